"""Generates the kernel table of DESIGN.md section 4 from the COMMITTED measurement record - no hand-typed number:

    python profiles/design_table.py [round tag, default r6]      (rewrites the block between the GENERATED markers of DESIGN.md)

Sources: profiles/<tag>_bench_default.json (the default bench.py line of the record lease: headline, gradient, workloads),
profiles/pmc_latest.json (HBM bytes per unit from the PMC passes, FETCH_SIZE x 2 + WRITE_SIZE as the guide prescribes),
profiles/<tag>_<workload>_summary.json (rocprofv3 --kernel-trace --stats: per-launch durations of the dominant kernel),
profiles/<tag>_kres.txt (profiles/kres.sh on the objects of the same build: VGPRs, spilt registers, scratch bytes per lane)."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
TAG = sys.argv[1] if len(sys.argv) > 1 else "r6"
BEGIN, END = "<!-- BEGIN GENERATED KERNEL TABLE (profiles/design_table.py) -->", "<!-- END GENERATED KERNEL TABLE -->"


def load(name):
    try:
        return json.load(open(os.path.join(P, name)))
    except Exception:
        return None


def kres_table():
    out = {}
    try:
        for l in open(os.path.join(P, f"{TAG}_kres.txt")):
            m = re.match(r"(?:void )?(qd::\S.*?)\((?:qd::)?(?:SweepArgs|.*)\) vgpr (\d+) .*? vspill (\d+) sspill (\d+) lds \d+ scratch (\d+)", l.strip())
            if m:
                out[m.group(1).replace("void ", "")] = tuple(int(x) for x in m.groups()[1:])
    except Exception:
        pass
    return out


def sig(x, n=3):
    return "-" if x is None else f"{x:.{n}g}"


def main():
    bench = load(f"{TAG}_bench_default.json")
    if bench is None:
        raise SystemExit(f"profiles/{TAG}_bench_default.json not found")
    pmc = load("pmc_latest.json") or {}
    kres = kres_table()

    def prof(key):  # (summary of a profiled workload: its sweep kernels by total time, each with the mean and max / min of its timed launches)
        s = load(f"{TAG}_{key}_summary.json")
        if not s:
            return []
        ks = sorted(s.get("sweep_launches", {}).items(), key=lambda kv: -sum(kv[1]["durations_ms"]))
        return [(name.replace("void ", ""), d.get("timed_avg_ms", d.get("avg_ms_full_grid")), d.get("timed_max_over_min")) for name, d in ks[:2]]

    rows = []

    def add(label, key, v, kms, A, valu_spec, valu_meas, hbm, units_per_launch, alg_bytes):
        ent = pmc.get(key or "", {})
        traffic = ent.get("hbm_bytes_per_unit")
        # (the rocprofv3 columns only where the profile is of THIS command: a 250-step variant of the headline shares its kernels and its
        #  traffic per unit, not its launch durations)
        own = bool(key) and ent.get("units_per_launch") == units_per_launch
        ks = prof(key) if key else []
        names = " + ".join(f"`{k[0]}`" for k in ks) if ks else "-"
        durs = " + ".join(sig(k[1], 4) + (f" (max/min {k[2]:.3f})" if k[2] else "") for k in ks if k[1]) if (ks and own) else "-"
        regs = " ; ".join(f"{kres[k[0]][0]} / {kres[k[0]][1]} / {kres[k[0]][2]} / {kres[k[0]][3]} B" for k in ks if k[0] in kres) or "-"
        rows.append("| " + " | ".join([
            label, names, sig(kms, 4), durs, sig(v, 3), sig(A, 3), sig(valu_spec, 2), sig(valu_meas, 2), sig(hbm, 2),
            (f"{traffic / alg_bytes:.2g} ({traffic / 1e3:.3g} KB of {alg_bytes / 1e3:.3g} KB)" if traffic and alg_bytes else "-"), regs]) + " |")

    r = bench["roofline"]
    vk = "fp64_valu"
    add("**headline** C4 3x20 Lindblad 3600 x 2500, forward", "c4_fwd", bench["value"], r["kernel_ms_per_launch"], bench["config"]["rhs_applications_per_step"],
        r[vk]["frac"], r[vk]["frac_of_measured"], r["hbm"]["frac"], r["units_per_launch"], r["algorithmic_bytes_per_unit"])
    g = bench.get("gradient", {})
    if "roofline" in g:
        gr = g["roofline"]
        add("C4 gradient, full grid (chunks of initial conditions, one pass)", "c4_grad", g["value"], g["kernel_ms_per_evaluation"], g["rhs_applications_per_step"],
            gr[vk]["frac"], gr[vk]["frac_of_measured"], gr["hbm"]["frac"], gr["units_per_launch"], gr["algorithmic_bytes_per_unit"])
    names = {"c1": "C1 2x2 Schroedinger", "c2": "C2 2x2x2 Lindblad", "c3": "C3 2^4 Schroedinger", "c4": "C4 (250 / 500 steps)", "c5": "C5 2^5 Lindblad", "q4": "4-qubit open system",
             "q4j": "4-qubit open system, coupled", "c5j": "2^5 Lindblad, coupled", "l20": "20 x 20 Lindblad (one state, dim 160 000)", "n32": "32^4 Schroedinger (one state, dim 2^20)",
             "n4444": "4^4 Schroedinger"}
    for w in bench.get("workloads", []):
        if "v" not in w:
            continue
        opt = ", ".join(f"{k}={v}" for k, v in (w.get("o") or {}).items())
        label = f"{names.get(w['n'], w['n'])}, {'forward' if w['m'] == 'fwd' else 'gradient'}, {w['s']}" + (f", {w['d']}" if w["d"] != "f64" else "") + (f" ({opt})" if opt else "") + f" -> {w['sol']}"
        key = w["n"] + ("_f32" if w["d"] != "f64" else "") + ("_krylov" if w["sol"] == "krylov" else "") + "_" + w["m"]
        if w.get("o") and w["sol"] != "krylov":
            key = None  # (option variants of a workload share no profile)
        dim, real = w["dim"], 4 if w["d"] != "f64" else 8
        alg = (4 if w["m"] == "fwd" else 12) * real * dim
        cpu = f" [x {w['x']:.3g} of {w['cpu']['cores']} cores]" if "x" in w and "cpu" in w else ""
        add(label + cpu, key, w["v"], w["kms"], w["A"], None, w.get("valu"), w.get("hbm"), w["nt"] * w["ni"], alg)
    head = ("| workload -> solver path [GPU / CPU] | dominant kernel (rocprofv3) | kernel ms per evaluation (bench.py, HIP events) | rocprofv3 ms per launch (timed launches) | units/s | "
            "applications / step | valu frac of spec | of measured | hbm frac (algorithmic bytes) | PMC traffic / algorithmic | VGPR / vspill / sspill / scratch |\n|" + "---|" * 11)
    src = (f"Generated by `profiles/design_table.py {TAG}` from `profiles/{TAG}_bench_default.json` ({bench.get('n_gpus', 1)} GPU, value {bench['value']:.4g} units/s, "
           f"{bench['ms_per_step']:.1f} ms per step), `profiles/pmc_latest.json`, `profiles/{TAG}_*_summary.json`, `profiles/{TAG}_kres.txt`.  "
           "Fractions: canonical flops (SURVEY 8d) / kernel time / 78.6 TF fp64 (157.3 TF fp32-mixed) spec, resp. / the FMA rate measured on the device in the same run; "
           "rows without a spec fraction come from the compact `workloads` array (which carries the measured-rate fraction only).  "
           "These are the BUILDER's lease; the driver's `BENCH_rNN.json` is taken on another box of the pool - the headline moves by a few "
           "per cent from box to box (round 5: 584.0 ms in the builder's record, 610.8 ms on the driver's box).")
    block = BEGIN + "\n" + src + "\n\n" + head + "\n" + "\n".join(rows) + "\n" + END
    path = os.path.join(ROOT, "DESIGN.md")
    text = open(path).read()
    if BEGIN in text and END in text:
        text = text[:text.index(BEGIN)] + block + text[text.index(END) + len(END):]
        open(path, "w").write(text)
        print(f"DESIGN.md: table regenerated ({len(rows)} rows)")
    else:
        print(block)


if __name__ == "__main__":
    main()
