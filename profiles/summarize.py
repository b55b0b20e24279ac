"""Condenses a gpurun_out/prof_<tag>/ directory written by profiles/collect.sh into small files
that are committed under profiles/: per-kernel stats (rocprofv3 --stats) and per-launch HBM bytes
from the PMC passes (FETCH_SIZE / WRITE_SIZE are reported in KiB-like units of 1024 B;
MI355X_MICROARCH.md: on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x, other widths
uncalibrated -> both the raw and the doubled figure are kept)."""
import csv
import glob
import json
import os
import sys


def find(d, suffix):
    """newest matching file only: gpurun merges every call's output into the same directory"""
    fs = glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True)
    return [max(fs, key=os.path.getmtime)] if fs else []


def main():
    out, tag = sys.argv[1], sys.argv[2]
    res = {"tag": tag}
    stats = find(os.path.join(out, "stats"), "kernel_stats.csv")
    rows = []
    for f in stats:
        rows += list(csv.DictReader(open(f)))
    res["kernel_stats"] = rows
    with open(os.path.join(out, f"{tag}_kernel_stats.csv"), "w") as fh:
        if rows:
            w = csv.DictWriter(fh, fieldnames=list(rows[0].keys()))
            w.writeheader()
            w.writerows(rows)
    # every launch of the sweep kernels (kernel-trace): the bench command also launches them once on the small sample of its oracle
    # check (fewer workgroups, fewer steps), which pulls the --stats average below the duration of the timed launches;
    # `avg_ms_full_grid` averages the launches with the largest grid only and is the figure to hold against bench.py's
    # roofline.kernel_ms_per_launch.
    launches = {}
    for f in find(os.path.join(out, "stats"), "kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            if "k_forward" not in name and "k_adjoint" not in name:
                continue
            d = launches.setdefault(name.split("(")[0], {"durations_ms": [], "grid_x": []})
            d["durations_ms"].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e6)
            d["grid_x"].append(int(r["Grid_Size_X"]))
    for d in launches.values():
        gmax = max(d["grid_x"])
        full = [t for t, g in zip(d["durations_ms"], d["grid_x"]) if g == gmax]
        full = [t for t in full if t >= 0.5 * max(full)]  # (a one-state workload: the check launch has the full grid but few steps)
        d["avg_ms_full_grid"] = sum(full) / len(full)
        d["n_full_grid"] = len(full)
        d["total_ms_full_grid"] = sum(full)  # (a chunked gradient evaluation launches each sweep once per chunk: divide by the evaluations)
    # [r6] the TIMED launches of a one-launch-per-evaluation command: the last `steps` full-grid launches in front of the oracle check's small
    # one (the warm-up launches of a gmres run tune the preconditioner's degree and differ from one another) - min, max and mean of those
    try:
        line = json.loads([l for l in open(os.path.join(out, "bench_stats.log")) if l.startswith("{")][-1])
        steps = int(line["steps"])
        for d in launches.values():
            gmax = max(d["grid_x"])
            full = [t for t, g in zip(d["durations_ms"], d["grid_x"]) if g == gmax]
            if len(full) >= steps and len(full) <= steps + int(line["warmup"]) + 1:  # (one launch per evaluation)
                timed = full[-steps:] if len(full) == steps + int(line["warmup"]) else full[-steps - 1:-1]
                d["timed_launches_ms"] = timed
                d["timed_min_ms"], d["timed_max_ms"], d["timed_avg_ms"] = min(timed), max(timed), sum(timed) / len(timed)
                d["timed_max_over_min"] = max(timed) / min(timed)
    except Exception:
        pass
    res["sweep_launches"] = launches
    pmc = {}
    per_dispatch = {}  # [r6] kernel -> counter -> [(dispatch id, grid size, value)]: the TIMED launches of a command are the last `steps` ones
    for name, sub in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
        for f in find(os.path.join(out, sub), "counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                if r.get("Counter_Name") != name:
                    continue
                k = r["Kernel_Name"].split("(")[0]
                d = pmc.setdefault(k, {}).setdefault(name, [])
                d.append(float(r["Counter_Value"]))
                per_dispatch.setdefault(k, {}).setdefault(name, []).append((int(r["Dispatch_Id"]), int(r.get("Grid_Size", 0) or 0), float(r["Counter_Value"])))
    sq = {}
    for f in find(os.path.join(out, "pmc_sq"), "counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            sq.setdefault(k, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    summary = {}
    for k, v in pmc.items():
        f = v.get("FETCH_SIZE", [])
        w = v.get("WRITE_SIZE", [])
        fm = sum(f) / len(f) if f else None
        wm = sum(w) / len(w) if w else None
        summary[k] = {
            "launches": max(len(f), len(w)),
            "hbm_bytes_all_launches_fetch_x2": (2 * sum(f) + sum(w)) * 1024.0,
            "FETCH_SIZE_mean": fm, "WRITE_SIZE_mean": wm,
            "hbm_bytes_per_launch_raw": ((fm or 0) + (wm or 0)) * 1024.0,
            "hbm_bytes_per_launch_fetch_x2": (2 * (fm or 0) + (wm or 0)) * 1024.0,
        }
    for k, v in sq.items():
        summary.setdefault(k, {})["sq_mean"] = {c: sum(x) / len(x) for c, x in v.items()}
    res["pmc"] = summary
    # per-launch HBM bytes of the sweep kernels -> profiles/pmc_latest.json (read by bench.py for
    # roofline.traffic); key = "<workload>_<mode>", taken from a tag like r1_c2_fwd
    parts = tag.split("_")
    if len(parts) >= 3:
        key = "_".join(parts[1:])  # r2_c5_f32_fwd -> c5_f32_fwd
        tot, names = 0.0, []
        units = evals = None  # units (time steps x initial conditions) per evaluation of the profiled command, evaluations run
        try:
            line = json.loads([l for l in open(os.path.join(out, "bench_pmc_fetch.log")) if l.startswith("{")][-1])
            units = line["roofline"]["units_per_launch"]
            evals = line["steps"] + line["warmup"]
        except Exception:
            pass
        steps = None
        try:
            steps = int(line["steps"])
        except Exception:
            pass

        def timed_mean(k, counter):
            """mean over the TIMED launches of kernel k, or None where the kernel does not run once per evaluation (chunked gradient)"""
            v = sorted(per_dispatch.get(k, {}).get(counter, []))
            if not v or not steps or not evals:
                return None
            gmax = max(g for _, g, _ in v)
            full = [x for _, g, x in v if g == gmax]
            if len(full) == evals:        # (the oracle check's launch has a smaller grid)
                return sum(full[-steps:]) / steps
            if len(full) == evals + 1:    # (a one-state workload: the check launch has the full grid too)
                return sum(full[-steps - 1:-1]) / steps
            return None

        timed_only = True
        for k, v in summary.items():
            if ("k_forward" in k or "k_adjoint" in k) and "hbm_bytes_per_launch_fetch_x2" in v:
                # bytes per EVALUATION.  [r6] Where the kernel runs once per evaluation: the TIMED launches only (the warm-up launches of a
                # gmres run tune the preconditioner's degree - generic-path solves with their scratch vectors - and are not what is timed).
                # Otherwise all launches of the run (a chunked gradient launches each sweep once per chunk; the one small launch of
                # bench.py's oracle check is in the sum as well) over the evaluations of the run.
                tf, tw = timed_mean(k, "FETCH_SIZE"), timed_mean(k, "WRITE_SIZE")
                if tf is not None and tw is not None:
                    tot += (2.0 * tf + tw) * 1024.0
                    v["hbm_bytes_per_timed_launch_fetch_x2"] = (2.0 * tf + tw) * 1024.0
                else:
                    timed_only = False
                    tot += v["hbm_bytes_all_launches_fetch_x2"] / evals if evals else v["hbm_bytes_per_launch_fetch_x2"]
                names.append(k[:60])
        if names:
            latest_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pmc_latest.json")
            latest = json.load(open(latest_path)) if os.path.exists(latest_path) else {}
            import datetime
            from srchash import csrc_hash
            latest[key] = {"csrc_hash": csrc_hash(), "date": datetime.date.today().isoformat(),
                           "hbm_bytes_per_launch": tot, "units_per_launch": units,
                           "hbm_bytes_per_unit": (tot / units) if units else None, "kernels": names, "source": f"profiles/{tag}_summary.json",
                           "launches_counted": "timed launches" if timed_only else "all launches of the run / evaluations",
                           "correction": "FETCH_SIZE x2 (gfx950, calibrated on this kernel's 8-byte loads) + WRITE_SIZE, units of 1 KiB"}
            json.dump(latest, open(latest_path, "w"), indent=1)
            # gpurun only merges gpurun_out/ back: leave a copy there
            json.dump(latest, open(os.path.join(out, "pmc_latest.json"), "w"), indent=1)
    json.dump(res, open(os.path.join(out, f"{tag}_summary.json"), "w"), indent=1)
    print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk != "sq_mean"} for k, v in summary.items()}, indent=1)[:3000])
    for r in rows[:8]:
        print({k: r[k] for k in list(r.keys())[:6]})


if __name__ == "__main__":
    main()
