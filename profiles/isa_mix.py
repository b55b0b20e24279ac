"""Instruction mix of the loops of a kernel, from the disassembly of the gfx950 code object inside a hipcc object file.

    python profiles/isa_mix.py quandary_amd/csrc/build/qd_col.o 'k_forward_colILi2ELi5ELb1ELb1' [min_len]

Lists every natural loop (backward branch) of the kernel whose body has at least min_len instructions (default 150) with its static
instruction count by class: fp64 arithmetic, other VALU (integer / address / moves / fp32 / lane reads), DPP moves, LDS, global / scratch
memory, scalar, barriers.  The solver iteration of a sweep kernel is the innermost loop with exactly one s_barrier; the time-step loop is
the one around it.  Static counts are executed counts for the straight-line solver iteration; the step loop contains conditional parts
(trajectory stores, the 1 / (1 - alpha D) table of a changed step size) that are not executed on every pass."""
import collections
import os
import re
import subprocess
import sys
import tempfile

L = "/opt/rocm/lib/llvm/bin"


def disassemble(obj):
    t = tempfile.mkdtemp()
    subprocess.check_call([f"{L}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, f"{t}/fat.bin"])
    for trg in ("hipv4-amdgcn-amd-amdhsa--gfx950", "hip-amdgcn-amd-amdhsa--gfx950"):
        r = subprocess.run([f"{L}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={t}/fat.bin", f"--targets={trg}", f"--output={t}/dev.co"],
                           capture_output=True)
        if r.returncode == 0 and os.path.getsize(f"{t}/dev.co") > 0:
            break
    return subprocess.check_output([f"{L}/llvm-objdump", "-d", f"{t}/dev.co"], text=True).splitlines()


def classify(op):
    if op.startswith("v_") and op.endswith("_dpp"):
        return "dpp"
    if re.match(r"v_(fma|fmac|mul|add|min|max|rcp|div_scale|div_fmas|div_fixup|rsq|sqrt|trig_preop|ldexp|frexp|cvt_f64)\w*_f64", op) or op.startswith("v_mfma_f64"):
        return "fp64"
    if op.startswith("v_"):
        return "valu_other"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "flat_", "buffer_")):
        return "global"
    if op.startswith("scratch_"):
        return "scratch"
    if op == "s_barrier":
        return "barrier"
    if op == "s_waitcnt" or op == "s_nop":
        return "wait"
    return "scalar"


def main():
    obj, pat = sys.argv[1], sys.argv[2]
    min_len = int(sys.argv[3]) if len(sys.argv) > 3 else 150
    lines = disassemble(obj)
    start = next(i for i, l in enumerate(lines) if re.match(r"^[0-9a-f]+ <.*" + re.escape(pat) + r".*>:$", l))
    print("kernel:", subprocess.check_output(["c++filt", re.search(r"<(.*)>", lines[start]).group(1)], text=True).strip())
    ins = []
    for l in lines[start + 1:]:
        if re.match(r"^[0-9a-f]+ <.*>:$", l):
            break
        m = re.match(r"\s+(\S.*?)\s*//\s*([0-9A-F]+):", l)
        if m:
            ins.append((int(m.group(2), 16), m.group(1)))
    index = {a: i for i, (a, _) in enumerate(ins)}
    loops = []
    for i, (a, t) in enumerate(ins):
        if t.startswith(("s_cbranch", "s_branch")):
            off = int(t.split()[-1])
            off = off - 65536 if off >= 32768 else off
            tgt = a + 4 + 4 * off
            if tgt < a and tgt in index and i - index[tgt] + 1 >= min_len:
                loops.append((index[tgt], i))
    seen = set()
    for lo, hi in sorted(loops, key=lambda x: x[1] - x[0]):
        if any(abs(lo - a) < 8 and abs(hi - b) < 8 for a, b in seen):
            continue  # (several exits of one loop)
        seen.add((lo, hi))
        c = collections.Counter(classify(t.split()[0]) for _, t in ins[lo:hi + 1])
        ops = collections.Counter(t.split()[0] for _, t in ins[lo:hi + 1] if classify(t.split()[0]) in ("valu_other", "dpp"))
        n = hi - lo + 1
        valu = c["fp64"] + c["valu_other"] + c["dpp"]
        print(f"loop of {n} instructions: fp64 {c['fp64']}, other VALU {c['valu_other']}, DPP {c['dpp']}, LDS {c['lds']}, global {c['global']}, "
              f"scratch {c['scratch']}, scalar {c['scalar']}, waits {c['wait']}, barriers {c['barrier']}  | VALU total {valu}, fp64 share of VALU {c['fp64'] / max(valu, 1):.2f}")
        print("    other VALU by opcode:", ", ".join(f"{k} {v}" for k, v in ops.most_common(12)))


if __name__ == "__main__":
    main()
