"""Exposed memory latency in the loops of the kernels of an object file: every `s_waitcnt vmcnt(n)` of a loop body that (in static order)
forces a global LOAD issued fewer than DIST instructions earlier to return.  The vector-memory counter retires in order, so `vmcnt(n)` waits
for all but the youngest n operations (loads and stores alike) - a wait that the compiler keeps for a loop-carried register (filled by a load
before the loop, by arithmetic inside it) also drains a prefetch issued just before it.

    python profiles/wait_scan.py quandary_amd/csrc/build/qd_inst_4_0_1_0.o [name-pattern] [DIST=60]
"""
import re
import subprocess
import sys

sys.path.insert(0, __import__("os").path.dirname(__file__))
from isa_mix import disassemble  # noqa: E402


def kernels(lines):
    cur, name = None, None
    for l in lines:
        m = re.match(r"^[0-9a-f]+ <(.*)>:$", l)
        if m:
            if cur:
                yield name, cur
            name, cur = m.group(1), []
            continue
        m = re.match(r"\s+(\S.*?)\s*//\s*([0-9A-F]+):", l)
        if m and cur is not None:
            cur.append((int(m.group(2), 16), m.group(1)))
    if cur:
        yield name, cur


def scan(name, ins, dist):
    index = {a: i for i, (a, _) in enumerate(ins)}
    loops = set()
    for i, (a, t) in enumerate(ins):
        if t.startswith(("s_cbranch", "s_branch")):
            off = int(t.split()[-1])
            off = off - 65536 if off >= 32768 else off
            tgt = a + 4 + 4 * off
            if tgt < a and tgt in index and i - index[tgt] >= 40:
                loops.add((index[tgt], i))
    found = []
    for lo, hi in sorted(loops):
        vm = []  # (instruction index, is_load)
        for i in range(lo, hi + 1):
            t = ins[i][1]
            op = t.split()[0]
            if op.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")):
                vm.append((i, True))
            elif op.startswith(("global_store", "buffer_store", "flat_store", "global_atomic", "scratch_store")):
                vm.append((i, False))
            elif op == "s_waitcnt":
                m = re.search(r"vmcnt\((\d+)\)", t)
                if m:
                    n = int(m.group(1))
                    must = vm[:len(vm) - n] if n else vm
                    near = [i - j for j, ld in must if ld and i - j < dist]
                    if near:
                        found.append((hi - lo + 1, i - lo, n, min(near), len(near)))
                    vm = vm[len(vm) - n:] if n else []
    if found:
        dem = subprocess.check_output(["c++filt", name], text=True).strip()
        print(dem[:150])
        for f in sorted(set(found)):
            print("    loop of %d instr: at +%d s_waitcnt vmcnt(%d) drains %d load(s) issued as little as %d instructions earlier" % (f[0], f[1], f[2], f[4], f[3]))


def main():
    obj = sys.argv[1]
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    dist = int(sys.argv[3]) if len(sys.argv) > 3 else 60
    for name, ins in kernels(disassemble(obj)):
        if pat in name and len(ins) > 100:
            scan(name, ins, dist)


if __name__ == "__main__":
    main()
