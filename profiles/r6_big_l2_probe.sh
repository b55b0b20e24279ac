#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): L2 (TCC) hit / miss and fabric request counters of the global-memory sweeps (qd_big.h) on the
# reference's nlevels_32_32_32_32 case (n32) and the 20 x 20 Lindblad state (l20) - separate PMC passes, --kernel-trace only.
# usage: profiles/r6_big_l2_probe.sh  ->  gpurun_out/r6_big_l2_probe.json
set -u
export TMPDIR=/tmp
REPO=$PWD
OUT=$REPO/gpurun_out/big_l2
mkdir -p $OUT
cd /tmp
for W in n32 l20; do
  i=0
  for C in "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_sum" "TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/${W}_p$i -- python $REPO/bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline --no-workloads --no-gradient > $OUT/${W}_p$i.log 2>&1
  done
done
cd $REPO
python - $OUT <<'PY'
import csv, glob, json, os, sys
out = sys.argv[1]
res = {}
for w in ("n32", "l20"):
    acc = {}
    for f in glob.glob(os.path.join(out, w + "_p*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if "k_forward_big" not in k:
                continue
            e = acc.setdefault(r["Counter_Name"], [])
            e.append((int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0), float(r["Counter_Value"])))
    summ = {}
    for c, v in acc.items():
        gmax = max(g for g, _ in v)
        full = [x for g, x in v if g == gmax]
        full = [x for x in full if x >= 0.5 * max(full)]  # (the timed launches: the check launch has the full grid but few steps)
        summ[c] = sum(full) / len(full)
    res[w] = summ
json.dump(res, open(os.path.join(os.path.dirname(out), "r6_big_l2_probe.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
PY
