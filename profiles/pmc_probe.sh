#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): two rocprofv3 PMC passes (SQ counters only, no tracing domains besides --kernel-trace) of an
# arbitrary command; prints / stores the per-kernel sums.  usage: profiles/pmc_probe.sh <tag> <command...>
set -u
TAG=$1; shift
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/p1 -- "$@" > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_WAVES --output-format csv -d $OUT/p2 -- "$@" > $OUT/p2.log 2>&1
cd $REPO
python - $OUT $TAG <<'PY'
import csv, glob, json, os, sys
out, tag = sys.argv[1], sys.argv[2]
res = {}
for sub in ("p1", "p2"):
    for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if "k_forward" not in k and "k_adjoint" not in k:
                continue
            d = res.setdefault(k, {})
            e = d.setdefault(r["Counter_Name"], [0.0, 0])
            e[0] += float(r["Counter_Value"]); e[1] += 1
summary = {k: {c: v[0] / max(v[1], 1) for c, v in d.items()} | {"launches": max(v[1] for v in d.values())} for k, d in res.items()}
json.dump(summary, open(os.path.join(out, f"{tag}_pmc.json"), "w"), indent=1)
for k, d in summary.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:24s} {v:.4g}")
PY
