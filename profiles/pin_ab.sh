mkdir -p gpurun_out/pin
for v in pin nopin pin nopin; do
  if [ $v = pin ]; then L=quandary_amd/csrc/libquandary_amd.so; else L=quandary_amd/csrc/lib_nopin.so; fi
  python profiles/with_lib.py $L bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-gradient 2>/dev/null | tail -1 > gpurun_out/pin/bench_$v.json
  python - <<P
import json
d=json.loads(open("gpurun_out/pin/bench_$v.json").read())
print("$v", "headline", round(d["ms_per_step"],1), " ".join(f'{w["n"]}/{w["m"]}/{w["sol"]}/{w["d"][:3]}:{w["ms"]}' for w in d["workloads"]))
P
done
