set -e
for wl in globalmcmc glmala gk nf kde callback glmcmc_nf aglmcmc rtc nf_train; do
  timeout -k 10 400 python bench.py --workload $wl > gpurun_out/r02_final_bench_$wl.json 2> gpurun_out/r02_final_bench_$wl.err || echo "FAILED $wl"
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r02_final_bench_$wl.json")); print("$wl", d["value"], d["unit"], d["ms_per_step"])
except Exception as e:
    print("$wl", "ERR", e)
PY
done
