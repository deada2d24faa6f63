for ch in 65536 131072 262144 524288; do
  echo "== chains $ch"
  timeout -k 10 200 python bench.py --chains $ch --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('%.4g chain-steps/s kernel %.3f ms' % (j['value'], j['roofline']['kernel_ms']))" || exit 1
done
for wl in globalmcmc glmala gk; do
  it=2000; [ $wl = glmala ] && it=100; [ $wl = gk ] && it=500
  echo "== $wl"; timeout -k 10 300 python bench.py --workload $wl --iters $it --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('%.4g %s kernel %.3f ms' % (j['value'], j['unit'], j['roofline']['kernel_ms']))" || exit 1
done
