for v in exp_v0_select_newlog exp_v1_ifselect_newlog exp_v2_select_oldlog exp_v3_ifselect_oldlog; do
  echo "== $v"
  for i in 1 2; do GLABC_HIP_LIB=$PWD/gl-abc-mcmc_amd/csrc/$v.so timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(j['value'], j['roofline']['kernel_ms'])" || exit 1; done
done
