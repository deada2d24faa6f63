# A part of tools/profile_all.sh (a gpurun call is limited to 20 minutes): tools/profile_part.sh <tag> <workload> [<workload> ...]
# workloads: glmcmc glmala gk globalmcmc gamma nf nf_train | fast (glmcmc --fast-math) | benches (the lines without counters)
tag=$1; shift
for wl in "$@"; do
  case $wl in
    fast)
      bash tools/profile_workload.sh ${tag}fast glmcmc --fast-math > gpurun_out/${tag}_profile_glmcmc_fast.log 2>&1 || echo "FAILED profile fast" ;;
    benches)
      for b in kde callback glmcmc_nf aglmcmc rtc; do
        timeout -k 10 500 python3 bench.py --workload $b > gpurun_out/${tag}_bench_$b.json 2> gpurun_out/${tag}_bench_$b.err || echo "FAILED bench $b"
      done
      for n in 131072 524288; do
        timeout -k 10 300 python3 bench.py --chains $n --steps 5 --no-cpu-baseline > gpurun_out/${tag}_bench_glmcmc_$n.json 2>/dev/null || echo "FAILED bench $n"
      done ;;
    *)
      bash tools/profile_workload.sh $tag $wl > gpurun_out/${tag}_profile_$wl.log 2>&1 || echo "FAILED profile $wl" ;;
  esac
  echo "done $wl"
done
