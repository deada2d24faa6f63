# Every judged profile of a round on the GPU box (about 15 minutes): tools/profile_all.sh <tag>
# then, in the build container:  for each workload  python tools/summarise_workload.py <tag> <workload> r03 <kernel regex>
tag=$1
for wl in glmcmc glmala gk globalmcmc gamma nf nf_train; do
  bash tools/profile_workload.sh $tag $wl > gpurun_out/${tag}_profile_$wl.log 2>&1 || echo "FAILED profile $wl"
done
bash tools/profile_workload.sh ${tag}fast glmcmc --fast-math > gpurun_out/${tag}_profile_glmcmc_fast.log 2>&1 || echo "FAILED profile fast"
for wl in kde callback glmcmc_nf aglmcmc rtc; do
  timeout -k 10 500 python3 bench.py --workload $wl > gpurun_out/${tag}_bench_$wl.json 2> gpurun_out/${tag}_bench_$wl.err || echo "FAILED bench $wl"
done
for n in 131072 524288; do
  timeout -k 10 300 python3 bench.py --chains $n --steps 5 --no-cpu-baseline > gpurun_out/${tag}_bench_glmcmc_$n.json 2>/dev/null || echo "FAILED bench $n"
done
ls gpurun_out | grep "^$tag" | head -40
