# Collect the judged profiles of the default bench.py run on the GPU box: tools/profile_headline.sh <tag>
# kernel trace + stats, then three PMC passes (HBM read, HBM write, SQ) as MI355X_MICROARCH.md prescribes (separate runs)
set -e
tag=$1; out=$GRAFT_REPO_ROOT/gpurun_out/${tag}_glmcmc
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 bench.py > $out/bench.json 2> $out/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > $out/trace.json 2> $out/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $out/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $out/pmc_write.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $out/pmc_sq -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $out/pmc_sq.err
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY --output-format csv -d $out/pmc_sq2 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $out/pmc_sq2.err
for d in pmc_fetch pmc_write pmc_sq pmc_sq2; do python3 tools/pmc_summary.py $out/$d sampler_kernel; done
find $out/trace -name "*kernel_stats.csv" | head -1 | xargs head -3
tail -c 600 $out/bench.json
