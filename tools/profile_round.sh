# Collect the judged profiles of one workload on the GPU box: tools/profile_round.sh <tag> <workload>
# (kernel trace + stats, then one PMC pass; outputs under gpurun_out/<tag>_<workload>/)
set -e
tag=$1; wl=$2; out=$GRAFT_REPO_ROOT/gpurun_out/${tag}_${wl}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 bench.py --workload $wl > $out/bench.json 2> $out/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --workload $wl --steps 5 --warmup 1 --no-cpu-baseline > $out/trace.json 2> $out/trace.err
if [ "$wl" = nf ] || [ "$wl" = nf_train ]; then ctr="SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_LDS"; else ctr="SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES"; fi
rocprofv3 --pmc $ctr --output-format csv -d $out/pmc -- python3 bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline > $out/pmc.json 2> $out/pmc.err
python3 tools/pmc_summary.py $out/pmc _kernel
find $out/trace -name "*kernel_stats.csv" | head -1 | xargs head -4
cat $out/bench.json
