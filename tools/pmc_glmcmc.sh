# PMC passes of the headline bench for one launch geometry: tools/pmc_glmcmc.sh <tag> [extra bench.py flags...]
# (outputs under gpurun_out/<tag>/; separate passes, no tracing alongside the counters)
set -e
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
args="--steps 3 --warmup 1 --no-cpu-baseline $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py $args > $out/trace.json 2> $out/trace.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d $out/pmc1 -- python3 bench.py $args > $out/pmc1.json 2> $out/pmc1.err
rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU --output-format csv -d $out/pmc2 -- python3 bench.py $args > $out/pmc2.json 2> $out/pmc2.err
python3 tools/pmc_summary.py $out/pmc1 sampler_kernel
python3 tools/pmc_summary.py $out/pmc2 sampler_kernel
find $out/trace -name "*kernel_stats.csv" | head -1 | xargs head -3
