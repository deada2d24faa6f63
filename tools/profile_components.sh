# Per-component instruction table of the headline kernel on the GPU box: tools/profile_components.sh <tag>
# One rocprofv3 --pmc pass (SQ counters only) per variant of `bench.py`: full step, no history rows, no streaming sums,
# neither, and batch size 1 (one candidate per iteration).  Outputs under gpurun_out/<tag>_components/<variant>/.
set -e
tag=$1; out=$GRAFT_REPO_ROOT/gpurun_out/${tag}_components
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() {
  name=$1; shift
  python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline "$@" > $out/$name.bench.json 2> $out/$name.err
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $out/$name -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2>> $out/$name.err
  echo "== $name"; python3 tools/pmc_summary.py $out/$name sampler_kernel
}
run full
run no_history --no-history
run no_moments --no-moments
run bare --no-history --no-moments
run batch1 --batch 1
run batch16 --batch 16
