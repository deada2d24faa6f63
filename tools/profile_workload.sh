# Collect the judged profiles of one bench.py workload on the GPU box:
#   tools/profile_workload.sh <tag> <workload> [extra bench.py flags ...]
# -> gpurun_out/<tag>_<workload>/{bench.json, trace/, pmc_sq/, pmc_mem_r/, pmc_mem_w/, pmc_mfma/}
# One counter group per run, no tracing alongside the counters (MI355X_MICROARCH.md; gpurun refuses the combination).
set -e
tag=$1; wl=$2; shift 2
out=$GRAFT_REPO_ROOT/gpurun_out/${tag}_${wl}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 bench.py --workload $wl "$@" > $out/bench.json 2> $out/bench.err
short="--workload $wl --steps 3 --warmup 1 --no-cpu-baseline $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --workload $wl --no-cpu-baseline "$@" > $out/trace.json 2> $out/trace.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_sq -- python3 bench.py $short > $out/pmc_sq.json 2> $out/pmc_sq.err
case $wl in
  glmcmc|gk|globalmcmc|glmala|gamma)
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_mem_r -- python3 bench.py $short > /dev/null 2> $out/pmc_mem_r.err
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_mem_w -- python3 bench.py $short > /dev/null 2> $out/pmc_mem_w.err ;;
  nf|nf_train|glmcmc_nf)
    rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_LDS --output-format csv -d $out/pmc_mfma -- python3 bench.py $short > /dev/null 2> $out/pmc_mfma.err ;;
esac
sha256sum gl-abc-mcmc_amd/csrc/libglabc_hip.so | cut -c1-16 > $out/lib_sha16.txt
python3 -c "import bench; print(bench.src_sha16())" > $out/src_sha16.txt
tail -c 400 $out/bench.json
