# Where the wavefronts of the NF kernels spend their cycles (parked at a barrier / waitcnt, stalled at issue, issuing):
#   tools/pmc_nf_wait.sh <tag> [workload=nf]   -> gpurun_out/<tag>_nfwait/{pmc_a,pmc_b}/ + wait.txt
# One counter group per run, no tracing alongside the counters.
set -e
tag=$1; wl=${2:-nf}
out=$GRAFT_REPO_ROOT/gpurun_out/${tag}_nfwait
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
short="--workload $wl --steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_a -- python3 bench.py $short > /dev/null 2> $out/pmc_a.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_b -- python3 bench.py $short > /dev/null 2> $out/pmc_b.err
python3 - $out <<'PY' > $out/wait.txt
import glob, sys, pandas as pd
out = sys.argv[1]
for grp in ("pmc_a", "pmc_b"):
    fs = glob.glob(out + "/" + grp + "/**/*counter_collection.csv", recursive=True)
    df = pd.concat([pd.read_csv(f) for f in fs])
    df = df[df["Kernel_Name"].str.contains("nf_")]
    t = df.pivot_table(index=["Kernel_Name", "Dispatch_Id"], columns="Counter_Name", values="Counter_Value", aggfunc="sum").reset_index()
    print(t.groupby("Kernel_Name").mean(numeric_only=True).drop(columns=["Dispatch_Id"]).T.to_string())
PY
cat $out/wait.txt
