"""Turn the output of tools/profile_headline.sh (gpurun_out/<tag>_glmcmc) into the committed profiles/ files:
python tools/summarise_headline.py <tag> <profiles prefix, e.g. r01_f>"""
import collections, csv, glob, json, os, shutil, sys


def newest(pattern):
    """gpurun merges every call's files into the same directories: keep the most recent one"""
    return max(glob.glob(pattern, recursive=True), key=os.path.getmtime)


tag, prefix = sys.argv[1], sys.argv[2]
root = "gpurun_out/%s_glmcmc" % tag
vals = {}
for d in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2"):
    acc, cnt = collections.defaultdict(float), collections.defaultdict(int)
    for f in [newest(root + "/" + d + "/**/*counter_collection.csv")]:
        for r in csv.DictReader(open(f)):
            if "sampler_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"])
                cnt[r["Counter_Name"]] += 1
    for k in acc:
        vals[k] = acc[k] / cnt[k]
waves, steps = 1024.0, 2000.0
read, write = vals["FETCH_SIZE"] * 1024 * 2, vals["WRITE_SIZE"] * 1024
valu, salu = vals["SQ_INSTS_VALU"] / (waves * steps), vals["SQ_INSTS_SALU"] / (waves * steps)
wc, va = vals["SQ_WAVE_CYCLES"] * 4 / (waves * steps), vals["SQ_ACTIVE_INST_VALU"] * 4 / (waves * steps)
stats_file = newest(root + "/trace/**/*kernel_stats.csv")
kern = [r for r in csv.DictReader(open(stats_file)) if "sampler_kernel" in r["Name"]][0]
tr = json.loads(open(root + "/trace.json").read().strip().split("\n")[-1])
out = {
    "what": "rocprofv3 --pmc passes (one counter group per run) of `python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline` on "
            "MI355X; averages per launch of glabc::sampler_kernel<GLMCMC, D=2, YD=2, N=5, L=1, VAR_GAUSS_UNIT, SCHED_ILP> "
            "(65536 chains x 2000 iterations, history on)",
    "round": 2, "config": {"chains": 65536, "iters_per_launch": 2000, "history": True},
    "raw_avg_per_launch": vals,
    "hbm_traffic_bytes_per_launch": {"read": read, "write": write, "total": read + write,
                                     "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B read requests at 64 B); "
                                             "WRITE_SIZE exact; both in KiB"},
    "kernel_trace": {"avg_ns": float(kern["AverageNs"]), "min_ns": float(kern["MinNs"]), "max_ns": float(kern["MaxNs"]),
                     "calls": int(kern["Calls"]), "hip_event_ms_in_the_same_run": tr["roofline"]["kernel_ms"],
                     "source": "profiles/%s_kernel_stats_glmcmc.csv (rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 "
                               "--warmup 1 --no-cpu-baseline)" % prefix},
    "derived": {"valu_insts_per_wave_step": valu, "salu_insts_per_wave_step": salu, "wave_cycles_per_step": wc,
                "valu_active_cycles_per_step": va, "valu_active_fraction": va / wc, "cycles_per_valu_inst": va / valu,
                "clock_ghz_from_grbm": vals["GRBM_GUI_ACTIVE"] / 8 / (float(kern["AverageNs"]) * 1e-9) / 1e9}}
json.dump(out, open("profiles/%s_pmc_summary.json" % prefix, "w"), indent=1)
shutil.copy(stats_file, "profiles/%s_kernel_stats_glmcmc.csv" % prefix)
shutil.copy(root + "/bench.json", "profiles/%s_bench_glmcmc.json" % prefix)
b = json.loads(open(root + "/bench.json").read().strip().split("\n")[-1])
print(json.dumps(out["derived"], indent=1), out["hbm_traffic_bytes_per_launch"]["total"], out["kernel_trace"])
print("bench:", b["value"], b["ms_per_step"], b["roofline"]["kernel_ms"], b["roofline"]["frac"])
