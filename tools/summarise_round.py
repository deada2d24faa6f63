"""Turn the output of tools/profile_round.sh (gpurun_out/<tag>_<workload>) into committed profiles/ files:
python tools/summarise_round.py <tag> <workload> <profiles prefix, e.g. r02> <kernel-name substring>"""
import collections, csv, glob, json, os, shutil, sys


def newest(pattern):
    return max(glob.glob(pattern, recursive=True), key=os.path.getmtime)


tag, wl, prefix, pat = sys.argv[1:5]
root = "gpurun_out/%s_%s" % (tag, wl)
acc, cnt = collections.defaultdict(float), collections.defaultdict(int)
for r in csv.DictReader(open(newest(root + "/pmc/**/*counter_collection.csv"))):
    if pat in r["Kernel_Name"]:
        acc[r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[r["Counter_Name"]] += 1
vals = {k: acc[k] / cnt[k] for k in acc}
stats_file = newest(root + "/trace/**/*kernel_stats.csv")
kern = [r for r in csv.DictReader(open(stats_file)) if pat in r["Name"]]
bench = json.loads(open(root + "/bench.json").read().strip().split("\n")[-1])
cfg = bench.get("config", {})
out = {"what": "rocprofv3 --pmc pass of `python3 bench.py --workload %s --steps 3 --warmup 1 --no-cpu-baseline` on MI355X; averages "
               "per launch of the kernels whose name contains '%s'" % (wl, pat),
       "round": 2, "workload": wl, "config": {"chains": cfg.get("chains_per_gpu"), "iters_per_launch": cfg.get("iters_per_step")},
       "raw_avg_per_launch": vals,
       "kernel_trace": [{"name": k["Name"][:120], "avg_ns": float(k["AverageNs"]), "calls": int(k["Calls"])} for k in kern]}
if "SQ_INSTS_VALU" in vals and cfg.get("chains_per_gpu") and cfg.get("iters_per_step"):
    waves, steps = cfg["chains_per_gpu"] / 64.0, float(cfg["iters_per_step"])
    out["derived"] = {"valu_insts_per_wave_step": vals["SQ_INSTS_VALU"] / (waves * steps),
                      "salu_insts_per_wave_step": vals.get("SQ_INSTS_SALU", 0.0) / (waves * steps)}
    if "SQ_WAVE_CYCLES" in vals and "SQ_ACTIVE_INST_VALU" in vals:
        wc, va = vals["SQ_WAVE_CYCLES"] * 4 / (waves * steps), vals["SQ_ACTIVE_INST_VALU"] * 4 / (waves * steps)
        out["derived"].update(wave_cycles_per_step=wc, valu_active_fraction=va / wc,
                              cycles_per_valu_inst=va / out["derived"]["valu_insts_per_wave_step"])
if "SQ_VALU_MFMA_BUSY_CYCLES" in vals and "SQ_BUSY_CYCLES" in vals:
    # per kernel: SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs, SQ_BUSY_CYCLES over the 32 shader engines
    per = {}
    acc2, cnt2 = collections.defaultdict(lambda: collections.defaultdict(float)), collections.defaultdict(lambda: collections.defaultdict(int))
    for r in csv.DictReader(open(newest(root + "/pmc/**/*counter_collection.csv"))):
        if pat in r["Kernel_Name"]:
            k = r["Kernel_Name"].split("(")[0]
            acc2[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt2[k][r["Counter_Name"]] += 1
    for k in acc2:
        v = {c: acc2[k][c] / cnt2[k][c] for c in acc2[k]}
        cycles = v["SQ_BUSY_CYCLES"] / 32.0
        dur = [x for x in kern if x["Name"].split("(")[0] == k]
        per[k] = {"mfma_insts": v["SQ_INSTS_MFMA"], "mfma_busy_fraction": v["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cycles),
                  "kernel_cycles": cycles, "clock_ghz": cycles / float(dur[0]["AverageNs"]) if dur else None,
                  "avg_ns_kernel_trace": float(dur[0]["AverageNs"]) if dur else None,
                  "tflops_mfma_from_trace": v["SQ_INSTS_MFMA"] * 4096.0 / float(dur[0]["AverageNs"]) / 1e3 if dur else None}
    out["derived"] = {"per_kernel": per,
                      "note": "v_mfma_f32_32x32x2_f32 = 4096 flop per wave-instruction, 64 cycles of a SIMD's matrix pipe; busy fraction = "
                              "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles), kernel cycles = SQ_BUSY_CYCLES / 32 shader engines"}
json.dump(out, open("profiles/%s_pmc_%s.json" % (prefix, wl), "w"), indent=1)
shutil.copy(stats_file, "profiles/%s_kernel_stats_%s.csv" % (prefix, wl))
shutil.copy(root + "/bench.json", "profiles/%s_bench_%s.json" % (prefix, wl))
print(json.dumps(out.get("derived"), indent=1), out["kernel_trace"][:3])
