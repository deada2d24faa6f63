"""Turn the output of tools/profile_workload.sh (gpurun_out/<tag>_<workload>) into the committed profiles/ files
    python tools/summarise_workload.py <tag> <workload> <prefix, e.g. r03> <kernel-name regex>
-> profiles/<prefix>_pmc_<workload>.json (+ _kernel_stats_<workload>.csv, _bench_<workload>.json).  The JSON carries the first
16 hex digits of the SHA-256 of libglabc_hip.so the counters were taken with: bench.py uses the instruction counts only while
the library it runs is that very build."""
import collections, csv, glob, json, os, re, shutil, sys


def newest(pattern):
    """gpurun merges every call's files into the same directories: keep the most recent one"""
    files = glob.glob(pattern, recursive=True)
    return max(files, key=os.path.getmtime) if files else None


tag, wl, prefix, pat = sys.argv[1:5]
name = sys.argv[5] if len(sys.argv) > 5 else wl            # name of the profiles/ files (e.g. glmcmc_fast for `glmcmc --fast-math`)
rx = re.compile(pat)
root = "gpurun_out/%s_%s" % (tag, wl)
vals, per_kernel = {}, collections.defaultdict(dict)
for d in ("pmc_sq", "pmc_mem_r", "pmc_mem_w", "pmc_mfma"):
    f = newest(root + "/" + d + "/**/*counter_collection.csv")
    if not f:
        continue
    acc, cnt = collections.defaultdict(float), collections.defaultdict(int)
    acck, cntk = collections.defaultdict(float), collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        if rx.search(r["Kernel_Name"]):
            acc[r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[r["Counter_Name"]] += 1
            k = (r["Kernel_Name"].split("(")[0], r["Counter_Name"])
            acck[k] += float(r["Counter_Value"])
            cntk[k] += 1
    for k in acc:
        vals[k] = acc[k] / cnt[k]
    for (kn, c) in acck:
        per_kernel[kn][c] = acck[(kn, c)] / cntk[(kn, c)]


def src_sha():
    """hash of the kernel sources the profiled library was built from (bench.src_sha16): written on the GPU box by
    tools/profile_workload.sh; for an older run, taken from this tree if its library is still the profiled one"""
    p = root + "/src_sha16.txt"
    if os.path.exists(p):
        return open(p).read().strip()
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    import bench
    if bench.lib_sha16() == open(root + "/lib_sha16.txt").read().strip():
        return bench.src_sha16()
    return None


stats_file = newest(root + "/trace/**/*kernel_stats.csv")
kern = [r for r in csv.DictReader(open(stats_file)) if rx.search(r["Name"])]
bench = json.loads(open(root + "/bench.json").read().strip().split("\n")[-1])
cfg = bench.get("config", {})
out = {"what": "rocprofv3 --pmc passes (one counter group per run) of `python3 bench.py --workload %s --steps 3 --warmup 1 "
               "--no-cpu-baseline` on MI355X; averages per launch of the kernels matching /%s/%s" % (wl, pat, " (profiles name: %s)" % name if name != wl else ""),
       "round": 3, "workload": wl, "lib_sha16": open(root + "/lib_sha16.txt").read().strip(), "src_sha16": src_sha(),
       "config": {"chains": cfg.get("chains_per_gpu"), "iters_per_launch": cfg.get("iters_per_step"), "batch_size": cfg.get("batch_size")},
       "raw_avg_per_launch": vals,
       "kernel_trace": [{"name": k["Name"][:140], "avg_ns": float(k["AverageNs"]), "min_ns": float(k["MinNs"]), "calls": int(k["Calls"])}
                        for k in kern]}
if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
    read, write = vals["FETCH_SIZE"] * 1024 * 2, vals["WRITE_SIZE"] * 1024
    out["hbm_traffic_bytes_per_launch"] = {"read": read, "write": write, "total": read + write,
                                           "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B read requests at "
                                                   "64 B); WRITE_SIZE exact; both in KiB"}
if "SQ_INSTS_VALU" in vals and cfg.get("chains_per_gpu") and cfg.get("iters_per_step") and kern:
    groups, steps = -(-cfg["chains_per_gpu"] // 64), float(cfg["iters_per_step"])
    main = max(kern, key=lambda k: float(k["AverageNs"]) * int(k["Calls"]))
    ns = float(main["AverageNs"])
    clock = vals["GRBM_GUI_ACTIVE"] / 8 / ns if "GRBM_GUI_ACTIVE" in vals else None      # GHz: 8 XCDs count the kernel's cycles
    valu = vals["SQ_INSTS_VALU"] / (groups * steps)
    out["derived"] = {
        "groups_of_64_chains": groups, "waves_per_launch": vals.get("SQ_WAVES"),
        "valu_insts_per_group_step": valu, "salu_insts_per_group_step": vals.get("SQ_INSTS_SALU", 0.0) / (groups * steps),
        "note_group": "one group = 64 chains = one workgroup; a team kernel spreads the group's instructions over 2 - 4 wavefronts",
        "kernel_ns_trace": ns, "clock_ghz_from_grbm": clock,
        # wall-based: SIMD cycles that passed per vector instruction issued on it (1024 SIMDs share the launch evenly)
        "simd_cycles_per_valu_inst": (ns * clock * 1024.0) / vals["SQ_INSTS_VALU"] if clock else None,
        "wave_cycles_per_wave_step": vals["SQ_WAVE_CYCLES"] * 4 / (vals["SQ_WAVES"] * steps) if vals.get("SQ_WAVES") else None,
        "sq_active_inst_valu_x4_per_inst": vals["SQ_ACTIVE_INST_VALU"] * 4 / vals["SQ_INSTS_VALU"] if "SQ_ACTIVE_INST_VALU" in vals else None,
        "note_active": "SQ_ACTIVE_INST_VALU x 4 / SQ_INSTS_VALU is 4.04 - 4.12 for every kernel and occupancy measured here (one, two, "
                       "three wavefronts per SIMD): the counter charges a wave64 instruction its 4 issue cycles, it is not a "
                       "throughput -- simd_cycles_per_valu_inst is"}
if "SQ_VALU_MFMA_BUSY_CYCLES" in vals:
    per = {}
    for kn, v in per_kernel.items():
        if "SQ_VALU_MFMA_BUSY_CYCLES" not in v or "SQ_BUSY_CYCLES" not in v:
            continue
        cycles = v["SQ_BUSY_CYCLES"] / 32.0
        dur = [x for x in kern if x["Name"].split("(")[0] == kn]
        per[kn] = {"mfma_insts": v.get("SQ_INSTS_MFMA"), "mfma_busy_fraction": v["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cycles),
                   "kernel_cycles": cycles, "avg_ns_kernel_trace": float(dur[0]["AverageNs"]) if dur else None,
                   "clock_ghz": cycles / float(dur[0]["AverageNs"]) if dur else None,
                   "tflops_mfma_from_trace": v["SQ_INSTS_MFMA"] * 4096.0 / float(dur[0]["AverageNs"]) / 1e3 if dur and v.get("SQ_INSTS_MFMA") else None,
                   "lds_insts": v.get("SQ_INSTS_LDS"), "wait_inst_lds": v.get("SQ_WAIT_INST_LDS")}
    out["mfma"] = {"per_kernel": per,
                   "note": "v_mfma_f32_32x32x2_f32 = 4096 flop per wave-instruction, 64 cycles of a SIMD's matrix pipe; busy fraction = "
                           "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles), kernel cycles = SQ_BUSY_CYCLES / 32 shader engines"}
json.dump(out, open("profiles/%s_pmc_%s.json" % (prefix, name), "w"), indent=1)
shutil.copy(stats_file, "profiles/%s_kernel_stats_%s.csv" % (prefix, name))
shutil.copy(root + "/bench.json", "profiles/%s_bench_%s.json" % (prefix, name))
print(json.dumps(out.get("derived"), indent=1), json.dumps(out.get("mfma"), indent=1)[:1500], out["kernel_trace"][:3])
print("bench:", bench["value"], bench["unit"], bench["ms_per_step"])
