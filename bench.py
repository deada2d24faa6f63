#!/usr/bin/env python3
"""bench.py -- BASELINE.json's headline metric on MI355X.

Workload (BASELINE.json configs[1]): GLMCMC, iSIR batch N=5, global_frequency 0.9, on
Mixture_set(eps=0.05), theta_dim 2, 65 536 independent chains per GPU, float32, synthetic
initial states, Philox random stream; the reference's example proposals
(examples/Mixture.py:67-69).  One bench "step" = ONE fused kernel launch that advances
every chain of this GPU by --iters MH iterations, writes every iteration's Theta_Re row
(the reference records every iteration, GLMCMC.py:89,104) and streams the ESJD / moment
sums.  value = chain-iterations per second over all GPUs ("MH accept-steps/sec").

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Multi-GPU: one process per GPU, chains sharded by global chain id (rank r owns ids
[r*C, (r+1)*C)), no data-path collective; one RCCL all-gather of the per-chain ESJD /
moment sums at the end of the timed region (weak scaling: C chains per GPU).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(ROOT, "gl-abc-mcmc_amd"), os.path.join(ROOT, "tests")]

import numpy as np          # noqa: E402
import torch                # noqa: E402
import torch.distributed as dist   # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md); 6290 GB/s measured copy rate
EPS, GF, NBATCH, D = 0.05, 0.9, 5, 2


def descriptors(workload="glmcmc"):
    from glabcmcmc_amd import distribution
    from glabcmcmc_amd.examples.Mixture import Mixture_set
    if workload == "gk":            # BASELINE configs[3]: g-and-k, theta_dim 4, Uniform(0,10)^4 prior and importance proposal
        from glabcmcmc_amd.examples.GK import GK_set
        model = GK_set(0.6).descriptor()
        lp = distribution.DiagGaussian(4, torch.zeros(1, 4), torch.log(torch.tensor([0.15, 0.1, 0.2, 0.1]))).descriptor()
        ip = distribution.Uniform(4, torch.zeros(4), torch.full((4,), 10.0)).descriptor()
        return model, lp, ip
    if workload == "gamma":         # Gamma (distribution.py:90-137) as importance proposal AND prior of the fused kernel
        model = Mixture_set(EPS, prior=distribution.Gamma(torch.tensor([2.0, 2.0]), torch.tensor([1.0, 1.0]))).descriptor()
        lp = distribution.DiagGaussian(2, torch.zeros(1, 2), torch.log(torch.tensor([0.35, 0.35]))).descriptor()
        ip = distribution.Gamma(torch.tensor([4.0, 4.0]), torch.tensor([3.0, 3.0])).descriptor()
        return model, lp, ip
    model = Mixture_set(EPS).descriptor()
    lp = distribution.DiagGaussian(2, torch.zeros(1, 2), torch.log(torch.tensor([0.35, 0.35]))).descriptor()
    ip = distribution.DiagGaussian(2, torch.tensor([0.0, 0.0]), torch.tensor([0.0, 0.0])).descriptor()
    return model, lp, ip


def cpu_baseline(seconds_target=12.0):
    """The CPU checker (oracle/, a plain-C port of the reference loop, OpenMP over chains)
    timed on this host on a bounded sample of the same workload."""
    import oracle_lib
    model, lp, ip = descriptors()
    cores = min(len(os.sched_getaffinity(0)), 16)            # the GPU box's CPU share for one GPU
    os.environ["OMP_NUM_THREADS"] = str(cores)                # read by libgomp when the library is first loaded
    L = oracle_lib.load()

    def run(n, T):
        hc = oracle_lib.HostChains(np.zeros((n, 2), np.float32), np.zeros((n, 2), np.float32))
        hh = np.zeros((T, 2, n), np.float32)
        r, keep = oracle_lib.make_run(seed=1, step0=1, n_steps=T, gf=GF, batch=NBATCH, history=hh)
        cs = hc.struct()
        L.oracle_init_weights(C.byref(model), C.byref(ip), C.byref(cs))
        t = time.perf_counter()
        rc = L.oracle_glmcmc_steps(C.byref(model), C.byref(lp), C.byref(ip), C.byref(cs), C.byref(r))
        assert rc == 0
        return n * T / (time.perf_counter() - t)

    for _ in range(3):
        rate = run(8192, 100)                               # warm (threads, pages, clocks)
    n = 16384
    T = int(min(max(50, rate * seconds_target / n), 40000))
    rate = run(n, T)
    out = {"value": rate, "unit": "chain-steps/s", "cores": cores, "kind": "port",
           "sample": "oracle/glabc_oracle.c (C port of GLMCMC.py:58-104), OpenMP over %d threads, %d chains x %d "
                     "iterations of the bench workload" % (cores, n, T)}
    # the reference's own way of running -- one chain, one ATen operation at a time -- restated in oracle/aten_loop.py
    # (pinned to the reference's tapes bit for bit, tests/test_oracle_golden.py); ~3 s on one core of this host
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle"))
    import aten_loop
    out["reference_like"] = {"value": aten_loop.steps_per_second(3.0), "unit": "chain-steps/s", "cores": 1, "kind": "port",
                             "sample": "oracle/aten_loop.py: one chain, the reference's ATen operation sequence "
                                       "(GLMCMC.py:58-104, N = 5, gf 0.9), torch on one thread, ~3 s"}
    return out


def _omp_cores():
    cores = min(len(os.sched_getaffinity(0)), 16)            # the GPU box's CPU share for one GPU
    os.environ["OMP_NUM_THREADS"] = str(cores)                # read by libgomp when the library is first loaded
    return cores


def cpu_baseline_sampler(workload, batch, seconds_target=12.0):
    """The CPU checker's port of the workload's loop (oracle/glabc_oracle.c: GlobalMCMC.py:37-68, GLMALA.py:150-200, or
    GLMCMC.py:58-104 on the g-and-k / the Gamma configuration), OpenMP over chains, on a bounded sample of the bench workload."""
    import oracle_lib
    from glabcmcmc_amd import _capi
    model, lp, ip = descriptors(workload)
    cores = _omp_cores()
    L = oracle_lib.load()
    d = model.theta_dim
    gf = {"globalmcmc": 0.5, "glmala": 0.8, "gk": 0.9, "gamma": 0.9}[workload]
    mala = _capi.Mala(0.3, 0.3 ** 2, EPS ** 2, 100, 0)
    rng = np.random.default_rng(1)

    def run(n, T):
        if workload == "gk":
            theta0 = (rng.random((n, 4)) * 10).astype(np.float32)
            y0 = np.sort(rng.standard_normal((n, 8)) * 2 + 3, axis=1).astype(np.float32)
        else:
            theta0 = np.full((n, d), 1.4 if workload == "gamma" else 0.0, np.float32)
            y0 = (np.abs(theta0) + 0.2236068 * rng.standard_normal((n, d))).astype(np.float32)
        hc = oracle_lib.HostChains(theta0, y0)
        if workload == "glmala":
            hc.add_mala_state()
        hh = np.zeros((T, d, n), np.float32)
        r, keep = oracle_lib.make_run(seed=1, step0=1, n_steps=T, gf=gf, batch=batch, history=hh)
        cs = hc.struct()
        if workload == "glmala":
            assert L.oracle_glmala_init(C.byref(model), C.byref(cs)) == 0
        elif workload != "globalmcmc":
            assert L.oracle_init_weights(C.byref(model), C.byref(ip), C.byref(cs)) == 0
        t = time.perf_counter()
        if workload == "glmala":
            rc = L.oracle_glmala_steps(C.byref(model), C.byref(ip), C.byref(mala), C.byref(cs), C.byref(r))
        elif workload == "globalmcmc":
            rc = L.oracle_globalmcmc_steps(C.byref(model), C.byref(lp), C.byref(ip), C.byref(cs), C.byref(r))
        else:
            rc = L.oracle_glmcmc_steps(C.byref(model), C.byref(lp), C.byref(ip), C.byref(cs), C.byref(r))
        assert rc == 0
        return n * T / (time.perf_counter() - t)

    n = 4096 if workload == "glmala" else 16384
    for _ in range(2):
        rate = run(n, 4 if workload == "glmala" else 50)       # warm (threads, pages, clocks)
    T = int(min(max(4, rate * seconds_target / n), 40000))
    rate = run(n, T)
    what = {"globalmcmc": "GlobalMCMC.py:37-68", "glmala": "GLMALA.py:150-200", "gk": "GLMCMC.py:58-104 on the g-and-k Model",
            "gamma": "GLMCMC.py:58-104 with the Gamma prior / importance proposal"}[workload]
    return {"value": rate, "unit": "chain-steps/s", "cores": cores, "kind": "port",
            "sample": "oracle/glabc_oracle.c (C port of %s), OpenMP over %d threads, %d chains x %d iterations of the bench "
                      "workload" % (what, cores, n, T)}


def cpu_baseline_nf(flow_desc, rows_sample, rows_log_prob, couplings, seconds_target=12.0):
    """oracle_nf_sample + oracle_nf_log_prob (the checker's scalar restatement of the coupling stack) on a bounded number of
    rows, one thread (the functions are not threaded)."""
    import oracle_lib
    L = oracle_lib.load()

    def run(n_s, n_l):
        z, lq = np.empty((2, n_s), np.float32), np.empty(n_s, np.float32)
        x, lp = np.random.default_rng(0).standard_normal((2, n_l)).astype(np.float32), np.empty(n_l, np.float32)
        t = time.perf_counter()
        assert L.oracle_nf_sample(C.byref(flow_desc), None, 1234, 0, n_s, z.ctypes.data, lq.ctypes.data) == 0
        assert L.oracle_nf_log_prob(C.byref(flow_desc), x.ctypes.data, n_l, lp.ctypes.data) == 0
        return (n_s + n_l) / (time.perf_counter() - t)

    rate = run(500, 100)
    n = int(max(600, min(rate * seconds_target, rows_sample + rows_log_prob)))
    n_l = max(1, int(n * rows_log_prob / float(rows_sample + rows_log_prob)))
    rate = run(n - n_l, n_l)
    return {"value": rate, "unit": "rows/s", "cores": 1, "kind": "port",
            "sample": "oracle_nf_sample + oracle_nf_log_prob (oracle/glabc_oracle.c, scalar), %d + %d rows, %d couplings"
                      % (n - n_l, n_l, couplings)}


def cpu_baseline_nf_train(flow_desc, couplings, seconds_target=12.0):
    """oracle_nf_grad (forward_kld and its gradient, float32 states + double chain rule, scalar) on a bounded number of rows"""
    import oracle_lib
    from glabcmcmc_amd import _capi
    L = oracle_lib.load()
    gp = np.zeros(couplings * _capi.NF_COUPLING_FLOATS, np.float32)
    gb, loss = np.zeros(4, np.float32), np.zeros(1, np.float32)

    def run(n):
        x = (np.random.default_rng(0).standard_normal((2, n)) * 0.8 + 0.3).astype(np.float32)
        t = time.perf_counter()
        assert L.oracle_nf_grad(C.byref(flow_desc), x.ctypes.data, n, gp.ctypes.data, gb.ctypes.data, loss.ctypes.data) == 0
        return n / (time.perf_counter() - t)

    rate = run(300)
    n = int(max(300, rate * seconds_target))
    return {"value": run(n), "unit": "rows/s", "cores": 1, "kind": "port",
            "sample": "oracle_nf_grad (oracle/glabc_oracle.c: loss + gradient of forward_kld, scalar), %d rows, %d couplings; no "
                      "Adam step" % (n, couplings)}


def cpu_baseline_kde(kde_desc, n_centres, seconds_target=12.0):
    """oracle_kde_log_prob (the checker's scalar logsumexp over the centres) on a bounded number of points"""
    import oracle_lib
    L = oracle_lib.load()

    def run(n):
        pts = np.random.default_rng(0).standard_normal((2, n)).astype(np.float32)
        out = np.empty(n, np.float32)
        t = time.perf_counter()
        assert L.oracle_kde_log_prob(C.byref(kde_desc), pts.ctypes.data, n, out.ctypes.data) == 0
        return float(n) * n_centres / (time.perf_counter() - t)

    rate = run(200)
    n = int(max(200, rate * seconds_target / n_centres))
    return {"value": run(n), "unit": "pair-evaluations/s", "cores": 1, "kind": "port",
            "sample": "oracle_kde_log_prob (oracle/glabc_oracle.c, scalar), %d points x %d centres" % (n, n_centres)}


def lib_sha16():
    import hashlib
    from glabcmcmc_amd import _capi
    h = hashlib.sha256()
    with open(_capi.LIB_PATH, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 22), b""):
            h.update(chunk)
    return h.hexdigest()[:16]


def src_sha16():
    """SHA-256 over the sources libglabc_hip.so is built from (csrc/*.hip, *.h, Makefile, include/*.h), in sorted order: the
    second key of a counter file -- hipcc's objects are not byte-reproducible, so a clean rebuild of the SAME sources changes
    lib_sha16 but not this"""
    import glob
    import hashlib
    csrc = os.path.join(ROOT, "gl-abc-mcmc_amd", "csrc")
    files = sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "glabc_*.h")) + [os.path.join(csrc, "Makefile")] +
                   glob.glob(os.path.join(ROOT, "include", "*.h")))
    h = hashlib.sha256()
    for p in files:
        h.update(os.path.basename(p).encode() + b"\0")
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def counted(workload, n, K, batch=None):
    """profiles/r03_pmc_<workload>.json (tools/profile_workload.sh + summarise_workload.py): the rocprofv3 counters of exactly
    this configuration -- used ONLY while the library that runs is the build they were taken with (SHA-256 of
    libglabc_hip.so; or, after a clean rebuild, the same kernel sources: src_sha16) and the launch shape is the same; anything
    else returns None instead of a stale number."""
    try:
        with open(os.path.join(ROOT, "profiles", "r03_pmc_%s.json" % workload)) as f:
            p = json.load(f)
        cfg = p["config"]
        if cfg.get("chains") != n or cfg.get("iters_per_launch") != K:
            return None
        if p.get("lib_sha16") == lib_sha16():
            p["keyed_by"] = "libglabc_hip.so %s" % p["lib_sha16"]
        elif p.get("src_sha16") and p["src_sha16"] == src_sha16():
            p["keyed_by"] = "kernel sources %s (library rebuilt from the same sources)" % p["src_sha16"]
        else:
            return None
        if batch is not None and cfg.get("batch_size") not in (None, batch):
            return None
        return p
    except (OSError, KeyError, ValueError):
        return None


def valu_block(p, workload, kernel_ms):
    """roofline.valu: counted vector instructions x this run's kernel time, against the guide's vector issue peak"""
    d = p["derived"]
    insts = p["raw_avg_per_launch"]["SQ_INSTS_VALU"]
    wips = insts / (kernel_ms * 1e-3)
    peak = 1024 * 2.4e9 / 2              # MI355X_MICROARCH.md: wave64 v_fma_f32 = 2 cycles on a SIMD-32; 1024 SIMDs at 2.4 GHz
    return {"source": "profiles/r03_pmc_%s.json (rocprofv3 --pmc, %s) + this run's kernel time" % (workload, p["keyed_by"]),
            "valu_insts_per_launch": insts, "valu_insts_per_group_step": d["valu_insts_per_group_step"],
            "waves_per_launch": d.get("waves_per_launch"), "wave_insts_per_s": wips,
            "simd_cycles_per_valu_inst": 1024 * 2.4e9 * kernel_ms * 1e-3 / insts,
            "vector_peak_wave_insts_per_s": peak, "frac_of_vector_peak": wips / peak,
            "note": "one group = 64 chains; simd_cycles_per_valu_inst = SIMD cycles (2.4 GHz nominal) per vector instruction issued "
                    "on it, 2.0 at the guide's peak; the instruction mix of this path (v_mad_u64_u32, transcendental seeds, f64 in "
                    "GLMALA) saturates near 3.0 - 3.3 (DESIGN.md 4.1)"}


def bench_nf(args):
    """BASELINE configs[4]: RealNVP global proposal, 8 couplings x MLP[1,128,128,2], 65 536 chains x N=5 =
    327 680 rows per global step, on the f32 matrix cores.  One step = NF_model.sample(rows) + NF_model.log_prob
    of 65 536 current states (what one global step of GLMCMC_NF evaluates, GLMCMC_NFs.py:70-72,96-98)."""
    from glabcmcmc_amd.flows import RealNVP
    torch.cuda.set_device(0)
    torch.manual_seed(0)
    flow = RealNVP(args.couplings)
    with torch.no_grad():
        for c in flow.couplings:
            c.l3.weight.normal_(0, 0.3 / 128 ** 0.5)
            c.l3.bias.normal_(0, 0.1)
    flow = flow.cuda()
    rows, cur = args.chains * NBATCH, args.chains
    blob = flow.packed_params()
    f = flow.descriptor(blob)
    from glabcmcmc_amd import _capi
    lib = _capi.lib()
    z = torch.empty(2, rows, dtype=torch.float32, device="cuda")
    lq = torch.empty(rows, dtype=torch.float32, device="cuda")
    x = torch.randn(2, cur, device="cuda")
    lp = torch.empty(cur, dtype=torch.float32, device="cuda")

    def one_step(i):
        _capi.check(lib.glabc_nf_sample(C.byref(f), None, 1234, i * rows, rows, z.data_ptr(), lq.data_ptr(), None), "nf_sample")
        _capi.check(lib.glabc_nf_log_prob(C.byref(f), x.data_ptr(), cur, lp.data_ptr(), None), "nf_log_prob")

    for i in range(args.warmup):
        one_step(i)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i, (a, b) in enumerate(ev):
        a.record()
        one_step(i)
        b.record()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    flop = 2.0 * (128 + 128 * 128 + 2 * 128) * args.couplings * (rows + cur)        # 2*(1x128 + 128x128 + 128x2) per row-coupling
    mfma_flop = 2.0 * 128 * 128 * args.couplings * (rows + cur)
    out = {"metric": "NF proposal rows/sec (sample + log_prob), RealNVP %d couplings" % args.couplings,
           "value": (rows + cur) * args.steps / elapsed, "unit": "rows/s", "n_gpus": 1, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "GLMCMC_NF coupling stack: %d couplings x MLP[1,128,128,2], %d sample rows + %d log_prob rows"
                                  " per step (BASELINE configs[4])" % (args.couplings, rows, cur)},
           "roofline": {"bound": "mfma", "achieved": mfma_flop / (kernel_ms * 1e-3) / 1e12, "peak": 157.3, "unit": "TFLOP/s",
                        "frac": mfma_flop / (kernel_ms * 1e-3) / 1e12 / 157.3, "traffic": None,
                        "kernel": "glabc::nf_kernel<forward> + <inverse> (v_mfma_f32_32x32x2_f32)", "kernel_ms": kernel_ms,
                        "all_flop_TFLOPs": flop / (kernel_ms * 1e-3) / 1e12}}
    pm = counted("nf", None, None)
    if pm is not None and "mfma" in pm:
        out["roofline"]["mfma_counters"] = {"source": "profiles/r03_pmc_nf.json (%s)" % pm["keyed_by"],
                                            "per_kernel": pm["mfma"]["per_kernel"]}
    if not args.no_cpu_baseline:
        host_blob = blob.cpu().contiguous()                   # the checker reads host memory
        out["cpu_baseline"] = cpu_baseline_nf(flow.descriptor(host_blob), rows, cur, args.couplings)
    print(json.dumps(out), flush=True)


def bench_nf_train(args):
    """The training step of GLMCMC_NFs.py:112-124 at the pool size of BASELINE configs[4] in batched use: forward_kld over
    65 536 chains x (N=5 x step_size=20) = 6 553 600 resampled pool rows, 8 couplings; one step = loss + gradient
    (glabc_nf_grad: inverse pass + one backward launch per coupling + reduction) + Adam (glabc_adam_step).  The same step by
    torch autograd + torch.optim.Adam on the same device is timed beside it (on a quarter of the rows: its activations
    need ~1 KiB per row and coupling)."""
    from glabcmcmc_amd.flows import HipAdam, RealNVP
    import copy
    torch.cuda.set_device(0)
    torch.manual_seed(0)
    flow = RealNVP(args.couplings)
    with torch.no_grad():
        for c in flow.couplings:
            c.l3.weight.normal_(0, 0.3 / 128 ** 0.5)
            c.l3.bias.normal_(0, 0.1)
    flow = flow.cuda()
    rows = args.chains * NBATCH * 20
    x = torch.randn(2, rows, device="cuda") * 0.8 + 0.3
    opt = HipAdam(flow)
    for _ in range(args.warmup):
        opt.step(x, chain_major=True)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for a, b in ev:
        a.record()
        opt.gradient(x, chain_major=True)
        b.record()
    torch.cuda.synchronize()
    grad_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    t0 = time.perf_counter()
    for _ in range(args.steps):
        opt.step(x, chain_major=True)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    # torch autograd + Adam, same model, a quarter of the rows
    ref = copy.deepcopy(flow)
    topt = torch.optim.Adam(ref.parameters(), lr=5e-4, weight_decay=1e-5)
    xr = x[:, :rows // 4].t().contiguous()
    def torch_step():
        topt.zero_grad()
        loss = ref.forward_kld(xr)
        loss.backward()
        topt.step()
        return loss
    torch_step()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(3):
        torch_step()
    torch.cuda.synchronize()
    torch_rows_per_s = 3 * (rows // 4) / (time.perf_counter() - t1)
    mfma_flop = 4 * 2.0 * 128 * 128 * args.couplings * rows              # inverse pass + (1), (2), (3) of the backward sweep
    out = {"metric": "NF training rows/sec (forward_kld + gradient + Adam), RealNVP %d couplings" % args.couplings,
           "value": rows * args.steps / elapsed, "unit": "rows/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f32", "data": "synthetic",
           "config": {"workload": "GLMCMC_NF training step: %d couplings x MLP[1,128,128,2], %d pool rows (GLMCMC_NFs.py:112-124)"
                                  % (args.couplings, rows)},
           "torch_autograd_rows_per_s": torch_rows_per_s, "speedup_vs_torch_autograd": rows * args.steps / elapsed / torch_rows_per_s,
           "roofline": {"bound": "mfma", "achieved": mfma_flop / (grad_ms * 1e-3) / 1e12, "peak": 157.3, "unit": "TFLOP/s",
                        "frac": mfma_flop / (grad_ms * 1e-3) / 1e12 / 157.3, "traffic": None,
                        "kernel": "glabc::nf_kernel<inverse> + glabc::nf_backward_kernel x couplings (v_mfma_f32_32x32x2_f32)",
                        "kernel_ms": grad_ms}}
    pm = counted("nf_train", None, None)
    if pm is not None and "mfma" in pm:
        out["roofline"]["mfma_counters"] = {"source": "profiles/r03_pmc_nf_train.json (%s)" % pm["keyed_by"],
                                            "per_kernel": pm["mfma"]["per_kernel"]}
    if not args.no_cpu_baseline:
        host_blob = flow.packed_params().cpu().contiguous()   # the checker reads host memory
        out["cpu_baseline"] = cpu_baseline_nf_train(flow.descriptor(host_blob), args.couplings)
    print(json.dumps(out), flush=True)


def bench_kde(args):
    """KernelDensity.log_prob (kernel_density.py:96-128), the dense (points x centres) logsumexp of
    AGLMCMC's adaptive proposal: 524 288 evaluation points against 8192 weighted centres in d = 2 per step."""
    from glabcmcmc_amd import KernelDensity
    torch.cuda.set_device(0)
    torch.manual_seed(0)
    S, P = 8192, 8 * args.chains
    kde = KernelDensity(device="cuda", seed=1).fit(torch.randn(S, 2), torch.rand(S))
    pts = torch.randn(2, P, device="cuda")
    for _ in range(args.warmup):
        kde.log_prob_soa(pts)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for a, b in ev:
        a.record()
        out = kde.log_prob_soa(pts)
        b.record()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    algo_bytes = 4.0 * (3 * P + 3 * S + P)                      # points in, centres + log-weights in, densities out
    achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9
    cpu = None
    if not args.no_cpu_baseline:
        from glabcmcmc_amd import _capi
        kd = kde.descriptor()
        hx, hlw = kde._x.cpu().numpy().copy(), kde._log_w.cpu().numpy().copy()
        hk = _capi.Kde()
        hk.dim, hk.n_samples, hk.x, hk.log_w, hk.cum_q = kd.dim, kd.n_samples, hx.ctypes.data, hlw.ctypes.data, None
        for j in range(kd.dim):
            hk.bandwidth[j] = kd.bandwidth[j]
        hk.sum_log_bw, hk.c_2pi = kd.sum_log_bw, kd.c_2pi
        cpu = cpu_baseline_kde(hk, S)
    print(json.dumps({"metric": "KDE log-density pair evaluations/sec (points x centres), d = 2", "value": float(S) * P * args.steps / elapsed,
                      "cpu_baseline": cpu,
                      "unit": "pair-evaluations/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
                      "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                      "dtype": "f32", "data": "synthetic",
                      "config": {"workload": "KernelDensity.log_prob, %d points x %d centres, d=2 (AGLMCMC's adaptive proposal)" % (P, S)},
                      "finite": bool(torch.isfinite(out).all()),
                      "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                                   "traffic": None, "kernel": "kde_log_prob_kernel<2>", "kernel_ms": kernel_ms,
                                   "algorithmic_bytes_per_launch": algo_bytes,
                                   "note": "dense O(points x centres) arithmetic on O(points + centres) bytes: VALU-bound by "
                                           "construction (~60 vector instructions per pair over two passes); see DESIGN.md 4.3"}}), flush=True)


def bench_glmcmc_nf(args):
    """BASELINE configs[4] end to end: GLMCMC_NF (GLMCMC_NFs.py:43-186) at 65 536 chains, RealNVP with 8 couplings, N = 5,
    pools of step_size = 20 slices per chain, gf 0.9, one Adam step at the first pool refresh (Train_step = 1).  One bench
    step = --iters iterations of the whole loop: pool draws through the MFMA forward kernel (amortised 5 rows per chain and
    iteration), pool weights, the per-iteration NF-iSIR / RW-MH kernel and the log_prob refresh of the chains that moved."""
    import glabcmcmc_amd as g
    from glabcmcmc_amd.examples.Mixture import Mixture_set
    torch.cuda.set_device(0)
    n, K, N, step_size = args.chains, min(args.iters, 100), NBATCH, 20
    m = Mixture_set(EPS)
    lp = g.DiagGaussian(2, torch.zeros(1, 2), torch.log(torch.tensor([0.35, 0.35])))
    torch.manual_seed(0)
    from glabcmcmc_amd.flows import RealNVP
    flow = RealNVP(args.couplings)
    with torch.no_grad():
        for c in flow.couplings:
            c.l3.weight.normal_(0, 0.3 / 128 ** 0.5)
            c.l3.bias.normal_(0, 0.1)
    theta0 = torch.zeros(n, 2)
    y0 = (0.05 ** 0.5) * torch.randn(n, 2)
    st = {}

    def one_step(i):
        g.GLMCMC_NF(m, K + 1, theta0, y0, lp, None, GF, step_size, N, None, 1 if i == 0 else 0, num_layers=args.couplings,
                    seed=100 + i, flow=flow, return_device=True, verbose=False, state_out=st)

    for i in range(args.warmup):
        one_step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        one_step(args.warmup + i)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    per_iter = elapsed / (args.steps * K)
    rows_per_iter = float(N) * step_size * n * st["pools_drawn"] / K   # pool rows drawn per iteration (last run)
    moved = float(st["chains"].n_moves.double().sum()) / K            # log_prob refreshes per iteration (last run)
    mfma_flop = 2.0 * 128 * 128 * args.couplings * (rows_per_iter + moved + n / K)
    achieved = mfma_flop / per_iter / 1e12
    out = {"metric": "MH accept-steps/sec, GLMCMC_NF end to end (RealNVP %d couplings), 65 536 chains" % args.couplings,
           "value": float(n) / per_iter, "unit": "chain-steps/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f32", "data": "synthetic",
           "config": {"workload": "GLMCMC_NF, %d couplings x MLP[1,128,128,2], N=5, step_size=%d, gf=0.9, Mixture_set eps=0.05 "
                                  "(BASELINE configs[4]); host loop: 2 kernel launches per iteration + pool refreshes"
                                  % (args.couplings, step_size), "chains_per_gpu": n, "iters_per_step": K},
           "iterations_per_s": 1.0 / per_iter, "us_per_iteration": per_iter * 1e6,
           "moved_chains_per_iteration": moved, "pool_rows_per_iteration": rows_per_iter, "pools_per_step": st["pools_drawn"],
           "roofline": {"bound": "mfma", "achieved": achieved, "peak": 157.3, "unit": "TFLOP/s", "frac": achieved / 157.3,
                        "traffic": None, "kernel": "glabc::nf_kernel<forward, pairs> (pool draws)", "kernel_ms": None,
                        "note": "whole-loop rate: MFMA flops of the pool draws + log_prob refreshes over the wall time of the "
                                "loop (pool weights, step kernels, host launches included); the kernel alone: --workload nf"}}
    if not args.no_cpu_baseline:
        host_blob = flow.cpu().packed_params().cpu().contiguous()
        b = cpu_baseline_nf(flow.descriptor(host_blob), int(rows_per_iter), int(moved) + 1, args.couplings, seconds_target=10.0)
        rows_per_chain_step = (rows_per_iter + moved) / float(n)
        out["cpu_baseline"] = {"value": b["value"] / rows_per_chain_step, "unit": "chain-steps/s", "cores": 1, "kind": "port",
                               "sample": "the loop's dominant part only -- the flow's pool draws and log_prob refreshes, %.2f rows per "
                                         "chain-step -- through %s" % (rows_per_chain_step, b["sample"])}
    print(json.dumps(out), flush=True)


RTC_MIXTURE = """
GLABC_SIMULATOR void glabc_user_simulate(const float* theta, const float* eps, float* y)
{   /* examples/Mixture.py:19-23 written by a user as C: y = |theta| + sqrt(0.05) eps */
    for (int j = 0; j < GLABC_Y_DIM; ++j) y[j] = fabsf(theta[j]) + 0.2236068f * eps[j];
}
"""


def bench_rtc(args):
    """The headline workload (GLMCMC iSIR N=5, gf 0.9, 65 536 chains x --iters iterations per launch, history + sums) with the
    Model's simulator given as C SOURCE and compiled into the fused kernel at run time (compiled.CompiledModel, glabc_rtc_*):
    what a user's own simulator costs when it can be stated in C -- against `--workload callback` (the same Model as Python
    callbacks) and the built-in Mixture_set."""
    import glabcmcmc_amd as g
    from glabcmcmc_amd import _capi, engine
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    n, K, N = args.chains, args.iters, args.batch
    prior = g.DiagGaussian(2, torch.tensor([0.0, 0.0]), torch.tensor([0.0, 0.0]))
    t0 = time.perf_counter()
    cm = g.CompiledModel(2, 2, RTC_MIXTURE, prior, [1.5, 1.5], EPS)
    prog = cm.program(_capi.ALGO_GLMCMC, N)
    compile_s = time.perf_counter() - t0
    model = cm.descriptor()
    lp = g.DiagGaussian(2, torch.zeros(1, 2), torch.log(torch.tensor([0.35, 0.35]))).descriptor()
    ip = g.DiagGaussian(2, torch.tensor([0.0, 0.0]), torch.tensor([0.0, 0.0])).descriptor()
    gen = torch.Generator().manual_seed(1234)
    chains = engine.ChainBatch(torch.zeros(n, 2), (0.05 ** 0.5) * torch.randn(n, 2, generator=gen), dev)
    hist = torch.empty(K, 2, n, dtype=torch.float32, device=dev)
    mom = engine.Moments(n, 2, dev)
    idx = [0]

    def one_step():
        engine.run_steps("glabc_glmcmc_steps", model, lp, ip, chains, K, 1 + idx[0] * K, 20261003, GF, N, history=hist, moments=mom,
                         steps_per_launch=K, rtc_program=prog)
        idx[0] += 1

    for _ in range(args.warmup):
        one_step()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for a, b in ev:
        a.record()
        one_step()
        b.record()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    algo_bytes = 2 * 4 * 7 * n + 2 * 8 * 8 * n + 4 * 2 * n * K
    achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9
    esjd = mom.esjd()
    ok = torch.isfinite(esjd)
    out = {"metric": "MH accept-steps/sec, user simulator compiled into the fused kernel at run time, 65 536 chains, dim=2",
           "value": float(n) * K * args.steps / elapsed, "unit": "chain-steps/s", "n_gpus": 1, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "GLMCMC iSIR N=%d gf=0.9, CompiledModel (the Mixture simulator as user C source, hiprtc) eps=0.05 d=2"
                                  % N, "chains_per_gpu": n, "iters_per_step": K, "batch_size": N},
           "compile_seconds": compile_s, "esjd_mean": float(esjd[ok].double().mean()),
           "mean_theta_sq": float(mom.second_moment().diagonal(dim1=1, dim2=2).mean()), "analytic_mean_theta_sq": 2.081014,
           "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                        "traffic": None, "kernel": ("glabc::team_sampler_kernel<D=2, N=%d, VAR_GAUSS_UNIT, %d wavefronts per 64 chains> (hiprtc)" % (N, 3 if n <= 65536 else 2)
                                   if 16384 <= n <= 131072 and N >= 2 else "glabc::sampler_kernel<GLMCMC, D=2, N=%d, VAR_GAUSS_UNIT> (hiprtc)" % N),
                        "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": algo_bytes,
                        "note": "VALU-bound like the built-in kernel; the host picks the unit-Gaussian or the generic instantiation "
                                "per launch as for the built-in Models (default schedule, the user's simulator inlined)"}}
    if not args.no_cpu_baseline and N == NBATCH:
        out["cpu_baseline"] = cpu_baseline()                  # the same workload: the simulator is the built-in one written as C
        out["cpu_baseline"].pop("reference_like", None)
    print(json.dumps(out), flush=True)


def bench_aglmcmc(args):
    """SURVEY.md 8(f) f-4 end to end: AGLMCMC (AGLMCMC.py:44-289; adaptive KDE proposal, annealed threshold) at 65 536 chains
    sharing one density: per iteration KernelDensity.log_prob of the current states + the pool-iSIR / RW-MH kernel, per pool
    refresh the quantile update, KDE refit, 4x oversampled redraw, pool weights."""
    import glabcmcmc_amd as g
    from glabcmcmc_amd.examples.Mixture import Mixture_set
    import warnings
    torch.cuda.set_device(0)
    n, K, N, step_size = args.chains, min(args.iters, 60), NBATCH, 20
    m = Mixture_set(EPS)
    lp = g.DiagGaussian(2, torch.zeros(1, 2), torch.log(torch.tensor([0.35, 0.35])))
    isir = g.DiagGaussian(2, torch.tensor([0.0, 0.0]), torch.tensor([0.5, 0.5]))
    torch.manual_seed(0)
    theta0 = torch.zeros(n, 2)
    y0 = (0.05 ** 0.5) * torch.randn(n, 2)
    st = {}

    def one_step(i):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            g.AGLMCMC(m, K + 1, theta0, y0, lp, isir, None, 0.9, step_size, N, 0.8, 0.2, seed=50 + i, return_device=True,
                      verbose=False, state_out=st)

    for i in range(args.warmup):
        one_step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        one_step(args.warmup + i)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    per_iter = elapsed / (args.steps * K)
    out = {"metric": "MH accept-steps/sec, AGLMCMC end to end (adaptive KDE proposal), 65 536 chains",
           "value": float(n) / per_iter, "unit": "chain-steps/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f32", "data": "synthetic",
           "config": {"workload": "AGLMCMC N=5 step_size=%d gf=0.9 alpha=0.8 hat_eps_T=0.2, Mixture_set eps=0.05, one KDE shared by "
                                  "the chains (8192 training rows)" % step_size, "chains_per_gpu": n, "iters_per_step": K},
           "iterations_per_s": 1.0 / per_iter, "us_per_iteration": per_iter * 1e6,
           "kde_refits_last_step": st.get("num_train"),
           "roofline": {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                        "kernel": "kde_log_prob_kernel<2> (dense points x centres logsumexp)", "kernel_ms": None,
                        "note": "whole-loop rate; the dominant kernel is VALU-bound O(points x centres) work, rated by --workload kde"}}
    if not args.no_cpu_baseline:
        from glabcmcmc_amd import KernelDensity, _capi
        kde = KernelDensity(device="cuda", seed=1).fit(torch.randn(8192, 2), torch.rand(8192))
        kd = kde.descriptor()
        hx, hlw = kde._x.cpu().numpy().copy(), kde._log_w.cpu().numpy().copy()
        hk = _capi.Kde()
        hk.dim, hk.n_samples, hk.x, hk.log_w, hk.cum_q = kd.dim, kd.n_samples, hx.ctypes.data, hlw.ctypes.data, None
        for j in range(kd.dim):
            hk.bandwidth[j] = kd.bandwidth[j]
        hk.sum_log_bw, hk.c_2pi = kd.sum_log_bw, kd.c_2pi
        b = cpu_baseline_kde(hk, 8192, seconds_target=10.0)
        # per chain-step the loop evaluates the density of ~N pool rows (every refresh: N step_size rows per chain per step_size
        # global moves) against the 8192 centres
        pairs = float(N) * 8192
        out["cpu_baseline"] = {"value": b["value"] / pairs, "unit": "chain-steps/s", "cores": 1, "kind": "port",
                               "sample": "the loop's dominant part only -- KernelDensity.log_prob of the pool rows, ~%d pair "
                                         "evaluations per chain-step -- through %s" % (pairs, b["sample"])}
    print(json.dumps(out), flush=True)


def cpu_baseline_callback(n, iters, seconds_target=12.0):
    """The callback workload on the host: the CPU checker's split-phase twins (oracle_propose / oracle_select) around the SAME
    plain-torch Model evaluated on CPU tensors -- a batched-ATen CPU path of the loop bench_callback times on the GPU."""
    import oracle_lib
    from glabcmcmc_amd import _capi as A, distribution
    from glabcmcmc_amd.examples.UserModel import TorchMixture
    cores = min(len(os.sched_getaffinity(0)), 16)
    torch.set_num_threads(cores)
    L = oracle_lib.load()
    N = NBATCH
    model = TorchMixture(2, EPS)
    lp = distribution.DiagGaussian(2, torch.zeros(2), torch.log(torch.tensor([0.35, 0.35]))).descriptor()
    ip = distribution.DiagGaussian(2, torch.zeros(2), torch.zeros(2)).descriptor()
    R = N * n
    hc = oracle_lib.HostChains(np.zeros((n, 2), np.float32), (0.05 ** 0.5) * np.random.default_rng(0).standard_normal((n, 2)).astype(np.float32))
    cs = hc.struct()
    f32 = np.float32
    buf = dict(theta_prop=np.zeros((R, 2), f32), log_q=np.zeros(R, f32), log_u=np.zeros(n, f32), u_res=np.zeros(n, np.float64),
               is_global=np.zeros(n, np.int32), prior_cur=np.zeros(n, f32), kern_cur=np.zeros(n, f32))
    buf["prior_cur"][:] = model.prior_log_prob(torch.from_numpy(hc.theta.T.copy())).numpy()
    buf["kern_cur"][:] = model.calculate_log_kernel(torch.from_numpy(hc.y.T.copy())).numpy()
    io = A.StepIO(N, 2, 2, 0, buf["theta_prop"].ctypes.data, buf["log_q"].ctypes.data, None, buf["log_u"].ctypes.data,
                  buf["u_res"].ctypes.data, buf["is_global"].ctypes.data, None, None, None, buf["prior_cur"].ctypes.data,
                  buf["kern_cur"].ctypes.data, None)
    done, t0 = 0, time.perf_counter()
    while done < iters and (done < 5 or time.perf_counter() - t0 < seconds_target):
        run, keep = oracle_lib.make_run(seed=1, step0=1 + done, n_steps=1, gf=GF, batch=N)
        assert L.oracle_propose(0, C.byref(lp), C.byref(ip), C.byref(cs), C.byref(run), C.byref(io)) == 0
        th = torch.from_numpy(buf["theta_prop"])
        prior = model.prior_log_prob(th).numpy().astype(f32)
        y = model.generate_samples(th).contiguous()
        kern = model.calculate_log_kernel(y).numpy().astype(f32)
        yn = y.numpy()
        io.prior_prop, io.y_prop, io.kern_prop = prior.ctypes.data, yn.ctypes.data, kern.ctypes.data
        assert L.oracle_select(0, C.byref(ip), C.byref(cs), C.byref(run), C.byref(io)) == 0
        done += 1
    dt = time.perf_counter() - t0
    return {"value": n * done / dt, "unit": "chain-steps/s", "cores": cores, "kind": "port",
            "sample": "oracle_propose -> examples/UserModel.TorchMixture on CPU tensors (torch, %d threads) -> oracle_select, %d chains x "
                      "%d iterations of the callback workload" % (cores, n, done)}


def bench_callback(args):
    """SURVEY.md 8b-ii, the reference's plug-in API: GLMCMC (iSIR N=5, gf 0.9) with a user Model that is a plain-torch
    object -- no descriptor, so every iteration is glabc_propose -> the Model's callbacks on a (5 * chains, 2) batch ->
    glabc_select (generic.py).  One step = --iters iterations of all chains."""
    from glabcmcmc_amd import GLMCMC, _capi, distribution, engine
    from glabcmcmc_amd.examples.UserModel import TorchMixture
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    n, K, N = args.chains, min(args.iters, 200), NBATCH
    model = TorchMixture(2, EPS)
    lp = distribution.DiagGaussian(2, torch.zeros(2), torch.log(torch.tensor([0.35, 0.35])))
    ip = distribution.DiagGaussian(2, torch.zeros(2), torch.zeros(2))
    torch.manual_seed(0)
    state = {"theta": torch.zeros(n, 2), "y": (0.05 ** 0.5) * torch.randn(n, 2)}
    mom = engine.Moments(n, 2, dev)
    step_idx = [0]

    st_last = {}

    def one_step():
        st = st_last
        st.clear()
        GLMCMC(model, K + 1, state["theta"], state["y"], lp, None, GF, ip, N, seed=20261003 + step_idx[0], record_history=False,
               stats=mom, verbose=False, state_out=st, sentinel_redraw=not args.no_sentinel,
               graph=False if args.no_graph else "auto")
        ch = st["chains"]
        state["theta"], state["y"] = ch.theta.t(), ch.y.t()            # stay on the device (prepare() copies through the host)
        step_idx[0] += 1

    for _ in range(args.warmup):
        one_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    # the two HIP kernels of an iteration, timed on their own (HIP events on the launch stream, 100 launches each)
    lib = _capi.lib()
    R = N * n
    f32 = dict(dtype=torch.float32, device=dev)
    buf = dict(theta_prop=torch.zeros(R, 2, **f32), log_q=torch.zeros(R, **f32), log_u=torch.zeros(n, **f32),
               u_res=torch.zeros(n, dtype=torch.float64, device=dev), is_global=torch.zeros(n, dtype=torch.int32, device=dev),
               y_prop=torch.randn(R, 2, **f32), prior_prop=-torch.rand(R, **f32), kern_prop=-torch.rand(R, **f32),
               prior_cur=-torch.rand(n, **f32), kern_cur=-torch.rand(n, **f32))
    io = _capi.StepIO(N, 2, 2, 0, buf["theta_prop"].data_ptr(), buf["log_q"].data_ptr(), None, buf["log_u"].data_ptr(),
                      buf["u_res"].data_ptr(), buf["is_global"].data_ptr(), buf["y_prop"].data_ptr(), buf["prior_prop"].data_ptr(),
                      buf["kern_prop"].data_ptr(), buf["prior_cur"].data_ptr(), buf["kern_cur"].data_ptr(), None)
    chains = engine.ChainBatch(torch.zeros(n, 2), torch.zeros(n, 2), dev)
    cs, ms = chains.struct(), mom.struct()
    hist = torch.zeros(2, n, **f32)
    run = _capi.Run()
    run.seed, run.step0, run.n_steps, run.global_frequency, run.batch_size = 1, 1, 1, GF, N
    run.history, run.hist_stride, run.moments = hist.data_ptr(), n, C.pointer(ms)
    lpd, ipd = lp.descriptor(), ip.descriptor()

    def timed(fn, reps=100):
        for _ in range(5):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps

    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    ms_prop = timed(lambda: _capi.check(lib.glabc_propose(0, C.byref(lpd), C.byref(ipd), C.byref(cs), C.byref(run), C.byref(io),
                                                          stream), "glabc_propose"))
    ms_sel = timed(lambda: _capi.check(lib.glabc_select(0, C.byref(ipd), C.byref(cs), C.byref(run), C.byref(io), stream),
                                       "glabc_select"))
    bytes_prop = 4.0 * R * (2 + 1) + n * (4 + 8 + 4) + 4.0 * n * 2
    bytes_sel = 4.0 * R * 3 + n * (4 + 8 + 4) + n * (8 + 16) + 4.0 * n * 2 + 2 * 8.0 * 8 * n
    achieved = (bytes_prop + bytes_sel) / ((ms_prop + ms_sel) * 1e-3) / 1e9
    esjd = mom.esjd()
    ok = torch.isfinite(esjd)
    out = {"metric": "MH accept-steps/sec, user Model as torch callbacks (split-phase path), 65 536 chains, dim=2",
           "value": float(n) * K * args.steps / elapsed, "unit": "chain-steps/s", "n_gpus": 1, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "GLMCMC iSIR N=5 gf=0.9, examples/UserModel.TorchMixture (plain torch, no descriptor) eps=0.05 "
                                  "d=2: glabc_propose -> callbacks -> glabc_select per iteration", "chains_per_gpu": n,
                      "iters_per_step": K, "batch_size": N, "sentinel_redraw": not args.no_sentinel,
                      "hip_graph": bool(st_last.get("graph"))},
           "us_per_iteration": elapsed / (args.steps * K) * 1e6,
           "esjd_mean": float(esjd[ok].double().mean()),
           "mean_theta_sq": float(mom.second_moment().diagonal(dim1=1, dim2=2).mean()), "analytic_mean_theta_sq": 2.081014,
           "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                        "traffic": None, "kernel": "glabc::propose_kernel + glabc::select_kernel",
                        "kernel_ms": ms_prop + ms_sel, "propose_ms": ms_prop, "select_ms": ms_sel,
                        "algorithmic_bytes_per_launch": bytes_prop + bytes_sel,
                        "note": "the two HIP kernels take %.0f us of the %.0f us of an iteration; the rest is the Model's "
                                "~10 torch kernels and launch latency" % ((ms_prop + ms_sel) * 1e3, elapsed / (args.steps * K) * 1e6)}}
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_callback(n, 400)
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--chains", type=int, default=65536, help="chains per GPU")
    ap.add_argument("--iters", type=int, default=2000, help="MH iterations fused into one launch (= one bench step)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-history", action="store_true", help="diagnostic: do not write Theta_Re rows")
    ap.add_argument("--no-moments", action="store_true", help="diagnostic: do not stream the ESJD / moment sums (the JSON line "
                    "then carries no ESJD / moments)")
    ap.add_argument("--lanes", type=int, default=0, help="lanes per chain (0 = library default; geometry only)")
    ap.add_argument("--debug-flags", type=int, default=0, help="glabc_run.debug_flags (execution strategy only; results unchanged)")
    ap.add_argument("--batch", type=int, default=NBATCH, help="iSIR batch size N (default 5 = BASELINE configs[1]); above 16 the "
                    "wide kernel of glabc_wide.hip runs")
    ap.add_argument("--couplings", type=int, default=8, help="nf workload: number of couplings")
    ap.add_argument("--fast-math", action="store_true", help="glmcmc workload: the OPT-IN fast-transcendental variant "
                    "(glabc_run.math_mode = GLABC_MATH_FAST, include/glabc.h) -- another stream of normals, the same law; its own "
                    "bench line, never the headline")
    ap.add_argument("--no-graph", action="store_true", help="callback workload: launch every iteration instead of replaying a "
                    "captured hipGraph (only an iteration without the sentinel check can be captured)")
    ap.add_argument("--no-sentinel", action="store_true", help="callback workload: skip the GLMCMC.py:92-93 redraw check "
                    "(one device->host sync per iteration)")
    ap.add_argument("--rehearse-gloo", action="store_true",
                    help="rehearsal of the multi-rank control flow on ONE GPU: every rank uses cuda:0 and the collectives "
                         "run on gloo with CPU copies (numbers are meaningless; RCCL needs one GPU per rank)")
    ap.add_argument("--workload", default="glmcmc", choices=["glmcmc", "globalmcmc", "glmala", "nf", "gk", "kde", "callback", "glmcmc_nf", "aglmcmc", "rtc", "nf_train", "gamma"],
                    help="glmcmc = BASELINE configs[1] (the headline metric, default); globalmcmc = configs[0]'s "
                         "algorithm batched (gf 0.5); glmala = configs[2] (gf 0.8, N 5, tau 0.3, num_grad 100)")
    args = ap.parse_args()
    if args.workload == "nf":
        return bench_nf(args)
    if args.workload == "nf_train":
        return bench_nf_train(args)
    if args.workload == "kde":
        return bench_kde(args)
    if args.workload == "callback":
        return bench_callback(args)
    if args.workload == "glmcmc_nf":
        return bench_glmcmc_nf(args)
    if args.workload == "aglmcmc":
        return bench_aglmcmc(args)
    if args.workload == "rtc":
        return bench_rtc(args)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d"
                         % (args.gpus, world, args.gpus))
    if args.rehearse_gloo:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = torch.device("cpu") if args.rehearse_gloo else dev           # where collective payloads live
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse_gloo:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from glabcmcmc_amd import engine
    from glabcmcmc_amd.parallel import gather_chain_stats
    model, lp, ip = descriptors(args.workload)
    Dw = 4 if args.workload == "gk" else D
    n, K = args.chains, args.iters
    seed = 20261003

    # synthetic inputs, resident in HBM before the timed region: theta0 = 0 for every chain, y0 = |theta0| + sqrt(0.05) z
    # (SURVEY.md 8d).  They are a function of the GLOBAL chain id -- one generator for the whole job, each rank keeps the rows
    # of its shard -- so that a job gives the same chains however many ranks it is spread over (tests/test_parallel_gloo.py
    # rehearses 4 ranks against 1 and requires identical pooled statistics).
    g = torch.Generator(device="cpu").manual_seed(1234)
    lo, hi = rank * n, (rank + 1) * n
    if args.workload == "gk":
        from glabcmcmc_amd.examples.GK import GK_set
        theta0 = (torch.rand(world * n, 4, generator=g) * 10)[lo:hi].contiguous()
        z = torch.randn(world * n, 8, generator=g)[lo:hi].contiguous()
        y0 = GK_set(0.6).simulate_from_noise(theta0, z)             # CPU tensors: the torch formula of examples/GK.py
    else:
        theta0 = torch.full((n, D), 1.4) if args.workload == "gamma" else torch.zeros(n, D)     # inside the Gamma prior's support
        y0 = theta0.abs() + (0.05 ** 0.5) * torch.randn(world * n, D, generator=g)[lo:hi]
    chains = engine.ChainBatch(theta0, y0, dev, chain0=rank * n)
    engine.init_weights(model, ip, chains)
    hist = None if args.no_history else torch.empty(K, Dw, n, dtype=torch.float32, device=dev)
    mom = engine.Moments(n, Dw, dev)

    step_idx = [0]

    from glabcmcmc_amd import _capi
    gf = {"glmcmc": GF, "globalmcmc": 0.5, "glmala": 0.8, "gk": 0.9, "gamma": 0.9}[args.workload]
    mala = _capi.Mala(0.3, 0.3 ** 2, EPS ** 2, 100, 0)                  # README.md:128
    if args.workload == "glmala":
        chains.add_mala_state()
        engine.glmala_init(model, chains)

    def one_step():
        if args.workload == "glmala":
            engine.run_glmala_steps(model, ip, mala, chains, K, 1 + step_idx[0] * K, seed, gf, NBATCH, history=hist,
                                    moments=mom, steps_per_launch=K, lanes_per_chain=args.lanes)
        else:
            entry = "glabc_globalmcmc_steps" if args.workload == "globalmcmc" else "glabc_glmcmc_steps"
            engine.run_steps(entry, model, lp, ip, chains, K, 1 + step_idx[0] * K, seed, gf, args.batch,
                             history=hist, moments=None if args.no_moments else mom, steps_per_launch=K,
                             lanes_per_chain=args.lanes, debug_flags=args.debug_flags,
                             math_mode=_capi.MATH_FAST if args.fast_math else _capi.MATH_EXACT)
            if args.no_moments:
                mom.steps += K
        step_idx[0] += 1

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    gather_chain_stats(mom, world, via=cdev)           # untimed: RCCL sets up the all-gather's channels, torch loads its
    barrier()                                          # reduction kernels -- once, outside the measurement
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for a, b in ev:
        a.record()              # torch's current stream = the stream run_steps launches on
        one_step()
        b.record()
    stats = gather_chain_stats(mom, world, via=cdev)   # RCCL all-gather of the per-chain sums (no-op at N=1)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    total_steps = float(world) * n * K * args.steps
    value = total_steps / elapsed

    if rank == 0:
        # algorithmic HBM bytes of one launch: state in + out, history rows, moment sums in + out
        Dy = 8 if args.workload == "gk" else D
        S = Dw + Dy + 1 + 2                                # theta, y, log_w, flags, n_moves  (4-byte words)
        tri = Dw * (Dw + 1) // 2
        state_bytes = 2 * 4 * S * n + 2 * 8 * (Dw + 2 * tri) * n
        hist_bytes = 0 if args.no_history else 4 * Dw * n * K
        algo_bytes = state_bytes + hist_bytes
        achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9
        steps_all = mom.steps
        esjd_all = stats["esjd"]
        ok = torch.isfinite(esjd_all)
        # HBM bytes per launch measured with rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, gfx950
        # read correction applied) for exactly this configuration: profiles/r01_f_pmc_summary.json
        # the launch geometry glabc_glmcmc_steps picks (csrc/glabc_hip.hip run_sampler): teams of wavefronts for launches of
        # 16 384 .. 131 072 chains when the caller leaves the geometry to the library
        team_geometry = args.workload in ("glmcmc", "gk", "gamma") and 2 <= args.batch <= 16 and args.lanes == 0 and \
            not (args.debug_flags & 2) and ((args.debug_flags & 4) or 16384 <= n <= 131072)
        traffic, valu = None, None
        p = counted(args.workload + ("_fast" if args.fast_math else ""), n, K, args.batch if args.workload in ("glmcmc", "gk", "gamma") else None)
        if p is not None and args.lanes == 0 and args.debug_flags == 0 and not args.no_history and not args.no_moments:
            valu = valu_block(p, args.workload, kernel_ms)
            if "hbm_traffic_bytes_per_launch" in p:
                traffic = p["hbm_traffic_bytes_per_launch"]["total"]
        out = {
            "metric": "MH accept-steps/sec (whole node) + ESJD, 65 536 chains, Mixture_set dim=2" +
                      (" -- OPT-IN fast-transcendental variant (GLABC_MATH_FAST), not the headline" if args.fast_math else ""),
            "value": value, "unit": "chain-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": {"glmcmc": "GLMCMC iSIR N=%d gf=0.9, Mixture_set eps=0.05 d=2%s"
                                              % (args.batch, " (BASELINE configs[1])" if args.batch == NBATCH else ""),
                                    "globalmcmc": "GlobalMCMC gf=0.5, Mixture_set eps=0.05 d=2 (BASELINE configs[0], batched)",
                                    "glmala": "GLMALA iSIR N=5 gf=0.8 tau=0.3 num_grad=100, Mixture_set eps=0.05 d=2 "
                                              "(BASELINE configs[2])",
                                    "gk": "GLMCMC iSIR N=5 gf=0.9 on the g-and-k model (theta_dim 4, y_dim 8, eps 0.6), "
                                          "chains sharded over the GPUs (BASELINE configs[3])",
                                    "gamma": "GLMCMC iSIR N=%d gf=0.9, Mixture_set eps=0.05 d=2 with a Gamma(2,1) prior and a "
                                             "Gamma(4,3) importance proposal (distribution.py:90-137) in the fused kernel" % args.batch}[args.workload],
                       "chains_per_gpu": n, "iters_per_step": K, "batch_size": args.batch if args.workload in ("glmcmc", "gk", "gamma") else NBATCH,
                       "history": not args.no_history, "math_mode": "fast" if args.fast_math else "exact",
                       "lanes_per_chain": args.lanes or "auto",
                       "parallelism": "chains sharded over %d GPU(s), no data-path collective" % world},
            "esjd_mean": float(esjd_all[ok].double().mean()), "esjd_nan_frac": float(1.0 - ok.double().mean()),
            "mean_theta": stats["mean"], "mean_theta_sq": stats["mean_sq"],
            "analytic": {"mean_theta": 0.0, "mean_theta_sq": 2.081014} if args.workload not in ("gk", "gamma") else None,
            "moment_iters": steps_all,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": {"glmcmc": ("glabc::team_sampler_kernel<D=2, N=%d, %d wavefronts per 64 chains>"
                                               % (args.batch, 3 if n <= 65536 and args.batch >= 3 else 2)
                                               if team_geometry else "glabc::sampler_kernel<GLMCMC, D=2, N=%d>" % args.batch)
                                    if args.batch <= 16 else "glabc::wide_kernel<D=2, L> N=%d" % args.batch,
                                    "globalmcmc": "glabc::global_team_kernel<D=2> (2 wavefronts per 64 chains)" if args.lanes == 0 and
                                    not (args.debug_flags & 2) and ((args.debug_flags & 4) or 16384 <= n <= 131072)
                                    else "glabc::sampler_kernel<GLOBAL, D=2, N=1>",
                                    "glmala": "glabc::glmala_team_kernel<N=5, 2 wavefronts per 64 chains>" if n < 131072 and args.lanes in (0, 2)
                                    else "glabc::glmala_kernel<D=2, N=5>",
                                    "gk": "glabc::team_sampler_kernel<D=4, YD=8, N=%d>" % args.batch if team_geometry
                                    else "glabc::sampler_kernel<GLMCMC, D=4, YD=8, N=%d>" % args.batch,
                                    "gamma": ("glabc::team_sampler_kernel<D=2, N=%d, VAR_GAMMA, %d wavefronts per 64 chains>" % (args.batch, 3 if n <= 65536 and args.batch >= 3 else 2))
                                    if team_geometry else "glabc::sampler_kernel<GLMCMC, D=2, N=%d, VAR_GAMMA>" % args.batch}[args.workload], "kernel_ms": kernel_ms,
                         "algorithmic_bytes_per_launch": algo_bytes,
                         "bytes_per_chain_step": algo_bytes / (n * K),
                         "valu": valu,
                         "note": {"glmcmc": "the step is VALU-bound, not HBM-bound: ~1300 - 1500 vector instructions per chain-"
                                            "step against 8 algorithmic bytes (Philox + Box-Muller + densities); see DESIGN.md",
                                  "globalmcmc": "VALU-bound like the GLMCMC step (one candidate per iteration); see DESIGN.md 4.1",
                                  "glmala": "VALU-bound: a MALA move costs 400 simulations for its finite-difference gradient "
                                            "(~14 000 vector instructions per chain-step on average); see DESIGN.md 4.1b",
                                  "gk": "VALU-bound: 8 g-and-k variates (exp, tanh, pow) + a sort per candidate; see DESIGN.md",
                                  "gamma": "VALU-bound, float64: per candidate two Marsaglia-Tsang rejection loops (double log / sqrt) and four "
                                           "double-precision log(pdf) evaluations; see DESIGN.md"}[args.workload]},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline() if args.workload == "glmcmc" and args.batch == NBATCH and not args.fast_math else \
                cpu_baseline_sampler(args.workload, NBATCH if args.workload in ("globalmcmc", "glmala") else args.batch) \
                if args.workload != "glmcmc" else None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
