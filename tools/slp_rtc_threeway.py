"""Diagnostic (GPU box): where does the SLP-compiled run-time program differ?  For batch sizes 11 / 12 / 16 of the configuration
of tests/test_rtc.py::test_hip_self_check_refuses_a_miscompiled_kernel, with the vectorizer switched back ON for the run-time
compile (GLABC_RTC_OPTS) and the self-check off: fused run-time kernel vs split-phase path vs CPU checker (host-compiled
simulator), chains that differ pairwise.   python tools/slp_rtc_threeway.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gl-abc-mcmc_amd"), os.path.join(ROOT, "tests")]
os.environ["GLABC_RTC_LANES"] = "1"
os.environ["GLABC_RTC_SELF_CHECK"] = "0"
import glabcmcmc_amd as g_                      # noqa: E402
import oracle_lib                               # noqa: E402
from glabcmcmc_amd import _capi as A            # noqa: E402
from helpers import bits, make_dist             # noqa: E402
from test_rtc import NONLINEAR, host_simulator, user_model_desc   # noqa: E402

oracle = oracle_lib.load()
keep, fn = host_simulator(NONLINEAR, 3, 2, 4)
oracle.oracle_set_user_simulator(fn)
prior = make_dist(("gauss", [0.0, 0.5, 0.0], [1.5, 1.0, 2.0]))
local = make_dist(("gauss", [0.0, 0.0, 0.0], [0.45, 0.3, 0.6]))
n, T, seed = 512, 4, 20240229
gen = torch.Generator().manual_seed(seed)
theta0 = torch.tensor([0.0, 0.5, 0.0]) + torch.tensor([1.5, 1.0, 2.0]) * torch.randn(n, 3, generator=gen)
for opts in ("-fslp-vectorize", ""):
    os.environ["GLABC_RTC_OPTS"] = opts
    for N in (11, 12, 16):
        cm = g_.CompiledModel(3, 2, NONLINEAR, prior, [0.9, 0.6], 0.15, noise_dim=4)
        y0 = cm.simulate_from_noise(theta0, torch.randn(n, 4, generator=gen)).cpu()
        outs = {}
        for path, kw in (("fused", {}), ("generic", dict(sentinel_redraw=False, graph=False))):
            outs[path] = g_.GLMCMC(cm, T + 1, theta0, y0, local, None, 0.5, prior, N, seed=seed, verbose=False, path=path, **kw).numpy()
        m = user_model_desc(cm.descriptor(), 4)
        hc = oracle_lib.HostChains(theta0.numpy().copy(), y0.numpy().copy())
        hh = np.zeros((T, 3, n), np.float32)
        run, k2 = oracle_lib.make_run(seed=seed, step0=1, n_steps=T, gf=0.5, batch=N, history=hh)
        cs = hc.struct()
        ld, pd = local.descriptor(), prior.descriptor()
        assert oracle.oracle_init_weights(C.byref(m), C.byref(pd), C.byref(cs)) == 0
        assert oracle.oracle_glmcmc_steps(C.byref(m), C.byref(ld), C.byref(pd), C.byref(cs), C.byref(run)) == 0
        orc = np.concatenate([theta0.numpy()[None], hh.transpose(0, 2, 1)], axis=0)
        d = lambda a, b: int((bits(a) != bits(b)).any(axis=(0, 2)).sum())      # noqa: E731
        print("opts %-16r N %2d: fused!=generic %3d  fused!=oracle %3d  generic!=oracle %3d chains of %d"
              % (opts, N, d(outs["fused"], outs["generic"]), d(outs["fused"], orc), d(outs["generic"], orc), n), flush=True)
