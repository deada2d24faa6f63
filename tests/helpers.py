"""Shared helpers for the tests: golden loading and descriptor construction."""
import os

import numpy as np
import torch

from glabcmcmc_amd import distribution
from glabcmcmc_amd.examples.Mixture import Mixture_set

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as f:
        g = {k: f[k] for k in f.files}
    if "cfg" in g:
        g["cfg"] = eval(str(g["cfg"]), {"__builtins__": {}}, {"dict": dict})     # repr() of a plain dict
    return g


def make_dist(spec):
    """('gauss', loc, scale) / ('uniform', low, high) / ('gamma', shape, rate) -> host-mirror distribution object"""
    kind = spec[0]
    if kind == "gamma":
        return distribution.Gamma(torch.tensor(spec[1], dtype=torch.float32), torch.tensor(spec[2], dtype=torch.float32))
    if kind == "gauss":
        return distribution.DiagGaussian(len(spec[1]), torch.tensor(spec[1], dtype=torch.float32),
                                         torch.log(torch.tensor(spec[2], dtype=torch.float32)))
    if kind == "uniform":
        return distribution.Uniform(len(spec[1]), torch.tensor(spec[1], dtype=torch.float32),
                                    torch.tensor(spec[2], dtype=torch.float32))
    raise ValueError(kind)


def descriptors(cfg, consts=None):
    """(model, local, global) descriptors of a test configuration.  `consts` = a golden fixture: the
    host-computed float32 constants (exp(log_scale), log(eps) ...) are then taken from the fixture, i.e.
    the values the reference computed on the machine that generated the golden chains -- torch's CPU
    log / exp differ in the last bits between CPU types, and GLMALA's chains depend on them."""
    gk = cfg.get("model") == "gk"
    if gk:
        from glabcmcmc_amd.examples.GK import GK_set
        model = GK_set(cfg["epsilon"]).descriptor()
    else:
        model = Mixture_set(cfg["epsilon"]).descriptor()
    local, glob = make_dist(cfg["local"]).descriptor(), make_dist(cfg["global"]).descriptor()
    if consts is not None:
        for j in range(0 if gk else 2):
            model.noise.p1[j] = float(consts["c_noise_log_scale"][j])
            model.noise.p2[j] = float(consts["c_noise_scale"][j])
        model.kern_log_scale = float(consts["c_kern_log_scale"][0])
        model.kern_scale = float(consts["c_kern_scale"][0])
        for d, tag in ((local, "local"), (glob, "global")):
            if "c_%s_p1" % tag in consts:
                for j in range(d.dim):
                    d.p1[j] = float(consts["c_%s_p1" % tag][j])
                    d.p2[j] = float(consts["c_%s_p2" % tag][j])
    return model, local, glob


SAMPLER_GOLDENS = [
    "glmcmc_philox_bench", "glmcmc_philox_n8", "glmcmc_philox_n1", "glmcmc_philox_n3", "glmcmc_philox_n16",
    "glmcmc_philox_n32", "glmcmc_philox_n100",
    "glmcmc_philox_uniform", "globalmcmc_philox_bench", "globalmcmc_philox_wide",
    "glmcmc_tape_small", "globalmcmc_tape_small",
    "glmcmc_philox_gk", "glmcmc_philox_gk_gauss", "globalmcmc_philox_gk",
]

# reference run with a correctly rounded torch.sqrt (see make_golden.py): bit parity expected
GLMALA_GOLDENS_EXACT = ["glmala_philox_bench_ieee", "glmala_philox_local_ieee", "glmala_philox_uniform_ieee",
                        "glmala_philox_allglobal"]
# reference run as-is (torch.sqrt = MKL VML, 1 ulp low for 0.65 % of float32 inputs): parity until the
# first sqrt-ulp event of a chain, which the float32 finite-difference prior gradient then amplifies
GLMALA_GOLDENS_MKL = ["glmala_philox_bench", "glmala_philox_local", "glmala_philox_alllocal", "glmala_philox_uniform"]
GLMALA_GOLDENS = GLMALA_GOLDENS_EXACT + GLMALA_GOLDENS_MKL


def mala_params(cfg):
    """glabc_mala with tau**2 and epsilon**2 evaluated in Python floats, as GLMALA.py:43,90 do"""
    from glabcmcmc_amd import _capi as A
    return A.Mala(float(cfg["tau"]), float(cfg["tau"]) ** 2, float(cfg["epsilon"]) ** 2, int(cfg["num_grad"]), 0)


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
