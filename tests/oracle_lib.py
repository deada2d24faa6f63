"""ctypes binding of oracle/libglabc_oracle.so (the CPU checker) for the tests.

Test infrastructure.  The structures are the product's declarations of
include/glabc.h (glabcmcmc_amd._capi); the oracle's entry points take the same
structs with HOST pointers and no stream argument.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from glabcmcmc_amd import _capi as A

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "libglabc_oracle.so")

_P = C.POINTER
_SIG = {
    "oracle_aten_rowsum_f32": (C.c_float, [C.c_void_p, C.c_int]),
    "oracle_dist_log_prob": (C.c_int, [_P(A.Dist), C.c_void_p, C.c_int64, C.c_void_p]),
    "oracle_dist_forward": (C.c_int, [_P(A.Dist), C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "oracle_model_prior_log_prob": (C.c_int, [_P(A.Model), C.c_void_p, C.c_int64, C.c_void_p]),
    "oracle_model_discrepancy": (C.c_int, [_P(A.Model), C.c_void_p, C.c_int64, C.c_void_p]),
    "oracle_model_log_kernel": (C.c_int, [_P(A.Model), C.c_void_p, C.c_int64, C.c_void_p]),
    "oracle_model_simulate": (C.c_int, [_P(A.Model), C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "oracle_init_weights": (C.c_int, [_P(A.Model), _P(A.Dist), _P(A.Chains)]),
    "oracle_glmcmc_steps": (C.c_int, [_P(A.Model), _P(A.Dist), _P(A.Dist), _P(A.Chains), _P(A.Run)]),
    "oracle_globalmcmc_steps": (C.c_int, [_P(A.Model), _P(A.Dist), _P(A.Dist), _P(A.Chains), _P(A.Run)]),
    "oracle_aten_rowsum_f64": (C.c_double, [C.c_void_p, C.c_int]),
    "oracle_grad_noise": (None, [C.c_uint64, C.c_uint64, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "oracle_numerical_gradient": (C.c_int, [_P(A.Model), _P(A.Mala), C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32,
                                            C.c_int, C.c_void_p]),
    "oracle_glmala_init": (C.c_int, [_P(A.Model), _P(A.Chains)]),
    "oracle_glmala_steps": (C.c_int, [_P(A.Model), _P(A.Dist), _P(A.Mala), _P(A.Chains), _P(A.Run)]),
    "oracle_exp_v": (None, [C.c_void_p, C.c_int64, C.c_void_p]),
    "oracle_log_v": (None, [C.c_void_p, C.c_int64, C.c_void_p]),
    "oracle_nf_sample": (C.c_int, [_P(A.Flow), C.c_void_p, C.c_uint64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]),
    "oracle_nf_log_prob": (C.c_int, [_P(A.Flow), C.c_void_p, C.c_int64, C.c_void_p]),
    "oracle_nf_grad": (C.c_int, [_P(A.Flow), C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "oracle_adam_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_double, C.c_double, C.c_double,
                                   C.c_double, C.c_double, C.c_int32]),
    "oracle_pool_weights": (C.c_int, [_P(A.Model), C.c_void_p, C.c_void_p, C.c_int64, C.c_uint64, C.c_int64, C.c_void_p,
                                      C.c_void_p]),
    "oracle_glmcmc_nf_step": (C.c_int, [_P(A.Model), _P(A.Dist), _P(A.Pool), _P(A.Chains), _P(A.Run)]),
    "oracle_kde_fit": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_void_p]),
    "oracle_kde_log_prob": (C.c_int, [_P(A.Kde), C.c_void_p, C.c_int64, C.c_void_p]),
    "oracle_kde_sample": (C.c_int, [_P(A.Kde), C.c_int64, C.c_uint64, C.c_int64, C.c_void_p]),
    "oracle_kde_draws": (None, [C.c_uint64, C.c_int64, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]),
    "oracle_kde_train_weights": (C.c_int, [_P(A.Model), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "oracle_dist_forward_philox": (C.c_int, [_P(A.Dist), C.c_int64, C.c_uint64, C.c_int64, C.c_void_p, C.c_void_p]),
    "oracle_gamma_log_prob": (C.c_int, [_P(A.GammaDesc), C.c_void_p, C.c_int64, C.c_void_p]),
    "oracle_gamma_forward": (C.c_int, [_P(A.GammaDesc), C.c_int64, C.c_uint64, C.c_int64, C.c_void_p, C.c_void_p]),
    "oracle_esjd": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int64, C.c_int64, C.c_void_p]),
    "oracle_philox4x32_10": (None, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "oracle_expf_v": (None, [C.c_void_p, C.c_int64, C.c_void_p]),
    "oracle_fx_quantize_v": (None, [C.c_void_p, C.c_int64, C.c_void_p]),
    "oracle_fx_both": (None, [C.c_void_p, C.c_int64, C.c_void_p]),
    "oracle_expf_b_v": (None, [C.c_void_p, C.c_int64, C.c_void_p]),
    "oracle_logf_v": (None, [C.c_void_p, C.c_int64, C.c_void_p]),
    "oracle_sincos2pi_v": (None, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "oracle_normal_pair_v": (None, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "oracle_uniforms_v": (None, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "oracle_propose": (C.c_int, [C.c_int, _P(A.Dist), _P(A.Dist), _P(A.Chains), _P(A.Run), _P(A.StepIO)]),
    "oracle_propose_redraw": (C.c_int, [_P(A.Dist), _P(A.Chains), _P(A.Run), _P(A.StepIO), C.c_int32, C.c_void_p]),
    "oracle_select": (C.c_int, [C.c_int, _P(A.Dist), _P(A.Chains), _P(A.Run), _P(A.StepIO)]),
    "oracle_step_draws": (None, [C.c_uint64, C.c_uint64, C.c_uint32, C.c_int, C.c_int, C.c_int,
                                 C.c_void_p, C.c_void_p, C.c_void_p]),
}

_lib = None


def load():
    global _lib
    if _lib is None:
        # always through make: a checker older than its source must not be what the tests trust (on the GPU box the
        # prebuilt library travels with the snapshot and make finds it up to date)
        subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])
        # GLABC_ORACLE_LIB: another build of the same checker (e.g. one compiled with -fsanitize=address,undefined, run with
        # LD_PRELOAD=libasan.so -- the CPU suite then walks every checker path under the sanitizers)
        h = C.CDLL(os.environ.get("GLABC_ORACLE_LIB") or LIB_PATH)
        for name, (res, args) in _SIG.items():
            fn = getattr(h, name)
            fn.restype = res
            fn.argtypes = args
        _lib = h
    return _lib


def ptr(a):
    """address of a C-contiguous numpy array (None -> NULL)"""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data


class HostChains:
    """Chain state as numpy arrays in the chain-major layout of glabc_chains."""

    def __init__(self, theta, y, chain0=0, with_isir=True):
        theta = np.asarray(theta, np.float32)
        y = np.asarray(y, np.float32)
        n = theta.shape[0]
        self.n = n
        self.theta = np.array(theta.T, dtype=np.float32, order="C", copy=True)      # [d][n]; never an alias of the caller's array
        self.y = np.array(y.T, dtype=np.float32, order="C", copy=True)
        self.log_w = np.zeros(n, np.float32) if with_isir else None
        self.flags = np.full(n, A.FLAG_LOCAL, np.uint32) if with_isir else None
        self.n_moves = np.zeros(n, np.uint32)
        self.chain0 = chain0

    def struct(self):
        return A.Chains(self.n, self.chain0, self.n, ptr(self.theta), ptr(self.y), ptr(self.log_w),
                        ptr(self.flags), ptr(self.n_moves), ptr(getattr(self, "theta64", None)),
                        ptr(getattr(self, "y64", None)), ptr(getattr(self, "log_w64", None)),
                        ptr(getattr(self, "grad", None)))

    def add_mala_state(self):
        """the float64 state arrays of GLMALA (glabc_chains.theta64 ...)"""
        self.theta64 = np.zeros_like(self.theta, dtype=np.float64)
        self.y64 = np.zeros_like(self.y, dtype=np.float64)
        self.log_w64 = np.zeros(self.n, np.float64)
        self.grad = np.zeros_like(self.theta, dtype=np.float64)
        return self


class HostMoments:
    def __init__(self, n, d):
        tri = d * (d + 1) // 2
        self.sum_theta = np.zeros((d, n), np.float64)
        self.sum_outer = np.zeros((tri, n), np.float64)
        self.sum_jump = np.zeros((tri, n), np.float64)

    def struct(self):
        return A.Moments(ptr(self.sum_theta), ptr(self.sum_outer), ptr(self.sum_jump))


def make_run(seed=0, step0=1, n_steps=0, gf=0.0, batch=1, history=None, moments=None, tape=None):
    """glabc_run for host arrays; returns (struct, keepalive)."""
    keep = [history, moments, tape]
    m = moments.struct() if moments is not None else None
    t = None
    if tape is not None:
        u, r, z, n_prop = tape
        t = A.Tape(ptr(u), ptr(r), ptr(z), n_prop, 0)
    keep += [m, t]
    run = A.Run(seed, step0, n_steps, gf, batch, ptr(history),
                history.shape[-1] if history is not None else 0,
                C.pointer(m) if m is not None else None,
                C.pointer(t) if t is not None else None)
    return run, keep
