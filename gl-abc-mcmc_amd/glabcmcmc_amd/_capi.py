"""ctypes declarations of include/glabc.h and the loader of the HIP library.

The structures below are the Python spelling of the C ABI in ``include/glabc.h``
(field for field, same order).  ``lib()`` loads ``libglabc_hip.so`` -- the
hand-written gfx950 kernels plus their C entry points, built in-tree by
``__graft_entry__.build()`` -- and raises if it is missing: there is no CPU or
PyTorch fallback for the sampler hot path.
"""
import ctypes as C
import os

MAX_DIM = 8
MAX_BATCH = 16            # register kernels (and every sampler but GLMCMC)
MAX_BATCH_WIDE = 4096      # glabc_glmcmc_steps: lane groups of a wavefront share a chain's proposals

DIST_DIAG_GAUSS = 0
DIST_UNIFORM = 1
DIST_GAMMA = 2

SIM_ABS_GAUSS = 0
SIM_GK = 1
SIM_USER = 2

RTC_PRIOR_LOG_PROB, RTC_DISCREPANCY, RTC_LOG_KERNEL = 0, 1, 2      # glabc_rtc_model_rows `what`
VERSION = 301                  # include/glabc.h GLABC_VERSION
STREAM_LAYOUT = 2              # include/glabc.h GLABC_STREAM_LAYOUT

FLAG_LOCAL = 1
DEBUG_EXACT_INDEX = 1          # glabc_run.debug_flags
DEBUG_NO_TEAM = 2              # glabc_glmcmc_steps: never / always the two-wavefront team geometry (csrc/glabc_team.h)
DEBUG_TEAM = 4
DEBUG_DEFAULT_SCHEDULE = 8     # one lane per chain: the default-schedule objects also for small launches
FLAG_TH64 = 2
FLAG_LW64 = 4
FLAG_HAS_GRAD = 8

OK = 0

_f8 = C.c_float * MAX_DIM


class Dist(C.Structure):
    """struct glabc_dist"""
    _fields_ = [
        ("kind", C.c_int32),
        ("dim", C.c_int32),
        ("p0", _f8),
        ("p1", _f8),
        ("p2", _f8),
        ("p3", _f8),
        ("c0", C.c_float),
    ]


class Model(C.Structure):
    """struct glabc_model"""
    _fields_ = [
        ("sim_kind", C.c_int32),
        ("theta_dim", C.c_int32),
        ("y_dim", C.c_int32),
        ("gk_c", C.c_float),
        ("prior", Dist),
        ("noise", Dist),
        ("y_obs", _f8),
        ("kern_log_scale", C.c_float),
        ("kern_scale", C.c_float),
        ("kern_c0", C.c_float),
        ("epsilon", C.c_float),
    ]


class Chains(C.Structure):
    """struct glabc_chains (device pointers as integers)"""
    _fields_ = [
        ("n_chains", C.c_int64),
        ("chain0", C.c_int64),
        ("stride", C.c_int64),
        ("theta", C.c_void_p),
        ("y", C.c_void_p),
        ("log_w", C.c_void_p),
        ("flags", C.c_void_p),
        ("n_moves", C.c_void_p),
        ("theta64", C.c_void_p),
        ("y64", C.c_void_p),
        ("log_w64", C.c_void_p),
        ("grad", C.c_void_p),
    ]


class Mala(C.Structure):
    """struct glabc_mala"""
    _fields_ = [
        ("tau", C.c_double),
        ("tau_sq", C.c_double),
        ("eps_sq", C.c_double),
        ("num_grad", C.c_int32),
        ("reserved", C.c_int32),
    ]


NF_HIDDEN = 128
NF_COUPLING_FLOATS = 128 * 128 + 128 + 128 + 4 * 128 + 4


class Flow(C.Structure):
    """struct glabc_flow"""
    _fields_ = [
        ("n_couplings", C.c_int32),
        ("hidden", C.c_int32),
        ("params", C.c_void_p),
        ("base_loc", C.c_float * 2),
        ("base_log_scale", C.c_float * 2),
        ("base_scale", C.c_float * 2),
        ("base_c0", C.c_float),
        ("reserved", C.c_int32),
    ]


class Pool(C.Structure):
    """struct glabc_pool"""
    _fields_ = [
        ("theta", C.c_void_p),
        ("x", C.c_void_p),
        ("w", C.c_void_p),
        ("log_q_old", C.c_void_p),
        ("kk", C.c_void_p),
        ("step_size", C.c_int32),
        ("reserved", C.c_int32),
        ("moved_idx", C.c_void_p),
        ("n_moved", C.c_void_p),
        ("n_moved_reset", C.c_void_p),
    ]


class Kde(C.Structure):
    """struct glabc_kde"""
    _fields_ = [
        ("dim", C.c_int32),
        ("reserved", C.c_int32),
        ("n_samples", C.c_int64),
        ("x", C.c_void_p),
        ("log_w", C.c_void_p),
        ("cum_q", C.c_void_p),
        ("bandwidth", C.c_float * MAX_DIM),
        ("sum_log_bw", C.c_float),
        ("c_2pi", C.c_float),
    ]


class GammaDesc(C.Structure):
    """struct glabc_gamma"""
    _fields_ = [
        ("dim", C.c_int32),
        ("reserved", C.c_int32),
        ("shape", C.c_double * MAX_DIM),
        ("scale", C.c_double * MAX_DIM),
        ("gammaln", C.c_double * MAX_DIM),
    ]


class Moments(C.Structure):
    """struct glabc_moments"""
    _fields_ = [
        ("sum_theta", C.c_void_p),
        ("sum_outer", C.c_void_p),
        ("sum_jump", C.c_void_p),
    ]


class Tape(C.Structure):
    """struct glabc_tape"""
    _fields_ = [
        ("u", C.c_void_p),
        ("r", C.c_void_p),
        ("z", C.c_void_p),
        ("n_prop", C.c_int32),
        ("reserved", C.c_int32),
    ]


class StepIO(C.Structure):
    """struct glabc_step_io"""
    _fields_ = [
        ("n_prop", C.c_int32),
        ("theta_dim", C.c_int32),
        ("y_dim", C.c_int32),
        ("noise_dim", C.c_int32),
        ("theta_prop", C.c_void_p),
        ("log_q", C.c_void_p),
        ("sim_noise", C.c_void_p),
        ("log_u", C.c_void_p),
        ("u_res", C.c_void_p),
        ("is_global", C.c_void_p),
        ("y_prop", C.c_void_p),
        ("prior_prop", C.c_void_p),
        ("kern_prop", C.c_void_p),
        ("prior_cur", C.c_void_p),
        ("kern_cur", C.c_void_p),
        ("q_cur", C.c_void_p),
        ("n_valid", C.c_void_p),
    ]


ALGO_GLMCMC = 0
ALGO_GLOBALMCMC = 1
ALGO_GLMALA = 2
SLOT_REDRAW = 0x40000000


class DrawsOut(C.Structure):
    """struct glabc_draws_out"""
    _fields_ = [("u", C.c_void_p), ("r", C.c_void_p), ("z", C.c_void_p)]


MATH_EXACT, MATH_FAST = 0, 1       # glabc_run.math_mode


class Run(C.Structure):
    """struct glabc_run"""
    _fields_ = [
        ("seed", C.c_uint64),
        ("step0", C.c_uint32),
        ("n_steps", C.c_int32),
        ("global_frequency", C.c_float),
        ("batch_size", C.c_int32),
        ("history", C.c_void_p),
        ("hist_stride", C.c_int64),
        ("moments", C.POINTER(Moments)),
        ("tape", C.POINTER(Tape)),
        ("lanes_per_chain", C.c_int32),
        ("debug_flags", C.c_int32),
        ("step0_device", C.c_void_p),
        ("global_frequency_per_chain", C.c_void_p),
        ("math_mode", C.c_int32),
        ("reserved", C.c_int32),
        ("dump_draws", C.POINTER(DrawsOut)),
    ]


_P = C.POINTER

# name -> (restype, argtypes); the same table describes oracle_* twins minus the stream
ENTRY_POINTS = {
    "glabc_glmcmc_steps": (C.c_int, [_P(Model), _P(Dist), _P(Dist), _P(Chains), _P(Run), C.c_void_p]),
    "glabc_globalmcmc_steps": (C.c_int, [_P(Model), _P(Dist), _P(Dist), _P(Chains), _P(Run), C.c_void_p]),
    "glabc_glmala_steps": (C.c_int, [_P(Model), _P(Dist), _P(Mala), _P(Chains), _P(Run), C.c_void_p]),
    "glabc_glmala_init": (C.c_int, [_P(Model), _P(Chains), C.c_void_p]),
    "glabc_nf_sample": (C.c_int, [_P(Flow), C.c_void_p, C.c_uint64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "glabc_nf_log_prob": (C.c_int, [_P(Flow), C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "glabc_nf_log_prob_indexed": (C.c_int, [_P(Flow), C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                            C.c_void_p]),
    "glabc_nf_inverse": (C.c_int, [_P(Flow), C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "glabc_nf_grad_workspace": (C.c_int, [C.c_int32, C.c_int64, _P(C.c_int64)]),
    "glabc_nf_grad": (C.c_int, [_P(Flow), C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_void_p]),
    "glabc_adam_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_double, C.c_double, C.c_double,
                                  C.c_double, C.c_double, C.c_int32, C.c_void_p]),
    "glabc_pool_weights": (C.c_int, [_P(Model), C.c_void_p, C.c_void_p, C.c_int64, C.c_uint64, C.c_int64, C.c_void_p,
                                     C.c_void_p, C.c_void_p]),
    "glabc_glmcmc_nf_step": (C.c_int, [_P(Model), _P(Dist), _P(Pool), _P(Chains), _P(Run), C.c_void_p]),
    "glabc_kde_fit": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_void_p, C.c_void_p, C.c_void_p]),
    "glabc_kde_log_prob": (C.c_int, [_P(Kde), C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "glabc_kde_log_prob_indexed": (C.c_int, [_P(Kde), C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                             C.c_void_p]),
    "glabc_kde_sample": (C.c_int, [_P(Kde), C.c_int64, C.c_uint64, C.c_int64, C.c_void_p, C.c_void_p]),
    "glabc_kde_train_weights": (C.c_int, [_P(Model), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "glabc_init_weights": (C.c_int, [_P(Model), _P(Dist), _P(Chains), C.c_void_p]),
    "glabc_rtc_compile": (C.c_int, [C.c_char_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P(C.c_void_p), C.c_char_p,
                                    C.c_int64]),
    "glabc_rtc_steps": (C.c_int, [C.c_void_p, _P(Model), _P(Dist), _P(Dist), _P(Chains), _P(Run), C.c_void_p]),
    "glabc_rtc_simulate": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "glabc_rtc_release": (None, [C.c_void_p]),
    "glabc_rtc_hooks": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "glabc_rtc_model_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "glabc_propose": (C.c_int, [C.c_int, _P(Dist), _P(Dist), _P(Chains), _P(Run), _P(StepIO), C.c_void_p]),
    "glabc_propose_redraw": (C.c_int, [_P(Dist), _P(Chains), _P(Run), _P(StepIO), C.c_int32, C.c_void_p, C.c_void_p]),
    "glabc_select": (C.c_int, [C.c_int, _P(Dist), _P(Chains), _P(Run), _P(StepIO), C.c_void_p]),
    "glabc_model_simulate": (C.c_int, [_P(Model), C.c_void_p, C.c_void_p, C.c_int64, C.c_uint64, C.c_int64, C.c_void_p,
                                       C.c_void_p]),
    "glabc_dist_log_prob": (C.c_int, [_P(Dist), C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "glabc_dist_forward": (C.c_int, [_P(Dist), C.c_int64, C.c_uint64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "glabc_gamma_log_prob": (C.c_int, [_P(GammaDesc), C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "glabc_gamma_forward": (C.c_int, [_P(GammaDesc), C.c_int64, C.c_uint64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "glabc_model_prior_log_prob": (C.c_int, [_P(Model), C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "glabc_model_discrepancy": (C.c_int, [_P(Model), C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "glabc_model_log_kernel": (C.c_int, [_P(Model), C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "glabc_esjd": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]),
    "glabc_moments_esjd": (C.c_int, [_P(Moments), C.c_int64, C.c_int32, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]),
    "glabc_selftest_numerics": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "glabc_selftest_sqrt": (C.c_int, [C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]),
    "glabc_selftest_rowsum": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "glabc_version": (C.c_int, []),
    "glabc_stream_layout": (C.c_int, []),
    "glabc_status_string": (C.c_char_p, [C.c_int]),
    "glabc_last_hip_error": (C.c_int, []),
}

LIB_NAME = "libglabc_hip.so"
_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
# GLABC_HIP_LIB: load another build of the same ABI (kernel experiments); default = the in-tree build
LIB_PATH = os.environ.get("GLABC_HIP_LIB") or os.path.join(os.path.dirname(_PKG_DIR), "csrc", LIB_NAME)

_lib = None


class HipLibraryMissing(RuntimeError):
    pass


def bind(path):
    """ctypes handle of a build of the C-ABI library with every entry point's signature set; refuses another ABI version"""
    handle = C.CDLL(path)
    for name, (res, args) in ENTRY_POINTS.items():
        fn = getattr(handle, name)     # AttributeError here = ABI mismatch, deliberately loud
        fn.restype = res
        fn.argtypes = args
    got = handle.glabc_version()
    if got != VERSION:                 # a stale library would misread the argument structs (include/glabc.h GLABC_VERSION)
        raise HipLibraryMissing("%s is version %d, this binding is for %d: rebuild it (python __graft_entry__.py build)"
                                % (path, got, VERSION))
    return handle


def lib():
    """The loaded C-ABI library; raises HipLibraryMissing if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipLibraryMissing(
                "%s not found: build it with `python __graft_entry__.py build` "
                "(hipcc --offload-arch=gfx950). The sampler hot path has no CPU fallback." % LIB_PATH)
        _lib = bind(LIB_PATH)
    return _lib


def check(status, what):
    if status != OK:
        msg = lib().glabc_status_string(status)
        raise RuntimeError("%s failed: %s (status %d, hipError %d)" % (
            what, msg.decode() if msg else "?", status, lib().glabc_last_hip_error()))
