"""Device-side chain state and the launch loop shared by the sampler functions.

Everything here is plumbing around the C ABI (include/glabc.h): torch is used for
device memory and streams only.  The layout in HBM is chain-major
(structure-of-arrays, chain index innermost):

    theta   float32 [d][C]        y      float32 [y_dim][C]
    log_w   float32 [C]           flags  uint32  [C]        n_moves uint32 [C]
    history float32 [T][d][C]     (row t = Theta_Re[step0 + t])
    moments float64 [d][C], [tri(d)][C], [tri(d)][C]

so each wavefront touches 256 contiguous bytes per component, and a GPU's shard
of a multi-GPU run is a contiguous range of global chain ids [chain0, chain0+C).
"""
import ctypes as C

import torch

from . import _capi

MAX_STEPS_PER_LAUNCH = 1 << 16


def require_device(device=None):
    if not torch.cuda.is_available():
        raise RuntimeError("glabcmcmc_amd samplers run on an MI355X through libglabc_hip.so; "
                           "no HIP device is visible and there is no CPU fallback")
    dev = torch.device("cuda" if device is None else device)
    if dev.type != "cuda":
        raise RuntimeError("glabcmcmc_amd samplers need a cuda (HIP) device, got %r" % (device,))
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    return dev


def draw_seed(seed):
    """Philox key of a run.  None -> 62 bits from torch's global generator, so that
    torch.manual_seed(...) makes runs reproducible as it does for the reference."""
    if seed is None:
        return int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())
    return int(seed) & 0xFFFFFFFFFFFFFFFF


class ChainBatch:
    """State of C chains on one GPU (glabc_chains)."""

    def __init__(self, theta0, y0, device, chain0=0):
        theta0 = torch.as_tensor(theta0, dtype=torch.float32)
        y0 = torch.as_tensor(y0, dtype=torch.float32)
        if theta0.dim() == 1:
            theta0 = theta0.view(1, -1)
        y0 = y0.reshape(-1, y0.shape[-1]) if y0.dim() > 1 else y0.view(1, -1)
        if y0.shape[0] == 1 and theta0.shape[0] > 1:
            y0 = y0.expand(theta0.shape[0], -1)
        if y0.shape[0] != theta0.shape[0]:
            raise ValueError("Initial_y has %d rows for %d chains" % (y0.shape[0], theta0.shape[0]))
        self.device = device
        self.n = theta0.shape[0]
        self.d = theta0.shape[1]
        self.yd = y0.shape[1]
        self.chain0 = int(chain0)
        self.theta = theta0.t().contiguous().to(device)            # [d][C]
        self.y = y0.t().contiguous().to(device)                    # [yd][C]
        self.log_w = torch.zeros(self.n, dtype=torch.float32, device=device)
        self.flags = torch.full((self.n,), _capi.FLAG_LOCAL, dtype=torch.int32, device=device)
        self.n_moves = torch.zeros(self.n, dtype=torch.int32, device=device)

        self.theta64 = self.y64 = self.log_w64 = self.grad = None

    def add_mala_state(self):
        """float64 state arrays of GLMALA (glabc_chains.theta64 / y64 / log_w64 / grad)"""
        self.theta64 = torch.zeros(self.d, self.n, dtype=torch.float64, device=self.device)
        self.y64 = torch.zeros(self.yd, self.n, dtype=torch.float64, device=self.device)
        self.log_w64 = torch.zeros(self.n, dtype=torch.float64, device=self.device)
        self.grad = torch.zeros(self.d, self.n, dtype=torch.float64, device=self.device)
        return self

    def struct(self):
        ptr = lambda t: None if t is None else t.data_ptr()          # noqa: E731
        return _capi.Chains(self.n, self.chain0, self.n, self.theta.data_ptr(), self.y.data_ptr(),
                            self.log_w.data_ptr(), self.flags.data_ptr(), self.n_moves.data_ptr(),
                            ptr(self.theta64), ptr(self.y64), ptr(self.log_w64), ptr(self.grad))

    def theta_rows(self):
        """(C, d) view of the current states"""
        return self.theta.t()


class Moments:
    """Per-chain streaming sums (glabc_moments) and what is derived from them."""

    def __init__(self, n, d, device):
        tri = d * (d + 1) // 2
        self.n, self.d, self.steps = n, d, 0
        self.sum_theta = torch.zeros(d, n, dtype=torch.float64, device=device)
        self.sum_outer = torch.zeros(tri, n, dtype=torch.float64, device=device)
        self.sum_jump = torch.zeros(tri, n, dtype=torch.float64, device=device)

    def struct(self):
        return _capi.Moments(self.sum_theta.data_ptr(), self.sum_outer.data_ptr(), self.sum_jump.data_ptr())

    def _full(self, tri_rows):
        d = self.d
        m = torch.empty(self.n, d, d, dtype=torch.float64, device=tri_rows.device)
        k = 0
        for p in range(d):
            for q in range(p, d):
                m[:, p, q] = tri_rows[k]
                m[:, q, p] = tri_rows[k]
                k += 1
        return m

    def esjd(self):
        """ESJD.py:21-24 per chain from the streamed jump sums: det(sum dd^T / n)^(1/d)."""
        out = torch.empty(self.n, dtype=torch.float32, device=self.sum_jump.device)
        ms = self.struct()
        stream = torch.cuda.current_stream(out.device).cuda_stream
        with torch.cuda.device(out.device):
            _capi.check(_capi.lib().glabc_moments_esjd(C.byref(ms), self.steps, self.d, self.n, self.n, out.data_ptr(),
                                                       C.c_void_p(stream)), "glabc_moments_esjd")
        return out

    def mean(self):
        return (self.sum_theta / float(self.steps)).t()

    def second_moment(self):
        return self._full(self.sum_outer) / float(self.steps)

    def packed(self):
        """One row per chain [sum_theta | sum_outer | sum_jump] for an all-gather."""
        return torch.cat([self.sum_theta, self.sum_outer, self.sum_jump], dim=0).t().contiguous()


def run_steps(entry, model_desc, local_desc, global_desc, chains, n_steps, step0, seed, global_frequency, batch_size,
              history=None, moments=None, steps_per_launch=None, lanes_per_chain=0, debug_flags=0, gf_per_chain=None,
              rtc_program=None, math_mode=0, dump_draws=None, mirror=None):
    """Advance `chains` by n_steps iterations with the C-ABI entry point `entry`
    ('glabc_glmcmc_steps' / 'glabc_globalmcmc_steps'), K iterations per launch.

    history: None or float32 tensor [n_steps][d][C] on the chains' device.
    lanes_per_chain: 0 = let the library choose from the chain count; 1 / 2 / 4 force the
    launch geometry (results are identical for every choice).
    gf_per_chain: None or float32 device tensor [C] replacing global_frequency chain by chain.
    rtc_program: handle of glabc_rtc_compile -- the launches then go to glabc_rtc_steps (a Model whose simulator was compiled
    into the kernel at run time, compiled.CompiledModel) with the same arguments.
    math_mode: _capi.MATH_EXACT (default: the specified arithmetic, reproduced bit for bit by the CPU checker) or _capi.MATH_FAST
    (glabc_glmcmc_steps only, opt-in: hardware transcendentals -- the same law from another stream of normals, include/glabc.h).
    dump_draws: MATH_FAST only -- (u [C][n_steps][2] float32, r [C][n_steps] float64, z [C][n_steps][N][d + y_dim] float32) device
    tensors that receive the draws the kernel used, in the layout of glabc_tape (one launch: steps_per_launch >= n_steps).
    mirror: None or the _host.HostMirror of the tensor `history` is rows 1.. of: every launch's rows are copied to pinned host
    memory on a second stream while the next launch runs (launches are then cut to mirror.steps_per_launch rows).
    """
    lib = _capi.lib()
    if rtc_program is not None:
        entry = "glabc_rtc_steps"

        def fn(*a):
            return lib.glabc_rtc_steps(rtc_program, *a)
    else:
        fn = getattr(lib, entry)
    k_max = int(steps_per_launch or MAX_STEPS_PER_LAUNCH)
    if mirror is not None:
        k_max = min(k_max, int(mirror.steps_per_launch(steps_per_launch)))
    cs = chains.struct()
    ms = moments.struct() if moments is not None else None
    stream = torch.cuda.current_stream(chains.device).cuda_stream
    done = 0
    with torch.cuda.device(chains.device):
        while done < n_steps:
            k = min(k_max, n_steps - done)
            run = _capi.Run()
            run.seed = seed
            run.step0 = step0 + done
            run.n_steps = k
            run.global_frequency = float(global_frequency)
            run.batch_size = int(batch_size or 1)
            run.lanes_per_chain = int(lanes_per_chain)
            run.debug_flags = int(debug_flags)
            run.math_mode = int(math_mode)
            if dump_draws is not None:
                if k != n_steps:
                    raise ValueError("dump_draws covers one launch: steps_per_launch must be >= n_steps")
                do = _capi.DrawsOut(dump_draws[0].data_ptr(), dump_draws[1].data_ptr(), dump_draws[2].data_ptr())
                run.dump_draws = C.pointer(do)
            if gf_per_chain is not None:
                run.global_frequency_per_chain = gf_per_chain.data_ptr()
            if history is not None:
                run.history = history[done].data_ptr()
                run.hist_stride = chains.n
            if ms is not None:
                run.moments = C.pointer(ms)
            _capi.check(fn(C.byref(model_desc), C.byref(local_desc), C.byref(global_desc), C.byref(cs), C.byref(run),
                           C.c_void_p(stream)), entry)
            done += k
            if mirror is not None:
                mirror.rows_done(step0 + done)                     # history row i = iteration i (row 0 = Initial_theta)
    if moments is not None:
        moments.steps += n_steps


def glmala_init(model_desc, chains):
    lib = _capi.lib()
    cs = chains.struct()
    stream = torch.cuda.current_stream(chains.device).cuda_stream
    with torch.cuda.device(chains.device):
        _capi.check(lib.glabc_glmala_init(C.byref(model_desc), C.byref(cs), C.c_void_p(stream)), "glabc_glmala_init")


def run_glmala_steps(model_desc, importance_desc, mala, chains, n_steps, step0, seed, global_frequency, batch_size,
                     history=None, moments=None, steps_per_launch=None, lanes_per_chain=0, mirror=None):
    """GLMALA twin of run_steps (entry point glabc_glmala_steps); `mala` is a _capi.Mala.
    lanes_per_chain: 0 = let the library choose; 1 = 64 chains per wavefront, 2 = 32 (the other 32 lanes only help with the
    wave-cooperative gradient) -- launch geometry, results are identical."""
    lib = _capi.lib()
    k_max = int(steps_per_launch or MAX_STEPS_PER_LAUNCH)
    if mirror is not None:
        k_max = min(k_max, int(mirror.steps_per_launch(steps_per_launch)))
    cs = chains.struct()
    ms = moments.struct() if moments is not None else None
    stream = torch.cuda.current_stream(chains.device).cuda_stream
    done = 0
    with torch.cuda.device(chains.device):
        while done < n_steps:
            k = min(k_max, n_steps - done)
            run = _capi.Run()
            run.seed = seed
            run.step0 = step0 + done
            run.n_steps = k
            run.global_frequency = float(global_frequency)
            run.batch_size = int(batch_size)
            run.lanes_per_chain = int(lanes_per_chain)
            if history is not None:
                run.history = history[done].data_ptr()
                run.hist_stride = chains.n
            if ms is not None:
                run.moments = C.pointer(ms)
            _capi.check(lib.glabc_glmala_steps(C.byref(model_desc), C.byref(importance_desc), C.byref(mala), C.byref(cs),
                                               C.byref(run), C.c_void_p(stream)), "glabc_glmala_steps")
            done += k
            if mirror is not None:
                mirror.rows_done(step0 + done)
    if moments is not None:
        moments.steps += n_steps


def init_weights(model_desc, importance_desc, chains):
    lib = _capi.lib()
    cs = chains.struct()
    stream = torch.cuda.current_stream(chains.device).cuda_stream
    with torch.cuda.device(chains.device):
        _capi.check(lib.glabc_init_weights(C.byref(model_desc), C.byref(importance_desc), C.byref(cs),
                                           C.c_void_p(stream)), "glabc_init_weights")


def model_descriptor(abc_set):
    """The Model object's glabc_model.  The fused kernels evaluate the Model callbacks
    in registers, so the object must describe itself (examples/Mixture.py does); an
    arbitrary Python callback cannot be fused and is refused loudly."""
    fn = getattr(abc_set, "descriptor", None)
    if fn is None:
        raise TypeError("%s has no descriptor(): the HIP samplers need a Model that exposes its simulator / "
                        "prior / kernel as a glabc_model (see glabcmcmc_amd.examples.Mixture.Mixture_set)"
                        % type(abc_set).__name__)
    return fn()
