"""CompiledModel -- a user's simulator inside the fused kernel (run-time compilation, ``glabc_rtc_compile``).

The reference's Model is a Python object whose ``generate_samples`` may be anything (examples/Mixture.py:13-26).  Python
cannot run inside a GPU kernel, so such a Model goes through the split-phase path (``generic.py``, ~10^8.9 chain-steps/s).
If the simulator can be written as a few lines of C, ``CompiledModel`` compiles the library's own sampler code around it
(hiprtc, a second or so per configuration, cached) and the samplers run it at the speed of the built-in Models
(~10^10.3 chain-steps/s):

    SIM = '''
    GLABC_SIMULATOR void glabc_user_simulate(const float* theta, const float* eps, float* y)
    {   /* theta[GLABC_THETA_DIM], eps[GLABC_NOISE_DIM] standard normals -> y[GLABC_Y_DIM] */
        for (int j = 0; j < GLABC_Y_DIM; ++j) y[j] = fabsf(theta[j]) + 0.2236068f * eps[j];
    }'''
    model = CompiledModel(theta_dim=2, y_dim=2, simulator_source=SIM, prior=DiagGaussian(2, zeros, zeros),
                          y_obs=[1.5, 1.5], epsilon=0.05)
    MCMCRunner(model).run_glmcmc(...)

The rest of the Model defaults to what the reference's example uses: a ``DiagGaussian`` / ``Uniform`` prior, the Euclidean
discrepancy to ``y_obs`` and the Gaussian ABC kernel of width ``epsilon`` (examples/Mixture.py:28-45) -- and each of them can be
user source too: the same string may define, announced by a ``#define`` each,

    #define GLABC_USER_PRIOR 1
    GLABC_SIMULATOR float glabc_user_prior_log_prob(const float* theta) { ... }               /* Mixture.py:28-31 */
    #define GLABC_USER_DISCREPANCY 1
    GLABC_SIMULATOR float glabc_user_discrepancy(const float* y, const float* y_obs) { ... }  /* Mixture.py:33-36 */
    #define GLABC_USER_KERNEL 1
    GLABC_SIMULATOR float glabc_user_log_kernel(float dis, float scale) { ... }               /* Mixture.py:38-45 */

which the fused kernel then calls in place of the descriptor's forms (``prior`` stays a required argument: with a user prior it
only says where the self-check starts its chains).  The object also
implements the full duck-typed protocol (``generate_samples`` through the compiled simulator on rows, ``prior_log_prob`` /
``discrepancy`` / ``calculate_log_kernel`` through the row-wise kernels), so it works with every sampler, the split-phase
path included.  Use + - * / fmaf sqrtf fabsf and the ``glabc_*`` functions of include/glabc_numerics.h (``glabc_expf``,
``glabc_logf``, ``glabc_sincos2pi`` ...) for results that a CPU build of the same source reproduces bit for bit.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _capi, distribution, engine
from .distribution import _fill, _launch_rowwise


class SimulatorCompileError(RuntimeError):
    pass


class SimulatorSelfCheckError(RuntimeError):
    pass


class CompiledModel:
    def __init__(self, theta_dim, y_dim, simulator_source, prior, y_obs, epsilon, noise_dim=None):
        self.theta_dim, self.y_dim = int(theta_dim), int(y_dim)
        self.noise_dim = int(y_dim if noise_dim is None else noise_dim)
        self.simulator_source = str(simulator_source)
        self.prior = prior
        self.y_obs = torch.as_tensor(y_obs, dtype=torch.float32).reshape(1, -1)
        if self.y_obs.shape[1] != self.y_dim:
            raise ValueError("y_obs has %d entries, y_dim is %d" % (self.y_obs.shape[1], self.y_dim))
        self.epsilon = epsilon
        self._programs = {}
        self._checked = set()
        self._failed = {}                                 # (algorithm, batch size) -> message of the failed self-check
        self.user_prior, self.user_discrepancy, self.user_kernel = (self._announces(m) for m in
                                                                    ("GLABC_USER_PRIOR", "GLABC_USER_DISCREPANCY", "GLABC_USER_KERNEL"))

    def _announces(self, macro):
        import re
        return re.search(r"define[ \t]+%s(?![A-Za-z0-9_])" % macro, self.simulator_source) is not None

    # ---- run-time compiled programs, one per (algorithm, batch size) -------------------------------------------------
    def program(self, algo, batch_size=1):
        key = (int(algo), 1 if algo == _capi.ALGO_GLOBALMCMC else int(batch_size))
        if key in self._failed:
            raise SimulatorSelfCheckError(self._failed[key])
        if key not in self._programs:
            handle = C.c_void_p()
            log = C.create_string_buffer(1 << 16)
            rc = _capi.lib().glabc_rtc_compile(self.simulator_source.encode(), key[0], self.theta_dim, self.y_dim, self.noise_dim,
                                               key[1], C.byref(handle), log, len(log))
            if rc != _capi.OK:
                raise SimulatorCompileError("glabc_rtc_compile failed (status %d):\n%s" % (rc, log.value.decode(errors="replace")))
            self._programs[key] = handle
        if key not in self._checked and os.environ.get("GLABC_RTC_SELF_CHECK", "1") != "0":
            self._checked.add(key)                       # before the check: self_check re-enters program() through the samplers
            try:
                self.self_check(*key)
            except BaseException as exc:
                # a program that failed (or did not finish) its check must never be handed out: release it, and keep the
                # verdict so that every later call raises again instead of sampling with a miscompiled kernel
                self._checked.discard(key)
                handle = self._programs.pop(key, None)
                if handle is not None:
                    _capi.lib().glabc_rtc_release(handle)
                if isinstance(exc, SimulatorSelfCheckError):
                    self._failed[key] = str(exc)
                raise
        return self._programs[key]

    def self_check(self, algo, batch_size=1, n_chains=512, steps=4, seed=20240229):
        """The freshly compiled kernel against the split-phase path on the same Philox streams: a few iterations of
        `n_chains` synthetic chains through both (same simulator binary for the rows, library kernels for everything
        else -- the path tests/test_generic_path.py holds to the CPU checker), compared bit for bit.  Runs once per program
        (a few ms); GLABC_RTC_SELF_CHECK=0 skips it.  It exists because the run-time compiler has miscompiled this very
        kernel before: a mismatch raises instead of returning samples from the wrong law."""
        from .GlobalMCMC import GlobalMCMC
        from .GLMCMC import GLMCMC
        dev = engine.require_device(None)
        g = torch.Generator().manual_seed(seed)
        d = self.theta_dim
        pd = self.prior.descriptor()
        loc = torch.tensor([pd.p0[j] for j in range(d)])
        scale = torch.tensor([pd.p2[j] for j in range(d)])
        if pd.kind == _capi.DIST_UNIFORM:                             # p0 = low, p2 = high - low
            theta0 = loc + scale * torch.rand(n_chains, d, generator=g)
            imp = distribution.Uniform(d, loc, loc + scale)
            spread = scale / 4.0
        else:
            theta0 = loc + scale * torch.randn(n_chains, d, generator=g)
            imp = distribution.DiagGaussian(d, loc.clone(), torch.log(scale))
            spread = scale
        local = distribution.DiagGaussian(d, torch.zeros(d), torch.log(0.3 * spread))
        y0 = self.simulate_from_noise(theta0, torch.randn(n_chains, self.noise_dim, generator=g)).cpu()
        out = []
        for path, kw in (("fused", {}), ("generic", dict(sentinel_redraw=False, graph=False))):
            if algo == _capi.ALGO_GLMCMC:
                r = GLMCMC(self, steps + 1, theta0, y0, local, None, 0.5, imp, batch_size, seed=seed, device=dev, verbose=False,
                           path=path, **kw)
            else:
                r = GlobalMCMC(self, steps + 1, theta0, y0, imp, None, 0.5, local, seed=seed, device=dev, verbose=False, path=path,
                               **kw)
            out.append(r[1].contiguous().view(torch.int32))
        if not torch.equal(out[0], out[1]):
            bad = int((out[0] != out[1]).any(dim=-1).any(dim=0).sum()) if out[0].dim() == 3 else -1
            raise SimulatorSelfCheckError(
                "the run-time compiled kernel (algorithm %d, batch size %d) disagrees with the split-phase path on %d of %d "
                "chains after %d iterations: refusing to sample with it (use path='generic', and please report the "
                "simulator source)" % (algo, batch_size, bad, n_chains, steps))
        return True

    def __del__(self):
        try:
            for h in self._programs.values():
                _capi.lib().glabc_rtc_release(h)
        except Exception:                       # interpreter shutdown
            pass

    # ---- the duck-typed Model protocol (examples/Mixture.py:5-53) ----------------------------------------------------
    def simulate_from_noise(self, theta, eps):
        """generate_samples(theta, 1) with the simulator's standard normals supplied (rows on the device)"""
        dev = engine.require_device(theta.device if theta.is_cuda else None)
        th = theta.detach().to(device=dev, dtype=torch.float32).reshape(-1, self.theta_dim).contiguous()
        ee = eps.detach().to(device=dev, dtype=torch.float32).reshape(th.shape[0], self.noise_dim).contiguous()
        y = torch.empty(th.shape[0], self.y_dim, dtype=torch.float32, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        with torch.cuda.device(dev):
            _capi.check(_capi.lib().glabc_rtc_simulate(self.program(_capi.ALGO_GLOBALMCMC), th.data_ptr(), ee.data_ptr(),
                                                       th.shape[0], y.data_ptr(), C.c_void_p(stream)), "glabc_rtc_simulate")
        return y if theta.is_cuda else y.cpu()

    def generate_samples(self, theta, num_samples=1):
        theta = theta.reshape(-1, self.theta_dim)
        if num_samples != 1:
            theta = theta.repeat_interleave(num_samples, dim=0) if theta.shape[0] > 1 else theta.expand(num_samples, -1)
        eps = torch.randn(theta.shape[0], self.noise_dim, device=theta.device)
        return self.simulate_from_noise(theta, eps)

    def _dev(self, t):
        return t if t.is_cuda else t.to(engine.require_device(None))

    def _rows(self, what, rows, epsilon=None):
        """glabc_rtc_model_rows: a callback the source replaces, on rows, through the functions the fused kernel calls"""
        x = self._dev(rows).detach().to(torch.float32).contiguous()
        out = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
        m = self.descriptor(epsilon)
        stream = torch.cuda.current_stream(x.device).cuda_stream
        with torch.cuda.device(x.device):
            _capi.check(_capi.lib().glabc_rtc_model_rows(self.program(_capi.ALGO_GLOBALMCMC), C.byref(m), what, x.data_ptr(),
                                                         x.shape[0], out.data_ptr(), C.c_void_p(stream)), "glabc_rtc_model_rows")
        return out

    def prior_log_prob(self, samples):
        s = samples.reshape(-1, self.theta_dim)
        if self.user_prior:
            out = self._rows(_capi.RTC_PRIOR_LOG_PROB, s)
        else:
            out = _launch_rowwise("glabc_model_prior_log_prob", self.descriptor(), self._dev(s), "prior_log_prob")
        return out if samples.is_cuda else out.cpu()

    def discrepancy(self, y):
        yy = y.reshape(-1, self.y_dim)
        if self.user_discrepancy:
            out = self._rows(_capi.RTC_DISCREPANCY, yy)
        else:
            out = _launch_rowwise("glabc_model_discrepancy", self.descriptor(), self._dev(yy), "discrepancy")
        return out if y.is_cuda else out.cpu()

    def calculate_log_kernel(self, y, epsilon=None):
        yy = y.reshape(-1, self.y_dim)
        if self.user_discrepancy or self.user_kernel:
            out = self._rows(_capi.RTC_LOG_KERNEL, yy, epsilon)
        else:
            out = _launch_rowwise("glabc_model_log_kernel", self.descriptor(epsilon), self._dev(yy), "calculate_log_kernel")
        return out if y.is_cuda else out.cpu()

    def descriptor(self, epsilon=None):
        if epsilon is None:
            epsilon = self.epsilon
        m = _capi.Model()
        m.sim_kind = _capi.SIM_USER
        m.theta_dim, m.y_dim = self.theta_dim, self.y_dim
        m.prior = self.prior.descriptor()
        m.noise = distribution.DiagGaussian(self.noise_dim, torch.zeros(self.noise_dim), torch.zeros(self.noise_dim)).descriptor()
        _fill(m.y_obs, self.y_obs.reshape(-1))
        kern = distribution.DiagGaussian(1, loc=torch.tensor([0.0]), log_scale=torch.log(torch.tensor([epsilon]))).descriptor()
        m.kern_log_scale, m.kern_scale, m.kern_c0 = kern.p1[0], kern.p2[0], kern.c0
        m.epsilon = float(np.float32(epsilon))
        return m
