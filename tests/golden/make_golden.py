"""Generate the golden fixtures of tests/golden/ from the reference itself.

Runs ONLY in the build container (it imports /root/reference); the GPU box and
the test-suite see only the .npz files this script writes.  Usage:

    python tests/golden/make_golden.py            # all fixtures (~5 min)
    python tests/golden/make_golden.py primitives # one group

How the reference is driven
  * its modules are imported as they lie in /root/reference (an empty package
    object named `glabcmcmc` with __path__ pointing there lets `import
    glabcmcmc.GLMCMC` etc. work without executing the package's __init__, which
    would pull in the third-party `normflows` that is not installed here);
  * `torch.rand`, `torch.randn` and `np.random.uniform` are replaced, for the
    duration of one run, by functions that hand out pre-computed numbers in the
    order the loops ask for them (a "tape"); the loops themselves
    (GLMCMC.py:58-104, GlobalMCMC.py:37-68) run unmodified, one chain at a time;
  * "philox" fixtures compute the tape from the Philox stream specified in
    include/glabc_numerics.h (through the oracle library's oracle_step_draws),
    so only seeds and the resulting chains are stored and BOTH the CPU oracle
    and the gfx950 kernels can be compared with the reference bit for bit;
    "tape" fixtures use NumPy-generated numbers and store the tape as well.
"""
import contextlib
import io
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "gl-abc-mcmc_amd")]
sys.dont_write_bytecode = True

REF = "/root/reference"
pkg = types.ModuleType("glabcmcmc")
pkg.__path__ = [os.path.join(REF, "glabcmcmc")]
sys.modules["glabcmcmc"] = pkg
sys.path.insert(0, os.path.join(REF, "glabcmcmc", "examples"))

import glabcmcmc.distribution as rdist          # noqa: E402
import glabcmcmc.GLMCMC as rglmcmc              # noqa: E402
import glabcmcmc.GlobalMCMC as rglobal          # noqa: E402
import glabcmcmc.GLMALA as rglmala              # noqa: E402
import glabcmcmc.ESJD as resjd                  # noqa: E402
from Mixture import Mixture_set                 # noqa: E402
from glabcmcmc_amd.examples.GK import GK_set     # noqa: E402  (the build's own g-and-k Model: BASELINE config 4 has
                                                 #  no model in the reference tree; it is DRIVEN by the reference's loops)

import oracle_lib                               # noqa: E402

torch.set_num_threads(1)


# --------------------------------------------------------------------------- tape
class Tape:
    """Numbers for one chain: u[T,2] (branch, accept), r[T] f64, z[T,P,d+yd]."""

    def __init__(self, u, r, z, d, gf, isir, grad_fn=None):
        self.u, self.r, self.z, self.d, self.gf, self.isir = u, r, z, d, gf, isir
        self.t = -1
        self.expect = "branch"
        self.k = 0
        # GLMALA: numberical_gradient_logABC reseeds torch before each block of simulations
        # (GLMALA.py:76,80); the patched manual_seed marks the next randn as gradient noise of
        # (g, coordinate): g = 0 gradient at Theta_old (first local move only), g = 1 at the proposal
        self.grad_fn = grad_fn
        self.grad_done = False
        self.ms_count = 0
        self.ms_init = False
        self.pending = None

    def manual_seed(self, seed):
        if self.ms_count == 0:
            self.ms_init = not self.grad_done
        idx = self.ms_count // 2
        self.ms_count += 1
        if self.ms_init:
            g, k = (0, idx) if idx < self.d else (1, idx - self.d)
        else:
            g, k = 1, idx
        self.pending = (g, k)
        self.grad_done = True

    def rand(self, *size, **kw):
        if len(size) == 1 and isinstance(size[0], (tuple, list)):
            size = tuple(size[0])
        if len(size) == 2:                          # Uniform.forward: rand((n, d))
            return self._noise(size)
        assert size == (1,), size
        if self.expect == "branch":
            self.t += 1
            self.k = 0
            self.ms_count = 0
            v = self.u[self.t, 0]
            if not (self.isir and v < self.gf):
                self.expect = "accept"
        else:
            v = self.u[self.t, 1]
            self.expect = "branch"
        return torch.tensor([v], dtype=torch.float32)

    def _noise(self, shape):
        if self.pending is not None:
            g, k = self.pending
            self.pending = None
            out = self.grad_fn(self.t, g, k)
            assert tuple(shape) == out.shape, (shape, out.shape)
            return torch.from_numpy(out)
        n, dd = shape
        lo = 0 if self.k == 0 else self.d
        out = self.z[self.t, :n, lo:lo + dd]
        self.k += 1
        return torch.from_numpy(np.ascontiguousarray(out))

    def randn(self, *size, **kw):
        if len(size) == 1 and isinstance(size[0], (tuple, list)):
            size = tuple(size[0])
        return self._noise(size)

    def uniform(self, lo=0, hi=1):
        return float(self.r[self.t])


def _ieee_sqrt(t):
    """correctly rounded square root (NumPy's) in place of torch.sqrt: this torch build's CPU sqrt goes
    through MKL VML 'high accuracy' mode, which is 1 ulp low for 0.65 % of float32 inputs"""
    return torch.from_numpy(np.sqrt(t.detach().numpy()))


@contextlib.contextmanager
def patched(tape, ieee_sqrt=False):
    import secrets
    saved = (torch.rand, torch.randn, np.random.uniform, torch.manual_seed, np.random.seed, secrets.randbelow,
             torch.sqrt)
    torch.rand, torch.randn, np.random.uniform = tape.rand, tape.randn, tape.uniform
    torch.manual_seed, np.random.seed, secrets.randbelow = tape.manual_seed, (lambda s: None), (lambda n: 12345)
    if ieee_sqrt:
        torch.sqrt = _ieee_sqrt
    try:
        with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
            yield
    finally:
        (torch.rand, torch.randn, np.random.uniform, torch.manual_seed, np.random.seed, secrets.randbelow,
         torch.sqrt) = saved


def make_dist(spec):
    kind = spec[0]
    if kind == "gauss":
        _, loc, scale = spec
        return rdist.DiagGaussian(len(loc), torch.tensor(loc, dtype=torch.float32),
                                  torch.log(torch.tensor(scale, dtype=torch.float32)))
    if kind == "uniform":
        _, low, high = spec
        return rdist.Uniform(len(low), torch.tensor(low, dtype=torch.float32), torch.tensor(high, dtype=torch.float32))
    raise ValueError(kind)


def philox_tape(L, seed, chain, T, P, d, yd, uniform_prop_global, uniform_prop_local, gf):
    """Draws of steps 1..T of one chain from the specified Philox stream."""
    u = np.zeros((T, 2), np.float32)
    r = np.zeros(T, np.float64)
    z = np.zeros((T, P, d + yd), np.float32)
    u2 = np.zeros(2, np.float32)
    rr = np.zeros(1, np.float64)
    zz = np.zeros((P, d + yd), np.float32)
    w = np.zeros(4, np.uint32)
    for t in range(T):
        L.oracle_step_draws(seed, chain, t + 1, P, d, yd, u2.ctypes.data, rr.ctypes.data, zz.ctypes.data)
        u[t], r[t], z[t] = u2, rr[0], zz
        is_global = u2[0] < np.float32(gf)
        if (uniform_prop_global and is_global) or (uniform_prop_local and not is_global):
            # a Uniform proposal takes the raw words of its slots as [0,1) uniforms
            spp = (d + (d & 1) + yd + 3) // 4
            key = np.array([seed & 0xFFFFFFFF, seed >> 32], np.uint32)
            for j in range(P):
                words = []
                for b in range(spp):
                    ctr = np.array([chain & 0xFFFFFFFF, chain >> 32, t + 1, 1 + j * spp + b], np.uint32)
                    L.oracle_philox4x32_10(ctr.ctypes.data, key.ctypes.data, w.ctypes.data)
                    words += list(w)
                z[t, j, :d] = (np.array(words[:d], np.uint32) >> 8).astype(np.float32) * np.float32(2.0 ** -24)
    return u, r, z


def numpy_tape(rng, T, P, d, yd, uniform_prop):
    u = (rng.integers(0, 1 << 24, (T, 2)).astype(np.float32) * np.float32(2.0 ** -24))
    r = rng.random(T)
    z = rng.standard_normal((T, P, d + yd)).astype(np.float32)
    if uniform_prop:
        z[:, :, :d] = rng.integers(0, 1 << 24, (T, P, d)).astype(np.float32) * np.float32(2.0 ** -24)
    return u, r, z


def reference_constants(cfg):
    """The float32 constants the reference computes on the host for this configuration, with THIS
    machine's torch (log / exp differ in the last bits between CPU types): stored in the fixture so the
    tests rebuild exactly the descriptors the golden chains were produced with."""
    out = {}
    out["c_noise_log_scale"] = torch.log(torch.tensor([0.05, 0.05]).sqrt()).numpy()                    # Mixture.py:19
    out["c_noise_scale"] = torch.exp(torch.log(torch.tensor([0.05, 0.05]).sqrt())).numpy()
    out["c_kern_log_scale"] = torch.log(torch.tensor([cfg["epsilon"]])).numpy()                        # Mixture.py:42-43
    out["c_kern_scale"] = torch.exp(torch.log(torch.tensor([cfg["epsilon"]]))).numpy()
    for tag in ("local", "global"):
        spec = cfg[tag]
        if spec[0] == "gauss":
            ls = torch.log(torch.tensor(spec[2], dtype=torch.float32))
            out["c_%s_p1" % tag] = ls.numpy()
            out["c_%s_p2" % tag] = torch.exp(ls).numpy()
    return out


def make_model(cfg):
    return GK_set(cfg["epsilon"]) if cfg.get("model") == "gk" else Mixture_set(cfg["epsilon"])


def run_reference(algo, cfg, theta0, y0, tape):
    model = make_model(cfg)
    local = make_dist(cfg["local"])
    glob = make_dist(cfg["global"])
    T = cfg["T"]
    th0 = torch.from_numpy(theta0.copy())
    yy0 = torch.from_numpy(y0.copy()).view(1, -1)
    with patched(tape, cfg.get("ieee_sqrt", False)):
        if algo == "glmcmc":
            out = rglmcmc.GLMCMC(model, T + 1, th0, yy0, local, None, cfg["gf"], glob, cfg["N"])
        elif algo == "glmala":
            out = rglmala.GLMALA(model, T + 1, th0, yy0, cfg["tau"], cfg["num_grad"], None, cfg["gf"], glob, cfg["N"])
        else:
            out = rglobal.GlobalMCMC(model, T + 1, th0, yy0, glob, None, cfg["gf"], local)
    assert tape.t == T - 1, (tape.t, T)
    return out.numpy().copy()


def sampler_fixture(name, algo, cfg, mode):
    L = oracle_lib.load()
    gk = cfg.get("model") == "gk"
    d, yd = (4, 8) if gk else (2, 2)
    C, T, N = cfg["C"], cfg["T"], cfg["N"]
    P = N if algo in ("glmcmc", "glmala") else 1
    rng = np.random.default_rng(cfg["seed"] + 1000)
    if gk:
        theta0 = rng.uniform(0.5, 5.0, (C, d)).astype(np.float32)
        zz = torch.from_numpy(rng.standard_normal((C, yd)).astype(np.float32))
        saved = torch.randn
        torch.randn = lambda *a, **k: zz
        try:
            y0 = GK_set(cfg["epsilon"]).generate_samples(torch.from_numpy(theta0)).numpy().copy()
        finally:
            torch.randn = saved
    else:
        theta0 = (rng.standard_normal((C, d)) * cfg.get("theta0_sd", 0.0)).astype(np.float32)
        y0 = (np.abs(theta0) + np.sqrt(np.float32(0.05)) * rng.standard_normal((C, yd)).astype(np.float32)).astype(np.float32)
    ug = cfg["global"][0] == "uniform"
    ul = cfg["local"][0] == "uniform"
    chains = np.zeros((T + 1, C, d), np.float32)
    tapes = []
    for c in range(C):
        if mode == "philox":
            u, r, z = philox_tape(L, cfg["seed"], cfg.get("chain0", 0) + c, T, P, d, yd, ug, ul, cfg["gf"])
        else:
            assert ug == ul
            u, r, z = numpy_tape(rng, T, P, d, yd, ug)
            tapes.append((u, r, z))
        grad_fn = None
        if algo == "glmala":
            num, chain_id = cfg["num_grad"], cfg.get("chain0", 0) + c

            def grad_fn(t, g, k, num=num, chain_id=chain_id):
                out = np.zeros((num, yd), np.float32)
                L.oracle_grad_noise(cfg["seed"], chain_id, t + 1, g, k, num, yd, out.ctypes.data)
                return out
        tape = Tape(u, r, z, d, np.float32(cfg["gf"]), algo in ("glmcmc", "glmala"), grad_fn)
        chains[:, c, :] = run_reference(algo, cfg, theta0[c], y0[c], tape)
        print("\r%s chain %d/%d" % (name, c + 1, C), end="", flush=True)
    print()
    out = dict(algo=algo, mode=mode, theta0=theta0, y0=y0, chains=chains,
               cfg=np.array(repr(cfg)), **reference_constants(cfg))
    if mode == "tape":
        out["tape_u"] = np.stack([t[0] for t in tapes])
        out["tape_r"] = np.stack([t[1] for t in tapes])
        out["tape_z"] = np.stack([t[2] for t in tapes])
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    moves = (np.abs(np.diff(chains, axis=0)).sum(-1) > 0).sum(0)
    print("   moves/chain: mean %.1f min %d max %d" % (moves.mean(), moves.min(), moves.max()))


G2 = lambda s: ("gauss", [0.0, 0.0], [s, s])          # noqa: E731

SAMPLER_FIXTURES = {
    # BASELINE config 2's algorithm and parameters (examples/Mixture.py:67-73)
    "glmcmc_philox_bench": ("glmcmc", dict(epsilon=0.05, gf=0.9, N=5, C=96, T=1500, seed=20261003,
                                           local=G2(0.35), **{"global": G2(1.0)}), "philox"),
    # wider kernel -> many moves; N=8 exercises the >=8-element torch.sum order
    "glmcmc_philox_n8": ("glmcmc", dict(epsilon=0.3, gf=0.5, N=8, C=24, T=600, seed=77, chain0=5000000000,
                                        theta0_sd=1.0, local=G2(0.35), **{"global": G2(1.0)}), "philox"),
    "glmcmc_philox_n1": ("glmcmc", dict(epsilon=0.3, gf=0.6, N=1, C=16, T=400, seed=3,
                                        theta0_sd=1.0, local=G2(0.5), **{"global": G2(1.2)}), "philox"),
    "glmcmc_philox_n3": ("glmcmc", dict(epsilon=0.2, gf=0.8, N=3, C=16, T=400, seed=4,
                                        theta0_sd=1.0, local=G2(0.35), **{"global": ("gauss", [0.5, -0.25], [1.5, 0.75])}), "philox"),
    "glmcmc_philox_n16": ("glmcmc", dict(epsilon=0.3, gf=0.9, N=16, C=12, T=300, seed=5,
                                         theta0_sd=1.0, local=G2(0.35), **{"global": G2(1.0)}), "philox"),
    # batches beyond the register kernels: 8 / 16 lanes of a wavefront share a chain (glabc_wide.hip); torch.sum's 8-wide path
    "glmcmc_philox_n32": ("glmcmc", dict(epsilon=0.3, gf=0.8, N=32, C=8, T=250, seed=7, theta0_sd=1.0,
                                         local=G2(0.35), **{"global": G2(1.0)}), "philox"),
    "glmcmc_philox_n100": ("glmcmc", dict(epsilon=0.1, gf=0.7, N=100, C=6, T=150, seed=8, theta0_sd=1.0, chain0=123456789012,
                                          local=G2(0.35), **{"global": ("gauss", [0.3, -0.2], [1.4, 1.1])}), "philox"),
    "glmcmc_philox_uniform": ("glmcmc", dict(epsilon=0.3, gf=0.7, N=4, C=16, T=500, seed=6, theta0_sd=1.0,
                                             local=("uniform", [-0.5, -0.5], [0.5, 0.5]),
                                             **{"global": ("uniform", [-3.0, -3.0], [3.0, 3.0])}), "philox"),
    # BASELINE config 1's algorithm (GlobalMCMC, gf 0.5)
    "globalmcmc_philox_bench": ("globalmcmc", dict(epsilon=0.05, gf=0.5, N=1, C=48, T=2000, seed=11,
                                                   local=G2(0.35), **{"global": G2(1.0)}), "philox"),
    "globalmcmc_philox_wide": ("globalmcmc", dict(epsilon=0.3, gf=0.5, N=1, C=16, T=600, seed=12, theta0_sd=1.0,
                                                  local=G2(0.35), **{"global": G2(1.0)}), "philox"),
    # BASELINE config 3's algorithm and parameters (README.md:128: gf 0.8, N 5, tau 0.3, num_grad 100)
    "glmala_philox_bench": ("glmala", dict(epsilon=0.05, gf=0.8, N=5, tau=0.3, num_grad=100, C=24, T=500, seed=31,
                                           local=G2(0.35), **{"global": G2(1.0)}), "philox"),
    # mostly local moves, wide kernel: many accepted MALA moves -> float64 state, stale weights
    "glmala_philox_local": ("glmala", dict(epsilon=0.3, gf=0.3, N=3, tau=0.25, num_grad=10, C=16, T=300, seed=32,
                                           theta0_sd=1.0, local=G2(0.35), **{"global": G2(1.0)}), "philox"),
    "glmala_philox_alllocal": ("glmala", dict(epsilon=0.3, gf=0.0, N=2, tau=0.2, num_grad=7, C=8, T=200, seed=33,
                                              theta0_sd=1.0, local=G2(0.35), **{"global": G2(1.0)}), "philox"),
    "glmala_philox_allglobal": ("glmala", dict(epsilon=0.3, gf=1.0, N=4, tau=0.3, num_grad=5, C=8, T=200, seed=34,
                                               theta0_sd=1.0, local=G2(0.35), **{"global": ("gauss", [0.2, -0.1], [1.2, 0.9])}), "philox"),
    # the same two configurations with torch.sqrt replaced by a correctly rounded sqrt for the run: GLMALA's
    # chains depend on the last bit of every simulated discrepancy (the float32 finite-difference prior
    # gradient, GLMALA.py:84-85, turns a 1-ulp change of theta into a 1e-2 change of the drift), and this
    # torch build's sqrt (MKL VML) is not correctly rounded -- see DESIGN.md
    "glmala_philox_bench_ieee": ("glmala", dict(epsilon=0.05, gf=0.8, N=5, tau=0.3, num_grad=100, C=24, T=500, seed=31,
                                                ieee_sqrt=True, local=G2(0.35), **{"global": G2(1.0)}), "philox"),
    "glmala_philox_local_ieee": ("glmala", dict(epsilon=0.3, gf=0.3, N=3, tau=0.25, num_grad=10, C=16, T=300, seed=32,
                                                ieee_sqrt=True, theta0_sd=1.0, local=G2(0.35), **{"global": G2(1.0)}), "philox"),
    "glmala_philox_uniform_ieee": ("glmala", dict(epsilon=0.3, gf=0.5, N=4, tau=0.3, num_grad=12, C=8, T=300, seed=35,
                                                  ieee_sqrt=True, theta0_sd=1.0, local=G2(0.35),
                                                  **{"global": ("uniform", [-3.0, -3.0], [3.0, 3.0])}), "philox"),
    "glmala_philox_uniform": ("glmala", dict(epsilon=0.3, gf=0.5, N=4, tau=0.3, num_grad=12, C=8, T=300, seed=35,
                                             theta0_sd=1.0, local=G2(0.35), **{"global": ("uniform", [-3.0, -3.0], [3.0, 3.0])}), "philox"),
    # BASELINE config 4's model (g-and-k, theta_dim 4, the build's own Model class) driven by the reference's loops
    "glmcmc_philox_gk": ("glmcmc", dict(model="gk", epsilon=1.0, gf=0.8, N=5, C=16, T=600, seed=51,
                                        local=("gauss", [0.0] * 4, [0.15] * 4),
                                        **{"global": ("uniform", [0.0] * 4, [10.0] * 4)}), "philox"),
    "glmcmc_philox_gk_gauss": ("glmcmc", dict(model="gk", epsilon=0.7, gf=0.5, N=3, C=12, T=500, seed=52,
                                              local=("gauss", [0.0] * 4, [0.2, 0.1, 0.2, 0.1]),
                                              **{"global": ("gauss", [3.0, 1.5, 2.0, 1.0], [2.0, 1.0, 1.5, 0.7])}), "philox"),
    "globalmcmc_philox_gk": ("globalmcmc", dict(model="gk", epsilon=1.0, gf=0.5, N=1, C=12, T=800, seed=53,
                                                local=("gauss", [0.0] * 4, [0.15] * 4),
                                                **{"global": ("uniform", [0.0] * 4, [10.0] * 4)}), "philox"),
    # stored-tape fixtures (NumPy numbers; independent of the Philox specification)
    "glmcmc_tape_small": ("glmcmc", dict(epsilon=0.2, gf=0.8, N=5, C=8, T=300, seed=21, theta0_sd=1.0,
                                         local=G2(0.35), **{"global": G2(1.0)}), "tape"),
    "globalmcmc_tape_small": ("globalmcmc", dict(epsilon=0.2, gf=0.5, N=1, C=8, T=300, seed=22, theta0_sd=1.0,
                                                 local=G2(0.35), **{"global": G2(1.0)}), "tape"),
}


# --------------------------------------------------------------------------- primitives
def primitives():
    rng = np.random.default_rng(5)
    out = {}
    # DiagGaussian.log_prob / forward (distribution.py:166-181)
    for tag, loc, scale in (("std", [0.0, 0.0], [1.0, 1.0]), ("lp", [0.0, 0.0], [0.35, 0.35]),
                            ("gen", [0.5, -0.25], [1.5, 0.75]), ("d1", [0.0], [0.05]),
                            ("d4", [0.1, -0.2, 0.3, 0.0], [0.5, 1.0, 2.0, 0.1]),
                            ("d7", [0.0] * 7, [1.0, 0.5, 0.25, 2.0, 1.5, 0.75, 0.3]),
                            ("d8", [0.1] * 8, [1.0, 0.5, 0.25, 2.0, 1.5, 0.75, 0.3, 0.9])):
        g = make_dist(("gauss", loc, scale))
        d = len(loc)
        z = (rng.standard_normal((257, d)) * 2).astype(np.float32)
        eps = rng.standard_normal((257, d)).astype(np.float32)
        out["dg_%s_loc" % tag] = np.array(loc, np.float32)
        out["dg_%s_log_scale" % tag] = g.log_scale.numpy()
        out["dg_%s_scale" % tag] = torch.exp(g.log_scale).numpy()
        out["dg_%s_z" % tag] = z
        out["dg_%s_log_prob" % tag] = g.log_prob(torch.from_numpy(z)).numpy()
        saved = torch.randn
        torch.randn = lambda *a, **k: torch.from_numpy(eps)
        try:
            zz, lp = g.forward(257)
        finally:
            torch.randn = saved
        out["dg_%s_eps" % tag] = eps
        out["dg_%s_fwd_z" % tag] = zz.numpy()
        out["dg_%s_fwd_log_p" % tag] = lp.numpy()
    # Uniform (distribution.py:50-86)
    for tag, low, high in (("box", [-3.0, -3.0], [3.0, 3.0]), ("inc", [-0.5, -0.5], [0.5, 0.5]),
                           ("d4", [0.0] * 4, [10.0] * 4)):
        g = make_dist(("uniform", low, high))
        d = len(low)
        z = (rng.uniform(-1.2, 1.2, (200, d)) * np.abs(np.array(high))).astype(np.float32)
        z[0] = np.array(low, np.float32)             # closed interval: the bounds are inside
        z[1] = np.array(high, np.float32)
        u = rng.integers(0, 1 << 24, (200, d)).astype(np.float32) * np.float32(2.0 ** -24)
        out["un_%s_low" % tag] = np.array(low, np.float32)
        out["un_%s_high" % tag] = np.array(high, np.float32)
        out["un_%s_log_prob_val" % tag] = np.float32(g.log_prob_val.item())
        out["un_%s_z" % tag] = z
        out["un_%s_log_prob" % tag] = g.log_prob(torch.from_numpy(z)).numpy()
        saved = torch.rand
        torch.rand = lambda *a, **k: torch.from_numpy(u)
        try:
            zz, lp = g.forward(200)
        finally:
            torch.rand = saved
        out["un_%s_u" % tag] = u
        out["un_%s_fwd_z" % tag] = zz.numpy()
        out["un_%s_fwd_log_p" % tag] = lp.numpy()
    out["un_default_log_prob_val"] = np.float32(rdist.Uniform(2).log_prob_val.item())
    # Mixture_set callbacks (examples/Mixture.py:13-45)
    for eps_k in (0.05, 0.3):
        m = Mixture_set(eps_k)
        tag = "mix_%g" % eps_k
        theta = (rng.standard_normal((300, 2)) * 1.5).astype(np.float32)
        noise = rng.standard_normal((300, 2)).astype(np.float32)
        saved = torch.randn
        torch.randn = lambda *a, **k: torch.from_numpy(noise)
        try:
            y = m.generate_samples(torch.from_numpy(theta), 1).numpy()
        finally:
            torch.randn = saved
        out[tag + "_theta"] = theta
        out[tag + "_noise"] = noise
        out[tag + "_y"] = y
        out[tag + "_prior"] = m.prior_log_prob(torch.from_numpy(theta)).numpy()
        out[tag + "_dis"] = m.discrepancy(torch.from_numpy(y)).numpy()
        out[tag + "_logk"] = m.calculate_log_kernel(torch.from_numpy(y)).numpy()
        out[tag + "_noise_scale"] = torch.exp(torch.log(torch.tensor([0.05, 0.05]).sqrt())).numpy()
        out[tag + "_noise_log_scale"] = torch.log(torch.tensor([0.05, 0.05]).sqrt()).numpy()
        out[tag + "_kern_log_scale"] = torch.log(torch.tensor([eps_k])).numpy()
        out[tag + "_kern_scale"] = torch.exp(torch.log(torch.tensor([eps_k]))).numpy()
    # esjd (ESJD.py:2-25)
    chains = []
    vals = []
    for T, d in ((5, 2), (300, 2), (2000, 2), (300, 4), (50, 1), (400, 3)):
        x = np.cumsum((rng.random((T, d)) < 0.2) * rng.standard_normal((T, d)), axis=0).astype(np.float32)
        chains.append(x)
        vals.append(float(resjd.esjd(torch.from_numpy(x))))
    out["esjd_known"] = np.float32(resjd.esjd(torch.tensor([[0, 0], [1, 0], [1, 2], [1, 2], [0, 1.0]])))
    for i, (x, v) in enumerate(zip(chains, vals)):
        out["esjd_chain_%d" % i] = x
        out["esjd_value_%d" % i] = np.float32(v)
    # torch.sum association (GLMCMC.py:82): rows of n = 2..17 float32 weights
    for n in range(1, 18):
        x = (rng.standard_normal((64, n)) * rng.choice([1e-3, 1.0, 1e3], (64, n))).astype(np.float32)
        out["rowsum_%d_x" % n] = x
        out["rowsum_%d_sum" % n] = np.array([torch.sum(torch.from_numpy(r)).item() for r in x], np.float32)
    # weight_sampling edge cases (GLMCMC.py:7-22)
    ws = []
    for w, ran in (([0.2, 0.3, 0.5], 0.0), ([0.2, 0.3, 0.5], 0.19999), ([0.2, 0.3, 0.5], 0.2), ([0.2, 0.3, 0.5], 0.999999),
                   ([0.0, 0.0, 1.0], 0.5), ([0.5, 0.25], 0.9), ([float("nan"), 0.5], 0.1), ([1.0], 0.3)):
        saved = np.random.uniform
        np.random.uniform = lambda a, b, ran=ran: ran
        try:
            ind = rglmcmc.weight_sampling(list(map(float, w)))
        finally:
            np.random.uniform = saved
        ws.append((w, ran, -1 if ind is None else ind))
    out["weight_sampling_cases"] = np.array(repr(ws))
    # ---- everything below was added after round 1: appended so that the arrays above keep their random stream ----
    # Gamma.log_prob (distribution.py:123-137): float64 through scipy.stats.gamma.pdf; negative coordinates, an exact zero,
    # the bulk, and tails far enough out for the pdf to underflow (-inf although a logpdf would be finite)
    for tag, shape, rate in (("a", [2.0, 3.0], [1.0, 2.0]), ("b", [0.5], [3.0]), ("c", [1.0, 7.5, 2.25], [0.5, 1.5, 4.0])):
        g = rdist.Gamma(torch.tensor(shape), torch.tensor(rate))
        k = len(shape)
        z = rng.gamma(np.array(shape), 1.0 / np.array(rate), (300, k))
        z[:6] = -np.abs(rng.standard_normal((6, k))) * (np.array(shape) / np.array(rate))
        z[6, 0] = 0.0
        z[7:40] *= np.exp(rng.uniform(0.0, 6.5, (33, 1)))
        z[40:48] *= np.exp(rng.uniform(-12.0, -3.0, (8, 1)))
        out["gm_%s_shape" % tag] = np.array(shape, np.float32)
        out["gm_%s_rate" % tag] = np.array(rate, np.float32)
        out["gm_%s_z" % tag] = z
        out["gm_%s_log_prob" % tag] = g.log_prob(torch.from_numpy(z)).numpy()
    # torch.sum rows long enough for the 4-accumulator and cascade levels of ATen's sum (iSIR batches beyond 16)
    for n in (18, 31, 32, 33, 63, 64, 65, 100, 128, 255, 256, 257, 511, 512, 513, 1000, 2049, 4099):
        x = (rng.standard_normal((8, n)) * rng.choice([1e-3, 1.0, 1e3], (8, n))).astype(np.float32)
        out["rowsum_%d_x" % n] = x
        out["rowsum_%d_sum" % n] = np.array([torch.sum(torch.from_numpy(r)).item() for r in x], np.float32)
    # resample (GLMCMC_NFs.py:29-40): systematic resampling; the module imports the third-party normflows at its top
    # (absent here) but resample itself is plain torch -- an empty placeholder module satisfies the import
    sys.modules.setdefault("normflows", types.ModuleType("normflows"))
    import glabcmcmc.GLMCMC_NFs as rnf
    cases = []
    for i, (P, N, kind) in enumerate(((10, 10, "flat"), (1000, 1000, "exp"), (1000, 1000, "short"), (257, 300, "spiky"),
                                      (64, 7, "exp"), (5, 40, "short"), (1000, 1000, "zeros"))):
        w = {"flat": np.ones(P), "exp": np.exp(rng.standard_normal(P) * 2), "short": np.exp(rng.standard_normal(P)),
             "spiky": np.where(rng.random(P) < 0.02, 1.0, 1e-9), "zeros": np.where(rng.random(P) < 0.5, 0.0, rng.random(P))}[kind]
        w = (w / w.sum()).astype(np.float32)
        if kind == "short":
            w = (w * np.float32(0.83)).astype(np.float32)          # cumulative sum ends below 1: the last draws find no index
        u0 = np.float32(rng.integers(0, 1 << 24)) * np.float32(2.0 ** -24)
        saved = torch.rand
        torch.rand = lambda *a, u0=u0, **k: torch.tensor([u0])
        try:
            idx = rnf.resample(torch.from_numpy(w), N)
        finally:
            torch.rand = saved
        out["resample_%d_w" % i], out["resample_%d_u0" % i] = w, u0
        out["resample_%d_idx" % i] = idx.numpy().astype(np.int64)
        cases.append((i, P, N, kind, int(idx.numel())))
    out["resample_cases"] = np.array(repr(cases))
    # Gamma.forward (distribution.py:106-121): the reference draws scipy.stats.gamma.rvs from NumPy's generator; here rvs is
    # replaced by the build's specified draw (include/glabc_numerics.h glabc_gamma_draw, through the CPU checker) so that
    # the reference's forward() returns (z, log_prob(z)) for exactly the variates the device kernel makes
    import ctypes
    import scipy.stats
    from glabcmcmc_amd import distribution as bdist
    L = oracle_lib.load()
    for tag, shape, rate, seed, row0 in (("a", [2.0, 3.0], [1.0, 2.0], 11, 0), ("b", [0.5], [3.0], 12, 10 ** 12),
                                         ("c", [1.0, 7.5, 0.3], [0.5, 1.5, 4.0], 13, 77)):
        g = rdist.Gamma(torch.tensor(shape), torch.tensor(rate))
        desc = bdist.Gamma(torch.tensor(shape), torch.tensor(rate)).gamma_descriptor()
        n, k = 400, len(shape)
        zz, lp = np.empty((n, k)), np.empty(n)
        assert L.oracle_gamma_forward(ctypes.byref(desc), n, seed, row0, zz.ctypes.data, lp.ctypes.data) == 0
        saved = scipy.stats.gamma.rvs
        rdist.gamma.rvs = lambda a, scale=1, size=None, zz=zz: zz.copy()
        try:
            z_ref, lp_ref = g.forward(n)
        finally:
            rdist.gamma.rvs = saved
        out["gf_%s_shape" % tag], out["gf_%s_rate" % tag] = np.array(shape, np.float32), np.array(rate, np.float32)
        out["gf_%s_seed_row0" % tag] = np.array([seed, row0], np.int64)
        out["gf_%s_z" % tag], out["gf_%s_log_p" % tag] = z_ref.numpy(), lp_ref.numpy()
    # esjd (ESJD.py:2-25) beyond theta_dim 4: continues the esjd_chain_<i> sequence above
    for i, (T, d) in enumerate(((300, 6), (500, 8), (60, 5)), start=len(chains)):
        x = np.cumsum((rng.random((T, d)) < 0.3) * rng.standard_normal((T, d)), axis=0).astype(np.float32)
        out["esjd_chain_%d" % i] = x
        out["esjd_value_%d" % i] = np.float32(float(resjd.esjd(torch.from_numpy(x))))
    np.savez_compressed(os.path.join(HERE, "primitives.npz"), **out)
    print("primitives: %d arrays" % len(out))


def csv_fixture():
    """The CSV side effect as the reference writes it (GlobalMCMC.py:70-76 -- the variant that re-writes the previous
    block at the tail, SURVEY B9 -- and GLMCMC.py:105-111) for num_ite = 12 005: the chain and the SHA-256 of the file."""
    import hashlib
    import tempfile
    L = oracle_lib.load()
    out = {}
    T = 12004
    for algo, cfg in (("globalmcmc", dict(epsilon=0.3, gf=0.5, N=1, T=T, seed=61, local=G2(0.35), **{"global": G2(1.0)})),
                      ("glmcmc", dict(epsilon=0.3, gf=0.7, N=3, T=T, seed=62, local=G2(0.35), **{"global": G2(1.0)}))):
        P = cfg["N"]
        u, r, z = philox_tape(L, cfg["seed"], 0, T, P, 2, 2, False, False, cfg["gf"])
        tape = Tape(u, r, z, 2, np.float32(cfg["gf"]), algo == "glmcmc")
        theta0 = np.array([1.4, -1.3], np.float32)
        y0 = np.array([1.5, 1.45], np.float32)
        path = os.path.join(tempfile.mkdtemp(), "chain.csv")
        model, local, glob = make_model(cfg), make_dist(cfg["local"]), make_dist(cfg["global"])
        with patched(tape):
            if algo == "glmcmc":
                chain = rglmcmc.GLMCMC(model, T + 1, torch.from_numpy(theta0), torch.from_numpy(y0).view(1, -1), local, path,
                                       cfg["gf"], glob, cfg["N"])
            else:
                chain = rglobal.GlobalMCMC(model, T + 1, torch.from_numpy(theta0), torch.from_numpy(y0).view(1, -1), glob, path,
                                           cfg["gf"], local)
        data = open(path, "rb").read()
        lines = data.decode().splitlines()
        out[algo + "_chain"] = chain.numpy().copy()
        out[algo + "_sha256"] = np.array(hashlib.sha256(data).hexdigest())
        out[algo + "_n_lines"] = np.array(len(lines))
        out[algo + "_head"] = np.array("\n".join(lines[:3]))
        out[algo + "_tail"] = np.array("\n".join(lines[-3:]))
        out[algo + "_cfg"] = np.array(repr(cfg))
        print("csv %s: %d lines for %d iterations, %d bytes" % (algo, len(lines), T + 1, len(data)))
    np.savez_compressed(os.path.join(HERE, "csv.npz"), **out)


def gradient_fixture():
    """numberical_gradient_logABC (GLMALA.py:46-95) on a grid of thetas, noise from the Philox gradient slots,
    torch.sqrt correctly rounded (see _ieee_sqrt)."""
    import secrets
    L = oracle_lib.load()
    cfg = dict(epsilon=0.05, tau=0.3, num_grad=100, seed=41, local=G2(0.35), **{"global": G2(1.0)})
    rng = np.random.default_rng(41)
    n = 64
    theta = (rng.standard_normal((n, 2)) * 1.5).astype(np.float32)
    chain = rng.integers(0, 1 << 40, n)
    step = rng.integers(1, 1 << 20, n)
    grads = np.zeros((n, 2))
    model = Mixture_set(cfg["epsilon"])
    for i in range(n):
        noise = [np.zeros((cfg["num_grad"], 2), np.float32) for _ in range(2)]
        for k in range(2):
            L.oracle_grad_noise(cfg["seed"], int(chain[i]), int(step[i]), 1, k, cfg["num_grad"], 2, noise[k].ctypes.data)
        calls = [0]

        def randn(*a, **kw):
            k = calls[0] // 2
            calls[0] += 1
            return torch.from_numpy(noise[k])
        saved = (torch.manual_seed, torch.randn, np.random.seed, secrets.randbelow, torch.sqrt)
        torch.manual_seed, torch.randn, np.random.seed, secrets.randbelow, torch.sqrt = \
            (lambda s: None), randn, (lambda s: None), (lambda m: 7), _ieee_sqrt
        try:
            grads[i] = rglmala.numberical_gradient_logABC(model, torch.from_numpy(theta[i]), cfg["num_grad"]).numpy().ravel()
        finally:
            torch.manual_seed, torch.randn, np.random.seed, secrets.randbelow, torch.sqrt = saved
    np.savez_compressed(os.path.join(HERE, "glmala_gradient.npz"), theta=theta, chain=chain, step=step, grad=grads,
                        cfg=np.array(repr(cfg)), **reference_constants(cfg))
    print("glmala_gradient: %d points" % n)


class AGTape(Tape):
    """Tape for the reference's AGLMCMC (AGLMCMC.py:44-289): the per-iteration draws of the base tape (branch / accept
    uniforms, resampling double, the local move's 2 x (1, d) normals) plus the pool-side draws of the build's AGLMCMC
    (glabcmcmc_amd/AGLMCMC.py): the initial ISIR pool, the simulator noise of each pool, and for every KDE refit the
    4P centre-index uniforms + kernel noise.  torch.multinomial is replaced by the inverse-CDF rule of glabc_kde_sample
    (integer weights rint(w 2^40), first index whose inclusive prefix sum exceeds floor(u total))."""

    def __init__(self, L, key, T, P, d, gf):
        from glabcmcmc_amd.AGLMCMC import ISIR_STREAM, KDE_STREAM, POOL_STREAM
        u, r, z = philox_tape(L, key, 0, T, 1, d, d, False, False, gf)
        super().__init__(u, r, z, d, gf, True)
        self.L, self.key, self.P = L, key, P
        self.streams = (key ^ ISIR_STREAM, key ^ POOL_STREAM, key ^ KDE_STREAM)
        self.pools_drawn = 0          # simulator-noise blocks handed out
        self.init_done = False
        self.kde_calls = 0
        self.kde_u = None

    def _row_normals(self, seed, row0, n):
        unit = rdist.DiagGaussian(self.d, torch.zeros(self.d), torch.zeros(self.d))
        from glabcmcmc_amd import distribution as bdist
        desc = bdist.DiagGaussian(self.d, torch.zeros(self.d), torch.zeros(self.d)).descriptor()
        z = np.zeros((self.d, n), np.float32)
        lp = np.zeros(n, np.float32)
        import ctypes
        assert self.L.oracle_dist_forward_philox(ctypes.byref(desc), n, seed, row0, z.ctypes.data, lp.ctypes.data) == 0
        return torch.from_numpy(np.ascontiguousarray(z.T))

    def randn(self, *size, **kw):
        if len(size) == 1 and isinstance(size[0], (tuple, list)):
            size = tuple(size[0])
        n = size[0]
        if n == self.P:
            if not self.init_done:                                         # ISIR_prop.forward(P), AGLMCMC.py:80
                self.init_done = True
                return self._row_normals(self.streams[0], 0, n)
            k = self.pools_drawn                                           # generate_samples(Theta_prop0), :90 / :232
            self.pools_drawn += 1
            return self._row_normals(self.streams[1], k * self.P, n)
        if n == 4 * self.P:                                                # KDE.sample noise, kernel_density.py:147
            nrm = np.zeros((n, self.d), np.float32)
            uu = np.zeros(n, np.float64)
            self.L.oracle_kde_draws(self.streams[2], (self.kde_calls - 1) * n, n, self.d, uu.ctypes.data, nrm.ctypes.data)
            return torch.from_numpy(nrm)
        return self._noise(size)

    def multinomial(self, weights, n, replacement=True):
        assert replacement and n == 4 * self.P
        nrm = np.zeros((n, self.d), np.float32)
        uu = np.zeros(n, np.float64)
        self.L.oracle_kde_draws(self.streams[2], self.kde_calls * n, n, self.d, uu.ctypes.data, nrm.ctypes.data)
        self.kde_calls += 1
        wq = np.rint(weights.detach().numpy().astype(np.float64) * 2.0 ** 40).astype(np.int64)
        cum = np.cumsum(wq)
        target = np.floor(uu * float(cum[-1])).astype(np.int64)
        idx = np.minimum(np.searchsorted(cum, target, side="right"), len(cum) - 1)
        return torch.from_numpy(idx)


def aglmcmc_fixture():
    """The reference's AGLMCMC, unmodified, on the AGTape: its chain comes back through the CSV it writes (the function
    itself returns None, AGLMCMC.py:283-288)."""
    import csv
    import tempfile
    import glabcmcmc.AGLMCMC as raglmcmc
    L = oracle_lib.load()
    cfgs = {"aglmcmc_philox": dict(epsilon=0.3, seed=3, T=1500, gf=0.6, step_size=40, batch_size=5, alpha=0.8, hat_eps_T=0.5,
                                   local=G2(0.35), **{"global": ("gauss", [0.0, 0.0], [1.6487212, 1.6487212])})}
    for name, cfg in cfgs.items():
        T, P = cfg["T"], cfg["step_size"] * cfg["batch_size"]
        tape = AGTape(L, cfg["seed"], T, P, 2, cfg["gf"])
        model = Mixture_set(cfg["epsilon"])
        local = make_dist(cfg["local"])
        isir = rdist.DiagGaussian(2, torch.tensor([0.0, 0.0]), torch.tensor([0.5, 0.5]))
        theta0 = torch.tensor([1.5, 1.5])
        y0 = torch.tensor([[1.4, 1.7]])
        saved = torch.multinomial
        torch.multinomial = tape.multinomial
        out = os.path.join(tempfile.mkdtemp(), "chain.csv")
        try:
            with patched(tape):
                raglmcmc.AGLMCMC(model, T, theta0, y0, local, isir, out, cfg["gf"], cfg["step_size"], cfg["batch_size"],
                                 cfg["alpha"], cfg["hat_eps_T"], device="cpu")
        finally:
            torch.multinomial = saved
        rows = np.array([[np.float32(v) for v in row] for row in csv.reader(open(out))], np.float32)
        assert rows.shape == (T, 2), rows.shape
        np.savez_compressed(os.path.join(HERE, name + ".npz"), chain=rows, theta0=theta0.numpy(), y0=y0.numpy(),
                            kde_refits=np.array(tape.kde_calls), cfg=np.array(repr(cfg)), **reference_constants(cfg))
        print("%s: %d rows, %d KDE refits, %d iterations consumed" % (name, len(rows), tape.kde_calls, tape.t + 1))


def kde_fixture():
    """KernelDensity.fit / log_prob (kernel_density.py:70-128) and AGLMCMC's training weights (AGLMCMC.py:199-201,
    Mixture.py:47-53) evaluated by the reference on the CPU."""
    import glabcmcmc.kernel_density as rkde
    rng = np.random.default_rng(77)
    out = {}
    cases = [("a", 2, 400, True, "silverman"), ("b", 1, 64, False, "scott"), ("c", 3, 200, True, 0.3),
             ("d", 4, 150, True, torch.tensor([0.2, 0.5, 0.1, 1.5])), ("e", 2, 3000, True, "silverman")]
    for tag, d, n, weighted, bw in cases:
        X = (rng.standard_normal((n, d)) * rng.uniform(0.3, 2.0, d) + rng.uniform(-1, 1, d)).astype(np.float32)
        w = np.exp(rng.standard_normal(n) * 1.5).astype(np.float32) if weighted else None
        if tag == "c":
            w[:20] = 1e-30
        pts = np.concatenate([X[:40] + 0.1 * rng.standard_normal((40, d)), 4 * rng.standard_normal((40, d)),
                              60 * rng.standard_normal((16, d))]).astype(np.float32)
        k = rkde.KernelDensity(bandwidth=bw, device="cpu")
        k.fit(torch.from_numpy(X), None if w is None else torch.from_numpy(w))
        lp = k.log_prob(torch.from_numpy(pts))
        bw_out = k.bandwidth if isinstance(k.bandwidth, torch.Tensor) else torch.ones(d) * k.bandwidth
        out.update({tag + "_X": X, tag + "_pts": pts, tag + "_bandwidth": bw_out.numpy().astype(np.float32),
                    tag + "_weights": k.weights.numpy(), tag + "_log_prob": lp.numpy()})
        if w is not None:
            out[tag + "_w"] = w
        if not isinstance(bw, str):
            out[tag + "_bw_in"] = np.asarray(bw, np.float32).reshape(-1)
    out["cases"] = np.array(repr([(t, d, n, wt, bw if isinstance(bw, str) else "fixed") for t, d, n, wt, bw in cases]))
    # training weights under an annealed threshold
    model = Mixture_set(0.05)
    n = 512
    theta = (rng.standard_normal((n, 2)) * 1.2).astype(np.float32)
    dis = np.abs(rng.standard_normal(n) * 1.5).astype(np.float32)
    logq = (rng.standard_normal(n) - 2).astype(np.float32)
    for j, eps in enumerate([0.05, 0.7311, 2.5]):
        lw = model.prior_log_prob(torch.from_numpy(theta)) + model.calculate_log_kernel_dis(torch.from_numpy(dis), eps) \
            - torch.from_numpy(logq)
        out["tw%d" % j] = torch.exp(lw).numpy()
        ls = torch.log(torch.tensor([eps]))
        out["tw%d_consts" % j] = np.array([eps, float(ls), float(torch.exp(ls))], np.float64)
    out.update(tw_theta=theta, tw_dis=dis, tw_logq=logq)
    cfg = dict(epsilon=0.05, local=G2(0.35), **{"global": G2(1.0)})
    out.update(reference_constants(cfg))
    out["cfg"] = np.array(repr(cfg))
    np.savez_compressed(os.path.join(HERE, "kde.npz"), **out)
    print("kde: %d arrays" % len(out))


def gamma_candidates_fixture():
    """Gamma as the importance proposal INSIDE the samplers (include/glabc.h GLABC_DIST_GAMMA): the variates come from the
    chain's Gamma slots (include/glabc_numerics.h glabc_gamma_draw_candidate, through the CPU checker's test hook); the
    reference's own Gamma.forward (distribution.py:106-121), with scipy's draw replaced by exactly those variates, returns
    (z, log_prob(z)) -- stored, so that the checker's (and the kernels') log q' is pinned to the reference's float64
    log(pdf) sum.  Also Gamma.log_prob (distribution.py:123-137) at float32 points, as a prior is evaluated."""
    import ctypes
    import scipy.stats
    from glabcmcmc_amd import distribution as bdist
    L = oracle_lib.load()
    L.oracle_gamma_candidates.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int,
                                          ctypes.c_void_p, ctypes.c_void_p]
    rng = np.random.default_rng(424242)
    out, cases = {}, []
    for i, (shape, rate, seed, chain, step, N) in enumerate((
            ([4.0, 4.0], [3.0, 3.0], 5, 0, 1, 64), ([0.5, 2.5], [1.0, 0.25], 77, 10 ** 12 + 3, 4000000000, 40),
            ([1.0], [2.0], 9, 123456, 17, 100), ([7.5, 0.3, 1.0, 2.0], [1.5, 4.0, 0.5, 1.0], 2026, 65535, 2000, 33),
            ([3.0, 3.0, 3.0, 3.0, 3.0, 3.0, 3.0, 3.0], [1.0, 2.0, 3.0, 4.0, 0.5, 0.25, 8.0, 1.5], 1, 1, 1, 12))):
        k = len(shape)
        g = rdist.Gamma(torch.tensor(shape), torch.tensor(rate))
        desc = bdist.Gamma(torch.tensor(shape), torch.tensor(rate)).descriptor()
        zz, lp = np.empty((N, k)), np.empty(N)
        assert L.oracle_gamma_candidates(ctypes.byref(desc), seed, chain, step, N, zz.ctypes.data, lp.ctypes.data) == 0
        saved = scipy.stats.gamma.rvs
        rdist.gamma.rvs = lambda a, scale=1, size=None, zz=zz: zz.copy()
        try:
            z_ref, lp_ref = g.forward(N)
        finally:
            rdist.gamma.rvs = saved
        assert np.array_equal(z_ref.numpy(), zz)
        # log_prob at float32 points (a prior's use): positive, zero, negative, far in the tail (pdf underflows -> -inf)
        pts = np.abs(rng.standard_normal((200, k))).astype(np.float32) * np.float32(2.0)
        pts[0, 0], pts[1, 0], pts[2, 0], pts[3, 0] = 0.0, -0.5, 3000.0, 1e-30
        lp_pts = g.log_prob(torch.from_numpy(pts).double()).numpy()
        out["gc_%d_shape" % i], out["gc_%d_rate" % i] = np.array(shape, np.float32), np.array(rate, np.float32)
        out["gc_%d_z" % i], out["gc_%d_log_p" % i] = z_ref.numpy(), lp_ref.numpy()
        out["gc_%d_pts" % i], out["gc_%d_pts_log_prob" % i] = pts, lp_pts
        cases.append((i, seed, chain, step, N))
    out["cases"] = np.array(repr(cases))
    np.savez_compressed(os.path.join(HERE, "gamma_candidates.npz"), **out)
    print("gamma_candidates: %d arrays" % len(out))


if __name__ == "__main__":
    want = sys.argv[1:]
    if not want or "gamma_candidates" in want:
        gamma_candidates_fixture()
    if not want or "kde" in want:
        kde_fixture()
    if not want or "aglmcmc" in want:
        aglmcmc_fixture()
    if not want or "primitives" in want:
        primitives()
    if not want or "csv" in want:
        csv_fixture()
    if not want or "glmala_gradient" in want:
        gradient_fixture()
    for name, (algo, cfg, mode) in SAMPLER_FIXTURES.items():
        if not want or name in want or algo in want:
            sampler_fixture(name, algo, cfg, mode)
