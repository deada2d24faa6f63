"""Level-1 restatement of the hot path: ONE chain, one ATen operation at a time, the way the reference runs it.

TEST INFRASTRUCTURE ONLY (like everything under oracle/): used by tests/ (pinned against the tapes the reference's own loops
were driven with, tests/test_oracle_golden.py) and by bench.py's cpu_baseline leg, where it stands for the reference's CPU
path on the GPU box's host cores -- the reference itself cannot travel there.  The product never imports it.

What it restates (arithmetic and operation order; every tensor operation below is one ATen dispatch on a 1..N-row CPU tensor,
which is what the reference spends its time on):
  * distribution.py:166-181  DiagGaussian.forward / log_prob  (exp(log_scale) recomputed per call, true division, x*x)
  * distribution.py:73-86    Uniform.forward / log_prob
  * examples/Mixture.py:13-45 Mixture_set: |theta| + sqrt(0.05) N(0, I), standard-normal prior, Euclidean discrepancy,
                             Gaussian ABC kernel -- with the per-call construction of the distribution's constants
  * GLMCMC.py:7-22           weight_sampling (double running sum over a Python list)
  * GLMCMC.py:58-104         the GLMCMC iteration (iSIR global move with the cached log-weight and its dirty flag, RW-MH local
                             move with the prior-sentinel redraw)
  * GlobalMCMC.py:37-68      the GlobalMCMC iteration (independence MH / RW-MH)
Random numbers come from a `draws` object: TorchDraws = the reference's generators (torch global generator + NumPy's global
uniform), TapeDraws = the numbers of a stored tape in the order the loop asks for them.
"""
import math

import numpy as np
import torch

SENTINEL = 7 * math.log(1e-10)          # GLMCMC.py:92: a prior's "outside the support" value


# ------------------------------------------------------------------------------------------------------- random numbers
class TorchDraws:
    def branch(self, t):
        return torch.rand(1)

    accept = branch

    def proposal(self, t, n, d, uniform=False):
        return torch.rand(n, d) if uniform else torch.randn(n, d)

    def simulator(self, t, n, d):
        return torch.randn(n, d)

    def resample(self, t):
        return np.random.uniform(0, 1)


class TapeDraws:
    """u[T][2] (branch, accept), r[T] float64, z[T][P][d + y_dim] (proposal noise | simulator noise) of one chain"""

    def __init__(self, u, r, z, d):
        self.u, self.r, self.z, self.d = u, r, z, d

    def branch(self, t):
        return torch.tensor([self.u[t, 0]], dtype=torch.float32)

    def accept(self, t):
        return torch.tensor([self.u[t, 1]], dtype=torch.float32)

    def proposal(self, t, n, d, uniform=False):
        return torch.from_numpy(np.ascontiguousarray(self.z[t, :n, :d]))

    def simulator(self, t, n, d):
        return torch.from_numpy(np.ascontiguousarray(self.z[t, :n, self.d:self.d + d]))

    def resample(self, t):
        return float(self.r[t])


# ------------------------------------------------------------------------------------------------------- distributions
class Gauss:
    """distribution.py:143-203"""

    def __init__(self, loc, log_scale):
        self.loc, self.log_scale = loc, log_scale
        self.d = loc.numel()
        self.uniform = False

    def forward(self, eps):
        z = self.loc + torch.exp(self.log_scale) * eps                                           # :170
        log_p = -0.5 * self.d * np.log(2 * np.pi) - torch.sum(self.log_scale + 0.5 * torch.pow(eps, 2), 1)   # :171-173
        return z, log_p

    def log_prob(self, z):
        eps = (z - self.loc) / torch.exp(self.log_scale)                                         # :178
        return -0.5 * self.d * np.log(2 * np.pi) - torch.sum(self.log_scale + 0.5 * torch.pow(eps, 2), 1)


class Box:
    """distribution.py:50-86"""

    def __init__(self, low, high):
        self.low, self.high = low, high
        self.d = low.numel()
        self.uniform = True
        self.log_prob_val = -torch.log(torch.prod(high - low))                                   # :71

    def forward(self, eps):
        z = self.low + (self.high - self.low) * eps                                              # :77
        return z, self.log_prob_val * torch.ones(eps.shape[0])

    def log_prob(self, z):
        log_p = self.log_prob_val * torch.ones(z.shape[0])
        outside = torch.any(torch.logical_or(z < self.low, z > self.high).reshape(z.shape[0], -1), dim=-1)
        log_p[outside] = -np.inf                                                                 # :85
        return log_p


def make_distribution(spec):
    if spec[0] == "gauss":
        return Gauss(torch.tensor(spec[1], dtype=torch.float32), torch.log(torch.tensor(spec[2], dtype=torch.float32)))
    if spec[0] == "uniform":
        return Box(torch.tensor(spec[1], dtype=torch.float32), torch.tensor(spec[2], dtype=torch.float32))
    raise ValueError(spec[0])


# ------------------------------------------------------------------------------------------------------- the example Model
class Mixture:
    """examples/Mixture.py:5-53; the constants are rebuilt on every call, as there"""

    def __init__(self, epsilon, y_obs=(1.5, 1.5)):
        self.epsilon = epsilon
        self.y_obs = torch.tensor([list(y_obs)], dtype=torch.float32)
        self.theta_dim = self.y_obs.shape[1]

    def simulate(self, theta, eps):                                                              # :13-26
        noise = Gauss(torch.zeros(self.theta_dim), torch.log(torch.tensor([0.05] * self.theta_dim).sqrt()))
        return torch.abs(theta) + noise.forward(eps)[0]

    def prior(self, theta):                                                                      # :28-31
        return Gauss(torch.zeros(self.theta_dim), torch.zeros(self.theta_dim)).log_prob(theta)

    def discrepancy(self, y):                                                                    # :33-36
        return torch.sqrt(torch.sum((y - self.y_obs.view(1, -1)) ** 2, dim=1))

    def log_kernel(self, y):                                                                     # :38-45
        dis = self.discrepancy(y.view(-1, self.y_obs.shape[1]))
        k = Gauss(torch.tensor([0.0]), torch.log(torch.tensor([self.epsilon])))
        return k.log_prob(dis.view(-1, 1))


def weight_sampling(w_list, ran):                                                                # GLMCMC.py:7-22
    s = 0
    for j in range(len(w_list)):
        s += w_list[j]
        if ran < s:
            return j
    return None


# ------------------------------------------------------------------------------------------------------- the loops
def glmcmc(model, T, theta0, y0, local, importance, gf, N, draws):
    """GLMCMC.py:48-104 for one chain: returns Theta_Re (T + 1, d)"""
    theta = theta0.view(1, -1)
    y = y0.view(1, -1)
    dirty = True                                                                                 # `local`, :50
    lw_old = (model.prior(theta) + model.log_kernel(y) - importance.log_prob(theta)).view(-1)    # :52-55
    out = torch.zeros(T + 1, theta.shape[1])
    out[0, :] = theta.clone()
    for i in range(1, T + 1):
        t = i - 1
        if draws.branch(t) < gf:                                                                 # :59
            if dirty:                                                                            # :60-64
                lw_old = (model.prior(theta) + model.log_kernel(y) - importance.log_prob(theta)).view(-1)
            dirty = False
            prop, lq = importance.forward(draws.proposal(t, N, theta.shape[1], importance.uniform))   # :66
            keep = torch.all(~torch.isnan(prop), dim=1)                                          # :67-70
            prop, lq = prop[keep].clone(), lq[keep].clone()
            x = model.simulate(prop, draws.simulator(t, prop.shape[0], y.shape[1]))              # :71
            lw0 = model.prior(prop) + model.log_kernel(x) - lq                                   # :72-74
            lw = torch.cat((lw_old, lw0))
            cand = torch.cat((theta, prop), dim=0)
            xs = torch.cat((y.view(1, -1), x), dim=0)
            w = torch.exp(lw)                                                                    # :78
            w[torch.isnan(w)] = 0.0                                                              # :80-81
            w = w / torch.sum(w)                                                                 # :82
            ind = weight_sampling(w.tolist(), draws.resample(t))                                 # :83
            if ind is not None and ind != 0:                                                     # :84-88
                theta = cand[ind, :].clone().view(1, -1)
                lw_old = lw[ind].clone().view(-1)
                y = xs[ind, :].clone().view(1, -1)
        else:
            prop = local.forward(draws.proposal(t, 1, theta.shape[1], local.uniform))[0] + theta  # :91
            while model.prior(prop) == SENTINEL:                                                 # :92-93 (never for this Model)
                prop = local.forward(draws.proposal(t, 1, theta.shape[1], local.uniform))[0] + theta
            x = model.simulate(prop, draws.simulator(t, 1, y.shape[1]))[0,].clone()              # :94-95
            log_acc = model.prior(prop) + model.log_kernel(x) - model.prior(theta) - model.log_kernel(y)   # :96-97
            if torch.log(draws.accept(t)) < log_acc:                                             # :98-99
                dirty = True
                theta = prop.clone()
                y = x.clone().view(1, -1)
        out[i, :] = theta.clone()
    return out


def globalmcmc(model, T, theta0, y0, global_prop, local, gf, draws):
    """GlobalMCMC.py:31-68 for one chain"""
    theta = theta0.view(1, -1)
    y = y0.view(1, -1)
    out = torch.zeros(T + 1, theta.shape[1])
    out[0, :] = theta.clone()
    for i in range(1, T + 1):
        t = i - 1
        if draws.branch(t) < gf:                                                                 # :39
            prop, lq = global_prop.forward(draws.proposal(t, 1, theta.shape[1], global_prop.uniform))   # :40
            x = model.simulate(prop, draws.simulator(t, 1, y.shape[1]))[0,].clone()
            log_acc = model.prior(prop) + model.log_kernel(x) + global_prop.log_prob(theta) - lq \
                - model.prior(theta) - model.log_kernel(y)                                       # :44-46
        else:
            prop = local.forward(draws.proposal(t, 1, theta.shape[1], local.uniform))[0] + theta  # :56
            x = model.simulate(prop, draws.simulator(t, 1, y.shape[1]))[0,].clone()
            log_acc = model.prior(prop) + model.log_kernel(x) - model.prior(theta) - model.log_kernel(y)   # :60-61
        if torch.log(draws.accept(t)) < log_acc:                                                 # :47,62
            theta = prop.clone()
            y = x.clone().view(1, -1)
        out[i, :] = theta.clone()
    return out


def steps_per_second(seconds=3.0, epsilon=0.05, gf=0.9, N=5):
    """BASELINE configs[1]'s chain (GLMCMC, N = 5, gf 0.9, Mixture_set eps 0.05) on ONE core for about `seconds`:
    iterations per second of this process, torch limited to one thread as the reference's per-chain process would be"""
    import time
    threads = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        model = Mixture(epsilon)
        local = make_distribution(("gauss", [0.0, 0.0], [0.35, 0.35]))
        imp = make_distribution(("gauss", [0.0, 0.0], [1.0, 1.0]))
        theta0 = torch.zeros(2)
        y0 = model.simulate(theta0.view(1, -1), torch.randn(1, 2))
        glmcmc(model, 200, theta0, y0, local, imp, gf, N, TorchDraws())                          # warm-up
        done, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            glmcmc(model, 500, theta0, y0, local, imp, gf, N, TorchDraws())
            done += 500
        return done / (time.perf_counter() - t0)
    finally:
        torch.set_num_threads(threads)
