"""The drop-in return path: GLMCMC(...) -> Theta_Re in HOST memory (the reference returns a CPU tensor, GLMCMC.py:137), 65 536
chains x 2000 iterations, end to end on the wall clock.  python tools/host_return_bench.py  (needs an MI355X)

  mirror   the default since round 3: rows leave for pinned host memory launch by launch while the next launch computes
           (_host.HostMirror)
  after    the round-2 path: one pinned copy after the run
  device   return_device=True: no copy at all (what bench.py times)
"""
import sys, time
sys.path.insert(0, "gl-abc-mcmc_amd")
import torch
import glabcmcmc_amd as g
from glabcmcmc_amd import _host, distribution
from glabcmcmc_amd.examples.Mixture import Mixture_set

n, T = 65536, 2000
gen = torch.Generator().manual_seed(0)
theta0 = torch.randn(n, 2, generator=gen)
y0 = theta0.abs() + (0.05 ** 0.5) * torch.randn(n, 2, generator=gen)
lp = distribution.DiagGaussian(2, torch.zeros(2), torch.log(torch.tensor([0.35, 0.35])))
ip = distribution.DiagGaussian(2, torch.zeros(2), torch.zeros(2))
model = Mixture_set(0.05)


def run(**kw):
    return g.GLMCMC(model, T + 1, theta0, y0, lp, None, 0.9, ip, 5, seed=1, verbose=False, **kw)


def timed(name, fn, reps=4):
    out = fn(); torch.cuda.synchronize()
    keep = out.clone() if out.is_cuda else out.clone()
    del out                                                  # a caller that consumes and drops the result: torch's caching host
    t0 = time.perf_counter()                                 # allocator then hands the same pinned block to the next run
    for _ in range(reps):
        out = fn()
        shape, dev = tuple(out.shape), out.device
        del out
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print("%-8s %.1f ms per run of %d x %d -> %.2e chain-steps/s  (%s, %s)" % (name, dt * 1e3, n, T, n * T / dt, shape, dev), flush=True)
    return keep


t0 = time.perf_counter(); pin = torch.empty(T + 1, 2, n, pin_memory=True); t1 = time.perf_counter()
print("a fresh pinned buffer of %.2f GB: %.1f ms" % (pin.numel() * 4 / 1e9, (t1 - t0) * 1e3)); del pin
t0 = time.perf_counter(); pin = torch.empty(T + 1, 2, n, pin_memory=True); t1 = time.perf_counter()
print("the same from the caching host allocator: %.2f ms" % ((t1 - t0) * 1e3)); del pin
a = timed("mirror", run)
wanted = _host.HostMirror.wanted
_host.HostMirror.wanted = staticmethod(lambda *a_: False)
b = timed("after", run)
_host.HostMirror.wanted = wanted
c = timed("device", lambda: run(return_device=True))
assert torch.equal(a, b) and torch.equal(a, c.cpu())
print("same rows from all three")
