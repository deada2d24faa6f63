"""Single-chain iteration rates of the host-orchestrated samplers (launch-bound): python tools/host_loop_bench.py"""
import sys, time
sys.path.insert(0, "gl-abc-mcmc_amd")
import torch
from glabcmcmc_amd import AGLMCMC, GLMCMC, GLMCMC_NF, distribution
from glabcmcmc_amd.examples.Mixture import Mixture_set
torch.manual_seed(0)
M = Mixture_set(0.3)
lp = distribution.DiagGaussian(2, loc=torch.zeros(1, 2), log_scale=torch.log(torch.tensor([0.35, 0.35])))
ip = distribution.DiagGaussian(2, torch.tensor([0.0, 0.0]), torch.tensor([0.5, 0.5]))
for C in (1, 4096):
    th0 = torch.zeros(C, 2) + 1.5 if C > 1 else torch.tensor([1.5, 1.5])
    y0 = M.generate_samples(th0)
    T = 3000
    for name, fn in (("GLMCMC", lambda: GLMCMC(M, T, th0, y0, lp, None, 0.6, ip, 5, seed=1, verbose=False)),
                     ("GLMCMC_NF(32 couplings)", lambda: GLMCMC_NF(M, T, th0, y0, lp, None, 0.6, 100, 5, None, 3, seed=1, verbose=False)),
                     ("AGLMCMC", lambda: AGLMCMC(M, T, th0, y0, lp, ip, None, 0.6, 100, 5, 0.8, 0.5, seed=1, verbose=False))):
        fn() if name == "GLMCMC" else None
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print("C=%5d %-24s %7.1f iterations/s  (%.3g chain-steps/s)" % (C, name, T / dt, C * T / dt))
