import sys, time; sys.path.insert(0, "gl-abc-mcmc_amd")
import torch
from glabcmcmc_amd import KernelDensity
torch.manual_seed(0)
for S, P in ((2048, 65536), (8192, 524288), (500, 500), (500, 1)):
    k = KernelDensity(device="cuda", seed=1).fit(torch.randn(S, 2), torch.rand(S))
    pts = torch.randn(2, P, device="cuda")
    for _ in range(2): k.log_prob_soa(pts)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): k.log_prob_soa(pts)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print("S=%d P=%d  %.3f ms  %.3g pair-evals/s" % (S, P, dt * 1e3, S * P / dt))
