"""Throughput of streaming.stream_history at the benchmark shape (history to a .npy in /dev/shm): python tools/stream_bench.py"""
import os, sys, time
sys.path.insert(0, "gl-abc-mcmc_amd")
import torch
from glabcmcmc_amd import distribution, engine, streaming
from glabcmcmc_amd.examples.Mixture import Mixture_set
M = Mixture_set(0.05)
model = M.descriptor()
lp = distribution.DiagGaussian(2, loc=torch.zeros(1, 2), log_scale=torch.log(torch.tensor([0.35, 0.35]))).descriptor()
ip = distribution.DiagGaussian(2, torch.tensor([0.0, 0.0]), torch.tensor([0.0, 0.0])).descriptor()
n, T = 65536, 12000
dev = torch.device("cuda", 0)
chains = engine.ChainBatch(torch.zeros(n, 2), torch.zeros(n, 2) + 0.2, dev)
engine.init_weights(model, ip, chains)
path = "/dev/shm/glabc_stream_bench.npy"
for rep in range(2):
    t0 = time.perf_counter()
    streaming.stream_history("glabc_glmcmc_steps", model, lp, ip, chains, T, 1, 0.9, 5, path, step0=1 + rep * T, block=2000)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("run %d: %d chains x %d iterations, %.2f GB of history in %.2f s -> %.3g chain-steps/s, %.1f GB/s to the file" %
          (rep, n, T, n * T * 8 / 1e9, dt, n * T / dt, n * T * 8 / dt / 1e9))
os.remove(path)
