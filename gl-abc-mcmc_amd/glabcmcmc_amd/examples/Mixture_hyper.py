"""Global-frequency sweep (reference: examples/Mixture_hyper.py:12-41) on the GPU.

The reference scores each global_frequency by  esjd(chain) / mean-seconds-per-iteration  over 10 seeds x 11
frequencies of 1 000-iteration single-chain runs (Mixture_hyper.py:23-39) and reports the arg-max.  Here every
(frequency, seed) cell is a batch of independent chains in one fused launch; the score is the same quantity --
ESJD (ESJD.py, mean over the cell's chains) divided by the measured seconds per iteration of that launch.

    python -m glabcmcmc_amd.examples.Mixture_hyper          (needs an MI355X)
"""
import numpy as np
import torch

from .. import distribution, engine
from .Mixture import Mixture_set

GLOBAL_FREQUENCIES = [0, 0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9, 1]        # Mixture_hyper.py:24
SEEDS = [1, 2, 3, 4, 5, 6, 7, 8, 9, 10]                                          # Mixture_hyper.py:23


def sweep_fused(epsilon=0.05, num_ite=1000, chains_per_cell=4096, batch_size=5, frequencies=GLOBAL_FREQUENCIES, seeds=SEEDS,
                device=None):
    """The whole (seed x frequency) grid as ONE fused launch: cell (i, j) is a block of chains_per_cell chains with
    per-chain global_frequency = frequencies[j] (glabc_run.global_frequency_per_chain); `seeds` only counts the
    replicates -- the chains of a block are independent through their chain ids.  -> esjd [seeds x frequencies].
    (There is no per-cell time in a fused launch: use sweep() for the reference's ESJD-per-second score.)"""
    dev = engine.require_device(device)
    Model = Mixture_set(epsilon)
    model = Model.descriptor()
    lp = distribution.DiagGaussian(2, loc=torch.zeros(1, 2), log_scale=torch.log(torch.tensor([0.35, 0.35]))).descriptor()
    ip = distribution.DiagGaussian(2, torch.tensor([0.0, 0.0]), torch.tensor([0.0, 0.0])).descriptor()
    cells, m = len(seeds) * len(frequencies), int(chains_per_cell)
    n = cells * m
    g = torch.Generator().manual_seed(1)
    chains = engine.ChainBatch(torch.zeros(n, 2), (0.05 ** 0.5) * torch.randn(n, 2, generator=g), dev)
    engine.init_weights(model, ip, chains)
    gf = torch.tensor(frequencies, dtype=torch.float32).repeat(len(seeds)).repeat_interleave(m).to(dev)
    mom = engine.Moments(n, 2, dev)
    engine.run_steps("glabc_glmcmc_steps", model, lp, ip, chains, num_ite - 1, 1, 1, 0.0, batch_size, moments=mom, gf_per_chain=gf)
    e = mom.esjd().view(cells, m).double()
    e = torch.where(torch.isfinite(e), e, torch.zeros_like(e))
    return e.mean(dim=1).view(len(seeds), len(frequencies)).cpu().numpy()


def sweep(epsilon=0.05, num_ite=1000, chains_per_cell=4096, batch_size=5, frequencies=GLOBAL_FREQUENCIES, seeds=SEEDS,
          device=None, verbose=True):
    """-> dict(best_gf, resjd_mean [len(frequencies)], esjd [seeds x frequencies], sec_per_iter [seeds x frequencies])"""
    dev = engine.require_device(device)
    Model = Mixture_set(epsilon)
    model = Model.descriptor()
    lp = distribution.DiagGaussian(2, loc=torch.zeros(1, 2), log_scale=torch.log(torch.tensor([0.35, 0.35]))).descriptor()
    ip = distribution.DiagGaussian(2, torch.tensor([0.0, 0.0]), torch.tensor([0.0, 0.0])).descriptor()
    n = int(chains_per_cell)
    esjd = np.zeros((len(seeds), len(frequencies)))
    sec = np.zeros_like(esjd)
    for i, seed in enumerate(seeds):
        g = torch.Generator().manual_seed(seed)
        theta0 = torch.zeros(n, 2)
        y0 = (0.05 ** 0.5) * torch.randn(n, 2, generator=g)                       # y0 = generate_samples(theta0), Mixture_hyper.py:16
        for j, gf in enumerate(frequencies):
            chains = engine.ChainBatch(theta0, y0, dev)
            engine.init_weights(model, ip, chains)
            mom = engine.Moments(n, 2, dev)
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record()
            engine.run_steps("glabc_glmcmc_steps", model, lp, ip, chains, num_ite - 1, 1, seed, gf, batch_size, moments=mom)
            t1.record()
            torch.cuda.synchronize(dev)
            e = mom.esjd()
            esjd[i, j] = float(e[torch.isfinite(e)].double().mean())
            sec[i, j] = t0.elapsed_time(t1) * 1e-3 / (num_ite - 1)
    resjd = esjd / sec                                                              # Mixture_hyper.py:37
    resjd_mean = resjd.mean(axis=0)                                                 # :38
    best = frequencies[int(np.argmax(resjd_mean))]                                  # :39
    if verbose:
        print("*****************************")
        print(f"The best global frequency: {best}")                                # :40-41
    return {"best_gf": best, "resjd_mean": resjd_mean, "esjd": esjd, "sec_per_iter": sec}


if __name__ == "__main__":
    out = sweep()
    for gf, e, r in zip(GLOBAL_FREQUENCIES, out["esjd"].mean(0), out["resjd_mean"]):
        print("gf %.1f  ESJD %.5f  ESJD/sec-per-iter %.4g" % (gf, e, r))
