"""A large sample of the reference's GLMALA law, for a statistical parity test of the gfx950 kernel.

Runs ONLY in the build container (imports /root/reference).  The reference's GLMALA (GLMALA.py:118-230) runs
UNMODIFIED and AS IS -- its own torch.sqrt, its own torch / NumPy global generators, its reseeding inside every
gradient (GLMALA.py:74-81) -- one chain at a time, BASELINE config 3 (Mixture_set eps 0.05, gf 0.8, N 5, tau 0.3,
num_grad 100; theta0 = 0, y0 = |theta0| + sqrt(0.05) z), T iterations per chain.  The only substitution: the OS
entropy of `secrets.randbelow` is replaced by a seeded NumPy generator, so this script is reproducible.

Per chain a few time averages are kept (E|theta_j|, E theta_j^2, jump second moments, ESJD, move rate, final state);
the fixture stores their means over chunks of chains, from which the test derives pooled means and standard errors.

    python tests/golden/make_glmala_stats.py run  [--workers 5] [--chains 150000] [--out /tmp/glmala_stats]
    python tests/golden/make_glmala_stats.py collect [--out /tmp/glmala_stats]     # -> tests/golden/glmala_stats.npz

The per-chain rows live only in --out (scratch); the fixture keeps the chunk means.  To EXTEND a committed fixture after the
scratch directory is gone, run further chunks (`run --first-chunk K0`, K0 beyond every chunk the fixture was made from: chain
ids, and with them every seed, are k * 64 + c) and `collect --merge`: the fixture's chunk means and the new ones are pooled
(mean of chunk means; standard error = their standard deviation / sqrt(number of chunks) -- chunks are independent and of
equal size, so this is the same estimator as the per-chain one).
"""
import argparse
import contextlib
import glob
import io
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
CFG = dict(epsilon=0.05, gf=0.8, N=5, tau=0.3, num_grad=100, T=500)
CHUNK = 64
STAT_NAMES = ["mean_abs_0", "mean_abs_1", "mean_sq_0", "mean_sq_1", "jump_00", "jump_01", "jump_11", "esjd",
              "move_rate", "final_abs_0", "final_abs_1", "final_sq_0", "final_sq_1"]


def _import_reference():
    sys.dont_write_bytecode = True
    pkg = types.ModuleType("glabcmcmc")
    pkg.__path__ = [os.path.join(REF, "glabcmcmc")]
    sys.modules["glabcmcmc"] = pkg
    sys.path.insert(0, os.path.join(REF, "glabcmcmc", "examples"))
    import glabcmcmc.distribution as rdist
    import glabcmcmc.GLMALA as rglmala
    from Mixture import Mixture_set
    return rdist, rglmala, Mixture_set


def chain_stats(chain):
    """chain (T+1, 2) float32 -> the STAT_NAMES row (float64)"""
    x = chain[1:].astype(np.float64)
    d = np.diff(chain.astype(np.float64), axis=0)
    T = x.shape[0]
    jj = d.T @ d / T
    det = jj[0, 0] * jj[1, 1] - jj[0, 1] ** 2
    moved = (np.abs(np.diff(chain, axis=0)).sum(1) > 0).mean()
    return np.array([np.abs(x[:, 0]).mean(), np.abs(x[:, 1]).mean(), (x[:, 0] ** 2).mean(), (x[:, 1] ** 2).mean(),
                     jj[0, 0], jj[0, 1], jj[1, 1], np.sqrt(max(det, 0.0)), moved,
                     abs(x[-1, 0]), abs(x[-1, 1]), x[-1, 0] ** 2, x[-1, 1] ** 2])


def run_chunk(args):
    k, out_dir = args
    path = os.path.join(out_dir, "chunk_%06d.npy" % k)
    if os.path.exists(path):
        return k
    import secrets
    rdist, rglmala, Mixture_set = _import_reference()
    torch.set_num_threads(1)
    model = Mixture_set(CFG["epsilon"])
    imp = rdist.DiagGaussian(2, torch.tensor([0.0, 0.0]), torch.tensor([0.0, 0.0]))
    rows = np.zeros((CHUNK, len(STAT_NAMES)))
    saved = secrets.randbelow
    try:
        for c in range(CHUNK):
            cid = k * CHUNK + c
            ent = np.random.Generator(np.random.PCG64(1_000_003 * cid + 17))
            secrets.randbelow = lambda n, ent=ent: int(ent.integers(0, n))
            torch.manual_seed(2_000_003 * cid + 5)
            np.random.seed((3_000_017 * cid + 11) % (2 ** 32))
            theta0 = torch.zeros(2)
            y0 = (torch.abs(theta0) + (0.05 ** 0.5) * torch.randn(2)).view(1, -1)
            with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
                out = rglmala.GLMALA(model, CFG["T"] + 1, theta0, y0, CFG["tau"], CFG["num_grad"], None, CFG["gf"], imp,
                                     CFG["N"])
            rows[c] = chain_stats(out.numpy())
    finally:
        secrets.randbelow = saved
    np.save(path + ".tmp.npy", rows)
    os.replace(path + ".tmp.npy", path)
    return k


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["run", "collect"])
    ap.add_argument("--workers", type=int, default=5)
    ap.add_argument("--chains", type=int, default=150000)
    ap.add_argument("--out", default="/tmp/glmala_stats")
    ap.add_argument("--first-chunk", type=int, default=0)
    ap.add_argument("--merge", action="store_true", help="collect: pool with the chunk means of the committed fixture")
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    if a.what == "run":
        import multiprocessing as mp
        n_chunks = (a.chains + CHUNK - 1) // CHUNK
        with mp.Pool(a.workers) as pool:
            for i, k in enumerate(pool.imap_unordered(run_chunk, [(k, a.out) for k in range(a.first_chunk, a.first_chunk + n_chunks)])):
                if i % 20 == 0:
                    print("chunk %d done (%d / %d)" % (k, i + 1, n_chunks), flush=True)
        return
    files = sorted(glob.glob(os.path.join(a.out, "chunk_*.npy")))
    files = [f for f in files if not f.endswith(".tmp.npy")]
    rows = np.stack([np.load(f) for f in files])                 # (chunks, CHUNK, stats)
    fixture = os.path.join(HERE, "glmala_stats.npz")
    if a.merge:
        old = np.load(fixture)
        assert int(old["chunk"]) == CHUNK and str(old["cfg"]) == repr(CFG) and str(old["names"]) == repr(STAT_NAMES)
        ids = [int(os.path.basename(f)[6:12]) for f in files]
        assert min(ids) >= old["chunk_means"].shape[0], "new chunks must lie beyond the fixture's (seeds are per chain id)"
        cm = np.concatenate([old["chunk_means"], rows.mean(1)])
        n_chains = cm.shape[0] * CHUNK
        mean, se = cm.mean(0), cm.std(0, ddof=1) / np.sqrt(cm.shape[0])
    else:
        flat = rows.reshape(-1, rows.shape[-1])
        cm, n_chains = rows.mean(1), flat.shape[0]
        mean, se = flat.mean(0), flat.std(0, ddof=1) / np.sqrt(flat.shape[0])
    np.savez_compressed(fixture, chunk_means=cm, n_chains=np.array(n_chains),
                        chunk=np.array(CHUNK), mean=mean, se=se, names=np.array(repr(STAT_NAMES)), cfg=np.array(repr(CFG)))
    for n, m, s in zip(STAT_NAMES, mean, se):
        print("%-12s %.6f +- %.6f  (rel %.2e)" % (n, m, s, s / abs(m) if m else 0))
    print("%d chains in %d chunks (%d new files)" % (n_chains, cm.shape[0], len(files)))


if __name__ == "__main__":
    main()
