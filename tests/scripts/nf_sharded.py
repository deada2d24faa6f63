"""Two ranks (gloo) on one GPU, chains sharded, ONE flow shared: run under torch.distributed.run by tests/test_nf.py.
Every rank prints one JSON line: checksum of its final flow parameters, number of training steps, pooled E|theta|."""
import hashlib
import json
import os
import sys

import torch
import torch.distributed as dist


def main():
    kind = sys.argv[1]                       # fused | generic
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    import glabcmcmc_amd as g
    from glabcmcmc_amd.examples.Mixture import Mixture_set
    from glabcmcmc_amd.examples.UserModel import TorchMixture
    from glabcmcmc_amd.parallel import shard_range
    n_total = 1000
    chain0, n = shard_range(n_total, rank, world)
    model = Mixture_set(0.3) if kind == "fused" else TorchMixture(2, 0.3)
    lp = g.DiagGaussian(2, torch.zeros(1, 2), torch.log(torch.tensor([0.35, 0.35])))
    gen = torch.Generator().manual_seed(1)
    th_all = 1.3 * (torch.randint(0, 2, (n_total, 2), generator=gen).float() * 2 - 1)
    y_all = th_all.abs() + 0.2236 * torch.randn(n_total, 2, generator=gen)
    from glabcmcmc_amd.flows import RealNVP
    torch.manual_seed(5)
    flow = RealNVP(3)                          # the SAME initial flow on every rank (replicated weights)
    torch.manual_seed(100 + rank)              # ... but different generator states for the Models' own noise
    st = {}
    out = g.GLMCMC_NF(model, 81, th_all[chain0:chain0 + n], y_all[chain0:chain0 + n], lp, None, 0.7, 4, 5, None, 4, flow=flow,
                      seed=7, chain0=chain0, state_out=st, lr=5e-3, verbose=False, process_group=True)
    blob = st["flow"].packed_params().cpu().numpy().tobytes()
    line = json.dumps({"rank": rank, "n": n, "flow_sha": hashlib.sha256(blob).hexdigest(), "num_train": st["num_train"],
                      "pools": st["pools_drawn"], "mean_abs": float(out[40:].abs().mean()), "finite": bool(torch.isfinite(out).all()),
                       "loss": [round(float(v), 6) for v in st["loss_hist"]]})
    os.write(1, (line + "\n").encode())       # one write per rank: the two ranks share the pipe
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
