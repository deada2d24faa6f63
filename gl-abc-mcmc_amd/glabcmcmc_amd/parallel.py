"""Multi-GPU: chains shard embarrassingly.

Rank r of W owns the global chain ids [r*C, (r+1)*C) -- ``chain0 = r*C`` in
``glabc_chains`` -- and because the Philox counter carries the GLOBAL chain id a W-GPU
run is, bit for bit, the concatenation of W single-GPU runs.  Nothing is exchanged while
sampling.  The only collective is one all-gather (RCCL over xGMI when the backend is
"nccl") of the per-chain streaming sums at checkpoint / ESJD time: per chain
d + 2*tri(d) float64 words (64 B for d = 2; 4 MiB per GPU at 65 536 chains) --
latency-bound, one call, no bucketing needed.

The one exception: GLMCMC_NF with ONE flow shared by all chains.  Its weights are replicated; at a training
step every rank differentiates forward_kld over ITS pool rows and `average_gradients` all-reduces the row-weighted mean
(0.55 MB for 8 couplings, once per pool refresh at most Train_step times) before the identical Adam update on every rank.
"""
import torch
import torch.distributed as dist

from . import engine


def shard_range(n_total, rank, world):
    """Contiguous chain-id range of `rank`: (chain0, n_local); remainders go to the low ranks."""
    base, rem = divmod(int(n_total), int(world))
    n_local = base + (1 if rank < rem else 0)
    chain0 = rank * base + min(rank, rem)
    return chain0, n_local


def gather_rows(local_rows, world, return_counts=False):
    """All-gather per-rank blocks [k][n_r] (chain-major) into [k][sum n_r], ranks in order, i.e. in global chain-id
    order.  Shards may be ragged (shard_range hands the remainder of n_total % world to the low ranks): the block
    widths are exchanged first (world int64 words), every block is padded to the widest, one all-gather moves the
    payload, the padding is cut away.  Works on any backend (gloo on CPU tensors in the tests, nccl = RCCL on device
    tensors in production)."""
    if world == 1:
        return (local_rows, [local_rows.shape[1]]) if return_counts else local_rows
    k, n = local_rows.shape
    mine = torch.tensor([n], dtype=torch.int64, device=local_rows.device)
    widths = torch.empty(world, dtype=torch.int64, device=local_rows.device)
    dist.all_gather_into_tensor(widths, mine)
    counts = [int(v) for v in widths.tolist()]
    n_max = max(counts)
    block = local_rows.contiguous()
    if n < n_max:
        block = torch.cat([block, block.new_zeros(k, n_max - n)], dim=1)
    out = torch.empty(world * k, n_max, dtype=local_rows.dtype, device=local_rows.device)
    dist.all_gather_into_tensor(out, block)
    out = out.view(world, k, n_max)
    if min(counts) == n_max:
        rows = out.permute(1, 0, 2).reshape(k, world * n_max).contiguous()
    else:
        rows = torch.cat([out[r, :, :counts[r]] for r in range(world)], dim=1).contiguous()
    return (rows, counts) if return_counts else rows


def gather_moments(mom, world, via=None):
    """engine.Moments of ALL chains of the job (every rank gets the same object).  `via`: device the
    collective payload is staged on (default: where the sums live; "cpu" for a gloo rehearsal)."""
    if world == 1:
        return mom
    d = mom.d
    tri = d * (d + 1) // 2
    packed = torch.cat([mom.sum_theta, mom.sum_outer, mom.sum_jump], dim=0)        # [d + 2 tri][n]
    home = packed.device
    if via is not None and torch.device(via) != home:
        allrows, counts = gather_rows(packed.to(via), world, return_counts=True)
        allrows = allrows.to(home)
    else:
        allrows, counts = gather_rows(packed, world, return_counts=True)
    out = engine.Moments.__new__(engine.Moments)
    out.n, out.d, out.steps = sum(counts), d, mom.steps
    out.sum_theta = allrows[:d].contiguous()
    out.sum_outer = allrows[d:d + tri].contiguous()
    out.sum_jump = allrows[d + tri:].contiguous()
    return out


def gather_chain_stats(mom, world, via=None):
    """Per-chain ESJD (ESJD.py:21-24, via glabc_moments_esjd) and pooled posterior moments
    of the whole job from the all-gathered streaming sums."""
    allm = gather_moments(mom, world, via)
    steps = float(allm.steps)
    return {
        "esjd": allm.esjd(),
        "mean": float((allm.sum_theta.sum(dim=1) / (steps * allm.n)).mean()),
        "mean_sq": float(torch.stack([allm.sum_outer[k].sum() for k in _diag(allm.d)]).mean() / (steps * allm.n)),
        "n_chains": allm.n,
    }


def _diag(d):
    """positions of the diagonal entries in the row-major upper triangle"""
    pos, k = [], 0
    for p in range(d):
        pos.append(k)
        k += d - p
    return pos


def average_gradients(tensors, n_local, group=None, via=None):
    """In place: every tensor (a mean over this rank's n_local rows -- gradient blobs, the loss) becomes the mean over the
    rows of ALL ranks, sum_r n_r t_r / sum_r n_r: one all-reduce of the concatenated payload.  `via`: device the payload is
    staged on ("cpu" for a gloo rehearsal of device tensors)."""
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    if world == 1:
        return tensors
    flat = torch.cat([t.reshape(-1).double() * float(n_local) for t in tensors] +
                     [torch.tensor([float(n_local)], dtype=torch.float64, device=tensors[0].device)])
    home = flat.device
    if via is not None and torch.device(via) != home:
        flat = flat.to(via)
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat = flat.to(home)
    total = flat[-1]
    at = 0
    for t in tensors:
        k = t.numel()
        t.copy_((flat[at:at + k] / total).to(t.dtype).view_as(t))
        at += k
    return tensors


def max_over_ranks(value, group=None, device=None):
    """the largest of the ranks' integers (one all-reduce of a single word; `device`: where the word lives -- the GPU for
    nccl = RCCL, "cpu" for gloo)"""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return int(value)
    t = torch.tensor([int(value)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return int(t.item())
