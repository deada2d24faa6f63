"""world_size-2 test of the multi-GPU plumbing on CPU (gloo): shard ranges, the
all-gather of per-chain sums in global chain-id order."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from glabcmcmc_amd import engine
        from glabcmcmc_amd.parallel import gather_moments, gather_rows, shard_range
        chain0, n = shard_range(n_total, rank, world)
        ids = torch.arange(chain0, chain0 + n, dtype=torch.float64)
        rows = torch.stack([ids, ids * 10 + 1, -ids])                       # [k=3][n], value encodes the chain id
        allrows = gather_rows(rows, world)
        mom = engine.Moments.__new__(engine.Moments)
        mom.n, mom.d, mom.steps = n, 2, 7
        mom.sum_theta = torch.stack([ids, ids + 0.5])
        mom.sum_outer = torch.stack([ids * 2, ids * 3, ids * 4])
        mom.sum_jump = torch.stack([ids * 5, ids * 6, ids * 7])
        allm = gather_moments(mom, world)
        q.put((rank, chain0, n, allrows, allm.n, allm.steps, allm.sum_theta, allm.sum_outer, allm.sum_jump))
    finally:
        dist.destroy_process_group()


import pytest


@pytest.mark.parametrize("world,n_total", [(2, 64), (2, 65), (3, 65), (3, 2)])
def test_gather_in_chain_id_order(world, n_total):
    """equal shards, ragged shards (65 chains over 2 and 3 ranks) and a rank with nothing to contribute"""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    ids = torch.arange(n_total, dtype=torch.float64)
    from glabcmcmc_amd.parallel import shard_range
    assert [(r[1], r[2]) for r in res] == [shard_range(n_total, r, world) for r in range(world)]
    for r in res:
        assert torch.equal(r[3], torch.stack([ids, ids * 10 + 1, -ids]))
        assert r[4] == n_total and r[5] == 7
        assert torch.equal(r[6], torch.stack([ids, ids + 0.5]))
        assert torch.equal(r[7], torch.stack([ids * 2, ids * 3, ids * 4]))
        assert torch.equal(r[8], torch.stack([ids * 5, ids * 6, ids * 7]))


def test_shard_range_covers_everything():
    from glabcmcmc_amd.parallel import shard_range
    for n_total in (1, 7, 64, 65536, 524288, 1000003):
        for world in (1, 2, 3, 4, 8):
            spans = [shard_range(n_total, r, world) for r in range(world)]
            assert spans[0][0] == 0
            for (a, n), (b, _) in zip(spans, spans[1:]):
                assert a + n == b
            assert spans[-1][0] + spans[-1][1] == n_total
            assert max(s[1] for s in spans) - min(s[1] for s in spans) <= 1


def _grad_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from glabcmcmc_amd.parallel import average_gradients, max_over_ranks
        n_local = [100, 300, 50][rank]
        g = torch.full((4, 5), float(rank + 1))                       # "a mean over this rank's rows"
        b = torch.tensor([10.0 * (rank + 1), -1.0])
        loss = torch.tensor([float(rank)])
        average_gradients([g, b, loss], n_local)
        q.put((rank, g, b, loss, max_over_ranks([3, 9, 4][rank], None, "cpu")))
    finally:
        dist.destroy_process_group()


def test_row_weighted_gradient_average_and_refresh_vote():
    """the shared flow of a sharded GLMCMC_NF: per-rank mean gradients become the mean over all ranks' rows (weights =
    rows per rank) on every rank; the pool-refresh decision is the maximum over the ranks"""
    world, port = 3, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_grad_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    w = torch.tensor([100.0, 300.0, 50.0])
    want_g = float((w * torch.tensor([1.0, 2.0, 3.0])).sum() / w.sum())
    want_b0 = float((w * torch.tensor([10.0, 20.0, 30.0])).sum() / w.sum())
    want_loss = float((w * torch.tensor([0.0, 1.0, 2.0])).sum() / w.sum())
    for r in res:
        assert torch.allclose(r[1], torch.full((4, 5), want_g)) and abs(float(r[2][0]) - want_b0) < 1e-5
        assert abs(float(r[2][1]) + 1.0) < 1e-6 and abs(float(r[3]) - want_loss) < 1e-6 and r[4] == 9


@pytest.mark.gpu
@pytest.mark.parametrize("workload,chains", [("glmcmc", 16384), ("gk", 8192)])
def test_bench_rehearsal_four_ranks_on_one_gpu_equals_one_rank(workload, chains):
    """The driver's multi-GPU run, rehearsed: `bench.py --gpus 4 --rehearse-gloo` -- four processes launched exactly as the
    driver launches them (torch.distributed.run, one rank per process), all on the one leased GPU, collectives on gloo -- against
    ONE rank that runs the same 4 x `chains` chains.  Chains are sharded by global chain id and the synthetic inputs are a
    function of it, so the pooled statistics (ESJD, moments) must be identical to the last bit; what the driver's 8-GPU run
    adds is only the backend string ('nccl') and one GPU per rank.  (Four ranks, not eight: the GPU box allows six processes on
    its card.)"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--workload", workload, "--steps", "2", "--warmup", "1", "--iters", "150", "--no-cpu-baseline"]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    one = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--chains", str(4 * chains)] + common,
                         env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert one.returncode == 0, one.stderr[-2000:]
    four = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr",
                           "127.0.0.1", "--master-port", "29547", os.path.join(root, "bench.py"), "--gpus", "4", "--rehearse-gloo",
                           "--chains", str(chains)] + common, env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert four.returncode == 0, four.stderr[-2000:]
    a = json.loads([l for l in one.stdout.strip().split("\n") if l.startswith("{")][-1])
    b = json.loads([l for l in four.stdout.strip().split("\n") if l.startswith("{")][-1])
    assert (a["n_gpus"], b["n_gpus"]) == (1, 4) and b["scaling"] == "weak"
    assert a["config"]["chains_per_gpu"] == 4 * b["config"]["chains_per_gpu"]
    for key in ("esjd_mean", "mean_theta", "mean_theta_sq", "moment_iters", "esjd_nan_frac"):
        assert a[key] == b[key], (key, a[key], b[key])
    assert a["esjd_mean"] > 0
