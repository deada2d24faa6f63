"""The sampler loops for Models that are plain Python objects (the reference's duck-typed Model protocol).

The reference calls ``ABCset.generate_samples / prior_log_prob / calculate_log_kernel`` from inside its loops
(GLMCMC.py:71-74,94-97; GlobalMCMC.py:41-46,57-61; protocol: examples/Mixture.py:5-53).  A Model that exposes
``descriptor()`` is compiled into the fused gfx950 kernels; every other Model -- a user's own simulator -- runs here:
one iteration for ALL chains is

    glabc_propose            HIP: every random draw of the iteration (the fused kernels' Philox slots), candidates
    Model callbacks          the user's code on one (batch_size * n_chains, dim) batch -- CUDA tensors, or CPU tensors
                             for Models written against CPU tensors (``callback_device``)
    [glabc_propose_redraw]   HIP: the prior-sentinel redraw loop of GLMCMC.py:92-93
    glabc_select             HIP: iSIR weights / torch.sum order / double-precision index or the MH test, state update,
                             Theta_Re row, ESJD and moment sums

so the decisions are taken by the same arithmetic as in the fused kernels, and the Model is called
``batch_size * n_chains`` rows at a time instead of ``batch_size`` rows.  Proposal objects without a ``glabc_dist``
descriptor (``Gamma``, ``GaussianMixture``, a user's class) are callbacks too: ``forward`` / ``sample`` / ``log_prob``.

Optional extension of the protocol: a Model with ``noise_dim`` and ``simulate_from_noise(theta, eps)`` receives the
simulator's standard normals from the run's Philox stream (reproducible from ``seed``, independent of sharding); without
it ``generate_samples(theta, 1)`` draws from whatever generator the Model uses.
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _capi, _host, engine

SENTINEL = float(np.float32(7 * math.log(1e-10)))           # GLMCMC.py:92


def try_descriptor(obj):
    """obj.descriptor() or None when the object cannot describe itself as a glabc_dist / glabc_model"""
    fn = getattr(obj, "descriptor", None)
    if fn is None:
        return None
    try:
        return fn()
    except (NotImplementedError, ValueError):
        return None


def dist_descriptor(obj, dim, gamma=False):
    """the object's glabc_dist if it has one that a sampler can draw from (DiagGaussian / Uniform of the right dimension; a
    Gamma only where the caller says the kernels know it -- `gamma`: the importance / global proposal of glabc_propose)"""
    d = try_descriptor(obj)
    if d is None or not isinstance(d, _capi.Dist):
        return None                                          # e.g. GaussianMixture, a user's class
    if d.kind == _capi.DIST_GAMMA and not gamma:
        return None                                          # a callback there: forward() / log_prob() on the device
    if d.dim != dim:
        raise ValueError("proposal dimension %d does not match Model.theta_dim = %d" % (d.dim, dim))
    return d


def fused_supported(ABCset, proposals, batch_size, max_batch=None, max_dim=8, gamma_ok=False):
    """Can the fused kernels (glabc_glmcmc_steps / glabc_globalmcmc_steps / glabc_glmala_steps) run this configuration?
    gamma_ok: the entry point knows GLABC_DIST_GAMMA as the LAST proposal (importance / global) and as the Model's prior --
    GLMCMC and GlobalMCMC on the |theta| + noise Model up to theta_dim 4 (include/glabc.h)."""
    m = try_descriptor(ABCset)
    if m is None or not isinstance(m, _capi.Model):
        return False
    gamma = m.prior.kind == _capi.DIST_GAMMA
    for i, p in enumerate(proposals):
        d = try_descriptor(p)
        if d is None or not isinstance(d, _capi.Dist) or d.dim != m.theta_dim:
            return False
        if d.kind == _capi.DIST_GAMMA:
            if i != len(proposals) - 1:
                return False                                 # a Gamma local increment: callback
            gamma = True
    if gamma and not (gamma_ok and m.sim_kind == _capi.SIM_ABS_GAUSS and m.theta_dim <= 4):
        return False
    if m.sim_kind == _capi.SIM_USER:                         # compiled.CompiledModel: register kernels only, compiled per batch size
        return hasattr(ABCset, "program") and (batch_size is None or 1 <= int(batch_size) <= _capi.MAX_BATCH)
    if m.sim_kind == _capi.SIM_ABS_GAUSS:                    # instantiated for theta_dim 1..8 (GLMALA and batch sizes > 16: 1..4)
        if not 1 <= m.theta_dim <= max_dim:
            return False
        if m.theta_dim > 4 and batch_size is not None and int(batch_size) > _capi.MAX_BATCH:
            return False
    return batch_size is None or 1 <= int(batch_size) <= (max_batch or _capi.MAX_BATCH)


class ModelCallbacks:
    """Evaluates a duck-typed Model on candidate batches, on the device its code can work with."""

    def __init__(self, abc_set, device, callback_device="auto"):
        self.m = abc_set
        self.device = device
        if callback_device not in ("auto", "cuda", "cpu"):
            raise ValueError("callback_device must be 'auto', 'cuda' or 'cpu'")
        self.auto = callback_device == "auto"
        self.where = None if self.auto else callback_device
        self.noise_dim = int(getattr(abc_set, "noise_dim", 0)) if hasattr(abc_set, "simulate_from_noise") else 0

    def _back(self, t, rows):
        t = torch.as_tensor(t)
        return t.detach().to(device=self.device, dtype=torch.float32).reshape(rows, -1).contiguous()

    def _call(self, fn):
        """fn(on_cuda: bool)"""
        return fn(self.where != "cpu")

    def probe(self, theta_rows, y_rows):
        """'auto': decide once, before the loop, where the callbacks run.  A Model written against CPU tensors (the
        reference's own examples/Mixture.py mixes its CPU constants into the arithmetic) raises on CUDA tensors; it then
        gets CPU copies of every batch.  All three callbacks are tried on a few rows; the decision is final."""
        if not self.auto:
            return
        try:
            self.where = "cuda"
            th = theta_rows[:2].contiguous()
            self.prior(th)
            noise = torch.zeros(th.shape[0], self.noise_dim, device=th.device) if self.noise_dim else None
            self.kernel(self.simulate(th, noise))
            self.kernel(y_rows[:2].contiguous())
        except (RuntimeError, TypeError, ValueError):
            self.where = "cpu"

    def prior(self, theta):
        rows = theta.shape[0]
        return self._call(lambda cuda: self._back(self.m.prior_log_prob(theta if cuda else theta.cpu()), rows)).view(-1)

    def kernel(self, y):
        rows = y.shape[0]
        return self._call(lambda cuda: self._back(self.m.calculate_log_kernel(y if cuda else y.cpu()), rows)).view(-1)

    def simulate(self, theta, noise):
        rows = theta.shape[0]
        if self.noise_dim:
            return self._call(lambda cuda: self._back(
                self.m.simulate_from_noise(theta if cuda else theta.cpu(), noise if cuda else noise.cpu()), rows))
        return self._call(lambda cuda: self._back(self.m.generate_samples(theta if cuda else theta.cpu(), 1), rows))


class ProposalCallbacks:
    """A proposal object without a glabc_dist descriptor: forward / sample / log_prob as callbacks."""

    def __init__(self, dist, device):
        self.dist, self.device = dist, device

    def _dev(self, t, rows):
        return torch.as_tensor(t).detach().to(device=self.device, dtype=torch.float32).reshape(rows, -1).contiguous()

    def _draw(self, n):
        try:
            return self.dist.forward(n, device=self.device)        # the build's Gamma draws on the device (glabc_gamma_forward)
        except TypeError:
            return self.dist.forward(n)                            # anyone else's distribution: its own generator

    def forward(self, n):
        z, lp = self._draw(n)
        return self._dev(z, n), self._dev(lp, n).view(-1)

    def sample(self, n):
        return self._dev(self._draw(n)[0], n)

    def log_prob(self, theta):
        try:
            out = self.dist.log_prob(theta)
        except (RuntimeError, TypeError, ValueError):
            out = self.dist.log_prob(theta.cpu())
        return self._dev(out, theta.shape[0]).view(-1)


def run(algo, ABCset, num_ite, Initial_theta, Initial_y, Local_Proposal, Global_Proposal, filelocation, global_frequency,
        batch_size, csv_variant, *, seed=None, device=None, chain0=0, record_history=True, stats=None, return_device=False,
        verbose=True, state_out=None, callback_device="auto", sentinel_redraw=True, max_redraws=100000, max_graph_rounds=32,
        progress=None, graph="auto"):
    """GLMCMC (algo = _capi.ALGO_GLMCMC, GLMCMC.py:24-137) or GlobalMCMC (_capi.ALGO_GLOBALMCMC, GlobalMCMC.py:6-98) with
    the Model -- and, if need be, the proposals -- as callbacks.  Same return value and side effects as the fused path.

    graph: 'auto' | True | False.  An iteration whose work never visits the host -- descriptor proposals, callbacks on CUDA
    tensors, no sentinel check (GlobalMCMC, or sentinel_redraw=False) -- is captured ONCE as a hipGraph (torch.cuda.graph:
    the two HIP kernels with the iteration index in device memory, glabc_run.step0_device, plus the Model's own kernels) and
    replayed; 'auto' falls back to launching eagerly when the capture is not possible (e.g. a callback that synchronises).
    With the sentinel check on (GLMCMC's default) the replay is speculative, see below: same results as the eager loop; a prior
    that does return the sentinel keeps the run a replayed graph, with up to max_graph_rounds redraw rounds inside it."""
    lib = _capi.lib()
    dev, chains, single = _host.prepare(ABCset, Initial_theta, Initial_y, device, chain0)
    n, d, yd = chains.n, chains.d, chains.yd
    N = int(batch_size) if algo == _capi.ALGO_GLMCMC else 1
    if N < 1:
        raise ValueError("batch_size must be >= 1")
    key = engine.draw_seed(seed)
    model = ModelCallbacks(ABCset, dev, callback_device)
    local_desc = dist_descriptor(Local_Proposal, d) if Local_Proposal is not None else None
    global_desc = dist_descriptor(Global_Proposal, d, gamma=True)
    local_cb = ProposalCallbacks(Local_Proposal, dev) if (local_desc is None and Local_Proposal is not None) else None
    global_cb = ProposalCallbacks(Global_Proposal, dev) if global_desc is None else None
    if Local_Proposal is None and float(global_frequency) < 1:
        raise ValueError("a local proposal is needed unless global_frequency >= 1")

    R = N * n
    f32 = dict(dtype=torch.float32, device=dev)
    theta_prop = torch.zeros(R, d, **f32)
    log_q = torch.zeros(R, **f32)
    nd = model.noise_dim
    sim_noise = torch.zeros(R, nd, **f32) if nd else None
    log_u = torch.zeros(n, **f32)
    u_res = torch.zeros(n, dtype=torch.float64, device=dev)
    is_global = torch.zeros(n, dtype=torch.int32, device=dev)
    n_redrawn = torch.zeros(1, dtype=torch.int32, device=dev)
    model.probe(chains.theta.t(), chains.y.t())
    # callbacks of the initial state (GLMCMC.py:52-55): carried from here on by glabc_select
    prior_cur = model.prior(chains.theta.t().contiguous()).clone()
    kern_cur = model.kernel(chains.y.t().contiguous()).clone()
    hist = _host.allocate_history(num_ite, chains, record_history)

    io = _capi.StepIO()
    io.n_prop, io.theta_dim, io.y_dim, io.noise_dim = N, d, yd, nd
    io.theta_prop, io.log_q = theta_prop.data_ptr(), log_q.data_ptr()
    io.sim_noise = sim_noise.data_ptr() if nd else None
    io.log_u, io.u_res, io.is_global = log_u.data_ptr(), u_res.data_ptr(), is_global.data_ptr()
    io.prior_cur, io.kern_cur = prior_cur.data_ptr(), kern_cur.data_ptr()
    cs = chains.struct()
    ms = stats.struct() if stats is not None else None
    run_ = _capi.Run()
    run_.seed, run_.n_steps, run_.global_frequency, run_.batch_size, run_.hist_stride = key, 1, float(global_frequency), N, n
    if ms is not None:
        run_.moments = C.pointer(ms)
    hist_ptr, hist_row_bytes = (hist.data_ptr(), hist[0].numel() * 4) if hist is not None else (0, 0)
    lp = C.byref(local_desc) if local_desc is not None else None
    gp = C.byref(global_desc) if global_desc is not None else None

    col_index = torch.arange(n, device=dev).view(1, n)
    n_valid = torch.full((n,), N, dtype=torch.int32, device=dev)
    if global_cb is not None and algo == _capi.ALGO_GLMCMC:
        io.n_valid = n_valid.data_ptr()
    keep = {}                                              # tensors whose addresses the current StepIO holds
    speculating = [False]                                  # inside a captured iteration: bounded redraw rounds + a count
    graph_rounds = [0]                                     # redraw rounds (GLMCMC.py:92-93) held by the captured iteration

    def iteration(i):
        """one iteration on torch's current stream; i = None: the index is read from step_t on the device (graph)"""
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        if i is not None:
            run_.step0 = i
            run_.history = hist_ptr + i * hist_row_bytes if hist is not None else None
        _capi.check(lib.glabc_propose(algo, lp, gp, C.byref(cs), C.byref(run_), C.byref(io), stream), "glabc_propose")
        if global_cb is not None or local_cb is not None:
            glob_rows = (is_global != 0)
            if global_cb is not None:                                  # Importance_Proposal.forward(batch_size), GLMCMC.py:66
                z, lq = global_cb.forward(R)
                if local_cb is None:                                   # keep the kernel's local-move rows
                    keep_rows = torch.zeros(R, dtype=torch.bool, device=dev)
                    keep_rows[:n] = ~glob_rows
                    z = torch.where(keep_rows.view(-1, 1), theta_prop, z)
                theta_prop.copy_(z)
                log_q.copy_(lq)
                if algo == _capi.ALGO_GLMCMC:
                    # GLMCMC.py:67-70: proposals with a NaN coordinate are dropped BEFORE the Model sees them; the weight vector
                    # is then shorter.  Only a callback proposal can produce them.  Stays on the device: the chain's valid rows
                    # move to the front in their order (the k-th survivor is simulated with the k-th noise row, as the
                    # reference's generate_samples(Theta_prop0) draws for the shortened batch), glabc_select gets the count.
                    bad = torch.isnan(theta_prop).any(1).view(N, n) & glob_rows.view(1, n)
                    order = torch.argsort(bad.to(torch.uint8), dim=0, stable=True)
                    rows = (order * n + col_index).view(-1)
                    theta_prop.copy_(theta_prop[rows])
                    log_q.copy_(log_q[rows])
                    n_valid.copy_((N - bad.sum(0)).to(torch.int32))
            if local_cb is not None:                                   # Local_Proposal.sample(1) + Theta_old, GLMCMC.py:91
                row0_local = local_cb.sample(n) + chains.theta.t()
                theta_prop[:n] = torch.where(glob_rows.view(-1, 1), theta_prop[:n], row0_local)
        prior_prop = model.prior(theta_prop)                           # GLMCMC.py:74,92,96
        io.prior_prop = prior_prop.data_ptr()
        if sentinel_redraw and algo == _capi.ALGO_GLMCMC and speculating[0]:
            # captured form: the reference's loop (GLMCMC.py:92-93) ends on a host read, which a hipGraph cannot hold, so the
            # capture holds a BOUNDED number of its rounds -- redraw the local candidates whose prior is the sentinel (a
            # round without one is a no-op: the kernel touches nothing and the prior of an unchanged row is the same number),
            # evaluate the prior again -- and one more launch of the redraw kernel that only COUNTS the candidates still at the
            # sentinel after them (its device counter is never zeroed here).  The replay loop below reads the count once per
            # segment; if it is ever non-zero it restores the segment's start state and captures again with more rounds.
            for rnd in range(1, graph_rounds[0] + 1):
                _capi.check(lib.glabc_propose_redraw(lp, C.byref(cs), C.byref(run_), C.byref(io), rnd, n_redrawn.data_ptr(),
                                                     stream), "glabc_propose_redraw")
                prior_prop[:n] = model.prior(theta_prop[:n])
            _capi.check(lib.glabc_propose_redraw(lp, C.byref(cs), C.byref(run_), C.byref(io), graph_rounds[0] + 1,
                                                 violations.data_ptr(), stream), "glabc_propose_redraw")
        elif sentinel_redraw and algo == _capi.ALGO_GLMCMC:            # GLMCMC.py:92-93
            for rnd in range(1, max_redraws + 1):
                if local_cb is None:
                    n_redrawn.zero_()
                    _capi.check(lib.glabc_propose_redraw(lp, C.byref(cs), C.byref(run_), C.byref(io), rnd,
                                                         n_redrawn.data_ptr(), stream), "glabc_propose_redraw")
                    if int(n_redrawn.item()) == 0:
                        break
                    prior_prop[:n] = model.prior(theta_prop[:n])
                else:
                    again = (is_global == 0) & (prior_prop[:n] == SENTINEL)
                    k = int(again.sum().item())
                    if k == 0:
                        break
                    theta_prop[:n][again] = local_cb.sample(k) + chains.theta.t()[again]
                    prior_prop[:n] = model.prior(theta_prop[:n])
            else:
                raise RuntimeError("the local proposal keeps landing where prior_log_prob returns the sentinel "
                                   "7*log(1e-10) (GLMCMC.py:92-93) after %d redraws" % max_redraws)
        y_prop = model.simulate(theta_prop, sim_noise)                 # GLMCMC.py:71,94
        if y_prop.shape[1] != yd:
            raise ValueError("generate_samples returned %d columns, Initial_y has %d" % (y_prop.shape[1], yd))
        kern_prop = model.kernel(y_prop)                               # GLMCMC.py:72,96
        io.y_prop, io.kern_prop = y_prop.data_ptr(), kern_prop.data_ptr()
        keep.update(prior=prior_prop, y=y_prop, kern=kern_prop)
        if global_cb is not None:                                      # Importance_Proposal.log_prob(Theta_old), GLMCMC.py:63
            keep["q"] = global_cb.log_prob(chains.theta.t().contiguous())
            io.q_cur = keep["q"].data_ptr()
        _capi.check(lib.glabc_select(algo, gp, C.byref(cs), C.byref(run_), C.byref(io), stream), "glabc_select")

    # Graph replay.  Without a sentinel check (GlobalMCMC, sentinel_redraw=False) an iteration never visits the host.  WITH it
    # (the default for GLMCMC) the replay is speculative: almost no prior ever returns the sentinel, so the first capture holds
    # no redraw round at all and only counts the local candidates that hit it; the count is read once per segment of 64
    # iterations.  When it is non-zero the segment's start state is restored and the iteration is captured AGAIN with 2, 4, ...
    # max_graph_rounds redraw rounds inside the graph (device side, no host read): a Model whose prior does return the sentinel
    # keeps running as a replayed graph.  Only a prior that still returns it after max_graph_rounds redraws of one candidate sends
    # the rest of the run through the eager loop.  The Philox draws depend on (chain, iteration, round) only, so every form gives
    # the eager path's chains.
    plain = local_cb is None and global_cb is None and progress is None
    speculative = plain and sentinel_redraw and algo == _capi.ALGO_GLMCMC
    capturable = plain
    if graph is True and not capturable:
        raise ValueError("graph=True needs descriptor proposals and no progress callback")
    use_graph = capturable and graph in ("auto", True) and num_ite > 8
    violations = torch.zeros(1, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        i = 1
        if use_graph:
            # three eager iterations (they also settle where the callbacks run), then one captured iteration replayed
            for _ in range(3):
                iteration(i)
                i += 1
            use_graph = model.where != "cpu"
        if use_graph:
            saved = [t.clone() for t in (chains.theta, chains.y, chains.log_w, chains.flags, chains.n_moves, prior_cur, kern_cur)]
            live = [chains.theta, chains.y, chains.log_w, chains.flags, chains.n_moves, prior_cur, kern_cur]
            if stats is not None:
                live += [stats.sum_theta, stats.sum_outer, stats.sum_jump]
                saved += [t.clone() for t in (stats.sum_theta, stats.sum_outer, stats.sum_jump)]

            def snapshot():
                for a_, b_ in zip(saved, live):
                    a_.copy_(b_)

            def restore():
                for a_, b_ in zip(saved, live):
                    b_.copy_(a_)

            step_t = torch.tensor([i], dtype=torch.int32, device=dev)          # the iteration index, on the device
            side = torch.cuda.Stream(dev)

            def capture():
                """one iteration (graph_rounds[0] redraw rounds inside) as a hipGraph reading its index from step_t; the
                warm-up run of the captured form IS iteration i of the chains.  None when it cannot be captured."""
                step_t.fill_(i)
                violations.zero_()
                run_.step0, run_.step0_device = 1, step_t.data_ptr()
                run_.history = hist_ptr + hist_row_bytes if hist is not None else None   # row 0 of `history` = iteration 1
                g_ = torch.cuda.CUDAGraph()
                try:
                    side.wait_stream(torch.cuda.current_stream(dev))
                    with torch.cuda.stream(side):                               # warm the captured form once (step_t advances)
                        iteration(None)
                        step_t.add_(1)
                    torch.cuda.current_stream(dev).wait_stream(side)
                    with torch.cuda.graph(g_):
                        iteration(None)
                        step_t.add_(1)
                except Exception:                                               # not capturable after all: launch eagerly
                    if graph is True:
                        raise
                    torch.cuda.synchronize(dev)
                    g_ = None
                finally:
                    run_.step0_device = None
                return g_

            speculating[0] = speculative
            snapshot()
            g = capture()
            done = 1 if g is not None else 0                                    # the warm-up iteration belongs to the first segment
            if g is None:                                                       # (a failed capture may have run part of an iteration)
                restore()
            while g is not None and i + done <= num_ite:
                k = min(64 - done, num_ite - i - done)
                for _ in range(k):
                    g.replay()
                done += k
                if speculative and int(violations.item()) != 0:                 # one synchronisation per segment
                    restore()                                                   # back to iteration i
                    if state_out is not None:
                        state_out.setdefault("graph_rolled_back_at", i)
                    graph_rounds[0] = max(2, 2 * graph_rounds[0])
                    if graph_rounds[0] > max_graph_rounds:
                        g = None
                        break
                    g = capture()
                    done = 1 if g is not None else 0
                    if g is None:
                        restore()
                    continue
                i += done
                done = 0
                if i >= num_ite:
                    break
                if speculative:
                    snapshot()
            speculating[0] = False
            if g is not None:
                i = num_ite
                if state_out is not None:
                    state_out["graph"] = True
                    state_out["graph_redraw_rounds"] = graph_rounds[0]
        for i in range(i, num_ite):
            iteration(i)
            if progress is not None:
                progress(i)
    if stats is not None:
        stats.steps += num_ite - 1
    if state_out is not None:
        state_out.update(chains=chains, prior_cur=prior_cur, kern_cur=kern_cur, callback_device=model.where)
    return _host.finish(hist, chains, single, filelocation, csv_variant, verbose and single, return_device)


# ------------------------------------------------------------------------------------------- pool samplers (GLMCMC_NF, AGLMCMC)
def _noise_seed(key, chain0):
    """Seed of the torch generator that draws a callback Model's simulator noise (pool rows, MALA gradient estimates).  Every
    rank of a sharded run is given the same `seed` and its own chain0 (parallel.shard_range); the generator's stream must
    differ between shards, or chains on different GPUs would share their simulator noise -- so chain0 is hashed in
    (splitmix64 finaliser).  One shard (chain0 = 0) keeps the seed's own stream."""
    z = (int(chain0) * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    z ^= z >> 31
    return (int(key) ^ z) & 0x7FFFFFFFFFFFFFFF


class PoolSampler:
    """One chain-state + iteration engine shared by the callback forms of GLMCMC_NF and AGLMCMC: every chain owns a pool of
    P = batch_size*step_size proposals (row r = p*n + c), a global move is the iSIR step against the chain's next slice
    (GLMCMC_NFs.py:90-111 / AGLMCMC.py:125-172), a local move the random-walk MH step (:142-152 / :251-272).  The Model's
    methods evaluate a pool once (`load_pool`) and the local-move candidates every iteration; glabc_propose draws, glabc_select
    decides.  `log_q_old` (the proposal's log-density of the current states) is the caller's to keep current."""

    def __init__(self, ABCset, num_ite, Initial_theta, Initial_y, Local_Proposal, global_frequency, step_size, batch_size,
                 seed, device, chain0, callback_device):
        self.lib = _capi.lib()
        self.dev, self.chains, self.single = _host.prepare(ABCset, Initial_theta, Initial_y, device, chain0)
        dev, chains = self.dev, self.chains
        self.n, self.d, self.yd = chains.n, chains.d, chains.yd
        n, d = self.n, self.d
        self.N, self.S = int(batch_size), int(step_size)
        self.rows = self.N * self.S * n
        R = self.N * n
        self.key = engine.draw_seed(seed)
        self.model = ModelCallbacks(ABCset, dev, callback_device)
        self.local_desc = dist_descriptor(Local_Proposal, d)
        self.local_cb = ProposalCallbacks(Local_Proposal, dev) if self.local_desc is None else None
        self.gen = torch.Generator(device=dev)
        self.gen.manual_seed(_noise_seed(self.key, chain0))
        f32 = dict(dtype=torch.float32, device=dev)
        self.theta_prop = torch.zeros(R, d, **f32)
        self.log_q = torch.zeros(R, **f32)
        self.nd = self.model.noise_dim
        self.sim_noise = torch.zeros(R, self.nd, **f32) if self.nd else None
        self.log_u = torch.zeros(n, **f32)
        self.u_res = torch.zeros(n, dtype=torch.float64, device=dev)
        self.is_global = torch.zeros(n, dtype=torch.int32, device=dev)
        self.model.probe(chains.theta.t(), chains.y.t())
        self.prior_cur = self.model.prior(chains.theta.t().contiguous()).clone()
        self.kern_cur = self.model.kernel(chains.y.t().contiguous()).clone()
        self.hist = _host.allocate_history(num_ite, chains, True)
        self.kk = torch.zeros(n, dtype=torch.int64, device=dev)
        self.log_q_old = torch.zeros(n, **f32)
        self._chain_ids = torch.arange(n, device=dev).view(1, n)
        self._slot = torch.arange(self.N, device=dev).view(self.N, 1)
        self.pool = {}
        io = self.io = _capi.StepIO()
        io.n_prop, io.theta_dim, io.y_dim, io.noise_dim = self.N, d, self.yd, self.nd
        io.theta_prop, io.log_q = self.theta_prop.data_ptr(), self.log_q.data_ptr()
        io.sim_noise = self.sim_noise.data_ptr() if self.nd else None
        io.log_u, io.u_res, io.is_global = self.log_u.data_ptr(), self.u_res.data_ptr(), self.is_global.data_ptr()
        io.prior_cur, io.kern_cur = self.prior_cur.data_ptr(), self.kern_cur.data_ptr()
        self.cs = chains.struct()
        run_ = self.run_ = _capi.Run()
        run_.seed, run_.n_steps, run_.global_frequency, run_.batch_size, run_.hist_stride = \
            self.key, 1, float(global_frequency), self.N, n
        self._hist_ptr, self._hist_row_bytes = self.hist.data_ptr(), self.hist[0].numel() * 4
        self.stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        self._countdown = self.S
        self.group = None

    def load_pool(self, theta_rows, lq):
        """theta_rows (rows, d), lq (rows,): simulate and evaluate the pool (GLMCMC_NFs.py:73-78 / AGLMCMC.py:90-100)"""
        eps = torch.randn(self.rows, self.nd, generator=self.gen, device=self.dev) if self.nd else None
        x = self.model.simulate(theta_rows, eps)
        self.pool = dict(theta=theta_rows, lq=lq, x=x, prior=self.model.prior(theta_rows), kern=self.model.kernel(x))
        self.kk.zero_()
        self._countdown = self.S
        return self.pool

    def pool_weights(self):
        """weight0 = exp(prior + kernel - log q), NaN -> 0 (GLMCMC_NFs.py:79-85)"""
        w = torch.exp(self.pool["prior"] + self.pool["kern"] - self.pool["lq"])
        return torch.where(torch.isnan(w), torch.zeros_like(w), w)

    def step(self, i):
        """iteration i for every chain; returns the mask of the chains that moved (on the device)"""
        lib, n, N, io, run_, pool, chains = self.lib, self.n, self.N, self.io, self.run_, self.pool, self.chains
        run_.step0, run_.history = i, self._hist_ptr + i * self._hist_row_bytes
        lp = C.byref(self.local_desc) if self.local_desc is not None else None
        _capi.check(lib.glabc_propose(_capi.ALGO_GLMCMC, lp, None, C.byref(self.cs), C.byref(run_), C.byref(io), self.stream),
                    "glabc_propose")                                                            # branch, uniforms, local candidates
        glob = (self.is_global & 1) != 0
        if self.local_cb is not None:
            self.theta_prop[:n] = self.local_cb.sample(n) + chains.theta.t()
        src = (((self.kk.clamp(max=self.S - 1) * N).view(1, n) + self._slot) * n + self._chain_ids).view(-1)   # next slices
        loc = ~glob
        th_loc = torch.where(loc.view(-1, 1), self.theta_prop[:n], pool["theta"][src[:n]])       # local move: row 0 of the chain
        self.theta_prop.copy_(pool["theta"][src])
        self.theta_prop[:n] = th_loc
        y_loc = self.model.simulate(th_loc, self.sim_noise[:n].contiguous() if self.nd else None)
        y_prop, prior_prop, kern_prop = pool["x"][src], pool["prior"][src], pool["kern"][src]
        y_prop[:n] = torch.where(loc.view(-1, 1), y_loc, y_prop[:n])
        prior_prop[:n] = torch.where(loc, self.model.prior(th_loc), prior_prop[:n])
        kern_prop[:n] = torch.where(loc, self.model.kernel(y_loc), kern_prop[:n])
        self.log_q.copy_(pool["lq"][src])
        chains.flags.fill_(_capi.FLAG_LOCAL)        # the current state's weight is recomputed at every global move
        io.y_prop, io.prior_prop, io.kern_prop = y_prop.data_ptr(), prior_prop.data_ptr(), kern_prop.data_ptr()
        io.q_cur = self.log_q_old.data_ptr()
        _capi.check(lib.glabc_select(_capi.ALGO_GLMCMC, None, C.byref(self.cs), C.byref(run_), C.byref(io), self.stream),
                    "glabc_select")
        self.kk += glob
        self._countdown -= 1
        return (self.is_global & 2) != 0

    def pool_used_up(self):
        """has some chain used its step_size slices?  (a chain uses at most one per iteration: the device is asked -- a
        synchronisation -- only when that has become possible)"""
        if self._countdown > 0:
            return False
        used = int(self.kk.max().item())
        if self.group is not None:                 # ranks sharing one proposal refresh together: the fullest pool decides
            from .parallel import max_over_ranks
            g = None if self.group is True else self.group
            used = max_over_ranks(used, g, self.dev if torch.distributed.get_backend(g) == "nccl" else "cpu")
        if used < self.S:
            self._countdown = self.S - used
            return False
        return True

    def finish(self, filelocation, csv_variant, verbose, return_device):
        return _host.finish(self.hist, self.chains, self.single, filelocation, csv_variant, verbose and self.single, return_device)


def run_glmcmc_nf(ABCset, num_ite, Initial_theta, Initial_y, Local_Proposal, filelocation, global_frequency, step_size,
                  batch_size, base, Train_step, *, num_layers=32, seed=None, device=None, chain0=0, return_device=False,
                  verbose=True, flow=None, lr=5e-4, weight_decay=1e-5, state_out=None, callback_device="auto", process_group=None):
    """GLMCMC_NF (GLMCMC_NFs.py:43-186) with the Model as callbacks.  The flow's kernels are the fused path's
    (glabc_nf_sample for the pools, glabc_nf_log_prob_indexed for NF_model.log_prob(Theta_old), HipAdam for the training
    step); the Model is evaluated through its own methods: once per pool on all of its rows (generate_samples, prior_log_prob,
    calculate_log_kernel: GLMCMC_NFs.py:73-85,128-140) and once per iteration on the local-move candidates (:142-146); the
    iSIR index / MH test / state update / Theta_Re row are glabc_propose + glabc_select (PoolSampler).  Same schedule as the
    fused path: every chain owns a pool, all pools are redrawn as soon as one chain has used its step_size slices, one flow."""
    from .flows import HipAdam, RealNVP
    from .GLMCMC_NFs import resample
    ps = PoolSampler(ABCset, num_ite, Initial_theta, Initial_y, Local_Proposal, global_frequency, step_size, batch_size, seed,
                     device, chain0, callback_device)
    if ps.d != 2:
        raise ValueError("the RealNVP of GLMCMC_NF is built for theta_dim = 2 (MLP([1,128,128,2]), GLMCMC_NFs.py:56)")
    lib, dev, n, rows, chains = ps.lib, ps.dev, ps.n, ps.rows, ps.chains
    if flow is None:
        flow = RealNVP(num_layers, base if isinstance(base, torch.nn.Module) else None)
    flow = flow.to(dev)
    optimizer = HipAdam(flow, lr=lr, weight_decay=weight_decay)                                  # GLMCMC_NFs.py:63
    ps.group = process_group
    via = None if process_group is None or torch.distributed.get_backend(None if process_group is True else process_group) == "nccl" \
        else "cpu"
    count = torch.zeros(1, dtype=torch.int32, device=dev)
    losses = []

    def draw_pool(refresh_id):
        flow.eval()
        z, lq = flow.sample(rows, seed=ps.key ^ 0x9E3779B97F4A7C15,
                            row0=(refresh_id << 44) + chains.chain0 * ps.N * ps.S)                # :70-72 / 125-127; per shard
        ps.load_pool(z.contiguous(), lq)

    def flow_state():
        blob = flow.packed_params()
        return blob, flow.descriptor(blob)

    def log_prob_all():
        with torch.cuda.device(dev):
            _capi.check(lib.glabc_nf_log_prob(C.byref(fdesc), chains.theta.data_ptr(), n, ps.log_q_old.data_ptr(), ps.stream),
                        "glabc_nf_log_prob")

    refresh, num_train = 0, 0
    draw_pool(refresh)
    blob, fdesc = flow_state()
    log_prob_all()
    with torch.cuda.device(dev):
        for i in range(1, num_ite):
            moved = ps.step(i)
            order = torch.argsort(moved, descending=True, stable=True).to(torch.int32)            # NF_model.log_prob(Theta_old), :96-98:
            count.copy_(moved.sum().to(torch.int32).view(1))                                      # only where the state changed
            _capi.check(lib.glabc_nf_log_prob_indexed(C.byref(fdesc), chains.theta.data_ptr(), n, order.data_ptr(),
                                                      count.data_ptr(), n, ps.log_q_old.data_ptr(), ps.stream),
                        "glabc_nf_log_prob_indexed")
            if not ps.pool_used_up():                                                             # :112
                continue
            if num_train < Train_step:                                                            # :114-124
                w = ps.pool_weights()
                idx = resample(w / torch.sum(w), rows)
                losses.append(optimizer.step(ps.pool["theta"][idx], group=process_group, via=via))
                num_train += 1
                blob, fdesc = flow_state()
                log_prob_all()
            refresh += 1
            draw_pool(refresh)
    if state_out is not None:
        state_out.update(chains=chains, flow=flow, loss_hist=losses, num_train=num_train, pools_drawn=refresh + 1,
                         callback_device=ps.model.where)
    return ps.finish(filelocation, "global", verbose, return_device)


def run_aglmcmc(ABCset, num_ite, Initial_theta, Initial_y, Local_Proposal, Initial_ISIR_prop, filelocation, global_frequency,
                step_size, batch_size, alpha, hat_eps_T, *, seed=None, device=None, chain0=0, return_device=False, verbose=True,
                max_train=None, state_out=None, callback_device="auto"):
    """AGLMCMC (AGLMCMC.py:44-289) with the Model as callbacks: PoolSampler for the iterations, the Model's discrepancy /
    calculate_log_kernel_dis (or calculate_log_kernel(y, epsilon)) / prior_log_prob for the annealed training weights
    (:179-211), the build's KernelDensity kernels for the adaptive proposal (:214-229).  Same schedule and the same max_train
    rule as the fused path (AGLMCMC.py of the build)."""
    import warnings
    from .kernel_density import KernelDensity
    ps = PoolSampler(ABCset, num_ite, Initial_theta, Initial_y, Local_Proposal, global_frequency, step_size, batch_size, seed,
                     device, chain0, callback_device)
    dev, n, d, rows, chains, model = ps.dev, ps.n, ps.d, ps.rows, ps.chains, ps.model
    if not hasattr(ABCset, "discrepancy"):
        raise TypeError("AGLMCMC needs Model.discrepancy (AGLMCMC.py:93)")
    isir = ProposalCallbacks(Initial_ISIR_prop, dev)
    isir_desc = dist_descriptor(Initial_ISIR_prop, d)
    drawn = [0]

    def isir_forward():
        """Initial_ISIR_prop.forward(rows): on the device from the Philox stream when the proposal has a descriptor"""
        if isir_desc is None:
            return isir.forward(rows)
        z = torch.empty(d, rows, dtype=torch.float32, device=dev)
        lq = torch.empty(rows, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _capi.check(ps.lib.glabc_dist_forward(C.byref(isir_desc), rows, ps.key ^ 0x9E3779B97F4A7C15, (drawn[0] << 44) +
                                                  chains.chain0 * ps.N * ps.S, z.data_ptr(), lq.data_ptr(), ps.stream), "glabc_dist_forward")
        drawn[0] += 1
        return z.t().contiguous(), lq

    def discrepancy(y):
        return model._call(lambda cuda: model._back(ABCset.discrepancy(y if cuda else y.cpu()), y.shape[0])).view(-1)

    def kernel_of_discrepancy(dis, x, eps):                                                       # calculate_log_kernel_dis, :199
        if hasattr(ABCset, "calculate_log_kernel_dis"):
            return model._call(lambda cuda: model._back(ABCset.calculate_log_kernel_dis(dis if cuda else dis.cpu(), eps),
                                                        dis.shape[0])).view(-1)
        return model._call(lambda cuda: model._back(ABCset.calculate_log_kernel(x if cuda else x.cpu(), eps), x.shape[0])).view(-1)

    def load(theta_rows, lq):
        pool = ps.load_pool(theta_rows, lq)
        pool["dis"] = discrepancy(pool["x"])                                                      # :93 / 236

    th, lq = isir_forward()                                                                       # :80-81
    load(th, lq)
    KDE, kde_rows, warned = None, 0, False
    num_train, eps_num, hat_eps = 0, 0, 1000000.0                                                 # :119
    with torch.cuda.device(dev):
        for i in range(1, num_ite):
            if KDE is None:                                                                       # :137-140
                ps.log_q_old.copy_(isir.log_prob(chains.theta.t().contiguous()))
            else:
                ps.log_q_old.copy_(KDE.log_prob_soa(chains.theta))
            ps.step(i)
            if not ps.pool_used_up():                                                             # :175
                continue
            pool = ps.pool
            dis0 = pool["dis"]
            if hat_eps > hat_eps_T:                                                               # :179-196
                eps_num += 1
                num_a = torch.sum(dis0 < hat_eps)
                valid = dis0[~torch.isnan(dis0)]
                if valid.numel() > 0:
                    q = torch.clamp((alpha * num_a / valid.shape[0]).to(dis0.dtype), 0.0, 1.0)
                    if valid.numel() > (1 << 24):                                                 # torch.quantile's input limit
                        valid = valid[:: (valid.numel() >> 24) + 1]
                    hat_eps = float(torch.quantile(valid, q))
                hat_eps = max(hat_eps, float(hat_eps_T))
            tw = torch.exp(pool["prior"] + kernel_of_discrepancy(dis0, pool["x"], hat_eps) - pool["lq"])   # :199-202
            tw = torch.where(torch.isnan(tw), torch.zeros_like(tw), tw)
            cap = rows if (max_train is None and n == 1) else int(8192 if max_train is None else max_train)
            if cap < rows and not warned:
                warned = True
                warnings.warn('AGLMCMC: the adaptive KDE is trained on the first %d of %d pool rows (max_train)' % (cap, rows))
            m = min(rows, cap)
            keep = tw[:m] > 0                                                                     # :207-208
            if bool(keep.any()):
                KDE = KernelDensity(bandwidth='silverman', device=dev, seed=ps.key ^ 0xD1B54A32D192ED03)
                KDE.fit(pool["theta"][:m][keep], tw[:m][keep])                                    # :211-215
                num_train += 1
            if KDE is None:                                                                       # no usable weights yet
                th, lq = isir_forward()
                load(th, lq)
                continue
            got, parts = 0, []
            while got < rows:                                                                     # :220-226
                cand = KDE.sample_soa(4 * rows, row0=kde_rows).t().contiguous()
                kde_rows += 4 * rows
                sel = cand[model.prior(cand) > float(np.log(10 ** (-10)))]
                parts.append(sel)
                got += sel.shape[0]
                if sel.shape[0] == 0 and len(parts) > 8:
                    raise RuntimeError("the KDE proposal has left the prior's support")
            theta_new = torch.cat(parts, 0)[:rows].contiguous()
            load(theta_new, KDE.log_prob_soa(theta_new.t().contiguous()))                         # :229-249
    if state_out is not None:
        state_out.update(chains=chains, kde=KDE, hat_eps=hat_eps, num_train=num_train, eps_num=eps_num, pool=ps.pool,
                         callback_device=model.where)
    return ps.finish(filelocation, "aglmcmc", verbose, return_device)


# ----------------------------------------------------------------------------------------------------------- GLMALA
def _numerical_gradient(model, theta, num, eps_sq, gen, seeds, d_step=1e-1):
    """numberical_gradient_logABC (GLMALA.py:46-95) for a batch of rows: central difference (step 0.1) of the synthetic
    log-likelihood -1/2 log(Sigma + eps^2) - 1/2 mu^2 / (Sigma + eps^2) of `num` simulated discrepancies, the +/- sides on
    common random numbers, plus the float32 central difference (h = 1e-5) of the prior.  theta is cast to float32 as the
    reference does (:62); the statistics are float64 (:70-71)."""
    theta = theta.float()
    L, d = theta.shape
    dev = theta.device
    mu = torch.empty(2, L, d, dtype=torch.float64, device=dev)
    var = torch.empty(2, L, d, dtype=torch.float64, device=dev)
    grad_prior = torch.empty(L, d, dtype=torch.float64, device=dev)
    for k in range(d):
        e_k = torch.zeros(d, device=dev)
        e_k[k] = 1.0
        eps = torch.randn(L * num, model.noise_dim, generator=gen, device=dev) if model.noise_dim else None
        for side, sign in enumerate((1.0, -1.0)):
            rows = (theta + sign * d_step * e_k).repeat_interleave(num, dim=0)                  # :78,82
            if eps is None:                                                                      # :76-77,80-81
                torch.manual_seed(int(seeds[k]))
                np.random.seed(int(seeds[k]))
            dis = model.discrepancy(model.simulate(rows, eps)).view(L, num).double()
            mu[side, :, k] = dis.mean(dim=1)                                                     # :86-89
            var[side, :, k] = dis.var(dim=1)
        grad_prior[:, k] = ((model.prior(theta + e_k * 0.00001) - model.prior(theta - e_k * 0.00001))
                            / (2 * 0.00001)).double()                                            # :84-85 (float32 difference)
    logp = -0.5 * torch.log(var + eps_sq) - 0.5 * mu ** 2 / (var + eps_sq)                       # :90-93
    return (logp[0] - logp[1]) / (2 * d_step) + grad_prior                                       # :94-95


def run_glmala(ABCset, num_ite, Initial_theta, Initial_y, tau, num_grad, filelocation, global_frequency, Importance_Proposal,
               batch_size, *, seed=None, device=None, chain0=0, record_history=True, stats=None, return_device=False,
               verbose=True, state_out=None, callback_device="auto", progress=None):
    """GLMALA (GLMALA.py:118-230) with the Model as callbacks.  The iSIR global move is glabc_propose -> callbacks ->
    glabc_select as in GLMCMC; the MALA local move (GLMALA.py:182-200) -- gradient, drift, reverse density -- is evaluated in
    float64 torch operations on the chains that take it, and its accept / state update / Theta_Re row again by glabc_select.
    Reference behaviours kept: log_weight_old is not refreshed after MALA moves, the cached gradient is not
    refreshed after iSIR moves, the prior gradient is a float32 finite difference (B3), and Theta_old becomes a float64 tensor
    at a chain's first ACCEPTED MALA move and stays one (GLMALA.py:43,197-198: the drift is float64): `theta64` holds the
    state in double (float32-exact values until the chain's `th64` bit is set), the next proposal and the reverse density
    start from it -- `z*tau + Theta_old` is a float32 addition before the switch and a float64 one after, as in the fused
    kernel (GLABC_FLAG_TH64).  Theta_Re is float32 in either case (GLMALA.py:148,200).  y_old is not kept in double: it enters
    only through calculate_log_kernel(y_old), which is carried from the iteration that proposed it."""
    lib = _capi.lib()
    dev, chains, single = _host.prepare(ABCset, Initial_theta, Initial_y, device, chain0)
    n, d, yd = chains.n, chains.d, chains.yd
    N = int(batch_size)
    key = engine.draw_seed(seed)
    model = ModelCallbacks(ABCset, dev, callback_device)
    if not hasattr(ABCset, "discrepancy"):
        raise TypeError("GLMALA needs Model.discrepancy (GLMALA.py:78)")
    global_desc = dist_descriptor(Importance_Proposal, d, gamma=True)
    global_cb = ProposalCallbacks(Importance_Proposal, dev) if global_desc is None else None
    tau = float(tau)
    eps_sq = float(ABCset.epsilon) ** 2                                                          # GLMALA.py:90
    gen = torch.Generator(device=dev)
    gen.manual_seed(_noise_seed(key, chain0))
    host_rng = np.random.Generator(np.random.PCG64(_noise_seed(key, chain0)))

    def discrepancy(y):
        return model._call(lambda cuda: model._back(ABCset.discrepancy(y if cuda else y.cpu()), y.shape[0])).view(-1)
    model.discrepancy = discrepancy

    R = N * n
    f32 = dict(dtype=torch.float32, device=dev)
    theta_prop = torch.zeros(R, d, **f32)
    log_q = torch.zeros(R, **f32)
    nd = model.noise_dim
    sim_noise = torch.zeros(R, nd, **f32) if nd else None
    log_u = torch.zeros(n, **f32)
    u_res = torch.zeros(n, dtype=torch.float64, device=dev)
    is_global = torch.zeros(n, dtype=torch.int32, device=dev)
    model.probe(chains.theta.t(), chains.y.t())
    prior_cur = model.prior(chains.theta.t().contiguous()).clone()
    kern_cur = model.kernel(chains.y.t().contiguous()).clone()
    grad = torch.zeros(n, d, dtype=torch.float64, device=dev)                                    # grad_logABC_Theta_old, :146
    has_grad = torch.zeros(n, dtype=torch.bool, device=dev)
    theta64 = chains.theta.t().double().contiguous()                                             # Theta_old, (n, d)
    th64 = torch.zeros(n, dtype=torch.bool, device=dev)                                          # ... is a float64 tensor
    hist = _host.allocate_history(num_ite, chains, record_history)

    io = _capi.StepIO()
    io.n_prop, io.theta_dim, io.y_dim, io.noise_dim = N, d, yd, nd
    io.theta_prop, io.log_q = theta_prop.data_ptr(), log_q.data_ptr()
    io.sim_noise = sim_noise.data_ptr() if nd else None
    io.log_u, io.u_res, io.is_global = log_u.data_ptr(), u_res.data_ptr(), is_global.data_ptr()
    io.prior_cur, io.kern_cur = prior_cur.data_ptr(), kern_cur.data_ptr()
    cs = chains.struct()
    ms = stats.struct() if stats is not None else None
    run_ = _capi.Run()
    run_.seed, run_.n_steps, run_.global_frequency, run_.batch_size, run_.hist_stride = key, 1, float(global_frequency), N, n
    if ms is not None:
        run_.moments = C.pointer(ms)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    hist_ptr, hist_row_bytes = (hist.data_ptr(), hist[0].numel() * 4) if hist is not None else (0, 0)
    gp = C.byref(global_desc) if global_desc is not None else None
    c_norm = -0.5 * d * math.log(2 * math.pi)

    with torch.cuda.device(dev):
        for i in range(1, num_ite):
            run_.step0 = i
            run_.history = hist_ptr + i * hist_row_bytes if hist is not None else None
            _capi.check(lib.glabc_propose(_capi.ALGO_GLMALA, None, gp, C.byref(cs), C.byref(run_), C.byref(io), stream),
                        "glabc_propose")
            if global_cb is not None:                                                           # GLMALA.py:158
                z, lq = global_cb.forward(R)
                theta_prop.copy_(z)
                log_q.copy_(lq)
            # ---- iSIR candidates of every chain (rows of chains on the local branch are overwritten below) :158-165
            prior_prop = model.prior(theta_prop)
            y_prop = model.simulate(theta_prop, sim_noise)
            kern_prop = model.kernel(y_prop)
            # ---- MALA move of the chains on the local branch, GLMALA.py:182-200
            idx = torch.nonzero(is_global == 0).view(-1)
            L = int(idx.numel())
            g_new = None
            if L:
                th_old = theta64[idx]                                                            # (L, d) float64 (float32-exact before the switch)
                wide = th64[idx].view(-1, 1)
                need = ~has_grad[idx]
                if bool(need.any()):                                                             # :183-184 (theta.float(), :62)
                    sub = idx[need]
                    grad[sub] = _numerical_gradient(model, th_old[need].float(), int(num_grad), eps_sq, gen,
                                                    host_rng.integers(0, 2 ** 32, d))
                    has_grad[sub] = True
                g_old = grad[idx]
                z = torch.randn(L, d, generator=gen, device=dev)                                 # Local_proposal_forward, :25-44
                zt = z * tau                                                                     # float32
                base = torch.where(wide, zt.double() + th_old, (zt + th_old.float()).double())   # float64 add once Theta_old is float64
                th_new = base + g_old * tau ** 2 / 2                                             # float64, :43
                logq_fwd = c_norm - (0.5 * z ** 2).sum(1)
                g_new = _numerical_gradient(model, th_new, int(num_grad), eps_sq, gen, host_rng.integers(0, 2 ** 32, d))   # :187
                eps1 = torch.randn(L, nd, generator=gen, device=dev) if nd else None
                y_new = model.simulate(th_new, eps1)                                             # :188-189
                e_rev = (th_old - th_new - g_new * tau ** 2 / 2) / tau                           # log_proposal, :97-116
                rev = c_norm - (0.5 * e_rev ** 2).sum(1)
                theta_prop[idx] = th_new.float()
                y_prop[idx] = y_new
                prior_prop[idx] = model.prior(th_new)
                kern_prop[idx] = model.kernel(y_new)
                log_q[idx] = (rev - logq_fwd.double()).float()                                   # the proposal terms of :190-193
            io.prior_prop, io.y_prop, io.kern_prop = prior_prop.data_ptr(), y_prop.data_ptr(), kern_prop.data_ptr()
            if global_cb is not None:
                q_cur = global_cb.log_prob(chains.theta.t().contiguous())
                io.q_cur = q_cur.data_ptr()
            _capi.check(lib.glabc_select(_capi.ALGO_GLMALA, gp, C.byref(cs), C.byref(run_), C.byref(io), stream), "glabc_select")
            gm = ((is_global & 1) != 0) & ((is_global & 2) != 0)                                 # an accepted iSIR move: a float32 candidate
            theta64[gm] = chains.theta.t()[gm].double()
            if L:                                                                                # :194-199
                moved = (is_global[idx] & 2) != 0
                grad[idx[moved]] = g_new[moved]
                theta64[idx[moved]] = th_new[moved]                                              # :197 Theta_old = Theta_prop (float64)
                th64[idx[moved]] = True
            if progress is not None:
                progress(i)
    if stats is not None:
        stats.steps += num_ite - 1
    if state_out is not None:
        state_out.update(chains=chains, grad=grad, has_grad=has_grad, theta64=theta64, th64=th64, callback_device=model.where)
    return _host.finish(hist, chains, single, filelocation, "global", verbose and single, return_device)
