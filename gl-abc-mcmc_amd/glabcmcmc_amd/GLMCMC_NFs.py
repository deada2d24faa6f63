"""GLMCMC_NF -- iSIR against a pool of normalizing-flow proposals + random-walk MH local move
(reference: GLMCMC_NFs.py:43-186).

Same positional signature as the reference function (``base`` is accepted for signature parity; pass a
``flows.BaseDiagGaussian`` or None -- the reference passes ``nf.distributions.base.DiagGaussian(2)``).  The flow is
``flows.RealNVP`` (the restatement of the normflows model the reference builds, GLMCMC_NFs.py:51-61):

* pool draw  ``NF_model.sample(batch_size*step_size)`` per chain  -> ``glabc_nf_sample`` (f32 MFMA), GLMCMC_NFs.py:70-72
* pool weights (simulate, prior, kernel, exp)                     -> ``glabc_pool_weights``,        :73-85
* per iteration: iSIR against the next pool slice / RW-MH          -> ``glabc_glmcmc_nf_step``,        :92-111,141-152
                 ``NF_model.log_prob(Theta_old)`` (:96-98) is a pure function of the state and the flow: it is kept per
                 chain and re-evaluated only for the chains that moved (``glabc_nf_log_prob_indexed``, f32 MFMA, the list
                 and its length stay on the device) and for all chains after an optimizer step -- same values, a few per
                 cent of the work
* when a pool is used up: at most ``Train_step`` Adam steps on the forward KL of a systematically resampled pool
  (:112-124, ``resample`` :29-40) -- ``flows.HipAdam``: hand-written backward on the matrix cores (``glabc_nf_grad``) and
  ``glabc_adam_step`` -- then a new pool (:125-140).

Batched use (``Initial_theta`` of shape (C, d)): all chains share ONE flow; every chain owns a pool; pools are
redrawn together as soon as one chain has used its ``step_size`` slices.  With C = 1 this is the reference's schedule.
Keyword-only extras: ``num_layers`` (reference: 32, GLMCMC_NFs.py:51), ``seed``, ``device``, ``chain0``,
``return_device``, ``verbose``, ``flow`` (reuse / inspect the model), ``lr`` / ``weight_decay`` (:63), ``process_group``
(chains sharded over ranks that share the flow: the ranks refresh their pools together -- the fullest pool of any rank
decides, one all-reduce of a word at each check -- and gradients are averaged over the ranks' pools before each Adam step;
give every rank the same ``num_ite``, ``step_size`` and initial flow).
"""
import ctypes as C

import torch

from . import _capi, _host, engine
from .flows import HipAdam, RealNVP


def resample(W, N, u0=None):
    """Systematic resampling (GLMCMC_NFs.py:29-40): draw i of u_i = (u0 + i)/N goes to the first index j whose cumulative
    weight exceeds it; draws at or beyond the last cumulative weight find no index and are dropped, as the reference's
    counting loop drops them (the result then has fewer than N entries).  The reference's float32 ``torch.cumsum`` on the CPU
    accumulates in double and rounds every partial sum to float32 -- spelled out here so that the device scan (whose order of
    additions differs) gives the same float32 values."""
    if u0 is None:
        u0 = torch.rand(1, device=W.device)
    u = (torch.as_tensor(u0, dtype=torch.float32, device=W.device).view(1) + torch.arange(N, device=W.device)) / N
    Psum = torch.cumsum(W.double(), dim=0).to(W.dtype)
    idx = torch.searchsorted(Psum, u, right=True)               # Psum[j-1] <= u_i < Psum[j]  <=>  index j gets u_i
    return idx[idx < W.numel()]


def GLMCMC_NF(ABCset, num_ite, Initial_theta, Initial_y, Local_Proposal,
              filelocation, global_frequency, step_size, batch_size, base, Train_step, *,
              num_layers=32, seed=None, device=None, chain0=0, return_device=False, verbose=True, flow=None,
              lr=5e-4, weight_decay=1e-5, state_out=None, path="auto", process_group=None, **generic_kw):
    if path not in ("auto", "fused", "generic"):
        raise ValueError("path must be 'auto', 'fused' or 'generic'")
    from . import generic
    desc = generic.try_descriptor(ABCset)
    builtin = isinstance(desc, _capi.Model) and desc.sim_kind in (_capi.SIM_ABS_GAUSS, _capi.SIM_GK) and \
        desc.prior.kind != _capi.DIST_GAMMA and \
        generic.dist_descriptor(Local_Proposal, desc.theta_dim) is not None
    if path == "generic" or (path == "auto" and not builtin):
        # a Model given as callbacks (or a CompiledModel): pools, local moves and weights through the Model's own methods
        return generic.run_glmcmc_nf(ABCset, num_ite, Initial_theta, Initial_y, Local_Proposal, filelocation, global_frequency,
                                     step_size, batch_size, base, Train_step, num_layers=num_layers, seed=seed, device=device,
                                     chain0=chain0, return_device=return_device, verbose=verbose, flow=flow, lr=lr,
                                     weight_decay=weight_decay, state_out=state_out, process_group=process_group, **generic_kw)
    if generic_kw:
        raise TypeError("unexpected keyword arguments for the fused path: %s" % sorted(generic_kw))
    lib = _capi.lib()
    model = engine.model_descriptor(ABCset)
    local = Local_Proposal.descriptor()
    dev, chains, single = _host.prepare(ABCset, Initial_theta, Initial_y, device, chain0)
    if chains.d != 2:
        raise ValueError("the RealNVP of GLMCMC_NF is built for theta_dim = 2 (MLP([1,128,128,2]), GLMCMC_NFs.py:56)")
    n, N, P = chains.n, int(batch_size), int(batch_size) * int(step_size)
    key = engine.draw_seed(seed)
    if flow is None:
        flow = RealNVP(num_layers, base if isinstance(base, torch.nn.Module) else None)
    flow = flow.to(dev)
    optimizer = HipAdam(flow, lr=lr, weight_decay=weight_decay)                                  # :63, state on the device
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    hist = _host.allocate_history(num_ite, chains, True)
    kk = torch.zeros(n, dtype=torch.int32, device=dev)
    pool = {}
    losses = []

    def draw_pool(refresh_id):
        rows = P * n
        flow.eval()
        row0 = (refresh_id << 44) + chains.chain0 * P          # pool refresh, then this shard's rows: disjoint over ranks
        z, lq = flow.sample(rows, seed=key ^ 0x9E3779B97F4A7C15, row0=row0)                      # row r = p*n + c
        theta = z.t().contiguous()                                                                 # [2][rows]
        x = torch.empty(chains.yd, rows, dtype=torch.float32, device=dev)
        w = torch.empty(rows, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _capi.check(lib.glabc_pool_weights(C.byref(model), theta.data_ptr(), lq.data_ptr(), rows, key ^ 0x5851F42D4C957F2D,
                                               row0, x.data_ptr(), w.data_ptr(), stream), "glabc_pool_weights")
        pool.update(theta=theta, x=x, w=w, lq=lq)
        kk.zero_()

    refresh, num_train = 0, 0
    draw_pool(refresh)
    log_q_old = torch.empty(n, dtype=torch.float32, device=dev)
    moved_idx = torch.zeros(n, dtype=torch.int32, device=dev)
    n_moved = torch.zeros(2, dtype=torch.int32, device=dev)                                        # counters of odd / even iterations
    blob = flow.packed_params()                                                                    # repacked only after an optimizer step
    fdesc = flow.descriptor(blob)

    def log_prob_all():                                                                            # :96-98 for every chain
        with torch.cuda.device(dev):
            _capi.check(lib.glabc_nf_log_prob(C.byref(fdesc), chains.theta.data_ptr(), n, log_q_old.data_ptr(), stream),
                        "glabc_nf_log_prob")
    log_prob_all()
    # a chain uses at most one pool slice per iteration, so the pools cannot run out before `countdown` more
    # iterations: the device is only asked (a sync) when that is possible -- same schedule as checking every iteration
    countdown = int(step_size)
    cs = chains.struct()                                                                          # the state arrays never move
    run = _capi.Run()
    run.seed, run.n_steps, run.global_frequency, run.batch_size, run.hist_stride = key, 1, float(global_frequency), N, n
    hist_ptr, hist_row_bytes = hist.data_ptr(), hist[0].numel() * 4
    cnt = [n_moved[0:].data_ptr(), n_moved[1:].data_ptr()]
    for i in range(1, num_ite):
        with torch.cuda.device(dev):
            pd = _capi.Pool(pool["theta"].data_ptr(), pool["x"].data_ptr(), pool["w"].data_ptr(), log_q_old.data_ptr(),
                            kk.data_ptr(), int(step_size), 0, moved_idx.data_ptr(), cnt[i & 1], cnt[(i + 1) & 1])
            run.step0, run.history = i, hist_ptr + i * hist_row_bytes
            _capi.check(lib.glabc_glmcmc_nf_step(C.byref(model), C.byref(local), C.byref(pd), C.byref(cs), C.byref(run),
                                                 stream), "glabc_glmcmc_nf_step")
            _capi.check(lib.glabc_nf_log_prob_indexed(C.byref(fdesc), chains.theta.data_ptr(), n, moved_idx.data_ptr(),
                                                      cnt[i & 1], n, log_q_old.data_ptr(), stream),
                        "glabc_nf_log_prob_indexed")                                               # :96-98 for the chains that moved
        countdown -= 1
        if countdown > 0:
            continue
        used = int(kk.max().item())
        if process_group is not None:              # ranks sharing the flow refresh (and train) together: the fullest pool decides
            from .parallel import max_over_ranks
            used = max_over_ranks(used, None if process_group is True else process_group,
                                  dev if torch.distributed.get_backend(None if process_group is True else process_group) == "nccl" else "cpu")
        if used < int(step_size):                                                                  # :112
            countdown = int(step_size) - used
            continue
        if num_train < Train_step:                                                                 # :114-124
            w = pool["w"]
            Train_weight = w / torch.sum(w)
            idx = resample(Train_weight, w.numel())
            # zero_grad, forward_kld, backward unless the loss is NaN / inf, optimizer.step(): hand-written backward + Adam
            loss = optimizer.step(pool["theta"][:, idx], chain_major=True, group=process_group,
                                  via=None if process_group is None or torch.distributed.get_backend(
                                      None if process_group is True else process_group) == "nccl" else "cpu")
            num_train += 1
            losses.append(loss)
            blob = flow.packed_params()
            fdesc = flow.descriptor(blob)
            log_prob_all()                                                                         # the flow changed under every chain
        refresh += 1
        draw_pool(refresh)                                                                         # :125-140
        countdown = int(step_size)
    if state_out is not None:
        state_out.update(chains=chains, flow=flow, loss_hist=losses, num_train=num_train, pools_drawn=refresh + 1)
    return _host.finish(hist, chains, single, filelocation, "global", verbose and single, return_device)
