"""AGLMCMC -- iSIR against a pool drawn from an adaptive KDE proposal, threshold annealing, RW-MH local move
(reference: AGLMCMC.py:44-289).

Same positional signature as the reference.  Per iteration the chain state never leaves the GPU:

* initial pool ``Initial_ISIR_prop.forward(batch_size*step_size)``        -> ``glabc_dist_forward``,      AGLMCMC.py:80-81
* pool simulate / discrepancy / weights                                   -> ``glabc_pool_weights`` +
                                                                             ``glabc_model_discrepancy``, :90-112, 231-249
* proposal density of the current state (ISIR proposal, later the KDE)    -> ``glabc_dist_log_prob`` /
                                                                             ``glabc_kde_log_prob``,      :137-140
* iSIR against the next pool slice, or the RW-MH local move               -> ``glabc_glmcmc_nf_step``,    :125-172, 251-272
* every ``step_size`` global moves: anneal ``hat_eps`` to the ``alpha*num_a/n`` quantile of the pool's discrepancies
  (:179-196, ``torch.quantile`` on the device), training weights under ``hat_eps`` -> ``glabc_kde_train_weights`` (:199-211),
  ``KernelDensity.fit`` (:214-215), ``4x`` oversampled draw filtered by the prior (:220-226), ``KDE.log_prob`` of the new
  pool (:229) and its weights (:231-249).

Fixed with respect to the reference: ``Theta_Re`` has ``num_ite`` rows (the reference allocates 10 000
and raises IndexError beyond, :117) and is returned (the reference falls off the end and returns None).

Batched use (``Initial_theta`` (C, d)): all chains share ONE adaptive proposal; every chain owns a pool; pools are
refreshed together as soon as one chain has used its ``step_size`` slices, and the KDE is trained on the first
``max_train`` pool rows (rows are slice-major, so every chain contributes).  With C = 1 this is the reference's schedule.
"""
import ctypes as C
import warnings

import numpy as np
import torch

from . import _capi, _host, engine
from .kernel_density import KernelDensity

_LOG_PRIOR_FLOOR = float(np.log(10 ** (-10)))                                       # AGLMCMC.py:224
# Philox keys of the three pool-side streams (key ^ constant); the per-iteration draws use the run key itself
ISIR_STREAM, POOL_STREAM, KDE_STREAM = 0x9E3779B97F4A7C15, 0x5851F42D4C957F2D, 0xD1B54A32D192ED03


def AGLMCMC(ABCset, num_ite, Initial_theta, Initial_y, Local_Proposal, Initial_ISIR_prop,
            filelocation, global_frequency, step_size, batch_size, alpha, hat_eps_T, device=None, *,
            seed=None, chain0=0, return_device=False, verbose=True, max_train=None, state_out=None, path="auto",
            check_density_cache=False, **generic_kw):
    if path not in ("auto", "fused", "generic"):
        raise ValueError("path must be 'auto', 'fused' or 'generic'")
    from . import generic
    desc = generic.try_descriptor(ABCset)
    builtin = isinstance(desc, _capi.Model) and desc.sim_kind in (_capi.SIM_ABS_GAUSS, _capi.SIM_GK) and \
        desc.prior.kind != _capi.DIST_GAMMA and \
        generic.dist_descriptor(Local_Proposal, desc.theta_dim) is not None and \
        generic.dist_descriptor(Initial_ISIR_prop, desc.theta_dim) is not None
    if path == "generic" or (path == "auto" and not builtin):
        # a Model (or proposals) given as callbacks: generic.run_aglmcmc
        return generic.run_aglmcmc(ABCset, num_ite, Initial_theta, Initial_y, Local_Proposal, Initial_ISIR_prop, filelocation,
                                   global_frequency, step_size, batch_size, alpha, hat_eps_T, seed=seed, device=device,
                                   chain0=chain0, return_device=return_device, verbose=verbose, max_train=max_train,
                                   state_out=state_out, **generic_kw)
    if generic_kw:
        raise TypeError("unexpected keyword arguments for the fused path: %s" % sorted(generic_kw))
    lib = _capi.lib()
    model = engine.model_descriptor(ABCset)
    local = Local_Proposal.descriptor()
    isir = Initial_ISIR_prop.descriptor()
    dev, chains, single = _host.prepare(ABCset, Initial_theta, Initial_y, device, chain0)
    n, d, N, P = chains.n, chains.d, int(batch_size), int(batch_size) * int(step_size)
    rows = P * n
    key = engine.draw_seed(seed)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    hist = _host.allocate_history(num_ite, chains, True)
    kk = torch.zeros(n, dtype=torch.int32, device=dev)
    pool = {}
    refresh = [0]

    def weigh_pool(theta, lq):
        """x0, dis0, weight0 of a pool (AGLMCMC.py:90-112 / 231-249); row r = p*n + c"""
        x = torch.empty(chains.yd, rows, dtype=torch.float32, device=dev)
        w = torch.empty(rows, dtype=torch.float32, device=dev)
        dis = torch.empty(rows, dtype=torch.float32, device=dev)
        x_rows = None
        with torch.cuda.device(dev):
            _capi.check(lib.glabc_pool_weights(C.byref(model), theta.data_ptr(), lq.data_ptr(), rows, key ^ POOL_STREAM,
                                               refresh[0] * rows, x.data_ptr(), w.data_ptr(), stream), "glabc_pool_weights")
            x_rows = x.t().contiguous()
            _capi.check(lib.glabc_model_discrepancy(C.byref(model), x_rows.data_ptr(), rows, dis.data_ptr(), stream),
                        "glabc_model_discrepancy")
        pool.update(theta=theta, x=x, w=w, lq=lq, dis=dis)
        kk.zero_()
        refresh[0] += 1

    theta0 = torch.empty(d, rows, dtype=torch.float32, device=dev)
    lq0 = torch.empty(rows, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _capi.check(lib.glabc_dist_forward(C.byref(isir), rows, key ^ ISIR_STREAM, 0, theta0.data_ptr(), lq0.data_ptr(),
                                           stream), "glabc_dist_forward")                        # :80-81
    weigh_pool(theta0, lq0)

    KDE = None
    kde_rows = [0]
    warned = []
    num_train, eps_num = 0, 0
    hat_eps = 1000000.0                                                                           # :119
    log_q_old = torch.empty(n, dtype=torch.float32, device=dev)
    # a chain uses at most one pool slice per iteration: the device is only asked (a sync) whether a pool is used up
    # when that has become possible -- the same schedule as checking after every iteration
    countdown = int(step_size)
    cs = chains.struct()                                                                          # the state arrays never move
    run = _capi.Run()
    run.seed, run.n_steps, run.global_frequency, run.batch_size, run.hist_stride = key, 1, float(global_frequency), N, n
    hist_ptr, hist_row_bytes = hist.data_ptr(), hist[0].numel() * 4
    # The proposal density of a chain's current state (:137-140) is a pure function of the state and the proposal: with the
    # KDE -- an O(centres) sum per chain -- it is kept per chain and re-evaluated only for the chains the step reports as moved
    # (glabc_kde_log_prob_indexed, list and count on the device) and for all chains after a refit; same values as evaluating
    # every chain every iteration.
    moved_idx = torch.zeros(n, dtype=torch.int32, device=dev)
    n_moved = torch.zeros(2, dtype=torch.int32, device=dev)                                        # counters of odd / even iterations
    cnt = [n_moved[0:].data_ptr(), n_moved[1:].data_ptr()]
    kde_current = None                                                                            # the KDE log_q_old was computed with
    for i in range(1, num_ite):
        with torch.cuda.device(dev):
            if KDE is None:                                                                       # :137-140
                th_rows = chains.theta.t().contiguous()
                _capi.check(lib.glabc_dist_log_prob(C.byref(isir), th_rows.data_ptr(), n, log_q_old.data_ptr(), stream),
                            "glabc_dist_log_prob")
            elif kde_current is not KDE:
                log_q_old = KDE.log_prob_soa(chains.theta)
                kde_current = KDE
            else:
                KDE.log_prob_soa_indexed(chains.theta, moved_idx, n_moved[(i + 1) & 1:], log_q_old)   # the previous step's movers
                if check_density_cache and not torch.equal(log_q_old.view(torch.int32), KDE.log_prob_soa(chains.theta).view(torch.int32)):
                    raise AssertionError("AGLMCMC: the cached proposal densities differ from a full evaluation at iteration %d" % i)
            pd = _capi.Pool(pool["theta"].data_ptr(), pool["x"].data_ptr(), pool["w"].data_ptr(), log_q_old.data_ptr(),
                            kk.data_ptr(), int(step_size), 0, moved_idx.data_ptr(), cnt[i & 1], cnt[(i + 1) & 1])
            run.step0, run.history = i, hist_ptr + i * hist_row_bytes
            _capi.check(lib.glabc_glmcmc_nf_step(C.byref(model), C.byref(local), C.byref(pd), C.byref(cs), C.byref(run),
                                                 stream), "glabc_glmcmc_nf_step")                 # :125-172, 251-272
        countdown -= 1
        if countdown > 0:
            continue
        used = int(kk.max().item())
        if used < int(step_size):                                                                 # :175
            countdown = int(step_size) - used
            continue
        countdown = int(step_size)
        dis0 = pool["dis"]
        if hat_eps > hat_eps_T:                                                                   # :179-196
            eps_num += 1
            num_a = torch.sum(dis0 < hat_eps)
            valid = dis0[~torch.isnan(dis0)]
            if valid.numel() > 0:
                q = torch.clamp((alpha * num_a / valid.shape[0]).to(dis0.dtype), 0.0, 1.0)
                if valid.numel() > (1 << 24):                                                     # torch.quantile's input limit
                    valid = valid[:: (valid.numel() >> 24) + 1]
                hat_eps = float(torch.quantile(valid, q))
            hat_eps = max(hat_eps, float(hat_eps_T))
        train_model = ABCset.descriptor(hat_eps)                                                  # calculate_log_kernel_dis(dis0, hat_eps), :199
        tw = torch.empty(rows, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _capi.check(lib.glabc_kde_train_weights(C.byref(train_model), pool["theta"].data_ptr(), dis0.data_ptr(),
                                                    pool["lq"].data_ptr(), rows, tw.data_ptr(), stream), "glabc_kde_train_weights")
        # the reference trains its KDE on all batch_size*step_size pool rows (AGLMCMC.py:199-215): with one chain that is
        # what happens here (max_train None); a batch of chains shares ONE density, whose O(points x centres) evaluation is
        # capped at 8192 centres unless the caller asks for more
        cap = rows if (max_train is None and n == 1) else int(8192 if max_train is None else max_train)
        if cap < rows and not warned:
            warned.append(True)
            warnings.warn('AGLMCMC: the adaptive KDE is trained on the first %d of %d pool rows (max_train)' % (cap, rows))
        m = min(rows, cap)
        keep = tw[:m] > 0                                                                         # :207-208
        if bool(keep.any()):
            KDE = KernelDensity(bandwidth='silverman', device=dev, seed=key ^ KDE_STREAM)
            KDE.fit(pool["theta"][:, :m].t()[keep], tw[:m][keep])                                 # :211-215
            num_train += 1
        if KDE is None:                                                                           # no usable weights yet: keep the ISIR proposal
            theta_new = torch.empty(d, rows, dtype=torch.float32, device=dev)
            lq_new = torch.empty(rows, dtype=torch.float32, device=dev)
            with torch.cuda.device(dev):
                _capi.check(lib.glabc_dist_forward(C.byref(isir), rows, key ^ ISIR_STREAM, refresh[0] * rows,
                                                   theta_new.data_ptr(), lq_new.data_ptr(), stream), "glabc_dist_forward")
            weigh_pool(theta_new, lq_new)
            continue
        got, parts = 0, []
        while got < rows:                                                                         # :220-226 (4x oversampling, prior floor)
            cand = KDE.sample_soa(4 * rows, row0=kde_rows[0])                                     # one stream across all refits
            kde_rows[0] += 4 * rows
            ok = ABCset.prior_log_prob(cand.t().contiguous()) > _LOG_PRIOR_FLOOR
            sel = cand[:, ok]
            parts.append(sel)
            got += sel.shape[1]
            if sel.shape[1] == 0 and len(parts) > 8:
                raise RuntimeError("the KDE proposal has left the prior's support")
        theta_new = torch.cat(parts, 1)[:, :rows].contiguous()
        weigh_pool(theta_new, KDE.log_prob_soa(theta_new))                                        # :229-249
    if state_out is not None:
        state_out.update(chains=chains, kde=KDE, hat_eps=hat_eps, num_train=num_train, eps_num=eps_num, pool=pool)
    return _host.finish(hist, chains, single, filelocation, "aglmcmc", verbose and single, return_device)
