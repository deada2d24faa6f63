"""GLMCMC -- iSIR global move + random-walk MH local move (reference: GLMCMC.py:24-137).

Same positional signature as the reference function.  The per-iteration body
(GLMCMC.py:58-104) runs inside one fused gfx950 kernel launch per K iterations
(`glabc_glmcmc_steps`, include/glabc.h) for all chains at once:

* ``Initial_theta`` of shape (d,) / (1, d): one chain; returns the reference's
  ``Theta_Re`` -- a (num_ite, d) float32 CPU tensor, row 0 = Initial_theta.
* ``Initial_theta`` of shape (C, d): C independent chains (``Initial_y`` (C, y_dim));
  returns (num_ite, C, d).

Keyword-only extras (all optional): ``seed`` (Philox key; default drawn from torch's
global generator), ``device``, ``chain0`` (global id of the first chain -- the shard
offset in a multi-GPU run), ``record_history`` (False: return None, keep only
``stats``), ``stats`` (an ``engine.Moments`` to accumulate ESJD / moment sums into),
``return_device`` (leave the result on the GPU), ``steps_per_launch``, ``verbose``.

Dispatch (``path``): a Model and proposals that describe themselves (``descriptor()``; theta_dim 1..4 or the g-and-k
shape; batch_size <= 4096) run in the fused kernels (above 16 proposals lane groups of a wavefront share a chain); ANY other Model object implementing the reference's callbacks
(``generate_samples / prior_log_prob / calculate_log_kernel``, examples/Mixture.py:5-53), any theta_dim, any batch_size
and any proposal object run through the split-phase path of ``generic.py`` (``glabc_propose`` -> callbacks ->
``glabc_select``).  ``path="generic"`` forces the latter; ``path="fused"`` raises instead of falling back.
Generic-path extras: ``callback_device`` ('auto' | 'cuda' | 'cpu'), ``sentinel_redraw`` (GLMCMC.py:92-93, default on).
"""
from . import _capi, _host, engine, generic


def GLMCMC(ABCset, num_ite, Initial_theta, Initial_y, Local_Proposal,
           filelocation, global_frequency=0, Importance_Proposal=None, batch_size=None, *,
           seed=None, device=None, chain0=0, record_history=True, stats=None, return_device=False,
           steps_per_launch=None, verbose=True, state_out=None, path="auto", fast_math=False, **generic_kw):
    if Importance_Proposal is None or batch_size is None:
        raise ValueError("GLMCMC needs Importance_Proposal and batch_size (GLMCMC.py:54,66)")
    if path not in ("auto", "fused", "generic"):
        raise ValueError("path must be 'auto', 'fused' or 'generic'")
    if fast_math and path == "generic":
        raise ValueError("fast_math is a variant of the fused kernel (glabc_run.math_mode = GLABC_MATH_FAST)")
    if path == "generic" or (path == "auto" and not fast_math and not generic.fused_supported(ABCset, (Local_Proposal, Importance_Proposal),
                                                                             batch_size, _capi.MAX_BATCH_WIDE, gamma_ok=True)):
        return generic.run(_capi.ALGO_GLMCMC, ABCset, num_ite, Initial_theta, Initial_y, Local_Proposal, Importance_Proposal,
                           filelocation, global_frequency, batch_size, "glmcmc", seed=seed, device=device, chain0=chain0,
                           record_history=record_history, stats=stats, return_device=return_device, verbose=verbose,
                           state_out=state_out, **generic_kw)
    if generic_kw:
        raise TypeError("unexpected keyword arguments for the fused path: %s" % sorted(generic_kw))
    model = engine.model_descriptor(ABCset)
    local = Local_Proposal.descriptor()
    imp = Importance_Proposal.descriptor()
    dev, chains, single = _host.prepare(ABCset, Initial_theta, Initial_y, device, chain0)
    hist = _host.allocate_history(num_ite, chains, record_history)
    mirror = _host.HostMirror(hist) if _host.HostMirror.wanted(hist, single, return_device) else None   # rows leave for the host while the kernels run
    rtc = None
    if model.sim_kind == _capi.SIM_USER:                           # compiled.CompiledModel: the simulator is run-time compiled C
        rtc = ABCset.program(_capi.ALGO_GLMCMC, batch_size)       # (log_weight_old is computed at the first global move: `local` starts set)
    else:
        engine.init_weights(model, imp, chains)                    # GLMCMC.py:52-55
    engine.run_steps("glabc_glmcmc_steps", model, local, imp, chains, num_ite - 1, 1, engine.draw_seed(seed),
                     global_frequency, batch_size, history=None if hist is None else hist[1:], moments=stats,
                     steps_per_launch=steps_per_launch, rtc_program=rtc, mirror=mirror,
                     math_mode=_capi.MATH_FAST if fast_math else _capi.MATH_EXACT)
    if state_out is not None:
        state_out["chains"] = chains
    return _host.finish(hist, chains, single, filelocation, "glmcmc", verbose and single, return_device, mirror=mirror)
