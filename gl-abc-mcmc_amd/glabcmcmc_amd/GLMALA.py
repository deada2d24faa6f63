"""GLMALA -- iSIR global move + MALA local move (reference: GLMALA.py:118-230).

Same positional signature as the reference function.  The loop body (GLMALA.py:150-200) --
including the common-random-number finite-difference gradient of GLMALA.py:46-95 -- runs in
the fused gfx950 kernel behind ``glabc_glmala_steps`` (include/glabc.h).  Shapes, return value
and keyword-only extras as in ``GLMCMC``.

Reference behaviours that are reproduced on purpose: the chain's state
becomes float64 after the first accepted MALA move, ``log_weight_old`` is never refreshed after
MALA moves (B1), the cached gradient is not refreshed after iSIR moves, the prior gradient is a
float32 finite difference (B3).  Not reproducible: the reference reseeds torch / NumPy from
``secrets`` inside every gradient (B2), so two reference runs never agree; here the gradient noise
is a Philox sub-stream of the run's seed.
"""
from . import _capi, _host, engine, generic


def GLMALA(ABCset, num_ite, Initial_theta, Initial_y, tau, num_grad,
           filelocation, global_frequency=0, Importance_Proposal=None, batch_size=None, *,
           seed=None, device=None, chain0=0, record_history=True, stats=None, return_device=False,
           steps_per_launch=None, verbose=True, state_out=None, path="auto", **generic_kw):
    if Importance_Proposal is None or batch_size is None:
        raise ValueError("GLMALA needs Importance_Proposal and batch_size (GLMALA.py:155,158)")
    if path not in ("auto", "fused", "generic"):
        raise ValueError("path must be 'auto', 'fused' or 'generic'")
    fused_ok = generic.fused_supported(ABCset, (Importance_Proposal,), batch_size, max_dim=4) and \
        engine.model_descriptor(ABCset).sim_kind == _capi.SIM_ABS_GAUSS
    if path == "generic" or (path == "auto" and not fused_ok):
        # a Model given as callbacks (generic.py): iSIR through glabc_propose / glabc_select, the MALA move's gradient
        # through the Model's generate_samples / discrepancy in float64 torch operations
        return generic.run_glmala(ABCset, num_ite, Initial_theta, Initial_y, tau, num_grad, filelocation, global_frequency,
                                  Importance_Proposal, batch_size, seed=seed, device=device, chain0=chain0,
                                  record_history=record_history, stats=stats, return_device=return_device, verbose=verbose,
                                  state_out=state_out, **generic_kw)
    if generic_kw:
        raise TypeError("unexpected keyword arguments for the fused path: %s" % sorted(generic_kw))
    model = engine.model_descriptor(ABCset)
    imp = Importance_Proposal.descriptor()
    mala = _capi.Mala(float(tau), float(tau) ** 2, float(ABCset.epsilon) ** 2, int(num_grad), 0)   # GLMALA.py:43,90
    dev, chains, single = _host.prepare(ABCset, Initial_theta, Initial_y, device, chain0)
    chains.add_mala_state()
    hist = _host.allocate_history(num_ite, chains, record_history)
    mirror = _host.HostMirror(hist) if _host.HostMirror.wanted(hist, single, return_device) else None   # rows leave for the host while the kernels run
    engine.glmala_init(model, chains)                                  # GLMALA.py:143-149
    engine.run_glmala_steps(model, imp, mala, chains, num_ite - 1, 1, engine.draw_seed(seed), global_frequency,
                            batch_size, history=None if hist is None else hist[1:], moments=stats,
                            steps_per_launch=steps_per_launch, mirror=mirror)
    if state_out is not None:
        state_out["chains"] = chains
    return _host.finish(hist, chains, single, filelocation, "global", verbose and single, return_device, mirror=mirror)
