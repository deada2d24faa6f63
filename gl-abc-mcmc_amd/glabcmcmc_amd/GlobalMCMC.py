"""GlobalMCMC -- independence-MH global move + random-walk MH local move
(reference: GlobalMCMC.py:6-98).  Same positional signature; the loop body
(GlobalMCMC.py:37-68) is the fused gfx950 kernel behind ``glabc_globalmcmc_steps``.
Shapes, return value, keyword-only extras and the fused / generic dispatch (``path``) as in ``GLMCMC``."""
from . import _capi, _host, engine, generic


def GlobalMCMC(ABCset, num_ite, Initial_theta, Initial_y,
               Global_Proposal, filelocation, global_frequency, Local_Proposal=None, *,
               seed=None, device=None, chain0=0, record_history=True, stats=None, return_device=False,
               steps_per_launch=None, verbose=True, state_out=None, path="auto", **generic_kw):
    if Local_Proposal is None:
        if global_frequency < 1:
            raise ValueError("GlobalMCMC needs Local_Proposal unless global_frequency >= 1 (GlobalMCMC.py:56)")
        Local_Proposal = Global_Proposal
    if path not in ("auto", "fused", "generic"):
        raise ValueError("path must be 'auto', 'fused' or 'generic'")
    if path == "generic" or (path == "auto" and not generic.fused_supported(ABCset, (Local_Proposal, Global_Proposal), 1, gamma_ok=True)):
        return generic.run(_capi.ALGO_GLOBALMCMC, ABCset, num_ite, Initial_theta, Initial_y, Local_Proposal, Global_Proposal,
                           filelocation, global_frequency, 1, "global", seed=seed, device=device, chain0=chain0,
                           record_history=record_history, stats=stats, return_device=return_device, verbose=verbose,
                           state_out=state_out, **generic_kw)
    if generic_kw:
        raise TypeError("unexpected keyword arguments for the fused path: %s" % sorted(generic_kw))
    model = engine.model_descriptor(ABCset)
    local = Local_Proposal.descriptor()
    glob = Global_Proposal.descriptor()
    dev, chains, single = _host.prepare(ABCset, Initial_theta, Initial_y, device, chain0)
    hist = _host.allocate_history(num_ite, chains, record_history)
    mirror = _host.HostMirror(hist) if _host.HostMirror.wanted(hist, single, return_device) else None   # rows leave for the host while the kernels run
    rtc = ABCset.program(_capi.ALGO_GLOBALMCMC) if model.sim_kind == _capi.SIM_USER else None     # compiled.CompiledModel
    engine.run_steps("glabc_globalmcmc_steps", model, local, glob, chains, num_ite - 1, 1, engine.draw_seed(seed),
                     global_frequency, 1, history=None if hist is None else hist[1:], moments=stats,
                     steps_per_launch=steps_per_launch, rtc_program=rtc, mirror=mirror)
    if state_out is not None:
        state_out["chains"] = chains
    return _host.finish(hist, chains, single, filelocation, "global", verbose and single, return_device, mirror=mirror)
