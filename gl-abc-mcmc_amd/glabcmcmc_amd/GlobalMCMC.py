"""GlobalMCMC -- independence-MH global move + random-walk MH local move
(reference: GlobalMCMC.py:6-98).  Same positional signature; the loop body
(GlobalMCMC.py:37-68) is the fused gfx950 kernel behind ``glabc_globalmcmc_steps``.
Shapes, return value and keyword-only extras as in ``GLMCMC``."""
from . import _host, engine


def GlobalMCMC(ABCset, num_ite, Initial_theta, Initial_y,
               Global_Proposal, filelocation, global_frequency, Local_Proposal=None, *,
               seed=None, device=None, chain0=0, record_history=True, stats=None, return_device=False,
               steps_per_launch=None, verbose=True, state_out=None):
    if Local_Proposal is None:
        if global_frequency < 1:
            raise ValueError("GlobalMCMC needs Local_Proposal unless global_frequency >= 1 (GlobalMCMC.py:56)")
        Local_Proposal = Global_Proposal
    model = engine.model_descriptor(ABCset)
    local = Local_Proposal.descriptor()
    glob = Global_Proposal.descriptor()
    dev, chains, single = _host.prepare(ABCset, Initial_theta, Initial_y, device, chain0)
    hist = _host.allocate_history(num_ite, chains, record_history)
    engine.run_steps("glabc_globalmcmc_steps", model, local, glob, chains, num_ite - 1, 1, engine.draw_seed(seed),
                     global_frequency, 1, history=None if hist is None else hist[1:], moments=stats,
                     steps_per_launch=steps_per_launch)
    if state_out is not None:
        state_out["chains"] = chains
    return _host.finish(hist, chains, single, filelocation, "global", verbose and single, return_device)
