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
"""
from . import _host, engine


def GLMCMC(ABCset, num_ite, Initial_theta, Initial_y, Local_Proposal,
           filelocation, global_frequency=0, Importance_Proposal=None, batch_size=None, *,
           seed=None, device=None, chain0=0, record_history=True, stats=None, return_device=False,
           steps_per_launch=None, verbose=True, state_out=None):
    if Importance_Proposal is None or batch_size is None:
        raise ValueError("GLMCMC needs Importance_Proposal and batch_size (GLMCMC.py:54,66)")
    model = engine.model_descriptor(ABCset)
    local = Local_Proposal.descriptor()
    imp = Importance_Proposal.descriptor()
    dev, chains, single = _host.prepare(ABCset, Initial_theta, Initial_y, device, chain0)
    hist = _host.allocate_history(num_ite, chains, record_history)
    engine.init_weights(model, imp, chains)                        # GLMCMC.py:52-55
    engine.run_steps("glabc_glmcmc_steps", model, local, imp, chains, num_ite - 1, 1, engine.draw_seed(seed),
                     global_frequency, batch_size, history=None if hist is None else hist[1:], moments=stats,
                     steps_per_launch=steps_per_launch)
    if state_out is not None:
        state_out["chains"] = chains
    return _host.finish(hist, chains, single, filelocation, "glmcmc", verbose and single, return_device)
