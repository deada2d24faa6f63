"""AGLMCMC -- adaptive KDE-proposal sampler (reference: AGLMCMC.py).  Out of the hot-path
scope; present so `run_aglmcmc` resolves."""


def AGLMCMC(ABCset, num_ite, Initial_theta, Initial_y, Local_Proposal, Initial_ISIR_prop, filelocation,
            global_frequency, step_size, batch_size, alpha, hat_eps_T, **kw):
    raise NotImplementedError("AGLMCMC is outside the accelerated hot path (SURVEY.md 8(f) f-4)")
