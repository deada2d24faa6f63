"""GLMCMC_NF -- normalizing-flow global proposal (reference: GLMCMC_NFs.py:43-186).
The RealNVP coupling kernels (MFMA) are not built yet."""


def GLMCMC_NF(ABCset, num_ite, Initial_theta, Initial_y, Local_Proposal,
              filelocation, global_frequency, step_size, batch_size, base, Train_step, **kw):
    raise NotImplementedError("GLMCMC_NF: HIP coupling kernels not implemented yet (no CPU fallback by design)")
