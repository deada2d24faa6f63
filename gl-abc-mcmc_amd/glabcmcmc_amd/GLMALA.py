"""GLMALA -- iSIR global move + MALA local move (reference: GLMALA.py:118-230).
The fused gfx950 kernel for this sampler is not built yet (SURVEY.md section 8 rows
a4-a6); the function exists so the package surface matches the reference and fails
loudly rather than falling back to a CPU loop."""


def GLMALA(ABCset, num_ite, Initial_theta, Initial_y, tau, num_grad,
           filelocation, global_frequency=0, Importance_Proposal=None, batch_size=None, **kw):
    raise NotImplementedError("GLMALA: HIP kernel not implemented yet (no CPU fallback by design)")
