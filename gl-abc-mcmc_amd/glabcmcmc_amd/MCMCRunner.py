"""MCMCRunner -- the facade of the reference (MCMCRunner.py:6-121), same constructor and
``run_*`` signatures; extra keyword arguments are forwarded to the sampler functions.

A Model that is a plain Python object (no ``descriptor()``) runs through the split-phase path (generic.py).  With
``graph='auto'`` (the default) one iteration of that path -- the two HIP kernels AND the Model's own callbacks -- is captured
once as a hipGraph and replayed.  A capture freezes everything the callbacks do on the HOST: Python / NumPy scalars read per
call, counters, data-dependent Python branches, copies from pageable host memory.  Callbacks that are pure functions of
their tensor arguments (and of torch's CUDA generator, which graphs handle) are safe -- that is every Model written like
examples/Mixture.py; anything else must pass ``graph=False`` (same chains, launched eagerly).  A capture that cannot be
made at all (a callback that synchronises) falls back to eager launches by itself."""
import os


class MCMCRunner:
    def __init__(self, abc_set, output_dir='./'):
        self.abc_set = abc_set
        self.output_dir = output_dir
        os.makedirs(output_dir, exist_ok=True)

    def _path(self, output_file):
        return None if output_file is None else os.path.join(self.output_dir, output_file)

    def run_global_mcmc(self, num_iterations, initial_theta, initial_y, global_frequency, local_proposal,
                        global_proposal, output_file='global_mcmc_results.csv', **kw):
        from .GlobalMCMC import GlobalMCMC
        return GlobalMCMC(ABCset=self.abc_set, num_ite=num_iterations, Initial_theta=initial_theta,
                          Initial_y=initial_y, Global_Proposal=global_proposal, filelocation=self._path(output_file),
                          global_frequency=global_frequency, Local_Proposal=local_proposal, **kw)

    def run_glmcmc(self, num_iterations, initial_theta, initial_y, global_frequency, local_proposal,
                   importance_proposal, batch_size, output_file='glmcmc_results.csv', **kw):
        from .GLMCMC import GLMCMC
        return GLMCMC(ABCset=self.abc_set, num_ite=num_iterations, Initial_theta=initial_theta, Initial_y=initial_y,
                      Local_Proposal=local_proposal, filelocation=self._path(output_file),
                      global_frequency=global_frequency, Importance_Proposal=importance_proposal,
                      batch_size=batch_size, **kw)

    def run_aglmcmc(self, num_iterations, initial_theta, initial_y, global_frequency, local_proposal,
                    Initial_ISIR_prop, batch_size, step_size, alpha, hat_eps_T, output_file='glmcmc_results.csv', **kw):
        from .AGLMCMC import AGLMCMC
        return AGLMCMC(ABCset=self.abc_set, num_ite=num_iterations, Initial_theta=initial_theta, Initial_y=initial_y,
                       Local_Proposal=local_proposal, Initial_ISIR_prop=Initial_ISIR_prop,
                       filelocation=self._path(output_file), global_frequency=global_frequency, step_size=step_size,
                       batch_size=batch_size, alpha=alpha, hat_eps_T=hat_eps_T, **kw)

    def run_glmala(self, num_iterations, initial_theta, initial_y, global_frequency, importance_proposal,
                   batch_size, tau, num_grad, output_file='glmala_results.csv', **kw):
        from .GLMALA import GLMALA
        return GLMALA(ABCset=self.abc_set, num_ite=num_iterations, Initial_theta=initial_theta, Initial_y=initial_y,
                      tau=tau, num_grad=num_grad, filelocation=self._path(output_file),
                      global_frequency=global_frequency, Importance_Proposal=importance_proposal,
                      batch_size=batch_size, **kw)

    def run_glmcmc_nf(self, num_iterations, initial_theta, initial_y, global_frequency, local_proposal,
                      importance_proposal_base, batch_size, step_size, train_steps,
                      output_file='glmcmc_nf_results.csv', **kw):
        from .GLMCMC_NFs import GLMCMC_NF
        return GLMCMC_NF(ABCset=self.abc_set, num_ite=num_iterations, Initial_theta=initial_theta,
                         Initial_y=initial_y, Local_Proposal=local_proposal, filelocation=self._path(output_file),
                         global_frequency=global_frequency, step_size=step_size, batch_size=batch_size,
                         base=importance_proposal_base, Train_step=train_steps, **kw)
