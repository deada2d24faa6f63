"""A user's own Model as C source: all four callbacks of the reference's protocol (examples/Mixture.py:13-45 -- generate_samples,
prior_log_prob, discrepancy, calculate_log_kernel) compiled INTO the fused kernel at run time (``CompiledModel``, hiprtc).

The arithmetic is the reference's example again -- y = |theta| + N(0, 0.05 I), prior N(0, I), Gaussian ABC kernel on the
Euclidean distance to y_obs = (1.5, 1.5) -- written by hand, so its posterior is known in closed form
and the run can be checked; a real user would write their own simulator, prior, distance and kernel in the same four
functions.  Every sampler takes the object: the fused GLMCMC / GlobalMCMC kernels run the compiled callbacks, GLMALA, the
wide batches and the pool samplers reach them through the Model protocol's methods.

    python -m glabcmcmc_amd.examples.CompiledUserModel        # 65 536 chains through MCMCRunner.run_glmcmc, fused
"""
import torch

SOURCE = """
GLABC_SIMULATOR void glabc_user_simulate(const float* theta, const float* eps, float* y)
{   /* generate_samples, Mixture.py:13-26: theta[GLABC_THETA_DIM], eps[GLABC_NOISE_DIM] standard normals -> y[GLABC_Y_DIM] */
    for (int j = 0; j < GLABC_Y_DIM; ++j) y[j] = fabsf(theta[j]) + 0.2236068f * eps[j];
}
#define GLABC_USER_PRIOR 1
GLABC_SIMULATOR float glabc_user_prior_log_prob(const float* theta)
{   /* prior_log_prob, Mixture.py:28-31: N(0, I) */
    float s = 0.0f;
    for (int j = 0; j < GLABC_THETA_DIM; ++j) s += theta[j] * theta[j];
    return -0.9189385f * GLABC_THETA_DIM - 0.5f * s;
}
#define GLABC_USER_DISCREPANCY 1
GLABC_SIMULATOR float glabc_user_discrepancy(const float* y, const float* y_obs)
{   /* discrepancy, Mixture.py:33-36 */
    float s = 0.0f;
    for (int j = 0; j < GLABC_Y_DIM; ++j) s += (y[j] - y_obs[j]) * (y[j] - y_obs[j]);
    return sqrtf(s);
}
#define GLABC_USER_KERNEL 1
GLABC_SIMULATOR float glabc_user_log_kernel(float dis, float scale)
{   /* calculate_log_kernel, Mixture.py:38-45: N(0, scale^2) density of the discrepancy */
    const float e = dis / scale;
    return -0.9189385f - glabc_logf(scale) - 0.5f * (e * e);
}
"""


def build(epsilon=0.05):
    from .. import distribution
    from ..compiled import CompiledModel
    return CompiledModel(theta_dim=2, y_dim=2, simulator_source=SOURCE, prior=distribution.DiagGaussian(2, torch.zeros(2), torch.zeros(2)),
                         y_obs=[1.5, 1.5], epsilon=epsilon)


def analytic_second_moment(epsilon=0.05):
    v = 0.05 + epsilon ** 2
    return (1.5 / (1 + v)) ** 2 + v / (1 + v)


def main(n_chains=65536, num_ite=600, burn=200):
    from .. import distribution, engine
    from ..MCMCRunner import MCMCRunner
    model = build()
    lp = distribution.DiagGaussian(2, torch.zeros(2), torch.log(torch.tensor([0.35, 0.35])))
    ip = distribution.DiagGaussian(2, torch.zeros(2), torch.zeros(2))
    theta0 = torch.zeros(n_chains, 2)
    y0 = model.generate_samples(theta0)
    state = {}
    runner = MCMCRunner(model)
    runner.run_glmcmc(burn + 1, theta0, y0, 0.9, lp, ip, 5, output_file=None, record_history=False, verbose=False, seed=1, state_out=state)
    chains = state["chains"]
    stats = engine.Moments(n_chains, 2, engine.require_device())
    runner.run_glmcmc(num_ite + 1, chains.theta_rows().cpu(), chains.y.t().contiguous().cpu(), 0.9, lp, ip, 5, output_file=None,
                      record_history=False, stats=stats, verbose=False, seed=2)
    got = stats.second_moment().diagonal(dim1=1, dim2=2).mean(0).tolist()
    print("E theta^2 = %s   (stationary value %.4f)" % (got, analytic_second_moment()))
    return got


if __name__ == "__main__":
    main()
