/*
 * glabc.h -- C ABI of the MI355X-native GL-ABC-MCMC hot path.
 *
 * The reference (caofff/GL-ABC-MCMC, /root/reference) has no FFI of its own:
 * its hot path is the body of three Python loops,
 *     GlobalMCMC  glabcmcmc/GlobalMCMC.py:37-68
 *     GLMCMC      glabcmcmc/GLMCMC.py:58-104
 *     GLMALA      glabcmcmc/GLMALA.py:150-200
 * calling distribution.{DiagGaussian,Uniform,Gamma} (glabcmcmc/distribution.py)
 * and the user's Model callbacks (glabcmcmc/examples/Mixture.py:13-45).  Each
 * entry point below replaces one of those loop bodies (or one distribution /
 * ESJD.py call) for a whole batch of independent chains; the Python package
 * glabcmcmc_amd binds them with ctypes (INTEGRATION.md shows the stub a
 * maintainer of the reference would add).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no C++ / torch types.
 *   - every pointer inside glabc_chains / glabc_run is a DEVICE pointer
 *     (hipMalloc'd or a torch CUDA tensor's data_ptr); descriptor structs
 *     themselves are read on the host at call time and may live on the stack.
 *   - calls are stream-ordered on `stream` (a hipStream_t passed as void*;
 *     NULL = the null stream), allocate nothing, keep no global state, and
 *     are safe to call concurrently on different streams / chain sets.
 *   - return 0 on success or a negative glabc_status; never throw.
 *   - chain state is structure-of-arrays, chain index innermost
 *     ("chain-major"): component j of chain c is at  base[j*stride + c], so a
 *     64-lane wavefront reads 256 contiguous bytes per component.
 *   - random numbers: Philox4x32-10 keyed by the 64-bit seed with counter
 *     (global chain id, step, slot) -- include/glabc_numerics.h -- so results
 *     do not depend on launch geometry, on n_steps per call, or on how chains
 *     are sharded over GPUs (chain0 carries the shard offset).
 */
#ifndef GLABC_H
#define GLABC_H

#if !defined(__HIPCC_RTC__)
#include <stdint.h>
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define GLABC_VERSION 301          /* 0.3.1: bumped whenever a struct of this header changes (the Python binding refuses a library
                                      of another version: a stale .so would misread the argument blocks) */
/* Layout of the random stream (include/glabc_numerics.h): which Philox (counter, slot, word) a draw of (chain, iteration) reads.
 * A checkpoint stores it; resuming on another layout would continue on a different stream.  2 = simulator normals start at an
 * even word (round 2), NF pool rows keyed (refresh << 44) + chain0 * P + r. */
#define GLABC_STREAM_LAYOUT 2
#define GLABC_MAX_DIM 8            /* theta_dim, y_dim, distribution dim */
#define GLABC_MAX_BATCH 16         /* iSIR proposals per step held in registers (one to four lanes per chain) */
#define GLABC_MAX_BATCH_WIDE 4096  /* glabc_glmcmc_steps beyond that: 8 to 64 lanes of a wavefront share a chain's proposals */

typedef enum glabc_status {
    GLABC_OK = 0,
    GLABC_ERR_NULL = -1,           /* a required pointer is NULL */
    GLABC_ERR_DIM = -2,            /* dimension out of range / not compiled in */
    GLABC_ERR_KIND = -3,           /* distribution / simulator kind not supported by this entry point */
    GLABC_ERR_ARG = -4,            /* bad scalar argument (n_steps < 0, batch_size out of range, ...) */
    GLABC_ERR_LAUNCH = -5,         /* hipLaunchKernel failed; see glabc_last_hip_error() */
    GLABC_ERR_NO_DEVICE = -6       /* no HIP device / wrong architecture */
} glabc_status;

/* ---- distributions: glabcmcmc/distribution.py ---------------------------- */
typedef enum glabc_dist_kind {
    GLABC_DIST_DIAG_GAUSS = 0,     /* distribution.py:143-181 */
    GLABC_DIST_UNIFORM = 1,        /* distribution.py:50-86   */
    GLABC_DIST_GAMMA = 2           /* distribution.py:90-137  */
} glabc_dist_kind;

typedef struct glabc_dist {
    int32_t kind;                  /* glabc_dist_kind */
    int32_t dim;
    float p0[GLABC_MAX_DIM];       /* DiagGaussian loc        | Uniform low        | Gamma shape */
    float p1[GLABC_MAX_DIM];       /* DiagGaussian log_scale  | Uniform high       | Gamma rate  */
    float p2[GLABC_MAX_DIM];       /* DiagGaussian exp(log_scale) exactly as the caller's exp returned it
                                      (distribution.py:170,178 recompute it per call; the round trip
                                      exp(log(s)) != s matters for bit parity) | Uniform high-low
                                      | Gamma scale = 1/rate as the reference forms it (float32 division, distribution.py:118,133) */
    float p3[GLABC_MAX_DIM];       /* Gamma scipy.special.gammaln(shape) (a float32 number: the reference keeps Shape as a float32
                                      array, distribution.py:103) | unused otherwise */
    float c0;                      /* DiagGaussian f32(-0.5*dim*log(2*pi)) (distribution.py:171,177)
                                      | Uniform log_prob_val (distribution.py:71) | unused */
} glabc_dist;

/* GLABC_DIST_GAMMA inside the samplers (glabc_glmcmc_steps incl. batch sizes beyond GLABC_MAX_BATCH, glabc_globalmcmc_steps,
 * glabc_propose / glabc_select, glabc_init_weights, glabc_dist_log_prob, the row-wise Model callbacks): accepted as the
 * importance / global proposal and as the Model's prior (not as the local increment, not as simulator noise).  The reference's
 * Gamma is float64 (distribution.py:106-137); the chains' state is float32 here, so
 *   forward:  z_q = glabc_gamma_draw_candidate(shape_q; chain, iteration, candidate j, coordinate q) * scale_q in double
 *             (include/glabc_numerics.h: Marsaglia-Tsang on the chain's Philox stream), theta'_q = (float) z_q,
 *             log q' = (float) sum_q log(pdf(z_q))  -- the density of the double variate, summed in torch.sum's float64 order
 *   log_prob: (float) sum_q log(pdf((double) theta_q)), -inf outside the support or where the pdf underflows (distribution.py:136)
 * which is what the split-phase path does with a Gamma object's callbacks (results rounded to float32 for glabc_select). */

/* ---- the Model callbacks: glabcmcmc/examples/Mixture.py:5-53 ------------- */
typedef enum glabc_sim_kind {
    GLABC_SIM_ABS_GAUSS = 0,       /* y = |theta| + noise, noise ~ DiagGaussian   Mixture.py:13-26 */
    GLABC_SIM_GK = 1,              /* g-and-k order statistics (BASELINE config 4; no counterpart in the reference tree --
                                      the Model is the build's own, glabcmcmc_amd/examples/GK.py): theta = (A, B, g, k),
                                      y = sort_j( A + B (1 + c tanh(g z_j/2)) (1 + z_j^2)^k z_j ), z_j ~ N(0,1), j < y_dim = 8 */
    GLABC_SIM_USER = 2             /* the caller's simulator, C source compiled into the fused kernel at run time (glabc_rtc_compile):
                                      y[y_dim] = glabc_user_simulate(theta[theta_dim], eps[noise.dim] standard normals); only
                                      glabc_rtc_steps and the row-wise Model callbacks accept it */
} glabc_sim_kind;

typedef struct glabc_model {
    int32_t sim_kind;              /* glabc_sim_kind */
    int32_t theta_dim;             /* Mixture.py:8  */
    int32_t y_dim;                 /* Mixture.py:10 */
    float gk_c;                    /* GLABC_SIM_GK: the constant c (0.8) */
    glabc_dist prior;              /* prior_log_prob      Mixture.py:28-31 */
    glabc_dist noise;              /* generate_samples    Mixture.py:19    */
    float y_obs[GLABC_MAX_DIM];    /* discrepancy = ||y - y_obs||_2         Mixture.py:9,33-36 */
    float kern_log_scale;          /* log(f32 epsilon)      calculate_log_kernel, Mixture.py:42-43 */
    float kern_scale;              /* exp(log(f32 epsilon)) */
    float kern_c0;                 /* f32(-0.5*log(2*pi))   */
    float epsilon;                 /* epsilon itself (GLMALA.py:90 uses epsilon**2 in double) */
} glabc_model;

/* ---- chain state ----------------------------------------------------------- */
#define GLABC_FLAG_LOCAL 1u        /* the reference's `local` dirty flag, GLMCMC.py:50,65,100 */
/* GLMALA only.  The reference's tensors change dtype while a chain runs: the gradient estimate is
 * float64 (GLMALA.py:70-72), so the first ACCEPTED MALA move turns Theta_old / y_old into float64
 * tensors (GLMALA.py:43,197-198) and every later torch.cat keeps them float64; log_weight_old takes
 * the dtype the state has when it is first computed (GLMALA.py:152-156) and keeps it.  The two
 * sticky bits below record in which precision the reference would be computing. */
#define GLABC_FLAG_TH64 2u         /* Theta_old, y_old are float64 tensors */
#define GLABC_FLAG_LW64 4u         /* log_weight_old (and hence the iSIR weights) are float64 */
#define GLABC_FLAG_HAS_GRAD 8u     /* grad_logABC_Theta_old is not None, GLMALA.py:146,183-184 */

typedef struct glabc_chains {
    int64_t n_chains;              /* chains in this call (this GPU's shard) */
    int64_t chain0;                /* global id of chain 0 of the shard (Philox counter words 0,1) */
    int64_t stride;                /* elements between components, >= n_chains */
    float* theta;                  /* [theta_dim][stride]  Theta_old  GLMCMC.py:48 */
    float* y;                      /* [y_dim][stride]      y_old      GLMCMC.py:49 */
    float* log_w;                  /* [stride] log_weight_old GLMCMC.py:53-55 (iSIR samplers; else NULL) */
    uint32_t* flags;               /* [stride] GLABC_FLAG_* (iSIR samplers; else NULL) */
    uint32_t* n_moves;             /* [stride] accepted moves, num_acc GLMCMC.py:51,88,101; NULL = not counted */
    /* GLMALA state (NULL for the other samplers): the authoritative copy of the state in double
     * (float32-exact values while the corresponding GLABC_FLAG_*64 bit is clear); theta / y above
     * then receive the float32 casts that Theta_Re records (GLMALA.py:180,200) */
    double* theta64;               /* [theta_dim][stride] */
    double* y64;                   /* [y_dim][stride] */
    double* log_w64;               /* [stride] */
    double* grad;                  /* [theta_dim][stride] grad_logABC_Theta_old, GLMALA.py:146,199 */
} glabc_chains;

/* Per-chain streaming sums for ESJD.py:17-24 and posterior moments, updated
 * once per iteration inside the kernel (so a run needs no history to report
 * them).  tri(d) = d(d+1)/2 entries, row-major upper triangle. */
typedef struct glabc_moments {
    double* sum_theta;             /* [theta_dim][stride]        sum_t theta_t            */
    double* sum_outer;             /* [tri(theta_dim)][stride]   sum_t theta_t theta_t^T  */
    double* sum_jump;              /* [tri(theta_dim)][stride]   sum_t (theta_t - theta_{t-1})(..)^T */
} glabc_moments;

/* Replayed random numbers (tests, common-random-number studies): when
 * glabc_run.tape is non-NULL glabc_glmcmc_steps / glabc_globalmcmc_steps read their draws
 * from here (device arrays covering exactly the call's n_steps) instead of Philox.
 * Layout is chain-outermost, as recorded by tests/golden/make_golden.py.  (GLMALA and the
 * GLMCMC_NF entry points do not take a tape.) */
typedef struct glabc_tape {
    const float* u;                /* [n_chains][n_steps][2]  (branch, accept) f32 in [0,1) */
    const double* r;               /* [n_chains][n_steps]     resampling uniform f64 in [0,1) */
    const float* z;                /* [n_chains][n_steps][n_prop][theta_dim + y_dim] N(0,1) draws:
                                      proposal noise then simulator noise; the local move uses row 0 */
    int32_t n_prop;                /* rows per step on the tape */
    int32_t reserved;
} glabc_tape;

typedef struct glabc_run {
    uint64_t seed;                 /* Philox key */
    uint32_t step0;                /* iteration index of the first step of this call (the reference's
                                      loop variable i starts at 1: GLMCMC.py:58); Philox counter word 2 */
    int32_t n_steps;               /* iterations fused into this launch (K) */
    float global_frequency;        /* GLMCMC.py:59 */
    int32_t batch_size;            /* iSIR proposals N, GLMCMC.py:66: 1..GLABC_MAX_BATCH_WIDE for glabc_glmcmc_steps (no tape above
                                      GLABC_MAX_BATCH), 1..GLABC_MAX_BATCH for the other samplers; ignored by GlobalMCMC */
    float* history;                /* NULL or [n_steps][theta_dim][hist_stride]: Theta_Re rows i=step0.. GLMCMC.py:89,104 */
    int64_t hist_stride;           /* >= n_chains */
    const glabc_moments* moments;  /* NULL or accumulators (same stride as chains) */
    const glabc_tape* tape;        /* NULL = Philox */
    int32_t lanes_per_chain;       /* launch geometry only, never changes results: 0 = choose, or 1 / 2 / 4 lanes cooperating on
                                      one chain's batch_size proposals (8 / 16 / 32 / 64 when batch_size > GLABC_MAX_BATCH) */
    int32_t debug_flags;           /* 0, or GLABC_DEBUG_* bits: execution strategy only, never changes results */
    const uint32_t* step0_device;  /* glabc_propose / glabc_propose_redraw / glabc_select only: NULL, or a DEVICE word holding the
                                      iteration index of this call.  The kernels then read the index from it (Philox counter
                                      word 2) and write the Theta_Re row (index - step0) of `history`, so ONE captured hipGraph
                                      of an iteration can be replayed while the word is incremented between replays */
    const float* global_frequency_per_chain;   /* NULL, or device array [n_chains] that replaces global_frequency chain by
                                      chain -- a hyper-parameter grid (examples/Mixture_hyper.py:24) is then one launch.
                                      glabc_glmcmc_steps / glabc_globalmcmc_steps only */
    int32_t math_mode;             /* GLABC_MATH_EXACT (0, default) or GLABC_MATH_FAST, see below */
    int32_t reserved;
    const struct glabc_draws_out* dump_draws;  /* GLABC_MATH_FAST only: NULL, or where the kernel WRITES the draws it used */
} glabc_run;

/* glabc_run.math_mode -- an OPT-IN variant of glabc_glmcmc_steps (batch size 2..GLABC_MAX_BATCH, theta_dim <= 4, the
 * |theta| + noise simulator; anything else returns GLABC_ERR_ARG).
 *   GLABC_MATH_EXACT  every elementary function is the specified sequence of IEEE operations of include/glabc_numerics.h: the
 *                     CPU checker reproduces every bit.  The default, and what every parity statement in this repository is about.
 *   GLABC_MATH_FAST   Box-Muller with the hardware's v_log_f32 / v_sqrt_f32 / v_sin_f32 / v_cos_f32, the iSIR weights with
 *                     v_exp_f32, the accept uniform's logarithm with v_log_f32, the discrepancy's square root with v_sqrt_f32 and
 *                     the division by the kernel width with v_rcp_f32 (about 1 ulp each).  The normals are then NOT the checker's:
 *                     the run samples the same law from another stream.  Parity is shown the teacher-forced way (SURVEY.md
 *                     appendix A.4): with dump_draws set the kernel records every uniform and normal it used in the glabc_tape
 *                     layout; the CPU checker replays that tape through the EXACT arithmetic and must take the same accept /
 *                     resample decisions except where a decision lies within rounding of its threshold
 *                     (tests/test_hip_parity.py::test_fast_math_*).  Philox, the candidates' arithmetic (IEEE + - x), torch.sum's
 *                     order and the double-precision index are unchanged. */
#define GLABC_MATH_EXACT 0
#define GLABC_MATH_FAST 1
typedef struct glabc_draws_out {   /* device arrays covering exactly the call's n_steps, layout of glabc_tape */
    float* u;                      /* [n_chains][n_steps][2] */
    double* r;                     /* [n_chains][n_steps] */
    float* z;                      /* [n_chains][n_steps][batch_size][theta_dim + y_dim] */
} glabc_draws_out;

/* glabc_run.debug_flags: the iSIR index (GLMCMC.py:7-22) is normally found from float32 reciprocal-multiplied
 * weights and recomputed the reference's way (IEEE divisions, double partial sums) only when the resampling uniform
 * lies within 4e-6 of a partial sum (the two cannot disagree otherwise: the fast partial sums are within 1.3e-6 of
 * the reference's).  This bit takes the reference's path always -- tests use it to show both give the same chains. */
#define GLABC_DEBUG_EXACT_INDEX 1
/* glabc_glmcmc_steps, batch_size 2..GLABC_MAX_BATCH: launches of at most two wavefronts per SIMD run as TEAMS of two wavefronts per
 * 64 chains (csrc/glabc_team.h: one holds the chains and takes the decisions, the other evaluates half of the candidates one
 * iteration ahead).  These bits forbid / force that geometry -- tests use them to show that both give the same chains. */
#define GLABC_DEBUG_NO_TEAM 2
#define GLABC_DEBUG_TEAM 4
/* (Round 3: teams of two / three wavefronts also run the Gamma variant, the g-and-k Model, GlobalMCMC -- glabc_globalmcmc_steps: a
 * helper wavefront draws an iteration's random numbers a chunk of iterations ahead -- and the run-time compiled kernels of
 * glabc_rtc_steps; the same two bits force / forbid the geometry there.) */
/* One lane per chain, launches of at most two wavefronts per SIMD: the library picks the build of the kernels scheduled for
 * instruction-level parallelism; this bit picks the default-schedule build (the one larger launches get) -- so that a test
 * can walk EVERY instantiation with small launches (tests/test_slp_twin.py). */
#define GLABC_DEBUG_DEFAULT_SCHEDULE 8

/* ---- entry points ------------------------------------------------------------ */

/* GLMCMC.py:58-104 -- iSIR global move with probability global_frequency,
 * else random-walk MH local move.  `local` is the zero-mean increment
 * distribution (GLMCMC.py:91), `importance` the absolute proposal (GLMCMC.py:66). */
int glabc_glmcmc_steps(const glabc_model* model, const glabc_dist* local, const glabc_dist* importance,
                       const glabc_chains* chains, const glabc_run* run, void* stream);

/* GlobalMCMC.py:37-68 -- independence MH global move / random-walk MH local move. */
int glabc_globalmcmc_steps(const glabc_model* model, const glabc_dist* local, const glabc_dist* global,
                           const glabc_chains* chains, const glabc_run* run, void* stream);

/* GLMALA.py:150-200 -- iSIR global move (GLMALA.py:151-180) / MALA local move whose drift is the
 * common-random-number central-difference gradient of a synthetic-likelihood log-ABC
 * (numberical_gradient_logABC, GLMALA.py:46-95; Local_proposal_forward :25-44; log_proposal :97-116).
 * tau_sq = tau**2 and eps_sq = epsilon**2 are passed as the caller's double-precision values
 * (Python evaluates them in float, GLMALA.py:43,90).  chains->theta64 / y64 / log_w64 / grad are
 * required; initialise them with glabc_glmala_init. */
typedef struct glabc_mala {
    double tau;                    /* GLMALA.py:118 */
    double tau_sq;                 /* tau ** 2 */
    double eps_sq;                 /* ABCset.epsilon ** 2 */
    int32_t num_grad;              /* simulations per finite-difference side, GLMALA.py:46 `num` */
    int32_t reserved;
} glabc_mala;

int glabc_glmala_steps(const glabc_model* model, const glabc_dist* importance, const glabc_mala* mala,
                       const glabc_chains* chains, const glabc_run* run, void* stream);

/* GLMALA.py:143-149: theta64 / y64 <- theta / y, flags <- LOCAL, no gradient yet. */
int glabc_glmala_init(const glabc_model* model, const glabc_chains* chains, void* stream);

/* ---- RealNVP coupling stack of GLMCMC_NF (GLMCMC_NFs.py:51-61,70-72,96-98) -------------------------
 * theta_dim = 2, hidden width 128 (nf.nets.MLP([1, 128, 128, 2]), GLMCMC_NFs.py:56).  One coupling's
 * parameters are one block of GLABC_NF_COUPLING_FLOATS floats in DEVICE memory:
 *     W2^T [k][i] (128*128, input-major) | W1[128] | b1[128] | (b2[i], W3[0][i], W3[1][i], 0) x 128 | b3[2], 0, 0
 * (W1: Linear(1,128).weight[:,0]; W2: Linear(128,128).weight [out][in]; W3: Linear(128,2).weight; row 0 of the
 * last layer is the shift, row 1 the log-scale: normflows AffineCoupling, param[:, 0::2] / param[:, 1::2]).
 * Rows are chain-major: x[0*n + r], x[1*n + r]. */
#define GLABC_NF_HIDDEN 128
#define GLABC_NF_COUPLING_FLOATS (128 * 128 + 128 + 128 + 4 * 128 + 4)

typedef struct glabc_flow {
    int32_t n_couplings;           /* GLMCMC_NFs.py:51 num_layers (32 there) */
    int32_t hidden;                /* must be GLABC_NF_HIDDEN */
    const float* params;           /* device, [n_couplings][GLABC_NF_COUPLING_FLOATS] */
    float base_loc[2];             /* nf.distributions.base.DiagGaussian(2): loc, log_scale, exp(log_scale) */
    float base_log_scale[2];
    float base_scale[2];
    float base_c0;                 /* f32(-0.5*2*log(2 pi)) */
    int32_t reserved;
} glabc_flow;

/* NF_model.sample(n): z = base(eps) pushed FORWARD through every coupling + swap; log_q = base log-density minus
 * the accumulated log-determinants.  eps[2][n] = the base noise, or NULL to draw it from Philox
 * (key = seed, counter = (row0 + r, 0, 0)). */
int glabc_nf_sample(const glabc_flow* flow, const float* eps, uint64_t seed, int64_t row0, int64_t n_rows,
                    float* z_out, float* log_q, void* stream);

/* NF_model.log_prob(x): x pulled back through the couplings in reverse order (inverse pass) + base log_prob. */
int glabc_nf_log_prob(const glabc_flow* flow, const float* x, int64_t n_rows, float* log_q, void* stream);

/* NF_model.log_prob for the rows listed in idx[0 .. *n_dev - 1] only: x = (theta[idx], theta[stride + idx]) (the chain-major
 * state arrays), result to log_q[idx].  *n_dev is read on the device (no host synchronisation); max_rows bounds it (the grid).
 * Same arithmetic per row as glabc_nf_log_prob. */
int glabc_nf_log_prob_indexed(const glabc_flow* flow, const float* theta, int64_t stride, const int32_t* idx,
                              const int32_t* n_dev, int64_t max_rows, float* log_q, void* stream);

/* NF_model.log_prob(x) AND the base-space point it is evaluated at: z_out[2][n] = x pulled back through every coupling;
 * trace (may be NULL): [n_couplings][n] the conditioner input every coupling saw on the way (what the training step needs to
 * open exactly the ReLU gates the forward evaluation opened). */
int glabc_nf_inverse(const glabc_flow* flow, const float* x, int64_t n_rows, float* z_out, float* log_q, float* trace,
                     void* stream);

/* ---- the training step of GLMCMC_NFs.py:63,112-124 -----------------------------------------------------------------------
 * loss = NF_model.forward_kld(x) = -mean(NF_model.log_prob(x)) over x[2][n_rows] and its gradient with respect to every
 * parameter, differentiated by hand (the reference lets autograd do it, GLMCMC_NFs.py:120-122): grad_params has the layout
 * of flow->params ([n_couplings][GLABC_NF_COUPLING_FLOATS], padding entries 0), grad_base = d/d(loc0, loc1, log_scale0,
 * log_scale1) of the base distribution, *loss is a device scalar.  workspace: device memory of at least
 * glabc_nf_grad_workspace bytes, contents undefined afterwards.  The derivative is that of the float32 evaluation's active
 * set: every ReLU gate is opened exactly where glabc_nf_log_prob's arithmetic opened it (the conditioner inputs of the
 * downward pass are kept, 4 bytes per row and coupling).  Floating-point parity (sums over rows run on the matrix cores in
 * float32, reduced over workgroups in a fixed order): reproducible to the bit from run to run, within 2e-4 of each tensor's
 * largest entry of the checker's double-precision sums over the same active set (tests/test_nf_train.py). */
int glabc_nf_grad_workspace(int32_t n_couplings, int64_t n_rows, int64_t* bytes);
int glabc_nf_grad(const glabc_flow* flow, const float* x, int64_t n_rows, void* workspace, int64_t workspace_bytes,
                  float* grad_params, float* grad_base, float* loss, void* stream);

/* torch.optim.Adam.step() (GLMCMC_NFs.py:63,123: lr 5e-4, weight_decay 1e-5 added to the gradient, no amsgrad) on `count`
 * float32 values, step = 1, 2, ...: exp_avg <- lerp(exp_avg, g, 1 - beta1); exp_avg_sq <- beta2 exp_avg_sq + (1 - beta2) g g;
 * p <- p - lr / (1 - beta1^step) * exp_avg / (sqrt(exp_avg_sq) / sqrt(1 - beta2^step) + eps). */
int glabc_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t count, double lr, double beta1,
                    double beta2, double eps, double weight_decay, int32_t step, void* stream);

/* ---- GLMCMC_NF (GLMCMC_NFs.py:43-186): iSIR against a pool of flow proposals -----------------------------
 * A pool holds P = batch_size*step_size proposals per chain, row r = p*n_chains + c (slice kk of chain c =
 * rows p in [kk*batch_size, (kk+1)*batch_size)), arrays chain-major [dim][P*n_chains].
 *
 * glabc_pool_weights, GLMCMC_NFs.py:73-85 / 128-140: x = generate_samples(theta) with Philox noise keyed by
 * (row_id0 + r), w = exp(prior(theta) + K(x) - log_q), NaN -> 0. */
int glabc_pool_weights(const glabc_model* model, const float* theta, const float* log_q, int64_t n_rows,
                       uint64_t seed, int64_t row_id0, float* x_out, float* w_out, void* stream);

/* One iteration (GLMCMC_NFs.py:90-111,141-152) for every chain: with probability global_frequency the iSIR
 * move against the chain's next pool slice -- weights cat(exp(prior + K - log_q_old), pool_w[slice]) normalised
 * with torch.sum, index by the double running sum (GLMCMC_NFs.py:11-26), kk += 1 -- else the random-walk MH
 * local move (same draws as glabc_glmcmc_steps).  log_q_old[c] = NF_model.log_prob(Theta_old) (glabc_nf_log_prob
 * on chains->theta).  run->n_steps must be 1; run->history (if any) receives the row. */
typedef struct glabc_pool {
    const float* theta;            /* [theta_dim][P*n_chains] */
    const float* x;                /* [y_dim][P*n_chains] */
    const float* w;                /* [P*n_chains] */
    const float* log_q_old;        /* [n_chains] */
    int32_t* kk;                   /* [n_chains] slices already consumed, GLMCMC_NFs.py:86,111 */
    int32_t step_size;             /* slices per pool */
    int32_t reserved;
    /* optional (all three NULL or moved_idx + n_moved set): the chains that moved in this iteration, so that the caller
     * re-evaluates NF_model.log_prob(Theta_old) (GLMCMC_NFs.py:96-98, a pure function of the state and the flow) only for
     * them -- glabc_nf_log_prob_indexed.  The order of the list is unspecified. */
    int32_t* moved_idx;            /* [n_chains] local chain indices, entries 0 .. *n_moved - 1 */
    int32_t* n_moved;              /* device counter, incremented atomically; the caller or n_moved_reset zeroes it */
    int32_t* n_moved_reset;        /* NULL or a counter this call sets to 0 (the one the NEXT iteration will count into) */
} glabc_pool;

int glabc_glmcmc_nf_step(const glabc_model* model, const glabc_dist* local, const glabc_pool* pool,
                         const glabc_chains* chains, const glabc_run* run, void* stream);

/* GLMCMC.py:52-55 -- (re)initialise log_w = prior + log-kernel - q(theta) and set
 * GLABC_FLAG_LOCAL for every chain. */
int glabc_init_weights(const glabc_model* model, const glabc_dist* importance,
                       const glabc_chains* chains, void* stream);

/* ---- split-phase iteration for Models whose callbacks are arbitrary code -------------------------------------------
 * The reference's plug-in API is the duck-typed Model protocol (examples/Mixture.py:5-53): the sampler loops call
 * ABCset.generate_samples / prior_log_prob / calculate_log_kernel (GLMCMC.py:71-74,94-97; GlobalMCMC.py:41-46,57-61).
 * A Model that cannot describe itself as a glabc_model runs one iteration in three phases:
 *
 *   glabc_propose   every random draw of the iteration (same Philox slots as the fused kernels): the branch / accept /
 *                   resampling numbers of each chain and its n_prop candidates theta' with their proposal log-density
 *   the callbacks   generate_samples, prior_log_prob, calculate_log_kernel on the candidate rows -- PyTorch on the same
 *                   device, or anything else that fills y_prop / prior_prop / kern_prop
 *   glabc_select    iSIR weights, torch.sum normalisation, double-precision index (GLMCMC.py:74-88, 7-22) or the MH test
 *                   (GLMCMC.py:96-103; GlobalMCMC.py:44-52,60-67), state update, Theta_Re row, streaming sums
 *
 * Candidate j of chain c is row r = j*n_chains + c of every candidate array (the Model sees one (n_prop*n_chains, dim)
 * row-major batch).  A chain on the local branch uses row j = 0 only (theta' = Theta_old + increment, GLMCMC.py:91); its
 * other rows still hold global candidates and are ignored by glabc_select.  All pointers are device pointers.
 * theta_dim and y_dim are free in glabc_select (loops); a descriptor proposal limits theta_dim to GLABC_MAX_DIM. */
typedef enum glabc_algo {
    GLABC_ALGO_GLMCMC = 0,         /* GLMCMC.py:58-104 */
    GLABC_ALGO_GLOBALMCMC = 1,     /* GlobalMCMC.py:37-68 */
    GLABC_ALGO_GLMALA = 2          /* GLMALA.py:150-200: the iSIR move as GLMCMC; on the local branch the caller evaluates the
                                      MALA proposal (gradient, drift, GLMALA.py:182-193) and puts theta', y', prior', K' in
                                      row c and the proposal terms log_proposal(theta', grad', Theta_old) - log q_forward in
                                      log_q[c]; glabc_select takes the accept decision.  `local` stays clear after an accepted
                                      MALA move (GLMALA.py:195-199) */
} glabc_algo;

typedef struct glabc_step_io {
    int32_t n_prop;                /* candidate rows per chain: batch_size (GLMCMC.py:66) | 1 (GlobalMCMC) */
    int32_t theta_dim;
    int32_t y_dim;
    int32_t noise_dim;             /* standard normals per candidate handed to the simulator (0 = the Model draws its own) */
    /* written by glabc_propose */
    float* theta_prop;             /* [n_prop*n_chains][theta_dim] */
    float* log_q;                  /* [n_prop*n_chains] forward()'s log_p of a global candidate (GLMCMC.py:66, GlobalMCMC.py:40) */
    float* sim_noise;              /* NULL or [n_prop*n_chains][noise_dim]: words DP + i of the candidate's Philox blocks,
                                      DP = theta_dim rounded up to even (the fused kernels' simulator draws) */
    float* log_u;                  /* [n_chains] log(torch.rand(1)), GLMCMC.py:98 */
    double* u_res;                 /* [n_chains] np.random.uniform(0,1), GLMCMC.py:17 */
    int32_t* is_global;            /* [n_chains] bit 0: torch.rand(1) < global_frequency, GLMCMC.py:59; glabc_select sets
                                      bit 1 when the chain moved in this iteration */
    /* written by the Model callbacks, read by glabc_select */
    const float* y_prop;           /* [n_prop*n_chains][y_dim]  generate_samples(theta_prop, 1) */
    const float* prior_prop;       /* [n_prop*n_chains]         prior_log_prob(theta_prop) */
    const float* kern_prop;        /* [n_prop*n_chains]         calculate_log_kernel(y_prop) */
    /* callbacks of the CURRENT state, carried from the iteration in which it was proposed (the reference re-evaluates
     * them every iteration, GLMCMC.py:61-64,97; pure functions of the state) -- maintained by glabc_select */
    float* prior_cur;              /* [n_chains] prior_log_prob(Theta_old) */
    float* kern_cur;               /* [n_chains] calculate_log_kernel(y_old) */
    const float* q_cur;            /* NULL (glabc_select evaluates the descriptor `global` at Theta_old) or [n_chains]
                                      Importance_Proposal.log_prob(Theta_old) from the caller */
    const int32_t* n_valid;        /* NULL, or [n_chains]: GLMCMC.py:67-70 removes the proposals that have a NaN coordinate BEFORE
                                      the Model sees them -- the weight vector is then shorter (torch.sum adds fewer terms, the index
                                      counts the survivors).  Only a proposal given as a callback can produce such rows; the caller
                                      moves them behind the chain's valid candidates (keeping their order) and says how many are
                                      left: glabc_select uses rows 0 .. n_valid[c]-1 of chain c.  Ignored on the local branch. */
} glabc_step_io;

/* Philox slots of the local move's redraws (GLMCMC.py:92-93): round k of a chain reads blocks GLABC_SLOT_REDRAW + k*2 + b */
#define GLABC_SLOT_REDRAW 0x40000000u

/* Draws of iteration run->step0 for every chain (run->n_steps must be 1).  `local` / `global` may be NULL: the caller then
 * fills the corresponding rows of theta_prop / log_q itself (proposal objects without a descriptor). */
int glabc_propose(int algo, const glabc_dist* local, const glabc_dist* global, const glabc_chains* chains,
                  const glabc_run* run, const glabc_step_io* io, void* stream);

/* GLMCMC.py:92-93, `while prior_log_prob(Theta_prop) == 7*log(1e-10): redraw`: for every chain on the local branch whose
 * prior_prop[c] equals the sentinel, row c of theta_prop <- Theta_old + a fresh increment (redraw round `round` >= 1) and
 * n_redrawn (device counter, caller-zeroed) is incremented.  The caller re-evaluates prior_prop and repeats until 0. */
int glabc_propose_redraw(const glabc_dist* local, const glabc_chains* chains, const glabc_run* run, const glabc_step_io* io,
                         int32_t round, int32_t* n_redrawn, void* stream);

/* The decision and the state update of iteration run->step0 (run->n_steps must be 1; run->history, if any, receives the
 * Theta_Re row; run->moments the streaming sums).  chains->log_w / flags are required for GLABC_ALGO_GLMCMC. */
int glabc_select(int algo, const glabc_dist* global, const glabc_chains* chains, const glabc_run* run,
                 const glabc_step_io* io, void* stream);

/* ---- user simulators inside the fused kernel (run-time compilation, hiprtc) ----------------------------------------------
 * The reference's generate_samples is any Python method (examples/Mixture.py:13-26).  Its GPU counterpart is a few lines of C:
 *
 *     GLABC_SIMULATOR void glabc_user_simulate(const float* theta, const float* eps, float* y)
 *     {   // theta[GLABC_THETA_DIM], eps[GLABC_NOISE_DIM] standard normals -> y[GLABC_Y_DIM]
 *         for (int j = 0; j < GLABC_Y_DIM; ++j) y[j] = fabsf(theta[j]) + 0.2236068f * eps[j];
 *     }
 *
 * glabc_rtc_compile builds the library's own sampler code (Philox slots, proposals, prior, discrepancy ||y - y_obs||, Gaussian
 * ABC kernel, iSIR / MH decisions, history, sums -- the source of the built-in kernels, embedded in the library) around that
 * function for one configuration (algorithm, dimensions, batch size) and returns a handle; glabc_rtc_steps is then
 * glabc_glmcmc_steps / glabc_globalmcmc_steps for a glabc_model with sim_kind = GLABC_SIM_USER (noise.dim = the normals the
 * simulator consumes).  Written with + - * / fma, sqrtf and the glabc_* functions of glabc_numerics.h the simulator gives
 * results a CPU build of the same source reproduces bit for bit.  log (may be NULL) receives the compiler's messages. */
typedef struct glabc_rtc_program glabc_rtc_program;
int glabc_rtc_compile(const char* simulator_source, int32_t algo, int32_t theta_dim, int32_t y_dim, int32_t noise_dim,
                      int32_t batch_size, glabc_rtc_program** out, char* log, int64_t log_size);
int glabc_rtc_steps(const glabc_rtc_program* program, const glabc_model* model, const glabc_dist* local, const glabc_dist* global,
                    const glabc_chains* chains, const glabc_run* run, void* stream);
/* generate_samples(theta, 1) of the compiled simulator on n row-major points: theta[n][theta_dim], eps[n][noise_dim] -> y[n][y_dim] */
int glabc_rtc_simulate(const glabc_rtc_program* program, const float* theta, const float* eps, int64_t n, float* y, void* stream);
void glabc_rtc_release(glabc_rtc_program* program);

/* The Model's OTHER callbacks as user source (examples/Mixture.py:28-45 are Python methods too).  The same source string may
 * define, each announced by a #define in that source, any of
 *     #define GLABC_USER_PRIOR 1
 *     GLABC_SIMULATOR float glabc_user_prior_log_prob(const float* theta)               // prior_log_prob, Mixture.py:28-31
 *     #define GLABC_USER_DISCREPANCY 1
 *     GLABC_SIMULATOR float glabc_user_discrepancy(const float* y, const float* y_obs)  // discrepancy, Mixture.py:33-36
 *     #define GLABC_USER_KERNEL 1
 *     GLABC_SIMULATOR float glabc_user_log_kernel(float dis, float scale)               // calculate_log_kernel of a discrepancy,
 *                                                                                      // Mixture.py:38-45; scale = model->kern_scale
 * and the fused kernel calls them where it would evaluate the descriptor's prior / Euclidean discrepancy / Gaussian kernel
 * (a callback that is not announced stays the descriptor's; model->prior must still be a valid placeholder).  glabc_rtc_hooks
 * reports what a program replaces; glabc_rtc_model_rows evaluates prior_log_prob (user priors only: descriptor priors have
 * glabc_model_prior_log_prob), discrepancy or calculate_log_kernel on n row-major points through the very functions the fused
 * kernel calls: in[n][theta_dim] or in[n][y_dim] -> out[n]. */
#define GLABC_RTC_PRIOR_LOG_PROB 0
#define GLABC_RTC_DISCREPANCY 1
#define GLABC_RTC_LOG_KERNEL 2
int glabc_rtc_hooks(const glabc_rtc_program* program, int32_t* user_prior, int32_t* user_discrepancy, int32_t* user_kernel);
int glabc_rtc_model_rows(const glabc_rtc_program* program, const glabc_model* model, int32_t what, const float* in, int64_t n,
                         float* out, void* stream);

/* generate_samples(theta, 1) of a descriptor Model on n row-major points (Mixture.py:13-26 regime n x 1):
 * theta[n][theta_dim], eps[n][y_dim] standard normals (NULL: drawn from Philox(seed; row0 + r, 0, b), pairs of words)
 * -> y[n][y_dim].  The standalone form of the simulator the fused kernels inline. */
int glabc_model_simulate(const glabc_model* model, const float* theta, const float* eps, int64_t n, uint64_t seed,
                         int64_t row0, float* y, void* stream);

/* distribution.py:176-181 / 81-86 / 123-137: log_prob of n row-major points z[n][dim] -> out[n]. */
int glabc_dist_log_prob(const glabc_dist* dist, const float* z, int64_t n, float* out, void* stream);

/* BaseDistribution.forward(n), distribution.py:165-173 / 73-78, drawn on the device: row r (global id row0 + r)
 * takes its dim variates from Philox(seed; id, 0, b), b = 0..ceil(dim/4)-1 -- DiagGaussian: Box-Muller pairs of
 * words (2i, 2i+1), z = loc + exp(log_scale)*eps; Uniform: one float32 uniform per word, z = low + (high-low)*u --
 * and returns forward()'s log_p.  z_out is dimension-major [dim][n]. */
int glabc_dist_forward(const glabc_dist* dist, int64_t n, uint64_t seed, int64_t row0, float* z_out, float* log_p_out,
                       void* stream);

/* Gamma.log_prob, distribution.py:123-137: float64, log(scipy.stats.gamma.pdf(z, shape, scale=1/rate)) with -inf where
 * the pdf is 0 (it underflows earlier than a logpdf would -- reproduced), summed over the dimensions.
 *   pdf_j = exp((shape_j - 1) log(x) - x - gammaln(shape_j)) / scale_j,  x = z_j / scale_j,  0 outside x > 0
 * gammaln[] is supplied by the caller (scipy.special.gammaln(shape) on the host; a constant of the distribution).
 * z[n][dim] row-major float64 -> out[n] float64. */
typedef struct glabc_gamma {
    int32_t dim;
    int32_t reserved;
    double shape[GLABC_MAX_DIM];
    double scale[GLABC_MAX_DIM];   /* 1/rate as the reference forms it (float32 reciprocal, distribution.py:133) */
    double gammaln[GLABC_MAX_DIM];
} glabc_gamma;

int glabc_gamma_log_prob(const glabc_gamma* dist, const double* z, int64_t n, double* out, void* stream);

/* Gamma.forward(n), distribution.py:106-121, drawn on the device: z[r][j] = glabc_gamma_draw(shape_j; Philox(seed; row0 + r,
 * j, attempt)) * scale_j (include/glabc_numerics.h: Marsaglia-Tsang in double; the reference draws scipy.stats.gamma.rvs from
 * NumPy's generator) and log_p[r] = Gamma.log_prob(z[r]).  z_out[n][dim] row-major float64, log_p_out[n] float64. */
int glabc_gamma_forward(const glabc_gamma* dist, int64_t n, uint64_t seed, int64_t row0, double* z_out, double* log_p_out,
                        void* stream);

/* Model callbacks on n row-major points (Mixture.py:28-45): used by the host
 * mirror's Model class and by the parity tests. */
int glabc_model_prior_log_prob(const glabc_model* model, const float* theta, int64_t n, float* out, void* stream);
int glabc_model_discrepancy(const glabc_model* model, const float* y, int64_t n, float* out, void* stream);
int glabc_model_log_kernel(const glabc_model* model, const float* y, int64_t n, float* out, void* stream);

/* ESJD.py:2-25 for a batch of chains: history [n_rows][theta_dim][stride]
 * (chain-major, as written by the samplers) -> esjd[n_chains] =
 * det(D^T D / (n_rows-1))^(1/theta_dim), D = consecutive differences. */
int glabc_esjd(const float* history, int64_t n_rows, int32_t theta_dim, int64_t n_chains, int64_t stride,
               float* esjd_out, void* stream);

/* The same quantity from the streamed jump sums (glabc_moments.sum_jump after n_steps
 * iterations): esjd[c] = det(sum_jump_c / n_steps)^(1/theta_dim).  No history needed. */
int glabc_moments_esjd(const glabc_moments* moments, int64_t n_steps, int32_t theta_dim, int64_t n_chains,
                       int64_t stride, float* esjd_out, void* stream);

/* ---- KernelDensity, the adaptive proposal of AGLMCMC (kernel_density.py:4-177) --------
 * A weighted Gaussian KDE over n_samples centres.  All arrays are device pointers; x is dimension-major.
 *
 * glabc_kde_fit (kernel_density.py:70-94, weighted_std :40-68): weights = w_raw / sum(w_raw) (w_raw NULL: 1/n);
 * bandwidth = h * weighted_std(x, weights, unbiased) with h the Silverman / Scott factor the host formed in Python
 * floats (:24-33), or bw_fixed[dim] (host array) when the caller passes an explicit bandwidth (then h is ignored).
 * Sums are float64 in a fixed order (256 strided partials, then a binary tree) instead of torch.sum's float32
 * cascade; everything else is the reference's float32 arithmetic.  Outputs: weights_out[n], log_w_out[n] =
 * log(weights + 1e-10) (:125), wq_out[n] = rint(weights * 2^40) (the integer weights glabc_kde_sample draws
 * from; the caller turns them into inclusive prefix sums -- integer, so any scan gives the same numbers) and
 * consts_out[dim + 2] = bandwidth[0..dim), sum(log(bandwidth)) (:122), 0.5*dim*log(2 pi) (:121). */
int glabc_kde_fit(const float* x, const float* w_raw, int64_t n_samples, int32_t dim, double h, const float* bw_fixed,
                  float* weights_out, float* log_w_out, int64_t* wq_out, float* consts_out, void* stream);

typedef struct glabc_kde {
    int32_t dim;                   /* 1..GLABC_MAX_DIM */
    int32_t reserved;
    int64_t n_samples;
    const float* x;                /* [dim][n_samples] kernel centres */
    const float* log_w;            /* [n_samples] log(weights + 1e-10) */
    const int64_t* cum_q;          /* [n_samples] inclusive prefix sums of wq (glabc_kde_sample only; else may be NULL) */
    float bandwidth[GLABC_MAX_DIM];
    float sum_log_bw;              /* consts_out[dim] */
    float c_2pi;                   /* consts_out[dim + 1] */
} glabc_kde;

/* KernelDensity.log_prob, kernel_density.py:96-128: out[p] = logsumexp_s( ((-0.5 * sum_d ((pt_pd - x_sd)*(1/bw_d))^2
 * - c_2pi) - sum_log_bw) + log_w_s ), the inner float32 operations in the reference's order; the logsumexp is
 * max + log(sum exp(. - max)) with the sum taken exactly in 2^-40 fixed point (order-independent, so the lanes of a
 * wavefront can share one point).  pts is dimension-major [dim][n_points]. */
int glabc_kde_log_prob(const glabc_kde* kde, const float* pts, int64_t n_points, float* out, void* stream);

/* The same for the points listed in idx[0 .. *n_dev - 1] only: point i = (pts[idx[i]], pts[stride + idx[i]], ...), result to
 * out[idx[i]].  *n_dev is read on the device (no host synchronisation); max_points bounds it (the grid).  AGLMCMC keeps the
 * proposal density of every chain's current state (AGLMCMC.py:137-140, a pure function of the state and the fitted KDE) and
 * refreshes it for the chains that moved. */
int glabc_kde_log_prob_indexed(const glabc_kde* kde, const float* pts, int64_t stride, const int32_t* idx, const int32_t* n_dev,
                               int64_t max_points, float* out, void* stream);

/* KernelDensity.sample, kernel_density.py:130-150: row r (global id row0 + r) picks centre j = first index with
 * cum_q[j] > floor(u * cum_q[n-1]), u the float64 uniform of Philox(seed; id, 0, 0) words 0-1 (torch.multinomial
 * with replacement is the same inverse-CDF draw on torch's generator), and adds bandwidth_d * normal_d (normals from
 * words 2-3 of block 0, then blocks 1..).  out is dimension-major [dim][n]. */
int glabc_kde_sample(const glabc_kde* kde, int64_t n, uint64_t seed, int64_t row0, float* out, void* stream);

/* AGLMCMC.py:199-204 (and :104-109): weights of a proposal pool from its stored discrepancies,
 * w[r] = exp((prior(theta_r) + calculate_log_kernel_dis(dis_r)) - log_q[r]), NaN -> 0.  The threshold is the one in
 * model->kern_* (the caller passes a copy of its descriptor with the annealed eps_hat, Mixture.py:47-53).
 * theta is dimension-major [theta_dim][n]. */
int glabc_kde_train_weights(const glabc_model* model, const float* theta, const float* dis, const float* log_q, int64_t n,
                            float* w_out, void* stream);

/* Test hooks (not part of the sampling API): evaluate include/glabc_numerics.h on the device.
 * op 0 expf, 1 logf, 2 sin(2 pi u), 3 cos(2 pi u) on float bit patterns in[n] -> out[n];
 * op 4 / 5: the two normals of glabc_normal_pair(in[2i], in[2i+1]).  glabc_selftest_sqrt counts the
 * floats with bit pattern in [first_bits, last_bits] whose glabc_sqrtf_normal differs from the
 * exactly rounded square root (*mismatches is a device counter the caller zeroed). */
int glabc_selftest_numerics(int op, const uint32_t* in, uint32_t* out, int64_t n, void* stream);
int glabc_selftest_sqrt(uint32_t first_bits, uint32_t last_bits, uint64_t* mismatches, void* stream);
/* torch.sum's float32 association for rows of any length as glabc_select evaluates it: x[n_rows][n] -> out[n_rows] */
int glabc_selftest_rowsum(const float* x, int32_t n_rows, int32_t n, float* out, void* stream);

int glabc_version(void);
int glabc_stream_layout(void);
const char* glabc_status_string(int status);
int glabc_last_hip_error(void);    /* hipError_t of the most recent failed launch on this thread */

#ifdef __cplusplus
}
#endif
#endif /* GLABC_H */
