/* oracle/oracle_capi.h -- TEST INFRASTRUCTURE.  Flat C view of the CPU oracle
 * (oracle/sepaihrd_oracle.hpp) for ctypes.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load liboracle.so. */
#ifndef SEPAIHRD_ORACLE_CAPI_H
#define SEPAIHRD_ORACLE_CAPI_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_problem {
    int32_t n, n_times, n_obs, n_beta, n_kappa, n_params, solver, constraint_mode;
    const double *times, *N, *M; /* M column-major n x n */
    const double *a, *h_infec, *p, *h, *icu, *d_H, *d_ICU, *d_community;
    const double *beta_end_times, *beta_values;   /* n_beta (may be 0) */
    const double *kappa_end_times, *kappa_values; /* n_kappa >= 1, baseline first */
    const double *initial_state;                  /* 11 n */
    const double *obs_H, *obs_ICU, *obs_D;        /* n_obs x n row-major */
    const double *lower, *upper, *sigmas;         /* n_params; lower = NaN -> no bounds entry */
    const char *param_names;                      /* '\n'-joined, n_params names */
    const char *npi_names;                        /* '\n'-joined, n_kappa-1 names */
    double beta, theta, sigma, gamma_p, gamma_A, gamma_I, gamma_H, gamma_ICU;
    double multipliers[8]; /* E0,P0,A0,I0,H0,ICU0,R0,D0 */
    double runup_days, seed_exposed;
    double abs_err, rel_err, dt_hint;
} oracle_problem;

void *oracle_create(const oracle_problem *pb, char *err, int errlen);
void oracle_destroy(void *h);
void oracle_set_constraint_mode(void *h, int mode);
void oracle_set_max_attempts(void *h, long max_attempts);

/* theta: B x P chain-major.  traj (nullable): B x T x 11n.  ll_parts (nullable): B x 3. */
int oracle_eval_batch(void *h, const double *theta, int B, double *loglik, int32_t *status,
                      int32_t *n_accept, int32_t *n_reject, double *ll_parts, double *traj,
                      int nthreads);
/* RHS with the model updated by theta (theta may be NULL -> base parameters). */
int oracle_rhs(void *h, const double *theta, const double *x, double t, double *dxdt);
double oracle_beta_kappa(void *h, const double *theta, double t, double *beta, double *kappa);
double oracle_poisson_loglik(const double *sim, const double *obs, int rows, int cols);
int oracle_apply_constraints(void *h, int mode, const double *in, double *out);
/* chain b: mt19937(seed0 + b), theta_b = constrain(base + sigma * N(0,1)), reflect or clamp per mode */
int oracle_jitter_draws(void *h, int mode, const double *base, uint32_t seed0, int B, double *out);
uint64_t oracle_cache_hash(const double *p, int size);
/* n standard normals from mt19937(seed) with ONE persistent std::normal_distribution */
void oracle_std_normals(uint32_t seed, int count, double *out);
/* Adaptive Metropolis, one chain. samples: (iterations/thinning + 1) x P capacity given by caller.
 * two_pass_covariance: 0 = covariance refresh from running co-moments (oracle::RunningMoments), 1 = the literal two
 * passes over the whole chain history of recomputeFullCovariance (MetropolisHastingsSampler.cpp:168-199). */
int oracle_mh(void *h, int iterations, int burn_in, int adaptation_period, int thinning,
              double reg_eps, double target_acc, int adapt_scale, const double *x0, uint32_t seed,
              double *best, double *best_value, int32_t *accepted, double *final_scale,
              unsigned char *accept_trace, double *samples, double *sample_values,
              int32_t *n_samples, double *final_cov, int two_pass_covariance);
/* Posterior ensemble summaries from a FIXED initial state (SimulationRunner::runSimulation):
 * ppc [6][n_probs][Tp][n], sero [n_probs][T], status [S]; returns Tp, n_valid via pointers. */
int oracle_ensemble(void *h, const double *theta, int S, const double *probs, int n_probs, double *ppc,
                    double *sero, int32_t *status, int32_t *n_valid, int nthreads);
/* HillClimbingOptimizer restated (seeded, virtual threads).  trace: [iterations] current logL. */
int oracle_hc(void *h, int iterations, int cloud_size_multiplier, int threads, const double *x0, uint32_t seed,
              double *best, double *best_value, double *final_cov, double *trace, long *evaluations);
/* ParticleSwarmOptimization restated (seeded; one evaluation at a time).  cfg: 17 doubles, order in oracle_capi.cpp.
 * x0 nullable (no warm-start particle).  trace: [iterations] global best.  Returns 2 when an integration throws. */
int oracle_pso(void *h, const double *cfg, const double *x0, uint32_t seed, double *best, double *best_value,
               double *final_cov, double *trace, long *evaluations);
/* SEPAIHRDGradientObjectiveFunction::evaluate_with_gradient restated; returns 0, or 2 when an integration throws */
int oracle_gradient(void *h, const double *theta, double epsilon, double *value, double *grad);
/* NUTSSampler restated (seeded) over the finite-difference gradient objective; returns the number of samples or -2 */
int oracle_nuts(void *h, int iterations, int adaptation_window, double delta_target, int max_tree_depth, double fd_epsilon,
                int constraint_mode, const double *theta0, uint32_t seed, double *samples, double *values,
                double *eps_trace, int32_t *depth_trace, double *best, double *best_value, long *gradient_calls);
/* ModelCalibrator restated: HC phase (clamp) -> covariance conditioning -> one MH chain (reflect).
 * samples capacity: (mh_iterations / thinning + 1) x P.  All outputs nullable except best/best_value. */
int oracle_calibrate(void *h, int hc_iterations, int cloud_size_multiplier, int threads, uint32_t hc_seed,
                     int mh_iterations, int burn_in, int adaptation_period, int thinning, uint32_t mh_seed,
                     const double *x0, double *best, double *best_value, double *initial_value,
                     double *phase1_best_value, double *phase2_cov, unsigned char *accept_trace,
                     double *samples, double *sample_values, double *mcmc_objective_values, int32_t *n_samples);
/* ModelCalibrator.cpp:93-131 on a given covariance (row-major P x P) with the handle's sigmas */
int oracle_condition_covariance(void *h, const double *cov, double *out);
/* BASELINE config 0 plumbing (CPU only): AgeSIRModel RHS and a Dopri5 run.  C row-major n x n, state [S,I,R] x n. */
void oracle_sir_rhs(int n, const double *N, const double *C, const double *gamma, double q, double scale_C,
                    const double *state, double *deriv);
int oracle_sir_simulate(int n, const double *N, const double *C, const double *gamma, double q, double scale_C,
                        const double *init, const double *times, int n_times, double abs_err, double rel_err,
                        double *traj, int32_t *n_accept, int32_t *n_reject);
/* model fields after updateModelParameters(theta): [beta, theta, sigma, gamma_p, gamma_A, gamma_I, gamma_H, gamma_ICU,
 * a[n], h_infec[n], p[n], h[n], icu[n], d_H[n], d_ICU[n], d_community[n], beta_values[n_beta], kappa_values[n_kappa]];
 * returns the number of doubles written, or -1 when the update throws */
int oracle_model_parameters(void *h, const double *theta, double *out);
/* trajectories of S samples from the problem's initial state AS GIVEN (SimulationRunner::runSimulation):
 * traj [S][T][11 n], status [S] */
int oracle_simulate_samples(void *h, const double *theta, int S, double *traj, int32_t *status, int nthreads);
/* returns the number of indices written (ResultAggregator.cpp:246-266) */
int oracle_ppc_select(int n_samples, int num_for_ppc, uint32_t seed, int32_t *out);
int oracle_num_threads(void);

#ifdef __cplusplus
}
#endif
#endif
