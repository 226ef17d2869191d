/* tests/c_abi/c_abi_smoke.c -- the boundary used from plain C (gcc, not g++): include/sepaihrd_hip.h is
 * the only header, plain pointers and sizes are the only types.  A 2-age problem is built by hand, evaluated
 * for three parameter vectors through the host-pointer entry point, then the same chains are driven through
 * the device-resident sampler entry points for a few iterations.
 *   gcc -std=c99 -Iinclude tests/c_abi/c_abi_smoke.c -L<pkg> -lsepaihrd_hip -Wl,-rpath,<pkg> -lm */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "sepaihrd_hip.h"

#define N_AGE 2
#define T 40
#define P 3

int main(void) {
    double times[T], N[N_AGE] = {2.0e6, 1.0e6}, M[N_AGE * N_AGE] = {6.0, 2.0, 2.0, 3.0}; /* column-major */
    double a[N_AGE] = {1, 1}, h_infec[N_AGE] = {1, 1}, p[N_AGE] = {0.4, 0.2}, h[N_AGE] = {0.02, 0.1};
    double icu[N_AGE] = {0.1, 0.3}, d_H[N_AGE] = {0.02, 0.08}, d_ICU[N_AGE] = {0.2, 0.4}, d_comm[N_AGE] = {0, 0};
    double kappa_end[2] = {10.0, 1000.0}, kappa_val[2] = {1.0, 0.6};
    double x0[11 * N_AGE], obs_H[T * N_AGE], obs_ICU[T * N_AGE], obs_D[T * N_AGE];
    int32_t field[P] = {SEPAIHRD_F_BETA, SEPAIHRD_F_THETA, SEPAIHRD_F_KAPPA_VALUE}, index[P] = {0, 0, 1};
    double lower[P] = {0.01, 0.1, 0.1}, upper[P] = {1.0, 1.0, 1.5};
    uint8_t has_bounds[P] = {1, 1, 1};
    double theta[3 * P] = {0.06, 0.5, 0.6, 0.08, 0.4, 0.7, 5.0 /* reflected */, 0.5, 0.6};
    double loglik[3];
    int32_t status[3], n_acc[3], n_rej[3];
    char err[256] = {0};
    int i, t, rc;

    for (t = 0; t < T; ++t) times[t] = (double)t;
    memset(x0, 0, sizeof x0);
    for (i = 0; i < N_AGE; ++i) {
        x0[1 * N_AGE + i] = 50.0; x0[2 * N_AGE + i] = 30.0; x0[3 * N_AGE + i] = 20.0; x0[4 * N_AGE + i] = 40.0;
        x0[5 * N_AGE + i] = 8.0;  x0[6 * N_AGE + i] = 2.0;
        x0[i] = N[i] - 150.0;
    }
    for (t = 0; t < T; ++t)
        for (i = 0; i < N_AGE; ++i) {
            obs_H[t * N_AGE + i] = floor(3.0 + 0.4 * t * (i + 1));
            obs_ICU[t * N_AGE + i] = floor(1.0 + 0.1 * t * (i + 1));
            obs_D[t * N_AGE + i] = (t == 7 && i == 0) ? NAN : floor(0.05 * t * (i + 1)); /* a missing observation */
        }

    sepaihrd_problem pb;
    memset(&pb, 0, sizeof pb);
    pb.abi_version = SEPAIHRD_ABI_VERSION;
    pb.n_age = N_AGE; pb.n_times = T; pb.n_obs = T; pb.n_beta = 0; pb.n_kappa = 2; pb.n_params = P;
    pb.solver = SEPAIHRD_SOLVER_DOPRI5; pb.constraint_mode = SEPAIHRD_CONSTRAINT_REFLECT; pb.arith = SEPAIHRD_ARITH_FMA;
    pb.times = times; pb.N = N; pb.M = M; pb.a = a; pb.h_infec = h_infec; pb.p = p; pb.h = h; pb.icu = icu;
    pb.d_H = d_H; pb.d_ICU = d_ICU; pb.d_community = d_comm;
    pb.kappa_end_times = kappa_end; pb.kappa_values = kappa_val;
    pb.initial_state = x0; pb.obs_H = obs_H; pb.obs_ICU = obs_ICU; pb.obs_D = obs_D;
    pb.param_field = field; pb.param_index = index; pb.lower = lower; pb.upper = upper; pb.has_bounds = has_bounds;
    pb.beta = 0.06; pb.theta = 0.5; pb.sigma = 1.0 / 3; pb.gamma_p = 0.5; pb.gamma_A = 0.2; pb.gamma_I = 0.2;
    pb.gamma_H = 0.1; pb.gamma_ICU = 1.0 / 14;
    for (i = 0; i < 8; ++i) pb.multipliers[i] = 1.0;
    pb.runup_days = 0.0; pb.seed_exposed = 0.0; pb.abs_err = 1e-6; pb.rel_err = 1e-6; pb.dt_hint = 1.0;

    if (sepaihrd_abi_version() != SEPAIHRD_ABI_VERSION) { fprintf(stderr, "ABI version mismatch\n"); return 2; }
    sepaihrd_ctx* ctx = sepaihrd_create(&pb, -1, err, (int)sizeof err);
    if (!ctx) { fprintf(stderr, "sepaihrd_create: %s\n", err); return 3; }

    rc = sepaihrd_eval_batch(ctx, theta, 3, loglik, status, n_acc, n_rej, NULL, NULL);
    if (rc != SEPAIHRD_OK) { fprintf(stderr, "eval_batch: %s\n", sepaihrd_last_error(ctx)); return 4; }
    for (i = 0; i < 3; ++i)
        printf("chain %d: loglik %.6f status %d steps %d+%d\n", i, loglik[i], status[i], n_acc[i], n_rej[i]);
    /* the third vector's beta = 5.0 reflects into its bounds: 1 - (5 - 0.01 - 4 * 0.99 ... ) stays valid */
    for (i = 0; i < 3; ++i)
        if (status[i] != SEPAIHRD_STATUS_OK || !isfinite(loglik[i]) || n_acc[i] < T - 1) { fprintf(stderr, "unexpected result\n"); return 5; }

    /* a wrong ABI version must be refused with a message, not evaluated */
    pb.abi_version = SEPAIHRD_ABI_VERSION + 7;
    if (sepaihrd_create(&pb, -1, err, (int)sizeof err) != NULL || err[0] == 0) { fprintf(stderr, "version check missing\n"); return 6; }
    pb.abi_version = SEPAIHRD_ABI_VERSION;

    /* device-resident sampler: identity-scaled proposals z = 0 reproduce the current states */
    {
        double cov0[P * P] = {1e-4, 0, 0, 0, 1e-4, 0, 0, 0, 1e-4}, z[3 * P], scale[3] = {1, 1, 1}, cur[3], prop[3];
        uint8_t accept[3] = {1, 0, 1};
        int32_t rows[2] = {0, 1};
        double hist[3 * 2 * P];
        double start[3 * P];
        memcpy(start, theta, sizeof start);
        start[6] = 0.5;  /* inside the bounds, so that z = 0 proposes the state itself */
        memset(z, 0, sizeof z);
        double kept[3 * 2 * P];
        sepaihrd_mh_config cfg;
        sepaihrd_mh* mh;
        memset(&cfg, 0, sizeof cfg);
        cfg.chains = 3; cfg.iterations = 8; cfg.thinning = 1; cfg.adaptation_window = 4;
        cfg.covariance_mode = SEPAIHRD_MH_COV_RUNNING; cfg.reg_eps = 1e-6; cfg.scaling_factor = 2.38 * 2.38 / P;
        mh = sepaihrd_mh_create(ctx, &cfg, start, cov0);
        if (!mh) { fprintf(stderr, "mh_create: %s\n", sepaihrd_last_error(ctx)); return 7; }
        if (sepaihrd_mh_evaluate_current(mh, cur, NULL) != SEPAIHRD_OK || sepaihrd_mh_propose(mh, z, scale, prop, NULL) != SEPAIHRD_OK ||
            sepaihrd_mh_commit(mh, accept) != SEPAIHRD_OK || sepaihrd_mh_adapt(mh, 0.1, 1, 0) != SEPAIHRD_OK ||
            sepaihrd_mh_read_history(mh, rows, 2, hist) != SEPAIHRD_OK || sepaihrd_mh_read_samples(mh, 0, 2, kept) != SEPAIHRD_OK) {
            fprintf(stderr, "sampler entry point failed: %s\n", sepaihrd_last_error(ctx));
            return 8;
        }
        for (i = 0; i < 3; ++i)
            if (cur[i] != prop[i]) { fprintf(stderr, "z = 0 must propose the current state\n"); return 9; }
        if (sepaihrd_mh_history_length(mh) != 2 || sepaihrd_mh_sample_count(mh) != 2 || hist[0] != start[0] || hist[P] != start[0] ||
            memcmp(hist, kept, sizeof hist) != 0) { fprintf(stderr, "history mismatch\n"); return 10; }
        sepaihrd_mh_destroy(mh);
    }
    sepaihrd_destroy(ctx);
    printf("OK\n");
    return 0;
}
