// oracle/oracle_capi.cpp -- TEST INFRASTRUCTURE (see sepaihrd_oracle.hpp header).
#include "oracle_capi.h"
#include "sepaihrd_oracle.hpp"
#include <cstdio>
#include <cstring>
#include <sstream>
#if defined(_OPENMP)
#include <omp.h>
#endif

namespace {
std::vector<std::string> split_lines(const char* s) {
    std::vector<std::string> out;
    if (!s) return out;
    std::stringstream ss(s);
    std::string line;
    while (std::getline(ss, line, '\n'))
        if (!line.empty()) out.push_back(line);
    return out;
}
std::vector<double> vec(const double* p, int n) {
    return p ? std::vector<double>(p, p + n) : std::vector<double>();
}
struct Handle {
    oracle::Problem pb;
};
}  // namespace

extern "C" {

void* oracle_create(const oracle_problem* q, char* err, int errlen) {
    try {
        auto* h = new Handle();
        oracle::Problem& pb = h->pb;
        oracle::Params& P = pb.base;
        const int n = q->n;
        P.n = n;
        P.N = vec(q->N, n);
        P.M = vec(q->M, n * n);
        P.a = vec(q->a, n); P.h_infec = vec(q->h_infec, n); P.p = vec(q->p, n); P.h = vec(q->h, n);
        P.icu = vec(q->icu, n); P.d_H = vec(q->d_H, n); P.d_ICU = vec(q->d_ICU, n);
        P.d_community = vec(q->d_community, n);
        P.beta = q->beta; P.theta = q->theta; P.sigma = q->sigma; P.gamma_p = q->gamma_p;
        P.gamma_A = q->gamma_A; P.gamma_I = q->gamma_I; P.gamma_H = q->gamma_H; P.gamma_ICU = q->gamma_ICU;
        P.beta_end_times = vec(q->beta_end_times, q->n_beta);
        P.beta_values = vec(q->beta_values, q->n_beta);
        P.kappa_end_times = vec(q->kappa_end_times, q->n_kappa);
        P.kappa_values = vec(q->kappa_values, q->n_kappa);
        P.E0_multiplier = q->multipliers[0]; P.P0_multiplier = q->multipliers[1];
        P.A0_multiplier = q->multipliers[2]; P.I0_multiplier = q->multipliers[3];
        P.H0_multiplier = q->multipliers[4]; P.ICU0_multiplier = q->multipliers[5];
        P.R0_multiplier = q->multipliers[6]; P.D0_multiplier = q->multipliers[7];
        P.runup_days = q->runup_days; P.seed_exposed = q->seed_exposed;
        oracle::Model check(P);  // validates sizes
        pb.time_points = vec(q->times, q->n_times);
        pb.initial_state = vec(q->initial_state, oracle::NUM_COMPARTMENTS * n);
        pb.num_obs_rows = q->n_obs;
        pb.obs_H = vec(q->obs_H, q->n_obs * n);
        pb.obs_ICU = vec(q->obs_ICU, q->n_obs * n);
        pb.obs_D = vec(q->obs_D, q->n_obs * n);
        pb.solver = q->solver == 1 ? oracle::CASH_KARP54 : oracle::DOPRI5;
        pb.abs_err = q->abs_err; pb.rel_err = q->rel_err; pb.dt_hint = q->dt_hint;
        pb.pm.names = split_lines(q->param_names);
        pb.pm.npi_names = split_lines(q->npi_names);
        if ((int)pb.pm.names.size() != q->n_params) throw std::invalid_argument("param_names count");
        if ((int)pb.pm.npi_names.size() != q->n_kappa - 1) throw std::invalid_argument("npi_names count");
        for (int i = 0; i < q->n_params; ++i) {
            pb.pm.sigmas[pb.pm.names[i]] = q->sigmas ? q->sigmas[i] : 0.0;
            if (q->lower && q->upper && !std::isnan(q->lower[i]))
                pb.pm.bounds[pb.pm.names[i]] = {q->lower[i], q->upper[i]};
        }
        pb.pm.mode = q->constraint_mode == 1 ? oracle::MCMC_REFLECT : oracle::OPTIMIZATION_CLAMP;
        return h;
    } catch (const std::exception& e) {
        if (err && errlen > 0) std::snprintf(err, errlen, "%s", e.what());
        return nullptr;
    }
}

void oracle_destroy(void* h) { delete static_cast<Handle*>(h); }

void oracle_set_constraint_mode(void* h, int mode) {
    static_cast<Handle*>(h)->pb.pm.mode = mode == 1 ? oracle::MCMC_REFLECT : oracle::OPTIMIZATION_CLAMP;
}

void oracle_set_max_attempts(void* h, long max_attempts) { static_cast<Handle*>(h)->pb.max_attempts = max_attempts; }

int oracle_eval_batch(void* hv, const double* theta, int B, double* loglik, int32_t* status,
                      int32_t* n_accept, int32_t* n_reject, double* ll_parts, double* traj,
                      int nthreads) {
    const Handle* h = static_cast<Handle*>(hv);
    const int P = (int)h->pb.pm.names.size();
    const size_t traj_len = h->pb.time_points.size() * oracle::NUM_COMPARTMENTS * h->pb.base.n;
    (void)nthreads;
#if defined(_OPENMP)
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads > 0 ? nthreads : 1)
#endif
    for (int b = 0; b < B; ++b) {
        std::vector<double> th(theta + (size_t)b * P, theta + (size_t)(b + 1) * P);
        oracle::EvalInfo info;
        std::vector<double> tr;
        double v = oracle::objective(h->pb, th, &info, traj ? &tr : nullptr);
        loglik[b] = v;
        if (status) status[b] = info.status;
        if (n_accept) n_accept[b] = (int32_t)info.steps.accepted;
        if (n_reject) n_reject[b] = (int32_t)info.steps.rejected;
        if (ll_parts) {
            ll_parts[3 * b + 0] = info.ll_hosp;
            ll_parts[3 * b + 1] = info.ll_icu;
            ll_parts[3 * b + 2] = info.ll_deaths;
        }
        if (traj) {
            if (tr.size() == traj_len) std::memcpy(traj + (size_t)b * traj_len, tr.data(), traj_len * 8);
            else std::fill(traj + (size_t)b * traj_len, traj + (size_t)(b + 1) * traj_len, 0.0);
        }
    }
    return 0;
}

static int updated_model(const Handle* h, const double* theta, oracle::Model& m) {
    if (!theta) return 0;
    const int P = (int)h->pb.pm.names.size();
    try {
        h->pb.pm.updateModelParameters(std::vector<double>(theta, theta + P), m);
    } catch (...) { return 1; }
    return 0;
}

int oracle_rhs(void* hv, const double* theta, const double* x, double t, double* dxdt) {
    const Handle* h = static_cast<Handle*>(hv);
    oracle::Model m(h->pb.base);
    if (updated_model(h, theta, m)) return 1;
    m.rhs(x, dxdt, t);
    return 0;
}

double oracle_beta_kappa(void* hv, const double* theta, double t, double* beta, double* kappa) {
    const Handle* h = static_cast<Handle*>(hv);
    oracle::Model m(h->pb.base);
    updated_model(h, theta, m);
    const double b = m.beta(t), k = m.kappa(t);
    if (beta) *beta = b;
    if (kappa) *kappa = k;
    return b * k;
}

double oracle_poisson_loglik(const double* sim, const double* obs, int rows, int cols) {
    return oracle::poisson_loglik(sim, obs, rows, cols);
}

int oracle_apply_constraints(void* hv, int mode, const double* in, double* out) {
    const Handle* h = static_cast<Handle*>(hv);
    oracle::ParameterManager pm = h->pb.pm;
    pm.mode = mode == 1 ? oracle::MCMC_REFLECT : oracle::OPTIMIZATION_CLAMP;
    const int P = (int)pm.names.size();
    std::vector<double> c = pm.applyConstraints(std::vector<double>(in, in + P));
    std::copy(c.begin(), c.end(), out);
    return 0;
}

int oracle_jitter_draws(void* hv, int mode, const double* base, uint32_t seed0, int B, double* out) {
    const Handle* h = static_cast<Handle*>(hv);
    oracle::ParameterManager pm = h->pb.pm;
    pm.mode = mode == 1 ? oracle::MCMC_REFLECT : oracle::OPTIMIZATION_CLAMP;
    const int P = (int)pm.names.size();
    std::vector<double> b0(base, base + P);
    for (int b = 0; b < B; ++b) {
        std::vector<double> c = oracle::jitter_draw(pm, b0, seed0 + (uint32_t)b);
        std::copy(c.begin(), c.end(), out + (size_t)b * P);
    }
    return 0;
}

uint64_t oracle_cache_hash(const double* p, int size) { return oracle::cache_hash(p, size); }

void oracle_std_normals(uint32_t seed, int count, double* out) {
    std::mt19937 rng(seed);
    std::normal_distribution<double> normal(0.0, 1.0);
    for (int i = 0; i < count; ++i) out[i] = normal(rng);
}

int oracle_mh(void* hv, int iterations, int burn_in, int adaptation_period, int thinning,
              double reg_eps, double target_acc, int adapt_scale, const double* x0, uint32_t seed,
              double* best, double* best_value, int32_t* accepted, double* final_scale,
              unsigned char* accept_trace, double* samples, double* sample_values,
              int32_t* n_samples, double* final_cov, int two_pass_covariance) {
    const Handle* h = static_cast<Handle*>(hv);
    oracle::Problem pb = h->pb;  // private copy: the sampler flips the constraint mode
    const int P = (int)pb.pm.names.size();
    oracle::MHSettings cfg;
    cfg.iterations = iterations; cfg.burn_in = burn_in; cfg.adaptation_period = adaptation_period;
    cfg.thinning = std::max(1, thinning); cfg.regularization_epsilon = reg_eps;
    cfg.target_acceptance_rate = target_acc; cfg.adapt_scale = adapt_scale != 0;
    cfg.two_pass_covariance = two_pass_covariance != 0;
    oracle::Objective f = [&pb](const std::vector<double>& th) {
        oracle::EvalInfo info;
        double v = oracle::objective(pb, th, &info, nullptr);
        if (info.status >= 2) throw std::runtime_error("SimulationException");
        return v;
    };
    oracle::MHResult r = oracle::metropolis_hastings(cfg, std::vector<double>(x0, x0 + P), f, pb.pm, seed);
    if (best) std::copy(r.best.begin(), r.best.end(), best);
    if (best_value) *best_value = r.best_value;
    if (accepted) *accepted = r.accepted;
    if (final_scale) *final_scale = r.final_scale;
    if (accept_trace) std::copy(r.accept_trace.begin(), r.accept_trace.end(), accept_trace);
    if (n_samples) *n_samples = (int32_t)r.samples.size();
    if (samples)
        for (size_t i = 0; i < r.samples.size(); ++i)
            std::copy(r.samples[i].begin(), r.samples[i].end(), samples + i * P);
    if (sample_values) std::copy(r.sample_values.begin(), r.sample_values.end(), sample_values);
    if (final_cov) std::copy(r.final_cov.begin(), r.final_cov.end(), final_cov);
    return 0;
}

int oracle_ensemble(void* hv, const double* theta, int S, const double* probs, int n_probs, double* ppc,
                    double* sero, int32_t* status, int32_t* n_valid, int nthreads) {
    auto* h = static_cast<Handle*>(hv);
    const oracle::EnsembleSummary r = oracle::ensemble_summaries(
        h->pb, theta, S, std::vector<double>(probs, probs + n_probs), nthreads > 0 ? nthreads : 1);
    std::copy(r.ppc.begin(), r.ppc.end(), ppc);
    if (sero) std::copy(r.sero.begin(), r.sero.end(), sero);
    if (status) std::copy(r.status.begin(), r.status.end(), status);
    if (n_valid) *n_valid = r.n_valid;
    return r.Tp;
}

int oracle_hc(void* hv, int iterations, int cloud_size_multiplier, int threads, const double* x0, uint32_t seed,
              double* best, double* best_value, double* final_cov, double* trace, long* evaluations) {
    auto* h = static_cast<Handle*>(hv);
    const int P = static_cast<int>(h->pb.pm.names.size());
    oracle::HCSettings cfg;
    cfg.iterations = iterations; cfg.cloud_size_multiplier = cloud_size_multiplier; cfg.threads = threads;
    oracle::ParameterManager pm = h->pb.pm;
    pm.mode = oracle::OPTIMIZATION_CLAMP;  // ModelCalibrator.cpp:62-66
    oracle::Problem pb = h->pb;
    pb.pm.mode = oracle::OPTIMIZATION_CLAMP;
    auto fn = [&](const std::vector<double>& p) {
        oracle::EvalInfo info;
        const double v = oracle::objective(pb, p, &info);
        if (info.status >= 2) throw std::runtime_error("SimulationException");
        return v;
    };
    const oracle::HCResult r = oracle::hill_climbing(cfg, std::vector<double>(x0, x0 + P), fn, pm, seed);
    std::copy(r.best.begin(), r.best.end(), best);
    *best_value = r.best_value;
    std::copy(r.final_cov.begin(), r.final_cov.end(), final_cov);
    if (trace) std::copy(r.current_trace.begin(), r.current_trace.end(), trace);
    if (evaluations) *evaluations = r.evaluations;
    return 0;
}

// cfg: [iterations, swarm_size, max_stagnation, omega_start, omega_end, c1_initial, c1_final, c2_initial, c2_final,
//       variant, topology, use_opposition_learning, use_adaptive_parameters, restart_threshold, quantum_beta,
//       levy_alpha, deferred_personal_bests]
int oracle_pso(void* hv, const double* cfg_values, const double* x0, uint32_t seed, double* best, double* best_value,
               double* final_cov, double* trace, long* evaluations) {
    auto* h = static_cast<Handle*>(hv);
    const int P = static_cast<int>(h->pb.pm.names.size());
    oracle::PSOSettings cfg;
    const double* c = cfg_values;
    cfg.iterations = int(c[0]); cfg.swarm_size = int(c[1]); cfg.max_stagnation = int(c[2]);
    cfg.omega_start = c[3]; cfg.omega_end = c[4]; cfg.c1_initial = c[5]; cfg.c1_final = c[6];
    cfg.c2_initial = c[7]; cfg.c2_final = c[8]; cfg.variant = int(c[9]); cfg.topology = int(c[10]);
    cfg.use_opposition_learning = c[11] != 0.0; cfg.use_adaptive_parameters = c[12] != 0.0;
    cfg.restart_threshold = c[13]; cfg.quantum_beta = c[14]; cfg.levy_alpha = c[15];
    cfg.deferred_personal_bests = c[16] != 0.0;
    oracle::Problem pb = h->pb;
    pb.pm.mode = oracle::OPTIMIZATION_CLAMP;
    auto fn = [&](const std::vector<double>& p) {
        oracle::EvalInfo info;
        const double v = oracle::objective(pb, p, &info);
        if (info.status >= 2) throw std::runtime_error("SimulationException");
        return v;
    };
    std::vector<double> start;
    if (x0) start.assign(x0, x0 + P);
    try {
        const oracle::PSOResult r = oracle::particle_swarm(cfg, x0 ? &start : nullptr, fn, pb.pm, seed);
        std::copy(r.best.begin(), r.best.end(), best);
        *best_value = r.best_value;
        if (final_cov) std::copy(r.final_cov.begin(), r.final_cov.end(), final_cov);
        if (trace) std::copy(r.best_trace.begin(), r.best_trace.end(), trace);
        if (evaluations) *evaluations = r.evaluations;
    } catch (const std::exception&) { return 2; }
    return 0;
}

int oracle_gradient(void* hv, const double* theta, double epsilon, double* value, double* grad) {
    auto* h = static_cast<Handle*>(hv);
    const int P = static_cast<int>(h->pb.pm.names.size());
    std::vector<double> g;
    try {
        *value = oracle::evaluate_with_gradient(h->pb, std::vector<double>(theta, theta + P), g, epsilon);
    } catch (const std::exception&) { return 2; }
    std::copy(g.begin(), g.end(), grad);
    return 0;
}

// NUTSSampler restated over the finite-difference gradient objective (constraint mode as given: the reference
// runs it as phase 2, MCMC_REFLECT).  samples [iterations][P], values / eps_trace [iterations], depth_trace
// [iterations]; returns the number of samples stored, or -2 when an integration throws.
int oracle_nuts(void* hv, int iterations, int adaptation_window, double delta_target, int max_tree_depth, double fd_epsilon,
                int constraint_mode, const double* theta0, uint32_t seed, double* samples, double* values,
                double* eps_trace, int32_t* depth_trace, double* best, double* best_value, long* gradient_calls) {
    auto* h = static_cast<Handle*>(hv);
    oracle::Problem pb = h->pb;
    pb.pm.mode = constraint_mode == 0 ? oracle::OPTIMIZATION_CLAMP : oracle::MCMC_REFLECT;
    const int P = static_cast<int>(pb.pm.names.size());
    oracle::NUTSSettings cfg;
    cfg.iterations = iterations; cfg.adaptation_window = adaptation_window; cfg.delta_target = delta_target;
    cfg.max_tree_depth = max_tree_depth;
    auto grad_fn = [&](const std::vector<double>& th, std::vector<double>& g) {
        return oracle::evaluate_with_gradient(pb, th, g, fd_epsilon);
    };
    auto fn = [&](const std::vector<double>& th) {
        oracle::EvalInfo info;
        const double v = oracle::objective(pb, th, &info);
        if (info.status >= 2) throw std::runtime_error("SimulationException");
        return v;
    };
    try {
        const oracle::NUTSResult r = oracle::nuts(cfg, std::vector<double>(theta0, theta0 + P), grad_fn, fn, pb.pm, seed);
        const int ns = static_cast<int>(r.samples.size());
        for (int s = 0; s < ns; ++s) {
            std::copy(r.samples[s].begin(), r.samples[s].end(), samples + static_cast<size_t>(s) * P);
            values[s] = r.sample_values[s];
            eps_trace[s] = r.epsilon_trace[s];
            depth_trace[s] = r.depth_trace[s];
        }
        if (!r.best.empty()) std::copy(r.best.begin(), r.best.end(), best);
        *best_value = r.best_value;
        if (gradient_calls) *gradient_calls = r.gradient_calls;
        return ns;
    } catch (const std::exception&) { return -2; }
}

int oracle_calibrate(void* hv, int hc_iterations, int cloud_size_multiplier, int threads, uint32_t hc_seed,
                     int mh_iterations, int burn_in, int adaptation_period, int thinning, uint32_t mh_seed,
                     const double* x0, double* best, double* best_value, double* initial_value,
                     double* phase1_best_value, double* phase2_cov, unsigned char* accept_trace, double* samples,
                     double* sample_values, double* mcmc_objective_values, int32_t* n_samples) {
    auto* h = static_cast<Handle*>(hv);
    oracle::Problem pb = h->pb;  // private copy: the calibrator switches the constraint mode
    const int P = static_cast<int>(pb.pm.names.size());
    oracle::HCSettings hc;
    hc.iterations = hc_iterations; hc.cloud_size_multiplier = cloud_size_multiplier; hc.threads = threads;
    oracle::MHSettings mh;
    mh.iterations = mh_iterations; mh.burn_in = burn_in; mh.adaptation_period = adaptation_period;
    mh.thinning = std::max(1, thinning);
    oracle::Objective fn = [&pb](const std::vector<double>& p) {
        oracle::EvalInfo info;
        const double v = oracle::objective(pb, p, &info);
        if (info.status >= 2) throw std::runtime_error("SimulationException");
        return v;
    };
    const oracle::CalibrationResult r =
        oracle::calibrate(hc, mh, std::vector<double>(x0, x0 + P), fn, pb.pm, hc_seed, mh_seed);
    std::copy(r.best.begin(), r.best.end(), best);
    *best_value = r.best_value;
    if (initial_value) *initial_value = r.initial_value;
    if (phase1_best_value) *phase1_best_value = r.phase1.best_value;
    if (phase2_cov) std::copy(r.phase2_cov.begin(), r.phase2_cov.end(), phase2_cov);
    if (accept_trace) std::copy(r.phase2.accept_trace.begin(), r.phase2.accept_trace.end(), accept_trace);
    const int ns = static_cast<int>(r.phase2.samples.size());
    if (n_samples) *n_samples = ns;
    if (samples)
        for (int s = 0; s < ns; ++s) std::copy(r.phase2.samples[s].begin(), r.phase2.samples[s].end(), samples + static_cast<size_t>(s) * P);
    if (sample_values) std::copy(r.phase2.sample_values.begin(), r.phase2.sample_values.end(), sample_values);
    if (mcmc_objective_values) std::copy(r.mcmc_objective_values.begin(), r.mcmc_objective_values.end(), mcmc_objective_values);
    return 0;
}

int oracle_condition_covariance(void* hv, const double* cov, double* out) {
    auto* h = static_cast<Handle*>(hv);
    const oracle::ParameterManager& pm = h->pb.pm;
    const int P = static_cast<int>(pm.names.size());
    const std::vector<double> r = oracle::condition_phase1_covariance(
        std::vector<double>(cov, cov + static_cast<size_t>(P) * P), P, [&](int i) { return pm.sigmas.at(pm.names[i]); });
    std::copy(r.begin(), r.end(), out);
    return 0;
}

namespace {
oracle::SIRParams sir_params(int n, const double* N, const double* C, const double* gamma, double q, double scale_C) {
    oracle::SIRParams p;
    p.n = n; p.q = q; p.scale_C = scale_C;
    p.N.assign(N, N + n); p.C.assign(C, C + n * n); p.gamma.assign(gamma, gamma + n);
    return p;
}
}  // namespace

void oracle_sir_rhs(int n, const double* N, const double* C, const double* gamma, double q, double scale_C,
                    const double* state, double* deriv) {
    oracle::sir_rhs(sir_params(n, N, C, gamma, q, scale_C), state, deriv);
}

int oracle_sir_simulate(int n, const double* N, const double* C, const double* gamma, double q, double scale_C,
                        const double* init, const double* times, int n_times, double abs_err, double rel_err,
                        double* traj, int32_t* n_accept, int32_t* n_reject) {
    try {
        oracle::StepStats st;
        const std::vector<double> flat = oracle::sir_simulate(sir_params(n, N, C, gamma, q, scale_C),
                                                              oracle::state_type(init, init + 3 * n),
                                                              std::vector<double>(times, times + n_times), abs_err, rel_err, &st);
        std::copy(flat.begin(), flat.end(), traj);
        if (n_accept) *n_accept = static_cast<int32_t>(st.accepted);
        if (n_reject) *n_reject = static_cast<int32_t>(st.rejected);
        return 0;
    } catch (const std::exception&) { return 2; }
}

int oracle_model_parameters(void* hv, const double* theta, double* out) {
    auto* h = static_cast<Handle*>(hv);
    const int P = static_cast<int>(h->pb.pm.names.size());
    oracle::Model model(h->pb.base);
    try {
        h->pb.pm.updateModelParameters(std::vector<double>(theta, theta + P), model);
    } catch (...) { return -1; }
    const oracle::Params& m = model.P;
    std::vector<double> v = {m.beta, m.theta, m.sigma, m.gamma_p, m.gamma_A, m.gamma_I, m.gamma_H, m.gamma_ICU};
    for (const std::vector<double>* f : {&m.a, &m.h_infec, &m.p, &m.h, &m.icu, &m.d_H, &m.d_ICU, &m.d_community})
        v.insert(v.end(), f->begin(), f->end());
    v.insert(v.end(), m.beta_values.begin(), m.beta_values.end());
    v.insert(v.end(), m.kappa_values.begin(), m.kappa_values.end());
    std::copy(v.begin(), v.end(), out);
    return static_cast<int>(v.size());
}

int oracle_simulate_samples(void* hv, const double* theta, int S, double* traj, int32_t* status, int nthreads) {
    auto* h = static_cast<Handle*>(hv);
    const int P = static_cast<int>(h->pb.pm.names.size());
    const size_t per = h->pb.time_points.size() * static_cast<size_t>(oracle::NUM_COMPARTMENTS) * h->pb.base.n;
#pragma omp parallel for num_threads(nthreads > 0 ? nthreads : 1) schedule(dynamic)
    for (int s = 0; s < S; ++s) {
        std::vector<double> tr;
        status[s] = oracle::simulate_sample(h->pb, std::vector<double>(theta + static_cast<size_t>(s) * P, theta + static_cast<size_t>(s + 1) * P), tr);
        if (status[s] == 0) std::copy(tr.begin(), tr.end(), traj + static_cast<size_t>(s) * per);
    }
    return 0;
}

int oracle_ppc_select(int n_samples, int num_for_ppc, uint32_t seed, int32_t* out) {
    const std::vector<int> sel = oracle::select_ppc_samples(static_cast<size_t>(n_samples), num_for_ppc, seed);
    std::copy(sel.begin(), sel.end(), out);
    return static_cast<int>(sel.size());
}

int oracle_num_threads(void) {
#if defined(_OPENMP)
    return omp_get_max_threads();
#else
    return 1;
#endif
}

}  // extern "C"
