// host_capi.cpp -- flat C entry points over the C++ host mirror, for the Python test-suite only.
// The layout of `sepaihrd_problem` is reused as the carrier of the model / data arrays.
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>

#include "epidemic_hip/BatchedHillClimbing.hpp"
#include "epidemic_hip/BatchedParticleSwarm.hpp"
#include "epidemic_hip/HipModelCalibrator.hpp"
#include "epidemic_hip/HipNUTSSampler.hpp"
#include "epidemic_hip/HipPosteriorEnsemble.hpp"
#include "epidemic_hip/HipSEPAIHRD.hpp"
#include "sepaihrd_hip.h"
#include "sepaihrd_rng.inc"

using namespace epidemic;

namespace {
struct HostHandle {
    std::unique_ptr<HipSEPAIHRDParameterManager> pm;
    std::unique_ptr<SimulationCache> cache;
    std::unique_ptr<CalibrationData> data;
    std::unique_ptr<HipSEPAIHRDObjectiveFunction> obj;
    std::string error;
};
std::vector<std::string> split_lines(const char* s) {
    std::vector<std::string> out;
    if (!s) return out;
    std::stringstream ss(s);
    std::string line;
    while (std::getline(ss, line, '\n'))
        if (!line.empty()) out.push_back(line);
    return out;
}
Eigen::VectorXd vec(const double* p, int n) {
    Eigen::VectorXd v(n);
    for (int i = 0; i < n; ++i) v[i] = p ? p[i] : 0.0;
    return v;
}
thread_local std::string g_error;
thread_local double g_last_mh_loop_seconds = 0.0;
}  // namespace

namespace epidemic {
void canonical_queue_draw_sequence(uint32_t seed, int P, int rounds, const unsigned char* takes_uniform, double* normals, double* log_u);
}

extern "C" {

// test hook: see canonical_queue_draw_sequence (MultiChainMetropolisHastings.cpp)
void host_queue_draw_sequence(uint32_t seed, int P, int rounds, const unsigned char* takes_uniform, double* normals, double* log_u) {
    epidemic::canonical_queue_draw_sequence(seed, P, rounds, takes_uniform, normals, log_u);
}

const char* host_last_error(void) { return g_error.c_str(); }
// iteration loop of this thread's last device-resident sampler run, without set-up and read-back (seconds)
double host_last_mh_loop_seconds(void) { return g_last_mh_loop_seconds; }

// Test hook (no GPU needed): what the adapter raises for a per-chain status >= 2.  Returns 1 when it is the
// SimulationException the reference's solver wrapper throws (Dopri5SolverStrategy.cpp:38-42), 0 for any other type;
// the message goes to host_last_error.
int host_status_exception(int status) {
    try {
        HipSEPAIHRDObjectiveFunction::throwIntegrationFailure(status);
    } catch (const SimulationException& e) {
        g_error = e.what();
        return 1;
    } catch (const std::exception& e) {
        g_error = e.what();
        return 0;
    }
    return 0;
}

// names / npi_names: '\n'-joined; sigmas: [P].  Bounds come from pb->lower/upper.
// with_objective = 0 builds the parameter manager only (no device needed)
void* host_objective_create(const sepaihrd_problem* pb, const char* names, const char* npi_names,
                            const double* sigmas, int device, int cache_capacity, int with_objective) {
    try {
        const int n = pb->n_age;
        SEPAIHRDParameters mp;
        mp.N = vec(pb->N, n);
        mp.M_baseline = Eigen::MatrixXd(n, n);
        std::memcpy(mp.M_baseline.data(), pb->M, sizeof(double) * n * n);
        mp.a = vec(pb->a, n); mp.h_infec = vec(pb->h_infec, n); mp.p = vec(pb->p, n); mp.h = vec(pb->h, n);
        mp.icu = vec(pb->icu, n); mp.d_H = vec(pb->d_H, n); mp.d_ICU = vec(pb->d_ICU, n);
        mp.d_community = vec(pb->d_community, n);
        mp.beta = pb->beta; mp.theta = pb->theta; mp.sigma = pb->sigma; mp.gamma_p = pb->gamma_p;
        mp.gamma_A = pb->gamma_A; mp.gamma_I = pb->gamma_I; mp.gamma_H = pb->gamma_H; mp.gamma_ICU = pb->gamma_ICU;
        mp.beta_end_times.assign(pb->beta_end_times, pb->beta_end_times + pb->n_beta);
        mp.beta_values.assign(pb->beta_values, pb->beta_values + pb->n_beta);
        mp.kappa_end_times.assign(pb->kappa_end_times, pb->kappa_end_times + pb->n_kappa);
        mp.kappa_values.assign(pb->kappa_values, pb->kappa_values + pb->n_kappa);
        mp.E0_multiplier = pb->multipliers[0]; mp.P0_multiplier = pb->multipliers[1];
        mp.A0_multiplier = pb->multipliers[2]; mp.I0_multiplier = pb->multipliers[3];
        mp.H0_multiplier = pb->multipliers[4]; mp.ICU0_multiplier = pb->multipliers[5];
        mp.R0_multiplier = pb->multipliers[6]; mp.D0_multiplier = pb->multipliers[7];
        mp.runup_days = pb->runup_days; mp.seed_exposed = pb->seed_exposed;

        const std::vector<std::string> nm = split_lines(names);
        std::map<std::string, double> sg;
        std::map<std::string, std::pair<double, double>> bd;
        for (size_t i = 0; i < nm.size(); ++i) {
            sg[nm[i]] = sigmas[i];
            bd[nm[i]] = {pb->lower[i], pb->upper[i]};
        }
        auto h = std::make_unique<HostHandle>();
        h->pm = std::make_unique<HipSEPAIHRDParameterManager>(mp, nm, sg, bd, split_lines(npi_names));
        h->pm->setConstraintMode(pb->constraint_mode == SEPAIHRD_CONSTRAINT_REFLECT ? ConstraintMode::MCMC_REFLECT
                                                                                   : ConstraintMode::OPTIMIZATION_CLAMP);
        h->cache = std::make_unique<SimulationCache>(static_cast<size_t>(cache_capacity > 0 ? cache_capacity : 1000));
        if (!with_objective) return h.release();
        auto mat = [&](const double* src) {
            Eigen::MatrixXd m(pb->n_obs, n);
            for (int r = 0; r < pb->n_obs; ++r)
                for (int c = 0; c < n; ++c) m(r, c) = src[static_cast<size_t>(r) * n + c];
            return m;
        };
        h->data = std::make_unique<CalibrationData>(mat(pb->obs_H), mat(pb->obs_ICU), mat(pb->obs_D), mp.N);
        std::shared_ptr<IOdeSolverStrategy> solver;
        if (pb->solver == SEPAIHRD_SOLVER_CASH_KARP54) solver = std::make_shared<CashKarpSolverStrategy>();
        else solver = std::make_shared<Dopri5SolverStrategy>();
        h->obj = std::make_unique<HipSEPAIHRDObjectiveFunction>(
            *h->pm, *h->cache, *h->data, std::vector<double>(pb->times, pb->times + pb->n_times),
            vec(pb->initial_state, 11 * n), solver, pb->abs_err, pb->rel_err, device, pb->arith == SEPAIHRD_ARITH_FMA);
        return h.release();
    } catch (const std::exception& e) {
        g_error = e.what();
        return nullptr;
    }
}

void host_objective_destroy(void* hv) { delete static_cast<HostHandle*>(hv); }

// HipPosteriorEnsemble over the handle's parameter manager / data.  samples: n_samples x P.
// ppc: [6 series][5: lower_95, lower_90, median, upper_90, upper_95][T_pos][n]; selected: capacity
// max(n_samples, num_for_ppc) indices actually simulated; sero / rt (nullable): [5: q025,q05,median,q95,q975][T]
// over samples burn_in, burn_in + thinning, ...
int host_ensemble(void* hv, const sepaihrd_problem* pb, int device, const double* samples, int n_samples,
                  int num_for_ppc, uint32_t seed, double* ppc, int32_t* selected, int32_t* n_selected,
                  int32_t* samples_used, int burn_in, int thinning, double* sero, double* rt) {
    auto* h = static_cast<HostHandle*>(hv);
    try {
        const int n = pb->n_age;
        const size_t P = h->pm->getParameterCount();
        std::shared_ptr<IOdeSolverStrategy> solver;
        if (pb->solver == SEPAIHRD_SOLVER_CASH_KARP54) solver = std::make_shared<CashKarpSolverStrategy>();
        else solver = std::make_shared<Dopri5SolverStrategy>();
        const std::vector<double> times(pb->times, pb->times + pb->n_times);
        HipPosteriorEnsemble ens(*h->pm, *h->data, times, vec(pb->initial_state, 11 * n), solver, pb->abs_err, pb->rel_err,
                                 device, pb->arith == SEPAIHRD_ARITH_FMA);
        std::vector<Eigen::VectorXd> ps(static_cast<size_t>(n_samples), Eigen::VectorXd(static_cast<Eigen::Index>(P)));
        for (int s = 0; s < n_samples; ++s)
            for (size_t i = 0; i < P; ++i) ps[static_cast<size_t>(s)][static_cast<Eigen::Index>(i)] = samples[static_cast<size_t>(s) * P + i];
        const std::vector<int> sel = HipPosteriorEnsemble::selectSamples(ps.size(), num_for_ppc, seed);
        for (size_t i = 0; i < sel.size(); ++i) selected[i] = sel[i];
        *n_selected = static_cast<int32_t>(sel.size());
        const PosteriorPredictiveData d = ens.aggregatePosteriorPredictives(ps, num_for_ppc, seed);
        *samples_used = d.samples_used;
        const PosteriorPredictiveData::IncidenceData* series[6] = {&d.daily_hospitalizations, &d.daily_icu_admissions,
                                                                   &d.daily_deaths, &d.cumulative_hospitalizations,
                                                                   &d.cumulative_icu_admissions, &d.cumulative_deaths};
        const size_t Tp = d.time_points.size();
        for (int ser = 0; ser < 6; ++ser) {
            const Eigen::MatrixXd* m[5] = {&series[ser]->lower_95, &series[ser]->lower_90, &series[ser]->median,
                                           &series[ser]->upper_90, &series[ser]->upper_95};
            for (int q = 0; q < 5; ++q)
                for (size_t t = 0; t < Tp; ++t)
                    for (int a = 0; a < n; ++a)
                        ppc[((static_cast<size_t>(ser) * 5 + q) * Tp + t) * n + a] =
                            (*m[q])(static_cast<Eigen::Index>(t), a);
        }
        if (sero) {
            const auto agg = ens.aggregateSeroprevalence(ps, burn_in, thinning);
            const char* keys[5] = {"q025", "q05", "median", "q95", "q975"};
            size_t k = 0;
            for (double t : times) {
                const auto it = agg.find(t);
                for (int q = 0; q < 5; ++q)
                    sero[static_cast<size_t>(q) * times.size() + k] =
                        it == agg.end() ? std::numeric_limits<double>::quiet_NaN() : it->second.at(keys[q]);
                ++k;
            }
        }
        if (rt) {
            const auto agg = ens.aggregateRt(ps, burn_in, thinning);
            const char* keys[5] = {"q025", "q05", "median", "q95", "q975"};
            size_t k = 0;
            for (double t : times) {
                const auto it = agg.find(t);
                for (int q = 0; q < 5; ++q)
                    rt[static_cast<size_t>(q) * times.size() + k] =
                        it == agg.end() ? std::numeric_limits<double>::quiet_NaN() : it->second.at(keys[q]);
                ++k;
            }
        }
        return 0;
    } catch (const std::exception& e) {
        g_error = e.what();
        return 1;
    }
}

// returns 0 ok, 1 = exception thrown by calculate() (message in host_last_error)
int host_objective_calculate(void* hv, const double* theta, double* value) {
    auto* h = static_cast<HostHandle*>(hv);
    try {
        *value = h->obj->calculate(vec(theta, static_cast<int>(h->pm->getParameterCount())));
        return 0;
    } catch (const std::exception& e) {
        g_error = e.what();
        return 1;
    }
}

int host_objective_calculate_batch(void* hv, const double* thetas, int B, double* out, int* status) {
    auto* h = static_cast<HostHandle*>(hv);
    try {
        h->obj->calculateBatch(thetas, B, out, status);
        return 0;
    } catch (const std::exception& e) {
        g_error = e.what();
        return 1;
    }
}

void host_cache_stats(void* hv, long* calls, long* hits, long* size) {
    auto* h = static_cast<HostHandle*>(hv);
    *calls = static_cast<long>(h->cache->getLikelihoodCalls());
    *hits = static_cast<long>(h->cache->getLikelihoodHits());
    *size = static_cast<long>(h->cache->size());
}

int host_apply_constraints(void* hv, int mode, const double* in, double* out) {
    auto* h = static_cast<HostHandle*>(hv);
    const ConstraintMode keep = h->pm->getConstraintMode();
    h->pm->setConstraintMode(mode == 1 ? ConstraintMode::MCMC_REFLECT : ConstraintMode::OPTIMIZATION_CLAMP);
    const int P = static_cast<int>(h->pm->getParameterCount());
    const Eigen::VectorXd c = h->pm->applyConstraints(vec(in, P));
    for (int i = 0; i < P; ++i) out[i] = c[i];
    h->pm->setConstraintMode(keep);
    return 0;
}

int host_current_parameters(void* hv, double* out) {
    auto* h = static_cast<HostHandle*>(hv);
    const Eigen::VectorXd c = h->pm->getCurrentParameters();
    for (Eigen::Index i = 0; i < c.size(); ++i) out[i] = c[i];
    return 0;
}

// C chains of Adaptive Metropolis through the batched objective.  Outputs per chain:
// accepted[C], best_value[C], best[C*P], final_scale[C], accept_trace[C*(iterations-1)],
// n_samples (same for all chains), samples[C*n_samples*P], sample_values[C*n_samples].
int host_mh_run(void* hv, int C, const double* initial, uint32_t seed, int iterations, int burn_in,
                int adaptation_period, int thinning, double reg_eps, double target_acc, int adapt_scale,
                int use_scalar_interface, int32_t* accepted, double* best_value, double* best, double* final_scale,
                unsigned char* accept_trace, int32_t* n_samples, double* samples, double* sample_values,
                int two_pass_covariance, double* final_cov, int adaptation_window, int device_streams) {
    auto* h = static_cast<HostHandle*>(hv);
    try {
        const int P = static_cast<int>(h->pm->getParameterCount());
        MultiChainMetropolisHastings mh;
        mh.configure({{"mcmc_iterations", double(iterations)}, {"report_interval", 0.0}, {"write_checkpoints", 0.0}, {"write_trace", 0.0}, {"burn_in", double(burn_in)},
                      {"adaptation_period", double(adaptation_period)}, {"thinning", double(thinning)},
                      {"regularization_epsilon", reg_eps}, {"target_acceptance_rate", target_acc},
                      {"adapt_scale", double(adapt_scale)}, {"store_samples", 1.0},
                      {"two_pass_covariance", double(two_pass_covariance)}, {"adaptation_window", double(adaptation_window)},
                      {"device_streams", double(device_streams)}, {"keep_accept_traces", accept_trace ? 1.0 : 0.0}});
        mh.setSeed(seed);
        std::vector<OptimizationResult> res;
        if (use_scalar_interface == 1) {
            for (int c = 0; c < C; ++c) {
                mh.setSeed(seed + static_cast<uint32_t>(c));
                res.push_back(mh.optimize(vec(initial + static_cast<size_t>(c) * P, P), *h->obj, *h->pm));
                if (accept_trace)
                    std::copy(mh.acceptTraces()[0].begin(), mh.acceptTraces()[0].end(),
                              accept_trace + static_cast<size_t>(c) * (iterations - 1));
            }
        } else {
            const std::vector<double> init(initial, initial + static_cast<size_t>(C) * P);
            res = use_scalar_interface == 2 ? mh.optimizeChainsOnDevice(init, C, *h->obj, *h->pm)  // device-resident state
                                            : mh.optimizeChains(init, C, *h->obj, *h->pm);
            g_last_mh_loop_seconds = mh.lastLoopSeconds();
            if (accept_trace)
                for (int c = 0; c < C; ++c)
                    std::copy(mh.acceptTraces()[static_cast<size_t>(c)].begin(), mh.acceptTraces()[static_cast<size_t>(c)].end(),
                              accept_trace + static_cast<size_t>(c) * (iterations - 1));
        }
        const int ns = static_cast<int>(res[0].samples.size());
        if (n_samples) *n_samples = ns;
        for (int c = 0; c < C; ++c) {
            const OptimizationResult& r = res[static_cast<size_t>(c)];
            if (accepted) accepted[c] = static_cast<int32_t>(r.additionalStats.at("accepted_count"));
            if (best_value) best_value[c] = r.bestObjectiveValue;
            if (final_scale) final_scale[c] = r.additionalStats.at("final_scale");
            if (best) for (int i = 0; i < P; ++i) best[static_cast<size_t>(c) * P + i] = r.bestParameters[i];
            if (samples)
                for (int s = 0; s < ns; ++s)
                    for (int i = 0; i < P; ++i)
                        samples[(static_cast<size_t>(c) * ns + s) * P + i] = r.samples[static_cast<size_t>(s)][i];
            if (sample_values)
                for (int s = 0; s < ns; ++s) sample_values[static_cast<size_t>(c) * ns + s] = r.sampleObjectiveValues[static_cast<size_t>(s)];
            if (final_cov)
                for (int i = 0; i < P; ++i)
                    for (int j = 0; j < P; ++j) final_cov[(static_cast<size_t>(c) * P + i) * P + j] = r.finalCovariance(i, j);
        }
        return 0;
    } catch (const std::exception& e) {
        g_error = e.what();
        return 1;
    }
}

// BatchedHillClimbingOptimizer in the clamp mode ModelCalibrator sets for phase 1
// (ModelCalibrator.cpp:62-66).  use_scalar_interface hides calculateBatch from the optimizer, so that
// every objective value comes from IObjectiveFunction::calculate (one launch per value).
int host_hc_run(void* hv, const double* x0, uint32_t seed, int threads, int iterations, int cloud_size_multiplier,
                int use_scalar_interface, double* best, double* best_value, double* final_cov, double* trace,
                long* evaluations, long* launches) {
    auto* h = static_cast<HostHandle*>(hv);
    try {
        const int P = static_cast<int>(h->pm->getParameterCount());
        h->pm->setConstraintMode(ConstraintMode::OPTIMIZATION_CLAMP);
        BatchedHillClimbingOptimizer hc;
        hc.configure({{"iterations", double(iterations)}, {"cloud_size_multiplier", double(cloud_size_multiplier)},
                      {"threads", double(threads)}, {"seed", double(seed)}});
        struct ScalarOnly : IObjectiveFunction {
            IObjectiveFunction& inner;
            explicit ScalarOnly(IObjectiveFunction& o) : inner(o) {}
            double calculate(const Eigen::VectorXd& p) const override { return inner.calculate(p); }
            const std::vector<std::string>& getParameterNames() const override { return inner.getParameterNames(); }
        } scalar(*h->obj);
        const OptimizationResult r = use_scalar_interface ? hc.optimize(vec(x0, P), scalar, *h->pm)
                                                          : hc.optimize(vec(x0, P), *h->obj, *h->pm);
        for (int i = 0; i < P; ++i) best[i] = r.bestParameters[i];
        *best_value = r.bestObjectiveValue;
        for (int a = 0; a < P; ++a)
            for (int b = 0; b < P; ++b) final_cov[static_cast<size_t>(a) * P + b] = r.finalCovariance(a, b);
        if (trace) std::copy(hc.currentTrace().begin(), hc.currentTrace().end(), trace);
        if (evaluations) *evaluations = hc.evaluations();
        if (launches) *launches = hc.launches();
        return 0;
    } catch (const std::exception& e) {
        g_error = e.what();
        return 1;
    }
}

// BatchedParticleSwarmOptimization.  settings: n_settings (key, value) pairs, keys as in pso_settings.txt plus `seed`.
int host_pso_run(void* hv, const double* x0, const char* const* keys, const double* values, int n_settings,
                 double* best, double* best_value, double* final_cov, double* trace, long* evaluations, long* launches) {
    auto* h = static_cast<HostHandle*>(hv);
    try {
        const int P = static_cast<int>(h->pm->getParameterCount());
        h->pm->setConstraintMode(ConstraintMode::OPTIMIZATION_CLAMP);
        std::map<std::string, double> settings;
        for (int i = 0; i < n_settings; ++i) settings[keys[i]] = values[i];
        BatchedParticleSwarmOptimization pso;
        pso.configure(settings);
        const OptimizationResult r = pso.optimize(x0 ? vec(x0, P) : Eigen::VectorXd(), *h->obj, *h->pm);
        for (int i = 0; i < P; ++i) best[i] = r.bestParameters[i];
        *best_value = r.bestObjectiveValue;
        if (final_cov)
            for (int a = 0; a < P; ++a)
                for (int b = 0; b < P; ++b) final_cov[static_cast<size_t>(a) * P + b] = r.finalCovariance(a, b);
        if (trace) std::copy(pso.bestTrace().begin(), pso.bestTrace().end(), trace);
        if (evaluations) *evaluations = pso.evaluations();
        if (launches) *launches = pso.launches();
        return 0;
    } catch (const std::exception& e) {
        g_error = e.what();
        return 1;
    }
}

namespace {
void copy_calibration(const HipModelCalibrator& cal, int P, int mh_iterations, double* best, double* best_value,
                      double* initial_value, double* phase1_best_value, double* phase2_cov, unsigned char* accept_trace,
                      double* samples, double* sample_values, double* mcmc_objective_values, int32_t* n_samples) {
    for (int i = 0; i < P; ++i) best[i] = cal.getBestParameterVector()[i];
    *best_value = cal.getBestObjectiveValue();
    if (initial_value) *initial_value = cal.getInitialObjectiveValue();
    if (phase1_best_value) *phase1_best_value = cal.getPhase1Result().bestObjectiveValue;
    if (phase2_cov)
        for (int a = 0; a < P; ++a)
            for (int b = 0; b < P; ++b) phase2_cov[static_cast<size_t>(a) * P + b] = cal.getPhase2Covariance()(a, b);
    const auto& res = cal.getPhase2Results();
    const int ns = static_cast<int>(res[0].samples.size());
    if (n_samples) *n_samples = ns;
    for (size_t c = 0; c < res.size(); ++c) {
        if (accept_trace)
            std::copy(cal.acceptTraces()[c].begin(), cal.acceptTraces()[c].end(), accept_trace + c * static_cast<size_t>(mh_iterations - 1));
        for (int s = 0; s < ns; ++s) {
            if (samples)
                for (int i = 0; i < P; ++i) samples[(c * ns + s) * P + i] = res[c].samples[static_cast<size_t>(s)][i];
            if (sample_values) sample_values[c * ns + s] = res[c].sampleObjectiveValues[static_cast<size_t>(s)];
        }
    }
    if (mcmc_objective_values) std::copy(cal.getMCMCObjectiveValues().begin(), cal.getMCMCObjectiveValues().end(), mcmc_objective_values);
}
}  // namespace

// HipModelCalibrator: two-phase calibration.  Per-chain outputs are chain-major; n_samples is per chain.
int host_calibrate(void* hv, int hc_iterations, int cloud_size_multiplier, int threads, uint32_t hc_seed,
                   int mh_iterations, int burn_in, int adaptation_period, int thinning, uint32_t mh_seed, int chains,
                   double* best, double* best_value, double* initial_value, double* phase1_best_value,
                   double* phase2_cov, unsigned char* accept_trace, double* samples, double* sample_values,
                   double* mcmc_objective_values, int32_t* n_samples) {
    auto* h = static_cast<HostHandle*>(hv);
    try {
        const int P = static_cast<int>(h->pm->getParameterCount());
        h->pm->setConstraintMode(ConstraintMode::OPTIMIZATION_CLAMP);  // mode at construction time
        HipModelCalibrator cal(*h->pm, *h->obj);
        cal.calibrate({{"iterations", double(hc_iterations)}, {"cloud_size_multiplier", double(cloud_size_multiplier)},
                       {"threads", double(threads)}, {"seed", double(hc_seed)}},
                      {{"mcmc_iterations", double(mh_iterations)}, {"report_interval", 0.0}, {"write_checkpoints", 0.0}, {"write_trace", 0.0}, {"burn_in", double(burn_in)},
                       {"adaptation_period", double(adaptation_period)}, {"thinning", double(thinning)},
                       {"seed", double(mh_seed)}, {"store_samples", 1.0}},
                      chains);
        copy_calibration(cal, P, mh_iterations, best, best_value, initial_value, phase1_best_value, phase2_cov, accept_trace,
                         samples, sample_values, mcmc_objective_values, n_samples);
        return 0;
    } catch (const std::exception& e) {
        g_error = e.what();
        return 1;
    }
}

// SEPAIHRDModelCalibration::runPSOMCMC (SEPAIHRDModelCalibration.cpp:179-208): phase 1 = particle swarm.
int host_calibrate_pso(void* hv, const char* const* keys, const double* values, int n_settings, int mh_iterations,
                       int burn_in, int adaptation_period, int thinning, uint32_t mh_seed, int chains, double* best,
                       double* best_value, double* initial_value, double* phase1_best_value, double* phase2_cov,
                       unsigned char* accept_trace, double* samples, double* sample_values,
                       double* mcmc_objective_values, int32_t* n_samples) {
    auto* h = static_cast<HostHandle*>(hv);
    try {
        const int P = static_cast<int>(h->pm->getParameterCount());
        h->pm->setConstraintMode(ConstraintMode::OPTIMIZATION_CLAMP);
        std::map<std::string, double> phase1;
        for (int i = 0; i < n_settings; ++i) phase1[keys[i]] = values[i];
        HipModelCalibrator cal(*h->pm, *h->obj);
        cal.setPhase1Algorithm(std::make_unique<BatchedParticleSwarmOptimization>());
        cal.calibrate(phase1,
                      {{"mcmc_iterations", double(mh_iterations)}, {"report_interval", 0.0}, {"write_checkpoints", 0.0}, {"write_trace", 0.0}, {"burn_in", double(burn_in)},
                       {"adaptation_period", double(adaptation_period)}, {"thinning", double(thinning)},
                       {"seed", double(mh_seed)}, {"store_samples", 1.0}},
                      chains);
        copy_calibration(cal, P, mh_iterations, best, best_value, initial_value, phase1_best_value, phase2_cov, accept_trace,
                         samples, sample_values, mcmc_objective_values, n_samples);
        return 0;
    } catch (const std::exception& e) {
        g_error = e.what();
        return 1;
    }
}

// HipSEPAIHRDGradientObjectiveFunction::evaluate_with_gradient over the handle's parameter manager / data.
// returns 0 ok, 1 = exception (message in host_last_error)
int host_gradient(void* hv, const sepaihrd_problem* pb, int device, const double* theta, double epsilon, double* value,
                  double* grad) {
    auto* h = static_cast<HostHandle*>(hv);
    try {
        const int n = pb->n_age;
        const int P = static_cast<int>(h->pm->getParameterCount());
        std::shared_ptr<IOdeSolverStrategy> solver;
        if (pb->solver == SEPAIHRD_SOLVER_CASH_KARP54) solver = std::make_shared<CashKarpSolverStrategy>();
        else solver = std::make_shared<Dopri5SolverStrategy>();
        SimulationCache cache(16);
        HipSEPAIHRDGradientObjectiveFunction obj(*h->pm, cache, *h->data, std::vector<double>(pb->times, pb->times + pb->n_times),
                                                 vec(pb->initial_state, 11 * n), solver, pb->abs_err, pb->rel_err, device,
                                                 pb->arith == SEPAIHRD_ARITH_FMA);
        obj.epsilon_ = epsilon;
        Eigen::VectorXd g;
        IGradientObjectiveFunction& iface = obj;  // through the interface NUTS uses (NUTSSampler.cpp:80)
        *value = iface.evaluate_with_gradient(vec(theta, P), g);
        for (int i = 0; i < P; ++i) grad[i] = g[i];
        return 0;
    } catch (const std::exception& e) {
        g_error = e.what();
        return 1;
    }
}

// HipNUTSSampler over a HipSEPAIHRDGradientObjectiveFunction built on the handle's parameter manager / data
// (SEPAIHRDModelCalibration::runNUTS runs it as phase 2: MCMC_REFLECT).  Outputs as oracle_nuts; returns the number
// of samples, or -1 on an exception (message in host_last_error).
int host_nuts_run(void* hv, const sepaihrd_problem* pb, int device, int iterations, int adaptation_window, double delta_target,
                  int max_tree_depth, double fd_epsilon, int constraint_mode, const double* theta0, uint32_t seed,
                  double* samples, double* values, double* eps_trace, int32_t* depth_trace, double* best, double* best_value,
                  long* gradient_calls, long* gradient_launches) {
    auto* h = static_cast<HostHandle*>(hv);
    try {
        const int n = pb->n_age;
        const int P = static_cast<int>(h->pm->getParameterCount());
        h->pm->setConstraintMode(constraint_mode == 0 ? ConstraintMode::OPTIMIZATION_CLAMP : ConstraintMode::MCMC_REFLECT);
        std::shared_ptr<IOdeSolverStrategy> solver;
        if (pb->solver == SEPAIHRD_SOLVER_CASH_KARP54) solver = std::make_shared<CashKarpSolverStrategy>();
        else solver = std::make_shared<Dopri5SolverStrategy>();
        SimulationCache cache(1000);
        HipSEPAIHRDGradientObjectiveFunction obj(*h->pm, cache, *h->data, std::vector<double>(pb->times, pb->times + pb->n_times),
                                                 vec(pb->initial_state, 11 * n), solver, pb->abs_err, pb->rel_err, device,
                                                 pb->arith == SEPAIHRD_ARITH_FMA);
        obj.epsilon_ = fd_epsilon;
        HipNUTSSampler nuts;
        nuts.configure({{"nuts_iterations", double(iterations)}, {"nuts_adaptation_window", double(adaptation_window)},
                        {"nuts_delta_target", delta_target}, {"nuts_max_tree_depth", double(max_tree_depth)},
                        {"seed", double(seed)}});
        IObjectiveFunction& iface = static_cast<HipSEPAIHRDObjectiveFunction&>(obj);
        const OptimizationResult r = nuts.optimize(vec(theta0, P), iface, *h->pm);
        const int ns = static_cast<int>(r.samples.size());
        for (int s = 0; s < ns; ++s) {
            for (int i = 0; i < P; ++i) samples[static_cast<size_t>(s) * P + i] = r.samples[static_cast<size_t>(s)][i];
            values[s] = r.sampleObjectiveValues[static_cast<size_t>(s)];
            eps_trace[s] = nuts.epsilonTrace()[static_cast<size_t>(s)];
            depth_trace[s] = nuts.depthTrace()[static_cast<size_t>(s)];
        }
        if (r.bestParameters.size() == P)
            for (int i = 0; i < P; ++i) best[i] = r.bestParameters[i];
        *best_value = r.bestObjectiveValue;
        if (gradient_calls) *gradient_calls = nuts.gradientCalls();
        if (gradient_launches) *gradient_launches = nuts.gradientLaunches();
        return ns;
    } catch (const std::exception& e) {
        g_error = e.what();
        return -1;
    }
}

// Test hook (no GPU): the device's restatement of glibc's log (csrc/sepaihrd_rng.inc), compiled for the host
void host_glibc_log(const double* x, int n, double* out) {
    for (int i = 0; i < n; ++i) out[i] = sepaihrd_rng::glibc_log(x[i]);
}

// the same for glibc's exp (the device's scale adaptation, exp(log_scale_))
void host_glibc_exp(const double* x, int n, double* out) {
    for (int i = 0; i < n; ++i) out[i] = sepaihrd_rng::glibc_exp(x[i]);
}

// The device-resident sampler with the reference's reporting and trace files switched ON (MetropolisHastingsSampler.cpp:
// 363-383,399-411,440-469): progress lines into `log_path` (one per line), files into `dir`; device_streams = 0 draws on the
// host.  Outputs: every chain's samples [C][n_samples][P] and their values, *fell_back (the libm self-check refused the
// device streams), failure counts [3].
int host_mh_run_reported(void* hv, int C, const double* initial, uint32_t seed, int iterations, int burn_in, int adaptation_period,
                         int thinning, int report_interval, int checkpoint_chains, int device_state, int device_streams,
                         const char* dir, const char* log_path, double* samples, double* sample_values, int32_t* n_samples,
                         int* fell_back, long* failures) {
    auto* h = static_cast<HostHandle*>(hv);
    try {
        const int P = static_cast<int>(h->pm->getParameterCount());
        MultiChainMetropolisHastings mh;
        mh.configure({{"mcmc_iterations", double(iterations)}, {"burn_in", double(burn_in)}, {"adaptation_period", double(adaptation_period)},
                      {"thinning", double(thinning)}, {"report_interval", double(report_interval)}, {"write_checkpoints", 1.0},
                      {"write_trace", 1.0}, {"checkpoint_chains", double(checkpoint_chains)}, {"device_streams", double(device_streams)},
                      {"keep_accept_traces", 0.0}});
        mh.setSeed(seed);
        mh.setOutputDirectory(dir ? dir : "");
        std::ofstream log;
        if (log_path) {
            log.open(log_path);
            mh.setProgressSink([&log](const std::string& level, const std::string& msg) { log << level << " " << msg << std::endl; });
        }
        const std::vector<double> init(initial, initial + static_cast<size_t>(C) * P);
        const std::vector<OptimizationResult> res = device_state ? mh.optimizeChainsOnDevice(init, C, *h->obj, *h->pm)
                                                                 : mh.optimizeChains(init, C, *h->obj, *h->pm);
        const int ns = static_cast<int>(res[0].samples.size());
        if (n_samples) *n_samples = ns;
        for (int c = 0; c < C; ++c)
            for (int s = 0; s < ns; ++s) {
                if (samples)
                    for (int i = 0; i < P; ++i) samples[(static_cast<size_t>(c) * ns + s) * P + i] = res[static_cast<size_t>(c)].samples[static_cast<size_t>(s)][i];
                if (sample_values) sample_values[static_cast<size_t>(c) * ns + s] = res[static_cast<size_t>(c)].sampleObjectiveValues[static_cast<size_t>(s)];
            }
        if (fell_back) *fell_back = mh.deviceStreamsFellBack() ? 1 : 0;
        if (failures)
            for (size_t k = 0; k < 3; ++k) failures[k] = k < mh.failureCounts().size() ? mh.failureCounts()[k] : 0;
        return 0;
    } catch (const std::exception& e) {
        g_error = e.what();
        return 1;
    }
}

// The host twin of sepaihrd_device_libm_check (no GPU): csrc/sepaihrd_rng.inc's log / exp compiled for the host, on the SAME
// fixed arguments, against this process's std::log / std::exp.  Counts of differing arguments.
void host_libm_selfcheck(int* n_args, int* n_log_diff, int* n_exp_diff) {
    int dl = 0, de = 0;
    for (int i = 0; i < sepaihrd_rng::LIBM_CHECK_N; ++i) {
        volatile double xl = sepaihrd_rng::libm_check_log_arg(i), xe = sepaihrd_rng::libm_check_exp_arg(i);
        const double a = sepaihrd_rng::glibc_log(xl), b = std::log(xl), c = sepaihrd_rng::glibc_exp(xe), d = std::exp(xe);
        if (std::memcmp(&a, &b, sizeof(double)) != 0) ++dl;
        if (std::memcmp(&c, &d, sizeof(double)) != 0) ++de;
    }
    if (n_args) *n_args = sepaihrd_rng::LIBM_CHECK_N;
    if (n_log_diff) *n_log_diff = dl;
    if (n_exp_diff) *n_exp_diff = de;
}
// the self-check arguments themselves (tests look at what they cover)
void host_libm_selfcheck_args(double* log_args, double* exp_args) {
    for (int i = 0; i < sepaihrd_rng::LIBM_CHECK_N; ++i) {
        log_args[i] = sepaihrd_rng::libm_check_log_arg(i);
        exp_args[i] = sepaihrd_rng::libm_check_exp_arg(i);
    }
}

// SEPAIHRD_ARITH_* the reference-shaped constructors select (environment SEPAIHRD_ARITH): what bench.py's default --arith
// must equal (tests/test_host_logic.py)
int host_default_arith() { return HipSEPAIHRDObjectiveFunction::defaultArithmeticIsFma() ? SEPAIHRD_ARITH_FMA : SEPAIHRD_ARITH_STRICT; }

// Pure host (no GPU): exact-sort quantiles across chains of every column of a summary table, out [n_probs][width].
int host_summary_quantiles(const double* table, int rows, int width, const double* probs, int n_probs, double* out) {
    try {
        const std::vector<double> q = MultiChainMetropolisHastings::summaryQuantiles(
            std::vector<double>(table, table + static_cast<size_t>(rows) * width), width, std::vector<double>(probs, probs + n_probs));
        std::copy(q.begin(), q.end(), out);
        return 0;
    } catch (const std::exception& e) {
        g_error = e.what();
        return 1;
    }
}

// optimizeChainGroupsOnDevice with stored samples, then the all-gather of the per-chain summary records over the
// groups' devices.  records [C][2P+2] (host concatenation, chain order), gathered [G][C][2P+2] (what each group's device
// holds afterwards), samples [C][n_samples][P], best_value [C], accepted [C]; *backend_used: SEPAIHRD_GATHER_RCCL / _HOST.
int host_mh_groups_summaries(void** handles, int G, int C, const double* initial, uint32_t seed, int iterations, int burn_in,
                             int adaptation_period, int thinning, int backend, double* records, double* gathered, double* samples,
                             double* best_value, int32_t* accepted, int32_t* backend_used) {
    try {
        auto* h0 = static_cast<HostHandle*>(handles[0]);
        const int P = static_cast<int>(h0->pm->getParameterCount());
        std::vector<HipSEPAIHRDObjectiveFunction*> objs;
        for (int g = 0; g < G; ++g) objs.push_back(static_cast<HostHandle*>(handles[g])->obj.get());
        MultiChainMetropolisHastings mh;
        mh.configure({{"mcmc_iterations", double(iterations)}, {"report_interval", 0.0}, {"write_checkpoints", 0.0}, {"write_trace", 0.0}, {"burn_in", double(burn_in)},
                      {"adaptation_period", double(adaptation_period)}, {"thinning", double(thinning)}, {"store_samples", 1.0}});
        mh.setSeed(seed);
        const std::vector<OptimizationResult> res =
            mh.optimizeChainGroupsOnDevice(std::vector<double>(initial, initial + static_cast<size_t>(C) * P), C, objs, *h0->pm);
        const std::vector<double>& rec = mh.chainSummaries();
        if (records) std::copy(rec.begin(), rec.end(), records);
        const int used = mh.gatherChainSummaries(objs, backend);
        if (backend_used) *backend_used = used;
        if (gathered)
            for (int g = 0; g < G; ++g) {
                const std::vector<double> t = mh.gatheredSummaries(*objs[static_cast<size_t>(g)]);
                std::copy(t.begin(), t.end(), gathered + static_cast<size_t>(g) * rec.size());
            }
        const size_t ns = res[0].samples.size();
        for (int c = 0; c < C; ++c) {
            const OptimizationResult& r = res[static_cast<size_t>(c)];
            if (best_value) best_value[c] = r.bestObjectiveValue;
            if (accepted) accepted[c] = static_cast<int32_t>(r.additionalStats.at("accepted_count"));
            if (samples)
                for (size_t k = 0; k < ns; ++k)
                    for (int i = 0; i < P; ++i) samples[(static_cast<size_t>(c) * ns + k) * P + i] = r.samples[k][i];
        }
        return 0;
    } catch (const std::exception& e) {
        g_error = e.what();
        return 1;
    }
}

// optimizeChainGroupsOnDevice over G handles (one device context each; parameter manager of the first).
// Outputs as host_mh_run, without the samples.
int host_mh_run_groups(void** handles, int G, int C, const double* initial, uint32_t seed, int iterations, int burn_in,
                       int adaptation_period, int thinning, int32_t* accepted, double* best_value, double* best,
                       unsigned char* accept_trace) {
    try {
        auto* h0 = static_cast<HostHandle*>(handles[0]);
        const int P = static_cast<int>(h0->pm->getParameterCount());
        std::vector<HipSEPAIHRDObjectiveFunction*> objs;
        for (int g = 0; g < G; ++g) objs.push_back(static_cast<HostHandle*>(handles[g])->obj.get());
        MultiChainMetropolisHastings mh;
        mh.configure({{"mcmc_iterations", double(iterations)}, {"report_interval", 0.0}, {"write_checkpoints", 0.0}, {"write_trace", 0.0}, {"burn_in", double(burn_in)},
                      {"adaptation_period", double(adaptation_period)}, {"thinning", double(thinning)},
                      {"store_samples", 0.0}});
        mh.setSeed(seed);
        const std::vector<OptimizationResult> res =
            mh.optimizeChainGroupsOnDevice(std::vector<double>(initial, initial + static_cast<size_t>(C) * P), C, objs, *h0->pm);
        g_last_mh_loop_seconds = mh.lastLoopSeconds();
        for (int c = 0; c < C; ++c) {
            const OptimizationResult& r = res[static_cast<size_t>(c)];
            if (accepted) accepted[c] = static_cast<int32_t>(r.additionalStats.at("accepted_count"));
            if (best_value) best_value[c] = r.bestObjectiveValue;
            if (best) for (int i = 0; i < P; ++i) best[static_cast<size_t>(c) * P + i] = r.bestParameters[i];
            if (accept_trace)
                std::copy(mh.acceptTraces()[static_cast<size_t>(c)].begin(), mh.acceptTraces()[static_cast<size_t>(c)].end(),
                          accept_trace + static_cast<size_t>(c) * (iterations - 1));
        }
        return 0;
    } catch (const std::exception& e) {
        g_error = e.what();
        return 1;
    }
}

}  // extern "C"

// The reference-shaped constructors (SEPAIHRDModelCalibration.cpp:84-118): model -> parameter manager -> objective,
// argument for argument, (a) with a HipSEPAIHRDParameterManager built from the model and (b) with an IParameterManager
// that is NOT one (forwards to a private manager: stands for the reference's own SEPAIHRDParameterManager), whose
// constraint mode is flipped behind the objective's back between evaluations.
// values: [2 managers][2 modes: clamp, reflect][B]; model_back: the calibrated entries read back from the MODEL after
// updateModelParameters(thetas[0]) in clamp mode, [P].
namespace {
class ForwardingManager : public IParameterManager {
public:
    explicit ForwardingManager(HipSEPAIHRDParameterManager& inner) : in_(inner) {}
    Eigen::VectorXd getCurrentParameters() const override { return in_.getCurrentParameters(); }
    void updateModelParameters(const Eigen::VectorXd& p) override { in_.updateModelParameters(p); }
    const std::vector<std::string>& getParameterNames() const override { return in_.getParameterNames(); }
    size_t getParameterCount() const override { return in_.getParameterCount(); }
    double getSigmaForParamIndex(int i) const override { return in_.getSigmaForParamIndex(i); }
    Eigen::VectorXd applyConstraints(const Eigen::VectorXd& p) const override { return in_.applyConstraints(p); }
    int getIndexForParam(const std::string& n) const override { return in_.getIndexForParam(n); }
    double getLowerBoundForParamIndex(int i) const override { return in_.getLowerBoundForParamIndex(i); }
    double getUpperBoundForParamIndex(int i) const override { return in_.getUpperBoundForParamIndex(i); }
    void setMode(ConstraintMode m) { in_.setConstraintMode(m); }
private:
    HipSEPAIHRDParameterManager& in_;
};
}  // namespace

extern "C" int host_reference_constructors(const sepaihrd_problem* pb, const char* names, const char* npi_names,
                                           const double* sigmas, const double* thetas, int B, double* values,
                                           double* model_back) {
    try {
        const int n = pb->n_age;
        SEPAIHRDParameters mp;
        mp.N = vec(pb->N, n);
        mp.M_baseline = Eigen::MatrixXd(n, n);
        std::memcpy(mp.M_baseline.data(), pb->M, sizeof(double) * n * n);
        mp.a = vec(pb->a, n); mp.h_infec = vec(pb->h_infec, n); mp.p = vec(pb->p, n); mp.h = vec(pb->h, n);
        mp.icu = vec(pb->icu, n); mp.d_H = vec(pb->d_H, n); mp.d_ICU = vec(pb->d_ICU, n);
        mp.d_community = vec(pb->d_community, n);
        mp.beta = pb->beta; mp.theta = pb->theta; mp.sigma = pb->sigma; mp.gamma_p = pb->gamma_p;
        mp.gamma_A = pb->gamma_A; mp.gamma_I = pb->gamma_I; mp.gamma_H = pb->gamma_H; mp.gamma_ICU = pb->gamma_ICU;
        mp.beta_end_times.assign(pb->beta_end_times, pb->beta_end_times + pb->n_beta);
        mp.beta_values.assign(pb->beta_values, pb->beta_values + pb->n_beta);
        mp.E0_multiplier = pb->multipliers[0]; mp.P0_multiplier = pb->multipliers[1];
        mp.A0_multiplier = pb->multipliers[2]; mp.I0_multiplier = pb->multipliers[3];
        mp.H0_multiplier = pb->multipliers[4]; mp.ICU0_multiplier = pb->multipliers[5];
        mp.R0_multiplier = pb->multipliers[6]; mp.D0_multiplier = pb->multipliers[7];
        mp.runup_days = pb->runup_days; mp.seed_exposed = pb->seed_exposed;
        const std::vector<std::string> nm = split_lines(names);
        std::map<std::string, double> sg;
        std::map<std::string, std::pair<double, double>> bd;
        for (size_t i = 0; i < nm.size(); ++i) {
            sg[nm[i]] = sigmas[i];
            bd[nm[i]] = {pb->lower[i], pb->upper[i]};
        }
        // main.cpp:222-242: the strategy takes the schedule after the baseline period
        auto npi = std::make_shared<PiecewiseConstantNpiStrategy>(
            std::vector<double>(pb->kappa_end_times + 1, pb->kappa_end_times + pb->n_kappa),
            std::vector<double>(pb->kappa_values + 1, pb->kappa_values + pb->n_kappa),
            std::map<std::string, std::pair<double, double>>{}, pb->kappa_values[0], pb->kappa_end_times[0], true,
            split_lines(npi_names));
        auto model = std::make_shared<AgeSEPAIHRDModel>(mp, npi);
        auto mat = [&](const double* src) {
            Eigen::MatrixXd m(pb->n_obs, n);
            for (int r = 0; r < pb->n_obs; ++r)
                for (int c = 0; c < n; ++c) m(r, c) = src[static_cast<size_t>(r) * n + c];
            return m;
        };
        const CalibrationData data(mat(pb->obs_H), mat(pb->obs_ICU), mat(pb->obs_D), mp.N);
        std::shared_ptr<IOdeSolverStrategy> solver;
        if (pb->solver == SEPAIHRD_SOLVER_CASH_KARP54) solver = std::make_shared<CashKarpSolverStrategy>();
        else solver = std::make_shared<Dopri5SolverStrategy>();
        const std::vector<double> times(pb->times, pb->times + pb->n_times);
        const Eigen::VectorXd x0 = vec(pb->initial_state, 11 * n);
        const size_t P = nm.size();

        // (a) the two make_unique calls of setupCalibrator with the class names changed
        auto pm = std::make_unique<HipSEPAIHRDParameterManager>(model, nm, sg, bd);
        SimulationCache cache_a(4);
        auto objective = std::make_unique<HipSEPAIHRDObjectiveFunction>(model, *pm, cache_a, data, times, x0, solver,
                                                                         pb->abs_err, pb->rel_err);
        // (b) a manager of another type, mode changed without telling the objective
        HipSEPAIHRDParameterManager inner(model->getModelParameters(), nm, sg, bd, split_lines(npi_names));
        ForwardingManager foreign(inner);
        SimulationCache cache_b(4);
        HipSEPAIHRDObjectiveFunction objective_b(model, foreign, cache_b, data, times, x0, solver, pb->abs_err, pb->rel_err);
        for (int mode = 0; mode < 2; ++mode) {
            const ConstraintMode m = mode ? ConstraintMode::MCMC_REFLECT : ConstraintMode::OPTIMIZATION_CLAMP;
            pm->setConstraintMode(m);
            foreign.setMode(m);
            for (int b = 0; b < B; ++b) {
                const Eigen::VectorXd th = vec(thetas + static_cast<size_t>(b) * P, static_cast<int>(P));
                cache_a.clear();
                cache_b.clear();
                values[(0 * 2 + mode) * static_cast<size_t>(B) + b] = objective->calculate(th);
                values[(1 * 2 + mode) * static_cast<size_t>(B) + b] = objective_b.calculate(th);
            }
        }
        pm->setConstraintMode(ConstraintMode::OPTIMIZATION_CLAMP);
        pm->updateModelParameters(vec(thetas, static_cast<int>(P)));
        HipSEPAIHRDParameterManager readback(model, nm, sg, bd);  // reads the model the first manager wrote into
        const Eigen::VectorXd cur = readback.getCurrentParameters();
        for (size_t i = 0; i < P; ++i) model_back[i] = cur[static_cast<Eigen::Index>(i)];
        return 0;
    } catch (const std::exception& e) {
        g_error = e.what();
        return 1;
    }
}

// Model-side holders without a device: kappa(t) of PiecewiseConstantNpiStrategy at nt times; the model's
// getModelParameters() schedule (baseline first), state size and first / last state names.
extern "C" int host_model_holders(const double* ends_after, const double* values_after, int n_after, double baseline,
                                  double baseline_end, const double* t, int nt, double* kappa_out, double* sched_ends,
                                  double* sched_values, int* state_size, char* first_last, int first_last_len) {
    try {
        auto npi = std::make_shared<PiecewiseConstantNpiStrategy>(std::vector<double>(ends_after, ends_after + n_after),
                                                                   std::vector<double>(values_after, values_after + n_after),
                                                                   std::map<std::string, std::pair<double, double>>{}, baseline,
                                                                   baseline_end);
        for (int i = 0; i < nt; ++i) kappa_out[i] = npi->getReductionFactor(t[i]);
        SEPAIHRDParameters mp;
        const int n = 3;
        mp.N = Eigen::VectorXd::Constant(n, 1000.0);
        mp.M_baseline = Eigen::MatrixXd::Identity(n, n);
        for (Eigen::VectorXd* v : {&mp.a, &mp.h_infec, &mp.p, &mp.h, &mp.icu, &mp.d_H, &mp.d_ICU}) *v = Eigen::VectorXd::Constant(n, 0.1);
        AgeSEPAIHRDModel model(mp, npi);
        const SEPAIHRDParameters back = model.getModelParameters();
        for (size_t k = 0; k < back.kappa_values.size(); ++k) { sched_ends[k] = back.kappa_end_times[k]; sched_values[k] = back.kappa_values[k]; }
        *state_size = model.getStateSize();
        const std::vector<std::string> names = model.getStateNames();
        std::snprintf(first_last, static_cast<size_t>(first_last_len), "%s %s %s %d", names.front().c_str(), names.back().c_str(),
                      npi->getNpiParamName(0).c_str(), static_cast<int>(model.clone()->getNumAgeClasses()));
        bool threw = false;
        try {
            std::vector<double> x(33, 0.0), dx(33, 0.0);
            model.computeDerivatives(x, dx, 0.0);
        } catch (const SimulationException&) { threw = true; }
        return threw ? 0 : 2;  // there is no host right-hand side
    } catch (const std::exception& e) {
        g_error = e.what();
        return 1;
    }
}
