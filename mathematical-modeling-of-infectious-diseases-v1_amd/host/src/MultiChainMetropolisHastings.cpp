// MultiChainMetropolisHastings.cpp -- Adaptive-Metropolis (Haario) with Robbins-Monro global scale,
// the reference's MetropolisHastingsSampler (src/sir_age_structured/optimizers/
// MetropolisHastingsSampler.cpp:65-412), advanced for C independent chains in lock-step so that
// step 3 of every iteration ("evaluate likelihood", :312) is one batched device call.
//
// Per-chain state is what the reference keeps as sampler members: current_covariance_,
// proposal_cholesky_, running_mean_, log_scale_/global_scale_, recent_accepts_, chain_history_
// and its own std::mt19937.  Host work per iteration is O(C P^2) and runs under OpenMP.
#include "epidemic_hip/HipSEPAIHRD.hpp"

#include <algorithm>
#include <cmath>

namespace epidemic {

struct MultiChainMetropolisHastings::Chain {
    std::mt19937 gen;
    std::vector<double> x, prop, cov, chol, mean;
    double lp = 0.0, log_scale = 0.0, scale = 1.0;
    std::deque<int> recent;
    int emergency = 0, accepted = 0;
    std::vector<double> history;  // (t+1) x P, every state of the chain (covariance re-estimation)
    size_t history_len = 0;
};

namespace {
// lower Cholesky factor, row by row; false if not positive definite (Eigen::LLT::info() != Success)
bool cholesky(const std::vector<double>& A, int P, std::vector<double>& L) {
    std::vector<double> out(static_cast<size_t>(P) * P, 0.0);
    for (int j = 0; j < P; ++j) {
        double d = A[static_cast<size_t>(j) * P + j];
        for (int k = 0; k < j; ++k) d -= out[static_cast<size_t>(j) * P + k] * out[static_cast<size_t>(j) * P + k];
        if (!(d > 0.0)) return false;
        const double ljj = std::sqrt(d);
        out[static_cast<size_t>(j) * P + j] = ljj;
        for (int i = j + 1; i < P; ++i) {
            double s = A[static_cast<size_t>(i) * P + j];
            for (int k = 0; k < j; ++k) s -= out[static_cast<size_t>(i) * P + k] * out[static_cast<size_t>(j) * P + k];
            out[static_cast<size_t>(i) * P + j] = s / ljj;
        }
    }
    L.swap(out);
    return true;
}
inline double sanitize(double v) { return (std::isnan(v) || std::isinf(v)) ? -1e18 : v; }  // safeEvaluate :65-74
}  // namespace

void MultiChainMetropolisHastings::configure(const std::map<std::string, double>& settings) {
    auto get = [&](const char* key, double def) {
        auto it = settings.find(key);
        return it != settings.end() ? it->second : def;
    };
    iterations_ = static_cast<int>(get("mcmc_iterations", 10000.0));
    burn_in_ = static_cast<int>(get("burn_in", 1000.0));
    adaptation_period_ = static_cast<int>(get("adaptation_period", 100.0));
    thinning_ = std::max(1, static_cast<int>(get("thinning", 1.0)));
    regularization_epsilon_ = get("regularization_epsilon", 1e-6);
    target_acceptance_rate_ = get("target_acceptance_rate", 0.234);
    adapt_scale_ = get("adapt_scale", 1.0) != 0.0;
    store_samples_ = get("store_samples", 1.0) != 0.0;
}

OptimizationResult MultiChainMetropolisHastings::optimize(const Eigen::VectorXd& x0, IObjectiveFunction& objective,
                                                          IParameterManager& pm) {
    const int P = static_cast<int>(x0.size());
    std::vector<double> init(x0.data(), x0.data() + P);
    BatchEval eval = [&](const double* th, int B, double* out) {
        for (int b = 0; b < B; ++b) {
            Eigen::VectorXd v(P);
            for (int i = 0; i < P; ++i) v[i] = th[static_cast<size_t>(b) * P + i];
            try { out[b] = sanitize(objective.calculate(v)); } catch (...) { out[b] = -1e18; }
        }
    };
    return run(init, 1, eval, pm)[0];
}

std::vector<OptimizationResult> MultiChainMetropolisHastings::optimizeChains(const std::vector<double>& initial, int C,
                                                                             IBatchObjectiveFunction& objective,
                                                                             IParameterManager& pm) {
    std::vector<int> status;
    BatchEval eval = [&](const double* th, int B, double* out) {
        status.resize(static_cast<size_t>(B));
        objective.calculateBatch(th, B, out, status.data());
        for (int b = 0; b < B; ++b)  // an integration failure is the exception safeEvaluate swallows
            out[b] = status[static_cast<size_t>(b)] >= 2 ? -1e18 : sanitize(out[b]);
    };
    return run(initial, C, eval, pm);
}

std::vector<OptimizationResult> MultiChainMetropolisHastings::run(const std::vector<double>& initial, int C,
                                                                  const BatchEval& eval, IParameterManager& pm) {
    // :207-209 reflection mode for valid Bayesian sampling
    if (auto* hpm = dynamic_cast<HipSEPAIHRDParameterManager*>(&pm)) hpm->setConstraintMode(ConstraintMode::MCMC_REFLECT);
    const int P = static_cast<int>(pm.getParameterCount());
    if (static_cast<int>(initial.size()) != C * P) throw InvalidParameterException("MetropolisHastingsSampler", "initial size != C*P");
    const double scaling_factor = (2.38 * 2.38) / static_cast<double>(P);
    const size_t PP = static_cast<size_t>(P) * P;

    std::vector<Chain> chains(static_cast<size_t>(C));
    std::vector<double> batch(static_cast<size_t>(C) * P), values(static_cast<size_t>(C));
    for (int c = 0; c < C; ++c) {
        Chain& ch = chains[static_cast<size_t>(c)];
        ch.gen.seed(seed_ + static_cast<uint32_t>(c));
        ch.x.assign(initial.begin() + static_cast<size_t>(c) * P, initial.begin() + static_cast<size_t>(c + 1) * P);
        ch.prop.resize(static_cast<size_t>(P));
        if (initial_cov_.size() == PP) {  // warm start from phase 1 (:219-223): no 2.38^2/P scaling
            ch.cov = initial_cov_;
        } else {
            ch.cov.assign(PP, 0.0);
            for (int i = 0; i < P; ++i) {  // :226-237
                const double s = pm.getSigmaForParamIndex(i);
                ch.cov[static_cast<size_t>(i) * P + i] = (s > 0 ? s * s : 1e-6);
            }
            for (double& v : ch.cov) v *= scaling_factor;
        }
        for (int i = 0; i < P; ++i) ch.cov[static_cast<size_t>(i) * P + i] += regularization_epsilon_;
        if (!cholesky(ch.cov, P, ch.chol)) {  // :240-246
            ch.chol.assign(PP, 0.0);
            for (int i = 0; i < P; ++i) ch.chol[static_cast<size_t>(i) * P + i] = 0.1;
        }
        ch.mean = ch.x;
        ch.history.reserve(static_cast<size_t>(iterations_) * P);
        std::copy(ch.x.begin(), ch.x.end(), batch.begin() + static_cast<size_t>(c) * P);
    }
    eval(batch.data(), C, values.data());  // initial state :257

    std::vector<OptimizationResult> results(static_cast<size_t>(C));
    traces_.assign(static_cast<size_t>(C), {});
    auto to_eigen = [P](const std::vector<double>& v) {
        Eigen::VectorXd e(P);
        for (int i = 0; i < P; ++i) e[i] = v[static_cast<size_t>(i)];
        return e;
    };
    for (int c = 0; c < C; ++c) {
        Chain& ch = chains[static_cast<size_t>(c)];
        ch.lp = values[static_cast<size_t>(c)];
        ch.history.insert(ch.history.end(), ch.x.begin(), ch.x.end());
        ch.history_len = 1;
        OptimizationResult& r = results[static_cast<size_t>(c)];
        if (store_samples_) { r.samples.push_back(to_eigen(ch.x)); r.sampleObjectiveValues.push_back(ch.lp); }
        r.bestParameters = to_eigen(ch.x);
        r.bestObjectiveValue = ch.lp;
        traces_[static_cast<size_t>(c)].reserve(static_cast<size_t>(std::max(iterations_ - 1, 0)));
    }

    for (int t = 1; t < iterations_; ++t) {
        // ---- 1+2: adaptation and proposal, independent per chain
#pragma omp parallel for schedule(static)
        for (int c = 0; c < C; ++c) {
            Chain& ch = chains[static_cast<size_t>(c)];
            if (t > burn_in_) {
                {   // updateCovarianceRank1 :154-166, gamma = 10/(t+100)
                    const double* ns = &ch.history[(ch.history_len - 1) * static_cast<size_t>(P)];
                    const double gamma = 10.0 / (t + 100.0);
                    std::vector<double> diff(static_cast<size_t>(P));
                    for (int i = 0; i < P; ++i) diff[static_cast<size_t>(i)] = ns[i] - ch.mean[static_cast<size_t>(i)];
                    for (int i = 0; i < P; ++i) ch.mean[static_cast<size_t>(i)] += gamma * diff[static_cast<size_t>(i)];
                    for (int i = 0; i < P; ++i)
                        for (int j = 0; j < P; ++j)
                            ch.cov[static_cast<size_t>(i) * P + j] =
                                (1.0 - gamma) * ch.cov[static_cast<size_t>(i) * P + j] +
                                gamma * (diff[static_cast<size_t>(i)] * diff[static_cast<size_t>(j)]);
                }
                if (t % adaptation_period_ == 0) {
                    if (ch.history_len >= static_cast<size_t>(P) + 10) {  // recomputeFullCovariance :168-199
                        std::vector<double> mean(static_cast<size_t>(P), 0.0), acc(PP, 0.0);
                        for (size_t s = 0; s < ch.history_len; ++s)
                            for (int i = 0; i < P; ++i) mean[static_cast<size_t>(i)] += ch.history[s * P + i];
                        for (int i = 0; i < P; ++i) mean[static_cast<size_t>(i)] /= static_cast<double>(ch.history_len);
                        ch.mean = mean;
                        for (size_t s = 0; s < ch.history_len; ++s) {
                            const double* row = &ch.history[s * P];
                            for (int i = 0; i < P; ++i) {
                                const double di = row[i] - mean[static_cast<size_t>(i)];
                                for (int j = 0; j < P; ++j)
                                    acc[static_cast<size_t>(i) * P + j] += di * (row[j] - mean[static_cast<size_t>(j)]);
                            }
                        }
                        const double denom = double(ch.history_len - 1);
                        for (int i = 0; i < P; ++i)
                            for (int j = 0; j < P; ++j)
                                ch.cov[static_cast<size_t>(i) * P + j] =
                                    scaling_factor * (acc[static_cast<size_t>(i) * P + j] / denom) +
                                    (i == j ? regularization_epsilon_ : 0.0);
                        cholesky(ch.cov, P, ch.chol);  // kept only on success
                    }
                    std::vector<double> stable = ch.cov;  // :295-300: epsilon once more before the kept LLT
                    for (int i = 0; i < P; ++i) stable[static_cast<size_t>(i) * P + i] += regularization_epsilon_;
                    cholesky(stable, P, ch.chol);
                }
            }
            // generateProposal :91-102
            std::vector<double> z(static_cast<size_t>(P));
            {
                std::normal_distribution<double> dist(0.0, 1.0);
                for (int i = 0; i < P; ++i) z[static_cast<size_t>(i)] = dist(ch.gen);
            }
            Eigen::VectorXd raw(P);
            for (int i = 0; i < P; ++i) {
                double s = 0.0;
                for (int j = 0; j <= i; ++j) s += ch.chol[static_cast<size_t>(i) * P + j] * z[static_cast<size_t>(j)];
                raw[i] = ch.x[static_cast<size_t>(i)] + ch.scale * s;
            }
            const Eigen::VectorXd cons = pm.applyConstraints(raw);  // :309 (const, thread-safe)
            for (int i = 0; i < P; ++i) {
                ch.prop[static_cast<size_t>(i)] = cons[i];
                batch[static_cast<size_t>(c) * P + i] = cons[i];
            }
        }
        // ---- 3: one batched evaluation for all chains
        eval(batch.data(), C, values.data());
        // ---- 4-7: accept / reject, scale adaptation, bookkeeping
#pragma omp parallel for schedule(static)
        for (int c = 0; c < C; ++c) {
            Chain& ch = chains[static_cast<size_t>(c)];
            OptimizationResult& r = results[static_cast<size_t>(c)];
            const double prop_lp = values[static_cast<size_t>(c)];
            const double log_ratio = prop_lp - ch.lp;
            bool accept = false;
            if (log_ratio >= 0.0) accept = true;
            else {
                std::uniform_real_distribution<double> u_dist(0.0, 1.0);  // stateless: same draws as one shared object
                if (std::log(u_dist(ch.gen)) < log_ratio) accept = true;
            }
            if (accept) {
                ch.x = ch.prop;
                ch.lp = prop_lp;
                ch.accepted++;
                if (ch.lp > r.bestObjectiveValue) { r.bestObjectiveValue = ch.lp; r.bestParameters = to_eigen(ch.x); }
            }
            traces_[static_cast<size_t>(c)].push_back(accept ? 1 : 0);
            if (adapt_scale_) {  // adaptGlobalScale :104-152
                ch.recent.push_back(accept ? 1 : 0);
                if (ch.recent.size() > 1000) ch.recent.pop_front();
                double rate = 0.0;
                if (!ch.recent.empty()) {
                    int sum = 0;
                    for (int a : ch.recent) sum += a;
                    rate = static_cast<double>(sum) / ch.recent.size();
                }
                if (ch.recent.size() >= 1000 && rate < 0.001) { ch.log_scale -= 0.7; ch.emergency++; }
                else if (rate < 0.02 && ch.recent.size() >= 500) {
                    double g = 5.0 / std::sqrt(static_cast<double>(t) + 1.0);
                    g = std::min(g, 0.3);
                    ch.log_scale += g * (0.0 - target_acceptance_rate_);
                } else {
                    double g = 1.0 / std::sqrt(static_cast<double>(t) + 1.0);
                    g = std::min(g, 0.1);
                    ch.log_scale += g * ((accept ? 1.0 : 0.0) - target_acceptance_rate_);
                }
                if (ch.scale <= 0.011 && rate > 0.15 && rate < 0.30) ch.log_scale += 0.01;
                ch.log_scale = std::max(std::min(ch.log_scale, 2.3), -6.9);
                ch.scale = std::exp(ch.log_scale);
            }
            ch.history.insert(ch.history.end(), ch.x.begin(), ch.x.end());
            ch.history_len++;
            if (store_samples_ && (t % thinning_ == 0)) {
                r.samples.push_back(to_eigen(ch.x));
                r.sampleObjectiveValues.push_back(ch.lp);
            }
        }
    }

    for (int c = 0; c < C; ++c) {
        Chain& ch = chains[static_cast<size_t>(c)];
        OptimizationResult& r = results[static_cast<size_t>(c)];
        r.finalCovariance = Eigen::MatrixXd(P, P);
        for (int i = 0; i < P; ++i)
            for (int j = 0; j < P; ++j) r.finalCovariance(i, j) = ch.cov[static_cast<size_t>(i) * P + j];
        r.additionalStats["acceptance_rate"] = static_cast<double>(ch.accepted) / iterations_;  // :387
        r.additionalStats["accepted_count"] = ch.accepted;
        r.additionalStats["final_scale"] = ch.scale;
        r.additionalStats["burn_in"] = static_cast<double>(burn_in_);
        r.additionalStats["total_iterations"] = static_cast<double>(iterations_);
    }
    return results;
}

}  // namespace epidemic
