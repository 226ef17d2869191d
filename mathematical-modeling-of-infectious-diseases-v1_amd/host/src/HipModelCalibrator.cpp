// host/src/HipModelCalibrator.cpp -- see the header for the reference lines mirrored.
#include "epidemic_hip/HipModelCalibrator.hpp"

#include <algorithm>
#include <cmath>
#include <limits>

namespace epidemic {

namespace {

// cyclic Jacobi for a symmetric matrix (row-major P x P); eigenvalues ascending, evecs column k <-> evals[k]
void jacobi_eigen_sym(std::vector<double> A, int P, std::vector<double>& evals, std::vector<double>& evecs) {
    const size_t n = static_cast<size_t>(P);
    std::vector<double> V(n * n, 0.0);
    for (size_t i = 0; i < n; ++i) V[i * n + i] = 1.0;
    for (int sweep = 0; sweep < 100; ++sweep) {
        double off = 0.0, diag = 0.0;
        for (size_t a = 0; a < n; ++a)
            for (size_t b = 0; b < n; ++b) (a == b ? diag : off) += A[a * n + b] * A[a * n + b];
        if (off <= 1e-30 * diag || off == 0.0) break;
        for (size_t p = 0; p + 1 < n; ++p)
            for (size_t q = p + 1; q < n; ++q) {
                const double apq = A[p * n + q];
                if (apq == 0.0) continue;
                const double theta = (A[q * n + q] - A[p * n + p]) / (2.0 * apq);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), sn = t * c;
                for (size_t k = 0; k < n; ++k) {
                    const double akp = A[k * n + p], akq = A[k * n + q];
                    A[k * n + p] = c * akp - sn * akq;
                    A[k * n + q] = sn * akp + c * akq;
                }
                for (size_t k = 0; k < n; ++k) {
                    const double apk = A[p * n + k], aqk = A[q * n + k];
                    A[p * n + k] = c * apk - sn * aqk;
                    A[q * n + k] = sn * apk + c * aqk;
                }
                for (size_t k = 0; k < n; ++k) {
                    const double vkp = V[k * n + p], vkq = V[k * n + q];
                    V[k * n + p] = c * vkp - sn * vkq;
                    V[k * n + q] = sn * vkp + c * vkq;
                }
            }
    }
    std::vector<size_t> order(n);
    for (size_t i = 0; i < n; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return A[a * n + a] < A[b * n + b]; });
    evals.resize(n);
    evecs.assign(n * n, 0.0);
    for (size_t k = 0; k < n; ++k) {
        evals[k] = A[order[k] * n + order[k]];
        for (size_t a = 0; a < n; ++a) evecs[a * n + k] = V[a * n + order[k]];
    }
}

}  // namespace

Eigen::MatrixXd conditionPhase1Covariance(const Eigen::MatrixXd& cov_in, const IParameterManager& pm) {
    const int P = static_cast<int>(cov_in.rows());
    const size_t n = static_cast<size_t>(P);
    std::vector<double> cov(n * n);
    for (int a = 0; a < P; ++a)
        for (int b = 0; b < P; ++b) cov[static_cast<size_t>(a) * n + b] = 0.5 * (cov_in(a, b) + cov_in(b, a));  // :100
    std::vector<double> evals, evecs;
    jacobi_eigen_sym(cov, P, evals, evecs);
    for (int i = 0; i < P; ++i) {  // :107-111: eigenvalue i (ascending) is floored with parameter i's sigma
        const double min_var = std::pow(pm.getSigmaForParamIndex(i) * 0.1, 2);
        evals[static_cast<size_t>(i)] = std::max(evals[static_cast<size_t>(i)], min_var);
    }
    std::vector<double> ql(n * n);
    for (size_t a = 0; a < n; ++a)
        for (size_t k = 0; k < n; ++k) ql[a * n + k] = evecs[a * n + k] * evals[k];
    Eigen::MatrixXd out(P, P);
    double tr = 0.0;
    for (size_t a = 0; a < n; ++a)
        for (size_t b = 0; b < n; ++b) {
            double sum = 0.0;
            for (size_t k = 0; k < n; ++k) sum += ql[a * n + k] * evecs[b * n + k];  // Q L' Q^T (:114)
            out(static_cast<Eigen::Index>(a), static_cast<Eigen::Index>(b)) = sum * 4.0;  // :117
            if (a == b) tr += sum * 4.0;
        }
    const double eps = 1e-8 * tr / P;  // :120-121
    for (int a = 0; a < P; ++a) out(a, a) += eps;
    return out;
}

HipModelCalibrator::HipModelCalibrator(HipSEPAIHRDParameterManager& parameterManager,
                                       HipSEPAIHRDObjectiveFunction& objective)
    : pm_(parameterManager), obj_(objective) {
    if (pm_.getParameterNames() != obj_.getParameterNames())  // :32-34
        throw InvalidParameterException("ModelCalibrator", "Parameter names mismatch between ParameterManager and ObjectiveFunction.");
    best_ = pm_.getCurrentParameters();
    best_value_ = obj_.calculate(best_);
    initial_value_ = best_value_;
    if (std::isnan(best_value_) || std::isinf(best_value_)) best_value_ = -std::numeric_limits<double>::infinity();
}

void HipModelCalibrator::calibrate(const std::map<std::string, double>& phase1_settings,
                                   const std::map<std::string, double>& phase2_settings, int chains) {
    const int P = static_cast<int>(pm_.getParameterCount());
    const int C = std::max(1, chains);
    // phase 1 (:57-77)
    pm_.setConstraintMode(ConstraintMode::OPTIMIZATION_CLAMP);
    if (!phase1_algo_) phase1_algo_ = std::make_unique<BatchedHillClimbingOptimizer>();
    phase1_algo_->configure(phase1_settings);
    phase1_ = phase1_algo_->optimize(best_, obj_, pm_);
    if (phase1_.bestObjectiveValue > best_value_) {
        best_value_ = phase1_.bestObjectiveValue;
        best_ = phase1_.bestParameters;
    }
    // phase 2 (:80-146)
    pm_.setConstraintMode(ConstraintMode::MCMC_REFLECT);
    MultiChainMetropolisHastings mh;
    mh.configure(phase2_settings);
    auto it = phase2_settings.find("seed");
    if (it != phase2_settings.end()) mh.setSeed(static_cast<uint32_t>(it->second));
    if (phase1_.finalCovariance.rows() > 0) {
        phase2_cov_ = conditionPhase1Covariance(phase1_.finalCovariance, pm_);
        mh.setInitialCovariance(phase2_cov_);
    }
    std::vector<double> init(static_cast<size_t>(C) * P);
    for (int c = 0; c < C; ++c)
        for (int i = 0; i < P; ++i) init[static_cast<size_t>(c) * P + i] = best_[i];
    phase2_ = mh.optimizeChainsOnDevice(init, C, obj_, pm_);  // sampler state resident in HBM, same numbers
    traces_ = mh.acceptTraces();
    for (const OptimizationResult& r : phase2_)
        if (r.bestObjectiveValue > best_value_) {
            best_value_ = r.bestObjectiveValue;
            best_ = r.bestParameters;
        }
    // objective value of every stored sample (:141-144), all chains in one launch
    std::vector<double> thetas;
    for (const OptimizationResult& r : phase2_)
        for (const Eigen::VectorXd& smp : r.samples)
            for (int i = 0; i < P; ++i) thetas.push_back(smp[i]);
    const int B = static_cast<int>(thetas.size() / static_cast<size_t>(P));
    mcmc_values_.assign(static_cast<size_t>(B), 0.0);
    if (B > 0) obj_.calculateBatch(thetas.data(), B, mcmc_values_.data());  // throws like calculate() on integration failure
    pm_.updateModelParameters(best_);  // :151
}

}  // namespace epidemic
