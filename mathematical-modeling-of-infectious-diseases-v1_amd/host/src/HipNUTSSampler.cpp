// host/src/HipNUTSSampler.cpp -- see the header for the reference lines mirrored.
#include "epidemic_hip/HipNUTSSampler.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>

namespace epidemic {

void HipNUTSSampler::configure(const std::map<std::string, double>& settings) {  // :24-39
    auto get = [&](const char* name, double def) {
        auto it = settings.find(name);
        return it != settings.end() ? it->second : def;
    };
    num_iterations_ = static_cast<int>(get("nuts_iterations", 2000.0));
    adaptation_window_ = static_cast<int>(get("nuts_adaptation_window", 500.0));
    delta_target_ = get("nuts_delta_target", 0.8);
    max_tree_depth_ = static_cast<int>(get("nuts_max_tree_depth", 10.0));
    seed_ = static_cast<uint32_t>(get("seed", 1.0));
}

double HipNUTSSampler::dot(const Vec& a, const Vec& b) const {
    double s = 0.0;
    for (int i = 0; i < P_; ++i) s += a[static_cast<size_t>(i)] * b[static_cast<size_t>(i)];
    return s;
}

double HipNUTSSampler::gradient(const Vec& theta, Vec& grad) {
    ++gradient_calls_;
    for (const Evaluated& e : recent_)
        if (std::memcmp(e.theta.data(), theta.data(), sizeof(double) * static_cast<size_t>(P_)) == 0) {
            grad = e.grad;
            return e.value;
        }
    ++gradient_launches_;
    Eigen::VectorXd th(P_), g(P_);
    for (int i = 0; i < P_; ++i) th[i] = theta[static_cast<size_t>(i)];
    const double value = grad_obj_->evaluate_with_gradient(th, g);
    grad.resize(static_cast<size_t>(P_));
    for (int i = 0; i < P_; ++i) grad[static_cast<size_t>(i)] = g[i];
    recent_.push_front({theta, grad, value});
    if (recent_.size() > 4) recent_.pop_back();
    return value;
}

static double clip_gradient(std::vector<double>& g, double max_norm) {
    double sq = 0.0;
    for (double v : g) sq += v * v;
    const double nrm = std::sqrt(sq);
    if (nrm > max_norm)
        for (double& v : g) v *= max_norm / nrm;
    return nrm;
}

void HipNUTSSampler::leapfrog(Vec& theta, Vec& r, double epsilon) {  // :290-318
    Vec grad;
    gradient(theta, grad);
    clip_gradient(grad, MAX_GRAD_NORM);
    for (int i = 0; i < P_; ++i) r[static_cast<size_t>(i)] += 0.5 * epsilon * grad[static_cast<size_t>(i)];
    for (int i = 0; i < P_; ++i) theta[static_cast<size_t>(i)] += epsilon * r[static_cast<size_t>(i)];
    Eigen::VectorXd th(P_);
    for (int i = 0; i < P_; ++i) th[i] = theta[static_cast<size_t>(i)];
    const Eigen::VectorXd constrained = pm_->applyConstraints(th);
    for (int i = 0; i < P_; ++i) theta[static_cast<size_t>(i)] = constrained[i];
    gradient(theta, grad);
    clip_gradient(grad, MAX_GRAD_NORM);
    for (int i = 0; i < P_; ++i) r[static_cast<size_t>(i)] += 0.5 * epsilon * grad[static_cast<size_t>(i)];
}

bool HipNUTSSampler::checkNoUTurn(const Vec& theta_minus, const Vec& theta_plus, const Vec& r_minus, const Vec& r_plus) const {
    double dot_minus = 0.0, dot_plus = 0.0;  // :409-422
    for (int i = 0; i < P_; ++i) {
        const size_t u = static_cast<size_t>(i);
        const double d = theta_plus[u] - theta_minus[u];
        dot_minus += d * r_minus[u];
        dot_plus += d * r_plus[u];
    }
    return dot_minus >= 0 && dot_plus >= 0;
}

void HipNUTSSampler::buildTree(const Vec& theta, const Vec& r, double log_u_slice, int v, int j, double epsilon, double H0,
                               Tree& tree) {  // :321-406
    if (j == 0) {
        Vec theta_prime = theta, r_prime = r, grad;
        leapfrog(theta_prime, r_prime, v * epsilon);
        const double log_p = gradient(theta_prime, grad);  // the vector leapfrog() just evaluated
        const double H_prime = log_p - 0.5 * dot(r_prime, r_prime);
        tree.n_valid = (log_u_slice <= H_prime) ? 1 : 0;
        tree.s = (log_u_slice < H_prime + DELTA_MAX);
        tree.theta_minus = theta_prime;
        tree.theta_plus = theta_prime;
        tree.r_minus = r_prime;
        tree.r_plus = r_prime;
        tree.theta_prime = theta_prime;
        tree.alpha = std::min(1.0, std::exp(H_prime - H0));
        tree.n_alpha = 1;
        return;
    }
    Tree left;
    buildTree(theta, r, log_u_slice, v, j - 1, epsilon, H0, left);
    if (!left.s) {
        tree = left;
        return;
    }
    Tree right;
    if (v == -1) {
        buildTree(left.theta_minus, left.r_minus, log_u_slice, v, j - 1, epsilon, H0, right);
        tree.theta_minus = right.theta_minus;
        tree.r_minus = right.r_minus;
        tree.theta_plus = left.theta_plus;
        tree.r_plus = left.r_plus;
    } else {
        buildTree(left.theta_plus, left.r_plus, log_u_slice, v, j - 1, epsilon, H0, right);
        tree.theta_minus = left.theta_minus;
        tree.r_minus = left.r_minus;
        tree.theta_plus = right.theta_plus;
        tree.r_plus = right.r_plus;
    }
    if (right.s) {
        tree.n_valid = left.n_valid + right.n_valid;
        const double prob = tree.n_valid > 0 ? static_cast<double>(right.n_valid) / static_cast<double>(tree.n_valid) : 0.0;
        if (std::uniform_real_distribution<>(0.0, 1.0)(rng_) < prob) tree.theta_prime = right.theta_prime;
        else tree.theta_prime = left.theta_prime;
        tree.alpha = left.alpha + right.alpha;
        tree.n_alpha = left.n_alpha + right.n_alpha;
        const bool no_uturn = checkNoUTurn(tree.theta_minus, tree.theta_plus, tree.r_minus, tree.r_plus);
        tree.s = left.s && right.s && no_uturn;
    } else {
        tree.theta_prime = left.theta_prime;
        tree.n_valid = left.n_valid;
        tree.s = false;
        tree.alpha = left.alpha;
        tree.n_alpha = left.n_alpha;
    }
}

double HipNUTSSampler::findReasonableEpsilon(const Vec& theta) {  // :230-287
    double avg_scale = 0.0;
    for (int i = 0; i < P_; ++i) avg_scale += pm_->getSigmaForParamIndex(i);
    avg_scale /= P_;
    double epsilon = avg_scale * 0.1;
    epsilon = std::max(1e-6, std::min(epsilon, 0.1));
    std::normal_distribution<> normal(0.0, 1.0);
    Vec r(static_cast<size_t>(P_)), grad;
    for (int i = 0; i < P_; ++i) r[static_cast<size_t>(i)] = normal(rng_);
    const double log_p = gradient(theta, grad);
    if (!std::isfinite(log_p)) return epsilon;
    const double H0 = log_p - 0.5 * dot(r, r);
    Vec theta_prime = theta, r_prime = r;
    leapfrog(theta_prime, r_prime, epsilon);
    double log_p_prime = gradient(theta_prime, grad);
    double H_prime = log_p_prime - 0.5 * dot(r_prime, r_prime);
    double accept_prob = std::exp(std::min(0.0, H_prime - H0));
    for (int iter = 0; iter < 5; ++iter) {
        if (accept_prob < 0.1 && epsilon > 1e-8) epsilon *= 0.5;
        else if (accept_prob > 0.9 && epsilon < 1.0) epsilon *= 1.5;
        else break;
        theta_prime = theta;
        r_prime = r;
        leapfrog(theta_prime, r_prime, epsilon);
        log_p_prime = gradient(theta_prime, grad);
        if (!std::isfinite(log_p_prime)) {
            epsilon *= 0.5;
            continue;
        }
        H_prime = log_p_prime - 0.5 * dot(r_prime, r_prime);
        accept_prob = std::exp(std::min(0.0, H_prime - H0));
    }
    return epsilon;
}

OptimizationResult HipNUTSSampler::optimize(const Eigen::VectorXd& initialParameters, IObjectiveFunction& objectiveFunction,
                                            IParameterManager& parameterManager) {
    grad_obj_ = dynamic_cast<IGradientObjectiveFunction*>(&objectiveFunction);
    if (!grad_obj_)  // :51-54
        throw InvalidParameterException("NUTSSampler", "Objective function must implement IGradientObjectiveFunction for NUTS.");
    pm_ = &parameterManager;
    P_ = static_cast<int>(initialParameters.size());
    rng_.seed(seed_);
    recent_.clear();
    epsilon_trace_.clear();
    depth_trace_.clear();
    gradient_calls_ = gradient_launches_ = 0;

    OptimizationResult result;
    Vec theta_m(static_cast<size_t>(P_));
    for (int i = 0; i < P_; ++i) theta_m[static_cast<size_t>(i)] = initialParameters[i];
    double epsilon = findReasonableEpsilon(theta_m);
    const double mu = std::log(10.0 * epsilon);  // dual averaging, :66-72
    double epsilon_bar = epsilon, H_bar = 0.0;
    const double gamma = 0.05, t0 = 10.0, kappa = 0.75;

    for (int m = 1; m <= num_iterations_; ++m) {
        std::normal_distribution<> normal(0.0, 1.0);
        Vec r0(static_cast<size_t>(P_)), grad;
        for (int i = 0; i < P_; ++i) r0[static_cast<size_t>(i)] = normal(rng_);
        const double log_p = gradient(theta_m, grad);
        clip_gradient(grad, MAX_GRAD_NORM);
        if (!std::isfinite(log_p)) {  // :101-108
            if (!result.samples.empty()) {
                result.samples.push_back(result.samples.back());
                result.sampleObjectiveValues.push_back(result.sampleObjectiveValues.back());
                epsilon_trace_.push_back(epsilon);
                depth_trace_.push_back(-1);
            }
            continue;
        }
        const double H0 = log_p - 0.5 * dot(r0, r0);
        const double log_u_slice = H0 - std::exponential_distribution<>(1.0)(rng_);
        Vec theta_minus = theta_m, theta_plus = theta_m, r_minus = r0, r_plus = r0, theta_next = theta_m;
        int j = 0, n = 1, n_alpha = 0;
        bool s = true;
        double alpha = 0.0;
        while (s && j < max_tree_depth_) {  // :133-163
            const int v = (std::uniform_int_distribution<>(0, 1)(rng_) * 2) - 1;
            Tree subtree;
            if (v == -1) {
                buildTree(theta_minus, r_minus, log_u_slice, v, j, epsilon, H0, subtree);
                theta_minus = subtree.theta_minus;
                r_minus = subtree.r_minus;
            } else {
                buildTree(theta_plus, r_plus, log_u_slice, v, j, epsilon, H0, subtree);
                theta_plus = subtree.theta_plus;
                r_plus = subtree.r_plus;
            }
            if (subtree.s && checkNoUTurn(theta_minus, theta_plus, r_minus, r_plus)) {
                const double acceptance_prob = static_cast<double>(subtree.n_valid) / static_cast<double>(n + subtree.n_valid);
                if (std::uniform_real_distribution<>(0.0, 1.0)(rng_) < acceptance_prob) theta_next = subtree.theta_prime;
                n += subtree.n_valid;
                alpha += subtree.alpha;
                n_alpha += subtree.n_alpha;
                j++;
            } else {
                s = false;
            }
        }
        theta_m = theta_next;
        if (m <= adaptation_window_) {  // :166-183
            const double avg_alpha = n_alpha > 0 ? alpha / n_alpha : 0.0;
            const double eta = 1.0 / (m + t0);
            H_bar = (1.0 - eta) * H_bar + eta * (delta_target_ - avg_alpha);
            const double log_epsilon = mu - (std::sqrt(m) / gamma) * H_bar;
            epsilon = std::exp(log_epsilon);
            const double m_kappa = std::pow(m, -kappa);
            const double log_epsilon_bar = m_kappa * log_epsilon + (1.0 - m_kappa) * std::log(epsilon_bar);
            epsilon_bar = std::exp(log_epsilon_bar);
        } else {
            epsilon = epsilon_bar;
        }
        Eigen::VectorXd th(P_);
        for (int i = 0; i < P_; ++i) th[i] = theta_m[static_cast<size_t>(i)];
        const Eigen::VectorXd constrained_theta = parameterManager.applyConstraints(th);
        result.samples.push_back(constrained_theta);
        const double final_obj = objectiveFunction.calculate(constrained_theta);
        result.sampleObjectiveValues.push_back(final_obj);
        if (final_obj > result.bestObjectiveValue) {
            result.bestObjectiveValue = final_obj;
            result.bestParameters = constrained_theta;
        }
        epsilon_trace_.push_back(epsilon);
        depth_trace_.push_back(j);
    }
    result.additionalStats["gradient_calls"] = static_cast<double>(gradient_calls_);
    result.additionalStats["gradient_launches"] = static_cast<double>(gradient_launches_);
    return result;
}

}  // namespace epidemic
