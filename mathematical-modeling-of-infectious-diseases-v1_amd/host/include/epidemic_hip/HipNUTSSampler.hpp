// host/include/epidemic_hip/HipNUTSSampler.hpp
//
// NUTSSampler (src/model/optimizers/NUTSSampler.cpp:16-428, include/model/optimizers/NUTSSampler.hpp) over the
// device gradient objective: every evaluate_with_gradient() is the centre value plus ONE launch of the P
// perturbed simulations (HipSEPAIHRDGradientObjectiveFunction).  The algorithm, its constants and the order of
// its random draws are the reference's.  Two build-side points:
//   * the reference evaluates the gradient THREE times per tree leaf -- twice inside leapfrog() (:298, :311) and
//     once more at the leaf's end point (:341) -- and the third call, like the first call of the NEXT leaf of the
//     same subtree, repeats a parameter vector just evaluated.  The objective is a pure function of theta, so
//     those repeats are served from the last evaluations (compared bit for bit): the same numbers for about a
//     third of the launches.  gradientCalls() counts the reference's calls, gradientLaunches() the ones that ran.
//   * rng_ is seeded from std::random_device in the reference (:21); configure() takes `seed`.
#pragma once
#include <cstdint>
#include <deque>
#include <map>
#include <random>
#include <string>
#include <vector>

#include "epidemic_hip/Interfaces.hpp"

namespace epidemic {

class HipNUTSSampler : public IOptimizationAlgorithm {
public:
    void configure(const std::map<std::string, double>& settings) override;
    OptimizationResult optimize(const Eigen::VectorXd& initialParameters, IObjectiveFunction& objectiveFunction,
                                IParameterManager& parameterManager) override;
    const std::vector<double>& epsilonTrace() const { return epsilon_trace_; }  // step size after every iteration
    const std::vector<int>& depthTrace() const { return depth_trace_; }        // tree depth reached
    long gradientCalls() const { return gradient_calls_; }
    long gradientLaunches() const { return gradient_launches_; }

private:
    using Vec = std::vector<double>;
    struct Tree {
        Vec theta_minus, theta_plus, r_minus, r_plus, theta_prime;
        int n_valid = 0;
        bool s = false;
        double alpha = 0.0;
        int n_alpha = 0;
    };
    double gradient(const Vec& theta, Vec& grad);
    double findReasonableEpsilon(const Vec& theta);
    void leapfrog(Vec& theta, Vec& r, double epsilon);
    void buildTree(const Vec& theta, const Vec& r, double log_u_slice, int v, int j, double epsilon, double H0, Tree& tree);
    bool checkNoUTurn(const Vec& theta_minus, const Vec& theta_plus, const Vec& r_minus, const Vec& r_plus) const;
    double dot(const Vec& a, const Vec& b) const;

    int num_iterations_ = 2000, adaptation_window_ = 500, max_tree_depth_ = 10;
    double delta_target_ = 0.8;
    uint32_t seed_ = 1;
    static constexpr double DELTA_MAX = 1000.0;      // NUTSSampler.hpp:124
    static constexpr double MAX_GRAD_NORM = 1000.0;  // NUTSSampler.cpp:296

    // run state
    int P_ = 0;
    std::mt19937 rng_;
    IGradientObjectiveFunction* grad_obj_ = nullptr;
    IParameterManager* pm_ = nullptr;
    struct Evaluated { Vec theta, grad; double value; };
    std::deque<Evaluated> recent_;  // the last evaluations, newest first
    std::vector<double> epsilon_trace_;
    std::vector<int> depth_trace_;
    long gradient_calls_ = 0, gradient_launches_ = 0;
};

}  // namespace epidemic
