// host/include/epidemic_hip/BatchedHillClimbing.hpp
//
// HillClimbingOptimizer (src/sir_age_structured/optimizers/HillClimbingOptimizer.cpp:24-353,
// include/sir_age_structured/optimizers/HillClimbingOptimizer.hpp) with its objective calls
// grouped into device launches.  Per iteration the reference makes
//   num_candidates calls under `#pragma omp parallel for` (:222-228)      -> ONE batch
//   up to 10 backtracking calls, one after another (:60-75)               -> ONE batch: the candidates
//       depend only on the step length, the first improving one is taken
//   up to 12 expansion calls with a moving anchor (:89-104)               -> ONE batch: candidate i
//       depends only on candidate i-1 (not on its value) as long as every earlier one improved,
//       and the loop stops at the first that does not
// so the values consumed, and therefore the whole search path, are those of the one-call-at-a-time
// loop (tests compare with the CPU restatement of exactly that loop).
//
// configure() takes the reference's keys (iterations, report_interval, cloud_size_multiplier) plus two
// build-side ones: `threads` (the reference sizes the cloud by omp_get_max_threads(), :157-163, which
// means nothing for a device) and `seed` (the reference seeds from std::random_device, :112,172).
// Virtual thread t owns the candidates OpenMP's static schedule would give it, with its own
// mt19937 (seed_seq of four master draws) and its own persistent normal_distribution (:173-183).
#pragma once
#include <cstdint>
#include <functional>
#include <map>
#include <string>
#include <vector>

#include "epidemic_hip/Interfaces.hpp"

namespace epidemic {

class BatchedHillClimbingOptimizer : public IOptimizationAlgorithm {
public:
    void configure(const std::map<std::string, double>& settings) override;
    // scalar interface (any IObjectiveFunction); uses calculateBatch when the objective offers it
    OptimizationResult optimize(const Eigen::VectorXd& initialParameters, IObjectiveFunction& objectiveFunction,
                                IParameterManager& parameterManager) override;
    const std::vector<double>& currentTrace() const { return trace_; }   // current logL after each iteration
    long evaluations() const { return evaluations_; }                   // objective values requested
    long launches() const { return launches_; }                         // batches issued
private:
    using BatchEval = std::function<void(const double*, int, double*)>;
    OptimizationResult run(const Eigen::VectorXd& x0, const BatchEval& eval, IParameterManager& pm);
    int iterations_ = 2000, report_interval_ = 100, cloud_size_multiplier_ = 8, threads_ = 16;
    uint32_t seed_ = 1;
    std::vector<double> trace_;
    long evaluations_ = 0, launches_ = 0;
};

}  // namespace epidemic
