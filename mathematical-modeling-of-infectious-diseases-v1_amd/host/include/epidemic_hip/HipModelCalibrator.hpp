// host/include/epidemic_hip/HipModelCalibrator.hpp
//
// ModelCalibrator (src/sir_age_structured/ModelCalibrator.cpp:13-159,
// include/sir_age_structured/ModelCalibrator.hpp) over the device objective: the two-phase
// calibration run of SEPAIHRDModelCalibration::runHillClimbingMCMC
// (src/model/SEPAIHRDModelCalibration.cpp:150-178) with both phases on batched launches --
// phase 1 BatchedHillClimbingOptimizer in OPTIMIZATION_CLAMP mode, conditioning of its covariance
// (:93-131), phase 2 MultiChainMetropolisHastings in MCMC_REFLECT mode warm-started from it, the
// objective value of every stored sample (:141-144) in one launch.
// `chains` > 1 runs that many phase-2 chains in lock-step from the phase-1 optimum (chain c draws
// from mt19937(seed + c); chain 0 is the reference's single chain).
// The eigen-decomposition the reference takes from Eigen::SelfAdjointEigenSolver is a cyclic Jacobi
// iteration here (eigenpairs ascending, like Eigen returns them).
#pragma once
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "epidemic_hip/BatchedHillClimbing.hpp"
#include "epidemic_hip/HipSEPAIHRD.hpp"

namespace epidemic {

// ModelCalibrator.cpp:93-131; cov and the result are P x P
Eigen::MatrixXd conditionPhase1Covariance(const Eigen::MatrixXd& cov, const IParameterManager& pm);

class HipModelCalibrator {
public:
    // evaluates the initial objective value like the reference's constructor (:36-45)
    HipModelCalibrator(HipSEPAIHRDParameterManager& parameterManager, HipSEPAIHRDObjectiveFunction& objective);
    // Phase-1 algorithm, as the map the reference's constructor takes under PHASE1_NAME (ModelCalibrator.hpp:60):
    // BatchedHillClimbingOptimizer unless set (runHillClimbingMCMC); BatchedParticleSwarmOptimization gives
    // SEPAIHRDModelCalibration::runPSOMCMC (SEPAIHRDModelCalibration.cpp:179-208).
    void setPhase1Algorithm(std::unique_ptr<IOptimizationAlgorithm> algorithm) { phase1_algo_ = std::move(algorithm); }
    void calibrate(const std::map<std::string, double>& phase1_settings,
                   const std::map<std::string, double>& phase2_settings, int chains = 1);
    const Eigen::VectorXd& getBestParameterVector() const { return best_; }
    double getBestObjectiveValue() const { return best_value_; }
    double getInitialObjectiveValue() const { return initial_value_; }
    const OptimizationResult& getPhase1Result() const { return phase1_; }
    const std::vector<OptimizationResult>& getPhase2Results() const { return phase2_; }
    const Eigen::MatrixXd& getPhase2Covariance() const { return phase2_cov_; }
    const std::vector<Eigen::VectorXd>& getMCMCSamples() const { return phase2_.empty() ? empty_ : phase2_[0].samples; }
    // objective value of every stored sample of every chain, chain-major (:141-144)
    const std::vector<double>& getMCMCObjectiveValues() const { return mcmc_values_; }
    const std::vector<std::vector<unsigned char>>& acceptTraces() const { return traces_; }
private:
    HipSEPAIHRDParameterManager& pm_;
    HipSEPAIHRDObjectiveFunction& obj_;
    std::unique_ptr<IOptimizationAlgorithm> phase1_algo_;
    Eigen::VectorXd best_;
    double best_value_ = 0.0, initial_value_ = 0.0;
    OptimizationResult phase1_;
    std::vector<OptimizationResult> phase2_;
    Eigen::MatrixXd phase2_cov_;
    std::vector<double> mcmc_values_;
    std::vector<std::vector<unsigned char>> traces_;
    std::vector<Eigen::VectorXd> empty_;
};

}  // namespace epidemic
