// host/include/epidemic_hip/HipPosteriorEnsemble.hpp
//
// Post-calibration ensemble on the device: the second consumer of the integrator
// (SURVEY 8f rank 1).  Mirrors, for this path only,
//   ResultAggregator::aggregatePosteriorPredictives   src/model/ResultAggregator.cpp:174-396
//     (result type PosteriorPredictiveData, include/model/AnalysisTypes.hpp:44-62)
//   the seroprevalence aggregation of PostCalibrationAnalyser::analyzeMCMCRunsInBatches
//     src/model/PostCalibrationAnalyser.cpp:209-246,303-343 (AggregatedStats per time point)
// Every selected sample is simulated from the GIVEN initial state, as
// SimulationRunner::runSimulation does (src/model/SimulationRunner.cpp:24-104), in ONE device
// launch; the per-time quantiles come from an exact sort on the device.  Differences from the
// reference, by design: the incidence quantiles use the exact-sort rule of
// PostCalibrationAnalyser.cpp:316-326 instead of Boost.Accumulators' order-dependent P^2 estimate;
// the Rt values are the Perron root of the n x n block that carries the next-generation matrix's spectrum
// (power iteration) instead of Eigen::EigenSolver on the 4n x 4n matrix: same number to rounding.
#pragma once
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "epidemic_hip/HipSEPAIHRD.hpp"

namespace epidemic {

struct PosteriorPredictiveData {
    std::vector<double> time_points;  // the output times >= 0
    struct IncidenceData {
        Eigen::MatrixXd median, lower_90, upper_90, lower_95, upper_95;  // T_pos x n_age
        Eigen::MatrixXd observed;
    };
    IncidenceData daily_hospitalizations, daily_icu_admissions, daily_deaths;
    IncidenceData cumulative_hospitalizations, cumulative_icu_admissions, cumulative_deaths;
    int samples_used = 0;  // simulations that were valid (build-side addition)
};

using AggregatedStats = std::map<std::string, double>;  // "median", "q025", "q975", "q05", "q95" (+ "mean", "std_dev")

// include/model/AnalysisTypes.hpp:14-39 (kappa_values omitted: they are the sample's own parameters)
struct EssentialMetrics {
    double R0 = 0.0, overall_IFR = 0.0, overall_attack_rate = 0.0, peak_hospital_occupancy = 0.0, peak_ICU_occupancy = 0.0,
           time_to_peak_hospital = 0.0, time_to_peak_ICU = 0.0, total_cumulative_deaths = 0.0;
    double max_Rt = 0.0, min_Rt = 1e6, final_Rt = 0.0, seroprevalence_at_target_day = 0.0;
    std::vector<double> age_specific_IFR, age_specific_IHR, age_specific_IICUR, age_specific_attack_rate;
};

class HipPosteriorEnsemble {
public:
    HipPosteriorEnsemble(HipSEPAIHRDParameterManager& parameterManager, const CalibrationData& observed_data,
                         const std::vector<double>& time_points, const Eigen::VectorXd& initial_state,
                         std::shared_ptr<IOdeSolverStrategy> solver_strategy, double abs_error = 1.0e-6,
                         double rel_error = 1.0e-6, int device = -1, bool fma_arithmetic = false);

    // sample selection as ResultAggregator.cpp:246-266: num_samples_for_ppc draws WITH replacement from
    // mt19937(random_seed) + uniform_int_distribution when 0 < num < size, else every sample in order.
    // random_seed = 0 means std::random_device in the reference; here it is rejected (reproducibility).
    static std::vector<int> selectSamples(size_t n_samples, int num_samples_for_ppc, unsigned int random_seed);

    PosteriorPredictiveData aggregatePosteriorPredictives(const std::vector<Eigen::VectorXd>& param_samples,
                                                          int num_samples_for_ppc, unsigned int random_seed);

    // samples burn_in, burn_in + thinning, ... (PostCalibrationAnalyser.cpp:209); one entry per output time
    std::map<double, AggregatedStats> aggregateSeroprevalence(const std::vector<Eigen::VectorXd>& param_samples,
                                                              int burn_in, int thinning);
    // effective reproduction number per output time over the same samples ("Rt_aggregated_with_uncertainty",
    // PostCalibrationAnalyser.cpp:233-236,342; MetricsCalculator::calculateRtTrajectory :172-197)
    std::map<double, AggregatedStats> aggregateRt(const std::vector<Eigen::VectorXd>& param_samples, int burn_in,
                                                  int thinning);
    // one EssentialMetrics row per valid simulation of the same samples (MetricsCalculator.cpp:8-170: what the
    // reference writes to mcmc_batches/batch_k.csv), and its summary table with the metric names of
    // ResultAggregator::aggregateBatchMetrics (:35-85): mean, std_dev, exact median / q025 / q975 over ALL rows
    // instead of P^2 per batch pooled by aggregateAllBatches (:87-172)
    std::vector<EssentialMetrics> calculateEssentialMetrics(const std::vector<Eigen::VectorXd>& param_samples, int burn_in,
                                                            int thinning);
    static std::map<std::string, AggregatedStats> aggregateMetrics(const std::vector<EssentialMetrics>& rows);

private:
    void run(const std::vector<double>& thetas, int S, bool want_sero);
    HipSEPAIHRDParameterManager& pm_;
    const CalibrationData& data_;
    std::vector<double> time_points_;
    SimulationCache cache_;
    std::unique_ptr<HipSEPAIHRDObjectiveFunction> objective_;
    int n_ = 0, t_pos_ = 0;
    std::vector<double> ppc_, sero_, rt_, metrics_;  // [6][5][T_pos][n], [5][T], [5][T], [S][12 + 4 n]
    int metrics_rows_ = 0;
    int n_valid_ = 0;
};

}  // namespace epidemic
