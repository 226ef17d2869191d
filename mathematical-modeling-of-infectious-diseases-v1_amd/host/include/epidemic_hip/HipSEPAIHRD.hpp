// HipSEPAIHRD.hpp -- host-side (C++) mirror of the reference's plug-in surface for the MCMC
// likelihood path, backed by the C ABI of include/sepaihrd_hip.h.
//
//   HipSEPAIHRDParameterManager   <- SEPAIHRDParameterManager
//                                    (include/model/parameters/SEPAIHRDParameterManager.hpp,
//                                     src/model/parameters/SEPAIHRDParameterManager.cpp)
//   HipSEPAIHRDObjectiveFunction  <- SEPAIHRDObjectiveFunction
//                                    (include/model/objectives/SEPAIHRDObjectiveFunction.hpp:49-58,
//                                     src/model/objectives/SEPAIHRDObjectiveFunction.cpp)
//   SimulationCache               <- src/sir_age_structured/caching/SimulationCache.cpp
//   MultiChainMetropolisHastings  <- MetropolisHastingsSampler
//                                    (src/sir_age_structured/optimizers/MetropolisHastingsSampler.cpp)
//                                    run for C independent chains in lock-step so that every
//                                    iteration is ONE batched device evaluation.
#pragma once
#include <cstdint>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <random>
#include <string>
#include <unordered_map>

#include "AgeSEPAIHRDModel.hpp"
#include "Interfaces.hpp"

struct sepaihrd_ctx;

namespace epidemic {

// SEPAIHRDParameters, PiecewiseConstantNpiStrategy, AgeSEPAIHRDModel: AgeSEPAIHRDModel.hpp

// include/model/parameters/SEPAIHRDParameterManager.hpp:22-25
enum class ConstraintMode { OPTIMIZATION_CLAMP = 0, MCMC_REFLECT = 1 };

// observed matrices as SEPAIHRDObjectiveFunction reads them from CalibrationData (T_obs x n)
class CalibrationData {
public:
    CalibrationData(const Eigen::MatrixXd& new_hospitalizations, const Eigen::MatrixXd& new_icu,
                    const Eigen::MatrixXd& new_deaths, const Eigen::VectorXd& population)
        : hosp_(new_hospitalizations), icu_(new_icu), deaths_(new_deaths), pop_(population) {}
    const Eigen::MatrixXd& getNewHospitalizations() const { return hosp_; }
    const Eigen::MatrixXd& getNewICU() const { return icu_; }
    const Eigen::MatrixXd& getNewDeaths() const { return deaths_; }
    const Eigen::VectorXd& getPopulationByAgeGroup() const { return pop_; }
private:
    Eigen::MatrixXd hosp_, icu_, deaths_;
    Eigen::VectorXd pop_;
};

class SimulationCache : public ISimulationCache {
public:
    explicit SimulationCache(size_t max_size = 1000);
    size_t computeHash(const Eigen::VectorXd& params) const;  // theta quantised to 1e-8
    bool getLikelihood(size_t key, double& value);
    void storeLikelihood(size_t key, double value);
    std::optional<double> get(const Eigen::VectorXd& parameters) override;
    void set(const Eigen::VectorXd& parameters, double result) override;
    void clear() override;
    size_t size() const override;
    std::string createCacheKey(const Eigen::VectorXd& parameters) const override;
    bool getLikelihood(const std::string& key, double& value) override;
    void storeLikelihood(const std::string& key, double value) override;
    size_t getLikelihoodCalls() const { return calls_; }
    size_t getLikelihoodHits() const { return hits_; }
private:
    struct Entry { double value; uint64_t freq, tick; };
    size_t capacity_;
    uint64_t tick_ = 0;
    size_t calls_ = 0, hits_ = 0;
    std::unordered_map<size_t, Entry> map_;
};

class HipSEPAIHRDParameterManager : public IParameterManager {
public:
    // The reference's constructor (include/model/parameters/SEPAIHRDParameterManager.hpp:40-45,
    // SEPAIHRDModelCalibration.cpp:84-87): parameters and calibratable kappa names are read from the model
    // (getModelParameters(), the PiecewiseConstantNpiStrategy's names); updateModelParameters() writes back into it.
    HipSEPAIHRDParameterManager(std::shared_ptr<AgeSEPAIHRDModel> model, const std::vector<std::string>& params_to_calibrate,
                                const std::map<std::string, double>& proposal_sigmas,
                                const std::map<std::string, std::pair<double, double>>& param_bounds);
    // npi_param_names: names of kappa_values[1..] (PiecewiseConstantNpiStrategy's calibratable names;
    // empty -> "kappa_2", "kappa_3", ...)
    HipSEPAIHRDParameterManager(const SEPAIHRDParameters& model_params,
                                const std::vector<std::string>& params_to_calibrate,
                                const std::map<std::string, double>& proposal_sigmas,
                                const std::map<std::string, std::pair<double, double>>& param_bounds,
                                const std::vector<std::string>& npi_param_names = {});
    Eigen::VectorXd getCurrentParameters() const override;
    void updateModelParameters(const Eigen::VectorXd& parameters) override;
    const std::vector<std::string>& getParameterNames() const override { return names_; }
    size_t getParameterCount() const override { return names_.size(); }
    double getSigmaForParamIndex(int index) const override;
    Eigen::VectorXd applyConstraints(const Eigen::VectorXd& parameters) const override;
    int getIndexForParam(const std::string& name) const override;
    double getLowerBoundForParamIndex(int idx) const override;
    double getUpperBoundForParamIndex(int idx) const override;
    void setConstraintMode(ConstraintMode m) { mode_ = m; }
    ConstraintMode getConstraintMode() const { return mode_; }
    // resolved theta -> field map (enum sepaihrd_field / index), done once
    const std::vector<int32_t>& fieldCodes() const { return field_; }
    const std::vector<int32_t>& fieldIndices() const { return index_; }
    const SEPAIHRDParameters& modelParameters() const { return params_; }
    bool hasBounds(int idx) const { return has_bounds_[static_cast<size_t>(idx)] != 0; }
private:
    double* slot(int field, int index);
    std::shared_ptr<AgeSEPAIHRDModel> model_;  // null for the struct-taking constructor
    SEPAIHRDParameters params_;
    std::vector<std::string> names_, npi_names_;
    std::vector<double> sigma_, lower_, upper_;
    std::vector<uint8_t> has_bounds_;
    std::vector<int32_t> field_, index_;
    ConstraintMode mode_ = ConstraintMode::OPTIMIZATION_CLAMP;
};

class HipSEPAIHRDObjectiveFunction : public virtual IObjectiveFunction, public IBatchObjectiveFunction {
public:
    // SEPAIHRDObjectiveFunction's constructor, argument for argument
    // (include/model/objectives/SEPAIHRDObjectiveFunction.hpp:49-58; built at SEPAIHRDModelCalibration.cpp:94-118).
    // parameterManager may be ANY IParameterManager whose names follow the reference's naming -- a
    // HipSEPAIHRDParameterManager is used as it is; for any other (the reference's own SEPAIHRDParameterManager) the
    // name -> field map is resolved here from getParameterNames(), the bounds and sigmas from its getters and the base
    // values from model->getModelParameters(), and its CURRENT constraint mode is read before every evaluation by
    // probing applyConstraints() with a vector beyond the bounds (the interface has no mode getter).
    // Device and arithmetic: environment SEPAIHRD_DEVICE (default: current device), SEPAIHRD_ARITH=fma|strict
    // (default fma -- the arithmetic bench.py's `value` is measured in; over the reference's own run length, 4096 chains x
    // 100 000 iterations, it takes every accept decision the way strict does: profiles/r04_fma_vs_strict_100k.json;
    // strict = the CPU build's operation sequence, the mode the oracle parity tests compare bit for bit).
    HipSEPAIHRDObjectiveFunction(std::shared_ptr<AgeSEPAIHRDModel> model, IParameterManager& parameterManager,
                                 ISimulationCache& cache, const CalibrationData& calibration_data,
                                 const std::vector<double>& time_points, const Eigen::VectorXd& initial_state,
                                 std::shared_ptr<IOdeSolverStrategy> solver_strategy, double abs_error = 1.0e-6,
                                 double rel_error = 1.0e-6);
    // Same argument order and meaning as SEPAIHRDObjectiveFunction's constructor; the model object is
    // the parameter manager's SEPAIHRDParameters (the device needs no host model instance).
    HipSEPAIHRDObjectiveFunction(HipSEPAIHRDParameterManager& parameterManager, ISimulationCache& cache,
                                 const CalibrationData& calibration_data,
                                 const std::vector<double>& time_points, const Eigen::VectorXd& initial_state,
                                 std::shared_ptr<IOdeSolverStrategy> solver_strategy, double abs_error = 1.0e-6,
                                 double rel_error = 1.0e-6, int device = -1, bool fma_arithmetic = true);
    ~HipSEPAIHRDObjectiveFunction() override;
    // the arithmetic the reference-shaped constructors select right now (environment SEPAIHRD_ARITH; fma unless "strict")
    static bool defaultArithmeticIsFma() { return environmentFma(); }
    HipSEPAIHRDObjectiveFunction(const HipSEPAIHRDObjectiveFunction&) = delete;
    HipSEPAIHRDObjectiveFunction& operator=(const HipSEPAIHRDObjectiveFunction&) = delete;

    double calculate(const Eigen::VectorXd& parameters) const override;
    const std::vector<std::string>& getParameterNames() const override;
    void calculateBatch(const double* thetas, int B, double* out, int* status = nullptr) const override;
    // the SimulationException a per-chain status >= SEPAIHRD_STATUS_STEP_FAILURE stands for (2 odeint's 500 rejections,
    // 3 step-attempt budget, 4 hand-off between wavefronts timed out); never returns
    [[noreturn]] static void throwIntegrationFailure(int status);
    // per-chain step counters of the last calculateBatch (diagnostics)
    const std::vector<int32_t>& lastAccepted() const { return n_acc_; }
    const std::vector<int32_t>& lastRejected() const { return n_rej_; }
    // build-side accessors for other device callers of the same problem (HipPosteriorEnsemble)
    sepaihrd_ctx* deviceContext() const { return ctx_; }
    void syncDeviceConstraintMode() const { syncConstraintMode(); }
protected:
    // builds the device context for this problem; multipliers_override (8 values) replaces the model's
    // E0..D0 multipliers as the base values of the non-calibrated ones
    static sepaihrd_ctx* createContext(const HipSEPAIHRDParameterManager& pm, const CalibrationData& data,
                                       const std::vector<double>& time_points, const Eigen::VectorXd& initial_state,
                                       const std::shared_ptr<IOdeSolverStrategy>& solver_strategy, double abs_error,
                                       double rel_error, int device, bool fma_arithmetic,
                                       const double* multipliers_override);
    void syncConstraintMode() const;
    // resolves the manager argument of the model-taking constructors: `owned` is filled when it is not a Hip manager
    static HipSEPAIHRDParameterManager& resolveManager(const std::shared_ptr<AgeSEPAIHRDModel>& model, IParameterManager& given,
                                                       std::unique_ptr<HipSEPAIHRDParameterManager>& owned);
    static int environmentDevice();
    static bool environmentFma();
    std::unique_ptr<HipSEPAIHRDParameterManager> owned_pm_;  // declared before pm_: it may be what pm_ refers to
    IParameterManager* foreign_pm_ = nullptr;                // the caller's manager when it is not a Hip one
    HipSEPAIHRDParameterManager& pm_;
    ISimulationCache& cache_;
    sepaihrd_ctx* ctx_ = nullptr;
    mutable int device_mode_ = -1;
    mutable std::vector<int32_t> n_acc_, n_rej_, status_;
};

// SEPAIHRDGradientObjectiveFunction (include/model/objectives/SEPAIHRDGradientObjectiveFunction.hpp,
// src/model/objectives/SEPAIHRDGradientObjectiveFunction.cpp:15-171): forward differences, all P
// perturbed simulations in ONE launch.  The perturbed runs follow that file's own rules, not
// calculate()'s: fresh clamp-mode constraints, initial state ALWAYS scaled by the multipliers (read
// unconstrained from the perturbed vector, 1.0 when not calibrated), invalid when the non-S total is
// above N or negative (entry 0), likelihood over ALL output rows (a row count different from the
// observations' makes every entry (lowest() - f) / eps, as in the reference).
class HipSEPAIHRDGradientObjectiveFunction : public HipSEPAIHRDObjectiveFunction, public IGradientObjectiveFunction {
public:
    // SEPAIHRDGradientObjectiveFunction's constructor (SEPAIHRDModelCalibration.cpp:96-104), as above
    HipSEPAIHRDGradientObjectiveFunction(std::shared_ptr<AgeSEPAIHRDModel> model, IParameterManager& parameterManager,
                                         ISimulationCache& cache, const CalibrationData& calibration_data,
                                         const std::vector<double>& time_points, const Eigen::VectorXd& initial_state,
                                         std::shared_ptr<IOdeSolverStrategy> solver_strategy, double abs_error = 1.0e-6,
                                         double rel_error = 1.0e-6);
    HipSEPAIHRDGradientObjectiveFunction(HipSEPAIHRDParameterManager& parameterManager, ISimulationCache& cache,
                                         const CalibrationData& calibration_data, const std::vector<double>& time_points,
                                         const Eigen::VectorXd& initial_state,
                                         std::shared_ptr<IOdeSolverStrategy> solver_strategy, double abs_error = 1.0e-6,
                                         double rel_error = 1.0e-6, int device = -1, bool fma_arithmetic = true);
    ~HipSEPAIHRDGradientObjectiveFunction() override;
    double epsilon_ = 1e-4;  // public in the reference too (:28)
    double evaluate_with_gradient(const Eigen::VectorXd& params, Eigen::VectorXd& grad) const override;
    double calculate(const Eigen::VectorXd& parameters) const override { return HipSEPAIHRDObjectiveFunction::calculate(parameters); }
    const std::vector<std::string>& getParameterNames() const override { return HipSEPAIHRDObjectiveFunction::getParameterNames(); }
private:
    void buildGradientContext(const CalibrationData& data, const std::vector<double>& time_points,
                              const std::shared_ptr<IOdeSolverStrategy>& solver_strategy, double abs_error, double rel_error,
                              int device, bool fma_arithmetic);
    bool initialStateValid(const double* plus) const;
    sepaihrd_ctx* grad_ctx_ = nullptr;
    Eigen::VectorXd initial_state_;
    size_t n_times_ = 0, n_obs_rows_ = 0;
    double first_time_ = 0.0;
    std::vector<int> mult_index_;  // parameter index of E0..D0 multiplier, -1 = not calibrated
};

// MetropolisHastingsSampler for many independent chains.  configure() takes the reference's
// settings keys (mcmc_iterations, burn_in, adaptation_period, thinning, regularization_epsilon,
// target_acceptance_rate, adapt_scale, store_samples).  A seed replaces the reference's
// std::random_device (build-side addition): chain c draws from std::mt19937(seed + c) with the
// reference's draw order (fresh normal_distribution per proposal, uniform only when log_ratio < 0).
class MultiChainMetropolisHastings : public IOptimizationAlgorithm {
public:
    void configure(const std::map<std::string, double>& settings) override;
    void setSeed(uint32_t seed) { seed_ = seed; }
    // MetropolisHastingsSampler::setInitialCovariance (MetropolisHastingsSampler.cpp:52-63): every chain
    // warm-starts from this covariance instead of diag(sigma^2) 2.38^2/P; a non-square matrix clears it
    void setInitialCovariance(const Eigen::MatrixXd& cov) {
        initial_cov_.clear();
        if (cov.rows() > 0 && cov.rows() == cov.cols())
            for (Eigen::Index a = 0; a < cov.rows(); ++a)
                for (Eigen::Index b = 0; b < cov.cols(); ++b) initial_cov_.push_back(cov(a, b));
    }
    // one chain through the scalar interface (drop-in for MetropolisHastingsSampler)
    OptimizationResult optimize(const Eigen::VectorXd& initialParameters, IObjectiveFunction& objectiveFunction,
                                IParameterManager& parameterManager) override;
    // C chains in lock-step; initial is C x P chain-major
    std::vector<OptimizationResult> optimizeChains(const std::vector<double>& initial, int C,
                                                   IBatchObjectiveFunction& objective,
                                                   IParameterManager& parameterManager);
    // The same sampler with the per-chain adaptation state (covariance, Cholesky factor, running mean,
    // chain history) resident on the device: the O(C t P^2) covariance refresh and the C t P doubles of
    // history stay next to the likelihood kernel, the host keeps the random streams, the accept test and
    // the scale adaptation.  Same numbers as optimizeChains (strict arithmetic: bit for bit).
    std::vector<OptimizationResult> optimizeChainsOnDevice(const std::vector<double>& initial, int C,
                                                           HipSEPAIHRDObjectiveFunction& objective,
                                                           IParameterManager& parameterManager);
    // optimizeChainsOnDevice for G = objectives.size() contiguous groups of chains, one host thread, one
    // device context and one stream per group: while one group waits for its evaluation the others draw,
    // decide and launch, so the device sees several evaluations in flight (a batch of a few thousand chains
    // leaves most SIMDs idle).  Chain c draws from mt19937(seed + c) whatever the grouping:
    // the results are those of one group.
    std::vector<OptimizationResult> optimizeChainGroupsOnDevice(const std::vector<double>& initial, int C,
                                                                const std::vector<HipSEPAIHRDObjectiveFunction*>& objectives,
                                                                IParameterManager& parameterManager);
    // Per-chain summary records of the last device-resident run, chain-major [C][2 P + 2]: posterior means and variances
    // over the stored samples after burn-in, the chain's best value, its accepted proposals (formed on the
    // device, sepaihrd_mh_summary_records).  They also stay on each objective's device (sepaihrd_records_buffer 0).
    const std::vector<double>& chainSummaries() const { return summary_records_; }
    // The one exchange of the path: after optimizeChainGroupsOnDevice, every group's device receives the records of ALL
    // chains (RCCL ncclAllGather over the groups' devices when they are distinct and librccl loads, staging through the host
    // otherwise), so that the ensemble statistics the reference forms serially (ResultAggregator.cpp:35-172) can be
    // formed next to the data on any device.  backend: SEPAIHRD_GATHER_AUTO / _RCCL / _HOST; returns the one used.
    int gatherChainSummaries(const std::vector<HipSEPAIHRDObjectiveFunction*>& objectives, int backend = 0);
    // the gathered table as one group's device holds it (after gatherChainSummaries)
    std::vector<double> gatheredSummaries(HipSEPAIHRDObjectiveFunction& objective) const;
    // exact-sort quantiles across chains of every record column, probs x width (the rule of the reference's trajectory
    // quantiles, PostCalibrationAnalyser.cpp:303-340: v[floor(pos)] (1 - f) + v[floor(pos) + 1] f, pos = q (n - 1))
    static std::vector<double> summaryQuantiles(const std::vector<double>& table, int width, const std::vector<double>& probs);
    void setHostThreads(int n) { host_threads_ = n; }  // per-run cap on the OpenMP team (0 = the CPU share)
    // recomputeFullCovariance (MetropolisHastingsSampler.cpp:168-199) as the reference writes it -- two passes over the
    // whole chain history, O(t P^2) per refresh and every state kept -- instead of the running co-moments the sampler
    // carries by default (O(P^2) per refresh, no history; same running mean bit for bit, covariance equal to ~1e-13)
    void setTwoPassCovariance(bool on) { two_pass_covariance_ = on; }
    // Where the chains' std::mt19937 streams are drawn in optimizeChainsOnDevice: on the device (default: mt19937,
    // generate_canonical, the polar method and glibc's log written out for it, csrc/sepaihrd_rng.inc -- the same values from
    // the same stream positions) or on the host (libstdc++ itself; what sets the pace beyond a few thousand chains)
    void setDeviceStreams(bool on) { device_streams_ = on; }
    // The device's log / exp restate one libm build (glibc 2.35, x86-64, FMA variants).  Before a run lets the device draw,
    // they are compared with this host's std::log / std::exp on 4096 fixed arguments (sepaihrd_device_libm_check); on any
    // difference the run keeps draws and scale adaptation on the host, says so through the progress sink, and this is true.
    bool deviceStreamsFellBack() const { return device_streams_fell_back_; }
    // Progress reports and trace files of MetropolisHastingsSampler::optimize (MetropolisHastingsSampler.cpp:363-383,399-411,
    // 440-469): every `report_interval` iterations the line "Iter: .. | LogPost: .. | Best: .. | AccRate: .. | Scale: .." and,
    // with `write_checkpoints`, posterior_trace_checkpoint.csv (the last <= 5000 thinned samples); at the end
    // posterior_trace_final.csv and, with `write_trace`, posterior_trace.csv -- all three settings keys are the reference's,
    // with its defaults (100, 1, 1; report_interval <= 0 switches the reports off).  The reference has ONE chain: here the
    // first `checkpoint_chains` chains (settings key, default 1) report and write; chain 0 uses the reference's file names,
    // chain k > 0 gets `_chain<k>` in front of `.csv`.  A device-resident run does not wait for any of this: the values
    // leave the device behind the iteration they belong to (sepaihrd_mh_snapshot_begin) and a writer thread formats them.
    // Directory: <project root>/data/mcmc_samples like the reference (FileUtils::getProjectRoot's rule) unless set here.
    void setOutputDirectory(const std::string& dir) { output_dir_ = dir; }
    // where the progress lines go (default: stdout, in the layout of the reference's Logger); level is INFO or WARNING
    void setProgressSink(std::function<void(const std::string& level, const std::string& message)> sink) { progress_sink_ = std::move(sink); }
    // evaluations the accept tests of the last device-resident run saw fail: status 2 (500 rejections), 3 (attempt budget),
    // 4 (hand-off guard of the evaluation kernel).  They count as -1e18 like a throwing objective; a non-zero last entry is
    // an error of this build and makes the run throw.
    const std::vector<long>& failureCounts() const { return failure_counts_; }
    const std::vector<std::vector<unsigned char>>& acceptTraces() const { return traces_; }
    void setKeepAcceptTraces(bool on) { keep_traces_ = on; }  // off for long runs of many chains (65 536 x 100 000 = 6.5 GB)
    // wall time of the iteration loop of the last device-resident run (proposal 1 staged .. last accept test), without
    // the set-up before it (history allocation, initial values) and the read-back after it; groups: the slowest group
    double lastLoopSeconds() const { return last_loop_seconds_; }
private:
    struct Chain;
    using BatchEval = std::function<void(const double*, int, double*)>;
    std::vector<OptimizationResult> run(const std::vector<double>& initial, int C, const BatchEval& eval,
                                        IParameterManager& pm);
    int iterations_ = 10000, burn_in_ = 1000, adaptation_period_ = 100, thinning_ = 1;
    double regularization_epsilon_ = 1e-6, target_acceptance_rate_ = 0.234;
    bool adapt_scale_ = true, store_samples_ = true;
    uint32_t seed_ = 1;
    std::vector<std::vector<unsigned char>> traces_;
    std::vector<double> initial_cov_;  // row-major P x P, empty = none
    int host_threads_ = 0;
    bool two_pass_covariance_ = false;
    bool device_streams_ = true;
    bool keep_traces_ = true;  // per-iteration accept flags of every chain (acceptTraces()): C x iterations bytes on the host
    int adaptation_window_ = 0;  // ring of newest states on the device (0: adaptation period + 1, at least 128)
    double last_loop_seconds_ = 0.0;
    std::vector<double> summary_records_;  // [C][2 P + 2] of the last device-resident run
    std::vector<int> group_rows_;          // chains per group of the last grouped run
    int summary_width_ = 0;
    int report_interval_ = 100, checkpoint_chains_ = 1;
    bool write_checkpoints_ = true, write_trace_ = true;
    std::string output_dir_;
    std::function<void(const std::string&, const std::string&)> progress_sink_;
    bool device_streams_fell_back_ = false;
    std::vector<long> failure_counts_;
    struct Reporter;
    friend struct Reporter;
};

}  // namespace epidemic
