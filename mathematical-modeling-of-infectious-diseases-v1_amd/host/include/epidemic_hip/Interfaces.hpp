// Interfaces.hpp -- the reference's plug-in surface for the likelihood path, restated so that the
// HIP-backed classes are drop-in replacements.  Signatures follow (paths under /root/reference):
//   IObjectiveFunction      include/sir_age_structured/interfaces/IObjectiveFunction.hpp:13-31
//   IParameterManager       include/sir_age_structured/interfaces/IParameterManager.hpp:17-82
//   IOptimizationAlgorithm  include/sir_age_structured/interfaces/IOptimizationAlgorithm.hpp:18-54
//   ISimulationCache        include/sir_age_structured/interfaces/ISimulationCache.hpp:13-62
//   IOdeSolverStrategy      include/sir_age_structured/interfaces/IOdeSolverStrategy.hpp:18-43
//   IEpidemicModel          include/sir_age_structured/interfaces/IEpidemicModel.hpp:24-92
//   INpiStrategy            include/model/interfaces/INpiStrategy.hpp:14-65
//   exceptions              include/exceptions/Exceptions.hpp:18-174
// Inside the reference tree these declarations are replaced by the reference's own headers (see
// INTEGRATION.md); nothing here adds or changes a virtual.
#pragma once
#include <functional>
#include <limits>
#include <map>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "DenseCompat.hpp"

namespace epidemic {

class ModelException : public std::runtime_error {
public:
    ModelException(const std::string& where, const std::string& what)
        : std::runtime_error("[" + where + "] " + what) {}
};
class InvalidParameterException : public ModelException { using ModelException::ModelException; };
class SimulationException : public ModelException { using ModelException::ModelException; };

// The model-side surface (six pure virtuals).  On the device path computeDerivatives is never called -- the RHS
// runs inside the HIP kernel -- but the objective's constructor takes a model through this interface, like the
// reference's, and reads its parameters (AgeSEPAIHRDModel.hpp).
class IEpidemicModel {
public:
    virtual ~IEpidemicModel() = default;
    virtual void computeDerivatives(const std::vector<double>& state, std::vector<double>& derivatives, double time) = 0;
    virtual void applyIntervention(const std::string& name, double time, const Eigen::VectorXd& params) = 0;
    virtual void reset() = 0;
    virtual int getStateSize() const = 0;
    virtual std::vector<std::string> getStateNames() const = 0;
    virtual int getNumAgeClasses() const = 0;
};

class INpiStrategy {
public:
    virtual ~INpiStrategy() = default;
    virtual double getReductionFactor(double time) const = 0;
    virtual const std::vector<double>& getEndTimes() const = 0;
    virtual std::vector<double> getValues() const = 0;
    virtual double getBaselineKappa() const = 0;
    virtual double getBaselinePeriodEndTime() const = 0;
    virtual void setValues(const std::vector<double>& new_values) = 0;
    virtual std::shared_ptr<INpiStrategy> clone() const = 0;
    virtual double getLowerBoundForParamIndex(int idx) const = 0;
    virtual double getUpperBoundForParamIndex(int idx) const = 0;
};

class IObjectiveFunction {
public:
    virtual ~IObjectiveFunction() = default;
    virtual double calculate(const Eigen::VectorXd& parameters) const = 0;
    virtual const std::vector<std::string>& getParameterNames() const = 0;
};

// Build-side addition: B independent calculate() calls in one device launch.  thetas is B x P,
// chain-major (one parameter vector after another); out receives B objective values.  Per-chain
// failures follow calculate(): lowest() for invalid theta; an integration failure of any chain
// throws SimulationException after the whole batch has been evaluated (status[] tells which).
class IBatchObjectiveFunction {
public:
    virtual ~IBatchObjectiveFunction() = default;
    virtual void calculateBatch(const double* thetas, int B, double* out, int* status = nullptr) const = 0;
};

// include/model/interfaces/IGradientObjectiveFunction.hpp
class IGradientObjectiveFunction : public virtual IObjectiveFunction {
public:
    virtual ~IGradientObjectiveFunction() = default;
    virtual double evaluate_with_gradient(const Eigen::VectorXd& params, Eigen::VectorXd& grad) const = 0;
};

class IParameterManager {
public:
    virtual ~IParameterManager() = default;
    virtual Eigen::VectorXd getCurrentParameters() const = 0;
    virtual void updateModelParameters(const Eigen::VectorXd& parameters) = 0;
    virtual const std::vector<std::string>& getParameterNames() const = 0;
    virtual size_t getParameterCount() const = 0;
    virtual double getSigmaForParamIndex(int index) const = 0;
    virtual Eigen::VectorXd applyConstraints(const Eigen::VectorXd& parameters) const = 0;
    virtual int getIndexForParam(const std::string& name) const = 0;
    virtual double getLowerBoundForParamIndex(int idx) const = 0;
    virtual double getUpperBoundForParamIndex(int idx) const = 0;
};

class ISimulationCache {
public:
    virtual ~ISimulationCache() = default;
    virtual std::optional<double> get(const Eigen::VectorXd& parameters) = 0;
    virtual void set(const Eigen::VectorXd& parameters, double result) = 0;
    virtual void clear() = 0;
    virtual size_t size() const = 0;
    virtual std::string createCacheKey(const Eigen::VectorXd& parameters) const = 0;
    virtual bool getLikelihood(const std::string& key, double& value) = 0;
    virtual void storeLikelihood(const std::string& key, double value) = 0;
};

using state_type = std::vector<double>;
// The solver is selected by the DYNAMIC TYPE of the strategy object, as in the reference.  On the
// device path integrate() is never called (the stepper runs inside the HIP kernel); the objects are
// selectors.  A host integrate() is deliberately absent: there is no CPU fallback.
class IOdeSolverStrategy {
public:
    virtual ~IOdeSolverStrategy() = default;
    virtual void integrate(const std::function<void(const state_type&, state_type&, double)>& system,
                           state_type& initial_state, const std::vector<double>& times, double dt_hint,
                           std::function<void(const state_type&, double)> observer, double abs_error,
                           double rel_error) const = 0;
};
class Dopri5SolverStrategy : public IOdeSolverStrategy {
public:
    void integrate(const std::function<void(const state_type&, state_type&, double)>&, state_type&,
                   const std::vector<double>&, double, std::function<void(const state_type&, double)>, double,
                   double) const override {
        throw SimulationException("Dopri5SolverStrategy::integrate",
                                  "host integration is not built: this strategy selects the HIP Dopri5 kernel");
    }
};
class CashKarpSolverStrategy : public IOdeSolverStrategy {
public:
    void integrate(const std::function<void(const state_type&, state_type&, double)>&, state_type&,
                   const std::vector<double>&, double, std::function<void(const state_type&, double)>, double,
                   double) const override {
        throw SimulationException("CashKarpSolverStrategy::integrate",
                                  "host integration is not built: this strategy selects the HIP Cash-Karp kernel");
    }
};

struct OptimizationResult {
    Eigen::VectorXd bestParameters;
    double bestObjectiveValue = -std::numeric_limits<double>::infinity();
    std::vector<Eigen::VectorXd> samples;
    std::vector<double> sampleObjectiveValues;
    std::map<std::string, double> additionalStats;
    Eigen::MatrixXd finalCovariance;
};

class IOptimizationAlgorithm {
public:
    virtual ~IOptimizationAlgorithm() = default;
    virtual OptimizationResult optimize(const Eigen::VectorXd& initialParameters,
                                        IObjectiveFunction& objectiveFunction,
                                        IParameterManager& parameterManager) = 0;
    virtual void configure(const std::map<std::string, double>& settings) = 0;
};

}  // namespace epidemic
