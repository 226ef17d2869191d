// AgeSEPAIHRDModel.hpp -- the MODEL-side types the reference's constructors take, so that the HIP-backed
// objective and parameter manager are built with the reference's own argument lists:
//
//   SEPAIHRDParameters             include/model/parameters/SEPAIHRDParameters.hpp:20-124 (fields on this path)
//   PiecewiseConstantNpiStrategy   include/model/PieceWiseConstantNPIStrategy.hpp, src/model/PieceWiseConstantNPIStrategy.cpp
//   AgeSEPAIHRDModel               include/model/AgeSEPAIHRDModel.hpp, src/model/AgeSEPAIHRDModel.cpp:230-368
//
// Inside the reference tree the reference's own headers define these names and this file is not used (the adapter
// reads a model only through getNumAgeClasses(), getModelParameters(), setModelParameters() and getNpiStrategy()).
// Here they are holders of the parameters: the derivative evaluation itself exists only as the HIP kernel
// (csrc/sepaihrd_kernels.hip), and computeDerivatives() says so instead of offering a CPU path.
#pragma once
#include <algorithm>
#include <map>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "Interfaces.hpp"

namespace epidemic {

struct SEPAIHRDParameters {
    Eigen::VectorXd N;
    Eigen::MatrixXd M_baseline;
    double contact_matrix_scaling_factor = 1.0;
    double beta = 0.0;
    std::vector<double> beta_end_times, beta_values;
    Eigen::VectorXd a, h_infec;
    double theta = 0, sigma = 0, gamma_p = 0, gamma_A = 0, gamma_I = 0, gamma_H = 0, gamma_ICU = 0;
    Eigen::VectorXd p, h, icu, d_H, d_ICU, d_community;
    std::vector<double> kappa_end_times, kappa_values;  // baseline period first
    double E0_multiplier = 1, P0_multiplier = 1, A0_multiplier = 1, I0_multiplier = 1, H0_multiplier = 1,
           ICU0_multiplier = 1, R0_multiplier = 1, D0_multiplier = 1;
    double runup_days = 30.0, seed_exposed = 10.0;
};

// kappa(t): a fixed (or calibratable) baseline value up to baseline_period_end_time, then one value per period,
// period k ending at end_times[k] inclusive, the last value for ever after.
class PiecewiseConstantNpiStrategy : public INpiStrategy {
public:
    PiecewiseConstantNpiStrategy(const std::vector<double>& npi_end_times_after_baseline,
                                 const std::vector<double>& npi_values_after_baseline,
                                 const std::map<std::string, std::pair<double, double>>& param_specific_bounds = {},
                                 double baseline_kappa = 1.0, double baseline_period_end_time = 13.0,
                                 bool fixed_baseline = true, const std::vector<std::string>& param_names_for_npi_values = {})
        : ends_(npi_end_times_after_baseline), values_(npi_values_after_baseline), bounds_(param_specific_bounds),
          baseline_(baseline_kappa), baseline_end_(baseline_period_end_time), fixed_(fixed_baseline),
          names_(param_names_for_npi_values) {
        const char* W = "PiecewiseConstantNpiStrategy";
        if (baseline_end_ < 0.0) throw InvalidParameterException(W, "Baseline period end time must be non-negative.");
        if (ends_.size() != values_.size()) throw InvalidParameterException(W, "NPI end times vector size must match NPI values vector size.");
        if (!names_.empty() && names_.size() != values_.size())
            throw InvalidParameterException(W, "Explicit NPI parameter names vector size must match NPI values vector size if provided.");
        double prev = baseline_end_;
        for (double t : ends_) {
            if (!(t > prev)) throw InvalidParameterException(W, "NPI end times must increase strictly from the baseline period's end.");
            prev = t;
        }
        if (baseline_ < 0.0) throw InvalidParameterException(W, "Baseline kappa value must be non-negative.");
        for (double k : values_)
            if (k < 0.0) throw InvalidParameterException(W, "NPI kappa values must be non-negative.");
        if (names_.empty())  // kappa_1 is the baseline
            for (size_t i = 0; i < values_.size(); ++i) names_.push_back("kappa_" + std::to_string(i + 2));
    }
    // PieceWiseConstantNPIStrategy.cpp:86-127 without its monotone-time cache (a pure function of t)
    double getReductionFactor(double time) const override {
        if (time < 0 || time <= baseline_end_ || ends_.empty()) return baseline_;
        const size_t idx = static_cast<size_t>(std::lower_bound(ends_.begin(), ends_.end(), time) - ends_.begin());
        return idx >= values_.size() ? values_.back() : values_[idx];
    }
    const std::vector<double>& getEndTimes() const override { return ends_; }
    std::vector<double> getValues() const override {  // baseline first
        std::vector<double> all(1, baseline_);
        all.insert(all.end(), values_.begin(), values_.end());
        return all;
    }
    double getBaselineKappa() const override { return baseline_; }
    double getBaselinePeriodEndTime() const override { return baseline_end_; }
    void setValues(const std::vector<double>& new_values) override {
        if (new_values.size() != values_.size())
            throw InvalidParameterException("PiecewiseConstantNpiStrategy::setValues", "New NPI values vector size must match existing number of changeable NPI periods.");
        for (double k : new_values)
            if (k < 0.0) throw InvalidParameterException("PiecewiseConstantNpiStrategy::setValues", "NPI kappa values must be non-negative.");
        values_ = new_values;
    }
    std::shared_ptr<INpiStrategy> clone() const override {
        return std::make_shared<PiecewiseConstantNpiStrategy>(ends_, values_, bounds_, baseline_, baseline_end_, fixed_, names_);
    }
    size_t getNumCalibratableNpiParams() const { return values_.size() + (fixed_ ? 0 : 1); }
    std::string getNpiParamName(int calibratable_idx) const {
        if (calibratable_idx < 0 || static_cast<size_t>(calibratable_idx) >= getNumCalibratableNpiParams())
            throw InvalidParameterException("PiecewiseConstantNpiStrategy::getNpiParamName", "calibratable_idx out of range.");
        if (!fixed_) return calibratable_idx == 0 ? std::string("kappa_baseline") : names_[static_cast<size_t>(calibratable_idx - 1)];
        return names_[static_cast<size_t>(calibratable_idx)];
    }
    bool isBaselineFixed() const { return fixed_; }
    double getLowerBoundForParamIndex(int idx) const override { return bound(idx).first; }
    double getUpperBoundForParamIndex(int idx) const override { return bound(idx).second; }
private:
    std::pair<double, double> bound(int idx) const {
        auto it = bounds_.find(getNpiParamName(idx));
        return it != bounds_.end() ? it->second : std::make_pair(0.0, 1.0);
    }
    std::vector<double> ends_, values_;
    std::map<std::string, std::pair<double, double>> bounds_;
    double baseline_, baseline_end_;
    bool fixed_;
    std::vector<std::string> names_;
};

class AgeSEPAIHRDModel final : public IEpidemicModel {
public:
    AgeSEPAIHRDModel(const SEPAIHRDParameters& params, std::shared_ptr<INpiStrategy> npi_strategy_ptr)
        : p_(params), npi_(std::move(npi_strategy_ptr)) {
        const char* W = "AgeSEPAIHRDModel";
        const Eigen::Index n = p_.N.size();
        if (n <= 0) throw InvalidParameterException(W, "Invalid parameters: population vector is empty.");
        if (!npi_) throw InvalidParameterException(W, "NPI strategy pointer cannot be null.");
        if (p_.M_baseline.rows() != n || p_.M_baseline.cols() != n) throw InvalidParameterException(W, "Contact matrix must be n x n.");
        for (const Eigen::VectorXd* v : {&p_.a, &p_.h_infec, &p_.p, &p_.h, &p_.icu, &p_.d_H, &p_.d_ICU})
            if (v->size() != n) throw InvalidParameterException(W, "Age-specific parameter vector of the wrong size.");
        if (p_.d_community.size() == 0) p_.d_community = Eigen::VectorXd::Zero(n);
        p_.kappa_end_times.clear();
        p_.kappa_values.clear();  // the schedule lives in the strategy
    }
    std::shared_ptr<AgeSEPAIHRDModel> clone() const { return std::make_shared<AgeSEPAIHRDModel>(p_, npi_->clone()); }

    void computeDerivatives(const std::vector<double>&, std::vector<double>&, double) override {
        throw SimulationException("AgeSEPAIHRDModel::computeDerivatives",
                                  "host derivative evaluation is not built: the model's right-hand side runs inside the HIP kernel");
    }
    void applyIntervention(const std::string& name, double, const Eigen::VectorXd&) override {
        throw ModelException("AgeSEPAIHRDModel::applyIntervention", "interventions are not part of the likelihood path: " + name);
    }
    void reset() override {}
    int getStateSize() const override { return 11 * getNumAgeClasses(); }
    std::vector<std::string> getStateNames() const override {  // AgeSEPAIHRDModel.cpp:251-259
        static const char* comp[11] = {"S", "E", "P", "A", "I", "H", "ICU", "R", "D", "CumH", "CumICU"};
        std::vector<std::string> names;
        for (const char* c : comp)
            for (int i = 0; i < getNumAgeClasses(); ++i) names.push_back(std::string(c) + std::to_string(i));
        return names;
    }
    int getNumAgeClasses() const override { return static_cast<int>(p_.N.size()); }

    std::shared_ptr<INpiStrategy> getNpiStrategy() const { return npi_; }
    // AgeSEPAIHRDModel.cpp:294-323: the struct carries the strategy's schedule, baseline period first
    SEPAIHRDParameters getModelParameters() const {
        SEPAIHRDParameters out = p_;
        out.kappa_end_times.assign(1, npi_->getBaselinePeriodEndTime());
        const std::vector<double>& ends = npi_->getEndTimes();
        out.kappa_end_times.insert(out.kappa_end_times.end(), ends.begin(), ends.end());
        out.kappa_values = npi_->getValues();
        return out;
    }
    // :325-363 -- everything but the NPI schedule, which setValues() of the strategy carries
    void setModelParameters(const SEPAIHRDParameters& params) {
        if (params.N.size() != p_.N.size()) throw InvalidParameterException("setModelParameters", "Size mismatch.");
        const Eigen::Index n = p_.N.size();
        p_ = params;
        if (p_.d_community.size() == 0) p_.d_community = Eigen::VectorXd::Zero(n);
        p_.kappa_end_times.clear();
        p_.kappa_values.clear();
    }
private:
    SEPAIHRDParameters p_;
    std::shared_ptr<INpiStrategy> npi_;
};

}  // namespace epidemic
