// host/src/HipPosteriorEnsemble.cpp -- see the header for the reference lines mirrored.
#include "epidemic_hip/HipPosteriorEnsemble.hpp"

#include <algorithm>
#include <cmath>
#include <functional>
#include <numeric>
#include <random>

#include "sepaihrd_hip.h"

namespace epidemic {

namespace {
// probabilities in the order the device call is made with; ResultAggregator.cpp:224
const double kProbs[5] = {0.025, 0.05, 0.5, 0.95, 0.975};
}  // namespace

HipPosteriorEnsemble::HipPosteriorEnsemble(HipSEPAIHRDParameterManager& parameterManager,
                                           const CalibrationData& observed_data,
                                           const std::vector<double>& time_points, const Eigen::VectorXd& initial_state,
                                           std::shared_ptr<IOdeSolverStrategy> solver_strategy, double abs_error,
                                           double rel_error, int device, bool fma_arithmetic)
    : pm_(parameterManager), data_(observed_data), time_points_(time_points), cache_(1) {
    objective_ = std::make_unique<HipSEPAIHRDObjectiveFunction>(pm_, cache_, data_, time_points_, initial_state,
                                                                std::move(solver_strategy), abs_error, rel_error, device,
                                                                fma_arithmetic);
    if (sepaihrd_set_initial_state_mode(objective_->deviceContext(), SEPAIHRD_INIT_FIXED) != SEPAIHRD_OK)
        throw ModelException("HipPosteriorEnsemble", "sepaihrd_set_initial_state_mode failed");
    n_ = static_cast<int>(pm_.modelParameters().N.size());
    for (double t : time_points_) t_pos_ += (t >= 0.0);
}

std::vector<int> HipPosteriorEnsemble::selectSamples(size_t n_samples, int num_samples_for_ppc, unsigned int random_seed) {
    std::vector<int> selected;
    if (num_samples_for_ppc > 0 && static_cast<size_t>(num_samples_for_ppc) < n_samples) {
        if (random_seed == 0)
            throw InvalidParameterException("HipPosteriorEnsemble", "random_seed 0 (random_device) is not reproducible; pass a seed");
        std::mt19937 gen;
        gen.seed(random_seed);
        std::uniform_int_distribution<> distrib(0, static_cast<int>(n_samples) - 1);
        selected.reserve(static_cast<size_t>(num_samples_for_ppc));
        for (int i = 0; i < num_samples_for_ppc; ++i) selected.push_back(distrib(gen));
    } else {
        selected.resize(n_samples);
        std::iota(selected.begin(), selected.end(), 0);
    }
    return selected;
}

void HipPosteriorEnsemble::run(const std::vector<double>& thetas, int S, bool want_sero) {
    sepaihrd_ctx* ctx = objective_->deviceContext();
    objective_->syncDeviceConstraintMode();
    ppc_.assign(static_cast<size_t>(6) * 5 * t_pos_ * n_, 0.0);
    sero_.assign(want_sero ? static_cast<size_t>(5) * time_points_.size() : 0, 0.0);
    int32_t nv = 0;
    rt_.assign(want_sero ? static_cast<size_t>(5) * time_points_.size() : 0, 0.0);
    metrics_.assign(want_sero ? static_cast<size_t>(S) * (12 + 4 * n_) : 0, 0.0);
    metrics_rows_ = want_sero ? S : 0;
    const int rc = sepaihrd_ensemble_quantiles(ctx, thetas.data(), S, kProbs, 5, ppc_.data(),
                                               want_sero ? sero_.data() : nullptr, want_sero ? rt_.data() : nullptr,
                                               want_sero ? metrics_.data() : nullptr, nullptr, &nv);
    if (rc != SEPAIHRD_OK)
        throw ModelException("HipPosteriorEnsemble", std::string("sepaihrd_ensemble_quantiles: ") + sepaihrd_last_error(ctx));
    n_valid_ = nv;
}

PosteriorPredictiveData HipPosteriorEnsemble::aggregatePosteriorPredictives(
    const std::vector<Eigen::VectorXd>& param_samples, int num_samples_for_ppc, unsigned int random_seed) {
    PosteriorPredictiveData out;
    for (double t : time_points_)
        if (t >= 0.0) out.time_points.push_back(t);
    if (out.time_points.empty()) return PosteriorPredictiveData();  // :203-206
    out.daily_hospitalizations.observed = data_.getNewHospitalizations();
    out.daily_icu_admissions.observed = data_.getNewICU();
    out.daily_deaths.observed = data_.getNewDeaths();
    if (param_samples.empty()) return out;  // :222-225
    const std::vector<int> selected = selectSamples(param_samples.size(), num_samples_for_ppc, random_seed);
    const size_t P = pm_.getParameterCount();
    std::vector<double> thetas(selected.size() * P);
    for (size_t s = 0; s < selected.size(); ++s) {
        const Eigen::VectorXd& v = param_samples[static_cast<size_t>(selected[s])];
        if (static_cast<size_t>(v.size()) != P) throw InvalidParameterException("HipPosteriorEnsemble", "sample size mismatch");
        for (size_t i = 0; i < P; ++i) thetas[s * P + i] = v[static_cast<Eigen::Index>(i)];
    }
    run(thetas, static_cast<int>(selected.size()), false);
    out.samples_used = n_valid_;
    PosteriorPredictiveData::IncidenceData* series[6] = {&out.daily_hospitalizations, &out.daily_icu_admissions,
                                                         &out.daily_deaths, &out.cumulative_hospitalizations,
                                                         &out.cumulative_icu_admissions, &out.cumulative_deaths};
    for (int ser = 0; ser < 6; ++ser) {
        Eigen::MatrixXd* dst[5] = {&series[ser]->lower_95, &series[ser]->lower_90, &series[ser]->median,
                                   &series[ser]->upper_90, &series[ser]->upper_95};  // order of kProbs
        for (int p = 0; p < 5; ++p) {
            dst[p]->resize(t_pos_, n_);
            for (int t = 0; t < t_pos_; ++t)
                for (int a = 0; a < n_; ++a)
                    (*dst[p])(t, a) = ppc_[((static_cast<size_t>(ser) * 5 + p) * t_pos_ + t) * n_ + a];
        }
    }
    return out;
}

std::map<double, AggregatedStats> HipPosteriorEnsemble::aggregateSeroprevalence(
    const std::vector<Eigen::VectorXd>& param_samples, int burn_in, int thinning) {
    std::map<double, AggregatedStats> out;
    if (param_samples.empty() || burn_in >= static_cast<int>(param_samples.size()) || thinning <= 0) return out;  // :180-189
    const size_t P = pm_.getParameterCount();
    std::vector<double> thetas;
    int S = 0;
    for (size_t i = static_cast<size_t>(burn_in); i < param_samples.size(); i += static_cast<size_t>(thinning)) {
        const Eigen::VectorXd& v = param_samples[i];
        if (static_cast<size_t>(v.size()) != P) throw InvalidParameterException("HipPosteriorEnsemble", "sample size mismatch");
        for (size_t k = 0; k < P; ++k) thetas.push_back(v[static_cast<Eigen::Index>(k)]);
        ++S;
    }
    run(thetas, S, true);
    if (n_valid_ == 0) return out;
    const size_t T = time_points_.size();
    const char* keys[5] = {"q025", "q05", "median", "q95", "q975"};  // order of kProbs
    for (size_t k = 0; k < T; ++k) {
        AggregatedStats st;
        for (int p = 0; p < 5; ++p) st[keys[p]] = sero_[static_cast<size_t>(p) * T + k];
        out[time_points_[k]] = st;
    }
    return out;
}

std::map<double, AggregatedStats> HipPosteriorEnsemble::aggregateRt(const std::vector<Eigen::VectorXd>& param_samples,
                                                                    int burn_in, int thinning) {
    // same ensemble, same launch as the seroprevalence: PostCalibrationAnalyser.cpp:233-236,342
    const std::map<double, AggregatedStats> sero = aggregateSeroprevalence(param_samples, burn_in, thinning);
    std::map<double, AggregatedStats> out;
    if (sero.empty()) return out;
    const size_t T = time_points_.size();
    const char* keys[5] = {"q025", "q05", "median", "q95", "q975"};
    for (size_t k = 0; k < T; ++k) {
        AggregatedStats st;
        for (int p = 0; p < 5; ++p) st[keys[p]] = rt_[static_cast<size_t>(p) * T + k];
        out[time_points_[k]] = st;
    }
    return out;
}

std::vector<EssentialMetrics> HipPosteriorEnsemble::calculateEssentialMetrics(
    const std::vector<Eigen::VectorXd>& param_samples, int burn_in, int thinning) {
    std::vector<EssentialMetrics> out;
    if (aggregateSeroprevalence(param_samples, burn_in, thinning).empty()) return out;  // same launch fills the table
    const size_t width = static_cast<size_t>(12 + 4 * n_);
    for (int s = 0; s < metrics_rows_; ++s) {
        const double* r = &metrics_[static_cast<size_t>(s) * width];
        if (std::isnan(r[0])) continue;  // invalid simulation: skipped (PostCalibrationAnalyser.cpp:222-226)
        EssentialMetrics m;
        m.R0 = r[0]; m.overall_IFR = r[1]; m.overall_attack_rate = r[2]; m.peak_hospital_occupancy = r[3];
        m.peak_ICU_occupancy = r[4]; m.time_to_peak_hospital = r[5]; m.time_to_peak_ICU = r[6];
        m.total_cumulative_deaths = r[7]; m.max_Rt = r[8]; m.min_Rt = r[9]; m.final_Rt = r[10];
        m.seroprevalence_at_target_day = r[11];
        for (int a = 0; a < n_; ++a) {
            m.age_specific_IFR.push_back(r[12 + 4 * a + 0]);
            m.age_specific_IHR.push_back(r[12 + 4 * a + 1]);
            m.age_specific_IICUR.push_back(r[12 + 4 * a + 2]);
            m.age_specific_attack_rate.push_back(r[12 + 4 * a + 3]);
        }
        out.push_back(std::move(m));
    }
    return out;
}

std::map<std::string, AggregatedStats> HipPosteriorEnsemble::aggregateMetrics(const std::vector<EssentialMetrics>& rows) {
    std::map<std::string, AggregatedStats> result;
    if (rows.empty()) return result;
    const int n = static_cast<int>(rows[0].age_specific_IFR.size());
    std::vector<std::pair<std::string, std::function<double(const EssentialMetrics&)>>> cols = {
        {"R0", [](const EssentialMetrics& m) { return m.R0; }},
        {"overall_IFR", [](const EssentialMetrics& m) { return m.overall_IFR; }},
        {"overall_attack_rate", [](const EssentialMetrics& m) { return m.overall_attack_rate; }},
        {"peak_hospital", [](const EssentialMetrics& m) { return m.peak_hospital_occupancy; }},
        {"peak_ICU", [](const EssentialMetrics& m) { return m.peak_ICU_occupancy; }},
        {"time_to_peak_hospital", [](const EssentialMetrics& m) { return m.time_to_peak_hospital; }},
        {"time_to_peak_ICU", [](const EssentialMetrics& m) { return m.time_to_peak_ICU; }},
        {"total_deaths", [](const EssentialMetrics& m) { return m.total_cumulative_deaths; }},
        {"max_Rt", [](const EssentialMetrics& m) { return m.max_Rt; }},
        {"min_Rt", [](const EssentialMetrics& m) { return m.min_Rt; }},
        {"final_Rt", [](const EssentialMetrics& m) { return m.final_Rt; }},
        {"seroprevalence_day64", [](const EssentialMetrics& m) { return m.seroprevalence_at_target_day; }}};
    for (int a = 0; a < n; ++a) {
        cols.push_back({"IFR_age_" + std::to_string(a), [a](const EssentialMetrics& m) { return m.age_specific_IFR[static_cast<size_t>(a)]; }});
        cols.push_back({"IHR_age_" + std::to_string(a), [a](const EssentialMetrics& m) { return m.age_specific_IHR[static_cast<size_t>(a)]; }});
        cols.push_back({"IICUR_age_" + std::to_string(a), [a](const EssentialMetrics& m) { return m.age_specific_IICUR[static_cast<size_t>(a)]; }});
        cols.push_back({"AttackRate_age_" + std::to_string(a), [a](const EssentialMetrics& m) { return m.age_specific_attack_rate[static_cast<size_t>(a)]; }});
    }
    auto quantile = [](const std::vector<double>& v, double q) {  // PostCalibrationAnalyser.cpp:316-326
        const double pos = q * (v.size() - 1);
        const size_t idx = static_cast<size_t>(pos);
        const double frac = pos - idx;
        return idx + 1 < v.size() ? v[idx] * (1.0 - frac) + v[idx + 1] * frac : v[idx];
    };
    for (const auto& col : cols) {
        std::vector<double> v;
        for (const EssentialMetrics& m : rows) v.push_back(col.second(m));
        double mean = 0.0;
        for (double x : v) mean += x;
        mean /= v.size();
        double var = 0.0;  // ba::variance(lazy): the population variance
        for (double x : v) var += (x - mean) * (x - mean);
        var /= v.size();
        std::sort(v.begin(), v.end());
        AggregatedStats st;
        st["mean"] = mean; st["std_dev"] = std::sqrt(var);
        st["median"] = quantile(v, 0.5); st["q025"] = quantile(v, 0.025); st["q975"] = quantile(v, 0.975);
        result[col.first] = st;
    }
    return result;
}

}  // namespace epidemic
