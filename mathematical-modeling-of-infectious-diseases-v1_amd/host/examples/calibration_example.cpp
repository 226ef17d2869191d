// host/examples/calibration_example.cpp
//
// What a caller of the reference writes, against the device path: the model and its NPI strategy (main.cpp:222-242),
// the objects SEPAIHRDModelCalibration::setupCalibrator builds from them
// (src/model/SEPAIHRDModelCalibration.cpp:84-118) WITH THE REFERENCE'S OWN ARGUMENT LISTS -- only the two class names
// differ -- and the run of runHillClimbingMCMC (:150-178), followed by the post-calibration ensemble
// (src/model/main.cpp:505-560).  Self-contained: a 4-age-group problem with synthetic observations.
//
//   make -C host examples && host/examples/calibration_example      (needs an MI355X)
#include <cmath>
#include <cstdio>
#include <memory>

#include "epidemic_hip/HipModelCalibrator.hpp"
#include "epidemic_hip/HipPosteriorEnsemble.hpp"
#include "epidemic_hip/HipSEPAIHRD.hpp"

using namespace epidemic;

int main() {
    const int n = 4, days = 90;
    SEPAIHRDParameters mp;
    mp.N = Eigen::VectorXd(n);
    const double N[4] = {3.0e6, 4.0e6, 2.0e6, 1.0e6};
    const double M[4][4] = {{7, 5, 2, 1}, {5, 8, 3, 1.5}, {2, 3, 4, 2}, {1, 1.5, 2, 3}};
    mp.M_baseline = Eigen::MatrixXd(n, n);
    for (int i = 0; i < n; ++i) {
        mp.N[i] = N[i];
        for (int j = 0; j < n; ++j) mp.M_baseline(i, j) = M[i][j];
    }
    auto vec4 = [&](double a, double b, double c, double d) {
        Eigen::VectorXd v(n);
        v[0] = a; v[1] = b; v[2] = c; v[3] = d;
        return v;
    };
    mp.a = vec4(1, 1, 1, 1); mp.h_infec = vec4(1, 1, 1, 1);
    mp.p = vec4(0.4, 0.3, 0.2, 0.1); mp.h = vec4(0.01, 0.03, 0.08, 0.15); mp.icu = vec4(0.05, 0.1, 0.25, 0.4);
    mp.d_H = vec4(0.01, 0.02, 0.05, 0.1); mp.d_ICU = vec4(0.2, 0.3, 0.4, 0.5); mp.d_community = vec4(0, 0, 0, 0);
    mp.beta = 0.05; mp.theta = 0.5; mp.sigma = 1.0 / 3; mp.gamma_p = 0.5; mp.gamma_A = 0.2; mp.gamma_I = 0.2;
    mp.gamma_H = 0.1; mp.gamma_ICU = 1.0 / 14;
    mp.runup_days = 0.0; mp.seed_exposed = 0.0;  // multiplier branch of the initial state
    // kappa(t): baseline 1.0 until day 13, then three calibratable periods (named kappa_2 .. kappa_4 by default)
    auto npi = std::make_shared<PiecewiseConstantNpiStrategy>(std::vector<double>{40, 70, 305}, std::vector<double>{0.5, 0.7, 0.9},
                                                               std::map<std::string, std::pair<double, double>>{}, 1.0, 13.0);
    auto model = std::make_shared<AgeSEPAIHRDModel>(mp, npi);

    const std::vector<std::string> names = {"beta", "theta", "kappa_2", "kappa_3", "kappa_4"};
    std::map<std::string, double> sigmas = {{"beta", 0.005}, {"theta", 0.02}, {"kappa_2", 0.05}, {"kappa_3", 0.05}, {"kappa_4", 0.05}};
    std::map<std::string, std::pair<double, double>> bounds = {
        {"beta", {0.01, 1.0}}, {"theta", {0.1, 1.0}}, {"kappa_2", {0.1, 1.5}}, {"kappa_3", {0.1, 1.5}}, {"kappa_4", {0.1, 1.5}}};
    // reference: std::make_unique<SEPAIHRDParameterManager>(model_, params_to_calibrate_, proposal_sigmas_, param_bounds_)
    HipSEPAIHRDParameterManager pm(model, names, sigmas, bounds);

    std::vector<double> times(days);
    for (int t = 0; t < days; ++t) times[static_cast<size_t>(t)] = t;
    Eigen::VectorXd x0(11 * n);
    for (int i = 0; i < 11 * n; ++i) x0[i] = 0.0;
    for (int a = 0; a < n; ++a) {
        x0[1 * n + a] = 40.0 + 10.0 * a;  // E
        x0[2 * n + a] = 25.0;             // P
        x0[3 * n + a] = 15.0;             // A
        x0[4 * n + a] = 30.0 + 15.0 * a;  // I
        x0[5 * n + a] = 10.0 + 3.0 * a;   // H
        x0[6 * n + a] = 3.0 + 3.0 * a;    // ICU
        double non_s = 0.0;
        for (int c = 1; c <= 8; ++c) non_s += x0[c * n + a];
        x0[a] = N[a] - non_s;
    }
    // synthetic observations: a smooth wave per stream and age group
    Eigen::MatrixXd oH(days, n), oI(days, n), oD(days, n);
    for (int t = 0; t < days; ++t)
        for (int a = 0; a < n; ++a) {
            const double wave = std::exp(-0.5 * std::pow((t - 35.0) / 14.0, 2.0));
            oH(t, a) = std::floor(4.0 + (30.0 + 25.0 * a) * wave);
            oI(t, a) = std::floor(1.0 + (6.0 + 8.0 * a) * wave);
            oD(t, a) = std::floor((1.0 + 5.0 * a) * wave);
        }
    CalibrationData data(oH, oI, oD, mp.N);
    SimulationCache cache;
    auto solver = std::make_shared<Dopri5SolverStrategy>();

    try {
        // reference: std::make_unique<SEPAIHRDObjectiveFunction>(model_, *parameterManager, *cache_, observed_data_,
        //                                                         time_points_, initial_state_cached_, solver_strategy_)
        HipSEPAIHRDObjectiveFunction objective(model, pm, cache, data, times, x0, solver);
        HipModelCalibrator calibrator(pm, objective);
        std::printf("initial log-likelihood   %.6f\n", calibrator.getInitialObjectiveValue());

        calibrator.calibrate({{"iterations", 25}, {"cloud_size_multiplier", 4}, {"threads", 8}, {"seed", 11}},
                             {{"mcmc_iterations", 600}, {"burn_in", 200}, {"adaptation_period", 100}, {"thinning", 5}, {"seed", 12},
                              // the reference's reporting keys: a progress line every 200 iterations, no trace files from an example
                              {"report_interval", 200}, {"write_checkpoints", 0}, {"write_trace", 0}},
                             /*chains=*/8);
        std::printf("phase 1 (hill climbing)  %.6f\n", calibrator.getPhase1Result().bestObjectiveValue);
        std::printf("best after MCMC          %.6f  (8 chains x 600 iterations)\n", calibrator.getBestObjectiveValue());
        const Eigen::VectorXd& best = calibrator.getBestParameterVector();
        for (size_t i = 0; i < names.size(); ++i) std::printf("  %-8s %.6f\n", names[i].c_str(), best[static_cast<Eigen::Index>(i)]);

        HipSEPAIHRDGradientObjectiveFunction gradient(model, pm, cache, data, times, x0, solver);
        Eigen::VectorXd g;
        const double f = gradient.evaluate_with_gradient(best, g);
        std::printf("gradient at the optimum  f = %.6f  d/dbeta = %.4g  d/dtheta = %.4g\n", f, g[0], g[1]);

        HipPosteriorEnsemble ensemble(pm, data, times, x0, solver);
        const PosteriorPredictiveData ppc = ensemble.aggregatePosteriorPredictives(calibrator.getMCMCSamples(), -1, 1);
        const auto sero = ensemble.aggregateSeroprevalence(calibrator.getMCMCSamples(), 40, 2);
        std::printf("posterior predictive     day 35, oldest group: hospital admissions median %.2f [%.2f, %.2f], observed %.0f (%d samples)\n",
                    ppc.daily_hospitalizations.median(35, 3), ppc.daily_hospitalizations.lower_95(35, 3),
                    ppc.daily_hospitalizations.upper_95(35, 3), oH(35, 3), ppc.samples_used);
        std::printf("seroprevalence day 89    median %.5f\n", sero.at(89.0).at("median"));
        const auto rt = ensemble.aggregateRt(calibrator.getMCMCSamples(), 40, 2);
        const std::vector<EssentialMetrics> rows = ensemble.calculateEssentialMetrics(calibrator.getMCMCSamples(), 40, 2);
        const auto summary = HipPosteriorEnsemble::aggregateMetrics(rows);
        std::printf("Rt day 0 / day 89        median %.3f / %.3f\n", rt.at(0.0).at("median"), rt.at(89.0).at("median"));
        std::printf("essential metrics        %zu rows; R0 median %.3f [%.3f, %.3f], peak hospital occupancy mean %.1f, IFR oldest group mean %.4f\n",
                    rows.size(), summary.at("R0").at("median"), summary.at("R0").at("q025"), summary.at("R0").at("q975"),
                    summary.at("peak_hospital").at("mean"), summary.at("IFR_age_3").at("mean"));

        const bool ok = std::isfinite(calibrator.getBestObjectiveValue()) &&
                        calibrator.getBestObjectiveValue() >= calibrator.getInitialObjectiveValue() &&
                        ppc.samples_used == static_cast<int>(calibrator.getMCMCSamples().size()) &&
                        rows.size() == (calibrator.getMCMCSamples().size() - 40 + 1) / 2 && summary.at("R0").at("median") > 0.0;
        std::printf(ok ? "OK\n" : "FAILED\n");
        return ok ? 0 : 1;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 2;
    }
}
