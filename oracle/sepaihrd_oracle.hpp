// =============================================================================
// oracle/sepaihrd_oracle.hpp  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// CPU restatement (plain C++17, no dependencies) of the reference's MCMC
// likelihood hot path.  Only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may link or call this.  The product path (HIP kernels behind
// include/sepaihrd_hip.h) never routes through it.
//
// PARITY STATUS: "parity unpinned" for the integrator.  The adaptive RK
// arithmetic lives in Boost.Odeint (find_package(Boost) in the reference's
// CMakeLists.txt:34, version unpinned, sources absent from /root/reference and
// from this image) and the reference holds no test that pins a trajectory or a
// full log-likelihood value.  The integrator below restates Boost.Odeint's
// published algorithm (integrate_times + controlled_runge_kutta +
// runge_kutta_dopri5 / runge_kutta_cash_karp54) and is anchored on the
// reference's call sites.  What IS pinned by the reference's own tests:
//   * the Poisson log-likelihood closed form
//     (tests/model/SEPAIHRDObjectivefunctionTest.cpp:688-752, tol 1e-8);
//   * structural properties of calculate() (same file :334-685).
// Independent cross-checks (SciPy DOP853 / Radau at rtol 1e-12, mpmath RHS
// values) are committed under tests/golden/ with their generating script, and
// tests/test_oracle_independent_steps.py pins one step of each stepper, the error
// norm / accept rule and both step-size rules against SciPy's RK45 tableau and
// the published Cash-Karp tableau (no code shared with this file).
//
// Every function cites the reference file:line it follows (paths relative to
// /root/reference).
// =============================================================================
#pragma once
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <deque>
#include <functional>
#include <limits>
#include <map>
#include <random>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace oracle {

using state_type = std::vector<double>;

// include/model/ModelConstants.hpp:9,18,22
constexpr double MIN_POPULATION_FOR_DIVISION = 1e-9;
constexpr int NUM_COMPARTMENTS = 11;
constexpr int NUM_POPULATION_COMPARTMENTS = 9;
constexpr int MAX_AGE_CLASSES = 64;  // oracle-side bound (the reference is generic in n)

// -----------------------------------------------------------------------------
// Model parameters: include/model/parameters/SEPAIHRDParameters.hpp:20-124
// M is stored column-major like Eigen::MatrixXd (M[j*n+i] == M(i,j)).
// kappa_end_times / kappa_values include the baseline period at index 0, as
// AgeSEPAIHRDModel::getModelParameters returns them
// (src/model/AgeSEPAIHRDModel.cpp:316-321).
// -----------------------------------------------------------------------------
struct Params {
    int n = 0;
    std::vector<double> N, M;
    double beta = 0, theta = 0, sigma = 0, gamma_p = 0, gamma_A = 0, gamma_I = 0,
           gamma_H = 0, gamma_ICU = 0;
    std::vector<double> a, h_infec, p, h, icu, d_H, d_ICU, d_community;
    std::vector<double> beta_end_times, beta_values;
    std::vector<double> kappa_end_times, kappa_values;
    double E0_multiplier = 1, P0_multiplier = 1, A0_multiplier = 1, I0_multiplier = 1,
           H0_multiplier = 1, ICU0_multiplier = 1, R0_multiplier = 1, D0_multiplier = 1;
    double runup_days = 30.0, seed_exposed = 10.0;
};

// -----------------------------------------------------------------------------
// Piecewise-constant schedules.
// kappa(t): src/model/PieceWiseConstantNPIStrategy.cpp:86-127
// beta(t) : src/model/PiecewiseConstantParameterStrategy.cpp:37-74, built in
//           src/model/AgeSEPAIHRDModel.cpp:352-362 (first entry = baseline).
// The reference keeps a mutable "last period" cache; both of its branches agree
// with the pure lower_bound lookup written here.
// -----------------------------------------------------------------------------
inline double piecewise_value(const std::vector<double>& ends_after, const double* vals_after,
                              double baseline_value, double baseline_end, double t) {
    if (t <= baseline_end) return baseline_value;
    if (ends_after.empty()) return baseline_value;
    auto it = std::lower_bound(ends_after.begin(), ends_after.end(), t);
    if (it == ends_after.end()) return vals_after[ends_after.size() - 1];
    return vals_after[it - ends_after.begin()];
}

inline double kappa_at(const Params& P, double t) {
    // PieceWiseConstantNPIStrategy.cpp:87-89: t < 0 -> baseline
    if (t < 0) return P.kappa_values.front();
    std::vector<double> ends(P.kappa_end_times.begin() + 1, P.kappa_end_times.end());
    return piecewise_value(ends, P.kappa_values.data() + 1, P.kappa_values.front(),
                           P.kappa_end_times.front(), t);
}

inline double beta_at(const Params& P, double t) {
    // AgeSEPAIHRDModel.cpp:352-368: schedule only when sizes match and non-empty
    if (P.beta_values.empty() || P.beta_end_times.empty() ||
        P.beta_values.size() != P.beta_end_times.size())
        return P.beta;
    std::vector<double> ends(P.beta_end_times.begin() + 1, P.beta_end_times.end());
    return piecewise_value(ends, P.beta_values.data() + 1, P.beta_values.front(),
                           P.beta_end_times.front(), t);
}

// -----------------------------------------------------------------------------
// Model with the pre-computed members the reference keeps.
// -----------------------------------------------------------------------------
struct Model {
    Params P;
    std::vector<double> inv_N;
    // lookups hoisted out of the RHS (same values as kappa_at/beta_at)
    std::vector<double> kappa_ends_after, beta_ends_after;
    bool has_beta_schedule = false;

    explicit Model(const Params& p) { set(p); }
    // src/model/AgeSEPAIHRDModel.cpp:325-363 setModelParameters
    void set(const Params& p) {
        P = p;
        if (P.n <= 0 || P.n > MAX_AGE_CLASSES) throw std::invalid_argument("Model: bad n");
        if (P.kappa_values.empty() || P.kappa_values.size() != P.kappa_end_times.size())
            throw std::invalid_argument("Model: kappa schedule needs >= 1 (baseline) entry");
        if (P.d_community.empty()) P.d_community.assign(P.n, 0.0);
        inv_N.resize(P.n);
        for (int i = 0; i < P.n; ++i)
            inv_N[i] = (P.N[i] > MIN_POPULATION_FOR_DIVISION) ? (1.0 / P.N[i]) : 0.0;
        kappa_ends_after.assign(P.kappa_end_times.begin() + (P.kappa_end_times.empty() ? 0 : 1),
                                P.kappa_end_times.end());
        has_beta_schedule = !P.beta_values.empty() && !P.beta_end_times.empty() &&
                            P.beta_values.size() == P.beta_end_times.size();
        if (has_beta_schedule)
            beta_ends_after.assign(P.beta_end_times.begin() + 1, P.beta_end_times.end());
        else
            beta_ends_after.clear();
    }
    double kappa(double t) const {
        if (t < 0) return P.kappa_values.front();
        return piecewise_value(kappa_ends_after, P.kappa_values.data() + 1,
                               P.kappa_values.front(), P.kappa_end_times.front(), t);
    }
    double beta(double t) const {
        if (!has_beta_schedule) return P.beta;
        return piecewise_value(beta_ends_after, P.beta_values.data() + 1, P.beta_values.front(),
                               P.beta_end_times.front(), t);
    }

    // src/model/AgeSEPAIHRDModel.cpp:101-228 computeDerivatives.
    // Operation order is kept statement by statement (no FMA: the reference
    // builds with -O3 on baseline x86-64, CMakeLists.txt:25-29).
    void rhs(const double* x, double* dx, double t) const {
        const int n = P.n;
        const double* S = x;
        const double* E = x + n;
        const double* Pp = x + 2 * n;
        const double* A = x + 3 * n;
        const double* I = x + 4 * n;
        const double* H = x + 5 * n;
        const double* ICU = x + 6 * n;
        double inf_pressure[MAX_AGE_CLASSES], lambda[MAX_AGE_CLASSES];  // reference: workspace_ members
        for (int i = 0; i < n; ++i) lambda[i] = 0.0;  // :161-164
        for (int i = 0; i < n; ++i) {  // :153-157
            double total_inf = Pp[i] + A[i] + P.theta * I[i];
            inf_pressure[i] = total_inf * P.h_infec[i] * inv_N[i];
        }
        for (int j = 0; j < n; ++j) {  // :166-174, column-major walk
            const double inf_j = inf_pressure[j];
            for (int i = 0; i < n; ++i) lambda[i] += P.M[j * n + i] * inf_j;
        }
        const double beta_eff = beta(t) * kappa(t);  // :176-178
        for (int i = 0; i < n; ++i) lambda[i] *= beta_eff * P.a[i];  // :180-183
        for (int i = 0; i < n; ++i) {  // :195-227
            double lambda_val = std::max(0.0, lambda[i]);
            double flow_SE = lambda_val * S[i];
            double flow_EP = P.sigma * E[i];
            double flow_P_out = P.gamma_p * Pp[i];
            double flow_PA = P.p[i] * flow_P_out;
            double flow_PI = flow_P_out - flow_PA;
            double flow_IH = P.h[i] * I[i];
            double flow_IR = P.gamma_I * I[i];
            double flow_ID_community = P.d_community[i] * I[i];
            double I_out = flow_IR + flow_IH + flow_ID_community;
            double flow_H_ICU = P.icu[i] * H[i];
            double H_out = P.gamma_H * H[i] + P.d_H[i] * H[i] + flow_H_ICU;
            double ICU_out = (P.gamma_ICU + P.d_ICU[i]) * ICU[i];
            dx[0 * n + i] = -flow_SE;
            dx[1 * n + i] = flow_SE - flow_EP;
            dx[2 * n + i] = flow_EP - flow_P_out;
            dx[3 * n + i] = flow_PA - P.gamma_A * A[i];
            dx[4 * n + i] = flow_PI - I_out;
            dx[5 * n + i] = flow_IH - H_out;
            dx[6 * n + i] = flow_H_ICU - ICU_out;
            dx[7 * n + i] = P.gamma_A * A[i] + flow_IR + P.gamma_H * H[i] + P.gamma_ICU * ICU[i];
            dx[8 * n + i] = P.d_H[i] * H[i] + P.d_ICU[i] * ICU[i] + flow_ID_community;
            dx[9 * n + i] = flow_IH;
            dx[10 * n + i] = flow_H_ICU;
        }
    }
};

// -----------------------------------------------------------------------------
// Boost.Odeint restatement (third-party, version unpinned, NOT in the image).
// Call sites: src/sir_age_structured/solvers/Dopri5SolverStrategy.cpp:28-37,
//             src/sir_age_structured/solvers/CashKarpSolverStrategy.cpp:18-25.
// Published algorithm restated:
//   boost/numeric/odeint/integrate/detail/integrate_times.hpp (controlled tag)
//   boost/numeric/odeint/stepper/controlled_runge_kutta.hpp
//       (default_error_checker, default_step_adjuster, try_step for the
//        explicit_error_stepper_tag and explicit_error_stepper_fsal_tag)
//   boost/numeric/odeint/stepper/runge_kutta_dopri5.hpp (do_step_impl)
//   boost/numeric/odeint/stepper/runge_kutta_cash_karp54.hpp + generic RK
// -----------------------------------------------------------------------------
enum Solver { DOPRI5 = 0, CASH_KARP54 = 1 };

struct step_adjustment_error : std::runtime_error {
    step_adjustment_error()
        : std::runtime_error("Max number of iterations exceeded (500). A new step size was not found.") {}
};

// Build-side guard shared with the HIP path (the reference / odeint have none): with a degenerate
// tolerance the step size underflows to 0 and zero-length steps are "accepted" forever.
struct step_budget_error : std::runtime_error {
    step_budget_error() : std::runtime_error("step attempt budget exhausted") {}
};

struct StepStats {
    long accepted = 0, rejected = 0, rhs_calls = 0;
    long max_attempts = 1000000;
};

using System = std::function<void(const double*, double*, double)>;

// default_error_checker::error: max_i |xerr_i| / (eps_abs + eps_rel*(a_x*|x_i| + a_dxdt*dt*|dxdt_i|)),
// a_x = a_dxdt = 1; norm_inf via max(init, |v|) starting from 0.
inline double error_norm(const state_type& x_old, const state_type& dxdt_old, state_type& xerr,
                         double dt, double eps_abs, double eps_rel) {
    const double a_x = 1.0, a_dxdt_dt = 1.0 * dt;
    double m = 0.0;
    for (size_t i = 0; i < xerr.size(); ++i) {
        xerr[i] = std::abs(xerr[i]) /
                  (eps_abs + eps_rel * (a_x * std::abs(x_old[i]) + a_dxdt_dt * std::abs(dxdt_old[i])));
        m = std::max(m, std::abs(xerr[i]));
    }
    return m;
}
// default_step_adjuster (max_dt = 0): order = 5, error_order = 4 for both steppers.
inline double decrease_step(double dt, double error, int error_order) {
    dt *= std::max(9.0 / 10.0 * std::pow(error, -1.0 / (error_order - 1)), 1.0 / 5.0);
    return dt;
}
inline double increase_step(double dt, double error, int stepper_order) {
    if (error < 0.5) {
        error = std::max(std::pow(5.0, -static_cast<double>(stepper_order)), error);
        dt *= 9.0 / 10.0 * std::pow(error, -1.0 / stepper_order);
    }
    return dt;
}

struct Dopri5Tableau {
    // runge_kutta_dopri5::do_step_impl: every coefficient is a quotient of two
    // doubles; the error weights dc_i are DIFFERENCES of two rounded quotients.
    static constexpr double a2 = 1.0 / 5, a3 = 3.0 / 10, a4 = 4.0 / 5, a5 = 8.0 / 9;
    static constexpr double b21 = 1.0 / 5;
    static constexpr double b31 = 3.0 / 40, b32 = 9.0 / 40;
    static constexpr double b41 = 44.0 / 45, b42 = -56.0 / 15, b43 = 32.0 / 9;
    static constexpr double b51 = 19372.0 / 6561, b52 = -25360.0 / 2187, b53 = 64448.0 / 6561,
                            b54 = -212.0 / 729;
    static constexpr double b61 = 9017.0 / 3168, b62 = -355.0 / 33, b63 = 46732.0 / 5247,
                            b64 = 49.0 / 176, b65 = -5103.0 / 18656;
    static constexpr double c1 = 35.0 / 384, c3 = 500.0 / 1113, c4 = 125.0 / 192,
                            c5 = -2187.0 / 6784, c6 = 11.0 / 84;
    static constexpr double dc1 = c1 - 5179.0 / 57600, dc3 = c3 - 7571.0 / 16695,
                            dc4 = c4 - 393.0 / 640, dc5 = c5 - (-92097.0 / 339200),
                            dc6 = c6 - 187.0 / 2100, dc7 = -1.0 / 40;
};

struct CashKarpTableau {
    // rk54_ck_coefficients_{a1..a5,b,db,c}
    static constexpr double c2 = 1.0 / 5, c3 = 3.0 / 10, c4 = 3.0 / 5, c5 = 1.0, c6 = 7.0 / 8;
    static constexpr double a21 = 1.0 / 5;
    static constexpr double a31 = 3.0 / 40, a32 = 9.0 / 40;
    static constexpr double a41 = 3.0 / 10, a42 = -9.0 / 10, a43 = 6.0 / 5;
    static constexpr double a51 = -11.0 / 54, a52 = 5.0 / 2, a53 = -70.0 / 27, a54 = 35.0 / 27;
    static constexpr double a61 = 1631.0 / 55296, a62 = 175.0 / 512, a63 = 575.0 / 13824,
                            a64 = 44275.0 / 110592, a65 = 253.0 / 4096;
    static constexpr double b1 = 37.0 / 378, b3 = 250.0 / 621, b4 = 125.0 / 594, b6 = 512.0 / 1771;
    static constexpr double db1 = 37.0 / 378 - 2825.0 / 27648, db3 = 250.0 / 621 - 18575.0 / 48384,
                            db4 = 125.0 / 594 - 13525.0 / 55296, db5 = -277.0 / 14336,
                            db6 = 512.0 / 1771 - 1.0 / 4;
};

// controlled_runge_kutta< runge_kutta_dopri5 > (FSAL) ------------------------
struct ControlledDopri5 {
    double eps_abs, eps_rel;
    bool first_call = true;
    state_type dxdt, xnew, dxdtnew, xerr, xtmp, k2, k3, k4, k5, k6;
    StepStats* st;
    ControlledDopri5(double a, double r, StepStats* s) : eps_abs(a), eps_rel(r), st(s) {}

    // returns true on success; t and dt updated like try_step(sys, x, t, dt)
    bool try_step(const System& sys, state_type& x, double& t, double& dt) {
        const size_t m = x.size();
        if (dxdt.size() != m || first_call) {  // try_step_v1: resize || m_first_call -> initialize
            dxdt.resize(m);
            sys(x.data(), dxdt.data(), t);
            if (st) st->rhs_calls++;
            first_call = false;
        }
        xnew.resize(m); dxdtnew.resize(m); xerr.resize(m); xtmp.resize(m);
        k2.resize(m); k3.resize(m); k4.resize(m); k5.resize(m); k6.resize(m);
        using T = Dopri5Tableau;
        const double* k1 = dxdt.data();
        {   // scale_sumN: t1 = a1*t2 + a2*t3 + ... evaluated left to right, a1 = 1.0
            const double f1 = dt * T::b21;
            for (size_t i = 0; i < m; ++i) xtmp[i] = 1.0 * x[i] + f1 * k1[i];
            sys(xtmp.data(), k2.data(), t + dt * T::a2);
        }
        {
            const double f1 = dt * T::b31, f2 = dt * T::b32;
            for (size_t i = 0; i < m; ++i) xtmp[i] = 1.0 * x[i] + f1 * k1[i] + f2 * k2[i];
            sys(xtmp.data(), k3.data(), t + dt * T::a3);
        }
        {
            const double f1 = dt * T::b41, f2 = dt * T::b42, f3 = dt * T::b43;
            for (size_t i = 0; i < m; ++i) xtmp[i] = 1.0 * x[i] + f1 * k1[i] + f2 * k2[i] + f3 * k3[i];
            sys(xtmp.data(), k4.data(), t + dt * T::a4);
        }
        {
            const double f1 = dt * T::b51, f2 = dt * T::b52, f3 = dt * T::b53, f4 = dt * T::b54;
            for (size_t i = 0; i < m; ++i)
                xtmp[i] = 1.0 * x[i] + f1 * k1[i] + f2 * k2[i] + f3 * k3[i] + f4 * k4[i];
            sys(xtmp.data(), k5.data(), t + dt * T::a5);
        }
        {
            const double f1 = dt * T::b61, f2 = dt * T::b62, f3 = dt * T::b63, f4 = dt * T::b64,
                         f5 = dt * T::b65;
            for (size_t i = 0; i < m; ++i)
                xtmp[i] = 1.0 * x[i] + f1 * k1[i] + f2 * k2[i] + f3 * k3[i] + f4 * k4[i] + f5 * k5[i];
            sys(xtmp.data(), k6.data(), t + dt);
        }
        {
            const double f1 = dt * T::c1, f3 = dt * T::c3, f4 = dt * T::c4, f5 = dt * T::c5,
                         f6 = dt * T::c6;
            for (size_t i = 0; i < m; ++i)
                xnew[i] = 1.0 * x[i] + f1 * k1[i] + f3 * k3[i] + f4 * k4[i] + f5 * k5[i] + f6 * k6[i];
            sys(xnew.data(), dxdtnew.data(), t + dt);  // the new derivative (FSAL)
        }
        if (st) st->rhs_calls += 6;
        {
            const double e1 = dt * T::dc1, e3 = dt * T::dc3, e4 = dt * T::dc4, e5 = dt * T::dc5,
                         e6 = dt * T::dc6, e7 = dt * T::dc7;
            for (size_t i = 0; i < m; ++i)
                xerr[i] = e1 * k1[i] + e3 * k3[i] + e4 * k4[i] + e5 * k5[i] + e6 * k6[i] + e7 * dxdtnew[i];
        }
        const double max_rel_err = error_norm(x, dxdt, xerr, dt, eps_abs, eps_rel);
        if (max_rel_err > 1.0) {
            dt = decrease_step(dt, max_rel_err, 4);
            if (st) st->rejected++;
            return false;
        }
        t += dt;
        dt = increase_step(dt, max_rel_err, 5);
        x = xnew;        // copied only on success
        dxdt = dxdtnew;
        if (st) st->accepted++;
        return true;
    }
};

// controlled_runge_kutta< runge_kutta_cash_karp54 > (non-FSAL, generic RK) ----
struct ControlledCashKarp {
    double eps_abs, eps_rel;
    state_type dxdt, xnew, xerr, xtmp, k2, k3, k4, k5, k6;
    StepStats* st;
    ControlledCashKarp(double a, double r, StepStats* s) : eps_abs(a), eps_rel(r), st(s) {}

    bool try_step(const System& sys, state_type& x, double& t, double& dt) {
        const size_t m = x.size();
        dxdt.resize(m); xnew.resize(m); xerr.resize(m); xtmp.resize(m);
        k2.resize(m); k3.resize(m); k4.resize(m); k5.resize(m); k6.resize(m);
        sys(x.data(), dxdt.data(), t);  // try_step_v1: sys(x, m_dxdt, t) at EVERY attempt
        using T = CashKarpTableau;
        const double* k1 = dxdt.data();
        // generic_rk_scale_sum: coefficients a[i]*dt; zero coefficients contribute +0.0
        {
            const double f1 = T::a21 * dt;
            for (size_t i = 0; i < m; ++i) xtmp[i] = 1.0 * x[i] + f1 * k1[i];
            sys(xtmp.data(), k2.data(), t + T::c2 * dt);
        }
        {
            const double f1 = T::a31 * dt, f2 = T::a32 * dt;
            for (size_t i = 0; i < m; ++i) xtmp[i] = 1.0 * x[i] + f1 * k1[i] + f2 * k2[i];
            sys(xtmp.data(), k3.data(), t + T::c3 * dt);
        }
        {
            const double f1 = T::a41 * dt, f2 = T::a42 * dt, f3 = T::a43 * dt;
            for (size_t i = 0; i < m; ++i) xtmp[i] = 1.0 * x[i] + f1 * k1[i] + f2 * k2[i] + f3 * k3[i];
            sys(xtmp.data(), k4.data(), t + T::c4 * dt);
        }
        {
            const double f1 = T::a51 * dt, f2 = T::a52 * dt, f3 = T::a53 * dt, f4 = T::a54 * dt;
            for (size_t i = 0; i < m; ++i)
                xtmp[i] = 1.0 * x[i] + f1 * k1[i] + f2 * k2[i] + f3 * k3[i] + f4 * k4[i];
            sys(xtmp.data(), k5.data(), t + T::c5 * dt);
        }
        {
            const double f1 = T::a61 * dt, f2 = T::a62 * dt, f3 = T::a63 * dt, f4 = T::a64 * dt,
                         f5 = T::a65 * dt;
            for (size_t i = 0; i < m; ++i)
                xtmp[i] = 1.0 * x[i] + f1 * k1[i] + f2 * k2[i] + f3 * k3[i] + f4 * k4[i] + f5 * k5[i];
            sys(xtmp.data(), k6.data(), t + T::c6 * dt);
        }
        if (st) st->rhs_calls += 6;
        {
            const double f1 = T::b1 * dt, f2 = 0.0 * dt, f3 = T::b3 * dt, f4 = T::b4 * dt,
                         f5 = 0.0 * dt, f6 = T::b6 * dt;
            for (size_t i = 0; i < m; ++i)
                xnew[i] = 1.0 * x[i] + f1 * k1[i] + f2 * k2[i] + f3 * k3[i] + f4 * k4[i] + f5 * k5[i] +
                          f6 * k6[i];
        }
        {
            const double e1 = T::db1 * dt, e2 = 0.0 * dt, e3 = T::db3 * dt, e4 = T::db4 * dt,
                         e5 = T::db5 * dt, e6 = T::db6 * dt;
            for (size_t i = 0; i < m; ++i)
                xerr[i] = e1 * k1[i] + e2 * k2[i] + e3 * k3[i] + e4 * k4[i] + e5 * k5[i] + e6 * k6[i];
        }
        const double max_rel_err = error_norm(x, dxdt, xerr, dt, eps_abs, eps_rel);
        if (max_rel_err > 1.0) {
            dt = decrease_step(dt, max_rel_err, 4);
            if (st) st->rejected++;
            return false;
        }
        t += dt;
        dt = increase_step(dt, max_rel_err, 5);
        x = xnew;
        if (st) st->accepted++;
        return true;
    }
};

// integrate_times( controlled stepper, sys, x, times_begin, times_end, dt, obs )
template <class Stepper>
inline size_t integrate_times(Stepper& stepper, const System& sys, state_type& x,
                              const std::vector<double>& times, double dt,
                              const std::function<void(const state_type&, double)>& obs,
                              long max_attempts = 1000000) {
    size_t steps = 0;
    long attempts = 0;
    int fails = 0;  // failed_step_checker, max 500
    auto it = times.begin();
    const auto end = times.end();
    if (it == end) return 0;
    while (true) {
        double current_time = *it++;
        obs(x, current_time);
        if (it == end) break;
        // less_with_sign(t1, t2, dt>0): t2 - t1 > epsilon
        while ((*it - current_time) > std::numeric_limits<double>::epsilon()) {
            double current_dt = std::min(dt, *it - current_time);  // min_abs, dt > 0
            if (attempts++ >= max_attempts) throw step_budget_error();
            if (stepper.try_step(sys, x, current_time, current_dt)) {
                ++steps;
                fails = 0;
                dt = std::max(dt, current_dt);  // max_abs
            } else {
                if (fails++ >= 500) throw step_adjustment_error();
                dt = current_dt;
            }
        }
    }
    return steps;
}

// -----------------------------------------------------------------------------
// Simulator::run grid validation (src/sir_age_structured/Simulator.cpp:60-150)
// -----------------------------------------------------------------------------
struct SimulationResult {
    std::vector<double> time_points;
    std::vector<double> flat;  // T x m, row = output time (one allocation instead of one per day)
    size_t m = 0;
    const double* row(size_t k) const { return flat.data() + k * m; }
};

// `res` is caller-provided scratch so that a worker thread reuses one buffer across evaluations
// (a fresh >128 KB vector per call is an mmap + page faults under the process-wide mm lock, which
// serialises OpenMP workers).
inline void simulate(SimulationResult& res, const Model& model, const state_type& init,
                     const std::vector<double>& times, Solver solver, double dt_hint,
                     double abs_err, double rel_err, StepStats* st = nullptr) {
    if (static_cast<int>(init.size()) != NUM_COMPARTMENTS * model.P.n)
        throw std::invalid_argument("Simulator::run: initial state size mismatch");
    if (times.empty()) throw std::invalid_argument("Simulator::run: empty output time points");
    for (size_t i = 1; i < times.size(); ++i)
        if (times[i] <= times[i - 1])
            throw std::invalid_argument("Simulator::run: time points must be strictly increasing");
    res.m = init.size();
    res.time_points.clear();
    res.flat.clear();
    res.time_points.reserve(times.size());
    res.flat.reserve(times.size() * init.size());
    state_type x = init;
    System sys = [&model](const double* xs, double* dx, double t) { model.rhs(xs, dx, t); };
    auto obs = [&res](const state_type& s, double t) {
        res.time_points.push_back(t);
        res.flat.insert(res.flat.end(), s.begin(), s.end());
    };
    if (solver == DOPRI5) {
        ControlledDopri5 stp(abs_err, rel_err, st);
        integrate_times(stp, sys, x, times, dt_hint, obs, st ? st->max_attempts : 1000000);
    } else {
        ControlledCashKarp stp(abs_err, rel_err, st);
        integrate_times(stp, sys, x, times, dt_hint, obs, st ? st->max_attempts : 1000000);
    }
}

// -----------------------------------------------------------------------------
// Poisson log-likelihood of one stream, serial summation order.
// src/model/objectives/SEPAIHRDObjectiveFunction.cpp:241-279.  The reference
// wraps the row loop in an OpenMP reduction when rows*cols >= 256; its result
// then depends on the thread count, so parity is defined against this serial
// (1-thread) order.  sim, obs: row-major rows x cols.
// -----------------------------------------------------------------------------
inline double poisson_loglik(const double* sim, const double* obs, int rows, int cols) {
    const double epsilon = 1e-10;
    double log_likelihood = 0.0;
    for (int i = 0; i < rows; ++i) {
        double row_sum = 0.0;
        for (int j = 0; j < cols; ++j) {
            const double o = obs[i * cols + j];
            if (o >= 0.0 && std::isfinite(o)) {
                double s = sim[i * cols + j];
                if (s < 0.0) s = 0.0;
                s += epsilon;
                row_sum += (o * std::log(s) - s);
            }
        }
        log_likelihood += row_sum;
    }
    return log_likelihood;
}

// -----------------------------------------------------------------------------
// Parameter manager: src/model/parameters/SEPAIHRDParameterManager.cpp
// -----------------------------------------------------------------------------
enum ConstraintMode { OPTIMIZATION_CLAMP = 0, MCMC_REFLECT = 1 };

// :302-313
inline double reflect_bound(double value, double minb, double maxb) {
    if (minb >= maxb) return minb;
    double width = maxb - minb;
    double y = std::fmod(value - minb, 2.0 * width);
    if (y < 0) y += 2.0 * width;
    if (y <= width) return minb + y;
    return maxb - (y - width);
}

struct ParameterManager {
    std::vector<std::string> names;
    std::map<std::string, double> sigmas;
    std::map<std::string, std::pair<double, double>> bounds;
    // names of the NPI strategy's calibratable (after-baseline) values, in order
    // (PieceWiseConstantNPIStrategy.cpp:55-61: default "kappa_<i+2>")
    std::vector<std::string> npi_names;
    ConstraintMode mode = OPTIMIZATION_CLAMP;

    // :315-347
    std::vector<double> applyConstraints(const std::vector<double>& p) const {
        if (p.size() != names.size()) throw std::invalid_argument("applyConstraints: size mismatch");
        std::vector<double> c = p;
        for (size_t i = 0; i < names.size(); ++i) {
            auto it = bounds.find(names[i]);
            if (it != bounds.end()) {
                double minb = it->second.first, maxb = it->second.second;
                if (minb > maxb) std::swap(minb, maxb);
                if (mode == OPTIMIZATION_CLAMP) c[i] = std::min(std::max(p[i], minb), maxb);
                else c[i] = reflect_bound(p[i], minb, maxb);
            } else {
                if (mode == OPTIMIZATION_CLAMP) c[i] = std::max(0.0, p[i]);
                else c[i] = std::abs(p[i]);
            }
        }
        return c;
    }

    static bool starts(const std::string& s, const char* pre) { return s.rfind(pre, 0) == 0; }

    // :164-287 updateModelParameters(theta, target_model).  Throws on the same
    // conditions the reference throws on (the objective maps any throw to lowest()).
    void updateModelParameters(const std::vector<double>& theta, Model& model) const {
        if (theta.size() != names.size()) throw std::invalid_argument("size mismatch");
        std::vector<double> c = applyConstraints(theta);
        Params up = model.P;
        std::vector<double> npi_vals(up.kappa_values.begin() + 1, up.kappa_values.end());
        bool npi_update = false;
        for (size_t i = 0; i < names.size(); ++i) {
            const std::string& name = names[i];
            const double value = c[i];
            if (name == "beta") up.beta = value;
            else if (starts(name, "beta_")) {
                size_t idx = std::stoul(name.substr(5)) - 1;
                if (idx < up.beta_values.size()) up.beta_values[idx] = value;
                else throw std::invalid_argument("Beta index out of range: " + name);
            }
            else if (name == "theta") up.theta = value;
            else if (name == "sigma") up.sigma = value;
            else if (name == "gamma_p") up.gamma_p = value;
            else if (name == "gamma_A") up.gamma_A = value;
            else if (name == "gamma_I") up.gamma_I = value;
            else if (name == "gamma_H") up.gamma_H = value;
            else if (name == "gamma_ICU") up.gamma_ICU = value;
            else if (starts(name, "a_")) up.a.at(std::stoul(name.substr(2))) = value;
            else if (starts(name, "h_infec_")) up.h_infec.at(std::stoul(name.substr(8))) = value;
            else if (starts(name, "p_")) up.p.at(std::stoul(name.substr(2))) = value;
            else if (starts(name, "h_")) up.h.at(std::stoul(name.substr(2))) = value;
            else if (starts(name, "icu_")) up.icu.at(std::stoul(name.substr(4))) = value;
            else if (starts(name, "d_H_")) up.d_H.at(std::stoul(name.substr(4))) = value;
            else if (starts(name, "d_ICU_")) up.d_ICU.at(std::stoul(name.substr(6))) = value;
            else if (starts(name, "d_community_")) {
                size_t idx = std::stoul(name.substr(12));
                if (up.d_community.empty()) up.d_community.assign(model.P.n, 0.0);
                if (idx < up.d_community.size()) up.d_community[idx] = value;
            }
            else if (name == "seed_exposed") up.seed_exposed = value;
            else if (name == "runup_days") up.runup_days = value;
            else if (name == "E0_multiplier") up.E0_multiplier = value;
            else if (name == "P0_multiplier") up.P0_multiplier = value;
            else if (name == "A0_multiplier") up.A0_multiplier = value;
            else if (name == "I0_multiplier") up.I0_multiplier = value;
            else if (name == "H0_multiplier") up.H0_multiplier = value;
            else if (name == "ICU0_multiplier") up.ICU0_multiplier = value;
            else if (name == "R0_multiplier") up.R0_multiplier = value;
            else if (name == "D0_multiplier") up.D0_multiplier = value;
            else if (starts(name, "kappa_")) {
                for (size_t k = 0; k < npi_names.size(); ++k)
                    if (npi_names[k] == name) { npi_vals[k] = value; npi_update = true; break; }
            }
            // unknown names: reference prints a warning and continues (:264-266)
        }
        if (npi_update) {
            // setCalibratableValues (PieceWiseConstantNPIStrategy.cpp:228-262): negative -> throw
            for (double v : npi_vals)
                if (v < 0.0) throw std::invalid_argument("All NPI kappa values must be non-negative.");
            for (size_t k = 0; k < npi_vals.size(); ++k) up.kappa_values[k + 1] = npi_vals[k];
        }
        model.set(up);
    }
};

// -----------------------------------------------------------------------------
// The objective: src/model/objectives/SEPAIHRDObjectiveFunction.cpp:22-50,62-235
// obs_*: row-major num_obs_rows x n.
// -----------------------------------------------------------------------------
struct Problem {
    Params base;
    ParameterManager pm;
    std::vector<double> time_points;
    state_type initial_state;
    int num_obs_rows = 0;
    std::vector<double> obs_H, obs_ICU, obs_D;
    Solver solver = DOPRI5;
    double abs_err = 1e-6, rel_err = 1e-6, dt_hint = 1.0;
    long max_attempts = 1000000;
};

struct EvalInfo {
    StepStats steps;
    int status = 0;  // 0 ok, 1 = returned lowest(), 2 = SimulationException would propagate, 3 = step budget
    double ll_hosp = 0, ll_icu = 0, ll_deaths = 0;
};

inline double objective(const Problem& pb, const std::vector<double>& theta, EvalInfo* info = nullptr,
                        std::vector<double>* traj = nullptr) {
    const double LOWEST = std::numeric_limits<double>::lowest();
    EvalInfo local;
    EvalInfo& inf = info ? *info : local;
    inf = EvalInfo{};
    const std::vector<double>& tp = pb.time_points;
    // ctor :39-46
    int runup_offset = 0;
    for (size_t i = 0; i < tp.size(); ++i)
        if (tp[i] >= 0.0) { runup_offset = static_cast<int>(i); break; }
    const int num_obs_points = static_cast<int>(tp.size()) - runup_offset;
    if (tp.empty()) { inf.status = 1; return LOWEST; }  // :110-112

    Model model(pb.base);
    const int n = model.P.n;
    double total_pop = 0.0;
    for (int i = 0; i < n; ++i) total_pop += model.P.N[i];  // Eigen sum(): sequential for small n
    std::vector<double> age_fraction(n, 0.0);
    if (total_pop > 0.0)
        for (int i = 0; i < n; ++i) age_fraction[i] = model.P.N[i] / total_pop;

    try {
        pb.pm.updateModelParameters(theta, model);  // :117-122
    } catch (...) { inf.status = 1; return LOWEST; }

    state_type init = pb.initial_state;  // :126
    const double runup_days = model.P.runup_days, seed_exposed = model.P.seed_exposed;
    if (runup_days > 0 && seed_exposed > 0) {  // :131-143
        for (int i = 0; i < n; ++i) {
            init[i + n] = seed_exposed * age_fraction[i];
            for (int c = 2; c <= 10; ++c) init[i + c * n] = 0.0;
        }
    } else {  // :145-152
        const double mult[8] = {model.P.E0_multiplier, model.P.P0_multiplier, model.P.A0_multiplier,
                                model.P.I0_multiplier, model.P.H0_multiplier, model.P.ICU0_multiplier,
                                model.P.R0_multiplier, model.P.D0_multiplier};
        for (int c = 1; c <= 8; ++c)
            for (int i = 0; i < n; ++i) init[c * n + i] *= mult[c - 1];
    }
    for (int i = 0; i < n; ++i) {  // :155-163
        double sum = 0;
        for (int j = 1; j < NUM_POPULATION_COMPARTMENTS; ++j) sum += init[j * n + i];
        if (sum > model.P.N[i]) { inf.status = 1; return LOWEST; }
        init[i] = model.P.N[i] - sum;
    }

    thread_local SimulationResult res;
    inf.steps.max_attempts = pb.max_attempts;
    try {
        simulate(res, model, init, tp, pb.solver, pb.dt_hint, pb.abs_err, pb.rel_err, &inf.steps);
    } catch (const step_adjustment_error&) {
        inf.status = 2;  // SimulationException propagates out of calculate() (no try/catch at :165)
        return LOWEST;
    } catch (const step_budget_error&) {
        inf.status = 3;
        return LOWEST;
    }
    const size_t T = tp.size();
    if (traj) *traj = res.flat;
    if (num_obs_points != pb.num_obs_rows) { inf.status = 1; return LOWEST; }  // :176-178

    // :191-215 daily incidence from the cumulative compartments D (8n), CumH (9n), CumICU (10n)
    std::vector<double> sim_hosp(T * n), sim_icu(T * n), sim_deaths(T * n);
    auto diff = [&](std::vector<double>& out, int comp) {
        for (int i = 0; i < n; ++i) out[i] = res.row(0)[comp * n + i] - init[comp * n + i];
        for (size_t k = 1; k < T; ++k)
            for (int i = 0; i < n; ++i)
                out[k * n + i] = res.row(k)[comp * n + i] - res.row(k - 1)[comp * n + i];
        for (double& v : out) v = std::max(v, 0.0);  // cwiseMax(0.0)
    };
    diff(sim_hosp, 9);
    diff(sim_icu, 10);
    diff(sim_deaths, 8);
    // :218-225 bottom num_obs_points rows
    const size_t off = static_cast<size_t>(runup_offset) * n;
    inf.ll_hosp = poisson_loglik(sim_hosp.data() + off, pb.obs_H.data(), num_obs_points, n);
    inf.ll_icu = poisson_loglik(sim_icu.data() + off, pb.obs_ICU.data(), num_obs_points, n);
    inf.ll_deaths = poisson_loglik(sim_deaths.data() + off, pb.obs_D.data(), num_obs_points, n);
    double total = inf.ll_hosp + inf.ll_icu + inf.ll_deaths;
    if (std::isnan(total) || std::isinf(total)) { total = LOWEST; inf.status = 1; }  // :227
    return total;
}

// -----------------------------------------------------------------------------
// SEPAIHRDGradientObjectiveFunction::evaluate_with_gradient
// (src/model/objectives/SEPAIHRDGradientObjectiveFunction.cpp:15-171): forward differences,
// eps_i = epsilon max(|theta_i|, epsilon).  The perturbed runs do NOT go through calculate():
//   * a fresh parameter manager (default OPTIMIZATION_CLAMP mode) updates a cloned model (:40-54);
//   * the initial state is ALWAYS problem.initial_state scaled by the multipliers, which are read
//     from the UNCONSTRAINED perturbed vector by name and default to 1.0 when not calibrated (:59-83);
//   * S by subtraction, invalid (gradient entry 0) when the non-S total exceeds N or is negative (:85-101);
//   * the three Poisson terms use ALL output rows, so a row count different from the observations'
//     makes each of them lowest() (calculateSingleLogLikelihood's dimension check) (:126-158);
//   * a non-finite sum becomes lowest(), which is finite, so the difference quotient is still formed.
// -----------------------------------------------------------------------------
inline double evaluate_with_gradient(const Problem& pb, const std::vector<double>& theta, std::vector<double>& grad,
                                     double epsilon = 1e-4) {
    const double LOWEST = std::numeric_limits<double>::lowest();
    const int P = static_cast<int>(theta.size());
    grad.assign(P, 0.0);
    EvalInfo info;
    const double f_center = objective(pb, theta, &info);
    if (info.status >= 2) throw std::runtime_error("SimulationException");
    if (!std::isfinite(f_center)) return f_center;
    const int n = pb.base.n;
    const size_t T = pb.time_points.size();
    static const char* mult_names[8] = {"E0_multiplier", "P0_multiplier", "A0_multiplier", "I0_multiplier",
                                        "H0_multiplier", "ICU0_multiplier", "R0_multiplier", "D0_multiplier"};
    for (int i = 0; i < P; ++i) {
        const double param_scale = std::max(std::abs(theta[i]), epsilon);
        const double eps_i = epsilon * param_scale;
        Model model(pb.base);
        ParameterManager temp = pb.pm;
        temp.mode = OPTIMIZATION_CLAMP;
        std::vector<double> plus = theta;
        plus[i] += eps_i;
        try {
            temp.updateModelParameters(plus, model);
        } catch (...) { grad[i] = 0.0; continue; }
        state_type init = pb.initial_state;
        for (int c = 1; c <= 8; ++c) {
            double mult = 1.0;
            for (int k = 0; k < P; ++k)
                if (pb.pm.names[k] == mult_names[c - 1]) { mult = plus[k]; break; }
            for (int a = 0; a < n; ++a) init[c * n + a] *= mult;
        }
        bool valid = true;
        for (int a = 0; a < n; ++a) {
            double sum = 0.0;
            for (int j = 1; j < NUM_POPULATION_COMPARTMENTS; ++j) sum += init[j * n + a];
            if (sum > model.P.N[a] || sum < 0) { valid = false; break; }
            init[a] = model.P.N[a] - sum;
        }
        if (!valid) { grad[i] = 0.0; continue; }
        thread_local SimulationResult res;
        StepStats st;
        st.max_attempts = pb.max_attempts;
        simulate(res, model, init, pb.time_points, pb.solver, pb.dt_hint, pb.abs_err, pb.rel_err, &st);
        double f_plus;
        if (static_cast<int>(T) != pb.num_obs_rows) {
            f_plus = LOWEST + LOWEST + LOWEST;  // -inf
        } else {
            std::vector<double> sh(T * n), si(T * n), sd(T * n);
            auto diff = [&](std::vector<double>& out, int comp) {
                for (int a = 0; a < n; ++a) out[a] = res.row(0)[comp * n + a] - init[comp * n + a];
                for (size_t k = 1; k < T; ++k)
                    for (int a = 0; a < n; ++a) out[k * n + a] = res.row(k)[comp * n + a] - res.row(k - 1)[comp * n + a];
                for (double& v : out) v = std::max(v, 0.0);
            };
            diff(sh, 9); diff(si, 10); diff(sd, 8);
            const int rows = static_cast<int>(T);
            f_plus = poisson_loglik(sh.data(), pb.obs_H.data(), rows, n) + poisson_loglik(si.data(), pb.obs_ICU.data(), rows, n) +
                     poisson_loglik(sd.data(), pb.obs_D.data(), rows, n);
        }
        if (std::isnan(f_plus) || std::isinf(f_plus)) f_plus = LOWEST;
        grad[i] = std::isfinite(f_plus) ? (f_plus - f_center) / eps_i : 0.0;
    }
    return f_center;
}

// -----------------------------------------------------------------------------
// Posterior ensemble (second consumer of the integrator).
//   SimulationRunner::runSimulation       src/model/SimulationRunner.cpp:24-104
//     theta -> model (updateModelParameters, PostCalibrationAnalyser.cpp:212-218), then the
//     simulation starts from the GIVEN initial state: no run-up / multiplier re-derivation.
//   ResultAggregator::aggregatePosteriorPredictives  src/model/ResultAggregator.cpp:297-345
//     daily = max(0, X(t) - X(t_prev)) on the output times >= 0, cumulative = running sums.
//   MetricsCalculator::calculateSeroprevalenceTrajectory  src/model/MetricsCalculator.cpp:199-226
//   quantile rule  PostCalibrationAnalyser.cpp:303-340 (exact sort, interpolate at q (n - 1)).
// The reference estimates the incidence quantiles with Boost.Accumulators' P^2 (third-party,
// order-dependent); this restatement and the device path use the exact-sort rule throughout.
// -----------------------------------------------------------------------------
inline int simulate_sample(const Problem& pb, const std::vector<double>& theta, std::vector<double>& traj) {
    Model model(pb.base);
    try {
        pb.pm.updateModelParameters(theta, model);
    } catch (...) { return 1; }
    thread_local SimulationResult res;
    StepStats st;
    st.max_attempts = pb.max_attempts;
    try {
        simulate(res, model, pb.initial_state, pb.time_points, pb.solver, pb.dt_hint, pb.abs_err, pb.rel_err, &st);
    } catch (const step_adjustment_error&) { return 2;
    } catch (const step_budget_error&) { return 3; }
    traj = res.flat;
    return 0;
}

// sample selection of ResultAggregator::aggregatePosteriorPredictives (ResultAggregator.cpp:246-266)
inline std::vector<int> select_ppc_samples(size_t n_samples, int num_samples_for_ppc, unsigned int seed) {
    std::vector<int> sel;
    if (num_samples_for_ppc > 0 && static_cast<size_t>(num_samples_for_ppc) < n_samples) {
        std::mt19937 gen;
        gen.seed(seed);
        std::uniform_int_distribution<> distrib(0, static_cast<int>(n_samples) - 1);
        for (int i = 0; i < num_samples_for_ppc; ++i) sel.push_back(distrib(gen));
    } else {
        for (size_t i = 0; i < n_samples; ++i) sel.push_back(static_cast<int>(i));
    }
    return sel;
}

inline double sorted_quantile(const std::vector<double>& v, double q) {
    const double pos = q * (v.size() - 1);
    const size_t idx = static_cast<size_t>(pos);
    const double frac = pos - idx;
    if (idx + 1 < v.size()) return v[idx] * (1.0 - frac) + v[idx + 1] * frac;
    return v[idx];
}

struct EnsembleSummary {
    int Tp = 0, n_valid = 0;
    std::vector<double> ppc;     // [6][n_probs][Tp][n]
    std::vector<double> sero;    // [n_probs][T]
    std::vector<int> status;     // [S]
};

inline EnsembleSummary ensemble_summaries(const Problem& pb, const double* thetas, int S,
                                          const std::vector<double>& probs, int nthreads = 1) {
    const std::vector<double>& tp = pb.time_points;
    const int T = static_cast<int>(tp.size()), n = pb.base.n, P = static_cast<int>(pb.pm.names.size());
    std::vector<int> pos_idx;
    for (int i = 0; i < T; ++i)
        if (tp[i] >= 0.0) pos_idx.push_back(i);
    EnsembleSummary out;
    const int Tp = out.Tp = static_cast<int>(pos_idx.size());
    const int np = static_cast<int>(probs.size());
    out.status.assign(S, 0);
    double total_pop = 0.0;
    for (int i = 0; i < n; ++i) total_pop += pb.base.N[i];
    // per-sample series, sample-major
    std::vector<double> series(static_cast<size_t>(S) * 6 * Tp * n), sero(static_cast<size_t>(S) * T);
#pragma omp parallel for num_threads(nthreads) schedule(dynamic)
    for (int s = 0; s < S; ++s) {
        std::vector<double> theta(thetas + static_cast<size_t>(s) * P, thetas + static_cast<size_t>(s + 1) * P), traj;
        out.status[s] = simulate_sample(pb, theta, traj);
        if (out.status[s] != 0) continue;
        const size_t W = static_cast<size_t>(NUM_COMPARTMENTS) * n;
        double* dst = &series[static_cast<size_t>(s) * 6 * Tp * n];
        const int comps[3] = {9, 10, 8};
        for (int ser = 0; ser < 3; ++ser) {
            std::vector<double> prev(n);
            const int first = pos_idx.empty() ? 0 : pos_idx[0];
            for (int a = 0; a < n; ++a)
                prev[a] = first > 0 ? traj[(first - 1) * W + comps[ser] * n + a] : pb.initial_state[comps[ser] * n + a];
            for (int t = 0; t < Tp; ++t) {
                const double* row = &traj[pos_idx[t] * W + comps[ser] * n];
                for (int a = 0; a < n; ++a) {
                    const double daily = std::max(0.0, row[a] - prev[a]);
                    prev[a] = row[a];
                    dst[(static_cast<size_t>(ser) * Tp + t) * n + a] = daily;
                    dst[(static_cast<size_t>(ser + 3) * Tp + t) * n + a] =
                        t == 0 ? daily : dst[(static_cast<size_t>(ser + 3) * Tp + t - 1) * n + a] + daily;
                }
            }
        }
        for (int k = 0; k < T; ++k) {
            double tot = 0.0;
            for (int a = 0; a < n; ++a) tot += traj[k * W + a];
            sero[static_cast<size_t>(s) * T + k] = (total_pop - tot) / total_pop;
        }
    }
    out.ppc.assign(static_cast<size_t>(6) * np * Tp * n, std::numeric_limits<double>::quiet_NaN());
    out.sero.assign(static_cast<size_t>(np) * T, std::numeric_limits<double>::quiet_NaN());
    for (int s = 0; s < S; ++s) out.n_valid += out.status[s] == 0;
    if (out.n_valid == 0) return out;
    std::vector<double> v;
    for (int ser = 0; ser < 6; ++ser)
        for (int t = 0; t < Tp; ++t)
            for (int a = 0; a < n; ++a) {
                v.clear();
                for (int s = 0; s < S; ++s)
                    if (out.status[s] == 0) v.push_back(series[static_cast<size_t>(s) * 6 * Tp * n + (static_cast<size_t>(ser) * Tp + t) * n + a]);
                std::sort(v.begin(), v.end());
                for (int p = 0; p < np; ++p)
                    out.ppc[((static_cast<size_t>(ser) * np + p) * Tp + t) * n + a] = sorted_quantile(v, probs[p]);
            }
    for (int k = 0; k < T; ++k) {
        v.clear();
        for (int s = 0; s < S; ++s)
            if (out.status[s] == 0) v.push_back(sero[static_cast<size_t>(s) * T + k]);
        std::sort(v.begin(), v.end());
        for (int p = 0; p < np; ++p) out.sero[static_cast<size_t>(p) * T + k] = sorted_quantile(v, probs[p]);
    }
    return out;
}

// -----------------------------------------------------------------------------
// SimulationCache::computeHash  (src/sir_age_structured/caching/SimulationCache.cpp:12-19,35-52)
// -----------------------------------------------------------------------------
inline uint64_t mix_hash(uint64_t k) {
    k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33;
    return k;
}
inline uint64_t cache_hash(const double* p, int size) {
    uint64_t seed = 0;
    for (int i = 0; i < size; ++i) {
        long long q = static_cast<long long>(p[i] * 1e8 + 0.5);
        uint64_t k = static_cast<uint64_t>(q);
        seed ^= mix_hash(k) + 0x9e3779b9ULL + (seed << 6) + (seed >> 2);
    }
    return seed;
}

// -----------------------------------------------------------------------------
// Synthetic parameter draws: the reference benchmark's jitter protocol
// (src/model/sepaihrd_objective_benchmark_main.cpp:412-413,452-460), one
// mt19937(seed) + one persistent normal_distribution per chain.
// -----------------------------------------------------------------------------
inline std::vector<double> jitter_draw(const ParameterManager& pm, const std::vector<double>& base,
                                       uint32_t seed) {
    std::mt19937 rng(seed);
    std::normal_distribution<double> normal(0.0, 1.0);
    std::vector<double> cand(base.size());
    for (size_t i = 0; i < base.size(); ++i) {
        const double s = pm.sigmas.at(pm.names[i]);
        cand[i] = base[i] + s * normal(rng);
    }
    return pm.applyConstraints(cand);
}

// -----------------------------------------------------------------------------
// Adaptive-Metropolis sampler restated with plain loops.
// src/sir_age_structured/optimizers/MetropolisHastingsSampler.cpp:65-412.
// A seed replaces the reference's std::random_device (:20-23) -- a build-side
// addition.  Eigen::LLT is replaced by a textbook
// lower Cholesky; Eigen's blocked evaluation order is not reproduced
// (parity unpinned for those bits).
// -----------------------------------------------------------------------------
struct MHSettings {
    int iterations = 10000, burn_in = 1000, adaptation_period = 100, thinning = 1;
    double regularization_epsilon = 1e-6, target_acceptance_rate = 0.234;
    bool adapt_scale = true;
    std::vector<double> initial_cov;  // setInitialCovariance (:52-63): P x P row-major, empty = none
    // recomputeFullCovariance (:168-199): false = running co-moments (RunningMoments below, O(P^2) per iteration whatever t),
    // true = the literal two passes over the whole history (O(t P^2) per refresh, every state kept)
    bool two_pass_covariance = false;
};

// -----------------------------------------------------------------------------
// The sample covariance of recomputeFullCovariance (:168-199) carried as running sums, so that a refresh costs
// O(P^2) instead of a walk over the whole chain history and no history has to be kept.
//   sum    plain running sum of the states, added in the order the reference's own loop adds them
//          (`for vec in chain_history_: mean += vec`, :171-174): sum / len IS the reference's mean, bit for bit;
//   mean,  Welford's recurrence for the centred second moment, n = number of states so far, d = x - mean (old):
//   m2        m2_ij += ((n - 1) / n) * (d_i * d_j);   mean_i += d_i * (1 / n)
//          (the symmetric form: the bits of m2_ij and m2_ji are the same, only j <= i is stored).
// At a refresh  cov_ij = scaling * (m2_ij / (len - 1)) + eps [i == j].  The reference forms the same matrix as
// `centered.adjoint() * centered` through Eigen's blocked GEMM (:184), whose summation order nothing in the
// reference tree pins; this recurrence is the order the build chose, stated once here and followed by the host
// library and the device kernels operation for operation (no contraction), so that all three give the same bits.
// Agreement with the two-pass form: ~1e-13 relative (tests/test_oracle_golden.py).
// -----------------------------------------------------------------------------
struct RunningMoments {
    int P = 0;
    long n = 0;
    std::vector<double> sum, mean, m2;  // m2: P x P row-major, entries j <= i
    explicit RunningMoments(int P_ = 0) : P(P_), sum(P_, 0.0), mean(P_, 0.0), m2(static_cast<size_t>(P_) * P_, 0.0) {}
    void push(const double* x) {
        n += 1;
        const double rn = 1.0 / static_cast<double>(n);
        const double w = static_cast<double>(n - 1) / static_cast<double>(n);
        std::vector<double> d(P);
        for (int i = 0; i < P; ++i) d[i] = x[i] - mean[i];
        for (int i = 0; i < P; ++i)
            for (int j = 0; j <= i; ++j) m2[static_cast<size_t>(i) * P + j] += w * (d[i] * d[j]);
        for (int i = 0; i < P; ++i) mean[i] += d[i] * rn;
        for (int i = 0; i < P; ++i) sum[i] += x[i];
    }
    // running_mean_ = mean (:175-176) and current_covariance_ (:184-190) of a history of n states
    void covariance(double scaling, double eps, std::vector<double>& running_mean, std::vector<double>& cov) const {
        const double len = static_cast<double>(n), denom = static_cast<double>(n - 1);
        for (int i = 0; i < P; ++i) running_mean[i] = sum[i] / len;
        for (int i = 0; i < P; ++i)
            for (int j = 0; j < P; ++j) {
                const double m = i >= j ? m2[static_cast<size_t>(i) * P + j] : m2[static_cast<size_t>(j) * P + i];
                cov[static_cast<size_t>(i) * P + j] = scaling * (m / denom) + (i == j ? eps : 0.0);
            }
    }
};

struct MHResult {
    std::vector<double> best;
    double best_value = -std::numeric_limits<double>::infinity();
    std::vector<std::vector<double>> samples;
    std::vector<double> sample_values;
    int accepted = 0;
    double final_scale = 1.0;
    std::vector<double> final_cov;  // P x P row-major
    std::vector<unsigned char> accept_trace;
};

inline bool cholesky_lower(const std::vector<double>& A, int P, std::vector<double>& L) {
    L.assign(static_cast<size_t>(P) * P, 0.0);
    for (int j = 0; j < P; ++j) {
        double d = A[j * P + j];
        for (int k = 0; k < j; ++k) d -= L[j * P + k] * L[j * P + k];
        if (!(d > 0.0)) return false;
        const double ljj = std::sqrt(d);
        L[j * P + j] = ljj;
        for (int i = j + 1; i < P; ++i) {
            double s = A[i * P + j];
            for (int k = 0; k < j; ++k) s -= L[i * P + k] * L[j * P + k];
            L[i * P + j] = s / ljj;
        }
    }
    return true;
}

using Objective = std::function<double(const std::vector<double>&)>;

inline MHResult metropolis_hastings(const MHSettings& cfg, const std::vector<double>& x0,
                                    const Objective& objective_fn, ParameterManager& pm,
                                    uint32_t seed) {
    pm.mode = MCMC_REFLECT;  // :207-209
    const int P = static_cast<int>(x0.size());
    std::mt19937 gen(seed);
    auto safe_eval = [&](const std::vector<double>& p) {  // :65-74
        try {
            double v = objective_fn(p);
            if (std::isnan(v) || std::isinf(v)) return -1e18;
            return v;
        } catch (...) { return -1e18; }
    };
    MHResult r;
    std::vector<double> cur = x0;
    std::vector<double> cov(static_cast<size_t>(P) * P, 0.0);
    const double scaling_factor = (2.38 * 2.38) / static_cast<double>(P);
    if (cfg.initial_cov.size() == cov.size()) {  // :219-223 warm start, no scaling
        cov = cfg.initial_cov;
    } else {
        for (int i = 0; i < P; ++i) {  // :226-233
            double s = pm.sigmas.at(pm.names[i]);
            cov[i * P + i] = (s > 0 ? s * s : 1e-6);
        }
        for (double& v : cov) v *= scaling_factor;
    }
    for (int i = 0; i < P; ++i) cov[i * P + i] += cfg.regularization_epsilon;  // :237
    std::vector<double> L;
    if (!cholesky_lower(cov, P, L)) {  // :240-246
        L.assign(static_cast<size_t>(P) * P, 0.0);
        for (int i = 0; i < P; ++i) L[i * P + i] = 0.1;
    }
    std::vector<double> running_mean = x0;
    double log_scale = 0.0, global_scale = 1.0;
    double cur_lp = safe_eval(cur);
    // chain_history_ (:262-264): kept whole only for the literal two-pass refresh; otherwise its running sums
    // and the newest state, which is all updateCovarianceRank1 reads (:157)
    std::vector<std::vector<double>> history;
    RunningMoments moments(P);
    size_t history_len = 1;
    if (cfg.two_pass_covariance) history.reserve(cfg.iterations);
    history.push_back(cur);
    moments.push(cur.data());
    r.samples.push_back(cur);
    r.sample_values.push_back(cur_lp);
    r.best = cur;
    r.best_value = cur_lp;
    std::deque<int> recent;
    int emergency = 0;
    std::uniform_real_distribution<double> u_dist(0.0, 1.0);
    r.accept_trace.reserve(cfg.iterations);

    for (int t = 1; t < cfg.iterations; ++t) {
        if (t > cfg.burn_in) {
            {   // updateCovarianceRank1 :154-166
                const std::vector<double>& ns = history.back();
                const double gamma = 10.0 / (t + 100.0);
                std::vector<double> diff(P);
                for (int i = 0; i < P; ++i) diff[i] = ns[i] - running_mean[i];
                for (int i = 0; i < P; ++i) running_mean[i] += gamma * diff[i];
                for (int i = 0; i < P; ++i)
                    for (int j = 0; j < P; ++j)
                        cov[i * P + j] = (1.0 - gamma) * cov[i * P + j] + gamma * (diff[i] * diff[j]);
            }
            if (t % cfg.adaptation_period == 0) {
                // recomputeFullCovariance :168-199
                if (history_len >= static_cast<size_t>(P) + 10 && !cfg.two_pass_covariance) {
                    moments.covariance(scaling_factor, cfg.regularization_epsilon, running_mean, cov);
                    std::vector<double> Ltry;
                    if (cholesky_lower(cov, P, Ltry)) L = Ltry;
                } else if (history_len >= static_cast<size_t>(P) + 10) {
                    std::vector<double> mean(P, 0.0);
                    for (const auto& v : history)
                        for (int i = 0; i < P; ++i) mean[i] += v[i];
                    for (int i = 0; i < P; ++i) mean[i] /= static_cast<double>(history.size());
                    running_mean = mean;
                    std::vector<double> c(static_cast<size_t>(P) * P, 0.0);
                    for (const auto& v : history)
                        for (int i = 0; i < P; ++i) {
                            const double di = v[i] - mean[i];
                            for (int j = 0; j < P; ++j) c[i * P + j] += di * (v[j] - mean[j]);
                        }
                    const double denom = double(history.size() - 1);
                    for (int i = 0; i < P; ++i)
                        for (int j = 0; j < P; ++j)
                            cov[i * P + j] = scaling_factor * (c[i * P + j] / denom) +
                                             (i == j ? cfg.regularization_epsilon : 0.0);
                    std::vector<double> Ltry;
                    if (cholesky_lower(cov, P, Ltry)) L = Ltry;
                }
                // :295-300  epsilon added again before the LLT that is actually kept
                std::vector<double> stable = cov;
                for (int i = 0; i < P; ++i) stable[i * P + i] += cfg.regularization_epsilon;
                std::vector<double> Ltry;
                if (cholesky_lower(stable, P, Ltry)) L = Ltry;
            }
        }
        // generateProposal :91-102 (fresh normal_distribution per call)
        std::vector<double> z(P);
        {
            std::normal_distribution<double> dist(0.0, 1.0);
            for (int i = 0; i < P; ++i) z[i] = dist(gen);
        }
        std::vector<double> raw(P);
        for (int i = 0; i < P; ++i) {
            double s = 0.0;
            for (int j = 0; j <= i; ++j) s += L[i * P + j] * z[j];
            raw[i] = cur[i] + global_scale * s;
        }
        std::vector<double> prop = pm.applyConstraints(raw);  // :309
        const double prop_lp = safe_eval(prop);               // :312
        const double log_ratio = prop_lp - cur_lp;
        bool accept = false;
        if (log_ratio >= 0.0) accept = true;
        else if (std::log(u_dist(gen)) < log_ratio) accept = true;  // U drawn only here (:327)
        if (accept) {
            cur = prop;
            cur_lp = prop_lp;
            r.accepted++;
            if (cur_lp > r.best_value) { r.best_value = cur_lp; r.best = cur; }
        }
        r.accept_trace.push_back(accept ? 1 : 0);
        if (cfg.adapt_scale) {  // adaptGlobalScale :104-152
            recent.push_back(accept ? 1 : 0);
            if (recent.size() > 1000) recent.pop_front();
            double rate = 0.0;
            if (!recent.empty()) {
                int sum = 0;
                for (int a : recent) sum += a;
                rate = static_cast<double>(sum) / recent.size();
            }
            if (recent.size() >= 1000 && rate < 0.001) {
                log_scale -= 0.7;
                emergency++;
            } else if (rate < 0.02 && recent.size() >= 500) {
                double g = 5.0 / std::sqrt(static_cast<double>(t) + 1.0);
                g = std::min(g, 0.3);
                log_scale += g * (0.0 - cfg.target_acceptance_rate);
            } else {
                double g = 1.0 / std::sqrt(static_cast<double>(t) + 1.0);
                g = std::min(g, 0.1);
                log_scale += g * ((accept ? 1.0 : 0.0) - cfg.target_acceptance_rate);
            }
            if (global_scale <= 0.011 && rate > 0.15 && rate < 0.30) log_scale += 0.01;
            log_scale = std::max(std::min(log_scale, 2.3), -6.9);
            global_scale = std::exp(log_scale);
        }
        if (cfg.two_pass_covariance) history.push_back(cur);
        else history.back() = cur;
        history_len++;
        moments.push(cur.data());
        if (t % cfg.thinning == 0) {
            r.samples.push_back(cur);
            r.sample_values.push_back(cur_lp);
        }
    }
    (void)emergency;
    r.final_scale = global_scale;
    r.final_cov = cov;
    return r;
}

// -----------------------------------------------------------------------------
// HillClimbingOptimizer (src/sir_age_structured/optimizers/HillClimbingOptimizer.cpp:24-353):
// candidate cloud (half correlated L z moves, half axis-aligned moves), winner, early accept,
// backtracking + moving-anchor expansion line search (:39-112), rank-one covariance adaptation with
// symmetrisation / jitter / diagonal floor (:279-307), Cholesky refresh every 10 iterations with
// regularisation fallback (:310-341).  One objective call at a time, in the reference's order.
// The reference seeds from std::random_device (:165-183); here the master generator takes a seed and
// "threads" is a setting (virtual threads: thread t owns the candidates OpenMP's static schedule
// would give it, with its own mt19937 and its own persistent normal_distribution, :186-214).
// Dense products are restated element-wise (Eigen's evaluation order is not pinned by the reference).
// -----------------------------------------------------------------------------
struct HCSettings {
    int iterations = 2000;
    int cloud_size_multiplier = 8;
    int threads = 1;
};

struct HCResult {
    std::vector<double> best;
    double best_value = 0.0;
    std::vector<double> final_cov;          // P x P row-major
    std::vector<double> current_trace;      // current_logL after every iteration
    long evaluations = 0;
};

inline HCResult hill_climbing(const HCSettings& cfg, const std::vector<double>& x0, const Objective& objective_fn,
                              ParameterManager& pm, uint32_t seed) {
    const int P = static_cast<int>(x0.size());
    HCResult r;
    auto safe_eval = [&](const std::vector<double>& p) {  // :24-32
        ++r.evaluations;
        try {
            double v = objective_fn(p);
            if (std::isnan(v) || std::isinf(v)) return -1e18;
            return v;
        } catch (...) { return -1e18; }
    };
    auto sigma_of = [&](int i) { return pm.sigmas.at(pm.names[i]); };
    auto add = [&](const std::vector<double>& a, const std::vector<double>& b, double sb) {
        std::vector<double> o(P);
        for (int i = 0; i < P; ++i) o[i] = a[i] + b[i] * sb;
        return o;
    };
    r.best = x0;
    r.best_value = safe_eval(x0);
    std::vector<double> cur = x0, prev = x0;
    double cur_l = r.best_value;
    std::vector<double> cov(static_cast<size_t>(P) * P, 0.0), L;
    for (int i = 0; i < P; ++i) {  // :146-150
        const double s = sigma_of(i);
        cov[i * P + i] = (s > 0 ? s * s : 1e-4);
    }
    cholesky_lower(cov, P, L);  // :153
    const int V = std::max(1, cfg.threads);
    const int nc = std::max(4, V * std::max(1, cfg.cloud_size_multiplier));  // :163
    std::mt19937 master(seed);
    std::vector<std::mt19937> rngs(V);
    std::vector<std::normal_distribution<double>> norms(V);
    for (int t = 0; t < V; ++t) {  // :179-183
        const uint32_t a = static_cast<uint32_t>(master()), b = static_cast<uint32_t>(master()),
                       c = static_cast<uint32_t>(master()), d = static_cast<uint32_t>(master());
        std::seed_seq sq{a, b, c, d};
        rngs[t].seed(sq);
        norms[t] = std::normal_distribution<double>(0.0, 1.0);
    }
    // static schedule: thread t owns a contiguous block, the first nc % V threads one more
    std::vector<int> owner(nc);
    {
        const int q = nc / V, rem = nc % V;
        int i = 0;
        for (int t = 0; t < V; ++t)
            for (int k = 0; k < q + (t < rem ? 1 : 0); ++k) owner[i++] = t;
    }
    auto line_search = [&](std::vector<double>& params, double& logL, const std::vector<double>& direction) {  // :39-112
        const double shrinkage = 0.5, growth = 2.0;
        const int max_backtrack = 10, max_expansion = 12;
        double step = 1.0;
        std::vector<double> improved = params;
        double improved_l = logL;
        bool found = false;
        for (int i = 0; i < max_backtrack; ++i) {
            const std::vector<double> cand = pm.applyConstraints(add(params, direction, step));
            double sq = 0.0;
            for (int k = 0; k < P; ++k) sq += (cand[k] - params[k]) * (cand[k] - params[k]);
            if (sq < 1e-16) break;
            const double l = safe_eval(cand);
            if (l > improved_l) { improved = cand; improved_l = l; found = true; break; }
            step *= shrinkage;
        }
        if (!found) return false;
        std::vector<double> best = improved;
        double best_l = improved_l;
        std::vector<double> cur_step(P);
        for (int k = 0; k < P; ++k) cur_step[k] = improved[k] - params[k];
        for (int i = 0; i < max_expansion; ++i) {
            for (int k = 0; k < P; ++k) cur_step[k] *= growth;
            const std::vector<double> cand = pm.applyConstraints(add(best, cur_step, 1.0));
            const double l = safe_eval(cand);
            if (l > best_l) { best = cand; best_l = l; } else break;
        }
        params = best;
        logL = best_l;
        return true;
    };
    std::vector<std::vector<double>> cand(nc, std::vector<double>(P)), ccand(nc, std::vector<double>(P));
    std::vector<double> scores(nc, -1e18), z(P);
    for (int iter = 0; iter < cfg.iterations; ++iter) {
        for (int i = 0; i < nc; ++i) {  // :192-214, per owner thread in index order
            std::mt19937& g = rngs[owner[i]];
            std::normal_distribution<double>& nd = norms[owner[i]];
            if (i < nc / 2) {
                for (int k = 0; k < P; ++k) z[k] = nd(g);
                for (int a = 0; a < P; ++a) {
                    double sum = 0.0;
                    for (int b = 0; b <= a; ++b) sum += L[a * P + b] * z[b];
                    cand[i][a] = sum;
                }
            } else {
                std::uniform_int_distribution<int> param_dist(0, P - 1);
                const int idx = param_dist(g);
                const double sg = std::sqrt(cov[idx * P + idx]);
                std::fill(cand[i].begin(), cand[i].end(), 0.0);
                cand[i][idx] = sg * nd(g);
            }
        }
        for (int i = 0; i < nc; ++i) {  // :222-228
            ccand[i] = pm.applyConstraints(add(cur, cand[i], 1.0));
            scores[i] = safe_eval(ccand[i]);
        }
        int best_idx = -1;
        double best_val = -1e18;
        for (int i = 0; i < nc; ++i)
            if (scores[i] > best_val) { best_val = scores[i]; best_idx = i; }
        bool moved = false;
        if (best_idx != -1 && best_val > -1e18) {  // :241-262
            const std::vector<double> best_point = ccand[best_idx];
            std::vector<double> dir(P);
            for (int k = 0; k < P; ++k) dir[k] = best_point[k] - cur[k];
            if (best_val > cur_l) { cur = best_point; cur_l = best_val; moved = true; }
            const bool ls = line_search(cur, cur_l, dir);
            moved = moved || ls;
        }
        if (moved) {  // :265-308
            if (cur_l > r.best_value) { r.best_value = cur_l; r.best = cur; }
            std::vector<double> st(P);
            double norm2 = 0.0;
            for (int k = 0; k < P; ++k) { st[k] = cur[k] - prev[k]; norm2 += st[k] * st[k]; }
            if (norm2 > 1e-14) {
                const double alpha = 2.0 / (P + 2.0);
                for (int a = 0; a < P; ++a)
                    for (int b = 0; b < P; ++b) {
                        double v = cov[a * P + b] * (1.0 - alpha);
                        v += alpha * (st[a] * st[b]);
                        cov[a * P + b] = v;
                    }
                std::vector<double> sym(cov.size());
                for (int a = 0; a < P; ++a)
                    for (int b = 0; b < P; ++b) sym[a * P + b] = 0.5 * (cov[a * P + b] + cov[b * P + a]);
                cov = sym;
                double tr = 0.0;
                for (int a = 0; a < P; ++a) tr += cov[a * P + a];
                const double jitter = 1e-8 * tr / P;
                for (int a = 0; a < P; ++a) cov[a * P + a] += jitter;
                for (int a = 0; a < P; ++a) {
                    double mv = sigma_of(a);
                    mv = (mv > 0 ? mv * mv * 0.01 : 1e-8);
                    if (cov[a * P + a] < mv) cov[a * P + a] = mv;
                }
            }
            prev = cur;
        }
        if (iter > 0 && iter % 10 == 0) {  // :313-341
            std::vector<double> Ln;
            if (cholesky_lower(cov, P, Ln)) {
                L = Ln;
            } else {
                double tr = 0.0;
                for (int a = 0; a < P; ++a) tr += cov[a * P + a];
                double lambda = 1e-6 * tr / P;
                bool regularized = false;
                for (int attempt = 0; attempt < 5; ++attempt) {
                    for (int a = 0; a < P; ++a) cov[a * P + a] += lambda;
                    if (cholesky_lower(cov, P, Ln)) { L = Ln; regularized = true; break; }
                    lambda *= 10.0;
                }
                if (!regularized) {
                    L.assign(static_cast<size_t>(P) * P, 0.0);
                    for (int a = 0; a < P; ++a) L[a * P + a] = std::sqrt(cov[a * P + a]);
                    for (int a = 0; a < P; ++a)
                        for (int b = 0; b < P; ++b)
                            if (a != b) cov[a * P + b] = 0.0;
                }
            }
        }
        r.current_trace.push_back(cur_l);
    }
    r.final_cov = cov;
    return r;
}

// -----------------------------------------------------------------------------
// ModelCalibrator (src/sir_age_structured/ModelCalibrator.cpp:13-159): initial objective value,
// phase 1 in OPTIMIZATION_CLAMP mode, conditioning of the phase-1 covariance (symmetrise, eigenvalue
// floor at (0.1 sigma_i)^2 with the i-th SMALLEST eigenvalue paired with parameter i as the reference
// does, x4 inflation, trace jitter :93-131), phase 2 in MCMC_REFLECT mode warm-started from it,
// objective value of every stored sample (:141-144).
// The reference takes the eigen-decomposition from Eigen::SelfAdjointEigenSolver (not pinned); the
// restatement uses a cyclic Jacobi iteration, eigenpairs sorted ascending like Eigen returns them.
// -----------------------------------------------------------------------------
inline void jacobi_eigen_sym(std::vector<double> A, int P, std::vector<double>& evals, std::vector<double>& evecs) {
    std::vector<double> V(static_cast<size_t>(P) * P, 0.0);
    for (int i = 0; i < P; ++i) V[i * P + i] = 1.0;
    for (int sweep = 0; sweep < 100; ++sweep) {
        double off = 0.0, diag = 0.0;
        for (int a = 0; a < P; ++a)
            for (int b = 0; b < P; ++b) (a == b ? diag : off) += A[a * P + b] * A[a * P + b];
        if (off <= 1e-30 * diag || off == 0.0) break;
        for (int p = 0; p < P - 1; ++p)
            for (int q = p + 1; q < P; ++q) {
                const double apq = A[p * P + q];
                if (apq == 0.0) continue;
                const double theta = (A[q * P + q] - A[p * P + p]) / (2.0 * apq);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), sn = t * c;
                for (int k = 0; k < P; ++k) {  // columns p, q
                    const double akp = A[k * P + p], akq = A[k * P + q];
                    A[k * P + p] = c * akp - sn * akq;
                    A[k * P + q] = sn * akp + c * akq;
                }
                for (int k = 0; k < P; ++k) {  // rows p, q
                    const double apk = A[p * P + k], aqk = A[q * P + k];
                    A[p * P + k] = c * apk - sn * aqk;
                    A[q * P + k] = sn * apk + c * aqk;
                }
                for (int k = 0; k < P; ++k) {
                    const double vkp = V[k * P + p], vkq = V[k * P + q];
                    V[k * P + p] = c * vkp - sn * vkq;
                    V[k * P + q] = sn * vkp + c * vkq;
                }
            }
    }
    std::vector<int> order(P);
    for (int i = 0; i < P; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return A[a * P + a] < A[b * P + b]; });
    evals.resize(P);
    evecs.assign(static_cast<size_t>(P) * P, 0.0);
    for (int k = 0; k < P; ++k) {
        evals[k] = A[order[k] * P + order[k]];
        for (int a = 0; a < P; ++a) evecs[a * P + k] = V[a * P + order[k]];
    }
}

inline std::vector<double> condition_phase1_covariance(const std::vector<double>& cov_in, int P,
                                                       const std::function<double(int)>& sigma_of) {
    std::vector<double> cov(cov_in.size());
    for (int a = 0; a < P; ++a)
        for (int b = 0; b < P; ++b) cov[a * P + b] = 0.5 * (cov_in[a * P + b] + cov_in[b * P + a]);  // :100
    std::vector<double> evals, evecs;
    jacobi_eigen_sym(cov, P, evals, evecs);
    for (int i = 0; i < P; ++i) {  // :107-111
        const double min_var = std::pow(sigma_of(i) * 0.1, 2);
        evals[i] = std::max(evals[i], min_var);
    }
    std::vector<double> ql(cov.size()), out(cov.size());
    for (int a = 0; a < P; ++a)
        for (int k = 0; k < P; ++k) ql[a * P + k] = evecs[a * P + k] * evals[k];
    for (int a = 0; a < P; ++a)
        for (int b = 0; b < P; ++b) {  // Q L' Q^T (:114)
            double sum = 0.0;
            for (int k = 0; k < P; ++k) sum += ql[a * P + k] * evecs[b * P + k];
            out[a * P + b] = sum * 4.0;  // :117
        }
    double tr = 0.0;
    for (int a = 0; a < P; ++a) tr += out[a * P + a];
    const double eps = 1e-8 * tr / P;  // :120-121
    for (int a = 0; a < P; ++a) out[a * P + a] += eps;
    return out;
}

struct CalibrationResult {
    double initial_value = 0.0;
    HCResult phase1;
    MHResult phase2;
    std::vector<double> phase2_cov;  // conditioned covariance handed to the sampler
    std::vector<double> best;
    double best_value = 0.0;
    std::vector<double> mcmc_objective_values;
};

// objective_fn must evaluate through the SAME ParameterManager object `pm` (its mode is switched)
inline CalibrationResult calibrate(const HCSettings& hc_cfg, MHSettings mh_cfg, const std::vector<double>& x0,
                                   const Objective& objective_fn, ParameterManager& pm, uint32_t hc_seed,
                                   uint32_t mh_seed) {
    const int P = static_cast<int>(x0.size());
    CalibrationResult r;
    r.best = x0;
    r.best_value = objective_fn(x0);  // :36-45
    r.initial_value = r.best_value;
    if (std::isnan(r.best_value) || std::isinf(r.best_value)) r.best_value = -std::numeric_limits<double>::infinity();
    pm.mode = OPTIMIZATION_CLAMP;  // :62-66
    r.phase1 = hill_climbing(hc_cfg, r.best, objective_fn, pm, hc_seed);
    if (r.phase1.best_value > r.best_value) { r.best_value = r.phase1.best_value; r.best = r.phase1.best; }
    pm.mode = MCMC_REFLECT;  // :85-89
    if (!r.phase1.final_cov.empty()) {
        r.phase2_cov = condition_phase1_covariance(r.phase1.final_cov, P, [&](int i) { return pm.sigmas.at(pm.names[i]); });
        mh_cfg.initial_cov = r.phase2_cov;
    }
    r.phase2 = metropolis_hastings(mh_cfg, r.best, objective_fn, pm, mh_seed);
    if (r.phase2.best_value > r.best_value) { r.best_value = r.phase2.best_value; r.best = r.phase2.best; }
    for (const auto& smp : r.phase2.samples) r.mcmc_objective_values.push_back(objective_fn(smp));  // :141-144
    return r;
}

// -----------------------------------------------------------------------------
// ParticleSwarmOptimization (src/model/optimizers/ParticleSwarmOptimizer.cpp:105-948) restated as the reference
// runs it with use_parallel = 0: ONE particle at a time -- draw its neighbourhood best, move it, evaluate it,
// update its personal best -- so that a later particle of the same iteration already sees the earlier ones'
// new personal bests (:377-432).  `deferred_personal_bests` holds those updates back until the iteration's loop
// has finished, which is the outcome the reference's `omp parallel for` aims at and the order the batched
// product path (one launch per iteration) necessarily has; with the GLOBAL_BEST topology no particle reads
// another's personal best inside the loop and the two orders coincide.
// The reference seeds `rng_` from std::random_device (ParticleSwarmOptimizer.hpp:228); here it takes a seed.
// One deviation, marked below: the HYBRID variant's quantum branch receives the mean personal best (the
// reference passes an empty vector there, :356-359 with :407-409, and indexes it).
// -----------------------------------------------------------------------------
struct PSOSettings {
    int iterations = 100, swarm_size = 30, max_stagnation = 50;
    double omega_start = 0.9, omega_end = 0.4, c1_initial = 2.5, c1_final = 0.5, c2_initial = 0.5, c2_final = 2.5;
    int variant = 0;   // 0 standard, 1 quantum, 2 adaptive, 3 Levy flight, 4 hybrid
    int topology = 0;  // 0 global best, 1 ring (k = 2), 2 von Neumann grid, 3 random dynamic
    bool use_opposition_learning = false, use_adaptive_parameters = false;
    double restart_threshold = 1e-6, quantum_beta = 1.0, levy_alpha = 1.5;
    bool deferred_personal_bests = false;
};

struct PSOResult {
    std::vector<double> best;
    double best_value = 0.0;
    std::vector<double> final_cov;   // P x P row-major
    std::vector<double> best_trace;  // global best after every iteration
    long evaluations = 0;
};

inline PSOResult particle_swarm(const PSOSettings& cfg, const std::vector<double>* x0, const Objective& objective_fn,
                                const ParameterManager& pm, uint32_t seed) {
    const int P = static_cast<int>(pm.names.size());
    const int S = cfg.swarm_size;
    PSOResult r;
    struct Ptl {  // ParticleSwarmOptimizer.hpp `particle`
        std::vector<double> x, v, pb_x, q_x;
        double pb = -std::numeric_limits<double>::infinity(), fit = -std::numeric_limits<double>::infinity(), rate = 0.0;
        int wins = 0, moves = 0;
    };
    std::vector<double> lb(P), ub(P);
    for (int k = 0; k < P; ++k) {
        const auto& bd = pm.bounds.at(pm.names[k]);  // getLower/UpperBoundForParamIndex throw without bounds
        lb[k] = bd.first;
        ub[k] = bd.second;
    }
    auto f = [&](const std::vector<double>& p) { ++r.evaluations; return objective_fn(p); };
    auto clip = [](double v, double lo, double hi) { return v < lo ? lo : (hi < v ? hi : v); };  // std::clamp
    std::mt19937 master(seed);
    std::uniform_real_distribution<> master_u(0.0, 1.0);  // uniform_dist_ / normal_dist_: members, state persists
    std::normal_distribution<> master_n(0.0, 1.0);
    std::vector<Ptl> sw(S);
    std::vector<double> g_x(P, 0.0);
    double g = -std::numeric_limits<double>::infinity();
    int stagnation = 0;
    auto draw_seeds = [&]() {
        std::vector<unsigned int> sd(S);
        for (int i = 0; i < S; ++i) sd[i] = static_cast<unsigned int>(master());
        return sd;
    };

    // ---- initializeSwarm :249-328
    {
        const std::vector<unsigned int> sd = draw_seeds();
        for (int i = 0; i < S; ++i) {
            Ptl& p = sw[i];
            p.x.resize(P); p.v.resize(P);
            std::mt19937 lr(sd[i]);
            std::uniform_real_distribution<> lu(0.0, 1.0);
            for (int k = 0; k < P; ++k)
                p.x[k] = (i == 0 && x0) ? clip((*x0)[k], lb[k], ub[k]) : lb[k] + lu(lr) * (ub[k] - lb[k]);
            for (int k = 0; k < P; ++k) {
                const double vmax = 0.2 * (ub[k] - lb[k]);
                p.v[k] = -vmax + 2 * vmax * lu(lr);
            }
            p.fit = f(p.x);
            p.pb_x = p.x; p.pb = p.fit; p.q_x = p.x;
        }
        if (cfg.use_opposition_learning) {  // :527-574 then :309-314
            std::vector<Ptl> mirror(S);
            for (int i = 0; i < S; ++i) {
                mirror[i].x.resize(P); mirror[i].v.resize(P);
                for (int k = 0; k < P; ++k) {
                    mirror[i].x[k] = lb[k] + ub[k] - sw[i].x[k];
                    mirror[i].v[k] = -sw[i].v[k];
                }
                mirror[i].q_x = mirror[i].x; mirror[i].pb_x = mirror[i].x;  // pb stays -inf: never evaluated
            }
            std::vector<std::pair<double, int>> order;
            for (int i = 0; i < S; ++i) { order.push_back({sw[i].pb, i}); order.push_back({mirror[i].pb, i + S}); }
            std::sort(order.begin(), order.end(), [](const auto& a, const auto& b) { return a.first > b.first; });
            std::vector<Ptl> chosen(S);
            for (int i = 0; i < S; ++i) chosen[i] = order[i].second < S ? sw[order[i].second] : mirror[order[i].second - S];
            sw = std::move(chosen);
            for (int i = 0; i < S; ++i) { sw[i].fit = f(sw[i].x); sw[i].pb = sw[i].fit; sw[i].pb_x = sw[i].x; }
        }
        for (const Ptl& p : sw) if (p.pb > g) { g = p.pb; g_x = p.pb_x; }
    }

    auto neighbours = [&](int i) {  // getNeighbors :836-906
        std::vector<int> nb;
        if (cfg.topology == 0) { for (int j = 0; j < S; ++j) nb.push_back(j); }
        else if (cfg.topology == 1) {
            nb.push_back(i);
            for (int j = 1; j <= 2; ++j) { nb.push_back((i - j + S) % S); nb.push_back((i + j) % S); }
        } else if (cfg.topology == 2) {
            const int gs = static_cast<int>(std::ceil(std::sqrt(S)));
            const int row = i / gs, col = i % gs;
            nb.push_back(i);
            if (row > 0 && (row - 1) * gs + col < S) nb.push_back((row - 1) * gs + col);
            if (row < gs - 1 && (row + 1) * gs + col < S) nb.push_back((row + 1) * gs + col);
            if (col > 0 && row * gs + col - 1 < S) nb.push_back(row * gs + col - 1);
            if (col < gs - 1 && row * gs + col + 1 < S) nb.push_back(row * gs + col + 1);
        } else {
            nb.push_back(i);
            std::vector<int> cand;
            for (int j = 0; j < S; ++j) if (j != i) cand.push_back(j);
            std::shuffle(cand.begin(), cand.end(), master);
            nb.insert(nb.end(), cand.begin(), cand.begin() + std::min(4, static_cast<int>(cand.size())));
        }
        return nb;
    };
    auto neighbourhood_best = [&](int i) {  // :816-834
        std::vector<double> bx = sw[i].pb_x;
        double bv = sw[i].pb;
        for (int j : neighbours(i))
            if (j >= 0 && j < S && sw[j].pb > bv) { bv = sw[j].pb; bx = sw[j].pb_x; }
        return bx;
    };
    auto move_standard = [&](Ptl& p, const std::vector<double>& lbest, double omega, double c1, double c2,
                             std::mt19937& lr) {  // :576-618
        std::uniform_real_distribution<> lu(0.0, 1.0);
        std::vector<double> r1(P), r2(P);
        for (int k = 0; k < P; ++k) { r1[k] = lu(lr); r2[k] = lu(lr); }
        for (int k = 0; k < P; ++k) {
            const double cognitive = c1 * (r1[k] * (p.pb_x[k] - p.x[k]));
            const double social = c2 * (r2[k] * (lbest[k] - p.x[k]));
            double v = omega * p.v[k] + cognitive + social;
            const double vmax = 0.2 * (ub[k] - lb[k]);
            v = clip(v, -vmax, vmax);
            double x = p.x[k] + v;
            if (x < lb[k]) { x = lb[k] + std::abs(x - lb[k]); v *= -0.5; }
            else if (x > ub[k]) { x = ub[k] - std::abs(x - ub[k]); v *= -0.5; }
            p.x[k] = clip(x, lb[k], ub[k]);
            p.v[k] = v;
        }
    };
    auto move_quantum = [&](Ptl& p, const std::vector<double>& mbest, int iter, std::mt19937& lr) {  // :620-653
        std::uniform_real_distribution<> lu(0.0, 1.0);
        const double phi = lu(lr);
        const double beta = cfg.quantum_beta * (1.0 - 0.5 * static_cast<double>(iter) / cfg.iterations);
        for (int k = 0; k < P; ++k) {
            const double attractor = phi * p.pb_x[k] + (1 - phi) * g_x[k];
            const double u = lu(lr);
            const double L = 2.0 * beta * std::abs(mbest[k] - p.x[k]);
            const double side = lu(lr);
            p.x[k] = clip(side < 0.5 ? attractor + L * std::log(1.0 / u) : attractor - L * std::log(1.0 / u), lb[k], ub[k]);
        }
        p.q_x = p.x;
    };
    auto move_levy = [&](Ptl& p, double omega, double c1, double c2, std::mt19937& lr) {  // :655-680, :908-934
        move_standard(p, g_x, omega, c1, c2, lr);
        std::uniform_real_distribution<> lu(0.0, 1.0);
        if (lu(lr) < 0.1 * (1.0 + p.rate)) {
            const double a = cfg.levy_alpha;
            std::vector<double> step(P);
            for (int k = 0; k < P; ++k) {  // Mantegna; a fresh normal_distribution per number
                const double sigma_u = std::pow(std::tgamma(1 + a) * std::sin(M_PI * a / 2) /
                                                    (std::tgamma((1 + a) / 2) * a * std::pow(2, (a - 1) / 2)),
                                                1.0 / a);
                std::normal_distribution<> ln(0.0, 1.0);
                const double u = ln(lr) * sigma_u;
                const double v = std::max(std::abs(ln(lr)), 1e-10);
                step[k] = clip(u / std::pow(v, 1.0 / a), -100.0, 100.0);
            }
            const double step_scale = 0.01 * (1.0 - stagnation / static_cast<double>(cfg.max_stagnation));
            for (int k = 0; k < P; ++k) p.x[k] = clip(p.x[k] + step_scale * (ub[k] - lb[k]) * step[k], lb[k], ub[k]);
        }
    };
    auto restart = [&]() {  // restartSwarm :742-814, keep_best_count = 3
        const int keep = 3;
        std::sort(sw.begin(), sw.end(), [](const Ptl& a, const Ptl& b) { return a.pb > b.pb; });
        const std::vector<Ptl> elite(sw.begin(), sw.begin() + std::min(keep, S));
        const std::vector<unsigned int> sd = draw_seeds();
        for (int i = keep; i < S; ++i) {
            std::mt19937 lr(sd[i]);
            std::uniform_real_distribution<> lu(0.0, 1.0);
            std::normal_distribution<> ln(0.0, 1.0);
            const Ptl& e = elite[static_cast<size_t>(i) % elite.size()];
            Ptl& p = sw[i];
            for (int k = 0; k < P; ++k) {
                if (lu(lr) < 0.7) {
                    const double sigma = 0.3 * (ub[k] - lb[k]) * (1.0 + 0.5 * lu(lr));
                    p.x[k] = e.x[k] + sigma * ln(lr);
                } else {
                    p.x[k] = lb[k] + lu(lr) * (ub[k] - lb[k]);
                }
                p.x[k] = clip(p.x[k], lb[k], ub[k]);
                const double vmax = 0.2 * (ub[k] - lb[k]);
                p.v[k] = -vmax + 2 * vmax * lu(lr);
            }
            p.fit = f(p.x);
            p.pb_x = p.x; p.pb = p.fit; p.q_x = p.x;
            p.wins = 0; p.moves = 0; p.rate = 0.0;
        }
        for (int i = 0; i < std::min(keep, S); ++i) sw[i] = elite[i];
        g = sw[0].pb; g_x = sw[0].pb_x;
    };

    double previous_g = -std::numeric_limits<double>::infinity();
    for (int iter = 0; iter < cfg.iterations; ++iter) {  // :127-215
        if (std::abs(g - previous_g) < cfg.restart_threshold) {
            if (++stagnation > cfg.max_stagnation) { restart(); stagnation = 0; }
        } else {
            stagnation = 0;
        }
        previous_g = g;

        // ---- updateParticles :330-432
        double omega, c1, c2;
        const double ratio = (cfg.iterations > 1) ? static_cast<double>(iter) / (cfg.iterations - 1) : 0.0;
        if (cfg.use_adaptive_parameters) {
            double mean_d = 0.0, max_d = 0.0;  // calculateEvolutionaryFactor :445-479
            for (const Ptl& p : sw) {
                double sq = 0.0;
                for (int k = 0; k < P; ++k) sq += (p.x[k] - g_x[k]) * (p.x[k] - g_x[k]);
                mean_d += std::sqrt(sq);
                max_d = std::max(max_d, std::sqrt(sq));
            }
            mean_d /= S;
            double mean_f = 0.0, max_f = -std::numeric_limits<double>::infinity(), min_f = std::numeric_limits<double>::infinity();
            for (const Ptl& p : sw) { mean_f += p.fit; max_f = std::max(max_f, p.fit); min_f = std::min(min_f, p.fit); }
            mean_f /= S;
            const double range = (max_f - min_f) > 1e-10 ? (max_f - min_f) : 1e-10;
            const double ef = 0.5 * ((max_d > 0) ? mean_d / max_d : 0.0) + 0.5 * (1.0 - (max_f - mean_f) / range);
            if (ef > 0.7) { omega = 0.9 - 0.2 * ratio; c1 = 1.5 + 0.5 * std::sin(ratio * M_PI); c2 = 1.5 - 0.5 * std::sin(ratio * M_PI); }
            else if (ef > 0.4) { omega = 0.7 - 0.3 * ratio; c1 = 2.0 - ratio; c2 = 1.0 + ratio; }
            else if (ef > 0.2) { omega = 0.4 - 0.3 * ratio; c1 = 1.0 - 0.5 * ratio; c2 = 2.0 + 0.5 * ratio; }
            else { omega = 0.9 + 0.1 * master_u(master); c1 = 2.5 + master_u(master); c2 = 0.5 + master_u(master); }
            omega = clip(omega, 0.1, 1.0); c1 = clip(c1, 0.0, 4.0); c2 = clip(c2, 0.0, 4.0);
        } else {
            omega = cfg.omega_start + (cfg.omega_end - cfg.omega_start) * ratio;
            c1 = cfg.c1_initial + (cfg.c1_final - cfg.c1_initial) * ratio;
            c2 = cfg.c2_initial + (cfg.c2_final - cfg.c2_initial) * ratio;
        }
        std::vector<double> mbest;
        if (cfg.variant == 1 || cfg.variant == 4 /* deviation: see the header comment */) {
            mbest.assign(P, 0.0);
            for (const Ptl& p : sw) for (int k = 0; k < P; ++k) mbest[k] += p.pb_x[k];
            for (double& m : mbest) m /= S;
        }
        const std::vector<unsigned int> sd = draw_seeds();
        // deferred: every neighbourhood best is read before the first particle moves.  The RANDOM_DYNAMIC
        // topology's shuffles draw from the master generator in particle order either way.
        std::vector<std::vector<double>> lbests;
        if (cfg.deferred_personal_bests)
            for (int i = 0; i < S; ++i) lbests.push_back(cfg.topology == 0 ? g_x : neighbourhood_best(i));
        for (int i = 0; i < S; ++i) {
            Ptl& p = sw[i];
            std::mt19937 lr(sd[i]);
            std::uniform_real_distribution<> lu(0.0, 1.0);
            const std::vector<double> lbest =
                cfg.deferred_personal_bests ? lbests[i] : (cfg.topology == 0 ? g_x : neighbourhood_best(i));
            if (cfg.variant == 0 || cfg.variant == 2) move_standard(p, lbest, omega, c1, c2, lr);
            else if (cfg.variant == 1) move_quantum(p, mbest, iter, lr);
            else if (cfg.variant == 3) move_levy(p, omega, c1, c2, lr);
            else {
                if (p.rate < 0.3 && lu(lr) < 0.5) move_levy(p, omega, c1, c2, lr);
                else if (p.rate > 0.7 && lu(lr) < 0.3) move_quantum(p, mbest, iter, lr);
                else move_standard(p, lbest, omega, c1, c2, lr);
            }
            const double nf = f(p.x);
            p.fit = nf;
            p.moves++;
            if (nf > p.pb) { p.pb = nf; p.pb_x = p.x; p.wins++; }
            p.rate = p.moves > 0 ? static_cast<double>(p.wins) / p.moves : 0.0;
        }
        for (const Ptl& p : sw) if (p.pb > g) { g = p.pb; g_x = p.pb_x; }  // :148-155

        if ((cfg.variant == 2 || cfg.variant == 4) && iter % 5 == 0) {  // ELS :158-179, :706-740
            auto it = std::max_element(sw.begin(), sw.end(), [](const Ptl& a, const Ptl& b) { return a.pb < b.pb; });
            Ptl& b = *it;
            std::vector<double> trial = b.x;
            double sigma_scale = 0.1 * std::exp(-2.0 * b.rate);
            for (int attempt = 0; attempt < 3; ++attempt) {
                for (int k = 0; k < P; ++k)
                    trial[k] = clip(b.x[k] + sigma_scale * (ub[k] - lb[k]) * master_n(master), lb[k], ub[k]);
                const double tf = f(trial);
                if (tf > b.pb) { b.x = trial; b.pb_x = trial; b.pb = tf; b.fit = tf; break; }
                sigma_scale *= 0.5;
            }
            if (b.pb > g) { g = b.pb; g_x = b.pb_x; }
        }
        r.best_trace.push_back(g);
    }
    r.best = g_x;
    r.best_value = g;
    std::vector<double> mean(P, 0.0);  // :221-239
    for (const Ptl& p : sw) for (int k = 0; k < P; ++k) mean[k] += p.pb_x[k];
    for (double& m : mean) m /= S;
    r.final_cov.assign(static_cast<size_t>(P) * P, 0.0);
    for (const Ptl& p : sw)
        for (int a = 0; a < P; ++a)
            for (int b = 0; b < P; ++b) r.final_cov[a * P + b] += (p.pb_x[a] - mean[a]) * (p.pb_x[b] - mean[b]);
    for (double& c : r.final_cov) c /= (S - 1);
    for (int a = 0; a < P; ++a) r.final_cov[a * P + a] += 1e-6;
    return r;
}

// -----------------------------------------------------------------------------
// NUTSSampler (src/model/optimizers/NUTSSampler.cpp:41-428, include/model/optimizers/NUTSSampler.hpp): the
// single-phase No-U-Turn sampler over the finite-difference gradient objective -- heuristic initial step size
// (:230-287), leapfrog with the gradient norm clipped at 1000 (:290-318), recursive tree doubling (Hoffman & Gelman
// 2014, algorithm 6, :321-406), U-turn test (:409-422), dual averaging inside the adaptation window (:166-183).
// Every gradient the reference evaluates is evaluated here, in its order: THREE per tree leaf (two in the leapfrog,
// one more at the leaf's end point), `gradient_calls` counts them.  Random draws in the reference's order: a fresh
// distribution object per draw site (normal for the momentum, exponential for the slice, uniform_int for the
// direction, uniform_real for the subtree choices).  The reference seeds from std::random_device (:21); here a seed.
// Eigen's norm() / dot() are restated as plain left-to-right sums (their vectorised order is not pinned).
// -----------------------------------------------------------------------------
struct NUTSSettings {
    int iterations = 2000, adaptation_window = 500, max_tree_depth = 10;
    double delta_target = 0.8;
};
struct NUTSResult {
    std::vector<std::vector<double>> samples;
    std::vector<double> sample_values, epsilon_trace;
    std::vector<int> depth_trace;
    std::vector<double> best;
    double best_value = -std::numeric_limits<double>::infinity();
    long gradient_calls = 0;
};
using GradObjective = std::function<double(const std::vector<double>&, std::vector<double>&)>;

inline NUTSResult nuts(const NUTSSettings& cfg, const std::vector<double>& theta0, const GradObjective& grad_fn,
                       const Objective& objective_fn, const ParameterManager& pm, uint32_t seed) {
    const int P = static_cast<int>(theta0.size());
    using Vec = std::vector<double>;
    NUTSResult out;
    std::mt19937 rng(seed);
    constexpr double DELTA_MAX = 1000.0, MAX_GRAD_NORM = 1000.0;
    auto dot = [&](const Vec& a, const Vec& b) { double s = 0.0; for (int i = 0; i < P; ++i) s += a[i] * b[i]; return s; };
    auto gradient = [&](const Vec& th, Vec& g) { ++out.gradient_calls; g.assign(P, 0.0); return grad_fn(th, g); };
    auto clip = [&](Vec& g) {
        const double nrm = std::sqrt(dot(g, g));
        if (nrm > MAX_GRAD_NORM) for (double& v : g) v *= MAX_GRAD_NORM / nrm;
        return nrm;
    };
    auto leapfrog = [&](Vec& th, Vec& r, double eps) {  // :290-318
        Vec g;
        gradient(th, g); clip(g);
        for (int i = 0; i < P; ++i) r[i] += 0.5 * eps * g[i];
        for (int i = 0; i < P; ++i) th[i] += eps * r[i];
        th = pm.applyConstraints(th);
        gradient(th, g); clip(g);
        for (int i = 0; i < P; ++i) r[i] += 0.5 * eps * g[i];
    };
    auto no_uturn = [&](const Vec& tm, const Vec& tp, const Vec& rm, const Vec& rp) {  // :409-422
        double dm = 0.0, dp = 0.0;
        for (int i = 0; i < P; ++i) { dm += (tp[i] - tm[i]) * rm[i]; dp += (tp[i] - tm[i]) * rp[i]; }
        return dm >= 0 && dp >= 0;
    };
    struct Tree { Vec theta_minus, theta_plus, r_minus, r_plus, theta_prime; int n_valid = 0; bool s = false; double alpha = 0.0; int n_alpha = 0; };
    std::function<void(const Vec&, const Vec&, double, int, int, double, double, Tree&)> build =
        [&](const Vec& th, const Vec& r, double log_u, int v, int j, double eps, double H0, Tree& tree) {  // :321-406
            if (j == 0) {
                Vec tp = th, rp = r, g;
                leapfrog(tp, rp, v * eps);
                const double log_p = gradient(tp, g);
                const double Hp = log_p - 0.5 * dot(rp, rp);
                tree.n_valid = (log_u <= Hp) ? 1 : 0;
                tree.s = (log_u < Hp + DELTA_MAX);
                tree.theta_minus = tree.theta_plus = tree.theta_prime = tp;
                tree.r_minus = tree.r_plus = rp;
                tree.alpha = std::min(1.0, std::exp(Hp - H0));
                tree.n_alpha = 1;
                return;
            }
            Tree left;
            build(th, r, log_u, v, j - 1, eps, H0, left);
            if (!left.s) { tree = left; return; }
            Tree right;
            if (v == -1) {
                build(left.theta_minus, left.r_minus, log_u, v, j - 1, eps, H0, right);
                tree.theta_minus = right.theta_minus; tree.r_minus = right.r_minus;
                tree.theta_plus = left.theta_plus; tree.r_plus = left.r_plus;
            } else {
                build(left.theta_plus, left.r_plus, log_u, v, j - 1, eps, H0, right);
                tree.theta_minus = left.theta_minus; tree.r_minus = left.r_minus;
                tree.theta_plus = right.theta_plus; tree.r_plus = right.r_plus;
            }
            if (right.s) {
                tree.n_valid = left.n_valid + right.n_valid;
                const double prob = tree.n_valid > 0 ? static_cast<double>(right.n_valid) / static_cast<double>(tree.n_valid) : 0.0;
                tree.theta_prime = (std::uniform_real_distribution<>(0.0, 1.0)(rng) < prob) ? right.theta_prime : left.theta_prime;
                tree.alpha = left.alpha + right.alpha;
                tree.n_alpha = left.n_alpha + right.n_alpha;
                tree.s = left.s && right.s && no_uturn(tree.theta_minus, tree.theta_plus, tree.r_minus, tree.r_plus);
            } else {
                tree.theta_prime = left.theta_prime;
                tree.n_valid = left.n_valid;
                tree.s = false;
                tree.alpha = left.alpha;
                tree.n_alpha = left.n_alpha;
            }
        };

    // ---- findReasonableEpsilon :230-287
    double epsilon;
    {
        double avg = 0.0;
        for (int i = 0; i < P; ++i) avg += pm.sigmas.at(pm.names[i]);
        avg /= P;
        epsilon = std::max(1e-6, std::min(avg * 0.1, 0.1));
        std::normal_distribution<> normal(0.0, 1.0);
        Vec r(P), g;
        for (int i = 0; i < P; ++i) r[i] = normal(rng);
        const double log_p = gradient(theta0, g);
        if (std::isfinite(log_p)) {
            const double H0 = log_p - 0.5 * dot(r, r);
            Vec tp = theta0, rp = r;
            leapfrog(tp, rp, epsilon);
            double lpp = gradient(tp, g);
            double Hp = lpp - 0.5 * dot(rp, rp);
            double accept_prob = std::exp(std::min(0.0, Hp - H0));
            for (int iter = 0; iter < 5; ++iter) {
                if (accept_prob < 0.1 && epsilon > 1e-8) epsilon *= 0.5;
                else if (accept_prob > 0.9 && epsilon < 1.0) epsilon *= 1.5;
                else break;
                tp = theta0; rp = r;
                leapfrog(tp, rp, epsilon);
                lpp = gradient(tp, g);
                if (!std::isfinite(lpp)) { epsilon *= 0.5; continue; }
                Hp = lpp - 0.5 * dot(rp, rp);
                accept_prob = std::exp(std::min(0.0, Hp - H0));
            }
        }
    }
    const double mu = std::log(10.0 * epsilon);
    double epsilon_bar = epsilon, H_bar = 0.0;
    const double gamma = 0.05, t0 = 10.0, kappa = 0.75;
    Vec theta_m = theta0;
    for (int m = 1; m <= cfg.iterations; ++m) {  // :72-214
        std::normal_distribution<> normal(0.0, 1.0);
        Vec r0(P), g;
        for (int i = 0; i < P; ++i) r0[i] = normal(rng);
        const double log_p = gradient(theta_m, g);
        clip(g);
        if (!std::isfinite(log_p)) {
            if (!out.samples.empty()) {
                out.samples.push_back(out.samples.back());
                out.sample_values.push_back(out.sample_values.back());
                out.epsilon_trace.push_back(epsilon);
                out.depth_trace.push_back(-1);
            }
            continue;
        }
        const double H0 = log_p - 0.5 * dot(r0, r0);
        const double log_u = H0 - std::exponential_distribution<>(1.0)(rng);
        Vec theta_minus = theta_m, theta_plus = theta_m, r_minus = r0, r_plus = r0, theta_next = theta_m;
        int j = 0, n = 1, n_alpha = 0;
        bool s = true;
        double alpha = 0.0;
        while (s && j < cfg.max_tree_depth) {
            const int v = (std::uniform_int_distribution<>(0, 1)(rng) * 2) - 1;
            Tree sub;
            if (v == -1) {
                build(theta_minus, r_minus, log_u, v, j, epsilon, H0, sub);
                theta_minus = sub.theta_minus; r_minus = sub.r_minus;
            } else {
                build(theta_plus, r_plus, log_u, v, j, epsilon, H0, sub);
                theta_plus = sub.theta_plus; r_plus = sub.r_plus;
            }
            if (sub.s && no_uturn(theta_minus, theta_plus, r_minus, r_plus)) {
                const double acceptance = static_cast<double>(sub.n_valid) / static_cast<double>(n + sub.n_valid);
                if (std::uniform_real_distribution<>(0.0, 1.0)(rng) < acceptance) theta_next = sub.theta_prime;
                n += sub.n_valid;
                alpha += sub.alpha;
                n_alpha += sub.n_alpha;
                j++;
            } else {
                s = false;
            }
        }
        theta_m = theta_next;
        if (m <= cfg.adaptation_window) {  // :166-183
            const double avg_alpha = n_alpha > 0 ? alpha / n_alpha : 0.0;
            const double eta = 1.0 / (m + t0);
            H_bar = (1.0 - eta) * H_bar + eta * (cfg.delta_target - avg_alpha);
            const double log_eps = mu - (std::sqrt(m) / gamma) * H_bar;
            epsilon = std::exp(log_eps);
            const double m_kappa = std::pow(m, -kappa);
            epsilon_bar = std::exp(m_kappa * log_eps + (1.0 - m_kappa) * std::log(epsilon_bar));
        } else {
            epsilon = epsilon_bar;
        }
        const Vec constrained = pm.applyConstraints(theta_m);
        out.samples.push_back(constrained);
        const double value = objective_fn(constrained);
        out.sample_values.push_back(value);
        out.epsilon_trace.push_back(epsilon);
        out.depth_trace.push_back(j);
        if (value > out.best_value) { out.best_value = value; out.best = constrained; }
    }
    return out;
}

// -----------------------------------------------------------------------------
// BASELINE config 0 ("plumbing", CPU only): the age-structured SIR model behind the same interfaces,
// AgeSIRModel::computeDerivatives (src/sir_age_structured/AgeSIRModel.cpp:106-139):
//   lambda = q (C_current (I / N)) with I/N = 0 where N <= 1e-9, clipped at 0; dS = -lambda S,
//   dI = lambda S - gamma I, dR = gamma I; a negative derivative of a compartment below 1e-9 is zeroed.
// C_current = baseline contact matrix * scale (:97-103).  State layout [S(n), I(n), R(n)].
// The reference has NO fixed-step RK4 strategy (BASELINE config 0 names one): the run goes through the
// same controlled Dopri5 / integrate_times restatement as the SEPAIHRD path.
// -----------------------------------------------------------------------------
struct SIRParams {
    int n = 0;
    std::vector<double> N, C, gamma;  // C row-major n x n (baseline), gamma per age
    double q = 0.0, scale_C = 1.0;
};

inline void sir_rhs(const SIRParams& p, const double* x, double* dx) {
    const int n = p.n;
    std::vector<double> ion(n, 0.0);
    for (int j = 0; j < n; ++j)
        if (p.N[j] > 1e-9) ion[j] = x[n + j] / p.N[j];
    for (int i = 0; i < n; ++i) {
        double acc = 0.0;
        for (int j = 0; j < n; ++j) acc += (p.C[i * n + j] * p.scale_C) * ion[j];
        double lambda = p.q * acc;
        lambda = std::max(lambda, 0.0);
        const double S = x[i], I = x[n + i], R = x[2 * n + i];
        double dS = -lambda * S, dI = lambda * S - p.gamma[i] * I, dR = p.gamma[i] * I;
        if (S < 1e-9 && dS < 0) dS = 0.0;
        if (I < 1e-9 && dI < 0) dI = 0.0;
        if (R < 1e-9 && dR < 0) dR = 0.0;
        dx[i] = dS; dx[n + i] = dI; dx[2 * n + i] = dR;
    }
}

// trajectory [T][3n] at the output times, Dopri5 via the same integrate_times restatement
inline std::vector<double> sir_simulate(const SIRParams& p, const state_type& init, const std::vector<double>& times,
                                        double abs_err, double rel_err, StepStats* stats = nullptr) {
    System sys = [&p](const double* xs, double* dx, double) { sir_rhs(p, xs, dx); };
    std::vector<double> flat;
    flat.reserve(times.size() * init.size());
    auto obs = [&flat](const state_type& x, double) { flat.insert(flat.end(), x.begin(), x.end()); };
    state_type x = init;
    ControlledDopri5 stepper(abs_err, rel_err, stats);
    integrate_times(stepper, sys, x, times, 1.0, obs, stats ? stats->max_attempts : 1000000);
    return flat;
}

}  // namespace oracle
