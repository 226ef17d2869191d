// =============================================================================
// csrc/sepaihrd_kernels.hip -- hand-written gfx950 (CDNA4) kernels for one
// SEPAIHRD objective evaluation per chain:
//     theta -> constrained model parameters -> initial state -> adaptive RK
//     (Dopri5 FSAL / Cash-Karp 5(4)) over the output grid -> daily incidence ->
//     3-stream Poisson log-likelihood.
//
// Mapping (MI355X-first):
//   * one LANE per (chain, age class); a chain is a group of LPC = pow2(n) adjacent
//     lanes, a 64-wide wavefront integrates 64/LPC chains (16 for the 4-age model);
//   * the whole 11-compartment state of an age class, all RK stage derivatives and
//     the chain's parameters live in that lane's VGPRs -- HBM is touched once at the
//     start (theta) and once at the end (log-likelihood, counters);
//   * the only cross-lane traffic is the age-contact contraction
//     lambda_i = sum_j M(i,j) pi_j (DPP quad broadcasts for n<=4, wave shuffles above),
//     the max-norm of the RK error estimate and the per-day likelihood row sum;
//   * step-size control is per chain: lanes of one chain always agree, chains of one
//     wavefront may take different numbers of steps (the wave runs until its slowest
//     chain is done);
//   * theta of the wave's chains is staged through LDS with coalesced loads, the
//     per-chain beta/kappa schedules stay in LDS.
//
// This file is compiled twice: -DSEPAIHRD_ARITH_FMA=0 -ffp-contract=off (same
// operation sequence as the CPU build of the reference, CMakeLists.txt:25-29) and
// -DSEPAIHRD_ARITH_FMA=1 -ffp-contract=fast.
//
// Reference behaviour followed (paths under /root/reference):
//   RHS                 src/model/AgeSEPAIHRDModel.cpp:101-228
//   beta(t), kappa(t)   src/model/PiecewiseConstantParameterStrategy.cpp:37-74,
//                       src/model/PieceWiseConstantNPIStrategy.cpp:86-127
//   constraints         src/model/parameters/SEPAIHRDParameterManager.cpp:302-347
//   theta -> model      src/model/parameters/SEPAIHRDParameterManager.cpp:164-287
//   objective           src/model/objectives/SEPAIHRDObjectiveFunction.cpp:62-279
//   integrator          Boost.Odeint integrate_times + controlled_runge_kutta<dopri5 |
//                       cash_karp54> as called from
//                       src/sir_age_structured/solvers/{Dopri5,CashKarp}SolverStrategy.cpp
// =============================================================================
#include <hip/hip_runtime.h>
#include <float.h>
#include <math.h>
#include <stdint.h>

#include <utility>

#include "sepaihrd_device.h"

#ifndef SEPAIHRD_ARITH_FMA
#error "compile with -DSEPAIHRD_ARITH_FMA=0 or 1"
#endif

namespace sepaihrd {
namespace {

// ----------------------------------------------------------------------------------
// cross-lane helpers: a chain is LPC adjacent lanes
// ----------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// value held by lane J of my chain group
template <int LPC, int J>
__device__ __forceinline__ double group_bcast(double v) {
    if constexpr (LPC == 1) {
        return v;
    } else if constexpr (LPC == 2) {
        return dpp_move<(J) | (J << 2) | ((2 + J) << 4) | ((2 + J) << 6)>(v);  // quad_perm
    } else if constexpr (LPC == 4) {
        return dpp_move<J * 0x55>(v);  // quad_perm:[J,J,J,J]
    } else {
        return __shfl(v, J, LPC);
    }
}

// NaN-ignoring max (std::max(init, v) with init never NaN: odeint's norm_inf)
__device__ __forceinline__ double max_keep(double m, double v) { return (m < v) ? v : m; }

template <int LPC>
__device__ __forceinline__ double group_max(double m) {
    if constexpr (LPC == 1) {
        return m;
    } else if constexpr (LPC == 2) {
        return max_keep(m, dpp_move<0xB1>(m));  // quad_perm:[1,0,3,2]
    } else if constexpr (LPC == 4) {
        m = max_keep(m, dpp_move<0xB1>(m));
        return max_keep(m, dpp_move<0x4E>(m));  // quad_perm:[2,3,0,1]
    } else {
#pragma unroll
        for (int off = LPC / 2; off > 0; off >>= 1) m = max_keep(m, __shfl_xor(m, off));
        return m;
    }
}

// any() over my chain group
template <int LPC>
__device__ __forceinline__ bool group_any(bool pred, int lane) {
    const unsigned long long b = __ballot(pred);
    if constexpr (LPC == 64) {
        return b != 0ull;
    } else {
        const unsigned long long mask = ((1ull << LPC) - 1ull) << (lane & ~(LPC - 1));
        return (b & mask) != 0ull;
    }
}

// ----------------------------------------------------------------------------------
// tableaus, written as Boost.Odeint writes them: quotients of doubles; the Dopri5 error
// weights are DIFFERENCES of rounded quotients (runge_kutta_dopri5.hpp do_step_impl).
// ----------------------------------------------------------------------------------
namespace dp {
constexpr double a2 = 1.0 / 5, a3 = 3.0 / 10, a4 = 4.0 / 5, a5 = 8.0 / 9;
constexpr double b21 = 1.0 / 5;
constexpr double b31 = 3.0 / 40, b32 = 9.0 / 40;
constexpr double b41 = 44.0 / 45, b42 = -56.0 / 15, b43 = 32.0 / 9;
constexpr double b51 = 19372.0 / 6561, b52 = -25360.0 / 2187, b53 = 64448.0 / 6561, b54 = -212.0 / 729;
constexpr double b61 = 9017.0 / 3168, b62 = -355.0 / 33, b63 = 46732.0 / 5247, b64 = 49.0 / 176,
                 b65 = -5103.0 / 18656;
constexpr double c1 = 35.0 / 384, c3 = 500.0 / 1113, c4 = 125.0 / 192, c5 = -2187.0 / 6784, c6 = 11.0 / 84;
constexpr double dc1 = c1 - 5179.0 / 57600, dc3 = c3 - 7571.0 / 16695, dc4 = c4 - 393.0 / 640,
                 dc5 = c5 - (-92097.0 / 339200), dc6 = c6 - 187.0 / 2100, dc7 = -1.0 / 40;
}  // namespace dp
namespace ck {
constexpr double c2 = 1.0 / 5, c3 = 3.0 / 10, c4 = 3.0 / 5, c5 = 1.0, c6 = 7.0 / 8;
constexpr double a21 = 1.0 / 5;
constexpr double a31 = 3.0 / 40, a32 = 9.0 / 40;
constexpr double a41 = 3.0 / 10, a42 = -9.0 / 10, a43 = 6.0 / 5;
constexpr double a51 = -11.0 / 54, a52 = 5.0 / 2, a53 = -70.0 / 27, a54 = 35.0 / 27;
constexpr double a61 = 1631.0 / 55296, a62 = 175.0 / 512, a63 = 575.0 / 13824, a64 = 44275.0 / 110592,
                 a65 = 253.0 / 4096;
constexpr double b1 = 37.0 / 378, b3 = 250.0 / 621, b4 = 125.0 / 594, b6 = 512.0 / 1771;
constexpr double db1 = 37.0 / 378 - 2825.0 / 27648, db3 = 250.0 / 621 - 18575.0 / 48384,
                 db4 = 125.0 / 594 - 13525.0 / 55296, db5 = -277.0 / 14336, db6 = 512.0 / 1771 - 1.0 / 4;
}  // namespace ck

// ----------------------------------------------------------------------------------
// per-lane model record
// ----------------------------------------------------------------------------------
template <int LPC>
struct LaneModel {
    double theta, sigma, gamma_p, gamma_A, gamma_I, gamma_H, gamma_ICU;
    double a, h_infec, p, h, icu, d_H, d_ICU, d_comm, inv_N;
    double Mrow[LPC];
};

// AgeSEPAIHRDModel::computeDerivatives for this lane's age class.
// beta_eff = beta(t) * kappa(t) of the chain.
template <int LPC>
__device__ __forceinline__ void rhs(const LaneModel<LPC>& q, const double (&x)[NUM_COMP],
                                    double (&dx)[NUM_COMP], double beta_eff) {
    const double S = x[0], E = x[1], P = x[2], A = x[3], I = x[4], H = x[5], ICU = x[6];
    const double total_inf = P + A + q.theta * I;
    const double inf_pressure = total_inf * q.h_infec * q.inv_N;
    double lambda = 0.0;
    // lambda_i += M(i,j) * pi_j, j ascending (column-major walk of the reference)
    [&]<int... J>(std::integer_sequence<int, J...>) {
        ((lambda += q.Mrow[J] * group_bcast<LPC, J>(inf_pressure)), ...);
    }(std::make_integer_sequence<int, LPC>{});
    lambda *= beta_eff * q.a;
    const double lambda_val = (0.0 < lambda) ? lambda : 0.0;  // std::max(0.0, lambda)

    const double flow_SE = lambda_val * S;
    const double flow_EP = q.sigma * E;
    const double flow_P_out = q.gamma_p * P;
    const double flow_PA = q.p * flow_P_out;
    const double flow_PI = flow_P_out - flow_PA;
    const double flow_IH = q.h * I;
    const double flow_IR = q.gamma_I * I;
    const double flow_ID_community = q.d_comm * I;
    const double I_out = flow_IR + flow_IH + flow_ID_community;
    const double flow_H_ICU = q.icu * H;
    const double H_out = q.gamma_H * H + q.d_H * H + flow_H_ICU;
    const double ICU_out = (q.gamma_ICU + q.d_ICU) * ICU;

    dx[0] = -flow_SE;
    dx[1] = flow_SE - flow_EP;
    dx[2] = flow_EP - flow_P_out;
    dx[3] = flow_PA - q.gamma_A * A;
    dx[4] = flow_PI - I_out;
    dx[5] = flow_IH - H_out;
    dx[6] = flow_H_ICU - ICU_out;
    dx[7] = q.gamma_A * A + flow_IR + q.gamma_H * H + q.gamma_ICU * ICU;
    dx[8] = q.d_H * H + q.d_ICU * ICU + flow_ID_community;
    dx[9] = flow_IH;
    dx[10] = flow_H_ICU;
}

// SEPAIHRDParameterManager.cpp:302-313 / :326-343
__device__ __forceinline__ double reflect_bound(double value, double minb, double maxb) {
    if (minb >= maxb) return minb;
    const double width = maxb - minb;
    double y = fmod(value - minb, 2.0 * width);
    if (y < 0) y += 2.0 * width;
    if (y <= width) return minb + y;
    return maxb - (y - width);
}
__device__ __forceinline__ double constrain(double v, double lo, double hi, int has_bounds, int mode) {
    if (has_bounds) {
        if (lo > hi) { const double t = lo; lo = hi; hi = t; }
        if (mode == 0) {
            const double m = (v < lo) ? lo : v;  // std::max(v, lo)
            return (hi < m) ? hi : m;            // std::min(m, hi)
        }
        return reflect_bound(v, lo, hi);
    }
    if (mode == 0) return (0.0 < v) ? v : 0.0;  // std::max(0.0, v)
    return fabs(v);
}

// piecewise-constant lookup: value index = #(ends < t), clamped to the last period
__device__ __forceinline__ int period_index(const double* __restrict__ ends, int count, double t) {
    int idx = 0;
    for (int k = 0; k < count; ++k) idx += (t > ends[k]) ? 1 : 0;
    return idx < count - 1 ? idx : count - 1;
}

struct Schedule {
    const double* sb;  // LDS: this chain's beta values  [nb]
    const double* sk;  // LDS: this chain's kappa values [nk]
    double beta_const; // used when nb == 0
    // cached segment (lo, hi] on which beta*kappa is constant
    double lo, hi, bk;
};

__device__ __forceinline__ double beta_kappa_at(const DevProblem& pb, const Schedule& s, double t) {
    const double beta = (pb.nb > 0) ? s.sb[period_index(pb.beta_ends, pb.nb, t)] : s.beta_const;
    const double kappa = s.sk[period_index(pb.kappa_ends, pb.nk, t)];
    return beta * kappa;
}

__device__ __forceinline__ void refresh_segment(const DevProblem& pb, Schedule& s, double t) {
    double lo = -INFINITY, hi = INFINITY;
    double beta = s.beta_const;
    if (pb.nb > 0) {
        const int ib = period_index(pb.beta_ends, pb.nb, t);
        beta = s.sb[ib];
        if (ib > 0) lo = pb.beta_ends[ib - 1];
        if (ib < pb.nb - 1) hi = pb.beta_ends[ib];
    }
    const int ik = period_index(pb.kappa_ends, pb.nk, t);
    if (ik > 0) lo = fmax(lo, pb.kappa_ends[ik - 1]);
    if (ik < pb.nk - 1) hi = fmin(hi, pb.kappa_ends[ik]);
    s.lo = lo;
    s.hi = hi;
    s.bk = beta * s.sk[ik];
}

#define SEP_UNROLL _Pragma("unroll")

// ----------------------------------------------------------------------------------
// the evaluation kernel: block = one wavefront = 64/LPC chains
// ----------------------------------------------------------------------------------
template <int LPC, int SOLVER, int ARITH_FMA>
__global__ __launch_bounds__(WAVE) void sepaihrd_eval_kernel(const DevProblem pb,
                                                              const double* __restrict__ theta,
                                                              const int B, const EvalOutputs out) {
    constexpr int CPW = WAVE / LPC;
    extern __shared__ double lds[];
    const int lane = threadIdx.x;
    const int grp = lane / LPC;
    const int age = lane % LPC;
    const long long chain0 = (long long)blockIdx.x * CPW;
    const int chains_here = (B - chain0) < CPW ? (int)(B - chain0) : CPW;
    const bool chain_valid = grp < chains_here;
    // lanes of a group past the end of the batch shadow group 0 (always valid): they follow
    // the same control flow and never store.
    const int g = chain_valid ? grp : 0;
    const long long chain = chain0 + g;
    const int P = pb.P;

    // ---- 1. coalesced load of theta for the wave's chains, constrained on the way into LDS
    //         (applyConstraints inside updateModelParameters, PM.cpp:173)
    {
        const int total = chains_here * P;
        const double* src = theta + chain0 * P;
        for (int idx = lane; idx < total; idx += WAVE) {
            const int p = idx % P;
            lds[idx] = constrain(src[idx], pb.lower[p], pb.upper[p], pb.has_bounds[p], pb.constraint_mode);
        }
    }
    __syncthreads();
    const double* th = lds + g * P;
    double* sched = lds + CPW * P + grp * (pb.nb + pb.nk);  // own region even for shadow groups

    auto scalar_slot = [&](int slot) -> double {
        const int s = pb.src_scalar[slot];
        return s >= 0 ? th[s] : pb.base_scalar[slot];
    };
    auto vec_slot = [&](int field) -> double {
        const int s = pb.src_vec[field * LPC + age];
        return s >= 0 ? th[s] : pb.base_vec[field * LPC + age];
    };

    // ---- 2. theta -> model (updateModelParameters + setModelParameters)
    LaneModel<LPC> q;
    q.theta = scalar_slot(SS_THETA);
    q.sigma = scalar_slot(SS_SIGMA);
    q.gamma_p = scalar_slot(SS_GAMMA_P);
    q.gamma_A = scalar_slot(SS_GAMMA_A);
    q.gamma_I = scalar_slot(SS_GAMMA_I);
    q.gamma_H = scalar_slot(SS_GAMMA_H);
    q.gamma_ICU = scalar_slot(SS_GAMMA_ICU);
    q.a = vec_slot(VF_A);
    q.h_infec = vec_slot(VF_H_INFEC);
    q.p = vec_slot(VF_P);
    q.h = vec_slot(VF_H);
    q.icu = vec_slot(VF_ICU);
    q.d_H = vec_slot(VF_D_H);
    q.d_ICU = vec_slot(VF_D_ICU);
    q.d_comm = vec_slot(VF_D_COMM);
    const double Ni = pb.N[age];
    q.inv_N = (Ni > 1e-9) ? (1.0 / Ni) : 0.0;  // AgeSEPAIHRDModel.cpp:332-334
    SEP_UNROLL
    for (int j = 0; j < LPC; ++j) q.Mrow[j] = pb.Mrow[age * LPC + j];

    Schedule sch;
    sch.sb = sched;
    sch.sk = sched + pb.nb;
    sch.beta_const = scalar_slot(SS_BETA);
    for (int k = age; k < pb.nb + pb.nk; k += LPC) sched[k] = scalar_slot(SS_SCHEDULE0 + k);
    __syncthreads();

    int status = 0;
    if (pb.kappa_calibrated) {  // setCalibratableValues: any after-baseline kappa < 0 -> throw -> lowest()
        bool neg = false;
        for (int k = 1; k < pb.nk; ++k) neg |= (sch.sk[k] < 0.0);
        if (neg) status = 1;
    }
    if (!pb.obs_rows_match) status = 1;  // SEPAIHRDObjectiveFunction.cpp:176-178

    // ---- 3. initial state (SEPAIHRDObjectiveFunction.cpp:124-163)
    double x[NUM_COMP];
    SEP_UNROLL
    for (int c = 0; c < NUM_COMP; ++c) x[c] = pb.init_state[c * LPC + age];
    {
        const double runup_days = scalar_slot(SS_RUNUP_DAYS);
        const double seed_exposed = scalar_slot(SS_SEED_EXPOSED);
        if (runup_days > 0 && seed_exposed > 0) {
            x[1] = seed_exposed * pb.age_fraction[age];
            SEP_UNROLL
            for (int c = 2; c < NUM_COMP; ++c) x[c] = 0.0;
        } else {
            SEP_UNROLL
            for (int c = 1; c <= 8; ++c) x[c] *= scalar_slot(SS_E0_MULT + (c - 1));
        }
        double sum = 0;
        SEP_UNROLL
        for (int c = 1; c < NUM_POP_COMP; ++c) sum += x[c];
        if (group_any<LPC>(sum > Ni, lane)) status = 1;
        x[0] = Ni - sum;
    }

    // incidence bookkeeping: previous observed values of D, CumH, CumICU (row 0 vs init_state)
    double prevD = x[8], prevH = x[9], prevICU = x[10];
    double llH = 0.0, llICU = 0.0, llD = 0.0;
    int n_acc = 0, n_rej = 0;
    const int T = pb.T;
    const int n_real = pb.n;

    auto observe = [&](int k) {
        double incH = x[9] - prevH, incICU = x[10] - prevICU, incD = x[8] - prevD;
        incH = (incH < 0.0) ? 0.0 : incH;  // cwiseMax(0.0)
        incICU = (incICU < 0.0) ? 0.0 : incICU;
        incD = (incD < 0.0) ? 0.0 : incD;
        prevH = x[9]; prevICU = x[10]; prevD = x[8];
        if (k >= pb.runup_offset) {
            const int row = k - pb.runup_offset;
            const double* o = pb.obs + (size_t)row * LPC + age;
            const size_t stream_stride = (size_t)pb.n_obs * LPC;
            const double oH = o[0], oI = o[stream_stride], oD = o[2 * stream_stride];
            const double eps = 1e-10;
            auto term = [&](double obs, double sim) -> double {
                if (obs >= 0.0 && isfinite(obs)) {
                    if (sim < 0.0) sim = 0.0;
                    sim += eps;
                    return obs * log(sim) - sim;
                }
                return 0.0;
            };
            const double tH = term(oH, incH), tI = term(oI, incICU), tD = term(oD, incD);
            // row_sum over ages in ascending order (serial order of calculateSingleLogLikelihood)
            auto row_sum = [&](double tv) -> double {
                double rs = 0.0;
                [&]<int... J>(std::integer_sequence<int, J...>) {
                    ((rs += group_bcast<LPC, J>(tv)), ...);
                }(std::make_integer_sequence<int, LPC>{});
                return rs;
            };
            llH += row_sum(tH);
            llICU += row_sum(tI);
            llD += row_sum(tD);
        }
        if (out.traj != nullptr && chain_valid && age < n_real) {
            double* dst = out.traj + ((size_t)chain * T + k) * (NUM_COMP * n_real) + age;
            SEP_UNROLL
            for (int c = 0; c < NUM_COMP; ++c) dst[c * n_real] = x[c];
        }
    };

    // ---- 4. integrate_times(controlled stepper, ..., times, dt_hint, observer)
    bool active = (status == 0) && (T > 0);
    int k_next = 1;  // index of the next output time to reach
    double t = (T > 0) ? pb.times[0] : 0.0;
    double t_next = (T > 1) ? pb.times[1] : t;
    double dt = pb.dt_hint;
    int fails = 0;
    int attempts = 0;
    if (active) observe(0);
    if (T <= 1) active = false;

    sch.lo = INFINITY; sch.hi = -INFINITY; sch.bk = 0.0;  // empty segment: first step refreshes
    double k1[NUM_COMP];
    if (SOLVER == 0) {  // controlled FSAL stepper: initialize() at the first try_step
        const double bk0 = beta_kappa_at(pb, sch, t);
        rhs<LPC>(q, x, k1, bk0);
    }

    const double eps_abs = pb.abs_tol, eps_rel = pb.rel_tol;

    while (__ballot(active) != 0ull) {
        // min_abs(dt, t_next - t); finished chains idle with a harmless unit step
        const double cur = active ? fmin(dt, t_next - t) : 1.0;

        // stage times and beta*kappa at those times
        double tau[7], bks[7];
        if (SOLVER == 0) {
            tau[0] = t;  // unused (k1 is the FSAL derivative)
            tau[1] = t + cur * dp::a2; tau[2] = t + cur * dp::a3; tau[3] = t + cur * dp::a4;
            tau[4] = t + cur * dp::a5; tau[5] = t + cur; tau[6] = t + cur;
        } else {
            tau[0] = t;
            tau[1] = t + ck::c2 * cur; tau[2] = t + ck::c3 * cur; tau[3] = t + ck::c4 * cur;
            tau[4] = t + ck::c5 * cur; tau[5] = t + ck::c6 * cur; tau[6] = t + cur;
        }
        {
            const double tmin = (SOLVER == 0) ? tau[1] : tau[0];
            const double tmax = (SOLVER == 0) ? tau[6] : tau[4];
            const bool in_seg = (tmin > sch.lo) && (tmax <= sch.hi);
            if (__ballot(active && !in_seg) != 0ull) {
                SEP_UNROLL
                for (int s = 0; s < 7; ++s) bks[s] = beta_kappa_at(pb, sch, tau[s]);
                refresh_segment(pb, sch, tau[6]);
            } else {
                SEP_UNROLL
                for (int s = 0; s < 7; ++s) bks[s] = sch.bk;
            }
        }

        double k2[NUM_COMP], k3[NUM_COMP], k4[NUM_COMP], k5[NUM_COMP], k6[NUM_COMP];
        double xt[NUM_COMP], xnew[NUM_COMP], xerr[NUM_COMP];
        double k7[NUM_COMP];

        if (SOLVER == 0) {
            // runge_kutta_dopri5::do_step_impl -- scale_sumN left to right, factors dt*b
            { const double f1 = cur * dp::b21;
              SEP_UNROLL for (int c = 0; c < NUM_COMP; ++c) xt[c] = x[c] + f1 * k1[c];
              rhs<LPC>(q, xt, k2, bks[1]); }
            { const double f1 = cur * dp::b31, f2 = cur * dp::b32;
              SEP_UNROLL for (int c = 0; c < NUM_COMP; ++c) xt[c] = x[c] + f1 * k1[c] + f2 * k2[c];
              rhs<LPC>(q, xt, k3, bks[2]); }
            { const double f1 = cur * dp::b41, f2 = cur * dp::b42, f3 = cur * dp::b43;
              SEP_UNROLL for (int c = 0; c < NUM_COMP; ++c) xt[c] = x[c] + f1 * k1[c] + f2 * k2[c] + f3 * k3[c];
              rhs<LPC>(q, xt, k4, bks[3]); }
            { const double f1 = cur * dp::b51, f2 = cur * dp::b52, f3 = cur * dp::b53, f4 = cur * dp::b54;
              SEP_UNROLL for (int c = 0; c < NUM_COMP; ++c)
                  xt[c] = x[c] + f1 * k1[c] + f2 * k2[c] + f3 * k3[c] + f4 * k4[c];
              rhs<LPC>(q, xt, k5, bks[4]); }
            { const double f1 = cur * dp::b61, f2 = cur * dp::b62, f3 = cur * dp::b63, f4 = cur * dp::b64,
                           f5 = cur * dp::b65;
              SEP_UNROLL for (int c = 0; c < NUM_COMP; ++c)
                  xt[c] = x[c] + f1 * k1[c] + f2 * k2[c] + f3 * k3[c] + f4 * k4[c] + f5 * k5[c];
              rhs<LPC>(q, xt, k6, bks[5]); }
            { const double f1 = cur * dp::c1, f3 = cur * dp::c3, f4 = cur * dp::c4, f5 = cur * dp::c5,
                           f6 = cur * dp::c6;
              SEP_UNROLL for (int c = 0; c < NUM_COMP; ++c)
                  xnew[c] = x[c] + f1 * k1[c] + f3 * k3[c] + f4 * k4[c] + f5 * k5[c] + f6 * k6[c];
              rhs<LPC>(q, xnew, k7, bks[6]); }
            { const double e1 = cur * dp::dc1, e3 = cur * dp::dc3, e4 = cur * dp::dc4, e5 = cur * dp::dc5,
                           e6 = cur * dp::dc6, e7 = cur * dp::dc7;
              SEP_UNROLL for (int c = 0; c < NUM_COMP; ++c)
                  xerr[c] = e1 * k1[c] + e3 * k3[c] + e4 * k4[c] + e5 * k5[c] + e6 * k6[c] + e7 * k7[c]; }
        } else {
            // controlled_runge_kutta<cash_karp54>::try_step: sys(x, dxdt, t) at EVERY attempt
            rhs<LPC>(q, x, k1, bks[0]);
            { const double f1 = ck::a21 * cur;
              SEP_UNROLL for (int c = 0; c < NUM_COMP; ++c) xt[c] = x[c] + f1 * k1[c];
              rhs<LPC>(q, xt, k2, bks[1]); }
            { const double f1 = ck::a31 * cur, f2 = ck::a32 * cur;
              SEP_UNROLL for (int c = 0; c < NUM_COMP; ++c) xt[c] = x[c] + f1 * k1[c] + f2 * k2[c];
              rhs<LPC>(q, xt, k3, bks[2]); }
            { const double f1 = ck::a41 * cur, f2 = ck::a42 * cur, f3 = ck::a43 * cur;
              SEP_UNROLL for (int c = 0; c < NUM_COMP; ++c) xt[c] = x[c] + f1 * k1[c] + f2 * k2[c] + f3 * k3[c];
              rhs<LPC>(q, xt, k4, bks[3]); }
            { const double f1 = ck::a51 * cur, f2 = ck::a52 * cur, f3 = ck::a53 * cur, f4 = ck::a54 * cur;
              SEP_UNROLL for (int c = 0; c < NUM_COMP; ++c)
                  xt[c] = x[c] + f1 * k1[c] + f2 * k2[c] + f3 * k3[c] + f4 * k4[c];
              rhs<LPC>(q, xt, k5, bks[4]); }
            { const double f1 = ck::a61 * cur, f2 = ck::a62 * cur, f3 = ck::a63 * cur, f4 = ck::a64 * cur,
                           f5 = ck::a65 * cur;
              SEP_UNROLL for (int c = 0; c < NUM_COMP; ++c)
                  xt[c] = x[c] + f1 * k1[c] + f2 * k2[c] + f3 * k3[c] + f4 * k4[c] + f5 * k5[c];
              rhs<LPC>(q, xt, k6, bks[5]); }
            // zero tableau entries (b2 = b5 = 0, db2 = 0) contribute an exact +0.0 in the reference
            { const double f1 = ck::b1 * cur, f3 = ck::b3 * cur, f4 = ck::b4 * cur, f6 = ck::b6 * cur;
              SEP_UNROLL for (int c = 0; c < NUM_COMP; ++c)
                  xnew[c] = x[c] + f1 * k1[c] + f3 * k3[c] + f4 * k4[c] + f6 * k6[c]; }
            { const double e1 = ck::db1 * cur, e3 = ck::db3 * cur, e4 = ck::db4 * cur, e5 = ck::db5 * cur,
                           e6 = ck::db6 * cur;
              SEP_UNROLL for (int c = 0; c < NUM_COMP; ++c)
                  xerr[c] = e1 * k1[c] + e3 * k3[c] + e4 * k4[c] + e5 * k5[c] + e6 * k6[c]; }
        }

        // default_error_checker: max_i |xerr_i| / (eps_abs + eps_rel (|x_i| + dt |dxdt_i|)), start-of-step x, dxdt
        double err = 0.0;
        SEP_UNROLL
        for (int c = 0; c < NUM_COMP; ++c) {
            const double e = fabs(xerr[c]) / (eps_abs + eps_rel * (fabs(x[c]) + cur * fabs(k1[c])));
            err = max_keep(err, e);
        }
        err = group_max<LPC>(err);

        const bool reject = err > 1.0;
        ++attempts;
        // default_step_adjuster: powers are only evaluated where some lane needs them
        const bool need_dec = active && reject;
        // growth can only matter while dt is below the largest output gap (dt = max(dt, grown))
        const bool need_inc = active && !reject && (err < 0.5) && (dt < pb.max_gap);
        double cur_after = cur;
        if (__ballot(need_dec) != 0ull) {
            const double f = fmax(9.0 / 10.0 * pow(err, -1.0 / (4 - 1)), 1.0 / 5.0);
            if (need_dec) cur_after = cur * f;
        }
        if (__ballot(need_inc) != 0ull) {
            const double e2 = fmax(pow(5.0, -5.0), err);
            const double f = 9.0 / 10.0 * pow(e2, -1.0 / 5);
            if (need_inc) cur_after = cur * f;
        }

        if (active) {
            if (reject) {
                ++n_rej;
                dt = cur_after;  // dt = current_dt (reduced)
                if (fails++ >= 500) { status = 2; active = false; }
            } else {
                ++n_acc;
                fails = 0;
                t += cur;
                SEP_UNROLL
                for (int c = 0; c < NUM_COMP; ++c) x[c] = xnew[c];
                if (SOLVER == 0) {
                    SEP_UNROLL
                    for (int c = 0; c < NUM_COMP; ++c) k1[c] = k7[c];
                }
                dt = fmax(dt, cur_after);  // max_abs(dt, current_dt)
                // less_with_sign(t, t_next, dt): t_next - t > epsilon
                if (!((t_next - t) > DBL_EPSILON)) {
                    t = t_next;  // integrate_times re-reads the exact grid time
                    observe(k_next);
                    ++k_next;
                    if (k_next >= T) active = false;
                    else t_next = pb.times[k_next];
                }
            }
            if (active && attempts >= pb.max_attempts) { status = 3; active = false; }
        }
    }

    // ---- 5. total (SEPAIHRDObjectiveFunction.cpp:222-227)
    if (chain_valid && age == 0) {
        double total = llH + llICU + llD;
        if (status == 0 && (isnan(total) || isinf(total))) status = 1;
        if (status != 0) total = -DBL_MAX;
        out.loglik[chain] = total;
        if (out.status) out.status[chain] = status;
        if (out.n_accept) out.n_accept[chain] = n_acc;
        if (out.n_reject) out.n_reject[chain] = n_rej;
        if (out.ll_parts) {
            out.ll_parts[3 * chain + 0] = llH;
            out.ll_parts[3 * chain + 1] = llICU;
            out.ll_parts[3 * chain + 2] = llD;
        }
    }
}

// ----------------------------------------------------------------------------------
// launch plumbing
// ----------------------------------------------------------------------------------
template <int LPC, int SOLVER>
int launch_one(const DevProblem& pb, const double* d_theta, int B, const EvalOutputs& out, void* stream) {
    constexpr int CPW = WAVE / LPC;
    const int blocks = (B + CPW - 1) / CPW;
    if (blocks <= 0) return 0;
    const size_t lds = eval_lds_bytes(pb);
    hipLaunchKernelGGL((sepaihrd_eval_kernel<LPC, SOLVER, SEPAIHRD_ARITH_FMA>), dim3(blocks), dim3(WAVE), lds,
                       static_cast<hipStream_t>(stream), pb, d_theta, B, out);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

template <int LPC, int SOLVER>
int info_one(const DevProblem& pb, LaunchInfo* info, const char* name) {
    hipFuncAttributes attr;
    if (hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(&sepaihrd_eval_kernel<LPC, SOLVER, SEPAIHRD_ARITH_FMA>)) !=
        hipSuccess)
        return -3;
    info->vgprs = attr.numRegs;
    info->sgprs = 0;
    info->lds_static = (int)attr.sharedSizeBytes;
    info->scratch = (int)attr.localSizeBytes;
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, sepaihrd_eval_kernel<LPC, SOLVER, SEPAIHRD_ARITH_FMA>, WAVE,
                                                     eval_lds_bytes(pb)) != hipSuccess)
        nb = -1;
    info->max_blocks_per_cu = nb;
    info->name = name;
    return 0;
}

#define SEP_DISPATCH(FN, ...)                                                                  \
    switch (pb.lpc) {                                                                          \
        case 1: return solver == 0 ? FN<1, 0>(__VA_ARGS__) : FN<1, 1>(__VA_ARGS__);            \
        case 2: return solver == 0 ? FN<2, 0>(__VA_ARGS__) : FN<2, 1>(__VA_ARGS__);            \
        case 4: return solver == 0 ? FN<4, 0>(__VA_ARGS__) : FN<4, 1>(__VA_ARGS__);            \
        case 8: return solver == 0 ? FN<8, 0>(__VA_ARGS__) : FN<8, 1>(__VA_ARGS__);            \
        case 16: return solver == 0 ? FN<16, 0>(__VA_ARGS__) : FN<16, 1>(__VA_ARGS__);         \
        default: return -4;                                                                    \
    }

}  // namespace

#if SEPAIHRD_ARITH_FMA
#define SEP_LAUNCH launch_eval_fma
#define SEP_INFO kernel_info_fma
#define SEP_NAME "sepaihrd_eval_kernel[fma]"
#else
#define SEP_LAUNCH launch_eval_strict
#define SEP_INFO kernel_info_strict
#define SEP_NAME "sepaihrd_eval_kernel[strict]"
#endif

int SEP_LAUNCH(const DevProblem& pb, int solver, const double* d_theta, int B, const EvalOutputs& out,
               void* stream) {
    SEP_DISPATCH(launch_one, pb, d_theta, B, out, stream)
}
int SEP_INFO(const DevProblem& pb, int solver, LaunchInfo* info) {
    SEP_DISPATCH(info_one, pb, info, SEP_NAME)
}

}  // namespace sepaihrd
