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
//     lambda_i = sum_j M(i,j) pi_j (DPP quad_perm broadcasts for n <= 4; for n = 8, 16 DPP row_newbcast, in the
//     tolerance build fused into the multiply-add as v_fmac_f64_dpp), the max-norm of the RK error estimate and
//     the per-day likelihood row sum;
//   * batches of up to 4096 chains of a problem with n <= 4 run a second form of the integrator with SIXTEEN lanes
//     per chain (sepaihrd_lane_split.inc: the compartments of an age class spread over a quad of lanes), so that a
//     batch too small to fill the chip still puts a wave on every SIMD; bit-identical results;
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
#include <stdlib.h>

#include <utility>

#include "sepaihrd_device.h"

#ifndef SEPAIHRD_DOPRI5_WPS2
#define SEPAIHRD_DOPRI5_WPS2 0
#endif
#ifndef SEPAIHRD_LL_STATE_IN_LDS
#define SEPAIHRD_LL_STATE_IN_LDS 1  // the inline likelihood's loop-carried state in LDS in the two-waves-per-SIMD builds (see LL_IN_LDS)
#endif
#ifndef SEPAIHRD_ARITH_FMA
#error "compile with -DSEPAIHRD_ARITH_FMA=0 or 1"
#endif

namespace sepaihrd {
namespace {

#include "sepaihrd_dev_common.inc"  // cross-lane helpers, tableaus, constraints, schedule lookup, log_pos

// ----------------------------------------------------------------------------------
// per-lane model record
// ----------------------------------------------------------------------------------
#if SEPAIHRD_ARITH_FMA
// tolerance mode folds per-chain constants once per evaluation instead of once per RHS call:
//   the contact row := (a_i M(i, j)) (h_infec_j / N_j) (MF_H_INFEC := h_infec / N is only its ingredient), MF_R_I := gamma_I + h + d_community,
//   MF_R_H := gamma_H + d_H + icu, MF_R_ICU := gamma_ICU + d_ICU  (8 instructions fewer per call)
//   MF_PG := p gamma_p, MF_PI := gamma_p - p gamma_p.
// The derivatives are written with explicit fma() in ONE fixed association, the one the 16-lane small-batch form
// (sepaihrd_lane_split.inc) evaluates, so that a chain's result does not depend on which of the two kernels -- i.e.
// on which batch size -- evaluated it.
enum ModelField { MF_THETA = 0, MF_SIGMA, MF_GAMMA_P, MF_GAMMA_A, MF_GAMMA_I, MF_GAMMA_H, MF_GAMMA_ICU, MF_A,
                  MF_H_INFEC, MF_P, MF_H, MF_ICU, MF_D_H, MF_D_ICU, MF_D_COMM, MF_INV_N, MF_R_I, MF_R_H, MF_R_ICU,
                  MF_PG, MF_PI, MF_MROW0 };
#else
enum ModelField { MF_THETA = 0, MF_SIGMA, MF_GAMMA_P, MF_GAMMA_A, MF_GAMMA_I, MF_GAMMA_H, MF_GAMMA_ICU, MF_A,
                  MF_H_INFEC, MF_P, MF_H, MF_ICU, MF_D_H, MF_D_ICU, MF_D_COMM, MF_INV_N, MF_MROW0 };
#endif
template <int LPC>
struct LaneModel {  // VGPR-resident (an LDS-resident variant measured no gain at 1 wave/SIMD and lost at 2)
    double v[MF_MROW0 + LPC];
    __device__ __forceinline__ double get(int f) const { return v[f]; }
    __device__ __forceinline__ void set(int f, double val) { v[f] = val; }
};

// AgeSEPAIHRDModel::computeDerivatives for this lane's age class.
// beta_eff = beta(t) * kappa(t) of the chain.
// Branch-probability hints of the 4-lane / 16-age time loop: they make the common path of an attempt fall through (a taken
// branch costs a lone wave 12-15 cycles).  Tolerance build only: measured in the strict build the same hints cost 1.5-5 %
// (c3 3.31 -> 3.44 ms, c5 28.9 -> 30.5 ms) -- its code is laid out well as it is.
#if SEPAIHRD_ARITH_FMA
#define SEP_RARELY(x) __builtin_expect((x), 0)
#define SEP_MOSTLY(x) __builtin_expect((x), 1)
#else
#define SEP_RARELY(x) (x)
#define SEP_MOSTLY(x) (x)
#endif

template <int LPC>
__device__ __forceinline__ void rhs(const LaneModel<LPC>& q, const double (&x)[NUM_COMP],
                                    double (&dx)[NUM_COMP], double beta_eff) {
    const double S = x[0], E = x[1], P = x[2], A = x[3], I = x[4], H = x[5], ICU = x[6];
#if SEPAIHRD_ARITH_FMA
    const double total_inf = fma(q.get(MF_THETA), I, P + A);
    const double inf_pressure = total_inf;  // h_infec_j / N_j rides in the contact row (column j) since round 2
#else
    const double total_inf = P + A + q.get(MF_THETA) * I;
    const double inf_pressure = total_inf * q.get(MF_H_INFEC) * q.get(MF_INV_N);
#endif
    // lambda_i = 0.0 + M(i,0) pi_0 + M(i,1) pi_1 + ..., j ascending (column-major walk of the reference).
    // The leading "0.0 +" only turns a -0.0 first product into +0.0, which max(0.0, .) below does anyway.
#if SEPAIHRD_ARITH_FMA
    // fma(m_0, pi_0, +0.0) = round(m_0 pi_0), then fma(m_j, pi_j, .) for j = 1, 2, ...: written out, because left to
    // itself the contraction pass fuses the FIRST product of "m0 pi0 + m1 pi1" and rounds the second
    double lambda;
    if constexpr (LPC == 8 || LPC == 16) {
        lambda = 0.0;
        [&]<int... J>(std::integer_sequence<int, J...>) {
            (([&] {
                if constexpr (LPC == 16) {
                    fmac_row_bcast<J, 0xf, J == 0>(lambda, inf_pressure, q.get(MF_MROW0 + J));
                } else {  // two chains per row: lanes 0-7 take lane J, lanes 8-15 lane 8 + J
                    fmac_row_bcast<J, 0x3, J == 0>(lambda, inf_pressure, q.get(MF_MROW0 + J));
                    fmac_row_bcast<8 + J, 0xc, false>(lambda, inf_pressure, q.get(MF_MROW0 + J));
                }
            }()), ...);
        }(std::make_integer_sequence<int, LPC>{});
    } else {
        lambda = q.get(MF_MROW0) * group_bcast<LPC, 0>(inf_pressure);
        [&]<int... J>(std::integer_sequence<int, J...>) {
            ((lambda = fma(q.get(MF_MROW0 + J + 1), group_bcast<LPC, J + 1>(inf_pressure), lambda)), ...);
        }(std::make_integer_sequence<int, LPC - 1>{});
    }
#else
    double lambda = q.get(MF_MROW0) * group_bcast<LPC, 0>(inf_pressure);
    [&]<int... J>(std::integer_sequence<int, J...>) {
        ((lambda += q.get(MF_MROW0 + J + 1) * group_bcast<LPC, J + 1>(inf_pressure)), ...);
    }(std::make_integer_sequence<int, LPC - 1>{});
#endif
#if SEPAIHRD_ARITH_FMA
    lambda *= beta_eff;  // a_i folded into the contact row
#else
    lambda *= beta_eff * q.get(MF_A);
#endif
    const double lambda_val = (0.0 < lambda) ? lambda : 0.0;  // std::max(0.0, lambda)

    const double flow_SE = lambda_val * S;
#if SEPAIHRD_ARITH_FMA
    const double flow_IH = q.get(MF_H) * I;        // also d CumH
    const double flow_H_ICU = q.get(MF_ICU) * H;   // also d CumICU
    dx[0] = -flow_SE;
    dx[1] = fma(-q.get(MF_SIGMA), E, flow_SE);
    dx[2] = fma(-q.get(MF_GAMMA_P), P, q.get(MF_SIGMA) * E);
    dx[3] = fma(-q.get(MF_GAMMA_A), A, q.get(MF_PG) * P);
    dx[4] = fma(-q.get(MF_R_I), I, q.get(MF_PI) * P);
    dx[5] = fma(-q.get(MF_R_H), H, flow_IH);
    dx[6] = fma(-q.get(MF_R_ICU), ICU, flow_H_ICU);
    dx[7] = fma(q.get(MF_GAMMA_ICU), ICU, fma(q.get(MF_GAMMA_A), A, fma(q.get(MF_GAMMA_H), H, q.get(MF_GAMMA_I) * I)));
    dx[8] = fma(q.get(MF_D_ICU), ICU, fma(q.get(MF_D_COMM), I, q.get(MF_D_H) * H));  // H, I, ICU: the order the 16-lane form's slot-2 stream visits them in
    dx[9] = flow_IH;
    dx[10] = flow_H_ICU;
#else
    const double flow_EP = q.get(MF_SIGMA) * E;
    const double flow_P_out = q.get(MF_GAMMA_P) * P;
    const double flow_PA = q.get(MF_P) * flow_P_out;
    const double flow_PI = flow_P_out - flow_PA;
    const double flow_IH = q.get(MF_H) * I;
    const double flow_IR = q.get(MF_GAMMA_I) * I;
    const double flow_ID_community = q.get(MF_D_COMM) * I;
    const double flow_H_ICU = q.get(MF_ICU) * H;

    dx[0] = -flow_SE;
    dx[1] = flow_SE - flow_EP;
    dx[2] = flow_EP - flow_P_out;
    dx[3] = flow_PA - q.get(MF_GAMMA_A) * A;
    const double I_out = flow_IR + flow_IH + flow_ID_community;
    const double H_out = q.get(MF_GAMMA_H) * H + q.get(MF_D_H) * H + flow_H_ICU;
    const double ICU_out = (q.get(MF_GAMMA_ICU) + q.get(MF_D_ICU)) * ICU;
    dx[4] = flow_PI - I_out;
    dx[5] = flow_IH - H_out;
    dx[6] = flow_H_ICU - ICU_out;
    dx[7] = q.get(MF_GAMMA_A) * A + flow_IR + q.get(MF_GAMMA_H) * H + q.get(MF_GAMMA_ICU) * ICU;
    dx[8] = q.get(MF_D_H) * H + q.get(MF_D_ICU) * ICU + flow_ID_community;
    dx[9] = flow_IH;
    dx[10] = flow_H_ICU;
#endif
}

// Diagnostic build only (-DSEPAIHRD_STAMPS): s_memtime stamps at section boundaries of the RK loop,
// summed per wave and written to out.ll_parts of the wave's first chain.  Never in the product build.
#ifdef SEPAIHRD_STAMPS
#define SEP_STAMP(var)                                                                 \
    do {                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                             \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");     \
        __builtin_amdgcn_sched_barrier(0);                                             \
    } while (0)
#else
#define SEP_STAMP(var) do { } while (0)
#endif

// |e| / s of the error norm.  strict: the IEEE division of the CPU build.  fma: Newton-refined
// reciprocal (4 instructions instead of 15); s > 0 is a tolerance scale, far from the overflow /
// underflow cases the IEEE sequence guards against.
__device__ __forceinline__ double quotient(double e, double s) {
#if SEPAIHRD_ARITH_FMA
    double r = __builtin_amdgcn_rcp(s);   // ~2^-26 relative
    r = fma(fma(-s, r, 1.0), r, r);       // one Newton step: ~2^-52
    return e * r;                         // a couple of ulp: the value only drives the step-size rule
#else
    return e / s;
#endif
}

// exp for the step-size controller: p = k ln2 + r, |r| <= ln2/2, degree-13 Taylor (remainder < 5e-18).
__device__ __forceinline__ double exp_ctl(double p) {
    constexpr double log2e = 1.44269504088896338700e+00;
    constexpr double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    const double kd = rint(p * log2e);
    double r = fma(-kd, ln2_hi, p);
    r = fma(-kd, ln2_lo, r);
    double q = 1.0 / 6227020800.0;
    q = fma(q, r, 1.0 / 479001600.0);
    q = fma(q, r, 1.0 / 39916800.0);
    q = fma(q, r, 1.0 / 3628800.0);
    q = fma(q, r, 1.0 / 362880.0);
    q = fma(q, r, 1.0 / 40320.0);
    q = fma(q, r, 1.0 / 5040.0);
    q = fma(q, r, 1.0 / 720.0);
    q = fma(q, r, 1.0 / 120.0);
    q = fma(q, r, 1.0 / 24.0);
    q = fma(q, r, 1.0 / 6.0);
    q = fma(q, r, 0.5);
    q = fma(q, r, 1.0);
    q = fma(q, r, 1.0);
    return ldexp(q, (int)kd);
}
// x^c for the controller's err^(-1/3), err^(-1/5): exp(c log x), a few ulp -- the reference calls
// std::pow (libm, not bit-pinned); the result only scales the next trial step.
__device__ __forceinline__ double pow_ctl(double x, double c) {
#if SEPAIHRD_ARITH_FMA
    // tolerance mode: the factor only scales the next TRIAL step, whose local error the controller checks
    // again; the hardware's single-precision log2 / exp2 (~1e-7 relative) are exact enough and cost five
    // instructions instead of fifty-seven.  err = +inf gives 0 (floored at 1/5 by the caller), like pow.
    const float l2 = __builtin_amdgcn_logf((float)x);
    return (double)__builtin_amdgcn_exp2f((float)c * l2);
#else
    return exp_ctl(c * log_ctl(x));
#endif
}

#if SEPAIHRD_ARITH_FMA
// Stage coefficients through DPP (tolerance build).  Every stage sum multiplies the chain's step `cur` by tableau
// constants: 26 v_mul_f64 per attempt, each constant a 64-bit literal that has to sit in (or be re-materialised into)
// an SGPR pair.  A chain is a 16-lane DPP row here, so lane l of the row computes cur * coefficient[l] ONCE (two
// multiplications fill two 16-entry vectors) and every term of a stage sum after the first takes its factor from there
// as the row_newbcast source of the v_fmac_f64_dpp that adds the term: the broadcast costs nothing.  The first term of a
// sum (x + f1 k1: a three-operand fma) and the two leading products of the error estimate keep their own per-lane
// factor.  Same products, same fmas, same order as written out term by term (and as the 4-lane kernel): same bits.
enum QuadCoefDopri5 { QD_B32 = 0, QD_B42, QD_B43, QD_B52, QD_B53, QD_B54, QD_B62, QD_B63, QD_B64, QD_B65, QD_C3, QD_C4, QD_C5,
                      QD_C6, QD_DC4, QD_DC5, QD_DC6 /* vector B lane 0 */, QD_DC7 /* vector B lane 1 */ };
enum QuadCoefCashKarp { QK_A32 = 0, QK_A42, QK_A43, QK_A52, QK_A53, QK_A54, QK_A62, QK_A63, QK_A64, QK_A65, QK_B3, QK_B4, QK_B6,
                        QK_DB4, QK_DB5, QK_DB6 };
__device__ const double QUAD_COEF[2][2][16] = {
    {{dp::b32, dp::b42, dp::b43, dp::b52, dp::b53, dp::b54, dp::b62, dp::b63, dp::b64, dp::b65, dp::c3, dp::c4, dp::c5, dp::c6,
      dp::dc4, dp::dc5},
     {dp::dc6, dp::dc7, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}},
    {{ck::a32, ck::a42, ck::a43, ck::a52, ck::a53, ck::a54, ck::a62, ck::a63, ck::a64, ck::a65, ck::b3, ck::b4, ck::b6, ck::db4,
      ck::db5, ck::db6},
     {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}}};
// acc += (cur * coefficient IDX of the chain's row) * k
template <int IDX>
__device__ __forceinline__ void stage_term(double& acc, double vecA, double vecB, double k) {
    if constexpr (IDX < 16) fmac_row_bcast<IDX, 0xf, false>(acc, vecA, k);
    else fmac_row_bcast<IDX - 16, 0xf, false>(acc, vecB, k);
}


// The RK stages of one attempt with the row-coefficient vectors: k2..k7 (k1 for Cash-Karp), the new state and the
// error estimate.  rhs_call(x_in, k_out, beta_kappa) evaluates the right-hand side.  N = values per lane.
// Phase pads of the 16-age RK body of the tolerance build (N = 11 values per lane, one wave per SIMD: configs[4]; DESIGN.md 4,
// "The strict build's two speeds"): one 4-byte s_nop in front of (even bits) or behind (odd bits) the RHS of stage i / 2 when the
// bit is set in SEPAIHRD_FMA16_STAGE_PADS, tied to a value of that point so that it stays where it is written.  The mask is chosen
// with tools/check_code_phase.py (8-byte encodings off phase at 0.9 cycles, a pad at 4.3) and confirmed on the GPU.
#ifndef SEPAIHRD_FMA16_STAGE_PADS
#define SEPAIHRD_FMA16_STAGE_PADS 194  // greedy front-to-back search: 231 -> 180 of the body's 458 eight-byte encodings off phase, c5 18.40 -> 18.25 ms
#endif
#ifndef SEPAIHRD_FMA3_STAGE_PADS   // the same pad points in the 16-lane form's body (N = 3 values per lane)
#define SEPAIHRD_FMA3_STAGE_PADS 0
#endif
#define SEP_ROW_STAGE_PAD(i, v) do { if ((((N == NUM_COMP) ? SEPAIHRD_FMA16_STAGE_PADS : (N == 3) ? SEPAIHRD_FMA3_STAGE_PADS : 0) >> (i)) & 1) asm volatile("s_nop 0" : "+v"(v)); } while (0)

template <int SOLVER, int N, class RHS>
__device__ __forceinline__ void rk_stages_row_coef(const double cur, const double vecA, const double vecB, const double (&x)[N],
                                                   double (&k1)[N], double (&k2)[N], double (&k3)[N], double (&k4)[N], double (&k5)[N],
                                                   double (&k6)[N], double (&k7)[N], double (&xnew)[N], double (&xerr)[N],
                                                   const double (&bks)[7], RHS&& rhs_call) {
    double xt[N];
        if (SOLVER == 0) {
            { const double f1 = cur * dp::b21;
              SEP_UNROLL for (int c = 0; c < N; ++c) xt[c] = fma(f1, k1[c], x[c]);
              SEP_ROW_STAGE_PAD(0, xt[0]); rhs_call(xt, k2, bks[1]); }
            SEP_ROW_STAGE_PAD(1, k2[0]);
            { const double f1 = cur * dp::b31;
              SEP_UNROLL for (int c = 0; c < N; ++c) { xt[c] = fma(f1, k1[c], x[c]); stage_term<QD_B32>(xt[c], vecA, vecB, k2[c]); }
              SEP_ROW_STAGE_PAD(2, xt[0]); rhs_call(xt, k3, bks[2]); }
            SEP_ROW_STAGE_PAD(3, k3[0]);
            { const double f1 = cur * dp::b41;
              SEP_UNROLL for (int c = 0; c < N; ++c) {
                  xt[c] = fma(f1, k1[c], x[c]); stage_term<QD_B42>(xt[c], vecA, vecB, k2[c]); stage_term<QD_B43>(xt[c], vecA, vecB, k3[c]); }
              SEP_ROW_STAGE_PAD(4, xt[0]); rhs_call(xt, k4, bks[3]); }
            SEP_ROW_STAGE_PAD(5, k4[0]);
            { const double f1 = cur * dp::b51;
              SEP_UNROLL for (int c = 0; c < N; ++c) {
                  xt[c] = fma(f1, k1[c], x[c]); stage_term<QD_B52>(xt[c], vecA, vecB, k2[c]); stage_term<QD_B53>(xt[c], vecA, vecB, k3[c]);
                  stage_term<QD_B54>(xt[c], vecA, vecB, k4[c]); }
              SEP_ROW_STAGE_PAD(6, xt[0]); rhs_call(xt, k5, bks[4]); }
            SEP_ROW_STAGE_PAD(7, k5[0]);
            { const double f1 = cur * dp::b61;
              SEP_UNROLL for (int c = 0; c < N; ++c) {
                  xt[c] = fma(f1, k1[c], x[c]); stage_term<QD_B62>(xt[c], vecA, vecB, k2[c]); stage_term<QD_B63>(xt[c], vecA, vecB, k3[c]);
                  stage_term<QD_B64>(xt[c], vecA, vecB, k4[c]); stage_term<QD_B65>(xt[c], vecA, vecB, k5[c]); }
              SEP_ROW_STAGE_PAD(8, xt[0]); rhs_call(xt, k6, bks[5]); }
            SEP_ROW_STAGE_PAD(9, k6[0]);
            { const double f1 = cur * dp::c1;
              SEP_UNROLL for (int c = 0; c < N; ++c) {
                  xnew[c] = fma(f1, k1[c], x[c]); stage_term<QD_C3>(xnew[c], vecA, vecB, k3[c]); stage_term<QD_C4>(xnew[c], vecA, vecB, k4[c]);
                  stage_term<QD_C5>(xnew[c], vecA, vecB, k5[c]); stage_term<QD_C6>(xnew[c], vecA, vecB, k6[c]); }
              SEP_ROW_STAGE_PAD(10, xnew[0]); rhs_call(xnew, k7, bks[6]); }
            SEP_ROW_STAGE_PAD(11, k7[0]);
            // e1 k1 + e3 k3 + ...: of two leading products the contraction pass rounds the second and fuses the first
            // (see the 4-lane kernel's listing); written out so that it does not depend on the pass
            { const double e1 = cur * dp::dc1, e3 = cur * dp::dc3;
              SEP_UNROLL for (int c = 0; c < N; ++c) {
                  xerr[c] = fma(e1, k1[c], e3 * k3[c]); stage_term<QD_DC4>(xerr[c], vecA, vecB, k4[c]); stage_term<QD_DC5>(xerr[c], vecA, vecB, k5[c]);
                  stage_term<QD_DC6>(xerr[c], vecA, vecB, k6[c]); stage_term<QD_DC7>(xerr[c], vecA, vecB, k7[c]); } }
        } else {
            rhs_call(x, k1, bks[0]);
            { const double f1 = ck::a21 * cur;
              SEP_UNROLL for (int c = 0; c < N; ++c) xt[c] = fma(f1, k1[c], x[c]);
              rhs_call(xt, k2, bks[1]); }
            { const double f1 = ck::a31 * cur;
              SEP_UNROLL for (int c = 0; c < N; ++c) { xt[c] = fma(f1, k1[c], x[c]); stage_term<QK_A32>(xt[c], vecA, vecB, k2[c]); }
              rhs_call(xt, k3, bks[2]); }
            { const double f1 = ck::a41 * cur;
              SEP_UNROLL for (int c = 0; c < N; ++c) {
                  xt[c] = fma(f1, k1[c], x[c]); stage_term<QK_A42>(xt[c], vecA, vecB, k2[c]); stage_term<QK_A43>(xt[c], vecA, vecB, k3[c]); }
              rhs_call(xt, k4, bks[3]); }
            { const double f1 = ck::a51 * cur;
              SEP_UNROLL for (int c = 0; c < N; ++c) {
                  xt[c] = fma(f1, k1[c], x[c]); stage_term<QK_A52>(xt[c], vecA, vecB, k2[c]); stage_term<QK_A53>(xt[c], vecA, vecB, k3[c]);
                  stage_term<QK_A54>(xt[c], vecA, vecB, k4[c]); }
              rhs_call(xt, k5, bks[4]); }
            { const double f1 = ck::a61 * cur;
              SEP_UNROLL for (int c = 0; c < N; ++c) {
                  xt[c] = fma(f1, k1[c], x[c]); stage_term<QK_A62>(xt[c], vecA, vecB, k2[c]); stage_term<QK_A63>(xt[c], vecA, vecB, k3[c]);
                  stage_term<QK_A64>(xt[c], vecA, vecB, k4[c]); stage_term<QK_A65>(xt[c], vecA, vecB, k5[c]); }
              rhs_call(xt, k6, bks[5]); }
            { const double f1 = ck::b1 * cur;
              SEP_UNROLL for (int c = 0; c < N; ++c) {
                  xnew[c] = fma(f1, k1[c], x[c]); stage_term<QK_B3>(xnew[c], vecA, vecB, k3[c]); stage_term<QK_B4>(xnew[c], vecA, vecB, k4[c]);
                  stage_term<QK_B6>(xnew[c], vecA, vecB, k6[c]); } }
            // here the pass rounds the FIRST product and fuses the second (it rounds the one whose coefficient is
            // negative: db1 < 0 < db3, while dc3 < 0 < dc1 above)
            { const double e1 = ck::db1 * cur, e3 = ck::db3 * cur;
              SEP_UNROLL for (int c = 0; c < N; ++c) {
                  xerr[c] = fma(e3, k3[c], e1 * k1[c]); stage_term<QK_DB4>(xerr[c], vecA, vecB, k4[c]); stage_term<QK_DB5>(xerr[c], vecA, vecB, k5[c]);
                  stage_term<QK_DB6>(xerr[c], vecA, vecB, k6[c]); } }
        }
}
#endif

// ----------------------------------------------------------------------------------
// the evaluation kernel: block = one wavefront = 64/LPC chains
// ----------------------------------------------------------------------------------
// WPS = waves per SIMD the register allocation is budgeted for.  1 (up to 512 unified VGPRs) is the
// default; the Cash-Karp stepper (no FSAL derivative, fewer live stage vectors) also gets a 2-wave
// build, launched when the batch is large enough to put two waves on every SIMD: measured +13..22 %
// there, while the 4-age Dopri5 body loses at 2 waves/SIMD with the logs inline (its spills go to scratch; it reaches
// two waves the other way, with the likelihood as a separate pass).  Round 4: the SIXTEEN-age Dopri5 integrator of the
// tolerance build takes the 2-wave budget as well -- measured in one box on configs[4] (tools/ab.sh): 17.92 -> 17.01 ms
// per 32 768-chain step (+5.3 %) although 40 registers go to scratch.
// INLINE_LL selects where the Poisson terms are evaluated:
//   false  the integrator parks the daily increments of D, CumH, CumICU in HBM and a separate pass,
//          parallel over (chain, day, age), evaluates the likelihood -- best while the batch does not
//          fill the chip (4096 chains: 1.24 -> 1.09 ms per step);
//   true   three logs per output inside the wave, observation records prefetched by LDS-DMA -- best
//          at saturation, where the separate pass would add HBM traffic to the same VALU work.
// does the tolerance build's Dopri5 integrator of this lane count have a two-waves-per-SIMD form? (the comment at split_pays)
template <int LPC>
constexpr bool dopri5_two_waves() { return SEPAIHRD_ARITH_FMA != 0 && ((LPC == 4 && SEPAIHRD_LL_STATE_IN_LDS != 0) || LPC == 16); }
// One-wave-per-SIMD forms that are launched for at most one wave per SIMD (<= 1024 workgroups; above that their two-wave
// sibling takes over) declare a high accumulation register they never touch: the allocation passes 256 registers, two of
// their waves no longer fit a SIMD, and the dispatcher cannot pair them up while other SIMDs stay empty (see split_pays).
template <int LPC, int SOLVER, int WPS>
constexpr bool one_wave_per_simd_only() { return WPS == 1 && (SOLVER == 1 || dopri5_two_waves<LPC>()); }

template <int LPC, int SOLVER, int ARITH_FMA, int WPS, bool INLINE_LL>
__global__ __launch_bounds__(WAVE, WPS) void sepaihrd_eval_kernel(const DevProblem pb,
                                                              const double* __restrict__ theta,
                                                              const int B, const EvalOutputs out,
                                                              const int cum_chains) {
    constexpr int CPW = WAVE / LPC;
    if constexpr (one_wave_per_simd_only<LPC, SOLVER, WPS>()) asm volatile("" ::: "a31");
    extern __shared__ __attribute__((aligned(16))) double lds[];
    // the inline-likelihood builds take the next grid time from the observation record: no output grid in their LDS (it was
    // 8 KB per wave at 1001 days -- with 2 KB more for the log table a CU held 7 waves of the 16-age kernel instead of 8)
    double* const lds_times = lds;                     // [T] output grid (separate-pass builds only)
    double* const lds_rec = lds + (INLINE_LL ? 0 : times_pad(pb));  // [2][64 lanes][2]  LDS-DMA landing zone (INLINE_LL)
    double* const lds_mends = lds_rec + LDS_REC_DOUBLES;  // [nm_pad]
    double* const lds_bk = lds_mends + pb.nm_pad;      // [CPW][nm + 1]
    double* const lds_theta = lds_bk + CPW * (pb.nm + 1);
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
            lds_theta[idx] = constrain(src[idx], pb.lower[p], pb.upper[p], pb.has_bounds[p], pb.constraint_mode);
        }
    }
    for (int k = lane; k < pb.nm_pad; k += WAVE) lds_mends[k] = pb.mends[k];
    if constexpr (!INLINE_LL) {
        for (int k = lane; k < pb.T; k += WAVE) lds_times[k] = pb.times[k];
    }
    if constexpr (INLINE_LL) stage_log_table(lane, WAVE);  // the Poisson term's log reads its table from LDS
    __syncthreads();
    const double* th = lds_theta + g * P;

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
    q.set(MF_THETA, scalar_slot(SS_THETA));
    q.set(MF_SIGMA, scalar_slot(SS_SIGMA));
    q.set(MF_GAMMA_P, scalar_slot(SS_GAMMA_P));
    q.set(MF_GAMMA_A, scalar_slot(SS_GAMMA_A));
    q.set(MF_GAMMA_I, scalar_slot(SS_GAMMA_I));
    q.set(MF_GAMMA_H, scalar_slot(SS_GAMMA_H));
    q.set(MF_GAMMA_ICU, scalar_slot(SS_GAMMA_ICU));
    q.set(MF_A, vec_slot(VF_A));
    q.set(MF_H_INFEC, vec_slot(VF_H_INFEC));
    q.set(MF_P, vec_slot(VF_P));
    q.set(MF_H, vec_slot(VF_H));
    q.set(MF_ICU, vec_slot(VF_ICU));
    q.set(MF_D_H, vec_slot(VF_D_H));
    q.set(MF_D_ICU, vec_slot(VF_D_ICU));
    q.set(MF_D_COMM, vec_slot(VF_D_COMM));
    const double Ni = pb.N[age];
    q.set(MF_INV_N, (Ni > 1e-9) ? (1.0 / Ni) : 0.0);  // AgeSEPAIHRDModel.cpp:332-334
#if SEPAIHRD_ARITH_FMA
    q.set(MF_H_INFEC, q.get(MF_H_INFEC) * q.get(MF_INV_N));
    q.set(MF_R_I, q.get(MF_GAMMA_I) + q.get(MF_H) + q.get(MF_D_COMM));
    q.set(MF_R_H, q.get(MF_GAMMA_H) + q.get(MF_D_H) + q.get(MF_ICU));
    q.set(MF_R_ICU, q.get(MF_GAMMA_ICU) + q.get(MF_D_ICU));
    q.set(MF_PG, rounded_here(q.get(MF_P) * q.get(MF_GAMMA_P)));
    q.set(MF_PI, q.get(MF_GAMMA_P) - q.get(MF_PG));
    // column j of the contact row also carries age class j's h_infec / N: (a_i M(i,j)) * c_j, so that the pressure
    // P + A + theta I needs no scaling of its own in the RHS (one multiplication fewer per call, in every form)
    [&]<int... J>(std::integer_sequence<int, J...>) {
        ((q.set(MF_MROW0 + J, q.get(MF_A) * pb.Mrow[age * LPC + J] * group_bcast<LPC, J>(q.get(MF_H_INFEC)))), ...);
    }(std::make_integer_sequence<int, LPC>{});
#else
    SEP_UNROLL
    for (int j = 0; j < LPC; ++j) q.set(MF_MROW0 + j, pb.Mrow[age * LPC + j]);
#endif

    // beta(t) kappa(t) on every merged segment: "current_beta * reduction_factor" of the RHS, once per chain
    Schedule sch;
    sch.me = lds_mends;
    {
        double* bkv = lds_bk + grp * (pb.nm + 1);  // own region even for shadow groups
        for (int j = age; j <= pb.nm; j += LPC) {
            const double beta = (pb.nb > 0) ? scalar_slot(SS_SCHEDULE0 + pb.seg_ib[j]) : scalar_slot(SS_BETA);
            const double kappa = scalar_slot(SS_SCHEDULE0 + pb.nb + pb.seg_ik[j]);
            bkv[j] = beta * kappa;
        }
        sch.bkv = bkv;
    }
    __syncthreads();

    int status = 0;
    if (pb.kappa_calibrated) {  // setCalibratableValues: any after-baseline kappa < 0 -> throw -> lowest()
        bool neg = false;
        for (int k = 1; k < pb.nk; ++k) neg |= (scalar_slot(SS_SCHEDULE0 + pb.nb + k) < 0.0);
        if (neg) status = 1;
    }
    if (!pb.obs_rows_match && pb.init_mode == 0) status = 1;  // SEPAIHRDObjectiveFunction.cpp:176-178

    // ---- 3. initial state (SEPAIHRDObjectiveFunction.cpp:124-163)
    double x[NUM_COMP];
    SEP_UNROLL
    for (int c = 0; c < NUM_COMP; ++c) x[c] = pb.init_state[c * LPC + age];
    if (pb.init_mode != 1) {  // 1: problem.initial_state as given (SimulationRunner.cpp:24-104)
        const double runup_days = scalar_slot(SS_RUNUP_DAYS);
        const double seed_exposed = scalar_slot(SS_SEED_EXPOSED);
        // 2: the finite-difference objective always scales by the multipliers and also rejects a
        //    negative non-S total (SEPAIHRDGradientObjectiveFunction.cpp:55-99)
        if (pb.init_mode == 0 && runup_days > 0 && seed_exposed > 0) {
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
        if (group_any<LPC>(sum > Ni || (pb.init_mode == 2 && sum < 0), lane)) status = 1;
        x[0] = Ni - sum;
    }

    int n_acc = 0, n_rej = 0;
    const int T = pb.T;
    const int n_real = pb.n;

    // Observer at output index k for the chains with do_it set.  The likelihood only needs the daily
    // increments of D, CumH, CumICU and never feeds back into the dynamics.
    double* const cum_lane = out.cum + cum_index(T, (size_t)(chain0 + grp) * LPC + age, 0, 0);  // own column, shadow groups too
    double prevD = x[8], prevH = x[9], prevICU = x[10];  // row 0: X(0) - init_state = 0
    double llH = 0.0, llICU = 0.0, llD = 0.0;             // INLINE_LL accumulators
    // Two-waves-per-SIMD builds of the tolerance arithmetic keep that state in LDS (LL_IN_LDS, round 4): with it the 4-age
    // Dopri5 integrator fits 256 registers WITH its logs inline (253, nothing spilled; 273 without), so configs[3]'s batch no
    // longer parks its increments in HBM (1.26 GB written and read back per 32 768-chain step): 1.92-2.00 -> 1.84 ms per step.
    // Dopri5 only: the Cash-Karp integrator has the registers to spare (252) and LOSES 2.4 % when its observer goes through LDS
    // (configs[2], same box: 3.015 -> 3.09 ms).
    constexpr bool LL_IN_LDS = INLINE_LL && SEPAIHRD_LL_STATE_IN_LDS != 0 && SEPAIHRD_ARITH_FMA != 0 && WPS == 2 && SOLVER == 0;
    double* const lds_ll = lds_theta + CPW * pb.P;  // [6][WAVE] behind the theta staging area (eval_lds_bytes budgets it)

    // INLINE_LL: grid record of output k for this lane, {obs_H, obs_ICU, obs_D, times[k+1]}, requested by
    // LDS-DMA (global_load_lds_dwordx4 x2: no VGPR destination) one whole RK step before it is read.
    const double* grid_lane = pb.grid + (size_t)age * 4;
    auto request_record = [&](int k) {
        const double* src = grid_lane + (size_t)k * (LPC * 4);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)lds_rec, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 2),
                                         (__attribute__((address_space(3))) void*)(lds_rec + 2 * WAVE), 16, 0, 0);
    };

    auto store_traj = [&](int k) {
        if (out.traj != nullptr && chain_valid && age < n_real) {
            double* tdst = out.traj + ((size_t)chain * T + k) * (NUM_COMP * n_real) + age;
            SEP_UNROLL
            for (int c = 0; c < NUM_COMP; ++c) tdst[c * n_real] = x[c];
        }
    };
    // split form: park the raw increments (one coalesced 512-B store per wave and compartment)
    auto observe_store = [&](bool do_it, int k) {
        if (do_it) {
            double* dst = cum_lane + (size_t)k * CUM_ROW_DOUBLES;
            dst[0] = x[8] - prevD;
            dst[WAVE] = x[9] - prevH;
            dst[2 * WAVE] = x[10] - prevICU;
            prevD = x[8]; prevH = x[9]; prevICU = x[10];
            store_traj(k);
        }
    };
    // inline form: all lanes execute, the terms of chains without do_it are discarded.  Records of
    // outputs before t = 0 and of padded ages carry NaN observations, which the Poisson term skips.
    // LL_IN_LDS: the observer's loop-carried state (the three previous values, the three stream sums) lives in LDS between
    // outputs instead of in twelve registers that are dead through the whole RK body; the theta staging area is free by then
    auto observe_inline = [&](bool do_it, int k, double oH, double oI, double oD) {
        if constexpr (LL_IN_LDS) {
            prevH = lds_ll[lane]; prevICU = lds_ll[WAVE + lane]; prevD = lds_ll[2 * WAVE + lane];
            llH = lds_ll[3 * WAVE + lane]; llICU = lds_ll[4 * WAVE + lane]; llD = lds_ll[5 * WAVE + lane];
        }
        double incH = x[9] - prevH, incICU = x[10] - prevICU, incD = x[8] - prevD;
        incH = (incH < 0.0) ? 0.0 : incH;  // cwiseMax(0.0)
        incICU = (incICU < 0.0) ? 0.0 : incICU;
        incD = (incD < 0.0) ? 0.0 : incD;
        prevH = do_it ? x[9] : prevH;
        prevICU = do_it ? x[10] : prevICU;
        prevD = do_it ? x[8] : prevD;
        const double sim[3] = {incH + 1e-10, incICU + 1e-10, incD + 1e-10};
        double lg[3];
        log_pos3(sim, lg);
        auto term = [&](double obs, int s) -> double {
            const double v = obs * lg[s] - sim[s];
            return (do_it && obs >= 0.0 && isfinite(obs)) ? v : 0.0;
        };
        const double tH = term(oH, 0), tI = term(oI, 1), tD = term(oD, 2);
        const double tv3[3] = {tH, tI, tD};
        double rs3[3];
        row_sums3<LPC>(tv3, rs3);  // ages ascending: calculateSingleLogLikelihood's inner loop
        llH += rs3[0];
        llICU += rs3[1];
        llD += rs3[2];
        if constexpr (LL_IN_LDS) {
            lds_ll[lane] = prevH; lds_ll[WAVE + lane] = prevICU; lds_ll[2 * WAVE + lane] = prevD;
            lds_ll[3 * WAVE + lane] = llH; lds_ll[4 * WAVE + lane] = llICU; lds_ll[5 * WAVE + lane] = llD;
        }
        if (do_it) store_traj(k);
    };

    // ---- 4. integrate_times(controlled stepper, ..., times, dt_hint, observer)
    bool active = (status == 0);
    int k_next = 1;  // index of the next output time to reach
    double t = pb.times[0];
    double t_next;
    double dt = pb.dt_hint;
    int fails = 0;
    int attempts = 0;
    t_next = (T > 1) ? pb.times[1] : t;
    if constexpr (INLINE_LL) {
        const double oH = grid_lane[0], oI = grid_lane[1], oD = grid_lane[2];  // record 0 is read directly
        if constexpr (LL_IN_LDS) {
            lds_ll[lane] = prevH; lds_ll[WAVE + lane] = prevICU; lds_ll[2 * WAVE + lane] = prevD;
            lds_ll[3 * WAVE + lane] = 0.0; lds_ll[4 * WAVE + lane] = 0.0; lds_ll[5 * WAVE + lane] = 0.0;
        }
        observe_inline(active, 0, oH, oI, oD);
        if (active && T > 1) request_record(1);
    } else {
        observe_store(active && chain_valid, 0);
    }
    if (T <= 1) active = false;

    sch.lo = INFINITY; sch.hi = -INFINITY; sch.bk = 0.0;  // empty segment: first step refreshes
    double k1[NUM_COMP];
    if (SOLVER == 0) {  // controlled FSAL stepper: initialize() at the first try_step
        int c0, c1;
        segment_index2(sch, pb.nm_pad, t, t, c0, c1);
        rhs<LPC>(q, x, k1, sch.bkv[c0]);
    }

    const double eps_abs = pb.abs_tol, eps_rel = pb.rel_tol;
#if SEPAIHRD_ARITH_FMA
    // sixteen lanes per chain = one DPP row per chain: the stage coefficients come as row broadcasts (see QUAD_COEF)
    constexpr bool ROW_COEF = (LPC == 16);
    const double coefA = ROW_COEF ? QUAD_COEF[SOLVER][0][lane & 15] : 0.0, coefB = ROW_COEF ? QUAD_COEF[SOLVER][1][lane & 15] : 0.0;
#endif

#ifdef SEPAIHRD_STAMPS
    unsigned long long st0 = 0, st1 = 0, st2 = 0, st3 = 0, st4 = 0;
    unsigned long long acc_head = 0, acc_body = 0, acc_err = 0, acc_tail = 0;
    unsigned long long st5 = 0, st6 = 0, acc_errA = 0, acc_tailA = 0;
#endif
#if SEPAIHRD_ARITH_FMA
    while (__ballot(active) != 0ull && attempts < pb.max_attempts) {  // the build-side guard rides in the loop condition
#else
    while (__ballot(active) != 0ull) {
#endif
        SEP_STAMP(st0);
        // min_abs(dt, t_next - t); finished chains idle with a harmless unit step
        const double cur = active ? fmin(dt, t_next - t) : 1.0;
#if SEPAIHRD_ARITH_FMA
        double vecA = 0.0, vecB = 0.0;
        if constexpr (ROW_COEF)
            asm("v_mul_f64 %0, %2, %3\n\tv_mul_f64 %1, %2, %4\n\ts_nop 1" : "=&v"(vecA), "=&v"(vecB) : "v"(cur), "v"(coefA), "v"(coefB));
#endif

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
            if (SEP_RARELY(__ballot(active && !in_seg) != 0ull)) {  // rare: the common path falls through
                // some chain's step leaves its cached segment: look the stages up again (rare)
                int c_lo, c_hi;
                segment_index2(sch, pb.nm_pad, tmin, tmax, c_lo, c_hi);
                const double v_lo = sch.bkv[c_lo], v_hi = sch.bkv[c_hi];
                if (__ballot(c_hi > c_lo + 1) == 0ull) {
                    // at most one breakpoint b inside the step: stages <= b take v_lo, later ones v_hi
                    const double bpt = (c_lo < pb.nm) ? sch.me[c_lo] : INFINITY;
                    SEP_UNROLL
                    for (int s = 0; s < 7; ++s) bks[s] = (tau[s] > bpt) ? v_hi : v_lo;
                } else {
                    for (int s = 0; s < 7; ++s) {
                        int ca, cb;
                        segment_index2(sch, pb.nm_pad, tau[s], tau[s], ca, cb);
                        bks[s] = sch.bkv[ca];
                    }
                }
                sch.lo = (c_hi > 0) ? sch.me[c_hi - 1] : -INFINITY;
                sch.hi = (c_hi < pb.nm) ? sch.me[c_hi] : INFINITY;
                sch.bk = v_hi;
                // every LDS read of this (rare) path is complete before it rejoins: the stage code then carries no
                // s_waitcnt of its own (six per attempt otherwise, one per stage's beta*kappa)
                __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
            } else {
                SEP_UNROLL
                for (int s = 0; s < 7; ++s) bks[s] = sch.bk;
            }
        }

        SEP_STAMP(st1);
        double k2[NUM_COMP], k3[NUM_COMP], k4[NUM_COMP], k5[NUM_COMP], k6[NUM_COMP];
        double xt[NUM_COMP], xnew[NUM_COMP], xerr[NUM_COMP];
        double k7[NUM_COMP];

#if SEPAIHRD_ARITH_FMA
        if constexpr (ROW_COEF) {
            rk_stages_row_coef<SOLVER, NUM_COMP>(cur, vecA, vecB, x, k1, k2, k3, k4, k5, k6, k7, xnew, xerr, bks,
                                                 [&](const double (&xin)[NUM_COMP], double (&kout)[NUM_COMP], double bk) { rhs<LPC>(q, xin, kout, bk); });
        } else
#endif
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

        // default_error_checker: err = max_i |xerr_i| / (eps_abs + eps_rel (|x_i| + dt |dxdt_i|)) with the
        // start-of-step x, dxdt; reject iff err > 1.  |e| <= s  =>  fl(|e|/s) <= 1 exactly (division is
        // monotone), so the 11 divisions are only needed when some quotient may exceed 1 or when the
        // value of err feeds the step-size controller (dt below the largest output gap).
        SEP_STAMP(st2);
        double sc[NUM_COMP], ea[NUM_COMP];
        bool over = false;
        SEP_UNROLL
        for (int c = 0; c < NUM_COMP; ++c) {
            sc[c] = eps_abs + eps_rel * (fabs(x[c]) + cur * fabs(k1[c]));
            ea[c] = fabs(xerr[c]);
            over |= (ea[c] > sc[c]);
        }
        // growth multiplies the trial step by at most 0.9 * (5^-5)^(-1/5) = 4.5, so it can only raise
        // dt = max(dt, grown) when 4.5 cur > dt, and it only matters while dt < the largest output gap
        const bool grow_relevant = (dt < pb.max_gap) && (4.5000001 * cur > dt);
        const bool need_err = active && (over || grow_relevant);
        double err = 0.0;
        if (__ballot(need_err) != 0ull) {
            SEP_UNROLL
            for (int c = 0; c < NUM_COMP; ++c) err = max_keep(err, quotient(ea[c], sc[c]));
            err = group_max<LPC>(err);
        }

        SEP_STAMP(st5);
        const bool reject = err > 1.0;
        ++attempts;
        // default_step_adjuster.  decrease: dt *= max(0.9 err^(-1/3), 0.2); increase (err < 0.5):
        // dt *= 0.9 max(5^-5, err)^(-1/5).  One exp(c log x) serves both directions; it is evaluated only
        // when some lane needs it, and growth only matters while dt is below the largest output gap
        // (dt = max(dt, grown) cannot change min(dt, gap) otherwise).
        const bool need_dec = active && reject;
        const bool need_inc = active && !reject && (err < 0.5) && grow_relevant;
#if SEPAIHRD_ARITH_FMA
        // Tolerance build: what only a rejection or a step-size change touches lives in the block only those attempts enter (see
        // the 16-lane form, sepaihrd_lane_split.inc: when no chain of the wave rejects or grows, cur_after = cur <= dt, so dt keeps
        // its value, every active chain accepted and nothing can give up).  The strict build keeps the form below.
        const bool acc = active && !reject;
        if (SEP_RARELY(__ballot(need_dec || need_inc) != 0ull)) {
            const double arg = max_moved_uniform(err, 1.0 / 3125.0);  // 5^-5: the floor of the increase rule; a rejected step has err > 1
            const double expo = need_dec ? -1.0 / (4 - 1) : -1.0 / 5;
            const double pw = 9.0 / 10.0 * pow_ctl(arg, expo);
            const double f = fmax(pw, 1.0 / 5.0);  // the floor of the decrease rule; an increase has pw > 1 (err < 0.5)
            const double cur_after = (need_dec || need_inc) ? cur * f : cur;
            const bool rej = need_dec;
            n_rej += rej ? 1 : 0;
            // failure: dt = reduced current_dt; success: dt = max_abs(dt, current_dt)
            dt = rej ? cur_after : (acc ? fmax(dt, cur_after) : dt);
            // failed_step_checker: throws when 500 consecutive failures precede this one
            if (rej && fails >= 500) { status = 2; active = false; }
            fails = acc ? 0 : fails + (rej ? 1 : 0);
        } else {
            fails = 0;
        }
        n_acc += acc ? 1 : 0;
        SEP_STAMP(st3);
        {
#else
        double cur_after = cur;
        if (SEP_RARELY(__ballot(need_dec || need_inc) != 0ull)) {
            const double arg = max_moved_uniform(err, 1.0 / 3125.0);  // 5^-5: the floor of the increase rule; a rejected step has err > 1
            const double expo = need_dec ? -1.0 / (4 - 1) : -1.0 / 5;
            const double pw = 9.0 / 10.0 * pow_ctl(arg, expo);
            const double f = fmax(pw, 1.0 / 5.0);  // the floor of the decrease rule; an increase has pw > 1 (err < 0.5)
            if (need_dec || need_inc) cur_after = cur * f;
        }

        SEP_STAMP(st3);
        {
            const bool rej = active && reject;
            const bool acc = active && !reject;
            n_rej += rej ? 1 : 0;
            n_acc += acc ? 1 : 0;
            // failure: dt = reduced current_dt; success: dt = max_abs(dt, current_dt)
            dt = rej ? cur_after : (acc ? fmax(dt, cur_after) : dt);
            // failed_step_checker: throws when 500 consecutive failures precede this one
            if (rej && fails >= 500) { status = 2; active = false; }
            fails = acc ? 0 : fails + (rej ? 1 : 0);
#endif
            if (acc) {
                t += cur;
                SEP_UNROLL
                for (int c = 0; c < NUM_COMP; ++c) x[c] = xnew[c];
                if (SOLVER == 0) {
                    SEP_UNROLL
                    for (int c = 0; c < NUM_COMP; ++c) k1[c] = k7[c];
                }
            }
            SEP_STAMP(st6);
            // less_with_sign(t, t_next, dt): t_next - t > epsilon
            const bool reached = acc && !((t_next - t) > DBL_EPSILON);
            if constexpr (INLINE_LL) {
                if (SEP_MOSTLY(__ballot(reached) != 0ull)) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the record requested a step ago
                    const double2 ra = *reinterpret_cast<const double2*>(lds_rec + 2 * lane);
                    const double2 rb = *reinterpret_cast<const double2*>(lds_rec + 2 * WAVE + 2 * lane);
                    observe_inline(reached, k_next, ra.x, ra.y, rb.x);
                    if (reached) {
                        t = t_next;  // integrate_times re-reads the exact grid time
                        ++k_next;
                        t_next = rb.y;
                        if (k_next >= T) active = false;
                        else request_record(k_next);
                    }
                }
            } else if (reached) {
                observe_store(chain_valid, k_next);
                t = t_next;  // integrate_times re-reads the exact grid time
                ++k_next;
                if (k_next >= T) active = false;
                else t_next = lds_times[k_next];
            }
#if !SEPAIHRD_ARITH_FMA
            if (active && attempts >= pb.max_attempts) { status = 3; active = false; }
#endif
        }
        SEP_STAMP(st4);
#ifdef SEPAIHRD_STAMPS
        acc_head += st1 - st0; acc_body += st2 - st1; acc_err += st3 - st2; acc_tail += st4 - st3;
        acc_errA += st5 - st2;                       // scale/compare + division path (before the pow paths)
        if (st6 > st3) acc_tailA += st6 - st3;       // reject/accept state update (before observe)
        st6 = 0;
#endif
    }

#if SEPAIHRD_ARITH_FMA
    if (active) status = 3;  // chains still integrating when the guard ended the loop
#endif
    // ---- 5. integrator status and step counters; the likelihood pass finishes the evaluation
    if (chain_valid && age == 0) {
        if constexpr (INLINE_LL) {  // total (SEPAIHRDObjectiveFunction.cpp:222-227)
            if constexpr (LL_IN_LDS) { llH = lds_ll[3 * WAVE + lane]; llICU = lds_ll[4 * WAVE + lane]; llD = lds_ll[5 * WAVE + lane]; }
            double total = (llH + llICU) + llD;
            if (status == 0 && (isnan(total) || isinf(total))) status = 1;
            if (status != 0) total = -DBL_MAX;
            out.loglik[chain] = total;
            if (out.status) out.status[chain] = status;
#ifndef SEPAIHRD_STAMPS
            if (out.ll_parts) {
                out.ll_parts[3 * chain + 0] = llH;
                out.ll_parts[3 * chain + 1] = llICU;
                out.ll_parts[3 * chain + 2] = llD;
            }
#endif
        } else {
            out.wstatus[chain] = status;  // the likelihood pass finishes the evaluation
        }
        if (out.n_accept) out.n_accept[chain] = n_acc;
        if (out.n_reject) out.n_reject[chain] = n_rej;
#ifdef SEPAIHRD_STAMPS
        if (out.ll_parts) {
            double* dbg = out.ll_parts + 3 * chain;
            if (grp == 0) { dbg[0] = (double)acc_head; dbg[1] = (double)acc_body; dbg[2] = (double)acc_err; }
            if (grp == 1) { dbg[0] = (double)acc_tail; dbg[1] = (double)attempts; dbg[2] = (double)acc_errA; }
            if (grp == 2) dbg[0] = (double)acc_tailA;
        }
#endif
    }
}

// ----------------------------------------------------------------------------------
// Likelihood pass 1: one lane per (chain, age) and one block row per output day.
// incidence = max(increment, 0), Poisson term
// obs log(sim + 1e-10) - (sim + 1e-10) for valid observations, summed over the ages of the chain in
// ascending order (the reference's inner loop, SEPAIHRDObjectiveFunction.cpp:264-276).
// ----------------------------------------------------------------------------------
constexpr int LL_DAYS_PER_BLOCK = 4;  // one wave per day: fewer, larger workgroups for the dispatcher
// (chain, stream) waves from which the serial walk beats the (chain, day, age)-parallel kernel + its rows[] round trip:
// measured (Dopri5, tolerance build) 16 384 chains 0.32 vs 0.22 ms, 32 768 chains 0.43 vs 0.45, 65 536 0.75 vs 0.85,
// 131 072 1.11 vs 1.66 -- the walk wants at least 1.5 waves on every SIMD
constexpr int LL_SERIAL_MIN_WAVES = 1536;
template <int LPC>
__global__ __launch_bounds__(WAVE * LL_DAYS_PER_BLOCK) void sepaihrd_ll_terms_kernel(const DevProblem pb, const int B,
                                                                                    const EvalOutputs out,
                                                                                    const int cum_chains) {
    const int lane = threadIdx.x % WAVE;
    const size_t col = (size_t)blockIdx.x * WAVE + lane;  // chain * LPC + age
    const int k = blockIdx.y * LL_DAYS_PER_BLOCK + threadIdx.x / WAVE;
    stage_log_table(threadIdx.x, WAVE * LL_DAYS_PER_BLOCK);
    __syncthreads();
    if (k >= pb.T) return;
    const size_t chain = col / LPC;
    const int age = (int)(col % LPC);
    const bool valid = chain < (size_t)B;
    const size_t c = valid ? col : 0;
    if (k < pb.runup_offset) {  // outputs before t = 0 carry no observation: every term is 0 (wave-uniform)
        if (valid && age == 0) {
            double* dst = out.rows + (size_t)k * 3 * cum_chains + chain;
            dst[0] = 0.0;
            dst[cum_chains] = 0.0;
            dst[2 * (size_t)cum_chains] = 0.0;
        }
        return;
    }
    const double* cur = out.cum + cum_index(pb.T, c, k, 0);
    const double* rec = pb.grid + ((size_t)k * LPC + age) * 4;  // {obs_H, obs_ICU, obs_D, t_{k+1}}
    double rs[3], tv3[3];
    SEP_UNROLL
    for (int s = 0; s < 3; ++s) {
        const int comp = (s == 0) ? 1 : (s == 1) ? 2 : 0;  // cum rows are D, CumH, CumICU; streams are H, ICU, D
        double inc = cur[(size_t)comp * WAVE];  // X(k) - X(k-1), written by the integrator
        inc = (inc < 0.0) ? 0.0 : inc;            // cwiseMax(0.0)
        const double obs = rec[s];
        const double sim = inc + 1e-10;
        const double v = obs * log_pos(sim) - sim;
        tv3[s] = (valid && obs >= 0.0 && isfinite(obs)) ? v : 0.0;
    }
    row_sums3<LPC>(tv3, rs);
    if (valid && age == 0) {
        double* dst = out.rows + (size_t)k * 3 * cum_chains + chain;
        dst[0] = rs[0];
        dst[cum_chains] = rs[1];
        dst[2 * (size_t)cum_chains] = rs[2];
    }
}

// Likelihood pass for saturating batches: PARTS adjacent lanes per (chain, stream) walk the days and, within a day, the
// ages -- the reference's two nested serial sums -- so there is no rows[] round trip; the increments of a lane's ages
// arrive in one vector load, the observations are wave-uniform (scalar loads).  Same operations in the same order as the
// (chain, day, age)-parallel kernel above, so the same bits: a day's row sum is ((t_0 + t_1) + t_2) + ... with the partial sum
// handed from lane to lane (part p adds its ages onto what part p - 1 hands over, one DPP move per hand-over), and the LAST part
// adds the rows in day order.  The stream's sum goes to rows[0][stream][chain] and the reduce kernel finishes with n_rows = 1.
// PARTS is a matter of balance, not of arithmetic: with one lane per (chain, stream) a 32 768-chain batch is 1536 waves on
// 1024 SIMDs -- half of them hold two waves, half one, and the kernel takes as long as the SIMDs with two (round 4: 0.377 ms,
// of which ~0.1 ms is that imbalance); two lanes per (chain, stream) make it 3072 waves, three on every SIMD.
template <int LPC, int PARTS>
__global__ __launch_bounds__(WAVE) void sepaihrd_ll_serial_kernel(const DevProblem pb, const int B, const EvalOutputs out,
                                                                   const int cum_chains) {
    static_assert(PARTS == 1 || PARTS == 2 || PARTS == 4, "a (chain, stream) is one lane, a pair or a quad");
    static_assert(LPC % PARTS == 0, "every part takes the same number of ages");
    constexpr int AGES = LPC / PARTS;                  // ages per lane
    stage_log_table(threadIdx.x, WAVE);
    __syncthreads();
    const int t = blockIdx.x * WAVE + threadIdx.x;
    const int chain = t / PARTS, part = t % PARTS;
    const int s = blockIdx.y;  // stream: H, ICU, D
    const bool valid = chain < B;
    const size_t c = valid ? (size_t)chain : 0;
    const int comp = (s == 0) ? 1 : (s == 1) ? 2 : 0;  // cum rows are D, CumH, CumICU
    const double* cur = out.cum + cum_index(pb.T, c * LPC + part * AGES, 0, comp);
    double acc = 0.0;
    // the sums are a dependent chain, the loads are not: DAYS days of increments are requested at a time
    constexpr int DAYS = (AGES <= 4) ? 8 : (AGES == 8 ? 4 : 2);
    for (int k0 = pb.runup_offset; k0 < pb.T; k0 += DAYS) {
        double inc[DAYS][AGES];
        SEP_UNROLL
        for (int d = 0; d < DAYS; ++d) {
            const int k = (k0 + d < pb.T) ? k0 + d : pb.T - 1;
            const double* row = cur + (size_t)k * CUM_ROW_DOUBLES;
            if constexpr (AGES % 4 == 0) {
                SEP_UNROLL
                for (int a = 0; a < AGES; a += 4) {
                    const double4 v = *reinterpret_cast<const double4*>(row + a);
                    inc[d][a] = v.x; inc[d][a + 1] = v.y; inc[d][a + 2] = v.z; inc[d][a + 3] = v.w;
                }
            } else if constexpr (AGES % 2 == 0) {
                SEP_UNROLL
                for (int a = 0; a < AGES; a += 2) {
                    const double2 v = *reinterpret_cast<const double2*>(row + a);
                    inc[d][a] = v.x; inc[d][a + 1] = v.y;
                }
            } else {
                SEP_UNROLL
                for (int a = 0; a < AGES; ++a) inc[d][a] = row[a];
            }
        }
        SEP_UNROLL
        for (int d = 0; d < DAYS; ++d) {
            const int k = k0 + d;
            if (k >= pb.T) break;
            const double* rec = pb.grid + ((size_t)k * LPC + part * AGES) * 4 + s;  // obs of the lane's a-th age at rec[4 a]
            double tv[AGES];
            SEP_UNROLL
            for (int a = 0; a < AGES; ++a) {
                const double x = (inc[d][a] < 0.0) ? 0.0 : inc[d][a];  // cwiseMax(0.0)
                const double obs = rec[4 * a];
                const double sim = x + 1e-10;
                const double v = obs * log_pos(sim) - sim;
                tv[a] = (obs >= 0.0 && isfinite(obs)) ? v : 0.0;
            }
            // ages ascending across the parts: part 0 starts the sum ("0.0 +" dropped: value-identical), every later part
            // continues the one its left neighbour hands over
            double r = tv[0];
            if constexpr (PARTS == 1) {
                SEP_UNROLL
                for (int a = 1; a < AGES; ++a) r += tv[a];
            } else {
                SEP_UNROLL
                for (int a = 1; a < AGES; ++a) r += tv[a];          // part 0's own ages (the others redo theirs below)
                SEP_UNROLL
                for (int p = 1; p < PARTS; ++p) {
                    // lane of part p takes the running sum of part p - 1 (quad_perm: every lane reads its left neighbour)
                    const double left = dpp_move<0x90>(r);           // quad_perm:[0,0,1,2]
                    if (part == p) {
                        r = left + tv[0];
                        SEP_UNROLL
                        for (int a = 1; a < AGES; ++a) r += tv[a];
                    }
                }
            }
            acc += r;  // meaningful in the last part's lane
        }
    }
    if (valid && part == PARTS - 1) out.rows[(size_t)s * cum_chains + chain] = acc;
}

// Likelihood pass 2: one lane per (chain, stream) adds the daily row sums in day order (the serial
// "log_likelihood += row_sum" of the reference) and lane 0 of each triple forms
// total = (hosp + icu) + deaths; NaN / Inf -> lowest() (SEPAIHRDObjectiveFunction.cpp:222-227).
// The additions are a dependent chain, the loads are not: 64 days are requested at a time.
__global__ __launch_bounds__(WAVE) void sepaihrd_ll_reduce_kernel(const DevProblem pb, const int B,
                                                                   const EvalOutputs out, const int cum_chains,
                                                                   const int n_rows) {
    // lane = 4 * chain_in_block + stream (stream 3 idles): 16 chains per wave, quad = one chain
    const int lane = threadIdx.x;
    const int stream = lane & 3;
    const int chain = blockIdx.x * 16 + (lane >> 2);
    const bool valid = chain < B && stream < 3;
    const int ch = chain < B ? chain : 0;
    const double* src = out.rows + (size_t)(stream < 3 ? stream : 0) * cum_chains + ch;
    const size_t step = (size_t)3 * cum_chains;
    double acc = 0.0;
    int k = 0;
    constexpr int IN_FLIGHT = 64;  // 401 rows = 7 round trips instead of 26 (16 in flight: 16 us per 4096 chains)
    for (; k + IN_FLIGHT <= n_rows; k += IN_FLIGHT) {
        double v[IN_FLIGHT];
        SEP_UNROLL
        for (int j = 0; j < IN_FLIGHT; ++j) v[j] = src[(size_t)(k + j) * step];
        SEP_UNROLL
        for (int j = 0; j < IN_FLIGHT; ++j) acc += v[j];
    }
    for (; k + 16 <= n_rows; k += 16) {
        double v[16];
        SEP_UNROLL
        for (int j = 0; j < 16; ++j) v[j] = src[(size_t)(k + j) * step];
        SEP_UNROLL
        for (int j = 0; j < 16; ++j) acc += v[j];
    }
    for (; k < n_rows; ++k) acc += src[(size_t)k * step];
    const double h = group_bcast<4, 0>(acc), i = group_bcast<4, 1>(acc), d = group_bcast<4, 2>(acc);
    if (!(chain < B) || stream != 0) return;
    int status = out.wstatus[chain];
    double total = (h + i) + d;
    if (status == 0 && (isnan(total) || isinf(total))) status = 1;
    if (status != 0) total = -DBL_MAX;
    out.loglik[chain] = total;
    if (out.status) out.status[chain] = status;
#ifndef SEPAIHRD_STAMPS
    if (out.ll_parts) {
        const bool ok = out.wstatus[chain] == 0;
        out.ll_parts[3 * chain + 0] = ok ? h : 0.0;
        out.ll_parts[3 * chain + 1] = ok ? i : 0.0;
        out.ll_parts[3 * chain + 2] = ok ? d : 0.0;
    }
#endif
    (void)valid;
}

#if SEPAIHRD_ARITH_FMA
#define SEP_QUAD_NAME "sepaihrd_eval_quad_kernel[fma]"
#else
#define SEP_QUAD_NAME "sepaihrd_eval_quad_kernel[strict]"
#endif
#include "sepaihrd_lane_split.inc"  // 16-lanes-per-chain form for small batches of the 4-age model

// Which form integrates a batch of B chains of a 4-age problem: the 16-lane form up to QUAD_MAX_CHAINS chains, unless
// the context asks for one form (sepaihrd_set_integrator_form: the parity tests run the same chains through both).
// Experiment builds (-DSEPAIHRD_EXPERIMENTS, tools/ only): SEPAIHRD_LANE_SPLIT=0|1 overrides from the environment.
inline bool lane_split_wanted(const DevProblem& pb, int B) {
#ifdef SEPAIHRD_EXPERIMENTS
    static const int mode = [] {
        const char* e = getenv("SEPAIHRD_LANE_SPLIT");
        return e == nullptr ? -1 : atoi(e);
    }();
    if (mode >= 0) return mode != 0;
#endif
    return pb.form == 0 ? B <= QUAD_MAX_CHAINS : pb.form == 2;
}
#if SEPAIHRD_ARITH_FMA && defined(SEPAIHRD_EXPERIMENTS)
#include "sepaihrd_wave_chain.inc"  // one wavefront per chain (measured, loses everywhere; not in the shipped library)
#define SEPAIHRD_HAVE_WAVE_CHAIN 1
#else
#define SEPAIHRD_HAVE_WAVE_CHAIN 0
#endif

// (chain, stream) waves from which the separate likelihood pass walks the days in one lane per (chain, stream); both
// forms give the same bits.  Experiment builds: SEPAIHRD_LL_SERIAL_MIN_WAVES=n overrides.
template <int LPC>
inline int ll_serial_min_waves() {
#ifdef SEPAIHRD_EXPERIMENTS
    static const int v = [] {
        const char* e = getenv("SEPAIHRD_LL_SERIAL_MIN_WAVES");
        return e == nullptr ? -1 : atoi(e);
    }();
    if (v >= 0) return v;
#endif
    // 16 lanes per chain (round 3, 1001 days): the walk takes 3.3 - 4.4 ms whatever the batch (one lane walks 16 ages x 1001
    // days), the parallel kernel 1.3 / 2.5 / 5.0 ms at 8192 / 16 384 / 32 768 chains: they cross near 28 000 chains
    return LPC >= 16 ? 1344 : LL_SERIAL_MIN_WAVES;
}

// ----------------------------------------------------------------------------------
// launch plumbing
// ----------------------------------------------------------------------------------
// Does a launch of `blocks` wavefronts evaluate the likelihood as a separate pass over increments parked in HBM
// (T 3 n 8 bytes per evaluation, written once and read once)?  Up to one wave per SIMD the chip is not full and the pass
// always wins (three logs per output leave the serial critical path of every wave).  Beyond that it pays only where
// the integrator without the inline logs gains a second wave per SIMD from it: its register count says so (<= 256:
// Dopri5 in fma arithmetic with up to 4 age classes -- 15.1 M vs 13.9 M evals/s at 32 768 chains).  The 16-age
// integrator needs 274 registers either way: parking bought it nothing and cost 12.6 GB written + read per 32 768-chain
// step (22 % of the step in the pass that reads them back); it keeps its logs inline.
// The Dopri5 integrator of the tolerance build takes the two-waves-per-SIMD register budget where that was MEASURED to pay
// (tools/ab.sh, one box each): 4 lanes per chain with the likelihood's state in LDS (253 registers, nothing spilled: configs[3]
// 1.92-2.00 -> 1.84 ms per step) and 16 lanes per chain (256 registers, 16 spilled: configs[4] +3-5 %).  Other lane counts keep
// one wave per SIMD (8 lanes per chain would spill 31 registers; unmeasured).
// (dopri5_two_waves<LPC>() is defined in front of the kernel, which asks it too)

template <int LPC, int SOLVER>
inline bool split_pays(size_t blocks) {
#ifdef SEPAIHRD_EXPERIMENTS
    static const int forced = [] {
        const char* e = getenv("SEPAIHRD_SPLIT_LL");
        return e == nullptr ? -1 : atoi(e);
    }();
    if (forced >= 0 && blocks > (size_t)SPLIT_LL_MAX_BLOCKS) return forced != 0;
#endif
    // Between half a wave and one wave per SIMD the tolerance build's Dopri5 integrator keeps its likelihood inline since
    // round 4.  The separate-pass kernel fits two of its 255-register waves on a SIMD, and behind any other kernel that has
    // touched tens of MB the dispatcher does pair them up while other SIMDs stay empty (16 384 chains: 0.95 ms in a loop of
    // evaluations, 1.46 ms behind a 64-MB elementwise kernel or inside the sampler; tools/probe_dispatch_placement.py); the
    // inline kernel needs 273 registers, cannot be paired, and with the table log costs 1.10 ms against 0.95 + 0.21.
    if (SEPAIHRD_ARITH_FMA != 0 && SOLVER == 0 && LPC >= 2 && blocks > (size_t)SPLIT_LL_MAX_BLOCKS / 2 && blocks <= (size_t)SPLIT_LL_MAX_BLOCKS) return false;
    if (blocks <= (size_t)SPLIT_LL_MAX_BLOCKS) return true;
#ifdef SEPAIHRD_SPLIT_ONLY_BELOW_MAX  // A/B builds: the inline forms everywhere above SEPAIHRD_SPLIT_LL_MAX_BLOCKS
    return false;
#endif
    if (!(SEPAIHRD_ARITH_FMA != 0 && SOLVER == 0)) return false;
    // from two waves per SIMD on the tolerance build's 4-age Dopri5 integrator keeps its logs inline at 256 registers (LL_IN_LDS)
    if (dopri5_two_waves<LPC>()) return false;  // (above 1024 workgroups: the two-wave inline form)
    static const bool second_wave = [] {
        hipFuncAttributes attr;
        return hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(&sepaihrd_eval_kernel<LPC, SOLVER, SEPAIHRD_ARITH_FMA, 1, false>)) == hipSuccess &&
               attr.numRegs <= 256;
    }();
    return second_wave;
}

template <int LPC, int SOLVER, int WPS, bool INLINE_LL>
int launch_wps(const DevProblem& pb, const double* d_theta, int blocks, int B, const EvalOutputs& out, void* stream) {
    const size_t lds = eval_lds_bytes(pb, INLINE_LL);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int cum_chains = blocks * (WAVE / LPC);  // columns incl. the shadow groups of the last wave
    if constexpr (!INLINE_LL) {
        if (out.cum == nullptr || out.rows == nullptr || out.wstatus == nullptr) return -5;  // see needs_workspace_one()
    }
    hipLaunchKernelGGL((sepaihrd_eval_kernel<LPC, SOLVER, SEPAIHRD_ARITH_FMA, WPS, INLINE_LL>), dim3(blocks), dim3(WAVE), lds,
                       st, pb, d_theta, B, out, cum_chains);
    if (out.ev_after_integrator) (void)hipEventRecord(static_cast<hipEvent_t>(out.ev_after_integrator), st);
    if constexpr (!INLINE_LL) {
        if ((B + WAVE - 1) / WAVE >= ll_serial_min_waves<LPC>() / 3) {
            // lanes per (chain, stream): one from three waves per SIMD on; below that a pair, so that every SIMD holds the same
            // number of waves (32 768 chains: 1536 waves -> 3072)
            const int waves1 = 3 * ((B + WAVE - 1) / WAVE);
            if (LPC % 2 == 0 && waves1 < 3 * 1024) {
                if constexpr (LPC % 2 == 0)
                    hipLaunchKernelGGL((sepaihrd_ll_serial_kernel<LPC, 2>), dim3((2 * B + WAVE - 1) / WAVE, 3), dim3(WAVE), 0, st, pb, B, out, cum_chains);
            } else {
                hipLaunchKernelGGL((sepaihrd_ll_serial_kernel<LPC, 1>), dim3((B + WAVE - 1) / WAVE, 3), dim3(WAVE), 0, st, pb, B, out, cum_chains);
            }
            hipLaunchKernelGGL(sepaihrd_ll_reduce_kernel, dim3((B + 15) / 16), dim3(WAVE), 0, st, pb, B, out, cum_chains, 1);
        } else {
            hipLaunchKernelGGL((sepaihrd_ll_terms_kernel<LPC>), dim3(blocks, (pb.T + LL_DAYS_PER_BLOCK - 1) / LL_DAYS_PER_BLOCK),
                               dim3(WAVE * LL_DAYS_PER_BLOCK), 0, st, pb, B, out, cum_chains);
            hipLaunchKernelGGL(sepaihrd_ll_reduce_kernel, dim3((B + 15) / 16), dim3(WAVE), 0, st, pb, B, out, cum_chains, pb.T);
        }
    }
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

// Does a launch of B chains park the daily increments in the ctx-owned workspace (cum / rows / wstatus)?  The ONE
// place that decides it: launch_one below takes the same branches, and the C ABI sizes the workspace from this.
template <int LPC, int SOLVER>
int needs_workspace_one(const DevProblem& pb, int B, int force_split) {
    constexpr int CPW = WAVE / LPC;
    const int blocks = (B + CPW - 1) / CPW;
    if (blocks <= 0) return 0;
#if SEPAIHRD_HAVE_WAVE_CHAIN
    if constexpr (LPC == 4) {
        if (wave_chain_wanted(B)) return 1;  // the one-wave-per-chain form parks its increments
    }
#endif
    if constexpr (LPC == 4) {
        // the 16-lane form evaluates the likelihood on consumer waves of the same workgroup: no workspace
        if (lane_split_wanted(pb, B)) return (quad_fused_wanted() && !force_split && quad_fused_lds_bytes(pb) <= QUAD_FUSED_MAX_LDS) ? 0 : 1;
    }
    return (split_pays<LPC, SOLVER>((size_t)blocks) || force_split) ? 1 : 0;
}

template <int LPC, int SOLVER>
int launch_one(const DevProblem& pb, const double* d_theta, int B, const EvalOutputs& out, void* stream) {
    constexpr int CPW = WAVE / LPC;
    const int blocks = (B + CPW - 1) / CPW;
    if (blocks <= 0) return 0;
#if SEPAIHRD_HAVE_WAVE_CHAIN
    if constexpr (LPC == 4) {
        if (wave_chain_wanted(B)) return launch_wave_chain<SOLVER>(pb, d_theta, B, out, stream);
    }
#endif
    if constexpr (LPC == 4) {
        if (lane_split_wanted(pb, B)) return launch_quad<SOLVER>(pb, d_theta, B, out, stream);
    }
    // 1024 SIMDs: up to one wave per SIMD the chip is not full and the separate likelihood pass wins
    if (split_pays<LPC, SOLVER>((size_t)blocks) || out.force_split)
        return launch_wps<LPC, SOLVER, 1, false>(pb, d_theta, blocks, B, out, stream);
    if constexpr (SOLVER == 1 || (SEPAIHRD_ARITH_FMA && (SEPAIHRD_DOPRI5_WPS2 || dopri5_two_waves<LPC>()))) {
        // the two-wave form as soon as some SIMD has to hold two waves (its one-wave sibling cannot: one_wave_per_simd_only)
        if (blocks > 1024) return launch_wps<LPC, SOLVER, 2, true>(pb, d_theta, blocks, B, out, stream);
    }
    return launch_wps<LPC, SOLVER, 1, true>(pb, d_theta, blocks, B, out, stream);
}

template <typename K>
int info_of(K kernel, const DevProblem& pb, int lanes, LaunchInfo* info, const char* name, int ll_form, int block_threads = WAVE,
            size_t lds_bytes = 0) {
    info->likelihood_form = ll_form;
    hipFuncAttributes attr;
    if (hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(kernel)) != hipSuccess) return -3;
    info->vgprs = attr.numRegs;
    info->sgprs = 0;
    info->lds_static = (int)attr.sharedSizeBytes;
    info->scratch = (int)attr.localSizeBytes;
    int nb = 0;
    if (lds_bytes == 0) lds_bytes = eval_lds_bytes(pb, ll_form == LL_FORM_INLINE);
    info->lds_dynamic = (int)lds_bytes;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, block_threads, lds_bytes) != hipSuccess) nb = -1;
    info->max_blocks_per_cu = nb;
    info->lanes_per_chain = lanes;
    info->name = name;
    return 0;
}

template <int LPC, int SOLVER>
int info_one(const DevProblem& pb, int batch, LaunchInfo* info, const char* name) {
    constexpr int CPW = WAVE / LPC;
#if SEPAIHRD_HAVE_WAVE_CHAIN
    if constexpr (LPC == 4) {
        if (batch > 0 && wave_chain_wanted(batch))
            return info_of(&sepaihrd_eval_wave_kernel<SOLVER>, pb, WAVE, info, "sepaihrd_eval_wave_kernel[fma]", LL_FORM_SEPARATE_PASS, WAVE, wave_chain_lds_bytes(pb));
    }
#endif
    if constexpr (LPC == 4) {
        if (batch > 0 && lane_split_wanted(pb, batch))
            return (quad_fused_wanted() && quad_fused_lds_bytes(pb) <= QUAD_FUSED_MAX_LDS)
                       ? info_of(&sepaihrd_eval_quad_kernel<SOLVER, SEPAIHRD_ARITH_FMA, true, false>, pb, QUAD_LANES, info, SEP_QUAD_NAME "+ll", LL_FORM_CONSUMER_WAVES, 8 * WAVE, quad_fused_lds_bytes(pb))  // the form an evaluation without trajectories launches
                       : info_of(&sepaihrd_eval_quad_kernel<SOLVER, SEPAIHRD_ARITH_FMA, false>, pb, QUAD_LANES, info, SEP_QUAD_NAME, LL_FORM_SEPARATE_PASS);
    }
    // the same branches as launch_one (batch <= 0: a batch that fills the chip)
    const size_t blocks = batch > 0 ? (size_t)((batch + CPW - 1) / CPW) : (size_t)1 << 20;
    if (split_pays<LPC, SOLVER>(blocks))
        return info_of(&sepaihrd_eval_kernel<LPC, SOLVER, SEPAIHRD_ARITH_FMA, 1, false>, pb, LPC, info, name, LL_FORM_SEPARATE_PASS);
    if constexpr (SOLVER == 1 || (SEPAIHRD_ARITH_FMA && (SEPAIHRD_DOPRI5_WPS2 || dopri5_two_waves<LPC>()))) {
        if (blocks > 1024) return info_of(&sepaihrd_eval_kernel<LPC, SOLVER, SEPAIHRD_ARITH_FMA, 2, true>, pb, LPC, info, name, LL_FORM_INLINE);
    }
    return info_of(&sepaihrd_eval_kernel<LPC, SOLVER, SEPAIHRD_ARITH_FMA, 1, true>, pb, LPC, info, name, LL_FORM_INLINE);
}

#if SEPAIHRD_ARITH_FMA
// the Poisson term's log on caller-given arguments (sepaihrd_device_log_values: the tests compare it with the host's std::log;
// one copy, in the tolerance build's translation unit -- log_pos() is the same explicit operation sequence in both builds)
__global__ __launch_bounds__(WAVE) void log_values_kernel(const double* __restrict__ x, const int n, double* __restrict__ out) {
    stage_log_table(threadIdx.x, WAVE);
    __syncthreads();
    const int i = blockIdx.x * WAVE + threadIdx.x;
    if (i < n) out[i] = log_pos(x[i]);
}
#endif

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
#define SEP_NEEDS_WS launch_needs_workspace_fma
#define SEP_INFO kernel_info_fma
#define SEP_PHASED phase_pass_applied_fma
#define SEP_NAME "sepaihrd_eval_kernel[fma]"
#else
#define SEP_LAUNCH launch_eval_strict
#define SEP_NEEDS_WS launch_needs_workspace_strict
#define SEP_INFO kernel_info_strict
#define SEP_PHASED phase_pass_applied_strict
#define SEP_NAME "sepaihrd_eval_kernel[strict]"
#endif

int SEP_LAUNCH(const DevProblem& pb, int solver, const double* d_theta, int B, const EvalOutputs& out,
               void* stream) {
    SEP_DISPATCH(launch_one, pb, d_theta, B, out, stream)
}
int SEP_NEEDS_WS(const DevProblem& pb, int solver, int B, int force_split) {
    SEP_DISPATCH(needs_workspace_one, pb, B, force_split)
}
int SEP_INFO(const DevProblem& pb, int solver, int batch, LaunchInfo* info) {
    SEP_DISPATCH(info_one, pb, batch, info, SEP_NAME)
}
// csrc/Makefile compiles the host half of this file with -DSEPAIHRD_PHASE_PASS_APPLIED=1 when the device half it embeds came
// out of csrc/phase_pass.py, and with =0 when that step failed and the plain compile was used
#ifndef SEPAIHRD_PHASE_PASS_APPLIED
#define SEPAIHRD_PHASE_PASS_APPLIED 0
#endif
int SEP_PHASED() { return SEPAIHRD_PHASE_PASS_APPLIED; }
#if SEPAIHRD_ARITH_FMA
int poisson_log_values(const double* d_x, int n, double* d_out, void* stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(log_values_kernel, dim3((unsigned)((n + WAVE - 1) / WAVE)), dim3(WAVE), 0, (hipStream_t)stream, d_x, n, d_out);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
#endif

}  // namespace sepaihrd
