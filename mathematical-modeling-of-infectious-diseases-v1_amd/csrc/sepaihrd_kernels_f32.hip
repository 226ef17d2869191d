// =============================================================================
// csrc/sepaihrd_kernels_f32.hip -- the fp32-state arm of BASELINE configs[4] ("16 age groups ... fp32 vs fp64
// tolerance sweep"): the same evaluation as sepaihrd_kernels.hip -- theta -> model -> initial state -> adaptive RK
// (Dopri5 FSAL / Cash-Karp) over the output grid -> 3-stream Poisson log-likelihood -- with the ODE STATE, the stage
// derivatives and the model coefficients in fp32, and everything a 24-bit number cannot carry kept out of fp32:
//
//   * the log-likelihood: every Poisson term (fp64 log, log_pos) and every sum is fp64 (SURVEY.md section 7:
//     magnitude ~1e6, differences of O(1) decide an acceptance);
//   * the daily INCREMENTS the likelihood reads.  The reference differences cumulative compartments (D, CumH, CumICU
//     reach 1e4..1e6 while a day adds 0..1e3, SEPAIHRDObjectiveFunction.cpp:191-215), which a 24-bit state cannot
//     resolve.  Nothing reads those compartments (or R) back -- their derivatives depend on A, I, H, ICU only -- so
//     here they are integrated as PER-INTERVAL accumulators: the fp32 slot holds the amount added since the last
//     output (that IS the increment, at full fp32 relative precision), and at every output it is folded into an fp64
//     running total and reset.  The error norm still scales them by the size of the whole compartment
//     (|total + accumulator|), like the reference's |x_i|;
//   * time: t, dt and the output grid stay fp64 (a handful of instructions per attempt; fp32 at t = 1000 resolves 6e-5);
//   * theta, its constraints and the folded per-chain constants are formed in fp64 and rounded once.
//
// What fp32 buys on MI355X: a wave64 fp32 VALU instruction occupies the SIMD for 2 cycles instead of 4 and the
// kernel needs about half the registers, so three to four waves share a SIMD where the fp64 n = 16 kernel runs one.
// What it costs is measured by tools/sweep_c5.py: rounding noise of ~1e-7 per operation
// against tolerances down to 1e-6.  There is no bit-exactness contract for this arm (the reference has no fp32
// path); its test is accuracy against the fp64 kernel at each tolerance.
//
// Mapping: one lane per (chain, age class), LPC = pow2(n) in {4, 8, 16} lanes per chain, block = one wavefront.
// Likelihood inline (no workspace): each lane keeps fp64 partial sums of its own age class over the days and the
// ages are added once at the end (the summation order differs from the reference's day-major order -- allowed
// here, see above).  Contraction is on (-ffp-contract=fast) and the derivatives are written with explicit fmaf().
// =============================================================================
#include <hip/hip_runtime.h>
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include <utility>

#include "sepaihrd_device.h"

namespace sepaihrd {
namespace {

#include "sepaihrd_dev_common.inc"

// ---------------------------------------------------------------------------------- 32-bit cross-lane helpers
template <int CTRL>
__device__ __forceinline__ float dpp_move_f(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xf, 0xf, true));
}
template <int LPC>
__device__ __forceinline__ float group_max_f(float m) {
    if constexpr (LPC >= 2) m = fmaxf(m, dpp_move_f<0xB1>(m));   // quad_perm:[1,0,3,2]
    if constexpr (LPC >= 4) m = fmaxf(m, dpp_move_f<0x4E>(m));   // quad_perm:[2,3,0,1]
    if constexpr (LPC >= 8) m = fmaxf(m, dpp_move_f<0x141>(m));  // row_half_mirror
    if constexpr (LPC >= 16) m = fmaxf(m, dpp_move_f<0x140>(m)); // row_mirror
    return m;
}
// acc += m * (lane LANE of src's 16-lane row) in the lanes of banks BANKS: v_fmac_f32 with its first source through DPP
template <int LANE, int BANKS, bool FIRST>
__device__ __forceinline__ void fmac_row_bcast_f(float& acc, float src, float m) {
    if constexpr (FIRST)
        asm("s_nop 1\n\tv_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:%4"
            : "+v"(acc) : "v"(src), "v"(m), "n"(LANE), "n"(BANKS));
    else
        asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:%4"
            : "+v"(acc) : "v"(src), "v"(m), "n"(LANE), "n"(BANKS));
}
// sum over the LPC lanes of my chain (order free in this arm)
template <int LPC>
__device__ __forceinline__ double group_sum(double v) {
    if constexpr (LPC >= 2) v += dpp_move<0xB1>(v);
    if constexpr (LPC >= 4) v += dpp_move<0x4E>(v);
    if constexpr (LPC >= 8) v += dpp_move<0x141>(v);
    if constexpr (LPC >= 16) v += dpp_move<0x140>(v);
    return v;
}

constexpr int F32_SLOTS = 7;          // per-lane fp64 values parked in LDS (see the kernel)
constexpr int NQ = 4;                 // quadrature slots: R, D, CumH, CumICU (state indices 7..10)
constexpr int ND = NUM_COMP - NQ;     // dynamic compartments S..ICU

// per-lane model record, fp32, constants folded once per evaluation (as the fp64 tolerance build folds them)
template <int LPC>
struct LaneModelF {
    float theta, sigma, gamma_p, gamma_A, gamma_I, gamma_H, gamma_ICU;
    float c_inf;                      // h_infec / N
    float h, icu, d_H, d_ICU, d_comm;
    float r_I, r_H, r_ICU, pg, pi;    // gamma_I + h + d_comm, gamma_H + d_H + icu, gamma_ICU + d_ICU, p gamma_p, gamma_p - p gamma_p
    float m[LPC];                     // a_i M(i, .)
};

// AgeSEPAIHRDModel::computeDerivatives (AgeSEPAIHRDModel.cpp:101-228) for this lane's age class, fp32.
// y = {S, E, P, A, I, H, ICU, accR, accD, accCumH, accCumICU}: the last four only receive.
template <int LPC>
__device__ __forceinline__ void rhs_f(const LaneModelF<LPC>& q, const float (&y)[NUM_COMP], float (&dy)[NUM_COMP], float beta_eff) {
    const float S = y[0], E = y[1], P = y[2], A = y[3], I = y[4], H = y[5], ICU = y[6];
    const float pressure = fmaf(q.theta, I, P + A) * q.c_inf;
    float lambda = 0.0f;
    if constexpr (LPC == 16) {
        [&]<int... J>(std::integer_sequence<int, J...>) {
            ((fmac_row_bcast_f<J, 0xf, J == 0>(lambda, pressure, q.m[J])), ...);
        }(std::make_integer_sequence<int, 16>{});
    } else if constexpr (LPC == 8) {  // two chains per 16-lane row: lanes 0-7 take lane J, lanes 8-15 lane 8 + J
        [&]<int... J>(std::integer_sequence<int, J...>) {
            ((fmac_row_bcast_f<J, 0x3, J == 0>(lambda, pressure, q.m[J]), fmac_row_bcast_f<8 + J, 0xc, false>(lambda, pressure, q.m[J])), ...);
        }(std::make_integer_sequence<int, 8>{});
    } else {
        static_assert(LPC == 4, "fp32 arm: 4, 8 or 16 lanes per chain");
        lambda = q.m[0] * dpp_move_f<0x00>(pressure);
        lambda = fmaf(q.m[1], dpp_move_f<0x55>(pressure), lambda);
        lambda = fmaf(q.m[2], dpp_move_f<0xAA>(pressure), lambda);
        lambda = fmaf(q.m[3], dpp_move_f<0xFF>(pressure), lambda);
    }
    lambda *= beta_eff;
    const float flow_SE = fmaxf(lambda, 0.0f) * S;
    const float flow_IH = q.h * I;        // also d CumH
    const float flow_H_ICU = q.icu * H;   // also d CumICU
    dy[0] = -flow_SE;
    dy[1] = fmaf(-q.sigma, E, flow_SE);
    dy[2] = fmaf(-q.gamma_p, P, q.sigma * E);
    dy[3] = fmaf(-q.gamma_A, A, q.pg * P);
    dy[4] = fmaf(-q.r_I, I, q.pi * P);
    dy[5] = fmaf(-q.r_H, H, flow_IH);
    dy[6] = fmaf(-q.r_ICU, ICU, flow_H_ICU);
    dy[7] = fmaf(q.gamma_ICU, ICU, fmaf(q.gamma_A, A, fmaf(q.gamma_H, H, q.gamma_I * I)));
    dy[8] = fmaf(q.d_H, H, fmaf(q.d_comm, I, q.d_ICU * ICU));
    dy[9] = flow_IH;
    dy[10] = flow_H_ICU;
}

// tableau coefficients rounded to fp32 once (constexpr): the controller's exactness class is fp32 anyway
#define F(x) (static_cast<float>(x))

template <int LPC, int SOLVER, int WPS>
__global__ __launch_bounds__(WAVE, WPS) void sepaihrd_eval_f32_kernel(const DevProblem pb, const double* __restrict__ theta,
                                                                     const int B, const EvalOutputs out) {
    constexpr int CPW = WAVE / LPC;
    // WPS = 1: the build launched for at most one wave per SIMD (<= 1024 workgroups).  It declares an accumulation register
    // it never touches, which takes its allocation past 256 registers: two of its waves do not fit a SIMD and the
    // dispatcher cannot pair them up while other SIMDs idle (4096 chains of the 16-age problem behind a 64-MB kernel:
    // 2.04 -> 3.04 ms with the two-wave build; sepaihrd_kernels.hip one_wave_per_simd_only, tools/probe_dispatch_placement.py)
    if constexpr (WPS == 1) asm volatile("" ::: "a127");
    extern __shared__ __attribute__((aligned(16))) double lds[];
    // no output grid here (the next grid time rides in the observation records): half the fp64 kernels' LDS, so that
    // LDS does not cap the waves per CU below what the registers allow
    double* const lds_rec = lds;                          // [2][64 lanes][2]  LDS-DMA landing zone
    double* const lds_slots = lds_rec + LDS_REC_DOUBLES;  // [7][64 lanes]  per-lane fp64 values touched once per OUTPUT:
                                                          //   totals of R, D, CumH, CumICU; likelihood sums H, ICU, D
    double* const lds_mends = lds_slots + F32_SLOTS * WAVE;  // [nm_pad]
    double* const lds_bk = lds_mends + pb.nm_pad;         // [CPW][nm + 1]
    double* const lds_theta = lds_bk + CPW * (pb.nm + 1);
    const int lane = threadIdx.x;
    double* const slot = lds_slots + lane;                // slot[c * WAVE]
    const int grp = lane / LPC;
    const int age = lane % LPC;
    const long long chain0 = (long long)blockIdx.x * CPW;
    const int chains_here = (B - chain0) < CPW ? (int)(B - chain0) : CPW;
    const bool chain_valid = grp < chains_here;
    const int g = chain_valid ? grp : 0;  // lanes past the end of the batch shadow group 0 and never store
    const long long chain = chain0 + g;
    const int P = pb.P;

    // ---- 1. theta, constrained on the way into LDS (fp64: a bound is a bound)
    {
        const int total = chains_here * P;
        const double* src = theta + chain0 * P;
        for (int idx = lane; idx < total; idx += WAVE) {
            const int p = idx % P;
            lds_theta[idx] = constrain(src[idx], pb.lower[p], pb.upper[p], pb.has_bounds[p], pb.constraint_mode);
        }
    }
    for (int k = lane; k < pb.nm_pad; k += WAVE) lds_mends[k] = pb.mends[k];
    stage_log_table(lane, WAVE);  // the (fp64) Poisson term's log reads its table from LDS
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

    // ---- 2. theta -> model, folded in fp64, rounded once
    LaneModelF<LPC> q;
    const double Ni = pb.N[age];
    {
        const double sigma = scalar_slot(SS_SIGMA), gamma_p = scalar_slot(SS_GAMMA_P), gamma_A = scalar_slot(SS_GAMMA_A),
                     gamma_I = scalar_slot(SS_GAMMA_I), gamma_H = scalar_slot(SS_GAMMA_H), gamma_ICU = scalar_slot(SS_GAMMA_ICU);
        const double a = vec_slot(VF_A), h_infec = vec_slot(VF_H_INFEC), p = vec_slot(VF_P), h = vec_slot(VF_H),
                     icu = vec_slot(VF_ICU), d_H = vec_slot(VF_D_H), d_ICU = vec_slot(VF_D_ICU), d_comm = vec_slot(VF_D_COMM);
        const double inv_N = (Ni > 1e-9) ? (1.0 / Ni) : 0.0;
        q.theta = F(scalar_slot(SS_THETA)); q.sigma = F(sigma); q.gamma_p = F(gamma_p); q.gamma_A = F(gamma_A);
        q.gamma_I = F(gamma_I); q.gamma_H = F(gamma_H); q.gamma_ICU = F(gamma_ICU);
        q.c_inf = F(h_infec * inv_N);
        q.h = F(h); q.icu = F(icu); q.d_H = F(d_H); q.d_ICU = F(d_ICU); q.d_comm = F(d_comm);
        q.r_I = F(gamma_I + h + d_comm); q.r_H = F(gamma_H + d_H + icu); q.r_ICU = F(gamma_ICU + d_ICU);
        q.pg = F(p * gamma_p); q.pi = F(gamma_p - p * gamma_p);
        SEP_UNROLL
        for (int j = 0; j < LPC; ++j) q.m[j] = F(a * pb.Mrow[age * LPC + j]);
    }

    Schedule sch;
    sch.me = lds_mends;
    {
        double* bkv = lds_bk + grp * (pb.nm + 1);
        for (int j = age; j <= pb.nm; j += LPC) {
            const double beta = (pb.nb > 0) ? scalar_slot(SS_SCHEDULE0 + pb.seg_ib[j]) : scalar_slot(SS_BETA);
            const double kappa = scalar_slot(SS_SCHEDULE0 + pb.nb + pb.seg_ik[j]);
            bkv[j] = beta * kappa;
        }
        sch.bkv = bkv;
    }
    __syncthreads();

    int status = 0;
    if (pb.kappa_calibrated) {
        bool neg = false;
        for (int k = 1; k < pb.nk; ++k) neg |= (scalar_slot(SS_SCHEDULE0 + pb.nb + k) < 0.0);
        if (neg) status = 1;
    }
    if (!pb.obs_rows_match && pb.init_mode == 0) status = 1;

    // ---- 3. initial state in fp64 (SEPAIHRDObjectiveFunction.cpp:124-163), then split: S..ICU to fp32, the four
    //         quadrature compartments to fp64 totals with empty fp32 accumulators
    float y[NUM_COMP];
    float totf[NQ];  // size of the whole compartment for the error norm's |x_i|, refreshed at every output
    {
        double x[NUM_COMP];
        SEP_UNROLL
        for (int c = 0; c < NUM_COMP; ++c) x[c] = pb.init_state[c * LPC + age];
        if (pb.init_mode != 1) {
            const double runup_days = scalar_slot(SS_RUNUP_DAYS);
            const double seed_exposed = scalar_slot(SS_SEED_EXPOSED);
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
        SEP_UNROLL
        for (int c = 0; c < ND; ++c) y[c] = F(x[c]);
        SEP_UNROLL
        for (int c = 0; c < NQ; ++c) { slot[c * WAVE] = x[ND + c]; totf[c] = F(x[ND + c]); y[ND + c] = 0.0f; }
        SEP_UNROLL
        for (int c = NQ; c < F32_SLOTS; ++c) slot[c * WAVE] = 0.0;  // likelihood sums of this lane's age class
    }

    int n_acc = 0, n_rej = 0;
    const int T = pb.T;
    const int n_real = pb.n;

    const double* grid_lane = pb.grid + (size_t)age * 4;
    auto request_record = [&](int k) {  // {obs_H, obs_ICU, obs_D, times[k+1]} of output k by LDS-DMA, a step ahead
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
            for (int c = 0; c < ND; ++c) tdst[c * n_real] = (double)y[c];
            SEP_UNROLL
            for (int c = 0; c < NQ; ++c) tdst[(ND + c) * n_real] = slot[c * WAVE];  // folded just before
        }
    };
    // observer at output k for the chains with do_it: the accumulators ARE the increments (cwiseMax(0) applied),
    // Poisson terms and sums in fp64; then fold into the totals and restart the accumulators
    auto observe = [&](bool do_it, int k, double oH, double oI, double oD) {
        auto term = [&](double obs, float inc) -> double {
            const double sim = (double)fmaxf(inc, 0.0f) + 1e-10;
            const double v = obs * log_pos(sim) - sim;
            return (do_it && obs >= 0.0 && isfinite(obs)) ? v : 0.0;
        };
        // one log at a time: interleaved, the three fp64 logs hold ~70 registers at once -- the widest point of the kernel
        slot[4 * WAVE] += term(oH, y[9]);
        __builtin_amdgcn_sched_barrier(0);
        slot[5 * WAVE] += term(oI, y[10]);
        __builtin_amdgcn_sched_barrier(0);
        slot[6 * WAVE] += term(oD, y[8]);
        __builtin_amdgcn_sched_barrier(0);
        if (do_it) {
            SEP_UNROLL
            for (int c = 0; c < NQ; ++c) {
                const double tc = slot[c * WAVE] + (double)y[ND + c];
                slot[c * WAVE] = tc;
                totf[c] = F(tc);
                y[ND + c] = 0.0f;
            }
            store_traj(k);
        }
    };

    // ---- 4. integrate_times(controlled stepper, ..., times, dt_hint, observer); time bookkeeping in fp64
    bool active = (status == 0);
    int k_next = 1;
    double t = pb.times[0];
    double t_next = (T > 1) ? pb.times[1] : t;
    double dt = pb.dt_hint;
    int fails = 0, attempts = 0;
    {
        const double oH = grid_lane[0], oI = grid_lane[1], oD = grid_lane[2];
        observe(active, 0, oH, oI, oD);  // row 0: X(0) - init_state = 0
        if (active && T > 1) request_record(1);
    }
    if (T <= 1) active = false;

    sch.lo = INFINITY; sch.hi = -INFINITY; sch.bk = 0.0;
    float k1[NUM_COMP];
    if (SOLVER == 0) {
        int c0, c1;
        segment_index2(sch, pb.nm_pad, t, t, c0, c1);
        rhs_f<LPC>(q, y, k1, F(sch.bkv[c0]));
    }
    const float eps_abs = F(pb.abs_tol), eps_rel = F(pb.rel_tol);

    while (__ballot(active) != 0ull) {
        const double cur_d = active ? fmin(dt, t_next - t) : 1.0;  // min_abs(dt, t_next - t)
        const float cur = F(cur_d);

        // beta(t) kappa(t) at the stage times (fp64 times against the fp64 breakpoints; cached segment)
        float bks[7];
        {
            double tau[7];
            if (SOLVER == 0) {
                tau[0] = t; tau[1] = t + cur_d * dp::a2; tau[2] = t + cur_d * dp::a3; tau[3] = t + cur_d * dp::a4;
                tau[4] = t + cur_d * dp::a5; tau[5] = t + cur_d; tau[6] = t + cur_d;
            } else {
                tau[0] = t; tau[1] = t + ck::c2 * cur_d; tau[2] = t + ck::c3 * cur_d; tau[3] = t + ck::c4 * cur_d;
                tau[4] = t + ck::c5 * cur_d; tau[5] = t + ck::c6 * cur_d; tau[6] = t + cur_d;
            }
            const double tmin = (SOLVER == 0) ? tau[1] : tau[0];
            const double tmax = (SOLVER == 0) ? tau[6] : tau[4];
            const bool in_seg = (tmin > sch.lo) && (tmax <= sch.hi);
            if (__ballot(active && !in_seg) != 0ull) {
                for (int s = 0; s < 7; ++s) {
                    int ca, cb;
                    segment_index2(sch, pb.nm_pad, tau[s], tau[s], ca, cb);
                    bks[s] = F(sch.bkv[ca]);
                }
                int c_lo, c_hi;
                segment_index2(sch, pb.nm_pad, tmin, tmax, c_lo, c_hi);
                sch.lo = (c_hi > 0) ? sch.me[c_hi - 1] : -INFINITY;
                sch.hi = (c_hi < pb.nm) ? sch.me[c_hi] : INFINITY;
                sch.bk = sch.bkv[c_hi];
            } else {
                const float v = F(sch.bk);
                SEP_UNROLL
                for (int s = 0; s < 7; ++s) bks[s] = v;
            }
        }

        float k2[NUM_COMP], k3[NUM_COMP], k4[NUM_COMP], k5[NUM_COMP], k6[NUM_COMP], k7[NUM_COMP];
        float yt[NUM_COMP], ynew[NUM_COMP], yerr[NUM_COMP];
        // The four quadrature slots feed nothing back, so their stage values are never formed, and their new value
        // (qn) and error estimate (qe) are summed as the stages complete: k2..k6 of those slots die at once (20
        // registers less at the widest point of the loop).
        float qn[NQ], qe[NQ];
        auto quad_init = [&](float wn, float we) {
            SEP_UNROLL for (int c = 0; c < NQ; ++c) { qn[c] = fmaf(wn, k1[ND + c], y[ND + c]); qe[c] = we * k1[ND + c]; }
        };
        auto quad_add = [&](const float (&k)[NUM_COMP], float wn, float we) {
            SEP_UNROLL for (int c = 0; c < NQ; ++c) { qn[c] = fmaf(wn, k[ND + c], qn[c]); qe[c] = fmaf(we, k[ND + c], qe[c]); }
        };
        SEP_UNROLL for (int c = ND; c < NUM_COMP; ++c) yt[c] = 0.0f;  // never read by the right-hand side
        if (SOLVER == 0) {
            quad_init(cur * F(dp::c1), cur * F(dp::dc1));
            { const float f1 = cur * F(dp::b21);
              SEP_UNROLL for (int c = 0; c < ND; ++c) yt[c] = fmaf(f1, k1[c], y[c]);
              rhs_f<LPC>(q, yt, k2, bks[1]); }
            { const float f1 = cur * F(dp::b31), f2 = cur * F(dp::b32);
              SEP_UNROLL for (int c = 0; c < ND; ++c) yt[c] = fmaf(f2, k2[c], fmaf(f1, k1[c], y[c]));
              rhs_f<LPC>(q, yt, k3, bks[2]);
              quad_add(k3, cur * F(dp::c3), cur * F(dp::dc3)); }
            { const float f1 = cur * F(dp::b41), f2 = cur * F(dp::b42), f3 = cur * F(dp::b43);
              SEP_UNROLL for (int c = 0; c < ND; ++c) yt[c] = fmaf(f3, k3[c], fmaf(f2, k2[c], fmaf(f1, k1[c], y[c])));
              rhs_f<LPC>(q, yt, k4, bks[3]);
              quad_add(k4, cur * F(dp::c4), cur * F(dp::dc4)); }
            { const float f1 = cur * F(dp::b51), f2 = cur * F(dp::b52), f3 = cur * F(dp::b53), f4 = cur * F(dp::b54);
              SEP_UNROLL for (int c = 0; c < ND; ++c)
                  yt[c] = fmaf(f4, k4[c], fmaf(f3, k3[c], fmaf(f2, k2[c], fmaf(f1, k1[c], y[c]))));
              rhs_f<LPC>(q, yt, k5, bks[4]);
              quad_add(k5, cur * F(dp::c5), cur * F(dp::dc5)); }
            { const float f1 = cur * F(dp::b61), f2 = cur * F(dp::b62), f3 = cur * F(dp::b63), f4 = cur * F(dp::b64),
                          f5 = cur * F(dp::b65);
              SEP_UNROLL for (int c = 0; c < ND; ++c)
                  yt[c] = fmaf(f5, k5[c], fmaf(f4, k4[c], fmaf(f3, k3[c], fmaf(f2, k2[c], fmaf(f1, k1[c], y[c])))));
              rhs_f<LPC>(q, yt, k6, bks[5]);
              quad_add(k6, cur * F(dp::c6), cur * F(dp::dc6)); }
            { const float f1 = cur * F(dp::c1), f3 = cur * F(dp::c3), f4 = cur * F(dp::c4), f5 = cur * F(dp::c5),
                          f6 = cur * F(dp::c6);
              SEP_UNROLL for (int c = 0; c < ND; ++c)
                  ynew[c] = fmaf(f6, k6[c], fmaf(f5, k5[c], fmaf(f4, k4[c], fmaf(f3, k3[c], fmaf(f1, k1[c], y[c])))));
              SEP_UNROLL for (int c = 0; c < NQ; ++c) ynew[ND + c] = qn[c];
              rhs_f<LPC>(q, ynew, k7, bks[6]); }
            { const float e1 = cur * F(dp::dc1), e3 = cur * F(dp::dc3), e4 = cur * F(dp::dc4), e5 = cur * F(dp::dc5),
                          e6 = cur * F(dp::dc6), e7 = cur * F(dp::dc7);
              SEP_UNROLL for (int c = 0; c < ND; ++c)
                  yerr[c] = fmaf(e7, k7[c], fmaf(e6, k6[c], fmaf(e5, k5[c], fmaf(e4, k4[c], fmaf(e3, k3[c], e1 * k1[c])))));
              SEP_UNROLL for (int c = 0; c < NQ; ++c) yerr[ND + c] = fmaf(e7, k7[ND + c], qe[c]); }
        } else {
            rhs_f<LPC>(q, y, k1, bks[0]);
            quad_init(cur * F(ck::b1), cur * F(ck::db1));
            { const float f1 = cur * F(ck::a21);
              SEP_UNROLL for (int c = 0; c < ND; ++c) yt[c] = fmaf(f1, k1[c], y[c]);
              rhs_f<LPC>(q, yt, k2, bks[1]); }
            { const float f1 = cur * F(ck::a31), f2 = cur * F(ck::a32);
              SEP_UNROLL for (int c = 0; c < ND; ++c) yt[c] = fmaf(f2, k2[c], fmaf(f1, k1[c], y[c]));
              rhs_f<LPC>(q, yt, k3, bks[2]);
              quad_add(k3, cur * F(ck::b3), cur * F(ck::db3)); }
            { const float f1 = cur * F(ck::a41), f2 = cur * F(ck::a42), f3 = cur * F(ck::a43);
              SEP_UNROLL for (int c = 0; c < ND; ++c) yt[c] = fmaf(f3, k3[c], fmaf(f2, k2[c], fmaf(f1, k1[c], y[c])));
              rhs_f<LPC>(q, yt, k4, bks[3]);
              quad_add(k4, cur * F(ck::b4), cur * F(ck::db4)); }
            { const float f1 = cur * F(ck::a51), f2 = cur * F(ck::a52), f3 = cur * F(ck::a53), f4 = cur * F(ck::a54);
              SEP_UNROLL for (int c = 0; c < ND; ++c)
                  yt[c] = fmaf(f4, k4[c], fmaf(f3, k3[c], fmaf(f2, k2[c], fmaf(f1, k1[c], y[c]))));
              rhs_f<LPC>(q, yt, k5, bks[4]);
              quad_add(k5, 0.0f, cur * F(ck::db5)); }
            { const float f1 = cur * F(ck::a61), f2 = cur * F(ck::a62), f3 = cur * F(ck::a63), f4 = cur * F(ck::a64),
                          f5 = cur * F(ck::a65);
              SEP_UNROLL for (int c = 0; c < ND; ++c)
                  yt[c] = fmaf(f5, k5[c], fmaf(f4, k4[c], fmaf(f3, k3[c], fmaf(f2, k2[c], fmaf(f1, k1[c], y[c])))));
              rhs_f<LPC>(q, yt, k6, bks[5]);
              quad_add(k6, cur * F(ck::b6), cur * F(ck::db6)); }
            { const float f1 = cur * F(ck::b1), f3 = cur * F(ck::b3), f4 = cur * F(ck::b4), f6 = cur * F(ck::b6);
              SEP_UNROLL for (int c = 0; c < ND; ++c)
                  ynew[c] = fmaf(f6, k6[c], fmaf(f4, k4[c], fmaf(f3, k3[c], fmaf(f1, k1[c], y[c]))));
              SEP_UNROLL for (int c = 0; c < NQ; ++c) ynew[ND + c] = qn[c]; }
            { const float e1 = cur * F(ck::db1), e3 = cur * F(ck::db3), e4 = cur * F(ck::db4), e5 = cur * F(ck::db5),
                          e6 = cur * F(ck::db6);
              SEP_UNROLL for (int c = 0; c < ND; ++c)
                  yerr[c] = fmaf(e6, k6[c], fmaf(e5, k5[c], fmaf(e4, k4[c], fmaf(e3, k3[c], e1 * k1[c]))));
              SEP_UNROLL for (int c = 0; c < NQ; ++c) yerr[ND + c] = qe[c]; }
        }

        // default_error_checker: err = max_i |yerr_i| / (eps_abs + eps_rel (|x_i| + dt |dxdt_i|)), start-of-step values;
        // x_i of a quadrature compartment = its running total + the accumulator
        float err = 0.0f;
        SEP_UNROLL
        for (int c = 0; c < NUM_COMP; ++c) {
            const float xabs = (c < ND) ? fabsf(y[c]) : fabsf(totf[c - ND] + y[c]);
            const float sc = fmaf(eps_rel, fmaf(cur, fabsf(k1[c]), xabs), eps_abs);
            err = fmaxf(err, fabsf(yerr[c]) * __builtin_amdgcn_rcpf(sc));
        }
        err = group_max_f<LPC>(err);
        const bool reject = err > 1.0f;
        ++attempts;
        // default_step_adjuster: decrease dt *= max(0.9 err^(-1/3), 0.2); increase (err < 0.5): dt *= 0.9 max(5^-5, err)^(-1/5)
        const bool grow_relevant = (dt < pb.max_gap) && (4.5000001 * cur_d > dt);
        const bool need_dec = active && reject;
        const bool need_inc = active && !reject && (err < 0.5f) && grow_relevant;
        double cur_after = cur_d;
        if (__ballot(need_dec || need_inc) != 0ull) {
            const float arg = need_dec ? err : fmaxf(1.0f / 3125.0f, err);
            const float expo = need_dec ? -1.0f / 3.0f : -1.0f / 5.0f;
            const float pw = 0.9f * __builtin_amdgcn_exp2f(expo * __builtin_amdgcn_logf(arg));
            const float f = need_dec ? fmaxf(pw, 0.2f) : pw;
            if (need_dec || need_inc) cur_after = cur_d * (double)f;
        }
        {
            const bool rej = active && reject;
            const bool acc = active && !reject;
            n_rej += rej ? 1 : 0;
            n_acc += acc ? 1 : 0;
            dt = rej ? cur_after : (acc ? fmax(dt, cur_after) : dt);
            if (rej && fails >= 500) { status = 2; active = false; }  // failed_step_checker
            fails = acc ? 0 : fails + (rej ? 1 : 0);
            if (acc) {
                t += cur_d;
                SEP_UNROLL
                for (int c = 0; c < NUM_COMP; ++c) y[c] = ynew[c];
                if (SOLVER == 0) {
                    SEP_UNROLL
                    for (int c = 0; c < NUM_COMP; ++c) k1[c] = k7[c];
                }
            }
            const bool reached = acc && !((t_next - t) > DBL_EPSILON);  // less_with_sign(t, t_next, dt)
            if (__ballot(reached) != 0ull) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the record requested a step ago
                const double2 ra = *reinterpret_cast<const double2*>(lds_rec + 2 * lane);
                const double2 rb = *reinterpret_cast<const double2*>(lds_rec + 2 * WAVE + 2 * lane);
                observe(reached, k_next, ra.x, ra.y, rb.x);
                if (reached) {
                    t = t_next;  // integrate_times re-reads the exact grid time
                    ++k_next;
                    t_next = rb.y;
                    if (k_next >= T) active = false;
                    else request_record(k_next);
                }
            }
            if (active && attempts >= pb.max_attempts) { status = 3; active = false; }
        }
    }

    // ---- 5. total = (hosp + icu) + deaths, ages added once (SEPAIHRDObjectiveFunction.cpp:222-227)
    const double sH = group_sum<LPC>(slot[4 * WAVE]), sI = group_sum<LPC>(slot[5 * WAVE]), sD = group_sum<LPC>(slot[6 * WAVE]);
    if (chain_valid && age == 0) {
        double total = (sH + sI) + sD;
        if (status == 0 && (isnan(total) || isinf(total))) status = 1;
        if (status != 0) total = -DBL_MAX;
        out.loglik[chain] = total;
        if (out.status) out.status[chain] = status;
        if (out.ll_parts) {
            out.ll_parts[3 * chain + 0] = sH;
            out.ll_parts[3 * chain + 1] = sI;
            out.ll_parts[3 * chain + 2] = sD;
        }
        if (out.n_accept) out.n_accept[chain] = n_acc;
        if (out.n_reject) out.n_reject[chain] = n_rej;
    }
}
#undef F

// Two waves per SIMD: the loop holds ~210 registers (seven stage vectors, the model record, the fp64 time and
// schedule scalars).  Measured on the configs[4] workload: a 3-wave budget (168 registers) spills ~40 values to
// scratch inside the loop and is 15 % slower (2.26 M vs 2.66 M evals/s at 1e-6), so two it is.
template <int LPC>
constexpr int f32_waves_per_simd() { return 2; }
inline size_t f32_lds_bytes(const DevProblem& pb) {
    const int cpw = WAVE / pb.lpc;
    return ((size_t)LDS_REC_DOUBLES + (size_t)F32_SLOTS * WAVE + pb.nm_pad + (size_t)cpw * (pb.nm + 1) + (size_t)cpw * pb.P) * sizeof(double);
}

template <int LPC, int SOLVER>
int launch_f32_one(const DevProblem& pb, const double* d_theta, int B, const EvalOutputs& out, void* stream) {
    constexpr int CPW = WAVE / LPC;
    const int blocks = (B + CPW - 1) / CPW;
    if (blocks <= 0) return 0;
    if (blocks <= 1024)
        hipLaunchKernelGGL((sepaihrd_eval_f32_kernel<LPC, SOLVER, 1>), dim3(blocks), dim3(WAVE), f32_lds_bytes(pb), static_cast<hipStream_t>(stream), pb,
                           d_theta, B, out);
    else
        hipLaunchKernelGGL((sepaihrd_eval_f32_kernel<LPC, SOLVER, f32_waves_per_simd<LPC>()>), dim3(blocks), dim3(WAVE),
                           f32_lds_bytes(pb), static_cast<hipStream_t>(stream), pb, d_theta, B, out);
    if (out.ev_after_integrator) (void)hipEventRecord(static_cast<hipEvent_t>(out.ev_after_integrator), static_cast<hipStream_t>(stream));
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

template <int LPC, int SOLVER>
int info_f32_one(const DevProblem& pb, LaunchInfo* info) {
    auto kernel = &sepaihrd_eval_f32_kernel<LPC, SOLVER, f32_waves_per_simd<LPC>()>;
    hipFuncAttributes attr;
    if (hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(kernel)) != hipSuccess) return -3;
    info->vgprs = attr.numRegs;
    info->sgprs = 0;
    info->lds_static = (int)attr.sharedSizeBytes;
    info->lds_dynamic = (int)f32_lds_bytes(pb);
    info->scratch = (int)attr.localSizeBytes;
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, WAVE, f32_lds_bytes(pb)) != hipSuccess) nb = -1;
    info->max_blocks_per_cu = nb;
    info->lanes_per_chain = LPC;
    info->name = "sepaihrd_eval_f32_kernel";
    return 0;
}

}  // namespace

int launch_eval_f32(const DevProblem& pb, int solver, const double* d_theta, int B, const EvalOutputs& out, void* stream) {
    switch (pb.lpc) {
        case 4: return solver == 0 ? launch_f32_one<4, 0>(pb, d_theta, B, out, stream) : launch_f32_one<4, 1>(pb, d_theta, B, out, stream);
        case 8: return solver == 0 ? launch_f32_one<8, 0>(pb, d_theta, B, out, stream) : launch_f32_one<8, 1>(pb, d_theta, B, out, stream);
        case 16: return solver == 0 ? launch_f32_one<16, 0>(pb, d_theta, B, out, stream) : launch_f32_one<16, 1>(pb, d_theta, B, out, stream);
        default: return -4;  // 1 or 2 age classes: not built in fp32
    }
}
int kernel_info_f32(const DevProblem& pb, int solver, LaunchInfo* info) {
    switch (pb.lpc) {
        case 4: return solver == 0 ? info_f32_one<4, 0>(pb, info) : info_f32_one<4, 1>(pb, info);
        case 8: return solver == 0 ? info_f32_one<8, 0>(pb, info) : info_f32_one<8, 1>(pb, info);
        case 16: return solver == 0 ? info_f32_one<16, 0>(pb, info) : info_f32_one<16, 1>(pb, info);
        default: return -4;
    }
}

}  // namespace sepaihrd
