// =============================================================================
// csrc/sepaihrd_ensemble.hip -- posterior-ensemble summaries on gfx950.
//
// Second consumer of the integrator: one simulation per stored
// posterior sample, then per-time quantiles across the samples.
//
// Reference behaviour followed (paths under /root/reference):
//   series     src/model/ResultAggregator.cpp:297-345   daily = max(0, X(t) - X(t_prev)) for the
//              output times t >= 0 (previous point = last run-up point or the initial state),
//              cumulative = running sum of the daily values in time order; X in {CumH, CumICU, D}
//   sero       src/model/MetricsCalculator.cpp:199-226  (sum N - sum_a S_a(t)) / sum N, every time
//   Rt         src/model/ReproductionNumberCalculator.cpp:55-171  spectral radius of F V^-1 (below)
//   quantile   src/model/PostCalibrationAnalyser.cpp:303-340  exact sort, pos = q (n - 1),
//              v[floor pos] (1 - frac) + v[floor pos + 1] frac
// The reference feeds the six incidence series through Boost.Accumulators' P^2 estimator
// (ResultAggregator.cpp:226-244, order-dependent, third-party); here every series uses the exact
// sort rule of PostCalibrationAnalyser, as SURVEY 8f prescribes.
//
// Layout: the integrator leaves the increments in cum[T][3][S*lpc] (sample-major columns).  Pass 1
// (one lane per (sample, age)) walks the days and writes every series value into
// vals[segment][S_pad] with the SAMPLE index contiguous, so that pass 2 (one workgroup per segment)
// loads its segment coalesced, sorts it in LDS (bitonic, S_pad <= 16384 doubles = 128 KiB) and
// interpolates the quantiles.  Samples whose integration failed are +inf and sort to the end; the
// quantile positions use the count of valid samples.
// Compiled with -ffp-contract=off: the interpolation is the CPU build's operation sequence.
// =============================================================================
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include <cstring>
#include <vector>

#include <rocprim/device/device_segmented_radix_sort.hpp>

#include "sepaihrd_device.h"

namespace sepaihrd {
namespace {

__global__ __launch_bounds__(256) void ensemble_count_valid_kernel(const int32_t* wstatus, int S, int32_t* n_valid) {
    __shared__ int part[256];
    int c = 0;
    for (int s = threadIdx.x; s < S; s += 256) c += (wstatus[s] == 0);
    part[threadIdx.x] = c;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) *n_valid = part[0];
}

// Pass 1: lane = (sample, age) column of the integrator workspace.
__global__ __launch_bounds__(WAVE) void ensemble_series_kernel(const EnsembleArgs a) {
    const size_t col = (size_t)blockIdx.x * WAVE + threadIdx.x;
    const int s = (int)(col / a.lpc);
    const int age = (int)(col % a.lpc);
    if (s >= a.S_pad || age >= a.n) return;
    const bool ok = s < a.S && a.wstatus[s] == 0;
    const double inf = INFINITY;
    const size_t seg_stride = (size_t)a.S_pad;
    // series order: daily H, daily ICU, daily D, cumulative H, cumulative ICU, cumulative D
    // cum rows are D, CumH, CumICU
    const int comp_of[3] = {1, 2, 0};
    double run[3] = {0.0, 0.0, 0.0};
    // the running sums are a dependent chain, the loads are not: 8 days are requested at a time
    constexpr int DAYS = 8;
    for (int t0 = 0; t0 < a.Tp; t0 += DAYS) {
        double inc[DAYS][3];
#pragma unroll
        for (int d = 0; d < DAYS; ++d) {
            const size_t k = (size_t)(a.runup_offset + (t0 + d < a.Tp ? t0 + d : a.Tp - 1));
#pragma unroll
            for (int ser = 0; ser < 3; ++ser) inc[d][ser] = ok ? a.cum[cum_index(a.T, col, k, comp_of[ser])] : 0.0;
        }
#pragma unroll
        for (int d = 0; d < DAYS; ++d) {
            const int t = t0 + d;
            if (t >= a.Tp) break;
#pragma unroll
            for (int ser = 0; ser < 3; ++ser) {
                double daily = inf, cumulative = inf;
                if (ok) {
                    daily = (0.0 < inc[d][ser]) ? inc[d][ser] : 0.0;  // std::max(0.0, cur - prev)
                    run[ser] = (t == 0) ? daily : run[ser] + daily;   // row(t) = row(t-1) + daily.row(t)
                    cumulative = run[ser];
                }
                a.vals[(((size_t)ser * a.Tp + t) * a.n + age) * seg_stride + s] = daily;
                a.vals[(((size_t)(ser + 3) * a.Tp + t) * a.n + age) * seg_stride + s] = cumulative;
            }
        }
    }
    if (a.traj != nullptr && a.sero_out != nullptr && age == 0) {
        double* sero = a.vals + (size_t)6 * a.Tp * a.n * seg_stride;
        for (int k = 0; k < a.T; ++k) {
            double v = inf;
            if (ok) {
                const double* row = a.traj + ((size_t)s * a.T + k) * (NUM_COMP * a.n);  // S block first
                double tot = 0.0;
                for (int j = 0; j < a.n; ++j) tot += row[j];
                v = (a.total_pop - tot) / a.total_pop;
            }
            sero[(size_t)k * seg_stride + s] = v;
        }
    }
}

// SEPAIHRDParameterManager.cpp:302-313 / :326-343 (same code as the evaluation kernel's)
__device__ __forceinline__ double ens_reflect_bound(double value, double minb, double maxb) {
    if (minb >= maxb) return minb;
    const double width = maxb - minb;
    double y = fmod(value - minb, 2.0 * width);
    if (y < 0) y += 2.0 * width;
    if (y <= width) return minb + y;
    return maxb - (y - width);
}
__device__ __forceinline__ double ens_constrain(const DevProblem& pb, const double* th, int p) {
    const double v = th[p];
    double lo = pb.lower[p], hi = pb.upper[p];
    if (pb.has_bounds[p]) {
        if (lo > hi) { const double t = lo; lo = hi; hi = t; }
        if (pb.constraint_mode == 0) {
            const double m = (v < lo) ? lo : v;
            return (hi < m) ? hi : m;
        }
        return ens_reflect_bound(v, lo, hi);
    }
    if (pb.constraint_mode == 0) return (0.0 < v) ? v : 0.0;
    return fabs(v);
}
__device__ __forceinline__ double ens_scalar(const DevProblem& pb, const double* th, int slot) {
    const int src = pb.src_scalar[slot];
    return src >= 0 ? ens_constrain(pb, th, src) : pb.base_scalar[slot];
}
__device__ __forceinline__ double ens_vec(const DevProblem& pb, const double* th, int field, int age) {
    const int src = pb.src_vec[field * pb.lpc + age];
    return src >= 0 ? ens_constrain(pb, th, src) : pb.base_vec[field * pb.lpc + age];
}

// Effective reproduction number of sample s at output time k
// (ReproductionNumberCalculator::calculateRt, src/model/ReproductionNumberCalculator.cpp:55-92,95-171):
// spectral radius of the next-generation matrix K = F V^-1 over the states (E, P, A, I) x age.
// F has entries only in its E rows: F(E_i, {P_j, A_j}) = T_ij, F(E_i, I_j) = theta T_ij with
// T_ij = max(0, beta(t) kappa(t) M_ij a_i h_infec_j S_i / N_j); V is block lower-triangular
// (sigma, gamma_p, p gamma_p, (1-p) gamma_p, gamma_A, gamma_I + h), so V^-1 restricted to the E columns is
// closed form and the non-zero spectrum of K is that of the n x n block
//   K_EE(i, j) = T_ij (1/gamma_p + p_j/gamma_A + theta (1 - p_j)/(gamma_I + h_j)).
// K_EE is non-negative: its spectral radius is its Perron root, taken by power iteration in the
// max-norm (the reference calls Eigen::EigenSolver on the full 4n x 4n matrix: same number to rounding).
constexpr int RT_MAX_AGE = 16;
__global__ __launch_bounds__(WAVE) void ensemble_rt_kernel(const EnsembleArgs a, const DevProblem pb, const double* theta) {
    const size_t idx = (size_t)blockIdx.x * WAVE + threadIdx.x;
    const int s = (int)(idx % a.S_pad);  // sample fastest: the segment row is written coalesced
    const int k = (int)(idx / a.S_pad);
    if (k >= a.T) return;
    double* out = a.vals + ((size_t)a.rt_segment0 + k) * a.S_pad;
    if (!(s < a.S && a.wstatus[s] == 0)) { out[s] = INFINITY; return; }
    const double* th = theta + (size_t)s * pb.P;
    const int n = a.n;
    const double t = pb.times[k];
    int seg = 0;
    for (int j = 0; j < pb.nm; ++j) seg += (pb.mends[j] < t) ? 1 : 0;
    const double beta = (pb.nb > 0) ? ens_scalar(pb, th, SS_SCHEDULE0 + pb.seg_ib[seg]) : ens_scalar(pb, th, SS_BETA);
    const double kappa = ens_scalar(pb, th, SS_SCHEDULE0 + pb.nb + pb.seg_ik[seg]);
    const double theta_i = ens_scalar(pb, th, SS_THETA);
    const double gamma_p = ens_scalar(pb, th, SS_GAMMA_P), gamma_A = ens_scalar(pb, th, SS_GAMMA_A),
                 gamma_I = ens_scalar(pb, th, SS_GAMMA_I);
    const double* row = a.traj + ((size_t)s * a.T + k) * (NUM_COMP * n);  // S block first
    double ci[RT_MAX_AGE], gj[RT_MAX_AGE], v[RT_MAX_AGE], w[RT_MAX_AGE];
    for (int i = 0; i < n; ++i) {
        ci[i] = beta * kappa * ens_vec(pb, th, VF_A, i) * row[i];  // beta kappa a_i S_i
        const double Nj = pb.N[i];
        const double pj = ens_vec(pb, th, VF_P, i), hj = ens_vec(pb, th, VF_H, i);
        const double dwell = 1.0 / gamma_p + pj / gamma_A + theta_i * (1.0 - pj) / (gamma_I + hj);
        gj[i] = (Nj < 1e-9) ? 0.0 : ens_vec(pb, th, VF_H_INFEC, i) / Nj * dwell;
        v[i] = 1.0;
    }
    double lambda = 0.0;
    for (int it = 0; it < 2000; ++it) {
        double m = 0.0;
        for (int i = 0; i < n; ++i) {
            double acc = 0.0;
            for (int j = 0; j < n; ++j) {
                const double tij = ci[i] * pb.Mrow[i * pb.lpc + j] * gj[j];
                acc += ((0.0 < tij) ? tij : 0.0) * v[j];
            }
            w[i] = acc;
            m = (acc > m) ? acc : m;
        }
        if (!(m > 0.0)) { lambda = 0.0; break; }
        for (int i = 0; i < n; ++i) v[i] = w[i] / m;
        const bool done = fabs(m - lambda) <= 1e-15 * m;
        lambda = m;
        if (done) break;
    }
    out[s] = lambda;
}

// Per-sample summary metrics, one thread per sample
// (MetricsCalculator::calculateEssentialMetrics, src/model/MetricsCalculator.cpp:8-170).  Row layout:
//   [0] R0  [1] overall_IFR  [2] overall_attack_rate  [3] peak_hospital  [4] peak_ICU
//   [5] time_to_peak_hospital  [6] time_to_peak_ICU  [7] total_deaths  [8] max_Rt  [9] min_Rt  [10] final_Rt
//   [11] seroprevalence at the output time closest to day 64, then per age: IFR, IHR, IICUR, attack rate
// Quirks kept: cumulative infections = the non-S part of the initial state + sum_t lambda_t S_t dt with
// lambda_t = beta kappa(t) M (P + A + theta I)/N using the CONSTANT beta (no schedule, no a_i, no h_infec) and
// dt = 1 for the first time point (:103-113); max_Rt starts at 0, min_Rt at 1e6 (AnalysisTypes.hpp:26-27);
// peaks move on strict ">" only (:91-98); ratios need more than one infection and are clipped to [0, 1]
// (:139-157).  R0 = spectral radius of F V^-1 with S = N, beta(0), kappa(0) and no clipping of F (:22-52).
constexpr int METRIC_SCALARS = 12;
__global__ __launch_bounds__(WAVE) void ensemble_metrics_kernel(const EnsembleArgs a, const DevProblem pb, const double* theta) {
    const int s = blockIdx.x * WAVE + threadIdx.x;
    if (s >= a.S) return;
    const int n = a.n, width = METRIC_SCALARS + 4 * n;
    double* out = a.metrics_out + (size_t)s * width;
    if (a.wstatus[s] != 0) {
        for (int i = 0; i < width; ++i) out[i] = NAN;
        return;
    }
    const double* th = theta + (size_t)s * pb.P;
    const double beta_c = ens_scalar(pb, th, SS_BETA), theta_i = ens_scalar(pb, th, SS_THETA);
    const double gamma_p = ens_scalar(pb, th, SS_GAMMA_P), gamma_A = ens_scalar(pb, th, SS_GAMMA_A),
                 gamma_I = ens_scalar(pb, th, SS_GAMMA_I);
    const double* traj = a.traj + (size_t)s * a.T * (NUM_COMP * n);
    const double* rt = a.vals + (size_t)a.rt_segment0 * a.S_pad + s;  // Rt(s, k) at stride S_pad
    double total_pop = 0.0;
    for (int i = 0; i < n; ++i) total_pop += pb.N[i];

    // R0
    double r0 = 0.0;
    {
        int seg = 0;
        for (int j = 0; j < pb.nm; ++j) seg += (pb.mends[j] < 0.0) ? 1 : 0;
        const double beta0 = (pb.nb > 0) ? ens_scalar(pb, th, SS_SCHEDULE0 + pb.seg_ib[seg]) : beta_c;
        const double kappa0 = ens_scalar(pb, th, SS_SCHEDULE0 + pb.nb + pb.seg_ik[seg]);
        double ci[RT_MAX_AGE], gj[RT_MAX_AGE], v[RT_MAX_AGE], w[RT_MAX_AGE];
        for (int i = 0; i < n; ++i) {
            ci[i] = beta0 * kappa0 * ens_vec(pb, th, VF_A, i) * pb.N[i];
            const double pj = ens_vec(pb, th, VF_P, i), hj = ens_vec(pb, th, VF_H, i);
            const double dwell = 1.0 / gamma_p + pj / gamma_A + theta_i * (1.0 - pj) / (gamma_I + hj);
            gj[i] = (pb.N[i] < 1e-9) ? 0.0 : ens_vec(pb, th, VF_H_INFEC, i) / pb.N[i] * dwell;
            v[i] = 1.0;
        }
        for (int it = 0; it < 2000; ++it) {
            double m = 0.0;
            for (int i = 0; i < n; ++i) {
                double acc = 0.0;
                for (int j = 0; j < n; ++j) acc += ci[i] * pb.Mrow[i * pb.lpc + j] * gj[j] * v[j];
                w[i] = acc;
                m = (fabs(acc) > m) ? fabs(acc) : m;
            }
            if (!(m > 0.0)) { r0 = 0.0; break; }
            for (int i = 0; i < n; ++i) v[i] = w[i] / m;
            const bool done = fabs(m - r0) <= 1e-15 * m;
            r0 = m;
            if (done) break;
        }
    }

    double cum_inf[RT_MAX_AGE];
    for (int i = 0; i < n; ++i) {
        double c = 0.0;
        for (int comp = 1; comp <= 7; ++comp) c += pb.init_state[comp * pb.lpc + i];  // E0 + P0 + ... + R0
        cum_inf[i] = c;
    }
    int target = 0;
    {
        double best = INFINITY;
        for (int k = 0; k < a.T; ++k) {
            const double d = fabs(pb.times[k] - 64.0);
            if (d < best) { best = d; target = k; }
        }
    }
    double peak_h = 0.0, peak_icu = 0.0, t_peak_h = 0.0, t_peak_icu = 0.0;
    double max_rt = 0.0, min_rt = 1e6, final_rt = 0.0, sero64 = 0.0;
    for (int k = 0; k < a.T; ++k) {
        const double* row = traj + (size_t)k * (NUM_COMP * n);
        const double t = pb.times[k];
        const double dt = (k > 0) ? (t - pb.times[k - 1]) : 1.0;
        const double r = rt[(size_t)k * a.S_pad];
        max_rt = (max_rt < r) ? r : max_rt;
        min_rt = (r < min_rt) ? r : min_rt;
        if (k == a.T - 1) final_rt = r;
        double tot_h = 0.0, tot_icu = 0.0, tot_s = 0.0;
        for (int i = 0; i < n; ++i) { tot_h += row[5 * n + i]; tot_icu += row[6 * n + i]; tot_s += row[i]; }
        if (tot_h > peak_h) { peak_h = tot_h; t_peak_h = t; }
        if (tot_icu > peak_icu) { peak_icu = tot_icu; t_peak_icu = t; }
        int seg = 0;
        for (int j = 0; j < pb.nm; ++j) seg += (pb.mends[j] < t) ? 1 : 0;
        const double kappa = ens_scalar(pb, th, SS_SCHEDULE0 + pb.nb + pb.seg_ik[seg]);
        for (int i = 0; i < n; ++i) {
            double acc = 0.0;
            for (int j = 0; j < n; ++j) {
                const double load = (pb.N[j] > 1e-9) ? (row[2 * n + j] + row[3 * n + j] + theta_i * row[4 * n + j]) / pb.N[j] : 0.0;
                acc += pb.Mrow[i * pb.lpc + j] * load;
            }
            cum_inf[i] += (beta_c * kappa) * acc * row[i] * dt;
        }
        if (k == target) sero64 = (total_pop - tot_s) / total_pop;
    }
    const double* last = traj + (size_t)(a.T - 1) * (NUM_COMP * n);
    double sum_inf = 0.0, sum_deaths = 0.0;
    for (int i = 0; i < n; ++i) {
        const double deaths = last[8 * n + i] - pb.init_state[8 * pb.lpc + i];
        const double hosp = last[9 * n + i] - pb.init_state[9 * pb.lpc + i];
        const double icu = last[10 * n + i] - pb.init_state[10 * pb.lpc + i];
        sum_inf += cum_inf[i];
        sum_deaths += deaths;
        double ifr = 0.0, ihr = 0.0, iicur = 0.0;
        if (cum_inf[i] > 1.0) {
            ifr = fmax(0.0, fmin(deaths / cum_inf[i], 1.0));
            ihr = fmax(0.0, fmin(hosp / cum_inf[i], 1.0));
            iicur = fmax(0.0, fmin(icu / cum_inf[i], 1.0));
        }
        out[METRIC_SCALARS + 4 * i + 0] = ifr;
        out[METRIC_SCALARS + 4 * i + 1] = ihr;
        out[METRIC_SCALARS + 4 * i + 2] = iicur;
        out[METRIC_SCALARS + 4 * i + 3] = (pb.N[i] > 0) ? cum_inf[i] / pb.N[i] : 0.0;
    }
    out[0] = r0;
    out[1] = (sum_inf > 1e-9) ? sum_deaths / sum_inf : 0.0;
    out[2] = sum_inf / total_pop;
    out[3] = peak_h; out[4] = peak_icu; out[5] = t_peak_h; out[6] = t_peak_icu;
    out[7] = sum_deaths;
    out[8] = max_rt; out[9] = min_rt; out[10] = final_rt; out[11] = sero64;
}

// quantile p of sorted segment `sid` (nv valid values first, +inf after them) into its output slot
__device__ __forceinline__ void write_quantile(const EnsembleArgs& a, int n_series_segments, size_t sid, int p, int nv,
                                               const double* seg) {
    double r = NAN;
    if (nv > 0) {
        const double pos = a.probs[p] * (double)(size_t)(nv - 1);
        const size_t idx = (size_t)pos;
        const double frac = pos - (double)idx;
        r = (idx + 1 < (size_t)nv) ? seg[idx] * (1.0 - frac) + seg[idx + 1] * frac : seg[idx];
    }
    if ((int)sid < n_series_segments) {
        const int age = (int)(sid % a.n);
        const int t = (int)((sid / a.n) % a.Tp);
        const int ser = (int)(sid / ((size_t)a.n * a.Tp));
        a.q_out[(((size_t)ser * a.n_probs + p) * a.Tp + t) * a.n + age] = r;
    } else if (a.rt_out != nullptr && (int)sid >= a.rt_segment0) {
        a.rt_out[(size_t)p * a.T + (sid - (size_t)a.rt_segment0)] = r;
    } else {
        const size_t k = sid - (size_t)n_series_segments;
        a.sero_out[(size_t)p * a.T + k] = r;
    }
}

// Pass 2: one workgroup per segment; bitonic sort in LDS, then the interpolated quantiles.
__global__ void ensemble_quantile_kernel(const EnsembleArgs a, const int n_series_segments) {
    extern __shared__ double seg[];
    const int Np = a.S_pad;
    const int tid = threadIdx.x, BS = blockDim.x;
    const size_t sid = blockIdx.x;
    const double* src = a.vals + sid * (size_t)Np;
    for (int i = tid; i < Np; i += BS) seg[i] = src[i];
    __syncthreads();
    for (int k = 2; k <= Np; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < (Np >> 1); i += BS) {
                const int lo = ((i & ~(j - 1)) << 1) | (i & (j - 1));  // index with bit j clear
                const int hi = lo | j;
                const bool up = (lo & k) == 0;
                const double x = seg[lo], y = seg[hi];
                if ((x > y) == up) { seg[lo] = y; seg[hi] = x; }
            }
            __syncthreads();
        }
    }
    const int nv = *a.n_valid;
    for (int p = tid; p < a.n_probs; p += BS) write_quantile(a, n_series_segments, sid, p, nv, seg);
}

// Large ensembles (more samples than fit LDS): the segments of a group were sorted in global memory by the
// library's segmented radix sort; one thread per (segment, probability) interpolates.
__global__ void ensemble_quantile_sorted_kernel(const EnsembleArgs a, const int n_series_segments, const double* sorted,
                                                const int first_segment, const int n_group) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)n_group * a.n_probs) return;
    const int g = (int)(idx / a.n_probs), p = (int)(idx % a.n_probs);
    write_quantile(a, n_series_segments, (size_t)(first_segment + g), p, *a.n_valid, sorted + (size_t)g * a.S_pad);
}

}  // namespace

int launch_ensemble_summaries(const EnsembleArgs& a, void* stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool in_lds = a.S_pad <= ENSEMBLE_MAX_SAMPLES;
    if (a.S <= 0 || a.S_pad < WAVE || a.S > a.S_pad || (in_lds ? (a.S_pad & (a.S_pad - 1)) != 0 : a.S_pad % WAVE != 0)) return -4;
    hipLaunchKernelGGL(ensemble_count_valid_kernel, dim3(1), dim3(256), 0, st, a.wstatus, a.S, a.n_valid);
    const size_t cols = (size_t)a.S_pad * a.lpc;
    hipLaunchKernelGGL(ensemble_series_kernel, dim3((unsigned)((cols + WAVE - 1) / WAVE)), dim3(WAVE), 0, st, a);
    const int n_series_segments = 6 * a.Tp * a.n;
    const bool sero = a.traj != nullptr && a.sero_out != nullptr;
    const bool rt = a.traj != nullptr && a.rt_out != nullptr;
    if (rt && (a.n > RT_MAX_AGE || a.rt_segment0 != n_series_segments + (sero ? a.T : 0))) return -4;
    if (rt) {
        const size_t cells = (size_t)a.S_pad * a.T;
        hipLaunchKernelGGL(ensemble_rt_kernel, dim3((unsigned)((cells + WAVE - 1) / WAVE)), dim3(WAVE), 0, st, a, *a.pb, a.theta);
    }
    if (a.metrics_out != nullptr) {
        if (!rt) return -4;  // the table needs the Rt values
        hipLaunchKernelGGL(ensemble_metrics_kernel, dim3((unsigned)((a.S + WAVE - 1) / WAVE)), dim3(WAVE), 0, st, a, *a.pb, a.theta);
    }
    const int segments = n_series_segments + (sero ? a.T : 0) + (rt ? a.T : 0);
    if (!in_lds) {
        // segments of S_pad doubles sorted in groups by rocPRIM's segmented radix sort (8 passes over the keys)
        // into a scratch buffer, quantiles picked from it; group size bounded by the scratch buffer
        int group = (int)(a.sort_scratch_doubles / (size_t)a.S_pad);
        if (group > segments) group = segments;
        if (group < 1 || (size_t)group * a.S_pad >= (size_t)1 << 31) return -4;
        std::vector<unsigned> offs((size_t)group + 1);
        for (int g = 0; g <= group; ++g) offs[(size_t)g] = (unsigned)((size_t)g * a.S_pad);
        unsigned* d_offs = nullptr;
        if (hipMalloc((void**)&d_offs, offs.size() * sizeof(unsigned)) != hipSuccess) return -3;
        int rc = 0;
        if (hipMemcpyAsync(d_offs, offs.data(), offs.size() * sizeof(unsigned), hipMemcpyHostToDevice, st) != hipSuccess) rc = -3;
        void* tmp = nullptr;
        size_t tmp_bytes = 0;
        for (int first = 0; first < segments && rc == 0; first += group) {
            const int ng = (segments - first < group) ? segments - first : group;
            const double* in = a.vals + (size_t)first * a.S_pad;
            const unsigned size = (unsigned)((size_t)ng * a.S_pad);
            size_t need = 0;
            if (rocprim::segmented_radix_sort_keys(nullptr, need, in, a.sort_scratch, size, (unsigned)ng, d_offs, d_offs + 1, 0, 64,
                                                   st) != hipSuccess) { rc = -3; break; }
            if (need > tmp_bytes) {
                if (tmp) (void)hipFree(tmp);
                tmp = nullptr;
                if (hipMalloc(&tmp, need) != hipSuccess) { rc = -3; break; }
                tmp_bytes = need;
            }
            if (rocprim::segmented_radix_sort_keys(tmp, tmp_bytes, in, a.sort_scratch, size, (unsigned)ng, d_offs, d_offs + 1, 0, 64,
                                                   st) != hipSuccess) { rc = -3; break; }
            const size_t work = (size_t)ng * a.n_probs;
            hipLaunchKernelGGL(ensemble_quantile_sorted_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, st, a,
                               n_series_segments, a.sort_scratch, first, ng);
        }
        (void)hipStreamSynchronize(st);
        if (tmp) (void)hipFree(tmp);
        (void)hipFree(d_offs);
        return (rc == 0 && hipGetLastError() == hipSuccess) ? 0 : -3;
    }
    const int threads = a.S_pad / 2 < 1024 ? a.S_pad / 2 : 1024;
    const size_t lds = (size_t)a.S_pad * sizeof(double);
    if (lds > 48 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(&ensemble_quantile_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess)
        return -3;
    hipLaunchKernelGGL(ensemble_quantile_kernel, dim3(segments), dim3(threads), lds, st, a, n_series_segments);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace sepaihrd
