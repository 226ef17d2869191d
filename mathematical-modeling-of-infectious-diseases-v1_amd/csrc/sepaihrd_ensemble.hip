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
            for (int ser = 0; ser < 3; ++ser) inc[d][ser] = ok ? a.cum[(k * 3 + comp_of[ser]) * a.cum_stride + col] : 0.0;
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
    for (int p = tid; p < a.n_probs; p += BS) {
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
        } else {
            const size_t k = sid - (size_t)n_series_segments;
            a.sero_out[(size_t)p * a.T + k] = r;
        }
    }
}

}  // namespace

int launch_ensemble_summaries(const EnsembleArgs& a, void* stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (a.S <= 0 || a.S_pad < WAVE || (a.S_pad & (a.S_pad - 1)) != 0 || a.S_pad > ENSEMBLE_MAX_SAMPLES || a.S > a.S_pad)
        return -4;
    hipLaunchKernelGGL(ensemble_count_valid_kernel, dim3(1), dim3(256), 0, st, a.wstatus, a.S, a.n_valid);
    const size_t cols = (size_t)a.S_pad * a.lpc;
    hipLaunchKernelGGL(ensemble_series_kernel, dim3((unsigned)((cols + WAVE - 1) / WAVE)), dim3(WAVE), 0, st, a);
    const int n_series_segments = 6 * a.Tp * a.n;
    const bool sero = a.traj != nullptr && a.sero_out != nullptr;
    const int segments = n_series_segments + (sero ? a.T : 0);
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
