// =============================================================================
// csrc/sepaihrd_sampler.hip -- per-chain Adaptive-Metropolis state resident in HBM.
//
// The reference's MetropolisHastingsSampler keeps, per chain, the current state, the proposal
// covariance, its Cholesky factor, the running mean and the whole chain history, and every
// `adaptation_period` iterations recomputes the covariance from that history in two passes
// (src/sir_age_structured/optimizers/MetropolisHastingsSampler.cpp:154-199,295-300).  For one chain
// that is nothing; for thousands of chains in lock-step it is O(C t P^2) work per refresh and C t P
// doubles of history, which belongs next to the likelihood kernel, not on the host.
//
// What stays on the host: the random streams (std::mt19937 + libstdc++ distributions, whose draw
// order depends on the accept test) and the scalar scale adaptation.
// What lives here (one launch each, all chains):
//   propose   prop = applyConstraints(x + scale * L z)                      (:91-102,309)
//   test      accept when log_ratio >= 0 or log(u) < log_ratio              (:318-331)
//   commit    x <- prop where accepted; the state goes to the ring of the newest states and,
//             every thinning-th one, to the sample store                    (:332-340,354-360)
//   rank-one  cov <- (1-g) cov + g d d^T, mean <- mean + g d, d = x_new - mean   (:154-166), queued
//   moments   running sum / Welford mean / centred second moment of every state (oracle::RunningMoments):
//             the covariance refresh (:168-199) costs O(P^2) whatever the length of the chain
//   two-pass  the same refresh as the reference writes it, over a ring that holds the whole history
//   cholesky  lower factor, kept only when the matrix is positive definite  (:240-246,295-300)
// Every sum runs in a fixed order (states ascending, k ascending in the Cholesky recurrences) and this
// file is compiled with -ffp-contract=off, so the numbers are those of the host loop and of the oracle
// bit for bit (tests compare accept traces, samples and covariances).
// =============================================================================
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "sepaihrd_device.h"
#include "sepaihrd_rng.inc"

namespace sepaihrd {
namespace {

// SEPAIHRDParameterManager.cpp:302-313 / :326-343 (same code as the evaluation kernel's)
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
            const double m = (v < lo) ? lo : v;
            return (hi < m) ? hi : m;
        }
        return reflect_bound(v, lo, hi);
    }
    if (mode == 0) return (0.0 < v) ? v : 0.0;
    return fabs(v);
}

// where state `row` of chain c lives: its ring slot, and its place in the sample store (nullptr: not a stored state)
__device__ __forceinline__ double* ring_row(const SamplerState& s, const int c, const int row) {
    return s.hist + ((size_t)c * s.window + (size_t)(row % s.window)) * s.P;
}
__device__ __forceinline__ double* store_row(const SamplerState& s, const int c, const int row) {
    if (s.n_store <= 0 || row % s.thinning != 0 || row / s.thinning >= s.n_store) return nullptr;
    return s.store + ((size_t)c * s.n_store + (size_t)(row / s.thinning)) * s.P;
}

// prop_i = constrain(x_i + scale * sum_{j <= i} L_ij z_j), j ascending
// sum_{j <= i} L(i, j) z_j, j ascending (the reference's row-times-vector order), from the packed columns of the factor.
// The additions are a dependent chain, the loads are not: eight are requested at a time (sixteen gain nothing more) (one at a time the longest row
// of a 62-parameter chain took 62 memory latencies: 27 us per proposal whatever the bandwidth).
__device__ __forceinline__ double chol_row_dot(const double* __restrict__ Lc, const double* __restrict__ zc, const int i, const int P) {
    double sum = 0.0;
    int j = 0, off = 0;
    for (; j + 8 <= i + 1; j += 8) {
        double l[8], zz[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            l[u] = Lc[off + (i - (j + u))];
            zz[u] = zc[j + u];
            off += P - (j + u);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) sum += l[u] * zz[u];
    }
    for (; j <= i; off += P - j, ++j) sum += Lc[off + (i - j)] * zc[j];
    return sum;
}

// chol_row_dot for two vectors of normals at once (the two continuations of a chain's stream): every element of the factor
// is read ONCE.  Per vector the products and additions are chol_row_dot's, in its order.
__device__ __forceinline__ void chol_row_dot2(const double* __restrict__ Lc, const double* za, const double* zb,
                                              const int i, const int P, double& out_a, double& out_b) {
    double sa = 0.0, sb = 0.0;
    int j = 0, off = 0;
    for (; j + 8 <= i + 1; j += 8) {
        double l[8], a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            l[u] = Lc[off + (i - (j + u))];
            a[u] = za[j + u];
            b[u] = zb[j + u];
            off += P - (j + u);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) { sa += l[u] * a[u]; sb += l[u] * b[u]; }
    }
    for (; j <= i; off += P - j, ++j) { const double l = Lc[off + (i - j)]; sa += l * za[j]; sb += l * zb[j]; }
    out_a = sa;
    out_b = sb;
}

__global__ void mh_propose_kernel(const SamplerState s, const DevProblem pb, const double* z, const double* scale) {
    const int c = blockIdx.x;
    const int P = s.P;
    for (int i = threadIdx.x; i < P; i += blockDim.x) {
        // the factor is stored as its lower triangle, COLUMN by column (column j = L(j..P-1, j) at offset
        // j P - j (j - 1) / 2): for a fixed j the threads i = j .. P-1 read one contiguous run, instead of P different
        // cache lines as with rows (51 -> ~25 us per 4096-chain proposal), and no zeros of the upper triangle travel
        // (the proposal is bound by reading the factors: 126 -> 64 MB per 4096 chains x 62 parameters)
        const double* Lc = s.chol + (size_t)c * P * P;
        const double* zc = z + (size_t)c * P;
        const double sum = chol_row_dot(Lc, zc, i, P);
        const double raw = s.x[(size_t)c * P + i] + scale[c] * sum;
        s.prop[(size_t)c * P + i] = constrain(raw, pb.lower[i], pb.upper[i], pb.has_bounds[i], pb.constraint_mode);
    }
}

// adaptGlobalScale(accepted, step) of one chain (:104-152), the window of the last 1000 flags as a ring with its running sum;
// returns global_scale_ = exp(log_scale_) (glibc's exp, csrc/sepaihrd_rng.inc).  One thread per chain.
__device__ __forceinline__ double adapt_global_scale(const SamplerState& s, const int c, const bool accepted, const int step) {
    if (!s.adapt_scale) return s.scale[c];
    int32_t* meta = s.recent_meta + 4 * (size_t)c;
    uint8_t* ring = s.recent + 1000 * (size_t)c;
    int pos = meta[0], len = meta[1], sum = meta[2];
    if (len == 1000) sum -= ring[pos]; else ++len;   // push_back, pop_front beyond 1000 (:107-110)
    ring[pos] = accepted ? 1 : 0;
    sum += accepted ? 1 : 0;
    pos = (pos + 1) % 1000;
    meta[0] = pos; meta[1] = len; meta[2] = sum;
    const double rate = (double)sum / (double)len;
    double ls = s.log_scale[c];
    const double target = s.target_rate;
    if (len >= 1000 && rate < 0.001) {
        ls -= 0.7;
        meta[3] += 1;
    } else if (rate < 0.02 && len >= 500) {
        double g = 5.0 / sqrt((double)step + 1.0);
        g = (0.3 < g) ? 0.3 : g;                     // std::min(gamma_fast, 0.3)
        ls += g * (0.0 - target);
    } else {
        double g = 1.0 / sqrt((double)step + 1.0);
        g = (0.1 < g) ? 0.1 : g;
        ls += g * ((accepted ? 1.0 : 0.0) - target);
    }
    if (s.scale[c] <= 0.011 && rate > 0.15 && rate < 0.30) ls += 0.01;   // the scale BEFORE this update (:146-148)
    ls = (2.3 < ls) ? 2.3 : ls;                      // std::max(std::min(log_scale_, 2.3), -6.9)
    ls = (ls < -6.9) ? -6.9 : ls;
    const double sc = sepaihrd_rng::glibc_exp(ls);
    s.log_scale[c] = ls;
    s.scale[c] = sc;
    return sc;
}
// An evaluation that FAILED (status 2 odeint's 500 rejections, 3 the attempt budget, 4 the hand-off guard between an
// integrating wave and its likelihood wave) enters the accept test as -1e18 like a throwing objective (safeEvaluate :65-74) and
// would otherwise be indistinguishable from an ordinary rejection: counted here, read with sepaihrd_mh_read_failure_counts.
__device__ __forceinline__ void count_failure(const SamplerState& s, const int32_t st) {
    if (st >= 2 && s.fail_counts != nullptr) atomicAdd(&s.fail_counts[(st > 4 ? 4 : st) - 2], 1u);
}
// what else the device keeps of a test's outcome: the value of every stored sample and the accept trace
__device__ __forceinline__ void record_outcome(const SamplerState& s, const int c, const int row, const bool accepted, const double lp_now) {
    if (s.lp_store != nullptr && s.n_store > 0 && row % s.thinning == 0 && row / s.thinning < s.n_store)
        s.lp_store[(size_t)c * s.n_store + row / s.thinning] = lp_now;
    if (s.trace != nullptr && row >= 1) s.trace[(size_t)(row - 1) * s.C + c] = accepted ? 1 : 0;
}

// The accept test of MetropolisHastingsSampler::optimize (:318-331) for every chain, on the device: the evaluation's
// value goes through safeEvaluate's rule (:65-74; a failed integration, status >= 2, counts as -1e18 like a throwing
// objective), log_ratio = proposed - current, accepted when log_ratio >= 0 (no uniform drawn) or log(u) < log_ratio.
// log(u) comes from the host (the chain's mt19937 stream, drawn and logged while the evaluation ran), as do BOTH outcomes
// of the scale adaptation; the kernel picks.  flags: bit 0 accepted, bit 1 best value of the chain so far, bit 2 the test
// took no uniform (the next normals are the other continuation's).  values = what the test compared (for the host's
// bookkeeping, which runs while the NEXT evaluation does).  Comparisons and one subtraction: no rounding to differ in.
__global__ void mh_accept_kernel(const SamplerState s, const int row, const double* __restrict__ loglik, const int32_t* __restrict__ status,
                                 const double* __restrict__ log_u, const double* __restrict__ scale_reject,
                                 const double* __restrict__ scale_accept, double* lp, double* best_lp, double* scale_sel,
                                 uint8_t* flags, double* values) {
    const int C = s.C;
    int32_t* const accepted = s.accepted;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double v = loglik[c];
    count_failure(s, status[c]);
    if (status[c] >= 2 || isnan(v) || isinf(v)) v = -1e18;
    const double log_ratio = v - lp[c];
    const bool no_uniform = log_ratio >= 0.0;
    const bool acc = no_uniform || (log_u[c] < log_ratio);
    uint8_t f = (acc ? 1 : 0) | (no_uniform ? 4 : 0);
    if (acc) {
        lp[c] = v;
        accepted[c] += 1;
        if (v > best_lp[c]) { best_lp[c] = v; f |= 2; }
    }
    flags[c] = f;
    scale_sel[c] = s.device_scale ? adapt_global_scale(s, c, acc, row) : (acc ? scale_accept[c] : scale_reject[c]);
    values[c] = v;
    record_outcome(s, c, row, acc, lp[c]);
}

// proposal with the normals of the continuation the test took (flags bit 2) and the scale it selected
__global__ void mh_propose_select_kernel(const SamplerState s, const DevProblem pb, const double* z_uniform, const double* z_plain,
                                         const uint8_t* flags, const double* scale) {
    const int c = blockIdx.x;
    const int P = s.P;
    const double* zc = ((flags[c] & 4) ? z_plain : z_uniform) + (size_t)c * P;
    for (int i = threadIdx.x; i < P; i += blockDim.x) {
        const double* Lc = s.chol + (size_t)c * P * P;
        const double sum = chol_row_dot(Lc, zc, i, P);
        const double raw = s.x[(size_t)c * P + i] + scale[c] * sum;
        s.prop[(size_t)c * P + i] = constrain(raw, pb.lower[i], pb.upper[i], pb.has_bounds[i], pb.constraint_mode);
    }
}

// Test, commit and next proposal of one chain in ONE launch (block = chain), for the iterations without a covariance
// refresh in between (all but one per adaptation period): three dependent small kernels cost the device more in
// dispatch gaps than in work.  The arithmetic is that of mh_accept_kernel, mh_commit_kernel and
// mh_propose_select_kernel, in that order.
// Two waves per chain.  What takes the time is L z -- a row is P / 8 dependent round trips to memory -- and it does not
// depend on the test, only WHICH of the two continuations' normals it is formed with does: wave 0 forms it for both (every
// element of the factor read once, chol_row_dot2) while wave 1 runs the test (serial in one lane: the accept rule, the scale
// adaptation with its exponential, the outcome records) and fetches the states the commit needs.  The launch is then as long
// as the longer of the two plus the commit, not their sum (21.5 -> ~14 us at 4096 chains).
constexpr int FUSED_ROWS_PER_LANE = (200 + WAVE - 1) / WAVE;  // P <= 200 (sepaihrd_mh_create)
__global__ __launch_bounds__(2 * WAVE) void mh_test_commit_propose_kernel(const SamplerState s, const DevProblem pb,
        const double* __restrict__ loglik, const int32_t* __restrict__ status, const double* __restrict__ log_u,
        const double* __restrict__ scale_reject, const double* __restrict__ scale_accept, double* lp, double* best_lp, double* scale_sel,
        uint8_t* flags, double* values, const double* z_uniform, const double* z_plain, const int row,
        const double* __restrict__ lz_uniform, const double* __restrict__ lz_plain) {
    __shared__ double xs[200];
    __shared__ double zu[200], zp[200];
    __shared__ double sc_sh;
    __shared__ int f_sh;
    const int c = blockIdx.x;
    const int P = s.P;
    const int lane = threadIdx.x & (WAVE - 1);
    const bool dot_wave = threadIdx.x < WAVE;
    double su[FUSED_ROWS_PER_LANE], sp[FUSED_ROWS_PER_LANE];       // wave 0: L z of both continuations, rows lane + 64 q
    double x_cur[FUSED_ROWS_PER_LANE], x_prop[FUSED_ROWS_PER_LANE];  // wave 1: the chain's state and its proposal
    if (dot_wave && lz_uniform != nullptr) {
        // L z of both continuations was formed while the evaluation ran (mh_lz_kernel, same chol_row_dot2): two loads
#pragma unroll
        for (int q = 0; q < FUSED_ROWS_PER_LANE; ++q) {
            const int i = lane + q * WAVE;
            su[q] = i < P ? lz_uniform[(size_t)c * P + i] : 0.0;
            sp[q] = i < P ? lz_plain[(size_t)c * P + i] : 0.0;
        }
    } else if (dot_wave) {
        for (int i = lane; i < P; i += WAVE) {
            zu[i] = z_uniform[(size_t)c * P + i];
            zp[i] = z_plain[(size_t)c * P + i];
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");  // one wave wrote what it reads: LDS is in order within a wave
        __builtin_amdgcn_wave_barrier();
        const double* Lc = s.chol + (size_t)c * P * P;
#pragma unroll
        for (int q = 0; q < FUSED_ROWS_PER_LANE; ++q) {
            const int i = lane + q * WAVE;
            su[q] = sp[q] = 0.0;
            if (i < P) chol_row_dot2(Lc, zu, zp, i, P, su[q], sp[q]);
        }
    } else {
#pragma unroll
        for (int q = 0; q < FUSED_ROWS_PER_LANE; ++q) {  // requested before the serial part below, consumed after it
            const int i = lane + q * WAVE;
            x_cur[q] = i < P ? s.x[(size_t)c * P + i] : 0.0;
            x_prop[q] = i < P ? s.prop[(size_t)c * P + i] : 0.0;
        }
        if (lane == 0) {
            double v = loglik[c];
            count_failure(s, status[c]);
            if (status[c] >= 2 || isnan(v) || isinf(v)) v = -1e18;
            const double log_ratio = v - lp[c];
            const bool no_uniform = log_ratio >= 0.0;
            const bool acc = no_uniform || (log_u[c] < log_ratio);
            int f = (acc ? 1 : 0) | (no_uniform ? 4 : 0);
            if (acc) {
                lp[c] = v;
                s.accepted[c] += 1;
                if (v > best_lp[c]) { best_lp[c] = v; f |= 2; }
            }
            const double sc = s.device_scale ? adapt_global_scale(s, c, acc, row) : (acc ? scale_accept[c] : scale_reject[c]);
            record_outcome(s, c, row, acc, lp[c]);
            flags[c] = (uint8_t)f;
            scale_sel[c] = sc;
            values[c] = v;
            f_sh = f;
            sc_sh = sc;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int f = f_sh;
        double* const ring = ring_row(s, c, row);
        double* const kept = store_row(s, c, row);
#pragma unroll
        for (int q = 0; q < FUSED_ROWS_PER_LANE; ++q) {
            const int i = lane + q * WAVE;
            if (i < P) {
                const size_t idx = (size_t)c * P + i;
                const double v = (f & 1) ? x_prop[q] : x_cur[q];
                if (f & 1) s.x[idx] = v;
                if (f & 2) s.best[idx] = x_prop[q];
                ring[i] = v;
                if (kept) kept[i] = v;
                xs[i] = v;
            }
        }
    }
    __syncthreads();
    if (dot_wave) {
        const int f = f_sh;
        const double sc = sc_sh;
#pragma unroll
        for (int q = 0; q < FUSED_ROWS_PER_LANE; ++q) {
            const int i = lane + q * WAVE;
            if (i < P) {
                const double sum = (f & 4) ? sp[q] : su[q];
                const double raw = xs[i] + sc * sum;
                s.prop[(size_t)c * P + i] = constrain(raw, pb.lower[i], pb.upper[i], pb.has_bounds[i], pb.constraint_mode);
            }
        }
    }
}

// L z for the normals of BOTH continuations of the next test, ahead of it: the product needs the factor and the normals, not
// the running evaluation, and reading the factor (15.6 KB per chain at P = 62: 64 MB per 4096 chains) is what the fused
// test + commit + proposal launch spent its 18 us on.  Queued behind the draws on the copy stream it runs beside the
// evaluation; the launch between two evaluations is then the test and two loads per row.  One wave per chain, the same
// chol_row_dot2 on the same operands as the fused kernel's first wave: the same bits.
__global__ __launch_bounds__(WAVE) void mh_lz_kernel(const SamplerState s, const double* __restrict__ z_uniform, const double* __restrict__ z_plain,
                                                     double* __restrict__ lz_uniform, double* __restrict__ lz_plain) {
    __shared__ double zu[200], zp[200];
    const int c = blockIdx.x, P = s.P, lane = threadIdx.x;
    for (int i = lane; i < P; i += WAVE) {
        zu[i] = z_uniform[(size_t)c * P + i];
        zp[i] = z_plain[(size_t)c * P + i];
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const double* Lc = s.chol + (size_t)c * P * P;
    for (int i = lane; i < P; i += WAVE) {
        double a, b;
        chol_row_dot2(Lc, zu, zp, i, P, a, b);
        lz_uniform[(size_t)c * P + i] = a;
        lz_plain[(size_t)c * P + i] = b;
    }
}

// accepted_known: the accept byte comes from mh_accept_kernel, which has counted it already
__global__ void mh_commit_kernel(const SamplerState s, const uint8_t* accept, const int row, const int accepted_known) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)s.C * s.P) return;
    const int c = (int)(idx / s.P), i = (int)(idx % s.P);
    double v = s.x[idx];
    // accept byte: bit 0 = the proposal was accepted, bit 1 = it is the chain's best state so far (caller's bookkeeping)
    if (accept != nullptr && (accept[c] & 1)) {
        v = s.prop[idx];
        s.x[idx] = v;
        if (i == 0 && !accepted_known) s.accepted[c] += 1;
    }
    if (accept != nullptr && (accept[c] & 2)) s.best[idx] = s.prop[idx];
    ring_row(s, c, row)[i] = v;
    double* const kept = store_row(s, c, row);
    if (kept) kept[i] = v;
}

// The rank-one updates of updateCovarianceRank1 (:154-166), n of them in one launch: update k reads state rows[k]
// with gammas[k], in order; d uses the mean BEFORE that update.  The reference performs one per iteration (:286-288)
// but reads the result only at a refresh that does not recompute from the history (fewer than P + 10 states) and at
// the end of the run -- every recompute overwrites covariance AND mean -- so the library queues the updates and runs
// them when (if) someone looks: one pass over the C P^2 covariances per READ instead of per iteration (252 MB of
// traffic at 4096 chains x 62 parameters).  Each thread replays the mean recurrence of its i and j next to its
// covariance entry: the operations and their order are those of one update per iteration, so the bits are the same.
__global__ void mh_rank1_catchup_cov_kernel(const SamplerState s, const int32_t* rows, const double* gammas, const int n) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t PP = (size_t)s.P * s.P;
    if (idx >= (size_t)s.C * PP) return;
    const int c = (int)(idx / PP);
    const int i = (int)((idx % PP) / s.P), j = (int)(idx % s.P);
    double mi = s.mean[(size_t)c * s.P + i], mj = s.mean[(size_t)c * s.P + j];
    double cov = s.cov[idx];
    for (int r = 0; r < n; ++r) {
        const double g = gammas[r];
        const double* h = ring_row(s, c, rows[r]);
        const double di = h[i] - mi, dj = h[j] - mj;
        cov = (1.0 - g) * cov + g * (di * dj);
        mi += g * di;
        mj += g * dj;
    }
    s.cov[idx] = cov;
}
__global__ void mh_rank1_catchup_mean_kernel(const SamplerState s, const int32_t* rows, const double* gammas, const int n) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)s.C * s.P) return;
    const int c = (int)(idx / s.P), i = (int)(idx % s.P);
    double m = s.mean[idx];
    for (int r = 0; r < n; ++r) m += gammas[r] * (ring_row(s, c, rows[r])[i] - m);
    s.mean[idx] = m;
}

// States row0 .. row0 + n - 1 of every chain enter its running sums (oracle::RunningMoments::push, one state after
// the other): state r is the (r + 1)-th, so  rn = 1 / (r + 1),  w = r / (r + 1),  d = x - wmean,
//     m2_ij += w * (d_i * d_j)  (j <= i),   wmean_i += d_i * rn,   sum_i += x_i.
// One workgroup per chain.  The deviations d of a chunk of states are formed first -- thread i walks the chunk with
// its own mean recurrence, the only sequential part -- and parked in LDS; then every thread adds the chunk to its
// entries of m2, states ascending.  emit_len > 0: the refresh of recomputeFullCovariance (:175-190) for a history of
// emit_len states follows at once: running mean = sum / len (the reference's own mean, same additions in the same
// order) and cov_ij = scaling * (m2_ij / (len - 1)) + eps [i == j], both triangles written.
constexpr int MOMENT_THREADS = 256;
__global__ __launch_bounds__(MOMENT_THREADS) void mh_moments_catchup_kernel(const SamplerState s, const int row0, const int n,
                                                                            const int chunk_rows, const int emit_len) {
    extern __shared__ double dev[];  // [chunk_rows][P] deviations, then [chunk_rows] weights w
    const int c = blockIdx.x, P = s.P, tid = threadIdx.x;
    double* const wts = dev + (size_t)chunk_rows * P;
    double* const m2 = s.m2 + (size_t)c * P * P;
    for (int base = 0; base < n; base += chunk_rows) {
        const int rows = min(chunk_rows, n - base);
        // the chunk's states into LDS with every thread (coalesced rows), then thread i turns its column into deviations in
        // place: the walk is the only sequential part and should not sit out a global-memory latency per state
        for (int e = tid; e < rows * P; e += MOMENT_THREADS) {
            const int r = e / P, i = e - r * P;
            dev[e] = ring_row(s, c, row0 + base + r)[i];
        }
        __syncthreads();
        for (int i = tid; i < P; i += MOMENT_THREADS) {
            double mean = s.wmean[(size_t)c * P + i], sum = s.sum[(size_t)c * P + i];
            int r = 0;
            for (; r + 4 <= rows; r += 4) {  // the reads and the reciprocals of four states ahead of the recurrence that needs them
                double x[4], rn[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    x[u] = dev[(size_t)(r + u) * P + i];
                    rn[u] = 1.0 / (double)(row0 + base + r + u + 1);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const double d = x[u] - mean;
                    dev[(size_t)(r + u) * P + i] = d;
                    mean += d * rn[u];
                    sum += x[u];
                }
            }
            for (; r < rows; ++r) {
                const int row = row0 + base + r;
                const double x = dev[(size_t)r * P + i];
                const double rn = 1.0 / (double)(row + 1);
                const double d = x - mean;
                dev[(size_t)r * P + i] = d;
                mean += d * rn;
                sum += x;
            }
            s.wmean[(size_t)c * P + i] = mean;
            s.sum[(size_t)c * P + i] = sum;
        }
        for (int r = tid; r < rows; r += MOMENT_THREADS) wts[r] = (double)(row0 + base + r) / (double)(row0 + base + r + 1);
        __syncthreads();
        const bool emit = emit_len > 0 && base + rows >= n;
        const double denom = (double)(emit_len - 1);
        for (int e = tid; e < P * P; e += MOMENT_THREADS) {
            const int i = e / P, j = e - i * P;
            if (j > i) continue;
            double acc = m2[e];
            // states ascending, one addition each: that chain is the recurrence's order.  The products do not depend on it:
            // eight states' LDS reads and products are formed first, so a state costs the chain one addition, not a round trip.
            // (What bounds the kernel after that is LDS bandwidth -- two 8-byte reads per product, nothing kept in registers;
            // batched state reads and four interleaved entries per thread on top changed nothing: 575 against 546 us.)
            int r = 0;
            for (; r + 8 <= rows; r += 8) {
                double t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) t[u] = wts[r + u] * (dev[(size_t)(r + u) * P + i] * dev[(size_t)(r + u) * P + j]);
#pragma unroll
                for (int u = 0; u < 8; ++u) acc += t[u];
            }
            for (; r < rows; ++r) acc += wts[r] * (dev[(size_t)r * P + i] * dev[(size_t)r * P + j]);
            m2[e] = acc;
            if (emit) {
                const double v = s.scaling * (acc / denom) + (i == j ? s.reg_eps : 0.0);
                s.cov[((size_t)c * P + i) * P + j] = v;
                s.cov[((size_t)c * P + j) * P + i] = v;
            }
        }
        __syncthreads();
    }
    if (emit_len > 0) {
        if (n <= 0) {  // nothing was pending: the refresh alone
            const double denom = (double)(emit_len - 1);
            for (int e = tid; e < P * P; e += MOMENT_THREADS) {
                const int i = e / P, j = e - i * P;
                if (j > i) continue;
                const double v = s.scaling * (m2[e] / denom) + (i == j ? s.reg_eps : 0.0);
                s.cov[((size_t)c * P + i) * P + j] = v;
                s.cov[((size_t)c * P + j) * P + i] = v;
            }
        }
        for (int i = tid; i < P; i += MOMENT_THREADS) s.mean[(size_t)c * P + i] = s.sum[(size_t)c * P + i] / (double)emit_len;
    }
}

// SURVEY 8(e)'s per-chain summary record from the sample store: mean_i = (sum_s x_si) / S and the unbiased variance
// sum_s (x_si - mean_i)^2 / (S - 1) over samples first .. n_samples - 1 (two passes, samples ascending), the chain's best
// value and its accepted proposals.  Thread = (chain, parameter).
__global__ void mh_summary_kernel(const SamplerState s, const double* __restrict__ best_lp, const int first, const int n_samples, double* out) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)s.C * s.P) return;
    const int c = (int)(idx / s.P), i = (int)(idx % s.P);
    const int W = 2 * s.P + 2;
    const double* col = s.store + ((size_t)c * s.n_store + first) * s.P + i;
    const int S = n_samples - first;
    double sum = 0.0;
    for (int k = 0; k < S; ++k) sum += col[(size_t)k * s.P];
    const double mean = sum / (double)S;
    double ss = 0.0;
    for (int k = 0; k < S; ++k) {
        const double d = col[(size_t)k * s.P] - mean;
        ss += d * d;
    }
    out[(size_t)c * W + i] = mean;
    out[(size_t)c * W + s.P + i] = S > 1 ? ss / (double)(S - 1) : 0.0;
    if (i == 0) {
        out[(size_t)c * W + 2 * s.P] = best_lp[c];
        out[(size_t)c * W + 2 * s.P + 1] = (double)s.accepted[c];
    }
}

// What a progress report / checkpoint of the run reads (MetropolisHastingsSampler.cpp:363-383,440-469), for the n listed chains,
// gathered into one contiguous block so that it can leave the device on another stream while the run goes on:
//   out[k] = [value, best value, scale, accepted | samples first .. first + count - 1, [count][P] | their values [count]]
// Launched on the sampler's own stream right behind the iteration it reports: the values are that iteration's.
__global__ void mh_snapshot_kernel(const SamplerState s, const double* __restrict__ lp, const double* __restrict__ best_lp,
                                   const int32_t* __restrict__ chains, const int first, const int count, double* out) {
    const int k = blockIdx.x;
    const int c = chains[k];
    const size_t width = 4 + (size_t)count * ((size_t)s.P + 1);
    double* const o = out + (size_t)k * width;
    if (threadIdx.x == 0) {
        o[0] = lp[c];
        o[1] = best_lp[c];
        o[2] = s.scale != nullptr ? s.scale[c] : 1.0;
        o[3] = (double)s.accepted[c];
    }
    const double* src = s.store + ((size_t)c * s.n_store + first) * s.P;
    for (size_t i = threadIdx.x; i < (size_t)count * s.P; i += blockDim.x) o[4 + i] = src[i];
    if (s.lp_store != nullptr)
        for (int i = threadIdx.x; i < count; i += blockDim.x) o[4 + (size_t)count * s.P + i] = s.lp_store[(size_t)c * s.n_store + first + i];
}

// pass 1 of recomputeFullCovariance: mean_i = (sum_s h[s][i]) / len, s ascending
__global__ void mh_full_mean_kernel(const SamplerState s, const int len) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)s.C * s.P) return;
    const int c = (int)(idx / s.P), i = (int)(idx % s.P);
    const double* h = s.hist + (size_t)c * s.window * s.P + i;
    double m = 0.0;
    for (int r = 0; r < len; ++r) m += h[(size_t)r * s.P];
    s.mean[idx] = m / (double)len;
}

// pass 2: acc_ij = sum_s (h[s][i] - mean_i) (h[s][j] - mean_j), s ascending;
// cov_ij = scaling (acc_ij / (len - 1)) + (i == j ? eps : 0).  A thread owns one j and IT rows i:
// each history row is read once per IT rows instead of once per row.
constexpr int IT = 8;
__global__ __launch_bounds__(WAVE) void mh_full_cov_kernel(const SamplerState s, const int len) {
    const int c = blockIdx.x;
    const int i0 = blockIdx.y * IT;
    const int j = blockIdx.z * WAVE + threadIdx.x;
    const int P = s.P;
    const bool live = j < P;
    const double* h = s.hist + (size_t)c * s.window * P;
    const double* m = s.mean + (size_t)c * P;
    const double mj = live ? m[j] : 0.0;
    double mi[IT], acc[IT];
#pragma unroll
    for (int k = 0; k < IT; ++k) { mi[k] = (i0 + k < P) ? m[i0 + k] : 0.0; acc[k] = 0.0; }
    for (int r = 0; r < len; ++r) {
        const double* row = h + (size_t)r * P;
        const double dj = (live ? row[j] : 0.0) - mj;
#pragma unroll
        for (int k = 0; k < IT; ++k) {
            const double di = ((i0 + k < P) ? row[i0 + k] : 0.0) - mi[k];
            acc[k] += di * dj;
        }
    }
    if (!live) return;
    const double denom = (double)(len - 1);
#pragma unroll
    for (int k = 0; k < IT; ++k) {
        const int i = i0 + k;
        if (i < P) s.cov[((size_t)c * P + i) * P + j] = s.scaling * (acc[k] / denom) + (i == j ? s.reg_eps : 0.0);
    }
}

// Lower Cholesky factor of cov (+ diag_add on the diagonal), one workgroup per chain, factor built in
// LDS (packed lower triangle: P <= 200 in 160 KiB).  on_failure: 0 = leave chol untouched ("kept only on success"), 1 = 0.1 I (:242-244).
// n_tries = 2: when cov + diag_add is not positive definite the factor of cov + diag_add2 is tried in the same launch.  The
// adaptation-period step factorises cov (kept on success, :190-197) and then cov + eps I (kept on success, :295-300): the second
// overwrites the first whenever it succeeds, so it is tried FIRST and the plain matrix only if it fails -- the same factor in
// every case with one factorisation instead of two.
__global__ __launch_bounds__(WAVE) void mh_cholesky_kernel(const SamplerState s, const double diag_add, const double diag_add2, const int n_tries,
                                                           const int on_failure) {
    extern __shared__ double L[];  // lower triangle packed row by row: (i, j) at i (i + 1) / 2 + j, j <= i
    __shared__ int ok;
    const int c = blockIdx.x, P = s.P, tid = threadIdx.x;
    const double* A = s.cov + (size_t)c * P * P;
    for (int attempt = 0; attempt < n_tries; ++attempt) {
        const double add = attempt == 0 ? diag_add : diag_add2;
        __syncthreads();
        if (tid == 0) ok = 1;
        // the matrix's lower triangle goes into L first (row-wise reads, eight rows in flight) and is factorised in place: a
        // column step then touches LDS only, instead of one strided global read per row and column
        for (int i0 = 0; i0 < P; i0 += 8) {
            double a[8][(200 + WAVE - 1) / WAVE];
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int q = 0; q < (200 + WAVE - 1) / WAVE; ++q) {
                    const int i = i0 + u, j = tid + q * WAVE;
                    a[u][q] = (i < P && j <= i) ? A[(size_t)i * P + j] : 0.0;
                }
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int q = 0; q < (200 + WAVE - 1) / WAVE; ++q) {
                    const int i = i0 + u, j = tid + q * WAVE;
                    if (i < P && j <= i) L[i * (i + 1) / 2 + j] = a[u][q];
                }
        }
        __syncthreads();
        // v - sum_k L_ik L_jk, k ascending, one subtraction per k: the products do not depend on that chain, eight of them
        // (and their LDS reads) are formed ahead of the subtractions that consume them
        auto minus_dot = [&](double v, const double* __restrict__ ra, const double* __restrict__ rb, const int len) -> double {
            int k = 0;
            for (; k + 8 <= len; k += 8) {
                double t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) t[u] = ra[k + u] * rb[k + u];
#pragma unroll
                for (int u = 0; u < 8; ++u) v -= t[u];
            }
            for (; k < len; ++k) v -= ra[k] * rb[k];
            return v;
        };
        for (int j = 0; j < P; ++j) {
            const int rj = j * (j + 1) / 2;
            if (tid == (j % WAVE)) {
                const double d = minus_dot(L[rj + j] + add, L + rj, L + rj, j);
                if (!(d > 0.0)) ok = 0;
                else L[rj + j] = sqrt(d);
            }
            __syncthreads();
            if (!ok) break;
            const double ljj = L[rj + j];
            for (int i = j + 1 + tid; i < P; i += WAVE) {
                const int ri = i * (i + 1) / 2;
                L[ri + j] = minus_dot(L[ri + j], L + ri, L + rj, j) / ljj;
            }
            __syncthreads();
        }
        if (ok) break;
    }
    double* dst = s.chol + (size_t)c * P * P;  // packed columns of the lower triangle (see mh_propose_kernel)
    if (ok || on_failure == 1) {
        for (int j = 0, off = 0; j < P; off += P - j, ++j)
            for (int i = j + tid; i < P; i += WAVE) dst[off + (i - j)] = ok ? L[i * (i + 1) / 2 + j] : (i == j ? 0.1 : 0.0);
    }
}

// ---- the chains' random streams on the device (csrc/sepaihrd_rng.inc has the why and the arithmetic) ----
constexpr int MT_N = 624;

// std::mt19937::seed(value): x[0] = value, x[i] = 1812433253 (x[i-1] ^ (x[i-1] >> 30)) + i; the first output twists
__global__ void mh_seed_kernel(const SamplerState s, const uint32_t seed0) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= s.C) return;
    uint32_t* g = s.mt + (size_t)c * MT_N;
    uint32_t x = seed0 + (uint32_t)c;
    g[0] = x;
    for (int i = 1; i < MT_N; ++i) {
        x = 1812433253u * (x ^ (x >> 30)) + (uint32_t)i;
        g[i] = x;
    }
    s.mt_idx[c] = MT_N;
    s.mt_used[2 * c] = 0;
    s.mt_used[2 * c + 1] = 0;
}

// One wave per chain.  The state lives in LDS while the wave works: st[a] the state the stream is in, st[a ^ 1] the one
// after it (the twist: word i from words i, i + 1 and the word 397 ahead -- three runs of independent words and the last).
// Polar method in parallel: an attempt ALWAYS takes two canonicals, so attempt k of a draw reads canonicals 2k, 2k + 1 from
// the draw's first one whatever the attempts before it decided; the lanes evaluate 64 attempts at once, a ballot ranks the
// accepted ones, and the first ceil(P / 2) of them are the distribution's pairs (y mult, x mult) in order -- the values and
// the number of words consumed are those of the sequential loop.
__global__ __launch_bounds__(WAVE) void mh_draw_kernel(const SamplerState s, const uint8_t* __restrict__ flags, const int first,
                                                       double* __restrict__ log_u, double* __restrict__ z_uniform,
                                                       double* __restrict__ z_plain, const int want_normals) {
    using namespace sepaihrd_rng;
    __shared__ uint32_t st[2][MT_N];
    __shared__ int last_lane[2];
    const int c = blockIdx.x, lane = threadIdx.x, P = s.P;
    uint32_t* const g = s.mt + (size_t)c * MT_N;
    for (int i = lane; i < MT_N; i += WAVE) st[0][i] = g[i];
    __syncthreads();
    auto twist = [&](int from) {  // st[from] -> st[from ^ 1]
        const uint32_t* a = st[from];
        uint32_t* b = st[from ^ 1];
        for (int i = lane; i < 227; i += WAVE) b[i] = mt_twist_word(a[i], a[i + 1], a[i + 397]);
        __syncthreads();
        for (int i = 227 + lane; i < 454; i += WAVE) b[i] = mt_twist_word(a[i], a[i + 1], b[i - 227]);
        __syncthreads();
        for (int i = 454 + lane; i < 623; i += WAVE) b[i] = mt_twist_word(a[i], a[i + 1], b[i - 227]);
        __syncthreads();
        if (lane == 0) b[623] = mt_twist_word(a[623], b[0], b[396]);
        __syncthreads();
    };
    // ---- the stream moves by what the previous test's continuation took
    int cur = 0;
    int idx = s.mt_idx[c];
    int consumed = first ? 0 : ((flags[c] & 4) ? s.mt_used[2 * c + 1] : s.mt_used[2 * c]);
    bool moved_on = false;
    while (idx + consumed >= MT_N) {
        twist(cur);
        cur ^= 1;
        consumed -= MT_N - idx;
        idx = 0;
        moved_on = true;
    }
    idx += consumed;
    if (moved_on)
        for (int i = lane; i < MT_N; i += WAVE) g[i] = st[cur][i];
    if (lane == 0) s.mt_idx[c] = idx;
    // ---- look ahead from there (nothing below changes the stored stream)
    int a = cur, apos = idx;  // the state the look-ahead is in and its position in it
    bool have_next = false;
    auto word = [&](int w) -> uint32_t {  // tempered output w words ahead of the look-ahead's position
        const int q = apos + w;
        return mt_temper(q < MT_N ? st[a][q] : st[a ^ 1][q - MT_N]);
    };
    const int npairs = (P + 1) / 2;
    int have[2] = {0, 0};       // pairs found so far: [0] the continuation with the uniform in front, [1] without
    bool done[2] = {first != 0 || !want_normals, !want_normals};
    int used_words[2] = {2, 0};
    double* const zdst[2] = {z_uniform + (size_t)c * P, (first ? z_uniform : z_plain) + (size_t)c * P};
    for (int round = 0;; ++round) {
        while (apos >= MT_N) {  // the look-ahead has left its state
            if (!have_next) twist(a);
            a ^= 1;
            apos -= MT_N;
            have_next = false;
        }
        if (apos + 4 * WAVE + 2 > MT_N && !have_next) { twist(a); have_next = true; }
        if (round == 0 && lane == 0 && !first) log_u[c] = glibc_log(mt_canonical(word(0), word(1)));
        if (done[0] && done[1]) break;
        const uint32_t w0 = word(4 * lane), w1 = word(4 * lane + 1), w2 = word(4 * lane + 2), w3 = word(4 * lane + 3),
                       w4 = word(4 * lane + 4), w5 = word(4 * lane + 5);
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            if (done[v]) continue;  // wave-uniform
            const double c0 = v == 0 ? mt_canonical(w2, w3) : mt_canonical(w0, w1);
            const double c1 = v == 0 ? mt_canonical(w4, w5) : mt_canonical(w2, w3);
            const double x = 2.0 * c0 - 1.0, y = 2.0 * c1 - 1.0;
            const double r2 = x * x + y * y;
            const bool ok = !(r2 > 1.0 || r2 == 0.0);
            const unsigned long long mask = __ballot(ok);
            const int rank = __popcll(mask & ((1ull << lane) - 1ull));
            const int need = npairs - have[v];
            if (ok && rank < need) {
                const double mult = sqrt(-2 * glibc_log(r2) / r2);
                const int i = 2 * (have[v] + rank);
                zdst[v][i] = (y * mult) * 1.0 + 0.0;
                if (i + 1 < P) zdst[v][i + 1] = (x * mult) * 1.0 + 0.0;
                if (rank == need - 1) last_lane[v] = lane;
            }
            const int found = __popcll(mask);
            if (found >= need) {
                __syncthreads();
                used_words[v] = (v == 0 ? 2 : 0) + 4 * WAVE * round + 4 * (last_lane[v] + 1);
                done[v] = true;
            } else {
                have[v] += found;
            }
        }
        apos += 4 * WAVE;
    }
    if (lane == 0) {
        s.mt_used[2 * c] = first ? used_words[1] : used_words[0];
        s.mt_used[2 * c + 1] = used_words[1];
    }
}

}  // namespace

static inline unsigned blocks_for(size_t n, unsigned bs) { return (unsigned)((n + bs - 1) / bs); }

// the device's log / exp on the self-check arguments of csrc/sepaihrd_rng.inc; the arguments go back with the values so that
// the host compares on exactly the doubles the device used.  out: [log args N | log values N | exp args N | exp values N]
__global__ void libm_check_kernel(double* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    constexpr int N = sepaihrd_rng::LIBM_CHECK_N;
    if (i >= N) return;
    const double xl = sepaihrd_rng::libm_check_log_arg(i), xe = sepaihrd_rng::libm_check_exp_arg(i);
    out[i] = xl;
    out[N + i] = sepaihrd_rng::glibc_log(xl);
    out[2 * N + i] = xe;
    out[3 * N + i] = sepaihrd_rng::glibc_exp(xe);
}
int sampler_libm_check_values(double* d_out, void* stream) {
    hipLaunchKernelGGL(libm_check_kernel, dim3(blocks_for(sepaihrd_rng::LIBM_CHECK_N, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), d_out);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
int sampler_libm_check_count() { return sepaihrd_rng::LIBM_CHECK_N; }

int sampler_seed_streams(const SamplerState& s, uint32_t seed0, void* stream) {
    hipLaunchKernelGGL(mh_seed_kernel, dim3(blocks_for((size_t)s.C, 64)), dim3(64), 0, static_cast<hipStream_t>(stream), s, seed0);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
int sampler_draw(const SamplerState& s, const uint8_t* d_flags, int first, double* d_log_u, double* d_z_uniform, double* d_z_plain,
                 int want_normals, void* stream) {
    hipLaunchKernelGGL(mh_draw_kernel, dim3(s.C), dim3(WAVE), 0, static_cast<hipStream_t>(stream), s, d_flags, first, d_log_u, d_z_uniform,
                       d_z_plain, want_normals);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int sampler_propose(const SamplerState& s, const DevProblem& pb, const double* d_z, const double* d_scale, void* stream) {
    hipLaunchKernelGGL(mh_propose_kernel, dim3(s.C), dim3(WAVE), 0, static_cast<hipStream_t>(stream), s, pb, d_z, d_scale);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

// rows of the staged normals that the host drew again (chains whose accept test took the branch the speculative
// draw did not assume): z[chain[k]][:] = rows[k][:]
__global__ void mh_patch_normals_kernel(double* z, const int32_t* chain, const double* rows, const int n_patch, const int P) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)n_patch * P) return;
    const int k = (int)(idx / P), i = (int)(idx % P);
    z[(size_t)chain[k] * P + i] = rows[idx];
}
int sampler_patch_normals(double* d_z, const int32_t* d_chain, const double* d_rows, int n_patch, int P, void* stream) {
    if (n_patch <= 0) return 0;
    hipLaunchKernelGGL(mh_patch_normals_kernel, dim3(blocks_for((size_t)n_patch * P, 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), d_z, d_chain, d_rows, n_patch, P);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int sampler_accept_test(const SamplerState& s, const int row, const double* d_loglik, const int32_t* d_status, const double* d_log_u, const double* d_scale_reject,
                        const double* d_scale_accept, double* d_lp, double* d_best_lp, double* d_scale_sel, uint8_t* d_flags,
                        double* d_values, void* stream) {
    hipLaunchKernelGGL(mh_accept_kernel, dim3(blocks_for((size_t)s.C, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), s, row, d_loglik,
                       d_status, d_log_u, d_scale_reject, d_scale_accept, d_lp, d_best_lp, d_scale_sel, d_flags, d_values);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
int sampler_propose_select(const SamplerState& s, const DevProblem& pb, const double* d_z_uniform, const double* d_z_plain,
                           const uint8_t* d_flags, const double* d_scale, void* stream) {
    hipLaunchKernelGGL(mh_propose_select_kernel, dim3(s.C), dim3(WAVE), 0, static_cast<hipStream_t>(stream), s, pb, d_z_uniform,
                       d_z_plain, d_flags, d_scale);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int sampler_test_commit_propose(const SamplerState& s, const DevProblem& pb, const double* d_loglik, const int32_t* d_status,
                                const double* d_log_u, const double* d_scale_reject, const double* d_scale_accept, double* d_lp,
                                double* d_best_lp, double* d_scale_sel, uint8_t* d_flags, double* d_values, const double* d_z_uniform,
                                const double* d_z_plain, int row, void* stream, const double* d_lz_uniform, const double* d_lz_plain) {
    if (s.P > 200) return -3;
    hipLaunchKernelGGL(mh_test_commit_propose_kernel, dim3(s.C), dim3(2 * WAVE), 0, static_cast<hipStream_t>(stream), s, pb, d_loglik, d_status,
                       d_log_u, d_scale_reject, d_scale_accept, d_lp, d_best_lp, d_scale_sel, d_flags, d_values, d_z_uniform, d_z_plain, row,
                       d_lz_uniform, d_lz_plain);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
int sampler_lz(const SamplerState& s, const double* d_z_uniform, const double* d_z_plain, double* d_lz_uniform, double* d_lz_plain, void* stream) {
    if (s.P > 200) return -3;
    hipLaunchKernelGGL(mh_lz_kernel, dim3(s.C), dim3(WAVE), 0, static_cast<hipStream_t>(stream), s, d_z_uniform, d_z_plain, d_lz_uniform, d_lz_plain);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

// d_accept written by sampler_accept_test (flags) has been counted there; a caller's own accept bytes are counted here
int sampler_commit(const SamplerState& s, const uint8_t* d_accept, int row, void* stream) {
    return sampler_commit_counted(s, d_accept, row, 0, stream);
}
int sampler_commit_counted(const SamplerState& s, const uint8_t* d_accept, int row, int accepted_known, void* stream) {
    hipLaunchKernelGGL(mh_commit_kernel, dim3(blocks_for((size_t)s.C * s.P, 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       s, d_accept, row, accepted_known);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int sampler_rank1_catchup(const SamplerState& s, const int32_t* d_rows, const double* d_gammas, int n, void* stream) {
    if (n <= 0) return 0;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(mh_rank1_catchup_cov_kernel, dim3(blocks_for((size_t)s.C * s.P * s.P, 256)), dim3(256), 0, st, s, d_rows, d_gammas, n);
    hipLaunchKernelGGL(mh_rank1_catchup_mean_kernel, dim3(blocks_for((size_t)s.C * s.P, 256)), dim3(256), 0, st, s, d_rows, d_gammas, n);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int sampler_moments_catchup(const SamplerState& s, int row0, int n, int emit_len, void* stream) {
    if (n <= 0 && emit_len <= 0) return 0;
    // deviations of a chunk of states in LDS: up to 64 KiB of the 160 KiB (P <= 200: at least 40 states per chunk)
    int chunk = (int)((64 * 1024) / ((size_t)(s.P + 1) * sizeof(double)));
    if (chunk > n) chunk = n > 0 ? n : 1;
    const size_t lds = (size_t)chunk * (s.P + 1) * sizeof(double);
    if (lds > 48 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(&mh_moments_catchup_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            64 * 1024) != hipSuccess)
        return -3;
    hipLaunchKernelGGL(mh_moments_catchup_kernel, dim3(s.C), dim3(MOMENT_THREADS), lds, static_cast<hipStream_t>(stream), s, row0, n,
                       chunk, emit_len);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int sampler_summary_records(const SamplerState& s, const double* d_best_lp, int first_sample, int n_samples, double* d_out, void* stream) {
    if (s.n_store <= 0 || first_sample < 0 || n_samples > s.n_store || first_sample >= n_samples) return -1;
    hipLaunchKernelGGL(mh_summary_kernel, dim3(blocks_for((size_t)s.C * s.P, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), s,
                       d_best_lp, first_sample, n_samples, d_out);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int sampler_snapshot(const SamplerState& s, const double* d_lp, const double* d_best_lp, const int32_t* d_chains, int n, int first,
                     int count, double* d_out, void* stream) {
    hipLaunchKernelGGL(mh_snapshot_kernel, dim3(n), dim3(256), 0, static_cast<hipStream_t>(stream), s, d_lp, d_best_lp, d_chains, first, count, d_out);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int sampler_full_covariance(const SamplerState& s, int len, void* stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(mh_full_mean_kernel, dim3(blocks_for((size_t)s.C * s.P, 256)), dim3(256), 0, st, s, len);
    hipLaunchKernelGGL(mh_full_cov_kernel, dim3(s.C, (s.P + IT - 1) / IT, (s.P + WAVE - 1) / WAVE), dim3(WAVE), 0, st, s, len);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

static int launch_cholesky(const SamplerState& s, double diag_add, double diag_add2, int n_tries, int on_failure, void* stream) {
    const size_t lds = (size_t)s.P * (s.P + 1) / 2 * sizeof(double);
    if (lds > 48 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(&mh_cholesky_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess)
        return -3;
    hipLaunchKernelGGL(mh_cholesky_kernel, dim3(s.C), dim3(WAVE), lds, static_cast<hipStream_t>(stream), s, diag_add, diag_add2, n_tries, on_failure);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
int sampler_cholesky(const SamplerState& s, double diag_add, int on_failure, void* stream) {
    return launch_cholesky(s, diag_add, 0.0, 1, on_failure, stream);
}
int sampler_cholesky_refresh(const SamplerState& s, void* stream) {
    return launch_cholesky(s, s.reg_eps, 0.0, 2, 0, stream);
}

}  // namespace sepaihrd
