// csrc/sepaihrd_device.h -- structures shared by the C-ABI host code (sepaihrd_capi.cpp)
// and the HIP kernels (sepaihrd_kernels.hip).  Internal: not part of the C ABI.
#pragma once
#include <stdint.h>

namespace sepaihrd {

constexpr int NUM_COMP = 11;          // S,E,P,A,I,H,ICU,R,D,CumH,CumICU
constexpr int NUM_POP_COMP = 9;       // S..D
constexpr int WAVE = 64;              // CDNA wavefront

// scalar slots of a chain's parameter record (order fixed, see build_slot_tables)
enum ScalarSlot {
    SS_BETA = 0, SS_THETA, SS_SIGMA, SS_GAMMA_P, SS_GAMMA_A, SS_GAMMA_I, SS_GAMMA_H, SS_GAMMA_ICU,
    SS_E0_MULT, SS_P0_MULT, SS_A0_MULT, SS_I0_MULT, SS_H0_MULT, SS_ICU0_MULT, SS_R0_MULT, SS_D0_MULT,
    SS_RUNUP_DAYS, SS_SEED_EXPOSED,
    SS_SCHEDULE0  // beta_values[0..nb) then kappa_values[0..nk)
};
// per-age vector fields
enum VecField { VF_A = 0, VF_H_INFEC, VF_P, VF_H, VF_ICU, VF_D_H, VF_D_ICU, VF_D_COMM, VF_COUNT };

// Problem data resident in HBM (uploaded once per ctx).  Every per-age table is padded
// to `lpc` (lanes per chain = n rounded up to a power of two) so that lane `age` can
// index it directly; padded ages have N = 0, zero state, zero rates and NaN observations.
struct DevProblem {
    int32_t n, lpc, T, n_obs, runup_offset, nb, nk, P, ns;
    int32_t constraint_mode, kappa_calibrated, max_attempts, obs_rows_match;
    int32_t init_mode;  // 0: from theta (objective), 1: problem.initial_state as given (ensemble), 2: multipliers always (FD gradient)
    int32_t form;       // SEPAIHRD_FORM_*: which form of the integrator a launch uses (0 = by batch size); same bits either way
    double abs_tol, rel_tol, dt_hint, max_gap;
    const double* times;         // [T]
    // per output point k and lane (age): {obs_H, obs_ICU, obs_D, times[k+1]} -- 32 bytes, fetched by
    // two 16-byte LDS-DMA loads one RK step ahead of use.  NaN observations for k < runup_offset
    // and for padded ages; times[T] := times[T-1].
    const double* grid;          // [T][lpc][4]
    const double* lower;         // [P]
    const double* upper;         // [P]
    const int32_t* has_bounds;   // [P]
    const int32_t* src_scalar;   // [ns] theta index feeding the slot, or -1
    const double* base_scalar;   // [ns]
    const int32_t* src_vec;      // [VF_COUNT][lpc]
    const double* base_vec;      // [VF_COUNT][lpc]
    const double* N;             // [lpc]
    const double* age_fraction;  // [lpc]  N_i / sum(N)
    const double* Mrow;          // [lpc][lpc]  Mrow[i*lpc+j] = M(i,j)
    const double* init_state;    // [11][lpc]
    const double* beta_ends;     // [nb]
    const double* kappa_ends;    // [nk]
    // merged schedule: union of the beta and kappa end times, sorted; segment j = (mends[j-1], mends[j]],
    // segment nm = (mends[nm-1], +inf).  seg_ib / seg_ik: value index of each schedule on segment j.
    int32_t nm, nm_pad;          // nm_pad = nm rounded up to even (padded with +inf)
    const double* mends;         // [nm_pad]
    const int32_t* seg_ib;       // [nm + 1]
    const int32_t* seg_ik;       // [nm + 1]
};

struct EvalOutputs {
    double* loglik;     // [B]
    int32_t* status;    // [B] or null
    int32_t* n_accept;  // [B] or null
    int32_t* n_reject;  // [B] or null
    double* ll_parts;   // [B][3] or null
    double* traj;       // [B][T][11][n] or null
    // workspace of the likelihood pass (ctx-owned):
    //   cum  [T][3][Bc*lpc]  daily increments of D, CumH, CumICU of every lane (coalesced rows)
    //   rows [T][3][Bc]      per-day row sums of the three streams
    //   wstatus [Bc]         integrator status per chain (always written, `status` may be null)
    double* cum;
    double* rows;
    int32_t* wstatus;
    void* ev_after_integrator;  // optional hipEvent_t recorded between the integrator kernel and the likelihood pass
    int32_t force_split;        // always park the increments in `cum` (ensemble summaries read them)
};
inline size_t workspace_cum_doubles(const DevProblem& pb, size_t chains) { return (size_t)pb.T * 3 * chains * pb.lpc; }
// Layout of the parked daily increments (D, CumH, CumICU of every (chain, age) column and output day): WAVE-major,
// cum[column / 64][T][3][64].  A wave of the integrator writes its own contiguous T x 1.5 KB, day after day -- one or two
// pages for the whole run.  (Through round 4's first builds it was cum[T][3][columns]: successive days of a wave lay
// 24 B x columns apart, every output touched three new pages, and the kernel ran 1.6 x slower behind any other kernel
// that had pushed the page-table lines out of the cache: 16 384 chains, 0.91 -> 1.46 ms, tools/probe_dispatch_placement.py.)
constexpr size_t CUM_ROW_DOUBLES = 3 * WAVE;  // one day of one wave
#if defined(__HIPCC__)
__host__ __device__
#endif
inline size_t cum_index(int T, size_t column, int day, int comp) {
    return ((column / WAVE) * (size_t)T + (size_t)day) * CUM_ROW_DOUBLES + (size_t)comp * WAVE + (column % WAVE);
}
inline size_t workspace_rows_doubles(const DevProblem& pb, size_t chains) { return (size_t)pb.T * 3 * chains; }

// sepaihrd_kernel_info::likelihood_form (include/sepaihrd_hip.h: SEPAIHRD_LL_*)
constexpr int LL_FORM_INLINE = 0, LL_FORM_SEPARATE_PASS = 1, LL_FORM_CONSUMER_WAVES = 2;
struct LaunchInfo {
    int vgprs, sgprs, lds_static, lds_dynamic, scratch, max_blocks_per_cu;
    int lanes_per_chain;  // of the kernel a launch of the given batch uses
    int likelihood_form;  // LL_FORM_*: inline in the integrator, separate pass over parked increments, consumer waves
    const char* name;
};

// implemented twice, once per arithmetic mode (separate translation units of the same source)
int launch_eval_strict(const DevProblem& pb, int solver, const double* d_theta, int B,
                       const EvalOutputs& out, void* stream);
int launch_eval_fma(const DevProblem& pb, int solver, const double* d_theta, int B,
                    const EvalOutputs& out, void* stream);
// 1 when a launch of B chains parks its daily increments in the ctx-owned workspace (EvalOutputs::cum / rows /
// wstatus must then be sized for it), 0 when the likelihood is evaluated inside the integrator, < 0 on error.
// Decided in the kernel translation unit, next to the launch code that takes the same branches.
int launch_needs_workspace_strict(const DevProblem& pb, int solver, int B, int force_split);
int launch_needs_workspace_fma(const DevProblem& pb, int solver, int B, int force_split);
// fp32-state arm (csrc/sepaihrd_kernels_f32.hip): likelihood inline, no workspace; -4 for lanes-per-chain it is not built for
int launch_eval_f32(const DevProblem& pb, int solver, const double* d_theta, int B, const EvalOutputs& out, void* stream);
int kernel_info_f32(const DevProblem& pb, int solver, LaunchInfo* info);
// batch <= 0: the large-batch kernel
int kernel_info_strict(const DevProblem& pb, int solver, int batch, LaunchInfo* info);
int kernel_info_fma(const DevProblem& pb, int solver, int batch, LaunchInfo* info);
// 1 when the translation unit's device code went through csrc/phase_pass.py (csrc/Makefile), 0 for the plain compile
int phase_pass_applied_strict();
int phase_pass_applied_fma();
// the Poisson term's log (csrc/sepaihrd_dev_common.inc log_pos) on n device-resident arguments
int poisson_log_values(const double* d_x, int n, double* d_out, void* stream);

// ---- posterior ensemble summaries (csrc/sepaihrd_ensemble.hip) ----
constexpr int ENSEMBLE_MAX_SAMPLES = 16384;  // one sorted segment lives in LDS (128 KiB of 160 KiB)
struct EnsembleArgs {
    int S, S_pad;             // samples; stride of a segment: a power of two >= 64 (LDS sort) or a multiple of 64
    int lpc, n, T, Tp;        // lanes per chain, ages, output times, output times with t >= 0
    int runup_offset;         // index of the first output time >= 0
    int n_probs;
    size_t cum_stride;        // columns of the integrator workspace (launch chains * lpc)
    const double* cum;        // daily increments of D, CumH, CumICU: see cum_index()
    const int32_t* wstatus;   // [S] integrator status
    const double* traj;       // [S][T][11][n] or null (seroprevalence needs S(t))
    double total_pop;
    double* vals;             // [(6 Tp n) + T (sero) + T (Rt)][S_pad] series values, one sortable segment per row
    const double* probs;      // [n_probs] device
    double* q_out;            // [6][n_probs][Tp][n] device
    double* sero_out;         // [n_probs][T] device or null
    int32_t* n_valid;         // [1] device
    // effective reproduction number (needs traj): one more block of T segments starting at rt_segment0
    double* rt_out;           // [n_probs][T] device or null
    int rt_segment0;
    const DevProblem* pb;     // host pointer to the ctx's problem (kernel argument by value)
    const double* theta;      // [S][P] device, the samples
    double* metrics_out;      // [S][12 + 4 n] per-sample summary metrics, device or null (needs rt_out)
    // ensembles beyond the LDS sort (S_pad > ENSEMBLE_MAX_SAMPLES): scratch for groups of globally sorted segments
    double* sort_scratch;
    size_t sort_scratch_doubles;
};
int launch_ensemble_summaries(const EnsembleArgs& a, void* stream);

inline int lanes_per_chain(int n) {
    int l = 1;
    while (l < n) l <<= 1;
    return l;
}
// LDS carve (one wavefront per block), in doubles:
//   [0, T_pad)                    output times (T rounded up to even)
//   [.., + 256)                   observation-record landing zone (inline-likelihood build)
//   [.., + nm_pad)                merged period end times (+inf padded to an even count)
//   [.., + cpw*(nm+1))            per-chain beta*kappa of every merged segment
//   [.., + cpw*P)                 constrained theta of the wave's chains (prologue only)
constexpr int MAX_TIMES = 12288;  // output grid staged in LDS (96 KiB at the cap)
constexpr int LOG_TABLE_LDS_BYTES = 128 * 2 * 8;  // lds_log_table (csrc/sepaihrd_dev_common.inc): static LDS of every kernel that takes logs
constexpr int LDS_REC_DOUBLES = 2 * WAVE * 2;  // LDS-DMA landing zone of the inline-likelihood build
#ifndef SEPAIHRD_SPLIT_LL_MAX_BLOCKS
#define SEPAIHRD_SPLIT_LL_MAX_BLOCKS 1024
#endif
constexpr int SPLIT_LL_MAX_BLOCKS = SEPAIHRD_SPLIT_LL_MAX_BLOCKS;  // waves up to which the separate likelihood pass is always used
// Which form of the likelihood a launch uses is decided in the kernel translation unit (split_pays in
// csrc/sepaihrd_kernels.hip): up to one wave per SIMD the chip is not full and the separate pass always wins; beyond that it
// wins only where the integrator WITHOUT the inline logs fits two waves per SIMD (Dopri5 in fma arithmetic up to 4 age
// classes: 256 registers, 15.1 M vs 13.9 M evals/s at 32 768 chains) -- elsewhere the parked increments are pure traffic.
#if defined(__HIPCC__)
#define SEP_HOST_DEVICE __host__ __device__
#else
#define SEP_HOST_DEVICE
#endif
SEP_HOST_DEVICE inline int times_pad(const DevProblem& pb) { return (pb.T + 1) & ~1; }
// dynamic LDS of one wave of sepaihrd_eval_kernel; inline_ll: the builds that evaluate the likelihood inside the integrator
// keep no output grid in LDS (the next grid time rides in the observation record)
inline size_t eval_lds_bytes(const DevProblem& pb, bool inline_ll = false) {
    const int cpw = WAVE / pb.lpc;
    // ... + 6 doubles per lane: the inline likelihood's state in the two-waves-per-SIMD builds (LL_IN_LDS, sepaihrd_kernels.hip)
    return ((size_t)(inline_ll ? 0 : times_pad(pb)) + LDS_REC_DOUBLES + pb.nm_pad + (size_t)cpw * (pb.nm + 1) + (size_t)cpw * pb.P + 6 * WAVE) *
           sizeof(double);
}

// ---- Adaptive-Metropolis state of C chains resident in HBM (csrc/sepaihrd_sampler.hip) ----
// chain_history_ of the reference (MetropolisHastingsSampler.cpp:262-264,354) is read in three places only: its newest
// state by updateCovarianceRank1 (:157), all of it by recomputeFullCovariance (:168-199), every thinning-th state as the
// run's samples (:357-360).  Kept here: a RING of the last `window` states (pending rank-one and co-moment updates read
// it), the running sums that make the refresh O(P^2) (oracle::RunningMoments states the recurrence), and the thinned
// samples.  With covariance_mode two-pass the ring holds every state (window = rows of the whole run) and the refresh
// walks it as the reference does.
struct SamplerState {
    int32_t C, P;
    int32_t window;           // rows of the ring per chain; state r lives in slot r % window
    int32_t thinning, n_store;  // states r with r % thinning == 0 are also kept as sample r / thinning (n_store = 0: none)
    double scaling, reg_eps;  // 2.38^2 / P and the regularisation epsilon of the covariance refresh
    double* x;      // [C][P] current state
    double* prop;   // [C][P] last proposal (after applyConstraints)
    double* cov;    // [C][P][P] proposal covariance, row-major
    double* chol;   // [C] x (P P allocated) its lower Cholesky factor, the lower triangle column by column: L(i, j), i >= j, at j P - j (j - 1) / 2 + (i - j)
    double* mean;   // [C][P] running mean (running_mean_)
    double* hist;   // [C][window][P] ring of the newest states
    double* best;   // [C][P] the best state so far (updated by commit where bit 1 of the chain's accept byte is set)
    double* store;  // [C][n_store][P] the thinned samples
    double* sum;    // [C][P] running sum of all states, in the order of the reference's mean loop (:171-174)
    double* wmean;  // [C][P] Welford mean of all states
    double* m2;     // [C][P][P] centred second moment, entries j <= i
    int32_t* accepted;  // [C] accepted proposals so far
    // the chains' std::mt19937 streams when the device draws (csrc/sepaihrd_rng.inc): state words, position in the state,
    // and the words the two continuations of the pending accept test take ([c][0] with the uniform, [c][1] without)
    uint32_t* mt;       // [C][624]
    int32_t* mt_idx;    // [C]
    int32_t* mt_used;   // [C][2]
    // adaptGlobalScale (MetropolisHastingsSampler.cpp:104-152) per chain when the device keeps it: log_scale_ / global_scale_,
    // the window of the last 1000 accept flags as a ring with its running sum, the emergency counter
    int32_t device_scale;   // 0: the caller supplies both candidates of the next scale with every test
    int32_t adapt_scale;    // adapt_scale_ of the reference's settings
    double target_rate;     // target_acceptance_rate_
    double* log_scale;      // [C]
    double* scale;          // [C]
    uint8_t* recent;        // [C][1000]
    int32_t* recent_meta;   // [C][4]: position, length, sum, emergency shrinks
    double* lp_store;       // [C][n_store] the chain's value at every stored sample (sampleObjectiveValues)
    uint8_t* trace;         // [iterations - 1][C] accept flags of every test, or null
    uint32_t* fail_counts;  // [3] evaluations the accept test saw with status 2, 3, 4 (whole sampler), or null
};
// the device's glibc_log / glibc_exp on the N = sampler_libm_check_count() self-check arguments of csrc/sepaihrd_rng.inc:
// d_out [log args N | log values N | exp args N | exp values N]
int sampler_libm_check_values(double* d_out, void* stream);
int sampler_libm_check_count();
// progress / checkpoint block of n chains: per chain [value, best, scale, accepted | samples first.. [count][P] | values [count]]
int sampler_snapshot(const SamplerState& s, const double* d_lp, const double* d_best_lp, const int32_t* d_chains, int n, int first,
                     int count, double* d_out, void* stream);
int sampler_propose(const SamplerState& s, const DevProblem& pb, const double* d_z, const double* d_scale, void* stream);
int sampler_commit(const SamplerState& s, const uint8_t* d_accept, int row, void* stream);
int sampler_commit_counted(const SamplerState& s, const uint8_t* d_accept, int row, int accepted_known, void* stream);
// the accept test on the device (flags: bit 0 accepted, bit 1 best so far, bit 2 no uniform drawn) and the proposal that
// follows it with the normals of the continuation taken
// row: the state the test's outcome becomes (= the iteration t of the test); used by the device-kept scale adaptation,
// the value store and the accept trace
int sampler_accept_test(const SamplerState& s, const int row, const double* d_loglik, const int32_t* d_status, const double* d_log_u, const double* d_scale_reject,
                        const double* d_scale_accept, double* d_lp, double* d_best_lp, double* d_scale_sel, uint8_t* d_flags,
                        double* d_values, void* stream);
// the three in one launch (block = chain), for iterations without a covariance refresh between commit and proposal
int sampler_test_commit_propose(const SamplerState& s, const DevProblem& pb, const double* d_loglik, const int32_t* d_status,
                                const double* d_log_u, const double* d_scale_reject, const double* d_scale_accept, double* d_lp,
                                double* d_best_lp, double* d_scale_sel, uint8_t* d_flags, double* d_values, const double* d_z_uniform,
                                const double* d_z_plain, int row, void* stream, const double* d_lz_uniform = nullptr,
                                const double* d_lz_plain = nullptr);
// L z of both continuations' normals ahead of the test (d_lz_*: [C][P]); the fused launch above then takes them instead of
// reading the factor itself
int sampler_lz(const SamplerState& s, const double* d_z_uniform, const double* d_z_plain, double* d_lz_uniform, double* d_lz_plain, void* stream);
int sampler_propose_select(const SamplerState& s, const DevProblem& pb, const double* d_z_uniform, const double* d_z_plain,
                           const uint8_t* d_flags, const double* d_scale, void* stream);
int sampler_patch_normals(double* d_z, const int32_t* d_chain, const double* d_rows, int n_patch, int P, void* stream);
// n queued rank-one updates (:154-166) in order: update k reads state d_rows[k] (still in the ring) with d_gammas[k]
int sampler_rank1_catchup(const SamplerState& s, const int32_t* d_rows, const double* d_gammas, int n, void* stream);
// states row0 .. row0 + n - 1 (still in the ring; they are states number row0 + 1 .. of the chain) enter the running
// sums; emit_len > 0: then running mean and covariance of a history of emit_len states are written (:175-190)
int sampler_moments_catchup(const SamplerState& s, int row0, int n, int emit_len, void* stream);
// recomputeFullCovariance as the reference writes it, two passes over states 0 .. len - 1 (the ring must hold them all)
int sampler_full_covariance(const SamplerState& s, int len, void* stream);
int sampler_cholesky(const SamplerState& s, double diag_add, int on_failure, void* stream);
// the two factorisations of the adaptation-period step after a full recompute (cov, then cov + eps I, each kept on
// success) as one: cov + eps I first, cov only if that fails -- the same factor in every case
int sampler_cholesky_refresh(const SamplerState& s, void* stream);
// std::mt19937(seed0 + chain) for every chain
int sampler_seed_streams(const SamplerState& s, uint32_t seed0, void* stream);
// The draws of one accept test and of the proposal after it, from every chain's stream: log(u) of the uniform the test
// takes when log_ratio < 0, the P normals that follow it (z_uniform) and the P normals from the same position when no
// uniform is taken (z_plain).  The stream first moves by what the PREVIOUS test's continuation used (d_flags bit 2 picks
// it; first != 0: the stream's start, only z_plain's values are drawn and written to z_uniform).
int sampler_draw(const SamplerState& s, const uint8_t* d_flags, int first, double* d_log_u, double* d_z_uniform, double* d_z_plain,
                 int want_normals, void* stream);
// per-chain summary record of SURVEY 8(e): [P means | P variances (n - 1) | best value | accepted proposals] over the
// stored samples first_sample .. n_samples - 1, written to d_out [C][2 P + 2]
int sampler_summary_records(const SamplerState& s, const double* d_best_lp, int first_sample, int n_samples, double* d_out, void* stream);

}  // namespace sepaihrd
