// csrc/sepaihrd_capi.cpp -- implementation of the C ABI declared in include/sepaihrd_hip.h.
// Host side only: validates the problem, resolves the theta -> field map into slot tables,
// uploads the problem to HBM once, and launches the HIP kernels.  There is no CPU
// evaluation path in this library.
#include "sepaihrd_hip.h"

#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#include "sepaihrd_device.h"

using namespace sepaihrd;

struct sepaihrd_ctx {
    int device = 0;
    int solver = 0;
    int arith = 0;
    int precision = 0;  // SEPAIHRD_PRECISION_F64 / _F32
    DevProblem dp{};
    std::vector<void*> allocs;
    // host copies needed by sepaihrd_apply_constraints
    std::vector<double> lower, upper;
    std::vector<uint8_t> has_bounds;
    int n = 0, T = 0, P = 0;
    std::vector<double> host_N;  // population sizes (ensemble seroprevalence)
    // buffers of sepaihrd_ensemble_quantiles, kept between calls (grow-only): allocating tens of GB per call
    // costs more than the kernels at large ensembles
    void* ens_buf[9] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t ens_cap[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    std::string last_error;
    // staging buffers for the host-pointer entry point (grown on demand)
    size_t cap_B = 0;
    hipStream_t own_stream = nullptr;  // sepaihrd_eval_batch_begin / _end
    int pending_B = 0;
    bool cap_traj = false;
    double* d_theta = nullptr;
    double* d_loglik = nullptr;
    int32_t* d_status = nullptr;
    int32_t* d_nacc = nullptr;
    int32_t* d_nrej = nullptr;
    double* d_parts = nullptr;
    // d_loglik .. d_parts are views into ONE allocation (results slab: [loglik B][parts 3B][status B][accepted B][rejected B]),
    // fetched with one copy into a page-locked mirror of the same layout
    void* d_results = nullptr;
    void* h_results = nullptr;
    double* d_traj = nullptr;
    size_t cap_traj_elems = 0;
    // likelihood-pass workspace (device), sized for ws_chains chains
    size_t ws_chains = 0;
    double* ws_cum = nullptr;
    double* ws_rows = nullptr;
    int32_t* ws_status = nullptr;
    size_t ws_budget_bytes = (size_t)24 << 30;  // larger batches are evaluated in chunks of chains
    // ONE evaluation in flight per context (they share the workspace above): the last launch sequence leaves an
    // event, and a launch on a different stream waits for it first (free when the stream is the same)
    hipEvent_t busy_event = nullptr;
    hipStream_t busy_stream = nullptr;
    bool busy_valid = false;
    // a stream the LIBRARY owns (a sampler's) on which launches after the first record nothing: every marker between two
    // kernels of a stream is dispatch latency (3-5 us each in the sampler loop).  The event is re-recorded lazily when
    // another stream asks (fence_before); only for an owned stream, whose handle is known to be alive
    hipStream_t lazy_stream = nullptr;
    // sepaihrd_device_libm_check: -1 not run yet, else the number of self-check arguments on which the device's log / exp
    // restatements differ from this process's libm
    int libm_log_diff = -1, libm_exp_diff = -1;
    // optional per-kernel timing (HIP events on the launch stream), see sepaihrd_set_timing
    bool timing = false;
    int timing_period = 1;     // events around every timing_period-th launch sequence only
    long timing_seen = 0;
    std::vector<hipEvent_t> ev;  // triples: before integrator, after integrator, after likelihood pass
    size_t ev_used = 0;
    // per-chain summary records: the table of the chains this context ran, and the gathered table of all
    double* rec_buf[2] = {nullptr, nullptr};
    size_t rec_cap[2] = {0, 0};
};

namespace {

void set_err(char* err, int errlen, const std::string& msg) {
    if (err && errlen > 0) std::snprintf(err, (size_t)errlen, "%s", msg.c_str());
}

#define HIP_TRY(expr, ctx, fail)                                                             \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) {                                                              \
            (ctx)->last_error = std::string(#expr) + ": " + hipGetErrorString(e_);           \
            fail;                                                                            \
        }                                                                                    \
    } while (0)

template <class T>
const T* upload(sepaihrd_ctx* ctx, const std::vector<T>& v, bool& ok) {
    if (v.empty()) {
        // keep a valid (1-element) allocation so kernels may form the pointer
        void* p = nullptr;
        if (hipMalloc(&p, sizeof(T)) != hipSuccess) { ok = false; return nullptr; }
        ctx->allocs.push_back(p);
        return static_cast<const T*>(p);
    }
    void* p = nullptr;
    if (hipMalloc(&p, v.size() * sizeof(T)) != hipSuccess) { ok = false; return nullptr; }
    ctx->allocs.push_back(p);
    if (hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) ok = false;
    return static_cast<const T*>(p);
}

// SEPAIHRDParameterManager.cpp:302-313
double reflect_bound_host(double value, double minb, double maxb) {
    if (minb >= maxb) return minb;
    const double width = maxb - minb;
    double y = std::fmod(value - minb, 2.0 * width);
    if (y < 0) y += 2.0 * width;
    if (y <= width) return minb + y;
    return maxb - (y - width);
}

void free_workspace(sepaihrd_ctx* c) {
    if (c->ws_cum) (void)hipFree(c->ws_cum);
    if (c->ws_rows) (void)hipFree(c->ws_rows);
    if (c->ws_status) (void)hipFree(c->ws_status);
    c->ws_cum = c->ws_rows = nullptr;
    c->ws_status = nullptr;
    c->ws_chains = 0;
}

// chains per launch so that the cumulative-compartment workspace stays within the budget
size_t chunk_chains(const sepaihrd_ctx* c, size_t B) {
    const size_t cpw = (size_t)(WAVE / c->dp.lpc);
    const size_t per_chain = (size_t)c->dp.T * 3 * c->dp.lpc * sizeof(double);
    size_t fit = std::max<size_t>(cpw, (c->ws_budget_bytes / per_chain) / cpw * cpw);
    const size_t want = (B + cpw - 1) / cpw * cpw;
    return std::min(fit, want);
}

int ensure_workspace(sepaihrd_ctx* c, size_t chains) {
    if (chains <= c->ws_chains) return SEPAIHRD_OK;
    free_workspace(c);
    if (hipMalloc((void**)&c->ws_cum, workspace_cum_doubles(c->dp, chains) * sizeof(double)) != hipSuccess ||
        hipMalloc((void**)&c->ws_rows, workspace_rows_doubles(c->dp, chains) * sizeof(double)) != hipSuccess ||
        hipMalloc((void**)&c->ws_status, chains * sizeof(int32_t)) != hipSuccess) {
        free_workspace(c);
        c->last_error = "likelihood workspace allocation failed";
        return SEPAIHRD_E_HIP;
    }
    c->ws_chains = chains;
    return SEPAIHRD_OK;
}

// Does a launch of B chains use the ctx-owned workspace?  Decided by the kernel translation unit (the launch code
// takes the same branches); < 0: unsupported lanes-per-chain.
int needs_workspace(const sepaihrd_ctx* c, int B, int force_split) {
    if (c->precision == SEPAIHRD_PRECISION_F32 && !force_split) return c->dp.lpc >= 4 ? 0 : -4;  // likelihood inline
    return c->arith == SEPAIHRD_ARITH_FMA ? launch_needs_workspace_fma(c->dp, c->solver, B, force_split)
                                          : launch_needs_workspace_strict(c->dp, c->solver, B, force_split);
}

bool stream_is_capturing(hipStream_t st) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    return hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
}

// Order this launch sequence after the previous one of the same context when it runs on another stream.
// Inside a stream capture nothing is recorded or waited for (the captured graph owns its ordering).
int fence_before(sepaihrd_ctx* c, hipStream_t st) {
    if (!c->busy_valid || c->busy_stream == st || stream_is_capturing(st)) return SEPAIHRD_OK;
    // launches after the first on a sampler's stream recorded nothing: mark the end of what is queued there now, for
    // this other stream to wait on
    if (c->lazy_stream != nullptr && c->busy_stream == c->lazy_stream) (void)hipEventRecord(c->busy_event, c->busy_stream);
    if (hipStreamWaitEvent(st, c->busy_event, 0) != hipSuccess) {
        c->last_error = "hipStreamWaitEvent on the context's previous evaluation failed";
        return SEPAIHRD_E_HIP;
    }
    return SEPAIHRD_OK;
}
void fence_after(sepaihrd_ctx* c, hipStream_t st) {
    if (c->busy_valid && c->busy_stream == st && st == c->lazy_stream) return;  // recorded lazily (fence_before)
    if (stream_is_capturing(st)) return;
    if (!c->busy_event && hipEventCreateWithFlags(&c->busy_event, hipEventDisableTiming) != hipSuccess) {
        c->busy_event = nullptr;
        return;
    }
    if (hipEventRecord(c->busy_event, st) == hipSuccess) { c->busy_stream = st; c->busy_valid = true; }
}

void free_staging(sepaihrd_ctx* c) {
    void* ptrs[] = {c->d_theta, c->d_results, c->d_traj};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    if (c->h_results) (void)hipHostFree(c->h_results);
    c->d_theta = c->d_loglik = c->d_parts = c->d_traj = nullptr;
    c->d_status = c->d_nacc = c->d_nrej = nullptr;
    c->d_results = c->h_results = nullptr;
    c->cap_B = 0;
    c->cap_traj_elems = 0;
}

constexpr size_t RESULT_BYTES_PER_CHAIN = 4 * sizeof(double) + 3 * sizeof(int32_t);

// Staging of the host-pointer entry points for B chains (keeps a trajectory buffer that is already there).  The result
// views are laid out for THIS batch at the front of the slab, so that one copy fetches them whatever the capacity.
int ensure_staging(sepaihrd_ctx* ctx, size_t B) {
    if (B > ctx->cap_B) {
        const size_t keep_traj = ctx->cap_traj_elems;
        double* keep = ctx->d_traj;
        ctx->d_traj = nullptr;
        free_staging(ctx);
        ctx->d_traj = keep;
        ctx->cap_traj_elems = keep_traj;
        HIP_TRY(hipMalloc((void**)&ctx->d_theta, B * ctx->P * sizeof(double)), ctx, return SEPAIHRD_E_HIP);
        HIP_TRY(hipMalloc(&ctx->d_results, B * RESULT_BYTES_PER_CHAIN), ctx, return SEPAIHRD_E_HIP);
        HIP_TRY(hipHostMalloc(&ctx->h_results, B * RESULT_BYTES_PER_CHAIN, hipHostMallocDefault), ctx, return SEPAIHRD_E_HIP);
        ctx->cap_B = B;
    }
    ctx->d_loglik = static_cast<double*>(ctx->d_results);
    ctx->d_parts = ctx->d_loglik + B;
    ctx->d_status = reinterpret_cast<int32_t*>(ctx->d_parts + 3 * B);
    ctx->d_nacc = ctx->d_status + B;
    ctx->d_nrej = ctx->d_nacc + B;
    return SEPAIHRD_OK;
}

// One copy of the batch's results (the span up to the last array asked for) and the wait; then the caller's arrays are
// filled from the page-locked mirror.
int fetch_results(sepaihrd_ctx* ctx, hipStream_t st, int B, double* loglik, int32_t* status, int32_t* n_accept, int32_t* n_reject,
                  double* ll_parts) {
    const size_t n = (size_t)B;
    char* const h = static_cast<char*>(ctx->h_results);
    const size_t off_parts = n * sizeof(double), off_status = 4 * n * sizeof(double);
    const size_t off_nacc = off_status + n * sizeof(int32_t), off_nrej = off_nacc + n * sizeof(int32_t);
    size_t hi = n * sizeof(double);  // the log-likelihoods, always
    if (ll_parts) hi = off_status;
    if (status) hi = off_nacc;
    if (n_accept) hi = off_nrej;
    if (n_reject) hi = off_nrej + n * sizeof(int32_t);
    HIP_TRY(hipMemcpyAsync(h, ctx->d_results, hi, hipMemcpyDeviceToHost, st), ctx, return SEPAIHRD_E_HIP);
    HIP_TRY(hipStreamSynchronize(st), ctx, return SEPAIHRD_E_HIP);
    if (loglik) std::memcpy(loglik, h, n * sizeof(double));
    if (ll_parts) std::memcpy(ll_parts, h + off_parts, n * 3 * sizeof(double));
    if (status) std::memcpy(status, h + off_status, n * sizeof(int32_t));
    if (n_accept) std::memcpy(n_accept, h + off_nacc, n * sizeof(int32_t));
    if (n_reject) std::memcpy(n_reject, h + off_nrej, n * sizeof(int32_t));
    return SEPAIHRD_OK;
}

}  // namespace

extern "C" {

int sepaihrd_abi_version(void) { return SEPAIHRD_ABI_VERSION; }

sepaihrd_ctx* sepaihrd_create(const sepaihrd_problem* pb, int device, char* err, int errlen) {
    if (!pb) { set_err(err, errlen, "problem is NULL"); return nullptr; }
    if (pb->abi_version != SEPAIHRD_ABI_VERSION) { set_err(err, errlen, "ABI version mismatch"); return nullptr; }
    const int n = pb->n_age, T = pb->n_times, P = pb->n_params, nb = pb->n_beta, nk = pb->n_kappa;
    if (n < 1 || n > SEPAIHRD_MAX_AGE_CLASSES) { set_err(err, errlen, "n_age out of range [1,64]"); return nullptr; }
    if (lanes_per_chain(n) > 16) {
        set_err(err, errlen, "n_age > 16 not built in this version"); return nullptr;
    }
    if (T < 1) { set_err(err, errlen, "n_times must be >= 1"); return nullptr; }
    if (T > MAX_TIMES) { set_err(err, errlen, "n_times exceeds the LDS-staged grid limit (12288)"); return nullptr; }
    if (P < 1) { set_err(err, errlen, "n_params must be >= 1"); return nullptr; }
    if (nk < 1 || nk > SEPAIHRD_MAX_SCHEDULE || nb < 0 || nb > SEPAIHRD_MAX_SCHEDULE) {
        set_err(err, errlen, "schedule lengths out of range (n_kappa >= 1)"); return nullptr;
    }
    if (!pb->times || !pb->N || !pb->M || !pb->a || !pb->h_infec || !pb->p || !pb->h || !pb->icu ||
        !pb->d_H || !pb->d_ICU || !pb->kappa_end_times || !pb->kappa_values || !pb->initial_state ||
        !pb->param_field || !pb->param_index || (nb > 0 && (!pb->beta_end_times || !pb->beta_values)) ||
        (pb->n_obs > 0 && (!pb->obs_H || !pb->obs_ICU || !pb->obs_D))) {
        set_err(err, errlen, "a required array pointer is NULL"); return nullptr;
    }
    if (pb->solver != SEPAIHRD_SOLVER_DOPRI5 && pb->solver != SEPAIHRD_SOLVER_CASH_KARP54) {
        set_err(err, errlen, "unknown solver"); return nullptr;
    }
    // Simulator::run grid validation (Simulator.cpp:78-88) and ctor checks (:15-44)
    for (int i = 1; i < T; ++i)
        if (!(pb->times[i] > pb->times[i - 1])) {
            set_err(err, errlen, "time points must be strictly increasing"); return nullptr;
        }
    if (pb->abs_err < 0 || pb->rel_err < 0) { set_err(err, errlen, "negative error tolerance"); return nullptr; }
    if (!(pb->dt_hint > 0)) { set_err(err, errlen, "dt_hint must be positive"); return nullptr; }
    // schedule validation: PiecewiseConstantNpiStrategy ctor (PieceWiseConstantNPIStrategy.cpp:24-54),
    // PiecewiseConstantParameterStrategy ctor (PiecewiseConstantParameterStrategy.cpp:22-34)
    if (pb->kappa_end_times[0] < 0.0) { set_err(err, errlen, "kappa baseline end time must be non-negative"); return nullptr; }
    for (int k = 1; k < nk; ++k)
        if (!(pb->kappa_end_times[k] > pb->kappa_end_times[k - 1])) {
            set_err(err, errlen, "kappa end times must be strictly increasing"); return nullptr;
        }
    for (int k = 1; k < nb; ++k)
        if (!(pb->beta_end_times[k] > pb->beta_end_times[k - 1])) {
            set_err(err, errlen, "beta end times must be strictly increasing"); return nullptr;
        }

    int ndev = 0;
    {
        const hipError_t e = hipGetDeviceCount(&ndev);
        if (e != hipSuccess || ndev <= 0) {
            set_err(err, errlen, std::string("no HIP device available (this library has no CPU fallback): "
                                             "hipGetDeviceCount -> ") + hipGetErrorString(e) + ", count " +
                                     std::to_string(ndev));
            return nullptr;
        }
    }
    if (device < 0) {
        if (hipGetDevice(&device) != hipSuccess) { set_err(err, errlen, "hipGetDevice failed"); return nullptr; }
    }
    if (device >= ndev) { set_err(err, errlen, "device index out of range"); return nullptr; }
    if (hipSetDevice(device) != hipSuccess) { set_err(err, errlen, "hipSetDevice failed"); return nullptr; }

    auto* ctx = new sepaihrd_ctx();
    ctx->device = device;
    ctx->solver = pb->solver;
    ctx->arith = pb->arith == SEPAIHRD_ARITH_FMA ? SEPAIHRD_ARITH_FMA : SEPAIHRD_ARITH_STRICT;
    if (pb->precision != SEPAIHRD_PRECISION_F64 && pb->precision != SEPAIHRD_PRECISION_F32) {
        set_err(err, errlen, "unknown precision"); delete ctx; return nullptr;
    }
    if (pb->precision == SEPAIHRD_PRECISION_F32 && lanes_per_chain(n) < 4) {
        set_err(err, errlen, "the fp32-state arm is built for 3 to 16 age classes"); delete ctx; return nullptr;
    }
    ctx->precision = pb->precision;
    ctx->n = n; ctx->T = T; ctx->P = P;
    ctx->host_N.assign(pb->N, pb->N + n);
    if (const char* mb = std::getenv("SEPAIHRD_WORKSPACE_MB")) {  // likelihood-workspace budget (default 24 GiB)
        const long v = std::atol(mb);
        if (v > 0) ctx->ws_budget_bytes = (size_t)v << 20;
    }

    const int lpc = lanes_per_chain(n);
    const int ns = SS_SCHEDULE0 + nb + nk;

    // ---- slot tables: later theta entries overwrite earlier ones, as the reference's
    //      sequential loop over names does (SEPAIHRDParameterManager.cpp:197-267)
    std::vector<int32_t> src_scalar(ns, -1);
    std::vector<double> base_scalar(ns, 0.0);
    base_scalar[SS_BETA] = pb->beta; base_scalar[SS_THETA] = pb->theta; base_scalar[SS_SIGMA] = pb->sigma;
    base_scalar[SS_GAMMA_P] = pb->gamma_p; base_scalar[SS_GAMMA_A] = pb->gamma_A;
    base_scalar[SS_GAMMA_I] = pb->gamma_I; base_scalar[SS_GAMMA_H] = pb->gamma_H;
    base_scalar[SS_GAMMA_ICU] = pb->gamma_ICU;
    for (int i = 0; i < 8; ++i) base_scalar[SS_E0_MULT + i] = pb->multipliers[i];
    base_scalar[SS_RUNUP_DAYS] = pb->runup_days; base_scalar[SS_SEED_EXPOSED] = pb->seed_exposed;
    for (int k = 0; k < nb; ++k) base_scalar[SS_SCHEDULE0 + k] = pb->beta_values[k];
    for (int k = 0; k < nk; ++k) base_scalar[SS_SCHEDULE0 + nb + k] = pb->kappa_values[k];

    std::vector<int32_t> src_vec((size_t)VF_COUNT * lpc, -1);
    std::vector<double> base_vec((size_t)VF_COUNT * lpc, 0.0);
    const double* vec_base_ptr[VF_COUNT] = {pb->a, pb->h_infec, pb->p, pb->h, pb->icu, pb->d_H, pb->d_ICU,
                                            pb->d_community};
    for (int f = 0; f < VF_COUNT; ++f)
        for (int i = 0; i < n; ++i) base_vec[(size_t)f * lpc + i] = vec_base_ptr[f] ? vec_base_ptr[f][i] : 0.0;

    int kappa_calibrated = 0;
    for (int p = 0; p < P; ++p) {
        const int f = pb->param_field[p], idx = pb->param_index[p];
        auto bad = [&](const char* what) {
            set_err(err, errlen, std::string("param ") + std::to_string(p) + ": " + what);
            delete ctx;
            return static_cast<sepaihrd_ctx*>(nullptr);
        };
        if (f == SEPAIHRD_F_NONE) continue;
        if (f >= SEPAIHRD_F_BETA && f <= SEPAIHRD_F_SEED_EXPOSED) {
            src_scalar[f] = p;
        } else if (f == SEPAIHRD_F_BETA_VALUE) {
            if (idx < 0 || idx >= nb) return bad("beta index out of range");
            src_scalar[SS_SCHEDULE0 + idx] = p;
        } else if (f == SEPAIHRD_F_KAPPA_VALUE) {
            if (idx < 1 || idx >= nk) return bad("kappa index out of range (index 0 is the fixed baseline)");
            src_scalar[SS_SCHEDULE0 + nb + idx] = p;
            kappa_calibrated = 1;
        } else if (f >= SEPAIHRD_F_A && f <= SEPAIHRD_F_D_COMMUNITY) {
            if (idx < 0 || idx >= n) return bad("age index out of range");
            src_vec[(size_t)(f - SEPAIHRD_F_A) * lpc + idx] = p;
        } else {
            return bad("unknown field code");
        }
    }

    // ---- padded per-age tables
    std::vector<double> Npad(lpc, 0.0), fracpad(lpc, 0.0), Mrow((size_t)lpc * lpc, 0.0),
        init((size_t)NUM_COMP * lpc, 0.0);
    double total_pop = 0.0;
    for (int i = 0; i < n; ++i) { Npad[i] = pb->N[i]; total_pop += pb->N[i]; }
    if (total_pop > 0.0)  // SEPAIHRDObjectiveFunction.cpp:103-108
        for (int i = 0; i < n; ++i) fracpad[i] = pb->N[i] / total_pop;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) Mrow[(size_t)i * lpc + j] = pb->M[(size_t)j * n + i];
    for (int c = 0; c < NUM_COMP; ++c)
        for (int i = 0; i < n; ++i) init[(size_t)c * lpc + i] = pb->initial_state[(size_t)c * n + i];

    int runup_offset = 0;  // SEPAIHRDObjectiveFunction.cpp:39-46
    for (int i = 0; i < T; ++i)
        if (pb->times[i] >= 0.0) { runup_offset = i; break; }
    const int num_obs_points = T - runup_offset;
    const int n_obs = pb->n_obs;
    const double qnan = std::numeric_limits<double>::quiet_NaN();
    // grid records: [T][lpc][4] = {obs_H, obs_ICU, obs_D, times[k+1]}
    std::vector<double> grid((size_t)T * lpc * 4, qnan);
    const double* obs_ptr[3] = {pb->obs_H, pb->obs_ICU, pb->obs_D};
    for (int k = 0; k < T; ++k)
        for (int i = 0; i < lpc; ++i) {
            double* rec = &grid[((size_t)k * lpc + i) * 4];
            const int r = k - runup_offset;
            if (i < n && r >= 0 && r < n_obs)
                for (int s = 0; s < 3; ++s) rec[s] = obs_ptr[s][(size_t)r * n + i];
            rec[3] = pb->times[k + 1 < T ? k + 1 : T - 1];
        }

    // merged beta/kappa schedule (see DevProblem)
    std::vector<double> mends;
    for (int k = 0; k < nb; ++k) mends.push_back(pb->beta_end_times[k]);
    for (int k = 0; k < nk; ++k) mends.push_back(pb->kappa_end_times[k]);
    std::sort(mends.begin(), mends.end());
    mends.erase(std::unique(mends.begin(), mends.end()), mends.end());
    const int nm = (int)mends.size();
    std::vector<int32_t> seg_ib(nm + 1, 0), seg_ik(nm + 1, 0);
    for (int j = 0; j <= nm; ++j) {
        const double trep = j < nm ? mends[j] : std::numeric_limits<double>::infinity();
        int cb = 0, ckk = 0;
        for (int k = 0; k < nb; ++k) cb += (trep > pb->beta_end_times[k]) ? 1 : 0;
        for (int k = 0; k < nk; ++k) ckk += (trep > pb->kappa_end_times[k]) ? 1 : 0;
        seg_ib[j] = nb > 0 ? std::min(cb, nb - 1) : 0;
        seg_ik[j] = std::min(ckk, nk - 1);
    }
    if (nm & 1) mends.push_back(std::numeric_limits<double>::infinity());

    double max_gap = 0.0;
    for (int i = 1; i < T; ++i) max_gap = std::max(max_gap, pb->times[i] - pb->times[i - 1]);

    ctx->lower.assign(P, 0.0); ctx->upper.assign(P, 0.0); ctx->has_bounds.assign(P, 0);
    std::vector<int32_t> hb(P, 0);
    for (int p = 0; p < P; ++p) {
        const bool has = pb->has_bounds ? pb->has_bounds[p] != 0 : (pb->lower && pb->upper);
        ctx->has_bounds[p] = has ? 1 : 0;
        hb[p] = has ? 1 : 0;
        if (has) { ctx->lower[p] = pb->lower[p]; ctx->upper[p] = pb->upper[p]; }
    }

    DevProblem& d = ctx->dp;
    d.n = n; d.lpc = lpc; d.T = T; d.n_obs = n_obs; d.runup_offset = runup_offset; d.nb = nb; d.nk = nk;
    d.P = P; d.ns = ns;
    d.constraint_mode = pb->constraint_mode == SEPAIHRD_CONSTRAINT_REFLECT ? 1 : 0;
    d.kappa_calibrated = kappa_calibrated;
    d.max_attempts = pb->max_attempts > 0 ? pb->max_attempts : 1000000;
    d.obs_rows_match = (num_obs_points == n_obs) ? 1 : 0;
    d.abs_tol = pb->abs_err; d.rel_tol = pb->rel_err; d.dt_hint = pb->dt_hint; d.max_gap = max_gap;

    bool ok = true;
    d.times = upload(ctx, std::vector<double>(pb->times, pb->times + T), ok);
    d.grid = upload(ctx, grid, ok);
    d.lower = upload(ctx, ctx->lower, ok);
    d.upper = upload(ctx, ctx->upper, ok);
    d.has_bounds = upload(ctx, hb, ok);
    d.src_scalar = upload(ctx, src_scalar, ok);
    d.base_scalar = upload(ctx, base_scalar, ok);
    d.src_vec = upload(ctx, src_vec, ok);
    d.base_vec = upload(ctx, base_vec, ok);
    d.N = upload(ctx, Npad, ok);
    d.age_fraction = upload(ctx, fracpad, ok);
    d.Mrow = upload(ctx, Mrow, ok);
    d.init_state = upload(ctx, init, ok);
    d.beta_ends = upload(ctx, std::vector<double>(pb->beta_end_times, pb->beta_end_times + nb), ok);
    d.kappa_ends = upload(ctx, std::vector<double>(pb->kappa_end_times, pb->kappa_end_times + nk), ok);
    d.nm = nm; d.nm_pad = (int)mends.size();
    d.mends = upload(ctx, mends, ok);
    d.seg_ib = upload(ctx, seg_ib, ok);
    d.seg_ik = upload(ctx, seg_ik, ok);
    if (!ok) {
        set_err(err, errlen, "device allocation / upload failed");
        sepaihrd_destroy(ctx);
        return nullptr;
    }
    if (eval_lds_bytes(d) + LOG_TABLE_LDS_BYTES > 64 * 1024) {  // dynamic + the static 2 KB of the Poisson term's log table
        set_err(err, errlen, "n_params too large for the LDS staging buffer");
        sepaihrd_destroy(ctx);
        return nullptr;
    }
    return ctx;
}

void sepaihrd_destroy(sepaihrd_ctx* ctx) {
    if (!ctx) return;
    for (double* b : ctx->rec_buf) if (b) { (void)hipSetDevice(ctx->device); (void)hipFree(b); }
    (void)hipSetDevice(ctx->device);
    if (ctx->own_stream) { (void)hipStreamSynchronize(ctx->own_stream); (void)hipStreamDestroy(ctx->own_stream); }
    free_staging(ctx);
    free_workspace(ctx);
    for (void* p : ctx->ens_buf)
        if (p) (void)hipFree(p);
    for (hipEvent_t e : ctx->ev) (void)hipEventDestroy(e);
    if (ctx->busy_event) (void)hipEventDestroy(ctx->busy_event);
    for (void* p : ctx->allocs) (void)hipFree(p);
    delete ctx;
}

const char* sepaihrd_last_error(const sepaihrd_ctx* ctx) { return ctx ? ctx->last_error.c_str() : "ctx is NULL"; }

int sepaihrd_set_constraint_mode(sepaihrd_ctx* ctx, int mode) {
    if (!ctx || (mode != SEPAIHRD_CONSTRAINT_CLAMP && mode != SEPAIHRD_CONSTRAINT_REFLECT))
        return SEPAIHRD_E_INVALID_ARG;
    ctx->dp.constraint_mode = mode;
    return SEPAIHRD_OK;
}

int sepaihrd_set_arith(sepaihrd_ctx* ctx, int arith) {
    if (!ctx || (arith != SEPAIHRD_ARITH_STRICT && arith != SEPAIHRD_ARITH_FMA)) return SEPAIHRD_E_INVALID_ARG;
    ctx->arith = arith;
    return SEPAIHRD_OK;
}

int sepaihrd_set_precision(sepaihrd_ctx* ctx, int precision) {
    if (!ctx || (precision != SEPAIHRD_PRECISION_F64 && precision != SEPAIHRD_PRECISION_F32)) return SEPAIHRD_E_INVALID_ARG;
    if (precision == SEPAIHRD_PRECISION_F32 && ctx->dp.lpc < 4) {
        ctx->last_error = "the fp32-state arm is built for 3 to 16 age classes";
        return SEPAIHRD_E_UNSUPPORTED;
    }
    ctx->precision = precision;
    return SEPAIHRD_OK;
}

int sepaihrd_reserve(sepaihrd_ctx* ctx, int max_B) {
    if (!ctx || max_B < 0) return SEPAIHRD_E_INVALID_ARG;
    HIP_TRY(hipSetDevice(ctx->device), ctx, return SEPAIHRD_E_HIP);
    return ensure_workspace(ctx, chunk_chains(ctx, (size_t)max_B));
}

int sepaihrd_eval_batch_device(sepaihrd_ctx* ctx, const double* d_theta, int B, double* d_loglik,
                               int32_t* d_status, int32_t* d_n_accept, int32_t* d_n_reject,
                               double* d_ll_parts, double* d_traj, void* stream) {
    if (!ctx) return SEPAIHRD_E_INVALID_ARG;
    if (B < 0 || (B > 0 && (!d_theta || !d_loglik))) {
        ctx->last_error = "eval_batch_device: NULL theta/loglik or negative B";
        return SEPAIHRD_E_INVALID_ARG;
    }
    if (B == 0) return SEPAIHRD_OK;
    // The workspace grows on the first call for a larger batch (an allocation: call sepaihrd_reserve
    // beforehand when the launch must be allocation-free, e.g. under stream capture).
    // batches that fill the chip use the inline-likelihood kernel and need no workspace / chunking
    const int needs = needs_workspace(ctx, B, 0);
    if (needs < 0) { ctx->last_error = "unsupported lanes-per-chain"; return SEPAIHRD_E_UNSUPPORTED; }
    const bool split = needs != 0;
    const size_t chunk = split ? chunk_chains(ctx, (size_t)B) : (size_t)B;
    if (split) {
        const int rc = ensure_workspace(ctx, chunk);
        if (rc != SEPAIHRD_OK) return rc;
    }
    {
        const int rc = fence_before(ctx, static_cast<hipStream_t>(stream));
        if (rc != SEPAIHRD_OK) return rc;
    }
    const size_t traj_per_chain = (size_t)ctx->T * NUM_COMP * ctx->n;
    for (size_t off = 0; off < (size_t)B; off += chunk) {
        const int nb = (int)std::min(chunk, (size_t)B - off);
        hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
        if (ctx->timing && (ctx->timing_seen++ % ctx->timing_period) == 0) {
            while (ctx->ev.size() < ctx->ev_used + 3) {
                hipEvent_t e;
                HIP_TRY(hipEventCreate(&e), ctx, return SEPAIHRD_E_HIP);
                ctx->ev.push_back(e);
            }
            e0 = ctx->ev[ctx->ev_used]; e1 = ctx->ev[ctx->ev_used + 1]; e2 = ctx->ev[ctx->ev_used + 2];
            ctx->ev_used += 3;
            (void)hipEventRecord(e0, static_cast<hipStream_t>(stream));
        }
        EvalOutputs out{d_loglik + off,
                        d_status ? d_status + off : nullptr,
                        d_n_accept ? d_n_accept + off : nullptr,
                        d_n_reject ? d_n_reject + off : nullptr,
                        d_ll_parts ? d_ll_parts + 3 * off : nullptr,
                        d_traj ? d_traj + off * traj_per_chain : nullptr,
                        ctx->ws_cum, ctx->ws_rows, ctx->ws_status, e1, 0};
        const double* th = d_theta + off * (size_t)ctx->P;
        const int rc = ctx->precision == SEPAIHRD_PRECISION_F32 ? launch_eval_f32(ctx->dp, ctx->solver, th, nb, out, stream)
                       : ctx->arith == SEPAIHRD_ARITH_FMA       ? launch_eval_fma(ctx->dp, ctx->solver, th, nb, out, stream)
                                                                : launch_eval_strict(ctx->dp, ctx->solver, th, nb, out, stream);
        if (rc != 0) {
            ctx->last_error = rc == -4 ? "unsupported lanes-per-chain"
                              : rc == -5 ? "launch needs the likelihood workspace but none was sized for it"
                                         : "kernel launch failed";
            return rc == -4 ? SEPAIHRD_E_UNSUPPORTED : SEPAIHRD_E_HIP;
        }
        if (e2) (void)hipEventRecord(e2, static_cast<hipStream_t>(stream));
    }
    fence_after(ctx, static_cast<hipStream_t>(stream));
    return SEPAIHRD_OK;
}

int sepaihrd_set_timing(sepaihrd_ctx* ctx, int enable) {
    if (!ctx) return SEPAIHRD_E_INVALID_ARG;
    ctx->timing = enable != 0;
    ctx->timing_period = enable > 1 ? enable : 1;
    ctx->timing_seen = 0;
    ctx->ev_used = 0;
    return SEPAIHRD_OK;
}

int sepaihrd_get_timing(sepaihrd_ctx* ctx, double* integrator_ms, double* likelihood_ms, int* launches) {
    if (!ctx || !integrator_ms || !likelihood_ms || !launches) return SEPAIHRD_E_INVALID_ARG;
    double a = 0.0, b = 0.0;
    const size_t n = ctx->ev_used / 3;
    for (size_t i = 0; i < n; ++i) {
        float m0 = 0.f, m1 = 0.f;
        HIP_TRY(hipEventSynchronize(ctx->ev[3 * i + 2]), ctx, return SEPAIHRD_E_HIP);
        HIP_TRY(hipEventElapsedTime(&m0, ctx->ev[3 * i], ctx->ev[3 * i + 1]), ctx, return SEPAIHRD_E_HIP);
        HIP_TRY(hipEventElapsedTime(&m1, ctx->ev[3 * i + 1], ctx->ev[3 * i + 2]), ctx, return SEPAIHRD_E_HIP);
        a += m0; b += m1;
    }
    *integrator_ms = a; *likelihood_ms = b; *launches = (int)n;
    ctx->ev_used = 0;
    return SEPAIHRD_OK;
}

int sepaihrd_eval_batch(sepaihrd_ctx* ctx, const double* theta, int B, double* loglik, int32_t* status,
                        int32_t* n_accept, int32_t* n_reject, double* ll_parts, double* traj) {
    if (!ctx) return SEPAIHRD_E_INVALID_ARG;
    if (B < 0 || (B > 0 && (!theta || !loglik))) {
        ctx->last_error = "eval_batch: NULL theta/loglik or negative B";
        return SEPAIHRD_E_INVALID_ARG;
    }
    if (B == 0) return SEPAIHRD_OK;
    if (!traj) {  // the common case: asynchronous copies on the context's stream and ONE wait
        const int rc = sepaihrd_eval_batch_begin(ctx, theta, B);
        return rc != SEPAIHRD_OK ? rc : sepaihrd_eval_batch_end(ctx, loglik, status, n_accept, n_reject, ll_parts);
    }
    HIP_TRY(hipSetDevice(ctx->device), ctx, return SEPAIHRD_E_HIP);
    const size_t traj_elems = traj ? (size_t)B * ctx->T * NUM_COMP * ctx->n : 0;
    if (ctx->pending_B > 0) { ctx->last_error = "eval_batch: a sepaihrd_eval_batch_begin is pending"; return SEPAIHRD_E_INVALID_ARG; }
    {
        const int rc = ensure_staging(ctx, (size_t)B);
        if (rc != SEPAIHRD_OK) return rc;
    }
    if (traj_elems > ctx->cap_traj_elems) {
        if (ctx->d_traj) (void)hipFree(ctx->d_traj);
        ctx->d_traj = nullptr;
        ctx->cap_traj_elems = 0;
        HIP_TRY(hipMalloc((void**)&ctx->d_traj, traj_elems * sizeof(double)), ctx, return SEPAIHRD_E_HIP);
        ctx->cap_traj_elems = traj_elems;
    }
    HIP_TRY(hipMemcpy(ctx->d_theta, theta, (size_t)B * ctx->P * sizeof(double), hipMemcpyHostToDevice), ctx,
            return SEPAIHRD_E_HIP);
    const int rc = sepaihrd_eval_batch_device(ctx, ctx->d_theta, B, ctx->d_loglik, ctx->d_status, ctx->d_nacc,
                                              ctx->d_nrej, ctx->d_parts, traj ? ctx->d_traj : nullptr, nullptr);
    if (rc != SEPAIHRD_OK) return rc;
    HIP_TRY(hipDeviceSynchronize(), ctx, return SEPAIHRD_E_HIP);
    {
        const int rcf = fetch_results(ctx, nullptr, B, loglik, status, n_accept, n_reject, ll_parts);
        if (rcf != SEPAIHRD_OK) return rcf;
    }
    if (traj)
        HIP_TRY(hipMemcpy(traj, ctx->d_traj, traj_elems * sizeof(double), hipMemcpyDeviceToHost), ctx,
                return SEPAIHRD_E_HIP);
    return SEPAIHRD_OK;
}

// The host-pointer evaluation in two halves on a stream of the context's own, so that a caller with two contexts
// (the finite-difference objective: centre value and perturbed batch) has both in flight at once.
int sepaihrd_eval_batch_begin(sepaihrd_ctx* ctx, const double* theta, int B) {
    if (!ctx) return SEPAIHRD_E_INVALID_ARG;
    if (B <= 0 || !theta) { ctx->last_error = "eval_batch_begin: NULL theta or B <= 0"; return SEPAIHRD_E_INVALID_ARG; }
    if (ctx->pending_B > 0) { ctx->last_error = "eval_batch_begin: the previous begin has no end yet"; return SEPAIHRD_E_INVALID_ARG; }
    HIP_TRY(hipSetDevice(ctx->device), ctx, return SEPAIHRD_E_HIP);
    if (!ctx->own_stream) HIP_TRY(hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking), ctx, return SEPAIHRD_E_HIP);
    {
        const int rc = ensure_staging(ctx, (size_t)B);
        if (rc != SEPAIHRD_OK) return rc;
    }
    if (sepaihrd_reserve(ctx, B) != SEPAIHRD_OK) return SEPAIHRD_E_HIP;  // workspace growth is not stream-ordered
    HIP_TRY(hipMemcpyAsync(ctx->d_theta, theta, (size_t)B * ctx->P * sizeof(double), hipMemcpyHostToDevice, ctx->own_stream), ctx,
            return SEPAIHRD_E_HIP);
    const int rc = sepaihrd_eval_batch_device(ctx, ctx->d_theta, B, ctx->d_loglik, ctx->d_status, ctx->d_nacc, ctx->d_nrej,
                                              ctx->d_parts, nullptr, ctx->own_stream);
    if (rc != SEPAIHRD_OK) return rc;
    ctx->pending_B = B;
    return SEPAIHRD_OK;
}

int sepaihrd_eval_batch_end(sepaihrd_ctx* ctx, double* loglik, int32_t* status, int32_t* n_accept, int32_t* n_reject,
                            double* ll_parts) {
    if (!ctx) return SEPAIHRD_E_INVALID_ARG;
    const int B = ctx->pending_B;
    if (B <= 0) { ctx->last_error = "eval_batch_end: nothing pending"; return SEPAIHRD_E_INVALID_ARG; }
    ctx->pending_B = 0;
    HIP_TRY(hipSetDevice(ctx->device), ctx, return SEPAIHRD_E_HIP);
    return fetch_results(ctx, ctx->own_stream, B, loglik, status, n_accept, n_reject, ll_parts);
}

int sepaihrd_set_initial_state_mode(sepaihrd_ctx* ctx, int mode) {
    if (!ctx || mode < SEPAIHRD_INIT_FROM_THETA || mode > SEPAIHRD_INIT_MULTIPLIERS) return SEPAIHRD_E_INVALID_ARG;
    ctx->dp.init_mode = mode;
    return SEPAIHRD_OK;
}

int sepaihrd_set_integrator_form(sepaihrd_ctx* ctx, int form) {
    if (!ctx || form < SEPAIHRD_FORM_AUTO || form > SEPAIHRD_FORM_QUAD) return SEPAIHRD_E_INVALID_ARG;
    if (form == SEPAIHRD_FORM_QUAD && ctx->dp.lpc != 4) {
        ctx->last_error = "set_integrator_form: the sixteen-lanes-per-chain form exists for problems of 3 or 4 age classes";
        return SEPAIHRD_E_UNSUPPORTED;
    }
    ctx->dp.form = form;
    return SEPAIHRD_OK;
}

int sepaihrd_ensemble_quantiles(sepaihrd_ctx* ctx, const double* theta, int S, const double* probs, int n_probs,
                                double* ppc_quantiles, double* sero_quantiles, double* rt_quantiles, double* metrics,
                                int32_t* status, int32_t* n_valid) {
    if (!ctx) return SEPAIHRD_E_INVALID_ARG;
    if (S <= 0 || !theta || !probs || n_probs <= 0 || n_probs > 1024 || !ppc_quantiles) {
        ctx->last_error = "ensemble_quantiles: need S > 0, theta, probs (1..1024) and ppc_quantiles";
        return SEPAIHRD_E_INVALID_ARG;
    }
    for (int p = 0; p < n_probs; ++p)
        if (!(probs[p] >= 0.0 && probs[p] <= 1.0)) {
            ctx->last_error = "ensemble_quantiles: probabilities must lie in [0, 1]";
            return SEPAIHRD_E_INVALID_ARG;
        }
    if (ctx->pending_B > 0) {
        ctx->last_error = "ensemble_quantiles: a sepaihrd_eval_batch_begin is pending on this context";
        return SEPAIHRD_E_INVALID_ARG;
    }
    if (ctx->precision != SEPAIHRD_PRECISION_F64) {
        ctx->last_error = "ensemble_quantiles: the ensemble summaries read the fp64 integrator's parked increments (set precision F64)";
        return SEPAIHRD_E_UNSUPPORTED;
    }
    HIP_TRY(hipSetDevice(ctx->device), ctx, return SEPAIHRD_E_HIP);
    const DevProblem& dp = ctx->dp;
    const int Tp = dp.T - dp.runup_offset;
    if (Tp <= 0) {
        ctx->last_error = "ensemble_quantiles: no output time >= 0";
        return SEPAIHRD_E_INVALID_ARG;
    }
    int S_pad = WAVE;
    while (S_pad < S && S_pad < ENSEMBLE_MAX_SAMPLES) S_pad <<= 1;
    const bool big = S > ENSEMBLE_MAX_SAMPLES;  // segments sorted in global memory instead of LDS
    if (big) S_pad = (S + WAVE - 1) / WAVE * WAVE;
    const size_t cpw = (size_t)(WAVE / dp.lpc);
    const size_t chains = ((size_t)S + cpw - 1) / cpw * cpw;
    int rc = ensure_workspace(ctx, chains);
    if (rc != SEPAIHRD_OK) return rc;

    const bool want_sero = sero_quantiles != nullptr, want_metrics = metrics != nullptr;
    const bool want_rt = rt_quantiles != nullptr || want_metrics;  // the metric table reads the Rt values
    const bool want_traj = want_sero || want_rt;
    if (want_rt && dp.n > 16) {
        ctx->last_error = "ensemble_quantiles: Rt trajectories are built for at most 16 age classes";
        return SEPAIHRD_E_UNSUPPORTED;
    }
    const size_t n_ppc = (size_t)6 * n_probs * Tp * dp.n;
    const size_t n_sero = want_sero ? (size_t)n_probs * dp.T : 0;
    const size_t n_rt = want_rt ? (size_t)n_probs * dp.T : 0;
    const size_t n_vals = ((size_t)6 * Tp * dp.n + (want_sero ? dp.T : 0) + (want_rt ? dp.T : 0)) * S_pad;
    const size_t n_traj = want_traj ? (size_t)S * dp.T * NUM_COMP * dp.n : 0;
    double *d_theta = nullptr, *d_ll = nullptr, *d_vals = nullptr, *d_traj = nullptr, *d_probs = nullptr, *d_q = nullptr,
           *d_metrics = nullptr, *d_scratch = nullptr;
    // scratch of the global sort: up to 2 GiB, at least one segment
    const size_t n_scratch = big ? std::max<size_t>((size_t)S_pad, std::min<size_t>(n_vals, (size_t)1 << 28) / S_pad * S_pad) : 0;
    const size_t n_metrics = want_metrics ? (size_t)S * (12 + 4 * dp.n) : 0;
    int32_t* d_nv = nullptr;
    auto cleanup = [&]() {};  // the buffers stay with the context
    int slot = 0;
    auto dalloc = [&](void** p, size_t bytes) {
        const int k = slot++;
        if (bytes == 0) bytes = 8;
        if (ctx->ens_cap[k] < bytes) {
            if (ctx->ens_buf[k]) (void)hipFree(ctx->ens_buf[k]);
            ctx->ens_buf[k] = nullptr;
            ctx->ens_cap[k] = 0;
            if (hipMalloc(&ctx->ens_buf[k], bytes) != hipSuccess) return false;
            ctx->ens_cap[k] = bytes;
        }
        *p = ctx->ens_buf[k];
        return true;
    };
    if (!dalloc((void**)&d_theta, (size_t)S * ctx->P * sizeof(double)) || !dalloc((void**)&d_ll, (size_t)S * sizeof(double)) ||
        !dalloc((void**)&d_vals, n_vals * sizeof(double)) || !dalloc((void**)&d_traj, n_traj * sizeof(double)) ||
        !dalloc((void**)&d_probs, (size_t)n_probs * sizeof(double)) || !dalloc((void**)&d_q, (n_ppc + n_sero + n_rt) * sizeof(double)) ||
        !dalloc((void**)&d_nv, sizeof(int32_t)) || !dalloc((void**)&d_metrics, n_metrics * sizeof(double)) ||
        !dalloc((void**)&d_scratch, n_scratch * sizeof(double))) {
        cleanup();
        ctx->last_error = "ensemble_quantiles: device allocation failed";
        return SEPAIHRD_E_HIP;
    }
    HIP_TRY(hipMemcpy(d_theta, theta, (size_t)S * ctx->P * sizeof(double), hipMemcpyHostToDevice), ctx,
            { cleanup(); return SEPAIHRD_E_HIP; });
    HIP_TRY(hipMemcpy(d_probs, probs, (size_t)n_probs * sizeof(double), hipMemcpyHostToDevice), ctx,
            { cleanup(); return SEPAIHRD_E_HIP; });
    EvalOutputs out{d_ll, nullptr, nullptr, nullptr, nullptr, want_traj ? d_traj : nullptr,
                    ctx->ws_cum, ctx->ws_rows, ctx->ws_status, nullptr, 1};
    rc = fence_before(ctx, nullptr);  // an evaluation of this context may still be running on another stream
    if (rc != SEPAIHRD_OK) return rc;
    rc = ctx->arith == SEPAIHRD_ARITH_FMA ? launch_eval_fma(dp, ctx->solver, d_theta, S, out, nullptr)
                                          : launch_eval_strict(dp, ctx->solver, d_theta, S, out, nullptr);
    if (rc != 0) {
        cleanup();
        ctx->last_error = rc == -4 ? "unsupported lanes-per-chain" : "kernel launch failed";
        return rc == -4 ? SEPAIHRD_E_UNSUPPORTED : SEPAIHRD_E_HIP;
    }
    double total_pop = 0.0;
    for (int i = 0; i < dp.n; ++i) total_pop += ctx->host_N[(size_t)i];
    EnsembleArgs a{};
    a.S = S; a.S_pad = S_pad; a.lpc = dp.lpc; a.n = dp.n; a.T = dp.T; a.Tp = Tp; a.runup_offset = dp.runup_offset;
    a.n_probs = n_probs;
    a.cum_stride = chains * dp.lpc;
    a.cum = ctx->ws_cum; a.wstatus = ctx->ws_status; a.traj = want_traj ? d_traj : nullptr;
    a.total_pop = total_pop;
    a.vals = d_vals; a.probs = d_probs; a.q_out = d_q; a.sero_out = want_sero ? d_q + n_ppc : nullptr; a.n_valid = d_nv;
    a.rt_out = want_rt ? d_q + n_ppc + n_sero : nullptr;
    a.rt_segment0 = 6 * Tp * dp.n + (want_sero ? dp.T : 0);
    a.pb = &ctx->dp;
    a.theta = d_theta;
    a.metrics_out = want_metrics ? d_metrics : nullptr;
    a.sort_scratch = big ? d_scratch : nullptr;
    a.sort_scratch_doubles = n_scratch;
    rc = launch_ensemble_summaries(a, nullptr);
    if (rc != 0) {
        cleanup();
        ctx->last_error = "ensemble summary launch failed";
        return SEPAIHRD_E_HIP;
    }
    HIP_TRY(hipDeviceSynchronize(), ctx, { cleanup(); return SEPAIHRD_E_HIP; });
    HIP_TRY(hipMemcpy(ppc_quantiles, d_q, n_ppc * sizeof(double), hipMemcpyDeviceToHost), ctx, { cleanup(); return SEPAIHRD_E_HIP; });
    if (want_sero)
        HIP_TRY(hipMemcpy(sero_quantiles, d_q + n_ppc, n_sero * sizeof(double), hipMemcpyDeviceToHost), ctx,
                { cleanup(); return SEPAIHRD_E_HIP; });
    if (rt_quantiles)
        HIP_TRY(hipMemcpy(rt_quantiles, d_q + n_ppc + n_sero, n_rt * sizeof(double), hipMemcpyDeviceToHost), ctx,
                { cleanup(); return SEPAIHRD_E_HIP; });
    if (want_metrics)
        HIP_TRY(hipMemcpy(metrics, d_metrics, n_metrics * sizeof(double), hipMemcpyDeviceToHost), ctx,
                { cleanup(); return SEPAIHRD_E_HIP; });
    if (status)
        HIP_TRY(hipMemcpy(status, ctx->ws_status, (size_t)S * sizeof(int32_t), hipMemcpyDeviceToHost), ctx,
                { cleanup(); return SEPAIHRD_E_HIP; });
    if (n_valid)
        HIP_TRY(hipMemcpy(n_valid, d_nv, sizeof(int32_t), hipMemcpyDeviceToHost), ctx, { cleanup(); return SEPAIHRD_E_HIP; });
    cleanup();
    return SEPAIHRD_OK;
}

int sepaihrd_apply_constraints(const sepaihrd_ctx* ctx, int mode, const double* in, int B, double* out) {
    if (!ctx || !in || !out || B < 0) return SEPAIHRD_E_INVALID_ARG;
    const int P = ctx->P;
    for (int b = 0; b < B; ++b)
        for (int p = 0; p < P; ++p) {
            const double v = in[(size_t)b * P + p];
            double r;
            if (ctx->has_bounds[p]) {
                double lo = ctx->lower[p], hi = ctx->upper[p];
                if (lo > hi) std::swap(lo, hi);
                r = mode == SEPAIHRD_CONSTRAINT_CLAMP ? std::min(std::max(v, lo), hi) : reflect_bound_host(v, lo, hi);
            } else {
                r = mode == SEPAIHRD_CONSTRAINT_CLAMP ? std::max(0.0, v) : std::abs(v);
            }
            out[(size_t)b * P + p] = r;
        }
    return SEPAIHRD_OK;
}

int sepaihrd_get_kernel_info_for_batch(sepaihrd_ctx* ctx, int32_t batch_chains, sepaihrd_kernel_info* info) {
    if (!ctx || !info) return SEPAIHRD_E_INVALID_ARG;
    std::memset(info, 0, sizeof(*info));
    HIP_TRY(hipSetDevice(ctx->device), ctx, return SEPAIHRD_E_HIP);
    LaunchInfo li{};
    const int rc = ctx->precision == SEPAIHRD_PRECISION_F32 ? kernel_info_f32(ctx->dp, ctx->solver, &li)
                   : ctx->arith == SEPAIHRD_ARITH_FMA       ? kernel_info_fma(ctx->dp, ctx->solver, batch_chains, &li)
                                                            : kernel_info_strict(ctx->dp, ctx->solver, batch_chains, &li);
    if (rc != 0) { ctx->last_error = "kernel_info failed"; return SEPAIHRD_E_HIP; }
    info->lanes_per_chain = li.lanes_per_chain;
    info->chains_per_wave = WAVE / li.lanes_per_chain;
    info->block_threads = WAVE;
    info->vgprs = li.vgprs; info->sgprs = li.sgprs; info->scratch_bytes = li.scratch;
    info->lds_bytes = li.lds_static + li.lds_dynamic;
    info->max_blocks_per_cu = li.max_blocks_per_cu;
    info->likelihood_form = li.likelihood_form;
    info->phase_pass_applied = ctx->precision == SEPAIHRD_PRECISION_F32 ? 0  // the fp32-state kernel does not go through the pass
                               : ctx->arith == SEPAIHRD_ARITH_FMA       ? phase_pass_applied_fma()
                                                                        : phase_pass_applied_strict();
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, ctx->device), ctx, return SEPAIHRD_E_HIP);
    info->num_cus = prop.multiProcessorCount;
    std::snprintf(info->kernel_name, sizeof(info->kernel_name), "%s lpc=%d solver=%d", li.name, li.lanes_per_chain,
                  ctx->solver);
    std::snprintf(info->device_name, sizeof(info->device_name), "%s (%s)", prop.name, prop.gcnArchName);
    return SEPAIHRD_OK;
}

int sepaihrd_get_kernel_info(sepaihrd_ctx* ctx, sepaihrd_kernel_info* info) {
    return sepaihrd_get_kernel_info_for_batch(ctx, 0, info);
}

// ------------------------------------------------------------------ device-resident Adaptive Metropolis
struct sepaihrd_mh {
    sepaihrd_ctx* ctx = nullptr;
    SamplerState st{};
    int rows = 0;  // history rows written so far
    hipStream_t stream = nullptr;  // own non-blocking stream: several samplers (one per host thread) overlap
    double* d_z = nullptr;
    double* d_scale = nullptr;
    double* d_loglik = nullptr;
    int32_t* d_status = nullptr;
    uint8_t* d_accept = nullptr;
    int32_t* d_rows = nullptr;
    double* d_gather = nullptr;
    size_t gather_cap = 0;
    // sepaihrd_mh_stage_normals / sepaihrd_mh_step: a second normals buffer filled by a copy stream while the
    // evaluation runs, and ONE packed upload per iteration (accept flags, scales, re-drawn rows) from pinned memory
    double* d_z_stage = nullptr;
    double* d_lz = nullptr;          // [2][C][P] L z of both continuations, formed beside the evaluation (sampler_lz)
    bool lz_ready = false;           // d_lz holds the products of the normals staged for the coming test
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_staged = nullptr;
    bool staged = false;
    double* h_stage[2] = {nullptr, nullptr};  // pinned [C][P] each: the caller fills one while the other's copy may still run
    int stage_turn = 0;
    uint8_t* h_pack = nullptr;   // pinned
    void* h_fetch = nullptr;     // pinned mirror of [loglik C][status C] (one allocation on the device too: one copy per fetch)
    // accept test on the device (sepaihrd_mh_step_tested): current and best values per chain, the test's inputs
    // [log_u C][scale_reject C][scale_accept C][z_plain C*P] (page-locked mirror filled by the caller while the evaluation
    // runs), its outputs [values C doubles][flags C bytes] with their page-locked mirror, the scale it selected
    double* d_lp = nullptr;
    double* d_best_lp = nullptr;
    double* d_test = nullptr;
    double* h_test = nullptr;
    double* d_scale_sel = nullptr;
    void* d_test_out = nullptr;
    void* h_test_out = nullptr;
    hipEvent_t ev_test_up = nullptr, ev_tested = nullptr, ev_fetched = nullptr, ev_proposed = nullptr;
    hipEvent_t ev_last_proposed = nullptr;  // whichever of the two marks the end of the last proposal
    bool values_set = false, test_pending = false, proposed_once = false;
    uint8_t* d_pack = nullptr;
    size_t pack_bytes = 0, off_scale = 0, off_chain = 0, off_rows = 0;
    // rank-one covariance updates not yet applied (see mh_rank1_catchup_cov_kernel): the state each one reads and its
    // gamma, in the order they were asked for.  Uploaded as [rows n int32 | gammas n doubles] from a page-locked buffer.
    std::vector<int32_t> pending_row;
    std::vector<double> pending_gamma;
    void* d_r1 = nullptr;
    void* h_r1 = nullptr;
    size_t r1_cap = 0;
    hipEvent_t ev_r1 = nullptr;  // the last upload out of h_r1 has been copied
    bool r1_in_flight = false;
    // states [mom_rows, rows) have not entered the running sums yet (see mh_moments_catchup_kernel)
    int mom_rows = 0;
    bool device_rng = false;  // the chains' mt19937 streams live on the device (sepaihrd_mh_seed_streams)
    // a self-contained sampler lets its caller queue iterations ahead: at most ~128 of them (an event every 32 steps, four kept)
    hipEvent_t ev_ahead[4] = {nullptr, nullptr, nullptr, nullptr};
    bool ev_ahead_used[4] = {false, false, false, false};
    long auto_steps = 0;
    int covariance_mode = SEPAIHRD_MH_COV_RUNNING;
    int iterations = 0;
    double* d_summary = nullptr;
    // sepaihrd_mh_snapshot_begin / _end: a gather on the sampler's stream, the copy home on a stream of its own
    hipStream_t snap_stream = nullptr;
    hipEvent_t ev_snap_gathered = nullptr, ev_snap_done = nullptr;
    double* d_snap = nullptr;
    double* h_snap = nullptr;   // page-locked
    int32_t* d_snap_chains = nullptr;
    size_t snap_cap = 0, snap_chain_cap = 0;
    int snap_n = 0, snap_count = 0;
    bool snap_pending = false;
    std::vector<int32_t> snap_chain_list;  // what d_snap_chains holds
    std::vector<void*> allocs;
};

namespace {
// queue the rank-one update that precedes the next proposal: it reads the newest state (row rows - 1) with this gamma
void mh_queue_rank1(sepaihrd_mh* mh, double gamma) {
    mh->pending_row.push_back(mh->rows - 1);
    mh->pending_gamma.push_back(gamma);
}
// apply the queued updates in order (someone is about to read the covariance, or the ring is about to overwrite a
// state they read)
int mh_flush_rank1(sepaihrd_mh* mh) {
    const size_t n = mh->pending_gamma.size();
    if (n == 0) return 0;
    for (size_t k = 0; k < n; ++k)  // every state still in the ring (mh_before_commit keeps it so)
        if (mh->pending_row[k] < 0 || mh->pending_row[k] >= mh->rows || mh->pending_row[k] < mh->rows - mh->st.window) return -1;
    if (mh->r1_in_flight) {  // the page-locked buffer is still the source of the previous upload
        if (hipEventSynchronize(mh->ev_r1) != hipSuccess) return -3;
        mh->r1_in_flight = false;
    }
    if (n > mh->r1_cap) {
        const size_t cap = std::max<size_t>(2 * n, 256);
        if (mh->d_r1) (void)hipFree(mh->d_r1);
        if (mh->h_r1) (void)hipHostFree(mh->h_r1);
        mh->d_r1 = mh->h_r1 = nullptr;
        mh->r1_cap = 0;
        if (hipMalloc(&mh->d_r1, cap * (sizeof(double) + sizeof(int32_t))) != hipSuccess) return -3;
        if (hipHostMalloc(&mh->h_r1, cap * (sizeof(double) + sizeof(int32_t)), hipHostMallocDefault) != hipSuccess) return -3;
        mh->r1_cap = cap;
    }
    // [gammas cap doubles | rows cap int32]: both aligned whatever n
    std::memcpy(mh->h_r1, mh->pending_gamma.data(), n * sizeof(double));
    std::memcpy(static_cast<char*>(mh->h_r1) + mh->r1_cap * sizeof(double), mh->pending_row.data(), n * sizeof(int32_t));
    if (hipMemcpyAsync(mh->d_r1, mh->h_r1, mh->r1_cap * sizeof(double) + n * sizeof(int32_t), hipMemcpyHostToDevice, mh->stream) != hipSuccess) return -3;
    if (hipEventRecord(mh->ev_r1, mh->stream) != hipSuccess) return -3;
    mh->r1_in_flight = true;
    const int rc = sampler_rank1_catchup(mh->st, reinterpret_cast<const int32_t*>(static_cast<char*>(mh->d_r1) + mh->r1_cap * sizeof(double)),
                                         static_cast<const double*>(mh->d_r1), (int)n, mh->stream);
    mh->pending_gamma.clear();
    mh->pending_row.clear();
    return rc;
}
// the states committed since the last catch-up enter the running sums; emit_len > 0: and the refresh of
// recomputeFullCovariance for a history of emit_len states follows in the same launch
int mh_flush_moments(sepaihrd_mh* mh, int emit_len) {
    if (mh->covariance_mode != SEPAIHRD_MH_COV_RUNNING) return 0;
    const int n = mh->rows - mh->mom_rows;
    if (n <= 0 && emit_len <= 0) return 0;
    const int rc = sampler_moments_catchup(mh->st, mh->mom_rows, n, emit_len, mh->stream);
    mh->mom_rows = mh->rows;
    return rc;
}
// Before state `rows` is written into its ring slot: whatever is queued on the state that slot still holds runs first.
int mh_before_commit(sepaihrd_mh* mh) {
    const int oldest_kept = mh->rows + 1 - mh->st.window;  // after the commit the ring holds states oldest_kept .. rows
    int rc = 0;
    if (mh->covariance_mode == SEPAIHRD_MH_COV_RUNNING && mh->mom_rows < oldest_kept) rc = mh_flush_moments(mh, 0);
    if (rc == 0 && !mh->pending_row.empty() && mh->pending_row.front() < oldest_kept) rc = mh_flush_rank1(mh);
    return rc;
}
// the adaptation step before a proposal: 0 none, 1 rank-one update, 2 + Cholesky refresh of cov + eps I, 3 + full
// recompute first (which overwrites covariance and mean: queued rank-one updates are dropped unapplied)
int mh_adapt_step(sepaihrd_mh* mh, double gamma, int adapt) {
    int rc = 0;
    if (adapt >= 1) mh_queue_rank1(mh, gamma);
    if (adapt == 2) {
        rc = mh_flush_rank1(mh);
        if (rc == 0) rc = sampler_cholesky(mh->st, mh->st.reg_eps, 0, mh->stream);  // :295-300
    } else if (adapt == 3) {
        mh->pending_gamma.clear();
        mh->pending_row.clear();
        if (mh->covariance_mode == SEPAIHRD_MH_COV_RUNNING) rc = mh_flush_moments(mh, mh->rows);
        else rc = sampler_full_covariance(mh->st, mh->rows, mh->stream);
        if (rc == 0) rc = sampler_cholesky_refresh(mh->st, mh->stream);  // :190-197 and :295-300, each kept on success
    }
    return rc;
}

// log-likelihoods (and statuses) of the last evaluation: one copy into the page-locked mirror, the wait, two memcpy
int mh_fetch_values(sepaihrd_mh* mh, double* loglik, int32_t* status) {
    sepaihrd_ctx* ctx = mh->ctx;
    const size_t C = (size_t)mh->st.C;
    const size_t bytes = C * sizeof(double) + (status ? C * sizeof(int32_t) : 0);
    HIP_TRY(hipMemcpyAsync(mh->h_fetch, mh->d_loglik, bytes, hipMemcpyDeviceToHost, mh->stream), ctx, return SEPAIHRD_E_HIP);
    HIP_TRY(hipStreamSynchronize(mh->stream), ctx, return SEPAIHRD_E_HIP);
    std::memcpy(loglik, mh->h_fetch, C * sizeof(double));
    if (status) std::memcpy(status, static_cast<char*>(mh->h_fetch) + C * sizeof(double), C * sizeof(int32_t));
    return SEPAIHRD_OK;
}

int mh_eval(sepaihrd_mh* mh, const double* d_theta, double* loglik, int32_t* status) {
    sepaihrd_ctx* ctx = mh->ctx;
    const int C = mh->st.C;
    const int rc = sepaihrd_eval_batch_device(ctx, d_theta, C, mh->d_loglik, mh->d_status, nullptr, nullptr, nullptr, nullptr, mh->stream);
    if (rc != SEPAIHRD_OK) return rc;
    return mh_fetch_values(mh, loglik, status);
}
}  // namespace

sepaihrd_mh* sepaihrd_mh_create(sepaihrd_ctx* ctx, const sepaihrd_mh_config* cfg, const double* x0, const double* cov0) {
    if (!ctx) return nullptr;
    const int P = ctx->P;
    if (!cfg || cfg->chains <= 0 || cfg->iterations <= 0 || !x0 || !cov0 || P > 200 ||
        (cfg->covariance_mode != SEPAIHRD_MH_COV_RUNNING && cfg->covariance_mode != SEPAIHRD_MH_COV_TWO_PASS)) {
        ctx->last_error = "mh_create: need chains > 0, iterations > 0, a known covariance_mode, x0, cov0 and at most 200 parameters";
        return nullptr;
    }
    const int C = cfg->chains;
    const double reg_eps = cfg->reg_eps, scaling_factor = cfg->scaling_factor;
    // ring of the newest states: every state of the run for the two-pass refresh, else the window asked for
    // (at least 2: the newest state and the one a queued update may still read), never more than the run has
    int window = cfg->covariance_mode == SEPAIHRD_MH_COV_TWO_PASS ? cfg->iterations
                                                                   : std::min(cfg->iterations, std::max(cfg->adaptation_window > 0 ? cfg->adaptation_window : 128, 2));
    const int thinning = cfg->thinning > 0 ? cfg->thinning : 0;
    const int n_store = thinning > 0 ? (cfg->iterations - 1) / thinning + 1 : 0;
    if (hipSetDevice(ctx->device) != hipSuccess) { ctx->last_error = "hipSetDevice failed"; return nullptr; }
    const size_t CP = (size_t)C * P, CPP = CP * P;
    {
        // say what the state costs before allocating instead of failing somewhere inside
        const double need = (double)CP * ((double)window + (double)n_store) * 8.0 + 3.0 * (double)CPP * 8.0 + 10.0 * (double)CP * 8.0;  // ... + the C*P arrays (state, proposal, means, sums, normals, L z)
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && need > (double)free_b) {
            char msg[320];
            std::snprintf(msg, sizeof(msg),
                          "mh_create: sampler state needs %.3f GB (ring of newest states C*window*P = %d*%d*%d doubles, samples "
                          "C*%d*P, covariance + factor + second moment 3*C*P*P) but %.3f GB of device memory are free; run "
                          "fewer chains per sampler object%s",
                          need / 1e9, C, window, P, n_store, (double)free_b / 1e9,
                          cfg->covariance_mode == SEPAIHRD_MH_COV_TWO_PASS ? " or use SEPAIHRD_MH_COV_RUNNING (no full history)" : "");
            ctx->last_error = msg;
            return nullptr;
        }
    }
    auto* mh = new sepaihrd_mh();
    mh->ctx = ctx;
    SamplerState& st = mh->st;
    st.C = C; st.P = P; st.window = window; st.thinning = std::max(thinning, 1); st.n_store = n_store;
    st.scaling = scaling_factor; st.reg_eps = reg_eps;
    mh->covariance_mode = cfg->covariance_mode;
    mh->iterations = cfg->iterations;
    bool ok = true;
    auto dalloc = [&](void** p, size_t bytes) {
        if (!ok) return;
        if (hipMalloc(p, bytes) != hipSuccess) { ok = false; return; }
        mh->allocs.push_back(*p);
    };
    dalloc((void**)&st.x, CP * sizeof(double));
    dalloc((void**)&st.prop, CP * sizeof(double));
    dalloc((void**)&st.cov, CPP * sizeof(double));
    dalloc((void**)&st.chol, CPP * sizeof(double));
    dalloc((void**)&st.mean, CP * sizeof(double));
    dalloc((void**)&st.best, CP * sizeof(double));
    dalloc((void**)&st.hist, CP * (size_t)window * sizeof(double));
    if (n_store > 0) dalloc((void**)&st.store, CP * (size_t)n_store * sizeof(double));
    dalloc((void**)&st.sum, CP * sizeof(double));
    dalloc((void**)&st.wmean, CP * sizeof(double));
    dalloc((void**)&st.m2, CPP * sizeof(double));
    dalloc((void**)&st.accepted, (size_t)C * sizeof(int32_t));
    dalloc((void**)&st.fail_counts, 3 * sizeof(uint32_t));
    dalloc((void**)&st.mt, (size_t)C * 624 * sizeof(uint32_t));
    dalloc((void**)&st.mt_idx, (size_t)C * sizeof(int32_t));
    dalloc((void**)&st.mt_used, (size_t)C * 2 * sizeof(int32_t));
    dalloc((void**)&mh->d_summary, (size_t)C * (2 * (size_t)P + 2) * sizeof(double));
    dalloc((void**)&mh->d_z, CP * sizeof(double));
    dalloc((void**)&mh->d_scale, (size_t)C * sizeof(double));
    dalloc((void**)&mh->d_loglik, (size_t)C * (sizeof(double) + sizeof(int32_t)));
    mh->d_status = mh->d_loglik ? reinterpret_cast<int32_t*>(mh->d_loglik + C) : nullptr;
    dalloc((void**)&mh->d_accept, (size_t)C);
    dalloc((void**)&mh->d_z_stage, CP * sizeof(double));
    dalloc((void**)&mh->d_lz, 2 * CP * sizeof(double));
    dalloc((void**)&mh->d_lp, (size_t)C * sizeof(double));
    dalloc((void**)&mh->d_best_lp, (size_t)C * sizeof(double));
    dalloc((void**)&mh->d_test, (3 * (size_t)C + CP) * sizeof(double));
    dalloc((void**)&mh->d_scale_sel, (size_t)C * sizeof(double));
    dalloc(&mh->d_test_out, (size_t)C * (sizeof(double) + 1));
    {
        auto up8 = [](size_t v) { return (v + 7) & ~(size_t)7; };
        mh->off_scale = up8((size_t)C);
        mh->off_chain = mh->off_scale + (size_t)C * sizeof(double);
        mh->off_rows = up8(mh->off_chain + (size_t)C * sizeof(int32_t));
        mh->pack_bytes = mh->off_rows + CP * sizeof(double);
        dalloc((void**)&mh->d_pack, mh->pack_bytes);
        if (ok && hipHostMalloc((void**)&mh->h_pack, mh->pack_bytes, hipHostMallocDefault) != hipSuccess) { mh->h_pack = nullptr; ok = false; }
        if (ok && hipHostMalloc(&mh->h_fetch, (size_t)C * (sizeof(double) + sizeof(int32_t)), hipHostMallocDefault) != hipSuccess) { mh->h_fetch = nullptr; ok = false; }
        if (ok && hipHostMalloc((void**)&mh->h_test, (3 * (size_t)C + CP) * sizeof(double), hipHostMallocDefault) != hipSuccess) { mh->h_test = nullptr; ok = false; }
        if (ok && hipHostMalloc(&mh->h_test_out, (size_t)C * (sizeof(double) + 1), hipHostMallocDefault) != hipSuccess) { mh->h_test_out = nullptr; ok = false; }
        for (int b = 0; b < 2; ++b)
            if (ok && hipHostMalloc((void**)&mh->h_stage[b], CP * sizeof(double), hipHostMallocDefault) != hipSuccess) { mh->h_stage[b] = nullptr; ok = false; }
    }
    if (ok && sepaihrd_reserve(ctx, C) != SEPAIHRD_OK) ok = false;
    if (ok && hipStreamCreateWithFlags(&mh->stream, hipStreamNonBlocking) != hipSuccess) ok = false;
    if (ok && hipStreamCreateWithFlags(&mh->copy_stream, hipStreamNonBlocking) != hipSuccess) ok = false;
    if (ok && hipEventCreateWithFlags(&mh->ev_staged, hipEventDisableTiming) != hipSuccess) ok = false;
    for (hipEvent_t* e : {&mh->ev_test_up, &mh->ev_tested, &mh->ev_fetched, &mh->ev_proposed, &mh->ev_r1})
        if (ok && hipEventCreateWithFlags(e, hipEventDisableTiming) != hipSuccess) ok = false;
    if (ok) ok = hipMemsetAsync(mh->d_test_out, 0, (size_t)C * (sizeof(double) + 1), mh->stream) == hipSuccess;
    if (ok) ok = hipMemsetAsync(st.sum, 0, CP * sizeof(double), mh->stream) == hipSuccess &&
                 hipMemsetAsync(st.wmean, 0, CP * sizeof(double), mh->stream) == hipSuccess &&
                 hipMemsetAsync(st.m2, 0, CPP * sizeof(double), mh->stream) == hipSuccess &&
                 hipMemsetAsync(st.accepted, 0, (size_t)C * sizeof(int32_t), mh->stream) == hipSuccess &&
                 hipMemsetAsync(st.fail_counts, 0, 3 * sizeof(uint32_t), mh->stream) == hipSuccess;
    if (ok) {
        std::vector<double> cov_all(CPP);
        for (int c = 0; c < C; ++c) std::copy(cov0, cov0 + (size_t)P * P, cov_all.begin() + (size_t)c * P * P);
        ok = hipMemcpy(st.x, x0, CP * sizeof(double), hipMemcpyHostToDevice) == hipSuccess &&
             hipMemcpy(st.mean, x0, CP * sizeof(double), hipMemcpyHostToDevice) == hipSuccess &&
             hipMemcpy(st.best, x0, CP * sizeof(double), hipMemcpyHostToDevice) == hipSuccess &&
             hipMemcpy(st.cov, cov_all.data(), CPP * sizeof(double), hipMemcpyHostToDevice) == hipSuccess &&
             hipMemsetAsync(st.chol, 0, CPP * sizeof(double), mh->stream) == hipSuccess;
    }
    if (ok) ok = sampler_cholesky(st, 0.0, 1, mh->stream) == 0 && sampler_commit(st, nullptr, 0, mh->stream) == 0 &&
                 hipStreamSynchronize(mh->stream) == hipSuccess;
    if (!ok) {
        ctx->last_error = "mh_create: device allocation or initialisation failed";
        for (void* p : mh->allocs) (void)hipFree(p);
        if (mh->h_pack) (void)hipHostFree(mh->h_pack);
        if (mh->h_fetch) (void)hipHostFree(mh->h_fetch);
        if (mh->h_test) (void)hipHostFree(mh->h_test);
        if (mh->h_test_out) (void)hipHostFree(mh->h_test_out);
        for (double* b : mh->h_stage) if (b) (void)hipHostFree(b);
        if (mh->ev_staged) (void)hipEventDestroy(mh->ev_staged);
        for (hipEvent_t e : {mh->ev_test_up, mh->ev_tested, mh->ev_fetched, mh->ev_proposed, mh->ev_r1}) if (e) (void)hipEventDestroy(e);
        if (mh->copy_stream) (void)hipStreamDestroy(mh->copy_stream);
        if (mh->stream) (void)hipStreamDestroy(mh->stream);
        delete mh;
        return nullptr;
    }
    mh->rows = 1;
    ctx->lazy_stream = mh->stream;
    return mh;
}

void sepaihrd_mh_destroy(sepaihrd_mh* mh) {
    if (!mh) return;
    (void)hipSetDevice(mh->ctx->device);
    if (mh->copy_stream) { (void)hipStreamSynchronize(mh->copy_stream); (void)hipStreamDestroy(mh->copy_stream); }
    if (mh->stream) {
        (void)hipStreamSynchronize(mh->stream);
        sepaihrd_ctx* ctx = mh->ctx;
        if (ctx->lazy_stream == mh->stream) ctx->lazy_stream = nullptr;
        if (ctx->busy_stream == mh->stream) { ctx->busy_valid = false; ctx->busy_stream = nullptr; }  // all of it is done
        (void)hipStreamDestroy(mh->stream);
    }
    if (mh->ev_staged) (void)hipEventDestroy(mh->ev_staged);
    for (hipEvent_t e : {mh->ev_test_up, mh->ev_tested, mh->ev_fetched, mh->ev_proposed, mh->ev_r1}) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : mh->ev_ahead) if (e) (void)hipEventDestroy(e);
    if (mh->h_pack) (void)hipHostFree(mh->h_pack);
    if (mh->h_fetch) (void)hipHostFree(mh->h_fetch);
    if (mh->h_test) (void)hipHostFree(mh->h_test);
    if (mh->h_test_out) (void)hipHostFree(mh->h_test_out);
    for (double* b : mh->h_stage) if (b) (void)hipHostFree(b);
    for (void* p : mh->allocs) (void)hipFree(p);
    if (mh->snap_stream) { (void)hipStreamSynchronize(mh->snap_stream); (void)hipStreamDestroy(mh->snap_stream); }
    for (hipEvent_t e : {mh->ev_snap_gathered, mh->ev_snap_done}) if (e) (void)hipEventDestroy(e);
    if (mh->d_snap) (void)hipFree(mh->d_snap);
    if (mh->h_snap) (void)hipHostFree(mh->h_snap);
    if (mh->d_snap_chains) (void)hipFree(mh->d_snap_chains);
    if (mh->d_rows) (void)hipFree(mh->d_rows);
    if (mh->d_gather) (void)hipFree(mh->d_gather);
    if (mh->d_r1) (void)hipFree(mh->d_r1);
    if (mh->h_r1) (void)hipHostFree(mh->h_r1);
    delete mh;
}

int sepaihrd_mh_history_length(const sepaihrd_mh* mh) { return mh ? mh->rows : 0; }

int sepaihrd_mh_evaluate_current(sepaihrd_mh* mh, double* loglik, int32_t* status) {
    if (!mh || !loglik) return SEPAIHRD_E_INVALID_ARG;
    HIP_TRY(hipSetDevice(mh->ctx->device), mh->ctx, return SEPAIHRD_E_HIP);
    return mh_eval(mh, mh->st.x, loglik, status);
}

int sepaihrd_mh_propose(sepaihrd_mh* mh, const double* z, const double* scale, double* loglik, int32_t* status) {
    if (!mh || !z || !scale) return SEPAIHRD_E_INVALID_ARG;
    sepaihrd_ctx* ctx = mh->ctx;
    HIP_TRY(hipSetDevice(ctx->device), ctx, return SEPAIHRD_E_HIP);
    const size_t CP = (size_t)mh->st.C * mh->st.P;
    HIP_TRY(hipMemcpyAsync(mh->d_z, z, CP * sizeof(double), hipMemcpyHostToDevice, mh->stream), ctx, return SEPAIHRD_E_HIP);
    HIP_TRY(hipMemcpyAsync(mh->d_scale, scale, (size_t)mh->st.C * sizeof(double), hipMemcpyHostToDevice, mh->stream), ctx,
            return SEPAIHRD_E_HIP);
    if (sampler_propose(mh->st, ctx->dp, mh->d_z, mh->d_scale, mh->stream) != 0) {
        ctx->last_error = "mh_propose: launch failed";
        return SEPAIHRD_E_HIP;
    }
    if (!loglik)  // launch only: the caller overlaps host work and calls sepaihrd_mh_fetch
        return sepaihrd_eval_batch_device(ctx, mh->st.prop, mh->st.C, mh->d_loglik, mh->d_status, nullptr, nullptr, nullptr, nullptr,
                                          mh->stream);
    return mh_eval(mh, mh->st.prop, loglik, status);
}

double* sepaihrd_mh_staging_buffer(sepaihrd_mh* mh) { return mh ? mh->h_stage[mh->stage_turn] : nullptr; }

int sepaihrd_mh_stage_normals(sepaihrd_mh* mh, const double* z) {
    if (!mh || !z) return SEPAIHRD_E_INVALID_ARG;
    if (z == mh->h_stage[mh->stage_turn]) mh->stage_turn ^= 1;  // page-locked source: a real DMA; the next fill goes to the other buffer
    sepaihrd_ctx* ctx = mh->ctx;
    HIP_TRY(hipSetDevice(ctx->device), ctx, return SEPAIHRD_E_HIP);
    const size_t CP = (size_t)mh->st.C * mh->st.P;
    HIP_TRY(hipMemcpyAsync(mh->d_z_stage, z, CP * sizeof(double), hipMemcpyHostToDevice, mh->copy_stream), ctx, return SEPAIHRD_E_HIP);
    HIP_TRY(hipEventRecord(mh->ev_staged, mh->copy_stream), ctx, return SEPAIHRD_E_HIP);
    mh->staged = true;
    return SEPAIHRD_OK;
}

int sepaihrd_mh_step(sepaihrd_mh* mh, const uint8_t* accept, const double* scale, const int32_t* patch_chain, const double* patch_z,
                     int n_patch, double gamma, int adapt) {
    if (!mh || !scale || n_patch < 0 || (n_patch > 0 && (!patch_chain || !patch_z)) || adapt < 0 || adapt > 3) return SEPAIHRD_E_INVALID_ARG;
    sepaihrd_ctx* ctx = mh->ctx;
    const int C = mh->st.C, P = mh->st.P;
    if (!mh->staged) { ctx->last_error = "mh_step: no staged normals (call sepaihrd_mh_stage_normals first)"; return SEPAIHRD_E_INVALID_ARG; }
    if (n_patch > C) { ctx->last_error = "mh_step: more patched rows than chains"; return SEPAIHRD_E_INVALID_ARG; }
    for (int k = 0; k < n_patch; ++k)
        if (patch_chain[k] < 0 || patch_chain[k] >= C) { ctx->last_error = "mh_step: patched chain out of range"; return SEPAIHRD_E_INVALID_ARG; }
    if (accept && mh->rows >= mh->iterations) { ctx->last_error = "mh_step: more states than the sampler was created for"; return SEPAIHRD_E_INVALID_ARG; }
    HIP_TRY(hipSetDevice(ctx->device), ctx, return SEPAIHRD_E_HIP);
    hipStream_t st = mh->stream;
    // one upload: [accept C | scale C | chains n | rows n x P]
    if (accept) std::memcpy(mh->h_pack, accept, (size_t)C);
    std::memcpy(mh->h_pack + mh->off_scale, scale, (size_t)C * sizeof(double));
    if (n_patch > 0) {
        std::memcpy(mh->h_pack + mh->off_chain, patch_chain, (size_t)n_patch * sizeof(int32_t));
        // patch_z is a full [C][P] array of which the listed chains' rows are valid: gathered straight into the upload
        double* dst = reinterpret_cast<double*>(mh->h_pack + mh->off_rows);
        for (int k = 0; k < n_patch; ++k)
            std::memcpy(dst + (size_t)k * P, patch_z + (size_t)patch_chain[k] * P, (size_t)P * sizeof(double));
    }
    const size_t used = n_patch > 0 ? mh->off_rows + (size_t)n_patch * P * sizeof(double) : mh->off_chain;
    HIP_TRY(hipMemcpyAsync(mh->d_pack, mh->h_pack, used, hipMemcpyHostToDevice, st), ctx, return SEPAIHRD_E_HIP);
    int rc = 0;
    if (accept) {
        rc = mh_before_commit(mh);
        if (rc == 0) rc = sampler_commit(mh->st, mh->d_pack, mh->rows, st);
        if (rc == 0) mh->rows++;
    }
    if (rc == 0) rc = mh_adapt_step(mh, gamma, adapt);
    if (rc != 0) { ctx->last_error = "mh_step: launch failed"; return SEPAIHRD_E_HIP; }
    HIP_TRY(hipStreamWaitEvent(st, mh->ev_staged, 0), ctx, return SEPAIHRD_E_HIP);  // the staged normals have landed
    rc = sampler_patch_normals(mh->d_z_stage, reinterpret_cast<const int32_t*>(mh->d_pack + mh->off_chain),
                               reinterpret_cast<const double*>(mh->d_pack + mh->off_rows), n_patch, P, st);
    if (rc == 0) rc = sampler_propose(mh->st, ctx->dp, mh->d_z_stage, reinterpret_cast<const double*>(mh->d_pack + mh->off_scale), st);
    if (rc != 0) { ctx->last_error = "mh_step: launch failed"; return SEPAIHRD_E_HIP; }
    std::swap(mh->d_z, mh->d_z_stage);  // the next staging goes to the other buffer
    mh->staged = false;
    return sepaihrd_eval_batch_device(ctx, mh->st.prop, C, mh->d_loglik, mh->d_status, nullptr, nullptr, nullptr, nullptr, st);
}

int sepaihrd_mh_fetch(sepaihrd_mh* mh, double* loglik, int32_t* status) {
    if (!mh || !loglik) return SEPAIHRD_E_INVALID_ARG;
    sepaihrd_ctx* ctx = mh->ctx;
    HIP_TRY(hipSetDevice(ctx->device), ctx, return SEPAIHRD_E_HIP);
    return mh_fetch_values(mh, loglik, status);
}

int sepaihrd_mh_set_values(sepaihrd_mh* mh, const double* values) {
    if (!mh || !values) return SEPAIHRD_E_INVALID_ARG;
    sepaihrd_ctx* ctx = mh->ctx;
    HIP_TRY(hipSetDevice(ctx->device), ctx, return SEPAIHRD_E_HIP);
    const size_t bytes = (size_t)mh->st.C * sizeof(double);
    HIP_TRY(hipMemcpy(mh->d_lp, values, bytes, hipMemcpyHostToDevice), ctx, return SEPAIHRD_E_HIP);
    HIP_TRY(hipMemcpy(mh->d_best_lp, values, bytes, hipMemcpyHostToDevice), ctx, return SEPAIHRD_E_HIP);
    if (mh->st.lp_store)  // the value of sample 0 (:266-268)
        HIP_TRY(hipMemcpy2D(mh->st.lp_store, (size_t)mh->st.n_store * sizeof(double), values, sizeof(double), sizeof(double), (size_t)mh->st.C,
                            hipMemcpyHostToDevice), ctx, return SEPAIHRD_E_HIP);
    mh->values_set = true;
    return SEPAIHRD_OK;
}

double* sepaihrd_mh_test_buffer(sepaihrd_mh* mh) { return mh ? mh->h_test : nullptr; }

// The device's log / exp (csrc/sepaihrd_rng.inc) restate ONE libm build.  Before a sampler lets the device draw, they are
// evaluated on fixed arguments and compared, bit for bit, with the std::log / std::exp of THIS process.
int sepaihrd_device_libm_check(sepaihrd_ctx* ctx, int32_t* n_log_diff, int32_t* n_exp_diff) {
    if (!ctx) return SEPAIHRD_E_INVALID_ARG;
    if (ctx->libm_log_diff < 0) {
        HIP_TRY(hipSetDevice(ctx->device), ctx, return SEPAIHRD_E_HIP);
        const int N = sampler_libm_check_count();
        double* d_out = nullptr;
        HIP_TRY(hipMalloc((void**)&d_out, 4 * (size_t)N * sizeof(double)), ctx, return SEPAIHRD_E_HIP);
        std::vector<double> h(4 * (size_t)N);
        const bool ok = sampler_libm_check_values(d_out, nullptr) == 0 &&
                        hipMemcpy(h.data(), d_out, h.size() * sizeof(double), hipMemcpyDeviceToHost) == hipSuccess;
        (void)hipFree(d_out);
        if (!ok) { ctx->last_error = "device_libm_check: launch or copy failed"; return SEPAIHRD_E_HIP; }
        int dl = 0, de = 0;
        for (int i = 0; i < N; ++i) {
            // volatile: the comparison must be against the library call, not against a constant the compiler folded
            volatile double xl = h[(size_t)i], xe = h[2 * (size_t)N + i];
            const double hl = std::log(xl), he = std::exp(xe);
            if (std::memcmp(&hl, &h[(size_t)N + i], sizeof(double)) != 0) ++dl;
            if (std::memcmp(&he, &h[3 * (size_t)N + i], sizeof(double)) != 0) ++de;
        }
        // test hook: pretend this host's libm is another one (the fall-back to host-drawn streams is then exercised on a
        // box whose libm does match)
        if (const char* e = std::getenv("SEPAIHRD_LIBM_SELFCHECK")) if (std::string(e) == "fail") { dl += 1; de += 1; }
        ctx->libm_log_diff = dl;
        ctx->libm_exp_diff = de;
    }
    if (n_log_diff) *n_log_diff = ctx->libm_log_diff;
    if (n_exp_diff) *n_exp_diff = ctx->libm_exp_diff;
    return SEPAIHRD_OK;
}

namespace {
// does a batch of C chains fill the chip with two integrator waves per SIMD (the same threshold as launch_one's)?
bool mh_lz_ahead_wanted() {
    const char* e = std::getenv("SEPAIHRD_MH_LZ");
    return !(e && std::string(e) == "fused");
}
bool mh_draws_behind_the_evaluation(const sepaihrd_ctx* ctx, int C) {
    if (const char* e = std::getenv("SEPAIHRD_MH_DRAW")) {
        if (std::string(e) == "overlap") return false;
        if (std::string(e) == "serial") return true;
    }
    const long long chains_per_wave = WAVE / (ctx->dp.lpc > 0 ? ctx->dp.lpc : 1);
    return (C + chains_per_wave - 1) / chains_per_wave > 2 * 1024;  // more than one round of two waves per SIMD
}
}  // namespace

int sepaihrd_device_log_values(sepaihrd_ctx* ctx, const double* x, int32_t n, double* out) {
    if (!ctx || n < 0 || (n > 0 && (!x || !out))) return SEPAIHRD_E_INVALID_ARG;
    if (n == 0) return SEPAIHRD_OK;
    HIP_TRY(hipSetDevice(ctx->device), ctx, return SEPAIHRD_E_HIP);
    double* d = nullptr;
    HIP_TRY(hipMalloc((void**)&d, 2 * (size_t)n * sizeof(double)), ctx, return SEPAIHRD_E_HIP);
    const bool ok = hipMemcpy(d, x, (size_t)n * sizeof(double), hipMemcpyHostToDevice) == hipSuccess &&
                    poisson_log_values(d, n, d + n, nullptr) == 0 &&
                    hipMemcpy(out, d + n, (size_t)n * sizeof(double), hipMemcpyDeviceToHost) == hipSuccess;
    (void)hipFree(d);
    if (!ok) { ctx->last_error = "device_log_values: launch or copy failed"; return SEPAIHRD_E_HIP; }
    return SEPAIHRD_OK;
}

namespace {
// seed_streams / keep_scale_on_device refuse when the device functions are not this host's libm
int mh_require_matching_libm(sepaihrd_mh* mh, const char* who) {
    sepaihrd_ctx* ctx = mh->ctx;
    int32_t dl = 0, de = 0;
    const int rc = sepaihrd_device_libm_check(ctx, &dl, &de);
    if (rc != SEPAIHRD_OK) return rc;
    if (dl == 0 && de == 0) return SEPAIHRD_OK;
    char msg[384];
    std::snprintf(msg, sizeof(msg),
                  "%s: the device's log / exp (csrc/sepaihrd_rng.inc: glibc 2.35, x86-64, FMA variants) differ from this host's libm on %d "
                  "(log) and %d (exp) of %d self-check arguments: streams drawn on the device would not be the host's -- keep the draws "
                  "and the scale adaptation on the host (device_streams 0)", who, dl, de, sampler_libm_check_count());
    ctx->last_error = msg;
    return SEPAIHRD_E_UNSUPPORTED;
}
}  // namespace

int sepaihrd_mh_seed_streams(sepaihrd_mh* mh, uint32_t seed0) {
    if (!mh) return SEPAIHRD_E_INVALID_ARG;
    sepaihrd_ctx* ctx = mh->ctx;
    HIP_TRY(hipSetDevice(ctx->device), ctx, return SEPAIHRD_E_HIP);
    if (const int rc = mh_require_matching_libm(mh, "mh_seed_streams")) return rc;
    if (sampler_seed_streams(mh->st, seed0, mh->stream) != 0) { ctx->last_error = "mh_seed_streams: launch failed"; return SEPAIHRD_E_HIP; }
    HIP_TRY(hipStreamSynchronize(mh->stream), ctx, return SEPAIHRD_E_HIP);
    mh->device_rng = true;
    return SEPAIHRD_OK;
}

int sepaihrd_mh_keep_scale_on_device(sepaihrd_mh* mh, int adapt_scale, double target_rate, int keep_trace) {
    if (!mh) return SEPAIHRD_E_INVALID_ARG;
    sepaihrd_ctx* ctx = mh->ctx;
    HIP_TRY(hipSetDevice(ctx->device), ctx, return SEPAIHRD_E_HIP);
    if (mh->rows > 1) {
        ctx->last_error = "mh_keep_scale_on_device: call it before the first iteration (the accept window and the sample values start with the run)";
        return SEPAIHRD_E_INVALID_ARG;
    }
    if (adapt_scale)  // global_scale_ = exp(log_scale_) on the device
        if (const int rc = mh_require_matching_libm(mh, "mh_keep_scale_on_device")) return rc;
    SamplerState& st = mh->st;
    const size_t C = (size_t)st.C;
    {   // what this call adds to the sampler's state (sepaihrd_mh_create budgeted the rest): say so before failing inside
        const double need = (double)C * (1000.0 + 16.0 + 16.0) + (st.n_store > 0 && !st.lp_store ? (double)C * st.n_store * 8.0 : 0.0) +
                            (keep_trace && mh->iterations > 1 && !st.trace ? (double)C * (double)(mh->iterations - 1) : 0.0);
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && need > (double)free_b) {
            char msg[256];
            std::snprintf(msg, sizeof(msg), "mh_keep_scale_on_device: the accept window, the sample values and %s need %.3f GB but %.3f GB "
                          "of device memory are free", keep_trace ? "the accept trace (C * (iterations - 1) bytes)" : "no trace",
                          need / 1e9, (double)free_b / 1e9);
            ctx->last_error = msg;
            return SEPAIHRD_E_HIP;
        }
    }
    auto dalloc = [&](void** p, size_t bytes) -> bool {
        if (*p) return true;
        if (hipMalloc(p, bytes) != hipSuccess) return false;
        mh->allocs.push_back(*p);
        return true;
    };
    bool ok = dalloc((void**)&st.log_scale, C * sizeof(double)) && dalloc((void**)&st.scale, C * sizeof(double)) &&
              dalloc((void**)&st.recent, C * 1000) && dalloc((void**)&st.recent_meta, C * 4 * sizeof(int32_t));
    if (ok && st.n_store > 0) ok = dalloc((void**)&st.lp_store, C * st.n_store * sizeof(double));
    if (ok && keep_trace && mh->iterations > 1) ok = dalloc((void**)&st.trace, C * (size_t)(mh->iterations - 1));
    if (!ok) { ctx->last_error = "mh_keep_scale_on_device: device allocation failed"; return SEPAIHRD_E_HIP; }
    std::vector<double> ones(C, 1.0);
    HIP_TRY(hipMemset(st.log_scale, 0, C * sizeof(double)), ctx, return SEPAIHRD_E_HIP);            // log_scale_ = 0, global_scale_ = 1 (:252-253)
    HIP_TRY(hipMemcpy(st.scale, ones.data(), C * sizeof(double), hipMemcpyHostToDevice), ctx, return SEPAIHRD_E_HIP);
    HIP_TRY(hipMemset(st.recent, 0, C * 1000), ctx, return SEPAIHRD_E_HIP);
    HIP_TRY(hipMemset(st.recent_meta, 0, C * 4 * sizeof(int32_t)), ctx, return SEPAIHRD_E_HIP);
    if (st.trace) HIP_TRY(hipMemset(st.trace, 0, C * (size_t)(mh->iterations - 1)), ctx, return SEPAIHRD_E_HIP);
    if (st.lp_store && mh->values_set && mh->rows <= 1)  // called after sepaihrd_mh_set_values: sample 0's value is the chain's current one
        HIP_TRY(hipMemcpy2D(st.lp_store, (size_t)st.n_store * sizeof(double), mh->d_lp, sizeof(double), sizeof(double), C, hipMemcpyDeviceToDevice),
                ctx, return SEPAIHRD_E_HIP);
    st.adapt_scale = adapt_scale ? 1 : 0;
    st.target_rate = target_rate;
    st.device_scale = 1;
    return SEPAIHRD_OK;
}

int sepaihrd_mh_read_run_state(sepaihrd_mh* mh, double* values, double* best_values, double* scales, int32_t* accepted, int32_t* emergency) {
    if (!mh) return SEPAIHRD_E_INVALID_ARG;
    sepaihrd_ctx* ctx = mh->ctx;
    HIP_TRY(hipSetDevice(ctx->device), ctx, return SEPAIHRD_E_HIP);
    HIP_TRY(hipStreamSynchronize(mh->stream), ctx, return SEPAIHRD_E_HIP);
    const size_t C = (size_t)mh->st.C;
    if (values) HIP_TRY(hipMemcpy(values, mh->d_lp, C * sizeof(double), hipMemcpyDeviceToHost), ctx, return SEPAIHRD_E_HIP);
    if (best_values) HIP_TRY(hipMemcpy(best_values, mh->d_best_lp, C * sizeof(double), hipMemcpyDeviceToHost), ctx, return SEPAIHRD_E_HIP);
    if (scales) {
        if (!mh->st.scale) { ctx->last_error = "mh_read_run_state: the scale is not kept on the device"; return SEPAIHRD_E_INVALID_ARG; }
        HIP_TRY(hipMemcpy(scales, mh->st.scale, C * sizeof(double), hipMemcpyDeviceToHost), ctx, return SEPAIHRD_E_HIP);
    }
    if (accepted) HIP_TRY(hipMemcpy(accepted, mh->st.accepted, C * sizeof(int32_t), hipMemcpyDeviceToHost), ctx, return SEPAIHRD_E_HIP);
    if (emergency) {
        if (!mh->st.recent_meta) { ctx->last_error = "mh_read_run_state: the scale is not kept on the device"; return SEPAIHRD_E_INVALID_ARG; }
        HIP_TRY(hipMemcpy2D(emergency, sizeof(int32_t), mh->st.recent_meta + 3, 4 * sizeof(int32_t), sizeof(int32_t), C, hipMemcpyDeviceToHost), ctx,
                return SEPAIHRD_E_HIP);
    }
    return SEPAIHRD_OK;
}

// Progress reports and checkpoints of a run whose iterations are queued ahead (MetropolisHastingsSampler.cpp:363-383 reads the
// chain's value, best value, acceptance count, scale and -- for posterior_trace_checkpoint.csv -- its newest samples every
// report_interval iterations).  Reading them with the synchronous getters would drain the queue.
int sepaihrd_mh_snapshot_begin(sepaihrd_mh* mh, const int32_t* chains, int n, int first_sample, int count) {
    if (!mh || !chains || n <= 0 || first_sample < 0 || count < 0) return SEPAIHRD_E_INVALID_ARG;
    sepaihrd_ctx* ctx = mh->ctx;
    const int C = mh->st.C, P = mh->st.P;
    if (mh->snap_pending) { ctx->last_error = "mh_snapshot_begin: the previous snapshot has not been collected (sepaihrd_mh_snapshot_end)"; return SEPAIHRD_E_INVALID_ARG; }
    if (!mh->values_set) { ctx->last_error = "mh_snapshot_begin: the chains' values are unknown (sepaihrd_mh_set_values)"; return SEPAIHRD_E_INVALID_ARG; }
    if (count > 0 && first_sample + count > sepaihrd_mh_sample_count(mh)) { ctx->last_error = "mh_snapshot_begin: beyond the samples stored so far"; return SEPAIHRD_E_INVALID_ARG; }
    for (int k = 0; k < n; ++k)
        if (chains[k] < 0 || chains[k] >= C) { ctx->last_error = "mh_snapshot_begin: chain out of range"; return SEPAIHRD_E_INVALID_ARG; }
    HIP_TRY(hipSetDevice(ctx->device), ctx, return SEPAIHRD_E_HIP);
    if (!mh->snap_stream) {
        HIP_TRY(hipStreamCreateWithFlags(&mh->snap_stream, hipStreamNonBlocking), ctx, return SEPAIHRD_E_HIP);
        HIP_TRY(hipEventCreateWithFlags(&mh->ev_snap_gathered, hipEventDisableTiming), ctx, return SEPAIHRD_E_HIP);
        HIP_TRY(hipEventCreateWithFlags(&mh->ev_snap_done, hipEventDisableTiming), ctx, return SEPAIHRD_E_HIP);
    }
    const size_t width = 4 + (size_t)count * ((size_t)P + 1), need = (size_t)n * width;
    if (need > mh->snap_cap) {  // the previous snapshot has been collected: nothing reads the old buffers any more
        if (mh->d_snap) (void)hipFree(mh->d_snap);
        if (mh->h_snap) (void)hipHostFree(mh->h_snap);
        mh->d_snap = mh->h_snap = nullptr;
        mh->snap_cap = 0;
        HIP_TRY(hipMalloc((void**)&mh->d_snap, need * sizeof(double)), ctx, return SEPAIHRD_E_HIP);
        HIP_TRY(hipHostMalloc((void**)&mh->h_snap, need * sizeof(double), hipHostMallocDefault), ctx, return SEPAIHRD_E_HIP);
        mh->snap_cap = need;
    }
    if ((size_t)n > mh->snap_chain_cap) {
        if (mh->d_snap_chains) (void)hipFree(mh->d_snap_chains);
        mh->d_snap_chains = nullptr;
        mh->snap_chain_cap = 0;
        mh->snap_chain_list.clear();
        HIP_TRY(hipMalloc((void**)&mh->d_snap_chains, (size_t)n * sizeof(int32_t)), ctx, return SEPAIHRD_E_HIP);
        mh->snap_chain_cap = (size_t)n;
    }
    // the chain list is uploaded when it changes (every report of a run names the same chains): a blocking copy once, none after
    if (mh->snap_chain_list.size() != (size_t)n || !std::equal(chains, chains + n, mh->snap_chain_list.begin())) {
        HIP_TRY(hipMemcpy(mh->d_snap_chains, chains, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice), ctx, return SEPAIHRD_E_HIP);
        mh->snap_chain_list.assign(chains, chains + n);
    }
    if (sampler_snapshot(mh->st, mh->d_lp, mh->d_best_lp, mh->d_snap_chains, n, first_sample, count, mh->d_snap, mh->stream) != 0) {
        ctx->last_error = "mh_snapshot_begin: launch failed";
        return SEPAIHRD_E_HIP;
    }
    HIP_TRY(hipEventRecord(mh->ev_snap_gathered, mh->stream), ctx, return SEPAIHRD_E_HIP);
    HIP_TRY(hipStreamWaitEvent(mh->snap_stream, mh->ev_snap_gathered, 0), ctx, return SEPAIHRD_E_HIP);
    HIP_TRY(hipMemcpyAsync(mh->h_snap, mh->d_snap, need * sizeof(double), hipMemcpyDeviceToHost, mh->snap_stream), ctx, return SEPAIHRD_E_HIP);
    HIP_TRY(hipEventRecord(mh->ev_snap_done, mh->snap_stream), ctx, return SEPAIHRD_E_HIP);
    mh->snap_n = n;
    mh->snap_count = count;
    mh->snap_pending = true;
    return SEPAIHRD_OK;
}

// returns 1 while the snapshot has not landed (wait == 0 only), 0 with the outputs filled, < 0 on error.  May be called from
// another host thread than the one that queues the iterations (it touches the snapshot's own event and buffer only).
int sepaihrd_mh_snapshot_end(sepaihrd_mh* mh, int wait, double* state, double* samples, double* sample_values) {
    if (!mh) return SEPAIHRD_E_INVALID_ARG;
    if (!mh->snap_pending) return SEPAIHRD_E_INVALID_ARG;
    if (hipSetDevice(mh->ctx->device) != hipSuccess) return SEPAIHRD_E_HIP;
    if (!wait) {
        const hipError_t q = hipEventQuery(mh->ev_snap_done);
        if (q == hipErrorNotReady) return 1;
        if (q != hipSuccess) return SEPAIHRD_E_HIP;
    } else if (hipEventSynchronize(mh->ev_snap_done) != hipSuccess) return SEPAIHRD_E_HIP;
    const size_t P = (size_t)mh->st.P, count = (size_t)mh->snap_count, width = 4 + count * (P + 1);
    for (int k = 0; k < mh->snap_n; ++k) {
        const double* o = mh->h_snap + (size_t)k * width;
        if (state) std::memcpy(state + 4 * (size_t)k, o, 4 * sizeof(double));
        if (samples && count) std::memcpy(samples + (size_t)k * count * P, o + 4, count * P * sizeof(double));
        if (sample_values && count) std::memcpy(sample_values + (size_t)k * count, o + 4 + count * P, count * sizeof(double));
    }
    mh->snap_pending = false;
    return SEPAIHRD_OK;
}

int sepaihrd_mh_read_failure_counts(sepaihrd_mh* mh, int64_t counts[3]) {
    if (!mh || !counts) return SEPAIHRD_E_INVALID_ARG;
    sepaihrd_ctx* ctx = mh->ctx;
    HIP_TRY(hipSetDevice(ctx->device), ctx, return SEPAIHRD_E_HIP);
    HIP_TRY(hipStreamSynchronize(mh->stream), ctx, return SEPAIHRD_E_HIP);
    uint32_t h[3] = {0, 0, 0};
    HIP_TRY(hipMemcpy(h, mh->st.fail_counts, sizeof(h), hipMemcpyDeviceToHost), ctx, return SEPAIHRD_E_HIP);
    for (int k = 0; k < 3; ++k) counts[k] = (int64_t)h[k];
    return SEPAIHRD_OK;
}

int sepaihrd_mh_read_sample_values(sepaihrd_mh* mh, int first, int count, double* out) {
    if (!mh || !out || first < 0 || count <= 0) return SEPAIHRD_E_INVALID_ARG;
    sepaihrd_ctx* ctx = mh->ctx;
    if (!mh->st.lp_store || first + count > sepaihrd_mh_sample_count(mh)) {
        ctx->last_error = "mh_read_sample_values: not kept (sepaihrd_mh_keep_scale_on_device) or beyond the samples stored so far";
        return SEPAIHRD_E_INVALID_ARG;
    }
    HIP_TRY(hipSetDevice(ctx->device), ctx, return SEPAIHRD_E_HIP);
    HIP_TRY(hipStreamSynchronize(mh->stream), ctx, return SEPAIHRD_E_HIP);
    HIP_TRY(hipMemcpy2D(out, (size_t)count * sizeof(double), mh->st.lp_store + first, (size_t)mh->st.n_store * sizeof(double),
                        (size_t)count * sizeof(double), (size_t)mh->st.C, hipMemcpyDeviceToHost), ctx, return SEPAIHRD_E_HIP);
    return SEPAIHRD_OK;
}

int sepaihrd_mh_read_accept_trace(sepaihrd_mh* mh, uint8_t* out) {
    if (!mh || !out) return SEPAIHRD_E_INVALID_ARG;
    sepaihrd_ctx* ctx = mh->ctx;
    if (!mh->st.trace) { ctx->last_error = "mh_read_accept_trace: no trace kept"; return SEPAIHRD_E_INVALID_ARG; }
    HIP_TRY(hipSetDevice(ctx->device), ctx, return SEPAIHRD_E_HIP);
    HIP_TRY(hipStreamSynchronize(mh->stream), ctx, return SEPAIHRD_E_HIP);
    HIP_TRY(hipMemcpy(out, mh->st.trace, (size_t)mh->st.C * (size_t)(mh->iterations - 1), hipMemcpyDeviceToHost), ctx, return SEPAIHRD_E_HIP);
    return SEPAIHRD_OK;
}

int sepaihrd_mh_draw_first(sepaihrd_mh* mh) {
    if (!mh) return SEPAIHRD_E_INVALID_ARG;
    sepaihrd_ctx* ctx = mh->ctx;
    if (!mh->device_rng) { ctx->last_error = "mh_draw_first: call sepaihrd_mh_seed_streams first"; return SEPAIHRD_E_INVALID_ARG; }
    HIP_TRY(hipSetDevice(ctx->device), ctx, return SEPAIHRD_E_HIP);
    // proposal 1: the normals from the start of every stream, into the staged-normals buffer
    if (sampler_draw(mh->st, nullptr, 1, nullptr, mh->d_z_stage, nullptr, 1, mh->copy_stream) != 0) { ctx->last_error = "mh_draw_first: launch failed"; return SEPAIHRD_E_HIP; }
    HIP_TRY(hipEventRecord(mh->ev_staged, mh->copy_stream), ctx, return SEPAIHRD_E_HIP);
    mh->staged = true;
    return SEPAIHRD_OK;
}

int sepaihrd_mh_step_tested(sepaihrd_mh* mh, double gamma, int adapt, int last) {
    if (!mh || adapt < 0 || adapt > 3) return SEPAIHRD_E_INVALID_ARG;
    sepaihrd_ctx* ctx = mh->ctx;
    const int C = mh->st.C, P = mh->st.P;
    if (!mh->values_set) { ctx->last_error = "mh_step_tested: call sepaihrd_mh_set_values first"; return SEPAIHRD_E_INVALID_ARG; }
    const bool self_contained = mh->device_rng && mh->st.device_scale != 0;  // nothing of the caller's goes into the test
    if (self_contained && (mh->auto_steps++ % 32) == 0) {
        const int slot = (int)((mh->auto_steps / 32) % 4);
        HIP_TRY(hipSetDevice(ctx->device), ctx, return SEPAIHRD_E_HIP);
        if (!mh->ev_ahead[slot]) HIP_TRY(hipEventCreateWithFlags(&mh->ev_ahead[slot], hipEventDisableTiming), ctx, return SEPAIHRD_E_HIP);
        if (mh->ev_ahead_used[slot]) HIP_TRY(hipEventSynchronize(mh->ev_ahead[slot]), ctx, return SEPAIHRD_E_HIP);  // ~128 steps ago
        HIP_TRY(hipEventRecord(mh->ev_ahead[slot], mh->stream), ctx, return SEPAIHRD_E_HIP);
        mh->ev_ahead_used[slot] = true;
    }
    if (mh->test_pending && !self_contained) { ctx->last_error = "mh_step_tested: the previous test has not been fetched"; return SEPAIHRD_E_INVALID_ARG; }
    if (!last && !mh->staged && !mh->device_rng) { ctx->last_error = "mh_step_tested: no staged normals (call sepaihrd_mh_stage_normals first)"; return SEPAIHRD_E_INVALID_ARG; }
    if (mh->rows >= mh->iterations) { ctx->last_error = "mh_step_tested: more states than the sampler was created for"; return SEPAIHRD_E_INVALID_ARG; }
    HIP_TRY(hipSetDevice(ctx->device), ctx, return SEPAIHRD_E_HIP);
    hipStream_t st = mh->stream, cs = mh->copy_stream;
    const size_t CP = (size_t)C * P;
    mh->lz_ready = false;
    // the test's inputs travel on the copy stream while the evaluation still runs; they may not overwrite what the
    // previous proposal is reading
    if (mh->proposed_once) HIP_TRY(hipStreamWaitEvent(cs, mh->ev_last_proposed, 0), ctx, return SEPAIHRD_E_HIP);
    if (mh->device_rng) {
        // the caller supplies the two scale candidates only; log(u) and the normals of both continuations are drawn here,
        // from where the previous test left every chain's stream (its flags are final: the copy stream has just waited
        // for the kernel that wrote them)
        if (!self_contained)
            HIP_TRY(hipMemcpyAsync(mh->d_test + C, mh->h_test + C, 2 * (size_t)C * sizeof(double), hipMemcpyHostToDevice, cs), ctx,
                    return SEPAIHRD_E_HIP);
        uint8_t* const prev_flags = reinterpret_cast<uint8_t*>(static_cast<double*>(mh->d_test_out) + C);
        // The draws of test i + 1 need nothing of evaluation i, so they run beside it on the copy stream -- as long as the
        // evaluation leaves room.  A saturating batch (two 256-register waves on every SIMD) leaves none: the draw kernel's
        // 65 536 small waves then take the slots of retiring integrator waves and hold up the next ones (round 4, 65 536
        // chains: the evaluation stretched from 2.9 to 3.6 ms around 0.1 ms of draws).  From that size on the draws queue
        // behind the evaluation on the main stream instead (65 536 chains: 4.01 -> 3.43 ms per iteration; a batch of exactly
        // one round, 32 768 chains, still gains from the overlap -- 1.95 against 2.00 -- as the draws fill the round's tail;
        // a low-priority copy stream changed nothing).  SEPAIHRD_MH_DRAW=overlap|serial overrides (A/B runs).
        const bool serial_draw = mh_draws_behind_the_evaluation(ctx, C);
        if (sampler_draw(mh->st, prev_flags, 0, mh->d_test, mh->d_z_stage, mh->d_test + 3 * (size_t)C, last ? 0 : 1, serial_draw ? st : cs) != 0) {
            ctx->last_error = "mh_step_tested: draw launch failed";
            return SEPAIHRD_E_HIP;
        }
        mh->staged = true;
        // ... and, when the draws run beside the evaluation and no covariance refresh separates this test from its proposal,
        // L z of both continuations too: the launch between two evaluations then reads no factor (SEPAIHRD_MH_LZ=fused: A/B)
        if (!serial_draw && !last && adapt <= 1 && mh->st.P <= 200 && mh_lz_ahead_wanted()) {
            if (sampler_lz(mh->st, mh->d_z_stage, mh->d_test + 3 * (size_t)C, mh->d_lz, mh->d_lz + CP, cs) != 0) {
                ctx->last_error = "mh_step_tested: L z launch failed";
                return SEPAIHRD_E_HIP;
            }
            mh->lz_ready = true;
        }
    } else
    HIP_TRY(hipMemcpyAsync(mh->d_test, mh->h_test, (3 * (size_t)C + (last ? 0 : CP)) * sizeof(double), hipMemcpyHostToDevice, cs), ctx,
            return SEPAIHRD_E_HIP);
    HIP_TRY(hipEventRecord(mh->ev_test_up, cs), ctx, return SEPAIHRD_E_HIP);
    HIP_TRY(hipStreamWaitEvent(st, mh->ev_test_up, 0), ctx, return SEPAIHRD_E_HIP);
    double* const d_values = static_cast<double*>(mh->d_test_out);
    uint8_t* const d_flags = reinterpret_cast<uint8_t*>(d_values + C);
    if (mh_before_commit(mh) != 0) { ctx->last_error = "mh_step_tested: catch-up of the queued updates failed"; return SEPAIHRD_E_HIP; }
    if (!last && adapt <= 1) {
        // no covariance refresh between commit and proposal: test, commit and proposal in one launch
        // (the staged normals landed before the test's inputs: same copy stream, staged first -- ev_test_up covers them)
        if (sampler_test_commit_propose(mh->st, ctx->dp, mh->d_loglik, mh->d_status, mh->d_test, mh->d_test + C, mh->d_test + 2 * (size_t)C,
                                        mh->d_lp, mh->d_best_lp, mh->d_scale_sel, d_flags, d_values, mh->d_z_stage,
                                        mh->d_test + 3 * (size_t)C, mh->rows, st, mh->lz_ready ? mh->d_lz : nullptr,
                                        mh->lz_ready ? mh->d_lz + CP : nullptr) != 0) {
            ctx->last_error = "mh_step_tested: launch failed";
            return SEPAIHRD_E_HIP;
        }
        HIP_TRY(hipEventRecord(mh->ev_tested, st), ctx, return SEPAIHRD_E_HIP);  // one marker: tested AND proposed
        mh->ev_last_proposed = mh->ev_tested;
        HIP_TRY(hipStreamWaitEvent(cs, mh->ev_tested, 0), ctx, return SEPAIHRD_E_HIP);
        if (!self_contained) {  // the outcome for the caller's bookkeeping (a self-contained sampler keeps its own: read at the end)
            HIP_TRY(hipMemcpyAsync(mh->h_test_out, mh->d_test_out, (size_t)C * (sizeof(double) + 1), hipMemcpyDeviceToHost, cs), ctx,
                    return SEPAIHRD_E_HIP);
            HIP_TRY(hipEventRecord(mh->ev_fetched, cs), ctx, return SEPAIHRD_E_HIP);
            mh->test_pending = true;
        }
        mh->proposed_once = true;
        mh->rows++;
        if (adapt == 1) mh_queue_rank1(mh, gamma);  // as mh_adapt_step: the queued update names history row rows - 1
        std::swap(mh->d_z, mh->d_z_stage);
        mh->staged = false;
        return sepaihrd_eval_batch_device(ctx, mh->st.prop, C, mh->d_loglik, mh->d_status, nullptr, nullptr, nullptr, nullptr, st);
    }
    int rc = sampler_accept_test(mh->st, mh->rows, mh->d_loglik, mh->d_status, mh->d_test, mh->d_test + C, mh->d_test + 2 * (size_t)C, mh->d_lp,
                                 mh->d_best_lp, mh->d_scale_sel, d_flags, d_values, st);
    if (rc != 0) { ctx->last_error = "mh_step_tested: launch failed"; return SEPAIHRD_E_HIP; }
    HIP_TRY(hipEventRecord(mh->ev_tested, st), ctx, return SEPAIHRD_E_HIP);
    // the outcome goes back on the copy stream as soon as the test has run; the main stream does not wait for it
    HIP_TRY(hipStreamWaitEvent(cs, mh->ev_tested, 0), ctx, return SEPAIHRD_E_HIP);
    if (!self_contained) {
        HIP_TRY(hipMemcpyAsync(mh->h_test_out, mh->d_test_out, (size_t)C * (sizeof(double) + 1), hipMemcpyDeviceToHost, cs), ctx,
                return SEPAIHRD_E_HIP);
        HIP_TRY(hipEventRecord(mh->ev_fetched, cs), ctx, return SEPAIHRD_E_HIP);
        mh->test_pending = true;
    }
    rc = sampler_commit_counted(mh->st, d_flags, mh->rows, 1, st);
    if (rc != 0) { ctx->last_error = "mh_step_tested: launch failed"; return SEPAIHRD_E_HIP; }
    mh->rows++;
    if (last) return SEPAIHRD_OK;
    rc = mh_adapt_step(mh, gamma, adapt);
    if (rc != 0) { ctx->last_error = "mh_step_tested: launch failed"; return SEPAIHRD_E_HIP; }
    HIP_TRY(hipStreamWaitEvent(st, mh->ev_staged, 0), ctx, return SEPAIHRD_E_HIP);  // the staged normals have landed
    rc = sampler_propose_select(mh->st, ctx->dp, mh->d_z_stage, mh->d_test + 3 * (size_t)C, d_flags, mh->d_scale_sel, st);
    if (rc != 0) { ctx->last_error = "mh_step_tested: launch failed"; return SEPAIHRD_E_HIP; }
    HIP_TRY(hipEventRecord(mh->ev_proposed, st), ctx, return SEPAIHRD_E_HIP);
    mh->ev_last_proposed = mh->ev_proposed;
    mh->proposed_once = true;
    std::swap(mh->d_z, mh->d_z_stage);
    mh->staged = false;
    return sepaihrd_eval_batch_device(ctx, mh->st.prop, C, mh->d_loglik, mh->d_status, nullptr, nullptr, nullptr, nullptr, st);
}

int sepaihrd_mh_fetch_test(sepaihrd_mh* mh, double* values, uint8_t* flags) {
    if (!mh || !values || !flags) return SEPAIHRD_E_INVALID_ARG;
    sepaihrd_ctx* ctx = mh->ctx;
    if (!mh->test_pending) { ctx->last_error = "mh_fetch_test: no test pending"; return SEPAIHRD_E_INVALID_ARG; }
    HIP_TRY(hipSetDevice(ctx->device), ctx, return SEPAIHRD_E_HIP);
    HIP_TRY(hipEventSynchronize(mh->ev_fetched), ctx, return SEPAIHRD_E_HIP);
    const size_t C = (size_t)mh->st.C;
    std::memcpy(values, mh->h_test_out, C * sizeof(double));
    std::memcpy(flags, static_cast<const char*>(mh->h_test_out) + C * sizeof(double), C);
    mh->test_pending = false;
    return SEPAIHRD_OK;
}

int sepaihrd_mh_commit(sepaihrd_mh* mh, const uint8_t* accept) {
    if (!mh || !accept) return SEPAIHRD_E_INVALID_ARG;
    sepaihrd_ctx* ctx = mh->ctx;
    if (mh->rows >= mh->iterations) {
        ctx->last_error = "mh_commit: more states than the sampler was created for";
        return SEPAIHRD_E_INVALID_ARG;
    }
    HIP_TRY(hipSetDevice(ctx->device), ctx, return SEPAIHRD_E_HIP);
    if (mh_before_commit(mh) != 0) { ctx->last_error = "mh_commit: catch-up of the queued updates failed"; return SEPAIHRD_E_HIP; }
    HIP_TRY(hipMemcpyAsync(mh->d_accept, accept, (size_t)mh->st.C, hipMemcpyHostToDevice, mh->stream), ctx, return SEPAIHRD_E_HIP);
    if (sampler_commit(mh->st, mh->d_accept, mh->rows, mh->stream) != 0) {
        ctx->last_error = "mh_commit: launch failed";
        return SEPAIHRD_E_HIP;
    }
    mh->rows++;
    return SEPAIHRD_OK;
}

int sepaihrd_mh_adapt(sepaihrd_mh* mh, double gamma, int refresh, int recompute_full) {
    if (!mh) return SEPAIHRD_E_INVALID_ARG;
    sepaihrd_ctx* ctx = mh->ctx;
    HIP_TRY(hipSetDevice(ctx->device), ctx, return SEPAIHRD_E_HIP);
    const int rc = mh_adapt_step(mh, gamma, refresh ? (recompute_full ? 3 : 2) : 1);
    if (rc != 0) {
        ctx->last_error = "mh_adapt: launch failed";
        return SEPAIHRD_E_HIP;
    }
    return SEPAIHRD_OK;
}

int sepaihrd_mh_read_history(sepaihrd_mh* mh, const int32_t* rows, int n_rows, double* out) {
    if (!mh || !rows || n_rows <= 0 || !out) return SEPAIHRD_E_INVALID_ARG;
    sepaihrd_ctx* ctx = mh->ctx;
    HIP_TRY(hipSetDevice(ctx->device), ctx, return SEPAIHRD_E_HIP);
    const int C = mh->st.C, P = mh->st.P, W = mh->st.window;
    for (int r = 0; r < n_rows; ++r)
        if (rows[r] < 0 || rows[r] >= mh->rows || rows[r] < mh->rows - W) {
            ctx->last_error = "mh_read_history: state not committed yet, or no longer among the newest `window` states "
                              "(the thinned samples are read with sepaihrd_mh_read_samples)";
            return SEPAIHRD_E_INVALID_ARG;
        }
    HIP_TRY(hipStreamSynchronize(mh->stream), ctx, return SEPAIHRD_E_HIP);
    // strided copies straight out of the ring: state r of every chain = a 2-D region
    for (int r = 0; r < n_rows; ++r)
        HIP_TRY(hipMemcpy2D(out + (size_t)r * P, (size_t)n_rows * P * sizeof(double),
                            mh->st.hist + (size_t)(rows[r] % W) * P, (size_t)W * P * sizeof(double),
                            (size_t)P * sizeof(double), (size_t)C, hipMemcpyDeviceToHost),
                ctx, return SEPAIHRD_E_HIP);
    return SEPAIHRD_OK;
}

int sepaihrd_mh_sample_count(const sepaihrd_mh* mh) {
    if (!mh || mh->st.n_store <= 0 || mh->rows <= 0) return 0;
    return std::min(mh->st.n_store, (mh->rows - 1) / mh->st.thinning + 1);
}

int sepaihrd_mh_read_samples(sepaihrd_mh* mh, int first, int count, double* out) {
    if (!mh || !out || first < 0 || count <= 0) return SEPAIHRD_E_INVALID_ARG;
    sepaihrd_ctx* ctx = mh->ctx;
    if (first + count > sepaihrd_mh_sample_count(mh)) {
        ctx->last_error = "mh_read_samples: beyond the samples stored so far";
        return SEPAIHRD_E_INVALID_ARG;
    }
    HIP_TRY(hipSetDevice(ctx->device), ctx, return SEPAIHRD_E_HIP);
    HIP_TRY(hipStreamSynchronize(mh->stream), ctx, return SEPAIHRD_E_HIP);
    const size_t P = (size_t)mh->st.P;
    HIP_TRY(hipMemcpy2D(out, (size_t)count * P * sizeof(double), mh->st.store + (size_t)first * P,
                        (size_t)mh->st.n_store * P * sizeof(double), (size_t)count * P * sizeof(double), (size_t)mh->st.C,
                        hipMemcpyDeviceToHost),
            ctx, return SEPAIHRD_E_HIP);
    return SEPAIHRD_OK;
}

int sepaihrd_mh_summary_records(sepaihrd_mh* mh, int first_sample, double* out, double* d_out) {
    if (!mh || (!out && !d_out)) return SEPAIHRD_E_INVALID_ARG;
    sepaihrd_ctx* ctx = mh->ctx;
    const int ns = sepaihrd_mh_sample_count(mh);
    if (first_sample < 0 || first_sample >= ns) {
        ctx->last_error = "mh_summary_records: first_sample beyond the samples stored so far";
        return SEPAIHRD_E_INVALID_ARG;
    }
    if (!mh->values_set) { ctx->last_error = "mh_summary_records: the chains' values are unknown (sepaihrd_mh_set_values)"; return SEPAIHRD_E_INVALID_ARG; }
    HIP_TRY(hipSetDevice(ctx->device), ctx, return SEPAIHRD_E_HIP);
    double* const dst = d_out ? d_out : mh->d_summary;
    if (sampler_summary_records(mh->st, mh->d_best_lp, first_sample, ns, dst, mh->stream) != 0) {
        ctx->last_error = "mh_summary_records: launch failed";
        return SEPAIHRD_E_HIP;
    }
    HIP_TRY(hipStreamSynchronize(mh->stream), ctx, return SEPAIHRD_E_HIP);
    if (out)
        HIP_TRY(hipMemcpy(out, dst, (size_t)mh->st.C * (2 * (size_t)mh->st.P + 2) * sizeof(double), hipMemcpyDeviceToHost), ctx,
                return SEPAIHRD_E_HIP);
    return SEPAIHRD_OK;
}

int sepaihrd_mh_read_moments(sepaihrd_mh* mh, double* mean, double* m2) {
    if (!mh || (!mean && !m2)) return SEPAIHRD_E_INVALID_ARG;
    sepaihrd_ctx* ctx = mh->ctx;
    if (mh->covariance_mode != SEPAIHRD_MH_COV_RUNNING) { ctx->last_error = "mh_read_moments: the sampler keeps no running sums (two-pass mode)"; return SEPAIHRD_E_INVALID_ARG; }
    HIP_TRY(hipSetDevice(ctx->device), ctx, return SEPAIHRD_E_HIP);
    if (mh_flush_moments(mh, 0) != 0) { ctx->last_error = "mh_read_moments: catch-up failed"; return SEPAIHRD_E_HIP; }
    HIP_TRY(hipStreamSynchronize(mh->stream), ctx, return SEPAIHRD_E_HIP);
    const size_t CP = (size_t)mh->st.C * mh->st.P;
    if (mean) HIP_TRY(hipMemcpy(mean, mh->st.wmean, CP * sizeof(double), hipMemcpyDeviceToHost), ctx, return SEPAIHRD_E_HIP);
    if (m2) HIP_TRY(hipMemcpy(m2, mh->st.m2, CP * mh->st.P * sizeof(double), hipMemcpyDeviceToHost), ctx, return SEPAIHRD_E_HIP);
    return SEPAIHRD_OK;
}

int sepaihrd_mh_read_proposal(sepaihrd_mh* mh, double* prop) {
    if (!mh || !prop) return SEPAIHRD_E_INVALID_ARG;
    sepaihrd_ctx* ctx = mh->ctx;
    HIP_TRY(hipSetDevice(ctx->device), ctx, return SEPAIHRD_E_HIP);
    HIP_TRY(hipStreamSynchronize(mh->stream), ctx, return SEPAIHRD_E_HIP);
    HIP_TRY(hipMemcpy(prop, mh->st.prop, (size_t)mh->st.C * mh->st.P * sizeof(double), hipMemcpyDeviceToHost), ctx,
            return SEPAIHRD_E_HIP);
    return SEPAIHRD_OK;
}

int sepaihrd_mh_read_best(sepaihrd_mh* mh, double* best) {
    if (!mh || !best) return SEPAIHRD_E_INVALID_ARG;
    sepaihrd_ctx* ctx = mh->ctx;
    HIP_TRY(hipSetDevice(ctx->device), ctx, return SEPAIHRD_E_HIP);
    HIP_TRY(hipStreamSynchronize(mh->stream), ctx, return SEPAIHRD_E_HIP);
    HIP_TRY(hipMemcpy(best, mh->st.best, (size_t)mh->st.C * mh->st.P * sizeof(double), hipMemcpyDeviceToHost), ctx,
            return SEPAIHRD_E_HIP);
    return SEPAIHRD_OK;
}

int sepaihrd_mh_busy(sepaihrd_mh* mh) {
    if (!mh) return 0;
    return hipStreamQuery(mh->stream) == hipErrorNotReady ? 1 : 0;
}

int sepaihrd_mh_read_covariance(sepaihrd_mh* mh, double* cov) {
    if (!mh || !cov) return SEPAIHRD_E_INVALID_ARG;
    sepaihrd_ctx* ctx = mh->ctx;
    HIP_TRY(hipSetDevice(ctx->device), ctx, return SEPAIHRD_E_HIP);
    if (mh_flush_rank1(mh) != 0) { ctx->last_error = "mh_read_covariance: rank-one catch-up failed"; return SEPAIHRD_E_HIP; }
    HIP_TRY(hipStreamSynchronize(mh->stream), ctx, return SEPAIHRD_E_HIP);
    HIP_TRY(hipMemcpy(cov, mh->st.cov, (size_t)mh->st.C * mh->st.P * mh->st.P * sizeof(double), hipMemcpyDeviceToHost), ctx,
            return SEPAIHRD_E_HIP);
    return SEPAIHRD_OK;
}

// ------------------------------------------------------------------ summary records across devices
// The one exchange of the path: after sampling, every device gets the fixed-width per-chain records of ALL chains so that
// it can form the ensemble quantiles the reference computes serially (ResultAggregator.cpp:35-172).  One process drives
// several devices here (one context / host thread per device, as optimizeChainGroupsOnDevice runs them), so the
// collective is RCCL's single-process form: ncclCommInitAll over the contexts' devices and one ncclAllGather per device
// inside a group call -- xGMI is point to point, each device's block goes to its peers directly.  RCCL is loaded at first
// use (dlopen of librccl.so.1: the library has no link-time dependency on it, and a process that already carries a copy
// -- PyTorch's -- shares it); without it, or when two contexts share a device (RCCL wants one rank per device), the
// records are staged through the host.
double* sepaihrd_records_buffer(sepaihrd_ctx* ctx, int which, size_t doubles) {
    if (!ctx || which < 0 || which > 1) return nullptr;
    if (doubles > ctx->rec_cap[which]) {
        if (hipSetDevice(ctx->device) != hipSuccess) return nullptr;
        if (ctx->rec_buf[which]) (void)hipFree(ctx->rec_buf[which]);
        ctx->rec_buf[which] = nullptr;
        ctx->rec_cap[which] = 0;
        if (hipMalloc((void**)&ctx->rec_buf[which], doubles * sizeof(double)) != hipSuccess) {
            ctx->last_error = "records_buffer: device allocation failed";
            return nullptr;
        }
        ctx->rec_cap[which] = doubles;
    }
    return ctx->rec_buf[which];
}

int sepaihrd_read_records(sepaihrd_ctx* ctx, int which, double* out, size_t doubles) {
    if (!ctx || which < 0 || which > 1 || !out || doubles > ctx->rec_cap[which]) return SEPAIHRD_E_INVALID_ARG;
    HIP_TRY(hipSetDevice(ctx->device), ctx, return SEPAIHRD_E_HIP);
    HIP_TRY(hipMemcpy(out, ctx->rec_buf[which], doubles * sizeof(double), hipMemcpyDeviceToHost), ctx, return SEPAIHRD_E_HIP);
    return SEPAIHRD_OK;
}

int sepaihrd_write_records(sepaihrd_ctx* ctx, int which, const double* in, size_t doubles) {
    if (!ctx || which < 0 || which > 1 || !in) return SEPAIHRD_E_INVALID_ARG;
    if (!sepaihrd_records_buffer(ctx, which, doubles)) return SEPAIHRD_E_HIP;
    HIP_TRY(hipSetDevice(ctx->device), ctx, return SEPAIHRD_E_HIP);
    HIP_TRY(hipMemcpy(ctx->rec_buf[which], in, doubles * sizeof(double), hipMemcpyHostToDevice), ctx, return SEPAIHRD_E_HIP);
    return SEPAIHRD_OK;
}

extern "C++" {
namespace {
// the six RCCL entry points the gather needs, resolved once
struct RcclApi {
    void* lib = nullptr;
    int (*CommInitAll)(void**, int, const int*) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok = false;
};
const RcclApi& rccl_api() {
    static const RcclApi api = [] {
        RcclApi a;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            a.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (a.lib) break;
        }
        if (!a.lib) return a;
        a.CommInitAll = reinterpret_cast<decltype(a.CommInitAll)>(dlsym(a.lib, "ncclCommInitAll"));
        a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(a.lib, "ncclCommDestroy"));
        a.GroupStart = reinterpret_cast<decltype(a.GroupStart)>(dlsym(a.lib, "ncclGroupStart"));
        a.GroupEnd = reinterpret_cast<decltype(a.GroupEnd)>(dlsym(a.lib, "ncclGroupEnd"));
        a.AllGather = reinterpret_cast<decltype(a.AllGather)>(dlsym(a.lib, "ncclAllGather"));
        a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(a.lib, "ncclGetErrorString"));
        a.ok = a.CommInitAll && a.CommDestroy && a.GroupStart && a.GroupEnd && a.AllGather;
        return a;
    }();
    return api;
}
constexpr int NCCL_DOUBLE = 8;  // ncclFloat64 (rccl.h: ncclDataType_t)
}  // namespace
}  // extern "C++"

int sepaihrd_allgather_records(sepaihrd_ctx* const* ctxs, int n, const int32_t* rows, int width, int backend, int* backend_used) {
    if (!ctxs || n <= 0 || !rows || width <= 0 || backend < SEPAIHRD_GATHER_AUTO || backend > SEPAIHRD_GATHER_HOST) return SEPAIHRD_E_INVALID_ARG;
    for (int k = 0; k < n; ++k)
        if (!ctxs[k] || rows[k] < 0) return SEPAIHRD_E_INVALID_ARG;
    sepaihrd_ctx* c0 = ctxs[0];
    size_t total = 0;
    int max_rows = 0;
    for (int k = 0; k < n; ++k) { total += (size_t)rows[k]; max_rows = std::max(max_rows, (int)rows[k]); }
    for (int k = 0; k < n; ++k)
        if ((size_t)rows[k] * width > ctxs[k]->rec_cap[0]) { c0->last_error = "allgather_records: a context's local table (records_buffer 0) is smaller than rows * width"; return SEPAIHRD_E_INVALID_ARG; }
    bool distinct = true;
    for (int a = 0; a < n; ++a)
        for (int b = a + 1; b < n; ++b) distinct = distinct && ctxs[a]->device != ctxs[b]->device;
    const bool can_rccl = distinct && rccl_api().ok;
    if (backend == SEPAIHRD_GATHER_RCCL && !can_rccl) {
        c0->last_error = !distinct ? "allgather_records: RCCL wants one rank per device, two contexts share one" : "allgather_records: librccl could not be loaded";
        return SEPAIHRD_E_UNSUPPORTED;
    }
    const bool use_rccl = backend == SEPAIHRD_GATHER_RCCL || (backend == SEPAIHRD_GATHER_AUTO && can_rccl);
    if (backend_used) *backend_used = use_rccl ? SEPAIHRD_GATHER_RCCL : SEPAIHRD_GATHER_HOST;
    for (int k = 0; k < n; ++k)
        if (!sepaihrd_records_buffer(ctxs[k], 1, total * width)) { c0->last_error = "allgather_records: no room for the gathered table"; return SEPAIHRD_E_HIP; }

    if (!use_rccl) {  // host staging: every local table down, the whole table up
        std::vector<double> table(total * width);
        size_t off = 0;
        for (int k = 0; k < n; ++k) {
            HIP_TRY(hipSetDevice(ctxs[k]->device), c0, return SEPAIHRD_E_HIP);
            HIP_TRY(hipMemcpy(table.data() + off, ctxs[k]->rec_buf[0], (size_t)rows[k] * width * sizeof(double), hipMemcpyDeviceToHost), c0, return SEPAIHRD_E_HIP);
            off += (size_t)rows[k] * width;
        }
        for (int k = 0; k < n; ++k) {
            HIP_TRY(hipSetDevice(ctxs[k]->device), c0, return SEPAIHRD_E_HIP);
            HIP_TRY(hipMemcpy(ctxs[k]->rec_buf[1], table.data(), table.size() * sizeof(double), hipMemcpyHostToDevice), c0, return SEPAIHRD_E_HIP);
        }
        return SEPAIHRD_OK;
    }

    // RCCL: equal counts per rank, so every rank sends max_rows rows (its table padded) and the blocks are compacted after
    const RcclApi& rc = rccl_api();
    const size_t block = (size_t)max_rows * width;
    std::vector<int> devs(n);
    for (int k = 0; k < n; ++k) devs[k] = ctxs[k]->device;
    std::vector<void*> comms(n, nullptr);
    std::vector<double*> send(n, nullptr), recv(n, nullptr);
    std::vector<hipStream_t> streams(n, nullptr);
    int rc_code = 0;
    auto fail = [&](const char* what) {
        c0->last_error = std::string("allgather_records: ") + what + (rc_code && rc.GetErrorString ? std::string(": ") + rc.GetErrorString(rc_code) : std::string());
        for (int k = 0; k < n; ++k) {
            (void)hipSetDevice(devs[k]);
            if (send[k]) (void)hipFree(send[k]);
            if (recv[k]) (void)hipFree(recv[k]);
            if (streams[k]) (void)hipStreamDestroy(streams[k]);
            if (comms[k]) (void)rc.CommDestroy(comms[k]);
        }
        return SEPAIHRD_E_HIP;
    };
    for (int k = 0; k < n; ++k) {
        if (hipSetDevice(devs[k]) != hipSuccess || hipStreamCreateWithFlags(&streams[k], hipStreamNonBlocking) != hipSuccess ||
            hipMalloc((void**)&send[k], block * sizeof(double)) != hipSuccess || hipMalloc((void**)&recv[k], block * n * sizeof(double)) != hipSuccess ||
            hipMemsetAsync(send[k], 0, block * sizeof(double), streams[k]) != hipSuccess ||
            hipMemcpyAsync(send[k], ctxs[k]->rec_buf[0], (size_t)rows[k] * width * sizeof(double), hipMemcpyDeviceToDevice, streams[k]) != hipSuccess)
            return fail("staging buffers");
    }
    if ((rc_code = rc.CommInitAll(comms.data(), n, devs.data())) != 0) return fail("ncclCommInitAll");
    if ((rc_code = rc.GroupStart()) != 0) return fail("ncclGroupStart");
    for (int k = 0; k < n; ++k) {
        (void)hipSetDevice(devs[k]);
        if ((rc_code = rc.AllGather(send[k], recv[k], block, NCCL_DOUBLE, comms[k], streams[k])) != 0) { (void)rc.GroupEnd(); return fail("ncclAllGather"); }
    }
    if ((rc_code = rc.GroupEnd()) != 0) return fail("ncclGroupEnd");
    rc_code = 0;
    for (int k = 0; k < n; ++k) {  // compact: rank r's first rows[r] rows, in rank order
        (void)hipSetDevice(devs[k]);
        size_t off = 0;
        for (int r = 0; r < n; ++r) {
            if (rows[r] > 0 && hipMemcpyAsync(ctxs[k]->rec_buf[1] + off, recv[k] + (size_t)r * block, (size_t)rows[r] * width * sizeof(double),
                                              hipMemcpyDeviceToDevice, streams[k]) != hipSuccess)
                return fail("compaction");
            off += (size_t)rows[r] * width;
        }
    }
    for (int k = 0; k < n; ++k) {
        (void)hipSetDevice(devs[k]);
        if (hipStreamSynchronize(streams[k]) != hipSuccess) return fail("stream synchronisation");
    }
    for (int k = 0; k < n; ++k) {
        (void)hipSetDevice(devs[k]);
        (void)hipFree(send[k]); (void)hipFree(recv[k]); (void)hipStreamDestroy(streams[k]); (void)rc.CommDestroy(comms[k]);
    }
    return SEPAIHRD_OK;
}

}  // extern "C"
