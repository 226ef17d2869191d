// MultiChainMetropolisHastings.cpp -- Adaptive-Metropolis (Haario) with Robbins-Monro global scale,
// the reference's MetropolisHastingsSampler (src/sir_age_structured/optimizers/
// MetropolisHastingsSampler.cpp:65-412), advanced for C independent chains in lock-step so that
// step 3 of every iteration ("evaluate likelihood", :312) is one batched device call.
//
// Per-chain state is what the reference keeps as sampler members: current_covariance_,
// proposal_cholesky_, running_mean_, log_scale_/global_scale_, recent_accepts_, chain_history_
// and its own std::mt19937.  Host work per iteration is O(C P^2) and runs under OpenMP.
#include "epidemic_hip/HipSEPAIHRD.hpp"
#include "sepaihrd_hip.h"

#include <sched.h>

#include <algorithm>
#include <chrono>
#include <thread>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <limits>
#include <sstream>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace epidemic {

struct MultiChainMetropolisHastings::Chain {
    std::mt19937 gen;
    std::vector<double> x, prop, cov, chol, mean;
    double lp = 0.0, log_scale = 0.0, scale = 1.0;
    std::deque<int> recent;
    int emergency = 0, accepted = 0;
    std::vector<double> history;  // chain_history_: every state ((t+1) x P) for the two-pass refresh, else the newest only
    size_t history_len = 0;
    // recomputeFullCovariance from running sums (oracle::RunningMoments states the recurrence; the device kernels
    // follow it operation for operation): plain sum in the order of the reference's mean loop, Welford mean, centred
    // second moment (entries j <= i)
    std::vector<double> sum, wmean, m2;
    void push_moments(const double* x, int P) {
        const double n = static_cast<double>(history_len);  // the state just appended is the n-th
        const double rn = 1.0 / n, w = (n - 1.0) / n;
        std::vector<double> d(static_cast<size_t>(P));
        for (int i = 0; i < P; ++i) d[static_cast<size_t>(i)] = x[i] - wmean[static_cast<size_t>(i)];
        for (int i = 0; i < P; ++i)
            for (int j = 0; j <= i; ++j)
                m2[static_cast<size_t>(i) * P + j] += w * (d[static_cast<size_t>(i)] * d[static_cast<size_t>(j)]);
        for (int i = 0; i < P; ++i) wmean[static_cast<size_t>(i)] += d[static_cast<size_t>(i)] * rn;
        for (int i = 0; i < P; ++i) sum[static_cast<size_t>(i)] += x[i];
    }
};

namespace {
// lower Cholesky factor, row by row; false if not positive definite (Eigen::LLT::info() != Success)
bool cholesky(const std::vector<double>& A, int P, std::vector<double>& L) {
    std::vector<double> out(static_cast<size_t>(P) * P, 0.0);
    for (int j = 0; j < P; ++j) {
        double d = A[static_cast<size_t>(j) * P + j];
        for (int k = 0; k < j; ++k) d -= out[static_cast<size_t>(j) * P + k] * out[static_cast<size_t>(j) * P + k];
        if (!(d > 0.0)) return false;
        const double ljj = std::sqrt(d);
        out[static_cast<size_t>(j) * P + j] = ljj;
        for (int i = j + 1; i < P; ++i) {
            double s = A[static_cast<size_t>(i) * P + j];
            for (int k = 0; k < j; ++k) s -= out[static_cast<size_t>(i) * P + k] * out[static_cast<size_t>(j) * P + k];
            out[static_cast<size_t>(i) * P + j] = s / ljj;
        }
    }
    L.swap(out);
    return true;
}
// Host threads for the per-chain loops: the OpenMP default counts every hardware thread of the machine,
// a container usually owns a share of them (CPU affinity and / or a cgroup quota); oversubscribing the
// share turns every parallel region into a scheduling storm.  SEPAIHRD_CPU_THREADS overrides.
int host_thread_share() {
    int n = 1;
#ifdef _OPENMP
    n = omp_get_max_threads();
#endif
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = std::min(n, std::max(1, CPU_COUNT(&set)));
    if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char quota[64] = {0};
        double period = 0.0;
        if (std::fscanf(f, "%63s %lf", quota, &period) == 2 && std::strcmp(quota, "max") != 0 && period > 0.0)
            n = std::min(n, std::max(1, static_cast<int>(std::atof(quota) / period)));
        std::fclose(f);
    }
    if (const char* env = std::getenv("SEPAIHRD_CPU_THREADS")) n = std::max(1, std::min(n, std::atoi(env)));
    return std::max(1, n);
}

// std::normal_distribution<double>(0, 1) of libstdc++ (bits/random.tcc: Marsaglia polar method over
// generate_canonical<double, 53>) written out for std::mt19937: the same words consumed in the same order, the same
// double operations, so the same values -- without the long double arithmetic of generate_canonical and the
// distribution object's state.  A FRESH distribution per proposal (generateProposal :93-97): pairs are produced as
// (y mult, x mult) and the second value of the last pair is dropped when P is odd.  The device-resident loop below
// draws with it; optimizeChains() keeps std::normal_distribution, and the tests require both to give the same chains.
inline double canonical53(std::mt19937& g) {
    const double lo = static_cast<double>(g());               // sum += (urng() - min) * 1
    const double hi = static_cast<double>(g());               // sum += (urng() - min) * 2^32
    double ret = (lo + hi * 4294967296.0) / 18446744073709551616.0;
    if (__builtin_expect(ret >= 1.0, 0)) ret = std::nextafter(1.0, 0.0);
    return ret;
}
// A chain's random stream as a look-ahead queue of canonical uniforms (generate_canonical<double, 53>: always two
// 32-bit outputs each).  Both continuations of an accept test read the SAME queue -- the branch that draws the uniform
// takes element 0 for it and builds its normals from element 1 on, the other branch builds them from element 0 on -- and
// the test then consumes what the branch taken used.  Nothing is generated twice and no 2.5-KB generator is copied
// (the device-resident loop copied one per chain and iteration: 10 MB per iteration at 4096 chains).
struct CanonicalQueue {
    std::mt19937 gen;
    std::vector<double> buf;   // [head, tail): generated, not yet consumed
    size_t head = 0, tail = 0;
    void ensure(size_t n) {    // at least n unconsumed elements
        if (tail - head < n) refill(n);
    }
    void refill(size_t n) {
        if (buf.size() < std::max<size_t>(n, 320)) buf.resize(std::max<size_t>(2 * n, 320));  // 62 normals take ~80 elements
        if (head > 0) {
            std::memmove(buf.data(), buf.data() + head, (tail - head) * sizeof(double));
            tail -= head;
            head = 0;
        }
        // only what is asked for (plus a little): every chain consumes ~80 elements per iteration, and filling the
        // whole buffer would make ALL chains refill in the same iteration (they start together) -- a 0.5 ms hiccup of
        // the host every third iteration at 4096 chains, long enough for the device to run dry
        double* const b = buf.data();
        const size_t upto = std::min(buf.size(), n + 16);
        while (tail < upto) b[tail++] = canonical53(gen);
    }
    double at(size_t i) { ensure(i + 1); return buf[head + i]; }
    void consume(size_t n) { head += n; }
};
// std::normal_distribution<double> (polar method) over the queue from element `from` on; returns the elements it used
inline size_t standard_normals_from_queue(CanonicalQueue& q, size_t from, double* dst, int P) {
    size_t pos = from;
    q.ensure(from + static_cast<size_t>(P) + static_cast<size_t>(P) / 2 + 8);  // the usual need (27 % of the pairs are rejected)
    const double* b = q.buf.data() + q.head;
    size_t avail = q.tail - q.head;
    for (int i = 0; i < P; i += 2) {
        double x, y, r2;
        do {
            if (__builtin_expect(pos + 2 > avail, 0)) {  // a long run of rejections: more of the stream
                q.ensure(pos + 64);
                b = q.buf.data() + q.head;
                avail = q.tail - q.head;
            }
            x = 2.0 * b[pos] - 1.0;
            y = 2.0 * b[pos + 1] - 1.0;
            pos += 2;
            r2 = x * x + y * y;
        } while (r2 > 1.0 || r2 == 0.0);
        const double mult = std::sqrt(-2 * std::log(r2) / r2);
        dst[i] = (y * mult) * 1.0 + 0.0;
        if (i + 1 < P) dst[i + 1] = (x * mult) * 1.0 + 0.0;
    }
    return pos - from;
}
inline void draw_standard_normals(std::mt19937& g, double* dst, int P) {
    for (int i = 0; i < P; i += 2) {
        double x, y, r2;
        do {
            x = 2.0 * canonical53(g) - 1.0;
            y = 2.0 * canonical53(g) - 1.0;
            r2 = x * x + y * y;
        } while (r2 > 1.0 || r2 == 0.0);
        const double mult = std::sqrt(-2 * std::log(r2) / r2);
        dst[i] = (y * mult) * 1.0 + 0.0;                       // ret * stddev + mean
        if (i + 1 < P) dst[i + 1] = (x * mult) * 1.0 + 0.0;   // the saved value
    }
}

inline double sanitize(double v) { return (std::isnan(v) || std::isinf(v)) ? -1e18 : v; }  // safeEvaluate :65-74
}  // namespace

namespace {
// FileUtils::getProjectRoot (src/utils/FileUtils.cpp:24-45): the working directory or the nearest of its five parents
// that holds data/, include/ and src/; else the working directory
std::string project_root() {
    namespace fs = std::filesystem;
    std::error_code ec;
    fs::path cur = fs::current_path(ec);
    if (ec) return ".";
    std::vector<fs::path> roots{cur};
    fs::path p = cur;
    for (int i = 0; i < 5; ++i) {
        p = p.parent_path();
        if (!p.empty()) roots.push_back(p);
    }
    for (const fs::path& r : roots)
        if (fs::exists(r / "data", ec) && fs::exists(r / "include", ec) && fs::exists(r / "src", ec)) return fs::absolute(r).lexically_normal().string();
    return fs::absolute(cur).lexically_normal().string();
}
// the reference's Logger::formatLogMessage (include/utils/Logger.hpp:143-160)
std::string logger_line(const std::string& level, const std::string& source, const std::string& message) {
    std::ostringstream oss;
    const std::time_t now = std::chrono::system_clock::to_time_t(std::chrono::system_clock::now());
    std::tm tm_buf{};
    localtime_r(&now, &tm_buf);
    oss << std::put_time(&tm_buf, "%Y-%m-%d %H:%M:%S") << " " << (level == "WARNING" ? "[WARNING]" : "[INFO]   ") << " [" << source << "] " << message;
    return oss.str();
}
}  // namespace

// Progress lines and trace files of one run (MetropolisHastingsSampler.cpp:363-383,399-411,440-469) for the first k chains.
// The samples of a reported chain accumulate here as they arrive (host loops hand them over as they are stored, the
// device-resident loop in snapshots), so a checkpoint is always written from host memory.
struct MultiChainMetropolisHastings::Reporter {
    const MultiChainMetropolisHastings& mh;
    std::vector<std::string> names;
    int P = 0, k = 0;
    std::string dir;
    std::vector<std::vector<double>> samples, values;  // per reported chain: [n][P], [n]
    Reporter(const MultiChainMetropolisHastings& owner, IParameterManager& pm, int C) : mh(owner), names(pm.getParameterNames()) {
        P = static_cast<int>(pm.getParameterCount());
        k = std::max(0, std::min(owner.checkpoint_chains_, C));
        samples.assign(static_cast<size_t>(k), {});
        values.assign(static_cast<size_t>(k), {});
        if (filesWanted()) dir = owner.output_dir_.empty() ? project_root() + "/data/mcmc_samples" : owner.output_dir_;
    }
    bool reportsWanted() const { return k > 0 && mh.report_interval_ > 0; }
    bool filesWanted() const { return k > 0 && mh.store_samples_ && (mh.write_checkpoints_ || mh.write_trace_); }
    bool due(int t) const { return reportsWanted() && (t + 1) % mh.report_interval_ == 0; }  // :363
    void say(const std::string& level, const std::string& msg) const {
        if (mh.progress_sink_) mh.progress_sink_(level, msg);
        else std::cout << logger_line(level, "MetropolisHastingsSampler", msg) << std::endl;
    }
    void append(int chain, const double* rows, const double* vals, size_t n) {
        samples[static_cast<size_t>(chain)].insert(samples[static_cast<size_t>(chain)].end(), rows, rows + n * static_cast<size_t>(P));
        values[static_cast<size_t>(chain)].insert(values[static_cast<size_t>(chain)].end(), vals, vals + n);
    }
    // the progress line of iteration t for reported chain `chain` (:364-379)
    void line(int chain, int t, double lp, double best, long accepted, double scale) const {
        const double acc_rate = static_cast<double>(accepted) / (t + 1);
        char buf[256];
        std::snprintf(buf, sizeof(buf), "%sIter: %6d | LogPost: %.2f | Best: %.2f | AccRate: %.1f%% | Scale: %.3f",
                      chain > 0 ? ("[chain " + std::to_string(chain) + "] ").c_str() : "", t + 1, lp, best, acc_rate * 100.0, scale);
        say("INFO", buf);
        if (t > mh.burn_in_ && acc_rate < 0.05) say("WARNING", "Acceptance rate low (<5%). Scale adaptation working to correct.");
        else if (t > mh.burn_in_ && acc_rate > 0.50) say("WARNING", "Acceptance rate high (>50%). Scale adaptation working to correct.");
    }
    std::string path(const std::string& stem, int chain) const {
        return dir + "/" + stem + (chain > 0 ? "_chain" + std::to_string(chain) : "") + ".csv";
    }
    // saveCheckpoint / saveSamplesToCSV (:414-469): `iter,log_posterior,<names>`, rows `i,value,theta...` in
    // std::scientific << setprecision(6) (the manipulators stay set for the parameters); checkpoint: the last <= 5000 samples
    void writeFile(const std::string& file, int chain, size_t first) const {
        std::error_code ec;
        std::filesystem::create_directories(dir, ec);
        // written under another name and moved into place: a reader never sees half a checkpoint
        const std::string tmp = file + ".part";
        {
            std::ofstream out(tmp);
            if (!out.is_open()) { say("WARNING", "Failed to open CSV for writing: " + file); return; }
            out << "iter,log_posterior";
            for (const std::string& n : names) out << "," << n;
            out << "\n";
            const std::vector<double>& s = samples[static_cast<size_t>(chain)];
            const std::vector<double>& v = values[static_cast<size_t>(chain)];
            std::string row;
            char num[40];
            for (size_t i = first; i < v.size(); ++i) {
                row = std::to_string(i);
                std::snprintf(num, sizeof(num), ",%.6e", v[i]);
                row += num;
                for (int j = 0; j < P; ++j) {
                    std::snprintf(num, sizeof(num), ",%.6e", s[i * static_cast<size_t>(P) + j]);
                    row += num;
                }
                row += "\n";
                out << row;
            }
        }
        std::filesystem::rename(tmp, file, ec);
        if (ec) say("WARNING", "Failed to move " + tmp + " to " + file);
    }
    void checkpoint() const {  // :380-382 -> saveCheckpoint(result, pm, false)
        if (!(mh.write_checkpoints_ && mh.store_samples_)) return;
        for (int c = 0; c < k; ++c) {
            const size_t n = values[static_cast<size_t>(c)].size();
            if (n == 0) continue;
            writeFile(path("posterior_trace_checkpoint", c), c, n > 5000 ? n - 5000 : 0);
        }
    }
    void finish() const {  // :399-411
        if (!mh.store_samples_) return;
        for (int c = 0; c < k; ++c) {
            if (values[static_cast<size_t>(c)].empty()) continue;
            if (mh.write_checkpoints_) {
                writeFile(path("posterior_trace_final", c), c, 0);
                say("INFO", "Full posterior trace saved to: " + path("posterior_trace_final", c));
            }
            if (mh.write_trace_) {
                writeFile(path("posterior_trace", c), c, 0);
                say("INFO", "Full posterior trace saved to: " + path("posterior_trace", c));
            }
        }
    }
};

void MultiChainMetropolisHastings::configure(const std::map<std::string, double>& settings) {
    auto get = [&](const char* key, double def) {
        auto it = settings.find(key);
        return it != settings.end() ? it->second : def;
    };
    iterations_ = static_cast<int>(get("mcmc_iterations", 10000.0));
    burn_in_ = static_cast<int>(get("burn_in", 1000.0));
    adaptation_period_ = static_cast<int>(get("adaptation_period", 100.0));
    thinning_ = std::max(1, static_cast<int>(get("thinning", 1.0)));
    regularization_epsilon_ = get("regularization_epsilon", 1e-6);
    target_acceptance_rate_ = get("target_acceptance_rate", 0.234);
    adapt_scale_ = get("adapt_scale", 1.0) != 0.0;
    store_samples_ = get("store_samples", 1.0) != 0.0;
    // build-side keys (not in the reference's mcmc_settings.txt): the literal two-pass covariance refresh over the whole
    // history instead of running co-moments, and the ring of newest states the device keeps (default: the period, >= 128)
    two_pass_covariance_ = get("two_pass_covariance", two_pass_covariance_ ? 1.0 : 0.0) != 0.0;
    adaptation_window_ = static_cast<int>(get("adaptation_window", static_cast<double>(adaptation_window_)));
    device_streams_ = get("device_streams", device_streams_ ? 1.0 : 0.0) != 0.0;
    keep_traces_ = get("keep_accept_traces", keep_traces_ ? 1.0 : 0.0) != 0.0;
    // the reference's reporting and I/O keys (MetropolisHastingsSampler.cpp:34,41-43), and how many chains they cover here
    report_interval_ = static_cast<int>(get("report_interval", 100.0));
    write_checkpoints_ = get("write_checkpoints", 1.0) != 0.0;
    write_trace_ = get("write_trace", 1.0) != 0.0;
    checkpoint_chains_ = static_cast<int>(get("checkpoint_chains", static_cast<double>(checkpoint_chains_)));
}

OptimizationResult MultiChainMetropolisHastings::optimize(const Eigen::VectorXd& x0, IObjectiveFunction& objective,
                                                          IParameterManager& pm) {
    const int P = static_cast<int>(x0.size());
    std::vector<double> init(x0.data(), x0.data() + P);
    BatchEval eval = [&](const double* th, int B, double* out) {
        for (int b = 0; b < B; ++b) {
            Eigen::VectorXd v(P);
            for (int i = 0; i < P; ++i) v[i] = th[static_cast<size_t>(b) * P + i];
            try { out[b] = sanitize(objective.calculate(v)); } catch (...) { out[b] = -1e18; }
        }
    };
    return run(init, 1, eval, pm)[0];
}

std::vector<OptimizationResult> MultiChainMetropolisHastings::optimizeChains(const std::vector<double>& initial, int C,
                                                                             IBatchObjectiveFunction& objective,
                                                                             IParameterManager& pm) {
    std::vector<int> status;
    BatchEval eval = [&](const double* th, int B, double* out) {
        status.resize(static_cast<size_t>(B));
        objective.calculateBatch(th, B, out, status.data());
        for (int b = 0; b < B; ++b)  // an integration failure is the exception safeEvaluate swallows
            out[b] = status[static_cast<size_t>(b)] >= 2 ? -1e18 : sanitize(out[b]);
    };
    return run(initial, C, eval, pm);
}

std::vector<OptimizationResult> MultiChainMetropolisHastings::run(const std::vector<double>& initial, int C,
                                                                  const BatchEval& eval, IParameterManager& pm) {
    // :207-209 reflection mode for valid Bayesian sampling
    if (auto* hpm = dynamic_cast<HipSEPAIHRDParameterManager*>(&pm)) hpm->setConstraintMode(ConstraintMode::MCMC_REFLECT);
    const int P = static_cast<int>(pm.getParameterCount());
    if (static_cast<int>(initial.size()) != C * P) throw InvalidParameterException("MetropolisHastingsSampler", "initial size != C*P");
    const double scaling_factor = (2.38 * 2.38) / static_cast<double>(P);
    const size_t PP = static_cast<size_t>(P) * P;
    const int nthreads = std::min(host_threads_ > 0 ? std::min(host_threads_, host_thread_share()) : host_thread_share(), std::max(1, C));
    (void)nthreads;

    std::vector<Chain> chains(static_cast<size_t>(C));
    std::vector<double> batch(static_cast<size_t>(C) * P), values(static_cast<size_t>(C));
    for (int c = 0; c < C; ++c) {
        Chain& ch = chains[static_cast<size_t>(c)];
        ch.gen.seed(seed_ + static_cast<uint32_t>(c));
        ch.x.assign(initial.begin() + static_cast<size_t>(c) * P, initial.begin() + static_cast<size_t>(c + 1) * P);
        ch.prop.resize(static_cast<size_t>(P));
        if (initial_cov_.size() == PP) {  // warm start from phase 1 (:219-223): no 2.38^2/P scaling
            ch.cov = initial_cov_;
        } else {
            ch.cov.assign(PP, 0.0);
            for (int i = 0; i < P; ++i) {  // :226-237
                const double s = pm.getSigmaForParamIndex(i);
                ch.cov[static_cast<size_t>(i) * P + i] = (s > 0 ? s * s : 1e-6);
            }
            for (double& v : ch.cov) v *= scaling_factor;
        }
        for (int i = 0; i < P; ++i) ch.cov[static_cast<size_t>(i) * P + i] += regularization_epsilon_;
        if (!cholesky(ch.cov, P, ch.chol)) {  // :240-246
            ch.chol.assign(PP, 0.0);
            for (int i = 0; i < P; ++i) ch.chol[static_cast<size_t>(i) * P + i] = 0.1;
        }
        ch.mean = ch.x;
        if (two_pass_covariance_) ch.history.reserve(static_cast<size_t>(iterations_) * P);
        else {
            ch.sum.assign(static_cast<size_t>(P), 0.0);
            ch.wmean.assign(static_cast<size_t>(P), 0.0);
            ch.m2.assign(PP, 0.0);
        }
        std::copy(ch.x.begin(), ch.x.end(), batch.begin() + static_cast<size_t>(c) * P);
    }
    eval(batch.data(), C, values.data());  // initial state :257

    std::vector<OptimizationResult> results(static_cast<size_t>(C));
    Reporter reporter(*this, pm, C);
    traces_.assign(static_cast<size_t>(C), {});
    auto to_eigen = [P](const std::vector<double>& v) {
        Eigen::VectorXd e(P);
        for (int i = 0; i < P; ++i) e[i] = v[static_cast<size_t>(i)];
        return e;
    };
    for (int c = 0; c < C; ++c) {
        Chain& ch = chains[static_cast<size_t>(c)];
        ch.lp = values[static_cast<size_t>(c)];
        ch.history.insert(ch.history.end(), ch.x.begin(), ch.x.end());
        ch.history_len = 1;
        if (!two_pass_covariance_) ch.push_moments(ch.x.data(), P);
        OptimizationResult& r = results[static_cast<size_t>(c)];
        if (store_samples_) { r.samples.push_back(to_eigen(ch.x)); r.sampleObjectiveValues.push_back(ch.lp); }
        r.bestParameters = to_eigen(ch.x);
        r.bestObjectiveValue = ch.lp;
        if (keep_traces_) traces_[static_cast<size_t>(c)].reserve(static_cast<size_t>(std::max(iterations_ - 1, 0)));
        if (store_samples_ && c < reporter.k) reporter.append(c, ch.x.data(), &ch.lp, 1);
    }

    for (int t = 1; t < iterations_; ++t) {
        // ---- 1+2: adaptation and proposal, independent per chain
#pragma omp parallel for schedule(static) num_threads(nthreads)
        for (int c = 0; c < C; ++c) {
            Chain& ch = chains[static_cast<size_t>(c)];
            if (t > burn_in_) {
                {   // updateCovarianceRank1 :154-166, gamma = 10/(t+100)
                    const double* ns = &ch.history[ch.history.size() - static_cast<size_t>(P)];  // chain_history_.back()
                    const double gamma = 10.0 / (t + 100.0);
                    std::vector<double> diff(static_cast<size_t>(P));
                    for (int i = 0; i < P; ++i) diff[static_cast<size_t>(i)] = ns[i] - ch.mean[static_cast<size_t>(i)];
                    for (int i = 0; i < P; ++i) ch.mean[static_cast<size_t>(i)] += gamma * diff[static_cast<size_t>(i)];
                    for (int i = 0; i < P; ++i)
                        for (int j = 0; j < P; ++j)
                            ch.cov[static_cast<size_t>(i) * P + j] =
                                (1.0 - gamma) * ch.cov[static_cast<size_t>(i) * P + j] +
                                gamma * (diff[static_cast<size_t>(i)] * diff[static_cast<size_t>(j)]);
                }
                if (t % adaptation_period_ == 0) {
                    if (ch.history_len >= static_cast<size_t>(P) + 10 && !two_pass_covariance_) {  // recomputeFullCovariance :168-199
                        const double len = static_cast<double>(ch.history_len), denom = static_cast<double>(ch.history_len - 1);
                        for (int i = 0; i < P; ++i) ch.mean[static_cast<size_t>(i)] = ch.sum[static_cast<size_t>(i)] / len;
                        for (int i = 0; i < P; ++i)
                            for (int j = 0; j < P; ++j) {
                                const double m = i >= j ? ch.m2[static_cast<size_t>(i) * P + j] : ch.m2[static_cast<size_t>(j) * P + i];
                                ch.cov[static_cast<size_t>(i) * P + j] = scaling_factor * (m / denom) + (i == j ? regularization_epsilon_ : 0.0);
                            }
                        cholesky(ch.cov, P, ch.chol);  // kept only on success
                    } else if (ch.history_len >= static_cast<size_t>(P) + 10) {
                        std::vector<double> mean(static_cast<size_t>(P), 0.0), acc(PP, 0.0);
                        for (size_t s = 0; s < ch.history_len; ++s)
                            for (int i = 0; i < P; ++i) mean[static_cast<size_t>(i)] += ch.history[s * P + i];
                        for (int i = 0; i < P; ++i) mean[static_cast<size_t>(i)] /= static_cast<double>(ch.history_len);
                        ch.mean = mean;
                        for (size_t s = 0; s < ch.history_len; ++s) {
                            const double* row = &ch.history[s * P];
                            for (int i = 0; i < P; ++i) {
                                const double di = row[i] - mean[static_cast<size_t>(i)];
                                for (int j = 0; j < P; ++j)
                                    acc[static_cast<size_t>(i) * P + j] += di * (row[j] - mean[static_cast<size_t>(j)]);
                            }
                        }
                        const double denom = double(ch.history_len - 1);
                        for (int i = 0; i < P; ++i)
                            for (int j = 0; j < P; ++j)
                                ch.cov[static_cast<size_t>(i) * P + j] =
                                    scaling_factor * (acc[static_cast<size_t>(i) * P + j] / denom) +
                                    (i == j ? regularization_epsilon_ : 0.0);
                        cholesky(ch.cov, P, ch.chol);  // kept only on success
                    }
                    std::vector<double> stable = ch.cov;  // :295-300: epsilon once more before the kept LLT
                    for (int i = 0; i < P; ++i) stable[static_cast<size_t>(i) * P + i] += regularization_epsilon_;
                    cholesky(stable, P, ch.chol);
                }
            }
            // generateProposal :91-102
            std::vector<double> z(static_cast<size_t>(P));
            {
                std::normal_distribution<double> dist(0.0, 1.0);
                for (int i = 0; i < P; ++i) z[static_cast<size_t>(i)] = dist(ch.gen);
            }
            Eigen::VectorXd raw(P);
            for (int i = 0; i < P; ++i) {
                double s = 0.0;
                for (int j = 0; j <= i; ++j) s += ch.chol[static_cast<size_t>(i) * P + j] * z[static_cast<size_t>(j)];
                raw[i] = ch.x[static_cast<size_t>(i)] + ch.scale * s;
            }
            const Eigen::VectorXd cons = pm.applyConstraints(raw);  // :309 (const, thread-safe)
            for (int i = 0; i < P; ++i) {
                ch.prop[static_cast<size_t>(i)] = cons[i];
                batch[static_cast<size_t>(c) * P + i] = cons[i];
            }
        }
        // ---- 3: one batched evaluation for all chains
        eval(batch.data(), C, values.data());
        // ---- 4-7: accept / reject, scale adaptation, bookkeeping
#pragma omp parallel for schedule(static) num_threads(nthreads)
        for (int c = 0; c < C; ++c) {
            Chain& ch = chains[static_cast<size_t>(c)];
            OptimizationResult& r = results[static_cast<size_t>(c)];
            const double prop_lp = values[static_cast<size_t>(c)];
            const double log_ratio = prop_lp - ch.lp;
            bool accept = false;
            if (log_ratio >= 0.0) accept = true;
            else {
                std::uniform_real_distribution<double> u_dist(0.0, 1.0);  // stateless: same draws as one shared object
                if (std::log(u_dist(ch.gen)) < log_ratio) accept = true;
            }
            if (accept) {
                ch.x = ch.prop;
                ch.lp = prop_lp;
                ch.accepted++;
                if (ch.lp > r.bestObjectiveValue) { r.bestObjectiveValue = ch.lp; r.bestParameters = to_eigen(ch.x); }
            }
            if (keep_traces_) traces_[static_cast<size_t>(c)].push_back(accept ? 1 : 0);
            if (adapt_scale_) {  // adaptGlobalScale :104-152
                ch.recent.push_back(accept ? 1 : 0);
                if (ch.recent.size() > 1000) ch.recent.pop_front();
                double rate = 0.0;
                if (!ch.recent.empty()) {
                    int sum = 0;
                    for (int a : ch.recent) sum += a;
                    rate = static_cast<double>(sum) / ch.recent.size();
                }
                if (ch.recent.size() >= 1000 && rate < 0.001) { ch.log_scale -= 0.7; ch.emergency++; }
                else if (rate < 0.02 && ch.recent.size() >= 500) {
                    double g = 5.0 / std::sqrt(static_cast<double>(t) + 1.0);
                    g = std::min(g, 0.3);
                    ch.log_scale += g * (0.0 - target_acceptance_rate_);
                } else {
                    double g = 1.0 / std::sqrt(static_cast<double>(t) + 1.0);
                    g = std::min(g, 0.1);
                    ch.log_scale += g * ((accept ? 1.0 : 0.0) - target_acceptance_rate_);
                }
                if (ch.scale <= 0.011 && rate > 0.15 && rate < 0.30) ch.log_scale += 0.01;
                ch.log_scale = std::max(std::min(ch.log_scale, 2.3), -6.9);
                ch.scale = std::exp(ch.log_scale);
            }
            if (two_pass_covariance_) ch.history.insert(ch.history.end(), ch.x.begin(), ch.x.end());
            else std::copy(ch.x.begin(), ch.x.end(), ch.history.begin());  // the newest state is all the rank-one update reads
            ch.history_len++;
            if (!two_pass_covariance_) ch.push_moments(ch.x.data(), P);
            if (store_samples_ && (t % thinning_ == 0)) {
                r.samples.push_back(to_eigen(ch.x));
                r.sampleObjectiveValues.push_back(ch.lp);
            }
        }
        // ---- 8: logging and checkpointing of the reported chains (:362-383)
        for (int c = 0; c < reporter.k; ++c) {
            const Chain& ch = chains[static_cast<size_t>(c)];
            if (store_samples_ && (t % thinning_ == 0)) reporter.append(c, ch.x.data(), &ch.lp, 1);
            if (reporter.due(t)) reporter.line(c, t, ch.lp, results[static_cast<size_t>(c)].bestObjectiveValue, ch.accepted, ch.scale);
        }
        if (reporter.due(t)) reporter.checkpoint();
    }

    for (int c = 0; c < C; ++c) {
        Chain& ch = chains[static_cast<size_t>(c)];
        OptimizationResult& r = results[static_cast<size_t>(c)];
        r.finalCovariance = Eigen::MatrixXd(P, P);
        for (int i = 0; i < P; ++i)
            for (int j = 0; j < P; ++j) r.finalCovariance(i, j) = ch.cov[static_cast<size_t>(i) * P + j];
        r.additionalStats["acceptance_rate"] = static_cast<double>(ch.accepted) / iterations_;  // :387
        r.additionalStats["accepted_count"] = ch.accepted;
        r.additionalStats["final_scale"] = ch.scale;
        r.additionalStats["burn_in"] = static_cast<double>(burn_in_);
        r.additionalStats["total_iterations"] = static_cast<double>(iterations_);
    }
    reporter.finish();
    return results;
}

// ------------------------------------------------------------------ device-resident state
std::vector<OptimizationResult> MultiChainMetropolisHastings::optimizeChainsOnDevice(
    const std::vector<double>& initial, int C, HipSEPAIHRDObjectiveFunction& objective, IParameterManager& pm) {
    if (auto* spm = dynamic_cast<HipSEPAIHRDParameterManager*>(&pm)) spm->setConstraintMode(ConstraintMode::MCMC_REFLECT);  // :207-209
    objective.syncDeviceConstraintMode();
    const int P = static_cast<int>(pm.getParameterCount());
    if (static_cast<int>(initial.size()) != C * P) throw InvalidParameterException("MetropolisHastingsSampler", "initial size != C*P");
    const double scaling_factor = (2.38 * 2.38) / static_cast<double>(P);
    const size_t PP = static_cast<size_t>(P) * P;
    const int nthreads = std::min(host_threads_ > 0 ? std::min(host_threads_, host_thread_share()) : host_thread_share(), std::max(1, C));
    (void)nthreads;

    // initial covariance of every chain (:219-237)
    std::vector<double> cov0;
    if (initial_cov_.size() == PP) {
        cov0 = initial_cov_;
    } else {
        cov0.assign(PP, 0.0);
        for (int i = 0; i < P; ++i) {
            const double s = pm.getSigmaForParamIndex(i);
            cov0[static_cast<size_t>(i) * P + i] = (s > 0 ? s * s : 1e-6);
        }
        for (double& v : cov0) v *= scaling_factor;
    }
    for (int i = 0; i < P; ++i) cov0[static_cast<size_t>(i) * P + i] += regularization_epsilon_;

    sepaihrd_ctx* ctx = objective.deviceContext();
    Reporter reporter(*this, pm, C);
    // the device may draw the streams only if its log / exp ARE this host's libm (sepaihrd_device_libm_check)
    bool device_streams = this->device_streams_;
    device_streams_fell_back_ = false;
    if (device_streams) {
        int32_t n_log = 0, n_exp = 0;
        if (sepaihrd_device_libm_check(ctx, &n_log, &n_exp) != SEPAIHRD_OK)
            throw ModelException("MetropolisHastingsSampler", std::string("sepaihrd_device_libm_check: ") + sepaihrd_last_error(ctx));
        if (n_log != 0 || n_exp != 0) {
            device_streams = false;
            device_streams_fell_back_ = true;
            reporter.say("WARNING", "the device's log / exp differ from this host's libm on " + std::to_string(n_log) + " (log) and " +
                                        std::to_string(n_exp) + " (exp) of 4096 self-check arguments (csrc/sepaihrd_rng.inc restates glibc 2.35's "
                                        "x86-64 FMA variants): random streams and scale adaptation stay on the host for this run");
        }
    }
    sepaihrd_mh_config mcfg{};
    mcfg.chains = C;
    mcfg.iterations = std::max(iterations_, 1);
    mcfg.thinning = store_samples_ ? thinning_ : 0;
    // every catch-up of the queued updates then falls on a refresh (one launch per adaptation period)
    mcfg.adaptation_window = adaptation_window_ > 0 ? adaptation_window_ : std::max(adaptation_period_ + 1, 128);
    mcfg.covariance_mode = two_pass_covariance_ ? SEPAIHRD_MH_COV_TWO_PASS : SEPAIHRD_MH_COV_RUNNING;
    mcfg.reg_eps = regularization_epsilon_;
    mcfg.scaling_factor = scaling_factor;
    sepaihrd_mh* mh = sepaihrd_mh_create(ctx, &mcfg, initial.data(), cov0.data());
    if (!mh) throw ModelException("MetropolisHastingsSampler", std::string("sepaihrd_mh_create: ") + sepaihrd_last_error(ctx));
    struct Guard { sepaihrd_mh* p; ~Guard() { sepaihrd_mh_destroy(p); } } guard{mh};
    auto check = [&](int rc, const char* what) {
        if (rc != SEPAIHRD_OK) throw ModelException("MetropolisHastingsSampler", std::string(what) + ": " + sepaihrd_last_error(ctx));
    };

    // What the host keeps per chain.  The random stream is consumed in the reference's order: the normals of a
    // proposal, then ONE uniform only if log_ratio < 0 (:327), then the next proposal's normals.  While the device
    // evaluates proposal t the host draws BOTH continuations of proposal t + 1 from the chain's queue (with and without
    // the uniform in front) and hands them to the device, which runs the accept test itself; the host's own state
    // (stream position, accept window, scale) catches up one evaluation later.
    struct Light {
        CanonicalQueue rng;   // the chain's stream, with look-ahead
        size_t used_likely = 0, used_alt = 0;  // queue elements the two continuations of the pending accept test take
        double log_u = 0.0;   // log of the uniform of the branch that draws it
        double lp = 0.0, log_scale = 0.0, scale = 1.0, best = 0.0;
        std::vector<unsigned char> recent;  // ring of the last 1000 accept flags (:107-110) + their sum
        size_t recent_pos = 0, recent_len = 0;
        int recent_sum = 0;
        int emergency = 0, accepted = 0;
        double cand_log_scale[2] = {0.0, 0.0}, cand_scale[2] = {1.0, 1.0};  // adaptGlobalScale for reject / accept, prepared ahead
        std::vector<double> best_x;
        std::vector<double> sample_values;
    };
    std::vector<Light> chains(static_cast<size_t>(C));
    const size_t CP = static_cast<size_t>(C) * P;
    std::vector<double> values(static_cast<size_t>(C)), scale(static_cast<size_t>(C));
    double* z_next = sepaihrd_mh_staging_buffer(mh);  // page-locked; re-fetched after every staging (two alternate)
    std::vector<int32_t> status(static_cast<size_t>(C));
    auto sanitize_all = [&]() {
        for (int c = 0; c < C; ++c)
            values[static_cast<size_t>(c)] = status[static_cast<size_t>(c)] >= 2 ? -1e18 : sanitize(values[static_cast<size_t>(c)]);
    };
    check(sepaihrd_mh_evaluate_current(mh, values.data(), status.data()), "mh_evaluate_current");  // :257
    sanitize_all();
    traces_.assign(static_cast<size_t>(C), {});
    // generateProposal :91-102 over the chain's queue from element `from` on; returns the elements used
    auto draw_normals = [P](CanonicalQueue& q, size_t from, double* dst) { return standard_normals_from_queue(q, from, dst, P); };
    // adaptGlobalScale (:104-152) as a function of the accept flag: the new log-scale, without touching the chain.
    // Evaluated twice per chain AHEAD of the accept test (while the device works) so that the exp() of the branch
    // taken is ready, and once more in the test itself for the ring bookkeeping -- same arithmetic both times.
    const double target = target_acceptance_rate_;
    auto next_log_scale = [target](const Light& ch, bool acc, int t, bool* emergency_hit) -> double {
        size_t len = ch.recent_len;
        int sum = ch.recent_sum;
        if (len == 1000) sum -= ch.recent[ch.recent_pos]; else ++len;
        sum += acc ? 1 : 0;
        const double rate = static_cast<double>(sum) / static_cast<double>(len);
        double ls = ch.log_scale;
        *emergency_hit = false;
        if (len >= 1000 && rate < 0.001) { ls -= 0.7; *emergency_hit = true; }
        else if (rate < 0.02 && len >= 500) {
            double g = 5.0 / std::sqrt(static_cast<double>(t) + 1.0);
            g = std::min(g, 0.3);
            ls += g * (0.0 - target);
        } else {
            double g = 1.0 / std::sqrt(static_cast<double>(t) + 1.0);
            g = std::min(g, 0.1);
            ls += g * ((acc ? 1.0 : 0.0) - target);
        }
        if (ch.scale <= 0.011 && rate > 0.15 && rate < 0.30) ls += 0.01;
        return std::max(std::min(ls, 2.3), -6.9);
    };
#pragma omp parallel for schedule(static) num_threads(nthreads)
    for (int c = 0; c < C; ++c) {
        Light& ch = chains[static_cast<size_t>(c)];
        ch.rng.gen.seed(seed_ + static_cast<uint32_t>(c));
        ch.lp = ch.best = values[static_cast<size_t>(c)];
        ch.best_x.assign(initial.begin() + static_cast<size_t>(c) * P, initial.begin() + static_cast<size_t>(c + 1) * P);
        ch.recent.assign(1000, 0);
        if (store_samples_) ch.sample_values.push_back(ch.lp);
        if (keep_traces_) traces_[static_cast<size_t>(c)].reserve(static_cast<size_t>(std::max(iterations_ - 1, 0)));
        if (!device_streams) ch.rng.consume(draw_normals(ch.rng, 0, &z_next[static_cast<size_t>(c) * P]));  // proposal 1
        scale[static_cast<size_t>(c)] = ch.scale;
    }

    const bool profile = std::getenv("SEPAIHRD_MH_PROFILE") != nullptr;
    double t_launch = 0, t_prepare = 0, t_wait = 0, t_book = 0;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<double>(b - a).count();
    };
    // adaptation step that precedes proposal t (:279-301; the history then holds t rows, states 0 .. t-1)
    auto adapt_mode = [&](int t) -> int {
        if (t <= burn_in_) return 0;
        if (t % adaptation_period_ != 0) return 1;
        return static_cast<size_t>(t) >= static_cast<size_t>(P) + 10 ? 3 : 2;
    };
    // The accept test itself runs on the device (sepaihrd_mh_step_tested): the host hands it log(u), both outcomes of
    // the scale adaptation and the normals of both continuations BEFORE the evaluation is over, so the device goes from
    // one evaluation to the next without waiting for the host; the host learns the outcome (flags, the values compared)
    // at the start of the next evaluation and does its bookkeeping for iteration t while evaluation t + 1 runs.
    // With the streams on the device the scale adaptation goes there too (adaptGlobalScale in the test kernel): the sampler is
    // self-contained, this loop only queues iterations, and what the host kept per chain is read back at the end
    if (device_streams) check(sepaihrd_mh_keep_scale_on_device(mh, adapt_scale_ ? 1 : 0, target_acceptance_rate_, keep_traces_ ? 1 : 0), "mh_keep_scale_on_device");
    check(sepaihrd_mh_set_values(mh, values.data()), "mh_set_values");
    double* const test = sepaihrd_mh_test_buffer(mh);  // [log_u C][scale if rejected C][scale if accepted C][z_plain C*P]
    double* const t_log_u = test;
    double* const t_scale_reject = test + C;
    double* const t_scale_accept = test + 2 * static_cast<size_t>(C);
    double* const z_plain = test + 3 * static_cast<size_t>(C);
    std::vector<uint8_t> flags(static_cast<size_t>(C));
    // what the host keeps of iteration t once the test's outcome is known (:327-371 and adaptGlobalScale :104-152)
    auto book = [&](int t, bool drew_next) {
#pragma omp parallel for schedule(static) num_threads(nthreads)
        for (int c = 0; c < C; ++c) {
            Light& ch = chains[static_cast<size_t>(c)];
            const uint8_t f = flags[static_cast<size_t>(c)];
            const bool acc = (f & 1) != 0;
            // the stream moves by what the continuation taken drew: no uniform (bit 2) -> the plain normals only
            if (!device_streams) ch.rng.consume((f & 4) ? (drew_next ? ch.used_alt : 0) : ch.used_likely);
            if (acc) {
                ch.lp = values[static_cast<size_t>(c)];
                ch.accepted++;
                if (f & 2) ch.best = ch.lp;  // the device compared against the same best value and keeps the state
            }
            if (keep_traces_) traces_[static_cast<size_t>(c)].push_back(acc ? 1 : 0);
            if (adapt_scale_) {
                bool emergency_hit;
                const double ls = next_log_scale(ch, acc, t, &emergency_hit);
                if (ch.recent_len == 1000) ch.recent_sum -= ch.recent[ch.recent_pos]; else ch.recent_len++;
                ch.recent[ch.recent_pos] = acc ? 1 : 0;
                ch.recent_sum += acc ? 1 : 0;
                ch.recent_pos = (ch.recent_pos + 1) % 1000;
                if (emergency_hit) ch.emergency++;
                ch.log_scale = ls;
                ch.scale = (ls == ch.cand_log_scale[acc ? 1 : 0]) ? ch.cand_scale[acc ? 1 : 0] : std::exp(ls);
            }
            if (store_samples_ && (t % thinning_ == 0)) ch.sample_values.push_back(ch.lp);
        }
    };
    // reports and checkpoints of the first reporter.k chains
    std::vector<int32_t> rep_chains(static_cast<size_t>(reporter.k));
    for (int c = 0; c < reporter.k; ++c) rep_chains[static_cast<size_t>(c)] = c;
    int rep_fetched = 0;  // samples of the reported chains already handed to the reporter
    // host-side streams: the host has the chains' values, counts and scales of iteration t; the new samples of the reported
    // chains come through the same snapshot, waited for at once (this loop waits for the device every iteration anyway)
    auto report_from_host = [&](int t) {
        const int ns_now = store_samples_ ? std::min(sepaihrd_mh_sample_count(mh), t / thinning_ + 1) : 0;
        const int count = std::max(ns_now - rep_fetched, 0);
        const size_t k = static_cast<size_t>(reporter.k), n = static_cast<size_t>(count);
        std::vector<double> smp(k * n * P);
        if (n) {
            check(sepaihrd_mh_snapshot_begin(mh, rep_chains.data(), reporter.k, rep_fetched, count), "mh_snapshot_begin");
            check(sepaihrd_mh_snapshot_end(mh, 1, nullptr, smp.data(), nullptr), "mh_snapshot_end");
        }
        for (size_t c = 0; c < k; ++c) {
            const Light& ch = chains[c];
            if (n) reporter.append(static_cast<int>(c), &smp[c * n * P], &ch.sample_values[static_cast<size_t>(rep_fetched)], n);
            reporter.line(static_cast<int>(c), t, ch.lp, ch.best, ch.accepted, ch.scale);
        }
        rep_fetched = ns_now;
        reporter.checkpoint();
    };
    const auto loop_begin = now();
    if (device_streams) check(sepaihrd_mh_seed_streams(mh, seed_), "mh_seed_streams");  // chain c: mt19937(seed + c), as the host's
    if (iterations_ > 1) {  // proposal 1: nothing to test yet
        if (device_streams) {
            check(sepaihrd_mh_draw_first(mh), "mh_draw_first");
        } else {
            check(sepaihrd_mh_stage_normals(mh, z_next), "mh_stage_normals");
            z_next = sepaihrd_mh_staging_buffer(mh);
        }
        check(sepaihrd_mh_step(mh, nullptr, scale.data(), nullptr, nullptr, 0, 10.0 / (1 + 100.0), adapt_mode(1)), "mh_step");
    }
    if (device_streams) {
        // The run is queued ahead of the device; a report must not wait for it.  Behind the iteration it belongs to, a gather of
        // the reported chains' values and new samples is queued and copied home on a stream of its own
        // (sepaihrd_mh_snapshot_begin); a writer thread waits for THAT, formats the line and rewrites the checkpoint file.
        std::thread writer;
        std::string writer_error;
        auto queue_report = [&](int t) {
            if (writer.joinable()) writer.join();  // one snapshot in flight; reports are report_interval iterations apart
            if (!writer_error.empty()) throw ModelException("MetropolisHastingsSampler", writer_error);
            const int ns_now = store_samples_ ? sepaihrd_mh_sample_count(mh) : 0;
            const int first = rep_fetched, count = std::max(ns_now - rep_fetched, 0);
            check(sepaihrd_mh_snapshot_begin(mh, rep_chains.data(), reporter.k, first, count), "mh_snapshot_begin");
            rep_fetched = ns_now;
            writer = std::thread([&, t, count]() {
                const size_t k = static_cast<size_t>(reporter.k), n = static_cast<size_t>(count);
                std::vector<double> state(4 * k), smp(k * n * P), vals(k * n);
                if (sepaihrd_mh_snapshot_end(mh, 1, state.data(), n ? smp.data() : nullptr, n ? vals.data() : nullptr) != SEPAIHRD_OK) {
                    writer_error = "sepaihrd_mh_snapshot_end failed";
                    return;
                }
                for (size_t c = 0; c < k; ++c) {
                    if (n) reporter.append(static_cast<int>(c), &smp[c * n * P], &vals[c * n], n);
                    reporter.line(static_cast<int>(c), t, state[4 * c], state[4 * c + 1], static_cast<long>(state[4 * c + 3]), state[4 * c + 2]);
                }
                reporter.checkpoint();
            });
        };
        struct Joiner { std::thread& t; ~Joiner() { if (t.joinable()) t.join(); } } joiner{writer};
        for (int t = 1; t < iterations_; ++t) {
            check(sepaihrd_mh_step_tested(mh, 10.0 / ((t + 1) + 100.0), adapt_mode(t + 1), t + 1 < iterations_ ? 0 : 1), "mh_step_tested");
            if (reporter.due(t)) queue_report(t);
        }
        if (writer.joinable()) writer.join();
        if (!writer_error.empty()) throw ModelException("MetropolisHastingsSampler", writer_error);
        std::vector<double> lp_all(static_cast<size_t>(C)), best_all(static_cast<size_t>(C)), scale_all(static_cast<size_t>(C));
        std::vector<int32_t> acc_all(static_cast<size_t>(C)), emergency_all(static_cast<size_t>(C));
        check(sepaihrd_mh_read_run_state(mh, lp_all.data(), best_all.data(), scale_all.data(), acc_all.data(), emergency_all.data()), "mh_read_run_state");  // waits
        last_loop_seconds_ = secs(loop_begin, now());
        const int ns_dev = store_samples_ ? sepaihrd_mh_sample_count(mh) : 0;
        std::vector<double> sv;
        if (ns_dev > 0) {
            sv.resize(static_cast<size_t>(C) * ns_dev);
            check(sepaihrd_mh_read_sample_values(mh, 0, ns_dev, sv.data()), "mh_read_sample_values");
        }
        std::vector<uint8_t> tr;
        if (keep_traces_ && iterations_ > 1) {
            tr.resize(static_cast<size_t>(C) * (iterations_ - 1));
            check(sepaihrd_mh_read_accept_trace(mh, tr.data()), "mh_read_accept_trace");
        }
#pragma omp parallel for schedule(static) num_threads(nthreads)
        for (int c = 0; c < C; ++c) {
            Light& ch = chains[static_cast<size_t>(c)];
            ch.lp = lp_all[static_cast<size_t>(c)];
            ch.best = best_all[static_cast<size_t>(c)];
            ch.scale = scale_all[static_cast<size_t>(c)];
            ch.accepted = acc_all[static_cast<size_t>(c)];
            ch.emergency = emergency_all[static_cast<size_t>(c)];
            if (ns_dev > 0) ch.sample_values.assign(sv.begin() + static_cast<size_t>(c) * ns_dev, sv.begin() + static_cast<size_t>(c + 1) * ns_dev);
            if (!tr.empty()) {
                auto& out = traces_[static_cast<size_t>(c)];
                out.resize(static_cast<size_t>(iterations_ - 1));
                for (int t = 0; t + 1 < iterations_; ++t) out[static_cast<size_t>(t)] = tr[static_cast<size_t>(t) * C + c];
            }
        }
    } else {
    for (int t = 1; t < iterations_; ++t) {
        const auto p0 = now();
        const bool more = t + 1 < iterations_;
        if (t > 1) {  // the outcome of test t - 1 (the device ran it when evaluation t - 1 ended)
            check(sepaihrd_mh_fetch_test(mh, values.data(), flags.data()), "mh_fetch_test");
            const auto p0b = now();
            t_wait += secs(p0, p0b);
            book(t - 1, true);
            if (reporter.due(t - 1)) report_from_host(t - 1);
            t_book += secs(p0b, now());
        }
        const auto p1 = now();
        // ---- while the device evaluates proposal t: the inputs of test t and both continuations of proposal t + 1
#pragma omp parallel for schedule(static) num_threads(nthreads)
        for (int c = 0; c < C; ++c) {
            Light& ch = chains[static_cast<size_t>(c)];
            if (!device_streams) {
                // uniform_real_distribution<double>(0, 1) returns the canonical itself (c * (1 - 0) + 0)
                ch.log_u = std::log(ch.rng.at(0));                                     // :327
                t_log_u[c] = ch.log_u;
                ch.used_likely = 1 + (more ? draw_normals(ch.rng, 1, &z_next[static_cast<size_t>(c) * P]) : 0);
                if (more) ch.used_alt = draw_normals(ch.rng, 0, &z_plain[static_cast<size_t>(c) * P]);
            }
            if (adapt_scale_) {
                bool e;
                for (int a = 0; a < 2; ++a) {
                    ch.cand_log_scale[a] = next_log_scale(ch, a == 1, t, &e);
                    ch.cand_scale[a] = std::exp(ch.cand_log_scale[a]);
                }
                t_scale_reject[c] = ch.cand_scale[0];
                t_scale_accept[c] = ch.cand_scale[1];
            } else {
                t_scale_reject[c] = t_scale_accept[c] = ch.scale;
            }
        }
        if (more && !device_streams) {
            check(sepaihrd_mh_stage_normals(mh, z_next), "mh_stage_normals");
            z_next = sepaihrd_mh_staging_buffer(mh);
        }
        const auto p2 = now();
        // test t -> commit -> adapt -> proposal t + 1 -> evaluation t + 1: queued behind evaluation t
        check(sepaihrd_mh_step_tested(mh, 10.0 / ((t + 1) + 100.0), adapt_mode(t + 1), more ? 0 : 1), "mh_step_tested");
        const auto p3 = now();
        t_prepare += secs(p1, p2); t_launch += secs(p2, p3);
    }
    if (iterations_ > 1) {
        check(sepaihrd_mh_fetch_test(mh, values.data(), flags.data()), "mh_fetch_test");
        book(iterations_ - 1, false);
        if (reporter.due(iterations_ - 1)) report_from_host(iterations_ - 1);
    }
    last_loop_seconds_ = secs(loop_begin, now());
    }  // host-side streams
    {   // the best states were kept on the device
        std::vector<double> best_all(CP);
        check(sepaihrd_mh_read_best(mh, best_all.data()), "mh_read_best");
        for (int c = 0; c < C; ++c)
            chains[static_cast<size_t>(c)].best_x.assign(best_all.begin() + static_cast<size_t>(c) * P, best_all.begin() + static_cast<size_t>(c + 1) * P);
    }
    if (profile)
        std::fprintf(stderr, "[mh profile] per iteration ms: wait for the previous test %.3f  bookkeeping %.3f  prepare test and both continuations %.3f  upload + launches %.3f\n",
                     1e3 * t_wait / iterations_, 1e3 * t_book / iterations_, 1e3 * t_prepare / iterations_, 1e3 * t_launch / iterations_);

    std::vector<double> rows, covs(static_cast<size_t>(C) * PP);
    const int ns = store_samples_ ? sepaihrd_mh_sample_count(mh) : 0;  // state 0 and every thinning-th one (:266-268,357-360)
    summary_records_.clear();
    summary_width_ = 2 * P + 2;
    group_rows_.assign(1, C);
    if (ns > 0 && iterations_ > 1) {
        // per-chain summary over the samples after burn-in (all of them when none is): formed on the device, left in the
        // context's records buffer for a later all-gather, copied here
        int first = burn_in_ / thinning_ + 1;
        if (first >= ns) first = 0;
        double* d_rec = sepaihrd_records_buffer(ctx, 0, static_cast<size_t>(C) * summary_width_);
        if (!d_rec) throw ModelException("MetropolisHastingsSampler", std::string("sepaihrd_records_buffer: ") + sepaihrd_last_error(ctx));
        summary_records_.resize(static_cast<size_t>(C) * summary_width_);
        check(sepaihrd_mh_summary_records(mh, first, summary_records_.data(), d_rec), "mh_summary_records");
    }
    if (ns > 0) {
        rows.resize(static_cast<size_t>(C) * ns * P);
        check(sepaihrd_mh_read_samples(mh, 0, ns, rows.data()), "mh_read_samples");
    }
    check(sepaihrd_mh_read_covariance(mh, covs.data()), "mh_read_covariance");
    std::vector<OptimizationResult> results(static_cast<size_t>(C));
    for (int c = 0; c < C; ++c) {
        Light& ch = chains[static_cast<size_t>(c)];
        OptimizationResult& r = results[static_cast<size_t>(c)];
        r.bestParameters = Eigen::VectorXd(P);
        for (int i = 0; i < P; ++i) r.bestParameters[i] = ch.best_x[static_cast<size_t>(i)];
        r.bestObjectiveValue = ch.best;
        for (int s = 0; s < ns; ++s) {
            Eigen::VectorXd v(P);
            for (int i = 0; i < P; ++i) v[i] = rows[(static_cast<size_t>(c) * ns + s) * P + i];
            r.samples.push_back(v);
        }
        r.sampleObjectiveValues = ch.sample_values;
        r.finalCovariance = Eigen::MatrixXd(P, P);
        for (int i = 0; i < P; ++i)
            for (int j = 0; j < P; ++j) r.finalCovariance(i, j) = covs[static_cast<size_t>(c) * PP + static_cast<size_t>(i) * P + j];
        r.additionalStats["acceptance_rate"] = static_cast<double>(ch.accepted) / iterations_;
        r.additionalStats["accepted_count"] = ch.accepted;
        r.additionalStats["final_scale"] = ch.scale;
        r.additionalStats["burn_in"] = static_cast<double>(burn_in_);
        r.additionalStats["total_iterations"] = static_cast<double>(iterations_);
    }
    {   // evaluations the accept tests saw FAIL (they counted as -1e18, like an objective that throws)
        int64_t fc[3] = {0, 0, 0};
        check(sepaihrd_mh_read_failure_counts(mh, fc), "mh_read_failure_counts");
        failure_counts_.assign({static_cast<long>(fc[0]), static_cast<long>(fc[1]), static_cast<long>(fc[2])});
        for (OptimizationResult& r : results) r.additionalStats["failed_evaluations_all_chains"] = static_cast<double>(fc[0] + fc[1] + fc[2]);
        if (fc[0] + fc[1] > 0)
            reporter.say("WARNING", std::to_string(fc[0]) + " evaluations ended in odeint's 500 consecutive step rejections and " + std::to_string(fc[1]) +
                                        " in the attempt budget; their proposals were rejected like an objective that throws");
        if (fc[2] > 0)
            throw ModelException("MetropolisHastingsSampler", std::to_string(fc[2]) + " evaluations reported SEPAIHRD_STATUS_PIPELINE (the hand-off between an "
                                 "integrating wave and its likelihood wave timed out): a defect of the evaluation kernel, not of the model -- the run's accept decisions are not to be trusted");
    }
    // the trace files of the reported chains, from everything the run stored (:399-411)
    for (int c = 0; c < reporter.k && ns > 0; ++c) {
        reporter.samples[static_cast<size_t>(c)].assign(rows.begin() + static_cast<size_t>(c) * ns * P, rows.begin() + static_cast<size_t>(c + 1) * ns * P);
        reporter.values[static_cast<size_t>(c)] = chains[static_cast<size_t>(c)].sample_values;
    }
    reporter.finish();
    return results;
}

// Test hook (tests/test_host_logic.py, no GPU needed): the draws of `rounds` consecutive accept tests of one chain through
// the look-ahead queue exactly as optimizeChainsOnDevice makes them -- log(u) of the element in front, the P normals of
// the continuation that takes the uniform (takes_uniform[r] != 0) or of the one that does not, then the stream moves by
// what that continuation used.  Both continuations are drawn first, as in the loop.
void canonical_queue_draw_sequence(uint32_t seed, int P, int rounds, const unsigned char* takes_uniform, double* normals,
                                   double* log_u) {
    CanonicalQueue q;
    q.gen.seed(seed);
    std::vector<double> with_u(static_cast<size_t>(P)), plain(static_cast<size_t>(P));
    for (int r = 0; r < rounds; ++r) {
        log_u[r] = std::log(q.at(0));
        const size_t used_with = 1 + standard_normals_from_queue(q, 1, with_u.data(), P);
        const size_t used_plain = standard_normals_from_queue(q, 0, plain.data(), P);
        const bool u = takes_uniform[r] != 0;
        std::copy((u ? with_u : plain).begin(), (u ? with_u : plain).end(), normals + static_cast<size_t>(r) * P);
        q.consume(u ? used_with : used_plain);
    }
}

std::vector<OptimizationResult> MultiChainMetropolisHastings::optimizeChainGroupsOnDevice(
    const std::vector<double>& initial, int C, const std::vector<HipSEPAIHRDObjectiveFunction*>& objectives,
    IParameterManager& pm) {
    const int G = static_cast<int>(objectives.size());
    if (G <= 0) throw InvalidParameterException("MetropolisHastingsSampler", "no objective given");
    const int P = static_cast<int>(pm.getParameterCount());
    if (static_cast<int>(initial.size()) != C * P) throw InvalidParameterException("MetropolisHastingsSampler", "initial size != C*P");
    if (G == 1 || C < G) return optimizeChainsOnDevice(initial, C, *objectives[0], pm);
    if (auto* spm = dynamic_cast<HipSEPAIHRDParameterManager*>(&pm)) spm->setConstraintMode(ConstraintMode::MCMC_REFLECT);
    std::vector<int> first(static_cast<size_t>(G) + 1, 0);
    for (int g = 0; g < G; ++g) first[static_cast<size_t>(g) + 1] = first[static_cast<size_t>(g)] + C / G + (g < C % G ? 1 : 0);
    const int share = host_threads_ > 0 ? std::min(host_threads_, host_thread_share()) : host_thread_share();
    std::vector<std::vector<OptimizationResult>> parts(static_cast<size_t>(G));
    std::vector<std::vector<std::vector<unsigned char>>> part_traces(static_cast<size_t>(G));
    std::vector<std::string> errors(static_cast<size_t>(G));
    std::vector<double> loop_secs(static_cast<size_t>(G), 0.0);
    std::vector<std::vector<double>> part_records(static_cast<size_t>(G));
    std::vector<int> fell_back(static_cast<size_t>(G), 0);
    std::vector<std::vector<long>> part_failures(static_cast<size_t>(G));
    std::vector<std::thread> workers;
    for (int g = 0; g < G; ++g) {
        workers.emplace_back([&, g]() {
            try {
                MultiChainMetropolisHastings local = *this;  // settings + initial covariance
                local.seed_ = seed_ + static_cast<uint32_t>(first[static_cast<size_t>(g)]);
                local.host_threads_ = std::max(1, share / G);
                if (g > 0) local.checkpoint_chains_ = 0;  // the reports and trace files cover the run's first chains: group 0's
                const int c0 = first[static_cast<size_t>(g)], c1 = first[static_cast<size_t>(g) + 1];
                const std::vector<double> init(initial.begin() + static_cast<size_t>(c0) * P, initial.begin() + static_cast<size_t>(c1) * P);
                parts[static_cast<size_t>(g)] = local.optimizeChainsOnDevice(init, c1 - c0, *objectives[static_cast<size_t>(g)], pm);
                part_traces[static_cast<size_t>(g)] = local.traces_;
                part_records[static_cast<size_t>(g)] = local.summary_records_;
                loop_secs[static_cast<size_t>(g)] = local.last_loop_seconds_;
                fell_back[static_cast<size_t>(g)] = local.device_streams_fell_back_ ? 1 : 0;
                part_failures[static_cast<size_t>(g)] = local.failure_counts_;
            } catch (const std::exception& e) { errors[static_cast<size_t>(g)] = e.what(); }
        });
    }
    for (std::thread& w : workers) w.join();
    for (const std::string& e : errors)
        if (!e.empty()) throw ModelException("MetropolisHastingsSampler", e);
    std::vector<OptimizationResult> results;
    traces_.clear();
    last_loop_seconds_ = *std::max_element(loop_secs.begin(), loop_secs.end());
    device_streams_fell_back_ = std::any_of(fell_back.begin(), fell_back.end(), [](int v) { return v != 0; });
    failure_counts_.assign(3, 0);
    for (const std::vector<long>& f : part_failures)
        for (size_t k = 0; k < f.size() && k < 3; ++k) failure_counts_[k] += f[k];
    summary_records_.clear();
    summary_width_ = 2 * P + 2;
    group_rows_.clear();
    for (int g = 0; g < G; ++g) {
        for (OptimizationResult& r : parts[static_cast<size_t>(g)]) results.push_back(std::move(r));
        for (auto& t : part_traces[static_cast<size_t>(g)]) traces_.push_back(std::move(t));
        summary_records_.insert(summary_records_.end(), part_records[static_cast<size_t>(g)].begin(), part_records[static_cast<size_t>(g)].end());
        group_rows_.push_back(first[static_cast<size_t>(g) + 1] - first[static_cast<size_t>(g)]);
    }
    return results;
}

int MultiChainMetropolisHastings::gatherChainSummaries(const std::vector<HipSEPAIHRDObjectiveFunction*>& objectives, int backend) {
    const int G = static_cast<int>(objectives.size());
    if (G <= 0 || static_cast<size_t>(G) != group_rows_.size() || summary_records_.empty())
        throw InvalidParameterException("MetropolisHastingsSampler", "gatherChainSummaries: run optimizeChainGroupsOnDevice over these objectives first (with stored samples)");
    std::vector<sepaihrd_ctx*> ctxs;
    std::vector<int32_t> rows;
    for (int g = 0; g < G; ++g) {
        ctxs.push_back(objectives[static_cast<size_t>(g)]->deviceContext());
        rows.push_back(group_rows_[static_cast<size_t>(g)]);
    }
    int used = 0;
    if (sepaihrd_allgather_records(ctxs.data(), G, rows.data(), summary_width_, backend, &used) != SEPAIHRD_OK)
        throw ModelException("MetropolisHastingsSampler", std::string("sepaihrd_allgather_records: ") + sepaihrd_last_error(ctxs[0]));
    return used;
}

std::vector<double> MultiChainMetropolisHastings::gatheredSummaries(HipSEPAIHRDObjectiveFunction& objective) const {
    std::vector<double> out(summary_records_.size());
    if (sepaihrd_read_records(objective.deviceContext(), 1, out.data(), out.size()) != SEPAIHRD_OK)
        throw ModelException("MetropolisHastingsSampler", std::string("sepaihrd_read_records: ") + sepaihrd_last_error(objective.deviceContext()));
    return out;
}

std::vector<double> MultiChainMetropolisHastings::summaryQuantiles(const std::vector<double>& table, int width, const std::vector<double>& probs) {
    if (width <= 0 || table.size() % static_cast<size_t>(width) != 0)
        throw InvalidParameterException("MetropolisHastingsSampler", "summaryQuantiles: table size is not a multiple of the record width");
    const size_t n = table.size() / static_cast<size_t>(width);
    std::vector<double> out(probs.size() * static_cast<size_t>(width), std::numeric_limits<double>::quiet_NaN());
    if (n == 0) return out;
    std::vector<double> col(n);
    for (int w = 0; w < width; ++w) {
        for (size_t c = 0; c < n; ++c) col[c] = table[c * width + w];
        std::sort(col.begin(), col.end());
        for (size_t q = 0; q < probs.size(); ++q) {
            const double pos = probs[q] * static_cast<double>(n - 1);
            const size_t lo = static_cast<size_t>(std::floor(pos));
            const size_t hi = std::min(lo + 1, n - 1);
            const double f = pos - static_cast<double>(lo);
            out[q * width + w] = col[lo] * (1.0 - f) + col[hi] * f;
        }
    }
    return out;
}

}  // namespace epidemic
