// host/src/BatchedHillClimbing.cpp -- see the header for the reference lines mirrored.
#include "epidemic_hip/BatchedHillClimbing.hpp"

#include <algorithm>
#include <cmath>
#include <random>

namespace epidemic {

namespace {

// Eigen::LLT restated (lower factor, row-major P x P); false = not positive definite
bool llt_lower(const std::vector<double>& A, int P, std::vector<double>& L) {
    L.assign(static_cast<size_t>(P) * P, 0.0);
    for (int j = 0; j < P; ++j) {
        double d = A[static_cast<size_t>(j) * P + j];
        for (int k = 0; k < j; ++k) d -= L[static_cast<size_t>(j) * P + k] * L[static_cast<size_t>(j) * P + k];
        if (!(d > 0.0)) return false;
        const double ljj = std::sqrt(d);
        L[static_cast<size_t>(j) * P + j] = ljj;
        for (int i = j + 1; i < P; ++i) {
            double s = A[static_cast<size_t>(i) * P + j];
            for (int k = 0; k < j; ++k) s -= L[static_cast<size_t>(i) * P + k] * L[static_cast<size_t>(j) * P + k];
            L[static_cast<size_t>(i) * P + j] = s / ljj;
        }
    }
    return true;
}

inline double sanitize(double v) { return (std::isnan(v) || std::isinf(v)) ? -1e18 : v; }  // safe_evaluate :24-32

}  // namespace

void BatchedHillClimbingOptimizer::configure(const std::map<std::string, double>& settings) {
    auto get = [&](const std::string& key, double def) {
        auto it = settings.find(key);
        return (it != settings.end()) ? it->second : def;
    };
    iterations_ = static_cast<int>(get("iterations", 2000.0));
    report_interval_ = static_cast<int>(get("report_interval", 100.0));
    cloud_size_multiplier_ = std::max(1, static_cast<int>(get("cloud_size_multiplier", 8.0)));
    threads_ = std::max(1, static_cast<int>(get("threads", 16.0)));
    seed_ = static_cast<uint32_t>(get("seed", 1.0));
}

OptimizationResult BatchedHillClimbingOptimizer::optimize(const Eigen::VectorXd& x0, IObjectiveFunction& objective,
                                                          IParameterManager& pm) {
    const int P = static_cast<int>(x0.size());
    auto* batch = dynamic_cast<IBatchObjectiveFunction*>(&objective);
    std::vector<int> status;
    BatchEval eval;
    if (batch) {
        eval = [&, batch](const double* th, int B, double* out) {
            status.resize(static_cast<size_t>(B));
            batch->calculateBatch(th, B, out, status.data());
            for (int b = 0; b < B; ++b)  // an integration failure is the exception safe_evaluate swallows
                out[b] = status[static_cast<size_t>(b)] >= 2 ? -1e18 : sanitize(out[b]);
        };
    } else {
        eval = [&, P](const double* th, int B, double* out) {
            for (int b = 0; b < B; ++b) {
                Eigen::VectorXd v(P);
                for (int i = 0; i < P; ++i) v[i] = th[static_cast<size_t>(b) * P + i];
                try { out[b] = sanitize(objective.calculate(v)); } catch (...) { out[b] = -1e18; }
            }
        };
    }
    return run(x0, eval, pm);
}

OptimizationResult BatchedHillClimbingOptimizer::run(const Eigen::VectorXd& x0, const BatchEval& eval_raw,
                                                     IParameterManager& pm) {
    const int P = static_cast<int>(x0.size());
    const size_t PP = static_cast<size_t>(P) * P;
    trace_.clear();
    evaluations_ = launches_ = 0;
    auto eval = [&](const std::vector<double>& thetas, int B, std::vector<double>& out) {
        out.resize(static_cast<size_t>(B));
        if (B == 0) return;
        eval_raw(thetas.data(), B, out.data());
        evaluations_ += B;
        ++launches_;
    };
    auto constrain_into = [&](const std::vector<double>& p, double* dst) {
        Eigen::VectorXd v(P);
        for (int i = 0; i < P; ++i) v[i] = p[static_cast<size_t>(i)];
        const Eigen::VectorXd c = pm.applyConstraints(v);
        for (int i = 0; i < P; ++i) dst[i] = c[i];
    };

    OptimizationResult result;
    std::vector<double> cur(x0.data(), x0.data() + P), prev = cur, best = cur;
    std::vector<double> one;
    eval(cur, 1, one);  // :136
    double cur_l = one[0], best_l = one[0];

    std::vector<double> cov(PP, 0.0), L;
    for (int i = 0; i < P; ++i) {  // :146-150
        const double s = pm.getSigmaForParamIndex(i);
        cov[static_cast<size_t>(i) * P + i] = (s > 0 ? s * s : 1e-4);
    }
    llt_lower(cov, P, L);  // :153

    const int V = threads_;
    const int nc = std::max(4, V * cloud_size_multiplier_);  // :163
    std::mt19937 master(seed_);
    std::vector<std::mt19937> rngs(static_cast<size_t>(V));
    std::vector<std::normal_distribution<double>> norms(static_cast<size_t>(V));
    for (int t = 0; t < V; ++t) {  // :179-183
        const uint32_t a = static_cast<uint32_t>(master()), b = static_cast<uint32_t>(master()),
                       c = static_cast<uint32_t>(master()), d = static_cast<uint32_t>(master());
        std::seed_seq sq{a, b, c, d};
        rngs[static_cast<size_t>(t)].seed(sq);
        norms[static_cast<size_t>(t)] = std::normal_distribution<double>(0.0, 1.0);
    }
    std::vector<int> owner(static_cast<size_t>(nc));  // schedule(static): contiguous blocks
    {
        const int q = nc / V, rem = nc % V;
        size_t i = 0;
        for (int t = 0; t < V; ++t)
            for (int k = 0; k < q + (t < rem ? 1 : 0); ++k) owner[i++] = t;
    }

    std::vector<double> steps(static_cast<size_t>(nc) * P), cloud(static_cast<size_t>(nc) * P), scores, z(static_cast<size_t>(P));
    std::vector<double> tmp(static_cast<size_t>(P)), ls_batch, ls_val;

    // performRobustLineSearch (:39-112) with each phase's calls issued as one batch
    auto line_search = [&](std::vector<double>& params, double& logL, const std::vector<double>& direction) {
        const double shrinkage = 0.5, growth = 2.0;
        const int max_backtrack = 10, max_expansion = 12;
        // phase 1: the candidates of the backtracking loop; it stops BEFORE evaluating the first
        // candidate closer than 1e-8 to the current point
        ls_batch.assign(static_cast<size_t>(max_backtrack) * P, 0.0);
        int nb = 0;
        double step = 1.0;
        for (int i = 0; i < max_backtrack; ++i) {
            for (int k = 0; k < P; ++k) tmp[static_cast<size_t>(k)] = params[static_cast<size_t>(k)] + direction[static_cast<size_t>(k)] * step;
            double* cand = &ls_batch[static_cast<size_t>(nb) * P];
            constrain_into(tmp, cand);
            double sq = 0.0;
            for (int k = 0; k < P; ++k) sq += (cand[k] - params[static_cast<size_t>(k)]) * (cand[k] - params[static_cast<size_t>(k)]);
            if (sq < 1e-16) break;
            ++nb;
            step *= shrinkage;
        }
        eval(ls_batch, nb, ls_val);
        int hit = -1;
        for (int i = 0; i < nb; ++i)
            if (ls_val[static_cast<size_t>(i)] > logL) { hit = i; break; }
        if (hit < 0) return false;
        std::vector<double> improved(ls_batch.begin() + static_cast<size_t>(hit) * P, ls_batch.begin() + static_cast<size_t>(hit + 1) * P);
        const double improved_l = ls_val[static_cast<size_t>(hit)];
        // phase 2: candidate i = constrain(candidate i-1 + step 2^i), anchor = previous candidate
        std::vector<double> cur_step(static_cast<size_t>(P)), anchor = improved;
        for (int k = 0; k < P; ++k) cur_step[static_cast<size_t>(k)] = improved[static_cast<size_t>(k)] - params[static_cast<size_t>(k)];
        ls_batch.assign(static_cast<size_t>(max_expansion) * P, 0.0);
        for (int i = 0; i < max_expansion; ++i) {
            for (int k = 0; k < P; ++k) {
                cur_step[static_cast<size_t>(k)] *= growth;
                tmp[static_cast<size_t>(k)] = anchor[static_cast<size_t>(k)] + cur_step[static_cast<size_t>(k)];
            }
            constrain_into(tmp, &ls_batch[static_cast<size_t>(i) * P]);
            anchor.assign(ls_batch.begin() + static_cast<size_t>(i) * P, ls_batch.begin() + static_cast<size_t>(i + 1) * P);
        }
        eval(ls_batch, max_expansion, ls_val);
        std::vector<double> bestp = improved;
        double bestl = improved_l;
        for (int i = 0; i < max_expansion; ++i) {
            if (ls_val[static_cast<size_t>(i)] > bestl) {
                bestl = ls_val[static_cast<size_t>(i)];
                bestp.assign(ls_batch.begin() + static_cast<size_t>(i) * P, ls_batch.begin() + static_cast<size_t>(i + 1) * P);
            } else {
                break;
            }
        }
        params = bestp;
        logL = bestl;
        return true;
    };

    for (int iter = 0; iter < iterations_; ++iter) {
        for (int i = 0; i < nc; ++i) {  // :192-214
            std::mt19937& g = rngs[static_cast<size_t>(owner[static_cast<size_t>(i)])];
            std::normal_distribution<double>& nd = norms[static_cast<size_t>(owner[static_cast<size_t>(i)])];
            double* st = &steps[static_cast<size_t>(i) * P];
            if (i < nc / 2) {
                for (int k = 0; k < P; ++k) z[static_cast<size_t>(k)] = nd(g);
                for (int a = 0; a < P; ++a) {
                    double sum = 0.0;
                    for (int b = 0; b <= a; ++b) sum += L[static_cast<size_t>(a) * P + b] * z[static_cast<size_t>(b)];
                    st[a] = sum;
                }
            } else {
                std::uniform_int_distribution<int> param_dist(0, P - 1);
                const int idx = param_dist(g);
                const double sg = std::sqrt(cov[static_cast<size_t>(idx) * P + idx]);
                std::fill(st, st + P, 0.0);
                st[idx] = sg * nd(g);
            }
        }
        for (int i = 0; i < nc; ++i) {  // :222-228, one launch
            for (int k = 0; k < P; ++k) tmp[static_cast<size_t>(k)] = cur[static_cast<size_t>(k)] + steps[static_cast<size_t>(i) * P + k];
            constrain_into(tmp, &cloud[static_cast<size_t>(i) * P]);
        }
        eval(cloud, nc, scores);
        int best_idx = -1;
        double best_val = -1e18;
        for (int i = 0; i < nc; ++i)
            if (scores[static_cast<size_t>(i)] > best_val) { best_val = scores[static_cast<size_t>(i)]; best_idx = i; }
        bool moved = false;
        if (best_idx != -1 && best_val > -1e18) {  // :241-262
            const std::vector<double> point(cloud.begin() + static_cast<size_t>(best_idx) * P,
                                            cloud.begin() + static_cast<size_t>(best_idx + 1) * P);
            std::vector<double> dir(static_cast<size_t>(P));
            for (int k = 0; k < P; ++k) dir[static_cast<size_t>(k)] = point[static_cast<size_t>(k)] - cur[static_cast<size_t>(k)];
            if (best_val > cur_l) { cur = point; cur_l = best_val; moved = true; }
            const bool ls = line_search(cur, cur_l, dir);
            moved = moved || ls;
        }
        if (moved) {  // :265-308
            if (cur_l > best_l) { best_l = cur_l; best = cur; }
            std::vector<double> st(static_cast<size_t>(P));
            double norm2 = 0.0;
            for (int k = 0; k < P; ++k) {
                st[static_cast<size_t>(k)] = cur[static_cast<size_t>(k)] - prev[static_cast<size_t>(k)];
                norm2 += st[static_cast<size_t>(k)] * st[static_cast<size_t>(k)];
            }
            if (norm2 > 1e-14) {
                const double alpha = 2.0 / (P + 2.0);
                for (int a = 0; a < P; ++a)
                    for (int b = 0; b < P; ++b) {
                        double v = cov[static_cast<size_t>(a) * P + b] * (1.0 - alpha);
                        v += alpha * (st[static_cast<size_t>(a)] * st[static_cast<size_t>(b)]);
                        cov[static_cast<size_t>(a) * P + b] = v;
                    }
                std::vector<double> sym(PP);
                for (int a = 0; a < P; ++a)
                    for (int b = 0; b < P; ++b)
                        sym[static_cast<size_t>(a) * P + b] = 0.5 * (cov[static_cast<size_t>(a) * P + b] + cov[static_cast<size_t>(b) * P + a]);
                cov.swap(sym);
                double tr = 0.0;
                for (int a = 0; a < P; ++a) tr += cov[static_cast<size_t>(a) * P + a];
                const double jitter = 1e-8 * tr / P;
                for (int a = 0; a < P; ++a) cov[static_cast<size_t>(a) * P + a] += jitter;
                for (int a = 0; a < P; ++a) {
                    double mv = pm.getSigmaForParamIndex(a);
                    mv = (mv > 0 ? mv * mv * 0.01 : 1e-8);
                    if (cov[static_cast<size_t>(a) * P + a] < mv) cov[static_cast<size_t>(a) * P + a] = mv;
                }
            }
            prev = cur;
        }
        if (iter > 0 && iter % 10 == 0) {  // :313-341
            std::vector<double> Ln;
            if (llt_lower(cov, P, Ln)) {
                L.swap(Ln);
            } else {
                double tr = 0.0;
                for (int a = 0; a < P; ++a) tr += cov[static_cast<size_t>(a) * P + a];
                double lambda = 1e-6 * tr / P;
                bool regularized = false;
                for (int attempt = 0; attempt < 5; ++attempt) {
                    for (int a = 0; a < P; ++a) cov[static_cast<size_t>(a) * P + a] += lambda;
                    if (llt_lower(cov, P, Ln)) { L.swap(Ln); regularized = true; break; }
                    lambda *= 10.0;
                }
                if (!regularized) {
                    L.assign(PP, 0.0);
                    for (int a = 0; a < P; ++a) L[static_cast<size_t>(a) * P + a] = std::sqrt(cov[static_cast<size_t>(a) * P + a]);
                    for (int a = 0; a < P; ++a)
                        for (int b = 0; b < P; ++b)
                            if (a != b) cov[static_cast<size_t>(a) * P + b] = 0.0;
                }
            }
        }
        trace_.push_back(cur_l);
    }

    result.bestParameters = Eigen::VectorXd(P);
    for (int i = 0; i < P; ++i) result.bestParameters[i] = best[static_cast<size_t>(i)];
    result.bestObjectiveValue = best_l;
    result.finalCovariance = Eigen::MatrixXd(P, P);  // :346, transferred to the sampler by ModelCalibrator.cpp:93-131
    for (int a = 0; a < P; ++a)
        for (int b = 0; b < P; ++b) result.finalCovariance(a, b) = cov[static_cast<size_t>(a) * P + b];
    result.additionalStats["objective_evaluations"] = static_cast<double>(evaluations_);
    result.additionalStats["device_launches"] = static_cast<double>(launches_);
    return result;
}

}  // namespace epidemic
