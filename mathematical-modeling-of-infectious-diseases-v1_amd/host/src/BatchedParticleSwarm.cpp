// host/src/BatchedParticleSwarm.cpp -- see the header for the reference lines mirrored.
#include "epidemic_hip/BatchedParticleSwarm.hpp"

#include <algorithm>
#include <cmath>
#include <functional>
#include <limits>
#include <stdexcept>
#include <string>

namespace epidemic {

namespace {
inline double clampd(double v, double lo, double hi) { return std::clamp(v, lo, hi); }
}  // namespace

// Settings arrive as one key -> double map (pso_settings.txt).  Each known key is a row of the table below: where
// the value goes and the range it must lie in -- the same keys and the same accept / reject decisions as
// ParticleSwarmOptimization::configure (ParticleSwarmOptimizer.cpp:10-95); unknown keys are ignored like there
// (use_parallel and log_evolutionary_state have nothing to switch on this path).
void BatchedParticleSwarmOptimization::configure(const std::map<std::string, double>& settings) {
    enum Range { ANY, POSITIVE, NON_NEGATIVE, ENUM_0_4, ENUM_0_3 };
    struct Setting {
        const char* key;
        Range range;
        std::function<void(double)> store;
    };
    auto as_int = [](int& dst) { return [&dst](double v) { dst = static_cast<int>(v); }; };
    auto as_double = [](double& dst) { return [&dst](double v) { dst = v; }; };
    auto as_flag = [](bool& dst) { return [&dst](double v) { dst = v != 0.0; }; };
    const Setting table[] = {
        {"iterations", POSITIVE, as_int(iterations_)},
        {"swarm_size", POSITIVE, as_int(swarm_size_)},
        {"report_interval", POSITIVE, as_int(report_interval_)},
        {"max_stagnation", POSITIVE, as_int(max_stagnation_)},
        {"omega_start", NON_NEGATIVE, as_double(omega_start_)},
        {"omega_end", NON_NEGATIVE, as_double(omega_end_)},
        {"c1_initial", NON_NEGATIVE, as_double(c1_initial_)},
        {"c1_final", NON_NEGATIVE, as_double(c1_final_)},
        {"c2_initial", NON_NEGATIVE, as_double(c2_initial_)},
        {"c2_final", NON_NEGATIVE, as_double(c2_final_)},
        {"variant", ENUM_0_4, [this](double v) { variant_ = static_cast<PSOVariant>(static_cast<int>(v)); }},
        {"topology", ENUM_0_3, [this](double v) { topology_ = static_cast<TopologyType>(static_cast<int>(v)); }},
        {"use_opposition_learning", ANY, as_flag(use_opposition_learning_)},
        {"use_adaptive_parameters", ANY, as_flag(use_adaptive_parameters_)},
        {"diversity_threshold", ANY, as_double(diversity_threshold_)},
        {"restart_threshold", ANY, as_double(restart_threshold_)},
        {"quantum_beta", ANY, as_double(quantum_beta_)},
        {"levy_alpha", ANY, as_double(levy_alpha_)},
        {"seed", NON_NEGATIVE, [this](double v) { seed_ = static_cast<uint32_t>(std::fmod(v, 4294967296.0)); }},  // build-side: replaces std::random_device (:578)
    };
    for (const Setting& row : table) {
        const auto it = settings.find(row.key);
        if (it == settings.end()) continue;
        const double v = it->second;
        // the enum ranges are checked on the double itself: a float-to-int conversion of a value the int cannot hold
        // (a seed of 3e9 in an ANY row) is undefined behaviour
        const bool ok = row.range == ANY || (row.range == POSITIVE && v > 0) || (row.range == NON_NEGATIVE && v >= 0) ||
                        (row.range == ENUM_0_4 && v >= 0.0 && v < 5.0) || (row.range == ENUM_0_3 && v >= 0.0 && v < 4.0);
        if (!ok) {
            static const char* what[] = {"", "a positive number", "zero or a positive number", "one of 0..4", "one of 0..3"};
            throw std::invalid_argument(std::string("particle swarm setting '") + row.key + "' = " + std::to_string(v) +
                                        ": expected " + what[row.range]);
        }
        row.store(v);
    }
}

void BatchedParticleSwarmOptimization::evaluate(const std::vector<const std::vector<double>*>& points,
                                                std::vector<double>& values) {
    const int B = static_cast<int>(points.size());
    values.resize(static_cast<size_t>(B));
    if (B == 0) return;
    std::vector<double> flat(static_cast<size_t>(B) * n_);
    for (int b = 0; b < B; ++b) std::copy(points[static_cast<size_t>(b)]->begin(), points[static_cast<size_t>(b)]->end(), flat.begin() + static_cast<size_t>(b) * n_);
    eval_(flat.data(), B, values.data());
    evaluations_ += B;
    ++launches_;
}

OptimizationResult BatchedParticleSwarmOptimization::optimize(const Eigen::VectorXd& initialParameters,
                                                              IObjectiveFunction& objective, IParameterManager& pm) {
    n_ = static_cast<int>(pm.getParameterCount());
    lb_.resize(static_cast<size_t>(n_));
    ub_.resize(static_cast<size_t>(n_));
    for (int k = 0; k < n_; ++k) {
        lb_[static_cast<size_t>(k)] = pm.getLowerBoundForParamIndex(k);
        ub_[static_cast<size_t>(k)] = pm.getUpperBoundForParamIndex(k);
    }
    auto* batch = dynamic_cast<IBatchObjectiveFunction*>(&objective);
    std::vector<int> status;
    if (batch) {
        eval_ = [&, batch](const double* th, int B, double* out) {
            status.resize(static_cast<size_t>(B));
            batch->calculateBatch(th, B, out, status.data());
            for (int b = 0; b < B; ++b)  // calculate() lets SimulationException escape and the swarm loop does not catch it
                if (status[static_cast<size_t>(b)] >= 2)
                    throw SimulationException("Dopri5SolverStrategy::integrate", "Boost.Odeint integration failed: step size adjustment");
        };
    } else {
        eval_ = [&](const double* th, int B, double* out) {
            for (int b = 0; b < B; ++b) {
                Eigen::VectorXd v(n_);
                for (int i = 0; i < n_; ++i) v[i] = th[static_cast<size_t>(b) * n_ + i];
                out[b] = objective.calculate(v);
            }
        };
    }
    eval_one_ = [&](const std::vector<double>& p) {
        std::vector<double> v;
        evaluate({&p}, v);
        return v[0];
    };
    rng_.seed(seed_);
    uniform_dist_.reset();
    normal_dist_.reset();
    stagnation_counter_ = 0;
    trace_.clear();
    evaluations_ = launches_ = 0;

    std::vector<double> init;
    if (initialParameters.size() == n_) init.assign(initialParameters.data(), initialParameters.data() + n_);
    initializeSwarm(init.empty() ? nullptr : &init);

    double previous_gbest = -std::numeric_limits<double>::infinity();
    for (int iter = 0; iter < iterations_; ++iter) {
        if (std::abs(gbest_value_ - previous_gbest) < restart_threshold_) {  // :129-141
            stagnation_counter_++;
            if (stagnation_counter_ > max_stagnation_) {
                restartSwarm();
                stagnation_counter_ = 0;
            }
        } else {
            stagnation_counter_ = 0;
        }
        previous_gbest = gbest_value_;
        updateParticles(iter);
        for (const Particle& p : swarm_)  // :148-155
            if (p.pbest_value > gbest_value_) {
                gbest_value_ = p.pbest_value;
                gbest_position_ = p.pbest_position;
            }
        if ((variant_ == PSOVariant::ADAPTIVE || variant_ == PSOVariant::HYBRID) && (iter % 5 == 0)) {  // :158-179
            auto best_it = std::max_element(swarm_.begin(), swarm_.end(),
                                            [](const Particle& a, const Particle& b) { return a.pbest_value < b.pbest_value; });
            applyElitistLearningStrategy(*best_it);
            if (best_it->pbest_value > gbest_value_) {
                gbest_value_ = best_it->pbest_value;
                gbest_position_ = best_it->pbest_position;
            }
        }
        trace_.push_back(gbest_value_);
    }

    OptimizationResult result;
    result.bestParameters = Eigen::VectorXd(n_);
    for (int i = 0; i < n_; ++i) result.bestParameters[i] = gbest_position_[static_cast<size_t>(i)];
    result.bestObjectiveValue = gbest_value_;
    // covariance of the personal bests for the sampler (:221-239)
    std::vector<double> mean(static_cast<size_t>(n_), 0.0);
    for (const Particle& p : swarm_)
        for (int k = 0; k < n_; ++k) mean[static_cast<size_t>(k)] += p.pbest_position[static_cast<size_t>(k)];
    for (double& m : mean) m /= swarm_size_;
    result.finalCovariance = Eigen::MatrixXd::Zero(n_, n_);
    for (const Particle& p : swarm_)
        for (int a = 0; a < n_; ++a)
            for (int b = 0; b < n_; ++b)
                result.finalCovariance(a, b) += (p.pbest_position[static_cast<size_t>(a)] - mean[static_cast<size_t>(a)]) *
                                                (p.pbest_position[static_cast<size_t>(b)] - mean[static_cast<size_t>(b)]);
    for (int a = 0; a < n_; ++a)
        for (int b = 0; b < n_; ++b) result.finalCovariance(a, b) /= (swarm_size_ - 1);
    for (int a = 0; a < n_; ++a) result.finalCovariance(a, a) += 1e-6;
    result.additionalStats["objective_evaluations"] = static_cast<double>(evaluations_);
    result.additionalStats["device_launches"] = static_cast<double>(launches_);
    return result;
}

void BatchedParticleSwarmOptimization::initializeSwarm(const std::vector<double>* initial) {  // :249-328
    swarm_.assign(static_cast<size_t>(swarm_size_), Particle());
    gbest_value_ = -std::numeric_limits<double>::infinity();
    gbest_position_.assign(static_cast<size_t>(n_), 0.0);
    std::vector<unsigned int> seeds(static_cast<size_t>(swarm_size_));
    for (int i = 0; i < swarm_size_; ++i) seeds[static_cast<size_t>(i)] = static_cast<unsigned int>(rng_());
    for (int i = 0; i < swarm_size_; ++i) {
        Particle& p = swarm_[static_cast<size_t>(i)];
        p.position.resize(static_cast<size_t>(n_));
        p.velocity.resize(static_cast<size_t>(n_));
        std::mt19937 local_rng(seeds[static_cast<size_t>(i)]);
        std::uniform_real_distribution<> local_uniform(0.0, 1.0);
        if (i == 0 && initial != nullptr) {
            for (int k = 0; k < n_; ++k)
                p.position[static_cast<size_t>(k)] = clampd((*initial)[static_cast<size_t>(k)], lb_[static_cast<size_t>(k)], ub_[static_cast<size_t>(k)]);
        } else {
            for (int k = 0; k < n_; ++k)
                p.position[static_cast<size_t>(k)] = lb_[static_cast<size_t>(k)] + local_uniform(local_rng) * (ub_[static_cast<size_t>(k)] - lb_[static_cast<size_t>(k)]);
        }
        for (int k = 0; k < n_; ++k) {
            const double vmax = 0.2 * (ub_[static_cast<size_t>(k)] - lb_[static_cast<size_t>(k)]);
            p.velocity[static_cast<size_t>(k)] = -vmax + 2 * vmax * local_uniform(local_rng);
        }
    }
    std::vector<const std::vector<double>*> pts;
    std::vector<double> vals;
    for (const Particle& p : swarm_) pts.push_back(&p.position);
    evaluate(pts, vals);
    for (int i = 0; i < swarm_size_; ++i) {
        Particle& p = swarm_[static_cast<size_t>(i)];
        p.current_fitness = vals[static_cast<size_t>(i)];
        p.pbest_position = p.position;
        p.pbest_value = p.current_fitness;
        p.quantum_position = p.position;
    }
    if (use_opposition_learning_) {  // :306-315
        oppositionBasedInitialization();
        pts.clear();
        for (const Particle& p : swarm_) pts.push_back(&p.position);
        evaluate(pts, vals);
        for (int i = 0; i < swarm_size_; ++i) {
            Particle& p = swarm_[static_cast<size_t>(i)];
            p.current_fitness = vals[static_cast<size_t>(i)];
            p.pbest_value = p.current_fitness;
            p.pbest_position = p.position;
        }
    }
    for (const Particle& p : swarm_)
        if (p.pbest_value > gbest_value_) {
            gbest_value_ = p.pbest_value;
            gbest_position_ = p.pbest_position;
        }
}

void BatchedParticleSwarmOptimization::oppositionBasedInitialization() {  // :527-574
    std::vector<Particle> opposite(static_cast<size_t>(swarm_size_));
    for (int i = 0; i < swarm_size_; ++i) {
        Particle& o = opposite[static_cast<size_t>(i)];
        const Particle& p = swarm_[static_cast<size_t>(i)];
        o.position.resize(static_cast<size_t>(n_));
        o.velocity.resize(static_cast<size_t>(n_));
        for (int k = 0; k < n_; ++k) {
            o.position[static_cast<size_t>(k)] = lb_[static_cast<size_t>(k)] + ub_[static_cast<size_t>(k)] - p.position[static_cast<size_t>(k)];
            o.velocity[static_cast<size_t>(k)] = -p.velocity[static_cast<size_t>(k)];
        }
        o.quantum_position = o.position;
        o.pbest_position = o.position;  // pbest_value stays -inf: the mirrored particles are not evaluated here
    }
    std::vector<std::pair<double, int>> fitness_indices;
    fitness_indices.reserve(static_cast<size_t>(2 * swarm_size_));
    for (int i = 0; i < swarm_size_; ++i) {
        fitness_indices.push_back({swarm_[static_cast<size_t>(i)].pbest_value, i});
        fitness_indices.push_back({opposite[static_cast<size_t>(i)].pbest_value, i + swarm_size_});
    }
    std::sort(fitness_indices.begin(), fitness_indices.end(), [](const auto& a, const auto& b) { return a.first > b.first; });
    std::vector<Particle> next(static_cast<size_t>(swarm_size_));
    for (int i = 0; i < swarm_size_; ++i) {
        const int idx = fitness_indices[static_cast<size_t>(i)].second;
        next[static_cast<size_t>(i)] = idx < swarm_size_ ? swarm_[static_cast<size_t>(idx)] : opposite[static_cast<size_t>(idx - swarm_size_)];
    }
    swarm_ = std::move(next);
}

double BatchedParticleSwarmOptimization::calculateEvolutionaryFactor() const {  // :445-479
    double mean_distance = 0.0, max_distance = 0.0;
    for (const Particle& p : swarm_) {
        double sq = 0.0;
        for (int k = 0; k < n_; ++k) {
            const double d = p.position[static_cast<size_t>(k)] - gbest_position_[static_cast<size_t>(k)];
            sq += d * d;
        }
        const double dist = std::sqrt(sq);
        mean_distance += dist;
        max_distance = std::max(max_distance, dist);
    }
    mean_distance /= swarm_size_;
    double mean_fitness = 0.0, max_fitness = -std::numeric_limits<double>::infinity(),
           min_fitness = std::numeric_limits<double>::infinity();
    for (const Particle& p : swarm_) {
        mean_fitness += p.current_fitness;
        max_fitness = std::max(max_fitness, p.current_fitness);
        min_fitness = std::min(min_fitness, p.current_fitness);
    }
    mean_fitness /= swarm_size_;
    const double fitness_range = (max_fitness - min_fitness) > 1e-10 ? (max_fitness - min_fitness) : 1e-10;
    const double distance_factor = (max_distance > 0) ? mean_distance / max_distance : 0.0;
    const double fitness_factor = (max_fitness - mean_fitness) / fitness_range;
    return 0.5 * distance_factor + 0.5 * (1.0 - fitness_factor);
}

BatchedParticleSwarmOptimization::EvolutionaryState BatchedParticleSwarmOptimization::estimateEvolutionaryState() const {
    const double f = calculateEvolutionaryFactor();  // :434-443
    if (f > 0.7) return EvolutionaryState::EXPLORATION;
    if (f > 0.4) return EvolutionaryState::EXPLOITATION;
    if (f > 0.2) return EvolutionaryState::CONVERGENCE;
    return EvolutionaryState::JUMPING_OUT;
}

void BatchedParticleSwarmOptimization::adaptParameters(EvolutionaryState state, int iteration, double& omega, double& c1,
                                                       double& c2) {  // :481-525
    const double ratio = (iterations_ > 1) ? static_cast<double>(iteration) / (iterations_ - 1) : 0.0;
    switch (state) {
        case EvolutionaryState::EXPLORATION:
            omega = 0.9 - 0.2 * ratio;
            c1 = 1.5 + 0.5 * std::sin(ratio * M_PI);
            c2 = 1.5 - 0.5 * std::sin(ratio * M_PI);
            break;
        case EvolutionaryState::EXPLOITATION:
            omega = 0.7 - 0.3 * ratio;
            c1 = 2.0 - ratio;
            c2 = 1.0 + ratio;
            break;
        case EvolutionaryState::CONVERGENCE:
            omega = 0.4 - 0.3 * ratio;
            c1 = 1.0 - 0.5 * ratio;
            c2 = 2.0 + 0.5 * ratio;
            break;
        case EvolutionaryState::JUMPING_OUT:
            omega = 0.9 + 0.1 * uniform_dist_(rng_);
            c1 = 2.5 + uniform_dist_(rng_);
            c2 = 0.5 + uniform_dist_(rng_);
            break;
    }
    omega = clampd(omega, 0.1, 1.0);
    c1 = clampd(c1, 0.0, 4.0);
    c2 = clampd(c2, 0.0, 4.0);
}

std::vector<int> BatchedParticleSwarmOptimization::getNeighbors(int particle_idx) {  // :836-906
    std::vector<int> neighbors;
    switch (topology_) {
        case TopologyType::GLOBAL_BEST:
            for (int i = 0; i < swarm_size_; ++i) neighbors.push_back(i);
            break;
        case TopologyType::LOCAL_BEST: {
            const int k = 2;
            neighbors.push_back(particle_idx);
            for (int j = 1; j <= k; ++j) {
                neighbors.push_back((particle_idx - j + swarm_size_) % swarm_size_);
                neighbors.push_back((particle_idx + j) % swarm_size_);
            }
        } break;
        case TopologyType::VON_NEUMANN: {
            const int grid_size = static_cast<int>(std::ceil(std::sqrt(swarm_size_)));
            const int row = particle_idx / grid_size, col = particle_idx % grid_size;
            neighbors.push_back(particle_idx);
            if (row > 0) { const int idx = (row - 1) * grid_size + col; if (idx < swarm_size_) neighbors.push_back(idx); }
            if (row < grid_size - 1) { const int idx = (row + 1) * grid_size + col; if (idx < swarm_size_) neighbors.push_back(idx); }
            if (col > 0) { const int idx = row * grid_size + (col - 1); if (idx < swarm_size_) neighbors.push_back(idx); }
            if (col < grid_size - 1) { const int idx = row * grid_size + (col + 1); if (idx < swarm_size_) neighbors.push_back(idx); }
        } break;
        case TopologyType::RANDOM_DYNAMIC: {
            neighbors.push_back(particle_idx);
            std::vector<int> candidates;
            for (int i = 0; i < swarm_size_; ++i)
                if (i != particle_idx) candidates.push_back(i);
            std::shuffle(candidates.begin(), candidates.end(), rng_);
            const int k = std::min(4, static_cast<int>(candidates.size()));
            neighbors.insert(neighbors.end(), candidates.begin(), candidates.begin() + k);
        } break;
    }
    return neighbors;
}

std::vector<double> BatchedParticleSwarmOptimization::getNeighborhoodBest(int particle_idx) {  // :816-834
    const std::vector<int> neighbors = getNeighbors(particle_idx);
    std::vector<double> best_position = swarm_[static_cast<size_t>(particle_idx)].pbest_position;
    double best_value = swarm_[static_cast<size_t>(particle_idx)].pbest_value;
    for (int j : neighbors)
        if (j >= 0 && j < swarm_size_ && swarm_[static_cast<size_t>(j)].pbest_value > best_value) {
            best_value = swarm_[static_cast<size_t>(j)].pbest_value;
            best_position = swarm_[static_cast<size_t>(j)].pbest_position;
        }
    return best_position;
}

void BatchedParticleSwarmOptimization::standardUpdate(Particle& p, const std::vector<double>& lbest, double omega, double c1,
                                                      double c2, std::mt19937& rng) {  // :576-618
    std::uniform_real_distribution<> local_uniform(0.0, 1.0);
    std::vector<double> r1(static_cast<size_t>(n_)), r2(static_cast<size_t>(n_));
    for (int i = 0; i < n_; ++i) {
        r1[static_cast<size_t>(i)] = local_uniform(rng);
        r2[static_cast<size_t>(i)] = local_uniform(rng);
    }
    for (int k = 0; k < n_; ++k) {
        const size_t u = static_cast<size_t>(k);
        const double cognitive = c1 * (r1[u] * (p.pbest_position[u] - p.position[u]));
        const double social = c2 * (r2[u] * (lbest[u] - p.position[u]));
        p.velocity[u] = omega * p.velocity[u] + cognitive + social;
    }
    for (int k = 0; k < n_; ++k) {
        const size_t u = static_cast<size_t>(k);
        const double vmax = 0.2 * (ub_[u] - lb_[u]);
        p.velocity[u] = clampd(p.velocity[u], -vmax, vmax);
    }
    for (int k = 0; k < n_; ++k) p.position[static_cast<size_t>(k)] += p.velocity[static_cast<size_t>(k)];
    for (int k = 0; k < n_; ++k) {
        const size_t u = static_cast<size_t>(k);
        if (p.position[u] < lb_[u]) {
            p.position[u] = lb_[u] + std::abs(p.position[u] - lb_[u]);
            p.velocity[u] *= -0.5;
        } else if (p.position[u] > ub_[u]) {
            p.position[u] = ub_[u] - std::abs(p.position[u] - ub_[u]);
            p.velocity[u] *= -0.5;
        }
        p.position[u] = clampd(p.position[u], lb_[u], ub_[u]);
    }
}

void BatchedParticleSwarmOptimization::quantumUpdate(Particle& p, const std::vector<double>& mean_best, int iteration,
                                                     std::mt19937& rng) {  // :620-653
    std::uniform_real_distribution<> local_uniform(0.0, 1.0);
    const double phi = local_uniform(rng);
    std::vector<double> attractor(static_cast<size_t>(n_));
    for (int k = 0; k < n_; ++k)
        attractor[static_cast<size_t>(k)] = phi * p.pbest_position[static_cast<size_t>(k)] + (1 - phi) * gbest_position_[static_cast<size_t>(k)];
    const double beta = quantum_beta_ * (1.0 - 0.5 * static_cast<double>(iteration) / iterations_);
    for (int k = 0; k < n_; ++k) {
        const size_t u = static_cast<size_t>(k);
        const double uu = local_uniform(rng);
        const double L = 2.0 * beta * std::abs(mean_best[u] - p.position[u]);
        if (local_uniform(rng) < 0.5) p.position[u] = attractor[u] + L * std::log(1.0 / uu);
        else p.position[u] = attractor[u] - L * std::log(1.0 / uu);
        p.position[u] = clampd(p.position[u], lb_[u], ub_[u]);
    }
    p.quantum_position = p.position;
}

double BatchedParticleSwarmOptimization::levyNumber(std::mt19937& rng) const {  // Mantegna, :918-934
    const double sigma_u = std::pow(std::tgamma(1 + levy_alpha_) * std::sin(M_PI * levy_alpha_ / 2) /
                                        (std::tgamma((1 + levy_alpha_) / 2) * levy_alpha_ * std::pow(2, (levy_alpha_ - 1) / 2)),
                                    1.0 / levy_alpha_);
    std::normal_distribution<> local_normal(0.0, 1.0);
    const double u = local_normal(rng) * sigma_u;
    const double v = std::max(std::abs(local_normal(rng)), 1e-10);
    const double levy_step = u / std::pow(v, 1.0 / levy_alpha_);
    return clampd(levy_step, -100.0, 100.0);
}

void BatchedParticleSwarmOptimization::levyUpdate(Particle& p, double omega, double c1, double c2, std::mt19937& rng) {  // :655-680
    standardUpdate(p, gbest_position_, omega, c1, c2, rng);
    std::uniform_real_distribution<> local_uniform(0.0, 1.0);
    const double levy_prob = 0.1 * (1.0 + p.success_rate);
    if (local_uniform(rng) < levy_prob) {
        std::vector<double> step(static_cast<size_t>(n_));
        for (int i = 0; i < n_; ++i) step[static_cast<size_t>(i)] = levyNumber(rng);
        const double step_scale = 0.01 * (1.0 - stagnation_counter_ / static_cast<double>(max_stagnation_));
        for (int k = 0; k < n_; ++k) {
            const size_t u = static_cast<size_t>(k);
            const double scale = step_scale * (ub_[u] - lb_[u]);
            p.position[u] += scale * step[u];
            p.position[u] = clampd(p.position[u], lb_[u], ub_[u]);
        }
    }
}

void BatchedParticleSwarmOptimization::updateParticles(int iteration) {  // :330-432
    double omega = omega_start_, c1 = c1_initial_, c2 = c2_initial_;
    if (use_adaptive_parameters_) {
        adaptParameters(estimateEvolutionaryState(), iteration, omega, c1, c2);
    } else {
        const double ratio = (iterations_ > 1) ? static_cast<double>(iteration) / (iterations_ - 1) : 0.0;
        omega = omega_start_ + (omega_end_ - omega_start_) * ratio;
        c1 = c1_initial_ + (c1_final_ - c1_initial_) * ratio;
        c2 = c2_initial_ + (c2_final_ - c2_initial_) * ratio;
    }
    // calculateMeanBestPosition :936-947.  The reference fills it for QUANTUM only (:356-359) and then hands the
    // EMPTY vector to quantumPSOUpdate from the HYBRID branch (:407-409), which reads past its end; the mean of the
    // personal bests at the start of the iteration is what that method is written for, so HYBRID gets it too.
    std::vector<double> mean_best;
    if (variant_ == PSOVariant::QUANTUM || variant_ == PSOVariant::HYBRID) {
        mean_best.assign(static_cast<size_t>(n_), 0.0);
        for (const Particle& p : swarm_)
            for (int k = 0; k < n_; ++k) mean_best[static_cast<size_t>(k)] += p.pbest_position[static_cast<size_t>(k)];
        for (double& m : mean_best) m /= swarm_size_;
    }
    std::vector<unsigned int> seeds(static_cast<size_t>(swarm_size_));
    for (int i = 0; i < swarm_size_; ++i) seeds[static_cast<size_t>(i)] = static_cast<unsigned int>(rng_());
    // neighbourhood bests as they stand at the start of the iteration (see the header)
    std::vector<std::vector<double>> lbest(static_cast<size_t>(swarm_size_));
    for (int i = 0; i < swarm_size_; ++i)
        lbest[static_cast<size_t>(i)] = (topology_ == TopologyType::GLOBAL_BEST) ? gbest_position_ : getNeighborhoodBest(i);
    for (int i = 0; i < swarm_size_; ++i) {
        Particle& p = swarm_[static_cast<size_t>(i)];
        std::mt19937 local_rng(seeds[static_cast<size_t>(i)]);
        std::uniform_real_distribution<> local_uniform(0.0, 1.0);
        switch (variant_) {
            case PSOVariant::STANDARD:
            case PSOVariant::ADAPTIVE:
                standardUpdate(p, lbest[static_cast<size_t>(i)], omega, c1, c2, local_rng);
                break;
            case PSOVariant::QUANTUM:
                quantumUpdate(p, mean_best, iteration, local_rng);
                break;
            case PSOVariant::LEVY_FLIGHT:
                levyUpdate(p, omega, c1, c2, local_rng);
                break;
            case PSOVariant::HYBRID:
                if (p.success_rate < 0.3 && local_uniform(local_rng) < 0.5) {
                    levyUpdate(p, omega, c1, c2, local_rng);
                } else if (p.success_rate > 0.7 && local_uniform(local_rng) < 0.3) {
                    quantumUpdate(p, mean_best, iteration, local_rng);
                } else {
                    standardUpdate(p, lbest[static_cast<size_t>(i)], omega, c1, c2, local_rng);
                }
                break;
        }
    }
    std::vector<const std::vector<double>*> pts;
    std::vector<double> vals;
    for (const Particle& p : swarm_) pts.push_back(&p.position);
    evaluate(pts, vals);
    for (int i = 0; i < swarm_size_; ++i) {
        Particle& p = swarm_[static_cast<size_t>(i)];
        const double new_fitness = vals[static_cast<size_t>(i)];
        p.current_fitness = new_fitness;
        p.total_updates++;
        if (new_fitness > p.pbest_value) {
            p.pbest_value = new_fitness;
            p.pbest_position = p.position;
            p.success_count++;
        }
        p.success_rate = (p.total_updates > 0) ? static_cast<double>(p.success_count) / p.total_updates : 0.0;
    }
}

void BatchedParticleSwarmOptimization::applyElitistLearningStrategy(Particle& best) {  // :706-740
    std::vector<double> trial = best.position;
    double sigma_scale = 0.1 * std::exp(-2.0 * best.success_rate);
    for (int attempt = 0; attempt < 3; ++attempt) {
        for (int k = 0; k < n_; ++k) {
            const size_t u = static_cast<size_t>(k);
            const double sigma = sigma_scale * (ub_[u] - lb_[u]);
            trial[u] = best.position[u] + sigma * normal_dist_(rng_);
            trial[u] = clampd(trial[u], lb_[u], ub_[u]);
        }
        const double trial_fitness = eval_one_(trial);
        if (trial_fitness > best.pbest_value) {
            best.position = trial;
            best.pbest_position = trial;
            best.pbest_value = trial_fitness;
            best.current_fitness = trial_fitness;
            break;
        }
        sigma_scale *= 0.5;
    }
}

void BatchedParticleSwarmOptimization::restartSwarm(int keep_best_count) {  // :742-814
    std::sort(swarm_.begin(), swarm_.end(), [](const Particle& a, const Particle& b) { return a.pbest_value > b.pbest_value; });
    const std::vector<Particle> elite(swarm_.begin(), swarm_.begin() + std::min(keep_best_count, swarm_size_));
    std::vector<unsigned int> seeds(static_cast<size_t>(swarm_size_));
    for (int i = 0; i < swarm_size_; ++i) seeds[static_cast<size_t>(i)] = static_cast<unsigned int>(rng_());
    for (int i = keep_best_count; i < swarm_size_; ++i) {
        Particle& p = swarm_[static_cast<size_t>(i)];
        std::mt19937 local_rng(seeds[static_cast<size_t>(i)]);
        std::uniform_real_distribution<> local_uniform(0.0, 1.0);
        std::normal_distribution<> local_normal(0.0, 1.0);
        const size_t elite_idx = static_cast<size_t>(i) % elite.size();
        for (int k = 0; k < n_; ++k) {
            const size_t u = static_cast<size_t>(k);
            if (local_uniform(local_rng) < 0.7) {
                const double range = ub_[u] - lb_[u];
                const double sigma = 0.3 * range * (1.0 + 0.5 * local_uniform(local_rng));
                p.position[u] = elite[elite_idx].position[u] + sigma * local_normal(local_rng);
            } else {
                p.position[u] = lb_[u] + local_uniform(local_rng) * (ub_[u] - lb_[u]);
            }
            p.position[u] = clampd(p.position[u], lb_[u], ub_[u]);
            const double vmax = 0.2 * (ub_[u] - lb_[u]);
            p.velocity[u] = -vmax + 2 * vmax * local_uniform(local_rng);
        }
    }
    std::vector<const std::vector<double>*> pts;
    std::vector<double> vals;
    for (int i = keep_best_count; i < swarm_size_; ++i) pts.push_back(&swarm_[static_cast<size_t>(i)].position);
    evaluate(pts, vals);
    for (int i = keep_best_count; i < swarm_size_; ++i) {
        Particle& p = swarm_[static_cast<size_t>(i)];
        p.current_fitness = vals[static_cast<size_t>(i - keep_best_count)];
        p.pbest_position = p.position;
        p.pbest_value = p.current_fitness;
        p.quantum_position = p.position;
        p.success_count = 0;
        p.total_updates = 0;
        p.success_rate = 0.0;
    }
    for (int i = 0; i < std::min(keep_best_count, swarm_size_); ++i) swarm_[static_cast<size_t>(i)] = elite[static_cast<size_t>(i)];
    gbest_value_ = swarm_[0].pbest_value;
    gbest_position_ = swarm_[0].pbest_position;
}

}  // namespace epidemic
