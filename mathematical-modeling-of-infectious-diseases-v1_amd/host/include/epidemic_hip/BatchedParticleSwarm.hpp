// host/include/epidemic_hip/BatchedParticleSwarm.hpp
//
// ParticleSwarmOptimization (src/model/optimizers/ParticleSwarmOptimizer.cpp:10-948,
// include/model/optimizers/ParticleSwarmOptimizer.hpp) with the swarm's objective calls grouped into one
// device launch per phase: the reference evaluates inside `#pragma omp parallel for
// if(use_parallel_)` loops over the particles (:283-312 initialisation, :377-432 update, :771-805 restart).
//
// Evaluation order.  Inside the reference's update loop a particle reads its neighbours' personal bests while
// other iterations of the same loop may already have rewritten them (serially: particles with a lower index
// have; in parallel: a race).  Here every particle of an iteration sees the personal bests as they stood when
// the iteration began, moves, and only then is the whole swarm evaluated -- the outcome the parallel loop aims
// at.  With the GLOBAL_BEST topology nothing is read from other particles during the loop, so this IS the
// reference's serial result.
//
// Everything else follows the reference: the five variants (standard, quantum, adaptive, Levy flight, hybrid),
// the four topologies, opposition-based initialisation (whose mirrored particles are never evaluated before the
// selection, :527-574), evolutionary-state parameter adaptation, elitist learning every fifth iteration (its three
// trials stay one call at a time: each consumes random draws only if the previous one failed), stagnation
// restarts, and the swarm covariance handed to the sampler (:221-239).  The reference seeds `rng_` from
// std::random_device (:578); configure() here takes a `seed`.
#pragma once
#include <cstdint>
#include <functional>
#include <map>
#include <random>
#include <string>
#include <vector>

#include "epidemic_hip/Interfaces.hpp"

namespace epidemic {

class BatchedParticleSwarmOptimization : public IOptimizationAlgorithm {
public:
    enum class PSOVariant { STANDARD = 0, QUANTUM = 1, ADAPTIVE = 2, LEVY_FLIGHT = 3, HYBRID = 4 };
    enum class TopologyType { GLOBAL_BEST = 0, LOCAL_BEST = 1, VON_NEUMANN = 2, RANDOM_DYNAMIC = 3 };

    void configure(const std::map<std::string, double>& settings) override;
    OptimizationResult optimize(const Eigen::VectorXd& initialParameters, IObjectiveFunction& objectiveFunction,
                                IParameterManager& parameterManager) override;
    const std::vector<double>& bestTrace() const { return trace_; }  // global best after every iteration
    long evaluations() const { return evaluations_; }
    long launches() const { return launches_; }

private:
    struct Particle {
        std::vector<double> position, velocity, pbest_position, quantum_position;
        double pbest_value = -std::numeric_limits<double>::infinity();
        double current_fitness = -std::numeric_limits<double>::infinity();
        double success_rate = 0.0;
        int success_count = 0, total_updates = 0;
    };
    enum class EvolutionaryState { EXPLORATION, EXPLOITATION, CONVERGENCE, JUMPING_OUT };
    using BatchEval = std::function<void(const double*, int, double*)>;

    void evaluate(const std::vector<const std::vector<double>*>& points, std::vector<double>& values);
    void initializeSwarm(const std::vector<double>* initial);
    void updateParticles(int iteration);
    void restartSwarm(int keep_best_count = 3);
    void oppositionBasedInitialization();
    void applyElitistLearningStrategy(Particle& best);
    double calculateEvolutionaryFactor() const;
    EvolutionaryState estimateEvolutionaryState() const;
    void adaptParameters(EvolutionaryState state, int iteration, double& omega, double& c1, double& c2);
    std::vector<int> getNeighbors(int particle_idx);
    std::vector<double> getNeighborhoodBest(int particle_idx);
    void standardUpdate(Particle& p, const std::vector<double>& lbest, double omega, double c1, double c2, std::mt19937& rng);
    void quantumUpdate(Particle& p, const std::vector<double>& mean_best, int iteration, std::mt19937& rng);
    void levyUpdate(Particle& p, double omega, double c1, double c2, std::mt19937& rng);
    double levyNumber(std::mt19937& rng) const;

    // settings (defaults of ParticleSwarmOptimizer.hpp:172-231)
    int iterations_ = 100, swarm_size_ = 30, report_interval_ = 10, max_stagnation_ = 50;
    double omega_start_ = 0.9, omega_end_ = 0.4, c1_initial_ = 2.5, c1_final_ = 0.5, c2_initial_ = 0.5, c2_final_ = 2.5;
    PSOVariant variant_ = PSOVariant::STANDARD;
    TopologyType topology_ = TopologyType::GLOBAL_BEST;
    bool use_opposition_learning_ = false, use_adaptive_parameters_ = false;
    double diversity_threshold_ = 0.1, restart_threshold_ = 1e-6, quantum_beta_ = 1.0, levy_alpha_ = 1.5;
    uint32_t seed_ = 1;

    // run state
    int n_ = 0, stagnation_counter_ = 0;
    std::vector<double> lb_, ub_;
    std::vector<Particle> swarm_;
    std::vector<double> gbest_position_;
    double gbest_value_ = 0.0;
    std::mt19937 rng_;
    std::uniform_real_distribution<> uniform_dist_{0.0, 1.0};
    std::normal_distribution<> normal_dist_{0.0, 1.0};
    BatchEval eval_;
    std::function<double(const std::vector<double>&)> eval_one_;
    std::vector<double> trace_;
    long evaluations_ = 0, launches_ = 0;
};

}  // namespace epidemic
