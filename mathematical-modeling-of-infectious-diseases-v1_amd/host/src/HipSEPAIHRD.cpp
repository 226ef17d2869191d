// HipSEPAIHRD.cpp -- parameter manager, cache and objective adapters above the C ABI.
#include "epidemic_hip/HipSEPAIHRD.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>

#include "sepaihrd_hip.h"

namespace epidemic {

namespace {
bool starts_with(const std::string& s, const char* prefix) { return s.rfind(prefix, 0) == 0; }

size_t parse_index(const std::string& name, size_t prefix_len, const char* where) {
    try {
        return std::stoul(name.substr(prefix_len));
    } catch (const std::exception&) {
        throw InvalidParameterException(where, "Could not parse index from parameter: " + name);
    }
}

// SEPAIHRDParameterManager.cpp:302-313
double reflectBound(double value, double minb, double maxb) {
    if (minb >= maxb) return minb;
    const double width = maxb - minb;
    double y = std::fmod(value - minb, 2.0 * width);
    if (y < 0) y += 2.0 * width;
    if (y <= width) return minb + y;
    return maxb - (y - width);
}

// MurmurHash3 finaliser used by SimulationCache.cpp:12-19
inline uint64_t mix_hash(uint64_t k) {
    k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33;
    return k;
}
}  // namespace

// ------------------------------------------------------------------ SimulationCache
SimulationCache::SimulationCache(size_t max_size) : capacity_(max_size) {
    if (max_size == 0) throw std::invalid_argument("SimulationCache: max_size must be > 0.");
}
size_t SimulationCache::computeHash(const Eigen::VectorXd& params) const {
    uint64_t seed = 0;
    for (Eigen::Index i = 0; i < params.size(); ++i) {
        const long long quantized = static_cast<long long>(params[i] * 1e8 + 0.5);
        seed ^= mix_hash(static_cast<uint64_t>(quantized)) + 0x9e3779b9ULL + (seed << 6) + (seed >> 2);
    }
    return static_cast<size_t>(seed);
}
bool SimulationCache::getLikelihood(size_t key, double& value) {
    ++calls_;
    auto it = map_.find(key);
    if (it == map_.end()) return false;
    ++hits_;
    it->second.freq++;
    it->second.tick = ++tick_;
    value = it->second.value;  // returned on hash equality alone, as in the reference
    return true;
}
void SimulationCache::storeLikelihood(size_t key, double value) {
    auto it = map_.find(key);
    if (it != map_.end()) { it->second.value = value; it->second.tick = ++tick_; return; }
    if (map_.size() >= capacity_) {  // least frequently used, ties by age (SimulationCache.cpp:74-105)
        auto victim = map_.begin();
        for (auto e = map_.begin(); e != map_.end(); ++e)
            if (e->second.freq < victim->second.freq ||
                (e->second.freq == victim->second.freq && e->second.tick < victim->second.tick))
                victim = e;
        map_.erase(victim);
    }
    map_[key] = Entry{value, 1, ++tick_};
}
std::optional<double> SimulationCache::get(const Eigen::VectorXd& p) {
    double v;
    if (getLikelihood(computeHash(p), v)) return v;
    return std::nullopt;
}
void SimulationCache::set(const Eigen::VectorXd& p, double r) { storeLikelihood(computeHash(p), r); }
void SimulationCache::clear() { map_.clear(); }
size_t SimulationCache::size() const { return map_.size(); }
std::string SimulationCache::createCacheKey(const Eigen::VectorXd& p) const { return std::to_string(computeHash(p)); }
bool SimulationCache::getLikelihood(const std::string& key, double& value) {
    return getLikelihood(static_cast<size_t>(std::stoull(key)), value);
}
void SimulationCache::storeLikelihood(const std::string& key, double value) {
    storeLikelihood(static_cast<size_t>(std::stoull(key)), value);
}

// ------------------------------------------------------------------ parameter manager
HipSEPAIHRDParameterManager::HipSEPAIHRDParameterManager(
    const SEPAIHRDParameters& model_params, const std::vector<std::string>& params_to_calibrate,
    const std::map<std::string, double>& proposal_sigmas,
    const std::map<std::string, std::pair<double, double>>& param_bounds,
    const std::vector<std::string>& npi_param_names)
    : params_(model_params), names_(params_to_calibrate), npi_names_(npi_param_names) {
    const char* W = "SEPAIHRDParameterManager";
    if (names_.empty()) throw InvalidParameterException(W, "Parameter names list (params_to_calibrate) cannot be empty.");
    const size_t n = static_cast<size_t>(params_.N.size());
    if (params_.kappa_values.empty() || params_.kappa_values.size() != params_.kappa_end_times.size())
        throw InvalidParameterException(W, "kappa schedule needs a baseline entry");
    if (params_.d_community.size() == 0) params_.d_community = Eigen::VectorXd::Zero(static_cast<Eigen::Index>(n));
    if (npi_names_.empty())
        for (size_t k = 1; k < params_.kappa_values.size(); ++k) npi_names_.push_back("kappa_" + std::to_string(k + 1));
    if (npi_names_.size() != params_.kappa_values.size() - 1)
        throw InvalidParameterException(W, "NPI parameter names must match the after-baseline kappa values");

    struct AgePrefix { const char* prefix; int field; };
    // dispatch order of updateModelParameters (:221-228): "h_infec_" is tested before "h_"
    static const AgePrefix age_prefixes[] = {
        {"a_", SEPAIHRD_F_A}, {"h_infec_", SEPAIHRD_F_H_INFEC}, {"p_", SEPAIHRD_F_P}, {"h_", SEPAIHRD_F_H},
        {"icu_", SEPAIHRD_F_ICU}, {"d_H_", SEPAIHRD_F_D_H}, {"d_ICU_", SEPAIHRD_F_D_ICU},
        {"d_community_", SEPAIHRD_F_D_COMMUNITY}};
    static const std::pair<const char*, int> scalars_a[] = {
        {"theta", SEPAIHRD_F_THETA}, {"sigma", SEPAIHRD_F_SIGMA}, {"gamma_p", SEPAIHRD_F_GAMMA_P},
        {"gamma_A", SEPAIHRD_F_GAMMA_A}, {"gamma_I", SEPAIHRD_F_GAMMA_I}, {"gamma_H", SEPAIHRD_F_GAMMA_H},
        {"gamma_ICU", SEPAIHRD_F_GAMMA_ICU}};
    static const std::pair<const char*, int> scalars_b[] = {
        {"seed_exposed", SEPAIHRD_F_SEED_EXPOSED}, {"runup_days", SEPAIHRD_F_RUNUP_DAYS},
        {"E0_multiplier", SEPAIHRD_F_E0_MULT}, {"P0_multiplier", SEPAIHRD_F_P0_MULT},
        {"A0_multiplier", SEPAIHRD_F_A0_MULT}, {"I0_multiplier", SEPAIHRD_F_I0_MULT},
        {"H0_multiplier", SEPAIHRD_F_H0_MULT}, {"ICU0_multiplier", SEPAIHRD_F_ICU0_MULT},
        {"R0_multiplier", SEPAIHRD_F_R0_MULT}, {"D0_multiplier", SEPAIHRD_F_D0_MULT}};

    for (const std::string& name : names_) {
        auto sg = proposal_sigmas.find(name);
        if (sg == proposal_sigmas.end()) throw InvalidParameterException(W, "Missing proposal sigma for parameter: " + name);
        auto bd = param_bounds.find(name);
        if (bd == param_bounds.end()) throw InvalidParameterException(W, "Missing bounds for parameter: " + name);
        sigma_.push_back(sg->second);
        lower_.push_back(bd->second.first);
        upper_.push_back(bd->second.second);
        has_bounds_.push_back(1);

        int field = SEPAIHRD_F_NONE, index = 0;
        bool done = false;
        if (name == "beta") { field = SEPAIHRD_F_BETA; done = true; }
        else if (starts_with(name, "beta_")) {
            const size_t k = parse_index(name, 5, W);
            if (k < 1 || k > params_.beta_values.size())
                throw InvalidParameterException("updateModelParameters", "Beta index out of range for name: " + name);
            field = SEPAIHRD_F_BETA_VALUE; index = static_cast<int>(k - 1); done = true;
        }
        for (const auto& s : scalars_a) if (!done && name == s.first) { field = s.second; done = true; }
        for (const auto& ap : age_prefixes)
            if (!done && starts_with(name, ap.prefix)) {
                const size_t idx = parse_index(name, std::strlen(ap.prefix), W);
                if (idx >= n) {
                    if (ap.field == SEPAIHRD_F_D_COMMUNITY) { field = SEPAIHRD_F_NONE; done = true; break; }
                    throw InvalidParameterException(W, "Invalid age index for parameter " + name);
                }
                field = ap.field; index = static_cast<int>(idx); done = true;
            }
        for (const auto& s : scalars_b) if (!done && name == s.first) { field = s.second; done = true; }
        if (!done && starts_with(name, "kappa_")) {
            bool found = false;
            for (size_t k = 0; k < npi_names_.size(); ++k)
                if (npi_names_[k] == name) { field = SEPAIHRD_F_KAPPA_VALUE; index = static_cast<int>(k + 1); found = true; break; }
            if (!found) {
                if (name == "kappa_1" || name == "kappa_baseline")
                    throw InvalidParameterException(W, "Parameter '" + name + "' refers to the fixed baseline kappa and cannot be in params_to_calibrate.");
                throw InvalidParameterException(W, "NPI parameter '" + name + "' in params_to_calibrate is not recognized as a calibratable NPI parameter by the strategy.");
            }
            done = true;
        }
        field_.push_back(field);  // SEPAIHRD_F_NONE: unknown name, ignored with a warning in the reference
        index_.push_back(index);
    }
}

namespace {
// names of kappa_values[1..] as the model's strategy spells them (SEPAIHRDParameterManager.cpp:63-88)
std::vector<std::string> npiNamesOf(const std::shared_ptr<AgeSEPAIHRDModel>& model) {
    if (!model) throw InvalidParameterException("SEPAIHRDParameterManager", "Model pointer cannot be null.");
    auto base = model->getNpiStrategy();
    auto piecewise = std::dynamic_pointer_cast<PiecewiseConstantNpiStrategy>(base);
    if (!piecewise)
        throw ModelException("SEPAIHRDParameterManager", "NPI strategy is not a PiecewiseConstantNpiStrategy, cannot resolve kappa names.");
    if (!piecewise->isBaselineFixed())
        throw InvalidParameterException("SEPAIHRDParameterManager",
                                        "a calibratable baseline kappa (kappa_baseline) is not built on the device path: kappa_1 is fixed");
    std::vector<std::string> names;
    for (size_t k = 0; k < piecewise->getNumCalibratableNpiParams(); ++k) names.push_back(piecewise->getNpiParamName(static_cast<int>(k)));
    return names;
}
const SEPAIHRDParameters& checkedParameters(const std::shared_ptr<AgeSEPAIHRDModel>& model, SEPAIHRDParameters& storage) {
    if (!model) throw InvalidParameterException("SEPAIHRDParameterManager", "Model pointer cannot be null.");
    storage = model->getModelParameters();
    return storage;
}
}  // namespace

HipSEPAIHRDParameterManager::HipSEPAIHRDParameterManager(std::shared_ptr<AgeSEPAIHRDModel> model,
                                                         const std::vector<std::string>& params_to_calibrate,
                                                         const std::map<std::string, double>& proposal_sigmas,
                                                         const std::map<std::string, std::pair<double, double>>& param_bounds)
    : HipSEPAIHRDParameterManager([&]() -> SEPAIHRDParameters { SEPAIHRDParameters p; return checkedParameters(model, p); }(),
                                  params_to_calibrate, proposal_sigmas, param_bounds, npiNamesOf(model)) {
    model_ = std::move(model);
}

double* HipSEPAIHRDParameterManager::slot(int field, int index) {
    SEPAIHRDParameters& p = params_;
    switch (field) {
        case SEPAIHRD_F_BETA: return &p.beta;
        case SEPAIHRD_F_THETA: return &p.theta;
        case SEPAIHRD_F_SIGMA: return &p.sigma;
        case SEPAIHRD_F_GAMMA_P: return &p.gamma_p;
        case SEPAIHRD_F_GAMMA_A: return &p.gamma_A;
        case SEPAIHRD_F_GAMMA_I: return &p.gamma_I;
        case SEPAIHRD_F_GAMMA_H: return &p.gamma_H;
        case SEPAIHRD_F_GAMMA_ICU: return &p.gamma_ICU;
        case SEPAIHRD_F_E0_MULT: return &p.E0_multiplier;
        case SEPAIHRD_F_P0_MULT: return &p.P0_multiplier;
        case SEPAIHRD_F_A0_MULT: return &p.A0_multiplier;
        case SEPAIHRD_F_I0_MULT: return &p.I0_multiplier;
        case SEPAIHRD_F_H0_MULT: return &p.H0_multiplier;
        case SEPAIHRD_F_ICU0_MULT: return &p.ICU0_multiplier;
        case SEPAIHRD_F_R0_MULT: return &p.R0_multiplier;
        case SEPAIHRD_F_D0_MULT: return &p.D0_multiplier;
        case SEPAIHRD_F_RUNUP_DAYS: return &p.runup_days;
        case SEPAIHRD_F_SEED_EXPOSED: return &p.seed_exposed;
        case SEPAIHRD_F_BETA_VALUE: return &p.beta_values[static_cast<size_t>(index)];
        case SEPAIHRD_F_KAPPA_VALUE: return &p.kappa_values[static_cast<size_t>(index)];
        case SEPAIHRD_F_A: return &p.a[index];
        case SEPAIHRD_F_H_INFEC: return &p.h_infec[index];
        case SEPAIHRD_F_P: return &p.p[index];
        case SEPAIHRD_F_H: return &p.h[index];
        case SEPAIHRD_F_ICU: return &p.icu[index];
        case SEPAIHRD_F_D_H: return &p.d_H[index];
        case SEPAIHRD_F_D_ICU: return &p.d_ICU[index];
        case SEPAIHRD_F_D_COMMUNITY: return &p.d_community[index];
        default: return nullptr;
    }
}

Eigen::VectorXd HipSEPAIHRDParameterManager::getCurrentParameters() const {
    Eigen::VectorXd out(static_cast<Eigen::Index>(names_.size()));
    auto* self = const_cast<HipSEPAIHRDParameterManager*>(this);
    for (size_t i = 0; i < names_.size(); ++i) {
        const double* s = self->slot(field_[i], index_[i]);
        if (!s) throw InvalidParameterException("SEPAIHRDParameterManager::getCurrentParameters", "Unknown parameter name: " + names_[i]);
        out[static_cast<Eigen::Index>(i)] = *s;
    }
    return out;
}

void HipSEPAIHRDParameterManager::updateModelParameters(const Eigen::VectorXd& parameters) {
    if (static_cast<size_t>(parameters.size()) != names_.size())
        throw InvalidParameterException("SEPAIHRDParameterManager::updateModelParameters", "Parameter vector size mismatch.");
    const Eigen::VectorXd c = applyConstraints(parameters);
    bool kappa_touched = false;
    for (size_t i = 0; i < names_.size(); ++i) {
        if (field_[i] == SEPAIHRD_F_KAPPA_VALUE) kappa_touched = true;
        if (double* s = slot(field_[i], index_[i])) *s = c[static_cast<Eigen::Index>(i)];
    }
    if (kappa_touched)
        for (size_t k = 1; k < params_.kappa_values.size(); ++k)
            if (params_.kappa_values[k] < 0.0)
                throw InvalidParameterException("PiecewiseConstantNpiStrategy::setCalibratableValues", "All NPI kappa values must be non-negative.");
    if (model_) {  // the reference's manager writes into the model it was given (:269-286)
        model_->setModelParameters(params_);
        if (kappa_touched)
            model_->getNpiStrategy()->setValues(std::vector<double>(params_.kappa_values.begin() + 1, params_.kappa_values.end()));
    }
}

double HipSEPAIHRDParameterManager::getSigmaForParamIndex(int index) const {
    if (index < 0 || static_cast<size_t>(index) >= names_.size())
        throw std::out_of_range("SEPAIHRDParameterManager::getSigmaForParamIndex: Index out of bounds.");
    return sigma_[static_cast<size_t>(index)];
}

Eigen::VectorXd HipSEPAIHRDParameterManager::applyConstraints(const Eigen::VectorXd& parameters) const {
    if (static_cast<size_t>(parameters.size()) != names_.size())
        throw InvalidParameterException("SEPAIHRDParameterManager::applyConstraints", "Parameter vector size mismatch.");
    Eigen::VectorXd c = parameters;
    for (size_t i = 0; i < names_.size(); ++i) {
        const Eigen::Index k = static_cast<Eigen::Index>(i);
        if (has_bounds_[i]) {
            double minb = lower_[i], maxb = upper_[i];
            if (minb > maxb) std::swap(minb, maxb);
            c[k] = mode_ == ConstraintMode::OPTIMIZATION_CLAMP ? std::min(std::max(parameters[k], minb), maxb)
                                                               : reflectBound(parameters[k], minb, maxb);
        } else {
            c[k] = mode_ == ConstraintMode::OPTIMIZATION_CLAMP ? std::max(0.0, parameters[k]) : std::abs(parameters[k]);
        }
    }
    return c;
}

int HipSEPAIHRDParameterManager::getIndexForParam(const std::string& name) const {
    auto it = std::find(names_.begin(), names_.end(), name);
    return it == names_.end() ? -1 : static_cast<int>(it - names_.begin());
}
double HipSEPAIHRDParameterManager::getLowerBoundForParamIndex(int idx) const {
    if (idx < 0 || static_cast<size_t>(idx) >= names_.size())
        throw std::out_of_range("SEPAIHRDParameterManager::getLowerBoundForParamIndex: Index out of bounds.");
    return lower_[static_cast<size_t>(idx)];
}
double HipSEPAIHRDParameterManager::getUpperBoundForParamIndex(int idx) const {
    if (idx < 0 || static_cast<size_t>(idx) >= names_.size())
        throw std::out_of_range("SEPAIHRDParameterManager::getUpperBoundForParamIndex: Index out of bounds.");
    return upper_[static_cast<size_t>(idx)];
}

// ------------------------------------------------------------------ objective
HipSEPAIHRDObjectiveFunction::HipSEPAIHRDObjectiveFunction(
    HipSEPAIHRDParameterManager& parameterManager, ISimulationCache& cache, const CalibrationData& data,
    const std::vector<double>& time_points, const Eigen::VectorXd& initial_state,
    std::shared_ptr<IOdeSolverStrategy> solver_strategy, double abs_error, double rel_error, int device,
    bool fma_arithmetic)
    : pm_(parameterManager), cache_(cache) {
    ctx_ = createContext(pm_, data, time_points, initial_state, solver_strategy, abs_error, rel_error, device,
                         fma_arithmetic, nullptr);
    device_mode_ = pm_.getConstraintMode() == ConstraintMode::MCMC_REFLECT ? SEPAIHRD_CONSTRAINT_REFLECT : SEPAIHRD_CONSTRAINT_CLAMP;
}

sepaihrd_ctx* HipSEPAIHRDObjectiveFunction::createContext(
    const HipSEPAIHRDParameterManager& pm_, const CalibrationData& data, const std::vector<double>& time_points,
    const Eigen::VectorXd& initial_state, const std::shared_ptr<IOdeSolverStrategy>& solver_strategy, double abs_error,
    double rel_error, int device, bool fma_arithmetic, const double* multipliers_override) {
    const char* W = "SEPAIHRDObjectiveFunction";
    const SEPAIHRDParameters& mp = pm_.modelParameters();
    const int n = static_cast<int>(mp.N.size());
    if (!solver_strategy) throw InvalidParameterException("Simulator::Simulator", "Solver strategy pointer cannot be null.");
    int solver;
    if (dynamic_cast<Dopri5SolverStrategy*>(solver_strategy.get())) solver = SEPAIHRD_SOLVER_DOPRI5;
    else if (dynamic_cast<CashKarpSolverStrategy*>(solver_strategy.get())) solver = SEPAIHRD_SOLVER_CASH_KARP54;
    else throw InvalidParameterException(W, "solver strategy has no HIP kernel (Dopri5 and Cash-Karp are built)");
    if (initial_state.size() != 11 * n)
        throw InvalidParameterException("Simulator::run", "Initial state size does not match model state size.");
    if (time_points.empty()) throw InvalidParameterException("Simulator::run", "Output time points vector cannot be empty.");

    const int T_obs = static_cast<int>(data.getNewDeaths().rows());
    auto row_major = [&](const Eigen::MatrixXd& m) {
        std::vector<double> v(static_cast<size_t>(m.rows() * m.cols()));
        for (Eigen::Index i = 0; i < m.rows(); ++i)
            for (Eigen::Index j = 0; j < m.cols(); ++j) v[static_cast<size_t>(i * m.cols() + j)] = m(i, j);
        return v;
    };
    const std::vector<double> oH = row_major(data.getNewHospitalizations()), oI = row_major(data.getNewICU()),
                              oD = row_major(data.getNewDeaths());
    const size_t P = pm_.getParameterCount();
    std::vector<double> lo(P), hi(P);
    std::vector<uint8_t> has(P);
    for (size_t i = 0; i < P; ++i) {
        lo[i] = pm_.getLowerBoundForParamIndex(static_cast<int>(i));
        hi[i] = pm_.getUpperBoundForParamIndex(static_cast<int>(i));
        has[i] = pm_.hasBounds(static_cast<int>(i)) ? 1 : 0;
        // finite-difference context: the multipliers reach the initial state UNCONSTRAINED
        // (SEPAIHRDGradientObjectiveFunction.cpp:59-83 reads them from params_plus)
        const int32_t f = pm_.fieldCodes()[i];
        if (multipliers_override && f >= SEPAIHRD_F_E0_MULT && f <= SEPAIHRD_F_D0_MULT) {
            lo[i] = -std::numeric_limits<double>::infinity();
            hi[i] = std::numeric_limits<double>::infinity();
            has[i] = 1;
        }
    }

    sepaihrd_problem pb;
    std::memset(&pb, 0, sizeof(pb));
    pb.abi_version = SEPAIHRD_ABI_VERSION;
    pb.n_age = n;
    pb.n_times = static_cast<int32_t>(time_points.size());
    pb.n_obs = T_obs;
    pb.n_beta = static_cast<int32_t>(mp.beta_values.size() == mp.beta_end_times.size() ? mp.beta_values.size() : 0);
    pb.n_kappa = static_cast<int32_t>(mp.kappa_values.size());
    pb.n_params = static_cast<int32_t>(P);
    pb.solver = solver;
    pb.constraint_mode = pm_.getConstraintMode() == ConstraintMode::MCMC_REFLECT ? SEPAIHRD_CONSTRAINT_REFLECT : SEPAIHRD_CONSTRAINT_CLAMP;
    pb.arith = fma_arithmetic ? SEPAIHRD_ARITH_FMA : SEPAIHRD_ARITH_STRICT;
    pb.times = time_points.data();
    pb.N = mp.N.data();
    pb.M = mp.M_baseline.data();  // column-major, like Eigen
    pb.a = mp.a.data(); pb.h_infec = mp.h_infec.data(); pb.p = mp.p.data(); pb.h = mp.h.data();
    pb.icu = mp.icu.data(); pb.d_H = mp.d_H.data(); pb.d_ICU = mp.d_ICU.data(); pb.d_community = mp.d_community.data();
    pb.beta_end_times = mp.beta_end_times.data(); pb.beta_values = mp.beta_values.data();
    pb.kappa_end_times = mp.kappa_end_times.data(); pb.kappa_values = mp.kappa_values.data();
    pb.initial_state = initial_state.data();
    pb.obs_H = oH.data(); pb.obs_ICU = oI.data(); pb.obs_D = oD.data();
    pb.param_field = pm_.fieldCodes().data(); pb.param_index = pm_.fieldIndices().data();
    pb.lower = lo.data(); pb.upper = hi.data(); pb.has_bounds = has.data();
    pb.beta = mp.beta; pb.theta = mp.theta; pb.sigma = mp.sigma; pb.gamma_p = mp.gamma_p; pb.gamma_A = mp.gamma_A;
    pb.gamma_I = mp.gamma_I; pb.gamma_H = mp.gamma_H; pb.gamma_ICU = mp.gamma_ICU;
    const double mult[8] = {mp.E0_multiplier, mp.P0_multiplier, mp.A0_multiplier, mp.I0_multiplier,
                            mp.H0_multiplier, mp.ICU0_multiplier, mp.R0_multiplier, mp.D0_multiplier};
    std::memcpy(pb.multipliers, multipliers_override ? multipliers_override : mult, sizeof(mult));
    pb.runup_days = mp.runup_days; pb.seed_exposed = mp.seed_exposed;
    pb.abs_err = abs_error; pb.rel_err = rel_error; pb.dt_hint = 1.0;  // simulator built with dt = 1.0 (:113-114)

    char err[512] = {0};
    sepaihrd_ctx* ctx = sepaihrd_create(&pb, device, err, sizeof(err));
    if (!ctx) throw ModelException(W, std::string("sepaihrd_create failed: ") + err);
    return ctx;
}

int HipSEPAIHRDObjectiveFunction::environmentDevice() {
    const char* e = std::getenv("SEPAIHRD_DEVICE");
    return e ? std::atoi(e) : -1;
}
// The arithmetic of the reference-shaped constructors: `fma` unless SEPAIHRD_ARITH=strict.  Settled by measurement
// (profiles/r04_fma_vs_strict_100k.json): the headline problem's 4096 chains over the reference's own 100 000
// iterations, same seeds, once in each arithmetic -- 409 595 904 accept decisions, none different, samples bit-identical.
bool HipSEPAIHRDObjectiveFunction::environmentFma() {
    const char* e = std::getenv("SEPAIHRD_ARITH");
    return !(e && std::string(e) == "strict");
}

HipSEPAIHRDParameterManager& HipSEPAIHRDObjectiveFunction::resolveManager(
    const std::shared_ptr<AgeSEPAIHRDModel>& model, IParameterManager& given, std::unique_ptr<HipSEPAIHRDParameterManager>& owned) {
    const char* W = "SEPAIHRDObjectiveFunction";
    if (!model) throw InvalidParameterException(W, "Model pointer cannot be null.");
    if (auto* hip = dynamic_cast<HipSEPAIHRDParameterManager*>(&given)) {
        if (static_cast<int>(hip->modelParameters().N.size()) != model->getNumAgeClasses())
            throw InvalidParameterException(W, "parameter manager and model disagree on the number of age classes");
        return *hip;
    }
    // any other manager: same names, sigmas and bounds, resolved once against the model's parameters
    const std::vector<std::string>& names = given.getParameterNames();
    std::map<std::string, double> sigmas;
    std::map<std::string, std::pair<double, double>> bounds;
    for (size_t i = 0; i < names.size(); ++i) {
        const int k = static_cast<int>(i);
        sigmas[names[i]] = given.getSigmaForParamIndex(k);
        bounds[names[i]] = {given.getLowerBoundForParamIndex(k), given.getUpperBoundForParamIndex(k)};
    }
    owned = std::make_unique<HipSEPAIHRDParameterManager>(model->getModelParameters(), names, sigmas, bounds, npiNamesOf(model));
    return *owned;
}

HipSEPAIHRDObjectiveFunction::HipSEPAIHRDObjectiveFunction(
    std::shared_ptr<AgeSEPAIHRDModel> model, IParameterManager& parameterManager, ISimulationCache& cache,
    const CalibrationData& data, const std::vector<double>& time_points, const Eigen::VectorXd& initial_state,
    std::shared_ptr<IOdeSolverStrategy> solver_strategy, double abs_error, double rel_error)
    : pm_(resolveManager(model, parameterManager, owned_pm_)), cache_(cache) {
    if (owned_pm_) foreign_pm_ = &parameterManager;
    syncConstraintMode();  // a foreign manager's mode, before the context is built with it
    ctx_ = createContext(pm_, data, time_points, initial_state, solver_strategy, abs_error, rel_error, environmentDevice(),
                         environmentFma(), nullptr);
    device_mode_ = pm_.getConstraintMode() == ConstraintMode::MCMC_REFLECT ? SEPAIHRD_CONSTRAINT_REFLECT : SEPAIHRD_CONSTRAINT_CLAMP;
}

HipSEPAIHRDObjectiveFunction::~HipSEPAIHRDObjectiveFunction() { sepaihrd_destroy(ctx_); }

const std::vector<std::string>& HipSEPAIHRDObjectiveFunction::getParameterNames() const { return pm_.getParameterNames(); }

void HipSEPAIHRDObjectiveFunction::syncConstraintMode() const {
    // the reference consults the SHARED parameter manager's mode_ on every evaluation
    if (foreign_pm_) {
        // IParameterManager has no mode getter: a value a quarter width above the upper bound comes back AS the bound
        // under OPTIMIZATION_CLAMP and mirrored below it under MCMC_REFLECT (SEPAIHRDParameterManager.cpp:302-347)
        const size_t P = pm_.getParameterCount();
        Eigen::VectorXd probe = pm_.getCurrentParameters();
        int k = -1;
        for (size_t i = 0; i < P && k < 0; ++i) {
            const double lo = pm_.getLowerBoundForParamIndex(static_cast<int>(i)), hi = pm_.getUpperBoundForParamIndex(static_cast<int>(i));
            if (hi > lo && std::isfinite(hi - lo)) k = static_cast<int>(i);
        }
        if (k >= 0) {
            const double lo = pm_.getLowerBoundForParamIndex(k), hi = pm_.getUpperBoundForParamIndex(k);
            for (size_t i = 0; i < P; ++i) {  // the other entries sit inside their bounds
                const double l = pm_.getLowerBoundForParamIndex(static_cast<int>(i)), h = pm_.getUpperBoundForParamIndex(static_cast<int>(i));
                probe[static_cast<Eigen::Index>(i)] = std::min(std::max(probe[static_cast<Eigen::Index>(i)], std::min(l, h)), std::max(l, h));
            }
            probe[k] = hi + 0.25 * (hi - lo);
            const double back = foreign_pm_->applyConstraints(probe)[k];
            pm_.setConstraintMode(back < hi ? ConstraintMode::MCMC_REFLECT : ConstraintMode::OPTIMIZATION_CLAMP);
        }
    }
    const int want = pm_.getConstraintMode() == ConstraintMode::MCMC_REFLECT ? SEPAIHRD_CONSTRAINT_REFLECT : SEPAIHRD_CONSTRAINT_CLAMP;
    if (want != device_mode_) { sepaihrd_set_constraint_mode(ctx_, want); device_mode_ = want; }
}

void HipSEPAIHRDObjectiveFunction::calculateBatch(const double* thetas, int B, double* out, int* status) const {
    syncConstraintMode();
    status_.resize(static_cast<size_t>(B)); n_acc_.resize(static_cast<size_t>(B)); n_rej_.resize(static_cast<size_t>(B));
    const int rc = sepaihrd_eval_batch(ctx_, thetas, B, out, status_.data(), n_acc_.data(), n_rej_.data(), nullptr, nullptr);
    if (rc != SEPAIHRD_OK) throw ModelException("SEPAIHRDObjectiveFunction::calculateBatch", sepaihrd_last_error(ctx_));
    int worst = 0;
    for (int b = 0; b < B; ++b) {
        if (status) status[b] = status_[static_cast<size_t>(b)];
        worst = std::max(worst, static_cast<int>(status_[static_cast<size_t>(b)]));
    }
    if (worst >= SEPAIHRD_STATUS_STEP_FAILURE && !status) throwIntegrationFailure(worst);
}

// Per-chain status >= 2: the evaluation produced no value.  The reference lets the solver's exception leave calculate()
// (no try/catch around Simulator::run, SEPAIHRDObjectiveFunction.cpp:165; Dopri5SolverStrategy.cpp:38-42 rethrows
// odeint's as SimulationException) and the samplers' safeEvaluate turns it into -1e18; same exception type here, the
// message says which of the three causes it was.
void HipSEPAIHRDObjectiveFunction::throwIntegrationFailure(int status) {
    switch (status) {
        case SEPAIHRD_STATUS_STEP_FAILURE:
            throw SimulationException("Dopri5SolverStrategy::integrate", "Boost.Odeint integration failed: step size adjustment");
        case SEPAIHRD_STATUS_STEP_BUDGET:
            throw SimulationException("HipSEPAIHRDObjectiveFunction", "integration stopped: step-attempt budget exhausted (SEPAIHRD_STATUS_STEP_BUDGET)");
        case SEPAIHRD_STATUS_PIPELINE:
            throw SimulationException("HipSEPAIHRDObjectiveFunction",
                                      "no value: the hand-off between the integrating wavefront and its likelihood wavefront timed out "
                                      "(SEPAIHRD_STATUS_PIPELINE)");
        default:
            throw SimulationException("HipSEPAIHRDObjectiveFunction", "evaluation failed with per-chain status " + std::to_string(status));
    }
}

double HipSEPAIHRDObjectiveFunction::calculate(const Eigen::VectorXd& parameters) const {
    // cache probe before anything else (SEPAIHRDObjectiveFunction.cpp:63-77)
    auto* fast = dynamic_cast<SimulationCache*>(&cache_);
    size_t fast_key = 0;
    std::string key;
    double cached;
    if (fast) {
        fast_key = fast->computeHash(parameters);
        if (fast->getLikelihood(fast_key, cached)) return cached;
    } else {
        key = cache_.createCacheKey(parameters);
        if (cache_.getLikelihood(key, cached)) return cached;
    }
    if (static_cast<size_t>(parameters.size()) != pm_.getParameterCount())
        return std::numeric_limits<double>::lowest();  // updateModelParameters throws -> lowest() (:117-122)
    double value = 0.0;
    int st = 0;
    calculateBatch(parameters.data(), 1, &value, &st);
    if (st >= SEPAIHRD_STATUS_STEP_FAILURE)  // propagates out of calculate(); samplers map it to -1e18
        throw SimulationException("Dopri5SolverStrategy::integrate", "Boost.Odeint integration failed: step size adjustment");
    if (fast) fast->storeLikelihood(fast_key, value);
    else cache_.storeLikelihood(key, value);
    return value;
}

// ------------------------------------------------------------------ finite-difference gradient objective
HipSEPAIHRDGradientObjectiveFunction::HipSEPAIHRDGradientObjectiveFunction(
    HipSEPAIHRDParameterManager& parameterManager, ISimulationCache& cache, const CalibrationData& data,
    const std::vector<double>& time_points, const Eigen::VectorXd& initial_state,
    std::shared_ptr<IOdeSolverStrategy> solver_strategy, double abs_error, double rel_error, int device,
    bool fma_arithmetic)
    : HipSEPAIHRDObjectiveFunction(parameterManager, cache, data, time_points, initial_state, solver_strategy, abs_error,
                                   rel_error, device, fma_arithmetic),
      initial_state_(initial_state), n_times_(time_points.size()),
      n_obs_rows_(static_cast<size_t>(data.getNewDeaths().rows())), first_time_(time_points.empty() ? 0.0 : time_points.front()) {
    buildGradientContext(data, time_points, solver_strategy, abs_error, rel_error, device, fma_arithmetic);
}

HipSEPAIHRDGradientObjectiveFunction::HipSEPAIHRDGradientObjectiveFunction(
    std::shared_ptr<AgeSEPAIHRDModel> model, IParameterManager& parameterManager, ISimulationCache& cache,
    const CalibrationData& data, const std::vector<double>& time_points, const Eigen::VectorXd& initial_state,
    std::shared_ptr<IOdeSolverStrategy> solver_strategy, double abs_error, double rel_error)
    : HipSEPAIHRDObjectiveFunction(std::move(model), parameterManager, cache, data, time_points, initial_state, solver_strategy,
                                   abs_error, rel_error),
      initial_state_(initial_state), n_times_(time_points.size()),
      n_obs_rows_(static_cast<size_t>(data.getNewDeaths().rows())), first_time_(time_points.empty() ? 0.0 : time_points.front()) {
    buildGradientContext(data, time_points, solver_strategy, abs_error, rel_error, environmentDevice(), environmentFma());
}

void HipSEPAIHRDGradientObjectiveFunction::buildGradientContext(const CalibrationData& data, const std::vector<double>& time_points,
                                                                const std::shared_ptr<IOdeSolverStrategy>& solver_strategy,
                                                                double abs_error, double rel_error, int device, bool fma_arithmetic) {
    const Eigen::VectorXd& initial_state = initial_state_;
    // perturbed runs: multipliers default to 1.0 unless calibrated (:59-72), always the multiplier rule
    const double ones[8] = {1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0};
    const ConstraintMode keep = pm_.getConstraintMode();
    pm_.setConstraintMode(ConstraintMode::OPTIMIZATION_CLAMP);  // the temporary manager's default mode (:40-43)
    try {
        grad_ctx_ = createContext(pm_, data, time_points, initial_state, solver_strategy, abs_error, rel_error, device,
                                  fma_arithmetic, ones);
    } catch (...) {
        pm_.setConstraintMode(keep);
        throw;
    }
    pm_.setConstraintMode(keep);
    sepaihrd_set_initial_state_mode(grad_ctx_, SEPAIHRD_INIT_MULTIPLIERS);
    const std::vector<int32_t>& codes = pm_.fieldCodes();
    mult_index_.assign(8, -1);
    for (size_t k = 0; k < codes.size(); ++k)
        if (codes[k] >= SEPAIHRD_F_E0_MULT && codes[k] <= SEPAIHRD_F_D0_MULT && mult_index_[static_cast<size_t>(codes[k] - SEPAIHRD_F_E0_MULT)] < 0)
            mult_index_[static_cast<size_t>(codes[k] - SEPAIHRD_F_E0_MULT)] = static_cast<int>(k);
}

HipSEPAIHRDGradientObjectiveFunction::~HipSEPAIHRDGradientObjectiveFunction() { sepaihrd_destroy(grad_ctx_); }

bool HipSEPAIHRDGradientObjectiveFunction::initialStateValid(const double* plus) const {
    const Eigen::VectorXd& N = pm_.modelParameters().N;
    const int n = static_cast<int>(N.size());
    for (int k = 0; k < n; ++k) {
        double sum = 0.0;
        for (int j = 1; j <= 8; ++j) {  // E .. D: the population compartments besides S (:88-92)
            const int idx = mult_index_[static_cast<size_t>(j - 1)];
            const double mult = idx >= 0 ? plus[idx] : 1.0;
            sum += initial_state_[j * n + k] * mult;
        }
        if (sum > N[k] || sum < 0) return false;
    }
    return true;
}

double HipSEPAIHRDGradientObjectiveFunction::evaluate_with_gradient(const Eigen::VectorXd& params, Eigen::VectorXd& grad) const {
    const double LOWEST = std::numeric_limits<double>::lowest();
    const int P = static_cast<int>(params.size());
    grad.resize(P);
    std::vector<double> plus(static_cast<size_t>(P) * P), eps(static_cast<size_t>(P));
    for (int i = 0; i < P; ++i) {
        const double param_scale = std::max(std::abs(params[i]), epsilon_);
        eps[static_cast<size_t>(i)] = epsilon_ * param_scale;
        for (int k = 0; k < P; ++k) plus[static_cast<size_t>(i) * P + k] = params[k];
        plus[static_cast<size_t>(i) * P + i] += eps[static_cast<size_t>(i)];
    }
    std::vector<double> f_plus(static_cast<size_t>(P), LOWEST);
    std::vector<int32_t> status(static_cast<size_t>(P), 0);
    const bool rows_match = n_times_ == n_obs_rows_;
    if (rows_match && first_time_ < 0.0)
        throw InvalidParameterException("SEPAIHRDGradientObjectiveFunction",
                                        "output grids that start before t = 0 with one observation row per output are not built");
    // The P perturbed simulations do not depend on the centre value: they are launched first, on their context's own
    // stream, and run while calculate() evaluates the centre (the reference's order of the two is not observable).
    if (rows_match && sepaihrd_eval_batch_begin(grad_ctx_, plus.data(), P) != SEPAIHRD_OK)
        throw ModelException("SEPAIHRDGradientObjectiveFunction", sepaihrd_last_error(grad_ctx_));
    double f_center;
    try {
        f_center = HipSEPAIHRDObjectiveFunction::calculate(params);  // :22
    } catch (...) {
        if (rows_match) (void)sepaihrd_eval_batch_end(grad_ctx_, nullptr, nullptr, nullptr, nullptr, nullptr);
        throw;
    }
    if (rows_match && sepaihrd_eval_batch_end(grad_ctx_, f_plus.data(), status.data(), nullptr, nullptr, nullptr) != SEPAIHRD_OK)
        throw ModelException("SEPAIHRDGradientObjectiveFunction", sepaihrd_last_error(grad_ctx_));
    if (!std::isfinite(f_center)) {  // :24-29
        for (int i = 0; i < P; ++i) grad[i] = 0.0;
        return f_center;
    }
    for (int i = 0; i < P; ++i) {
        const size_t u = static_cast<size_t>(i);
        const bool valid = initialStateValid(&plus[u * P]);
        if (!valid) { grad[i] = 0.0; continue; }  // :94-104
        double fp;
        if (!rows_match) {
            // every stream fails calculateSingleLogLikelihood's dimension check: lowest() three times,
            // -inf, then mapped to lowest() (:148-155); the simulation result is not used
            fp = LOWEST;
        } else {
            if (status[u] >= SEPAIHRD_STATUS_STEP_FAILURE) throwIntegrationFailure(status[u]);
            fp = f_plus[u];  // status 1 with a valid state: non-finite likelihood, already lowest()
        }
        grad[i] = std::isfinite(fp) ? (fp - f_center) / eps[u] : 0.0;  // :163-167
    }
    return f_center;
}

}  // namespace epidemic
