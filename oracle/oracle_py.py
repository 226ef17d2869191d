"""ctypes binding of the CPU oracle (oracle/liboracle.so) -- TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle.so")
_dp = C.POINTER(C.c_double)


class oracle_problem(C.Structure):
    _fields_ = [
        ("n", C.c_int32), ("n_times", C.c_int32), ("n_obs", C.c_int32), ("n_beta", C.c_int32),
        ("n_kappa", C.c_int32), ("n_params", C.c_int32), ("solver", C.c_int32), ("constraint_mode", C.c_int32),
        ("times", _dp), ("N", _dp), ("M", _dp),
        ("a", _dp), ("h_infec", _dp), ("p", _dp), ("h", _dp), ("icu", _dp), ("d_H", _dp), ("d_ICU", _dp),
        ("d_community", _dp),
        ("beta_end_times", _dp), ("beta_values", _dp), ("kappa_end_times", _dp), ("kappa_values", _dp),
        ("initial_state", _dp), ("obs_H", _dp), ("obs_ICU", _dp), ("obs_D", _dp),
        ("lower", _dp), ("upper", _dp), ("sigmas", _dp),
        ("param_names", C.c_char_p), ("npi_names", C.c_char_p),
        ("beta", C.c_double), ("theta", C.c_double), ("sigma", C.c_double), ("gamma_p", C.c_double),
        ("gamma_A", C.c_double), ("gamma_I", C.c_double), ("gamma_H", C.c_double), ("gamma_ICU", C.c_double),
        ("multipliers", C.c_double * 8), ("runup_days", C.c_double), ("seed_exposed", C.c_double),
        ("abs_err", C.c_double), ("rel_err", C.c_double), ("dt_hint", C.c_double),
    ]


_lib = None


def build(native: bool = False) -> None:
    import subprocess
    target = "liboracle_native.so" if native else "liboracle.so"
    subprocess.run(["make", "-C", _HERE, target], check=True, capture_output=True)


def load(path: str | None = None) -> C.CDLL:
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        build(native=p.endswith("_native.so"))
    lib = C.CDLL(p)
    vp = C.c_void_p
    lib.oracle_create.restype = vp
    lib.oracle_create.argtypes = [C.POINTER(oracle_problem), C.c_char_p, C.c_int]
    lib.oracle_destroy.argtypes = [vp]
    lib.oracle_set_constraint_mode.argtypes = [vp, C.c_int]
    lib.oracle_set_max_attempts.argtypes = [vp, C.c_long]
    lib.oracle_eval_batch.argtypes = [vp, vp, C.c_int, vp, vp, vp, vp, vp, vp, C.c_int]
    lib.oracle_rhs.argtypes = [vp, vp, vp, C.c_double, vp]
    lib.oracle_beta_kappa.restype = C.c_double
    lib.oracle_beta_kappa.argtypes = [vp, vp, C.c_double, vp, vp]
    lib.oracle_poisson_loglik.restype = C.c_double
    lib.oracle_poisson_loglik.argtypes = [vp, vp, C.c_int, C.c_int]
    lib.oracle_apply_constraints.argtypes = [vp, C.c_int, vp, vp]
    lib.oracle_jitter_draws.argtypes = [vp, C.c_int, vp, C.c_uint32, C.c_int, vp]
    lib.oracle_cache_hash.restype = C.c_uint64
    lib.oracle_cache_hash.argtypes = [vp, C.c_int]
    lib.oracle_std_normals.argtypes = [C.c_uint32, C.c_int, vp]
    lib.oracle_mh.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, vp,
                              C.c_uint32, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_int]
    lib.oracle_num_threads.restype = C.c_int
    lib.oracle_hc.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, C.c_uint32, vp, vp, vp, vp, vp]
    lib.oracle_pso.argtypes = [vp, vp, vp, C.c_uint32, vp, vp, vp, vp, vp]
    lib.oracle_gradient.argtypes = [vp, vp, C.c_double, vp, vp]
    lib.oracle_nuts.argtypes = [vp, C.c_int, C.c_int, C.c_double, C.c_int, C.c_double, C.c_int, vp, C.c_uint32, vp, vp, vp, vp,
                                vp, vp, vp]
    lib.oracle_calibrate.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_uint32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.oracle_condition_covariance.argtypes = [vp, vp, vp]
    lib.oracle_sir_rhs.restype = None
    lib.oracle_sir_rhs.argtypes = [C.c_int, vp, vp, vp, C.c_double, C.c_double, vp, vp]
    lib.oracle_sir_simulate.argtypes = [C.c_int, vp, vp, vp, C.c_double, C.c_double, vp, vp, C.c_int, C.c_double,
                                        C.c_double, vp, vp, vp]
    lib.oracle_model_parameters.argtypes = [vp, vp, vp]
    lib.oracle_simulate_samples.argtypes = [vp, vp, C.c_int, vp, vp, C.c_int]
    lib.oracle_ppc_select.argtypes = [C.c_int, C.c_int, C.c_uint32, vp]
    lib.oracle_ensemble.argtypes = [vp, vp, C.c_int, vp, C.c_int, vp, vp, vp, vp, C.c_int]
    if path is None:
        _lib = lib
    return lib


class Oracle:
    """CPU restatement of SEPAIHRDObjectiveFunction for a SEPAIHRDProblem (the checker)."""

    def __init__(self, pb, lib_path: str | None = None):
        self.lib = load(lib_path)
        self.pb = pb
        self._keep = []
        s = oracle_problem()
        s.n, s.n_times, s.n_obs = pb.n, pb.n_times, pb.n_obs
        s.n_beta, s.n_kappa, s.n_params = len(pb.beta_values), len(pb.kappa_values), pb.n_params
        s.solver, s.constraint_mode = pb.solver, pb.constraint_mode

        def dbl(x):
            a = np.ascontiguousarray(x, dtype=np.float64)
            if a.size == 0:
                a = np.zeros(1)
            self._keep.append(a)
            return a.ctypes.data_as(_dp)

        s.times, s.N = dbl(pb.times), dbl(pb.N)
        s.M = dbl(np.asarray(pb.M).ravel(order="F"))
        for name in ("a", "h_infec", "p", "h", "icu", "d_H", "d_ICU", "d_community", "beta_end_times",
                     "beta_values", "kappa_end_times", "kappa_values", "initial_state"):
            setattr(s, name, dbl(getattr(pb, name)))
        s.obs_H, s.obs_ICU, s.obs_D = dbl(pb.obs_H), dbl(pb.obs_ICU), dbl(pb.obs_D)
        lo, hi, has = pb.bounds_arrays()
        lo = np.where(has.astype(bool), lo, np.nan)
        s.lower, s.upper, s.sigmas = dbl(lo), dbl(hi), dbl(pb.sigma_array())
        s.param_names = "\n".join(pb.param_names).encode()
        s.npi_names = "\n".join(pb.npi_names).encode()
        s.beta, s.theta, s.sigma, s.gamma_p = pb.beta, pb.theta, pb.sigma, pb.gamma_p
        s.gamma_A, s.gamma_I, s.gamma_H, s.gamma_ICU = pb.gamma_A, pb.gamma_I, pb.gamma_H, pb.gamma_ICU
        for i in range(8):
            s.multipliers[i] = float(pb.multipliers[i])
        s.runup_days, s.seed_exposed = pb.runup_days, pb.seed_exposed
        s.abs_err, s.rel_err, s.dt_hint = pb.abs_err, pb.rel_err, pb.dt_hint
        err = C.create_string_buffer(512)
        self.h = self.lib.oracle_create(C.byref(s), err, len(err))
        if not self.h:
            raise RuntimeError("oracle_create failed: " + err.value.decode())
        self.P, self.n, self.T = pb.n_params, pb.n, pb.n_times
        self.times = np.asarray(pb.times, dtype=np.float64)

    def __del__(self):
        try:
            if self.h:
                self.lib.oracle_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def set_constraint_mode(self, mode: int):
        self.lib.oracle_set_constraint_mode(self.h, mode)

    def set_max_attempts(self, n: int):
        self.lib.oracle_set_max_attempts(self.h, n)

    def eval_batch(self, theta, want_traj: bool = False, nthreads: int = 0) -> dict:
        th = np.ascontiguousarray(np.atleast_2d(theta), dtype=np.float64)
        assert th.shape[1] == self.P
        B = th.shape[0]
        out = {"loglik": np.empty(B), "status": np.empty(B, dtype=np.int32),
               "n_accept": np.empty(B, dtype=np.int32), "n_reject": np.empty(B, dtype=np.int32),
               "ll_parts": np.empty((B, 3))}
        traj = np.empty((B, self.T, 11 * self.n)) if want_traj else None
        if nthreads <= 0:
            nthreads = self.lib.oracle_num_threads()
        self.lib.oracle_eval_batch(self.h, th.ctypes.data, B, out["loglik"].ctypes.data,
                                   out["status"].ctypes.data, out["n_accept"].ctypes.data,
                                   out["n_reject"].ctypes.data, out["ll_parts"].ctypes.data,
                                   traj.ctypes.data if want_traj else None, nthreads)
        if want_traj:
            out["traj"] = traj
        return out

    def ensemble_quantiles(self, theta, probs, nthreads: int = 0) -> dict:
        """Posterior-ensemble summaries from the problem's initial state as given (fixed-state runs)."""
        th = np.ascontiguousarray(np.atleast_2d(theta), dtype=np.float64)
        pr = np.ascontiguousarray(probs, dtype=np.float64)
        S, npb = th.shape[0], pr.size
        Tp = int(np.sum(np.asarray(self.times) >= 0.0))
        ppc = np.empty((6, npb, Tp, self.n))
        sero = np.empty((npb, self.T))
        status = np.empty(S, dtype=np.int32)
        nv = C.c_int32(0)
        if nthreads <= 0:
            nthreads = self.lib.oracle_num_threads()
        tp = self.lib.oracle_ensemble(self.h, th.ctypes.data, S, pr.ctypes.data, npb, ppc.ctypes.data,
                                      sero.ctypes.data, status.ctypes.data, C.byref(nv), nthreads)
        assert tp == Tp
        return {"ppc": ppc, "sero": sero, "status": status, "n_valid": nv.value}

    def hill_climbing(self, x0, seed: int, iterations: int, cloud_size_multiplier: int = 8, threads: int = 1) -> dict:
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        best = np.empty(self.P)
        cov = np.empty((self.P, self.P))
        trace = np.empty(iterations)
        bv = C.c_double(0.0)
        ne = C.c_long(0)
        self.lib.oracle_hc(self.h, iterations, cloud_size_multiplier, threads, x0.ctypes.data, seed, best.ctypes.data,
                           C.byref(bv), cov.ctypes.data, trace.ctypes.data, C.byref(ne))
        return {"best": best, "best_value": bv.value, "final_cov": cov, "trace": trace, "evaluations": ne.value}

    PSO_KEYS = ("iterations", "swarm_size", "max_stagnation", "omega_start", "omega_end", "c1_initial", "c1_final",
                "c2_initial", "c2_final", "variant", "topology", "use_opposition_learning", "use_adaptive_parameters",
                "restart_threshold", "quantum_beta", "levy_alpha", "deferred_personal_bests")
    PSO_DEFAULTS = (100, 30, 50, 0.9, 0.4, 2.5, 0.5, 0.5, 2.5, 0, 0, 0, 0, 1e-6, 1.0, 1.5, 0)

    def particle_swarm(self, x0, seed: int, **settings) -> dict:
        """ParticleSwarmOptimization restated; settings as in pso_settings.txt (+ deferred_personal_bests)."""
        cfg = dict(zip(self.PSO_KEYS, self.PSO_DEFAULTS))
        unknown = set(settings) - set(cfg)
        if unknown:
            raise KeyError(f"unknown PSO settings: {sorted(unknown)}")
        cfg.update(settings)
        vals = np.array([float(cfg[k]) for k in self.PSO_KEYS])
        x0p = None
        if x0 is not None:
            x0 = np.ascontiguousarray(x0, dtype=np.float64)
            x0p = x0.ctypes.data
        iters = int(cfg["iterations"])
        best = np.empty(self.P)
        cov = np.empty((self.P, self.P))
        trace = np.empty(iters)
        bv = C.c_double(0.0)
        ne = C.c_long(0)
        rc = self.lib.oracle_pso(self.h, vals.ctypes.data, x0p, seed, best.ctypes.data, C.byref(bv), cov.ctypes.data,
                                 trace.ctypes.data, C.byref(ne))
        if rc != 0:
            raise RuntimeError("SimulationException")
        return {"best": best, "best_value": bv.value, "final_cov": cov, "trace": trace, "evaluations": ne.value}

    def evaluate_with_gradient(self, theta, epsilon: float = 1e-4):
        th = np.ascontiguousarray(theta, dtype=np.float64)
        g = np.empty(self.P)
        v = C.c_double(0.0)
        rc = self.lib.oracle_gradient(self.h, th.ctypes.data, epsilon, C.byref(v), g.ctypes.data)
        if rc != 0:
            raise RuntimeError("SimulationException")
        return v.value, g

    def nuts(self, theta0, seed: int, iterations: int, adaptation_window: int, max_tree_depth: int = 10,
             delta_target: float = 0.8, fd_epsilon: float = 1e-4, constraint_mode: int = 1) -> dict:
        """NUTSSampler restated over the finite-difference gradient objective."""
        th = np.ascontiguousarray(theta0, dtype=np.float64)
        samples = np.empty((iterations, self.P))
        values, eps = np.empty(iterations), np.empty(iterations)
        depth = np.empty(iterations, dtype=np.int32)
        best = np.empty(self.P)
        bv = C.c_double(0.0)
        ng = C.c_long(0)
        ns = self.lib.oracle_nuts(self.h, iterations, adaptation_window, delta_target, max_tree_depth, fd_epsilon,
                                  constraint_mode, th.ctypes.data, seed, samples.ctypes.data, values.ctypes.data,
                                  eps.ctypes.data, depth.ctypes.data, best.ctypes.data, C.byref(bv), C.byref(ng))
        if ns < 0:
            raise RuntimeError("SimulationException")
        return {"samples": samples[:ns], "sample_values": values[:ns], "epsilon_trace": eps[:ns], "depth_trace": depth[:ns],
                "best": best, "best_value": bv.value, "gradient_calls": ng.value}

    def model_parameters(self, theta) -> dict:
        """Model fields after updateModelParameters(theta) (SEPAIHRDParameterManager.cpp:164-287)."""
        th = np.ascontiguousarray(theta, dtype=np.float64)
        n, nb, nk = self.n, len(self.pb.beta_values), len(self.pb.kappa_values)
        buf = np.empty(8 + 8 * n + nb + nk)
        k = self.lib.oracle_model_parameters(self.h, th.ctypes.data, buf.ctypes.data)
        if k < 0:
            raise ValueError("updateModelParameters throws for this theta")
        out = dict(zip(("beta", "theta", "sigma", "gamma_p", "gamma_A", "gamma_I", "gamma_H", "gamma_ICU"), buf[:8]))
        o = 8
        for name in ("a", "h_infec", "p", "h", "icu", "d_H", "d_ICU", "d_community"):
            out[name] = buf[o:o + n].copy(); o += n
        out["beta_values"] = buf[o:o + nb].copy(); o += nb
        out["kappa_values"] = buf[o:o + nk].copy()
        return out

    def simulate_samples(self, theta, nthreads: int = 0) -> dict:
        """Trajectories from the problem's initial state as given (fixed-state ensemble runs)."""
        th = np.ascontiguousarray(np.atleast_2d(theta), dtype=np.float64)
        S = th.shape[0]
        traj = np.zeros((S, self.T, 11 * self.n))
        status = np.empty(S, dtype=np.int32)
        if nthreads <= 0:
            nthreads = self.lib.oracle_num_threads()
        self.lib.oracle_simulate_samples(self.h, th.ctypes.data, S, traj.ctypes.data, status.ctypes.data, nthreads)
        return {"traj": traj, "status": status}

    def condition_covariance(self, cov) -> np.ndarray:
        c = np.ascontiguousarray(cov, dtype=np.float64)
        out = np.empty_like(c)
        self.lib.oracle_condition_covariance(self.h, c.ctypes.data, out.ctypes.data)
        return out

    def calibrate(self, x0, hc_seed: int, mh_seed: int, hc_iterations: int, mh_iterations: int, burn_in: int,
                  cloud_size_multiplier: int = 8, threads: int = 1, adaptation_period: int = 100,
                  thinning: int = 1) -> dict:
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        cap = mh_iterations // max(1, thinning) + 1
        out = {"best": np.empty(self.P), "phase2_cov": np.empty((self.P, self.P)),
               "accept_trace": np.empty(mh_iterations - 1, dtype=np.uint8), "samples": np.empty((cap, self.P)),
               "sample_values": np.empty(cap), "mcmc_objective_values": np.empty(cap)}
        bv, iv, p1 = C.c_double(0.0), C.c_double(0.0), C.c_double(0.0)
        ns = C.c_int32(0)
        self.lib.oracle_calibrate(self.h, hc_iterations, cloud_size_multiplier, threads, hc_seed, mh_iterations, burn_in,
                                  adaptation_period, thinning, mh_seed, x0.ctypes.data, out["best"].ctypes.data,
                                  C.byref(bv), C.byref(iv), C.byref(p1), out["phase2_cov"].ctypes.data,
                                  out["accept_trace"].ctypes.data, out["samples"].ctypes.data,
                                  out["sample_values"].ctypes.data, out["mcmc_objective_values"].ctypes.data,
                                  C.byref(ns))
        n = ns.value
        out.update(best_value=bv.value, initial_value=iv.value, phase1_best_value=p1.value, n_samples=n)
        for k in ("samples", "sample_values", "mcmc_objective_values"):
            out[k] = out[k][:n]
        return out

    def calculate(self, theta) -> float:
        return float(self.eval_batch(theta, nthreads=1)["loglik"][0])

    def rhs(self, x, t, theta=None) -> np.ndarray:
        x = np.ascontiguousarray(x, dtype=np.float64)
        dx = np.empty_like(x)
        th = None if theta is None else np.ascontiguousarray(theta, dtype=np.float64)
        rc = self.lib.oracle_rhs(self.h, th.ctypes.data if th is not None else None, x.ctypes.data, float(t),
                                 dx.ctypes.data)
        if rc:
            raise RuntimeError("updateModelParameters threw")
        return dx

    def beta_kappa(self, t, theta=None):
        b, k = C.c_double(), C.c_double()
        th = None if theta is None else np.ascontiguousarray(theta, dtype=np.float64)
        self.lib.oracle_beta_kappa(self.h, th.ctypes.data if th is not None else None, float(t), C.byref(b),
                                   C.byref(k))
        return b.value, k.value

    def apply_constraints(self, theta, mode: int) -> np.ndarray:
        th = np.ascontiguousarray(np.atleast_2d(theta), dtype=np.float64)
        out = np.empty_like(th)
        for b in range(th.shape[0]):
            self.lib.oracle_apply_constraints(self.h, mode, th[b].ctypes.data, out[b].ctypes.data)
        return out

    def jitter_draws(self, base, seed0: int, B: int, mode: int = 1) -> np.ndarray:
        base = np.ascontiguousarray(base, dtype=np.float64)
        out = np.empty((B, self.P))
        self.lib.oracle_jitter_draws(self.h, mode, base.ctypes.data, seed0, B, out.ctypes.data)
        return out

    def metropolis_hastings(self, x0, seed: int, iterations: int, burn_in: int, adaptation_period: int = 100,
                            thinning: int = 1, reg_eps: float = 1e-6, target_acc: float = 0.234,
                            adapt_scale: bool = True, two_pass_covariance: bool = False) -> dict:
        P = self.P
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        cap = iterations // max(1, thinning) + 2
        best = np.empty(P)
        best_value, final_scale = C.c_double(), C.c_double()
        accepted, n_samples = C.c_int32(), C.c_int32()
        trace = np.zeros(max(iterations - 1, 1), dtype=np.uint8)
        samples = np.empty((cap, P))
        values = np.empty(cap)
        cov = np.empty((P, P))
        self.lib.oracle_mh(self.h, iterations, burn_in, adaptation_period, thinning, reg_eps, target_acc,
                           int(adapt_scale), x0.ctypes.data, seed, best.ctypes.data, C.byref(best_value),
                           C.byref(accepted), C.byref(final_scale), trace.ctypes.data, samples.ctypes.data,
                           values.ctypes.data, C.byref(n_samples), cov.ctypes.data, int(two_pass_covariance))
        ns = n_samples.value
        return {"best": best, "best_value": best_value.value, "accepted": accepted.value,
                "final_scale": final_scale.value, "accept_trace": trace[:iterations - 1],
                "samples": samples[:ns], "sample_values": values[:ns], "final_cov": cov}


def sir_rhs(N, Cm, gamma, q, scale_C, state) -> np.ndarray:
    """AgeSIRModel::computeDerivatives (BASELINE config 0 plumbing)."""
    N, Cm, gamma, state = (np.ascontiguousarray(a, dtype=np.float64) for a in (N, Cm, gamma, state))
    out = np.empty_like(state)
    load().oracle_sir_rhs(len(N), N.ctypes.data, Cm.ctypes.data, gamma.ctypes.data, q, scale_C, state.ctypes.data,
                          out.ctypes.data)
    return out


def sir_simulate(N, Cm, gamma, q, scale_C, init, times, abs_err=1e-6, rel_err=1e-6) -> dict:
    N, Cm, gamma, init, times = (np.ascontiguousarray(a, dtype=np.float64) for a in (N, Cm, gamma, init, times))
    traj = np.empty((len(times), init.size))
    na, nr = C.c_int32(0), C.c_int32(0)
    rc = load().oracle_sir_simulate(len(N), N.ctypes.data, Cm.ctypes.data, gamma.ctypes.data, q, scale_C, init.ctypes.data,
                                    times.ctypes.data, len(times), abs_err, rel_err, traj.ctypes.data, C.byref(na),
                                    C.byref(nr))
    if rc != 0:
        raise RuntimeError("sir_simulate failed")
    return {"traj": traj, "n_accept": na.value, "n_reject": nr.value}


def ppc_select(n_samples: int, num_for_ppc: int, seed: int) -> np.ndarray:
    out = np.empty(max(n_samples, num_for_ppc, 1), dtype=np.int32)
    k = load().oracle_ppc_select(n_samples, num_for_ppc, seed, out.ctypes.data)
    return out[:k].copy()


def poisson_loglik(sim, obs) -> float:
    lib = load()
    sim = np.ascontiguousarray(sim, dtype=np.float64)
    obs = np.ascontiguousarray(obs, dtype=np.float64)
    return lib.oracle_poisson_loglik(sim.ctypes.data, obs.ctypes.data, sim.shape[0], sim.shape[1])


def cache_hash(p) -> int:
    p = np.ascontiguousarray(p, dtype=np.float64)
    return load().oracle_cache_hash(p.ctypes.data, p.size)


def std_normals(seed: int, count: int) -> np.ndarray:
    out = np.empty(count)
    load().oracle_std_normals(seed, count, out.ctypes.data)
    return out
