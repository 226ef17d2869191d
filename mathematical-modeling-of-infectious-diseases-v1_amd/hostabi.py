"""ctypes view of the C++ host mirror (host/, libsepaihrd_host.so) -- test plumbing.

The C++ classes (HipSEPAIHRDObjectiveFunction, HipSEPAIHRDParameterManager, SimulationCache,
MultiChainMetropolisHastings) are what a reference maintainer links against; the flat functions
of host/src/host_capi.cpp only exist so that the Python test-suite can drive them.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import hipabi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsepaihrd_host.so")
_lib = None


def load_library() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    hipabi.load_library()  # torch-first HIP runtime + libsepaihrd_hip.so, then the host library
    path = os.environ.get("SEPAIHRD_HOST_LIB") or LIB_PATH  # env override: experiment builds only
    if not os.path.exists(path):
        raise FileNotFoundError(f"{path} not found: run __graft_entry__.build()")
    lib = C.CDLL(path)
    vp = C.c_void_p
    lib.host_last_error.restype = C.c_char_p
    lib.host_last_mh_loop_seconds.restype = C.c_double
    lib.host_objective_create.restype = vp
    lib.host_objective_create.argtypes = [C.POINTER(hipabi.sepaihrd_problem), C.c_char_p, C.c_char_p, vp, C.c_int,
                                          C.c_int, C.c_int]
    lib.host_objective_destroy.argtypes = [vp]
    lib.host_model_holders.argtypes = [vp, vp, C.c_int, C.c_double, C.c_double, vp, C.c_int, vp, vp, vp, C.POINTER(C.c_int),
                                       C.c_char_p, C.c_int]
    lib.host_reference_constructors.argtypes = [C.POINTER(hipabi.sepaihrd_problem), C.c_char_p, C.c_char_p, vp, vp, C.c_int, vp, vp]
    lib.host_objective_calculate.argtypes = [vp, vp, C.POINTER(C.c_double)]
    lib.host_objective_calculate_batch.argtypes = [vp, vp, C.c_int, vp, vp]
    lib.host_cache_stats.argtypes = [vp, C.POINTER(C.c_long), C.POINTER(C.c_long), C.POINTER(C.c_long)]
    lib.host_apply_constraints.argtypes = [vp, C.c_int, vp, vp]
    lib.host_current_parameters.argtypes = [vp, vp]
    lib.host_mh_run.argtypes = [vp, C.c_int, vp, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double,
                                C.c_double, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, C.c_int, vp, C.c_int, C.c_int]
    lib.host_mh_run_groups.argtypes = [vp, C.c_int, C.c_int, vp, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp]
    lib.host_gradient.argtypes = [vp, C.POINTER(hipabi.sepaihrd_problem), C.c_int, vp, C.c_double, vp, vp]
    lib.host_calibrate.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_uint32, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.host_hc_run.argtypes = [vp, vp, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp]
    lib.host_calibrate_pso.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_int,
                                       vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.host_nuts_run.argtypes = [vp, C.POINTER(hipabi.sepaihrd_problem), C.c_int, C.c_int, C.c_int, C.c_double, C.c_int,
                                  C.c_double, C.c_int, vp, C.c_uint32, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.host_pso_run.argtypes = [vp, vp, vp, vp, C.c_int, vp, vp, vp, vp, vp, vp]
    lib.host_ensemble.argtypes = [vp, C.POINTER(hipabi.sepaihrd_problem), C.c_int, vp, C.c_int, C.c_int, C.c_uint32,
                                  vp, vp, vp, vp, C.c_int, C.c_int, vp, vp]
    _lib = lib
    return lib


class HostObjective:
    """HipSEPAIHRDObjectiveFunction + its parameter manager + a SimulationCache (C++ objects)."""

    def __init__(self, pb, device: int = -1, cache_capacity: int = 1000, with_objective: bool = True):
        self.lib = load_library()
        self.pb = pb
        keep: list = []
        st = hipabi.build_problem_struct(pb, keep)
        sig = np.ascontiguousarray(pb.sigma_array())
        self.h = self.lib.host_objective_create(C.byref(st), "\n".join(pb.param_names).encode(),
                                                "\n".join(pb.npi_names).encode(), sig.ctypes.data, device,
                                                cache_capacity, int(with_objective))
        if not self.h:
            raise RuntimeError("host_objective_create failed: " + self.lib.host_last_error().decode())
        self.P = pb.n_params

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.host_objective_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def hill_climbing(self, x0, seed: int, iterations: int, cloud_size_multiplier: int = 8, threads: int = 16,
                      use_scalar_interface: bool = False) -> dict:
        """BatchedHillClimbingOptimizer::optimize in OPTIMIZATION_CLAMP mode (calibration phase 1)."""
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        best = np.empty(self.P)
        cov = np.empty((self.P, self.P))
        trace = np.empty(iterations)
        bv = C.c_double(0.0)
        ne, nl = C.c_long(0), C.c_long(0)
        rc = self.lib.host_hc_run(self.h, x0.ctypes.data, seed, threads, iterations, cloud_size_multiplier,
                                  int(use_scalar_interface), best.ctypes.data, C.byref(bv), cov.ctypes.data,
                                  trace.ctypes.data, C.byref(ne), C.byref(nl))
        if rc != 0:
            raise RuntimeError("host_hc_run: " + self.lib.host_last_error().decode())
        return {"best": best, "best_value": bv.value, "final_cov": cov, "trace": trace,
                "evaluations": ne.value, "launches": nl.value}

    def particle_swarm(self, x0, seed: int, **settings) -> dict:
        """BatchedParticleSwarmOptimization::optimize in OPTIMIZATION_CLAMP mode; settings as in pso_settings.txt."""
        settings = dict(settings, seed=seed)
        keys = (C.c_char_p * len(settings))(*[k.encode() for k in settings])
        vals = np.array([float(v) for v in settings.values()])
        x0p = None
        if x0 is not None:
            x0 = np.ascontiguousarray(x0, dtype=np.float64)
            x0p = x0.ctypes.data
        iters = int(settings.get("iterations", 100))
        best = np.empty(self.P)
        cov = np.empty((self.P, self.P))
        trace = np.empty(iters)
        bv = C.c_double(0.0)
        ne, nl = C.c_long(0), C.c_long(0)
        rc = self.lib.host_pso_run(self.h, x0p, keys, vals.ctypes.data, len(settings), best.ctypes.data, C.byref(bv),
                                   cov.ctypes.data, trace.ctypes.data, C.byref(ne), C.byref(nl))
        if rc != 0:
            raise RuntimeError("host_pso_run: " + self.lib.host_last_error().decode())
        return {"best": best, "best_value": bv.value, "final_cov": cov, "trace": trace,
                "evaluations": ne.value, "launches": nl.value}

    def evaluate_with_gradient(self, theta, epsilon: float = 1e-4, device: int = -1):
        """HipSEPAIHRDGradientObjectiveFunction::evaluate_with_gradient: (value, gradient)."""
        th = np.ascontiguousarray(theta, dtype=np.float64)
        keep: list = []
        st = hipabi.build_problem_struct(self.pb, keep)
        g = np.empty(self.P)
        v = C.c_double(0.0)
        rc = self.lib.host_gradient(self.h, C.byref(st), device, th.ctypes.data, epsilon, C.byref(v), g.ctypes.data)
        if rc != 0:
            raise RuntimeError("host_gradient: " + self.lib.host_last_error().decode())
        return v.value, g

    def nuts(self, theta0, seed: int, iterations: int, adaptation_window: int, max_tree_depth: int = 10,
             delta_target: float = 0.8, fd_epsilon: float = 1e-4, constraint_mode: int = 1, device: int = -1) -> dict:
        """HipNUTSSampler::optimize over a HipSEPAIHRDGradientObjectiveFunction (SEPAIHRDModelCalibration::runNUTS)."""
        th = np.ascontiguousarray(theta0, dtype=np.float64)
        keep: list = []
        st = hipabi.build_problem_struct(self.pb, keep)
        samples = np.empty((iterations, self.P))
        values, eps = np.empty(iterations), np.empty(iterations)
        depth = np.empty(iterations, dtype=np.int32)
        best = np.empty(self.P)
        bv = C.c_double(0.0)
        nc, nl = C.c_long(0), C.c_long(0)
        ns = self.lib.host_nuts_run(self.h, C.byref(st), device, iterations, adaptation_window, delta_target,
                                    max_tree_depth, fd_epsilon, constraint_mode, th.ctypes.data, seed,
                                    samples.ctypes.data, values.ctypes.data, eps.ctypes.data, depth.ctypes.data,
                                    best.ctypes.data, C.byref(bv), C.byref(nc), C.byref(nl))
        if ns < 0:
            raise RuntimeError("host_nuts_run: " + self.lib.host_last_error().decode())
        return {"samples": samples[:ns], "sample_values": values[:ns], "epsilon_trace": eps[:ns], "depth_trace": depth[:ns],
                "best": best, "best_value": bv.value, "gradient_calls": nc.value, "gradient_launches": nl.value}

    def calibrate(self, hc_seed: int, mh_seed: int, hc_iterations: int, mh_iterations: int, burn_in: int,
                  cloud_size_multiplier: int = 8, threads: int = 16, adaptation_period: int = 100, thinning: int = 1,
                  chains: int = 1) -> dict:
        """HipModelCalibrator: HC (clamp) -> covariance conditioning -> `chains` MH chains (reflect)."""
        cap = 1 + (mh_iterations - 1) // max(1, thinning)  # t = 0 and every thinning-th iteration after it
        out = {"best": np.empty(self.P), "phase2_cov": np.empty((self.P, self.P)),
               "accept_trace": np.empty((chains, mh_iterations - 1), dtype=np.uint8),
               "samples": np.empty((chains, cap, self.P)), "sample_values": np.empty((chains, cap)),
               "mcmc_objective_values": np.empty((chains, cap))}
        bv, iv, p1 = C.c_double(0.0), C.c_double(0.0), C.c_double(0.0)
        ns = C.c_int32(0)
        rc = self.lib.host_calibrate(self.h, hc_iterations, cloud_size_multiplier, threads, hc_seed, mh_iterations,
                                     burn_in, adaptation_period, thinning, mh_seed, chains, out["best"].ctypes.data,
                                     C.byref(bv), C.byref(iv), C.byref(p1), out["phase2_cov"].ctypes.data,
                                     out["accept_trace"].ctypes.data, out["samples"].ctypes.data,
                                     out["sample_values"].ctypes.data, out["mcmc_objective_values"].ctypes.data,
                                     C.byref(ns))
        if rc != 0:
            raise RuntimeError("host_calibrate: " + self.lib.host_last_error().decode())
        n = ns.value
        assert n == cap, (n, cap)
        out.update(best_value=bv.value, initial_value=iv.value, phase1_best_value=p1.value, n_samples=n)
        return out

    def calibrate_pso(self, pso_settings: dict, mh_seed: int, mh_iterations: int, burn_in: int,
                      adaptation_period: int = 100, thinning: int = 1, chains: int = 1) -> dict:
        """HipModelCalibrator with the particle swarm as phase 1 (SEPAIHRDModelCalibration::runPSOMCMC)."""
        cap = 1 + (mh_iterations - 1) // max(1, thinning)
        out = {"best": np.empty(self.P), "phase2_cov": np.empty((self.P, self.P)),
               "accept_trace": np.empty((chains, mh_iterations - 1), dtype=np.uint8),
               "samples": np.empty((chains, cap, self.P)), "sample_values": np.empty((chains, cap)),
               "mcmc_objective_values": np.empty((chains, cap))}
        keys = (C.c_char_p * len(pso_settings))(*[k.encode() for k in pso_settings])
        vals = np.array([float(v) for v in pso_settings.values()])
        bv, iv, p1 = C.c_double(0.0), C.c_double(0.0), C.c_double(0.0)
        ns = C.c_int32(0)
        rc = self.lib.host_calibrate_pso(self.h, keys, vals.ctypes.data, len(pso_settings), mh_iterations, burn_in,
                                         adaptation_period, thinning, mh_seed, chains, out["best"].ctypes.data,
                                         C.byref(bv), C.byref(iv), C.byref(p1), out["phase2_cov"].ctypes.data,
                                         out["accept_trace"].ctypes.data, out["samples"].ctypes.data,
                                         out["sample_values"].ctypes.data, out["mcmc_objective_values"].ctypes.data,
                                         C.byref(ns))
        if rc != 0:
            raise RuntimeError("host_calibrate_pso: " + self.lib.host_last_error().decode())
        assert ns.value == cap, (ns.value, cap)
        out.update(best_value=bv.value, initial_value=iv.value, phase1_best_value=p1.value, n_samples=ns.value)
        return out

    def posterior_ensemble(self, samples, num_for_ppc: int, seed: int, burn_in: int = 0, thinning: int = 1,
                           want_sero: bool = True, want_rt: bool = False, device: int = -1) -> dict:
        """HipPosteriorEnsemble::aggregatePosteriorPredictives (+ aggregateSeroprevalence) over this handle's
        parameter manager and data.  ppc: [6][5: lower_95, lower_90, median, upper_90, upper_95][T_pos][n]."""
        ps = np.ascontiguousarray(np.atleast_2d(samples), dtype=np.float64)
        keep: list = []
        st = hipabi.build_problem_struct(self.pb, keep)
        n, T = self.pb.n, self.pb.n_times
        Tp = int(np.sum(np.asarray(self.pb.times) >= 0.0))
        ppc = np.empty((6, 5, Tp, n))
        sel = np.empty(max(ps.shape[0], num_for_ppc, 1), dtype=np.int32)
        nsel, used = C.c_int32(0), C.c_int32(0)
        sero = np.empty((5, T)) if want_sero else None
        rt = np.empty((5, T)) if want_rt else None
        rc = self.lib.host_ensemble(self.h, C.byref(st), device, ps.ctypes.data, ps.shape[0], num_for_ppc, seed,
                                    ppc.ctypes.data, sel.ctypes.data, C.byref(nsel), C.byref(used), burn_in, thinning,
                                    sero.ctypes.data if want_sero else None, rt.ctypes.data if want_rt else None)
        if rc != 0:
            raise RuntimeError("host_ensemble: " + self.lib.host_last_error().decode())
        return {"ppc": ppc, "selected": sel[:nsel.value].copy(), "samples_used": used.value, "sero": sero, "rt": rt}

    def calculate(self, theta) -> float:
        th = np.ascontiguousarray(theta, dtype=np.float64)
        v = C.c_double()
        if self.lib.host_objective_calculate(self.h, th.ctypes.data, C.byref(v)):
            raise RuntimeError(self.lib.host_last_error().decode())
        return v.value

    def calculate_batch(self, thetas):
        th = np.ascontiguousarray(thetas, dtype=np.float64)
        B = th.shape[0]
        out = np.empty(B)
        status = np.empty(B, dtype=np.int32)
        if self.lib.host_objective_calculate_batch(self.h, th.ctypes.data, B, out.ctypes.data, status.ctypes.data):
            raise RuntimeError(self.lib.host_last_error().decode())
        return out, status

    def cache_stats(self):
        a, b, c = C.c_long(), C.c_long(), C.c_long()
        self.lib.host_cache_stats(self.h, C.byref(a), C.byref(b), C.byref(c))
        return {"calls": a.value, "hits": b.value, "size": c.value}

    def apply_constraints(self, theta, mode: int) -> np.ndarray:
        th = np.ascontiguousarray(np.atleast_2d(theta), dtype=np.float64)
        out = np.empty_like(th)
        for b in range(th.shape[0]):
            self.lib.host_apply_constraints(self.h, mode, th[b].ctypes.data, out[b].ctypes.data)
        return out

    def current_parameters(self) -> np.ndarray:
        out = np.empty(self.P)
        self.lib.host_current_parameters(self.h, out.ctypes.data)
        return out

    def metropolis_hastings_reported(self, initial, seed: int, iterations: int, burn_in: int, out_dir: str, log_path: str,
                                     adaptation_period: int = 100, thinning: int = 1, report_interval: int = 100,
                                     checkpoint_chains: int = 1, device_state: bool = True, device_streams: bool = True) -> dict:
        """The sampler with the reference's progress reports and trace files ON (MetropolisHastingsSampler.cpp:363-383,
        399-411,440-469): lines into log_path, posterior_trace_checkpoint.csv / _final.csv / posterior_trace.csv into out_dir."""
        x0 = np.ascontiguousarray(np.atleast_2d(initial), dtype=np.float64)
        Cn, P = x0.shape
        n_s = 1 + (max(iterations, 1) - 1) // max(1, thinning)
        samples, values = np.zeros((Cn, n_s, P)), np.zeros((Cn, n_s))
        ns, fb = C.c_int32(), C.c_int()
        failures = (C.c_long * 3)()
        self.lib.host_mh_run_reported.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint32] + [C.c_int] * 8 + \
            [C.c_char_p, C.c_char_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int), C.POINTER(C.c_long)]
        rc = self.lib.host_mh_run_reported(self.h, Cn, x0.ctypes.data, seed, iterations, burn_in, adaptation_period, thinning,
                                           report_interval, checkpoint_chains, int(device_state), int(device_streams),
                                           out_dir.encode(), log_path.encode(), samples.ctypes.data, values.ctypes.data,
                                           C.byref(ns), C.byref(fb), failures)
        if rc:
            raise RuntimeError(self.lib.host_last_error().decode())
        assert ns.value == n_s
        return {"samples": samples, "sample_values": values, "fell_back": bool(fb.value), "failures": list(failures)}

    def metropolis_hastings(self, initial, seed: int, iterations: int, burn_in: int, adaptation_period: int = 100,
                            thinning: int = 1, reg_eps: float = 1e-6, target_acc: float = 0.234,
                            adapt_scale: bool = True, scalar_interface: bool = False, device_state: bool = False,
                            two_pass_covariance: bool = False, want_trace: bool = True, adaptation_window: int = 0,
                            device_streams: bool = True) -> dict:
        """MultiChainMetropolisHastings (MetropolisHastingsSampler.cpp:201-412 for C lock-step chains): host-state loop,
        scalar interface, or (device_state) adaptation state resident in HBM.  two_pass_covariance: the reference's
        literal covariance refresh over the whole history instead of running co-moments.  device_streams (device_state
        only): the chains' mt19937 streams drawn on the device (default) or on the host."""
        x0 = np.ascontiguousarray(np.atleast_2d(initial), dtype=np.float64)
        Cn, P = x0.shape
        accepted = np.zeros(Cn, dtype=np.int32)
        best_value, final_scale = np.zeros(Cn), np.zeros(Cn)
        best = np.zeros((Cn, P))
        trace = np.zeros((Cn, max(iterations - 1, 1)), dtype=np.uint8) if want_trace else None
        ns = C.c_int32()
        # the C side packs with its own n_samples: t = 0 plus every thinning-th t
        n_s = 1 + (max(iterations, 1) - 1) // max(1, thinning)
        samples = np.zeros((Cn, n_s, P))
        values = np.zeros((Cn, n_s))
        cov = np.zeros((Cn, P, P))
        rc = self.lib.host_mh_run(self.h, Cn, x0.ctypes.data, seed, iterations, burn_in, adaptation_period, thinning,
                                  reg_eps, target_acc, int(adapt_scale), 2 if device_state else int(scalar_interface),
                                  accepted.ctypes.data,
                                  best_value.ctypes.data, best.ctypes.data, final_scale.ctypes.data,
                                  trace.ctypes.data if want_trace else None, C.byref(ns), samples.ctypes.data,
                                  values.ctypes.data, int(two_pass_covariance), cov.ctypes.data, int(adaptation_window),
                                  int(device_streams))
        if rc:
            raise RuntimeError(self.lib.host_last_error().decode())
        assert ns.value == n_s
        return {"accepted": accepted, "best_value": best_value, "best": best, "final_scale": final_scale,
                "accept_trace": trace[:, :iterations - 1] if want_trace else None, "samples": samples,
                "sample_values": values, "final_cov": cov,
                "loop_seconds": float(self.lib.host_last_mh_loop_seconds())}


def libm_selfcheck() -> dict:
    """The host twin of sepaihrd_device_libm_check (no device): csrc/sepaihrd_rng.inc's log / exp compiled for the host
    against this process's std::log / std::exp on the self-check arguments; also returns the arguments."""
    lib = load_library()
    n, dl, de = C.c_int(), C.c_int(), C.c_int()
    lib.host_libm_selfcheck.argtypes = [C.POINTER(C.c_int)] * 3
    lib.host_libm_selfcheck.restype = None
    lib.host_libm_selfcheck(C.byref(n), C.byref(dl), C.byref(de))
    la, ea = np.empty(n.value), np.empty(n.value)
    lib.host_libm_selfcheck_args.argtypes = [C.c_void_p, C.c_void_p]
    lib.host_libm_selfcheck_args.restype = None
    lib.host_libm_selfcheck_args(la.ctypes.data, ea.ctypes.data)
    return {"n": n.value, "log_diff": dl.value, "exp_diff": de.value, "log_args": la, "exp_args": ea}


def metropolis_hastings_groups(objectives, initial, seed: int, iterations: int, burn_in: int, adaptation_period: int = 100,
                               thinning: int = 1) -> dict:
    """MultiChainMetropolisHastings::optimizeChainGroupsOnDevice over len(objectives) HostObjective handles
    (one device context, stream and host thread per group of chains)."""
    lib = load_library()
    x0 = np.ascontiguousarray(np.atleast_2d(initial), dtype=np.float64)
    Cn, P = x0.shape
    G = len(objectives)
    handles = (C.c_void_p * G)(*[o.h for o in objectives])
    accepted = np.zeros(Cn, dtype=np.int32)
    best_value = np.zeros(Cn)
    best = np.zeros((Cn, P))
    trace = np.zeros((Cn, max(iterations - 1, 1)), dtype=np.uint8)
    rc = lib.host_mh_run_groups(handles, G, Cn, x0.ctypes.data, seed, iterations, burn_in, adaptation_period, thinning,
                                accepted.ctypes.data, best_value.ctypes.data, best.ctypes.data, trace.ctypes.data)
    if rc:
        raise RuntimeError(lib.host_last_error().decode())
    return {"accepted": accepted, "best_value": best_value, "best": best, "accept_trace": trace[:, :iterations - 1],
            "loop_seconds": float(lib.host_last_mh_loop_seconds())}


def reference_constructors(pb, thetas) -> dict:
    """The objective and the parameter manager built with the REFERENCE's constructor argument lists
    (model first; SEPAIHRDModelCalibration.cpp:84-118): values[manager][mode][b] of calculate() for a Hip manager
    built from the model and for a manager of another type, and the calibrated entries read back from the model
    after updateModelParameters(thetas[0])."""
    lib = load_library()
    keep: list = []
    st = hipabi.build_problem_struct(pb, keep)
    sig = np.ascontiguousarray(pb.sigma_array())
    th = np.ascontiguousarray(np.atleast_2d(thetas), dtype=np.float64)
    B, P = th.shape
    values = np.empty((2, 2, B))
    back = np.empty(P)
    if lib.host_reference_constructors(C.byref(st), "\n".join(pb.param_names).encode(), "\n".join(pb.npi_names).encode(),
                                       sig.ctypes.data, th.ctypes.data, B, values.ctypes.data, back.ctypes.data):
        raise RuntimeError("host_reference_constructors: " + lib.host_last_error().decode())
    return {"values": values, "model_back": back}


def model_holders(ends_after, values_after, baseline, baseline_end, t) -> dict:
    """PiecewiseConstantNpiStrategy::getReductionFactor at times t and what AgeSEPAIHRDModel reports (no device)."""
    lib = load_library()
    ea = np.ascontiguousarray(ends_after, dtype=np.float64)
    va = np.ascontiguousarray(values_after, dtype=np.float64)
    tt = np.ascontiguousarray(t, dtype=np.float64)
    kap = np.empty(len(tt))
    se, sv = np.empty(len(ea) + 1), np.empty(len(ea) + 1)
    size = C.c_int()
    buf = C.create_string_buffer(128)
    rc = lib.host_model_holders(ea.ctypes.data, va.ctypes.data, len(ea), baseline, baseline_end, tt.ctypes.data, len(tt),
                                kap.ctypes.data, se.ctypes.data, sv.ctypes.data, C.byref(size), buf, len(buf))
    if rc:
        raise RuntimeError("host_model_holders: rc %d %s" % (rc, lib.host_last_error().decode()))
    return {"kappa": kap, "schedule_ends": se, "schedule_values": sv, "state_size": size.value, "names": buf.value.decode()}


def summary_quantiles(table, probs) -> np.ndarray:
    """MultiChainMetropolisHastings::summaryQuantiles (pure host): exact-sort quantiles across chains of every column."""
    lib = load_library()
    t = np.ascontiguousarray(table, dtype=np.float64)
    q = np.ascontiguousarray(probs, dtype=np.float64)
    out = np.empty((len(q), t.shape[1]))
    lib.host_summary_quantiles.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    if lib.host_summary_quantiles(t.ctypes.data, t.shape[0], t.shape[1], q.ctypes.data, len(q), out.ctypes.data):
        raise RuntimeError(lib.host_last_error().decode())
    return out


def metropolis_hastings_group_summaries(objectives, initial, seed: int, iterations: int, burn_in: int, adaptation_period: int = 100,
                                        thinning: int = 1, backend: int = 0) -> dict:
    """optimizeChainGroupsOnDevice + gatherChainSummaries: the per-chain summary records (host concatenation), the table
    every group's device holds after the all-gather, and the samples they were formed from."""
    lib = load_library()
    x0 = np.ascontiguousarray(np.atleast_2d(initial), dtype=np.float64)
    Cn, P = x0.shape
    G = len(objectives)
    W = 2 * P + 2
    n_s = 1 + (max(iterations, 1) - 1) // max(1, thinning)
    handles = (C.c_void_p * G)(*[o.h for o in objectives])
    records, gathered = np.zeros((Cn, W)), np.zeros((G, Cn, W))
    samples, best_value, accepted = np.zeros((Cn, n_s, P)), np.zeros(Cn), np.zeros(Cn, dtype=np.int32)
    used = C.c_int32(-1)
    lib.host_mh_groups_summaries.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int,
                                             C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    rc = lib.host_mh_groups_summaries(handles, G, Cn, x0.ctypes.data, seed, iterations, burn_in, adaptation_period, thinning, backend,
                                      records.ctypes.data, gathered.ctypes.data, samples.ctypes.data, best_value.ctypes.data,
                                      accepted.ctypes.data, C.byref(used))
    if rc:
        raise RuntimeError(lib.host_last_error().decode())
    return {"records": records, "gathered": gathered, "samples": samples, "best_value": best_value, "accepted": accepted,
            "backend_used": int(used.value)}


def default_arith() -> int:
    """ARITH_* the reference-shaped constructors of the C++ adapters select (environment SEPAIHRD_ARITH; fma unless
    "strict").  No device needed."""
    lib = load_library()
    lib.host_default_arith.restype = C.c_int
    return int(lib.host_default_arith())


def glibc_log(x) -> np.ndarray:
    """csrc/sepaihrd_rng.inc's restatement of glibc's log, compiled for the host (test hook)."""
    lib = load_library()
    a = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty_like(a)
    lib.host_glibc_log.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    lib.host_glibc_log.restype = None
    lib.host_glibc_log(a.ctypes.data, a.size, out.ctypes.data)
    return out


def glibc_exp(x) -> np.ndarray:
    """csrc/sepaihrd_rng.inc's restatement of glibc's exp, compiled for the host (test hook)."""
    lib = load_library()
    a = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty_like(a)
    lib.host_glibc_exp.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    lib.host_glibc_exp.restype = None
    lib.host_glibc_exp(a.ctypes.data, a.size, out.ctypes.data)
    return out
