"""ctypes view of the C ABI in include/sepaihrd_hip.h (test / bench plumbing).

The reference-side binding a C++ maintainer would add is the adapter in ``host/``
(see INTEGRATION.md); this module is what tests/ and bench.py use to drive the same
entry points from Python.  No compute happens here and there is no fallback: if
``libsepaihrd_hip.so`` is missing or no HIP device is present the constructors raise.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

from .problem import SEPAIHRDProblem

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsepaihrd_hip.so")
ABI_VERSION = 3
LOWEST = -np.finfo(np.float64).max

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_up = C.POINTER(C.c_uint8)


class sepaihrd_problem(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("n_age", C.c_int32), ("n_times", C.c_int32), ("n_obs", C.c_int32),
        ("n_beta", C.c_int32), ("n_kappa", C.c_int32), ("n_params", C.c_int32), ("solver", C.c_int32),
        ("constraint_mode", C.c_int32), ("arith", C.c_int32), ("max_attempts", C.c_int32),
        ("precision", C.c_int32),
        ("times", _dp), ("N", _dp), ("M", _dp),
        ("a", _dp), ("h_infec", _dp), ("p", _dp), ("h", _dp), ("icu", _dp), ("d_H", _dp), ("d_ICU", _dp),
        ("d_community", _dp),
        ("beta_end_times", _dp), ("beta_values", _dp), ("kappa_end_times", _dp), ("kappa_values", _dp),
        ("initial_state", _dp), ("obs_H", _dp), ("obs_ICU", _dp), ("obs_D", _dp),
        ("param_field", _ip), ("param_index", _ip), ("lower", _dp), ("upper", _dp), ("has_bounds", _up),
        ("beta", C.c_double), ("theta", C.c_double), ("sigma", C.c_double), ("gamma_p", C.c_double),
        ("gamma_A", C.c_double), ("gamma_I", C.c_double), ("gamma_H", C.c_double), ("gamma_ICU", C.c_double),
        ("multipliers", C.c_double * 8), ("runup_days", C.c_double), ("seed_exposed", C.c_double),
        ("abs_err", C.c_double), ("rel_err", C.c_double), ("dt_hint", C.c_double),
    ]


class sepaihrd_kernel_info(C.Structure):
    _fields_ = [
        ("lanes_per_chain", C.c_int32), ("chains_per_wave", C.c_int32), ("block_threads", C.c_int32),
        ("vgprs", C.c_int32), ("sgprs", C.c_int32), ("lds_bytes", C.c_int32), ("scratch_bytes", C.c_int32),
        ("max_blocks_per_cu", C.c_int32), ("num_cus", C.c_int32), ("likelihood_form", C.c_int32),
        ("phase_pass_applied", C.c_int32),
        ("kernel_name", C.c_char * 128), ("device_name", C.c_char * 128),
    ]


# every symbol include/sepaihrd_hip.h declares
class sepaihrd_mh_config(C.Structure):
    """include/sepaihrd_hip.h: struct sepaihrd_mh_config"""
    _fields_ = [("chains", C.c_int32), ("iterations", C.c_int32), ("thinning", C.c_int32),
                ("adaptation_window", C.c_int32), ("covariance_mode", C.c_int32), ("reserved", C.c_int32),
                ("reg_eps", C.c_double), ("scaling_factor", C.c_double)]


MH_COV_RUNNING, MH_COV_TWO_PASS = 0, 1
FORM_AUTO, FORM_LANE_PER_AGE, FORM_QUAD = 0, 1, 2
LL_INLINE, LL_SEPARATE_PASS, LL_CONSUMER_WAVES = 0, 1, 2
GATHER_AUTO, GATHER_RCCL, GATHER_HOST = 0, 1, 2


def mh_create(lib, ctx, chains: int, iterations: int, x0: np.ndarray, cov0: np.ndarray, reg_eps: float = 1e-6,
              scaling_factor: Optional[float] = None, thinning: int = 1, adaptation_window: int = 0,
              covariance_mode: int = MH_COV_RUNNING):
    """sepaihrd_mh_create with its config struct filled; returns the opaque sampler handle (None on failure)."""
    P = x0.shape[-1]
    cfg = sepaihrd_mh_config(chains, iterations, thinning, adaptation_window, covariance_mode, 0, reg_eps,
                             scaling_factor if scaling_factor is not None else 2.38 * 2.38 / P)
    x0 = np.ascontiguousarray(x0, dtype=np.float64)
    cov0 = np.ascontiguousarray(cov0, dtype=np.float64)
    return lib.sepaihrd_mh_create(ctx, C.byref(cfg), x0.ctypes.data, cov0.ctypes.data)


EXPORTED_SYMBOLS = (
    "sepaihrd_create", "sepaihrd_destroy", "sepaihrd_last_error", "sepaihrd_abi_version",
    "sepaihrd_set_constraint_mode", "sepaihrd_set_arith", "sepaihrd_set_precision", "sepaihrd_set_integrator_form", "sepaihrd_eval_batch",
    "sepaihrd_eval_batch_device", "sepaihrd_eval_batch_begin", "sepaihrd_eval_batch_end", "sepaihrd_apply_constraints", "sepaihrd_get_kernel_info", "sepaihrd_get_kernel_info_for_batch", "sepaihrd_reserve",
    "sepaihrd_set_timing", "sepaihrd_get_timing", "sepaihrd_set_initial_state_mode",
    "sepaihrd_ensemble_quantiles", "sepaihrd_mh_create", "sepaihrd_mh_destroy", "sepaihrd_mh_evaluate_current",
    "sepaihrd_mh_propose", "sepaihrd_mh_fetch", "sepaihrd_mh_stage_normals", "sepaihrd_mh_staging_buffer", "sepaihrd_mh_step", "sepaihrd_mh_read_best", "sepaihrd_mh_busy", "sepaihrd_mh_set_values", "sepaihrd_mh_test_buffer",
    "sepaihrd_mh_step_tested", "sepaihrd_mh_fetch_test", "sepaihrd_mh_commit", "sepaihrd_mh_adapt", "sepaihrd_mh_read_history",
    "sepaihrd_mh_read_covariance", "sepaihrd_mh_read_proposal", "sepaihrd_mh_history_length",
    "sepaihrd_mh_sample_count", "sepaihrd_mh_read_samples", "sepaihrd_mh_read_moments", "sepaihrd_mh_summary_records",
    "sepaihrd_records_buffer", "sepaihrd_allgather_records", "sepaihrd_read_records", "sepaihrd_write_records",
    "sepaihrd_mh_seed_streams", "sepaihrd_mh_draw_first", "sepaihrd_mh_keep_scale_on_device", "sepaihrd_mh_read_run_state",
    "sepaihrd_mh_read_sample_values", "sepaihrd_mh_read_accept_trace",
    "sepaihrd_device_libm_check", "sepaihrd_mh_read_failure_counts", "sepaihrd_mh_snapshot_begin", "sepaihrd_mh_snapshot_end",
    "sepaihrd_device_log_values",
)

_lib = None


def load_library(path: Optional[str] = None) -> C.CDLL:
    """dlopen the HIP library; raises (never falls back) when it is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("SEPAIHRD_HIP_LIB") or LIB_PATH  # env override: experiment builds only
    # One HIP runtime per process: PyTorch-ROCm wheels bundle their own libamdhip64.so.7 /
    # libhsa-runtime64 and a second copy (the system ROCm the library was linked against) cannot
    # open the device once the first has.  Importing torch FIRST makes the dynamic linker resolve
    # our NEEDED libamdhip64.so.7 to the already-loaded copy (same SONAME), so torch tensors,
    # torch streams and these kernels share one runtime.  Without torch (C++ hosts) the system
    # runtime from the library's RUNPATH is used.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(p):
        raise FileNotFoundError(
            f"{p} not found: build it with __graft_entry__.build() (hipcc --offload-arch=gfx950); "
            "the HIP path has no CPU fallback")
    lib = C.CDLL(p)
    vp = C.c_void_p
    lib.sepaihrd_create.restype = vp
    lib.sepaihrd_create.argtypes = [C.POINTER(sepaihrd_problem), C.c_int, C.c_char_p, C.c_int]
    lib.sepaihrd_destroy.restype = None
    lib.sepaihrd_destroy.argtypes = [vp]
    lib.sepaihrd_last_error.restype = C.c_char_p
    lib.sepaihrd_last_error.argtypes = [vp]
    lib.sepaihrd_abi_version.restype = C.c_int
    lib.sepaihrd_set_constraint_mode.argtypes = [vp, C.c_int]
    lib.sepaihrd_set_arith.argtypes = [vp, C.c_int]
    lib.sepaihrd_set_precision.argtypes = [vp, C.c_int]
    lib.sepaihrd_set_integrator_form.argtypes = [vp, C.c_int]
    lib.sepaihrd_eval_batch.argtypes = [vp, vp, C.c_int, vp, vp, vp, vp, vp, vp]
    lib.sepaihrd_eval_batch_device.argtypes = [vp, vp, C.c_int, vp, vp, vp, vp, vp, vp, vp]
    lib.sepaihrd_apply_constraints.argtypes = [vp, C.c_int, vp, C.c_int, vp]
    lib.sepaihrd_get_kernel_info.argtypes = [vp, C.POINTER(sepaihrd_kernel_info)]
    lib.sepaihrd_get_kernel_info_for_batch.argtypes = [vp, C.c_int32, C.POINTER(sepaihrd_kernel_info)]
    lib.sepaihrd_reserve.argtypes = [vp, C.c_int]
    lib.sepaihrd_set_timing.argtypes = [vp, C.c_int]
    lib.sepaihrd_get_timing.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int)]
    lib.sepaihrd_mh_create.restype = vp
    lib.sepaihrd_mh_create.argtypes = [vp, C.POINTER(sepaihrd_mh_config), vp, vp]
    lib.sepaihrd_mh_sample_count.argtypes = [vp]
    lib.sepaihrd_mh_read_samples.argtypes = [vp, C.c_int, C.c_int, vp]
    lib.sepaihrd_mh_read_moments.argtypes = [vp, vp, vp]
    lib.sepaihrd_mh_summary_records.argtypes = [vp, C.c_int, vp, vp]
    lib.sepaihrd_records_buffer.restype = vp
    lib.sepaihrd_records_buffer.argtypes = [vp, C.c_int, C.c_size_t]
    lib.sepaihrd_allgather_records.argtypes = [C.POINTER(vp), C.c_int, vp, C.c_int, C.c_int, C.POINTER(C.c_int)]
    lib.sepaihrd_read_records.argtypes = [vp, C.c_int, vp, C.c_size_t]
    lib.sepaihrd_write_records.argtypes = [vp, C.c_int, vp, C.c_size_t]
    lib.sepaihrd_mh_set_values.argtypes = [vp, vp]
    # every entry point a Python caller may reach takes its pointers as pointers (a bare int would be cut to 32 bits)
    lib.sepaihrd_abi_version.argtypes = []
    lib.sepaihrd_eval_batch_begin.argtypes = [vp, vp, C.c_int]
    lib.sepaihrd_eval_batch_end.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.sepaihrd_mh_stage_normals.argtypes = [vp, vp]
    lib.sepaihrd_mh_staging_buffer.restype = vp
    lib.sepaihrd_mh_staging_buffer.argtypes = [vp]
    lib.sepaihrd_mh_step.argtypes = [vp, vp, vp, vp, vp, C.c_int, C.c_double, C.c_int]
    lib.sepaihrd_mh_test_buffer.restype = vp
    lib.sepaihrd_mh_test_buffer.argtypes = [vp]
    lib.sepaihrd_mh_step_tested.argtypes = [vp, C.c_double, C.c_int, C.c_int]
    lib.sepaihrd_mh_fetch_test.argtypes = [vp, vp, vp]
    lib.sepaihrd_mh_read_best.argtypes = [vp, vp]
    lib.sepaihrd_mh_busy.argtypes = [vp]
    lib.sepaihrd_mh_seed_streams.argtypes = [vp, C.c_uint32]
    lib.sepaihrd_mh_draw_first.argtypes = [vp]
    lib.sepaihrd_mh_keep_scale_on_device.argtypes = [vp, C.c_int, C.c_double, C.c_int]
    lib.sepaihrd_mh_read_run_state.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.sepaihrd_mh_read_sample_values.argtypes = [vp, C.c_int, C.c_int, vp]
    lib.sepaihrd_mh_read_accept_trace.argtypes = [vp, vp]
    lib.sepaihrd_device_libm_check.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.sepaihrd_device_log_values.argtypes = [vp, vp, C.c_int32, vp]
    lib.sepaihrd_mh_read_failure_counts.argtypes = [vp, C.POINTER(C.c_int64)]
    lib.sepaihrd_mh_snapshot_begin.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int]
    lib.sepaihrd_mh_snapshot_end.argtypes = [vp, C.c_int, vp, vp, vp]
    lib.sepaihrd_mh_destroy.restype = None
    lib.sepaihrd_mh_destroy.argtypes = [vp]
    lib.sepaihrd_mh_evaluate_current.argtypes = [vp, vp, vp]
    lib.sepaihrd_mh_propose.argtypes = [vp, vp, vp, vp, vp]
    lib.sepaihrd_mh_fetch.argtypes = [vp, vp, vp]
    lib.sepaihrd_mh_commit.argtypes = [vp, vp]
    lib.sepaihrd_mh_adapt.argtypes = [vp, C.c_double, C.c_int, C.c_int]
    lib.sepaihrd_mh_read_history.argtypes = [vp, vp, C.c_int, vp]
    lib.sepaihrd_mh_read_covariance.argtypes = [vp, vp]
    lib.sepaihrd_mh_read_proposal.argtypes = [vp, vp]
    lib.sepaihrd_mh_history_length.argtypes = [vp]
    lib.sepaihrd_set_initial_state_mode.argtypes = [vp, C.c_int]
    lib.sepaihrd_ensemble_quantiles.argtypes = [vp, vp, C.c_int, vp, C.c_int, vp, vp, vp, vp, vp, vp]
    if path is None:
        _lib = lib
    return lib


def _ptr(a: np.ndarray, typ):
    return a.ctypes.data_as(typ)


def build_problem_struct(pb: SEPAIHRDProblem, keep: list) -> sepaihrd_problem:
    """Fill the C struct; ``keep`` receives the arrays that must outlive the call."""
    n = pb.n
    s = sepaihrd_problem()
    s.abi_version = ABI_VERSION
    s.n_age, s.n_times, s.n_obs = n, pb.n_times, pb.n_obs
    s.n_beta, s.n_kappa, s.n_params = len(pb.beta_values), len(pb.kappa_values), pb.n_params
    s.solver, s.constraint_mode, s.arith = pb.solver, pb.constraint_mode, pb.arith
    s.max_attempts = int(getattr(pb, "max_attempts", 0))
    s.precision = int(getattr(pb, "precision", 0))

    def dbl(x):
        a = np.ascontiguousarray(x, dtype=np.float64)
        if a.size == 0:
            a = np.zeros(1)
        keep.append(a)
        return _ptr(a, _dp)

    s.times, s.N = dbl(pb.times), dbl(pb.N)
    s.M = dbl(np.asfortranarray(pb.M).ravel(order="F"))  # column-major like Eigen
    for name in ("a", "h_infec", "p", "h", "icu", "d_H", "d_ICU", "d_community", "beta_end_times",
                 "beta_values", "kappa_end_times", "kappa_values", "initial_state"):
        setattr(s, name, dbl(getattr(pb, name)))
    s.obs_H, s.obs_ICU, s.obs_D = dbl(pb.obs_H), dbl(pb.obs_ICU), dbl(pb.obs_D)
    codes, idxs = pb.field_map()
    lo, hi, has = pb.bounds_arrays()
    keep.extend([codes, idxs, has])
    s.param_field, s.param_index = _ptr(codes, _ip), _ptr(idxs, _ip)
    s.lower, s.upper, s.has_bounds = dbl(lo), dbl(hi), _ptr(has, _up)
    s.beta, s.theta, s.sigma, s.gamma_p = pb.beta, pb.theta, pb.sigma, pb.gamma_p
    s.gamma_A, s.gamma_I, s.gamma_H, s.gamma_ICU = pb.gamma_A, pb.gamma_I, pb.gamma_H, pb.gamma_ICU
    for i in range(8):
        s.multipliers[i] = float(pb.multipliers[i])
    s.runup_days, s.seed_exposed = pb.runup_days, pb.seed_exposed
    s.abs_err, s.rel_err, s.dt_hint = pb.abs_err, pb.rel_err, pb.dt_hint
    return s


class HipObjective:
    """Batched SEPAIHRD objective on one MI355X; mirrors IObjectiveFunction for B thetas."""

    def __init__(self, pb: SEPAIHRDProblem, device: int = -1):
        self.lib = load_library()
        self.pb = pb
        keep: list = []
        st = build_problem_struct(pb, keep)
        err = C.create_string_buffer(512)
        self.ctx = self.lib.sepaihrd_create(C.byref(st), device, err, len(err))
        if not self.ctx:
            raise RuntimeError("sepaihrd_create failed: " + err.value.decode())
        self.P, self.n, self.T = pb.n_params, pb.n, pb.n_times

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.sepaihrd_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int, what: str):
        if rc != 0:
            raise RuntimeError(f"{what} failed ({rc}): " + self.lib.sepaihrd_last_error(self.ctx).decode())

    def getParameterNames(self):
        return list(self.pb.param_names)

    def set_constraint_mode(self, mode: int):
        self._check(self.lib.sepaihrd_set_constraint_mode(self.ctx, mode), "set_constraint_mode")

    def set_arith(self, arith: int):
        self._check(self.lib.sepaihrd_set_arith(self.ctx, arith), "set_arith")

    def set_precision(self, precision: int):
        """PRECISION_F64 (reference arithmetic) or PRECISION_F32 (fp32 state, fp64 likelihood: BASELINE configs[4])."""
        self._check(self.lib.sepaihrd_set_precision(self.ctx, int(precision)), "set_precision")

    def set_integrator_form(self, form: int):
        """FORM_AUTO (by batch size), FORM_LANE_PER_AGE or FORM_QUAD (sixteen lanes per chain): same bits either way."""
        self._check(self.lib.sepaihrd_set_integrator_form(self.ctx, int(form)), "set_integrator_form")

    def calculate(self, theta) -> float:
        return float(self.eval_batch(np.asarray(theta, dtype=np.float64)[None, :])["loglik"][0])

    def eval_batch(self, theta, want_traj: bool = False) -> dict:
        th = np.ascontiguousarray(theta, dtype=np.float64)
        if th.ndim != 2 or th.shape[1] != self.P:
            raise ValueError(f"theta must be B x {self.P}")
        B = th.shape[0]
        out = {
            "loglik": np.empty(B), "status": np.empty(B, dtype=np.int32),
            "n_accept": np.empty(B, dtype=np.int32), "n_reject": np.empty(B, dtype=np.int32),
            "ll_parts": np.empty((B, 3)),
        }
        traj = np.empty((B, self.T, 11 * self.n)) if want_traj else None
        rc = self.lib.sepaihrd_eval_batch(
            self.ctx, th.ctypes.data, B, out["loglik"].ctypes.data, out["status"].ctypes.data,
            out["n_accept"].ctypes.data, out["n_reject"].ctypes.data, out["ll_parts"].ctypes.data,
            traj.ctypes.data if want_traj else None)
        self._check(rc, "sepaihrd_eval_batch")
        if want_traj:
            out["traj"] = traj
        return out

    def eval_batch_device(self, d_theta, d_loglik, d_status=None, d_n_accept=None, d_n_reject=None,
                          d_ll_parts=None, d_traj=None, stream: int = 0, B: Optional[int] = None):
        """Arguments are torch CUDA tensors (or raw device addresses); async on ``stream``."""
        def addr(t):
            if t is None:
                return None
            return t if isinstance(t, int) else t.data_ptr()
        if B is None:
            B = int(d_theta.shape[0])
        rc = self.lib.sepaihrd_eval_batch_device(self.ctx, addr(d_theta), B, addr(d_loglik), addr(d_status),
                                                 addr(d_n_accept), addr(d_n_reject), addr(d_ll_parts),
                                                 addr(d_traj), stream if stream else None)
        self._check(rc, "sepaihrd_eval_batch_device")

    def set_timing(self, enable):
        """True / 1: events around every launch; k > 1: around every k-th launch; False / 0: off."""
        self._check(self.lib.sepaihrd_set_timing(self.ctx, int(enable)), "sepaihrd_set_timing")

    def get_timing(self) -> dict:
        a, b, n = C.c_double(), C.c_double(), C.c_int()
        self._check(self.lib.sepaihrd_get_timing(self.ctx, C.byref(a), C.byref(b), C.byref(n)), "sepaihrd_get_timing")
        return {"integrator_ms": a.value, "likelihood_ms": b.value, "launches": n.value}

    def set_initial_state_mode(self, mode: int):
        """0: x(t0) derived from theta (objective); 1: problem.initial_state as given (ensemble runs)."""
        self._check(self.lib.sepaihrd_set_initial_state_mode(self.ctx, int(mode)), "set_initial_state_mode")

    def ensemble_quantiles(self, theta, probs, want_sero: bool = True, want_rt: bool = False,
                           want_metrics: bool = False) -> dict:
        """Posterior-ensemble summaries (ResultAggregator.cpp:297-345, MetricsCalculator.cpp:199-226):
        ppc [6][n_probs][T_pos][n], sero [n_probs][T], status [S], n_valid."""
        th = np.ascontiguousarray(np.atleast_2d(theta), dtype=np.float64)
        pr = np.ascontiguousarray(probs, dtype=np.float64)
        S, npb = th.shape[0], pr.size
        Tp = int(np.sum(np.asarray(self.pb.times) >= 0.0))
        ppc = np.empty((6, npb, Tp, self.pb.n))
        sero = np.empty((npb, self.pb.n_times)) if want_sero else None
        rt = np.empty((npb, self.pb.n_times)) if want_rt else None
        met = np.empty((S, 12 + 4 * self.pb.n)) if want_metrics else None
        status = np.empty(S, dtype=np.int32)
        nv = C.c_int32(0)
        self._check(self.lib.sepaihrd_ensemble_quantiles(
            self.ctx, th.ctypes.data, S, pr.ctypes.data, npb, ppc.ctypes.data,
            sero.ctypes.data if want_sero else None, rt.ctypes.data if want_rt else None,
            met.ctypes.data if want_metrics else None, status.ctypes.data, C.byref(nv)), "ensemble_quantiles")
        out = {"ppc": ppc, "status": status, "n_valid": nv.value}
        if want_sero:
            out["sero"] = sero
        if want_rt:
            out["rt"] = rt
        if want_metrics:
            out["metrics"] = met
        return out

    def reserve(self, max_B: int):
        self._check(self.lib.sepaihrd_reserve(self.ctx, int(max_B)), "sepaihrd_reserve")

    def apply_constraints(self, theta, mode: int) -> np.ndarray:
        th = np.ascontiguousarray(np.atleast_2d(theta), dtype=np.float64)
        out = np.empty_like(th)
        self._check(self.lib.sepaihrd_apply_constraints(self.ctx, mode, th.ctypes.data, th.shape[0],
                                                        out.ctypes.data), "apply_constraints")
        return out

    def device_libm_check(self):
        """(n_log_diff, n_exp_diff): arguments of the run-time self-check on which the device's log / exp restatements
        differ from this process's libm (0, 0 = the device may draw the chains' streams)."""
        a, b = C.c_int32(-1), C.c_int32(-1)
        self._check(self.lib.sepaihrd_device_libm_check(self.ctx, C.byref(a), C.byref(b)), "device_libm_check")
        return a.value, b.value

    def device_log_values(self, x) -> np.ndarray:
        """The Poisson term's log as the device evaluates it (csrc/sepaihrd_dev_common.inc log_pos), on positive normal x."""
        x = np.ascontiguousarray(x, dtype=np.float64).ravel()
        out = np.empty_like(x)
        self._check(self.lib.sepaihrd_device_log_values(self.ctx, x.ctypes.data, x.size, out.ctypes.data), "device_log_values")
        return out

    def kernel_info(self, batch: int = 0) -> dict:
        """Resource report of the integrator kernel a launch of `batch` chains uses (0: the large-batch kernel)."""
        info = sepaihrd_kernel_info()
        self._check(self.lib.sepaihrd_get_kernel_info_for_batch(self.ctx, int(batch), C.byref(info)), "get_kernel_info")
        d = {k: getattr(info, k) for k, _ in info._fields_}
        d["kernel_name"] = info.kernel_name.decode()
        d["device_name"] = info.device_name.decode()
        return d
