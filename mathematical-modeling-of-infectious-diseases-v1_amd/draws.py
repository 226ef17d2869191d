"""Synthetic parameter draws for batches of chains (host-side workload generator).

Mirrors the jitter protocol of the reference's own benchmark
(src/model/sepaihrd_objective_benchmark_main.cpp:412-413,452-460): chain b owns
``std::mt19937(seed0 + b)`` and ONE persistent ``std::normal_distribution<double>``;
``theta_i = base_i + sigma_i * N(0,1)`` for i = 0..P-1, then ``applyConstraints``.

The libstdc++ draw order is reproduced exactly (bits/random.tcc, GCC 11):
``generate_canonical<double,53>`` consumes two 32-bit words (low word first) and the
normal distribution is Marsaglia's polar method returning ``y*mult`` first and caching
``x*mult`` for the next call.
"""
from __future__ import annotations

import math

import numpy as np

_TWO32 = 4294967296.0
_TWO64 = 18446744073709551616.0


def mt19937_words(seed: int, count: int) -> np.ndarray:
    """First ``count`` outputs of std::mt19937(seed) (numpy's legacy seeding is init_genrand)."""
    rs = np.random.RandomState(int(seed) & 0xFFFFFFFF)
    return rs.randint(0, 2 ** 32, size=count, dtype=np.uint64).astype(np.float64)


def _canonical(w_lo: np.ndarray, w_hi: np.ndarray) -> np.ndarray:
    r = (w_lo + w_hi * _TWO32) / _TWO64
    return np.where(r >= 1.0, np.nextafter(1.0, 0.0), r)


def std_normals_batch(seeds: np.ndarray, count: int) -> np.ndarray:
    """count standard normals for every seed, libstdc++ order.  Returns len(seeds) x count."""
    seeds = np.asarray(seeds)
    B = len(seeds)
    pairs = (count + 1) // 2
    budget = 4 * pairs * 2 + 64  # words per chain; grown on demand
    while True:
        words = np.stack([mt19937_words(s, budget) for s in seeds])
        pos = np.zeros(B, dtype=np.int64)
        out = np.empty((B, 2 * pairs))
        rows = np.arange(B)
        ok_budget = True
        for k in range(pairs):
            todo = np.ones(B, dtype=bool)
            x = np.zeros(B)
            y = np.zeros(B)
            r2 = np.zeros(B)
            while todo.any():
                if (pos[todo] + 4 > budget).any():
                    ok_budget = False
                    break
                idx = rows[todo]
                p = pos[todo]
                xs = 2.0 * _canonical(words[idx, p], words[idx, p + 1]) - 1.0
                ys = 2.0 * _canonical(words[idx, p + 2], words[idx, p + 3]) - 1.0
                rr = xs * xs + ys * ys
                x[idx], y[idx], r2[idx] = xs, ys, rr
                pos[idx] += 4
                todo[idx] = (rr > 1.0) | (rr == 0.0)
            if not ok_budget:
                break
            # libm log (std::log of the host), not numpy's SIMD log: they differ by 1 ulp in ~1e-4 of calls
            logr2 = np.fromiter((math.log(v) for v in r2), dtype=np.float64, count=B)
            mult = np.sqrt(-2.0 * logr2 / r2)
            out[:, 2 * k] = y * mult      # returned first
            out[:, 2 * k + 1] = x * mult  # saved, returned by the next call
        if ok_budget:
            return out[:, :count]
        budget *= 2


def reflect_bound(v: np.ndarray, lo: np.ndarray, hi: np.ndarray) -> np.ndarray:
    """SEPAIHRDParameterManager.cpp:302-313 (np.fmod == std::fmod)."""
    width = hi - lo
    safe = np.where(width > 0, width, 1.0)
    y = np.fmod(v - lo, 2.0 * safe)
    y = np.where(y < 0, y + 2.0 * safe, y)
    r = np.where(y <= safe, lo + y, hi - (y - safe))
    return np.where(lo >= hi, lo, r)


def apply_constraints(theta: np.ndarray, lo, hi, has, mode: int) -> np.ndarray:
    """SEPAIHRDParameterManager::applyConstraints (:315-347); mode 0 clamp, 1 reflect."""
    theta = np.asarray(theta, dtype=np.float64)
    lo2, hi2 = np.minimum(lo, hi), np.maximum(lo, hi)
    has = np.asarray(has).astype(bool)
    if mode == 0:
        bounded = np.minimum(np.maximum(theta, lo2), hi2)
        free = np.maximum(0.0, theta)
    else:
        bounded = reflect_bound(theta, lo2, hi2)
        free = np.abs(theta)
    return np.where(has, bounded, free)


def jitter_draws(pb, seed0: int, B: int, mode: int = 1, base=None) -> np.ndarray:
    """B x P matrix of constrained draws; chain b uses mt19937(seed0 + b)."""
    base = np.asarray(pb.base_theta if base is None else base, dtype=np.float64)
    z = std_normals_batch(np.arange(seed0, seed0 + B, dtype=np.int64), pb.n_params)
    cand = base[None, :] + pb.sigma_array()[None, :] * z
    lo, hi, has = pb.bounds_arrays()
    return apply_constraints(cand, lo, hi, has, mode)
