"""Deterministic synthetic encoded queries for bench.py and the parity tests.

Counter-based ``splitmix64(seed, index)`` -> U[0,1), so the same matrix is produced by any NumPy
version and by the C side.  Rows mimic the reference encoder (QuerySampler.py:200-221): d/2
(upper, lower) pairs scaled to [0, 1000], 1..10 active range predicates per query, inactive pairs at
their (0, 1000) default; labels are log2 of an independence-model cardinality on a Covertype-sized
table, so y lies in about [0, 19] like the real log2(card).
"""
from __future__ import annotations

import numpy as np

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(seed: int, idx) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = (np.asarray(idx, dtype=np.uint64) + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(seed) * np.uint64(0xD1B54A32D192ED03)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def uniform01(seed: int, idx) -> np.ndarray:
    return (splitmix64(seed, idx) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def synthetic_queries(n: int, d: int, seed: int = 0, join_block: bool = False, table_rows: float = 581012.0):
    """X [n, d] float64 (encoder layout), Y [n, 1] = log2(card)."""
    n_join = (d // 8) if join_block else 0
    pairs = (d - 3 * n_join) // 2
    assert pairs >= 1, "d too small"
    stride = 3 * pairs + 2 + n_join
    base = np.arange(n, dtype=np.uint64)[:, None] * np.uint64(stride)
    col = np.arange(pairs, dtype=np.uint64)[None, :]
    key = uniform01(seed, base + col)
    u1 = uniform01(seed, base + np.uint64(pairs) + col) * 1000.0
    u2 = uniform01(seed, base + np.uint64(2 * pairs) + col) * 1000.0
    pmax = min(10, pairs)
    p_r = 1 + np.floor(uniform01(seed, base[:, 0] + np.uint64(3 * pairs)) * pmax).astype(np.int64)
    rank = np.argsort(np.argsort(key, axis=1), axis=1)
    active = rank < p_r[:, None]
    upper, lower = np.maximum(u1, u2), np.minimum(u1, u2)
    X = np.zeros((n, d), dtype=np.float64)
    X[:, 0:2 * pairs:2] = np.where(active, upper, 0.0)
    X[:, 1:2 * pairs:2] = np.where(active, lower, 1000.0)
    sel = np.where(active, (upper - lower) / 1000.0, 1.0)
    card = table_rows * np.prod(sel, axis=1)
    if n_join:
        ju = uniform01(seed, base + np.uint64(3 * pairs + 2) + np.arange(n_join, dtype=np.uint64)[None, :])
        jact = ju < 0.35
        X[:, 2 * pairs + 2::3][:, :n_join] = jact.astype(np.float64)
        card = card * np.power(4.0, jact.sum(axis=1))
    Y = np.maximum(0.0, np.log2(np.maximum(card, 1e-300)))[:, None]
    return X, Y
