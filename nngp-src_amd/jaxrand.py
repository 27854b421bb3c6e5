"""The reference's pool draw, restated: ``jax.random.choice(PRNGKey(10), n, (k,), replace=False, p=p)``
(reference active/ActiveLearner.py:50-53, its default ``--biased_sample True``, active/active_train.py:62).

jax is not installable here, so this is a restatement of the published algorithm of the pinned
jax 0.3.23 (``nngp.yaml:78``; ``jax/_src/random.py`` and ``jax/_src/prng.py`` of that release), float64 as the reference
runs it (``jax_enable_x64``):

* ``PRNGKey(seed)``            -> the key pair ``(seed >> 32, seed & 0xffffffff)`` (threefry_seed);
* ``_random_bits(key, 64, (n,))`` -> ``threefry_2x32(key, iota(2 n))``: the counters are split into halves
  ``x0 = [0, n)``, ``x1 = [n, 2 n)``, one Threefry-2x32 (20 rounds) block per pair, ``bits_i = y0_i << 32 | y1_i``;
* ``uniform(key, (n,), float64, tiny, 1)`` -> ``max(tiny, f * (1 - tiny) + tiny)`` with
  ``f = bitcast(bits >> 12 | 0x3ff0000000000000) - 1``;
* ``gumbel``                    -> ``-log(-log(u))``;
* ``choice(..., replace=False, p)`` -> the Gumbel top-k trick: ``argsort(-gumbel - log(p))[:k]`` (stable sort).

Pinned by the Random123 known-answer vectors of Threefry-2x32-20 (the vectors jax's own ``random_test.py`` checks;
``tests/test_host.py``).  NOT pinned against jax itself -- the surrounding steps follow the release's source as
restated above, and ``log`` is the platform's: two keys closer than one ulp could swap.  The same draw runs on the
device (``nngp_pool_select``, csrc/posterior.hip) and in the host build of the ABI (the checker's C restatement of include/nngp_hip.h).
"""
from __future__ import annotations

import numpy as np

_ROT = ((13, 15, 26, 6), (17, 29, 16, 24))


def _rotl(x, r):
    return ((x << np.uint32(r)) | (x >> np.uint32(32 - r))).astype(np.uint32)


def threefry2x32(key, x0, x1):
    """Threefry-2x32, 20 rounds (Salmon et al., SC'11; Random123 ``threefry2x32_R(20, ...)``): uint32 arrays in, out."""
    k0, k1 = np.uint32(key[0]), np.uint32(key[1])
    ks = (k0, k1, np.uint32(k0 ^ k1 ^ np.uint32(0x1BD11BDA)))
    x0 = (np.asarray(x0, dtype=np.uint32) + ks[0]).astype(np.uint32)
    x1 = (np.asarray(x1, dtype=np.uint32) + ks[1]).astype(np.uint32)
    for g in range(5):
        for r in _ROT[g & 1]:
            x0 = (x0 + x1).astype(np.uint32)
            x1 = _rotl(x1, r) ^ x0
        x0 = (x0 + ks[(g + 1) % 3]).astype(np.uint32)
        x1 = (x1 + ks[(g + 2) % 3] + np.uint32(g + 1)).astype(np.uint32)
    return x0, x1


def prng_key(seed: int):
    seed = int(seed)
    return (np.uint32((seed >> 32) & 0xFFFFFFFF), np.uint32(seed & 0xFFFFFFFF))


def random_bits64(key, n: int) -> np.ndarray:
    with np.errstate(over="ignore"):
        y0, y1 = threefry2x32(key, np.arange(n, dtype=np.uint32), np.arange(n, 2 * n, dtype=np.uint64).astype(np.uint32))
    return (y0.astype(np.uint64) << np.uint64(32)) | y1.astype(np.uint64)


def uniform64(key, n: int) -> np.ndarray:
    """``jax.random.uniform(key, (n,), float64, minval=tiny, maxval=1)``."""
    tiny = np.finfo(np.float64).tiny
    bits = (random_bits64(key, n) >> np.uint64(12)) | np.uint64(0x3FF0000000000000)
    f = bits.view(np.float64) - 1.0
    return np.maximum(tiny, f * (1.0 - tiny) + tiny)


def gumbel64(key, n: int) -> np.ndarray:
    return -np.log(-np.log(uniform64(key, n)))


def choice_without_replacement(seed: int, n: int, k: int, p) -> np.ndarray:
    """``jax.random.choice(PRNGKey(seed), n, (k,), replace=False, p=p)`` in draw order."""
    p = np.asarray(p, dtype=np.float64)
    assert p.shape == (n,) and 0 <= k <= n
    with np.errstate(divide="ignore"):
        g = -gumbel64(prng_key(seed), n) - np.log(p)
    return np.argsort(g, kind="stable")[:k]
