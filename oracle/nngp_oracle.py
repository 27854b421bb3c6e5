"""CPU oracle (NumPy float64) for the NNGP/NTK hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this module.  The product path (``nngp-src_amd``) never does; it fails
loudly when the HIP library is missing.

PARITY UNPINNED: the arithmetic of this path lives in a third-party dependency that
is absent from /root/reference: ``neural-tangents==0.6.1`` on ``jax==0.3.23`` /
``jaxlib==0.3.22`` (reference ``nngp.yaml:78-79,88``).  The reference ships no test,
golden vector or fixture for the path, and neither package
is installable here, so this file restates the *published* closed forms and is pinned
by the known-answer tests of SURVEY.md section 8c (Cho-Saul arc-cosine identity,
structural identities), not by neural-tangents output.

What is restated, and the reference call site each function stands behind:

* ``kernel_fn``  -- ``stax.serial(stax.Dense(512), stax.Relu(), stax.Dense(1))``'s
  ``kernel_fn(x1, x2, get)`` (reference ``train.py:161-164``, ``estimator.py:27-30``):
  input stage ``K0 = x1 x2^T / d``; Dense ``K <- w^2 K + b^2``, ``Theta <- K_new + w^2 Theta``;
  ReLU (``ABRelu(0, 1)``): ``s = sqrt(max(q1 q2 - K^2, 0))``, ``theta = atan2(s, K)``,
  ``K <- (s + (pi - theta) K) / 2pi``, ``Theta <- Theta (pi - theta) / 2pi``, ``q <- q / 2``.
* ``fit`` / ``predict`` -- ``nt.predict.gradient_descent_mse_ensemble(kernel_fn, X, Y,
  diag_reg=1e-3)`` and its ``predict_fn(x_test, get, compute_cov=True)``
  (reference ``train.py:171-172,157-158``; ``estimator.py:34-35,66-67``):
  ``A = K_dd + diag_reg * trace(K_dd)/N * I`` (relative regulariser),
  ``C = cho_factor(A)``, ``alpha = cho_solve(C, Y)``, ``mean = K_td alpha``,
  ``cov = K_tt - K_td A^-1 K_dt`` (nngp) or the NTK form documented in ``predict``.
"""
from __future__ import annotations

import collections
import math

import numpy as np
import scipy.linalg

Arch = collections.namedtuple("Arch", ["w_std", "b_std"])  # one entry per Dense layer


def make_arch(n_relu: int = 1, w_std=1.0, b_std=0.0) -> Arch:
    """Dense,(Relu,Dense)*n_relu with shared or per-layer std (reference: n_relu=1, W_std=1, b_std=0)."""
    nd = n_relu + 1
    w = [float(w_std)] * nd if np.isscalar(w_std) else [float(v) for v in w_std]
    b = [float(b_std)] * nd if np.isscalar(b_std) else [float(v) for v in b_std]
    assert len(w) == nd and len(b) == nd
    return Arch(tuple(w), tuple(b))


def diag_kernel(q, arch: Arch):
    """K(x, x) and Theta(x, x) from q = |x|^2/d alone (s = 0, theta = 0 on the diagonal)."""
    k = np.asarray(q, dtype=np.float64)
    t = np.zeros_like(k)
    nd = len(arch.w_std)
    for layer in range(nd):
        w2, b2 = arch.w_std[layer] ** 2, arch.b_std[layer] ** 2
        k = w2 * k + b2
        t = k + w2 * t
        if layer < nd - 1:
            k = 0.5 * k
            t = 0.5 * t
    return k, t


def kernel_fn(x1, x2=None, get="nngp", arch: Arch = None):
    """Closed-form kernel of Dense,(Relu,Dense)^n.  Follows train.py:161-164 (see module doc).

    x1: [N1, d], x2: [N2, d] or None (=> x2 = x1).  get in {'nngp','ntk',('nngp','ntk')}.
    """
    arch = arch or make_arch()
    x1 = np.asarray(x1, dtype=np.float64)
    sym = x2 is None
    x2 = x1 if sym else np.asarray(x2, dtype=np.float64)
    d = x1.shape[1]
    k = (x1 @ x2.T) / d
    q1 = np.sum(x1 * x1, axis=1) / d
    q2 = np.sum(x2 * x2, axis=1) / d
    t = np.zeros_like(k)
    nd = len(arch.w_std)
    for layer in range(nd):
        w2, b2 = arch.w_std[layer] ** 2, arch.b_std[layer] ** 2
        k = w2 * k + b2
        q1 = w2 * q1 + b2
        q2 = w2 * q2 + b2
        t = k + w2 * t
        if layer < nd - 1:
            prod = np.outer(q1, q2)
            s = np.sqrt(np.maximum(prod - k * k, 0.0))
            theta = np.arctan2(s, k)
            theta = np.where((s == 0.0) & (k == 0.0), np.pi / 2, theta)
            kdot = (np.pi - theta) / (2 * np.pi)
            k = s / (2 * np.pi) + kdot * k
            t = kdot * t
            q1 = 0.5 * q1
            q2 = 0.5 * q2
    if isinstance(get, (tuple, list)):
        return tuple({"nngp": k, "ntk": t}[g] for g in get)
    return {"nngp": k, "ntk": t}[get]


class Posterior:
    """Exact GP posterior; follows gradient_descent_mse_ensemble at train.py:171-172."""

    def __init__(self, x_train, y_train, arch: Arch = None, diag_reg: float = 1e-3,
                 diag_reg_absolute_scale: bool = False):
        self.arch = arch or make_arch()
        self.x = np.asarray(x_train, dtype=np.float64)
        self.y = np.asarray(y_train, dtype=np.float64).reshape(self.x.shape[0], -1)
        self.diag_reg = diag_reg
        self.absolute = diag_reg_absolute_scale
        self._cache = {}

    def _reg(self, k):
        n = k.shape[0]
        scale = 1.0 if self.absolute else np.trace(k) / n
        return k + self.diag_reg * scale * np.eye(n)

    def _factor(self, get):
        if get not in self._cache:
            k_dd = kernel_fn(self.x, None, get, self.arch)
            c = scipy.linalg.cho_factor(self._reg(k_dd), lower=True)
            alpha = scipy.linalg.cho_solve(c, self.y)
            self._cache[get] = (k_dd, c, alpha)
        return self._cache[get]

    def predict(self, x_test=None, get="nngp", compute_cov=True):
        """predict_fn(x_test=..., get=..., compute_cov=...) -- train.py:157-158."""
        k_dd, c, alpha = self._factor(get)
        if x_test is None:
            k_td = k_dd
            nngp_tt = k_dd if get == "nngp" else kernel_fn(self.x, None, "nngp", self.arch)
        else:
            x_test = np.asarray(x_test, dtype=np.float64)
            k_td = kernel_fn(x_test, self.x, get, self.arch)
            nngp_tt = kernel_fn(x_test, None, "nngp", self.arch)
        mean = k_td @ alpha
        if not compute_cov:
            return mean
        if get == "nngp":
            cov = nngp_tt - k_td @ scipy.linalg.cho_solve(c, k_td.T)
        else:
            # NTK ensemble covariance at t = inf:
            #   K_tt + Th_td Th~^-1 K_dd Th~^-1 Th_dt - (Th_td Th~^-1 K_dt + h.c.)
            nngp_dd = kernel_fn(self.x, None, "nngp", self.arch)
            nngp_td = nngp_dd if x_test is None else kernel_fn(x_test, self.x, "nngp", self.arch)
            z = scipy.linalg.cho_solve(c, k_td.T)            # Th~^-1 Th_dt   [N, M]
            cov = nngp_tt + z.T @ nngp_dd @ z - (nngp_td @ z + (nngp_td @ z).T)
        return mean, cov


# ----------------------------------------------------------------------------------------
# Independent cross-checks used by tests to pin the restated closed form.
# ----------------------------------------------------------------------------------------
def cho_saul_arccos1(x1, x2):
    """Degree-1 arc-cosine kernel of Cho & Saul (2009): (1/pi)|x||y| (sin t + (pi - t) cos t)."""
    x1 = np.asarray(x1, dtype=np.float64)
    x2 = np.asarray(x2, dtype=np.float64)
    n1 = np.linalg.norm(x1, axis=1)[:, None]
    n2 = np.linalg.norm(x2, axis=1)[None, :]
    cos = np.clip((x1 @ x2.T) / np.where(n1 * n2 == 0, 1.0, n1 * n2), -1.0, 1.0)
    th = np.arccos(cos)
    return (n1 * n2 / np.pi) * (np.sin(th) + (np.pi - th) * cos)


def monte_carlo_relu_nngp(x1, x2, width=200000, seed=0):
    """Finite-width check: E_w[relu(w.x/sqrt(d)) relu(w.x'/sqrt(d))] for w ~ N(0, I)."""
    rng = np.random.default_rng(seed)
    x1 = np.asarray(x1, dtype=np.float64)
    x2 = np.asarray(x2, dtype=np.float64)
    d = x1.shape[1]
    w = rng.standard_normal((d, width))
    h1 = np.maximum(x1 @ w / math.sqrt(d), 0.0)
    h2 = np.maximum(x2 @ w / math.sqrt(d), 0.0)
    return h1 @ h2.T / width
