"""80-bit (numpy longdouble) GP posterior variance: the referee between the HIP path and the float64 oracle on fits whose
condition number puts the float64 oracle's own rounding (cond * eps64, amplified by the cancellation in the variance)
near the test tolerance.  Test infrastructure, like everything under oracle/: never imported by the product.
The kernel matrices are the float64 oracle's (oracle/nngp_oracle.py kernel_fn); only the factorisation, the solves and the
variance sums run in extended precision.  Follows predict_fn(get, compute_cov=True): train.py:157-158 (NNGP form) and the
NTK ensemble covariance at t = inf (nngp_oracle.Posterior.predict)."""
import numpy as np

LD = np.longdouble


def cholesky_ld(a):
    """Lower Cholesky factor of a (float64 in, longdouble out), left-looking by columns."""
    n = a.shape[0]
    l = np.zeros((n, n), dtype=LD)
    a = a.astype(LD)
    for j in range(n):
        row = l[j, :j]
        l[j, j] = np.sqrt(a[j, j] - row @ row)
        if j + 1 < n:
            l[j + 1:, j] = (a[j + 1:, j] - l[j + 1:, :j] @ row) / l[j, j]
    return l


def solve_ld(l, b):
    """(L L^T)^-1 b for b [n, m] in longdouble."""
    n = l.shape[0]
    x = b.astype(LD).copy()
    for j in range(n):                      # forward: L y = b
        x[j] = (x[j] - l[j, :j] @ x[:j]) / l[j, j]
    for j in range(n - 1, -1, -1):          # backward: L^T x = y
        x[j] = (x[j] - l[j + 1:, j] @ x[j + 1:]) / l[j, j]
    return x


def posterior_variance_ld(a_dd, cross, prior_diag, nngp_dd=None, nngp_cross=None):
    """diag of the posterior covariance.  a_dd = kernel + reg I of `get` on the training rows, cross = kernel of `get`
    [M, N], prior_diag = NNGP K_tt diagonal.  NNGP: prior - diag(cross a_dd^-1 cross^T).  NTK: additionally the NNGP
    kernels on the training rows and across."""
    l = cholesky_ld(a_dd)
    z = solve_ld(l, cross.T)               # [N, M]
    cross_ld = cross.astype(LD)
    if nngp_dd is None:
        return prior_diag.astype(LD) - np.sum(cross_ld.T * z, axis=0)
    w = nngp_dd.astype(LD) @ z
    return prior_diag.astype(LD) + np.sum(z * w, axis=0) - 2 * np.sum(nngp_cross.astype(LD).T * z, axis=0)
