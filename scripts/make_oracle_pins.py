#!/usr/bin/env python3
"""Independent pins for the CPU oracle -> tests/golden/oracle_pins.json (CPU only, ~10 min).

Nothing in here uses the closed forms the oracle restates (arc-cosine map, Theta recursion):

* ``finite_width``: random fully connected ReLU networks in the NTK parameterisation of ``stax.Dense``
  (z = W_std / sqrt(fan_in) * W h + b_std * b, W, b ~ N(0, 1); reference architecture train.py:161-164 widened to
  n_relu hidden layers), width 4096, 400 / 64 / 48 seeds for 1 / 2 / 3 hidden layers.  NNGP estimate = mean over seeds of E[f(x) f(x') | hidden layers]
  (the output layer's Gaussian weights integrated exactly: W_std^2 a.a'/width + b_std^2); NTK estimate =
  mean over seeds of the Jacobian inner product sum_theta df(x)/dtheta df(x')/dtheta, Jacobians by torch autograd.
  Stored: mean and standard error per entry.
* ``integral``: the two Gaussian expectations that define a ReLU layer, E[relu(u) relu(v)] and
  E[relu'(u) relu'(v)] for (u, v) ~ N(0, [[q1, k], [k, q2]]), evaluated with mpmath at 40 digits by nested
  quadrature of the DEFINITION (no arc-cosine formula), chained through the layers with
  K <- w^2 K + b^2, Theta <- K + w^2 Theta, Theta <- Theta * E[relu' relu'] (the chain rule of the NTK).

tests/test_oracle.py compares oracle.kernel_fn against both.  The reference itself holds no fixture for this path and
neural-tangents is absent, so this cannot turn parity "green" -- it removes "author checked author".
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "oracle_pins.json")

X6 = np.array([[1.0, 2.0, 3.0, 4.0, 0.5, 0.0, 1.5, 2.5],
               [4.0, 3.0, 2.0, 1.0, 0.0, 0.5, 2.5, 1.5],
               [0.0, 0.0, 0.0, 5.0, 0.0, 0.0, 0.0, 1.0],
               [-1.0, -2.0, -3.0, -4.0, 0.5, 0.0, -1.5, 2.5],
               [2.0, 2.0, 2.0, 2.0, 2.0, 2.0, 2.0, 2.0],
               [0.3, -0.7, 1.1, 0.0, -2.0, 0.9, 0.0, 0.4]])
X4 = np.array([[1, 2, 3, 4], [4, 3, 2, 1], [0, 0, 0, 1000], [-1, -2, -3, -4]], dtype=np.float64)  # SURVEY.md 8c KAT inputs


def finite_width(x, n_relu, w_std, b_std, width, seeds, seed0=0):
    import torch
    torch.set_num_threads(max(1, (os.cpu_count() or 2) - 1))
    xt = torch.tensor(x, dtype=torch.float64)
    n, d = xt.shape
    sizes = [d] + [width] * n_relu + [1]
    nn_acc, nn_sq = np.zeros((n, n)), np.zeros((n, n))
    tk_acc, tk_sq = np.zeros((n, n)), np.zeros((n, n))
    for s in range(seeds):
        g = torch.Generator().manual_seed(seed0 + s)
        ws = [torch.randn(sizes[i + 1], sizes[i], generator=g, dtype=torch.float64).requires_grad_() for i in range(len(sizes) - 1)]
        bs = [torch.randn(sizes[i + 1], generator=g, dtype=torch.float64).requires_grad_() for i in range(len(sizes) - 1)]
        h = xt
        for i, (w, b) in enumerate(zip(ws, bs)):
            if i == len(ws) - 1:
                last = h.detach()  # activations feeding the output layer
            h = (w_std / np.sqrt(sizes[i])) * h @ w.T + b_std * b
            if i < len(ws) - 1:
                h = torch.relu(h)
        f = h[:, 0]
        params = ws + (bs if b_std != 0.0 else [])
        jac = []
        for i in range(n):
            gr = torch.autograd.grad(f[i], params, retain_graph=True)
            jac.append(torch.cat([t.reshape(-1) for t in gr]))
        jac = torch.stack(jac)
        tk = (jac @ jac.T).numpy()
        # NNGP: E over the output layer's own weights of f(x) f(x') given the hidden activations (exact), Monte-Carlo over
        # the layers below: w_std^2 a.a' / width + b_std^2 -- concentrates like 1/sqrt(width) instead of O(1) per seed
        nn = (w_std ** 2) * (last @ last.T).numpy() / sizes[-2] + b_std ** 2
        nn_acc += nn; nn_sq += nn * nn; tk_acc += tk; tk_sq += tk * tk
    def stats(acc, sq):
        mean = acc / seeds
        var = np.maximum(sq / seeds - mean * mean, 0.0) * seeds / max(seeds - 1, 1)
        return mean.tolist(), np.sqrt(var / seeds).tolist()
    nm, ne = stats(nn_acc, nn_sq)
    tm, te = stats(tk_acc, tk_sq)
    return {"x": np.asarray(x).tolist(), "n_relu": n_relu, "w_std": w_std, "b_std": b_std, "width": width, "seeds": seeds,
            "nngp_mean": nm, "nngp_stderr": ne, "ntk_mean": tm, "ntk_stderr": te}


def relu_expectations(q1, q2, k, mp):
    """(E[relu(u) relu(v)], E[step(u) step(v)]) for (u, v) ~ N(0, [[q1, k], [k, q2]]) by nested quadrature."""
    q1, q2, k = mp.mpf(q1), mp.mpf(q2), mp.mpf(k)
    if q1 == 0 or q2 == 0:
        return mp.mpf(0), mp.mpf(1) / 4  # degenerate: relu(0) = 0; relu'(0) convention = 1/2 each side
    c = k / mp.sqrt(q1 * q2)
    c = max(min(c, mp.mpf(1)), mp.mpf(-1))
    s = mp.sqrt(1 - c * c)
    phi = lambda z: mp.exp(-z * z / 2) / mp.sqrt(2 * mp.pi)
    if s == 0:  # collinear: v = sign(c) sqrt(q2 / q1) u
        if c > 0:
            return mp.sqrt(q1 * q2) * mp.quad(lambda z: z * z * phi(z), [0, 1, 4, 12, 40]), mp.mpf(1) / 2
        return mp.mpf(0), mp.mpf(0)
    # u = sqrt(q1) z1 > 0 <=> z1 > 0;  v = sqrt(q2) (c z1 + s z2) > 0 <=> z2 > lo(z1) = -c z1 / s.  The inner Gaussian
    # integrals over z2 are elementary: int_lo^inf phi = erfc(lo / sqrt 2) / 2 =: Q(lo), int_lo^inf z2 phi = phi(lo);
    # the outer one is done by quadrature (no arc-cosine closed form anywhere).
    Q = lambda t: mp.erfc(t / mp.sqrt(2)) / 2
    pts = [0, 1, 4, 12, 40]
    rr = mp.sqrt(q1 * q2) * mp.quad(lambda z1: z1 * phi(z1) * (c * z1 * Q(-c * z1 / s) + s * phi(-c * z1 / s)), pts)
    ss = mp.quad(lambda z1: phi(z1) * Q(-c * z1 / s), pts)
    return rr, ss


def integral_kernel(x, n_relu, w_std, b_std, pairs, digits=40):
    import mpmath as mp
    mp.mp.dps = digits
    d = x.shape[1]
    out = []
    for (i, j) in pairs:
        xi, xj = [mp.mpf(float(v)) for v in x[i]], [mp.mpf(float(v)) for v in x[j]]
        k = sum(a * b for a, b in zip(xi, xj)) / d
        q1 = sum(a * a for a in xi) / d
        q2 = sum(b * b for b in xj) / d
        t = mp.mpf(0)
        w2, b2 = mp.mpf(w_std) ** 2, mp.mpf(b_std) ** 2
        for layer in range(n_relu + 1):
            k, q1, q2 = w2 * k + b2, w2 * q1 + b2, w2 * q2 + b2
            t = k + w2 * t
            if layer < n_relu:
                rr, ss = relu_expectations(q1, q2, k, mp)
                r1, _ = relu_expectations(q1, q1, q1, mp)
                r2, _ = relu_expectations(q2, q2, q2, mp)
                k, t, q1, q2 = rr, t * ss, r1, r2
        out.append({"i": i, "j": j, "nngp": mp.nstr(k, 30), "ntk": mp.nstr(t, 30)})
    return {"x": np.asarray(x).tolist(), "n_relu": n_relu, "w_std": w_std, "b_std": b_std, "digits": digits, "entries": out}


def main():
    width = int(os.environ.get("PIN_WIDTH", "4096"))
    # seeds per depth: a seed of the 3-hidden-layer net costs two 4096 x 4096 Gaussian matrices and 6 backward passes (~10 s)
    seeds = {1: int(os.environ.get("PIN_SEEDS_1", "400")), 2: int(os.environ.get("PIN_SEEDS_2", "64")), 3: int(os.environ.get("PIN_SEEDS_3", "48"))}
    pins = {"generator": "scripts/make_oracle_pins.py", "finite_width": [], "integral": []}
    t0 = time.time()
    for n_relu in (1, 3):
        for b_std in (0.0, 0.3):
            pins["finite_width"].append(finite_width(X6, n_relu, 1.0, b_std, width, seeds[n_relu], seed0=1000 * n_relu + int(10 * b_std)))
            print("finite width n_relu=%d b=%.1f done at %.0f s" % (n_relu, b_std, time.time() - t0), flush=True)
    pins["finite_width"].append(finite_width(X6, 2, 1.4, 0.2, width, seeds[2], seed0=77))
    pairs = [(0, 0), (0, 1), (0, 2), (0, 3), (1, 3), (2, 3)]
    for n_relu in (1, 3):
        for b_std in (0.0, 0.3):
            pins["integral"].append(integral_kernel(X4, n_relu, 1.0, b_std, pairs))
            print("integral n_relu=%d b=%.1f done at %.0f s" % (n_relu, b_std, time.time() - t0), flush=True)
    pins["integral"].append(integral_kernel(X6, 2, 1.4, 0.2, [(0, 1), (1, 5), (3, 4)]))
    with open(OUT, "w") as f:
        json.dump(pins, f)
    print("wrote", OUT, "in %.0f s" % (time.time() - t0))


if __name__ == "__main__":
    sys.exit(main())
