"""Pins the CPU oracle (oracle/nngp_oracle.py and the C restatement) -- runs without a GPU.

The reference holds no golden vectors for this path ("parity unpinned"), so the oracle
is pinned by (i) the known-answer values of SURVEY.md 8c, (ii) the Cho-Saul arc-cosine identity,
(iii) a finite-width Monte-Carlo network, (iv) structural identities, (v) independent linear algebra.
"""
import os

import numpy as np
import pytest
import scipy.linalg

import c_oracle
import nngp_oracle as o

X4 = np.array([[1, 2, 3, 4], [4, 3, 2, 1], [0, 0, 0, 1000], [-1, -2, -3, -4]], dtype=np.float64)


def test_kat_nngp_ntk_one_relu():
    a = o.make_arch(1)
    K = o.kernel_fn(X4, None, "nngp", a)
    T = o.kernel_fn(X4, None, "ntk", a)
    np.testing.assert_allclose(K[0], [3.75, 2.7204019972683966, 529.18491951711405, 0.0], rtol=1e-14, atol=1e-13)
    np.testing.assert_allclose(np.diag(K), [3.75, 3.75, 125000, 3.75], rtol=1e-14)
    np.testing.assert_allclose(K[1, 3], 0.2204019972683966, rtol=1e-13)
    np.testing.assert_allclose(K[2, 3], 29.184919517114082, rtol=1e-13)
    np.testing.assert_allclose(T[0], [7.5, 4.5511008152653218, 909.49402191888396, 0.0], rtol=1e-14, atol=1e-13)
    np.testing.assert_allclose(T[1, 3], -0.44889918473467816, rtol=1e-13)


def test_kat_three_relu():
    a = o.make_arch(3)
    np.testing.assert_allclose(o.kernel_fn(X4, None, "nngp", a)[0],
                               [0.9375, 0.7527212742689593, 142.36081896611202, 0.46287289706284834], rtol=1e-13)
    np.testing.assert_allclose(o.kernel_fn(X4, None, "ntk", a)[0],
                               [3.75, 1.9872369045510607, 394.24645589889656, 0.64285184651525862], rtol=1e-13)


def test_kat_posterior():
    post = o.Posterior(X4[:3], [1, 2, 3], o.make_arch(1), diag_reg=1e-3)
    xt = np.array([[2, 2, 2, 2], [1, 0, 0, 0]], dtype=np.float64)
    m, c = post.predict(xt, "nngp", True)
    np.testing.assert_allclose(m.ravel(), [0.11370915795217168, 0.02263792470241081], rtol=1e-12)
    np.testing.assert_allclose(c, [[1.1519693646083808, 0.18694115108475184], [0.18694115108475184, 0.1078032110983804]], rtol=1e-12)
    m, c = post.predict(xt, "ntk", True)
    np.testing.assert_allclose(m.ravel(), [0.12412394197183788, 0.02119708767313084], rtol=1e-12)
    np.testing.assert_allclose(c, [[1.0623580088157483, 0.17640899132719248], [0.17640899132719248, 0.10678768603197643]], rtol=1e-12)


def test_cho_saul_identity():
    rng = np.random.default_rng(1)
    x = rng.normal(size=(40, 7)) * rng.uniform(0.1, 30, size=(40, 1))
    K = o.kernel_fn(x, None, "nngp", o.make_arch(1))
    ref = 0.5 * o.cho_saul_arccos1(x, x) / x.shape[1]
    assert np.abs(K - ref).max() <= 1e-14 * np.abs(K).max()


def test_monte_carlo_finite_width():
    rng = np.random.default_rng(2)
    x = rng.uniform(0, 1, size=(6, 5))
    K = o.kernel_fn(x, None, "nngp", o.make_arch(1))
    mc = o.monte_carlo_relu_nngp(x, x, width=400000, seed=3)
    assert np.abs(K - mc).max() / np.abs(K).max() < 5e-3


def test_structural_identities():
    rng = np.random.default_rng(3)
    x = rng.uniform(0, 1000, size=(30, 6))
    for n_relu in (1, 2, 4):
        a = o.make_arch(n_relu)
        K = o.kernel_fn(x, None, "nngp", a)
        q = np.sum(x * x, axis=1) / x.shape[1]
        np.testing.assert_allclose(np.diag(K), q / 2 ** n_relu, rtol=1e-12)          # diag halves per ReLU
        np.testing.assert_allclose(K, K.T, rtol=0, atol=1e-9 * np.abs(K).max())
        assert np.linalg.eigvalsh(K).min() >= -1e-9 * np.trace(K)
        kd, td = o.diag_kernel(q, a)
        np.testing.assert_allclose(kd, np.diag(K), rtol=1e-12)
        np.testing.assert_allclose(td, np.diag(o.kernel_fn(x, None, "ntk", a)), rtol=1e-7)
    a = o.make_arch(1)
    u, v = np.array([[1.0, 2.0, 0.0]]), np.array([[-2.0, -4.0, 0.0]])
    assert abs(o.kernel_fn(u, v, "nngp", a)[0, 0]) < 1e-12                             # antiparallel -> 0
    e1, e2 = np.array([[3.0, 0.0]]), np.array([[0.0, 5.0]])
    np.testing.assert_allclose(o.kernel_fn(e1, e2, "nngp", a)[0, 0], np.sqrt(4.5 * 12.5) / (2 * np.pi), rtol=1e-13)
    z = np.zeros((1, 3))
    assert o.kernel_fn(z, u, "nngp", a)[0, 0] == 0.0 and o.kernel_fn(z, u, "ntk", a)[0, 0] == 0.0
    # homogeneity: K(l x, l x') = l^2 K(x, x') with b_std = 0 (SURVEY 8a notes)
    np.testing.assert_allclose(o.kernel_fn(3 * x, None, "nngp", a), 9 * o.kernel_fn(x, None, "nngp", a), rtol=1e-12)


def test_posterior_interpolates_as_reg_vanishes():
    rng = np.random.default_rng(4)
    x = rng.uniform(0, 1, size=(25, 4))
    y = rng.normal(size=(25, 1))
    m = o.Posterior(x, y, o.make_arch(1), diag_reg=1e-12).predict(None, "nngp", False)
    np.testing.assert_allclose(m, y, atol=1e-5)


def test_posterior_against_plain_solve():
    rng = np.random.default_rng(5)
    x = rng.uniform(0, 1000, size=(60, 8)); y = rng.normal(size=(60, 1)); xt = rng.uniform(0, 1000, size=(9, 8))
    a = o.make_arch(2, 1.3, 0.2)
    K = o.kernel_fn(x, None, "nngp", a)
    A = K + 1e-3 * np.trace(K) / 60 * np.eye(60)
    ktd = o.kernel_fn(xt, x, "nngp", a)
    mean = ktd @ np.linalg.solve(A, y)
    cov = o.kernel_fn(xt, None, "nngp", a) - ktd @ np.linalg.solve(A, ktd.T)
    m2, c2 = o.Posterior(x, y, a, diag_reg=1e-3).predict(xt, "nngp", True)
    np.testing.assert_allclose(m2, mean, rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(c2, cov, rtol=1e-8, atol=1e-9 * np.abs(cov).max())


@pytest.mark.parametrize("n,d,n_relu,w,b", [(130, 20, 1, 1.0, 0.0), (97, 64, 3, 1.0, 0.0), (64, 5, 2, 1.5, 0.3)])
def test_c_oracle_matches_numpy(n, d, n_relu, w, b):
    rng = np.random.default_rng(n)
    x = rng.uniform(0, 1000, size=(n, d)); y = rng.normal(size=(n, 1)) * 4; xt = rng.uniform(0, 1000, size=(17, d))
    a = o.make_arch(n_relu, w, b)
    K = o.kernel_fn(x, None, "nngp", a)
    np.testing.assert_allclose(c_oracle.kernel_build(x, None, "nngp", a.w_std, a.b_std), K, rtol=1e-12, atol=1e-12 * np.abs(K).max())
    Kr = o.kernel_fn(xt, x, "ntk", a)
    np.testing.assert_allclose(c_oracle.kernel_build(xt, x, "ntk", a.w_std, a.b_std), Kr, rtol=1e-9, atol=1e-12 * np.abs(Kr).max())
    A = K + 1e-3 * np.trace(K) / n * np.eye(n)
    L, info = c_oracle.potrf_lower(A)
    assert info == 0
    np.testing.assert_allclose(L, scipy.linalg.cholesky(A, lower=True), rtol=1e-9, atol=1e-11 * np.abs(A).max())
    model = c_oracle.fit(x, y, a.w_std, a.b_std)
    post = o.Posterior(x, y, a, diag_reg=1e-3)
    m, c = post.predict(xt, "nngp", True)
    mc, vc = c_oracle.predict_nngp(model, xt, 1)
    np.testing.assert_allclose(mc, m, rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(vc, np.diag(c), rtol=1e-7, atol=1e-9 * np.abs(c).max())
    _, cc = c_oracle.predict_nngp(model, xt, 2)
    np.testing.assert_allclose(cc, c, rtol=1e-7, atol=1e-9 * np.abs(c).max())


def test_c_oracle_detects_indefinite():
    a = np.array([[1.0, 2.0], [2.0, 1.0]])
    _, info = c_oracle.potrf_lower(a)
    assert info == 2


def test_golden_forest_fixture_reproduces(golden_dir):
    g = np.load(os.path.join(golden_dir, "forest_n256_m64.npz"))
    a = o.make_arch(1)
    post = o.Posterior(g["X_train"], g["Y_train"], a, diag_reg=1e-3)
    m, c = post.predict(g["X_test"], "nngp", True)
    np.testing.assert_allclose(m, g["nngp_mean"], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(np.diag(c), g["nngp_var"], rtol=1e-7)
    mt = post.predict(g["X_test"], "ntk", False)
    np.testing.assert_allclose(mt, g["ntk_mean"], rtol=1e-9, atol=1e-9)
    model = c_oracle.fit(g["X_train"], g["Y_train"], a.w_std, a.b_std)
    mc, vc = c_oracle.predict_nngp(model, g["X_test"], 1)
    np.testing.assert_allclose(mc, g["nngp_mean"], rtol=1e-8, atol=1e-8)
    np.testing.assert_allclose(vc, g["nngp_var"], rtol=1e-6)


# ---- independent pins (tests/golden/oracle_pins.json, scripts/make_oracle_pins.py): nothing below was produced by the
# ---- closed forms the oracle restates ----
def _pins(golden_dir):
    import json
    return json.load(open(os.path.join(golden_dir, "oracle_pins.json")))


def test_finite_width_networks_pin_nngp_and_ntk(golden_dir):
    """Random ReLU networks of width 4096 in the NTK parameterisation of stax.Dense (train.py:161-164 widened to n_relu
    hidden layers; W_std, b_std incl. bias): the mean over seeds of the Jacobian inner product (torch autograd) is the
    empirical NTK, the mean of W_std^2 a.a'/width + b_std^2 the NNGP kernel.  The oracle's closed forms must lie within
    4.5 standard errors (+ the O(1/width) finite-width bias) of both, entry by entry."""
    for case in _pins(golden_dir)["finite_width"]:
        x = np.array(case["x"])
        a = o.make_arch(case["n_relu"], case["w_std"], case["b_std"])
        for get in ("nngp", "ntk"):
            K = o.kernel_fn(x, None, get, a)
            mean, se = np.array(case[get + "_mean"]), np.array(case[get + "_stderr"])
            bias = 3.0 / case["width"] * np.sqrt(np.outer(np.diag(K), np.diag(K)))
            dev = np.abs(K - mean) / (4.5 * se + bias)
            assert dev.max() < 1.0, (case["n_relu"], case["b_std"], get, dev.max())
            # and the estimate is sharp enough to mean something: the typical standard error is ~1 % of the kernel's scale or better
            assert np.median(se) < 0.02 * np.abs(K).max(), (get, np.median(se), np.abs(K).max())


def test_gaussian_integral_definition_pins_the_closed_form(golden_dir):
    """E[relu(u) relu(v)] and E[relu'(u) relu'(v)] by 40-digit quadrature of the Gaussian integrals that DEFINE a ReLU
    layer (mpmath), chained through the layers with the NTK chain rule -- no arc-cosine formula involved.  The oracle
    agrees to float64 rounding, for 1 and 3 hidden layers, with and without bias, and for W_std != 1."""
    for case in _pins(golden_dir)["integral"]:
        x = np.array(case["x"])
        a = o.make_arch(case["n_relu"], case["w_std"], case["b_std"])
        K, T = o.kernel_fn(x, None, "nngp", a), o.kernel_fn(x, None, "ntk", a)
        scale_k, scale_t = np.abs(K).max(), np.abs(T).max()
        for e in case["entries"]:
            i, j = e["i"], e["j"]
            assert abs(K[i, j] - float(e["nngp"])) <= 4e-15 * max(abs(float(e["nngp"])), 1e-3 * scale_k), (case["n_relu"], case["b_std"], e, K[i, j])
            assert abs(T[i, j] - float(e["ntk"])) <= 4e-15 * max(abs(float(e["ntk"])), 1e-3 * scale_t), (case["n_relu"], case["b_std"], e, T[i, j])
            # the C restatement too
        Kc = c_oracle.kernel_build(x, None, "nngp", a.w_std, a.b_std)
        Tc = c_oracle.kernel_build(x, None, "ntk", a.w_std, a.b_std)
        for e in case["entries"]:
            assert abs(Kc[e["i"], e["j"]] - float(e["nngp"])) <= 1e-13 * scale_k and abs(Tc[e["i"], e["j"]] - float(e["ntk"])) <= 1e-13 * scale_t


def test_pin_generator_runs_live_at_small_width():
    """The committed fixture's generator, run here at a size that takes seconds (width 512, 12 seeds; one integral entry):
    looser, but shows the fixture is reproducible from scripts/make_oracle_pins.py."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_oracle_pins", os.path.join(os.path.dirname(__file__), "..", "scripts", "make_oracle_pins.py"))
    P = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(P)
    r = P.finite_width(P.X6, 2, 1.0, 0.3, 512, 12, seed0=5)
    a = o.make_arch(2, 1.0, 0.3)
    for get in ("nngp", "ntk"):
        K = o.kernel_fn(P.X6, None, get, a)
        mean, se = np.array(r[get + "_mean"]), np.array(r[get + "_stderr"])
        assert (np.abs(K - mean) / (5 * se + 0.02 * np.abs(K).max())).max() < 1.0
    g = P.integral_kernel(P.X4, 1, 1.0, 0.0, [(0, 1)], digits=30)
    assert abs(float(g["entries"][0]["nngp"]) - 2.7204019972683966) < 1e-14 and abs(float(g["entries"][0]["ntk"]) - 4.5511008152653218) < 1e-14
