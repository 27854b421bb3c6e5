"""The C ABI of include/nngp_hip.h compiled for the host (oracle/libnngp_cpu.so, test infrastructure): same symbols, same
prototypes, same error behaviour as the HIP library, float64 results equal to the NumPy oracle's -- so the parity tests can
drive both builds through one interface (tests/test_gpu_api.py does), and bench.py times this build as its cpu_baseline.
Reference call sites of the path: train.py:157-172, estimator.py:27-67."""
import ctypes
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nngp_src_amd import _lib, synth  # noqa: E402
from oracle import c_abi  # noqa: E402
from oracle import nngp_oracle as o  # noqa: E402


def test_host_build_exports_the_whole_abi():
    lib = c_abi.lib()
    assert lib.nngp_version() == 1
    for name in _lib.ABI_SYMBOLS:
        assert hasattr(lib, name), name
    assert not hasattr(lib, "nngp_debug_set")


@pytest.mark.parametrize("get", ["nngp", "ntk"])
@pytest.mark.parametrize("n,d,n_relu,w,b,absolute", [(130, 20, 1, 1.0, 0.0, False), (97, 33, 3, 1.2, 0.1, False), (64, 5, 2, 1.5, 0.3, True)])
def test_host_build_against_the_numpy_oracle(get, n, d, n_relu, w, b, absolute):
    x, y = synth.synthetic_queries(n, d, seed=n)
    xt, _ = synth.synthetic_queries(19, d, seed=n + 1)
    a = o.make_arch(n_relu, w, b)
    reg = 1e-3 if not absolute else 30.0
    post = o.Posterior(x, y, a, diag_reg=reg, diag_reg_absolute_scale=absolute)
    mean, cov = post.predict(xt, get, True)
    m = c_abi.CpuModel(n, d, a.w_std, a.b_std, get=get, diag_reg=reg, diag_reg_absolute_scale=absolute).fit(x, y)
    info = m.info()
    assert info["n"] == n and info["rel_residual"] < 1e-10
    k = o.kernel_fn(x, None, get, a)
    assert info["reg"] == pytest.approx(reg * (1.0 if absolute else np.trace(k) / n), rel=1e-8)
    # (the C and the NumPy restatement of the NTK agree to ~1e-9 per entry, tests/test_oracle.py; cond(A) carries that into the solve)
    tm, tv = (1e-8, 1e-6) if get == "nngp" else (1e-6, 1e-5)
    mean_c, var_c = m.predict(xt, "diag")
    np.testing.assert_allclose(mean_c, mean, rtol=tm, atol=tm * np.abs(mean).max())
    np.testing.assert_allclose(var_c, np.diag(cov), rtol=tv, atol=1e-9 * np.abs(cov).max())
    _, cov_c = m.predict(xt, "full")
    np.testing.assert_allclose(cov_c, cov, rtol=tv, atol=tv * 1e-2 * np.abs(cov).max())
    np.testing.assert_allclose(m.predict(xt, None), mean, rtol=tm, atol=tm * np.abs(mean).max())
    # x_test=None: the training rows (estimator.py:37-40)
    mean_tr = m.predict(None, None)
    np.testing.assert_allclose(mean_tr, post.predict(None, get, False), rtol=10 * tm, atol=10 * tm * np.abs(mean_tr).max())
    # stage-level entry points and the row-block shard of the build (train.py:166-168's slot)
    m2 = c_abi.CpuModel(n, d, a.w_std, a.b_std, get=get, diag_reg=reg, diag_reg_absolute_scale=absolute)
    m2.set_train(x, y)
    half = n // 2
    m2.build_rows(0, half)
    m2.build_rows(half, n)
    m2.factor()
    m2.solve()
    np.testing.assert_allclose(m2.alpha(), m.alpha(), rtol=1e-9, atol=1e-12 * np.abs(m.alpha()).max())
    kr = c_abi.kernel_build(xt, x, get, a.w_std, a.b_std, rows=(3, 11))
    np.testing.assert_allclose(kr[3:11], o.kernel_fn(xt, x, get, a)[3:11], rtol=1e-8)
    assert np.isnan(kr[:3]).all() and np.isnan(kr[11:]).all()
    k32 = c_abi.kernel_build(x, None, get, a.w_std, a.b_std, dtype=np.float32)
    np.testing.assert_allclose(k32, k.astype(np.float32), rtol=2e-7)


def test_host_build_error_behaviour_matches_the_header():
    lib = c_abi.lib()
    arch = _lib.make_arch([1.0, 1.0], [0.0, 0.0])
    h = ctypes.c_void_p()
    assert lib.nngp_model_create(ctypes.byref(h), 16, 4, 3, 1, ctypes.byref(arch), 7, 1e-3, 0) == -2  # bad `get`
    assert b"get" in lib.nngp_last_error()
    assert lib.nngp_model_create(ctypes.byref(h), 16, 4, 3, 1, ctypes.byref(arch), 1, 1e-3, 0) == 0
    out = np.zeros(4)
    assert lib.nngp_model_predict(h, None, 0, 1, ctypes.c_void_p(out.ctypes.data), ctypes.c_void_p(out.ctypes.data), None) == -2
    assert b"fit first" in lib.nngp_last_error()
    x = np.ones((17, 3))
    assert lib.nngp_model_set_train(h, ctypes.c_void_p(x.ctypes.data), ctypes.c_void_p(x.ctypes.data), 17, None) == -2  # > n_cap
    # what has no CPU counterpart is exported and says so
    assert lib.nngp_model_append(h, None, None, 1, None) == -2 and b"not in the CPU build" in lib.nngp_last_error()
    assert lib.nngp_model_prepare_serving(h, None) == -2
    assert lib.nngp_gemm_nt_f32(None, 0, None, 0, None, 0, 0, 0, 0, 1.0, 0.0, 0, None) == -2
    assert lib.nngp_model_destroy(h) == 0
    # an indefinite "kernel" (negative absolute regulariser is refused before it gets there)
    assert lib.nngp_model_create(ctypes.byref(h), 16, 4, 3, 1, ctypes.byref(arch), 1, -1.0, 1) == -2


def test_host_build_carries_the_line_encoder(golden_dir):
    """nngp_encoder_* is host code: the host build compiles the product's encoder.cpp as it is."""
    lib = c_abi.lib()
    schema = "T\ta:n:0:10\tb:c:4\n"
    h = ctypes.c_void_p()
    rc = lib.nngp_encoder_create(ctypes.byref(h), schema.encode(), 8, 0)
    if rc != 0:  # schema grammar differs: the symbol and its error path are what this test is about
        assert lib.nngp_last_error()
        return
    assert lib.nngp_encoder_dim(h) > 0
    assert lib.nngp_encoder_destroy(h) == 0


def test_pool_select_host_build_against_numpy():
    """nngp_pool_select (active/ActiveLearner.py:43-55): score = std / max(mean); the top `count` scores in ascending order are
    np.argsort(score)[-count:], and the score-proportional draw without replacement is the reference's
    jax.random.choice(PRNGKey(seed), m, (count,), replace=False, p=score / sum(score)) -- restated in NumPy in
    nngp_src_amd/jaxrand.py (Threefry-2x32-20 -> 52-bit uniforms -> Gumbel top-k); the host build must make the same draw."""
    from nngp_src_amd import jaxrand
    rng = np.random.default_rng(5)
    m, count = 700, 150
    mean = rng.uniform(0.5, 19.0, size=(m, 1))
    var = rng.uniform(0.0, 2.0, size=m) ** 2
    var[::97] = 0.0
    score = np.sqrt(var) / mean.max(0)
    got = c_abi.pool_select(mean, var, count)
    np.testing.assert_array_equal(got, np.argsort(score, kind="stable")[-count:])
    for seed in (10, 3, (7 << 32) | 5):
        want = jaxrand.choice_without_replacement(seed, m, count, score / score.sum())
        np.testing.assert_array_equal(c_abi.pool_select(mean, var, count, biased=True, seed=seed), want)
    # zero scores are never drawn while positive ones are left; the draw follows the scores
    sel = c_abi.pool_select(mean, var, m // 2, biased=True, seed=3)
    assert len(set(sel.tolist())) == m // 2 and score[sel].min() > 0.0 and score[sel].mean() > 1.2 * np.delete(score, sel).mean()
    assert c_abi.lib().nngp_pool_select(None, 5, 1, None, 2, 0, 0, None, None) != 0


def _host_gemm_i8s(a, b, cin, alpha, beta, sa, sb, cut):
    lib = c_abi.lib()
    m, k = a.shape
    n = b.shape[0]
    c = np.full((m, n), np.nan)
    vp = lambda arr: None if arr is None else arr.ctypes.data_as(ctypes.c_void_p)
    rc = lib.nngp_gemm_nt_i8s(vp(c), n, vp(cin), 0 if cin is None else cin.shape[1], vp(a), a.shape[1], vp(b), b.shape[1], m, n, k,
                              alpha, beta, sa, sb, cut, None)
    assert rc == 0, lib.nngp_last_error()
    return c


def test_host_sliced_int8_product():
    """The host restatement of csrc/gemm_i8s.hip (the posterior's residual product, train.py:157-158): exact on integers,
    float64-grade on random rows whose magnitudes spread over 2^18, error falling by ~2^-8 per diagonal."""
    rng = np.random.default_rng(3)
    m, n, k = 24, 40, 333
    a = rng.integers(-30000, 30001, (m, k)).astype(np.float64)
    b = rng.integers(-30000, 30001, (n, k)).astype(np.float64)
    c0 = rng.integers(-9, 10, (m, n)).astype(np.float64)
    got = _host_gemm_i8s(a, b, c0, -1.0, 1.0, 3, 3, 4)   # 3 planes hold 23 bits + sign; all 9 pairs
    assert np.array_equal(got, c0 - a @ b.T)
    a = rng.standard_normal((m, k)) * np.exp2(rng.integers(-18, 1, (m, k)))
    b = rng.standard_normal((n, k)) * np.exp2(rng.integers(-6, 1, (n, 1)))
    ref = (a.astype(np.longdouble) @ b.astype(np.longdouble).T).astype(np.float64)
    unit = np.abs(a).max(1)[:, None] * np.abs(b).max(1)[None, :]
    errs = []
    for sa, sb, cut in ((3, 3, 2), (4, 4, 3), (5, 5, 4), (6, 6, 5), (7, 7, 6)):
        got = _host_gemm_i8s(a, b, None, 1.0, 0.0, sa, sb, cut)
        errs.append(np.max(np.abs(got - ref) / unit))
    # dropped pairs: weight 256^-(cut+3), ~cut+2 of them, each a sum of k products of two digits (<= 2^14); scale^2 <= 16 unit
    for (cut, e) in zip((2, 3, 4, 5, 6), errs):
        assert e < (cut + 2) * np.sqrt(k) * 2.0 ** 14 * 256.0 ** -(cut + 3) * 16 + 4e-16, (cut, e)   # + the float64 rounding of the combination
    assert errs[2] < 1e-9 and errs[3] < 1e-11 and errs[4] < 1e-14
