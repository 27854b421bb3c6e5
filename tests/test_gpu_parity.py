"""Parity tests proper (-m gpu): the HIP path, called through the C ABI, against the float64 oracle.

Tolerances: kernel build is float64 arithmetic -> 1e-11 relative; posterior means must meet the
north-star gate of 1e-4 relative (we assert 1e-6: CG on the float64 kernel converges to the float64
answer); variances 1e-3 relative (float32 triangular solve); float32 MFMA GEMM is checked bit-exactly
on integer data and to float32 rounding on random data.
"""
import json
import time
import os

import numpy as np
import pytest
import scipy.linalg
import torch

import c_oracle
import nngp_oracle as o
from nngp_src_amd import synth
from nngp_src_amd.model import GPModel

import gpu_util as G

pytestmark = pytest.mark.gpu


# ---------------------------------------------------------------------------- kernel build (a1)
@pytest.mark.parametrize("n,d,n_relu,w,b", [(1, 1, 1, 1.0, 0.0), (63, 20, 1, 1.0, 0.0), (65, 3, 1, 1.0, 0.0),
                                            (130, 79, 3, 1.0, 0.0), (257, 64, 2, 1.4, 0.25), (200, 256, 1, 1.0, 0.0)])
def test_kernel_build_symmetric(n, d, n_relu, w, b):
    rng = np.random.default_rng(n * 7 + d)
    x = rng.uniform(0, 1000, size=(n, d))
    a = o.make_arch(n_relu, w, b)
    out = G.kernel_build(x, None, a.w_std, a.b_std)
    K, T = o.kernel_fn(x, None, ("nngp", "ntk"), a)
    assert not np.isnan(out["nngp"]).any() and not np.isnan(out["ntk"]).any()
    np.testing.assert_allclose(out["nngp"], K, rtol=1e-11, atol=1e-12 * np.abs(K).max())
    # off-diagonal NTK to 1e-9; the diagonal of the oracle carries sqrt(rounding) noise (theta ~ 1e-8), ours is exact
    np.testing.assert_allclose(out["ntk"], T, rtol=1e-7, atol=1e-9 * np.abs(T).max())
    assert np.array_equal(out["nngp"], out["nngp"].T) and np.array_equal(out["ntk"], out["ntk"].T)
    q = np.sum(x * x, axis=1) / d
    kd, td = o.diag_kernel(q, a)
    np.testing.assert_allclose(np.diag(out["nngp"]), kd, rtol=1e-13)
    np.testing.assert_allclose(np.diag(out["ntk"]), td, rtol=1e-13)


@pytest.mark.parametrize("n1,n2,d", [(5, 300, 20), (129, 64, 7), (64, 129, 128), (300, 1, 20)])
def test_kernel_build_rectangular_and_padded_ld(n1, n2, d):
    rng = np.random.default_rng(n1 + n2)
    x1, x2 = rng.uniform(0, 1000, size=(n1, d)), rng.uniform(0, 1000, size=(n2, d))
    a = o.make_arch(1)
    out = G.kernel_build(x1, x2, a.w_std, a.b_std, ld=n2 + 3)
    K, T = o.kernel_fn(x1, x2, ("nngp", "ntk"), a)
    np.testing.assert_allclose(out["nngp"][:, :n2], K, rtol=1e-11, atol=1e-12 * np.abs(K).max())
    np.testing.assert_allclose(out["ntk"][:, :n2], T, rtol=1e-9, atol=1e-12 * np.abs(T).max())
    assert np.isnan(out["nngp"][:, n2:]).all()  # padding columns untouched


def test_kernel_build_row_shard_equals_full():
    x, _ = synth.synthetic_queries(333, 64, seed=3)
    a = o.make_arch(3)
    full = G.kernel_build(x, None, a.w_std, a.b_std, get=("nngp",))["nngp"]
    parts = np.full_like(full, np.nan)
    for r0, r1 in [(0, 100), (100, 101), (101, 333)]:
        blk = G.kernel_build(x, None, a.w_std, a.b_std, get=("nngp",), rows=(r0, r1))["nngp"]
        assert np.isnan(blk[:r0]).all() and np.isnan(blk[r1:]).all()
        parts[r0:r1] = blk[r0:r1]
    np.testing.assert_allclose(parts, full, rtol=1e-13, atol=0)
    np.testing.assert_allclose(full, o.kernel_fn(x, None, "nngp", a), rtol=1e-11)


def test_kernel_build_edge_values():
    a = o.make_arch(1)
    x = np.array([[1, 2, 3, 4], [4, 3, 2, 1], [0, 0, 0, 1000], [-1, -2, -3, -4], [0, 0, 0, 0], [1, 2, 3, 4]], dtype=np.float64)
    out = G.kernel_build(x, None, a.w_std, a.b_std)
    np.testing.assert_allclose(out["nngp"][0, :4], [3.75, 2.7204019972683966, 529.18491951711405, 0.0], rtol=1e-13, atol=1e-12)
    np.testing.assert_allclose(out["ntk"][0, :4], [7.5, 4.5511008152653218, 909.49402191888396, 0.0], rtol=1e-13, atol=1e-12)
    assert np.all(out["nngp"][4] == 0) and np.all(out["ntk"][4] == 0)           # zero vector
    np.testing.assert_allclose(out["nngp"][0, 5], 3.75, rtol=1e-9)              # duplicate rows: s ~ sqrt(rounding)
    f32 = G.kernel_build(x, None, a.w_std, a.b_std, dtype=torch.float32)
    np.testing.assert_allclose(f32["nngp"], out["nngp"], rtol=2e-7, atol=1e-30)


@pytest.mark.parametrize("n_relu,w", [(2, 1.0), (3, 1.3), (4, 0.8)])
def test_kernel_build_composite_relu_map(n_relu, w):
    """Round 4 (reference op: kernel_fn of stax.serial(Dense, Relu, ..., Dense), train.py:161-164, whose Dense layers have no bias):
    without biases the layer recursion is sqrt(q q') A F(theta0) with ONE univariate function of the first layer's angle, which
    the kernel evaluates by a checked piecewise polynomial (kernel_build.hip: comp_table) instead of a sqrt and an arctangent per
    layer.  Entries against a 40-digit evaluation of the recursion itself -- random pairs, duplicate, nearly parallel, antiparallel,
    orthogonal and zero rows -- and against the per-layer path (timing-knob key 5 = 63); the diagonal is the closed form, bit for bit."""
    import mpmath as mp
    mp.mp.dps = 40
    rng = np.random.default_rng(17 + n_relu)
    x = rng.uniform(-1.0, 3.0, size=(260, 16))
    x[1] = x[0]
    x[2] = x[0] * (1.0 + 1e-9) + 1e-7 * rng.standard_normal(16)
    x[3] = -x[0]
    x[4] = 0.0
    x[5] = 0.0; x[5, 0] = 2.0
    x[6] = 0.0; x[6, 1] = 5.0
    a = o.make_arch(n_relu, w, 0.0)
    got = G.kernel_build(x, None, a.w_std, a.b_std, get=("nngp",), knobs=True)["nngp"]
    from nngp_src_amd import _lib as L
    L.load(knobs=True).nngp_debug_set(5, 63)
    try:
        layered = G.kernel_build(x, None, a.w_std, a.b_std, get=("nngp",), knobs=True)["nngp"]
    finally:
        L.load(knobs=True).nngp_debug_set(5, 0)
    assert np.array_equal(got, got.T) and np.array_equal(np.diag(got), np.diag(layered))
    scale = np.sqrt(np.outer(np.diag(got), np.diag(got)))
    assert np.max(np.abs(got - layered) / np.maximum(scale, 1e-300)) < 2e-13   # the per-layer path loses digits on nearly parallel rows
    assert np.all(got[4] == 0.0)

    def exact(i, j):
        q1, q2, k = (mp.fsum(mp.mpf(float(v)) ** 2 for v in x[i]) / 16, mp.fsum(mp.mpf(float(v)) ** 2 for v in x[j]) / 16,
                     mp.fsum(mp.mpf(float(u)) * mp.mpf(float(v)) for u, v in zip(x[i], x[j])) / 16)
        w2 = mp.mpf(float(a.w_std[0])) ** 2
        for l in range(n_relu + 1):
            q1, q2, k = w2 * q1, w2 * q2, w2 * k
            if l < n_relu:
                c = k / mp.sqrt(q1 * q2)
                th = mp.acos(max(mp.mpf(-1), min(mp.mpf(1), c)))
                k = mp.sqrt(q1 * q2) * (mp.sin(th) + (mp.pi - th) * mp.cos(th)) / (2 * mp.pi)
                q1, q2 = q1 / 2, q2 / 2
        return k
    pairs = [(1, 0), (2, 0), (3, 0), (6, 5), (5, 0), (7, 0)] + [tuple(rng.integers(7, 260, 2)) for _ in range(60)]
    worst = 0.0
    for i, j in pairs:
        if i == j:
            continue
        ref = exact(int(i), int(j))
        worst = max(worst, float(abs(mp.mpf(float(got[i, j])) - ref) / mp.mpf(float(scale[i, j]))))
    # the Gram entry itself carries ~1e-16 d of rounding; a duplicate / nearly parallel pair turns that into sqrt(rounding) of angle
    assert worst < 3e-14, worst


def test_kernel_build_empty_inputs():
    a = o.make_arch(1)
    out = G.kernel_build(np.zeros((0, 5)), None, a.w_std, a.b_std)
    assert out["nngp"].shape == (0, 0)


# ---------------------------------------------------------------------------- float32 MFMA GEMM
def test_gemm_exact_on_integers_asymmetric():
    """A = I against an asymmetric B, then small-integer operands: exact in float32 (catches lane-map swaps)."""
    torch.manual_seed(0)
    m, n, k = 256, 384, 128
    b = torch.randint(-8, 9, (n, k), device=G.dev()).float()
    a = torch.zeros((m, k), device=G.dev())
    a[torch.arange(128), torch.arange(128)] = 1.0
    c = torch.full((m, n), float("nan"), device=G.dev())
    G.gemm_nt(c, a, b, 1.0, 0.0)
    assert torch.equal(c[:128], b[:, :128].T.contiguous()) and torch.all(c[128:] == 0)
    a = torch.randint(-8, 9, (m, k), device=G.dev()).float()
    c0 = torch.randint(-8, 9, (m, n), device=G.dev()).float()
    c = c0.clone()
    G.gemm_nt(c, a, b, -1.0, 1.0)
    ref = c0.double() - a.double() @ b.double().T
    assert torch.equal(c.double(), ref)


def test_gemm_is_a_k_ordered_fp32_fma_chain():
    """f32-input MFMA is an exact fp32 fma chain (guide: MFMA numerics); emulate the kernel's k order on the host
    and require bit-for-bit equality.  Order: 32-wide K tiles from the high end of K down (small Cholesky terms
    first), inside a tile per 8-wide step s, per kk: k = 8s+kk then 8s+4+kk."""
    torch.manual_seed(5)
    m = n = 128
    k = 256
    a = torch.randn((m, k), device=G.dev()); b = torch.randn((n, k), device=G.dev())
    c = torch.zeros((m, n), device=G.dev())
    G.gemm_nt(c, a, b, 1.0, 0.0)
    got = c.cpu().numpy()
    an, bn = a.cpu().numpy(), b.cpu().numpy()

    def chain(order):
        acc = np.zeros((m, n), dtype=np.float32)
        for kk in order:
            prod = an[:, kk].astype(np.float64)[:, None] * bn[:, kk].astype(np.float64)[None, :]  # exact in f64
            acc = (acc.astype(np.float64) + prod).astype(np.float32)                               # one rounding (fma)
        return acc

    tiles = list(reversed(range(k // 32)))
    base = [32 * t + 8 * s + kk + 4 * h for t in tiles for s in range(4) for kk in range(4) for h in (0, 1)]
    alt = [32 * t + 8 * s + kk + 4 * h for t in tiles for s in range(4) for kk in range(4) for h in (1, 0)]
    ok = np.array_equal(got, chain(base)) or np.array_equal(got, chain(alt))
    if not ok:
        d0 = np.abs(got - chain(base)).max()
        pytest.fail("GEMM is not bit-identical to the emulated fma chain (max diff %g)" % d0)


@pytest.mark.parametrize("m,n,k", [(128, 128, 128), (384, 256, 512), (1024, 1024, 2048), (2048, 2048, 256), (1536, 1024, 128)])
def test_gemm_random(m, n, k):
    torch.manual_seed(1)
    a = torch.randn((m, k), device=G.dev()); b = torch.randn((n, k), device=G.dev()); c0 = torch.randn((m, n), device=G.dev())
    c = c0.clone()
    G.gemm_nt(c, a, b, -1.0, 1.0)
    ref = c0.double() - a.double() @ b.double().T
    bound = c0.abs().double() + a.abs().double() @ b.abs().double().T   # sum |terms| per element
    tol = 4 * (k + 1) ** 0.5 * 6e-8  # random-walk bound of a k-long fp32 fma chain, relative to sum |terms|
    assert ((c.double() - ref).abs() <= tol * bound).all(), ((c.double() - ref).abs() / bound).max().item()
    # strided views (sub-blocks of a larger matrix) and the SYRK form
    big = torch.randn((m + 128, k + 256), device=G.dev())
    av = big[128:, 256:]
    c = torch.zeros((m, m), device=G.dev())
    G.gemm_nt(c, av, av, 1.0, 0.0, lower_only=True)
    ref = (av.double() @ av.double().T)
    idx = torch.arange(m, device=G.dev())
    elem_lower = idx[:, None] >= idx[None, :]                 # what the Cholesky reads back
    tile_upper = (idx[:, None] // 128) < (idx[None, :] // 128)  # 128-tiles strictly above the diagonal
    bound = av.abs().double() @ av.abs().double().T
    assert ((c.double() - ref).abs() <= tol * bound)[elem_lower].all()
    assert torch.all(c[tile_upper] == 0)  # tiles above the diagonal are never touched


def test_gemm_split_f16_exact_on_integers():
    """Split-float16 GEMM: integers up to 2^11 are exact in the hi plane, so the product is exact in float32
    (asymmetric operands and A = I catch lane-map and tile-index mistakes; K blocks are walked high to low)."""
    torch.manual_seed(11)
    m, n, k = 384, 640, 96  # not multiples of the 256 tile: edge tiles, masked stores
    b = torch.randint(-8, 9, (n, k), device=G.dev()).float()
    a = torch.zeros((m, k), device=G.dev())
    a[torch.arange(96), torch.arange(96)] = 1.0
    c = torch.full((m, n), float("nan"), device=G.dev())
    G.gemm_nt_h3(c, a, b, 1.0, 0.0, 1.0)
    assert torch.equal(c[:96], b[:, :96].T.contiguous()) and torch.all(c[96:] == 0)
    a = torch.randint(-8, 9, (m, k), device=G.dev()).float()
    c0 = torch.randint(-8, 9, (m, n), device=G.dev()).float()
    c = c0.clone()
    G.gemm_nt_h3(c, a, b, -1.0, 1.0, 4.0)  # a power-of-two scale does not change an exact result
    assert torch.equal(c.double(), c0.double() - a.double() @ b.double().T)


@pytest.mark.parametrize("m,n,k", [(128, 128, 32), (1024, 768, 512), (2304, 2304, 1024)])
def test_gemm_split_f16_random(m, n, k):
    """Float32-grade accuracy of the three-product split (hi*hi + hi*lo + lo*hi, float32 accumulation) against float64,
    with row magnitudes spread over 2^-6..1; and the lower-only form touches exactly the 128-tiles on/below the diagonal."""
    torch.manual_seed(3)
    scale_rows = torch.exp2(torch.randint(-6, 1, (m, 1), device=G.dev()).float())
    a = torch.randn((m, k), device=G.dev()) * scale_rows
    b = torch.randn((n, k), device=G.dev())
    c0 = torch.randn((m, n), device=G.dev())
    c = c0.clone()
    G.gemm_nt_h3(c, a, b, -1.0, 1.0, 2.0 ** 10)
    ref = c0.double() - a.double() @ b.double().T
    bound = c0.abs().double() + a.abs().double() @ b.abs().double().T
    # operand truncation 2^-22 per factor + a k-long float32 accumulation, relative to sum |terms|
    tol = 2.0 ** -20 + 4 * (k + 1) ** 0.5 * 6e-8
    assert ((c.double() - ref).abs() <= tol * bound).all(), ((c.double() - ref).abs() / bound).max().item()
    if m == n:
        c = torch.zeros((m, m), device=G.dev())
        G.gemm_nt_h3(c, a, a, 1.0, 0.0, 2.0 ** 10, lower_only=True)
        ref = a.double() @ a.double().T
        idx = torch.arange(m, device=G.dev())
        elem_lower = idx[:, None] >= idx[None, :]
        tile_upper = (idx[:, None] // 128) < (idx[None, :] // 128)
        bound = a.abs().double() @ a.abs().double().T
        assert ((c.double() - ref).abs() <= tol * bound)[elem_lower].all()
        assert torch.all(c[tile_upper] == 0)


def test_gemm_sliced_int8_exact_on_integers():
    """Sliced int8 product (csrc/gemm_i8s.hip; the posterior's residual product, train.py:157-158): with A = I against an
    asymmetric B and with integer operands below 2^23 (three 8-bit planes, all nine pairs) every step is exact -- catches
    lane-map, digit and tile-index mistakes.  Edge tiles (m, n not multiples of 256), k not a multiple of 128."""
    torch.manual_seed(21)
    m, n, k = 384, 640, 200
    b = torch.randint(-30000, 30001, (n, k), device=G.dev()).double()
    a = torch.zeros((m, k), device=G.dev(), dtype=torch.float64)
    a[torch.arange(200), torch.arange(200)] = 1.0
    c = torch.full((m, n), float("nan"), device=G.dev(), dtype=torch.float64)
    G.gemm_nt_i8s(c, None, a, b, 1.0, 0.0, 3, 3, 4)
    assert torch.equal(c[:200], b[:, :200].T.contiguous()) and torch.all(c[200:] == 0)
    a = torch.randint(-30000, 30001, (m, k), device=G.dev()).double()
    c0 = torch.randint(-8, 9, (m, n), device=G.dev()).double()
    c = torch.full_like(c0, float("nan"))
    G.gemm_nt_i8s(c, c0, a, b, -1.0, 1.0, 3, 3, 4)
    assert torch.equal(c, c0 - a @ b.T)


def test_gemm_sliced_int8_is_the_host_restatement_bit_for_bit():
    """Every step of the sliced product is exact or one IEEE operation in a fixed order: the host restatement
    (oracle/nngp_cpu_abi.c) reproduces the device result bit for bit, on rows whose magnitudes spread over 2^18."""
    from oracle import c_abi
    import ctypes
    rng = np.random.default_rng(8)
    m, n, k = 256, 384, 700
    a = rng.standard_normal((m, k)) * np.exp2(rng.integers(-18, 1, (m, k)))
    b = rng.standard_normal((n, k)) * np.exp2(rng.integers(-6, 1, (n, 1)))
    c0 = rng.standard_normal((m, n))
    host = np.full((m, n), np.nan)
    lib = c_abi.lib()
    vp = lambda arr: arr.ctypes.data_as(ctypes.c_void_p)
    for sa, sb, cut in ((5, 5, 4), (4, 6, 3), (7, 7, 6)):
        assert lib.nngp_gemm_nt_i8s(vp(host), n, vp(c0), n, vp(a), k, vp(b), k, m, n, k, -1.0, 1.0, sa, sb, cut, None) == 0
        c = torch.full((m, n), float("nan"), device=G.dev(), dtype=torch.float64)
        G.gemm_nt_i8s(c, torch.from_numpy(c0).to(G.dev()), torch.from_numpy(a).to(G.dev()), torch.from_numpy(b).to(G.dev()),
                      -1.0, 1.0, sa, sb, cut)
        assert np.array_equal(c.cpu().numpy(), host), (sa, sb, cut, np.abs(c.cpu().numpy() - host).max())
    ref = c0 - (a.astype(np.longdouble) @ b.astype(np.longdouble).T).astype(np.float64)
    unit = np.abs(a).max(1)[:, None] * np.abs(b).max(1)[None, :]
    assert np.max(np.abs(host - ref) / unit) < 2e-14   # 7 x 7 planes, cut 6: float64 grade proper (what is left is the rounding of c0 - product)


def test_gemm_sliced_int8_k_chunks_cannot_overflow():
    """K beyond one chunk on the worst case the digits allow: every entry of a row equal (top digit 126 or 127), the lower
    digits -128 or 127 by choice of the value, all products of one sign -- the int32 accumulators hold up to 2^30.9 per chunk.
    The host restatement sums in int64: equality means no accumulator wrapped.  5 x 5 planes (up to 5 pairs on a diagonal): chunks
    of 16384; 3 x 5 planes (round 4: at most 3 pairs on a diagonal): ONE chunk of 40960, 3 x 40960 x 2^14 = 2^30.9."""
    from oracle import c_abi
    import ctypes
    m, n = 128, 256
    lib = c_abi.lib()
    vp = lambda arr: arr.ctypes.data_as(ctypes.c_void_p)
    for sa, k in ((5, 2 * 16384 + 640), (3, 40960)):
        for digits in ((126, -128, -128, -128, -128), (125, 127, 127, 127, 127)):
            # every entry = -(sum digits 256^-(i+1)) of the row scale 1 (row maximum 0.49.. = f 2^-1 with f <= 126/128)
            frac = sum(d * 256.0 ** -(i + 1) for i, d in enumerate(digits))
            a = np.full((m, k), -sum(d * 256.0 ** -(i + 1) for i, d in enumerate(digits[:sa])))
            b = np.full((n, k), -frac)
            b[5] *= 0.5
            host = np.full((1, 8), np.nan)
            assert lib.nngp_gemm_nt_i8s(vp(host), 8, None, 0, vp(a[:1].copy()), k, vp(b[:8].copy()), k, 1, 8, k, 1.0, 0.0, sa, 5, 4, None) == 0
            c = torch.full((m, n), float("nan"), device=G.dev(), dtype=torch.float64)
            G.gemm_nt_i8s(c, None, torch.from_numpy(a).to(G.dev()), torch.from_numpy(b).to(G.dev()), 1.0, 0.0, sa, 5, 4)
            got = c.cpu().numpy()
            assert np.all(got[:, np.arange(n) != 5] == host[0, 0]) and np.all(got[:, 5] == host[0, 5])
            assert abs(host[0, 0] - (a[0] @ b[0])) < 1e-9 * abs(a[0] @ b[0])


@pytest.mark.parametrize("m,n,k", [(128, 128, 64), (1024, 1280, 4096), (512, 2304, 20000)])
def test_gemm_sliced_int8_random(m, n, k):
    """Float64-grade accuracy of 5 x 5 planes with cut 4 against an 80-bit product, relative to the row maxima."""
    rng = np.random.default_rng(m + k)
    a = rng.standard_normal((m, k)) * np.exp2(rng.integers(-12, 1, (m, k)))
    b = rng.standard_normal((n, k)) * np.exp2(rng.integers(-6, 1, (n, 1)))
    at, bt = torch.from_numpy(a).to(G.dev()), torch.from_numpy(b).to(G.dev())
    c = torch.full((m, n), float("nan"), device=G.dev(), dtype=torch.float64)
    G.gemm_nt_i8s(c, None, at, bt, 1.0, 0.0)
    rows = rng.choice(m, 24, replace=False)
    ref = (a[rows].astype(np.longdouble) @ b.astype(np.longdouble).T).astype(np.float64)
    unit = np.abs(a[rows]).max(1)[:, None] * np.abs(b).max(1)[None, :]
    err = np.max(np.abs(c.cpu().numpy()[rows] - ref) / unit)
    assert err < 6 * np.sqrt(k) * 2.0 ** 14 * 256.0 ** -7 * 16, err   # six dropped pairs of weight 256^-7; scale^2 <= 16 unit
    # and against the float64 matrix pipe on the whole product
    c64 = torch.full_like(c, float("nan"))
    G.gemm_nt_f64(c64, None, at, bt, 1.0, 0.0) if k % 16 == 0 else None
    if k % 16 == 0:
        assert torch.max((c - c64).abs() / (at.abs().amax(1)[:, None] * bt.abs().amax(1)[None, :])).item() < 6 * np.sqrt(k) * 2.0 ** 14 * 256.0 ** -7 * 16


def test_gemm_split_f16_agrees_with_f32_mfma_in_the_cholesky():
    """The look-ahead Cholesky with split-float16 trailing updates against the same factorisation on the float32 MFMA
    (debug key 2 = 2): both factor the same matrix to float32 accuracy."""
    n = 4096
    x, y = synth.synthetic_queries(n, 32, seed=21)
    from nngp_src_amd import _lib
    lib = _lib.load(knobs=True)  # the A/B switch only exists in libnngp_hip_knobs.so
    res = {}
    for key2 in (0, 2):
        lib.nngp_debug_set(2, key2)
        try:
            mdl = GPModel(n, 32, [1.0, 1.0], [0.0, 0.0], diag_reg=1e-3, knobs=True).fit(x / 1000.0, y)
            res[key2] = (mdl.info(), mdl.alpha().cpu().numpy().copy())
            mdl.close()
        finally:
            lib.nngp_debug_set(2, 0)
    for key2 in (0, 2):
        info = res[key2][0]
        assert info["clamped_pivots"] == 0 and info["rel_residual"] < 1e-9 and info["refine_iters"] <= 8, info
    d = np.linalg.norm(res[0][1] - res[2][1]) / np.linalg.norm(res[2][1])
    assert d < 1e-8, d  # alpha is the float64 CG answer either way


@pytest.mark.parametrize("n", [9300, 12800])
def test_grouped_cholesky_matches_the_one_column_form(n):
    """Round 3's grouped look-ahead Cholesky (block columns in groups of 4: K = 1024 updates inside a group, ONE K = 4096 pass over
    the columns beyond it, issued as multi-region multi-panel split-float16 launches) against the same factorisation with one
    block column per group (debug key 2 = 11, the round-2 schedule): both are float32 factors of the same matrix -- the float64
    CG answer alpha agrees to 1e-9, neither clamps a pivot, the CG takes the same number of iterations (+-1), and two runs of
    the grouped form give the same bits (the multi-region launches pull tiles from shared counters in any order; the result
    must not depend on it)."""
    from nngp_src_amd import _lib
    x, y = synth.synthetic_queries(n, 48, seed=31)
    lib = _lib.load(knobs=True)
    res = {}
    for key2 in (11, 0, 0):
        lib.nngp_debug_set(2, key2)
        try:
            mdl = GPModel(n, 48, [1.0, 1.0, 1.0], [0.0, 0.0, 0.0], diag_reg=1e-3, knobs=True).fit(x, y)
            a32, _ = mdl.factor_buffers()
            res.setdefault(key2, []).append((mdl.info(), mdl.alpha().cpu().numpy().copy(), torch.tril(a32[:n, :n]).clone()))
            mdl.close()
        finally:
            lib.nngp_debug_set(2, 0)
    (i1, a1, l1), (i4, a4, l4), (_, a4b, l4b) = res[11][0], res[0][0], res[0][1]
    assert i1["clamped_pivots"] == 0 and i4["clamped_pivots"] == 0 and i4["rel_residual"] < 1e-9
    assert abs(i1["refine_iters"] - i4["refine_iters"]) <= 1, (i1, i4)
    assert np.linalg.norm(a1 - a4) <= 1e-9 * np.linalg.norm(a1)
    assert torch.equal(l4, l4b) and np.array_equal(a4, a4b)
    # the two factors are float32 factors of one matrix: they differ by rounding only
    assert (l1 - l4).abs().max().item() <= 2e-3 * l1.abs().max().item()


@pytest.mark.parametrize("n", [9300, 13440])
def test_cholesky_schedule_with_the_panel_solves_off_the_update_stream(n):
    """Round 4's experimental schedule of the grouped Cholesky (debug key 8 bit 1, knobs build only: panel solves on the panel / bulk
    streams, trailing updates alone on the update stream -- measured no faster than round 3's, see potrf.hip) against the product's
    schedule.  Every tile sees the same arithmetic in the same order: the factors, alpha and the posterior of a block of queries
    must agree BIT FOR BIT -- any read-modify-write race between the streams would show here.  Also with a far chunk's next-column
    region in its own launch (16) and the finished diagonal blocks inverted on a side stream under the last block columns (32).
    The variant with the early part of the next group's first diagonal-block update on its own stream (4; it costs a CG iteration
    at N = 32768) differs by float32 rounding in those diagonal blocks only; alpha, the float64 CG answer, agrees to 1e-9."""
    from nngp_src_amd import _lib
    x, y = synth.synthetic_queries(n, 48, seed=41)
    xt, _ = synth.synthetic_queries(300, 48, seed=42)
    lib = _lib.load(knobs=True)
    res = {}
    for key8 in (0, 1, 1, 17, 33, 5):
        lib.nngp_debug_set(8, key8)  # read when the model (its streams) is created
        try:
            mdl = GPModel(n, 48, [1.0, 1.0, 1.0], [0.0, 0.0, 0.0], diag_reg=1e-3, knobs=True).fit(x, y)
            a32, _ = mdl.factor_buffers()
            mean, var = mdl.predict(xt, cov="diag")
            res.setdefault(key8, []).append((mdl.info(), mdl.alpha().cpu().numpy().copy(), torch.tril(a32[:n, :n]).clone(), mean, var))
            mdl.close()
        finally:
            lib.nngp_debug_set(8, 0)
    (i3, a3, l3, m3, v3) = res[0][0]
    for key8 in (1, 17, 33):
        for (i4, a4, l4, m4, v4) in res[key8]:
            assert i4["clamped_pivots"] == 0
            assert torch.equal(l3, l4) and np.array_equal(a3, a4), key8
            assert np.array_equal(m3, m4) and np.array_equal(v3, v4), key8  # the inverted blocks too
    (i0, a0, l0, _, _) = res[5][0]
    assert i0["clamped_pivots"] == 0 and i0["rel_residual"] < 1e-9 and abs(i0["refine_iters"] - i3["refine_iters"]) <= 1
    assert np.linalg.norm(a0 - a3) <= 1e-9 * np.linalg.norm(a3)
    assert (l0 - l3).abs().max().item() <= 2e-3 * l3.abs().max().item()


@pytest.mark.parametrize("rows", [128, 512, 24576 + 128])  # 64x128 and 128x128 workgroup tiles
def test_gemm_in_place_inverse_block(rows):
    torch.manual_seed(2)
    bmat = torch.randn((rows, 128), device=G.dev()); inv = torch.randn((128, 128), device=G.dev())
    ref = bmat.double() @ inv.double().T
    G.gemm_nt(bmat, bmat, inv, 1.0, 0.0)
    assert (bmat.double() - ref).abs().max().item() < 1e-4


def test_gemm_f64_mfma():
    """float64 MFMA GEMM: exact on small integers (asymmetric operands catch C/D lane-map mistakes), 1e-13 on random."""
    torch.manual_seed(7)
    m, n, k = 256, 384, 144
    a = torch.randint(-8, 9, (m, k), device=G.dev()).double(); b = torch.randint(-8, 9, (n, k), device=G.dev()).double()
    cin = torch.randint(-8, 9, (m, n), device=G.dev()).double()
    c = torch.full((m, n), float("nan"), dtype=torch.float64, device=G.dev())
    G.gemm_nt_f64(c, cin, a, b, -1.0, 1.0)
    assert torch.equal(c, cin - a @ b.T)
    eye = torch.zeros((128, 128), dtype=torch.float64, device=G.dev()); eye.fill_diagonal_(1.0)
    bb = torch.randn((256, 128), dtype=torch.float64, device=G.dev())
    c = torch.empty((128, 256), dtype=torch.float64, device=G.dev())
    G.gemm_nt_f64(c, None, eye, bb, 1.0, 0.0)
    assert torch.equal(c, bb.T.contiguous())
    a = torch.randn((384, 1024), dtype=torch.float64, device=G.dev()); b = torch.randn((256, 1024), dtype=torch.float64, device=G.dev())
    c = torch.randn((384, 256), dtype=torch.float64, device=G.dev()); ref = 0.5 * c + 2.0 * a @ b.T
    G.gemm_nt_f64(c, c, a, b, 2.0, 0.5)
    assert (c - ref).abs().max().item() < 1e-11
    # few tiles and a long K: the split-K path (partial tiles + fixed-order reduction) -- exact on integers, in place,
    # and the same bits on every run
    m, n, k = 128, 384, 4096 + 48
    a = torch.randint(-8, 9, (m, k), device=G.dev()).double(); b = torch.randint(-8, 9, (n, k), device=G.dev()).double()
    cin = torch.randint(-8, 9, (m, n), device=G.dev()).double()
    c = cin.clone()
    G.gemm_nt_f64(c, c, a, b, -1.0, 3.0)
    assert torch.equal(c, 3.0 * cin - a @ b.T)
    a = torch.randn((256, 8192), dtype=torch.float64, device=G.dev()); b = torch.randn((640, 8192), dtype=torch.float64, device=G.dev())
    c1 = torch.empty((256, 640), dtype=torch.float64, device=G.dev()); c2 = torch.empty_like(c1)
    G.gemm_nt_f64(c1, None, a, b, 1.0, 0.0)
    G.gemm_nt_f64(c2, None, a, b, 1.0, 0.0)
    assert torch.equal(c1, c2) and (c1 - a @ b.T).abs().max().item() < 1e-10
    # large products (enough tiles that no split K is taken; an odd multiple of 128 rows): exact on integers (asymmetric operands,
    # A = I catches lane-map and tile-index mistakes), in place with beta, and the same bits on every run
    m, n, k = 640, 33 * 128, 208
    a = torch.randint(-8, 9, (m, k), device=G.dev()).double(); b = torch.randint(-8, 9, (n, k), device=G.dev()).double()
    cin = torch.randint(-8, 9, (m, n), device=G.dev()).double()
    c = cin.clone()
    G.gemm_nt_f64(c, c, a, b, -1.0, 3.0)
    assert torch.equal(c, 3.0 * cin - a @ b.T)
    eye = torch.zeros((640, 640), dtype=torch.float64, device=G.dev()); eye.fill_diagonal_(1.0)
    bb = torch.randn((33 * 128, 640), dtype=torch.float64, device=G.dev())
    c = torch.full((640, 33 * 128), float("nan"), dtype=torch.float64, device=G.dev())
    G.gemm_nt_f64(c, None, eye, bb, 1.0, 0.0)
    assert torch.equal(c, bb.T.contiguous())
    a = torch.randn((1024, 4096 + 16), dtype=torch.float64, device=G.dev()); b = torch.randn((4224, 4096 + 16), dtype=torch.float64, device=G.dev())
    outs = []
    for _ in range(3):
        c = torch.empty((1024, 4224), dtype=torch.float64, device=G.dev())
        G.gemm_nt_f64(c, None, a, b, 1.0, 0.0)
        outs.append(c)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    assert (outs[0] - a @ b.T).abs().max().item() < 1e-10


# ---------------------------------------------------------------------------- Cholesky / TRSM
def _spd(n, seed, cond_reg=1e-3):
    x, _ = synth.synthetic_queries(n, 32, seed=seed)
    K = o.kernel_fn(x / 1000.0, None, "nngp", o.make_arch(1))
    return K + cond_reg * np.trace(K) / n * np.eye(n)


@pytest.mark.parametrize("n", [128, 256, 640, 2048, 5120])
def test_potrf_against_lapack(n):
    A = _spd(n, n)
    a = torch.from_numpy(np.tril(A).astype(np.float32)).to(G.dev())
    a += torch.triu(torch.full((n, n), 7.0, device=G.dev()), 1)  # garbage above the diagonal must be ignored
    dinv, clamped = G.potrf(a)
    assert clamped == 0
    L = torch.tril(a).double().cpu().numpy()
    Lref = scipy.linalg.cholesky(A.astype(np.float32).astype(np.float64), lower=True)
    # backward error of the factorisation at float32 level
    assert np.linalg.norm(L @ L.T - A) / np.linalg.norm(A) < 5e-6
    assert G.rel_l2(L, Lref) < 2e-3  # forward error ~ cond * eps32
    for blk in range(n // 128):
        Lb = L[blk * 128:(blk + 1) * 128, blk * 128:(blk + 1) * 128]
        Xb = dinv[blk].double().cpu().numpy()
        assert np.allclose(np.triu(Xb, 1), 0.0)
        assert np.abs(Xb @ Lb - np.eye(128)).max() < 2e-3


@pytest.mark.parametrize("pos", [5, 31, 32, 70, 100, 127])
def test_potrf_flags_indefinite(pos):
    """A non-positive pivot is clamped and counted, whichever wave of the leaf leads its 32-column sub-block (the follower
    waves recompute the clamped pivot from the published column and must arrive at the same factor); the rest stays finite."""
    a = torch.eye(128, device=G.dev()) * 2.0
    a[pos, pos] = -1.0
    dinv, clamped = G.potrf(a)
    assert clamped >= 1
    assert torch.isfinite(torch.tril(a)).all() and torch.isfinite(dinv).all()
    keep = [i for i in range(128) if i != pos]
    L = torch.tril(a).double().cpu().numpy()
    assert np.allclose(np.diag(L)[keep], np.sqrt(2.0), rtol=1e-6)


def test_trsm_right_lower_transposed():
    n, m = 640, 256
    A = _spd(n, 11)
    a = torch.from_numpy(np.tril(A).astype(np.float32)).to(G.dev())
    dinv, _ = G.potrf(a)
    torch.manual_seed(3)
    b = torch.randn((m, n), device=G.dev())
    x = b.clone()
    G.trsm(x, a, dinv)
    L = torch.tril(a).double()
    resid = (x.double() @ L.T - b.double()).abs().max().item()
    assert resid < 5e-4, resid


@pytest.mark.parametrize("wide", [False, True])
@pytest.mark.parametrize("n,rows", [(9300, 1024), (10800, 600), (12288, 256), (8192, 1000)])
def test_blocked_solves_of_the_posterior_against_scipy(n, rows, wide):
    """The posterior's blocked triangular solves (reference: the cho_solve inside predict_fn, train.py:157-158) through
    nngp_model_apply_factor against scipy.linalg.solve_triangular on the model's own float32 factor, at sizes that are not multiples
    of the step (9300 -> 9344 = 9 x 1024 + 128 = 4 x 2048 + 1152; 10800 -> 10880 = 5 x 2048 + 640), row counts that are not
    multiples of the tile, and blocks below the float16 path's thresholds.  `wide`: round 4's 2048-column steps (two 1024-column
    panels per split-float16 update; debug key 9 = 2, knobs build -- measured and not adopted, solve.hip).  Gate: the residual
    X L^T - B  against the scale |X| |L|^T of its terms (what a backward-stable float32 substitution leaves is ~ n eps of that;
    the blocked form multiplies by inverted diagonal blocks, so a few times more), and the solution against the float64 solve
    within that residual's reach."""
    import scipy.linalg as sla
    from nngp_src_amd import _lib
    x, y = synth.synthetic_queries(n, 24, seed=51)
    if wide:
        _lib.load(knobs=True).nngp_debug_set(9, 2)  # read when the model's workspaces are sized
    try:
        model = GPModel(n, 24, [1.0, 1.0], [0.0, 0.0], diag_reg=1e-3, knobs=wide).fit(x, y)
    finally:
        if wide:
            _lib.load(knobs=True).nngp_debug_set(9, 0)
    if wide:
        _lib.load(knobs=True).nngp_debug_set(9, 2)
    a32, _ = model.factor_buffers()
    L = torch.tril(a32[:n, :n]).double().cpu().numpy()
    rng = np.random.default_rng(n + rows)
    B = rng.standard_normal((rows, n)).astype(np.float32) * (1.0 + 10.0 * rng.random((rows, 1)).astype(np.float32))
    # forward half
    X = model.apply_factor(torch.from_numpy(B.copy()).to(G.dev())).cpu().numpy().astype(np.float64)
    scale = np.abs(X) @ np.abs(L).T
    res = np.abs(X @ L.T - B)
    assert (res / scale).max() < 2e-5, (res / scale).max()
    Xref = sla.solve_triangular(L, B.astype(np.float64).T, lower=True).T
    assert np.linalg.norm(X - Xref) <= 2e-3 * np.linalg.norm(Xref), np.linalg.norm(X - Xref) / np.linalg.norm(Xref)
    # both halves: Z L L^T = B
    Z = model.apply_factor(torch.from_numpy(B.copy()).to(G.dev()), both_halves=True).cpu().numpy().astype(np.float64)
    Y = Z @ L                       # should solve Y L^T = B like X
    scale2 = np.abs(Y) @ np.abs(L).T + (np.abs(Z) @ np.abs(L)) @ np.abs(L).T
    res2 = np.abs(Y @ L.T - B)
    assert (res2 / scale2).max() < 2e-5, (res2 / scale2).max()
    model.close()
    if wide:
        _lib.load(knobs=True).nngp_debug_set(9, 0)


@pytest.mark.parametrize("n", [9300, 12288])
def test_transposed_split_copy_from_the_panel_solves_is_the_separate_pass_bit_for_bit(n):
    """Round 4: the Cholesky's fused panel solves also write the TRANSPOSED float16-split copy of L (operand of the posterior's
    "B L^-1" solves) instead of a separate pass over the finished factor beside a predict's first solve (k_split_lower_t; debug key
    9 = 8 restores it, knobs build).  Same values, same rounding: the posterior of a block large enough for the float16-pipe
    solves must come out bit for bit the same, and so must both halves of the solve through nngp_model_apply_factor."""
    from nngp_src_amd import _lib
    x, y = synth.synthetic_queries(n, 24, seed=71)
    xt, _ = synth.synthetic_queries(1024, 24, seed=72)
    lib = _lib.load(knobs=True)
    B = torch.randn((1024, n), device=G.dev(), dtype=torch.float32)
    out = {}
    for key9 in (8, 0):
        lib.nngp_debug_set(9, key9)  # read by the factorisation
        try:
            model = GPModel(n, 24, [1.0, 1.0], [0.0, 0.0], diag_reg=1e-3, knobs=True).fit(x, y)
            mean, var = model.predict(xt, cov="diag")
            z = model.apply_factor(B.clone(), both_halves=True).clone()
            out[key9] = (mean, var, z)
            model.close()
        finally:
            lib.nngp_debug_set(9, 0)
    assert np.array_equal(out[0][0], out[8][0]) and np.array_equal(out[0][1], out[8][1]) and torch.equal(out[0][2], out[8][2])


def test_triangular_inverse_blocks_skip_their_zero_tiles_bit_for_bit():
    """Round 4: the blocked solves multiply by the inverted diagonal blocks with a float32 GEMM that walks, per column tile, only the
    k tiles where the (triangular) block is non-zero.  The skipped terms are exact zeros: the solve must return the bits of the
    full product (debug key 9 = 4, knobs build)."""
    from nngp_src_amd import _lib
    n, rows = 5000, 384
    x, y = synth.synthetic_queries(n, 24, seed=61)
    lib = _lib.load(knobs=True)
    model = GPModel(n, 24, [1.0, 1.0], [0.0, 0.0], diag_reg=1e-3, knobs=True).fit(x, y)
    B = torch.randn((rows, n), device=G.dev(), dtype=torch.float32)
    out = {}
    for key9 in (0, 4):
        lib.nngp_debug_set(9, key9)
        try:
            out[key9] = (model.apply_factor(B.clone()).clone(), model.apply_factor(B.clone(), both_halves=True).clone())
        finally:
            lib.nngp_debug_set(9, 0)
    assert torch.equal(out[0][0], out[4][0]) and torch.equal(out[0][1], out[4][1])
    assert torch.isfinite(out[0][1]).all()
    model.close()


# ---------------------------------------------------------------------------- fit / predict (a3, a4)
def _fit_and_check(x, y, xt, n_relu=1, get="nngp", w=1.0, b=0.0):
    a = o.make_arch(n_relu, w, b)
    model = GPModel(x.shape[0], x.shape[1], a.w_std, a.b_std, get=get, diag_reg=1e-3, ny=y.shape[1])
    model.fit(x, y)
    info = model.info()
    post = o.Posterior(x, y, a, diag_reg=1e-3)
    K = post._factor(get)[0]
    np.testing.assert_allclose(info["reg"], 1e-3 * np.trace(K) / x.shape[0], rtol=1e-8)  # oracle diag carries sqrt(rounding) noise
    assert info["clamped_pivots"] == 0 and info["rel_residual"] <= 1e-10, info
    alpha = model.alpha().cpu().numpy()
    assert G.rel_l2(alpha, post._factor(get)[2]) < 1e-6
    return model, post, info


@pytest.mark.parametrize("n,m,d,n_relu", [(100, 7, 20, 1), (128, 128, 64, 1), (700, 130, 64, 3), (1500, 64, 128, 1)])
def test_fit_predict_nngp(n, m, d, n_relu):
    x, y = synth.synthetic_queries(n, d, seed=0)
    xt, _ = synth.synthetic_queries(m, d, seed=1)
    model, post, info = _fit_and_check(x, y, xt, n_relu)
    mean_ref, cov_ref = post.predict(xt, "nngp", True)
    mean, var = model.predict(xt, cov="diag")
    l2, elem = G.mean_gate(mean, mean_ref)
    assert l2 < 1e-6 and elem < 1e-6, (l2, elem, info)
    np.testing.assert_allclose(var, np.diag(cov_ref), rtol=1e-5, atol=1e-9 * np.abs(cov_ref).max())
    mean2, cov = model.predict(xt, cov="full")
    np.testing.assert_allclose(mean2, mean, rtol=1e-6, atol=1e-6 * np.abs(mean).max())  # corrected through different rows Z
    assert np.abs(cov - cov_ref).max() < 1e-6 * np.abs(np.diag(cov_ref)).max()
    assert np.array_equal(cov, cov.T)
    np.testing.assert_allclose(np.diag(cov), var, rtol=1e-6)  # full covariance: level 2; diag: level 1 (the default)
    model.set_refine(3)  # two sweeps + second-order formula
    var3 = model.predict(xt, cov="diag")[1]
    _, cov3 = model.predict(xt, cov="full")
    np.testing.assert_allclose(var3, np.diag(cov_ref), rtol=1e-7, atol=1e-10 * np.abs(cov_ref).max())
    assert np.abs(cov3 - cov_ref).max() < 1e-8 * np.abs(np.diag(cov_ref)).max()
    # float32-only covariance (set_refine(0)): accurate to ~cond*eps32 of the PRIOR variance only
    model.set_refine(0)
    var32 = model.predict(xt, cov="diag")[1]
    _, cov32 = model.predict(xt, cov="full")
    prior = o.diag_kernel(np.sum(xt * xt, axis=1) / d, o.make_arch(n_relu))[0]
    assert np.abs(var32 - np.diag(cov_ref)).max() < 1e-4 * prior.max()
    assert np.abs(cov32 - cov_ref).max() < 1e-4 * prior.max()
    model.set_refine(2)
    np.testing.assert_allclose(model.predict(xt, cov=False), mean, rtol=1e-6, atol=1e-6 * np.abs(mean).max())  # full CG vs early stop + correction
    # x_test=None: predictions on the training rows (estimator.py:37-40)
    mean_tr, var_tr = model.predict(None, cov="diag")
    mtr_ref, ctr_ref = post.predict(None, "nngp", True)
    assert G.mean_gate(mean_tr, mtr_ref)[0] < 1e-6
    np.testing.assert_allclose(var_tr, np.diag(ctr_ref), rtol=1e-4, atol=1e-9 * np.abs(ctr_ref).max())


def test_fit_predict_ntk_mean_and_two_outputs():
    x, y = synth.synthetic_queries(300, 64, seed=0)
    xt, _ = synth.synthetic_queries(50, 64, seed=1)
    y2 = np.concatenate([y, np.sin(y)], axis=1)
    model, post, _ = _fit_and_check(x, y2, xt, n_relu=2, get="ntk", w=1.2, b=0.1)
    mean = model.predict(xt, cov=False)
    assert mean.shape == (50, 2)
    m_ref, c_ref = post.predict(xt, "ntk", True)
    assert G.mean_gate(mean, m_ref)[0] < 1e-6
    # NTK ensemble covariance: K_tt + Z K_dd Z^T - (K_td Z^T + h.c.)
    _, var = model.predict(xt, cov="diag")
    np.testing.assert_allclose(var, np.diag(c_ref), rtol=1e-4, atol=1e-8 * np.abs(c_ref).max())
    _, cov = model.predict(xt, cov="full")
    assert np.abs(cov - c_ref).max() < 1e-5 * np.abs(np.diag(c_ref)).max()
    mtr, vtr = model.predict(None, cov="diag")
    mtr_ref, ctr_ref = post.predict(None, "ntk", True)
    assert G.mean_gate(mtr, mtr_ref)[0] < 1e-6
    np.testing.assert_allclose(vtr, np.diag(ctr_ref), rtol=1e-3, atol=1e-7 * np.abs(ctr_ref).max())


def test_forest_golden_config1(golden_dir):
    """BASELINE.json configs[0]: forest queries, N=1000 train / 200 test, 3-layer ReLU NNGP."""
    g = np.load(os.path.join(golden_dir, "forest_n1000_m200.npz"))
    a = o.make_arch(1)
    for get in ("nngp", "ntk"):
        model = GPModel(1000, 20, a.w_std, a.b_std, get=get, diag_reg=1e-3).fit(g["X_train"], g["Y_train"])
        info = model.info()
        assert info["clamped_pivots"] == 0 and info["rel_residual"] < 1e-10, info
        if get == "nngp":
            mean, var = model.predict(g["X_test"], cov="diag")
            np.testing.assert_allclose(var, g["nngp_var"], rtol=1e-5)
            _, cov = model.predict(g["X_test"][:16], cov="full")
            assert np.abs(cov - g["nngp_cov16"]).max() < 1e-6 * np.abs(np.diag(g["nngp_cov16"])).max()
        else:
            mean, var = model.predict(g["X_test"], cov="diag")
            np.testing.assert_allclose(var, g["ntk_var"], rtol=1e-4)
        l2, elem = G.mean_gate(mean, g[get + "_mean"])
        assert l2 < 1e-6 and elem < 1e-5, (get, l2, elem, info)
        from nngp_src_amd.util import q_error_profile
        pa, pb = q_error_profile((mean - g["Y_test"]).ravel()), q_error_profile((g[get + "_mean"] - g["Y_test"]).ravel())
        for key in pa:
            assert abs(pa[key] - pb[key]) <= 1e-3 * abs(pb[key]), (key, pa[key], pb[key])


def test_medium_size_against_c_oracle():
    """N=4096 (32 Cholesky leaves, deep recursion): posterior means vs the C float64 oracle."""
    n, m, d = 4096, 256, 64
    x, y = synth.synthetic_queries(n, d, seed=0)
    xt, _ = synth.synthetic_queries(m, d, seed=1)
    a = o.make_arch(1)
    ref = c_oracle.fit(x, y, a.w_std, a.b_std)
    mean_ref, var_ref = c_oracle.predict_nngp(ref, xt, 1)
    model = GPModel(n, d, a.w_std, a.b_std, diag_reg=1e-3).fit(x, y)
    info = model.info()
    mean, var = model.predict(xt, cov="diag")
    l2, elem = G.mean_gate(mean, mean_ref)
    assert info["clamped_pivots"] == 0 and l2 < 1e-6 and elem < 1e-6, (l2, elem, info)
    np.testing.assert_allclose(var, var_ref, rtol=1e-4, atol=1e-9 * np.abs(var_ref).max())
    assert info["refine_iters"] <= 20, info


def test_large_size_properties():
    """BASELINE configs[1] size (N=8192, d=64): size-independent properties instead of an O(N^3) CPU oracle:
    float64 residual of the solve, symmetry of K, K alpha + reg alpha = y on a row sample, and linearity in y."""
    n, d = 8192, 64
    x, y = synth.synthetic_queries(n, d, seed=0)
    a = o.make_arch(1)
    model = GPModel(n, d, a.w_std, a.b_std, diag_reg=1e-3).fit(x, y)
    info = model.info()
    assert info["clamped_pivots"] == 0 and info["rel_residual"] < 1e-10, info
    alpha = model.alpha().cpu().numpy()
    rows = np.arange(0, n, 257)
    Krows = o.kernel_fn(x[rows], x, "nngp", a)
    lhs = Krows @ alpha + info["reg"] * alpha[rows]
    assert np.abs(lhs - y[rows]).max() < 1e-7 * np.abs(y).max()
    kbuf, ld = model.kernel_buffer()
    blk = kbuf[:512, :512].cpu().numpy()
    assert np.array_equal(blk, blk.T)
    np.testing.assert_allclose(blk[:8], o.kernel_fn(x[:8], x[:512], "nngp", a), rtol=1e-11)
    mean_tr = model.predict(None, cov=False)
    np.testing.assert_allclose(mean_tr, y - info["reg"] * alpha, atol=1e-7 * np.abs(y).max())  # K alpha = y - reg alpha
    model2 = GPModel(n, d, a.w_std, a.b_std, diag_reg=1e-3).fit(x, 3.0 * y)
    assert G.rel_l2(model2.alpha().cpu().numpy(), 3.0 * alpha) < 1e-8


def test_lookahead_cholesky_matches_recursion():
    """N=8192 takes the two-stream look-ahead driver (block columns of 1024); it must agree with the recursion bit for bit
    on the factor (same kernels, same per-tile arithmetic order) and with the oracle rows on alpha."""
    from nngp_src_amd import _lib
    n, d = 8192, 32
    x, y = synth.synthetic_queries(n, d, seed=5)
    a = o.make_arch(1)
    lib = _lib.load(knobs=True)  # the A/B switch only exists in libnngp_hip_knobs.so
    alphas = []
    for disable in (0, 1):
        lib.nngp_debug_set(2, disable)
        model = GPModel(n, d, a.w_std, a.b_std, diag_reg=1e-3, knobs=True).fit(x, y)
        info = model.info()
        assert info["clamped_pivots"] == 0 and info["rel_residual"] < 1e-10, info
        alphas.append(model.alpha().cpu().numpy())
        model.close()
    lib.nngp_debug_set(2, 0)
    assert G.rel_l2(alphas[0], alphas[1]) < 1e-9
    rows = np.arange(0, n, 511)
    Krows = o.kernel_fn(x[rows], x, "nngp", a)
    reg = 1e-3 * np.mean(np.sum(x * x, axis=1) / d / 2)
    assert np.abs(Krows @ alphas[0] + reg * alphas[0][rows] - y[rows]).max() < 1e-7 * np.abs(y).max()


@pytest.mark.parametrize("n,nb", [(3000, 512), (9300, 1024)])
def test_right_looking_block_columns_single_rank(n, nb):
    """The block-column pieces (factor_panel / factor_update_cols) that the multi-GPU Cholesky deals out, run on one rank.  The larger
    case takes the fused float16-pipe panel solve and the multi-region split-float16 updates (four target block columns per launch)."""
    from nngp_src_amd import distributed
    d = 32
    x, y = synth.synthetic_queries(n, d, seed=7)
    a = o.make_arch(1)
    ref = GPModel(n, d, a.w_std, a.b_std, diag_reg=1e-3).fit(x, y)
    model = GPModel(n, d, a.w_std, a.b_std, diag_reg=1e-3)
    model.set_train(x, y)
    model.build_rows(0, n)
    distributed.distributed_factor(model, nb=nb)
    model.solve()
    info = model.info()
    assert info["clamped_pivots"] == 0 and info["rel_residual"] < 1e-10, info
    assert G.rel_l2(model.alpha().cpu().numpy(), ref.alpha().cpu().numpy()) < 1e-9
    xt, _ = synth.synthetic_queries(40, d, seed=8)
    m1, v1 = model.predict(xt, cov="diag")
    m0, v0 = ref.predict(xt, cov="diag")
    assert G.mean_gate(m1, m0)[0] < 1e-9
    np.testing.assert_allclose(v1, v0, rtol=1e-6)


def test_full_forest_run_against_c_oracle(golden_dir):
    """The reference's own forest run at full size (train.py defaults: 18 000 queries, seed-10 split -> N=10 800 train,
    M=3 600 test, d=20): posterior means, variances and the q-error profile against the float64 C oracle."""
    from nngp_src_amd import encoder as enc, util
    import contextlib, io
    g = np.load(os.path.join(golden_dir, "forest_queries.npz"))
    bounds, cards = g["bounds"], g["cards"]
    sent = np.iinfo(np.int32).min
    loader = enc.GeneralQueryEncoder()
    lo, hi = g["col_lo"], g["col_hi"]
    X = np.tile(np.array([0.0, 1000.0]), (bounds.shape[0], 10))
    active = bounds[:, :, 0] != sent
    scaled = (bounds.astype(np.float64) - lo[None, :, None]) / (hi - lo)[None, :, None] * 1000.0
    X[:, 0::2] = np.where(active, scaled[:, :, 0], 0.0)
    X[:, 1::2] = np.where(active, scaled[:, :, 1], 1000.0)
    q0 = [(c, float(bounds[0, c, 0]), float(bounds[0, c, 1])) for c in range(10) if active[0, c]]
    np.testing.assert_array_equal(X[0], loader.transform_to_1d_array(q0))  # vectorised encoding == the encoder
    Y = np.log2(cards.astype(np.float64))[:, None]
    with contextlib.redirect_stdout(io.StringIO()):
        Xtr, Ytr, _, Xte, Yte, _, _, _, _ = util.train_test_val_split(X, Y, 0.6, 0.2)
    assert Xtr.shape == (10800, 20) and Xte.shape == (3600, 20)
    a = o.make_arch(1)
    ref = c_oracle.fit(Xtr, Ytr, a.w_std, a.b_std)
    mean_ref, var_ref = c_oracle.predict_nngp(ref, Xte, 1)
    model = GPModel(10800, 20, a.w_std, a.b_std, diag_reg=1e-3).fit(Xtr, Ytr)
    info = model.info()
    mean, var = model.predict(Xte, cov="diag")
    l2, elem = G.mean_gate(mean, mean_ref)
    assert info["clamped_pivots"] == 0 and info["rel_residual"] < 1e-10, info
    assert l2 < 1e-6 and elem < 1e-5, (l2, elem, info)
    np.testing.assert_allclose(var, var_ref, rtol=1e-4)
    pa, pb = util.q_error_profile((mean - Yte).ravel()), util.q_error_profile((mean_ref - Yte).ravel())
    for key in pa:
        assert abs(pa[key] - pb[key]) <= 1e-4 * abs(pb[key]), (key, pa[key], pb[key])
    print("forest full run: cg_iters=%d, q-error median %.4f mean %.3f" % (info["refine_iters"], pa["median"], pa["mean"]))


@pytest.mark.parametrize("n,m,d,n_relu,ny", [(4100, 130, 7, 1, 1), (4224, 1, 20, 2, 2), (5003, 1000, 12, 1, 1), (6143, 257, 20, 3, 1),
                                             (6143, 1500, 20, 1, 1)])  # the last one is large enough for the float16-pipe solves
def test_lookahead_sizes_not_multiples_of_the_blocks(n, m, d, n_relu, ny):
    """N just above the look-ahead threshold and not a multiple of 128 / 1024, M not a multiple of 128: the tail block
    column, the split copies of the factor and the float16-pipe solves of the posterior against the float64 C oracle."""
    x, y0 = synth.synthetic_queries(n, d, seed=n)
    xt, _ = synth.synthetic_queries(m, d, seed=n + 1)
    y = np.concatenate([y0 * (c + 1) + c for c in range(ny)], axis=1)
    a = o.make_arch(n_relu)
    model = GPModel(n, d, a.w_std, a.b_std, diag_reg=1e-3, ny=ny).fit(x, y)
    info = model.info()
    assert info["clamped_pivots"] == 0 and info["rel_residual"] < 1e-9 and info["refine_iters"] <= 10, info
    mean, var = model.predict(xt, cov="diag")
    for c in range(ny):
        ref = c_oracle.fit(x, y[:, c:c + 1], a.w_std, a.b_std)
        mean_ref, var_ref = c_oracle.predict_nngp(ref, xt, 1)
        l2, elem = G.mean_gate(mean[:, c:c + 1], mean_ref)
        assert l2 < 1e-6 and elem < 1e-5, (c, l2, elem, info)
        if c == 0:
            np.testing.assert_allclose(var, var_ref, rtol=1e-5, atol=1e-9 * np.abs(var_ref).max())
    # the float32-only covariance goes through the float16-pipe forward solve too
    model.set_refine(0)
    _, var0 = model.predict(xt, cov="diag")
    assert np.max(np.abs(var0 - var_ref)) <= 2e-2 * np.max(np.abs(var_ref))
    model.close()


def test_ntk_on_the_lookahead_path_and_append():
    """NTK kernel at a size that takes the look-ahead factorisation and the float16-pipe solves: mean and ensemble
    covariance against the NumPy oracle, then 300 appended rows against a refit."""
    n, b, d = 4300, 300, 16
    x, y = synth.synthetic_queries(n + b, d, seed=11)
    xt, _ = synth.synthetic_queries(96, d, seed=12)
    a = o.make_arch(2, 1.1, 0.05)
    model = GPModel(n + b, d, a.w_std, a.b_std, get="ntk", diag_reg=1e-3).fit(x[:n], y[:n])
    post = o.Posterior(x[:n], y[:n], a, diag_reg=1e-3)
    mean, var = model.predict(xt, cov="diag")
    m_ref, c_ref = post.predict(xt, "ntk", True)
    assert G.mean_gate(mean, m_ref)[0] < 1e-6
    np.testing.assert_allclose(var, np.diag(c_ref), rtol=1e-4, atol=1e-8 * np.abs(c_ref).max())
    model.append(x[n:], y[n:])
    ref = GPModel(n + b, d, a.w_std, a.b_std, get="ntk", diag_reg=1e-3).fit(x, y)
    a1, a2 = model.alpha().cpu().numpy(), ref.alpha().cpu().numpy()
    assert np.linalg.norm(a1 - a2) <= 1e-7 * np.linalg.norm(a2)
    m1, v1 = model.predict(xt, cov="diag")
    m2, v2 = ref.predict(xt, cov="diag")
    assert np.allclose(m1, m2, rtol=1e-8, atol=1e-8 * np.abs(m2).max())
    np.testing.assert_allclose(v1, v2, rtol=1e-5, atol=1e-8 * np.abs(v2).max())
    # the appended fit directly against the float64 oracle on the concatenated set
    m_or, c_or = o.Posterior(x, y, a, diag_reg=1e-3).predict(xt, "ntk", True)
    assert G.mean_gate(m1, m_or)[0] < 1e-6
    np.testing.assert_allclose(v1, np.diag(c_or), rtol=1e-4, atol=1e-8 * np.abs(c_or).max())
    model.close(); ref.close()


def test_edge_cases_small_and_degenerate():
    """n = 1, duplicated training rows (K singular without the regulariser), zero test rows, many output columns."""
    a = o.make_arch(1)
    # a single training query
    x1 = np.array([[3.0, 4.0, 0.0, 1000.0]]); y1 = np.array([[5.0]])
    m1 = GPModel(1, 4, a.w_std, a.b_std, diag_reg=1e-3).fit(x1, y1)
    xt = np.array([[3.0, 4.0, 0.0, 1000.0], [1.0, 0.0, 0.0, 0.0]])
    mean, var = m1.predict(xt, cov="diag")
    mref, cref = o.Posterior(x1, y1, a, diag_reg=1e-3).predict(xt, "nngp", True)
    np.testing.assert_allclose(mean, mref, rtol=1e-9)
    np.testing.assert_allclose(var, np.diag(cref), rtol=1e-6)
    # duplicated rows: K is singular, K + reg I is not
    x, y = synth.synthetic_queries(150, 20, seed=3)
    xd = np.vstack([x, x[:50]]); yd = np.vstack([y, y[:50] + 0.5])
    md = GPModel(200, 20, a.w_std, a.b_std, diag_reg=1e-3).fit(xd, yd)
    info = md.info()
    assert info["clamped_pivots"] == 0 and info["rel_residual"] < 1e-9, info
    xt2, _ = synth.synthetic_queries(30, 20, seed=4)
    mean, var = md.predict(xt2, cov="diag")
    mref, cref = o.Posterior(xd, yd, a, diag_reg=1e-3).predict(xt2, "nngp", True)
    assert G.mean_gate(mean, mref)[0] < 1e-6
    np.testing.assert_allclose(var, np.diag(cref), rtol=1e-4)
    # no test rows
    mean0, var0 = md.predict(np.zeros((0, 20)), cov="diag")
    assert mean0.shape == (0, 1) and var0.shape == (0,)
    # three output columns share one factorisation
    y3 = np.concatenate([yd, yd ** 2 / 10.0, np.cos(yd)], axis=1)
    m3 = GPModel(200, 20, a.w_std, a.b_std, diag_reg=1e-3, ny=3).fit(xd, y3)
    mean3, var3 = m3.predict(xt2, cov="diag")
    mref3, _ = o.Posterior(xd, y3, a, diag_reg=1e-3).predict(xt2, "nngp", True)
    assert mean3.shape == (30, 3) and G.mean_gate(mean3, mref3)[0] < 1e-6
    np.testing.assert_allclose(var3, var, rtol=1e-9)
    # all-zero queries (|x| = 0: the arc-cosine map's 0/0 corner), orthogonal and opposite queries (theta = pi/2, pi),
    # in the training set and in the test set; with and without biases
    for b_std in (0.0, 0.2):
        az = o.make_arch(2, 1.1, b_std)
        xz = np.vstack([x[:100], np.zeros((2, 20)), -x[:3], np.eye(20)[:4] * 7.0])
        yz = np.vstack([y[:100], [[0.0], [0.1]], y[:3] + 1.0, [[1.0], [2.0], [3.0], [4.0]]])
        mz = GPModel(len(xz), 20, az.w_std, az.b_std, diag_reg=1e-3).fit(xz, yz)
        xtz = np.vstack([np.zeros((1, 20)), -x[5:7], np.eye(20)[6:8] * 3.0, xt2[:5]])
        mean, var = mz.predict(xtz, cov="diag")
        mref, cref = o.Posterior(xz, yz, az, diag_reg=1e-3).predict(xtz, "nngp", True)
        assert np.isfinite(mean).all() and np.isfinite(var).all()
        assert G.mean_gate(mean, mref)[0] < 1e-6
        np.testing.assert_allclose(var, np.diag(cref), rtol=1e-4, atol=1e-9 * np.abs(cref).max())
        mz.close()
    # wrong shapes are rejected on the host
    with pytest.raises(ValueError):
        md.predict(np.zeros((3, 19)))
    with pytest.raises(ValueError):
        GPModel(10, 4, a.w_std, a.b_std).fit(np.zeros((11, 4)), np.zeros((11, 1)))


# ---------------------------------------------------------------------------- seeded random sweep over the whole operator surface
def _sweep_case(seed):
    r = np.random.default_rng(1000 + seed)
    get = "ntk" if seed % 3 == 2 else "nngp"
    n = int(r.integers(40, 1600 if get == "ntk" else 5200))
    if os.environ.get("NNGP_SWEEP_NMAX"):  # exploration at other sizes (NNGP cases only; the NumPy NTK oracle is slow)
        get, n = "nngp", int(r.integers(int(os.environ.get("NNGP_SWEEP_NMIN", "40")), int(os.environ["NNGP_SWEEP_NMAX"])))
        if os.environ.get("NNGP_SWEEP_NTK") and seed % 2:  # NTK at these sizes: mean only (C oracle alpha)
            get = "ntk"
    return dict(seed=seed, get=get, n=n, m=int(r.integers(1, 400)), d=int(r.choice([2, 3, 7, 20, 64, 128, 200, 256])),
                n_relu=int(r.integers(1, 5)), w=float(r.uniform(0.6, 1.8)), b=float(r.choice([0.0, 0.05, 0.3])),
                diag_reg=float(r.choice([1e-4, 1e-3, 1e-2])), absolute=bool(r.integers(0, 4) == 0), join=bool(r.integers(0, 3) == 0))


@pytest.mark.parametrize("seed", range(int(os.environ.get("NNGP_SWEEP_CASES", "12"))))  # more cases: set the variable
def test_random_sweep_against_the_float64_oracle(seed):
    """Random architecture (depth, W_std, b_std), regulariser (size, relative/absolute), sizes off every block grid and
    both encodings: float64 kernel, alpha, mean and variance of the HIP path against the float64 oracle.  One line per
    case is appended to gpurun_out/parity_sweep.jsonl (summary committed as profiles/*_parity_sweep.jsonl)."""
    c = _sweep_case(seed)
    if c["absolute"]:  # diag_reg_absolute_scale: reg is not scaled by trace/N, so give it the kernel's own scale (~|x|^2/d)
        c["diag_reg"] *= 1e5
    x, y = synth.synthetic_queries(c["n"], c["d"], seed=seed, join_block=c["join"] and c["d"] >= 8)
    xt, _ = synth.synthetic_queries(c["m"], c["d"], seed=seed + 100, join_block=c["join"] and c["d"] >= 8)
    a = o.make_arch(c["n_relu"], c["w"], c["b"])
    model = GPModel(c["n"], c["d"], a.w_std, a.b_std, get=c["get"], diag_reg=c["diag_reg"],
                    diag_reg_absolute_scale=c["absolute"]).fit(x, y)
    if os.environ.get("NNGP_SWEEP_SERVING"):  # exploration: the same cases through the explicit-inverse serving mode
        model.prepare_serving()
    if os.environ.get("NNGP_SWEEP_LEVEL"):    # exploration: another covariance precision level
        model.set_refine(int(os.environ["NNGP_SWEEP_LEVEL"]))
    mean, var = model.predict(xt, cov="diag")  # before info(): the alpha CG stops early, the mean is corrected through Z
    cov_iters, sweep_est = model.cov_iters(), model.sweep_estimate()[1]
    info = model.info()
    shift = model.factor_shift() / info["reg"]
    _, cov = model.predict(xt[:64], cov="full")
    prior = o.diag_kernel(np.sum(xt * xt, axis=1) / c["d"], a)[0].max()  # var = prior - ...: resolution eps64 * prior
    assert np.abs(np.diag(cov) - var[:64]).max() <= 1e-4 * np.abs(var[:64]).max() + 1e-13 * prior  # level 2 vs level 1
    if c["get"] == "nngp":
        ref = c_oracle.fit(x, y, a.w_std, a.b_std, diag_reg=c["diag_reg"], absolute=c["absolute"])
        mean_ref, var_ref = c_oracle.predict_nngp(ref, xt, 1)
        alpha_ref, reg_ref = ref["alpha"], ref["reg"]
    elif c["n"] > 1600:  # exploration only: the NumPy oracle is too slow here; mean through the C oracle's alpha
        ref = c_oracle.fit(x, y, a.w_std, a.b_std, get="ntk", diag_reg=c["diag_reg"], absolute=c["absolute"])
        mean_ref = c_oracle.kernel_build(xt, x, "ntk", a.w_std, a.b_std) @ ref["alpha"]
        var_ref, alpha_ref, reg_ref = var, ref["alpha"], ref["reg"]
    else:
        post = o.Posterior(x, y, a, diag_reg=c["diag_reg"], diag_reg_absolute_scale=c["absolute"])
        mean_ref, cov_ref = post.predict(xt, "ntk", True)
        var_ref, alpha_ref = np.diag(cov_ref), post._factor("ntk")[2]
        reg_ref = c["diag_reg"] * (1.0 if c["absolute"] else np.trace(post._factor("ntk")[0]) / c["n"])
    l2, elem = G.mean_gate(mean, mean_ref)
    row = dict(c, reg_rel=abs(info["reg"] - reg_ref) / reg_ref, cg_iters=info["refine_iters"], cov_iters=cov_iters, sweep_est=sweep_est, factor_shift=shift,
               clamped=info["clamped_pivots"],
               alpha_rel_l2=G.rel_l2(model.alpha().cpu().numpy(), alpha_ref), mean_rel_l2=l2, mean_elem=elem,
               var_max_rel=float(np.max(np.abs(var - var_ref.ravel()) / np.maximum(np.abs(var_ref.ravel()), 1e-9 * np.abs(var_ref).max() + 1e-11 * prior))))
    try:
        os.makedirs("gpurun_out", exist_ok=True)
        with open("gpurun_out/parity_sweep.jsonl", "a") as f:
            f.write(json.dumps(row) + "\n")
    except OSError:
        pass
    model.close()
    assert row["clamped"] == 0 and row["reg_rel"] < 1e-8, row
    if c["get"] == "nngp" and row["cg_iters"] >= 8 and not os.environ.get("NNGP_SWEEP_SERVING"):
        assert cov_iters > 0 and row["var_max_rel"] < 1e-6, row  # weak preconditioner: the rows must have gone on by CG
    # small regularisers (1e-4 relative, or absolute on a large-trace kernel) raise cond(K + reg I): alpha itself is
    # then determined to ~cond * eps64 only, the mean stays at the gate
    # (gates of SURVEY.md 7: 1e-4 norm-wise and elementwise; 149 of 150 cases are within 1e-6 / 5e-6, the last -- seed 119,
    # NTK, d = 2, four layers, cond ~ 1e8 -- at 7e-7 / 1.1e-5)
    assert row["mean_rel_l2"] < 1e-6 and row["mean_elem"] < 3e-5, row
    # Variance gate of SURVEY.md 7: 1e-3 relative.  Held here at 3e-4: the float64 oracle itself is that far from an 80-bit
    # referee on the worst-conditioned cases (d = 2 .. 3, diag_reg 1e-4: rows whose variance is 1e-8 of the prior), and
    # last-bit differences between the two kernel builds move those variances by 1e-4
    # (tests/test_gpu_extended_precision.py; profiles/r2_extended_precision.jsonl).  143 of 150 cases are within 3e-5.
    assert row["var_max_rel"] < 3e-4, row


@pytest.mark.parametrize("seed", [0, 1, 2, 3, 4, 5])
def test_level2_variances_keep_the_tighter_gates(seed):
    """The default covariance level went from 2 to 1 in round 2 and the sweep's gates were widened with it (variance 1e-4 ->
    3e-4, mean 1e-5 -> 3e-5 elementwise).  Level 2 (one float64 correction sweep + the second-order formula) is still a
    supported setting: the same random cases at set_refine(2) against the float64 oracle at the gates the sweep had before."""
    c = _sweep_case(seed)
    if c["get"] != "nngp":
        pytest.skip("NNGP cases only (the NTK covariance always runs two sweeps)")
    if c["absolute"]:
        c["diag_reg"] *= 1e5
    x, y = synth.synthetic_queries(c["n"], c["d"], seed=seed, join_block=c["join"] and c["d"] >= 8)
    xt, _ = synth.synthetic_queries(c["m"], c["d"], seed=seed + 100, join_block=c["join"] and c["d"] >= 8)
    a = o.make_arch(c["n_relu"], c["w"], c["b"])
    model = GPModel(c["n"], c["d"], a.w_std, a.b_std, get="nngp", diag_reg=c["diag_reg"], diag_reg_absolute_scale=c["absolute"]).fit(x, y)
    model.set_refine(2)
    mean, var = model.predict(xt, cov="diag")
    ref = c_oracle.fit(x, y, a.w_std, a.b_std, diag_reg=c["diag_reg"], absolute=c["absolute"])
    mean_ref, var_ref = c_oracle.predict_nngp(ref, xt, 1)
    prior = o.diag_kernel(np.sum(xt * xt, axis=1) / c["d"], a)[0].max()
    l2, elem = G.mean_gate(mean, mean_ref)
    vrel = float(np.max(np.abs(var - var_ref.ravel()) / np.maximum(np.abs(var_ref.ravel()), 1e-9 * np.abs(var_ref).max() + 1e-11 * prior)))
    model.close()
    assert l2 < 1e-6 and elem < 1e-5 and vrel < 1e-4, (c, l2, elem, vrel)


def test_adaptive_covariance_when_the_alpha_solve_says_nothing():
    """The ill-conditioned fit of the sweep (seed 6: d=3, 4-layer, diag_reg=1e-4, cond 2.6e7) with y = 0: the alpha solve
    converges at once, so only the row flag can tell that the fixed sweeps were not enough.  The covariance does not
    depend on y: it must equal the one of the fit with the real targets (which continues by CG because of its 13
    alpha iterations), at the diag and the full level."""
    c = _sweep_case(6)
    x, y = synth.synthetic_queries(c["n"], c["d"], seed=6)
    xt, _ = synth.synthetic_queries(c["m"], c["d"], seed=106)
    a = o.make_arch(c["n_relu"], c["w"], c["b"])
    ref_model = GPModel(c["n"], c["d"], a.w_std, a.b_std, diag_reg=c["diag_reg"]).fit(x, y)
    _, var_ref = ref_model.predict(xt, cov="diag")
    assert ref_model.info()["refine_iters"] >= 8 and ref_model.cov_iters() > 0
    model = GPModel(c["n"], c["d"], a.w_std, a.b_std, diag_reg=c["diag_reg"], knobs=True).fit(x, np.zeros_like(y))
    assert model.info()["refine_iters"] <= 1
    mean, var = model.predict(xt, cov="diag")
    assert model.cov_iters() > 0
    assert np.all(mean == 0.0)
    np.testing.assert_allclose(var, var_ref, rtol=1e-6)
    _, cov = model.predict(xt, cov="full")
    assert model.cov_iters() > 0
    np.testing.assert_allclose(np.diag(cov), var_ref, rtol=1e-6)
    # fixed sweeps only (what levels >= 2 did before): far off
    model.debug_set(6, 1)  # (the model above runs on libnngp_hip_knobs.so for this switch)
    try:
        _, var_fixed = model.predict(xt, cov="diag")
    finally:
        model.debug_set(6, 0)
    assert model.cov_iters() == 0 and np.max(np.abs(var_fixed - var_ref) / var_ref) > 1e-3
    # x_test=None (estimator.py:37-40): variances at the training rows, against the float64 C oracle
    mean_tr, var_tr = ref_model.predict(None, cov="diag")
    assert ref_model.cov_iters() > 0
    orc = c_oracle.fit(x, y, a.w_std, a.b_std, diag_reg=c["diag_reg"])
    mtr_ref, vtr_ref = c_oracle.predict_nngp(orc, x, 1)
    assert G.mean_gate(mean_tr, mtr_ref)[0] < 1e-6
    np.testing.assert_allclose(var_tr, vtr_ref.ravel(), rtol=1e-5)
    model.close(); ref_model.close()


@pytest.mark.parametrize("log2_scale", [-12, 9])
def test_power_of_two_input_scaling_is_exact(log2_scale):
    """With b_std = 0 the ReLU kernel is homogeneous of degree 2 in x, and the relative regulariser scales with it, so
    scaling every query by 2^k leaves the posterior mean unchanged and scales the variance by 4^k.  Powers of two commute
    with every rounding on the path (float64 build, float32 factor, the power-of-two scale of the float16 split, CG), so
    the results must agree to the last bit -- a size-independent check of the scale handling of the float16-pipe Cholesky
    (N = 4500 takes the look-ahead path)."""
    n, m, d = 4500, 200, 24
    x, y = synth.synthetic_queries(n, d, seed=3)
    xt, _ = synth.synthetic_queries(m, d, seed=4)
    a = o.make_arch(2)
    base = GPModel(n, d, a.w_std, a.b_std, diag_reg=1e-3).fit(x, y)
    mean0, var0 = base.predict(xt, cov="diag")
    c = 2.0 ** log2_scale
    scaled = GPModel(n, d, a.w_std, a.b_std, diag_reg=1e-3).fit(x * c, y)
    mean1, var1 = scaled.predict(xt * c, cov="diag")
    assert scaled.info()["refine_iters"] == base.info()["refine_iters"]
    assert np.array_equal(mean1, mean0)
    assert np.array_equal(var1, var0 * c * c)
    base.close(); scaled.close()


def test_float32_factor_breakdown_is_retried_with_a_larger_shift():
    """cond(K + reg I) far beyond 1 / eps32 (d = 2, 4-layer, absolute diag_reg = 10 on a kernel of scale 3e6; found by the
    random sweep at N = 8675): the float32 factorisation clamps pivots and what it leaves gives NaN in the first CG
    step.  nngp_model_factor must notice, refactor with a larger shift, and the float64 solves must still deliver the
    oracle's mean and variance."""
    n, m, d = 3000, 120, 2
    x, y = synth.synthetic_queries(n, d, seed=1)
    xt, _ = synth.synthetic_queries(m, d, seed=101)
    a = o.make_arch(3, 1.6294680774493502, 0.0)
    model = GPModel(n, d, a.w_std, a.b_std, diag_reg=10.0, diag_reg_absolute_scale=True).fit(x, y)
    info = model.info()
    assert model.factor_shift() > info["reg"] and info["clamped_pivots"] == 0, (model.factor_shift(), info)
    assert info["rel_residual"] <= 1e-10, info
    mean, var = model.predict(xt, cov="diag")
    assert model.cov_iters() > 0
    ref = c_oracle.fit(x, y, a.w_std, a.b_std, diag_reg=10.0, absolute=True)
    mean_ref, var_ref = c_oracle.predict_nngp(ref, xt, 1)
    l2, elem = G.mean_gate(mean, mean_ref)
    assert l2 < 1e-6 and elem < 1e-5, (l2, elem, info)
    np.testing.assert_allclose(var, var_ref.ravel(), rtol=1e-4)
    model.close()


def test_early_stopped_cg_mean_correction():
    """With a covariance, predict stops the alpha CG at 1e-6 and corrects the mean through the covariance rows:
    mu = K_td a_k + Z r_k.  The mean must match the float64 oracle as well as the fully converged solve does, info() and
    alpha() must take the solve up again and deliver the converged alpha, and a later mean-only predict must agree."""
    n, m, d = 3000, 100, 24
    x, y = synth.synthetic_queries(n, d, seed=21)
    xt, _ = synth.synthetic_queries(m, d, seed=22)
    a = o.make_arch(2)
    ref = c_oracle.fit(x, y, a.w_std, a.b_std)
    mean_ref, var_ref = c_oracle.predict_nngp(ref, xt, 1)
    model = GPModel(n, d, a.w_std, a.b_std, diag_reg=1e-3)
    model.set_train(x, y); model.build_rows(0, n); model.factor(); model.solve()   # no info(): the solve stays deferred
    mean, var = model.predict(xt, cov="diag")                                      # early stop + correction
    l2, elem = G.mean_gate(mean, mean_ref)
    assert l2 < 1e-7 and elem < 1e-6, (l2, elem)
    np.testing.assert_allclose(var, var_ref.ravel(), rtol=1e-5)
    mean_full, _ = model.predict(xt[:50], cov="full")                              # same state, other rows Z
    assert G.mean_gate(mean_full, mean_ref[:50])[0] < 1e-7
    info = model.info()                                                            # resumes the CG to its tolerance
    assert info["rel_residual"] <= 1e-10 and 3 <= info["refine_iters"] <= 8, info
    assert G.rel_l2(model.alpha().cpu().numpy(), ref["alpha"]) < 1e-8
    mean_only = model.predict(xt, cov=False)
    assert G.mean_gate(mean_only, mean_ref)[0] < 1e-9
    np.testing.assert_allclose(mean_only, mean, rtol=1e-6, atol=1e-6 * np.abs(mean).max())
    model.close()


def test_a_lost_completion_in_a_persistent_solve_costs_an_error_code_not_the_gpu():
    """Round 5: each blocked triangular solve of the posterior (reference: the cho_solve inside predict_fn, train.py:157-158) is one
    persistent launch whose workgroups wait on device counters (csrc/trsm_tickets.hip).  Every wait is bounded: here one item's
    completion is deliberately lost (test hook NNGP_TK_FAULT: ticket 40 is processed but never published, waits give up after 50 ms
    instead of 2 s).  The launch must drain by itself -- every workgroup sees the error word at its next poll or ticket -- the next
    call on the model must return the error code with a message, and the model must then serve correct results through the
    step-by-step solves."""
    import scipy.linalg as sla
    n, rows = 9300, 1024   # (1024 x 9344 right-hand-side entries: above the threshold of the float16-pipe solves)
    x, y = synth.synthetic_queries(n, 24, seed=51)
    model = GPModel(n, 24, [1.0, 1.0], [0.0, 0.0], diag_reg=1e-3).fit(x, y)
    a32, _ = model.factor_buffers()
    L = torch.tril(a32[:n, :n]).double().cpu().numpy()
    rng = np.random.default_rng(3)
    B = rng.standard_normal((rows, n)).astype(np.float32)
    good = model.apply_factor(torch.from_numpy(B.copy()).to(G.dev())).cpu().numpy().astype(np.float64)   # the healthy persistent solve
    os.environ["NNGP_TK_FAULT"] = "40"
    try:
        t0 = time.perf_counter()
        model.apply_factor(torch.from_numpy(B.copy()).to(G.dev()))      # enqueues the faulty launch: asynchronous, no error yet
        torch.cuda.synchronize()
        assert time.perf_counter() - t0 < 5.0, "the launch did not drain within its bounded waits"
    finally:
        del os.environ["NNGP_TK_FAULT"]
    with pytest.raises(Exception) as err:                                # the next call reports it ...
        model.apply_factor(torch.from_numpy(B.copy()).to(G.dev()))
    assert "gave up waiting" in str(err.value)
    X = model.apply_factor(torch.from_numpy(B.copy()).to(G.dev())).cpu().numpy().astype(np.float64)   # ... and the model goes on, step by step
    Xref = sla.solve_triangular(L, B.astype(np.float64).T, lower=True).T
    assert np.linalg.norm(X - Xref) <= 2e-3 * np.linalg.norm(Xref)
    assert np.linalg.norm(good - Xref) <= 2e-3 * np.linalg.norm(Xref)
    model.close()


def test_persistent_solves_with_one_table_per_xcd_return_the_same_bits():
    """Round 5, optional form (NNGP_TK_QUEUES=8, read when a model is created): the item table of a persistent solve as eight tables,
    one per XCD, a workgroup drawing from the table of the XCD it runs on.  The arithmetic of every tile and the order of the updates
    a tile receives are those of the single table, so both halves of the solve must come out bit for bit the same -- and the launch
    must drain (tests/test_host.py proves the protocol; here it runs on the chip)."""
    n, rows = 9300, 1024
    x, y = synth.synthetic_queries(n, 24, seed=52)
    rng = np.random.default_rng(4)
    B = rng.standard_normal((rows, n)).astype(np.float32)
    out = []
    for queues in ("1", "8"):
        os.environ["NNGP_TK_QUEUES"] = queues
        try:
            model = GPModel(n, 24, [1.0, 1.0], [0.0, 0.0], diag_reg=1e-3).fit(x, y)
        finally:
            del os.environ["NNGP_TK_QUEUES"]
        t0 = time.perf_counter()
        fwd = model.apply_factor(torch.from_numpy(B.copy()).to(G.dev())).cpu().numpy()
        both = model.apply_factor(torch.from_numpy(B.copy()).to(G.dev()), both_halves=True).cpu().numpy()
        assert time.perf_counter() - t0 < 5.0
        out.append((fwd, both))
        model.close()
    assert np.isfinite(out[0][0]).all() and np.isfinite(out[0][1]).all()
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
