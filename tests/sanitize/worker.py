"""Runs inside a process with libasan/libubsan preloaded (tests/test_sanitizers.py): drives the sanitizer builds of the
host C++ (csrc/encoder.cpp) and of the C oracle (oracle/nngp_oracle.c) through ctypes and checks their results."""
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
host_so, oracle_so = sys.argv[1], sys.argv[2]

# ---- C oracle ----
import c_oracle
import nngp_oracle as o
c_oracle._LIB_PATH = oracle_so
c_oracle._lib = None
rng = np.random.default_rng(0)
for n, d, n_relu, m in [(1, 1, 1, 1), (7, 3, 1, 2), (65, 20, 3, 9), (130, 5, 2, 17), (257, 64, 1, 33)]:
    a = o.make_arch(n_relu, 1.2, 0.1)
    x = rng.uniform(0, 1000, size=(n, d)); y = rng.normal(size=(n, 2)); xt = rng.uniform(0, 1000, size=(m, d))
    for get in ("nngp", "ntk"):
        # (the NTK of numerically parallel rows -- all rows when d = 1 -- is sqrt(eps)-sensitive in any float64 evaluation)
        tol = 1e-9 if get == "nngp" else 1e-6
        K = c_oracle.kernel_build(x, None, get, a.w_std, a.b_std)
        np.testing.assert_allclose(K, o.kernel_fn(x, None, get, a), rtol=tol, atol=1e-12 * np.abs(K).max())
        Kr = c_oracle.kernel_build(xt, x, get, a.w_std, a.b_std)
        np.testing.assert_allclose(Kr, o.kernel_fn(xt, x, get, a), rtol=tol, atol=1e-12 * np.abs(Kr).max())
    model = c_oracle.fit(x, y, a.w_std, a.b_std)
    mean, cov = c_oracle.predict_nngp(model, xt, 2)
    mref, cref = o.Posterior(x, y, a, diag_reg=1e-3).predict(xt, "nngp", True)
    np.testing.assert_allclose(mean, mref, rtol=1e-7, atol=1e-8)
    np.testing.assert_allclose(cov, cref, rtol=1e-6, atol=1e-8 * np.abs(cref).max())
    _, var = c_oracle.predict_nngp(model, xt, 1)
    np.testing.assert_allclose(var, np.diag(cref), rtol=1e-6, atol=1e-8 * np.abs(cref).max())
_, info = c_oracle.potrf_lower(np.array([[1.0, 2.0], [2.0, 1.0]]))
assert info == 2

# ---- host C++: query-line encoder ----
lib = ctypes.CDLL(host_so)
vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32
lib.nngp_last_error.restype = ctypes.c_char_p
lib.nngp_encoder_create.argtypes = [ctypes.POINTER(vp), ctypes.c_char_p, i32, i32]
lib.nngp_encoder_destroy.argtypes = [vp]
lib.nngp_encoder_dim.argtypes = [vp]
lib.nngp_encoder_encode.argtypes = [vp, ctypes.c_char_p, i64, i32, vp, vp, i64, ctypes.POINTER(i64)]
gold = json.load(open(os.path.join(ROOT, "tests", "golden", "encoder_ref.json")))


def schema_text(tables):
    lines = []
    for t in tables:
        lines.append("table %s" % t["name"])
        for c in t["columns"]:  # [name, kind, lo, hi, num_categories]
            lines.append("cat %s %d" % (c[0], int(c[4])) if c[1] == "categorical" else "num %s %r %r" % (c[0], float(c[2]), float(c[3])))
    return "\n".join(lines).encode()


def encode(h, lines, with_card):
    dim = lib.nngp_encoder_dim(h)
    text = "\n".join(lines).encode()
    cap = text.count(b"\n") + 1
    x = np.full((cap, dim), np.nan); cards = np.full((cap,), np.nan); n = i64(0)
    rc = lib.nngp_encoder_encode(h, text, len(text), int(with_card), x.ctypes.data_as(vp), cards.ctypes.data_as(vp), cap, ctypes.byref(n))
    return rc, x[: n.value], cards[: n.value]


h = vp()
assert lib.nngp_encoder_create(ctypes.byref(h), schema_text(gold["join_tables"]), 64, 0) == 0, lib.nngp_last_error()
rc, x, _ = encode(h, [g["line"] for g in gold["join"]], False)
assert rc == 0 and np.array_equal(x, np.array([g["x"] for g in gold["join"]]))
# malformed input must come back as an error code, never as a memory error
bad = ["", "@", "@@@@@", "nosuch@A,1,2@", gold["join"][0]["line"] + "@", ",,,,@,,,#,,@,,", "t1,t1,t1@@@@@@@@", "\x00\xff\xfe",
       gold["join"][0]["line"][: len(gold["join"][0]["line"]) // 2], "a" * 5000, "@".join(["x"] * 40)]
for b in bad:
    rc, _, _ = encode(h, [b], False)
    rc2, _, _ = encode(h, [b], True)
for trial in range(300):  # byte-level mutations of valid lines
    line = bytearray(gold["join"][trial % len(gold["join"])]["line"].encode())
    for _ in range(rng.integers(1, 4)):
        pos = int(rng.integers(0, len(line)))
        op = int(rng.integers(0, 3))
        if op == 0:
            line[pos] = int(rng.integers(1, 256))
        elif op == 1:
            del line[pos]
        else:
            line.insert(pos, int(rng.choice(list(b"@#,-.0123456789eE"))))
    text = bytes(line).replace(b"\n", b" ")
    n = i64(0)
    dim = lib.nngp_encoder_dim(h)
    xb = np.empty((2, dim)); cb = np.empty((2,))
    lib.nngp_encoder_encode(h, text, len(text), trial & 1, xb.ctypes.data_as(vp), cb.ctypes.data_as(vp), 2, ctypes.byref(n))
# capacity limit: more lines than max_lines is an error, not an overflow
text = "\n".join(g["line"] for g in gold["join"]).encode()
xb = np.empty((1, lib.nngp_encoder_dim(h))); n = i64(0)
assert lib.nngp_encoder_encode(h, text, len(text), 0, xb.ctypes.data_as(vp), None, 1, ctypes.byref(n)) != 0
lib.nngp_encoder_destroy(h)
# single-table mode on the forest golden lines (column ranges come with the fixture's vectors: use the product's table)
sys.path.insert(0, ROOT)
for bad_schema in [b"", b"table", b"num a 0 1", b"table t\nnum a", b"table t\ncat c x", b"table a\ntable b\n"]:
    hh = vp()
    rc = lib.nngp_encoder_create(ctypes.byref(hh), bad_schema, 64, 1)
    if rc == 0:
        lib.nngp_encoder_destroy(hh)
assert lib.nngp_encoder_create(ctypes.byref(h), b"table t\nnum a 0 1", 0, 0) != 0
print("SANITIZE_WORKER_OK")
