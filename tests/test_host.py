"""Host-side logic and the C-ABI surface -- no GPU needed (no compute call is made)."""
import ctypes
import hashlib
import io
import json
import os
import re
import contextlib

import numpy as np
import pytest

import nngp_src_amd as pkg
from nngp_src_amd import _lib, encoder as enc, stax, synth, util, distributed

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol():
    lib = _lib.load()
    assert lib.nngp_version() == 1
    header = open(os.path.join(ROOT, "include", "nngp_hip.h")).read()
    declared = set(re.findall(r"\b(nngp_[a-z0-9_]+)\s*\(", header))
    declared -= {"nngp_model", "nngp_comm"}
    # nngp_debug_set is declared under #ifdef NNGP_TIMING_KNOBS: exported by libnngp_hip_knobs.so only
    assert declared == set(_lib.ABI_SYMBOLS) | {"nngp_debug_set"}, declared ^ set(_lib.ABI_SYMBOLS)
    for name in _lib.ABI_SYMBOLS:
        assert hasattr(lib, name), name
    assert not hasattr(lib, "nngp_debug_set"), "the product library must not export the timing knobs"
    knobs = _lib.load(knobs=True)
    for name in _lib.ABI_SYMBOLS + ("nngp_debug_set",):
        assert hasattr(knobs, name), name
    assert knobs.nngp_debug_set(99, 1) != 0 and b"key out of range" in knobs.nngp_last_error()
    assert ctypes.sizeof(_lib.NngpArch) == 8 + 2 * 16 * 8
    assert ctypes.sizeof(_lib.NngpFitInfo) == 3 * 8 + 2 * 4 + 2 * 8


def test_argument_validation_without_gpu():
    lib = _lib.load()
    arch = _lib.make_arch([1.0, 1.0], [0.0, 0.0])
    rc = lib.nngp_kernel_build(None, 4, None, 4, 3, ctypes.byref(arch), 1, None, None, 4, 0, 4, None)
    assert rc != 0 and b"x1" in lib.nngp_last_error()
    h = ctypes.c_void_p()
    rc = lib.nngp_model_create(ctypes.byref(h), 0, 0, 3, 1, ctypes.byref(arch), 1, 1e-3, 0)
    assert rc != 0 and b"bad sizes" in lib.nngp_last_error()
    bad = _lib.NngpArch(); bad.n_dense = 0
    rc = lib.nngp_model_create(ctypes.byref(h), 8, 0, 3, 1, ctypes.byref(bad), 1, 1e-3, 0)
    assert rc != 0 and b"n_dense" in lib.nngp_last_error()
    with pytest.raises(_lib.NngpError):
        _lib.check(rc)


def test_comm_entry_points_validate_arguments_without_gpu():
    """Section-e entry points (RCCL bound lazily by dlopen): argument errors come back as codes, nothing is loaded."""
    lib = _lib.load()
    h = ctypes.c_void_p()
    assert lib.nngp_comm_create(ctypes.byref(h), None, 2, 0) != 0 and b"NULL" in lib.nngp_last_error()
    ident = ctypes.create_string_buffer(128)
    assert lib.nngp_comm_create(ctypes.byref(h), ident, 2, 5) != 0 and b"world/rank" in lib.nngp_last_error()
    assert lib.nngp_allgather_rows(None, 8, 8, _lib.DTYPE_F64, None, None) != 0
    assert lib.nngp_bcast(None, 8, _lib.DTYPE_F32, 0, None, None) != 0
    assert lib.nngp_comm_destroy(None) == 0


def test_product_never_touches_the_oracle():
    """The product path must not import, link or execute anything under oracle/ (and has no CPU fallback)."""
    src = os.path.join(ROOT, "nngp-src_amd")
    for dirpath, _, files in os.walk(src):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                text = open(os.path.join(dirpath, fn), errors="replace").read()
                assert not re.search(r"nngp_oracle|c_oracle|libnngp_oracle|oracle/", text), fn


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    _, _, kernel_fn = stax.serial(stax.Dense(512), stax.Relu(), stax.Dense(1))
    with pytest.raises(_lib.NngpError, match="no CPU fallback"):
        kernel_fn(np.ones((4, 3)), None, "nngp")
    with pytest.raises(_lib.NngpError):
        pkg.predict.gradient_descent_mse_ensemble(kernel_fn, np.ones((4, 3)), np.ones((4, 1)), diag_reg=1e-3)(
            x_test=np.ones((2, 3)), get="nngp", compute_cov=True)


def test_stax_topology():
    init_fn, apply_fn, kernel_fn = stax.serial(stax.Dense(512, W_std=1.5, b_std=0.1), stax.Relu(), stax.Dense(64),
                                               stax.Relu(), stax.Dense(1))
    assert kernel_fn.w_std == (1.5, 1.0, 1.0) and kernel_fn.b_std == (0.1, 0.0, 0.0) and kernel_fn.n_relu == 2
    assert pkg.batch(kernel_fn, device_count=0, batch_size=0) is kernel_fn  # the reference configuration
    with pytest.raises(NotImplementedError):
        stax.serial(stax.Relu(), stax.Dense(1))
    with pytest.raises(NotImplementedError):
        stax.serial(stax.Dense(4), stax.Dense(1))
    shape, params = init_fn(0, (10, 7))
    assert shape == (10, 1) and [w.shape for w, _ in params] == [(7, 512), (512, 64), (64, 1)]
    assert apply_fn(params, np.ones((10, 7))).shape == (10, 1)


def test_finite_width_network_approaches_closed_form():
    import nngp_oracle as o  # tests may use the oracle as the checker
    rng = np.random.default_rng(0)
    x = rng.uniform(0, 1, size=(5, 6))
    init_fn, apply_fn, _ = stax.serial(stax.Dense(4096), stax.Relu(), stax.Dense(1))
    outs = []
    for seed in range(300):
        _, params = init_fn(seed, x.shape)
        outs.append(apply_fn(params, x)[:, 0])
    emp = np.cov(np.array(outs).T, bias=True)
    K = o.kernel_fn(x, None, "nngp", o.make_arch(1))
    assert np.abs(emp - K).max() / np.abs(K).max() < 0.35  # 300 draws: ~10-25 % sampling error


def test_encoder_matches_reference_golden(golden_dir):
    gold = json.load(open(os.path.join(golden_dir, "encoder_ref.json")))
    loader = enc.GeneralQueryEncoder()
    for item in gold["forest"]:
        pred_list, card = loader.parse_line(item["line"])
        assert card > 0
        np.testing.assert_array_equal(loader.transform_to_1d_array(pred_list), np.array(item["x"]))
    tables = [enc.TableEncoder(t["name"], [enc.ColumnSpec(c[0], c[1], c[2], c[3], int(c[4])) for c in t["columns"]], 64)
              for t in gold["join_tables"]]
    je = enc.NNGPEncoder(tables)
    assert je.join_feat_dim == gold["join_feat_dim"]
    for item in gold["join"]:
        np.testing.assert_array_equal(je.parse_line_without_card_then_encode(item["line"]), np.array(item["x"]))
        tids, preds, joins, card = je.parse_line(item["line"] + "@77")
        assert card == 77
    with pytest.raises(AssertionError, match="Query Format Error"):
        je.parse_line_without_card_then_encode("orders,cust@o_id,1,0")


def test_encoder_defaults_and_forest_data(golden_dir):
    loader = enc.GeneralQueryEncoder()
    assert loader.total_feat_dim == 20
    np.testing.assert_array_equal(loader.transform_to_1d_array([]), np.tile([0.0, 1000.0], 10))
    g = np.load(os.path.join(golden_dir, "forest_queries.npz"))
    bounds, cards = g["bounds"], g["cards"]
    assert bounds.shape == (18000, 10, 2) and cards.min() >= 1
    assert list(g["files"])[:2] == ["query_10.txt", "query_2.txt"]  # sorted(os.listdir) order
    sent = np.iinfo(np.int32).min
    q0 = [(c, float(bounds[0, c, 0]), float(bounds[0, c, 1])) for c in range(10) if bounds[0, c, 0] != sent]
    assert len(q0) == 10  # query_10.txt: ten predicates
    x0 = loader.transform_to_1d_array(q0)
    assert (x0 >= 0).all() and (x0 <= 1000).all()


def test_join_loader_and_aux_filter(tmp_path):
    tables = [enc.TableEncoder("a", [enc.numerical("k", 0, 10), enc.numerical("v", 0, 100)], 64),
              enc.TableEncoder("b", [enc.numerical("k", 0, 10), enc.categorical("c", 70)], 64)]
    je = enc.NNGPEncoder(tables)
    assert je.feat_dim == 4 + 2 + 2 + 3
    (tmp_path / "q1.txt").write_text("a@v,50,10@@100\na,b@@c,1,69@a,b,k@8\n")
    (tmp_path / "join_query_aux.txt").write_text("a@v,5,1@@4@2.0@0.1\na@v,9,2@@16@500.0@0.1\n")
    q, cards, infos = je.load_queries(str(tmp_path), use_aux=False)
    assert cards == [100, 8] and infos[1].num_joins == 1 and infos[1].num_table == 2
    q, cards, infos = je.load_queries(str(tmp_path), use_aux=True, q_error_threshold=100.0, coef_var_threshold=1.0)
    assert cards == [16, 100, 8]  # the well-predicted aux line (q_error 2 < 100, cv 0.1 < 1) is dropped
    X, Y = je.transform_to_arrays(q, cards)
    assert X.shape == (3, 11) and Y[0, 0] == 4.0
    np.testing.assert_array_equal(X[2, 6:8], [float(2 ** 62), float(2 ** 58)])  # factorised: MSB-first chunks of 64
    assert X[2, 10] == 1.0 and X[1, 10] == 0.0


def test_split_pin(golden_dir):
    pin = json.load(open(os.path.join(golden_dir, "split_pin.json")))
    idx = np.asarray(util.split_indices(pin["n"], pin["seed"]), dtype=np.int64)
    assert idx[:8].tolist() == [2636, 10465, 12327, 1458, 15087, 17515, 4, 10311] == pin["first"]
    assert hashlib.sha256(idx.tobytes()).hexdigest() == pin["sha256"]
    assert pin["sha256"].startswith("9bc86dcff2246369")
    X = np.arange(40.0).reshape(20, 2); Y = np.arange(20.0).reshape(20, 1)
    with contextlib.redirect_stdout(io.StringIO()):
        out = util.train_test_val_split(X, Y, 0.6, 0.2, all_query_infos=list(range(20)), max_num_train=5)
    assert out[0].shape == (5, 2) and out[3].shape == (4, 2) and out[6].shape == (4, 2) and len(out[2]) == 5


def test_prediction_statistics_format():
    errors = np.log2(np.array([0.5, 1.0, 2.0, 4.0, 8.0]))
    infos = [util.QueryInfo(1, 0, p, False, False) for p in (2, 2, 3, 3, 3)]
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        util.PredictionStatistics().get_prediction_details(errors, infos, partition_keys="num_table")
    text = buf.getvalue()
    assert "Query attributes:num_table=1" in text and "# Queries = 5" in text
    assert "Min/Max: 0.500000000000000 / 8.000000000000000" in text
    assert "Mean: 3.10000000" in text and "Median: 2.00000000" in text
    assert "25%/75% Quantiles: 1.00000000 / 4.00000000" in text
    # more than 6 partitions are merged pairwise (util.py:129-140)
    infos = [util.QueryInfo(1, 0, p, False, False) for p in range(8)]
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        util.PredictionStatistics().get_prediction_details(np.zeros(8), infos, partition_keys="num_predicates")
    assert buf.getvalue().count("Query attributes") == 4 and "# Queries = 2" in buf.getvalue()


def test_synthetic_generator_is_deterministic():
    X, Y = synth.synthetic_queries(256, 64, seed=0)
    X2, Y2 = synth.synthetic_queries(256, 64, seed=0)
    assert np.array_equal(X, X2) and np.array_equal(Y, Y2)
    assert hashlib.sha256(X.tobytes()).hexdigest()[:16] == synth_pin()
    assert X.min() >= 0 and X.max() <= 1000 and (X[:, 0::2] >= 0).all()
    active = X[:, 1::2] != 1000.0
    assert active.sum(axis=1).min() >= 1 and active.sum(axis=1).max() <= 10
    assert (X[:, 0::2] >= np.where(active, X[:, 1::2], 0)).all()  # upper >= lower on active pairs
    assert 0 <= Y.min() and Y.max() <= 19.2
    Xj, _ = synth.synthetic_queries(64, 256, seed=1, join_block=True)
    assert set(np.unique(Xj[:, 256 - 96:][:, 2::3])) <= {0.0, 1.0}


def synth_pin():
    return "618243c2ba9827d5"


def test_row_partition():
    for n, w in [(1000, 8), (128, 3), (7, 8), (65536, 8)]:
        parts = [distributed.row_partition(n, w, r) for r in range(w)]
        assert parts[0][0] == 0 and parts[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
        assert all(r1 - r0 <= distributed.row_chunk(n, w) for r0, r1 in parts)


def test_native_encoder_is_bit_identical(golden_dir):
    """The C++ encoder in libnngp_hip.so (host code, no GPU needed) against the golden vectors the REFERENCE encoder
    produced, and against the Python restatement on fuzzed lines."""
    gold = json.load(open(os.path.join(golden_dir, "encoder_ref.json")))
    single = enc.NativeEncoder.from_encoder(enc.GeneralQueryEncoder())
    assert single.feat_dim == 20
    x, cards = single.encode_lines([g["line"] for g in gold["forest"]], with_card=True)
    np.testing.assert_array_equal(x, np.array([g["x"] for g in gold["forest"]]))
    assert cards[0] == int(gold["forest"][0]["line"].split("@")[1])
    tables = [enc.TableEncoder(t["name"], [enc.ColumnSpec(c[0], c[1], c[2], c[3], int(c[4])) for c in t["columns"]], 64)
              for t in gold["join_tables"]]
    py = enc.NNGPEncoder(tables)
    nat = enc.NativeEncoder.from_encoder(py)
    assert nat.feat_dim == py.feat_dim
    np.testing.assert_array_equal(nat.encode_lines([g["line"] for g in gold["join"]]), np.array([g["x"] for g in gold["join"]]))
    x, cards = nat.encode_lines([g["line"] + "@%d" % (7 + i) for i, g in enumerate(gold["join"])], with_card=True)
    np.testing.assert_array_equal(x, np.array([g["x"] for g in gold["join"]]))
    assert cards.tolist() == [7 + i for i in range(len(gold["join"]))]
    # fuzz against the Python encoder
    rng = np.random.default_rng(0)
    names = [t.table_name for t in tables]
    lines = []
    for _ in range(200):
        k = rng.integers(1, len(tables) + 1)
        tids = sorted(rng.choice(len(tables), size=k, replace=False).tolist())
        preds = []
        for t in tids:
            ps = []
            for c in tables[t].columns:
                if rng.random() < 0.5:
                    continue
                if c.kind == "categorical":
                    cats = rng.choice(c.num_categories, size=rng.integers(1, 4), replace=False)
                    ps.append("%s,%s" % (c.name, ",".join(str(int(v)) for v in cats)))
                else:
                    a, b = sorted(rng.uniform(c.lo - 5, c.hi + 5, 2), reverse=True)
                    ps.append("%s,%.3f,%.3f" % (c.name, a, b))
            preds.append("#".join(ps))
        joins = [j for j in py.all_join_infos if j.t1_id in tids and j.t2_id in tids and rng.random() < 0.7]
        jstr = "#".join("%s,%s,%s" % (names[j.t1_id], names[j.t2_id], j.col_name) for j in joins)
        lines.append(",".join(names[t] for t in tids) + "@" + "@".join(preds) + "@" + jstr)
    want = np.array([py.parse_line_without_card_then_encode(l) for l in lines])
    np.testing.assert_array_equal(nat.encode_lines(lines), want)
    with pytest.raises(_lib.NngpError, match="Query Format Error"):
        nat.encode_lines(["orders,cust@o_id,1,0"])
    with pytest.raises(_lib.NngpError, match="unknown column"):
        single.encode_lines(["Z,1,0@5"], with_card=True)


def test_threefry_known_answers_and_the_restated_jax_draw():
    """jaxrand.py restates jax.random.choice(PRNGKey(10), n, (k,), replace=False, p) of the pinned jax 0.3.23
    (reference active/ActiveLearner.py:50-53).  Pins: the Random123 known-answer vectors of Threefry-2x32-20 (the three
    vectors jax's own random_test.py checks); jax.random.uniform(PRNGKey(0)) = 0.41845703 in float32 -- the widely quoted first
    draw, which exercises the key layout, the counter layout and the mantissa fill of the same pipeline at 32 bits; then the
    statistics of the 64-bit draw.  Unpinned against jax itself (not installable here)."""
    from nngp_src_amd import jaxrand as J
    u32 = lambda v: np.array([v], dtype=np.uint32)
    with np.errstate(over="ignore"):
        for key, ctr, want in [((0, 0), (0, 0), (0x6b200159, 0x99ba4efe)),
                               ((0xffffffff, 0xffffffff), (0xffffffff, 0xffffffff), (0x1cb996fc, 0xbb002be7)),
                               ((0x13198a2e, 0x03707344), (0x243f6a88, 0x85a308d3), (0xc4923a9c, 0x483df7a0))]:
            y0, y1 = J.threefry2x32((np.uint32(key[0]), np.uint32(key[1])), u32(ctr[0]), u32(ctr[1]))
            assert (int(y0[0]), int(y1[0])) == want
        # float32 scalar draw of jax: one block on the padded counter pair (0, 0), first word, 23 mantissa bits
        y0, _ = J.threefry2x32(J.prng_key(0), u32(0), u32(0))
    f = ((y0 >> np.uint32(9)) | np.uint32(0x3F800000)).view(np.float32) - np.float32(1.0)
    assert abs(float(f[0]) - 0.41845703) < 1e-7
    assert J.prng_key(10) == (0, 10) and J.prng_key((3 << 32) | 9) == (3, 9)
    u = J.uniform64(J.prng_key(10), 200000)
    assert 0.0 < u.min() and u.max() < 1.0 and abs(u.mean() - 0.5) < 3e-3 and abs(np.mean(u < 0.1) - 0.1) < 3e-3
    # the draw: distinct indices, deterministic, proportional to p (first draw frequencies over many seeds)
    p = np.array([0.5, 0.25, 0.125, 0.125])
    first = np.array([J.choice_without_replacement(seed, 4, 2, p)[0] for seed in range(4000)])
    freq = np.bincount(first, minlength=4) / 4000.0
    assert np.abs(freq - p).max() < 0.03
    a = J.choice_without_replacement(10, 1000, 100, np.full(1000, 1e-3))
    assert len(set(a.tolist())) == 100 and np.array_equal(a, J.choice_without_replacement(10, 1000, 100, np.full(1000, 1e-3)))
    # zero-probability entries are never drawn
    pz = np.array([0.0, 0.5, 0.0, 0.5])
    assert set(J.choice_without_replacement(10, 4, 2, pz).tolist()) == {1, 3}


def test_active_train_cli_flags_and_split():
    """active_train.py mirrors the reference driver's flags (active/active_train.py:54-107) -- including `--biased_sample` parsed with
    type=bool, i.e. any non-empty value is True -- and its 20 / 60 / 20 split of the seed-10 shuffle (:26-27)."""
    from nngp_src_amd import active_train
    from nngp_src_amd.util import train_test_val_split
    a = active_train.parse_args([])
    assert a.biased_sample is True and a.budget == 1000 and a.active_iters == 3 and a.kernel_type == "nngp" and a.chunk_size == 10
    assert active_train.parse_args(["--biased_sample", "False"]).biased_sample is True   # the reference's type=bool
    assert active_train.parse_args(["--top_k"]).biased_sample is False
    b = active_train.parse_args(["--budget", "50", "--active_iters", "2", "--kernel_type", "ntk", "--relations", "a,b"])
    assert (b.budget, b.active_iters, b.kernel_type, b.join_query) == (50, 2, "ntk", True)
    X = np.arange(1000, dtype=np.float64).reshape(500, 2)
    Y = np.arange(500, dtype=np.float64).reshape(500, 1)
    out = active_train.split_20_60_20(X, Y)
    ref = train_test_val_split(X, Y, train_frac=0.2, test_frac=0.6)
    assert out[0].shape[0] == 100 and out[3].shape[0] == 300 and out[6].shape[0] == 100
    for g, r in zip(out, ref):
        assert (g is None and r is None) or np.array_equal(np.asarray(g), np.asarray(r))
    from nngp_src_amd.active import ActiveLearner
    assert ActiveLearner().biased_sample is True  # the reference's default


def _replay_ticket_table(items, mt, nb, tail, backward):
    """Walk a ticket table in order with the kernel's own counters (csrc/trsm_tickets.hip): an item may start only if the wait
    condition the kernel polls already holds through items with LOWER tickets -- then no workgroup can ever wait on work nobody has
    started -- and at the end every tile has received every block column exactly once, in order."""
    KQ = 4          # split items per (row tile, block column): 32 rows each
    ct = lambda J: tail if J == nb - 1 else 8
    ctiles = (nb - 1) * 8 + tail
    pos = (lambda J: nb - 1 - J) if backward else (lambda J: J)
    xd = np.zeros((mt, nb), np.int64); xs = np.zeros((mt, nb), np.int64); bs = np.zeros((mt, nb), np.int64)
    up = np.zeros((mt, ctiles), np.int64)
    seen = set()
    big = 0
    for t, (word, r, cq, J) in enumerate(items):
        typ, npan = word & 15, (word >> 4) & 15
        key = (typ, r, cq, J)
        assert key not in seen, ("duplicate item", t, key)
        seen.add(key)
        assert r >= 0 and 0 <= J < nb
        if typ in (3, 4):    # update: 128 x 128 tile (3) or the 2 x 2 group of tiles (2 r, 2 r + 1) x (2 cq, 2 cq + 1) (4); panels J, J -/+ 1, ...
            rows = [r] if typ == 3 else [2 * r, 2 * r + 1]
            cols = [cq] if typ == 3 else [2 * cq, 2 * cq + 1]
            assert 1 <= npan <= 4 and rows[-1] < mt and cols[-1] < ctiles and cols[0] // 8 == cols[-1] // 8
            big += typ == 4
            first = J + npan - 1 if backward else J - npan + 1      # earliest block column of the item
            assert 0 <= first < nb
            Jt = cols[0] // 8
            assert cols[-1] - 8 * Jt < ct(Jt)
            assert pos(J) < pos(Jt), ("an update from a block column that is not solved before its target", t)
            if npan > 1:
                assert ct(J) == 8 and ct(first) == 8, ("a tail-width panel grouped with others", t)
            for rr in rows:
                assert xs[rr, J] == KQ, ("update before the split copy of its latest block column", t)
                for cc in cols:
                    assert up[rr, cc] == pos(first), ("update out of order", t, up[rr, cc], pos(first))
                    up[rr, cc] += npan
        elif typ == 5:  # the chain update of a tile from the split rows of the block right before its own
            assert npan == 1 and r < mt and 0 <= cq < ctiles
            Jt = cq // 8
            assert cq - 8 * Jt < ct(Jt) and pos(J) == pos(Jt) - 1, ("a merged update that does not come from the previous block", t)
            assert bs[r, J] == KQ, ("merged update before the split rows of its source block", t)
            assert up[r, cq] == pos(J), ("merged update out of order", t)
            up[r, cq] += 1
        elif typ == 1:  # diagonal-product tile
            assert r < mt and cq // 8 == J and cq - 8 * J < ct(J)
            assert bs[r, J] == KQ, ("diagonal product before its operand is split", t)
            xd[r, J] += 1
        elif typ == 2:  # split of a solved block
            assert r < mt and 0 <= cq < KQ and xd[r, J] == ct(J), ("split of a block whose diagonal tiles are not all done", t)
            xs[r, J] += 1
        else:           # split of an updated block
            assert typ == 0 and r < mt and 0 <= cq < KQ
            for c in range(8 * J, 8 * J + ct(J)):
                assert up[r, c] == pos(J), ("split of a block that has not received all its updates", t)
            bs[r, J] += 1
    for r in range(mt):
        for J in range(nb):
            assert xd[r, J] == ct(J) and xs[r, J] == KQ and bs[r, J] == KQ
            for c in range(8 * J, 8 * J + ct(J)):
                assert up[r, c] == pos(J)
    return len(items), big


@pytest.mark.parametrize("mt,nb,tail,workers", [(8, 32, 8, 512), (1, 2, 8, 512), (2, 3, 1, 16), (29, 11, 5, 512), (8, 8, 8, 512),
                                                (3, 10, 8, 1), (8, 64, 8, 512), (5, 9, 3, 97)])
def test_persistent_solve_ticket_tables_only_wait_on_lower_tickets(mt, nb, tail, workers):
    """Round 5: the posterior's blocked triangular solves (reference: the cho_solve inside predict_fn, train.py:157-158) run as one
    persistent launch each, whose workgroups take work items in ticket order and wait on device counters.  Hang freedom rests on ONE
    property of the host-built table -- every item's dependencies hold lower tickets -- which this test proves for the bench shapes
    (cfg3: 8 x 32, cfg4: 8 x 64, cfg2: 8 x 8, the forest run: 29 x 11 with a 640-wide tail), degenerate ones and odd worker counts,
    forward and backward, by replaying the table against the kernel's wait conditions."""
    lib = _lib.load()
    for backward, merged in ((0, 1), (1, 1), (0, 0), (1, 0)):
        count = ctypes.c_int64(0)
        assert lib.nngp_trsm_ticket_order(mt, nb, tail, backward, workers, merged, None, 0, ctypes.byref(count)) == 0, lib.nngp_last_error()
        buf = np.zeros((count.value, 4), np.int32)
        assert lib.nngp_trsm_ticket_order(mt, nb, tail, backward, workers, merged, buf.ctypes.data_as(ctypes.c_void_p), count.value,
                                          ctypes.byref(count)) == 0, lib.nngp_last_error()
        assert merged == int((buf[:, 0] & 15 == 5).any()) or (mt < 2 and merged == 1 and not (buf[:, 0] & 15 == 5).any())
        n, big = _replay_ticket_table([tuple(int(v) for v in row) for row in buf], mt, nb, tail, bool(backward))
        assert n == count.value
        assert big > 0 or mt < 2 or nb < 4, "no 256 x 256 bulk items in a shape that has room for them"
        # the table is a pure function of the shape: run-to-run bitwise reproducibility of the solves depends on it
        buf2 = np.zeros_like(buf)
        assert lib.nngp_trsm_ticket_order(mt, nb, tail, backward, workers, merged, buf2.ctypes.data_as(ctypes.c_void_p), count.value,
                                          ctypes.byref(count)) == 0
        assert np.array_equal(buf, buf2)
    assert lib.nngp_trsm_ticket_order(0, 4, 8, 0, 512, 1, None, 0, ctypes.byref(count)) != 0


class _TicketState:
    """The kernel's counters (csrc/trsm_tickets.hip) for the draw-protocol test: `ready` is the wait condition an item polls,
    `apply` what its completion publishes."""
    KQ = 4

    def __init__(self, mt, nb, tail, backward):
        self.mt, self.nb, self.tail, self.backward = mt, nb, tail, backward
        self.ctiles = (nb - 1) * 8 + tail
        self.xd = np.zeros((mt, nb), np.int64); self.xs = np.zeros((mt, nb), np.int64); self.bs = np.zeros((mt, nb), np.int64)
        self.up = np.zeros((mt, self.ctiles), np.int64)

    def ct(self, J):
        return self.tail if J == self.nb - 1 else 8

    def pos(self, J):
        return self.nb - 1 - J if self.backward else J

    def _update_targets(self, item):
        word, r, cq, J = item
        typ, npan = word & 15, (word >> 4) & 15
        rows = [r] if typ in (3, 5) else [2 * r, 2 * r + 1]
        cols = [cq] if typ in (3, 5) else [2 * cq, 2 * cq + 1]
        first = J if typ == 5 else (J + npan - 1 if self.backward else J - npan + 1)
        return typ, npan, rows, cols, first

    def ready(self, item):
        word, r, cq, J = item
        typ = word & 15
        if typ in (3, 4, 5):
            typ, npan, rows, cols, first = self._update_targets(item)
            src = self.bs if typ == 5 else self.xs
            return all(src[rr, J] == self.KQ and all(self.up[rr, cc] == self.pos(first) for cc in cols) for rr in rows)
        if typ == 1:
            return self.bs[r, J] == self.KQ
        if typ == 2:
            return self.xd[r, J] == self.ct(J)
        return all(self.up[r, c] == self.pos(J) for c in range(8 * J, 8 * J + self.ct(J)))

    def apply(self, item):
        word, r, cq, J = item
        typ = word & 15
        if typ in (3, 4, 5):
            typ, npan, rows, cols, first = self._update_targets(item)
            for rr in rows:
                for cc in cols:
                    self.up[rr, cc] += npan
        elif typ == 1:
            self.xd[r, J] += 1
        elif typ == 2:
            self.xs[r, J] += 1
        else:
            self.bs[r, J] += 1


@pytest.mark.parametrize("mt,nb,tail", [(8, 32, 8), (8, 8, 8), (4, 16, 8), (29, 11, 5), (16, 6, 8), (3, 5, 2), (1, 3, 8)])
def test_persistent_solve_queues_drain_whoever_draws_from_them(mt, nb, tail):
    """On MI355X the table is eight tables, one per XCD (nngp_trsm_ticket_queues): a workgroup draws from the table of the XCD it runs
    on and, once that has run out, from the others.  What keeps such a launch from hanging: (1) the global start order of the host's
    schedule has every item behind its dependencies (the single-table property), (2) every table holds its items in that order.
    Then the earliest unfinished item is at the head of its table and the workgroups drawing there hold nothing later than it.  The
    test checks (1) and (2) and then RUNS the draw protocol -- any number of workgroups per XCD, at least one (the product uses the
    eight tables only for launches of >= 224 workgroups, which the hardware deals round-robin over the XCDs; an XCD without any would
    end in the kernel's bounded waits and the step-by-step fallback), items finishing in random order -- to the end."""
    lib = _lib.load()
    rng = np.random.default_rng(mt * 1000 + nb * 10 + tail)
    for backward, merged in ((0, 0), (1, 0), (0, 1)):
        count = ctypes.c_int64(0)
        assert lib.nngp_trsm_ticket_queues(mt, nb, tail, backward, 256, merged, 8, None, None, 0, ctypes.byref(count)) == 0, lib.nngp_last_error()
        buf = np.zeros((count.value, 4), np.int32)
        qof = np.zeros(count.value, np.int32)
        assert lib.nngp_trsm_ticket_queues(mt, nb, tail, backward, 256, merged, 8, buf.ctypes.data_as(ctypes.c_void_p),
                                           qof.ctypes.data_as(ctypes.c_void_p), count.value, ctypes.byref(count)) == 0, lib.nngp_last_error()
        items = [tuple(int(v) for v in row) for row in buf]
        n, _ = _replay_ticket_table(items, mt, nb, tail, bool(backward))   # (1)
        assert n == count.value and qof.min() >= 0 and qof.max() < 8
        if mt >= 8:
            assert len(set(qof.tolist())) == 8, "a shape with four pairs of row tiles leaves a queue empty"
        # the bulk items of one pair of row tiles sit in the queues of that pair only
        for (word, r, cq, J), q in zip(items, qof):
            if word & 15 == 4 and mt // 2 in (1, 2, 4, 8):
                per = 8 // (mt // 2)
                assert r * per <= q < (r + 1) * per, "a bulk item outside the queues of its pair of row tiles"
        tables = [[i for i in range(n) if qof[i] == q] for q in range(8)]   # (2): by construction of the device tables (tk_solve)
        for workers_per_q in ([1] * 8, [4] * 8, [32] * 8, [2, 1, 3, 1, 1, 5, 1, 2]):
            state = _TicketState(mt, nb, tail, bool(backward))
            head = [0] * 8
            held = []   # (item index, home queue of the worker)
            idle = [q for q in range(8) for _ in range(workers_per_q[q])]
            done = 0

            def draw(q):
                for k in range(8):
                    qq = (q + k) % 8
                    if head[qq] < len(tables[qq]):
                        head[qq] += 1
                        return tables[qq][head[qq] - 1]
                return None

            while done < n:
                still_idle = []
                for q in idle:
                    i = draw(q)
                    if i is None:
                        still_idle.append(q)
                    else:
                        held.append((i, q))
                idle = still_idle
                runnable = [k for k, (i, q) in enumerate(held) if state.ready(items[i])]
                assert runnable, ("the launch would hang", mt, nb, tail, backward, merged, workers_per_q, done, n)
                k = runnable[int(rng.integers(len(runnable)))]
                i, q = held.pop(k)
                state.apply(items[i])
                done += 1
                idle.append(q)


def test_persistent_solve_kernel_has_no_loop_the_threads_of_a_wave_leave_apart(tmp_path):
    """The first GPU run of csrc/trsm_tickets.hip hung with every workgroup holding its first ticket: hipcc had threaded thread 0's
    path (publish -> next ticket) across the back edge of the item loop, which turned the rest of the loop into an INNER loop that the
    other threads of wave 0 never leave -- lane 0 then waits for its wave to reconverge, forever, outside every bounded wait.  The
    kernel now has one thread-0 block per iteration; this test keeps it that way by looking at what hipcc makes of it: the item loop
    at depth 1 and, inside it, only the poll loop, the stage loops and the barrier-only loop at depth 2 -- no depth-3 loop."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    src = os.path.join(ROOT, "nngp-src_amd", "csrc", "trsm_tickets.hip")
    out = str(tmp_path / "tk.s")
    subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-S", "--cuda-device-only", src, "-o", out],
                   check=True, capture_output=True, timeout=600)
    text = open(out).read()
    start = text.index("k_trsm_tickets")
    end = text.index(".end_amdhsa_kernel", start) if ".end_amdhsa_kernel" in text[start:] else len(text)
    body = text[start:end]
    body = body[:body.index("s_endpgm")] if "s_endpgm" in body else body
    depths = [int(m) for m in re.findall(r"Loop Header: Depth=(\d+)", body)]
    assert depths and max(depths) == 2 and depths.count(1) == 1, depths
    assert "s_memrealtime" in body and "buffer_inv sc1" in body and "buffer_wbl2 sc1" in body   # bounded polls, acquire, release
