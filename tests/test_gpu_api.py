"""The reference-shaped Python surface on the GPU (-m gpu): stax / batch / predict / Estimator / train CLI."""
import contextlib
import io
import os
import types

import numpy as np
import pytest
import torch

import nngp_oracle as o
import nngp_src_amd as nt
from nngp_src_amd import encoder as enc, stax, synth, train as train_cli
from nngp_src_amd.estimator import Estimator
from nngp_src_amd.model import GPModel
from nngp_src_amd import _lib as _lib_mod

import gpu_util as G

pytestmark = pytest.mark.gpu


def test_reference_call_sequence():
    """The exact call sequence of train.py:161-172,157-158 with the reference's hyper-parameters."""
    x, y = synth.synthetic_queries(400, 20, seed=0)
    xt, _ = synth.synthetic_queries(33, 20, seed=1)
    init_fn, apply_fn, kernel_fn = stax.serial(stax.Dense(512), stax.Relu(), stax.Dense(1))
    kernel_fn = nt.batch(kernel_fn, device_count=0, batch_size=0)
    predict_fn = nt.predict.gradient_descent_mse_ensemble(kernel_fn, x, y, diag_reg=1e-3)
    pred_mean, pred_cov = predict_fn(x_test=xt, get="nngp", compute_cov=True)
    assert pred_mean.shape == (33, 1) and pred_cov.shape == (33, 33)
    post = o.Posterior(x, y, o.make_arch(1), diag_reg=1e-3)
    m_ref, c_ref = post.predict(xt, "nngp", True)
    assert G.mean_gate(pred_mean, m_ref)[0] < 1e-6
    np.testing.assert_allclose(np.sqrt(np.diag(pred_cov)), np.sqrt(np.diag(c_ref)), rtol=1e-5)
    g_ntk = predict_fn(x_test=xt, get="ntk", compute_cov=True)   # train.py --kernel_type ntk
    m_ntk, c_ntk = post.predict(xt, "ntk", True)
    assert G.mean_gate(g_ntk.mean, m_ntk)[0] < 1e-6
    assert np.abs(g_ntk.covariance - c_ntk).max() < 1e-5 * np.abs(np.diag(c_ntk)).max()
    # mean only / ntk / tuple get / kernel_fn forms
    # (with a covariance the CG for alpha stops early and the mean is corrected through the covariance rows: same value
    # to ~1e-9, not the same bits)
    np.testing.assert_allclose(predict_fn(x_test=xt, get="nngp", compute_cov=False), pred_mean, rtol=1e-6, atol=1e-6 * np.abs(pred_mean).max())
    both = predict_fn(x_test=xt, get=("nngp", "ntk"))
    assert G.mean_gate(both.ntk, post.predict(xt, "ntk", False))[0] < 1e-6
    k = kernel_fn(xt, x, "nngp")
    np.testing.assert_allclose(k, o.kernel_fn(xt, x, "nngp", o.make_arch(1)), rtol=1e-11)
    kk = kernel_fn(xt, None, ("nngp", "ntk"))
    np.testing.assert_allclose(kk.ntk, o.kernel_fn(xt, None, "ntk", o.make_arch(1)), rtol=1e-7)
    assert kernel_fn(xt).nngp.shape == (33, 33)
    with pytest.raises(NotImplementedError):
        predict_fn(t=1.0, x_test=xt, get="nngp")
    with pytest.raises(ValueError):
        kernel_fn(xt, x[:, :5], "nngp")


def test_batch_tiles_serially():
    x, _ = synth.synthetic_queries(256, 64, seed=2)
    _, _, kernel_fn = stax.serial(stax.Dense(512), stax.Relu(), stax.Dense(512), stax.Relu(), stax.Dense(1))
    full = kernel_fn(x, None, "nngp")
    tiled = nt.batch(kernel_fn, batch_size=64, device_count=0)(x, None, "nngp")
    np.testing.assert_allclose(tiled, full, rtol=1e-13)


def _toy_encoder():
    tables = [enc.TableEncoder("a", [enc.numerical("k", 0, 10), enc.numerical("v", 0, 100), enc.numerical("w", -5, 5)], 64),
              enc.TableEncoder("b", [enc.numerical("k", 0, 10), enc.categorical("c", 70)], 64)]
    return enc.NNGPEncoder(tables)


def test_estimator_serving_path(tmp_path):
    rng = np.random.default_rng(0)
    lines = []
    for i in range(300):
        up, lo = sorted(rng.uniform(0, 100, 2), reverse=True)
        wu, wl = sorted(rng.uniform(-5, 5, 2), reverse=True)
        card = max(1, int(5000 * (up - lo) / 100 * (wu - wl) / 10))
        if i % 3 == 0:
            lines.append("a,b@v,%.2f,%.2f#w,%.2f,%.2f@c,%d,%d@a,b,k@%d" % (up, lo, wu, wl, i % 70, (i * 7) % 70, card * 3))
        else:
            lines.append("a@v,%.2f,%.2f#w,%.2f,%.2f@@%d" % (up, lo, wu, wl, card))
    (tmp_path / "q.txt").write_text("\n".join(lines) + "\n")
    je = _toy_encoder()
    with contextlib.redirect_stdout(io.StringIO()) as buf:
        est = Estimator("toy", "", str(tmp_path), encoder=je)
        est.load_model()
        serve = [l.rsplit("@", 1)[0] for l in lines[:40]]
        pred_mean, pred_std = est.predict(serve)
    assert "Model construction complete." in buf.getvalue() and "prediction time=" in buf.getvalue()
    assert pred_mean.shape == (40,) and pred_std.shape == (40,)
    post = o.Posterior(est.X_train, est.Y_train, o.make_arch(1), diag_reg=1e-3)
    xt = np.array([je.parse_line_without_card_then_encode(l) for l in serve])
    m_ref, c_ref = post.predict(xt, "nngp", True)
    assert G.mean_gate(pred_mean, m_ref)[0] < 1e-6
    plain = np.array(["@c," not in l for l in serve])
    np.testing.assert_allclose(pred_std[plain], np.sqrt(np.diag(c_ref))[plain], rtol=1e-4)
    # rows with factorised categorical codes carry features ~2^62 (chunk_size 64, as in the reference): kernel
    # entries span 1e5 .. 1e36 and cond(K) is far beyond float32; the float64 refinement (serving mode: explicit
    # inverse + second-order formula) still delivers the oracle's standard deviations
    np.testing.assert_allclose(pred_std[~plain], np.sqrt(np.diag(c_ref))[~plain], rtol=1e-5)
    # the same through the solve path (no explicit inverse)
    with contextlib.redirect_stdout(io.StringIO()):
        est2 = Estimator("toy", "", str(tmp_path), encoder=je, serving=False)
        est2.load_model()
        mean2, std2 = est2.predict(serve)
    assert G.mean_gate(mean2, m_ref)[0] < 1e-6
    np.testing.assert_allclose(std2, np.sqrt(np.diag(c_ref)), rtol=1e-3)


def test_train_cli_on_forest_queries(golden_dir, tmp_path):
    """train.py --kernel_type nngp on the reference's forest queries (config 1: 1000 train / 200 test)."""
    g = np.load(os.path.join(golden_dir, "forest_queries.npz"))
    g = {k: g[k] for k in g.files}  # NpzFile decompresses on every access
    sent = np.iinfo(np.int32).min
    names = "ABCDEFGHIJ"
    per_file = 2000
    for fi, fn in enumerate(g["files"]):
        with open(tmp_path / str(fn), "w") as f:
            for i in range(fi * per_file, (fi + 1) * per_file):
                preds = ["%s,%d,%d" % (names[c], g["bounds"][i, c, 0], g["bounds"][i, c, 1]) for c in range(10)
                         if g["bounds"][i, c, 0] != sent]
                f.write("#".join(preds) + "@%d\n" % g["cards"][i])
    args = train_cli.make_parser().parse_args(["--kernel_type", "nngp", "--query_path", str(tmp_path),
                                               "--max_num_train", "1000", "--max_num_test", "200"])
    args.join_query = False
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        res = train_cli.main(args)
    text = buf.getvalue()
    for needle in ("number of query: 18000", "(1000, 20) (200, 20)", "Kernel construction in", "Mean Square Error:",
                   "Inference time=", "Predict Result Profile of 200 Queries:", "Query attributes:num_table=1"):
        assert needle in text, needle
    gold = np.load(os.path.join(golden_dir, "forest_n1000_m200.npz"))
    assert G.mean_gate(res["pred_mean"], gold["nngp_mean"])[0] < 1e-6
    np.testing.assert_allclose(res["pred_std"], np.sqrt(gold["nngp_var"]), rtol=1e-5)


def test_active_learning_loop(golden_dir):
    """active/ActiveLearner.py control flow on forest fixture rows: each round moves the most uncertain pool queries
    into the training set; selections match an oracle-driven run of the same loop and the validation error drops."""
    from nngp_src_amd.active import ActiveLearner
    g = np.load(os.path.join(golden_dir, "forest_n1000_m200.npz"))
    X, Y = g["X_train"], g["Y_train"]
    Xtr, Ytr, Xpool, Ypool, Xval, Yval = X[:200], Y[:200], X[200:800], Y[200:800], g["X_test"], g["Y_test"]
    _, _, kernel_fn = stax.serial(stax.Dense(512), stax.Relu(), stax.Dense(1))
    learner = ActiveLearner(budget=150, active_iters=2, kernel_type="nngp", biased_sample=False)
    with contextlib.redirect_stdout(io.StringIO()) as buf:
        learner.active_train(kernel_fn, Xtr, Ytr, Xpool, Ypool, Xval, Yval)
    assert buf.getvalue().count("Test MSE Loss:") == 3 and "# Training samples: 500" in buf.getvalue()
    assert learner.history[-1] < learner.history[0]
    # first selection against the oracle's posterior on the same split
    post = o.Posterior(Xtr, Ytr, o.make_arch(1), diag_reg=1e-3)
    m_ref, c_ref = post.predict(Xpool, "nngp", True)
    score = np.sqrt(np.diag(c_ref)) / np.max(m_ref, 0)
    want = set(np.argsort(score)[-150:].tolist())
    learner2 = ActiveLearner(budget=150, active_iters=0, biased_sample=False)
    with contextlib.redirect_stdout(io.StringIO()):
        pf = learner2.train(kernel_fn, Xtr, Ytr)
    got = set(learner2.active_test(pf, Xpool).tolist())
    assert len(got ^ want) <= 2  # identical up to ties at the selection boundary


def test_active_train_driver_with_the_reference_default_draw(golden_dir):
    """active/active_train.py:21-51 end to end on forest fixture rows with the reference's defaults (--biased_sample True): the
    20 / 60 / 20 split, three fits (the later two extend the factor), and the pool draw of every round equal to the restated
    jax.random.choice(PRNGKey(10), ...) on the oracle's posterior of the same training set (identical up to scores that differ
    in the last digits: the device posterior against the float64 oracle)."""
    from nngp_src_amd import active_train, jaxrand
    from nngp_src_amd.active import ActiveLearner
    g = np.load(os.path.join(golden_dir, "forest_n1000_m200.npz"))
    X = np.vstack([g["X_train"], g["X_test"]])
    Y = np.vstack([g["Y_train"].reshape(-1, 1), g["Y_test"].reshape(-1, 1)])
    args = active_train.parse_args(["--budget", "120", "--active_iters", "2"])
    with contextlib.redirect_stdout(io.StringIO()) as buf:
        learner = active_train.main(args, data=(X, Y, None))
    out = buf.getvalue()
    n = X.shape[0]
    assert "number of query: %d" % n in out and out.count("Test MSE Loss:") == 3
    assert "# Initial Training samples: %d" % int(n * 0.2) in out and "Active Iteration 1: Selection 120" in out
    assert "# Training samples: %d" % (int(n * 0.2) + 240) in out
    assert learner.biased_sample is True and learner.history[-1] < learner.history[0] * 1.5
    # the first round's draw against the oracle
    Xtr, Ytr, _, Xpool, Ypool, _, _, _, _ = active_train.split_20_60_20(X, Y)
    post = o.Posterior(np.asarray(Xtr), np.asarray(Ytr), o.make_arch(1), diag_reg=1e-3)
    m_ref, c_ref = post.predict(np.asarray(Xpool), "nngp", True)
    score = np.sqrt(np.diag(c_ref)) / np.max(m_ref, 0)
    want = jaxrand.choice_without_replacement(10, score.shape[0], 120, score / score.sum())
    learner2 = ActiveLearner(budget=120, active_iters=0)
    _, _, kernel_fn = stax.serial(stax.Dense(512), stax.Relu(), stax.Dense(1))
    with contextlib.redirect_stdout(io.StringIO()):
        pf = learner2.train(kernel_fn, np.asarray(Xtr), np.asarray(Ytr))
    got = learner2.active_test(pf, np.asarray(Xpool))
    assert got.shape == (120,) and len(set(got.tolist()) ^ set(want.tolist())) <= 2
    assert np.mean(got == want) > 0.9  # draw ORDER too, wherever the scores agree to the last digits


@pytest.mark.parametrize("n0,b,ncap", [(1500, 700, 2200), (4200, 1000, 6000), (4096, 128, 4224)])
def test_append_rows_matches_full_refit(n0, b, ncap):
    """nngp_model_append: extending the factor (L10 by a blocked solve, L11 by a small Cholesky) gives the same
    float64 alpha and posterior as a full refit on the concatenated training set; twice in a row for the middle case."""
    d = 24
    x, y = synth.synthetic_queries(ncap, d, seed=3)
    xt, _ = synth.synthetic_queries(200, d, seed=4)
    w, bb = [1.0, 1.0], [0.0, 0.0]
    steps = [n0, n0 + b] + ([ncap] if n0 + b < ncap else [])
    inc = GPModel(ncap, d, w, bb, diag_reg=1e-3).fit(x[:n0], y[:n0])
    for prev, cur in zip(steps[:-1], steps[1:]):
        inc.append(x[prev:cur], y[prev:cur])
        info = inc.info()
        assert info["n"] == cur and info["clamped_pivots"] == 0 and info["rel_residual"] < 1e-9 and info["refine_iters"] <= 10, info
        ref = GPModel(cur, d, w, bb, diag_reg=1e-3).fit(x[:cur], y[:cur])
        a_inc, a_ref = inc.alpha().cpu().numpy(), ref.alpha().cpu().numpy()
        assert np.linalg.norm(a_inc - a_ref) <= 1e-7 * np.linalg.norm(a_ref)
        m_inc, v_inc = inc.predict(xt, cov="diag")
        m_ref, v_ref = ref.predict(xt, cov="diag")
        assert np.allclose(m_inc, m_ref, rtol=1e-8, atol=1e-8 * np.abs(m_ref).max())
        # default level 1; the extended factor is a weaker preconditioner (its old part belongs to the previous regulariser)
        assert np.allclose(v_inc, v_ref, rtol=1e-4, atol=1e-9 * np.abs(v_ref).max())
        _, v_inc2 = inc.set_refine(2).predict(xt, cov="diag")
        inc.set_refine(1)
        assert np.allclose(v_inc2, v_ref, rtol=1e-5, atol=1e-9 * np.abs(v_ref).max())
        # ... and DIRECTLY against the float64 oracle fitted on the concatenated set (not only HIP against HIP)
        m_or, c_or = o.Posterior(x[:cur], y[:cur], o.make_arch(1), diag_reg=1e-3).predict(xt, "nngp", True)
        assert G.mean_gate(m_inc, m_or)[0] < 1e-6
        np.testing.assert_allclose(v_inc, np.diag(c_or), rtol=3e-4, atol=1e-9 * np.abs(c_or).max())
        np.testing.assert_allclose(v_inc2, np.diag(c_or), rtol=1e-4, atol=1e-9 * np.abs(c_or).max())
        # the train-train kernel in HBM is the full symmetric matrix of the concatenated set
        k_inc, ld = inc.kernel_buffer()
        k_ref, ld_ref = ref.kernel_buffer()
        assert torch.allclose(k_inc[:cur, :cur], k_ref[:cur, :cur], rtol=1e-13, atol=0.0)  # (row build vs mirrored tiles)
        assert torch.equal(k_inc[:cur, :cur], k_inc[:cur, :cur].T)
        ref.close()
    inc.close()


@pytest.mark.parametrize("get,n,d,reg", [("nngp", 2500, 20, 1e-3), ("ntk", 1300, 16, 1e-3), ("nngp", 2706, 3, 1e-4), ("nngp", 130, 5, 1e-3)])
def test_serving_mode_matches_the_solve_path(get, n, d, reg):
    """nngp_model_prepare_serving: predictions through the explicit float64 inverse against the solve path at level 3
    (diag, full, x_test=None), on a well-conditioned fit of each kernel and on the ill-conditioned NNGP fit of the sweep;
    append drops the inverse."""
    from nngp_src_amd import synth
    from nngp_src_amd.model import GPModel
    x, y = synth.synthetic_queries(n + 40, d, seed=6)
    xt, _ = synth.synthetic_queries(150, d, seed=106)
    arch = ([0.96] * 4, [0.05] * 4) if d == 3 else ([1.0] * 3, [0.0] * 3)
    model = GPModel(n + 40, d, arch[0], arch[1], get=get, diag_reg=reg).fit(x[:n], y[:n])
    model.set_refine(3)
    mean0, var0 = model.predict(xt, cov="diag")
    _, cov0 = model.predict(xt[:40], cov="full")
    mtr0, vtr0 = model.predict(None, cov="diag")
    model.prepare_serving()
    model.set_refine(2)
    mean1, var1 = model.predict(xt, cov="diag")
    assert model.cov_iters() == 0
    _, cov1 = model.predict(xt[:40], cov="full")
    mtr1, vtr1 = model.predict(None, cov="diag")
    for got, want in ((mean1, mean0), (mtr1, mtr0)):
        np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-6 * np.abs(want).max())
    np.testing.assert_allclose(var1, var0, rtol=2e-6)
    assert np.abs(cov1 - cov0).max() <= 2e-6 * np.abs(np.diag(cov0)).max()
    assert np.array_equal(cov1, cov1.T)
    np.testing.assert_allclose(vtr1, vtr0, rtol=1e-4, atol=1e-9 * np.abs(vtr0).max())
    for few in (1, 3, 8, 13):  # a handful of queries: the streaming (skinny) products instead of the 128-row MFMA GEMM
        mean_f, var_f = model.predict(xt[:few], cov="diag")
        np.testing.assert_allclose(mean_f, mean0[:few], rtol=1e-6, atol=1e-6 * np.abs(mean0).max())
        np.testing.assert_allclose(var_f, var0[:few], rtol=2e-6)
    # the inverse belongs to the fit it was built from
    model.append(x[n:], y[n:])
    ref = GPModel(n + 40, d, arch[0], arch[1], get=get, diag_reg=reg).fit(x, y)
    _, var2 = model.predict(xt, cov="diag")
    _, var3 = ref.predict(xt, cov="diag")
    np.testing.assert_allclose(var2, var3, rtol=1e-5, atol=1e-9 * np.abs(var3).max())
    model.close(); ref.close()


def test_c_abi_error_codes_instead_of_crashes():
    """Stages called out of order, NULL outputs, capacities exceeded: every entry point must return a negative code and
    leave a message in nngp_last_error -- never touch the GPU with bad arguments (include/nngp_hip.h: error behaviour)."""
    import ctypes
    from nngp_src_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    arch = _lib.make_arch([1.0, 1.0], [0.0, 0.0])
    h = ctypes.c_void_p()

    def failed(rc, needle):
        msg = (lib.nngp_last_error() or b"").decode()
        return rc < 0 and needle in msg, (rc, msg)

    assert failed(lib.nngp_model_create(ctypes.byref(h), 0, 0, 4, 1, ctypes.byref(arch), _lib.GET_NNGP, 1e-3, 0), "bad sizes")[0]
    assert failed(lib.nngp_model_create(ctypes.byref(h), 300, 0, 4, 1, ctypes.byref(arch), 7, 1e-3, 0), "get must be")[0]
    assert lib.nngp_model_create(ctypes.byref(h), 300, 16, 4, 1, ctypes.byref(arch), _lib.GET_NNGP, 1e-3, 0) == 0
    x = torch.rand((300, 4), dtype=torch.float64, device=dev) * 1000
    y = torch.rand((300, 1), dtype=torch.float64, device=dev)
    out = torch.empty((16, 1), dtype=torch.float64, device=dev)
    var = torch.empty((16,), dtype=torch.float64, device=dev)
    s = _lib.stream_ptr()
    p = _lib.ptr
    for call, needle in [
        (lambda: lib.nngp_model_predict(h, p(x[:16]), 16, 1, p(out), p(var), s), "fit the model first"),
        (lambda: lib.nngp_model_build_rows(h, 0, 10, s), "set_train first"),
        (lambda: lib.nngp_model_factor(h, s), "build the kernel rows first"),
        (lambda: lib.nngp_model_solve(h, 0, 0.0, s), "factor first"),
        (lambda: lib.nngp_model_append(h, p(x[:4]), p(y[:4]), 4, s), "fit the model first"),
        (lambda: lib.nngp_model_prepare_serving(h, s), "fit the model first"),
        (lambda: lib.nngp_model_set_train(h, p(x), p(y), 301, s), "outside"),
        (lambda: lib.nngp_model_set_train(h, None, p(y), 300, s), "NULL"),
        (lambda: lib.nngp_model_set_refine(h, 99), "level"),
    ]:
        ok, info = failed(call(), needle)
        assert ok, info
    assert lib.nngp_model_fit(h, p(x[:280]), p(y[:280]), 280, s) == 0
    for call, needle in [
        (lambda: lib.nngp_model_predict(h, p(x[:16]), 16, 5, p(out), p(var), s), "cov_mode"),
        (lambda: lib.nngp_model_predict(h, p(x[:16]), 16, 1, p(out), None, s), "NULL output"),
        (lambda: lib.nngp_model_predict(h, p(x[:16]), -1, 0, p(out), None, s), "negative"),
        (lambda: lib.nngp_model_build_rows(h, 5, 281, s), "bad row range"),
        (lambda: lib.nngp_model_append(h, p(x[280:]), p(y[280:]), 21, s), "exceed"),
    ]:
        ok, info = failed(call(), needle)
        assert ok, info
    # the handle still works after all that
    assert lib.nngp_model_predict(h, p(x[:16]), 16, 1, p(out), p(var), s) == 0
    torch.cuda.synchronize()
    assert torch.isfinite(out).all() and (var > 0).all()
    assert lib.nngp_model_append(h, p(x[280:]), p(y[280:]), 20, s) == 0 and lib.nngp_model_solve(h, 0, 0.0, s) == 0
    assert lib.nngp_model_predict(h, p(x[:16]), 16, 1, p(out), p(var), s) == 0
    torch.cuda.synchronize()
    assert torch.isfinite(out).all() and (var > 0).all()
    # the int8 product's entry point and the residual diagnostics
    a = torch.rand((128, 64), dtype=torch.float64, device=dev)
    c = torch.empty((128, 128), dtype=torch.float64, device=dev)
    for call, needle in [
        (lambda: lib.nngp_gemm_nt_i8s(p(c), 128, None, 0, p(a), 64, p(a), 64, 100, 128, 64, 1.0, 0.0, 5, 5, 4, s), "multiples of"),
        (lambda: lib.nngp_gemm_nt_i8s(p(c), 128, None, 0, p(a), 64, p(a), 64, 128, 128, 64, 1.0, 0.0, 9, 5, 4, s), "planes per operand"),
        (lambda: lib.nngp_gemm_nt_i8s(p(c), 128, None, 0, None, 64, p(a), 64, 128, 128, 64, 1.0, 0.0, 5, 5, 4, s), "multiples of"),
        (lambda: lib.nngp_model_residual_floor(None, None, None), "NULL model"),
        (lambda: lib.nngp_model_residual_timer(None, 1), "NULL model"),
    ]:
        ok, info = failed(call(), needle)
        assert ok, info
    ratio, off = ctypes.c_double(0.0), ctypes.c_int32(7)
    assert lib.nngp_model_residual_floor(h, ctypes.byref(ratio), ctypes.byref(off)) == 0 and ratio.value == -1.0 and off.value == 0
    lib.nngp_model_destroy(h)


def test_checkpoint_save_and_resume(tmp_path):
    """the reference keeps (X, factor, alpha) in a closure only.  GPModel.save writes what defines the fit
    (X, Y, architecture, regulariser; alpha as the check value), GPModel.load refits on the device -- faster than reading a
    factor back -- and must reproduce alpha, the means and the variances; appended rows are part of the state."""
    n0, b, d = 900, 300, 24
    x, y = synth.synthetic_queries(n0 + b, d, seed=12)
    xt, _ = synth.synthetic_queries(77, d, seed=13)
    model = GPModel(n0 + b, d, [1.2, 1.0, 0.9], [0.1, 0.0, 0.2], get="nngp", diag_reg=2e-3).fit(x[:n0], y[:n0])
    model.append(x[n0:], y[n0:])
    mean, var = model.predict(xt, cov="diag")
    path = str(tmp_path / "fit.npz")
    model.save(path)
    again = GPModel.load(path)
    assert again.n == n0 + b and again.get == "nngp"
    mean2, var2 = again.predict(xt, cov="diag")
    np.testing.assert_allclose(mean2, mean, rtol=1e-8, atol=1e-9 * np.abs(mean).max())
    np.testing.assert_allclose(var2, var, rtol=1e-5)
    # a checkpoint that does not reproduce is refused
    z = dict(np.load(path))
    z["alpha"] = z["alpha"] * 1.001
    np.savez(str(tmp_path / "bad.npz"), **z)
    with pytest.raises(Exception):
        GPModel.load(str(tmp_path / "bad.npz"))


def test_update_timer_reports_the_trailing_update_launches():
    """bench.py's `roofline` is measured live: nngp_model_update_timer puts HIP events around every split-float16 update launch
    of the factorisation.  N = 5120 = 5 block columns, grouped form with 4 columns per group (potrf.hip): block columns 0..2 are
    applied to the rest of their group at once -- rows below the next diagonal block x the group's remaining columns, on or below
    the diagonal (the diagonal blocks themselves are float32 GEMMs) -- so three launches; the group's far update has only the
    last diagonal block to reach, which is a float32 GEMM too.  Their algorithmic work is 2 x entries x (panel width; the first
    panel keeps 64 lead columns on the float32 MFMA).  The numbers must not change the fit."""
    n, d = 5120, 16
    x, y = synth.synthetic_queries(n, d, seed=21)
    model = GPModel(n, d, [1.0, 1.0], [0.0, 0.0], diag_reg=1e-3)
    model.fit(x, y)
    a0 = model.alpha().cpu().numpy()
    model.update_timer(True)
    model.fit(x, y)
    launches, ms, flops = model.update_timer_read()
    assert launches == 3 and ms > 0.0

    def entries(rows, cols, shift):  # (i, j), i < rows, j < cols, j <= i + shift
        return sum(min(cols, i + shift + 1) for i in range(rows))

    want = 0.0
    for k in range(3):
        m = n - 1024 * (k + 1)                  # rows below block column k
        wn = min(m, 1024 * (3 - k))             # columns of group 0 after block column k
        want += 2.0 * entries(m - 1024, wn, 1024) * (1024 - (64 if k == 0 else 0))
    assert flops == pytest.approx(want, rel=1e-12)
    np.testing.assert_array_equal(model.alpha().cpu().numpy(), a0)
    model.update_timer(False)
    model.fit(x, y)
    assert model.update_timer_read()[0] == 0
    # a model too small for the look-ahead factorisation still answers (no launches)
    small = GPModel(512, d, [1.0, 1.0], [0.0, 0.0], diag_reg=1e-3).fit(x[:512], y[:512])
    small.update_timer(True)
    small.fit(x[:512], y[:512])
    assert small.update_timer_read()[0] == 0


@pytest.mark.parametrize("get", ["nngp", "ntk"])
def test_hip_build_and_host_build_of_the_abi_agree(get):
    """One interface, two builds: the call sequence create / set_train / build_rows / factor / solve / alpha / info / predict
    through libnngp_hip.so (device pointers, mixed precision) and through the host build of the same header
    (oracle/libnngp_cpu.so: host pointers, float64 -- the checker) gives the same numbers."""
    from oracle import c_abi
    n, d, m = 1100, 24, 70
    x, y = synth.synthetic_queries(n, d, seed=31)
    xt, _ = synth.synthetic_queries(m, d, seed=32)
    w, b = [1.1, 1.0, 0.9], [0.05, 0.0, 0.1]
    hip = GPModel(n, d, w, b, get=get, diag_reg=1e-3)
    cpu = c_abi.CpuModel(n, d, w, b, get=get, diag_reg=1e-3)
    for model in (hip, cpu):
        model.set_train(x, y)
        model.build_rows(0, n // 2)
        model.build_rows(n // 2, n)
        model.factor()
        model.solve()
    a_hip, a_cpu = hip.alpha().cpu().numpy(), cpu.alpha()
    assert np.linalg.norm(a_hip - a_cpu) <= 1e-7 * np.linalg.norm(a_cpu)
    i_hip, i_cpu = hip.info(), cpu.info()
    assert i_hip["n"] == i_cpu["n"] == n and i_hip["reg"] == pytest.approx(i_cpu["reg"], rel=1e-8)
    mean_h, var_h = hip.predict(xt, cov="diag")
    mean_c, var_c = cpu.predict(xt, "diag")
    assert np.linalg.norm(mean_h - mean_c) <= 1e-7 * np.linalg.norm(mean_c)
    np.testing.assert_allclose(var_h, var_c, rtol=1e-4)
    _, cov_h = hip.predict(xt, cov="full")
    _, cov_c = cpu.predict(xt, "full")
    np.testing.assert_allclose(cov_h, cov_c, rtol=1e-3, atol=1e-6 * np.abs(cov_c).max())


@pytest.mark.parametrize("n,ld", [(1, 8), (127, 128), (128, 160), (1000, 1024), (4133, 4224)])
def test_symmetric_product_reads_the_lower_triangle_only(n, ld):
    """nngp_symv_f64 (the alpha CG's product): y = (A + c I) x from the lower triangle of a symmetric matrix held in a padded
    buffer.  The strict upper triangle and everything beyond row / column n are poisoned with NaN: they must not be read."""
    import torch
    from nngp_src_amd import _lib
    lib = _lib.load()
    g = torch.Generator(device="cpu").manual_seed(n)
    b = torch.randn((n, n), generator=g, dtype=torch.float64)
    a = (b + b.T) / 2
    x = torch.randn((n,), generator=g, dtype=torch.float64)
    want = a @ x + 0.75 * x
    rows = ((n + 127) // 128) * 128
    buf = torch.full((rows, ld), float("nan"), dtype=torch.float64)
    buf[:n, :n] = torch.tril(a) + torch.triu(torch.full((n, n), float("nan"), dtype=torch.float64), diagonal=1)
    dev = _lib.require_gpu()
    buf_d, x_d = buf.to(dev), x.to(dev)
    y_d = torch.full((n,), float("nan"), dtype=torch.float64, device=dev)
    _lib.check(lib.nngp_symv_f64(_lib.ptr(buf_d), ld, n, _lib.ptr(x_d), _lib.ptr(y_d), 0.75, _lib.stream_ptr()))
    torch.cuda.synchronize()
    y = y_d.cpu()
    assert torch.isfinite(y).all()
    assert float((y - want).abs().max()) <= 1e-12 * float(want.abs().max() + 1.0) * max(1.0, n ** 0.5)
    y2 = torch.empty_like(y_d)
    _lib.check(lib.nngp_symv_f64(_lib.ptr(buf_d), ld, n, _lib.ptr(x_d), _lib.ptr(y2), 0.75, _lib.stream_ptr()))
    torch.cuda.synchronize()
    assert torch.equal(y2.cpu(), y)  # fixed summation order


@pytest.mark.parametrize("get,n,d", [("nngp", 6144, 32), ("ntk", 3072, 20)])
def test_fit_and_predict_are_bitwise_reproducible(get, n, d):
    """Two models, same inputs: alpha, means, variances and the full covariance are bit-identical.  Nothing on the path may
    depend on which workgroup took which tile from the work counters, on the order partial sums arrive in (the CG's dot
    products and symmetric product sum fixed slots in a fixed order), or on what an earlier fit left in the workspaces."""
    x, y = synth.synthetic_queries(n, d, seed=31)
    xt, _ = synth.synthetic_queries(700, d, seed=32)
    outs = []
    for rep in range(2):
        model = GPModel(n, d, [1.1, 0.9, 1.0], [0.05, 0.0, 0.1], get=get, diag_reg=1e-3)
        if rep == 1:  # a different fit first: stale workspace contents must not leak into the second one
            model.fit(x[::-1].copy(), y[::-1].copy())
            model.predict(xt[:100], cov="diag")
        model.fit(x, y)
        mean, var = model.predict(xt, cov="diag")
        _, cov = model.predict(xt[:96], cov="full")
        outs.append((model.alpha().cpu().numpy(), np.asarray(mean), np.asarray(var), np.asarray(cov), model.info()["refine_iters"]))
        model.close()
    for a, b in zip(outs[0], outs[1]):
        np.testing.assert_array_equal(a, b)


def test_pool_select_on_the_device_matches_the_host_build(golden_dir):
    """the pool scoring of the active-learning loop runs on the GPU (nngp_pool_select) -- only the selected
    indices travel.  Against the host build of the same entry point on the oracle's posterior (top-k: identical up to ties at
    the boundary; score-proportional draw: the same counter-based keys, so the same draw wherever the scores agree)."""
    from oracle import c_abi
    g = np.load(os.path.join(golden_dir, "forest_n1000_m200.npz"))
    X, Y = g["X_train"], g["Y_train"]
    Xtr, Ytr, Xpool = X[:300], Y[:300], X[300:1000]
    model = GPModel(300, X.shape[1], [1.0, 1.0], [0.0, 0.0], diag_reg=1e-3).fit(Xtr, Ytr)
    m_ref, c_ref = o.Posterior(Xtr, Ytr, o.make_arch(1), diag_reg=1e-3).predict(Xpool, "nngp", True)
    v_ref = np.diag(c_ref)
    for biased in (False, True):
        got = model.select_pool(Xpool, 150, biased=biased, seed=10)
        want = c_abi.pool_select(m_ref, v_ref, 150, biased=biased, seed=10)
        assert got.shape == (150,) and len(set(got.tolist())) == 150
        assert len(set(got.tolist()) ^ set(want.tolist())) <= 2, (biased, len(set(got.tolist()) ^ set(want.tolist())))
    # the device kernels on the oracle's own numbers: exactly the host build's indices, in order
    lib = _lib_mod.load()
    md, vd = _lib_mod.to_device_f64(np.ascontiguousarray(m_ref), G.dev()), _lib_mod.to_device_f64(np.ascontiguousarray(v_ref), G.dev())
    for biased in (0, 1):
        idx = torch.empty((150,), dtype=torch.int64, device=md.device)
        _lib_mod.check(lib.nngp_pool_select(_lib_mod.ptr(md), md.shape[0], 1, _lib_mod.ptr(vd), 150, biased, 10, _lib_mod.ptr(idx),
                                            _lib_mod.stream_ptr()))
        np.testing.assert_array_equal(idx.cpu().numpy(), c_abi.pool_select(m_ref, v_ref, 150, biased=bool(biased), seed=10))
    model.close()


@pytest.mark.parametrize("get,n,mt", [("nngp", 2600, 300), ("ntk", 4200, 520)])
def test_digit_planes_cut_from_the_lower_triangle_are_the_row_cut_bit_for_bit(get, n, mt):
    """Round 4: the digit planes of the kernel matrix (operand of the posterior's int8 residual product; reference op: the
    covariance of predict_fn, train.py:157-158) are cut from the lower triangle of the bitwise symmetric kernel buffer, every entry
    read once (k_i8s_slice_sym: 128 x 128 tiles, direct with the rows' scales and transposed with the columns').  Same digits as the
    row-by-row cut (timing-knob key 5 = 62): every number downstream is bit-identical -- 5 planes (NNGP) and 7 planes + the NNGP
    kernel beside an NTK fit."""
    x, y = synth.synthetic_queries(n, 24, seed=51)
    xt, _ = synth.synthetic_queries(mt, 24, seed=52)
    model = GPModel(n, 24, [1.0, 1.1], [0.0, 0.1], get=get, diag_reg=1e-3, knobs=True)
    outs = []
    for key in (0, 62):
        model.debug_set(5, key)
        model.fit(x, y)
        model.residual_timer(True)
        mean, var = model.predict(xt, cov="diag")
        assert model.residual_timer_read()[0] >= 1   # the int8 path ran
        outs.append((np.asarray(mean), np.asarray(var)))
    model.debug_set(5, 0)
    model.close()
    np.testing.assert_array_equal(outs[0][0], outs[1][0])
    np.testing.assert_array_equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("get", ["nngp", "ntk"])
def test_later_residuals_with_five_rounded_planes_against_seven_and_float64(get):
    """Round 4 (late): the FINE int8 products -- later residuals of the correction sweeps, the NTK's W = Z K_dd (reference op: the
    covariance of predict_fn(..., compute_cov=True), train.py:157-158) -- go on with z rounded to 40 bits below its row maximum
    (five planes, written back: 25 plane products instead of 28).  An iterate good to ~1e-8 does not see 9e-13: variances against
    seven planes (timing-knob key 5 = 64) and against the float64 pipe (key 5 = 50), NNGP at level 3 and NTK at its default."""
    n, mt = 4300, 600   # np >= 4096 and >= 512 padded test rows: the FINE path
    x, y = synth.synthetic_queries(n, 24, seed=61)
    xt, _ = synth.synthetic_queries(mt, 24, seed=62)
    model = GPModel(n, 24, [1.0, 1.0, 1.0], [0.0, 0.0, 0.0], get=get, diag_reg=1e-3, knobs=True).fit(x, y)
    if get == "nngp":
        model.set_refine(3)   # two correction sweeps: the second residual is a FINE product (level 2 has only the first, COARSE one)
    out = {}
    for key in (0, 64, 50):
        model.debug_set(5, key)
        model.residual_timer(True)
        _, var = model.predict(xt, cov="diag")
        launches, _, flops, ops = model.residual_timer_read()
        out[key] = (np.asarray(var), launches, ops / flops if flops else 0.0)
    model.debug_set(5, 0)
    model.close()
    assert out[50][1] == 0 and out[0][1] >= 2 and out[0][1] == out[64][1]      # float64 pipe: no int8 launches
    assert out[0][2] < out[64][2] <= 28.0                                      # fewer plane products per launch on average
    for key in (64, 50):
        # (2e-8: rounding z to 40 bits below its row maximum moves a variance by what the sweeps amplify it to -- 3e-9 with round 4's
        # step-by-step solves, 7.5e-9 with round 5's merged chain updates, whose z0 differs in its last bits; variances are gated at 1e-5)
        assert np.max(np.abs(out[0][0] - out[key][0]) / np.abs(out[key][0])) < 2e-8, key


def test_int8_residual_path_against_the_float64_residual():
    """The level-1 variance with its residual product on the int8 pipe (exact digit planes, gemm_i8s.hip; reference op: the
    covariance of predict_fn(..., compute_cov=True), train.py:157-158) against the same predict with the float64 product
    (timing-knob key 5 = 50) and against the oracle; the live timer says which path ran: M > 128 rows on a fit of N >= 2048 take
    it, a small block of rows and the NTK's later residuals do not."""
    x, y = synth.synthetic_queries(2600, 24, seed=41)
    xt, _ = synth.synthetic_queries(300, 24, seed=42)
    model = GPModel(2600, 24, [1.0, 1.0, 1.0], [0.0, 0.0, 0.0], diag_reg=1e-3, knobs=True).fit(x, y)
    model.residual_timer(True)
    model.trsm_timer(True)
    mean_i8, var_i8 = model.predict(xt, cov="diag")
    solves, sms, sflops = model.trsm_timer_read()   # level 1: forward + backward solve of k, forward solve of the residual
    assert solves == 3 and sflops == 3.0 * 2688 * 2688 * 384 and sms > 0.0 and model.trsm_timer_read()[0] == 0
    model.trsm_timer(False)
    launches, ms, flops, ops = model.residual_timer_read()
    assert launches == 1 and ops == 12 * flops and flops == 2.0 * 384 * 2688 * 2688 and ms > 0.0  # 3 x 5 planes, i + j <= 4: 12 pairs
    ratio, distrusted = model.residual_floor()   # the guard's estimate of what the dropped digit pairs cost these variances
    assert 0.0 < ratio < 1e-5 and not distrusted, ratio
    model.predict(xt[:100], cov="diag")
    assert model.residual_timer_read()[0] == 0   # 128 padded rows: float64 pipe
    model.debug_set(5, 50)
    mean_64, var_64 = model.predict(xt, cov="diag")
    assert model.residual_timer_read()[0] == 0
    model.debug_set(5, 0)
    # the mean passes through Z only in its early-stop correction Z r_alpha, and the int8 path goes on with Z rounded to 24 bits
    assert np.max(np.abs(mean_i8 - mean_64)) <= 1e-10 * np.max(np.abs(mean_64))
    assert np.max(np.abs(var_i8 - var_64) / np.abs(var_64)) < 1e-6
    import c_oracle
    post = c_oracle.fit(x, y, [1.0, 1.0, 1.0], [0.0, 0.0, 0.0], diag_reg=1e-3)
    mean_ref, var_ref = c_oracle.predict_nngp(post, xt, 1)
    assert np.max(np.abs(var_i8 - var_ref[:300]) / np.abs(var_ref[:300])) < 1e-5
    assert np.max(np.abs(mean_i8 - mean_ref) / np.maximum(1.0, np.abs(mean_ref))) < 1e-6
    # round 4: the guard looks at EVERY level-1 batch, not only the first of a fit (a later batch may sit closer to training rows: smaller
    # variances under the same absolute floor).  A later batch's estimate travels to the host without a wait and is read when the NEXT
    # predict starts: the reported ratio is the largest seen, and a batch that trips the threshold (key 5 = 56: threshold 0) sends the
    # fit to the float64 pipe from the next predict on.
    _, var_near = model.predict(x[:300] * (1.0 + 1e-7), cov="diag")
    assert model.residual_timer_read()[0] == 1 and var_near.mean() < 0.5 * var_i8.mean()
    torch.cuda.synchronize()
    model.predict(xt, cov="diag")
    ratio2, distrusted2 = model.residual_floor()
    assert ratio2 >= ratio and ratio2 < 1e-5 and not distrusted2 and model.residual_timer_read()[0] == 1, (ratio, ratio2, distrusted2)
    # (round 5: residual_floor() waits for a pending estimate and folds it in -- the one above is consumed; the next batch leaves a new one)
    model.predict(xt, cov="diag")
    torch.cuda.synchronize()
    assert model.residual_timer_read()[0] == 1
    model.debug_set(5, 56)
    _, var_d = model.predict(xt, cov="diag")       # its start reads the previous batch's estimate against threshold 0
    model.debug_set(5, 0)
    assert model.residual_floor()[1] and model.residual_timer_read()[0] == 0
    np.testing.assert_array_equal(var_d, var_64)
    model.close()
    # no room for the digit planes (timing-knob key 5 = 55: every workspace allocation of the int8 path fails): the model stays on
    # the float64 pipe, with the float64 path's results
    full = GPModel(2600, 24, [1.0, 1.0, 1.0], [0.0, 0.0, 0.0], diag_reg=1e-3, knobs=True).fit(x, y)
    full.debug_set(5, 55)
    full.residual_timer(True)
    mean_f, var_f = full.predict(xt, cov="diag")
    full.debug_set(5, 0)
    assert full.residual_timer_read()[0] == 0
    np.testing.assert_array_equal(var_f, var_64)
    _, var_f2 = full.predict(xt, cov="diag")   # ... and does not try again
    assert full.residual_timer_read()[0] == 0
    np.testing.assert_array_equal(var_f2, var_64)
    full.close()
    # the guard fires (key 5 = 56: threshold 0): the first predict is redone on the float64 pipe and the fit stays there
    wary = GPModel(2600, 24, [1.0, 1.0, 1.0], [0.0, 0.0, 0.0], diag_reg=1e-3, knobs=True).fit(x, y)
    wary.debug_set(5, 56)
    wary.residual_timer(True)
    _, var_w = wary.predict(xt, cov="diag")
    wary.debug_set(5, 0)
    assert wary.residual_timer_read()[0] == 1 and wary.residual_floor()[1]
    np.testing.assert_array_equal(var_w, var_64)
    _, var_w2 = wary.predict(xt, cov="diag")
    assert wary.residual_timer_read()[0] == 0
    np.testing.assert_array_equal(var_w2, var_64)
    wary.fit(x, y)                                  # a new fit is trusted again until measured
    wary.predict(xt, cov="diag")
    assert wary.residual_timer_read()[0] == 1 and not wary.residual_floor()[1]
    wary.close()
    # NTK: the first correction sweep's residual only (one launch per predict), the later residual and W = Z K_dd on the float64 pipe
    ntk = GPModel(2600, 24, [1.0, 1.0], [0.0, 0.0], get="ntk", diag_reg=1e-3, knobs=True).fit(x, y)
    ntk.residual_timer(True)
    _, v_i8 = ntk.predict(xt, cov="diag")
    assert ntk.residual_timer_read()[0] == 1
    ntk.debug_set(5, 50)
    _, v_64 = ntk.predict(xt, cov="diag")
    ntk.debug_set(5, 0)
    assert np.max(np.abs(v_i8 - v_64) / np.abs(v_64)) < 1e-7
    ntk.close()


def test_row_sharded_layout_refuses_what_it_cannot_serve():
    """Round-4 advice: after nngp_model_factor_input_rows the model's float64 kernel holds this rank's rows only (shard32.py; the
    nt.batch device slot of train.py:166-168) -- a covariance, a replicated CG, an appended fit or a serving inverse computed from it
    would be silently wrong.  They must come back as error codes; the mean of test rows (alpha installed by the caller) still works."""
    import ctypes
    from nngp_src_amd import _lib
    lib = _lib.load()
    n, d = 2304, 16
    x, y = synth.synthetic_queries(n, d, seed=3)
    xt, _ = synth.synthetic_queries(256, d, seed=4)
    model = GPModel(n + 128, d, [1.0, 1.0], [0.0, 0.0], diag_reg=1e-3)
    h, s, p = model.handle, _lib.stream_ptr(), _lib.ptr

    def failed(rc, needle):
        msg = (lib.nngp_last_error() or b"").decode()
        return rc < 0 and needle in msg, (rc, msg)

    model.set_train(x, y)
    model.build_rows(0, n)
    ok, info = failed(lib.nngp_model_factor_input_complete(h), "no rows of the factor input")
    assert ok, info
    assert lib.nngp_model_factor_input_rows(h, 0, n, 1.0, s) == 0
    assert lib.nngp_model_factor_input_complete(h) == 0
    model.factor()
    ok, info = failed(lib.nngp_model_solve(h, 0, 0.0, s), "row-sharded")
    assert ok, info
    # alpha from the single-GPU fit of the same data stands in for the caller's sharded CG
    ref = GPModel(n, d, [1.0, 1.0], [0.0, 0.0], diag_reg=1e-3).fit(x, y)
    alpha = ref.alpha().contiguous()
    mean_ref = ref.predict(xt, cov=None)
    assert lib.nngp_model_set_alpha(h, p(alpha), 5, 1e-11, s) == 0
    xd = _lib.to_device_f64(xt)
    mean = torch.empty((256, 1), dtype=torch.float64, device=G.dev())
    var = torch.empty((256,), dtype=torch.float64, device=G.dev())
    for mode in (_lib.COV_DIAG, _lib.COV_FULL):
        ok, info = failed(lib.nngp_model_predict(h, p(xd), 256, mode, p(mean), p(var), s), "row-sharded")
        assert ok, info
    ok, info = failed(lib.nngp_model_predict(h, None, 0, _lib.COV_NONE, p(mean), None, s), "row-sharded")   # x_test=None reads K_dd as the cross kernel
    assert ok, info
    ok, info = failed(lib.nngp_model_prepare_serving(h, s), "row-sharded")
    assert ok, info
    xn = _lib.to_device_f64(x[:64]); yn = _lib.to_device_f64(y[:64])
    ok, info = failed(lib.nngp_model_append(h, p(xn), p(yn), 64, s), "row-sharded")
    assert ok, info
    assert lib.nngp_model_predict(h, p(xd), 256, _lib.COV_NONE, p(mean), None, s) == 0
    assert G.mean_gate(mean.cpu().numpy(), mean_ref)[0] < 1e-9
    model.close(); ref.close()


@pytest.mark.parametrize("n,rows", [(2500, 300), (9300, 384)])
def test_every_entry_point_on_a_fresh_fit_in_an_unusual_order(n, rows):
    """Round 4 lost a GPU session to a null device pointer: nngp_model_apply_factor(both halves) reached the float32 solve path before
    anything had built L^T.  The operand is now built by the function that reads it (apply_inverse_f32), so no order of calls can
    reach it unbuilt.  A freshly created and fitted model takes every public entry point here in an order no other test uses --
    both halves of the factor FIRST -- and must answer with results or error codes, never with a fault (include/nngp_hip.h: error
    behaviour).  n = 2500: the float32 solve path; n = 9300, 384 rows: the persistent float16-pipe solves."""
    import ctypes
    import scipy.linalg as sla
    from nngp_src_amd import _lib
    lib = _lib.load()
    d = 12
    x, y = synth.synthetic_queries(n, d, seed=21)
    xt, _ = synth.synthetic_queries(rows, d, seed=22)
    model = GPModel(n + 256, d, [1.0, 1.0], [0.0, 0.0], diag_reg=1e-3).fit(x, y)
    h, s, p = model.handle, _lib.stream_ptr(), _lib.ptr
    rng = np.random.default_rng(5)
    B = rng.standard_normal((rows, n)).astype(np.float32)
    Z = model.apply_factor(torch.from_numpy(B.copy()).to(G.dev()), both_halves=True).cpu().numpy().astype(np.float64)   # first call on the handle
    a32, _ = model.factor_buffers()
    L = torch.tril(a32[:n, :n]).double().cpu().numpy()
    Zref = sla.cho_solve((L, True), B.astype(np.float64).T).T
    assert np.linalg.norm(Z - Zref) <= 5e-3 * np.linalg.norm(Zref)
    r = torch.from_numpy(rng.standard_normal(n)).to(G.dev())
    z = torch.empty_like(r)
    assert lib.nngp_model_precond(h, p(r), p(z), s) == 0
    q = torch.empty((100,), dtype=torch.float64, device=G.dev())
    assert lib.nngp_model_matvec_rows(h, p(r), p(q), 50, 150, s) == 0
    model.set_refine(2)
    mean2, var2 = model.predict(xt, cov="diag")
    model.set_refine(0)
    mean0, var0 = model.predict(xt, cov="diag")
    model.set_refine(1)
    mean1, cov1 = model.predict(xt, cov="full")
    assert np.allclose(mean0, mean2, rtol=0, atol=1e-8 * np.abs(mean2).max()) and np.allclose(np.diag(cov1), var2, rtol=1e-4)
    assert np.allclose(var0, var2, rtol=5e-2)
    model.prepare_serving()
    mean_s, var_s = model.predict(xt, cov="diag")
    assert np.allclose(var_s, var2, rtol=1e-4)
    X1 = model.apply_factor(torch.from_numpy(B.copy()).to(G.dev())).cpu().numpy().astype(np.float64)
    assert np.isfinite(X1).all()
    xa, ya = synth.synthetic_queries(200, d, seed=23)
    model.append(xa, ya)
    mean_a, var_a = model.predict(xt, cov="diag")
    assert np.isfinite(mean_a).all() and (var_a > 0).all()
    Z2 = model.apply_factor(torch.from_numpy(np.ascontiguousarray(rng.standard_normal((rows, n + 200)).astype(np.float32))).to(G.dev()), both_halves=True)
    assert torch.isfinite(Z2).all()
    assert model.info()["clamped_pivots"] == 0
    model.close()


def test_predicts_on_a_reserved_model_allocate_nothing():
    """SURVEY.md 8b ownership rule (workspace is allocated ahead, never inside the timed launch functions): a model created with
    m_cap -- GPModel then calls nngp_model_reserve(m_cap, diag) -- makes no device allocation in its first predict after a fit, nor
    in its tenth, nor after a refit; nngp_alloc_count is the library's own count of its hipMalloc calls."""
    from nngp_src_amd import _lib
    lib = _lib.load()
    n, d, mrows = 4352, 16, 512
    x, y = synth.synthetic_queries(n, d, seed=8)
    xt, _ = synth.synthetic_queries(mrows, d, seed=9)
    model = GPModel(n, d, [1.0, 1.0, 1.0], [0.0, 0.0, 0.0], diag_reg=1e-3, m_cap=mrows)
    xd = _lib.to_device_f64(xt)
    model.fit(x, y)
    torch.cuda.synchronize()
    c0 = lib.nngp_alloc_count()
    assert c0 > 0
    first = model.predict(xd, cov="diag")
    assert lib.nngp_alloc_count() == c0, "the first predict after a fit allocated device memory"
    for _ in range(9):
        last = model.predict(xd, cov="diag")
    assert lib.nngp_alloc_count() == c0, "a later predict allocated device memory"
    assert np.array_equal(first[0], last[0]) and np.array_equal(first[1], last[1])
    model.fit(x, y)
    model.predict(xd, cov="diag")
    model.predict(xd[:300], cov="diag")
    assert lib.nngp_alloc_count() == c0, "a refit or a smaller batch allocated device memory"
    model.close()
