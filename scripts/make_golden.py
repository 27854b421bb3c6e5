#!/usr/bin/env python3
"""Generate tests/golden/* (run in the build container only; /root/reference is read, never copied).

Fixtures are DATA: encoded inputs and expected outputs.
  forest_queries.npz      the 18 000 forest range queries of the reference (Queries/forest_data/*.txt) as
                          int32 bounds + int64 cardinalities, in the reference's load order
  forest_n1000_m200.npz   config 1: seed-10 split, first 1000 train / 200 test rows, encoded X/Y plus the
                          float64 oracle's posterior (nngp + ntk)
  forest_n256_m64.npz     a small slice of the same for quick GPU parity tests
  encoder_ref.json        encodings produced by the REFERENCE encoder (neuroestimator/estimator/encoder.py,
                          imported by file path) for forest lines and a toy 3-table join schema: pins
                          nngp-src_amd/encoder.py against the reference itself
  split_pin.json          the seed-10 permutation pin (util.py:271-293)
"""
import hashlib
import importlib.util
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")

import nngp_oracle as oracle  # noqa: E402
from nngp_src_amd import encoder as enc  # noqa: E402
from nngp_src_amd import util  # noqa: E402


def load_reference_encoder():
    spec = importlib.util.spec_from_file_location("ref_encoder", os.path.join(REF, "neuroestimator/estimator/encoder.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    import pandas as pd
    os.makedirs(OUT, exist_ok=True)
    qdir = os.path.join(REF, "Queries", "forest_data")
    loader = enc.GeneralQueryEncoder(enc.FOREST_COLUMNS, "forest", 64)
    queries, cards, infos = loader.load_queries(qdir, verbose=False)
    X, Y = loader.transform_to_arrays(queries, cards)
    assert X.shape == (18000, 20)

    # ---- compact copy of the query data ----
    SENT = np.iinfo(np.int32).min
    bounds = np.full((len(queries), 10, 2), SENT, dtype=np.int32)
    for i, q in enumerate(queries):
        for (c, up, lo) in q:
            assert float(int(up)) == up and float(int(lo)) == lo
            bounds[i, c] = (int(up), int(lo))
    np.savez_compressed(os.path.join(OUT, "forest_queries.npz"), bounds=bounds, cards=np.asarray(cards, dtype=np.int64),
                        col_lo=np.array([c.lo for c in enc.FOREST_COLUMNS], dtype=np.float64),
                        col_hi=np.array([c.hi for c in enc.FOREST_COLUMNS], dtype=np.float64),
                        files=np.array(sorted(os.listdir(qdir))))

    # ---- reference encoder cross-check + golden encodings ----
    ref = load_reference_encoder()
    names = [c.name for c in enc.FOREST_COLUMNS]
    df = pd.DataFrame({c.name: [float(c.lo), float(c.hi)] for c in enc.FOREST_COLUMNS})
    ref_table = ref.Table(df, ["numerical"] * 10, "forest", 64)
    lines = []
    for fn in sorted(os.listdir(qdir)):
        with open(os.path.join(qdir, fn)) as f:
            lines += [next(f).strip() for _ in range(3)]
    forest_gold = []
    for line in lines:
        pl = ref_table.parse_predicates(line.split("@")[0].strip())
        v = ref_table.predicate_encoding(pl)
        mine = loader.transform_to_1d_array(loader.parse_line(line)[0])
        assert np.array_equal(v, mine), line
        forest_gold.append({"line": line, "x": v.tolist()})
    # all 18000 through the reference encoder as well
    for i in range(0, 18000, 97):
        pl = [(c, float(u), float(l)) for (c, u, l) in queries[i]]
        assert np.array_equal(ref_table.predicate_encoding(pl), X[i])

    # toy join schema: 3 tables, numerical + categorical columns, shared keys
    t_specs = {
        "orders": (["o_id", "o_cust", "o_total", "o_status"], ["numerical", "numerical", "numerical", "categorical"]),
        "cust": (["o_cust", "c_age", "c_region"], ["numerical", "numerical", "categorical"]),
        "items": (["o_id", "i_price", "i_qty"], ["numerical", "numerical", "numerical"]),
    }
    rng = np.random.default_rng(7)
    dfs, ref_tables, my_tables = {}, [], []
    for tname, (cols, kinds) in t_specs.items():
        data = {}
        for c, k in zip(cols, kinds):
            data[c] = rng.integers(0, 150, 400) if k == "categorical" else np.round(rng.uniform(-50, 500, 400), 1)
        dfs[tname] = pd.DataFrame(data)
        ref_tables.append(ref.Table(dfs[tname].copy(), kinds, tname, 64))
        my_tables.append(enc.TableEncoder.from_dataframe(dfs[tname].copy(), kinds, tname, 64))
    ref_enc = ref.NNGPEncoder(ref_tables)
    my_enc = enc.NNGPEncoder(my_tables)
    join_lines = [
        "orders,cust@o_total,300.5,20#o_status,3,77,129@c_age,60,18@orders,cust,o_cust",
        "orders@o_id,100,5@",
        "orders,items,cust@@i_price,99.5,1.5#i_qty,10,2@c_region,0,20,31@orders,items,o_id#orders,cust,o_cust",
        "cust@@",
        "items,orders@i_qty,400,-10@o_status,1@items,orders,o_id",
    ]
    join_gold = []
    for line in join_lines:
        v = ref_enc.parse_line_without_card_then_encode(line)
        mine = my_enc.parse_line_without_card_then_encode(line)
        assert v.shape == mine.shape and np.array_equal(v, mine), (line, v, mine)
        tids, pl, ji, card = ref_enc.parse_line(line + "@12345")
        assert card == 12345
        join_gold.append({"line": line, "x": v.tolist()})
    tables_meta = [{"name": t.table_name, "columns": [list(map(lambda z: z if isinstance(z, str) else float(z), c)) for c in t.columns]}
                   for t in my_tables]
    with open(os.path.join(OUT, "encoder_ref.json"), "w") as f:
        json.dump({"source": "neuroestimator/estimator/encoder.py (reference, imported by file path)",
                   "forest": forest_gold, "join_tables": tables_meta, "join": join_gold,
                   "join_feat_dim": int(ref_enc.join_feat_dim)}, f)

    # ---- split pin ----
    idx = np.asarray(util.split_indices(18000, 10), dtype=np.int64)
    with open(os.path.join(OUT, "split_pin.json"), "w") as f:
        json.dump({"n": 18000, "seed": 10, "first": idx[:8].tolist(), "sha256": hashlib.sha256(idx.tobytes()).hexdigest()}, f)

    # ---- config-1 fixtures with oracle outputs ----
    Xtr, Ytr, qtr, Xte, Yte, qte, _, _, _ = util.train_test_val_split(X, Y, 0.6, 0.2, all_query_infos=infos)
    arch = oracle.make_arch(1)
    for tag, n, m in (("forest_n1000_m200", 1000, 200), ("forest_n256_m64", 256, 64)):
        xa, ya, xb, yb = Xtr[:n], Ytr[:n], Xte[:m], Yte[:m]
        post = oracle.Posterior(xa, ya, arch, diag_reg=1e-3)
        mean_n, cov_n = post.predict(xb, "nngp", True)
        mean_t, cov_t = post.predict(xb, "ntk", True)
        np.savez_compressed(os.path.join(OUT, tag + ".npz"), X_train=xa, Y_train=ya, X_test=xb, Y_test=yb,
                            num_predicates_test=np.array([q.num_predicates for q in qte[:m]], dtype=np.int32),
                            nngp_mean=mean_n, nngp_var=np.diag(cov_n).copy(), nngp_cov16=cov_n[:16, :16].copy(),
                            ntk_mean=mean_t, ntk_var=np.diag(cov_t).copy(),
                            k_dd_corner=oracle.kernel_fn(xa[:8], None, "nngp", arch),
                            t_dd_corner=oracle.kernel_fn(xa[:8], None, "ntk", arch))
        prof = util.q_error_profile((mean_n - yb).ravel())
        print(tag, "oracle q-error median %.4f mean %.3f" % (prof["median"], prof["mean"]))
    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    main()
