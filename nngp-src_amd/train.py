"""``python -m nngp_src_amd.train --kernel_type nngp ...`` -- drop-in for the reference driver
(train.py:153-203 ``NNGP_train_and_test``, train.py:224-246 ``main``, flags train.py:252-287).

Prints the same lines as the reference (number of query, shapes, "Kernel construction in ... seconds.",
"Mean Square Error: ...", "Inference time=... seconds", then the q-error profile) with the GP running on
the MI355X through libnngp_hip.so.
"""
from __future__ import annotations

import datetime
import os
from argparse import ArgumentDefaultsHelpFormatter, ArgumentParser

import numpy as np

from . import stax, predict as nt_predict
from .batching import batch
from .encoder import FOREST_COLUMNS, GeneralQueryEncoder, TableEncoder
from .util import PredictionStatistics, train_test_val_split

pred_stat = PredictionStatistics()


def build_kernel_fn(n_relu: int = 1):
    layers = [stax.Dense(512)]
    for _ in range(n_relu):
        layers += [stax.Relu(), stax.Dense(512)]
    layers[-1] = stax.Dense(1)
    return stax.serial(*layers)


def NNGP_train_and_test(args, X_train, Y_train, X_test, Y_test, query_infos_train=None, query_infos_test=None):
    def prediction(pred_fn, X_test, kernel_type="nngp", compute_cov=True):
        pred_mean, pred_cov = pred_fn(x_test=X_test, get=kernel_type, compute_cov=compute_cov)
        return pred_mean, pred_cov

    init_fn, apply_fn, kernel_fn = build_kernel_fn(getattr(args, "n_relu", 1))
    kernel_fn = batch(kernel_fn, device_count=0, batch_size=0)
    start = datetime.datetime.now()
    predict_fn = nt_predict.gradient_descent_mse_ensemble(kernel_fn, X_train, Y_train, diag_reg=1e-3)
    duration = (datetime.datetime.now() - start).total_seconds()
    print('Kernel construction in %s seconds.' % duration)

    cov_mode = True if getattr(args, "full_cov", False) else "diag"
    pred_mean, pred_cov = prediction(predict_fn, X_test, kernel_type=args.kernel_type, compute_cov=cov_mode)
    pred_std = np.sqrt(np.diag(pred_cov)) if cov_mode is True else np.sqrt(pred_cov)

    mse = np.sum(np.power(pred_mean - Y_test, 2))
    print("Mean Square Error: {}".format(mse))

    print(X_test.shape, Y_test.shape)
    start = datetime.datetime.now()
    pred_mean, pred_cov = prediction(predict_fn, X_test, kernel_type=args.kernel_type, compute_cov=cov_mode)
    duration = (datetime.datetime.now() - start).total_seconds()
    print("Inference time={} seconds".format(duration))

    errors = np.ravel(np.array(pred_mean - Y_test))
    pred_stat.get_prediction_details(errors, query_infos_test, partition_keys='num_table')
    return {"pred_mean": np.ravel(pred_mean), "pred_std": np.ravel(pred_std), "errors": errors, "mse": float(mse),
            "fit_info": predict_fn.model_for(args.kernel_type).info()}


def load_training_data(args):
    """datasets.load_training_data for the single-table case (datasets.py:301-346)."""
    relation = args.relations.split(',')[0].strip()
    if relation != 'forest':
        raise NotImplementedError("only the forest relation ships with column metadata; others need their CSV")
    csv = os.path.join(args.data_path or "", "forest.csv")
    if args.data_path and os.path.exists(csv):
        import pandas as pd
        names = [c.name for c in FOREST_COLUMNS]
        df = pd.read_csv(csv, header=None, usecols=list(range(10)), names=names)
        cols = TableEncoder.from_dataframe(df, ['numerical'] * 10, args.names, args.chunk_size).columns
        loader = GeneralQueryEncoder(cols, args.names, args.chunk_size)
    else:
        loader = GeneralQueryEncoder(FOREST_COLUMNS, args.names, args.chunk_size)
    print("feature dim={}".format(loader.total_feat_dim))
    all_queries, all_cards, all_query_infos = loader.load_queries(args.query_path)
    X, Y = loader.transform_to_arrays(all_queries, all_cards)
    return X, Y, all_query_infos


def main(args):
    if args.join_query:
        raise NotImplementedError("join schemas need their benchmark CSVs; use Estimator(encoder=...)")
    X, Y, all_query_infos = load_training_data(args)
    print("number of query: {}".format(X.shape[0]))
    X_train, Y_train, qi_train, X_test, Y_test, qi_test, _, _, _ = train_test_val_split(
        X, Y, train_frac=0.6, test_frac=0.2, all_query_infos=all_query_infos, max_num_train=args.max_num_train)
    if args.max_num_test is not None:
        X_test, Y_test, qi_test = X_test[:args.max_num_test], Y_test[:args.max_num_test], qi_test[:args.max_num_test]
    print(X_train.shape, X_test.shape)
    print(Y_train.shape, Y_test.shape)
    if args.kernel_type == 'gp':
        raise NotImplementedError("--kernel_type gp is broken in the reference (train.py:114 uses an undefined jit)")
    return NNGP_train_and_test(args, X_train, Y_train, X_test, Y_test, qi_train, qi_test)


def make_parser():
    parser = ArgumentParser("NNGP/NTK estimator", formatter_class=ArgumentDefaultsHelpFormatter, conflict_handler="resolve")
    parser.add_argument("--chunk_size", default=64, type=int, help="dimension of factorized encoding")
    parser.add_argument("--kernel_type", type=str, default='nngp', help='nngp, ntk')
    parser.add_argument("--feat_encode", type=str, default='dnn-encoder', help='dnn-encoder,one-hot')
    parser.add_argument('--no-cuda', action='store_true', default=True, help='kept for flag parity; ignored')
    parser.add_argument("--relations", type=str, default='forest')
    parser.add_argument("--names", type=str, default='forest')
    parser.add_argument("--query_path", type=str, default='Queries/forest_data')
    parser.add_argument("--data_path", type=str, default='')
    parser.add_argument("--schema_name", type=str, default='imdb_simple', help='yelp, tpcds, tpch')
    # additions (config 1 of BASELINE.json is not reachable from the reference CLI)
    parser.add_argument("--max_num_train", type=int, default=None)
    parser.add_argument("--max_num_test", type=int, default=None)
    parser.add_argument("--n_relu", type=int, default=1, help="hidden ReLU layers (reference: 1)")
    parser.add_argument("--full_cov", action='store_true', help="form the full M x M covariance like the reference")
    return parser


if __name__ == "__main__":
    args = make_parser().parse_args()
    args.cuda = True
    args.join_query = len(args.relations.split(',')) > 1
    print(args)
    main(args)
