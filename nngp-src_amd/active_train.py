"""``python -m nngp_src_amd.active_train --budget 1000 --active_iters 3 ...`` -- drop-in for the reference's active-learning driver
(active/active_train.py:21-51 ``main``, flags :54-107).

Same flow and prints: load the encoded queries, 20 / 60 / 20 split with seed 10 (train / pool / validation:
``train_test_val_split(X, Y, train_frac=0.2, test_frac=0.6)``, active_train.py:26-27), ``Dense(512)-Relu-Dense(1)``, then
``ActiveLearner.active_train`` -- fit, score the pool by predictive std / max(mean), move ``--budget`` queries into the training set
(default: the std-proportional draw of ``--biased_sample True``, the reference's ``jax.random.choice`` restated in ``jaxrand.py``),
refit, ``--active_iters`` times; the GP runs on the MI355X through libnngp_hip.so and every refit after the first EXTENDS the
factor (``nngp_model_append``).

Data: the reference reads ``schemas.load_training_schema_data(args)`` (benchmark CSVs of yelp / tpcds / tpch / imdb, not shipped);
a registered schema (``schemas.register_schema``) is loaded the same way, otherwise the single-table loader of ``train.py`` -- the
alternative the reference keeps commented out at active_train.py:22 -- serves ``--relations forest``.
"""
from __future__ import annotations

from argparse import ArgumentDefaultsHelpFormatter, ArgumentParser

import numpy as np

from . import schemas
from .active import ActiveLearner
from .train import build_kernel_fn, load_training_data
from .util import train_test_val_split


def split_20_60_20(X, Y, all_query_infos=None):
    """active_train.py:26-27: train 20 %, pool ("test") 60 %, validation 20 % of the seed-10 shuffle."""
    return train_test_val_split(X, Y, train_frac=0.2, test_frac=0.6, all_query_infos=all_query_infos)


def main(args, data=None):
    if data is not None:
        X, Y, all_query_infos = data
    elif getattr(args, "schema_name", None) in schemas.SCHEMAS:
        X, Y, _ = schemas.load_training_schema_data(args.schema_name, args.data_path, args.query_path, args.chunk_size)
        all_query_infos = None
    else:
        X, Y, all_query_infos = load_training_data(args)
    num_queries = X.shape[0]
    print("number of query: {}".format(num_queries))
    X_train, Y_train, query_infos_train, X_test, Y_test, query_infos_test, X_val, Y_val, query_infos_val = \
        split_20_60_20(X, Y, all_query_infos)
    X_train, Y_train = np.asarray(X_train), np.asarray(Y_train)
    X_test, Y_test = np.asarray(X_test), np.asarray(Y_test)
    X_val = np.asarray(X_val) if X_val is not None else None
    Y_val = np.asarray(Y_val) if Y_val is not None else None
    print(X_train.shape, X_test.shape)
    print(Y_train.shape, Y_test.shape)
    init_fn, apply_fn, kernel_fn = build_kernel_fn(getattr(args, "n_relu", 1))
    active_learner = ActiveLearner(args)
    active_learner.active_train(kernel_fn, X_train, Y_train, X_test, Y_test, X_val, Y_val, query_infos_val)
    return active_learner


def _ref_bool(text):
    """The reference declares ``--biased_sample`` with ``type=bool`` (active_train.py:62): ANY non-empty value, 'False' included,
    parses as True.  Kept (a drop-in keeps the flag's behaviour); ``--top_k`` below is the way to the deterministic selection."""
    return bool(text)


def make_parser():
    parser = ArgumentParser("NNGP estimator", formatter_class=ArgumentDefaultsHelpFormatter, conflict_handler="resolve")
    parser.add_argument('--kernel_type', type=str, default="nngp", help='nngp, ntk')
    parser.add_argument("--chunk_size", default=10, type=int, help="dimension of factorized encoding")
    parser.add_argument("--feat_encode", type=str, default='dnn-encoder', help='dnn-encoder,one-hot')
    parser.add_argument('--no-cuda', action='store_true', default=True, help='kept for flag parity; ignored')
    parser.add_argument("--biased_sample", default=True, type=_ref_bool, help="Enable Biased sampling for test set selection")
    parser.add_argument('--active_iters', type=int, default=3, help='Num of iterators of active learning.')
    parser.add_argument('--budget', type=int, default=1000, help='Selected Queries budget Per Iteration.')
    parser.add_argument("--relations", type=str, default='forest')
    parser.add_argument("--names", type=str, default='forest')
    parser.add_argument("--query_path", type=str, default='Queries/forest_data')
    parser.add_argument("--data_path", type=str, default='')
    parser.add_argument("--schema_name", type=str, default='tpch', help='yelp, tpcds, tpch')
    # additions
    parser.add_argument("--top_k", action='store_true', help="select the `budget` largest scores instead of the biased draw")
    parser.add_argument("--n_relu", type=int, default=1, help="hidden ReLU layers (reference: 1)")
    parser.add_argument("--max_num_train", type=int, default=None)
    return parser


def parse_args(argv=None):
    args = make_parser().parse_args(argv)
    if args.top_k:
        args.biased_sample = False
    args.cuda = True
    args.join_query = len(args.relations.split(',')) > 1
    return args


if __name__ == "__main__":
    args = parse_args()
    print(args)
    main(args)
