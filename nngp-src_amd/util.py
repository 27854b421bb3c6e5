"""Host utilities on the hot path's edges: the deterministic split and the q-error profile printer.

Restates reference ``util.py:271-293`` (``train_test_val_split``) and ``util.py:111-167``
(``PredictionStatistics``) so that the same queries are evaluated and the printed profile diffs
cleanly against the reference's stdout.
"""
from __future__ import annotations

import collections
import random

import numpy as np

QueryInfo = collections.namedtuple("QueryInfo", ["num_table", "num_joins", "num_predicates", "is_equal_join",
                                                 "is_multi_key"])


def split_indices(num_instances: int, seed: int = 10):
    """The permutation the reference applies before slicing (util.py:276-278): CPython's
    ``random.seed(seed); random.shuffle(list(range(n)))``."""
    indices = list(range(num_instances))
    random.seed(seed)
    random.shuffle(indices)
    return indices


def train_test_val_split(X, Y, train_frac=0.6, test_frac=0.2, seed=10, all_query_infos=None, max_num_train=None):
    """util.py:271-293 -- 60/20/20 contiguous slices of the seed-10 shuffle; optional train truncation."""
    num_instances = X.shape[0]
    print("# instances = {}".format(num_instances))
    num_train, num_test = int(train_frac * num_instances), int(test_frac * num_instances)
    indices = split_indices(num_instances, seed)
    X, Y = X[indices, :], Y[indices, :]
    if all_query_infos is not None:
        all_query_infos = [all_query_infos[idx] for idx in indices]
    X_train, Y_train = X[:num_train, :], Y[:num_train, :]
    X_test, Y_test = X[num_train: num_train + num_test, :], Y[num_train: num_train + num_test, :]
    has_val = train_frac + test_frac < 1
    X_val = X[num_train + num_test:, :] if has_val else None
    Y_val = Y[num_train + num_test:, :] if has_val else None
    qi_train = all_query_infos[:num_train] if all_query_infos is not None else None
    qi_test = all_query_infos[num_train: num_train + num_test] if all_query_infos is not None else None
    qi_val = all_query_infos[num_train + num_test:] if all_query_infos is not None and has_val else None
    if max_num_train is not None and max_num_train <= num_train:
        qi_train = qi_train[:max_num_train] if qi_train is not None else None
        X_train = X_train[:max_num_train]
        Y_train = Y_train[:max_num_train]
    return X_train, Y_train, qi_train, X_test, Y_test, qi_test, X_val, Y_val, qi_val


class PredictionStatistics(object):
    """q-error profile of log2-space errors, partitioned by QueryInfo keys (util.py:107-167)."""

    def __init__(self):
        self.keys = ['num_table', 'num_joins', 'num_predicates']

    def get_prediction_details(self, errors, query_infos=None, partition_keys=''):
        if query_infos is None or not partition_keys:
            self.get_prediction_statistics(errors)
            return
        partition_keys = [key.strip() for key in partition_keys.strip().split(',')]
        for key in partition_keys:
            assert key in self.keys, "Unsupported partition key!"
        partition_errors = {}
        for error, query_info in zip(np.asarray(errors).tolist(), query_infos):
            query_attrs = tuple(getattr(query_info, key) for key in partition_keys)
            partition_errors.setdefault(query_attrs, []).append(error)
        # shrink the result display size: merge adjacent partitions when there are more than 6 (util.py:129-140)
        if len(partition_errors) > 6:
            ordered = [(attrs, partition_errors[attrs]) for attrs in sorted(partition_errors.keys())]
            merged = {}
            for i, (attrs, errs) in enumerate(ordered):
                if i % 2 == 0 and i < len(ordered) - 1:
                    continue
                elif i % 2 == 1:
                    errs += ordered[i - 1][1]
                    merged[attrs] = errs
                else:
                    merged[attrs] = errs
            partition_errors = merged
        for query_attrs in sorted(partition_errors.keys()):
            info_str = ["{}={}".format(key, attr) for key, attr in zip(partition_keys, list(query_attrs))]
            print('Query attributes:' + ','.join(info_str))
            print('# Queries = {}'.format(len(partition_errors[query_attrs])))
            self.get_prediction_statistics(np.array(partition_errors[query_attrs]))

    def get_prediction_statistics(self, errors):
        errors = np.power(2.0, np.asarray(errors, dtype=np.float64))  # back from log2 scale (signed q-error)
        lower, upper = np.quantile(errors, 0.25), np.quantile(errors, 0.75)
        print("<" * 80)
        print("Predict Result Profile of {} Queries:".format(len(errors)))
        print("Min/Max: {:.15f} / {:.15f}".format(np.min(errors), np.max(errors)))
        print("Mean: {:.8f}".format(np.mean(errors)))
        print("Median: {:.8f}".format(np.median(errors)))
        print("25%/75% Quantiles: {:.8f} / {:.8f}".format(lower, upper))
        print("5%/95% Quantiles: {:.8f} / {:.8f}".format(np.quantile(errors, 0.05), np.quantile(errors, 0.95)))
        print(">" * 80)
        return abs(upper - lower)


def q_error_profile(errors) -> dict:
    """The numbers get_prediction_statistics prints, as a dict (used by parity tests and bench notes)."""
    e = np.power(2.0, np.asarray(errors, dtype=np.float64))
    return {"min": float(e.min()), "max": float(e.max()), "mean": float(e.mean()), "median": float(np.median(e)),
            "q25": float(np.quantile(e, 0.25)), "q75": float(np.quantile(e, 0.75)),
            "q05": float(np.quantile(e, 0.05)), "q95": float(np.quantile(e, 0.95))}
