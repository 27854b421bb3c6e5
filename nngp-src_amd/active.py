"""Active-learning loop over the GP hot path (reference active/ActiveLearner.py:15-77).

Same control flow as the reference: fit, score a pool by predictive standard deviation relative to max(mean),
move the `budget` most uncertain (or std-proportionally sampled) pool queries into the training set, refit.
The first fit is a full kernel build + Cholesky on the GPU; every later round only builds the kernel rows of the
queries it adds and EXTENDS the factor (``GPModel.append`` / ``nngp_model_append``: a blocked triangular solve against
the existing factor plus a small Cholesky, ~b N^2 flops instead of N^3/3), then re-solves alpha in float64.  The model
handle is sized for the final training set, so nothing is reallocated between iterations.  Differences from the reference: only
diag(cov) is requested.  The biased sampler -- the reference's default, ``--biased_sample True`` (active_train.py:62) -- restates
``jax.random.choice(PRNGKey(10), n, (k,), replace=False, p=std_prob)`` step by step (``jaxrand.py``: Threefry-2x32-20 -> 52-bit
uniforms -> Gumbel top-k), on the device (``nngp_pool_select``), in the host build and in the NumPy fallback below: the draw is the
reference's draw index for index wherever jax's own ``log`` and ours round alike (pinned by the Random123 vectors of Threefry, not
against jax itself, which is not installable here).  NaN scores (a NaN variance or mean): the reference's argsort puts NaN LAST,
i.e. its top-k path would select them first and its biased path never; here a NaN score never wins in either path.
"""
from __future__ import annotations

import numpy as np

from .model import GPModel
from .util import PredictionStatistics


class ActiveLearner(object):
    def __init__(self, args=None, budget=1000, active_iters=3, kernel_type="nngp", biased_sample=True):
        self.args = args
        self.budget = getattr(args, "budget", budget)
        self.active_iters = getattr(args, "active_iters", active_iters)
        self.kernel_type = getattr(args, "kernel_type", kernel_type)
        self.biased_sample = getattr(args, "biased_sample", biased_sample)
        self.pred_stat = PredictionStatistics()
        self._model = None
        self._fitted = None      # (X, Y) of the last fit, to recognise an appended training set
        self.incremental = getattr(args, "incremental", True)
        self.history = []

    # -- reference: ActiveLearner.train (ActiveLearner.py:23-31) --
    def train(self, kernel_fn, X_train, Y_train, X_test=None, Y_test=None, n_cap=None):
        X_train = np.ascontiguousarray(X_train, dtype=np.float64)
        Y_train = np.ascontiguousarray(Y_train, dtype=np.float64).reshape(X_train.shape[0], -1)
        n, d = X_train.shape
        if self._model is None or self._model.n_cap < n or self._model.d != d or self._model.get != self.kernel_type:
            if self._model is not None:
                self._model.close()
            self._fitted = None
            self._model = GPModel(max(n, n_cap or n), d, kernel_fn.w_std, kernel_fn.b_std, get=self.kernel_type,
                                  diag_reg=1e-3, ny=Y_train.shape[1])
        # When the new training set extends the fitted one (the loop below appends the selected pool queries), only the
        # new kernel rows are built and the factor is extended (GPModel.append) instead of a full refit.
        prev = self._fitted
        if (self.incremental and prev is not None and self._model.n == prev[0].shape[0] and 128 <= prev[0].shape[0] < n
                and np.array_equal(X_train[:prev[0].shape[0]], prev[0]) and np.array_equal(Y_train[:prev[0].shape[0]], prev[1])):
            n_old = prev[0].shape[0]
            self._model.append(X_train[n_old:], Y_train[n_old:])
        else:
            self._model.fit(X_train, Y_train)
        self._fitted = (X_train, Y_train)

        def predict_fn(x_test=None, get=None, compute_cov=False):
            assert get in (None, self.kernel_type)
            if compute_cov:
                return self._model.predict(x_test, cov="diag" if compute_cov == "diag" else "full")
            return self._model.predict(x_test, cov=False)

        predict_fn.learner = self  # active_test scores the pool on the device through the learner's CURRENT model
        return predict_fn

    # -- reference: ActiveLearner.test (ActiveLearner.py:33-40) --
    def test(self, predict_fn, X_val, Y_val, query_infos_val=None, kernel_type="nngp", compute_cov=True):
        pred_mean = predict_fn(x_test=X_val, get=kernel_type, compute_cov=False)
        errors = pred_mean - Y_val
        mse = float(np.mean(np.power(errors, 2.0)))
        print("Test MSE Loss:{}".format(mse))
        self.pred_stat.get_prediction_details(np.ravel(errors), query_infos_val, partition_keys='num_predicates')
        return mse

    # -- reference: ActiveLearner.active_test (ActiveLearner.py:43-55) --
    def active_test(self, predict_fn, X_test, kernel_type="nngp"):
        num_test = X_test.shape[0]
        num_select = self.budget if num_test > self.budget else num_test
        # the model is looked up NOW: a predict_fn kept across a train() that replaced the model would hold a closed handle
        model = self._model if getattr(predict_fn, "learner", None) is self else None
        if model is not None and num_test > 0:
            assert kernel_type == model.get, "active_test: the fitted model is a %r posterior" % model.get
            # std / max(mean), top-`budget` or the std-proportional draw, all on the GPU: only the indices come back
            return model.select_pool(X_test, num_select, biased=self.biased_sample, seed=10)
        pred_mean, pred_var = predict_fn(x_test=X_test, get=kernel_type, compute_cov="diag")
        pred_std = np.sqrt(np.maximum(pred_var, 0.0))
        pred_std = pred_std / np.max(pred_mean, 0)
        pred_std = np.reshape(pred_std, (num_test,))
        if self.biased_sample:
            from .jaxrand import choice_without_replacement
            std_prob = pred_std / np.sum(pred_std)
            return choice_without_replacement(10, num_test, num_select, std_prob)
        return np.argsort(pred_std)[-num_select:]

    # -- reference: ActiveLearner.merge_data (ActiveLearner.py:57-65) --
    def merge_data(self, select_indices, X_train, Y_train, X_test, Y_test):
        X_train_new = np.vstack((X_train, X_test[select_indices]))
        Y_train_new = np.vstack((Y_train, Y_test[select_indices]))
        keep = np.setdiff1d(np.arange(X_test.shape[0]), np.asarray(select_indices))
        return X_train_new, Y_train_new, X_test[keep], Y_test[keep]

    # -- reference: ActiveLearner.active_train (ActiveLearner.py:67-77) --
    def active_train(self, kernel_fn, X_train, Y_train, X_test, Y_test, X_val, Y_val, query_infos_val=None):
        print("# Initial Training samples: {}".format(X_train.shape[0]))
        n_cap = X_train.shape[0] + min(self.budget * self.active_iters, X_test.shape[0])
        predict_fn = self.train(kernel_fn, X_train, Y_train, n_cap=n_cap)
        self.history = [self.test(predict_fn, X_val, Y_val, query_infos_val, self.kernel_type)]
        for i in range(self.active_iters):
            select_indices = self.active_test(predict_fn, X_test, self.kernel_type)
            print("Active Iteration {}: Selection {}".format(i, select_indices.shape[0]))
            X_train, Y_train, X_test, Y_test = self.merge_data(select_indices, X_train, Y_train, X_test, Y_test)
            print("# Training samples: {}".format(X_train.shape[0]))
            predict_fn = self.train(kernel_fn, X_train, Y_train, n_cap=n_cap)
            self.history.append(self.test(predict_fn, X_val, Y_val, query_infos_val, self.kernel_type))
        return predict_fn
