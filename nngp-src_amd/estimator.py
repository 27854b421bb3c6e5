"""``Estimator`` -- the PostgreSQL-facing serving class (reference neuroestimator/estimator/estimator.py:16-68).

Same constructor, ``load_model()`` and ``predict(query_lines) -> (pred_mean[M], pred_std[M])`` as the
reference.  Differences that do not change results: the GP lives in HBM behind the C ABI; ``load_model``
fits (kernel build + Cholesky + alpha) instead of also forming the N x N posterior covariance the
reference computes only to warm its cache (estimator.py:37-40); ``predict`` asks the device for
diag(cov) only, which is all the reference consumes (estimator.py:55).
"""
from __future__ import annotations

import datetime

import numpy as np

from . import stax, predict as nt_predict
from .batching import batch


class Estimator(object):
    def __init__(self, schema_name: str, data_path: str, train_query_path: str, chunk_size: int = 64,
                 use_aux: bool = False, q_error_threshold: float = 100.0, coef_var_threshold: float = 1.0,
                 encoder=None, kernel_type: str = "nngp", serving: bool = True):
        self.schema_name = schema_name
        self.data_path = data_path
        self.train_query_path = train_query_path
        self.chunk_size = chunk_size
        self.kernel_type = kernel_type
        self.serving = serving  # load_model also builds the explicit float64 inverse: predict = one product per batch
        print("loading schema and training data ... This may take seconds ...")
        if encoder is None:
            from .schemas import load_training_schema_data
            X_train, Y_train, self.nngp_encoder = load_training_schema_data(
                schema_name, data_path, train_query_path, chunk_size, use_aux, q_error_threshold, coef_var_threshold)
        else:
            self.nngp_encoder = encoder
            queries, cards, _ = encoder.load_queries(train_query_path, use_aux, q_error_threshold, coef_var_threshold)
            X_train, Y_train = encoder.transform_to_arrays(queries, cards)
        self.X_train, self.Y_train = np.asarray(X_train, dtype=np.float64), np.asarray(Y_train, dtype=np.float64)
        try:
            from .encoder import NativeEncoder
            self._native = NativeEncoder.from_encoder(self.nngp_encoder)
        except Exception:  # a user-supplied encoder object without table metadata: keep its Python path
            self._native = None
        print("Building model kernel ...")
        init_fn, apply_fn, kernel_fn = stax.serial(stax.Dense(512), stax.Relu(), stax.Dense(1))
        kernel_fn = batch(kernel_fn, device_count=0, batch_size=0)
        self.predict_fn = nt_predict.gradient_descent_mse_ensemble(kernel_fn, self.X_train, self.Y_train, diag_reg=1e-3)

    def load_model(self):
        model = self.predict_fn.model_for(self.kernel_type)  # kernel build + Cholesky + alpha, cached in HBM
        if self.serving:
            model.prepare_serving()
        n = self.X_train.shape[0]
        print((n, model.ny), (n, n))  # the shapes the reference prints (estimator.py:39)
        print("Model construction complete.")

    def predict(self, query_lines):
        start = datetime.datetime.now()
        if self._native is not None:  # one native call for the whole batch (estimator.py:46-49 loops in Python)
            X_test = self._native.encode_lines(query_lines, with_card=False)
        else:
            X_test = [self.nngp_encoder.parse_line_without_card_then_encode(line) for line in query_lines]
        X_test = np.asarray(X_test, dtype=np.float64).reshape(len(query_lines), self.X_train.shape[1])
        pred_mean, pred_var = self._nngp_prediction(X_test)
        duration = (datetime.datetime.now() - start).total_seconds()
        print("prediction time={} seconds".format(duration))
        pred_std = np.sqrt(pred_var)
        return pred_mean.ravel(), pred_std.ravel()

    def _nngp_prediction(self, X_test, kernel_type=None, compute_cov="diag"):
        g = self.predict_fn(x_test=X_test, get=kernel_type or self.kernel_type, compute_cov=compute_cov)
        return g.mean, g.covariance
