"""``nt.predict``-shaped front end (reference train.py:171-172,157-158; estimator.py:34-35,66-67).

    predict_fn = predict.gradient_descent_mse_ensemble(kernel_fn, X_train, Y_train, diag_reg=1e-3)
    mean, cov = predict_fn(x_test=X_test, get='nngp', compute_cov=True)

Semantics follow the reference's use of neural-tangents 0.6.1: infinite-time (t=None) closed form,
``diag_reg`` relative to trace(K)/N unless ``diag_reg_absolute_scale``, lazy construction with the
train-train kernel, its factor and alpha cached per ``get`` inside the closure.
"""
from __future__ import annotations

import collections

import numpy as np

from .model import GPModel

Gaussian = collections.namedtuple("Gaussian", ["mean", "covariance"])


def gradient_descent_mse_ensemble(kernel_fn, x_train, y_train, learning_rate: float = 1.0, diag_reg: float = 0.0,
                                  diag_reg_absolute_scale: bool = False, trace_axes=(-1,), **kernel_fn_kwargs):
    if kernel_fn_kwargs:
        raise NotImplementedError("kernel_fn kwargs are not supported: %s" % sorted(kernel_fn_kwargs))
    x_train = np.ascontiguousarray(x_train, dtype=np.float64)
    y_train = np.ascontiguousarray(y_train, dtype=np.float64)
    if x_train.ndim != 2:
        raise ValueError("x_train must be [N, d]")
    if y_train.ndim == 1:
        y_train = y_train[:, None]
    if y_train.shape[0] != x_train.shape[0]:
        raise ValueError("x_train / y_train row mismatch: %d vs %d" % (x_train.shape[0], y_train.shape[0]))
    w_std, b_std = kernel_fn.w_std, kernel_fn.b_std
    models = {}

    def model_for(get: str) -> GPModel:
        if get not in models:
            m = GPModel(x_train.shape[0], x_train.shape[1], w_std, b_std, get=get, diag_reg=diag_reg,
                        diag_reg_absolute_scale=diag_reg_absolute_scale, ny=y_train.shape[1])
            m.fit(x_train, y_train)
            models[get] = m
        return models[get]

    def predict_fn(t=None, x_test=None, get=None, compute_cov=False):
        if t is not None:
            raise NotImplementedError("only the infinite-time posterior (t=None) is implemented, as the reference uses")
        gets = ("nngp", "ntk") if get is None else ((get,) if isinstance(get, str) else tuple(get))
        out = []
        for g in gets:
            if g not in ("nngp", "ntk"):
                raise ValueError("get must be 'nngp' or 'ntk', got %r" % (g,))
            m = model_for(g)
            if compute_cov:
                mean, cov = m.predict(x_test, cov="diag" if compute_cov == "diag" else "full")
                out.append(Gaussian(mean, cov))
            else:
                out.append(m.predict(x_test, cov=False))
        if isinstance(get, str):
            return out[0]
        return collections.namedtuple("Predictions", gets)(*out)

    predict_fn.model_for = model_for
    return predict_fn
