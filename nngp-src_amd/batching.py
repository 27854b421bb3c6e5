"""``nt.batch`` slot (reference train.py:166-168, estimator.py:31-33, active/ActiveLearner.py:24-26).

The reference calls ``nt.batch(kernel_fn, device_count=0, batch_size=0)``, which is the identity.  The
two knobs keep their meaning here: ``batch_size > 0`` tiles the kernel build serially over row blocks,
``device_count > 0`` shards row blocks over the ranks of the active ``torch.distributed`` group and
all-gathers them (RCCL over xGMI on MI355X) -- see distributed.py.
"""
from __future__ import annotations

import numpy as np


def batch(kernel_fn, batch_size: int = 0, device_count: int = -1, store_on_device: bool = True):
    if batch_size in (0, None) and device_count in (0, None):
        return kernel_fn  # the reference configuration: identity

    def batched_kernel_fn(x1, x2=None, get=None, **kw):
        from . import distributed
        n1 = int(np.shape(x1)[0])
        if device_count and device_count != 0 and distributed.world_size() > 1:
            return distributed.sharded_kernel(kernel_fn, x1, x2, get)
        if not batch_size or batch_size >= n1:
            return kernel_fn(x1, x2, get, **kw)
        if n1 % batch_size != 0:
            raise ValueError("batch_size (%d) must divide the number of rows (%d)" % (batch_size, n1))
        blocks = [kernel_fn(x1, x2, get, rows=(r, r + batch_size), **kw) for r in range(0, n1, batch_size)]
        if isinstance(blocks[0], tuple):
            return type(blocks[0])(*[np.concatenate([b[i] for b in blocks], axis=0) for i in range(len(blocks[0]))])
        return np.concatenate(blocks, axis=0)

    for attr in ("w_std", "b_std", "n_relu"):
        setattr(batched_kernel_fn, attr, getattr(kernel_fn, attr))
    batched_kernel_fn.inner = kernel_fn
    return batched_kernel_fn
