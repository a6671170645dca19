"""numpy restatement of the sparse half of Occupancy_Initialization.forward
(models/occupancy_initialization.py:131-174 of the reference) — test infrastructure.
spconv layers: parity unpinned (see oracle/sparse.py); the wiring (BN -> ELAN -> 3 residual
blocks -> SubM(32->1) -> BN) follows the reference lines cited."""
import numpy as np

from . import sparse as OS


def _conv(sd, prefix, x, nbr):
    w, b = sd[prefix + ".weight"], sd[prefix + ".bias"]
    return OS.sparse_conv(x, nbr if w.shape[0] == 27 else None, w, b)


def _block(sd, prefix, x, nbr):
    """SubMconv3dBlock: conv -> LN -> ReLU (models/modules.py:440-452)"""
    y = _conv(sd, prefix + ".conv", x, nbr)
    return OS.layernorm_rows(y, sd[prefix + ".ln.weight"], sd[prefix + ".ln.bias"], post_relu=True)


def sparse_elan(sd, prefix, x, nbr):
    """models/modules.py:401-438"""
    f1 = _block(sd, prefix + ".conv1", x, nbr)
    f2 = _block(sd, prefix + ".conv2", x, nbr)
    parts = [f1, f2]
    for name in ("conv3", "conv4", "conv5", "conv6"):
        parts.append(_block(sd, f"{prefix}.{name}", parts[-1], nbr))
    return _block(sd, prefix + ".conv7", np.concatenate(parts, 1), nbr)


def sparse_stack(sd, var, coords, interval):
    """var f32[N,32] on coords int32[N,4] (one batch element) -> logit f32[N,1]"""
    nbr = OS.kernel_map(coords, coords, 3, interval)
    x = OS.batchnorm_train(var, sd["norm0.weight"], sd["norm0.bias"])
    x = sparse_elan(sd, "similary_1", x, nbr)
    for i in (1, 2, 3):
        y = _conv(sd, f"subm{i}", x, nbr)
        x = OS.layernorm_rows(y, sd[f"norm{i}.weight"], sd[f"norm{i}.bias"], residual=x, pre_relu=True)
    y = _conv(sd, "subm4", x, nbr)
    return OS.batchnorm_train(y, sd["norm4.weight"], sd["norm4.bias"])
