"""BatchNorm folding (eval mode) of a reference-style state_dict, conv by conv.

models/seg_hrnet.py builds every conv as Conv2d(+bias?) -> BatchNorm2d(eps=1e-5) (-> ReLU);
callers always run net.eval() (val.py:95, demo.py:80), so BN is the per-channel affine
y = (x - mean) * gamma / sqrt(var + eps) + beta and folds into the conv exactly:
w' = w * s,  b' = (b - mean) * s + beta  with s = gamma / sqrt(var + eps).  Done in float64.
"""
from __future__ import annotations

import numpy as np

BN_EPS = 1e-5


def fold_conv(sd, name: str, bn: str, has_bias: bool):
    """-> (w f32 [cout,cin,k,k] C-contiguous, b f32 [cout])."""
    w = sd[name + ".weight"].detach().cpu().double().numpy()
    cout = w.shape[0]
    b = sd[name + ".bias"].detach().cpu().double().numpy() if has_bias else np.zeros(cout)
    if bn:
        gamma = sd[bn + ".weight"].detach().cpu().double().numpy()
        beta = sd[bn + ".bias"].detach().cpu().double().numpy()
        mean = sd[bn + ".running_mean"].detach().cpu().double().numpy()
        var = sd[bn + ".running_var"].detach().cpu().double().numpy()
        s = gamma / np.sqrt(var + BN_EPS)
        w = w * s[:, None, None, None]
        b = (b - mean) * s + beta
    return np.ascontiguousarray(w, dtype=np.float32), np.ascontiguousarray(b, dtype=np.float32)
