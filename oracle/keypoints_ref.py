"""ORACLE (test infrastructure, never shipped as product): CPU restatement of the
reference heatmap -> keypoint post-processing.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

What it restates (citations relative to /root/reference):
  * demo.py:172-185, val.py:151-164   caller arg-max: torch.max(dim=3) then torch.max(dim=2)
  * inference.py:22-51                get_max_preds (np.argmax of the flattened plane)
  * inference.py:75-94                my_taylor: log-space quadratic sub-pixel refine
  * inference.py:136-152              get_final: clamp to 1e-10, refine every keypoint
  * val.py:172-180, demo.py:195-200   top-k selection and crop->image back-projection

Pure numpy + math, generalised from the reference's batch-1 (`hm[0]`, inference.py:148)
to per-sample.  Pinned by tests/golden/keypoints_*.npz which were produced by calling the
REAL inference.get_max_preds / inference.get_final (tests/golden/make_golden.py).
"""
from __future__ import annotations

import heapq
import math

import numpy as np


def argmax_keypoints(hm: np.ndarray):
    """hm [N,K,H,W] f32 -> coords [N,K,2] f32 (x=col, y=row), maxvals [N,K] f32.

    First-occurrence tie-break in row-major order (two-stage torch.max of demo.py:172-173
    and np.argmax of inference.py:35 agree).  maxvals are the RAW maxima (demo.py:185).
    """
    assert hm.ndim == 4
    n, k, h, w = hm.shape
    flat = hm.reshape(n, k, -1)
    idx = np.argmax(flat, axis=2)
    maxvals = np.take_along_axis(flat, idx[..., None], axis=2)[..., 0].astype(np.float32)
    coords = np.empty((n, k, 2), np.float32)
    coords[..., 0] = (idx % w).astype(np.float32)
    coords[..., 1] = (idx // w).astype(np.float32)
    return coords, maxvals


def refine_one(plane: np.ndarray, coord: np.ndarray) -> np.ndarray:
    """inference.py:75-94 on a plane already clamped with max(., 1e-10)."""
    h, w = plane.shape
    px, py = int(coord[0]), int(coord[1])
    out = coord.astype(np.float32).copy()
    if 1 < px < w - 2 and 1 < py < h - 2:
        lg = lambda v: math.log(float(v))
        hx = 0.5 * (lg(plane[py][px + 1]) - lg(plane[py][px - 1]))
        hy = 0.5 * (lg(plane[py + 1][px]) - lg(plane[py - 1][px]))
        hxx = 0.25 * (lg(plane[py][px + 2]) - 2 * lg(plane[py][px]) + lg(plane[py][px - 2]))
        hyy = 0.25 * (lg(plane[py + 2][px]) - 2 * lg(plane[py][px]) + lg(plane[py - 2][px]))
        if hxx != 0 and hyy != 0:
            off = [-hx / hxx, -hy / hyy]
            if off[0] < 1 and off[1] < 1:            # signed compare, both-or-neither (:92)
                out += np.asarray(off)               # f64 offset added into f32 coords (:93)
    return out


def refine_keypoints(hm: np.ndarray, coords: np.ndarray) -> np.ndarray:
    """get_final (inference.py:136-152) for every sample of the batch."""
    hmc = np.maximum(hm, 1e-10)                      # :141
    out = np.empty_like(coords, dtype=np.float32)
    for n in range(hm.shape[0]):
        for k in range(hm.shape[1]):
            out[n, k] = refine_one(hmc[n, k], coords[n, k])
    return out


def heatmaps_to_keypoints(hm: np.ndarray) -> np.ndarray:
    """Full a15+a16 path: [N,K,H,W] -> [N,K,3] = (x, y, peak)."""
    coords, maxvals = argmax_keypoints(hm)
    ref = refine_keypoints(hm, coords)
    return np.concatenate([ref, maxvals[..., None]], axis=2).astype(np.float32)


def select_topk(maxvals, thresh: float, min_k: int):
    """val.py:172-177 (thresh=0.8, min_k=24) / demo.py:195-200 (thresh=0.6, min_k=0)."""
    mv = [float(v) for v in maxvals]
    large_k = int(np.sum(np.asarray(mv) > thresh))
    large_k = max(large_k, min_k)
    return heapq.nlargest(large_k, range(len(mv)), mv.__getitem__)


def crop_to_image(preds: np.ndarray, rate: float, x: float, y: float) -> np.ndarray:
    """val.py:180: ori_preds = preds * (1 / rate) + [x, y]."""
    return preds * (1 / rate) + [x, y]
