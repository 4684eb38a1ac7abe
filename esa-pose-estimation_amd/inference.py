"""Heatmaps -> keypoints, the drop-in for the reference's inference.py + caller glue.

Reference being replaced:
  * demo.py:172-185 / val.py:151-164   two-stage torch.max + per-keypoint .cpu().item() loop
  * inference.py:22-51                 get_max_preds
  * inference.py:136-152, 75-94        get_final -> my_taylor
  * val.py:172-180 / demo.py:195-200   top-k by peak value, crop -> image coordinates

`heatmaps_to_keypoints` is the fused GPU path ([N,K,H,W] on the device -> [N,K,3] on the
device, one kernel, no host sync); `get_max_preds` / `get_final` keep the reference's names
and numpy-in/numpy-out contract for callers that are not rewritten, but run the same kernel.
"""
from __future__ import annotations

import ctypes as C
import heapq

import numpy as np
import torch

from . import _lib


def _keypoints(heat: torch.Tensor, want_index: bool):
    if not isinstance(heat, torch.Tensor) or heat.dim() != 4:
        raise ValueError("expected a 4-D tensor [N, K, H, W]")
    if not heat.is_cuda:
        raise RuntimeError("heatmaps_to_keypoints runs on the GPU only (no CPU fallback)")
    if heat.dtype != torch.float32:
        raise TypeError(f"expected float32 heatmaps, got {heat.dtype}")
    # heat-maps that come straight out of a forward carry the per-tile maxima their output-layer kernel found
    # (hrnet._Runtime.forward): finishing over those gives the same bits without reading the maps again.  The note is
    # honoured only for this very tensor object, unmodified since (views, clones and in-place edits take the full sweep).
    note = getattr(heat, "_esa_partials", None)
    if note is not None:
        try:
            fresh = note[2] == heat._version
        except RuntimeError:                # inference tensor: no version counter
            fresh = False
        if not fresh or not heat.is_contiguous() or note[0].device != heat.device:
            note = None
    heat = heat.contiguous()
    n, k, h, w = heat.shape
    kp = torch.empty((n, k, 3), dtype=torch.float32, device=heat.device)
    idx = torch.empty((n, k), dtype=torch.int32, device=heat.device) if want_index else None
    stream = torch.cuda.current_stream(heat.device).cuda_stream
    with torch.cuda.device(heat.device):
        if note is not None:
            _lib.check(_lib.lib().esahrnet_keypoints_finish(heat.data_ptr(), note[0].data_ptr(), note[1], n, k, h, w,
                                                            kp.data_ptr(), idx.data_ptr() if want_index else None,
                                                            C.c_void_p(stream)))
        else:
            _lib.check(_lib.lib().esahrnet_keypoints_ex(heat.data_ptr(), n, k, h, w, kp.data_ptr(),
                                                        idx.data_ptr() if want_index else None, C.c_void_p(stream)))
    return kp, idx


def heatmaps_to_keypoints(heat: torch.Tensor) -> torch.Tensor:
    """f32 cuda [N,K,H,W] -> f32 cuda [N,K,3] = (x, y, peak); x=col, y=row, 0-based,
    sub-pixel refined exactly as inference.my_taylor does; peak is the raw maximum."""
    return _keypoints(heat, False)[0]


def _to_device(hm):
    if isinstance(hm, np.ndarray):
        assert hm.ndim == 4, 'batch_images should be 4-ndim'          # inference.py:29
        if not torch.cuda.is_available():
            raise RuntimeError("no GPU visible: the keypoint kernel has no CPU fallback")
        return torch.from_numpy(np.ascontiguousarray(hm, dtype=np.float32)).cuda()
    return hm


def get_max_preds(batch_heatmaps):
    """inference.py:22-51 contract: -> (preds [N,K,2] f32 integer coordinates, maxvals [N,K,1]).
    Runs the same HIP kernel as heatmaps_to_keypoints (keypoints.hip), which also hands back the flat index of
    its first-occurrence arg-max: preds = (idx % W, idx // W), maxvals = the raw peak."""
    assert isinstance(batch_heatmaps, (np.ndarray, torch.Tensor)), \
        'batch_heatmaps should be numpy.ndarray'
    t = _to_device(batch_heatmaps)
    kp, idx = _keypoints(t, True)
    w = t.shape[3]
    idx = idx.cpu().numpy()
    preds = np.stack([(idx % w).astype(np.float32), (idx // w).astype(np.float32)], axis=2)
    return preds, kp[..., 2:3].cpu().numpy()


def get_final(hm, coords=None):
    """inference.py:136-152 contract for one sample: hm [1,K,H,W] -> refined preds [K,2].
    `coords` (the caller's integer arg-max list) is accepted and ignored: the fused kernel
    recomputes the identical arg-max."""
    t = _to_device(hm)
    kp = heatmaps_to_keypoints(t[:1])
    return kp[0, :, :2].cpu().numpy()


def select_keypoints(maxvals, thresh: float = 0.8, min_k: int = 24):
    """val.py:172-177 (thresh .8, at least 24) / demo.py:195-200 (thresh .6, min_k 0):
    indices of the keypoints handed to PnP, largest peak first."""
    mv = [float(v) for v in maxvals]
    large_k = int(np.sum(np.asarray(mv) > thresh))
    large_k = max(large_k, min_k)
    return heapq.nlargest(large_k, range(len(mv)), mv.__getitem__)


def crop_to_image(preds, rate, x, y):
    """val.py:180: ori_preds = preds * (1 / rate) + [x, y]."""
    return np.asarray(preds) * (1 / rate) + [x, y]
