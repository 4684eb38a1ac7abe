"""Loader contract in front of the path (SURVEY.md §8f NEXT-2): bbox -> clamped crop box -> GPU crop.

Reference being mirrored: ESAValDataSet.__getitem__ (data_load_val.py:103-195): the detector box
(x, y, x2, y2) is squared around its centre, scaled by 1.05, shifted back inside the 1920x1200 frame,
the crop is edge-padded, resized to `scale`, divided by 255 and normalised with mean 0.485 / std 0.229;
the caller later needs `bbox` (crop origin) and `rate` (= scale / size) to map keypoints back
(val.py:180).  `val_box` is the integer box arithmetic (host, a few ints per image); the pixel work runs
in crops.hip on the frames already resident on the GPU.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib

IMG_W, IMG_H = 1920, 1200            # data_load_val.py:77-78
MEAN_VAL, STD = 0.485, 0.229         # data_load_val.py:86  (the TRAIN loaders use 0.449: data_load4.py:81)


def val_box(bbox, img_w: int = IMG_W, img_h: int = IMG_H, k: float = 1.05):
    """data_load_val.py:127-158 -> ([x_new, y_new, w_new, h_new], size); (w_new, h_new) are the far corner."""
    x, y, w, h = bbox
    c0 = int((x + w) / 2)
    c1 = int((y + h) / 2)
    size = int(max((w - x), (h - y)) / 2)
    x_new, y_new = int(c0 - k * size), int(c1 - k * size)
    w_new, h_new = int(c0 + k * size), int(c1 + k * size)
    if x_new < 0:
        w_new -= x_new
        x_new = 0
    if y_new < 0:
        h_new -= y_new
        y_new = 0
    if w_new > img_w:
        x_new = x_new + img_w - w_new
        if x_new < 0:
            x_new = 0
        w_new = img_w
    if h_new > img_h:
        y_new = y_new + img_h - h_new
        if y_new < 0:
            y_new = 0
        h_new = img_h
    return [x_new, y_new, w_new, h_new], max(w_new - x_new, h_new - y_new)


def crop_batch(frames: torch.Tensor, bboxes, scale: int = 256, mean: float = MEAN_VAL, std: float = STD):
    """frames: uint8 cuda [N, H, W] (gray camera frames); bboxes: N detector boxes (x, y, x2, y2).
    -> (crops f32 cuda [N,1,scale,scale], boxes [N][4] ints, rates [N]) — image, bbox, rate of
    data_load_val.py:195."""
    if not (isinstance(frames, torch.Tensor) and frames.is_cuda and frames.dtype == torch.uint8 and frames.dim() == 3):
        raise TypeError("frames must be a uint8 CUDA tensor [N, H, W] (no CPU fallback)")
    frames = frames.contiguous()
    n, fh, fw = frames.shape
    if len(bboxes) != n:
        raise ValueError(f"{len(bboxes)} detector boxes for {n} frames (crop_kernel reads one box per frame)")
    boxes, rates = [], []
    for b in bboxes:
        box, size = val_box(b, fw, fh)
        if box[2] <= box[0] or box[3] <= box[1]:
            raise ValueError(f"empty crop box {box} from detector box {list(b)}")
        boxes.append(box)
        rates.append(1.0 if size == scale else scale / size)
    bt = torch.tensor(boxes, dtype=torch.int32, device=frames.device)
    out = torch.empty((n, 1, scale, scale), dtype=torch.float32, device=frames.device)
    stream = torch.cuda.current_stream(frames.device).cuda_stream
    with torch.cuda.device(frames.device):
        _lib.check(_lib.lib().esahrnet_crops(frames.data_ptr(), n, fh, fw, bt.data_ptr(), scale, mean, std,
                                             out.data_ptr(), C.c_void_p(stream)))
    return out, boxes, rates
