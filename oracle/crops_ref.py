"""ORACLE (test infrastructure): CPU restatement of the crop / pad / resize / normalise stage.

Follows data_load_val.py:127-187 for the box arithmetic and the (swapped) edge pad, and restates
OpenCV's published 8-bit INTER_LINEAR resize (half-pixel centres, 11-bit coefficients, int32 horizontal
pass, ((b0*(r0>>4))>>16 + (b1*(r1>>4))>>16 + 2) >> 2 vertical pass) for cv2.resize.  PARITY UNPINNED
against cv2 itself: it is not installable here and the reference holds no image fixture."""
import numpy as np


def val_box(bbox, img_w=1920, img_h=1200, k=1.05):
    x, y, w, h = bbox
    c0, c1 = int((x + w) / 2), int((y + h) / 2)
    size = int(max((w - x), (h - y)) / 2)
    x_new, y_new, w_new, h_new = int(c0 - k * size), int(c1 - k * size), int(c0 + k * size), int(c1 + k * size)
    if x_new < 0:
        w_new -= x_new; x_new = 0
    if y_new < 0:
        h_new -= y_new; y_new = 0
    if w_new > img_w:
        x_new = max(x_new + img_w - w_new, 0); w_new = img_w
    if h_new > img_h:
        y_new = max(y_new + img_h - h_new, 0); h_new = img_h
    return [x_new, y_new, w_new, h_new], max(w_new - x_new, h_new - y_new)


def _coef(dst, src):
    scale = src / dst
    f = ((np.arange(dst, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = f - s.astype(np.float32)
    lo, hi = s < 0, s >= src - 1
    f[lo | hi] = 0.0
    s[lo] = 0
    s[hi] = src - 1
    a0 = np.rint((np.float32(1.0) - f) * np.float32(2048.0)).astype(np.int64)
    a1 = np.rint(f * np.float32(2048.0)).astype(np.int64)
    return s, np.minimum(s + 1, src - 1), a0, a1


def resize_u8_linear(img, dst_h, dst_w):
    src_h, src_w = img.shape
    sx0, sx1, ax0, ax1 = _coef(dst_w, src_w)
    sy0, sy1, by0, by1 = _coef(dst_h, src_h)
    im = img.astype(np.int64)
    rows = im[:, sx0] * ax0 + im[:, sx1] * ax1                    # [src_h, dst_w]
    r0, r1 = rows[sy0], rows[sy1]
    v = (((by0[:, None] * (r0 >> 4)) >> 16) + ((by1[:, None] * (r1 >> 4)) >> 16) + 2) >> 2
    return np.clip(v, 0, 255).astype(np.uint8)


def crop_one(frame, bbox, scale=256, mean=0.485, std=0.229):
    box, size = val_box(bbox, frame.shape[1], frame.shape[0])
    x0, y0, x1, y1 = box
    image = frame[y0:y1, x0:x1]
    xs, ys = x1 - x0, y1 - y0
    if xs != size or ys != size:
        image = np.pad(image, ((0, size - xs), (0, size - ys)), 'edge')     # data_load_val.py:168 (sic)
    rate = 1.0 if size == scale else scale / size
    image = resize_u8_linear(image, scale, scale)
    t = image.astype(np.float32) / np.float32(255.0)
    return ((t - np.float32(mean)) / np.float32(std))[None], box, rate
