"""Seed-reproducible synthetic weights and SPEED-shaped crops.

There is no network for checkpoints or datasets, so parity tests and bench.py run on
random-init weights of the reference architecture.  The generator is a counter-based
hash (splitmix64) so that the GPU box regenerates bit-identical tensors from (name, seed)
without shipping 35.6 MB of weights and without depending on torch's RNG stream.

Recipe (SURVEY.md §8d "Synthetic inputs"): conv weight ~ N(0, sqrt(g/fan_in)) with g=0.5 so
activations stay O(1) through the residual/fuse sums (the reference's own std=1e-3 init,
models/seg_hrnet.py:479, makes every activation ~0 and a 1e-3 tolerance vacuous; He init
g=2 puts the reference's own fp32 noise floor above 1e-3).  BN gets non-trivial affine and
running statistics so that BN folding is really exercised.
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np
import torch

_GOLD = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def _fnv1a(s: str) -> np.uint64:
    h = 0xCBF29CE484222325
    for ch in s.encode():
        h = ((h ^ ch) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return np.uint64(h)


def _splitmix(z: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def _uniform(name: str, seed: int, n: int, stream: int) -> np.ndarray:
    """n doubles in (0,1), a pure function of (name, seed, stream, index)."""
    with np.errstate(over="ignore"):
        base = _splitmix(np.array([_fnv1a(name) ^ np.uint64(seed * 2 + 1)], np.uint64))[0]
        base = base + np.uint64(stream) * np.uint64(0xD1B54A32D192ED03)
        idx = np.arange(1, n + 1, dtype=np.uint64)
        bits = _splitmix(base + idx * _GOLD)
    return ((bits >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)


def normal(name: str, seed: int, shape, std: float = 1.0, mean: float = 0.0) -> np.ndarray:
    n = int(np.prod(shape)) if len(shape) else 1
    u1 = _uniform(name, seed, n, 0)
    u2 = _uniform(name, seed, n, 1)
    z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
    return (mean + std * z).astype(np.float32).reshape(shape)


def uniform(name: str, seed: int, shape, lo: float, hi: float) -> np.ndarray:
    n = int(np.prod(shape)) if len(shape) else 1
    return (lo + (hi - lo) * _uniform(name, seed, n, 0)).astype(np.float32).reshape(shape)


def make_state_dict(shapes, seed: int = 0, gain: float = 0.5) -> "OrderedDict[str, torch.Tensor]":
    """``shapes``: mapping name -> shape (e.g. ``{k: v.shape for k, v in net.state_dict().items()}``)."""
    out = OrderedDict()
    for name, shape in shapes.items():
        shape = tuple(shape)
        if name.endswith("num_batches_tracked"):
            out[name] = torch.zeros((), dtype=torch.long)
        elif name.endswith("running_mean"):
            out[name] = torch.from_numpy(normal(name, seed, shape, 0.1))
        elif name.endswith("running_var"):
            out[name] = torch.from_numpy(uniform(name, seed, shape, 0.5, 1.5))
        elif len(shape) == 4:                                   # conv weight
            fan_in = shape[1] * shape[2] * shape[3]
            out[name] = torch.from_numpy(normal(name, seed, shape, float(np.sqrt(gain / fan_in))))
        elif name.endswith(".weight"):                          # BN gamma
            out[name] = torch.from_numpy(uniform(name, seed, shape, 0.5, 1.5))
        elif name.endswith(".bias"):                            # conv bias / BN beta
            out[name] = torch.from_numpy(normal(name, seed, shape, 0.1))
        else:
            raise KeyError(f"synth: do not know how to fill {name!r}")
    return out


def make_crops(n: int, cin: int, h: int, w: int, seed: int = 0) -> torch.Tensor:
    """SPEED-shaped normalised crops: f32 [n, cin, h, w] ~ N(0,1) (real crops normalised by
    mean~0.45 / std 0.229 land in about [-2, 2.4]; data_load_val.py:86)."""
    return torch.from_numpy(normal(f"crops{cin}x{h}x{w}", seed, (n, cin, h, w)))


def make_gaussian_heatmaps(n: int, k: int, h: int, w: int, seed: int = 0, sigma: float = 2.0,
                           noise: float = 0.01) -> torch.Tensor:
    """Heatmaps with one sub-pixel-centred Gaussian blob per plane plus a little noise —
    the shape the trained network emits (data_load4.py:54-58 labels, sigma = gauss_size)."""
    cx = uniform("hm_cx", seed, (n, k), 4.0, w - 5.0).astype(np.float64)
    cy = uniform("hm_cy", seed, (n, k), 4.0, h - 5.0).astype(np.float64)
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float64)
    g = np.exp(-((xs[None, None] - cx[..., None, None]) ** 2 +
                 (ys[None, None] - cy[..., None, None]) ** 2) / (2 * sigma * sigma))
    g = g + noise * normal("hm_noise", seed, (n, k, h, w)).astype(np.float64)
    return torch.from_numpy(g.astype(np.float32))


# ---- synthetic SPEED-shaped evaluation set (BASELINE.json configs[4]; SURVEY.md §8f NEXT-2 "12 000-image set") ----
ESA_CAMERA = np.array([[3003.41297, 0.0, 960.0], [0.0, 3003.41297, 600.0], [0.0, 0.0, 1.0]])   # lib/utils/base_utils.py:250-252


def make_scene(n: int, k: int, seed: int = 0, img_w: int = 1920, img_h: int = 1200):
    """n images of one rigid k-keypoint model under known random poses, as the pipeline meets them:
    -> dict(kp3d [k,3], q [n,4] = [w,x,y,z], t [n,3], uv [n,k,2] image-pixel projections with the ESA camera,
            bboxes [n][4] = (x, y, x2, y2) detector-style boxes around the projections).
    Poses are drawn so that the whole model stays inside the 1920x1200 frame (SPEED's spacecraft always is)."""
    kp3d = uniform("scene_kp3d", seed, (k, 3), -0.6, 0.6).astype(np.float64)
    rv = normal("scene_rv", seed, (n, 4)).astype(np.float64)
    q = rv / np.linalg.norm(rv, axis=1, keepdims=True)
    tz = uniform("scene_tz", seed, (n,), 4.0, 14.0).astype(np.float64)
    fx, cx, cy = ESA_CAMERA[0, 0], ESA_CAMERA[0, 2], ESA_CAMERA[1, 2]
    # image-plane centre anywhere that keeps a 1.1 m radius model inside the frame
    r_px = fx * 1.1 / tz
    u0 = cx + (uniform("scene_u", seed, (n,), -1.0, 1.0) * np.maximum(img_w / 2 - r_px - 8, 0)).astype(np.float64)
    v0 = cy + (uniform("scene_v", seed, (n,), -1.0, 1.0) * np.maximum(img_h / 2 - r_px - 8, 0)).astype(np.float64)
    t = np.stack([(u0 - cx) * tz / fx, (v0 - cy) * tz / fx, tz], 1)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], 1),
                  np.stack([2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)], 1),
                  np.stack([2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], 1)], 1)
    pc = np.einsum("nij,kj->nki", R, kp3d) + t[:, None, :]
    uv = np.stack([fx * pc[..., 0] / pc[..., 2] + cx, fx * pc[..., 1] / pc[..., 2] + cy], 2)
    lo, hi = uv.min(1), uv.max(1)
    pad = 0.08 * (hi - lo).max(1, keepdims=True) + 4.0            # a detector box is a little loose
    bboxes = np.concatenate([np.floor(lo - pad), np.ceil(hi + pad)], 1).astype(int)
    bboxes[:, [0, 2]] = np.clip(bboxes[:, [0, 2]], 0, img_w)
    bboxes[:, [1, 3]] = np.clip(bboxes[:, [1, 3]], 0, img_h)
    return dict(kp3d=kp3d, q=q, t=t, uv=uv, bboxes=bboxes.tolist())


def render_heatmaps(centers: torch.Tensor, size: int, sigma: float = 2.0) -> torch.Tensor:
    """What a trained network emits for keypoints at `centers` [N,K,2] (crop pixels, x then y): sigma-2 Gaussian
    blobs (data_load4.py:54-58 labels) -> f32 [N,K,size,size] on centers' device."""
    ax = torch.arange(size, dtype=torch.float32, device=centers.device)
    dx = (ax[None, None, :] - centers[..., 0:1]) ** 2            # [N,K,S]
    dy = (ax[None, None, :] - centers[..., 1:2]) ** 2
    return torch.exp(-(dy[..., :, None] + dx[..., None, :]) / (2.0 * sigma * sigma))
