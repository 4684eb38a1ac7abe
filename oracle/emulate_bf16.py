"""ORACLE (test infrastructure, never shipped): CPU emulation of the product's single-pass bf16 mode
(esahrnet_cfg.precision = 1, BASELINE.json configs[3]) — what the HIP kernels of that mode compute, restated with
torch CPU ops, so that the GPU path can be checked TIGHTLY (a few bf16 roundings that fall the other way) and not
only against the fp32 reference at the ~1e-2 that bf16 storage costs.

Arithmetic being emulated (plan.hip with cfg.precision == 1 runs the op-by-op plan):
  * BatchNorm folded into every convolution in float64, then rounded to f32 (fold.py);
  * conv1 (stem): f32 input crop x f32 weights on the f32 VALU -> + bias, ReLU -> stored as bf16;
  * every other convolution: bf16 activations x bf16(folded weight), f32 accumulate, + f32 bias, + bf16 residual,
    ReLU, stored as bf16 (one rounding, round-to-nearest-even);
  * cross-resolution fuse: f32 sum of the bf16 terms, lower-resolution terms bilinearly up-sampled in f32
    (align_corners=False), ReLU, stored as bf16;
  * last_layer[0] (1x1 over the concat) evaluated per branch at the branch's resolution (bias in the branch-0
    slice), each slice stored as bf16, up-sampled + summed + ReLU by the fuse kernel -> bf16; last_layer[3] as any conv;
  * output_layer: f32 VALU on [up x2 (align_corners=True) of the bf16 map, raw f32 crop] with f32 weights -> f32.
Only tests/ may import this.  Topology restated from models/seg_hrnet.py:425-473 as in oracle/hrnet_ref.py (which is
the oracle pinned against the reference's own outputs); this file adds nothing but the roundings.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def q(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.bfloat16).to(torch.float32)


def _fold(sd, name, bn):
    w = sd[name + ".weight"].double()
    b = sd.get(name + ".bias")
    b = torch.zeros(w.shape[0], dtype=torch.float64) if b is None else b.double()
    if bn:
        g = sd[bn + ".weight"].double() / torch.sqrt(sd[bn + ".running_var"].double() + 1e-5)
        w = w * g[:, None, None, None]
        b = (b - sd[bn + ".running_mean"].double()) * g + sd[bn + ".bias"].double()
    return w.float(), b.float()


def forward(sd: dict, cfg: dict, x0: torch.Tensor, taps: dict | None = None) -> torch.Tensor:
    def tap(name, t):
        if taps is not None:
            taps[name] = t
        return t

    def conv(name, bn, x, stride=1, relu=False, res=None, wslice=None, use_bias=True, quant_w=True):
        w, b = _fold(sd, name, bn)
        if wslice is not None:
            w = w[:, wslice[0]:wslice[1]].contiguous()
        y = F.conv2d(x, q(w) if quant_w else w, None, stride, (w.shape[-1] - 1) // 2)
        if use_bias:
            y = y + b[None, :, None, None]
        if res is not None:
            y = y + res
        return q(F.relu(y) if relu else y)

    up = lambda t, size: F.interpolate(t, size=size, mode="bilinear", align_corners=False)
    x = conv("conv1", "bn1", x0, relu=True, quant_w=False)
    tap("stem1", x)
    x = conv("conv2", "bn2", x, 2, relu=True)
    tap("stem2", x)

    def block(p, x):
        res = x
        if (p + ".downsample.0.weight") in sd:
            res = conv(p + ".downsample.0", p + ".downsample.1", x)
        o = conv(p + ".conv1", p + ".bn1", x, relu=True)
        return conv(p + ".conv2", p + ".bn2", o, relu=True, res=res)

    for k in range(cfg["blocks"][0][0]):
        x = block(f"layer1.{k}", x)
    tap("layer1", x)
    ys = [x]
    for s in (2, 3, 4):
        nb = len(cfg["blocks"][s - 1])
        t = f"transition{s - 1}"
        xs = []
        for i in range(nb):
            if i < len(ys):
                if (f"{t}.{i}.0.weight") in sd:              # width change on an existing branch (stage 2, branch 0)
                    xs.append(conv(f"{t}.{i}.0", f"{t}.{i}.1", ys[i], relu=True))
                else:
                    xs.append(ys[i])
            else:
                xs.append(conv(f"{t}.{i}.0.0", f"{t}.{i}.0.1", ys[-1], 2, relu=True))
        for m in range(cfg["modules"][s - 1]):
            p = f"stage{s}.{m}"
            for b in range(nb):
                for k in range(cfg["blocks"][s - 1][b]):
                    xs[b] = block(f"{p}.branches.{b}.{k}", xs[b])
            outs = []
            for i in range(nb):
                y = None
                for j in range(nb):
                    if j == i:
                        tt = xs[j]
                    elif j > i:
                        tt = up(conv(f"{p}.fuse_layers.{i}.{j}.0", f"{p}.fuse_layers.{i}.{j}.1", xs[j]), xs[i].shape[-2:])
                    else:
                        tt = xs[j]
                        for k in range(i - j):
                            qn = f"{p}.fuse_layers.{i}.{j}.{k}"
                            tt = conv(qn + ".0", qn + ".1", tt, 2, relu=(k != i - j - 1))
                    y = tt if y is None else y + tt
                outs.append(q(F.relu(y)))
            xs = outs
        ys = xs
        for b, tt in enumerate(ys):
            tap(f"stage{s}.{b}", tt)
    off, acc = 0, None
    for b, tt in enumerate(ys):
        c = tt.shape[1]
        tb = conv("last_layer.0", "last_layer.1", tt, wslice=(off, off + c), use_bias=(b == 0))
        off += c
        tb = tb if b == 0 else up(tb, ys[0].shape[-2:])
        acc = tb if acc is None else acc + tb
    h = q(F.relu(acc))
    tap("head0", h)
    h = conv("last_layer.3", "last_layer.4", h, relu=True)
    tap("head3", h)
    h = F.interpolate(h, scale_factor=2, mode="bilinear", align_corners=True)
    return F.conv2d(torch.cat([h, x0], 1), sd["output_layer.0.weight"], sd["output_layer.0.bias"], padding=1)
