"""ORACLE-side numerics study (test infrastructure): emulate the product's split-bf16
("bf16x3") arithmetic on the CPU to predict its heatmap error before/without a GPU.

Every activation and folded weight is represented as hi + lo with hi = bf16(v),
lo = bf16(v - hi); a convolution is conv(hi_a,hi_w) + conv(hi_a,lo_w) + conv(lo_a,hi_w) with
fp32 accumulation (the lo*lo term is dropped), exactly what the MFMA kernels do.
Run:  python -m oracle.emulate_split_bf16
"""
import sys

import numpy as np
import torch
import torch.nn.functional as F

from oracle import hrnet_ref


def split(t):
    hi = t.to(torch.bfloat16).to(torch.float32)
    lo = (t - hi).to(torch.bfloat16).to(torch.float32)
    return hi, lo


def rq(t):
    hi, lo = split(t)
    return hi + lo


class Emu:
    def __init__(self, sd, terms=3):
        self.sd, self.terms = sd, terms

    def conv(self, name, bn, x, stride=1, relu=False, res=None):
        sd = self.sd
        w = sd[name + ".weight"].double()
        b = sd.get(name + ".bias")
        b = torch.zeros(w.shape[0], dtype=torch.float64) if b is None else b.double()
        if bn:
            g = sd[bn + ".weight"].double() / torch.sqrt(sd[bn + ".running_var"].double() + 1e-5)
            w = w * g[:, None, None, None]
            b = (b - sd[bn + ".running_mean"].double()) * g + sd[bn + ".bias"].double()
        w, b = w.float(), b.float()
        wh, wl = split(w)
        xh, xl = split(x)
        pad = (w.shape[-1] - 1) // 2
        y = F.conv2d(xh, wh, None, stride, pad)
        if self.terms >= 3:
            y = y + F.conv2d(xh, wl, None, stride, pad) + F.conv2d(xl, wh, None, stride, pad)
        y = y + b[None, :, None, None]
        if res is not None:
            y = y + res
        if relu:
            y = F.relu(y)
        return rq(y) if self.terms >= 3 else y.to(torch.bfloat16).float()


def forward(sd, cfg, x0, terms=3):
    e = Emu(sd, terms)
    up = lambda t, size: F.interpolate(t, size=size, mode="bilinear", align_corners=False)
    x = e.conv("conv1", "bn1", x0, relu=True)
    x = e.conv("conv2", "bn2", x, 2, relu=True)

    def block(p, x):
        res = x
        if (p + ".downsample.0.weight") in sd:
            res = e.conv(p + ".downsample.0", p + ".downsample.1", x)
        o = e.conv(p + ".conv1", p + ".bn1", x, relu=True)
        return e.conv(p + ".conv2", p + ".bn2", o, relu=True, res=res)
    for k in range(cfg["blocks"][0][0]):
        x = block(f"layer1.{k}", x)
    ys = [x]
    for s in (2, 3, 4):
        nb = len(cfg["blocks"][s - 1])
        xs = list(ys) + [e.conv(f"transition{s-1}.{nb-1}.0.0", f"transition{s-1}.{nb-1}.0.1", ys[-1], 2, relu=True)]
        p = f"stage{s}.0"
        for b in range(nb):
            for k in range(cfg["blocks"][s - 1][b]):
                xs[b] = block(f"{p}.branches.{b}.{k}", xs[b])
        outs = []
        for i in range(nb):
            y = None
            for j in range(nb):
                if j == i:
                    t = xs[j]
                elif j > i:
                    t = up(e.conv(f"{p}.fuse_layers.{i}.{j}.0", f"{p}.fuse_layers.{i}.{j}.1", xs[j]), xs[i].shape[-2:])
                else:
                    t = xs[j]
                    for k in range(i - j):
                        q = f"{p}.fuse_layers.{i}.{j}.{k}"
                        t = e.conv(q + ".0", q + ".1", t, 2, relu=(k != i - j - 1))
                y = t if y is None else y + t
            outs.append(rq(F.relu(y)))
        ys = outs
    # head: 1x1 480->480 evaluated per branch at native resolution, then upsampled (linearity)
    sdh = dict(sd)
    w = sd["last_layer.0.weight"]
    off, acc = 0, None
    for b, t in enumerate(ys):
        c = t.shape[1]
        sdh[f"_h{b}.weight"] = w[:, off:off + c].contiguous()
        off += c
    g = sd["last_layer.1.weight"].double() / torch.sqrt(sd["last_layer.1.running_var"].double() + 1e-5)
    bias = ((sd["last_layer.0.bias"].double() - sd["last_layer.1.running_mean"].double()) * g + sd["last_layer.1.bias"].double()).float()
    for b, t in enumerate(ys):
        sdh[f"_h{b}.weight"] = (sdh[f"_h{b}.weight"].double() * g[:, None, None, None]).float()
    e2 = Emu(sdh, terms)
    for b, t in enumerate(ys):
        tb = e2.conv(f"_h{b}", None, t)
        tb = tb if b == 0 else up(tb, ys[0].shape[-2:])
        acc = tb if acc is None else acc + tb
    h = rq(F.relu(acc + bias[None, :, None, None]))
    h = e.conv("last_layer.3", "last_layer.4", h, relu=True)
    h = F.interpolate(h, scale_factor=2, mode="bilinear", align_corners=True)
    return F.conv2d(torch.cat([h, x0], 1), sd["output_layer.0.weight"], sd["output_layer.0.bias"], padding=1)


if __name__ == "__main__":
    sys.path.insert(0, ".")
    import esa_pose_estimation_amd.synth as synth
    g = np.load("tests/golden/w32_hrnet2_256.npz")
    cfg = hrnet_ref.default_cfg(1, 11)
    shapes = {str(k): tuple(int(x) for x in s.split(",")) if s else () for k, s in zip(g["state_keys"], g["state_shapes"])}
    sd = synth.make_state_dict(shapes, seed=0)
    x = synth.make_crops(1, 1, 256, 256, seed=0)
    with torch.no_grad():
        for terms in (3, 1):
            y = forward(sd, cfg, x, terms).numpy()
            d = np.abs(y - g["out"])
            print(f"terms={terms}: Linf {d.max():.3e} mean {d.mean():.3e}  argmax moved in "
                  f"{(y.reshape(11,-1).argmax(-1) != g['plane_argmax'][0]).sum()} of 11 planes")
