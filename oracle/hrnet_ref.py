"""ORACLE (test infrastructure, never shipped as product): CPU restatement of the
reference HRNet keypoint-heatmap forward.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
file.  The product path (esa-pose-estimation_amd/) never does.

What it restates (citations relative to /root/reference):
  * models/seg_hrnet.py:425-473 / models/seg_hrnet2.py:426-472   HighResolutionNet.forward
  * models/seg_hrnet.py:225-249                                  HighResolutionModule.forward
  * models/seg_hrnet.py:45-61                                    BasicBlock.forward
  * models/seg_hrnet.py:343-377                                  transition layers
  * models/seg_hrnet.py:176-220                                  fuse layers
  * models/seg_hrnet.py:313-340                                  last_layer / output_layer head

Form: a *functional* interpreter over a reference-style ``state_dict`` (name -> tensor)
using torch.nn.functional on CPU in fp32 (or fp64 for noise-floor studies).  There is no
nn.Module tree here on purpose: the oracle shares no code with the product's module
(esa-pose-estimation_amd/hrnet.py) nor with the reference's.

Pinning: tests/test_oracle_golden.py checks this restatement against fixtures under
tests/golden/ that were produced by importing the REAL reference model in the dev
container (tests/golden/make_golden.py).  Parity is therefore pinned for the model;
see oracle/keypoints_ref.py for the post-processing half.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

BN_EPS = 1e-5  # nn.BatchNorm2d default, seg_hrnet.py:22


def default_cfg(cin: int = 1, num_keypoints: int = 11, widths=(32, 64, 128, 256),
                blocks=((2,), (2, 2), (2, 2, 2), (4, 4, 4, 4)), modules=(1, 1, 1, 1),
                stem_width: int = 64, final_conv_kernel: int = 1, variant: int = 0) -> dict:
    """Stage table of config/default.py:39-74 as a plain dict.

    cin=1,K=11 is seg_hrnet2.py:265,324;  cin=3,K=32 is seg_hrnet.py:265,324;
    variant=1, cin=1, K=30 is seg_hrnet3.py (CBAM, 3x3 last_layer[0], 64-channel stem skip).
    """
    return dict(cin=cin, num_keypoints=num_keypoints, widths=tuple(widths),
                blocks=tuple(tuple(b) for b in blocks), modules=tuple(modules),
                stem_width=stem_width, final_conv_kernel=final_conv_kernel, variant=variant)


# ----------------------------------------------------------------------------- helpers
def _conv(sd, name, x, stride=1):
    w = sd[name + ".weight"]
    b = sd.get(name + ".bias")
    k = w.shape[-1]
    return F.conv2d(x, w.to(x.dtype), None if b is None else b.to(x.dtype),
                    stride=stride, padding=(k - 1) // 2)


def _bn(sd, name, x):
    # eval-mode BatchNorm2d: (x - mean) / sqrt(var + eps) * gamma + beta
    return F.batch_norm(x, sd[name + ".running_mean"].to(x.dtype),
                        sd[name + ".running_var"].to(x.dtype),
                        sd[name + ".weight"].to(x.dtype), sd[name + ".bias"].to(x.dtype),
                        training=False, eps=BN_EPS)


def _cbam(sd, p, x):
    """seg_hrnet3.py:32-61: out = ca(x) * x; out = sa(out) * out.  ``p`` = owner prefix ('' = the net)."""
    q = p + "." if p else ""
    w0, w2 = sd[q + "ca.fc.0.weight"].to(x.dtype), sd[q + "ca.fc.2.weight"].to(x.dtype)

    def fc(v):
        return F.conv2d(F.relu(F.conv2d(v, w0)), w2)
    ca = torch.sigmoid(fc(F.adaptive_avg_pool2d(x, 1)) + fc(F.adaptive_max_pool2d(x, 1)))
    x = ca * x
    m = torch.cat([torch.mean(x, dim=1, keepdim=True), torch.max(x, dim=1, keepdim=True)[0]], dim=1)
    sa = torch.sigmoid(F.conv2d(m, sd[q + "sa.conv1.weight"].to(x.dtype), padding=3))
    return sa * x


def _basic_block(sd, p, x):
    """seg_hrnet.py:45-61 (seg_hrnet3.py:82-103 when the block owns CBAM weights).
    ``p`` is the block prefix, e.g. 'layer1.0'."""
    res = x
    out = F.relu(_bn(sd, p + ".bn1", _conv(sd, p + ".conv1", x)))
    out = _bn(sd, p + ".bn2", _conv(sd, p + ".conv2", out))
    if (p + ".ca.fc.0.weight") in sd:                # seg_hrnet3.py:90-91
        out = _cbam(sd, p, out)
    if (p + ".downsample.0.weight") in sd:           # seg_hrnet.py:55-56
        res = _bn(sd, p + ".downsample.1", _conv(sd, p + ".downsample.0", x))
    return F.relu(out + res)


def _hr_module(sd, p, xs, nblocks):
    """seg_hrnet.py:225-249. ``xs`` list of branch tensors (highest resolution first)."""
    nb = len(xs)
    xs = [x for x in xs]
    for b in range(nb):
        for k in range(nblocks[b]):
            xs[b] = _basic_block(sd, f"{p}.branches.{b}.{k}", xs[b])
    if nb == 1:
        return xs
    outs = []
    for i in range(nb):
        y = None
        for j in range(nb):
            if j == i:
                t = xs[j]
            elif j > i:
                t = _bn(sd, f"{p}.fuse_layers.{i}.{j}.1",
                        _conv(sd, f"{p}.fuse_layers.{i}.{j}.0", xs[j]))
                # F.interpolate(size=..., mode='bilinear') == align_corners=False (:241-244)
                t = F.interpolate(t, size=xs[i].shape[-2:], mode="bilinear",
                                  align_corners=False)
            else:
                t = xs[j]
                for k in range(i - j):
                    q = f"{p}.fuse_layers.{i}.{j}.{k}"
                    t = _bn(sd, q + ".1", _conv(sd, q + ".0", t, stride=2))
                    if k != i - j - 1:               # ReLU on all but the last (:208-216)
                        t = F.relu(t)
            y = t if y is None else y + t
        outs.append(F.relu(y))
    return outs


def _transition(sd, p, ys, n_new):
    """seg_hrnet.py:343-377 + the wiring of :434-457 (new branch fed from ys[-1])."""
    xs = []
    n_pre = len(ys)
    for i in range(n_new):
        if i < n_pre:
            if (f"{p}.{i}.0.weight") in sd:          # width change -> 3x3 s1 conv+BN+ReLU
                xs.append(F.relu(_bn(sd, f"{p}.{i}.1", _conv(sd, f"{p}.{i}.0", ys[i]))))
            else:
                xs.append(ys[i])
        else:
            t = ys[-1]
            for j in range(i + 1 - n_pre):
                q = f"{p}.{i}.{j}"
                t = F.relu(_bn(sd, q + ".1", _conv(sd, q + ".0", t, stride=2)))
            xs.append(t)
    return xs


def forward(sd: dict, cfg: dict, x0: torch.Tensor, taps: dict | None = None) -> torch.Tensor:
    """x0: [N, cin, H, W] -> raw heatmaps [N, K, H, W]  (seg_hrnet.py:425-473).

    ``taps`` (optional dict) receives named intermediate tensors for per-layer parity
    debugging of the HIP path.
    """
    def tap(name, t):
        if taps is not None:
            taps[name] = t
        return t

    v3 = cfg.get("variant", 0) == 1
    skip = _conv(sd, "conv1", x0)                                      # seg_hrnet3.py:473: pre-BN, kept
    tap("stem_raw", skip)
    x = F.relu(_bn(sd, "bn1", skip))                                   # :426-428
    tap("stem1", x)
    x = F.relu(_bn(sd, "bn2", _conv(sd, "conv2", x, stride=2)))        # :429-431
    tap("stem2", x)
    for k in range(cfg["blocks"][0][0]):                               # layer1 :432
        x = _basic_block(sd, f"layer1.{k}", x)
    tap("layer1", x)
    ys = [x]
    for s in (2, 3, 4):
        nb = len(cfg["blocks"][s - 1])
        xs = _transition(sd, f"transition{s - 1}", ys, nb)
        for m in range(cfg["modules"][s - 1]):
            xs = _hr_module(sd, f"stage{s}.{m}", xs, cfg["blocks"][s - 1])
        ys = xs
        for b, t in enumerate(ys):
            tap(f"stage{s}.{b}", t)
    # :461-466  F.upsample(mode='bilinear') == align_corners=False
    size = ys[0].shape[-2:]
    cat = torch.cat([ys[0]] + [F.interpolate(t, size=size, mode="bilinear", align_corners=False)
                               for t in ys[1:]], 1)
    h = F.relu(_bn(sd, "last_layer.1", _conv(sd, "last_layer.0", cat)))         # :313-321
    tap("head0", h)
    h = F.relu(_bn(sd, "last_layer.4", _conv(sd, "last_layer.3", h)))           # :322-329
    tap("head3", h)
    # nn.UpsamplingBilinear2d(scale_factor=2) == align_corners=True (:330)
    h = F.interpolate(h, scale_factor=2, mode="bilinear", align_corners=True)
    if v3:                                                                       # seg_hrnet3.py:516-518
        return _conv(sd, "output_layer.0", torch.cat([h, _cbam(sd, "", skip)], 1))
    return _conv(sd, "output_layer.0", torch.cat([h, x0], 1))                   # :332-340, :469


def conv_flops(cfg: dict, H: int, W: int) -> int:
    """Direct-convolution FLOPs (2*MAC) per crop, the figure SURVEY.md §8d quotes."""
    total = 0
    for c in enumerate_convs(cfg):
        oh, ow = H // c["out_div"], W // c["out_div"]
        total += 2 * oh * ow * c["cout"] * c["cin"] * c["k"] * c["k"]
    return total


def enumerate_convs(cfg: dict) -> list:
    """Independent enumeration of every Conv2d (name, cin, cout, k, stride, resolution).

    ``out_div``: output resolution divisor relative to the input crop.
    Used by the tests to cross-check the product's C++ plan and the SURVEY MAC counts.
    """
    out = []
    w = cfg["widths"]
    sw = cfg["stem_width"]

    def add(name, cin, cout, k, stride, out_div, bias=False, bn=None, relu=False):
        out.append(dict(name=name, cin=cin, cout=cout, k=k, stride=stride, out_div=out_div,
                        bias=bias, bn=bn, relu=relu))

    add("conv1", cfg["cin"], sw, 3, 1, 1, bn="bn1", relu=True)
    add("conv2", sw, sw, 3, 2, 2, bn="bn2", relu=True)
    c_in = sw
    for k in range(cfg["blocks"][0][0]):
        p = f"layer1.{k}"
        add(p + ".conv1", c_in, w[0], 3, 1, 2, bn=p + ".bn1", relu=True)
        add(p + ".conv2", w[0], w[0], 3, 1, 2, bn=p + ".bn2", relu=True)
        if c_in != w[0]:
            add(p + ".downsample.0", c_in, w[0], 1, 1, 2, bn=p + ".downsample.1")
        c_in = w[0]
    pre = [w[0]]
    for s in (2, 3, 4):
        nb = len(cfg["blocks"][s - 1])
        cur = list(w[:nb])
        t = f"transition{s - 1}"
        for i in range(nb):
            if i < len(pre):
                if pre[i] != cur[i]:
                    add(f"{t}.{i}.0", pre[i], cur[i], 3, 1, 2 << i, bn=f"{t}.{i}.1", relu=True)
            else:
                n = i + 1 - len(pre)
                for j in range(n):
                    co = cur[i] if j == n - 1 else pre[-1]
                    add(f"{t}.{i}.{j}.0", pre[-1], co, 3, 2, 2 << (len(pre) + j),
                        bn=f"{t}.{i}.{j}.1", relu=True)
        for m in range(cfg["modules"][s - 1]):
            p = f"stage{s}.{m}"
            for b in range(nb):
                for k in range(cfg["blocks"][s - 1][b]):
                    q = f"{p}.branches.{b}.{k}"
                    add(q + ".conv1", cur[b], cur[b], 3, 1, 2 << b, bn=q + ".bn1", relu=True)
                    add(q + ".conv2", cur[b], cur[b], 3, 1, 2 << b, bn=q + ".bn2", relu=True)
            for i in range(nb):
                for j in range(nb):
                    if j > i:
                        q = f"{p}.fuse_layers.{i}.{j}"
                        add(q + ".0", cur[j], cur[i], 1, 1, 2 << j, bn=q + ".1")
                    elif j < i:
                        for k in range(i - j):
                            q = f"{p}.fuse_layers.{i}.{j}.{k}"
                            last = k == i - j - 1
                            add(q + ".0", cur[j], cur[i] if last else cur[j], 3, 2,
                                2 << (j + k + 1), bn=q + ".1", relu=not last)
        pre = cur
    tot = sum(pre)
    K = cfg["num_keypoints"]
    fk = cfg["final_conv_kernel"]
    v3 = cfg.get("variant", 0) == 1
    add("last_layer.0", tot, tot, 3 if v3 else 1, 1, 2, bias=True, bn="last_layer.1", relu=True)
    add("last_layer.3", tot, K, fk, 1, 2, bias=True, bn="last_layer.4", relu=True)
    add("output_layer.0", K + (sw if v3 else cfg["cin"]), K, 3, 1, 1, bias=True)
    return out
