#!/usr/bin/env python3
"""Per-launch table of one forward (HIP-event durations from esahrnet_forward_timed)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from esa_pose_estimation_amd import config, seg_hrnet, seg_hrnet2, seg_hrnet3, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--hw", type=int, default=256)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--variant", default="seg_hrnet2")
ap.add_argument("--widths", default="32,64,128,256", help="branch widths (48,96,192,384 = W48)")
ap.add_argument("--precision", default="bf16x3", choices=["bf16x3", "bf16"])
a = ap.parse_args()
mod = {"seg_hrnet2": seg_hrnet2, "seg_hrnet": seg_hrnet, "seg_hrnet3": seg_hrnet3}[a.variant]
net = mod.get_seg_model(config.make_config(widths=tuple(int(v) for v in a.widths.split(','))), precision=a.precision)
net.load_state_dict(synth.make_state_dict({k: v.shape for k, v in net.state_dict().items()}, seed=0))
net = net.cuda().eval()
x = synth.make_crops(a.batch, net._cin, a.hw, a.hw, seed=1).cuda()
with torch.no_grad():
    net(x)
    acc = None
    for _ in range(a.reps):
        _, ops = net.forward_timed(x)
        if acc is None:
            acc = [dict(o, ms=0.0) for o in ops]
        for q, o in zip(acc, ops):
            q["ms"] += o["ms"] / a.reps
tot = sum(o["ms"] for o in acc)
print(f"{'#':>3} {'kernel':40s} {'label':44s} {'us':>8s} {'%':>5s} {'TF/s':>7s} {'GB/s':>7s}")
for i, o in enumerate(acc):
    tf = o["flops"] / (o["ms"] * 1e-3) / 1e12 if o["ms"] > 0 else 0
    gb = o["bytes"] / (o["ms"] * 1e-3) / 1e9 if o["ms"] > 0 else 0
    print(f"{i:3d} {o['kernel'][:40]:40s} {o['label'][:44]:44s} {o['ms']*1e3:8.1f} {100*o['ms']/tot:5.1f} {tf:7.1f} {gb:7.0f}")
print(f"total {tot:.3f} ms  -> {a.batch / tot * 1e3:.0f} crops/s (eager, events)")
