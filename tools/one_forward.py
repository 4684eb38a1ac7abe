#!/usr/bin/env python3
"""A few eager forwards + keypoints (for rocprofv3 --pmc / --kernel-trace runs)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from esa_pose_estimation_amd import config, inference, seg_hrnet2, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--hw", type=int, default=256)
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
net = seg_hrnet2.get_seg_model(config.make_config())
net.load_state_dict(synth.make_state_dict({k: v.shape for k, v in net.state_dict().items()}, seed=0))
net = net.cuda().eval()
x = synth.make_crops(a.batch, 1, a.hw, a.hw, seed=1).cuda()
with torch.no_grad():
    for _ in range(a.reps):
        kp = inference.heatmaps_to_keypoints(net(x))
torch.cuda.synchronize()
print("ok", tuple(kp.shape))
