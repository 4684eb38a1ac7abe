import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from esa_pose_estimation_amd import inference
h = torch.randn(32, 11, 256, 256, device="cuda")
for _ in range(5): kp = inference.heatmaps_to_keypoints(h)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): kp = inference.heatmaps_to_keypoints(h)
e1.record(); torch.cuda.synchronize()
print("keypoints us/call", e0.elapsed_time(e1) / 50 * 1e3)
