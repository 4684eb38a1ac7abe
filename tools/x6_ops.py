"""Per-launch table of one fp32-mode forward (HIP events per launch): python tools/x6_ops.py [batch]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from esa_pose_estimation_amd import config, seg_hrnet2, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
net = seg_hrnet2.get_seg_model(config.make_config())
sd = synth.make_state_dict({k: v.shape for k, v in net.state_dict().items()}, seed=0)
net.load_state_dict(sd)
net = net.cuda().eval()
x = synth.make_crops(n, 1, 256, 256, seed=1).cuda()
with torch.no_grad():
    for _ in range(3):
        y, ops = net.forward_timed(x)
    acc = [dict(o) for o in ops]
    for _ in range(4):
        y, ops = net.forward_timed(x)
        for a, o in zip(acc, ops):
            a["ms"] += o["ms"]
tot = 0
for a in acc:
    ms = a["ms"] / 5
    tot += ms
    tf = a["flops"] / (ms * 1e-3) / 1e12 if ms > 0 else 0
    print(f"{ms * 1e3:8.1f} us {tf:7.1f} TF {a['bytes'] / (ms * 1e-3) / 1e9 if ms > 0 else 0:7.0f} GB/s  {a['kernel']:34s} {a['label'][:90]}")
print(f"total {tot:.3f} ms")
