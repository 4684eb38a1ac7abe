"""Debug: which intermediate tensor of a forward depends on the crop's POSITION in the batch?  Runs the batch and a
permutation of it with every intermediate kept (ESAHRNET_TAP_ALL=1 adds every convolution output) and prints the
tensors that differ.  usage: dbg_nondet.py [variant] [widths] [hw] [batch]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from esa_pose_estimation_amd import config, seg_hrnet, seg_hrnet2, seg_hrnet3, synth
variant = sys.argv[1] if len(sys.argv) > 1 else "seg_hrnet2"
widths = tuple(int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "32,64,128,256").split(","))
hw = int(sys.argv[3]) if len(sys.argv) > 3 else 256
batch = int(sys.argv[4]) if len(sys.argv) > 4 else 32
mod = {"seg_hrnet2": seg_hrnet2, "seg_hrnet": seg_hrnet, "seg_hrnet3": seg_hrnet3}[variant]
net = mod.get_seg_model(config.make_config(widths=widths))
net.load_state_dict(synth.make_state_dict({k: v.shape for k, v in net.state_dict().items()}, seed=0))
net = net.cuda().eval()
x = synth.make_crops(batch, net._cin, hw, hw, seed=5).cuda()
perm = torch.randperm(batch, generator=torch.Generator().manual_seed(0)).cuda()
with torch.no_grad():
    a = net.taps(x)
    b = net.taps(x[perm])
for k in a:
    ta, tb = a[k], b[k]
    if ta.shape[0] != batch:
        continue
    d = (ta[perm] - tb).abs()
    bad = (d > 0).sum().item()
    if bad:
        idx = (d > 0).nonzero()[:5].tolist()
        print(f"{k:40s} shape {tuple(ta.shape)} mismatches {bad} max {d.max().item():.3e} first {idx}")
print(f"{variant} widths {widths} {hw}x{hw} batch {batch}: {len(a)} tensors compared")
