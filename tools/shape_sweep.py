#!/usr/bin/env python3
"""Every legal crop size must run: forward + keypoints over a sweep of even sizes (odd level sizes, tile remainders, tiny and
large crops) for the three variants and both precisions; reports launch failures and non-finite outputs."""
import itertools, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from esa_pose_estimation_amd import config, inference, seg_hrnet, seg_hrnet2, seg_hrnet3, synth

sizes = [16, 18, 22, 34, 50, 66, 98, 130, 200, 258, 322]


def run(log=print):
    bad = 0
    for variant, mod, prec in (("seg_hrnet2", seg_hrnet2, "bf16x3"), ("seg_hrnet3", seg_hrnet3, "bf16x3"), ("seg_hrnet", seg_hrnet, "bf16x3"),
                               ("seg_hrnet2", seg_hrnet2, "bf16")):
        net = mod.get_seg_model(config.make_config(widths=(32, 64, 128, 256)), precision=prec) if prec != "bf16x3" else mod.get_seg_model(config.make_config())
        net.load_state_dict(synth.make_state_dict({k: v.shape for k, v in net.state_dict().items()}, seed=0))
        net = net.cuda().eval()
        n_ok = 0
        for h, w in itertools.product(sizes, sizes):
            if (sizes.index(h) + sizes.index(w)) % 3 and h != w:      # a third of the off-diagonal pairs
                continue
            x = synth.make_crops(2, net._cin, h, w, seed=1).cuda()
            try:
                with torch.no_grad():
                    y = net(x)
                    kp = inference.heatmaps_to_keypoints(y)
                torch.cuda.synchronize()
                if not (torch.isfinite(y).all() and torch.isfinite(kp).all()):
                    log("NON-FINITE", variant, prec, h, w); bad += 1
                else:
                    n_ok += 1
            except Exception as e:      # noqa: BLE001
                log("FAIL", variant, prec, h, w, str(e)[:120]); bad += 1
        log("%s %s ok: %d" % (variant, prec, n_ok))
    log("bad: %d" % bad)
    return bad


if __name__ == "__main__":
    sys.exit(1 if run() else 0)
