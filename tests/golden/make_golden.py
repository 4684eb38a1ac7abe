#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the REAL reference.

Run in the dev container only (needs /root/reference):   python tests/golden/make_golden.py

The reference publishes no known-answer tests, so every golden vector is an
output of the reference's own code imported from /root/reference with three shims
(SURVEY.md §8c): ``np.int = int`` (seg_hrnet.py:311), a dict-with-attribute-access stand-in
for the yacs config (config/default.py:39-74), and an empty ``cv2`` module so that
``inference.py`` imports (cv2 is only used by dead code there).  Weights/inputs come from the
repo's seed-reproducible generator (esa-pose-estimation_amd/synth.py), so a fixture stores
only (cfg, seed, expected output) — data, never reference source.
"""
import os
import sys
import types
import warnings

import numpy as np

np.int = int                                           # shim 1
sys.modules.setdefault("cv2", types.ModuleType("cv2"))  # shim 3
REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REF)
sys.path.insert(1, ROOT)
warnings.filterwarnings("ignore")

import torch  # noqa: E402

import esa_pose_estimation_amd.synth as synth  # noqa: E402
from models import seg_hrnet, seg_hrnet2, seg_hrnet3  # noqa: E402  (reference)
import inference as ref_inference  # noqa: E402  (reference)

OUT = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(8)


class AD(dict):                                        # shim 2
    __getattr__ = dict.__getitem__


def ref_cfg(widths, blocks):
    def stage(i):
        nb = len(blocks[i])
        return AD(NUM_MODULES=1, NUM_BRANCHES=nb, NUM_BLOCKS=list(blocks[i]),
                  NUM_CHANNELS=list(widths[:nb]), BLOCK="BASIC", FUSE_METHOD="SUM")
    extra = AD(FINAL_CONV_KERNEL=1, STAGE1=stage(0), STAGE2=stage(1), STAGE3=stage(2), STAGE4=stage(3))
    return AD(MODEL=AD(PRETRAINED="", EXTRA=AD(HIGH_RESOLUTION_NET=extra)))


DEFAULT_BLOCKS = ((2,), (2, 2), (2, 2, 2), (4, 4, 4, 4))


def build_ref(variant, widths, blocks, seed, gain=0.5):
    mod = {"seg_hrnet": seg_hrnet, "seg_hrnet2": seg_hrnet2, "seg_hrnet3": seg_hrnet3}[variant]
    net = mod.get_seg_model(ref_cfg(widths, blocks)).eval()
    sd = synth.make_state_dict({k: v.shape for k, v in net.state_dict().items()}, seed=seed, gain=gain)
    net.load_state_dict(sd, strict=True)
    return net, sd


def full_net(tag, variant, widths, blocks, n, hw, seed, subsample=1, taps=False, gain=0.5):
    cin = 3 if variant == "seg_hrnet" else 1
    net, sd = build_ref(variant, widths, blocks, seed, gain)
    x = synth.make_crops(n, cin, hw, hw, seed=seed)
    act_absmax = [0.0]

    def hook(_m, _i, o):                                # largest activation any conv / BN / ReLU emits
        if isinstance(o, torch.Tensor):
            act_absmax[0] = max(act_absmax[0], float(o.abs().max()))
    hooks = [m.register_forward_hook(hook) for m in net.modules() if not list(m.children())]
    with torch.no_grad():
        y = net(x)
        for h_ in hooks:
            h_.remove()
        y64 = net.double()(x.double())
    y = y.numpy()
    rec = dict(variant=variant, widths=np.asarray(widths), blocks_flat=np.asarray(sum(blocks, ())),
               n=n, hw=hw, seed=seed, subsample=subsample, gain=gain, act_absmax=act_absmax[0],
               out=y[:, :, ::subsample, ::subsample].copy(),
               out_absmax=np.abs(y).max(), out_sum=np.float64(y.astype(np.float64).sum()),
               plane_max=y.reshape(n, y.shape[1], -1).max(-1),
               plane_argmax=y.reshape(n, y.shape[1], -1).argmax(-1),
               fp32_vs_fp64_linf=np.abs(y - y64.numpy()).max(),
               n_state_tensors=len(sd),
               state_keys=np.asarray(list(sd.keys())),
               state_shapes=np.asarray([",".join(map(str, v.shape)) for v in sd.values()]))
    np.savez_compressed(os.path.join(OUT, tag + ".npz"), **rec)
    print(f"{tag}: out {y.shape} absmax {rec['out_absmax']:.4f} max|act| {act_absmax[0]:.2f} fp32-vs-fp64 Linf "
          f"{rec['fp32_vs_fp64_linf']:.3e}")


def hr_module(tag, nb, widths, nblocks, hw, seed):
    """Reference HighResolutionModule (seg_hrnet.py:105-249) in isolation."""
    w = list(widths[:nb])
    m = seg_hrnet.HighResolutionModule(nb, seg_hrnet.BasicBlock, list(nblocks), list(w), list(w),
                                       "SUM", True).eval()
    sd = synth.make_state_dict({k: v.shape for k, v in m.state_dict().items()}, seed=seed)
    m.load_state_dict(sd, strict=True)
    xs = [synth.normal(f"{tag}.x{b}", seed, (2, w[b], hw >> b, hw >> b)) for b in range(nb)]
    with torch.no_grad():
        ys = m([torch.from_numpy(x) for x in xs])
    rec = dict(nb=nb, widths=np.asarray(w), nblocks=np.asarray(nblocks), hw=hw, seed=seed)
    for b in range(nb):
        rec[f"y{b}"] = ys[b].numpy()
    np.savez_compressed(os.path.join(OUT, tag + ".npz"), **rec)
    print(f"{tag}: {[tuple(y.shape) for y in ys]}")


def keypoints(tag, hm, note):
    """Reference get_max_preds + the caller's two-stage torch.max + get_final."""
    n, k, h, w = hm.shape
    coords_np, maxvals_np = ref_inference.get_max_preds(hm.copy())
    t = torch.from_numpy(hm)
    a, b = torch.max(t, dim=3)                      # demo.py:172
    c, d = torch.max(a, dim=2)                      # demo.py:173
    refined = np.zeros((n, k, 2), np.float32)
    caller_xy = np.zeros((n, k, 2), np.float32)
    caller_max = np.zeros((n, k), np.float32)
    for i in range(n):
        co = []
        for j in range(k):
            co.append(np.array([b[i][j][d[i][j]].item(), d[i][j].item()], dtype=np.float32))
            caller_max[i, j] = a[i][j][d[i][j]].item()
        caller_xy[i] = np.asarray(co)
        # get_final indexes hm[0] (inference.py:148): hand it one sample at a time
        refined[i] = np.asarray(ref_inference.get_final(hm[i:i + 1].copy(), [c_.copy() for c_ in co]))
    np.savez_compressed(os.path.join(OUT, tag + ".npz"), hm=hm, coords=coords_np,
                        maxvals=maxvals_np[..., 0], caller_xy=caller_xy, caller_max=caller_max,
                        refined=refined, note=note)
    print(f"{tag}: hm {hm.shape}")


def adversarial_planes():
    h = w = 32
    planes = []
    p = np.zeros((h, w), np.float32); p[3, 1] = p[3, 7] = p[9, 2] = p[20, 20] = 1.0    # 4-way tie
    planes.append(p)
    planes.append(np.full((h, w), 0.25, np.float32))                                   # all equal -> (0,0)
    p = np.zeros((h, w), np.float32); p[0, 5] = 2.0; planes.append(p)                  # peak on top border
    p = np.zeros((h, w), np.float32); p[10, 1] = 2.0; planes.append(p)                 # px == 1: no refine
    p = np.zeros((h, w), np.float32); p[10, w - 2] = 2.0; planes.append(p)             # px == W-2
    p = np.zeros((h, w), np.float32); p[10, w - 3] = 2.0; p[10, w - 4] = 1.0; planes.append(p)  # px == W-3 refined
    p = -np.ones((h, w), np.float32); p[12, 12] = -0.5; planes.append(p)               # all negative -> clamp, hxx=0
    ys, xs = np.mgrid[0:h, 0:w]
    g = np.exp(-((xs - 15.3) ** 2 + (ys - 9.8) ** 2) / 8.0).astype(np.float32); planes.append(g)  # true gaussian
    def cross(left, right):
        q = np.full((h, w), 1e-3, np.float32)
        q[10, 15] = 1.0; q[9, 15] = q[11, 15] = 0.6; q[8, 15] = q[12, 15] = 0.2
        q[10, 14] = left; q[10, 16] = right; q[10, 13] = q[10, 17] = 0.95
        return q
    planes.append(cross(0.9, 0.1))                                                     # offset x ~ -43: applied
    planes.append(cross(0.1, 0.9))                                                     # offset x ~ +43: rejected
    p = g.copy(); p[10, 14] = p[10, 16] = p[10, 15]; p[10, 13] = p[10, 17] = p[10, 15]; planes.append(p)  # hxx == 0
    p = g.copy(); p[8, 15] = -3.0; planes.append(p)                                    # negative neighbour -> clamp
    return np.stack(planes)[None]


def main():
    tiny = (8, 16, 32, 64)
    w32 = (32, 64, 128, 256)
    if "--range" in sys.argv:       # dynamic-range goldens only (round 2): weight gain 1.0, SURVEY.md §8d
        full_net("w32_hrnet2_128_g1", "seg_hrnet2", w32, DEFAULT_BLOCKS, 1, 128, seed=3, gain=1.0)
        full_net("w32_hrnet2_256_g1", "seg_hrnet2", w32, DEFAULT_BLOCKS, 1, 256, seed=0, subsample=2, gain=1.0)
        return
    full_net("tiny_hrnet2_64", "seg_hrnet2", tiny, DEFAULT_BLOCKS, 2, 64, seed=1)
    full_net("tiny_hrnet_64", "seg_hrnet", tiny, DEFAULT_BLOCKS, 1, 64, seed=2)
    full_net("w32_hrnet2_128", "seg_hrnet2", w32, DEFAULT_BLOCKS, 1, 128, seed=3)
    full_net("w32_hrnet2_256", "seg_hrnet2", w32, DEFAULT_BLOCKS, 1, 256, seed=0)
    full_net("w32_hrnet_256", "seg_hrnet", w32, DEFAULT_BLOCKS, 1, 256, seed=0, subsample=4)
    # seg_hrnet3 (CBAM; ChannelAttention needs C // 16 >= 1, so the small case uses widths 16..128)
    full_net("small_hrnet3_64", "seg_hrnet3", (16, 32, 64, 128), DEFAULT_BLOCKS, 2, 64, seed=4)
    full_net("w32_hrnet3_128", "seg_hrnet3", w32, DEFAULT_BLOCKS, 1, 128, seed=5)
    hr_module("hrmodule2", 2, (16, 32), (2, 2), 32, seed=11)
    hr_module("hrmodule3", 3, (16, 32, 64), (1, 2, 1), 32, seed=12)
    hr_module("hrmodule4", 4, (8, 16, 32, 64), (1, 1, 1, 2), 32, seed=13)
    keypoints("keypoints_gauss", synth.make_gaussian_heatmaps(2, 11, 64, 64, seed=5).numpy(),
              "sigma-2 blobs + noise")
    keypoints("keypoints_adversarial", adversarial_planes(), "ties/borders/clamp/offset-sign cases")
    keypoints("keypoints_randn", synth.normal("kp_randn", 7, (1, 6, 48, 40)), "raw N(0,1) planes, H!=W")


if __name__ == "__main__":
    main()
