#!/usr/bin/env python3
"""BASELINE config 5 in miniature: synthetic SPEED-shaped frames -> crops -> HRNet -> keypoints -> host PnP -> CSV.

    python tools/e2e_submission.py --images 1024 --batch 32 [--variant seg_hrnet3] [--out DIR]

Frames are random 1200x1920 uint8 images uploaded from pinned host memory batch by batch on a copy stream, double-
buffered against the compute of the previous batch (so the GPU stage below includes the PCIe copy, overlapped), detector boxes are random squares; the weights are the seed-reproducible random set, so the
network's own keypoints are noise — the run measures the plumbing and the stage rates (the host stage by default on
keypoints of random true poses, see --net-keypoints):
  gpu stage : H2D frames + crop/resize/normalise + forward + arg-max/refine + D2H of [N,K,3]
  host stage: top-k, back-projection, EPnP + RANSAC + weighted LM, quaternion, CSV row   (native C++ by default)
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from esa_pose_estimation_amd import config, crops, inference, pipeline, pnp, synth  # noqa: E402
import esa_pose_estimation_amd as pkg  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--images", type=int, default=1024)
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--variant", default="seg_hrnet2", choices=["seg_hrnet2", "seg_hrnet", "seg_hrnet3"])
ap.add_argument("--scale", type=int, default=256)
ap.add_argument("--out", default="gpurun_out")
ap.add_argument("--workers", type=int, default=0, help="numpy host stage: processes (0 = in-process)")
ap.add_argument("--numpy-pnp", action="store_true", help="host stage by the numpy restatement instead of the native solver")
ap.add_argument("--threads", type=int, default=0, help="native host stage: worker threads (0 = all allowed cores, max 16)")
ap.add_argument("--net-keypoints", action="store_true",
                help="solve poses from the (random-weight) network's keypoints: RANSAC never finds a consensus, i.e. the "
                     "worst case of the host stage; default: keypoints of random true poses projected into the crop + 0.5 px noise")
a = ap.parse_args()

mod = getattr(__import__("esa_pose_estimation_amd." + a.variant), a.variant)
net = mod.get_seg_model(config.make_config())
net.load_state_dict(synth.make_state_dict({k: v.shape for k, v in net.state_dict().items()}, seed=0))
net = net.cuda().eval()
K3 = net.num_keypoints
kp3d = synth.uniform("e2e_kp3d", 1, (K3, 3), -0.6, 0.6).astype(np.float64)
Kcam = np.array([[3003.41297, 0.0, 960.0], [0.0, 3003.41297, 600.0], [0.0, 0.0, 1.0]])

rng = np.random.default_rng(0)
nb = (a.images + a.batch - 1) // a.batch
pin = torch.empty((a.batch, 1200, 1920), dtype=torch.uint8).pin_memory()
pin.numpy()[:] = rng.integers(0, 256, size=(a.batch, 1200, 1920), dtype=np.uint8)     # one pinned batch, re-used


def boxes(n):
    c = rng.uniform([300, 250], [1620, 950], size=(n, 2))
    s = rng.uniform(120, 500, size=(n, 1))
    return np.concatenate([c - s / 2, c + s / 2], 1).astype(int).tolist()


writer = pipeline.SubmissionWriter()
t_gpu = t_host = 0.0
kps, metas = [], []
copy_stream = torch.cuda.Stream()
dev_frames = [torch.empty((a.batch, 1200, 1920), dtype=torch.uint8, device="cuda") for _ in range(2)]
copied = [torch.cuda.Event(), torch.cuda.Event()]
consumed = [torch.cuda.Event(), torch.cuda.Event()]


def upload(slot):                               # H2D of the next batch on the copy stream, behind the last reader of the slot
    with torch.cuda.stream(copy_stream):
        copy_stream.wait_event(consumed[slot])
        dev_frames[slot].copy_(pin, non_blocking=True)
        copied[slot].record(copy_stream)


with torch.no_grad():
    for ev in consumed:
        ev.record()
    upload(0)
    for b in range(nb + 1):                     # first batch = warm-up (weights fold/upload, workspace)
        n = min(a.batch, a.images - max(b - 1, 0) * a.batch) if b else a.batch
        bb = boxes(n)
        if b == 1:
            torch.cuda.synchronize()
            t_start = time.perf_counter()
        slot = b & 1
        if b < nb:
            upload(slot ^ 1)                    # overlaps the compute of this batch
        torch.cuda.current_stream().wait_event(copied[slot])
        x, bx, rates = crops.crop_batch(dev_frames[slot][:n], bb, a.scale)
        consumed[slot].record()
        kp = inference.heatmaps_to_keypoints(net(x)).cpu().numpy()
        if b:
            kps.append(kp)
            metas.append((bx, rates))
    torch.cuda.synchronize()
    t_gpu = time.perf_counter() - t_start
if not a.net_keypoints:                         # what a trained network would hand over: projections of a true pose
    for kp, (bx, rates) in zip(kps, metas):
        for i in range(len(bx)):
            R = pnp.rodrigues(rng.uniform(-1.0, 1.0, 3))
            tv = np.array([rng.uniform(-0.3, 0.3), rng.uniform(-0.3, 0.3), rng.uniform(5.0, 12.0)])
            p2 = pnp.project(kp3d, R, tv, Kcam) + rng.normal(0, 0.5, (K3, 2))
            kp[i, :, :2] = (p2 - np.array([bx[i][0], bx[i][1]])) * rates[i]
            kp[i, :, 2] = rng.uniform(0.5, 1.0, K3)
done = 0
pool = pipeline.pose_pool(a.workers) if (a.workers > 1 and a.numpy_pnp) else None
pipeline.poses_from_keypoints(kps[0][:1], metas[0][0][:1], metas[0][1][:1], kp3d, Kcam, 0.0, min(24, K3))   # warm the library
if pool is not None:                              # spawn the workers outside the timed region
    list(pool.map(int, range(a.workers)))
t0 = time.perf_counter()
for kp, (bx, rates) in zip(kps, metas):
    for q, t in pipeline.poses_from_keypoints(kp, bx, rates, kp3d, Kcam, thresh=0.0, min_k=min(24, K3), pool=pool,
                                              native=not a.numpy_pnp, threads=a.threads):
        writer.append_test(f"img{done:06d}.jpg", q, t)
        done += 1
t_host = time.perf_counter() - t0
if pool is not None:
    pool.shutdown()
os.makedirs(a.out, exist_ok=True)
path = writer.export(out_dir=a.out, suffix="e2e")
print(f"{a.variant}: {done} images, batch {a.batch}: gpu stage {done / t_gpu:.0f} images/s "
      f"({t_gpu / nb * 1e3:.2f} ms per batch incl. {a.batch * 2.3:.0f} MB H2D), "
      f"host PnP stage {done / t_host:.0f} images/s ({'numpy, %d process(es)' % max(a.workers, 1) if a.numpy_pnp else 'native'}); CSV: {path}")
