"""Batch-N evaluation / submission driver (SURVEY.md §8f NEXT-4).

Reference being mirrored: the per-image loop of val.py:136-233 (batch 1, >=150 blocking .item() reads
per image) and submission.py:6-52.  Here: frames and detector boxes in, poses out, N crops at a time:
  crops.crop_batch -> net -> inference.heatmaps_to_keypoints -> (parallel.gather_keypoints) -> ONE D2H
  copy of [N, K, 3] -> host: top-k, back-projection, EPnP + RANSAC, peak-weighted refinement, quaternion.
`SubmissionWriter` keeps the reference's API and CSV format (filename, q0..q3, r0..r2, sorted by name).
"""
from __future__ import annotations

import contextlib
import csv
import os
from concurrent.futures import ProcessPoolExecutor
from datetime import datetime

import numpy as np
import torch

from . import crops, inference, parallel, pnp


class SubmissionWriter:
    """submission.py:6-52."""

    def __init__(self):
        self.test_results = []
        self.real_test_results = []

    def _append(self, filename, q, r, real):
        (self.real_test_results if real else self.test_results).append(
            {'filename': filename, 'q': list(q), 'r': list(r)})

    def append_test(self, filename, q, r):
        self._append(filename, q, r, real=False)

    def append_real_test(self, filename, q, r):
        self._append(filename, q, r, real=True)

    def export(self, out_dir='', suffix=None):
        sorted_test = sorted(self.test_results, key=lambda k: k['filename'])
        sorted_real_test = sorted(self.real_test_results, key=lambda k: k['filename'])
        if suffix is None:
            suffix = datetime.now().strftime("%Y%m%d-%H%M")
        submission_path = os.path.join(out_dir, 'submission_{}.csv'.format(suffix))
        with open(submission_path, 'w') as f:
            w = csv.writer(f, lineterminator='\n')
            for result in (sorted_test + sorted_real_test):
                w.writerow([result['filename'], *(result['q'] + result['r'])])
        return submission_path


def _blas_single_thread():
    """The PnP stage is thousands of 6x6 .. 12x12 LAPACK calls: a multi-threaded BLAS spends its time waking
    threads (measured 62 ms vs 4.5 ms per image with OpenBLAS on 8 cores), so it runs single-threaded."""
    try:
        from threadpoolctl import threadpool_limits
        return threadpool_limits(limits=1)
    except Exception:                                   # noqa: BLE001 - optional dependency
        return contextlib.nullcontext()


def _pose_job(args):
    kp, kp3d, K, xy, rate, thresh, min_k = args
    with _blas_single_thread():
        q, t, _ = pnp.keypoints_to_pose(kp, kp3d, K, xy, rate, thresh=thresh, min_k=min_k)
    return q, t


def poses_from_keypoints(kp, boxes, rates, kp3d, K, thresh: float = 0.8, min_k: int = 24, pool=None,
                         native: bool = True, threads: int = 0):
    """Host stage of val.py:172-224 for a batch: kp [N,K,3] (numpy) -> list of (q [w,x,y,z], t).
    native=True: the C++ solver of the library (`esahrnet_pnp_batch`, `threads` worker threads, ~100 us per image
    and thread); native=False: the numpy restatement it is tested against (optionally over a process `pool`)."""
    K = np.asarray(K, np.float64)
    if native:
        q, t = pnp.keypoints_to_pose_batch(kp, kp3d, K, [(b[0], b[1]) for b in boxes], rates, thresh, min_k, threads)
        return [(q[i], t[i]) for i in range(len(boxes))]
    jobs = [(kp[i], kp3d, K, (boxes[i][0], boxes[i][1]), rates[i], thresh, min_k) for i in range(len(boxes))]
    if pool is not None:
        return list(pool.map(_pose_job, jobs, chunksize=max(1, len(jobs) // 32)))
    return [_pose_job(j) for j in jobs]


def pose_pool(workers: int):
    """Process pool for the host PnP stage (images are independent)."""
    return ProcessPoolExecutor(max_workers=workers)


def estimate_poses(net, frames: torch.Tensor, bboxes, kp3d, K, scale: int = 256, thresh: float = 0.8,
                   min_k: int = 24, distributed: bool = False, pool=None, native: bool = True):
    """One batch of the val.py:136-233 loop.  frames uint8 cuda [N,H,W]; bboxes N x (x, y, x2, y2);
    kp3d [K3, 3] model keypoints; K camera matrix.  -> list of (q [w,x,y,z], t) per image."""
    x, boxes, rates = crops.crop_batch(frames, bboxes, scale)
    with torch.no_grad():
        if distributed:
            kp = parallel.sharded_keypoints(net, x)
        else:
            kp = inference.heatmaps_to_keypoints(net(x))
    kp = kp.cpu().numpy()                                   # the only device->host copy: N*K*3 floats
    return poses_from_keypoints(kp, boxes, rates, kp3d, K, thresh, min_k, pool, native)


def run_submission(net, batches, kp3d, K, writer: SubmissionWriter, real: bool = False, **kw):
    """`batches` yields (names, frames_u8_cuda, bboxes); appends every pose to `writer`."""
    for names, frames, bboxes in batches:
        for name, (q, t) in zip(names, estimate_poses(net, frames, bboxes, kp3d, K, **kw)):
            (writer.append_real_test if real else writer.append_test)(name, q, t)
    return writer
