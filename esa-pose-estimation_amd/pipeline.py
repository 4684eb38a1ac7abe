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
import logging
import os
from concurrent.futures import ProcessPoolExecutor
from datetime import datetime

import numpy as np
import torch

from . import crops, inference, parallel, pnp

logger = logging.getLogger(__name__)


class SubmissionWriter:
    """Collects (filename, q, r) rows and writes the ESA submission CSV.

    Same interface and same file, byte for byte, as the reference's writer (submission.py:6-52; pinned by
    tests/golden/submission_*.csv, which tests/golden/make_submission_golden.py produced by running the
    reference's class): one row `filename,q0,q1,q2,q3,r0,r1,r2` per image, the synthetic test set first and the
    real test set after it, each sorted by filename (stable, so duplicates keep their insertion order), values
    written with csv's default str() formatting, '\n' line ends.  `run_submission` only needs the two append
    methods, so the reference's own writer object can be passed in its place."""

    def __init__(self):
        self._rows = {False: [], True: []}          # real? -> [(filename, [q0..q3, r0..r2])]

    # the reference exposes the two lists; keep them readable for callers that look at them
    @property
    def test_results(self):
        return [{'filename': f, 'q': v[:4], 'r': v[4:]} for f, v in self._rows[False]]

    @property
    def real_test_results(self):
        return [{'filename': f, 'q': v[:4], 'r': v[4:]} for f, v in self._rows[True]]

    def append_test(self, filename, q, r):
        self._rows[False].append((filename, list(q) + list(r)))

    def append_real_test(self, filename, q, r):
        self._rows[True].append((filename, list(q) + list(r)))

    def export(self, out_dir='', suffix=None):
        if suffix is None:
            suffix = datetime.now().strftime("%Y%m%d-%H%M")
        path = os.path.join(out_dir, f'submission_{suffix}.csv')
        with open(path, 'w') as f:
            out = csv.writer(f, lineterminator='\n')
            for real in (False, True):
                for name, values in sorted(self._rows[real], key=lambda row: row[0]):
                    out.writerow([name, *values])
        return path


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


FALLBACK_POSE = ((1.0, 0.0, 0.0, 0.0), (0.0, 0.0, 10.0))   # identity attitude, 10 m down the boresight


class PoseFailure(ValueError):
    """No pose for an image (fewer than 4 usable keypoints, or RANSAC found no consensus)."""


def estimate_poses(net, frames: torch.Tensor, bboxes, kp3d, K, scale: int = 256, thresh: float = 0.8,
                   min_k: int = 24, distributed: bool = False, pool=None, native: bool = True,
                   on_fail: str = "raise"):
    """One batch of the val.py:136-233 loop.  frames uint8 cuda [N,H,W]; bboxes N x (x, y, x2, y2);
    kp3d [K3, 3] model keypoints; K camera matrix.  -> list of (q [w,x,y,z], t) per image.
    An image without a solution (the native solver reports it as a NaN row; the reference would die inside
    cv2.solvePnPRansac) raises PoseFailure, or with on_fail="nan" is returned as the NaN row for the caller to
    deal with — it is never passed on silently."""
    x, boxes, rates = crops.crop_batch(frames, bboxes, scale)
    with torch.no_grad():
        if distributed:
            kp = parallel.sharded_keypoints(net, x)
        else:
            kp = inference.heatmaps_to_keypoints(net(x))
    kp = kp.cpu().numpy()                                   # the only device->host copy: N*K*3 floats
    poses = poses_from_keypoints(kp, boxes, rates, kp3d, K, thresh, min_k, pool, native)
    if on_fail == "raise":
        bad = [i for i, (q, t) in enumerate(poses) if not (np.all(np.isfinite(q)) and np.all(np.isfinite(t)))]
        if bad:
            raise PoseFailure(f"no pose for batch positions {bad}")
    return poses


def run_submission(net, batches, kp3d, K, writer, real: bool = False, on_fail: str = "fallback", **kw):
    """`batches` yields (names, frames_u8_cuda, bboxes); appends every pose to `writer` (ours or the reference's
    SubmissionWriter: anything with append_test / append_real_test).  A submission needs a finite row for every
    image, so an image without a solution gets FALLBACK_POSE and is logged and listed in `writer.failed`
    (on_fail="fallback"), or stops the run (on_fail="raise")."""
    failed = []
    for names, frames, bboxes in batches:
        poses = estimate_poses(net, frames, bboxes, kp3d, K, on_fail="nan", **kw)
        for name, (q, t) in zip(names, poses):
            if not (np.all(np.isfinite(q)) and np.all(np.isfinite(t))):
                if on_fail == "raise":
                    raise PoseFailure(f"no pose for {name}")
                logger.warning("no pose for %s: writing the fallback pose", name)
                failed.append(name)
                q, t = FALLBACK_POSE
            (writer.append_real_test if real else writer.append_test)(name, q, t)
    try:
        writer.failed = getattr(writer, "failed", []) + failed
    except AttributeError:
        pass
    return writer
