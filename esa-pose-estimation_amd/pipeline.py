"""Batch-N evaluation / submission driver (SURVEY.md §8f NEXT-4).

Reference being mirrored: the per-image loop of val.py:136-233 (batch 1, >=150 blocking .item() reads
per image) and submission.py:6-52.  Here: frames and detector boxes in, poses out, N crops at a time:
  crops.crop_batch -> net -> inference.heatmaps_to_keypoints -> (parallel.gather_keypoints) -> ONE D2H
  copy of [N, K, 3] -> host: top-k, back-projection, EPnP + RANSAC, peak-weighted refinement, quaternion.
`SubmissionWriter` keeps the reference's API and CSV format (filename, q0..q3, r0..r2, sorted by name).
"""
from __future__ import annotations

import csv
import os
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


def estimate_poses(net, frames: torch.Tensor, bboxes, kp3d, K, scale: int = 256, thresh: float = 0.8,
                   min_k: int = 24, distributed: bool = False):
    """One batch of the val.py:136-233 loop.  frames uint8 cuda [N,H,W]; bboxes N x (x, y, x2, y2);
    kp3d [K3, 3] model keypoints; K camera matrix.  -> list of (q [w,x,y,z], t) per image."""
    x, boxes, rates = crops.crop_batch(frames, bboxes, scale)
    with torch.no_grad():
        if distributed:
            kp = parallel.sharded_keypoints(net, x)
        else:
            kp = inference.heatmaps_to_keypoints(net(x))
    kp = kp.cpu().numpy()                                   # the only device->host copy: N*K*3 floats
    out = []
    for i in range(len(boxes)):
        q, t, _ = pnp.keypoints_to_pose(kp[i], kp3d, np.asarray(K, np.float64), (boxes[i][0], boxes[i][1]),
                                        rates[i], thresh=thresh, min_k=min_k)
        out.append((q, t))
    return out


def run_submission(net, batches, kp3d, K, writer: SubmissionWriter, real: bool = False, **kw):
    """`batches` yields (names, frames_u8_cuda, bboxes); appends every pose to `writer`."""
    for names, frames, bboxes in batches:
        for name, (q, t) in zip(names, estimate_poses(net, frames, bboxes, kp3d, K, **kw)):
            (writer.append_real_test if real else writer.append_test)(name, q, t)
    return writer
