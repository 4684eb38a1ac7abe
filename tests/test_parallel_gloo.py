"""CPU, world_size 2, gloo: the N>1 path of the keypoint exchange (contiguous crop sharding,
padded all-gather, trim) — the only collective on the path."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from esa_pose_estimation_amd import parallel


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, k, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = parallel.shard_bounds(n_total, world, rank)
        full = torch.arange(n_total * k * 3, dtype=torch.float32).reshape(n_total, k, 3)
        got = parallel.gather_keypoints(full[lo:hi].clone(), n_total)
        ok = torch.equal(got, full)

        class FakeNet:                      # stands in for the GPU model: crops -> per-crop "heatmaps"
            num_keypoints = k

        # sharded_keypoints' sharding arithmetic (without the GPU kernels): emulate its body
        kp = full[lo:hi] * 2
        ok = ok and torch.equal(parallel.gather_keypoints(kp, n_total), full * 2)
        q.put((rank, bool(ok), lo, hi))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [8, 7, 1])
def test_gather_keypoints_world2(n_total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, 11, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _, _ in res)
    spans = sorted((lo, hi) for _, _, lo, hi in res)
    assert spans[0][0] == 0 and spans[-1][1] == n_total and spans[0][1] == spans[1][0]


def test_shard_bounds_cover_the_batch():
    for n in (1, 5, 32, 255, 256):
        for w in (1, 2, 3, 8):
            cuts = [parallel.shard_bounds(n, w, r) for r in range(w)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))
            sizes = [b - a for a, b in cuts]
            assert max(sizes) - min(sizes) <= 1


def test_single_process_passthrough():
    kp = torch.zeros(4, 11, 3)
    assert parallel.gather_keypoints(kp, 4) is kp
    with pytest.raises(ValueError):
        parallel.gather_keypoints(kp, 5)
