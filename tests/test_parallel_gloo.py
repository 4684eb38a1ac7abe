"""CPU, world_size 2, gloo: the N>1 path of the keypoint exchange (contiguous crop sharding,
padded all-gather, trim) — the only collective on the path — through the SAME functions
the GPU path runs: parallel.sharded_keypoints (with a stub net and a stub keypoint function, the HIP
kernels being GPU-only) and bench.py's measurement loop (parallel.make_sharded_step / timed_steps)."""
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


class StubNet:
    """Stands in for the GPU model: crops [n,1,H,W] -> 'heat-maps' [n,K,H,W] that encode (crop id, k)."""
    def __init__(self, k):
        self.num_keypoints = k
        self.calls = []

    def __call__(self, x):
        self.calls.append(int(x.shape[0]))
        ks = torch.arange(self.num_keypoints, dtype=torch.float32).view(1, -1, 1, 1)
        return x[:, :1] * 100.0 + ks                      # value = 100 * crop id + k everywhere


def _stub_keypoints(heat):
    v = heat[:, :, 0, 0]
    return torch.stack([v, v + 0.25, v + 0.5], dim=2)     # [n,K,3]


def _expected(n_total, k):
    ids = torch.arange(n_total, dtype=torch.float32).view(-1, 1) * 100.0 + torch.arange(k, dtype=torch.float32).view(1, -1)
    return torch.stack([ids, ids + 0.25, ids + 0.5], dim=2)


def _worker(rank, world, port, n_total, k, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = parallel.shard_bounds(n_total, world, rank)
        full = torch.arange(n_total * k * 3, dtype=torch.float32).reshape(n_total, k, 3)
        ok = torch.equal(parallel.gather_keypoints(full[lo:hi].clone(), n_total), full)
        # the product function itself: every rank holds the whole batch, runs its slice, all get everything
        crops = torch.arange(n_total, dtype=torch.float32).view(-1, 1, 1, 1).expand(n_total, 1, 4, 4).contiguous()
        net = StubNet(k)
        got = parallel.sharded_keypoints(net, crops, keypoints_fn=_stub_keypoints)
        ok = ok and torch.equal(got, _expected(n_total, k))
        ok = ok and net.calls == ([hi - lo] if hi > lo else [])          # an empty shard never calls the net
        # bench.py's N > 1 loop: a "graph replay" (returns a static tensor) followed by the all-gather
        static = _stub_keypoints(net(crops[lo:hi])) if hi > lo else crops.new_zeros((0, k, 3))
        replays = []

        def local_step():
            replays.append(1)
            return static
        step = parallel.make_sharded_step(local_step, n_total)
        elapsed, out = parallel.timed_steps(step, steps=3, warmup=2, device=torch.device("cpu"))
        ok = ok and torch.equal(out, _expected(n_total, k)) and len(replays) == 5 and elapsed > 0
        # MAX over ranks: every rank reports the same elapsed time
        t = torch.tensor([elapsed], dtype=torch.float64)
        both = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(both, t)
        ok = ok and all(float(b) == float(t) for b in both)
        q.put((rank, bool(ok), lo, hi))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [8, 7, 1])          # even split, uneven tail, an EMPTY shard on rank 1
def test_sharded_keypoints_world2(n_total):
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
    # the measurement loop without a process group: plain timing, no collective
    n = []
    elapsed, out = parallel.timed_steps(parallel.make_sharded_step(lambda: (n.append(1), kp)[1], 4), steps=4, warmup=1)
    assert out is kp and len(n) == 5 and elapsed >= 0
    net = StubNet(11)
    crops = torch.arange(3, dtype=torch.float32).view(-1, 1, 1, 1).expand(3, 1, 2, 2).contiguous()
    assert torch.equal(parallel.sharded_keypoints(net, crops, keypoints_fn=_stub_keypoints), _expected(3, 11))
