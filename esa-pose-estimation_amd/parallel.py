"""Data-parallel crop sharding over the GPUs of one node (one process per GPU, RCCL/xGMI).

The reference's only multi-GPU mechanism is single-process nn.DataParallel, which gathers the
full [N,K,H,W] heatmaps onto GPU 0 (main.py:254, val.py:382; 92 MB per rank at batch 32).  Crops
are independent through the whole path, so here every rank runs its contiguous
slice of the batch and the ONLY exchange is one all-gather of the per-rank [n_local, K, 3]
keypoints (132 B per crop at K=11) before the host-side PnP — latency-bound, no ring tuning.
Uneven tails are padded to the largest shard and trimmed after the gather.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_bounds(n_total: int, world_size: int, rank: int):
    """Contiguous split; the first (n_total % world_size) ranks get one extra crop."""
    base, extra = divmod(n_total, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_keypoints(local_kp: torch.Tensor, n_total: int, group=None) -> torch.Tensor:
    """local_kp [n_local, K, 3] (this rank's shard, rank order = crop order) -> [n_total, K, 3]
    on every rank.  One all_gather (RCCL on GPU tensors, gloo on CPU tensors)."""
    if not (dist.is_available() and dist.is_initialized()):
        if local_kp.shape[0] != n_total:
            raise ValueError("no process group but shard size != batch size")
        return local_kp
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lo, hi = shard_bounds(n_total, world, rank)
    if local_kp.shape[0] != hi - lo:
        raise ValueError(f"rank {rank}: shard has {local_kp.shape[0]} crops, expected {hi - lo}")
    n_max = -(-n_total // world)
    k = local_kp.shape[1]
    padded = local_kp.new_zeros((n_max, k, 3))
    padded[: hi - lo] = local_kp
    flat = local_kp.new_empty((world * n_max, k, 3))
    dist.all_gather_into_tensor(flat, padded.contiguous(), group=group)
    out = flat.view(world, n_max, k, 3)
    parts = []
    for r in range(world):
        a, b = shard_bounds(n_total, world, r)
        parts.append(out[r, : b - a])
    return torch.cat(parts, 0)


def sharded_keypoints(net, crops: torch.Tensor, group=None, keypoints_fn=None) -> torch.Tensor:
    """Every rank holds (or can index) the full batch `crops` [N,Cin,H,W]; each runs its slice
    through `net` + the fused keypoint kernel and all ranks return the full [N,K,3].
    `keypoints_fn` (default: inference.heatmaps_to_keypoints, GPU only) maps the rank's heat-maps to [n,K,3]."""
    if keypoints_fn is None:
        from .inference import heatmaps_to_keypoints as keypoints_fn
    heatmaps_to_keypoints = keypoints_fn
    n_total = crops.shape[0]
    if dist.is_available() and dist.is_initialized():
        lo, hi = shard_bounds(n_total, dist.get_world_size(group), dist.get_rank(group))
    else:
        lo, hi = 0, n_total
    k = net.num_keypoints
    if hi > lo:
        kp = heatmaps_to_keypoints(net(crops[lo:hi]))
    else:
        kp = crops.new_zeros((0, k, 3))
    return gather_keypoints(kp, n_total, group)


# ---- the measurement protocol of bench.py, importable so that the N > 1 branch runs under gloo on the CPU too ----
def make_sharded_step(local_step, n_total: int, group=None):
    """step() = this rank's shard through `local_step` (eager forward + keypoints, or a HIP-graph replay that
    returns its static output tensor), then the path's one exchange: the keypoint all-gather."""
    def step():
        return gather_keypoints(local_step(), n_total, group)
    return step


def timed_steps(step, steps: int, warmup: int, sync=None, device=None, group=None):
    """W untimed steps, then exactly K steps bracketed by (device sync + barrier) on both sides; returns
    (elapsed seconds = MAX over ranks, output of the last step).  `sync`: callable draining the device
    (torch.cuda.synchronize on a GPU, nothing on the CPU)."""
    import time
    sync = sync or (lambda: None)
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    out = None
    for _ in range(warmup):
        out = step()
    sync()
    if multi:
        dist.barrier(group)
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    sync()
    if multi:
        dist.barrier(group)
    elapsed = time.perf_counter() - t0
    if multi:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        elapsed = float(t.item())
    return elapsed, out
