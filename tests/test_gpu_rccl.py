"""RCCL on the hardware at hand: the path's only exchange — the all-gather of the per-rank keypoints
 — executed by the `nccl` backend (= RCCL on ROCm) with world size 1 on cuda:0, through the very functions
bench.py's N > 1 branch uses (parallel.make_sharded_step / timed_steps / gather_keypoints / sharded_keypoints).  The
multi-rank logic (uneven tails, empty shards) is covered under gloo in tests/test_parallel_gloo.py; an 8-GPU node is
only available to the driver."""
import os

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


def test_rccl_world_size_1_drives_the_sharded_step():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU")
    from esa_pose_estimation_amd import config, inference, parallel, seg_hrnet2, synth
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        assert dist.get_backend() == "nccl"
        net = seg_hrnet2.get_seg_model(config.make_config(widths=(16, 32, 64, 128)))
        sd = synth.make_state_dict({k: v.shape for k, v in net.state_dict().items()}, seed=5)
        net.load_state_dict(sd)
        net = net.to(dev).eval()
        n = 6
        x = synth.make_crops(n, 1, 64, 64, seed=5).to(dev)
        with torch.no_grad():
            ref = inference.heatmaps_to_keypoints(net(x))
            # (1) the library entry point: shard (the whole batch at world size 1) + RCCL all_gather_into_tensor
            kp = parallel.sharded_keypoints(net, x)
            assert kp.is_cuda and kp.shape == (n, 11, 3) and torch.equal(kp, ref)
            # (2) bench.py's N > 1 loop: local step -> gather, timed with barrier + MAX-over-ranks all_reduce on the GPU
            step = parallel.make_sharded_step(lambda: inference.heatmaps_to_keypoints(net(x)), n)
            elapsed, out = parallel.timed_steps(step, steps=3, warmup=1, sync=torch.cuda.synchronize, device=dev)
            assert elapsed > 0 and torch.equal(out, ref)
            # (3) the collective itself on a device tensor that is NOT the net's output
            t = torch.arange(n * 11 * 3, dtype=torch.float32, device=dev).view(n, 11, 3)
            assert torch.equal(parallel.gather_keypoints(t, n), t)
            dist.barrier()
            s = torch.tensor([3.5], dtype=torch.float64, device=dev)
            dist.all_reduce(s, op=dist.ReduceOp.MAX)
            assert float(s) == 3.5
    finally:
        dist.destroy_process_group()
