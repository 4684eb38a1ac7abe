#!/usr/bin/env python3
"""bench.py — the reference's headline metric on MI355X: HRNet-W32 256x256 crops/sec.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one pass of the hot path over one batch of synthetic SPEED-shaped crops already
resident in HBM:  x [B,1,256,256] f32 -> HRNet forward (seg_hrnet2 topology, widths 32/64/128/256)
-> [B,11,256,256] heatmaps -> fused arg-max + sub-pixel refine -> [B,11,3] keypoints
(+ for N>1 the RCCL all-gather of the keypoints, the path's only exchange).  B = 32 per GPU
(BASELINE.json configs[1]); weak scaling: N GPUs process N*32 crops per step (configs[2] at N=8).

One JSON line on stdout (rank 0), carrying also
  "roofline":     dominant kernel (3x3 stride-1 split-bf16 MFMA convolution) — algorithmic FLOPs
                  per launch / HIP-event duration per launch, against the bf16x3 MFMA peak;
  "cpu_baseline": the CPU oracle (torch-CPU restatement of the reference forward + numpy
                  post-processing) timed on this box's host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

# gfx950 dense peaks (/opt/skills/guides/MI355X_MICROARCH.md): bf16 MFMA ~2.5 PFLOP/s, f32 157.3 TFLOP/s.
# The split-bf16 scheme issues 3 bf16 MFMA FLOPs per algorithmic FLOP, so the ceiling for
# ALGORITHMIC FLOP/s of the convolution kernels is 2500/3 TFLOP/s.
PEAK_BF16_TFLOPS = 2500.0
PEAK_BF16X3_TFLOPS = PEAK_BF16_TFLOPS / 3.0
PEAK_F32_TFLOPS = 157.3


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="crops per GPU per step")
    ap.add_argument("--hw", type=int, default=256)
    ap.add_argument("--variant", default="seg_hrnet2", choices=["seg_hrnet2", "seg_hrnet"])
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of HIP-graph replay")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU-baseline time budget")
    ap.add_argument("--profile-steps", type=int, default=3, help="instrumented forwards for the roofline leg")
    return ap.parse_args()


def host_cores() -> int:
    """CPU threads this process may really use: affinity mask, capped by the cgroup CPU quota
    (a GPU box hands each job a share of the host, e.g. 16 of 256 hardware threads) and, when no
    quota is readable, by 16 — a batch-1 CNN forward does not scale past that anyway."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    quota = int(txt[0]) / int(txt[1])
            else:
                q = int(txt[0])
                if q > 0:
                    quota = q / int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            break
        except Exception:
            continue
    if quota:
        n = min(n, max(1, int(quota + 0.5)))
    else:
        n = min(n, 16)
    return n


def cpu_baseline(sd, cfg, variant, hw, budget_s):
    """Oracle (kind 'port') on the host cores: batch-1 forwards + numpy post-processing."""
    from esa_pose_estimation_amd import synth
    from oracle import hrnet_ref, keypoints_ref
    cores = host_cores()
    torch.set_num_threads(cores)
    x = synth.make_crops(1, cfg["cin"], hw, hw, seed=0)

    def one():
        with torch.no_grad():
            y = hrnet_ref.forward(sd, cfg, x)
        keypoints_ref.heatmaps_to_keypoints(y.numpy())

    t_w = time.perf_counter()
    for _ in range(2):
        one()
        if time.perf_counter() - t_w > budget_s:       # pathological host: keep the run bounded
            break
    times = []
    t_end = time.perf_counter() + budget_s
    while time.perf_counter() < t_end and len(times) < 200:
        t0 = time.perf_counter()
        one()
        times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    return {"value": round(1.0 / med, 3), "unit": "crops/s", "cores": cores, "kind": "port",
            "sample": f"{len(times)} batch-1 forwards of {variant} W32 {hw}x{hw} + numpy arg-max/refine "
                      f"(oracle/hrnet_ref.py, torch {torch.__version__} CPU, median {med * 1e3:.1f} ms/crop)"}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit(f"--gpus {args.gpus} needs torch.distributed.run with --nproc-per-node {args.gpus}")
        args.gpus = world
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)

    from esa_pose_estimation_amd import config, inference, parallel, synth
    import esa_pose_estimation_amd as pkg
    mod = getattr(__import__("esa_pose_estimation_amd." + args.variant), args.variant)
    net = mod.get_seg_model(config.make_config())
    sd = synth.make_state_dict({k: v.shape for k, v in net.state_dict().items()}, seed=0)
    net.load_state_dict(sd, strict=True)
    net = net.to(dev).eval()
    cin, K = net._cin, net.num_keypoints
    B, hw = args.batch, args.hw
    n_total = B * world
    # every rank generates only its own shard of the global synthetic batch
    x = synth.make_crops(B, cin, hw, hw, seed=1000 + rank).to(dev)

    def local_step():
        heat = net(x)
        return inference.heatmaps_to_keypoints(heat)

    with torch.no_grad():
        kp = local_step()                      # folds + uploads weights, allocates workspace
        torch.cuda.synchronize()
        graph = None
        if not args.no_graph:
            try:
                s = torch.cuda.Stream()
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    local_step()
                torch.cuda.current_stream().wait_stream(s)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    kp = local_step()
            except Exception as e:             # noqa: BLE001
                print(f"[bench] HIP graph capture failed ({e}); running eager", file=sys.stderr)
                graph = None

        def step():
            if graph is not None:
                graph.replay()
                out = kp
            else:
                out = local_step()
            if world > 1:
                return parallel.gather_keypoints(out, n_total)
            return out

        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            allkp = step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        assert allkp.shape == (n_total, K, 3) and bool(torch.isfinite(allkp).all())

        # ---- roofline leg (rank 0): per-launch HIP-event durations of instrumented forwards ----
        roof = None
        breakdown = None
        if rank == 0:
            per = None
            for _ in range(max(1, args.profile_steps)):
                _, ops = net.forward_timed(x)
                if per is None:
                    per = [dict(o, ms=0.0) for o in ops]
                for a, o in zip(per, ops):
                    a["ms"] += o["ms"] / max(1, args.profile_steps)
            groups = {}
            for o in per:
                gk = groups.setdefault(o["kernel"], dict(launches=0, ms=0.0, flops=0.0, bytes=0.0))
                gk["launches"] += 1
                gk["ms"] += o["ms"]
                gk["flops"] += o["flops"]
                gk["bytes"] += o["bytes"]
            # dominant kernel = the matrix-core kernel with the largest share of the step
            dominant = max((k for k, v in groups.items() if v["flops"] > 0), key=lambda k: groups[k]["ms"])
            d = groups[dominant]
            ach = d["flops"] / (d["ms"] * 1e-3) / 1e12
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tpath):
                try:
                    traffic = json.load(open(tpath)).get(dominant)
                except Exception:
                    traffic = None
            roof = {"bound": "mfma", "kernel": dominant, "achieved": round(ach, 2),
                    "peak": round(PEAK_BF16X3_TFLOPS, 1), "unit": "TFLOP/s",
                    "frac": round(ach / PEAK_BF16X3_TFLOPS, 4), "traffic": traffic,
                    "launches_per_step": d["launches"],
                    "avg_launch_us": round(d["ms"] * 1e3 / d["launches"], 2),
                    "algorithmic_gflop_per_launch": round(d["flops"] / d["launches"] / 1e9, 3),
                    "frac_of_f32_peak": round(ach / PEAK_F32_TFLOPS, 4),
                    "peak_note": "split-bf16: 3 bf16 MFMA FLOPs per algorithmic FLOP -> ceiling 2500/3 TFLOP/s"}
            tot_ms = sum(v["ms"] for v in groups.values())
            breakdown = {k: {"launches": v["launches"], "ms": round(v["ms"], 4),
                             "share": round(v["ms"] / tot_ms, 4),
                             "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2) if v["ms"] > 0 else 0.0,
                             "gbps": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1) if v["ms"] > 0 else 0.0}
                         for k, v in groups.items()}

    cpu = None
    if rank == 0 and not args.no_cpu_baseline:
        from oracle import hrnet_ref
        cpu = cpu_baseline(sd, hrnet_ref.default_cfg(cin, K), args.variant, hw, args.cpu_seconds)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = n_total * args.steps / elapsed
        flops_crop = net.flops_per_crop(hw, hw)
        line = {
            "metric": "HRNet-W32 256x256 crops/sec (heatmaps + fused argmax/refine keypoints)",
            "value": round(value, 1), "unit": "crops/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16x3 (split-bf16 MFMA, f32 accumulate; f32 VALU stem/head)",
            "data": "synthetic",
            "config": {"workload": f"{args.variant} HRNet-W32 {hw}x{hw}, batch {B}/GPU, {K} keypoints, "
                                   f"fp32 NCHW in -> heatmaps -> keypoints (BASELINE configs[{1 if world == 1 else 2}])",
                       "global_batch": n_total, "per_gpu_batch": B, "parallelism": f"dp{world}",
                       "hip_graph": graph is not None,
                       "algorithmic_gflop_per_crop": round(flops_crop / 1e9, 3),
                       "whole_net_algorithmic_tflops": round(value * flops_crop / 1e12 / world, 2)},
            "roofline": roof, "cpu_baseline": cpu, "kernel_breakdown": breakdown,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
