#!/usr/bin/env python3
"""bench.py — the reference's headline metric on MI355X: HRNet-W32 256x256 crops/sec.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one pass of the hot path over one batch of synthetic SPEED-shaped crops already
resident in HBM:  x [B,1,256,256] f32 -> HRNet forward (seg_hrnet2 topology, widths 32/64/128/256)
-> [B,11,256,256] heatmaps -> fused arg-max + sub-pixel refine -> [B,11,3] keypoints
(+ for N>1 the RCCL all-gather of the keypoints, the path's only exchange).  B = 32 per GPU
(BASELINE.json configs[1]); weak scaling: N GPUs process N*32 crops per step (configs[2] at N=8).
Arithmetic: the fp32-grade mode (precision "fp32" = bf16x6, include/esahrnet.h) — configs[1] says fp32; the faster
split-bf16 mode is an `extras` line, never `value`.

One JSON line on stdout (rank 0), carrying also
  "roofline":     dominant kernel (the 3x3 stride-1 bf16x6 MFMA convolution, conv_x6.hip) — algorithmic FLOPs
                  per launch / HIP-event duration per launch, against the bf16x6 ceiling 2500/6 TFLOP/s;
  "cpu_baseline": the CPU oracle (torch-CPU restatement of the reference forward + numpy
                  post-processing) timed on this box's host cores on a bounded sample: batch 1 on all
                  cores (the headline row), batch 1 on one thread and batch 32 on all cores;
  "extras" (N = 1 only, all measured AFTER the timed region, none of them is `value`):
      latency_b1_us        one crop, graph-replayed and eager (the reference calls the net per image, val.py:112)
      sustained            the same step over >= 2000 back-to-back replays (clocks settle under load)
      config3_w48_384_bf16 BASELINE.json configs[3]: HRNet-W48 384x384 batch 64 in the single-pass bf16 mode,
                           with its own roofline block (MFMA and HBM fractions of its dominant kernel)
      seg_hrnet3           the CBAM network val.py:380 instantiates, batch 32 256x256, with its own roofline block.

Other workloads as the headline of an explicit run (never what the driver runs):
    python bench.py --workload w48-bf16        (configs[3])
    python bench.py --variant seg_hrnet3
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

# gfx950 dense peaks (/opt/skills/guides/MI355X_MICROARCH.md): bf16 MFMA ~2.5 PFLOP/s, f32 157.3 TFLOP/s, HBM3E 8 TB/s.
# The split-bf16 scheme issues 3 bf16 MFMA FLOPs per algorithmic FLOP, so the ceiling for
# ALGORITHMIC FLOP/s of the convolution kernels is 2500/3 TFLOP/s; the single-pass bf16 mode prices against 2500.
PEAK_BF16_TFLOPS = 2500.0
PEAK_BF16X3_TFLOPS = PEAK_BF16_TFLOPS / 3.0
# fp32-grade mode ("fp32" = bf16x6: exact 3-term bf16 split, 6 MFMA FLOPs per algorithmic FLOP)
PEAK_BF16X6_TFLOPS = PEAK_BF16_TFLOPS / 6.0
DTYPES = {
    "fp32": "bf16x6 (fp32-grade: every f32 operand split exactly into 3 bf16 terms, 6 bf16 MFMAs per product, f32 accumulate; "
            "f32 NHWC activations; f32 VALU stem conv1 / output layer) — heat-maps closer to fp64 than the fp32 CPU reference",
    "bf16x3": "bf16x3 (split-bf16 MFMA, ~16 significand bits per operand, f32 accumulate; f32 VALU stem/head) — NOT fp32",
    "bf16": "bf16 (single-pass bf16 MFMA, f32 accumulate, f32 bias epilogue; f32 VALU stem conv1 / output layer)",
}
PEAKS = {"fp32": PEAK_BF16X6_TFLOPS, "bf16x3": PEAK_BF16X3_TFLOPS, "bf16": PEAK_BF16_TFLOPS}
PEAK_NOTES = {
    "fp32": "bf16x6: 6 bf16 MFMA FLOPs per algorithmic FLOP -> ceiling 2500/6 = 416.7 TFLOP/s (2.65 x the f32 vector/matrix peak)",
    "bf16x3": "split-bf16: 3 bf16 MFMA FLOPs per algorithmic FLOP -> ceiling 2500/3 TFLOP/s",
    "bf16": ("single-pass bf16: one MFMA FLOP per algorithmic FLOP -> ceiling 2500 TFLOP/s; layer-by-layer bf16 "
             "is HBM-bound on the wide-resolution branches, hence both fractions (compulsory bytes / 8 TB/s)"),
}
PEAK_F32_TFLOPS = 157.3
PEAK_HBM_GBPS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=None, help="crops per GPU per step (default: 32; 64 for w48-bf16)")
    ap.add_argument("--hw", type=int, default=None)
    ap.add_argument("--workload", default="w32", choices=["w32", "w48-bf16", "w48"],
                    help="w32: BASELINE configs[1] (headline); w48-bf16: configs[3]; w48: the same net in split-bf16")
    ap.add_argument("--variant", default="seg_hrnet2", choices=["seg_hrnet2", "seg_hrnet", "seg_hrnet3"])
    ap.add_argument("--precision", default=None, choices=["fp32", "bf16x3", "bf16"],
                    help="default: fp32 (fp32-grade bf16x6, BASELINE configs[1]); w48-bf16: bf16; seg_hrnet3: bf16x3")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of HIP-graph replay")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="time budget of the headline CPU-baseline row")
    ap.add_argument("--sustained-steps", type=int, default=2000)
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


def _time_oracle(sd, cfg, hw, batch, threads, budget_s, min_iters=3, max_iters=200):
    from esa_pose_estimation_amd import synth
    from oracle import hrnet_ref, keypoints_ref
    torch.set_num_threads(threads)
    x = synth.make_crops(batch, cfg["cin"], hw, hw, seed=0)

    def one():
        with torch.no_grad():
            y = hrnet_ref.forward(sd, cfg, x)
        keypoints_ref.heatmaps_to_keypoints(y.numpy())

    t_w = time.perf_counter()
    one()                                               # warm-up (bounded: a pathological host stops here)
    if time.perf_counter() - t_w < budget_s / 4:
        one()
    times = []
    t_end = time.perf_counter() + budget_s
    while (time.perf_counter() < t_end or len(times) < min_iters) and len(times) < max_iters:
        t0 = time.perf_counter()
        one()
        times.append(time.perf_counter() - t0)
        if len(times) >= min_iters and time.perf_counter() - t_w > 3 * budget_s:
            break
    med = float(np.median(times))
    return batch / med, med, len(times)


def cpu_baseline(sd, cfg, variant, hw, budget_s):
    """Oracle (kind 'port') on the host cores.  Headline row: batch-1 forwards + numpy post-processing on all
    cores; BASELINE.md §2 also asks for 1 thread and for batch 32 — shorter budgets, same protocol."""
    cores = host_cores()
    v, med, n = _time_oracle(sd, cfg, hw, 1, cores, budget_s)
    out = {"value": round(v, 3), "unit": "crops/s", "cores": cores, "kind": "port",
           "sample": f"{n} batch-1 forwards of {variant} W32 {hw}x{hw} + numpy arg-max/refine "
                     f"(oracle/hrnet_ref.py, torch {torch.__version__} CPU, median {med * 1e3:.1f} ms/crop)"}
    rows = []
    for batch, threads, budget in ((1, 1, budget_s / 2), (32, cores, budget_s / 2)):
        try:
            v, med, n = _time_oracle(sd, cfg, hw, batch, threads, budget, min_iters=2)
            rows.append({"batch": batch, "threads": threads, "value": round(v, 3), "unit": "crops/s",
                         "median_ms_per_forward": round(med * 1e3, 1), "forwards": n})
        except Exception as e:                          # noqa: BLE001 - a baseline row must not sink the bench line
            rows.append({"batch": batch, "threads": threads, "error": str(e)})
    out["rows"] = rows
    torch.set_num_threads(cores)
    return out


def build_net(variant, widths, precision, dev, seed=0):
    from esa_pose_estimation_amd import config, synth
    mod = getattr(__import__("esa_pose_estimation_amd." + variant), variant)
    net = mod.get_seg_model(config.make_config(widths=widths), precision=precision)
    sd = synth.make_state_dict({k: v.shape for k, v in net.state_dict().items()}, seed=seed)
    net.load_state_dict(sd, strict=True)
    return net.to(dev).eval(), sd


def capture(local_step, enabled=True):
    """HIP-graph capture of one local step (torch.cuda.graph); (None, None) when disabled or refused."""
    if not enabled:
        return None, None
    try:
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            local_step()
        torch.cuda.current_stream().wait_stream(s)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = local_step()
        return graph, out
    except Exception as e:                              # noqa: BLE001
        print(f"[bench] HIP graph capture failed ({e}); running eager", file=sys.stderr)
        return None, None


def roofline_leg(net, x, profile_steps, precision):
    """Per-launch HIP-event durations of instrumented forwards -> (roofline block, per-kernel breakdown)."""
    per = None
    for _ in range(max(1, profile_steps)):
        _, ops = net.forward_timed(x)
        if per is None:
            per = [dict(o, ms=0.0) for o in ops]
        for a, o in zip(per, ops):
            a["ms"] += o["ms"] / max(1, profile_steps)
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
    gbps = d["bytes"] / (d["ms"] * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get(dominant)
        except Exception:
            traffic = None
    peak = PEAKS[precision]
    roof = {"bound": "mfma", "kernel": dominant, "achieved": round(ach, 2), "peak": round(peak, 1),
            "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": traffic,
            "traffic_source": "profiles/traffic.json (committed rocprofv3 --pmc pass of this workload, HBM bytes per launch; "
                              "not measured in this run)" if traffic is not None else None,
            "launches_per_step": d["launches"], "avg_launch_us": round(d["ms"] * 1e3 / d["launches"], 2),
            "algorithmic_gflop_per_launch": round(d["flops"] / d["launches"] / 1e9, 3),
            "frac_of_f32_peak": round(ach / PEAK_F32_TFLOPS, 4),
            "algorithmic_gbps": round(gbps, 1), "frac_of_hbm_peak": round(gbps / PEAK_HBM_GBPS, 4),
            "peak_note": PEAK_NOTES[precision]}
    tot_ms = sum(v["ms"] for v in groups.values())
    breakdown = {k: {"launches": v["launches"], "ms": round(v["ms"], 4), "share": round(v["ms"] / tot_ms, 4),
                     "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2) if v["ms"] > 0 else 0.0,
                     "gbps": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1) if v["ms"] > 0 else 0.0}
                 for k, v in groups.items()}
    roof["whole_step_event_ms"] = round(tot_ms, 4)
    roof["launches_per_forward"] = len(per)
    return roof, breakdown


def side_workload(tag, variant, widths, precision, batch, hw, dev, steps, warmup, profile_steps):
    """A second workload measured with the same protocol (graph replay, K steps between syncs) for `extras`."""
    from esa_pose_estimation_amd import inference, parallel, synth
    net, _ = build_net(variant, widths, precision, dev)
    x = synth.make_crops(batch, net._cin, hw, hw, seed=2000).to(dev)

    def local_step():
        return inference.heatmaps_to_keypoints(net(x))

    with torch.no_grad():
        local_step()
        torch.cuda.synchronize()
        graph, out = capture(local_step)

        def step():
            if graph is not None:
                graph.replay()
                return out
            return local_step()
        elapsed, kp = parallel.timed_steps(step, steps, warmup, sync=torch.cuda.synchronize)
        assert bool(torch.isfinite(kp).all())
        roof, breakdown = roofline_leg(net, x, profile_steps, precision)
    flops_crop = net.flops_per_crop(hw, hw)
    value = batch * steps / elapsed
    res = {"workload": tag, "value": round(value, 1), "unit": "crops/s", "ms_per_step": round(elapsed / steps * 1e3, 4),
           "batch": batch, "steps": steps, "hip_graph": graph is not None, "launches": roof["launches_per_forward"],
           "dtype": DTYPES[precision],
           "algorithmic_gflop_per_crop": round(flops_crop / 1e9, 3),
           "whole_net_algorithmic_tflops": round(value * flops_crop / 1e12, 2),
           "roofline": roof, "kernel_breakdown": breakdown}
    del net, x, graph, out
    torch.cuda.empty_cache()
    return res


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

    from esa_pose_estimation_amd import inference, parallel, synth
    w48 = args.workload.startswith("w48")
    widths = (48, 96, 192, 384) if w48 else (32, 64, 128, 256)
    precision = args.precision or ("bf16" if args.workload == "w48-bf16" else "bf16x3" if w48 else "fp32")
    B = args.batch or (64 if w48 else 32)
    hw = args.hw or (384 if w48 else 256)
    net, sd = build_net(args.variant, widths, precision, dev)
    cin, K = net._cin, net.num_keypoints
    n_total = B * world
    # every rank generates only its own shard of the global synthetic batch
    x = synth.make_crops(B, cin, hw, hw, seed=1000 + rank).to(dev)

    def local_step():
        return inference.heatmaps_to_keypoints(net(x))

    with torch.no_grad():
        local_step()                           # folds + uploads weights, allocates workspace
        torch.cuda.synchronize()
        graph, kp = capture(local_step, not args.no_graph)

        def replay_or_eager():
            if graph is not None:
                graph.replay()
                return kp
            return local_step()

        step = parallel.make_sharded_step(replay_or_eager, n_total) if world > 1 else replay_or_eager
        elapsed, allkp = parallel.timed_steps(step, args.steps, args.warmup, sync=torch.cuda.synchronize, device=dev)
        assert allkp.shape == (n_total, K, 3) and bool(torch.isfinite(allkp).all())

        roof = breakdown = None
        if rank == 0:
            roof, breakdown = roofline_leg(net, x, args.profile_steps, precision)

        extras = None
        if rank == 0 and world == 1 and not args.no_extras and args.workload == "w32" and args.variant == "seg_hrnet2" and precision == "fp32":
            extras = {}
            # ---- batch-1 latency (val.py:112 calls the net once per image) --------------------------------------
            x1 = x[:1].clone()

            def step1():
                return inference.heatmaps_to_keypoints(net(x1))
            step1()
            torch.cuda.synchronize()
            g1, _ = capture(step1)
            lat = {}
            for name, fn in (("graph", (lambda: g1.replay()) if g1 is not None else None), ("eager", step1)):
                if fn is None:
                    continue
                for _ in range(20):
                    fn()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(200):
                    fn()
                    torch.cuda.synchronize()                # latency: the caller waits for every crop
                lat[name] = round((time.perf_counter() - t0) / 200 * 1e6, 1)
            host = []
            for _ in range(10):                             # few calls from an idle queue: no back-pressure from the GPU
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(4):
                    step1()
                host.append((time.perf_counter() - t0) / 4)
            lat["eager_host_enqueue"] = round(float(np.median(host)) * 1e6, 1)
            torch.cuda.synchronize()
            lat["note"] = ("one W32 256x256 crop, forward + keypoints, wall time per call including the wait for the "
                           "result; eager_host_enqueue = host time of one eager call (checks + ~90 launches)")
            extras["latency_b1_us"] = lat
            del g1
            # ---- sustained throughput -----------------------------------------------------------------------------
            if args.sustained_steps > 0:
                el, _ = parallel.timed_steps(replay_or_eager, args.sustained_steps, 10, sync=torch.cuda.synchronize)
                extras["sustained"] = {"value": round(B * args.sustained_steps / el, 1), "unit": "crops/s",
                                       "steps": args.sustained_steps,
                                       "ms_per_step": round(el / args.sustained_steps * 1e3, 4),
                                       "seconds": round(el, 2)}
            # ---- BASELINE configs[3] and the production network ----------------------------------------------------
            for key, a in (("w32_bf16x3_opt_in", ("the headline workload in the split-bf16 fast mode (precision='bf16x3': NOT fp32; "
                                                  "heat-map error ~1e-5 of scale instead of ~5e-7)",
                                                  "seg_hrnet2", (32, 64, 128, 256), "bf16x3", 32, 256)),
                           ("config3_w48_384_bf16", ("HRNet-W48 384x384 batch 64, bf16 storage / f32 accumulate (BASELINE configs[3])",
                                                     "seg_hrnet2", (48, 96, 192, 384), "bf16", 64, 384)),
                           ("seg_hrnet3", ("seg_hrnet3 (CBAM, val.py:380) W32 256x256 batch 32, 30 keypoints, fp32-grade",
                                           "seg_hrnet3", (32, 64, 128, 256), "fp32", 32, 256)),
                           ("seg_hrnet3_bf16x3_opt_in", ("seg_hrnet3 in the split-bf16 fast mode (precision='bf16x3': NOT fp32)",
                                                         "seg_hrnet3", (32, 64, 128, 256), "bf16x3", 32, 256))):
                try:
                    extras[key] = side_workload(*a, dev, steps=max(5, min(args.steps, 20)), warmup=3,
                                                profile_steps=max(1, min(args.profile_steps, 2)))
                except Exception as e:                  # noqa: BLE001 - an extra must not sink the headline
                    extras[key] = {"error": f"{type(e).__name__}: {e}"}

    cpu = None
    if rank == 0 and not args.no_cpu_baseline:
        from oracle import hrnet_ref
        variant_id = 1 if args.variant == "seg_hrnet3" else 0
        cpu = cpu_baseline(sd, hrnet_ref.default_cfg(cin, K, widths=widths, variant=variant_id), args.variant, hw,
                           args.cpu_seconds)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = n_total * args.steps / elapsed
        flops_crop = net.flops_per_crop(hw, hw)
        wname = "W48" if w48 else "W32"
        cfg_idx = 3 if args.workload == "w48-bf16" else (1 if world == 1 else 2)
        line = {
            "metric": f"HRNet-{wname} {hw}x{hw} crops/sec (heatmaps + fused argmax/refine keypoints)",
            "value": round(value, 1), "unit": "crops/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": DTYPES[precision],
            "data": "synthetic",
            "config": {"workload": f"{args.variant} HRNet-{wname} {hw}x{hw}, batch {B}/GPU, {K} keypoints, "
                                   f"fp32 NCHW in -> heatmaps -> keypoints (BASELINE configs[{cfg_idx}])",
                       "global_batch": n_total, "per_gpu_batch": B, "parallelism": f"dp{world}",
                       "hip_graph": graph is not None, "launches_per_forward": roof["launches_per_forward"] if roof else None,
                       "algorithmic_gflop_per_crop": round(flops_crop / 1e9, 3),
                       "whole_net_algorithmic_tflops": round(value * flops_crop / 1e12 / world, 2)},
            "roofline": roof, "cpu_baseline": cpu, "kernel_breakdown": breakdown, "extras": extras,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
