"""GPU parity of the single-pass bf16 mode (esahrnet_cfg.precision = 1; BASELINE.json configs[3]: "bf16 with fp32
BN accumulate").  Activations and weights are stored ONCE as bf16 (half the bytes of the split format, one MFMA per
product instead of three), accumulation and the folded-BN bias epilogue are f32.

Tolerances, stated up front:
  * operators: the kernel's result must be the bf16 ROUNDING of the exact result on bf16-rounded operands —
    |y - exact| <= 2^-8 |exact| + 2e-6 (half an ulp of bf16 is 2^-9; f32 accumulation noise on top);
  * whole network vs oracle/emulate_bf16.py (the same arithmetic restated on the CPU): the two differ only where an
    f32 sum lands on the other side of a bf16 rounding boundary (the summation order differs); one flip is one bf16
    ulp (0.4 %) of one activation, ~2.5e-4 of all elements flip per layer and ~60 layers of 3x3 receptive fields
    spread them: measured, the heat-maps of the two agree no better (mean 3e-4 .. 9e-4, worst 5e-3 .. 7e-3) than
    each agrees with the fp32 reference — the emulation pins the MODE (what is rounded where), the operator tests
    above pin the ARITHMETIC exactly.  Asserted: mean <= 2e-3, worst <= 1.5e-2, every stored intermediate within
    2^-6 of its scale;
  * whole network vs the fp32 reference (golden fixtures of the REAL reference / the fp32 oracle): bf16 storage
    cannot meet 1e-3 — SURVEY.md §8d measured 9.8e-3 for the reference under bf16 autocast and says to expect ~1e-2:
    asserted L_inf <= 3e-2, mean-abs <= 4e-3; reported besides: keypoint shift in pixels and arg-max flips."""
import ctypes as C
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL_EMU = 1.5e-2
TOL_EMU_MEAN = 2e-3
TOL_F32_LINF = 3e-2
TOL_F32_MEAN = 4e-3


def qb(t):
    return t.to(torch.bfloat16).to(torch.float32)


@pytest.fixture(scope="module")
def env():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU")
    from esa_pose_estimation_amd import _lib, config, inference, seg_hrnet, seg_hrnet2, synth
    from oracle import emulate_bf16, hrnet_ref, keypoints_ref
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    return dict(lib=_lib.lib(), L=_lib, config=config, inference=inference, seg_hrnet=seg_hrnet, seg_hrnet2=seg_hrnet2,
                synth=synth, emu=emulate_bf16, hrnet_ref=hrnet_ref, kref=keypoints_ref)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


CONV_CASES = [
    # n, cin, cout, h, w, k, stride, relu, res
    (2, 64, 64, 32, 32, 3, 1, True, True),
    (1, 128, 64, 16, 48, 3, 1, True, False),
    (1, 48, 96, 32, 32, 3, 2, True, False),        # W48 transition: 48 -> 96 (padded 64 -> 128)
    (2, 64, 64, 34, 30, 3, 2, False, False),       # partial tiles, stride 2
    (1, 192, 192, 24, 24, 3, 1, True, True),       # 3 blocks of 64 input channels
    (1, 384, 384, 12, 12, 3, 1, True, True),       # W48 deepest branch: 6 blocks
    (1, 32, 32, 7, 5, 3, 1, False, False),         # image smaller than a tile, channels padded 32 -> 64
    (1, 128, 32, 16, 16, 1, 1, False, False),      # 1x1 fuse-up
    (1, 96, 720, 8, 8, 1, 1, False, False),        # last_layer[0] slice of W48 branch 1
    (1, 720, 11, 20, 24, 1, 1, True, False),       # last_layer[3] of W48: 12 blocks in registers
    (2, 480, 480, 6, 10, 1, 1, False, False),      # 8 blocks
    (16, 64, 64, 64, 64, 3, 1, True, True),        # network scale: several steps per workgroup
    (16, 128, 128, 64, 64, 3, 2, True, False),
]


@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "x".join(map(str, c)))
def test_op_conv_bf16_is_the_rounding_of_the_exact_result(env, case):
    n, cin, cout, h, w, k, stride, relu, use_res = case
    synth, lib, L = env["synth"], env["lib"], env["L"]
    x = torch.from_numpy(synth.normal("bx", 1, (n, cin, h, w)))
    wt = torch.from_numpy(synth.normal("bw", 2, (cout, cin, k, k), float(np.sqrt(1.0 / (cin * k * k)))))
    b = torch.from_numpy(synth.normal("bb", 3, (cout,), 0.1))
    ref = F.conv2d(qb(x).double(), qb(wt).double(), b.double(), stride=stride, padding=(k - 1) // 2)
    res = None
    if use_res:
        res = torch.from_numpy(synth.normal("br", 4, tuple(ref.shape)))
        ref = ref + qb(res).double()
    if relu:
        ref = F.relu(ref)
    xd = x.cuda()
    rd = res.cuda() if use_res else None
    outs = []
    for _ in range(2):
        y = torch.full(tuple(ref.shape), float("nan"), device="cuda")
        L.check(lib.esahrnet_op_conv_ex(xd.data_ptr(), n, cin, h, w, wt.numpy().ctypes.data_as(C.c_void_p),
                                        b.numpy().ctypes.data_as(C.c_void_p), cout, k, stride, int(relu),
                                        rd.data_ptr() if use_res else None, y.data_ptr(), 1, _stream()))
        torch.cuda.synchronize()
        outs.append(y.cpu())
    y = outs[0].double()
    assert torch.equal(outs[0], outs[1])
    assert torch.equal(qb(outs[0]), outs[0])                     # the output IS bf16 data
    bound = ref.abs() * 2.0 ** -8 + 2e-6
    bad = (y - ref).abs() > bound
    assert not bool(bad.any()), ((y - ref).abs().max().item(), int(bad.sum()))


def test_op_fuse_bf16(env):
    synth, lib, L = env["synth"], env["lib"], env["L"]
    n, c, h, w = 2, 96, 24, 40
    sizes = [(24, 40), (12, 20), (6, 10), (3, 5)]
    xs = [torch.from_numpy(synth.normal(f"gx{i}", 5, (n, c, a, b))) for i, (a, b) in enumerate(sizes)]
    ref = qb(xs[0]).clone()
    for t in xs[1:]:
        ref = ref + F.interpolate(qb(t), size=(h, w), mode="bilinear", align_corners=False)
    ref = F.relu(ref).double()
    xd = [t.cuda() for t in xs]
    ptrs = (C.c_void_p * 4)(*[t.data_ptr() for t in xd])
    hs = (C.c_int * 4)(*[s[0] for s in sizes])
    ws = (C.c_int * 4)(*[s[1] for s in sizes])
    y = torch.empty((n, c, h, w), device="cuda")
    L.check(lib.esahrnet_op_fuse_ex(ptrs, hs, ws, 4, n, c, h, w, 1, y.data_ptr(), 1, _stream()))
    torch.cuda.synchronize()
    assert bool(((y.cpu().double() - ref).abs() <= ref.abs() * 2.0 ** -8 + 2e-6).all())


def _build(env, variant, widths, seed, gain=0.5):
    net = env[variant].get_seg_model(env["config"].make_config(widths=widths), precision="bf16")
    sd = env["synth"].make_state_dict({k: v.shape for k, v in net.state_dict().items()}, seed=seed, gain=gain)
    net.load_state_dict(sd, strict=True)
    return net.cuda().eval(), sd


def _report(tag, y, ref32, emu, kref):
    d = np.abs(y - ref32)
    kp_y, kp_r = kref.heatmaps_to_keypoints(y), kref.heatmaps_to_keypoints(ref32)
    shift = np.hypot(*(kp_y[..., :2] - kp_r[..., :2]).reshape(-1, 2).T)
    flips = int((y.reshape(*y.shape[:2], -1).argmax(-1) != ref32.reshape(*y.shape[:2], -1).argmax(-1)).sum())
    e = np.abs(y - emu).max()
    em = np.abs(y - emu).mean()
    print(f"{tag}: vs fp32 reference L_inf {d.max():.3e} mean-abs {d.mean():.3e} (|out| max {np.abs(ref32).max():.2f}); "
          f"keypoint shift median {np.median(shift):.3f} px max {shift.max():.2f} px, arg-max flips {flips} of {shift.size}; "
          f"vs bf16 emulation L_inf {e:.3e} mean-abs {em:.3e}")
    assert em <= TOL_EMU_MEAN, em
    return d.max(), d.mean(), e


@pytest.mark.parametrize("tag", ["tiny_hrnet2_64", "tiny_hrnet_64", "w32_hrnet2_128", "w32_hrnet2_256"])
def test_bf16_net_vs_reference_golden_and_emulation(env, golden_dir, tag):
    g = np.load(os.path.join(golden_dir, tag + ".npz"), allow_pickle=False)
    variant = str(g["variant"])
    cin, K = (3, 32) if variant == "seg_hrnet" else (1, 11)
    widths = tuple(int(v) for v in g["widths"])
    net, sd = _build(env, variant, widths, int(g["seed"]))
    x = env["synth"].make_crops(int(g["n"]), cin, int(g["hw"]), int(g["hw"]), seed=int(g["seed"]))
    cfg = env["hrnet_ref"].default_cfg(cin, K, widths=widths)
    with torch.no_grad():
        y = net(x.cuda()).cpu().numpy()
        emu = env["emu"].forward(sd, cfg, x).numpy()
    assert np.isfinite(y).all()
    s = int(g["subsample"])
    linf, mean, e = _report(tag, y[:, :, ::s, ::s], g["out"], emu[:, :, ::s, ::s], env["kref"])
    assert e <= TOL_EMU, e
    assert linf <= TOL_F32_LINF and mean <= TOL_F32_MEAN, (linf, mean)


def test_bf16_intermediates_match_emulation(env):
    net, sd = _build(env, "seg_hrnet2", (32, 64, 128, 256), 4)
    x = env["synth"].make_crops(1, 1, 96, 64, seed=4)
    taps_emu = {}
    with torch.no_grad():
        env["emu"].forward(sd, env["hrnet_ref"].default_cfg(1, 11), x, taps_emu)
        taps = net.taps(x.cuda())
    assert {"stem1", "stem2", "layer1", "stage2.0", "stage4.3"} <= set(taps)
    assert ("head0" in taps and "head3" in taps) != ("head3_fused" in taps)     # exactly one head alternative ran
    worst = {}
    for name, ref in taps_emu.items():
        if name == "head3" and "head3_fused" in taps:
            name = "head3_fused"            # head_fused_bf.hip: the 720-channel head0 never exists
        elif name not in taps:
            continue
        got = taps[name].cpu()
        assert got.shape == ref.shape, (name, got.shape, ref.shape)
        assert torch.equal(qb(got), got), name                        # stored tensors are bf16 data
        err = (got - ref).abs()
        worst[name] = float(err.max())
        # a flipped rounding is one bf16 ulp of the value: allow a few ulps, relative to the tensor's scale
        assert float(err.max()) <= 2.0 ** -6 * float(ref.abs().max()), (name, float(err.max()))
    print("bf16 taps vs emulation, worst abs diff:", {k: f"{v:.2e}" for k, v in worst.items()})


def test_bf16_unfused_head_alternative(env, monkeypatch):
    """ESAHRNET_BF_UNFUSED_HEAD=1: slice 0 + fuse + 1x1 instead of head_fused_bf (the alternative odd geometries take)."""
    monkeypatch.setenv("ESAHRNET_BF_UNFUSED_HEAD", "1")
    net, sd = _build(env, "seg_hrnet2", (32, 64, 128, 256), 3)
    x = env["synth"].make_crops(1, 1, 128, 128, seed=3)
    with torch.no_grad():
        emu = env["emu"].forward(sd, env["hrnet_ref"].default_cfg(1, 11), x)
        y, ops = net.forward_timed(x.cuda())
    assert "head_fused_bf" not in {o["kernel"] for o in ops}
    assert (y.cpu() - emu).abs().max().item() <= TOL_EMU and (y.cpu() - emu).abs().mean().item() <= TOL_EMU_MEAN


@pytest.mark.parametrize("hw", [(48, 80), (16, 16), (18, 34), (104, 72)])
def test_bf16_odd_shapes_match_emulation(env, hw):
    net, sd = _build(env, "seg_hrnet2", (32, 64, 128, 256), 6)
    x = env["synth"].make_crops(2, 1, hw[0], hw[1], seed=6)
    with torch.no_grad():
        emu = env["emu"].forward(sd, env["hrnet_ref"].default_cfg(1, 11), x)
        y = net(x.cuda()).cpu()
    assert (y - emu).abs().max().item() <= TOL_EMU


@pytest.mark.parametrize("hw", [(128, 128), (104, 72), (70, 50)])
def test_bf16_head_interpolation_on_matrix_cores_matches_valu_form(env, monkeypatch, hw):
    """head_fused_bf.hip evaluates sum_b up(t_b) as T . U on the matrix cores (U = interpolation weights: exact in bf16 for the
    2x / 4x / 8x grids, hi + lo otherwise — 104x72 and 70x50 have branch grids that are not exact decimations).  The first
    form (four VALU taps per branch, ESAHRNET_BF_HEAD_VALU=1) computes the same sums in another order: the two must agree
    to the rounding of h0 to bf16."""
    net, sd = _build(env, "seg_hrnet2", (32, 64, 128, 256), 8)
    x = env["synth"].make_crops(2, 1, hw[0], hw[1], seed=8).cuda()
    with torch.no_grad():
        monkeypatch.delenv("ESAHRNET_BF_HEAD_VALU", raising=False)
        y_mf, ops = net.forward_timed(x)
        monkeypatch.setenv("ESAHRNET_BF_HEAD_VALU", "1")
        y_valu = net(x)
    if "head_fused_bf" not in {o["kernel"] for o in ops}:
        pytest.skip("this geometry runs the unfused head alternative")
    scale = max(1.0, y_valu.abs().max().item())
    assert (y_mf - y_valu).abs().max().item() <= 4e-3 * scale
    assert (y_mf - y_valu).abs().mean().item() <= 4e-4 * scale


def test_bf16_w48_384_config3(env):
    """BASELINE configs[3]: HRNet-W48 (48/96/192/384), 384x384, bf16 — n = 2 against the emulation and the fp32
    oracle, then the batch-64 workload itself through size-independent properties (crops are independent: every
    sample equals its own batch-1 forward bit for bit; permutation equivariance; input untouched)."""
    widths = (48, 96, 192, 384)
    net, sd = _build(env, "seg_hrnet2", widths, 21)
    synth = env["synth"]
    cfg = env["hrnet_ref"].default_cfg(1, 11, widths=widths)
    x2 = synth.make_crops(2, 1, 384, 384, seed=21)
    with torch.no_grad():
        ref = env["hrnet_ref"].forward(sd, cfg, x2).numpy()
        emu = env["emu"].forward(sd, cfg, x2).numpy()
        y2 = net(x2.cuda()).cpu().numpy()
    linf, mean, e = _report("W48 384x384 bf16", y2, ref, emu, env["kref"])
    assert e <= TOL_EMU and linf <= TOL_F32_LINF and mean <= TOL_F32_MEAN, (e, linf, mean)
    x = torch.cat([x2[:1], synth.make_crops(63, 1, 384, 384, seed=77)]).cuda()
    xc = x.clone()
    with torch.no_grad():
        y = net(x)
        singles = {i: net(x[i:i + 1]) for i in (0, 13, 63)}
        perm = torch.randperm(64, generator=torch.Generator().manual_seed(1)).cuda()
        yp = net(x[perm])
    torch.cuda.synchronize()
    assert torch.equal(x, xc)
    for i, ys in singles.items():
        assert torch.equal(y[i:i + 1], ys), i
    assert torch.equal(yp, y[perm])
    assert np.array_equal(y[0].cpu().numpy(), y2[0])
    kp = env["inference"].heatmaps_to_keypoints(y)
    assert kp.shape == (64, 11, 3) and bool(torch.isfinite(kp).all())


def test_bf16_plan_reports_its_kernels(env):
    net, sd = _build(env, "seg_hrnet2", (48, 96, 192, 384), 21)
    x = env["synth"].make_crops(2, 1, 128, 128, seed=1).cuda()
    with torch.no_grad():
        y, ops = net.forward_timed(x)
    kernels = {o["kernel"] for o in ops}
    print(sorted(kernels))
    assert "conv_s2c32_kernel<1, 8, 4, false, true>" in kernels and "conv1x1_kernel<bf16>" in kernels
    assert "head_fused_bf" in kernels
    assert not any(k.startswith(("bblock32", "head_fused2", "head_t", "stem_fused", "conv_mfma")) for k in kernels)
    assert all(o["bytes"] > 0 for o in ops)
