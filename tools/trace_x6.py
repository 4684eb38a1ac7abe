"""Phase trace of conv_x6_kernel (debug build: ESA_HIPCC_FLAGS=-DX6_TRACE=1 python esa-pose-estimation_amd/build.py --force).
Runs one 3x3 stride-1 convolution and prints, per traced workgroup, the cycles between the stamps of every step:
barrier wait | phase-0 preamble (decode, tile loads) | phase 0 | phase 1 (+ reloads) | phase 2 | tail."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from esa_pose_estimation_amd import _lib, synth  # noqa: E402

STEM = len(sys.argv) > 1 and sys.argv[1] == "stem"        # build with -DX6_TRACE=3: one seg_hrnet2 forward, the fused stem's stamps
if STEM:
    sys.argv = sys.argv[:1]
n, cin, cout, h, w = [int(v) for v in (sys.argv[1:6] if len(sys.argv) > 5 else (128, 64, 64, 64, 64))]
stride = int(sys.argv[6]) if len(sys.argv) > 6 else 1        # build with -DX6_TRACE=<stride>
lib = _lib.lib()
raw = C.CDLL(_lib.LIB_PATH) if hasattr(_lib, "LIB_PATH") else lib
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
x = torch.from_numpy(synth.normal("x", 1, (n, cin, h, w))).cuda()
wt = synth.normal("w", 2, (cout, cin, 3, 3), float(np.sqrt(1.0 / (cin * 9))))
b = synth.normal("b", 3, (cout,), 0.1)
y = torch.empty((n, cout, (h + stride - 1) // stride, (w + stride - 1) // stride), device="cuda")
if STEM:
    from esa_pose_estimation_amd import config, seg_hrnet2
    net = seg_hrnet2.get_seg_model(config.make_config())
    net.load_state_dict(synth.make_state_dict({k: v.shape for k, v in net.state_dict().items()}, seed=0))
    net = net.cuda().eval()
    xs = synth.make_crops(32, 1, 256, 256, seed=1).cuda()
    with torch.no_grad():
        for _ in range(3):
            net(xs)
for _ in range(0 if STEM else 2):
    _lib.check(lib.esahrnet_op_conv_ex(x.data_ptr(), n, cin, h, w, wt.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p),
                                       cout, 3, stride, 1, None, y.data_ptr(), 2, st))
torch.cuda.synchronize()
buf = np.zeros(64 * 32 * 8, np.uint64)
fn = raw.esa_debug_x6_trace
fn.argtypes = [C.c_void_p]
assert fn(buf.ctypes.data_as(C.c_void_p)) == 0
t = buf.reshape(64, 32, 8).astype(np.int64)
names = ["barrier", "pre", "phase0", "phase1", "phase2", "tail"]
for wg in (0, 1, 7, 33):
    steps = [s for s in range(32) if t[wg, s, 6]]
    if len(steps) > 2:
        dc = float(t[wg, steps[-1], 0] - t[wg, steps[0], 0])
        dr = float(t[wg, steps[-1], 7] - t[wg, steps[0], 7])
        print(f"workgroup {wg}: shader clock {dc / dr * 0.1:.3f} GHz over {len(steps) - 1} steps ({dc / (len(steps) - 1):.0f} cycles per step)")
    else:
        print(f"workgroup {wg}")
    for s in range(32):
        if t[wg, s, 6] == 0:
            break
        d = [int(t[wg, s, i + 1] - t[wg, s, i]) for i in range(6)]
        gap = int(t[wg, s + 1, 0] - t[wg, s, 6]) if s + 1 < 32 and t[wg, s + 1, 0] else 0
        print(f"  step {s:2d}: " + "  ".join(f"{nm} {v:6d}" for nm, v in zip(names, d)) + f"  next {gap:5d}  total {int(t[wg, s, 6] - t[wg, s, 0]):6d}")

wg = np.zeros(1024 * 4, np.uint64)
fn2 = raw.esa_debug_x6_wg
fn2.argtypes = [C.c_void_p]
assert fn2(wg.ctypes.data_as(C.c_void_p)) == 0
wg = wg.reshape(1024, 4).astype(np.int64)
live = wg[:, 1] > 0
G = int(live.sum())
t0 = wg[live, 0].min()
st = (wg[live, 0] - t0) * 0.01
en = (wg[live, 1] - t0) * 0.01
print(f"{G} workgroups; start 0 .. {st.max():.1f} us; end {en.min():.1f} .. {en.max():.1f} us; steps per workgroup {wg[live, 2].min()} .. {wg[live, 2].max()}")
for name, sel in (("first half", slice(0, G // 2)), ("second half", slice(G // 2, G))):
    e = en[sel]; s0 = st[sel]
    print(f"  {name}: start median {np.median(s0):6.1f} us, end min / median / max {e.min():6.1f} / {np.median(e):6.1f} / {e.max():6.1f} us")
cu = {}
for b in range(G):
    hw = int(wg[b, 3])
    key = ((hw >> 8) & 0xf, (hw >> 13) & 0x7, (hw >> 16) & 0xf) if False else hw & ~0x3f  # everything but wave / simd bits
    cu.setdefault(key, []).append(b)
pairs = [v for v in cu.values() if len(v) == 2]
print(f"  {len(pairs)} CU slots with exactly two traced workgroups; of them {sum(1 for a, b in pairs if (a < G // 2) != (b < G // 2))} pair the two halves of the grid")

d = (wg[:G, 1] - wg[:G, 0]) * 0.01
xcc = (wg[:G, 3] >> 0) & 0  # placeholder
print("duration (us) by workgroup index decile:", " ".join(f"{d[i * G // 10:(i + 1) * G // 10].mean():.0f}" for i in range(10)))
print("workgroup 0 duration", f"{d[0]:.1f}", "min", f"{d.min():.1f}", "max", f"{d.max():.1f}")
hw = wg[:G, 3]
for name, sh, w in (("se_id", 13, 3), ("cu_id", 8, 4), ("sh_id", 12, 1)):
    v = (hw >> sh) & ((1 << w) - 1)
    print(name, " ".join(f"{int(k)}:{d[v == k].mean():.0f}" for k in np.unique(v)))

for b in (0, 1, 7, 33):
    pro = (int(t[b, 0, 7]) - int(wg[b, 0])) * 0.01
    last = max(s_ for s_ in range(32) if t[b, s_, 6])
    loop = (int(t[b, last, 7]) - int(t[b, 0, 7])) * 0.01
    print(f"workgroup {b}: prologue {pro:.1f} us, steps 0..{last - 1} {loop:.1f} us, whole {d[b]:.1f} us")
