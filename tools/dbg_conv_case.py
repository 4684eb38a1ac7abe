"""Debug: one convolution through esahrnet_op_conv at a network-like size against torch CPU; prints where it differs.
usage: dbg_conv_case.py n cin cout h w k stride relu res"""
import ctypes as C, sys, os
import numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from esa_pose_estimation_amd import _lib as L, synth
n, cin, cout, h, w, k, stride, relu, use_res = [int(v) for v in sys.argv[1:10]]
lib = L.lib()
x = torch.from_numpy(synth.normal("opx", 1, (n, cin, h, w)))
wt = torch.from_numpy(synth.normal("opw", 2, (cout, cin, k, k), float(np.sqrt(1.0 / (cin * k * k)))))
b = torch.from_numpy(synth.normal("opb", 3, (cout,), 0.1))
ref = F.conv2d(x, wt, b, stride=stride, padding=(k - 1) // 2)
res = None
if use_res:
    res = torch.from_numpy(synth.normal("opr", 4, tuple(ref.shape)))
    ref = ref + res
if relu:
    ref = F.relu(ref)
xd = x.cuda(); rd = res.cuda() if use_res else None
for rep in range(3):
    y = torch.full(tuple(ref.shape), float("nan"), device="cuda")
    L.check(lib.esahrnet_op_conv(xd.data_ptr(), n, cin, h, w, wt.numpy().ctypes.data_as(C.c_void_p),
                                 b.numpy().ctypes.data_as(C.c_void_p), cout, k, stride, int(relu),
                                 rd.data_ptr() if use_res else None, y.data_ptr(), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    d = (y.cpu() - ref).abs()
    bad = (d > 1e-3) | torch.isnan(d)
    print(f"rep {rep}: max err {d[~torch.isnan(d)].max().item():.3e}  bad elements {int(bad.sum())}")
    if bad.any():
        idx = bad.nonzero()
        print("  first bad (n, c, y, x):", idx[:6].tolist())
        cols = sorted(set((idx[:, 3] % 16).tolist())); rows = sorted(set((idx[:, 2] % 4).tolist()))
        print("  x % 16 of bad:", cols, " y % 4:", rows, " channels//16:", sorted(set((idx[:, 1] // 16).tolist())), " images:", sorted(set(idx[:, 0].tolist()))[:10])

# ---- which tap / chunk explains a wrong element?  (only when something failed in the last repetition)
if bad.any():
    import itertools
    nchunks = cin // 32
    pad = (k - 1) // 2
    xp = F.pad(x, (pad, pad, pad, pad))
    yc = y.cpu()
    shown = 0
    for (bn, bc, by, bx) in idx[:200].tolist():
        diff = (yc[bn, bc, by, bx] - ref[bn, bc, by, bx]).item()
        if use_res and relu and ref[bn, bc, by, bx] == 0:
            continue
        best = None
        for c, ky, kx in itertools.product(range(nchunks), range(k), range(k)):
            xs = xp[bn, c * 32:(c + 1) * 32, by * stride + ky, bx * stride + kx]
            ws = wt[bc, c * 32:(c + 1) * 32, ky, kx]
            t = float((xs * ws).sum())
            # candidates: contribution missing (X read as zero), or X taken from the previous / next chunk (stale plane)
            cands = {"missing": -t}
            for dc in (-1, 1):
                if 0 <= c + dc < nchunks:
                    xo = xp[bn, (c + dc) * 32:(c + dc + 1) * 32, by * stride + ky, bx * stride + kx]
                    cands[f"x from chunk {c + dc}"] = float((xo * ws).sum()) - t
            for name, val in cands.items():
                e = abs(val - diff)
                if best is None or e < best[0]:
                    best = (e, c, ky, kx, name, val)
        print(f"  ({bn},{bc},{by},{bx}) diff {diff:+.4f}: best single-tap explanation chunk {best[1]} tap ({best[2]},{best[3]}) {best[4]} -> {best[5]:+.4f} (residual {best[0]:.2e})")
        shown += 1
        if shown >= 8:
            break
