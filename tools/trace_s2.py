"""Phase timeline of the stream kernel (conv_s2c32.hip built with -DS2_TRACE=<Cinp>): runs one forward and prints,
per traced wave, the cycle stamps of each step's phases.  Debug tool; needs ESA_HIPCC_FLAGS="-DS2_TRACE=64" build."""
import ctypes, sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from esa_pose_estimation_amd import config, seg_hrnet2, synth, _lib

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 32
net = seg_hrnet2.get_seg_model(config.make_config(widths=(32, 64, 128, 256)))
net.load_state_dict(synth.make_state_dict({k: v.shape for k, v in net.state_dict().items()}, seed=0))
net = net.cuda().eval()
x = synth.make_crops(batch, net._cin, 256, 256, seed=1).cuda()
with torch.no_grad():
    for _ in range(3):
        net(x)
torch.cuda.synchronize()
lib = _lib.lib()
buf = np.zeros(64 * 4 * 8 * 16, dtype=np.uint64)
lib.esa_debug_s2_trace.argtypes = [ctypes.c_void_p]
rc = lib.esa_debug_s2_trace(buf.ctypes.data)
assert rc == 0, rc
t = buf.reshape(64, 4, 8, 16).astype(np.int64)
base = t[:, :, 0, 0].min()
names = ["start", "bar1", "ldsw", "bar2", "pref", "kx0", "kx1", "kx2", "epi"]
for wg in (0, 32):
    for w in range(4):
        hw = int(t[wg, w, 0, 15])
        print(f"wg {wg} wave {w} hw_id {hw:#x}")
        for s in range(8):
            r = t[wg, w, s]
            if r[0] == 0:
                continue
            cells = " ".join(f"{names[e]}={int(r[e] - base) if r[e] else -1:7d}" for e in range(9))
            print(f"   step {s}: {cells}")
life = [(int(t[wg, w, 0, 11] - t[wg, w, 0, 10]), int(t[wg, w, 0, 12] - t[wg, w, 0, 9]), int(t[wg, w, 0, 0] - t[wg, w, 0, 10]))
        for wg in range(64) for w in range(4) if t[wg, w, 0, 11]]
print("wave lifetime: ticks median", np.median([l[0] for l in life]), " wall(100MHz) median", np.median([l[1] for l in life]),
      " prologue ticks median", np.median([l[2] for l in life]))
w0 = t[:, :, 0, 9][t[:, :, 0, 11] > 0]; w1 = t[:, :, 0, 12][t[:, :, 0, 11] > 0]
print("first start -> last end over traced WGs (100 MHz ticks):", int(w1.max() - w0.min()), " start spread", int(w0.max() - w0.min()))
wg = np.zeros(2048, dtype=np.uint64)
lib.esa_debug_s2_wg.argtypes = [ctypes.c_void_p]
assert lib.esa_debug_s2_wg(wg.ctypes.data) == 0
wg = wg.reshape(1024, 2).astype(np.int64)
wg = wg[wg[:, 0] > 0]
t0 = wg[:, 0].min()
st = np.sort(wg[:, 0] - t0); en = np.sort(wg[:, 1] - t0)
print(f"{len(wg)} workgroups: start (10 ns ticks) p0 {st[0]} p25 {st[len(st)//4]} p50 {st[len(st)//2]} p75 {st[3*len(st)//4]} p100 {st[-1]};"
      f" end p0 {en[0]} p50 {en[len(en)//2]} p100 {en[-1]}; lifetime median {np.median(wg[:,1]-wg[:,0])}")
for x in range(8):
    e = wg[x::8, 1] - t0
    print(f"  XCD {x}: end min {e.min()} median {int(np.median(e))} max {e.max()}")
e = wg[:, 1] - t0
half = len(wg) // 2
print("  first-half WGs end median", int(np.median(e[:half])), " second-half", int(np.median(e[half:])))
order = np.argsort(wg[:, 0] - t0)
print("start by blockIdx (every 32nd):", [(int(i), int(wg[i, 0] - t0)) for i in range(0, len(wg), 32)])
# aggregate: mean phase durations over all traced waves/steps
d = {}
for wg in range(64):
    for w in range(4):
        for s in range(8):
            r = t[wg, w, s]
            if r[0] == 0 or r[7] == 0:
                continue
            seq = [0, 1, 2, 3, 4, 5, 6, 7]
            for a, b in zip(seq[:-1], seq[1:]):
                d.setdefault(f"{names[a]}->{names[b]}", []).append(int(r[b] - r[a]))
            if r[8]:
                d.setdefault("kx2->epi", []).append(int(r[8] - r[7]))
            if s + 1 < 8 and t[wg, w, s + 1, 0]:
                d.setdefault("step total", []).append(int(t[wg, w, s + 1, 0] - r[0]))
for half, rng in (("first WG of a CU", range(0, 32)), ("second WG of a CU", range(32, 64))):
    tot = []
    for wg_ in rng:
        for w in range(4):
            r = t[wg_, w]
            st_ = [int(r[s_ + 1, 0] - r[s_, 0]) for s_ in range(3) if r[s_ + 1, 0] and r[s_, 0] and r[s_ + 1, 0] > r[s_, 0]]
            if len(st_) == 3:
                tot.append(st_ + [int(r[3, 8] - r[3, 0])])
    if tot:
        print(half, "step durations (cycles, median over waves) steps 0..3:", np.median(np.array(tot), axis=0))
for k, v in d.items():
    print(f"{k:14s} mean {np.mean(v):8.0f}  median {np.median(v):8.0f}  min {np.min(v):7d} max {np.max(v):7d}  n {len(v)}")
