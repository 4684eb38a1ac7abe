import sys, os, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np, torch, torch.nn.functional as F
from esa_pose_estimation_amd import _lib as L, synth
lib = L.lib()
n, cin, cout, h, w = 1, 128, 32, 16, 16
x = torch.from_numpy(synth.normal("opx", 1, (n, cin, h, w)))
wt = torch.from_numpy(synth.normal("opw", 2, (cout, cin, 1, 1), float(np.sqrt(1.0 / cin))))
b = torch.from_numpy(synth.normal("opb", 3, (cout,), 0.1))
ref = F.conv2d(x.double(), wt.double(), b.double())
y = torch.full(tuple(ref.shape), float("nan"), device="cuda")
L.check(lib.esahrnet_op_conv(x.cuda().data_ptr(), n, cin, h, w, wt.numpy().ctypes.data_as(C.c_void_p), b.numpy().ctypes.data_as(C.c_void_p), cout, 1, 1, 0, None, y.data_ptr(), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
torch.cuda.synchronize()
e = (y.cpu().double() - ref).abs()[0]
print("max", e.max().item(), "per-channel max", e.amax(dim=(1, 2)).numpy().round(5))
print("per-col max", e.amax(dim=(0, 1)).numpy().round(5))
def bf(t): return t.to(torch.bfloat16).to(torch.float64)
X = x.double()[0].reshape(cin, -1); Wm = wt.double().reshape(cout, cin)
Xh = bf(X.float()); Xl = bf((X - Xh).float()); Wh = bf(Wm.float()); Wl = bf((Wm - Wh).float())
full = Wh @ Xh + Wh @ Xl + Wl @ Xh + b.double()[:, None]
yy = y.cpu().double()[0].reshape(cout, -1)
for name, t in [("full", full), ("no WlXh", full - Wl @ Xh), ("no WhXl", full - Wh @ Xl), ("hh only", Wh @ Xh + b.double()[:, None])]:
    e = (yy - t).abs()
    print(name, "ch0", e[0].max().item(), "ch1", e[1].max().item(), "ch2", e[2].max().item(), "ch3", e[3].max().item())
