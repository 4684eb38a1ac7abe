"""Kernel-time survey of conv_x6 launches (run under rocprofv3 --kernel-trace; durations come from the trace):
    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d <out> -- python3 $REPO/tools/x6_opbench.py
Each case runs esahrnet_op_conv_ex(precision=2) three times; tools/x6_opbench_report.py prints TFLOP/s per case."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CASES = [  # n, cin, cout, h, w, k, stride
    (32, 64, 64, 64, 64, 3, 1), (128, 64, 64, 64, 64, 3, 1),
    (32, 128, 128, 32, 32, 3, 1), (128, 128, 128, 32, 32, 3, 1),
    (32, 256, 256, 16, 16, 3, 1), (128, 256, 256, 16, 16, 3, 1),
    (32, 32, 32, 128, 128, 3, 1), (128, 32, 32, 128, 128, 3, 1),
    (32, 64, 64, 256, 256, 3, 2), (32, 32, 64, 128, 128, 3, 2), (32, 64, 128, 64, 64, 3, 2),
    (32, 32, 480, 128, 128, 1, 1), (32, 64, 480, 64, 64, 1, 1), (32, 256, 32, 16, 16, 1, 1),
]

def main():
    from esa_pose_estimation_amd import _lib, synth
    lib = _lib.lib()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for (n, cin, cout, h, w, k, s) in CASES:
        x = torch.from_numpy(synth.normal("x", 1, (n, cin, h, w))).cuda()
        wt = synth.normal("w", 2, (cout, cin, k, k), float(np.sqrt(1.0 / (cin * k * k))))
        b = synth.normal("b", 3, (cout,), 0.1)
        oh, ow = ((h + 1) // 2, (w + 1) // 2) if s == 2 else (h, w)
        y = torch.empty((n, cout, oh, ow), device="cuda")
        for _ in range(3):
            _lib.check(lib.esahrnet_op_conv_ex(x.data_ptr(), n, cin, h, w, wt.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p),
                                               cout, k, s, 1, None, y.data_ptr(), 2, st))
        torch.cuda.synchronize()
        print("case", n, cin, cout, h, w, k, s, flush=True)


if __name__ == "__main__":
    main()
