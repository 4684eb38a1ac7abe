// crops.hip — crop + edge-pad + bilinear resize + normalise, straight from the 8-bit camera frame
// to the network's f32 NCHW input (SURVEY.md §8f NEXT-2).
//
// Replaces, per image, the host work of ESAValDataSet.__getitem__ (data_load_val.py:139-187):
// numpy slicing of the clamped box, np.pad(..., 'edge'), cv2.resize(image, (scale, scale))
// (INTER_LINEAR on uint8) and torchvision ToTensor + Normalize(mean, std).  Bug-compatible with the
// reference's pad call, which pads ROWS by the width deficit and COLUMNS by the height deficit
// (data_load_val.py:168) — only reachable when clamping at the frame border made the box non-square.
//
// The resize restates OpenCV's published 8-bit INTER_LINEAR: half-pixel centres, coefficients
// quantised to 11 bits (INTER_RESIZE_COEF_BITS), horizontal pass in int32, vertical pass
// ((b0*(r0>>4))>>16) + ((b1*(r1>>4))>>16) + 2) >> 2.  PARITY UNPINNED against cv2 itself (not
// installable here); bit-exact against oracle/crops_ref.py, which restates the same arithmetic.
#include "kernels.h"

namespace esa {
namespace {

__device__ __forceinline__ void coef(int d, int src, int dst, int& s0, int& s1, int& a0, int& a1) {
    const double scale = (double)src / (double)dst;
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) { f = 0.f; s = 0; }
    if (s >= src - 1) { f = 0.f; s = src - 1; }
    s0 = s;
    s1 = min(s + 1, src - 1);
    a0 = (int)rintf((1.f - f) * 2048.f);     // saturate_cast<short>: round to nearest even
    a1 = (int)rintf(f * 2048.f);
}

__global__ __launch_bounds__(256) void crop_kernel(const unsigned char* frames, const int* boxes, float* out, int N,
                                                   int FH, int FW, int S, float mean, float std_) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)N * S * S) return;
    const int dx = (int)(idx % S);
    const int dy = (int)((idx / S) % S);
    const int n = (int)(idx / ((long long)S * S));
    const int x0 = boxes[n * 4 + 0], y0 = boxes[n * 4 + 1], x1 = boxes[n * 4 + 2], y1 = boxes[n * 4 + 3];
    const int xs = x1 - x0, ys = y1 - y0, size = max(xs, ys);
    const int rows = ys + (size - xs), cols = xs + (size - ys);       // reference's swapped pad amounts
    int sx0, sx1, ax0, ax1, sy0, sy1, by0, by1;
    coef(dx, cols, S, sx0, sx1, ax0, ax1);
    coef(dy, rows, S, sy0, sy1, by0, by1);
    const unsigned char* f = frames + (size_t)n * FH * FW;
    // the box comes from the host (crops.val_box keeps it inside the frame); a C-ABI caller's bad box must not
    // become an out-of-bounds read, so the source row / column are clamped to the frame as well
    auto px = [&](int r, int c) {
        const int yy = min(max(y0 + min(r, ys - 1), 0), FH - 1), xx = min(max(x0 + min(c, xs - 1), 0), FW - 1);
        return (int)f[(size_t)yy * FW + xx];
    };
    const int r0 = px(sy0, sx0) * ax0 + px(sy0, sx1) * ax1;
    const int r1 = px(sy1, sx0) * ax0 + px(sy1, sx1) * ax1;
    int v = (((by0 * (r0 >> 4)) >> 16) + ((by1 * (r1 >> 4)) >> 16) + 2) >> 2;
    v = min(max(v, 0), 255);
    out[idx] = ((float)v / 255.f - mean) / std_;
}

}  // namespace

int launch_crops(const unsigned char* frames, const int* boxes, float* out, int N, int FH, int FW, int S,
                 float mean, float std_, hipStream_t s) {
    const long long total = (long long)N * S * S;
    if (total <= 0 || FH <= 0 || FW <= 0) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(crop_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, frames, boxes, out, N, FH,
                       FW, S, mean, std_);
    return (int)hipGetLastError();
}

}  // namespace esa
