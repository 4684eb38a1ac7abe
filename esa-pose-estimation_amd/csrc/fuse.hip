// fuse.hip — cross-resolution fuse on SB tensors:  y = [relu]( sum_i  up_i(x_i) ).
//
// Replaces the summation of HighResolutionModule.forward (models/seg_hrnet.py:232-247): terms at
// the output resolution are read directly, lower-resolution terms (the 1x1 conv + BN of
// fuse_layers[i][j], j > i, evaluated on the low-resolution grid) are bilinearly up-sampled on
// the fly with F.interpolate's align_corners=False rule (:241-244).  The same kernel implements
// the pre-head up-sample + concat + last_layer[0] of :461-468 after the 1x1 conv has been pushed
// through the (linear) up-sampling, see plan.cpp.
//
// Memory-bound elementwise op: one thread per (pixel, 8-channel group), group index fastest
// across lanes so a pixel's channels are read and written as contiguous 32-byte pieces.
#include "kernels.h"
#include "sb.h"

namespace esa {
namespace {

struct Lerp {
    int i0, i1;
    float l0, l1;
};

// ATen area_pixel_compute_source_index(scale, dst, align_corners=false): src = scale*(dst+0.5)-0.5,
// clamped at 0; i1 = i0 + (i0 < in-1).
__device__ __forceinline__ Lerp lerp_ac_false(int dst, int in, int out) {
    const float scale = (float)in / (float)out;
    float src = scale * ((float)dst + 0.5f) - 0.5f;
    src = src < 0.f ? 0.f : src;
    Lerp r;
    r.i0 = min((int)src, in - 1);
    r.i1 = r.i0 + (r.i0 < in - 1 ? 1 : 0);
    r.l1 = src - (float)r.i0;
    r.l0 = 1.f - r.l1;
    return r;
}

__device__ __forceinline__ void load_group(const char* base, size_t pix, int Cp, int c8, float v[8]) {
    const char* a = base + pix * (size_t)(Cp * 4) + c8 * 32;
    join8(*reinterpret_cast<const uint4*>(a), *reinterpret_cast<const uint4*>(a + 16), v);
}

__global__ __launch_bounds__(256) void fuse_kernel(FuseParams p, long long total) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int G = p.Cp >> 3;
    const int c8 = (int)(idx % G);
    long long pix = idx / G;
    const int x = (int)(pix % p.W);
    long long row = pix / p.W;
    const int y = (int)(row % p.H);
    const int n = (int)(row / p.H);

    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
    for (int t = 0; t < p.nterms; ++t) {
        const int h = p.h[t], w = p.w[t];
        float v[8];
        if (h == p.H && w == p.W) {
            load_group(p.x[t], (size_t)pix, p.Cp, c8, v);
        } else {
            const Lerp ly = lerp_ac_false(y, h, p.H), lx = lerp_ac_false(x, w, p.W);
            const size_t r0 = ((size_t)n * h + ly.i0) * w, r1 = ((size_t)n * h + ly.i1) * w;
            float v00[8], v01[8], v10[8], v11[8];
            load_group(p.x[t], r0 + lx.i0, p.Cp, c8, v00);
            load_group(p.x[t], r0 + lx.i1, p.Cp, c8, v01);
            load_group(p.x[t], r1 + lx.i0, p.Cp, c8, v10);
            load_group(p.x[t], r1 + lx.i1, p.Cp, c8, v11);
#pragma unroll
            for (int i = 0; i < 8; ++i)
                v[i] = ly.l0 * (lx.l0 * v00[i] + lx.l1 * v01[i]) + ly.l1 * (lx.l0 * v10[i] + lx.l1 * v11[i]);
        }
        if (t == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = v[i];
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] += v[i];
        }
    }
    if (p.relu) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = relu1(acc[i]);
    }
    uint4 hi, lo;
    split8(acc, hi, lo);
    char* o = p.y + (size_t)pix * (size_t)(p.Cp * 4) + c8 * 32;
    *reinterpret_cast<uint4*>(o) = hi;
    *reinterpret_cast<uint4*>(o + 16) = lo;
}

}  // namespace

int launch_fuse(const FuseParams& p, hipStream_t stream) {
    if ((p.Cp & 7) || p.nterms < 1 || p.nterms > 4) return (int)hipErrorInvalidValue;
    const long long total = (long long)p.N * p.H * p.W * (p.Cp >> 3);
    const long long nblk = (total + 255) / 256;
    if (nblk <= 0 || nblk > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(fuse_kernel, dim3((unsigned)nblk), dim3(256), 0, stream, p, total);
    return (int)hipGetLastError();
}

}  // namespace esa
