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

// lerp with the scale in/out handed over by the host (same f32 quotient as ATen computes)
__device__ __forceinline__ Lerp lerp_scaled(int dst, int in, float scale) {
    float src = scale * ((float)dst + 0.5f) - 0.5f;
    src = src < 0.f ? 0.f : src;
    Lerp r;
    r.i0 = min((int)src, in - 1);
    r.i1 = r.i0 + (r.i0 < in - 1 ? 1 : 0);
    r.l1 = src - (float)r.i0;
    r.l0 = 1.f - r.l1;
    return r;
}

struct FuseScales {
    float sy[4], sx[4];
};

// One workgroup row = one output row (blockIdx.y = n*H + y): the row decomposition and the vertical
// interpolation are wave-uniform, all per-lane index math is 32-bit.  NS same-resolution terms come
// first, then NU lower-resolution terms (the host orders them); both counts are compile-time.
// one 8-channel group of a pixel -> 8 floats (SB: hi + lo chunks, 32 bytes; BF: one 16-byte chunk)
template <int FMT>
__device__ __forceinline__ void load8(const char* pix, int c8, float v[8]) {
    if (FMT == FMT_BF) {
        unpack8_bf16(*reinterpret_cast<const uint4*>(pix + c8 * 16), v);
    } else if (FMT == FMT_F32) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(pix + c8 * 32), b = *reinterpret_cast<const f32x4*>(pix + c8 * 32 + 16);
        v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
    } else {
        join8(*reinterpret_cast<const uint4*>(pix + c8 * 32), *reinterpret_cast<const uint4*>(pix + c8 * 32 + 16), v);
    }
}

template <int NS, int NU, int FMT = FMT_SB>
__global__ __launch_bounds__(256) void fuse_kernel(FuseParams p, FuseScales fs) {
    constexpr bool BF = FMT == FMT_BF;
    const int G = p.Cp >> 3;
    const unsigned u = blockIdx.x * 256u + threadIdx.x;
    if (u >= (unsigned)(p.W * G)) return;
    const int x = (int)(u / (unsigned)G);
    const int c8 = (int)(u - (unsigned)x * (unsigned)G);
    const int row = blockIdx.y;                   // n*H + y
    const int n = row / p.H, y = row - n * p.H;
    const int pixb = p.Cp * (BF ? 2 : 4);

    float acc[8];
    if (NS > 0) {
        load8<FMT>(p.x[0] + ((size_t)row * p.W + x) * (size_t)pixb, c8, acc);
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = 0.f;
    }
#pragma unroll
    for (int t = 1; t < NS; ++t) {
        float v[8];
        load8<FMT>(p.x[t] + ((size_t)row * p.W + x) * (size_t)pixb, c8, v);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] += v[i];
    }
#pragma unroll
    for (int k = 0; k < NU; ++k) {
        const int t = NS + k;
        const int h = p.h[t], w = p.w[t];
        const Lerp ly = lerp_scaled(y, h, fs.sy[t]), lx = lerp_scaled(x, w, fs.sx[t]);
        const char* r0 = p.x[t] + ((size_t)n * h + ly.i0) * w * (size_t)pixb;
        const char* r1 = p.x[t] + ((size_t)n * h + ly.i1) * w * (size_t)pixb;
        const int o0 = lx.i0 * pixb, o1 = lx.i1 * pixb;
        float v00[8], v01[8], v10[8], v11[8];
        load8<FMT>(r0 + o0, c8, v00);
        load8<FMT>(r0 + o1, c8, v01);
        load8<FMT>(r1 + o0, c8, v10);
        load8<FMT>(r1 + o1, c8, v11);
#pragma unroll
        for (int i = 0; i < 8; ++i)
            acc[i] += ly.l0 * (lx.l0 * v00[i] + lx.l1 * v01[i]) + ly.l1 * (lx.l0 * v10[i] + lx.l1 * v11[i]);
    }
    if (p.relu) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = relu1(acc[i]);
    }
    if (BF) {
        *reinterpret_cast<uint4*>(p.y + ((size_t)row * p.W + x) * (size_t)pixb + c8 * 16) = pack8_bf16(acc);
        return;
    }
    if (FMT == FMT_F32) {
        char* o = p.y + ((size_t)row * p.W + x) * (size_t)pixb + c8 * 32;
        *reinterpret_cast<f32x4*>(o) = f32x4{acc[0], acc[1], acc[2], acc[3]};
        *reinterpret_cast<f32x4*>(o + 16) = f32x4{acc[4], acc[5], acc[6], acc[7]};
        return;
    }
    uint4 hi, lo;
    split8(acc, hi, lo);
    char* o = p.y + ((size_t)row * p.W + x) * (size_t)pixb + c8 * 32;
    *reinterpret_cast<uint4*>(o) = hi;
    *reinterpret_cast<uint4*>(o + 16) = lo;
}

template <int NS, int NU>
int launch_fuse_t(const FuseParams& p, const FuseScales& fs, hipStream_t stream) {
    const long long per_row = (long long)p.W * (p.Cp >> 3);
    const long long rows = (long long)p.N * p.H;
    if (per_row <= 0 || rows <= 0 || rows > 0x7fffffffLL || per_row > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    const dim3 grid((unsigned)((per_row + 255) / 256), (unsigned)rows);
    if (p.fmt == FMT_BF) hipLaunchKernelGGL((fuse_kernel<NS, NU, FMT_BF>), grid, dim3(256), 0, stream, p, fs);
    else if (p.fmt == FMT_F32) hipLaunchKernelGGL((fuse_kernel<NS, NU, FMT_F32>), grid, dim3(256), 0, stream, p, fs);
    else hipLaunchKernelGGL((fuse_kernel<NS, NU, FMT_SB>), grid, dim3(256), 0, stream, p, fs);
    return (int)hipGetLastError();
}

}  // namespace

int launch_fuse(const FuseParams& p_in, hipStream_t stream) {
    if ((p_in.Cp & 7) || p_in.nterms < 1 || p_in.nterms > 4) return (int)hipErrorInvalidValue;
    // same-resolution terms first (the sum is re-associated; every term is an exact f32 value of SB data)
    FuseParams p = p_in;
    FuseScales fs{};
    int ns = 0, k = 0;
    for (int pass = 0; pass < 2; ++pass)
        for (int t = 0; t < p_in.nterms; ++t) {
            const bool same = p_in.h[t] == p_in.H && p_in.w[t] == p_in.W;
            if (same != (pass == 0)) continue;
            p.x[k] = p_in.x[t]; p.h[k] = p_in.h[t]; p.w[k] = p_in.w[t];
            fs.sy[k] = (float)p_in.h[t] / (float)p_in.H;
            fs.sx[k] = (float)p_in.w[t] / (float)p_in.W;
            ++k;
            if (same) ++ns;
        }
    const int nu = p.nterms - ns;
    switch (ns * 10 + nu) {
        case 1: return launch_fuse_t<0, 1>(p, fs, stream);
        case 2: return launch_fuse_t<0, 2>(p, fs, stream);
        case 3: return launch_fuse_t<0, 3>(p, fs, stream);
        case 4: return launch_fuse_t<0, 4>(p, fs, stream);
        case 10: return launch_fuse_t<1, 0>(p, fs, stream);
        case 11: return launch_fuse_t<1, 1>(p, fs, stream);
        case 12: return launch_fuse_t<1, 2>(p, fs, stream);
        case 13: return launch_fuse_t<1, 3>(p, fs, stream);
        case 20: return launch_fuse_t<2, 0>(p, fs, stream);
        case 21: return launch_fuse_t<2, 1>(p, fs, stream);
        case 22: return launch_fuse_t<2, 2>(p, fs, stream);
        case 30: return launch_fuse_t<3, 0>(p, fs, stream);
        case 31: return launch_fuse_t<3, 1>(p, fs, stream);
        case 40: return launch_fuse_t<4, 0>(p, fs, stream);
    }
    return (int)hipErrorInvalidValue;
}

}  // namespace esa
