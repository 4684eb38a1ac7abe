// head.hip — head tail: UpsamplingBilinear2d(x2, align_corners=True) of last_layer[3..5]'s
// output, concat with the raw input crop, 3x3 conv (+bias, no BN, no activation) -> raw
// heatmaps, f32 NCHW.
//
// Replaces last_layer[6] + output_layer of models/seg_hrnet.py:330-340, 469.  The up-sampled
// K-channel map and the concat tensor are never materialised: a workgroup builds the halo'd
// (8+2)x(32+2) concat tile in LDS (bilinear taps taken straight from the half-resolution SB
// tensor, raw input straight from the caller's NCHW crop) and runs the (K+cin)*9*K MACs per
// pixel on the f32 VALU with wave-uniform (scalar) weights.  K*(K+cin)*9 = 1188 MACs per pixel
// for the 11-keypoint variant: 0.5 % of the network, HBM-write-bound (K*4 B per pixel out).
#include "kernels.h"
#include "sb.h"

namespace esa {
namespace {

constexpr int FTH = 8, FTW = 32;            // output tile
constexpr int FIH = FTH + 2, FIW = FTW + 2; // halo'd input tile
constexpr int FROW = FIW + 1;               // LDS row pitch (floats)

struct LerpT {
    int i0, i1;
    float l0, l1;
};
// ATen align_corners=True: scale = (in-1)/(out-1), src = scale*dst.
__device__ __forceinline__ LerpT lerp_ac_true(int dst, int in, int out) {
    const float scale = out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
    const float src = scale * (float)dst;
    LerpT r;
    r.i0 = min((int)src, in - 1);
    r.i1 = r.i0 + (r.i0 < in - 1 ? 1 : 0);
    r.l1 = src - (float)r.i0;
    r.l0 = 1.f - r.l1;
    return r;
}

template <int KT>
__global__ __launch_bounds__(256) void final_kernel(FinalParams p, int tiles_x, int tiles_y) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* tile = reinterpret_cast<float*>(smem);       // [K+cin][FIH][FROW]
    const int CT = p.K + p.cin;
    int b = blockIdx.x;
    const int tx = b % tiles_x; b /= tiles_x;
    const int ty = b % tiles_y;
    const int n = b / tiles_y;
    const int oy0 = ty * FTH, ox0 = tx * FTW;

    // ---- stage: up-sampled keypoint channels, 8 at a time per thread ------------------------
    const int G = (p.K + 7) >> 3;
    for (int u = threadIdx.x; u < FIH * FIW * G; u += 256) {
        const int c8 = u % G;
        const int q = u / G;
        const int py = q / FIW, px = q - py * FIW;
        const int gy = oy0 - 1 + py, gx = ox0 - 1 + px;
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = 0.f;
        if (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) {
            const LerpT ly = lerp_ac_true(gy, p.h, p.H), lx = lerp_ac_true(gx, p.wd, p.W);
            const size_t r0 = ((size_t)n * p.h + ly.i0) * p.wd, r1 = ((size_t)n * p.h + ly.i1) * p.wd;
            const size_t ps = (size_t)p.Cp * 4;
            float v00[8], v01[8], v10[8], v11[8];
            const char* a;
            a = p.h3 + (r0 + lx.i0) * ps + c8 * 32;
            join8(*reinterpret_cast<const uint4*>(a), *reinterpret_cast<const uint4*>(a + 16), v00);
            a = p.h3 + (r0 + lx.i1) * ps + c8 * 32;
            join8(*reinterpret_cast<const uint4*>(a), *reinterpret_cast<const uint4*>(a + 16), v01);
            a = p.h3 + (r1 + lx.i0) * ps + c8 * 32;
            join8(*reinterpret_cast<const uint4*>(a), *reinterpret_cast<const uint4*>(a + 16), v10);
            a = p.h3 + (r1 + lx.i1) * ps + c8 * 32;
            join8(*reinterpret_cast<const uint4*>(a), *reinterpret_cast<const uint4*>(a + 16), v11);
#pragma unroll
            for (int i = 0; i < 8; ++i)
                v[i] = ly.l0 * (lx.l0 * v00[i] + lx.l1 * v01[i]) + ly.l1 * (lx.l0 * v10[i] + lx.l1 * v11[i]);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = c8 * 8 + i;
            if (c < p.K) tile[(c * FIH + py) * FROW + px] = v[i];
        }
    }
    // ---- stage: raw input channels ---------------------------------------------------------------
    for (int u = threadIdx.x; u < FIH * FIW * p.cin; u += 256) {
        const int px = u % FIW;
        const int r = u / FIW;
        const int py = r % FIH, ci = r / FIH;
        const int gy = oy0 - 1 + py, gx = ox0 - 1 + px;
        float v = 0.f;
        if (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W)
            v = p.x0[(((size_t)n * p.cin + ci) * p.H + gy) * p.W + gx];
        tile[((p.K + ci) * FIH + py) * FROW + px] = v;
    }
    __syncthreads();

    // ---- compute: one pixel per thread, all K outputs ------------------------------------------
    const int lx = threadIdx.x & (FTW - 1), lyy = threadIdx.x / FTW;
    float acc[KT];
#pragma unroll
    for (int k = 0; k < KT; ++k) acc[k] = p.bias[k];
    for (int c = 0; c < CT; ++c) {
        const float* tp = tile + (c * FIH + lyy) * FROW + lx;
        const float* wp = p.w + (size_t)c * 9 * KT;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const float v = tp[(tap / 3) * FROW + (tap % 3)];
#pragma unroll
            for (int k = 0; k < KT; ++k) acc[k] = fmaf(v, wp[tap * KT + k], acc[k]);
        }
    }
    const int oy = oy0 + lyy, ox = ox0 + lx;
    if (oy < p.H && ox < p.W) {
#pragma unroll
        for (int k = 0; k < KT; ++k)
            if (k < p.K) p.out[(((size_t)n * p.K + k) * p.H + oy) * p.W + ox] = acc[k];
    }
}

template <int KT>
int launch_final_t(const FinalParams& p, hipStream_t stream) {
    const int tiles_x = (p.W + FTW - 1) / FTW, tiles_y = (p.H + FTH - 1) / FTH;
    const long long nblk = (long long)p.N * tiles_x * tiles_y;
    if (nblk <= 0 || nblk > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    const size_t lds = (size_t)(p.K + p.cin) * FIH * FROW * sizeof(float);
    auto kern = final_kernel<KT>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    if (lds > 64 * 1024) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), lds, stream, p, tiles_x, tiles_y);
    return (int)hipGetLastError();
}

}  // namespace

// padded output-channel count the weights/bias of the final conv must be packed with
int final_kt(int K) { return K <= 11 ? 11 : (K <= 16 ? 16 : (K <= 32 ? 32 : -1)); }

int launch_final(const FinalParams& p, hipStream_t stream) {
    switch (final_kt(p.K)) {
        case 11: return launch_final_t<11>(p, stream);
        case 16: return launch_final_t<16>(p, stream);
        case 32: return launch_final_t<32>(p, stream);
    }
    return (int)hipErrorInvalidValue;
}

}  // namespace esa
