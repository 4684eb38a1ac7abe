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

// ---- 2x2 pixel blocks (fp32-grade mode, even H and W, every up-sampled term at ratio >= 2) --------------------------------
// The kernel above issues 2 + 8 NU 16-byte loads per 32 bytes it writes: with three up-sampled terms the texture path (64
// B/clk/CU), not HBM, sets its rate (2.8-3.3 TB/s on the full-resolution levels).  Here a thread owns a 2x2 block of
// pixels x 8 channels: at ratio >= 2 the four pixels' taps lie in a 3x3 neighbourhood of the source — rows {i0(Y), i1(Y),
// i1(Y+1)}, and i0(Y+1) is one of the first two — and in ONE 2x2 cell for exact ratios 4 and 8, so the block loads 9 (or 4)
// source pixels per term instead of 16.  Every pixel evaluates exactly the expression of fuse_kernel on exactly its own
// four taps (picked from the loaded set by v_cndmask).
template <int NS, int NU>
__global__ __launch_bounds__(256) void fuse2x2_kernel(FuseParams p, FuseScales fs, int cell_mask) {
    const int G = p.Cp >> 3, W2 = p.W >> 1, H2 = p.H >> 1;
    const unsigned u = blockIdx.x * 256u + threadIdx.x;
    if (u >= (unsigned)(W2 * G)) return;
    const int bx = (int)(u / (unsigned)G);
    const int c8 = (int)(u - (unsigned)bx * (unsigned)G);
    const int brow = blockIdx.y;                  // n*H2 + by
    const int n = brow / H2, by = brow - n * H2;
    const int X = 2 * bx, Y = 2 * by;
    const int pixb = p.Cp * 4;
    const size_t row0 = (size_t)n * p.H + Y;      // image row of the block's first pixel row

    float acc[4][8];                              // pixel j = dy*2 + dx
    if (NS > 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            load8<FMT_F32>(p.x[0] + ((row0 + (j >> 1)) * p.W + X + (j & 1)) * (size_t)pixb, c8, acc[j]);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[j][i] = 0.f;
    }
#pragma unroll
    for (int t = 1; t < NS; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v[8];
            load8<FMT_F32>(p.x[t] + ((row0 + (j >> 1)) * p.W + X + (j & 1)) * (size_t)pixb, c8, v);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[j][i] += v[i];
        }
#pragma unroll
    for (int k = 0; k < NU; ++k) {
        const int t = NS + k;
        const int h = p.h[t], w = p.w[t];
        const Lerp ly[2] = {lerp_scaled(Y, h, fs.sy[t]), lerp_scaled(Y + 1, h, fs.sy[t])};
        const Lerp lx[2] = {lerp_scaled(X, w, fs.sx[t]), lerp_scaled(X + 1, w, fs.sx[t])};
        // loaded rows a = i0(Y), b = i1(Y), c = i1(Y+1); pixel row 1 reads (i0(Y+1) in {a, b}, c); columns likewise
        const char* base = p.x[t] + (size_t)n * h * w * (size_t)pixb;
        const char* ra = base + (size_t)ly[0].i0 * w * (size_t)pixb;
        const char* rb = base + (size_t)ly[0].i1 * w * (size_t)pixb;
        const char* rc = base + (size_t)ly[1].i1 * w * (size_t)pixb;
        const int ca = lx[0].i0 * pixb, cb = lx[0].i1 * pixb, cc = lx[1].i1 * pixb;
        const bool sely = ly[1].i0 != ly[0].i0, selx = lx[1].i0 != lx[0].i0;
        const bool cell = (cell_mask >> t) & 1;   // exact ratio 4 / 8 in both directions: the block's pixels share one 2x2 cell
        if (cell) {         // (wave-uniform) one 2x2 cell serves the four pixels: no selection
            float v00[8], v01[8], v10[8], v11[8];
            load8<FMT_F32>(ra + ca, c8, v00);
            load8<FMT_F32>(ra + cb, c8, v01);
            load8<FMT_F32>(rb + ca, c8, v10);
            load8<FMT_F32>(rb + cb, c8, v11);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int dy = j >> 1, dx = j & 1;
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    acc[j][i] += ly[dy].l0 * (lx[dx].l0 * v00[i] + lx[dx].l1 * v01[i]) + ly[dy].l1 * (lx[dx].l0 * v10[i] + lx[dx].l1 * v11[i]);
            }
        } else {
            float v[3][3][8];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const char* rr = r == 0 ? ra : r == 1 ? rb : rc;
                load8<FMT_F32>(rr + ca, c8, v[r][0]);
                load8<FMT_F32>(rr + cb, c8, v[r][1]);
                load8<FMT_F32>(rr + cc, c8, v[r][2]);
            }
            // first column of the dx = 1 pixels (i0(X+1) is column a or b), per loaded row; then the first row of the
            // dy = 1 pixels (row a or b) — the second taps are column c / row c
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float c0[3][2], c1[3][2];       // [row][dx]: the pixel column's two taps in loaded row r
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    c0[r][0] = v[r][0][i]; c1[r][0] = v[r][1][i];
                    c0[r][1] = selx ? v[r][1][i] : v[r][0][i]; c1[r][1] = v[r][2][i];
                }
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const float a0 = lx[dx].l0 * c0[0][dx] + lx[dx].l1 * c1[0][dx];      // row a, interpolated along x
                    const float b0 = lx[dx].l0 * c0[1][dx] + lx[dx].l1 * c1[1][dx];      // row b
                    const float g0 = lx[dx].l0 * c0[2][dx] + lx[dx].l1 * c1[2][dx];      // row c
                    acc[dx][i] += ly[0].l0 * a0 + ly[0].l1 * b0;
                    acc[2 + dx][i] += ly[1].l0 * (sely ? b0 : a0) + ly[1].l1 * g0;
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);      // one term's sources in registers at a time
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (p.relu) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[j][i] = relu1(acc[j][i]);
        }
        char* o = p.y + ((row0 + (j >> 1)) * p.W + X + (j & 1)) * (size_t)pixb + c8 * 32;
        *reinterpret_cast<f32x4*>(o) = f32x4{acc[j][0], acc[j][1], acc[j][2], acc[j][3]};
        *reinterpret_cast<f32x4*>(o + 16) = f32x4{acc[j][4], acc[j][5], acc[j][6], acc[j][7]};
    }
}

template <int NS, int NU>
int launch_fuse_t(const FuseParams& p, const FuseScales& fs, hipStream_t stream) {
    // fp32-grade mode, even grid, every up-sampled term at half the resolution or less: the 2x2-block kernel
    bool blocks = p.fmt == FMT_F32 && NU > 0 && !(p.H & 1) && !(p.W & 1);
    int cell_mask = 0;
    for (int t = NS; t < NS + NU && blocks; ++t) {
        if (2 * p.h[t] > p.H || 2 * p.w[t] > p.W) blocks = false;
        const bool exact = p.h[t] > 0 && p.w[t] > 0 && p.H % p.h[t] == 0 && p.W % p.w[t] == 0;
        const int ry = exact ? p.H / p.h[t] : 0, rx = exact ? p.W / p.w[t] : 0;
        if ((ry == 4 || ry == 8) && (rx == 4 || rx == 8)) cell_mask |= 1 << t;
    }
    if (blocks) {
        const long long per_row = (long long)(p.W >> 1) * (p.Cp >> 3), rows = (long long)p.N * (p.H >> 1);
        if (per_row <= 0 || rows <= 0 || rows > 0x7fffffffLL || per_row > 0x7fffffffLL) return (int)hipErrorInvalidValue;
        hipLaunchKernelGGL((fuse2x2_kernel<NS, NU>), dim3((unsigned)((per_row + 255) / 256), (unsigned)rows), dim3(256), 0, stream, p, fs, cell_mask);
        return (int)hipGetLastError();
    }
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
