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
#include "devstate.h"
#include "kernels.h"
#include "sb.h"

namespace esa {
namespace {

constexpr int FTW = 32;                     // output tile: 32 columns x 8 * RPT rows (a thread owns RPT vertically adjacent pixels)
constexpr int FIW = FTW + 2;                // halo'd input tile width
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

__device__ __forceinline__ LerpT lerp_ac_scaled(int dst, int in, float scale) {      // lerp_ac_true with its scale given
    const float src = scale * (float)dst;
    LerpT r;
    r.i0 = min((int)src, in - 1);
    r.i1 = r.i0 + (r.i0 < in - 1 ? 1 : 0);
    r.l1 = src - (float)r.i0;
    r.l0 = 1.f - r.l1;
    return r;
}

// RPT = 2: the two pixels of a thread share 12 of their 18 tile reads per channel and every weight, and their
// multiply-adds pair up as v_pk_fma_f32 (same order per pixel as RPT = 1) — 102 -> ~60 us at W32 256^2.
template <int KT, int RPT>
__global__ __launch_bounds__(256) void final_kernel(FinalParams p, int tiles_x, int tiles_y) {
    constexpr int FTH = 8 * RPT, FIH = FTH + 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* tile = reinterpret_cast<float*>(smem);       // [K+cin][FIH][FROW]
    const int CT = p.K + p.cin;
    int b = blockIdx.x;
    const int tx = b % tiles_x; b /= tiles_x;
    const int ty = b % tiles_y;
    const int n = b / tiles_y;
    const int oy0 = ty * FTH, ox0 = tx * FTW;

    // ---- stage: up-sampled keypoint channels, 8 at a time per thread ------------------------
    constexpr int G = (KT + 7) >> 3;        // (K <= KT: groups past K hold nothing the compute loop reads)
    // ATen align_corners=True: scale = (in-1)/(out-1), src = scale*dst — the two divisions once per thread, not per unit
    const float sc_y = p.H > 1 ? (float)(p.h - 1) / (float)(p.H - 1) : 0.f, sc_x = p.W > 1 ? (float)(p.wd - 1) / (float)(p.W - 1) : 0.f;
    for (int u = threadIdx.x; u < FIH * FIW * G; u += 256) {
        const int c8 = u % G;
        const int q = u / G;
        const int py = q / FIW, px = q - py * FIW;
        const int gy = oy0 - 1 + py, gx = ox0 - 1 + px;
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = 0.f;
        if (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) {
            const LerpT ly = lerp_ac_scaled(gy, p.h, sc_y), lx = lerp_ac_scaled(gx, p.wd, sc_x);
            const size_t r0 = ((size_t)n * p.h + ly.i0) * p.wd, r1 = ((size_t)n * p.h + ly.i1) * p.wd;
            const size_t ps = (size_t)p.Cp * (p.fmt == FMT_BF ? 2 : 4);
            float v00[8], v01[8], v10[8], v11[8];
            auto ld = [&](size_t pix, float v_[8]) {
                if (p.fmt == FMT_BF) {
                    unpack8_bf16(*reinterpret_cast<const uint4*>(p.h3 + pix * ps + c8 * 16), v_);
                } else if (p.fmt == FMT_F32) {
                    const float* a = reinterpret_cast<const float*>(p.h3 + pix * ps) + c8 * 8;
                    const f32x4 a0 = *reinterpret_cast<const f32x4*>(a), a1 = *reinterpret_cast<const f32x4*>(a + 4);
                    v_[0] = a0[0]; v_[1] = a0[1]; v_[2] = a0[2]; v_[3] = a0[3]; v_[4] = a1[0]; v_[5] = a1[1]; v_[6] = a1[2]; v_[7] = a1[3];
                } else {
                    const char* a = p.h3 + pix * ps + c8 * 32;
                    join8(*reinterpret_cast<const uint4*>(a), *reinterpret_cast<const uint4*>(a + 16), v_);
                }
            };
            ld(r0 + lx.i0, v00);
            ld(r0 + lx.i1, v01);
            ld(r1 + lx.i0, v10);
            ld(r1 + lx.i1, v11);
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

    // ---- compute: RPT pixels per thread, all K outputs ----------------------------------------
    const int lx = threadIdx.x & (FTW - 1), lyy = (threadIdx.x / FTW) * RPT;
    if constexpr (RPT == 2) {
        f32x2 acc[KT];
#pragma unroll
        for (int k = 0; k < KT; ++k) acc[k] = f32x2{p.bias[k], p.bias[k]};
        for (int c = 0; c < CT; ++c) {
            const float* tp = tile + (c * FIH + lyy) * FROW + lx;
            const float* wp = p.w + (size_t)c * 9 * KT;
            float r[4][3];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) r[i][j] = tp[i * FROW + j];
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const f32x2 v = {r[tap / 3][tap % 3], r[tap / 3 + 1][tap % 3]};
#pragma unroll
                for (int k = 0; k < KT; ++k) {
                    const float w = wp[tap * KT + k];
                    acc[k] = __builtin_elementwise_fma(v, f32x2{w, w}, acc[k]);
                }
            }
        }
        const int ox = ox0 + lx;
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int oy = oy0 + lyy + rr;
            if (oy < p.H && ox < p.W) {
#pragma unroll
                for (int k = 0; k < KT; ++k)
                    if (k < p.K) p.out[(((size_t)n * p.K + k) * p.H + oy) * p.W + ox] = acc[k][rr];
            }
        }
    } else {
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
}
template <int KT, int RPT>
int launch_final_rt(const FinalParams& p, hipStream_t stream) {
    constexpr int FTH = 8 * RPT, FIH = FTH + 2;
    const int tiles_x = (p.W + FTW - 1) / FTW, tiles_y = (p.H + FTH - 1) / FTH;
    const long long nblk = (long long)p.N * tiles_x * tiles_y;
    if (nblk <= 0 || nblk > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    const size_t lds = (size_t)(p.K + p.cin) * FIH * FROW * sizeof(float);
    auto kern = final_kernel<KT, RPT>;
    if (const int e_ = ensure_dyn_lds(reinterpret_cast<const void*>(kern), 64 * 1024)) return e_;
    if (lds > 64 * 1024) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), lds, stream, p, tiles_x, tiles_y);
    return (int)hipGetLastError();
}
template <int KT>
int launch_final_t(const FinalParams& p, hipStream_t stream) {
    // two rows per thread where the 18-row tile fits the 64 KB the kernel may ask for (same multiply-add order either way)
    if ((size_t)(p.K + p.cin) * 18 * FROW * sizeof(float) <= 64 * 1024 && KT <= 16) return launch_final_rt<KT, 2>(p, stream);
    return launch_final_rt<KT, 1>(p, stream);
}


// ---------------------------------------------------------------------------------------------------
// Matrix-core version of the same op.  The contraction index is (tap, channel) with the K+cin concat
// channels padded to CG groups of 8: K-slot group G = tap*CG + cg, four groups per 32-wide MFMA chunk.
// The workgroup builds the halo'd 18x18 concat tile ONCE in LDS as split-bf16 operand planes
// [part][cg][pixel][16 B] (bilinear taps + raw input, then hi/lo split), and every tap is just another
// LDS address of that tile (no im2col).  A = tile fragments (rows = 16 pixels of an output row),
// B = weight fragments (columns = output channels), so D[pixel][cout] leaves 4 consecutive pixels of
// one heat-map in a lane: 16-byte stores into the f32 NCHW output.
constexpr int MTH = 16, MTW = 16;                 // output tile
constexpr int MIH = MTH + 2, MIW = MTW + 2;
constexpr int MPLANE = ((MIH * MIW * 16 + 255) / 256) * 256;       // 5376 B, multiple of 256 B (bank-congruent planes)
constexpr int MSRC = 12;                           // source window of a tile (rows and columns), see the staging
__host__ __device__ constexpr bool final_wreg(int cg, int m) { return m == 1 && (9 * cg + 3) / 4 <= 7; }

template <int CG, int M>
__global__ __launch_bounds__(256, 2) void final_mfma_kernel(FinalParams p, const uint4* __restrict__ wpk, int tiles_x,
                                                           int tiles_y) {
    constexpr int NG = 9 * CG;                    // K-slot groups carrying weights
    constexpr int NCH = (NG + 3) / 4;             // 32-wide chunks
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const tile = smem;                      // [part][cg][pixel][16 B]
    // small weight sets stay in registers (10 fragments per lane for K <= 16, cin = 1): 10 KB of LDS less per workgroup,
    // i.e. five instead of three workgroups per CU, and no LDS reads for the B operand
    constexpr bool WREG = final_wreg(CG, M);
    char* const wl = smem + 2 * CG * MPLANE;      // [m][chunk][part][lane][16 B]   (absent with WREG)
    float* const srcf = reinterpret_cast<float*>(wl + (WREG ? 0 : M * NCH * 2048));      // [MSRC][MSRC][CG][8] f32 source window
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i16 = lane & 15, g = lane >> 4;
    int b = xcd_contiguous(blockIdx.x, gridDim.x);
    const int tx = b % tiles_x; b /= tiles_x;
    const int ty = b % tiles_y;
    const int n = b / tiles_y;
    const int oy0 = ty * MTH, ox0 = tx * MTW;

    // ---- weights -> registers or LDS ------------------------------------------------------------------
    uint4 wr[WREG ? 2 * NCH : 1];
    if (WREG) {
#pragma unroll
        for (int c = 0; c < 2 * NCH; ++c) wr[c] = wpk[c * 64 + lane];          // [chunk][part][lane]
    } else {
        for (int u = tid; u < M * NCH * 2 * 64; u += 256) *reinterpret_cast<uint4*>(wl + u * 16) = wpk[u];
    }

    // ---- source window of the up-sampled keypoint channels -> LDS as f32, each source pixel unpacked ONCE ---------
    // (x2 with align_corners=True: an 18-wide strip of the output covers at most 11 source columns; 12 with slack)
    const int NKG = (p.K + 7) >> 3;               // channel groups that hold keypoint channels
    const int gy_lo = max(oy0 - 1, 0), gy_hi = min(oy0 + MTH, p.H - 1);
    const int gx_lo = max(ox0 - 1, 0), gx_hi = min(ox0 + MTW, p.W - 1);
    const int sy0 = lerp_ac_true(gy_lo, p.h, p.H).i0, sx0 = lerp_ac_true(gx_lo, p.wd, p.W).i0;
    const int srows = min(lerp_ac_true(gy_hi, p.h, p.H).i1 - sy0 + 1, MSRC);
    const int scols = min(lerp_ac_true(gx_hi, p.wd, p.W).i1 - sx0 + 1, MSRC);
    for (int u = tid; u < srows * scols * NKG; u += 256) {
        const int kg = u % NKG;
        const int q = u / NKG;
        const int sy = q / scols, sx = q - sy * scols;
        float v[8];
        if (p.fmt == FMT_BF) {         // single-bf16 tensor (precision = 1): 2 bytes per channel
            const char* a = p.h3 + (((size_t)n * p.h + sy0 + sy) * p.wd + sx0 + sx) * ((size_t)p.Cp * 2) + kg * 16;
            unpack8_bf16(*reinterpret_cast<const uint4*>(a), v);
        } else {
            const char* a = p.h3 + (((size_t)n * p.h + sy0 + sy) * p.wd + sx0 + sx) * ((size_t)p.Cp * 4) + kg * 32;
            join8(*reinterpret_cast<const uint4*>(a), *reinterpret_cast<const uint4*>(a + 16), v);
        }
        float* o = srcf + ((sy * MSRC + sx) * CG + kg) * 8;
        *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(o + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
    __syncthreads();

    // ---- concat tile -> LDS operand planes: unit = (pixel, channel group) --------------------------
    for (int u = tid; u < MIH * MIW * CG; u += 256) {
        const int cg = u % CG;
        const int q = u / CG;
        const int py = q / MIW, px = q - py * MIW;
        const int gy = oy0 - 1 + py, gx = ox0 - 1 + px;
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = 0.f;
        if (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) {
            if (cg * 8 < p.K) {                   // up-sampled keypoint channels of this group
                const LerpT ly = lerp_ac_true(gy, p.h, p.H), lx = lerp_ac_true(gx, p.wd, p.W);
                const int r0 = min(ly.i0 - sy0, MSRC - 1) * MSRC, r1 = min(ly.i1 - sy0, MSRC - 1) * MSRC;
                const int c0 = min(lx.i0 - sx0, MSRC - 1), c1 = min(lx.i1 - sx0, MSRC - 1);
                const float* a00 = srcf + ((r0 + c0) * CG + cg) * 8;
                const float* a01 = srcf + ((r0 + c1) * CG + cg) * 8;
                const float* a10 = srcf + ((r1 + c0) * CG + cg) * 8;
                const float* a11 = srcf + ((r1 + c1) * CG + cg) * 8;
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const float4 v00 = *reinterpret_cast<const float4*>(a00 + 4 * hf), v01 = *reinterpret_cast<const float4*>(a01 + 4 * hf);
                    const float4 v10 = *reinterpret_cast<const float4*>(a10 + 4 * hf), v11 = *reinterpret_cast<const float4*>(a11 + 4 * hf);
                    v[4 * hf + 0] = ly.l0 * (lx.l0 * v00.x + lx.l1 * v01.x) + ly.l1 * (lx.l0 * v10.x + lx.l1 * v11.x);
                    v[4 * hf + 1] = ly.l0 * (lx.l0 * v00.y + lx.l1 * v01.y) + ly.l1 * (lx.l0 * v10.y + lx.l1 * v11.y);
                    v[4 * hf + 2] = ly.l0 * (lx.l0 * v00.z + lx.l1 * v01.z) + ly.l1 * (lx.l0 * v10.z + lx.l1 * v11.z);
                    v[4 * hf + 3] = ly.l0 * (lx.l0 * v00.w + lx.l1 * v01.w) + ly.l1 * (lx.l0 * v10.w + lx.l1 * v11.w);
                }
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (cg * 8 + i >= p.K) v[i] = 0.f;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {         // raw input channels that fall into this group
                const int ci = cg * 8 + i - p.K;
                if (ci >= 0 && ci < p.cin) v[i] = p.x0[(((size_t)n * p.cin + ci) * p.H + gy) * p.W + gx];
            }
        }
        uint4 hi, lo;
        split8(v, hi, lo);
        *reinterpret_cast<uint4*>(tile + cg * MPLANE + q * 16) = hi;
        *reinterpret_cast<uint4*>(tile + (CG + cg) * MPLANE + q * 16) = lo;
    }

    // per-lane tile offset of K-slot group (chunk, g): plane of its channel group + the tap's pixel shift
    int offp[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        int G = c * 4 + g;
        G = G < NG ? G : 0;                       // padded groups have zero weights: any valid address
        const int tap = G / CG, cg = G - tap * CG;
        offp[c] = cg * MPLANE + ((tap / 3) * MIW + (tap % 3) + i16) * 16;
    }
    __syncthreads();

    // ---- 4 output rows per wave -----------------------------------------------------------------
    f32x4 acc[M][4];
#pragma unroll
    for (int m = 0; m < M; ++m) {
        const float bv = p.bias[m * 16 + i16];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[m][t] = f32x4{bv, bv, bv, bv};
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        bf16x8 wh[M], wlo[M];
#pragma unroll
        for (int m = 0; m < M; ++m) {
            if (WREG) {
                wh[m] = __builtin_bit_cast(bf16x8, wr[WREG ? 2 * c : 0]);
                wlo[m] = __builtin_bit_cast(bf16x8, wr[WREG ? 2 * c + 1 : 0]);
            } else {
                wh[m] = *reinterpret_cast<const bf16x8*>(wl + (((m * NCH + c) * 2 + 0) * 64 + lane) * 16);
                wlo[m] = *reinterpret_cast<const bf16x8*>(wl + (((m * NCH + c) * 2 + 1) * 64 + lane) * 16);
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const char* a = tile + offp[c] + ((wave * 4 + t) * MIW) * 16;
            const bf16x8 xh = *reinterpret_cast<const bf16x8*>(a);
            const bf16x8 xl = *reinterpret_cast<const bf16x8*>(a + CG * MPLANE);
#pragma unroll
            for (int m = 0; m < M; ++m) {
                acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xl, wh[m], acc[m][t], 0, 0, 0);
                acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh, wlo[m], acc[m][t], 0, 0, 0);
                acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh, wh[m], acc[m][t], 0, 0, 0);
            }
        }
    }
    // D[pixel][cout]: lane = cout i16 (+16 m), registers = pixels ox0 + 4g .. +3 of row oy
#pragma unroll
    for (int m = 0; m < M; ++m) {
        const int k = m * 16 + i16;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int oy = oy0 + wave * 4 + t, ox = ox0 + g * 4;
            if (k < p.K && oy < p.H && ox < p.W) {
                float* o = p.out + (((size_t)n * p.K + k) * p.H + oy) * p.W + ox;
                if (ox + 3 < p.W && (p.W & 3) == 0) *reinterpret_cast<f32x4*>(o) = acc[m][t];
                else
                    for (int r = 0; r < 4 && ox + r < p.W; ++r) o[r] = acc[m][t][r];
            }
        }
    }
    // ---- optional: the tile's first row-major maximum of every heat-map (keypoints.hip finishes over the tiles) ----
    if (p.part) {
        float* const pv = srcf + MSRC * MSRC * CG * 8;          // [wave][M*16] value, then [wave][M*16] index
        int* const pi = reinterpret_cast<int*>(pv + 4 * M * 16);
#pragma unroll
        for (int m = 0; m < M; ++m) {
            float bv = -INFINITY;
            int bi = 0x7fffffff;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int oy = oy0 + wave * 4 + t;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ox = ox0 + g * 4 + r;
                    if (oy < p.H && ox < p.W) argmax_take(acc[m][t][r], oy * p.W + ox, bv, bi);
                }
            }
#pragma unroll
            for (int off = 16; off <= 32; off <<= 1) {
                const float ov = __shfl_xor(bv, off);
                const int oi = __shfl_xor(bi, off);
                argmax_take(ov, oi, bv, bi);
            }
            if (g == 0) { pv[wave * (M * 16) + m * 16 + i16] = bv; pi[wave * (M * 16) + m * 16 + i16] = bi; }
        }
        __syncthreads();
        if (tid < M * 16 && tid < p.K) {
            float bv = pv[tid];
            int bi = pi[tid];
#pragma unroll
            for (int w = 1; w < 4; ++w) argmax_take(pv[w * (M * 16) + tid], pi[w * (M * 16) + tid], bv, bi);
            const int tile_id = ty * tiles_x + tx;
            p.part[((size_t)n * p.K + tid) * (size_t)(tiles_x * tiles_y) + tile_id] = make_float2(bv, __int_as_float(bi));
        }
    }
}

template <int CG, int M>
int launch_final_mfma_t(const FinalParams& p, hipStream_t stream) {
    constexpr int NCH = (9 * CG + 3) / 4;
    const int tiles_x = (p.W + MTW - 1) / MTW, tiles_y = (p.H + MTH - 1) / MTH;
    const long long nblk = (long long)p.N * tiles_x * tiles_y;
    if (nblk <= 0 || nblk > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    const int lds = 2 * CG * MPLANE + (final_wreg(CG, M) ? 0 : M * NCH * 2048) + MSRC * MSRC * CG * 32 + 4 * M * 16 * 8;
    auto kern = final_mfma_kernel<CG, M>;
    if (const int e_ = ensure_dyn_lds(reinterpret_cast<const void*>(kern), lds)) return e_;
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), lds, stream, p, p.wpk, tiles_x, tiles_y);
    return (int)hipGetLastError();
}

static inline uint16_t fb16(float f) {
    uint32_t u;
    __builtin_memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static inline float fb16f(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
}

}  // namespace

// padded output-channel count the weights/bias of the final conv must be packed with
int final_kt(int K) { return K <= 11 ? 11 : (K <= 16 ? 16 : (K <= 32 ? 32 : -1)); }

// MFMA path: channel groups CG = ceil((K+cin)/8) in {2..5}, M = ceil(K/16) in {1,2}
bool final_mfma_supported(int K, int cin) {
    const int cg = (K + cin + 7) / 8;
    return K >= 1 && K <= 32 && cg >= 2 && cg <= 5;
}
int final_part_tiles(int K, int cin, int H, int W) {
    if (!final_mfma_supported(K, cin) || H <= 0 || W <= 0) return 0;
    return ((W + MTW - 1) / MTW) * ((H + MTH - 1) / MTH);
}
size_t final_mfma_bytes(int K, int cin) {
    const int cg = (K + cin + 7) / 8, m = (K + 15) / 16, nch = (9 * cg + 3) / 4;
    return (size_t)m * nch * 2048;
}
// w: [K][K+cin][3][3] (reference layout) -> [m][chunk][hi|lo][lane][8]: lane (cout = m*16 + (l&15), g = l>>4),
// element j <-> K-slot group G = chunk*4 + g = tap*CG + cg, channel cg*8 + j
void pack_final_mfma(const float* w, int K, int cin, void* dst) {
    uint16_t* d = static_cast<uint16_t*>(dst);
    const int ct = K + cin, cg_n = (ct + 7) / 8, m_n = (K + 15) / 16, nch = (9 * cg_n + 3) / 4;
    for (int m = 0; m < m_n; ++m)
        for (int c = 0; c < nch; ++c)
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 8; ++j) {
                    const int co = m * 16 + (l & 15), G = c * 4 + (l >> 4);
                    float v = 0.f;
                    if (G < 9 * cg_n) {
                        const int tap = G / cg_n, ch = (G % cg_n) * 8 + j;
                        if (co < K && ch < ct) v = w[((size_t)co * ct + ch) * 9 + tap];
                    }
                    const uint16_t hi = fb16(v), lo = fb16(v - fb16f(hi));
                    const size_t base = (((size_t)m * nch + c) * 2) * 512;
                    d[base + l * 8 + j] = hi;
                    d[base + 512 + l * 8 + j] = lo;
                }
}

int launch_final(const FinalParams& p, hipStream_t stream) {
    if (p.wpk && final_mfma_supported(p.K, p.cin)) {
        const int cg = (p.K + p.cin + 7) / 8, m = (p.K + 15) / 16;
        switch (cg * 10 + m) {
            case 21: return launch_final_mfma_t<2, 1>(p, stream);
            case 31: return launch_final_mfma_t<3, 1>(p, stream);
            case 32: return launch_final_mfma_t<3, 2>(p, stream);
            case 42: return launch_final_mfma_t<4, 2>(p, stream);
            case 52: return launch_final_mfma_t<5, 2>(p, stream);
        }
    }
    switch (final_kt(p.K)) {
        case 11: return launch_final_t<11>(p, stream);
        case 16: return launch_final_t<16>(p, stream);
        case 32: return launch_final_t<32>(p, stream);
    }
    return (int)hipErrorInvalidValue;
}

}  // namespace esa
