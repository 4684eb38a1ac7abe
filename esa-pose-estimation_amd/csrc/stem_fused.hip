// stem_fused.hip — the whole stem in one kernel:
//   conv1 3x3 s1 (Cin -> 64) + bn1 + ReLU  ->  conv2 3x3 s2 (64 -> 64) + bn2 + ReLU
// f32 NCHW crop in, SB [N][H/2][W/2][64] out.
//
// Replaces conv1/bn1/relu/conv2/bn2/relu of models/seg_hrnet.py:265-270, 426-431.  The 64-channel
// full-resolution tensor between the two convolutions (16.8 MB per crop, the largest tensor of the
// network: 537 MB written and read again per 32-crop batch) is never materialised: a workgroup
// recomputes the 9 x 33 conv1 pixels its stride-2 output tile needs on the f32 VALU (K = 9*Cin is
// far too short for the matrix cores), writes them split-bf16 straight into the LDS operand planes
// of the MFMA convolution, and runs conv2 from there exactly like conv_mfma<3,2,4,2>.
//
// Work split of the conv1 phase: wave w owns the 8-channel k-group w of the current 32-channel
// chunk (so its 72*Cin weights are wave-uniform -> scalar registers), lanes walk the 297 pixels.
// conv1 pixels outside the image are forced to 0 (they are conv2's zero padding, not conv1
// evaluated on padding).
#include "conv_cfg.h"
#include "devstate.h"
#include "kernels.h"
#include "sb.h"

#ifndef SF_DO_CONV1
#define SF_DO_CONV1 1
#endif
#ifndef SF_DO_MFMA
#define SF_DO_MFMA 1
#endif
#ifndef SF_DO_WDMA
#define SF_DO_WDMA 1
#endif

namespace esa {
namespace {

constexpr int SF_TH = 4, SF_MT = 2;
using SC = ConvCfg<3, 2, SF_TH, SF_MT>;
constexpr int RH = SC::IH + 2, RW = SC::IW + 2;          // raw crop tile (conv1 halo): 11 x 35
constexpr int RPITCH = RW + 1;

template <int CIN, int CT>
__global__ __launch_bounds__(NTHREADS, 2) void stem_fused_kernel(StemFusedParams p,
                                                                const float* __restrict__ w1g,
                                                                const float* __restrict__ b1g, int tiles_x,
                                                                int tiles_y, int ctiles) {
    // w1g/b1g (= p.w1/p.bias1) are separate const __restrict__ kernel arguments so that the compiler
    // may treat the wave-uniform conv1 weights as invariant and fetch them with scalar loads.
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* xs = smem;
    char* wsm = smem + SC::XBYTES;
    float* raw = reinterpret_cast<float*>(smem + SC::XBYTES + SC::WBYTES);   // [CIN][RH][RPITCH]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int b = xcd_contiguous(blockIdx.x, gridDim.x);
    const int ct = b % ctiles; b /= ctiles;
    const int tx = b % tiles_x; b /= tiles_x;
    const int ty = b % tiles_y;
    const int n = b / tiles_y;
    const int oy0 = ty * SF_TH, ox0 = tx * TW;
    const int nchunks = p.Cmid >> 5;
    const int my0 = oy0 * 2 - 1, mx0 = ox0 * 2 - 1;       // conv1-output (mid) coords of LDS pixel (0,0)

    // ---- raw crop tile -> LDS (zero padded) -----------------------------------------------------
    for (int u = tid; u < CIN * RH * RW; u += NTHREADS) {
        const int rx = u % RW;
        const int r = u / RW;
        const int ry = r % RH, ci = r / RH;
        const int gy = my0 - 1 + ry, gx = mx0 - 1 + rx;
        float v = 0.f;
        if (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W)
            v = p.x[(((size_t)n * CIN + ci) * p.H + gy) * p.W + gx];
        raw[(ci * RH + ry) * RPITCH + rx] = v;
    }

    // CT cout tiles (of 16*SF_MT channels) per workgroup share one conv1 evaluation
    f32x4 acc[CT][SF_MT];
    const int g = lane >> 4;
    {
#pragma unroll
        for (int k = 0; k < CT; ++k)
#pragma unroll
            for (int m = 0; m < SF_MT; ++m)
                acc[k][m] = *reinterpret_cast<const f32x4*>(p.bias2 + ((ct * CT + k) * SF_MT + m) * 16 + g * 4);
    }
    const char* xrd = xs + SC::plane_off(2 * g) + ((wave * 2) * SC::IW + (lane & 15) * 2) * 16;
    const char* wrd = wsm + lane * 16;
    const int wv = __builtin_amdgcn_readfirstlane(wave);

#define SF_DMA_W(K, CH)                                                                            \
    {                                                                                              \
        const uint4* wbase = p.w2 + (size_t)((ct * CT + (K)) * SF_MT) * nchunks * (SC::TAPS * 128);  \
        _Pragma("unroll") for (int it = 0; it < SC::WITER; ++it) {                                 \
            const int ub = (it * 4 + wave) * 64;                                                   \
            if (SF_DO_WDMA && ub < SC::WUNITS) {                                                   \
                const int mt = ub / (SC::TAPS * 128), rem = ub - mt * (SC::TAPS * 128);            \
                const uint4* src = wbase + ((size_t)mt * nchunks + (CH)) * (SC::TAPS * 128) + rem + lane; \
                dma16(src, wsm + __builtin_amdgcn_readfirstlane(ub) * 16);                         \
            }                                                                                      \
        }                                                                                          \
    }
    for (int c = 0; c < nchunks; ++c) {
        __syncthreads();                 // raw tile ready (c == 0) / previous chunk's MFMAs done
        SF_DMA_W(0, c)                   // lands while the VALU evaluates conv1
        // ---- conv1 for channels [32c + 8*wave, +8) of all 297 tile pixels -> LDS operand planes ----
        {
            const float* __restrict__ w1 = w1g + (size_t)((c * 4 + wv) * CIN) * 72;   // [cin][9][8], wave-uniform
            const float* __restrict__ b1 = b1g + (c * 4 + wv) * 8;
            char* ph = xs + SC::plane_off(2 * wv);
            char* pl = xs + SC::plane_off(2 * wv + 1);
            constexpr int QIT = (SC::NPIX + 63) / 64;                  // 5 pixels per lane
            int rbase[QIT];
            bool inside[QIT];
            float a[QIT][8];
#pragma unroll
            for (int it = 0; it < QIT; ++it) {
                const int q = min(lane + it * 64, SC::NPIX - 1);
                const int qy = q / SC::IW, qx = q - qy * SC::IW;
                rbase[it] = qy * RPITCH + qx;
                const int my = my0 + qy, mx = mx0 + qx;
                inside[it] = my >= 0 && my < p.H && mx >= 0 && mx < p.W;
#pragma unroll
                for (int i = 0; i < 8; ++i) a[it][i] = b1[i];
            }
            // tap-major: 8 wave-uniform weights at a time against the lane's 5 pixels; the input-channel
            // loop stays rolled so that only 72 weights (one channel) sit in scalar registers at a time
#pragma unroll 1
            for (int ci = 0; ci < (SF_DO_CONV1 ? CIN : 0); ++ci)
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    float wt[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) wt[i] = w1[(ci * 9 + tap) * 8 + i];
#pragma unroll
                    for (int it = 0; it < QIT; ++it) {
                        const float v = raw[ci * RH * RPITCH + rbase[it] + (tap / 3) * RPITCH + (tap % 3)];
#pragma unroll
                        for (int i = 0; i < 8; ++i) a[it][i] = fmaf(v, wt[i], a[it][i]);
                    }
                }
#pragma unroll
            for (int it = 0; it < QIT; ++it) {
                const int q = lane + it * 64;
#pragma unroll
                for (int i = 0; i < 8; ++i) a[it][i] = inside[it] ? relu1(a[it][i]) : 0.f;
                uint4 hi, lo;
                split8(a[it], hi, lo);
                if (q < SC::NPIX) {
                    *reinterpret_cast<uint4*>(ph + q * 16) = hi;
                    *reinterpret_cast<uint4*>(pl + q * 16) = lo;
                }
            }
        }
        // ---- conv2 on the matrix cores, one cout tile at a time (weights: global -> LDS DMA) ---------
#pragma unroll
        for (int k = 0; k < CT; ++k) {
            if (k) {
                __syncthreads();         // previous cout tile's MFMAs done reading the weight buffer
                SF_DMA_W(k, c)
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the W DMA of this wave has landed
            __syncthreads();             // (k == 0: also publishes the conv1 planes)
#pragma unroll
            for (int kx = 0; kx < (SF_DO_MFMA ? 3 : 0); ++kx) {
                bf16x8 wh[3][SF_MT], wl[3][SF_MT];
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int m = 0; m < SF_MT; ++m) {
                        wh[ky][m] = *reinterpret_cast<const bf16x8*>(wrd + ((m * 9 + ky * 3 + kx) * 2 + 0) * 1024);
                        wl[ky][m] = *reinterpret_cast<const bf16x8*>(wrd + ((m * 9 + ky * 3 + kx) * 2 + 1) * 1024);
                    }
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int off = (ky * SC::IW + kx) * 16;
                    const bf16x8 xh = *reinterpret_cast<const bf16x8*>(xrd + off);
                    const bf16x8 xo = *reinterpret_cast<const bf16x8*>(xrd + off + SC::LO_OFF);
#pragma unroll
                    for (int m = 0; m < SF_MT; ++m) {
                        acc[k][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[ky][m], xh, acc[k][m], 0, 0, 0);
                        acc[k][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[ky][m], xo, acc[k][m], 0, 0, 0);
                        acc[k][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[ky][m], xh, acc[k][m], 0, 0, 0);
                    }
                }
            }
        }
    }

#undef SF_DMA_W
    // ---- epilogue: ReLU, split, store ---------------------------------------------------------------
    const int ox = ox0 + (lane & 15), oy = oy0 + wave;
    const bool inr = oy < p.OH && ox < p.OW;
#pragma unroll
    for (int km = 0; km < CT * SF_MT; ++km) {
        const int k = km / SF_MT, m = km % SF_MT;
        const int co = ((ct * CT + k) * SF_MT + m) * 16 + g * 4;
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = relu1(acc[k][m][i]);
        uint2 hi, lo;
        split4(v, hi, lo);
        const uint4 ch = quad_to_chunk(hi, lo);                    // all lanes; only the store is predicated
        if (inr) *reinterpret_cast<uint4*>(p.y + ((size_t)(n * p.OH + oy) * p.OW + ox) * (size_t)(p.Coutp * 4) + chunk_ofs(co, g)) = ch;
    }
}

// ---------------------------------------------------------------------------------------------------
// Coutp == 64 (the reference stem): one cout tile per wave instead of one output row per wave.  A wave's
// 18 weight fragments per chunk are then private — loaded global -> registers while the VALU evaluates
// conv1, never through LDS (no weight DMA, no weight barriers, 42 KB of LDS instead of 79 KB: three
// workgroups per CU) — and every input-row fragment it reads from LDS feeds 4 output rows' taps
// (54 LDS reads per 108 MFMAs instead of 108).
#ifndef SF_OCC
#define SF_OCC 2
#endif
template <int CIN>
__global__ __launch_bounds__(NTHREADS, SF_OCC) void stem_fused64_kernel(StemFusedParams p,
                                                                  const float* __restrict__ w1g,
                                                                  const float* __restrict__ b1g, int tiles_x,
                                                                  int tiles_y) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* xs = smem;
    float* raw = reinterpret_cast<float*>(smem + SC::XBYTES);               // [CIN][RH][RPITCH]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int b = xcd_contiguous(blockIdx.x, gridDim.x);
    const int tx = b % tiles_x; b /= tiles_x;
    const int ty = b % tiles_y;
    const int n = b / tiles_y;
    const int oy0 = ty * SF_TH, ox0 = tx * TW;
    const int nchunks = p.Cmid >> 5;
    const int my0 = oy0 * 2 - 1, mx0 = ox0 * 2 - 1;

    for (int u = tid; u < CIN * RH * RW; u += NTHREADS) {
        const int rx = u % RW;
        const int r = u / RW;
        const int ry = r % RH, ci = r / RH;
        const int gy = my0 - 1 + ry, gx = mx0 - 1 + rx;
        float v = 0.f;
        if (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W)
            v = p.x[(((size_t)n * CIN + ci) * p.H + gy) * p.W + gx];
        raw[(ci * RH + ry) * RPITCH + rx] = v;
    }

    const int g = lane >> 4;
    f32x4 acc[SF_TH];
    {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias2 + wave * 16 + g * 4);
#pragma unroll
        for (int t = 0; t < SF_TH; ++t) acc[t] = bv;
    }
    const char* xrd = xs + SC::plane_off(2 * g) + ((lane & 15) * 2) * 16;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
    const u32x4* wsrc = reinterpret_cast<const u32x4*>(p.w2) + (size_t)wv * nchunks * (SC::TAPS * 128) + lane;

    for (int c = 0; c < nchunks; ++c) {
        // this wave's conv2 weights of the chunk: [tap][hi|lo], in flight during the conv1 phase
        u32x4 wf[SC::TAPS][2];
#pragma unroll
        for (int tap = 0; tap < SC::TAPS; ++tap)
#pragma unroll
            for (int part = 0; part < 2; ++part)
                wf[tap][part] = SF_DO_WDMA ? wsrc[(size_t)c * (SC::TAPS * 128) + (tap * 2 + part) * 64] : u32x4{0, 0, 0, 0};
        __syncthreads();                 // raw tile ready (c == 0) / previous chunk's MFMAs done
        {
            const float* __restrict__ w1 = w1g + (size_t)((c * 4 + wv) * CIN) * 72;
            const float* __restrict__ b1 = b1g + (c * 4 + wv) * 8;
            char* ph = xs + SC::plane_off(2 * wv);
            char* pl = xs + SC::plane_off(2 * wv + 1);
            constexpr int QIT = (SC::NPIX + 63) / 64;
            int rbase[QIT];
            bool inside[QIT];
            float a[QIT][8];
#pragma unroll
            for (int it = 0; it < QIT; ++it) {
                const int q = min(lane + it * 64, SC::NPIX - 1);
                const int qy = q / SC::IW, qx = q - qy * SC::IW;
                rbase[it] = qy * RPITCH + qx;
                const int my = my0 + qy, mx = mx0 + qx;
                inside[it] = my >= 0 && my < p.H && mx >= 0 && mx < p.W;
#pragma unroll
                for (int i = 0; i < 8; ++i) a[it][i] = b1[i];
            }
#pragma unroll 1
            for (int ci = 0; ci < (SF_DO_CONV1 ? CIN : 0); ++ci)
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    float wt[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) wt[i] = w1[(ci * 9 + tap) * 8 + i];
#pragma unroll
                    for (int it = 0; it < QIT; ++it) {
                        const float v = raw[ci * RH * RPITCH + rbase[it] + (tap / 3) * RPITCH + (tap % 3)];
#pragma unroll
                        for (int i = 0; i < 8; ++i) a[it][i] = fmaf(v, wt[i], a[it][i]);
                    }
                }
#pragma unroll
            for (int it = 0; it < QIT; ++it) {
                const int q = lane + it * 64;
#pragma unroll
                for (int i = 0; i < 8; ++i) a[it][i] = inside[it] ? relu1(a[it][i]) : 0.f;
                uint4 hi, lo;
                split8(a[it], hi, lo);
                if (q < SC::NPIX) {
                    *reinterpret_cast<uint4*>(ph + q * 16) = hi;
                    *reinterpret_cast<uint4*>(pl + q * 16) = lo;
                }
            }
        }
        __syncthreads();                 // conv1 planes of this chunk are complete
        // conv2: cout tile `wave`, all SF_TH output rows; row i of the tile feeds output row t = (i - ky) / 2
#pragma unroll
        for (int kx = 0; kx < (SF_DO_MFMA ? 3 : 0); ++kx)
#pragma unroll
            for (int i = 0; i < SC::IH; ++i) {
                const int off = (i * SC::IW + kx) * 16;
                const bf16x8 xh = *reinterpret_cast<const bf16x8*>(xrd + off);
                const bf16x8 xo = *reinterpret_cast<const bf16x8*>(xrd + off + SC::LO_OFF);
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int d = i - ky;
                    if (d >= 0 && (d & 1) == 0 && d / 2 < SF_TH) {
                        const int t = d / 2;
                        const bf16x8 wh = __builtin_bit_cast(bf16x8, wf[ky * 3 + kx][0]);
                        const bf16x8 wl = __builtin_bit_cast(bf16x8, wf[ky * 3 + kx][1]);
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, xh, acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xo, acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xh, acc[t], 0, 0, 0);
                    }
                }
            }
    }
    // ---- epilogue: ReLU, split, 16-byte chunk stores ------------------------------------------------
    const int ox = ox0 + (lane & 15);
#pragma unroll
    for (int t = 0; t < SF_TH; ++t) {
        const int oy = oy0 + t;
        const int co = wave * 16 + g * 4;
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = relu1(acc[t][i]);
        uint2 hi, lo;
        split4(v, hi, lo);
        const uint4 ch = quad_to_chunk(hi, lo);
        if (oy < p.OH && ox < p.OW)
            *reinterpret_cast<uint4*>(p.y + ((size_t)(n * p.OH + oy) * p.OW + ox) * (size_t)(p.Coutp * 4) + chunk_ofs(co, g)) = ch;
    }
}

template <int CIN>
int launch_sf64(const StemFusedParams& p, hipStream_t stream) {
    auto kern = stem_fused64_kernel<CIN>;
    const int lds = SC::XBYTES + CIN * RH * RPITCH * (int)sizeof(float);
    if (const int e_ = ensure_dyn_lds(reinterpret_cast<const void*>(kern), lds)) return e_;
    const int tiles_x = (p.OW + TW - 1) / TW, tiles_y = (p.OH + SF_TH - 1) / SF_TH;
    const long long nblk = (long long)p.N * tiles_x * tiles_y;
    if (nblk <= 0 || nblk > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(NTHREADS), lds, stream, p, p.w1, p.bias1, tiles_x, tiles_y);
    return (int)hipGetLastError();
}

template <int CIN, int CT>
int launch_sf2(const StemFusedParams& p, hipStream_t stream) {
    auto kern = stem_fused_kernel<CIN, CT>;
    const int lds = SC::XBYTES + SC::WBYTES + CIN * RH * RPITCH * (int)sizeof(float);
    if (const int e_ = ensure_dyn_lds(reinterpret_cast<const void*>(kern), lds)) return e_;
    const int tiles_x = (p.OW + TW - 1) / TW, tiles_y = (p.OH + SF_TH - 1) / SF_TH;
    const int ctiles = p.Coutp / (16 * SF_MT * CT);
    const long long nblk = (long long)p.N * tiles_x * tiles_y * ctiles;
    if (nblk <= 0 || nblk > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(NTHREADS), lds, stream, p, p.w1, p.bias1, tiles_x, tiles_y, ctiles);
    return (int)hipGetLastError();
}

#ifndef SF_V2
#define SF_V2 1
#endif
template <int CIN>
int launch_sf(const StemFusedParams& p, hipStream_t stream) {
    if (SF_V2 && p.Coutp == 64) return launch_sf64<CIN>(p, stream);
    // two cout tiles per workgroup halve the conv1 recomputation when the layer has an even number
    if ((p.Coutp / (16 * SF_MT)) % 2 == 0) return launch_sf2<CIN, 2>(p, stream);
    return launch_sf2<CIN, 1>(p, stream);
}

}  // namespace

int launch_stem_fused(const StemFusedParams& p, hipStream_t stream) {
    if ((p.Cmid & 31) || (p.Coutp & 31)) return (int)hipErrorInvalidValue;
    switch (p.cin) {
        case 1: return launch_sf<1>(p, stream);
        case 2: return launch_sf<2>(p, stream);
        case 3: return launch_sf<3>(p, stream);
        case 4: return launch_sf<4>(p, stream);
    }
    return (int)hipErrorInvalidValue;
}

}  // namespace esa
