// conv_mfma.hip — im2col-free 3x3 / 1x1 convolution (+ folded-BN bias, residual, ReLU) on the
// gfx950 matrix cores in split-bf16 ("bf16x3") arithmetic.
//
// Replaces (reference, cuDNN via ATen): every nn.Conv2d + BatchNorm2d + ReLU (+ residual add) of
// models/seg_hrnet.py except the stem conv1 and output_layer (BasicBlock :45-61, transitions
// :343-377, fuse layers :176-220, last_layer :313-329).
//
// Formulation.  Implicit GEMM  D[cout][pixel] += W[cout][cin] * X[cin][pixel+tap]  summed over
// the k*k taps and 32-channel chunks of cin; nothing like an im2col matrix ever exists: a
// workgroup stages ONE halo'd input tile per chunk in LDS and every tap is just a different
// LDS address of the same tile.  Each f32 product a*b is evaluated as
//     a_hi*b_hi + a_hi*b_lo + a_lo*b_hi      (a = a_hi + a_lo, both bf16; f32 accumulate)
// = three v_mfma_f32_16x16x32_bf16 per 16x16x32 tile: ~16 mantissa bits per operand, heatmap
// L_inf ~1e-5 against the fp32 reference (oracle/emulate_split_bf16.py), at 16/3 = 5.3x the
// f32-MFMA rate.  Activations (SB layout, sb.h) and weights are pre-split, so the inner loop
// contains no conversion: LDS -> ds_read_b128 -> MFMA.
//
// Tiling.  Workgroup = 256 threads = 4 waves; output tile = TH rows x 16 columns of pixels x
// (16*MT) output channels.  Wave w owns rows [w*TH/4, (w+1)*TH/4) and all MT cout tiles:
// accumulators acc[MT][NT] of one 16(cout) x 16(pixel) MFMA tile each.
//
// LDS image of the input tile: 8 planes (k-group g = 0..3 x part hi/lo), each [IH*IW] pixels x
// 16 B.  A wave's B-operand read (lane l -> pixel l&15, k-group l>>4) then touches 16
// consecutive 16-B slots per plane pair, which is conflict-free for ds_read_b128's lane
// groups when plane strides are multiples of 256 B (stride-2 convs: odd k-groups are shifted by
// one slot so that even/odd pixels of the two k-groups in a lane group interleave).
// The staging writes use a diagonal (pixel, chunk) -> lane map so that the 8 lanes of one
// ds_write_b128 group hit 8 different slots while the global reads still cover whole 128-B
// lines.
//
// Pipeline.  The input tile of chunk c+1 is loaded into registers before the MFMAs of chunk c
// and written to LDS after them (issue-early / write-late); the weight fragments of a chunk go
// global -> LDS by DMA (global_load_lds, no registers); two workgroups per CU cover each
// other's staging phases.
#include "kernels.h"
#include "sb.h"

namespace esa {

namespace {

constexpr int TW = 16;        // output columns per workgroup tile = one MFMA N-tile
constexpr int NTHREADS = 256;

// 16 B per lane global -> LDS DMA; LDS destination = wave-uniform base (+ lane*16 by hardware).
__device__ __forceinline__ void dma16(const void* gsrc, char* lds_uniform_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_uniform_base, 16, 0, 0);
}

template <int KS, int S, int TH, int MT>
struct ConvCfg {
    static constexpr int PAD = (KS - 1) / 2;
    static constexpr int TAPS = KS * KS;
    static constexpr int NT = TH / 4;                       // pixel-row tiles per wave
    static constexpr int IH = (TH - 1) * S + KS;
    static constexpr int IW = (TW - 1) * S + KS;
    static constexpr int NPIX = IH * IW;
    static constexpr int PLANE = ((NPIX * 16 + 16 + 255) / 256) * 256;
    static constexpr int XBYTES = 8 * PLANE;
    static constexpr int WBYTES = TAPS * MT * 2048;
    static constexpr int XUNITS = ((NPIX + 7) / 8) * 64;    // 16-B units, whole 8-pixel groups
    static constexpr int XITER = (XUNITS + NTHREADS - 1) / NTHREADS;
    static constexpr int WUNITS = TAPS * MT * 128;
    static constexpr int WITER = (WUNITS + NTHREADS - 1) / NTHREADS;
    static constexpr int LDS_BYTES = XBYTES + WBYTES;
    __host__ __device__ static constexpr int plane_off(int j) {
        return j * PLANE + ((S == 2 && ((j >> 1) & 1)) ? 16 : 0);
    }
};

template <int KS, int S, int TH, int MT>
__global__ __launch_bounds__(NTHREADS, 2) void conv_mfma_kernel(ConvParams p, int tiles_x,
                                                               int tiles_y, int ctiles) {
    using C = ConvCfg<KS, S, TH, MT>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* xs = smem;
    char* wsm = smem + C::XBYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;

    int b = blockIdx.x;
    const int ct = b % ctiles; b /= ctiles;
    const int tx = b % tiles_x; b /= tiles_x;
    const int ty = b % tiles_y;
    const int n = b / tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int nchunks = p.Cinp >> 5;

    // ---- staging maps (chunk-invariant) ----------------------------------------------------
    // X: unit u -> 8-pixel group (u>>6), lane (r = lane>>3, t = lane&7): pixel q = grp*8 + t,
    // 16-B chunk j = (t + r) & 7 of that pixel's 128-B channel chunk (diagonal map, see header).
    int xg[C::XITER];      // byte offset of (pixel, j) inside this image, -1 = zero padding
    const int pix_stride = p.Cinp * 4;
    const int jst = ((lane & 7) + (lane >> 3)) & 7;
    const int q0 = wave * 8 + (lane & 7);                    // pixel of iteration 0; +32 per iteration
#pragma unroll
    for (int it = 0; it < C::XITER; ++it) {
        const int q = q0 + it * 32;
        const int qy = q / C::IW, qx = q - qy * C::IW;
        const int gy = oy0 * S - C::PAD + qy, gx = ox0 * S - C::PAD + qx;
        const bool inside = q < C::NPIX && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
        xg[it] = inside ? ((gy * p.W + gx) * pix_stride + jst * 16) : -1;
    }
    char* xwr = xs + C::plane_off(jst) + q0 * 16;             // LDS write address, +512 per iteration
    const char* xn = p.x + (size_t)n * p.H * p.W * pix_stride;   // 32-bit offsets inside one image
    const uint4* wbase = p.w + (size_t)(ct * MT) * nchunks * (C::TAPS * 128);

    uint4 xr[C::XITER];
    // issue-early half of the X staging: global -> registers
#define ESA_PREFETCH_X(CH)                                                                        \
    {                                                                                             \
        _Pragma("unroll") for (int it = 0; it < C::XITER; ++it) {                                 \
            uint4 v = make_uint4(0, 0, 0, 0);                                                     \
            if (xg[it] >= 0) v = *reinterpret_cast<const uint4*>(xn + xg[it] + (CH) * 128);       \
            xr[it] = v;                                                                           \
        }                                                                                         \
    }
    // write-late half: registers -> LDS
#define ESA_COMMIT_X()                                                                            \
    {                                                                                             \
        _Pragma("unroll") for (int it = 0; it < C::XITER; ++it)                                   \
            if (q0 + it * 32 < C::NPIX) *reinterpret_cast<uint4*>(xwr + it * 512) = xr[it];       \
    }
    // W: straight copy of [MT][TAPS][hi/lo][1 KB] fragments, global -> LDS DMA (no registers);
    // one 1-KB piece per wave-instruction, LDS destination = wave-uniform base + lane*16.
#define ESA_DMA_W(CH)                                                                             \
    {                                                                                             \
        _Pragma("unroll") for (int it = 0; it < C::WITER; ++it) {                                 \
            const int ub = (it * 4 + wave) * 64;                                                  \
            if (ub < C::WUNITS) {                                                                 \
                const int mt = ub / (C::TAPS * 128), rem = ub - mt * (C::TAPS * 128);             \
                const uint4* src = wbase + ((size_t)mt * nchunks + (CH)) * (C::TAPS * 128) + rem + lane; \
                dma16(src, wsm + __builtin_amdgcn_readfirstlane(ub) * 16);                        \
            }                                                                                     \
        }                                                                                         \
    }

    f32x4 acc[MT][C::NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < C::NT; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    // per-lane LDS read bases
    const int g = lane >> 4;
    const char* xrd = xs + C::plane_off(2 * g) + ((wave * C::NT * S) * C::IW + (lane & 15) * S) * 16;
    const char* wrd = wsm + lane * 16;
    constexpr int ROWS = (C::NT - 1) * S + KS;               // input rows one wave touches

    ESA_PREFETCH_X(0)
    for (int c = 0; c < nchunks; ++c) {
        if (c) __syncthreads();          // everyone finished reading the previous chunk
        ESA_DMA_W(c)
        ESA_COMMIT_X()
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the W DMA of this wave has landed
        __syncthreads();
        if (c + 1 < nchunks) ESA_PREFETCH_X(c + 1)
        // kx-major order: an input-row fragment (row i, column shift kx) feeds up to KS taps
        // (output rows i-ky), so each fragment is read from LDS once instead of KS times.
#pragma unroll
        for (int kx = 0; kx < KS; ++kx) {
            bf16x8 wh[KS][MT], wl[KS][MT];
#pragma unroll
            for (int ky = 0; ky < KS; ++ky)
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    wh[ky][m] = *reinterpret_cast<const bf16x8*>(wrd + ((m * C::TAPS + ky * KS + kx) * 2 + 0) * 1024);
                    wl[ky][m] = *reinterpret_cast<const bf16x8*>(wrd + ((m * C::TAPS + ky * KS + kx) * 2 + 1) * 1024);
                }
#pragma unroll
            for (int i = 0; i < ROWS; ++i) {
                const int off = (i * C::IW + kx) * 16;
                const bf16x8 xh = *reinterpret_cast<const bf16x8*>(xrd + off);
                const bf16x8 xo = *reinterpret_cast<const bf16x8*>(xrd + off + C::PLANE);
#pragma unroll
                for (int ky = 0; ky < KS; ++ky) {
                    const int d = i - ky;
                    if (d >= 0 && d % S == 0 && d / S < C::NT) {
                        const int t = d / S;
#pragma unroll
                        for (int m = 0; m < MT; ++m) {
                            acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[ky][m], xh, acc[m][t], 0, 0, 0);
                            acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[ky][m], xo, acc[m][t], 0, 0, 0);
                            acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[ky][m], xh, acc[m][t], 0, 0, 0);
                        }
                    }
                }
            }
        }
    }
#undef ESA_PREFETCH_X
#undef ESA_COMMIT_X
#undef ESA_DMA_W

    // ---- epilogue: bias, residual, ReLU, split, store -----------------------------------------
    // D tile: column (lane&15) = pixel, rows (lane>>4)*4 + r = cout  -> 4 consecutive channels.
    const int ox = ox0 + (lane & 15);
    const int opix = p.Coutp * 4;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int co = (ct * MT + m) * 16 + g * 4;                 // first of this lane's 4 couts
        const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + co);
        const int cofs = (co >> 3) * 32 + ((co >> 2) & 1) * 8;     // byte offset inside the pixel
#pragma unroll
        for (int t = 0; t < C::NT; ++t) {
            const int oy = oy0 + wave * C::NT + t;
            if (oy < p.OH && ox < p.OW) {
                const size_t o = ((size_t)(n * p.OH + oy) * p.OW + ox) * opix + cofs;
                float v[4] = {acc[m][t][0] + bv[0], acc[m][t][1] + bv[1], acc[m][t][2] + bv[2],
                              acc[m][t][3] + bv[3]};
                if (p.res) {
                    const uint2 rh = *reinterpret_cast<const uint2*>(p.res + o);
                    const uint2 rl = *reinterpret_cast<const uint2*>(p.res + o + 16);
                    float r[4];
                    join4(rh, rl, r);
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] += r[i];
                }
                if (p.relu) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
                }
                if (p.out_f32) {
                    const size_t of = ((size_t)(n * p.OH + oy) * p.OW + ox) * opix + (size_t)co * 4;
                    *reinterpret_cast<f32x4*>(p.y + of) = f32x4{v[0], v[1], v[2], v[3]};
                } else {
                    uint2 hi, lo;
                    split4(v, hi, lo);
                    *reinterpret_cast<uint2*>(p.y + o) = hi;
                    *reinterpret_cast<uint2*>(p.y + o + 16) = lo;
                }
            }
        }
    }
}

template <int KS, int S, int TH, int MT>
int launch_t(const ConvParams& p, hipStream_t stream) {
    using C = ConvCfg<KS, S, TH, MT>;
    static bool attr_set = false;
    auto kern = conv_mfma_kernel<KS, S, TH, MT>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const int tiles_x = (p.OW + TW - 1) / TW, tiles_y = (p.OH + TH - 1) / TH;
    const int ctiles = p.Coutp / (16 * MT);
    const long long nblk = (long long)p.N * tiles_y * tiles_x * ctiles;
    if (nblk <= 0 || nblk > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(NTHREADS), C::LDS_BYTES, stream, p, tiles_x,
                       tiles_y, ctiles);
    return (int)hipGetLastError();
}

}  // namespace

int launch_conv(const ConvParams& p, int k, int stride, hipStream_t stream) {
    if ((p.Cinp & 31) || (p.Coutp & 31)) return (int)hipErrorInvalidValue;
    // one image is addressed with 32-bit byte offsets inside the kernel
    if ((long long)p.H * p.W * p.Cinp * 4 > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    if (k == 3 && stride == 1) return launch_t<3, 1, 16, 2>(p, stream);
    if (k == 3 && stride == 2) return launch_t<3, 2, 4, 2>(p, stream);
    if (k == 1 && stride == 1) return launch_t<1, 1, 16, 2>(p, stream);
    return (int)hipErrorInvalidValue;
}

size_t packed_weight_bytes(int coutp, int cinp, int k) {
    return (size_t)(coutp / 16) * (cinp / 32) * k * k * 2048;
}

static inline uint16_t host_bf16(float f) {
    uint32_t u;
    __builtin_memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);       // round to nearest even (finite inputs)
    return (uint16_t)(u >> 16);
}
static inline float host_bf16_to_f32(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
}

// Layout: [cout16 tile][cin32 chunk][tap][part hi/lo][lane 0..63][8 x bf16], lane l holding
// W[cout = tile*16 + (l&15)][cin = chunk*32 + 8*(l>>4) + j] — the MFMA 16x16x32 A-operand map.
void pack_conv_weights(const float* w, int cout, int cin, int k, int coutp, int cinp, void* dst) {
    uint16_t* d = static_cast<uint16_t*>(dst);
    const int taps = k * k, nch = cinp / 32;
    for (int t16 = 0; t16 < coutp / 16; ++t16)
        for (int c = 0; c < nch; ++c)
            for (int tap = 0; tap < taps; ++tap)
                for (int l = 0; l < 64; ++l)
                    for (int j = 0; j < 8; ++j) {
                        const int co = t16 * 16 + (l & 15), ci = c * 32 + 8 * (l >> 4) + j;
                        float v = 0.f;
                        if (co < cout && ci < cin) v = w[((size_t)co * cin + ci) * taps + tap];
                        const uint16_t hi = host_bf16(v);
                        const uint16_t lo = host_bf16(v - host_bf16_to_f32(hi));
                        const size_t base = ((((size_t)t16 * nch + c) * taps + tap) * 2) * 512;
                        d[base + l * 8 + j] = hi;
                        d[base + 512 + l * 8 + j] = lo;
                    }
}

}  // namespace esa
