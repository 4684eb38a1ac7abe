// conv_s2c32.hip — 3x3 stride-2 convolution (+ folded-BN bias, ReLU) on the matrix cores, split-bf16,
// weights in registers: the fuse-down chains and the transition layers
// (models/seg_hrnet.py:176-220 fuse_layers[i][j], j < i; :343-377 transition layers).
//
// A stride-2 conv reads four input pixels per output pixel, so the generic tiling (conv_mfma<3,2,4,2>:
// one output row per wave, 32 couts per workgroup) is LDS- and L2-bound: 54 LDS reads per 54 MFMAs, and
// the 9x33 input tile is staged once per 32-cout slice.  With a single 32-channel chunk the whole
// weight set of a 16-cout tile is 18 fragments = 72 VGPRs, so here
//   * a wave owns one cout tile for the workgroup's lifetime (persistent workgroups, cout-tile index
//     constant per workgroup): its weights are loaded from global ONCE, never touch LDS;
//   * every wave sweeps all rows of the pixel tile: each input-row fragment it reads from LDS feeds up to
//     two taps of two output rows (54 reads per 108 MFMAs at 64 couts per workgroup);
//   * the input tile is staged once per 64 (or 32) couts; the next item's tile is prefetched into
//     registers while the current one is consumed (issue-early / write-late, as in conv_mfma.hip).
// LDS holds only the 8 operand planes of the input tile (40 KB).
#include "conv_cfg.h"
#include "kernels.h"
#include "sb.h"

namespace esa {
namespace {


// MW = cout tiles (of 16) per workgroup: 4 -> wave = cout tile, 4 output rows each;
//                                        2 -> wave = (cout tile, row half), 2 output rows each
// The workgroup walks a stream of (item, chunk) steps; X of step s+1 is prefetched into registers while
// step s is consumed; the weight registers are a ring of three kx-thirds: as soon as phase kx of step s
// is done, the (kx) third of step s+1 is loaded into the same registers (3 phases of cover, no extra
// VGPRs).  Single-chunk layers whose cout slice does not change keep their weights for the whole launch.
template <int S, int TH, int MW>
__global__ __launch_bounds__(NTHREADS, 2) void conv_s2c32_kernel(ConvParams p, int tiles_x, int tiles_y, int ctiles,
                                                                int nitems) {
    using S2C = ConvCfg<3, S, TH, 2>;
    constexpr int S2_TH = TH;
    constexpr int RG = 4 / MW;                  // row groups
    constexpr int NT = S2_TH / RG;              // output rows per wave
    constexpr int ROWS = (NT - 1) * S + 3;      // input rows a wave touches
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* xs = smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4;
    const int mw = wave % MW, rg = wave / MW;
    const int G = gridDim.x;
    const int nchunks = p.Cinp >> 5;
    const int pixb = p.Cinp * 4;
    int item = xcd_contiguous(blockIdx.x, G);
    if (item >= nitems) return;

    // ---- staging map of the item being prefetched (same unit map as conv_mfma.hip) ---------------
    int xg[S2C::XITER];
    const int jst = tid & 7, q0 = tid >> 3;
    char* xwr = xs + S2C::plane_off(jst) + q0 * 16;
    const char* xn;
    int s_n, s_oy0, s_ox0, s_ct;
#define S2_DECODE(ITEM)                                                                           \
    {                                                                                             \
        int b_ = (ITEM);                                                                          \
        s_ct = b_ % ctiles; b_ /= ctiles;                                                         \
        const int tx_ = b_ % tiles_x; b_ /= tiles_x;                                              \
        const int ty_ = b_ % tiles_y;                                                             \
        s_n = b_ / tiles_y;                                                                       \
        s_oy0 = ty_ * S2_TH; s_ox0 = tx_ * TW;                                                    \
        xn = p.x + (size_t)s_n * p.H * p.W * pixb;                                                \
        _Pragma("unroll") for (int it = 0; it < S2C::XITER; ++it) {                               \
            const int q = q0 + it * 32;                                                           \
            const int qy = q / S2C::IW, qx = q - qy * S2C::IW;                                    \
            const int gy = s_oy0 * S - 1 + qy, gx = s_ox0 * S - 1 + qx;                           \
            const bool inside = q < S2C::NPIX && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;      \
            xg[it] = inside ? ((gy * p.W + gx) * pixb + jst * 16) : -1;                           \
        }                                                                                         \
    }
    typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
    u32x4 xr[S2C::XITER];
#define S2_PREFETCH(CH)                                                                           \
    {                                                                                             \
        _Pragma("unroll") for (int it = 0; it < S2C::XITER; ++it) {                               \
            u32x4 v = {0, 0, 0, 0};                                                               \
            if (xg[it] >= 0) v = *reinterpret_cast<const u32x4*>(xn + xg[it] + (CH) * 128);       \
            xr[it] = v;                                                                           \
        }                                                                                         \
    }
    // weights of (cout slice CT, chunk CH), third KX -> registers wh/wl[ky*3 + KX]
#define S2_LOAD_W(CT, CH, KX)                                                                     \
    {                                                                                             \
        const uint4* ws_ = p.w + ((size_t)((CT) * MW + mw) * nchunks + (CH)) * (9 * 128) + lane;  \
        _Pragma("unroll") for (int ky = 0; ky < 3; ++ky) {                                        \
            wh[ky * 3 + (KX)] = __builtin_bit_cast(bf16x8, ws_[((ky * 3 + (KX)) * 2 + 0) * 64]);  \
            wl[ky * 3 + (KX)] = __builtin_bit_cast(bf16x8, ws_[((ky * 3 + (KX)) * 2 + 1) * 64]);  \
        }                                                                                         \
    }

    const char* xrd = xs + S2C::plane_off(2 * g) + ((rg * NT * S) * S2C::IW + (lane & 15) * S) * 16;
    const int opix = p.Coutp * 4;
    bf16x8 wh[9], wl[9];

    S2_DECODE(item)
    S2_PREFETCH(0)
    S2_LOAD_W(s_ct, 0, 0)
    S2_LOAD_W(s_ct, 0, 1)
    S2_LOAD_W(s_ct, 0, 2)
    int n = s_n, oy0 = s_oy0, ox0 = s_ox0, ct = s_ct;      // the item being computed
    int c = 0;
    bool first = true;
    f32x4 acc[NT];
    while (true) {
        const bool last_chunk = c + 1 == nchunks;
        const bool more = !last_chunk || item + G < nitems;          // is there a step s+1?
        if (!first) __syncthreads();            // previous step's MFMAs are done reading the planes
        first = false;
#pragma unroll
        for (int it = 0; it < S2C::XITER; ++it)
            if (q0 + it * 32 < S2C::NPIX) *reinterpret_cast<u32x4*>(xwr + it * 512) = xr[it];
        __syncthreads();
        // step s+1: next chunk of this item, or chunk 0 of the workgroup's next item
        int nct = ct, nch = c + 1;
        if (last_chunk) {
            nch = 0;
            if (more) {
                S2_DECODE(item + G)
                nct = s_ct;
            }
        }
        if (more) S2_PREFETCH(nch)
        const bool reload = more && (nchunks > 1 || nct != ct);
        const int co = (ct * MW + mw) * 16 + g * 4;
        if (c == 0) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + co);
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = bv;
        }
        // residual (last chunk only): two halves of the wave's rows, each loaded a phase or two before it
        // is folded into the accumulators, as 16-byte chunks (sb.h)
        constexpr int NH = NT >= 2 ? NT / 2 : 1;
        uint4 rc[NH];
        const bool do_res = last_chunk && p.res != nullptr;
        const int rox = ox0 + (lane & 15);
#define S2_RES_LOAD(HALF)                                                                         \
        if (do_res) {                                                                             \
            _Pragma("unroll") for (int t = 0; t < NH; ++t) {                                      \
                const int oy_ = oy0 + rg * NT + (HALF) * NH + t;                                  \
                rc[t] = make_uint4(0, 0, 0, 0);                                                   \
                if ((HALF) * NH + t < NT && oy_ < p.OH && rox < p.OW)                             \
                    rc[t] = *reinterpret_cast<const uint4*>(p.res + ((size_t)(n * p.OH + oy_) * p.OW + rox) * opix + chunk_ofs(co, g)); \
            }                                                                                     \
        }
#define S2_RES_ADD(HALF)                                                                          \
        if (do_res) {                                                                             \
            _Pragma("unroll") for (int t = 0; t < NH; ++t)                                        \
                if ((HALF) * NH + t < NT) {                                                       \
                    uint2 rh_, rl_;                                                               \
                    chunk_to_quad(rc[t], rh_, rl_);                                               \
                    float r_[4];                                                                  \
                    join4(rh_, rl_, r_);                                                          \
                    _Pragma("unroll") for (int i = 0; i < 4; ++i) acc[(HALF) * NH + t][i] += r_[i]; \
                }                                                                                 \
        }
        S2_RES_LOAD(0)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
#pragma unroll
            for (int i = 0; i < ROWS; ++i) {
                const int off = (i * S2C::IW + kx) * 16;
                const bf16x8 xh = *reinterpret_cast<const bf16x8*>(xrd + off);
                const bf16x8 xo = *reinterpret_cast<const bf16x8*>(xrd + off + S2C::LO_OFF);
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int d = i - ky;
                    if (d >= 0 && d % S == 0 && d / S < NT) {
                        const int t = d / S;
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[ky * 3 + kx], xh, acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[ky * 3 + kx], xo, acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[ky * 3 + kx], xh, acc[t], 0, 0, 0);
                    }
                }
            }
            if (reload) S2_LOAD_W(nct, nch, kx)     // this third is free: refill it for step s+1
            if (kx == 1) {
                S2_RES_ADD(0)
                if (NT >= 2) S2_RES_LOAD(1)
            }
            if (kx == 2 && NT >= 2) S2_RES_ADD(1)
        }
#undef S2_RES_LOAD
#undef S2_RES_ADD
        if (last_chunk) {
            // ---- epilogue: ReLU, split, 16-byte chunk stores ---------------------------------------
            const int ox = ox0 + (lane & 15);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int oy = oy0 + rg * NT + t;
                float v[4] = {acc[t][0], acc[t][1], acc[t][2], acc[t][3]};
                {
                    const int rfl = relu_floor(p.relu);          // branch-free (see sb.h)
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = relu_opt(v[i], rfl);
                }
                uint2 hi, lo;
                split4(v, hi, lo);
                const uint4 ch = quad_to_chunk(hi, lo);
                if (oy < p.OH && ox < p.OW)
                    *reinterpret_cast<uint4*>(p.y + ((size_t)(n * p.OH + oy) * p.OW + ox) * opix + chunk_ofs(co, g)) = ch;
            }
            if (!more) break;
            item += G;
            n = s_n; oy0 = s_oy0; ox0 = s_ox0; ct = s_ct;
            c = 0;
        } else {
            ++c;
        }
    }
#undef S2_DECODE
#undef S2_PREFETCH
#undef S2_LOAD_W
}

template <int S, int TH, int MW>
int launch_s2c32_t(const ConvParams& p, hipStream_t stream) {
    using S2C = ConvCfg<3, S, TH, 2>;
    constexpr int S2_TH = TH;
    auto kern = conv_s2c32_kernel<S, TH, MW>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, S2C::XBYTES);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const int tiles_x = (p.OW + TW - 1) / TW, tiles_y = (p.OH + S2_TH - 1) / S2_TH;
    const int ctiles = p.Coutp / (16 * MW);
    const long long nitems = (long long)p.N * tiles_y * tiles_x * ctiles;
    if (nitems <= 0 || nitems > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    static int slots = 0;
    if (!slots) {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) {
            hipDeviceProp_t prop;
            if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
        }
        slots = 2 * cus;
    }
    int grid = (int)(nitems < slots ? nitems : slots);
    if (grid > ctiles) grid -= grid % ctiles;   // grid stride keeps the cout slice of a workgroup constant
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NTHREADS), S2C::XBYTES, stream, p, tiles_x, tiles_y, ctiles,
                       (int)nitems);
    return (int)hipGetLastError();
}

}  // namespace

bool conv_s2c32_supported(const ConvParams& p) {
    return (p.Cinp & 31) == 0 && p.Cinp >= 32 && (p.Coutp & 31) == 0 && !p.out_f32 &&
           (long long)p.H * p.W * p.Cinp * 4 <= 0x7fffffffLL;
}

int launch_conv_s2c32(const ConvParams& p, hipStream_t stream) {
    if (!conv_s2c32_supported(p)) return (int)hipErrorInvalidValue;
    return (p.Coutp % 64 == 0) ? launch_s2c32_t<2, 4, 4>(p, stream) : launch_s2c32_t<2, 4, 2>(p, stream);
}

// the same scheme for stride 1 (TH = 16, a wave = one cout tile x 8 rows, 32 couts per workgroup)
int launch_conv_s1w(const ConvParams& p, hipStream_t stream) {
    if (!conv_s2c32_supported(p)) return (int)hipErrorInvalidValue;
    return launch_s2c32_t<1, 16, 2>(p, stream);
}

}  // namespace esa
