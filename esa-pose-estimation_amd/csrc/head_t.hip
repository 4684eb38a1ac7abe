// head_t.hip — low-resolution head terms t_b = W_b·x_b (b = 2, 3) in the transposed "T layout" that
// head_fused2.hip interpolates on the matrix cores.
//
// Replaces (reference): the input-channel slices [:, C0+C1 : ...] of last_layer[0] (1x1 480->480,
// models/seg_hrnet.py:313-321) applied to branches 2 and 3 — by linearity the 1x1 conv is evaluated
// on each branch's own grid and up-sampled afterwards (plan.hip).
//
// T layout of one branch: [N][h][Ctp/32][part hi|lo][32 ch][XP] bf16, stored column = x + HT_PAD
// (columns outside [HT_PAD, HT_PAD + w) are written as zeros): the contraction index of the
// interpolation GEMM (source pixels of a row) is contiguous per channel, so an 8-pixel run is one
// MFMA A-operand fragment of one lane.
//
// One wave = 16 stored columns of one row x all Ctp output channels: the x fragments of the 16 pixels
// stay in registers (A operand: rows = pixels), the weight fragments stream through LDS in 32-channel
// chunks (B operand: columns = output channels), so D[pixel][cout] puts 4 consecutive pixels of one
// channel in one lane -> 8-byte stores of hi / lo along the row.
#include "devstate.h"
#include "kernels.h"
#include "sb.h"

namespace esa {
namespace {

template <int NCH>
__global__ __launch_bounds__(256, 2) void head_t_kernel(HeadTParams p, int tiles_per_row, long long ntiles, int csplit) {
    constexpr int WFR = 4 * NCH;                   // 1-KB weight fragments per 32-channel chunk
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    const long long tile = (long long)blockIdx.x * 4 + wave;
    const bool live = tile < ntiles;
    const int k = (int)(tile % tiles_per_row);
    const long long row = tile / tiles_per_row;                 // n*h + sy
    const int col = k * 16 - HT_PAD + i;
    const bool valid = live && col >= 0 && col < p.w;
    const int nchunks = p.Ctp >> 5;
    // blockIdx.y splits the output-channel chunks (small grids: more workgroups, each re-loading its x fragments)
    const int c_begin = (int)blockIdx.y * csplit, c_end = min(c_begin + csplit, nchunks);

    bf16x8 xh[NCH], xl[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        uint4 h = make_uint4(0, 0, 0, 0), l = make_uint4(0, 0, 0, 0);
        if (valid) {
            const char* a = p.x + ((size_t)row * p.w + col) * (size_t)(p.Cinp * 4) + c * 128 + g * 32;
            h = *reinterpret_cast<const uint4*>(a);
            l = *reinterpret_cast<const uint4*>(a + 16);
        }
        xh[c] = __builtin_bit_cast(bf16x8, h);
        xl[c] = __builtin_bit_cast(bf16x8, l);
    }

    typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
    const u32x4* wsrc = reinterpret_cast<const u32x4*>(p.wt);
    u32x4 wreg[NCH];
#define HT_PREFETCH(CH)                                                                       \
    {                                                                                         \
        _Pragma("unroll") for (int it = 0; it < NCH; ++it)                                    \
            wreg[it] = wsrc[(size_t)(CH) * WFR * 64 + it * 256 + tid];                        \
    }
#define HT_COMMIT(BUF)                                                                        \
    {                                                                                         \
        _Pragma("unroll") for (int it = 0; it < NCH; ++it)                                    \
            *reinterpret_cast<u32x4*>(smem + (BUF) * (WFR * 1024) + (it * 256 + tid) * 16) = wreg[it]; \
    }
    HT_PREFETCH(c_begin)
    HT_COMMIT(0)
    __syncthreads();
    char* trow = p.t + (size_t)row * nchunks * 64 * (size_t)(p.XP * 2) + (size_t)(k * 16 + (g >> 1) * 8) * 2;
    for (int cc = c_begin; cc < c_end; ++cc) {
        const int buf = (cc - c_begin) & 1;
        if (cc + 1 < c_end) HT_PREFETCH(cc + 1)
        const char* wb = smem + buf * (WFR * 1024) + lane * 16;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const bf16x8 wh = *reinterpret_cast<const bf16x8*>(wb + ((m * NCH + c) * 2 + 0) * 1024);
                const bf16x8 wl = *reinterpret_cast<const bf16x8*>(wb + ((m * NCH + c) * 2 + 1) * 1024);
                d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xl[c], wh, d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh[c], wl, d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh[c], wh, d, 0, 0, 0);
            }
            {   // rows g, g^1 of the D tile hold 8 consecutive pixels of a channel: swap halves so that the
                // odd row stores the hi part and the even row the lo part, 16 bytes each (sb.h)
                const float v[4] = {d[0], d[1], d[2], d[3]};
                uint2 hi, lo;
                split4(v, hi, lo);
                const uint4 ch = quad_to_chunk(hi, lo);
                char* o = trow + ((size_t)(cc * 2 + ((g & 1) ? 0 : 1)) * 32 + m * 16 + i) * (size_t)(p.XP * 2);
                if (live) *reinterpret_cast<uint4*>(o) = ch;
            }
        }
        if (cc + 1 < c_end) {
            HT_COMMIT(buf ^ 1)
            __syncthreads();
        }
    }
#undef HT_PREFETCH
#undef HT_COMMIT
}

template <int NCH>
int launch_head_t_n(const HeadTParams& p, hipStream_t stream) {
    auto kern = head_t_kernel<NCH>;
    const int lds = 2 * 4 * NCH * 1024;
    if (const int e_ = ensure_dyn_lds(reinterpret_cast<const void*>(kern), lds)) return e_;
    const int tiles_per_row = p.XP / 16;
    const long long ntiles = (long long)p.N * p.h * tiles_per_row;
    const long long nblk = (ntiles + 3) / 4;
    if (nblk <= 0 || nblk > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    // aim for >= 4 workgroups per CU: split the chunk loop over blockIdx.y when the pixel grid is small
    const int nchunks = p.Ctp >> 5;
    int parts = 1;
    while (nblk * parts < 1024 && parts * 2 <= nchunks) parts *= 2;
    const int csplit = (nchunks + parts - 1) / parts;
    const int ny = (nchunks + csplit - 1) / csplit;
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk, (unsigned)ny), dim3(256), lds, stream, p, tiles_per_row, ntiles, csplit);
    return (int)hipGetLastError();
}

}  // namespace

int head_t_xp(int w) { return ((w + 8 + 15) / 16) * 16; }

bool head_t_supported(int Cinp) {
    const int n = Cinp / 32;
    return (Cinp % 32) == 0 && (n == 2 || n == 3 || n == 4 || n == 6 || n == 8 || n == 12);
}

int launch_head_t(const HeadTParams& p, hipStream_t stream) {
    if ((p.Ctp & 31) || (p.XP & 15) || p.XP < p.w + HT_PAD) return (int)hipErrorInvalidValue;
    switch (p.Cinp / 32) {
        case 2: return launch_head_t_n<2>(p, stream);
        case 3: return launch_head_t_n<3>(p, stream);
        case 4: return launch_head_t_n<4>(p, stream);
        case 6: return launch_head_t_n<6>(p, stream);
        case 8: return launch_head_t_n<8>(p, stream);
        case 12: return launch_head_t_n<12>(p, stream);
    }
    return (int)hipErrorInvalidValue;
}

}  // namespace esa
