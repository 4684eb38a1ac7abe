// conv1x1.hip — 1x1 convolution (+ folded-BN bias, optional ReLU) for the small fuse-up layers:
// fuse_layers[i][j], j > i (models/seg_hrnet.py:176-220: 1x1 conv + BN on the low-resolution branch,
// up-sampled afterwards by fuse.hip).
//
// These layers are tiny (256 -> 32 channels on a 16x16 grid is 67 MFLOP per batch) and the tiled kernel
// of conv_mfma.hip walks their input channels in 32-wide chunks, one global -> LDS round trip per chunk:
// 8 serial round trips for 0.2 us of math.  Here a wave owns 16 pixels of a row and holds ALL their
// input channels in registers (one round trip, B operand: columns = pixels); the weights stream through
// LDS one 32-cout chunk at a time (A operand: rows = output channels), the next chunk's fragments in
// flight while the current one is consumed.  D[cout][pixel] leaves 4 consecutive channels of a pixel
// per lane: 16-byte SB chunk stores (sb.h).
#include <algorithm>

#include "devstate.h"
#include "kernels.h"
#include "sb.h"

namespace esa {
namespace {

// BF (ConvParams::bf): NCH counts 64-channel blocks of 128 bytes; the (hi, lo) fragment pair of a chunk becomes the
// two K-steps of a block (sb.h), 2 MFMAs instead of 3, and the epilogue packs 4 channels into 8 bytes.
// PG: 16-pixel groups per wave.  With 2 the wave reads every weight fragment once for 32 pixels — twice the MFMAs per barrier
// and per LDS byte; used for the big bf16 launches, whose chunks are only 4 .. 8 MFMAs deep per group.
template <int NCH, bool BF, int PG = 1>
__device__ __forceinline__ void conv1x1_body(const ConvParams& p, const long long ntiles, const int tiles_per_row,
                                             const int bid, const int G, const int split = 0, const int nsplit = 1) {
    constexpr int WFR = 4 * NCH;                   // 1-KB weight fragments per 32-cout chunk
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    bool valid[PG];
    size_t pixel[PG];                               // (n*H + y) * W + x of the lane's pixel in group pg
#pragma unroll
    for (int pg = 0; pg < PG; ++pg) {
        const long long tile = ((long long)xcd_contiguous(bid, G) * 4 + wave) * PG + pg;
        const int k = (int)(tile % tiles_per_row);
        const long long row = tile / tiles_per_row;                 // n*H + y
        const int col = k * 16 + i;
        valid[pg] = tile < ntiles && col < p.W;
        pixel[pg] = (size_t)row * p.W + col;
    }
    // very wide outputs (the nine-tap products of head_gather.hip: 135 chunks) are cut into `nsplit` ranges of 32-cout
    // chunks, one workgroup each: more workgroups than CUs on a 16x16 grid, and a shorter serial chunk loop
    const int per = ((p.Coutp >> 5) + nsplit - 1) / nsplit;
    const int c_begin = split * per, nchunks = min(c_begin + per, p.Coutp >> 5);

    bf16x8 xh[PG][NCH], xl[PG][NCH];
#pragma unroll
    for (int pg = 0; pg < PG; ++pg)
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            uint4 h = make_uint4(0, 0, 0, 0), l = make_uint4(0, 0, 0, 0);
            if (valid[pg]) {
                const char* a = p.x + pixel[pg] * (size_t)(p.Cinp * (BF ? 2 : 4)) + c * 128 + g * (BF ? 16 : 32);
                h = *reinterpret_cast<const uint4*>(a);
                l = *reinterpret_cast<const uint4*>(a + (BF ? 64 : 16));
            }
            xh[pg][c] = __builtin_bit_cast(bf16x8, h);
            xl[pg][c] = __builtin_bit_cast(bf16x8, l);
        }

    typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
    const u32x4* wsrc = reinterpret_cast<const u32x4*>(p.w);
    u32x4 wreg[NCH];
    // (the loads of the loop are unconditional and pinned at the top of an iteration — hipcc otherwise sinks them to the LDS
    // writes that consume them; the bias sits in LDS: as a global load it was waited for in front of every cout tile's first MFMA)
#define C1_PREFETCH(CH)                                                                       \
    {                                                                                         \
        _Pragma("unroll") for (int it = 0; it < NCH; ++it)                                    \
            wreg[it] = wsrc[(size_t)(CH) * WFR * 64 + it * 256 + tid];                        \
        __builtin_amdgcn_sched_barrier(0);                                                    \
    }
#define C1_COMMIT(BUF)                                                                        \
    {                                                                                         \
        _Pragma("unroll") for (int it = 0; it < NCH; ++it)                                    \
            *reinterpret_cast<u32x4*>(smem + (BUF) * (WFR * 1024) + (it * 256 + tid) * 16) = wreg[it]; \
    }
    if (c_begin >= nchunks) return;
    float* const bias_s = reinterpret_cast<float*>(smem + 2 * WFR * 1024);
    for (int k = tid; k < (nchunks - c_begin) * 32; k += 256) bias_s[k] = p.bias[c_begin * 32 + k];
    C1_PREFETCH(c_begin)
    C1_COMMIT(0)
    __syncthreads();
    char* orow[PG];
#pragma unroll
    for (int pg = 0; pg < PG; ++pg) orow[pg] = p.y + pixel[pg] * (size_t)(p.Coutp * (BF ? 2 : 4));
    for (int cc = c_begin; cc < nchunks; ++cc) {
        const int buf = (cc - c_begin) & 1;
        C1_PREFETCH(cc + 1 < nchunks ? cc + 1 : cc)
        const char* wb = smem + buf * (WFR * 1024) + lane * 16;
        uint2 pk[PG][2];                             // BF: the two cout tiles' packed quads, stored together below
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int co = cc * 32 + m * 16 + g * 4;
            const f32x4 bv = *reinterpret_cast<const f32x4*>(bias_s + (cc - c_begin) * 32 + m * 16 + g * 4);
            f32x4 d[PG];
#pragma unroll
            for (int pg = 0; pg < PG; ++pg) d[pg] = bv;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const bf16x8 wh = *reinterpret_cast<const bf16x8*>(wb + ((m * NCH + c) * 2 + 0) * 1024);
                const bf16x8 wl = *reinterpret_cast<const bf16x8*>(wb + ((m * NCH + c) * 2 + 1) * 1024);
#pragma unroll
                for (int pg = 0; pg < PG; ++pg) {
                    if (BF) {
                        d[pg] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xh[pg][c], d[pg], 0, 0, 0);
                        d[pg] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, xl[pg][c], d[pg], 0, 0, 0);
                    } else {
                        d[pg] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, xh[pg][c], d[pg], 0, 0, 0);
                        d[pg] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xl[pg][c], d[pg], 0, 0, 0);
                        d[pg] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xh[pg][c], d[pg], 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int pg = 0; pg < PG; ++pg) {
                float v[4] = {d[pg][0], d[pg][1], d[pg][2], d[pg][3]};
                {
                    const int rfl = relu_floor(p.relu);          // branch-free (see sb.h)
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = relu_opt(v[r], rfl);
                }
                if (BF) {
                    pk[pg][m] = pack4_bf16(v);
                } else if (p.out_f32) {     // plain f32 NHWC (same pixel pitch): the nine-tap products read by head_gather.hip
                    if (valid[pg]) *reinterpret_cast<float4*>(orow[pg] + co * 4) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
                    uint2 hi, lo;
                    split4(v, hi, lo);
                    const uint4 ch = quad_to_chunk(hi, lo);
                    if (valid[pg]) *reinterpret_cast<uint4*>(orow[pg] + chunk_ofs(co, g)) = ch;
                }
            }
        }
        if (BF) {
            // A lane holds 8 bytes of each cout tile (couts 4g .. 4g+3).  v_permlane16_swap trades the odd lane rows of the
            // first tile for the even rows of the second: an even row g then owns couts 4g .. 4g+7 of tile 0, an odd row couts
            // 4(g-1) .. 4(g-1)+7 of tile 1 — ONE 16-byte store per lane, 64 contiguous bytes per pixel and chunk (the 8-byte
            // stores left 32-byte pieces: half an HBM burst each).  All lanes execute the swap; only the store is predicated.
            const int cob = cc * 32 + ((g & 1) ? 16 + (g - 1) * 4 : g * 4);
#pragma unroll
            for (int pg = 0; pg < PG; ++pg) {
                const auto sx = __builtin_amdgcn_permlane16_swap(pk[pg][0].x, pk[pg][1].x, false, false);
                const auto sy = __builtin_amdgcn_permlane16_swap(pk[pg][0].y, pk[pg][1].y, false, false);
                if (valid[pg]) *reinterpret_cast<uint4*>(orow[pg] + cob * 2) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        C1_COMMIT(buf ^ 1)
        __syncthreads();
    }
#undef C1_PREFETCH
#undef C1_COMMIT
}

template <int NCH, bool BF = false, int PG = 1>
__global__ __launch_bounds__(256, 2) void conv1x1_kernel(ConvParams p, long long ntiles, int tiles_per_row) {
    conv1x1_body<NCH, BF, PG>(p, ntiles, tiles_per_row, (int)blockIdx.x, (int)gridDim.x, (int)blockIdx.y, (int)gridDim.y);
}

// Several independent 1x1 convolutions in one launch (the fuse-up convolutions of an HRModule, models/seg_hrnet.py:
// 176-197: each reads one branch, none reads another's output).  Workgroups [start[j], start[j+1]) run convolution j as
// its own launch would; the body is instantiated for every input depth the plans have (2 / 4 / 8 chunks of 32 channels).
constexpr int C1_MAXJOBS = 6;
struct C1Jobs {
    ConvParams p[C1_MAXJOBS];
    long long ntiles[C1_MAXJOBS];
    int tiles_per_row[C1_MAXJOBS];
    int start[C1_MAXJOBS + 1];
    int njobs;
};
__global__ __launch_bounds__(256, 2) void conv1x1_jobs_kernel(C1Jobs jobs) {
    const int b = (int)blockIdx.x;
    int j = 0;
#pragma unroll
    for (int k = 1; k < C1_MAXJOBS; ++k) j += (k < jobs.njobs && b >= jobs.start[k]) ? 1 : 0;
    const int bid = b - jobs.start[j], G = jobs.start[j + 1] - jobs.start[j];
    const ConvParams& p = jobs.p[j];
    switch (p.Cinp >> 5) {
        case 2: conv1x1_body<2, false>(p, jobs.ntiles[j], jobs.tiles_per_row[j], bid, G); break;
        case 4: conv1x1_body<4, false>(p, jobs.ntiles[j], jobs.tiles_per_row[j], bid, G); break;
        default: conv1x1_body<8, false>(p, jobs.ntiles[j], jobs.tiles_per_row[j], bid, G); break;
    }
}

template <int NCH, bool BF = false>
int launch_conv1x1_n(const ConvParams& p, hipStream_t stream) {
    const int lds = 2 * 4 * NCH * 1024 + p.Coutp * 4;        // two weight buffers + the bias
    const int tiles_per_row = (p.W + 15) / 16;
    const long long ntiles = (long long)p.N * p.H * tiles_per_row;
    // two 16-pixel groups per wave for the big bf16 launches (same MFMAs in the same order on every accumulator: the result
    // does not depend on the choice)
    if constexpr (BF && NCH <= 4) {
        if (ntiles >= 32LL * device_cus()) {
            auto kern2 = conv1x1_kernel<NCH, BF, 2>;
            if (const int e_ = ensure_dyn_lds(reinterpret_cast<const void*>(kern2), lds)) return e_;
            const long long nblk2 = (ntiles + 7) / 8;
            if (nblk2 > 0x7fffffffLL) return (int)hipErrorInvalidValue;
            hipLaunchKernelGGL(kern2, dim3((unsigned)nblk2, 1u), dim3(256), lds, stream, p, ntiles, tiles_per_row);
            return (int)hipGetLastError();
        }
    }
    auto kern = conv1x1_kernel<NCH, BF>;
    if (const int e_ = ensure_dyn_lds(reinterpret_cast<const void*>(kern), lds)) return e_;
    const long long nblk = (ntiles + 3) / 4;
    if (nblk <= 0 || nblk > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    int nsplit = 1;
    if ((p.Coutp >> 5) >= 32) nsplit = (int)std::min<long long>(8, std::max<long long>(1, (4LL * device_cus() + nblk - 1) / nblk));
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk, (unsigned)nsplit), dim3(256), lds, stream, p, ntiles, tiles_per_row);
    return (int)hipGetLastError();
}

}  // namespace

// weights must be packed with pack_conv_weights(k = 1): [Coutp/16][Cinp/32][hi|lo][64] fragments — the
// 4*NCH fragments of a 32-cout chunk are contiguous, which is what the staging above relies on
bool conv1x1_supported(const ConvParams& p) {
    if ((p.fmt == FMT_BF)) {         // blocks of 64 input channels; every width of the BF plans (64 .. 768 channels)
        const int n = p.Cinp / 64;
        return (p.Cinp % 64) == 0 && (p.Coutp % 64) == 0 && !p.res && !p.out_f32 && p.H == p.OH && p.W == p.OW &&
               (n == 1 || n == 2 || n == 3 || n == 4 || n == 6 || n == 8 || n == 12);
    }
    const int n = p.Cinp / 32;
    return (p.Cinp % 32) == 0 && (p.Coutp % 32) == 0 && !p.res && p.H == p.OH && p.W == p.OW &&
           (n == 2 || n == 3 || n == 4 || n == 6 || n == 8 || n == 12);
}

int launch_conv1x1(const ConvParams& p, hipStream_t stream) {
    if (!conv1x1_supported(p)) return (int)hipErrorInvalidValue;
    if ((p.fmt == FMT_BF)) {
        switch (p.Cinp / 64) {
            case 1: return launch_conv1x1_n<1, true>(p, stream);
            case 2: return launch_conv1x1_n<2, true>(p, stream);
            case 3: return launch_conv1x1_n<3, true>(p, stream);
            case 4: return launch_conv1x1_n<4, true>(p, stream);
            case 6: return launch_conv1x1_n<6, true>(p, stream);
            case 8: return launch_conv1x1_n<8, true>(p, stream);
            case 12: return launch_conv1x1_n<12, true>(p, stream);
        }
        return (int)hipErrorInvalidValue;
    }
    switch (p.Cinp / 32) {
        case 2: return launch_conv1x1_n<2>(p, stream);
        case 3: return launch_conv1x1_n<3>(p, stream);
        case 4: return launch_conv1x1_n<4>(p, stream);
        case 6: return launch_conv1x1_n<6>(p, stream);
        case 8: return launch_conv1x1_n<8>(p, stream);
        case 12: return launch_conv1x1_n<12>(p, stream);
    }
    return (int)hipErrorInvalidValue;
}

bool conv1x1_jobs_supported(const ConvParams* ps, int n) {
    if (n < 2 || n > C1_MAXJOBS) return false;
    for (int j = 0; j < n; ++j) {
        const int nch = ps[j].Cinp / 32;
        if ((ps[j].fmt != FMT_SB) || ps[j].out_f32 || !conv1x1_supported(ps[j]) || (nch != 2 && nch != 4 && nch != 8)) return false;
    }
    return true;
}

int launch_conv1x1_jobs(const ConvParams* ps, int n, hipStream_t stream) {
    if (!conv1x1_jobs_supported(ps, n)) return (int)hipErrorInvalidValue;
    C1Jobs jobs{};
    jobs.njobs = n;
    int at = 0;
    for (int j = 0; j < n; ++j) {
        const ConvParams& p = ps[j];
        jobs.p[j] = p;
        jobs.tiles_per_row[j] = (p.W + 15) / 16;
        jobs.ntiles[j] = (long long)p.N * p.H * jobs.tiles_per_row[j];
        const long long nblk = (jobs.ntiles[j] + 3) / 4;
        if (nblk <= 0 || at + nblk > 0x7fffffffLL) return (int)hipErrorInvalidValue;
        jobs.start[j] = at;
        at += (int)nblk;
    }
    for (int k = n; k <= C1_MAXJOBS; ++k) jobs.start[k] = at;
    int coutp_max = 0;
    for (int j = 0; j < n; ++j) coutp_max = ps[j].Coutp > coutp_max ? ps[j].Coutp : coutp_max;
    const int lds = 2 * 4 * 8 * 1024 + coutp_max * 4;       // the deepest body's two weight buffers + a member's bias
    if (const int e_ = ensure_dyn_lds(reinterpret_cast<const void*>(conv1x1_jobs_kernel), lds)) return e_;
    hipLaunchKernelGGL(conv1x1_jobs_kernel, dim3((unsigned)at), dim3(256), lds, stream, jobs);
    return (int)hipGetLastError();
}

}  // namespace esa
