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
// groups when the two planes are congruent mod 256 B (stride-2 convs: one slot apart so that
// even/odd pixels of the two k-groups interleave).  Staging reads whole 128-B lines (8
// consecutive lanes = the 8 pieces of one pixel) and scatters them to the 8 planes; the plane
// skews keep those ds_write_b128 at most 2-way conflicted (see ConvCfg::plane_off).
//
// Pipeline.  Persistent workgroups (2 per CU) walk a strided list of (image, tile, cout-tile) work
// items.  The input tile of the NEXT (item, chunk) is loaded into registers before the MFMAs of
// the current one and written to LDS after them (issue-early / write-late), also across item
// boundaries, so the epilogue stores of item i overlap the staging of item i+1; the weight
// fragments of a chunk go global -> LDS by DMA (global_load_lds, no registers) and stay resident
// across items for single-chunk layers (Cin = 32).
#include "conv_cfg.h"
#include "devstate.h"
#include "kernels.h"
#include "sb.h"

// ablation switches for tuning experiments (all 1 in the shipped build)
#ifndef ESA_DO_XLOAD
#define ESA_DO_XLOAD 1
#endif
#ifndef ESA_DO_MFMA
#define ESA_DO_MFMA 1
#endif
#ifndef ESA_DO_STORE
#define ESA_DO_STORE 1
#endif
#ifndef ESA_DO_WDMA
#define ESA_DO_WDMA 1
#endif
#ifndef ESA_DO_RES
#define ESA_DO_RES 1
#endif
#ifndef ESA_XCD_REMAP
#define ESA_XCD_REMAP 1
#endif
#ifndef ESA_CONV_RING
#define ESA_CONV_RING 1         // weight-thirds ring for 3x3 convs with >= 2 input chunks
#endif

namespace esa {

namespace {


template <int KS, int S, int TH, int MT, bool PERSIST>
__global__ __launch_bounds__(NTHREADS, 2) void conv_mfma_kernel(ConvParams p, int tiles_x,
                                                               int tiles_y, int ctiles, int nitems) {
    using C = ConvCfg<KS, S, TH, MT>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* xs = smem;
    char* wsm = smem + C::XBYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int nchunks = p.Cinp >> 5;
    const int pix_stride = p.Cinp * 4;

    // Persistent workgroups: work item = (image, tile, cout tile), cout tile fastest.  Blocks b and
    // b+8 share an XCD (round-robin dispatch), so with the remap below every XCD owns a contiguous
    // run of items and the cout-tile siblings that re-read one input tile hit the same L2.
    // (PERSIST = false: one item per workgroup, same code with the cross-item prefetch compiled out.)
    const int G = gridDim.x;
    int item = blockIdx.x;
    if (ESA_XCD_REMAP && (G & 7) == 0) item = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);

    // ---- staging map of the item being PREFETCHED (chunk-invariant) ---------------------------
    // X: unit u = it*256 + tid -> pixel q = u>>3, 16-B piece j = u&7 of that pixel's 128-B channel
    // chunk: 8 consecutive lanes read one whole 128-B line (fully coalesced).
    int xg[C::XITER];      // byte offset of (pixel, j) inside the image, -1 = zero padding
    const int jst = tid & 7;
    const int q0 = tid >> 3;                                  // pixel of iteration 0; +32 per iteration
    char* xwr = xs + C::plane_off(jst) + q0 * 16;             // LDS write address, +512 per iteration
    const char* xn;                                           // image base of the prefetched item
    int s_n, s_oy0, s_ox0, s_ct;
#define ESA_DECODE(ITEM)                                                                          \
    {                                                                                             \
        int b_ = (ITEM);                                                                          \
        s_ct = b_ % ctiles; b_ /= ctiles;                                                         \
        const int tx_ = b_ % tiles_x; b_ /= tiles_x;                                              \
        const int ty_ = b_ % tiles_y;                                                             \
        s_n = b_ / tiles_y;                                                                       \
        s_oy0 = ty_ * TH; s_ox0 = tx_ * TW;                                                       \
        xn = p.x + (size_t)s_n * p.H * p.W * pix_stride;                                          \
        _Pragma("unroll") for (int it = 0; it < C::XITER; ++it) {                                 \
            const int q = q0 + it * 32;                                                           \
            const int qy = q / C::IW, qx = q - qy * C::IW;                                        \
            const int gy = s_oy0 * S - C::PAD + qy, gx = s_ox0 * S - C::PAD + qx;                 \
            const bool inside = q < C::NPIX && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;        \
            xg[it] = inside ? ((gy * p.W + gx) * pix_stride + jst * 16) : -1;                     \
        }                                                                                         \
    }

    uint4 xr[C::XITER];
    // issue-early half of the X staging: global -> registers
#define ESA_PREFETCH_X(CH)                                                                        \
    {                                                                                             \
        _Pragma("unroll") for (int it = 0; it < C::XITER; ++it) {                                 \
            uint4 v = make_uint4(0, 0, 0, 0);                                                     \
            if (ESA_DO_XLOAD && xg[it] >= 0) v = *reinterpret_cast<const uint4*>(xn + xg[it] + (CH) * 128); \
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

    // per-lane LDS read bases
    const int g = lane >> 4;
    const char* xrd = xs + C::plane_off(2 * g) + ((wave * C::NT * S) * C::IW + (lane & 15) * S) * 16;
    const char* wrd = wsm + lane * 16;
    constexpr int ROWS = (C::NT - 1) * S + KS;               // input rows one wave touches
    const int opix = p.Coutp * 4;

    if (item >= nitems) return;
    ESA_DECODE(item)
    ESA_PREFETCH_X(0)
    int res_ct = -1;          // cout tile whose weights are resident in LDS (single-chunk layers)
    bool first = true;
    while (item < nitems) {
        const int n = s_n, oy0 = s_oy0, ox0 = s_ox0, ct = s_ct;   // the item being computed
        const uint4* wbase = p.w + (size_t)(ct * MT) * nchunks * (C::TAPS * 128);
        const int next = item + G;
        // accumulators start at bias (+ residual): the residual loads are issued here, a whole
        // staging phase ahead of the first MFMA that consumes them, and the epilogue has no loads.
        // D tile: column (lane&15) = pixel, rows (lane>>4)*4 + r = cout -> 4 consecutive channels.
        const int ox = ox0 + (lane & 15);
        f32x4 acc[MT][C::NT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int co = (ct * MT + m) * 16 + g * 4;                 // first of this lane's 4 couts
            const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + co);
#pragma unroll
            for (int t = 0; t < C::NT; ++t) {
                acc[m][t] = bv;
                const int oy = oy0 + wave * C::NT + t;
                if (p.res) {                   // 16-byte chunk per lane, halves swapped into quads (sb.h)
                    uint4 rc = make_uint4(0, 0, 0, 0);
                    if (oy < p.OH && ox < p.OW)
                        rc = *reinterpret_cast<const uint4*>(p.res + ((size_t)(n * p.OH + oy) * p.OW + ox) * opix + chunk_ofs(co, g));
                    uint2 rh, rl;
                    chunk_to_quad(rc, rh, rl);
                    float r[4];
                    join4(rh, rl, r);
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[m][t][i] += r[i];
                }
            }
        }

        for (int c = 0; c < nchunks; ++c) {
            if (!first) __syncthreads();     // everyone finished reading the previous chunk / item
            first = false;
            if (nchunks > 1 || ct != res_ct) {
                ESA_DMA_W(c)
                res_ct = ct;
                ESA_COMMIT_X()
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the W DMA of this wave has landed
            } else {                         // weights already resident: nothing to wait for but X
                ESA_COMMIT_X()
            }
            __syncthreads();
            if (c + 1 < nchunks) {
                ESA_PREFETCH_X(c + 1)
            } else if (PERSIST && next < nitems) {   // cross-item pipelining: chunk 0 of the next item
                ESA_DECODE(next)
                ESA_PREFETCH_X(0)
            }
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
                    const bf16x8 xo = *reinterpret_cast<const bf16x8*>(xrd + off + C::LO_OFF);
#pragma unroll
                    for (int ky = 0; ky < KS; ++ky) {
                        const int d = i - ky;
                        if (ESA_DO_MFMA && d >= 0 && d % S == 0 && d / S < C::NT) {
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

        // ---- epilogue: ReLU, split, store (no loads) -------------------------------------------
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int co = (ct * MT + m) * 16 + g * 4;
#pragma unroll
            for (int t = 0; t < C::NT; ++t) {
                const int oy = oy0 + wave * C::NT + t;
                const bool inr = (ESA_DO_STORE || acc[m][t][0] == 123.456f) && oy < p.OH && ox < p.OW;
                float v[4] = {acc[m][t][0], acc[m][t][1], acc[m][t][2], acc[m][t][3]};
                {
                    const int rfl = relu_floor(p.relu);          // branch-free (see sb.h)
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = relu_opt(v[i], rfl);
                }
                if (p.out_f32) {
                    const size_t of = ((size_t)(n * p.OH + oy) * p.OW + ox) * opix + (size_t)co * 4;
                    if (inr) *reinterpret_cast<f32x4*>(p.y + of) = f32x4{v[0], v[1], v[2], v[3]};
                } else {
                    uint2 hi, lo;
                    split4(v, hi, lo);
                    const uint4 ch = quad_to_chunk(hi, lo);        // all lanes; only the store is predicated
                    if (inr) *reinterpret_cast<uint4*>(p.y + ((size_t)(n * p.OH + oy) * p.OW + ox) * opix + chunk_ofs(co, g)) = ch;
                }
            }
        }
        if (!PERSIST) break;
        item = next;
    }
#undef ESA_DECODE
#undef ESA_PREFETCH_X
#undef ESA_COMMIT_X
#undef ESA_DMA_W
}

// ---------------------------------------------------------------------------------------------------
// 3x3 convolution with >= 2 input-channel chunks: same tiling and MFMA loop, but the weight
// fragments of chunk c+1 are DMA'd into LDS *while chunk c is being computed*.  The compute loop is
// kx-major, so the weight image is kept as three 12-KB "thirds" (one per kx): as soon as every wave
// has finished phase kx of chunk c (a barrier) slot kx is refilled with the kx-third of chunk c+1,
// which is not read before phase kx of the next chunk.  No extra LDS, two more barriers per chunk,
// and the L2 -> LDS weight latency (the exposed cost of the plain kernel on deep layers: 37 KB per
// workgroup and chunk, re-fetched by every pixel tile) leaves the critical path.
// LDS weight layout here: [kx][mt][ky][hi|lo][1 KB].
template <int S, int TH, int MT>
__global__ __launch_bounds__(NTHREADS, 2) void conv_mfma_ring_kernel(ConvParams p, int tiles_x,
                                                                    int tiles_y, int ctiles) {
    using C = ConvCfg<3, S, TH, MT>;
    constexpr int THIRD_FRAGS = MT * 3 * 2;                 // 1-KB fragments per kx-third
    constexpr int THIRD_BYTES = THIRD_FRAGS * 1024;
    constexpr int DPW = (THIRD_FRAGS + 3) / 4;              // DMA instructions per wave per third
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* xs = smem;
    char* wsm = smem + C::XBYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nchunks = p.Cinp >> 5;
    const int pix_stride = p.Cinp * 4;
    // blocks b and b+8 share an XCD (round-robin dispatch): give every XCD a contiguous run of items, so
    // the cout-tile siblings that re-read one input tile (and neighbouring tiles' halos) hit the same L2
    int b = blockIdx.x;
    if (ESA_XCD_REMAP && (gridDim.x & 7) == 0) b = (b & 7) * (gridDim.x >> 3) + (b >> 3);
    const int ct = b % ctiles; b /= ctiles;
    const int tx = b % tiles_x; b /= tiles_x;
    const int ty = b % tiles_y;
    const int n = b / tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW;

    int xg[C::XITER];
    const int jst = tid & 7, q0 = tid >> 3;
#pragma unroll
    for (int it = 0; it < C::XITER; ++it) {
        const int q = q0 + it * 32;
        const int qy = q / C::IW, qx = q - qy * C::IW;
        const int gy = oy0 * S - C::PAD + qy, gx = ox0 * S - C::PAD + qx;
        const bool inside = q < C::NPIX && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
        xg[it] = inside ? ((gy * p.W + gx) * pix_stride + jst * 16) : -1;
    }
    char* xwr = xs + C::plane_off(jst) + q0 * 16;
    const char* xn = p.x + (size_t)n * p.H * p.W * pix_stride;
    const uint4* wbase = p.w + (size_t)(ct * MT) * nchunks * (C::TAPS * 128);
    uint4 xr[C::XITER];
#define RING_PREFETCH_X(CH)                                                                       \
    {                                                                                             \
        _Pragma("unroll") for (int it = 0; it < C::XITER; ++it) {                                 \
            uint4 v = make_uint4(0, 0, 0, 0);                                                     \
            if (ESA_DO_XLOAD && xg[it] >= 0) v = *reinterpret_cast<const uint4*>(xn + xg[it] + (CH) * 128); \
            xr[it] = v;                                                                           \
        }                                                                                         \
    }
    // third KX of chunk CH: fragment f = (mt, ky, part) -> LDS slot KX; one fragment per wave-instruction
#define RING_DMA_W(CH, KX)                                                                        \
    {                                                                                             \
        _Pragma("unroll") for (int it = 0; it < DPW; ++it) {                                      \
            const int f = it * 4 + wave;                                                          \
            if (ESA_DO_WDMA && f < THIRD_FRAGS) {                                                 \
                const int mt = f / 6, r6 = f - mt * 6, ky = r6 >> 1, part = r6 & 1;               \
                const uint4* src = wbase + ((size_t)mt * nchunks + (CH)) * (C::TAPS * 128) +      \
                                   ((ky * 3 + (KX)) * 2 + part) * 64 + lane;                      \
                dma16(src, wsm + (KX) * THIRD_BYTES + __builtin_amdgcn_readfirstlane(f) * 1024);  \
            }                                                                                     \
        }                                                                                         \
    }

    const int g = lane >> 4;
    const int ox = ox0 + (lane & 15);
    const int opix = p.Coutp * 4;
    f32x4 acc[MT][C::NT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int co = (ct * MT + m) * 16 + g * 4;
        const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + co);
#pragma unroll
        for (int t = 0; t < C::NT; ++t) {
            acc[m][t] = bv;
            const int oy = oy0 + wave * C::NT + t;
            if (ESA_DO_RES && p.res) {         // 16-byte chunk per lane, halves swapped into quads (sb.h)
                uint4 rc = make_uint4(0, 0, 0, 0);
                if (oy < p.OH && ox < p.OW)
                    rc = *reinterpret_cast<const uint4*>(p.res + ((size_t)(n * p.OH + oy) * p.OW + ox) * opix + chunk_ofs(co, g));
                uint2 rh, rl;
                chunk_to_quad(rc, rh, rl);
                float r[4];
                join4(rh, rl, r);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[m][t][i] += r[i];
            }
        }
    }
    const char* xrd = xs + C::plane_off(2 * g) + ((wave * C::NT * S) * C::IW + (lane & 15) * S) * 16;
    const char* wrd = wsm + lane * 16;
    constexpr int ROWS = (C::NT - 1) * S + 3;

    RING_DMA_W(0, 0)
    RING_DMA_W(0, 1)
    RING_DMA_W(0, 2)
    RING_PREFETCH_X(0)
    for (int c = 0; c < nchunks; ++c) {
        if (c) __syncthreads();              // phase 2 of chunk c-1 done: X planes and slot 2 are free
        // commit X(c); the wait hipcc puts in front of it (vmcnt(0): an LDS-DMA is pending) also
        // retires the slot-0/slot-1 DMAs of this chunk, issued one and two MFMA phases ago
#pragma unroll
        for (int it = 0; it < C::XITER; ++it)
            if (q0 + it * 32 < C::NPIX) *reinterpret_cast<uint4*>(xwr + it * 512) = xr[it];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                     // X(c), W(c,0), W(c,1) visible (c == 0: W(0,2) too)
        if (c) RING_DMA_W(c, 2)              // lands during phases 0 and 1
        if (c + 1 < nchunks) RING_PREFETCH_X(c + 1)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            bf16x8 wh[3][MT], wl[3][MT];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    wh[ky][m] = *reinterpret_cast<const bf16x8*>(wrd + kx * THIRD_BYTES + ((m * 3 + ky) * 2 + 0) * 1024);
                    wl[ky][m] = *reinterpret_cast<const bf16x8*>(wrd + kx * THIRD_BYTES + ((m * 3 + ky) * 2 + 1) * 1024);
                }
#pragma unroll
            for (int i = 0; i < ROWS; ++i) {
                const int off = (i * C::IW + kx) * 16;
                const bf16x8 xh = *reinterpret_cast<const bf16x8*>(xrd + off);
                const bf16x8 xo = *reinterpret_cast<const bf16x8*>(xrd + off + C::LO_OFF);
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int d = i - ky;
                    if (ESA_DO_MFMA && d >= 0 && d % S == 0 && d / S < C::NT) {
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
            // slot kx is free once every wave is past phase kx: refill it for the next chunk.  The
            // barrier after phase 1 also publishes slot 2 of THIS chunk (issued after the commit); its
            // wait sits before the refill of slot 1 so that it never drains a just-issued DMA.
            if (kx == 0 && c + 1 < nchunks) {
                __syncthreads();
                RING_DMA_W(c + 1, 0)
            }
            if (kx == 1) {
                if (c) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (c || c + 1 < nchunks) __syncthreads();
                if (c + 1 < nchunks) RING_DMA_W(c + 1, 1)
            }
        }
    }
#undef RING_PREFETCH_X
#undef RING_DMA_W
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int co = (ct * MT + m) * 16 + g * 4;
#pragma unroll
        for (int t = 0; t < C::NT; ++t) {
            const int oy = oy0 + wave * C::NT + t;
            const bool inr = (ESA_DO_STORE || acc[m][t][0] == 123.456f) && oy < p.OH && ox < p.OW;
            float v[4] = {acc[m][t][0], acc[m][t][1], acc[m][t][2], acc[m][t][3]};
            {
                const int rfl = relu_floor(p.relu);          // branch-free (see sb.h)
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = relu_opt(v[i], rfl);
            }
            if (p.out_f32) {
                const size_t of = ((size_t)(n * p.OH + oy) * p.OW + ox) * opix + (size_t)co * 4;
                if (inr) *reinterpret_cast<f32x4*>(p.y + of) = f32x4{v[0], v[1], v[2], v[3]};
            } else {
                uint2 hi, lo;
                split4(v, hi, lo);
                const uint4 ch = quad_to_chunk(hi, lo);            // all lanes; only the store is predicated
                if (inr) *reinterpret_cast<uint4*>(p.y + ((size_t)(n * p.OH + oy) * p.OW + ox) * opix + chunk_ofs(co, g)) = ch;
            }
        }
    }
}

template <int S, int TH, int MT>
int launch_ring(const ConvParams& p, hipStream_t stream) {
    using C = ConvCfg<3, S, TH, MT>;
    auto kern = conv_mfma_ring_kernel<S, TH, MT>;
    if (const int e_ = ensure_dyn_lds(reinterpret_cast<const void*>(kern), C::LDS_BYTES)) return e_;
    const int tiles_x = (p.OW + TW - 1) / TW, tiles_y = (p.OH + TH - 1) / TH;
    const int ctiles = p.Coutp / (16 * MT);
    const long long nitems = (long long)p.N * tiles_y * tiles_x * ctiles;
    if (nitems <= 0 || nitems > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(kern, dim3((unsigned)nitems), dim3(NTHREADS), C::LDS_BYTES, stream, p, tiles_x, tiles_y, ctiles);
    return (int)hipGetLastError();
}

int persistent_grid(long long nitems) {
    const int slots = 2 * device_cus();       // two workgroups per CU (80 KB LDS each)
    return (int)(nitems < slots ? nitems : slots);
}

// PERSIST: persistent workgroups with cross-item prefetch and resident weights.  Measured on
// MI355X (round 1): it pays only for single-chunk layers (Cin = 32: the weights stay in LDS for the
// whole launch); for deeper layers its ~70 extra VGPRs cost more than the pipelining returns.
template <int KS, int S, int TH, int MT, bool PERSIST>
int launch_tp(const ConvParams& p, hipStream_t stream) {
    using C = ConvCfg<KS, S, TH, MT>;
    auto kern = conv_mfma_kernel<KS, S, TH, MT, PERSIST>;
    if (const int e_ = ensure_dyn_lds(reinterpret_cast<const void*>(kern), C::LDS_BYTES)) return e_;
    const int tiles_x = (p.OW + TW - 1) / TW, tiles_y = (p.OH + TH - 1) / TH;
    const int ctiles = p.Coutp / (16 * MT);
    const long long nitems = (long long)p.N * tiles_y * tiles_x * ctiles;
    if (nitems <= 0 || nitems > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    const int grid = PERSIST ? persistent_grid(nitems) : (int)nitems;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NTHREADS), C::LDS_BYTES, stream, p, tiles_x,
                       tiles_y, ctiles, (int)nitems);
    return (int)hipGetLastError();
}

template <int KS, int S, int TH, int MT>
int launch_t(const ConvParams& p, hipStream_t stream) {
#ifndef ESA_PERSIST_S1
#define ESA_PERSIST_S1 1
#endif
    if (KS == 3 && p.Cinp == 32 && (S == 2 || ESA_PERSIST_S1)) return launch_tp<KS, S, TH, MT, true>(p, stream);
#if ESA_CONV_RING
    if constexpr (KS == 3) {
        if (p.Cinp > 32) return launch_ring<S, TH, MT>(p, stream);
    }
#endif
    return launch_tp<KS, S, TH, MT, false>(p, stream);
}

}  // namespace

#ifndef ESA_S1W
#define ESA_S1W 1
#endif
#ifndef ESA_S2C32
#define ESA_S2C32 1
#endif
#ifndef ESA_C1X1
#define ESA_C1X1 1
#endif

namespace {
// which kernel serves a convolution.  The choice must not depend on the batch size: a crop's result is
// bit-identical in any batch (tests/test_gpu_parity.py).
enum Choice { C_NONE, C_STREAM_S1, C_TILE_S1_8, C_TILE_S1_16, C_STREAM_S2, C_TILE_S2, C_1X1, C_TILE_1X1, C_X6 };
Choice choose_conv(const ConvParams& p, int k, int stride) {
    if (p.fmt == FMT_F32) return conv_x6_supported(p, k, stride) ? C_X6 : C_NONE;      // f32 tensors: bf16x6 arithmetic (conv_x6.hip)
    if ((p.fmt == FMT_BF)) {         // single-bf16 tensors: the stream kernel (every 3x3) and the register 1x1 kernel only
        if (k == 3 && stride == 1) return conv_s2c32_supported(p) ? C_STREAM_S1 : C_NONE;
        if (k == 3 && stride == 2) return conv_s2c32_supported(p) ? C_STREAM_S2 : C_NONE;
        if (k == 1 && stride == 1) return conv1x1_supported(p) ? C_1X1 : C_NONE;
        return C_NONE;
    }
    if ((p.Cinp & 31) || (p.Coutp & 31)) return C_NONE;
    // one image is addressed with 32-bit byte offsets inside the kernels
    if ((long long)p.H * p.W * p.Cinp * 4 > 0x7fffffffLL) return C_NONE;
    if (k == 3 && stride == 1) {
        // the register-weight stream kernel (conv_s2c32.hip: cross-item prefetch, two barriers per chunk, no weight
        // traffic through LDS) wherever an image has at least 8 of its 16x16x32-cout items: measured 5-15 % faster
        // than the LDS weight ring on the 64/128/256-channel branches
        const long long items_per_image = (long long)((p.OH + 15) / 16) * ((p.OW + 15) / 16) * (p.Coutp / 32);
        // (very deep contractions amortise the ring's prologue and share each weight chunk between four waves
        // through LDS: the 480 -> 480 3x3 of seg_hrnet3 runs 436 TFLOP/s on the ring, 400 on the stream kernel)
#ifndef ESA_S1W_MAXC
#define ESA_S1W_MAXC 256
#endif
        if (ESA_S1W && items_per_image >= 8 && p.Cinp <= ESA_S1W_MAXC && conv_s2c32_supported(p)) return C_STREAM_S1;
        // deep, small-resolution layers have too few 16x16 tiles to fill 2 workgroups on every CU: halve the
        // tile height there
        if (p.Cinp > 32 && items_per_image < 12 && p.OH > 8) return C_TILE_S1_8;
        return C_TILE_S1_16;
    }
    if (k == 3 && stride == 2) return ESA_S2C32 && conv_s2c32_supported(p) ? C_STREAM_S2 : C_TILE_S2;
    if (k == 1 && stride == 1) return ESA_C1X1 && conv1x1_supported(p) ? C_1X1 : C_TILE_1X1;
    return C_NONE;
}
}  // namespace

// does launch_conv run the stride-1 stream kernel for this 3x3 convolution?
bool conv_is_stream_s1(const ConvParams& p) { return choose_conv(p, 3, 1) == C_STREAM_S1; }

int launch_conv(const ConvParams& p, int k, int stride, hipStream_t stream) {
    switch (choose_conv(p, k, stride)) {
        case C_STREAM_S1: return launch_conv_s1w(p, stream);
        case C_TILE_S1_8: return launch_t<3, 1, 8, 2>(p, stream);
        case C_TILE_S1_16: return launch_t<3, 1, 16, 2>(p, stream);
        case C_STREAM_S2: return launch_conv_s2c32(p, stream);
        case C_TILE_S2: return launch_t<3, 2, 4, 2>(p, stream);
        case C_1X1: return launch_conv1x1(p, stream);
        case C_TILE_1X1: return launch_t<1, 1, 16, 2>(p, stream);
        case C_X6: return launch_conv_x6(p, k, stride, stream);
        default: return (int)hipErrorInvalidValue;
    }
}

// name of the __global__ function launch_conv() runs for these parameters, as rocprofv3 prints it (without
// namespaces and argument list) — the per-launch tables of bench.py / tools/profile_ops.py group by it
const char* conv_kernel_name(const ConvParams& p, int k, int stride) {
    const bool ring = ESA_CONV_RING && p.Cinp > 32;
    const bool persist = p.Cinp == 32;
    if (p.fmt == FMT_F32) return choose_conv(p, k, stride) == C_X6 ? conv_x6_kernel_name(p, k, stride) : "none";
    if ((p.fmt == FMT_BF)) {
        switch (choose_conv(p, k, stride)) {
            case C_STREAM_S1: return "conv_s2c32_kernel<1, 8, 4, false, true>";
            case C_STREAM_S2: return "conv_s2c32_kernel<2, 4, 4, false, true>";
            case C_1X1: return "conv1x1_kernel<bf16>";
            default: return "none";
        }
    }
    switch (choose_conv(p, k, stride)) {
        case C_STREAM_S1: return use_th16(p) ? "conv_s2c32_kernel<1, 16, 4, false>" : p.Coutp % 64 == 0 ? "conv_s2c32_kernel<1, 8, 4, false>" : "conv_s2c32_kernel<1, 8, 2, false>";
        case C_TILE_S1_8: return ring ? "conv_mfma_ring_kernel<1, 8, 2>" : persist ? "conv_mfma_kernel<3, 1, 8, 2, true>" : "conv_mfma_kernel<3, 1, 8, 2, false>";
        case C_TILE_S1_16: return ring ? "conv_mfma_ring_kernel<1, 16, 2>" : persist ? "conv_mfma_kernel<3, 1, 16, 2, true>" : "conv_mfma_kernel<3, 1, 16, 2, false>";
        case C_STREAM_S2: return p.Coutp % 64 == 0 ? "conv_s2c32_kernel<2, 4, 4, false>" : "conv_s2c32_kernel<2, 4, 2, false>";
        case C_TILE_S2: return ring ? "conv_mfma_ring_kernel<2, 4, 2>" : persist ? "conv_mfma_kernel<3, 2, 4, 2, true>" : "conv_mfma_kernel<3, 2, 4, 2, false>";
        case C_1X1: return "conv1x1_kernel";
        case C_TILE_1X1: return "conv_mfma_kernel<1, 1, 16, 2, false>";
        default: return "none";
    }
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

size_t packed_weight_bytes_bf(int coutp, int cinp, int k) {
    return (size_t)(coutp / 16) * (cinp / 64) * k * k * 2048;
}

// BF mode (kernels.h): [cout16 tile][cin64 block][tap][K-step][lane 0..63][8 x bf16]; one rounding per weight
void pack_conv_weights_bf(const float* w, int cout, int cin, int k, int coutp, int cinp, void* dst) {
    uint16_t* d = static_cast<uint16_t*>(dst);
    const int taps = k * k, nblk = cinp / 64;
    for (int t16 = 0; t16 < coutp / 16; ++t16)
        for (int c = 0; c < nblk; ++c)
            for (int tap = 0; tap < taps; ++tap)
                for (int step = 0; step < 2; ++step)
                    for (int l = 0; l < 64; ++l)
                        for (int j = 0; j < 8; ++j) {
                            const int co = t16 * 16 + (l & 15), ci = c * 64 + step * 32 + 8 * (l >> 4) + j;
                            float v = 0.f;
                            if (co < cout && ci < cin) v = w[((size_t)co * cin + ci) * taps + tap];
                            d[(((((size_t)t16 * nblk + c) * taps + tap) * 2) + step) * 512 + l * 8 + j] = host_bf16(v);
                        }
}

}  // namespace esa
