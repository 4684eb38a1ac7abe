// conv_s2c32.hip — the "stream" 3x3 convolution (+ folded-BN bias, optional residual, ReLU) on the matrix
// cores, split-bf16, weights in registers:
//   stride 2: the fuse-down chains and the transition layers
//             (models/seg_hrnet.py:176-220 fuse_layers[i][j], j < i; :343-377 transition layers);
//   stride 1: the BasicBlock 3x3 convs of the 64-, 128- and 256-channel branches and of layer1.0
//             (models/seg_hrnet.py:32-61).
//
// The generic tiling (conv_mfma.hip: one output row per wave, weights AND input through LDS) is LDS-bound:
// 54 LDS reads per 54 MFMAs.  With 32-channel chunks the whole weight set of a 16-cout tile is
// 18 fragments = 72 VGPRs, so here
//   * a wave owns one cout tile for the workgroup's lifetime (persistent workgroups, cout-tile index
//     constant per workgroup): its weights come from global straight into registers, never touch LDS;
//   * every wave sweeps all rows of the pixel tile: each input-row fragment it reads from LDS feeds up to
//     three taps of three output rows (60 reads per 216 MFMAs at stride 1);
//   * the input tile is staged once per 64 (or 32) couts; the next step's tile is prefetched into
//     registers while the current one is consumed (issue-early / write-late, as in conv_mfma.hip).
// LDS holds only the 8 operand planes of the input tile (24..40 KB).
//
// Everything around the MFMA stream is kept short and branch-free, because a wave that is not issuing
// MFMAs leaves its SIMD's matrix pipe to ONE other wave (2 workgroups per CU):
//   * global accesses go through buffer descriptors: an out-of-image pixel / row / column is a single
//     select of an out-of-range offset (loads return 0 = the zero padding, stores are dropped), no
//     exec-mask branches, 32-bit offsets, the image base in the scalar offset of the LOADS (the stores keep it in
//     the vector offset: store-data hazard, see the epilogue);
//   * the item decode divides by multiplication (host-computed reciprocals);
//   * bias and weights of the next step are loaded a phase or more ahead of their use;
//   * LDS operand reads run one or two rows ahead of the MFMAs that consume them;
//   * in the last chunk of an item the epilogue of each output row rides inside the last MFMA phase.
#include "conv_cfg.h"
#include "devstate.h"
#include "kernels.h"
#include "sb.h"
#include <algorithm>
#include <atomic>
#include <type_traits>

#ifdef S2_TRACE
// debug build only (tools/trace_s2.py, -DS2_TRACE=<Cinp>): per-wave cycle stamps of the stream kernel's phases
__device__ unsigned long long g_s2_trace[64 * 4 * 8 * 16];
__device__ unsigned long long g_s2_wg[1024 * 2];       // wall clock (100 MHz) at start / end of every workgroup
extern "C" int esa_debug_s2_trace(void* dst) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_s2_trace), sizeof(g_s2_trace));
}
extern "C" int esa_debug_s2_wg(void* dst) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_s2_wg), sizeof(g_s2_wg));
}
#endif

// timing experiments (tools/ablate_any.sh): bit mask of parts compiled OUT — 1 X global loads, 2 X LDS writes,
// 4 weight reloads, 8 MFMAs, 16 barriers, 32 epilogue (all but one row), 64 LDS operand reads, 128 residual.
// 0 in the product build.
#ifndef S2_ABL
#define S2_ABL 0
#endif
#ifndef S2_INPHASE
#define S2_INPHASE 1      // epilogue rows inside the last MFMA phase (0: after it)
#endif
#ifndef S2_RD
#define S2_RD 2          // LDS read-ahead distance (input rows)
#endif

namespace esa {
namespace {

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

constexpr uint32_t OOB = 0x80000000u;      // offset >= every descriptor's num_records (all < 2^31)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

// tile stream geometry; m_* = floor((2^32 - 1) / d): q = umulhi(b, m) is b / d or one less
struct StreamGeo {
    int tiles_x, tiles_y, ctiles, nitems;
    uint32_t m_ct, m_tx, m_ty;
};
__device__ __forceinline__ int div_magic(int b, int d, uint32_t m) {
    int q = (int)__umulhi((uint32_t)b, m);
    if (b - q * d >= d) ++q;
    return q;
}

// MW = cout tiles (of 16) per workgroup: 4 -> wave = cout tile, TH output rows each;
//                                        2 -> wave = (cout tile, row half), TH/2 output rows each
// The workgroup walks a stream of (item, chunk) steps; X of step s+1 is prefetched into registers while
// step s is consumed; the weight registers are a ring of three kx-thirds: as soon as phase kx of step s
// is done, the (kx) third of step s+1 is loaded into the same registers (3 phases of cover, no extra
// VGPRs).  Single-chunk layers whose cout slice does not change keep their weights for the whole launch.
// MH: multi-head launch (ConvParams::nheads > 1): the output tensor, its channel count and the ReLU flag depend on
// the cout slice of the item.
// BF: the tensors are single bf16 (sb.h "BF", esahrnet_cfg.precision 1).  A step stages one 128-byte block of 64
// channels per pixel — the same loads, planes and reads as a 32-channel split chunk — whose plane pair (2g, 2g+1) holds
// the block's two MFMA K-steps instead of (hi, lo); the weight registers wh / wl hold the K-step 0 / 1 fragments; a
// tap of a row is 2 MFMAs for 64 channels instead of 3 for 32; accumulators leave as 8 bytes of bf16 per lane.
// DB: two tile buffers in LDS and ONE barrier per step.  The tile of step s+1 (in registers since step s-1) is written
// into the other buffer right behind the barrier that opens step s, while the first operand reads of step s are in
// flight, and the loads of step s+2 are issued behind it; the barrier that opens step s+1 then publishes that tile
// and retires the reads of step s at once.  (Single-buffered: barrier, write, barrier, and the LDS-read pipe starts
// cold behind the second one.)
// OCC: workgroups per CU the register budget is cut for (2: 256 VGPRs per wave; 1: 512 — the 16-row tile).
// The body is shared by the one-convolution kernel and by conv_s2c32_jobs_kernel (several independent convolutions in
// one launch): `bid` / `G` are the workgroup's index and the grid size WITHIN its convolution.
template <int S, int TH, int MW, bool MH, bool BF, bool DB, int OCC>
__device__ __forceinline__ void conv_s2c32_body(const ConvParams& p, const StreamGeo& geo, const int bid, const int G) {
    using S2C = ConvCfg<3, S, TH, 2>;
    constexpr int RG = 4 / MW;                  // row groups
    constexpr int NT = TH / RG;                 // output rows per wave
    constexpr int ROWS = (NT - 1) * S + 3;      // input rows a wave touches
#ifndef S2_RD84
#define S2_RD84 1
#endif
    // LDS read-ahead in input rows: as many as the register budget allows (the bf16 64-cout variant spills 8 VGPRs at 1
    // and its big layers are HBM-bound: 0)
    constexpr int RD = OCC == 1 ? 2 : (S == 1 && MW == 2) ? S2_RD : (S == 1 && MW == 4) ? (BF ? 0 : S2_RD84) : 1;
    constexpr bool INPHASE = S2_INPHASE;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* xs = smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4;
    const int mw = wave % MW, rg = wave / MW;
    constexpr int EB = BF ? 2 : 4;             // bytes per channel
    const int nchunks = p.Cinp >> (BF ? 6 : 5);
    const int pixb = p.Cinp * EB;
    int opix = p.Coutp * EB;                                             // (per item in a multi-head launch)
    const int ximg = p.H * p.W * pixb;
    int yimg = p.OH * p.OW * opix;                                       // bytes per image (< 2^31, host-checked)
    int item = xcd_contiguous(bid, G);
    if (item >= geo.nitems) return;
#ifdef S2_TRACE
    const unsigned long long t_begin = clock64();
    const bool wgon = S == 1 && p.Cinp == S2_TRACE && bid < 1024 && tid == 0;
    if (wgon) g_s2_wg[bid * 2] = wall_clock64();
#endif
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x, (uint32_t)p.N * (uint32_t)ximg);
    __amdgpu_buffer_rsrc_t ry = make_rsrc(MH ? p.yh[0] : p.y, (uint32_t)p.N * (uint32_t)yimg);
    int relu = p.relu;
    const __amdgpu_buffer_rsrc_t rr = make_rsrc(p.res ? p.res : p.y, p.res ? (uint32_t)p.N * (uint32_t)yimg : 0u);

    // ---- staging map: thread -> (operand plane jst, tile pixel q0 + 32*it), fixed for the launch ----
    const int jst = tid & 7, q0 = tid >> 3;
    char* xwr = xs + S2C::plane_off(BF ? bf_plane_of_chunk(jst) : jst) + q0 * 16;
    int qyx[S2C::XITER];                        // tile-local (row << 8 | column) of the pixel, -1 beyond the tile
#pragma unroll
    for (int it = 0; it < S2C::XITER; ++it) {
        const int q = q0 + it * 32;
        const int qy = q / S2C::IW, qx = q - qy * S2C::IW;
        qyx[it] = q < S2C::NPIX ? (qy << 8 | qx) : -1;
    }
    uint32_t xg[S2C::XITER];                    // byte offsets inside the image of the item being prefetched
    int s_n, s_oy0, s_ox0, s_ct;
    auto decode = [&](int it_) {
        const int q1 = div_magic(it_, geo.ctiles, geo.m_ct);
        s_ct = it_ - q1 * geo.ctiles;
        const int q2 = div_magic(q1, geo.tiles_x, geo.m_tx);
        const int tx = q1 - q2 * geo.tiles_x;
        s_n = div_magic(q2, geo.tiles_y, geo.m_ty);
        const int ty = q2 - s_n * geo.tiles_y;
        s_oy0 = ty * TH;
        s_ox0 = tx * TW;
        const int gy0 = s_oy0 * S - 1, gx0 = s_ox0 * S - 1;
#pragma unroll
        for (int it = 0; it < S2C::XITER; ++it) {
            const int gy = gy0 + (qyx[it] >> 8), gx = gx0 + (qyx[it] & 255);
            const bool inside = qyx[it] >= 0 && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
            xg[it] = inside ? (uint32_t)((gy * p.W + gx) * pixb + jst * 16) : OOB;
        }
    };
    u32x4 xr[S2C::XITER];
    auto prefetch = [&](int n_, int ch) {
        const int so = n_ * ximg + ch * 128;
#pragma unroll
        for (int it = 0; it < S2C::XITER; ++it)
            if (!(S2_ABL & 1)) xr[it] = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)xg[it], so, 0);
    };
    // weights of (cout slice CT, chunk CH), third KX -> registers wh/wl[ky*3 + KX]
    bf16x8 wh[9], wl[9];
#define S2_LOAD_W(CT, CH, KX)                                                                     \
    {                                                                                             \
        const uint4* ws_ = p.w + ((size_t)((CT) * MW + mw) * nchunks + (CH)) * (9 * 128) + lane;  \
        _Pragma("unroll") for (int ky = 0; ky < 3; ++ky) {                                        \
            wh[ky * 3 + (KX)] = __builtin_bit_cast(bf16x8, ws_[((ky * 3 + (KX)) * 2 + 0) * 64]);  \
            wl[ky * 3 + (KX)] = __builtin_bit_cast(bf16x8, ws_[((ky * 3 + (KX)) * 2 + 1) * 64]);  \
        }                                                                                         \
    }

    const char* xrd0 = xs + S2C::plane_off(2 * g) + ((rg * NT * S) * S2C::IW + (lane & 15) * S) * 16;
    auto write_tile = [&](int b_) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < S2C::XITER; ++it)
            if (!(S2_ABL & 2) && q0 + it * 32 < S2C::NPIX)
                *reinterpret_cast<u32x4*>(xwr + (DB ? b_ * S2C::XBYTES : 0) + it * 512) = xr[it];
    };

    decode(item);
    prefetch(s_n, 0);
    int n = s_n, oy0 = s_oy0, ox0 = s_ox0, ct = s_ct;      // the item being computed
    int c = 0;
    bool first = true;
    f32x4 acc[NT];
    // DB look-ahead: step1 = the step behind the current one (its tile is in xr), coordinates of its item in s1_*
    int buf = 0, item1 = item, c1 = 0, s1_n = s_n, s1_oy0 = s_oy0, s1_ox0 = s_ox0, s1_ct = s_ct;
    bool ok1 = false;
    if (DB) {
        write_tile(0);                                           // tile of step 0; published by the first barrier
        ok1 = nchunks > 1 || item + G < geo.nitems;
        if (nchunks > 1) {
            c1 = 1;
        } else if (ok1) {
            item1 = item + G;
            decode(item1);
        }
        s1_n = s_n; s1_oy0 = s_oy0; s1_ox0 = s_ox0; s1_ct = s_ct;
        prefetch(s_n, c1);
    }
    // the first weights go out BEHIND the tile loads, as in every later step: the counted vmcnt that hipcc puts in front
    // of the tile commit at the loop head is the minimum over the loop entry and the back edge, and with nothing younger
    // than the tile on the entry path it would drain the weight loads of every step
    __builtin_amdgcn_sched_barrier(0);          // (hipcc would hoist the weight loads above the tile loads)
    S2_LOAD_W(ct, 0, 0)
    S2_LOAD_W(ct, 0, 1)
    S2_LOAD_W(ct, 0, 2)
    f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + (ct * MW + mw) * 16 + g * 4);
#ifdef S2_TRACE
    int tstep = 0;
    // traced: the first 32 workgroups and the first 32 of the second half (the second workgroup of a CU)
    const int tslot = bid < 32 ? bid : bid - (G >> 1) + 32;
    const bool ton = S == 1 && p.Cinp == S2_TRACE && tslot >= 0 && tslot < 64 && (bid < 32 || bid >= (G >> 1)) && lane == 0;
    unsigned long long* const trow = g_s2_trace + ((ton ? tslot : 0) * 4 + wave) * 8 * 16;
#define TR(EV) if (ton && tstep < 8) trow[tstep * 16 + (EV)] = clock64();
    if (ton) {
        trow[15] = __builtin_amdgcn_s_getreg((31 << 11) | 4);      // HW_ID
        trow[9] = wall_clock64();
        trow[10] = t_begin;
    }
#else
#define TR(EV)
#endif
    while (true) {
        TR(0)
        const bool last_chunk = c + 1 == nchunks;
        const bool more = DB ? ok1 : (!last_chunk || item + G < geo.nitems);          // is there a step s+1?
        const char* xrd = xrd0 + (DB ? buf * S2C::XBYTES : 0);
        bf16x8 fh[RD + 1], fo[RD + 1];
#define S2_READ(IDX)                                                                              \
        {                                                                                         \
            const int off_ = (((IDX) % ROWS) * S2C::IW + (IDX) / ROWS) * 16;                      \
            if (!(S2_ABL & 64)) {                                                                 \
                fh[(IDX) % (RD + 1)] = *reinterpret_cast<const bf16x8*>(xrd + off_);              \
                fo[(IDX) % (RD + 1)] = *reinterpret_cast<const bf16x8*>(xrd + off_ + S2C::LO_OFF); \
            } else if ((IDX) < RD + 1) {                                                          \
                fh[(IDX) % (RD + 1)] = wh[(IDX) % 9];                                             \
                fo[(IDX) % (RD + 1)] = wl[(IDX) % 9];                                             \
            }                                                                                     \
        }
        int nct = ct, nch = c + 1;
        int item2 = item1, c2 = c1;
        bool ok2 = false;
        if (DB) {
            if (!(S2_ABL & 16)) __syncthreads();      // tile of this step published, reads of the previous step retired
            TR(3)
#pragma unroll
            for (int r = 0; r < RD; ++r) S2_READ(r)    // start the operand pipe before the staging work
            __builtin_amdgcn_sched_barrier(0);
            if (ok1) write_tile(buf ^ 1);              // tile of step s+1 -> the buffer step s-1 was read from
            // step s+2 = the step behind step1: next chunk of its item, or chunk 0 of the workgroup's next item (beyond
            // the end of the stream the last tile is fetched again: harmless, keeps the loop free of branches around loads)
            nct = s1_ct; nch = c1;
            ok2 = ok1;
            if (c1 + 1 < nchunks) {
                c2 = c1 + 1;
            } else {
                c2 = 0;
                item2 = item1 + G;
                ok2 = ok1 && item2 < geo.nitems;
                if (ok2) decode(item2);
            }
            prefetch(s_n, c2);
            TR(4)
        } else {
            if (!first && !(S2_ABL & 16)) __syncthreads();            // previous step's MFMAs are done reading the planes
            first = false;
            TR(1)
            write_tile(0);
            TR(2)
            if (!(S2_ABL & 16)) __syncthreads();
            TR(3)
            // step s+1: next chunk of this item, or chunk 0 of the workgroup's next item (the last step of the
            // stream prefetches its own tile again: harmless, keeps the loop free of branches around loads)
            if (last_chunk) {
                nch = 0;
                if (more) {
                    decode(item + G);
                    nct = s_ct;
                }
            }
            prefetch(s_n, nch);
            TR(4)
        }
        constexpr bool S2_STATIC_LOADS = DB;
        // DB variant: loads are issued UNCONDITIONALLY (the weight thirds of the next step even when they are the ones already held,
        // the residual rows even when the layer has none: an out-of-range offset costs an issue slot, no traffic): with a
        // load behind a run-time condition hipcc cannot count what is in flight and falls back to vmcnt(0) — observed:
        // the tile commit at the start of every step waited for the weight third issued just before the barrier, the
        // residual fold for the third issued just before it.  (Costs redundant weight loads on single-chunk layers; measured
        // on MI355X, batch 32: with two workgroups per CU neither this nor the second barrier moves a layer — the partner
        // wave fills every stall — so the variant is used where a launch has at most one workgroup per CU.)
        const bool reload = S2_STATIC_LOADS ? !(S2_ABL & 4) : (!(S2_ABL & 4) && more && (nchunks > 1 || nct != ct));
        int co = (ct * MW + mw) * 16 + g * 4;                   // cout of the lane's accumulator quad ...
        if (MH) {                                               // ... inside its head's tensor
            const int cb = (ct * MW + mw) * 16;                 // per wave: a 64-cout slice may span two heads
            const int hsel = (cb >= p.hb[1] ? 1 : 0) + (p.nheads > 2 && cb >= p.hb[2] ? 1 : 0);
            co -= p.hb[hsel];
            opix = (p.hb[hsel + 1] - p.hb[hsel]) * EB;
            yimg = p.OH * p.OW * opix;
            ry = make_rsrc(p.yh[hsel], (uint32_t)p.N * (uint32_t)yimg);
            relu = p.hrelu[hsel];
        }
        if (c == 0) {
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = bv;
        }
        // output rows of this wave: 32-bit offsets inside the image, OOB for pixels outside it
        const int rox = ox0 + (lane & 15), roy = oy0 + rg * NT;
        const uint32_t o0 = rox < p.OW ? (uint32_t)((roy * p.OW + rox) * opix + (BF ? co * 2 : chunk_ofs(co, g))) : OOB;
        const int orow = p.OW * opix, yso = n * yimg;
        const int nrows = p.OH - roy;                                   // rows t < nrows exist
        // residual (last chunk only): the first half of the wave's rows is loaded at the start of the step and folded
        // into the accumulators after the second phase; the second half is loaded then and folded in by the row's
        // epilogue.  16-byte chunks (sb.h).
        // Residual (last chunk only): the first half of the wave's rows is loaded at the start of the step and folded into
        // the accumulators after the second phase; the second half is loaded then and folded in while the last phase runs.
        //   split format: 16-byte chunks, halves swapped into accumulator quads (sb.h), f32 adds on the VALU;
        //   bf16 format:  through the matrix pipe, acc += I·r — one more MFMA per output row whose A operand is a 16x32
        //                 selector (idA) and whose B operand is the residual AS IT LIES IN MEMORY: lane (pixel, g < 2)
        //                 loads the 16-byte chunk of 8 channels 8g.. of the wave's cout tile, lanes g >= 2 nothing (an
        //                 out-of-range offset).  16-byte loads instead of 8-byte ones, no unpacking: 287 -> 214 us on the
        //                 64-channel 192x192 layers of W48, which are HBM-bound.  (Tried for the split format too, hi and lo
        //                 chunks in g < 2 / g >= 2: correct, ~70 VALU instructions fewer per item, but the selector's 4
        //                 registers pushed the kernel from 2 to 18 spilled VGPRs and every layer lost 10-15 %.)
        constexpr int NH = NT >= 2 ? NT / 2 : 1;
        u32x4 rc[NH];
        const int rfl = relu_floor(relu);                        // branch-free optional ReLU (see sb.h)
        const bool do_res = !(S2_ABL & 128) && last_chunk && p.res != nullptr;
        const int ctb = (ct * MW + mw) * 16;                    // first cout of the wave's tile (no residual in multi-head launches)
        const uint32_t r0 = !BF ? o0 : (rox < p.OW && g < 2 ? (uint32_t)((roy * p.OW + rox) * opix + (ctb + 8 * g) * 2) : OOB);
        auto res_load = [&](int half) __attribute__((always_inline)) {
#pragma unroll
            for (int t = 0; t < NH; ++t) {
                const int tt = half * NH + t;
                const uint32_t ro_ = (tt < NT && tt < nrows && (!S2_STATIC_LOADS || do_res)) ? r0 + (uint32_t)(tt * orow) : OOB;
                rc[t] = __builtin_amdgcn_raw_buffer_load_b128(rr, (int)ro_, yso, 0);
            }
        };
        // bf16 format: the selector A[r][k] = 1 where k = 8g + j is cout r of the tile (g < 2).  Rebuilt in every step
        // that folds a residual (8 VALU) instead of living in 4 VGPRs for the whole launch (the empty asm pins it here).
        u32x4 idw = {0u, 0u, 0u, 0u};
        if (BF && do_res) {
            int jj = (lane & 15) - 8 * g;
            asm volatile("" : "+v"(jj));
            const uint32_t one = (jj & 1) ? 0x3f800000u : 0x00003f80u;
            const bool on = jj >= 0 && jj < 8 && g < 2;
#pragma unroll
            for (int k = 0; k < 4; ++k) idw[k] = (on && (jj >> 1) == k) ? one : 0u;
        }
        const bf16x8 idA = __builtin_bit_cast(bf16x8, idw);
        auto res_fold = [&](int t) __attribute__((always_inline)) {          // acc[t] += residual row t (its chunk in rc[t % NH])
            if (BF) {
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(idA, __builtin_bit_cast(bf16x8, rc[t % NH]), acc[t], 0, 0, 0);
            } else {
                uint2 rh_, rl_;
                chunk_to_quad(make_uint4(rc[t % NH][0], rc[t % NH][1], rc[t % NH][2], rc[t % NH][3]), rh_, rl_);
                float r_[4];
                join4(rh_, rl_, r_);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[t][i] += r_[i];
            }
        };
        auto epilogue_row = [&](int t, auto res_c) __attribute__((always_inline)) {
            if (!BF && decltype(res_c)::value && NT >= 2 && t >= NH) res_fold(t);
            float v[4] = {acc[t][0], acc[t][1], acc[t][2], acc[t][3]};
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = relu_opt(v[i], rfl);
            const uint32_t so_ = t < nrows ? o0 + (uint32_t)(t * orow) : OOB;
            if (BF) {       // 4 channels of one pixel = 8 bytes per lane; the four rows of 16 lanes fill 32 contiguous bytes
                typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
                const uint2 pk = pack4_bf16(v);
                __builtin_amdgcn_raw_buffer_store_b64(u32x2{pk.x, pk.y}, ry, (int)(so_ + (uint32_t)yso), 0, 0);
                return;
            }
            uint2 hi, lo;
            split4(v, hi, lo);
            const uint4 ch = quad_to_chunk(hi, lo);
            const u32x4 cv = {ch.x, ch.y, ch.z, ch.w};
            // The image base goes into the VECTOR offset, the scalar offset stays the constant 0: gfx950 reads the
            // data registers of a 16-byte buffer store late (lanes 12..15 of every row of 16), so the instruction
            // behind it must not overwrite them — and hipcc 7.2 pads that hazard only for stores WITHOUT an
            // soffset register (tools/ubench/store_data_war.hip; with `yso` as soffset the epilogue of the next
            // row landed in this row's columns 12..15 whenever the scheduler put it right behind the store).
            __builtin_amdgcn_raw_buffer_store_b128(cv, ry, (int)(so_ + (uint32_t)yso), 0, 0);
        };
        if (S2_STATIC_LOADS || do_res) res_load(0);
        // one kx phase.  The third phase of a last chunk is straight-line code of its own: the epilogue of output
        // row t (ReLU, split, chunk swap, store: ~20 VALU instructions) is emitted one input row after the row's
        // last MFMA and rides in the issue shadow of the next row's MFMAs instead of running after the phase with
        // the matrix pipe idle (measured: the epilogues were 13 % of a layer, 6 % of the forward).
        auto phase = [&](auto kx_c, auto last_c, auto res_c) __attribute__((always_inline)) {
            constexpr int kx = decltype(kx_c)::value;
            constexpr bool EPI = INPHASE && decltype(last_c)::value && kx == 2 && !(S2_ABL & 32);
#pragma unroll
            for (int i = 0; i < ROWS; ++i) {
                const int idx = kx * ROWS + i;
                if (idx + RD < 3 * ROWS) {
                    S2_READ(idx + RD)
                    __builtin_amdgcn_sched_barrier(0);       // keep the read ahead of this row's MFMAs
                }
                // output row (i - 3) / S got its last MFMA one input row ago
                if (EPI && i >= 3 && (i - 3) % S == 0 && (i - 3) / S < NT) epilogue_row((i - 3) / S, res_c);
                const bf16x8 xh = fh[idx % (RD + 1)];
                const bf16x8 xo = fo[idx % (RD + 1)];
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int d = i - ky;
                    if (!(S2_ABL & 8) && d >= 0 && d % S == 0 && d / S < NT) {
                        const int t = d / S;
                        if (BF) {
                            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[ky * 3 + kx], xh, acc[t], 0, 0, 0);
                            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[ky * 3 + kx], xo, acc[t], 0, 0, 0);
                        } else {
                            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[ky * 3 + kx], xh, acc[t], 0, 0, 0);
                            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[ky * 3 + kx], xo, acc[t], 0, 0, 0);
                            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[ky * 3 + kx], xh, acc[t], 0, 0, 0);
                        }
                        // the row's last MFMA of the item: the second half of the residual rows rides behind it
                        if (BF && decltype(res_c)::value && kx == 2 && ky == 2 && NT >= 2 && t >= NH) res_fold(t);
                    }
                }
            }
            if (EPI) {
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    if (t * S + 3 > ROWS - 1) epilogue_row(t, res_c);      // rows finished by the final input rows
            }
        };
        if (!DB) {
#pragma unroll
            for (int r = 0; r < RD; ++r) S2_READ(r)
        }
        phase(std::integral_constant<int, 0>{}, std::false_type{}, std::false_type{});
        if (reload) S2_LOAD_W(nct, nch, 0)          // this third is free: refill it for step s+1
        TR(5)
        phase(std::integral_constant<int, 1>{}, std::false_type{}, std::false_type{});
        if (reload) S2_LOAD_W(nct, nch, 1)
        if (do_res) {
#pragma unroll
            for (int t = 0; t < NH; ++t) res_fold(t);
            if (!S2_STATIC_LOADS && NT >= 2) res_load(1);
        }
        if (S2_STATIC_LOADS && NT >= 2) res_load(1);
        TR(6)
        if (!last_chunk) phase(std::integral_constant<int, 2>{}, std::false_type{}, std::false_type{});
        else if (do_res) phase(std::integral_constant<int, 2>{}, std::true_type{}, std::true_type{});
        else phase(std::integral_constant<int, 2>{}, std::true_type{}, std::false_type{});
        if (reload) S2_LOAD_W(nct, nch, 2)
        TR(7)
#undef S2_READ
        if (last_chunk && !(INPHASE && !(S2_ABL & 32))) {
#pragma unroll
            for (int t = 0; t < ((S2_ABL & 32) ? 1 : NT); ++t) {
                if (do_res) epilogue_row(t, std::true_type{});
                else epilogue_row(t, std::false_type{});
            }
        }
        if (last_chunk) {
            TR(8)
#ifdef S2_TRACE
            if (ton && !more) {
                trow[11] = clock64();
                trow[12] = wall_clock64();
            }
#endif
#ifdef S2_TRACE
            if (wgon && !more) g_s2_wg[bid * 2 + 1] = wall_clock64();
#endif
            if (!more) break;
            if (nct != ct) bv = *reinterpret_cast<const f32x4*>(p.bias + (nct * MW + mw) * 16 + g * 4);
            item += G;
            if (DB) { n = s1_n; oy0 = s1_oy0; ox0 = s1_ox0; ct = s1_ct; }
            else { n = s_n; oy0 = s_oy0; ox0 = s_ox0; ct = s_ct; }
            c = 0;
        } else {
            ++c;
        }
        if (DB) {                                   // step1 <- step2
            item1 = item2; c1 = c2; ok1 = ok2;
            s1_n = s_n; s1_oy0 = s_oy0; s1_ox0 = s_ox0; s1_ct = s_ct;
            buf ^= 1;
        }
#ifdef S2_TRACE
        ++tstep;
#endif
    }
#undef TR
#undef S2_LOAD_W
}

template <int S, int TH, int MW, bool MH = false, bool BF = false, bool DB = false, int OCC = 2>
__global__ __launch_bounds__(NTHREADS, OCC) void conv_s2c32_kernel(ConvParams p, StreamGeo geo) {
    conv_s2c32_body<S, TH, MW, MH, BF, DB, OCC>(p, geo, (int)blockIdx.x, (int)gridDim.x);
}

// Several INDEPENDENT convolutions (the same-depth 3x3s of the 64/128/256-channel branches of an HRModule,
// models/seg_hrnet.py:143-174: the branches do not talk to each other between two fuse layers) in one launch.
// Workgroups [start[j], start[j+1]) run convolution j exactly as its own launch would — same items per workgroup,
// same order, same bits — but the launch gap is paid once, the 256-channel convolution (one workgroup per CU on its
// own, i.e. one wave per SIMD) shares the CUs with the others, and one convolution's tail overlaps the next one's
// start.  Longest workgroups first (the dispatcher hands out workgroups in index order).
// A convolution with a multiple of 64 couts runs the 64-cout tiling (MW = 4), one with 32 (mod 64) the 32-cout one (MW = 2):
// both bodies live in the kernel, a workgroup takes the one its convolution needs.  The 32-channel branch's convolutions
// are HBM-bound (67 MB in, 67 MB out for 9.7 GFLOP), the deeper branches' matrix-bound: side by side on the CUs they
// overlap instead of adding up.
constexpr int MAXJOBS = 4;
struct StreamJobs {
    ConvParams p[MAXJOBS];
    StreamGeo geo[MAXJOBS];
    int start[MAXJOBS + 1];
    int mw[MAXJOBS];
    int njobs;
};
template <int S, int TH, bool BF>
__global__ __launch_bounds__(NTHREADS, 2) void conv_s2c32_jobs_kernel(StreamJobs jobs) {
    const int b = (int)blockIdx.x;
    int j = 0;
#pragma unroll
    for (int k = 1; k < MAXJOBS; ++k) j += (k < jobs.njobs && b >= jobs.start[k]) ? 1 : 0;
    const int bid = b - jobs.start[j], G = jobs.start[j + 1] - jobs.start[j];
    if (!BF && jobs.mw[j] == 2) conv_s2c32_body<S, TH, 2, false, BF, false, 2>(jobs.p[j], jobs.geo[j], bid, G);
    else conv_s2c32_body<S, TH, 4, false, BF, false, 2>(jobs.p[j], jobs.geo[j], bid, G);
}

uint32_t magic_of(int d) { return (uint32_t)(0xffffffffull / (uint64_t)d); }

// images one launch may cover: the kernels address whole tensors with 31-bit byte offsets (buffer descriptors, OOB
// marker 2^31).  Larger batches are cut into several launches over image ranges HERE, so that the kernel serving a
// layer — and with it every bit of a crop's result — never depends on the batch size.
std::atomic<long long> g_launch_limit{0x7fffffffLL};      // bytes; lowered only by tests (esahrnet_debug_set_launch_limit)
int images_per_launch(const ConvParams& p) {
    const int eb = (p.fmt == FMT_BF) ? 2 : 4;
    long long per = (long long)p.H * p.W * p.Cinp * eb;
    if (p.nheads > 1) {
        for (int h = 0; h < p.nheads; ++h) per = std::max(per, (long long)p.OH * p.OW * (p.hb[h + 1] - p.hb[h]) * eb);
    } else {
        per = std::max(per, (long long)p.OH * p.OW * p.Coutp * eb);
    }
    return (int)std::min<long long>(p.N, g_launch_limit.load(std::memory_order_relaxed) / std::max(per, 1LL));
}

#ifndef S2_DB
#define S2_DB 1          // 1: stride-1 launches with at most one workgroup per CU run the double-buffered variant
#endif
template <int S, int TH, int MW, bool MH, bool BF, bool DB, int OCC = 2>
int launch_s2c32_k(const ConvParams& p, const StreamGeo& geo, int grid, hipStream_t stream) {
    using S2C = ConvCfg<3, S, TH, 2>;
    constexpr int LDS = (DB ? 2 : 1) * S2C::XBYTES;
    auto kern = conv_s2c32_kernel<S, TH, MW, MH, BF, DB, OCC>;
    if (const int e_ = ensure_dyn_lds(reinterpret_cast<const void*>(kern), LDS)) return e_;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NTHREADS), LDS, stream, p, geo);
    return (int)hipGetLastError();
}

template <int S, int TH, int MW, bool MH = false, bool BF = false>
int launch_s2c32_t(const ConvParams& p, hipStream_t stream) {
    constexpr int EB = BF ? 2 : 4;
    const int nmax = images_per_launch(p);
    if (nmax < 1) return (int)hipErrorInvalidValue;
    if (p.N > nmax) {
        for (int n0 = 0; n0 < p.N; n0 += nmax) {
            ConvParams q = p;
            q.N = std::min(nmax, p.N - n0);
            q.x = p.x + (size_t)n0 * p.H * p.W * p.Cinp * EB;
            if (p.y) q.y = p.y + (size_t)n0 * p.OH * p.OW * p.Coutp * EB;
            if (p.res) q.res = p.res + (size_t)n0 * p.OH * p.OW * p.Coutp * EB;
            for (int h = 0; h < p.nheads && h < 3; ++h)
                if (p.yh[h]) q.yh[h] = p.yh[h] + (size_t)n0 * p.OH * p.OW * (p.hb[h + 1] - p.hb[h]) * EB;
            if (const int e = launch_s2c32_t<S, TH, MW, MH, BF>(q, stream)) return e;
        }
        return 0;
    }
    StreamGeo geo;
    geo.tiles_x = (p.OW + TW - 1) / TW;
    geo.tiles_y = (p.OH + TH - 1) / TH;
    geo.ctiles = p.Coutp / (16 * MW);
    const long long nitems = (long long)p.N * geo.tiles_y * geo.tiles_x * geo.ctiles;
    if (nitems <= 0 || nitems > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    geo.nitems = (int)nitems;
    geo.m_ct = magic_of(geo.ctiles);
    geo.m_tx = magic_of(geo.tiles_x);
    geo.m_ty = magic_of(geo.tiles_y);
    const int cus = device_cus(), slots = (TH == 16 ? 1 : 2) * cus;
    int grid = (int)(nitems < slots ? nitems : slots);
    if (grid > geo.ctiles) grid -= grid % geo.ctiles;   // grid stride keeps the cout slice of a workgroup constant
    // All variants run the same MFMAs in the same order on every accumulator: a crop's result does not depend on which
    // one a batch size selects.
    if constexpr (TH == 16) {
        return launch_s2c32_k<S, TH, MW, MH, BF, true, 1>(p, geo, grid, stream);
    } else {
        if constexpr (S2_DB && S == 1 && !MH) {
            if (nitems <= cus) return launch_s2c32_k<S, TH, MW, MH, BF, true>(p, geo, grid, stream);
        }
        return launch_s2c32_k<S, TH, MW, MH, BF, false>(p, geo, grid, stream);
    }
}

}  // namespace

// test hook: bytes one stream-kernel launch may address (default and maximum 2^31 - 1)
void set_stream_launch_limit(long long bytes) {
    g_launch_limit.store(bytes > 0 && bytes < 0x7fffffffLL ? bytes : 0x7fffffffLL, std::memory_order_relaxed);
}

// one IMAGE must be addressable with 31-bit byte offsets (buffer descriptors, OOB marker 2^31); a batch that is not
// is cut into image ranges by the launcher (images_per_launch), so this predicate does not look at N
bool conv_s2c32_supported(const ConvParams& p) {
    if ((p.fmt == FMT_BF)) return (p.Cinp & 63) == 0 && p.Cinp >= 64 && (p.Coutp & 63) == 0 && !p.out_f32 && p.nheads <= 1 &&
                     (long long)p.H * p.W * p.Cinp * 2 < 0x7fffffffLL && (long long)p.OH * p.OW * p.Coutp * 2 < 0x7fffffffLL;
    return (p.Cinp & 31) == 0 && p.Cinp >= 32 && (p.Coutp & 31) == 0 && !p.out_f32 &&
           (long long)p.H * p.W * p.Cinp * 4 < 0x7fffffffLL &&
           (long long)p.OH * p.OW * p.Coutp * 4 < 0x7fffffffLL;
}

int launch_conv_s2c32(const ConvParams& p, hipStream_t stream) {
    if (!conv_s2c32_supported(p)) return (int)hipErrorInvalidValue;
    if ((p.fmt == FMT_BF)) return launch_s2c32_t<2, 4, 4, false, true>(p, stream);
    return (p.Coutp % 64 == 0) ? launch_s2c32_t<2, 4, 4>(p, stream) : launch_s2c32_t<2, 4, 2>(p, stream);
}

bool conv_s2c32_multi_supported(const ConvParams& p) {
    if ((p.fmt == FMT_BF)) return false;
    if (p.nheads < 2 || p.nheads > 3 || p.res || p.out_f32 || p.hb[0] != 0 || p.hb[p.nheads] != p.Coutp) return false;
    if ((p.Cinp & 31) || p.Cinp < 32 || (long long)p.H * p.W * p.Cinp * 4 >= 0x7fffffffLL) return false;
    for (int h = 0; h < p.nheads; ++h) {
        const int c = p.hb[h + 1] - p.hb[h];
        if (c <= 0 || (c & 31) || !p.yh[h] || (long long)p.OH * p.OW * c * 4 >= 0x7fffffffLL) return false;
    }
    return true;
}

// heads are multiples of 32 couts and a wave owns 16: every wave lies inside one head
int launch_conv_s2c32_multi(const ConvParams& p, hipStream_t stream) {
    if (!conv_s2c32_multi_supported(p)) return (int)hipErrorInvalidValue;
    return (p.Coutp % 64 == 0) ? launch_s2c32_t<2, 4, 4, true>(p, stream) : launch_s2c32_t<2, 4, 2, true>(p, stream);
}

// the same scheme for stride 1: 64 couts x 8 rows per workgroup (wave = cout tile) where the layer has a
// multiple of 64 couts, else 32 couts x 8 rows (wave = cout tile x 4 rows; the 16-row tile of earlier builds sat at
// 256 VGPRs and was 4-14 % slower)
#ifndef S2_TH16
#define S2_TH16 0
#endif
// Experiment kept for the record (-DS2_TH16=1): ONE workgroup per CU with the full register file on a 16-row tile
// (64 couts x 16 rows x 16 columns, 432 MFMAs per wave and barrier, read-ahead 2).  Measured on MI355X, batch 32:
// 31-33 us against 25-26 us on the 64-channel layers, 27 against 22-23 on the 128-channel ones — a lone wave per SIMD
// loses more to its own stalls than the longer step wins back — and the residual is folded at another point of the
// accumulation, so results are not bit-identical to the 8-row tile.  Not used.
bool use_th16(const ConvParams& p) {
    if (!S2_TH16 || (p.fmt == FMT_BF) || p.Coutp % 64 != 0 || p.OH < 16) return false;
    const long long items16 = (long long)p.N * ((p.OH + 15) / 16) * ((p.OW + 15) / 16) * (p.Coutp / 64);
    return items16 >= device_cus();
}

int launch_conv_s1w(const ConvParams& p, hipStream_t stream) {
    if (!conv_s2c32_supported(p)) return (int)hipErrorInvalidValue;
    if ((p.fmt == FMT_BF)) return launch_s2c32_t<1, 8, 4, false, true>(p, stream);
#if S2_TH16
    if (use_th16(p)) return launch_s2c32_t<1, 16, 4>(p, stream);
#endif
    if (p.Coutp % 64 == 0) return launch_s2c32_t<1, 8, 4>(p, stream);
    return launch_s2c32_t<1, 8, 2>(p, stream);
}

// ---- several independent stride-1 convolutions in one launch (conv_s2c32_jobs_kernel) ----------------------------
bool conv_jobs_supported(const ConvParams* ps, int n, int stride) {
    if (n < 2 || n > MAXJOBS || (stride != 1 && stride != 2)) return false;
    for (int j = 0; j < n; ++j) {
        const ConvParams& p = ps[j];
        if (p.fmt != ps[0].fmt || p.nheads > 1 || p.out_f32 || p.Coutp % ((p.fmt == FMT_BF) ? 64 : 32) != 0 || !conv_s2c32_supported(p)) return false;
        if (images_per_launch(p) < p.N || (stride == 1 && (p.H != p.OH || p.W != p.OW || use_th16(p)))) return false;
        if (stride == 2 && (p.OH != (p.H + 1) / 2 || p.OW != (p.W + 1) / 2 || p.res)) return false;
    }
    return true;
}

template <int S, int TH, bool BF>
static int launch_jobs_t(const ConvParams* ps, int n, hipStream_t stream) {
    using S2C = ConvCfg<3, S, TH, 2>;
    StreamJobs jobs{};
    const int slots = 2 * device_cus();
    struct J { int idx, grid; long long steps; };
    J order[MAXJOBS];
    StreamGeo geos[MAXJOBS];
    int mws[MAXJOBS];
    for (int j = 0; j < n; ++j) {
        const ConvParams& p = ps[j];
        StreamGeo& geo = geos[j];
        geo.tiles_x = (p.OW + TW - 1) / TW;
        geo.tiles_y = (p.OH + TH - 1) / TH;
        const int mw = p.Coutp % 64 == 0 ? 4 : 2;
        mws[j] = mw;
        geo.ctiles = p.Coutp / (16 * mw);
        const long long nitems = (long long)p.N * geo.tiles_y * geo.tiles_x * geo.ctiles;
        if (nitems <= 0 || nitems > 0x7fffffffLL) return (int)hipErrorInvalidValue;
        geo.nitems = (int)nitems;
        geo.m_ct = magic_of(geo.ctiles);
        geo.m_tx = magic_of(geo.tiles_x);
        geo.m_ty = magic_of(geo.tiles_y);
        int grid = (int)(nitems < slots ? nitems : slots);
        if (grid > geo.ctiles) grid -= grid % geo.ctiles;
        order[j] = {j, grid, ((nitems + grid - 1) / grid) * (p.Cinp >> (BF ? 6 : 5))};
    }
    std::sort(order, order + n, [](const J& a, const J& b) { return a.steps > b.steps; });   // longest workgroups first
    jobs.njobs = n;
    int at = 0;
    for (int k = 0; k < n; ++k) {
        jobs.p[k] = ps[order[k].idx];
        jobs.geo[k] = geos[order[k].idx];
        jobs.mw[k] = mws[order[k].idx];
        jobs.start[k] = at;
        at += order[k].grid;
    }
    for (int k = n; k <= MAXJOBS; ++k) jobs.start[k] = at;
    auto kern = conv_s2c32_jobs_kernel<S, TH, BF>;
    if (const int e_ = ensure_dyn_lds(reinterpret_cast<const void*>(kern), S2C::XBYTES)) return e_;
    hipLaunchKernelGGL(kern, dim3((unsigned)at), dim3(NTHREADS), S2C::XBYTES, stream, jobs);
    return (int)hipGetLastError();
}

int launch_conv_jobs(const ConvParams* ps, int n, int stride, hipStream_t stream) {
    if (!conv_jobs_supported(ps, n, stride)) return (int)hipErrorInvalidValue;
    if (stride == 1) return (ps[0].fmt == FMT_BF) ? launch_jobs_t<1, 8, true>(ps, n, stream) : launch_jobs_t<1, 8, false>(ps, n, stream);
    return (ps[0].fmt == FMT_BF) ? launch_jobs_t<2, 4, true>(ps, n, stream) : launch_jobs_t<2, 4, false>(ps, n, stream);
}

}  // namespace esa
