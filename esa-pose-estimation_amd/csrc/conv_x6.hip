// conv_x6.hip — 3x3 (stride 1 / 2) and 1x1 convolution (+ folded-BN bias, optional residual, ReLU) in fp32-grade
// arithmetic on the bf16 matrix cores: the "bf16x6" mode (esahrnet_cfg.precision 2), tensors in plain f32 NHWC ("F32").
//
// Replaces (reference, cuDNN fp32 via ATen): every nn.Conv2d + BatchNorm2d + ReLU (+ residual add) of
// models/seg_hrnet.py except the stem conv1 and output_layer — BasicBlock :45-61, transitions :343-377, fuse layers
// :176-220, last_layer :313-329 — at the precision the reference computes in (BASELINE configs[1]: fp32).
//
// Arithmetic.  Every f32 operand v is split EXACTLY into three bf16 terms, v = v0 + v1 + v2 (round-to-nearest-even at
// each stage: 8 + 8 + 8 significand bits, fp32 exponent range), and a product a*b is evaluated as
//     a0*b0 + a0*b1 + a1*b0 + a0*b2 + a1*b1 + a2*b0            (six v_mfma_f32_16x16x32_bf16, f32 accumulation)
// — the dropped terms a1*b2 + a2*b1 + a2*b2 are below 2^-26 |a*b|.  Measured on MI355X against fp64
// (tools/ubench/x6_numerics.hip, K = 288 .. 4320): rms error 2.4e-8 of sum|a*b| for signed and 4.6e-7 for all-positive
// products; a sequential f32 fmaf chain (and v_mfma_f32_16x16x4_f32, which is bit-identical to it) has 2.8e-8 / 7.2e-7;
// all nine products buy nothing over six.  Ceiling: 2.5 PFLOP/s / 6 = 417 TFLOP/s algorithmic = 2.65 x the f32 peak.
// Weights are split on the host (pack_conv_weights_x6); activations stay f32 in HBM (4 bytes per channel, the bytes of
// the split-bf16 format) and are split by the staging threads on their way into LDS, once per workgroup and tile — the
// epilogue stores accumulators as they are.
//
// Tiling.  256 threads = 4 waves, ONE workgroup per CU (up to 512 VGPRs per lane): a wave owns one 16-cout MFMA tile x NR
// rows x 16 columns; every B fragment (16 pixels x 32 channels of one term) read from LDS feeds 6 MFMAs per tap and up
// to three taps; the wave's weights — 9 taps x 3 terms = 27 fragments of a 32-channel chunk — live in registers and are
// refilled kx-third by kx-third for the next step as soon as a phase is done.  The workgroup covers 16*CT couts x
// (4/CT)*NR rows (64 couts x 16 rows at stride 1); it walks a persistent stream of (item, chunk) steps with two tile
// buffers in LDS and ONE barrier per step: the tile of step s+1 is loaded, split and written while step s computes.
// A step is 864 MFMAs per wave at stride 1 (13.8 k cycles) against 3 barriers and 216 MFMAs in the split-bf16 stream
// kernel (conv_s2c32.hip).  (A wave with TWO cout tiles — twice the MFMAs per LDS read — needs 216 weight registers and
// spilled at 512; LDS reads are 2.6 k of a step's 13.8 k cycles as it is.)
//
// K order inside a 32-channel chunk.  A staging thread (pixel, sg) loads two 16-byte quads of the pixel's 128-byte
// chunk: channels 4sg .. 4sg+3 and 16+4sg .. 16+4sg+3 — four consecutive lanes read 64 contiguous bytes per load —
// so MFMA K index (k-group sg, element j) is channel  j < 4 ? 4sg + j : 16 + 4sg + (j - 4);  the packed weights use
// the same permutation (x6_chan_of_k).
#include <algorithm>
#include <cstdio>
#include <type_traits>

#include "conv_cfg.h"
#include "devstate.h"
#include "kernels.h"
#include "sb.h"

#ifdef X6_TRACE
// debug build only (tools/trace_x6.py, -DX6_TRACE=1): shader-clock stamps of the first workgroups' wave 0, 8 events x 32 steps
__device__ unsigned long long g_x6_trace[64 * 32 * 8];
extern "C" int esa_debug_x6_trace(void* dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_x6_trace), sizeof(g_x6_trace)); }
__device__ unsigned long long g_x6_wg[1024 * 4];      // per workgroup: wall clock (100 MHz) at start / end, steps, HW_ID
extern "C" int esa_debug_x6_wg(void* dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_x6_wg), sizeof(g_x6_wg)); }
#endif

// timing experiments only (tools/trace_x6.py): bit mask of parts compiled OUT — 1 tile global loads, 2 split + LDS writes,
// 4 weight reloads, 8 LDS operand reads, 16 residual loads + output stores, 32 barrier.  0 in the product build.
#ifndef X6_ABL
#define X6_ABL 0
#endif
#ifndef X6_KG_SKEW
#define X6_KG_SKEW 1
#endif
#ifndef X6_W2BUF
#define X6_W2BUF 1      // 3x3: two weight thirds in registers (72 VGPRs), the step loop unrolled by two
#endif
#ifndef X6_READ_PIN
#define X6_READ_PIN 1
#endif
#ifndef X6_PREF_DIST
#define X6_PREF_DIST 1      // (2: measured, no change — the reads cost issue cycles, not exposed latency)
#endif
#ifndef X6_S2_PREF
#define X6_S2_PREF 1
#endif
#ifndef X6_C1_MINCOUT
#define X6_C1_MINCOUT 256
#endif
#ifndef X6_C1_JOBS
#define X6_C1_JOBS 0       // merged launch of the register-resident 1x1 kernel for a module's fuse-up 1x1s: measured slower
#endif
#ifndef X6_S2_SINGLE
#define X6_S2_SINGLE 1    // stride-2 kernels (and the fused stem): 4-row tiles in ONE LDS buffer, two barriers per step
#endif
#ifndef X6_DEEP
#define X6_DEEP 0
#endif
#ifndef X6_PRIO_FLIP
#define X6_PRIO_FLIP 1
#endif

namespace esa {

__host__ __device__ constexpr int x6_chan_of_k(int sg, int j) { return j < 4 ? 4 * sg + j : 16 + 4 * sg + (j - 4); }

size_t packed_weight_bytes_x6(int coutp, int cinp, int k) { return (size_t)(coutp / 16) * (cinp / 32) * k * k * 3 * 1024; }

static inline uint16_t x6_host_bf16(float f) {
    uint32_t u;
    __builtin_memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);       // round to nearest even (finite inputs)
    return (uint16_t)(u >> 16);
}
static inline float x6_host_f32(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
}

// [cout16 tile][cin32 chunk][tap][term 0..2][lane 0..63][8 x bf16], lane l holding
// W[cout = tile*16 + (l&15)][cin = chunk*32 + x6_chan_of_k(l>>4, j)] split into three bf16 terms
void pack_conv_weights_x6(const float* w, int cout, int cin, int k, int coutp, int cinp, void* dst) {
    uint16_t* d = static_cast<uint16_t*>(dst);
    const int taps = k * k, nch = cinp / 32;
    for (int t16 = 0; t16 < coutp / 16; ++t16)
        for (int c = 0; c < nch; ++c)
            for (int tap = 0; tap < taps; ++tap)
                for (int l = 0; l < 64; ++l)
                    for (int j = 0; j < 8; ++j) {
                        const int co = t16 * 16 + (l & 15), ci = c * 32 + x6_chan_of_k(l >> 4, j);
                        float v = 0.f;
                        if (co < cout && ci < cin) v = w[((size_t)co * cin + ci) * taps + tap];
                        const size_t base = ((((size_t)t16 * nch + c) * taps + tap) * 3) * 512;
                        float r = v;
                        for (int t = 0; t < 3; ++t) {
                            const uint16_t h = x6_host_bf16(r);
                            d[base + t * 512 + l * 8 + j] = h;
                            r -= x6_host_f32(h);
                        }
                    }
}

namespace {

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
constexpr uint32_t X6_OOB = 0x80000000u;      // offset >= every descriptor's num_records (all < 2^31)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t x6_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

// (a, b) -> three dwords of packed bf16 (a in the low half): a = h + m + l exactly, likewise b
__device__ __forceinline__ void x6_split_pair(float a, float b, uint32_t& h, uint32_t& m, uint32_t& l) {
    const f32x2 v = {a, b};
    h = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
    const f32x2 r = {a - __uint_as_float(h << 16), b - __uint_as_float(h & 0xffff0000u)};
    m = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, bf16x2));
    const f32x2 s = {r[0] - __uint_as_float(m << 16), r[1] - __uint_as_float(m & 0xffff0000u)};
    l = __builtin_bit_cast(uint32_t, __builtin_convertvector(s, bf16x2));
}

// compile-time loop: f(std::integral_constant<int, I>{}) for I in [B, E)
template <int B, int E, class F>
__device__ __forceinline__ void x6_static_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        x6_static_for<B + 1, E>(f);
    }
}
// staging units of a phase's row r: the last XH rows take one unit each (phases shorter than XH rows several in row 0);
// unit u of the half goes to row max(0, NR - XH + u)
__host__ __device__ constexpr int x6_unit_row(int NR, int XH, int u) { return NR - XH + u > 0 ? NR - XH + u : 0; }
__host__ __device__ constexpr int x6_units_in_row(int NR, int XH, int nunits, int r) {
    int n = 0;
    for (int u = 0; u < nunits; ++u) n += x6_unit_row(NR, XH, u) == r ? 1 : 0;
    return n;
}
__host__ __device__ constexpr int x6_first_unit_in_row(int NR, int XH, int nunits, int r) {
    for (int u = 0; u < nunits; ++u)
        if (x6_unit_row(NR, XH, u) == r) return u;
    return 0;
}

// NBUF: tile buffers in LDS.  2: the tile of step s+1 is written while step s computes, ONE barrier per step.  1 (stride 2:
// the input tile is 4x the output tile, and 4-row tiles double-buffered leave room for one workgroup per CU only): the
// whole next tile is loaded into registers during the step and written between two barriers behind it — the write phase
// (split, or the fused stem's conv1: pure VALU) of one workgroup runs beside the MFMA phases of the CU's other one.
template <int KS, int S, int CT, int NR, int NW, int NBUF = 2>
struct X6Cfg {
    static constexpr int NT = NW * 64;                      // threads per workgroup (NW waves: 4 = one per SIMD, 8 = two)
    static constexpr int RG = NW / CT;                      // row groups
    static constexpr int TH = RG * NR;                      // output rows per workgroup tile
    static constexpr int PAD = (KS - 1) / 2;
    static constexpr int TAPS = KS * KS;
    static constexpr int IH = (TH - 1) * S + KS;
    static constexpr int IW = (TW - 1) * S + KS;
    static constexpr int NPIX = IH * IW;
    static constexpr int PLANE = ((NPIX * 16 + 128 + 255) / 256) * 256;
    static constexpr int XBYTES = 12 * PLANE;               // one tile buffer: 4 k-groups x 3 terms
    static constexpr int LDS = NBUF * XBYTES;
    static constexpr int XITER = (NPIX * 4 + NT - 1) / NT;
    static constexpr int ROWS = (NR - 1) * S + KS;          // input rows a wave touches
    // B-operand reads (ds_read_b128: lane groups pair k-groups {0,1} and {2,3}, one term per instruction) want the
    // planes of a k-group pair congruent mod 256 B at stride 1 and one 16-byte slot apart at stride 2 (conv_cfg.h)
    // Staging writes (a ds_write_b128 group = 2 pixels x the 4 k-groups): k-groups {0,1} share their banks by the read rule;
    // the pair {2,3} sits 64 bytes further so that a pixel's four writes are 2-way, not 4-way, conflicts (X6_KG_SKEW)
    __host__ __device__ static constexpr int kg_skew(int g) { return (S == 2 ? (g & 1) * 16 : 0) + (X6_KG_SKEW ? (g >> 1) * 64 : 0); }
    __host__ __device__ static constexpr int plane_off(int g, int t) { return (g * 3 + t) * PLANE + kg_skew(g); }
    static_assert(LDS <= 160 * 1024, "tile buffers exceed the CU's LDS");
};

// tile stream geometry; m_* = floor((2^32 - 1) / d): q = umulhi(b, m) is b / d or one less
struct X6Geo {
    int tiles_x, tiles_y, ctiles, nitems;
    uint32_t m_ct, m_tx, m_ty;
};
__device__ __forceinline__ int x6_div(int b, int d, uint32_t m) {
    int q = (int)__umulhi((uint32_t)b, m);
    if (b - q * d >= d) ++q;
    return q;
}

struct X6Pos {            // one (item, chunk) step of the workgroup's stream
    int item, c, n, oy0, ox0, ct;
    int ok;                 // (an int: a trailing bool makes hipcc copy the struct's padding bytes through scratch memory)
};

// STEM (fused stem, launch_stem_fused_x6): the input tensor of this 3x3 stride-2 convolution — conv1 + bn1 + ReLU of the raw
// crop (models/seg_hrnet.py:426-428), 64 channels at full resolution: 537 MB per batch-32 step written and read back — is
// never materialised.  A staging unit loads the 3x3 neighbourhood of its pixel from the f32 NCHW crop (cin = 1) and forms
// its 8 channels on the f32 VALU (bias first, then the nine taps in order: bit for bit what stem_kernel computes), ReLU,
// then splits them like any other unit.  Everything behind the staging is the ordinary kernel.
struct X6StemSrc {
    const float* x0;      // f32 [N][1][H][W]
    const float* w1;      // f32 [2 chunks][4 k-groups][9 taps + bias][8]: channel chunk*32 + x6_chan_of_k(sg, j)
};
template <int KS, int S, int CT, int NR, int NW, bool STEM = false, int NBUF = 2>
__device__ __forceinline__ void x6_body(const ConvParams& p, const X6Geo& geo, const int bid, const int G, const X6StemSrc stem = X6StemSrc{}) {
    using C = X6Cfg<KS, S, CT, NR, NW, NBUF>;
    constexpr int QSTEP = C::NT / 4;            // tile pixels staged per iteration
    constexpr int TAPS = C::TAPS, ROWS = C::ROWS, XITER = C::XITER;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, px = lane & 15;
    const int cw = wave % CT, rg = wave / CT;
    const int nchunks = p.Cinp >> 5;
    const int pixb = p.Cinp * 4, opix = p.Coutp * 4;
    const uint32_t ximg = (uint32_t)p.H * p.W * pixb, yimg = (uint32_t)p.OH * p.OW * opix;     // < 2^31, host-checked

    X6Pos cur;
    cur.item = xcd_contiguous(bid, G);
    if (cur.item >= geo.nitems) return;
    auto decode = [&](X6Pos& q) {
        const int q1 = x6_div(q.item, geo.ctiles, geo.m_ct);
        q.ct = q.item - q1 * geo.ctiles;
        const int q2 = x6_div(q1, geo.tiles_x, geo.m_tx);
        const int tx = q1 - q2 * geo.tiles_x;
        q.n = x6_div(q2, geo.tiles_y, geo.m_ty);
        const int ty = q2 - q.n * geo.tiles_y;
        q.oy0 = ty * C::TH;
        q.ox0 = tx * TW;
    };
    auto advance = [&](const X6Pos& a) {
        X6Pos q = a;
        if (q.c + 1 < nchunks) {
            ++q.c;
        } else {
            q.c = 0;
            q.item += G;
            q.ok = a.ok && q.item < geo.nitems;
            if (q.ok) decode(q);
        }
        return q;
    };

    float* const w1s = reinterpret_cast<float*>(smem + C::LDS);     // STEM: conv1's folded weights + bias, [chunk][sg][10][8]
    // STEM: the raw crop under a tile of conv1's output (tile + one pixel each side, zeros beyond the image: conv1's padding),
    // two buffers: the crop of step s+2 is loaded at the start of step s (one or two loads per thread), written here at its
    // end and consumed by the staging of step s+1 — every conv1 input then comes from LDS with a compile-time offset (read
    // per unit from global memory, the nine bounds-checked addresses per unit cost more VALU than conv1 itself).
    constexpr int RW = C::IW + 2, RH = C::IH + 2, RAWN = RW * RH, RIT = STEM ? (RAWN + C::NT - 1) / C::NT : 1;
    float* const rawt = w1s + 2 * 4 * 10 * 8;
    if (STEM) {
        for (int i = tid; i < 2 * 4 * 10 * 8; i += C::NT) w1s[i] = stem.w1[i];      // (published by the prologue's barrier)
    }
    // ---- staging map: thread -> (k-group sg, tile pixel q0 + QSTEP*it), fixed for the launch ----
    // (measured: 8 consecutive lanes = 8 pixels of ONE k-group makes the ds_write_b128 groups conflict-free but the global loads of
    // a lane quad touch four lines: 1 % fewer cycles per step in the trace, nothing in the network)
    const int sg = tid & 3, q0 = tid >> 2;
    int qyx[XITER];                             // tile-local (row << 8 | column), -1 beyond the tile
#pragma unroll
    for (int it = 0; it < XITER; ++it) {
        const int q = q0 + it * QSTEP;
        const int qy = q / C::IW, qx = q - qy * C::IW;
        qyx[it] = q < C::NPIX ? (qy << 8 | qx) : -1;
    }
    int roff[STEM ? XITER : 1];                 // STEM: the unit's pixel in the raw crop tile (its 3x3 window's first element)
    if constexpr (STEM) {
#pragma unroll
        for (int it = 0; it < XITER; ++it) roff[it] = qyx[it] >= 0 ? (qyx[it] >> 8) * RW + (qyx[it] & 255) : 0;
    }
    // The tile of step s+1 is staged INSIDE step s in two halves (units [0, XH) and [XH, XITER)): loads at the start of a
    // phase, split + LDS writes behind it — a unit lives in registers for one phase, not for a whole step.
    // DEEP (stride 2 and 1x1: steps of 24-108 MFMAs per wave, shorter than a trip to HBM): the WHOLE tile of step s+2 is
    // loaded at the start of step s into a second register set (xn) and handed to xr at the start of step s+1 — two steps of
    // cover instead of one phase.  Measured (X6_DEEP=1): no change — the stride-2 steps (36 MFMAs per phase and wave) are bound
    // by what surrounds the MFMAs (item decode, address generation, 27 weight loads, barrier: 5 k of a step's 7 k cycles), not
    // by load latency.  Off by default.
    constexpr bool DEEP = !STEM && NBUF == 2 && (S == 2 || KS == 1) && X6_DEEP;
    constexpr int XH = (KS == 1 || DEEP || NBUF == 1) ? XITER : (XITER + 1) / 2;
    u32x4 xr[STEM ? 1 : XH][2];                 // the half in flight: two quads of 4 channels per unit
    u32x4 xn[DEEP ? XITER : 1][2];              // DEEP: the tile two steps ahead
    float rawreg[RIT];                          // STEM: this thread's pixels of the raw crop two steps ahead
    // byte offsets of the staged item's units inside its image (X6_OOB beyond the image: the zero padding); they change with
    // the item, not with the chunk: recomputed by tile_offsets() when the stream moves to a new item
    uint32_t xoff[STEM ? 1 : XITER];
    auto tile_offsets = [&](const X6Pos& q) __attribute__((always_inline)) {
        if constexpr (!STEM) {
            const int gy0 = q.oy0 * S - C::PAD, gx0 = q.ox0 * S - C::PAD;
#pragma unroll
            for (int it = 0; it < XITER; ++it) {
                const int gy = gy0 + (qyx[it] >> 8), gx = gx0 + (qyx[it] & 255);
                const bool inside = qyx[it] >= 0 && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
                xoff[it] = inside ? (uint32_t)((gy * p.W + gx) * pixb + sg * 16) : X6_OOB;
            }
        }
    };
    auto load_tile = [&](const X6Pos& q, auto half_c) __attribute__((always_inline)) {
        constexpr int i0 = decltype(half_c)::value * XH;
        if constexpr (STEM) return;       // (raw_load / stage_stem below)
        const __amdgpu_buffer_rsrc_t rx = x6_rsrc(p.x + (size_t)q.n * ximg, q.ok ? ximg : 0u);      // (no step behind: zeros, no traffic)
        const int so = q.c * 128;
#pragma unroll
        for (int it = i0; it < i0 + XH && it < XITER; ++it) {
            const uint32_t off = xoff[STEM ? 0 : it];
            if (!(X6_ABL & 1)) {
                xr[it - i0][0] = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)off, so, 0);
                xr[it - i0][1] = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)off, so + 64, 0);
            }
        }
    };
    auto load_next = [&](const X6Pos& q) __attribute__((always_inline)) {       // DEEP: all units of q's tile -> xn
        const __amdgpu_buffer_rsrc_t rx = x6_rsrc(p.x + (size_t)q.n * ximg, q.ok ? ximg : 0u);
        const int gy0 = q.oy0 * S - C::PAD, gx0 = q.ox0 * S - C::PAD;
        const int so = q.c * 128;
#pragma unroll
        for (int it = 0; it < (DEEP ? XITER : 0); ++it) {
            const int gy = gy0 + (qyx[it] >> 8), gx = gx0 + (qyx[it] & 255);
            const bool inside = qyx[it] >= 0 && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
            const uint32_t off = inside ? (uint32_t)((gy * p.W + gx) * pixb + sg * 16) : X6_OOB;
            xn[it][0] = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)off, so, 0);
            xn[it][1] = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)off, so + 64, 0);
        }
    };
    char* const xwr = smem + q0 * 16 + sg * (3 * C::PLANE) + (S == 2 ? (sg & 1) * 16 : 0) + (X6_KG_SKEW ? (sg >> 1) * 64 : 0);      // plane_off(sg, 0) + pixel slot
    // unit `it` of the half in flight: split into three exact bf16 terms, three 16-byte LDS writes
    // split into three exact bf16 terms + three 16-byte LDS writes of one unit's 8 channels
    auto emit_unit = [&](int buf, int it, const u32x4 (&src)[2]) __attribute__((always_inline)) {
        u32x4 t0, t1, t2;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                uint32_t a, b, c;
                x6_split_pair(__uint_as_float(src[h][2 * k]), __uint_as_float(src[h][2 * k + 1]), a, b, c);
                t0[2 * h + k] = a; t1[2 * h + k] = b; t2[2 * h + k] = c;
            }
        char* o = xwr + buf * C::XBYTES + it * (QSTEP * 16);
        if ((it + 1) * QSTEP <= C::NPIX || q0 + it * QSTEP < C::NPIX) {       // (only the tile's last unit is partial)
            *reinterpret_cast<u32x4*>(o) = t0;
            *reinterpret_cast<u32x4*>(o + C::PLANE) = t1;
            *reinterpret_cast<u32x4*>(o + 2 * C::PLANE) = t2;
        }
    };
    // STEM: the raw crop of position q -> registers (zeros beyond the image), registers -> LDS
    auto raw_load = [&](const X6Pos& q) __attribute__((always_inline)) {
        if constexpr (STEM) {
            const uint32_t rimg = (uint32_t)p.H * p.W * 4;
            const __amdgpu_buffer_rsrc_t rx = x6_rsrc(stem.x0 + (size_t)q.n * p.H * p.W, q.ok ? rimg : 0u);
            const int gy0 = q.oy0 * S - C::PAD - 1, gx0 = q.ox0 * S - C::PAD - 1;
#pragma unroll
            for (int k = 0; k < RIT; ++k) {
                const int i = tid + k * C::NT;
                const int ry = i / RW, rx_ = i - ry * RW;
                const int yy = gy0 + ry, xx = gx0 + rx_;
                const bool in = i < RAWN && (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;
                rawreg[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, in ? (yy * p.W + xx) * 4 : (int)X6_OOB, 0, 0));
            }
        }
    };
    auto raw_commit = [&](int rb) __attribute__((always_inline)) {
        if constexpr (STEM) {
#pragma unroll
            for (int k = 0; k < RIT; ++k)
                if (RAWN % C::NT == 0 || k + 1 < RIT || tid + k * C::NT < RAWN) rawt[rb * RAWN + tid + k * C::NT] = rawreg[k];
        }
    };
    // STEM: conv1 + bn1 + ReLU of ALL units of position q's tile (channels q.c*32 + x6_chan_of_k(sg, 0..7)) from raw crop
    // buffer rb -> tile buffer buf.  Taps outermost: a tap's 8 weights (LDS) serve every unit.
    auto stage_stem = [&](int buf, const X6Pos& q, int rb) __attribute__((always_inline)) {
        if constexpr (STEM) {
            const float* wt = w1s + (q.c * 4 + sg) * 80;
            const float* rt = rawt + rb * RAWN;
            const int gy0 = q.oy0 * S - C::PAD, gx0 = q.ox0 * S - C::PAD;
            constexpr int UG = XITER <= 3 ? 3 : 2;       // units per pass (their 8-channel accumulators: 8 registers each)
            x6_static_for<0, (XITER + UG - 1) / UG>([&](auto g_c) __attribute__((always_inline)) {
                constexpr int u0 = decltype(g_c)::value * UG, un = XITER - u0 < UG ? XITER - u0 : UG;
                f32x4 a0[un], a1[un];
                {
                    const f32x4 b0 = *reinterpret_cast<const f32x4*>(wt + 72), b1 = *reinterpret_cast<const f32x4*>(wt + 76);
#pragma unroll
                    for (int u = 0; u < un; ++u) { a0[u] = b0; a1[u] = b1; }
                }
                x6_static_for<0, 9>([&](auto t_c) __attribute__((always_inline)) {
                    constexpr int t = decltype(t_c)::value;
                    const f32x4 w0 = *reinterpret_cast<const f32x4*>(wt + t * 8), w1v = *reinterpret_cast<const f32x4*>(wt + t * 8 + 4);
                    x6_static_for<0, un>([&](auto u_c) __attribute__((always_inline)) {
                        constexpr int u = decltype(u_c)::value;
                        const float v = rt[roff[u0 + u] + (t / 3) * RW + t % 3];
                        a0[u] = __builtin_elementwise_fma(f32x4{v, v, v, v}, w0, a0[u]);
                        a1[u] = __builtin_elementwise_fma(f32x4{v, v, v, v}, w1v, a1[u]);
                    });
                });
                x6_static_for<0, un>([&](auto u_c) __attribute__((always_inline)) {
                    constexpr int u = decltype(u_c)::value, it = u0 + u;
                    // outside conv1's output: conv2's zero padding, not conv1 of the padded crop
                    const int gy = gy0 + (qyx[it] >> 8), gx = gx0 + (qyx[it] & 255);
                    const uint32_t m = (qyx[it] >= 0 && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W) ? 0xffffffffu : 0u;
                    u32x4 src[2];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        src[0][k] = __float_as_uint(relu1(a0[u][k])) & m;
                        src[1][k] = __float_as_uint(relu1(a1[u][k])) & m;
                    }
                    emit_unit(buf, it, src);
                });
                __builtin_amdgcn_sched_barrier(0);
            });
        }
    };
    auto write_unit = [&](int buf, int i0, int it) __attribute__((always_inline)) {
        {
            if (it >= XITER || (X6_ABL & 2)) return;
            u32x4 t0, t1, t2;
            u32x4 src[2];
            if constexpr (STEM) {
                src[0] = u32x4{0u, 0u, 0u, 0u}; src[1] = src[0];      // (never called: stage_stem)
            } else {
                src[0] = xr[it - i0][0];
                src[1] = xr[it - i0][1];
            }
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    uint32_t a, b, c;
                    x6_split_pair(__uint_as_float(src[h][2 * k]), __uint_as_float(src[h][2 * k + 1]), a, b, c);
                    t0[2 * h + k] = a; t1[2 * h + k] = b; t2[2 * h + k] = c;
                }
            char* o = xwr + buf * C::XBYTES + it * (QSTEP * 16);
            if ((it + 1) * QSTEP <= C::NPIX || q0 + it * QSTEP < C::NPIX) {       // (only the tile's last unit is partial)
                *reinterpret_cast<u32x4*>(o) = t0;
                *reinterpret_cast<u32x4*>(o + C::PLANE) = t1;
                *reinterpret_cast<u32x4*>(o + 2 * C::PLANE) = t2;
            }
        }
    };

    // ---- weights in registers; third kx = taps ky*KS + kx (KS fragments x 3 terms) ----
    // W2 (3x3): TWO thirds live — phase i (three per step, counted across steps) multiplies from buffer i & 1 while the third
    // of phase i + 1 lands in the other one, which phase i - 1 has finished with: 72 weight registers instead of 108.  The
    // buffer of a phase is a compile-time index, so the step loop is unrolled by two (PAR = step parity).  1x1: wf[1][3].
    constexpr bool W2 = KS == 3 && X6_W2BUF;
    bf16x8 wf[W2 ? 2 * KS : TAPS][3];
    auto load_w = [&](int ct, int ch, int kx, int b) __attribute__((always_inline)) {     // b: buffer (W2 only)
        const uint4* ws = p.w + ((size_t)((ct * CT + cw) * nchunks + ch) * TAPS) * 3 * 64 + lane;
        if (X6_ABL & 4) return;
#pragma unroll
        for (int ky = 0; ky < KS; ++ky)
#pragma unroll
            for (int t = 0; t < 3; ++t)
                wf[W2 ? b * KS + ky : ky * KS + kx][t] = __builtin_bit_cast(bf16x8, ws[((ky * KS + kx) * 3 + t) * 64]);
    };
    auto load_bias = [&](int ct) { return *reinterpret_cast<const f32x4*>(p.bias + (ct * CT + cw) * 16 + g * 4); };

    // ---- prologue: tile of step 0 into buffer 0, tile of step 1 in flight, weights of step 0 ----
    cur.c = 0;
    cur.ok = true;
    decode(cur);
    constexpr auto H0 = std::integral_constant<int, 0>{};
    constexpr auto H1 = std::integral_constant<int, 1>{};
    auto write_tile = [&](int buf, auto half_c) __attribute__((always_inline)) {
        constexpr int i0 = decltype(half_c)::value * XH;
#pragma unroll
        for (int it = i0; it < i0 + XH; ++it) write_unit(buf, i0, it);
    };
    tile_offsets(cur);
    if constexpr (STEM) {
        raw_load(cur);
        raw_commit(0);
        __syncthreads();            // (also publishes w1s)
        stage_stem(0, cur, 0);
    } else {
        load_tile(cur, H0);
        write_tile(0, H0);
        if constexpr (XH < XITER) {
            load_tile(cur, H1);
            write_tile(0, H1);
        }
    }
    X6Pos nxt = advance(cur);
    if (nxt.c == 0) tile_offsets(nxt);
    if constexpr (STEM) {
        raw_load(nxt);
        raw_commit(1);              // (published by the loop's first barrier)
    }
    int sidx = 0;
    if constexpr (DEEP) load_next(nxt);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (W2) {
        load_w(cur.ct, 0, 0, 0);            // step 0's first third; the others are loaded one phase ahead inside the steps
    } else {
#pragma unroll
        for (int kx = 0; kx < KS; ++kx) load_w(cur.ct, 0, kx, 0);
    }
    f32x4 bv = load_bias(cur.ct);
    asm volatile("" : "+v"(bv));                // (waited for here, once: at the loop head hipcc would otherwise merge "bias pending
                                                // behind 27 weight loads" into every iteration's state and drain vmcnt(0) there)

    const char* const xrd0 = smem + ((rg * NR * S) * C::IW + px * S) * 16 + g * (3 * C::PLANE) + (S == 2 ? (g & 1) * 16 : 0) +
                             (X6_KG_SKEW ? (g >> 1) * 64 : 0);       // plane_off(g, 0) + pixel slot
    f32x4 acc[NR];
    int buf = 0;
    int pstep = 0;
#ifdef X6_TRACE
    int tstep = 0;
    constexpr bool TSEL = KS == 3 && (X6_TRACE == 3 ? STEM : (S == X6_TRACE && !STEM));      // (-DX6_TRACE=3: the fused stem)
    const bool ton = TSEL && bid < 64 && tid == 0;
    const bool wgon = TSEL && bid < 1024 && tid == 0;
    if (wgon) {
        g_x6_wg[bid * 4] = __builtin_amdgcn_s_memrealtime();
        g_x6_wg[bid * 4 + 3] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    }
#define X6_TR(EV) if (ton && tstep < 32) g_x6_trace[(bid * 32 + tstep) * 8 + (EV)] = __builtin_amdgcn_s_memtime();
#else
#define X6_TR(EV)
#endif
    auto step = [&](auto par_c) __attribute__((always_inline)) -> bool {      // true: that was this workgroup's last step
        constexpr int PAR = decltype(par_c)::value;
        constexpr int B0 = PAR, B1 = PAR ^ 1;       // W2: buffers of phases 0 / 2 and of phase 1
        // Two workgroups share a CU (X6_MODE 2): the hardware arbitrates the matrix pipe by priority, then AGE — left alone
        // the older workgroup runs at full speed, the younger one on the leftovers, and finishes its equal share of the items
        // alone on a half-empty CU.  The priority alternates per step between the grid's halves (workgroups b and b + G/2
        // usually share a CU), so both progress at the same average rate.
        if (X6_PRIO_FLIP) {
            if (((pstep++) ^ (bid * 2 >= G ? 1 : 0)) & 1) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
        }
        X6_TR(0)
#ifdef X6_TRACE
        if (ton && tstep < 32) g_x6_trace[(bid * 32 + tstep) * 8 + 7] = __builtin_amdgcn_s_memrealtime();      // 100 MHz wall clock
#endif
        if (!(X6_ABL & 32)) __syncthreads();        // tile of this step published; every wave is done reading the other buffer
        X6_TR(1)
        const bool last_chunk = cur.c + 1 == nchunks;
        X6Pos nn = nxt;
        if constexpr (STEM) {       // the raw crop two steps ahead
            nn = advance(nxt);
            raw_load(nn);
        }
        // Every load of the loop is UNCONDITIONAL (the next step's weight thirds even when they are the ones already held or
        // no step follows, the residual rows and the output stores of a step that does not end its item with out-of-range
        // offsets: an issue slot, no traffic): with a load behind a run-time branch hipcc cannot count what is in flight at
        // the join and drains vmcnt(0) there — measured, a phase then started 1-2 k cycles late behind the L2 latency of the
        // weight third issued just before it.
        const int wct = nxt.ok ? nxt.ct : cur.ct, wch = nxt.ok ? nxt.c : cur.c;
        if (cur.c == 0) {
#pragma unroll
            for (int t = 0; t < NR; ++t) acc[t] = bv;
        }
        // bias of the next step's cout slice: the OLDEST load of the step, so that the wait in front of the next item's
        // accumulator start leaves the weight thirds issued behind it in flight
        f32x4 bvn = load_bias(wct);
        __builtin_amdgcn_sched_barrier(0);
        // output addressing of this wave's rows; the residual rows (last chunk) are loaded a few rows ahead of their use
        const int co0 = (cur.ct * CT + cw) * 16 + g * 4;
        const int rox = cur.ox0 + px, roy = cur.oy0 + rg * NR;
        const uint32_t o0 = (rox < p.OW && last_chunk) ? (uint32_t)((roy * p.OW + rox) * opix + co0 * 4) : X6_OOB;
        const int orow = p.OW * opix;
        const int nrows = p.OH - roy;
        const bool do_res = p.res != nullptr;
        const __amdgpu_buffer_rsrc_t rr = x6_rsrc((do_res ? p.res : p.y) + (size_t)cur.n * yimg, do_res ? yimg : 0u);
        const __amdgpu_buffer_rsrc_t ry = x6_rsrc(p.y + (size_t)cur.n * yimg, yimg);
        const int rfl = relu_floor(p.relu);
        constexpr int RCN = NBUF == 1 ? 2 : NR < 6 ? NR : 6;       // residual rows in flight (a row's load is issued RCN - 1 rows before its
                                                                   // epilogue; the single-buffer stride-2 kernels are short of registers)
        u32x4 rc[RCN];
        // (unconditional: without a residual the descriptor is empty and the loads return zeros — a load behind a run-time
        // condition makes hipcc drain vmcnt(0) at the join)
        auto res_load = [&](int t) __attribute__((always_inline)) {
            const uint32_t ro = t < nrows ? o0 + (uint32_t)(t * orow) : X6_OOB;
            if (!(X6_ABL & 16)) rc[t % RCN] = __builtin_amdgcn_raw_buffer_load_b128(rr, (int)ro, 0, 0);
        };
        const char* const xrd = xrd0 + buf * C::XBYTES;
        // One kx phase.  Accumulation order (tools/ubench/x6_numerics.hip: 7x less rounding error than one long chain, better
        // than a 16-way blocked f32 sum on the CPU): the 3 taps x 6 products of an output row and cout tile are summed in a
        // FRESH accumulator — the fifteen low-order products first, while the sum is small, the three a0*b0 products last —
        // and added to the item's running sum by one VALU add, one row late (the chain's result is not waited for).
        // The B fragments of KS input rows stay in registers: every LDS read feeds the taps of up to three output rows.
        // `wh`: half of the next step's tile (in registers since the start of this phase) that is split and written to the
        // other LDS buffer during the LAST rows of this phase, one unit per row, in the issue shadow of the row's MFMAs
        // (-1: none).  LDS operand reads run one output row ahead of the MFMAs that consume them, pinned by a
        // sched_barrier: left alone, hipcc sinks each ds_read next to its first use and waits for it.
        auto phase = [&](auto kx_c, auto epi_c, auto wh_c, auto wb_c) __attribute__((always_inline)) {
            constexpr int kx = decltype(kx_c)::value;
            constexpr int WB = decltype(wb_c)::value;            // W2: weight buffer of this phase
            constexpr bool EPI = decltype(epi_c)::value;         // last phase of the item: rows leave as they complete
            constexpr int WH = decltype(wh_c)::value;
            // LDS operand reads run one output row ahead (RING = the input rows of output rows r and r + 1) — except in the
            // single-buffer stride-2 kernels, which are short of registers: there a row's new input rows are read at its start
            constexpr bool PREF = NBUF == 2 || X6_S2_PREF;      // (the single-buffer kernels: since the two-thirds weight registers)
            constexpr int PD = PREF ? (S == 1 && KS == 3 ? X6_PREF_DIST : 1) : 0;      // output rows the LDS reads run ahead
            constexpr int RING = KS + PD * S;
            constexpr int NSLOT = 6 * KS;                        // MFMAs of a row
            bf16x8 xw[RING][3];
            bf16x8 dsink[3] = {};         // (X6_ABL & 128 only)
            auto read_rows = [&](int lo, int hi) __attribute__((always_inline)) {
                if ((X6_ABL & 8) && lo > 0) return;
                if ((X6_ABL & 128) && lo > 0) {     // (timing experiment: the reads are issued, nothing waits for them)
#pragma unroll
                    for (int j = lo; j <= hi; ++j)
#pragma unroll
                        for (int t = 0; t < 3; ++t)
                            asm volatile("ds_read_b128 %0, %1" : "+v"(dsink[t]) : "v"((uint32_t)(uintptr_t)(xrd - smem) + (uint32_t)((j * C::IW + kx) * 16 + t * C::PLANE)));
                    return;
                }
#pragma unroll
                for (int j = lo; j <= hi; ++j)
#pragma unroll
                    for (int t = 0; t < 3; ++t)
                        xw[j % RING][t] = *reinterpret_cast<const bf16x8*>(xrd + (j * C::IW + kx) * 16 + t * C::PLANE);
            };
            read_rows(0, KS - 1 + (PD > 1 ? (PD - 1) * S : 0));
            if (EPI) {
#pragma unroll
                for (int t = 0; t < RCN - 1 && t < NR; ++t) res_load(t);
            }
            f32x4 sm = {0.f, 0.f, 0.f, 0.f}, tprev = {0.f, 0.f, 0.f, 0.f};
            constexpr int NUNITS = WH < 0 ? 0 : (XITER - WH * XH < XH ? XITER - WH * XH : XH);      // units of this phase's half
            x6_static_for<0, NR + 1>([&](auto r_c) __attribute__((always_inline)) {
                constexpr int r = decltype(r_c)::value;
                // One scheduling region per output row: its MFMAs, the LDS reads of the rows that output row r + 1 adds to the
                // window, the running-sum add and epilogue of row r - 1, the split + LDS writes of this row's staging units.
                // (The wave issues in order and hipcc puts everything else behind the row's MFMAs; what fills the matrix pipe
                // meanwhile is the SIMD's other wave.)
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (PREF && r + PD < NR) {
                    read_rows((r + PD - 1) * S + KS, (r + PD) * S + KS - 1);
                    if (X6_READ_PIN) __builtin_amdgcn_sched_barrier(0);     // (else hipcc places the reads BEHIND this row's MFMAs)
                }
                if constexpr (!PREF && r > 0 && r < NR) read_rows((r - 1) * S + KS, r * S + KS - 1);
                if constexpr (r < NR) {
                    sm = f32x4{0.f, 0.f, 0.f, 0.f};
                    x6_static_for<0, NSLOT>([&](auto i_c) __attribute__((always_inline)) {
                        constexpr int i = decltype(i_c)::value;
                        // chain order: the five low-order products of every tap first, the KS a0*b0 products last
                        constexpr int ky = i < 5 * KS ? i / 5 : i - 5 * KS, pr = i < 5 * KS ? i % 5 : 5;
                        constexpr int tp = ky * KS + kx, w = (r * S + ky) % RING;
                        constexpr int wa = pr == 0 ? 2 : (pr == 2 || pr == 3) ? 1 : 0, xb = pr == 1 ? 2 : (pr == 2 || pr == 4) ? 1 : 0;
                        sm = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[W2 ? WB * KS + ky : tp][wa], xw[w][xb], sm, 0, 0, 0);
                    });
                }
                if constexpr (r > 0) {
                    acc[r - 1] += tprev;
                    // (pinned: LLVM otherwise sinks these adds to the end of the step — sixteen fresh sums per phase parked in
                    // registers, then bursts of adds)
                    asm volatile("" : "+v"(acc[r - 1]));
                    if constexpr (EPI) {          // row r - 1 leaves
                        u32x4 eo = {0u, 0u, 0u, 0u};
                        if (last_chunk) {         // (a uniform branch around VALU only: the store below stays unconditional — its
                                                  // offset is out of range when the step does not end its item)
#pragma unroll
                            for (int k = 0; k < 4; ++k) eo[k] = __float_as_uint(relu_opt(acc[r - 1][k] + __uint_as_float(rc[(r - 1) % RCN][k]), rfl));
                        }
                        const uint32_t so = (r - 1) < nrows ? o0 + (uint32_t)((r - 1) * orow) : X6_OOB;
                        // (image base in the descriptor, scalar offset the constant 0: see the store-data hazard note in
                        // conv_s2c32.hip — hipcc pads 16-byte buffer stores only when they carry no SGPR soffset)
                        if (!(X6_ABL & 16)) __builtin_amdgcn_raw_buffer_store_b128(eo, ry, (int)so, 0, 0);
                        else asm volatile("" :: "v"(eo));
                    }
                }
                if constexpr (EPI && r + RCN - 1 < NR) res_load(r + RCN - 1);      // into the slot the epilogue above has just read
                // staging units of this row: the last XH rows take one each (short phases several)
                if constexpr (!STEM && r < NR) {
                    constexpr int ucount = x6_units_in_row(NR, XH, NUNITS, r), ufirst = x6_first_unit_in_row(NR, XH, NUNITS, r);
                    x6_static_for<0, ucount>([&](auto u_c) __attribute__((always_inline)) {
                        write_unit(buf ^ 1, WH * XH, WH * XH + ufirst + decltype(u_c)::value);
                    });
                }
                tprev = sm;
            });
            if (X6_ABL & 128) asm volatile("s_waitcnt lgkmcnt(0)" :: "v"(dsink[0]), "v"(dsink[1]), "v"(dsink[2]));
            __builtin_amdgcn_sched_barrier(0);
        };
        constexpr auto F = std::false_type{};
        constexpr auto T = std::true_type{};
        constexpr auto W0 = std::integral_constant<int, 0>{};
        constexpr auto W1 = std::integral_constant<int, XH < XITER ? 1 : -1>{};
        constexpr auto WN = std::integral_constant<int, -1>{};
        constexpr auto IB0 = std::integral_constant<int, B0>{};
        constexpr auto IB1 = std::integral_constant<int, B1>{};
        if constexpr (KS == 3 && NBUF == 1) {
            // single tile buffer: the whole next tile -> registers now, -> LDS between the two barriers behind the phases
            load_tile(nxt, H0);
            if constexpr (W2) load_w(cur.ct, cur.c, 1, B1);       // this step's second third (the buffer phase 2 of the step before has left)
            __builtin_amdgcn_sched_barrier(0);
            X6_TR(2)
            phase(std::integral_constant<int, 0>{}, F, WN, IB0);
            asm volatile("" : "+v"(bvn));
            X6_TR(3)
            if constexpr (W2) load_w(cur.ct, cur.c, 2, B0);       // this step's last third
            else load_w(wct, wch, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            phase(std::integral_constant<int, 1>{}, F, WN, IB1);
            X6_TR(4)
            if constexpr (W2) load_w(wct, wch, 0, B1);            // the next step's first third
            else load_w(wct, wch, 1, 0);
            __builtin_amdgcn_sched_barrier(0);
            phase(std::integral_constant<int, 2>{}, T, WN, IB0);
            X6_TR(5)
            if constexpr (!W2) load_w(wct, wch, 2, 0);
            __syncthreads();            // every wave is done reading the tile
            if constexpr (STEM) stage_stem(0, nxt, (sidx + 1) & 1);
            else write_tile(0, H0);     // (published by the barrier at the top of the next step)
        } else if constexpr (KS == 3) {
            // The tile of step s+1 goes to the other buffer (free since the barrier) half by half: loads at the start of phases
            // 0 / 1, split + LDS writes in those phases' last rows.  The weight third a phase has used is refilled for step
            // s+1 right behind it.  (The sched_barriers keep hipcc from hoisting those loads into the phase before: the
            // registers they fill are the ones that phase is still reading.)
            if constexpr (DEEP) {
#pragma unroll
                for (int it = 0; it < XITER; ++it) { xr[it][0] = xn[it][0]; xr[it][1] = xn[it][1]; }
                load_next(advance(nxt));
            } else {
                load_tile(nxt, H0);
            }
            if constexpr (W2) load_w(cur.ct, cur.c, 1, B1);       // this step's second third (the buffer phase 2 of the step before has left)
            __builtin_amdgcn_sched_barrier(0);
            X6_TR(2)
            phase(std::integral_constant<int, 0>{}, F, W0, IB0);
            asm volatile("" : "+v"(bvn));       // the bias load is complete here (older than the tile half just consumed): waited
                                                // for now, with a counted vmcnt, not at the next step's start behind the weight loads
            X6_TR(3)
            if constexpr (STEM) {       // conv1 of the next tile, between two phases (inside one the operand ring leaves no registers)
                stage_stem(buf ^ 1, nxt, (sidx + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (XH < XITER) load_tile(nxt, H1);
            if constexpr (W2) load_w(cur.ct, cur.c, 2, B0);       // this step's last third
            else load_w(wct, wch, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            phase(std::integral_constant<int, 1>{}, F, W1, IB1);
            X6_TR(4)
            if constexpr (W2) load_w(wct, wch, 0, B1);            // the next step's first third
            else load_w(wct, wch, 1, 0);
            __builtin_amdgcn_sched_barrier(0);
            phase(std::integral_constant<int, 2>{}, T, WN, IB0);
            X6_TR(5)
            if constexpr (!W2) load_w(wct, wch, 2, 0);
        } else {
            if constexpr (DEEP) {
#pragma unroll
                for (int it = 0; it < XITER; ++it) { xr[it][0] = xn[it][0]; xr[it][1] = xn[it][1]; }
                load_next(advance(nxt));
            } else {
                load_tile(nxt, H0);
            }
            __builtin_amdgcn_sched_barrier(0);
            phase(std::integral_constant<int, 0>{}, T, W0, IB0);
            asm volatile("" : "+v"(bvn));
            load_w(wct, wch, 0, 0);
        }
        bv = bvn;
        X6_TR(6)
#ifdef X6_TRACE
        ++tstep;
#endif
#ifdef X6_TRACE
        if (wgon && !nxt.ok) { g_x6_wg[bid * 4 + 1] = __builtin_amdgcn_s_memrealtime(); g_x6_wg[bid * 4 + 2] = tstep; }
#endif
        if constexpr (STEM) raw_commit(sidx & 1);       // crop of step s+2 (this buffer's readers were step s-1's staging)
        ++sidx;
        if (!nxt.ok) return true;
        cur = nxt;
        if constexpr (STEM) {
            nxt = nn;
        } else {
            nxt = advance(cur);
            if (nxt.c == 0) tile_offsets(nxt);      // (a uniform branch around integer VALU only)
        }
        if constexpr (NBUF == 2) buf ^= 1;
        return false;
    };
    if constexpr (W2) {
        while (true) {
            if (step(std::integral_constant<int, 0>{})) break;
            if (step(std::integral_constant<int, 1>{})) break;
        }
    } else {
        while (!step(std::integral_constant<int, 0>{})) {}
    }
}

// OCC: workgroups per CU the register / LDS budget is cut for
template <int KS, int S, int CT, int NR, int NW, int OCC, int NBUF>
__global__ __launch_bounds__(NW * 64, OCC * NW / 4) void conv_x6_kernel(ConvParams p, X6Geo geo) {
    static_assert(OCC * X6Cfg<KS, S, CT, NR, NW, NBUF>::LDS <= 160 * 1024, "tile buffers of OCC workgroups exceed the CU's LDS");
    x6_body<KS, S, CT, NR, NW, false, NBUF>(p, geo, (int)blockIdx.x, (int)gridDim.x);
}

// fused stem: conv1 (VALU, inside the staging) -> conv2 3x3 stride 2 (models/seg_hrnet.py:426-431), cin = 1
constexpr int X6_S2_NBUF = X6_S2_SINGLE ? 1 : 2;        // stride 2: 4-row tiles in one LDS buffer / 2-row tiles in two
constexpr int X6_S2_NR4 = X6_S2_SINGLE ? 4 : 2, X6_S2_NR2 = X6_S2_SINGLE ? 2 : 1;
// the stem keeps 2-row tiles in two buffers: its staging holds 9 raw pixels per unit and the 4-row tile's 5 units spill
constexpr int X6_STEM_NR = 4, X6_STEM_NBUF = 1;
__global__ __launch_bounds__(NTHREADS, 2) void stem_x6_kernel(ConvParams p, X6Geo geo, X6StemSrc stem) {
    x6_body<3, 2, 4, X6_STEM_NR, 4, true, X6_STEM_NBUF>(p, geo, (int)blockIdx.x, (int)gridDim.x, stem);
}

// Several INDEPENDENT convolutions of one kind (the same-depth 3x3s of an HRModule's branches, the same-depth links of its
// fuse-down chains, its fuse-up 1x1s: models/seg_hrnet.py:143-220 — none reads another's output) in one launch.
// Workgroups [start[j], start[j+1]) run convolution j exactly as its own launch would — same items per workgroup, same
// order, same bits — but the launch gap is paid once and one convolution's tail overlaps the next one's start (at batch 32
// a branch convolution is only 4 steps per workgroup).  Longest workgroups first.  Both tilings (64 and 32 couts per
// workgroup) live in the kernel; a workgroup takes the one its convolution needs.  Two workgroups per CU (X6_MODE 2).
constexpr int X6_MAXJOBS = 6;
struct X6Jobs {
    ConvParams p[X6_MAXJOBS];
    X6Geo geo[X6_MAXJOBS];
    int start[X6_MAXJOBS + 1];
    int ct[X6_MAXJOBS];
    int njobs;
};
template <int KS, int S>
struct X6JobCfg {
    static constexpr int NR4 = S == 2 ? X6_S2_NR4 : 8, NR2 = S == 2 ? X6_S2_NR2 : 4;      // rows per wave of the 64- / 32-cout tiling (X6_MODE 2)
    static constexpr int NBUF = S == 2 ? X6_S2_NBUF : 2;
    using C4 = X6Cfg<KS, S, 4, NR4, 4, NBUF>;
    using C2 = X6Cfg<KS, S, 2, NR2, 4, NBUF>;
    static constexpr int LDS = C4::LDS > C2::LDS ? C4::LDS : C2::LDS;
};
template <int KS, int S>
__global__ __launch_bounds__(NTHREADS, 2) void conv_x6_jobs_kernel(X6Jobs jobs) {
    const int b = (int)blockIdx.x;
    int j = 0;
#pragma unroll
    for (int k = 1; k < X6_MAXJOBS; ++k) j += (k < jobs.njobs && b >= jobs.start[k]) ? 1 : 0;
    const int bid = b - jobs.start[j], G = jobs.start[j + 1] - jobs.start[j];
    if (jobs.ct[j] == 4) x6_body<KS, S, 4, X6JobCfg<KS, S>::NR4, 4, false, X6JobCfg<KS, S>::NBUF>(jobs.p[j], jobs.geo[j], bid, G);
    else x6_body<KS, S, 2, X6JobCfg<KS, S>::NR2, 4, false, X6JobCfg<KS, S>::NBUF>(jobs.p[j], jobs.geo[j], bid, G);
}

uint32_t x6_magic(int d) { return (uint32_t)(0xffffffffull / (uint64_t)d); }

template <int KS, int S>
int launch_x6_jobs_t(const ConvParams* ps, int n, hipStream_t stream) {
    using J = X6JobCfg<KS, S>;
    X6Jobs jobs{};
    const int slots = 2 * device_cus();
    struct Ord { int idx, grid; long long steps; };
    Ord order[X6_MAXJOBS];
    X6Geo geos[X6_MAXJOBS];
    int cts[X6_MAXJOBS];
    for (int j = 0; j < n; ++j) {
        const ConvParams& p = ps[j];
        X6Geo& geo = geos[j];
        const int ct = p.Coutp % 64 == 0 ? 4 : 2;
        const int th = ct == 4 ? J::C4::TH : J::C2::TH;
        cts[j] = ct;
        geo.tiles_x = (p.OW + TW - 1) / TW;
        geo.tiles_y = (p.OH + th - 1) / th;
        geo.ctiles = p.Coutp / (16 * ct);
        const long long nitems = (long long)p.N * geo.tiles_y * geo.tiles_x * geo.ctiles;
        if (nitems <= 0 || nitems > 0x7fffffffLL) return (int)hipErrorInvalidValue;
        geo.nitems = (int)nitems;
        geo.m_ct = x6_magic(geo.ctiles);
        geo.m_tx = x6_magic(geo.tiles_x);
        geo.m_ty = x6_magic(geo.tiles_y);
        int grid = (int)(nitems < slots ? nitems : slots);
        if (grid > geo.ctiles) grid -= grid % geo.ctiles;
        order[j] = {j, grid, ((nitems + grid - 1) / grid) * (p.Cinp >> 5)};
    }
    std::sort(order, order + n, [](const Ord& a, const Ord& b) { return a.steps > b.steps; });      // longest workgroups first
    jobs.njobs = n;
    int at = 0;
    for (int k = 0; k < n; ++k) {
        jobs.p[k] = ps[order[k].idx];
        jobs.geo[k] = geos[order[k].idx];
        jobs.ct[k] = cts[order[k].idx];
        jobs.start[k] = at;
        at += order[k].grid;
    }
    for (int k = n; k <= X6_MAXJOBS; ++k) jobs.start[k] = at;
    auto kern = conv_x6_jobs_kernel<KS, S>;
    if (const int e_ = ensure_dyn_lds(reinterpret_cast<const void*>(kern), J::LDS)) return e_;
    hipLaunchKernelGGL(kern, dim3((unsigned)at), dim3(NTHREADS), J::LDS, stream, jobs);
    return (int)hipGetLastError();
}

template <int KS, int S, int CT, int NR, int NW, int OCC, int NBUF>
int launch_x6_t(const ConvParams& p, hipStream_t stream) {
    using C = X6Cfg<KS, S, CT, NR, NW, NBUF>;
    X6Geo geo;
    geo.tiles_x = (p.OW + TW - 1) / TW;
    geo.tiles_y = (p.OH + C::TH - 1) / C::TH;
    geo.ctiles = p.Coutp / (16 * CT);
    const long long nitems = (long long)p.N * geo.tiles_y * geo.tiles_x * geo.ctiles;
    if (nitems <= 0 || nitems > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    geo.nitems = (int)nitems;
    geo.m_ct = x6_magic(geo.ctiles);
    geo.m_tx = x6_magic(geo.tiles_x);
    geo.m_ty = x6_magic(geo.tiles_y);
    const int cus = OCC * device_cus();
    int grid = (int)(nitems < cus ? nitems : cus);
    if (grid > geo.ctiles) grid -= grid % geo.ctiles;   // grid stride keeps the cout slice of a workgroup constant
    auto kern = conv_x6_kernel<KS, S, CT, NR, NW, OCC, NBUF>;
    if (const int e_ = ensure_dyn_lds(reinterpret_cast<const void*>(kern), C::LDS)) return e_;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(C::NT), C::LDS, stream, p, geo);
    return (int)hipGetLastError();
}

}  // namespace

// one IMAGE must be addressable with 31-bit byte offsets (per-image buffer descriptors, OOB marker 2^31); the batch
// size is unlimited (the image base is a 64-bit pointer)
bool conv_x6_supported(const ConvParams& p, int k, int stride) {
    if (p.fmt != FMT_F32 || p.out_f32 || p.nheads > 1) return false;
    if (!((k == 3 && (stride == 1 || stride == 2)) || (k == 1 && stride == 1))) return false;
    if ((p.Cinp & 31) || p.Cinp < 32 || (p.Coutp & 31) || p.Coutp < 32) return false;
    if (stride == 1 && (p.OH != p.H || p.OW != p.W)) return false;
    if (stride == 2 && (p.OH != (p.H + 1) / 2 || p.OW != (p.W + 1) / 2)) return false;
    return (long long)p.H * p.W * p.Cinp * 4 < 0x7fffffffLL && (long long)p.OH * p.OW * p.Coutp * 4 < 0x7fffffffLL;
}

// The tiling depends on the layer's shape only, never on the batch size: a crop's result is bit-identical in any batch.
//   couts % 64 == 0: 64 couts x 16 rows per workgroup (wave = cout tile x 16 rows); x 8 rows where an image is at most 16
//                    rows high (the 256-channel 16x16 branch: 2 tiles per image instead of 1, the chip has 256 CUs)
//   else           : 32 couts x 16 rows                (wave = cout tile x 8 rows)
// stride 2: 4 output rows per workgroup in every case (the input tile is 9 x 33 pixels).
static int x6_variant(const ConvParams& p, int k, int stride) {
    const int ct = p.Coutp % 64 == 0 ? 4 : 2;
    if (stride == 2) return ct == 4 ? 0 : 1;
    if (ct == 4) return p.OH <= 16 ? 2 : 3;
    return 4;
}

// X6_MODE — how a CU is filled (measured on MI355X, 64 -> 64 channels 64x64, batch 128, cycles per step of 864 MFMAs per
// SIMD; 13.8 k = matrix pipe never idle):
//   0: ONE workgroup of 4 waves (one per SIMD, 512 registers, 16-row tile)                      20.6 k
//   1: ONE workgroup of 8 waves (two per SIMD, 16-row tile)                                     19.0 k — the SIMD partners meet at
//      the SAME barrier every step: the older wave wins the matrix pipe, finishes its step and waits 4.2 k cycles for the
//      younger one, which then runs alone; with the barrier compiled out (wrong results) the same code takes 14.8 k
//   2: TWO workgroups of 4 waves (8-row tiles): SIMD partners belong to different workgroups and never wait for each other;
//      a workgroup's own four waves sit on four SIMDs, do the same work and reach their barrier together
#ifndef X6_MODE
#define X6_MODE 2
#endif
#if X6_MODE == 1
#define X6_KARGS(KS_, S_, CT_, NR1_) KS_, S_, CT_, (NR1_) / 2, 8, 1, 2
#elif X6_MODE == 2
#define X6_KARGS(KS_, S_, CT_, NR1_) KS_, S_, CT_, ((S_) == 2 ? (X6_S2_SINGLE ? (NR1_) : (NR1_) / 2) : ((NR1_) > 8 ? 8 : (NR1_) == 8 && (CT_) == 2 ? 4 : (NR1_))), 4, 2, ((S_) == 2 ? X6_S2_NBUF : 2)
#else
#define X6_KARGS(KS_, S_, CT_, NR1_) KS_, S_, CT_, NR1_, 4, 1, 2
#endif
int launch_conv_x6(const ConvParams& p, int k, int stride, hipStream_t stream) {
    if (!conv_x6_supported(p, k, stride)) return (int)hipErrorInvalidValue;
    if (k == 1 && conv1x1_x6_supported(p)) return launch_conv1x1_x6(p, stream);
    const int v = x6_variant(p, k, stride);
    if (k == 3) {
        switch (v) {
            case 0: return launch_x6_t<X6_KARGS(3, 2, 4, 4)>(p, stream);
            case 1: return launch_x6_t<X6_KARGS(3, 2, 2, 2)>(p, stream);
            case 2: return launch_x6_t<X6_KARGS(3, 1, 4, 8)>(p, stream);
            case 3: return launch_x6_t<X6_KARGS(3, 1, 4, 16)>(p, stream);
            default: return launch_x6_t<X6_KARGS(3, 1, 2, 8)>(p, stream);
        }
    }
    switch (v) {
        case 2: return launch_x6_t<X6_KARGS(1, 1, 4, 8)>(p, stream);
        case 3: return launch_x6_t<X6_KARGS(1, 1, 4, 16)>(p, stream);
        default: return launch_x6_t<X6_KARGS(1, 1, 2, 8)>(p, stream);
    }
}

// ---- 1x1 convolutions with all input channels of a wave's pixels in registers (conv1x1_x6_kernel) -----------------------
// The tile-stream kernel above stages the input once per (tile, 32- or 64-cout slice): a 1x1 with 480 couts (the
// last_layer[0] slices on the low-resolution branches, models/seg_hrnet.py:313-321 after the linearity trick) re-reads and
// re-splits its input fifteen times for steps of 24-48 MFMAs.  Here — the structure of conv1x1.hip in bf16x6 arithmetic —
// a wave owns PG groups of 16 pixels of a row, loads ALL their input channels once (two quads per chunk, x6 K order),
// splits them once into the three exact bf16 terms (registers for the whole launch), and walks the output channels 16 at
// a time; the weight fragments of a cout tile (3 * NCH KB) stream through a double-buffered LDS image shared by the four
// waves.  Per cout tile the 6 * NCH products run as one chain from a fresh accumulator, low-order products first.
template <int NCH, int PG>
__device__ __forceinline__ void x6_c1_body(const ConvParams& p, const long long ntiles, const int tiles_per_row, const int bid, const int G,
                                           const int m0, const int m1) {       // cout tiles [m0, m1) of 16
    constexpr int WFR = 3 * NCH;                    // 1-KB weight fragments per 16-cout tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    bool valid[PG];
    size_t pixel[PG];                               // (n*H + y) * W + x of the lane's pixel in group pg
#pragma unroll
    for (int pg = 0; pg < PG; ++pg) {
        const long long tile = ((long long)xcd_contiguous(bid, G) * 4 + wave) * PG + pg;
        const int k = (int)(tile % tiles_per_row);
        const long long row = tile / tiles_per_row;
        const int col = k * 16 + i;
        valid[pg] = tile < ntiles && col < p.W;
        pixel[pg] = (size_t)row * p.W + col;
    }
    bf16x8 xf[PG][NCH][3];
#pragma unroll
    for (int pg = 0; pg < PG; ++pg)
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = {0.f, 0.f, 0.f, 0.f};
            if (valid[pg]) {
                const float* a = reinterpret_cast<const float*>(p.x) + pixel[pg] * (size_t)p.Cinp + c * 32 + g * 4;
                lo = *reinterpret_cast<const f32x4*>(a);
                hi = *reinterpret_cast<const f32x4*>(a + 16);
            }
            u32x4 t[3];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float va = k < 2 ? lo[2 * k] : hi[2 * k - 4], vb = k < 2 ? lo[2 * k + 1] : hi[2 * k - 3];
                uint32_t h_, m_, l_;
                x6_split_pair(va, vb, h_, m_, l_);
                t[0][k] = h_; t[1][k] = m_; t[2][k] = l_;
            }
#pragma unroll
            for (int q = 0; q < 3; ++q) xf[pg][c][q] = __builtin_bit_cast(bf16x8, t[q]);
        }
    constexpr int WIT = (WFR * 64 + NTHREADS - 1) / NTHREADS;
    const u32x4* wsrc = reinterpret_cast<const u32x4*>(p.w);
    u32x4 wreg[WIT];
    // (every load of the loop unconditional — clamped indices — and pinned at the top of the iteration: left alone hipcc sinks
    // the prefetch to the LDS writes that consume it; the bias of the slice sits in LDS: a global load per cout tile was
    // waited for at the head of every iteration)
    auto prefetch = [&](int m) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < WIT; ++it) {
            const int u = it * NTHREADS + tid < WFR * 64 ? it * NTHREADS + tid : WFR * 64 - 1;
            wreg[it] = wsrc[(size_t)m * WFR * 64 + u];
        }
    };
    auto commit = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < WIT; ++it)
            if (it * NTHREADS + tid < WFR * 64) *reinterpret_cast<u32x4*>(smem + buf * (WFR * 1024) + (it * NTHREADS + tid) * 16) = wreg[it];
    };
    float* const bias_s = reinterpret_cast<float*>(smem + 2 * WFR * 1024);
    for (int i = tid; i < (m1 - m0) * 16; i += NTHREADS) bias_s[i] = p.bias[m0 * 16 + i];
    prefetch(m0);
    commit(0);
    __syncthreads();
    const int rfl = relu_floor(p.relu);
    for (int m = m0; m < m1; ++m) {
        const int buf = (m - m0) & 1;
        prefetch(m + 1 < m1 ? m + 1 : m);
        __builtin_amdgcn_sched_barrier(0);
        const char* wb = smem + buf * (WFR * 1024) + lane * 16;
        const f32x4 bv = *reinterpret_cast<const f32x4*>(bias_s + (m - m0) * 16 + g * 4);
        f32x4 d[PG];
#pragma unroll
        for (int pg = 0; pg < PG; ++pg) d[pg] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            bf16x8 wa[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) wa[q] = *reinterpret_cast<const bf16x8*>(wb + (c * 3 + q) * 1024);
#pragma unroll
            for (int pg = 0; pg < PG; ++pg) {
                d[pg] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[2], xf[pg][c][0], d[pg], 0, 0, 0);
                d[pg] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[0], xf[pg][c][2], d[pg], 0, 0, 0);
                d[pg] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[1], xf[pg][c][1], d[pg], 0, 0, 0);
                d[pg] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[1], xf[pg][c][0], d[pg], 0, 0, 0);
                d[pg] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[0], xf[pg][c][1], d[pg], 0, 0, 0);
            }
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const bf16x8 w0 = *reinterpret_cast<const bf16x8*>(wb + (c * 3) * 1024);
#pragma unroll
            for (int pg = 0; pg < PG; ++pg) d[pg] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, xf[pg][c][0], d[pg], 0, 0, 0);
        }
#pragma unroll
        for (int pg = 0; pg < PG; ++pg) {
            u32x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = __float_as_uint(relu_opt(d[pg][k] + bv[k], rfl));
            if (valid[pg]) *reinterpret_cast<u32x4*>(p.y + (pixel[pg] * (size_t)p.Coutp + m * 16 + g * 4) * 4) = o;
        }
        __builtin_amdgcn_sched_barrier(0);
        commit(buf ^ 1);
        __syncthreads();
    }
}

template <int NCH, int PG>
__global__ __launch_bounds__(NTHREADS, 2) void conv1x1_x6_kernel(ConvParams p, long long ntiles, int tiles_per_row, int per) {
    // blockIdx.y: a slice of `per` cout tiles (wide outputs on small grids: the pixel groups alone do not fill the chip)
    const int m0 = (int)blockIdx.y * per, ntile16 = p.Coutp >> 4;
    x6_c1_body<NCH, PG>(p, ntiles, tiles_per_row, (int)blockIdx.x, (int)gridDim.x, m0, m0 + per < ntile16 ? m0 + per : ntile16);
}

// the fuse-up 1x1 convolutions of an HRModule (models/seg_hrnet.py:176-197) in one launch
constexpr int X6_C1_MAXJOBS = 6;
struct X6C1Jobs {
    ConvParams p[X6_C1_MAXJOBS];
    long long ntiles[X6_C1_MAXJOBS];
    int tiles_per_row[X6_C1_MAXJOBS];
    int start[X6_C1_MAXJOBS + 1];
    int njobs;
};
__global__ __launch_bounds__(NTHREADS, 2) void conv1x1_x6_jobs_kernel(X6C1Jobs jobs) {
    const int b = (int)blockIdx.x;
    int j = 0;
#pragma unroll
    for (int k = 1; k < X6_C1_MAXJOBS; ++k) j += (k < jobs.njobs && b >= jobs.start[k]) ? 1 : 0;
    const int bid = b - jobs.start[j], G = jobs.start[j + 1] - jobs.start[j];
    const ConvParams& p = jobs.p[j];
    switch (p.Cinp >> 5) {
        case 2: x6_c1_body<2, 1>(p, jobs.ntiles[j], jobs.tiles_per_row[j], bid, G, 0, p.Coutp >> 4); break;
        case 4: x6_c1_body<4, 1>(p, jobs.ntiles[j], jobs.tiles_per_row[j], bid, G, 0, p.Coutp >> 4); break;
        default: x6_c1_body<8, 1>(p, jobs.ntiles[j], jobs.tiles_per_row[j], bid, G, 0, p.Coutp >> 4); break;
    }
}

template <int NCH, int PG>
int launch_c1_x6_t(const ConvParams& p, hipStream_t stream) {
    const int lds = 2 * 3 * NCH * 1024;
    const int tiles_per_row = (p.W + 15) / 16;
    const long long ntiles = (long long)p.N * p.H * tiles_per_row;
    const long long nblk = (ntiles + 4 * PG - 1) / (4 * PG);
    if (nblk <= 0 || nblk > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    // cout slices: enough workgroups for four per CU, at least four cout tiles each (a slice re-loads and re-splits the input)
    const int ntile16 = p.Coutp >> 4;
    long long want = (4LL * device_cus() + nblk - 1) / nblk;
    if (want > ntile16 / 4) want = ntile16 / 4;
    const int slices = want < 1 ? 1 : (int)want;
    const int per = (ntile16 + slices - 1) / slices;
    auto kern = conv1x1_x6_kernel<NCH, PG>;
    if (const int e_ = ensure_dyn_lds(reinterpret_cast<const void*>(kern), lds + per * 64)) return e_;      // + the slice's bias
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk, (unsigned)((ntile16 + per - 1) / per)), dim3(NTHREADS), lds + per * 64, stream, p, ntiles, tiles_per_row, per);
    return (int)hipGetLastError();
}

// ---- fused stem of the fp32-grade mode ----------------------------------------------------------------------------------
bool stem_fused_x6_supported(int cin, int cmid, int coutp) { return X6_MODE == 2 && cin == 1 && cmid == 64 && coutp % 64 == 0; }

// conv1's folded weights [64][1][3][3] + bias [64] -> [2 chunks][4 k-groups][9 taps + bias][8] f32 (X6StemSrc::w1)
void pack_stem_w1_x6(const float* w, const float* b, float* dst) {
    for (int c = 0; c < 2; ++c)
        for (int sg = 0; sg < 4; ++sg)
            for (int j = 0; j < 8; ++j) {
                const int ch = c * 32 + x6_chan_of_k(sg, j);
                float* o = dst + (c * 4 + sg) * 80;
                for (int t = 0; t < 9; ++t) o[t * 8 + j] = w[ch * 9 + t];
                o[72 + j] = b[ch];
            }
}

int launch_stem_fused_x6(const StemFusedParams& sp, hipStream_t stream) {
    if (!stem_fused_x6_supported(sp.cin, sp.Cmid, sp.Coutp)) return (int)hipErrorInvalidValue;
    using C = X6Cfg<3, 2, 4, X6_STEM_NR, 4, X6_STEM_NBUF>;
    ConvParams p{};
    p.x = nullptr; p.y = sp.y; p.res = nullptr; p.w = sp.w2; p.bias = sp.bias2;
    p.N = sp.N; p.H = sp.H; p.W = sp.W; p.OH = sp.OH; p.OW = sp.OW; p.Cinp = sp.Cmid; p.Coutp = sp.Coutp; p.relu = 1; p.fmt = FMT_F32;
    if (sp.OH != (sp.H + 1) / 2 || sp.OW != (sp.W + 1) / 2 || (long long)sp.H * sp.W * 4 >= 0x7fffffffLL ||
        (long long)sp.OH * sp.OW * sp.Coutp * 4 >= 0x7fffffffLL)
        return (int)hipErrorInvalidValue;
    X6Geo geo;
    geo.tiles_x = (p.OW + TW - 1) / TW;
    geo.tiles_y = (p.OH + C::TH - 1) / C::TH;
    geo.ctiles = p.Coutp / 64;
    const long long nitems = (long long)p.N * geo.tiles_y * geo.tiles_x * geo.ctiles;
    if (nitems <= 0 || nitems > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    geo.nitems = (int)nitems;
    geo.m_ct = x6_magic(geo.ctiles);
    geo.m_tx = x6_magic(geo.tiles_x);
    geo.m_ty = x6_magic(geo.tiles_y);
    const int slots = 2 * device_cus();
    int grid = (int)(nitems < slots ? nitems : slots);
    if (grid > geo.ctiles) grid -= grid % geo.ctiles;
    constexpr int LDS = C::LDS + 2 * 4 * 10 * 8 * 4 + 2 * (C::IW + 2) * (C::IH + 2) * 4;       // tiles, conv1 weights, raw crops
    if (const int e_ = ensure_dyn_lds(reinterpret_cast<const void*>(stem_x6_kernel), LDS)) return e_;
    X6StemSrc src{sp.x, sp.w1};
    hipLaunchKernelGGL(stem_x6_kernel, dim3((unsigned)grid), dim3(NTHREADS), LDS, stream, p, geo, src);
    return (int)hipGetLastError();
}

// 1x1 with 64 / 128 / 256 (or 32 / 96 / 192) input channels and no residual: the register-resident form
bool conv1x1_x6_supported(const ConvParams& p) {
    const int n = p.Cinp / 32;
    // measured (batch 32) against the tile-stream kernel: 64 -> 480 on a 64x64 grid 135 -> 85 us, 128 -> 480 on 32x32 62 -> 48,
    // 256 -> 480 on 16x16 30 -> 23 (with the cout range cut into slices, blockIdx.y, where the pixel groups alone leave CUs
    // idle) — so: every WIDE output (256 couts and more: the head's products; the fuse-up 1x1s of a module stay with the
    // tile-stream kernel, which also serves their merged launch — a rule on the layer, never on the batch: the two kernels
    // sum in different orders)
    return p.fmt == FMT_F32 && !p.res && !p.out_f32 && p.nheads <= 1 && (p.Cinp % 32) == 0 && (p.Coutp % 16) == 0 && p.H == p.OH &&
           p.W == p.OW && (n == 1 || n == 2 || n == 3 || n == 4 || n == 6 || n == 8) && p.Coutp >= X6_C1_MINCOUT;
}
int launch_conv1x1_x6(const ConvParams& p, hipStream_t stream) {
    if (!conv1x1_x6_supported(p)) return (int)hipErrorInvalidValue;
    switch (p.Cinp / 32) {
        case 1: return launch_c1_x6_t<1, 2>(p, stream);
        case 2: return launch_c1_x6_t<2, 2>(p, stream);
        case 3: return launch_c1_x6_t<3, 2>(p, stream);
        case 4: return launch_c1_x6_t<4, 2>(p, stream);
        case 6: return launch_c1_x6_t<6, 1>(p, stream);
        default: return launch_c1_x6_t<8, 1>(p, stream);
    }
}
bool conv1x1_x6_jobs_supported(const ConvParams* ps, int n) {
    if (n < 2 || n > X6_C1_MAXJOBS || !X6_C1_JOBS) return false;      // (fuse-up 1x1s: 24 -> 30, 32 -> 48 us: off, a build flag)
    for (int j = 0; j < n; ++j) {
        const int nch = ps[j].Cinp / 32;
        if (!conv1x1_x6_supported(ps[j]) || (nch != 2 && nch != 4 && nch != 8)) return false;
    }
    return true;
}
int launch_conv1x1_x6_jobs(const ConvParams* ps, int n, hipStream_t stream) {
    if (!conv1x1_x6_jobs_supported(ps, n)) return (int)hipErrorInvalidValue;
    X6C1Jobs jobs{};
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
    for (int k = n; k <= X6_C1_MAXJOBS; ++k) jobs.start[k] = at;
    int coutp_max = 0;
    for (int j = 0; j < n; ++j) coutp_max = ps[j].Coutp > coutp_max ? ps[j].Coutp : coutp_max;
    const int lds = 2 * 3 * 8 * 1024 + coutp_max * 4;       // weights of the widest member + a member's bias
    if (const int e_ = ensure_dyn_lds(reinterpret_cast<const void*>(conv1x1_x6_jobs_kernel), lds)) return e_;
    hipLaunchKernelGGL(conv1x1_x6_jobs_kernel, dim3((unsigned)at), dim3(NTHREADS), lds, stream, jobs);
    return (int)hipGetLastError();
}

// 2..6 independent convolutions of one kernel size and stride (each one a conv_x6 launch on its own) as ONE launch
bool conv_x6_jobs_supported(const ConvParams* ps, int n, int k, int stride) {
    if (X6_MODE != 2 || n < 2 || n > X6_MAXJOBS) return false;
    for (int j = 0; j < n; ++j)
        if (!conv_x6_supported(ps[j], k, stride)) return false;
    return true;
}

int launch_conv_x6_jobs(const ConvParams* ps, int n, int k, int stride, hipStream_t stream) {
    if (!conv_x6_jobs_supported(ps, n, k, stride)) return (int)hipErrorInvalidValue;
    if (k == 1 && conv1x1_x6_jobs_supported(ps, n)) return launch_conv1x1_x6_jobs(ps, n, stream);
    if (k == 1) return launch_x6_jobs_t<1, 1>(ps, n, stream);
    return stride == 1 ? launch_x6_jobs_t<3, 1>(ps, n, stream) : launch_x6_jobs_t<3, 2>(ps, n, stream);
}

const char* conv_x6_kernel_name(const ConvParams& p, int k, int stride) {
    // as rocprofv3 prints the instantiations of launch_conv_x6 (X6_KARGS evaluated), by x6_variant
#if X6_MODE == 2
    static const char* const n3[] = {X6_S2_SINGLE ? "conv_x6_kernel<3, 2, 4, 4, 4, 2, 1>" : "conv_x6_kernel<3, 2, 4, 2, 4, 2, 2>",
                                     X6_S2_SINGLE ? "conv_x6_kernel<3, 2, 2, 2, 4, 2, 1>" : "conv_x6_kernel<3, 2, 2, 1, 4, 2, 2>",
                                     "conv_x6_kernel<3, 1, 4, 8, 4, 2, 2>", "conv_x6_kernel<3, 1, 4, 8, 4, 2, 2>", "conv_x6_kernel<3, 1, 2, 4, 4, 2, 2>"};
    static const char* const n1[] = {"", "", "conv_x6_kernel<1, 1, 4, 8, 4, 2, 2>", "conv_x6_kernel<1, 1, 4, 8, 4, 2, 2>", "conv_x6_kernel<1, 1, 2, 4, 4, 2, 2>"};
#elif X6_MODE == 1
    static const char* const n3[] = {"conv_x6_kernel<3, 2, 4, 2, 8, 1, 2>", "conv_x6_kernel<3, 2, 2, 1, 8, 1, 2>", "conv_x6_kernel<3, 1, 4, 4, 8, 1, 2>",
                                     "conv_x6_kernel<3, 1, 4, 8, 8, 1, 2>", "conv_x6_kernel<3, 1, 2, 4, 8, 1, 2>"};
    static const char* const n1[] = {"", "", "conv_x6_kernel<1, 1, 4, 4, 8, 1, 2>", "conv_x6_kernel<1, 1, 4, 8, 8, 1, 2>", "conv_x6_kernel<1, 1, 2, 4, 8, 1, 2>"};
#else
    static const char* const n3[] = {"conv_x6_kernel<3, 2, 4, 4, 4, 1, 2>", "conv_x6_kernel<3, 2, 2, 2, 4, 1, 2>", "conv_x6_kernel<3, 1, 4, 8, 4, 1, 2>",
                                     "conv_x6_kernel<3, 1, 4, 16, 4, 1, 2>", "conv_x6_kernel<3, 1, 2, 8, 4, 1, 2>"};
    static const char* const n1[] = {"", "", "conv_x6_kernel<1, 1, 4, 8, 4, 1, 2>", "conv_x6_kernel<1, 1, 4, 16, 4, 1, 2>", "conv_x6_kernel<1, 1, 2, 8, 4, 1, 2>"};
#endif
    if (k == 1 && conv1x1_x6_supported(p)) return "conv1x1_x6_kernel";
    const int v = x6_variant(p, k, stride);
    return k == 3 ? n3[v] : n1[v];
}

}  // namespace esa
