// head_fused2.hip — last_layer[0..5] in one kernel with the bilinear up-sampling done by the matrix
// cores (second generation of head_fused.hip, which remains the fallback for geometries this one
// does not cover):
//
//   h0 = ReLU( W0·x0 + bias0 + sum_{b=1..3} U_b·t_b )          t_b = W_b·x_b on branch b's grid
//   h3 = ReLU( W3·h0 + bias3 )
//
// Replaces (reference): the three F.upsample + torch.cat of models/seg_hrnet.py:461-466 and
// last_layer[0..5] (:313-329).  Bilinear interpolation is a linear map over source pixels, and for
// one output row only two source rows per branch carry non-zero weight, so per 16-pixel output row
//     sum_b U_b·t_b  =  T[32 ch][64 slots] · U[64 slots][16 px]
// with the 64 contraction slots = 2 rows x (16 + 8 + 8) source columns of the three branches: two
// K-chunks of the 16x16x32 MFMA.  U (the interpolation weights) depends on the output row and column
// only — built once per wave, kept in registers; for the 2x/4x/8x ratios its entries are exact in bf16
// (numerators < 256), otherwise a lo part is carried (ULO).  T comes from LDS as [slot group][part]
// [channel][8 source pixels] bf16 — the "T layout":
//   * branch 1 (64 of the 480 input channels, 252 MB if materialised at batch 32): t_1 of the
//     workgroup's 11x16 source window is computed here, per 32-channel chunk, by 11 of the 16 waves
//     (A = x_1 fragments held in registers, B = W_1 fragments), and written to LDS straight from the
//     accumulator layout (lane = channel, 4 consecutive pixels);
//   * branches 2, 3: produced in T layout by head_t.hip, staged with 8-byte loads.
// The sum accumulates in the same registers as W0·x0; ReLU + split turn the accumulator into the
// B operand of W3 (permuted K order, as in head_fused.hip).  One barrier per 32-channel chunk; the
// chunk c+1 data (t_1 tile, t_2/t_3 windows, W0/W3/bias) and the W_1 fragments of chunk c+2 are
// produced while chunk c is consumed.
#include "devstate.h"
#include "kernels.h"
#include "sb.h"

// ablation switches for tuning experiments (all 1 in the shipped build)
#ifndef H2_DO_T1
#define H2_DO_T1 1
#endif
#ifndef H2_DO_INTERP
#define H2_DO_INTERP 1
#endif
#ifndef H2_DO_W0
#define H2_DO_W0 1
#endif
#ifndef H2_DO_W3
#define H2_DO_W3 1
#endif
#ifndef H2_NW
#define H2_NW 8                 // waves (= output rows) per workgroup: 8 -> two workgroups per CU
#endif
#ifndef H2_DO_BAR
#define H2_DO_BAR 1
#endif

namespace esa {
namespace {

constexpr int HTW = 16;                         // tile width = one MFMA N-tile; tile height = waves per workgroup
// max source rows per branch and tile of NW output rows (2x / 4x / 8x coarser grids, +3 for the taps)
constexpr int rows1(int nw) { return nw / 2 + 3; }
constexpr int rows2(int nw) { return nw / 4 + 3; }
constexpr int rows3(int nw) { return nw / 8 + 3; }

struct Lerp2 {
    int i0, i1;
    float l0, l1;
};
__device__ __host__ inline Lerp2 lerp2(int dst, int in, int out) {     // ATen align_corners=False
    const float scale = (float)in / (float)out;
    float src = scale * ((float)dst + 0.5f) - 0.5f;
    src = src < 0.f ? 0.f : src;
    Lerp2 r;
    r.i0 = (int)src < in - 1 ? (int)src : in - 1;
    r.i1 = r.i0 + (r.i0 < in - 1 ? 1 : 0);
    r.l1 = src - (float)r.i0;
    r.l0 = 1.f - r.l1;
    return r;
}

template <int NW, int NCH0, int NCH1, int M3, bool ULO>
// (the wide variants need > 80 KB of LDS, i.e. one workgroup per CU anyway: let them have 256 VGPRs)
__global__ __launch_bounds__(NW * 64, (NCH0 > 1 || NCH1 > 2) ? 2 : 4) void head_fused2_kernel(Head2Params p, int tiles_x, int tiles_y) {
    constexpr int H2THREADS = NW * 64;
    constexpr int R1 = rows1(NW), R2 = rows2(NW), R3 = rows3(NW);
    constexpr int T1OFF = 0, T2OFF = R1 * 2 * 1024, T3OFF = T2OFF + R2 * 1024, TBUF = T3OFF + R3 * 1024;
    constexpr int W03FR = 4 * NCH0 + 2 * M3;             // W0 then W3 fragments of one chunk
    constexpr int W03S = W03FR * 1024 + 256;             // + 32 bias floats
    constexpr int W1FR = 4 * NCH1;
    constexpr int W1S = W1FR * 1024;
    constexpr int WLD = (W03FR + 1 + NW - 1) / NW;       // W0/W3/bias items per wave and chunk
    constexpr int W1LD = (W1FR + NW - 1) / NW;           // W_1 fragments per wave and chunk
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const tb0 = smem;
    char* const w03b = smem + 2 * TBUF;
    char* const w1b = w03b + 2 * W03S;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int px = lane & 15, q = lane >> 4;
    int b_ = xcd_contiguous(blockIdx.x, gridDim.x);
    const int tx = b_ % tiles_x; b_ /= tiles_x;
    const int ty = b_ % tiles_y;
    const int n = b_ / tiles_y;
    const int oy0 = ty * NW, ox0 = tx * HTW;
#ifdef H2_NCHUNKS
    const int nchunks = H2_NCHUNKS;         // timing experiments only
#else
    const int nchunks = p.Ctp >> 5;
#endif

    // ---- source windows of the three low-resolution branches (workgroup-uniform) ---------------
    int ry0[3], rh[3], ws[3], rw0;
    {
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const Lerp2 a = lerp2(oy0, p.th[b], p.H), e = lerp2(min(oy0 + NW - 1, p.H - 1), p.th[b], p.H);
            const Lerp2 c = lerp2(ox0, p.tw[b], p.W);
            ry0[b] = a.i0; rh[b] = e.i1 - a.i0 + 1;
            // window slot 0 in source columns: branch 1 starts at the first needed column, branches
            // 2/3 at the 8-byte aligned stored column below it (stored column = x + HT_PAD)
            ws[b] = b == 0 ? c.i0 : ((c.i0 + HT_PAD) & ~3) - HT_PAD;
        }
        const Lerp2 d = lerp2(min(ox0 + HTW - 1, p.W - 1), p.tw[0], p.W);
        rw0 = d.i1 - ws[0] + 1;
    }

    // ---- staging maps.  Everything below is branch-free on purpose: every thread issues the same four
    // loads per chunk and the same four LDS writes (threads without a unit of their own repeat another
    // thread's — same data to the same address), so the compiler can keep two generations of loads in
    // flight with counted waits instead of draining vmcnt at every control-flow join.
    // t_2 / t_3 windows: unit = (row, part, channel, 4-pixel half), 8 B each
    const char* sg[2];
    int sstride[2], sdst[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int n2 = rh[1] * 128, n3 = rh[2] * 128;
        const int u = (it * H2THREADS + tid) % (n2 + n3);
        const int b = u < n2 ? 1 : 2;
        const int v = u < n2 ? u : u - n2;
        const int r = v >> 7, part = (v >> 6) & 1, ch = (v >> 1) & 31, half = v & 1;
        const int xp = b == 1 ? p.xp2 : p.xp3;
        const char* base = b == 1 ? p.t2 : p.t3;
        sg[it] = base + ((((size_t)n * p.th[b] + ry0[b] + r) * nchunks * 2 + part) * 32 + ch) * (size_t)(xp * 2)
                 + (size_t)(ws[b] + HT_PAD + half * 4) * 2;
        sstride[it] = 128 * xp;                          // bytes between chunks: 2 parts x 32 ch x XP x 2 B
        sdst[it] = (b == 1 ? T2OFF : T3OFF) + v * 8;
    }
    // W0 / W3 fragments and bias0 of a chunk: item f = j*NW + wave (items past the bias item repeat it,
    // its lanes >= 8 repeat lanes 0..7); W_1 fragments: (j*NW + wave) % W1FR
    const uint4* wsrc[WLD];
    int wstep[WLD], wdst[WLD];
#pragma unroll
    for (int j = 0; j < WLD; ++j) {
        const int f = j * NW + wave < W03FR ? j * NW + wave : W03FR;
        if (f < 4 * NCH0) { wsrc[j] = p.w0 + f * 64 + lane; wstep[j] = 4 * NCH0 * 64; }
        else if (f < W03FR) { wsrc[j] = p.w3 + ((size_t)((f - 4 * NCH0) >> 1) * nchunks * 2 + ((f - 4 * NCH0) & 1)) * 64 + lane; wstep[j] = 128; }
        else { wsrc[j] = reinterpret_cast<const uint4*>(p.bias0) + (lane & 7); wstep[j] = 8; }
        wdst[j] = f < W03FR ? (f * 64 + lane) * 16 : W03FR * 1024 + (lane & 7) * 16;
    }
    const uint4* w1src[W1LD];
    int w1dst[W1LD];
#pragma unroll
    for (int j = 0; j < W1LD; ++j) {
        const int f = (j * NW + wave) % W1FR;
        w1src[j] = p.w1 + f * 64 + lane;
        w1dst[j] = (f * 64 + lane) * 16;
    }
    const int clast = nchunks - 1;
    // two staging register sets (A, B): the loads of chunk c+2 are issued while chunk c is consumed
    // and committed to LDS a whole iteration later — one workgroup per CU has nobody else to hide
    // the ~2 us load latency behind.  Chunk indices past the end are clamped (a redundant reload).
    typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;      // (HIP's uint4 struct in an array ends up in scratch)
    typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
    u32x2 srA[2], srB[2];
    u32x4 wregA[WLD], w1regA[W1LD], wregB[WLD], w1regB[W1LD];
    // chunk CH: t_2/t_3 windows + W0/W3/bias0; chunk CH1: W_1 fragments
#define H2_PREFETCH(CH, CH1, S)                                                               \
    {                                                                                         \
        const int ch_ = min((CH), clast), ch1_ = min((CH1), clast);                           \
        _Pragma("unroll") for (int it = 0; it < 2; ++it)                                      \
            sr##S[it] = *reinterpret_cast<const u32x2*>(sg[it] + (size_t)ch_ * sstride[it]);  \
        _Pragma("unroll") for (int j = 0; j < WLD; ++j) wreg##S[j] = *reinterpret_cast<const u32x4*>(wsrc[j] + (size_t)ch_ * wstep[j]); \
        _Pragma("unroll") for (int j = 0; j < W1LD; ++j) w1reg##S[j] = *reinterpret_cast<const u32x4*>(w1src[j] + (size_t)ch1_ * W1FR * 64); \
    }
#define H2_COMMIT(CH, CH1, S)                                                                 \
    {                                                                                         \
        _Pragma("unroll") for (int it = 0; it < 2; ++it)                                      \
            *reinterpret_cast<u32x2*>(tb0 + ((CH) & 1) * TBUF + sdst[it]) = sr##S[it];        \
        _Pragma("unroll") for (int j = 0; j < WLD; ++j)                                       \
            *reinterpret_cast<u32x4*>(w03b + ((CH) & 1) * W03S + wdst[j]) = wreg##S[j];       \
        _Pragma("unroll") for (int j = 0; j < W1LD; ++j)                                      \
            *reinterpret_cast<u32x4*>(w1b + ((CH1) & 1) * W1S + w1dst[j]) = w1reg##S[j];      \
    }

    // ---- per-lane constants -------------------------------------------------------------------------
    const int ox = ox0 + px, oy = oy0 + wave;
    const bool in = ox < p.W && oy < p.H;
    bf16x8 x0h[NCH0], x0l[NCH0];
#pragma unroll
    for (int c = 0; c < NCH0; ++c) {
        uint4 h = make_uint4(0, 0, 0, 0), l = make_uint4(0, 0, 0, 0);
        if (in) {
            const char* a = p.x0 + (((size_t)n * p.H + oy) * p.W + ox) * (size_t)(p.C0p * 4) + c * 128 + q * 32;
            h = *reinterpret_cast<const uint4*>(a);
            l = *reinterpret_cast<const uint4*>(a + 16);
        }
        x0h[c] = __builtin_bit_cast(bf16x8, h);
        x0l[c] = __builtin_bit_cast(bf16x8, l);
    }
    // x_1 fragments of this wave's source row (waves >= rh[0] have none): A operand, rows = pixels
    const bool t1wave = wave < rh[0];
    bf16x8 x1h[NCH1], x1l[NCH1];
#pragma unroll
    for (int c = 0; c < NCH1; ++c) {
        uint4 h = make_uint4(0, 0, 0, 0), l = make_uint4(0, 0, 0, 0);
        if (t1wave && px < rw0) {
            const char* a = p.x1 + (((size_t)n * p.th[0] + ry0[0] + wave) * p.tw[0] + ws[0] + px) * (size_t)(p.C1p * 4)
                            + c * 128 + q * 32;
            h = *reinterpret_cast<const uint4*>(a);
            l = *reinterpret_cast<const uint4*>(a + 16);
        }
        x1h[c] = __builtin_bit_cast(bf16x8, h);
        x1l[c] = __builtin_bit_cast(bf16x8, l);
    }
    // interpolation operand U (B: column = this lane's pixel, K group = q) and the LDS offsets of the
    // matching T fragments (A: row = channel px, K group = q).  Slot groups:
    //   chunk 0:  q = (row sel << 1 | column group) of branch 1;   chunk 1:  q = (branch 2|3) << 1 | row sel
    bf16x8 uh[2], ul[2];
    int offA[2];
    {
        const int oxc = min(ox, p.W - 1), oyc = min(oy, p.H - 1);
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
            const int b = kc == 0 ? 0 : 1 + (q >> 1);
            const int sel = kc == 0 ? (q >> 1) : (q & 1);
            const int cg = kc == 0 ? (q & 1) : 0;
            const Lerp2 lx = lerp2(oxc, p.tw[b], p.W), ly = lerp2(oyc, p.th[b], p.H);
            const float wy = sel ? ly.l1 : ly.l0;
            const int row = (sel ? ly.i1 : ly.i0) - ry0[b];
            offA[kc] = (b == 0 ? T1OFF + (row * 2 + cg) * 1024 : (b == 1 ? T2OFF : T3OFF) + row * 1024) + px * 16;
            float u[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int sc = ws[b] + cg * 8 + j;
                u[j] = wy * ((sc == lx.i0 ? lx.l0 : 0.f) + (sc == lx.i1 ? lx.l1 : 0.f));
            }
            uint4 hb, lb;
            split8(u, hb, lb);
            uh[kc] = __builtin_bit_cast(bf16x8, hb);
            ul[kc] = __builtin_bit_cast(bf16x8, lb);
        }
    }

    // t_1 of chunk CH (W_1 fragments of that chunk are in w1 buffer CH&1) -> T buffer CH&1
#define H2_T1(CH)                                                                             \
    if (H2_DO_T1 && t1wave) {                                                                 \
        const char* wb1 = w1b + ((CH) & 1) * W1S + lane * 16;                                 \
        char* td = tb0 + ((CH) & 1) * TBUF + T1OFF + (wave * 2 + (q >> 1)) * 1024 + px * 16 + (q & 1) * 8; \
        _Pragma("unroll") for (int m = 0; m < 2; ++m) {                                       \
            f32x4 d = {0.f, 0.f, 0.f, 0.f};                                                   \
            _Pragma("unroll") for (int c = 0; c < NCH1; ++c) {                                \
                const bf16x8 wh = *reinterpret_cast<const bf16x8*>(wb1 + ((m * NCH1 + c) * 2 + 0) * 1024); \
                const bf16x8 wl = *reinterpret_cast<const bf16x8*>(wb1 + ((m * NCH1 + c) * 2 + 1) * 1024); \
                d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x1l[c], wh, d, 0, 0, 0);          \
                d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x1h[c], wl, d, 0, 0, 0);          \
                d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x1h[c], wh, d, 0, 0, 0);          \
            }                                                                                 \
            const float v_[4] = {d[0], d[1], d[2], d[3]};                                     \
            uint2 hi_, lo_;                                                                   \
            split4(v_, hi_, lo_);                                                             \
            *reinterpret_cast<uint2*>(td + m * 256) = hi_;                                    \
            *reinterpret_cast<uint2*>(td + m * 256 + 512) = lo_;                              \
        }                                                                                     \
    }

    f32x4 acc3[M3];
#pragma unroll
    for (int m = 0; m < M3; ++m) acc3[m] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- prologue: W_1(0); then chunk 0's data + W_1(1) while t_1(0) is computed; set A then takes
    // chunk 1 + W_1(2), which iteration 0 commits ---------------------------------------------------------
    H2_PREFETCH(0, 0, B)
    H2_COMMIT(0, 0, B)
    __syncthreads();
    H2_PREFETCH(0, 1, A)
    H2_T1(0)
    H2_COMMIT(0, 1, A)
    __syncthreads();
    H2_PREFETCH(1, 2, A)

    // iteration cc: issue the loads of (cc+2, cc+3) into set SP, consume chunk cc, produce t_1(cc+1),
    // commit set SC = (cc+1, cc+2) issued one iteration ago
#define H2_ITER(SP, SC)                                                                       \
    {                                                                                         \
        H2_PREFETCH(cc + 2, cc + 3, SP)                                                       \
        h2_consume(cc);                                                                       \
        if (cc + 1 < nchunks) {                                                               \
            H2_T1(cc + 1)                                                                     \
            H2_COMMIT(cc + 1, cc + 2, SC)   /* nobody reads T/W03 buffer (cc+1)&1 or W_1 buffer cc&1 now */ \
            if (H2_DO_BAR) __syncthreads();                                                   \
        }                                                                                     \
    }
    auto h2_consume = [&](int cc) __attribute__((always_inline)) {
        const char* tb = tb0 + (cc & 1) * TBUF;
        const char* wb = w03b + (cc & 1) * W03S;
        f32x4 a[2];
        a[0] = *reinterpret_cast<const f32x4*>(wb + W03FR * 1024 + q * 16);
        a[1] = *reinterpret_cast<const f32x4*>(wb + W03FR * 1024 + 64 + q * 16);
        // (1) W0·x0
#pragma unroll
        for (int m = 0; m < (H2_DO_W0 ? 2 : 0); ++m)
#pragma unroll
            for (int c = 0; c < NCH0; ++c) {
                const bf16x8 ah = *reinterpret_cast<const bf16x8*>(wb + lane * 16 + ((m * NCH0 + c) * 2 + 0) * 1024);
                const bf16x8 al = *reinterpret_cast<const bf16x8*>(wb + lane * 16 + ((m * NCH0 + c) * 2 + 1) * 1024);
                a[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, x0h[c], a[m], 0, 0, 0);
                a[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, x0l[c], a[m], 0, 0, 0);
                a[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, x0h[c], a[m], 0, 0, 0);
            }
        // (2) + sum_b U_b·t_b
#pragma unroll
        for (int kc = 0; kc < (H2_DO_INTERP ? 2 : 0); ++kc)
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const bf16x8 th_ = *reinterpret_cast<const bf16x8*>(tb + offA[kc] + m * 256);
                const bf16x8 tl_ = *reinterpret_cast<const bf16x8*>(tb + offA[kc] + m * 256 + 512);
                a[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tl_, uh[kc], a[m], 0, 0, 0);
                if (ULO) a[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(th_, ul[kc], a[m], 0, 0, 0);
                a[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(th_, uh[kc], a[m], 0, 0, 0);
            }
        // (3) ReLU, split: the accumulator pair is the B operand of W3 under the permuted K order
        float v[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[i] = relu1(a[0][i]);
            v[4 + i] = relu1(a[1][i]);
        }
        uint4 hb, lb;
        split8(v, hb, lb);
        const bf16x8 hh = __builtin_bit_cast(bf16x8, hb), hl = __builtin_bit_cast(bf16x8, lb);
#pragma unroll
        for (int m = 0; m < (H2_DO_W3 ? M3 : 0); ++m) {
            const bf16x8 a3h = *reinterpret_cast<const bf16x8*>(wb + lane * 16 + (4 * NCH0 + m * 2 + 0) * 1024);
            const bf16x8 a3l = *reinterpret_cast<const bf16x8*>(wb + lane * 16 + (4 * NCH0 + m * 2 + 1) * 1024);
            acc3[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3l, hh, acc3[m], 0, 0, 0);
            acc3[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3h, hl, acc3[m], 0, 0, 0);
            acc3[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3h, hh, acc3[m], 0, 0, 0);
        }
    };
    for (int cc = 0; cc < nchunks; ++cc) {
        H2_ITER(B, A)
        if (++cc >= nchunks) break;
        H2_ITER(A, B)
    }
#undef H2_ITER
#undef H2_PREFETCH
#undef H2_COMMIT
#undef H2_T1

    // ---- epilogue: h3 = ReLU(acc3 + bias3) -> SB [N][H][W][C3p] --------------------------------
    if (in) {
        char* o = p.y + (((size_t)n * p.H + oy) * p.W + ox) * (size_t)(p.C3p * 4);
#pragma unroll
        for (int m = 0; m < M3; ++m) {
            const int co = m * 16 + q * 4;
            const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias3 + co);
            const int cofs = (co >> 3) * 32 + ((co >> 2) & 1) * 8;
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = relu1(acc3[m][i] + bv[i]);
            uint2 hi, lo;
            split4(v, hi, lo);
            *reinterpret_cast<uint2*>(o + cofs) = hi;
            *reinterpret_cast<uint2*>(o + cofs + 16) = lo;
        }
        for (int c = M3 * 16 + q * 4; c < p.C3p; c += 16) {      // keep the padded channels exact zeros
            const int zo = (c >> 3) * 32 + ((c >> 2) & 1) * 8;
            *reinterpret_cast<uint2*>(o + zo) = make_uint2(0, 0);
            *reinterpret_cast<uint2*>(o + zo + 16) = make_uint2(0, 0);
        }
    }
}

template <int NW, int NCH0, int NCH1, int M3, bool ULO>
int launch_head2_t(const Head2Params& p, hipStream_t stream) {
    auto kern = head_fused2_kernel<NW, NCH0, NCH1, M3, ULO>;
    const int lds = 2 * (rows1(NW) * 2 + rows2(NW) + rows3(NW)) * 1024 + 2 * ((4 * NCH0 + 2 * M3) * 1024 + 256) +
                    2 * 4 * NCH1 * 1024;
    if (const int e_ = ensure_dyn_lds(reinterpret_cast<const void*>(kern), lds)) return e_;
    const int tiles_x = (p.W + HTW - 1) / HTW, tiles_y = (p.H + NW - 1) / NW;
    const long long nblk = (long long)p.N * tiles_x * tiles_y;
    if (nblk <= 0 || nblk > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(NW * 64), lds, stream, p, tiles_x, tiles_y);
    return (int)hipGetLastError();
}

template <int NCH0, int NCH1, int M3>
int launch_head2_u(const Head2Params& p, bool ulo, hipStream_t stream) {
    return ulo ? launch_head2_t<H2_NW, NCH0, NCH1, M3, true>(p, stream)
               : launch_head2_t<H2_NW, NCH0, NCH1, M3, false>(p, stream);
}

inline bool bf16_exact(float f) {
    uint32_t u;
    __builtin_memcpy(&u, &f, 4);
    return (u & 0xffffu) == 0;
}

}  // namespace

// Geometry check (host mirror of the kernel's window arithmetic) for every tile of the output grid;
// *ulo is set when some interpolation weight is not exactly representable in bf16.
bool head_fused2_supported(int H, int W, const int th[3], const int tw[3], int C0p, int C1p, int K, bool* ulo) {
    if (C0p != 32 && C0p != 64) return false;
    if (C1p != 64 && C1p != 96) return false;
    if (K < 1 || K > 32) return false;
    constexpr int NW = H2_NW;
    const int rmax[3] = {rows1(NW), rows2(NW), rows3(NW)};
    int rsum = 0;
    bool need_lo = false;
    for (int b = 0; b < 3; ++b) {
        if (th[b] < 1 || tw[b] < 1 || th[b] > H || tw[b] > W) return false;
        int rb = 0;
        for (int o = 0; o < H; o += NW) {
            const Lerp2 a = lerp2(o, th[b], H), e = lerp2(o + NW - 1 < H - 1 ? o + NW - 1 : H - 1, th[b], H);
            if (e.i1 - a.i0 + 1 > rmax[b]) return false;
            rb = e.i1 - a.i0 + 1 > rb ? e.i1 - a.i0 + 1 : rb;
        }
        if (b > 0) rsum += rb;
        for (int o = 0; o < W; o += HTW) {
            const Lerp2 c = lerp2(o, tw[b], W), d = lerp2(o + HTW - 1 < W - 1 ? o + HTW - 1 : W - 1, tw[b], W);
            const int ws = b == 0 ? c.i0 : ((c.i0 + HT_PAD) & ~3) - HT_PAD;
            if (d.i1 - ws + 1 > (b == 0 ? 16 : 8)) return false;
        }
        // every product ly.l? * lx.l? must be a bf16 number, or the kernel carries U's lo part
        for (int y = 0; y < H && !need_lo; ++y) {
            const Lerp2 ly = lerp2(y, th[b], H);
            for (int x = 0; x < W; ++x) {
                const Lerp2 lx = lerp2(x, tw[b], W);
                const float wx0 = lx.i0 == lx.i1 ? lx.l0 + lx.l1 : lx.l0, wx1 = lx.l1;
                if (!bf16_exact(ly.l0 * wx0) || !bf16_exact(ly.l0 * wx1) || !bf16_exact(ly.l1 * wx0) ||
                    !bf16_exact(ly.l1 * wx1)) { need_lo = true; break; }
            }
        }
    }
    if (rsum * 128 > 2 * NW * 64) return false;      // t_2 + t_3 staging units: two per thread
    if (ulo) *ulo = need_lo;
    return true;
}

int launch_head2(const Head2Params& p, bool ulo, hipStream_t stream) {
    if (p.Ctp & 31) return (int)hipErrorInvalidValue;
    const int m3 = p.K <= 16 ? 1 : 2;
    const int key = (p.C0p / 32) * 100 + (p.C1p / 32) * 10 + m3;
    switch (key) {
        case 121: return launch_head2_u<1, 2, 1>(p, ulo, stream);
        case 122: return launch_head2_u<1, 2, 2>(p, ulo, stream);
        case 231: return launch_head2_u<2, 3, 1>(p, ulo, stream);
        case 232: return launch_head2_u<2, 3, 2>(p, ulo, stream);
    }
    return (int)hipErrorInvalidValue;
}

}  // namespace esa
