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
#include "kernels.h"
#include "sb.h"

namespace esa {
namespace {

constexpr int HT = 16;
constexpr int H2THREADS = 1024;
constexpr int R1 = 11, R2 = 7, R3 = 5;          // max source rows per branch and tile
constexpr int T1OFF = 0;
constexpr int T2OFF = R1 * 2 * 1024;
constexpr int T3OFF = T2OFF + R2 * 1024;
constexpr int TBUF = T3OFF + R3 * 1024;         // 34816 B

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

template <int NCH0, int NCH1, int M3, bool ULO>
__global__ __launch_bounds__(H2THREADS, 1) void head_fused2_kernel(Head2Params p, int tiles_x, int tiles_y) {
    constexpr int W03FR = 4 * NCH0 + 2 * M3;             // W0 then W3 fragments of one chunk
    constexpr int W03S = W03FR * 1024 + 256;             // + 32 bias floats
    constexpr int W1FR = 4 * NCH1;
    constexpr int W1S = W1FR * 1024;
    static_assert(W03FR < 16 && W1FR <= 16, "one weight fragment per wave");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const tb0 = smem;
    char* const w03b = smem + 2 * TBUF;
    char* const w1b = w03b + 2 * W03S;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int px = lane & 15, q = lane >> 4;
    int b_ = blockIdx.x;
    const int tx = b_ % tiles_x; b_ /= tiles_x;
    const int ty = b_ % tiles_y;
    const int n = b_ / tiles_y;
    const int oy0 = ty * HT, ox0 = tx * HT;
    const int nchunks = p.Ctp >> 5;

    // ---- source windows of the three low-resolution branches (workgroup-uniform) ---------------
    int ry0[3], rh[3], ws[3], rw0;
    {
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const Lerp2 a = lerp2(oy0, p.th[b], p.H), e = lerp2(min(oy0 + HT - 1, p.H - 1), p.th[b], p.H);
            const Lerp2 c = lerp2(ox0, p.tw[b], p.W);
            ry0[b] = a.i0; rh[b] = e.i1 - a.i0 + 1;
            // window slot 0 in source columns: branch 1 starts at the first needed column, branches
            // 2/3 at the 8-byte aligned stored column below it (stored column = x + HT_PAD)
            ws[b] = b == 0 ? c.i0 : ((c.i0 + HT_PAD) & ~3) - HT_PAD;
        }
        const Lerp2 d = lerp2(min(ox0 + HT - 1, p.W - 1), p.tw[0], p.W);
        rw0 = d.i1 - ws[0] + 1;
    }

    // ---- staging map of the t_2 / t_3 windows: unit = (row, part, channel, 4-pixel half) ----------
    const char* sg[2];
    int sstride[2], sdst[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int u = it * H2THREADS + tid;
        const int n2 = rh[1] * 128, n3 = rh[2] * 128;
        sg[it] = nullptr; sstride[it] = 0; sdst[it] = 0;
        if (u < n2 + n3) {
            const int b = u < n2 ? 1 : 2;
            const int v = u < n2 ? u : u - n2;
            const int r = v >> 7, part = (v >> 6) & 1, ch = (v >> 1) & 31, half = v & 1;
            const int xp = b == 1 ? p.xp2 : p.xp3;
            const char* base = b == 1 ? p.t2 : p.t3;
            sg[it] = base + ((((size_t)n * p.th[b] + ry0[b] + r) * nchunks * 2 + part) * 32 + ch) * (size_t)(xp * 2)
                     + (size_t)(ws[b] + HT_PAD + half * 4) * 2;
            sstride[it] = 128 * xp;                      // bytes between chunks: 2 parts x 32 ch x XP x 2 B
            sdst[it] = (b == 1 ? T2OFF : T3OFF) + v * 8;
        }
    }
    uint2 sr[2];
    uint4 wreg, w1reg;
    // chunk CH: t_2/t_3 windows + W0/W3/bias0; chunk CH1: W_1 fragments
#define H2_PREFETCH(CH, CH1)                                                                  \
    {                                                                                         \
        if ((CH) < nchunks) {                                                                 \
            _Pragma("unroll") for (int it = 0; it < 2; ++it) {                                \
                uint2 v = make_uint2(0, 0);                                                   \
                if (sg[it]) v = *reinterpret_cast<const uint2*>(sg[it] + (size_t)(CH) * sstride[it]); \
                sr[it] = v;                                                                   \
            }                                                                                 \
            if (wave < 4 * NCH0)                                                              \
                wreg = p.w0[((size_t)(CH) * 4 * NCH0 + wave) * 64 + lane];                    \
            else if (wave < W03FR)                                                            \
                wreg = p.w3[((size_t)(((wave - 4 * NCH0) >> 1) * nchunks + (CH)) * 2 + ((wave - 4 * NCH0) & 1)) * 64 + lane]; \
            else if (wave == W03FR && lane < 8)                                               \
                wreg = *reinterpret_cast<const uint4*>(p.bias0 + (CH) * 32 + lane * 4);       \
        }                                                                                     \
        if ((CH1) < nchunks && wave < W1FR) w1reg = p.w1[((size_t)(CH1) * W1FR + wave) * 64 + lane]; \
    }
#define H2_COMMIT(CH, CH1)                                                                    \
    {                                                                                         \
        if ((CH) < nchunks) {                                                                 \
            _Pragma("unroll") for (int it = 0; it < 2; ++it)                                  \
                if (sg[it]) *reinterpret_cast<uint2*>(tb0 + ((CH) & 1) * TBUF + sdst[it]) = sr[it]; \
            if (wave < W03FR || (wave == W03FR && lane < 8))                                  \
                *reinterpret_cast<uint4*>(w03b + ((CH) & 1) * W03S + tid * 16) = wreg;        \
        }                                                                                     \
        if ((CH1) < nchunks && wave < W1FR)                                                   \
            *reinterpret_cast<uint4*>(w1b + ((CH1) & 1) * W1S + tid * 16) = w1reg;            \
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
    if (t1wave) {                                                                             \
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

    // ---- prologue: W_1(0); then chunk 0's data + W_1(1) while t_1(0) is computed -------------------
    H2_PREFETCH(nchunks, 0)
    H2_COMMIT(nchunks, 0)
    __syncthreads();
    H2_PREFETCH(0, 1)
    H2_T1(0)
    H2_COMMIT(0, 1)
    __syncthreads();

    for (int cc = 0; cc < nchunks; ++cc) {
        H2_PREFETCH(cc + 1, cc + 2)
        const char* tb = tb0 + (cc & 1) * TBUF;
        const char* wb = w03b + (cc & 1) * W03S;
        f32x4 a[2];
        a[0] = *reinterpret_cast<const f32x4*>(wb + W03FR * 1024 + q * 16);
        a[1] = *reinterpret_cast<const f32x4*>(wb + W03FR * 1024 + 64 + q * 16);
        // (1) W0·x0
#pragma unroll
        for (int m = 0; m < 2; ++m)
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
        for (int kc = 0; kc < 2; ++kc)
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
        for (int m = 0; m < M3; ++m) {
            const bf16x8 a3h = *reinterpret_cast<const bf16x8*>(wb + lane * 16 + (4 * NCH0 + m * 2 + 0) * 1024);
            const bf16x8 a3l = *reinterpret_cast<const bf16x8*>(wb + lane * 16 + (4 * NCH0 + m * 2 + 1) * 1024);
            acc3[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3l, hh, acc3[m], 0, 0, 0);
            acc3[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3h, hl, acc3[m], 0, 0, 0);
            acc3[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3h, hh, acc3[m], 0, 0, 0);
        }
        if (cc + 1 < nchunks) {
            H2_T1(cc + 1)
            H2_COMMIT(cc + 1, cc + 2)       // nobody reads T/W03 buffer (cc+1)&1 or W_1 buffer cc&1 now
            __syncthreads();
        }
    }
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

template <int NCH0, int NCH1, int M3, bool ULO>
int launch_head2_t(const Head2Params& p, hipStream_t stream) {
    auto kern = head_fused2_kernel<NCH0, NCH1, M3, ULO>;
    static bool attr_set = false;
    const int lds = 2 * TBUF + 2 * ((4 * NCH0 + 2 * M3) * 1024 + 256) + 2 * 4 * NCH1 * 1024;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const int tiles_x = (p.W + HT - 1) / HT, tiles_y = (p.H + HT - 1) / HT;
    const long long nblk = (long long)p.N * tiles_x * tiles_y;
    if (nblk <= 0 || nblk > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(H2THREADS), lds, stream, p, tiles_x, tiles_y);
    return (int)hipGetLastError();
}

template <int NCH0, int NCH1, int M3>
int launch_head2_u(const Head2Params& p, bool ulo, hipStream_t stream) {
    return ulo ? launch_head2_t<NCH0, NCH1, M3, true>(p, stream) : launch_head2_t<NCH0, NCH1, M3, false>(p, stream);
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
    const int rmax[3] = {R1, R2, R3};
    bool need_lo = false;
    for (int b = 0; b < 3; ++b) {
        if (th[b] < 1 || tw[b] < 1 || th[b] > H || tw[b] > W) return false;
        for (int o = 0; o < H; o += HT) {
            const Lerp2 a = lerp2(o, th[b], H), e = lerp2(o + HT - 1 < H - 1 ? o + HT - 1 : H - 1, th[b], H);
            if (e.i1 - a.i0 + 1 > rmax[b]) return false;
        }
        for (int o = 0; o < W; o += HT) {
            const Lerp2 c = lerp2(o, tw[b], W), d = lerp2(o + HT - 1 < W - 1 ? o + HT - 1 : W - 1, tw[b], W);
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
