// head_x6.hip — the whole of last_layer[0..5] in one kernel for the fp32-grade mode (f32 NHWC tensors, bf16x6
// arithmetic, conv_x6.hip), never materialising the 480-channel tensors:
//
//   h0 = ReLU( W0·x0  +  sum_{b=1..3} bilinear_up(t_b)  +  bias0 )        (480 ch @ H/2)
//   h3 = ReLU( W3·h0 + bias3 )                                            (K   ch @ H/2)
//
// Replaces (reference): the three F.upsample + torch.cat of models/seg_hrnet.py:461-466 and last_layer[0..5]
// (1x1 480->480 + BN + ReLU, 1x1 480->K + BN + ReLU, :313-329).  The 1x1 convolution is pushed through the (linear)
// bilinear up-sampling: t_b = W_b·x_b is computed on branch b's own grid by conv_x6 and only interpolated here (f32
// VALU, ATen's align_corners=False weights); W_0 acts on the full-resolution branch directly.  Same structure as
// head_fused.hip (the split-bf16 mode's first-generation head) — one workgroup = 16x16 pixels = 16 waves, one 16-pixel
// row per wave, per 32-channel chunk of h0:
//   (1) a[2 M-tiles] = W0[chunk]·x0 — six MFMAs per tile and 32 input channels on the three exact bf16 terms of x0
//       (registers for the whole kernel) and of W0 (LDS-staged), low-order products first;
//   (2) + bias0 + the bilinear taps of t_1..t_3 from an LDS-staged f32 tile of this chunk;
//   (3) ReLU, exact 3-term split — the accumulator layout (lane = pixel, 4+4 consecutive channels) IS the B fragment of
//       conv_x6's K order (x6_chan_of_k), so h0 goes straight back into the matrix cores: six MFMAs per K-tile into a
//       FRESH accumulator, added to acc3 by one VALU add per chunk (tools/ubench/x6_numerics.hip: the rounding of a
//       long MFMA chain is what separates a bf16x6 sum from a blocked f32 sum);
// h0 never leaves the register file.  In the op-by-op plan these tensors cost 4 GB of HBM traffic per batch-32 step.
#include <type_traits>

#include "devstate.h"
#include "kernels.h"
#include "sb.h"

namespace esa {
namespace {

constexpr int HT = 16;                 // tile width (pixels)
constexpr int HTY = 8;                 // tile height: one row per wave, 8 waves (256 VGPRs each: the three-term fragments of
                                       // x0, W0, W3 and h0 do not fit the 128 of head_fused's 16-wave workgroup)
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
constexpr int PIXB = 144;              // LDS pixel pitch of the staged f32 t-tiles (128 B + 16 B pad)
constexpr int RMAX1 = 11, RMAX2 = 7, RMAX3 = 5;   // max source-region width per low-res branch (16 output columns)
constexpr int RMAY1 = 7, RMAY2 = 5, RMAY3 = 4;    // max source-region height (8 output rows)
constexpr int REG_PIX = RMAY1 * RMAX1 + RMAY2 * RMAX2 + RMAY3 * RMAX3;   // 132 pixels
constexpr int BUF_BYTES = REG_PIX * PIXB;                                // 28080 B per buffer

// a - b on two / four floats as packed instructions (hipcc lowers a vector fsub to one v_sub_f32 per element; the packed
// add takes the negation as a source modifier)
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) {
    f32x2 d;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ f32x4 pk_sub(f32x4 a, f32x4 b) {
    const f32x2 lo = pk_sub(f32x2{a[0], a[1]}, f32x2{b[0], b[1]}), hi = pk_sub(f32x2{a[2], a[3]}, f32x2{b[2], b[3]});
    return f32x4{lo[0], lo[1], hi[0], hi[1]};
}
// v0 + w * d on four floats as two packed FMAs
__device__ __forceinline__ f32x4 pk_lerp(f32x4 v0, float w, f32x4 d) {
    const f32x2 w2 = {w, w};
    const f32x2 lo = __builtin_elementwise_fma(f32x2{d[0], d[1]}, w2, f32x2{v0[0], v0[1]});
    const f32x2 hi = __builtin_elementwise_fma(f32x2{d[2], d[3]}, w2, f32x2{v0[2], v0[3]});
    return f32x4{lo[0], lo[1], hi[0], hi[1]};
}
// 8 floats (two accumulator quads: K order x6_chan_of_k) -> three bf16x8 fragments, v = t0 + t1 + t2 exactly
__device__ __forceinline__ void x6_split8(const float v[8], bf16x8 out[3]) {
    u32x4 t[3];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const f32x2 a = {v[2 * k], v[2 * k + 1]};
        const uint32_t h = __builtin_bit_cast(uint32_t, __builtin_convertvector(a, bf16x2));
        const f32x2 r = pk_sub(a, f32x2{__uint_as_float(h << 16), __uint_as_float(h & 0xffff0000u)});
        const uint32_t m = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, bf16x2));
        const f32x2 q = pk_sub(r, f32x2{__uint_as_float(m << 16), __uint_as_float(m & 0xffff0000u)});
        t[0][k] = h; t[1][k] = m; t[2][k] = __builtin_bit_cast(uint32_t, __builtin_convertvector(q, bf16x2));
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) out[i] = __builtin_bit_cast(bf16x8, t[i]);
}
// six products of one 16x16x32 tile pair into `d`: the five low-order ones first, a0*b0 last
__device__ __forceinline__ f32x4 x6_mma(const bf16x8 a[3], const bf16x8 b[3], f32x4 d) {
    d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], d, 0, 0, 0);
    d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], d, 0, 0, 0);
    d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], d, 0, 0, 0);
    d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], d, 0, 0, 0);
    d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], d, 0, 0, 0);
    d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], d, 0, 0, 0);
    return d;
}

struct LerpF {
    int i0, i1;
    float l0, l1;
};
__device__ __forceinline__ LerpF lerp_false(int dst, int in, int out) {     // align_corners=False
    const float scale = (float)in / (float)out;
    float src = scale * ((float)dst + 0.5f) - 0.5f;
    src = src < 0.f ? 0.f : src;
    LerpF r;
    r.i0 = min((int)src, in - 1);
    r.i1 = r.i0 + (r.i0 < in - 1 ? 1 : 0);
    r.l1 = src - (float)r.i0;
    r.l0 = 1.f - r.l1;
    return r;
}

constexpr int HTHREADS = 512;          // 8 waves

template <int NCH0, int M3>
__global__ __launch_bounds__(HTHREADS, 2) void head_x6_kernel(HeadParams p, int tiles_x, int tiles_y) {
    constexpr int WFR = 2 * NCH0 * 3 + M3 * 3;          // 1-KB weight fragments per chunk (W0 then W3), three terms each
    constexpr int WBYTES = WFR * 1024;
    constexpr int STRIDE = BUF_BYTES + WBYTES;          // one LDS buffer: t tiles, then weights
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int px = lane & 15, q = lane >> 4;
    int b_ = blockIdx.x;
    const int tx = b_ % tiles_x; b_ /= tiles_x;
    const int ty = b_ % tiles_y;
    const int n = b_ / tiles_y;
    const int oy0 = ty * HTY, ox0 = tx * HT;
    const int nchunks = p.Ctp >> 5;

    // ---- source regions of the three low-resolution terms (workgroup-uniform) ---------------
    int ry0[3], rx0[3], rh[3], rw[3], rbase[3];
    {
        int base = 0;
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const LerpF a = lerp_false(oy0, p.th[b], p.H), e = lerp_false(min(oy0 + HTY - 1, p.H - 1), p.th[b], p.H);
            const LerpF c = lerp_false(ox0, p.tw[b], p.W), d = lerp_false(min(ox0 + HT - 1, p.W - 1), p.tw[b], p.W);
            ry0[b] = a.i0; rh[b] = e.i1 - a.i0 + 1;
            rx0[b] = c.i0; rw[b] = d.i1 - c.i0 + 1;
            rbase[b] = base;
            base += rh[b] * rw[b];
        }
    }
    const int npix_stage = rbase[2] + rh[2] * rw[2];          // <= REG_PIX (validated on the host)

    // ---- staging map: unit u = it*1024 + tid -> staged pixel s = u>>3, 16-B piece j = u&7 -------
    constexpr int SIT = (REG_PIX * 8 + HTHREADS - 1) / HTHREADS;           // 2
    const char* sg[SIT];
#pragma unroll
    for (int it = 0; it < SIT; ++it) {
        const int s_ = it * (HTHREADS / 8) + (tid >> 3);
        sg[it] = nullptr;
        if (s_ < npix_stage) {
            const int b = s_ >= rbase[2] ? 2 : (s_ >= rbase[1] ? 1 : 0);
            const int r = s_ - rbase[b];
            const int yy = ry0[b] + r / rw[b], xx = rx0[b] + r % rw[b];
            sg[it] = p.t[b] + (((size_t)n * p.th[b] + yy) * p.tw[b] + xx) * (size_t)(p.Ctp * 4) + (tid & 7) * 16;
        }
    }
    const int lane_lds = (tid >> 3) * PIXB + (tid & 7) * 16;
    // weights of one chunk: fragments [W0 m0 c0 t0, t1, t2, ... | W3 m t0, t1, t2], 64 uint4 each
    constexpr int WIT = (WFR * 64 + HTHREADS - 1) / HTHREADS;     // weight units (16 B) per thread
    uint4 sr[SIT], wreg[WIT];
#define HEAD_PREFETCH(CH)                                                                     \
    {                                                                                         \
        _Pragma("unroll") for (int it = 0; it < SIT; ++it) {                                  \
            uint4 v = make_uint4(0, 0, 0, 0);                                                 \
            if (sg[it]) v = *reinterpret_cast<const uint4*>(sg[it] + (CH) * 128);             \
            sr[it] = v;                                                                       \
        }                                                                                     \
        _Pragma("unroll") for (int wi = 0; wi < WIT; ++wi) {                                  \
            const int wf = wi * (HTHREADS / 64) + (tid >> 6);      /* fragment index */        \
            if (wf < WFR) {                                                                   \
                const uint4* src = wf < 6 * NCH0                                              \
                    ? p.w0 + ((size_t)(CH) * 6 * NCH0 + wf) * 64 + lane                       \
                    : p.w3 + ((size_t)(((wf - 6 * NCH0) / 3) * nchunks + (CH)) * 3 + ((wf - 6 * NCH0) % 3)) * 64 + lane; \
                wreg[wi] = *src;                                                              \
            }                                                                                 \
        }                                                                                     \
    }
#define HEAD_COMMIT(BUF)                                                                      \
    {                                                                                         \
        _Pragma("unroll") for (int it = 0; it < SIT; ++it)                                    \
            if (sg[it]) *reinterpret_cast<uint4*>(smem + (BUF) * STRIDE + it * (HTHREADS / 8) * PIXB + lane_lds) = sr[it]; \
        _Pragma("unroll") for (int wi = 0; wi < WIT; ++wi)                                    \
            if (wi * (HTHREADS / 64) + (tid >> 6) < WFR)                                      \
                *reinterpret_cast<uint4*>(smem + (BUF) * STRIDE + BUF_BYTES + (wi * HTHREADS + tid) * 16) = wreg[wi]; \
    }

    // ---- per-lane constants: this wave's x0 fragment, interpolation coefficients ----------------
    const int ox = ox0 + px, oy = oy0 + wave;
    const bool in = ox < p.W && oy < p.H;
    bf16x8 xf[NCH0][3];
#pragma unroll
    for (int c = 0; c < NCH0; ++c) {
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (in) {       // the two quads of conv_x6's K order: channels 4q .. 4q+3 and 16+4q .. 16+4q+3 of the chunk
            const float* a = reinterpret_cast<const float*>(p.x0) + (((size_t)n * p.H + oy) * p.W + ox) * (size_t)p.C0p + c * 32 + q * 4;
            const f32x4 lo = *reinterpret_cast<const f32x4*>(a), hi = *reinterpret_cast<const f32x4*>(a + 16);
#pragma unroll
            for (int i = 0; i < 4; ++i) { v[i] = lo[i]; v[4 + i] = hi[i]; }
        }
        x6_split8(v, xf[c]);
    }
    int o00[3], o01[3], o10[3], o11[3];     // LDS byte offsets of the 4 taps (incl. region base, q*16)
    float w00[3], w01[3], w10[3], w11[3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        const LerpF lx = lerp_false(min(ox, p.W - 1), p.tw[b], p.W);
        const LerpF ly = lerp_false(min(oy, p.H - 1), p.th[b], p.H);
        const int r0 = (rbase[b] + (ly.i0 - ry0[b]) * rw[b]) * PIXB, r1 = (rbase[b] + (ly.i1 - ry0[b]) * rw[b]) * PIXB;
        const int c0 = (lx.i0 - rx0[b]) * PIXB + q * 16, c1 = (lx.i1 - rx0[b]) * PIXB + q * 16;
        o00[b] = r0 + c0; o01[b] = r0 + c1; o10[b] = r1 + c0; o11[b] = r1 + c1;
        w00[b] = ly.l0 * lx.l0; w01[b] = ly.l0 * lx.l1; w10[b] = ly.l1 * lx.l0; w11[b] = ly.l1 * lx.l1;
    }

    f32x4 acc3[M3];
#pragma unroll
    for (int m = 0; m < M3; ++m) acc3[m] = f32x4{0.f, 0.f, 0.f, 0.f};

    HEAD_PREFETCH(0)
    HEAD_COMMIT(0)
    __syncthreads();
    for (int cc = 0; cc < nchunks; ++cc) {
        const int buf = cc & 1;
        if (cc + 1 < nchunks) HEAD_PREFETCH(cc + 1)
        const char* tb = smem + buf * STRIDE;
        const char* wb = tb + BUF_BYTES + lane * 16;
        const f32x4 bias_lo = *reinterpret_cast<const f32x4*>(p.bias0 + cc * 32 + q * 4);
        const f32x4 bias_hi = *reinterpret_cast<const f32x4*>(p.bias0 + cc * 32 + 16 + q * 4);
        f32x4 a[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int c = 0; c < NCH0; ++c) {
                bf16x8 wa[3];
#pragma unroll
                for (int t = 0; t < 3; ++t) wa[t] = *reinterpret_cast<const bf16x8*>(wb + ((m * NCH0 + c) * 3 + t) * 1024);
                a[m] = x6_mma(wa, xf[c], a[m]);
            }
        a[0] += bias_lo;
        a[1] += bias_hi;
        f32x4 s[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const f32x4 v00 = *reinterpret_cast<const f32x4*>(tb + o00[b] + m * 64);
                const f32x4 v01 = *reinterpret_cast<const f32x4*>(tb + o01[b] + m * 64);
                const f32x4 v10 = *reinterpret_cast<const f32x4*>(tb + o10[b] + m * 64);
                const f32x4 v11 = *reinterpret_cast<const f32x4*>(tb + o11[b] + m * 64);
                s[m] += w00[b] * v00 + w01[b] * v01 + w10[b] * v10 + w11[b] * v11;
                // keep at most one (branch, M-tile)'s 4 taps in flight: 128-VGPR budget at 16 waves
                __builtin_amdgcn_sched_barrier(0);
            }
        float v[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[i] = relu1(a[0][i] + s[0][i]);
            v[4 + i] = relu1(a[1][i] + s[1][i]);
        }
        bf16x8 hf[3];
        x6_split8(v, hf);
#pragma unroll
        for (int m = 0; m < M3; ++m) {
            bf16x8 w3[3];
#pragma unroll
            for (int t = 0; t < 3; ++t) w3[t] = *reinterpret_cast<const bf16x8*>(wb + (6 * NCH0 + m * 3 + t) * 1024);
            acc3[m] += x6_mma(w3, hf, f32x4{0.f, 0.f, 0.f, 0.f});
        }
        if (cc + 1 < nchunks) {
            HEAD_COMMIT(buf ^ 1)          // nobody reads buffer buf^1 during this iteration
            __syncthreads();
        }
    }
#undef HEAD_PREFETCH
#undef HEAD_COMMIT

    // ---- epilogue: h3 = ReLU(acc3 + bias3) -> f32 [N][H][W][C3p] -----------------------------
    if (in) {
        float* o = reinterpret_cast<float*>(p.y) + (((size_t)n * p.H + oy) * p.W + ox) * (size_t)p.C3p;
#pragma unroll
        for (int m = 0; m < M3; ++m) {
            const int co = m * 16 + q * 4;
            const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias3 + co);
            f32x4 v;
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = relu1(acc3[m][i] + bv[i]);
            *reinterpret_cast<f32x4*>(o + co) = v;
        }
        for (int c = M3 * 16 + q * 4; c < p.C3p; c += 16)      // keep the padded channels exact zeros
            *reinterpret_cast<f32x4*>(o + c) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
}

// ---- second form: exact 2x / 4x / 8x grids (the network's own: branch b lives on H >> (b + 1)) ---------------------------
// One workgroup = 16x16 pixels = 4 waves; a lane owns a 2x2 pixel BLOCK (and, as before, 4 + 4 channels of the chunk):
// the four pixels of a block share their bilinear sources — 3x3 pixels of the 2x branch, 2x2 of the 4x and 8x branches,
// 17 LDS reads per 16 channels where one pixel per lane takes 12 each — and the W0 / W3 fragments read from LDS feed four
// 16-pixel MFMA column groups (group j = pixel (j & 1, j >> 1) of every block).  2.8x fewer LDS reads per pixel than the
// first form, which the LDS pipe bounds (33 ds_read_b128 per 16 pixels and chunk).  The interpolation is separable
// (v0 + l (v1 - v0) along x, then y: exact where the clamped border duplicates a source); sources beyond the grid are
// clamped while STAGING, so a lane's taps need no border cases.
constexpr int V2_SIDE0 = 10, V2_SIDE1 = 6, V2_SIDE2 = 4;                 // source-region sides of a 16x16 tile
constexpr int V2_PIX = V2_SIDE0 * V2_SIDE0 + V2_SIDE1 * V2_SIDE1 + V2_SIDE2 * V2_SIDE2;     // 152
constexpr int V2_BUF = 160 * PIXB;                                       // 5 staging rows of 32 pixels (152 used): 23040 B
constexpr int V2_THREADS = 256;
// compile-time loop: f(std::integral_constant<int, I>{}) for I in [B, E) (arrays indexed by a run-time loop counter next
// to a sched_barrier stay in scratch memory)
template <int B, int E, class F>
__device__ __forceinline__ void hx_static_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        hx_static_for<B + 1, E>(f);
    }
}

template <int NCH0, int M3>
__global__ __launch_bounds__(V2_THREADS, 2) void head_x6_v2_kernel(HeadParams p, int tiles_x, int tiles_y) {
    constexpr int WFR = 2 * NCH0 * 3 + M3 * 3;
    constexpr int WBYTES = ((WFR + 3) & ~3) * 1024;     // (whole rows of 4 fragments: the staging writes are unconditional)
    constexpr int STRIDE = V2_BUF + WBYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pxl = lane & 15, q = lane >> 4;
    int b_ = blockIdx.x;
    const int tx = b_ % tiles_x; b_ /= tiles_x;
    const int ty = b_ % tiles_y;
    const int n = b_ / tiles_y;
    const int oy0 = ty * 16, ox0 = tx * 16;
    const int nchunks = p.Ctp >> 5;
    constexpr int side[3] = {V2_SIDE0, V2_SIDE1, V2_SIDE2};
    constexpr int rbase[3] = {0, V2_SIDE0 * V2_SIDE0, V2_SIDE0 * V2_SIDE0 + V2_SIDE1 * V2_SIDE1};

    // ---- staging map: unit u = it*256 + tid -> staged pixel u >> 3, 16-byte piece u & 7 (sources clamped into the grid;
    // the 8 slots past the last region re-load its last pixel: every load and LDS write of the loop is unconditional)
    constexpr int SIT = (V2_PIX * 8 + V2_THREADS - 1) / V2_THREADS;       // 5
    uint32_t so[SIT];               // byte offset inside the unit's tensor t[b]
#pragma unroll
    for (int it = 0; it < SIT; ++it) {
        const int s_ = min(it * (V2_THREADS / 8) + (tid >> 3), V2_PIX - 1);
        const int b = s_ >= rbase[2] ? 2 : (s_ >= rbase[1] ? 1 : 0);
        const int sd = b == 2 ? V2_SIDE2 : (b == 1 ? V2_SIDE1 : V2_SIDE0);
        const int r = s_ - (b == 2 ? rbase[2] : (b == 1 ? rbase[1] : 0));
        const int yy = min(max((oy0 >> (b + 1)) - 1 + r / sd, 0), p.th[b] - 1);
        const int xx = min(max((ox0 >> (b + 1)) - 1 + r % sd, 0), p.tw[b] - 1);
        so[it] = (uint32_t)(((n * p.th[b] + yy) * p.tw[b] + xx) * (p.Ctp * 4) + (tid & 7) * 16);
    }
    const int lane_lds = (tid >> 3) * PIXB + (tid & 7) * 16;
    constexpr int WIT = (WFR * 64 + V2_THREADS - 1) / V2_THREADS;
    // The next chunk's tiles and weights travel through registers in TWO halves (units [0, UH) at the start of the
    // iteration, [UH, NU) in its middle; unit u < SIT: tile row u, else weight row u - SIT), each written to the other LDS
    // buffer — which nobody reads during this iteration — before the next is loaded: 16 registers in flight, not 32.
    // (rows 0..2 of the tile map are the 2x branch's; row 3 holds its last 4 pixels, then the 4x branch's; row 4 the 4x
    // branch's last 8, then the 8x branch's)
    constexpr int NU = SIT + WIT, UH = (NU + 1) / 2;
    u32x4 sr[UH];
    auto prefetch = [&](int ch, auto half_c) __attribute__((always_inline)) {
        constexpr int u0 = decltype(half_c)::value * UH;
        hx_static_for<u0, (u0 + UH < NU ? u0 + UH : NU)>([&](auto u_c) __attribute__((always_inline)) {
            constexpr int u = decltype(u_c)::value;
            if constexpr (u < SIT) {
                const int s_ = u * (V2_THREADS / 8) + (tid >> 3);
                const char* base = u < 3 ? p.t[0] : u == 3 ? (s_ < rbase[1] ? p.t[0] : p.t[1]) : (s_ < rbase[2] ? p.t[1] : p.t[2]);
                sr[u - u0] = *reinterpret_cast<const u32x4*>(base + so[u < SIT ? u : 0] + ch * 128);
            } else {
                const int wf = min((u - SIT) * (V2_THREADS / 64) + (tid >> 6), WFR - 1);
                const uint4* src = wf < 6 * NCH0
                    ? p.w0 + ((size_t)ch * 6 * NCH0 + wf) * 64 + lane
                    : p.w3 + ((size_t)(((wf - 6 * NCH0) / 3) * nchunks + ch) * 3 + ((wf - 6 * NCH0) % 3)) * 64 + lane;
                sr[u - u0] = *reinterpret_cast<const u32x4*>(src);
            }
        });
    };
    auto commit = [&](int buf, auto half_c) __attribute__((always_inline)) {
        constexpr int u0 = decltype(half_c)::value * UH;
        hx_static_for<u0, (u0 + UH < NU ? u0 + UH : NU)>([&](auto u_c) __attribute__((always_inline)) {
            constexpr int u = decltype(u_c)::value;
            if constexpr (u < SIT)
                *reinterpret_cast<u32x4*>(smem + buf * STRIDE + u * (V2_THREADS / 8) * PIXB + lane_lds) = sr[u - u0];
            else
                *reinterpret_cast<u32x4*>(smem + buf * STRIDE + V2_BUF + ((u - SIT) * V2_THREADS + tid) * 16) = sr[u - u0];
        });
    };
    constexpr auto HA = std::integral_constant<int, 0>{};
    constexpr auto HB = std::integral_constant<int, 1>{};
    float* const bias_s = reinterpret_cast<float*>(smem + 2 * STRIDE);      // bias0, whole (published by the first barrier)
    for (int i = tid; i < p.Ctp; i += V2_THREADS) bias_s[i] = p.bias0[i];

    // ---- this lane's block: pixels (X + dx, Y + dy), group j = dy*2 + dx ----------------------
    const int bx = pxl & 7, by = pxl >> 3;
    const int X = ox0 + 2 * bx, Y = oy0 + 4 * wave + 2 * by;
    bf16x8 xf[4][NCH0][3];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int ox = X + (j & 1), oy = Y + (j >> 1);
        const bool in = ox < p.W && oy < p.H;
#pragma unroll
        for (int c = 0; c < NCH0; ++c) {
            float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (in) {
                const float* a = reinterpret_cast<const float*>(p.x0) + (((size_t)n * p.H + oy) * p.W + ox) * (size_t)p.C0p + c * 32 + q * 4;
                const f32x4 lo = *reinterpret_cast<const f32x4*>(a), hi = *reinterpret_cast<const f32x4*>(a + 16);
#pragma unroll
                for (int i = 0; i < 4; ++i) { v[i] = lo[i]; v[4 + i] = hi[i]; }
            }
            x6_split8(v, xf[j][c]);
        }
    }
    // bilinear geometry (align_corners=False, scale exactly 2^-(b+1)): src = (dst + 0.5) / s - 0.5 = (2 dst + 1 - s) / 2s
    int toff[3];                    // LDS byte offset of the block's first source pixel (+ q*16)
    float lx0[3], ly0[3];           // weight of the second tap of pixel dx = 0 / dy = 0; the other pixel's: + 1/s (2x branch: 0.25
                                    // against the next source pair)
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        const int sh = b + 2;       // log2(2s)
        const int ny = 2 * Y + 1 - (2 << b), nx = 2 * X + 1 - (2 << b);
        const int yb = ny >> sh, xb = nx >> sh;     // floor (arithmetic shift); the dy = 1 / dx = 1 pixel: + (b == 0 ? 1 : 0)
        const float inv = 1.f / (float)(1 << sh);
        ly0[b] = (float)(ny - (yb << sh)) * inv;
        lx0[b] = (float)(nx - (xb << sh)) * inv;
        const int lr = yb - ((oy0 >> (b + 1)) - 1), lc = xb - ((ox0 >> (b + 1)) - 1);
        toff[b] = (rbase[b] + lr * side[b] + lc) * PIXB + q * 16;
    }

    f32x4 acc3[4][M3];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int m = 0; m < M3; ++m) acc3[j][m] = f32x4{0.f, 0.f, 0.f, 0.f};

    prefetch(0, HA);
    commit(0, HA);
    prefetch(0, HB);
    commit(0, HB);
    __syncthreads();
    for (int cc = 0; cc < nchunks; ++cc) {
        const int buf = cc & 1;
        const int nc = min(cc + 1, nchunks - 1);      // (the last iteration re-loads its own chunk: no branch around loads)
        prefetch(nc, HA);
        __builtin_amdgcn_sched_barrier(0);            // (left alone, hipcc sinks every load to the LDS write that consumes it)
        const char* tb = smem + buf * STRIDE;
        const char* wb = tb + V2_BUF + lane * 16;
        float v[4][8];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            if (m == 1) {
                __builtin_amdgcn_sched_barrier(0);
                commit(buf ^ 1, HA);
                prefetch(nc, HB);
                __builtin_amdgcn_sched_barrier(0);
            }
            const f32x4 bias = *reinterpret_cast<const f32x4*>(bias_s + cc * 32 + m * 16 + q * 4);
            f32x4 a[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) a[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < NCH0; ++c) {
                bf16x8 wa[3];
#pragma unroll
                for (int t = 0; t < 3; ++t) wa[t] = *reinterpret_cast<const bf16x8*>(wb + ((m * NCH0 + c) * 3 + t) * 1024);
#pragma unroll
                for (int j = 0; j < 4; ++j) a[j] = x6_mma(wa, xf[j][c], a[j]);
            }
            // up-sampled terms of the block's four pixels, 4 channels (m*16 + 4q ..)
            f32x4 s[4];
            {       // 2x branch: 3x3 sources; pixel (dx, dy) takes columns dx, dx+1 and rows dy, dy+1
                const char* t0 = tb + toff[0] + m * 64;
                f32x4 h[3][2];
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const f32x4 v0 = *reinterpret_cast<const f32x4*>(t0 + (r * V2_SIDE0) * PIXB);
                    const f32x4 v1 = *reinterpret_cast<const f32x4*>(t0 + (r * V2_SIDE0 + 1) * PIXB);
                    const f32x4 v2 = *reinterpret_cast<const f32x4*>(t0 + (r * V2_SIDE0 + 2) * PIXB);
                    h[r][0] = pk_lerp(v0, 0.75f, pk_sub(v1, v0));         // (X and Y are even: the 2x branch's weights are constants)
                    h[r][1] = pk_lerp(v1, 0.25f, pk_sub(v2, v1));
                }
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    s[dx] = pk_lerp(h[0][dx], 0.75f, pk_sub(h[1][dx], h[0][dx]));
                    s[2 + dx] = pk_lerp(h[1][dx], 0.25f, pk_sub(h[2][dx], h[1][dx]));
                }
            }
#pragma unroll
            for (int b = 1; b < 3; ++b) {       // 4x / 8x branches: the block's pixels share one 2x2 source cell
                const char* t0 = tb + toff[b] + m * 64;
                const int sd = b == 1 ? V2_SIDE1 : V2_SIDE2;
                const float step = b == 1 ? 0.25f : 0.125f;         // 1/s: the weight step between the block's two pixels
                f32x4 h[2][2];
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const f32x4 v0 = *reinterpret_cast<const f32x4*>(t0 + (r * sd) * PIXB);
                    const f32x4 v1 = *reinterpret_cast<const f32x4*>(t0 + (r * sd + 1) * PIXB);
                    const f32x4 d = pk_sub(v1, v0);
                    h[r][0] = pk_lerp(v0, lx0[b], d);
                    h[r][1] = pk_lerp(h[r][0], step, d);
                }
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const f32x4 d = pk_sub(h[1][dx], h[0][dx]);
                    const f32x4 o0 = pk_lerp(h[0][dx], ly0[b], d);
                    s[dx] += o0;
                    s[2 + dx] += pk_lerp(o0, step, d);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) v[j][m * 4 + i] = relu1((a[j][i] + bias[i]) + s[j][i]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bf16x8 hf[3];
            x6_split8(v[j], hf);
#pragma unroll
            for (int m = 0; m < M3; ++m) {
                bf16x8 w3[3];
#pragma unroll
                for (int t = 0; t < 3; ++t) w3[t] = *reinterpret_cast<const bf16x8*>(wb + (6 * NCH0 + m * 3 + t) * 1024);
                acc3[j][m] += x6_mma(w3, hf, f32x4{0.f, 0.f, 0.f, 0.f});
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        commit(buf ^ 1, HB);
        __syncthreads();
    }

#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int ox = X + (j & 1), oy = Y + (j >> 1);
        if (ox < p.W && oy < p.H) {
            float* o = reinterpret_cast<float*>(p.y) + (((size_t)n * p.H + oy) * p.W + ox) * (size_t)p.C3p;
#pragma unroll
            for (int m = 0; m < M3; ++m) {
                const int co = m * 16 + q * 4;
                const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias3 + co);
                f32x4 r;
#pragma unroll
                for (int i = 0; i < 4; ++i) r[i] = relu1(acc3[j][m][i] + bv[i]);
                *reinterpret_cast<f32x4*>(o + co) = r;
            }
            for (int c = M3 * 16 + q * 4; c < p.C3p; c += 16)
                *reinterpret_cast<f32x4*>(o + c) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
}

template <int NCH0, int M3>
int launch_head_x6_v2_t(const HeadParams& p, hipStream_t stream) {
    auto kern = head_x6_v2_kernel<NCH0, M3>;
    const int lds = 2 * (V2_BUF + ((2 * NCH0 * 3 + M3 * 3 + 3) & ~3) * 1024) + p.Ctp * 4;
    if (const int e_ = ensure_dyn_lds(reinterpret_cast<const void*>(kern), lds)) return e_;
    const int tiles_x = (p.W + 15) / 16, tiles_y = (p.H + 15) / 16;
    const long long nblk = (long long)p.N * tiles_x * tiles_y;
    if (nblk <= 0 || nblk > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(V2_THREADS), lds, stream, p, tiles_x, tiles_y);
    return (int)hipGetLastError();
}

// the second form's grids: branch b at exactly H >> (b + 1) x W >> (b + 1)
bool head_x6_exact_grids(const HeadParams& p) {
    for (int b = 0; b < 3; ++b) {
        if ((p.th[b] << (b + 1)) != p.H || (p.tw[b] << (b + 1)) != p.W) return false;
        if ((long long)p.N * p.th[b] * p.tw[b] * p.Ctp * 4 >= 0x7fffffffLL) return false;      // 32-bit staging offsets
    }
    return true;
}

template <int NCH0, int M3>
int launch_head_x6_t(const HeadParams& p, hipStream_t stream) {
    auto kern = head_x6_kernel<NCH0, M3>;
    const int lds = 2 * (BUF_BYTES + (2 * NCH0 * 3 + M3 * 3) * 1024);
    if (const int e_ = ensure_dyn_lds(reinterpret_cast<const void*>(kern), lds)) return e_;
    const int tiles_x = (p.W + HT - 1) / HT, tiles_y = (p.H + HTY - 1) / HTY;
    const long long nblk = (long long)p.N * tiles_x * tiles_y;
    if (nblk <= 0 || nblk > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(HTHREADS), lds, stream, p, tiles_x, tiles_y);
    return (int)hipGetLastError();
}

}  // namespace

static inline void hx6_lerp_host(int dst, int in, int out, int& i0, int& i1) {
    const float scale = (float)in / (float)out;
    float src = scale * ((float)dst + 0.5f) - 0.5f;
    src = src < 0.f ? 0.f : src;
    i0 = (int)src < in - 1 ? (int)src : in - 1;
    i1 = i0 + (i0 < in - 1 ? 1 : 0);
}

bool head_x6_supported(int H, int W, const int th[3], const int tw[3], int C0p, int K) {
    if (C0p != 32 && C0p != 64) return false;
    if (K < 1 || K > 32) return false;
    const int rmx[3] = {RMAX1, RMAX2, RMAX3}, rmy[3] = {RMAY1, RMAY2, RMAY3};
    for (int b = 0; b < 3; ++b) {
        for (int o = 0; o < H; o += HTY) {
            int a0, a1, e0, e1;
            hx6_lerp_host(o, th[b], H, a0, a1);
            hx6_lerp_host(o + HTY - 1 < H - 1 ? o + HTY - 1 : H - 1, th[b], H, e0, e1);
            if (e1 - a0 + 1 > rmy[b]) return false;
        }
        for (int o = 0; o < W; o += HT) {
            int a0, a1, e0, e1;
            hx6_lerp_host(o, tw[b], W, a0, a1);
            hx6_lerp_host(o + HT - 1 < W - 1 ? o + HT - 1 : W - 1, tw[b], W, e0, e1);
            if (e1 - a0 + 1 > rmx[b]) return false;
        }
    }
    return true;
}

// x0, t[b], y: f32 NHWC; w0 = pack_conv_weights_x6(k = 1) of last_layer[0]'s branch-0 slice [Ctp/16][C0p/32][3][64],
// w3 = pack_conv_weights_x6(k = 1) of last_layer[3] [M3][Ctp/32][3][64]; the source-region geometry is checked by
// head_x6_supported
int launch_head_x6(const HeadParams& p, hipStream_t stream) {
    if (p.Ctp & 31) return (int)hipErrorInvalidValue;
    const int m3 = p.K <= 16 ? 1 : 2;
    if (p.C3p < 16 * m3 || (p.C3p & 15)) return (int)hipErrorInvalidValue;
    if (head_x6_exact_grids(p)) {
        if (p.C0p == 32 && m3 == 1) return launch_head_x6_v2_t<1, 1>(p, stream);
        if (p.C0p == 32 && m3 == 2) return launch_head_x6_v2_t<1, 2>(p, stream);
    }
    if (p.C0p == 32 && m3 == 1) return launch_head_x6_t<1, 1>(p, stream);
    if (p.C0p == 32 && m3 == 2) return launch_head_x6_t<1, 2>(p, stream);
    if (p.C0p == 64 && m3 == 1) return launch_head_x6_t<2, 1>(p, stream);
    if (p.C0p == 64 && m3 == 2) return launch_head_x6_t<2, 2>(p, stream);
    return (int)hipErrorInvalidValue;
}

}  // namespace esa
