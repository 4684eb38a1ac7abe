// head_fused.hip — the whole of last_layer[0..5] in one kernel, never materialising the
// 480-channel tensors:
//
//   h0 = ReLU( W0·x0  +  sum_{b=1..3} bilinear_up(t_b)  +  bias0 )        (480 ch @ H/2)
//   h3 = ReLU( W3·h0 + bias3 )                                            (K   ch @ H/2)
//
// Replaces (reference): the three F.upsample + torch.cat of models/seg_hrnet.py:461-466 and
// last_layer[0..5] (1x1 480->480 + BN + ReLU, 1x1 480->K + BN + ReLU, :313-329).  The 1x1
// convolution is pushed through the (linear) bilinear up-sampling: t_b = W_b·x_b is computed
// on branch b's own grid by conv_mfma (f32 NHWC output) and only interpolated here; W_0 acts on
// the full-resolution branch directly.  See plan.hip for the algebra.
//
// One workgroup = 16x16 pixels = 16 waves, one 16-pixel row per wave (the kernel is VALU/LDS
// heavy — 12 bilinear taps per h0 element — so it wants many small waves, not fat ones).
// Per 32-channel chunk of h0:
//   (1) acc0[2 M-tiles] = bias0 + W0[chunk]·x0   — 6 split-bf16 MFMAs; the wave's x0 fragment
//       lives in registers for the whole kernel, the chunk's W0/W3 fragments are staged in LDS;
//   (2) acc0 += bilinear taps of t_1..t_3 read from an LDS-staged f32 tile of this chunk
//       (pixel pitch 144 B: conflict-free ds_read_b128 for the 2:1 source-pixel sharing);
//   (3) ReLU, split to hi/lo bf16 — the accumulator layout (lane = pixel, 4+4 consecutive
//       channels) IS a valid B-operand fragment under a permuted K order, so h0 goes straight
//       back into the matrix core:  acc3 += W3[chunk]·h0  (W3 is packed with the same K order);
// h0 never leaves the register file.  Staging of chunk c+1 (global -> registers) is issued
// before the math of chunk c and written to the other LDS buffer after it.
// Epilogue: bias3 + ReLU + split -> SB store of h3.
#include "devstate.h"
#include "kernels.h"
#include "sb.h"

namespace esa {
namespace {

constexpr int HT = 16;                 // tile edge (pixels)
constexpr int PIXB = 144;              // LDS pixel pitch of the staged f32 t-tiles (128 B + 16 B pad)
constexpr int RMAX1 = 11, RMAX2 = 7, RMAX3 = 5;   // max source-region edge per low-res branch
constexpr int REG_PIX = RMAX1 * RMAX1 + RMAX2 * RMAX2 + RMAX3 * RMAX3;   // 195 pixels
constexpr int BUF_BYTES = REG_PIX * PIXB;                                // 28080 B per buffer

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

constexpr int HTHREADS = 1024;         // 16 waves

template <int NCH0, int M3>
__global__ __launch_bounds__(HTHREADS, 1) void head_fused_kernel(HeadParams p, int tiles_x, int tiles_y) {
    constexpr int WFR = 2 * NCH0 * 2 + M3 * 2;          // 1-KB weight fragments per chunk (W0 then W3)
    constexpr int WBYTES = WFR * 1024;
    constexpr int STRIDE = BUF_BYTES + WBYTES;          // one LDS buffer: t tiles, then weights
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int px = lane & 15, q = lane >> 4;
    int b_ = blockIdx.x;
    const int tx = b_ % tiles_x; b_ /= tiles_x;
    const int ty = b_ % tiles_y;
    const int n = b_ / tiles_y;
    const int oy0 = ty * HT, ox0 = tx * HT;
    const int nchunks = p.Ctp >> 5;

    // ---- source regions of the three low-resolution terms (workgroup-uniform) ---------------
    int ry0[3], rx0[3], rh[3], rw[3], rbase[3];
    {
        int base = 0;
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const LerpF a = lerp_false(oy0, p.th[b], p.H), e = lerp_false(min(oy0 + HT - 1, p.H - 1), p.th[b], p.H);
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
    // weights of one chunk: fragments [W0 m0 c0 hi, lo, ... | W3 m hi, lo], 64 uint4 each
    const bool wthread = tid < WFR * 64;
    const int wf = tid >> 6;                                   // fragment index for weight staging
    uint4 sr[SIT], wreg;
#define HEAD_PREFETCH(CH)                                                                     \
    {                                                                                         \
        _Pragma("unroll") for (int it = 0; it < SIT; ++it) {                                  \
            uint4 v = make_uint4(0, 0, 0, 0);                                                 \
            if (sg[it]) v = *reinterpret_cast<const uint4*>(sg[it] + (CH) * 128);             \
            sr[it] = v;                                                                       \
        }                                                                                     \
        if (wthread) {                                                                        \
            const uint4* src = wf < 4 * NCH0                                                  \
                ? p.w0 + ((size_t)(CH) * 4 * NCH0 + wf) * 64 + lane                           \
                : p.w3 + ((size_t)(((wf - 4 * NCH0) >> 1) * nchunks + (CH)) * 2 + ((wf - 4 * NCH0) & 1)) * 64 + lane; \
            wreg = *src;                                                                      \
        }                                                                                     \
    }
#define HEAD_COMMIT(BUF)                                                                      \
    {                                                                                         \
        _Pragma("unroll") for (int it = 0; it < SIT; ++it)                                    \
            if (sg[it]) *reinterpret_cast<uint4*>(smem + (BUF) * STRIDE + it * (HTHREADS / 8) * PIXB + lane_lds) = sr[it]; \
        if (wthread) *reinterpret_cast<uint4*>(smem + (BUF) * STRIDE + BUF_BYTES + tid * 16) = wreg; \
    }

    // ---- per-lane constants: this wave's x0 fragment, interpolation coefficients ----------------
    const int ox = ox0 + px, oy = oy0 + wave;
    const bool in = ox < p.W && oy < p.H;
    bf16x8 xh[NCH0], xl[NCH0];
#pragma unroll
    for (int c = 0; c < NCH0; ++c) {
        uint4 h = make_uint4(0, 0, 0, 0), l = make_uint4(0, 0, 0, 0);
        if (in) {
            const char* a = p.x0 + (((size_t)n * p.H + oy) * p.W + ox) * (size_t)(p.C0p * 4) + c * 128 + q * 32;
            h = *reinterpret_cast<const uint4*>(a);
            l = *reinterpret_cast<const uint4*>(a + 16);
        }
        xh[c] = __builtin_bit_cast(bf16x8, h);
        xl[c] = __builtin_bit_cast(bf16x8, l);
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
        f32x4 a[2] = {bias_lo, bias_hi};
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int c = 0; c < NCH0; ++c) {
                const bf16x8 ah = *reinterpret_cast<const bf16x8*>(wb + ((m * NCH0 + c) * 2 + 0) * 1024);
                const bf16x8 al = *reinterpret_cast<const bf16x8*>(wb + ((m * NCH0 + c) * 2 + 1) * 1024);
                a[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, xh[c], a[m], 0, 0, 0);
                a[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, xl[c], a[m], 0, 0, 0);
                a[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, xh[c], a[m], 0, 0, 0);
            }
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
        uint4 hb, lb;
        split8(v, hb, lb);
        const bf16x8 hh = __builtin_bit_cast(bf16x8, hb), hl = __builtin_bit_cast(bf16x8, lb);
#pragma unroll
        for (int m = 0; m < M3; ++m) {
            const bf16x8 a3h = *reinterpret_cast<const bf16x8*>(wb + (4 * NCH0 + m * 2 + 0) * 1024);
            const bf16x8 a3l = *reinterpret_cast<const bf16x8*>(wb + (4 * NCH0 + m * 2 + 1) * 1024);
            acc3[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3l, hh, acc3[m], 0, 0, 0);
            acc3[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3h, hl, acc3[m], 0, 0, 0);
            acc3[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3h, hh, acc3[m], 0, 0, 0);
        }
        if (cc + 1 < nchunks) {
            HEAD_COMMIT(buf ^ 1)          // nobody reads buffer buf^1 during this iteration
            __syncthreads();
        }
    }
#undef HEAD_PREFETCH
#undef HEAD_COMMIT

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

template <int NCH0, int M3>
int launch_head_t(const HeadParams& p, hipStream_t stream) {
    auto kern = head_fused_kernel<NCH0, M3>;
    const int lds = 2 * (BUF_BYTES + (2 * NCH0 * 2 + M3 * 2) * 1024);
    if (const int e_ = ensure_dyn_lds(reinterpret_cast<const void*>(kern), lds)) return e_;
    const int tiles_x = (p.W + HT - 1) / HT, tiles_y = (p.H + HT - 1) / HT;
    const long long nblk = (long long)p.N * tiles_x * tiles_y;
    if (nblk <= 0 || nblk > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(HTHREADS), lds, stream, p, tiles_x, tiles_y);
    return (int)hipGetLastError();
}

// host mirror of lerp_false for the region-size validation
inline void lerp_host(int dst, int in, int out, int& i0, int& i1) {
    const float scale = (float)in / (float)out;
    float src = scale * ((float)dst + 0.5f) - 0.5f;
    src = src < 0.f ? 0.f : src;
    i0 = (int)src < in - 1 ? (int)src : in - 1;
    i1 = i0 + (i0 < in - 1 ? 1 : 0);
}

}  // namespace

bool head_fused_supported(int H, int W, const int th[3], const int tw[3], int C0p, int K) {
    if (C0p != 32 && C0p != 64) return false;
    if (K < 1 || K > 32) return false;
    const int rmax[3] = {RMAX1, RMAX2, RMAX3};
    for (int b = 0; b < 3; ++b) {
        for (int o = 0; o < H; o += HT) {
            int a0, a1, e0, e1;
            lerp_host(o, th[b], H, a0, a1);
            lerp_host(o + HT - 1 < H - 1 ? o + HT - 1 : H - 1, th[b], H, e0, e1);
            if (e1 - a0 + 1 > rmax[b]) return false;
        }
        for (int o = 0; o < W; o += HT) {
            int a0, a1, e0, e1;
            lerp_host(o, tw[b], W, a0, a1);
            lerp_host(o + HT - 1 < W - 1 ? o + HT - 1 : W - 1, tw[b], W, e0, e1);
            if (e1 - a0 + 1 > rmax[b]) return false;
        }
    }
    return true;
}

int launch_head(const HeadParams& p, hipStream_t stream) {
    if (p.Ctp & 31) return (int)hipErrorInvalidValue;
    const int m3 = p.K <= 16 ? 1 : 2;
    if (p.C0p == 32 && m3 == 1) return launch_head_t<1, 1>(p, stream);
    if (p.C0p == 32 && m3 == 2) return launch_head_t<1, 2>(p, stream);
    if (p.C0p == 64 && m3 == 1) return launch_head_t<2, 1>(p, stream);
    if (p.C0p == 64 && m3 == 2) return launch_head_t<2, 2>(p, stream);
    return (int)hipErrorInvalidValue;
}

// W3 [K][Ct] (1x1) -> [M3][Ctp/32][hi|lo][lane][8] with the permuted K order of the h0 fragment:
// lane (r = l&15, g = l>>4), element j  <->  channel chunk*32 + (j < 4 ? 4g + j : 16 + 4g + j - 4)
static inline uint16_t hb16(float f) {
    uint32_t u;
    __builtin_memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static inline float hb16f(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
}
size_t head_w3_bytes(int K, int Ctp) { return (size_t)(K <= 16 ? 1 : 2) * (Ctp / 32) * 2048; }
void pack_head_w3(const float* w, int K, int Ct, int Ctp, void* dst) {
    uint16_t* d = static_cast<uint16_t*>(dst);
    const int m3 = K <= 16 ? 1 : 2, nch = Ctp / 32;
    for (int m = 0; m < m3; ++m)
        for (int c = 0; c < nch; ++c)
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 8; ++j) {
                    const int r = m * 16 + (l & 15), g = l >> 4;
                    const int ch = c * 32 + (j < 4 ? 4 * g + j : 16 + 4 * g + j - 4);
                    float v = 0.f;
                    if (r < K && ch < Ct) v = w[(size_t)r * Ct + ch];
                    const uint16_t hi = hb16(v), lo = hb16(v - hb16f(hi));
                    const size_t base = (((size_t)m * nch + c) * 2) * 512;
                    d[base + l * 8 + j] = hi;
                    d[base + 512 + l * 8 + j] = lo;
                }
}

}  // namespace esa
