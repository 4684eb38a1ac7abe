// head_fused_bf.hip — last_layer[0..5] in one kernel for the single-pass bf16 mode (esahrnet_cfg.precision = 1),
// the 720-channel (W48) / 480-channel (W32) tensors of models/seg_hrnet.py:461-468, 313-329 never materialised:
//
//   h0 = ReLU( W0·x0 + bias0 + sum_{b=1..3} bilinear_up(t_b) )      (Ct channels @ H/2, registers only)
//   h3 = ReLU( W3·h0 + bias3 )                                      (K  channels @ H/2, stored as bf16)
//
// t_b = W_b·x_b are the last_layer[0] slices evaluated on branch b's own grid by the bf16 1x1 kernel (a 1x1
// convolution commutes with bilinear interpolation, plan.hip); they arrive as BF tensors [N][th][tw][Ctp].
// Same scheme as head_fused.hip (split format), with one MFMA per product and 2-byte operands:
// one workgroup = 8 rows x 16 pixels = 8 waves, one row of 16 pixels per wave (16 waves of 128 VGPRs spilled 70
// registers; 8 waves have 256 each).  Per 32-channel chunk of h0:
//   (1) acc[2 cout tiles] = bias0 + W0[chunk]·x0: the wave's x0 fragments (both K-steps of every 64-channel block)
//       live in registers for the whole kernel, the chunk's W0 / W3 fragments are staged in LDS;
//   (2) += 4 bilinear taps x 3 branches, read as 8-byte (4-channel) pieces from an LDS-staged bf16 tile of the chunk
//       (80-byte pixel pitch) and widened to f32 by a shift;
//   (3) ReLU, one rounding to bf16: the accumulator layout (lane = pixel, 4 + 4 consecutive channels) IS a B-operand
//       fragment under the permuted K order pack_head_w3_bf gives W3, so h0 goes straight back into the matrix core.
// Staging of chunk c+1 (global -> registers) is issued before the math of chunk c and written to the other LDS buffer
// after it.  Unpinned against the reference by itself; the bf16 network tests cover it (tests/test_gpu_bf16.py).
#include <algorithm>

#include "devstate.h"
#include "kernels.h"
#include "sb.h"

namespace esa {
namespace {

constexpr int BHT = 16;                 // tile width (pixels)
constexpr int BHY = 8;                  // tile height (rows = waves)
constexpr int BPIX = 80;                // LDS pixel pitch of the staged bf16 t-tiles (64 B + 16 B pad)
constexpr int BR1 = 11, BR2 = 7, BR3 = 5;           // max source-region edge per low-resolution branch
constexpr int BREG = BR1 * BR1 + BR2 * BR2 + BR3 * BR3;   // 195 pixels
constexpr int BBUF = ((BREG * BPIX + 255) / 256) * 256;
constexpr int BTHREADS = 64 * BHY;
// MF (matrix-core interpolation, the default): the staged t-tiles are kept TRANSPOSED, T[branch][source row][channel 0..31]
// [source column], 16 / 8 / 8 columns per row for the three branches, so that 8 consecutive source columns of one channel
// are one 16-byte A-operand piece; see the kernel
constexpr int T1OFF = 0, T2OFF = BR1 * 32 * 16 * 2, T3OFF = T2OFF + BR2 * 32 * 8 * 2, TBYTES = T3OFF + BR3 * 32 * 8 * 2;     // 17 408 B
static_assert(TBYTES % 256 == 0 && BR1 <= 16 && BR2 <= 8 && BR3 <= 8, "T layout");

struct LerpB {
    int i0, i1;
    float l0, l1;
};
__device__ __host__ inline LerpB lerp_false_b(int dst, int in, int out) {     // align_corners=False (ATen)
#pragma clang fp contract(off)      // host (window limits) and device must see the same source indices: no fma here
    const float scale = (float)in / (float)out;
    float src = scale * ((float)dst + 0.5f) - 0.5f;
    src = src < 0.f ? 0.f : src;
    LerpB r;
    r.i0 = (int)src < in - 1 ? (int)src : in - 1;
    r.i1 = r.i0 + (r.i0 < in - 1 ? 1 : 0);
    r.l1 = src - (float)r.i0;
    r.l0 = 1.f - r.l1;
    return r;
}

// value of lane ^ 4 (bit 2 of the lane = `hi`): two DPP row shifts and a select — no trip through the LDS crossbar, which
// the kernel already saturates (a __shfl_xor is a ds_bpermute)
__device__ __forceinline__ uint32_t lane_xor4(uint32_t v, bool hi) {
    const uint32_t up = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x104, 0xf, 0xf, true);      // row_shl:4: lane i <- lane i + 4
    const uint32_t dn = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);      // row_shr:4: lane i <- lane i - 4
    return hi ? dn : up;
}

// MF: sum_b U_b·t_b on the matrix cores.  Bilinear interpolation is linear over source pixels and one output row draws on two
// source rows per branch, so for the wave's row of 16 pixels  sum_b U_b·t_b = T[ch][64 slots] · U[64 slots][16 px], slots =
// 2 rows x (16 + 8 + 8) source columns = two K-steps (the scheme of head_fused2.hip for the split format).  U depends on the
// output row and column only: built once per wave, kept in registers, exact in bf16 for the 2x / 4x / 8x grids (numerators
// < 256); other ratios carry a lo part (`ulo`: two more MFMAs).  The accumulator is the one W0·x0 already uses.  Replaces 96
// multiply-adds + 96 bf16 widenings per lane and chunk (the kernel was VALU-bound: 2.65 ms at W48 384x384 batch 64).
template <int NB0, int M3, bool MF>
__global__ __launch_bounds__(BTHREADS, MF ? 4 : 1) void head_fused_bf_kernel(HeadParams p, int tiles_x, int tiles_y, int ulo) {
    constexpr int WFR = 2 * NB0 * 2 + M3;               // 1-KB weight fragments per chunk: W0 [m][block][step], W3 [m]
    constexpr int WBYTES = WFR * 1024;
    constexpr int TB = MF ? TBYTES : BBUF;
    constexpr int STRIDE = TB + WBYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int px = lane & 15, q = lane >> 4;
    int b_ = xcd_contiguous(blockIdx.x, gridDim.x);
    const int tx = b_ % tiles_x; b_ /= tiles_x;
    const int ty = b_ % tiles_y;
    const int n = b_ / tiles_y;
    const int oy0 = ty * BHY, ox0 = tx * BHT;
    const int nchunks = p.Ctp >> 5;

    int ry0[3], rx0[3], rh[3], rw[3], rbase[3];
    {
        int base = 0;
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const LerpB a = lerp_false_b(oy0, p.th[b], p.H), e = lerp_false_b(min(oy0 + BHY - 1, p.H - 1), p.th[b], p.H);
            const LerpB c = lerp_false_b(ox0, p.tw[b], p.W), d = lerp_false_b(min(ox0 + BHT - 1, p.W - 1), p.tw[b], p.W);
            ry0[b] = a.i0; rh[b] = e.i1 - a.i0 + 1;
            rx0[b] = c.i0; rw[b] = d.i1 - c.i0 + 1;
            rbase[b] = base;
            base += rh[b] * rw[b];
        }
    }
    const int npix_stage = rbase[2] + rh[2] * rw[2];          // <= BREG (validated on the host)

    // staging: thread -> (staged pixel s = tid >> 2, 16-byte piece j = tid & 3 of the chunk's 64 bytes)
    const char* sg = nullptr;
    bool twrite = false;                        // MF: this lane writes its pair's four dwords
    int pbase[3] = {0, 0, 0}, npairs = 0;       // MF: column pairs per branch region
    if (MF) {
#pragma unroll
        for (int b = 0; b < 3; ++b) { pbase[b] = npairs; npairs += rh[b] * ((rw[b] + 1) >> 1); }
    }
    if (!MF) {
        const int s_ = tid >> 2;
        if (s_ < npix_stage) {
            const int b = s_ >= rbase[2] ? 2 : (s_ >= rbase[1] ? 1 : 0);
            const int r = s_ - rbase[b];
            const int yy = ry0[b] + r / rw[b], xx = rx0[b] + r % rw[b];
            sg = p.t[b] + (((size_t)n * p.th[b] + yy) * p.tw[b] + xx) * (size_t)(p.Ctp * 2) + (tid & 3) * 16;
        }
    }
    int lane_lds = (tid >> 2) * BPIX + (tid & 3) * 16;
    int tstride = 0;                            // MF: bytes between consecutive channels of the thread's staged pixel
    if (MF) {
        // thread = (column pair u = tid >> 3, column of the pair hc = bit 2, 16-byte piece j = tid & 3 = channels 8j .. 8j+7):
        // the two lanes of a pair swap halves (commit) so that each writes FOUR dwords [even column | odd column] — channels
        // 8j .. 8j+3 by the even-column lane, 8j+4 .. 8j+7 by the odd-column one — instead of eight 2-byte pieces
        const int u = tid >> 3, hc = (tid >> 2) & 1;
        if (u < npairs) {
            const int b = u >= pbase[2] ? 2 : (u >= pbase[1] ? 1 : 0);
            const int r = u - pbase[b], pw = (rw[b] + 1) >> 1;
            const int row = r / pw, cp = r % pw, col = 2 * cp + hc;
            const int cw = b == 0 ? 16 : 8;
            tstride = cw * 2;
            twrite = true;
            lane_lds = (b == 0 ? T1OFF : b == 1 ? T2OFF : T3OFF) + ((row * 32 + (tid & 3) * 8 + 4 * hc) * cw + 2 * cp) * 2;
            if (col < rw[b])
                sg = p.t[b] + (((size_t)n * p.th[b] + ry0[b] + row) * p.tw[b] + rx0[b] + col) * (size_t)(p.Ctp * 2) + (tid & 3) * 16;
        }
        // columns beyond a window keep their zeros (their U entries are zero, but 0 x stale NaN bits would not be)
        for (int i = tid * 16; i < 2 * STRIDE; i += BTHREADS * 16)
            if (i % STRIDE < TB) *reinterpret_cast<uint4*>(smem + i) = make_uint4(0, 0, 0, 0);
        __syncthreads();
    }
    constexpr int WIT = (WFR * 64 + BTHREADS - 1) / BTHREADS;     // weight staging iterations (one fragment per wave each)
    uint4 sr, wreg[WIT];
#define HB_PREFETCH(CH)                                                                       \
    {                                                                                         \
        sr = make_uint4(0, 0, 0, 0);                                                          \
        if (sg) sr = *reinterpret_cast<const uint4*>(sg + (CH) * 64);                         \
        _Pragma("unroll") for (int wi = 0; wi < WIT; ++wi) {                                  \
            const int wf = wi * BHY + wave;                                                   \
            if (wf < WFR) {                                                                   \
                const uint4* src = wf < 4 * NB0                                               \
                    ? p.w0 + ((size_t)(CH) * 4 * NB0 + wf) * 64 + lane                        \
                    : p.w3 + ((size_t)(wf - 4 * NB0) * nchunks + (CH)) * 64 + lane;           \
                wreg[wi] = *src;                                                              \
            }                                                                                 \
        }                                                                                     \
    }
#define HB_COMMIT(BUF)                                                                        \
    {                                                                                         \
        if (MF) {                                                                             \
            const bool hc_ = (tid >> 2) & 1;                                                  \
            const uint32_t r0_ = lane_xor4(hc_ ? sr.x : sr.z, hc_), r1_ = lane_xor4(hc_ ? sr.y : sr.w, hc_);    \
            const uint32_t e0_ = hc_ ? r0_ : sr.x, e1_ = hc_ ? r1_ : sr.y;                    \
            const uint32_t o0_ = hc_ ? sr.z : r0_, o1_ = hc_ ? sr.w : r1_;                    \
            if (twrite) {                                                                     \
                char* d_ = smem + (BUF) * STRIDE + lane_lds;                                  \
                *reinterpret_cast<uint32_t*>(d_) = (e0_ & 0xffffu) | (o0_ << 16);             \
                *reinterpret_cast<uint32_t*>(d_ + tstride) = (e0_ >> 16) | (o0_ & 0xffff0000u);       \
                *reinterpret_cast<uint32_t*>(d_ + 2 * tstride) = (e1_ & 0xffffu) | (o1_ << 16);       \
                *reinterpret_cast<uint32_t*>(d_ + 3 * tstride) = (e1_ >> 16) | (o1_ & 0xffff0000u);   \
            }                                                                                 \
        } else if (sg) *reinterpret_cast<uint4*>(smem + (BUF) * STRIDE + lane_lds) = sr;      \
        _Pragma("unroll") for (int wi = 0; wi < WIT; ++wi)                                    \
            if (wi * BHY + wave < WFR)                                                        \
                *reinterpret_cast<uint4*>(smem + (BUF) * STRIDE + TB + ((wi * BHY + wave) * 64 + lane) * 16) = wreg[wi]; \
    }

    const int ox = ox0 + px, oy = oy0 + wave;
    const bool in = ox < p.W && oy < p.H;
    bf16x8 x0[NB0], x1[NB0];                      // K-step 0 / 1 of every 64-channel block of branch 0
#pragma unroll
    for (int c = 0; c < NB0; ++c) {
        uint4 h = make_uint4(0, 0, 0, 0), l = make_uint4(0, 0, 0, 0);
        if (in) {
            const char* a = p.x0 + (((size_t)n * p.H + oy) * p.W + ox) * (size_t)(p.C0p * 2) + c * 128 + q * 16;
            h = *reinterpret_cast<const uint4*>(a);
            l = *reinterpret_cast<const uint4*>(a + 64);
        }
        x0[c] = __builtin_bit_cast(bf16x8, h);
        x1[c] = __builtin_bit_cast(bf16x8, l);
    }
    // MF: interpolation operand U (B operand: lane = pixel px, k-group q = 8 slots) and the lane's T pieces (A operand:
    // lane & 15 = channel, k-group q).  K-step 0 = t_1: q = 2 * (row 0/1) + (columns 0-7 / 8-15);  K-step 1: q = 0/1 t_2 row
    // 0/1, q = 2/3 t_3 row 0/1, columns 0-7.
    bf16x8 u0 = {}, u1 = {}, u0l = {}, u1l = {};
    int toff0 = 0, toff1 = 0;
    if (MF) {
        float wa[8], wb_[8];
        const int ch = lane & 15;
        {
            const LerpB lx = lerp_false_b(min(ox, p.W - 1), p.tw[0], p.W), ly = lerp_false_b(min(oy, p.H - 1), p.th[0], p.H);
            const float wr = (q >> 1) ? ly.l1 : ly.l0;
            const int row = ((q >> 1) ? ly.i1 : ly.i0) - ry0[0];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = (q & 1) * 8 + j + rx0[0];
                wa[j] = wr * ((c == lx.i0 ? lx.l0 : 0.f) + (c == lx.i1 ? lx.l1 : 0.f));
            }
            toff0 = T1OFF + ((row * 32 + ch) * 16 + (q & 1) * 8) * 2;
        }
        {
            const int b = 1 + (q >> 1);
            const LerpB lx = lerp_false_b(min(ox, p.W - 1), p.tw[b], p.W), ly = lerp_false_b(min(oy, p.H - 1), p.th[b], p.H);
            const float wr = (q & 1) ? ly.l1 : ly.l0;
            const int row = ((q & 1) ? ly.i1 : ly.i0) - ry0[b];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = j + rx0[b];
                wb_[j] = wr * ((c == lx.i0 ? lx.l0 : 0.f) + (c == lx.i1 ? lx.l1 : 0.f));
            }
            toff1 = (b == 1 ? T2OFF : T3OFF) + ((row * 32 + ch) * 8) * 2;
        }
        const uint4 ha = pack8_bf16(wa), hb = pack8_bf16(wb_);
        u0 = __builtin_bit_cast(bf16x8, ha);
        u1 = __builtin_bit_cast(bf16x8, hb);
        float ra[8], rb[8];
        unpack8_bf16(ha, ra);
        unpack8_bf16(hb, rb);
#pragma unroll
        for (int j = 0; j < 8; ++j) { ra[j] = wa[j] - ra[j]; rb[j] = wb_[j] - rb[j]; }
        u0l = __builtin_bit_cast(bf16x8, pack8_bf16(ra));
        u1l = __builtin_bit_cast(bf16x8, pack8_bf16(rb));
    }
    int o00[3], o01[3], o10[3], o11[3];     // (VALU form) LDS byte offsets of the 4 taps (incl. region base, q*8)
    float w00[3], w01[3], w10[3], w11[3];
#pragma unroll
    for (int b = 0; b < 3 && !MF; ++b) {
        const LerpB lx = lerp_false_b(min(ox, p.W - 1), p.tw[b], p.W);
        const LerpB ly = lerp_false_b(min(oy, p.H - 1), p.th[b], p.H);
        const int r0 = (rbase[b] + (ly.i0 - ry0[b]) * rw[b]) * BPIX, r1 = (rbase[b] + (ly.i1 - ry0[b]) * rw[b]) * BPIX;
        const int c0 = (lx.i0 - rx0[b]) * BPIX + q * 8, c1 = (lx.i1 - rx0[b]) * BPIX + q * 8;
        o00[b] = r0 + c0; o01[b] = r0 + c1; o10[b] = r1 + c0; o11[b] = r1 + c1;
        w00[b] = ly.l0 * lx.l0; w01[b] = ly.l0 * lx.l1; w10[b] = ly.l1 * lx.l0; w11[b] = ly.l1 * lx.l1;
    }

    f32x4 acc3[M3];
#pragma unroll
    for (int m = 0; m < M3; ++m) acc3[m] = f32x4{0.f, 0.f, 0.f, 0.f};

    HB_PREFETCH(0)
    HB_COMMIT(0)
    __syncthreads();
    for (int cc = 0; cc < nchunks; ++cc) {
        const int buf = cc & 1;
        if (cc + 1 < nchunks) HB_PREFETCH(cc + 1)
        const char* tb = smem + buf * STRIDE;
        const char* wb = tb + TB + lane * 16;
        f32x4 a[2] = {*reinterpret_cast<const f32x4*>(p.bias0 + cc * 32 + q * 4),
                      *reinterpret_cast<const f32x4*>(p.bias0 + cc * 32 + 16 + q * 4)};
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int c = 0; c < NB0; ++c) {
                const bf16x8 w0 = *reinterpret_cast<const bf16x8*>(wb + ((m * NB0 + c) * 2 + 0) * 1024);
                const bf16x8 w1 = *reinterpret_cast<const bf16x8*>(wb + ((m * NB0 + c) * 2 + 1) * 1024);
                a[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, x0[c], a[m], 0, 0, 0);
                a[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, x1[c], a[m], 0, 0, 0);
            }
        f32x4 s[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        if (MF) {
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const bf16x8 t0 = *reinterpret_cast<const bf16x8*>(tb + toff0 + m * (16 * 16 * 2));
                const bf16x8 t1 = *reinterpret_cast<const bf16x8*>(tb + toff1 + m * (16 * 8 * 2));
                a[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(t0, u0, a[m], 0, 0, 0);
                a[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(t1, u1, a[m], 0, 0, 0);
                if (ulo) {
                    a[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(t0, u0l, a[m], 0, 0, 0);
                    a[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(t1, u1l, a[m], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int b = 0; b < 3 && !MF; ++b)
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                float v00[4], v01[4], v10[4], v11[4];
                unpack4_bf16(*reinterpret_cast<const uint2*>(tb + o00[b] + m * 32), v00);
                unpack4_bf16(*reinterpret_cast<const uint2*>(tb + o01[b] + m * 32), v01);
                unpack4_bf16(*reinterpret_cast<const uint2*>(tb + o10[b] + m * 32), v10);
                unpack4_bf16(*reinterpret_cast<const uint2*>(tb + o11[b] + m * 32), v11);
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    s[m][i] += w00[b] * v00[i] + w01[b] * v01[i] + w10[b] * v10[i] + w11[b] * v11[i];
                // keep at most one (branch, cout tile)'s 4 taps in flight: 128-VGPR budget at 16 waves
                __builtin_amdgcn_sched_barrier(0);
            }
        float v[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[i] = relu1(a[0][i] + s[0][i]);
            v[4 + i] = relu1(a[1][i] + s[1][i]);
        }
        const bf16x8 hh = __builtin_bit_cast(bf16x8, pack8_bf16(v));
#pragma unroll
        for (int m = 0; m < M3; ++m) {
            const bf16x8 a3 = *reinterpret_cast<const bf16x8*>(wb + (4 * NB0 + m) * 1024);
            acc3[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3, hh, acc3[m], 0, 0, 0);
        }
        if (cc + 1 < nchunks) {
            HB_COMMIT(buf ^ 1)          // nobody reads buffer buf^1 during this iteration
            __syncthreads();
        }
    }
#undef HB_PREFETCH
#undef HB_COMMIT

    // ---- epilogue: h3 = ReLU(acc3 + bias3) -> BF [N][H][W][C3p] (padded channels written as zeros) --------------
    if (in) {
        char* o = p.y + (((size_t)n * p.H + oy) * p.W + ox) * (size_t)(p.C3p * 2);
#pragma unroll
        for (int m = 0; m < M3; ++m) {
            const int co = m * 16 + q * 4;
            const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias3 + co);
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = relu1(acc3[m][i] + bv[i]);
            *reinterpret_cast<uint2*>(o + co * 2) = pack4_bf16(v);
        }
        for (int c = M3 * 16 + q * 4; c < p.C3p; c += 16) *reinterpret_cast<uint2*>(o + c * 2) = make_uint2(0, 0);
    }
}

template <int NB0, int M3, bool MF>
int launch_head_bf_t(const HeadParams& p, hipStream_t stream) {
    auto kern = head_fused_bf_kernel<NB0, M3, MF>;
    const int lds = 2 * ((MF ? TBYTES : BBUF) + (2 * NB0 * 2 + M3) * 1024);
    // exact bf16 interpolation weights: every branch grid is the 2x / 4x / 8x decimation of branch 0's
    int ulo = 0;
    for (int b = 0; b < 3; ++b) ulo |= (p.th[b] << (b + 1)) != p.H || (p.tw[b] << (b + 1)) != p.W;
    if (const int e_ = ensure_dyn_lds(reinterpret_cast<const void*>(kern), lds)) return e_;
    const int tiles_x = (p.W + BHT - 1) / BHT, tiles_y = (p.H + BHY - 1) / BHY;
    const long long nblk = (long long)p.N * tiles_x * tiles_y;
    if (nblk <= 0 || nblk > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(BTHREADS), lds, stream, p, tiles_x, tiles_y, ulo);
    return (int)hipGetLastError();
}

static inline uint16_t hbf16(float f) {
    uint32_t u;
    __builtin_memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

}  // namespace

// geometry: the same source-region bounds as head_fused.hip (branch grids at 1/2, 1/4, 1/8 of branch 0's)
bool head_fused_bf_supported(int H, int W, const int th[3], const int tw[3], int C0p, int K) {
    if (!((C0p == 64 || C0p == 128) && head_fused_supported(H, W, th, tw, 64, K))) return false;
    // the staging threads of a workgroup must cover a tile's source windows: 64 column pairs (matrix-core form) / 128 pixels
    // (lerp_false_b is compiled without fma contraction on both sides, so these are the device's own numbers)
    int rows[3] = {0, 0, 0}, cols[3] = {0, 0, 0};
    for (int b = 0; b < 3; ++b) {
        for (int o = 0; o < H; o += BHY)
            rows[b] = std::max(rows[b], lerp_false_b(std::min(o + BHY - 1, H - 1), th[b], H).i1 - lerp_false_b(o, th[b], H).i0 + 1);
        for (int o = 0; o < W; o += BHT)
            cols[b] = std::max(cols[b], lerp_false_b(std::min(o + BHT - 1, W - 1), tw[b], W).i1 - lerp_false_b(o, tw[b], W).i0 + 1);
    }
    int pairs = 0, pix = 0;
    for (int b = 0; b < 3; ++b) { pairs += rows[b] * ((cols[b] + 1) / 2); pix += rows[b] * cols[b]; }
    return pairs <= BTHREADS / 8 && pix <= BTHREADS / 4 && rows[0] <= BR1 && rows[1] <= BR2 && rows[2] <= BR3 &&
           cols[0] <= 16 && cols[1] <= 8 && cols[2] <= 8;
}

int launch_head_bf(const HeadParams& p, hipStream_t stream) {
    if ((p.Ctp & 31) || (p.C3p & 3)) return (int)hipErrorInvalidValue;
    const int m3 = p.K <= 16 ? 1 : 2;
    const bool valu = getenv("ESAHRNET_BF_HEAD_VALU") != nullptr;       // the first-generation (VALU interpolation) form (A/B, tests)
    if (valu) {
        if (p.C0p == 64 && m3 == 1) return launch_head_bf_t<1, 1, false>(p, stream);
        if (p.C0p == 64 && m3 == 2) return launch_head_bf_t<1, 2, false>(p, stream);
        if (p.C0p == 128 && m3 == 1) return launch_head_bf_t<2, 1, false>(p, stream);
        if (p.C0p == 128 && m3 == 2) return launch_head_bf_t<2, 2, false>(p, stream);
        return (int)hipErrorInvalidValue;
    }
    if (p.C0p == 64 && m3 == 1) return launch_head_bf_t<1, 1, true>(p, stream);
    if (p.C0p == 64 && m3 == 2) return launch_head_bf_t<1, 2, true>(p, stream);
    if (p.C0p == 128 && m3 == 1) return launch_head_bf_t<2, 1, true>(p, stream);
    if (p.C0p == 128 && m3 == 2) return launch_head_bf_t<2, 2, true>(p, stream);
    return (int)hipErrorInvalidValue;
}

// W3 [K][Ct] (1x1) -> [M3][Ctp/32][lane][8] bf16 with the permuted K order of the h0 fragment (head_fused.hip):
// lane (r = l&15, g = l>>4), element j  <->  channel chunk*32 + (j < 4 ? 4g + j : 16 + 4g + j - 4)
size_t head_w3_bf_bytes(int K, int Ctp) { return (size_t)(K <= 16 ? 1 : 2) * (Ctp / 32) * 1024; }
void pack_head_w3_bf(const float* w, int K, int Ct, int Ctp, void* dst) {
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
                    d[(((size_t)m * nch + c) * 64 + l) * 8 + j] = hbf16(v);
                }
}

}  // namespace esa
