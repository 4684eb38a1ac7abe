// head_gather.hip — seg_hrnet3 head: the share of last_layer[0] that comes from the low-resolution branches.
//
// models/seg_hrnet3.py:506-515 up-samples branches 1..3 to branch 0's grid (bilinear, align_corners=False), concatenates
// 480 channels and runs a 3x3 480 -> 480 convolution at that resolution: 68 of the network's 92 GFLOP per crop, of which 80 %
// multiply interpolated copies of 32x32 and 16x16 images.  Convolution and interpolation are both linear and the channel
// mixing W_tap commutes with the (channel-wise) interpolation U:
//
//     conv3x3(U x)(p) = sum_tap W_tap (U x)(p + d_tap) = sum_tap U(W_tap x)(p + d_tap)        (terms with p + d_tap outside: 0)
//
// so for branches 2 and 3 the nine 1x1 products z_tap = W_tap x are formed ON THE LOW-RESOLUTION GRID by the 1x1 kernel (a
// 9*480-channel output; 16x / 64x fewer pixels than the direct form) and this kernel evaluates the right-hand side: for every
// output pixel the nine shifted bilinear samples of the nine z planes, summed.  Branch 0 and the up-sampled branch 1 stay in a
// direct 3x3 convolution that takes this kernel's result as its residual input (then bias and ReLU).
//
// z channel order (chosen by the weight packing in plan.hip): o = (cout / 8) * 72 + tap * 8 + cout % 8, so the nine taps of an
// 8-channel group are 288 contiguous bytes of a z pixel.
//
// Workgroup = a 16 x 32 tile of output pixels x `gpw` channel groups, one group at a time:
//   stage   the z pixels (plain f32) under the tile (+ the one-pixel conv halo) of both branches, 288 bytes each, into LDS —
//           the next group's pieces are fetched into registers before the x pass and committed behind it;
//   x pass  b[branch][dy][low-res row][tile column] = sum_dx sum_2 lx * z[tap(dy, dx)][row][column sample]      (f32 x 8 in LDS)
//   y pass  out[row][column] = sum_branch sum_dy sum_2 ly * b[branch][dy][row sample][column]  ->  split-bf16 store.
// The separable form costs 6 + 6 multiply-adds per branch and output element instead of 36; the sample positions and
// weights of the tile's columns and rows are tabulated in LDS once per workgroup.
#include <algorithm>

#include "devstate.h"
#include "kernels.h"
#include "sb.h"

namespace esa {
namespace {

constexpr int GT_H = 16, GT_W = 32;

struct Lerp1 { int i0, i1; float l0, l1; };
// PyTorch's area_pixel_compute_source_index, align_corners = False (the F.upsample default the reference relies on)
__device__ __host__ inline Lerp1 lerp_half(int dst, int in, int out) {
    const float scale = (float)in / (float)out;
    float src = scale * ((float)dst + 0.5f) - 0.5f;
    src = src < 0.f ? 0.f : src;
    Lerp1 r;
    r.i0 = (int)src < in - 1 ? (int)src : in - 1;
    r.i1 = r.i0 + (r.i0 < in - 1 ? 1 : 0);
    r.l1 = src - (float)r.i0;
    r.l0 = 1.f - r.l1;
    return r;
}

// per-tile sample tables (LDS): entry = {offset of sample 0, offset of sample 1, weight 0, weight 1}; weight 0 = weight 1 = 0
// for positions outside the image (the convolution's zero padding)
struct Samp { int o0, o1; float l0, l1; };
constexpr int GT_T = 512;        // threads per workgroup: 16 rows of 32 columns
// GT_MAXP (template): staged 16-byte pieces a thread may hold for the next channel group — 6 (170 window pixels, 127 VGPRs: four
// waves per SIMD) for every power-of-two crop, 8 (227 pixels, 141 VGPRs) for odd level sizes whose windows are larger

// (HIP's second launch-bound argument is waves per SIMD.  Asking for 4 — two 8-wave workgroups per CU, 128 VGPRs — makes hipcc
// spill 85 registers; asked for 2 it allocates 127 for the 6-piece variant on its own, which gives the same occupancy.)
// F32OUT: the output tensor is plain f32 NHWC (fp32-grade mode) instead of split-bf16 — a template parameter, not a run-time
// flag: with both stores in one kernel hipcc allocates 162 VGPRs (three waves per SIMD, 0.9 -> 1.2 ms)
template <int GT_MAXP, bool F32OUT>
__global__ __launch_bounds__(GT_T, 2) void head_gather_kernel(GatherParams p, int tiles_x, int tiles_y) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    int t = (int)blockIdx.x;
    const int tx0 = (t % tiles_x) * GT_W; t /= tiles_x;
    const int ty0 = (t % tiles_y) * GT_H;
    const int n = t / tiles_y;
    // staged source window of each branch: rows [ylo, ylo + rows), columns [xlo, xlo + cols)
    int ylo[2], xlo[2], rows[2], cols[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        ylo[b] = lerp_half(max(ty0 - 1, 0), p.h[b], p.H).i0;
        rows[b] = lerp_half(min(ty0 + GT_H, p.H - 1), p.h[b], p.H).i1 - ylo[b] + 1;
        xlo[b] = lerp_half(max(tx0 - 1, 0), p.w[b], p.W).i0;
        cols[b] = lerp_half(min(tx0 + GT_W, p.W - 1), p.w[b], p.W).i1 - xlo[b] + 1;
    }
    char* zs[2];
    float* bs[2];
    zs[0] = smem;
    zs[1] = zs[0] + p.R[0] * p.Cc[0] * 288;
    bs[0] = reinterpret_cast<float*>(zs[1] + p.R[1] * p.Cc[1] * 288);
    bs[1] = bs[0] + 3 * p.R[0] * GT_W * 8;
    Samp* xt = reinterpret_cast<Samp*>(bs[1] + 3 * p.R[1] * GT_W * 8);    // [2][GT_W + 2]: column X = tx0 - 1 + i
    Samp* yt = xt + 2 * (GT_W + 2);                                        // [2][GT_H + 2]: row    Y = ty0 - 1 + i
    for (int i = tid; i < 2 * (GT_W + 2) + 2 * (GT_H + 2); i += GT_T) {
        const bool isx = i < 2 * (GT_W + 2);
        const int j = isx ? i : i - 2 * (GT_W + 2);
        const int span = isx ? GT_W + 2 : GT_H + 2;
        const int b = j / span, k = j % span;
        const int pos = (isx ? tx0 : ty0) - 1 + k, lim = isx ? p.W : p.H;
        Samp sm{0, 0, 0.f, 0.f};
        if (pos >= 0 && pos < lim) {
            const Lerp1 L = lerp_half(pos, isx ? (b ? p.w[1] : p.w[0]) : (b ? p.h[1] : p.h[0]), lim);
            const int lo = isx ? (b ? xlo[1] : xlo[0]) : (b ? ylo[1] : ylo[0]);
            sm.o0 = L.i0 - lo; sm.o1 = L.i1 - lo; sm.l0 = L.l0; sm.l1 = L.l1;
        }
        (isx ? xt : yt)[j] = sm;
    }
    const int g0 = (int)blockIdx.y * p.gpw;
    const int g1 = min(min(g0 + p.gpw, p.ngroups), p.nreal);
    // padding channels of the output: exact zeros (the direct convolution adds them as its residual)
    for (int cg = max(g0, p.nreal); cg < min(g0 + p.gpw, p.ngroups); ++cg)
        for (int u = tid; u < GT_H * GT_W; u += GT_T) {
            const int Y0 = ty0 + u / GT_W, X0 = tx0 + u % GT_W;
            if (Y0 >= p.H || X0 >= p.W) continue;
            char* o = p.y + (((size_t)n * p.H + Y0) * p.W + X0) * (size_t)(p.Cp * 4) + (size_t)cg * 32;
            *reinterpret_cast<uint4*>(o) = make_uint4(0, 0, 0, 0);
            *reinterpret_cast<uint4*>(o + 16) = make_uint4(0, 0, 0, 0);
        }
    // staging: piece i of the window = 16 bytes; thread holds pieces tid, tid + 256, ... of both branches in registers
    const int np0 = rows[0] * cols[0] * 18, npt = np0 + rows[1] * cols[1] * 18;
    // first staged z pixel of each branch
    const char* zg0 = p.z[0] + (((size_t)n * p.h[0] + ylo[0]) * p.w[0] + xlo[0]) * (size_t)p.zpix[0];
    const char* zg1 = p.z[1] + (((size_t)n * p.h[1] + ylo[1]) * p.w[1] + xlo[1]) * (size_t)p.zpix[1];
    uint4 pre[GT_MAXP];        // (zero-initialised: left undefined, hipcc keeps the array in scratch)
#pragma unroll
    for (int k = 0; k < GT_MAXP; ++k) pre[k] = make_uint4(0, 0, 0, 0);
    // (macros, not lambdas: captured by reference the register array `pre` ends up in scratch)
#define GT_PIECE(K)                                                                              \
    const int i_ = tid + (K) * GT_T;                                                              \
    const int b_ = i_ < np0 ? 0 : 1;                                                              \
    const int j_ = b_ ? i_ - np0 : i_;                                                            \
    const int piece_ = j_ % 18, px_ = j_ / 18;                                                    \
    const int cols_ = b_ ? cols[1] : cols[0];                                                     \
    const int c_ = px_ % cols_, r_ = px_ / cols_;
#define GT_FETCH(CG)                                                                              \
    _Pragma("unroll") for (int k = 0; k < GT_MAXP; ++k) {                                         \
        GT_PIECE(k)                                                                               \
        if (i_ < npt)                                                                             \
            pre[k] = *reinterpret_cast<const uint4*>((b_ ? zg1 : zg0) + ((size_t)r_ * (b_ ? p.w[1] : p.w[0]) + c_) *            \
                                                     (size_t)(b_ ? p.zpix[1] : p.zpix[0]) + (size_t)(CG) * 288 + piece_ * 16);    \
    }
#define GT_COMMIT()                                                                               \
    _Pragma("unroll") for (int k = 0; k < GT_MAXP; ++k) {                                         \
        GT_PIECE(k)                                                                               \
        if (i_ < npt) *reinterpret_cast<uint4*>((b_ ? zs[1] : zs[0]) + (r_ * (b_ ? p.Cc[1] : p.Cc[0]) + c_) * 288 + piece_ * 16) = pre[k]; \
    }
    if (g0 < g1) { GT_FETCH(g0) GT_COMMIT() }
    __syncthreads();
    const int px = tid & (GT_W - 1), slot0 = tid >> 5;           // a thread keeps its tile column in both passes; slot0 = 0..15
    for (int cg = g0; cg < g1; ++cg) {
        if (cg + 1 < g1) { GT_FETCH(cg + 1) }    // lands while the x pass runs
        // ---- x pass: slots = (branch, dy, staged row) ----
        const int sa = 3 * rows[0], sb_ = 3 * rows[1];
        for (int sl = slot0; sl < sa + sb_; sl += GT_T / GT_W) {
            const int b = sl < sa ? 0 : 1;
            const int v = b ? sl - sa : sl;
            const int rows_b = b ? rows[1] : rows[0], ccb = b ? p.Cc[1] : p.Cc[0], rb = b ? p.R[1] : p.R[0];
            const char* zsb = b ? zs[1] : zs[0];
            float* bsb = b ? bs[1] : bs[0];
            const int r = v % rows_b, dy = v / rows_b;
            float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const Samp sm = xt[b * (GT_W + 2) + px + dx];
                const float* a0 = reinterpret_cast<const float*>(zsb + (r * ccb + sm.o0) * 288 + (dy * 3 + dx) * 32);
                const float* a1 = reinterpret_cast<const float*>(zsb + (r * ccb + sm.o1) * 288 + (dy * 3 + dx) * 32);
                const float4 u0 = *reinterpret_cast<const float4*>(a0), u1 = *reinterpret_cast<const float4*>(a0 + 4);
                const float4 w0 = *reinterpret_cast<const float4*>(a1), w1 = *reinterpret_cast<const float4*>(a1 + 4);
                acc[0] += sm.l0 * u0.x + sm.l1 * w0.x; acc[1] += sm.l0 * u0.y + sm.l1 * w0.y;
                acc[2] += sm.l0 * u0.z + sm.l1 * w0.z; acc[3] += sm.l0 * u0.w + sm.l1 * w0.w;
                acc[4] += sm.l0 * u1.x + sm.l1 * w1.x; acc[5] += sm.l0 * u1.y + sm.l1 * w1.y;
                acc[6] += sm.l0 * u1.z + sm.l1 * w1.z; acc[7] += sm.l0 * u1.w + sm.l1 * w1.w;
            }
            // two planes of four channels each: lanes of a row are 16 bytes apart (32 would be a 2-way bank conflict)
            float* o = bsb + ((dy * rb + r) * 2 * GT_W + px) * 4;
            *reinterpret_cast<float4*>(o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
            *reinterpret_cast<float4*>(o + GT_W * 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
        }
        __syncthreads();
        if (cg + 1 < g1) { GT_COMMIT() }         // zs is free: nobody reads it before the next barrier
        // ---- y pass ----
        {
            const int py = slot0;
            const int Y0 = ty0 + py, X0 = tx0 + px;
            float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const Samp sm = yt[b * (GT_H + 2) + py + dy];
                    const float* a0 = bs[b] + ((dy * p.R[b] + sm.o0) * 2 * GT_W + px) * 4;
                    const float* a1 = bs[b] + ((dy * p.R[b] + sm.o1) * 2 * GT_W + px) * 4;
                    const float4 u0 = *reinterpret_cast<const float4*>(a0), u1 = *reinterpret_cast<const float4*>(a0 + GT_W * 4);
                    const float4 w0 = *reinterpret_cast<const float4*>(a1), w1 = *reinterpret_cast<const float4*>(a1 + GT_W * 4);
                    acc[0] += sm.l0 * u0.x + sm.l1 * w0.x; acc[1] += sm.l0 * u0.y + sm.l1 * w0.y;
                    acc[2] += sm.l0 * u0.z + sm.l1 * w0.z; acc[3] += sm.l0 * u0.w + sm.l1 * w0.w;
                    acc[4] += sm.l0 * u1.x + sm.l1 * w1.x; acc[5] += sm.l0 * u1.y + sm.l1 * w1.y;
                    acc[6] += sm.l0 * u1.z + sm.l1 * w1.z; acc[7] += sm.l0 * u1.w + sm.l1 * w1.w;
                }
            if (Y0 < p.H && X0 < p.W) {
                if constexpr (F32OUT) {
                    char* o = p.y + (((size_t)n * p.H + Y0) * p.W + X0) * (size_t)(p.Cp * 4) + (size_t)cg * 32;
                    *reinterpret_cast<f32x4*>(o) = f32x4{acc[0], acc[1], acc[2], acc[3]};
                    *reinterpret_cast<f32x4*>(o + 16) = f32x4{acc[4], acc[5], acc[6], acc[7]};
                } else {
                    uint4 hi, lo;
                    split8(acc, hi, lo);        // (the address AFTER the split: computed before it, 129 VGPRs = one wave per SIMD less)
                    char* o = p.y + (((size_t)n * p.H + Y0) * p.W + X0) * (size_t)(p.Cp * 4) + (size_t)cg * 32;
                    *reinterpret_cast<uint4*>(o) = hi;
                    *reinterpret_cast<uint4*>(o + 16) = lo;
                }
            }
        }
        __syncthreads();        // bs is rewritten by the next x pass; zs now holds the next group
    }
#undef GT_PIECE
#undef GT_FETCH
#undef GT_COMMIT
}

// largest source window any tile needs (exactly the kernel's own arithmetic)
void window_max(int in, int out, int tile, int* m) {
    *m = 0;
    for (int t0 = 0; t0 < out; t0 += tile) {
        const int lo = lerp_half(std::max(t0 - 1, 0), in, out).i0;
        const int hi = lerp_half(std::min(t0 + tile, out - 1), in, out).i1;
        *m = std::max(*m, hi - lo + 1);
    }
    *m += 1;        // the device may contract the source-index arithmetic into an fma: one sample of slack
}

size_t gather_lds(const GatherParams& p) {
    size_t b = 0;
    for (int i = 0; i < 2; ++i) b += (size_t)p.R[i] * p.Cc[i] * 288 + (size_t)3 * p.R[i] * GT_W * 32;
    return b + (2 * (GT_W + 2) + 2 * (GT_H + 2)) * 16;
}

void fill_windows(GatherParams& p) {
    for (int i = 0; i < 2; ++i) {
        window_max(p.h[i], p.H, GT_H, &p.R[i]);
        window_max(p.w[i], p.W, GT_W, &p.Cc[i]);
    }
}

}  // namespace

bool head_gather_supported(int H, int W, const int* h, const int* w, int Cp) {
    GatherParams p{};
    p.H = H; p.W = W; p.Cp = Cp;
    for (int i = 0; i < 2; ++i) { p.h[i] = h[i]; p.w[i] = w[i]; }
    if (H < 1 || W < 1 || h[0] < 1 || h[1] < 1 || w[0] < 1 || w[1] < 1 || (Cp & 7)) return false;
    fill_windows(p);
    // the windows must fit the LDS of a CU (two workgroups per CU up to 80 KB: every power-of-two crop; odd level sizes with
    // ratios a little under 4 / 8 need up to ~105 KB and run one per CU) and the registers the next group is staged through
    return gather_lds(p) <= 150 * 1024 && (p.R[0] * p.Cc[0] + p.R[1] * p.Cc[1]) * 18 <= 8 * GT_T;
}

int launch_head_gather(GatherParams p, hipStream_t stream, int fmt) {
    if (!head_gather_supported(p.H, p.W, p.h, p.w, p.Cp) || (fmt != FMT_SB && fmt != FMT_F32)) return (int)hipErrorInvalidValue;
    fill_windows(p);
    p.ngroups = p.Cp >> 3;
    p.nreal = (p.C + 7) >> 3;
    p.gpw = 6;              // measured 3 .. 30: flat within 5 %, 6 best
    const int tiles_x = (p.W + GT_W - 1) / GT_W, tiles_y = (p.H + GT_H - 1) / GT_H;
    const size_t lds = gather_lds(p);
    const long long nblk = (long long)tiles_x * tiles_y * p.N;
    if (nblk > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    const dim3 grid((unsigned)nblk, (unsigned)((p.ngroups + p.gpw - 1) / p.gpw));
    const bool six = (p.R[0] * p.Cc[0] + p.R[1] * p.Cc[1]) * 18 <= 6 * GT_T, f32 = fmt == FMT_F32;
    auto kern = six ? (f32 ? head_gather_kernel<6, true> : head_gather_kernel<6, false>)
                    : (f32 ? head_gather_kernel<8, true> : head_gather_kernel<8, false>);
    if (const int e_ = ensure_dyn_lds(reinterpret_cast<const void*>(kern), lds)) return e_;
    hipLaunchKernelGGL(kern, grid, dim3(GT_T), lds, stream, p, tiles_x, tiles_y);
    return (int)hipGetLastError();
}

}  // namespace esa
