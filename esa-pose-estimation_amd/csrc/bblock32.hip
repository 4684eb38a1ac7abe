// bblock32.hip — a whole BasicBlock of the 32-channel (highest-resolution) branch in one kernel:
//
//   y = ReLU( conv2( ReLU(conv1(x) + b1) ) + b2 + x )        3x3, stride 1, 32 -> 32 -> 32, BN folded
//
// Replaces BasicBlock.forward of models/seg_hrnet.py:45-61 for the blocks of layer1 / branch 0
// (10 of them, 20 of the network's 58 3x3 stride-1 convolutions).  On this branch a convolution
// moves 67 MB in and 67 MB out per 32-crop batch for only 9.7 GFLOP, i.e. it is HBM-bound even on
// the matrix cores; fusing the pair keeps the intermediate activation in LDS and halves the traffic
// (the residual is the block input itself and is re-read from L2, not from HBM).
//
// One persistent workgroup per CU (8 waves), both weight sets (2 x 36.9 KB of split-bf16 MFMA
// fragments) resident in LDS for the whole launch.  Per 16x16 output tile:
//   1. the 20x20 halo-2 input tile is committed from registers (it was prefetched during the
//      previous tile) into 8 LDS operand planes (see conv_cfg.h for the plane layout);
//   2. conv1 is evaluated on the 18x18 halo-1 "mid" tile: the 324 mid pixels are walked in LINEAR
//      order, 16 per MFMA N-tile (21 tiles over 8 waves), each tap being a constant LDS offset;
//   3. + b1, ReLU, pixels outside the image forced to 0 (they are conv2's zero padding), split to
//      hi/lo bf16 and written into operand planes that ALIAS the input tile (barrier in between);
//   4. conv2 runs from those planes exactly like conv_mfma<3,1,16,2> (kx-major fragment reuse),
//      its accumulators having been initialised with b2 + x long before;
//   5. ReLU, split, store.
#include "conv_cfg.h"
#include "devstate.h"
#include "kernels.h"
#include "sb.h"

#ifndef BB_DO_XLOAD
#define BB_DO_XLOAD 1           // ablation switches for tuning experiments (all 1 in the shipped build)
#endif
#ifndef BB_DO_STORE
#define BB_DO_STORE 1
#endif
#ifndef BB_DO_RES
#define BB_DO_RES 1
#endif

namespace esa {
namespace {

constexpr int BT = 16;                         // output tile edge
constexpr int XE = BT + 4, ME = BT + 2;        // input tile edge (halo 2), mid tile edge (halo 1)
constexpr int XPIX = XE * XE, MPIX = ME * ME;  // 400, 324
constexpr int XPLANE = ((XPIX * 16 + 128 + 255) / 256) * 256;   // 6656
constexpr int MPLANE = ((MPIX * 16 + 128 + 255) / 256) * 256;   // 5376
constexpr int XBYTES = 8 * XPLANE;             // 53248 (the mid planes, 43008 B, alias its start)
constexpr int WB = 9 * 2 * 2048;               // 36864: one conv's fragments [2 M-tiles][9 taps][hi|lo][1 KB]
constexpr int LDS_TOTAL = XBYTES + 2 * WB;     // 126976
constexpr int BTHREADS = 512;
constexpr int XIT = (XPIX * 8 + BTHREADS - 1) / BTHREADS;        // 7
constexpr int MTILES = (MPIX + 15) / 16;       // 21 linear N-tiles of the mid tile
constexpr int MT_PER_WAVE = (MTILES + 7) / 8;  // 3

__host__ __device__ constexpr int xplane_off(int j) { return j * XPLANE + (((j >> 2) * 2 + (j & 1)) * 16); }
__host__ __device__ constexpr int mplane_off(int j) { return j * MPLANE + (((j >> 2) * 2 + (j & 1)) * 16); }

__global__ __launch_bounds__(BTHREADS, 1) void bblock32_kernel(BlockParams p, int tiles_x, int tiles_y, int nitems) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* xs = smem;                   // input planes / mid planes
    char* w1s = smem + XBYTES;
    char* w2s = w1s + WB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, px = lane & 15;
    const int G = gridDim.x;
    int item = blockIdx.x;
    if ((G & 7) == 0) item = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);   // XCD-contiguous runs
    if (item >= nitems) return;

    // ---- both weight sets -> LDS once.  Deliberately through registers, not LDS-DMA: with a
    // global_load_lds anywhere in the kernel hipcc drains vmcnt(0) at every __syncthreads() and at
    // every use of an ordinary load, which would serialise the prefetch / store overlap below.
    for (int u = tid; u < 2 * WB / 16; u += BTHREADS) {
        const uint4 v = u < WB / 16 ? p.w1[u] : p.w2[u - WB / 16];
        *reinterpret_cast<uint4*>(w1s + u * 16) = v;
    }

    // ---- input-tile staging map: unit u = it*512 + tid -> pixel q = u>>3, piece j = u&7 ----------
    const int jst = tid & 7, q0 = tid >> 3;               // +64 pixels per iteration
    char* xwr = xs + xplane_off(jst) + q0 * 16;
    int xg[XIT];
    const char* xn;
    int s_n, s_oy0, s_ox0;
#define BB_DECODE(ITEM)                                                                          \
    {                                                                                            \
        int b_ = (ITEM);                                                                         \
        const int tx_ = b_ % tiles_x; b_ /= tiles_x;                                             \
        const int ty_ = b_ % tiles_y;                                                            \
        s_n = b_ / tiles_y; s_oy0 = ty_ * BT; s_ox0 = tx_ * BT;                                  \
        xn = p.x + (size_t)s_n * p.H * p.W * 128;                                                \
        _Pragma("unroll") for (int it = 0; it < XIT; ++it) {                                     \
            const int q = q0 + it * 64;                                                          \
            const int qy = q / XE, qx = q - qy * XE;                                             \
            const int gy = s_oy0 - 2 + qy, gx = s_ox0 - 2 + qx;                                  \
            const bool inside = q < XPIX && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;          \
            xg[it] = inside ? ((gy * p.W + gx) * 128 + jst * 16) : -1;                           \
        }                                                                                        \
    }
    uint4 xr[XIT];
#define BB_PREFETCH()                                                                            \
    {                                                                                            \
        _Pragma("unroll") for (int it = 0; it < XIT; ++it) {                                     \
            uint4 v = make_uint4(0, 0, 0, 0);                                                    \
            if (BB_DO_XLOAD && xg[it] >= 0) v = *reinterpret_cast<const uint4*>(xn + xg[it]);    \
            xr[it] = v;                                                                          \
        }                                                                                        \
    }

    // ---- per-lane read bases -------------------------------------------------------------------------
    // conv1: linear mid pixel idx -> input-tile pixel (my, mx) (tap (0,0)); k-group g picks the plane
    int x1[MT_PER_WAVE];          // LDS byte offset of the lane's input pixel for tile k, tap (0,0)
    int midx[MT_PER_WAVE];        // linear mid pixel index (>= MPIX: padding lane)
#pragma unroll
    for (int k = 0; k < MT_PER_WAVE; ++k) {
        const int idx = (wave + 8 * k) * 16 + px;
        midx[k] = idx;
        const int c = min(idx, MPIX - 1);
        const int my = c / ME, mx = c - my * ME;
        x1[k] = xplane_off(2 * g) + (my * XE + mx) * 16;
    }
    // conv2: rows 2*wave, 2*wave+1 of the 16x16 tile, mid planes with row pitch ME
    const char* m2 = xs + mplane_off(2 * g) + ((wave * 2) * ME + px) * 16;
    const char* w1r = w1s + lane * 16;
    const char* w2r = w2s + lane * 16;
    const f32x4 b1v[2] = {*reinterpret_cast<const f32x4*>(p.bias1 + g * 4),
                          *reinterpret_cast<const f32x4*>(p.bias1 + 16 + g * 4)};
    const f32x4 b2v[2] = {*reinterpret_cast<const f32x4*>(p.bias2 + g * 4),
                          *reinterpret_cast<const f32x4*>(p.bias2 + 16 + g * 4)};

    BB_DECODE(item)
    BB_PREFETCH()
    bool first = true;
    while (item < nitems) {
        const int n = s_n, oy0 = s_oy0, ox0 = s_ox0;
        const int next = item + G;
        if (!first) __syncthreads();          // previous tile's conv2 finished reading the mid planes
#pragma unroll
        for (int it = 0; it < XIT; ++it)
            if (q0 + it * 64 < XPIX) *reinterpret_cast<uint4*>(xwr + it * 1024) = xr[it];
        first = false;
        __syncthreads();
        if (next < nitems) {
            BB_DECODE(next)
            BB_PREFETCH()
        }
        // residual x (L2-hot): loaded here, unconditionally (clamped address, so the loads stay in this
        // basic block and the scheduling barrier pins them), folded into the conv2 accumulators after
        // conv1's MFMAs — a full MFMA phase later, so the latency is covered
        const int ox = ox0 + px;
        f32x4 acc2[2][2];
        uint4 rc[2][2];                        // one 16-byte chunk per lane (sb.h: chunk_to_quad)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                acc2[m][t] = b2v[m];
                const int oyc = min(oy0 + wave * 2 + t, p.H - 1), oxc = min(ox, p.W - 1);
                const char* r = p.x + ((size_t)(n * p.H + oyc) * p.W + oxc) * 128 + chunk_ofs(m * 16 + g * 4, g);
                rc[m][t] = BB_DO_RES ? *reinterpret_cast<const uint4*>(r) : make_uint4(0, 0, 0, 0);
            }
        __builtin_amdgcn_sched_barrier(0);

        // ---- conv1 on the linear mid pixels -------------------------------------------------------
        f32x4 acc1[MT_PER_WAVE][2];
#pragma unroll
        for (int k = 0; k < MT_PER_WAVE; ++k) { acc1[k][0] = b1v[0]; acc1[k][1] = b1v[1]; }
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            bf16x8 wh[2], wl[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                wh[m] = *reinterpret_cast<const bf16x8*>(w1r + ((m * 9 + tap) * 2 + 0) * 1024);
                wl[m] = *reinterpret_cast<const bf16x8*>(w1r + ((m * 9 + tap) * 2 + 1) * 1024);
            }
            const int toff = ((tap / 3) * XE + (tap % 3)) * 16;
#pragma unroll
            for (int k = 0; k < MT_PER_WAVE; ++k) {
                if (wave + 8 * k < MTILES) {              // wave-uniform
                    const bf16x8 xh = *reinterpret_cast<const bf16x8*>(xs + x1[k] + toff);
                    const bf16x8 xo = *reinterpret_cast<const bf16x8*>(xs + x1[k] + toff + XPLANE + 16);
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        acc1[k][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[m], xh, acc1[k][m], 0, 0, 0);
                        acc1[k][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[m], xo, acc1[k][m], 0, 0, 0);
                        acc1[k][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[m], xh, acc1[k][m], 0, 0, 0);
                    }
                }
            }
        }
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float rv[4];
                uint2 rh, rl;
                chunk_to_quad(rc[m][t], rh, rl);
                join4(rh, rl, rv);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc2[m][t][i] += rv[i];
            }
        __syncthreads();                      // every wave is done reading the input planes
        // ---- mid = ReLU(conv1) (0 outside the image) -> operand planes (aliasing the input tile) ----
#pragma unroll
        for (int k = 0; k < MT_PER_WAVE; ++k) {
            const int idx = midx[k];
            if (wave + 8 * k < MTILES && idx < MPIX) {
                const int my = idx / ME, mx = idx - my * ME;
                const int gy = oy0 - 1 + my, gx = ox0 - 1 + mx;
                const bool inside = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    float v[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = inside ? relu1(acc1[k][m][i]) : 0.f;
                    uint2 hi, lo;
                    split4(v, hi, lo);
                    // channels m*16 + g*4 .. +3  ->  8-channel group c8 = 2m + (g>>1), half g&1
                    const int c8 = 2 * m + (g >> 1);
                    char* d = xs + mplane_off(2 * c8) + idx * 16 + (g & 1) * 8;
                    *reinterpret_cast<uint2*>(d) = hi;
                    *reinterpret_cast<uint2*>(d + MPLANE + 16) = lo;
                }
            }
        }
        __syncthreads();
        // ---- conv2 from the mid planes (kx-major: each row fragment feeds up to 3 taps) -----------
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            bf16x8 wh[3][2], wl[3][2];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    wh[ky][m] = *reinterpret_cast<const bf16x8*>(w2r + ((m * 9 + ky * 3 + kx) * 2 + 0) * 1024);
                    wl[ky][m] = *reinterpret_cast<const bf16x8*>(w2r + ((m * 9 + ky * 3 + kx) * 2 + 1) * 1024);
                }
#pragma unroll
            for (int i = 0; i < 4; ++i) {                 // input rows 0..3 of this wave's 2 output rows
                const int off = (i * ME + kx) * 16;
                const bf16x8 xh = *reinterpret_cast<const bf16x8*>(m2 + off);
                const bf16x8 xo = *reinterpret_cast<const bf16x8*>(m2 + off + MPLANE + 16);
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int t = i - ky;
                    if (t >= 0 && t < 2) {
#pragma unroll
                        for (int m = 0; m < 2; ++m) {
                            acc2[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[ky][m], xh, acc2[m][t], 0, 0, 0);
                            acc2[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[ky][m], xo, acc2[m][t], 0, 0, 0);
                            acc2[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[ky][m], xh, acc2[m][t], 0, 0, 0);
                        }
                    }
                }
            }
        }
        // ---- epilogue ------------------------------------------------------------------------------
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int oy = oy0 + wave * 2 + t;
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = relu1(acc2[m][t][i]);
                uint2 hi, lo;
                split4(v, hi, lo);
                const uint4 ch = quad_to_chunk(hi, lo);            // all lanes; only the store is predicated
                if ((BB_DO_STORE || acc2[m][t][0] == 123.456f) && oy < p.H && ox < p.W)
                    *reinterpret_cast<uint4*>(p.y + ((size_t)(n * p.H + oy) * p.W + ox) * 128 + chunk_ofs(m * 16 + g * 4, g)) = ch;
            }
        item = next;
    }
#undef BB_DECODE
#undef BB_PREFETCH
}

}  // namespace

int launch_bblock32(const BlockParams& p, hipStream_t stream) {
    if ((long long)p.H * p.W * 128 > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    if (const int e_ = ensure_dyn_lds(reinterpret_cast<const void*>(bblock32_kernel), LDS_TOTAL)) return e_;
    const int cus = device_cus();
    const int tiles_x = (p.W + BT - 1) / BT, tiles_y = (p.H + BT - 1) / BT;
    const long long nitems = (long long)p.N * tiles_x * tiles_y;
    if (nitems <= 0 || nitems > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    const int grid = (int)(nitems < cus ? nitems : cus);
    hipLaunchKernelGGL(bblock32_kernel, dim3((unsigned)grid), dim3(BTHREADS), LDS_TOTAL, stream, p, tiles_x,
                       tiles_y, (int)nitems);
    return (int)hipGetLastError();
}

}  // namespace esa
