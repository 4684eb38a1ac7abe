// stem.hip — stem conv1: 3x3 s1 p1 (cin 1..4 -> cout) + folded BN + ReLU, f32 NCHW in, SB out.
//
// Replaces conv1+bn1+relu of models/seg_hrnet.py:265-267, 426-428.  K = 9*cin <= 36 is far
// too short for the matrix cores and the op is bound by its 256 B/pixel output write, so it is
// plain f32 VALU: one thread per (pixel, 8-channel group) with the group index fastest across
// lanes, i.e. the 8 lanes of a pixel write its 256 contiguous bytes and read the same 9*cin
// inputs (served once by the L1).
#include "kernels.h"
#include "sb.h"

namespace esa {
namespace {

__global__ __launch_bounds__(256) void stem_kernel(StemParams p, long long total) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* wsm = reinterpret_cast<float*>(smem);
    const int G = p.cout >> 3;
    const int wcount = G * p.cin * 72;
    for (int i = threadIdx.x; i < wcount; i += 256) wsm[i] = p.w[i];
    __syncthreads();

    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int c8 = (int)(idx % G);
    long long pix = idx / G;
    const int x = (int)(pix % p.W);
    pix /= p.W;
    const int y = (int)(pix % p.H);
    const int n = (int)(pix / p.H);

    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = p.bias[c8 * 8 + i];
    for (int ci = 0; ci < p.cin; ++ci) {
        const float* xp = p.x + ((size_t)(n * p.cin + ci) * p.H) * p.W;
        const float* wp = wsm + (c8 * p.cin + ci) * 72;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int yy = y + ky - 1;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int xx = x + kx - 1;
                float v = 0.f;
                if (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) v = xp[(size_t)yy * p.W + xx];
                const f32x4 w0 = *reinterpret_cast<const f32x4*>(wp + (ky * 3 + kx) * 8);
                const f32x4 w1 = *reinterpret_cast<const f32x4*>(wp + (ky * 3 + kx) * 8 + 4);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc[i] = fmaf(v, w0[i], acc[i]);
                    acc[4 + i] = fmaf(v, w1[i], acc[4 + i]);
                }
            }
        }
    }
    if (p.relu) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = relu1(acc[i]);
    }
    if (p.fmt == FMT_BF) {
        *reinterpret_cast<uint4*>(p.y + (((size_t)(n * p.H + y) * p.W + x) * p.cout) * 2 + c8 * 16) = pack8_bf16(acc);
        return;
    }
    if (p.fmt == FMT_F32) {
        float* o = reinterpret_cast<float*>(p.y) + ((size_t)(n * p.H + y) * p.W + x) * p.cout + c8 * 8;
        *reinterpret_cast<f32x4*>(o) = f32x4{acc[0], acc[1], acc[2], acc[3]};
        *reinterpret_cast<f32x4*>(o + 4) = f32x4{acc[4], acc[5], acc[6], acc[7]};
        return;
    }
    uint4 hi, lo;
    split8(acc, hi, lo);
    char* o = p.y + (((size_t)(n * p.H + y) * p.W + x) * p.cout) * 4 + c8 * 32;
    *reinterpret_cast<uint4*>(o) = hi;
    *reinterpret_cast<uint4*>(o + 16) = lo;
}

// cin == 1 (seg_hrnet2 / seg_hrnet3): the 72 weights of a thread's 8-channel group live in registers and the thread walks
// STEM_PX consecutive pixels of a row — the kernel above re-reads them from LDS for every pixel (288 bytes of LDS per
// 72 multiply-adds: LDS-bound at 1.7 TB/s of output), and each input sample now serves up to three pixels from a register.
constexpr int STEM_PX = 8;
// POOL (seg_hrnet3's raw skip tensor, whose CBAM needs the per-channel mean and maximum): grid (row pieces, n * H), and the
// block also reduces its 8 x (256 / G) pixels per channel into slab `y * gridDim.x + blockIdx.x` of `pool` — the layout
// pool_partial_kernel writes (cbam.hip), so that kernel's 537 MB read of the tensor just written is not needed.
template <bool POOL>
__global__ __launch_bounds__(256) void stem1_kernel(StemParams p, int xgroups, long long total, float* pool) {
    __shared__ float psum[POOL ? 256 * 8 : 1], pmax[POOL ? 256 * 8 : 1];
    const int G = p.cout >> 3;
    int c8, xg, y, n;
    bool live = true;
    if (POOL) {
        const int u = (int)blockIdx.x * 256 + (int)threadIdx.x;
        live = u < xgroups * G;
        c8 = u % G; xg = min(u / G, xgroups - 1);
        n = (int)blockIdx.y / p.H; y = (int)blockIdx.y - n * p.H;
    } else {
        const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
        if (idx >= total) return;
        c8 = (int)(idx % G);
        long long q = idx / G;
        xg = (int)(q % xgroups);
        q /= xgroups;
        y = (int)(q % p.H);
        n = (int)(q / p.H);
    }
    const int x0 = xg * STEM_PX;
    float ps[8], pm[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { ps[i] = 0.f; pm[i] = -INFINITY; }
    float w[72], b[8];
#pragma unroll
    for (int i = 0; i < 18; ++i) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(p.w + (size_t)c8 * 72 + i * 4);
        w[4 * i] = v[0]; w[4 * i + 1] = v[1]; w[4 * i + 2] = v[2]; w[4 * i + 3] = v[3];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) b[i] = p.bias[c8 * 8 + i];
    const float* xp = p.x + (size_t)n * p.H * p.W;
    float in[3][STEM_PX + 2];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int yy = y + ky - 1;
#pragma unroll
        for (int k = 0; k < STEM_PX + 2; ++k) {
            const int xx = x0 + k - 1;
            in[ky][k] = (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) ? xp[(size_t)yy * p.W + xx] : 0.f;
        }
    }
#pragma unroll
    for (int px = 0; px < STEM_PX; ++px) {
        const int x = x0 + px;
        float acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = b[i];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const float v = in[ky][px + kx];
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = fmaf(v, w[(ky * 3 + kx) * 8 + i], acc[i]);     // same order as stem_kernel
            }
        if (p.relu) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = relu1(acc[i]);
        }
        if (x >= p.W || !live) continue;
        if (POOL) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { ps[i] += acc[i]; pm[i] = fmaxf(pm[i], acc[i]); }
        }
        if (p.fmt == FMT_BF) {
            *reinterpret_cast<uint4*>(p.y + (((size_t)(n * p.H + y) * p.W + x) * p.cout) * 2 + c8 * 16) = pack8_bf16(acc);
        } else if (p.fmt == FMT_F32) {
            float* o = reinterpret_cast<float*>(p.y) + ((size_t)(n * p.H + y) * p.W + x) * p.cout + c8 * 8;
            *reinterpret_cast<f32x4*>(o) = f32x4{acc[0], acc[1], acc[2], acc[3]};
            *reinterpret_cast<f32x4*>(o + 4) = f32x4{acc[4], acc[5], acc[6], acc[7]};
        } else {
            uint4 hi, lo;
            split8(acc, hi, lo);
            char* o = p.y + (((size_t)(n * p.H + y) * p.W + x) * p.cout) * 4 + c8 * 32;
            *reinterpret_cast<uint4*>(o) = hi;
            *reinterpret_cast<uint4*>(o + 16) = lo;
        }
    }
    if (POOL) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { psum[threadIdx.x * 8 + i] = ps[i]; pmax[threadIdx.x * 8 + i] = pm[i]; }
        __syncthreads();
        if ((int)threadIdx.x < p.cout) {          // channel c = tid: over the block's pixel threads (thread = xg_local * G + c8)
            const int cg = threadIdx.x >> 3, i = threadIdx.x & 7;
            float a = 0.f, b = -INFINITY;
            for (int l = 0; l < 256 / G; ++l) { a += psum[(l * G + cg) * 8 + i]; b = fmaxf(b, pmax[(l * G + cg) * 8 + i]); }
            const size_t slab = (size_t)y * gridDim.x + blockIdx.x, P = (size_t)p.H * gridDim.x;
            float* o = pool + (((size_t)n * P + slab) * p.cout + threadIdx.x) * 2;
            o[0] = a; o[1] = b;
        }
    }
}

}  // namespace

// slabs launch_stem_pool writes per image (0: not available for these parameters)
int stem_pool_slabs(int cin, int cout, int H, int W) {
    if (cin != 1 || cout != 64 || H < 1 || W < 1) return 0;         // 256 threads = 32 pixel groups x 8 channel groups
    const int xgroups = (W + STEM_PX - 1) / STEM_PX;
    return H * ((xgroups * (cout >> 3) + 255) / 256);
}

// conv1 as launch_stem computes it + per-slab (sum, max) of every channel into pool[N][slabs][cout][2]
int launch_stem_pool(const StemParams& p, float* pool, hipStream_t stream) {
    const int slabs = stem_pool_slabs(p.cin, p.cout, p.H, p.W);
    if (!slabs || !pool || (p.fmt != FMT_SB && p.fmt != FMT_F32) || (long long)p.N * p.H > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    const int xgroups = (p.W + STEM_PX - 1) / STEM_PX;
    const dim3 grid((unsigned)((xgroups * (p.cout >> 3) + 255) / 256), (unsigned)(p.N * p.H));
    hipLaunchKernelGGL(stem1_kernel<true>, grid, dim3(256), 0, stream, p, xgroups, 0LL, pool);
    return (int)hipGetLastError();
}

int launch_stem(const StemParams& p, hipStream_t stream) {
    if ((p.cout & 31) || p.cin < 1 || p.cin > 4) return (int)hipErrorInvalidValue;
    if (p.cin == 1) {
        const int xgroups = (p.W + STEM_PX - 1) / STEM_PX;
        const long long total1 = (long long)p.N * p.H * xgroups * (p.cout >> 3);
        const long long nblk1 = (total1 + 255) / 256;
        if (nblk1 <= 0 || nblk1 > 0x7fffffffLL) return (int)hipErrorInvalidValue;
        hipLaunchKernelGGL(stem1_kernel<false>, dim3((unsigned)nblk1), dim3(256), 0, stream, p, xgroups, total1, nullptr);
        return (int)hipGetLastError();
    }
    const long long total = (long long)p.N * p.H * p.W * (p.cout >> 3);
    const long long nblk = (total + 255) / 256;
    if (nblk <= 0 || nblk > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    const size_t lds = (size_t)(p.cout >> 3) * p.cin * 72 * sizeof(float);
    hipLaunchKernelGGL(stem_kernel, dim3((unsigned)nblk), dim3(256), lds, stream, p, total);
    return (int)hipGetLastError();
}

}  // namespace esa
