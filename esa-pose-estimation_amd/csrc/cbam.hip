// cbam.hip — the CBAM attention of the seg_hrnet3 variant on SB tensors — and on the
// fp32-grade mode's plain f32 NHWC tensors: 8 channels are 32 bytes in both (sb.h: load8_fmt / store8_fmt).
//
// Replaces ChannelAttention / SpatialAttention of models/seg_hrnet3.py:32-61 and their use inside
// BasicBlock.forward (:90-91: out = ca(out)*out; out = sa(out)*out, before the residual add) and on
// the stem skip (:516-517).  Four small memory-bound kernels, none of them on the matrix cores:
//   pool_partial   per (n, c): sum and max over a slab of pixels            (AdaptiveAvg/MaxPool2d(1))
//   ca_mlp         finishes the pooling, fc = 1x1 C->C/16, ReLU, 1x1 C/16->C on both, add, sigmoid
//   cbam_maps      per pixel: mean_c and max_c of ca*x                      (torch.mean / torch.max, dim=1)
//   cbam_apply     per pixel: sa = sigmoid(conv7x7([mean, max])), y = [relu](sa*ca*x [+ res]) -> SB slice
// plus `resample_slice`: bilinear re-sampling (both align_corners conventions, or plain copy) of an
// SB tensor into a channel slice of a wider SB tensor — the torch.cat([x0, up(x1), up(x2), up(x3)])
// of :506-512 and the cat([up2(last_layer), cbam(x0)]) of :518, materialised because this variant's
// last_layer[0] is a 3x3 convolution (no push-through-the-upsampling trick as in plan.hip).
#include "kernels.h"
#include "sb.h"

namespace esa {
namespace {

__device__ __forceinline__ float sigmoidf(float v) { return 1.f / (1.f + expf(-v)); }

// ---- pool_partial: grid (P, N), 256 threads; thread = (pixel lane pl, channel group c8) ----------
__device__ __forceinline__ void pool_partial_body(const char* x, float* partial, int HW, int Cp, int P, int slab, int n,
                                                  float* ssum, float* smax, bool f32) {
    const int G = Cp >> 3, PL = 256 / G;
    const int S = (HW + P - 1) / P;
    const int p0 = slab * S, p1 = min(HW, p0 + S);
    const int tid = threadIdx.x, pl = tid / G, c8 = tid - pl * G;
    float s[8], m[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { s[i] = 0.f; m[i] = -INFINITY; }
    if (pl < PL)
        for (int p = p0 + pl; p < p1; p += PL) {
            const char* a = x + ((size_t)n * HW + p) * (size_t)(Cp * 4) + c8 * 32;
            float v[8];
            load8_fmt(a, v, f32);
#pragma unroll
            for (int i = 0; i < 8; ++i) { s[i] += v[i]; m[i] = fmaxf(m[i], v[i]); }
        }
#pragma unroll
    for (int i = 0; i < 8; ++i) { ssum[tid * 8 + i] = s[i]; smax[tid * 8 + i] = m[i]; }
    __syncthreads();
    if (tid < Cp) {                       // channel c = tid: reduce over the pixel lanes
        const int g = tid >> 3, i = tid & 7;
        float a = 0.f, b = -INFINITY;
        for (int l = 0; l < PL; ++l) { a += ssum[(l * G + g) * 8 + i]; b = fmaxf(b, smax[(l * G + g) * 8 + i]); }
        float* o = partial + (((size_t)n * P + slab) * Cp + tid) * 2;
        o[0] = a; o[1] = b;
    }
}
__global__ __launch_bounds__(256) void pool_partial_kernel(const char* x, float* partial, int HW, int Cp, int P, int f32) {
    __shared__ float ssum[256 * 8];
    __shared__ float smax[256 * 8];
    pool_partial_body(x, partial, HW, Cp, P, (int)blockIdx.x, (int)blockIdx.y, ssum, smax, f32 != 0);
}

// ---- ca_mlp: grid N, 256 threads ----------------------------------------------------------------------
__device__ __forceinline__ void ca_mlp_body(const float* partial, const float* w0, const float* w2, float* ca, int HW, int C,
                                            int Cp, int Cr, int P, int n, float* avg, float* mx, float* ha, float* hm) {
    const int tid = threadIdx.x;
    // finish the pooling: 8 slabs per step so that 8 loads are in flight (one dependent round trip per slab
    // made this kernel 18 us of pure latency); slab order of the sum is kept
    for (int c = tid; c < Cp; c += 256) {
        float a = 0.f, b = -INFINITY;
        const float2* q = reinterpret_cast<const float2*>(partial) + (size_t)n * P * Cp + c;
        int s = 0;
        for (; s + 8 <= P; s += 8) {
            float2 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = q[(size_t)(s + k) * Cp];
#pragma unroll
            for (int k = 0; k < 8; ++k) { a += v[k].x; b = fmaxf(b, v[k].y); }
        }
        for (; s < P; ++s) { const float2 v = q[(size_t)s * Cp]; a += v.x; b = fmaxf(b, v.y); }
        avg[c] = a / (float)HW; mx[c] = b;
    }
    __syncthreads();
    // fc[0] (no bias) + ReLU on both pooled vectors: 16 lanes per hidden unit, strided over the channels,
    // then a 16-lane butterfly (a single thread per unit walked all C channels serially: 18 us per launch)
    for (int j0 = 0; j0 < Cr; j0 += 16) {
        const int j = j0 + (tid >> 4), l = tid & 15;
        float a = 0.f, b = 0.f;
        if (j < Cr)
            for (int c = l; c < C; c += 16) { const float wv = w0[j * C + c]; a += wv * avg[c]; b += wv * mx[c]; }
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1) { a += __shfl_xor(a, off, 16); b += __shfl_xor(b, off, 16); }
        if (j < Cr && l == 0) { ha[j] = fmaxf(a, 0.f); hm[j] = fmaxf(b, 0.f); }
    }
    __syncthreads();
    for (int c = tid; c < Cp; c += 256) {
        float v = 0.f;
        if (c < C) {
            float a = 0.f, b = 0.f;
            for (int j = 0; j < Cr; ++j) { a += w2[c * Cr + j] * ha[j]; b += w2[c * Cr + j] * hm[j]; }
            v = sigmoidf(a + b);
        }
        ca[(size_t)n * Cp + c] = v;
    }
}
__global__ __launch_bounds__(256) void ca_mlp_kernel(const float* partial, const float* w0, const float* w2,
                                                     float* ca, int HW, int C, int Cp, int Cr, int P) {
    __shared__ float avg[512], mx[512], ha[64], hm[64];
    ca_mlp_body(partial, w0, w2, ca, HW, C, Cp, Cr, P, (int)blockIdx.x, avg, mx, ha, hm);
}

// ---- cbam_maps: thread = (pixel, 8-channel group), group fastest -> coalesced 32-B pieces; the G
// partial (sum, max) of a pixel meet in LDS ------------------------------------------------------------
__device__ __forceinline__ void cbam_maps_body(const char* x, const float* ca, float* maps, long long npix, int HW, int C, int Cp,
                                               long long block, float* ps, float* pm, bool f32) {
    const int G = Cp >> 3, PPB = 256 / G;                 // pixels per block
    const int tid = threadIdx.x, pl = tid / G, c8 = tid - pl * G;
    const long long pix = block * PPB + pl;
    float s = 0.f, m = -INFINITY;
    if (pl < PPB && pix < npix && c8 * 8 < C) {
        const int n = (int)(pix / HW);
        const float* cn = ca + (size_t)n * Cp + c8 * 8;
        const char* a = x + (size_t)pix * (size_t)(Cp * 4) + c8 * 32;
        float v[8];
        load8_fmt(a, v, f32);
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (c8 * 8 + i < C) { const float u = v[i] * cn[i]; s += u; m = fmaxf(m, u); }
    }
    ps[tid] = s; pm[tid] = m;
    __syncthreads();
    if (pl < PPB && pix < npix && c8 == 0) {
        float a = 0.f, b = -INFINITY;
        for (int g = 0; g < G; ++g) { a += ps[pl * G + g]; b = fmaxf(b, pm[pl * G + g]); }
        maps[pix * 2 + 0] = a / (float)C;
        maps[pix * 2 + 1] = b;
    }
}
__global__ __launch_bounds__(256) void cbam_maps_kernel(const char* x, const float* ca, float* maps, long long npix,
                                                        int HW, int C, int Cp, int f32) {
    __shared__ float ps[256], pm[256];
    cbam_maps_body(x, ca, maps, npix, HW, C, Cp, (long long)blockIdx.x, ps, pm, f32 != 0);
}

// ---- cbam_apply: thread = (pixel, 8-channel group); the pixel's first thread evaluates the 7x7
// spatial attention once and shares it through LDS ----------------------------------------------------
__device__ __forceinline__ void cbam_apply_body(const CbamApplyParams& p, long long npix, long long block, float* w, float* sas) {
    if (threadIdx.x < 98) w[threadIdx.x] = p.w_sa[threadIdx.x];
    __syncthreads();
    const int G = p.Cp >> 3, PPB = 256 / G;
    const int tid = threadIdx.x, pl = tid / G, c8 = tid - pl * G;
    const long long idx = block * PPB + pl;
    const bool live = pl < PPB && idx < npix;
    const int HW = p.H * p.W;
    int n = 0;
    float part = 0.f;
    if (live) {
        // the 49 taps of the pixel's 7x7 spatial attention are spread over its G threads (a single thread
        // walking all of them serialised 49 dependent loads while the others waited at the barrier)
        n = (int)(idx / HW);
        const int pix = (int)(idx - (long long)n * HW);
        const int y = pix / p.W, x = pix - y * p.W;
        for (int t = c8; t < 49; t += G) {
            const int ky = t / 7, kx = t - ky * 7;
            const int yy = y + ky - 3, xx = x + kx - 3;
            if (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) {
                const float2 mp = *reinterpret_cast<const float2*>(p.maps + (((size_t)n * p.H + yy) * p.W + xx) * 2);
                part += w[t] * mp.x + w[49 + t] * mp.y;
            }
        }
    }
    sas[tid] = part;
    __syncthreads();
    float sa_sum = 0.f;
    if (live)
        for (int j = 0; j < G; ++j) sa_sum += sas[pl * G + j];
    __syncthreads();
    if (live && c8 == 0) sas[pl] = sigmoidf(sa_sum);
    __syncthreads();
    if (!live) return;
    const float sa = sas[pl];
    const float* cn = p.ca + (size_t)n * p.Cp + c8 * 8;
    const bool f32 = p.fmt == FMT_F32;
    const char* a = p.x + (size_t)idx * (size_t)(p.Cp * 4) + c8 * 32;
    float v[8];
    load8_fmt(a, v, f32);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (sa * cn[i]) * v[i];
    if (p.res) {
        const char* r = p.res + (size_t)idx * (size_t)(p.Cp * 4) + c8 * 32;
        float rv[8];
        load8_fmt(r, rv, f32);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] += rv[i];
    }
    if (p.relu) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = fmaxf(v[i], 0.f);
    }
    store8_fmt(p.y + (size_t)idx * (size_t)p.y_pix_bytes + ((p.y_c0 >> 3) + c8) * 32, v, f32);
}

__global__ __launch_bounds__(256) void cbam_apply_kernel(CbamApplyParams p, long long npix) {
    __shared__ float w[98];
    __shared__ float sas[256];
    cbam_apply_body(p, npix, (long long)blockIdx.x, w, sas);
}

// ---- cbam_spatial: cbam_maps + cbam_apply in one pass over a 16 x 32 tile (+ 3-pixel halo for the 7x7) -------------
// The two kernels above read x twice and hand the 2-channel maps through HBM.  Here a workgroup forms the maps of its
// halo'd tile in LDS (the halo pixels' x comes out of L2: the neighbouring tiles read them too), evaluates the 7x7 spatial
// attention from there and applies it.  thread = (pixel, 8-channel group), the G = Cp/8 threads of a pixel are
// neighbouring lanes (G a power of two <= 32): channel reductions and the 49 taps are shared by xor-shuffles.
constexpr int CS_TH = 16, CS_TW = 32, CS_HH = CS_TH + 6, CS_HW = CS_TW + 6;
__device__ __forceinline__ void cbam_spatial_body(const CbamApplyParams& p, int tiles_x, int tiles_y, int block, float* mp, float* w) {
    const int tid = threadIdx.x;
    int b = block;
    const int tx0 = (b % tiles_x) * CS_TW; b /= tiles_x;
    const int ty0 = (b % tiles_y) * CS_TH;
    const int n = b / tiles_y;
    if (tid < 98) w[tid] = p.w_sa[tid];
    const int G = p.Cp >> 3, PPB = 256 / G;
    const int lg = tid & (G - 1), pl = tid / G;
    const bool f32 = p.fmt == FMT_F32;
    float cn[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) cn[i] = p.ca[(size_t)n * p.Cp + lg * 8 + i];
    const size_t img = (size_t)n * p.H * p.W;
    // ---- maps of the halo'd tile (zero outside the image: the 7x7 convolution's padding); four pixels per thread in
    // flight: a single dependent load per loop trip left the kernel latency-bound ----
    for (int q0 = pl; q0 < CS_HH * CS_HW; q0 += 4 * PPB) {
        uint4 h4[4], l4[4];
        bool inside[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int q = q0 + k * PPB;
            const int hy = q / CS_HW, hx = q - hy * CS_HW;
            const int Y = ty0 - 3 + hy, X = tx0 - 3 + hx;
            inside[k] = q < CS_HH * CS_HW && Y >= 0 && Y < p.H && X >= 0 && X < p.W;
            h4[k] = make_uint4(0, 0, 0, 0); l4[k] = make_uint4(0, 0, 0, 0);
            if (inside[k] && lg * 8 < p.C) {
                const char* a = p.x + (img + (size_t)Y * p.W + X) * (size_t)(p.Cp * 4) + lg * 32;
                h4[k] = *reinterpret_cast<const uint4*>(a);
                l4[k] = *reinterpret_cast<const uint4*>(a + 16);
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int q = q0 + k * PPB;
            float s = 0.f, m = -INFINITY;
            if (inside[k] && lg * 8 < p.C) {
                float v[8];
                join8_fmt(h4[k], l4[k], v, f32);
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (lg * 8 + i < p.C) { const float u = v[i] * cn[i]; s += u; m = fmaxf(m, u); }
            }
            for (int off = G >> 1; off >= 1; off >>= 1) { s += __shfl_xor(s, off); m = fmaxf(m, __shfl_xor(m, off)); }
            if (lg == 0 && q < CS_HH * CS_HW) { mp[q * 2] = inside[k] ? s / (float)p.C : 0.f; mp[q * 2 + 1] = inside[k] ? m : 0.f; }
        }
    }
    __syncthreads();
    // ---- spatial attention + apply ----
    for (int q = pl; q < CS_TH * CS_TW; q += PPB) {
        const int py = q / CS_TW, px = q - py * CS_TW;
        const int Y = ty0 + py, X = tx0 + px;
        float part = 0.f;
        for (int t = lg; t < 49; t += G) {
            const int ky = t / 7, kx = t - ky * 7;
            const float* mq = mp + ((py + ky) * CS_HW + px + kx) * 2;
            part += w[t] * mq[0] + w[49 + t] * mq[1];
        }
        for (int off = G >> 1; off >= 1; off >>= 1) part += __shfl_xor(part, off);
        if (Y >= p.H || X >= p.W) continue;
        const float sa = sigmoidf(part);
        const size_t idx = img + (size_t)Y * p.W + X;
        const char* a = p.x + idx * (size_t)(p.Cp * 4) + lg * 32;
        float v[8];
        load8_fmt(a, v, f32);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (sa * cn[i]) * v[i];
        if (p.res) {
            const char* r = p.res + idx * (size_t)(p.Cp * 4) + lg * 32;
            float rv[8];
            load8_fmt(r, rv, f32);
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] += rv[i];
        }
        if (p.relu) {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = fmaxf(v[i], 0.f);
        }
        store8_fmt(p.y + idx * (size_t)p.y_pix_bytes + ((p.y_c0 >> 3) + lg) * 32, v, f32);
    }
}

__global__ __launch_bounds__(256) void cbam_spatial_kernel(CbamApplyParams p, int tiles_x, int tiles_y) {
    __shared__ float mp[CS_HH * CS_HW * 2];
    __shared__ float w[98];
    cbam_spatial_body(p, tiles_x, tiles_y, (int)blockIdx.x, mp, w);
}

// ---- the same launches for several tensors at once: the CBAM blocks of the branches of an HRModule at one depth are
// independent (seg_hrnet3.py: every branch is its own Sequential of BasicBlocks), and on the 32x32 and 16x16 branches each of
// these kernels is a few microseconds of pure launch latency.  Workgroups [start[j], start[j+1]) run job j exactly as its own
// launch would. ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cbam_jobs_kernel(CbamJobs jobs) {
    __shared__ float sh[4096];
    __shared__ float sw[128];
    const int b = (int)blockIdx.x;
    int j = 0;
#pragma unroll
    for (int k = 1; k < CBAM_MAXJOBS; ++k) j += (k < jobs.n && b >= jobs.start[k]) ? 1 : 0;
    const CbamJob& q = jobs.j[j];
    const int bid = b - jobs.start[j];
    switch (q.kind) {
        case CBAM_POOL: pool_partial_body(q.ap.x, q.partial, q.HW, q.ap.Cp, q.P, bid % q.P, bid / q.P, sh, sh + 2048, q.ap.fmt == FMT_F32); break;
        case CBAM_MLP: ca_mlp_body(q.partial, q.w0, q.w2, q.ca, q.HW, q.ap.C, q.ap.Cp, q.Cr, q.P, bid, sh, sh + 512, sh + 1024, sh + 1088); break;
        case CBAM_MAPS: cbam_maps_body(q.ap.x, q.ap.ca, q.maps, (long long)q.ap.N * q.HW, q.HW, q.ap.C, q.ap.Cp, bid, sh, sh + 256, q.ap.fmt == FMT_F32); break;
        case CBAM_APPLY: cbam_apply_body(q.ap, (long long)q.ap.N * q.HW, bid, sw, sh); break;
        default: cbam_spatial_body(q.ap, q.tiles_x, q.tiles_y, bid, sh, sw); break;
    }
}

// ---- resample_slice: thread = (dst pixel, 8-channel group of the source) ---------------------------------
struct LerpR { int i0, i1; float l0, l1; };
__device__ __forceinline__ LerpR lerp_any(int dst, int in, int out, int align) {
    float src;
    if (align) {
        const float scale = out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
        src = scale * (float)dst;
    } else {
        const float scale = (float)in / (float)out;
        src = scale * ((float)dst + 0.5f) - 0.5f;
        src = src < 0.f ? 0.f : src;
    }
    LerpR r;
    r.i0 = min((int)src, in - 1);
    r.i1 = r.i0 + (r.i0 < in - 1 ? 1 : 0);
    r.l1 = src - (float)r.i0;
    r.l0 = 1.f - r.l1;
    return r;
}

__global__ __launch_bounds__(256) void resample_slice_kernel(ResampleParams p, long long total) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int G = (p.C + 7) >> 3;             // only the groups that hold real channels
    const int c8 = (int)(idx % G);
    long long pix = idx / G;
    const int x = (int)(pix % p.W);
    long long row = pix / p.W;
    const int y = (int)(row % p.H);
    const int n = (int)(row / p.H);
    float v[8];
    const bool f32 = p.fmt == FMT_F32;
    auto ld = [&](size_t sp, float* out) { load8_fmt(p.x + sp * (size_t)(p.Cp_src * 4) + c8 * 32, out, f32); };
    if (p.h == p.H && p.w == p.W) {
        ld((size_t)pix, v);
    } else {
        const LerpR ly = lerp_any(y, p.h, p.H, p.align), lx = lerp_any(x, p.w, p.W, p.align);
        const size_t r0 = ((size_t)n * p.h + ly.i0) * p.w, r1 = ((size_t)n * p.h + ly.i1) * p.w;
        float v00[8], v01[8], v10[8], v11[8];
        ld(r0 + lx.i0, v00); ld(r0 + lx.i1, v01); ld(r1 + lx.i0, v10); ld(r1 + lx.i1, v11);
#pragma unroll
        for (int i = 0; i < 8; ++i)
            v[i] = ly.l0 * (lx.l0 * v00[i] + lx.l1 * v01[i]) + ly.l1 * (lx.l0 * v10[i] + lx.l1 * v11[i]);
    }
    store8_fmt(p.y + (size_t)pix * (size_t)p.y_pix_bytes + ((p.y_c0 >> 3) + c8) * 32, v, f32);
}

__global__ __launch_bounds__(256) void zero_slice_kernel(char* y, long long npix, int y_pix_bytes, int c0, int ngroups) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= npix * ngroups) return;
    const long long pix = idx / ngroups;
    const int g = (int)(idx - pix * ngroups);
    char* o = y + (size_t)pix * (size_t)y_pix_bytes + ((c0 >> 3) + g) * 32;
    *reinterpret_cast<uint4*>(o) = make_uint4(0, 0, 0, 0);
    *reinterpret_cast<uint4*>(o + 16) = make_uint4(0, 0, 0, 0);
}

inline int blocks(long long total) { return (int)((total + 255) / 256); }

}  // namespace

int launch_pool_partial(const char* x, float* partial, int N, int HW, int Cp, int P, hipStream_t s, int fmt) {
    if ((Cp & 7) || Cp > 256 || Cp < 8 || (fmt != FMT_SB && fmt != FMT_F32)) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(pool_partial_kernel, dim3(P, N), dim3(256), 0, s, x, partial, HW, Cp, P, fmt == FMT_F32 ? 1 : 0);
    return (int)hipGetLastError();
}

int launch_ca_mlp(const float* partial, const float* w0, const float* w2, float* ca, int N, int HW, int C, int Cp,
                  int Cr, int P, hipStream_t s) {
    if (Cp > 512 || Cr > 64 || Cr < 1) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(ca_mlp_kernel, dim3(N), dim3(256), 0, s, partial, w0, w2, ca, HW, C, Cp, Cr, P);
    return (int)hipGetLastError();
}

int launch_cbam_maps(const char* x, const float* ca, float* maps, int N, int HW, int C, int Cp, hipStream_t s, int fmt) {
    const long long npix = (long long)N * HW;
    if ((Cp & 7) || Cp > 2048 || Cp < 8 || (fmt != FMT_SB && fmt != FMT_F32)) return (int)hipErrorInvalidValue;
    const int ppb = 256 / (Cp >> 3);
    hipLaunchKernelGGL(cbam_maps_kernel, dim3((unsigned)((npix + ppb - 1) / ppb)), dim3(256), 0, s, x, ca, maps, npix, HW, C, Cp, fmt == FMT_F32 ? 1 : 0);
    return (int)hipGetLastError();
}

int launch_cbam_apply(const CbamApplyParams& p, hipStream_t s) {
    const long long npix = (long long)p.N * p.H * p.W;
    if ((p.y_c0 & 7) || (p.Cp & 7) || p.Cp > 2048 || p.Cp < 8 || (p.fmt != FMT_SB && p.fmt != FMT_F32)) return (int)hipErrorInvalidValue;
    const int ppb = 256 / (p.Cp >> 3);
    hipLaunchKernelGGL(cbam_apply_kernel, dim3((unsigned)((npix + ppb - 1) / ppb)), dim3(256), 0, s, p, npix);
    return (int)hipGetLastError();
}

// workgroups job `q` needs (the grid its own launch would use); -1: invalid
long long cbam_job_blocks(CbamJob& q) {
    const int Cp = q.ap.Cp;
    if ((Cp & 7) || Cp < 8 || (q.ap.fmt != FMT_SB && q.ap.fmt != FMT_F32)) return -1;
    switch (q.kind) {
        case CBAM_POOL: return Cp > 256 ? -1 : (long long)q.P * q.ap.N;
        case CBAM_MLP: return (Cp > 512 || q.Cr > 64 || q.Cr < 1) ? -1 : q.ap.N;
        case CBAM_MAPS: { if (Cp > 2048) return -1; const int ppb = 256 / (Cp >> 3); return ((long long)q.ap.N * q.HW + ppb - 1) / ppb; }
        case CBAM_APPLY: { if (Cp > 2048 || (q.ap.y_c0 & 7)) return -1; const int ppb = 256 / (Cp >> 3); return ((long long)q.ap.N * q.HW + ppb - 1) / ppb; }
        case CBAM_SPATIAL:
            if (!cbam_spatial_supported(Cp) || (q.ap.y_c0 & 7) || q.ap.C < 1 || q.ap.C > Cp) return -1;
            q.tiles_x = (q.ap.W + CS_TW - 1) / CS_TW; q.tiles_y = (q.ap.H + CS_TH - 1) / CS_TH;
            return (long long)q.tiles_x * q.tiles_y * q.ap.N;
    }
    return -1;
}

int launch_cbam_jobs(const CbamJob* js, int n, hipStream_t s) {
    if (n < 1 || n > CBAM_MAXJOBS) return (int)hipErrorInvalidValue;
    CbamJobs jobs{};
    long long at = 0;
    for (int k = 0; k < n; ++k) {
        jobs.j[k] = js[k];
        const long long nb = cbam_job_blocks(jobs.j[k]);
        if (nb <= 0) return (int)hipErrorInvalidValue;
        jobs.start[k] = (int)at;
        at += nb;
        if (at > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    }
    for (int k = n; k <= CBAM_MAXJOBS; ++k) jobs.start[k] = (int)at;
    jobs.n = n;
    hipLaunchKernelGGL(cbam_jobs_kernel, dim3((unsigned)at), dim3(256), 0, s, jobs);
    return (int)hipGetLastError();
}

bool cbam_spatial_supported(int Cp) {
    const int G = Cp >> 3;
    return (Cp & 7) == 0 && G >= 1 && G <= 32 && (G & (G - 1)) == 0;
}

int launch_cbam_spatial(const CbamApplyParams& p, hipStream_t s) {
    if (!cbam_spatial_supported(p.Cp) || (p.y_c0 & 7) || p.C < 1 || p.C > p.Cp || (p.fmt != FMT_SB && p.fmt != FMT_F32)) return (int)hipErrorInvalidValue;
    const int tiles_x = (p.W + CS_TW - 1) / CS_TW, tiles_y = (p.H + CS_TH - 1) / CS_TH;
    const long long nblk = (long long)tiles_x * tiles_y * p.N;
    if (nblk <= 0 || nblk > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(cbam_spatial_kernel, dim3((unsigned)nblk), dim3(256), 0, s, p, tiles_x, tiles_y);
    return (int)hipGetLastError();
}

int launch_resample_slice(const ResampleParams& p, hipStream_t s) {
    if ((p.y_c0 & 7) || (p.Cp_src & 7) || (p.fmt != FMT_SB && p.fmt != FMT_F32)) return (int)hipErrorInvalidValue;
    const long long total = (long long)p.N * p.H * p.W * ((p.C + 7) >> 3);
    hipLaunchKernelGGL(resample_slice_kernel, dim3(blocks(total)), dim3(256), 0, s, p, total);
    return (int)hipGetLastError();
}

int launch_zero_slice(char* y, long long npix, int y_pix_bytes, int c0, int nchan, hipStream_t s) {
    if ((c0 & 7) || (nchan & 7) || nchan <= 0) return (int)hipErrorInvalidValue;
    const long long total = npix * (nchan >> 3);
    hipLaunchKernelGGL(zero_slice_kernel, dim3(blocks(total)), dim3(256), 0, s, y, npix, y_pix_bytes, c0, nchan >> 3);
    return (int)hipGetLastError();
}

}  // namespace esa
