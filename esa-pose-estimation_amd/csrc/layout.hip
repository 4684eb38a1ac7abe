// layout.hip — f32 NCHW <-> SB (split-bf16 NHWC) conversion.  Used by the per-operator test
// entry points and by esahrnet_tap_read; the network itself enters SB through the stem kernel
// and leaves it through the head kernel.
#include "kernels.h"
#include "sb.h"

namespace esa {
namespace {

template <bool BF>
__global__ __launch_bounds__(256) void nchw_to_sb_kernel(const float* x, int N, int C, int H, int W,
                                                         char* y, int Cp, long long total) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int G = Cp >> 3;
    const int c8 = (int)(idx % G);
    const long long pix = idx / G;
    const long long hw = (long long)H * W;
    const int n = (int)(pix / hw);
    const long long s = pix - (long long)n * hw;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = c8 * 8 + i;
        v[i] = c < C ? x[((size_t)n * C + c) * hw + s] : 0.f;
    }
    if (BF) {
        *reinterpret_cast<uint4*>(y + (size_t)pix * (size_t)(Cp * 2) + c8 * 16) = pack8_bf16(v);
        return;
    }
    uint4 hi, lo;
    split8(v, hi, lo);
    char* o = y + (size_t)pix * (size_t)(Cp * 4) + c8 * 32;
    *reinterpret_cast<uint4*>(o) = hi;
    *reinterpret_cast<uint4*>(o + 16) = lo;
}

template <bool BF>
__global__ __launch_bounds__(256) void sb_to_nchw_kernel(const char* x, int N, int C, int H, int W,
                                                         int Cp, float* y, long long total) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;   // over (n, c, s), s fastest
    if (idx >= total) return;
    const long long hw = (long long)H * W;
    const long long s = idx % hw;
    const long long nc = idx / hw;
    const int c = (int)(nc % C);
    const int n = (int)(nc / C);
    if (BF) {
        y[idx] = bf16_bits_to_f32(*reinterpret_cast<const unsigned short*>(x + ((size_t)n * hw + s) * (size_t)(Cp * 2) + c * 2));
        return;
    }
    const char* a = x + ((size_t)n * hw + s) * (size_t)(Cp * 4) + (c >> 3) * 32 + (c & 7) * 2;
    const uint32_t hi = *reinterpret_cast<const unsigned short*>(a);
    const uint32_t lo = *reinterpret_cast<const unsigned short*>(a + 16);
    y[idx] = bf16_bits_to_f32(hi) + bf16_bits_to_f32(lo);
}

// SB -> NCHW for the network output: thread = (pixel, 8-channel group), group fastest, so a wave reads whole pixels
// (contiguous 32-byte groups) and writes 64-byte runs of 16 consecutive pixels into each of its planes.  The element-wise
// kernel above (thread = output element) touches a different 128-byte line per lane for two bytes: 0.9 TB/s.
__global__ __launch_bounds__(256) void sb_to_nchw_groups_kernel(const char* x, int C, long long hw, int Cp, float* y, long long total) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int G = (C + 7) >> 3;
    const int g = (int)(idx % G);
    const long long pix = idx / G;                  // n * hw + s
    const long long n = pix / hw, sp = pix - n * hw;
    const char* a = x + (size_t)pix * (size_t)(Cp * 4) + g * 32;
    float v[8];
    join8(*reinterpret_cast<const uint4*>(a), *reinterpret_cast<const uint4*>(a + 16), v);
    float* o = y + ((size_t)n * C + (size_t)g * 8) * (size_t)hw + sp;
#pragma unroll
    for (int j = 0; j < 8; ++j)
        if (g * 8 + j < C) o[(size_t)j * hw] = v[j];
}

// plain f32 NHWC <-> f32 NCHW: thread = (pixel, 8-channel group), group fastest (a pixel's channels are contiguous)
__global__ __launch_bounds__(256) void nchw_to_f32_kernel(const float* x, int C, long long hw, float* y, int Cp, long long total) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int G = Cp >> 3;
    const int c8 = (int)(idx % G);
    const long long pix = idx / G;
    const long long n = pix / hw, s = pix - n * hw;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = c8 * 8 + i;
        v[i] = c < C ? x[((size_t)n * C + c) * hw + s] : 0.f;
    }
    float* o = y + (size_t)pix * Cp + c8 * 8;
    *reinterpret_cast<f32x4*>(o) = f32x4{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(o + 4) = f32x4{v[4], v[5], v[6], v[7]};
}
__global__ __launch_bounds__(256) void f32_to_nchw_kernel(const float* x, int C, long long hw, int Cp, float* y, long long total) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int G = (C + 7) >> 3;
    const int g = (int)(idx % G);
    const long long pix = idx / G;
    const long long n = pix / hw, sp = pix - n * hw;
    const float* a = x + (size_t)pix * Cp + g * 8;
    const f32x4 a0 = *reinterpret_cast<const f32x4*>(a), a1 = *reinterpret_cast<const f32x4*>(a + 4);
    const float v[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
    float* o = y + ((size_t)n * C + (size_t)g * 8) * (size_t)hw + sp;
#pragma unroll
    for (int j = 0; j < 8; ++j)
        if (g * 8 + j < C) o[(size_t)j * hw] = v[j];
}

}  // namespace

int launch_nchw_to_f32(const float* x, int N, int C, int H, int W, char* y, int Cp, hipStream_t s) {
    const long long total = (long long)N * H * W * (Cp >> 3);
    const long long nblk = (total + 255) / 256;
    if (nblk <= 0 || nblk > 0x7fffffffLL || (Cp & 7) || C > Cp) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(nchw_to_f32_kernel, dim3((unsigned)nblk), dim3(256), 0, s, x, C, (long long)H * W, reinterpret_cast<float*>(y), Cp, total);
    return (int)hipGetLastError();
}

int launch_f32_to_nchw(const char* x, int N, int C, int H, int W, int Cp, float* y, hipStream_t s) {
    const long long total = (long long)N * H * W * ((C + 7) >> 3);
    const long long nblk = (total + 255) / 256;
    if (nblk <= 0 || nblk > 0x7fffffffLL || C > Cp || (Cp & 7)) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(f32_to_nchw_kernel, dim3((unsigned)nblk), dim3(256), 0, s, reinterpret_cast<const float*>(x), C, (long long)H * W, Cp, y, total);
    return (int)hipGetLastError();
}

int launch_nchw_to_fmt(int fmt, const float* x, int N, int C, int H, int W, char* y, int Cp, hipStream_t s) {
    return fmt == FMT_BF ? launch_nchw_to_bf(x, N, C, H, W, y, Cp, s) : fmt == FMT_F32 ? launch_nchw_to_f32(x, N, C, H, W, y, Cp, s)
                                                                                          : launch_nchw_to_sb(x, N, C, H, W, y, Cp, s);
}
int launch_fmt_to_nchw(int fmt, const char* x, int N, int C, int H, int W, int Cp, float* y, hipStream_t s) {
    return fmt == FMT_BF ? launch_bf_to_nchw(x, N, C, H, W, Cp, y, s) : fmt == FMT_F32 ? launch_f32_to_nchw(x, N, C, H, W, Cp, y, s)
                                                                                          : launch_sb_to_nchw(x, N, C, H, W, Cp, y, s);
}

int launch_nchw_to_sb(const float* x, int N, int C, int H, int W, char* y, int Cp, hipStream_t s) {
    const long long total = (long long)N * H * W * (Cp >> 3);
    const long long nblk = (total + 255) / 256;
    if (nblk <= 0 || nblk > 0x7fffffffLL || (Cp & 7) || C > Cp) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(nchw_to_sb_kernel<false>, dim3((unsigned)nblk), dim3(256), 0, s, x, N, C, H, W, y, Cp, total);
    return (int)hipGetLastError();
}

int launch_nchw_to_bf(const float* x, int N, int C, int H, int W, char* y, int Cp, hipStream_t s) {
    const long long total = (long long)N * H * W * (Cp >> 3);
    const long long nblk = (total + 255) / 256;
    if (nblk <= 0 || nblk > 0x7fffffffLL || (Cp & 7) || C > Cp) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(nchw_to_sb_kernel<true>, dim3((unsigned)nblk), dim3(256), 0, s, x, N, C, H, W, y, Cp, total);
    return (int)hipGetLastError();
}

int launch_sb_to_nchw(const char* x, int N, int C, int H, int W, int Cp, float* y, hipStream_t s) {
    const long long total = (long long)N * H * W * ((C + 7) >> 3);
    const long long nblk = (total + 255) / 256;
    if (nblk <= 0 || nblk > 0x7fffffffLL || C > Cp || (Cp & 7)) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(sb_to_nchw_groups_kernel, dim3((unsigned)nblk), dim3(256), 0, s, x, C, (long long)H * W, Cp, y, total);
    return (int)hipGetLastError();
}

int launch_bf_to_nchw(const char* x, int N, int C, int H, int W, int Cp, float* y, hipStream_t s) {
    const long long total = (long long)N * C * H * W;
    const long long nblk = (total + 255) / 256;
    if (nblk <= 0 || nblk > 0x7fffffffLL || C > Cp) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(sb_to_nchw_kernel<true>, dim3((unsigned)nblk), dim3(256), 0, s, x, N, C, H, W, Cp, y, total);
    return (int)hipGetLastError();
}

}  // namespace esa
