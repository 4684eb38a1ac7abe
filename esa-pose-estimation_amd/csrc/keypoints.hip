// keypoints.hip — fused arg-max + sub-pixel refine:  f32 [planes][H][W] -> f32 [planes][3].
//
// Replaces (reference): the caller's two-stage torch.max + >=150 blocking .item() reads per
// image and the K x H x W device-to-host copy (demo.py:172-185, val.py:151-166), then
// inference.get_final -> my_taylor on the host (inference.py:136-152, 75-94).  One workgroup
// (16 waves) per (n,k) plane: every lane keeps (value, first index) over a strided 16-B sweep, a 64-lane
// DPP/shuffle reduction and one LDS step across the waves pick the first row-major maximum
// (== np.argmax / two-stage torch.max tie-break), then one lane evaluates the 9-tap
// log-quadratic offset in f64 exactly as the reference's Python floats do.
// NaN policy = the reference's: np.argmax / torch.max treat NaN as the maximum and return the
// FIRST NaN's index, the reported peak is NaN, and the refinement falls through (np.maximum(hm, 1e-10) and
// math.log propagate the NaN, `offset < 1` is then false): integer coordinates of the first NaN, peak NaN.
#include "kernels.h"

namespace esa {
namespace {

__device__ __forceinline__ void take(float v, int i, float& bv, int& bi) { argmax_take(v, i, bv, bi); }      // kernels.h

// integer peak (bi) of plane `pl` -> sub-pixel keypoint: the 9-tap log-quadratic offset in f64 exactly as the
// reference's Python floats do (inference.py:75-94, 136-152)
__device__ __forceinline__ void refine_and_store(const float* pl, int H, int W, int bi, float* kp3, int* idx_slot) {
    if (bi == 0x7fffffff) bi = 0;                        // all-NaN / all -inf plane
    const int px = bi % W, py = bi / W;
    float fx = (float)px, fy = (float)py;
    if (1 < px && px < W - 2 && 1 < py && py < H - 2) {   // inference.py:81
        // np.maximum(hm, 1e-10) of inference.py:141 (NaN-propagating, unlike fmaxf), then math.log in f64
        auto lg = [&](int yy, int xx) {
            const float v = pl[yy * W + xx];
            return log((double)(v < 1e-10f ? 1e-10f : v));
        };
        const double c = lg(py, px);
        const double hx = 0.5 * (lg(py, px + 1) - lg(py, px - 1));
        const double hy = 0.5 * (lg(py + 1, px) - lg(py - 1, px));
        const double hxx = 0.25 * (lg(py, px + 2) - 2 * c + lg(py, px - 2));
        const double hyy = 0.25 * (lg(py + 2, px) - 2 * c + lg(py - 2, px));
        if (hxx != 0 && hyy != 0) {
            const double ox = -hx / hxx, oy = -hy / hyy;
            if (ox < 1 && oy < 1) {                        // signed, both-or-neither (:92)
                fx = (float)((double)fx + ox);
                fy = (float)((double)fy + oy);
            }
        }
    }
    kp3[0] = fx;
    kp3[1] = fy;
    kp3[2] = pl[bi];
    if (idx_slot) *idx_slot = bi;
}

constexpr int KT = 1024;        // 16 waves per plane: the sweep is latency-bound, it wants loads in flight

__global__ __launch_bounds__(KT) void keypoints_kernel(const float* heat, int H, int W, float* kp, int* idx_out) {
    __shared__ float sv[KT / 64];
    __shared__ int si[KT / 64];
    const float* pl = heat + (size_t)blockIdx.x * H * W;
    const int total = H * W;
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    const int nvec = total >> 2;
    const bool aligned = ((reinterpret_cast<uintptr_t>(pl) & 15) == 0);
    if (aligned) {
        for (int v4 = threadIdx.x; v4 < nvec; v4 += KT) {
            const float4 q = reinterpret_cast<const float4*>(pl)[v4];
            take(q.x, v4 * 4 + 0, bv, bi);
            take(q.y, v4 * 4 + 1, bv, bi);
            take(q.z, v4 * 4 + 2, bv, bi);
            take(q.w, v4 * 4 + 3, bv, bi);
        }
        for (int i = nvec * 4 + threadIdx.x; i < total; i += KT) take(pl[i], i, bv, bi);
    } else {
        for (int i = threadIdx.x; i < total; i += KT) take(pl[i], i, bv, bi);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float ov = __shfl_xor(bv, off);
        const int oi = __shfl_xor(bi, off);
        take(ov, oi, bv, bi);
    }
    if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = bv; si[threadIdx.x >> 6] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < KT / 64; ++w) take(sv[w], si[w], bv, bi);
        refine_and_store(pl, H, W, bi, kp + (size_t)blockIdx.x * 3, idx_out ? idx_out + blockIdx.x : nullptr);
    }
}

// Finish over the per-tile maxima the output-layer kernel left behind (head.hip, FinalParams::part): one wave per plane
// reduces `ntiles` (value, index) pairs with the same ordering as the full sweep above, then refines on the heat-map —
// the 92 MB of heat-maps of a 32-crop batch are not read a second time.
__global__ __launch_bounds__(64) void keypoints_finish_kernel(const float* heat, const float2* part, int ntiles, int H, int W,
                                                              float* kp, int* idx_out) {
    const float2* pp = part + (size_t)blockIdx.x * ntiles;
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int t = threadIdx.x; t < ntiles; t += 64) {
        const float2 q = pp[t];
        take(q.x, __float_as_int(q.y), bv, bi);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float ov = __shfl_xor(bv, off);
        const int oi = __shfl_xor(bi, off);
        take(ov, oi, bv, bi);
    }
    if (threadIdx.x == 0)
        refine_and_store(heat + (size_t)blockIdx.x * H * W, H, W, bi, kp + (size_t)blockIdx.x * 3, idx_out ? idx_out + blockIdx.x : nullptr);
}

}  // namespace

int launch_keypoints_finish(const float* heat, const float2* part, int ntiles, int planes, int H, int W, float* kp, int* idx_out,
                            hipStream_t stream) {
    if (planes <= 0 || ntiles <= 0 || H <= 0 || W <= 0 || (long long)H * W > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(keypoints_finish_kernel, dim3((unsigned)planes), dim3(64), 0, stream, heat, part, ntiles, H, W, kp, idx_out);
    return (int)hipGetLastError();
}

int launch_keypoints(const float* heat, int planes, int H, int W, float* kp, int* idx_out, hipStream_t stream) {
    if (planes <= 0 || H <= 0 || W <= 0 || (long long)H * W > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(keypoints_kernel, dim3((unsigned)planes), dim3(KT), 0, stream, heat, H, W, kp, idx_out);
    return (int)hipGetLastError();
}

}  // namespace esa
