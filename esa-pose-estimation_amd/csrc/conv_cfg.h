// conv_cfg.h — tile geometry shared by the MFMA convolution kernels (conv_mfma.hip, stem_fused.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace esa {

constexpr int TW = 16;        // output columns per workgroup tile = one MFMA N-tile
constexpr int NTHREADS = 256;

// 16 B per lane global -> LDS DMA; LDS destination = wave-uniform base (+ lane*16 by hardware).
__device__ __forceinline__ void dma16(const void* gsrc, char* lds_uniform_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_uniform_base, 16, 0, 0);
}

template <int KS, int S, int TH, int MT>
struct ConvCfg {
    static constexpr int PAD = (KS - 1) / 2;
    static constexpr int TAPS = KS * KS;
    static constexpr int NT = TH / 4;                       // pixel-row tiles per wave
    static constexpr int IH = (TH - 1) * S + KS;
    static constexpr int IW = (TW - 1) * S + KS;
    static constexpr int NPIX = IH * IW;
    static constexpr int PLANE = ((NPIX * 16 + 128 + 255) / 256) * 256;
    static constexpr int XBYTES = 8 * PLANE;
    static constexpr int WBYTES = TAPS * MT * 2048;
    static constexpr int XUNITS = ((NPIX + 7) / 8) * 64;    // 16-B units, whole 8-pixel groups
    static constexpr int XITER = (XUNITS + NTHREADS - 1) / NTHREADS;
    static constexpr int WUNITS = TAPS * MT * 128;
    static constexpr int WITER = (WUNITS + NTHREADS - 1) / NTHREADS;
    static constexpr int LDS_BYTES = XBYTES + WBYTES;
    static constexpr int LO_OFF = PLANE + (S == 2 ? 64 : 16);   // plane_off(2g+1) - plane_off(2g)
    // plane j = 2*g + part (k-group g, part hi/lo).  Skews (16-B slots) are chosen for BOTH sides:
    //  reads  (ds_read_b128, lane groups pair k-groups {0,1} and {2,3}, one part per instruction):
    //         stride 1 needs planes g and g^1 congruent mod 256 B; stride 2 needs them one slot apart;
    //  writes (ds_write_b128, 8 consecutive lanes = the 8 planes of ONE pixel, i.e. one coalesced
    //         128-B global line): want distinct slots mod 128 B -> stride 2: all 8 distinct,
    //         stride 1: 4 distinct (2-way, hidden under the store's 13-cycle issue cost).
    __host__ __device__ static constexpr int plane_off(int j) {
        return j * PLANE + (S == 2 ? ((j >> 1) * 16 + (j & 1) * 64) : (((j >> 2) * 2 + (j & 1)) * 16));
    }
};

}  // namespace esa
