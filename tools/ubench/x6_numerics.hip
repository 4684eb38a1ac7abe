// x6_numerics.hip — how close to an f32 FMA chain does a K-long dot product get when every f32 operand is split into
// bf16 terms and the products run on v_mfma_f32_16x16x32_bf16 with f32 accumulation?
//   terms 2, products 3 ("bf16x3", the round-1/2 arithmetic)      a0b0 + a0b1 + a1b0
//   terms 3, products 6 ("bf16x6")                                a0b0 + a0b1 + a1b0 + a0b2 + a1b1 + a2b0
//   terms 3, products 9                                           all nine
// against an fp64 reference, next to the error of a sequential f32 fmaf chain (what the reference's CPU convolution
// roughly does) and of v_mfma_f32_16x16x4_f32.  Two data sets: N(0,1) x N(0,1) (signed products) and |N(0,1)| x |N(0,1)|
// (all products positive: a biased accumulator rounding would show as a drift growing with K).
//   hipcc --offload-arch=gfx950 -O3 x6_numerics.hip -o x6_numerics
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

static uint16_t bf16_rne(float v) {
    uint32_t u;
    memcpy(&u, &v, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static float bf16_f32(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
// v = t[0] + t[1] + t[2] exactly (round-to-nearest-even at every stage; 8 + 8 + 8 significand bits)
static void split3(float v, uint16_t t[3], bool trunc) {
    float r = v;
    for (int i = 0; i < 3; ++i) {
        if (trunc) {
            uint32_t u;
            memcpy(&u, &r, 4);
            t[i] = (uint16_t)(u >> 16);
        } else {
            t[i] = bf16_rne(r);
        }
        r -= bf16_f32(t[i]);
    }
}

// A: [3 terms][K/32][64 lanes][8]   lane l: row l&15, k = 32*kk + 8*(l>>4) + j     (16 x K)
// B: the same for the 16 columns
template <int MODE>   // 3, 6, 9 products;  0: f32 MFMA on the unsplit values
__global__ void dot_kernel(const uint16_t* A, const uint16_t* B, const float* Af, const float* Bf, float* D, int K) {
    const int lane = threadIdx.x, prob = blockIdx.x;
    const int ks = K / 32;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (MODE == 7 || MODE == 8) {
    } else if (MODE == 0) {
        // v_mfma_f32_16x16x4_f32: lane l holds A[l&15][k = l>>4], B[k = l>>4][l&15]
        for (int k4 = 0; k4 < K / 4; ++k4) {
            const float a = Af[((size_t)prob * 16 + (lane & 15)) * K + k4 * 4 + (lane >> 4)];
            const float b = Bf[((size_t)prob * 16 + (lane & 15)) * K + k4 * 4 + (lane >> 4)];
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
        }
    } else {
        const size_t term = (size_t)ks * 64 * 8;
        const uint16_t* a0 = A + (size_t)prob * 3 * term;
        const uint16_t* b0 = B + (size_t)prob * 3 * term;
        for (int kk = 0; kk < ks; ++kk) {
            bf16x8 a[3], b[3];
            for (int t = 0; t < 3; ++t) {
                a[t] = *reinterpret_cast<const bf16x8*>(a0 + t * term + ((size_t)kk * 64 + lane) * 8);
                b[t] = *reinterpret_cast<const bf16x8*>(b0 + t * term + ((size_t)kk * 64 + lane) * 8);
            }
            // small products first
            if (MODE >= 9) {
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[1], acc, 0, 0, 0);
            }
            if (MODE >= 6) {
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], acc, 0, 0, 0);
            }
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], acc, 0, 0, 0);
        }
    }
    if (MODE == 7 || MODE == 8) {
        // MODE 7: groups of 3 K-steps (the three ky taps of one kx phase) summed in a FRESH accumulator — the fifteen
        // low-order products first, the three main ones last — and added to the running sum by one VALU add per group.
        // MODE 8: the same with groups of 9 K-steps (a whole 32-channel chunk of a 3x3).
        const int GRP = MODE == 7 ? 3 : 9;
        acc = f32x4{0.f, 0.f, 0.f, 0.f};
        const size_t term = (size_t)ks * 64 * 8;
        const uint16_t* a0 = A + (size_t)prob * 3 * term;
        const uint16_t* b0 = B + (size_t)prob * 3 * term;
        for (int k0 = 0; k0 < ks; k0 += GRP) {
            f32x4 tmp = {0.f, 0.f, 0.f, 0.f};
            for (int pass = 0; pass < 2; ++pass)
                for (int kk = k0; kk < k0 + GRP && kk < ks; ++kk) {
                    bf16x8 a[3], b[3];
                    for (int t = 0; t < 3; ++t) {
                        a[t] = *reinterpret_cast<const bf16x8*>(a0 + t * term + ((size_t)kk * 64 + lane) * 8);
                        b[t] = *reinterpret_cast<const bf16x8*>(b0 + t * term + ((size_t)kk * 64 + lane) * 8);
                    }
                    if (pass == 0) {
                        tmp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], tmp, 0, 0, 0);
                        tmp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], tmp, 0, 0, 0);
                        tmp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], tmp, 0, 0, 0);
                        tmp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], tmp, 0, 0, 0);
                        tmp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], tmp, 0, 0, 0);
                    } else {
                        tmp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], tmp, 0, 0, 0);
                    }
                }
            for (int r = 0; r < 4; ++r) acc[r] += tmp[r];
        }
    }
    // D[row = 4*(lane>>4) + r][col = lane&15]
    for (int r = 0; r < 4; ++r) D[((size_t)prob * 16 + 4 * (lane >> 4) + r) * 16 + (lane & 15)] = acc[r];
}

struct Stat { double maxe = 0, sse = 0, bias = 0, scale = 0; size_t n = 0; };
static void add(Stat& s, double got, double ref, double sabs) {
    const double e = got - ref;
    s.maxe = std::max(s.maxe, std::fabs(e) / sabs);
    s.sse += (e / sabs) * (e / sabs);
    s.bias += e / sabs;
    s.n++;
}
static void show(const char* name, const Stat& s) {
    printf("  %-34s max %.3e  rms %.3e  mean %+.3e   (relative to sum|a*b|)\n", name, s.maxe, std::sqrt(s.sse / s.n), s.bias / s.n);
}

int main() {
    const int P = 64;
    for (int K : {288, 2304, 4320}) {
        for (int positive = 0; positive < 2; ++positive) {
            std::mt19937 rng(1234 + K + positive);
            std::normal_distribution<float> nd(0.f, 1.f);
            std::vector<float> Af((size_t)P * 16 * K), Bf((size_t)P * 16 * K);
            for (auto& v : Af) v = positive ? std::fabs(nd(rng)) : nd(rng);
            for (auto& v : Bf) v = (positive ? std::fabs(nd(rng)) : nd(rng)) * 0.05f;
            const int ks = K / 32;
            const size_t term = (size_t)ks * 64 * 8;
            float* dAf; float* dBf; float* dD;
            hipMalloc(&dAf, Af.size() * 4); hipMalloc(&dBf, Bf.size() * 4); hipMalloc(&dD, (size_t)P * 256 * 4);
            hipMemcpy(dAf, Af.data(), Af.size() * 4, hipMemcpyHostToDevice);
            hipMemcpy(dBf, Bf.data(), Bf.size() * 4, hipMemcpyHostToDevice);
            uint16_t *dA, *dB;
            hipMalloc(&dA, (size_t)P * 3 * term * 2); hipMalloc(&dB, (size_t)P * 3 * term * 2);
            // references
            std::vector<double> ref((size_t)P * 256), sabs((size_t)P * 256);
            std::vector<float> seq((size_t)P * 256);
            for (int p = 0; p < P; ++p)
                for (int r = 0; r < 16; ++r)
                    for (int c = 0; c < 16; ++c) {
                        double s = 0, sa = 0;
                        float f = 0.f;
                        for (int k = 0; k < K; ++k) {
                            const float a = Af[((size_t)p * 16 + r) * K + k], b = Bf[((size_t)p * 16 + c) * K + k];
                            s += (double)a * b; sa += std::fabs((double)a * b);
                            f = fmaf(a, b, f);
                        }
                        ref[(size_t)p * 256 + r * 16 + c] = s; sabs[(size_t)p * 256 + r * 16 + c] = sa;
                        seq[(size_t)p * 256 + r * 16 + c] = f;
                    }
            printf("K = %d, %s products\n", K, positive ? "all-positive" : "signed");
            Stat st;
            for (size_t i = 0; i < ref.size(); ++i) add(st, seq[i], ref[i], sabs[i]);
            show("host sequential f32 fmaf", st);
            {   // 16 interleaved f32 partial sums (a vectorised CPU dot product), combined pairwise
                Stat sb;
                for (int p = 0; p < P; ++p)
                    for (int r = 0; r < 16; ++r)
                        for (int c = 0; c < 16; ++c) {
                            float part[16] = {0};
                            for (int k = 0; k < K; ++k)
                                part[k & 15] = fmaf(Af[((size_t)p * 16 + r) * K + k], Bf[((size_t)p * 16 + c) * K + k], part[k & 15]);
                            for (int w = 8; w >= 1; w >>= 1)
                                for (int i = 0; i < w; ++i) part[i] += part[i + w];
                            add(sb, part[0], ref[(size_t)p * 256 + r * 16 + c], sabs[(size_t)p * 256 + r * 16 + c]);
                        }
                show("host f32, 16 partial sums", sb);
            }
            std::vector<float> D((size_t)P * 256);
            auto run = [&](int mode, const char* name) {
                switch (mode) {
                    case 0: dot_kernel<0><<<P, 64>>>(dA, dB, dAf, dBf, dD, K); break;
                    case 3: dot_kernel<3><<<P, 64>>>(dA, dB, dAf, dBf, dD, K); break;
                    case 6: dot_kernel<6><<<P, 64>>>(dA, dB, dAf, dBf, dD, K); break;
                    case 7: dot_kernel<7><<<P, 64>>>(dA, dB, dAf, dBf, dD, K); break;
                    case 8: dot_kernel<8><<<P, 64>>>(dA, dB, dAf, dBf, dD, K); break;
                    default: dot_kernel<9><<<P, 64>>>(dA, dB, dAf, dBf, dD, K); break;
                }
                hipDeviceSynchronize();
                hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
                Stat s;
                for (size_t i = 0; i < ref.size(); ++i) add(s, D[i], ref[i], sabs[i]);
                show(name, s);
            };
            run(0, "v_mfma_f32_16x16x4_f32");
            for (int trunc = 0; trunc < 2; ++trunc) {
                std::vector<uint16_t> A((size_t)P * 3 * term), B((size_t)P * 3 * term);
                for (int p = 0; p < P; ++p)
                    for (int kk = 0; kk < ks; ++kk)
                        for (int l = 0; l < 64; ++l)
                            for (int j = 0; j < 8; ++j) {
                                const int k = kk * 32 + 8 * (l >> 4) + j;
                                uint16_t ta[3], tb[3];
                                split3(Af[((size_t)p * 16 + (l & 15)) * K + k], ta, trunc);
                                split3(Bf[((size_t)p * 16 + (l & 15)) * K + k], tb, trunc);
                                for (int t = 0; t < 3; ++t) {
                                    A[(size_t)p * 3 * term + t * term + ((size_t)kk * 64 + l) * 8 + j] = ta[t];
                                    B[(size_t)p * 3 * term + t * term + ((size_t)kk * 64 + l) * 8 + j] = tb[t];
                                }
                            }
                hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice);
                hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
                run(3, trunc ? "bf16 x3 (2 terms, truncated)" : "bf16 x3 (2 terms, RNE)");
                run(6, trunc ? "bf16 x6 (3 terms, truncated)" : "bf16 x6 (3 terms, RNE)");
                run(9, trunc ? "bf16 x9 (3 terms, truncated)" : "bf16 x9 (3 terms, RNE)");
                if (!trunc) run(7, "bf16 x6, fresh acc per 3 K-steps");
                if (!trunc) run(8, "bf16 x6, fresh acc per 9 K-steps");
            }
            hipFree(dAf); hipFree(dBf); hipFree(dD); hipFree(dA); hipFree(dB);
        }
    }
    return 0;
}
