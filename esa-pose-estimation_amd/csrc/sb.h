// sb.h — the internal activation format and small device helpers (gfx950 only).
//
// "SB" = split-bf16 NHWC.  A tensor [N][H][W][Cp] (Cp = channels padded to a multiple of
// 32) stores every value v as hi = bf16(v), lo = bf16(v - hi), i.e. ~16 mantissa bits.
// Per pixel the Cp channels are laid out in groups of 8:  [c8][part(hi=0,lo=1)][8 x bf16],
// 32 bytes per group, Cp*4 bytes per pixel.  One 16-byte chunk is exactly one MFMA
// 16x16x32 bf16 operand fragment of one lane (8 consecutive K elements of one pixel), so
// the convolution kernels move chunks HBM -> LDS -> VGPR without ever repacking them.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Workgroups b and b+8 run on the same XCD (round-robin dispatch over the 8 XCDs, each with its own L2).
// Kernels whose neighbouring tiles share input (halos, interpolation windows, cout-tile siblings) map
// blockIdx through this so that every XCD owns one contiguous run of tiles and the shared lines are
// fetched from HBM / Infinity Cache once instead of once per XCD.
__device__ __forceinline__ int xcd_contiguous(int b, int nblocks) {
    return (nblocks & 7) == 0 ? (b & 7) * (nblocks >> 3) + (b >> 3) : b;
}

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ float bf16_bits_to_f32(uint32_t b) { return __uint_as_float(b << 16); }

// Round-to-nearest-even f32 -> bf16 bits (finite inputs; the network never produces NaN
// from finite weights/inputs).
__device__ __forceinline__ uint32_t f32_to_bf16_bits(float x) {
    __bf16 h = (__bf16)x;
    return (uint32_t)__builtin_bit_cast(unsigned short, h);
}

__device__ __forceinline__ void split_bf16(float v, uint32_t& hi, uint32_t& lo) {
    __bf16 h = (__bf16)v;
    float r = v - (float)h;
    __bf16 l = (__bf16)r;
    hi = (uint32_t)__builtin_bit_cast(unsigned short, h);
    lo = (uint32_t)__builtin_bit_cast(unsigned short, l);
}

typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

// Two floats -> one dword of hi bf16 (a in the low half) + one dword of lo bf16: v_cvt_pk_bf16_f32 on
// the pair, the residuals against the packed hi halves, v_cvt_pk_bf16_f32 again (6 VALU ops per pair;
// same round-to-nearest-even results as split_bf16 element by element).
__device__ __forceinline__ void split2(float a, float b, uint32_t& hi, uint32_t& lo) {
    const f32x2 v = {a, b};
    const uint32_t hb = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
    const f32x2 r = {a - __uint_as_float(hb << 16), b - __uint_as_float(hb & 0xffff0000u)};
    hi = hb;
    lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, bf16x2));
}

// max(v, 0) in one instruction: as signed integers every negative float (and -0) is < 0 and the
// non-negative floats keep their order, so v_max_i32 against 0 is ReLU (fmaxf costs a second,
// canonicalising v_max for values the compiler cannot prove quiet, e.g. MFMA results).
__device__ __forceinline__ float relu1(float v) {
    const int b = __builtin_bit_cast(int, v);
    return __builtin_bit_cast(float, b > 0 ? b : 0);
}

// Optional ReLU without a branch: floor = relu_floor(flag) is 0 (ReLU) or INT_MIN (identity), one v_max_i32.
// Keep it branch-free on purpose: with `if (relu) {...}` right behind the last MFMA of an accumulator, hipcc
// (ROCm 7.2) pads the MFMA -> VALU read hazard with s_nop only on the taken path, and the fall-through path
// reads the accumulator registers too early (observed: wrong low-order bits in the first two channels).
// (readfirstlane hides the two-valued origin of the floor: otherwise hipcc turns the max back into max-against-0
// plus a select)
__device__ __forceinline__ int relu_floor(int relu) { return relu ? 0 : (int)0x80000000; }
__device__ __forceinline__ float relu_opt(float v, int floor) {
    // __builtin_elementwise_max keeps it ONE v_max_i32; written as `b > floor ? b : floor` hipcc emits
    // v_cmp + s_nop + v_cndmask per element (12 instead of 4 instructions per accumulator quad)
    return __builtin_bit_cast(float, __builtin_elementwise_max(__builtin_bit_cast(int, v), floor));
}

// 4 floats -> 8 bytes of hi bf16 + 8 bytes of lo bf16
__device__ __forceinline__ void split4(const float v[4], uint2& hi, uint2& lo) {
    split2(v[0], v[1], hi.x, lo.x);
    split2(v[2], v[3], hi.y, lo.y);
}

// Accumulator quad <-> SB memory with 16-byte accesses.  An MFMA D tile leaves, in the lanes of 16-lane
// row q, 4 consecutive channels (hi 8 B, lo 8 B) of one pixel; rows q and q^1 together hold one
// 8-channel SB group [hi 16 B | lo 16 B].  v_permlane16_swap exchanges the halves so that the even
// row owns the whole lo chunk and the odd row the whole hi chunk: one 16-byte access per lane instead
// of two 8-byte ones (the 8-byte pattern ran the epilogue stores at ~3 TB/s).  Both helpers must be
// executed by all 64 lanes (predicate the memory access, not the swap).
__device__ __forceinline__ uint4 quad_to_chunk(uint2 hi, uint2 lo) {
    const auto r0 = __builtin_amdgcn_permlane16_swap(lo.x, hi.x, false, false);
    const auto r1 = __builtin_amdgcn_permlane16_swap(lo.y, hi.y, false, false);
    return make_uint4(r0[0], r1[0], r0[1], r1[1]);      // even rows: lo chunk, odd rows: hi chunk
}
__device__ __forceinline__ void chunk_to_quad(uint4 c, uint2& hi, uint2& lo) {
    const auto r0 = __builtin_amdgcn_permlane16_swap(c.x, c.z, false, false);
    const auto r1 = __builtin_amdgcn_permlane16_swap(c.y, c.w, false, false);
    lo = make_uint2(r0[0], r1[0]);
    hi = make_uint2(r0[1], r1[1]);
}
// byte offset inside the pixel of the chunk owned by a lane of row q for the quad of channels co..co+3
__device__ __forceinline__ int chunk_ofs(int co, int q) { return (co >> 3) * 32 + ((q & 1) ? 0 : 16); }

__device__ __forceinline__ void join4(uint2 hi, uint2 lo, float v[4]) {
    v[0] = bf16_bits_to_f32(hi.x & 0xffffu) + bf16_bits_to_f32(lo.x & 0xffffu);
    v[1] = bf16_bits_to_f32(hi.x >> 16) + bf16_bits_to_f32(lo.x >> 16);
    v[2] = bf16_bits_to_f32(hi.y & 0xffffu) + bf16_bits_to_f32(lo.y & 0xffffu);
    v[3] = bf16_bits_to_f32(hi.y >> 16) + bf16_bits_to_f32(lo.y >> 16);
}

// 8 floats <-> one 32-byte channel group (hi chunk, lo chunk)
__device__ __forceinline__ void split8(const float v[8], uint4& hi, uint4& lo) {
    uint2 h0, l0, h1, l1;
    split4(v, h0, l0);
    split4(v + 4, h1, l1);
    hi = make_uint4(h0.x, h0.y, h1.x, h1.y);
    lo = make_uint4(l0.x, l0.y, l1.x, l1.y);
}

__device__ __forceinline__ void join8(uint4 hi, uint4 lo, float v[8]) {
    join4(make_uint2(hi.x, hi.y), make_uint2(lo.x, lo.y), v);
    join4(make_uint2(hi.z, hi.w), make_uint2(lo.z, lo.w), v + 4);
}

// 8 channels of a pixel are 32 bytes in SB ([hi 16 B | lo 16 B]) and in the fp32-grade mode's plain f32 NHWC alike; the
// memory-bound kernels (cbam.hip, head_gather.hip, the raw stem) serve both through these (f32: a uniform run-time flag)
__device__ __forceinline__ void join8_fmt(uint4 a, uint4 b, float v[8], bool f32) {
    if (f32) {
        v[0] = __uint_as_float(a.x); v[1] = __uint_as_float(a.y); v[2] = __uint_as_float(a.z); v[3] = __uint_as_float(a.w);
        v[4] = __uint_as_float(b.x); v[5] = __uint_as_float(b.y); v[6] = __uint_as_float(b.z); v[7] = __uint_as_float(b.w);
    } else {
        join8(a, b, v);
    }
}
__device__ __forceinline__ void load8_fmt(const char* p, float v[8], bool f32) {
    join8_fmt(*reinterpret_cast<const uint4*>(p), *reinterpret_cast<const uint4*>(p + 16), v, f32);
}
__device__ __forceinline__ void store8_fmt(char* p, const float v[8], bool f32) {
    uint4 a, b;
    if (f32) {
        a = make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3]));
        b = make_uint4(__float_as_uint(v[4]), __float_as_uint(v[5]), __float_as_uint(v[6]), __float_as_uint(v[7]));
    } else {
        split8(v, a, b);
    }
    *reinterpret_cast<uint4*>(p) = a;
    *reinterpret_cast<uint4*>(p + 16) = b;
}

// ---- "BF" = single bf16 NHWC (esahrnet_cfg.precision == 1, BASELINE configs[3]) ---------------------------------
// A tensor [N][H][W][Cp] of plain bf16, Cp = channels padded to a multiple of 64 with exact zeros: 2 bytes per
// channel, half of SB.  A pixel's channels come in 64-channel blocks of 128 bytes = 8 chunks of 16 bytes (8
// consecutive channels each), natural order.  The convolution kernels stage one 128-byte block per pixel exactly as
// they stage one 32-channel SB chunk (same loads, same 8 LDS planes); what used to be the (hi, lo) plane pair of
// k-group g now holds the two MFMA K-steps of the block — chunk g (channels 8g..8g+7) and chunk 4+g (channels
// 32+8g..) — so a block costs 2 MFMAs per tap and row where the split format costs 3 for half as many channels.
__host__ __device__ __forceinline__ constexpr int bf_plane_of_chunk(int jst) { return 2 * (jst & 3) + (jst >> 2); }

// 4 floats -> 4 bf16 (8 bytes), round to nearest even (v_cvt_pk_bf16_f32)
__device__ __forceinline__ uint2 pack4_bf16(const float v[4]) {
    const f32x2 a = {v[0], v[1]}, b = {v[2], v[3]};
    return make_uint2(__builtin_bit_cast(uint32_t, __builtin_convertvector(a, bf16x2)),
                      __builtin_bit_cast(uint32_t, __builtin_convertvector(b, bf16x2)));
}
__device__ __forceinline__ void unpack4_bf16(uint2 c, float v[4]) {
    v[0] = __uint_as_float(c.x << 16);
    v[1] = __uint_as_float(c.x & 0xffff0000u);
    v[2] = __uint_as_float(c.y << 16);
    v[3] = __uint_as_float(c.y & 0xffff0000u);
}
__device__ __forceinline__ uint4 pack8_bf16(const float v[8]) {
    const uint2 a = pack4_bf16(v), b = pack4_bf16(v + 4);
    return make_uint4(a.x, a.y, b.x, b.y);
}
__device__ __forceinline__ void unpack8_bf16(uint4 c, float v[8]) {
    unpack4_bf16(make_uint2(c.x, c.y), v);
    unpack4_bf16(make_uint2(c.z, c.w), v + 4);
}
