// store_data_war.hip — may the data VGPRs of a buffer_store_dwordx4 (SGPR soffset, offen) be overwritten by the very
// next VALU instruction on gfx950?  hipcc pads this hazard only for stores WITHOUT an soffset register.
// Every wave stores a known pattern, overwrites dword W of the data registers K wait states later, and the host
// counts elements in memory that show the overwriting value.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int K, int W, int LOADS>
__global__ __launch_bounds__(256) void k(unsigned* out, const unsigned* cold, unsigned bytes, int iters, unsigned cold_words) {
    const unsigned gid = blockIdx.x * 256 + threadIdx.x;
    __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, (int)bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void*)cold, 0, (int)(cold_words * 4u), 0x00020000);
    unsigned seed = gid * 2654435761u + 99u;
    for (int it = 0; it < iters; ++it) {
        const unsigned off = ((unsigned)it * gridDim.x * 256u + gid) * 16u;       // one 16-byte slot per thread and iteration
        const unsigned soff = 0;
        seed = seed * 1664525u + 1013904223u;
        const unsigned coff = ((seed >> 4) % (cold_words / 4)) * 16u;             // keeps the memory pipeline of the CU busy
        unsigned t0, t1, t2, t3;
        asm volatile(
            "v_mov_b32 v100, 0x11110000\n v_mov_b32 v101, 0x22220000\n v_mov_b32 v102, 0x33330000\n v_mov_b32 v103, 0x44440000\n"
            ".rept %7\n buffer_load_dwordx4 v[104:107], %5, %6, 0 offen\n .endr\n"
            "s_nop 4\n"
            "buffer_store_dwordx4 v[100:103], %0, %1, %2 offen\n"
            ".if %3 > 0\n s_nop %3 - 1\n .endif\n"
            ".if %4 == 0\n v_mov_b32 v100, 0xdead0000\n .endif\n"
            ".if %4 == 1\n v_mov_b32 v101, 0xdead0000\n .endif\n"
            ".if %4 == 2\n v_mov_b32 v102, 0xdead0000\n .endif\n"
            ".if %4 == 3\n v_mov_b32 v103, 0xdead0000\n .endif\n"
            "s_waitcnt vmcnt(0)\n"
            :: "v"(off), "s"(ro), "s"(soff), "n"(K), "n"(W), "v"(coff), "s"(rc), "n"(LOADS)
            : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "memory");
    }
}
unsigned* dout; unsigned* cold; unsigned* hbuf;
const int GRID = 1024, ITERS = 8;
const size_t COLD_WORDS = 1ull << 26;
template <int K, int W, int LOADS> void run() {
    const size_t n = (size_t)GRID * 256 * ITERS * 4;
    CK(hipMemset(dout, 0, n * 4));
    k<K, W, LOADS><<<GRID, 256>>>(dout, cold, (unsigned)(n * 4), ITERS, (unsigned)COLD_WORDS);
    CK(hipMemcpy(hbuf, dout, n * 4, hipMemcpyDeviceToHost));
    const unsigned expect[4] = {0x11110000u, 0x22220000u, 0x33330000u, 0x44440000u};
    size_t bad = 0; unsigned long long lanes = 0;
    for (size_t i = 0; i < n; ++i)
        if (hbuf[i] != expect[i & 3]) { ++bad; lanes |= 1ull << ((i / 4) & 63); }
    printf("  dword %d overwritten after %d wait states, %d loads queued before the store: %8zu wrong dwords, lanes mask %016llx\n", W, K, LOADS, bad, lanes);
}
int main() {
    const size_t n = (size_t)GRID * 256 * ITERS * 4;
    CK(hipMalloc(&dout, n * 4)); CK(hipMalloc(&cold, COLD_WORDS * 4)); CK(hipMemset(cold, 0, COLD_WORDS * 4));
    hbuf = (unsigned*)malloc(n * 4);
    run<0, 0, 0>(); run<0, 1, 0>(); run<0, 2, 0>(); run<0, 3, 0>();
    run<1, 0, 0>(); run<1, 3, 0>(); run<2, 0, 0>(); run<2, 3, 0>(); run<3, 3, 0>(); run<4, 3, 0>(); run<8, 3, 0>();
    run<0, 0, 1>(); run<0, 3, 1>(); run<0, 3, 2>(); run<0, 3, 4>(); run<0, 0, 8>(); run<1, 3, 1>();
    return 0;
}
