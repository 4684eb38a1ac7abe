// store_war2.hip — companions of store_data_war.hip: which other source registers are read late on gfx950?
//  A  buffer_store_dwordx4: the voffset VGPR overwritten by the next instruction
//  B  global_store_dwordx4: a data VGPR overwritten after K wait states (hipcc pads 1 for FLAT stores > 64 bits)
//  C  ds_write_b128: a data VGPR overwritten by the next instruction
//  D  buffer_load_dwordx4: the voffset VGPR overwritten by the next instruction
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE, int K>
__global__ __launch_bounds__(256) void k(unsigned* out, const unsigned* src, unsigned bytes, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned lds[256 * 4];
    const unsigned gid = blockIdx.x * 256 + threadIdx.x;
    __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, (int)bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)bytes, 0x00020000);
    for (int it = 0; it < iters; ++it) {
        const unsigned off = ((unsigned)it * gridDim.x * 256u + gid) * 16u;
        unsigned* gp = out + off / 4;
        const unsigned soff = 0;
        if (MODE == 0)
            asm volatile("v_mov_b32 v100, 0x11110000\n v_mov_b32 v101, 0x22220000\n v_mov_b32 v102, 0x33330000\n v_mov_b32 v103, 0x44440000\n"
                         "v_mov_b32 v104, %0\n s_nop 4\n"
                         "buffer_store_dwordx4 v[100:103], v104, %1, %2 offen\n"
                         ".if %3 > 0\n s_nop %3 - 1\n .endif\n"
                         "v_mov_b32 v104, 0\n s_waitcnt vmcnt(0)\n"
                         :: "v"(off), "s"(ro), "s"(soff), "n"(K) : "v100", "v101", "v102", "v103", "v104", "memory");
        else if (MODE == 1)
            asm volatile("v_mov_b32 v100, 0x11110000\n v_mov_b32 v101, 0x22220000\n v_mov_b32 v102, 0x33330000\n v_mov_b32 v103, 0x44440000\n"
                         "s_nop 4\n"
                         "global_store_dwordx4 %0, v[100:103], off\n"
                         ".if %1 > 0\n s_nop %1 - 1\n .endif\n"
                         "v_mov_b32 v103, 0xdead0000\n s_waitcnt vmcnt(0)\n"
                         :: "v"(gp), "n"(K) : "v100", "v101", "v102", "v103", "memory");
        else if (MODE == 4)      // E global_store_dwordx2
            asm volatile("v_mov_b32 v100, 0x11110000\n v_mov_b32 v101, 0x22220000\n s_nop 4\n"
                         "global_store_dwordx2 %0, v[100:101], off\n"
                         ".if %1 > 0\n s_nop %1 - 1\n .endif\n"
                         "v_mov_b32 v101, 0xdead0000\n v_mov_b32 v100, 0xdead0000\n s_waitcnt vmcnt(0)\n"
                         :: "v"(gp), "n"(K) : "v100", "v101", "memory");
        else if (MODE == 5)      // F global_store_dword
            asm volatile("v_mov_b32 v100, 0x11110000\n s_nop 4\n"
                         "global_store_dword %0, v100, off\n"
                         ".if %1 > 0\n s_nop %1 - 1\n .endif\n"
                         "v_mov_b32 v100, 0xdead0000\n s_waitcnt vmcnt(0)\n"
                         :: "v"(gp), "n"(K) : "v100", "memory");
        else if (MODE == 6)      // G global_store_dwordx3
            asm volatile("v_mov_b32 v100, 0x11110000\n v_mov_b32 v101, 0x22220000\n v_mov_b32 v102, 0x33330000\n s_nop 4\n"
                         "global_store_dwordx3 %0, v[100:102], off\n"
                         ".if %1 > 0\n s_nop %1 - 1\n .endif\n"
                         "v_mov_b32 v102, 0xdead0000\n v_mov_b32 v100, 0xdead0000\n s_waitcnt vmcnt(0)\n"
                         :: "v"(gp), "n"(K) : "v100", "v101", "v102", "memory");
        else if (MODE == 2) {
            const unsigned la = (unsigned)(size_t)lds + threadIdx.x * 16;
            unsigned r0, r1, r2, r3;
            asm volatile("v_mov_b32 v100, 0x11110000\n v_mov_b32 v101, 0x22220000\n v_mov_b32 v102, 0x33330000\n v_mov_b32 v103, 0x44440000\n"
                         "s_nop 4\n"
                         "ds_write_b128 %4, v[100:103]\n"
                         ".if %5 > 0\n s_nop %5 - 1\n .endif\n"
                         "v_mov_b32 v103, 0xdead0000\n v_mov_b32 v100, 0xdead0000\n s_waitcnt lgkmcnt(0)\n"
                         "ds_read_b128 v[104:107], %4\n s_waitcnt lgkmcnt(0)\n"
                         "v_mov_b32 %0, v104\n v_mov_b32 %1, v105\n v_mov_b32 %2, v106\n v_mov_b32 %3, v107\n"
                         : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(la), "n"(K)
                         : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "memory");
            gp[0] = r0; gp[1] = r1; gp[2] = r2; gp[3] = r3;
        } else {
            unsigned r0, r1, r2, r3;
            asm volatile("v_mov_b32 v104, %4\n s_nop 4\n"
                         "buffer_load_dwordx4 v[100:103], v104, %5, %6 offen\n"
                         ".if %7 > 0\n s_nop %7 - 1\n .endif\n"
                         "v_mov_b32 v104, 0\n s_waitcnt vmcnt(0)\n"
                         "v_mov_b32 %0, v100\n v_mov_b32 %1, v101\n v_mov_b32 %2, v102\n v_mov_b32 %3, v103\n"
                         : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(off), "s"(rs), "s"(soff), "n"(K)
                         : "v100", "v101", "v102", "v103", "v104", "memory");
            gp[0] = r0; gp[1] = r1; gp[2] = r2; gp[3] = r3;
        }
    }
}
__global__ void fillsrc(unsigned* p, size_t n) {
    const unsigned e[4] = {0x11110000u, 0x22220000u, 0x33330000u, 0x44440000u};
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) p[i] = e[i & 3];
}
unsigned *dout, *dsrc, *hbuf;
const int GRID = 1024, ITERS = 8;
template <int MODE, int K> void run(const char* what) {
    const int nd = MODE == 4 ? 2 : MODE == 5 ? 1 : MODE == 6 ? 3 : 4;
    const size_t n = (size_t)GRID * 256 * ITERS * 4;
    CK(hipMemset(dout, 0, n * 4));
    k<MODE, K><<<GRID, 256>>>(dout, dsrc, (unsigned)(n * 4), ITERS);
    CK(hipMemcpy(hbuf, dout, n * 4, hipMemcpyDeviceToHost));
    const unsigned expect[4] = {0x11110000u, 0x22220000u, 0x33330000u, 0x44440000u};
    size_t bad = 0; unsigned long long lanes = 0;
    for (size_t i = 0; i < n; ++i)
        if ((int)(i & 3) < nd && hbuf[i] != expect[i & 3]) { ++bad; lanes |= 1ull << ((i / 4) & 63); }
    printf("%-58s %d wait states: %8zu wrong dwords, lanes %016llx\n", what, K, bad, lanes);
}
int main() {
    const size_t n = (size_t)GRID * 256 * ITERS * 4;
    CK(hipMalloc(&dout, n * 4)); CK(hipMalloc(&dsrc, n * 4));
    fillsrc<<<1024, 256>>>(dsrc, n);
    hbuf = (unsigned*)malloc(n * 4);
    run<0, 0>("A buffer_store_dwordx4: voffset VGPR overwritten after"); run<0, 1>("A buffer_store_dwordx4: voffset VGPR overwritten after");
    run<1, 0>("B global_store_dwordx4: data VGPR overwritten after"); run<1, 1>("B global_store_dwordx4: data VGPR overwritten after");
    run<1, 2>("B global_store_dwordx4: data VGPR overwritten after");
    run<2, 0>("C ds_write_b128: data VGPRs overwritten after"); run<2, 1>("C ds_write_b128: data VGPRs overwritten after");
    run<4, 0>("E global_store_dwordx2: data VGPRs overwritten after"); run<4, 1>("E global_store_dwordx2: data VGPRs overwritten after");
    run<5, 0>("F global_store_dword: data VGPR overwritten after");
    run<6, 0>("G global_store_dwordx3: data VGPRs overwritten after"); run<6, 1>("G global_store_dwordx3: data VGPRs overwritten after"); run<6, 2>("G global_store_dwordx3: data VGPRs overwritten after");
    run<3, 0>("D buffer_load_dwordx4: voffset VGPR overwritten after"); run<3, 1>("D buffer_load_dwordx4: voffset VGPR overwritten after");
    return 0;
}
