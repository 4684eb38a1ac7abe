// vmcnt_order.hip — do a buffer_load (MUBUF) and a younger global_load retire in issue order on gfx950, i.e. is
// `s_waitcnt vmcnt(1)` after {buffer_load cold line; global_load hot line} enough to read the buffer_load's data?
// Same question for {global_load cold; buffer_load hot}, {buffer_load cold; buffer_store} and {global cold; global hot}.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k(const unsigned* cold, const unsigned* hot, unsigned* sink, int* bad_out, int iters, unsigned cold_words) {
    const int lane = threadIdx.x & 63;
    const unsigned gid = blockIdx.x * 256 + threadIdx.x;
    __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void*)cold, 0, (int)(cold_words * 4u), 0x00020000);
    __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc((void*)hot, 0, 4096 * 4, 0x00020000);
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)sink, 0, 1 << 24, 0x00020000);
    __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc((void*)hot, 0, 0, 0x00020000);
    int bad = 0;
    unsigned seed = gid * 2654435761u + 12345u;
    for (int it = 0; it < iters; ++it) {
        seed = seed * 1664525u + 1013904223u;
        const unsigned ci = (seed >> 4) % cold_words;          // random cold word (one line per lane: worst latency spread)
        const unsigned hi = (lane * 4 + it) & 4095;
        unsigned got, other;
        const unsigned coff = ci * 4u, hoff = hi * 4u, soff = (gid * 4u) & ((1u << 24) - 1);
        const unsigned long long cptr = (unsigned long long)(cold + ci), hptr = (unsigned long long)(hot + hi);
        if (MODE == 0)        // buffer cold, global hot
            asm volatile("v_mov_b32 %0, 0xdeadbeef\n s_nop 1\n buffer_load_dword %0, %2, %3, 0 offen\n global_load_dword %1, %4, off\n"
                         "s_waitcnt vmcnt(1)\n v_mov_b32 %0, %0\n s_waitcnt vmcnt(0)\n"
                         : "=&v"(got), "=&v"(other) : "v"(coff), "s"(rc), "v"(hptr) : "memory");
        else if (MODE == 1)   // global cold, buffer hot
            asm volatile("v_mov_b32 %0, 0xdeadbeef\n s_nop 1\n global_load_dword %0, %2, off\n buffer_load_dword %1, %3, %4, 0 offen\n"
                         "s_waitcnt vmcnt(1)\n v_mov_b32 %0, %0\n s_waitcnt vmcnt(0)\n"
                         : "=&v"(got), "=&v"(other) : "v"(cptr), "v"(hoff), "s"(rh) : "memory");
        else if (MODE == 2)   // buffer cold, buffer store
            asm volatile("v_mov_b32 %0, 0xdeadbeef\n v_mov_b32 %1, 0\n s_nop 1\n buffer_load_dword %0, %2, %3, 0 offen\n buffer_store_dword %1, %4, %5, 0 offen\n"
                         "s_waitcnt vmcnt(1)\n v_mov_b32 %0, %0\n s_waitcnt vmcnt(0)\n"
                         : "=&v"(got), "=&v"(other) : "v"(coff), "s"(rc), "v"(soff), "s"(rs) : "memory");
        else if (MODE == 3)   // global cold, global hot
            asm volatile("v_mov_b32 %0, 0xdeadbeef\n s_nop 1\n global_load_dword %0, %2, off\n global_load_dword %1, %3, off\n"
                         "s_waitcnt vmcnt(1)\n v_mov_b32 %0, %0\n s_waitcnt vmcnt(0)\n"
                         : "=&v"(got), "=&v"(other) : "v"(cptr), "v"(hptr) : "memory");
        else if (MODE >= 5) { // buffer cold, then an OUT-OF-RANGE buffer op: 5 load all lanes, 6 load odd lanes, 7 store all lanes, 8 load via descriptor with 0 records
            const unsigned oob = 0x80000000u;
            const unsigned yoff = MODE == 6 ? ((lane & 1) ? oob : hoff) : oob;
            if (MODE == 7)
                asm volatile("v_mov_b32 %0, 0xdeadbeef\n v_mov_b32 %1, 0\n s_nop 1\n buffer_load_dword %0, %2, %3, 0 offen\n buffer_store_dword %1, %4, %5, 0 offen\n"
                             "s_waitcnt vmcnt(1)\n v_mov_b32 %0, %0\n s_waitcnt vmcnt(0)\n"
                             : "=&v"(got), "=&v"(other) : "v"(coff), "s"(rc), "v"(yoff), "s"(rs) : "memory");
            else if (MODE == 8)
                asm volatile("v_mov_b32 %0, 0xdeadbeef\n s_nop 1\n buffer_load_dword %0, %2, %3, 0 offen\n buffer_load_dword %1, %4, %5, 0 offen\n"
                             "s_waitcnt vmcnt(1)\n v_mov_b32 %0, %0\n s_waitcnt vmcnt(0)\n"
                             : "=&v"(got), "=&v"(other) : "v"(coff), "s"(rc), "v"(hoff), "s"(rz) : "memory");
            else
                asm volatile("v_mov_b32 %0, 0xdeadbeef\n s_nop 1\n buffer_load_dword %0, %2, %3, 0 offen\n buffer_load_dword %1, %4, %5, 0 offen\n"
                             "s_waitcnt vmcnt(1)\n v_mov_b32 %0, %0\n s_waitcnt vmcnt(0)\n"
                             : "=&v"(got), "=&v"(other) : "v"(coff), "s"(rc), "v"(yoff), "s"(rh) : "memory");
            if (MODE == 8 && other != 0) bad += 1 << 20;      // an empty descriptor must read zeros
        } else                // buffer cold, buffer hot
            asm volatile("v_mov_b32 %0, 0xdeadbeef\n s_nop 1\n buffer_load_dword %0, %2, %3, 0 offen\n buffer_load_dword %1, %4, %5, 0 offen\n"
                         "s_waitcnt vmcnt(1)\n v_mov_b32 %0, %0\n s_waitcnt vmcnt(0)\n"
                         : "=&v"(got), "=&v"(other) : "v"(coff), "s"(rc), "v"(hoff), "s"(rh) : "memory");
        bad += got != ci * 2654435761u;
    }
    if (bad) atomicAdd(bad_out, bad);
}
__global__ void fill(unsigned* p, size_t n) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) p[i] = (unsigned)i * 2654435761u;
}
int main() {
    const size_t cold_words = 1ull << 28;       // 1 GiB: misses every cache
    unsigned *cold, *hot, *sink; int* dbad;
    CK(hipMalloc(&cold, cold_words * 4)); CK(hipMalloc(&hot, 4096 * 4)); CK(hipMalloc(&sink, 1 << 24)); CK(hipMalloc(&dbad, 4));
    fill<<<4096, 256>>>(cold, cold_words); fill<<<16, 256>>>(hot, 4096);
    CK(hipDeviceSynchronize());
    const char* names[] = {"buffer_load cold ; global_load hot", "global_load cold ; buffer_load hot", "buffer_load cold ; buffer_store",
                           "global_load cold ; global_load hot", "buffer_load cold ; buffer_load hot",
                           "buffer_load cold ; buffer_load OUT OF RANGE", "buffer_load cold ; buffer_load half out of range",
                           "buffer_load cold ; buffer_store OUT OF RANGE", "buffer_load cold ; load via 0-record desc"};
    for (int mode = 0; mode < 9; ++mode) {
        CK(hipMemset(dbad, 0, 4));
        if (mode == 0) k<0><<<2048, 256>>>(cold, hot, sink, dbad, 200, (unsigned)cold_words);
        if (mode == 1) k<1><<<2048, 256>>>(cold, hot, sink, dbad, 200, (unsigned)cold_words);
        if (mode == 2) k<2><<<2048, 256>>>(cold, hot, sink, dbad, 200, (unsigned)cold_words);
        if (mode == 3) k<3><<<2048, 256>>>(cold, hot, sink, dbad, 200, (unsigned)cold_words);
        if (mode == 4) k<4><<<2048, 256>>>(cold, hot, sink, dbad, 200, (unsigned)cold_words);
        if (mode == 5) k<5><<<2048, 256>>>(cold, hot, sink, dbad, 200, (unsigned)cold_words);
        if (mode == 6) k<6><<<2048, 256>>>(cold, hot, sink, dbad, 200, (unsigned)cold_words);
        if (mode == 7) k<7><<<2048, 256>>>(cold, hot, sink, dbad, 200, (unsigned)cold_words);
        if (mode == 8) k<8><<<2048, 256>>>(cold, hot, sink, dbad, 200, (unsigned)cold_words);
        int h; CK(hipMemcpy(&h, dbad, 4, hipMemcpyDeviceToHost));
        printf("%-40s then vmcnt(1): %d stale reads of the older load out of %d\n", names[mode], h, 2048 * 256 * 200);
    }
    return 0;
}
