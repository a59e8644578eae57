// Which load flavours see another CU's sc0 / sc1 store through the XCD's L2, and what a poll costs (round 4).
// Block W stores, block R polls: the reader first caches the word with a plain load, then tells the writer to go (agent scope),
// the writer stores the new value with the flavour under test, the reader polls (bounded) with the flavour under test.
// hipcc --offload-arch=gfx950 -O2 tools/ubench_scope.hip -o tools/bin/ubench_scope && tools/bin/ubench_scope
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
__device__ __forceinline__ unsigned ld(unsigned* p, int fl) {
    unsigned v;
    switch (fl) {
        case 0: asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); break;
        case 1: asm volatile("global_load_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); break;
        case 2: asm volatile("global_load_dword %0, %1, off sc0 nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); break;
        case 3: asm volatile("buffer_inv sc0\n\tglobal_load_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); break;
        case 4: asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); break;
        case 5: asm volatile("global_load_dword %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); break;
        default: asm volatile("buffer_inv sc1\n\tglobal_load_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); break;
    }
    return v;
}
__device__ __forceinline__ void st(unsigned* p, unsigned v, int fl) {
    switch (fl) {
        case 0: asm volatile("global_store_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" :: "v"(p), "v"(v) : "memory"); break;
        case 1: asm volatile("global_store_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" :: "v"(p), "v"(v) : "memory"); break;
        default: asm volatile("global_store_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" :: "v"(p), "v"(v) : "memory"); break;
    }
}
__global__ void k(unsigned* word, unsigned* go, long long* out, int W, int R, int lfl, int sfl, int round) {
    if (threadIdx.x != 0) return;
    unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if ((int)blockIdx.x == R) {
        unsigned v0; asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v0) : "v"(word) : "memory");   // cache it in this CU's L1
        __hip_atomic_store(go, (unsigned)round, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        long long t0 = wall_clock64(), t1 = t0; int polls = 0; bool seen = false;
        while (wall_clock64() - t0 < 2000000) {   // 20 ms at 100 MHz
            ++polls;
            if (ld(word, lfl) == (unsigned)round) { seen = true; t1 = wall_clock64(); break; }
        }
        // cost of a poll that hits the settled value
        long long t2 = wall_clock64(); unsigned acc = 0;
        for (int i = 0; i < 64; ++i) acc += ld(word, lfl);
        long long t3 = wall_clock64();
        out[0] = seen; out[1] = polls; out[2] = t1 - t0; out[3] = (t3 - t2); out[4] = xcc; out[6] = acc;
    } else if ((int)blockIdx.x == W) {
        while (__hip_atomic_load(go, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != (unsigned)round) __builtin_amdgcn_s_sleep(8);
        for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(8);   // let the reader spin on its cached copy first
        st(word, (unsigned)round, sfl);
        out[5] = xcc;
    }
}
int main() {
    unsigned *word, *go; long long* out;
    hipMalloc(&word, 256); hipMalloc(&go, 256); hipMalloc(&out, 64);
    hipMemset(word, 0, 256); hipMemset(go, 0, 256);
    const char* ln[] = {"plain", "sc0", "sc0 nt", "inv sc0 + sc0", "sc1", "nt", "inv sc1 + sc0"};
    const char* sn[] = {"plain", "sc0", "sc1"};
    int round = 0;
    for (int pair = 0; pair < 2; ++pair) {
        const int W = 0, R = pair == 0 ? 8 : 3;   // blocks 0 and 8: the same XCD (blocks go round-robin over the 8 XCDs); 0 and 3: two XCDs
        for (int sfl = 0; sfl < 3; ++sfl)
            for (int lfl = 0; lfl < 7; ++lfl) {
                ++round;
                hipMemset(out, 0, 64);
                hipLaunchKernelGGL(k, dim3(16), dim3(64), 0, 0, word, go, out, W, R, lfl, sfl, round);
                hipDeviceSynchronize();
                long long h[8]; hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
                printf("blocks %d->%d (xcc %lld->%lld) store %-5s load %-14s: %s after %5lld polls, %7.2f us; settled poll %.2f us\n", W, R, h[5], h[4], sn[sfl], ln[lfl],
                       h[0] ? "seen" : "NOT SEEN", h[1], h[2] / 100.0, h[3] / 100.0 / 64);
            }
    }
    return 0;
}
