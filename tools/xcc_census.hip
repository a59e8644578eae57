// prints which XCC id each workgroup of a 512-block grid observes (HW_REG_XCC_ID), by blockIdx % 8
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* o) { unsigned x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x)); if (threadIdx.x == 0) o[blockIdx.x] = x; }
int main() {
    unsigned* d; hipMalloc(&d, 512 * 4); k<<<512, 64>>>(d); unsigned h[512]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int tab[8][16] = {};
    for (int b = 0; b < 512; ++b) tab[b % 8][h[b] & 15]++;
    printf("raw first 16:"); for (int b = 0; b < 16; ++b) printf(" %08x", h[b]); printf("\n");
    for (int r = 0; r < 8; ++r) { printf("b%%8=%d:", r); for (int x = 0; x < 16; ++x) printf(" %d", tab[r][x]); printf("\n"); }
    return 0;
}
