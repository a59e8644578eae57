// Development aid (round 6): how fast the hardware starts waves.  k_lbs_prep runs 4 000 short waves (one per frame) and takes
// 7.6 us + 3.4 us per 1000 frames whatever its waves do; this times EMPTY kernels of the same shapes.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/ubench_dispatch tools/ubench_dispatch.hip && tools/bin/ubench_dispatch
#include <hip/hip_runtime.h>
#include <cstdio>
template <int LDS, int SPIN>
__global__ void k(float* out) {
    __shared__ float s[LDS > 0 ? LDS : 1];
    if (LDS > 0) s[threadIdx.x % LDS] = 1.0f;
    float a = threadIdx.x;
    for (int i = 0; i < SPIN; ++i) a = a * 1.0001f + 0.5f;
    if (a == 123.456f) out[0] = a + (LDS > 0 ? s[0] : 0.0f);
}
template <int LDS, int SPIN>
static void run(const char* name, int grid, int block, float* d) {
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<LDS, SPIN>), dim3(grid), dim3(block), 0, 0, d);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k<LDS, SPIN>), dim3(grid), dim3(block), 0, 0, d);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s grid %6d x %4d threads = %6d waves: %7.2f us per launch (%.2f ns per wave)\n", name, grid, block, grid * block / 64, ms * 1e3 / 20, ms * 1e6 / 20 / (grid * block / 64));
}
int main() {
    float* d; hipMalloc(&d, 4);
    run<0, 0>("empty", 1, 64, d);
    run<0, 0>("empty", 250, 1024, d);
    run<0, 0>("empty", 1000, 256, d);
    run<0, 0>("empty", 4000, 64, d);
    run<0, 0>("empty", 4000, 256, d);
    run<0, 0>("empty", 16000, 64, d);
    run<1024, 0>("4 KB LDS", 1000, 256, d);
    run<0, 2000>("2000 dependent FMAs (~8k cycles)", 1000, 256, d);
    run<0, 2000>("2000 dependent FMAs (~8k cycles)", 250, 1024, d);
    run<0, 2000>("2000 dependent FMAs (~8k cycles)", 4000, 256, d);
    return 0;
}
