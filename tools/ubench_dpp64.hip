// Development aid (round 5): what a v_fmac_f64_dpp ... row_newbcast costs beside a plain v_fma_f64 and beside the LDS-broadcast form of the
// same update (ds_read_b64 of a word every lane reads + v_fma_f64), one wave per SIMD (the LDL^T panel's situation: chain_solve.hip,
// ldl_panel_eliminate_rows).   hipcc --offload-arch=gfx950 -O3 -o tools/bin/ubench_dpp64 tools/ubench_dpp64.hip && tools/bin/ubench_dpp64
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 512
template <int K> __device__ __forceinline__ void fmac_bc(double& acc, double m, double a) {
    asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(m), "v"(a), "n"(K));
}
__global__ __launch_bounds__(256, 1) void k(int mode, long long* cyc, double* sink) {
    __shared__ double lds[64];
    const int tid = threadIdx.x;
    if (tid < 64) lds[tid] = 1.0 + tid * 1e-9;
    __syncthreads();
    double a[16];
    for (int i = 0; i < 16; ++i) a[i] = tid * 1e-3 + i;
    const double m = 1.0 + tid * 1e-9, y = 1e-30 + tid * 1e-40;
    const long long t0 = clock64();
    if (mode == 0) {
        for (int it = 0; it < ITER; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) a[i] = fma(-m, y, a[i]);
        }
    } else if (mode == 1) {
        for (int it = 0; it < ITER; ++it) {
            fmac_bc<0>(a[0], m, y); fmac_bc<1>(a[1], m, y); fmac_bc<2>(a[2], m, y); fmac_bc<3>(a[3], m, y);
            fmac_bc<4>(a[4], m, y); fmac_bc<5>(a[5], m, y); fmac_bc<6>(a[6], m, y); fmac_bc<7>(a[7], m, y);
            fmac_bc<8>(a[8], m, y); fmac_bc<9>(a[9], m, y); fmac_bc<10>(a[10], m, y); fmac_bc<11>(a[11], m, y);
            fmac_bc<12>(a[12], m, y); fmac_bc<13>(a[13], m, y); fmac_bc<14>(a[14], m, y); fmac_bc<15>(a[15], m, y);
        }
    } else if (mode == 2) {   // 16 broadcast reads (every lane the same word), then 16 plain fmas with them
        for (int it = 0; it < ITER; ++it) {
            double b[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) b[i] = lds[(i + it) & 63];
#pragma unroll
            for (int i = 0; i < 16; ++i) a[i] = fma(-b[i], y, a[i]);
        }
    } else {   // v_readlane pair + fma with the scalar
        for (int it = 0; it < ITER; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int lo = __builtin_amdgcn_readlane(__double2loint(m), i), hi = __builtin_amdgcn_readlane(__double2hiint(m), i);
                a[i] = fma(-__hiloint2double(hi, lo), y, a[i]);
            }
        }
    }
    const long long t1 = clock64();
    double r = 0.0;
    for (int i = 0; i < 16; ++i) r += a[i];
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
    if (r == 123.456) sink[0] = r;
}
int main() {
    long long* cyc; double* sink;
    hipMalloc(&cyc, 8 * 8); hipMalloc(&sink, 8);
    const char* names[] = {"v_fma_f64, 16 independent accumulators", "v_fmac_f64_dpp row_newbcast, 16 independent accumulators",
                           "16 x ds_read_b64 (broadcast) + 16 x v_fma_f64", "16 x (2 v_readlane + v_fma_f64 with the scalar)"};
    for (int mode = 0; mode < 4; ++mode) {
        k<<<1, 256>>>(mode, cyc, sink); hipDeviceSynchronize();
        k<<<1, 256>>>(mode, cyc, sink); hipDeviceSynchronize();
        long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("%-58s %7.2f cycles per update (one wave per SIMD, %d x 16 updates)\n", names[mode], (double)c / (ITER * 16), ITER);
    }
    return 0;
}
