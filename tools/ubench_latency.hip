// Development aid: instruction / synchronisation latencies of one 256-thread workgroup on gfx950, measured with s_memtime
// around unrolled dependent chains (one wave per SIMD, as in k_chain_solve).  Build + run:
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/ubench_latency tools/ubench_latency.hip && tools/bin/ubench_latency
#include <hip/hip_runtime.h>
#include <cstdio>

#define N 512
__device__ __forceinline__ double rl(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ double dpp(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

__global__ __launch_bounds__(256) void k(long long* out, double* sink, double seed) {
    extern __shared__ double lds[];
    const int tid = threadIdx.x;
    lds[tid] = seed + tid; lds[256 + tid] = 1.0;
    __syncthreads();
    double x = seed + 1e-9 * tid, y = 1.0 + 1e-12 * tid;
    long long t0, t1;
    int s = 0;
#define REC() do { if (tid == 0) out[s] = t1 - t0; ++s; } while (0)
    // the timer reads are tied into the dependency chain: x "depends" on t0, and t1 is read after x exists
#define T0() do { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)); asm volatile("" : "+v"(x), "+v"(y) : "s"(t0)); } while (0)
#define T1() do { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) : "v"(x)); } while (0)
    // 0: dependent v_fma_f64 chain
    T0();
#pragma unroll
    for (int i = 0; i < N; ++i) x = fma(x, y, 1e-30);
    T1(); REC();
    // 1: 4 independent fma chains (issue rate)
    T0();
    double a = x, b = x + 1, c = x + 2, d = x + 3;
#pragma unroll
    for (int i = 0; i < N / 4; ++i) { a = fma(a, y, 1e-30); b = fma(b, y, 1e-30); c = fma(c, y, 1e-30); d = fma(d, y, 1e-30); }
    x = a + b + c + d;
    T1(); REC();
    // 2: readlane -> mul -> fma chain (back-substitution step)
    T0();
#pragma unroll
    for (int i = 0; i < N; ++i) { const double yj = rl(x, i & 63); const double dj = yj * y; x = fma(-1e-30, dj, x); }
    T1(); REC();
    // 3: as 2 with the exec-masked select
    T0();
#pragma unroll
    for (int i = 0; i < N; ++i) { const double yj = rl(x, i & 63); const double dj = yj * y; x = (tid == (i & 63)) ? dj : fma(-1e-30, dj, x); }
    T1(); REC();
    // 4: dependent LDS read chain (pointer chase)
    lds[512 + tid] = (double)((tid + 1) & 255);
    __syncthreads();
    T0();
    int p = tid + (int)(x * 0.0);
#pragma unroll 16
    for (int i = 0; i < N; ++i) p = (int)lds[512 + p];
    x += p;
    T1(); REC();
    // 5: LDS write -> barrier -> read round trip
    T0();
#pragma unroll 8
    for (int i = 0; i < N; ++i) { lds[tid] = x; __syncthreads(); x = lds[(tid + 64) & 255] * 0.5; }
    T1(); REC();
    // 6: bare barrier
    T0();
#pragma unroll 8
    for (int i = 0; i < N; ++i) { __syncthreads(); }
    T1(); REC();
    // 7: reciprocal + 2 Newton steps chain
    T0();
#pragma unroll
    for (int i = 0; i < N; ++i) { double r = __builtin_amdgcn_rcp(x); r = fma(fma(-x, r, 1.0), r, r); r = fma(fma(-x, r, 1.0), r, r); x = r + 1.0; }
    T1(); REC();
    // 8: IEEE divide chain
    T0();
#pragma unroll
    for (int i = 0; i < N; ++i) x = 1.0 / x + 1.0;
    T1(); REC();
    // 9: DPP wave reduction (6 steps) chain
    T0();
#pragma unroll
    for (int i = 0; i < N / 8; ++i) {
        x += dpp<0xB1>(x); x += dpp<0x4E>(x); x += dpp<0x141>(x); x += dpp<0x140>(x);
        x = rl(x, 0) + rl(x, 16) + rl(x, 32) + rl(x, 48);
    }
    T1(); REC();
    // 10: global load dependent chain (L2 hit): pointer chase through sink
    {
        const double* g = sink + 1024;
        T0();
        int q = tid + (int)(x * 0.0);
#pragma unroll 4
        for (int i = 0; i < 64; ++i) q = (int)g[q];
        x += q;
        T1(); REC();
    }
    // 11: sqrt chain
    T0();
#pragma unroll
    for (int i = 0; i < N; ++i) x = sqrt(x + 2.0);
    T1(); REC();
    // 12: readlane only chain (v -> s -> v)
    T0();
#pragma unroll
    for (int i = 0; i < N; ++i) x = rl(x, i & 63) + 1e-30;
    T1(); REC();
    // 13: mfma f64 16x16x4 dependent chain
    {
        typedef double v4d __attribute__((vector_size(32)));
        T0();
        v4d acc = {x, x, x, x};
#pragma unroll
        for (int i = 0; i < N / 4; ++i) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, acc, 0, 0, 0);
        x = acc[0] + acc[1] + acc[2] + acc[3];
        T1(); REC();
    }
    sink[tid] = x;
}

int main() {
    long long* d; double* sink;
    hipMalloc(&d, 64 * sizeof(long long));
    hipMalloc(&sink, 4096 * sizeof(double));
    double h[4096];
    for (int i = 0; i < 4096; ++i) h[i] = (i >= 1024 && i < 1280) ? (double)((i - 1024 + 1) & 255) : 0.0;
    hipMemcpy(sink, h, sizeof(h), hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) k<<<1, 256, 8192>>>(d, sink, 1.5);
    hipDeviceSynchronize();
    long long o[64];
    hipMemcpy(o, d, sizeof(o), hipMemcpyDeviceToHost);
    const char* nm[] = {"dependent v_fma_f64", "4 independent fma chains (per fma)", "readlane->mul->fma step", "same + masked select",
                        "LDS dependent read", "LDS write->barrier->read", "bare s_barrier (4 waves)", "rcp + 2 Newton (+add)", "IEEE 1/x (+add)",
                        "wave sum: 4 dpp + 4 readlane (per reduction)", "global (L2) dependent load", "sqrt f64 (+add)", "readlane->add", "mfma_f64_16x16x4 dependent"};
    const int cnt[] = {N, N, N, N, N, N, N, N, N, N / 8, 64, N, N, N / 4};
    // s_memtime counts at 100 MHz on this part?  print raw ticks per op; the shader clock ratio is printed by tools/prof_chain.py
    for (int i = 0; i < 14; ++i) printf("%-50s %8.2f ticks/op\n", nm[i], (double)o[i] / cnt[i]);
    return 0;
}
