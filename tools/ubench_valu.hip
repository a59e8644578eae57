// Development aid (round 5): f32 vector-pipe issue rates on gfx950, alone and beside a matrix-pipe wave, and the shader clock under load.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/ubench_valu tools/ubench_valu.hip && tools/bin/ubench_valu
// Every workgroup = 256 threads (one wave per SIMD); `wpe` workgroups per CU are made resident by the grid size (256 CUs x wpe).
// Modes per wave: 0 = 16 independent v_fma_f32 chains, 1 = 16 independent v_pk_fma_f32 chains, 2 = v_mfma_f32_16x16x32_f16 on
// 8 independent accumulators, 3 = ds_read_b128 stream.  A launch gives workgroup parity p the mode modes[p].
// Round 6 (the two cases MI355X_MICROARCH.md "Two waves per SIMD" names and round 5 did not measure):
//   modes 10 + n (n = 0..8): ONE wave whose stream is  [v_mfma_f32_16x16x32_f16 ; n x v_pk_fma_f32]  repeated (8 independent accumulators,
//                  16 independent packed chains) -- "MFMA cover": do the wave's own vector instructions ride under the matrix passes?
//   modes 30 + n: the same with n x v_fma_f32
//   prio: the role-1 wave runs at s_setprio `prio1` (0..3) -- a vector wave beside a matrix wave, with priority.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
#define ITER 2048

template <int N, bool PK>
__device__ __forceinline__ float interleaved(int tid) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
    half8 av, bv;
    for (int e = 0; e < 8; ++e) { av[e] = (_Float16)(tid * 1e-3f); bv[e] = (_Float16)1.0f; }
    f32x2 a[16];
    for (int i = 0; i < 16; ++i) a[i] = f32x2{tid * 1e-3f + i, 1.0f};
    const f32x2 y = {1.0f + tid * 1e-9f, 1.0f}, z = {1e-30f, 1e-30f};
    int c = 0;
    for (int it = 0; it < ITER / 4; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(av), "v"(bv));
#pragma unroll
                for (int q = 0; q < N; ++q) {
                    const int j = (u * 8 * N + i * N + q) & 15;
                    if (PK) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(y), "v"(z));
                    else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[j].x) : "v"(y.x), "v"(z.x));
                }
            }
    }
    float r = 0.0f;
    for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][3];
    for (int i = 0; i < 16; ++i) r += a[i].x + a[i].y;
    return r;
}

// [v_mfma_f32_16x16x4_f32 ; N x v_fma_f32] (the blend of the round-6 export kernel: T = W x A on the f32 matrix instruction, K = 4 joints)
template <int N>
__device__ __forceinline__ float interleaved_f32(int tid) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
    float av = tid * 1e-3f, bv = 1.0f;
    float a[16];
    for (int i = 0; i < 16; ++i) a[i] = tid * 1e-3f + i;
    const float y = 1.0f + tid * 1e-9f, z = 1e-30f;
    for (int it = 0; it < ITER / 4; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(av), "v"(bv));
#pragma unroll
                for (int q = 0; q < N; ++q) {
                    const int j = (u * 8 * N + i * N + q) & 15;
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(y), "v"(z));
                }
            }
    }
    float r = 0.0f;
    for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][3];
    for (int i = 0; i < 16; ++i) r += a[i];
    return r;
}

// [v_mfma_f32_16x16x32_f16 ; N x ds_read_b128 ; M x v_fma_f32]: LDS reads (and vector instructions) inside the wave's matrix stream
template <int N, int M>
__device__ __forceinline__ float interleaved_lds(int tid, const float* buf) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
    half8 av, bv;
    for (int e = 0; e < 8; ++e) { av[e] = (_Float16)(tid * 1e-3f); bv[e] = (_Float16)1.0f; }
    f32x4 d[4];
    for (int i = 0; i < 4; ++i) d[i] = f32x4{0, 0, 0, 0};
    float a[16];
    for (int i = 0; i < 16; ++i) a[i] = tid * 1e-3f + i;
    const float y = 1.0f + tid * 1e-9f, z = 1e-30f;
    const unsigned addr = (unsigned)(size_t)buf + (tid & 63) * 16;
    for (int it = 0; it < ITER / 4; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(av), "v"(bv));
#pragma unroll
                for (int q = 0; q < N; ++q) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d[(i * N + q) & 3]) : "v"(addr), "n"(1024 * ((0 * N + q) & 7)));
#pragma unroll
                for (int q = 0; q < M; ++q) {
                    const int j = (u * 8 * M + i * M + q) & 15;
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(y), "v"(z));
                }
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    float r = 0.0f;
    for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][3];
    for (int i = 0; i < 4; ++i) r += d[i][0] + d[i][2];
    for (int i = 0; i < 16; ++i) r += a[i];
    return r;
}

__global__ __launch_bounds__(256, 2) void k(int mode0, int mode1, int prio1, int* cu_seen, long long* cyc, long long* wall, float* sink) {
    __shared__ int role;
    __shared__ __attribute__((aligned(16))) float buf[4096];
    const int tid = threadIdx.x;
    if (tid == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg(4 | (8 << 6) | (7 << 11));
        const unsigned xc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));
        role = atomicAdd(&cu_seen[((xc & 15) << 8) | (hw & 255)], 1) & 1;
    }
    for (int i = tid; i < 4096; i += 256) buf[i] = 1.0f + i * 1e-6f;
    __syncthreads();
    const int mode = role ? mode1 : mode0;
    if (role && prio1 == 1) __builtin_amdgcn_s_setprio(1);
    if (role && prio1 == 2) __builtin_amdgcn_s_setprio(2);
    if (role && prio1 == 3) __builtin_amdgcn_s_setprio(3);
    const long long w0 = wall_clock64();
    const long long t0 = clock64();
    float r = 0.0f;
    if (mode == 0) {
        float a[16];
        for (int i = 0; i < 16; ++i) a[i] = tid * 1e-3f + i;
        const float y = 1.0f + tid * 1e-9f;
        for (int it = 0; it < ITER; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) a[i] = fmaf(a[i], y, 1e-30f);
        }
        for (int i = 0; i < 16; ++i) r += a[i];
    } else if (mode == 1) {
        f32x2 a[16];
        for (int i = 0; i < 16; ++i) a[i] = f32x2{tid * 1e-3f + i, 1.0f};
        const f32x2 y = {1.0f + tid * 1e-9f, 1.0f}, z = {1e-30f, 1e-30f};
        for (int it = 0; it < ITER; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) a[i] = __builtin_elementwise_fma(a[i], y, z);
        }
        for (int i = 0; i < 16; ++i) r += a[i].x + a[i].y;
    } else if (mode == 2) {
        f32x4 acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
        half8 av, bv;
        for (int e = 0; e < 8; ++e) { av[e] = (_Float16)(tid * 1e-3f); bv[e] = (_Float16)1.0f; }
        for (int it = 0; it < ITER / 4; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][3];
    } else if (mode == 4) {
        r = interleaved_f32<0>(tid);
    } else if (mode >= 50 && mode < 60) {
        switch (mode) {
            case 50: r = interleaved_f32<0>(tid); break; case 52: r = interleaved_f32<2>(tid); break; case 54: r = interleaved_f32<4>(tid); break;
            case 56: r = interleaved_f32<6>(tid); break; case 58: r = interleaved_f32<8>(tid); break; case 59: r = interleaved_f32<10>(tid); break;
        }
    } else if (mode >= 60 && mode < 70) {
        switch (mode) {
            case 61: r = interleaved_lds<1, 0>(tid, buf); break; case 62: r = interleaved_lds<2, 0>(tid, buf); break;
            case 63: r = interleaved_lds<1, 1>(tid, buf); break; case 64: r = interleaved_lds<1, 2>(tid, buf); break; case 65: r = interleaved_lds<2, 2>(tid, buf); break;
        }
    } else if (mode >= 10 && mode < 50) {
        switch (mode) {
#define CASE(n) case 10 + n: r = interleaved<n, true>(tid); break; case 30 + n: r = interleaved<n, false>(tid); break;
            CASE(0) CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(8)
#undef CASE
        }
    } else if (mode == 3) {
        f32x4 s4 = {0, 0, 0, 0};
        const int off = (tid & 63) * 4;
        for (int it = 0; it < ITER; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) s4 += *reinterpret_cast<const f32x4*>(&buf[(off + i * 256 + it) & 4092]);
        }
        r = s4.x + s4.y + s4.z + s4.w;
    }
    const long long t1 = clock64();
    const long long w1 = wall_clock64();
    if (tid == 0) { cyc[blockIdx.x] = ((t1 - t0) << 1) | role; wall[blockIdx.x] = w1 - w0; }
    if (r == 123.456f) sink[0] = r;
}

int main() {
    int* seen; long long *cyc, *wall; float* sink;
    hipMalloc(&seen, 4096 * 4); hipMalloc(&cyc, 1024 * 8); hipMalloc(&wall, 1024 * 8); hipMalloc(&sink, 4);
    auto name = [](int m) -> std::string {
        const char* names[] = {"v_fma_f32 x16 chains", "v_pk_fma_f32 x16 chains", "mfma 16x16x32 f16 x8 acc", "ds_read_b128"};
        if (m < 4) return names[m];
        if (m == 4) return "mfma 16x16x4 f32 x8 acc";
        if (m >= 50 && m < 60) { char b[64]; snprintf(b, 64, "[mfma_f32x4 ; %d x v_fma_f32]", m == 59 ? 10 : m - 50); return b; }
        if (m >= 60 && m < 70) { const int n[] = {0, 1, 2, 1, 1, 2}, v[] = {0, 0, 0, 1, 2, 2}; char b[64]; snprintf(b, 64, "[mfma ; %d ds_read_b128 ; %d v_fma]", n[m - 60], v[m - 60]); return b; }
        char b[64]; snprintf(b, 64, "[mfma ; %d x %s]", m >= 30 ? m - 30 : m - 10, m >= 30 ? "v_fma_f32" : "v_pk_fma_f32"); return b; };
    auto ninst_of = [](int m) { return m == 2 || m == 4 || m >= 10 ? ITER / 4 * 64 : ITER * 16; };     // interleaved modes: per MFMA (= per group)
    struct Cfg { int wpe, m0, m1, prio1; } cfgs[] = {{1, 0, 0, 0}, {2, 0, 0, 0}, {1, 1, 1, 0}, {2, 1, 1, 0}, {1, 2, 2, 0}, {2, 2, 2, 0}, {2, 2, 0, 0}, {2, 2, 1, 0}, {1, 3, 3, 0}, {2, 3, 3, 0}, {2, 2, 3, 0}, {2, 3, 1, 0},
        // round 6: priority for the vector wave beside the matrix wave
        {2, 2, 0, 1}, {2, 2, 0, 2}, {2, 2, 0, 3}, {2, 2, 1, 1}, {2, 2, 1, 2}, {2, 2, 1, 3},
        // round 6: one wave, its own vector instructions between its matrix instructions
        {1, 10, 10, 0}, {1, 11, 11, 0}, {1, 12, 12, 0}, {1, 13, 13, 0}, {1, 14, 14, 0}, {1, 15, 15, 0}, {1, 16, 16, 0}, {1, 18, 18, 0},
        {1, 31, 31, 0}, {1, 32, 32, 0}, {1, 33, 33, 0}, {1, 34, 34, 0}, {1, 35, 35, 0}, {1, 36, 36, 0}, {1, 38, 38, 0},
        // ... and two such waves per SIMD
        {2, 13, 13, 0}, {2, 33, 33, 0}, {2, 12, 12, 0},
        // round 6: the f32 matrix instruction (blend T = W x A, K = 4 joints), alone / two waves / with vector instructions inside / beside an f16 matrix wave
        {1, 4, 4, 0}, {2, 4, 4, 0}, {1, 52, 52, 0}, {1, 54, 54, 0}, {1, 56, 56, 0}, {1, 58, 58, 0}, {1, 59, 59, 0}, {2, 2, 4, 0}, {2, 2, 54, 0}, {2, 2, 56, 0}, {2, 32, 56, 0},
        // round 6: LDS reads (and vector instructions) inside the wave's own f16 matrix stream
        {1, 61, 61, 0}, {1, 62, 62, 0}, {1, 63, 63, 0}, {1, 64, 64, 0}, {1, 65, 65, 0}, {2, 64, 64, 0}};
    for (auto c : cfgs) {
        const int nb = 256 * c.wpe;
        hipMemset(seen, 0, 4096 * 4);
        k<<<nb, 256>>>(c.m0, c.m1, c.prio1, seen, cyc, wall, sink);
        hipDeviceSynchronize();
        hipMemset(seen, 0, 4096 * 4);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        k<<<nb, 256>>>(c.m0, c.m1, c.prio1, seen, cyc, wall, sink);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<long long> hc(nb), hw(nb);
        hipMemcpy(hc.data(), cyc, nb * 8, hipMemcpyDeviceToHost); hipMemcpy(hw.data(), wall, nb * 8, hipMemcpyDeviceToHost);
        double s[2] = {0, 0}, ws[2] = {0, 0}; int n[2] = {0, 0};
        for (int b = 0; b < nb; ++b) { const int role = hc[b] & 1; s[role] += (double)(hc[b] >> 1); ws[role] += (double)hw[b]; n[role]++; }
        printf("wg/CU %d | role0: %-26s", c.wpe, name(c.m0).c_str());
        if (n[0]) printf(" %6.2f cyc/inst (n=%d), clock %.2f GHz", s[0] / n[0] / ninst_of(c.m0), n[0], (s[0] / n[0]) / (ws[0] / n[0]) * 0.1);
        if (c.wpe == 2) { printf(" | role1 (prio %d): %-26s", c.prio1, name(c.m1).c_str()); if (n[1]) printf(" %6.2f cyc/inst (n=%d), clock %.2f GHz", s[1] / n[1] / ninst_of(c.m1), n[1], (s[1] / n[1]) / (ws[1] / n[1]) * 0.1); }
        printf(" | kernel %.1f us\n", ms * 1e3);
    }
    return 0;
}
