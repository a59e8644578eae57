// Store-pattern microbenchmark for the LBS export: how fast can 4000 x 6890 x 12 B be written
//   mode 0: linear, 16 B per lane (memset-like upper bound)
//   mode 1: the LBS epilogue pattern: tile = 128 vertices x 64 frames, wave = 32 vertices, half-wave h covers frames
//           fr = (r&3)+8(r>>2)+4h, per (lane, frame) one 12-byte non-temporal store (384-byte runs, row pitch 82 680 B)
//   mode 2: same tiles, but each wave writes whole 1536-byte tile rows (128 vertices of one frame) with 16 B per lane
// hipcc --offload-arch=gfx950 -O3 -o tools/bin/store_pattern tools/store_pattern.hip && tools/bin/store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x3u __attribute__((ext_vector_type(3), aligned(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k_linear(float* out, size_t n4) {
    f32x4 v = {1.f, 2.f, 3.f, 4.f};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
        __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(out) + i);
}

__global__ __launch_bounds__(256, 2) void k_tiles(float* out, int V, int F, int NVT, int mode) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int vt = blockIdx.x % NVT, ft = blockIdx.x / NVT;
    const int v0 = vt * 128, f0 = ft * 64;
    if (mode == 1) {
        const int v = v0 + wv * 32 + (lane & 31), h = lane >> 5;
        for (int nt = 0; nt < 2; ++nt)
            for (int r = 0; r < 16; ++r) {
                const int f = f0 + nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (f < F && v < V) {
                    f32x3u val = {(float)v, (float)f, 1.f};
                    __builtin_nontemporal_store(val, reinterpret_cast<f32x3u*>(out + ((size_t)f * V + v) * 3));
                }
            }
    } else {
        // a tile row = 128 vertices x 12 B = 1536 B = 96 lanes x 16 B: one wave writes rows wv, wv+4, ... (two rows per 3 stores... keep simple: 64 lanes x 16 B + 32 lanes x 16 B)
        for (int fr = wv; fr < 64; fr += 4) {
            const int f = f0 + fr;
            if (f >= F) continue;
            float* row = out + ((size_t)f * V + v0) * 3;
            const int nfl = min(128, V - v0) * 3;   // floats in this row
            for (int c = lane * 4; c < nfl; c += 256) {
                if (c + 4 <= nfl) { f32x3u a = {1.f, 2.f, 3.f}; __builtin_nontemporal_store(a, reinterpret_cast<f32x3u*>(row + c)); row[c + 3] = 4.f; }
                else for (int e = c; e < nfl; ++e) row[e] = 5.f;
            }
        }
    }
}

int main() {
    const int V = 6890, F = 4000;
    const size_t n = (size_t)V * F * 3;
    float* d; hipMalloc(&d, n * 4 + 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int NVT = (V + 127) / 128, NFT = (F + 63) / 64;
    for (int mode = 0; mode < 3; ++mode) {
        float best = 1e9;
        for (int rep = 0; rep < 6; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k_linear, dim3(256 * 8), dim3(256), 0, 0, d, n / 4);
            else hipLaunchKernelGGL(k_tiles, dim3(NVT * NFT), dim3(256), 0, 0, d, V, F, NVT, mode);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep > 0 && ms < best) best = ms;
        }
        printf("mode %d: %.1f us -> %.0f GB/s\n", mode, best * 1e3, n * 4 / (best * 1e-3) / 1e9);
    }
    return 0;
}
