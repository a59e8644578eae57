// Development aid: how long after issue does v_mfma_f32_32x32x16_f16 still READ its 4-register A / B operands on
// gfx950?  Inline assembly (the compiler's hazard recogniser does not look inside): A = B = all ones -> every output is 16.0; N idle
// slots after the MFMA the four A (or B) registers are overwritten with 2.0.  An output != 16 means the matrix pipe read the
// overwritten value, i.e. the hardware does not interlock that write-after-read and software has to keep N wait states.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/mfma_war_hazard tools/mfma_war_hazard.hip && tools/bin/mfma_war_hazard
#include <hip/hip_runtime.h>
#include <cstdio>

template <int N, int WHICH>   // WHICH: 0 = overwrite A, 1 = overwrite B
__global__ void k(float* out) {
    float o0, o1, o2, o3;
    asm volatile(
        "v_mov_b32 v100, 0x3c003c00\n v_mov_b32 v101, 0x3c003c00\n v_mov_b32 v102, 0x3c003c00\n v_mov_b32 v103, 0x3c003c00\n"
        "v_mov_b32 v104, 0x3c003c00\n v_mov_b32 v105, 0x3c003c00\n v_mov_b32 v106, 0x3c003c00\n v_mov_b32 v107, 0x3c003c00\n"
        "v_mov_b32 v108, 0\n v_mov_b32 v109, 0\n v_mov_b32 v110, 0\n v_mov_b32 v111, 0\n v_mov_b32 v112, 0\n v_mov_b32 v113, 0\n"
        "v_mov_b32 v114, 0\n v_mov_b32 v115, 0\n v_mov_b32 v116, 0\n v_mov_b32 v117, 0\n v_mov_b32 v118, 0\n v_mov_b32 v119, 0\n"
        "v_mov_b32 v120, 0\n v_mov_b32 v121, 0\n v_mov_b32 v122, 0\n v_mov_b32 v123, 0\n"
        "s_nop 15\n s_nop 15\n"
        "v_mfma_f32_32x32x16_f16 v[108:123], v[100:103], v[104:107], v[108:123]\n"
        ".if %4 > 0\n s_nop %4 - 1\n .endif\n"
        ".if %5 == 0\n"
        "v_mov_b32 v100, 0x40004000\n v_mov_b32 v101, 0x40004000\n v_mov_b32 v102, 0x40004000\n v_mov_b32 v103, 0x40004000\n"
        ".else\n"
        "v_mov_b32 v104, 0x40004000\n v_mov_b32 v105, 0x40004000\n v_mov_b32 v106, 0x40004000\n v_mov_b32 v107, 0x40004000\n"
        ".endif\n"
        "s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n"
        "v_mov_b32 %0, v108\n v_mov_b32 %1, v113\n v_mov_b32 %2, v118\n v_mov_b32 %3, v123\n"
        : "=v"(o0), "=v"(o1), "=v"(o2), "=v"(o3)
        : "n"(N), "n"(WHICH)
        : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115",
          "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123");
    out[threadIdx.x * 4 + 0] = o0; out[threadIdx.x * 4 + 1] = o1; out[threadIdx.x * 4 + 2] = o2; out[threadIdx.x * 4 + 3] = o3;
}

// Second experiment: the matrix pipe is BUSY with an earlier, independent MFMA of the same wavefront when the MFMA under test is
// issued; its A registers are overwritten N idle slots after its issue.  (An MFMA that queues behind another one may read its
// operands only when it starts executing.)
template <int N>
__global__ void k2(float* out) {
    float o0, o1, o2, o3;
    asm volatile(
        "v_mov_b32 v100, 0x3c003c00\n v_mov_b32 v101, 0x3c003c00\n v_mov_b32 v102, 0x3c003c00\n v_mov_b32 v103, 0x3c003c00\n"
        "v_mov_b32 v104, 0x3c003c00\n v_mov_b32 v105, 0x3c003c00\n v_mov_b32 v106, 0x3c003c00\n v_mov_b32 v107, 0x3c003c00\n"
        "v_mov_b32 v124, 0x3c003c00\n v_mov_b32 v125, 0x3c003c00\n v_mov_b32 v126, 0x3c003c00\n v_mov_b32 v127, 0x3c003c00\n"
        "v_mov_b32 v108, 0\n v_mov_b32 v109, 0\n v_mov_b32 v110, 0\n v_mov_b32 v111, 0\n v_mov_b32 v112, 0\n v_mov_b32 v113, 0\n"
        "v_mov_b32 v114, 0\n v_mov_b32 v115, 0\n v_mov_b32 v116, 0\n v_mov_b32 v117, 0\n v_mov_b32 v118, 0\n v_mov_b32 v119, 0\n"
        "v_mov_b32 v120, 0\n v_mov_b32 v121, 0\n v_mov_b32 v122, 0\n v_mov_b32 v123, 0\n"
        "v_mov_b32 v128, 0\n v_mov_b32 v129, 0\n v_mov_b32 v130, 0\n v_mov_b32 v131, 0\n v_mov_b32 v132, 0\n v_mov_b32 v133, 0\n"
        "v_mov_b32 v134, 0\n v_mov_b32 v135, 0\n v_mov_b32 v136, 0\n v_mov_b32 v137, 0\n v_mov_b32 v138, 0\n v_mov_b32 v139, 0\n"
        "v_mov_b32 v140, 0\n v_mov_b32 v141, 0\n v_mov_b32 v142, 0\n v_mov_b32 v143, 0\n"
        "s_nop 15\n s_nop 15\n"
        "v_mfma_f32_32x32x16_f16 v[108:123], v[100:103], v[104:107], v[108:123]\n"
        "v_mfma_f32_32x32x16_f16 v[128:143], v[124:127], v[104:107], v[128:143]\n"
        ".if %4 > 0\n s_nop %4 - 1\n .endif\n"
        "v_mov_b32 v124, 0x40004000\n v_mov_b32 v125, 0x40004000\n v_mov_b32 v126, 0x40004000\n v_mov_b32 v127, 0x40004000\n"
        "s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n"
        "v_mov_b32 %0, v128\n v_mov_b32 %1, v133\n v_mov_b32 %2, v138\n v_mov_b32 %3, v143\n"
        : "=v"(o0), "=v"(o1), "=v"(o2), "=v"(o3)
        : "n"(N)
        : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115",
          "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131",
          "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143");
    out[threadIdx.x * 4 + 0] = o0; out[threadIdx.x * 4 + 1] = o1; out[threadIdx.x * 4 + 2] = o2; out[threadIdx.x * 4 + 3] = o3;
}

template <int N>
void run2(float* d) {
    k2<N><<<1, 64>>>(d);
    float h[256];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int wrong = 0; float lo = 1e9f, hi = -1e9f;
    for (float v : h) { wrong += v != 16.0f; lo = v < lo ? v : lo; hi = v > hi ? v : hi; }
    printf("pipe busy; overwrite A of the 2nd MFMA %2d idle slots after its issue: %3d of 256 sampled outputs wrong (values %g .. %g)\n", N, wrong, lo, hi);
}

// Third experiment: the A registers are overwritten by an LDS load (ds_read_b128, data returns asynchronously) issued N idle slots
// after the MFMA.
template <int N>
__global__ void k3(float* out) {
    __shared__ unsigned int twos[256];
    twos[threadIdx.x] = 0x40004000u; twos[threadIdx.x + 64] = 0x40004000u; twos[threadIdx.x + 128] = 0x40004000u; twos[threadIdx.x + 192] = 0x40004000u;
    __syncthreads();
    float o0, o1, o2, o3;
    unsigned int addr = (unsigned int)(size_t)twos + threadIdx.x * 16;   // (LDS addresses are the low 32 bits of the generic pointer's offset)
    asm volatile(
        "v_mov_b32 v100, 0x3c003c00\n v_mov_b32 v101, 0x3c003c00\n v_mov_b32 v102, 0x3c003c00\n v_mov_b32 v103, 0x3c003c00\n"
        "v_mov_b32 v104, 0x3c003c00\n v_mov_b32 v105, 0x3c003c00\n v_mov_b32 v106, 0x3c003c00\n v_mov_b32 v107, 0x3c003c00\n"
        "v_mov_b32 v108, 0\n v_mov_b32 v109, 0\n v_mov_b32 v110, 0\n v_mov_b32 v111, 0\n v_mov_b32 v112, 0\n v_mov_b32 v113, 0\n"
        "v_mov_b32 v114, 0\n v_mov_b32 v115, 0\n v_mov_b32 v116, 0\n v_mov_b32 v117, 0\n v_mov_b32 v118, 0\n v_mov_b32 v119, 0\n"
        "v_mov_b32 v120, 0\n v_mov_b32 v121, 0\n v_mov_b32 v122, 0\n v_mov_b32 v123, 0\n"
        "s_nop 15\n s_nop 15\n"
        "v_mfma_f32_32x32x16_f16 v[108:123], v[100:103], v[104:107], v[108:123]\n"
        ".if %5 > 0\n s_nop %5 - 1\n .endif\n"
        "ds_read_b128 v[100:103], %4\n"
        "s_waitcnt lgkmcnt(0)\n"
        "s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n"
        "v_mov_b32 %0, v108\n v_mov_b32 %1, v113\n v_mov_b32 %2, v118\n v_mov_b32 %3, v123\n"
        : "=v"(o0), "=v"(o1), "=v"(o2), "=v"(o3)
        : "v"(addr), "n"(N)
        : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115",
          "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "memory");
    out[threadIdx.x * 4 + 0] = o0; out[threadIdx.x * 4 + 1] = o1; out[threadIdx.x * 4 + 2] = o2; out[threadIdx.x * 4 + 3] = o3;
}

template <int N>
void run3(float* d) {
    k3<N><<<1, 64>>>(d);
    float h[256];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int wrong = 0; float lo = 1e9f, hi = -1e9f;
    for (float v : h) { wrong += v != 16.0f; lo = v < lo ? v : lo; hi = v > hi ? v : hi; }
    printf("ds_read_b128 into A %2d idle slots after the MFMA: %3d of 256 sampled outputs wrong (values %g .. %g)\n", N, wrong, lo, hi);
}

template <int N, int WHICH>
void run(float* d) {
    k<N, WHICH><<<1, 64>>>(d);
    float h[256];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int wrong = 0; float lo = 1e9f, hi = -1e9f;
    for (float v : h) { wrong += v != 16.0f; lo = v < lo ? v : lo; hi = v > hi ? v : hi; }
    printf("overwrite %c, %2d idle slots after the MFMA: %3d of 256 sampled outputs wrong (values %g .. %g)\n", WHICH ? 'B' : 'A', N, wrong, lo, hi);
}

int main() {
    float* d;
    hipMalloc(&d, 256 * sizeof(float));
    run<0, 0>(d); run<1, 0>(d); run<2, 0>(d); run<3, 0>(d); run<4, 0>(d); run<5, 0>(d); run<6, 0>(d); run<7, 0>(d); run<8, 0>(d);
    run<10, 0>(d); run<12, 0>(d); run<16, 0>(d);
    run<0, 1>(d); run<1, 1>(d); run<2, 1>(d); run<3, 1>(d); run<4, 1>(d); run<5, 1>(d); run<6, 1>(d); run<7, 1>(d); run<8, 1>(d);
    run<10, 1>(d); run<12, 1>(d); run<16, 1>(d);
    run2<0>(d); run2<1>(d); run2<2>(d); run2<3>(d); run2<4>(d); run2<5>(d); run2<6>(d); run2<7>(d); run2<8>(d); run2<9>(d); run2<10>(d);
    run2<12>(d); run2<14>(d); run2<16>(d);
    run3<0>(d); run3<1>(d); run3<2>(d); run3<4>(d); run3<8>(d);
    return 0;
}
