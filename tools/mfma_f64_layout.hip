// Hardware fact check: operand / result lane maps of v_mfma_f64_16x16x4_f64 as JtJAcc (chain_solve.hip) assumes them:
//   lane l supplies A[l & 15][l >> 4] and B[l >> 4][l & 15]; register i of lane l returns D[(l >> 4) + 4 i][l & 15].
// Asymmetric operands (A[m][k] = 1 + m + 100 k, B[k][n] = 3 + 7 n + 1000 k), compared with a host product.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((vector_size(32)));
__global__ void k(double* out) {
    const int l = threadIdx.x;
    const double a = 1.0 + (l & 15) + 100.0 * (l >> 4);
    const double b = 3.0 + 7.0 * (l & 15) + 1000.0 * (l >> 4);
    v4d c = {0.0, 0.0, 0.0, 0.0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int i = 0; i < 4; ++i) out[l * 4 + i] = c[i];
}
int main() {
    double* d; hipMalloc(&d, 256 * 8);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    double h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int i = 0; i < 4; ++i) {
        const int m = (l >> 4) + 4 * i, n = l & 15;
        double ref = 0.0;
        for (int kk = 0; kk < 4; ++kk) ref += (1.0 + m + 100.0 * kk) * (3.0 + 7.0 * n + 1000.0 * kk);
        if (h[l * 4 + i] != ref) { if (bad < 5) printf("lane %d reg %d: got %g expected %g\n", l, i, h[l * 4 + i], ref); ++bad; }
    }
    printf("mfma_f64_16x16x4 layout: %s (%d mismatches)\n", bad ? "MISMATCH" : "as assumed", bad);
    return bad != 0;
}
