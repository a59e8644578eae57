// Batched full-mesh LBS, float32: verts[F][V][3] for F frames of pose variables.
// Replaces SmplModelLBS.r (src/moshpp/models/smpl_fast_derivatives.py:206-218,243-244 -> psbody
// verts_decorated) evaluated for a whole solved sequence at once.
//
// v0 (correctness baseline): one workgroup = 256 vertices of one frame, posedirs stored vertex-fastest so
// every load/store of a wave is one coalesced 256 B segment.  The MFMA version replaces this file's kernel.
#include "../../include/moshii.h"
#include "moshii_dev.h"

#include <vector>

struct Lbs32Model {
    float* v_shaped;
    float* posedirs_t;   // [9(K-1)][3][Vp]
    float* weights;      // [K][Vp]
    float* J;            // [K][3]
    int Vp;
};

extern "C" {
int moshii_internal_model_dims(moshii_model_t m, int* V, int* K);
const double* moshii_internal_vsh(moshii_model_t m);
const double* moshii_internal_posedirs(moshii_model_t m);
const double* moshii_internal_weights(moshii_model_t m);
const double* moshii_internal_J(moshii_model_t m);
void* moshii_internal_l32(moshii_model_t m);
void moshii_internal_l32_set_valid(moshii_model_t m, int v);
}

namespace {

__global__ void k_cvt_vsh(int n, const double* __restrict__ src, float* __restrict__ dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (float)src[i];
}

__global__ void k_cvt_posedirs(int V, int Vp, int nfeat, const double* __restrict__ src, float* __restrict__ dst) {
    // dst[(q*3 + i)*Vp + v] = src[(v*3 + i)*nfeat + q]
    const size_t total = (size_t)nfeat * 3 * Vp;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int v = (int)(idx % Vp);
        const int qi = (int)(idx / Vp);
        const int q = qi / 3, i = qi % 3;
        dst[idx] = (v < V) ? (float)src[((size_t)v * 3 + i) * nfeat + q] : 0.0f;
    }
}

__global__ void k_cvt_weights(int V, int Vp, int K, const double* __restrict__ src, float* __restrict__ dst) {
    const size_t total = (size_t)K * Vp;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int v = (int)(idx % Vp);
        const int j = (int)(idx / Vp);
        dst[idx] = (v < V) ? (float)src[(size_t)v * K + j] : 0.0f;
    }
}

__global__ __launch_bounds__(256) void k_lbs_f32_v0(ModelDev md, Lbs32Model lm, const float* __restrict__ pose,
                                                     const float* __restrict__ trans, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float smf[];
    const int K = md.K, P = md.P;
    float* fullpose = smf;            // P
    float* Rl = fullpose + P;         // K*9
    float* A = Rl + K * 9;            // K*12 : [Rw | tw - Rw J]
    float* feat = A + K * 12;         // K*9
    const int f = blockIdx.y, tid = threadIdx.x;
    const float* ps = pose + (size_t)f * md.NP;
    for (int d = tid; d < P; d += blockDim.x) {
        float v;
        if (d < md.body_dof) v = ps[d];
        else {
            const int h = d - md.body_dof;
            double acc = md.hands_mean[h];
            for (int i = 0; i < md.hand_dof; ++i) acc += (double)ps[md.body_dof + i] * md.comps[i * md.nhand_full + h];
            v = (float)acc;
        }
        fullpose[d] = v;
    }
    __syncthreads();
    if (tid < K) {
        const float x = fullpose[3 * tid], y = fullpose[3 * tid + 1], z = fullpose[3 * tid + 2];
        const float t2 = x * x + y * y + z * z;
        float a, b;
        if (t2 < 1e-6f) { a = 1.0f - t2 / 6.0f; b = 0.5f - t2 / 24.0f; }
        else { const float t = sqrtf(t2); a = sinf(t) / t; b = (1.0f - cosf(t)) / t2; }
        const float K2[9] = {x * x - t2, x * y, x * z, x * y, y * y - t2, y * z, x * z, y * z, z * z - t2};
        const float Km[9] = {0.0f, -z, y, z, 0.0f, -x, -y, x, 0.0f};
        for (int e = 0; e < 9; ++e) {
            const float id = (e == 0 || e == 4 || e == 8) ? 1.0f : 0.0f;
            const float r = id + a * Km[e] + b * K2[e];
            Rl[tid * 9 + e] = r;
            feat[tid * 9 + e] = r - id;
        }
    }
    __syncthreads();
    if (tid == 0) {
        float Rw[MOSHII_MAXK * 9], tw[MOSHII_MAXK * 3];
        for (int e = 0; e < 9; ++e) Rw[e] = Rl[e];
        for (int i = 0; i < 3; ++i) tw[i] = lm.J[i];
        for (int k = 1; k < K; ++k) {
            const int p = md.parents[k];
            for (int i = 0; i < 3; ++i) {
                for (int j = 0; j < 3; ++j)
                    Rw[k * 9 + i * 3 + j] = Rw[p * 9 + i * 3 + 0] * Rl[k * 9 + j] + Rw[p * 9 + i * 3 + 1] * Rl[k * 9 + 3 + j] + Rw[p * 9 + i * 3 + 2] * Rl[k * 9 + 6 + j];
                tw[k * 3 + i] = Rw[p * 9 + i * 3 + 0] * (lm.J[k * 3 + 0] - lm.J[p * 3 + 0]) + Rw[p * 9 + i * 3 + 1] * (lm.J[k * 3 + 1] - lm.J[p * 3 + 1]) +
                                Rw[p * 9 + i * 3 + 2] * (lm.J[k * 3 + 2] - lm.J[p * 3 + 2]) + tw[p * 3 + i];
            }
        }
        for (int k = 0; k < K; ++k)
            for (int i = 0; i < 3; ++i) {
                for (int j = 0; j < 3; ++j) A[k * 12 + i * 4 + j] = Rw[k * 9 + i * 3 + j];
                A[k * 12 + i * 4 + 3] = tw[k * 3 + i] - (Rw[k * 9 + i * 3 + 0] * lm.J[k * 3 + 0] + Rw[k * 9 + i * 3 + 1] * lm.J[k * 3 + 1] + Rw[k * 9 + i * 3 + 2] * lm.J[k * 3 + 2]);
            }
    }
    __syncthreads();
    const int v = blockIdx.x * blockDim.x + tid;
    if (v >= md.V) return;
    const int nfeat = 9 * (K - 1), Vp = lm.Vp;
    float vp[3] = {lm.v_shaped[v * 3 + 0], lm.v_shaped[v * 3 + 1], lm.v_shaped[v * 3 + 2]};
    for (int q = 0; q < nfeat; ++q) {
        const float fq = feat[9 + q];
        const float* pq = lm.posedirs_t + (size_t)q * 3 * Vp + v;
        vp[0] += pq[0] * fq; vp[1] += pq[Vp] * fq; vp[2] += pq[2 * Vp] * fq;
    }
    float T[12];
    for (int e = 0; e < 12; ++e) T[e] = 0.0f;
    for (int j = 0; j < K; ++j) {
        const float w = lm.weights[(size_t)j * Vp + v];
        if (w != 0.0f) for (int e = 0; e < 12; ++e) T[e] += w * A[j * 12 + e];
    }
    const float* tr = trans + (size_t)f * 3;
    float* o = out + ((size_t)f * md.V + v) * 3;
    for (int i = 0; i < 3; ++i) o[i] = T[i * 4 + 0] * vp[0] + T[i * 4 + 1] * vp[1] + T[i * 4 + 2] * vp[2] + T[i * 4 + 3] + tr[i];
}

}  // namespace

extern "C" int moshii_lbs32_prepare(moshii_model_t m) {
    int V, K;
    moshii_internal_model_dims(m, &V, &K);
    Lbs32Model* lm = (Lbs32Model*)moshii_internal_l32(m);
    const int Vp = (V + 63) & ~63;
    const int nfeat = 9 * (K - 1);
    if (!lm->v_shaped) {
        if (hipMalloc((void**)&lm->v_shaped, (size_t)V * 3 * sizeof(float)) != hipSuccess) return MOSHII_ERR_HIP;
        if (hipMalloc((void**)&lm->posedirs_t, (size_t)std::max(nfeat, 1) * 3 * Vp * sizeof(float)) != hipSuccess) return MOSHII_ERR_HIP;
        if (hipMalloc((void**)&lm->weights, (size_t)K * Vp * sizeof(float)) != hipSuccess) return MOSHII_ERR_HIP;
        if (hipMalloc((void**)&lm->J, (size_t)K * 3 * sizeof(float)) != hipSuccess) return MOSHII_ERR_HIP;
        lm->Vp = Vp;
        hipLaunchKernelGGL(k_cvt_posedirs, dim3(2048), dim3(256), 0, 0, V, Vp, nfeat, moshii_internal_posedirs(m), lm->posedirs_t);
        hipLaunchKernelGGL(k_cvt_weights, dim3(512), dim3(256), 0, 0, V, Vp, K, moshii_internal_weights(m), lm->weights);
    }
    hipLaunchKernelGGL(k_cvt_vsh, dim3((V * 3 + 255) / 256), dim3(256), 0, 0, V * 3, moshii_internal_vsh(m), lm->v_shaped);
    hipLaunchKernelGGL(k_cvt_vsh, dim3(1), dim3(256), 0, 0, K * 3, moshii_internal_J(m), lm->J);
    if (hipDeviceSynchronize() != hipSuccess) return MOSHII_ERR_HIP;
    moshii_internal_l32_set_valid(m, 1);
    return MOSHII_OK;
}

extern "C" hipError_t moshii_launch_lbs_f32(hipStream_t stream, const ModelDev* md, int F, const float* pose,
                                            const float* trans, float* verts, const void* lbs32) {
    const Lbs32Model lm = *(const Lbs32Model*)lbs32;
    const size_t lds = (size_t)(md->P + md->K * 30) * sizeof(float);
    hipLaunchKernelGGL(k_lbs_f32_v0, dim3((md->V + 255) / 256, F), dim3(256), lds, stream, *md, lm, pose, trans, verts);
    return hipGetLastError();
}
