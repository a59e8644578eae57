// Batched full-mesh LBS, float32 out: verts[F][V][3] for F frames of pose variables.
// Replaces SmplModelLBS.r (src/moshpp/models/smpl_fast_derivatives.py:206-218,243-244 -> psbody verts_decorated)
// evaluated for a whole solved sequence at once (mesh export of a Stage-II result).
//
// Per (vertex, frame) the work is 2*3*9(K-1) flop of pose correctives (SMPL-H: 2754) against 12 bytes of output, i.e.
// 230 flop/B: on the f32 pipes (157 TF) that is 19x above the HBM ridge, on the f16 matrix pipes (2.5 PF) it sits AT the
// ridge.  So the corrective contraction  C[v,i,f] = sum_q posedirs[v,i,q] * (R - I)[f,q]  runs on
// v_mfma_f32_32x32x16_f16 (f16 operands, scaled so posedirs stay in the normal range; f32 accumulate), and everything
// else is fused behind it so that HBM sees the 12 V F output bytes once:
//
//   k_lbs_prep   one 64-thread workgroup per frame: hand-PCA -> fullpose, Rodrigues, kinematic chain;
//                writes the skinning transforms A[j][f][12] (f32, translation folded with trans[f]) and the pose
//                feature rows featT[f][KP] (f16).
//   k_lbs_mfma   one workgroup (4 waves) per 128 vertices x 64 frames, two workgroups per CU (NT = 2; MOSHII_LBS_NT=4 selects the
//                earlier form: 128 frames, one workgroup per CU, results stored straight from the accumulators):
//                  * the 64 x KP feature panel is staged once in LDS (row pitch 16 x odd bytes: conflict-free b128 reads);
//                  * each wave owns 32 vertices x 3 coordinates x 2 frame tiles = 6 accumulators (96 registers) and streams its
//                    posedirs fragments from a fragment-major copy of the model (one contiguous 1 KiB record per wave-load,
//                    prefetched five k-steps ahead): 3 global + 2 LDS fragment loads feed 6 MFMAs;
//                  * the MFMA runs "features x posedirs": accumulator column (lane) = vertex, accumulator register = frame.
//                    Epilogue on that layout: a lane keeps ITS vertex's <= 8 skinning influences and rest position in
//                    registers for the whole tile; per 16-frame half tile the K joint transforms are staged in LDS
//                    ([joint][frame][12], joint blocks 784 B apart so that different joints fall on different banks, equal
//                    joints broadcast) and each (lane, frame) gathers only its own influences -- sparse skinning, no
//                    W x A GEMM (a dense-blend variant on the matrix pipe exists behind MOSHII_LBS_BLEND=mfma; slower as built);
//                  * the results of a half tile are exchanged through LDS and leave as whole 1536-byte tile rows (128 vertices of
//                    one frame, 16-byte streaming stores): tools/store_pattern.hip measures 5.5 TB/s for that pattern against
//                    3.1 TB/s for 384-byte runs per half-wave.
//                workgroup -> (vertex tile, frame tile) is XCD-aware: all frame tiles of one vertex tile run on the XCD
//                whose L2 already holds that tile's 356 KB of posedirs fragments.
//
// Accuracy: f16 operands give |err| ~ 2^-11 |posedirs| |R - I| sqrt(9(K-1)) ~ 5e-6 m for millimetre-scale correctives
// (tests bound it at 2e-5 m); moshii_lbs_forward_f64 is the reference-precision path.
#include "../../include/moshii.h"
#include "moshii_dev.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <cstdlib>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x3u __attribute__((ext_vector_type(3), aligned(4)));   // 12-byte, dword-aligned store unit
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));   // 16-byte, dword-aligned

#define LBS_NWMAX 8          // skinning influences per vertex (else the launch falls back to the plain kernel)
#define LBS_TV 128           // vertices per workgroup (32 per wave)
#define LBS_TF 128           // frames per workgroup (4 MFMA column tiles)

extern "C" {
int moshii_internal_model_dims(moshii_model_t m, int* V, int* K);
const double* moshii_internal_vsh(moshii_model_t m);
const double* moshii_internal_posedirs(moshii_model_t m);
const double* moshii_internal_weights(moshii_model_t m);
const double* moshii_internal_J(moshii_model_t m);
const double* moshii_internal_weights_host(moshii_model_t m);
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

// max |posedirs| (for the f16 scale)
__global__ void k_absmax(size_t n, const double* __restrict__ src, double* __restrict__ out) {
    __shared__ double red[256];
    double m = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) m = fmax(m, fabs(src[i]));
    red[threadIdx.x] = m;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + o]); __syncthreads(); }
    if (threadIdx.x == 0) out[blockIdx.x] = red[0];
}

// fragment-major f16 posedirs: record (vg, i, ks) holds, for lane l, the 8 values posedirs[v = 32 vg + (l & 31)][i][16 ks + 8 (l >> 5) + e]
__global__ void k_pack_pfrag(int V, int nfeat, int KS, int nvg, double pscale, const double* __restrict__ src, _Float16* __restrict__ dst) {
    const size_t total = (size_t)nvg * 3 * KS * 64 * 8;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int e = (int)(idx & 7);
        const int l = (int)((idx >> 3) & 63);
        size_t r = idx >> 9;
        const int ks = (int)(r % KS); r /= KS;
        const int i = (int)(r % 3);
        const int vg = (int)(r / 3);
        const int v = vg * 32 + (l & 31);
        const int q = ks * 16 + (l >> 5) * 8 + e;
        double val = 0.0;
        if (v < V && q < nfeat) val = src[((size_t)v * 3 + i) * nfeat + q] * pscale;
        dst[idx] = (_Float16)val;
    }
}

// ---- fallback kernel: one workgroup = 256 vertices of one frame (plain f32, dense weights) ----------------
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

// ---- per-frame preparation: joint transforms + f16 pose features ---------------------------------------
__global__ __launch_bounds__(256) void k_lbs_prep(ModelDev md, const float* __restrict__ Jf, int F, int KP,
                                                   const float* __restrict__ pose, const float* __restrict__ trans,
                                                   float* __restrict__ Atr, _Float16* __restrict__ featT,
                                                   _Float16* __restrict__ Ah, int KJ) {
    // one wavefront per frame, four frames per workgroup; everything a wave touches in LDS is its own
    __shared__ float s_fullpose[4][3 * MOSHII_MAXK];
    __shared__ float s_Rl[4][MOSHII_MAXK * 9], s_Rw[4][MOSHII_MAXK * 9], s_tw[4][MOSHII_MAXK * 3];
    const int K = md.K, P = md.P, wv = threadIdx.x >> 6, tid = threadIdx.x & 63;
    const int f = blockIdx.x * 4 + wv;
    if (f >= F) return;   // (whole wavefront; no workgroup barrier below)
    float* fullpose = s_fullpose[wv]; float* Rl = s_Rl[wv]; float* Rw = s_Rw[wv]; float* tw = s_tw[wv];
    const float* ps = pose + (size_t)f * md.NP;
    const int bd = md.body_dof, nhf = md.nhand_full;
    for (int d = tid; d < P; d += 64) {
        float v;
        if (d < bd) v = ps[d];
        else {
            const int h = d - bd;
            const int lo = md.col_lo[h], hi = md.col_hi[h];
            float a0 = md.hands_mean[h], a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;   // four independent chains: the loads overlap
            int i = lo;
            for (; i + 4 <= hi; i += 4) {
                a0 += ps[bd + i] * (float)md.comps[i * nhf + h];
                a1 += ps[bd + i + 1] * (float)md.comps[(i + 1) * nhf + h];
                a2 += ps[bd + i + 2] * (float)md.comps[(i + 2) * nhf + h];
                a3 += ps[bd + i + 3] * (float)md.comps[(i + 3) * nhf + h];
            }
            for (; i < hi; ++i) a0 += ps[bd + i] * (float)md.comps[i * nhf + h];
            v = (a0 + a1) + (a2 + a3);
        }
        fullpose[d] = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    _Float16* frow = featT + (size_t)f * KP;
    if (tid < K) {
        const float x = fullpose[3 * tid], y = fullpose[3 * tid + 1], z = fullpose[3 * tid + 2];
        const float t2 = x * x + y * y + z * z;
        float a, b;
        if (t2 < 1e-6f) { a = 1.0f - t2 / 6.0f; b = 0.5f - t2 / 24.0f; }
        else { const float t = sqrtf(t2); a = sinf(t) / t; b = (1.0f - cosf(t)) / t2; }
        const float K2[9] = {x * x - t2, x * y, x * z, x * y, y * y - t2, y * z, x * z, y * z, z * z - t2};
        const float Km[9] = {0.0f, -z, y, z, 0.0f, -x, -y, x, 0.0f};
#pragma unroll
        for (int e = 0; e < 9; ++e) {
            const float id = (e == 0 || e == 4 || e == 8) ? 1.0f : 0.0f;
            const float r = id + a * Km[e] + b * K2[e];
            Rl[tid * 9 + e] = r;
            if (tid >= 1) frow[(tid - 1) * 9 + e] = (_Float16)(a * Km[e] + b * K2[e]);   // R - I without the cancellation
        }
    }
    for (int q = 9 * (K - 1) + tid; q < KP; q += 64) frow[q] = (_Float16)0.0f;
    // kinematic chain inside the wavefront (in-order LDS), one tree level per step
    if (tid == 0) {
        for (int e = 0; e < 9; ++e) Rw[e] = Rl[e];
        for (int i = 0; i < 3; ++i) tw[i] = Jf[i];
    }
    const int lvl_of = (tid < K) ? md.depth[tid] : -1;
    const int p = (tid < K && tid > 0) ? md.parents[tid] : 0;
    float Jd[3] = {0.0f, 0.0f, 0.0f}, Jme[3] = {0.0f, 0.0f, 0.0f};
    if (tid < K) for (int i = 0; i < 3; ++i) { Jme[i] = Jf[tid * 3 + i]; Jd[i] = Jme[i] - Jf[p * 3 + i]; }
    for (int lvl = 1; lvl <= md.maxdepth; ++lvl) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (lvl_of == lvl) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const float p0 = Rw[p * 9 + i * 3 + 0], p1 = Rw[p * 9 + i * 3 + 1], p2 = Rw[p * 9 + i * 3 + 2];
#pragma unroll
                for (int j = 0; j < 3; ++j) Rw[tid * 9 + i * 3 + j] = p0 * Rl[tid * 9 + j] + p1 * Rl[tid * 9 + 3 + j] + p2 * Rl[tid * 9 + 6 + j];
                tw[tid * 3 + i] = p0 * Jd[0] + p1 * Jd[1] + p2 * Jd[2] + tw[p * 3 + i];
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (tid < K) {   // A_j = [Rw | tw - Rw J_j + trans]  (sum_j w_j = 1 lets the root translation ride in every joint)
        float4* o = reinterpret_cast<float4*>(Atr + ((size_t)tid * F + f) * 12);
        const float* tr = trans + (size_t)f * 3;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float r0 = Rw[tid * 9 + i * 3 + 0], r1 = Rw[tid * 9 + i * 3 + 1], r2 = Rw[tid * 9 + i * 3 + 2];
            const float t3 = tw[tid * 3 + i] - (r0 * Jme[0] + r1 * Jme[1] + r2 * Jme[2]) + tr[i];
            o[i] = make_float4(r0, r1, r2, t3);
            if (Ah != nullptr) {
                // the same row as f16 hi + lo in MFMA A-operand order: block of 8 frames, row rho = 4 slot + comp with
                // slot = 2 (fo & 3) + (fo >> 2), fo = f % 8 (so that accumulator register 4 a + c of half-wave h is frame a + 4 h)
                const int fo = f & 7, slot = 2 * (fo & 3) + (fo >> 2);
                const float vals[4] = {r0, r1, r2, t3};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const _Float16 hi = (_Float16)vals[c];
                    const _Float16 lo = (_Float16)(vals[c] - (float)hi);
                    const size_t base = ((((size_t)(f >> 3) * 3 + i) * 2) * 32 + (4 * slot + c)) * KJ + tid;
                    Ah[base] = hi;
                    Ah[base + (size_t)32 * KJ] = lo;
                }
            }
        }
    }
}

// ---- the MFMA kernel --------------------------------------------------------------------------------------
#define LBS_PITCH(KP) ((KP) + 8)   // halves; (KP + 8) * 2 bytes = 16 (2 KS + 1): 16 x odd  (464 -> 944 = 16 * 59)
#define LBS_FCH 8                  // frame tiles per L2 chunk: 1024 frames of transforms (K x 48 KB) stay L2-resident
#define LBS_JSTRIDE 1552   // bytes between the LDS transform blocks of consecutive joints: 32 frames x 48 B + 16 (bank shift)
#define LBS_JSTRIDE_H 784  // ... for the 16-frame half tiles of the two-workgroups-per-CU variant: 16 x 48 B + 16
#define LBS_SXP 392         // dwords per frame row of the NT = 2 result staging: 128 vertices x 3 + 8 (the two half-waves of a
                            // store instruction are 4 rows apart: 4 x 392 = 32 mod 64 banks)
__host__ __device__ inline size_t lbs_tl_bytes(int K, int nt) { return ((size_t)K * (nt == 2 ? LBS_JSTRIDE_H : LBS_JSTRIDE) + 15) & ~size_t(15); }
// dense-blend epilogue (KSJ = KJ / 16 > 0): A-operand stage of one 8-frame block, 192 rows of KJ halves + 16 bytes, then Sx [8][LBS_SXP]
__host__ __device__ inline size_t lbs_astage_bytes(int ksj) { return (size_t)192 * (ksj * 32 + 16); }
__host__ __device__ inline size_t lbs_region_bytes(int KP, int K, int nt, int ksj = 0) {
    const size_t panel = (size_t)nt * 32 * LBS_PITCH(KP) * 2;
    const size_t epi = (ksj > 0) ? lbs_astage_bytes(ksj) + (size_t)8 * LBS_SXP * 4
                                 : lbs_tl_bytes(K, nt) + (nt == 2 ? (size_t)16 * LBS_SXP * 4 : 0);
    return ((panel > epi ? panel : epi) + 15) & ~size_t(15);
}

// Blend + apply + store for one 32-frame tile, on the accumulator layout (lane = vertex column, register = frame row).
// Tl: this frame tile's joint transforms [K][LBS_JSTRIDE]; jw: this lane's influences {byte offset of the joint block, weight bits}.
// R0, RN: the accumulator registers handled by this call (all 16, or one half = 16 consecutive frames); FOFF: first frame held in Tl.
template <int NWT, int R0 = 0, int RN = 16, int FOFF = 0>
__device__ __forceinline__ void lbs_epilogue(const f32x16& ax, const f32x16& ay, const f32x16& az, float isc,
                                             const char* Tl, const int2 (&jw)[NWT], float vx, float vy, float vz,
                                             int V, int F, int fbase, int v, int lane, float* __restrict__ out, int dbg) {
    const int h = lane >> 5;
#pragma unroll
    for (int r = R0; r < R0 + RN; ++r) {
        const int fr = (r & 3) + 8 * (r >> 2) + 4 * h;   // frame inside the tile
        const char* Tf = Tl + (fr - FOFF) * 48;
        float4 A0[NWT], A1[NWT], A2[NWT];
#pragma unroll
        for (int s2 = 0; s2 < NWT; ++s2) {
            const float4* tp = reinterpret_cast<const float4*>(Tf + jw[s2].x);
            A0[s2] = tp[0]; A1[s2] = tp[1]; A2[s2] = tp[2];
        }
        float4 T0 = {0.f, 0.f, 0.f, 0.f}, T1 = T0, T2 = T0;
#pragma unroll
        for (int s2 = 0; s2 < NWT; ++s2) {
            const float w = __int_as_float(jw[s2].y);
            T0.x += w * A0[s2].x; T0.y += w * A0[s2].y; T0.z += w * A0[s2].z; T0.w += w * A0[s2].w;
            T1.x += w * A1[s2].x; T1.y += w * A1[s2].y; T1.z += w * A1[s2].z; T1.w += w * A1[s2].w;
            T2.x += w * A2[s2].x; T2.y += w * A2[s2].y; T2.z += w * A2[s2].z; T2.w += w * A2[s2].w;
        }
        const float px = vx + isc * ax[r], py = vy + isc * ay[r], pz = vz + isc * az[r];
        const int f = fbase + fr;
        if (f < F && v < V && !(dbg & 16)) {
            // one 12-byte store per lane; 32 consecutive lanes = 32 consecutive vertices = one contiguous 384-byte run
            // (streaming: the 165 MB of output must not evict the posedirs fragments the k-loop re-reads from L2)
            f32x3u val = {T0.x * px + T0.y * py + T0.z * pz + T0.w, T1.x * px + T1.y * py + T1.z * pz + T1.w,
                          T2.x * px + T2.y * py + T2.z * pz + T2.w};
            __builtin_nontemporal_store(val, reinterpret_cast<f32x3u*>(out + ((size_t)f * V + v) * 3));
        }
        if (r & 1) __builtin_amdgcn_sched_barrier(0);   // two frames at a time: their gathers overlap, the live set stays bounded
    }
}

// The NT = 2 form of the epilogue: same blend + apply, but the results of one 16-frame half tile go to an LDS staging area
// Sx[frame 16][LBS_SXP] (vertex-major inside a row) instead of to memory, so that the workgroup can then write whole
// 1536-byte tile rows (tools/store_pattern.hip: 5.5 TB/s for that pattern, 3.1 TB/s for 384-byte runs per half-wave).
template <int NWT, int R0, int FOFF>
__device__ __forceinline__ void lbs_epilogue_lds(const f32x16& ax, const f32x16& ay, const f32x16& az, float isc,
                                                 const char* Tl, const int2 (&jw)[NWT], float vx, float vy, float vz,
                                                 int vl, int lane, float* Sx) {
    const int h = lane >> 5;
#pragma unroll
    for (int r = R0; r < R0 + 8; ++r) {
        const int frl = (r & 3) + 8 * (r >> 2) + 4 * h - FOFF;   // frame inside the half tile
        const char* Tf = Tl + frl * 48;
        float4 A0[NWT], A1[NWT], A2[NWT];
#pragma unroll
        for (int s2 = 0; s2 < NWT; ++s2) {
            const float4* tp = reinterpret_cast<const float4*>(Tf + jw[s2].x);
            A0[s2] = tp[0]; A1[s2] = tp[1]; A2[s2] = tp[2];
        }
        float4 T0 = {0.f, 0.f, 0.f, 0.f}, T1 = T0, T2 = T0;
#pragma unroll
        for (int s2 = 0; s2 < NWT; ++s2) {
            const float w = __int_as_float(jw[s2].y);
            T0.x += w * A0[s2].x; T0.y += w * A0[s2].y; T0.z += w * A0[s2].z; T0.w += w * A0[s2].w;
            T1.x += w * A1[s2].x; T1.y += w * A1[s2].y; T1.z += w * A1[s2].z; T1.w += w * A1[s2].w;
            T2.x += w * A2[s2].x; T2.y += w * A2[s2].y; T2.z += w * A2[s2].z; T2.w += w * A2[s2].w;
        }
        const float px = vx + isc * ax[r], py = vy + isc * ay[r], pz = vz + isc * az[r];
        float* o = Sx + frl * LBS_SXP + vl * 3;
        o[0] = T0.x * px + T0.y * py + T0.z * pz + T0.w;
        o[1] = T1.x * px + T1.y * py + T1.z * pz + T1.w;
        o[2] = T2.x * px + T2.y * py + T2.z * pz + T2.w;
        if (r & 1) __builtin_amdgcn_sched_barrier(0);
    }
}

// NT = frame tiles (of 32) per workgroup.  NT = 4: one workgroup per CU (192 accumulator registers per lane).  NT = 2: half the
// accumulators and half the LDS, so that TWO workgroups share a CU and one's MFMA loop overlaps the other's epilogue
// gathers and the drain of its stores (with one wave per SIMD every phase of a tile is exposed back to back).
// KSJ > 0 (NT = 2 only): the skinning blend runs on the matrix pipe instead of gathering from LDS -- see the epilogue.
template <int NWT, int NT, int KSJ>
__global__ __launch_bounds__(256, (NT == 2) ? 2 : 1) void k_lbs_mfma(Lbs32Model lm, int V, int F, int NVT, int NFT, int NVX,
                                                                      float* __restrict__ out, int dbg_stop) {
    static_assert(KSJ == 0 || NT == 2, "the dense-blend epilogue is written for the two-workgroups-per-CU form");
    constexpr int TF = NT * 32;
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // XCD-aware tile order.  Workgroup b runs on XCD b % 8; each XCD owns the vertex tiles {xcd, xcd + 8, ...} and walks
    // (frame chunk, vertex tile, frame tile in chunk) with the frame tile fastest: the 356 KB of posedirs fragments of
    // a vertex tile are fetched once per chunk and then hit that XCD's L2, and a chunk's K x 1024 x 48 B of joint
    // transforms (2.5 MB for SMPL-H) stays L2-resident while every vertex tile of the XCD sweeps over it.
    const int b = blockIdx.x;
    const int xcd = b & 7, slot = b >> 3;
    const int FCH = dbg_stop >> 8;   // frame tiles per L2 chunk (host: LBS_FCH or MOSHII_LBS_FCH), packed above the debug bits
    dbg_stop &= 255;
    const int fl_t = slot % FCH, rest = slot / FCH;
    const int vt = xcd + 8 * (rest % NVX), ft = (rest / NVX) * FCH + fl_t;
    if (vt >= NVT || ft >= NFT) return;
    const int KP = lm.KP, KS = lm.KS, pitch = LBS_PITCH(KP);
    // LDS: one region R = max(feature panel [main loop], joint transforms of one frame tile [epilogue])
    _Float16* Bp = reinterpret_cast<_Float16*>(lds_raw);                           // main loop: [128][pitch] f16
    const int K = lm.K;
    char* Tl = lds_raw;                                                            // epilogue: [K][LBS_JSTRIDE]
    const int f0 = ft * TF, v0 = vt * LBS_TV;
    // stage the feature panel (rows beyond F are zero) and this tile's influences / rest vertices
    {
        const int chunks = KP / 8;   // 16-byte chunks per row (<= 64: one lane per chunk, one wave per row, no index division)
        for (int r0 = wv; r0 < TF; r0 += 4 * 8) {   // 8 independent 16-byte loads in flight per lane, then the LDS writes
            half8 v[8];
            const int cc = min(lane, chunks - 1);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int fr = min(f0 + r0 + 4 * k, F - 1);
                v[k] = *reinterpret_cast<const half8*>(lm.featT + (size_t)fr * KP + cc * 8);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int r = r0 + 4 * k;
                half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
                if (lane < chunks) *reinterpret_cast<half8*>(Bp + (size_t)r * pitch + cc * 8) = (f0 + r < F) ? v[k] : z;
            }
        }
    }
    // this lane's vertex (accumulator column): influences and rest position stay in registers for the whole tile
    const int vme = v0 + wv * 32 + (lane & 31);
    int2 jw[NWT];
#pragma unroll
    for (int s2 = 0; s2 < NWT; ++s2) jw[s2] = (KSJ == 0) ? lm.sjw[(size_t)vme * NWT + s2] : make_int2(0, 0);
    const float vx = lm.vsh_pad[(size_t)vme * 3 + 0], vy = lm.vsh_pad[(size_t)vme * 3 + 1], vz = lm.vsh_pad[(size_t)vme * 3 + 2];
    __syncthreads();
    // ---- main loop: acc[i][nt] (32 frames x 32 vertices) += featT(nt, ks) x Pfrag(i, ks)^T
    f32x16 acc[3][NT];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][nt][e] = 0.0f;
    const int vg = vt * 4 + wv;   // this wave's 32-vertex group
    const half8* ap = reinterpret_cast<const half8*>(lm.Pfrag) + ((size_t)vg * 3 * KS) * 64 + lane;
    const size_t astride = (size_t)KS * 64;   // coordinate stride in half8 units
    const _Float16* bp = Bp + (size_t)(lane & 31) * pitch + (lane >> 5) * 8;
    // Six rotating A-fragment sets (k-steps t .. t+5) and two B-fragment sets (t, t+1), addressed by NAME so that no register
    // copy ever waits on a load: step t computes from (A[t%6], B[t%2]) while the global loads for A[(t+5)%6] (five
    // k-steps = 1920 MFMA cycles ahead: covers an L2 miss with one wave per SIMD) and the LDS reads for B[(t+1)%2] fly.
    half8 aS[6][3], bS[2][NT];
#define LBS_LOAD_A(SET, KSTEP) { const int kk_ = min((KSTEP), KS - 1); _Pragma("unroll") for (int i = 0; i < 3; ++i) aS[SET][i] = ap[i * astride + (size_t)kk_ * 64]; }
#define LBS_LOAD_B(SET, KSTEP) { const int kk_ = min((KSTEP), KS - 1); _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) bS[SET][nt] = *reinterpret_cast<const half8*>(bp + (size_t)nt * 32 * pitch + kk_ * 16); }
    // (sched_barrier pins the issue order: without it hipcc sinks the prefetch loads down to their first use and the
    //  loop waits vmcnt(0) every k-step -- measured 89 cycles per MFMA instead of 32)
#define LBS_MMA(ASET, BSET) { __builtin_amdgcn_sched_barrier(0); _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) _Pragma("unroll") for (int i = 0; i < 3; ++i) \
        acc[i][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bS[BSET][nt], aS[ASET][i], acc[i][nt], 0, 0, 0); __builtin_amdgcn_sched_barrier(0); }
    LBS_LOAD_A(0, 0) LBS_LOAD_A(1, 1) LBS_LOAD_A(2, 2) LBS_LOAD_A(3, 3) LBS_LOAD_A(4, 4) LBS_LOAD_B(0, 0)
    int ks = 0;
    for (; ks + 6 <= KS; ks += 6) {
        LBS_LOAD_A(5, ks + 5) LBS_LOAD_B(1, ks + 1) LBS_MMA(0, 0)
        LBS_LOAD_A(0, ks + 6) LBS_LOAD_B(0, ks + 2) LBS_MMA(1, 1)
        LBS_LOAD_A(1, ks + 7) LBS_LOAD_B(1, ks + 3) LBS_MMA(2, 0)
        LBS_LOAD_A(2, ks + 8) LBS_LOAD_B(0, ks + 4) LBS_MMA(3, 1)
        LBS_LOAD_A(3, ks + 9) LBS_LOAD_B(1, ks + 5) LBS_MMA(4, 0)
        LBS_LOAD_A(4, ks + 10) LBS_LOAD_B(0, ks + 6) LBS_MMA(5, 1)
    }
    // remainder (KS mod 6 steps): same rotation; the fragments are already in flight, only B needs fetching
    if (ks < KS) { LBS_LOAD_B(1, ks + 1) LBS_MMA(0, 0) ++ks; }
    if (ks < KS) { LBS_LOAD_B(0, ks + 1) LBS_MMA(1, 1) ++ks; }
    if (ks < KS) { LBS_LOAD_B(1, ks + 1) LBS_MMA(2, 0) ++ks; }
    if (ks < KS) { LBS_LOAD_B(0, ks + 1) LBS_MMA(3, 1) ++ks; }
    if (ks < KS) { LBS_LOAD_B(1, ks + 1) LBS_MMA(4, 0) ++ks; }
#undef LBS_LOAD_A
#undef LBS_LOAD_B
#undef LBS_MMA
    // (MOSHII_LBS_STOP=2|4|16|32: phase timing by truncation / ablation -- 2: stop after the k-loop, 4: after the first
    //  frame tile, 16: no stores, 32: every gather reads joint 0)
    if (dbg_stop == 2) { if (acc[0][0][0] + acc[1][1][3] + acc[2][NT - 2][7] + acc[0][NT - 1][9] + acc[1][NT - 2][5] + acc[2][NT - 1][1] == 123.456f) out[0] = 1.0f; return; }
    if (dbg_stop & 32) for (int s2 = 0; s2 < NWT; ++s2) jw[s2].x = 0;
    // ---- epilogue, one 32-frame tile at a time.  The NEXT tile's joint transforms are pulled into registers (all of a
    // lane's <= 24 16-byte loads in flight at once) before the current tile is blended, and dropped into LDS behind an
    // LDS-only barrier -- a plain __syncthreads() would also wait for the tile's global stores to be acknowledged.
    const float isc = lm.inv_pscale;
#define LBS_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
    if constexpr (KSJ > 0) {
        // ---- dense blend on the matrix pipe.  The LDS-gather epilogue below moves 4 influences x 48 B per (vertex, frame) out
        // of LDS and is bound by the LDS return path (66 B/clk/CU measured, with or without bank conflicts).  Here the blended
        // transform rows  T[r][c](v, f) = sum_j W[v][j] A[j][f][r][c]  of 32 vertices x 8 frames come out of three
        // 32x32x16 MFMAs per 16 joints (f16 hi + lo operands: A_hi W_hi + A_lo W_hi + A_hi W_lo, f32 accumulate -- 2^-22
        // relative), with A staged once per 8-frame block in operand order ([row r][hi|lo][rho = 4 slot + c][joint]) by
        // k_lbs_prep and the weights W resident in registers as B fragments: 8 b128 LDS reads per 12 MFMAs.
        // Accumulator register 4 a + c of half-wave h then holds T[r][c] of frame (8 fb + a + 4 h), the frame whose
        // pose-corrected rest position sits in register a + 4 fb of the corrective accumulators.
        constexpr int KJ = KSJ * 16, PA = KJ * 2 + 16;          // A-stage row pitch in bytes (16 x odd: conflict-free b128 rows)
        constexpr int NCHT = (384 * KSJ + 255) / 256;            // 16-byte chunks per thread for one stage (192 rows x 2 KSJ)
        char* Ab = lds_raw;                                      // [192][PA]
        float* Sx = reinterpret_cast<float*>(lds_raw + lbs_astage_bytes(KSJ));   // [8][LBS_SXP]
        const int vl = wv * 32 + (lane & 31), h = lane >> 5;
        const int nfl = min(LBS_TV, V - v0) * 3;
        half8 Wh[KSJ], Wl[KSJ];
        {
            const half8* wp = reinterpret_cast<const half8*>(lm.Wfrag) + ((size_t)vg * KSJ * 2) * 64 + lane;
#pragma unroll
            for (int ks = 0; ks < KSJ; ++ks) { Wh[ks] = wp[(size_t)(ks * 2 + 0) * 64]; Wl[ks] = wp[(size_t)(ks * 2 + 1) * 64]; }
        }
        const int nfb8 = (F + 7) >> 3;
        float4 st[NCHT];
        auto fetch = [&](int q) {   // block of 8 frames number f0 / 8 + q (clamped: blocks past the end are never stored)
            const int fb8 = min((f0 >> 3) + q, nfb8 - 1);
            const float4* src = reinterpret_cast<const float4*>(lm.Ah + (size_t)fb8 * 192 * KJ);
#pragma unroll
            for (int i = 0; i < NCHT; ++i) st[i] = src[min(tid + 256 * i, 384 * KSJ - 1)];
        };
        auto put = [&]() {
#pragma unroll
            for (int i = 0; i < NCHT; ++i) {
                const int c = tid + 256 * i, row = c / (2 * KSJ), cc = c - row * (2 * KSJ);
                if (c < 384 * KSJ) *reinterpret_cast<float4*>(Ab + row * PA + cc * 16) = st[i];
            }
        };
        auto put_rows = [&](int fb) {   // the staged 8 frames x 128 vertices leave as whole 1536-byte tile rows
            if (dbg_stop & 16) return;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int q = tid + 256 * i, row = q / 96, c = (q - row * 96) * 4;
                const int f = fb + row;
                const float4 v4 = *reinterpret_cast<const float4*>(Sx + row * LBS_SXP + c);
                const f32x4u val = {v4.x, v4.y, v4.z, v4.w};
                float* o = out + ((size_t)f * V + v0) * 3 + c;
                if (f < F) {
                    if (c + 4 <= nfl) __builtin_nontemporal_store(val, reinterpret_cast<f32x4u*>(o));
                    else for (int e = 0; e < 4; ++e) if (c + e < nfl) o[e] = val[e];
                }
            }
        };
        fetch(0);
        LBS_LDS_BARRIER();   // every wave is done with the feature panel
        put();
        LBS_LDS_BARRIER();
#pragma unroll
        for (int q = 0; q < 2 * 4; ++q) {   // q = 4 nt + fb: compile-time accumulator indices
            const int nt = q >> 2, fb = q & 3;
            if (q + 1 < 8) fetch(q + 1);
            float ox[4], oy[4], oz[4];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                f32x16 D;
#pragma unroll
                for (int e = 0; e < 16; ++e) D[e] = 0.0f;
#pragma unroll
                for (int ks = 0; ks < KSJ; ++ks) {
                    const char* ar = Ab + ((r * 2) * 32 + (lane & 31)) * PA + (ks * 16 + 8 * h) * 2;
                    const half8 ah = *reinterpret_cast<const half8*>(ar);
                    const half8 al = *reinterpret_cast<const half8*>(ar + 32 * PA);
                    // (Round 1 put 8 idle issue slots behind each of these MFMAs, blaming a wrong tile on the matrix pipe still reading
                    //  its four-register A / B operands when a later write lands on them.  tools/mfma_war_hazard.hip tests exactly that
                    //  on the hardware -- VALU writes and ds_read_b128 returns into the operand registers 0..16 slots after issue, with
                    //  the pipe idle or busy: the product never changes, the hardware interlocks it -- and this variant passes its
                    //  parity test 60 times in a row without the slots.  The slots are gone; whatever corrupted that tile once was not
                    //  an operand hazard.  The issue order stays pinned.)
#define LBS_MFMA_SAFE(A_, B_) { D = __builtin_amdgcn_mfma_f32_32x32x16_f16(A_, B_, D, 0, 0, 0); __builtin_amdgcn_sched_barrier(0); \
                                }
                    LBS_MFMA_SAFE(ah, Wh[ks]) LBS_MFMA_SAFE(al, Wh[ks]) LBS_MFMA_SAFE(ah, Wl[ks])
#undef LBS_MFMA_SAFE
                }
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const float px = vx + isc * acc[0][nt][a + 4 * fb], py = vy + isc * acc[1][nt][a + 4 * fb], pz = vz + isc * acc[2][nt][a + 4 * fb];
                    const float o = D[4 * a + 0] * px + D[4 * a + 1] * py + D[4 * a + 2] * pz + D[4 * a + 3];
                    if (r == 0) ox[a] = o; else if (r == 1) oy[a] = o; else oz[a] = o;
                }
            }
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                float* o = Sx + (a + 4 * h) * LBS_SXP + vl * 3;
                o[0] = ox[a]; o[1] = oy[a]; o[2] = oz[a];
            }
            LBS_LDS_BARRIER();            // Sx complete, every wave done with this A stage
            if (q + 1 < 8) put();
            put_rows(f0 + 8 * q);
            LBS_LDS_BARRIER();            // next A stage visible, Sx free
        }
    } else if constexpr (NT == 4) {
    const int nchunk = K * 96;   // 16-byte chunks of one tile's transforms: [j][frame in tile][3]
    float4 tl0, tl1, tl2, tl3, tl4, tl5, tl6, tl7, tl8, tl9, tl10, tl11, tl12, tl13, tl14, tl15, tl16, tl17, tl18, tl19, tl20, tl21, tl22, tl23;   // (named scalars: hipcc keeps an array of these in scratch)
#define LBS_FETCH1(K_, FB) { const int c = min(tid + 256 * K_, nchunk - 1); const int j = c / 96, rem = c - j * 96, fl2 = rem / 3, q = rem - fl2 * 3; \
        tl##K_ = reinterpret_cast<const float4*>(lm.Atr + ((size_t)j * F + min((FB) + fl2, F - 1)) * 12)[q]; }
#define LBS_PUT1(K_) { const int c = tid + 256 * K_; const int j = c / 96, rem = c - j * 96; if (c < nchunk) *reinterpret_cast<float4*>(Tl + j * LBS_JSTRIDE + rem * 16) = tl##K_; }
#define LBS_FETCH_TL(FB) { LBS_FETCH1(0, FB) LBS_FETCH1(1, FB) LBS_FETCH1(2, FB) LBS_FETCH1(3, FB) LBS_FETCH1(4, FB) LBS_FETCH1(5, FB) LBS_FETCH1(6, FB) LBS_FETCH1(7, FB) LBS_FETCH1(8, FB) LBS_FETCH1(9, FB) LBS_FETCH1(10, FB) LBS_FETCH1(11, FB) LBS_FETCH1(12, FB) LBS_FETCH1(13, FB) LBS_FETCH1(14, FB) LBS_FETCH1(15, FB) LBS_FETCH1(16, FB) LBS_FETCH1(17, FB) LBS_FETCH1(18, FB) LBS_FETCH1(19, FB) LBS_FETCH1(20, FB) LBS_FETCH1(21, FB) LBS_FETCH1(22, FB) LBS_FETCH1(23, FB) }
#define LBS_PUT_TL() { LBS_PUT1(0) LBS_PUT1(1) LBS_PUT1(2) LBS_PUT1(3) LBS_PUT1(4) LBS_PUT1(5) LBS_PUT1(6) LBS_PUT1(7) LBS_PUT1(8) LBS_PUT1(9) LBS_PUT1(10) LBS_PUT1(11) LBS_PUT1(12) LBS_PUT1(13) LBS_PUT1(14) LBS_PUT1(15) LBS_PUT1(16) LBS_PUT1(17) LBS_PUT1(18) LBS_PUT1(19) LBS_PUT1(20) LBS_PUT1(21) LBS_PUT1(22) LBS_PUT1(23) }
    LBS_FETCH_TL(f0)
    LBS_LDS_BARRIER();   // every wave is done with the feature panel
    LBS_PUT_TL()
    LBS_LDS_BARRIER();
    // (one call per frame tile with a compile-time accumulator index: runtime indexing would push acc[][] to scratch)
    LBS_FETCH_TL(f0 + 32)
    lbs_epilogue<NWT>(acc[0][0], acc[1][0], acc[2][0], isc, Tl, jw, vx, vy, vz, V, F, f0 + 0, vme, lane, out, dbg_stop);
    if (dbg_stop == 4) { if (acc[1][1][3] + acc[2][2][7] + acc[0][3][9] + acc[0][1][0] + acc[2][1][1] == 123.456f) out[0] = 1.0f; return; }
    LBS_LDS_BARRIER(); LBS_PUT_TL() LBS_LDS_BARRIER();
    LBS_FETCH_TL(f0 + 64)
    lbs_epilogue<NWT>(acc[0][1], acc[1][1], acc[2][1], isc, Tl, jw, vx, vy, vz, V, F, f0 + 32, vme, lane, out, dbg_stop);
    LBS_LDS_BARRIER(); LBS_PUT_TL() LBS_LDS_BARRIER();
    LBS_FETCH_TL(f0 + 96)
    lbs_epilogue<NWT>(acc[0][2], acc[1][2], acc[2][2], isc, Tl, jw, vx, vy, vz, V, F, f0 + 64, vme, lane, out, dbg_stop);
    LBS_LDS_BARRIER(); LBS_PUT_TL() LBS_LDS_BARRIER();
    lbs_epilogue<NWT>(acc[0][3], acc[1][3], acc[2][3], isc, Tl, jw, vx, vy, vz, V, F, f0 + 96, vme, lane, out, dbg_stop);
#undef LBS_FETCH_TL
#undef LBS_PUT_TL
#undef LBS_FETCH1
#undef LBS_PUT1
    } else {
        // 16-frame half tiles: K x 16 x 48 B of transforms per stage (joint blocks LBS_JSTRIDE_H apart)
#pragma unroll
        for (int s2 = 0; s2 < NWT; ++s2) jw[s2].x = (jw[s2].x / LBS_JSTRIDE) * LBS_JSTRIDE_H;
        const int nchunk = K * 48;   // 16-byte chunks of one half tile: [j][frame][3]
        float4 tl0, tl1, tl2, tl3, tl4, tl5, tl6, tl7, tl8, tl9, tl10, tl11;
#define LBS_FETCH1(K_, FB) { const int c = min(tid + 256 * K_, nchunk - 1); const int j = c / 48, rem = c - j * 48, fl2 = rem / 3, q = rem - fl2 * 3; \
        tl##K_ = reinterpret_cast<const float4*>(lm.Atr + ((size_t)j * F + min((FB) + fl2, F - 1)) * 12)[q]; }
#define LBS_PUT1(K_) { const int c = tid + 256 * K_; const int j = c / 48, rem = c - j * 48; if (c < nchunk) *reinterpret_cast<float4*>(Tl + j * LBS_JSTRIDE_H + rem * 16) = tl##K_; }
#define LBS_FETCH_TL(FB) { LBS_FETCH1(0, FB) LBS_FETCH1(1, FB) LBS_FETCH1(2, FB) LBS_FETCH1(3, FB) LBS_FETCH1(4, FB) LBS_FETCH1(5, FB) LBS_FETCH1(6, FB) LBS_FETCH1(7, FB) LBS_FETCH1(8, FB) LBS_FETCH1(9, FB) LBS_FETCH1(10, FB) LBS_FETCH1(11, FB) }
#define LBS_PUT_TL() { LBS_PUT1(0) LBS_PUT1(1) LBS_PUT1(2) LBS_PUT1(3) LBS_PUT1(4) LBS_PUT1(5) LBS_PUT1(6) LBS_PUT1(7) LBS_PUT1(8) LBS_PUT1(9) LBS_PUT1(10) LBS_PUT1(11) }
        float* Sx = reinterpret_cast<float*>(lds_raw + lbs_tl_bytes(K, 2));   // [16][LBS_SXP] staged results of one half tile
        const int vl = wv * 32 + (lane & 31);
        const int nfl = min(LBS_TV, V - v0) * 3;   // valid floats of a tile row
        // cooperative store of the staged half tile: 16 rows x 96 chunks of 16 bytes, consecutive lanes on consecutive chunks
        auto put_rows = [&](int fb) {
            if (dbg_stop & 16) return;
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const int q = tid + 256 * i, row = q / 96, c = (q - row * 96) * 4;
                const int f = fb + row;
                const float4 v4 = *reinterpret_cast<const float4*>(Sx + row * LBS_SXP + c);   // (16-byte aligned: one ds_read_b128)
                const f32x4u val = {v4.x, v4.y, v4.z, v4.w};
                float* o = out + ((size_t)f * V + v0) * 3 + c;
                if (f < F) {
                    if (c + 4 <= nfl) __builtin_nontemporal_store(val, reinterpret_cast<f32x4u*>(o));
                    else for (int e = 0; e < 4; ++e) if (c + e < nfl) o[e] = val[e];
                }
            }
        };
        LBS_FETCH_TL(f0)
        LBS_LDS_BARRIER();   // every wave is done with the feature panel
        LBS_PUT_TL()
        LBS_LDS_BARRIER();
        LBS_FETCH_TL(f0 + 16)
        lbs_epilogue_lds<NWT, 0, 0>(acc[0][0], acc[1][0], acc[2][0], isc, Tl, jw, vx, vy, vz, vl, lane, Sx);
        LBS_LDS_BARRIER(); LBS_PUT_TL() put_rows(f0 + 0); LBS_LDS_BARRIER();
        LBS_FETCH_TL(f0 + 32)
        lbs_epilogue_lds<NWT, 8, 16>(acc[0][0], acc[1][0], acc[2][0], isc, Tl, jw, vx, vy, vz, vl, lane, Sx);
        LBS_LDS_BARRIER(); LBS_PUT_TL() put_rows(f0 + 16); LBS_LDS_BARRIER();
        LBS_FETCH_TL(f0 + 48)
        lbs_epilogue_lds<NWT, 0, 0>(acc[0][1], acc[1][1], acc[2][1], isc, Tl, jw, vx, vy, vz, vl, lane, Sx);
        LBS_LDS_BARRIER(); LBS_PUT_TL() put_rows(f0 + 32); LBS_LDS_BARRIER();
        lbs_epilogue_lds<NWT, 8, 16>(acc[0][1], acc[1][1], acc[2][1], isc, Tl, jw, vx, vy, vz, vl, lane, Sx);
        LBS_LDS_BARRIER(); put_rows(f0 + 48);
    }
#undef LBS_FETCH_TL
#undef LBS_PUT_TL
#undef LBS_FETCH1
#undef LBS_PUT1
#undef LBS_LDS_BARRIER
}

}  // namespace

static void free_ptr(void* p) { if (p) hipFree(p); }

extern "C" void moshii_lbs32_free(void* l32) {
    Lbs32Model* lm = (Lbs32Model*)l32;
    free_ptr(lm->v_shaped); free_ptr(lm->posedirs_t); free_ptr(lm->weights); free_ptr(lm->J);
    free_ptr(lm->Pfrag); free_ptr(lm->vsh_pad); free_ptr(lm->sjw);
    free_ptr(lm->Atr); free_ptr(lm->featT); free_ptr(lm->Ah); free_ptr(lm->Wfrag);
    memset(lm, 0, sizeof(*lm));
}

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
        // ---- MFMA-path model copy
        lm->mfma_ok = 0;
        const int Vp128 = (V + LBS_TV - 1) / LBS_TV * LBS_TV;
        const int KS = (nfeat + 15) / 16, KP = KS * 16;
        const int nvg = Vp128 / 32;
        lm->Vp128 = Vp128; lm->KS = KS; lm->KP = KP;
        // per-vertex influence lists (host; once per model)
        const double* wh = moshii_internal_weights_host(m);
        int NW = 1;
        for (int v = 0; v < V; ++v) {
            int c = 0;
            for (int j = 0; j < K; ++j) c += (wh[(size_t)v * K + j] != 0.0) ? 1 : 0;
            NW = std::max(NW, c);
        }
        bool ok = nfeat > 0 && NW <= LBS_NWMAX;
        lm->K = K; lm->NW = NW;
        const int NWT = (NW <= 4) ? 4 : 8;   // influences padded to the kernel's compile-time width (joint 0, weight 0)
        lm->NW = NWT;
        std::vector<int> sjw(ok ? (size_t)Vp128 * NWT * 2 : 0, 0);   // {byte offset of the joint's [32][12] f32 block, weight bits}
        for (int v = 0; v < V && ok; ++v) {
            int c = 0;
            for (int j = 0; j < K; ++j) {
                const double w = wh[(size_t)v * K + j];
                if (w != 0.0) {
                    const float wf = (float)w;
                    int bits; memcpy(&bits, &wf, 4);
                    sjw[((size_t)v * NWT + c) * 2 + 0] = j * 1552;   // LBS_JSTRIDE
                    sjw[((size_t)v * NWT + c) * 2 + 1] = bits;
                    ++c;
                }
            }
        }
        if (ok) {
            double* d_part = nullptr;
            if (hipMalloc((void**)&d_part, 256 * sizeof(double)) != hipSuccess) return MOSHII_ERR_HIP;
            hipLaunchKernelGGL(k_absmax, dim3(256), dim3(256), 0, 0, (size_t)V * 3 * nfeat, moshii_internal_posedirs(m), d_part);
            double part[256];
            if (hipMemcpy(part, d_part, sizeof(part), hipMemcpyDeviceToHost) != hipSuccess) return MOSHII_ERR_HIP;
            hipFree(d_part);
            double amax = 0.0;
            for (double p : part) amax = std::max(amax, p);
            // power-of-two scale that lifts the largest corrective to ~2^13: small entries stay normal in f16
            double pscale = 1.0;
            if (amax > 0.0) pscale = std::ldexp(1.0, 13 - (int)std::ceil(std::log2(amax)));
            lm->inv_pscale = (float)(1.0 / pscale);
            if (hipMalloc((void**)&lm->Pfrag, (size_t)nvg * 3 * KS * 64 * 8 * sizeof(_Float16)) != hipSuccess) return MOSHII_ERR_HIP;
            if (hipMalloc((void**)&lm->vsh_pad, (size_t)Vp128 * 3 * sizeof(float)) != hipSuccess) return MOSHII_ERR_HIP;
            if (hipMalloc((void**)&lm->sjw, sjw.size() * sizeof(int)) != hipSuccess) return MOSHII_ERR_HIP;
            hipMemcpy(lm->sjw, sjw.data(), sjw.size() * sizeof(int), hipMemcpyHostToDevice);
            hipLaunchKernelGGL(k_pack_pfrag, dim3(4096), dim3(256), 0, 0, V, nfeat, KS, nvg, pscale, moshii_internal_posedirs(m), lm->Pfrag);
            // dense-blend operands: weights as f16 hi + lo, B-operand fragment-major (lane l of group vg: vertex 32 vg + (l & 31),
            // joints 16 ks + 8 (l >> 5) .. + 8)
            const int KJ = (K + 15) / 16 * 16, KSJ = KJ / 16;
            lm->KJ = KJ;
            if (KJ <= 64) {
                std::vector<_Float16> wf((size_t)nvg * KSJ * 2 * 64 * 8);
                for (int g = 0; g < nvg; ++g)
                    for (int ks = 0; ks < KSJ; ++ks)
                        for (int l = 0; l < 64; ++l)
                            for (int e = 0; e < 8; ++e) {
                                const int v = g * 32 + (l & 31), j = ks * 16 + 8 * (l >> 5) + e;
                                const float w = (v < V && j < K) ? (float)wh[(size_t)v * K + j] : 0.0f;
                                const _Float16 hi = (_Float16)w;
                                const _Float16 lo = (_Float16)(w - (float)hi);
                                wf[((((size_t)g * KSJ + ks) * 2 + 0) * 64 + l) * 8 + e] = hi;
                                wf[((((size_t)g * KSJ + ks) * 2 + 1) * 64 + l) * 8 + e] = lo;
                            }
                if (hipMalloc((void**)&lm->Wfrag, wf.size() * sizeof(_Float16)) != hipSuccess) return MOSHII_ERR_HIP;
                hipMemcpy(lm->Wfrag, wf.data(), wf.size() * sizeof(_Float16), hipMemcpyHostToDevice);
            }
            lm->mfma_ok = 1;
        }
    }
    hipLaunchKernelGGL(k_cvt_vsh, dim3((V * 3 + 255) / 256), dim3(256), 0, 0, V * 3, moshii_internal_vsh(m), lm->v_shaped);
    hipLaunchKernelGGL(k_cvt_vsh, dim3(1), dim3(256), 0, 0, K * 3, moshii_internal_J(m), lm->J);
    if (lm->mfma_ok) {
        hipMemset(lm->vsh_pad, 0, (size_t)lm->Vp128 * 3 * sizeof(float));
        hipLaunchKernelGGL(k_cvt_vsh, dim3((V * 3 + 255) / 256), dim3(256), 0, 0, V * 3, moshii_internal_vsh(m), lm->vsh_pad);
    }
    if (hipDeviceSynchronize() != hipSuccess) return MOSHII_ERR_HIP;
    moshii_internal_l32_set_valid(m, 1);
    return MOSHII_OK;
}

extern "C" hipError_t moshii_launch_lbs_f32(hipStream_t stream, const ModelDev* md, int F, const float* pose,
                                            const float* trans, float* verts, void* lbs32) {
    Lbs32Model* lmp = (Lbs32Model*)lbs32;
    const bool force_v0 = getenv("MOSHII_LBS_PLAIN") != nullptr;
    if (!lmp->mfma_ok || force_v0) {
        const Lbs32Model lm = *lmp;
        const size_t lds = (size_t)(md->P + md->K * 30) * sizeof(float);
        hipLaunchKernelGGL(k_lbs_f32_v0, dim3((md->V + 255) / 256, F), dim3(256), lds, stream, *md, lm, pose, trans, verts);
        return hipGetLastError();
    }
    if (F > lmp->Fcap) {   // per-call scratch grows to the largest F seen (not stream-ordered: sync first)
        hipStreamSynchronize(stream);
        free_ptr(lmp->Atr); free_ptr(lmp->featT); free_ptr(lmp->Ah);
        lmp->Atr = nullptr; lmp->featT = nullptr; lmp->Ah = nullptr; lmp->Fcap = 0;
        hipError_t e = hipMalloc((void**)&lmp->Atr, (size_t)md->K * F * 12 * sizeof(float));
        if (e != hipSuccess) return e;
        e = hipMalloc((void**)&lmp->featT, (size_t)F * lmp->KP * sizeof(_Float16));
        if (e != hipSuccess) return e;
        if (lmp->Wfrag) {   // A-operand copy of the transforms; joints K .. KJ-1 stay zero
            const size_t nah = (size_t)((F + 7) / 8) * 192 * lmp->KJ;
            e = hipMalloc((void**)&lmp->Ah, nah * sizeof(_Float16));
            if (e != hipSuccess) return e;
            e = hipMemset(lmp->Ah, 0, nah * sizeof(_Float16));
            if (e != hipSuccess) return e;
        }
        lmp->Fcap = F;
    }
    const Lbs32Model lm = *lmp;
    // skinning blend: "mfma" = dense W x A on the matrix pipe (needs K <= 64), "gather" = sparse LDS gather of the influences
    bool dense = lm.Wfrag != nullptr && lm.KJ <= 64 && (lm.KJ == 16 || lm.KJ == 32 || lm.KJ == 64);
    if (const char* es = getenv("MOSHII_LBS_BLEND")) dense = dense && strcmp(es, "gather") != 0;
    else dense = false;
    hipLaunchKernelGGL(k_lbs_prep, dim3((F + 3) / 4), dim3(256), 0, stream, *md, lm.J, F, lm.KP, pose, trans, lm.Atr, lm.featT,
                       dense ? lm.Ah : (_Float16*)nullptr, lm.KJ);
    // frame tiles per workgroup: 2 = two workgroups per CU (default: measured 359 vs 394 us at F=4000), 4 = one per CU
    int nt = 2;
    if (const char* es = getenv("MOSHII_LBS_NT")) nt = (atoi(es) == 4) ? 4 : 2;
    if (dense) nt = 2;
    const int ksj = dense ? lm.KJ / 16 : 0;
    const int TF = nt * 32;
    const int NVT = lm.Vp128 / LBS_TV, NFT = (F + TF - 1) / TF;
    const int NVX = (NVT + 7) / 8;                         // vertex tiles per XCD
    int fch = LBS_FCH;
    if (const char* es = getenv("MOSHII_LBS_FCH")) fch = std::max(1, atoi(es));
    const int NCH = (NFT + fch - 1) / fch;         // frame chunks
    const int grid = 8 * NCH * NVX * fch;
    const size_t lds = lbs_region_bytes(lm.KP, lm.K, nt, ksj);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    auto kern = (nt == 2) ? ((lm.NW == 4) ? k_lbs_mfma<4, 2, 0> : k_lbs_mfma<8, 2, 0>) : ((lm.NW == 4) ? k_lbs_mfma<4, 4, 0> : k_lbs_mfma<8, 4, 0>);
    if (ksj == 4) kern = k_lbs_mfma<4, 2, 4>;
    else if (ksj == 2) kern = k_lbs_mfma<4, 2, 2>;
    else if (ksj == 1) kern = k_lbs_mfma<4, 2, 1>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (nt == 2) ? 80 * 1024 : 160 * 1024);
    if (e != hipSuccess) return e;
    int dbg_stop = 0;
    if (const char* es = getenv("MOSHII_LBS_STOP")) dbg_stop = atoi(es) & 255;
    dbg_stop |= fch << 8;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, lm, md->V, F, NVT, NFT, NVX, verts, dbg_stop);
    return hipGetLastError();
}
