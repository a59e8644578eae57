// Batched full-mesh LBS, float32 out: verts[F][V][3] for F frames of pose variables.
// Replaces SmplModelLBS.r (src/moshpp/models/smpl_fast_derivatives.py:206-218,243-244 -> psbody verts_decorated)
// evaluated for a whole solved sequence at once (mesh export of a Stage-II result).
//
// Per (vertex, frame) the work is 2*3*9(K-1) flop of pose correctives (SMPL-H: 2754) against 12 bytes of output, i.e.
// 230 flop/B: on the f32 pipes (157 TF) that is 19x above the HBM ridge, on the f16 matrix pipes (2.5 PF) it sits AT the
// ridge.  So the corrective contraction  C[v,i,f] = sum_q posedirs[v,i,q] * (R - I)[f,q]  runs on
// v_mfma_f32_16x16x32_f16 (f16 operands, scaled so posedirs stay in the normal range; f32 accumulate), and everything
// else is fused behind it so that HBM sees the 12 V F output bytes once.
//
//   k_lbs_prep   one wavefront per frame: hand-PCA -> fullpose, Rodrigues, kinematic chain; writes the skinning transforms
//                Atr[16-frame block][joint][frame][12] (f32, translation folded with trans[f]) -- one contiguous K x 768 B
//                block per 16 frames, the LDS image the export kernel copies -- and the pose features as MFMA B fragments
//                featF[128-frame tile][k-step][16-frame block][lane][8] (f16).
//   k_lbs_tile   ONE persistent workgroup per CU (4 waves, one per SIMD, the whole register file), a tile = 128 vertices x
//                128 frames, a wave = 32 vertices x 128 frames = 2 x 8 MFMA tiles of 16 x 16 per coordinate (192 accumulator
//                registers).  Round 3 rewrite; what the round-2 kernel (128 x 64 tiles, two workgroups per CU, 32x32x16
//                MFMAs with lane = vertex) lost its time on, and what replaces it:
//                  * its k-loop ran at 44 % of the MFMA rate behind an up-front copy of the whole feature panel (59 KB per
//                    tile at the ~11 B/clk a CU gets when every CU fetches at once).  Now the features stream through a
//                    three-slot LDS ring, one 8 KB k-step chunk at a time, loaded two steps ahead -- nothing is staged
//                    before the first MFMA -- and a tile covers 128 frames, so the posedirs fragments (the big operand, 19 MB)
//                    are streamed F / 128 times instead of F / 64;
//                  * its epilogue gathered each (vertex, frame)'s four joint transforms from LDS with lane = vertex: the 16
//                    lanes of a ds_read_b128 group hit 16 random joints, 2-3 of them on the same banks (66 B/clk/CU measured
//                    against 256).  The accumulators are now the other way round -- lane = frame, register = vertex -- so the
//                    16 lanes of a group read 16 consecutive frames of ONE joint (48 B apart, joint blocks 768 B = 3 x 256
//                    apart: conflict-free by construction), the lane's joint addresses and weights are per-(register, influence)
//                    constants kept in registers for the whole tile, and no address arithmetic is left in the loop;
//                  * the joint transforms of a 16-frame half tile are ONE contiguous block in memory and reach LDS by LDS-DMA
//                    (global_load_lds_dwordx4, double buffered, issued a half tile ahead): no staging registers, no ds_write
//                    pass, and the wait is a counted vmcnt that leaves the row stores in flight;
//                  * results leave through an LDS exchange as whole 1536-byte tile rows (16-byte streaming stores) as before;
//                    the exchange rows are 386 dwords apart, which keeps the lane = frame writes at two lanes per bank.
//                workgroup -> tile order is XCD-aware: XCD x owns the vertex tiles x, x + 8, ... (2.5 MB of fragments, L2
//                resident) and its 32 workgroups walk (frame tile, vertex tile) together.
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
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));   // 16-byte, dword-aligned store unit

#define LBS_NWMAX 8          // skinning influences per vertex (else the launch falls back to the plain kernel)
#define LBS_TV 128           // vertices per tile (32 per wave)
#define LBS_TF 128           // frames per tile (8 MFMA column tiles of 16)
#define LBS_SXP 386          // dwords per row of the result exchange: 128 vertices x 3 + 2.  A ds_write_b32 of the exchange has its 32
                             // lanes on 16 frames (rows) x 2 vertices 12 dwords apart: with rows 2 banks apart that is 16 distinct even
                             // banks, each hit twice (free on ds_write_b32); a 16-byte-aligned pitch would make it 4-way.  Rows are
                             // therefore 8-byte aligned only and are read back as pairs of ds_read_b64.
#define LBS_RING 3           // slots of the feature ring
#define LBS_CHUNK 8192       // bytes of one k-step's feature fragments: 8 frame blocks x 64 lanes x 16 B
#define LBS_JBYTES 768       // one joint's transforms for the 16 frames of a half tile (16 x 12 floats = 3 x 256 B)

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

// fragment-major f16 posedirs (MFMA A operand of v_mfma_f32_16x16x32_f16): record (vg, i, ks) holds, for lane l, the 8 values
// posedirs[v = 16 vg + (l & 15)][i][32 ks + 8 (l >> 4) + e]
__global__ void k_pack_pfrag(int V, int nfeat, int KS, int nvg, double pscale, const double* __restrict__ src, _Float16* __restrict__ dst) {
    const size_t total = (size_t)nvg * 3 * KS * 64 * 8;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int e = (int)(idx & 7);
        const int l = (int)((idx >> 3) & 63);
        size_t r = idx >> 9;
        const int ks = (int)(r % KS); r /= KS;
        const int i = (int)(r % 3);
        const int vg = (int)(r / 3);
        const int v = vg * 16 + (l & 15);
        const int q = ks * 32 + (l >> 4) * 8 + e;
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
__global__ __launch_bounds__(256, 4) void k_lbs_prep(ModelDev md, const float* __restrict__ Jf, int F, int KS, int KJ,
                                                   const float* __restrict__ pose, const float* __restrict__ trans,
                                                   float* __restrict__ Atr, _Float16* __restrict__ featF, long long* __restrict__ stamps) {
#define PREP_STAMP(K) { if (stamps != nullptr && blockIdx.x == 0 && threadIdx.x == 0) stamps[K] = clock64(); }
    // one wavefront per frame, four frames per workgroup; everything a wave touches in LDS is its own.  27 KB of static LDS + the
    // hand-component matrix (dynamic: hand_dof x nhand_full floats, 8.6 KB for SMPL-H / SMPL-X) and 81 registers: four workgroups
    // share a CU, so the 1000 workgroups of a 4000-frame export are resident at once.  (Its first form -- 69 KB, 169 registers, two
    // workgroups per CU -- timed the same 29 us under the profiler: the kernel is one latency chain per wave, cold loads and ~10
    // dependent tree levels, and residency was not what bounded it.)
    __shared__ float s_fullpose[4][3 * MOSHII_MAXK];
    __shared__ float s_R[4][MOSHII_MAXK * 9], s_tw[4][MOSHII_MAXK * 3];      // local rotations, turned into world rotations in place
    __shared__ float s_pose[4][3 * MOSHII_MAXK], s_hm[4][128], s_J[4][3 * MOSHII_MAXK];
    extern __shared__ float smf[];
    float* const s_comps = smf;
    __shared__ __attribute__((aligned(16))) _Float16 s_feat[4][16 * 32];   // the four frames' feature rows (KS <= 16 k-steps of 32), zero padded
    const int K = md.K, P = md.P, wv = threadIdx.x >> 6, tid = threadIdx.x & 63;
    const int fw = blockIdx.x * 4 + wv;          // this wave's frame; the last workgroup's spare waves redo frame F - 1 and write nothing
    const int f = min(fw, F - 1);
    PREP_STAMP(0)
    for (int q = tid; q < 16 * 32; q += 64) s_feat[wv][q] = (_Float16)0.0f;
    float* fullpose = s_fullpose[wv]; float* Rw = s_R[wv]; float* tw = s_tw[wv];
    const float* ps = pose + (size_t)f * md.NP;
    const int bd = md.body_dof, nhf = md.nhand_full;
    // Everything the frame needs from memory is fetched in ONE round of independent loads at the top (with 4 000 waves starting at
    // once a dependent round trip costs 1-2 000 cycles, and the first version made some thirty of them in a row: per-column loop
    // bounds, then components one at a time, then tree depth / parents, then the parents' joints): the hand-component matrix
    // (hand_dof x nhand_full <= 90 x 90, f64 -> f32, shared by the workgroup's four frames), the pose row, the hands' mean, the
    // tree and the rest joints -- staged in LDS, from where the rest of the kernel reads.
    const int hd = md.hand_dof, ncomp = hd * nhf;
    {
        double cst[16];    // (16 x 256 entries per pass: one pass for the 24 x 90 of the default hand spaces)
#pragma unroll
        for (int u = 0; u < 16; ++u) { const int i = threadIdx.x + 256 * u; cst[u] = (i < ncomp) ? md.comps[i] : 0.0; }
        float pv[3], hmv[2], jv[3];
#pragma unroll
        for (int u = 0; u < 3; ++u) { const int i = tid + 64 * u; pv[u] = (i < md.NP) ? ps[i] : 0.0f; }
#pragma unroll
        for (int u = 0; u < 2; ++u) { const int i = tid + 64 * u; hmv[u] = (i < nhf) ? (float)md.hands_mean[i] : 0.0f; }
#pragma unroll
        for (int i = 0; i < 3; ++i) jv[i] = (tid < K) ? Jf[tid * 3 + i] : 0.0f;
#pragma unroll
        for (int u = 0; u < 16; ++u) { const int i = threadIdx.x + 256 * u; if (i < ncomp) s_comps[i] = (float)cst[u]; }
        for (int i = threadIdx.x + 4096; i < ncomp; i += 256) s_comps[i] = (float)md.comps[i];   // (larger hand spaces: the rest, plainly)
#pragma unroll
        for (int u = 0; u < 3; ++u) { const int i = tid + 64 * u; if (i < md.NP) s_pose[wv][i] = pv[u]; }
#pragma unroll
        for (int u = 0; u < 2; ++u) { const int i = tid + 64 * u; if (i < nhf) s_hm[wv][i] = hmv[u]; }
#pragma unroll
        for (int i = 0; i < 3; ++i) if (tid < K) s_J[wv][tid * 3 + i] = jv[i];
    }
    const int lvl_of = (tid < K) ? md.depth[tid] : -1;
    const int p = (tid < K && tid > 0) ? md.parents[tid] : 0;
    __syncthreads();
    // fullpose = [pose[:body_dof], hands_mean + pose_hand . components]  (block diagonal: the other hand's entries are exact zeros)
    for (int d = tid; d < P; d += 64) {
        float v;
        if (d < bd) v = s_pose[wv][d];
        else {
            const int h = d - bd;
            float a0 = s_hm[wv][h], a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
            int i = 0;
            for (; i + 4 <= hd; i += 4) {
                a0 += s_pose[wv][bd + i] * s_comps[i * nhf + h];
                a1 += s_pose[wv][bd + i + 1] * s_comps[(i + 1) * nhf + h];
                a2 += s_pose[wv][bd + i + 2] * s_comps[(i + 2) * nhf + h];
                a3 += s_pose[wv][bd + i + 3] * s_comps[(i + 3) * nhf + h];
            }
            for (; i < hd; ++i) a0 += s_pose[wv][bd + i] * s_comps[i * nhf + h];
            v = (a0 + a1) + (a2 + a3);
        }
        fullpose[d] = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    PREP_STAMP(1)
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
            Rw[tid * 9 + e] = r;
            if (tid >= 1) s_feat[wv][(tid - 1) * 9 + e] = (_Float16)(a * Km[e] + b * K2[e]);   // R - I without the cancellation
        }
    }
    PREP_STAMP(2)
    // kinematic chain inside the wavefront (in-order LDS), one tree level per step
    if (tid == 0) for (int i = 0; i < 3; ++i) tw[i] = s_J[wv][i];      // (the root's local rotation is its world rotation)
    float Jd[3] = {0.0f, 0.0f, 0.0f}, Jme[3] = {0.0f, 0.0f, 0.0f};
    if (tid < K) for (int i = 0; i < 3; ++i) { Jme[i] = s_J[wv][tid * 3 + i]; Jd[i] = Jme[i] - s_J[wv][p * 3 + i]; }
    for (int lvl = 1; lvl <= md.maxdepth; ++lvl) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (lvl_of == lvl) {
            float rl[9];       // this joint's local rotation, read before its slot takes the world rotation
#pragma unroll
            for (int e = 0; e < 9; ++e) rl[e] = Rw[tid * 9 + e];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const float p0 = Rw[p * 9 + i * 3 + 0], p1 = Rw[p * 9 + i * 3 + 1], p2 = Rw[p * 9 + i * 3 + 2];
#pragma unroll
                for (int j = 0; j < 3; ++j) Rw[tid * 9 + i * 3 + j] = p0 * rl[j] + p1 * rl[3 + j] + p2 * rl[6 + j];
                tw[tid * 3 + i] = p0 * Jd[0] + p1 * Jd[1] + p2 * Jd[2] + tw[p * 3 + i];
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    PREP_STAMP(3)
    if (tid < K && fw < F) {   // A_j = [Rw | tw - Rw J_j + trans]  (sum_j w_j = 1 lets the root translation ride in every joint)
        f32x4* o = reinterpret_cast<f32x4*>(Atr + (((size_t)(f >> 4) * KJ + tid) * 16 + (f & 15)) * 12);
        const float* tr = trans + (size_t)f * 3;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float r0 = Rw[tid * 9 + i * 3 + 0], r1 = Rw[tid * 9 + i * 3 + 1], r2 = Rw[tid * 9 + i * 3 + 2];
            const float t3 = tw[tid * 3 + i] - (r0 * Jme[0] + r1 * Jme[1] + r2 * Jme[2]) + tr[i];
            const f32x4 row = {r0, r1, r2, t3};
            o[i] = row;
        }
    }
    PREP_STAMP(4)
    // B fragments of v_mfma_f32_16x16x32_f16: features 8 g .. 8 g + 7 of frame f are the 16 bytes of record (f / 128, g / 4,
    // (f / 16) % 8), lane (f % 16) + 16 (g % 4).  The workgroup's four frames are four consecutive lanes: thread (g, frame) writes
    // one 16-byte piece, four threads a 64-byte run (the first version wrote every feature as a 2-byte store of its own: 2.4 M write
    // requests per call, the waves a third of their time at the issue stage behind them).
    __syncthreads();
    {
        const int fi = threadIdx.x & 3, g = threadIdx.x >> 2, fo = blockIdx.x * 4 + fi;
        if (g < KS * 4 && fo < F) {
            const f32x4 piece = *reinterpret_cast<const f32x4*>(&s_feat[fi][g * 8]);
            _Float16* dst = featF + ((((size_t)(fo >> 7) * KS + (g >> 2)) * 8 + ((fo >> 4) & 7)) * 64 + (fo & 15) + 16 * (g & 3)) * 8;
            *reinterpret_cast<f32x4*>(dst) = piece;
        }
    }
    PREP_STAMP(5)
#undef PREP_STAMP
}

// ---- the export kernel ------------------------------------------------------------------------------------
// LDS map (bytes; every offset is a compile-time constant so that the epilogue's reads carry their buffer base as an immediate):
//   [0, 43008)        transforms of the half tile in work, buffer 0   (KJ <= 56 joints x 768 B)
//   [43008, 86016)    buffer 1
//   [86016, 110592)   feature ring, 3 slots of 8 KiB
//   [110592, 160000)  result exchange, two buffers of 16 rows of LBS_SXP dwords
#define LBS_TLMAX 43008
#define LBS_KJMAX 56
#define LBS_OFF_RING (2 * LBS_TLMAX)
#define LBS_OFF_SX (LBS_OFF_RING + LBS_RING * LBS_CHUNK)
#define LBS_SXBYTES (16 * LBS_SXP * 4)
#define LBS_LDS_BYTES (LBS_OFF_SX + 2 * LBS_SXBYTES)

#define LBS_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
// s_waitcnt vmcnt(N) only (expcnt / lgkmcnt fields at their "no wait" values): the LDS-DMA pieces are older than the N memory
// instructions issued behind them, which stay in flight
#define LBS_WAIT_VM(N) __builtin_amdgcn_s_waitcnt(0x0F70 | ((N) & 15) | (((N) >> 4) << 14))

// NWT: skinning influences per vertex (padded).  NVG: 16-vertex groups per wave -- 2: four waves per workgroup, one per SIMD, the
// whole register file each; 1: eight waves, two per SIMD with 256 registers each, so that one wave's LDS / memory latency is
// covered by the other's arithmetic (the workgroup tile is 128 vertices x 128 frames either way).
template <int NWT, int NVG>
__global__ __launch_bounds__(512 / NVG, 2 / NVG) void k_lbs_tile(Lbs32Model lm, int V, int F, int NVT, int NFT, float* __restrict__ out, int dbg) {
    constexpr int TPB = 512 / NVG, NWAVE = TPB / 64, WV = 16 * NVG;   // threads, waves, vertices per wave
    constexpr int RPW = 16 / NWAVE;                                  // exchange rows a wave stores per half tile
    // B-fragment sets: two (the fragments of step t + 1 are read while step t multiplies) for the four-wave form; ONE for the
    // eight-wave form -- its 256 registers per wave do not hold a second set: tried with two A sets instead of three, the loop was
    // free of scratch but the tile's vertex records were parked there, 266 us against 240
    constexpr int NB = NVG;
    constexpr int NA = 3;                                            // A-fragment sets = k-steps of lead of the posedirs loads + 1
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int KS = lm.KS, KJ = lm.KJ;
    const int tlb = KJ * LBS_JBYTES;                       // bytes of one half tile's transforms (a multiple of 1024)
    char* ring = lds_raw + LBS_OFF_RING;                   // [LBS_RING][8 frame blocks][64 lanes][16 B]
    char* Sx = lds_raw + LBS_OFF_SX;                       // [2][16 frames][LBS_SXP] f32
    // XCD-aware tile order.  Workgroup b runs on XCD b % 8; XCD x owns the vertex tiles {x, x + 8, ...}, whose posedirs
    // fragments (356 KB each) stay in that XCD's L2, and its workgroups walk (frame tile, vertex tile) side by side, so the
    // transforms and features of a few frame tiles are what else the L2 has to hold.
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
    const int NVX = (NVT - xcd + 7) >> 3;
    const int ntiles = NVX * NFT;
    const float isc = lm.inv_pscale;
    const int q4 = lane >> 4, fl = lane & 15;
    const int sxw = fl * (LBS_SXP * 4) + (wv * WV + 4 * q4) * 12;   // this lane's row / first vertex in the exchange (bytes)
    // Three rotating A-fragment sets (k-steps t .. t+2), two B-fragment sets (t, t+1), two feature-chunk register sets and the
    // three ring slots are addressed by NAME (the loop is unrolled six-fold) so that no register copy ever waits on a load.
    // Step t: barrier (chunk t + 1 is visible, every wave has left chunk t - 1), drop chunk t + 2 (fetched two steps ago) into
    // the slot chunk t - 1 occupied, fetch chunk t + 4 and the posedirs fragments of step t + 2, read the B fragments of step
    // t + 1 from LDS, issue the 24 NVG MFMAs of step t.
    half8 aS[NA][NVG][3], bS[NB][8];
    f32x4 gS[2][NVG];
#define LBS_LD_A(SET, KSTEP) { const int kk_ = min((KSTEP), KS - 1); _Pragma("unroll") for (int vg = 0; vg < NVG; ++vg) _Pragma("unroll") for (int c = 0; c < 3; ++c) \
        aS[SET][vg][c] = (ap + ((size_t)(vg * 3 + c) * KS + kk_) * 64)[lane]; }
#define LBS_LD_G(SET, KSTEP) { const int kk_ = min((KSTEP), KS - 1); _Pragma("unroll") for (int u = 0; u < NVG; ++u) gS[SET][u] = (fp + (size_t)kk_ * 512 + u * TPB)[tid]; }
#define LBS_ST_G(SET, SLOT) { _Pragma("unroll") for (int u = 0; u < NVG; ++u) *reinterpret_cast<f32x4*>(ring + (SLOT) * LBS_CHUNK + (tid + u * TPB) * 16) = gS[SET][u]; }
#define LBS_LD_B(SET, SLOT) { _Pragma("unroll") for (int t = 0; t < 8; ++t) bS[SET][t] = *reinterpret_cast<const half8*>(ring + (SLOT) * LBS_CHUNK + t * 1024 + lane * 16); }
    // The memory instructions of a step are spread BETWEEN its MFMAs (sched_barrier pins the order): a wave issues in order, and
    // a global load does not leave the issue stage while the CU's address unit is busy with the other waves' loads -- with all
    // loads of a step ahead of its MFMAs (the first version), every wave sat ~500 cycles behind the other seven's 1 KB loads
    // before its first MFMA (measured: 1 420 cycles per step against 768 of MFMA; 800 with the loads ablated).
#define LBS_MMA_T(ASET, BSET, T) { _Pragma("unroll") for (int vg = 0; vg < NVG; ++vg) _Pragma("unroll") for (int c = 0; c < 3; ++c) \
        acc[vg][T][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(aS[ASET][vg][c], bS[BSET][T], acc[vg][T][c], 0, 0, 0); __builtin_amdgcn_sched_barrier(0); }
#define LBS_LD_A1(SET, KSTEP, VG) { const int kk_ = min((KSTEP), KS - 1); _Pragma("unroll") for (int c = 0; c < 3; ++c) \
        aS[SET][VG][c] = (ap + ((size_t)((VG) * 3 + c) * KS + kk_) * 64)[lane]; __builtin_amdgcn_sched_barrier(0); }
#define LBS_STEP(S, KSTEP) { const bool ring_ = !(dbg & 8), lda_ = !(dbg & 4); \
        if (ring_) LBS_LDS_BARRIER(); \
        if constexpr (NB == 1) { LBS_LD_B(0, (S) % 3) } \
        __builtin_amdgcn_sched_barrier(0); \
        LBS_MMA_T((S) % NA, (S) % NB, 0) \
        if (ring_) { LBS_ST_G((S) % 2, ((S) + 2) % 3) } __builtin_amdgcn_sched_barrier(0); \
        LBS_MMA_T((S) % NA, (S) % NB, 1) \
        if (ring_ && (KSTEP) + 4 < KS) { LBS_LD_G((S) % 2, (KSTEP) + 4) } __builtin_amdgcn_sched_barrier(0);   /* (nothing is fetched past the last k-step) */ \
        LBS_MMA_T((S) % NA, (S) % NB, 2) \
        if constexpr (NA == 3) { if (lda_ && (KSTEP) + 2 < KS) LBS_LD_A1(((S) + 2) % 3, (KSTEP) + 2, 0) } \
        LBS_MMA_T((S) % NA, (S) % NB, 3) \
        if constexpr (NA == 3 && NVG == 2) { if (lda_ && (KSTEP) + 2 < KS) LBS_LD_A1(((S) + 2) % 3, (KSTEP) + 2, NVG - 1) } \
        LBS_MMA_T((S) % NA, (S) % NB, 4) \
        LBS_MMA_T((S) % NA, (S) % NB, 5) \
        if constexpr (NB == 2) { LBS_LD_B(((S) + 1) % 2, ((S) + 1) % 3) } __builtin_amdgcn_sched_barrier(0); \
        LBS_MMA_T((S) % NA, (S) % NB, 6) \
        LBS_MMA_T((S) % NA, (S) % NB, 7) \
        /* two A sets: the set this step multiplied with is free now -- the fragments of step t + 1 go into it behind the last MFMA */ \
        if constexpr (NA == 2) { if (lda_ && (KSTEP) + 2 < KS) { LBS_LD_A1((S) % 2, (KSTEP) + 2, 0) } } }
    // (MOSHII_LBS_STOP=16: workgroup 0 leaves clock stamps of its phases in the output buffer instead of vertices -- tools/lbs_bench.py prints them)
#define LBS_STAMP(K) { if ((dbg & 16) && blockIdx.x == 0 && tid == 0) reinterpret_cast<long long*>(out)[((idx - slot) / nslots) * 32 + (K)] = clock64(); }
    const int npieces = tlb >> 10;
    auto dma_piece = [&](const char* src, int buf, int p) {   // 1 KiB piece p of a half tile's transforms -> LDS buffer buf, by LDS-DMA
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)p * 1024 + lane * 16),
                                         (__attribute__((address_space(3))) void*)(lds_raw + buf * LBS_TLMAX + p * 1024), 16, 0, 0);
    };
    auto stage_dma = [&](const char* asrc, int h, int buf) { for (int p = wv; p < npieces; p += NWAVE) dma_piece(asrc + (size_t)h * tlb, buf, p); };
    if (slot < ntiles) stage_dma(reinterpret_cast<const char*>(lm.Atr) + (size_t)(slot / NVX) * 8 * tlb, 0, 0);
    for (int idx = slot; idx < ntiles; idx += nslots) {
        const int ft = idx / NVX, vt = xcd + 8 * (idx - ft * NVX);
        const int f0 = ft * LBS_TF, v0 = vt * LBS_TV;
        // ---- main loop: acc[vg][t][c] (16 vertices x 16 frames) += Pfrag(vg, c, ks) x featF(t, ks)
        f32x4 acc[NVG][8][3];
#pragma unroll
        for (int vg = 0; vg < NVG; ++vg)
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int c = 0; c < 3; ++c) acc[vg][t][c] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        // (wave-uniform bases: the loads take them as scalar pairs plus ONE lane-offset register instead of a 64-bit address each)
        const half8* ap = reinterpret_cast<const half8*>(lm.Pfrag) + ((size_t)(vt * 8 + wv * NVG) * 3 * KS) * 64;
        const f32x4* fp = reinterpret_cast<const f32x4*>(lm.featF) + (size_t)ft * KS * 512;
        LBS_STAMP(0)
        // the tile's 128 vertex records {rest position, NWT x (joint address, weight)}: threads 0 .. 127 fetch one each now and hold it
        // across the k-loop (3 + 2 NWT registers in two waves); behind the loop the records go through the (then idle) feature ring,
        // from where every lane picks up the 4 NVG vertices of its accumulator registers.  (Every lane fetching its own vertices
        // here kept 44 registers live across the loop -- spilled; fetching them behind the loop left the first half tile waiting
        // 4 000 cycles for memory.)
        float vrec[3]; int2 jrec[NWT];
        if (tid < LBS_TV) {
#pragma unroll
            for (int c = 0; c < 3; ++c) vrec[c] = lm.vsh_pad[(size_t)(v0 + tid) * 3 + c];
#pragma unroll
            for (int i = 0; i < NWT; ++i) jrec[i] = lm.sjw[(size_t)(v0 + tid) * NWT + i];
        }
        LBS_LD_G(0, 0) LBS_LD_G(1, 1) LBS_LD_A(0, 0) LBS_LD_A(1, 1)
        LBS_ST_G(0, 0) LBS_ST_G(1, 1)
        LBS_LD_G(0, 2) LBS_LD_G(1, 3)
        LBS_LDS_BARRIER();
        if constexpr (NB == 2) LBS_LD_B(0, 0)
        LBS_STAMP(1)
        int ks = 0;
        for (; ks + 6 <= KS; ks += 6) {
            LBS_STEP(0, ks) LBS_STEP(1, ks + 1) LBS_STEP(2, ks + 2) LBS_STEP(3, ks + 3) LBS_STEP(4, ks + 4) LBS_STEP(5, ks + 5)
        }
        if (ks < KS) { LBS_STEP(0, ks) ++ks; }
        if (ks < KS) { LBS_STEP(1, ks) ++ks; }
        if (ks < KS) { LBS_STEP(2, ks) ++ks; }
        if (ks < KS) { LBS_STEP(3, ks) ++ks; }
        if (ks < KS) { LBS_STEP(4, ks) ++ks; }
        // ---- epilogue: eight half tiles of 16 frames.  Per half tile: transforms in LDS (barrier), every lane blends and
        // applies its 4 NVG (vertex, frame) pairs and drops the results into the exchange (barrier), the workgroup writes 16 whole
        // tile rows.  The transforms of half tile h + 1 are on their way (LDS-DMA) while h is worked on.
        LBS_STAMP(2)
        // transforms of this tile / of this workgroup's next tile (whose first half tile is fetched during the last one of this tile:
        // issued behind the k-loop it kept the first half tile waiting 3 000 cycles)
        const char* asrc = reinterpret_cast<const char*>(lm.Atr) + (size_t)ft * 8 * tlb;
        const char* asrc_next = reinterpret_cast<const char*>(lm.Atr) + (size_t)((idx + nslots) / NVX) * 8 * tlb;
        constexpr int RECW = 4 + 2 * NWT;   // dwords per vertex record in LDS: x y z - | weights | joint addresses
        {
            LBS_LDS_BARRIER();   // every wave has read its last B fragments: the ring is free
            LBS_STAMP(28)
            if (tid < LBS_TV) {
                float* rec = reinterpret_cast<float*>(ring) + tid * RECW;
                rec[0] = vrec[0]; rec[1] = vrec[1]; rec[2] = vrec[2];
#pragma unroll
                for (int i = 0; i < NWT; ++i) { rec[4 + i] = __int_as_float(jrec[i].y); reinterpret_cast<int*>(rec)[4 + NWT + i] = jrec[i].x; }
            }
            LBS_LDS_BARRIER();
            LBS_STAMP(29)
        }
        // ---- this lane's vertices: register r of accumulator tile (vg, .) belongs to vertex v0 + WV wv + 16 vg + 4 (lane / 16) + r
        float vs[NVG][4][3], ww[NVG][4][NWT];
        int ja[NVG][4][NWT];
#pragma unroll
        for (int vg = 0; vg < NVG; ++vg)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float* rec = reinterpret_cast<const float*>(ring) + (wv * WV + vg * 16 + 4 * q4 + r) * RECW;
#pragma unroll
                for (int c = 0; c < 3; ++c) vs[vg][r][c] = rec[c];
#pragma unroll
                for (int i = 0; i < NWT; ++i) {
                    ww[vg][r][i] = rec[4 + i];
                    ja[vg][r][i] = reinterpret_cast<const int*>(rec)[4 + NWT + i] + fl * 48;   // byte offset inside a half tile's transform block
                }
            }
        // (MOSHII_LBS_STOP: phase timing by truncation / ablation -- 1: stop after the k-loop, 2: everything but the global stores,
        //  +4: the k-loop re-uses its first posedirs fragments, +8: ... and its first feature chunks, without the ring and its barriers)
        if (dbg & 1) {
            float sacc = 0.0f;
#pragma unroll
            for (int vg = 0; vg < NVG; ++vg)
#pragma unroll
                for (int t = 0; t < 8; ++t)
#pragma unroll
                    for (int c = 0; c < 3; ++c) sacc += acc[vg][t][c][0] + acc[vg][t][c][1] + acc[vg][t][c][2] + acc[vg][t][c][3];
            if (sacc == 123.456f) out[0] = vs[0][0][0] + ww[NVG - 1][3][0] + (float)ja[0][2][1];
            LBS_WAIT_VM(0);
            LBS_LDS_BARRIER();
            continue;
        }
        const int nfl = min(LBS_TV, V - v0) * 3;   // valid floats of a tile row
        // interior tile: every wave issues exactly 2 RPW store instructions per half tile, which is what the counted wait below relies on
        const bool full = (f0 + LBS_TF <= F) && (nfl == LBS_TV * 3);
        // The half tiles run as a LOOP (two per trip: the transform and exchange buffers alternate, and their bases are immediates
        // of the reads): unrolled eight-fold, the epilogue alone was 48 KB of straight-line code executed once per tile.  The
        // accumulators are registers and cannot be indexed by the trip count: a wave-uniform switch moves the half tile's 12 NVG
        // values out first (as rest position + corrective).  ONE barrier per half tile: with two exchange buffers, "transforms of
        // h visible" and "exchange of h - 1 complete" are the same barrier, and the rows of h - 1 are stored while the first
        // transform reads of h are in flight.
        constexpr int NPAIR = NWT / 2, NST = 4 * NVG * NPAIR;   // a stage = two influences of one (vertex, frame) item
        f32x4 A0[2][2], A1[2][2], A2[2][2];
        float pp[NVG][4][3];
#define LBS_GATHER(SET, ST, TOFF) { const int k_ = (ST) / NPAIR, p_ = (ST) % NPAIR; _Pragma("unroll") for (int i = 0; i < 2; ++i) { \
        const char* tp = lds_raw + (TOFF) + ja[k_ >> 2][k_ & 3][2 * p_ + i]; \
        A0[SET][i] = *reinterpret_cast<const f32x4*>(tp); A1[SET][i] = *reinterpret_cast<const f32x4*>(tp + 16); A2[SET][i] = *reinterpret_cast<const f32x4*>(tp + 32); } }
#define LBS_TAKE(H) { _Pragma("unroll") for (int vg = 0; vg < NVG; ++vg) _Pragma("unroll") for (int r = 0; r < 4; ++r) _Pragma("unroll") for (int c = 0; c < 3; ++c) \
        pp[vg][r][c] = fmaf(isc, acc[vg][H][c][r], vs[vg][r][c]); }
        // wave w writes rows w, w + NWAVE, ... of a half tile: 96 16-byte chunks per row = one full wave store + one half-wave store.
        // The stores of half tile h - 1 are issued one at a time between the stages of h: back to back, a wave's second store
        // waits at the issue stage while the address unit works through the other waves' (24 KB per half tile at 64 B/clk).
        auto row_store_part = [&](int h, int soff, int idx) {   // store instruction idx (0 .. 2 RPW - 1) of half tile h's rows
            const int i = idx >> 1, part = idx & 1;
            if (part == 1 && lane >= 32) return;
            const int row = wv + NWAVE * i, f = f0 + 16 * h + row;
            const f32x2* sp = reinterpret_cast<const f32x2*>(Sx + soff + row * (LBS_SXP * 4) + lane * 16 + part * 1024);
            const f32x2 lo = sp[0], hi = sp[1];
            const f32x4u val = {lo.x, lo.y, hi.x, hi.y};
            float* o = out + ((size_t)f * V + v0) * 3 + lane * 4 + part * 256;
            const int c = (lane + 64 * part) * 4;
            if (dbg & 18) return;
            // (streaming stores: the output must not evict the posedirs fragments the k-loop re-reads from L2)
            if (full) __builtin_nontemporal_store(val, reinterpret_cast<f32x4u*>(o));
            else if (f < F) {
                if (c + 4 <= nfl) __builtin_nontemporal_store(val, reinterpret_cast<f32x4u*>(o));
                else for (int e = 0; e < 4; ++e) if (c + e < nfl) o[e] = val[e];
            }
        };
#define LBS_HALF(H, TOFF, SOFF) { \
        /* the transforms of this half tile: its DMA pieces are older than the row stores of half tile H - 2, which stay in flight */ \
        if ((H) >= 2 && full && (dbg & ~16) == 0) LBS_WAIT_VM(2 * RPW); else LBS_WAIT_VM(0); \
        LBS_LDS_BARRIER(); \
        LBS_STAMP(3 + 3 * ((H) & 7)) \
        const char* dsrc_ = ((H) + 1 < 8) ? asrc + (size_t)((H) + 1) * tlb : asrc_next;   /* next half tile's transforms (the next tile's first) */ \
        const bool dgo_ = ((H) + 1 < 8) || (idx + nslots < ntiles); \
        for (int q = NST / 2; dgo_ && q * NWAVE + wv < npieces; ++q) dma_piece(dsrc_, ((H) + 1) & 1, q * NWAVE + wv); \
        switch (H) { case 0: LBS_TAKE(0) break; case 1: LBS_TAKE(1) break; case 2: LBS_TAKE(2) break; case 3: LBS_TAKE(3) break; \
                     case 4: LBS_TAKE(4) break; case 5: LBS_TAKE(5) break; case 6: LBS_TAKE(6) break; default: LBS_TAKE(7) break; } \
        LBS_GATHER(0, 0, TOFF) \
        LBS_STAMP(4 + 3 * ((H) & 7)) \
        float ox = 0.0f, oy = 0.0f, oz = 0.0f; \
        _Pragma("unroll") for (int st = 0; st < NST; ++st) { \
            if (st + 1 < NST) { if (st & 1) LBS_GATHER(0, st + 1, TOFF) else LBS_GATHER(1, st + 1, TOFF) } \
            /* memory instructions one at a time between the stages (the address unit serialises them): first the DMA pieces of the next \
               half tile, then the row stores of the previous one -- all stores younger than all pieces, which the counted wait relies on */ \
            if (st < NST / 2) { if (dgo_ && st * NWAVE + wv < npieces) dma_piece(dsrc_, ((H) + 1) & 1, st * NWAVE + wv); } \
            else if ((H) >= 1 && st - NST / 2 < 2 * RPW) row_store_part((H) - 1, LBS_SXBYTES - (SOFF), st - NST / 2); \
            const int k_ = st / NPAIR, p_ = st % NPAIR, set_ = st & 1; \
            const float px = pp[k_ >> 2][k_ & 3][0], py = pp[k_ >> 2][k_ & 3][1], pz = pp[k_ >> 2][k_ & 3][2]; \
            _Pragma("unroll") for (int i = 0; i < 2; ++i) { const float w = ww[k_ >> 2][k_ & 3][2 * p_ + i]; \
                ox = fmaf(w, fmaf(A0[set_][i].x, px, fmaf(A0[set_][i].y, py, fmaf(A0[set_][i].z, pz, A0[set_][i].w))), ox); \
                oy = fmaf(w, fmaf(A1[set_][i].x, px, fmaf(A1[set_][i].y, py, fmaf(A1[set_][i].z, pz, A1[set_][i].w))), oy); \
                oz = fmaf(w, fmaf(A2[set_][i].x, px, fmaf(A2[set_][i].y, py, fmaf(A2[set_][i].z, pz, A2[set_][i].w))), oz); } \
            if (p_ == NPAIR - 1) { float* so = reinterpret_cast<float*>(Sx + (SOFF) + sxw + ((k_ >> 2) * 16 + (k_ & 3)) * 12); so[0] = ox; so[1] = oy; so[2] = oz; ox = 0.0f; oy = 0.0f; oz = 0.0f; } \
        } \
        LBS_STAMP(5 + 3 * ((H) & 7)) }
#pragma unroll 1
        for (int h2 = 0; h2 < 8; h2 += 2) {
            LBS_HALF(h2, 0, 0)
            LBS_HALF(h2 + 1, LBS_TLMAX, LBS_SXBYTES)
        }
        LBS_LDS_BARRIER();   // the last exchange is complete
#pragma unroll
        for (int q = 0; q < 2 * RPW; ++q) row_store_part(7, LBS_SXBYTES, q);
        LBS_STAMP(27)
    }
#undef LBS_LD_A
#undef LBS_LD_G
#undef LBS_ST_G
#undef LBS_LD_B
#undef LBS_MMA_T
#undef LBS_LD_A1
#undef LBS_STEP
#undef LBS_GATHER
#undef LBS_TAKE
#undef LBS_HALF
#undef LBS_STAMP
}

}  // namespace

static void free_ptr(void* p) { if (p) hipFree(p); }

extern "C" void moshii_lbs32_free(void* l32) {
    Lbs32Model* lm = (Lbs32Model*)l32;
    free_ptr(lm->v_shaped); free_ptr(lm->posedirs_t); free_ptr(lm->weights); free_ptr(lm->J);
    free_ptr(lm->Pfrag); free_ptr(lm->vsh_pad); free_ptr(lm->sjw);
    free_ptr(lm->Atr); free_ptr(lm->featF);
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
        const int KS = (nfeat + 31) / 32;
        const int nvg = Vp128 / 16;
        lm->Vp128 = Vp128; lm->KS = KS;
        lm->KJ = (K + 3) & ~3;   // joints padded so that a half tile's transforms are whole 1 KiB DMA pieces
        // per-vertex influence lists (host; once per model)
        const double* wh = moshii_internal_weights_host(m);
        int NW = 1;
        for (int v = 0; v < V; ++v) {
            int c = 0;
            for (int j = 0; j < K; ++j) c += (wh[(size_t)v * K + j] != 0.0) ? 1 : 0;
            NW = std::max(NW, c);
        }
        bool ok = nfeat > 0 && NW <= LBS_NWMAX;
        lm->K = K;
        const int NWT = (NW <= 4) ? 4 : 8;   // influences padded to the kernel's compile-time width (joint 0, weight 0)
        lm->NW = NWT;
        std::vector<int> sjw(ok ? (size_t)Vp128 * NWT * 2 : 0, 0);   // {byte offset of the joint's [16][12] f32 block, weight bits}
        for (int v = 0; v < V && ok; ++v) {
            int c = 0;
            for (int j = 0; j < K; ++j) {
                const double w = wh[(size_t)v * K + j];
                if (w != 0.0) {
                    const float wf = (float)w;
                    int bits; memcpy(&bits, &wf, 4);
                    sjw[((size_t)v * NWT + c) * 2 + 0] = j * LBS_JBYTES;
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
            lm->mfma_ok = lm->KJ <= LBS_KJMAX;
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
    const bool force_v0 = getenv("MOSHII_LBS_PLAIN") != nullptr;   // the plain f32 kernel (tests compare the two)
    if (!lmp->mfma_ok || force_v0) {
        const Lbs32Model lm = *lmp;
        const size_t lds = (size_t)(md->P + md->K * 30) * sizeof(float);
        hipLaunchKernelGGL(k_lbs_f32_v0, dim3((md->V + 255) / 256, F), dim3(256), lds, stream, *md, lm, pose, trans, verts);
        return hipGetLastError();
    }
    const int Fpad = (F + LBS_TF - 1) / LBS_TF * LBS_TF;
    if (Fpad > lmp->Fcap) {   // per-call scratch grows to the largest F seen (not stream-ordered: sync first)
        hipStreamSynchronize(stream);
        free_ptr(lmp->Atr); free_ptr(lmp->featF);
        lmp->Atr = nullptr; lmp->featF = nullptr; lmp->Fcap = 0;
        const size_t na = (size_t)(Fpad / 16) * lmp->KJ * LBS_JBYTES, nf = (size_t)(Fpad / LBS_TF) * lmp->KS * LBS_CHUNK;
        hipError_t e = hipMalloc((void**)&lmp->Atr, na);
        if (e != hipSuccess) return e;
        e = hipMalloc((void**)&lmp->featF, nf);
        if (e != hipSuccess) return e;
        // frames beyond F, joints beyond K and feature columns beyond 9 (K - 1) are never written again: they stay zero
        e = hipMemset(lmp->Atr, 0, na);
        if (e != hipSuccess) return e;
        e = hipMemset(lmp->featF, 0, nf);
        if (e != hipSuccess) return e;
        lmp->Fcap = Fpad;
    }
    const Lbs32Model lm = *lmp;
    int dbg = 0;
    if (const char* es = getenv("MOSHII_LBS_STOP")) dbg = atoi(es) & 31;   // (development: phase timing by truncation / clock stamps; incomplete output)
    hipLaunchKernelGGL(k_lbs_prep, dim3((F + 3) / 4), dim3(256), (size_t)md->hand_dof * md->nhand_full * sizeof(float), stream, *md, lm.J, F, lm.KS, lm.KJ, pose, trans, lm.Atr, lm.featF,
                       (dbg & 16) ? reinterpret_cast<long long*>(verts) + 8 * 32 : (long long*)nullptr);
    const int NVT = lm.Vp128 / LBS_TV, NFT = Fpad / LBS_TF;
    int ncu = 0, devid = 0;
    hipGetDevice(&devid);
    hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, devid);
    // one workgroup per CU (the kernel takes the whole register file and most of the LDS), 8 XCDs
    const int nslots = std::max(1, std::min((ncu > 0 ? ncu : 256) / 8, ((NVT + 7) / 8) * NFT));
    // waves per workgroup: 8 (two per SIMD, the default) or 4 (MOSHII_LBS_WAVES=4: one per SIMD with twice the registers)
    int nwave = 8;
    if (const char* es = getenv("MOSHII_LBS_WAVES")) nwave = (atoi(es) == 4) ? 4 : 8;
    if (lm.NW != 4) nwave = 4;   // (eight influences per vertex: the register budget of the one-wave-per-SIMD form)
    auto kern = (lm.NW == 4) ? (nwave == 8 ? k_lbs_tile<4, 1> : k_lbs_tile<4, 2>) : k_lbs_tile<8, 2>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(8 * nslots), dim3(nwave * 64), LBS_LDS_BYTES, stream, lm, md->V, F, NVT, NFT, verts, dbg);
    return hipGetLastError();
}
