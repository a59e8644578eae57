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
//                block per 16 frames, from which the export kernel's waves gather their joints -- and the pose features as MFMA B fragments
//                featF[128-frame tile][k-step][16-frame block][lane][8] (f16).
//   k_lbs_export two persistent workgroups per CU (4 waves each), a tile = 64 vertices x 128 frames, a wave = 16 vertices x 128
//                frames = 8 MFMA tiles of 16 x 16 per coordinate (96 accumulator registers); k-loop with the features through a
//                three-slot LDS ring, blend over per-group joint lists with packed FMAs, rows out through an LDS exchange --
//                described at the kernel.  (Rounds 3/4: k_lbs_tile, one workgroup of eight waves per CU, 128 x 128 tiles, a
//                per-vertex four-influence gather from LDS: 216 us per 4000-frame SMPL-H export, 0.18 of the HBM peak.)
//
// Accuracy: f16 operands give |err| ~ 2^-11 |posedirs| |R - I| sqrt(9(K-1)) ~ 5e-6 m for millimetre-scale correctives
// (tests bound it at 2e-5 m); moshii_lbs_forward_f64 is the reference-precision path.
#include "../../include/moshii.h"
#include "moshii_dev.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <cstdlib>
#include <type_traits>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));   // 16-byte, dword-aligned store unit
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

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

// fragment-major f16 posedirs (MFMA A operand of v_mfma_f32_16x16x32_f16): record (group, i, ks) holds, for lane l, the 8 values
// posedirs[v = perm[16 group + (l & 15)]][i][32 ks + 8 (l >> 4) + e]  (perm: the export kernel's vertex groups, -1 = padding)
__global__ void k_pack_pfrag(int nfeat, int KS, int ng, double pscale, const int* __restrict__ perm, const double* __restrict__ src, _Float16* __restrict__ dst) {
    const size_t total = (size_t)ng * 3 * KS * 64 * 8;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int e = (int)(idx & 7);
        const int l = (int)((idx >> 3) & 63);
        size_t r = idx >> 9;
        const int ks = (int)(r % KS); r /= KS;
        const int i = (int)(r % 3);
        const int g = (int)(r / 3);
        const int v = perm[g * 16 + (l & 15)];
        const int q = ks * 32 + (l >> 4) * 8 + e;
        double val = 0.0;
        if (v >= 0 && q < nfeat) val = src[((size_t)v * 3 + i) * nfeat + q] * pscale;
        dst[idx] = (_Float16)val;
    }
}

// rest positions in group order, x pscale: vshs[slot] = {x, y, z, 0} of vertex perm[slot]
__global__ void k_pack_vsh(int n, float pscale, const int* __restrict__ perm, const double* __restrict__ vsh, float* __restrict__ dst) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const int v = perm[s];
    for (int c = 0; c < 3; ++c) dst[s * 4 + c] = v >= 0 ? (float)(vsh[(size_t)v * 3 + c] * (double)pscale) : 0.0f;
    dst[s * 4 + 3] = 0.0f;
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
// One wavefront per frame, lane = joint, four frames per workgroup.  Round 6 form -- everything of a frame lives in the wave's
// registers: the lane's rotation vector straight from the pose row (body joints) or as hands_mean + sum_i pose_hand[i] comps[i][.]
// with the f32 component rows fetched in ONE batch of independent loads (the model keeps an f32 copy: rounds 3-5 converted the
// 2 160 f64 components in every workgroup and ran the product as a per-lane LDS loop -- 62 % of the kernel's 15 us went by before
// Rodrigues started); Rodrigues; then the kinematic chain level by level with the parent's world transform fetched ACROSS LANES
// (ds_bpermute, 12 values a level) instead of through LDS arrays with a wave barrier per level (~770 cycles a level).  LDS holds
// only the four frames' f16 feature rows, from which 16-byte fragment pieces leave.
// per-joint constants of k_lbs_prep, packed so that a lane fetches everything it needs about its joint in ONE round of loads:
//   jtab[j] = 16 dwords: {parent (-1: root), rows n <= 16 of the hand-component window, i0, i1 | J_j | J_j - J_parent (root: J_0) | hands_mean of the
//             joint's three coordinates, is-hand-joint};   hcj[j][u] = {comps[i0 + u][3 columns of joint j], 0}, u < 16, zero beyond the window
__global__ void k_pack_jtab(ModelDev md, const float* __restrict__ Jf, const float* __restrict__ hcompf, const float* __restrict__ hmeanf,
                            float* __restrict__ jtab, float* __restrict__ hcj) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= md.K) return;
    int* ji = reinterpret_cast<int*>(jtab + j * 16);
    float* jf = jtab + j * 16;
    const int p = j > 0 ? md.parents[j] : -1;
    const int c0 = 3 * j, bd = md.body_dof, hd = md.hand_dof, nhf = md.nhand_full;
    int i0 = 0, i1 = 0, hand = 0;
    float hm[3] = {0.0f, 0.0f, 0.0f};
    if (c0 >= bd && hd > 0) {
        const int h = c0 - bd;
        hand = 1;
        i0 = min(min(md.col_lo[h], md.col_lo[h + 1]), md.col_lo[h + 2]);
        i1 = max(max(md.col_hi[h], md.col_hi[h + 1]), md.col_hi[h + 2]);
        if (i1 < i0) i1 = i0;
        for (int c = 0; c < 3; ++c) hm[c] = hmeanf[h + c];
        for (int u = 0; u < 16; ++u)
            for (int c = 0; c < 4; ++c) hcj[(j * 16 + u) * 4 + c] = (c < 3 && i0 + u < i1) ? hcompf[(size_t)(i0 + u) * nhf + h + c] : 0.0f;
    } else {
        for (int u = 0; u < 64; ++u) hcj[j * 64 + u] = 0.0f;
    }
    ji[0] = p; ji[1] = min(i1 - i0, 16); ji[2] = i0; ji[3] = i1;
    for (int i = 0; i < 3; ++i) {
        jf[4 + i] = Jf[j * 3 + i];
        jf[8 + i] = Jf[j * 3 + i] - (p >= 0 ? Jf[p * 3 + i] : 0.0f);
        jf[12 + i] = hm[i];
    }
    jf[7] = 0.0f; jf[11] = 0.0f; ji[15] = hand;
}

// sine in the export's preparation: the hardware instruction on the device (v_sin_f32 on t / 2 pi: absolute error ~1e-6 over the rotation
// angles a pose holds, below the f16 quantisation of the features it feeds; the library routine's exact range reduction costs ~50
// instructions a call in a kernel that 4 000 waves run at once), the library's in the host build of the emulation
#if defined(__HIP_DEVICE_COMPILE__)
#define LBS_SIN(x) __sinf(x)
#define LX_KEEP_F(x) __asm__ __volatile__("" : : "v"(x))
#define LX_KEEP_U(x) __asm__ __volatile__("" : : "v"(x))
#else
#define LBS_SIN(x) sinf(x)
#define LX_KEEP_F(x)
#define LX_KEEP_U(x)
#endif
#ifndef LBS_PREP_WAVES
#define LBS_PREP_WAVES 16      // frames (= waves) per workgroup of k_lbs_prep: a whole 16-frame block
#endif
__global__ __launch_bounds__(64 * LBS_PREP_WAVES, LBS_PREP_WAVES == 16 ? 1 : 4) void k_lbs_prep(ModelDev md, const float* __restrict__ jtab, const float* __restrict__ hcj, const float* __restrict__ hcompf,
                                                   int F, int KS, int KJ,
                                                   const float* __restrict__ pose, const float* __restrict__ trans,
                                                   float* __restrict__ Atr, _Float16* __restrict__ featF, int* __restrict__ varflag, int epoch,
                                                   long long* __restrict__ stamps) {
#define PREP_STAMP(K) { if (stamps != nullptr && blockIdx.x == 0 && threadIdx.x == 0) stamps[K] = clock64(); }
    __shared__ __attribute__((aligned(16))) _Float16 s_feat[LBS_PREP_WAVES][16 * 32];   // the workgroup's frames' feature rows (KS <= 16 k-steps of 32), zero padded
    const int K = md.K, wv = threadIdx.x >> 6, tid = threadIdx.x & 63;
    // Workgroup -> frames: a workgroup is 16 waves = one whole 16-frame block of the outputs (768-byte runs of a joint's transforms, 1 KB
    // feature records), completed by one CU in one L2 and leaving it as whole lines.  (What bounds the kernel is the instruction count:
    // 4 000 waves are four a SIMD, and a SIMD issues one vector instruction per ~4.5 cycles whichever wave it comes from -- an empty
    // kernel of this shape takes 2.2 us, one with 2 000 dependent FMAs a wave 16.4 us, tools/ubench_dispatch.hip; this one, ~700 vector +
    // ~300 other instructions a wave, 18-19 us.  Four or sixteen waves a workgroup: the same.)
#if LBS_PREP_WAVES == 16
    const int fbase = (int)blockIdx.x * 16;
#else
    const int fbase = (((int)blockIdx.x >> 5) * 8 + ((int)blockIdx.x & 7)) * 16 + (((int)blockIdx.x >> 3) & 3) * 4;
#endif
    const int fw = fbase + wv;                   // this wave's frame; spare waves behind the last frame redo frame F - 1 and write nothing
    const int f = __builtin_amdgcn_readfirstlane(min(fw, F - 1));
    PREP_STAMP(0)
    for (int q = tid; q < 16 * 32; q += 64) s_feat[wv][q] = (_Float16)0.0f;
    // (no instruction on the device, where a wave's lanes move together; the CPU emulation runs them one after another and meets here:
    //  the features written below must not be zeroed by a lane that comes later)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const float* ps = pose + (size_t)f * md.NP;
    const int bd = md.body_dof, nhf = md.nhand_full, hd = md.hand_dof;
    const int j = min(tid, K - 1);               // (lanes beyond the joints shadow joint K - 1 and write nothing)
    const bool act = tid < K;
    // ---- ONE round of independent loads: the joint's record, its window of the hand components, the lane's pose variables of this
    //      frame and of frame 0 (a dependent round trip costs 1-2 000 cycles with 4 000 waves starting at once; round 5 made thirty,
    //      this kernel's first form two)
    const f32x4* jt = reinterpret_cast<const f32x4*>(jtab) + j * 4;
    const f32x4 q0 = jt[0], q1 = jt[1], q2 = jt[2], q3 = jt[3];
    const unsigned* psb = reinterpret_cast<const unsigned*>(ps);
    const unsigned* p0b = reinterpret_cast<const unsigned*>(pose);
    const int c0 = 3 * j, cb = min(c0, max(bd - 3, 0));
    unsigned bv[3], b0[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) { bv[i] = psb[cb + i]; b0[i] = p0b[cb + i]; }
    const int hl = bd + min(tid, max(hd - 1, 0));
    const unsigned phb = hd > 0 ? psb[hl] : 0u, ph0 = hd > 0 ? p0b[hl] : 0u;
    // (the elements are copied to scalars first: __builtin_bit_cast applied to a vector ELEMENT expression read element 0 in the host build)
    const float q0x = q0.x, q0z = q0.z, q0w = q0.w, q3w = q3.w;
    const bool hand = __builtin_bit_cast(int, q3w) != 0;
    f32x4 cv[16];
    if (hand) {
#pragma unroll
        for (int u = 0; u < 16; ++u) cv[u] = reinterpret_cast<const f32x4*>(hcj)[j * 16 + u];
    } else {
#pragma unroll
        for (int u = 0; u < 16; ++u) cv[u] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    }
    const int p = __builtin_bit_cast(int, q0x), i0 = __builtin_bit_cast(int, q0z), i1 = __builtin_bit_cast(int, q0w);
    const float Jme[3] = {q1.x, q1.y, q1.z}, Jd[3] = {q2.x, q2.y, q2.z};
    // ---- which joints MOVE in this call: varflag[j] = epoch as soon as one frame's inputs of joint j differ (bitwise) from frame 0's.
    // A joint nobody marks has the same rotation -- the same nine pose features -- in every frame: its correctives are a constant of the
    // call (k_lbs_still; a body-only solve leaves the 30 hand joints of SMPL-H at the hand prior's mean: chmosh.py:626-647 with
    // optimize_fingers off, the reference's default).  The hand joints share one answer: their rotation vectors depend on the
    // hand-pose variables only.
    if (stamps != nullptr) {   // (development: which of the first loads the wave waits for)
        LX_KEEP_F(q0.x); PREP_STAMP(6)
        LX_KEEP_U(bv[0]); LX_KEEP_U(phb); PREP_STAMP(7)
        LX_KEEP_U(b0[0]); LX_KEEP_U(ph0); PREP_STAMP(8)
        LX_KEEP_F(cv[15].x); PREP_STAMP(9)
    }
    bool hands_differ = __ballot(tid < hd && phb != ph0) != 0ull;
    for (int base = 64; base < hd; base += 64) {   // (hand spaces of more than 64 variables: the same, 64 at a time)
        const int i = min(base + tid, hd - 1);
        hands_differ = hands_differ || __ballot(psb[bd + i] != p0b[bd + i]) != 0ull;
    }
    // ---- the lane's rotation vector: the pose variables themselves (body joints), or hands_mean + pose_hand . components over the
    //      joint's window of component rows, the pose variables fetched across lanes (lane i holds hand variable i)
    float rv[3];
    const float phf = __builtin_bit_cast(float, phb);
    {
        float hv[3] = {q3.x, q3.y, q3.z};
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const float pu = __shfl(phf, min(i0 + u, 63));       // (every lane takes part; rows past the window have zero components)
#pragma unroll
            for (int c = 0; c < 3; ++c) hv[c] = fmaf(pu, cv[u][c], hv[c]);
        }
        if (hand) {
            for (int i = max(i0 + 16, 0); i < i1; ++i) {          // (windows of more than 16 rows, hand spaces beyond 64 variables: the rest, plainly)
                const float pu = ps[bd + i];
#pragma unroll
                for (int c = 0; c < 3; ++c) hv[c] = fmaf(pu, hcompf[(size_t)i * nhf + (c0 - bd) + c], hv[c]);
            }
            if (hd > 64) {      // (the shuffle above reached variables 0 .. 63 only: redo the first 16 rows from memory)
#pragma unroll
                for (int c = 0; c < 3; ++c) hv[c] = (c == 0 ? q3.x : c == 1 ? q3.y : q3.z);
                for (int i = i0; i < i1; ++i) {
                    const float pu = ps[bd + i];
#pragma unroll
                    for (int c = 0; c < 3; ++c) hv[c] = fmaf(pu, hcompf[(size_t)i * nhf + (c0 - bd) + c], hv[c]);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) rv[i] = hand ? hv[i] : __builtin_bit_cast(float, bv[i]);
    }
    if (act && (hand ? hands_differ : (bv[0] != b0[0] || bv[1] != b0[1] || bv[2] != b0[2]))) varflag[j] = epoch;
    PREP_STAMP(1)
    // ---- Rodrigues: the local rotation Rl, and R - I (without the cancellation) as f16 features
    float Rl[9];
    {
        const float x = rv[0], y = rv[1], z = rv[2];
        const float t2 = x * x + y * y + z * z;
        float a, b;
        if (t2 < 1e-6f) { a = 1.0f - t2 / 6.0f; b = 0.5f - t2 / 24.0f; }
        else { const float t = sqrtf(t2); a = LBS_SIN(t) / t; const float sh = LBS_SIN(0.5f * t); b = 2.0f * sh * sh / t2; }   // (1 - cos t = 2 sin^2 (t / 2): no cancellation at small angles)
        const float K2[9] = {x * x - t2, x * y, x * z, x * y, y * y - t2, y * z, x * z, y * z, z * z - t2};
        const float Km[9] = {0.0f, -z, y, z, 0.0f, -x, -y, x, 0.0f};
#pragma unroll
        for (int e = 0; e < 9; ++e) {
            const float id = (e == 0 || e == 4 || e == 8) ? 1.0f : 0.0f;
            const float d = a * Km[e] + b * K2[e];
            Rl[e] = id + d;
            if (act && j >= 1) s_feat[wv][(j - 1) * 9 + e] = (_Float16)d;
        }
    }
    PREP_STAMP(2)
    // ---- kinematic chain by pointer jumping: every lane holds the transform (R, t) from its joint's frame to the frame of an ANCHOR
    // ancestor -- at first its parent: (Rl_j, J_j - J_parent); the root: (Rl_0, J_0), no anchor -- and in every round composes it with the
    // anchor's own (fetched across lanes: 12 values + the anchor's anchor) and adopts the anchor's anchor: the distance doubles, so
    // ceil(log2(depth + 1)) rounds (4 for the 10 levels of SMPL-H) replace one round per tree level
    float Rw[9], tw[3];
#pragma unroll
    for (int e = 0; e < 9; ++e) Rw[e] = Rl[e];
#pragma unroll
    for (int i = 0; i < 3; ++i) tw[i] = Jd[i];
    int anchor = act ? p : -1;
    for (int reach = 1; reach <= md.maxdepth; reach *= 2) {
        const int src = max(anchor, 0);
        float pr[9], pt[3];
#pragma unroll
        for (int e = 0; e < 9; ++e) pr[e] = __shfl(Rw[e], src);
#pragma unroll
        for (int i = 0; i < 3; ++i) pt[i] = __shfl(tw[i], src);
        const int pa = __shfl(anchor, src);
        if (anchor >= 0) {
            float Rn[9], tn[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
#pragma unroll
                for (int c = 0; c < 3; ++c) Rn[i * 3 + c] = pr[i * 3 + 0] * Rw[c] + pr[i * 3 + 1] * Rw[3 + c] + pr[i * 3 + 2] * Rw[6 + c];
                tn[i] = pr[i * 3 + 0] * tw[0] + pr[i * 3 + 1] * tw[1] + pr[i * 3 + 2] * tw[2] + pt[i];
            }
#pragma unroll
            for (int e = 0; e < 9; ++e) Rw[e] = Rn[e];
#pragma unroll
            for (int i = 0; i < 3; ++i) tw[i] = tn[i];
            anchor = pa;
        }
    }
    PREP_STAMP(3)
    if (act && fw < F) {   // A_j = [Rw | tw - Rw J_j + trans]  (sum_j w_j = 1 lets the root translation ride in every joint), row major:
                           // three 16-byte pieces (R_i0, R_i1, R_i2, t_i) -- each float is one B operand of the export kernel's blend
        f32x4* o = reinterpret_cast<f32x4*>(Atr + (((size_t)(f >> 4) * KJ + j) * 16 + (f & 15)) * 12);
        const float* tr = trans + (size_t)f * 3;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float r0 = Rw[i * 3 + 0], r1 = Rw[i * 3 + 1], r2 = Rw[i * 3 + 2];
            o[i] = f32x4{r0, r1, r2, tw[i] - (r0 * Jme[0] + r1 * Jme[1] + r2 * Jme[2]) + tr[i]};
        }
    }
    PREP_STAMP(4)
    // B fragments of v_mfma_f32_16x16x32_f16: features 8 g .. 8 g + 7 of frame f are the 16 bytes of record (f / 128, g / 4,
    // (f / 16) % 8), lane (f % 16) + 16 (g % 4).  The workgroup's frames are consecutive lanes: thread (g, frame) writes
    // one 16-byte piece, the frames' threads one contiguous run (the first version wrote every feature as a 2-byte store of its own: 2.4 M write
    // requests per call, the waves a third of their time at the issue stage behind them).
    __syncthreads();
    {
        const int fi = threadIdx.x & (LBS_PREP_WAVES - 1), g = threadIdx.x / LBS_PREP_WAVES, fo = fbase + fi;
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
// k_lbs_export: a workgroup = 4 waves (one per SIMD, <= 256 registers, 80 KB of LDS) so that TWO workgroups share a CU, each with
// its own tile of 64 vertices x 128 frames, and run in antiphase: while one is in its k-loop (matrix pipe: 360 MFMAs per wave) the
// other is in its blend epilogue (vector pipe + LDS).  The two pipes issue side by side from different waves of a SIMD, so a CU's
// tile pair costs max(matrix, vector) instead of their sum -- the round-3/4 kernel (ONE workgroup per CU, all eight waves in the
// same phase) paid the sum: 18 k cycles of k-loop with the vector pipe idle, then 26 k of epilogue with the matrix pipe idle.
//
// Wave w of a tile owns the vertex GROUP w: 16 of the tile's 64 vertices, chosen at model-prepare time (group_tiles below) so that
// a group depends on as few joints as possible -- its JOINT LIST, padded to rounds of LX_JR = 4 joints.  The MFMA accumulators are
// lane = frame, register = vertex (as before), and the blend runs over the group's joint list instead of over each vertex's own
// influences: per 16-frame block and round the wave copies the four joints' transforms (4 x 768 B, gathered by LDS-DMA with
// per-lane source addresses) into a wave-private LDS buffer, every lane reads its frame's 4 x 48 B ONCE for all four of its
// vertices (the round-3 form read 4 vertices x 4 influences x 48 B: the LDS pipe and ~450 instructions per block) and accumulates
// T_v += w_vj A_j as packed FMAs (v_pk_fma_f32: the transforms are stored as the pairs (R00,R10) (R01,R11) (R02,R12) (t0,t1)
// (R20,R21) (R22,t2), the weight is broadcast by op_sel) -- 96 packed FMAs per round, 24 instructions to apply T to the four
// corrected rest positions.  Vertices whose joints are not in a round have weight 0 there.  Nothing in the epilogue is shared
// between waves except the result exchange (un-permutes the group order, whole 768-byte tile rows leave as 16-byte streaming
// stores): ONE workgroup barrier per 16-frame block; the transforms need none (a wave waits for its own DMA with a counted vmcnt).
#define LX_TV 64             // vertices per tile (16 per wave)
#define LX_TF 128            // frames per tile (8 MFMA column tiles of 16)
#ifndef LX_XP
#define LX_XP 194            // dwords per row of the result exchange: 64 vertices x 3 + 2: the 16 frame rows of a two-dword write start
                             // on 16 distinct even banks
#endif
#define LX_JR 4              // joints per blend round
#ifndef LX_STAUX
#define LX_STAUX 2           // cache policy of the row stores: 2 = nt (streaming)
#endif
#ifndef LX_NRMAX
#define LX_NRMAX 6           // rounds per group the kernel's LDS tables hold (24 joints per 16 vertices; else the plain kernel runs)
#endif
#define LX_RING 3            // slots of the feature ring
#define LX_CHUNK 8192        // bytes of one k-step's feature fragments: 8 frame blocks x 64 lanes x 16 B
#define LX_JBYTES 768        // one joint's transforms for 16 frames (16 x 12 floats)
#define LX_OFF_SX (LX_RING * LX_CHUNK)                         // result exchange, two buffers of 16 rows
#define LX_SXBYTES (16 * LX_XP * 4)
#define LX_OFF_W (LX_OFF_SX + 2 * LX_SXBYTES)                  // weights [group][round][slot 16][joint 4] f32 (rounds >= 1 read them here)
#define LX_OFF_J (LX_OFF_W + 4 * LX_NRMAX * 256)               // joint lists [group][round][4] (byte offsets j x 768)
#define LX_LDS_BYTES (LX_OFF_J + 4 * LX_NRMAX * LX_JR * 4)     // 55 936 B: two workgroups per CU

#define LBS_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
// (wave_barrier: no instruction on the device, where a wave's lanes move together; the CPU emulation runs lanes one after another and
//  meets there -- a lane reads LDS bytes that the other lanes of its wave wrote)
#define LBS_WAVE_SYNC() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }

// LX_OPAQUE(x): the compiler may not treat x as loop-invariant behind this point (it would hoist every lane-only address computation
// of the epilogue out of the tile loop and park the results in registers the loops need -- measured: 35 spills ahead of the tile
// loop, reloaded inside the block loop); on the host / in the emulation nothing
#if defined(__HIP_DEVICE_COMPILE__)
#define LX_OPAQUE(x) __asm__ __volatile__("" : "+v"(x))
#define LX_KEEP(x) __asm__ __volatile__("" : : "v"(x))   // x stays in its registers, unused by anything else, up to here
#else
#define LX_OPAQUE(x)
#define LX_KEEP(x)
#endif

// ---- per call: the correctives of the joints that do not move ---------------------------------------------------------------------
// k_lbs_prep has marked the joints whose rotation differs between frames (varflag[j] == epoch).  The pose features of the others are
// the same in every frame, so their correctives  sum_{k >= 32 kseff} posedirs[v][k] feature[k]  are one constant per vertex for the
// whole call: one wavefront per vertex group evaluates them once -- the k-steps behind the last moving joint, on the fragments the
// export kernel would read, against frame 0's features -- and writes  pscale x (rest + still correctives)  in the layout of the rest
// positions (tab_vsc): the start value of the export kernel's accumulators, whose k-loop then ends at kseff.  A body-only Stage-II
// result (the reference's default: optimize_fingers off) keeps the 30 hand joints of SMPL-H still: 9 of 15 k-steps.
__global__ __launch_bounds__(64, 1) void k_lbs_still(Lbs32Model lm, const int* __restrict__ varflag, int epoch, int all_move) {
    const int lane = threadIdx.x, g = blockIdx.x, KS = lm.KS, q4 = lane >> 4, fl = lane & 15;
    int kseff = KS;
    if (!all_move) {
        const bool moves = lane >= 1 && lane < lm.K && varflag[min(lane, lm.K - 1)] == epoch;
        const unsigned long long mv = __ballot(moves);
        const int jl = mv ? 63 - __builtin_clzll(mv) : 0;
        kseff = min(KS, (9 * jl + 31) / 32);
    }
    const __amdgpu_buffer_rsrc_t rs_feat = __builtin_amdgcn_make_buffer_rsrc((void*)lm.featF, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_pf = __builtin_amdgcn_make_buffer_rsrc((void*)lm.Pfrag, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_tab = __builtin_amdgcn_make_buffer_rsrc((void*)lm.tables, 0, 0x7fffffff, 0x00020000);
    f32x4 vs[4], acc[3];
#pragma unroll
    for (int r = 0; r < 4; ++r) vs[r] = (f32x4)__builtin_amdgcn_raw_buffer_load_b128(rs_tab, (unsigned)(4 * q4 + r) * 16u, lm.tab_vshs + (unsigned)g * 256u, 0);
#pragma unroll
    for (int c = 0; c < 3; ++c) acc[c] = f32x4{vs[0][c], vs[1][c], vs[2][c], vs[3][c]};
    const unsigned ap = (unsigned)(g * 3 * KS) * 1024u;
    for (int kc = kseff; kc < KS; kc += 10) {    // (straight-line batches of ten steps -- one round trip for the nine still steps of a body-only
                                                 //  SMPL-H export; past the last step the last one again with a zero B operand)
        half8 ca[10][3], cb[10];
#pragma unroll
        for (int u = 0; u < 10; ++u) {
            const unsigned kk = (unsigned)min(kc + u, KS - 1);
#pragma unroll
            for (int c = 0; c < 3; ++c) ca[u][c] = (half8)__builtin_amdgcn_raw_buffer_load_b128(rs_pf, lane * 16u, ap + (c * (unsigned)KS + kk) * 1024u, 0);
            const u32x4 braw = __builtin_amdgcn_raw_buffer_load_b128(rs_feat, lane * 16u, kk * LX_CHUNK, 0);   // frame tile 0, frame block 0
            const unsigned keep = (kc + u < KS) ? 0xffffffffu : 0u;
            cb[u] = (half8)(braw & u32x4{keep, keep, keep, keep});
        }
#pragma unroll
        for (int u = 0; u < 10; ++u)
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ca[u][c], cb[u], acc[c], 0, 0, 0);
    }
    if (fl == 0) {   // column 0 = frame 0; register r = vertex slot 4 q4 + r
        float* o = reinterpret_cast<float*>(lm.tables + lm.tab_vsc) + ((size_t)g * 16 + 4 * q4) * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) *reinterpret_cast<f32x4*>(o + r * 4) = f32x4{acc[0][r], acc[1][r], acc[2][r], 0.0f};
    }
}

__global__ __launch_bounds__(256, 2) void k_lbs_export(Lbs32Model lm, int V, int F, int NVT, int NFT, float* __restrict__ out, const int* __restrict__ varflag, int epoch,
                                                     long long* __restrict__ dbgbuf, int dbg) {
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int KS = lm.KS, NRM = lm.NRM;
    // k-steps [kseff, KS) hold features of joints that do not move in this call (k_lbs_prep's varflag): the same in every frame -- their
    // correctives are a per-vertex constant of the call, which k_lbs_still has added to the rest positions the accumulators start at
    // (tab_vsc); the k-loop runs the first kseff steps.  (dbg & 8: treat every joint as moving.)
    int kseff = KS;
    if (!(dbg & 8)) {
        const bool moves = lane >= 1 && lane < lm.K && varflag[min(lane, lm.K - 1)] == epoch;
        const unsigned long long mv = __ballot(moves);
        const int jl = mv ? 63 - __builtin_clzll(mv) : 0;          // the last moving joint: its features end at 9 jl
        kseff = __builtin_amdgcn_readfirstlane(min(KS, (9 * jl + 31) / 32));
    }
    const unsigned tlb = (unsigned)lm.KJ * LX_JBYTES;      // bytes of one 16-frame block's transforms
    char* ring = lds_raw;                                  // [LX_RING][8 frame blocks][64 lanes][16 B]
    char* Sx = lds_raw + LX_OFF_SX;                        // [2][16 frames][LX_XP] f32
    // XCD-aware tile order.  Workgroup b runs on XCD b % 8; XCD x owns the vertex tiles {x, x + 8, ...}, whose posedirs fragments
    // stay in that XCD's L2, and its workgroups walk (frame tile, vertex tile) side by side.
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
    // Whole vertex tiles per XCD (SMPL-H: 108 = 13 x 8 + 4, so four XCDs carry 14 x NFT tiles against 13 x NFT and end ~20 us behind).
    // (dbg & 4: the even deal -- the partial round of vertex tiles dealt out as (vertex tile, frame tile) pairs, an eighth to every XCD
    //  in one contiguous run.  Measured twice, rounds 5 and 6, with two different kernels: every XCD then ends together -- and LATER
    //  (199.7 against 190.5 us per call): the XCDs that finish early leave the others a faster memory side for their last tiles.)
    const int nvb = NVT >> 3, nvr = NVT & 7;
    const bool olddeal = (dbg & 4) == 0;
    const int NVX = olddeal ? (NVT - xcd + 7) >> 3 : nvb;
    const int nreg = NVX * NFT;
    const int upairs = olddeal ? 0 : nvr * NFT, ulo = xcd * upairs / 8, uhi = (xcd + 1) * upairs / 8;
    const int ntiles = nreg + (uhi - ulo);
    const float isc = lm.inv_pscale;
    const int q4 = lane >> 4, fl = lane & 15;
    // (MOSHII_LBS_STOP=16: workgroup 0 leaves clock stamps of its phases in dbgbuf -- tools/lbs_bench.py prints them; the output stays complete)
#define LX_STAMP(K) { if (stamping && tid == 0) dbgbuf[stamp_tile * 24 + (K)] = clock64(); }
    const bool stamp_wg = (dbg & 16) && blockIdx.x == 0;
    bool stamping = false;
    int stamp_tile = 0;
    const long long wg_t0 = (dbg & 32) ? wall_clock64() : 0;   // (MOSHII_LBS_STOP=32: every workgroup leaves its start / end time in dbgbuf)
    // No LDS-DMA anywhere in this kernel (rounds 3/4 fetched the transforms with global_load_lds; the first form of this kernel the
    // features as well).  Measured this round: (1) the compiler books a FLAT-encoded LDS load as an access to both memories and turns
    // every wait it inserts while one is in flight into a full drain of both counters -- no read-ahead survives; (2) as BUFFER loads
    // (buffer_load_dwordx4 ... lds) its waits stay counted, but counted waits of my own across LDS loads + register loads + stores
    // returned stale LDS on the device (vmcnt(2) wrong, vmcnt(0) right: the three kinds do not retire in issue order relative to each
    // other); (3) MI355X_MICROARCH.md puts the landing rate of LDS-DMA at ~12 B/clk per CU, a fifth of what plain loads deliver --
    // this kernel wants 20-25.  So everything comes through registers: plain loads whose waits the compiler counts, then ds_write.
    // All hot loads and stores are BUFFER instructions: wave-uniform resource (4 SGPRs) + scalar offset + ONE 32-bit lane offset.  As
    // global_load / global_store the compiler kept a 64-bit address per lane and per stream in registers across the loops (hoisted out
    // of the tile loop as loop invariants: ~30 registers, spilled and reloaded inside the k-steps).
    // Resource words: base, no stride, 2^31 - 1 bytes, raw 32-bit data format.
    const __amdgpu_buffer_rsrc_t rs_atr = __builtin_amdgcn_make_buffer_rsrc((void*)lm.Atr, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_feat = __builtin_amdgcn_make_buffer_rsrc((void*)lm.featF, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_pf = __builtin_amdgcn_make_buffer_rsrc((void*)lm.Pfrag, 0, 0x7fffffff, 0x00020000);
    // (the small per-tile tables as well -- ONE kind of vector load in the kernel -- and all of them behind one resource: the model keeps
    //  them in one allocation, lm.tab_* are the tables' byte offsets in it; a resource is four scalar registers)
    const __amdgpu_buffer_rsrc_t rs_tab = __builtin_amdgcn_make_buffer_rsrc((void*)lm.tables, 0, 0x7fffffff, 0x00020000);
    // The blend T_v = sum_j w_vj A_j runs on v_mfma_f32_16x16x4_f32 (round 6): per 16-frame block and ROUND of four list joints,
    // D[vertex slot 4 q4 + r][frame fl] += W[slot][joint k] . A_k[frame][entry] for each of the 12 entries of the 3 x 4 transforms --
    // twelve matrix instructions where rounds 3-5 issued 96 packed (or 192 plain) vector FMAs per lane set.  The B operand of entry e
    // is ONE float per lane: lane (fl, k = lane >> 4) holds entry e of list joint k for frame fl -- the lane's own 48 bytes of that
    // joint's 16 x 48-byte run in Atr, three 16-byte loads straight from L2 into operand registers (no LDS staging, no wave sync);
    // the A operand is the lane's weight W[slot fl][joint k] (x 1 / pscale for the rotation entries: the accumulators carry
    // pscale x (rest + corrective)).  Accumulator layout = the k-loop's: lane = frame, register r = vertex slot 4 q4 + r.
    f32x4 sg[3];   // the NEXT item's transforms (this lane's joint, this lane's frame), loaded one item ahead
    auto load_item = [&](unsigned block, unsigned off) {   // block: byte offset of the 16-frame block in Atr (wave-uniform); off: lane offset
#pragma unroll
        for (int p = 0; p < 3; ++p) sg[p] = (f32x4)__builtin_amdgcn_raw_buffer_load_b128(rs_atr, off + 16u * p, block, 0);
    };
    const int* ljt = reinterpret_cast<const int*>(lds_raw + LX_OFF_J) + wv * NRM * LX_JR;   // this group's joint list in LDS
    half8 aS[3][3], bS[8];
    f32x4 gS[2][2];
#define LX_LD_A(SET, KSTEP) { const unsigned kk_ = (unsigned)min((KSTEP), kseff - 1); _Pragma("unroll") for (int c = 0; c < 3; ++c) \
        aS[SET][c] = (half8)__builtin_amdgcn_raw_buffer_load_b128(rs_pf, lane * 16u, ap + (c * (unsigned)KS + kk_) * 1024u, 0); __builtin_amdgcn_sched_barrier(0); }
#define LX_LD_G(SET, KSTEP) { const unsigned kk_ = (unsigned)min((KSTEP), kseff - 1); gS[SET][0] = (f32x4)__builtin_amdgcn_raw_buffer_load_b128(rs_feat, tid * 16u, fp + kk_ * LX_CHUNK, 0); \
        gS[SET][1] = (f32x4)__builtin_amdgcn_raw_buffer_load_b128(rs_feat, tid * 16u, fp + kk_ * LX_CHUNK + 4096u, 0); }
#define LX_ST_G(SET, SLOT) { *reinterpret_cast<f32x4*>(ring + (SLOT) * LX_CHUNK + tid * 16) = gS[SET][0]; *reinterpret_cast<f32x4*>(ring + (SLOT) * LX_CHUNK + (tid + 256) * 16) = gS[SET][1]; }
    // (four B-fragment registers: the fragments of frame blocks 0 .. 3 are read behind the barrier, block T + 4 goes into block T's
    //  register as soon as its three MFMAs are issued -- 144 cycles of MFMAs ahead of its own)
#ifdef LX_DBG_B8
#define LX_BI(T) (T)
#else
#define LX_BI(T) ((T) & 3)
#endif
#define LX_LD_B1(SLOT, T) { bS[LX_BI(T)] = *reinterpret_cast<const half8*>(ring + (SLOT) * LX_CHUNK + (T) * 1024 + lane * 16); }
#define LX_LD_B(SLOT) { LX_LD_B1(SLOT, 0) LX_LD_B1(SLOT, 1) LX_LD_B1(SLOT, 2) LX_LD_B1(SLOT, 3) }
#define LX_MMA_T(ASET, T) { _Pragma("unroll") for (int c = 0; c < 3; ++c) acc[T][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(aS[ASET][c], bS[LX_BI(T)], acc[T][c], 0, 0, 0); __builtin_amdgcn_sched_barrier(0); }
    // Step t.  Chunk c lives in ring slot c % 3.  Barrier: chunk t (and t + 1) are visible, every wave has left chunk t - 1; read the eight
    // B fragments of chunk t, drop chunk t + 2 (fetched two steps ago) into the slot chunk t - 1 occupied, fetch chunk t + 4 and the posedirs
    // fragments of step t + 2; the memory instructions sit BETWEEN the step's 24 MFMAs (a load does not leave the issue stage while the
    // address unit is busy with other waves' loads).
#define LX_STEP(S, KSTEP) { \
        LBS_LDS_BARRIER(); \
        LX_LD_B((S) % 3) __builtin_amdgcn_sched_barrier(0); \
        LX_MMA_T((S) % 3, 0) LX_LD_B1((S) % 3, 4) \
        if ((KSTEP) + 2 < kseff) { LX_ST_G((S) % 2, ((S) + 2) % 3) } __builtin_amdgcn_sched_barrier(0); \
        LX_MMA_T((S) % 3, 1) LX_LD_B1((S) % 3, 5) \
        if ((KSTEP) + 4 < kseff) { LX_LD_G((S) % 2, (KSTEP) + 4) } __builtin_amdgcn_sched_barrier(0); \
        LX_MMA_T((S) % 3, 2) LX_LD_B1((S) % 3, 6) \
        if ((KSTEP) + 2 < kseff) LX_LD_A(((S) + 2) % 3, (KSTEP) + 2) \
        LX_MMA_T((S) % 3, 3) LX_LD_B1((S) % 3, 7) __builtin_amdgcn_sched_barrier(0); \
        LX_MMA_T((S) % 3, 4) LX_MMA_T((S) % 3, 5) LX_MMA_T((S) % 3, 6) LX_MMA_T((S) % 3, 7) }
    unsigned off0 = 0;   // this lane's offset in a 16-frame block of transforms for the tile's round 0: its list joint + its frame
    for (int idx = slot; idx < ntiles; idx += nslots) {
        int ft, vt;
        if (idx < nreg) { ft = idx / NVX; vt = xcd + 8 * (idx - ft * NVX); }
        else { const int u = ulo + idx - nreg, e = u / NFT; ft = u - e * NFT; vt = 8 * nvb + e; }
        const int f0 = ft * LX_TF, v0 = vt * LX_TV, gi = vt * 4 + wv;
        if (stamp_wg) { stamp_tile = (idx - slot) / nslots; stamping = stamp_tile < 8; }
        LX_STAMP(0)
        if (stamping && tid == 0) dbgbuf[stamp_tile * 24 + 12] = wall_clock64();   // (100 MHz: the shader clock the stamps ran at)
        const int nr = __builtin_amdgcn_readfirstlane((int)__builtin_amdgcn_raw_buffer_load_b32(rs_tab, 0u, lm.tab_gnr + (unsigned)gi * 4u, 0));
        // ---- the tile's tables: the four groups' weights and joint lists into LDS (every wave is past the previous tile's last block)
        for (int i = tid; i < 64 * NRM; i += 256) {   // (4 groups x NRM rounds x 16 slots, 16 bytes each)
            const f32x4 wrow = (f32x4)__builtin_amdgcn_raw_buffer_load_b128(rs_tab, (unsigned)i * 16u, lm.tab_gw + (unsigned)(vt * 64 * NRM) * 16u, 0);
            *reinterpret_cast<f32x4*>(lds_raw + LX_OFF_W + i * 16) = wrow;
        }
        if (tid < 4 * NRM * LX_JR) reinterpret_cast<int*>(lds_raw + LX_OFF_J)[tid] = (int)__builtin_amdgcn_raw_buffer_load_b32(rs_tab, tid * 4u, lm.tab_gjid + (unsigned)(vt * 4 * NRM * LX_JR) * 4u, 0);
        // this lane's four vertices (slots 4 q4 .. 4 q4 + 3 of the group): exchange columns and scaled rest positions -- the accumulators
        // start AT the rest position (x pscale), so the k-loop delivers rest + corrective in one piece
        f32x4 acc[8][3];
        const unsigned ap = (unsigned)(gi * 3 * KS) * 1024u;      // byte offset of the group's fragments in Pfrag
        const unsigned fp = (unsigned)(ft * KS) * LX_CHUNK;      // byte offset of the frame tile's chunks in featF
        {
            f32x4 vs[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) vs[r] = (f32x4)__builtin_amdgcn_raw_buffer_load_b128(rs_tab, (unsigned)(4 * q4 + r) * 16u, lm.tab_vsc + (unsigned)gi * 256u, 0);
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int c = 0; c < 3; ++c) acc[t][c] = f32x4{vs[0][c], vs[1][c], vs[2][c], vs[3][c]};
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- main loop over the moving joints' k-steps: acc[t][c] (16 vertices x 16 frames) += Pfrag(group, c, ks) x featF(t, ks)
        if (kseff > 0) {
            LX_LD_G(0, 0) LX_LD_G(1, 1) LX_LD_A(0, 0) LX_LD_A(1, 1)
            LX_ST_G(0, 0) LX_ST_G(1, 1)
            LX_LD_G(0, 2) LX_LD_G(1, 3)
        }
        LX_STAMP(1)
        int ks = 0;
        for (; ks + 6 <= kseff; ks += 6) {
            LX_STEP(0, ks) LX_STEP(1, ks + 1) LX_STEP(2, ks + 2) LX_STEP(3, ks + 3) LX_STEP(4, ks + 4) LX_STEP(5, ks + 5)
        }
        if (ks < kseff) { LX_STEP(0, ks) ++ks; }
        if (ks < kseff) { LX_STEP(1, ks) ++ks; }
        if (ks < kseff) { LX_STEP(2, ks) ++ks; }
        if (ks < kseff) { LX_STEP(3, ks) ++ks; }
        if (ks < kseff) { LX_STEP(4, ks) ++ks; }
        LX_STAMP(2)
        // what only the epilogue needs is fetched behind the k-loop (held across it these 19 registers were spilled; the first item of a
        // tile waits for them -- the CU's other workgroup runs meanwhile): the lane's exchange columns, its offsets in a round-0 block
        // of transforms, the transforms of the first item
        const u32x4 xo = __builtin_amdgcn_raw_buffer_load_b128(rs_tab, q4 * 16u, lm.tab_gx + (unsigned)gi * 64u, 0);
        {
            unsigned lane_o = lane;
            LX_OPAQUE(lane_o);
            off0 = __builtin_amdgcn_raw_buffer_load_b32(rs_tab, (lane_o >> 4) * 4u, lm.tab_gjid + (unsigned)(gi * NRM * LX_JR) * 4u, 0) + (lane_o & 15u) * 48u;
        }
        // round 0's weights as the A operand: W[slot fl][joint q4], and the same x 1 / pscale for the rotation entries
        const float w0 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_tab, (unsigned)(fl * 16 + q4 * 4), lm.tab_gw + (unsigned)(gi * NRM) * 256u, 0));
        const float w0s = w0 * isc;
        load_item((unsigned)ft * 8u * tlb, off0);
        if (dbg & 1) {   // (MOSHII_LBS_STOP=1: stop behind the k-loop)
            float sacc = 0.0f;
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int c = 0; c < 3; ++c) sacc += acc[t][c][0] + acc[t][c][1] + acc[t][c][2] + acc[t][c][3];
            if (sacc == 123.456f) out[0] = sacc + (float)xo[0] + sg[0].x + sg[1].y + sg[2].z + w0s;
            LBS_LDS_BARRIER();
            continue;
        }
        // ---- epilogue: eight blocks of 16 frames x (this group's rounds).  The block body is expanded eight times (the accumulators are
        // registers: a rolled loop had to copy a block's 12 out through a branch tree -- ~60 moves per block).
        const unsigned abase = (unsigned)ft * 8u * tlb;   // byte offset of the tile's first 16-frame block in Atr
        const int nfl = min(LX_TV, V - v0) * 3;   // valid floats of a tile row
        const bool full = (f0 + LX_TF <= F) && (nfl == LX_TV * 3) && (dbg & 2) == 0;   // interior tile
        // row stores: a block's 16 rows x 768 B are 768 16-byte pieces, three per thread: piece k = 256 s + tid lies in row k / 48
        // (resource = the tile's first row: lane offsets stay below 2^31 whatever the size of the whole output)
        const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)(out + ((size_t)f0 * V + v0) * 3), 0, 0x7fffffff, 0x00020000);
        const unsigned rowb = (unsigned)V * 12u;   // bytes of an output row
        unsigned rp0, rr0;                         // piece 0 of this thread: row tid / 48, column piece tid % 48
        { unsigned tid_o = tid; LX_OPAQUE(tid_o); rr0 = tid_o / 48u; rp0 = tid_o - rr0 * 48u; }
        f32x4 rv[3];
        auto row_read = [&](int t) {   // the thread's three pieces of block t's rows, out of the exchange
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const unsigned pc = rp0 + 16u * s, w = pc >= 48u ? 1u : 0u, piece = pc - 48u * w, row = rr0 + 5u * s + w;
#if LX_XP % 4 == 0
                rv[s] = *reinterpret_cast<const f32x4*>(Sx + (t & 1) * LX_SXBYTES + row * (LX_XP * 4) + piece * 16);
#else
                const f32x2* sp = reinterpret_cast<const f32x2*>(Sx + (t & 1) * LX_SXBYTES + row * (LX_XP * 4) + piece * 16);
                const f32x2 lo = sp[0], hi = sp[1];
                rv[s] = f32x4{lo.x, lo.y, hi.x, hi.y};
#endif
            }
        };
        // The three stores go out back to back (addresses first); their data registers rv stay untouched until the stores have RETIRED:
        // s_waitcnt vmcnt(0) + LX_KEEP(rv) -- at once (`wait`: a tile's last block), or at the start of the next block, where the wave
        // waits for its transforms anyway.  Measured on the device: with ordinary code behind a buffer_store_dwordx4 -- the compiler
        // re-uses its data registers two wait states later, which is what the ISA asks for -- single floats of the stored pieces
        // arrived wrong in memory (lanes 12 .. 15 of every 16, only in the workgroup that shares its CU's address unit with an older
        // one, a few thousand floats per export): the unit reads a store's data out of the registers later than that when it is backed up.
        // lane offsets of the thread's three pieces in the tile's output rows (block-independent: the block is the scalar offset)
        unsigned voff[3];
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const unsigned pc = rp0 + 16u * s, w = pc >= 48u ? 1u : 0u, piece = pc - 48u * w, row = rr0 + 5u * s + w;
            voff[s] = row * rowb + piece * 16u;
        }
        const int wmode = (dbg & 2) ? 0 : full ? 1 : 2;    // (MOSHII_LBS_STOP=2: no row stores) / interior tile / edge tile
        auto row_write = [&](int t, bool wait) {
            const unsigned soff = (unsigned)(16 * t) * rowb;
            __builtin_amdgcn_sched_barrier(0);
            if (wmode == 1) {
                // interior tile: three unconditional streaming stores (nt: the output must not evict the posedirs fragments the k-loop
                // re-reads from L2).  (Round 5's form decided per store and per debug flag: ~15 scalar branches per block, and a taken
                // branch costs the wave ~30 cycles: the three stores took 480 cycles to issue.)
#pragma unroll
                for (int s = 0; s < 3; ++s) __builtin_amdgcn_raw_buffer_store_b128((u32x4)rv[s], rs_out, voff[s], soff, LX_STAUX);
            } else if (wmode == 2) {
#pragma nounroll
                for (int s = 0; s < 3; ++s) {
                    const unsigned pc = rp0 + 16u * s, w = pc >= 48u ? 1u : 0u, piece = pc - 48u * w, row = rr0 + 5u * s + w;
                    const f32x4 val = s == 0 ? rv[0] : s == 1 ? rv[1] : rv[2];
                    if (f0 + 16 * t + (int)row < F) {
                        if ((int)piece * 4 + 4 <= nfl) __builtin_amdgcn_raw_buffer_store_b128((u32x4)val, rs_out, row * rowb + piece * 16u, soff, 2);
                        else {   // a piece that straddles the end of a partial vertex tile's rows: float by float
                            float* o = out + ((size_t)(f0 + 16 * t + (int)row) * V + v0) * 3 + piece * 4;
                            for (int e = 0; e < 4; ++e) if ((int)piece * 4 + e < nfl) o[e] = val[e];
                        }
                    }
                }
            }
            if (wait) { __builtin_amdgcn_s_waitcnt(0x0F70); LX_KEEP(rv[0]); LX_KEEP(rv[1]); LX_KEEP(rv[2]); }   // vmcnt(0)
            __builtin_amdgcn_sched_barrier(0);
        };
        f32x4 T[3][4];     // T[c][e][r]: entry (c, e) of the blended 3 x 4 transform of vertex slot 4 q4 + r for this lane's frame
        const f32x4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
        // Block t: wait for everything in flight (vmcnt(0): explicit, because the compiler would count its own wait for the loads --
        // vmcnt(N), N = the vector-memory instructions issued behind them -- as if everything retired in issue order, and on this
        // device a store issued behind a load can retire ahead of it): this block's round-0 transforms are in sg, the previous block's
        // row stores have read rv.  Twelve matrix instructions; the next item's loads (a whole block to arrive in); the previous
        // block's rows out of the exchange and on their way; further rounds accumulate into T; apply; one workgroup barrier.
#define LX_BLEND(WS, WP, CIN) { \
        _Pragma("unroll") for (int c = 0; c < 3; ++c) { \
            T[c][0] = __builtin_amdgcn_mfma_f32_16x16x4f32((WS), sg[c].x, (CIN) ? zero4 : T[c][0], 0, 0, 0); \
            T[c][1] = __builtin_amdgcn_mfma_f32_16x16x4f32((WS), sg[c].y, (CIN) ? zero4 : T[c][1], 0, 0, 0); \
            T[c][2] = __builtin_amdgcn_mfma_f32_16x16x4f32((WS), sg[c].z, (CIN) ? zero4 : T[c][2], 0, 0, 0); \
            T[c][3] = __builtin_amdgcn_mfma_f32_16x16x4f32((WP), sg[c].w, (CIN) ? zero4 : T[c][3], 0, 0, 0); \
        } }
        // apply: out_v = T_v . (p_v, 1), p_v = pscale x (rest + corrective) out of the accumulators (1 / pscale rides in the rotation entries'
        // weights); into the exchange at the vertex's column.  (Packed FMAs -- two vertices an instruction -- measured: no faster.)
#define LX_APPLY(TT) { \
        unsigned fl_o = (unsigned)fl; LX_OPAQUE(fl_o);   /* (not a tile-loop invariant to be parked in a register: recomputed per block) */ \
        char* sx = Sx + ((TT) & 1) * LX_SXBYTES + fl_o * (LX_XP * 4); \
        _Pragma("unroll") for (int r = 0; r < 4; ++r) { \
            const float px = acc[TT][0][r], py = acc[TT][1][r], pz = acc[TT][2][r]; \
            float* so = reinterpret_cast<float*>(sx + xo[r]); \
            _Pragma("unroll") for (int c = 0; c < 3; ++c) so[c] = fmaf(T[c][0][r], px, fmaf(T[c][1][r], py, fmaf(T[c][2][r], pz, T[c][3][r]))); \
        } }
#define LX_BLOCK(TT) { \
        constexpr int t = (TT); \
        __builtin_amdgcn_s_waitcnt(0x0F70); \
        __builtin_amdgcn_sched_barrier(0); \
        if (t >= 2) { LX_KEEP(rv[0]); LX_KEEP(rv[1]); LX_KEEP(rv[2]); } \
        LX_BLEND(w0s, w0, true) \
        __builtin_amdgcn_sched_barrier(0); \
        if (nr > 1) load_item(abase + (unsigned)t * tlb, (unsigned)ljt[LX_JR + q4] + (unsigned)fl * 48u); \
        else if (t < 7) load_item(abase + (unsigned)(t + 1) * tlb, off0); \
        __builtin_amdgcn_sched_barrier(0); \
        if (t == 3) LX_STAMP(16) \
        if (t >= 1) { row_read(t - 1); if (t == 3) LX_STAMP(17) row_write(t - 1, false); } \
        __builtin_amdgcn_sched_barrier(0); \
        if (t == 3) LX_STAMP(13) \
        for (int q = 1; q < nr; ++q) { \
            const float wq = *reinterpret_cast<const float*>(lds_raw + LX_OFF_W + ((wv * NRM + q) * 16 + fl) * 16 + q4 * 4); \
            const float wqs = wq * isc; \
            __builtin_amdgcn_s_waitcnt(0x0F70); __builtin_amdgcn_sched_barrier(0); \
            if (t >= 1) { LX_KEEP(rv[0]); LX_KEEP(rv[1]); LX_KEEP(rv[2]); } \
            LX_BLEND(wqs, wq, false) \
            __builtin_amdgcn_sched_barrier(0); \
            if (q + 1 < nr) load_item(abase + (unsigned)t * tlb, (unsigned)ljt[(q + 1) * LX_JR + q4] + (unsigned)fl * 48u); \
            else if (t < 7) load_item(abase + (unsigned)(t + 1) * tlb, off0); \
            __builtin_amdgcn_sched_barrier(0); \
        } \
        if (t == 3) LX_STAMP(14) \
        LX_APPLY(TT) \
        if (t == 3) LX_STAMP(15) \
        LBS_LDS_BARRIER(); \
        LX_STAMP(3 + t) }
        // (A branch-free variant of the block for one-round groups on interior tiles -- the same steps as straight-line code, ~60 instructions
        //  instead of ~75 and a dozen scalar branches -- was measured slower, 170-178 against 151 us per call, with and without spills in
        //  its steady state, with counted or full waits: not kept.  Where in the block the previous block's rows are read and stored --
        //  behind the matrix instructions, behind the apply, or the read ahead of them -- makes no difference: 149-154 us all three.)
        LX_BLOCK(0) LX_BLOCK(1) LX_BLOCK(2) LX_BLOCK(3) LX_BLOCK(4) LX_BLOCK(5) LX_BLOCK(6) LX_BLOCK(7)
#undef LX_APPLY
#undef LX_BLEND
#undef LX_BLOCK
        __builtin_amdgcn_s_waitcnt(0x0F70); LX_KEEP(rv[0]); LX_KEEP(rv[1]); LX_KEEP(rv[2]);
        row_read(7);
        row_write(7, true);
        LX_STAMP(11)
    }
    if ((dbg & 32) && dbgbuf != nullptr && tid == 0 && blockIdx.x < 512) {
        dbgbuf[blockIdx.x * 2] = wg_t0;
        dbgbuf[blockIdx.x * 2 + 1] = wall_clock64();
    }
#undef LX_LD_G
#undef LX_ST_G
#undef LX_LD_A
#undef LX_LD_B
#undef LX_LD_B1
#undef LX_MMA_T
#undef LX_STEP
#undef LX_STAMP
}

}  // namespace

static void free_ptr(void* p) { if (p) hipFree(p); }

extern "C" void moshii_lbs32_free(void* l32) {
    Lbs32Model* lm = (Lbs32Model*)l32;
    free_ptr(lm->v_shaped); free_ptr(lm->posedirs_t); free_ptr(lm->weights); free_ptr(lm->J);
    free_ptr(lm->Pfrag); free_ptr(lm->perm); free_ptr(lm->tables); free_ptr(lm->dbgbuf);
    free_ptr(lm->Atr); free_ptr(lm->featF); free_ptr(lm->hcompf); free_ptr(lm->hmeanf); free_ptr(lm->varflag); free_ptr(lm->jtab); free_ptr(lm->hcj);
    memset(lm, 0, sizeof(*lm));
}

// Partition of every 64-vertex tile into four groups of 16 so that a group's vertices depend on few joints (its joint list is what
// the export kernel's blend walks, four joints per round; the workgroup meets at a barrier per block, so the tile's LARGEST list
// counts first, then the sum).  Start: the tile's vertices sorted by (strongest, second strongest) joint, cut into 16s; then pair
// swaps between groups while they lower (max rounds, sum of rounds, sum of joints).  mask[v] = the joints of vertex v (K <= 64).
// Returns order[tile * 64 + slot] = vertex id (or -1: padding behind V).
static std::vector<int> group_tiles(int V, int K, const double* wh, int NVT) {
    std::vector<int> order((size_t)NVT * LX_TV, -1);
    std::vector<unsigned long long> mask((size_t)NVT * LX_TV, 0ull);
    std::vector<int> key((size_t)NVT * LX_TV, 0);
    for (int v = 0; v < V; ++v) {
        int j1 = 0, j2 = -1;
        double w1 = -1.0, w2 = -1.0;
        unsigned long long m = 0;
        for (int j = 0; j < K; ++j) {
            const double w = std::fabs(wh[(size_t)v * K + j]);
            if (wh[(size_t)v * K + j] != 0.0) m |= 1ull << j;
            if (w > w1) { w2 = w1; j2 = j1; w1 = w; j1 = j; }
            else if (w > w2) { w2 = w; j2 = j; }
        }
        if (j2 < 0 || w2 <= 0.0) j2 = j1;
        mask[v] = m;
        key[v] = j1 * 64 + j2;
    }
    auto pc = [](unsigned long long m) { return __builtin_popcountll(m); };
    for (int t = 0; t < NVT; ++t) {
        int ids[LX_TV];
        for (int s = 0; s < LX_TV; ++s) ids[s] = t * LX_TV + s;
        std::stable_sort(ids, ids + LX_TV, [&](int a, int b) {
            const bool pa = a >= V, pb = b >= V;     // padding last
            if (pa != pb) return pb;
            return key[a] < key[b];
        });
        unsigned long long gm[4];
        auto umask = [&](int g) { unsigned long long m = 0; for (int s = 0; s < 16; ++s) m |= mask[ids[g * 16 + s]]; return m; };
        auto cost = [&](const unsigned long long* ms) {
            long mx = 0, sr = 0, sj = 0;
            for (int g = 0; g < 4; ++g) { const long n = pc(ms[g]), r = std::max(1L, (n + LX_JR - 1) / LX_JR); mx = std::max(mx, r); sr += r; sj += n; }
            return mx * 1000000L + sr * 1000L + sj;
        };
        for (int g = 0; g < 4; ++g) gm[g] = umask(g);
        long cur = cost(gm);
        for (int sweep = 0; sweep < 4; ++sweep) {
            bool improved = false;
            for (int a = 0; a < 4; ++a)
                for (int b = a + 1; b < 4; ++b)
                    for (int ia = 0; ia < 16; ++ia)
                        for (int ib = 0; ib < 16; ++ib) {
                            std::swap(ids[a * 16 + ia], ids[b * 16 + ib]);
                            unsigned long long ms[4] = {gm[0], gm[1], gm[2], gm[3]};
                            ms[a] = umask(a); ms[b] = umask(b);
                            const long c = cost(ms);
                            if (c < cur) { cur = c; gm[a] = ms[a]; gm[b] = ms[b]; improved = true; }
                            else std::swap(ids[a * 16 + ia], ids[b * 16 + ib]);
                        }
            if (!improved) break;
        }
        for (int s = 0; s < LX_TV; ++s) order[(size_t)t * LX_TV + s] = ids[s] < V ? ids[s] : -1;
    }
    return order;
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
        const int NVT = (V + LX_TV - 1) / LX_TV, NG = NVT * 4;
        const int KS = (nfeat + 31) / 32;
        lm->NVT = NVT; lm->KS = KS; lm->K = K;
        lm->KJ = (K + 3) & ~3;
        const double* wh = moshii_internal_weights_host(m);
        bool ok = nfeat > 0 && K <= 64;
        std::vector<int> order, gnr, gjid, gx;
        std::vector<float> gw;
        int NRM = 1;
        if (ok) {
            order = group_tiles(V, K, wh, NVT);
            std::vector<std::vector<int>> lists(NG);
            for (int g = 0; g < NG; ++g) {
                unsigned long long mk = 0;
                for (int s = 0; s < 16; ++s) {
                    const int v = order[(size_t)g * 16 + s];
                    if (v >= 0) for (int j = 0; j < K; ++j) if (wh[(size_t)v * K + j] != 0.0) mk |= 1ull << j;
                }
                for (int j = 0; j < K; ++j) if (mk >> j & 1) lists[g].push_back(j);
                NRM = std::max(NRM, ((int)lists[g].size() + LX_JR - 1) / LX_JR);
            }
            ok = NRM <= LX_NRMAX;
            if (ok) {
                gnr.assign(NG, 1); gjid.assign((size_t)NG * NRM * LX_JR, 0); gx.assign((size_t)NG * 16, 0);
                gw.assign((size_t)NG * NRM * 16 * LX_JR, 0.0f);
                for (int g = 0; g < NG; ++g) {
                    const int nj = (int)lists[g].size();
                    gnr[g] = std::max(1, (nj + LX_JR - 1) / LX_JR);
                    for (int i = 0; i < nj; ++i) gjid[(size_t)g * NRM * LX_JR + i] = lists[g][i] * LX_JBYTES;
                    for (int s = 0; s < 16; ++s) {
                        const int v = order[(size_t)g * 16 + s];
                        // padding keeps its own column of the tile (beyond the valid floats of a row: written to the exchange, never stored)
                        int col = v >= 0 ? v % LX_TV : -1;
                        if (col < 0) {   // the free columns of this tile, in slot order
                            const int tile = g / 4, nvalid = std::min(LX_TV, V - tile * LX_TV);
                            int npad_before = 0;
                            for (int s2 = 0; s2 < (g % 4) * 16 + s; ++s2) if (order[(size_t)tile * LX_TV + s2] < 0) ++npad_before;
                            col = nvalid + npad_before;
                        }
                        gx[(size_t)g * 16 + s] = col * 12;
                        if (v >= 0) for (int i = 0; i < nj; ++i)
                            gw[(((size_t)g * NRM + i / LX_JR) * 16 + s) * LX_JR + i % LX_JR] = (float)wh[(size_t)v * K + lists[g][i]];
                    }
                }
            }
        }
        lm->NRM = NRM;
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
            lm->pscale = (float)pscale;
            lm->inv_pscale = (float)(1.0 / pscale);
            if (hipMalloc((void**)&lm->Pfrag, (size_t)NG * 3 * KS * 64 * 8 * sizeof(_Float16)) != hipSuccess) return MOSHII_ERR_HIP;
            if (hipMalloc((void**)&lm->perm, order.size() * sizeof(int)) != hipSuccess) return MOSHII_ERR_HIP;
            // one allocation for the per-group tables (16-byte aligned parts): rounds | joint lists | exchange columns | rest positions | weights
            auto al16 = [](size_t n) { return (n + 15) & ~(size_t)15; };
            lm->tab_gnr = 0;
            lm->tab_gjid = (unsigned)al16(gnr.size() * sizeof(int));
            lm->tab_gx = lm->tab_gjid + (unsigned)al16(gjid.size() * sizeof(int));
            lm->tab_vshs = lm->tab_gx + (unsigned)al16(gx.size() * sizeof(int));
            lm->tab_gw = lm->tab_vshs + (unsigned)((size_t)NG * 16 * 4 * sizeof(float));
            lm->tab_vsc = lm->tab_gw + (unsigned)al16(gw.size() * sizeof(float));     // per call: rest + still correctives (k_lbs_still)
            const size_t tab_bytes = lm->tab_vsc + (size_t)NG * 16 * 4 * sizeof(float);
            if (hipMalloc((void**)&lm->tables, tab_bytes) != hipSuccess) return MOSHII_ERR_HIP;
            hipMemset(lm->tables, 0, tab_bytes);
            if (hipMalloc((void**)&lm->dbgbuf, 1024 * sizeof(long long)) != hipSuccess) return MOSHII_ERR_HIP;
            hipMemset(lm->dbgbuf, 0, 1024 * sizeof(long long));
            hipMemcpy(lm->perm, order.data(), order.size() * sizeof(int), hipMemcpyHostToDevice);
            hipMemcpy(lm->tables + lm->tab_gx, gx.data(), gx.size() * sizeof(int), hipMemcpyHostToDevice);
            hipMemcpy(lm->tables + lm->tab_gnr, gnr.data(), gnr.size() * sizeof(int), hipMemcpyHostToDevice);
            hipMemcpy(lm->tables + lm->tab_gjid, gjid.data(), gjid.size() * sizeof(int), hipMemcpyHostToDevice);
            hipMemcpy(lm->tables + lm->tab_gw, gw.data(), gw.size() * sizeof(float), hipMemcpyHostToDevice);
            hipLaunchKernelGGL(k_pack_pfrag, dim3(4096), dim3(256), 0, 0, nfeat, KS, NG, pscale, lm->perm, moshii_internal_posedirs(m), lm->Pfrag);
            lm->mfma_ok = 1;
        }
    }
    hipLaunchKernelGGL(k_cvt_vsh, dim3((V * 3 + 255) / 256), dim3(256), 0, 0, V * 3, moshii_internal_vsh(m), lm->v_shaped);
    hipLaunchKernelGGL(k_cvt_vsh, dim3(1), dim3(256), 0, 0, K * 3, moshii_internal_J(m), lm->J);
    lm->jtab_valid = 0;
    if (lm->mfma_ok)   // rest positions in group order, x pscale (the accumulators start there)
        hipLaunchKernelGGL(k_pack_vsh, dim3((lm->NVT * LX_TV + 255) / 256), dim3(256), 0, 0, lm->NVT * LX_TV, lm->pscale, lm->perm, moshii_internal_vsh(m), reinterpret_cast<float*>(lm->tables + lm->tab_vshs));
    if (hipDeviceSynchronize() != hipSuccess) return MOSHII_ERR_HIP;
    moshii_internal_l32_set_valid(m, 1);
    return MOSHII_OK;
}

extern "C" int moshii_internal_lbs_debug_times(void* lbs32, long long* out512x2) {   // (development: MOSHII_LBS_STOP=32)
    Lbs32Model* lm = (Lbs32Model*)lbs32;
    if (!lm->dbgbuf) return -1;
    hipDeviceSynchronize();
    return hipMemcpy(out512x2, lm->dbgbuf, 512 * 2 * sizeof(long long), hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
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
    // The export kernel addresses the per-call scratch through buffer resources (2^31 - 1 bytes, 32-bit offsets; a load beyond the range
    // returns zeros, silently): a call that would pass that is cut into sub-calls of whole frame tiles (SMPL-H: 860 000 frames a piece).
    {
        const long long per16 = (long long)lmp->KJ * LX_JBYTES, per128 = (long long)lmp->KS * LX_CHUNK;
        long long fmax = std::min(((0x7fffffffLL - 4096) / per16) * 16, ((0x7fffffffLL - 4096) / per128) * LX_TF);
        if (const char* es = getenv("MOSHII_LBS_FMAX")) fmax = std::min(fmax, (long long)std::max(LX_TF, atoi(es)));   // (tests: a small limit)
        fmax = fmax / LX_TF * LX_TF;
        if ((long long)F > fmax) {
            for (long long f0 = 0; f0 < F; f0 += fmax) {
                const int n = (int)std::min<long long>(fmax, F - f0);
                hipError_t e = moshii_launch_lbs_f32(stream, md, n, pose + (size_t)f0 * md->NP, trans + (size_t)f0 * 3, verts + (size_t)f0 * md->V * 3, lbs32);
                if (e != hipSuccess) return e;
            }
            return hipSuccess;
        }
    }
    const int Fpad = (F + LX_TF - 1) / LX_TF * LX_TF;
    if (Fpad > lmp->Fcap) {   // per-call scratch grows to the largest F seen (not stream-ordered: sync first)
        hipStreamSynchronize(stream);
        free_ptr(lmp->Atr); free_ptr(lmp->featF);
        lmp->Atr = nullptr; lmp->featF = nullptr; lmp->Fcap = 0;
        const size_t na = (size_t)(Fpad / 16) * lmp->KJ * LX_JBYTES, nf = (size_t)(Fpad / LX_TF) * lmp->KS * LX_CHUNK;
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
    if (!lmp->hmeanf) {   // f32 copies of the hand-pose map (stream-ordered: ahead of the first k_lbs_prep that reads them)
        const int nh = std::max(md->nhand_full, 1), nc = std::max(md->hand_dof * md->nhand_full, 1);
        hipError_t e = hipMalloc((void**)&lmp->hmeanf, (size_t)nh * sizeof(float));
        if (e != hipSuccess) return e;
        e = hipMalloc((void**)&lmp->hcompf, (size_t)nc * sizeof(float));
        if (e != hipSuccess) return e;
        if (md->hand_dof > 0) {
            hipLaunchKernelGGL(k_cvt_vsh, dim3((nh + 255) / 256), dim3(256), 0, stream, md->nhand_full, (const double*)md->hands_mean, lmp->hmeanf);
            hipLaunchKernelGGL(k_cvt_vsh, dim3((nc + 255) / 256), dim3(256), 0, stream, md->hand_dof * md->nhand_full, (const double*)md->comps, lmp->hcompf);
        }
    }
    if (!lmp->varflag) {   // k_lbs_prep marks the joints that move in a call with the call's number: never reset, never equal to an older call's
        hipError_t e = hipMalloc((void**)&lmp->varflag, 64 * sizeof(int));
        if (e != hipSuccess) return e;
        e = hipMemsetAsync(lmp->varflag, 0, 64 * sizeof(int), stream);
        if (e != hipSuccess) return e;
        lmp->epoch = 0;
    }
    lmp->epoch = lmp->epoch >= 0x7ffffff0 ? 1 : lmp->epoch + 1;
    if (!lmp->jtab) {
        hipError_t e = hipMalloc((void**)&lmp->jtab, 64 * 16 * sizeof(float));
        if (e != hipSuccess) return e;
        e = hipMalloc((void**)&lmp->hcj, 64 * 64 * sizeof(float));
        if (e != hipSuccess) return e;
        lmp->jtab_valid = 0;
    }
    if (!lmp->jtab_valid) {   // (the joints move with the betas: moshii_lbs32_prepare clears the mark)
        hipLaunchKernelGGL(k_pack_jtab, dim3(1), dim3(64), 0, stream, *md, lmp->J, lmp->hcompf, lmp->hmeanf, lmp->jtab, lmp->hcj);
        lmp->jtab_valid = 1;
    }
    const Lbs32Model lm = *lmp;
    int dbg = 0;
    if (const char* es = getenv("MOSHII_LBS_STOP")) dbg = atoi(es) & 127;   // (development: phase timing by truncation / clock stamps; incomplete output)
    hipLaunchKernelGGL(k_lbs_prep, dim3(LBS_PREP_WAVES == 16 ? (F + 15) / 16 : ((F + 127) / 128) * 32), dim3(64 * LBS_PREP_WAVES), 0, stream, *md, lm.jtab, lm.hcj, lm.hcompf, F, lm.KS, lm.KJ, pose, trans, lm.Atr, lm.featF, lm.varflag, lm.epoch,
                       (dbg & 16) ? lm.dbgbuf + 8 * 24 : (long long*)nullptr);
    hipLaunchKernelGGL(k_lbs_still, dim3(lm.NVT * 4), dim3(64), 0, stream, lm, lm.varflag, lm.epoch, (dbg & 8) ? 1 : 0);
    const int NVT = lm.NVT, NFT = Fpad / LX_TF;
    int ncu = 0, devid = 0;
    hipGetDevice(&devid);
    hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, devid);
    // two workgroups per CU (a workgroup takes half a CU's registers and LDS), 8 XCDs
    int nslots = std::max(1, std::min(2 * (ncu > 0 ? ncu : 256) / 8, ((NVT + 7) / 8) * NFT));
    if (const char* es = getenv("MOSHII_LBS_SLOTS")) nslots = std::max(1, std::min(nslots, atoi(es)));   // (development: fewer workgroups per XCD)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_lbs_export), hipFuncAttributeMaxDynamicSharedMemorySize, LX_LDS_BYTES);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_lbs_export, dim3(8 * nslots), dim3(256), LX_LDS_BYTES, stream, lm, md->V, F, NVT, NFT, verts, lm.varflag, lm.epoch, lm.dbgbuf, dbg);
    return hipGetLastError();
}
