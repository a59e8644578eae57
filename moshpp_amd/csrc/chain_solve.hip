// k_chain_solve: the MoSh++ Stage-II frame loop as ONE persistent HIP kernel for gfx950.
//
// Replaces chmosh.mosh_stageii's hot loop (src/moshpp/chmosh.py:584-724) together with everything it
// drives through chumpy: SmplModelLBS forward (models/smpl_fast_derivatives.py:185-244), its pose
// Jacobian (:246-258), TransformedLms (transformed_lm.py:130-162), the max-mixture prior
// (prior/gmm_prior_ch.py:53-85), the rigid first-frame init (rigid_transformations.py:39-83) and
// chumpy's minimize_dogleg (normal equations, solve, trust-region control).
//
// Mapping: one chain (sequence or chunk) per 256-thread workgroup = 4 waves, one per SIMD of a CU.
// All solver state is float64 and lives in LDS / registers; per evaluation the only L2 traffic is the
// free joints' share of the compact posedirs slice of the <= 3M attached vertices (vertex index fastest
// => coalesced), per frame M*3 observations in and one result row out.  J^T J is accumulated in
// registers as 16x16-thread outer-product tiles (lower triangle only) and eliminated (L D L^T) in
// registers with one LDS-published column and one barrier per step.  DESIGN.md section 4 has the
// per-phase timings and the rules this file follows (branch-free LDS traffic, loads batched ahead of use).
#include "moshii_dev.h"
#include <utility>
#include <type_traits>
#if defined(__HIP_DEVICE_COMPILE__)
#define MOSHII_OPAQUE(x) __asm__ __volatile__("" : "+v"(x))
#else
#define MOSHII_OPAQUE(x) do {} while (0)
#endif
__device__ __forceinline__ double uni64(double v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
#else
    return v;
#endif
}

namespace moshii {

typedef double v2d __attribute__((vector_size(16)));   // one 16-byte load

#ifdef MOSHII_PROFILE
__device__ long long g_prof[64];
// cooperative chains: wall-clock stamps (100 MHz, one clock for the whole device) of every exchange, per rank of chain 0:
// [rank][exchange][0 arrive, 1 posted, 2 everybody seen, 3 payload read]
#define MOSHII_TRACE_N 8192
__device__ long long g_trace[8][MOSHII_TRACE_N][4];
#define TRACE_STAMP(co, seq, ev) do { if (threadIdx.x == 0 && blockIdx.x < 8u * (co).G && (blockIdx.x & 7u) == 0u && (seq) < MOSHII_TRACE_N) g_trace[(co).rank][(seq)][(ev)] = wall_clock64(); } while (0)
#define PROF_BEGIN() long long _pt = clock64()
// (block 0 only: the one chain of a profiled run -- rank 0 of a cooperative chain)
#define PROF_LAP(slot) do { __syncthreads(); if (threadIdx.x == 0 && blockIdx.x == 0) { long long _n = clock64(); g_prof[slot] += _n - _pt; _pt = _n; } else { _pt = 0; } } while (0)
#define PROF_COUNT(slot) do { if (threadIdx.x == 0 && blockIdx.x == 0) g_prof[slot] += 1; } while (0)
// (laps inside helper functions: the running time stamp lives in g_prof[63]; PROF_MARK starts it -- one workgroup only)
#define PROF_MARK() do { if (threadIdx.x == 0 && blockIdx.x == 0) g_prof[63] = clock64(); } while (0)
#define PROF_LAP_EXT(slot) do { __syncthreads(); if (threadIdx.x == 0 && blockIdx.x == 0) { long long _n = clock64(); g_prof[slot] += _n - g_prof[63]; g_prof[63] = _n; } } while (0)
#define PROF_T(var) const long long var = clock64()
#define PROF_ACC(slot, var) do { if (threadIdx.x == 0 && blockIdx.x == 0) g_prof[slot] += clock64() - var; } while (0)   // thread 0's time since PROF_T
#else
#define PROF_BEGIN() do {} while (0)
#define PROF_LAP(slot) do {} while (0)
#define PROF_COUNT(slot) do {} while (0)
#define PROF_MARK() do {} while (0)
#define PROF_LAP_EXT(slot) do {} while (0)
#define PROF_T(var) do {} while (0)
#define PROF_ACC(slot, var) do {} while (0)
#define TRACE_STAMP(co, seq, ev) do {} while (0)
#endif

enum { S_KBEST = 0, S_PRIOR_SS = 1, S_FAIL = 2, S_TMP0 = 3, S_TMP1 = 4, S_TMP2 = 5, S_TMP3 = 6, S_PRIOR_REF = 7, S_PRIOR_KB0 = 8, S_BATON = 9, S_ABORT = 10,
       S_V0 = 11, S_V1 = 12,              // cooperative variant: this rank's share [v0, v1) of the frame's visible-marker list
       S_COOP_SEQ = 13, S_COOP_FAIL = 14  // ... number of exchanges completed; the group is broken (a rank did not show up)
     };

struct Ctx {
    double *pose, *trans, *pose_t, *trans_t, *pose_prev, *vtarget, *fullpose;
    double *feat, *B, *omega, *Rw, *tw, *Rloc, *acol, *Jl;
    double *vposed, *vpos, *msim, *res, *vconst, *vshp, *shp0;
    double *xb, *ell, *score, *px0, *ps0;
    double *g, *dsd, *dgn, *ddl, *y;
    double *red, *scal;
    unsigned long long* anc;
    int *visidx, *colpid, *colprior, *pid2prior, *jointslot, *kfree, *colq, *ksum, *kconst;
    double *big, *Jh, *Jrow, *Lm, *Trot, *xjs, *rest;
    int* tjs;
};

struct FrameParams {
    const double* obs;      // [M][3] of this frame
    double wt_data, wt_pose, wt_poseH, wt_velo;
    int has_velo, use_fingers, nobs;
    // extended variant (XT): jaw term, free shape block and its "stay" term are live in this phase
    double wt_poseF;
    int use_face, use_shape, has_stay;
    int v0, v1;             // the entries [v0, v1) of the visible-marker list whose Jacobian rows are built here (plain chain: all of them)
};

struct Sse { double data, prior, velo, hand, total, face, shape, stay; };

// Cross-lane moves on the DPP path (a v_mov per 32-bit half: no LDS crossbar round trip as with ds_bpermute).
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_f64(double v, int lane /* wave-uniform */) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
enum { DPP_QUAD_1032 = 0xB1, DPP_QUAD_2301 = 0x4E, DPP_ROW_HALF_MIRROR = 0x141, DPP_ROW_MIRROR = 0x140 };
// 64-bit DPP with row_newbcast:K (gfx90a on: the only lane control the double-precision ALU takes): every lane reads lane K of its own row
// of 16.  rowbcast_f64<K>(v) = that value; fmac_rowbcast<K>(acc, m, a): acc = fma(-m[lane K of the row], a, acc) in ONE instruction --
// what two v_readlane + an fma do when the broadcast value sits in every row (ldl_panel_eliminate_rows).  The source of a DPP read must
// not have been written by the two instructions before it (hazard the compiler handles for its own DPP, not inside asm).
#if defined(__HIP_DEVICE_COMPILE__)
// (the wait states ride INSIDE the statement that reads through DPP where the value was just produced -- rowbcast_f64, the first fma of a
//  column -- so that nothing the register allocator might place between the two, a copy of the value included, can undo them)
template <int K> __device__ __forceinline__ double rowbcast_f64(double v) {
    double r;
    asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "n"(K));
    return r;
}
template <int K, bool SETTLE = false> __device__ __forceinline__ void fmac_rowbcast(double& acc, double m, double a) {
    if constexpr (SETTLE) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(m), "v"(a), "n"(K));
    else asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(m), "v"(a), "n"(K));
}
#else
template <int K> __device__ __forceinline__ double rowbcast_f64(double v) { return __shfl(v, K, 16); }
template <int K, bool SETTLE = false> __device__ __forceinline__ void fmac_rowbcast(double& acc, double m, double a) { acc = fma(-__shfl(m, K, 16), a, acc); }
#endif

// Sum over the wavefront, the bitwise-identical total in every lane: an exchange butterfly inside each row of 16 lanes
// (pairs, quads, halves, row -- both partners add the same two numbers, so all 16 lanes agree), then the four row sums.
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_f64<DPP_QUAD_1032>(v);
    v += dpp_f64<DPP_QUAD_2301>(v);
    v += dpp_f64<DPP_ROW_HALF_MIRROR>(v);
    v += dpp_f64<DPP_ROW_MIRROR>(v);
    return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}
__device__ __forceinline__ void wave_sum3(double& a, double& b, double& c) {   // three independent reductions, interleaved
    a += dpp_f64<DPP_QUAD_1032>(a); b += dpp_f64<DPP_QUAD_1032>(b); c += dpp_f64<DPP_QUAD_1032>(c);
    a += dpp_f64<DPP_QUAD_2301>(a); b += dpp_f64<DPP_QUAD_2301>(b); c += dpp_f64<DPP_QUAD_2301>(c);
    a += dpp_f64<DPP_ROW_HALF_MIRROR>(a); b += dpp_f64<DPP_ROW_HALF_MIRROR>(b); c += dpp_f64<DPP_ROW_HALF_MIRROR>(c);
    a += dpp_f64<DPP_ROW_MIRROR>(a); b += dpp_f64<DPP_ROW_MIRROR>(b); c += dpp_f64<DPP_ROW_MIRROR>(c);
    a = (readlane_f64(a, 0) + readlane_f64(a, 16)) + (readlane_f64(a, 32) + readlane_f64(a, 48));
    b = (readlane_f64(b, 0) + readlane_f64(b, 16)) + (readlane_f64(b, 32) + readlane_f64(b, 48));
    c = (readlane_f64(c, 0) + readlane_f64(c, 16)) + (readlane_f64(c, 32) + readlane_f64(c, 48));
}
__device__ __forceinline__ double wave_max(double v) {
    v = fmax(v, dpp_f64<DPP_QUAD_1032>(v));
    v = fmax(v, dpp_f64<DPP_QUAD_2301>(v));
    v = fmax(v, dpp_f64<DPP_ROW_HALF_MIRROR>(v));
    v = fmax(v, dpp_f64<DPP_ROW_MIRROR>(v));
    return fmax(fmax(readlane_f64(v, 0), readlane_f64(v, 16)), fmax(readlane_f64(v, 32), readlane_f64(v, 48)));
}

// Sum over the 256-thread block; every thread returns the bitwise-identical total.
__device__ __forceinline__ void block_sum3(double& a, double& b, double& c, double* red) {
    wave_sum3(a, b, c);
    __syncthreads();
    const int tid = threadIdx.x;
    if ((tid & 63) == 0) { red[(tid >> 6) * 3 + 0] = a; red[(tid >> 6) * 3 + 1] = b; red[(tid >> 6) * 3 + 2] = c; }
    __syncthreads();
    a = (red[0] + red[3]) + (red[6] + red[9]);
    b = (red[1] + red[4]) + (red[7] + red[10]);
    c = (red[2] + red[5]) + (red[8] + red[11]);
}
// four sums at once / a maximum and three sums at once: each quantity's reduction tree is that of block_sum3 (same bits), one pair of
// barriers for all of them (the dogleg's bookkeeping was seven reductions per iteration: now three)
__device__ __forceinline__ void block_sum4(double& a, double& b, double& c, double& d, double* red) {
    wave_sum3(a, b, c);
    d = wave_sum(d);
    __syncthreads();
    const int tid = threadIdx.x;
    if ((tid & 63) == 0) { red[(tid >> 6) * 4 + 0] = a; red[(tid >> 6) * 4 + 1] = b; red[(tid >> 6) * 4 + 2] = c; red[(tid >> 6) * 4 + 3] = d; }
    __syncthreads();
    a = (red[0] + red[4]) + (red[8] + red[12]);
    b = (red[1] + red[5]) + (red[9] + red[13]);
    c = (red[2] + red[6]) + (red[10] + red[14]);
    d = (red[3] + red[7]) + (red[11] + red[15]);
}
__device__ __forceinline__ void block_max_sum3(double& mx, double& a, double& b, double& c, double* red) {
    wave_sum3(a, b, c);
    mx = wave_max(mx);
    __syncthreads();
    const int tid = threadIdx.x;
    if ((tid & 63) == 0) { red[(tid >> 6) * 4 + 0] = a; red[(tid >> 6) * 4 + 1] = b; red[(tid >> 6) * 4 + 2] = c; red[(tid >> 6) * 4 + 3] = mx; }
    __syncthreads();
    a = (red[0] + red[4]) + (red[8] + red[12]);
    b = (red[1] + red[5]) + (red[9] + red[13]);
    c = (red[2] + red[6]) + (red[10] + red[14]);
    mx = fmax(fmax(red[3], red[7]), fmax(red[11], red[15]));
}
__device__ __forceinline__ double block_sum(double a, double* red) {
    double b = 0.0, c = 0.0;
    block_sum3(a, b, c, red);
    return a;
}
__device__ __forceinline__ double block_max(double a, double* red) {
    a = wave_max(a);
    __syncthreads();
    const int tid = threadIdx.x;
    if ((tid & 63) == 0) red[tid >> 6] = a;
    __syncthreads();
    return fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
}

// ------------------------------------------------------------------------------------------------
// Cooperative chains: the exchange step between the G workgroups of one chain (moshii_dev.h: CoopDev).
// Protocol (cdna_hip_programming.md, Guideline 16, form R1): payload words are 8-byte agent-scope stores (write-through: no release
// fence), every storing wave drains its stores, ONE lane then stores the rank's flag = the exchange's sequence number; a reader polls
// the G flags (one lane each, relaxed), and reads the other ranks' payload with agent-scope loads (they bypass this CU's L1, which is
// never refreshed by another CU's stores).  Two slots per rank (parity of the sequence number): a rank can only be one exchange ahead
// of the slowest one -- it posts exchange s + 1 after it has read everybody's s, and needs everybody's s + 1 before it posts s + 2 --
// so slot s & 1 is never rewritten while somebody still reads it.  Every wait is bounded; a rank that gives up raises the group's
// abort word, everybody unwinds and the host reports the launch as failed (co-residency of the G workgroups is the caller's job).
// ------------------------------------------------------------------------------------------------
#if defined(__HIP_DEVICE_COMPILE__)
#define MOSHII_DRAIN_VMEM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#else
#define MOSHII_DRAIN_VMEM() do {} while (0)
#endif
__device__ __forceinline__ unsigned long long f64_bits(double v) { unsigned long long b; __builtin_memcpy(&b, &v, 8); return b; }
__device__ __forceinline__ double bits_f64(unsigned long long b) { double v; __builtin_memcpy(&v, &b, 8); return v; }
__device__ __forceinline__ void coop_st(MOSHII_GP(unsigned long long) p, double v) {
    __hip_atomic_store(p, f64_bits(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double coop_ld(MOSHII_GP(unsigned long long) p) {
    return bits_f64(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
// The same accesses at the scope of ONE XCD's L2 (the L2 = true forms below), for groups whose ranks share an XCD: stores sc0 --
// acknowledged by the L2, where an agent-scope (sc1) store is written through to the memory side before the storing wave's vmcnt moves --
// and loads sc0 nt: a plain sc0 load is served by this CU's L1 (tools/ubench_scope.hip: a word cached there is never seen to change, with
// or without buffer_inv sc0), a non-temporal one never is.  Round 4 ran every exchange this way when the ranks' XCC ids (s_getreg
// HW_REG_XCC_ID, compared in a first agent-scope exchange) agreed: correct, and NO faster -- 171.5 against 174.9 us per frame on the body
// solve, 652 against 649 us per cold frame in config 3's large exchange: the exchange is bound by the readers' round trips, which an
// sc1 load and an nt load pay alike (0.18 us a settled poll, either way), not by the write-through.  Not instantiated any more.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ void coop_st_l2(MOSHII_GP(unsigned long long) p, unsigned long long v) {
    asm volatile("global_store_dwordx2 %0, %1, off sc0" :: "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ unsigned long long coop_ld_l2(MOSHII_GP(unsigned long long) p) {
    unsigned long long v;
    asm volatile("global_load_dwordx2 %0, %1, off sc0 nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void coop_st32_l2(MOSHII_GP(unsigned int) p, unsigned v) {
    asm volatile("global_store_dword %0, %1, off sc0" :: "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ unsigned coop_ld32_l2(MOSHII_GP(unsigned int) p) {
    unsigned v;
    asm volatile("global_load_dword %0, %1, off sc0 nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
#endif
template <bool L2> __device__ __forceinline__ void coop_st64(MOSHII_GP(unsigned long long) p, unsigned long long v) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (L2) { coop_st_l2(p, v); return; }
#endif
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool L2> __device__ __forceinline__ unsigned long long coop_ld64(MOSHII_GP(unsigned long long) p) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (L2) return coop_ld_l2(p);
#endif
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool L2> __device__ __forceinline__ void coop_st32(MOSHII_GP(unsigned int) p, unsigned v) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (L2) { coop_st32_l2(p, v); return; }
#endif
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool L2> __device__ __forceinline__ unsigned coop_ld32(MOSHII_GP(unsigned int) p) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (L2) return coop_ld32_l2(p);
#endif
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned coop_begin(const Ctx& cx) { return (unsigned)__builtin_amdgcn_readfirstlane((int)cx.scal[S_COOP_SEQ]) + 1u; }   // (a scalar)
__device__ __forceinline__ MOSHII_GP(unsigned long long) coop_slot(const CoopCtx& co, unsigned seq, int r) {
    return co.slots + ((size_t)(seq & 1u) * co.G + r) * co.slot_doubles;
}
// 16-byte write-through stores / L1-bypassing loads of a slot (buffer instructions with the sc1 bit: 8-byte agent-scope accesses run at
// 0.54-0.70x the rate of 16-byte ones -- MI355X_MICROARCH.md); off16 = index of the 16-byte unit inside the slot
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MOSHII_COOP_NO_BUF)
typedef unsigned coop_u4 __attribute__((ext_vector_type(4)));
struct CoopSlot { __amdgpu_buffer_rsrc_t rs; };
__device__ __forceinline__ CoopSlot coop_slot16(const CoopCtx& co, unsigned seq, int r) {   // (seq, r: scalars)
    CoopSlot cs;
    cs.rs = __builtin_amdgcn_make_buffer_rsrc((void*)coop_slot(co, seq, r), 0, co.slot_doubles * 8, 0x00020000);
    return cs;
}
template <bool L2 = false>
__device__ __forceinline__ void coop_st16(const CoopSlot& cs, int off16, double a, double b) {
    const unsigned long long ba = f64_bits(a), bb = f64_bits(b);
    const coop_u4 v = {(unsigned)ba, (unsigned)(ba >> 32), (unsigned)bb, (unsigned)(bb >> 32)};
    if constexpr (L2) __builtin_amdgcn_raw_buffer_store_b128(v, cs.rs, off16 * 16, 0, 1);   // (aux 1: sc0)
    else __builtin_amdgcn_raw_buffer_store_b128(v, cs.rs, off16 * 16, 0, 16);               // (aux 16: sc1)
}
template <bool L2 = false>
__device__ __forceinline__ void coop_ld16(const CoopSlot& cs, int off16, double& a, double& b) {
    coop_u4 v;
    if constexpr (L2) v = __builtin_amdgcn_raw_buffer_load_b128(cs.rs, off16 * 16, 0, 3);   // (aux 3: sc0 nt)
    else v = __builtin_amdgcn_raw_buffer_load_b128(cs.rs, off16 * 16, 0, 16);
    a = bits_f64(((unsigned long long)v[1] << 32) | v[0]);
    b = bits_f64(((unsigned long long)v[3] << 32) | v[2]);
}
#else
struct CoopSlot { MOSHII_GP(unsigned long long) p; };
__device__ __forceinline__ CoopSlot coop_slot16(const CoopCtx& co, unsigned seq, int r) { CoopSlot cs; cs.p = coop_slot(co, seq, r); return cs; }
template <bool L2 = false>
__device__ __forceinline__ void coop_st16(const CoopSlot& cs, int off16, double a, double b) { coop_st(cs.p + 2 * (size_t)off16, a); coop_st(cs.p + 2 * (size_t)off16 + 1, b); }
template <bool L2 = false>
__device__ __forceinline__ void coop_ld16(const CoopSlot& cs, int off16, double& a, double& b) { a = coop_ld(cs.p + 2 * (size_t)off16); b = coop_ld(cs.p + 2 * (size_t)off16 + 1); }
#endif
// Test aid (MOSHII_COOP_SKEW=seed, tests/test_gpu_fullsize.py: the ranks' arrival order at the exchanges, randomised): the whole workgroup
// sleeps 0 .. 3 x 127 x 64 clocks (0 .. ~10 us), drawn from (seed, rank, exchange).  Results must not move by a bit.
__device__ __forceinline__ void coop_skew(const CoopCtx& co, unsigned seq) {
    if (co.skew != 0) {
        unsigned h = (seq * 2654435761u) ^ ((unsigned)co.rank * 40503u + (unsigned)co.skew * 977u);
        h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
        for (unsigned i = h & 3u; i > 0u; --i) __builtin_amdgcn_s_sleep(127);
    }
}
// All threads call, after their payload stores: publish this rank's slot, wait for every rank's.  false: the group is broken.
template <bool L2 = false>
__device__ __forceinline__ bool coop_publish_wait(const CoopCtx& co, const Ctx& cx, unsigned seq) {
    const int tid = threadIdx.x;
    coop_skew(co, seq);
    PROF_T(_tp0);
    MOSHII_DRAIN_VMEM();
    __syncthreads();
    PROF_ACC(33, _tp0);
    PROF_T(_tp1);
#if defined(MOSHII_COOP_FENCES)
    // Development variant (python -m moshpp_amd.build --variant=fences -DMOSHII_COOP_FENCES): the flag as a C++ release store / the read side
    // closed by an acquire fence, at agent scope.  On gfx950 that is a buffer_wbl2 sc1 before the flag and a buffer_inv sc1 behind the
    // poll -- an L2 write-back the written-through payload does not need, and the loss of this CU's L1 (model tables) at every exchange.
    // Measured beside the shipped protocol in DESIGN.md section 4b; the shipped one orders the same accesses by construction: payload
    // stores carry sc1 and are drained (vmcnt(0)) before the barrier that precedes the flag store; payload loads carry sc1 (never
    // served by an L1) and are issued after the poll has returned the flag.
    if (tid == 0) __hip_atomic_store(co.flags + co.rank, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
#else
    if (tid == 0) coop_st32<L2>(co.flags + co.rank, seq);
#endif
    TRACE_STAMP(co, seq, 1);
    if (tid < co.G && cx.scal[S_COOP_FAIL] == 0.0) {
        bool ok = true;
        unsigned spins = 0;
        while ((int)(coop_ld32<L2>(co.flags + tid) - seq) < 0) {
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 255u) == 0u)   // now and then: has somebody given up?  have we waited for ~0.1 s (an exchange takes microseconds)?
                if (spins > (1u << 21) || __hip_atomic_load(co.flags + co.G, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { ok = false; break; }
        }
        if (!ok) {
            __hip_atomic_store(co.flags + co.G, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            cx.scal[S_COOP_FAIL] = 1.0;
        }
    }
    __syncthreads();
#if defined(MOSHII_COOP_FENCES)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
    PROF_ACC(34, _tp1);
    TRACE_STAMP(co, seq, 2);
    return cx.scal[S_COOP_FAIL] == 0.0;
}
// All threads call once they have read what they need of the other ranks' slots.
__device__ __forceinline__ void coop_end(const Ctx& cx, unsigned seq) {
    __syncthreads();
    if (threadIdx.x == 0) cx.scal[S_COOP_SEQ] = (double)seq;
    __syncthreads();
}
// A few numbers per rank (NS <= 4 doubles): 8-byte {tag = sequence number, half a double} granules in the last 32 words of the slot --
// the data is its own flag (Guideline 16, form R2): no drain, no flag hop; a reader polls the granule until its tag is this exchange's.
// All threads call; on return land[r * NS + k] (LDS) holds value k of rank r.  Follow with coop_end().
template <int NS, bool L2>
__device__ __forceinline__ bool coop_exchange_small_(const CoopCtx& co, const Ctx& cx, unsigned seq, const double (&mine)[NS], double* land) {
    const int tid = threadIdx.x;
    const int goff = co.slot_doubles - 32;
    coop_skew(co, seq);
    if (tid < 2 * NS) {
        double v = mine[0];
#pragma unroll
        for (int k = 1; k < NS; ++k) if ((tid >> 1) == k) v = mine[k];
        const unsigned long long b = f64_bits(v);
        const unsigned half = (tid & 1) ? (unsigned)(b >> 32) : (unsigned)b;
        coop_st64<L2>(coop_slot(co, seq, co.rank) + goff + tid, ((unsigned long long)seq << 32) | half);
    }
    TRACE_STAMP(co, seq, 1);
    if (tid < co.G * 2 * NS && cx.scal[S_COOP_FAIL] == 0.0) {
        const int r = tid / (2 * NS), k = tid - r * 2 * NS;
        auto* g = coop_slot(co, seq, r) + goff + k;
        unsigned long long x;
        unsigned spins = 0;
        bool ok = true;
        while (((x = coop_ld64<L2>(g)) >> 32) != seq) {
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 255u) == 0u)
                if (spins > (1u << 21) || __hip_atomic_load(co.flags + co.G, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { ok = false; break; }
        }
        if (!ok) {
            __hip_atomic_store(co.flags + co.G, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            cx.scal[S_COOP_FAIL] = 1.0;
        }
        reinterpret_cast<unsigned*>(land)[tid] = (unsigned)x;   // (little-endian halves: word 2 (r NS + j) + h of land)
    }
    __syncthreads();
    TRACE_STAMP(co, seq, 2);
    return cx.scal[S_COOP_FAIL] == 0.0;
}
template <int NS>
__device__ __forceinline__ bool coop_exchange_small(const CoopCtx& co, const Ctx& cx, unsigned seq, const double (&mine)[NS], double* land) {
    return coop_exchange_small_<NS, false>(co, cx, seq, mine, land);
}

// Rodrigues + SO(3) left Jacobian; same formulas and small-angle switch as oracle/stageii_oracle.py:rodrigues.
__device__ __forceinline__ void rodrigues_dev(const double* r, double* R, double* Jl) {
    const double x = r[0], y = r[1], z = r[2];
    const double t2 = x * x + y * y + z * z;
    double a, b, c;
    if (t2 < 1e-6) {
        a = 1.0 - t2 / 6.0 + t2 * t2 / 120.0;
        b = 0.5 - t2 / 24.0 + t2 * t2 / 720.0;
        c = 1.0 / 6.0 - t2 / 120.0 + t2 * t2 / 5040.0;
    } else {
        const double t = sqrt(t2);
        double s, co;
        sincos(t, &s, &co);
        a = s / t;
        b = (1.0 - co) / t2;
        c = (t - s) / (t2 * t);
    }
    // K = [r]x ; K^2 = r r^T - t2 I
    const double K2[9] = {x * x - t2, x * y, x * z, x * y, y * y - t2, y * z, x * z, y * z, z * z - t2};
    const double Km[9] = {0.0, -z, y, z, 0.0, -x, -y, x, 0.0};
#pragma unroll
    for (int e = 0; e < 9; ++e) {
        const double id = (e == 0 || e == 4 || e == 8) ? 1.0 : 0.0;
        R[e] = id + a * Km[e] + b * K2[e];
        Jl[e] = id + b * Km[e] + c * K2[e];
    }
}

// rotation only (same formulas / small-angle switch as rodrigues_dev)
__device__ __forceinline__ void rodrigues_R(const double* r, double* R) {
    const double x = r[0], y = r[1], z = r[2];
    const double t2 = x * x + y * y + z * z;
    double a, b;
    if (t2 < 1e-6) {
        a = 1.0 - t2 / 6.0 + t2 * t2 / 120.0;
        b = 0.5 - t2 / 24.0 + t2 * t2 / 720.0;
    } else {
        const double t = sqrt(t2);
        double s, co;
        sincos(t, &s, &co);
        a = s / t;
        b = (1.0 - co) / t2;
    }
    const double K2[9] = {x * x - t2, x * y, x * z, x * y, y * y - t2, y * z, x * z, y * z, z * z - t2};
    const double Km[9] = {0.0, -z, y, z, 0.0, -x, -y, x, 0.0};
#pragma unroll
    for (int e = 0; e < 9; ++e) R[e] = ((e == 0 || e == 4 || e == 8) ? 1.0 : 0.0) + a * Km[e] + b * K2[e];
}

__device__ __forceinline__ void mat3_mul(const double* A, const double* Bm, double* C) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            C[i * 3 + j] = A[i * 3 + 0] * Bm[0 * 3 + j] + A[i * 3 + 1] * Bm[1 * 3 + j] + A[i * 3 + 2] * Bm[2 * 3 + j];
}
__device__ __forceinline__ void mat3_vec(const double* A, double x, double y, double z, double& ox, double& oy, double& oz) {
    ox = A[0] * x + A[1] * y + A[2] * z;
    oy = A[3] * x + A[4] * y + A[5] * z;
    oz = A[6] * x + A[7] * y + A[8] * z;
}

// marker from its three vertices (transformed_lm.py:138-159); optionally the 3x9 Jacobian wrt (v0,v1,v2).
__device__ __forceinline__ void marker_eval(const double* c, const double* v0, const double* v1, const double* v2,
                                            double* mk, double* L /* 27 or nullptr */) {
    const double e1[3] = {v1[0] - v0[0], v1[1] - v0[1], v1[2] - v0[2]};
    const double e2[3] = {v2[0] - v0[0], v2[1] - v0[1], v2[2] - v0[2]};
    const double l1 = sqrt(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]);
    const double f1[3] = {e1[0] / l1, e1[1] / l1, e1[2] / l1};
    const double nv[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
    const double ln = sqrt(nv[0] * nv[0] + nv[1] * nv[1] + nv[2] * nv[2]);
    const double f2[3] = {nv[0] / ln, nv[1] / ln, nv[2] / ln};
    const double f3[3] = {f1[1] * f2[2] - f1[2] * f2[1], f1[2] * f2[0] - f1[0] * f2[2], f1[0] * f2[1] - f1[1] * f2[0]};
#pragma unroll
    for (int i = 0; i < 3; ++i) mk[i] = v0[i] + c[0] * f1[i] + c[1] * f2[i] + c[2] * f3[i];
    if (L == nullptr) return;
    // D1 = (I - f1 f1^T)/l1 ; D2 = (I - f2 f2^T)/ln
    double D1[9], D2[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const double id = (i == j) ? 1.0 : 0.0;
            D1[i * 3 + j] = (id - f1[i] * f1[j]) / l1;
            D2[i * 3 + j] = (id - f2[i] * f2[j]) / ln;
        }
    // dn/de1 = -[e2]x ; dn/de2 = [e1]x
    const double N1[9] = {0.0, e2[2], -e2[1], -e2[2], 0.0, e2[0], e2[1], -e2[0], 0.0};
    const double N2[9] = {0.0, -e1[2], e1[1], e1[2], 0.0, -e1[0], -e1[1], e1[0], 0.0};
    double F21[9], F22[9];
    mat3_mul(D2, N1, F21);
    mat3_mul(D2, N2, F22);
    const double S1[9] = {0.0, -f1[2], f1[1], f1[2], 0.0, -f1[0], -f1[1], f1[0], 0.0};
    const double S2[9] = {0.0, -f2[2], f2[1], f2[2], 0.0, -f2[0], -f2[1], f2[0], 0.0};
    double T1[9], T2[9], T3[9];
    mat3_mul(S2, D1, T1);    // [f2]x D1
    mat3_mul(S1, F21, T2);   // [f1]x df2/de1
    mat3_mul(S1, F22, T3);   // [f1]x df2/de2
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int e = i * 3 + j;
            const double de1 = c[0] * D1[e] + c[1] * F21[e] + c[2] * (T2[e] - T1[e]);
            const double de2 = c[1] * F22[e] + c[2] * T3[e];
            const double id = (i == j) ? 1.0 : 0.0;
            L[i * 9 + j] = id - de1 - de2;
            L[i * 9 + 3 + j] = de1;
            L[i * 9 + 6 + j] = de2;
        }
}

// fullpose[d] = pose[d] for the leading body dofs, hands_mean + pose_hand . selected_components beyond
// (smpl_fast_derivatives.py:194-204); the two-accumulator order is part of the kernel's numerics
__device__ __forceinline__ double fullpose_entry(const ModelDev& md, const double* pose, int d) {
    const int bd = md.body_dof, nhf = md.nhand_full;
    if (d < bd) return pose[d];
    const int h = d - bd;
    const int lo = md.col_lo[h], hi = md.col_hi[h];   // components with a non-zero entry in column h
    double v0 = md.hands_mean[h], v1 = 0.0;
    int i = lo;
    for (; i + 2 <= hi; i += 2) {
        v0 += pose[bd + i] * md.comps[i * nhf + h];
        v1 += pose[bd + i + 1] * md.comps[(i + 1) * nhf + h];
    }
    if (i < hi) v0 += pose[bd + i] * md.comps[i * nhf + h];
    return v0 + v1;
}

// dst[a][i] = base[a][i] + sum_{k in klist} posedirs[a][i][9(k-1) .. 9k) . (R_k - I)   (cx.feat holds R - I of every joint).
// item = (coordinate i, vertex pair): every lane streams 16-byte pairs of the vertex-fastest posedirs slice (fully
// coalesced rows of Nvp doubles).  Only the joints that are FREE in the running solve change between evaluations, so the
// chain keeps the other joints' contribution in cx.vconst and each evaluation streams |klist| / (K-1) of the slice.
__device__ __forceinline__ void posedirs_partial(const Ctx& cx, const AttachDev& at, const int* klist, int nk,
                                                 const double* base, double* dst) {
    const int tid = threadIdx.x;
    const int Nv = at.Nv, Nvp = at.Nvp, Nvh = Nvp >> 1;
    for (int it = tid; it < 3 * Nvh; it += MOSHII_TPB) {
        const int i = it / Nvh, a2 = it - i * Nvh;
        double s0x = 0.0, s0y = 0.0, s1x = 0.0, s1y = 0.0, s2x = 0.0, s2y = 0.0;
        const auto* pp = gptr(reinterpret_cast<const v2d*>(at.Pt)) + (size_t)(i * 9) * Nvh + a2;
        // bursts of PB joints: all 9 PB row loads of a burst are issued before the first is consumed.  The stream runs at the
        // CU's L2 bandwidth only with >= 27 loads in flight per lane (measured: 9 in flight 39 us/frame, 27 in flight 30;
        // interleaving the loads with the multiplies of the previous joint was slower than either).
        constexpr int PB = 3;
        for (int idx0 = 0; idx0 < nk; idx0 += PB) {
            v2d q[PB][9];
            int kk[PB];
#pragma unroll
            for (int u = 0; u < PB; ++u) {
                kk[u] = klist[min(idx0 + u, nk - 1)];   // (past the list: a re-read that is not added)
                const auto* pk = pp + (size_t)((kk[u] - 1) * 27) * Nvh;
#pragma unroll
                for (int e = 0; e < 9; ++e) q[u][e] = pk[(size_t)e * Nvh];
            }
#pragma unroll
            for (int u = 0; u < PB; ++u) {
                if (idx0 + u < nk) {   // (uniform)
                    const double* f = &cx.feat[kk[u] * 9];
#pragma unroll
                    for (int e = 0; e < 9; e += 3) {
                        s0x += q[u][e][0] * f[e]; s0y += q[u][e][1] * f[e];
                        s1x += q[u][e + 1][0] * f[e + 1]; s1y += q[u][e + 1][1] * f[e + 1];
                        s2x += q[u][e + 2][0] * f[e + 2]; s2y += q[u][e + 2][1] * f[e + 2];
                    }
                }
            }
        }
        const int a = 2 * a2;
        if (a < Nv) dst[a * 3 + i] = base[a * 3 + i] + ((s0x + s1x) + s2x);
        if (a + 1 < Nv) dst[(a + 1) * 3 + i] = base[(a + 1) * 3 + i] + ((s0y + s1y) + s2y);
    }
}

// The same for the vertices [a_lo, a_hi) only (cooperative chains: a rank's share of the attached vertices).  A share is a few dozen
// vertices, i.e. far fewer (coordinate, vertex pair) items than threads: the joint list of an item is then dealt to JG = 2 or 4
// ADJACENT lanes (bursts of PB joints, round-robin), whose partial sums meet through DPP quad moves -- the pass is a latency chain per
// lane (one L2 round trip per burst), so this divides its length, not just its width.
__device__ __forceinline__ void posedirs_partial_range(const Ctx& cx, const AttachDev& at, const int* klist, int nk,
                                                       const double* base, double* dst, int a_lo, int a_hi) {
    // wavefronts 1 .. 3 only: wavefront 0 walks the kinematic chain meanwhile (eval_forward, F3), and a share is small enough for 192 lanes
    constexpr int LANES = MOSHII_TPB - 64;
    if (threadIdx.x < 64) return;
    const int tid = threadIdx.x - 64;
    const int Nvp = at.Nvp, Nvh = Nvp >> 1;
    const int p_lo = a_lo >> 1, np = ((a_hi + 1) >> 1) - p_lo;   // the vertex pairs that cover the range
    const int items = 3 * np;
    if (items <= 0) return;
    const int jsh = (items * 4 <= LANES) ? 2 : ((items * 2 <= LANES) ? 1 : 0);   // log2 of the lanes per item
    const int JG = 1 << jsh;
#ifndef MOSHII_COOP_PB
#define MOSHII_COOP_PB 3
#endif
    constexpr int PB = MOSHII_COOP_PB;   // joints per burst: 9 PB 16-byte loads in flight per lane (5 measured slower than 3: 181 vs 177.5 us per frame)
    for (int l0 = 0; l0 < items * JG; l0 += LANES) {   // (uniform trip count: the DPP moves below need whole wavefronts)
        const int lin = l0 + tid;
        const bool live = lin < items * JG;
        const int it = min(lin, items * JG - 1) >> jsh, jg = lin & (JG - 1);
        const int i = it / np, a2 = p_lo + (it - i * np);
        double s0x = 0.0, s0y = 0.0, s1x = 0.0, s1y = 0.0, s2x = 0.0, s2y = 0.0;
        const auto* pp = gptr(reinterpret_cast<const v2d*>(at.Pt)) + (size_t)(i * 9) * Nvh + a2;
        for (int idx0 = jg * PB; idx0 < nk; idx0 += JG * PB) {
            v2d q[PB][9];
            int kk[PB];
#pragma unroll
            for (int u = 0; u < PB; ++u) {
                kk[u] = klist[min(idx0 + u, nk - 1)];
                const auto* pk = pp + (size_t)((kk[u] - 1) * 27) * Nvh;
#pragma unroll
                for (int e = 0; e < 9; ++e) q[u][e] = pk[(size_t)e * Nvh];
            }
#pragma unroll
            for (int u = 0; u < PB; ++u) {
                if (idx0 + u < nk) {
                    const double* f = &cx.feat[kk[u] * 9];
#pragma unroll
                    for (int e = 0; e < 9; e += 3) {
                        s0x += q[u][e][0] * f[e]; s0y += q[u][e][1] * f[e];
                        s1x += q[u][e + 1][0] * f[e + 1]; s1y += q[u][e + 1][1] * f[e + 1];
                        s2x += q[u][e + 2][0] * f[e + 2]; s2y += q[u][e + 2][1] * f[e + 2];
                    }
                }
            }
        }
        double sx = (s0x + s1x) + s2x, sy = (s0y + s1y) + s2y;
        if (jsh >= 1) { sx += dpp_f64<DPP_QUAD_1032>(sx); sy += dpp_f64<DPP_QUAD_1032>(sy); }   // (uniform branches)
        if (jsh >= 2) { sx += dpp_f64<DPP_QUAD_2301>(sx); sy += dpp_f64<DPP_QUAD_2301>(sy); }
        const int a = 2 * a2;
        if (live && jg == 0) {
            if (a >= a_lo && a < a_hi) dst[a * 3 + i] = base[a * 3 + i] + sx;
            if (a + 1 >= a_lo && a + 1 < a_hi) dst[(a + 1) * 3 + i] = base[(a + 1) * 3 + i] + sy;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// forward evaluation of every residual block at (pose, trans): leaves joint transforms, posed
// attached vertices, simulated markers, weighted data residuals and the prior's l-vectors in LDS.
// ------------------------------------------------------------------------------------------------
// klist/nk: the joints whose pose-corrective contribution is summed here; vbase = rest vertices + the contribution of
// every other joint (cx.vconst, see posedirs_partial) -- or klist = all joints and vbase = v_shaped.
// XT (extended variant): the free shape coefficients s ride behind the pose variables (pose[NP .. NP + nshape)); the rest
// vertices and the regressed joints are re-shaped with them at every evaluation (smpl_fast_derivatives.py:186-191).
// COOP (cooperative chains): this workgroup is rank co.rank of co.G; it evaluates the markers [co.mlo, co.mhi) and their vertices, the
// rank co.prior_rank the prior; the partial sums of squares meet in an exchange, after which every rank holds the same totals.
template <bool XT, bool COOP = false>
__device__ Sse eval_forward(const Ctx& cx, const ModelDev& md, const AttachDev& at, const PriorDev& pr,
                            const OptsDev& op, const double* pose, const double* trans, const FrameParams& fp,
                            const uint8_t* visrow, const int* klist, int nk, const double* vbase, const bool light,
                            const CoopCtx& co = CoopCtx()) {
    const int tid = threadIdx.x;
    const int m_lo = COOP ? co.mlo : 0, m_hi = COOP ? co.mhi : at.M;   // this workgroup's markers ...
    const int a_lo = 3 * m_lo, a_hi = COOP ? 3 * co.mhi : at.Nv;        // ... and attached vertices (a = 3 m + s)
    const int K = md.K, P = md.P, bd = md.body_dof, hd = md.hand_dof, nhf = md.nhand_full;
    PROF_BEGIN(); PROF_COUNT(20);
    // `light` (wave-uniform): the forward state in LDS -- rotations, chain, vertices, simulated markers, the prior's argmin and
    // value -- is that of THIS point already (the evaluation that ended the previous frame was at it, with the same joint
    // lists); only what depends on the frame's data is redone: data residuals, velocity / finger / shape sums, the weights.
    // Same arithmetic on the same stored values: the result has the bits of a full evaluation.
    // (cooperative chains: a rank without markers -- the prior's -- needs none of the body's forward state)
    const bool body = !COOP || m_hi > m_lo;
    if (!light && body) {
    // F1: fullpose = [pose[:bd], hands_mean + pose_hand . comps]
    // (cooperative chains: while no hand coefficient is free in the running solve -- S_TMP2, set with the column tables -- the hand part
    //  stands as run_phase computed it with the fixed joints' correctives: its PCA sums are a latency chain of their own)
    const int d_hi = (COOP && hd > 0 && cx.scal[S_TMP2] == 0.0) ? min(P, bd) : P;
    for (int d = tid; d < d_hi; d += MOSHII_TPB) cx.fullpose[d] = fullpose_entry(md, pose, d);
    if constexpr (XT) {
        if (op.nshape > 0) {
            const int E = op.nshape, Nvp = at.Nvp;
            const double* shp = pose + md.NP;
            // (both sums: sixteen coefficients' loads in flight at a time -- one item per thread and 40 dependent round trips to the L2 made
            //  these two loops 17 us of a config-3 evaluation; the same two accumulators in the same order: the same bits)
            auto dot_s = [&](auto ld, double s0) {   // s0 + sum_e ld(e) shp[e], even e into s0, odd into s1 (a trailing odd one into s0)
                double s1 = 0.0;
                int e = 0;
                for (; e + 16 <= E; e += 16) {
                    double v[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) v[u] = ld(e + u);
#pragma unroll
                    for (int u = 0; u < 16; u += 2) { s0 += v[u] * shp[e + u]; s1 += v[u + 1] * shp[e + u + 1]; }
                }
                for (; e + 2 <= E; e += 2) { s0 += ld(e) * shp[e]; s1 += ld(e + 1) * shp[e + 1]; }
                if (e < E) s0 += ld(e) * shp[e];
                return s0 + s1;
            };
            for (int i = tid; i < 3 * K; i += MOSHII_TPB) {   // J = J0 + JS . s
                const auto* js = md.JS + (size_t)(i / 3) * E * 3 + (i % 3);
                cx.Jl[i] = dot_s([&](int e) { return js[e * 3]; }, md.J[i]);
            }
            const int na = COOP ? (a_hi - a_lo) : Nvp;   // (cooperative chains: this rank's vertices only)
            for (int it = tid; it < 3 * na; it += MOSHII_TPB) {   // rest vertices: vbase + S . s (vertex fastest: coalesced rows)
                const int i = it / na, a = (COOP ? a_lo : 0) + (it - i * na);
                const auto* sp = gptr(at.Ssh) + (size_t)i * Nvp + a;
                const size_t st = (size_t)3 * Nvp;
                const double sv = dot_s([&](int e) { return sp[e * st]; }, 0.0);
                if (a < at.Nv) cx.vshp[a * 3 + i] = vbase[a * 3 + i] + sv;
            }
            vbase = cx.vshp;
        }
    }
    __syncthreads();
    // F2: per joint rotation and pose feature R - I  (the Jacobian-only quantities are built in assemble())
    if (tid < K) {
        double R[9];
        rodrigues_R(&cx.fullpose[3 * tid], R);
#pragma unroll
        for (int e = 0; e < 9; ++e) {
            cx.Rloc[tid * 9 + e] = R[e];
            cx.feat[tid * 9 + e] = R[e] - ((e == 0 || e == 4 || e == 8) ? 1.0 : 0.0);
        }
        if (tid == 0) {
#pragma unroll
            for (int e = 0; e < 9; ++e) cx.Rw[e] = R[e];
            cx.tw[0] = cx.Jl[0]; cx.tw[1] = cx.Jl[1]; cx.tw[2] = cx.Jl[2];
        }
    }
    __syncthreads();
    PROF_LAP(0);
    // F3 (wave 0 only): kinematic chain G_j = G_par(j) . [R_j | J_j - J_par(j)], one tree level per step.  All K <= 64
    // joints live in one wavefront, whose LDS operations complete in order, so levels need no workgroup barrier;
    // waves 1..3 are already streaming posedirs (which needs only the local rotations) meanwhile.
#ifdef MOSHII_COOP_LEVEL_CHAIN
    constexpr bool JUMP = false;
#else
    constexpr bool JUMP = COOP;
#endif
    if (JUMP && tid < 64) {
        // Cooperative chains: the same products by pointer jumping -- every joint holds the composition of the local transforms from
        // itself up to (not including) its pointer `anc`, and in every step composes with what `anc` holds and takes over anc's pointer:
        // ceil(log2(depth + 1)) steps of one LDS round trip instead of one per tree level (the chain is what the other three wavefronts'
        // share of the pose correctives waits for here; the products are associated differently from the level-order walk: round-off).
        const bool act = tid < K;
        double R[9], t[3] = {0.0, 0.0, 0.0};
        int anc = -1;
        if (act) {
#pragma unroll
            for (int e = 0; e < 9; ++e) R[e] = cx.Rloc[tid * 9 + e];
            if (tid > 0) anc = md.parents[tid];
            const int pj = max(anc, 0);
#pragma unroll
            for (int e = 0; e < 3; ++e) t[e] = (tid > 0) ? cx.Jl[tid * 3 + e] - cx.Jl[pj * 3 + e] : cx.Jl[e];
        }
        for (int span = 1; span <= md.maxdepth; span *= 2) {   // (uniform)
            if (act) {
#pragma unroll
                for (int e = 0; e < 9; ++e) cx.Rw[tid * 9 + e] = R[e];
#pragma unroll
                for (int e = 0; e < 3; ++e) cx.tw[tid * 3 + e] = t[e];
                cx.jointslot[tid] = anc;   // (the table builder's scratch: free between table builds)
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (act && anc >= 0) {
                double Ra[9], Ro[9], ox, oy, oz;
#pragma unroll
                for (int e = 0; e < 9; ++e) Ra[e] = cx.Rw[anc * 9 + e];
                const double tax = cx.tw[anc * 3 + 0], tay = cx.tw[anc * 3 + 1], taz = cx.tw[anc * 3 + 2];
                const int aa = cx.jointslot[anc];
                mat3_mul(Ra, R, Ro);
                mat3_vec(Ra, t[0], t[1], t[2], ox, oy, oz);
#pragma unroll
                for (int e = 0; e < 9; ++e) R[e] = Ro[e];
                t[0] = ox + tax; t[1] = oy + tay; t[2] = oz + taz;
                anc = aa;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        if (act) {
#pragma unroll
            for (int e = 0; e < 9; ++e) cx.Rw[tid * 9 + e] = R[e];
#pragma unroll
            for (int e = 0; e < 3; ++e) cx.tw[tid * 3 + e] = t[e];
        }
    }
    if (!JUMP && tid < 64) {
        const int lvl_of = (tid < K) ? md.depth[tid] : -1;
        const int p = (tid < K && tid > 0) ? md.parents[tid] : 0;
        for (int lvl = 1; lvl <= md.maxdepth; ++lvl) {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (lvl_of == lvl) {
                double Rp[9], Rl[9], Ro[9];
#pragma unroll
                for (int e = 0; e < 9; ++e) { Rp[e] = cx.Rw[p * 9 + e]; Rl[e] = cx.Rloc[tid * 9 + e]; }
                mat3_mul(Rp, Rl, Ro);
#pragma unroll
                for (int e = 0; e < 9; ++e) cx.Rw[tid * 9 + e] = Ro[e];
                const double dx = cx.Jl[tid * 3 + 0] - cx.Jl[p * 3 + 0];
                const double dy = cx.Jl[tid * 3 + 1] - cx.Jl[p * 3 + 1];
                const double dz = cx.Jl[tid * 3 + 2] - cx.Jl[p * 3 + 2];
                double ox, oy, oz;
                mat3_vec(Rp, dx, dy, dz, ox, oy, oz);
                cx.tw[tid * 3 + 0] = ox + cx.tw[p * 3 + 0];
                cx.tw[tid * 3 + 1] = oy + cx.tw[p * 3 + 1];
                cx.tw[tid * 3 + 2] = oz + cx.tw[p * 3 + 2];
            }
        }
    }
    // F4: v_posed = vbase + sum_{k in klist} posedirs_k . vec(R_k - I) for the attached vertices.
    if constexpr (COOP) posedirs_partial_range(cx, at, klist, nk, vbase, cx.vposed, a_lo, a_hi);
    else posedirs_partial(cx, at, klist, nk, vbase, cx.vposed);
    __syncthreads();
    PROF_LAP(1);
    // F5: skinning  v = sum_j w_j (Rw_j (v_posed - J_j) + tw_j) + trans
    const int NW = at.NW;
    for (int a = a_lo + tid; a < a_hi; a += MOSHII_TPB) {
        const double px = cx.vposed[a * 3 + 0], py = cx.vposed[a * 3 + 1], pz = cx.vposed[a * 3 + 2];
        double ax = 0.0, ay = 0.0, az = 0.0;
        if (NW <= 4) {   // (uniform) the influences of the vertex in one round of loads (a rolled loop pays a memory round trip per influence)
            int jj[4];
            double wv[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) { jj[s] = gptr(at.wj)[a * NW + min(s, NW - 1)]; wv[s] = gptr(at.ww)[a * NW + min(s, NW - 1)]; }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                if (s < NW) {   // (uniform)
                    const int j = jj[s];
                    const double w = wv[s];
                    double ox, oy, oz;
                    mat3_vec(&cx.Rw[j * 9], px - cx.Jl[j * 3 + 0], py - cx.Jl[j * 3 + 1], pz - cx.Jl[j * 3 + 2], ox, oy, oz);
                    ax += w * (ox + cx.tw[j * 3 + 0]);
                    ay += w * (oy + cx.tw[j * 3 + 1]);
                    az += w * (oz + cx.tw[j * 3 + 2]);
                }
            }
        } else
        for (int s = 0; s < NW; ++s) {
            const int j = gptr(at.wj)[a * NW + s];
            const double w = gptr(at.ww)[a * NW + s];
            double ox, oy, oz;
            mat3_vec(&cx.Rw[j * 9], px - cx.Jl[j * 3 + 0], py - cx.Jl[j * 3 + 1], pz - cx.Jl[j * 3 + 2], ox, oy, oz);
            ax += w * (ox + cx.tw[j * 3 + 0]);
            ay += w * (oy + cx.tw[j * 3 + 1]);
            az += w * (oz + cx.tw[j * 3 + 2]);
        }
        cx.vpos[a * 3 + 0] = ax + trans[0];
        cx.vpos[a * 3 + 1] = ay + trans[1];
        cx.vpos[a * 3 + 2] = az + trans[2];
    }
    __syncthreads();
    }   // (!light)
    // F6: simulated markers + weighted data residual
    double sd = 0.0;
    for (int m = m_lo + tid; m < m_hi; m += MOSHII_TPB) {
        double mk[3];
        if (!light) {
            const double c[3] = {gptr(at.coef)[m * 3 + 0], gptr(at.coef)[m * 3 + 1], gptr(at.coef)[m * 3 + 2]};
            marker_eval(c, &cx.vpos[(3 * m + 0) * 3], &cx.vpos[(3 * m + 1) * 3], &cx.vpos[(3 * m + 2) * 3], mk, nullptr);
        } else {
            mk[0] = cx.msim[m * 3 + 0]; mk[1] = cx.msim[m * 3 + 1]; mk[2] = cx.msim[m * 3 + 2];
        }
        const bool v = visrow != nullptr && gptr(visrow)[m] != 0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            cx.msim[m * 3 + i] = mk[i];
            const double r = v ? fp.wt_data * (mk[i] - gptr(fp.obs)[m * 3 + i]) : 0.0;
            cx.res[m * 3 + i] = r;
            sd += r * r;
        }
    }
    PROF_LAP(2);
    // F8: velocity and finger terms
    double sv = 0.0, sh = 0.0;
    if (fp.has_velo)
        for (int i = tid; i < md.NP; i += MOSHII_TPB) { const double d = (pose[i] - cx.vtarget[i]) * fp.wt_velo; sv += d * d; }
    if (fp.use_fingers)
        for (int f = tid; f < op.nfinger; f += MOSHII_TPB) { const double d = pose[op.finger[f]] * fp.wt_poseH; sh += d * d; }
    double sf = 0.0, ss = 0.0, sy = 0.0;   // jaw term, shape regulariser, shape "stay" term (chmosh.py:685-699)
    if constexpr (XT) {
        if (fp.use_face)
            for (int f = tid; f < op.nface; f += MOSHII_TPB) { const double d = pose[op.face[f]] * fp.wt_poseF; sf += d * d; }
        if (fp.use_shape)
            for (int e = tid; e < op.nshape; e += MOSHII_TPB) {
                const double sv_ = pose[md.NP + e];
                const double d = sv_ * op.wt_shape;
                ss += d * d;
                if (fp.has_stay) { const double d2 = (sv_ - cx.shp0[e]) * op.wt_shape_stay; sy += d2 * d2; }
            }
    }
    PROF_LAP(40);
    // F7: prior: l_g = sqrt(.5) (x - mu_g) . L_g for every component, argmin of |l_g|^2 - log w_g
    const int np_ = op.nbody;
    double prior_ss = 0.0;
    if (np_ > 0 && light) prior_ss = cx.scal[S_PRIOR_SS];   // (stored by the evaluation that left this state; S_KBEST stands too)
    if (np_ > 0 && !light && (!COOP || co.rank == co.prior_rank)) {
        for (int b = tid; b < np_; b += MOSHII_TPB) cx.xb[b] = pose[op.body[b]];
        __syncthreads();
        const int G = pr.G;
        const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);   // (wave index as a scalar: scalar row addresses)
        const int c0 = min(lane, np_ - 1), c1 = min(lane + 64, np_ - 1);   // clamped: every load is in bounds, no branch
        // Scores |l_g|^2 of the components ga and gb (may be the same one), by all four wavefronts: wave w takes the rows
        // [w RC, (w + 1) RC) of L_g -- lane a accumulates column a (and a + 64), every load one contiguous run of a row, 2 x 16
        // row loads in flight per lane, each column sum split into an even-row and an odd-row chain -- the four partial column
        // sums meet in LDS (cx.ell) and wave 0 squares and reduces them.  The entries above the diagonal are stored zeros
        // (moshii_prior_create): no triangle mask.  The arithmetic of a component does not depend on its partner, so a score
        // is the same bits whichever pair it was computed in.  Ends with a barrier; cx.score[ga], cx.score[gb] are then valid.
        auto prior_pair = [&](int ga, int gb) {
            const int RC = 16 * ((np_ + 63) / 64);
            const int r_lo = wv * RC, r_hi = min(np_, r_lo + RC);
            // (the factors, the means and cx.xb are padded by 16 rows / entries: rows past the end are read and weighted 0)
            const auto* La = pr.chols + (size_t)ga * np_ * np_ + c0;
            const auto* Lb = pr.chols + (size_t)gb * np_ * np_ + c0;
            const auto* mua = pr.means + (size_t)ga * np_;
            const auto* mub = pr.means + (size_t)gb * np_;
            // x - mu_g of both components first goes to this wave's two slices of cx.ell (one coalesced load of the means per
            // lane; 63 scalar loads of them inside the fma chains cost as much as the matrix loads), is read back with uniform
            // addresses in the chains, and the slices then take the wave's partial column sums (a wave's LDS traffic is ordered).
            double* pa = cx.ell + (size_t)wv * np_;               // [2][4 waves][np_ (+ slack: the next slice, cx.score)]
            double* pb = cx.ell + (size_t)(4 + wv) * np_;
            for (int r = lane; r < np_; r += 64) { pa[r] = cx.xb[r] - mua[r]; pb[r] = cx.xb[r] - mub[r]; }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0, ha = 0.0, hb = 0.0;
            for (int r0 = r_lo; r0 < r_hi; r0 += 16) {
                double va[16], vb[16];
                const auto* Ra = La + (size_t)r0 * np_;
                const auto* Rb = Lb + (size_t)r0 * np_;
#pragma unroll
                for (int k = 0; k < 16; ++k) { va[k] = Ra[k * np_]; vb[k] = Rb[k * np_]; }
#pragma unroll
                for (int k = 0; k < 16; k += 2) {
                    const int r = r0 + k;
                    const double m0 = (r < r_hi) ? 1.0 : 0.0, m1 = (r + 1 < r_hi) ? 1.0 : 0.0;   // (uniform; rows past r_hi: whatever is there)
                    a0 = fma(pa[r] * m0, va[k], a0); a1 = fma(pa[r + 1] * m1, va[k + 1], a1);
                    b0 = fma(pb[r] * m0, vb[k], b0); b1 = fma(pb[r + 1] * m1, vb[k + 1], b1);
                }
                if (np_ > 64)   // columns 64.. (SMPL's 69-dof prior): non-zero in rows >= 64 only
                    for (int r = max(r0, 64); r < min(r0 + 16, r_hi); ++r) {
                        ha = fma(pa[r], La[(size_t)r * np_ + (c1 - c0)], ha);
                        hb = fma(pb[r], Lb[(size_t)r * np_ + (c1 - c0)], hb);
                    }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (lane < np_) { pa[lane] = a0 + a1; pb[lane] = b0 + b1; }
            if (lane + 64 < np_) { pa[lane + 64] = ha; pb[lane + 64] = hb; }
            __syncthreads();
            if (wv == 0) {
                double sqa = 0.0, sqb = 0.0, dum = 0.0;
                for (int c = lane; c < np_; c += 64) {
                    const double* qa = cx.ell + c;
                    const double* qb = cx.ell + (size_t)4 * np_ + c;
                    const double la = ((qa[0] + qa[np_]) + (qa[2 * np_] + qa[3 * np_])) * 0.70710678118654757;
                    const double lb = ((qb[0] + qb[np_]) + (qb[2 * np_] + qb[3 * np_])) * 0.70710678118654757;
                    sqa += la * la; sqb += lb * lb;
                }
                wave_sum3(sqa, sqb, dum);
                if (lane == 0) { cx.score[ga] = sqa; cx.score[gb] = sqb; }
            }
            __syncthreads();
        };
        // The max-mixture needs the argmin and its value only.  With a reference point x0 at which every component's
        // |l_g| is known, |l_g(x)| >= |l_g(x0)| - cnorm_g |x - x0| bounds every other component from below; if the
        // component that won last time, evaluated exactly, stays below all those bounds (with a margin far above rounding),
        // it IS the minimum and the other seven evaluations are skipped -- the result is what the full evaluation gives.
        int kb = 0;
        bool settled = false;
        PROF_LAP(41);
        if (cx.scal[S_PRIOR_REF] != 0.0) {
            double d2 = 0.0;
            for (int b = lane; b < np_; b += 64) { const double d = cx.xb[b] - cx.px0[b]; d2 += d * d; }
            const double dist = sqrt(wave_sum(d2));   // (every wave for itself: same data, same order, same bits)
            const int k0 = __builtin_amdgcn_readfirstlane((int)cx.scal[S_PRIOR_KB0]);
            prior_pair(k0, k0);
            const double val = cx.score[k0] + pr.neglogw[k0];
            // lane g tests component g (every wave for itself, same data: a uniform verdict without a barrier)
            const int gl = min(lane, G - 1);
            const double lb = fmax(0.0, cx.ps0[gl] - pr.cnorm[gl] * dist);
            const double LB = lb * lb + pr.neglogw[gl];
            const bool open_ = lane < G && lane != k0 && !(val + 1e-9 * (fabs(val) + fabs(LB) + 1.0) < LB);
            settled = __ballot(open_) == 0ull;
            kb = k0; prior_ss = val;
        }
        PROF_LAP(42);
        if (settled) PROF_COUNT(28);
        if (!settled) {
            PROF_COUNT(29);
            for (int ga = 0; ga < G; ga += 2) prior_pair(ga, min(ga + 1, G - 1));
            // argmin (lowest index on ties): lane g holds component g's total, every thread walks them by readlane
            const double sc = cx.score[min(lane, G - 1)] + pr.neglogw[min(lane, G - 1)];
            kb = 0;
            double best = readlane_f64(sc, 0);
            for (int gc = 1; gc < min(G, 64); ++gc) {
                const double v = readlane_f64(sc, gc);
                if (v < best) { best = v; kb = gc; }
            }
            prior_ss = best;
            // new reference: this point (the barriers of the evaluations above separate these writes from the reads of the test)
            for (int b = tid; b < np_; b += MOSHII_TPB) cx.px0[b] = cx.xb[b];
            if (tid < G) cx.ps0[tid] = sqrt(cx.score[tid]);
            if (tid == 0) { cx.scal[S_PRIOR_REF] = 1.0; cx.scal[S_PRIOR_KB0] = (double)kb; }
        }
        if (tid == 0) { cx.scal[S_KBEST] = (double)kb; cx.scal[S_PRIOR_SS] = prior_ss; }   // (read by assemble() / a light evaluation, many barriers later)
        PROF_LAP(43);
    }
    block_sum3(sd, sv, sh, cx.red);
    if constexpr (XT) block_sum3(sf, ss, sy, cx.red);
    if constexpr (COOP) {
        PROF_LAP(3);
        // the ranks' data sums of squares, added in rank order by every rank; the prior's value (and argmin) from the rank that has it
        const unsigned seq = coop_begin(cx);
        TRACE_STAMP(co, seq, 0);
        const double mine[3] = {sd, prior_ss, (np_ > 0) ? cx.scal[S_KBEST] : 0.0};
        double* land = cx.y;   // (free during an evaluation: the solve's reciprocal pivots / the assembly's diagonal terms are used up)
        sd = 0.0;
        if (coop_exchange_small<3>(co, cx, seq, mine, land)) {
            for (int r = 0; r < co.G; ++r) sd += land[3 * r];
            if (np_ > 0) {
                prior_ss = land[3 * co.prior_rank + 1];
                const double kbv = land[3 * co.prior_rank + 2];
                if (tid == 0 && co.rank != co.prior_rank) { cx.scal[S_PRIOR_SS] = prior_ss; cx.scal[S_KBEST] = kbv; }   // (what a light evaluation picks up)
            }
        }
        TRACE_STAMP(co, seq, 3);
        coop_end(cx, seq);
        PROF_LAP(31);
    } else {
        PROF_LAP(3);
    }
    Sse out;
    out.data = sd; out.velo = sv; out.hand = sh;
    out.face = sf; out.shape = ss; out.stay = sy;
    out.prior = (np_ > 0) ? fp.wt_pose * fp.wt_pose * prior_ss : 0.0;
    out.total = ((out.data + out.prior) + out.velo) + out.hand;
    if constexpr (XT) out.total += (sf + ss) + sy;
    return out;
}

// ------------------------------------------------------------------------------------------------
// J^T J on the matrix pipe.  The lower-triangle 16x16 tiles (bi, bj), numbered e = bi (bi + 1) / 2 + bj, are dealt to the
// four wavefronts round-robin (tile e belongs to wave e % 4); a wave accumulates its tiles with v_mfma_f64_16x16x4_f64
// over groups of four Jacobian rows: lane l supplies J[r0 + (l >> 4)][16 b + (l & 15)] -- the SAME fragment serves as the
// A operand of block-row b and as the B operand of block-column b, so a row group costs NBLK 8-byte LDS reads per lane
// for up to ceil(NE / 4) MFMAs (the former register-tile form read 2 NBLK operands per row per thread for NE fmas and ran
// at LDS speed).  Result layout (f64 MFMA): lane l, register i holds C[(l >> 4) + 4 i][l & 15].
// ------------------------------------------------------------------------------------------------
typedef double v4d __attribute__((vector_size(32)));

constexpr int tile_bi(int e) { int bi = 0; while ((bi + 1) * (bi + 2) / 2 <= e) ++bi; return bi; }
constexpr int tile_bj(int e) { return e - tile_bi(e) * (tile_bi(e) + 1) / 2; }

template <int NBLK>
struct JtJAcc {
    static constexpr int NE = NBLK * (NBLK + 1) / 2;
    static constexpr int NT = (NE + 3) / 4;      // tiles per wave
    static constexpr int XR = 16;                // tiles per exchange round (XR x 256 doubles of LDS)
    v4d c[NT];
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int t = 0; t < NT; ++t) c[t] = v4d{0.0, 0.0, 0.0, 0.0};
    }
    template <int W, int T>
    __device__ __forceinline__ void mm1(const double (&f)[NBLK]) {
        constexpr int e = 4 * T + W;
        if constexpr (e < NE) {
            constexpr int bi = tile_bi(e), bj = tile_bj(e);
            c[T] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[bi], f[bj], c[T], 0, 0, 0);
        }
    }
    template <int W, int... T>
    __device__ __forceinline__ void mm(const double (&f)[NBLK], std::integer_sequence<int, T...>) { (mm1<W, T>(f), ...); }
    // c += rows^T rows over this wave's tiles (rows: nr x LDJ in LDS; rows >= nr of the last group count as zero)
    template <int W>
    __device__ __forceinline__ void accumulate_w(const double* rows, int nr, int LDJ) {
        const int lane = threadIdx.x & 63, kr = lane >> 4, cc = lane & 15;
        for (int r0 = 0; r0 < nr; r0 += 8) {   // two row groups per trip: 2 NBLK reads in flight
            double f0[NBLK], f1[NBLK];
            const int ra = r0 + kr, rb = r0 + 4 + kr;
            const double* pa = rows + min(ra, nr - 1) * LDJ + cc;
            const double* pb = rows + min(rb, nr - 1) * LDJ + cc;
#pragma unroll
            for (int b = 0; b < NBLK; ++b) { f0[b] = pa[16 * b]; f1[b] = pb[16 * b]; }
            const double la = (ra < nr) ? 1.0 : 0.0, lb = (rb < nr) ? 1.0 : 0.0;   // (the clamped re-read of the last row must not count)
#pragma unroll
            for (int b = 0; b < NBLK; ++b) { f0[b] *= la; f1[b] *= lb; }
            mm<W>(f0, std::make_integer_sequence<int, NT>());
            mm<W>(f1, std::make_integer_sequence<int, NT>());
        }
    }
    __device__ __forceinline__ void accumulate(const double* rows, int nr, int LDJ) {
        switch (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) {   // wave-uniform: the tile indices of a wave are compile-time constants in its branch
            case 0: accumulate_w<0>(rows, nr, LDJ); break;
            case 1: accumulate_w<1>(rows, nr, LDJ); break;
            case 2: accumulate_w<2>(rows, nr, LDJ); break;
            default: accumulate_w<3>(rows, nr, LDJ); break;
        }
    }
};

// ------------------------------------------------------------------------------------------------
// J^T J in registers: thread (ty, tx) of a 16x16 grid owns A[bi*16+ty][bj*16+tx] for bj <= bi.
// (Tried for the 13-block solve, whose 91 entries per thread are 182 registers held across the whole dogleg loop of a kernel that spills
//  690 vector and 1067 scalar registers: the matrix parked in the chain's global scratch between its uses -- stored by the assembly, read
//  back by the two quadratic forms of an iteration and by the factorisation's block columns.  Spills 690 -> 566 only, the iteration 394 ->
//  431 us: the block columns' reads from the L2 sit on the factorisation's critical path.  The registers stay.)
// ------------------------------------------------------------------------------------------------
template <int NBLK>
struct AReg {
    static constexpr int NE = NBLK * (NBLK + 1) / 2;
    double a[NE];
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int e = 0; e < NE; ++e) a[e] = 0.0;
    }
    // a = the tiles of acc, re-dealt to the interleaved ownership above.  Register i of lane l of the owning wave holds entry
    // ((l >> 4) + 4 i, l & 15) of its tile, which is thread (wave i, lane l)'s entry here: the exchange is one 8-byte LDS word
    // per (tile, register, lane), written and read conflict-free.  X: XR x 256 doubles of LDS nobody else uses meanwhile
    // (callers: the Jacobian tile region after its last use); all threads call.
    __device__ __forceinline__ void take(const JtJAcc<NBLK>& acc, double* X) {
        const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
        constexpr int XR = JtJAcc<NBLK>::XR;
#pragma unroll
        for (int e0 = 0; e0 < NE; e0 += XR) {
            if (e0 > 0) __syncthreads();   // the previous round's reads are done
#pragma unroll
            for (int tt = 0; tt < XR / 4; ++tt) {
                const int t = e0 / 4 + tt;
                if (t < JtJAcc<NBLK>::NT) {
                    const int e = 4 * t + w;
                    if (e < NE) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) X[(e - e0) * 256 + i * 64 + lane] = acc.c[t][i];
                    }
                }
            }
            __syncthreads();
#pragma unroll
            for (int e = e0; e < e0 + XR; ++e)
                if (e < NE) a[e] = X[(e - e0) * 256 + w * 64 + lane];
        }
    }
    __device__ __forceinline__ void add_diag(const double* dvec /* LDS [n] */, int n) {
        const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
        if (ty != tx) return;
        int e = 0;
#pragma unroll
        for (int bi = 0; bi < NBLK; ++bi)
#pragma unroll
            for (int bj = 0; bj <= bi; ++bj) {
                if (bi == bj) { const int q = bi * 16 + ty; if (q < n) a[e] += dvec[q]; }
                ++e;
            }
    }
    // A[q1][q2] += scale * Pk[colprior[q1]][colprior[q2]]   (branch-free: clamped gathers, 0/1 factor)
    __device__ __forceinline__ void add_prior(double scale, MOSHII_GP(const double) Pk, int np_, const int* colprior, int n) {
        const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
        int pr_[NBLK], pc_[NBLK];
#pragma unroll
        for (int b = 0; b < NBLK; ++b) {
            const int q1 = b * 16 + ty, q2 = b * 16 + tx;
            const int p1 = colprior[min(q1, n - 1)], p2 = colprior[min(q2, n - 1)];
            pr_[b] = (q1 < n) ? p1 : -1;
            pc_[b] = (q2 < n) ? p2 : -1;
        }
        double v[NE];
        int e = 0;
#pragma unroll
        for (int bi = 0; bi < NBLK; ++bi)
#pragma unroll
            for (int bj = 0; bj <= bi; ++bj) { v[e] = Pk[max(pr_[bi], 0) * np_ + max(pc_[bj], 0)]; ++e; }
        e = 0;
#pragma unroll
        for (int bi = 0; bi < NBLK; ++bi)
#pragma unroll
            for (int bj = 0; bj <= bi; ++bj) { a[e] = fma(scale * ((pr_[bi] >= 0 && pc_[bj] >= 0) ? 1.0 : 0.0), v[e], a[e]); ++e; }
    }
    // sum_{q1,q2} A[q1][q2] x[q1] x[q2] over the full symmetric matrix (block partial; reduce outside).  Branch-free: the
    // 2 NBLK vector entries are fetched up front (zero beyond n), off-diagonal blocks count twice, diagonal blocks by ty/tx.
    __device__ __forceinline__ double quad_partial(const double* x, int n, int tid_) const {
        const int ty = tid_ >> 4, tx = tid_ & 15;
        double xr[NBLK], xc[NBLK];
#pragma unroll
        for (int b = 0; b < NBLK; ++b) {
            const int q1 = b * 16 + ty, q2 = b * 16 + tx;
            const double v1 = x[min(q1, n - 1)], v2 = x[min(q2, n - 1)];
            xr[b] = v1 * ((q1 < n) ? 1.0 : 0.0);   // (0/1 factors keep the reads unconditional)
            xc[b] = v2 * ((q2 < n) ? 1.0 : 0.0);
        }
        const double cd = (tx < ty) ? 2.0 : ((tx == ty) ? 1.0 : 0.0);
        double s = 0.0;
        int e = 0;
#pragma unroll
        for (int bi = 0; bi < NBLK; ++bi) {
            double r = 0.0;
#pragma unroll
            for (int bj = 0; bj <= bi; ++bj) { r += ((bj == bi) ? cd : 2.0) * a[e] * xc[bj]; ++e; }
            s += xr[bi] * r;
        }
        return s;
    }
};

// Solve A d = g by a right-looking L D L^T elimination carried out in REGISTERS: every thread keeps a working copy
// of its 16x16-interleaved entries of [A; g^T] (the right-hand side rides along as row n), and per column j
//   owners of column j publish c_ij = l_ij d_j (their current entries) to the packed factor in LDS,
//   one barrier, then every thread reads the <= 2 NBLK column entries it needs and updates its own registers
//   with c_ij c_kj / d_j.
// No thread ever reads an LDS word that is written in the same step (column j+1 has its own storage), so ONE
// workgroup barrier per column suffices; no square roots.  Wave 0 then back-substitutes
// x_j = (b_j - sum_{i>j} c_ij x_i) / d_j from the packed factor.  A itself is left untouched (the dogleg needs
// d^T A d afterwards).  Returns false on a non-positive pivot (the reference would fall back to lstsq; callers
// take the Cauchy step).
// NOT inlined, and handed LDS offsets instead of pointers: inside the frame loop this code shared the register file with
// the whole solver state -- the elimination step then reloaded spilled scalars from scratch memory (a vector-memory round
// trip, several per step, in the one loop of the kernel whose cost is pure latency).  As a function of its own it is
// allocated with nothing live but its arguments.
// (Two functions, elimination and back-substitution: as one, hipcc's register allocator crashes on it under the
// iterative-ILP scheduler the rest of this file is built with.)
// How a matrix crosses a function boundary.  A struct of more than 64 bytes is passed and returned through the stack --
// scratch memory, a store / load round trip at every call; a VECTOR of the same doubles travels in registers.  Up to 16
// entries per thread (NBLK <= 5) go as a vector, larger matrices as the struct (the emulation build, compiled by g++, has
// no such vector types: struct everywhere).
// assemble_fn's RESULT nevertheless goes back as the struct.  Returned as a vector, the callee keeps more of its state in the
// callee-saved half of the register file and saves / restores ~110 registers per lane to scratch at every call: alone on the
// GPU a chain is 1.5 % faster that way, but with a chain on every CU the extra scratch footprint (32 CUs share 4 MB of L2)
// costs 8 % (32-sequence leg 635 k -> 685 k frames/s with the struct return; the single-sequence bench is unchanged).
#ifdef MOSHII_ASM_VECRET
#define MOSHII_ASM_RET_VEC(NBLK) ((NBLK) * ((NBLK) + 1) / 2 <= 16)
#else
#define MOSHII_ASM_RET_VEC(NBLK) false
#endif
template <int NBLK, bool VEC = (NBLK * (NBLK + 1) / 2 <= 16)>
struct APass {
    typedef AReg<NBLK> type;
    static __device__ __forceinline__ type pack(const AReg<NBLK>& A) { return A; }
    static __device__ __forceinline__ AReg<NBLK> unpack(const type& v) { return v; }
};
#if defined(__clang__)
template <int NBLK>
struct APass<NBLK, true> {
    typedef double type __attribute__((ext_vector_type(NBLK * (NBLK + 1) / 2)));
    static __device__ __forceinline__ type pack(const AReg<NBLK>& A) {
        type v;
#pragma unroll
        for (int e = 0; e < AReg<NBLK>::NE; ++e) v[e] = A.a[e];
        return v;
    }
    static __device__ __forceinline__ AReg<NBLK> unpack(const type v) {
        AReg<NBLK> A;
#pragma unroll
        for (int e = 0; e < AReg<NBLK>::NE; ++e) A.a[e] = v[e];
        return A;
    }
};
#endif

// Elimination by 16-column PANELS (same arithmetic, operation for operation, as the round-1 column-pair elimination it replaced
// -- every entry receives w_ik = fma(-c_ij, c_kj pin_j, w_ik) for j ascending -- so the factor,
// the reciprocal pivots and the solver's trajectory have the same bits; what changes is who executes it).  The kernel's
// time is instruction count (one wavefront per SIMD: ~5 cycles an instruction, measured -- tools/ubench_latency.hip); the
// pair elimination spent ~150 instructions of EVERY wavefront and an LDS round trip per two columns.  Here, per panel:
//   A  the owners publish the panel's current entries into the packed factor (where they end up anyway),
//   B  wavefront 0 alone eliminates the 16 columns inside the panel: lane r holds row c0 + r (and row c0 + 64 + r when
//      the matrix has more than 64 rows) in registers, the pivot and the normalised pivot-column entries travel by
//      v_readlane -- no LDS, no barrier -- three instructions per (column, later column) pair,
//   C  everyone applies the panel to the tiles to its right: 16 rank-1 updates per tile from LDS reads that are broadcasts
//      (row entries) or spread over the banks (column entries), the next panel's entries included.
// Two barriers per 16 columns instead of eight.
// Up to four register blocks (the body solve) the factor is stored SQUARE -- row i at i (16 NBLK + 1), zeros right of the diagonal --
// instead of packed: twice the LDS (33 KB of a region that holds the Jacobian tiles otherwise), and every address in the elimination and
// the back-substitution is a row base plus a compile-time offset: no per-entry select of a spare word, no offset arithmetic per row.
template <int NBLK> constexpr bool ldl_square() { return NBLK <= 4; }
template <int NBLK>
struct LdlCtx {
    static constexpr bool SQ = ldl_square<NBLK>();
    static constexpr int LS = NBLK * 16 + 1;   // row stride of the square form (odd: a lane-per-row access spreads over the banks)
    double* Lp; double* pinv; double* Zr;
    int n, trash, zero, ty, tx, lane;
    int rS[NBLK], cS[NBLK];   // packed offsets of this thread's rows b 16 + ty / b 16 + tx (-1: beyond the border row)
};
// B: wavefront 0 eliminates the columns of panel P inside the panel.  FULL: all 16 columns exist (every panel but possibly
// the last) -- no per-column test, one basic block, in which the scheduler overlaps a column's updates with the next
// column's pivot chain.  Returns "a pivot was not positive".
template <int NBLK, int P, bool FULL>
__device__ __forceinline__ bool ldl_panel_eliminate(const LdlCtx<NBLK>& c) {
    constexpr int NS = (NBLK * 16 + 63) / 64;      // rows per lane of wavefront 0
    constexpr int c0 = 16 * P;
    double a[NS][16];
    int off[NS];
#pragma unroll
    for (int s_ = 0; s_ < NS; ++s_) {
        const int row = c0 + 64 * s_ + c.lane;
        if constexpr (LdlCtx<NBLK>::SQ) {
            off[s_] = (row <= c.n) ? row * LdlCtx<NBLK>::LS + c0 : -1;
            const double* rp = c.Lp + ((row < NBLK * 16) ? row * LdlCtx<NBLK>::LS + c0 : c.zero);   // (zeros right of the diagonal: stored)
            const int st = (row < NBLK * 16) ? 1 : 0;
#pragma unroll
            for (int k = 0; k < 16; ++k) a[s_][k] = rp[k * st];
        } else {
            off[s_] = (row <= c.n) ? row * (row + 1) / 2 + c0 : -1;
#pragma unroll
            for (int k = 0; k < 16; ++k) a[s_][k] = c.Lp[(off[s_] >= 0 && c0 + k <= row) ? off[s_] + k : c.zero];
        }
    }
    bool bad = false;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        if (FULL || c0 + j < c.n) {   // (uniform: the border row's own "column" is not eliminated)
            const double pj = readlane_f64(a[0][j], j);
            bad = bad || !(pj > 0.0);
            // 1 / pivot: hardware reciprocal + two Newton steps (pivot is positive and normal), shorter than the IEEE divide
            double pin = __builtin_amdgcn_rcp(pj);
            pin = fma(fma(-pj, pin, 1.0), pin, pin);
            pin = fma(fma(-pj, pin, 1.0), pin, pin);
            *((c.lane == 0) ? c.pinv + c0 + j : c.Lp + c.trash) = pin;
            const double lj = a[0][j] * pin;   // lane k: c_kj pin_j of row c0 + k
            // (all the later columns' multipliers first, then the updates: taken one at a time the compiler funnels every pair of
            //  v_readlane through the same scalar pair with a wait state before each fma -- 1.7 % of a frame; the same entries through
            //  LDS broadcasts instead of v_readlane time the same)
            double lkv[16];
#pragma unroll
            for (int k = j + 1; k < 16; ++k) lkv[k] = readlane_f64(lj, k);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = j + 1; k < 16; ++k)
#pragma unroll
                for (int s_ = 0; s_ < NS; ++s_) a[s_][k] = fma(-a[s_][j], lkv[k], a[s_][k]);
        }
    }
#pragma unroll
    for (int s_ = 0; s_ < NS; ++s_) {
        const int row = c0 + 64 * s_ + c.lane;
#pragma unroll
        for (int k = 0; k < 16; ++k) c.Lp[(off[s_] >= 0 && c0 + k <= row) ? off[s_] + k : c.trash] = a[s_][k];
    }
    return bad;
}
// B for the square form (<= 4 register blocks): the diagonal block's 16 rows sit in ALL FOUR rows of 16 lanes of wavefront 0 (lane l: row
// c0 + l % 16) beside the up to 48 rows below it (lane l: row c0 + 16 + l), so a pivot-column multiplier reaches every lane through a
// row broadcast inside the fma -- v_fmac_f64_dpp ... row_newbcast:k -- instead of two v_readlane and an fma per (column, later column)
// pair: ~500 instructions a panel instead of ~710 (~340 on the last, which has no rows below).  The same products and sums in the same
// order: the same bits.
template <int J, int K, bool FIRST = true>
__device__ __forceinline__ void ldl_rows_update(double (&a)[16], double lj) {   // a_k -= a_J l_kJ for k = K .. 15
    if constexpr (K < 16) {
        fmac_rowbcast<K, FIRST>(a[K], lj, a[J]);
        ldl_rows_update<J, K + 1, false>(a, lj);
    }
}
template <int NBLK, int P, bool FULL, int J>
__device__ __forceinline__ void ldl_rows_column(const LdlCtx<NBLK>& c, double* pinp, double (&ad)[16], double (&ab)[16], bool& bad) {
    constexpr int c0 = 16 * P;
    constexpr bool BELOW = c0 + 16 < NBLK * 16;
    if (FULL || c0 + J < c.n) {   // (uniform: the border row's own "column" is not eliminated)
        const double pj = rowbcast_f64<J>(ad[J]);
        bad = bad || !(pj > 0.0);
        double pin = __builtin_amdgcn_rcp(pj);
        pin = fma(fma(-pj, pin, 1.0), pin, pin);
        pin = fma(fma(-pj, pin, 1.0), pin, pin);
        pinp[J] = pin;   // (lane 0: pinv[c0 + J]; the others: their parking words)
        const double lj = ad[J] * pin;   // lane l: c_kj pin_j of diagonal row k = l % 16
        ldl_rows_update<J, J + 1>(ad, lj);
        if constexpr (BELOW) ldl_rows_update<J, J + 1>(ab, lj);
    }
    if constexpr (J + 1 < 16) ldl_rows_column<NBLK, P, FULL, J + 1>(c, pinp, ad, ab, bad);
}
// The same elimination for a panel whose 16 columns all exist, SOFTWARE-PIPELINED by hand.  Column J's work is (S) the pivot chain -- pivot
// out of row J, reciprocal, two Newton steps, the multipliers l_J: eight instructions, each waiting for the one before -- and (U) up to 29
// independent row updates that read l_J.  Only the FIRST update (row J + 1 of the diagonal block) feeds column J + 1's pivot; as statements
// in column order the wave nevertheless issued all of U(J) before S(J + 1) began (the updates are asm statements, which keep their order)
// and then sat through S(J + 1)'s latencies with nothing else to issue: ~100 + 8 (29 - 2 J) cycles a column.  Here column J + 1's chain
// starts right behind that first update and its eight instructions are dealt between seven chunks of the remaining updates of column J
// (sched_barrier pins the order), so a column costs the LONGER of the two instead of their sum.  Every register still receives the same
// products in the same order: the same bits.
template <int J, int LO, int HI, bool BELOW>
__device__ __forceinline__ void ldl_rows_rest(double (&ad)[16], double (&ab)[16], double lj) {   // items LO .. HI - 1 of U(J) without its first
    if constexpr (LO < HI) {
        constexpr int NAD = 14 - J;   // rows J + 2 .. 15 of the diagonal block, then rows J + 1 .. 15 of the block below
        if constexpr (LO < NAD) fmac_rowbcast<J + 2 + LO, false>(ad[J + 2 + LO], lj, ad[J]);
        else if constexpr (BELOW) fmac_rowbcast<J + 1 + (LO - NAD), false>(ab[J + 1 + (LO - NAD)], lj, ab[J]);
        ldl_rows_rest<J, LO + 1, HI, BELOW>(ad, ab, lj);
    }
}
template <bool BELOW, int J>
__device__ __forceinline__ void ldl_rows_pipe(double* pinp, double (&ad)[16], double (&ab)[16], double lj, bool& bad) {
    if constexpr (J + 1 < 16) {
        constexpr int NI = (14 - J) + (BELOW ? 15 - J : 0), CH = (NI + 6) / 7;
        constexpr int c1 = (CH < NI) ? CH : NI, c2 = (2 * CH < NI) ? 2 * CH : NI, c3 = (3 * CH < NI) ? 3 * CH : NI, c4 = (4 * CH < NI) ? 4 * CH : NI,
                      c5 = (5 * CH < NI) ? 5 * CH : NI, c6 = (6 * CH < NI) ? 6 * CH : NI;
        fmac_rowbcast<J + 1, true>(ad[J + 1], lj, ad[J]);   // row J + 1 of the diagonal block is final: column J + 1's chain can start
        const double pj = rowbcast_f64<J + 1>(ad[J + 1]);
        bad = bad || !(pj > 0.0);
        ldl_rows_rest<J, 0, c1, BELOW>(ad, ab, lj);
        __builtin_amdgcn_sched_barrier(0);
        double pin = __builtin_amdgcn_rcp(pj);
        ldl_rows_rest<J, c1, c2, BELOW>(ad, ab, lj);
        __builtin_amdgcn_sched_barrier(0);
        double er = fma(-pj, pin, 1.0);
        ldl_rows_rest<J, c2, c3, BELOW>(ad, ab, lj);
        __builtin_amdgcn_sched_barrier(0);
        pin = fma(er, pin, pin);
        ldl_rows_rest<J, c3, c4, BELOW>(ad, ab, lj);
        __builtin_amdgcn_sched_barrier(0);
        er = fma(-pj, pin, 1.0);
        ldl_rows_rest<J, c4, c5, BELOW>(ad, ab, lj);
        __builtin_amdgcn_sched_barrier(0);
        pin = fma(er, pin, pin);
        ldl_rows_rest<J, c5, c6, BELOW>(ad, ab, lj);
        __builtin_amdgcn_sched_barrier(0);
        pinp[J + 1] = pin;
        const double ljn = ad[J + 1] * pin;
        ldl_rows_rest<J, c6, NI, BELOW>(ad, ab, lj);
        __builtin_amdgcn_sched_barrier(0);
        ldl_rows_pipe<BELOW, J + 1>(pinp, ad, ab, ljn, bad);
    }
}
template <int NBLK, int P, bool FULL>
__device__ __forceinline__ bool ldl_panel_eliminate_rows(const LdlCtx<NBLK>& c) {
    constexpr int c0 = 16 * P, LS = LdlCtx<NBLK>::LS, LD = NBLK * 16;
    constexpr bool BELOW = c0 + 16 < LD;
    const int rd = c0 + (c.lane & 15), rb = c0 + 16 + c.lane;
    double ad[16], ab[16];
    {
        const double* pd = c.Lp + rd * LS + c0;
        const double* pb = c.Lp + ((BELOW && rb < LD) ? rb * LS + c0 : c.zero);
        const int sb = (BELOW && rb < LD) ? 1 : 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) { ad[k] = pd[k]; ab[k] = BELOW ? pb[k * sb] : 0.0; }
    }
    bool bad = false;
    double* const pinp = (c.lane == 0) ? c.pinv + c0 : c.Lp + c.trash;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MOSHII_LDL_NO_PIPE)
    if constexpr (FULL) {
        const double pj = rowbcast_f64<0>(ad[0]);
        bad = !(pj > 0.0);
        double pin = __builtin_amdgcn_rcp(pj);
        pin = fma(fma(-pj, pin, 1.0), pin, pin);
        pin = fma(fma(-pj, pin, 1.0), pin, pin);
        pinp[0] = pin;
        ldl_rows_pipe<BELOW, 0>(pinp, ad, ab, ad[0] * pin, bad);
    } else
#endif
    ldl_rows_column<NBLK, P, FULL, 0>(c, pinp, ad, ab, bad);
    {   // the factor's entries: the diagonal block's rows from the first row of lanes (on and left of the diagonal, rows up to the border row) ...
        const int offd = rd * LS + c0;
#pragma unroll
        for (int k = 0; k < 16; ++k) c.Lp[(c.lane < 16 && rd <= c.n && k <= (c.lane & 15)) ? offd + k : c.trash] = ad[k];
        if constexpr (BELOW) {   // ... and the rows below whole
            double* const pb = c.Lp + ((rb <= c.n) ? rb * LS + c0 : c.trash);
#pragma unroll
            for (int k = 0; k < 16; ++k) pb[k] = ab[k];
        }
    }
    return bad;
}
// C: panel P applied to every tile to its right
template <int NBLK, int P, bool FULL>
__device__ __forceinline__ void ldl_panel_apply(const LdlCtx<NBLK>& c, double (&w)[NBLK * (NBLK + 1) / 2]) {
    constexpr int c0 = 16 * P;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        if (FULL || c0 + j < c.n) {
            const double pin = c.pinv[c0 + j];
            double ci[NBLK], ck[NBLK];
#pragma unroll
            for (int b = P + 1; b < NBLK; ++b) {
                if constexpr (LdlCtx<NBLK>::SQ) {   // (every row of the square exists: rows beyond the border row hold zeros)
                    ci[b] = c.Lp[c.rS[b] + c0 + j];
                    ck[b] = c.Lp[c.cS[b] + c0 + j] * pin;
                } else {
                    ci[b] = c.Lp[(c.rS[b] >= 0) ? c.rS[b] + c0 + j : c.zero];
                    ck[b] = c.Lp[(c.cS[b] >= 0) ? c.cS[b] + c0 + j : c.zero] * pin;
                }
            }
#pragma unroll
            for (int bi = P + 1; bi < NBLK; ++bi)
#pragma unroll
                for (int bj = P + 1; bj <= bi; ++bj) w[bi * (bi + 1) / 2 + bj] = fma(-ci[bi], ck[bj], w[bi * (bi + 1) / 2 + bj]);
        }
    }
}
template <int NBLK, int P>
__device__ __forceinline__ bool ldl_panels(const LdlCtx<NBLK>& c, double (&w)[NBLK * (NBLK + 1) / 2]) {
    constexpr int c0 = 16 * P;
    if (c0 >= c.n) return true;   // (uniform)
    // A: the panel's entries as they stand (lower triangle of the diagonal tile, the tiles below it whole)
#pragma unroll
    for (int bi = P; bi < NBLK; ++bi) {
        const int q1 = bi * 16 + c.ty, q2 = c0 + c.tx;
        if constexpr (LdlCtx<NBLK>::SQ) c.Lp[c.rS[bi] + q2] = (q2 <= q1 && q1 <= c.n) ? w[bi * (bi + 1) / 2 + P] : 0.0;   // (zeros above the diagonal and below the border row)
        else c.Lp[(c.rS[bi] >= 0 && q2 <= q1) ? c.rS[bi] + q2 : c.trash] = w[bi * (bi + 1) / 2 + P];
    }
    __syncthreads();
    PROF_LAP_EXT(46);
    if (threadIdx.x < 64) {
        bool bad;
        if constexpr (LdlCtx<NBLK>::SQ && NBLK == 4) bad = (c0 + 16 <= c.n) ? ldl_panel_eliminate_rows<NBLK, P, true>(c) : ldl_panel_eliminate_rows<NBLK, P, false>(c);
        else bad = (c0 + 16 <= c.n) ? ldl_panel_eliminate<NBLK, P, true>(c) : ldl_panel_eliminate<NBLK, P, false>(c);
        if (c.lane == 0 && bad) c.Zr[1] = 1.0;
    }
    __syncthreads();
    PROF_LAP_EXT(44);
    if (c.Zr[1] != 0.0) return false;   // a non-positive pivot (every thread sees it): the caller takes the Cauchy step
    if constexpr (P + 1 < NBLK) {
        if (c0 + 16 <= c.n) ldl_panel_apply<NBLK, P, true>(c, w); else ldl_panel_apply<NBLK, P, false>(c, w);
        PROF_LAP_EXT(45);
        return ldl_panels<NBLK, P + 1>(c, w);
    }
    return true;
}

template <int NBLK>
__device__ __noinline__ bool ldl_factor(const typename APass<NBLK>::type Av, int o_Lp_, int o_g_, int o_pinv_, int n_) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const AReg<NBLK> A = APass<NBLK>::unpack(Av);
    // (function arguments arrive in vector registers: as scalars the loop bounds and LDS offsets below are SALU work)
    const int o_Lp = __builtin_amdgcn_readfirstlane(o_Lp_), o_g = __builtin_amdgcn_readfirstlane(o_g_);
    const int o_pinv = __builtin_amdgcn_readfirstlane(o_pinv_), n = __builtin_amdgcn_readfirstlane(n_);
    const double* const g = lds + o_g;
    const int tid = threadIdx.x;
    PROF_BEGIN(); PROF_COUNT(22);
    LdlCtx<NBLK> c;
    c.Lp = lds + o_Lp; c.pinv = lds + o_pinv; c.n = n;
    c.ty = tid >> 4; c.tx = tid & 15; c.lane = tid & 63;
    double w[AReg<NBLK>::NE];
    {
        int e = 0;
#pragma unroll
        for (int bi = 0; bi < NBLK; ++bi)
#pragma unroll
            for (int bj = 0; bj <= bi; ++bj) {
                const int q1 = bi * 16 + c.ty, q2 = bj * 16 + c.tx;
                const double gq = g[min(q2, n - 1)] * ((q2 < n) ? 1.0 : 0.0);   // (unconditional read)
                w[e] = (q1 == n) ? gq : A.a[e];
                ++e;
            }
    }
    // Lp: the packed factor, entry (i, j), j <= i, at i (i + 1) / 2 + j; row n is the right-hand side.  Behind it: 64 per-lane
    // trash words (stores that do not apply), a zero word (loads that do not apply), the panels' verdict.
    constexpr bool SQ = LdlCtx<NBLK>::SQ;
    constexpr int LS = LdlCtx<NBLK>::LS;
    const int fend = SQ ? NBLK * 16 * LS : (n + 1) * (n + 2) / 2;   // the factor's words
    c.trash = fend + c.lane; c.zero = fend + (SQ ? 80 : 64);   // (square form: 16 words more, a parked lane's 16-entry row store lands at trash .. trash + 15)
    c.Zr = c.Lp + c.zero;
#pragma unroll
    for (int b = 0; b < NBLK; ++b) {
        const int q1 = b * 16 + c.ty, q2 = b * 16 + c.tx;
        c.rS[b] = SQ ? q1 * LS : ((q1 <= n) ? q1 * (q1 + 1) / 2 : -1);
        c.cS[b] = SQ ? q2 * LS : ((q2 <= n) ? q2 * (q2 + 1) / 2 : -1);
    }
    if constexpr (SQ) {   // the tiles right of the diagonal tiles: zeros (the back-substitution reads whole rows)
#pragma unroll
        for (int bi = 0; bi < NBLK; ++bi)
#pragma unroll
            for (int bj = bi + 1; bj < NBLK; ++bj) c.Lp[c.rS[bi] + bj * 16 + c.tx] = 0.0;
    }
    if (tid == 0) { c.Zr[0] = 0.0; c.Zr[1] = 0.0; }
    PROF_MARK();
    const bool ok = ldl_panels<NBLK, 0>(c, w);
    __syncthreads();
    PROF_LAP(9);
    return ok;
}


template <int NBLK>
__device__ __noinline__ void ldl_backsub(int o_Lp_, int o_d_, int o_pinv_, int n_) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int o_Lp = __builtin_amdgcn_readfirstlane(o_Lp_), o_d = __builtin_amdgcn_readfirstlane(o_d_);
    const int o_pinv = __builtin_amdgcn_readfirstlane(o_pinv_), n = __builtin_amdgcn_readfirstlane(n_);
    const double* const Lp = lds + o_Lp;
    double* const d = lds + o_d;
    const double* const pinv = lds + o_pinv;
    const int tid = threadIdx.x;
    constexpr bool SQ = ldl_square<NBLK>();
    constexpr int LS = LdlCtx<NBLK>::LS;
    const int zero = SQ ? NBLK * 16 * LS + 80 : (n + 1) * (n + 2) / 2 + 64;
    PROF_BEGIN();
    // back substitution by wave 0: lane l owns unknowns l and l+64; the solved x_j is broadcast with v_readlane.
    // Rows of the factor and 1/d_j are fetched a whole group of U steps ahead (an LDS round trip is several times the
    // readlane -> multiply -> fma chain of a step); rows are read branch-free (lanes beyond the row read the zero word).
    // A step is readlane, multiply, fma and nothing else: no lane is ever overwritten with "its" solution inside the loop (the entries
    // of row j on and beyond the diagonal are zeros, so y_j stays what it was when step j broadcast it) -- every lane multiplies its
    // y by its own 1/d once, after the loop: the same product the step formed.  (Round 3's form selected lane j per step through the
    // exec mask and branched on j >= 0: 178 cycles a step.)  Steps below row 0 of the last group run with 1/d = 0: no effect.
    PROF_T(_tb0);
    if constexpr (SQ) {
        // Square factor: row j at a compile-time offset from the lane's base address, zeros on and right of the diagonal stored --
        // a step is multiply, readlane, read, fma with nothing to compute about addresses.  Every row 16 NBLK - 2 .. 1 is stepped
        // through: rows at and beyond the border row meet y = 0 in their lane (their products are zero: the border row's entries
        // and the zero rows behind it are finite), so n needs no test.
        if (tid < 64) {
            const double* row0 = Lp + tid;
            if (NBLK * 16 >= 64 || tid < NBLK * 16) lds[o_Lp + tid * LS + tid] = 0.0;   // (the diagonal holds the pivots: lane j's own entry of row j must not touch y_j; same wavefront: ordered)
            double y0 = Lp[(tid < n) ? n * LS + tid : zero];
            const double pl0 = pinv[min(tid, max(n - 1, 0))];
            // (all rows first: 16 NBLK - 2 reads whose only cost is their issue slots -- left to the scheduler they sit two steps ahead of
            //  their use and every other step waits out an LDS round trip)
            double L[NBLK * 16 - 1];
#pragma unroll
            for (int j = NBLK * 16 - 2; j >= 1; --j) L[j] = row0[j * LS];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = NBLK * 16 - 2; j >= 1; --j) {
                const double dj = readlane_f64(y0 * pl0, j);
                y0 = fma(-L[j], dj, y0);
            }
            if (tid < n) d[tid] = y0 * pl0;
        }
    } else
    if (tid < 64) {
        constexpr int U = 8;
        constexpr bool HI = NBLK > 4;   // unknowns 64.. exist
        const int base = n * (n + 1) / 2;
        double y0 = Lp[(tid < n) ? base + tid : zero];
        double y1 = HI ? Lp[(tid + 64 < n) ? base + tid + 64 : zero] : 0.0;
        const double pl0 = pinv[min(tid, max(n - 1, 0))];
        const double pl1 = HI ? pinv[min(tid + 64, max(n - 1, 0))] : 0.0;
        // The wavefront issues one instruction at a time, so a step costs its instruction count: the row fetch is kept to a scalar
        // subtract (the row offset j (j + 1) / 2 steps down by j), the address, the select of the zero word and the read -- round 3's
        // form recomputed the offset with a multiply and branched around a per-row read of 1 / d_j: 24 instructions a step, of which
        // the readlane -> multiply -> fma chain was six.  1 / d_j now comes from the lane that owns unknown j (every lane forms
        // y pinv, lane j's product is the one broadcast: the same product as before).
        double L0[U], L1[U];
        int jr = n - 1, offr = base - n;   // the next row to fetch and its offset jr (jr + 1) / 2   (scalars)
        auto fetch = [&](double& a0, double& a1) {   // (rows <= 0 have no entries: zeros)
            a0 = Lp[(tid < jr) ? offr + tid : zero];
            a1 = HI ? Lp[(tid + 64 < jr) ? offr + tid + 64 : zero] : 0.0;
            offr -= jr; --jr;
        };
#pragma unroll
        for (int u = 0; u < U; ++u) fetch(L0[u], L1[u]);
        for (int jb = n - 1; jb >= 1; jb -= U) {
            double N0[U], N1[U];
#pragma unroll
            for (int u = 0; u < U; ++u) fetch(N0[u], N1[u]);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int j = max(jb - u, 0);
                const double v0 = y0 * pl0;
                double dj = readlane_f64(v0, j & 63);
                if constexpr (HI) { const double v1 = y1 * pl1; const double dh = readlane_f64(v1, j & 63); dj = (j >= 64) ? dh : dj; }   // (a scalar select)
                y0 = fma(-L0[u], dj, y0);        // L0 / L1 are zero on and beyond the diagonal
                if (HI) y1 = fma(-L1[u], dj, y1);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) { L0[u] = N0[u]; L1[u] = N1[u]; }
        }
        if (tid < n) d[tid] = y0 * pl0;
        if (HI && tid + 64 < n) d[tid + 64] = y1 * pl1;
    }
    PROF_ACC(35, _tb0);
    __syncthreads();
    PROF_LAP(10);
}

template <int NBLK>
__device__ __forceinline__ bool ldl_solve(const AReg<NBLK>& A, int o_Lp, int o_g, int o_d, int o_pinv, int n) {
    if (!ldl_factor<NBLK>(APass<NBLK>::pack(A), o_Lp, o_g, o_pinv, n)) return false;
    ldl_backsub<NBLK>(o_Lp, o_d, o_pinv, n);
    return true;
}

// n > 127 unknowns (extended variant, NBLK > 8).  Neither a packed factor in LDS (151 KB for n = 194) nor a second
// register copy of the matrix (91 entries per thread) is affordable, so the elimination is LEFT-looking by 16-column
// block: only the current block column is live (<= NBLK entries per thread, taken from A, which stays untouched).  The packed
// factor -- entries c_ij = l_ij d_j, row n = the right-hand side -- is kept in this chain's GLOBAL scratch (L2-resident).
// Per block column b (round 4; round 3 eliminated two columns per barrier with everybody, updated with scalar fmas from L2 and
// streamed the back-substitution through one LDS panel: 130 + 110 + 61 us per solve at n = 194):
//   E  wavefront 0 eliminates the 16 columns of panel b inside the panel, lane = row (up to four rows per lane), pivots and
//      multipliers by v_readlane, exactly like ldl_panel_eliminate; the panel goes back to LDS (Wp) and out to the packed factor;
//   U1 MEANWHILE wavefronts 1..3 form, for block column b + 1, the products with the panels 0 .. b - 1 that are already final:
//      U[bi] = sum_c C[16 bi + r][c] C[16 (b + 1) + s][c] / d_c on the f64 matrix pipe (v_mfma_f64_16x16x4: 16 columns = 4 MFMAs
//      per tile, operands straight from L2, next group's loads in flight), and pass the tiles through LDS (Ux);
//   U2 everybody: W = A - U - (panel b's 16 columns, read from Wp in LDS), the block column b + 1 goes to Wp.
// Three barriers per 16 columns.  Back-substitution: wavefronts 1..3 stage 16 rows of the factor at a time in LDS (two buffers),
// wavefront 0 runs the steps (readlane -> fma, the rows fetched two steps ahead).
//   Lp (global): packed factor, entry (i, j) at i (i + 1) / 2 + j, rows 0..n; [trash .. trash + 265] spare words, [zero .. zero + 7] zeros.
//   Sl (LDS): [0 .. 63] per-lane trash, [64] zero, [65] "a pivot was not positive", [66 ..] Wp [CVR][17], Ux [NBLK][256]
//             (the back-substitution's two [16][CVR] buffers lie over Wp / Ux).
// E: wavefront 0 eliminates the 16 columns of panel P inside the panel (lane r = rows 16 P + r, + 64, ...; pivots and multipliers by
// v_readlane as in ldl_panel_eliminate), writes the panel back to Wp and out to the packed factor.
// (Tried: wavefront 0 on the diagonal block alone, multipliers through LDS, the rows below one per thread by everybody -- the same
//  arithmetic on four times the lanes.  Slower, 4.0 against 3.84 ms per cold frame: the phase is as long as the MFMA products the
//  other three wavefronts form meanwhile (U1), and the split adds a barrier and a second pass over Wp to every block column.)
template <int NBLK, int P, bool FULL>
__device__ __noinline__ bool big_panel_eliminate(double* Lp, double* Sl, double* pinv, int n, int lane) {
    constexpr int CVR = NBLK * 16, WS = 17, c0 = 16 * P;
    constexpr int NS = (CVR - c0 + 63) / 64;      // rows per lane
    double* Wp = Sl + 66;
    const int trashg = (n + 1) * (n + 2) / 2;
    double a[NS][16];
#pragma unroll
    for (int s_ = 0; s_ < NS; ++s_) {
        const int row = c0 + 64 * s_ + lane;
        const double* rp = (row < CVR) ? Wp + row * WS : Sl + 64;   // (rows beyond the matrix: zeros)
        const int st = (row < CVR) ? 1 : 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) a[s_][k] = rp[k * st];
    }
    bool bad = false;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        if (FULL || c0 + j < n) {   // (uniform: the border row's own "column" is not eliminated)
            const double pj = readlane_f64(a[0][j], j);
            bad = bad || !(pj > 0.0);
            double pin = __builtin_amdgcn_rcp(pj);
            pin = fma(fma(-pj, pin, 1.0), pin, pin);
            pin = fma(fma(-pj, pin, 1.0), pin, pin);
            *((lane == 0) ? pinv + c0 + j : Sl + lane) = pin;
            const double lj = a[0][j] * pin;
            double lkv[16];
#pragma unroll
            for (int k = j + 1; k < 16; ++k) lkv[k] = readlane_f64(lj, k);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = j + 1; k < 16; ++k)
#pragma unroll
                for (int s_ = 0; s_ < NS; ++s_) a[s_][k] = fma(-a[s_][j], lkv[k], a[s_][k]);
        }
    }
#pragma unroll
    for (int s_ = 0; s_ < NS; ++s_) {
        const int row = c0 + 64 * s_ + lane;
        // back to LDS for the next block column's update (rows below the diagonal block are what it reads) ...
        double* wp = (row < CVR) ? Wp + row * WS : Sl + lane;
        const int st = (row < CVR) ? 1 : 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) wp[k * st] = a[s_][k];
        // ... and out to the packed factor: entries on and left of the diagonal, rows up to the border row
        const int off = row * (row + 1) / 2 + c0;
#pragma unroll
        for (int k = 0; k < 16; ++k) Lp[(row <= n && c0 + k <= row) ? off + k : trashg + 10 + lane] = a[s_][k];
    }
    return bad;
}
// U1 (see above): one wavefront's tiles of block column bn against the final panels 0 .. bn - 2.  w3: 0..2 (wavefronts 1..3).
template <int NBLK>
__device__ __noinline__ void big_update_mfma(const double* Lp, double* Ux, const double* pinv, int n, int bn, int NB, int w3, int lane) {
    constexpr int TW = (NBLK + 2) / 3;   // tiles per wavefront at most
    const int zero = (n + 1) * (n + 2) / 2 + 1;
    const int m = lane & 15, kq = lane >> 4;
    const int rB = 16 * bn + m;
    const double* pB = Lp + ((rB <= n) ? rB * (rB + 1) / 2 : zero) + ((rB <= n) ? 4 * kq : 0);
    const int sB = (rB <= n) ? 1 : 0;
    const double* pA[TW];
    int sA[TW];
#pragma unroll
    for (int t = 0; t < TW; ++t) {
        const int rA = 16 * (bn + w3 + 3 * t) + m;
        pA[t] = Lp + ((rA <= n) ? rA * (rA + 1) / 2 : zero) + ((rA <= n) ? 4 * kq : 0);
        sA[t] = (rA <= n) ? 1 : 0;
    }
    v4d acc[TW];
#pragma unroll
    for (int t = 0; t < TW; ++t) acc[t] = v4d{0.0, 0.0, 0.0, 0.0};
    const int ng = bn - 1;   // groups of 16 final columns
    // The products are a few MFMAs; what costs is the round trip to the L2 for their operands -- so the groups go in chunks whose loads
    // are ALL in flight before the first is used (<= ~56 per lane: the vector-memory counter tells 63 apart): four groups at a time for a
    // wavefront with one or two tiles (late block columns: many groups), down to two with five tiles (early ones: few groups).
    int nt = 0;
#pragma unroll
    for (int t = 0; t < TW; ++t) nt += (bn + w3 + 3 * t < NB) ? 1 : 0;
    auto chunk = [&](auto gu_, int g0) {
        constexpr int GU = decltype(gu_)::value;
        double B_[GU][4], A_[GU][TW][4], P_[GU][4];
#pragma unroll
        for (int u = 0; u < GU; ++u) {
            const int g = min(g0 + u, ng - 1);   // (a chunk's surplus groups re-read the last one: multiplied by 0 below)
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_) { B_[u][s_] = pB[(16 * g + s_) * sB]; P_[u][s_] = pinv[16 * g + 4 * kq + s_]; }
#pragma unroll
            for (int t = 0; t < TW; ++t)
                if (bn + w3 + 3 * t < NB) {   // (uniform)
#pragma unroll
                    for (int s_ = 0; s_ < 4; ++s_) A_[u][t][s_] = pA[t][(16 * g + s_) * sA[t]];
                }
        }
#pragma unroll
        for (int u = 0; u < GU; ++u) {
            if (g0 + u < ng) {   // (uniform)
#pragma unroll
                for (int s_ = 0; s_ < 4; ++s_) {
                    const double bs = B_[u][s_] * P_[u][s_];
#pragma unroll
                    for (int t = 0; t < TW; ++t)
                        if (bn + w3 + 3 * t < NB) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(A_[u][t][s_], bs, acc[t], 0, 0, 0);
                }
            }
        }
    };
    if (nt <= 2) { for (int g0 = 0; g0 < ng; g0 += 4) chunk(std::integral_constant<int, 4>(), g0); }
    else if (nt == 3) { for (int g0 = 0; g0 < ng; g0 += 3) chunk(std::integral_constant<int, 3>(), g0); }
    else { for (int g0 = 0; g0 < ng; g0 += 2) chunk(std::integral_constant<int, 2>(), g0); }
    // register i of lane l holds entry ((l >> 4) + 4 i, l & 15) of the tile = thread (wavefront i, lane l)'s entry (AReg)
#pragma unroll
    for (int t = 0; t < TW; ++t)
        if (bn + w3 + 3 * t < NB) {
#pragma unroll
            for (int i = 0; i < 4; ++i) Ux[(bn + w3 + 3 * t) * 256 + i * 64 + lane] = acc[t][i];
        }
}
// back-substitution steps of one staged 16-row panel whose rows lie in lane group KK (unknowns 64 KK .. 64 KK + 63)
template <int NY, int KK>
__device__ __forceinline__ void big_backsub_panel(const double* sb, int CVR, int jlo, int jhi, int lane, double (&y)[NY], const double (&pl)[NY]) {
    double l0[NY], l1[NY];
    auto fetch = [&](int j, double (&l)[NY]) {
        const int r = max(j - jlo, 0);
#pragma unroll
        for (int k = 0; k <= KK; ++k) l[k] = sb[r * CVR + lane + 64 * k];
    };
    fetch(jhi, l0);
    for (int j = jhi; j >= jlo; j -= 2) {
        fetch(j - 1, l1);
        {
            const double dj = readlane_f64(y[KK] * pl[KK], j & 63);
#pragma unroll
            for (int k = 0; k <= KK; ++k) y[k] = fma(-l0[k], dj, y[k]);
        }
        fetch(j - 2, l0);
        if (j - 1 >= jlo) {
            const double dj = readlane_f64(y[KK] * pl[KK], (j - 1) & 63);
#pragma unroll
            for (int k = 0; k <= KK; ++k) y[k] = fma(-l1[k], dj, y[k]);
        }
    }
}
// block column B (elimination, update of B + 1) and on to B + 1; false: a pivot was not positive
template <int NBLK, int B>
__device__ __forceinline__ bool big_blocks(const AReg<NBLK>& A, double* Lp, double* Sl, const double* g, double* pinv, int n, int NB) {
    int tid = threadIdx.x;
    MOSHII_OPAQUE(tid);   // (inlined into the dogleg loop: what follows from the thread index is recomputed here, not hoisted out of the loop and spilled)
    const int ty = tid >> 4, tx = tid & 15, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int CVR = NBLK * 16, WS = 17;
    double* Wp = Sl + 66;
    double* Ux = Wp + CVR * WS;
    if (B >= NB) return true;   // (uniform)
    if (wv == 0) {
        const bool bad = (16 * B + 16 <= n) ? big_panel_eliminate<NBLK, B, true>(Lp, Sl, pinv, n, lane)
                                             : big_panel_eliminate<NBLK, B, false>(Lp, Sl, pinv, n, lane);
        if (bad && lane == 0) Sl[65] = 1.0;
    } else if (B + 1 < NB && B >= 1) {
        big_update_mfma<NBLK>(Lp, Ux, pinv, n, B + 1, NB, wv - 1, lane);
    }
    __syncthreads();
    PROF_LAP_EXT(14);
    if (Sl[65] != 0.0) return false;   // (every thread sees it)
    if constexpr (B + 1 < NBLK) {
        if (B + 1 < NB) {
            // W = A - U1 - (panel B's columns)
            // (tried: panel B's columns on the matrix pipe too, operands from Wp, added into the exchange tiles by their owners -- the same
            //  33 us per solve: this phase is its three barriers and the two passes over Ux / Wp, not the 16 x NBLK fmas)
            double W[NBLK];
            const int q2 = (B + 1) * 16 + tx;
            const double gq = g[min(q2, n - 1)] * ((q2 < n) ? 1.0 : 0.0);
#pragma unroll
            for (int bi = B + 1; bi < NBLK; ++bi) {
                const double v = (bi * 16 + ty == n) ? gq : A.a[bi * (bi + 1) / 2 + B + 1];
                W[bi] = (B >= 1 && bi < NB) ? v - Ux[bi * 256 + tid] : v;
            }
            const double* wk = Wp + q2 * WS;
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) {
                const double ck = wk[jj] * pinv[16 * B + jj];
#pragma unroll
                for (int bi = B + 1; bi < NBLK; ++bi) W[bi] = fma(-Wp[(bi * 16 + ty) * WS + jj], ck, W[bi]);
            }
            __syncthreads();   // every read of panel B in Wp is done
#pragma unroll
            for (int bi = B + 1; bi < NBLK; ++bi) {
                const int q1 = bi * 16 + ty;
                Wp[q1 * WS + tx] = (q1 > n || q2 > n) ? 0.0 : W[bi];
            }
            __syncthreads();
            PROF_LAP_EXT(13);
        }
        return big_blocks<NBLK, B + 1>(A, Lp, Sl, g, pinv, n, NB);
    }
    return true;
}
template <int NBLK>
__device__ bool ldl_big(const AReg<NBLK>& A, double* Lp, double* Sl, const double* g, double* d, double* pinv, int n) {
    int tid = threadIdx.x;
    MOSHII_OPAQUE(tid);
    const int ty = tid >> 4, tx = tid & 15, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    PROF_BEGIN(); PROF_COUNT(22);
    const int trash = (n + 1) * (n + 2) / 2, zero = trash + 1;
    constexpr int CVR = NBLK * 16, WS = 17;
    double* Wp = Sl + 66;
    const int NB = n / 16 + 1;   // block rows / columns, the border row included
    if (tid == 0) { Sl[64] = 0.0; Sl[65] = 0.0; }
    if (tid < 8) Lp[zero + tid] = 0.0;
    // block column 0 of [A; g^T] as it stands
    {
        const double gq = g[min(tx, n - 1)] * ((tx < n) ? 1.0 : 0.0);
#pragma unroll
        for (int bi = 0; bi < NBLK; ++bi) {
            const int q1 = bi * 16 + ty;
            const double v = (q1 == n) ? gq : A.a[bi * (bi + 1) / 2];
            Wp[q1 * WS + tx] = (q1 > n || tx > n) ? 0.0 : v;
        }
    }
    __syncthreads();
    PROF_MARK();
    const bool ok = big_blocks<NBLK, 0>(A, Lp, Sl, g, pinv, n, NB);
    __syncthreads();
    PROF_LAP(9);
    if (!ok) return false;
    // ---- back-substitution: x_j = (y_j - sum_{i > j} c_ij x_i) / d_j, rows n-1 .. 0, 16 at a time through LDS
    constexpr int NY = (CVR + 63) / 64;   // unknowns per lane of wavefront 0
    double* sb0 = Sl + 66;                // two [16][CVR] buffers
    const int base = n * (n + 1) / 2;
    double y[NY], pl[NY];
#pragma unroll
    for (int k = 0; k < NY; ++k) {
        y[k] = (wv == 0 && lane + 64 * k < n) ? Lp[base + lane + 64 * k] : 0.0;
        pl[k] = pinv[min(lane + 64 * k, max(n - 1, 0))];
    }
    auto stage = [&](int pnl) {   // rows 16 pnl .. 16 pnl + 15 (those below n), entries (j, 0 .. j-1), zero beyond: wavefronts 1..3
        double* sb = sb0 + (pnl & 1) * 16 * CVR;
        const int jlo = 16 * pnl, t3 = tid - 64;   // 192 threads, thread = column (a second round for columns 192.. where rows reach them)
        for (int i0 = 0; i0 < jlo + 15; i0 += MOSHII_TPB - 64) {
            const int i = i0 + t3;
            double v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {   // sixteen independent loads in flight
                const int j = jlo + r;
                v[r] = Lp[(i < j && j < n) ? j * (j + 1) / 2 + i : zero];
            }
            if (i < CVR) {
#pragma unroll
                for (int r = 0; r < 16; ++r) sb[r * CVR + i] = v[r];
            }
        }
        // (every column of the lane groups wavefront 0 reads for this panel -- 64 (pnl / 4 + 1) <= the rounds' 192 / 384 -- is written: zeros
        //  on and right of the diagonal)
    };
    const int PN = (n + 15) / 16;         // row panels 0 .. PN-1 hold rows 0 .. n-1
    if (wv != 0) stage(PN - 1);
    __syncthreads();
    for (int pnl = PN - 1; pnl >= 0; --pnl) {
        if (wv != 0) { if (pnl > 0) stage(pnl - 1); }
        else {
            const double* sb = sb0 + (pnl & 1) * 16 * CVR;
            const int jlo = 16 * pnl, jhi = min(jlo + 15, n - 1);
            switch (pnl >> 2) {   // (uniform) the lane group the panel's unknowns live in
                case 0: big_backsub_panel<NY, 0>(sb, CVR, jlo, jhi, lane, y, pl); break;
                case 1: if constexpr (NY > 1) big_backsub_panel<NY, 1>(sb, CVR, jlo, jhi, lane, y, pl); break;
                case 2: if constexpr (NY > 2) big_backsub_panel<NY, 2>(sb, CVR, jlo, jhi, lane, y, pl); break;
                default: if constexpr (NY > 3) big_backsub_panel<NY, 3>(sb, CVR, jlo, jhi, lane, y, pl); break;
            }
        }
        __syncthreads();
    }
    if (wv == 0) {
#pragma unroll
        for (int k = 0; k < NY; ++k) if (lane + 64 * k < n) d[lane + 64 * k] = y[k] * pl[k];
    }
    __syncthreads();
    PROF_LAP(10);
    return true;
}

template <int NBLK> __device__ __noinline__ bool coop_rs_reduce(unsigned seq_);   // (defined behind the context helpers below)

// ------------------------------------------------------------------------------------------------
// Normal equations at the point whose forward state is in LDS:  A = J^T J (registers), g = -J^T r (LDS).
// ------------------------------------------------------------------------------------------------
// Columns: [trans 3][free pose variables][free shape coefficients]; ncp = 3 + #pose columns, n - ncp = #shape columns (XT only).
// COOP: the rows of this rank's visible markers only (entries [fp.v0, fp.v1) of the list); the ranks' partial products -- and the prior
// rank's structured terms -- meet in an exchange before the tiles are re-dealt, after which every rank holds the same A and g.
template <int NBLK, bool XT, bool COOP = false>
__device__ void assemble(const Ctx& cx, const ChainLayout& ly, const ModelDev& md, const AttachDev& at,
                         const PriorDev& pr, const OptsDev& op, const double* pose, const FrameParams& fp,
                         int n, int ncp, int nkf, int nfree_hand, double* qs, AReg<NBLK>& A, const CoopCtx& co = CoopCtx()) {
    const int tid = threadIdx.x;
    const int LDJ = ly.LDJ, Tm = ly.Tm, NW = at.NW, Nvp = at.Nvp, bd = md.body_dof, nhf = md.nhand_full;
    const int nshp = XT ? n - ncp : 0;
    // per-vertex strides of the tile tables xjs / tjs: 4 NW + 2 doubles and NW + 1 ints.  With the dense strides (16 NW bytes) the
    // T1 items of a wavefront -- consecutive tile markers, 3 vertices apart -- fell onto two bank groups (PMC: 30 % of the kernel's
    // LDS-active cycles were bank conflicts); an odd number of 16-byte units per marker spreads them over all banks.
    const int XS = NW * 4 + 2, TS = NW + 1;
    PROF_BEGIN(); PROF_COUNT(21);
    JtJAcc<NBLK> acc;
    acc.zero();
    const bool rows_here = !COOP || co.mhi > co.mlo;   // (a cooperative rank without markers builds no Jacobian rows)
    if (rows_here) {
    for (int e = tid; e < 3 * Tm * LDJ; e += MOSHII_TPB) cx.Jrow[e] = 0.0;
    // Jacobian-only joint quantities at the current point (the forward state of the last evaluation is in LDS):
    // left-Jacobian columns a_c of each joint rotation and dR/dtheta_c = [a_c]x R
    if (tid < md.K) {
        double R[9], Jl[9];
        rodrigues_dev(&cx.fullpose[3 * tid], R, Jl);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const double ax = Jl[0 * 3 + c], ay = Jl[1 * 3 + c], az = Jl[2 * 3 + c];
            cx.acol[tid * 9 + c * 3 + 0] = ax;
            cx.acol[tid * 9 + c * 3 + 1] = ay;
            cx.acol[tid * 9 + c * 3 + 2] = az;
#pragma unroll
            for (int d = 0; d < 3; ++d) {   // column d of R
                const double rx = R[0 * 3 + d], ry = R[1 * 3 + d], rz = R[2 * 3 + d];
                cx.B[tid * 28 + c * 9 + 0 * 3 + d] = ay * rz - az * ry;
                cx.B[tid * 28 + c * 9 + 1 * 3 + d] = az * rx - ax * rz;
                cx.B[tid * 28 + c * 9 + 2 * 3 + d] = ax * ry - ay * rx;
            }
        }
    }
    __syncthreads();
    // world rotation axes omega_{k,c} = Rw_par(k) . a_{k,c}   (published by the barrier that ends T0)
    if (tid < 3 * md.K) {
        const int k = tid / 3, c = tid % 3;
        const double ax = cx.acol[k * 9 + c * 3 + 0], ay = cx.acol[k * 9 + c * 3 + 1], az = cx.acol[k * 9 + c * 3 + 2];
        double ox = ax, oy = ay, oz = az;
        if (k > 0) mat3_vec(&cx.Rw[md.parents[k] * 9], ax, ay, az, ox, oy, oz);
        cx.omega[k * 10 + c * 3 + 0] = ox; cx.omega[k * 10 + c * 3 + 1] = oy; cx.omega[k * 10 + c * 3 + 2] = oz;
    }
    }   // (rows_here)
    if constexpr (XT) {
        if (nshp > 0 && rows_here) {
            // shape derivative of the joint transforms at the current point (the restated lbs_derivatives_wrt_shape):
            //   dt_0 = JS_0, dt_j = dt_par + Rw_par (JS_j - JS_par)   (joint world positions, one tree level per step)
            //   q_j  = dt_j - Rw_j JS_j                               (so that dv/ds = Trot . S(v) + sum_j w_j q_j)
            // K x nshape x 3 doubles each: too large for LDS next to the Jacobian tiles, so they live in this chain's
            // global scratch (L2-resident; written here, read by the T1s items below).
            const int E = nshp, KE = md.K * E;
            double* dtv = qs;
            double* qv = qs + (size_t)KE * 3;
            for (int lvl = 0; lvl <= md.maxdepth; ++lvl) {
                // (the level's joints from a list built once per chain: scanning all K E items for those of this depth was 17 rounds of a
                //  division, a depth load and a branch per thread and level -- a third of this phase's 45 us at 55 joints x 80 coefficients)
                const auto* const bl = gptr(md.depth) + md.K;   // (ModelDev::depth: the sorted list and the level starts ride behind the depths)
                const int l0 = bl[md.K + lvl], nl = bl[md.K + lvl + 1] - l0;
                for (int ix = tid; ix < nl * E; ix += MOSHII_TPB) {
                    const int jl = ix / E, e = ix - jl * E;
                    const int j = bl[l0 + jl], it = j * E + e;
                    const auto* js = md.JS + (size_t)it * 3;
                    const double jx = js[0], jy = js[1], jz = js[2];
                    double dx = jx, dy = jy, dz = jz;
                    if (j > 0) {
                        const int p = md.parents[j];
                        const auto* jp = md.JS + ((size_t)p * E + e) * 3;
                        const double* dp = dtv + ((size_t)p * E + e) * 3;
                        double ox, oy, oz;
                        mat3_vec(&cx.Rw[p * 9], jx - jp[0], jy - jp[1], jz - jp[2], ox, oy, oz);
                        dx = dp[0] + ox; dy = dp[1] + oy; dz = dp[2] + oz;
                    }
                    dtv[(size_t)it * 3 + 0] = dx; dtv[(size_t)it * 3 + 1] = dy; dtv[(size_t)it * 3 + 2] = dz;
                    double rx, ry, rz;
                    mat3_vec(&cx.Rw[j * 9], jx, jy, jz, rx, ry, rz);
                    qv[(size_t)it * 3 + 0] = dx - rx; qv[(size_t)it * 3 + 1] = dy - ry; qv[(size_t)it * 3 + 2] = dz - rz;
                }
                __syncthreads();
            }
        }
    }
    const int v_lo = COOP ? fp.v0 : 0, v_hi = COOP ? fp.v1 : fp.nobs;
    for (int tile0 = v_lo; tile0 < v_hi; tile0 += Tm) {
        const int cnt = min(Tm, v_hi - tile0);
        const int ntv = 3 * cnt;
        // T0: per tile vertex blended rotation + rigidly-attached positions; per tile marker local Jacobian
        if (tid < ntv) {
            const int m = cx.visidx[tile0 + tid / 3];
            const int av = 3 * m + tid % 3;
            const double px = cx.vposed[av * 3 + 0], py = cx.vposed[av * 3 + 1], pz = cx.vposed[av * 3 + 2];
            double Tr[9];
#pragma unroll
            for (int e = 0; e < 9; ++e) Tr[e] = 0.0;
            auto influence = [&](int s, int j, double w) {
                double ox, oy, oz;
                mat3_vec(&cx.Rw[j * 9], px - cx.Jl[j * 3 + 0], py - cx.Jl[j * 3 + 1], pz - cx.Jl[j * 3 + 2], ox, oy, oz);
                cx.xjs[tid * XS + s * 4 + 0] = ox + cx.tw[j * 3 + 0];
                cx.xjs[tid * XS + s * 4 + 1] = oy + cx.tw[j * 3 + 1];
                cx.xjs[tid * XS + s * 4 + 2] = oz + cx.tw[j * 3 + 2];
                cx.xjs[tid * XS + s * 4 + 3] = w;
                cx.tjs[tid * TS + s] = j;
#pragma unroll
                for (int e = 0; e < 9; ++e) Tr[e] += w * cx.Rw[j * 9 + e];
            };
            if (NW <= 4) {   // (uniform; see eval_forward F5: the influences in one round of loads)
                int jj[4];
                double wv[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) { jj[s] = gptr(at.wj)[av * NW + min(s, NW - 1)]; wv[s] = gptr(at.ww)[av * NW + min(s, NW - 1)]; }
#pragma unroll
                for (int s = 0; s < 4; ++s) if (s < NW) influence(s, jj[s], wv[s]);
            } else {
                for (int s = 0; s < NW; ++s) influence(s, gptr(at.wj)[av * NW + s], gptr(at.ww)[av * NW + s]);
            }
#pragma unroll
            for (int e = 0; e < 9; ++e) cx.Trot[tid * 10 + e] = Tr[e];
        } else if (tid >= 128 && tid < 128 + cnt) {
            const int ml = tid - 128;
            const int m = cx.visidx[tile0 + ml];
            const double c[3] = {gptr(at.coef)[m * 3 + 0], gptr(at.coef)[m * 3 + 1], gptr(at.coef)[m * 3 + 2]};
            double mk[3], L[27];
            marker_eval(c, &cx.vpos[(3 * m + 0) * 3], &cx.vpos[(3 * m + 1) * 3], &cx.vpos[(3 * m + 2) * 3], mk, L);
#pragma unroll
            for (int sv = 0; sv < 3; ++sv)   // stored per vertex: [ml][sv][row*3 + x] in rows of 10 doubles (16-byte reads in T1)
#pragma unroll
                for (int row = 0; row < 3; ++row)
#pragma unroll
                    for (int x = 0; x < 3; ++x) cx.Lm[(ml * 3 + sv) * 10 + row * 3 + x] = L[row * 9 + sv * 3 + x];
#pragma unroll
            for (int i = 0; i < 3; ++i) cx.Jrow[(3 * ml + i) * LDJ + n] = cx.res[m * 3 + i];   // residual = column n of the tile (T3)
        }
        __syncthreads();
        PROF_LAP(4);
        // T1: marker rows, fused over the marker's three vertices: item = (tile marker, needed joint), marker fastest so
        // that the posedirs gathers of a wavefront fall into a few contiguous runs.  For each vertex
        //   dv/dtheta_{k,c} = omega_{k,c} x sum_{j in subtree(k)} w_j (x_j - t_k)  +  Trot . (P_k . vec([a_c]x R_k))
        // and the three columns are contracted at once with the marker's local 3x9 Jacobian L.
        const int bj0 = bd / 3;   // first hand joint (its dofs are hand-PCA coefficients, not pose variables)
        for (int it = tid; it < cnt * nkf; it += MOSHII_TPB) {
            const int kfi = it / cnt, ml = it - kfi * cnt;
            const int k = cx.kfree[kfi];
            const int m = cx.visidx[tile0 + ml];
            const unsigned long long mask = cx.anc[k];
            const double tkx = cx.tw[k * 3 + 0], tky = cx.tw[k * 3 + 1], tkz = cx.tw[k * 3 + 2];
            double om[10];   // (rows of 10 doubles: five 16-byte LDS reads)
            {
                const double2* o2 = reinterpret_cast<const double2*>(&cx.omega[k * 10]);
#pragma unroll
                for (int q = 0; q < 5; ++q) { const double2 t2 = o2[q]; om[2 * q] = t2.x; om[2 * q + 1] = t2.y; }
            }
            double r[9];
#pragma unroll
            for (int e = 0; e < 9; ++e) r[e] = 0.0;
            double Bk[28];   // dR_k/dtheta_c, c = 0..2: the same for the marker's three vertices (14 16-byte LDS reads, once per item)
            {
                const double2* bp2 = reinterpret_cast<const double2*>(&cx.B[k * 28]);
#pragma unroll
                for (int q = 0; q < 14; ++q) { const double2 t2 = bp2[q]; Bk[2 * q] = t2.x; Bk[2 * q + 1] = t2.y; }
            }
#pragma unroll
            for (int sv = 0; sv < 3; ++sv) {
                const int al = 3 * ml + sv;
                double ax = 0.0, ay = 0.0, az = 0.0;
                for (int s2 = 0; s2 < NW; ++s2) {   // joints of this vertex inside the subtree of k (branch-free: weight 0 otherwise)
                    const int j = cx.tjs[al * TS + s2];
                    const double2* xw = reinterpret_cast<const double2*>(&cx.xjs[al * XS + s2 * 4]);
                    const double2 xy = xw[0], zw = xw[1];   // two 16-byte reads; the weight rides with z, so it cannot be sunk into a branch
                    const double w = ((mask >> j) & 1ull) ? zw.y : 0.0;
                    ax += w * (xy.x - tkx);
                    ay += w * (xy.y - tky);
                    az += w * (zw.x - tkz);
                }
                double col[9];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    col[c * 3 + 0] = om[c * 3 + 1] * az - om[c * 3 + 2] * ay;
                    col[c * 3 + 1] = om[c * 3 + 2] * ax - om[c * 3 + 0] * az;
                    col[c * 3 + 2] = om[c * 3 + 0] * ay - om[c * 3 + 1] * ax;
                }
                {   // pose-corrective part; the root joint (k = 0) has none: it reads joint 1's record and scales by 0
                    const double cs = (k >= 1) ? 1.0 : 0.0;
                    const int kk = max(k, 1);
                    double pv[28];   // 14 x 16-byte loads per lane, contiguous across the markers of a wavefront (AttachDev::Pj)
                    const auto* pp = gptr(reinterpret_cast<const v2d*>(at.Pj)) + (size_t)((kk - 1) * 3 + sv) * 14 * at.M + m;
#pragma unroll
                    for (int q = 0; q < 14; ++q) { const v2d t2 = pp[(size_t)q * at.M]; pv[2 * q] = t2[0]; pv[2 * q + 1] = t2[1]; }
                    double Tr[10];
                    {
                        const double2* t2p = reinterpret_cast<const double2*>(&cx.Trot[al * 10]);
#pragma unroll
                        for (int q = 0; q < 5; ++q) { const double2 t2 = t2p[q]; Tr[2 * q] = t2.x; Tr[2 * q + 1] = t2.y; }
                    }
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const double* Bc = &Bk[c * 9];
                        double pd[3];
#pragma unroll
                        for (int i = 0; i < 3; ++i) {
                            double sacc = 0.0;
#pragma unroll
                            for (int e = 0; e < 9; ++e) sacc += pv[i * 9 + e] * Bc[e];
                            pd[i] = cs * sacc;
                        }
#pragma unroll
                        for (int i = 0; i < 3; ++i) col[c * 3 + i] += Tr[i * 3 + 0] * pd[0] + Tr[i * 3 + 1] * pd[1] + Tr[i * 3 + 2] * pd[2];
                    }
                }
                double Ls[10];
                {
                    const double2* l2 = reinterpret_cast<const double2*>(&cx.Lm[(ml * 3 + sv) * 10]);
#pragma unroll
                    for (int q = 0; q < 5; ++q) { const double2 t2 = l2[q]; Ls[2 * q] = t2.x; Ls[2 * q + 1] = t2.y; }
                }
#pragma unroll
                for (int row = 0; row < 3; ++row)
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        r[row * 3 + c] += Ls[row * 3 + 0] * col[c * 3 + 0] + Ls[row * 3 + 1] * col[c * 3 + 1] + Ls[row * 3 + 2] * col[c * 3 + 2];
            }
            if (k < bj0) {   // pose variables of a body joint: write the free ones straight into the Jacobian tile
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int q = cx.colq[3 * k + c];
                    if (q >= 0) {
                        cx.Jrow[(3 * ml + 0) * LDJ + q] = fp.wt_data * r[0 * 3 + c];
                        cx.Jrow[(3 * ml + 1) * LDJ + q] = fp.wt_data * r[1 * 3 + c];
                        cx.Jrow[(3 * ml + 2) * LDJ + q] = fp.wt_data * r[2 * 3 + c];
                    }
                }
            } else {         // hand joint: park d marker / d fullpose for the PCA contraction below
                double* jh = &cx.Jh[((size_t)ml * ly.nhj + (k - bj0)) * 9];
#pragma unroll
                for (int e = 0; e < 9; ++e) jh[e] = r[e];
            }
        }
        if constexpr (XT) {
            // (tried: an item's 9 + 36 loads all in flight before the first use instead of influence by influence -- T1 + T1s 160 -> 284 us per cold
            //  config-3 frame: assemble_fn<13> has no registers left for 45 more doubles per lane and spills them)
            // T1s: shape columns.  item = (tile marker, coefficient): dv/ds_e = Trot . S_e(v) + sum_s w_s q_{j_s, e} for the
            // marker's three vertices, contracted with the marker's local 3x9 Jacobian.
            if (nshp > 0) {
                const double* qv = qs + (size_t)md.K * nshp * 3;
                for (int it = tid; it < cnt * nshp; it += MOSHII_TPB) {
                    const int e = it / cnt, ml = it - e * cnt;
                    const int m = cx.visidx[tile0 + ml];
                    double r0 = 0.0, r1 = 0.0, r2 = 0.0;
#pragma unroll
                    for (int sv = 0; sv < 3; ++sv) {
                        const int al = 3 * ml + sv, av = 3 * m + sv;
                        const auto* sp = gptr(at.Ssh) + (size_t)e * 3 * Nvp + av;
                        const double sx = sp[0], sy = sp[Nvp], sz = sp[2 * Nvp];
                        const double* Tr = &cx.Trot[al * 10];
                        double dx = Tr[0] * sx + Tr[1] * sy + Tr[2] * sz;
                        double dy = Tr[3] * sx + Tr[4] * sy + Tr[5] * sz;
                        double dz = Tr[6] * sx + Tr[7] * sy + Tr[8] * sz;
                        for (int s2 = 0; s2 < NW; ++s2) {
                            const int j = cx.tjs[al * TS + s2];
                            const double w = cx.xjs[al * XS + s2 * 4 + 3];
                            const double* qq = qv + ((size_t)j * nshp + e) * 3;
                            dx += w * qq[0]; dy += w * qq[1]; dz += w * qq[2];
                        }
                        const double* Ls = &cx.Lm[(ml * 3 + sv) * 10];
                        r0 += Ls[0] * dx + Ls[1] * dy + Ls[2] * dz;
                        r1 += Ls[3] * dx + Ls[4] * dy + Ls[5] * dz;
                        r2 += Ls[6] * dx + Ls[7] * dy + Ls[8] * dz;
                    }
                    cx.Jrow[(3 * ml + 0) * LDJ + ncp + e] = fp.wt_data * r0;
                    cx.Jrow[(3 * ml + 1) * LDJ + ncp + e] = fp.wt_data * r1;
                    cx.Jrow[(3 * ml + 2) * LDJ + ncp + e] = fp.wt_data * r2;
                }
            }
        }
        // translation columns: d marker / d trans = I
        for (int it = tid; it < cnt * 9; it += MOSHII_TPB) {
            const int ml = it / 9, row = (it % 9) / 3, q = it % 3;
            cx.Jrow[(3 * ml + row) * LDJ + q] = (row == q) ? fp.wt_data : 0.0;
        }
        __syncthreads();
        PROF_LAP(5);
        // T2: hand-PCA columns: d marker / d pose[bd + i] = sum_h comps[i][h] d marker / d fullpose[bd + h]
        if (nfree_hand > 0) {
            for (int it = tid; it < cnt * nfree_hand; it += MOSHII_TPB) {
                const int ml = it / nfree_hand, q = ncp - nfree_hand + (it - ml * nfree_hand);   // hand columns are the tail of the pose columns
                const int i = cx.colpid[q] - bd;
                double r0 = 0.0, r1 = 0.0, r2 = 0.0;
                const int hlo = md.comp_lo[i], hhi = md.comp_hi[i];
                for (int h0 = hlo; h0 < hhi; h0 += 15) {   // fifteen components' loads in flight at a time (one by one: 45 round trips to the L2 per item)
                    double ccv[15];
#pragma unroll
                    for (int u = 0; u < 15; ++u) ccv[u] = md.comps[i * nhf + min(h0 + u, hhi - 1)] * ((h0 + u < hhi) ? 1.0 : 0.0);
#pragma unroll
                    for (int u = 0; u < 15; ++u) {
                        const int h = min(h0 + u, hhi - 1), hj = h / 3, c = h - 3 * hj;
                        const double cc = ccv[u];
                        const double* jh = &cx.Jh[((size_t)ml * ly.nhj + hj) * 9];
                        r0 += cc * jh[0 * 3 + c]; r1 += cc * jh[1 * 3 + c]; r2 += cc * jh[2 * 3 + c];
                    }
                }
                cx.Jrow[(3 * ml + 0) * LDJ + q] = fp.wt_data * r0;
                cx.Jrow[(3 * ml + 1) * LDJ + q] = fp.wt_data * r1;
                cx.Jrow[(3 * ml + 2) * LDJ + q] = fp.wt_data * r2;
            }
            __syncthreads();
        }
        PROF_LAP(6);
        // T3: [A; -g^T] += [Jt r]^T [Jt r]: the weighted residual rides as column n of the tile (n < LDJ), so the same
        // matrix-pipe pass that accumulates J^T J leaves J^T r in row n of the product -- no separate gradient loop
        acc.accumulate(cx.Jrow, ntv, LDJ);
        __syncthreads();
        PROF_LAP(7);
    }
    const int np_ = op.nbody;
    double pblk[COOP ? AReg<NBLK>::NE : 1];   // cooperative variant: the prior's block of A in the owners' layout, as received from the prior rank
    if constexpr (COOP) {
        // Slot layout in 16-byte units, thread index fastest (coalesced 1 KiB rows): units [0, 2 NT) x 256 = this thread's accumulator
        // registers (t, 2 h), (t, 2 h + 1); then (prior rank only) NUP units = its NE prior-block entries and its gradient entry.
        constexpr int NT = JtJAcc<NBLK>::NT, NE = AReg<NBLK>::NE, NUA = 2 * NT, NUP = (NE + 2) / 2;
        const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
        unsigned seq = coop_begin(cx);   // (the reduce-scatter form below moves on to a second exchange)
        TRACE_STAMP(co, seq, 0);
        double gqp = 0.0;
        auto exchange = [&](auto l2) {   // (l2: the access scope, see coop_st_l2)
            constexpr bool L2 = decltype(l2)::value;
            const CoopSlot mine = coop_slot16(co, seq, co.rank);
#pragma unroll
            for (int t = 0; t < NT; ++t)
                if (4 * t + wv < NE) {   // (uniform: the tile slots beyond the last tile hold zeros nobody reads)
                    coop_st16<L2>(mine, (2 * t) * MOSHII_TPB + tid, acc.c[t][0], acc.c[t][1]);
                    coop_st16<L2>(mine, (2 * t + 1) * MOSHII_TPB + tid, acc.c[t][2], acc.c[t][3]);
                }
            if (np_ > 0 && co.rank == co.prior_rank) {
                // the prior's share of the normal equations at this point (its argmin component kb was found by this rank's evaluation):
                // block w^2 (1/2 L L^T)[colprior, colprior] in the owners' layout, gradient -w^2 (1/2 L L^T)(x - mu) per column
                const int kb = (int)cx.scal[S_KBEST];
                const double w2 = fp.wt_pose * fp.wt_pose;
                AReg<NBLK> P;
                P.zero();
                P.add_prior(w2, pr.halfprec + (size_t)kb * np_ * np_, np_, cx.colprior, n);
                double gq = 0.0;
                if (tid < n) {
                    const int pb = cx.colprior[tid];
                    if (pb >= 0) {
                        const auto* Hk = pr.halfprec + (size_t)kb * np_ * np_ + pb;
                        const auto* mu = pr.means + (size_t)kb * np_;
                        double s0 = 0.0, s1 = 0.0;
                        for (int b0 = 0; b0 < np_; b0 += 32) {
                            double h[32];
#pragma unroll
                            for (int k = 0; k < 32; ++k) h[k] = Hk[(size_t)min(b0 + k, np_ - 1) * np_];
#pragma unroll
                            for (int k = 0; k < 32; k += 2) {
                                const int b = b0 + k;
                                s0 = fma(h[k] * ((b < np_) ? 1.0 : 0.0), cx.xb[min(b, np_ - 1)] - mu[min(b, np_ - 1)], s0);
                                s1 = fma(h[k + 1] * ((b + 1 < np_) ? 1.0 : 0.0), cx.xb[min(b + 1, np_ - 1)] - mu[min(b + 1, np_ - 1)], s1);
                            }
                        }
                        gq = -(w2 * (s0 + s1));
                    }
                }
                double pe[2 * NUP];
#pragma unroll
                for (int e = 0; e < 2 * NUP; ++e) pe[e] = (e < NE) ? P.a[e] : ((e == NE) ? gq : 0.0);
#pragma unroll
                for (int u = 0; u < NUP; ++u) coop_st16<L2>(mine, (NUA + u) * MOSHII_TPB + tid, pe[2 * u], pe[2 * u + 1]);
            }
            constexpr bool RS = NT >= 9;   // reduce-scatter + all-gather instead of the all-gather of every rank's whole slot (below)
            double own[RS ? 1 : 4 * NT];
            if constexpr (!RS) {
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int i = 0; i < 4; ++i) own[4 * t + i] = acc.c[t][i];
                acc.zero();
            }
#pragma unroll
            for (int e = 0; e < NE; ++e) pblk[e] = 0.0;
            if (coop_publish_wait<L2>(co, cx, seq)) {
                // Ranks in batches of RB: every load of a batch is in flight before the first is added (a rank-by-rank loop pays the
                // round trip to the memory side -- write-through lines are not kept in the L2 -- once per rank).  Rank order in the sums:
                // the same bits on every rank; the own share comes from the registers (what the slot holds).
                // (at most ~48 loads in flight per thread: with more than the 63 the vector-memory counter can tell apart -- 8 blocks, four ranks per
                //  batch: 72 + 19 -- the device build returned wrong sums, in the emulation as on paper nothing is wrong with it)
                constexpr int RB = (48 / (2 * NT) < 1) ? 1 : ((48 / (2 * NT) > 6) ? 6 : 48 / (2 * NT));
                const bool has_prior = np_ > 0;
                const CoopSlot sp = coop_slot16(co, seq, co.prior_rank);
                if constexpr (RS) {
                    // Large matrices (8 register blocks and up: 72 KB .. 184 KB per rank) as a REDUCE-SCATTER + ALL-GATHER: rank k sums tiles
                    // [k TPR, (k + 1) TPR) of every rank -- in rank order, the own share from registers: the bits of the all-gather form --,
                    // posts the sums in a second exchange and collects the other ranks'.  A rank reads 2 (G - 1) / G of a slot instead
                    // of G - 1 slots (config 3, eight ranks: 0.33 MB instead of 1.3 MB, which took 62 us of an assembly's ~170), and the
                    // ranks no longer stream the same slot at the same moment.
                    if (has_prior) {
                        double pv[2 * NUP];
#pragma unroll
                        for (int u = 0; u < NUP; ++u) coop_ld16<L2>(sp, (NUA + u) * MOSHII_TPB + tid, pv[2 * u], pv[2 * u + 1]);
#pragma unroll
                        for (int e = 0; e < NE; ++e) pblk[e] = pv[e];
                        gqp = pv[NE];
                    }
                    // (the reduction in a function of its own that works slot to slot -- this rank's own share is read back from its
                    //  slot like everybody else's: inside assemble_fn hipcc 7.2's register allocator crashes on it, and handing it the
                    //  accumulators by value faulted on the device)
                    const bool ok2 = coop_rs_reduce<NBLK>(seq);
                    seq += 1;
                    if (ok2) {
                        const int TPR = (NT + co.G - 1) / co.G;
#pragma unroll
                        for (int t = 0; t < NT; ++t)
                            if (4 * t + wv < NE) {   // (every load in flight: 2 NT <= 46)
                                const CoopSlot sr = coop_slot16(co, seq, t / TPR);
                                double w0, w1, w2, w3;
                                coop_ld16<L2>(sr, (2 * t) * MOSHII_TPB + tid, w0, w1);
                                coop_ld16<L2>(sr, (2 * t + 1) * MOSHII_TPB + tid, w2, w3);
                                acc.c[t] = v4d{w0, w1, w2, w3};
                            }
                    }
                } else {
                double pv[2 * NUP];
                if (has_prior && NUP + RB * NUA <= 48) {   // (the prior rank's units ride with the first batch where the counter allows)
#pragma unroll
                    for (int u = 0; u < NUP; ++u) coop_ld16<L2>(sp, (NUA + u) * MOSHII_TPB + tid, pv[2 * u], pv[2 * u + 1]);
                }
                for (int r0 = 0; r0 < co.G; r0 += RB) {
                    double v[RB][4 * NT];
#pragma unroll
                    for (int u = 0; u < RB; ++u) {
                        const CoopSlot sr = coop_slot16(co, seq, min(r0 + u, co.G - 1));
#pragma unroll
                        for (int t = 0; t < NT; ++t)
                            if (4 * t + wv < NE) {
                                coop_ld16<L2>(sr, (2 * t) * MOSHII_TPB + tid, v[u][4 * t], v[u][4 * t + 1]);
                                coop_ld16<L2>(sr, (2 * t + 1) * MOSHII_TPB + tid, v[u][4 * t + 2], v[u][4 * t + 3]);
                            }
                    }
#pragma unroll
                    for (int u = 0; u < RB; ++u) {
                        if (r0 + u < co.G) {   // (uniform)
                            const bool mine_ = (r0 + u) == co.rank;
#pragma unroll
                            for (int t = 0; t < NT; ++t)
                                if (4 * t + wv < NE) {
#pragma unroll
                                    for (int i = 0; i < 4; ++i) acc.c[t][i] += mine_ ? own[4 * t + i] : v[u][4 * t + i];
                                }
                        }
                    }
                }
                if (has_prior) {
                    if (NUP + RB * NUA > 48) {
#pragma unroll
                        for (int u = 0; u < NUP; ++u) coop_ld16<L2>(sp, (NUA + u) * MOSHII_TPB + tid, pv[2 * u], pv[2 * u + 1]);
                    }
#pragma unroll
                    for (int e = 0; e < NE; ++e) pblk[e] = pv[e];
                    gqp = pv[NE];
                }
                }   // (all-gather form)
            }
        };
        exchange(std::false_type());   // (the L2-scope form -- MOSHII_COOP_L2SCOPE builds -- bought nothing: see coop_st_l2)
        if (tid < n) cx.dgn[tid] = gqp;   // (free here: the Gauss-Newton step of the last iteration has been used up)
        TRACE_STAMP(co, seq, 3);
        coop_end(cx, seq);
        PROF_LAP(32);
    }
    A.take(acc, cx.big);   // (the tile loop ended with a barrier: the Jacobian tile region is free)
    {   // data-term gradient g = -J^T r: row n of the product (its owners: ty == n % 16 in block row n / 16)
        const int ty = tid >> 4, tx = tid & 15, bn = n >> 4;
        if (ty == (n & 15)) {
#pragma unroll
            for (int bi = 0; bi < NBLK; ++bi)
#pragma unroll
                for (int bj = 0; bj <= bi; ++bj)
                    if (bi == bn && bj * 16 + tx < n) cx.g[bj * 16 + tx] = -A.a[bi * (bi + 1) / 2 + bj];
        }
        __syncthreads();
    }
    // structured terms: prior (dense, precomputed 0.5 L L^T per component), velocity and finger terms (diagonal)
    double* dvec = cx.y;   // diagonal additions
    const int kb = (np_ > 0) ? (int)cx.scal[S_KBEST] : 0;
    for (int q = tid; q < n; q += MOSHII_TPB) {
        double dg = 0.0, gq = 0.0;
        if (XT && q >= ncp) {   // shape column: regulariser (+ "stay" term), both diagonal
            const int e = q - ncp;
            const double sv_ = pose[md.NP + e];
            const double w2 = op.wt_shape * op.wt_shape;
            dg += w2; gq -= w2 * sv_;
            if (fp.has_stay) { const double y2 = op.wt_shape_stay * op.wt_shape_stay; dg += y2; gq -= y2 * (sv_ - cx.shp0[e]); }
        } else if (q >= 3) {
            const int pid = cx.colpid[q];
            if (fp.has_velo) { const double w2 = fp.wt_velo * fp.wt_velo; dg += w2; gq -= w2 * (pose[pid] - cx.vtarget[pid]); }
            const int pb = cx.colprior[q];
            if (COOP) { gq += cx.dgn[q]; }   // (the prior rank's column sums, received above)
            // (tried: these column sums formed at the top of the assembly by wavefronts 1..3 while wavefront 0 does the joints' Rodrigues
            //  derivatives, parked in cx.dgn: `structured` 16 -> 6 us per frame and T0 19 -> 30 -- the two round trips are longer than what they hid behind)
            else if (pb >= 0) {   // prior gradient w^2 (1/2 L L^T)(x - mu): column pb of the symmetric half-precision, b uniform
                const auto* Hk = pr.halfprec + (size_t)kb * np_ * np_ + pb;
                const auto* mu = pr.means + (size_t)kb * np_;
                double s0 = 0.0, s1 = 0.0;
                for (int b0 = 0; b0 < np_; b0 += 32) {   // 32 column entries in flight per lane (two round trips for 63)
                    double h[32];
#pragma unroll
                    for (int k = 0; k < 32; ++k) h[k] = Hk[(size_t)min(b0 + k, np_ - 1) * np_];
#pragma unroll
                    for (int k = 0; k < 32; k += 2) {
                        const int b = b0 + k;
                        s0 = fma(h[k] * ((b < np_) ? 1.0 : 0.0), cx.xb[min(b, np_ - 1)] - mu[min(b, np_ - 1)], s0);
                        s1 = fma(h[k + 1] * ((b + 1 < np_) ? 1.0 : 0.0), cx.xb[min(b + 1, np_ - 1)] - mu[min(b + 1, np_ - 1)], s1);
                    }
                }
                gq -= fp.wt_pose * fp.wt_pose * (s0 + s1);
            }
            if (fp.use_fingers) {
                // finger ids are a contiguous tail of the pose vector in every supported model
                if (op.nfinger > 0 && pid >= op.finger[0] && pid <= op.finger[op.nfinger - 1]) {
                    const double w2 = fp.wt_poseH * fp.wt_poseH; dg += w2; gq -= w2 * pose[pid];
                }
            }
            if constexpr (XT) {
                if (fp.use_face && op.nface > 0 && pid >= op.face[0] && pid <= op.face[op.nface - 1]) {   // contiguous ids
                    const double w2 = fp.wt_poseF * fp.wt_poseF; dg += w2; gq -= w2 * pose[pid];
                }
            }
        }
        dvec[q] = dg;
        cx.g[q] += gq;
    }
    __syncthreads();
    A.add_diag(dvec, n);
    if constexpr (COOP) {
#pragma unroll
        for (int e = 0; e < AReg<NBLK>::NE; ++e) A.a[e] += pblk[e];
    } else {
        if (np_ > 0) A.add_prior(fp.wt_pose * fp.wt_pose, pr.halfprec + (size_t)kb * np_ * np_, np_, cx.colprior, n);
    }
    __syncthreads();
    PROF_LAP(8);
}

__device__ __forceinline__ Ctx make_ctx(double* lds, const ChainLayout& ly) {
    Ctx cx;
    cx.pose = lds + ly.o_pose; cx.trans = lds + ly.o_trans; cx.pose_t = lds + ly.o_pose_t; cx.trans_t = lds + ly.o_trans_t;
    cx.pose_prev = lds + ly.o_pose_prev; cx.vtarget = lds + ly.o_vtarget; cx.fullpose = lds + ly.o_fullpose;
    cx.Jl = lds + ly.o_Jl; cx.feat = lds + ly.o_feat; cx.B = lds + ly.o_B; cx.omega = lds + ly.o_omega; cx.Rw = lds + ly.o_Rw; cx.tw = lds + ly.o_tw;
    cx.Rloc = lds + ly.o_Rloc; cx.acol = lds + ly.o_acol;
    cx.vshp = lds + ly.o_vshp; cx.shp0 = lds + ly.o_shp0;
    cx.vconst = lds + ly.o_vconst; cx.vposed = lds + ly.o_vposed; cx.vpos = lds + ly.o_vpos; cx.msim = lds + ly.o_msim; cx.res = lds + ly.o_res;
    cx.xb = lds + ly.o_xb; cx.ell = lds + ly.o_ell; cx.score = lds + ly.o_score; cx.px0 = lds + ly.o_px0; cx.ps0 = lds + ly.o_ps0;
    cx.g = lds + ly.o_g; cx.dsd = lds + ly.o_dsd; cx.dgn = lds + ly.o_dgn; cx.ddl = lds + ly.o_ddl; cx.y = lds + ly.o_y;
    cx.red = lds + ly.o_red; cx.scal = lds + ly.o_scal;
    cx.anc = reinterpret_cast<unsigned long long*>(lds + ly.o_anc);
    int* ints = reinterpret_cast<int*>(lds + ly.o_ints);
    cx.visidx = ints + ly.i_visidx; cx.colpid = ints + ly.i_colpid; cx.colprior = ints + ly.i_colprior;
    cx.pid2prior = ints + ly.i_pid2prior; cx.jointslot = ints + ly.i_jointslot; cx.kfree = ints + ly.i_kfree;
    cx.colq = ints + ly.i_colq; cx.ksum = ints + ly.i_ksum; cx.kconst = ints + ly.i_kconst;
    cx.big = lds + ly.o_big;
    cx.Jh = cx.big + ly.t_Jh; cx.Jrow = cx.big + ly.t_Jrow; cx.Lm = cx.big + ly.t_Lm; cx.Trot = cx.big + ly.t_Trot;
    cx.xjs = cx.big + ly.t_xjs; cx.rest = cx.big + ly.t_rest;
    cx.tjs = reinterpret_cast<int*>(cx.big + ly.t_tjs);

    return cx;
}

// ------------------------------------------------------------------------------------------------
// The two big phases as functions of their own.  Inlined into the frame loop they shared one register allocation with the
// whole solver state (the kernel then spilled several hundred scalar and vector registers, with reloads inside the hot
// loops); compiled separately each starts from its few arguments and fetches its context -- layout, model, prior,
// options, attachment -- from the KernelCtx copy at the head of the LDS, as wave-uniform (scalar) values.
// ------------------------------------------------------------------------------------------------
template <class T>
__device__ __forceinline__ T uniform_load(const T* p) {
    static_assert(sizeof(T) % 4 == 0, "word-sized descriptors only");
    T out;
    int* o = reinterpret_cast<int*>(&out);
    const int* q = reinterpret_cast<const int*>(p);
#pragma unroll
    for (int i = 0; i < (int)(sizeof(T) / 4); ++i) o[i] = __builtin_amdgcn_readfirstlane(q[i]);
    return out;
}

// The kernel's own arguments -- layout, model, prior, options: ~1.3 KB of which a phase uses a few hundred bytes -- are read where they
// already are, the kernel-argument segment, with SCALAR loads (the segment's address travels through the KernelCtx; a function cannot ask
// for it: __builtin_amdgcn_kernarg_segment_ptr() is null outside the kernel with this toolchain).  From the LDS copy every dword cost a
// ds_read lane and a v_readfirstlane: ~100 of them at the head of every call, ten calls a frame.  KArgsMirror = k_chain_solve's parameter
// list as the struct the argument segment is laid out as.
struct KArgsMirror { const ChainDev* chains; ModelDev md; PriorDev pr; OptsDev op; ChainLayout ly; int n_chains; };
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MOSHII_NO_KARGS)
typedef const KArgsMirror __attribute__((address_space(4)))* KArgsPtr;
__device__ __forceinline__ KArgsPtr kargs_of(const KernelCtx* kc) {
    const unsigned long long v = kc->kargs;
    return (KArgsPtr)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)v));
}
#define MOSHII_CTX_FROM_KARGS(kc) const KArgsPtr ka_ = kargs_of(kc); const ChainLayout ly = ka_->ly; const ModelDev md = ka_->md; const PriorDev pr = ka_->pr; const OptsDev op = ka_->op;
#else
#define MOSHII_CTX_FROM_KARGS(kc) const ChainLayout ly = uniform_load(&(kc)->ly); const ModelDev md = uniform_load(&(kc)->md); const PriorDev pr = uniform_load(&(kc)->pr); const OptsDev op = uniform_load(&(kc)->op);
#endif

// Cooperative assembly, large matrices (assemble(): the reduce-scatter + all-gather form), after the wait of exchange `seq` in which
// every rank posted its partial products: this rank sums its tiles [rank TPR, (rank + 1) TPR) over the ranks -- in rank order, from zero:
// the bits of the all-gather form --, posts the sums at the same offsets in exchange seq + 1 and waits for everybody's.  Slot to slot
// (no large arguments); leaves exchange seq + 1 open (the caller reads the sums and ends it).  false: the group is broken.
template <int NBLK>
__device__ __noinline__ bool coop_rs_reduce(unsigned seq_) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const KernelCtx* kc = reinterpret_cast<const KernelCtx*>(lds);
    const ChainLayout ly = uniform_load(&kc->ly);
    const Ctx cx = make_ctx(lds, ly);
    const CoopCtx co = uniform_load(&kc->co);
    constexpr int NT = JtJAcc<NBLK>::NT, NE = AReg<NBLK>::NE;
    const int tid = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned seq = (unsigned)__builtin_amdgcn_readfirstlane((int)seq_);
    const int TPR = (NT + co.G - 1) / co.G, tlo = co.rank * TPR, thi = min(NT, tlo + TPR);
    v4d sums[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
        if (t >= tlo && t < thi && 4 * t + wv < NE) {   // (uniform)
            double v[MOSHII_COOP_MAXG][4];
#pragma unroll
            for (int r = 0; r < MOSHII_COOP_MAXG; ++r)
                if (r < co.G) {
                    const CoopSlot sr = coop_slot16(co, seq, r);
                    coop_ld16(sr, (2 * t) * MOSHII_TPB + tid, v[r][0], v[r][1]);
                    coop_ld16(sr, (2 * t + 1) * MOSHII_TPB + tid, v[r][2], v[r][3]);
                }
            v4d sum = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int r = 0; r < MOSHII_COOP_MAXG; ++r)
                if (r < co.G) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) sum[i] += v[r][i];
                }
            sums[t] = sum;
        }
    coop_end(cx, seq);
    seq = coop_begin(cx);
    const CoopSlot mine2 = coop_slot16(co, seq, co.rank);
#pragma unroll
    for (int t = 0; t < NT; ++t)
        if (t >= tlo && t < thi && 4 * t + wv < NE) {
            coop_st16(mine2, (2 * t) * MOSHII_TPB + tid, sums[t][0], sums[t][1]);
            coop_st16(mine2, (2 * t + 1) * MOSHII_TPB + tid, sums[t][2], sums[t][3]);
        }
    return coop_publish_wait(co, cx, seq);
}

template <bool XT, bool COOP>
__device__ __noinline__ Sse eval_forward_fn(const uint8_t* visrow, int o_pose, int o_trans, int i_klist, int nk,
                                            int o_vbase, int light) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const KernelCtx* kc = reinterpret_cast<const KernelCtx*>(lds);
    MOSHII_CTX_FROM_KARGS(kc)
    const AttachDev at = uniform_load(&kc->at);
    const FrameParams fp = uniform_load(reinterpret_cast<const FrameParams*>(kc->fp_raw));
    const Ctx cx = make_ctx(lds, ly);
    const int* ints = reinterpret_cast<const int*>(lds + ly.o_ints);
    CoopCtx co = CoopCtx();
    if constexpr (COOP) co = uniform_load(&kc->co);
    return eval_forward<XT, COOP>(cx, md, at, pr, op, lds + __builtin_amdgcn_readfirstlane(o_pose), lds + __builtin_amdgcn_readfirstlane(o_trans),
                                  fp, visrow, ints + __builtin_amdgcn_readfirstlane(i_klist), __builtin_amdgcn_readfirstlane(nk),
                                  lds + __builtin_amdgcn_readfirstlane(o_vbase), __builtin_amdgcn_readfirstlane(light) != 0, co);
}

#ifdef MOSHII_ASM_INLINE
#define MOSHII_ASM_LINKAGE __forceinline__
#else
#define MOSHII_ASM_LINKAGE __noinline__
#endif
template <int NBLK, bool XT, bool COOP>
__device__ MOSHII_ASM_LINKAGE typename APass<NBLK, MOSHII_ASM_RET_VEC(NBLK)>::type assemble_fn(int o_pose, int n, int ncp, int nkf, int nfree_hand, double* qs) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const KernelCtx* kc = reinterpret_cast<const KernelCtx*>(lds);
    MOSHII_CTX_FROM_KARGS(kc)
    const AttachDev at = uniform_load(&kc->at);
    const FrameParams fp = uniform_load(reinterpret_cast<const FrameParams*>(kc->fp_raw));
    const Ctx cx = make_ctx(lds, ly);
    AReg<NBLK> A;
    CoopCtx co = CoopCtx();
    if constexpr (COOP) co = uniform_load(&kc->co);
    assemble<NBLK, XT, COOP>(cx, ly, md, at, pr, op, lds + __builtin_amdgcn_readfirstlane(o_pose), fp, __builtin_amdgcn_readfirstlane(n),
                             __builtin_amdgcn_readfirstlane(ncp), __builtin_amdgcn_readfirstlane(nkf), __builtin_amdgcn_readfirstlane(nfree_hand), qs, A, co);
    return APass<NBLK, MOSHII_ASM_RET_VEC(NBLK)>::pack(A);
}

// Arun/Procrustes rigid init (rigid_transformations.py:39-83), serial on one thread (first solved frame only).
__device__ __noinline__ void rigid_init_serial(const Ctx& cx, const FrameParams& fp, const uint8_t* visrow, int M) {
    double am[3] = {0, 0, 0}, bm[3] = {0, 0, 0};
    int cnt = 0;
    for (int m = 0; m < M; ++m) if (visrow[m]) {
        for (int i = 0; i < 3; ++i) { am[i] += cx.msim[m * 3 + i]; bm[i] += fp.obs[m * 3 + i]; }
        ++cnt;
    }
    for (int i = 0; i < 3; ++i) { am[i] /= cnt; bm[i] /= cnt; }
    double G[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};   // C = sum (a - am)(b - bm)^T
    for (int m = 0; m < M; ++m) if (visrow[m])
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j)
            G[i * 3 + j] += (cx.msim[m * 3 + i] - am[i]) * (fp.obs[m * 3 + j] - bm[j]);
    // one-sided Jacobi: G V = U S
    double V[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < 2; ++p) for (int q = p + 1; q < 3; ++q) {
            double al = 0, be = 0, ga = 0;
            for (int i = 0; i < 3; ++i) { al += G[i * 3 + p] * G[i * 3 + p]; be += G[i * 3 + q] * G[i * 3 + q]; ga += G[i * 3 + p] * G[i * 3 + q]; }
            if (fabs(ga) <= 1e-300 || fabs(ga) <= 1e-17 * sqrt(al * be)) continue;
            off = fmax(off, fabs(ga) / sqrt(al * be));
            const double zeta = (be - al) / (2.0 * ga);
            const double t = ((zeta >= 0) ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
            const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
            for (int i = 0; i < 3; ++i) {
                const double gp = G[i * 3 + p], gq = G[i * 3 + q];
                G[i * 3 + p] = c * gp - s * gq; G[i * 3 + q] = s * gp + c * gq;
                const double vp = V[i * 3 + p], vq = V[i * 3 + q];
                V[i * 3 + p] = c * vp - s * vq; V[i * 3 + q] = s * vp + c * vq;
            }
        }
        if (off < 1e-15) break;
    }
    double sv[3], U[9];
    int imin = 0, imax = 0;
    for (int c = 0; c < 3; ++c) {
        sv[c] = sqrt(G[0 * 3 + c] * G[0 * 3 + c] + G[1 * 3 + c] * G[1 * 3 + c] + G[2 * 3 + c] * G[2 * 3 + c]);
        if (sv[c] < sv[imin]) imin = c;
        if (sv[c] > sv[imax]) imax = c;
    }
    for (int c = 0; c < 3; ++c) for (int i = 0; i < 3; ++i) U[i * 3 + c] = (sv[c] > 0) ? G[i * 3 + c] / sv[c] : 0.0;
    if (sv[imin] <= 1e-13 * sv[imax]) {   // rank-deficient: complete the basis
        const int c1 = (imin + 1) % 3, c2 = (imin + 2) % 3;
        U[0 * 3 + imin] = U[1 * 3 + c1] * U[2 * 3 + c2] - U[2 * 3 + c1] * U[1 * 3 + c2];
        U[1 * 3 + imin] = U[2 * 3 + c1] * U[0 * 3 + c2] - U[0 * 3 + c1] * U[2 * 3 + c2];
        U[2 * 3 + imin] = U[0 * 3 + c1] * U[1 * 3 + c2] - U[1 * 3 + c1] * U[0 * 3 + c2];
    }
    // here C = U S V^T with U from `a`-side rows: R = V U^T maps a -> b
    double R[9];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j)
        R[i * 3 + j] = V[i * 3 + 0] * U[j * 3 + 0] + V[i * 3 + 1] * U[j * 3 + 1] + V[i * 3 + 2] * U[j * 3 + 2];
    const double det = R[0] * (R[4] * R[8] - R[5] * R[7]) - R[1] * (R[3] * R[8] - R[5] * R[6]) + R[2] * (R[3] * R[7] - R[4] * R[6]);
    if (det < 0.0)
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R[i * 3 + j] -= 2.0 * V[i * 3 + imin] * U[j * 3 + imin];
    // cv2.Rodrigues(R): matrix -> axis-angle, angle in [0, pi]
    double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
    const double s = sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
    double c = (R[0] + R[4] + R[8] - 1.0) * 0.5;
    c = fmin(1.0, fmax(-1.0, c));
    const double theta = acos(c);
    double rv[3];
    if (s < 1e-5) {
        if (c > 0) { rv[0] = rv[1] = rv[2] = 0.0; }
        else {
            double x = sqrt(fmax((R[0] + 1.0) * 0.5, 0.0));
            double y = sqrt(fmax((R[4] + 1.0) * 0.5, 0.0)) * ((R[1] >= 0) ? 1.0 : -1.0);
            double z = sqrt(fmax((R[8] + 1.0) * 0.5, 0.0)) * ((R[2] >= 0) ? 1.0 : -1.0);
            if (fabs(x) < fabs(y) && fabs(x) < fabs(z) && ((R[5] > 0) != (y * z > 0))) z = -z;
            const double nn = sqrt(x * x + y * y + z * z);
            rv[0] = x * theta / nn; rv[1] = y * theta / nn; rv[2] = z * theta / nn;
        }
    } else {
        const double k = 0.5 * theta / s;
        rv[0] = rx * k; rv[1] = ry * k; rv[2] = rz * k;
    }
    cx.pose[0] = rv[0]; cx.pose[1] = rv[1]; cx.pose[2] = rv[2];
    for (int i = 0; i < 3; ++i) cx.trans[i] = bm[i] - (R[i * 3 + 0] * am[0] + R[i * 3 + 1] * am[1] + R[i * 3 + 2] * am[2]);
}

// ------------------------------------------------------------------------------------------------
// One solver phase = one ch.minimize(method='dogleg') call over x = [trans, pose[ids]]
// (oracle/stageii_oracle.py:minimize_dogleg), written as a single loop around ONE forward evaluation and
// ONE normal-equation assembly so that each is instantiated once in the kernel.
//   rigid     : before the solve, evaluate the markers at the current state and apply the Procrustes init
//   eval_only : no solve; just evaluate every term at the current state (per-frame record)
// ------------------------------------------------------------------------------------------------
// XT: the unknowns are x = [trans, pose[ids], shape coefficients (nshp of them, Step 2 only)]; the shape block is stored
// behind the pose variables (cx.pose[NP ..]), so a shape column q has colpid[q] = NP + e and moves with the same code.
template <int NBLK, bool XT, bool COOP>
__device__ Sse run_phase(const Ctx& cx, const ChainLayout& ly, const ModelDev& md, const AttachDev& at, const PriorDev& pr,
                         const OptsDev& op, const FrameParams& fp, const uint8_t* visrow, MOSHII_GP(const int) ids, int nids, int nshp,
                         double* qs, double e3,
                         bool rigid, bool eval_only, bool reuse, Sse& carried, bool& at_pose, int& fwd_set, int set_id, int& vc_key,
                         int& tab_key, int& n_iter, int& n_fev, int& fail, const CoopCtx& co) {
    const int tid = threadIdx.x;
    const int ncp = 3 + nids;
    const int n = ncp + (XT ? nshp : 0);
    const int NPX = XT ? ly.NPX : md.NP;
    // column tables + needed-joint lists of this free set; they stay in LDS until a solve with another set replaces them
    // (body-only Stage-II uses one set throughout: built once per chain)
    if (!eval_only && tab_key != set_id) {
        tab_key = set_id;
        for (int i = tid; i < md.NP; i += MOSHII_TPB) cx.colq[i] = -1;
        __syncthreads();
        for (int q = tid; q < n; q += MOSHII_TPB) {
            const int pid = (q < 3) ? -1 : ((q < ncp) ? ids[q - 3] : md.NP + (q - ncp));
            cx.colpid[q] = pid;
            cx.colprior[q] = (pid >= 0 && pid < md.NP) ? cx.pid2prior[pid] : -1;
            if (pid >= 0 && pid < md.NP) cx.colq[pid] = q;
        }
        if (tid == 0) {
            for (int k = 0; k < md.K; ++k) cx.jointslot[k] = -1;
            for (int i = 0; i < nids; ++i) {
                const int pid = ids[i];
                if (pid < md.body_dof) cx.jointslot[pid / 3] = 0;
                else for (int k = md.body_dof / 3; k < md.K; ++k) cx.jointslot[k] = 0;
            }
            int c = 0;
            for (int k = 0; k < md.K; ++k) if (cx.jointslot[k] == 0) { cx.jointslot[k] = c; cx.kfree[c] = k; ++c; }
            cx.scal[S_TMP0] = (double)c;
            int nh = 0;   // ids are sorted, so the hand-PCA coefficients (pid >= body_dof) are the trailing columns
            for (int i = 0; i < nids; ++i) nh += (ids[i] >= md.body_dof) ? 1 : 0;
            cx.scal[S_TMP2] = (double)nh;
            // joints whose rotation moves during this solve (their pose correctives are re-summed at every evaluation)
            // and the rest (summed once into cx.vconst); the root has no correctives
            int ns = 0, nc = 0;
            for (int k = 1; k < md.K; ++k) { if (cx.jointslot[k] >= 0) cx.ksum[ns++] = k; else cx.kconst[nc++] = k; }
            cx.scal[S_TMP3] = (double)ns;
        }
        __syncthreads();
    }
    const int nkf = (int)cx.scal[S_TMP0], nfree_hand = (int)cx.scal[S_TMP2];
    const int nks = (int)cx.scal[S_TMP3];   // (eval-only phases reuse the lists of the solve that preceded them)
    const int nkc = md.K - 1 - nks;
    if (vc_key != set_id) {
        // cx.vconst = v_shaped + correctives of the joints that stay fixed in this solve, at the current pose.  It stays
        // valid until a solve with a different free set runs (body-only Stage-II: computed once per chain).
        for (int d = tid; d < md.P; d += MOSHII_TPB) cx.fullpose[d] = fullpose_entry(md, cx.pose, d);
        __syncthreads();
        if (tid < md.K) {
            double R[9];
            rodrigues_R(&cx.fullpose[3 * tid], R);
#pragma unroll
            for (int e = 0; e < 9; ++e) cx.feat[tid * 9 + e] = R[e] - ((e == 0 || e == 4 || e == 8) ? 1.0 : 0.0);
        }
        __syncthreads();
        if constexpr (COOP) posedirs_partial_range(cx, at, cx.kconst, nkc, at.vsh, cx.vconst, 3 * co.mlo, 3 * co.mhi);
        else posedirs_partial(cx, at, cx.kconst, nkc, at.vsh, cx.vconst);
        __syncthreads();
        vc_key = set_id;
    }
    AReg<NBLK> A;
    A.zero();
    // the phase's parameters go to the context block at the head of the LDS, where eval_forward_fn / assemble_fn pick them up
    // (as a by-value argument they travelled through the stack in scratch memory at every call)
    if (tid == 0) {
        extern __shared__ __attribute__((aligned(16))) double lds[];
        static_assert(sizeof(FrameParams) <= sizeof(KernelCtx::fp_raw), "KernelCtx::fp_raw too small");
        *reinterpret_cast<FrameParams*>(reinterpret_cast<KernelCtx*>(lds)->fp_raw) = fp;
    }
    for (int i = tid; i < NPX; i += MOSHII_TPB) cx.pose_t[i] = cx.pose[i];
    if (tid < 3) cx.trans_t[tid] = cx.trans[tid];
    __syncthreads();
    Sse last;
    double sse = 0.0, delta = op.delta0, norm_sd = 0.0, norm_gn = 0.0, step = 0.0, p2 = 0.0, gd = 0.0;
    double dAd = 0.0, dl_dd = 0.0, dl_gs = 0.0, dl_ds = 0.0;   // d^T A d of the current step; the dogleg leg's sums of the current solve
    bool init = true, done = false, have_gn = false;
    int iteration = 0;
    // `reuse`: the previous phase of this frame ended with the forward state of the CURRENT point in LDS and the same
    // residual weights, so its evaluation (what ch.minimize recomputes on entry) is carried over instead of redone.
    bool skip_eval = reuse && !rigid;
    // at_pose && fwd_set == set_id: the forward state in LDS is that of the current point, left by an evaluation with this
    // free set's joint lists -- the previous frame's last one, or the previous phase's under other weights.  The entry
    // evaluation then only redoes what depends on the frame's data and weights (eval_forward's `light`).
    int light = (!rigid && at_pose && fwd_set == set_id) ? 1 : 0;
    while (true) {
        int tid = threadIdx.x; MOSHII_OPAQUE(tid);
        if (skip_eval) { last = carried; skip_eval = false; }
        else { PROF_T(_te); last = eval_forward_fn<XT, COOP>(visrow, ly.o_pose_t, ly.o_trans_t, ly.i_ksum, nks, ly.o_vconst, light); PROF_ACC(15, _te); }
        if constexpr (COOP) { if (cx.scal[S_COOP_FAIL] != 0.0) break; }   // the group is broken: unwind (the host reports the launch as failed)
        if (!(last.total == last.total)) { fail = 1; break; }   // a NaN objective: no comparison below would ever end the loop
        light = 0;
        at_pose = true;   // cleared below when a trial point is rejected
        fwd_set = set_id;
        if (rigid) {   // rigid_transformations.py:72-83 on the markers just simulated
            if constexpr (COOP) {   // every rank needs all the simulated markers: gather the ranks' rows (its own stay as they are)
                const unsigned seq = coop_begin(cx);
                auto gather = [&](auto l2) {   // (l2: the access scope, see coop_st_l2)
                    constexpr bool L2 = decltype(l2)::value;
                    auto* mine = coop_slot(co, seq, co.rank);
                    for (int i = 3 * co.mlo + tid; i < 3 * co.mhi; i += MOSHII_TPB) coop_st64<L2>(mine + i, f64_bits(cx.msim[i]));
                    if (tid == 0) coop_st64<L2>(mine + 3 * at.M, f64_bits((double)co.mlo));   // (where this rank's range starts)
                    if (coop_publish_wait<L2>(co, cx, seq)) {
                        const int M3 = 3 * at.M;
                        for (int i = tid; i < M3; i += MOSHII_TPB) {
                            // owner of marker i / 3: the rank whose range holds it (ranges are contiguous and ascending: count the boundaries passed)
                            int r = 0;
                            for (int q = 1; q < co.G; ++q) r += (i >= 3 * (int)bits_f64(coop_ld64<L2>(coop_slot(co, seq, q) + 3 * at.M))) ? 1 : 0;
                            if (r != co.rank) cx.msim[i] = bits_f64(coop_ld64<L2>(coop_slot(co, seq, r) + i));
                        }
                    }
                };
                gather(std::false_type());
                coop_end(cx, seq);
            }
            if (tid == 0) rigid_init_serial(cx, fp, visrow, at.M);
            __syncthreads();
            for (int i = tid; i < NPX; i += MOSHII_TPB) cx.pose_t[i] = cx.pose[i];
            if (tid < 3) cx.trans_t[tid] = cx.trans[tid];
            __syncthreads();
            rigid = false;
            continue;
        }
        if (eval_only) break;
        ++n_fev;
        PROF_T(_t1);
        bool improved = false, do_assemble = false;
        double rho = 0.0;
        if (init) {
            sse = uni64(last.total);
            do_assemble = true;
        } else {
            rho = sse - last.total;
            if (rho > 0.0) rho = rho / (2.0 * gd - dAd);   // (d^T A d: reduced with the step's norms when the trial point was formed)
            improved = rho > 0.0;
            at_pose = improved;
            if (improved) {
                for (int i = tid; i < NPX; i += MOSHII_TPB) cx.pose[i] = cx.pose_t[i];
                if (tid < 3) cx.trans[tid] = cx.trans_t[tid];
                __syncthreads();
                if (e3 > 0.0 && (sse - last.total) / sse < e3) done = true;
                else { do_assemble = true; sse = uni64(last.total); }
            }
        }
        PROF_ACC(23, _t1);
        double pp = 0.0, gg = 0.0, gAg = 0.0;
        if (do_assemble) {
            { PROF_T(_ta); A = APass<NBLK, MOSHII_ASM_RET_VEC(NBLK)>::unpack(assemble_fn<NBLK, XT, COOP>(ly.o_pose, n, ncp, nkf, nfree_hand, qs)); PROF_ACC(16, _ta); }
            if constexpr (COOP) { if (cx.scal[S_COOP_FAIL] != 0.0) break; }
            PROF_T(_t2);
            // the gradient's maximum, |p|^2 at the accepted point, |g|^2 and g^T A g in ONE block reduction
            double gm = 0.0;
            for (int q = tid; q < n; q += MOSHII_TPB) {
                const double gq = cx.g[q];
                const double pq = (q < 3) ? cx.trans[q] : cx.pose[cx.colpid[q]];
                gm = fmax(gm, fabs(gq)); pp += pq * pq; gg += gq * gq;
            }
            gAg = A.quad_partial(cx.g, n, tid);
            block_max_sum3(gm, pp, gg, gAg, cx.red);
            gm = uni64(gm); pp = uni64(pp); gg = uni64(gg); gAg = uni64(gAg);
            if (gm < 1e-15) done = true;
            PROF_ACC(24, _t2);
        }
        PROF_T(_t3);
        if (!init) {   // updateRadius + trust-region floor
            const double pnorm2 = (improved && do_assemble) ? pp : p2;   // (improved without an assembly: the e_3 stop -- done already)
            if (rho > 0.9) delta = fmax(delta, 2.5 * step);
            else if (rho < 0.05) delta *= 0.25;
            delta = uni64(delta);
            if (delta <= 1e-15 * sqrt(pnorm2)) done = true;
        }
        if (init || improved) {
            if (!init && iteration >= op.maxiter) done = true;
            if (done) break;
            // start_iteration: d_sd = |g|^2 / |J g|^2 g, with |J g|^2 = g^T A g
            ++iteration;
            const double csd = gg / gAg;
            norm_sd = uni64(fabs(csd) * sqrt(gg));
            for (int q = tid; q < n; q += MOSHII_TPB) cx.dsd[q] = csd * cx.g[q];
            have_gn = false;
            __syncthreads();
        } else if (done) {
            break;
        }
        init = false;
        PROF_ACC(25, _t3);
        PROF_T(_t4);
        // ---- update_step
        if (norm_sd >= delta) {
            const double sc = delta / norm_sd;
            for (int q = tid; q < n; q += MOSHII_TPB) cx.ddl[q] = sc * cx.dsd[q];
        } else {
            if (!have_gn) {
                // (BIG: factor in the chain's global scratch behind the shape-derivative arrays, LDS part at the head of `big`)
                constexpr bool BIG = XT && NBLK > 8;
                bool solved;
                PROF_T(_ts);
                if constexpr (BIG) solved = ldl_big<NBLK>(A, qs + (size_t)6 * md.K * op.nshape, cx.big, cx.g, cx.dgn, cx.y, n);
                else solved = ldl_solve<NBLK>(A, ly.o_big, ly.o_g, ly.o_dgn, ly.o_y, n);
                PROF_ACC(17, _ts);
                PROF_ACC(26, _t4);   // (includes the solve: subtract slot 17)
                if (!solved) {
                    fail = 1;
                    for (int q = tid; q < n; q += MOSHII_TPB) cx.dgn[q] = cx.dsd[q];
                    __syncthreads();
                }
                // |d_gn|^2 and the three sums of the dogleg leg, once per solve (a rejected step re-uses them with a smaller radius)
                double sg = 0.0;
                dl_dd = 0.0; dl_gs = 0.0; dl_ds = 0.0;
                for (int q = tid; q < n; q += MOSHII_TPB) {
                    const double dg_ = cx.dgn[q], dsq = cx.dsd[q], df = dg_ - dsq;
                    sg += dg_ * dg_; dl_dd += df * df; dl_gs += dg_ * dsq; dl_ds += df * dsq;
                }
                block_sum4(sg, dl_dd, dl_gs, dl_ds, cx.red);
                dl_dd = uni64(dl_dd); dl_gs = uni64(dl_gs); dl_ds = uni64(dl_ds);
                norm_gn = uni64(sqrt(sg));
                have_gn = true;
            }
            if (norm_gn <= delta) {
                for (int q = tid; q < n; q += MOSHII_TPB) cx.ddl[q] = cx.dgn[q];
            } else {
                const double delta_sq = delta * delta;
                const double sqnorm_sd = norm_sd * norm_sd;
                const double pnow = dl_dd * delta_sq + dl_gs * dl_gs - (norm_gn * norm_gn) * sqnorm_sd;
                const double beta = (delta_sq - sqnorm_sd) / (dl_ds + sqrt(pnow));
                for (int q = tid; q < n; q += MOSHII_TPB) cx.ddl[q] = cx.dsd[q] + beta * (cx.dgn[q] - cx.dsd[q]);
            }
        }
        __syncthreads();
        PROF_T(_t5);
        // ---- trial point and norms (+ d^T A d for the gain ratio of this step)
        double s2 = 0.0;
        p2 = 0.0; gd = 0.0;
        for (int q = tid; q < n; q += MOSHII_TPB) {
            const double dq = cx.ddl[q];
            const double pq = (q < 3) ? cx.trans[q] : cx.pose[cx.colpid[q]];
            s2 += dq * dq; p2 += pq * pq; gd += cx.g[q] * dq;
        }
        dAd = A.quad_partial(cx.ddl, n, tid);
        block_sum4(s2, p2, gd, dAd, cx.red);
        p2 = uni64(p2); gd = uni64(gd); dAd = uni64(dAd);
        step = uni64(sqrt(s2));
        if (step <= 1e-15 * sqrt(p2)) break;   // "small step size" stop
        for (int i = tid; i < NPX; i += MOSHII_TPB) cx.pose_t[i] = cx.pose[i];
        if (tid < 3) cx.trans_t[tid] = cx.trans[tid] + cx.ddl[tid];
        __syncthreads();
        for (int q = 3 + tid; q < n; q += MOSHII_TPB) cx.pose_t[cx.colpid[q]] += cx.ddl[q];
        __syncthreads();
        PROF_ACC(27, _t5);
    }
    n_iter += iteration;
    carried = last;
    return last;
}


// MINW = 1: one workgroup per CU, the compiler may use the whole 512-entry register file (lowest latency per chain);
// MINW = 2: registers capped at 256 so that two workgroups share a CU and cover each other's dependency stalls
//           (+6 % aggregate when there are more chains than CUs, 1.9x slower per chain: not instantiated since round 3).
// XT = true: the extended variant with the Step-2 extras of chmosh.py:685-699 (jaw term, free shape block); kept out of
//           the plain instantiations so that their register allocation and timings are untouched.
// COOP = true: cooperative chains -- G workgroups per chain (ChainDev::coop; see moshii_dev.h).  Block b is rank (b / 8) % G of chain
//           8 ((b / 8) / G) + b % 8: the ranks of a chain sit on blocks that are congruent modulo 8, i.e. (as blocks are observed to be dealt
//           to the XCDs round-robin) on CUs that share an L2 -- a matter of speed only.  A cooperative chain may be a repair chain of a
//           chunked solve (moshii_sequence_solve's host rounds): rank 0 alone reads the baton / boundary / re-join words other chains
//           write and sends its verdict to the other ranks at the top of every frame; the carry-on protocol of pass 1 (fuse_F) is the
//           plain chains'.  Rank 0 writes the per-frame rows, every rank its own simulated markers.
template <int NBLK, int MINW, bool XT, bool COOP = false>
__global__ __launch_bounds__(MOSHII_TPB, MINW) void k_chain_solve(const ChainDev* __restrict__ chains, ModelDev md, PriorDev pr,
                                                             OptsDev op, ChainLayout ly, int n_chains) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x;
    int chain_id = blockIdx.x;
    CoopCtx co = CoopCtx();
    if constexpr (COOP) {
        const int G = chains[0].coop.G;   // (the same for every chain of a launch)
        const int q = blockIdx.x >> 3;
        chain_id = 8 * (q / G) + (blockIdx.x & 7);
        if (chain_id >= n_chains) return;
        const ChainDev* c0 = chains + chain_id;
        co.G = G; co.rank = q % G; co.prior_rank = c0->coop.prior_rank;
        co.mlo = c0->coop.mlo[co.rank]; co.mhi = c0->coop.mlo[co.rank + 1];
        co.slot_doubles = c0->coop.slot_doubles; co.slots = c0->coop.slots; co.flags = c0->coop.flags; co.skew = c0->coop.skew;
    }
    const bool lead = !COOP || co.rank == 0;   // writes the rows every rank holds alike
    const ChainDev* chp = chains + chain_id;   // fields are (re)loaded where used: keeps SGPR pressure down
    const AttachDev at = *chp->att;
    const Ctx cx = make_ctx(lds, ly);
    if (tid == 0) {   // the descriptors for the separately compiled phases (eval_forward_fn / assemble_fn)
        KernelCtx* kc = reinterpret_cast<KernelCtx*>(lds);
        kc->ly = ly; kc->md = md; kc->pr = pr; kc->op = op; kc->at = at;
#if defined(__HIP_DEVICE_COMPILE__)
        kc->kargs = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
#if !defined(MOSHII_NO_KARGS)
        {   // KArgsMirror must BE this kernel's parameter list: a parameter added here and not there would hand the phase functions garbage.
            // (The parameter list as that one struct -- the same by construction -- does not compile: "Illegal instruction detected: Operand
            //  has incorrect register class" in one of the instantiations with hipcc 7.2.)  First, last and two middle fields against the arguments:
            const KArgsPtr kk = (KArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();
            if (kk->chains != chains || kk->n_chains != n_chains || kk->ly.total_doubles != ly.total_doubles || kk->md.V != md.V || kk->op.maxiter != op.maxiter)
                __builtin_trap();   // (the launch fails: moshii_chain_solve reports the HIP error)
        }
#endif
#endif
        if constexpr (COOP) { kc->co = co; cx.scal[S_COOP_SEQ] = 0.0; cx.scal[S_COOP_FAIL] = 0.0; }
    }

    const int NP = md.NP, M = at.M, F = chp->F, skip = chp->skip;
    bool has_prev, first;
    {
        const double* is = chp->init_state;
        const double* ip = chp->init_pose;
        const double* iv = chp->init_prev;
        const double* it = chp->init_trans;
        for (int i = tid; i < NP; i += MOSHII_TPB) {
            cx.pose[i] = is ? is[i] : (ip ? ip[i] : 0.0);
            cx.pose_prev[i] = is ? is[NP + i] : (iv ? iv[i] : 0.0);
            cx.vtarget[i] = 0.0;
            cx.pid2prior[i] = -1;
        }
        if (tid < 3) cx.trans[tid] = is ? is[2 * NP + tid] : (it ? it[tid] : 0.0);
        if constexpr (XT) {
            const double* ish = chp->init_shape;
            // (chunk hand-off states carry the coefficients behind the flags: [pose][pose_prev][trans][has_prev][first][shape])
            for (int e = tid; e < op.nshape; e += MOSHII_TPB) { cx.pose[NP + e] = is ? is[2 * NP + 5 + e] : (ish ? ish[e] : 0.0); cx.shp0[e] = 0.0; }
        }
        has_prev = is ? (is[2 * NP + 3] != 0.0) : (iv != nullptr);
        first = is ? (is[2 * NP + 4] != 0.0) : (chp->first != 0);
    }
    if (tid == 0) cx.scal[S_PRIOR_REF] = 0.0;   // no prior reference point yet (eval_forward)
    // the prior's partial-sum slices and the slack behind them are read past a wave's own entries (times 0) before the neighbouring
    // wave has necessarily written them on the first evaluation: keep every word finite from the start (stale LDS bits could be NaN)
    for (int i = tid; i < 8 * pr.npose + 16; i += MOSHII_TPB) cx.ell[i] = 0.0;
    if (tid < md.K) cx.anc[tid] = md.anc[tid];
    for (int i = tid; i < 3 * md.K; i += MOSHII_TPB) cx.Jl[i] = md.J[i];   // regressed joints: read in every phase, keep them in LDS
    __syncthreads();
    for (int b = tid; b < op.nbody; b += MOSHII_TPB) cx.pid2prior[op.body[b]] = b;
    __syncthreads();
    int vc_key = 0, tab_key = 0;   // which free set cx.vconst / the column tables were built for (0: none yet)
    int rejoin_run = 0;            // consecutive frames that reproduced the stored trajectory (repair chains)
    int bi_next = 0;               // next chunk boundary of a run-through repair chain
    bool fuse_on = false;          // a pass-1 chain that has carried on as the repair chain of the next chunk (ChainDev::fuse_F)
    bool at_pose = false;          // the forward state in LDS is that of (cx.pose, cx.trans) ...
    int fwd_set = 0;               // ... evaluated with the joint lists of this free set (run_phase)
    PROF_BEGIN();
#ifdef MOSHII_PROFILE
    const long long _wall0 = wall_clock64();
#endif

    bool tail_cut = false;
    for (int t = 0; t <= F; ++t) {
        PROF_T(_tf);
        if constexpr (!COOP && !XT) {   // (armed for body / finger solves only: moshii_sequence_solve runs extended solves without cooperative sweeps)
            // the tail of a pass-1 launch (ChainDev::tail_done): most of the other chunks are done and this one is far from it -- give the chunk up
            if (chp->tail_done != nullptr && chp->tail_can_cut != 0 && t < F && F - t > chp->tail_left) {
                if (tid == 0) cx.scal[S_ABORT] = (double)__hip_atomic_load(chp->tail_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __syncthreads();
                const bool cut = cx.scal[S_ABORT] >= (double)chp->tail_quota;
                __syncthreads();
                if (cut) {
                    // finite states whose flag word matches nothing: this chunk and the next fail the verification
                    auto spoil = [&](double* so, double flag) {
                        for (int i = tid; i < NP; i += MOSHII_TPB) { so[i] = cx.pose[i]; so[NP + i] = cx.pose_prev[i]; }
                        if (tid < 3) so[2 * NP + tid] = cx.trans[tid];
                        if (tid == 3) so[2 * NP + 3] = flag;
                        if (tid == 4) so[2 * NP + 4] = 0.0;
                    };
                    spoil(chp->entry_state, -1.0);      // (both slots exist for every chain of a chunked solve's first launch)
                    spoil(chp->final_state, -2.0);
                    if (tid == 0 && chp->tail_mark != nullptr) *chp->tail_mark = 0x7fffffff;   // (no sweep re-joins inside this chunk: its rows are a mix)
                    tail_cut = true;
                    break;
                }
            }
        }
        if constexpr (COOP) {
            // A cooperative REPAIR chain of a chunked solve (moshii_sequence_solve's host rounds).  Everything that depends on memory other
            // chains write -- the stop request of an upstream sweep, the negotiation at a chunk boundary, whether the last two frames
            // reproduced the stored rows -- is looked at by rank 0 alone, with the plain chain's code, and its verdict for this frame goes
            // to the other ranks in one small exchange: ranks that read such words for themselves could part ways.
            if (chp->baton != nullptr || chp->nb > 0 || chp->rejoin_tol > 0.0) {   // (descriptor constants: the same on every rank)
                bool halt = false;
                if (lead) {
                    const int S = 2 * NP + 5;
                    if (rejoin_run >= 2 && t < F) {   // the previous frame's verdict: pose and pose_prev both match, the stored rows (and final state) stand
                        if (tid == 0 && chp->frames_done) *chp->frames_done = t;
                        halt = true;
                    }
                    if (!halt && bi_next < chp->nb && t == chp->bnd[bi_next] - chp->bnd_off) {   // a chunk boundary (see the plain chain below)
                        double* s1 = chp->run_final + (size_t)bi_next * S;
                        for (int i = tid; i < NP; i += MOSHII_TPB) { s1[i] = cx.pose[i]; s1[NP + i] = cx.pose_prev[i]; }
                        if (tid < 3) s1[2 * NP + tid] = cx.trans[tid];
                        if (tid == 3) s1[2 * NP + 3] = has_prev ? 1.0 : 0.0;
                        if (tid == 4) s1[2 * NP + 4] = first ? 1.0 : 0.0;
                        if (chp->baton != nullptr) {
                            if (tid == 0) {
                                int* st = chp->baton + 2 * (chp->chunk0 + 1 + bi_next);
                                double go = 1.0;
                                if (__hip_atomic_load(st, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == 1) {
                                    __hip_atomic_store(st + 1, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                                    int spins = 0;
                                    while (__hip_atomic_load(st, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != 2) {
                                        __builtin_amdgcn_s_sleep(64);
                                        if (++spins > 20000) { go = 0.0; break; }
                                    }
                                }
                                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                                cx.scal[S_BATON] = go;
                            }
                            __syncthreads();
                            if (cx.scal[S_BATON] == 0.0) { if (tid == 0 && chp->frames_done) *chp->frames_done = t; halt = true; }
                            __syncthreads();
                        }
                        if (!halt) {
                            double* s2 = chp->run_entry + (size_t)(bi_next + 1) * S;
                            for (int i = tid; i < NP; i += MOSHII_TPB) { s2[i] = cx.pose[i]; s2[NP + i] = cx.pose_prev[i]; }
                            if (tid < 3) s2[2 * NP + tid] = cx.trans[tid];
                            if (tid == 3) s2[2 * NP + 3] = has_prev ? 1.0 : 0.0;
                            if (tid == 4) s2[2 * NP + 4] = first ? 1.0 : 0.0;
                            ++bi_next;
                        }
                    }
                    if (!halt && chp->baton != nullptr && t < F) {   // has an upstream chain of this round asked for this chain's territory?
                        if (tid == 0) cx.scal[S_ABORT] = (double)__hip_atomic_load(&chp->baton[2 * chp->chunk0 + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __syncthreads();
                        if (cx.scal[S_ABORT] != 0.0) {
                            if (tid == 0) {
                                chp->run_entry[(size_t)bi_next * S + 2 * NP + 3] = -1.0;
                                int* mark = chp->abort_at + chp->chunk0 + bi_next;
                                if (*mark < chp->bnd_off + t) *mark = chp->bnd_off + t;
                                if (chp->frames_done) *chp->frames_done = -t - 1;
                            }
                            halt = true;
                        }
                        __syncthreads();
                    }
                }
                const unsigned seq = coop_begin(cx);
                const double mine[1] = {halt ? 1.0 : 0.0};
                const bool okx = coop_exchange_small<1>(co, cx, seq, mine, cx.y);
                const double verdict = okx ? cx.y[0] : 1.0;   // (rank 0's word)
                coop_end(cx, seq);
                if (verdict != 0.0) break;
            }
        }
        // chunk hand-off states: moshii_sequence_solve checks a chunk's entry state against its predecessor's final one
        for (int which = 0; which < 2; ++which) {
            double* so = (which == 0) ? ((t == skip) ? chp->entry_state : nullptr) : ((t == F) ? chp->final_state : nullptr);
            if (so == nullptr || !lead) continue;
            for (int i = tid; i < NP; i += MOSHII_TPB) { so[i] = cx.pose[i]; so[NP + i] = cx.pose_prev[i]; }
            if (tid < 3) so[2 * NP + tid] = cx.trans[tid];
            if (tid == 3) so[2 * NP + 3] = has_prev ? 1.0 : 0.0;
            if (tid == 4) so[2 * NP + 4] = first ? 1.0 : 0.0;
            if constexpr (XT) for (int e = tid; e < op.nshape; e += MOSHII_TPB) so[2 * NP + 5 + e] = cx.pose[NP + e];
        }
        if (!COOP && chp->fuse_flags != nullptr) {
            const int S = 2 * NP + 5 + (XT ? op.nshape : 0);
            if (t == skip && chp->entry_state != nullptr) {   // the entry state is out: the left neighbour may compare with it
                __syncthreads();
                if (tid == 0) { __threadfence(); __hip_atomic_store(&chp->fuse_flags[3 * chp->fuse_c], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
            }
            if (chp->fuse_F == 0 && t == F) {                  // (a chain without a right neighbour: its rows are complete ...
                __syncthreads();
                if (tid == 0) {
                    __threadfence();
                    __hip_atomic_store(&chp->fuse_flags[3 * chp->fuse_c + 1], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                    // ... and it never carries on: its verdict is final too.  The chain of the chunk before it waits for BOTH flags
                    // before it takes this chunk over; without this store it waited out its whole patience and left the
                    // last chunk of every sequence to a host round.)
                    __hip_atomic_store(&chp->fuse_flags[3 * chp->fuse_c + 2], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            if (chp->fuse_F > 0 && t == chp->fuse_F) {
                // The end of this chain's own chunk (ChainDev::fuse_F): its end state ...
                double* sf = chp->run_final - S;
                for (int i = tid; i < NP; i += MOSHII_TPB) { sf[i] = cx.pose[i]; sf[NP + i] = cx.pose_prev[i]; }
                if (tid < 3) sf[2 * NP + tid] = cx.trans[tid];
                if (tid == 3) sf[2 * NP + 3] = has_prev ? 1.0 : 0.0;
                if (tid == 4) sf[2 * NP + 4] = first ? 1.0 : 0.0;
                if constexpr (XT) for (int e = tid; e < op.nshape; e += MOSHII_TPB) sf[2 * NP + 5 + e] = cx.pose[NP + e];
                __syncthreads();
                if (tid == 0) { __threadfence(); __hip_atomic_store(&chp->fuse_flags[3 * chp->fuse_c + 1], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }   // this chunk's rows are complete
                // ... against the entry state the next chunk's chain recorded a chunk ago (it only has to be waited for when the
                // chunks are not all resident -- then this chain gives up quickly and the host's rounds do the rest)
                bool carry = false;
                auto state_dev = [&](const double* en) {   // max |this chain's state - en| (flags must be equal; NaN counts as a miss)
                    double dv = 0.0, nn = 0.0;
                    for (int i = tid; i < NP; i += MOSHII_TPB) { dv = fmax(dv, fmax(fabs(cx.pose[i] - en[i]), fabs(cx.pose_prev[i] - en[NP + i]))); nn += en[i] + en[NP + i]; }
                    if (tid < 3) dv = fmax(dv, fabs(cx.trans[tid] - en[2 * NP + tid]));
                    if (tid == 3 && (has_prev ? 1.0 : 0.0) != en[2 * NP + 3]) dv = 1e300;
                    if (tid == 4 && (first ? 1.0 : 0.0) != en[2 * NP + 4]) dv = 1e300;
                    if constexpr (XT) for (int e = tid; e < op.nshape; e += MOSHII_TPB) dv = fmax(dv, fabs(cx.pose[NP + e] - en[2 * NP + 5 + e]));
                    if (!(nn == nn)) dv = 1e300;
                    return block_max(dv, cx.red);
                };
                auto wait_flag = [&](int* fl, int patience) {   // thread 0 waits (bounded) for *fl == 1; every thread gets the outcome
                    if (tid == 0) {
                        double go = 1.0;
                        int spins = 0;
                        while (__hip_atomic_load(fl, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != 1) {
                            __builtin_amdgcn_s_sleep(64);
                            if (++spins > patience) { go = 0.0; break; }
                        }
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                        cx.scal[S_BATON] = go;
                    }
                    __syncthreads();
                    const bool ok_ = cx.scal[S_BATON] != 0.0;
                    __syncthreads();
                    return ok_;
                };
                if (chp->fuse_has_next && wait_flag(chp->fuse_flags + 3 * (chp->fuse_c + 1), 2000)) {
                    const double dv = state_dev(chp->run_entry);
                    carry = !(dv <= chp->fuse_tol);
                    // A gross miss (another basin) is only this chain's to repair if its OWN chunk stands on firm ground: if the hand-off
                    // INTO this chunk missed grossly too, this chain's state is the suspect one and the sweep that repairs this chunk
                    // will deal with the next boundary when it gets there (the rule the host applies to its repair rounds).  The left
                    // neighbour finishes at about the same time; if it has not, the chain goes ahead.
                    if (carry && dv > 1e-6 && chp->fuse_has_prev && wait_flag(chp->fuse_flags + 3 * (chp->fuse_c - 1) + 1, 150)) {
                        const double* lf = chp->run_final - 2 * S;   // end state of the left neighbour
                        const double* oe = chp->run_entry - S;       // the entry state this chain recorded
                        double dl = 0.0;
                        for (int i = tid; i < 2 * NP + 3; i += MOSHII_TPB) dl = fmax(dl, fabs(lf[i] - oe[i]));
                        dl = block_max(dl, cx.red);
                        if (!(dl <= 1e-6)) carry = false;
                    }
                }
                // The verdict (a chain will start at the next chunk, or not) is in place before it is declared final: a sweep
                // arriving at the next boundary waits for that declaration before it looks.
                if (carry && tid == 0) __hip_atomic_store(&chp->baton[2 * chp->chunk0], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                __syncthreads();
                if (tid == 0) { __threadfence(); __hip_atomic_store(&chp->fuse_flags[3 * chp->fuse_c + 2], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
                if (!carry) { if (tid == 0 && chp->frames_done) *chp->frames_done = t; break; }
                // Carry on as the repair chain of the next chunk -- once that chunk's own chain has written its last row (from here on
                // its rows are compared against and replaced), unless an upstream sweep asks for this territory meanwhile.
                if (tid == 0) {
                    int* fl = chp->fuse_flags + 3 * (chp->fuse_c + 1) + 1;
                    double go = 1.0;
                    int spins = 0;
                    // (... AND has given its own verdict: until then it may still read the entry state it recorded -- the slot written
                    //  below -- for its "is my own chunk on firm ground" test; overwriting it earlier made that test compare this
                    //  chain's end state with itself)
                    while (__hip_atomic_load(fl, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != 1 ||
                           __hip_atomic_load(fl + 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != 1) {
                        if (__hip_atomic_load(&chp->baton[2 * chp->chunk0 + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { go = 0.0; break; }
                        __builtin_amdgcn_s_sleep(64);
                        if (++spins > 20000) { go = 0.0; break; }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    if (go != 0.0 && chp->fuse_count != nullptr) atomicAdd(chp->fuse_count, 1);
                    cx.scal[S_BATON] = go;
                }
                __syncthreads();
                if (cx.scal[S_BATON] == 0.0) { if (tid == 0 && chp->frames_done) *chp->frames_done = t; break; }
                double* s2 = chp->run_entry;   // the next chunk is entered with this state
                for (int i = tid; i < NP; i += MOSHII_TPB) { s2[i] = cx.pose[i]; s2[NP + i] = cx.pose_prev[i]; }
                if (tid < 3) s2[2 * NP + tid] = cx.trans[tid];
                if (tid == 3) s2[2 * NP + 3] = has_prev ? 1.0 : 0.0;
                if (tid == 4) s2[2 * NP + 4] = first ? 1.0 : 0.0;
                if constexpr (XT) for (int e = tid; e < op.nshape; e += MOSHII_TPB) s2[2 * NP + 5 + e] = cx.pose[NP + e];
                fuse_on = true;
            }
        }
        const bool sweeping = chp->fuse_F == 0 || fuse_on;   // (a pass-1 chain inside its own chunk is not a repair chain yet)
        if (!COOP && bi_next < chp->nb && t == chp->bnd[bi_next] - chp->bnd_off) {   // a chunk boundary inside a run-through repair chain
            const int S = 2 * NP + 5 + (XT ? op.nshape : 0);
            double* s1 = chp->run_final + (size_t)bi_next * S;          // end state of the chunk just left ...
            for (int i = tid; i < NP; i += MOSHII_TPB) { s1[i] = cx.pose[i]; s1[NP + i] = cx.pose_prev[i]; }
            if (tid < 3) s1[2 * NP + tid] = cx.trans[tid];
            if (tid == 3) s1[2 * NP + 3] = has_prev ? 1.0 : 0.0;
            if (tid == 4) s1[2 * NP + 4] = first ? 1.0 : 0.0;
            if constexpr (XT) for (int e = tid; e < op.nshape; e += MOSHII_TPB) s1[2 * NP + 5 + e] = cx.pose[NP + e];
            if (chp->baton != nullptr) {
                // Did a chain of this round start at the chunk being entered?  Its start state came from rows this chain has
                // just replaced, so it is told to stop and this chain goes on in its place once it has (it notices within one
                // frame).  Should it not stop within the patience below (it is not resident yet: more chains than CUs), this
                // chain ends here instead and the next round continues it -- what every chain did before there was a baton.
                if (tid == 0) {
                    int* st = chp->baton + 2 * (chp->chunk0 + 1 + bi_next);
                    double go = 1.0;
                    if (chp->fuse_flags != nullptr) {   // (first launch) has the chain of the chunk just left decided whether it carries on?
                        int* vd = chp->fuse_flags + 3 * (chp->chunk0 + bi_next) + 2;
                        int spins = 0;
                        while (__hip_atomic_load(vd, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != 1) {
                            __builtin_amdgcn_s_sleep(64);
                            if (++spins > 20000) { go = 0.0; break; }
                        }
                    }
                    if (go != 0.0 && __hip_atomic_load(st, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == 1) {
                        __hip_atomic_store(st + 1, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                        int spins = 0;
                        while (__hip_atomic_load(st, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != 2) {
                            __builtin_amdgcn_s_sleep(64);
                            if (++spins > 20000) { go = 0.0; break; }   // (~2 us per look: some tens of milliseconds)
                        }
                    }
                    if (go != 0.0 && chp->fuse_flags != nullptr) {   // ... and has that chunk's own pass-1 chain written its last row?
                        int* od = chp->fuse_flags + 3 * (chp->chunk0 + 1 + bi_next) + 1;
                        int spins = 0;
                        while (__hip_atomic_load(od, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != 1) {
                            __builtin_amdgcn_s_sleep(64);
                            if (++spins > 20000) { go = 0.0; break; }
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // the rows that chain stored are compared against below
                    cx.scal[S_BATON] = go;
                }
                __syncthreads();
                if (cx.scal[S_BATON] == 0.0) { if (tid == 0 && chp->frames_done) *chp->frames_done = t; break; }
            }
            double* s2 = chp->run_entry + (size_t)(bi_next + 1) * S;    // ... is the state the next chunk is entered with
            for (int i = tid; i < NP; i += MOSHII_TPB) { s2[i] = cx.pose[i]; s2[NP + i] = cx.pose_prev[i]; }
            if (tid < 3) s2[2 * NP + tid] = cx.trans[tid];
            if (tid == 3) s2[2 * NP + 3] = has_prev ? 1.0 : 0.0;
            if (tid == 4) s2[2 * NP + 4] = first ? 1.0 : 0.0;
            if constexpr (XT) for (int e = tid; e < op.nshape; e += MOSHII_TPB) s2[2 * NP + 5 + e] = cx.pose[NP + e];
            ++bi_next;
        }
        if (!COOP && chp->baton != nullptr && sweeping && t < F) {
            // Has an upstream chain of this round asked for this chain's territory (ChainDev::baton)?  Then stop here.  The rows
            // of the chunk this chain is in now switch from its own to older ones at frame t -- in the middle of a chunk, where
            // no hand-off check looks -- so the chunk's entry state is spoiled (an impossible flag value): unless the upstream
            // chain sweeps it (it rewrites the entry state when it gets there), the next verification fails it and it is
            // re-solved from its predecessor's end state, which this chain has just left behind.
            // (a slot of its own: S_BATON above is written by thread 0, possibly before the others have read here)
            if (tid == 0) cx.scal[S_ABORT] = (double)__hip_atomic_load(&chp->baton[2 * chp->chunk0 + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            if (cx.scal[S_ABORT] != 0.0) {
                if (tid == 0) {
                    const int S = 2 * NP + 5 + (XT ? op.nshape : 0);
                    chp->run_entry[(size_t)bi_next * S + 2 * NP + 3] = -1.0;
                    // ... and whoever re-solves this chunk starts by reproducing THIS chain's rows: it must not take that for
                    // having re-joined before it is past frame t
                    int* mark = chp->abort_at + chp->chunk0 + bi_next;
                    if (*mark < chp->bnd_off + t) *mark = chp->bnd_off + t;
                    if (chp->frames_done) *chp->frames_done = -t - 1;
                }
                break;
            }
        }
        if (t == F) break;
        const bool record = t >= skip;
        const uint8_t* visrow = chp->vis + (size_t)t * M;
        // visible-marker list (chmosh.py:591-594), kept in label order
        if (tid < 64) {   // wave 0: one ballot per 64 markers, list position = number of visible markers before this one
            int c = 0, c0 = 0, c1 = 0;
            for (int m0 = 0; m0 < M; m0 += 64) {
                const int m = m0 + tid;
                const bool v = m < M && gptr(visrow)[min(m, M - 1)] != 0;
                const unsigned long long mask = __ballot(v);
                if (v) cx.visidx[c + __popcll(mask & ((1ull << tid) - 1ull))] = m;
                c += __popcll(mask);
                if constexpr (COOP) {   // this rank's share of the list: the visible markers of [mlo, mhi) (the list is in label order)
                    c0 += __popcll(__ballot(v && m < co.mlo));
                    c1 += __popcll(__ballot(v && m < co.mhi));
                }
            }
            if (tid == 0) { cx.scal[S_TMP1] = (double)c; cx.scal[S_V0] = (double)c0; cx.scal[S_V1] = (double)c1; }
        }
        __syncthreads();
        const int nobs = (int)cx.scal[S_TMP1];
        if (nobs == 0) {   // chmosh.py:586-588
            int* st = chp->status;
            if (tid == 0 && st && record && lead) st[t] = 1;
            __syncthreads();
            continue;
        }
        FrameParams fp;
        fp.obs = chp->obs + (size_t)t * M * 3;
        fp.nobs = nobs;
        fp.v0 = COOP ? (int)cx.scal[S_V0] : 0;
        fp.v1 = COOP ? (int)cx.scal[S_V1] : nobs;
        const double n_miss = (double)(M - nobs);
        double anneal = 1.0;
        if (n_miss > 0.0) anneal = anneal + (n_miss / (double)M) * op.wt_annealing;   // :596-601
        fp.wt_data = op.wt_data * (op.num_train_markers / (double)nobs);              // :603
        const double wt_pose = op.wt_poseB * anneal;                                    // :604
        fp.wt_pose = wt_pose;
        fp.wt_poseH = op.wt_poseH * anneal;
        fp.wt_velo = op.wt_velo;
        fp.use_fingers = 0;
        fp.wt_poseF = 0.0; fp.use_face = 0; fp.use_shape = 0; fp.has_stay = 0;
        if constexpr (XT) {
            fp.wt_poseF = op.wt_poseF * anneal;                                          // :606
            // "extrap_dmpl" (:693-697): dmpl_prev was refreshed at :658-659, so the term pulls towards the value the
            // coefficients have on entering this frame -- from the second solved frame on
            fp.has_stay = (!first && op.wt_shape_stay != 0.0) ? 1 : 0;
            for (int e = tid; e < op.nshape; e += MOSHII_TPB) cx.shp0[e] = cx.pose[NP + e];
        }
        fp.has_velo = has_prev ? 1 : 0;
        if (has_prev)   // :624-626  target = pose.r + (pose.r - pose_prev)
            for (int i = tid; i < NP; i += MOSHII_TPB) cx.vtarget[i] = cx.pose[i] + (cx.pose[i] - cx.pose_prev[i]);
        if (!first) {   // :656-657
            for (int i = tid; i < NP; i += MOSHII_TPB) cx.pose_prev[i] = cx.pose[i];
            has_prev = true;
        }
        __syncthreads();
        int n_iter = 0, n_fev = 0, fail = 0;
        PROF_ACC(18, _tf);
        // phases: [rigid + annealed rounds x10 x5 x1 (first solved frame only, :629-655)] step 1 (:665-671),
        //         step 2 (:676-705), record (:712-724)
        Sse fin, carried;
        const bool same_sets = op.same_sets != 0;
        double prev_wt_pose = -1.0;
        int prev_terms = -1;
        for (int kind = first ? 0 : 3; kind < 6; ++kind) {
            const bool round = kind < 3;
            const bool step2 = kind == 4;
            fp.wt_pose = round ? ((kind == 0) ? 10.0 : (kind == 1) ? 5.0 : 1.0) * wt_pose : wt_pose;
            fp.use_fingers = (kind >= 4 && op.nfinger > 0) ? 1 : 0;
            if constexpr (XT) {
                fp.use_face = (kind >= 4 && op.nface > 0) ? 1 : 0;
                fp.use_shape = (kind >= 4 && op.nshape > 0) ? 1 : 0;
            }
            const int terms = fp.use_fingers | (fp.use_face << 1) | (fp.use_shape << 2);   // the Step-2-only residual blocks
            const bool reuse = at_pose && prev_wt_pose == fp.wt_pose && prev_terms == terms;
            fin = run_phase<NBLK, XT, COOP>(cx, ly, md, at, pr, op, fp, visrow, step2 ? op.step2 : op.step1, step2 ? op.n2 : op.n1,
                                  (XT && step2) ? op.nshape : 0, XT ? chp->qscratch + (COOP ? (size_t)co.rank * chp->coop.qstride : 0) : nullptr,
                                  round ? op.e3_first : op.e3, /*rigid=*/kind == 0, /*eval_only=*/kind == 5, reuse, carried, at_pose, fwd_set,
                                  /*set_id=*/(kind >= 4 && !same_sets) ? 2 : 1, vc_key, tab_key, n_iter, n_fev, fail, co);
            prev_wt_pose = fp.wt_pose; prev_terms = terms;
            if constexpr (COOP) { if (cx.scal[S_COOP_FAIL] != 0.0) break; }
        }
        if constexpr (COOP) {   // a rank did not show up: nothing of this frame is recorded; the host sees the group's abort word
            if (cx.scal[S_COOP_FAIL] != 0.0) break;
        }
        first = false;
        PROF_T(_tr);
        if (lead && record && sweeping && chp->rejoin_tol > 0.0 && chp->pose != nullptr && chp->trans != nullptr) {
            // repair chains: has this chain re-joined the trajectory already stored for this chunk?
            double dv = 0.0;
            const double* po = chp->pose + (size_t)t * NP;
            for (int i = tid; i < NP; i += MOSHII_TPB) dv = fmax(dv, fabs(cx.pose[i] - po[i]));
            if (tid < 3) dv = fmax(dv, fabs(cx.trans[tid] - chp->trans[t * 3 + tid]));
            if constexpr (XT)
                if (chp->shape != nullptr)
                    for (int e = tid; e < op.nshape; e += MOSHII_TPB) dv = fmax(dv, fabs(cx.pose[NP + e] - chp->shape[(size_t)t * op.nshape + e]));
            dv = block_max(dv, cx.red);
            rejoin_run = (dv <= chp->rejoin_tol) ? rejoin_run + 1 : 0;   // (NaN compares false)
        }
        if (record) {   // record
            double* o;
            if (lead && (o = chp->pose) != nullptr) for (int i = tid; i < NP; i += MOSHII_TPB) o[(size_t)t * NP + i] = cx.pose[i];
            if (lead && (o = chp->fullpose) != nullptr) for (int i = tid; i < md.P; i += MOSHII_TPB) o[(size_t)t * md.P + i] = cx.fullpose[i];
            if ((o = chp->msim) != nullptr) {   // (cooperative chains: every rank the rows of its own markers)
                const int i_lo = COOP ? 3 * co.mlo : 0, i_hi = COOP ? 3 * co.mhi : 3 * M;
                for (int i = i_lo + tid; i < i_hi; i += MOSHII_TPB) o[(size_t)t * 3 * M + i] = cx.msim[i];
            }
            if constexpr (XT)
                if (lead && (o = chp->shape) != nullptr) for (int e = tid; e < op.nshape; e += MOSHII_TPB) o[(size_t)t * op.nshape + e] = cx.pose[NP + e];
            if (tid == 0 && lead) {
                if ((o = chp->trans) != nullptr) { o[t * 3 + 0] = cx.trans[0]; o[t * 3 + 1] = cx.trans[1]; o[t * 3 + 2] = cx.trans[2]; }
                if ((o = chp->errs) != nullptr) {
                    o[t * 8 + 0] = fin.data; o[t * 8 + 1] = fin.prior; o[t * 8 + 2] = fin.velo; o[t * 8 + 3] = fin.hand;
                    o[t * 8 + 4] = XT ? fin.face : 0.0; o[t * 8 + 5] = XT ? fin.shape : 0.0; o[t * 8 + 6] = XT ? fin.stay : 0.0; o[t * 8 + 7] = 0.0;
                }
                int* oi;
                if ((oi = chp->iters) != nullptr) { oi[t * 2 + 0] = n_iter; oi[t * 2 + 1] = n_fev; }
                if ((oi = chp->status) != nullptr) oi[t] = fail ? -1 : 0;
            }
        }
        __syncthreads();
        if (lead && rejoin_run > 0 && chp->abort_at != nullptr) {
            // rows a stopped chain left in this chunk (ChainDev::abort_at) do not count: matching THEM says nothing about the
            // older rows behind them -- only frames at or past the mark do
            const int mark = __hip_atomic_load(chp->abort_at + chp->chunk0 + bi_next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (chp->bnd_off + t < mark) rejoin_run = 0;
        }
        // (cooperative chains: rank 0's verdict reaches the others at the top of the next frame)
        if (!COOP && rejoin_run >= 2 && t + 1 < F) { if (tid == 0 && chp->frames_done) *chp->frames_done = t + 1; break; }   // pose and pose_prev both match: the stored rows (and final state) stand
        if (t + 1 == F && tid == 0 && lead && chp->frames_done) *chp->frames_done = F;
        PROF_ACC(19, _tr);
    }
    if constexpr (COOP) {
        // a group that broke up inside a chunked solve: the rows of the chunk it stopped in change hands in mid-chunk, where no hand-off
        // check looks -- spoil that chunk's entry state so that the next verification re-solves it (as a stopped plain chain does)
        if (cx.scal[S_COOP_FAIL] != 0.0 && lead && tid == 0 && chp->run_entry != nullptr) {
            const int S = 2 * NP + 5;
            chp->run_entry[(size_t)bi_next * S + 2 * NP + 3] = -1.0;
            if (chp->abort_at != nullptr) chp->abort_at[chp->chunk0 + bi_next] = 0x7fffffff;   // (nothing behind the break counts as re-joined)
        }
        // (a cooperative repair chain: EVERY rank's rows -- the simulated markers are written rank by rank -- are out before rank 0 says so)
        if (chp->baton != nullptr && cx.scal[S_COOP_FAIL] == 0.0) {
            __syncthreads();
            if (tid == 0) __threadfence();
            const unsigned seq = coop_begin(cx);
            const double mine[1] = {0.0};
            coop_exchange_small<1>(co, cx, seq, mine, cx.y);
            coop_end(cx, seq);
        }
    }
    if (lead && chp->baton != nullptr) {   // this chain is out of the way: everything it stored is visible before the flag is
        __syncthreads();
        if (tid == 0) {
            __threadfence();
            __hip_atomic_store(&chp->baton[2 * chp->chunk0], 2, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if constexpr (!COOP && !XT) {
        if (chp->tail_done != nullptr) {   // this chain is done (or has given its chunk up: tail_cut): the others may count on it
            (void)tail_cut;
            __syncthreads();
            if (tid == 0) { __threadfence(); atomicAdd(chp->tail_done, 1); }
        }
    }
    PROF_LAP(12);
#ifdef MOSHII_PROFILE
    if (threadIdx.x == 0 && blockIdx.x == 0) g_prof[30] += wall_clock64() - _wall0;   // constant 100 MHz counter
#endif
}

#ifdef MOSHII_DEV_ONLY_NBLK4   // development builds: one size class (the 63-unknown body solve) compiles in seconds
template __global__ void k_chain_solve<4, 1, false, false>(const ChainDev*, ModelDev, PriorDev, OptsDev, ChainLayout, int);
template __global__ void k_chain_solve<4, 1, false, true>(const ChainDev*, ModelDev, PriorDev, OptsDev, ChainLayout, int);
#else
#define MOSHII_INSTANTIATE(N, XT_, COOP_) \
    template __global__ void k_chain_solve<N, 1, XT_, COOP_>(const ChainDev*, ModelDev, PriorDev, OptsDev, ChainLayout, int);
MOSHII_INSTANTIATE(2, false, false)
MOSHII_INSTANTIATE(4, false, false)
MOSHII_INSTANTIATE(5, false, false)
MOSHII_INSTANTIATE(7, false, false)
MOSHII_INSTANTIATE(8, false, false)
// extended variant: up to 3 + 111 pose + 80 expression unknowns (SMPL-X with fingers and face: NBLK = 13)
MOSHII_INSTANTIATE(5, true, false)
MOSHII_INSTANTIATE(8, true, false)
MOSHII_INSTANTIATE(10, true, false)
MOSHII_INSTANTIATE(13, true, false)
// cooperative chains (G workgroups per chain): the body solve and the solve with fingers
MOSHII_INSTANTIATE(4, false, true)
MOSHII_INSTANTIATE(5, false, true)
MOSHII_INSTANTIATE(7, false, true)
MOSHII_INSTANTIATE(8, false, true)
// ... and the extended variant (BASELINE config 3: 32 sequences x 8 workgroups = the chip)
MOSHII_INSTANTIATE(5, true, true)
MOSHII_INSTANTIATE(8, true, true)
MOSHII_INSTANTIATE(10, true, true)
MOSHII_INSTANTIATE(13, true, true)
#undef MOSHII_INSTANTIATE
#endif

// Simulated markers for explicit pose variables (TransformedLms.r), one frame per workgroup.
__global__ __launch_bounds__(MOSHII_TPB) void k_markers(const AttachDev* __restrict__ attp, ModelDev md, ChainLayout ly,
                                                        const double* __restrict__ pose, const double* __restrict__ trans,
                                                        double* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x;
    const AttachDev at = *attp;
    const Ctx cx = make_ctx(lds, ly);
    const int f = blockIdx.x;
    for (int i = tid; i < md.NP; i += MOSHII_TPB) cx.pose[i] = pose[(size_t)f * md.NP + i];
    if (tid < 3) cx.trans[tid] = trans[(size_t)f * 3 + tid];
    __syncthreads();
    FrameParams fp;
    fp.obs = nullptr; fp.wt_data = 0.0; fp.wt_pose = 0.0; fp.wt_poseH = 0.0; fp.wt_velo = 0.0;
    fp.has_velo = 0; fp.use_fingers = 0; fp.nobs = 0;
    fp.wt_poseF = 0.0; fp.use_face = 0; fp.use_shape = 0; fp.has_stay = 0;
    PriorDev pr; pr.G = 0; pr.npose = 0; pr.means = nullptr; pr.chols = nullptr; pr.halfprec = nullptr; pr.neglogw = nullptr;
    OptsDev op;
    op.nbody = 0; op.nfinger = 0; op.n1 = 0; op.n2 = 0; op.maxiter = 0;
    op.step1 = nullptr; op.step2 = nullptr; op.body = nullptr; op.finger = nullptr;
    op.nface = 0; op.nshape = 0; op.face = nullptr;
    for (int k = 1 + tid; k < md.K; k += MOSHII_TPB) cx.ksum[k - 1] = k;   // every joint's correctives, on top of v_shaped
    for (int i = tid; i < 3 * md.K; i += MOSHII_TPB) cx.Jl[i] = md.J[i];
    __syncthreads();
    fp.v0 = 0; fp.v1 = 0;
    eval_forward<false>(cx, md, at, pr, op, cx.pose, cx.trans, fp, nullptr, cx.ksum, md.K - 1, at.vsh, false);
    for (int i = tid; i < 3 * at.M; i += MOSHII_TPB) out[(size_t)f * 3 * at.M + i] = cx.msim[i];
}

}  // namespace moshii

// coop_g >= 1: cooperative chains, coop_g workgroups per chain (every ChainDev::coop of the launch is filled in for that group size);
// the grid is 8 coop_g ceil(n_chains / 8) blocks, of which those beyond the last chain return at once (block -> (chain, rank): k_chain_solve).
extern "C" hipError_t moshii_launch_chain_solve(int nblk, int two_per_cu, int xt, int n_chains, size_t lds_bytes, hipStream_t stream,
                                                const ChainDev* chains, const ModelDev* md, const PriorDev* pr,
                                                const OptsDev* op, const ChainLayout* ly, int coop_g) {
    using namespace moshii;
    void (*kern)(const ChainDev*, ModelDev, PriorDev, OptsDev, ChainLayout, int) = nullptr;
#ifdef MOSHII_DEV_ONLY_NBLK4
    if (xt || nblk != 4 || two_per_cu) return hipErrorInvalidValue;
    if (coop_g >= 1) kern = k_chain_solve<4, 1, false, true>; else kern = k_chain_solve<4, 1, false, false>;
#else
    if (coop_g >= 1 && xt) {
        switch (nblk) {
            case 5: kern = k_chain_solve<5, 1, true, true>; break;
            case 8: kern = k_chain_solve<8, 1, true, true>; break;
            case 10: kern = k_chain_solve<10, 1, true, true>; break;
            case 13: kern = k_chain_solve<13, 1, true, true>; break;
            default: return hipErrorInvalidValue;
        }
    } else if (coop_g >= 1) {
        switch (nblk) {
            case 4: kern = k_chain_solve<4, 1, false, true>; break;
            case 5: kern = k_chain_solve<5, 1, false, true>; break;
            case 7: kern = k_chain_solve<7, 1, false, true>; break;
            case 8: kern = k_chain_solve<8, 1, false, true>; break;
            default: return hipErrorInvalidValue;
        }
    } else if (xt) {
        switch (nblk) {
            case 5: kern = k_chain_solve<5, 1, true>; break;
            case 8: kern = k_chain_solve<8, 1, true>; break;
            case 10: kern = k_chain_solve<10, 1, true>; break;
            case 13: kern = k_chain_solve<13, 1, true>; break;
            default: return hipErrorInvalidValue;
        }
    } else switch (nblk * 2 + (two_per_cu ? 1 : 0)) {
        // (the 256-register instantiations <N, 2> -- two workgroups per CU, 1.9x slower per chain for +6 % aggregate throughput -- were
        //  removed in round 3; a build without them times 262-264 us per frame against 266-267 with them on the same box.  The "14 %
        //  slower without them" of an earlier commit was a slow box: profiles/r03_bench_line_slow_box.json)
        case 4: kern = k_chain_solve<2, 1, false>; break;
        case 8: kern = k_chain_solve<4, 1, false>; break;
        case 10: kern = k_chain_solve<5, 1, false>; break;
        case 14: kern = k_chain_solve<7, 1, false>; break;
        case 16: kern = k_chain_solve<8, 1, false>; break;
        default: return hipErrorInvalidValue;
    }
#endif
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
    const int grid = (coop_g >= 1) ? 8 * coop_g * ((n_chains + 7) / 8) : n_chains;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(MOSHII_TPB), lds_bytes, stream, chains, *md, *pr, *op, *ly, n_chains);
    return hipGetLastError();
}

extern "C" hipError_t moshii_launch_markers(int F, size_t lds_bytes, hipStream_t stream, const AttachDev* att,
                                            const ModelDev* md, const ChainLayout* ly, const double* pose,
                                            const double* trans, double* out) {
    using namespace moshii;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_markers), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_markers, dim3(F), dim3(MOSHII_TPB), lds_bytes, stream, att, *md, *ly, pose, trans, out);
    return hipGetLastError();
}

#ifdef MOSHII_PROFILE
extern "C" int moshii_prof_trace_read(long long* out /* [8][MOSHII_TRACE_N][4] */) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(moshii::g_trace), sizeof(long long) * 8 * MOSHII_TRACE_N * 4) == hipSuccess ? MOSHII_TRACE_N : -1;
}
extern "C" int moshii_prof_read(long long* out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(moshii::g_prof), sizeof(long long) * 64) != hipSuccess) return -1;
    if (reset) { long long z[64] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(moshii::g_prof), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#endif
