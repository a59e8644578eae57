// Internal device-side structures of libmoshii (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MOSHII_MAXK 64      // joints (ancestor sets are 64-bit masks)

// Pointers that a kernel LOADS from a descriptor (AttachDev / ChainDev fields) are generic to the compiler, and every access
// through them becomes a FLAT instruction: it counts against the LDS counter as well as the vector-memory one, returns out
// of order with respect to both, and so every wait behind it is a full drain (no partial vmcnt waits, no overlap with LDS
// reads).  gptr() types such a pointer as a global-memory address at its use site; kernel arguments do not need it.
#ifndef MOSHII_AS_GLOBAL
#define MOSHII_AS_GLOBAL __attribute__((address_space(1)))
#endif
template <class T> __device__ __forceinline__ const T MOSHII_AS_GLOBAL* gptr(const T* p) { return (const T MOSHII_AS_GLOBAL*)p; }
template <class T> __device__ __forceinline__ T MOSHII_AS_GLOBAL* gptr(T* p) { return (T MOSHII_AS_GLOBAL*)p; }
#if defined(__HIP_DEVICE_COMPILE__)
template <class T> __device__ __forceinline__ const T MOSHII_AS_GLOBAL* gptr(const T MOSHII_AS_GLOBAL* p) { return p; }   // (already typed)
#endif
// The by-value descriptors below (ModelDev, AttachDev, PriorDev, OptsDev) declare their device arrays with MOSHII_GP: a plain
// pointer to the host pass, a global-memory pointer to the device pass (same size and layout), so that device code reaches
// them with global loads wherever the descriptor itself came from (kernel argument, LDS copy, memory).
#if defined(__HIP_DEVICE_COMPILE__)
#define MOSHII_GP(T) T MOSHII_AS_GLOBAL*
#else
#define MOSHII_GP(T) T*
#endif
template <class T> __host__ __device__ inline MOSHII_GP(const T) as_gp(const T* p) { return (MOSHII_GP(const T))p; }   // (filling a descriptor)
template <class T> __host__ __device__ inline MOSHII_GP(T) as_gp_rw(T* p) { return (MOSHII_GP(T))p; }
#define MOSHII_TPB 256      // threads per chain workgroup: 4 waves, one per SIMD

struct ModelDev {
    int V, K, P, NP, body_dof, hand_dof, nhand_full, maxdepth;
    MOSHII_GP(const int) parents;              // [K]
    MOSHII_GP(const double) J;                 // [K][3] regressed joints (current betas)
    MOSHII_GP(const double) hands_mean;        // [nhand_full]
    MOSHII_GP(const double) comps;             // [hand_dof][nhand_full]
    MOSHII_GP(const int) comp_lo;              // [hand_dof] first non-zero column of each component row
    MOSHII_GP(const int) comp_hi;              // [hand_dof] one past the last non-zero column
    MOSHII_GP(const int) col_lo;               // [nhand_full] first component with a non-zero entry in this column
    MOSHII_GP(const int) col_hi;               // [nhand_full] one past the last such component
    MOSHII_GP(const unsigned long long) anc;   // [K] bit j set iff k is j or an ancestor of j
    MOSHII_GP(const int) depth;                // [K] depths, [K] the joints sorted by depth, [maxdepth + 2] where each depth starts in that list
    // free shape block (moshii_model_set_free_shape): dJ/ds, joint-major so that (joint, coefficient) items are contiguous
    int nshape;
    MOSHII_GP(const double) JS;                // [K][nshape][3]
};

struct AttachDev {
    int M, Nv, Nvp, NW;
    MOSHII_GP(const double) vsh;     // [Nv][3] v_shaped rows of the attached vertices (a = 3*m + s)
    MOSHII_GP(const double) Pt;      // [(K-1)*27][Nvp] posedirs slice, vertex index fastest (forward pass: coalesced rows)
    MOSHII_GP(const double) Pj;      // [(K-1)][3][14][M] x 16 bytes: the same slice for the Jacobian pass -- the 27 (+1 pad) entries of (joint, vertex
                           // a = 3 m + s) as 14 pairs, marker index fastest, so that a wavefront whose lanes are consecutive markers
                           // (the T1 items) reads contiguous 16-byte runs (the former 224-byte-record-per-lane form cost one
                           // cache-line look-up per lane per load)
    MOSHII_GP(const int) wj;         // [Nv][NW] joints with non-zero skinning weight (padded: joint 0, weight 0)
    MOSHII_GP(const double) ww;      // [Nv][NW]
    MOSHII_GP(const double) coef;    // [M][3]
    MOSHII_GP(const double) Ssh;     // [nshape][3][Nvp] rows of the free shape block, vertex index fastest (null when nshape == 0)
};

struct PriorDev {
    int G, npose;
    MOSHII_GP(const double) means;     // [G][npose]
    MOSHII_GP(const double) chols;     // [G][npose][npose] lower, L L^T = precision
    MOSHII_GP(const double) halfprec;  // [G][npose][npose] 0.5 * L L^T
    MOSHII_GP(const double) neglogw;   // [G] -log(weight)
    MOSHII_GP(const double) cnorm;     // [G] an upper bound of |L_g|_2 / sqrt(2): |l_g(x) - l_g(y)| <= cnorm_g |x - y| (the kernel's
                                       //     exact shortcut past components that cannot be the minimum)
};

struct OptsDev {
    double wt_data, wt_velo, wt_poseB, wt_poseH, wt_annealing, num_train_markers;
    double e3_first, e3, delta0;
    int maxiter, n1, n2, nbody, nfinger, same_sets;
    MOSHII_GP(const int) step1;
    MOSHII_GP(const int) step2;
    MOSHII_GP(const int) body;
    MOSHII_GP(const int) finger;
    // Step-2 extras (extended kernel variant only)
    double wt_poseF, wt_shape, wt_shape_stay;
    int nface, nshape;
    MOSHII_GP(const int) face;
};

// Cooperative chains (chain_solve.hip, COOP variant): ONE chain solved by G workgroups = G CUs.  Every rank holds the whole solver state
// in its own LDS and runs the whole dogleg control flow (same inputs, same code: same bits); what is SPLIT is the work that scales with
// the markers -- rank r owns the markers [mlo[r], mlo[r + 1]) and their attached vertices: pose correctives, skinning, marker frames,
// residuals, Jacobian rows, its share of J^T J -- and the max-mixture prior, which one rank (prior_rank, given fewer markers) evaluates
// for all.  The ranks meet twice per dogleg iteration: partial sums of squares after a residual evaluation, partial normal equations
// after an assembly (summed in rank order by every rank: identical results everywhere, no broadcast of a step).
#define MOSHII_COOP_MAXG 8
struct CoopDev {
    int G;                                   // ranks of the group (0 or 1: a plain chain)
    int prior_rank;
    int mlo[MOSHII_COOP_MAXG + 1];
    int slot_doubles;                        // doubles per (parity, rank) slot
    int qstride;                             // extended variant: doubles between the ranks' slices of ChainDev::qscratch
    int skew;                                // test aid (MOSHII_COOP_SKEW=seed): every rank is held back a pseudo-random 0 .. 10 us before each exchange
    MOSHII_GP(unsigned long long) slots;     // [2][G][slot_doubles] payload words (doubles as bit patterns: 8-byte agent-scope accesses)
    MOSHII_GP(unsigned int) flags;           // [G] sequence number of the last exchange each rank has posted, [G] = the group's abort word
};
// the rank's view, in KernelCtx
struct CoopCtx {
    int G, rank, prior_rank, mlo, mhi, slot_doubles, skew;
    MOSHII_GP(unsigned long long) slots;
    MOSHII_GP(unsigned int) flags;
};

struct ChainDev {
    const AttachDev* att;
    int F, first;
    const double* obs;
    const uint8_t* vis;
    const double* init_pose;
    const double* init_trans;
    const double* init_prev;
    double* pose;
    double* fullpose;
    double* trans;
    double* msim;
    double* errs;
    int* iters;
    int* status;
    // chunked sequences (moshii_sequence_solve): the first `skip` frames are warm-up -- solved, not recorded.
    // state vectors are [pose NP][pose_prev NP][trans 3][has_prev][first] = 2 NP + 5 doubles (+ [shape E] in the extended variant).
    int skip;
    const double* init_state;   // device; overrides init_pose/init_trans/init_prev/first when non-null
    double* entry_state;        // device or null: state on entering frame `skip`
    double* final_state;        // device or null: state after the last frame
    // repair chains that run on through following chunks: at relative frame bnd[i] the state is the end state of chunk
    // (first + i) and the entry state of chunk (first + i + 1)
    int nb;
    const int* bnd;
    double* run_final;          // final-state slot of this chain's first chunk (slots are contiguous by chunk)
    double* run_entry;          // entry-state slot of this chain's first chunk
    int* frames_done;           // device or null: number of frames this chain processed before it stopped
    // Several repair chains of one round run concurrently, each started at a failing chunk and running on until it re-joins the
    // stored trajectory.  When a chain reaches the chunk at which another chain of the round started, it takes that chain's
    // territory over (the downstream chain's start state came from rows this chain is replacing): baton[2 c] = 1 while the
    // chain started at chunk c runs, 2 once it has exited; baton[2 c + 1] = 1 asks it to stop at its next frame.
    int* baton;                 // device [2 x chunks of the launch] or null
    int* abort_at;              // device [chunks of the launch] or null: first frame (sequence numbering) a stopped chain did NOT re-solve
                                // in that chunk -- the stored rows change hands there, so no chain may declare itself re-joined
                                // before it has passed that frame
    int chunk0;                 // index of this chain's first chunk
    int bnd_off;                // bnd[] holds frame numbers of the sequence; this chain's frame 0 is frame bnd_off
    // free shape coefficients (extended kernel variant)
    const double* init_shape;   // device [nshape] or null
    double* shape;              // device [F][nshape] or null
    double* qscratch;           // device [2][K][nshape][3]: d(joint world position)/ds and q = dt - Rw . dJ/ds of the current point
    double rejoin_tol;          // > 0 (repair chains): stop once two consecutive solved frames reproduce the rows already in
                                // `pose`/`trans` to this tolerance -- the rest of the chunk is then the continuation within tol
    // Pass-1 chains that check their own right-hand hand-off (moshii_sequence_solve, all chunks resident): at frame fuse_F -- the
    // end of its own chunk -- the chain compares its state with the entry state the NEXT chunk's chain recorded (fuse_tol).  Equal:
    // it stops, as a pass-1 chain always did.  Different: it IS the sequential continuation the next chunk needs, so it carries on
    // as the repair chain of that chunk (the fields above -- nb, bnd, run_*, baton, abort_at, chunk0, rejoin_tol -- are set up for a
    // repair chain that starts at chunk fuse_c + 1) instead of leaving that to a later launch that waits for the slowest chunk of
    // pass 1.  fuse_flags[3 c] = 1 once chunk c's chain has recorded its entry state, fuse_flags[3 c + 1] = 1 once it has written
    // the last row of its own chunk (rows of a chunk are never compared against or overwritten before that), fuse_flags[3 c + 2] = 1
    // once it has decided whether it carries on (a sweep arriving at chunk c + 1 waits for that before it looks for a chain there).
    int fuse_F;                 // 0: plain chain
    int fuse_c;                 // index of this chain's own chunk
    int fuse_has_next;          // the next chunk belongs to the same sequence
    int fuse_has_prev;          // so does the previous one
    int* fuse_flags;            // device [3 x chunks of the launch]
    int* fuse_count;            // device: number of chains that carried on
    double fuse_tol;
    CoopDev coop;               // cooperative chains only (G >= 2 ... or 1 for tests); zero otherwise
    // Pass-1 chains of a chunked solve whose repairs are cooperative sweeps (round 6): the launch ends with its SLOWEST chunk -- 250 chunks of
    // the bench sequence: 154 .. 222 dogleg iterations, the launch takes 21.8 ms where the median chunk needs 13.4 (tools/first_frame.py) --
    // with nine CUs in ten idle at the end.  Every chain counts itself in tail_done when it ends; a chain that finds tail_quota others done
    // while it has more than tail_left frames to go gives up: it leaves spoiled (finite, never-matching) entry / final states and a mark that
    // no sweep may re-join inside its chunk, so that the host's rounds re-solve the chunk from its predecessor's end state -- a cooperative
    // sweep of 16 frames costs 2.5 ms beside the sweeps that run anyway.
    int* tail_done;             // device counter, or null
    int tail_quota, tail_left;
    int tail_can_cut;           // 0: this chain only counts (a sequence's first chunk has no predecessor to be re-solved from)
    int* tail_mark;             // device: abort_at slot of this chain's chunk
};

// LDS layout of the chain kernel: offsets in doubles from the dynamic-LDS base (all multiples of 2).
struct ChainLayout {
    int Mmax, Nvmax, NWmax, nmax, Tm, nkfmax, LDJ, nhj;
    int NPX;                   // pose variables + free shape coefficients (the shape block rides behind the pose in LDS)
    int o_vshp, o_shp0;        // shaped rest vertices of the current evaluation; shape coefficients at frame start
    int o_pose, o_trans, o_pose_t, o_trans_t, o_pose_prev, o_vtarget, o_fullpose;
    int o_feat, o_B, o_omega, o_Rw, o_tw, o_Rloc, o_acol, o_Jl;
    int o_vposed, o_vpos, o_msim, o_res, o_vconst;
    int o_xb, o_ell, o_score;
    int o_px0, o_ps0;          // prior reference point [npose] and sqrt of every component's score there [G]
    int o_g, o_dsd, o_dgn, o_ddl, o_y;
    int o_red, o_scal;
    int o_anc;                 // K x u64
    int o_ints;                // int region (offset in doubles)
    int o_big;                 // union: packed Cholesky factor | Jacobian tiles
    int big_doubles;
    int total_doubles;
    // int region sub-offsets (in ints)
    int i_visidx, i_colpid, i_colprior, i_pid2prior, i_jointslot, i_kfree, i_colq, i_ksum, i_kconst, i_total;
    // tile sub-offsets inside big (in doubles)
    int t_Jh, t_Jrow, t_Lm, t_Trot, t_xjs, t_rest, t_tjs;
};

// Copy of the chain kernel's descriptors at the head of its dynamic LDS (ChainLayout offsets start behind it): the phases
// of the solver that are compiled as functions of their own (chain_solve.hip) pick their context up from here instead of
// receiving several hundred bytes of arguments.
struct KernelCtx {
    ChainLayout ly;
    ModelDev md;
    PriorDev pr;
    OptsDev op;
    AttachDev at;
    double fp_raw[12];   // the running phase's per-frame parameters (chain_solve.hip: FrameParams), rewritten at every phase start
    CoopCtx co;          // cooperative variant only
    unsigned long long kargs;   // the kernel's argument segment (device builds: the functions read ly / md / pr / op from there with scalar loads)
};
#define MOSHII_KC_DOUBLES ((int)((sizeof(KernelCtx) + 15) / 16 * 2))

// single-precision / half-precision copies of the model for the full-mesh export kernels (lbs_forward.hip)
struct Lbs32Model {
    // plain f32 copy (fallback kernel)
    float* v_shaped;       // [V][3]
    float* posedirs_t;     // [9(K-1)][3][Vp64]  vertex fastest
    float* weights;        // [K][Vp64]
    float* J;              // [K][3]
    float* hcompf;         // [hand_dof][nhand_full] f32 copy of the hand-pose components (k_lbs_prep), made at the first export
    float* hmeanf;         // [nhand_full]
    float* jtab;           // [K][16] per-joint record of k_lbs_prep (k_pack_jtab), rebuilt when the joints change (jtab_valid)
    float* hcj;            // [K][16][4] the joint's window of hand-component rows
    int jtab_valid;
    int* varflag;          // [64] varflag[j] == epoch: joint j moves in the current export call (k_lbs_prep writes, k_lbs_export reads)
    int epoch;             // the export call's number
    int Vp;                // V padded to 64
    // MFMA path (lbs_forward.hip: k_lbs_prep + k_lbs_export).  Tiles of 64 vertices, each cut into four GROUPS of 16 (group_tiles)
    _Float16* Pfrag;       // [NVT*4 groups][3][KS][64 lanes][8]  posedirs * pscale, A-operand fragments of v_mfma_f32_16x16x32_f16
    int* perm;             // [NVT*64]     group order -> vertex id (-1: padding)
    char* tables;          // the per-group tables, one allocation; byte offsets of its parts:
    unsigned tab_gnr;      //   int   [NVT*4]              blend rounds of the group (four joints each)
    unsigned tab_gjid;     //   int   [NVT*4][NRM][4]      the group's joint list as byte offsets (j x 768) into a 16-frame block of Atr
    unsigned tab_gx;       //   int   [NVT*64]             byte offset of the vertex's column in a tile row of the result exchange (12 x local id)
    unsigned tab_vshs;     //   float [NVT*64][4]          rest positions in group order, x pscale
    unsigned tab_vsc;      //   float [NVT*64][4]          per call: rest positions + the correctives of the joints that do not move, x pscale (k_lbs_still)
    unsigned tab_gw;       //   float [NVT*4][NRM][16][4]  weights of the group's 16 vertices on the round's four joints
    long long* dbgbuf;     // [512][2] development: start / end time of every workgroup of the last export (MOSHII_LBS_STOP=32)
    int K, NRM;            // NRM: the largest round count over all groups
    float pscale, inv_pscale;
    int NVT, KS;           // vertex tiles; k-steps of 32 pose features
    int KJ;                // joints padded to a multiple of 4
    int mfma_ok;
    // per-call scratch, grown to the largest frame count seen (padded to whole 128-frame tiles)
    float* Atr;            // [Fcap/16][KJ][16][12]  joint transforms, row major (R_i0, R_i1, R_i2, t_i), i = 0 .. 2: each float one B operand of the blend
    _Float16* featF;       // [Fcap/128][KS][8][64 lanes][8]  pose features, B-operand fragments
    int Fcap;
};
