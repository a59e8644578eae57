/*
 * moshii.h -- C ABI of libmoshii.so: the MI355X-native MoSh++ Stage-II hot path.
 *
 * The reference (nghorbani/moshpp, pure Python) has no FFI; the boundary it offers is the
 * function-injection point `MoSh.mosh_stageii(mosh_stageii_func)` (src/moshpp/mosh_head.py:268-301,
 * call at :280-286) whose callee `chmosh.mosh_stageii` (src/moshpp/chmosh.py:458-741) is what this
 * library replaces.  Each entry point below cites the reference code whose arithmetic it takes over.
 * The Python mirror of the reference interface (moshpp_amd/chmosh.py) binds these with ctypes;
 * INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *   - plain C, caller-owned buffers, no global state except the per-process HIP context;
 *   - every function returns 0 on success, <0 on error; moshii_last_error() gives the message;
 *   - all floating-point data is IEEE double unless a name ends in _f32;
 *   - "host pointer" arguments are read during the call; handles own device copies;
 *   - thread-safe per handle (one host thread per handle at a time).
 *
 * Limits (one chain's whole solver state lives in the 160 KiB LDS of one CU; exceeding one gives MOSHII_ERR_UNSUPPORTED
 * with a message, nothing is truncated silently; the reference has no such limits)
 *   - joints K <= 64 (ancestor sets are 64-bit masks);
 *   - latent markers M <= 128 per attachment (SMPL-X layouts with body + finger + face markers: 89 in BASELINE config 3);
 *   - unknowns per solve (3 + free pose variables + free shape coefficients): <= 127 without, <= 207 with a jaw term or a
 *     free shape block (n_face > 0 or n_shape > 0);
 *   - mixture components of the prior x npose must fit beside the Jacobian tiles: G = 8, npose <= 69 as the reference
 *     ships them; the marker-tile size adapts, a layout that cannot fit at all is reported as an error.
 */
#ifndef MOSHII_H
#define MOSHII_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct moshii_model_s*  moshii_model_t;   /* body model resident in HBM                    */
typedef struct moshii_prior_s*  moshii_prior_t;   /* max-mixture pose prior resident in HBM        */
typedef struct moshii_attach_s* moshii_attach_t;  /* marker attachment: compact marker-vertex model */

#define MOSHII_OK               0
#define MOSHII_ERR_ARG         -1
#define MOSHII_ERR_HIP         -2
#define MOSHII_ERR_UNSUPPORTED -3
#define MOSHII_ERR_NO_DEVICE   -4
#define MOSHII_ERR_NUMERIC     -5

/* flags for moshii_chain_solve / moshii_sequence_solve / moshii_lbs_forward_* : where the big per-frame buffers live */
#define MOSHII_BUFFERS_HOST    0u   /* obs/vis/outputs are host pointers (library stages them)     */
#define MOSHII_BUFFERS_DEVICE  1u   /* obs/vis/outputs are device pointers; the launch is async on `stream` -- EXCEPT where cooperative
                                     * chains are used (below): those calls synchronise the stream before they return */

/* moshii_chain_solve and moshii_sequence_solve: COOPERATIVE chains -- a chain is solved by g workgroups (g CUs; 2 <= g <= 8) instead of
 * one.  The ranks split the work that scales with the markers (pose correctives, skinning, marker frames, Jacobian rows, J^T J) and one
 * of them evaluates the prior; they meet twice per dogleg iteration through device memory (csrc/moshii_dev.h: CoopDev).  Same
 * algorithm, same decisions; sums over markers are taken rank by rank, so results agree with a plain chain's to round-off (same dogleg
 * iteration counts), not bit for bit.  All variants are built: body / finger solves and the extended ones (face, free shape block).
 *   no MOSHII_COOP_GROUP word in `flags` (the value 0)  ->  the environment: MOSHII_COOP=g (moshii_chain_solve) / MOSHII_COOP_REPAIR=g
 *       (the repair chains of moshii_sequence_solve), "auto" or unset = THE LIBRARY'S CHOICE: cooperative with one rank per 256
 *       (marker, joint) Jacobian items plus one for the prior when every workgroup of the launch can be resident at once
 *       (g x n_chains <= CUs) and the solve is large enough to gain (SMPL-H / 53 markers: g = 6; MANO: plain); plain chains otherwise.
 *       A drop-in caller therefore gets cooperative chains for a single sequential chain -- 1.5x the one-workgroup rate.
 *   MOSHII_COOP_GROUP(1)  ->  plain chains (bit-reproducible; the call stays asynchronous with MOSHII_BUFFERS_DEVICE)
 *   MOSHII_COOP_GROUP(g), 2 <= g <= 8  ->  g workgroups per chain; falls back to plain chains when g x n_chains > CUs.
 * Cooperative calls synchronise the stream: a group whose ranks do not all become resident (device shared with another process) gives
 * up after ~0.1 s of waiting, the call is then repeated with plain chains (results as from MOSHII_COOP_GROUP(1)), a line goes to
 * stderr, and the library's own choice stays "plain" for the rest of the process.
 * (Test aid: MOSHII_COOP_SKEW=seed in the environment holds every rank back a pseudo-random 0 .. 10 us before each of its exchanges;
 *  results must not move by a bit -- tests/test_gpu_parity.py::test_cooperative_exchanges_under_randomised_rank_skew.) */
#define MOSHII_COOP_GROUP(g)   (((uint32_t)(g) & 0xffu) << 8)

const char* moshii_last_error(void);
/* 100: round 1-2 ABI.  101: moshii_stagei_desc grew by the trailing output pointer `init_sq` -- the struct carries no size field, so a
 * caller built against the 100 header must not call a 101 library's moshii_stagei_solve (it would read past the caller's struct);
 * check moshii_version() >= 101 before filling a moshii_stagei_desc declared from this header, and zero the struct first. */
int  moshii_version(void);
/* first 16 hex digits of the SHA-256 over the sources this binary was compiled from (python -m moshpp_amd.build computes the same
 * over the tree: a stale binary is detectable); "unknown" for a build outside build.py */
const char* moshii_source_hash(void);
int  moshii_device_count(void);
int  moshii_set_device(int device);
int  moshii_device_multiprocessors(void);   /* CUs of the current device (the chunk count moshii_sequence_solve picks by default) */

/* ---------------------------------------------------------------------------------------------
 * Body model.  Replaces load_surface_model + SmplModelLBS construction
 * (src/moshpp/models/smpl_fast_derivatives.py:52-244): the arrays of the model pickle plus the
 * pose-variable layout  fullpose = [pose[:body_dof], hands_mean + pose[body_dof:].selected_components]
 * (:194-204).  SMPL: body_dof=72, hand_dof=0.  SMPL-H: 66/2*dof_per_hand.  SMPL-X: 75/2*dof_per_hand.
 * MANO: 3/dof_per_hand.
 * ------------------------------------------------------------------------------------------- */
typedef struct moshii_model_desc {
    int32_t V;                      /* vertices                                                   */
    int32_t K;                      /* joints (<= 64)                                             */
    int32_t NB;                     /* shape coefficients held by shapedirs                       */
    int32_t body_dof;               /* leading fullpose dofs that are pose variables themselves   */
    int32_t hand_dof;               /* number of hand-PCA coefficients (0: none)                  */
    const int32_t* parents;         /* [K], parents[0] = -1 (kintree_table[0])                    */
    const double*  v_template;      /* [V][3]                                                     */
    const double*  shapedirs;       /* [V][3][NB]                                                 */
    const double*  posedirs;        /* [V][3][9(K-1)]                                             */
    const double*  weights;         /* [V][K] dense skinning weights                              */
    const double*  J_regressor;     /* [K][V] dense                                               */
    const double*  hands_mean;      /* [3K-body_dof] or NULL                                      */
    const double*  selected_components; /* [hand_dof][3K-body_dof] or NULL                        */
} moshii_model_desc;

int moshii_model_create(const moshii_model_desc* desc, moshii_model_t* out);
int moshii_model_destroy(moshii_model_t m);

/* v_shaped = v_template + shapedirs[:,:,:nb].betas ; J = J_regressor.v_shaped
 * (smpl_fast_derivatives.py:186-191; chmosh.py:499-500 writes the Stage-I betas).  Betas are frozen in
 * Stage-II, so this runs once per subject. */
int moshii_model_set_betas(moshii_model_t m, const double* betas, int32_t nb);

/* Declares shapedirs columns [start, start+count) as per-frame FREE variables of Stage-II Step 2: the expression
 * coefficients opt_model.betas[exp_start : exp_start+num_expressions] (chmosh.py:565-567, 687-688) or the DMPL
 * coefficients opt_model.betas[num_betas : num_betas+num_dmpls] (:513-514, 698-699).  Precomputes
 * JS = J_regressor . shapedirs[:,:,block] (the joints move with them, smpl_fast_derivatives.py:187-191); the solver
 * treats the coefficients as an OFFSET on the betas frozen by moshii_model_set_betas.  count = 0 clears it.
 * Call before moshii_attach_create (attachments gather their rows of the block). */
int moshii_model_set_free_shape(moshii_model_t m, int32_t start, int32_t count);

/* Regressed joints J[K][3] for the current betas (host buffer). */
int moshii_model_get_joints(moshii_model_t m, double* J_out);

/* Full-mesh LBS forward, SmplModelLBS.r (smpl_fast_derivatives.py:206-218,243-244 -> psbody
 * verts_decorated): pose[F][NP] (pose *variables*), trans[F][3] -> verts[F][V][3].
 * _f64: reference-precision path (used for the canonical body of TransformedCoeffs, chmosh.py:502).
 * _f32: batched export kernels: the pose-corrective contraction (3V x 9(K-1) x F) on the f16 matrix pipe (f32 accumulate), the skinning
 *       blend over per-group joint lists on the f32 matrix instruction.  |error| <= 2e-5 m against _f64.
 *       Joints whose pose variables are bitwise the same in every frame of a call (a body-only Stage-II result keeps the hand pose of
 *       SMPL-H / SMPL-X fixed) are noticed per call and their correctives evaluated once instead of per frame -- same result to
 *       round-off, nothing assumed about the input.  Any F: exports beyond the kernels' 2 GiB addressing range of per-call scratch
 *       (SMPL-H: 860 000 frames) are cut into sub-calls.  One export per model handle at a time (the per-call scratch belongs to the model).
 *       Buffers host or device per flags. */
int moshii_lbs_forward_f64(moshii_model_t m, int32_t F, const double* pose, const double* trans,
                           double* verts, uint32_t flags, void* stream);
int moshii_lbs_forward_f32(moshii_model_t m, int32_t F, const float* pose, const float* trans,
                           float* verts, uint32_t flags, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Pose prior.  Replaces create_gmm_body_prior / MaxMixtureComplete
 * (src/moshpp/prior/gmm_prior_ch.py:42-134).  The caller passes the already prepared arrays:
 * means[G][npose], chols[G][npose][npose] (lower Cholesky factors of the precisions, :122-123) and the
 * re-normalised weights[G] (:126-130).
 * ------------------------------------------------------------------------------------------- */
int moshii_prior_create(int32_t G, int32_t npose, const double* means, const double* chols,
                        const double* weights, moshii_prior_t* out);
int moshii_prior_destroy(moshii_prior_t p);

/* ---------------------------------------------------------------------------------------------
 * Marker attachment.  Replaces the numeric result of TransformedCoeffs (src/moshpp/transformed_lm.py:45-113)
 * as consumed by TransformedLms (:120-162): closest[M][3] vertex ids and coef[M][3].  Gathers the
 * rows of v_shaped / posedirs / weights of the <= 3M attached vertices into a compact slice
 * (requires moshii_model_set_betas first; re-create after changing betas).
 * ------------------------------------------------------------------------------------------- */
int moshii_attach_create(moshii_model_t m, int32_t M, const int32_t* closest, const double* coef,
                         moshii_attach_t* out);
int moshii_attach_destroy(moshii_attach_t a);

/* Simulated markers for given pose variables (TransformedLms.r): pose[F][NP], trans[F][3] -> markers[F][M][3].
 * Host buffers. */
int moshii_attach_markers(moshii_attach_t a, int32_t F, const double* pose, const double* trans,
                          double* markers);

/* ---------------------------------------------------------------------------------------------
 * Stage-II chain solve.  Replaces the frame loop of chmosh.mosh_stageii (src/moshpp/chmosh.py:584-724)
 * including every ch.minimize(method='dogleg') call (:651-653, 669-671, 703-705).
 * One chain = one sequence (or a contiguous chunk of one) walked frame by frame with warm start and
 * the 2-frame velocity term (:624-626, 656-657).  Chains of one call run concurrently, one workgroup
 * each; they share the model, prior and options but may have different attachments.
 * ------------------------------------------------------------------------------------------- */
typedef struct moshii_solve_opts {
    /* opt_settings.weights (support_data/conf/moshpp_conf.yaml:118-125) */
    double wt_data, wt_velo, wt_poseB, wt_poseH, wt_annealing;
    double num_train_markers;       /* 46, chmosh.py:460                                         */
    double e3_first, e3;            /* 1e-3 (first-frame rounds, :653) and 1e-2 (:671,705)       */
    double delta0;                  /* 0.5                                                       */
    int32_t maxiter;                /* opt_settings.maxiter (100)                                */
    int32_t n_step1;                /* free pose-variable ids of Step 1 (:665-668), sorted       */
    const int32_t* step1_ids;
    int32_t n_step2;                /* free pose-variable ids of Step 2 (:676-692), sorted       */
    const int32_t* step2_ids;
    int32_t n_body;                 /* pose_body_ids: the prior's argument (:613), == prior npose or 0 */
    const int32_t* body_ids;
    int32_t n_finger;               /* pose_finger_ids when optimize_fingers (:681-683), else 0  */
    const int32_t* finger_ids;
    /* Step-2 extras, all optional (0 / NULL = absent) */
    int32_t n_face;                 /* pose_face_ids when optimize_face (:560-563, 685-686): contiguous, also in step2_ids */
    const int32_t* face_ids;
    double  wt_poseF;               /* stageii_wt_poseF (annealed like poseB/H, :606)             */
    int32_t n_shape;                /* > 0: the block of moshii_model_set_free_shape is free in Step 2 (count must match) */
    double  wt_shape;               /* stageii_wt_expr (:687) or stageii_wt_dmpl (:698)           */
    double  wt_shape_stay;          /* DMPL: 6.0 = "extrap_dmpl" (:693-697).  dmpl_prev is refreshed (:658-659) before the
                                     * term is built, so it evaluates to (dmpl - dmpl at frame start) * 6 from the second
                                     * solved frame on.  0: no such term (expression).           */
} moshii_solve_opts;

#define MOSHII_NERR 8               /* per-frame SSE columns: data, poseB, velo, poseH, poseF, shape, shape_stay, 0 */

typedef struct moshii_chain_desc {
    moshii_attach_t attach;
    int32_t F;                      /* frames in this chain                                      */
    int32_t first_frame_schedule;   /* 1: rigid init + annealed rounds on the first solved frame (:629-655) */
    const double*  obs;             /* [F][M][3] metres, ordered like the latent labels (:591)   */
    const uint8_t* vis;             /* [F][M] 1 = label observed in this frame (markers_asdict)   */
    const double*  init_pose;       /* host [NP] or NULL (zeros)                                  */
    const double*  init_trans;      /* host [3]  or NULL (zeros)                                  */
    const double*  init_pose_prev;  /* host [NP] or NULL (no velocity term on the next frame)     */
    const double*  init_shape;      /* host [n_shape] or NULL (zeros): free shape coefficients     */
    /* outputs, one row per input frame; rows of frames without visible markers are left untouched
     * and flagged status = 1 (the reference skips them, :586-588) */
    double*  pose;                  /* [F][NP] pose variables                     (may be NULL)   */
    double*  fullpose;              /* [F][3K]  opt_model.fullpose (:720)                          */
    double*  trans;                 /* [F][3]                                                      */
    double*  markers_sim;           /* [F][M][3] simulated markers (all M; caller selects visible) */
    double*  errs;                  /* [F][MOSHII_NERR] SSE of every residual block (:712-714)     */
    int32_t* iters;                 /* [F][2] dogleg outer iterations, residual evaluations        */
    int32_t* status;                /* [F] 0 solved, 1 skipped (no markers), <0 numerical failure  */
    double*  shape;                 /* [F][n_shape] free shape coefficients: `expression` (:723-724) / `dmpls` (:721-722); may be NULL */
} moshii_chain_desc;

int moshii_chain_solve(moshii_model_t m, moshii_prior_t prior /* may be NULL when n_body == 0 */,
                       const moshii_solve_opts* opts, int32_t n_chains, const moshii_chain_desc* chains,
                       uint32_t flags, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Chunked sequence solve: the same frame loop (src/moshpp/chmosh.py:584-724), with each sequence cut into
 * contiguous chunks that are solved concurrently (one workgroup each) and then stitched so that the result
 * equals the sequential chain's to within `verify_tol`:
 *   pass 1  chunk c > 0 starts `warmup` frames early with the first-frame schedule (:629-655); those frames
 *           are solved but not recorded.  The chain map is a contraction (data term 400 vs velocity 2.5,
 *           moshpp_conf.yaml:118-125): the influence of the start state decays ~2.5x per frame
 *           (tools/chunk_deviation.py: 16 frames -> <1e-7 rad, 32 -> <1e-12).
 *   verify  the state (pose, pose_prev, trans) with which chunk c enters its first recorded frame is compared
 *           with the state its predecessor ended in; max|diff| <= verify_tol accepts the hand-off.
 *   repair  a chunk that fails is re-solved from its predecessor's exact end state (warm start + velocity
 *           term exactly as :624-626, 656-657), and its successor is re-verified; repeated until clean.
 * The call synchronises `stream` (the verification result is read on the host).
 * Free shape coefficients (n_shape > 0) travel with the hand-off states (verified and repaired like pose / trans).
 * ------------------------------------------------------------------------------------------- */
typedef struct moshii_sequence_desc {
    moshii_attach_t attach;
    int32_t F;
    const double*  obs;             /* [F][M][3]                                                   */
    const uint8_t* vis;             /* [F][M]                                                      */
    /* optional start state (host pointers; all NULL = the first-frame schedule of :629-655 on the first solved frame).
     * With init_pose / init_trans the sequence CONTINUES a chain: warm start from that state, and -- when init_pose_prev
     * is given too -- the velocity term from the first frame on (:624-626).  Used to shard one long sequence over
     * several GPUs: a rank re-solves its frame range from its left neighbour's end state (moshpp_amd/parallel.py). */
    const double*  init_pose;       /* [NP] or NULL                                                */
    const double*  init_trans;      /* [3]  (required with init_pose)                              */
    const double*  init_pose_prev;  /* [NP] or NULL                                                */
    double*  pose;                  /* outputs as in moshii_chain_desc, one row per input frame    */
    double*  fullpose;
    double*  trans;
    double*  markers_sim;
    double*  errs;
    int32_t* iters;
    int32_t* status;
    /* extended variant (n_shape > 0): the free coefficients travel with the chunk hand-off states */
    const double* init_shape;       /* [n_shape] start values when init_pose is given, or NULL (zeros) */
    double*       shape;            /* [F][n_shape] output, or NULL                                 */
} moshii_sequence_desc;

typedef struct moshii_chunk_opts {
    int32_t num_chunks;             /* chunks per sequence; 0 = fill the GPU (1 workgroup per CU)   */
    int32_t warmup;                 /* warm-up frames per chunk (NULL opts: 32)                     */
    double  verify_tol;             /* hand-off tolerance on pose [rad] / trans [m] (<= 0: 1e-11)   */
} moshii_chunk_opts;

typedef struct moshii_chunk_report {
    int32_t n_chunks, n_repaired, repair_rounds, warmup;
    double  max_handoff_dev;        /* largest accepted hand-off deviation                          */
    double  verify_tol;
} moshii_chunk_report;

/* Balanced chunk plan (pure host arithmetic, no device needed): writes starts[c] (first recorded frame) and
 * launch_starts[c] = max(0, starts[c] - warmup) (launch_starts[0] = 0); returns the number of chunks used
 * (<= min(num_chunks, cap, max(F,1))) or <0. */
int moshii_plan_chunks(int32_t F, int32_t num_chunks, int32_t warmup, int32_t cap, int32_t* starts, int32_t* launch_starts);

int moshii_sequence_solve(moshii_model_t m, moshii_prior_t prior, const moshii_solve_opts* opts, int32_t n_seq,
                          const moshii_sequence_desc* seqs, const moshii_chunk_opts* chunk_opts /* NULL: defaults */,
                          uint32_t flags, void* stream, moshii_chunk_report* report /* may be NULL */);

/* ---------------------------------------------------------------------------------------------
 * Stage-I: subject shape + latent marker placement from a handful of picked frames.  Replaces the numeric core of
 * mosh_stagei (src/moshpp/chmosh.py:177-447): prepare_mosh_markers_latent (:57-80), the per-frame rigid start
 * (:236-238), the annealing rounds of one dogleg each over [trans_f, markers_latent, pose_f[ids], betas[:nb]]
 * (:313-415) with the terms data / poseB / init_* / beta / surf (+ poseH in the last two rounds), and the outputs of
 * :417-447.  The marker attachment (TransformedCoeffs, transformed_lm.py:59-113) is re-evaluated at every evaluation
 * point, as the reference's dependency graph does.  All buffers are HOST pointers (the problem is a few kilobytes);
 * the model's own betas (moshii_model_set_betas) are ignored: the solve works on v_template + shapedirs[:, :, :nb].betas.
 * Not covered: opt_settings.extra_initial_rigid_adjustment.
 * ------------------------------------------------------------------------------------------- */
typedef struct moshii_stagei_desc {
    int32_t n_frames, M, n_faces, nb;       /* picked frames, latent markers, triangles, free betas             */
    const int32_t* faces;                   /* [n_faces][3] surface triangles (can_model.f)                      */
    const int32_t* marker_vids;             /* [M] marker_meta['marker_vids'] values                             */
    const double*  m2b;                     /* [M] distance from skin per marker (m2b_distance by type, :62-64)  */
    const double*  wt_init;                 /* [M] stagei_wt_init[_type] per marker, before annealing (:327-328) */
    const int32_t* n_obs;                   /* [n_frames] observed latent markers per frame                      */
    const int32_t* obs_ids;                 /* [sum n_obs] their latent ids (common_labels, :199-206)            */
    const double*  obs;                     /* [sum n_obs][3] metres                                             */
    const int32_t* exclude_vids;            /* vertices the attachment may not use (SMPL-X eyeballs) or NULL     */
    int32_t        n_exclude;
    const double*  betas_init;              /* [nb] or NULL (zeros)                                              */
    double wt_data, wt_poseB, wt_poseH, wt_betas, wt_surf;   /* stagei_wt_* (moshpp_conf.yaml:103-116)           */
    const double*  annealing;               /* [n_anneal] stagei_wt_annealing                                    */
    int32_t        n_anneal;
    const int32_t* pose_ids;                /* root + body (- toes) pose variables free in every round (:384-388) */
    int32_t        n_pose_ids;
    const int32_t* body_ids;                /* pose variables the GMM prior sees (pose_body_ids); n_body == prior npose */
    int32_t        n_body;
    const int32_t* finger_ids;              /* added (with poseH) in the last two rounds (:390-393); may be empty */
    int32_t        n_finger;
    /* optimize_face (chmosh.py:295-305, 394-398): per-frame expression coefficients betas[expr_start : expr_start + n_expr] and the
     * jaw pose ids become free in the last two rounds, with the terms expr (wt_expr) and poseF (wt_poseF).  Needs nb == 0: the
     * reference cannot share betas and free expressions either (:295-299) -- fold the fixed betas into the model's template.  */
    int32_t        n_expr, expr_start;
    const int32_t* face_ids;                /* pose_face_ids (66:69 for SMPL-X) or NULL                          */
    int32_t        n_face;
    double         wt_expr, wt_poseF;       /* stagei_wt_expr, stagei_wt_poseF                                   */
    const int32_t* head_ids;                /* [n_head] latent ids of the head markers, or NULL: no correlation term */
    const double*  head_corr;               /* [n_head_rows][n_head] `corr` of head_marker_corr_fname (:252-266, 362-369) */
    int32_t        n_head, n_head_rows;
    double         wt_init_head;            /* stagei_wt_init_body if the layout has a 'body' type, else stagei_wt_init */
    int32_t        maxiter;                 /* cfg.opt_settings.maxiter                                          */
    double         stagei_lr;               /* cfg.opt_settings.stagei_lr (dogleg e_3)                           */
    /* optional: frames of ONE problem spread over ranks (one process per GPU).  Every rank passes the same problem; rank r evaluates
     * the data / prior / finger rows of frames [frame_lo, frame_hi), exactly one rank (owns_shared_rows) the init / beta / surf / head
     * rows.  allreduce_sum sums `count` doubles in place over the ranks (host buffer): the normal equations [A | g] once per dogleg
     * iteration, a scalar per residual evaluation, the rigid starts and the final SSEs.  Every rank then takes the same step and
     * returns the same result.  This is the reference deployment's "shared betas" coupling (all frames share betas and the latent
     * markers) as a reduction; sharded == 0 ignores all of it. */
    int32_t  sharded, frame_lo, frame_hi, owns_shared_rows;
    int    (*allreduce_sum)(double* buf, int64_t count, void* user);
    void*    allreduce_user;
    /* outputs */
    double*  betas;                         /* [nb]                                                              */
    double*  markers_latent;                /* [M][3]                                                            */
    int32_t* markers_latent_vids;           /* [M] nearest canonical vertex of each latent marker (:420-422)     */
    double*  pose;                          /* [n_frames][NP]                                                    */
    double*  trans;                         /* [n_frames][3]                                                     */
    double*  markers_sim;                   /* [n_frames][M][3] or NULL: every latent marker simulated on every frame's body
                                             * (sharded: the frames are gathered with an all-reduce -- set it on EVERY rank or on none;
                                             * errs / init_sq may differ between ranks, their reductions are always issued)            */
    double*  expression;                    /* [n_frames][n_expr] or NULL                                        */
    double*  errs;                          /* [8] SSE of data, poseB, init, beta (expr when n_expr > 0), surf, poseH, init_head_corr, poseF */
    int32_t* iters;                         /* [1] dogleg outer iterations over all rounds                       */
    /* options added later (behind the outputs: a caller that zeroes the struct and knows nothing of them gets the old behaviour) */
    int32_t  extra_initial_rigid_adjustment;   /* cfg.opt_settings.extra_initial_rigid_adjustment (chmosh.py:230-232): before the annealing
                                                * rounds, one dogleg over every frame's root orientation + translation on the unweighted
                                                * marker residuals (e_3 = .001, delta_0 = .5, maxiter)                                   */
    int32_t  allreduce_on_device;              /* sharded only: allreduce_sum is handed DEVICE pointers (the solver's own buffers: the
                                                * Schur block / normal equations never visit the host; the few n-vectors and scalars are
                                                * staged through a device scratch).  The stream has been synchronised when the callback
                                                * runs and the callback returns after its collective has completed -- e.g. RCCL
                                                * (torch.distributed backend "nccl") on a tensor wrapped around the pointer.            */
    double*  init_sq;                          /* [M] or NULL (output): every latent marker's share of errs[2], the squared weighted
                                                * distance to its initial placement at the solution -- the reference keeps one `init_<type>`
                                                * entry per marker type in stagei_errs (chmosh.py:362-371); the caller sums by type       */
} moshii_stagei_desc;

int moshii_stagei_solve(moshii_model_t m, moshii_prior_t prior /* may be NULL */, const moshii_stagei_desc* desc, void* stream);

/* Introspection for benchmarks: name and dynamic-LDS bytes of the kernel the last moshii_chain_solve used. */
int moshii_last_launch_info(char* kernel_name, int32_t name_cap, int32_t* lds_bytes, int32_t* block_threads);

#ifdef __cplusplus
}
#endif
#endif /* MOSHII_H */
