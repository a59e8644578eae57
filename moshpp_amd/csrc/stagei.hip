// MoSh++ Stage-I on gfx950: shape (betas) + latent markers + one pose per picked frame, solved jointly (chmosh.py:83-455).
//
// Everything numeric runs on the device in f64: the canonical body, the re-evaluated marker attachment (3-NN, local frames,
// coefficients), the signed point-to-surface distance (exhaustive nearest triangle), the posed marker vertices with their pose and
// shape Jacobians, the residual rows and the dense Jacobian, the normal equations (tiled SYRK), the Cholesky solve.  The host
// drives Powell's dogleg (the trust-region bookkeeping on n-vectors) and the four annealing rounds.
//
// Kernels are written with block-strided loops over TID/NT.  tests/emu compiles this very file with g++ against a stand-in
// <hip/hip_runtime.h> that runs every workgroup as fibers on the CPU (test infrastructure: the product library is built from this
// file by hipcc only); the workgroup size S1_TPB and the lanes per vertex S1_VL can be set from the command line for that build
// (fewer fibers per workgroup: the kernels are correct for any power of two >= 64 / >= 1, the summation order follows them).
#include <hip/hip_runtime.h>
#include "stagei_views.h"
#include "../../include/moshii.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>

#define KERNEL __global__ void
#define KERNEL_LB(n) __global__ void __launch_bounds__(n)
#define DEVFN __device__ static inline
#define TID ((int)threadIdx.x)
#define NT ((int)blockDim.x)
#define BX ((int)blockIdx.x)
#define BY ((int)blockIdx.y)
#define SYNC() __syncthreads()
#define SHARED __shared__
#ifndef S1_TPB
#define S1_TPB 256
#endif
#define S1_CHOL_TPB 1024
#define S1_SURF_TPB (S1_TPB >= 256 ? 1024 : S1_TPB)
#define DYN_LDS(name) extern __shared__ double name[]
#define LAUNCH_LDS(k, gx, gy, nt, bytes, stream, ...) hipLaunchKernelGGL(k, dim3(gx, gy), dim3(nt), bytes, stream, __VA_ARGS__)
#define S1_TRSM_TPB 64
#ifndef S1_VL
#define S1_VL 64          // lanes that share one vertex's pose-corrective dot products
#endif
#define LAUNCH(k, gx, gy, nt, stream, ...) hipLaunchKernelGGL(k, dim3(gx, gy), dim3(nt), 0, stream, __VA_ARGS__)

#define S1_NMAX 4096     // unknowns (one column of the factor is staged in LDS)
#define S1_FSMAX 120     // per-frame unknowns the elimination kernel keeps in LDS (120 x 121 doubles = 116 KB)
#define S1_NWMAX 16      // non-zero skinning weights per vertex the vertex kernel keeps in registers

namespace {

struct S1Dims {
    int F, M, nb, NPZ;        // frames, markers, free betas, poses evaluated (F + canonical)
    int V, K, P, NP, body_dof, hand_dof, nhand_full, NBtot, nfeat;
    int nfaces, npid, nbody, nfinger, npose_prior, G;
    int n, ldn, R;            // unknowns, Jacobian pitch, residual rows
    int o_ml, o_pose, o_b;    // column offsets
    int r_data, r_prior, r_init, r_beta, r_surf, r_poseH, r_head;   // row offsets
    int nhead, nhead_rows;    // head-marker correlation term: C[nhead_rows][nhead]
    // the free shape block: shared betas (columns [0, nb) of shapedirs, one value set) or, per_frame = 1, the expression columns
    // [shape_start, shape_start + nb) with one value set per frame (optimize_face); shape_free: its columns exist in this round
    int per_frame, shape_start, shape_free, nface, r_poseF;
    int ncan;                 // canonical vertex list length = 9 M  [closest | closest0 | nearest-triangle vertices]
};

struct S1Ptr {
    // model
    const int* parents; const unsigned long long* anc;
    const double *vt, *shapedirs, *posedirs, *weights, *Jreg, *hands_mean, *comps;
    // prior
    const double *means, *chols, *neglogw;
    // problem constants
    const int* faces; const int* v2f_ptr; const int* v2f; const unsigned char* excl;
    const int* obs_ids; const int* obs_off; const double* obs; const double* m2b; const double* wt_init;
    const int* colmap;        // [NP] column of a pose variable inside one frame's block, -1 if frozen
    const int* body_ids; const int* finger_ids; const int* face_ids; double w_poseF;
    int* cl0; double* coef0;
    const int* head_ids; const double* head_C; double w_init_head;
    double *loss, *dinit;     // [M][3] init loss, [M][3][nb] d init / d betas (inputs of the head term)
    // evaluation point
    double *pose, *trans, *ml, *betas;
    // setup products
    double *J0, *JS;          // [K][3], [K][nb][3]
    // per pose
    double *fp, *Rl, *Jl, *Rw, *tw, *feat, *om, *Bm, *Jb, *q; int* featzero;
    // canonical body + attachment + surface
    double* can; int* cl; int* cl8; double *coef, *Fc, *dcdb; int* tv; double *sdist, *sdp, *sdabc; int* status;
    // vertex evaluations
    int* vlist;               // [NPZ][ncan] global vertex ids (frames use the first 3M entries of the canonical list)
    double *vv, *dvs, *dv, *Lb;
    // per (frame, attached vertex) forward state kept for the pose-Jacobian kernel: blended rotation, skinning influences,
    // rigidly attached positions
    double *vi_T, *vi_w, *vi_x; int *vi_n, *vi_j;
    // rows
    double *r, *Jm;
    // weights of the round: data, poseB, poseH, beta, surf, anneal
    double w_data, w_poseB, w_poseH, w_beta, w_surf, w_anneal;
};

// ---------------------------------------------------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------------------------------------------------
DEVFN void cross3(const double* a, const double* b, double* o) {
    o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}
DEVFN double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
DEVFN void matvec3(const double* Rm, const double* v, double* o) {
    for (int a = 0; a < 3; ++a) o[a] = Rm[3 * a] * v[0] + Rm[3 * a + 1] * v[1] + Rm[3 * a + 2] * v[2];
}
DEVFN void matmul3(const double* A, const double* B, double* C) {
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) C[3 * a + b] = A[3 * a] * B[b] + A[3 * a + 1] * B[3 + b] + A[3 * a + 2] * B[6 + b];
}
DEVFN void skew3(const double* v, double* Km) {
    Km[0] = 0; Km[1] = -v[2]; Km[2] = v[1]; Km[3] = v[2]; Km[4] = 0; Km[5] = -v[0]; Km[6] = -v[1]; Km[7] = v[0]; Km[8] = 0;
}

// R = I + a K + b K^2, Jl = I + b K + c K^2 with the series of the oracle below |r|^2 < 1e-6
DEVFN void rodrigues_dev(const double* r, double* Rm, double* Jm) {
    double t2 = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
    double a, b, c;
    if (t2 < 1e-6) {
        a = 1.0 - t2 / 6.0 + t2 * t2 / 120.0; b = 0.5 - t2 / 24.0 + t2 * t2 / 720.0; c = 1.0 / 6.0 - t2 / 120.0 + t2 * t2 / 5040.0;
    } else {
        double t = sqrt(t2);
        a = sin(t) / t; b = (1.0 - cos(t)) / t2; c = (t - sin(t)) / (t2 * t);
    }
    double Km[9], K2[9];
    skew3(r, Km); matmul3(Km, Km, K2);
    for (int i = 0; i < 9; ++i) {
        double e = (i == 0 || i == 4 || i == 8) ? 1.0 : 0.0;
        Rm[i] = e + a * Km[i] + b * K2[i];
        Jm[i] = e + b * Km[i] + c * K2[i];
    }
}

// frame vectors f1,f2,f3 (rows of Fm) of a vertex triple
DEVFN void frame_of(const double* v0, const double* v1, const double* v2, double* Fm) {
    double e1[3], e2[3], nv[3];
    for (int a = 0; a < 3; ++a) { e1[a] = v1[a] - v0[a]; e2[a] = v2[a] - v0[a]; }
    double l1 = sqrt(dot3(e1, e1));
    cross3(e1, e2, nv);
    double ln = sqrt(dot3(nv, nv));
    for (int a = 0; a < 3; ++a) { Fm[a] = e1[a] / l1; Fm[3 + a] = nv[a] / ln; }
    cross3(Fm, Fm + 3, Fm + 6);
}

// d(c0 f1 + c1 f2 + c2 f3)/d(v0,v1,v2): L[3][9] (without the identity of the anchor vertex)  -- markers_from_verts of the oracle
DEVFN void frame_jac(const double* v0, const double* v1, const double* v2, const double* c, double* L) {
    double e1[3], e2[3], nv[3], f1[3], f2[3];
    for (int a = 0; a < 3; ++a) { e1[a] = v1[a] - v0[a]; e2[a] = v2[a] - v0[a]; }
    double l1 = sqrt(dot3(e1, e1));
    cross3(e1, e2, nv);
    double ln = sqrt(dot3(nv, nv));
    for (int a = 0; a < 3; ++a) { f1[a] = e1[a] / l1; f2[a] = nv[a] / ln; }
    double D1[9], D2[9], S[9], T[9], df2e1[9], df2e2[9], sf1[9], sf2[9], df3e1[9], df3e2[9], tmp[9];
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) {
        double e = a == b ? 1.0 : 0.0;
        D1[3 * a + b] = (e - f1[a] * f1[b]) / l1;
        D2[3 * a + b] = (e - f2[a] * f2[b]) / ln;
    }
    skew3(e2, S); for (int i = 0; i < 9; ++i) S[i] = -S[i];      // dn/de1 = -[e2]x
    skew3(e1, T);                                                 // dn/de2 =  [e1]x
    matmul3(D2, S, df2e1); matmul3(D2, T, df2e2);
    skew3(f1, sf1); skew3(f2, sf2);
    matmul3(sf2, D1, tmp); matmul3(sf1, df2e1, df3e1);
    for (int i = 0; i < 9; ++i) df3e1[i] -= tmp[i];
    matmul3(sf1, df2e2, df3e2);
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) {
        double de1 = c[0] * D1[3 * a + b] + c[1] * df2e1[3 * a + b] + c[2] * df3e1[3 * a + b];
        double de2 = c[1] * df2e2[3 * a + b] + c[2] * df3e2[3 * a + b];
        L[9 * a + b] = -de1 - de2; L[9 * a + 3 + b] = de1; L[9 * a + 6 + b] = de2;
    }
}

// closest point on triangle (Ericson) with the part code: 0 interior, 1 ab, 2 bc, 3 ca, 4 a, 5 b, 6 c
DEVFN int closest_on_tri(const double* p, const double* a, const double* b, const double* c, double* q) {
    double ab[3], ac[3], ap[3], bp[3], cp[3];
    for (int i = 0; i < 3; ++i) { ab[i] = b[i] - a[i]; ac[i] = c[i] - a[i]; ap[i] = p[i] - a[i]; bp[i] = p[i] - b[i]; cp[i] = p[i] - c[i]; }
    double d1 = dot3(ab, ap), d2 = dot3(ac, ap);
    if (d1 <= 0 && d2 <= 0) { for (int i = 0; i < 3; ++i) q[i] = a[i]; return 4; }
    double d3 = dot3(ab, bp), d4 = dot3(ac, bp);
    if (d3 >= 0 && d4 <= d3) { for (int i = 0; i < 3; ++i) q[i] = b[i]; return 5; }
    double vc = d1 * d4 - d3 * d2;
    if (vc <= 0 && d1 >= 0 && d3 <= 0) { double t = d1 / (d1 - d3); for (int i = 0; i < 3; ++i) q[i] = a[i] + t * ab[i]; return 1; }
    double d5 = dot3(ab, cp), d6 = dot3(ac, cp);
    if (d6 >= 0 && d5 <= d6) { for (int i = 0; i < 3; ++i) q[i] = c[i]; return 6; }
    double vb = d5 * d2 - d1 * d6;
    if (vb <= 0 && d2 >= 0 && d6 <= 0) { double t = d2 / (d2 - d6); for (int i = 0; i < 3; ++i) q[i] = a[i] + t * ac[i]; return 3; }
    double va = d3 * d6 - d5 * d4;
    if (va <= 0 && (d4 - d3) >= 0 && (d5 - d6) >= 0) {
        double t = (d4 - d3) / ((d4 - d3) + (d5 - d6));
        for (int i = 0; i < 3; ++i) q[i] = b[i] + t * (c[i] - b[i]);
        return 2;
    }
    double den = 1.0 / (va + vb + vc);
    double v = vb * den, w = vc * den;
    for (int i = 0; i < 3; ++i) q[i] = a[i] + ab[i] * v + ac[i] * w;
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------------
// setup: J0 = Jreg . v_template, JS = Jreg . shapedirs[:, :, :nb]            grid (K, nb + 1)
// ---------------------------------------------------------------------------------------------------------------------------
KERNEL k_s1_setup(S1Dims d, S1Ptr p) {
    SHARED double red[3 * 256];
    int k = BX, e = BY;     // e == nb: template
    double acc[3] = {0, 0, 0};
    for (int v = TID; v < d.V; v += NT) {
        double w = p.Jreg[(size_t)k * d.V + v];
        if (w == 0.0) continue;
        for (int a = 0; a < 3; ++a)
            acc[a] += w * (e == d.nb ? p.vt[3 * v + a] : p.shapedirs[((size_t)v * 3 + a) * d.NBtot + d.shape_start + e]);
    }
    for (int a = 0; a < 3; ++a) red[a * 256 + TID] = acc[a];
    SYNC();
    if (TID == 0) {
        for (int a = 0; a < 3; ++a) {
            double s = 0;
            for (int t = 0; t < NT; ++t) s += red[a * 256 + t];
            if (e == d.nb) p.J0[3 * k + a] = s; else p.JS[((size_t)k * d.nb + e) * 3 + a] = s;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// per pose (grid NPZ): fullpose, Rodrigues, joints at the current betas, forward kinematics, shape derivative of the joint
// world positions (q), pose features, rotation axes, corrective derivative matrices
// ---------------------------------------------------------------------------------------------------------------------------
KERNEL k_s1_pose(S1Dims d, S1Ptr p) {
    int z = BX;
    const double* pose = p.pose + (size_t)z * d.NP;
    double* fp = p.fp + (size_t)z * d.P;
    double* Rl = p.Rl + (size_t)z * d.K * 9; double* Jl = p.Jl + (size_t)z * d.K * 9;
    double* Rw = p.Rw + (size_t)z * d.K * 9; double* tw = p.tw + (size_t)z * d.K * 3;
    double* Jb = p.Jb + (size_t)z * d.K * 3;
    double* q = p.q + (size_t)z * d.K * d.nb * 3;
    double* feat = p.feat + (size_t)z * d.nfeat;
    double* om = p.om + (size_t)z * d.K * 9; double* Bm = p.Bm + (size_t)z * d.K * 27;
    for (int i = TID; i < d.P; i += NT) {
        if (i < d.body_dof) fp[i] = pose[i];
        else {
            int c = i - d.body_dof;
            double s = p.hands_mean ? p.hands_mean[c] : 0.0;
            for (int h = 0; h < d.hand_dof; ++h) s += pose[d.body_dof + h] * p.comps[(size_t)h * d.nhand_full + c];
            fp[i] = s;
        }
    }
    for (int i = TID; i < 3 * d.K; i += NT) {
        double s = p.J0[i];
        int k = i / 3, a = i % 3;
        const double* bz = p.betas + (d.per_frame ? (size_t)z * d.nb : 0);
        for (int e = 0; e < d.nb; ++e) s += p.JS[((size_t)k * d.nb + e) * 3 + a] * bz[e];
        Jb[i] = s;
    }
    SYNC();
    for (int k = TID; k < d.K; k += NT) rodrigues_dev(fp + 3 * k, Rl + 9 * k, Jl + 9 * k);
    SYNC();
    // forward kinematics through LDS (the chain is serial; global round trips per joint would dominate)
    SHARED double sRl[64 * 9], sRw[64 * 9], sJb[64 * 3], stw[64 * 3]; SHARED int spar[64];
    for (int i = TID; i < 9 * d.K; i += NT) sRl[i] = Rl[i];
    for (int i = TID; i < 3 * d.K; i += NT) sJb[i] = Jb[i];
    for (int i = TID; i < d.K; i += NT) spar[i] = p.parents[i];
    SYNC();
    // the tree level by level, one thread per joint of the level (thread 0 walking all K joints through LDS was a quarter of this kernel)
    SHARED int sdep[64]; SHARED int smaxd;
    for (int j = TID; j < d.K; j += NT) {
        int dep = 0;
        for (int a = j; a > 0; a = spar[a]) ++dep;
        sdep[j] = dep;
    }
    SYNC();
    if (TID == 0) { int m = 0; for (int j = 0; j < d.K; ++j) m = (sdep[j] > m) ? sdep[j] : m; smaxd = m; }
    SYNC();
    const int maxd = smaxd;
    for (int lvl = 0; lvl <= maxd; ++lvl) {
        for (int j = TID; j < d.K; j += NT) {
            if (sdep[j] != lvl) continue;
            if (j == 0) {
                for (int i = 0; i < 9; ++i) sRw[i] = sRl[i];
                for (int a = 0; a < 3; ++a) stw[a] = sJb[a];
            } else {
                int pa = spar[j];
                matmul3(sRw + 9 * pa, sRl + 9 * j, sRw + 9 * j);
                double dj[3] = {sJb[3 * j] - sJb[3 * pa], sJb[3 * j + 1] - sJb[3 * pa + 1], sJb[3 * j + 2] - sJb[3 * pa + 2]}, o[3];
                matvec3(sRw + 9 * pa, dj, o);
                for (int a = 0; a < 3; ++a) stw[3 * j + a] = o[a] + stw[3 * pa + a];
            }
        }
        SYNC();
    }
    for (int i = TID; i < 9 * d.K; i += NT) Rw[i] = sRw[i];
    for (int i = TID; i < 3 * d.K; i += NT) tw[i] = stw[i];
    SYNC();
    // q_je = dt_je - Rw_j JS_je, dt_0 = JS_0, dt_j = dt_par + Rw_par (JS_j - JS_par): level by level, one thread per (joint, coefficient); the
    // parent's q is read back from global memory behind the level's barrier (one thread per coefficient walking the whole tree waited for a
    // round trip to the L2 at every joint: ~50 of this kernel's 67 us)
    for (int lvl = 0; lvl <= maxd; ++lvl) {
        for (int it = TID; it < d.K * d.nb; it += NT) {
            const int j = it / d.nb, e = it - j * d.nb;
            if (sdep[j] != lvl) continue;
            const double* js = p.JS + ((size_t)j * d.nb + e) * 3;
            double dt[3];
            if (j == 0) { for (int a = 0; a < 3; ++a) dt[a] = js[a]; }
            else {
                int pa = spar[j];
                const double* jp = p.JS + ((size_t)pa * d.nb + e) * 3;
                // dt of the parent is recovered from its q: dt_par = q_par + Rw_par JS_par
                double rp[3], dj[3] = {js[0] - jp[0], js[1] - jp[1], js[2] - jp[2]}, o[3];
                matvec3(sRw + 9 * pa, jp, rp);
                matvec3(sRw + 9 * pa, dj, o);
                for (int a = 0; a < 3; ++a) dt[a] = q[((size_t)pa * d.nb + e) * 3 + a] + rp[a] + o[a];
            }
            double rj[3];
            matvec3(sRw + 9 * j, js, rj);
            for (int a = 0; a < 3; ++a) q[((size_t)j * d.nb + e) * 3 + a] = dt[a] - rj[a];
        }
        SYNC();
    }
    for (int i = TID; i < d.nfeat; i += NT) {
        int k = 1 + i / 9, c = i % 9;
        feat[i] = Rl[9 * k + c] - ((c == 0 || c == 4 || c == 8) ? 1.0 : 0.0);
    }
    for (int i = TID; i < 3 * d.K; i += NT) {
        int k = i / 3, c = i % 3;
        double col[3] = {Jl[9 * k + c], Jl[9 * k + 3 + c], Jl[9 * k + 6 + c]};
        double o[3];
        if (k == 0) { for (int a = 0; a < 3; ++a) o[a] = col[a]; }
        else matvec3(Rw + 9 * p.parents[k], col, o);
        for (int a = 0; a < 3; ++a) om[9 * k + 3 * c + a] = o[a];
        double Sk[9];
        skew3(col, Sk);
        matmul3(Sk, Rl + 9 * k, Bm + 27 * k + 9 * c);
    }
    // a pose without any rotation away from the rest pose (the canonical one) has no corrective offsets
    SHARED int nzany;     // (every thread that sees a non-zero feature stores the same 1: no reduction)
    if (TID == 0) nzany = 0;
    SYNC();
    int mine = 0;
    for (int i = TID; i < d.nfeat; i += NT) if (sRl[9 * (1 + i / 9) + i % 9] - ((i % 9 == 0 || i % 9 == 4 || i % 9 == 8) ? 1.0 : 0.0) != 0.0) mine = 1;
    if (mine) nzany = 1;
    SYNC();
    if (TID == 0) p.featzero[z] = nzany ? 0 : 1;
}

// ---------------------------------------------------------------------------------------------------------------------------
// vertex evaluation: grid (ceil(nlist / S1_TPB), number of poses); pose index z = zbase + BY, list = vlist[z], nlist entries.
// mode 0: positions only into `out` [nlist][3] (canonical full mesh); 1: v; 2: v + dvs; 3: v + dvs + dv (pose Jacobian)
// ---------------------------------------------------------------------------------------------------------------------------
// vl: lanes that share one vertex's pose-corrective dot products (S1_VL; 1 for the canonical full mesh, whose pose has none: 6 890 vertices on
// one lane in 64 were 80 us per evaluation)
KERNEL k_s1_verts(S1Dims d, S1Ptr p, int zbase, int nlist, int mode, double* out, int full_mesh, int vl) {
    SHARED double part[S1_TPB * 3];
    int z = zbase + BY;
    const int lane = TID % vl, wv = TID / vl;
    int a = (BX * NT + TID) / vl;
    const bool valid = a < nlist;
    int v = 0;
    const double* Rw = p.Rw + (size_t)z * d.K * 9; const double* tw = p.tw + (size_t)z * d.K * 3;
    const double* Jb = p.Jb + (size_t)z * d.K * 3; const double* feat = p.feat + (size_t)z * d.nfeat;
    const double* tr = p.trans + 3 * z;
    double vs[3] = {0, 0, 0}, vp[3];
    if (valid) {
        v = full_mesh ? a : p.vlist[(size_t)z * d.ncan + a];
        const bool fz = p.featzero[z] != 0;
        for (int c = 0; c < 3; ++c) {
            double sacc = p.vt[3 * v + c];
            const double* sd = p.shapedirs + ((size_t)v * 3 + c) * d.NBtot + d.shape_start;
            const double* bz = p.betas + (d.per_frame ? (size_t)z * d.nb : 0);
            for (int e = 0; e < d.nb; ++e) sacc += sd[e] * bz[e];
            vs[c] = sacc;
            const double* pd = p.posedirs + ((size_t)v * 3 + c) * d.nfeat;
            double t = 0;
            if (!fz) for (int i = lane; i < d.nfeat; i += vl) t += pd[i] * feat[i];      // lanes split the 9(K-1) features
            part[TID * 3 + c] = t;
        }
    }
    SYNC();
    if (!valid || lane != 0) return;
    for (int c = 0; c < 3; ++c) {
        double t = 0;
        for (int l = 0; l < vl; ++l) t += part[(wv * vl + l) * 3 + c];
        vp[c] = vs[c] + t;
    }
    int jj[S1_NWMAX]; double wj[S1_NWMAX], xj[S1_NWMAX][3];
    int nw = 0;
    const double* wrow = p.weights + (size_t)v * d.K;
    for (int j0 = 0; j0 < d.K; j0 += 8) {      // eight weights in flight (read one at a time in front of the branch, the row's K loads were a
        double w8[8];                           //  chain of K round trips: 20 of this kernel's 21-33 us); the joints in order as before
#pragma unroll
        for (int u = 0; u < 8; ++u) w8[u] = (j0 + u < d.K) ? wrow[j0 + u] : 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const double w = w8[u];
            const int j = j0 + u;
            if (w == 0.0) continue;
            if (nw >= S1_NWMAX) { p.status[0] = 3; return; }
            jj[nw] = j; wj[nw] = w;
            double df[3] = {vp[0] - Jb[3 * j], vp[1] - Jb[3 * j + 1], vp[2] - Jb[3 * j + 2]}, o[3];
            matvec3(Rw + 9 * j, df, o);
            for (int c = 0; c < 3; ++c) xj[nw][c] = o[c] + tw[3 * j + c];
            ++nw;
        }
    }
    double pos[3] = {tr[0], tr[1], tr[2]}, Trot[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int s = 0; s < nw; ++s) {
        for (int c = 0; c < 3; ++c) pos[c] += wj[s] * xj[s][c];
        for (int i = 0; i < 9; ++i) Trot[i] += wj[s] * Rw[9 * jj[s] + i];
    }
    if (mode == 0) { for (int c = 0; c < 3; ++c) out[3 * (size_t)a + c] = pos[c]; return; }
    size_t slot = (size_t)z * d.ncan + a;
    for (int c = 0; c < 3; ++c) p.vv[3 * slot + c] = pos[c];
    if (mode < 2) return;
    if (d.nb) {
        const double* q = p.q + (size_t)z * d.K * d.nb * 3;
        for (int e = 0; e < d.nb; ++e) {
            const int ec = d.shape_start + e;
            double se[3] = {p.shapedirs[((size_t)v * 3 + 0) * d.NBtot + ec], p.shapedirs[((size_t)v * 3 + 1) * d.NBtot + ec],
                            p.shapedirs[((size_t)v * 3 + 2) * d.NBtot + ec]}, o[3];
            matvec3(Trot, se, o);
            for (int s = 0; s < nw; ++s) for (int c = 0; c < 3; ++c) o[c] += wj[s] * q[((size_t)jj[s] * d.nb + e) * 3 + c];
            for (int c = 0; c < 3; ++c) p.dvs[(slot * 3 + c) * d.nb + e] = o[c];
        }
    }
    if (mode < 3) return;
    {   // keep the forward state for k_s1_vjac (frames only: slots z * 3M + a)
        size_t fs = (size_t)z * 3 * d.M + a;
        for (int i = 0; i < 9; ++i) p.vi_T[fs * 9 + i] = Trot[i];
        p.vi_n[fs] = nw;
        for (int s2 = 0; s2 < nw; ++s2) {
            p.vi_j[fs * S1_NWMAX + s2] = jj[s2]; p.vi_w[fs * S1_NWMAX + s2] = wj[s2];
            for (int c = 0; c < 3; ++c) p.vi_x[(fs * S1_NWMAX + s2) * 3 + c] = xj[s2][c];
        }
    }
}

// pose Jacobian of the attached vertices: one thread per (vertex, joint k), its three dofs:
//   dv = omega_kc x (S_k - W_k tw_k) + Trot . (posedirs[v, :, 9(k-1):9k] . vec(B_kc))            grid (ceil(3M K / S1_TPB), F)
KERNEL k_s1_vjac(S1Dims d, S1Ptr p, int zbase) {
    int z = zbase + BY;
    int it = BX * NT + TID;
    if (it >= 3 * d.M * d.K) return;
    int a = it / d.K, k = it % d.K;
    int v = p.vlist[(size_t)z * d.ncan + a];
    size_t fs = (size_t)z * 3 * d.M + a;
    const double* tw = p.tw + (size_t)z * d.K * 3;
    const double* om = p.om + (size_t)z * d.K * 9; const double* Bm = p.Bm + (size_t)z * d.K * 27;
    const double* Trot = p.vi_T + fs * 9;
    const int nw = p.vi_n[fs];
    unsigned long long am = p.anc[k];
    double Sk[3] = {0, 0, 0}, Wk = 0;
    for (int s2 = 0; s2 < nw; ++s2) if ((am >> p.vi_j[fs * S1_NWMAX + s2]) & 1ull) {
        double w = p.vi_w[fs * S1_NWMAX + s2];
        Wk += w;
        for (int c = 0; c < 3; ++c) Sk[c] += w * p.vi_x[(fs * S1_NWMAX + s2) * 3 + c];
    }
    double arm[3] = {Sk[0] - Wk * tw[3 * k], Sk[1] - Wk * tw[3 * k + 1], Sk[2] - Wk * tw[3 * k + 2]};
    double* dv = p.dv + fs * 3 * d.P;      // [3][P]
    for (int c = 0; c < 3; ++c) {
        double o[3];
        cross3(om + 9 * k + 3 * c, arm, o);
        if (k >= 1) {
            double pc[3], t[3];
            for (int i = 0; i < 3; ++i) {
                const double* pd = p.posedirs + ((size_t)v * 3 + i) * d.nfeat + 9 * (k - 1);
                double sacc = 0;
                for (int e = 0; e < 9; ++e) sacc += pd[e] * Bm[27 * k + 9 * c + e];
                pc[i] = sacc;
            }
            matvec3(Trot, pc, t);
            for (int i = 0; i < 3; ++i) o[i] += t[i];
        }
        for (int i = 0; i < 3; ++i) dv[(size_t)i * d.P + 3 * k + c] = o[i];
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// attachment: the three nearest canonical vertices of every latent marker (grid M); ties go to the lower vertex id
// ---------------------------------------------------------------------------------------------------------------------------
// (value, index) minimum over the workgroup, lowest index on ties, left in bd[0] / bi[0]: a halving tree (NT is a power of two; the
// emulation build has NT = 1) -- thread 0 walking all 256 entries cost more than the search it finished
// (distance, index) minimum of the workgroup into bd[0] / bi[0], ties to the lower index -- so the result does not depend on the order of the
// comparisons: inside a wavefront by lane shuffles, across the wavefronts through their first words (the LDS tree it replaces took log2(NT)
// barriers a search, 8 searches a marker in k_s1_knn)
#define S1_BLOCK_ARGMIN(bd, bi) \
    { \
        double vd_ = bd[TID]; int vi_ = bi[TID]; \
        for (int o_ = 32; o_ > 0; o_ >>= 1) { \
            const double od_ = __shfl_down(vd_, o_); const int oi_ = __shfl_down(vi_, o_); \
            if (od_ < vd_ || (od_ == vd_ && oi_ < vi_)) { vd_ = od_; vi_ = oi_; } \
        } \
        SYNC(); \
        if ((TID & 63) == 0) { bd[TID >> 6] = vd_; bi[TID >> 6] = vi_; } \
        SYNC(); \
        if (TID == 0) { \
            for (int w_ = 1; w_ < (NT + 63) / 64; ++w_) \
                if (bd[w_] < vd_ || (bd[w_] == vd_ && bi[w_] < vi_)) { vd_ = bd[w_]; vi_ = bi[w_]; } \
            bd[0] = vd_; bi[0] = vi_; \
        } \
        SYNC(); \
    }

#define S1_NNK 8          // neighbours kept per marker (transformed_lm.py:73: the kd-tree query asks for 8)
KERNEL k_s1_knn(S1Dims d, S1Ptr p, int* cl8_out) {
    // ONE pass over the vertices: every thread keeps the 8 nearest of its own stride in registers, ordered by (distance, vertex id) -- a fixed
    // insertion network, no indexing by a run-time value -- then eight rounds of "the nearest head wins and its owner moves on".  (Round 3
    // made eight passes over V, one per neighbour: 2.7 x the work of the three-neighbour search it replaced, in every residual and Jacobian
    // evaluation of the dogleg.)
    SHARED double bd[256]; SHARED int bi[256];
    int m = BX;
    const double* x = p.ml + 3 * m;
    double kd[S1_NNK]; int ki[S1_NNK];
    for (int s = 0; s < S1_NNK; ++s) { kd[s] = 1e300; ki[s] = 0x7fffffff; }
    for (int v = TID; v < d.V; v += NT) {
        if (p.excl[v]) continue;
        double dx = x[0] - p.can[3 * v], dy = x[1] - p.can[3 * v + 1], dz = x[2] - p.can[3 * v + 2];
        double cd = dx * dx + dy * dy + dz * dz; int ci = v;
        if (cd < kd[S1_NNK - 1] || (cd == kd[S1_NNK - 1] && ci < ki[S1_NNK - 1])) {
            for (int s = 0; s < S1_NNK; ++s) {   // (cd, ci) sinks to its place, what it displaces moves on down
                const bool lt = cd < kd[s] || (cd == kd[s] && ci < ki[s]);
                const double td = lt ? kd[s] : cd; const int ti = lt ? ki[s] : ci;
                kd[s] = lt ? cd : kd[s]; ki[s] = lt ? ci : ki[s];
                cd = td; ci = ti;
            }
        }
    }
    int found[S1_NNK];
    for (int pass = 0; pass < S1_NNK; ++pass) {
        bd[TID] = kd[0]; bi[TID] = ki[0];
        SYNC();
        S1_BLOCK_ARGMIN(bd, bi)
        const int win = bi[0];
        found[pass] = win;
        SYNC();
        if (ki[0] == win) {   // this thread's head was taken: its list moves up
            for (int s = 0; s + 1 < S1_NNK; ++s) { kd[s] = kd[s + 1]; ki[s] = ki[s + 1]; }
            kd[S1_NNK - 1] = 1e300; ki[S1_NNK - 1] = 0x7fffffff;
        }
    }
    if (TID == 0) for (int s = 0; s < S1_NNK; ++s) cl8_out[S1_NNK * m + s] = found[s];
}

// The attachment's three vertices per marker out of its 8 nearest (transformed_lm.py:88-101): the two nearest and, normally, the
// third -- but while ANY marker's e1 x e2 vanishes (three collinear neighbours: the reference's nrm() turns 0 / 0 into a NaN and
// tests for it) the third neighbour of EVERY marker moves on to the next nearest.  One workgroup; status[2] is raised only if the
// eighth neighbour still leaves a collinear triple.
KERNEL k_s1_pick3(S1Dims d, S1Ptr p, const int* cl8, int* cl_out) {
    SHARED int any_bad;
    int nn = 3;
    for (;;) {
        if (TID == 0) any_bad = 0;
        SYNC();
        for (int m = TID; m < d.M; m += NT) {
            const int* c8 = cl8 + S1_NNK * m;
            const double* v0 = p.can + 3 * c8[0]; const double* v1 = p.can + 3 * c8[1]; const double* v2 = p.can + 3 * c8[nn - 1];
            double e1[3], e2[3], cr[3];
            for (int a = 0; a < 3; ++a) { e1[a] = v1[a] - v0[a]; e2[a] = v2[a] - v0[a]; }
            cross3(e1, e2, cr);
            if (dot3(cr, cr) == 0.0) any_bad = 1;
        }
        SYNC();
        const int bad = any_bad;
        SYNC();
        if (!bad || nn >= S1_NNK || nn >= d.M) { if (bad && TID == 0) p.status[2] = 1; break; }   // (the reference's loop bound is the marker count)
        ++nn;
    }
    for (int m = TID; m < d.M; m += NT) {
        cl_out[3 * m + 0] = cl8[S1_NNK * m + 0]; cl_out[3 * m + 1] = cl8[S1_NNK * m + 1]; cl_out[3 * m + 2] = cl8[S1_NNK * m + nn - 1];
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// signed distance of every latent marker to the canonical surface (grid M): exhaustive nearest triangle, part code, sign from
// the face / vertex normals, closed-form gradients wrt the marker and the triangle's vertices
// ---------------------------------------------------------------------------------------------------------------------------
DEVFN void vertex_normal(const S1Ptr& p, int v, double* n) {
    n[0] = n[1] = n[2] = 0;
    for (int i = p.v2f_ptr[v]; i < p.v2f_ptr[v + 1]; ++i) {
        const int* f = p.faces + 3 * p.v2f[i];
        double e1[3], e2[3], t[3];
        for (int a = 0; a < 3; ++a) { e1[a] = p.can[3 * f[1] + a] - p.can[3 * f[0] + a]; e2[a] = p.can[3 * f[2] + a] - p.can[3 * f[0] + a]; }
        cross3(e1, e2, t);
        for (int a = 0; a < 3; ++a) n[a] += t[a];
    }
    double ss = dot3(n, n);
    if (ss == 0) ss = 1e-10;
    double s = 1.0 / sqrt(ss);
    for (int a = 0; a < 3; ++a) n[a] *= s;
}

// (launched with S1_SURF_TPB threads: every thread's triangles are a chain of dependent gathers -- 59 of them at 256 threads were 40 of the
//  kernel's 47 us)
KERNEL k_s1_surface(S1Dims d, S1Ptr p) {
    SHARED double bd[S1_SURF_TPB]; SHARED int bi[S1_SURF_TPB];
    int m = BX;
    const double* x = p.ml + 3 * m;
    double best = 1e300; int besti = 0x7fffffff;
    for (int f = TID; f < d.nfaces; f += NT) {
        const int* fv = p.faces + 3 * f;
        double q[3];
        closest_on_tri(x, p.can + 3 * fv[0], p.can + 3 * fv[1], p.can + 3 * fv[2], q);
        double dx = x[0] - q[0], dy = x[1] - q[1], dz = x[2] - q[2];
        double d2 = dx * dx + dy * dy + dz * dz;
        if (d2 < best || (d2 == best && f < besti)) { best = d2; besti = f; }
    }
    bd[TID] = best; bi[TID] = besti;
    SYNC();
    S1_BLOCK_ARGMIN(bd, bi)
    if (TID == 0) {
        const int* fv = p.faces + 3 * bi[0];
        const double *A = p.can + 3 * fv[0], *B = p.can + 3 * fv[1], *C = p.can + 3 * fv[2];
        double q[3], diff[3], e1[3], e2[3], nrm[3], nh[3], nn[3];
        int part = closest_on_tri(x, A, B, C, q);
        for (int a = 0; a < 3; ++a) { diff[a] = x[a] - q[a]; e1[a] = B[a] - A[a]; e2[a] = C[a] - A[a]; }
        cross3(e1, e2, nrm);
        double s = sqrt(dot3(nrm, nrm));
        for (int a = 0; a < 3; ++a) nh[a] = nrm[a] / s;
        if (part == 0) { for (int a = 0; a < 3; ++a) nn[a] = nh[a]; }
        else if (part > 3) vertex_normal(p, fv[part - 4], nn);
        else {
            double n2[3];
            vertex_normal(p, fv[part - 1], nn); vertex_normal(p, fv[part % 3], n2);
            for (int a = 0; a < 3; ++a) nn[a] += n2[a];
        }
        double sd = dot3(diff, nn);
        double dir = sd > 0 ? 1.0 : (sd < 0 ? -1.0 : 0.0);
        double d2 = dot3(diff, diff), dist = sqrt(d2);
        p.sdist[m] = dist * dir;
        for (int s3 = 0; s3 < 3; ++s3) p.tv[3 * m + s3] = fv[s3];
        double* dp = p.sdp + 3 * m; double* da = p.sdabc + 9 * m;
        for (int i = 0; i < 9; ++i) da[i] = 0;
        if (part == 0) {
            double pa[3] = {x[0] - A[0], x[1] - A[1], x[2] - A[2]};
            double h = dot3(pa, nh), u[3], gb[3], gc[3];
            for (int a = 0; a < 3; ++a) u[a] = (pa[a] - h * nh[a]) / s;
            cross3(e2, u, gb); cross3(u, e1, gc);
            double sg = (h > 0 ? 1.0 : (h < 0 ? -1.0 : 0.0)) * dir;
            for (int a = 0; a < 3; ++a) { dp[a] = sg * nh[a]; da[3 + a] = sg * gb[a]; da[6 + a] = sg * gc[a]; da[a] = sg * (-nh[a] - gb[a] - gc[a]); }
        } else {
            double uh[3];
            for (int a = 0; a < 3; ++a) uh[a] = dist > 0 ? diff[a] / dist : 0.0;
            for (int a = 0; a < 3; ++a) dp[a] = dir * uh[a];
            if (part > 3) { for (int a = 0; a < 3; ++a) da[3 * (part - 4) + a] = -dir * uh[a]; }
            else {
                int i0 = part - 1, i1 = part % 3;
                const double* P0 = p.can + 3 * fv[i0]; const double* Q0 = p.can + 3 * fv[i1];
                double pq[3] = {Q0[0] - P0[0], Q0[1] - P0[1], Q0[2] - P0[2]}, np_[3] = {q[0] - P0[0], q[1] - P0[1], q[2] - P0[2]};
                double t = dot3(np_, pq) / dot3(pq, pq);
                for (int a = 0; a < 3; ++a) { da[3 * i0 + a] = -dir * (1 - t) * uh[a]; da[3 * i1 + a] = -dir * t * uh[a]; }
            }
        }
    }
}

// vertex lists: canonical pose gets [closest | closest0 | nearest-triangle vertices], every frame gets `closest`
KERNEL k_s1_lists(S1Dims d, S1Ptr p) {
    int i = BX * NT + TID;
    if (i >= 3 * d.M) return;
    for (int z = 0; z < d.F; ++z) p.vlist[(size_t)z * d.ncan + i] = p.cl[i];
    int* lc = p.vlist + (size_t)d.F * d.ncan;
    lc[i] = p.cl[i]; lc[3 * d.M + i] = p.cl0[i]; lc[6 * d.M + i] = p.tv[i];
}

// Arun / Procrustes fit R a + T ~= b (rigid_transformations.py:39-69) and cv2.Rodrigues(R) (:82); a, b [n][3]
DEVFN void rigid_fit(const double* A_, const double* B_, int cnt, double* rv, double* T) {
    double am[3] = {0, 0, 0}, bm[3] = {0, 0, 0};
    for (int m = 0; m < cnt; ++m) for (int i = 0; i < 3; ++i) { am[i] += A_[3 * m + i]; bm[i] += B_[3 * m + i]; }
    for (int i = 0; i < 3; ++i) { am[i] /= cnt; bm[i] /= cnt; }
    double G[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int m = 0; m < cnt; ++m) for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) G[i * 3 + j] += (A_[3 * m + i] - am[i]) * (B_[3 * m + j] - bm[j]);
    double V[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};      // one-sided Jacobi: G V = U S
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0;
        for (int p_ = 0; p_ < 2; ++p_) for (int q_ = p_ + 1; q_ < 3; ++q_) {
            double al = 0, be = 0, ga = 0;
            for (int i = 0; i < 3; ++i) { al += G[i * 3 + p_] * G[i * 3 + p_]; be += G[i * 3 + q_] * G[i * 3 + q_]; ga += G[i * 3 + p_] * G[i * 3 + q_]; }
            if (fabs(ga) <= 1e-300 || fabs(ga) <= 1e-17 * sqrt(al * be)) continue;
            off = fmax(off, fabs(ga) / sqrt(al * be));
            const double zeta = (be - al) / (2.0 * ga);
            const double t = ((zeta >= 0) ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
            const double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
            for (int i = 0; i < 3; ++i) {
                const double gp = G[i * 3 + p_], gq = G[i * 3 + q_];
                G[i * 3 + p_] = c * gp - sn * gq; G[i * 3 + q_] = sn * gp + c * gq;
                const double vp = V[i * 3 + p_], vq = V[i * 3 + q_];
                V[i * 3 + p_] = c * vp - sn * vq; V[i * 3 + q_] = sn * vp + c * vq;
            }
        }
        if (off < 1e-15) break;
    }
    double sv[3], U[9];
    int imin = 0, imax = 0;
    for (int c = 0; c < 3; ++c) {
        sv[c] = sqrt(G[c] * G[c] + G[3 + c] * G[3 + c] + G[6 + c] * G[6 + c]);
        if (sv[c] < sv[imin]) imin = c;
        if (sv[c] > sv[imax]) imax = c;
    }
    for (int c = 0; c < 3; ++c) for (int i = 0; i < 3; ++i) U[i * 3 + c] = (sv[c] > 0) ? G[i * 3 + c] / sv[c] : 0.0;
    if (sv[imin] <= 1e-13 * sv[imax]) {
        const int c1 = (imin + 1) % 3, c2 = (imin + 2) % 3;
        U[0 * 3 + imin] = U[1 * 3 + c1] * U[2 * 3 + c2] - U[2 * 3 + c1] * U[1 * 3 + c2];
        U[1 * 3 + imin] = U[2 * 3 + c1] * U[0 * 3 + c2] - U[0 * 3 + c1] * U[2 * 3 + c2];
        U[2 * 3 + imin] = U[0 * 3 + c1] * U[1 * 3 + c2] - U[1 * 3 + c1] * U[0 * 3 + c2];
    }
    double Rm[9];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Rm[i * 3 + j] = V[i * 3] * U[j * 3] + V[i * 3 + 1] * U[j * 3 + 1] + V[i * 3 + 2] * U[j * 3 + 2];
    const double det = Rm[0] * (Rm[4] * Rm[8] - Rm[5] * Rm[7]) - Rm[1] * (Rm[3] * Rm[8] - Rm[5] * Rm[6]) + Rm[2] * (Rm[3] * Rm[7] - Rm[4] * Rm[6]);
    if (det < 0.0) for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Rm[i * 3 + j] -= 2.0 * V[i * 3 + imin] * U[j * 3 + imin];
    double rx = Rm[7] - Rm[5], ry = Rm[2] - Rm[6], rz = Rm[3] - Rm[1];
    const double sn = sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
    double c = (Rm[0] + Rm[4] + Rm[8] - 1.0) * 0.5;
    c = fmin(1.0, fmax(-1.0, c));
    const double theta = acos(c);
    if (sn < 1e-5) {
        if (c > 0) { rv[0] = rv[1] = rv[2] = 0.0; }
        else {
            double x = sqrt(fmax((Rm[0] + 1.0) * 0.5, 0.0));
            double y = sqrt(fmax((Rm[4] + 1.0) * 0.5, 0.0)) * ((Rm[1] >= 0) ? 1.0 : -1.0);
            double z = sqrt(fmax((Rm[8] + 1.0) * 0.5, 0.0)) * ((Rm[2] >= 0) ? 1.0 : -1.0);
            if (fabs(x) < fabs(y) && fabs(x) < fabs(z) && ((Rm[5] > 0) != (y * z > 0))) z = -z;
            const double nn = sqrt(x * x + y * y + z * z);
            rv[0] = x * theta / nn; rv[1] = y * theta / nn; rv[2] = z * theta / nn;
        }
    } else {
        const double k = 0.5 * theta / sn;
        rv[0] = rx * k; rv[1] = ry * k; rv[2] = rz * k;
    }
    for (int i = 0; i < 3; ++i) T[i] = bm[i] - (Rm[i * 3] * am[0] + Rm[i * 3 + 1] * am[1] + Rm[i * 3 + 2] * am[2]);
}

// rigid start of every frame (grid F): markers simulated at the uploaded point vs the observations; out[f] = [rotvec, T]
KERNEL k_s1_rigid(S1Dims d, S1Ptr p, double* sim, double* out, int fbase) {
    int f = fbase + BX;
    int o0 = p.obs_off[f], nobs = p.obs_off[f + 1] - o0;
    const double* vv = p.vv + 3 * ((size_t)f * d.ncan);
    for (int k = TID; k < nobs; k += NT) {
        int m = p.obs_ids[o0 + k];
        const double* v0 = vv + 9 * m; const double* c = p.coef + 3 * m;
        double Fm[9];
        frame_of(v0, v0 + 3, v0 + 6, Fm);
        for (int a = 0; a < 3; ++a) sim[3 * (size_t)(o0 + k) + a] = v0[a] + c[0] * Fm[a] + c[1] * Fm[3 + a] + c[2] * Fm[6 + a];
    }
    SYNC();
    if (TID == 0) rigid_fit(sim + 3 * (size_t)o0, p.obs + 3 * (size_t)o0, nobs, out + 6 * f, out + 6 * f + 3);
}

// simulated markers of every frame at the uploaded point (all M latent markers): out[F][M][3]        grid F
KERNEL k_s1_simall(S1Dims d, S1Ptr p, double* out, int fbase) {
    int f = fbase + BX;
    const double* vv = p.vv + 3 * ((size_t)f * d.ncan);
    for (int m = TID; m < d.M; m += NT) {
        const double* v0 = vv + 9 * m; const double* c = p.coef + 3 * m;
        double Fm[9];
        frame_of(v0, v0 + 3, v0 + 6, Fm);
        for (int a = 0; a < 3; ++a) out[((size_t)f * d.M + m) * 3 + a] = v0[a] + c[0] * Fm[a] + c[1] * Fm[3 + a] + c[2] * Fm[6 + a];
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// shared rows (grid ceil(M / S1_TPB)): coefficients and canonical frames of the attachment, their shape derivative, the
// init / surf / beta rows.  want_J = 0 skips the Jacobian.
// ---------------------------------------------------------------------------------------------------------------------------
KERNEL k_s1_shared(S1Dims d, S1Ptr p, int want_J, int write_rows) {
    int m = BX * NT + TID;
    if (m >= d.M) return;
    const size_t zc = (size_t)d.F * d.ncan;                         // canonical slots
    const double* vc = p.vv + 3 * (zc + 3 * m);                     // v0c v1c v2c
    double Fc[9], diff[3], coef[3];
    frame_of(vc, vc + 3, vc + 6, Fc);
    for (int a = 0; a < 3; ++a) diff[a] = p.ml[3 * m + a] - vc[a];
    for (int i = 0; i < 3; ++i) coef[i] = dot3(diff, Fc + 3 * i);
    if (!(coef[0] == coef[0]) || !(coef[1] == coef[1]) || !(coef[2] == coef[2])) p.status[2] = 1;   // collinear neighbours
    for (int i = 0; i < 9; ++i) p.Fc[9 * m + i] = Fc[i];
    for (int i = 0; i < 3; ++i) p.coef[3 * m + i] = coef[i];
    const int nb = d.per_frame ? 0 : d.nb;          // the canonical body does not see per-frame expressions
    if (want_J && nb) {
        // dc_i/dbeta = (diff^T df_i/dV - f_i^T [I 0 0]) dV/dbeta
        const double* dV = p.dvs + (zc + 3 * m) * 3 * nb;           // [9][nb]
        for (int i = 0; i < 3; ++i) {
            double unit[3] = {0, 0, 0}, L[27], row[9];
            unit[i] = 1.0;
            frame_jac(vc, vc + 3, vc + 6, unit, L);
            for (int b = 0; b < 9; ++b) row[b] = diff[0] * L[b] + diff[1] * L[9 + b] + diff[2] * L[18 + b];
            for (int a = 0; a < 3; ++a) row[a] -= Fc[3 * i + a];
            for (int e = 0; e < nb; ++e) {
                double s = 0;
                for (int b = 0; b < 9; ++b) s += row[b] * dV[(size_t)b * nb + e];
                p.dcdb[((size_t)m * 3 + i) * nb + e] = s;
            }
        }
    }
    // init rows: (ml - init(betas)) wt_m anneal, init = v0 + F(can[closest0]) coef0
    {
        const double* v0 = p.vv + 3 * (zc + 3 * d.M + 3 * m);
        double F0[9], L0[27];
        frame_of(v0, v0 + 3, v0 + 6, F0);
        const double* c0 = p.coef0 + 3 * m;
        double w = p.wt_init[m] * p.w_anneal;
        for (int a = 0; a < 3; ++a) {
            double init = v0[a] + c0[0] * F0[a] + c0[1] * F0[3 + a] + c0[2] * F0[6 + a];
            if (write_rows) p.r[d.r_init + 3 * m + a] = (p.ml[3 * m + a] - init) * w;
            p.loss[3 * m + a] = p.ml[3 * m + a] - init;
        }
        if (want_J) {
            if (write_rows) for (int a = 0; a < 3; ++a) p.Jm[(size_t)(d.r_init + 3 * m + a) * d.ldn + d.o_ml + 3 * m + a] = w;
            if (nb) {
                frame_jac(v0, v0 + 3, v0 + 6, c0, L0);
                for (int a = 0; a < 3; ++a) L0[9 * a + a] += 1.0;
                const double* dV = p.dvs + (zc + 3 * d.M + 3 * m) * 3 * nb;
                for (int a = 0; a < 3; ++a) for (int e = 0; e < nb; ++e) {
                    double s = 0;
                    for (int b = 0; b < 9; ++b) s += L0[9 * a + b] * dV[(size_t)b * nb + e];
                    if (write_rows) p.Jm[(size_t)(d.r_init + 3 * m + a) * d.ldn + d.o_b + e] = -w * s;
                    p.dinit[((size_t)m * 3 + a) * nb + e] = s;
                }
            }
        }
    }
    // surf row
    if (!write_rows) return;
    p.r[d.r_surf + m] = (p.sdist[m] - p.m2b[m]) * p.w_surf;
    if (want_J) {
        double* Jr = p.Jm + (size_t)(d.r_surf + m) * d.ldn;
        for (int a = 0; a < 3; ++a) Jr[d.o_ml + 3 * m + a] = p.sdp[3 * m + a] * p.w_surf;
        if (nb) {
            const double* dV = p.dvs + (zc + 6 * d.M + 3 * m) * 3 * nb;
            for (int e = 0; e < nb; ++e) {
                double s = 0;
                for (int b = 0; b < 9; ++b) s += p.sdabc[9 * m + b] * dV[(size_t)b * nb + e];
                Jr[d.o_b + e] = s * p.w_surf;
            }
        }
    }
    if (m < nb) {
        p.r[d.r_beta + m] = p.betas[m] * p.w_beta;
        if (want_J) p.Jm[(size_t)(d.r_beta + m) * d.ldn + d.o_b + m] = p.w_beta;
    }
}

// head-marker correlation rows (chmosh.py:252-266, 362-369): C . init_loss[head ids] . wt      (one block, after k_s1_shared)
KERNEL k_s1_head(S1Dims d, S1Ptr p, int want_J) {
    for (int it = TID; it < d.nhead_rows * 3; it += NT) {
        int g = it / 3, a = it % 3;
        const double* Cg = p.head_C + (size_t)g * d.nhead;
        double s = 0;
        for (int h = 0; h < d.nhead; ++h) s += Cg[h] * p.loss[3 * p.head_ids[h] + a];
        p.r[d.r_head + it] = s * p.w_init_head;
        if (want_J) {
            double* Jr = p.Jm + (size_t)(d.r_head + it) * d.ldn;
            for (int h = 0; h < d.nhead; ++h) Jr[d.o_ml + 3 * p.head_ids[h] + a] = Cg[h] * p.w_init_head;
            for (int e = 0; e < (d.per_frame ? 0 : d.nb); ++e) {
                double t = 0;
                for (int h = 0; h < d.nhead; ++h) t += Cg[h] * p.dinit[((size_t)p.head_ids[h] * 3 + a) * d.nb + e];
                Jr[d.o_b + e] = -t * p.w_init_head;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// per-frame rows (grid F): data, poseB, poseH
// ---------------------------------------------------------------------------------------------------------------------------
KERNEL k_s1_rows(S1Dims d, S1Ptr p, int want_J, int fbase) {
    SHARED double score[64]; SHARED int kbest;
    int f = fbase + BX;
    const int M = d.M, nb = d.nb, npid = d.npid;
    const double* vv = p.vv + 3 * ((size_t)f * d.ncan);
    double* Lb = p.Lb + (size_t)f * M * 36;              // per marker: L[27] + posed frame F'[9]
    for (int m = TID; m < M; m += NT) {
        const double* v0 = vv + 9 * m;
        double L[27], Fp[9];
        frame_jac(v0, v0 + 3, v0 + 6, p.coef + 3 * m, L);
        for (int a = 0; a < 3; ++a) L[9 * a + a] += 1.0;
        frame_of(v0, v0 + 3, v0 + 6, Fp);
        for (int i = 0; i < 27; ++i) Lb[36 * m + i] = L[i];
        for (int i = 0; i < 9; ++i) Lb[36 * m + 27 + i] = Fp[i];
    }
    SYNC();
    int o0 = p.obs_off[f], nobs = p.obs_off[f + 1] - o0;
    const int colbase = d.o_pose + f * npid;
    // The pose columns of the data rows -- nobs x NP items of nine strided loads each, 24 a thread on one workgroup: 70 of a Jacobian call's 130 us
    // -- are dealt over the workgroups (frame, BY = 0 .. gridDim.y - 1); everything else is workgroup (frame, 0)'s.
    const int part = BY, nparts = (int)gridDim.y;
    auto pose_columns = [&]() {
        const double* dv = p.dv + (size_t)f * 3 * M * 3 * d.P;
        for (int it = part * NT + TID; it < nobs * d.NP; it += NT * nparts) {
            int k = it / d.NP, pid = it % d.NP, col = p.colmap[pid];
            if (col < 0) continue;
            int m = p.obs_ids[o0 + k];
            const double* L = Lb + 36 * m;
            double o[3] = {0, 0, 0};
            for (int s = 0; s < 3; ++s) for (int c = 0; c < 3; ++c) {
                const double* dvr = dv + ((size_t)(3 * m + s) * 3 + c) * d.P;
                double g;
                if (pid < d.body_dof) g = dvr[pid];
                else {
                    const double* cm = p.comps + (size_t)(pid - d.body_dof) * d.nhand_full;
                    g = 0;
                    for (int qd = 0; qd < d.nhand_full; ++qd) g += cm[qd] * dvr[d.body_dof + qd];
                }
                for (int a = 0; a < 3; ++a) o[a] += L[9 * a + 3 * s + c] * g;
            }
            for (int a = 0; a < 3; ++a) p.Jm[(size_t)(d.r_data + 3 * (o0 + k) + a) * d.ldn + colbase + col] = -p.w_data * o[a];
        }
    };
    if (part != 0) { if (want_J) pose_columns(); return; }
    // residual
    for (int i = TID; i < 3 * nobs; i += NT) {
        int k = i / 3, a = i % 3, m = p.obs_ids[o0 + k];
        const double* v0 = vv + 9 * m; const double* Fp = Lb + 36 * m + 27; const double* c = p.coef + 3 * m;
        double sim = v0[a] + c[0] * Fp[a] + c[1] * Fp[3 + a] + c[2] * Fp[6 + a];
        p.r[d.r_data + 3 * o0 + i] = (p.obs[3 * (size_t)(o0 + k) + a] - sim) * p.w_data;
    }
    if (want_J) {
        pose_columns();
        // trans, latent marker, betas
        for (int it = TID; it < nobs * 3; it += NT) {
            int k = it / 3, a = it % 3, m = p.obs_ids[o0 + k];
            double* Jr = p.Jm + (size_t)(d.r_data + 3 * (o0 + k) + a) * d.ldn;
            const double* L = Lb + 36 * m; const double* Fp = L + 27; const double* Fc = p.Fc + 9 * m;
            Jr[3 * f + a] = -p.w_data;
            for (int b = 0; b < 3; ++b) Jr[d.o_ml + 3 * m + b] = -p.w_data * (Fp[a] * Fc[b] + Fp[3 + a] * Fc[3 + b] + Fp[6 + a] * Fc[6 + b]);
            if (nb && d.shape_free) {
                const double* dV = p.dvs + ((size_t)f * d.ncan + 3 * m) * 3 * nb;
                const double* dc = p.dcdb + (size_t)m * 3 * nb;
                const int cb = d.o_b + (d.per_frame ? f * nb : 0);
                for (int e = 0; e < nb; ++e) {
                    double s = 0;
                    for (int b = 0; b < 9; ++b) s += L[9 * a + b] * dV[(size_t)b * nb + e];
                    if (!d.per_frame) for (int i = 0; i < 3; ++i) s += Fp[3 * i + a] * dc[(size_t)i * nb + e];
                    Jr[cb + e] = -p.w_data * s;
                }
            }
        }
    }
    // poseB: max-mixture prior on pose[body_ids]
    if (d.G > 0 && d.nbody > 0) {
        const int np_ = d.npose_prior;
        const double* pose = p.pose + (size_t)f * d.NP;
        double* ell = p.Lb + (size_t)d.F * M * 36 + (size_t)f * d.G * np_;      // scratch behind the L buffers
        for (int it = TID; it < d.G * np_; it += NT) {
            int g = it / np_, a = it % np_;
            double s = 0;
            for (int b = 0; b < np_; ++b) s += (pose[p.body_ids[b]] - p.means[(size_t)g * np_ + b]) * p.chols[((size_t)g * np_ + b) * np_ + a];
            ell[it] = 0.70710678118654752440 * s;
        }
        SYNC();
        for (int g = TID; g < d.G; g += NT) {
            double s = 0;
            for (int a = 0; a < np_; ++a) s += ell[g * np_ + a] * ell[g * np_ + a];
            score[g] = s + p.neglogw[g];
        }
        SYNC();
        if (TID == 0) { int kb = 0; for (int g = 1; g < d.G; ++g) if (score[g] < score[kb]) kb = g; kbest = kb; }
        SYNC();
        int g = kbest;
        int r0 = d.r_prior + f * (np_ + 1);
        for (int a = TID; a < np_; a += NT) p.r[r0 + a] = ell[g * np_ + a] * p.w_poseB;
        if (TID == 0) p.r[r0 + np_] = sqrt(p.neglogw[g]) * p.w_poseB;
        if (want_J) {
            for (int it = TID; it < np_ * np_; it += NT) {
                int a = it / np_, b = it % np_, col = p.colmap[p.body_ids[b]];
                if (col < 0) continue;
                p.Jm[(size_t)(r0 + a) * d.ldn + colbase + col] = p.w_poseB * 0.70710678118654752440 * p.chols[((size_t)g * np_ + b) * np_ + a];
            }
        }
        SYNC();
    }
    for (int k = TID; k < d.nface; k += NT) {          // poseF: the jaw (chmosh.py:394-396)
        int pid = p.face_ids[k];
        p.r[d.r_poseF + f * d.nface + k] = p.pose[(size_t)f * d.NP + pid] * p.w_poseF;
        if (want_J) p.Jm[(size_t)(d.r_poseF + f * d.nface + k) * d.ldn + colbase + p.colmap[pid]] = p.w_poseF;
    }
    if (d.per_frame && d.shape_free) {                  // expr: this frame's expression coefficients (:397)
        for (int e = TID; e < nb; e += NT) {
            p.r[d.r_beta + f * nb + e] = p.betas[(size_t)f * nb + e] * p.w_beta;
            if (want_J) p.Jm[(size_t)(d.r_beta + f * nb + e) * d.ldn + d.o_b + f * nb + e] = p.w_beta;
        }
    }
    for (int k = TID; k < d.nfinger; k += NT) {
        int pid = p.finger_ids[k];
        p.r[d.r_poseH + f * d.nfinger + k] = p.pose[(size_t)f * d.NP + pid] * p.w_poseH;
        if (want_J) p.Jm[(size_t)(d.r_poseH + f * d.nfinger + k) * d.ldn + colbase + p.colmap[pid]] = p.w_poseH;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// dense linear algebra on the row-major Jacobian Jm[R][ldn]
// ---------------------------------------------------------------------------------------------------------------------------
#define S1_T 32
#define S1_PB 32     // panel width of the blocked Cholesky further down
// flags[row chunk][column block] = 1 iff that 32 x 32 block of J holds a non-zero (most do not: a frame's pose columns appear only
// in that frame's rows)                                                    grid (ceil(R / 32), ceil(n / 32))
KERNEL k_s1_nzflags(const double* Jm, int R, int n, int ldn, int* flags) {
    SHARED int any;
    int rc = BX, cb = BY, mine = 0;
    if (TID == 0) any = 0;
    SYNC();
    for (int e = TID; e < S1_T * S1_T; e += NT) {
        int r = rc * S1_T + e / S1_T, c = cb * S1_T + e % S1_T;
        if (r < R && c < n && Jm[(size_t)r * ldn + c] != 0.0) mine = 1;
    }
    if (mine) any = 1;
    SYNC();
    if (TID == 0) flags[rc * ((n + S1_T - 1) / S1_T) + cb] = any;
}

#define S1_SYRK_CHUNKS 1024   // row chunks whose flags a tile keeps in LDS (32 768 rows; beyond: read one by one)
// A = J^T J, lower tiles (grid (nt, nt), tile (BX >= BY)); 256 threads, 2 x 2 outputs each; mirrored on write
KERNEL k_s1_syrk(const double* Jm, int R, int n, int ldn, double* A, const int* flags) {
    int ti = BX, tj = BY;
    if (tj > ti) return;
    __shared__ double Si[S1_T][S1_T + 1], Sj[S1_T][S1_T + 1];
    int tx = TID % 16, ty = TID / 16;
    double acc[2][2] = {{0, 0}, {0, 0}};
    const int ncb = (n + S1_T - 1) / S1_T;
    // which row chunks hold anything for this tile: fetched by all threads at once, so that the loop below knows its next chunk ahead of time
    SHARED unsigned char use[S1_SYRK_CHUNKS];
    const int nrc = (R + S1_T - 1) / S1_T;
    for (int rc = TID; rc < nrc && rc < S1_SYRK_CHUNKS; rc += NT) use[rc] = (flags[rc * ncb + ti] && flags[rc * ncb + tj]) ? 1 : 0;
    __syncthreads();
    // the next chunk's entries are fetched into registers while this one's products are formed (one chunk at a time the tile waited ~1.5 us for
    // every chunk's loads: 60-94 chunks on the shared columns' tiles)
    auto used = [&](int rc) -> bool {
        if (rc < S1_SYRK_CHUNKS) return use[rc] != 0;
        const int* fl = flags + rc * ncb; return fl[ti] && fl[tj];
    };
    auto next_used = [&](int rc) { while (rc < nrc && !used(rc)) ++rc; return rc; };
    constexpr int PER = S1_T * S1_T / 256;      // (launched with 256 threads)
    double ri[PER], rj[PER];
    auto fetch = [&](int rc) {
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int e = TID + u * 256, rr = e / S1_T, cc = e % S1_T;
            const int r = rc * S1_T + rr, ci = ti * S1_T + cc, cj = tj * S1_T + cc;
            ri[u] = (r < R && ci < n) ? Jm[(size_t)r * ldn + ci] : 0.0;
            rj[u] = (r < R && cj < n) ? Jm[(size_t)r * ldn + cj] : 0.0;
        }
    };
    int rc = next_used(0);
    if (rc < nrc) fetch(rc);
    while (rc < nrc) {
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int e = TID + u * 256, rr = e / S1_T, cc = e % S1_T;
            Si[rr][cc] = ri[u]; Sj[rr][cc] = rj[u];
        }
        __syncthreads();
        const int nx = next_used(rc + 1);
        if (nx < nrc) fetch(nx);
#pragma unroll 8
        for (int rr = 0; rr < S1_T; ++rr) {
            double a0 = Si[rr][ty], a1 = Si[rr][ty + 16], b0 = Sj[rr][tx], b1 = Sj[rr][tx + 16];
            acc[0][0] += a0 * b0; acc[0][1] += a0 * b1; acc[1][0] += a1 * b0; acc[1][1] += a1 * b1;
        }
        __syncthreads();
        rc = nx;
    }
    for (int u = 0; u < 2; ++u) for (int w = 0; w < 2; ++w) {
        int i = ti * S1_T + ty + 16 * u, j = tj * S1_T + tx + 16 * w;
        if (i < n && j < n) { A[(size_t)i * n + j] = acc[u][w]; A[(size_t)j * n + i] = acc[u][w]; }
    }
}

// partial[BY][n] = J[rows of chunk BY]^T r ; then y = sign * sum over chunks        grid (ceil(n / S1_TPB), S1_GT_CHUNKS)
#define S1_GT_CHUNKS 16
KERNEL k_s1_gemv_t(const double* Jm, const double* r, int R, int n, int ldn, double* partial) {
    int i = BX * NT + TID;
    if (i >= n) return;
    int per = (R + S1_GT_CHUNKS - 1) / S1_GT_CHUNKS, k0 = BY * per, k1 = (k0 + per < R) ? k0 + per : R;
    double sacc = 0;
    for (int k = k0; k < k1; ++k) sacc += Jm[(size_t)k * ldn + i] * r[k];
    partial[(size_t)BY * n + i] = sacc;
}
KERNEL k_s1_gemv_t_sum(const double* partial, int n, double sign, double* y) {
    int i = BX * NT + TID;
    if (i >= n) return;
    double sacc = 0;
    for (int c = 0; c < S1_GT_CHUNKS; ++c) sacc += partial[(size_t)c * n + i];
    y[i] = sign * sacc;
}

// y[R] = Mx[R][ld] . x[n]   (grid R; one block per row)
KERNEL k_s1_gemv(const double* Mx, const double* x, int n, int ld, double* y) {
    SHARED double red[256];
    int r = BX;
    double s = 0;
    for (int i = TID; i < n; i += NT) s += Mx[(size_t)r * ld + i] * x[i];
    red[TID] = s;
    SYNC();
    if (TID == 0) { double t = 0; for (int k = 0; k < NT; ++k) t += red[k]; y[r] = t; }
}

// the value lane `src` (a compile-time constant after unrolling) holds, in every lane: two v_readlane into a scalar pair (a handful of cycles;
// __shfl goes through ds_bpermute: ~2 500 of them made k_s1_chol_diag a 69 us kernel).  The CPU emulation keeps the shuffle.
DEVFN double s1_bcast(double v, int src) {
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, src), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), src);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
#else
    return __shfl(v, src);
#endif
}

// Blocked right-looking Cholesky of A[n][n] (lower), panel width S1_PB.  Per panel: k_s1_chol_diag (one workgroup: the diagonal
// block is factored and inverted in LDS), k_s1_chol_trsm (every row below: its 32 entries times the inverse) and k_s1_chol_update (one 32 x 32 tile of the
// trailing matrix per workgroup: A_ik -= L_i L_k^T).  status[1] = 1 if a pivot is not positive.
// GPU version: one wavefront, lane r keeps row r of the block in registers; pivots and columns travel by lane shuffles.
// Then lane c solves L x = e_c for column c of the inverse with the factor's entries broadcast the same way.
__global__ void __launch_bounds__(64) k_s1_chol_diag(double* A, int n, int j0, double* dinv, int* status) {
    const int jb = (n - j0) < S1_PB ? (n - j0) : S1_PB;
    const int r = threadIdx.x & 31;
    double a[S1_PB];
#pragma unroll
    for (int c = 0; c < S1_PB; ++c) a[c] = (r < jb && c <= r && c < jb) ? A[(size_t)(j0 + r) * n + j0 + c] : (r == c ? 1.0 : 0.0);
    int bad = 0;
#pragma unroll
    for (int c = 0; c < S1_PB; ++c) {
        double piv = s1_bcast(a[c], c);
        if (!(piv > 0)) { bad = 1; piv = 1.0; }
        const double dd = sqrt(piv);
        a[c] = (r == c) ? dd : a[c] / dd;            // rows above the diagonal hold zeros here
        const double lrc = a[c];
#pragma unroll
        for (int k = c + 1; k < S1_PB; ++k) {
            const double lkc = s1_bcast(a[c], k);
            if (r >= k) a[k] -= lrc * lkc;
        }
    }
    if (bad && threadIdx.x == 0) status[1] = 1;
    if (threadIdx.x < 32 && r < jb) {
#pragma unroll
        for (int c = 0; c < S1_PB; ++c) if (c <= r) A[(size_t)(j0 + r) * n + j0 + c] = a[c];
    }
    // inverse: lane c owns column c
    double x[S1_PB];
#pragma unroll
    for (int rr = 0; rr < S1_PB; ++rr) {
        double sacc = (rr == r) ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < S1_PB; ++k) if (k < rr) sacc -= s1_bcast(a[k], rr) * x[k];
        const double lrr = s1_bcast(a[rr], rr);   // (taken by every lane: the lanes left of the diagonal do not need it, but a shuffle inside
        x[rr] = (rr < r) ? 0.0 : sacc / lrr;    //  the conditional is a divergent collective to the CPU emulation of this file)
    }
    double* Di = dinv + (size_t)(j0 / S1_PB) * S1_PB * S1_PB;
    if (threadIdx.x < 32) {
#pragma unroll
        for (int rr = 0; rr < S1_PB; ++rr) Di[rr * S1_PB + r] = x[rr];
    }
}

// rows below the diagonal block: x = a . L_D^{-T}, i.e. x_c = sum_{k <= c} a_k Dinv[c][k]        grid ceil(rows / 64), 64 threads
KERNEL_LB(64) k_s1_chol_trsm(double* A, int n, int j0, const double* dinv) {
    SHARED double Di[S1_PB][S1_PB + 1];
    const int jb = (n - j0) < S1_PB ? (n - j0) : S1_PB;
    const double* Dg = dinv + (size_t)(j0 / S1_PB) * S1_PB * S1_PB;
    for (int e = TID; e < S1_PB * S1_PB; e += NT) Di[e / S1_PB][e % S1_PB] = Dg[e];
    SYNC();
    SHARED double Ap[S1_TRSM_TPB][S1_PB + 1];          // this workgroup's rows of the panel
    int i = j0 + jb + BX * NT + TID;
    if (i >= n) return;
    double* Ai = A + (size_t)i * n + j0;
    double* av = Ap[TID];
    for (int c = 0; c < S1_PB; ++c) av[c] = (c < jb) ? Ai[c] : 0.0;
    for (int c = S1_PB - 1; c >= 0; --c) {      // x_c needs a_k only for k <= c: overwrite in place, highest first
        double sacc = 0;
        for (int k = 0; k <= c; ++k) sacc += av[k] * Di[c][k];
        av[c] = sacc;
    }
    for (int c = 0; c < jb; ++c) Ai[c] = av[c];
}

// trailing update, tiles (BX >= BY) of the block that starts at row / column j0 + jb      grid (nt, nt), 256 threads
KERNEL k_s1_chol_update(double* A, int n, int j0, int jb) {
    int ti = BX, tj = BY;
    if (tj > ti) return;
    const int s0 = j0 + jb;
    __shared__ double Li[S1_PB][S1_PB + 1], Lk[S1_PB][S1_PB + 1];
    for (int e = TID; e < S1_PB * S1_PB; e += NT) {
        int r = e / S1_PB, c = e % S1_PB;
        int i = s0 + ti * S1_PB + r, k = s0 + tj * S1_PB + r;
        Li[r][c] = (i < n && c < jb) ? A[(size_t)i * n + j0 + c] : 0.0;
        Lk[r][c] = (k < n && c < jb) ? A[(size_t)k * n + j0 + c] : 0.0;
    }
    __syncthreads();
    int tx = TID % 16, ty = TID / 16;
    double acc[2][2] = {{0, 0}, {0, 0}};
#pragma unroll 8
    for (int c = 0; c < S1_PB; ++c) {
        double a0 = Li[ty][c], a1 = Li[ty + 16][c], b0 = Lk[tx][c], b1 = Lk[tx + 16][c];
        acc[0][0] += a0 * b0; acc[0][1] += a0 * b1; acc[1][0] += a1 * b0; acc[1][1] += a1 * b1;
    }
    for (int u = 0; u < 2; ++u) for (int w = 0; w < 2; ++w) {
        int i = s0 + ti * S1_PB + ty + 16 * u, k = s0 + tj * S1_PB + tx + 16 * w;
        if (i < n && k <= i) A[(size_t)i * n + k] -= acc[u][w];
    }
}

// L y = g, L^T x = y with the factor of the kernels above, one workgroup, panel by panel: the 32 x 32 diagonal block goes through
// LDS, the rest of the panel is applied in the coalesced direction (forward: one row per thread, 32 contiguous entries; backward:
// one earlier unknown per thread, reading 32 rows of the factor along the row).
KERNEL k_s1_tri_solve(const double* L, int n, const double* dinv, const double* g, double* out) {
    SHARED double y[S1_NMAX]; SHARED double Di[S1_PB][S1_PB + 1]; SHARED double yb[S1_PB];
    for (int i = TID; i < n; i += NT) y[i] = g[i];
    SYNC();
    for (int j0 = 0; j0 < n; j0 += S1_PB) {
        const int jb = (n - j0) < S1_PB ? (n - j0) : S1_PB;
        const double* Dg = dinv + (size_t)(j0 / S1_PB) * S1_PB * S1_PB;
        for (int e = TID; e < S1_PB * S1_PB; e += NT) Di[e / S1_PB][e % S1_PB] = Dg[e];
        SYNC();
        for (int c = TID; c < jb; c += NT) {            // y_blk <- L_D^{-1} y_blk
            double sacc = 0;
            for (int k = 0; k <= c; ++k) sacc += Di[c][k] * y[j0 + k];
            yb[c] = sacc;
        }
        SYNC();
        for (int c = TID; c < jb; c += NT) y[j0 + c] = yb[c];
        for (int i = j0 + jb + TID; i < n; i += NT) {
            const double* Li = L + (size_t)i * n + j0;
            double sacc = 0;
            for (int c = 0; c < jb; ++c) sacc += Li[c] * yb[c];
            y[i] -= sacc;
        }
        SYNC();
    }
    for (int j0 = ((n - 1) / S1_PB) * S1_PB; j0 >= 0; j0 -= S1_PB) {
        const int jb = (n - j0) < S1_PB ? (n - j0) : S1_PB;
        const double* Dg = dinv + (size_t)(j0 / S1_PB) * S1_PB * S1_PB;
        for (int e = TID; e < S1_PB * S1_PB; e += NT) Di[e / S1_PB][e % S1_PB] = Dg[e];
        SYNC();
        for (int c = TID; c < jb; c += NT) {            // x_blk <- L_D^{-T} y_blk
            double sacc = 0;
            for (int k = c; k < jb; ++k) sacc += Di[k][c] * y[j0 + k];
            yb[c] = sacc;
        }
        SYNC();
        for (int c = TID; c < jb; c += NT) y[j0 + c] = yb[c];
        for (int k = TID; k < j0; k += NT) {
            double sacc = 0;
            for (int c = 0; c < jb; ++c) sacc += L[(size_t)(j0 + c) * n + k] * yb[c];
            y[k] -= sacc;
        }
        SYNC();
    }
    for (int i = TID; i < n; i += NT) out[i] = y[i];
}

// ---------------------------------------------------------------------------------------------------------------------------
// Arrow structure of the Stage-I normal equations (the default solver; MOSHII_S1_SOLVER=dense switches it off): a frame's unknowns (trans, pose,
// per-frame expressions) are coupled to other frames only through the shared block (latent markers, betas).  Per frame one
// workgroup factors its diagonal block in LDS, inverts the factor in place and forms Y_f = L_f^{-1} A_fs, z_f = L_f^{-1} g_f; the
// shared block is then solved on the Schur complement S = A_ss - sum_f Y_f^T Y_f (a few panels of the blocked Cholesky instead of
// ~30), and every frame back-substitutes d_f = L_f^{-T} (z_f - Y_f d_s).
// ---------------------------------------------------------------------------------------------------------------------------
KERNEL k_s1_elim(const double* A, int n, const double* g, const int* fcols, int fs, const int* scols, int ns, int nsp,
                 double* Linv, double* Y, double* z, int* status, int fbase) {       // grid frames, dynamic LDS (fs (fs + 1) + 2 fs) doubles
    DYN_LDS(lds);
    const int f = fbase + BX, ld = fs + 1;
    double* Lm = lds; double* Bb = lds + (size_t)fs * ld;      // Bb: two columns of fs doubles
    const int* fc = fcols + (size_t)f * fs;
    for (int e = TID; e < fs * fs; e += NT) { int i = e / fs, j = e % fs; Lm[i * ld + j] = (j <= i) ? A[(size_t)fc[i] * n + fc[j]] : 0.0; }
    SYNC();
    // Cholesky, right-looking, in LDS.  ONE barrier a column (there were three: pivot, column scaling, update): every thread forms the pivot's
    // root and reciprocal for itself, the update multiplies the column's entries by it on the fly, and the column gets its final values during the
    // NEXT column's step, when nobody reads it any more.  The same operations on the same operands as before: the same bits.
    double ip_prev = 0.0, d_prev = 0.0;
    for (int c = 0; c < fs; ++c) {
        double v = Lm[c * ld + c];
        if (!(v > 0)) { if (TID == 0) status[1] = 1; v = 1.0; }
        const double dd = sqrt(v), ip = 1.0 / dd;
        if (c > 0) {
            for (int r = c + TID; r < fs; r += NT) Lm[r * ld + c - 1] *= ip_prev;
            if (TID == 0) Lm[(c - 1) * ld + c - 1] = d_prev;
        }
        for (int r = c + 1 + TID / 16; r < fs; r += NT / 16) {      // (entries (r, k <= r) dealt 16 columns wide: no index division, no idle upper half)
            const double lr = Lm[r * ld + c] * ip;
            for (int k = c + 1 + TID % 16; k <= r; k += 16) { const double lk = Lm[k * ld + c] * ip; Lm[r * ld + k] -= lr * lk; }
        }
        ip_prev = ip; d_prev = dd;
        SYNC();
    }
    if (TID == 0) Lm[(fs - 1) * ld + fs - 1] = d_prev;
    SYNC();
    // in-place inverse of the lower factor, column by column, last first: (trailing inverse) . column.  One barrier a column (there were two): a
    // step leaves its column in a buffer, from which the next step reads it while it writes it into the matrix.
    for (int j = fs - 1; j >= 0; --j) {
        double* Bc = Bb + (size_t)(j & 1) * fs;
        const double* Bn = Bb + (size_t)((j + 1) & 1) * fs;     // column j + 1 of the inverse (rows j + 1 ..), the previous step's
        const double ajj = 1.0 / Lm[j * ld + j];
        for (int i = j + 1 + TID; i < fs; i += NT) {
            double sacc = 0;
            sacc += Bn[i] * Lm[(j + 1) * ld + j];
            int k = j + 2;
            for (; k + 3 <= i; k += 4) {     // (four terms' operands in flight; the sum in k order as before)
                const double a0 = Lm[i * ld + k], a1 = Lm[i * ld + k + 1], a2 = Lm[i * ld + k + 2], a3 = Lm[i * ld + k + 3];
                const double b0 = Lm[k * ld + j], b1 = Lm[(k + 1) * ld + j], b2 = Lm[(k + 2) * ld + j], b3 = Lm[(k + 3) * ld + j];
                sacc += a0 * b0; sacc += a1 * b1; sacc += a2 * b2; sacc += a3 * b3;
            }
            for (; k <= i; ++k) sacc += Lm[i * ld + k] * Lm[k * ld + j];
            Bc[i] = -ajj * sacc;
        }
        if (TID == 0) Bc[j] = ajj;
        if (j + 1 < fs) for (int i = j + 1 + TID; i < fs; i += NT) Lm[i * ld + j + 1] = Bn[i];
        SYNC();
    }
    for (int i = TID; i < fs; i += NT) Lm[i * ld] = Bb[i];
    SYNC();
    double* Lg = Linv + (size_t)f * fs * fs;
    for (int e = TID; e < fs * fs; e += NT) { int i = e / fs, j = e % fs; Lg[e] = (j <= i) ? Lm[i * ld + j] : 0.0; }
}
// Y_f = L_f^{-1} A_fs (+ the gradient as one more column -> z_f) with the inverse factors k_s1_elim left in Linv.  grid (frames, ceil((ns + 1) / 32)):
// a workgroup stages the inverse factor and its 32 columns of A_fs in LDS (dynamic: fs (fs + 1) + 32 fs doubles); thread = (column, rows i = TID / 32
// mod NT / 32); the sums run over k ascending as they did inside k_s1_elim (where this was 44 items a thread on one workgroup per frame, operands from L2).
#define S1_YC 32
KERNEL k_s1_elim_y(const double* A, int n, const double* g, const int* fcols, int fs, const int* scols, int ns, int nsp,
                   const double* Linv, double* Y, double* z, int fbase) {
    DYN_LDS(lds);
    const int f = fbase + BX, ld = fs + 1, c0 = BY * S1_YC;
    double* Lm = lds; double* As = lds + (size_t)fs * ld;      // As[k][32]
    const int* fc = fcols + (size_t)f * fs;
    const double* Lg = Linv + (size_t)f * fs * fs;
    for (int e = TID; e < fs * fs; e += NT) { int i = e / fs, j = e % fs; Lm[i * ld + j] = Lg[e]; }
    for (int e = TID; e < fs * S1_YC; e += NT) {
        const int k = e / S1_YC, c = c0 + e % S1_YC;
        As[e] = (c < ns) ? A[(size_t)fc[k] * n + scols[c]] : (c == ns ? g[fc[k]] : 0.0);
    }
    SYNC();
    const int cl = TID % S1_YC, c = c0 + cl;
    if (c > ns) return;
    for (int i = TID / S1_YC; i < fs; i += NT / S1_YC) {
        double sacc = 0;
        for (int k = 0; k <= i; ++k) sacc += Lm[i * ld + k] * As[k * S1_YC + cl];
        if (c < ns) Y[((size_t)f * fs + i) * nsp + c] = sacc; else z[(size_t)f * fs + i] = sacc;
    }
}

// S = A_ss - T (T = Y^T Y from the tiled SYRK), h = g_s - Y^T z                                  grid ceil(ns / S1_TPB)
// grid ns: workgroup c writes row c of S along the row (one thread per row and a loop over its ns entries was 270 us of strided stores);
// the first ceil(ns / NT) workgroups also form h, one column per thread, rows in order as before
KERNEL k_s1_schur_sub(const double* A, int n, const double* g, const int* scols, int ns, const double* T, const double* Y, int nsp,
                      const double* z, int rows, double* S, double* h) {
    {
        const int c = BX;
        const size_t arow = (size_t)scols[c] * n;
        for (int k = TID; k < ns; k += NT) S[(size_t)c * ns + k] = A[arow + scols[k]] - T[(size_t)c * ns + k];
    }
    const int c = BX * NT + TID;
    if (c >= ns) return;
    double sacc = 0;
    int r = 0;
    for (; r + 8 <= rows; r += 8) {      // eight rows' operands in flight, the sum in row order as before
        double yv[8], zv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { yv[u] = Y[(size_t)(r + u) * nsp + c]; zv[u] = z[r + u]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) sacc += yv[u] * zv[u];
    }
    for (; r < rows; ++r) sacc += Y[(size_t)r * nsp + c] * z[r];
    h[c] = g[scols[c]] - sacc;
}

// d_f = L_f^{-T} (z_f - Y_f d_s), scattered into the full step; block 0 also scatters d_s                        grid F
KERNEL k_s1_back(const int* fcols, int fs, const int* scols, int ns, int nsp, const double* Linv, const double* Y, const double* z,
                 const double* ds, double* dfull, int fbase, int scatter_shared) {
    SHARED double w[S1_FSMAX];
    const int f = fbase + BX;
    for (int i = TID; i < fs; i += NT) {
        const double* Yr = Y + ((size_t)f * fs + i) * nsp;
        double sacc = z[(size_t)f * fs + i];
        for (int c = 0; c < ns; ++c) sacc -= Yr[c] * ds[c];
        w[i] = sacc;
    }
    SYNC();
    const double* Lg = Linv + (size_t)f * fs * fs;
    for (int k = TID; k < fs; k += NT) {
        double sacc = 0;
        for (int i = k; i < fs; ++i) sacc += Lg[(size_t)i * fs + k] * w[i];
        dfull[fcols[(size_t)f * fs + k]] = sacc;
    }
    if (BX == 0 && scatter_shared) for (int c = TID; c < ns; c += NT) dfull[scols[c]] = ds[c];
}

// ---------------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------------

struct DevPool {        // every device allocation of one solve; freed together
    std::vector<void*> ptrs;
    bool ok = true;
    template <class T> T* get(size_t count) {
        void* q = nullptr;
        if (hipMalloc(&q, std::max<size_t>(count, 1) * sizeof(T)) != hipSuccess) { ok = false; return nullptr; }
        ptrs.push_back(q);
        return (T*)q;
    }
    template <class T> T* put(const T* src, size_t count, hipStream_t st) {
        T* q = get<T>(count);
        if (q && count) hipMemcpyAsync(q, src, count * sizeof(T), hipMemcpyHostToDevice, st);
        return q;
    }
    ~DevPool() { for (void* q : ptrs) hipFree(q); }
};

static double nrm2(const std::vector<double>& v) { double s = 0; for (double x : v) s += x * x; return sqrt(s); }
static double dotv(const std::vector<double>& a, const std::vector<double>& b) { double s = 0; for (size_t i = 0; i < a.size(); ++i) s += a[i] * b[i]; return s; }

}  // namespace

// One Stage-I solve.  `mv` / `pv`: device views of the model / prior (moshii_api.hip fills them from the handles).
int moshii_stagei_core(const S1ModelView* mv, const S1PriorView* pv, const moshii_stagei_desc* ds, void* stream_, char* err, int errlen) {
    hipStream_t st = (hipStream_t)stream_;
    auto fail = [&](int code, const char* msg) { snprintf(err, errlen, "%s", msg); return code; };
    S1Dims d; memset(&d, 0, sizeof(d));
    S1Ptr p; memset(&p, 0, sizeof(p));
    d.F = ds->n_frames; d.M = ds->M; d.NPZ = d.F + 1;
    d.per_frame = ds->n_expr > 0 ? 1 : 0;
    d.nb = d.per_frame ? ds->n_expr : ds->nb; d.shape_start = d.per_frame ? ds->expr_start : 0;
    if (d.per_frame && ds->nb != 0) return fail(MOSHII_ERR_ARG, "stagei: free expressions need fixed betas (nb = 0; chmosh.py:295-299)");
    d.V = mv->V; d.K = mv->K; d.P = 3 * mv->K; d.NP = mv->NP; d.body_dof = mv->body_dof; d.hand_dof = mv->hand_dof;
    d.nhand_full = d.P - d.body_dof; d.NBtot = mv->NB; d.nfeat = 9 * (mv->K - 1);
    d.nfaces = ds->n_faces; d.nbody = ds->n_body; d.G = pv ? pv->G : 0; d.npose_prior = pv ? pv->npose : 0;
    d.ncan = 9 * d.M;
    if (d.F < 1 || d.M < 3 || d.nb < 0 || d.shape_start < 0 || d.shape_start + d.nb > d.NBtot) return fail(MOSHII_ERR_ARG, "stagei: bad sizes");
    if (pv && d.nbody != pv->npose) return fail(MOSHII_ERR_ARG, "stagei: prior size != number of body pose ids");
    if (d.G > 64) return fail(MOSHII_ERR_ARG, "stagei: more than 64 mixture components");
    const int F = d.F, M = d.M, nb = d.nb, NP = d.NP, K = d.K;
    int ntot_obs = 0;
    std::vector<int> obs_off(F + 1, 0);
    for (int f = 0; f < F; ++f) {
        if (ds->n_obs[f] < 3) return fail(MOSHII_ERR_ARG, "stagei: a picked frame has fewer than 3 observed layout markers (no rigid start possible)");
        ntot_obs += ds->n_obs[f]; obs_off[f + 1] = ntot_obs;
    }
    for (int i = 0; i < ntot_obs; ++i) if (ds->obs_ids[i] < 0 || ds->obs_ids[i] >= M) return fail(MOSHII_ERR_ARG, "stagei: observed marker id out of range");

    // frames of this rank (moshii_stagei_desc.sharded): every rank evaluates the canonical body and the attachment, the data / prior /
    // finger rows of its own frames, one rank the shared rows; normal equations and SSE are summed over the ranks through the
    // caller's all-reduce, after which every rank holds the same system and takes the same step
    const bool shard = ds->sharded && ds->allreduce_sum;
    const int f_lo = shard ? ds->frame_lo : 0, f_hi = shard ? ds->frame_hi : F, nown = f_hi - f_lo;
    const int own_shared = shard ? (ds->owns_shared_rows ? 1 : 0) : 1;
    // (a rank whose arguments are wrong must not just return: the others would wait in the first all-reduce.  Its verdict is summed
    //  over the ranks below, as soon as reduce() exists, and every rank fails together.)
    const bool bad_range = shard && (f_lo < 0 || f_hi > F || nown < 0);
    // the arrow solver back-substitutes the shared block on a rank with frames (the dense solver, MOSHII_S1_SOLVER=dense, has no such need)
    const char* solver_env0 = getenv("MOSHII_S1_SOLVER");
    const bool bad_owner = shard && own_shared && nown <= 0 && !(solver_env0 && strcmp(solver_env0, "dense") == 0);
    int reduce_rc = 0;
    DevPool pool;
    // sums over the ranks.  reduce(): a host vector; reduce_dev(): one of the solver's device buffers.  With allreduce_on_device the
    // callback works on device memory (RCCL): device buffers are summed in place, host vectors are staged through d_red; without it
    // the callback works on host memory (gloo, or nccl behind a copy) and device buffers take the round trip through `hred`.
    const bool red_dev = shard && ds->allreduce_on_device;
    double* d_red = nullptr; long long d_red_cap = 0;
    std::vector<double> hred;
    auto reduce = [&](double* buf, long long count) {
        if (!shard || count <= 0) return;
        if (!red_dev) { if (ds->allreduce_sum(buf, count, ds->allreduce_user) != 0) reduce_rc = 1; return; }
        if (count > d_red_cap) { d_red = pool.get<double>((size_t)count); d_red_cap = count; if (!pool.ok) { reduce_rc = 1; return; } }
        hipMemcpyAsync(d_red, buf, (size_t)count * 8, hipMemcpyHostToDevice, st);
        hipStreamSynchronize(st);
        if (ds->allreduce_sum(d_red, count, ds->allreduce_user) != 0) reduce_rc = 1;
        hipMemcpyAsync(buf, d_red, (size_t)count * 8, hipMemcpyDeviceToHost, st);
        hipStreamSynchronize(st);
    };
    auto reduce_dev = [&](double* dbuf, long long count) {
        if (!shard || count <= 0) return;
        hipStreamSynchronize(st);
        if (red_dev) { if (ds->allreduce_sum(dbuf, count, ds->allreduce_user) != 0) reduce_rc = 1; return; }
        hred.resize((size_t)count);
        hipMemcpyAsync(hred.data(), dbuf, (size_t)count * 8, hipMemcpyDeviceToHost, st);
        hipStreamSynchronize(st);
        if (ds->allreduce_sum(hred.data(), count, ds->allreduce_user) != 0) reduce_rc = 1;
        hipMemcpyAsync(dbuf, hred.data(), (size_t)count * 8, hipMemcpyHostToDevice, st);
        hipStreamSynchronize(st);
    };
    if (shard) {
        // The device path stages host vectors through d_red: that buffer is sized HERE, before the first collective, for the largest
        // vector the solve will ever stage (the n unknowns; 3 F M marker coordinates), so that no rank can drop out of a LATER all-reduce
        // because of it.  A rank that does not get it still joins the verdict sum -- through a 32-byte buffer -- and reports the failure
        // there, so that ALL ranks fail together instead of the others waiting in their first all-reduce for the process group's
        // timeout; only a rank that cannot even get those 32 bytes fails alone.
        double verdict[3] = {bad_range ? 1.0 : 0.0, bad_owner ? 1.0 : 0.0, 0.0};
        if (red_dev) {
            const long long cap0 = 3LL * M + (long long)F * (3 + NP) + (long long)nb * (d.per_frame ? F : 1) + 3LL * F * M + 6LL * F + 64;   // >= n, 3 F M, 6 F
            d_red = pool.get<double>((size_t)cap0);
            d_red_cap = pool.ok ? cap0 : 0;
            if (!pool.ok) {
                verdict[2] = 1.0;
                double* tiny = nullptr;
                if (hipMalloc((void**)&tiny, 4 * sizeof(double)) != hipSuccess)
                    return fail(MOSHII_ERR_HIP, "stagei: device allocation failed before the first all-reduce (the other ranks were not joined)");
                d_red = tiny; d_red_cap = 4;
                reduce(verdict, 3);
                hipFree(tiny);
                return fail(MOSHII_ERR_HIP, "stagei: device allocation of the all-reduce staging buffer failed on this rank (all ranks were told)");
            }
        }
        reduce(verdict, 3);
        if (reduce_rc) return fail(MOSHII_ERR_ARG, "stagei: the all-reduce callback failed");
        if (verdict[2] > 0) return fail(MOSHII_ERR_HIP, "stagei: device allocation of the all-reduce staging buffer failed on another rank");
        if (verdict[0] > 0) return fail(MOSHII_ERR_ARG, "stagei: frame range of a rank out of bounds");
        if (verdict[1] > 0) return fail(MOSHII_ERR_ARG, "stagei: the rank that owns the shared rows must own at least one frame");
    }
    // ---- constants
    p.parents = mv->parents; p.vt = mv->vt; p.shapedirs = mv->shapedirs; p.posedirs = mv->posedirs; p.weights = mv->weights;
    p.Jreg = mv->Jreg; p.hands_mean = mv->hands_mean; p.comps = mv->comps; p.anc = mv->anc;
    if (pv) { p.means = pv->means; p.chols = pv->chols; p.neglogw = pv->neglogw; }
    p.faces = pool.put(ds->faces, (size_t)3 * d.nfaces, st);
    {   // vertex -> incident faces
        std::vector<int> ptr(d.V + 1, 0), lst((size_t)3 * d.nfaces);
        for (int i = 0; i < 3 * d.nfaces; ++i) {
            if (ds->faces[i] < 0 || ds->faces[i] >= d.V) return fail(MOSHII_ERR_ARG, "stagei: face index out of range");
            ptr[ds->faces[i] + 1]++;
        }
        for (int v = 0; v < d.V; ++v) ptr[v + 1] += ptr[v];
        std::vector<int> fill(ptr.begin(), ptr.end() - 1);
        for (int f = 0; f < d.nfaces; ++f) for (int c = 0; c < 3; ++c) lst[fill[ds->faces[3 * f + c]]++] = f;
        p.v2f_ptr = pool.put(ptr.data(), ptr.size(), st); p.v2f = pool.put(lst.data(), lst.size(), st);
        hipStreamSynchronize(st);
    }
    {
        std::vector<unsigned char> ex(d.V, 0);
        for (int i = 0; i < ds->n_exclude; ++i) if (ds->exclude_vids[i] >= 0 && ds->exclude_vids[i] < d.V) ex[ds->exclude_vids[i]] = 1;
        p.excl = pool.put(ex.data(), ex.size(), st);
        hipStreamSynchronize(st);
    }
    p.obs_ids = pool.put(ds->obs_ids, ntot_obs, st); p.obs_off = pool.put(obs_off.data(), obs_off.size(), st);
    p.obs = pool.put(ds->obs, (size_t)3 * ntot_obs, st); p.m2b = pool.put(ds->m2b, M, st);
    p.body_ids = pool.put(ds->body_ids, d.nbody, st);
    int* d_colmap = pool.get<int>(NP); p.colmap = d_colmap;
    int* d_finger = pool.get<int>(std::max(1, ds->n_finger)); p.finger_ids = d_finger;
    const int nface_all = d.per_frame ? ds->n_face : 0;
    for (int k = 0; k < nface_all; ++k) if (ds->face_ids[k] < 0 || ds->face_ids[k] >= NP) return fail(MOSHII_ERR_ARG, "stagei: face pose id out of range");
    p.face_ids = pool.put(ds->face_ids, nface_all, st);
    p.cl0 = pool.get<int>(3 * M); p.coef0 = pool.get<double>(3 * M);
    p.loss = pool.get<double>(3 * M); p.dinit = pool.get<double>((size_t)3 * M * std::max(1, nb));
    d.nhead = ds->head_ids ? ds->n_head : 0; d.nhead_rows = ds->head_ids ? ds->n_head_rows : 0;
    std::vector<double> wt_init_eff(ds->wt_init, ds->wt_init + M);
    if (d.nhead_rows) {
        if (!ds->head_corr) return fail(MOSHII_ERR_ARG, "stagei: head_ids without head_corr");
        for (int h = 0; h < d.nhead; ++h) {
            if (ds->head_ids[h] < 0 || ds->head_ids[h] >= M) return fail(MOSHII_ERR_ARG, "stagei: head marker id out of range");
            wt_init_eff[ds->head_ids[h]] = 0.0;          // the head markers leave their per-type init rows (chmosh.py:364-367)
        }
        p.head_ids = pool.put(ds->head_ids, d.nhead, st); p.head_C = pool.put(ds->head_corr, (size_t)d.nhead_rows * d.nhead, st);
    }
    p.wt_init = pool.put(wt_init_eff.data(), M, st);
    hipStreamSynchronize(st);
    // the evaluation point in ONE allocation (pose | trans | latent markers | betas, each part on a 128-byte boundary): one copy per trial point
    // instead of four (565 small copies a solve were ~3 ms of host time)
    const size_t pt_pose = (size_t)d.NPZ * NP, pt_trans = (size_t)d.NPZ * 3, pt_ml = (size_t)3 * M, pt_betas = (size_t)std::max(1, nb) * (d.per_frame ? d.NPZ : 1);
    auto up16 = [](size_t c) { return (c + 15) & ~(size_t)15; };
    const size_t pt_o_trans = up16(pt_pose), pt_o_ml = pt_o_trans + up16(pt_trans), pt_o_betas = pt_o_ml + up16(pt_ml), pt_total = pt_o_betas + up16(pt_betas);
    double* d_point = pool.get<double>(pt_total);
    p.pose = d_point; p.trans = d_point ? d_point + pt_o_trans : nullptr;
    p.ml = d_point ? d_point + pt_o_ml : nullptr; p.betas = d_point ? d_point + pt_o_betas : nullptr;
    p.J0 = pool.get<double>(3 * K); p.JS = pool.get<double>((size_t)std::max(1, K * nb * 3));
    p.fp = pool.get<double>((size_t)d.NPZ * d.P); p.Rl = pool.get<double>((size_t)d.NPZ * K * 9); p.Jl = pool.get<double>((size_t)d.NPZ * K * 9);
    p.Rw = pool.get<double>((size_t)d.NPZ * K * 9); p.tw = pool.get<double>((size_t)d.NPZ * K * 3); p.feat = pool.get<double>((size_t)d.NPZ * d.nfeat);
    p.om = pool.get<double>((size_t)d.NPZ * K * 9); p.Bm = pool.get<double>((size_t)d.NPZ * K * 27); p.Jb = pool.get<double>((size_t)d.NPZ * K * 3);
    p.q = pool.get<double>((size_t)d.NPZ * K * std::max(1, nb) * 3); p.featzero = pool.get<int>(d.NPZ);
    p.can = pool.get<double>((size_t)3 * d.V); p.cl = pool.get<int>(3 * M); p.cl8 = pool.get<int>(S1_NNK * M); p.coef = pool.get<double>(3 * M); p.Fc = pool.get<double>(9 * M);
    p.dcdb = pool.get<double>((size_t)M * 3 * std::max(1, nb)); p.tv = pool.get<int>(3 * M); p.sdist = pool.get<double>(M);
    p.sdp = pool.get<double>(3 * M); p.sdabc = pool.get<double>(9 * M); p.status = pool.get<int>(4);
    p.vlist = pool.get<int>((size_t)d.NPZ * d.ncan); p.vv = pool.get<double>((size_t)d.NPZ * d.ncan * 3);
    p.dvs = pool.get<double>((size_t)d.NPZ * d.ncan * 3 * std::max(1, nb)); p.dv = pool.get<double>((size_t)F * 3 * M * 3 * d.P);
    p.vi_T = pool.get<double>((size_t)F * 3 * M * 9); p.vi_w = pool.get<double>((size_t)F * 3 * M * S1_NWMAX);
    p.vi_x = pool.get<double>((size_t)F * 3 * M * S1_NWMAX * 3); p.vi_n = pool.get<int>((size_t)F * 3 * M); p.vi_j = pool.get<int>((size_t)F * 3 * M * S1_NWMAX);
    p.Lb = pool.get<double>((size_t)F * M * 36 + (size_t)F * std::max(1, d.G * d.npose_prior));
    // largest problem of the rounds: all of body + fingers free
    const int npid_max = ds->n_pose_ids + ds->n_finger + nface_all;
    const int nsh_max = d.per_frame ? F * nb : nb;
    const int n_max = 3 * F + 3 * M + F * npid_max + nsh_max;
    const int R_max = 3 * ntot_obs + F * (d.G ? d.npose_prior + 1 : 0) + 3 * M + nsh_max + M + F * ds->n_finger + 3 * d.nhead_rows + F * nface_all;
    const int ld_max = (n_max + 15) & ~15;
    p.r = pool.get<double>(R_max); p.Jm = pool.get<double>((size_t)R_max * ld_max);
    if (n_max > S1_NMAX) return fail(MOSHII_ERR_ARG, "stagei: more than 4096 unknowns");
    double* d_A = pool.get<double>((size_t)n_max * n_max); double* d_L = pool.get<double>((size_t)n_max * n_max);
    // arrow-structured solver (per-frame elimination + Schur complement on the shared block) unless MOSHII_S1_SOLVER=dense asks for
    // the dense blocked Cholesky of the whole system.  Default since round 2: ten seeded problems (six SMPL-H seeds, fingers, SMPL-X,
    // SMPL, fixed betas) take the same dogleg iterations with both and end within 1.3e-10 (profiles/r02_stagei_schur_seeds.txt), the
    // arrow form 1.35-1.5x faster; sharded, it sums the 169 x 169 shared block instead of the 925 x 925 system.
    const char* solver_env = getenv("MOSHII_S1_SOLVER");
    const bool want_schur = !(solver_env && strcmp(solver_env, "dense") == 0);
    const int fs_max = 3 + npid_max + (d.per_frame ? nb : 0), ns_max = 3 * M + (d.per_frame ? 0 : nb), nsp_max = (ns_max + 15) & ~15;
    int *d_fcols = nullptr, *d_scols = nullptr, *d_ones = nullptr;
    double *d_Linv = nullptr, *d_Y = nullptr, *d_z = nullptr, *d_T = nullptr, *d_S = nullptr, *d_h = nullptr, *d_ds = nullptr;
    if (want_schur) {
        d_fcols = pool.get<int>((size_t)F * fs_max); d_scols = pool.get<int>(ns_max);
        d_Linv = pool.get<double>((size_t)F * fs_max * fs_max); d_Y = pool.get<double>((size_t)F * fs_max * nsp_max); d_z = pool.get<double>((size_t)F * fs_max);
        d_T = pool.get<double>((size_t)ns_max * ns_max); d_S = pool.get<double>((size_t)ns_max * ns_max); d_h = pool.get<double>(ns_max); d_ds = pool.get<double>(ns_max);
        d_ones = pool.get<int>((size_t)((F * fs_max + S1_T - 1) / S1_T) * ((ns_max + S1_T - 1) / S1_T));
    }
    double* d_dinv = pool.get<double>((size_t)(n_max / S1_PB + 1) * S1_PB * S1_PB); int* d_flags = pool.get<int>((size_t)((R_max + S1_T - 1) / S1_T) * ((n_max + S1_T - 1) / S1_T));
    double* d_g = pool.get<double>(n_max); double* d_part = pool.get<double>((size_t)S1_GT_CHUNKS * n_max); double* d_vec = pool.get<double>(n_max);
    double* d_out = pool.get<double>(std::max(n_max, R_max));
    if (!pool.ok) return fail(MOSHII_ERR_HIP, "stagei: device allocation failed");
    hipMemsetAsync(p.status, 0, 4 * sizeof(int), st);

    // ---- host state
    std::vector<double> pose((size_t)d.NPZ * NP, 0.0), trans((size_t)d.NPZ * 3, 0.0), ml(3 * M), betas((size_t)std::max(1, nb) * (d.per_frame ? d.NPZ : 1), 0.0);
    if (ds->betas_init && !d.per_frame) for (int e = 0; e < nb; ++e) betas[e] = ds->betas_init[e];
    std::vector<int> pose_ids, finger_ids, colmap(NP, -1);

    std::vector<double> hpoint(pt_total, 0.0);
    auto upload_point = [&]() {
        memcpy(hpoint.data(), pose.data(), std::min(pose.size(), pt_pose) * 8);
        memcpy(hpoint.data() + pt_o_trans, trans.data(), std::min(trans.size(), pt_trans) * 8);
        memcpy(hpoint.data() + pt_o_ml, ml.data(), std::min(ml.size(), pt_ml) * 8);
        memcpy(hpoint.data() + pt_o_betas, betas.data(), std::min(betas.size(), pt_betas) * 8);
        hipMemcpyAsync(d_point, hpoint.data(), pt_total * 8, hipMemcpyHostToDevice, st);
    };
    auto canonical = [&]() {   // pose kernels + canonical mesh at the uploaded point
        LAUNCH(k_s1_pose, d.NPZ, 1, S1_TPB, st, d, p);
        LAUNCH(k_s1_verts, (d.V + S1_TPB - 1) / S1_TPB, 1, S1_TPB, st, d, p, F, d.V, 0, p.can, 1, 1);
    };
    // evaluation of residual (and Jacobian) at the uploaded point
    int shared_rows_on = 1;   // 0 during the extra rigid adjustment: the init / beta / surf / head rows are not part of its objective
    auto evaluate = [&](int want_J) {
        canonical();
        LAUNCH(k_s1_knn, M, 1, S1_TPB, st, d, p, p.cl8);
        LAUNCH(k_s1_pick3, 1, 1, S1_TPB, st, d, p, p.cl8, p.cl);
        LAUNCH(k_s1_surface, M, 1, S1_SURF_TPB, st, d, p);
        LAUNCH(k_s1_lists, (3 * M + S1_TPB - 1) / S1_TPB, 1, S1_TPB, st, d, p);
        LAUNCH(k_s1_verts, (d.ncan * S1_VL + S1_TPB - 1) / S1_TPB, 1, S1_TPB, st, d, p, F, d.ncan, want_J ? 2 : 1, (double*)nullptr, 0, S1_VL);
        if (nown > 0) LAUNCH(k_s1_verts, (3 * M * S1_VL + S1_TPB - 1) / S1_TPB, nown, S1_TPB, st, d, p, f_lo, 3 * M, want_J ? 3 : 1, (double*)nullptr, 0, S1_VL);
        if (want_J && nown > 0) LAUNCH(k_s1_vjac, (3 * M * K + S1_TPB - 1) / S1_TPB, nown, S1_TPB, st, d, p, f_lo);
        if (want_J) hipMemsetAsync(p.Jm, 0, (size_t)d.R * d.ldn * 8, st);
        if (shard || !shared_rows_on) hipMemsetAsync(p.r, 0, (size_t)d.R * 8, st);          // rows of frames other ranks own (or switched off) stay zero here
        LAUNCH(k_s1_shared, (M + S1_TPB - 1) / S1_TPB, 1, S1_TPB, st, d, p, want_J, own_shared && shared_rows_on);
        if (d.nhead_rows && own_shared && shared_rows_on) LAUNCH(k_s1_head, 1, 1, S1_TPB, st, d, p, want_J);
        if (nown > 0) LAUNCH(k_s1_rows, nown, want_J ? 4 : 1, S1_TPB, st, d, p, want_J, f_lo);
    };
    auto fetch = [&](std::vector<double>& h, const double* dev, size_t count) {
        h.resize(count);
        hipMemcpyAsync(h.data(), dev, count * 8, hipMemcpyDeviceToHost, st);
        hipStreamSynchronize(st);
    };

    // ---- setup: regressed joints, initial latent markers (vertex + normal . m2b), frozen init attachment, rigid start
    LAUNCH(k_s1_setup, K, nb + 1, S1_TPB, st, d, p);
    upload_point();
    canonical();
    std::vector<double> can;
    fetch(can, p.can, (size_t)3 * d.V);
    {   // prepare_mosh_markers_latent (chmosh.py:57-67); normals from the incident faces
        for (int m = 0; m < M; ++m) {
            int v = ds->marker_vids[m];
            if (v < 0 || v >= d.V) return fail(MOSHII_ERR_ARG, "stagei: marker vertex id out of range");
            double nrm[3] = {0, 0, 0};
            for (int f = 0; f < d.nfaces; ++f) {
                const int* fv = ds->faces + 3 * f;
                if (fv[0] != v && fv[1] != v && fv[2] != v) continue;
                double e1[3], e2[3];
                for (int a = 0; a < 3; ++a) { e1[a] = can[3 * fv[1] + a] - can[3 * fv[0] + a]; e2[a] = can[3 * fv[2] + a] - can[3 * fv[0] + a]; }
                nrm[0] += e1[1] * e2[2] - e1[2] * e2[1]; nrm[1] += e1[2] * e2[0] - e1[0] * e2[2]; nrm[2] += e1[0] * e2[1] - e1[1] * e2[0];
            }
            double ss = nrm[0] * nrm[0] + nrm[1] * nrm[1] + nrm[2] * nrm[2];
            if (ss == 0) ss = 1e-10;
            for (int a = 0; a < 3; ++a) ml[3 * m + a] = can[3 * v + a] + nrm[a] / sqrt(ss) * ds->m2b[m];
        }
    }
    upload_point();
    // frozen attachment of the init markers (chmosh.py:188-190): closest0 / coef0 at the start point
    d.shape_free = d.per_frame ? 0 : 1;
    const int nsh0 = d.per_frame ? 0 : nb;
    d.npid = 0; d.n = 3 * F + 3 * M + nsh0; d.ldn = (d.n + 15) & ~15; d.o_ml = 3 * F; d.o_pose = 3 * F + 3 * M; d.o_b = d.o_pose;
    d.r_data = 0; d.r_prior = 3 * ntot_obs; d.r_init = d.r_prior; d.r_beta = d.r_init + 3 * M; d.r_surf = d.r_beta + nsh0; d.r_poseH = d.r_surf + M; d.r_head = d.r_poseH; d.r_poseF = d.r_head + 3 * d.nhead_rows; d.R = d.r_poseF;
    d.nfinger = 0; d.nface = 0;
    {
        int Gkeep = d.G; d.G = 0;                      // no prior rows while the column map is empty
        hipMemcpyAsync(d_colmap, colmap.data(), NP * sizeof(int), hipMemcpyHostToDevice, st);
        LAUNCH(k_s1_knn, M, 1, S1_TPB, st, d, p, p.cl8);
        LAUNCH(k_s1_pick3, 1, 1, S1_TPB, st, d, p, p.cl8, p.cl0);
        evaluate(0);                                    // fills coef (== coef0 here) and the posed markers of every frame
        hipMemcpyAsync(p.coef0, p.coef, 3 * M * 8, hipMemcpyDeviceToDevice, st);
        d.G = Gkeep;
    }
    {   // rigid start per frame (chmosh.py:236-238, rigid_transformations.py:39-83): Procrustes on the zero-pose markers
        double* d_sim = pool.get<double>((size_t)3 * std::max(1, ntot_obs)); double* d_rt = pool.get<double>(6 * F);
        if (!pool.ok) return fail(MOSHII_ERR_HIP, "stagei: device allocation failed");
        hipMemsetAsync(d_rt, 0, 6 * F * 8, st);
        if (nown > 0) LAUNCH(k_s1_rigid, nown, 1, S1_TPB, st, d, p, d_sim, d_rt, f_lo);
        std::vector<double> rt;
        fetch(rt, d_rt, 6 * F);
        reduce(rt.data(), 6 * F);
        for (int f = 0; f < F; ++f) for (int a = 0; a < 3; ++a) { pose[(size_t)f * NP + a] = rt[6 * f + a]; trans[3 * f + a] = rt[6 * f + 3 + a]; }
    }

    // ---- annealing rounds
    int total_iters = 0;
    std::vector<double> r, rnew, g, x, dsd, dgn, ddl, tmp, hbuf;
    // round -1 (opt_settings.extra_initial_rigid_adjustment, chmosh.py:230-232): the unweighted marker residuals of all frames
    // as a function of every frame's root orientation and translation only, e_3 = .001.  It runs through the same machinery:
    // the only pose columns are the root's, the latent-marker and shape columns are parked behind column n (written, never
    // read: the products take n columns of a row of ldn), the prior / init / beta / surf / head rows are off.
    const int Gall = d.G;
    for (int round = ds->extra_initial_rigid_adjustment ? -1 : 0; round < ds->n_anneal; ++round) {
        const bool rigid_round = round < 0;
        const double a = rigid_round ? 1.0 : ds->annealing[round];
        const bool detailed = !rigid_round && round > ds->n_anneal - 3;
        shared_rows_on = rigid_round ? 0 : 1;
        d.G = rigid_round ? 0 : Gall;
        pose_ids.assign(ds->pose_ids, ds->pose_ids + ds->n_pose_ids);
        if (rigid_round) { pose_ids.clear(); pose_ids.push_back(0); pose_ids.push_back(1); pose_ids.push_back(2); }
        finger_ids.clear();
        if (detailed) finger_ids.assign(ds->finger_ids, ds->finger_ids + ds->n_finger);
        pose_ids.insert(pose_ids.end(), finger_ids.begin(), finger_ids.end());
        d.nface = (detailed && d.per_frame) ? nface_all : 0;
        if (d.nface) pose_ids.insert(pose_ids.end(), ds->face_ids, ds->face_ids + d.nface);
        d.shape_free = d.per_frame ? (detailed ? 1 : 0) : 1;
        const int nsh = d.per_frame ? (d.shape_free ? F * nb : 0) : nb;
        std::sort(pose_ids.begin(), pose_ids.end());
        pose_ids.erase(std::unique(pose_ids.begin(), pose_ids.end()), pose_ids.end());
        std::fill(colmap.begin(), colmap.end(), -1);
        for (size_t c = 0; c < pose_ids.size(); ++c) colmap[pose_ids[c]] = (int)c;
        for (int b = 0; b < d.nbody; ++b) if (ds->body_ids[b] < 0 || ds->body_ids[b] >= NP) return fail(MOSHII_ERR_ARG, "stagei: body id out of range");
        d.npid = (int)pose_ids.size(); d.nfinger = (int)finger_ids.size();
        d.n = 3 * F + 3 * M + F * d.npid + nsh; d.ldn = (d.n + 15) & ~15;
        d.o_ml = 3 * F; d.o_pose = 3 * F + 3 * M; d.o_b = d.o_pose + F * d.npid;
        if (rigid_round) { d.n = 6 * F; d.ldn = ld_max; d.o_pose = 3 * F; d.o_ml = 6 * F; d.o_b = 6 * F + 3 * M; }
        d.r_data = 0; d.r_prior = 3 * ntot_obs; d.r_init = d.r_prior + F * (d.G ? d.npose_prior + 1 : 0);
        d.r_beta = d.r_init + 3 * M; d.r_surf = d.r_beta + nsh; d.r_poseH = d.r_surf + M; d.r_head = d.r_poseH + F * d.nfinger;
        d.r_poseF = d.r_head + 3 * d.nhead_rows; d.R = d.r_poseF + F * d.nface;
        p.w_anneal = a; p.w_data = rigid_round ? 1.0 : (ds->wt_data / a) * (46.0 / M); p.w_poseB = ds->wt_poseB * a; p.w_poseH = ds->wt_poseH * a;
        p.w_beta = (d.per_frame ? ds->wt_expr : ds->wt_betas) * a; p.w_surf = ds->wt_surf; p.w_init_head = ds->wt_init_head * a;
        p.w_poseF = ds->wt_poseF * a;
        hipMemcpyAsync(d_colmap, colmap.data(), NP * sizeof(int), hipMemcpyHostToDevice, st);
        if (d.nfinger) hipMemcpyAsync(d_finger, finger_ids.data(), d.nfinger * sizeof(int), hipMemcpyHostToDevice, st);
        const int fs = 3 + d.npid + ((d.per_frame && d.shape_free) ? nb : 0), ns = 3 * M + (d.per_frame ? 0 : nb), nsp = (ns + 15) & ~15;
        const bool schur = want_schur && fs <= S1_FSMAX && !rigid_round;
        if (schur) {
            std::vector<int> fcols((size_t)F * fs), scols(ns), ones((size_t)((F * fs + S1_T - 1) / S1_T) * ((ns + S1_T - 1) / S1_T), 1);
            for (int f = 0; f < F; ++f) {
                int* fc = fcols.data() + (size_t)f * fs;
                for (int c = 0; c < 3; ++c) fc[c] = 3 * f + c;
                for (int c = 0; c < d.npid; ++c) fc[3 + c] = d.o_pose + f * d.npid + c;
                for (int e = 0; e < fs - 3 - d.npid; ++e) fc[3 + d.npid + e] = d.o_b + f * nb + e;
            }
            for (int i = 0; i < 3 * M; ++i) scols[i] = d.o_ml + i;
            for (int e = 0; e < ns - 3 * M; ++e) scols[3 * M + e] = d.o_b + e;
            hipMemcpyAsync(d_fcols, fcols.data(), fcols.size() * sizeof(int), hipMemcpyHostToDevice, st);
            hipMemcpyAsync(d_scols, scols.data(), scols.size() * sizeof(int), hipMemcpyHostToDevice, st);
            hipMemcpyAsync(d_ones, ones.data(), ones.size() * sizeof(int), hipMemcpyHostToDevice, st);
        }
        hipStreamSynchronize(st);
        const int n = d.n, R = d.R;
        auto pack = [&](std::vector<double>& xx) {
            xx.resize(n);
            for (int f = 0; f < F; ++f) for (int c = 0; c < 3; ++c) xx[3 * f + c] = trans[3 * f + c];
            if (!rigid_round) for (int i = 0; i < 3 * M; ++i) xx[d.o_ml + i] = ml[i];
            for (int f = 0; f < F; ++f) for (int c = 0; c < d.npid; ++c) xx[d.o_pose + f * d.npid + c] = pose[(size_t)f * NP + pose_ids[c]];
            if (!rigid_round) for (int e = 0; e < nsh; ++e) xx[d.o_b + e] = betas[e];          // per-frame mode: [frame][coefficient], canonical row last
        };
        auto unpack = [&](const std::vector<double>& xx) {
            for (int f = 0; f < F; ++f) for (int c = 0; c < 3; ++c) trans[3 * f + c] = xx[3 * f + c];
            if (!rigid_round) for (int i = 0; i < 3 * M; ++i) ml[i] = xx[d.o_ml + i];
            for (int f = 0; f < F; ++f) for (int c = 0; c < d.npid; ++c) pose[(size_t)f * NP + pose_ids[c]] = xx[d.o_pose + f * d.npid + c];
            if (!rigid_round) for (int e = 0; e < nsh; ++e) betas[e] = xx[d.o_b + e];
        };
        auto eval_at = [&](const std::vector<double>& xx, int want_J, std::vector<double>& rr) -> double {
            unpack(xx); upload_point(); evaluate(want_J); fetch(rr, p.r, R);
            double sse_ = dotv(rr, rr);
            reduce(&sse_, 1);
            return sse_;
        };
        auto normal_eq = [&]() {     // A = J^T J, g = -J^T r on the device; g to the host
            int nt = (n + S1_T - 1) / S1_T;
            LAUNCH(k_s1_nzflags, (R + S1_T - 1) / S1_T, nt, 256, st, p.Jm, R, n, d.ldn, d_flags);
            LAUNCH(k_s1_syrk, nt, nt, 256, st, p.Jm, R, n, d.ldn, d_A, d_flags);
            LAUNCH(k_s1_gemv_t, (n + S1_TPB - 1) / S1_TPB, S1_GT_CHUNKS, S1_TPB, st, p.Jm, p.r, R, n, d.ldn, d_part);
            LAUNCH(k_s1_gemv_t_sum, (n + S1_TPB - 1) / S1_TPB, 1, S1_TPB, st, d_part, n, -1.0, d_g);
            if (shard && schur) {   // arrow-structured solver: A stays rank-local; only the gradient (and later the Schur block) is summed
                fetch(g, d_g, n);
                reduce(g.data(), n);
                return;
            }
            if (shard) {        // sum the ranks' normal equations A and g (MOSHII_S1_SOLVER=dense: a few MB per iteration)
                reduce_dev(d_A, (long long)n * n);
                reduce_dev(d_g, n);
            }
            fetch(g, d_g, n);
        };
        auto Ax = [&](const std::vector<double>& v, std::vector<double>& out) {   // A . v (A must still be unfactored)
            hipMemcpyAsync(d_vec, v.data(), n * 8, hipMemcpyHostToDevice, st);
            LAUNCH(k_s1_gemv, n, 1, S1_TPB, st, d_A, d_vec, n, n, d_out);
            fetch(out, d_out, n);
            if (shard && schur) reduce(out.data(), n);          // A is rank-local there: sum the products
        };
        // ---- Powell dogleg (chumpy minimize_dogleg as restated in oracle/stageii_oracle.py:minimize_dogleg)
        const double e1 = 1e-15, e2 = 1e-15, e3 = rigid_round ? .001 : ds->stagei_lr;
        pack(x);
        double sse = eval_at(x, 1, r);
        normal_eq();
        double delta = 0.5;
        bool done = false;
        int iteration = 0;
        auto norminf = [](const std::vector<double>& v) { double s = 0; for (double t : v) s = std::max(s, fabs(t)); return s; };
        if (norminf(g) < e1) done = true;
        std::vector<double> Ag, xt;
        while (!done) {
            ++iteration;
            // |J g|^2 = g^T A g
            Ax(g, Ag);
            double gg = dotv(g, g), gAg = dotv(g, Ag);
            dsd = g;
            for (double& t : dsd) t *= gg / gAg;
            bool have_gn = false;
            while (true) {
                double nsd = nrm2(dsd);
                if (nsd >= delta) { ddl = dsd; for (double& t : ddl) t *= delta / nsd; }
                else {
                    if (!have_gn && schur) {
                        const size_t lds_bytes = ((size_t)fs * (fs + 1) + 2 * (size_t)fs) * sizeof(double);
                        hipMemsetAsync(d_Y, 0, (size_t)F * fs * nsp * 8, st);
                        hipMemsetAsync(d_z, 0, (size_t)F * fs * 8, st);
                        hipFuncSetAttribute(reinterpret_cast<const void*>(k_s1_elim), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
                        // (sharded: every rank eliminates its own frames -- their rows live only here, so A_ff, A_fs and g_f are complete --
                        //  and contributes A_ss,r - Y_r^T Y_r and g_s,r - Y_r^T z_r; the sum over ranks is the Schur system: ns^2 + ns doubles)
                        if (nown > 0) LAUNCH_LDS(k_s1_elim, nown, 1, S1_TPB, lds_bytes, st, d_A, n, d_g, d_fcols, fs, d_scols, ns, nsp, d_Linv, d_Y, d_z, p.status, f_lo);
                        if (nown > 0) {
                            const size_t ybytes = ((size_t)fs * (fs + 1) + (size_t)S1_YC * fs) * sizeof(double);
                            hipFuncSetAttribute(reinterpret_cast<const void*>(k_s1_elim_y), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ybytes);
                            LAUNCH_LDS(k_s1_elim_y, nown, (ns + 1 + S1_YC - 1) / S1_YC, S1_TPB, ybytes, st, d_A, n, d_g, d_fcols, fs, d_scols, ns, nsp, d_Linv, d_Y, d_z, f_lo);
                        }
                        { int nt = (ns + S1_T - 1) / S1_T; LAUNCH(k_s1_syrk, nt, nt, 256, st, d_Y, F * fs, ns, nsp, d_T, d_ones); }
                        LAUNCH(k_s1_schur_sub, ns, 1, S1_TPB, st, d_A, n, d_g, d_scols, ns, d_T, d_Y, nsp, d_z, F * fs, d_S, d_h);
                        if (shard) {   // the only matrix that crosses ranks: the Schur system of the shared block, in place on the device
                            reduce_dev(d_S, (long long)ns * ns);
                            reduce_dev(d_h, ns);
                        }
                        for (int j0 = 0; j0 < ns; j0 += S1_PB) {
                            const int jb = std::min(S1_PB, ns - j0), rem = ns - j0 - jb;
                            LAUNCH(k_s1_chol_diag, 1, 1, S1_TRSM_TPB, st, d_S, ns, j0, d_dinv, p.status);
                            if (rem > 0) LAUNCH(k_s1_chol_trsm, (rem + S1_TRSM_TPB - 1) / S1_TRSM_TPB, 1, S1_TRSM_TPB, st, d_S, ns, j0, d_dinv);
                            if (rem > 0) { int nt = (rem + S1_PB - 1) / S1_PB; LAUNCH(k_s1_chol_update, nt, nt, 256, st, d_S, ns, j0, jb); }
                        }
                        LAUNCH(k_s1_tri_solve, 1, 1, S1_CHOL_TPB, st, d_S, ns, d_dinv, d_h, d_ds);
                        hipMemsetAsync(d_out, 0, (size_t)n * 8, st);
                        if (nown > 0) LAUNCH(k_s1_back, nown, 1, S1_TPB, st, d_fcols, fs, d_scols, ns, nsp, d_Linv, d_Y, d_z, d_ds, d_out, f_lo, own_shared);
                        else if (own_shared) return fail(MOSHII_ERR_ARG, "stagei: the rank that owns the shared rows must own at least one frame");
                        fetch(dgn, d_out, n);
                        if (shard) reduce(dgn.data(), n);
                        have_gn = true;
                    }
                    if (!have_gn) {
                        // the factorisation overwrites A: keep a copy for the rho denominator products
                        hipMemcpyAsync(d_L, d_A, (size_t)n * n * 8, hipMemcpyDeviceToDevice, st);
                        for (int j0 = 0; j0 < n; j0 += S1_PB) {
                            const int jb = std::min(S1_PB, n - j0), rem = n - j0 - jb;
                            LAUNCH(k_s1_chol_diag, 1, 1, S1_TRSM_TPB, st, d_L, n, j0, d_dinv, p.status);
                            if (rem > 0) LAUNCH(k_s1_chol_trsm, (rem + S1_TRSM_TPB - 1) / S1_TRSM_TPB, 1, S1_TRSM_TPB, st, d_L, n, j0, d_dinv);
                            if (rem > 0) { int nt = (rem + S1_PB - 1) / S1_PB; LAUNCH(k_s1_chol_update, nt, nt, 256, st, d_L, n, j0, jb); }
                        }
                        LAUNCH(k_s1_tri_solve, 1, 1, S1_CHOL_TPB, st, d_L, n, d_dinv, d_g, d_out);
                        fetch(dgn, d_out, n);
                        have_gn = true;
                    }
                    double ngn = nrm2(dgn);
                    if (ngn <= delta) ddl = dgn;
                    else {
                        double dsq = delta * delta, sq_sd = nsd * nsd, dd = 0, dsdd = 0;
                        for (int i = 0; i < n; ++i) { double df = dgn[i] - dsd[i]; dd += df * df; dsdd += df * dsd[i]; }
                        double gs = dotv(dgn, dsd);
                        double pnow = dd * dsq + gs * gs - ngn * ngn * sq_sd;
                        double beta = (dsq - sq_sd) / (dsdd + sqrt(pnow));
                        ddl.resize(n);
                        for (int i = 0; i < n; ++i) ddl[i] = dsd[i] + beta * (dgn[i] - dsd[i]);
                    }
                }
                double step = nrm2(ddl);
                bool improved = false;
                if (step <= e2 * nrm2(x)) done = true;
                else {
                    xt = x;
                    for (int i = 0; i < n; ++i) xt[i] += ddl[i];
                    const double sse_new = eval_at(xt, 0, rnew);
                    double rho = sse - sse_new;
                    if (rho > 0) {
                        Ax(ddl, tmp);
                        rho = rho / (2.0 * dotv(g, ddl) - dotv(ddl, tmp));
                    }
                    improved = rho > 0;
                    if (improved) {
                        x = xt;
                        if (e3 > 0 && (sse - sse_new) / sse < e3) done = true;
                        else {
                            sse = eval_at(x, 1, r);
                            normal_eq();
                            if (norminf(g) < e1) done = true;
                        }
                    }
                    if (rho > 0.9) delta = std::max(delta, 2.5 * step);
                    else if (rho < 0.05) delta *= 0.25;
                    if (delta <= e2 * nrm2(x)) done = true;
                }
                if (done || improved) break;
            }
            if (iteration >= ds->maxiter) done = true;
        }
        total_iters += iteration;
        unpack(x);
        if (round == ds->n_anneal - 1) {
            upload_point(); evaluate(0); fetch(r, p.r, R);
            if (ds->markers_sim) {       // stagei_markers_sim_all (chmosh.py:441): every latent marker on every frame's body
                double* d_simall = pool.get<double>((size_t)F * M * 3);
                if (!pool.ok) return fail(MOSHII_ERR_HIP, "stagei: device allocation failed");
                hipMemsetAsync(d_simall, 0, (size_t)F * M * 3 * 8, st);
                if (nown > 0) LAUNCH(k_s1_simall, nown, 1, S1_TPB, st, d, p, d_simall, f_lo);
                std::vector<double> sa;
                fetch(sa, d_simall, (size_t)F * M * 3);
                reduce(sa.data(), (long long)F * M * 3);
                memcpy(ds->markers_sim, sa.data(), sa.size() * 8);
            }
            auto sse_rows = [&](int lo, int hi) { double s = 0; for (int i = lo; i < hi; ++i) s += r[i] * r[i]; return s; };
            {   // (the collectives below are issued whether or not THIS rank asked for the output: ranks whose callers differ in which
                //  optional pointers they set must still issue the same sequence of all-reduces)
                std::vector<double> ev(8 + (size_t)M);
                ev[0] = sse_rows(d.r_data, d.r_prior); ev[1] = sse_rows(d.r_prior, d.r_init); ev[2] = sse_rows(d.r_init, d.r_beta);
                ev[3] = sse_rows(d.r_beta, d.r_surf); ev[4] = sse_rows(d.r_surf, d.r_poseH); ev[5] = sse_rows(d.r_poseH, d.r_head); ev[6] = sse_rows(d.r_head, d.r_poseF); ev[7] = sse_rows(d.r_poseF, d.R);
                // rows r_init + 3 m .. + 3: marker m's weighted offset from its initial placement; summed over ranks like errs[2]
                for (int mk = 0; mk < M; ++mk) ev[8 + mk] = sse_rows(d.r_init + 3 * mk, d.r_init + 3 * mk + 3);
                reduce(ev.data(), 8);
                reduce(ev.data() + 8, M);
                if (ds->errs) memcpy(ds->errs, ev.data(), 8 * sizeof(double));
                if (ds->init_sq) memcpy(ds->init_sq, ev.data() + 8, (size_t)M * sizeof(double));
            }
        }
        int hstat[4];
        hipMemcpyAsync(hstat, p.status, sizeof(hstat), hipMemcpyDeviceToHost, st);
        hipStreamSynchronize(st);
        if (reduce_rc) return fail(MOSHII_ERR_ARG, "stagei: the all-reduce callback failed");
        if (hstat[0] == 3) return fail(MOSHII_ERR_ARG, "stagei: a vertex has more than 16 non-zero skinning weights");
        if (hstat[1]) return fail(MOSHII_ERR_NUMERIC, "stagei: normal equations not positive definite");
        if (hstat[2]) return fail(MOSHII_ERR_NUMERIC, "stagei: a marker's nearest vertices are collinear for every one of its 8 neighbours (transformed_lm.py:94-101 runs out of candidates)");
    }
    if (hipGetLastError() != hipSuccess) return fail(MOSHII_ERR_HIP, "stagei: kernel launch failed");
    // ---- outputs; nearest canonical vertex of every latent marker (chmosh.py:420-422)
    upload_point(); canonical();
    fetch(can, p.can, (size_t)3 * d.V);
    for (int m = 0; m < M; ++m) {
        double best = 1e300; int bi = 0;
        for (int v = 0; v < d.V; ++v) {
            double s = 0;
            for (int a = 0; a < 3; ++a) { double t = ml[3 * m + a] - can[3 * v + a]; s += t * t; }
            if (s < best) { best = s; bi = v; }
        }
        if (ds->markers_latent_vids) ds->markers_latent_vids[m] = bi;
    }
    if (ds->betas && !d.per_frame) for (int e = 0; e < nb; ++e) ds->betas[e] = betas[e];
    if (ds->expression && d.per_frame) memcpy(ds->expression, betas.data(), (size_t)F * nb * 8);
    if (ds->markers_latent) memcpy(ds->markers_latent, ml.data(), 3 * M * 8);
    if (ds->pose) memcpy(ds->pose, pose.data(), (size_t)F * NP * 8);
    if (ds->trans) memcpy(ds->trans, trans.data(), (size_t)F * 3 * 8);
    if (ds->iters) ds->iters[0] = total_iters;
    return MOSHII_OK;
}
