// libmoshii C ABI (include/moshii.h): handles, setup kernels, host-side staging and launch logic.
#include "../../include/moshii.h"
#include "moshii_dev.h"
#include "stagei_views.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

extern "C" hipError_t moshii_launch_chain_solve(int nblk, int two_per_cu, int xt, int n_chains, size_t lds_bytes, hipStream_t stream,
                                                const ChainDev* chains, const ModelDev* md, const PriorDev* pr,
                                                const OptsDev* op, const ChainLayout* ly, int coop_g);
extern "C" hipError_t moshii_launch_markers(int F, size_t lds_bytes, hipStream_t stream, const AttachDev* att,
                                            const ModelDev* md, const ChainLayout* ly, const double* pose,
                                            const double* trans, double* out);
extern "C" hipError_t moshii_launch_lbs_f32(hipStream_t stream, const ModelDev* md, int F, const float* pose,
                                            const float* trans, float* verts, void* lbs32);

namespace {

thread_local std::string g_err;
int fail(int code, const std::string& msg) { g_err = msg; return code; }

#define HIP_TRY(expr)                                                                                  \
    do {                                                                                               \
        hipError_t _e = (expr);                                                                        \
        if (_e != hipSuccess)                                                                          \
            return fail(MOSHII_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));            \
    } while (0)

template <class T>
int dev_upload(const T* host, size_t n, T** out) {
    *out = nullptr;
    if (n == 0) return MOSHII_OK;
    HIP_TRY(hipMalloc((void**)out, n * sizeof(T)));
    HIP_TRY(hipMemcpy(*out, host, n * sizeof(T), hipMemcpyHostToDevice));
    return MOSHII_OK;
}

struct Scratch {   // growable device buffer reused across calls (small control data)
    char* ptr = nullptr;
    size_t cap = 0;
    hipStream_t last_stream = nullptr;
    bool used = false;
    int reserve(size_t bytes) {
        if (used) { hipStreamSynchronize(last_stream); used = false; }
        if (bytes <= cap) return MOSHII_OK;
        if (ptr) hipFree(ptr);
        ptr = nullptr; cap = 0;
        size_t want = std::max<size_t>(bytes, 1 << 16);
        if (hipMalloc((void**)&ptr, want) != hipSuccess) return fail(MOSHII_ERR_HIP, "scratch hipMalloc failed");
        cap = want;
        return MOSHII_OK;
    }
    ~Scratch() { if (ptr) hipFree(ptr); }
};

}  // namespace

struct moshii_model_s {
    int V = 0, K = 0, NB = 0, P = 0, NP = 0, body_dof = 0, hand_dof = 0, nhand_full = 0, maxdepth = 0;
    std::vector<int> parents, depth;
    std::vector<double> weights_host;   // [V][K] (attachment packing)
    double *d_vt = nullptr, *d_shapedirs = nullptr, *d_posedirs = nullptr, *d_weights = nullptr, *d_Jreg = nullptr;
    double *d_vsh = nullptr, *d_J = nullptr, *d_hands_mean = nullptr, *d_comps = nullptr;
    double* d_JS = nullptr;             // [K][nshape][3] (moshii_model_set_free_shape)
    int shape_start = 0, nshape = 0;
    Scratch qscratch;                   // per-chain shape-derivative scratch of the extended chain kernel
    Scratch coopbuf;                    // exchange slots + flags of cooperative chains (moshii_dev.h: CoopDev)
    Scratch handoff;                    // entry / final states and hand-off deviations of a chunked solve's chunks
    int *d_parents = nullptr, *d_depth = nullptr, *d_comp_lo = nullptr, *d_comp_hi = nullptr, *d_col_lo = nullptr, *d_col_hi = nullptr;
    unsigned long long* d_anc = nullptr;
    bool betas_set = false;
    Lbs32Model l32 = {};
    bool l32_valid = false;
    Scratch scratch;
    ModelDev dev() const {
        ModelDev md;
        md.V = V; md.K = K; md.P = P; md.NP = NP; md.body_dof = body_dof; md.hand_dof = hand_dof;
        md.nhand_full = nhand_full; md.maxdepth = maxdepth;
        md.parents = as_gp(d_parents); md.J = as_gp(d_J); md.hands_mean = as_gp(d_hands_mean); md.comps = as_gp(d_comps);
        md.comp_lo = as_gp(d_comp_lo); md.comp_hi = as_gp(d_comp_hi); md.col_lo = as_gp(d_col_lo); md.col_hi = as_gp(d_col_hi);
        md.anc = as_gp(d_anc); md.depth = as_gp(d_depth);
        md.nshape = nshape; md.JS = as_gp(d_JS);
        return md;
    }
};

struct moshii_prior_s {
    int G = 0, npose = 0;
    double *d_means = nullptr, *d_chols = nullptr, *d_halfprec = nullptr, *d_neglogw = nullptr, *d_cnorm = nullptr;
    PriorDev dev() const {
        PriorDev p; p.G = G; p.npose = npose; p.means = as_gp(d_means); p.chols = as_gp(d_chols); p.halfprec = as_gp(d_halfprec);
        p.neglogw = as_gp(d_neglogw); p.cnorm = as_gp(d_cnorm);
        return p;
    }
};

struct moshii_attach_s {
    moshii_model_t model = nullptr;
    int M = 0, Nv = 0, Nvp = 0, NW = 0;
    double *d_vsh = nullptr, *d_Pt = nullptr, *d_Pj = nullptr, *d_ww = nullptr, *d_coef = nullptr, *d_Ssh = nullptr;
    int nshape = 0;                     // free shape block gathered at creation (0: none)
    int* d_wj = nullptr;
    int* d_vids = nullptr;
    AttachDev* d_self = nullptr;
    AttachDev host_view;
};

namespace {

// ---- setup kernels --------------------------------------------------------------------------
__global__ void k_vshaped(int V, int NB, int nb, const double* __restrict__ vt, const double* __restrict__ sd,
                          const double* __restrict__ betas, double* __restrict__ vsh) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;   // (v, i)
    if (idx >= V * 3) return;
    const double* row = sd + (size_t)idx * NB;
    double s = 0.0;
    for (int b = 0; b < nb; ++b) s += row[b] * betas[b];
    vsh[idx] = vt[idx] + s;
}

__global__ void k_joints(int V, const double* __restrict__ Jreg, const double* __restrict__ vsh, double* __restrict__ J) {
    const int k = blockIdx.x;
    __shared__ double red[3][256];
    double s[3] = {0.0, 0.0, 0.0};
    for (int v = threadIdx.x; v < V; v += blockDim.x) {
        const double w = Jreg[(size_t)k * V + v];
        if (w != 0.0) { s[0] += w * vsh[v * 3 + 0]; s[1] += w * vsh[v * 3 + 1]; s[2] += w * vsh[v * 3 + 2]; }
    }
    for (int i = 0; i < 3; ++i) red[i][threadIdx.x] = s[i];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) for (int i = 0; i < 3; ++i) red[i][threadIdx.x] += red[i][threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x < 3) J[k * 3 + threadIdx.x] = red[threadIdx.x][0];
}

__global__ void k_pack_attach(int Nv, int Nvp, int K, const int* __restrict__ vids, const double* __restrict__ posedirs,
                              const double* __restrict__ vsh, double* __restrict__ Pt, double* __restrict__ Pj,
                              double* __restrict__ vsh_out) {
    const int nfeat = 9 * (K - 1);
    const size_t total = (size_t)(K - 1) * 27 * Nvp;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int a = (int)(idx % Nvp);
        const int r = (int)(idx / Nvp);           // (k-1)*27 + i*9 + e
        const int km1 = r / 27, q = r % 27;
        double v = 0.0;
        if (a < Nv) v = posedirs[((size_t)vids[a] * 3 + q / 9) * nfeat + 9 * km1 + q % 9];
        Pt[idx] = v;
        if (a < Nv) {   // Pj[k-1][s][q/2][m] as 16-byte pairs, marker index fastest (a = 3 m + s): see AttachDev
            const int M = Nv / 3, mk = a / 3, sv = a % 3;
            Pj[((((size_t)km1 * 3 + sv) * 14 + q / 2) * M + mk) * 2 + (q & 1)] = v;
            if (q == 26) Pj[((((size_t)km1 * 3 + sv) * 14 + 13) * M + mk) * 2 + 1] = 0.0;
        }
    }
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < Nv * 3) vsh_out[t] = vsh[(size_t)vids[t / 3] * 3 + t % 3];
}

// JS[k][e][i] = sum_v Jreg[k][v] shapedirs[v][i][start + e]: the regressed joints' derivative wrt the free shape block.
__global__ void k_shape_joints(int V, int NB, int start, int E, const double* __restrict__ Jreg, const double* __restrict__ sd,
                               double* __restrict__ JS) {
    const int k = blockIdx.x / E, e = blockIdx.x % E;
    __shared__ double red[3][256];
    double s[3] = {0.0, 0.0, 0.0};
    for (int v = threadIdx.x; v < V; v += blockDim.x) {
        const double w = Jreg[(size_t)k * V + v];
        if (w != 0.0)
            for (int i = 0; i < 3; ++i) s[i] += w * sd[((size_t)v * 3 + i) * NB + start + e];
    }
    for (int i = 0; i < 3; ++i) red[i][threadIdx.x] = s[i];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) for (int i = 0; i < 3; ++i) red[i][threadIdx.x] += red[i][threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x < 3) JS[((size_t)k * E + e) * 3 + threadIdx.x] = red[threadIdx.x][0];
}

// Ssh[e][i][a] = shapedirs[vids[a]][i][start + e]  (vertex index fastest; zero in the padding)
__global__ void k_pack_shape(int Nv, int Nvp, int NB, int start, int E, const int* __restrict__ vids,
                             const double* __restrict__ sd, double* __restrict__ Ssh) {
    const size_t total = (size_t)E * 3 * Nvp;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int a = (int)(idx % Nvp);
        const int r = (int)(idx / Nvp);   // e*3 + i
        Ssh[idx] = (a < Nv) ? sd[((size_t)vids[a] * 3 + r % 3) * NB + start + r / 3] : 0.0;
    }
}

// Reference-precision full-mesh LBS: one block = 256 vertices of one frame.
__global__ void k_lbs_f64(ModelDev md, const double* __restrict__ vsh, const double* __restrict__ posedirs,
                          const double* __restrict__ weights, const double* __restrict__ pose,
                          const double* __restrict__ trans, double* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int K = md.K, P = md.P;
    double* fullpose = sm;            // P
    double* Rl = fullpose + P;        // K*9
    double* Rw = Rl + K * 9;          // K*9
    double* tw = Rw + K * 9;          // K*3
    double* feat = tw + K * 3;        // K*9
    const int f = blockIdx.y, tid = threadIdx.x;
    const double* ps = pose + (size_t)f * md.NP;
    for (int d = tid; d < P; d += blockDim.x) {
        double v;
        if (d < md.body_dof) v = ps[d];
        else {
            const int h = d - md.body_dof;
            v = md.hands_mean[h];
            for (int i = 0; i < md.hand_dof; ++i) v += ps[md.body_dof + i] * md.comps[i * md.nhand_full + h];
        }
        fullpose[d] = v;
    }
    __syncthreads();
    if (tid < K) {
        const double x = fullpose[3 * tid], y = fullpose[3 * tid + 1], z = fullpose[3 * tid + 2];
        const double t2 = x * x + y * y + z * z;
        double a, b;
        if (t2 < 1e-6) { a = 1.0 - t2 / 6.0 + t2 * t2 / 120.0; b = 0.5 - t2 / 24.0 + t2 * t2 / 720.0; }
        else { const double t = sqrt(t2); a = sin(t) / t; b = (1.0 - cos(t)) / t2; }
        const double K2[9] = {x * x - t2, x * y, x * z, x * y, y * y - t2, y * z, x * z, y * z, z * z - t2};
        const double Km[9] = {0.0, -z, y, z, 0.0, -x, -y, x, 0.0};
        for (int e = 0; e < 9; ++e) {
            const double id = (e == 0 || e == 4 || e == 8) ? 1.0 : 0.0;
            const double r = id + a * Km[e] + b * K2[e];
            Rl[tid * 9 + e] = r;
            feat[tid * 9 + e] = r - id;
        }
    }
    __syncthreads();
    if (tid == 0) {   // serial chain: K <= 64 joints
        for (int e = 0; e < 9; ++e) Rw[e] = Rl[e];
        for (int i = 0; i < 3; ++i) tw[i] = md.J[i];
        for (int k = 1; k < K; ++k) {
            const int p = md.parents[k];
            for (int i = 0; i < 3; ++i) {
                for (int j = 0; j < 3; ++j)
                    Rw[k * 9 + i * 3 + j] = Rw[p * 9 + i * 3 + 0] * Rl[k * 9 + 0 * 3 + j] + Rw[p * 9 + i * 3 + 1] * Rl[k * 9 + 1 * 3 + j] +
                                            Rw[p * 9 + i * 3 + 2] * Rl[k * 9 + 2 * 3 + j];
                tw[k * 3 + i] = Rw[p * 9 + i * 3 + 0] * (md.J[k * 3 + 0] - md.J[p * 3 + 0]) + Rw[p * 9 + i * 3 + 1] * (md.J[k * 3 + 1] - md.J[p * 3 + 1]) +
                                Rw[p * 9 + i * 3 + 2] * (md.J[k * 3 + 2] - md.J[p * 3 + 2]) + tw[p * 3 + i];
            }
        }
    }
    __syncthreads();
    const int v = blockIdx.x * blockDim.x + tid;
    if (v >= md.V) return;
    const int nfeat = 9 * (K - 1);
    double vp[3];
    for (int i = 0; i < 3; ++i) {
        const double* row = posedirs + ((size_t)v * 3 + i) * nfeat;
        double s = 0.0;
        for (int q = 0; q < nfeat; ++q) s += row[q] * feat[9 + q];
        vp[i] = vsh[v * 3 + i] + s;
    }
    double acc[3] = {0.0, 0.0, 0.0};
    for (int j = 0; j < K; ++j) {
        const double w = weights[(size_t)v * K + j];
        if (w == 0.0) continue;
        const double dx = vp[0] - md.J[j * 3 + 0], dy = vp[1] - md.J[j * 3 + 1], dz = vp[2] - md.J[j * 3 + 2];
        for (int i = 0; i < 3; ++i)
            acc[i] += w * (Rw[j * 9 + i * 3 + 0] * dx + Rw[j * 9 + i * 3 + 1] * dy + Rw[j * 9 + i * 3 + 2] * dz + tw[j * 3 + i]);
    }
    const double* tr = trans + (size_t)f * 3;
    double* o = out + ((size_t)f * md.V + v) * 3;
    o[0] = acc[0] + tr[0]; o[1] = acc[1] + tr[1]; o[2] = acc[2] + tr[2];
}

// ---- LDS layout of the chain kernel ------------------------------------------------------------
int pick_nblk(int n, bool xt = false) {
    const int opts[5] = {2, 4, 5, 7, 8};
    const int opts_xt[4] = {5, 8, 10, 13};   // instantiations of the extended variant (chain_solve.hip)
    if (xt) { for (int o : opts_xt) if (o * 16 >= n + 1) return o; return -1; }
    for (int o : opts) if (o * 16 >= n + 1) return o;   // +1: the right-hand side rides along as row n (ldl_solve)
    return -1;
}

ChainLayout make_layout(const moshii_model_s* m, int Mmax, int Nvmax, int NWmax, int npose, int G, int nmax, int nkfmax, int Tm, int nblk = 0, int nhj = 0,
                        int nshape = 0) {
    ChainLayout ly;
    memset(&ly, 0, sizeof(ly));
    const int K = m->K, NP = m->NP, P = m->P;
    if (nblk <= 0) nblk = pick_nblk(nmax);
    const int LDJ = nblk * 16;
    ly.Mmax = Mmax; ly.Nvmax = Nvmax; ly.NWmax = NWmax; ly.nmax = nmax; ly.Tm = Tm; ly.nkfmax = nkfmax; ly.LDJ = LDJ; ly.nhj = nhj;
    int off = MOSHII_KC_DOUBLES;   // (the KernelCtx copy sits at the head of the LDS)
    auto take = [&](int nd) { int o = off; off += (nd + 1) & ~1; return o; };
    const int NPX = NP + nshape;   // free shape coefficients ride behind the pose variables
    ly.NPX = NPX;
    ly.o_pose = take(NPX); ly.o_trans = take(4); ly.o_pose_t = take(NPX); ly.o_trans_t = take(4);
    ly.o_vshp = take(nshape > 0 ? Nvmax * 3 : 0); ly.o_shp0 = take(nshape);
    ly.o_pose_prev = take(NP); ly.o_vtarget = take(NP); ly.o_fullpose = take(P);
    ly.o_feat = take(K * 9); ly.o_B = take(K * 28); ly.o_omega = take(K * 10); ly.o_Jl = take(K * 3); ly.o_Rw = take(K * 9); ly.o_tw = take(K * 3);
    ly.o_Rloc = take(K * 9); ly.o_acol = take(K * 9);
    ly.o_vconst = take(Nvmax * 3); ly.o_vposed = take(Nvmax * 3); ly.o_vpos = take(Nvmax * 3); ly.o_msim = take(Mmax * 3); ly.o_res = take(Mmax * 3);
    ly.o_xb = take(std::max(npose, 1) + 16);   // (+16: read unclamped in 16-row groups)
    ly.o_ell = take(std::max(8 * npose, 1) + 16); ly.o_score = take(std::max(G, 1));   // (ell: [2][4 waves][npose] + 16 zeros: read in 16-row groups)
    ly.o_px0 = take(std::max(npose, 1)); ly.o_ps0 = take(std::max(G, 1));
    ly.o_g = take(LDJ); ly.o_dsd = take(LDJ); ly.o_dgn = take(LDJ); ly.o_ddl = take(LDJ); ly.o_y = take(LDJ);
    ly.o_red = take(16); ly.o_scal = take(16);
    ly.o_anc = take(K);
    int io = 0;
    auto itake = [&](int ni) { int o = io; io += ni; return o; };
    ly.i_visidx = itake(Mmax); ly.i_colpid = itake(LDJ); ly.i_colprior = itake(LDJ); ly.i_pid2prior = itake(NP);
    ly.i_jointslot = itake(K); ly.i_kfree = itake(K); ly.i_colq = itake(NP); ly.i_ksum = itake(K); ly.i_kconst = itake(K);
    ly.i_total = io;
    ly.o_ints = take((io + 1) / 2);
    int t = 0;
    auto ttake = [&](int nd) { int o = t; t += (nd + 1) & ~1; return o; };
    ly.t_Jh = ttake(Tm * nhj * 9); ly.t_Jrow = ttake(3 * Tm * LDJ); ly.t_Lm = ttake(Tm * 30);
    ly.t_Trot = ttake(3 * Tm * 10); ly.t_xjs = ttake(3 * Tm * (NWmax * 4 + 2)); ly.t_rest = ttake(3 * Tm);   // (strides: assemble())
    ly.t_tjs = ttake((3 * Tm * (NWmax + 1) + 1) / 2);
    // packed factor + 64 per-lane trash words / zero word + the column broadcast buffer of ldl_solve; beyond 8 register blocks (extended variant)
    // the factor lives in global scratch and LDS keeps the broadcast buffer plus a 16-row panel
    // (ldl_big: a [LDJ][17] block column + nblk exchange tiles; its back-substitution lays two [16][LDJ] row buffers over them)
    const int chol = (nblk > 8) ? 66 + std::max(17 * LDJ + 256 * nblk, 32 * LDJ) + 4 : ((nblk <= 4) ? LDJ * (LDJ + 1) + 16 : (nmax + 1) * (nmax + 2) / 2) + 66 + 4 * LDJ + 4;   // (<= 4 blocks: the factor square, chain_solve.hip: ldl_square)
    ly.big_doubles = std::max(std::max(t, chol), 16 * 256);   // (16 x 256: the J^T J tile exchange, AReg::take)
    ly.o_big = take(ly.big_doubles);
    ly.total_doubles = off;
    return ly;
}

struct LaunchInfo { std::string name; int lds = 0; int threads = 0; } g_last;

}  // namespace

// ================================================================================================
extern "C" {

const char* moshii_last_error(void) { return g_err.c_str(); }
int moshii_version(void) { return 101; }   // 101: moshii_stagei_desc grew by init_sq (include/moshii.h)
#ifndef MOSHII_SRC_HASH
#define MOSHII_SRC_HASH "unknown"
#endif
const char* moshii_source_hash(void) { return MOSHII_SRC_HASH; }

int moshii_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
int moshii_set_device(int device) { HIP_TRY(hipSetDevice(device)); return MOSHII_OK; }

int moshii_device_multiprocessors(void) {
    int d = 0, n = 0;
    if (hipGetDevice(&d) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess) return 0;
    return n;
}

int moshii_model_create(const moshii_model_desc* d, moshii_model_t* out) {
    if (!d || !out) return fail(MOSHII_ERR_ARG, "null argument");
    if (d->K < 1 || d->K > MOSHII_MAXK) return fail(MOSHII_ERR_UNSUPPORTED, "K must be in [1,64]");
    if (moshii_device_count() < 1) return fail(MOSHII_ERR_NO_DEVICE, "no HIP device: libmoshii has no CPU path");
    const int P = 3 * d->K;
    if (d->body_dof < 3 || d->body_dof > P || d->body_dof % 3) return fail(MOSHII_ERR_ARG, "bad body_dof");
    const int nhf = P - d->body_dof;
    if ((d->hand_dof > 0) != (nhf > 0)) return fail(MOSHII_ERR_ARG, "hand_dof / body_dof mismatch");
    if (d->hand_dof > 0 && (!d->hands_mean || !d->selected_components)) return fail(MOSHII_ERR_ARG, "hand PCA arrays missing");
    auto* m = new moshii_model_s();
    m->V = d->V; m->K = d->K; m->NB = d->NB; m->P = P; m->body_dof = d->body_dof; m->hand_dof = d->hand_dof;
    m->nhand_full = nhf; m->NP = d->body_dof + d->hand_dof;
    m->parents.assign(d->parents, d->parents + d->K);
    m->depth.assign(d->K, 0);
    std::vector<unsigned long long> anc(d->K, 0ull);
    for (int j = 0; j < d->K; ++j) {
        int a = j, dep = 0;
        while (a >= 0) {
            anc[a] |= (1ull << j);
            const int p = m->parents[a];
            if (p >= a) { delete m; return fail(MOSHII_ERR_ARG, "parents must precede children"); }
            a = p;
            if (a >= 0) ++dep;
        }
        m->depth[j] = dep;
        m->maxdepth = std::max(m->maxdepth, dep);
    }
    m->weights_host.assign(d->weights, d->weights + (size_t)d->V * d->K);
    const size_t V = d->V, K = d->K, nfeat = 9 * (K - 1);
    int rc;
    if ((rc = dev_upload(d->v_template, V * 3, &m->d_vt))) return rc;
    if ((rc = dev_upload(d->shapedirs, V * 3 * d->NB, &m->d_shapedirs))) return rc;
    if ((rc = dev_upload(d->posedirs, V * 3 * nfeat, &m->d_posedirs))) return rc;
    if ((rc = dev_upload(d->weights, V * K, &m->d_weights))) return rc;
    if ((rc = dev_upload(d->J_regressor, K * V, &m->d_Jreg))) return rc;
    if ((rc = dev_upload(m->parents.data(), K, &m->d_parents))) return rc;
    {   // behind the depths: the joints sorted by depth ([K]) and where each depth starts ([maxdepth + 2]) -- the level-by-level walks of the kernels
        std::vector<int> dl(m->depth);
        for (int l = 0; l <= m->maxdepth; ++l) for (size_t j = 0; j < K; ++j) if (m->depth[j] == l) dl.push_back((int)j);
        int c = 0;
        for (int l = 0; l <= m->maxdepth + 1; ++l) { dl.push_back(c); for (size_t j = 0; j < K; ++j) c += (m->depth[j] == l) ? 1 : 0; }
        if ((rc = dev_upload(dl.data(), dl.size(), &m->d_depth))) return rc;
    }
    if ((rc = dev_upload(anc.data(), K, &m->d_anc))) return rc;
    if (d->hand_dof > 0) {
        std::vector<int> lo(d->hand_dof), hi(d->hand_dof);
        for (int i = 0; i < d->hand_dof; ++i) {
            int l = nhf, h = 0;
            for (int c = 0; c < nhf; ++c)
                if (d->selected_components[(size_t)i * nhf + c] != 0.0) { l = std::min(l, c); h = std::max(h, c + 1); }
            if (l > h) { l = 0; h = 0; }
            lo[i] = l; hi[i] = h;
        }
        if ((rc = dev_upload(d->hands_mean, (size_t)nhf, &m->d_hands_mean))) return rc;
        if ((rc = dev_upload(d->selected_components, (size_t)d->hand_dof * nhf, &m->d_comps))) return rc;
        if ((rc = dev_upload(lo.data(), lo.size(), &m->d_comp_lo))) return rc;
        if ((rc = dev_upload(hi.data(), hi.size(), &m->d_comp_hi))) return rc;
        std::vector<int> clo(nhf), chi(nhf);
        for (int c = 0; c < nhf; ++c) {
            int l = d->hand_dof, h = 0;
            for (int i = 0; i < d->hand_dof; ++i)
                if (d->selected_components[(size_t)i * nhf + c] != 0.0) { l = std::min(l, i); h = std::max(h, i + 1); }
            if (l > h) { l = 0; h = 0; }
            clo[c] = l; chi[c] = h;
        }
        if ((rc = dev_upload(clo.data(), clo.size(), &m->d_col_lo))) return rc;
        if ((rc = dev_upload(chi.data(), chi.size(), &m->d_col_hi))) return rc;
    }
    HIP_TRY(hipMalloc((void**)&m->d_vsh, V * 3 * sizeof(double)));
    HIP_TRY(hipMalloc((void**)&m->d_J, K * 3 * sizeof(double)));
    *out = m;
    std::vector<double> zero(std::max(1, d->NB), 0.0);
    return moshii_model_set_betas(m, zero.data(), d->NB);
}

extern "C" void moshii_lbs32_free(void* l32);   // lbs_forward.hip
static void free_l32(moshii_model_s* m) {
    moshii_lbs32_free(&m->l32);
    m->l32_valid = false;
}

int moshii_model_destroy(moshii_model_t m) {
    if (!m) return MOSHII_OK;
    hipDeviceSynchronize();
    void* ptrs[] = {m->d_vt, m->d_shapedirs, m->d_posedirs, m->d_weights, m->d_Jreg, m->d_vsh, m->d_J, m->d_hands_mean,
                    m->d_comps, m->d_parents, m->d_depth, m->d_comp_lo, m->d_comp_hi, m->d_col_lo, m->d_col_hi, m->d_anc, m->d_JS};
    for (void* p : ptrs) if (p) hipFree(p);
    free_l32(m);
    delete m;
    return MOSHII_OK;
}

int moshii_model_set_betas(moshii_model_t m, const double* betas, int32_t nb) {
    if (!m || (!betas && nb > 0)) return fail(MOSHII_ERR_ARG, "null argument");
    nb = std::min<int32_t>(nb, m->NB);
    double* d_b = nullptr;
    int rc = dev_upload(betas, (size_t)std::max(nb, 0), &d_b);
    if (rc) return rc;
    const int tot = m->V * 3;
    hipLaunchKernelGGL(k_vshaped, dim3((tot + 255) / 256), dim3(256), 0, 0, m->V, m->NB, nb, m->d_vt, m->d_shapedirs, d_b, m->d_vsh);
    hipLaunchKernelGGL(k_joints, dim3(m->K), dim3(256), 0, 0, m->V, m->d_Jreg, m->d_vsh, m->d_J);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    if (d_b) hipFree(d_b);
    m->betas_set = true;
    m->l32_valid = false;
    return MOSHII_OK;
}

int moshii_model_set_free_shape(moshii_model_t m, int32_t start, int32_t count) {
    if (!m || start < 0 || count < 0 || start + count > m->NB) return fail(MOSHII_ERR_ARG, "free shape block outside shapedirs");
    if (count > 125) return fail(MOSHII_ERR_UNSUPPORTED, "at most 125 free shape coefficients");
    HIP_TRY(hipDeviceSynchronize());
    if (m->d_JS) { hipFree(m->d_JS); m->d_JS = nullptr; }
    m->shape_start = start; m->nshape = count;
    if (count == 0) return MOSHII_OK;
    HIP_TRY(hipMalloc((void**)&m->d_JS, (size_t)m->K * count * 3 * sizeof(double)));
    hipLaunchKernelGGL(k_shape_joints, dim3(m->K * count), dim3(256), 0, 0, m->V, m->NB, start, count, m->d_Jreg, m->d_shapedirs, m->d_JS);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    return MOSHII_OK;
}

int moshii_model_get_joints(moshii_model_t m, double* J_out) {
    if (!m || !J_out) return fail(MOSHII_ERR_ARG, "null argument");
    HIP_TRY(hipMemcpy(J_out, m->d_J, (size_t)m->K * 3 * sizeof(double), hipMemcpyDeviceToHost));
    return MOSHII_OK;
}

int moshii_lbs_forward_f64(moshii_model_t m, int32_t F, const double* pose, const double* trans, double* verts,
                           uint32_t flags, void* stream_) {
    if (!m || !pose || !trans || !verts || F < 0) return fail(MOSHII_ERR_ARG, "bad argument");
    if (F == 0) return MOSHII_OK;
    hipStream_t stream = (hipStream_t)stream_;
    const bool dev = (flags & MOSHII_BUFFERS_DEVICE) != 0;
    const double *d_pose = pose, *d_trans = trans;
    double* d_out = verts;
    double *t_pose = nullptr, *t_trans = nullptr, *t_out = nullptr;
    if (!dev) {
        int rc;
        if ((rc = dev_upload(pose, (size_t)F * m->NP, &t_pose))) return rc;
        if ((rc = dev_upload(trans, (size_t)F * 3, &t_trans))) return rc;
        HIP_TRY(hipMalloc((void**)&t_out, (size_t)F * m->V * 3 * sizeof(double)));
        d_pose = t_pose; d_trans = t_trans; d_out = t_out;
    }
    const size_t lds = (size_t)(m->P + m->K * 30) * sizeof(double);
    hipLaunchKernelGGL(k_lbs_f64, dim3((m->V + 255) / 256, F), dim3(256), lds, stream, m->dev(), m->d_vsh, m->d_posedirs,
                       m->d_weights, d_pose, d_trans, d_out);
    HIP_TRY(hipGetLastError());
    if (!dev) {
        HIP_TRY(hipStreamSynchronize(stream));
        HIP_TRY(hipMemcpy(verts, t_out, (size_t)F * m->V * 3 * sizeof(double), hipMemcpyDeviceToHost));
        hipFree(t_pose); hipFree(t_trans); hipFree(t_out);
    }
    return MOSHII_OK;
}

// implemented in lbs_forward.hip
int moshii_lbs32_prepare(moshii_model_t m);

int moshii_lbs_forward_f32(moshii_model_t m, int32_t F, const float* pose, const float* trans, float* verts,
                           uint32_t flags, void* stream_) {
    if (!m || !pose || !trans || !verts || F < 0) return fail(MOSHII_ERR_ARG, "bad argument");
    if (F == 0) return MOSHII_OK;
    hipStream_t stream = (hipStream_t)stream_;
    if (!m->l32_valid) { int rc = moshii_lbs32_prepare(m); if (rc) return rc; }
    const bool dev = (flags & MOSHII_BUFFERS_DEVICE) != 0;
    const float *d_pose = pose, *d_trans = trans;
    float* d_out = verts;
    float *t_pose = nullptr, *t_trans = nullptr, *t_out = nullptr;
    if (!dev) {
        int rc;
        if ((rc = dev_upload(pose, (size_t)F * m->NP, &t_pose))) return rc;
        if ((rc = dev_upload(trans, (size_t)F * 3, &t_trans))) return rc;
        HIP_TRY(hipMalloc((void**)&t_out, (size_t)F * m->V * 3 * sizeof(float)));
        d_pose = t_pose; d_trans = t_trans; d_out = t_out;
    }
    ModelDev md = m->dev();
    HIP_TRY(moshii_launch_lbs_f32(stream, &md, F, d_pose, d_trans, d_out, &m->l32));
    if (!dev) {
        HIP_TRY(hipStreamSynchronize(stream));
        HIP_TRY(hipMemcpy(verts, t_out, (size_t)F * m->V * 3 * sizeof(float), hipMemcpyDeviceToHost));
        hipFree(t_pose); hipFree(t_trans); hipFree(t_out);
    }
    return MOSHII_OK;
}

int moshii_prior_create(int32_t G, int32_t npose, const double* means, const double* chols, const double* weights,
                        moshii_prior_t* out) {
    if (!means || !chols || !weights || !out || G < 1 || npose < 1) return fail(MOSHII_ERR_ARG, "bad argument");
    if (moshii_device_count() < 1) return fail(MOSHII_ERR_NO_DEVICE, "no HIP device: libmoshii has no CPU path");
    auto* p = new moshii_prior_s();
    p->G = G; p->npose = npose;
    const size_t nn = (size_t)npose * npose;
    std::vector<double> half((size_t)G * nn), nlw(G), cn(G), lower((size_t)G * nn, 0.0);   // lower: the factors with explicit zeros above the
    for (int g = 0; g < G; ++g) {                                                    // diagonal (the chain kernel reads whole rows)
        const double* L = chols + g * nn;
        for (int i = 0; i < npose; ++i)
            for (int j = 0; j <= i; ++j) lower[g * nn + (size_t)i * npose + j] = L[(size_t)i * npose + j];
        for (int i = 0; i < npose; ++i)
            for (int j = 0; j <= i; ++j) {
                double s = 0.0;
                for (int a = 0; a <= j; ++a) s += L[(size_t)i * npose + a] * L[(size_t)j * npose + a];   // (L L^T)_ij, L lower
                half[g * nn + (size_t)i * npose + j] = 0.5 * s;
                half[g * nn + (size_t)j * npose + i] = 0.5 * s;
            }
        nlw[g] = -std::log(weights[g]);
        // |L|_2 <= min(|L|_F, sqrt(|L|_1 |L|_inf)); a hair above, so that rounding here cannot make the bound too small
        double fro = 0.0, n1 = 0.0, ninf = 0.0;
        std::vector<double> colsum(npose, 0.0);
        for (int i = 0; i < npose; ++i) {
            double rs = 0.0;
            for (int j = 0; j <= i; ++j) { const double v = std::fabs(L[(size_t)i * npose + j]); fro += v * v; rs += v; colsum[j] += v; }
            ninf = std::max(ninf, rs);
        }
        for (int j = 0; j < npose; ++j) n1 = std::max(n1, colsum[j]);
        cn[g] = std::min(std::sqrt(fro), std::sqrt(n1 * ninf)) * 0.70710678118654757 * (1.0 + 1e-12);
    }
    int rc;
    // (means and factors carry 16 entries / rows of zero padding: the chain kernel reads whole 16-row groups unclamped)
    std::vector<double> means_p((size_t)G * npose + 16, 0.0);
    std::copy(means, means + (size_t)G * npose, means_p.begin());
    lower.resize((size_t)G * nn + (size_t)16 * npose, 0.0);
    if ((rc = dev_upload(means_p.data(), means_p.size(), &p->d_means))) return rc;
    if ((rc = dev_upload(lower.data(), lower.size(), &p->d_chols))) return rc;
    if ((rc = dev_upload(half.data(), half.size(), &p->d_halfprec))) return rc;
    if ((rc = dev_upload(nlw.data(), nlw.size(), &p->d_neglogw))) return rc;
    if ((rc = dev_upload(cn.data(), cn.size(), &p->d_cnorm))) return rc;
    *out = p;
    return MOSHII_OK;
}

int moshii_prior_destroy(moshii_prior_t p) {
    if (!p) return MOSHII_OK;
    hipDeviceSynchronize();
    void* ptrs[] = {p->d_means, p->d_chols, p->d_halfprec, p->d_neglogw, p->d_cnorm};
    for (void* q : ptrs) if (q) hipFree(q);
    delete p;
    return MOSHII_OK;
}

int moshii_attach_create(moshii_model_t m, int32_t M, const int32_t* closest, const double* coef, moshii_attach_t* out) {
    if (!m || !closest || !coef || !out || M < 1) return fail(MOSHII_ERR_ARG, "bad argument");
    if (M > 128) return fail(MOSHII_ERR_UNSUPPORTED, "at most 128 latent markers");
    auto* a = new moshii_attach_s();
    a->model = m; a->M = M; a->Nv = 3 * M; a->Nvp = (a->Nv + 7) & ~7;
    std::vector<int> vids(a->Nv);
    int NW = 1;
    for (int i = 0; i < a->Nv; ++i) {
        vids[i] = closest[i];
        if (vids[i] < 0 || vids[i] >= m->V) { delete a; return fail(MOSHII_ERR_ARG, "vertex id out of range"); }
        int c = 0;
        for (int j = 0; j < m->K; ++j) if (m->weights_host[(size_t)vids[i] * m->K + j] != 0.0) ++c;
        NW = std::max(NW, c);
    }
    a->NW = NW;
    std::vector<int> wj((size_t)a->Nv * NW, 0);
    std::vector<double> ww((size_t)a->Nv * NW, 0.0);
    for (int i = 0; i < a->Nv; ++i) {
        int c = 0;
        for (int j = 0; j < m->K; ++j) {
            const double w = m->weights_host[(size_t)vids[i] * m->K + j];
            if (w != 0.0) { wj[(size_t)i * NW + c] = j; ww[(size_t)i * NW + c] = w; ++c; }
        }
    }
    int rc;
    if ((rc = dev_upload(vids.data(), vids.size(), &a->d_vids))) return rc;
    if ((rc = dev_upload(wj.data(), wj.size(), &a->d_wj))) return rc;
    if ((rc = dev_upload(ww.data(), ww.size(), &a->d_ww))) return rc;
    if ((rc = dev_upload(coef, (size_t)M * 3, &a->d_coef))) return rc;
    HIP_TRY(hipMalloc((void**)&a->d_vsh, (size_t)a->Nv * 3 * sizeof(double)));
    const size_t npt = (size_t)(m->K - 1) * 27 * a->Nvp;
    HIP_TRY(hipMalloc((void**)&a->d_Pt, std::max<size_t>(npt, 1) * sizeof(double)));
    HIP_TRY(hipMalloc((void**)&a->d_Pj, std::max<size_t>((size_t)(m->K - 1) * 3 * 14 * M * 2, 1) * sizeof(double)));
    const int blocks = (int)std::min<size_t>(4096, std::max<size_t>((npt + 255) / 256, (size_t)(a->Nv * 3 + 255) / 256));
    hipLaunchKernelGGL(k_pack_attach, dim3(std::max(blocks, 1)), dim3(256), 0, 0, a->Nv, a->Nvp, m->K, a->d_vids, m->d_posedirs,
                       m->d_vsh, a->d_Pt, a->d_Pj, a->d_vsh);
    HIP_TRY(hipGetLastError());
    if (m->nshape > 0) {   // rows of the free shape block (moshii_model_set_free_shape)
        a->nshape = m->nshape;
        const size_t ns = (size_t)m->nshape * 3 * a->Nvp;
        HIP_TRY(hipMalloc((void**)&a->d_Ssh, ns * sizeof(double)));
        hipLaunchKernelGGL(k_pack_shape, dim3((unsigned)std::min<size_t>(4096, (ns + 255) / 256)), dim3(256), 0, 0, a->Nv, a->Nvp, m->NB,
                           m->shape_start, m->nshape, a->d_vids, m->d_shapedirs, a->d_Ssh);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipDeviceSynchronize());
    AttachDev& av = a->host_view;
    av.M = M; av.Nv = a->Nv; av.Nvp = a->Nvp; av.NW = NW;
    av.vsh = as_gp(a->d_vsh); av.Pt = as_gp(a->d_Pt); av.Pj = as_gp(a->d_Pj); av.wj = as_gp(a->d_wj); av.ww = as_gp(a->d_ww);
    av.coef = as_gp(a->d_coef); av.Ssh = as_gp(a->d_Ssh);
    if ((rc = dev_upload(&av, 1, &a->d_self))) return rc;
    *out = a;
    return MOSHII_OK;
}

int moshii_attach_destroy(moshii_attach_t a) {
    if (!a) return MOSHII_OK;
    hipDeviceSynchronize();
    void* ptrs[] = {a->d_vsh, a->d_Pt, a->d_Pj, a->d_ww, a->d_coef, a->d_wj, a->d_vids, a->d_self, a->d_Ssh};
    for (void* q : ptrs) if (q) hipFree(q);
    delete a;
    return MOSHII_OK;
}

int moshii_attach_markers(moshii_attach_t a, int32_t F, const double* pose, const double* trans, double* markers) {
    if (!a || !pose || !trans || !markers || F < 0) return fail(MOSHII_ERR_ARG, "bad argument");
    if (F == 0) return MOSHII_OK;
    moshii_model_t m = a->model;
    ChainLayout ly = make_layout(m, a->M, a->Nv, a->NW, 0, 0, 16, 1, 1);
    double *d_pose = nullptr, *d_trans = nullptr, *d_out = nullptr;
    int rc;
    if ((rc = dev_upload(pose, (size_t)F * m->NP, &d_pose))) return rc;
    if ((rc = dev_upload(trans, (size_t)F * 3, &d_trans))) return rc;
    HIP_TRY(hipMalloc((void**)&d_out, (size_t)F * a->M * 3 * sizeof(double)));
    ModelDev md = m->dev();
    HIP_TRY(moshii_launch_markers(F, (size_t)ly.total_doubles * sizeof(double), 0, a->d_self, &md, &ly, d_pose, d_trans, d_out));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(markers, d_out, (size_t)F * a->M * 3 * sizeof(double), hipMemcpyDeviceToHost));
    hipFree(d_pose); hipFree(d_trans); hipFree(d_out);
    return MOSHII_OK;
}

}  // extern "C" (reopened below)

// ---- shared launch preparation for moshii_chain_solve / moshii_sequence_solve -----------------------
namespace {

struct LaunchCfg {
    int nblk = 0;
    int two_per_cu = 0;
    int xt = 0;                 // extended kernel variant (jaw term / free shape block)
    int coop_g = 0;             // > 0: cooperative chains, this many workgroups per chain
    int coop_prior_rank = 0;
    int coop_slot_doubles = 0;
    double coop_prior_frac = 0.4;
    ChainLayout ly;
    size_t lds_bytes = 0;
    OptsDev od;
    PriorDev pd;
    ModelDev md;
};

// Validates the options, picks the J^T J register tiling and the LDS layout (marker-tile size Tm as large as the
// per-workgroup LDS budget allows) and uploads the id lists.  `extra_bytes` of the model's control scratch are
// reserved after the id lists; their device/host offsets come back through ctl_off.
// The cooperative-chain request of a call: MOSHII_COOP_GROUP(g) in `flags` (0 = no word: the environment variable `env`, else the library's
// choice; 1 = plain chains; 2 .. 8 = that many workgroups per chain).  Returns -1 (library's choice), 0 (plain), g, or -2 (out of range).
// Set once a cooperative group has broken up in this process (a rank did not become resident within the wait limit: the device is
// shared with another process, or CUs are masked): from then on the library's OWN choice is plain chains -- every further call would
// pay the wait limit and the repeated solve again.  An explicit MOSHII_COOP_GROUP(g) / MOSHII_COOP=g request is still honoured.
static std::atomic<bool> g_coop_broke_once{false};   // (process-wide and sticky; solves may run on several host threads)
static void note_coop_broken(const char* where) {
    if (!g_coop_broke_once.exchange(true)) fprintf(stderr, "[moshii] %s: a cooperative group broke up (the device is shared?); the call is repeated with plain chains, "
                                            "and the library's own choice is plain chains for the rest of this process\n", where);
}
// test aid: MOSHII_COOP_SKEW=seed (read at every call) -> CoopDev::skew, the ranks' arrival order at the exchanges randomised
int coop_skew_env() {
    const char* e = getenv("MOSHII_COOP_SKEW");
    return (e && *e) ? atoi(e) : 0;
}
int coop_request(uint32_t flags, const char* env) {
    int g = (int)((flags >> 8) & 0xffu);
    if (g == 0) {
        const char* e = getenv(env);
        if (!e) {
#ifdef MOSHII_EMULATION   // (the CPU emulation runs workgroups one after another unless told otherwise: a group would wait for itself)
            return 0;
#else
            return g_coop_broke_once.load() ? 0 : -1;
#endif
        }
        if (!strcmp(e, "auto")) return g_coop_broke_once.load() ? 0 : -1;   // (the library's choice, spelled out -- also how the emulation reaches it)
        g = atoi(e);
        if (g == 0) return 0;
    }
    if (g == 1) return 0;
    if (g < 0 || g > MOSHII_COOP_MAXG) return -2;
    return g;
}

// Cooperative chains: the markers [mlo[r], mlo[r + 1]) of rank r.  The ranks 0 .. G-2 get equal shares, the last rank -- which also
// evaluates the prior for the group -- `prior_frac` of one (MOSHII_COOP_PRIOR_FRAC; 1 without a prior).
void coop_split(int M, int G, double prior_frac, int* mlo) {
    const double w_last = (G > 1) ? prior_frac : 1.0;
    const double total = (G - 1) + w_last;
    double acc = 0.0;
    mlo[0] = 0;
    for (int r = 0; r < G; ++r) {
        acc += (r == G - 1) ? w_last : 1.0;
        mlo[r + 1] = (r == G - 1) ? M : std::min(M, (int)std::lround(M * acc / total));
        if (mlo[r + 1] < mlo[r]) mlo[r + 1] = mlo[r];
    }
}

int prepare_launch(moshii_model_t m, moshii_prior_t prior, const moshii_solve_opts* o, int Mmax, int Nvmax, int NWmax,
                   int n_workgroups, hipStream_t stream, size_t extra_bytes, LaunchCfg* cfg, size_t* ctl_off, int coop_g = 0) {
    if (!m->betas_set) return fail(MOSHII_ERR_ARG, "moshii_model_set_betas has not been called");
    if (o->n_body > 0 && (!prior || prior->npose != o->n_body)) return fail(MOSHII_ERR_ARG, "prior npose must equal n_body");
    if (o->n_step1 < 0 || o->n_step2 < 0 || o->n_step1 > m->NP || o->n_step2 > m->NP) return fail(MOSHII_ERR_ARG, "bad free-variable lists");
    const int NP = m->NP;
    const int nshape = o->n_shape;
    if (nshape < 0 || o->n_face < 0) return fail(MOSHII_ERR_ARG, "bad Step-2 extras");
    if (nshape > 0 && nshape != m->nshape) return fail(MOSHII_ERR_ARG, "n_shape must equal the block of moshii_model_set_free_shape");
    if (o->n_face > 0 && !o->face_ids) return fail(MOSHII_ERR_ARG, "face_ids missing");
    for (int i = 1; i < o->n_face; ++i)
        if (o->face_ids[i] != o->face_ids[i - 1] + 1) return fail(MOSHII_ERR_UNSUPPORTED, "face ids must be contiguous");
    const int xt = (nshape > 0 || o->n_face > 0) ? 1 : 0;
    const int nmax = 3 + std::max(o->n_step1, o->n_step2 + nshape);
    int nblk = pick_nblk(nmax, xt != 0);
    if (nblk < 0) return fail(MOSHII_ERR_UNSUPPORTED, xt ? "more than 207 unknowns per step" : "more than 125 free pose variables per step");
    if (const char* e = getenv("MOSHII_FORCE_NBLK")) nblk = std::max(nblk, atoi(e));
    auto count_kf = [&](const int32_t* ids, int n) {   // needed joints (superset over both steps)
        std::vector<char> need(m->K, 0);
        for (int i = 0; i < n; ++i) {
            if (ids[i] < 0 || ids[i] >= NP) return -1;
            if (ids[i] < m->body_dof) need[ids[i] / 3] = 1;
            else for (int k = m->body_dof / 3; k < m->K; ++k) need[k] = 1;
        }
        int c = 0; for (char b : need) c += b; return c;
    };
    const int kf1 = count_kf(o->step1_ids, o->n_step1), kf2 = count_kf(o->step2_ids, o->n_step2);
    if (kf1 < 0 || kf2 < 0) return fail(MOSHII_ERR_ARG, "free pose id out of range");
    const int nkfmax = std::max(1, std::max(kf1, kf2));
    if (o->n_finger > 0)
        for (int i = 1; i < o->n_finger; ++i)
            if (o->finger_ids[i] != o->finger_ids[i - 1] + 1) return fail(MOSHII_ERR_UNSUPPORTED, "finger ids must be contiguous");
    const int G = prior ? prior->G : 0, npose = prior ? prior->npose : 0;
    // LDS budget per workgroup: a lone chain may take the whole CU (160 KiB); a grid that fills the chip leaves
    // room for two workgroups per CU so that one chain's dependency stalls are covered by the other's work.
    int n_cu = 256;
    { int dev = 0; hipGetDevice(&dev); hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev); if (n_cu < 1) n_cu = 256; }
    // One workgroup per CU: a 256-register variant (two per CU) spilled and was measured slower per chain by 1.9x for +6 %
    // aggregate throughput (round 1, tools/gpu_occupancy.py); it was removed in round 3.
    const int two_per_cu = 0;
    (void)n_workgroups;
    int budget = two_per_cu ? 80 * 1024 : 160 * 1024;
    if (const char* e = getenv("MOSHII_LDS_BUDGET")) budget = atoi(e);
    // hand joints whose d marker / d fullpose must be parked for the PCA contraction (only when hand coefficients are free)
    bool hand_free = false;
    for (int i = 0; i < o->n_step1; ++i) hand_free |= o->step1_ids[i] >= m->body_dof;
    for (int i = 0; i < o->n_step2; ++i) hand_free |= o->step2_ids[i] >= m->body_dof;
    const int nhj = hand_free ? (m->K - m->body_dof / 3) : 0;
    // Marker-tile size Tm (T0 maps tile vertices to threads 0..127: 3 Tm <= 120).  The Jacobian rows of a tile are built by
    // (tile marker, needed joint) items, 256 at a time, so among the sizes that fit the LDS budget take the one with the
    // fewest item rounds + tiles over a fully visible frame (53 markers x 20 joints: 38 + 15 = 3 + 2 rounds, where two equal
    // tiles of 27 / 26 cost 3 + 3), weighted by what a round and a tile's fixed work cost (about 3 : 4).
    int Tm = std::min(40, std::max(2, Mmax));
    ChainLayout ly = make_layout(m, Mmax, Nvmax, NWmax, npose, G, nmax, nkfmax, Tm, nblk, nhj, nshape);
    // coop_g: -1 = the library's choice, 0 = plain chains, 2 .. 8 = that many workgroups per chain.  The choice: one rank per round of
    // (marker, joint) Jacobian items (256 threads build 256 of them at a time) plus one for the prior, when every workgroup of the
    // launch can be resident at once (n_workgroups chains x g <= CUs) and the solve is a plain body / finger solve.
    if (coop_g == -1) {
        const int item_ranks = (Mmax * nkfmax + MOSHII_TPB - 1) / MOSHII_TPB;
        coop_g = std::min(MOSHII_COOP_MAXG, item_ranks + (npose > 0 ? 1 : 0));
        if (coop_g < 3 || (!xt && nblk < 4)) coop_g = 0;   // (few items, or a solve so small -- MANO -- that the exchanges cost what the split saves: measured)
    }
    if (coop_g > 0 && ((!xt && nmax + 1 > 8 * 16) || (long long)coop_g * std::max(n_workgroups, 1) > n_cu)) coop_g = 0;   // (not built / not all resident: plain chains)
    if (coop_g > 0) {   // cooperative chains: a rank builds the rows of its own markers only -- one tile of its largest possible share
        if (!xt && nblk < 4) nblk = 4;   // (cooperative instantiations: 4, 5, 7, 8 register blocks; extended variant: 5, 8, 10, 13 as the plain one)
        if (nmax + 1 > nblk * 16) return fail(MOSHII_ERR_UNSUPPORTED, "cooperative chains: too many unknowns");
        if (coop_g > MOSHII_COOP_MAXG) return fail(MOSHII_ERR_ARG, "cooperative chains: at most 8 workgroups per chain");
        // with a prior: from three ranks on the last one does nothing but the prior (measured: its evaluation is as long as the others' forward pass)
        // (the extended variant's per-marker work -- shape columns, 13-block tiles -- outweighs the prior: its rank takes half a share of markers too;
        //  measured on config 3, eight ranks: 3.53 / 3.30 / 3.65 / 3.60 ms per cold frame at 0 / 0.5 / 0.8 / 1)
        cfg->coop_prior_frac = (npose > 0) ? ((coop_g >= 3) ? (xt ? 0.5 : 0.0) : 0.4) : 1.0;
        if (const char* e = getenv("MOSHII_COOP_PRIOR_FRAC")) cfg->coop_prior_frac = std::min(1.0, std::max(0.0, atof(e)));
        int mlo[MOSHII_COOP_MAXG + 1];
        coop_split(Mmax, coop_g, cfg->coop_prior_frac, mlo);
        int share = 2;
        for (int r = 0; r < coop_g; ++r) share = std::max(share, mlo[r + 1] - mlo[r]);
        Tm = std::min(40, share + 1);   // (+1: a chain with fewer markers than Mmax rounds its shares on its own)
        ly = make_layout(m, Mmax, Nvmax, NWmax, npose, G, nmax, nkfmax, Tm, nblk, nhj, nshape);
        while (Tm > 2 && (size_t)ly.total_doubles * 8 > (size_t)budget) {   // (a share larger than the LDS leaves room for: several tiles per rank)
            --Tm;
            ly = make_layout(m, Mmax, Nvmax, NWmax, npose, G, nmax, nkfmax, Tm, nblk, nhj, nshape);
        }
        const int NE = nblk * (nblk + 1) / 2, NT = (NE + 3) / 4;
        // slot: accumulators (2 NT 16-byte units per thread), prior block + gradient ((NE + 2) / 2 units), or 3 M marker coordinates; + 32 granule words
        cfg->coop_slot_doubles = std::max((4 * NT + 2 * ((NE + 2) / 2)) * MOSHII_TPB, 3 * Mmax + 2) + 32;
        cfg->coop_prior_rank = coop_g - 1;
    } else
    if (const char* e = getenv("MOSHII_TM")) {
        Tm = std::max(1, std::min(40, atoi(e)));
        ly = make_layout(m, Mmax, Nvmax, NWmax, npose, G, nmax, nkfmax, Tm, nblk, nhj, nshape);
    } else {
        int best = -1, best_cost = 0;
        for (int t = std::min(40, std::max(2, Mmax)); t >= 2; --t) {
            const ChainLayout lt = make_layout(m, Mmax, Nvmax, NWmax, npose, G, nmax, nkfmax, t, nblk, nhj, nshape);
            if ((size_t)lt.total_doubles * 8 > (size_t)budget) continue;
            const int full = Mmax / t, rem = Mmax % t;
            const int rounds = full * ((t * nkfmax + 255) / 256) + (rem ? (rem * nkfmax + 255) / 256 : 0);
            const int cost = 3 * rounds + 4 * (full + (rem ? 1 : 0));
            if (best < 0 || cost < best_cost) { best = t; best_cost = cost; }
        }
        if (best < 0) best = 2;   // (does not fit: reported below)
        Tm = best;
        ly = make_layout(m, Mmax, Nvmax, NWmax, npose, G, nmax, nkfmax, Tm, nblk, nhj, nshape);
    }
    const size_t lds_bytes = (size_t)ly.total_doubles * sizeof(double);
    if (lds_bytes > 160 * 1024) return fail(MOSHII_ERR_UNSUPPORTED, "problem does not fit the 160 KiB LDS of a CU");

    const size_t nids = (size_t)o->n_step1 + o->n_step2 + o->n_body + o->n_finger + o->n_face;
    const size_t need = sizeof(int) * (nids + 16) + 64 + extra_bytes;
    int rc = m->scratch.reserve(need);
    if (rc) return rc;
    std::vector<int> ids;
    ids.reserve(nids);
    const size_t e1 = 0, e2 = e1 + o->n_step1, eb = e2 + o->n_step2, ef = eb + o->n_body;
    ids.insert(ids.end(), o->step1_ids, o->step1_ids + o->n_step1);
    ids.insert(ids.end(), o->step2_ids, o->step2_ids + o->n_step2);
    ids.insert(ids.end(), o->body_ids, o->body_ids + o->n_body);
    ids.insert(ids.end(), o->finger_ids, o->finger_ids + o->n_finger);
    const size_t efc = ids.size();
    ids.insert(ids.end(), o->face_ids, o->face_ids + o->n_face);
    char* dbase = m->scratch.ptr;
    if (!ids.empty()) {
        HIP_TRY(hipMemcpyAsync(dbase, ids.data(), ids.size() * sizeof(int), hipMemcpyHostToDevice, stream));
        HIP_TRY(hipStreamSynchronize(stream));   // `ids` is pageable and goes out of scope
    }
    m->scratch.used = true; m->scratch.last_stream = stream;
    *ctl_off = (ids.size() * sizeof(int) + 63) & ~size_t(63);

    OptsDev& od = cfg->od;
    od.wt_data = o->wt_data; od.wt_velo = o->wt_velo; od.wt_poseB = o->wt_poseB; od.wt_poseH = o->wt_poseH;
    od.wt_annealing = o->wt_annealing; od.num_train_markers = o->num_train_markers;
    od.e3_first = o->e3_first; od.e3 = o->e3; od.delta0 = o->delta0; od.maxiter = o->maxiter;
    od.n1 = o->n_step1; od.n2 = o->n_step2; od.nbody = o->n_body; od.nfinger = o->n_finger;
    od.same_sets = (nshape == 0 && o->n_step1 == o->n_step2 && std::equal(o->step1_ids, o->step1_ids + o->n_step1, o->step2_ids)) ? 1 : 0;
    od.wt_poseF = o->wt_poseF; od.wt_shape = o->wt_shape; od.wt_shape_stay = o->wt_shape_stay;
    od.nface = o->n_face; od.nshape = nshape;
    const int* dids = (const int*)dbase;
    od.step1 = as_gp(dids + e1); od.step2 = as_gp(dids + e2); od.body = as_gp(dids + eb); od.finger = as_gp(dids + ef); od.face = as_gp(dids + efc);
    memset(&cfg->pd, 0, sizeof(cfg->pd));
    if (prior) cfg->pd = prior->dev();
    cfg->md = m->dev();
    cfg->nblk = nblk; cfg->two_per_cu = xt ? 0 : two_per_cu; cfg->xt = xt; cfg->ly = ly; cfg->lds_bytes = lds_bytes;
    cfg->coop_g = coop_g;
    return MOSHII_OK;
}

int launch_chains(const LaunchCfg& cfg, int n, const ChainDev* d_chains, hipStream_t stream) {
    HIP_TRY(moshii_launch_chain_solve(cfg.nblk, cfg.two_per_cu, cfg.xt, n, cfg.lds_bytes, stream, d_chains, &cfg.md, &cfg.pd, &cfg.od, &cfg.ly, cfg.coop_g));
    g_last.name = "k_chain_solve<" + std::to_string(cfg.nblk) + "," + std::to_string(cfg.two_per_cu ? 2 : 1) + (cfg.xt ? ",xt" : "") +
                  (cfg.coop_g > 0 ? ",coop" + std::to_string(cfg.coop_g) : "") + ">";
    g_last.lds = (int)cfg.lds_bytes; g_last.threads = MOSHII_TPB;
    return MOSHII_OK;
}

// host<->device staging of one sequence's per-frame buffers (MOSHII_BUFFERS_HOST callers)
struct Staged {
    double *obs = nullptr, *pose = nullptr, *fullpose = nullptr, *trans = nullptr, *msim = nullptr, *errs = nullptr;
    uint8_t* vis = nullptr;
    int *iters = nullptr, *status = nullptr;
    void release() {
        void* ptrs[] = {obs, vis, pose, fullpose, trans, msim, errs, iters, status};
        for (void* q : ptrs) if (q) hipFree(q);
        *this = Staged();
    }
};

struct FrameBufs {   // the per-frame arrays shared by moshii_chain_desc and moshii_sequence_desc
    int F, M;
    const double* obs; const uint8_t* vis;
    double *pose, *fullpose, *trans, *msim, *errs; int *iters, *status;
};

int stage_in(const FrameBufs& h, int NP, int P, hipStream_t stream, Staged* s) {
    const size_t Fz = std::max(h.F, 1), M = h.M;
    int rc;
    if ((rc = dev_upload(h.obs, (size_t)h.F * M * 3, &s->obs))) return rc;
    if ((rc = dev_upload(h.vis, (size_t)h.F * M, &s->vis))) return rc;
    HIP_TRY(hipMalloc((void**)&s->pose, Fz * NP * sizeof(double)));
    HIP_TRY(hipMalloc((void**)&s->fullpose, Fz * P * sizeof(double)));
    HIP_TRY(hipMalloc((void**)&s->trans, Fz * 3 * sizeof(double)));
    HIP_TRY(hipMalloc((void**)&s->msim, Fz * M * 3 * sizeof(double)));
    HIP_TRY(hipMalloc((void**)&s->errs, Fz * MOSHII_NERR * sizeof(double)));
    HIP_TRY(hipMalloc((void**)&s->iters, Fz * 2 * sizeof(int)));
    HIP_TRY(hipMalloc((void**)&s->status, Fz * sizeof(int)));
    HIP_TRY(hipMemsetAsync(s->pose, 0, Fz * NP * sizeof(double), stream));
    HIP_TRY(hipMemsetAsync(s->fullpose, 0, Fz * P * sizeof(double), stream));
    HIP_TRY(hipMemsetAsync(s->trans, 0, Fz * 3 * sizeof(double), stream));
    HIP_TRY(hipMemsetAsync(s->msim, 0, Fz * M * 3 * sizeof(double), stream));
    HIP_TRY(hipMemsetAsync(s->errs, 0, Fz * MOSHII_NERR * sizeof(double), stream));
    HIP_TRY(hipMemsetAsync(s->iters, 0, Fz * 2 * sizeof(int), stream));
    HIP_TRY(hipMemsetAsync(s->status, 0, Fz * sizeof(int), stream));
    return MOSHII_OK;
}

int stage_out(const FrameBufs& h, int NP, int P, Staged* s) {
    const size_t F = h.F, M = h.M;
    if (F) {
        if (h.pose) HIP_TRY(hipMemcpy(h.pose, s->pose, F * NP * sizeof(double), hipMemcpyDeviceToHost));
        if (h.fullpose) HIP_TRY(hipMemcpy(h.fullpose, s->fullpose, F * P * sizeof(double), hipMemcpyDeviceToHost));
        if (h.trans) HIP_TRY(hipMemcpy(h.trans, s->trans, F * 3 * sizeof(double), hipMemcpyDeviceToHost));
        if (h.msim) HIP_TRY(hipMemcpy(h.msim, s->msim, F * M * 3 * sizeof(double), hipMemcpyDeviceToHost));
        if (h.errs) HIP_TRY(hipMemcpy(h.errs, s->errs, F * MOSHII_NERR * sizeof(double), hipMemcpyDeviceToHost));
        if (h.iters) HIP_TRY(hipMemcpy(h.iters, s->iters, F * 2 * sizeof(int), hipMemcpyDeviceToHost));
        if (h.status) HIP_TRY(hipMemcpy(h.status, s->status, F * sizeof(int), hipMemcpyDeviceToHost));
    }
    s->release();
    return MOSHII_OK;
}

// max |entry_c - final_pred(c)| per chunk (pose, trans, and pose_prev when the velocity term is live); a flag
// mismatch (first-frame schedule pending / velocity term missing on one side) counts as infinite deviation.
__global__ void k_verify_chunks(int n, int NP, int E, const int* __restrict__ pred, const double* __restrict__ entry,
                                const double* __restrict__ fin, double* __restrict__ dev) {
    const int c = blockIdx.x;
    if (c >= n) return;
    const int p = pred[c];
    __shared__ double red[64];
    double d = 0.0;
    if (p >= 0) {
        const int S = 2 * NP + 5 + E;
        const double* a = entry + (size_t)c * S;
        const double* b = fin + (size_t)p * S;
        const bool flags_ok = (a[2 * NP + 3] == b[2 * NP + 3]) && (a[2 * NP + 4] == b[2 * NP + 4]);
        const bool hp = a[2 * NP + 3] != 0.0;
        bool nan = false;   // (fmax drops a NaN operand: track it separately, a NaN hand-off must never pass as deviation 0)
        for (int i = threadIdx.x; i < 2 * NP + 3; i += blockDim.x) {
            if (i >= NP && i < 2 * NP && !hp) continue;
            const double v = fabs(a[i] - b[i]);
            nan |= !(v == v);
            d = fmax(d, v);
        }
        for (int e = threadIdx.x; e < E; e += blockDim.x) {   // free shape block
            const double v = fabs(a[2 * NP + 5 + e] - b[2 * NP + 5 + e]);
            nan |= !(v == v);
            d = fmax(d, v);
        }
        if (nan) d = 2e300;          // a NaN state: reported as MOSHII_ERR_NUMERIC by the caller
        else if (a[2 * NP + 3] == -1.0 && b[2 * NP + 3] != -2.0) d = 5e299;   // this chunk's pass-1 chain gave the chunk up (ChainDev::tail_done): re-solve it, nothing else is wrong
        else if (b[2 * NP + 3] == -2.0) d = 4e299;                             // the predecessor was given up: its sweep hands over at the boundary
        else if (!flags_ok) d = 1e300;
    }
    red[threadIdx.x] = d;
    __syncthreads();
    for (int o = 32; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + o]); __syncthreads(); }
    if (threadIdx.x == 0) dev[c] = red[0];
}

}  // namespace

extern "C" {

int moshii_chain_solve(moshii_model_t m, moshii_prior_t prior, const moshii_solve_opts* o, int32_t n_chains,
                       const moshii_chain_desc* chains, uint32_t flags, void* stream_) {
    if (!m || !o || !chains || n_chains < 1) return fail(MOSHII_ERR_ARG, "bad argument");
    hipStream_t stream = (hipStream_t)stream_;
    const bool dev = (flags & MOSHII_BUFFERS_DEVICE) != 0;
    const int NP = m->NP, P = m->P;
    int Mmax = 0, Nvmax = 0, NWmax = 1;
    for (int c = 0; c < n_chains; ++c) {
        const moshii_chain_desc& ch = chains[c];
        if (!ch.attach || ch.attach->model != m || ch.F < 0 || !ch.obs || !ch.vis) return fail(MOSHII_ERR_ARG, "bad chain descriptor");
        Mmax = std::max(Mmax, ch.attach->M); Nvmax = std::max(Nvmax, ch.attach->Nv); NWmax = std::max(NWmax, ch.attach->NW);
    }
    // control block after the id lists: [ChainDev x n][init vectors]
    const int E = o->n_shape > 0 ? o->n_shape : 0;
    const size_t extra = sizeof(ChainDev) * n_chains + sizeof(double) * (size_t)n_chains * (2 * NP + 4 + E + 2) + 256;
    // cooperative chains (MOSHII_COOP_GROUP(g) in `flags`, or MOSHII_COOP=g): g workgroups per chain, all of them resident at once
    int coop_req = coop_request(flags, "MOSHII_COOP");
    if (coop_req < -1) return fail(MOSHII_ERR_ARG, "cooperative chains: group size out of range");
    LaunchCfg cfg;
    size_t ctl = 0;
    int rc = prepare_launch(m, prior, o, Mmax, Nvmax, NWmax, n_chains, stream, extra, &cfg, &ctl, coop_req);
    if (rc) return rc;
    const int coop_g = cfg.coop_g;   // (0: plain chains -- asked for, or the group does not fit the chip / the solve is an extended one)
    size_t coop_bytes_per_chain = 0;
    if (coop_g > 0) {
        coop_bytes_per_chain = ((size_t)2 * coop_g * cfg.coop_slot_doubles * sizeof(unsigned long long) + (size_t)(2 * coop_g + 2) * sizeof(unsigned) + 255) & ~size_t(255);
        if ((rc = m->coopbuf.reserve(coop_bytes_per_chain * n_chains))) return rc;
        m->coopbuf.used = true; m->coopbuf.last_stream = stream;
        HIP_TRY(hipMemsetAsync(m->coopbuf.ptr, 0, coop_bytes_per_chain * n_chains, stream));   // flags and abort words start at zero on EVERY call
    }
    char* dbase = m->scratch.ptr + ctl;
    // extended variant: per-chain scratch for the shape derivatives of the joint transforms ([2][K][E][3] doubles)
    const size_t nfac = (cfg.nblk > 8) ? (size_t)(cfg.ly.nmax + 1) * (cfg.ly.nmax + 2) / 2 + 12 + 256 : 0;   // global packed factor + trash / zero words + a spare word per thread (ldl_big)
    const size_t qbytes = ((size_t)2 * m->K * E * 3 + nfac) * sizeof(double);
    const int qranks = (coop_g > 0) ? coop_g : 1;   // (cooperative chains: a slice per rank -- every rank keeps its own derivative arrays / factor)
    if (E > 0) {
        if ((rc = m->qscratch.reserve(qbytes * n_chains * qranks))) return rc;
        m->qscratch.used = true; m->qscratch.last_stream = stream;
    }
    std::vector<double*> d_shape(n_chains, nullptr);
    std::vector<char> hostbuf(extra, 0);
    size_t off = sizeof(ChainDev) * n_chains;
    auto put = [&](const void* src, size_t bytes) { off = (off + 15) & ~size_t(15); size_t o2 = off; memcpy(hostbuf.data() + off, src, bytes); off += bytes; return o2; };
    std::vector<Staged> st(dev ? 0 : n_chains);
    std::vector<ChainDev> cds(n_chains);
    for (int c = 0; c < n_chains; ++c) {
        const moshii_chain_desc& ch = chains[c];
        ChainDev& cd = cds[c];
        memset(&cd, 0, sizeof(cd));
        cd.att = ch.attach->d_self; cd.F = ch.F; cd.first = ch.first_frame_schedule;
        if (coop_g > 0) {
            cd.coop.G = coop_g; cd.coop.prior_rank = cfg.coop_prior_rank; cd.coop.slot_doubles = cfg.coop_slot_doubles; cd.coop.skew = coop_skew_env();
            coop_split(ch.attach->M, coop_g, cfg.coop_prior_frac, cd.coop.mlo);
            char* cb = m->coopbuf.ptr + coop_bytes_per_chain * c;
            cd.coop.slots = as_gp_rw((unsigned long long*)cb);
            cd.coop.flags = as_gp_rw((unsigned*)(cb + (size_t)2 * coop_g * cfg.coop_slot_doubles * sizeof(unsigned long long)));
        }
        if (ch.init_pose) cd.init_pose = (const double*)(dbase + put(ch.init_pose, sizeof(double) * NP));
        if (ch.init_trans) cd.init_trans = (const double*)(dbase + put(ch.init_trans, sizeof(double) * 3));
        if (ch.init_pose_prev) cd.init_prev = (const double*)(dbase + put(ch.init_pose_prev, sizeof(double) * NP));
        if (E > 0) {
            if (ch.init_shape) cd.init_shape = (const double*)(dbase + put(ch.init_shape, sizeof(double) * E));
            cd.qscratch = (double*)(m->qscratch.ptr + qbytes * (size_t)c * qranks);
            cd.coop.qstride = (int)(qbytes / sizeof(double));
        }
        if (dev) {
            cd.obs = ch.obs; cd.vis = ch.vis; cd.pose = ch.pose; cd.fullpose = ch.fullpose; cd.trans = ch.trans;
            cd.msim = ch.markers_sim; cd.errs = ch.errs; cd.iters = ch.iters; cd.status = ch.status;
            cd.shape = E > 0 ? ch.shape : nullptr;
        } else {
            if (E > 0 && ch.shape && ch.F > 0) {
                HIP_TRY(hipMalloc((void**)&d_shape[c], (size_t)ch.F * E * sizeof(double)));
                HIP_TRY(hipMemsetAsync(d_shape[c], 0, (size_t)ch.F * E * sizeof(double), stream));
                cd.shape = d_shape[c];
            }
            const FrameBufs fb{ch.F, ch.attach->M, ch.obs, ch.vis, ch.pose, ch.fullpose, ch.trans, ch.markers_sim, ch.errs, ch.iters, ch.status};
            Staged& s = st[c];
            if ((rc = stage_in(fb, NP, P, stream, &s))) return rc;
            cd.obs = s.obs; cd.vis = s.vis; cd.pose = s.pose; cd.fullpose = s.fullpose; cd.trans = s.trans;
            cd.msim = s.msim; cd.errs = s.errs; cd.iters = s.iters; cd.status = s.status;
        }
    }
    if (off > extra) return fail(MOSHII_ERR_ARG, "internal: scratch overflow");
    memcpy(hostbuf.data(), cds.data(), sizeof(ChainDev) * n_chains);
    HIP_TRY(hipMemcpyAsync(dbase, hostbuf.data(), off, hipMemcpyHostToDevice, stream));
    HIP_TRY(hipStreamSynchronize(stream));   // hostbuf is pageable and goes out of scope
    if ((rc = launch_chains(cfg, n_chains, (const ChainDev*)dbase, stream))) return rc;
    if (coop_g > 0) {   // did every group stay whole?  (the call synchronises: a broken group has to be reported, not left in the rows)
        std::vector<unsigned> ab(n_chains, 0u);   // (one strided copy for all chains)
        HIP_TRY(hipMemcpy2DAsync(ab.data(), sizeof(unsigned), m->coopbuf.ptr + (size_t)2 * coop_g * cfg.coop_slot_doubles * sizeof(unsigned long long) + coop_g * sizeof(unsigned),
                                 coop_bytes_per_chain, sizeof(unsigned), (size_t)n_chains, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        bool broken = false;
        for (int c = 0; c < n_chains; ++c) broken |= ab[c] != 0u;
        if (broken) {
            // a workgroup of a group did not show up within the wait limit (the chip is shared with another process?): the same solve as
            // plain chains, over whatever the broken groups left in the rows
            note_coop_broken("moshii_chain_solve");
            if (!dev) for (int c = 0; c < n_chains; ++c) { st[c].release(); if (d_shape[c]) hipFree(d_shape[c]); }
            return moshii_chain_solve(m, prior, o, n_chains, chains, (flags & ~0xff00u) | (1u << 8), stream_);
        }
    }
    if (!dev) {
        HIP_TRY(hipStreamSynchronize(stream));
        for (int c = 0; c < n_chains; ++c) {
            const moshii_chain_desc& ch = chains[c];
            const FrameBufs fb{ch.F, ch.attach->M, ch.obs, ch.vis, ch.pose, ch.fullpose, ch.trans, ch.markers_sim, ch.errs, ch.iters, ch.status};
            if ((rc = stage_out(fb, NP, P, &st[c]))) return rc;
            if (d_shape[c]) {
                HIP_TRY(hipMemcpy(ch.shape, d_shape[c], (size_t)ch.F * E * sizeof(double), hipMemcpyDeviceToHost));
                hipFree(d_shape[c]);
            }
        }
    }
    return MOSHII_OK;
}

int moshii_plan_chunks(int32_t F, int32_t num_chunks, int32_t warmup, int32_t cap, int32_t* starts, int32_t* launch_starts) {
    if (F < 0 || num_chunks < 1 || warmup < 0 || cap < 1 || !starts || !launch_starts) return fail(MOSHII_ERR_ARG, "bad argument");
    int C = std::min<int64_t>(num_chunks, std::max(F, 1));
    C = std::min(C, cap);
    for (int c = 0; c < C; ++c) {
        starts[c] = (int32_t)(((int64_t)F * c) / C);                 // balanced: lengths differ by at most one frame
        launch_starts[c] = (c == 0) ? 0 : std::max(0, starts[c] - warmup);
    }
    return C;
}

int moshii_sequence_solve(moshii_model_t m, moshii_prior_t prior, const moshii_solve_opts* o, int32_t n_seq,
                          const moshii_sequence_desc* seqs, const moshii_chunk_opts* co, uint32_t flags, void* stream_,
                          moshii_chunk_report* report) {
    if (!m || !o || !seqs || n_seq < 1) return fail(MOSHII_ERR_ARG, "bad argument");
    hipStream_t stream = (hipStream_t)stream_;
    const bool dev = (flags & MOSHII_BUFFERS_DEVICE) != 0;
    const int NP = m->NP, P = m->P, E = o->n_shape, S = 2 * NP + 5 + E;   // hand-off state: [pose][pose_prev][trans][has_prev][first][shape]
    if (E > 0 && E != m->nshape) return fail(MOSHII_ERR_ARG, "n_shape does not match moshii_model_set_free_shape");
    const int warmup = co ? std::max(0, co->warmup) : 32;
    const double tol = (co && co->verify_tol > 0.0) ? co->verify_tol : 1e-11;
    const bool rejoin = getenv("MOSHII_NO_REJOIN") == nullptr;   // repair chains stop where they re-join the stored trajectory
    int Mmax = 0, Nvmax = 0, NWmax = 1;
    int64_t Ftot = 0;
    for (int q = 0; q < n_seq; ++q) {
        const moshii_sequence_desc& sq = seqs[q];
        if (!sq.attach || sq.attach->model != m || sq.F < 0 || !sq.obs || !sq.vis) return fail(MOSHII_ERR_ARG, "bad sequence descriptor");
        Mmax = std::max(Mmax, sq.attach->M); Nvmax = std::max(Nvmax, sq.attach->Nv); NWmax = std::max(NWmax, sq.attach->NW);
        Ftot += sq.F;
    }
    // ---- chunk plan
    int n_cu = 256;
    { int d = 0; hipGetDevice(&d); hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, d); if (n_cu < 1) n_cu = 256; }
    int want_total = co ? co->num_chunks : 0;   // chunks per sequence when > 0
    struct Chunk { int seq, s, e, a, pred; };
    std::vector<Chunk> chunks;
    {
        const int min_len = std::max(4, warmup / 2);
        const int64_t slots = n_cu;   // one workgroup per CU (see prepare_launch)
        std::vector<int32_t> st, ls;
        for (int q = 0; q < n_seq; ++q) {
            const int F = seqs[q].F;
            int C = want_total;
            if (C <= 0) {   // auto: fill the chip once, but keep chunks at least min_len frames long
                const int64_t share = std::max<int64_t>(1, (slots * F) / std::max<int64_t>(Ftot, 1));
                C = (int)std::max<int64_t>(1, std::min<int64_t>(share, F / min_len));
            }
            st.assign(std::max(C, 1), 0); ls.assign(std::max(C, 1), 0);
            C = moshii_plan_chunks(F, C, warmup, 1 << 20, st.data(), ls.data());
            if (C < 0) return C;
            for (int c = 0; c < C; ++c) {
                Chunk ck; ck.seq = q; ck.s = st[c]; ck.e = (c + 1 < C) ? st[c + 1] : F; ck.a = ls[c];
                ck.pred = (c == 0) ? -1 : (int)chunks.size() - 1;
                chunks.push_back(ck);
            }
        }
    }
    const int NC = (int)chunks.size();
    // ---- launch preparation + control block: [ChainDev x NC (pass 1)][ChainDev x NC (repairs)][pred x NC]
    const size_t extra = 2 * sizeof(ChainDev) * NC + 10 * sizeof(int) * NC + 256;
    LaunchCfg cfg;
    size_t ctl = 0;
    int rc = prepare_launch(m, prior, o, Mmax, Nvmax, NWmax, NC, stream, extra, &cfg, &ctl);
    if (rc) return rc;
    char* dbase = m->scratch.ptr + ctl;
    ChainDev* d_pass1 = (ChainDev*)dbase;
    ChainDev* d_repair = d_pass1 + NC;
    int* d_pred = (int*)(d_repair + NC);
    int* d_bnd = d_pred + NC;   // first frame of every chunk (the boundaries a run-through repair chain crosses)
    int* d_done = d_bnd + NC;   // frames processed per repair chain (diagnostics)
    int* d_baton = d_done + NC; // [2 NC] state / stop request per chunk (ChainDev::baton)
    int* d_abort_at = d_baton + 2 * NC;   // [NC] ChainDev::abort_at (kept across the rounds of this call)
    int* d_fuse = d_abort_at + NC;        // [3 NC] ChainDev::fuse_flags, then [1] ChainDev::fuse_count
    int* d_fuse_count = d_fuse + 3 * NC;
    // Pass-1 chains check their own right-hand hand-off and carry on as the repair chain of the next chunk when it misses
    // (ChainDev::fuse_F): possible when every chunk has a CU of its own for the whole launch, so that the flags they wait on are set.
    // Cooperative repair chains (MOSHII_COOP_GROUP(g) in `flags`, or MOSHII_COOP_REPAIR=g): the sweeps that bound a chunked solve run with g
    // workgroups each (csrc/moshii_dev.h: CoopDev), launched by the host's rounds -- a pass-1 chain cannot recruit CUs, so with them the
    // chains do not carry on inside the first launch.
    int coop_rep = coop_request(flags, "MOSHII_COOP_REPAIR");
    if (coop_rep < -1) return fail(MOSHII_ERR_ARG, "cooperative chains: group size out of range");
    if (coop_rep == -1) {   // the library's choice: as moshii_chain_solve picks it for one chain (prepare_launch)
        const int item_ranks = (Mmax * cfg.ly.nkfmax + MOSHII_TPB - 1) / MOSHII_TPB;
        coop_rep = std::min(MOSHII_COOP_MAXG, item_ranks + ((prior && o->n_body > 0) ? 1 : 0));
        if (coop_rep < 3 || cfg.ly.nmax + 1 > 8 * 16 || cfg.nblk < 4) coop_rep = 0;
    }
    if (coop_rep < 2 || E > 0 || o->n_face > 0 || !rejoin) coop_rep = 0;
    const bool fuse = rejoin && NC <= n_cu && getenv("MOSHII_NO_FUSE") == nullptr && coop_rep == 0;
    const bool trace = getenv("MOSHII_TRACE_REPAIR") != nullptr;
    std::vector<LaunchCfg> coop_cfgs(MOSHII_COOP_MAXG + 1);   // per group size, prepared when first used
    std::vector<char> coop_cfg_ready(MOSHII_COOP_MAXG + 1, 0);
    if (coop_rep >= 2) {
        // the exchange buffers of the repair rounds, once and for the largest round this call can launch (a buffer that grows from
        // round to round is a hipFree + hipMalloc in the middle of the solve)
        const int nb = std::max(4, cfg.nblk), NEc = nb * (nb + 1) / 2, NTc = (NEc + 3) / 4;
        const size_t slot = (size_t)std::max((4 * NTc + 2 * ((NEc + 2) / 2)) * MOSHII_TPB, 3 * Mmax + 2) + 32;
        const size_t per = ((size_t)2 * coop_rep * slot * sizeof(unsigned long long) + (size_t)(2 * coop_rep + 2) * sizeof(unsigned) + 255) & ~size_t(255);
        if ((rc = m->coopbuf.reserve(per * (size_t)std::min(NC, std::max(1, n_cu / 2))))) return rc;
    }
    // device buffers of this call: released on EVERY way out of the function (error returns included)
    struct Owned { std::vector<void*> p; ~Owned() { for (void* q : p) if (q) hipFree(q); } } owned;
    // (kept with the model between calls: three hipMalloc / hipFree pairs a call were ~0.3 ms of a 37 ms step)
    if ((rc = m->handoff.reserve(((size_t)2 * NC * S + NC) * sizeof(double)))) return rc;
    m->handoff.used = true; m->handoff.last_stream = stream;
    double* d_entry = (double*)m->handoff.ptr;
    double* d_final = d_entry + (size_t)NC * S;
    double* d_dev = d_final + (size_t)NC * S;
    // start states of sequences that continue a chain (moshii_sequence_desc.init_*): [pose][pose_prev][trans][has_prev][first = 0]
    double* d_init = nullptr;
    {
        bool any = false;
        for (int q = 0; q < n_seq; ++q) any |= seqs[q].init_pose != nullptr;
        if (any) {
            std::vector<double> hinit((size_t)n_seq * S, 0.0);
            for (int q = 0; q < n_seq; ++q) {
                const moshii_sequence_desc& sq = seqs[q];
                if (!sq.init_pose) continue;
                if (!sq.init_trans) return fail(MOSHII_ERR_ARG, "init_trans is required with init_pose");
                double* h = hinit.data() + (size_t)q * S;
                memcpy(h, sq.init_pose, sizeof(double) * NP);
                if (sq.init_pose_prev) memcpy(h + NP, sq.init_pose_prev, sizeof(double) * NP);
                memcpy(h + 2 * NP, sq.init_trans, sizeof(double) * 3);
                h[2 * NP + 3] = sq.init_pose_prev ? 1.0 : 0.0;
                h[2 * NP + 4] = 0.0;
                if (E > 0 && sq.init_shape) memcpy(h + 2 * NP + 5, sq.init_shape, sizeof(double) * E);
            }
            HIP_TRY(hipMalloc((void**)&d_init, hinit.size() * sizeof(double))); owned.p.push_back(d_init);
            HIP_TRY(hipMemcpy(d_init, hinit.data(), hinit.size() * sizeof(double), hipMemcpyHostToDevice));
        }
    }
    auto cleanup = [&]() {};   // (buffers are released by `owned`)
    std::vector<Staged> st(dev ? 0 : n_seq);
    std::vector<FrameBufs> fbs(n_seq), dbs(n_seq);
    for (int q = 0; q < n_seq; ++q) {
        const moshii_sequence_desc& sq = seqs[q];
        fbs[q] = FrameBufs{sq.F, sq.attach->M, sq.obs, sq.vis, sq.pose, sq.fullpose, sq.trans, sq.markers_sim, sq.errs, sq.iters, sq.status};
        dbs[q] = fbs[q];
        if (!dev) {
            Staged& s = st[q];
            if ((rc = stage_in(fbs[q], NP, P, stream, &s))) { cleanup(); return rc; }
            dbs[q].obs = s.obs; dbs[q].vis = s.vis; dbs[q].pose = s.pose; dbs[q].fullpose = s.fullpose; dbs[q].trans = s.trans;
            dbs[q].msim = s.msim; dbs[q].errs = s.errs; dbs[q].iters = s.iters; dbs[q].status = s.status;
        }
    }
    // extended variant: per-sequence shape rows (device) and per-chain scratch for the shape derivatives of the joint transforms
    std::vector<double*> shp(n_seq, nullptr);
    std::vector<char> shp_owned(n_seq, 0);
    size_t qbytes = 0;
    auto cleanup_shape = [&]() {};   // (released by `owned`)
    if (E > 0) {
        const size_t nfac = (cfg.nblk > 8) ? (size_t)(cfg.ly.nmax + 1) * (cfg.ly.nmax + 2) / 2 + 12 + 256 : 0;
        qbytes = ((size_t)2 * m->K * E * 3 + nfac) * sizeof(double);
        if ((rc = m->qscratch.reserve(qbytes * NC))) { cleanup(); return rc; }
        m->qscratch.used = true; m->qscratch.last_stream = stream;
        for (int q = 0; q < n_seq; ++q) {
            if (dev) { shp[q] = seqs[q].shape; continue; }
            if (seqs[q].F < 1) continue;
            if (hipMalloc((void**)&shp[q], (size_t)seqs[q].F * E * sizeof(double)) != hipSuccess) return fail(MOSHII_ERR_HIP, "hipMalloc failed");
            shp_owned[q] = 1; owned.p.push_back(shp[q]);
            hipMemsetAsync(shp[q], 0, (size_t)seqs[q].F * E * sizeof(double), stream);
        }
    }
    auto make_chain = [&](const Chunk& ck, int from, int idx, bool repair) {
        // a chain over frames [from, ck.e) of its sequence; rows are addressed relative to `from`
        const FrameBufs& b = dbs[ck.seq];
        const size_t M = b.M;
        ChainDev cd;
        memset(&cd, 0, sizeof(cd));
        cd.att = seqs[ck.seq].attach->d_self;
        cd.F = ck.e - from; cd.first = 1; cd.skip = ck.s - from;
        cd.obs = b.obs + (size_t)from * M * 3; cd.vis = b.vis + (size_t)from * M;
        cd.pose = b.pose ? b.pose + (size_t)from * NP : nullptr;
        cd.fullpose = b.fullpose ? b.fullpose + (size_t)from * P : nullptr;
        cd.trans = b.trans ? b.trans + (size_t)from * 3 : nullptr;
        cd.msim = b.msim ? b.msim + (size_t)from * M * 3 : nullptr;
        cd.errs = b.errs ? b.errs + (size_t)from * MOSHII_NERR : nullptr;
        cd.iters = b.iters ? b.iters + (size_t)from * 2 : nullptr;
        cd.status = b.status ? b.status + (size_t)from : nullptr;
        if (E > 0) {
            cd.shape = shp[ck.seq] ? shp[ck.seq] + (size_t)from * E : nullptr;
            cd.qscratch = (double*)(m->qscratch.ptr + qbytes * idx);      // repair chains reuse the slot of their first chunk
        }
        cd.final_state = d_final + (size_t)idx * S;
        // every chain records the state with which it enters its first recorded frame; a repair chain's is the
        // predecessor's end state it was started from, so if that predecessor is itself re-solved later the
        // mismatch shows up in the next verification and this chunk is repaired again
        cd.entry_state = d_entry + (size_t)idx * S;
        if (repair) {
            cd.init_state = d_final + (size_t)ck.pred * S;
            cd.rejoin_tol = rejoin ? tol : 0.0;
        } else if (ck.pred < 0 && d_init != nullptr && seqs[ck.seq].init_pose != nullptr) {
            cd.init_state = d_init + (size_t)ck.seq * S;   // the sequence continues a chain instead of starting one
        }
        return cd;
    };
    std::vector<ChainDev> cds(NC);
    std::vector<int> pred(NC);
    for (int c = 0; c < NC; ++c) { cds[c] = make_chain(chunks[c], chunks[c].a, c, false); pred[c] = chunks[c].pred; }
    if (fuse)
        for (int c = 0; c < NC; ++c) {
            ChainDev& cd = cds[c];
            cd.fuse_flags = d_fuse; cd.fuse_c = c; cd.fuse_count = d_fuse_count; cd.fuse_tol = tol;
            cd.fuse_has_prev = (chunks[c].pred >= 0) ? 1 : 0;
            const bool has_next = c + 1 < NC && chunks[c + 1].seq == chunks[c].seq;
            if (!has_next) continue;
            int cl = c + 1;   // one past the last chunk of the sequence
            while (cl < NC && chunks[cl].seq == chunks[c].seq) ++cl;
            // its own chunk as before, then -- if the hand-off to chunk c + 1 misses -- the repair chain that starts at chunk c + 1
            cd.fuse_F = chunks[c].e - chunks[c].a;
            cd.fuse_has_next = 1;
            cd.F = chunks[cl - 1].e - chunks[c].a;
            cd.final_state = d_final + (size_t)(cl - 1) * S;
            cd.nb = cl - 1 - (c + 1);
            cd.bnd = d_bnd + c + 2;
            cd.bnd_off = chunks[c].a;
            cd.run_final = d_final + (size_t)(c + 1) * S;
            cd.run_entry = d_entry + (size_t)(c + 1) * S;
            cd.baton = d_baton;
            cd.abort_at = d_abort_at;
            cd.chunk0 = c + 1;
            cd.rejoin_tol = tol;
        }
    // The tail of pass 1 (ChainDev::tail_done): with cooperative repair sweeps the first launch's chains do not carry on, so the launch lasts as
    // long as its slowest chunk while the CUs of the others idle.  Once all but a fifth of a chip's worth of chains have ended, a chain with
    // more than a few frames to go gives its chunk up to the sweeps (which re-solve 16 frames in 2.5 ms).  MOSHII_TAIL_CUT=0 switches it off.
    int n_tail_cut_armed = 0;
    {
        static const int tail_env = []{ const char* e = getenv("MOSHII_TAIL_CUT"); return e ? atoi(e) : -1; }();
        const int spare = tail_env > 0 ? tail_env : std::max(8, n_cu / 5);     // (a tenth of the chip: 65.4 k frames/s on the bench's six sequences; a fifth: 65.9 k; none: 63.4 k)
        // Only where it pays: every chunk on a CU of its own from the start (one wave of workgroups: the order of finishing is the order of
        // speed -- with 2 048 chunks the last to finish are simply the last to have started: measured 867 -> 617 k frames/s on the
        // 256-sequence job when armed there) and chunks so short that re-solving one costs less than the wait it ends (a 16-frame chunk:
        // 2.5 ms of a cooperative sweep; the 195-frame chunks of a 50 000-frame sequence: 30 ms -- 372 -> 226 k frames/s when armed).
        int longest = 0;
        for (int c = 0; c < NC; ++c) longest = std::max(longest, chunks[c].e - chunks[c].s);
        if (!fuse && coop_rep >= 2 && rejoin && tail_env != 0 && NC >= 4 * spare && NC <= n_cu && longest <= 32)
            for (int c = 0; c < NC; ++c) {
                ChainDev& cd = cds[c];
                cd.tail_done = d_fuse_count;                // (the carry-on counter: unused without the carry-on protocol, zeroed below)
                cd.tail_quota = NC - spare;
                cd.tail_left = 3;
                cd.tail_can_cut = chunks[c].pred >= 0 ? 1 : 0;
                cd.tail_mark = d_abort_at + c;
                n_tail_cut_armed += cd.tail_can_cut;
            }
    }
    HIP_TRY(hipMemcpyAsync(d_pass1, cds.data(), sizeof(ChainDev) * NC, hipMemcpyHostToDevice, stream));
    {   // the control words behind the descriptors in ONE copy: [pred][chunk starts][done][baton x 2 = 0][abort_at = -1: no mark][fuse flags x 3, fuse count = 0]
        std::vector<int> words((size_t)10 * NC + 1, 0);
        for (int c = 0; c < NC; ++c) { words[c] = pred[c]; words[(size_t)NC + c] = chunks[c].s; words[(size_t)5 * NC + c] = -1; }
        HIP_TRY(hipMemcpyAsync(d_pred, words.data(), sizeof(int) * words.size(), hipMemcpyHostToDevice, stream));
        HIP_TRY(hipStreamSynchronize(stream));   // (`words` goes out of scope)
    }
    if ((rc = launch_chains(cfg, NC, d_pass1, stream))) { cleanup(); return rc; }
    // ---- verify the hand-offs; re-solve (exactly, from the predecessor's final state) the chunks that fail
    std::vector<double> hdev(NC, 0.0);
    int n_repaired = 0, rounds = 0;
    if (fuse) {   // chains that carried on inside the first launch count as repairs (they did a repair chain's work)
        HIP_TRY(hipMemcpyAsync(&n_repaired, d_fuse_count, sizeof(int), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        if (trace) fprintf(stderr, "[moshii] pass 1: %d chains carried on into the next chunk\n", n_repaired);
    }
    double max_dev = 0.0;
    while (true) {
        hipLaunchKernelGGL(k_verify_chunks, dim3(NC), dim3(64), 0, stream, NC, NP, E, d_pred, d_entry, d_final, d_dev);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(hdev.data(), d_dev, sizeof(double) * NC, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        if (rounds == 0) if (const char* dump = getenv("MOSHII_DUMP_HANDOFF")) {   // diagnostics: first-pass hand-off deviations
            if (FILE* fp = fopen(dump, "w")) {
                for (int c = 0; c < NC; ++c) fprintf(fp, "%d %d %d %d %.6e\n", chunks[c].seq, chunks[c].a, chunks[c].s, chunks[c].e, hdev[c]);
                fclose(fp);
            }
        }
        std::vector<char> failing(NC, 0);
        for (int c = 0; c < NC; ++c) failing[c] = chunks[c].pred >= 0 && !(hdev[c] <= tol);
        if (trace && NC <= 16) { fprintf(stderr, "[moshii] hand-off deviations:"); for (int c = 0; c < NC; ++c) fprintf(stderr, " %.1e", hdev[c]); fprintf(stderr, "\n"); }
        for (int c = 0; c < NC; ++c)
            if (chunks[c].pred >= 0 && hdev[c] >= 2e300) {   // NaN in a hand-off state: repairing cannot make it verify
                cleanup_shape(); cleanup();
                return fail(MOSHII_ERR_NUMERIC, "a chunk hand-off state is NaN");
            }
        if (rounds > NC + 1) {   // every round makes at least the first failing hand-off of a sequence exact: NC rounds is the worst case
            cleanup_shape(); cleanup();
            return fail(MOSHII_ERR_NUMERIC, "chunk hand-offs did not verify within the round limit");
        }
        // Scheduling of the repair chains (any mistake here only costs time: whatever ends up inconsistent fails the next
        // verification).  A GROSS miss (> 1e-6) is a chunk whose fresh start sat in another basin: its repair chain may have
        // to run through several chunks before it re-joins, so it owns everything up to the next gross miss.  A SLIGHT
        // miss is a warm-up that had not quite converged: its chain re-joins within a few frames; slight misses that lie in
        // a gross chain's span wait for the next round (they may be swept anyway), the others are repaired right away.
        std::vector<int> todo;
        std::vector<char> todo_gross;
        {
            const double gross_dev = 1e-6;
            // ... except when the slight miss lies far (>= far_frames) behind the start of the gross chain whose span it is in:
            // gross chains re-join within ~140 frames on every sequence looked at, so such a chunk is repaired right away and
            // bounds that chain (should the chain ever get there, the next round continues it).
            static const int far_frames = []{ const char* e = getenv("MOSHII_FAR_FRAMES"); return e ? atoi(e) : 160; }();
            int seq = -1, span_start = 0;
            bool in_span = false;
            for (int c = 0; c < NC; ++c) {
                if (chunks[c].seq != seq) { seq = chunks[c].seq; in_span = false; }
                if (!failing[c]) continue;
                const int p = chunks[c].pred;
                // Chunks a pass-1 chain gave up in the launch's tail (5e299: ChainDev::tail_done) are re-solved like gross misses -- they
                // are the hard stretches, their sweeps run 30-70 frames -- but neither they nor their successors (4e299: the entry state
                // stands against a spoiled end state; the sweep hands over at the boundary) say anything about the chunks behind them:
                // as predecessors they do not hold back a gross miss's chain (they did: a cascade of one round per territory).
                const bool aftercut = hdev[c] == 4e299, p_given_up = failing[p] && hdev[p] >= 1e299 && hdev[p] < 1e300;
                const bool g = rejoin && hdev[c] > gross_dev, pg = rejoin && failing[p] && hdev[p] > gross_dev && !p_given_up;
                if (rejoin && aftercut) continue;
                if (!rejoin) { if (!failing[p]) { todo.push_back(c); todo_gross.push_back(0); } continue; }
                if (g) { if (!pg) { todo.push_back(c); todo_gross.push_back(1); in_span = true; span_start = chunks[c].s; } }
                else if (!in_span || (far_frames > 0 && chunks[c].s - span_start >= far_frames && !failing[p])) { todo.push_back(c); todo_gross.push_back(0); }
            }
        }
        if (todo.empty()) break;
        // Every repair chain starts at its failing chunk and runs on through the following chunks of its sequence until it
        // re-joins the stored trajectory.  Where it reaches the start of another chain of this round, it takes over from it
        // (ChainDev::baton): a cascade of adjacent wrong regions is swept by ONE chain in one round, while regions that turn
        // out to be independent are still repaired side by side.  (Before the baton each chain ended at the next chain's
        // start: a cascade cost one round per region, each as long as the longest chain of that round.)
        std::vector<ChainDev> rep(todo.size());
        std::vector<int> hbaton(2 * (size_t)NC, 0);
        for (size_t i = 0; i < todo.size(); ++i) hbaton[2 * (size_t)todo[i]] = 1;
        for (size_t i = 0; i < todo.size(); ++i) {
            const int c = todo[i];
            int cl = c + 1;   // one past the last chunk this chain may cover: the end of its sequence
            while (cl < NC && chunks[cl].seq == chunks[c].seq) ++cl;
            ChainDev cd = make_chain(chunks[c], chunks[c].s, c, true);
            cd.F = chunks[cl - 1].e - chunks[c].s;
            cd.final_state = d_final + (size_t)(cl - 1) * S;
            cd.nb = cl - 1 - c;
            cd.bnd = d_bnd + c + 1;
            cd.bnd_off = chunks[c].s;
            cd.run_final = d_final + (size_t)c * S;
            cd.run_entry = d_entry + (size_t)c * S;
            cd.baton = d_baton;
            cd.abort_at = d_abort_at;
            cd.chunk0 = c;
            if (!rejoin) { cd.F = chunks[c].e - chunks[c].s; cd.nb = 0; cd.final_state = d_final + (size_t)c * S; cd.baton = nullptr; cd.abort_at = nullptr; }   // (one chunk per chain)
            cd.frames_done = trace ? d_done + i : nullptr;
            rep[i] = cd;
        }
        // cooperative sweeps: as many workgroups per chain as the request and the chip allow (every workgroup of the launch resident)
        int g_round = 0;
        if (coop_rep >= 2) g_round = std::min(coop_rep, n_cu / (int)rep.size());
        const LaunchCfg* use = &cfg;
        if (g_round >= 2) {
            LaunchCfg& cc = coop_cfgs[g_round];
            if (!coop_cfg_ready[g_round]) {
                size_t ctl2 = 0;
                // (n_workgroups = 1: the residency of this round's chains has been settled just above)
                if ((rc = prepare_launch(m, prior, o, Mmax, Nvmax, NWmax, 1, stream, extra, &cc, &ctl2, g_round))) { cleanup(); return rc; }
                if (cc.coop_g != g_round) { cleanup(); return fail(MOSHII_ERR_ARG, "internal: cooperative repair launch not prepared"); }
                if (ctl2 != ctl) { cleanup(); return fail(MOSHII_ERR_ARG, "internal: control block moved"); }
                coop_cfg_ready[g_round] = 1;
            }
            const size_t per = ((size_t)2 * g_round * cc.coop_slot_doubles * sizeof(unsigned long long) + (size_t)(2 * g_round + 2) * sizeof(unsigned) + 255) & ~size_t(255);
            if ((rc = m->coopbuf.reserve(per * rep.size()))) { cleanup(); return rc; }
            m->coopbuf.used = true; m->coopbuf.last_stream = stream;
            HIP_TRY(hipMemsetAsync(m->coopbuf.ptr, 0, per * rep.size(), stream));
            for (size_t i = 0; i < rep.size(); ++i) {
                ChainDev& cd = rep[i];
                cd.coop.G = g_round; cd.coop.prior_rank = cc.coop_prior_rank; cd.coop.slot_doubles = cc.coop_slot_doubles; cd.coop.skew = coop_skew_env();
                    coop_split(seqs[chunks[todo[i]].seq].attach->M, g_round, cc.coop_prior_frac, cd.coop.mlo);
                char* cb = m->coopbuf.ptr + per * i;
                cd.coop.slots = as_gp_rw((unsigned long long*)cb);
                cd.coop.flags = as_gp_rw((unsigned*)(cb + (size_t)2 * g_round * cc.coop_slot_doubles * sizeof(unsigned long long)));
            }
            use = &cc;
        }
        HIP_TRY(hipMemcpyAsync(d_baton, hbaton.data(), sizeof(int) * hbaton.size(), hipMemcpyHostToDevice, stream));
        HIP_TRY(hipMemcpyAsync(d_repair, rep.data(), sizeof(ChainDev) * rep.size(), hipMemcpyHostToDevice, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        if ((rc = launch_chains(*use, (int)rep.size(), d_repair, stream))) { cleanup(); return rc; }
        if (g_round >= 2) {   // did every group stay whole?
            const size_t per = ((size_t)2 * g_round * use->coop_slot_doubles * sizeof(unsigned long long) + (size_t)(2 * g_round + 2) * sizeof(unsigned) + 255) & ~size_t(255);
            std::vector<unsigned> ab(rep.size(), 0u);   // (one strided copy: a copy per chain was 22 us a chain, 0.5-1 ms a round)
            HIP_TRY(hipMemcpy2DAsync(ab.data(), sizeof(unsigned), m->coopbuf.ptr + (size_t)2 * g_round * use->coop_slot_doubles * sizeof(unsigned long long) + g_round * sizeof(unsigned),
                                     per, sizeof(unsigned), rep.size(), hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipStreamSynchronize(stream));
            bool broken = false;
            for (size_t i = 0; i < rep.size(); ++i) broken |= ab[i] != 0u;
            if (broken) {   // (the chip is shared?)  Whatever the broken groups left is caught by the next verification -- they spoil the entry state of
                            // the chunk they stopped in -- and re-solved by plain chains from here on
                note_coop_broken("moshii_sequence_solve");
                coop_rep = 0;
            }
        }
        n_repaired += (int)rep.size();
        ++rounds;
        if (trace) {
            std::vector<int> done(rep.size(), -1);
            HIP_TRY(hipStreamSynchronize(stream));
            HIP_TRY(hipMemcpy(done.data(), d_done, sizeof(int) * rep.size(), hipMemcpyDeviceToHost));
            fprintf(stderr, "[moshii] repair round %d: %zu chains (chunk:dev:frames_done/limit)", rounds, rep.size());
            for (size_t i = 0; i < rep.size(); ++i) fprintf(stderr, " %d:%.0e:%d/%d", todo[i], hdev[todo[i]], done[i], rep[i].F);
            fprintf(stderr, "\n");
        }
    }
    for (int c = 0; c < NC; ++c) if (chunks[c].pred >= 0) max_dev = std::max(max_dev, hdev[c]);
    if (report) { report->n_chunks = NC; report->n_repaired = n_repaired; report->repair_rounds = rounds; report->max_handoff_dev = max_dev;
                  report->warmup = warmup; report->verify_tol = tol; }
    if (!dev)
        for (int q = 0; q < n_seq; ++q)
            if ((rc = stage_out(fbs[q], NP, P, &st[q]))) { cleanup_shape(); cleanup(); return rc; }
    if (!dev && E > 0) {
        hipStreamSynchronize(stream);
        for (int q = 0; q < n_seq; ++q)
            if (shp[q] && seqs[q].shape) hipMemcpy(seqs[q].shape, shp[q], (size_t)seqs[q].F * E * sizeof(double), hipMemcpyDeviceToHost);
    }
    cleanup_shape();
    cleanup();
    return MOSHII_OK;
}

int moshii_stagei_solve(moshii_model_t m, moshii_prior_t prior, const moshii_stagei_desc* desc, void* stream) {
    if (!m || !desc) return fail(MOSHII_ERR_ARG, "stagei: null model or desc");
    if (!desc->faces || !desc->marker_vids || !desc->m2b || !desc->wt_init || !desc->n_obs || !desc->obs_ids || !desc->obs ||
        !desc->annealing || !desc->pose_ids) return fail(MOSHII_ERR_ARG, "stagei: missing input array");
    S1ModelView mv;
    mv.V = m->V; mv.K = m->K; mv.NB = m->NB; mv.NP = m->NP; mv.body_dof = m->body_dof; mv.hand_dof = m->hand_dof;
    mv.parents = m->d_parents; mv.anc = m->d_anc; mv.vt = m->d_vt; mv.shapedirs = m->d_shapedirs; mv.posedirs = m->d_posedirs;
    mv.weights = m->d_weights; mv.Jreg = m->d_Jreg; mv.hands_mean = m->d_hands_mean; mv.comps = m->d_comps;
    S1PriorView pv;
    if (prior) { pv.G = prior->G; pv.npose = prior->npose; pv.means = prior->d_means; pv.chols = prior->d_chols; pv.neglogw = prior->d_neglogw; }
    char err[256] = {0};
    int rc = moshii_stagei_core(&mv, prior ? &pv : nullptr, desc, stream, err, sizeof(err));
    if (rc != MOSHII_OK) return fail(rc, err);
    return MOSHII_OK;
}

int moshii_last_launch_info(char* kernel_name, int32_t name_cap, int32_t* lds_bytes, int32_t* block_threads) {
    if (kernel_name && name_cap > 0) { strncpy(kernel_name, g_last.name.c_str(), name_cap - 1); kernel_name[name_cap - 1] = 0; }
    if (lds_bytes) *lds_bytes = g_last.lds;
    if (block_threads) *block_threads = g_last.threads;
    return MOSHII_OK;
}

}  // extern "C"

// accessors for lbs_forward.hip (keeps the handle layout private to this file)
extern "C" {
int moshii_internal_model_dims(moshii_model_t m, int* V, int* K) { *V = m->V; *K = m->K; return 0; }
const double* moshii_internal_vsh(moshii_model_t m) { return m->d_vsh; }
const double* moshii_internal_posedirs(moshii_model_t m) { return m->d_posedirs; }
const double* moshii_internal_weights(moshii_model_t m) { return m->d_weights; }
const double* moshii_internal_J(moshii_model_t m) { return m->d_J; }
const double* moshii_internal_weights_host(moshii_model_t m) { return m->weights_host.data(); }
void* moshii_internal_l32(moshii_model_t m) { return &m->l32; }
void moshii_internal_l32_set_valid(moshii_model_t m, int v) { m->l32_valid = v != 0; }
}
