// Device-pointer views handed to the Stage-I solver (stagei.hip).  No HIP types here: tests/emu compiles stagei.hip with g++.
#pragma once

struct S1ModelView {
    int V, K, NB, NP, body_dof, hand_dof;
    const int* parents;                 // [K]
    const unsigned long long* anc;      // [K] bit j set iff k is j or an ancestor of j
    const double *vt, *shapedirs, *posedirs, *weights, *Jreg;   // [V][3], [V][3][NB], [V][3][9(K-1)], [V][K], [K][V]
    const double *hands_mean, *comps;   // [3K - body_dof] or null, [hand_dof][3K - body_dof] or null
};

struct S1PriorView {
    int G, npose;
    const double *means, *chols, *neglogw;    // [G][npose], [G][npose][npose] (lower, L L^T = precision), [G]
};

struct moshii_stagei_desc;
int moshii_stagei_core(const S1ModelView* mv, const S1PriorView* pv, const moshii_stagei_desc* ds, void* stream, char* err, int errlen);
