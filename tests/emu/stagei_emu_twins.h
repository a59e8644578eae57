// TEST INFRASTRUCTURE.  Sequential restatements of the three Stage-I kernels whose GPU form is cooperative (LDS tiles shared by 256
// threads, wavefront shuffles) and therefore cannot run in the one-"thread"-per-block emulation build of moshpp_amd/csrc/stagei.hip
// (tests/emu/build_emu.py).  Included by that file under S1_EMU only, after its macro layer (KERNEL_LB, SHARED, SYNC, TID, NT) and
// S1_T / S1_PB are defined; the product library never sees this header.
#ifndef STAGEI_EMU_TWINS_H
#define STAGEI_EMU_TWINS_H

// one S1_T x S1_T tile of A = J^T J (k_s1_syrk)
static inline void s1_emu_syrk_tile(const double* Jm, int R, int n, int ldn, double* A, int ti, int tj) {
    for (int i = ti * S1_T; i < std::min(n, (ti + 1) * S1_T); ++i) for (int j = tj * S1_T; j < std::min(n, (tj + 1) * S1_T); ++j) {
        double s = 0;
        for (int r = 0; r < R; ++r) s += Jm[(size_t)r * ldn + i] * Jm[(size_t)r * ldn + j];
        A[(size_t)i * n + j] = s; A[(size_t)j * n + i] = s;
    }
}

// one S1_PB x S1_PB tile of the trailing update A_ik -= L_i L_k^T (k_s1_chol_update)
static inline void s1_emu_chol_update_tile(double* A, int n, int j0, int jb, int ti, int tj) {
    const int s0 = j0 + jb;
    for (int i = s0 + ti * S1_PB; i < std::min(n, s0 + (ti + 1) * S1_PB); ++i) for (int k = s0 + tj * S1_PB; k < std::min(n, s0 + (tj + 1) * S1_PB); ++k) {
        if (k > i) continue;
        double sacc = 0;
        for (int c = 0; c < jb; ++c) sacc += A[(size_t)i * n + j0 + c] * A[(size_t)k * n + j0 + c];
        A[(size_t)i * n + k] -= sacc;
    }
}

// factor + invert the diagonal block of a panel (k_s1_chol_diag)
KERNEL_LB(64) k_s1_chol_diag(double* A, int n, int j0, double* dinv, int* status) {
    SHARED double D[S1_PB][S1_PB + 1];
    const int jb = (n - j0) < S1_PB ? (n - j0) : S1_PB;
    for (int e = TID; e < S1_PB * S1_PB; e += NT) { int r = e / S1_PB, c = e % S1_PB; D[r][c] = (r < jb && c <= r) ? A[(size_t)(j0 + r) * n + j0 + c] : (r == c ? 1.0 : 0.0); }
    SYNC();
    for (int c = 0; c < jb; ++c) {
        if (TID == 0) {
            double v = D[c][c];
            if (!(v > 0)) { status[1] = 1; v = 1.0; }
            D[c][c] = sqrt(v);
        }
        SYNC();
        const double ip = 1.0 / D[c][c];
        for (int r = c + 1 + TID; r < jb; r += NT) D[r][c] *= ip;
        SYNC();
        // rank-1 update of the rows below: lane -> (row, column parity), no integer division
        for (int t = TID; t < 2 * S1_PB; t += NT) {
            const int r = t % S1_PB, h = t / S1_PB;
            if (r > c && r < jb) {
                const double lrc = D[r][c];
                for (int k = c + 1 + h; k <= r; k += 2) D[r][k] -= lrc * D[k][c];
            }
        }
        SYNC();
    }
    for (int e = TID; e < jb * jb; e += NT) { int r = e / jb, c = e % jb; if (c <= r) A[(size_t)(j0 + r) * n + j0 + c] = D[r][c]; }
    // inverse of the (padded, unit-extended) 32 x 32 factor, one column per thread: L x = e_c
    double* Di = dinv + (size_t)(j0 / S1_PB) * S1_PB * S1_PB;
    SHARED double X[S1_PB][S1_PB + 1];
    for (int c = TID; c < S1_PB; c += NT) {
        for (int r = 0; r < S1_PB; ++r) {
            double s0 = (r == c) ? 1.0 : 0.0, s1 = 0, s2 = 0, s3 = 0;        // four independent chains hide the LDS latency
            int k = c;
            for (; k + 3 < r; k += 4) {
                s0 -= D[r][k] * X[k][c]; s1 -= D[r][k + 1] * X[k + 1][c]; s2 -= D[r][k + 2] * X[k + 2][c]; s3 -= D[r][k + 3] * X[k + 3][c];
            }
            for (; k < r; ++k) s0 -= D[r][k] * X[k][c];
            X[r][c] = (r < c) ? 0.0 : ((s0 + s1) + (s2 + s3)) / D[r][r];
        }
        for (int r = 0; r < S1_PB; ++r) Di[r * S1_PB + c] = X[r][c];
    }
}


#endif
