/*
 * oracle_fast.c -- blocked, OpenMP-parallel Cholesky for the CPU oracle (TEST INFRASTRUCTURE ONLY).
 *
 * Same contract as orc_potrf_lower() (arma::chol(S,"lower"), src/gpirtMCMC.cpp:17,78,97) but
 * blocked right-looking so the n = 4096..8192 parity cases and bench.py's cpu_baseline leg finish
 * in seconds.  Validated against the unblocked restatement in tests/test_oracle_stages.py.
 * Compiled with -O3 -mavx2 -mfma -fopenmp (rounding differs from the unblocked code at the ulp
 * level, as between any two valid LAPACK builds).
 */
#include "gpirt_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define NB 96

/* unblocked factor of the nb x nb diagonal block stored with leading dimension lda */
static int potf2_block(double* A, int64_t lda, int nb)
{
    for (int j = 0; j < nb; ++j) {
        double* cj = A + (int64_t)j * lda;
        for (int k = 0; k < j; ++k) {
            const double* ck = A + (int64_t)k * lda;
            double ljk = ck[j];
            for (int i = j; i < nb; ++i) cj[i] -= ljk * ck[i];
        }
        double d = cj[j];
        if (!(d > 0.0)) return j + 1;
        d = sqrt(d);
        cj[j] = d;
        double inv = 1.0 / d;
        for (int i = j + 1; i < nb; ++i) cj[i] *= inv;
    }
    return 0;
}

/* C[0:mr, 0:nc] -= P[0:mr, 0:kb] * Q[0:nc, 0:kb]^T ; all column-major with leading dims */
static void gemm_nt_sub(double* restrict C, int64_t ldc, const double* restrict P, int64_t ldp,
                        const double* restrict Q, int64_t ldq, int64_t mr, int nc, int kb)
{
    int c = 0;
    for (; c + 4 <= nc; c += 4) {
        double* c0 = C + (int64_t)(c + 0) * ldc;
        double* c1 = C + (int64_t)(c + 1) * ldc;
        double* c2 = C + (int64_t)(c + 2) * ldc;
        double* c3 = C + (int64_t)(c + 3) * ldc;
        for (int p = 0; p < kb; ++p) {
            const double* pp = P + (int64_t)p * ldp;
            const double q0 = Q[c + 0 + (int64_t)p * ldq], q1 = Q[c + 1 + (int64_t)p * ldq];
            const double q2 = Q[c + 2 + (int64_t)p * ldq], q3 = Q[c + 3 + (int64_t)p * ldq];
#pragma omp simd
            for (int64_t i = 0; i < mr; ++i) {
                double v = pp[i];
                c0[i] -= v * q0; c1[i] -= v * q1; c2[i] -= v * q2; c3[i] -= v * q3;
            }
        }
    }
    for (; c < nc; ++c) {
        double* c0 = C + (int64_t)c * ldc;
        for (int p = 0; p < kb; ++p) {
            const double* pp = P + (int64_t)p * ldp;
            const double q0 = Q[c + (int64_t)p * ldq];
#pragma omp simd
            for (int64_t i = 0; i < mr; ++i) c0[i] -= pp[i] * q0;
        }
    }
}

int orc_potrf_lower_blocked(double* A, int64_t n, int nthreads)
{
    int info = 0;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
    for (int64_t k0 = 0; k0 < n && !info; k0 += NB) {
        int nb = (int)((n - k0) < NB ? (n - k0) : NB);
        double* Akk = A + k0 + k0 * n;
        int bi = potf2_block(Akk, n, nb);
        if (bi) { info = (int)(k0 + bi); break; }
        int64_t r0 = k0 + nb;          /* first trailing row */
        int64_t mr = n - r0;
        if (mr <= 0) break;
        /* panel: X * Lkk^T = A[r0:, k0:k0+nb]  (row chunks are independent) */
        const int64_t RC = 256;
#pragma omp parallel for schedule(dynamic)
        for (int64_t rc = 0; rc < mr; rc += RC) {
            int64_t rows = (mr - rc) < RC ? (mr - rc) : RC;
            double* X = A + r0 + rc + k0 * n;
            for (int j = 0; j < nb; ++j) {
                double* xj = X + (int64_t)j * n;
                for (int p = 0; p < j; ++p) {
                    const double* xp = X + (int64_t)p * n;
                    double l = Akk[j + (int64_t)p * n];
#pragma omp simd
                    for (int64_t i = 0; i < rows; ++i) xj[i] -= xp[i] * l;
                }
                double inv = 1.0 / Akk[j + (int64_t)j * n];
#pragma omp simd
                for (int64_t i = 0; i < rows; ++i) xj[i] *= inv;
            }
        }
        /* trailing update, lower blocks only: A[I, J] -= P[I,:] P[J,:]^T for I >= J */
        int64_t nblk = (mr + NB - 1) / NB;
#pragma omp parallel for schedule(dynamic)
        for (int64_t jb = 0; jb < nblk; ++jb) {
            int64_t c0 = r0 + jb * NB;
            int nc = (int)((n - c0) < NB ? (n - c0) : NB);
            const int64_t RB = 192;
            for (int64_t rr = c0; rr < n; rr += RB) {
                int64_t rows = (n - rr) < RB ? (n - rr) : RB;
                gemm_nt_sub(A + rr + c0 * n, n, A + rr + k0 * n, n, A + c0 + k0 * n, n, rows, nc, nb);
            }
        }
    }
    if (!info)
        for (int64_t j = 1; j < n; ++j) memset(A + j * n, 0, sizeof(double) * (size_t)j);
    return info;
}
