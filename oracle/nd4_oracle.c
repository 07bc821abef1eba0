/* nd4_oracle.c — see nd4_oracle.h.   *** TEST INFRASTRUCTURE, NOT PRODUCT CODE ***
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -shared -fPIC (oracle/Makefile).
 * Parity pinned against tests/golden (reference-generated) by tests/test_oracle_golden.py.
 */
#include "nd4_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ generator */
static inline uint32_t fmix32(uint32_t h) {
  h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16; return h;
}
double nd4o_uniform(uint32_t seed, uint32_t idx) {
  uint32_t hi = fmix32(idx ^ fmix32(seed));
  uint32_t lo = fmix32(hi + 0x9E3779B9u + idx);
  double m = (double)(hi >> 5) * 67108864.0 + (double)(lo >> 6);
  return m * 0x1p-52 - 1.0;
}
void nd4o_fill_uniform(uint32_t seed, uint32_t offset, int64_t n, double* out) {
  for (int64_t i = 0; i < n; i++) out[i] = nd4o_uniform(seed, offset + (uint32_t)i);
}

/* ------------------------------------------------------------------ matmul
 * matmul.js:49-53: for i: for k: for j: C[i,j] += A[i,k]*B[k,j]  (C zero-initialised, k ascending) */
static void matmul_ikj(int64_t I, int64_t K, int64_t J, const double* A, const double* B, double* C) {
  for (int64_t i = 0; i < I; i++) {
    double* c = C + i * J;
    for (int64_t j = 0; j < J; j++) c[j] = 0.0;
    for (int64_t k = 0; k < K; k++) {
      const double a = A[i * K + k];
      const double* b = B + k * J;
      for (int64_t j = 0; j < J; j++) c[j] += a * b[j];
    }
  }
}
void nd4o_matmul_batched(int64_t batch, int64_t I, int64_t K, int64_t J,
                         const double* A, int64_t sA, const double* B, int64_t sB, double* C) {
  for (int64_t b = 0; b < batch; b++) matmul_ikj(I, K, J, A + b * sA, B + b * sB, C + b * I * J);
}
int nd4o_matmul2(int ndimA, const int32_t* shA, const double* A,
                 int ndimB, const int32_t* shB, const double* B, int32_t* shC, double* C) {
  const int64_t I = shA[ndimA - 2], K = shA[ndimA - 1], J = shB[ndimB - 1];
  if (shB[ndimB - 2] != K) return -1;                          /* matmul.js:101-102 */
  const int nd = ndimA > ndimB ? ndimA : ndimB, nb = nd - 2;
  int32_t shape[32]; int64_t strA[32], strB[32];
  if (nd > 32) return -3;
  for (int d = 0; d < nb; d++) shape[d] = 1;
  /* matmul.js:110-116 common (broadcast) shape */
  for (int w = 0; w < 2; w++) {
    const int nda = w ? ndimB : ndimA; const int32_t* sh = w ? shB : shA;
    for (int i = nb, j = nda - 2; i-- > 0 && j-- > 0;) {
      if (shape[i] == 1) shape[i] = sh[j];
      else if (shape[i] != sh[j] && sh[j] != 1) return -2;
    }
  }
  /* element strides of the leading axes, 0 where the operand broadcasts */
  for (int w = 0; w < 2; w++) {
    const int nda = w ? ndimB : ndimA; const int32_t* sh = w ? shB : shA; int64_t* st = w ? strB : strA;
    int64_t s = w ? K * J : I * K;
    for (int i = nb - 1; i >= 0; i--) {
      const int j = i - nd + nda;
      if (j < 0) { st[i] = 0; continue; }
      st[i] = sh[j] > 1 ? s : 0; s *= sh[j];
    }
  }
  int64_t batch = 1; for (int d = 0; d < nb; d++) batch *= shape[d];
  int32_t idx[32] = {0};
  for (int64_t b = 0; b < batch; b++) {
    int64_t a = 0, bb = 0;
    for (int d = 0; d < nb; d++) { a += idx[d] * strA[d]; bb += idx[d] * strB[d]; }
    matmul_ikj(I, K, J, A + a, B + bb, C + b * I * J);
    for (int d = nb - 1; d >= 0; d--) { if (++idx[d] < shape[d]) break; idx[d] = 0; }
  }
  if (shC) { for (int d = 0; d < nb; d++) shC[d] = shape[d]; shC[nb] = (int32_t)I; shC[nb + 1] = (int32_t)J; }
  return 0;
}

/* ------------------------------------------------------------------ Givens helpers */
/* _giv_rot.js:22-37 */
void nd4o_giv_rot_qr(double a, double b, double* c, double* s, double* norm) {
  const double fa = fabs(a), fb = fabs(b);
  const double mx = fa > fb ? fa : fb;        /* Math.max; NaN falls through to the 0===max test as in JS? (NaN!==0) */
  if (0.0 == mx) { *c = 1; *s = 0; *norm = 0; return; }
  a /= mx; b /= mx;
  double n = sqrt(a * a + b * b);
  a /= n; b /= n; n *= mx;
  *c = a; *s = b; *norm = n;
}
/* _giv_rot.js:42-67: W_i' = c*W_i + s*W_j ; W_j' = c*W_j - s*W_i over n contiguous elements */
static inline void giv_rot_rows(double* W, int64_t n, int64_t i, int64_t j, double c, double s) {
  for (int64_t k = 0; k < n; k++) {
    const double wi = W[i + k], wj = W[j + k];
    W[i + k] = c * wi + s * wj;
    W[j + k] = c * wj - s * wi;
  }
}
/* _giv_rot.js:72-87: column rotation on an N x N matrix: W_i' = c*W_i - s*W_j ; W_j' = c*W_j + s*W_i */
static inline void giv_rot_cols(double* W, int64_t N, int64_t i, int64_t j, double c, double s) {
  for (int64_t k = 0; k < N; k++, i += N, j += N) {
    const double wi = W[i], wj = W[j];
    W[i] = c * wi - s * wj;
    W[j] = c * wj + s * wi;
  }
}
/* transpose_inplace.js:21-30 */
static void transpose_inplace(int64_t N, double* A) {
  for (int64_t i = 0; i < N - 1; i++)
    for (int64_t j = i + 1; j < N; j++) { double t = A[N * i + j]; A[N * i + j] = A[N * j + i]; A[N * j + i] = t; }
}

/* ------------------------------------------------------------------ QR */
void nd4o_qr_decomp_full(int64_t batch, int64_t M, int64_t N, const double* A, double* Q, double* R) {
  const int64_t B = 8;                                           /* qr.js:33 (64/8 for float64) */
  memcpy(R, A, sizeof(double) * batch * M * N);                  /* qr.js:37 */
  memset(Q, 0, sizeof(double) * batch * M * M);
  for (int64_t b = 0; b < batch; b++) {
    double* q = Q + b * M * M; double* r = R + b * M * N;
    for (int64_t i = 0; i < M; i++) q[M * i + i] = 1.0;          /* qr.js:52 */
    for (int64_t J = 0; J < N; J += B)                           /* qr.js:54-57 blocked order */
      for (int64_t I = J; I < M; I += B)
        for (int64_t i = I; i < I + B && i < M; i++)
          for (int64_t j = J; j < J + B && j < N && j < i; j++) {
            const int64_t ij = N * i + j, jj = N * j + j;
            const double R_ij = r[ij]; if (0.0 == R_ij) continue;            /* :60 */
            double c, s, norm; nd4o_giv_rot_qr(r[jj], R_ij, &c, &s, &norm);  /* :61-62 */
            r[ij] = 0.0; if (0.0 == s) continue;                             /* :63 */
            r[jj] = norm;                                                    /* :64 */
            giv_rot_rows(r, N - 1 - j, jj + 1, ij + 1, c, s);                /* :65-66 */
            giv_rot_rows(q, 1 + i, M * j, M * i, c, s);                      /* :67-68 */
          }
    transpose_inplace(M, q);                                     /* qr.js:70 */
  }
}

void nd4o_qr_decomp(int64_t batch, int64_t Nr, int64_t Mc, const double* A, double* Q, double* R) {
  /* reference naming (qr.js:88): [N,M] = rows, cols */
  const int64_t N = Nr, M = Mc;
  if (N <= M) { nd4o_qr_decomp_full(batch, N, M, A, Q, R); return; }        /* qr.js:91 */
  memcpy(Q, A, sizeof(double) * batch * N * M);                              /* qr.js:93 */
  memset(R, 0, sizeof(double) * batch * M * M);
  for (int64_t b = 0; b < batch; b++) {
    double* q = Q + b * N * M; double* r = R + b * M * M;
    for (int64_t i = 1; i < N; i++) {                                        /* qr.js:104-120 */
      const int64_t I = i < M ? i : M;
      for (int64_t j = 0; j < I; j++) {
        const int64_t ij = M * i + j, jj = M * j + j;
        const double R_ij = q[ij]; if (0.0 == R_ij) continue;
        double c, s, norm; nd4o_giv_rot_qr(q[jj], R_ij, &c, &s, &norm);
        if (s != 0.0) {
          if (c < 0) { c *= -1; s *= -1; norm *= -1; }                       /* :111-115 */
          giv_rot_rows(q, M - 1 - j, jj + 1, ij + 1, c, s);
          q[jj] = norm;
        }
        q[ij] = s;
      }
    }
    for (int64_t i = 0; i < M; i++)                                          /* qr.js:123-127 */
      for (int64_t j = i; j < M; j++) { r[M * i + j] = q[M * i + j]; q[M * i + j] = (i == j) ? 1.0 : 0.0; }
    for (int64_t i = N; --i > 0;) {                                          /* qr.js:130-138 */
      const int64_t I = i < M ? i : M;
      for (int64_t j = I; j-- > 0;) {
        const double s = q[M * i + j]; if (0.0 == s) continue;
        q[M * i + j] = 0.0;
        const double c = sqrt((1 - s) * (1 + s));
        giv_rot_rows(q, M - j, M * i + j, M * j + j, c, s);
      }
    }
  }
}

/* ------------------------------------------------------------------ LU (lu.js:24-81) */
void nd4o_lu_decomp(int64_t batch, int64_t N, const double* A, double* LU, int32_t* P) {
  memcpy(LU, A, sizeof(double) * batch * N * N);
  for (int64_t b = 0; b < batch; b++) {
    double* lu = LU + b * N * N; int32_t* p = P + b * N;
    for (int64_t i = 0; i < N; i++) p[i] = (int32_t)i;
    for (int64_t i = 0; i < N; i++) {
      double* row_i = lu + i * N;
      int64_t piv = i;                                                       /* :48-52 first strict max */
      for (int64_t j = i + 1; j < N; j++)
        if (fabs(lu[N * j + i]) > fabs(lu[N * piv + i])) piv = j;
      if (i != piv) {                                                        /* :54-62 */
        int32_t t = p[i]; p[i] = p[piv]; p[piv] = t;
        double* row_p = lu + piv * N;
        for (int64_t j = 0; j < N; j++) { double tmp = row_i[j]; row_i[j] = row_p[j]; row_p[j] = tmp; }
      }
      for (int64_t j = i + 1; j < N; j++) {                                  /* :65-73 */
        double* row_j = lu + j * N;
        const double scale = row_j[i] / row_i[i];
        row_j[i] = scale;
        for (int64_t k = i + 1; k < N; k++) row_j[k] -= scale * row_i[k];
      }
    }
  }
}

/* ------------------------------------------------------------------ triangular solves (tri.js) */
static void tril_solve1(int64_t M, int64_t N, int64_t O, const double* L, double* X) {        /* tri.js:61-70 */
  for (int64_t i = 0; i < M; i++) {
    for (int64_t k = 0; k < i; k++)
      for (int64_t j = 0; j < O; j++) X[O * i + j] -= L[N * i + k] * X[O * k + j];
    for (int64_t j = 0; j < O; j++) X[O * i + j] /= L[N * i + i];
  }
}
static void triu_solve1(int64_t M, int64_t N, int64_t O, const double* U, double* X) {        /* tri.js:87-94 */
  for (int64_t i = M; i-- > 0;)
    for (int64_t j = O; j-- > 0;) {
      for (int64_t k = M; --k > i;) X[O * i + j] -= U[N * i + k] * X[O * k + j];
      X[O * i + j] /= U[N * i + i];
    }
}
void nd4o_tril_solve(int64_t batch, int64_t M, int64_t O, const double* L, int64_t sL, double* X) {
  for (int64_t b = 0; b < batch; b++) tril_solve1(M, M, O, L + b * sL, X + b * M * O);
}
void nd4o_triu_solve(int64_t batch, int64_t M, int64_t O, const double* U, int64_t sU, double* X) {
  for (int64_t b = 0; b < batch; b++) triu_solve1(M, M, O, U + b * sU, X + b * M * O);
}
void nd4o_lu_solve(int64_t batch, int64_t N, int64_t J, const double* LU, int64_t sLU, const int32_t* P, int64_t sP,
                   const double* Y, int64_t sY, double* X) {
  for (int64_t b = 0; b < batch; b++) {
    const double* lu = LU + b * sLU; const int32_t* p = P + b * sP; const double* y = Y + b * sY; double* x = X + b * N * J;
    for (int64_t i = 0; i < N; i++)                                          /* lu.js:131-136 */
      for (int64_t j = 0; j < J; j++) x[J * i + j] = y[J * p[i] + j];
    for (int64_t i = 0; i < N; i++)                                          /* lu.js:139-142 */
      for (int64_t j = 0; j < J; j++)
        for (int64_t k = 0; k < i; k++) x[i * J + j] -= lu[N * i + k] * x[k * J + j];
    triu_solve1(N, N, J, lu, x);                                             /* lu.js:145 */
  }
}

/* ------------------------------------------------------------------ Cholesky (cholesky.js) */
/* src/la/cholesky.js:27-48 _cholesky_decomp with the Kahan accumulator of src/kahan_sum.js:22-46, preceded by the copy of
 * the lower triangle into a zeroed L (:63-68). Returns -1 where the reference throws (NaN pivot :43-44, NaN input via
 * KahanSum.set). */
int nd4o_cholesky_decomp(int64_t batch, int64_t N, const double* S, double* L) {
  for (int64_t b = 0; b < batch; b++) {
    const double* s = S + b * N * N; double* l = L + b * N * N;
    for (int64_t e = 0; e < N * N; e++) l[e] = 0.0;
    for (int64_t i = 0; i < N; i++)
      for (int64_t j = 0; j <= i; j++) l[N * i + j] = s[N * i + j];
    for (int64_t i = 0; i < N; i++)
      for (int64_t j = 0; j <= i; j++) {
        double sum = l[N * i + j], rst = 0.0;                      /* kahan.set */
        if (isnan(sum)) return -1;
        for (int64_t k = 0; k < j; k++) {                          /* kahan.add(-L_ik L_jk) */
          const double val = -l[N * i + k] * l[N * j + k];
          const double cor = val - rst, t = sum + cor;
          rst = (t - sum) - cor;
          sum = t;
        }
        if (i > j) l[N * i + j] = sum / l[N * j + j];
        else {
          l[N * i + i] = sqrt(sum);
          if (isnan(l[N * i + i])) return -1;
        }
      }
  }
  return 0;
}

static void tril_t_solve1(int64_t M, int64_t N, int64_t O, const double* L, double* X) {      /* tri.js:115-124 */
  for (int64_t k = M; k-- > 0;) {
    for (int64_t j = O; j-- > 0;) X[O * k + j] /= L[N * k + k];
    for (int64_t i = k; i-- > 0;)
      for (int64_t j = O; j-- > 0;) X[O * i + j] -= L[N * k + i] * X[O * k + j];
  }
}
/* src/la/cholesky.js:74-150 cholesky_solve core (:117-123): copy y, _tril_solve, _tril_t_solve */
void nd4o_cholesky_solve(int64_t batch, int64_t N, int64_t J, const double* L, int64_t sL, const double* Y, int64_t sY, double* X) {
  for (int64_t b = 0; b < batch; b++) {
    double* x = X + b * N * J; const double* y = Y + b * sY;
    for (int64_t e = 0; e < N * J; e++) x[e] = y[e];
    tril_solve1(N, N, J, L + b * sL, x);
    tril_t_solve1(N, N, J, L + b * sL, x);
  }
}

/* ------------------------------------------------------------------ LDL^T (ldl.js) */
/* src/la/ldl.js:47-64 _ldl_decomp after the lower-triangle copy of ldl_decomp (:82-87): packed LD, unit-L below the
 * diagonal, D on it, zeros above. No pivoting, no singularity check (a zero pivot propagates Inf/NaN like the reference). */
void nd4o_ldl_decomp(int64_t batch, int64_t N, const double* S, double* LD) {
  for (int64_t b = 0; b < batch; b++) {
    const double* s = S + b * N * N; double* ld = LD + b * N * N;
    for (int64_t e = 0; e < N * N; e++) ld[e] = 0.0;
    for (int64_t i = 0; i < N; i++)
      for (int64_t j = 0; j <= i; j++) ld[N * i + j] = s[N * i + j];
    for (int64_t j = 0; j < N; j++) {
      for (int64_t k = 0; k < j; k++) {
        const double V_k = ld[N * j + k] * ld[N * k + k];
        for (int64_t i = j; i < N; i++) ld[N * i + j] -= ld[N * i + k] * V_k;
      }
      for (int64_t i = j; ++i < N;) ld[N * i + j] /= ld[N * j + j];
    }
  }
}
/* src/la/ldl.js:93-130 _ldl_solve behind ldl_solve (:133-201): forward (unit L), scaling by D, backward (L^T) */
void nd4o_ldl_solve(int64_t batch, int64_t N, int64_t J, const double* LD, int64_t sLD, const double* Y, int64_t sY, double* X) {
  for (int64_t b = 0; b < batch; b++) {
    const double* ld = LD + b * sLD; const double* y = Y + b * sY; double* x = X + b * N * J;
    for (int64_t e = 0; e < N * J; e++) x[e] = y[e];
    for (int64_t i = 1; i < N; i++)
      for (int64_t k = 0; k < i; k++)
        for (int64_t j = 0; j < J; j++) x[J * i + j] -= ld[N * i + k] * x[J * k + j];
    for (int64_t i = 0; i < N; i++)
      for (int64_t j = 0; j < J; j++) x[J * i + j] /= ld[N * i + i];
    for (int64_t k = N; k-- > 1;)
      for (int64_t i = k; i-- > 0;)
        for (int64_t j = J; j-- > 0;) x[J * i + j] -= ld[N * k + i] * x[J * k + j];
  }
}

/* ------------------------------------------------------------------ Hessenberg (hessenberg.js) */
/* src/la/norm.js:22-67 FrobeniusNorm: scaled sum of squares (include / resultIncl) */
typedef struct { double sum, max; } fro_t;
static void fro_include(fro_t* f, double x) {
  x = fabs(x);
  if (x != 0) {
    if (f->max < x) { const double s = f->max / x; f->sum *= s * s; f->max = x; }
    x /= f->max;
    f->sum += x * x;
  }
}
static double fro_result_incl(const fro_t* f, double x) {
  x = fabs(x);
  double sum = f->sum, max = f->max;
  if (x != 0) {
    if (max < x) { const double s = max / x; sum *= s * s; max = x; }
    x /= max;
    sum += x * x;
  }
  return isfinite(max) ? sqrt(sum) * max : max;
}
/* src/la/hessenberg.js:27-86 _hessenberg_decomp on one matrix: A = U H U^T, H upper Hessenberg; rows are finished from
 * the bottom up with Householder reflectors acting on the leading columns; U (zero on entry) uses its last row as scratch. */
void nd4o_hessenberg_decomp(int64_t N, double* U, double* H) {
  for (int64_t i = N - 1; i-- > 0;) U[N * i + i] = 1.0;
  const int64_t lastRow = N * (N - 1);
  for (int64_t i = N; --i > 1;) {
    const int64_t rowI = N * i, ii = rowI + (i - 1);
    fro_t nrm = {0.0, 0.0};
    for (int64_t j = i - 1; j-- > 0;) fro_include(&nrm, H[rowI + j]);
    if (nrm.max == 0) continue;
    const double norm = fro_result_incl(&nrm, H[ii]) * (H[ii] > 0 ? -1 : +1);
    H[ii] -= norm;
    fro_include(&nrm, H[ii]);
    const double max = nrm.max, div = sqrt(nrm.sum);
    for (int64_t j = i; j-- > 0;) H[rowI + j] = H[rowI + j] / max * 1.4142135623730951 / div;     /* Math.SQRT2 */
    for (int64_t j = i; j-- > 0;) {                                   /* right of H */
      double sum = 0;
      for (int64_t k = i; k-- > 0;) sum += H[N * j + k] * H[rowI + k];
      for (int64_t k = i; k-- > 0;) H[N * j + k] -= H[rowI + k] * sum;
    }
    for (int64_t k = 0; k < N; k++) U[lastRow + k] = 0.0;              /* left of H */
    for (int64_t j = i; j-- > 0;)
      for (int64_t k = N; k-- > 0;) U[lastRow + k] += H[N * j + k] * H[rowI + j];
    for (int64_t j = i; j-- > 0;)
      for (int64_t k = N; k-- > 0;) H[N * j + k] -= H[rowI + j] * U[lastRow + k];
    for (int64_t j = N - 1; j-- > 0;) {                               /* right of U */
      double sum = 0;
      for (int64_t k = i; k-- > 0;) sum += U[N * j + k] * H[rowI + k];
      for (int64_t k = i; k-- > 0;) U[N * j + k] -= H[rowI + k] * sum;
    }
    for (int64_t k = rowI; k < ii; k++) H[k] = 0.0;
    H[ii] = norm;
  }
  for (int64_t k = lastRow; k < N * N - 1; k++) U[k] = 0.0;
  U[N * N - 1] = 1;
}

/* ------------------------------------------------------------------ bidiagonalisation (bidiag.js) */
static void transpose_inplace_sq(int64_t N, double* A) {                 /* transpose_inplace.js:21-30 */
  for (int64_t i = 0; i < N; i++)
    for (int64_t j = i + 1; j < N; j++) { const double t = A[N * i + j]; A[N * i + j] = A[N * j + i]; A[N * j + i] = t; }
}
/* Householder that leaves only entry `first` of row[first .. first+len-1] (bidiag.js:66-76 / :125-135 / :182-195): the
 * normalised vector (|v| = 1) overwrites v[first..]; returns 0 if there was nothing to eliminate (NORM.max === 0). */
static int bidiag_row_householder(const double* row, double* v, int64_t first, int64_t end, double* norm_out) {
  fro_t nrm = {0.0, 0.0};
  for (int64_t j = end; --j > first;) fro_include(&nrm, row[j]);
  if (nrm.max == 0) return 0;
  const double norm = fro_result_incl(&nrm, row[first]) * (row[first] > 0 ? -1 : +1);
  double head = row[first] - norm;
  fro_include(&nrm, head);
  const double max = nrm.max, div = sqrt(nrm.sum);
  for (int64_t j = first; j < end; j++) v[j] = (j == first ? head : row[j]) / max / div;
  *norm_out = norm;
  return 1;
}
/* X[j, first..end) -= 2 (X[j,:] . v) v for rows j in [j0, j1): "apply householder to right of ..." */
static void bidiag_apply_right(double* X, int64_t ld, int64_t j0, int64_t j1, const double* v, int64_t first, int64_t end) {
  for (int64_t j = j0; j < j1; j++) {
    double sum = 0;
    for (int64_t k = first; k < end; k++) sum += X[ld * j + k] * v[k];
    sum *= 2;
    for (int64_t k = first; k < end; k++) X[ld * j + k] -= v[k] * sum;
  }
}
/* src/la/bidiag.js:113-161 _bidiag_decomp_square: U, V zero on entry, B = A on entry */
static void bidiag_square1(int64_t N, double* U, double* B, double* V, double* tmp) {
  for (int64_t i = N; i-- > 0;) { U[N * i + i] = 1; V[N * i + i] = 1; }
  for (int64_t i = 0; i < N - 1; i++) {
    const int64_t ii = N * i + i;
    for (int64_t j = i; ++j < N;) {
      const int64_t ji = N * j + i;
      const double B_ji = B[ji]; if (B_ji == 0) continue;
      double c, s, norm; nd4o_giv_rot_qr(B[ii], B_ji, &c, &s, &norm);
      B[ji] = 0; if (s == 0) continue;
      B[ii] = norm;
      giv_rot_rows(B, N - 1 - i, ii + 1, ji + 1, c, s);
      giv_rot_rows(U, 1 + j, N * i, N * j, c, s);
    }
    double norm;
    if (!bidiag_row_householder(B + N * i, tmp, i + 1, N, &norm)) continue;
    for (int64_t j = i + 1; j < N; j++) B[N * i + j] = tmp[j];
    bidiag_apply_right(V, N, 0, N, B + N * i, i + 1, N);
    bidiag_apply_right(B, N, i + 1, N, B + N * i, i + 1, N);
    B[ii + 1] = norm;
    for (int64_t k = ii + 2; k < N * (i + 1); k++) B[k] = 0.0;
  }
  transpose_inplace_sq(N, V);
  transpose_inplace_sq(N, U);
}
/* src/la/bidiag.js:32-110 _bidiag_decomp_vert (M >= N): U [M,N] = A on entry, B, V [N,N] zero on entry */
static void bidiag_vert1(int64_t M, int64_t N, double* U, double* B, double* V, double* tmp) {
  for (int64_t i = N; i-- > 0;) V[N * i + i] = 1;
  for (int64_t i = 0; i < N; i++) {
    const int64_t ii = N * i + i;
    for (int64_t j = i; ++j < M;) {
      const int64_t ji = N * j + i;
      const double B_ji = U[ji]; if (B_ji == 0) continue;
      double c, s, norm; nd4o_giv_rot_qr(U[ii], B_ji, &c, &s, &norm);
      if (s != 0) {
        if (c < 0) { c *= -1; s *= -1; norm *= -1; }
        giv_rot_rows(U, N - 1 - i, ii + 1, ji + 1, c, s);
        U[ii] = norm;
      }
      U[ji] = s;
    }
    if (i < N - 2) {
      double norm;
      if (!bidiag_row_householder(U + N * i, tmp, i + 1, N, &norm)) continue;
      for (int64_t j = i + 1; j < N; j++) U[N * i + j] = tmp[j];
      bidiag_apply_right(V, N, 0, N, U + N * i, i + 1, N);
      bidiag_apply_right(U, N, i + 1, M, U + N * i, i + 1, N);
      U[ii + 1] = norm;
      for (int64_t k = ii + 2; k < N * (i + 1); k++) U[k] = 0.0;
    }
  }
  transpose_inplace_sq(N, V);
  for (int64_t i = N; --i > 0;) { B[N * i + i] = U[N * i + i]; B[N * (i - 1) + i] = U[N * (i - 1) + i]; }
  B[0] = U[0];
  for (int64_t i = 0; i < N; i++)
    for (int64_t j = i; j < N; j++) U[N * i + j] = (i == j) ? 1.0 : 0.0;
  for (int64_t i = N; i-- > 0;)
    for (int64_t j = M; --j > i;) {
      const double s = U[N * j + i]; if (s == 0) continue;
      U[N * j + i] = 0;
      const double c = sqrt((1 + s) * (1 - s));
      giv_rot_rows(U, N - i, N * j + i, N * i + i, c, s);
    }
}
/* src/la/bidiag.js:164-242 _bidiag_decomp_horiz (M < N): U [M,M], B [M,M+1] zero on entry; V [(M+1),N] holds A in its
 * rows 1..M on entry (bidiag.js:288-291) */
static void bidiag_horiz1(int64_t M, int64_t N, double* U, double* B, double* Vbuf) {
  double* V = Vbuf + N;                                              /* V_off + N */
  for (int64_t i = M; i-- > 0;) U[M * i + i] = 1;
  for (int64_t i = 0; i < M && i < N - 1; i++) {
    const int64_t ii = N * i + i;
    for (int64_t j = i; ++j < M;) {
      const int64_t ji = N * j + i;
      const double B_ji = V[ji]; if (B_ji == 0) continue;
      double c, s, norm; nd4o_giv_rot_qr(V[ii], B_ji, &c, &s, &norm);
      V[ji] = 0; if (s == 0) continue;
      V[ii] = norm;
      giv_rot_rows(V, N - 1 - i, ii + 1, ji + 1, c, s);
      giv_rot_rows(U, 1 + j, M * i, M * j, c, s);
    }
    double* above = V + N * (i - 1);                                 /* householder goes to row (i-1) of the shifted buffer */
    fro_t nrm = {0.0, 0.0};
    for (int64_t j = N - 1;;) {
      const double V_j = above[j] = V[N * i + j]; if (--j <= i) break;
      fro_include(&nrm, V_j);
    }
    if (nrm.max == 0) { above[i + 1] = 0; continue; }
    const double norm = fro_result_incl(&nrm, above[i + 1]) * (above[i + 1] > 0 ? -1 : +1);
    above[i + 1] -= norm;
    fro_include(&nrm, above[i + 1]);
    const double max = nrm.max, div = sqrt(nrm.sum);
    for (int64_t j = i; ++j < N;) above[j] = above[j] / max / div;
    bidiag_apply_right(V, N, i + 1, M, above, i + 1, N);
    V[ii + 1] = norm;
  }
  transpose_inplace_sq(M, U);
  for (int64_t i = M; i-- > 0;) {
    B[(M + 1) * i + (i + 1)] = i < N - 1 ? V[N * i + (i + 1)] : 0;
    B[(M + 1) * i + i] = V[N * i + i];
  }
  for (int64_t i = (M < N - 1 ? M : N - 1);;) {
    --i;
    for (int64_t k = 0; k < N; k++) V[N * i + k] = 0.0;
    V[N * i + (i + 1)] = 1;
    if (i < 0) break;
    const double* above = V + N * (i - 1);
    bidiag_apply_right(V, N, i, M, above, i + 1, N);
  }
}
/* src/la/bidiag.js:245-319 bidiag_decomp on one matrix: A [M,N] -> U [M,I], B [I,J], V [J,N], I = min(M,N), J = I (M >= N)
 * or I+1 (M < N). tmp: N doubles. */
void nd4o_bidiag_decomp(int64_t M, int64_t N, const double* A, double* U, double* B, double* V, double* tmp) {
  if (M == N) {
    for (int64_t e = 0; e < N * N; e++) { B[e] = A[e]; U[e] = 0.0; V[e] = 0.0; }
    bidiag_square1(N, U, B, V, tmp);
  } else if (M > N) {
    for (int64_t e = 0; e < M * N; e++) U[e] = A[e];
    for (int64_t e = 0; e < N * N; e++) { B[e] = 0.0; V[e] = 0.0; }
    bidiag_vert1(M, N, U, B, V, tmp);
  } else {
    for (int64_t e = 0; e < N; e++) V[e] = 0.0;
    for (int64_t e = 0; e < M * N; e++) V[N + e] = A[e];
    for (int64_t e = 0; e < M * M; e++) U[e] = 0.0;
    for (int64_t e = 0; e < M * (M + 1); e++) B[e] = 0.0;
    bidiag_horiz1(M, N, U, B, V);
  }
}

/* src/la/qr.js:146-183 _qr_decomp_inplace: Givens elimination of A (M x N) in place, the same rotations applied to the
 * rows of Y (M x L). The bundle /root/reference/dist/nd.js does not export this function, so it is pinned through the
 * reference's own test oracle (qr_test.js:213-225): A == R and Y == Q^T Y of qr_decomp_full. */
void nd4o_qr_decomp_inplace(int64_t M, int64_t N, int64_t L, double* A, double* Y) {
  for (int64_t i = 1; i < M; i++)
    for (int64_t j = 0; j < N && j < i; j++) {
      const int64_t ij = N * i + j, jj = N * j + j;
      const double A_ij = A[ij];
      if (A_ij == 0.0) continue;
      double c, s, norm;
      nd4o_giv_rot_qr(A[jj], A_ij, &c, &s, &norm);
      A[ij] = 0.0;
      if (s == 0.0) continue;
      A[jj] = norm;
      giv_rot_rows(A, N - 1 - j, jj + 1, ij + 1, c, s);
      giv_rot_rows(Y, L, L * j, L * i, c, s);
    }
}

/* src/la/qr.js:186-273 qr_lstsq core (:232-241): x[0:L] = Q^T y accumulated k-innermost, then _triu_solve(L,I,J) */
void nd4o_qr_lstsq(int64_t batch, int64_t N, int64_t M, int64_t I, int64_t J, const double* Q, int64_t sQ, const double* R, int64_t sR,
                   const double* Y, int64_t sY, double* X) {
  const int64_t L = M < I ? M : I;
  for (int64_t b = 0; b < batch; b++) {
    const double* q = Q + b * sQ; const double* r = R + b * sR; const double* y = Y + b * sY; double* x = X + b * I * J;
    for (int64_t e = 0; e < I * J; e++) x[e] = 0.0;
    for (int64_t i = 0; i < L; i++)
      for (int64_t j = 0; j < J; j++)
        for (int64_t k = 0; k < N; k++) x[i * J + j] += q[k * M + i] * y[k * J + j];
    triu_solve1(L, I, J, r, x);
  }
}

/* src/la/svd.js:100-228 svd_lstsq core (:165-201). Returns -1 on a non-finite singular value (:171-172). */
int nd4o_svd_lstsq(int64_t batch, int64_t N, int64_t M, int64_t I, int64_t J, const double* U, int64_t sU, const double* SV, int64_t sSv,
                   const double* V, int64_t sV, const double* Y, int64_t sY, double* X, double* tmp /* M*J */) {
  const double EPS = 1.4901161193847656e-08;                /* Math.sqrt(2^-52), exact */
  for (int64_t b = 0; b < batch; b++) {
    const double* u = U + b * sU; const double* sv = SV + b * sSv; const double* v = V + b * sV; const double* y = Y + b * sY;
    double* x = X + b * I * J;
    const double T = EPS * fabs(sv[0]);
    int64_t rank = M;
    for (int64_t r = 0; r < M; r++) {
      const double sv_r = fabs(sv[r]);
      if (!isfinite(sv_r)) return -1;
      if (sv_r <= T) { rank = r; break; }
    }
    for (int64_t e = 0; e < M * J; e++) tmp[e] = 0.0;
    for (int64_t e = 0; e < I * J; e++) x[e] = 0.0;
    for (int64_t k = 0; k < N; k++)
      for (int64_t i = 0; i < rank; i++)
        for (int64_t j = 0; j < J; j++) tmp[J * i + j] += u[M * k + i] * y[J * k + j];
    for (int64_t i = 0; i < rank; i++)
      for (int64_t j = 0; j < J; j++) tmp[J * i + j] /= sv[i];
    for (int64_t k = 0; k < rank; k++)
      for (int64_t i = 0; i < I; i++)
        for (int64_t j = 0; j < J; j++) x[J * i + j] += v[I * k + i] * tmp[J * k + j];
  }
  return 0;
}

/* ------------------------------------------------------------------ two-sided Jacobi SVD */
/* _svd_jac_utils.js:72-114 */
static void svd_jac_angles(double S_pp, double S_pq, double S_qp, double S_qq,
                           double* pca, double* psa, double* pcb, double* psb) {
  double x = atan2(S_qp - S_pq, S_qq + S_pp),
         y = atan2(S_qp + S_pq, S_qq - S_pp);
  const double a = (x - y) / 2, b = (x + y) / 2;
  double ca = cos(a), sa = sin(a), cb = cos(b), sb = sin(b);
  x = cb * (sa * S_qp + ca * S_pp) - sb * (sa * S_qq + ca * S_pq);
  y = sb * (ca * S_qp - sa * S_pp) + cb * (ca * S_qq - sa * S_pq);
  if (fabs(x) < fabs(y)) {
    double t = sa; sa = ca; ca = -t;            /* [sa,ca] = [ca,-sa] */
    t = cb; cb = sb; sb = -t;                   /* [cb,sb] = [sb,-cb] */
    x = y;
  }
  if (x < 0) { cb = -cb; sb = -sb; }
  *pca = ca; *psa = sa; *pcb = cb; *psb = sb;
}

/* stable merge sort of indices by descending key (ties keep ascending index; the reference's
 * Int32Array.sort((i,j)=>sv[j]-sv[i]) leaves tie order to the engine, _svd_jac_utils.js:158) */
static void sort_desc(int64_t n, const double* key, int32_t* ord, int32_t* tmp) {
  for (int64_t w = 1; w < n; w *= 2) {
    for (int64_t lo = 0; lo < n; lo += 2 * w) {
      int64_t mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n, i = lo, j = mid, k = lo;
      while (i < mid && j < hi) tmp[k++] = (key[ord[j]] > key[ord[i]]) ? ord[j++] : ord[i++];
      while (i < mid) tmp[k++] = ord[i++];
      while (j < hi) tmp[k++] = ord[j++];
    }
    memcpy(ord, tmp, sizeof(int32_t) * n);
  }
}

/* _svd_jac_utils.js:123-188 */
static void svd_jac_post(int64_t N, double* U, const double* S, double* V, double* sv, int32_t* ord, int32_t* tmp) {
  for (int64_t i = 0; i < N; i++) sv[i] = S[N * i + i];                      /* 1) */
  for (int64_t i = N; i-- > 0;) {                                            /* 2) make positive */
    const double s = sv[i];
    if (s < 0 || (s == 0 && signbit(s))) {
      sv[i] = -s;
      for (int64_t j = 0; j < N; j++) U[N * i + j] *= -1;
    }
  }
  for (int64_t i = 0; i < N; i++) ord[i] = (int32_t)i;                       /* 3) sort */
  sort_desc(N, sv, ord, tmp);
  for (int64_t i = 0; i < N; ++i)
    for (int64_t j = i;;) {
      int64_t t = ord[j]; ord[j] = (int32_t)j; j = t;
      if (j <= i) break;
      double* uI = U + (int64_t)ord[j] * N; double* uJ = U + j * N;
      double* vI = V + (int64_t)ord[j] * N; double* vJ = V + j * N;
      for (int64_t k = 0; k < N; k++) { double x = uI[k]; uI[k] = uJ[k]; uJ[k] = x; }
      for (int64_t k = 0; k < N; k++) { double x = vI[k]; vI[k] = vJ[k]; vJ[k] = x; }
      double x = sv[ord[j]]; sv[ord[j]] = sv[j]; sv[j] = x;
    }
  transpose_inplace(N, U);                                                   /* 4) */
}

int nd4o_svd_jac_2sided(int64_t batch, int64_t N, const double* A, double* U, double* sv, double* V) {
  if (N == 1) {                                                              /* svd_jac_2sided.js:66-78 */
    for (int64_t i = 0; i < batch; i++) {
      const double a = A[i];
      if (a < 0.0) { sv[i] = -a; U[i] = -1.0; } else { sv[i] = a; U[i] = 1.0; }
      V[i] = 1.0;
    }
    return 0;
  }
  const double eps = 0x1p-52, TOL = (N * eps) * (N * eps);                   /* :57 */
  const int64_t B = 8;
  double* S = (double*)malloc(sizeof(double) * N * N);
  int32_t* ord = (int32_t*)malloc(sizeof(int32_t) * 2 * N);
  int sweeps_max = 0;
  for (int64_t b = 0; b < batch; b++) {
    double* u = U + b * N * N; double* v = V + b * N * N;
    memcpy(S, A + b * N * N, sizeof(double) * N * N);
    for (int64_t i = 0; i < N; i++) for (int64_t j = 0; j < N; j++) u[N * i + j] = v[N * i + j] = (i == j);
    int sweeps = 0;
    for (int finished = 0; !finished;) {                                     /* :95-134 */
      finished = 1; sweeps++;
      for (int64_t Q = 0; Q < N; Q += B)
        for (int64_t P = 0; P <= Q; P += B)
          for (int64_t q = Q; q < Q + B && q < N; q++)
            for (int64_t p = P; p < P + B && p < q; p++) {
              const double S_pp = S[N * p + p], S_pq = S[N * p + q], S_qp = S[N * q + p], S_qq = S[N * q + q];
              if (!(S_pq * S_pq + S_qp * S_qp > fabs(S_pp * S_qq) * TOL)) continue;   /* :112 */
              finished = 0;
              double ca, sa, cb, sb; svd_jac_angles(S_pp, S_pq, S_qp, S_qq, &ca, &sa, &cb, &sb);
              giv_rot_rows(S, N, N * p, N * q, ca, sa);
              giv_rot_cols(S, N, p, q, cb, sb);
              S[N * p + q] = S[N * q + p] = 0.0;
              giv_rot_rows(u, N, N * p, N * q, ca, sa);
              giv_rot_rows(v, N, N * p, N * q, cb, -sb);
            }
    }
    if (sweeps > sweeps_max) sweeps_max = sweeps;
    svd_jac_post(N, u, S, v, sv + b * N, ord, ord + N);
  }
  free(S); free(ord);
  return sweeps_max;
}

/* ------------------------------------------------------------------ svd_decomp = svd_dc (bidiagonalisation + divide & conquer) */
#include "nd4_oracle_svd_dc.c"
