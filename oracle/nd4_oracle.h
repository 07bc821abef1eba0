/* nd4_oracle — CPU restatement of the nd4js `nd.la` hot path.   *** TEST INFRASTRUCTURE ***
 *
 * This library is the parity CHECKER and the `cpu_baseline` ("port") of bench.py. It is never
 * linked, imported or called by the product path (libnd4hip.so, package nd4js_amd): only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 *
 * Every function restates one reference function (file:line relative to /root/reference/),
 * keeping its exact floating-point operation order (plain IEEE fp64, no FMA: build with
 * -ffp-contract=off), so that matmul / LU / Givens-QR are bit-identical to the JS reference.
 * Parity is PINNED: tests/test_oracle_golden.py checks it against tests/golden (.npy files), which were
 * produced by the real reference bundle (oracle/gen_golden.js).
 */
#ifndef ND4_ORACLE_H
#define ND4_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* counter-based input generator shared by every language in this repo (not reference code) */
double nd4o_uniform(uint32_t seed, uint32_t idx);
void   nd4o_fill_uniform(uint32_t seed, uint32_t offset, int64_t n, double* out);

/* src/la/matmul.js:31-74 (matmul2_RR hot loop :49-53) with the broadcast odometer :44-70.
 * shapes are full NDArray shapes (ndim >= 2); C must hold prod(broadcast shape). Returns 0, or
 * -1 inner-dim mismatch, -2 not broadcast-compatible (matmul.js:95-116). shapeC (size
 * max(ndimA,ndimB)) is written when non-NULL. */
int nd4o_matmul2(int ndimA, const int32_t* shapeA, const double* A,
                 int ndimB, const int32_t* shapeB, const double* B,
                 int32_t* shapeC, double* C);
/* flat batched form: C[b] = A[b*strideA] * B[b*strideB]  (stride 0 = broadcast), i-k-j order */
void nd4o_matmul_batched(int64_t batch, int64_t I, int64_t K, int64_t J,
                         const double* A, int64_t strideA, const double* B, int64_t strideB, double* C);

/* src/la/_giv_rot.js:22-37 */
void nd4o_giv_rot_qr(double a, double b, double* c, double* s, double* norm);

/* src/la/qr.js:27-77  qr_decomp_full: A[batch,M,N] -> Q[batch,M,M], R[batch,M,N] */
void nd4o_qr_decomp_full(int64_t batch, int64_t M, int64_t N, const double* A, double* Q, double* R);
/* src/la/qr.js:80-145 qr_decomp (economic): Q[batch,M,min(M,N)], R[batch,min(M,N),N] */
void nd4o_qr_decomp(int64_t batch, int64_t M, int64_t N, const double* A, double* Q, double* R);

/* src/la/lu.js:24-81  LU[batch,N,N], P[batch,N] (permutation vector: A[P[i],:] = (L*U)[i,:]) */
void nd4o_lu_decomp(int64_t batch, int64_t N, const double* A, double* LU, int32_t* P);

/* src/la/tri.js:45-71 (_tril_solve) / :74-95 (_triu_solve) on [batch] M x M triangles (non-unit diagonal) and
 * [batch] M x O right-hand sides, in place on X (X is initialised with Y by the caller). strideT = 0 broadcasts T. */
void nd4o_tril_solve(int64_t batch, int64_t M, int64_t O, const double* L, int64_t strideL, double* X);
void nd4o_triu_solve(int64_t batch, int64_t M, int64_t O, const double* U, int64_t strideU, double* X);
/* src/la/lu.js:84-177 lu_solve core (:130-147): X = Y[P,:], forward substitution with the unit-lower part,
 * _triu_solve with the upper part. strides in elements, 0 = broadcast. X [batch, N, J]. */
void nd4o_lu_solve(int64_t batch, int64_t N, int64_t J, const double* LU, int64_t strideLU, const int32_t* P, int64_t strideP,
                   const double* Y, int64_t strideY, double* X);

/* src/la/cholesky.js:51-71 cholesky_decomp (kernel :27-48, Kahan sums): S [batch,N,N] (lower triangle read) -> L, upper
 * part zero. Returns -1 where the reference throws. */
int nd4o_cholesky_decomp(int64_t batch, int64_t N, const double* S, double* L);
/* src/la/cholesky.js:74-150 cholesky_solve core: X = L^-T L^-1 Y (strides in doubles, 0 = broadcast) */
void nd4o_cholesky_solve(int64_t batch, int64_t N, int64_t J, const double* L, int64_t strideL, const double* Y, int64_t strideY, double* X);

/* src/la/ldl.js:67-90 ldl_decomp (kernel :47-64): S [batch,N,N] (lower triangle read) -> packed LD */
void nd4o_ldl_decomp(int64_t batch, int64_t N, const double* S, double* LD);
/* src/la/ldl.js:133-201 ldl_solve core (:93-130): X = L^-T D^-1 L^-1 Y (strides in doubles, 0 = broadcast) */
void nd4o_ldl_solve(int64_t batch, int64_t N, int64_t J, const double* LD, int64_t strideLD, const double* Y, int64_t strideY, double* X);

/* src/la/hessenberg.js:27-86 on one matrix: U [N,N] zero on entry, H [N,N] = A on entry; A = U H U^T on exit */
void nd4o_hessenberg_decomp(int64_t N, double* U, double* H);

/* src/la/bidiag.js:245-319 bidiag_decomp (kernels :32-242) on one matrix: A [M,N] -> U [M,I], B [I,J] upper bidiagonal,
 * V [J,N] with A = U B V; I = min(M,N), J = I (M >= N) or I+1 (M < N). tmp: N doubles. */
void nd4o_bidiag_decomp(int64_t M, int64_t N, const double* A, double* U, double* B, double* V, double* tmp);

/* src/la/qr.js:146-183 _qr_decomp_inplace on one matrix: A [M,N] <- R, Y [M,L] <- Q^T Y */
void nd4o_qr_decomp_inplace(int64_t M, int64_t N, int64_t L, double* A, double* Y);

/* src/la/qr.js:186-273 qr_lstsq core: Q [N,M], R [M,I], Y [N,J] -> X [I,J] (strides in doubles, 0 = broadcast) */
void nd4o_qr_lstsq(int64_t batch, int64_t N, int64_t M, int64_t I, int64_t J, const double* Q, int64_t strideQ, const double* R, int64_t strideR,
                   const double* Y, int64_t strideY, double* X);
/* src/la/svd.js:100-228 svd_lstsq core: U [N,M], sv [M], V [M,I], Y [N,J] -> X [I,J]; tmp = M*J doubles.
 * Returns -1 if a singular value is NaN/Inf (the reference throws). */
int nd4o_svd_lstsq(int64_t batch, int64_t N, int64_t M, int64_t I, int64_t J, const double* U, int64_t strideU, const double* SV, int64_t strideSv,
                   const double* V, int64_t strideV, const double* Y, int64_t strideY, double* X, double* tmp);

/* src/la/svd_jac_2sided.js:30-144 (square input only; the rectangular pre-reduction :42-52 is
 * host-side composition) + _svd_jac_utils.js:72-114 (angles), :123-188 (post-processing).
 * U[batch,N,N], sv[batch,N], V[batch,N,N] (rows of V = right singular vectors). Returns sweeps. */
int nd4o_svd_jac_2sided(int64_t batch, int64_t N, const double* A, double* U, double* sv, double* V);

/* src/la/svd_dc.js:883-932 svd_dc (= nd.la.svd_decomp, svd.js:25; kernels :37-880 + bidiag.js:164-242), restated in
 * nd4_oracle_svd_dc.c: A [batch,M,N] -> U [batch,M,L], sv [batch,L], V [batch,L,N], L = min(M,N). Returns 0, the line number of
 * the reference assertion that failed, or -1 (out of memory). */
int nd4o_svd_dc(int64_t batch, int64_t M, int64_t N, const double* A, double* U, double* sv, double* V);

#ifdef __cplusplus
}
#endif
#endif
