// extern "C" entry points of include/nd4hip.h: argument validation and the *_dev (device pointer) forms. The host-pointer
// forms that the N-API shim binds live in nd4hip_host.hip.
#include "nd4hip_internal.h"
#include <cmath>

namespace {

constexpr size_t D = sizeof(double);

// The kernels carry the batch in gridDim.y/z (<= 65535): longer batches run in chunks of ND4_CHUNK matrices. `b0` is the
// first matrix of the chunk, `nb` its length; strides that are 0 (broadcast operand) stay 0.
constexpr int64_t ND4_CHUNK = 32768;
#define ND4_FOR_CHUNKS(batch) for (int64_t b0 = 0, nb = 0; (nb = ((batch) - b0 < ND4_CHUNK ? (batch) - b0 : ND4_CHUNK)) > 0; b0 += nb)

// algorithmic flop conventions of SURVEY.md 8(d) for nd4hip_profile_last
inline double nd4_flops_qr(int64_t M, int64_t N) {          // with explicit Q: 4 (M N^2 - N^3 / 3) for M >= N, 8/3 N^3 when square
  const double m = (double)M, n = (double)N;
  return M >= N ? 4.0 * (m * n * n - n * n * n / 3.0) : 2.0 * n * m * m - 2.0 / 3.0 * m * m * m + 4.0 / 3.0 * m * m * m;
}
inline double nd4_flops_svd(int64_t M, int64_t N) {         // Golub-Reinsch count with U, sv, V: 21 N^3 when square
  const double m = (double)(M >= N ? M : N), n = (double)(M >= N ? N : M);
  return 4.0 * m * m * n + 8.0 * m * n * n + 9.0 * n * n * n;
}

}  // namespace

// ------------------------------------------------------------------------------------ matmul
extern "C" int nd4hip_dgemm_batched_dev(nd4hip_handle* h, int64_t batch, int64_t I, int64_t K, int64_t J,
                                        const double* A, int64_t strideA, const double* B, int64_t strideB, double* C) {
  ND4_CHECK_ARG(h != nullptr, "nd4hip_dgemm_batched: NULL handle");
  Nd4DeviceGuard guard(h);
  Nd4Prof prof(h, "dgemm_batched", (double)(2.0 * batch * I * K * J), (double)(8.0 * (double)(batch * I * J + (strideA ? batch : 1) * I * K + (strideB ? batch : 1) * K * J)));
  ND4_CHECK_ARG(batch >= 0 && I >= 0 && K >= 0 && J >= 0, "nd4hip_dgemm_batched: negative extent");
  ND4_CHECK_ARG(strideA == 0 || strideA >= I * K, "nd4hip_dgemm_batched: strideA must be 0 or >= I*K");
  ND4_CHECK_ARG(strideB == 0 || strideB >= K * J, "nd4hip_dgemm_batched: strideB must be 0 or >= K*J");
  if (batch == 0 || I == 0 || J == 0) return 0;
  ND4_CHECK_ARG(A && B && C, "nd4hip_dgemm_batched: NULL matrix pointer");
  for (int64_t b0 = 0; b0 < batch; b0 += 32768) {             // gridDim.y limit
    const int64_t nb = batch - b0 < 32768 ? batch - b0 : 32768;
    ND4_TRY(nd4_gemm(h, false, false, I, J, K, 1.0, A + b0 * strideA, K, strideA, B + b0 * strideB, J, strideB,
                     0.0, C + b0 * I * J, J, I * J, nb));
  }
  return 0;
}


extern "C" int nd4hip_dgemm_ex_dev(nd4hip_handle* h, int transA, int transB, int64_t M, int64_t N, int64_t K,
                                   double alpha, const double* A, int64_t lda, const double* B, int64_t ldb,
                                   double beta, double* C, int64_t ldc) {
  ND4_CHECK_ARG(h != nullptr, "nd4hip_dgemm_ex: NULL handle");
  Nd4DeviceGuard guard(h);
  Nd4Prof prof(h, "dgemm_ex", (double)(2.0 * M * N * K), (double)(8.0 * (double)(M * K + K * N + M * N)));
  ND4_CHECK_ARG(M >= 0 && N >= 0 && K >= 0, "nd4hip_dgemm_ex: negative extent");
  ND4_CHECK_ARG(lda >= (transA ? M : K) && ldb >= (transB ? K : N) && ldc >= N, "nd4hip_dgemm_ex: leading dimension too small");
  if (M == 0 || N == 0) return 0;
  return nd4_gemm(h, transA != 0, transB != 0, M, N, K, alpha, A, lda, 0, B, ldb, 0, beta, C, ldc, 0, 1);
}

// ------------------------------------------------------------------------------------ LU
extern "C" int nd4hip_dgetrf_batched_dev(nd4hip_handle* h, int64_t batch, int64_t N, const double* A, double* LU, int32_t* P) {
  ND4_CHECK_ARG(h != nullptr, "nd4hip_dgetrf_batched: NULL handle");
  Nd4DeviceGuard guard(h);
  Nd4Prof prof(h, "dgetrf_batched", (double)(2.0 / 3.0 * batch * N * N * N), (double)(16.0 * batch * N * N));
  ND4_CHECK_ARG(batch >= 0 && N >= 0, "nd4hip_dgetrf_batched: negative extent");
  if (batch == 0 || N == 0) return 0;
  ND4_CHECK_ARG(A && LU && P, "nd4hip_dgetrf_batched: NULL pointer");
  ND4_FOR_CHUNKS(batch) ND4_TRY(nd4_getrf(h, nb, N, A + b0 * N * N, LU + b0 * N * N, P + b0 * N));
  return 0;
}

// ------------------------------------------------------------------------------------ solves
namespace {
// X[b] = Y[b*strideY] for every batch member (strideY = 0: the same Y for all)
int copy_rhs(nd4hip_handle* h, int64_t batch, int64_t rows, int64_t J, const double* Y, int64_t strideY, double* X) {
  return nd4_copy_matrix(h, rows, J, Y, J, X, J, batch, strideY, rows * J);
}
}  // namespace

extern "C" int nd4hip_dgetrs_batched_dev(nd4hip_handle* h, int64_t batch, int64_t N, int64_t J, const double* LU, int64_t strideLU,
                                         const int32_t* P, int64_t strideP, const double* Y, int64_t strideY, double* X) {
  ND4_CHECK_ARG(h != nullptr, "nd4hip_dgetrs_batched: NULL handle");
  Nd4DeviceGuard guard(h);
  Nd4Prof prof(h, "dgetrs_batched", (double)(2.0 * batch * N * N * J), (double)(8.0 * batch * (double)(N * N + 2 * N * J)));
  ND4_CHECK_ARG(batch >= 0 && N >= 0 && J >= 0, "nd4hip_dgetrs_batched: negative extent");
  ND4_CHECK_ARG((strideLU == 0 || strideLU >= N * N) && (strideP == 0 || strideP >= N) && (strideY == 0 || strideY >= N * J),
                "nd4hip_dgetrs_batched: a stride must be 0 or at least the size of one operand");
  if (batch == 0 || N == 0 || J == 0) return 0;
  ND4_CHECK_ARG(LU && P && Y && X, "nd4hip_dgetrs_batched: NULL pointer");
  ND4_FOR_CHUNKS(batch) ND4_TRY(nd4_getrs(h, nb, N, J, LU + b0 * strideLU, strideLU, P + b0 * strideP, strideP, Y + b0 * strideY, strideY, X + b0 * N * J));
  return 0;
}

extern "C" int nd4hip_dtrsm_batched_dev(nd4hip_handle* h, int upper, int unit_diag, int64_t batch, int64_t M, int64_t J,
                                        const double* T, int64_t strideT, const double* Y, int64_t strideY, double* X) {
  ND4_CHECK_ARG(h != nullptr, "nd4hip_dtrsm_batched: NULL handle");
  Nd4DeviceGuard guard(h);
  Nd4Prof prof(h, "dtrsm_batched", (double)(1.0 * batch * M * M * J), (double)(8.0 * batch * (double)(M * M / 2 + 2 * M * J)));
  ND4_CHECK_ARG(batch >= 0 && M >= 0 && J >= 0, "nd4hip_dtrsm_batched: negative extent");
  ND4_CHECK_ARG((strideT == 0 || strideT >= M * M) && (strideY == 0 || strideY >= M * J),
                "nd4hip_dtrsm_batched: a stride must be 0 or at least the size of one operand");
  if (batch == 0 || M == 0 || J == 0) return 0;
  ND4_CHECK_ARG(T && Y && X, "nd4hip_dtrsm_batched: NULL pointer");
  if (X != Y || strideY != M * J) ND4_TRY(copy_rhs(h, batch, M, J, Y, strideY, X));
  ND4_FOR_CHUNKS(batch) ND4_TRY(nd4_trsm(h, upper != 0, unit_diag != 0, nb, M, J, T + b0 * strideT, strideT, X + b0 * M * J));
  return 0;
}

// ---- least squares from a factorisation: qr_lstsq (qr.js:186-273), svd_lstsq / svd_solve (svd.js:66-228)
extern "C" int nd4hip_dqrls_batched_dev(nd4hip_handle* h, int64_t batch, int64_t N, int64_t M, int64_t I, int64_t J,
                                        const double* Q, int64_t strideQ, const double* R, int64_t strideR,
                                        const double* Y, int64_t strideY, double* X) {
  ND4_CHECK_ARG(h != nullptr, "nd4hip_dqrls_batched: NULL handle");
  Nd4DeviceGuard guard(h);
  Nd4Prof prof(h, "dqrls_batched", (double)(0.0), (double)(0.0));
  ND4_CHECK_ARG(batch >= 0 && N >= 0 && M >= 0 && I >= 0 && J >= 0, "nd4hip_dqrls_batched: negative extent");
  ND4_CHECK_ARG(I <= N, "qr_lstsq(Q,R,y): Under-determined systems not supported. Use rrqr instead.");      // qr.js:209
  ND4_CHECK_ARG((strideQ == 0 || strideQ >= N * M) && (strideR == 0 || strideR >= M * I) && (strideY == 0 || strideY >= N * J),
                "nd4hip_dqrls_batched: a stride must be 0 or at least the size of one operand");
  if (batch == 0 || I == 0 || J == 0) return 0;
  ND4_CHECK_ARG(X != nullptr, "nd4hip_dqrls_batched: NULL pointer");
  if (N == 0 || M == 0) { ND4_HIP(hipMemsetAsync(X, 0, D * (size_t)(batch * I * J), h->stream)); return 0; }
  ND4_CHECK_ARG(Q && R && Y, "nd4hip_dqrls_batched: NULL pointer");
  ND4_FOR_CHUNKS(batch) ND4_TRY(nd4_qrls(h, nb, N, M, I, J, Q + b0 * strideQ, strideQ, R + b0 * strideR, strideR, Y + b0 * strideY, strideY, X + b0 * I * J));
  return 0;
}

extern "C" int nd4hip_dsvdls_batched_dev(nd4hip_handle* h, int64_t batch, int64_t N, int64_t M, int64_t I, int64_t J,
                                         const double* U, int64_t strideU, const double* sv, int64_t strideSv,
                                         const double* V, int64_t strideV, const double* Y, int64_t strideY, double* X) {
  ND4_CHECK_ARG(h != nullptr, "nd4hip_dsvdls_batched: NULL handle");
  Nd4DeviceGuard guard(h);
  Nd4Prof prof(h, "dsvdls_batched", (double)(0.0), (double)(0.0));
  ND4_CHECK_ARG(batch >= 0 && N >= 0 && M >= 0 && I >= 0 && J >= 0, "nd4hip_dsvdls_batched: negative extent");
  ND4_CHECK_ARG((strideU == 0 || strideU >= N * M) && (strideSv == 0 || strideSv >= M) && (strideV == 0 || strideV >= M * I) &&
                (strideY == 0 || strideY >= N * J), "nd4hip_dsvdls_batched: a stride must be 0 or at least the size of one operand");
  if (batch == 0 || I == 0 || J == 0) return 0;
  ND4_CHECK_ARG(X != nullptr, "nd4hip_dsvdls_batched: NULL pointer");
  if (N == 0 || M == 0) { ND4_HIP(hipMemsetAsync(X, 0, D * (size_t)(batch * I * J), h->stream)); return 0; }
  ND4_CHECK_ARG(U && sv && V && Y, "nd4hip_dsvdls_batched: NULL pointer");
  ND4_FOR_CHUNKS(batch) ND4_TRY(nd4_svdls(h, nb, N, M, I, J, U + b0 * strideU, strideU, sv + b0 * strideSv, strideSv, V + b0 * strideV, strideV,
                                          Y + b0 * strideY, strideY, X + b0 * I * J));
  return 0;
}

// ---- Cholesky: cholesky_decomp (cholesky.js:51-71), cholesky_solve (:74-150)   
extern "C" int nd4hip_dpotrf_batched_dev(nd4hip_handle* h, int64_t batch, int64_t N, const double* S, double* L) {
  ND4_CHECK_ARG(h != nullptr, "nd4hip_dpotrf_batched: NULL handle");
  Nd4DeviceGuard guard(h);
  Nd4Prof prof(h, "dpotrf_batched", (double)(batch * N * N * N / 3.0), (double)(16.0 * batch * N * N));
  ND4_CHECK_ARG(batch >= 0 && N >= 0, "nd4hip_dpotrf_batched: negative extent");
  if (batch == 0 || N == 0) return 0;
  ND4_CHECK_ARG(S && L, "nd4hip_dpotrf_batched: NULL pointer");
  Nd4WsScope scope(h);
  void* p = nullptr;
  ND4_TRY(nd4_ws_alloc(h, sizeof(int) * (size_t)batch, &p));
  int* flags = static_cast<int*>(p);
  ND4_FOR_CHUNKS(batch) ND4_TRY(nd4_potrf(h, nb, N, S + b0 * N * N, L + b0 * N * N, flags + b0));
  // the reference throws on the first NaN pivot (cholesky.js:43-44): one small read-back decides it
  void* hp = nullptr;
  ND4_TRY(nd4_pinned(h, sizeof(int) * (size_t)batch, &hp));
  int* host = static_cast<int*>(hp);
  ND4_HIP(hipMemcpyAsync(host, flags, sizeof(int) * (size_t)batch, hipMemcpyDeviceToHost, h->stream));
  ND4_HIP(hipStreamSynchronize(h->stream));
  for (int64_t b = 0; b < batch; b++)
    if (host[b]) { nd4_set_error("Matrix contains NaNs or is (near) singular."); return ND4HIP_ERR_SINGULAR; }
  return 0;
}
extern "C" int nd4hip_dpotrs_batched_dev(nd4hip_handle* h, int64_t batch, int64_t N, int64_t J, const double* L, int64_t strideL,
                                         const double* Y, int64_t strideY, double* X) {
  ND4_CHECK_ARG(h != nullptr, "nd4hip_dpotrs_batched: NULL handle");
  Nd4DeviceGuard guard(h);
  Nd4Prof prof(h, "dpotrs_batched", (double)(2.0 * batch * N * N * J), (double)(8.0 * batch * (double)(N * N + 2 * N * J)));
  ND4_CHECK_ARG(batch >= 0 && N >= 0 && J >= 0, "nd4hip_dpotrs_batched: negative extent");
  ND4_CHECK_ARG((strideL == 0 || strideL >= N * N) && (strideY == 0 || strideY >= N * J),
                "nd4hip_dpotrs_batched: a stride must be 0 or at least the size of one operand");
  if (batch == 0 || N == 0 || J == 0) return 0;
  ND4_CHECK_ARG(L && Y && X, "nd4hip_dpotrs_batched: NULL pointer");
  ND4_FOR_CHUNKS(batch) ND4_TRY(nd4_potrs(h, nb, N, J, L + b0 * strideL, strideL, Y + b0 * strideY, strideY, X + b0 * N * J));
  return 0;
}

// ---- LDL^T: ldl_decomp (ldl.js:67-90), ldl_solve (:133-201)   
extern "C" int nd4hip_dldltrf_batched_dev(nd4hip_handle* h, int64_t batch, int64_t N, const double* S, double* LD) {
  ND4_CHECK_ARG(h != nullptr, "nd4hip_dldltrf_batched: NULL handle");
  Nd4DeviceGuard guard(h);
  Nd4Prof prof(h, "dldltrf_batched", (double)(batch * N * N * N / 3.0), (double)(16.0 * batch * N * N));
  ND4_CHECK_ARG(batch >= 0 && N >= 0, "nd4hip_dldltrf_batched: negative extent");
  if (batch == 0 || N == 0) return 0;
  ND4_CHECK_ARG(S && LD, "nd4hip_dldltrf_batched: NULL pointer");
  ND4_FOR_CHUNKS(batch) ND4_TRY(nd4_ldltrf(h, nb, N, S + b0 * N * N, LD + b0 * N * N));
  return 0;
}
extern "C" int nd4hip_dldltrs_batched_dev(nd4hip_handle* h, int64_t batch, int64_t N, int64_t J, const double* LD, int64_t strideLD,
                                          const double* Y, int64_t strideY, double* X) {
  ND4_CHECK_ARG(h != nullptr, "nd4hip_dldltrs_batched: NULL handle");
  Nd4DeviceGuard guard(h);
  Nd4Prof prof(h, "dldltrs_batched", (double)(2.0 * batch * N * N * J), (double)(8.0 * batch * (double)(N * N + 2 * N * J)));
  ND4_CHECK_ARG(batch >= 0 && N >= 0 && J >= 0, "nd4hip_dldltrs_batched: negative extent");
  ND4_CHECK_ARG((strideLD == 0 || strideLD >= N * N) && (strideY == 0 || strideY >= N * J),
                "nd4hip_dldltrs_batched: a stride must be 0 or at least the size of one operand");
  if (batch == 0 || N == 0 || J == 0) return 0;
  ND4_CHECK_ARG(LD && Y && X, "nd4hip_dldltrs_batched: NULL pointer");
  ND4_FOR_CHUNKS(batch) ND4_TRY(nd4_ldltrs(h, nb, N, J, LD + b0 * strideLD, strideLD, Y + b0 * strideY, strideY, X + b0 * N * J));
  return 0;
}

// ---- bidiag_decomp (bidiag.js:245-319)   
extern "C" int nd4hip_dgebrd_batched_dev(nd4hip_handle* h, int64_t batch, int64_t M, int64_t N, const double* A, double* U, double* B, double* V) {
  ND4_CHECK_ARG(h != nullptr, "nd4hip_dgebrd_batched: NULL handle");
  Nd4DeviceGuard guard(h);
  Nd4Prof prof(h, "dgebrd_batched", (double)(batch * nd4_flops_qr(M, N)), (double)(8.0 * batch * (double)(2 * M * N + M * M + N * N)));
  ND4_CHECK_ARG(batch >= 0 && M >= 0 && N >= 0, "nd4hip_dgebrd_batched: negative extent");
  if (batch == 0 || M == 0 || N == 0) return 0;
  ND4_CHECK_ARG(A && U && B && V, "nd4hip_dgebrd_batched: NULL pointer");
  { const int64_t K = M < N ? M : N, Jb = M >= N ? K : K + 1;
    ND4_FOR_CHUNKS(batch) ND4_TRY(nd4_gebrd(h, nb, M, N, A + b0 * M * N, U + b0 * M * K, B + b0 * K * Jb, V + b0 * Jb * N)); }
  return 0;
}

// ---- hessenberg_decomp (hessenberg.js:89-115)   
extern "C" int nd4hip_dgehrd_batched_dev(nd4hip_handle* h, int64_t batch, int64_t N, const double* A, double* U, double* H) {
  ND4_CHECK_ARG(h != nullptr, "nd4hip_dgehrd_batched: NULL handle");
  Nd4DeviceGuard guard(h);
  Nd4Prof prof(h, "dgehrd_batched", (double)(14.0 / 3.0 * batch * N * N * N), (double)(24.0 * batch * N * N));
  ND4_CHECK_ARG(batch >= 0 && N >= 0, "nd4hip_dgehrd_batched: negative extent");
  if (batch == 0 || N == 0) return 0;
  ND4_CHECK_ARG(A && U && H, "nd4hip_dgehrd_batched: NULL pointer");
  ND4_FOR_CHUNKS(batch) ND4_TRY(nd4_gehrd(h, nb, N, A + b0 * N * N, U + b0 * N * N, H + b0 * N * N));
  return 0;
}

// ------------------------------------------------------------------------------------ QR
extern "C" int nd4hip_dgeqrf_q_batched_dev(nd4hip_handle* h, int64_t batch, int64_t M, int64_t N, const double* A, double* Q, double* R) {
  ND4_CHECK_ARG(h != nullptr, "nd4hip_dgeqrf_q_batched: NULL handle");
  Nd4DeviceGuard guard(h);
  Nd4Prof prof(h, "dgeqrf_q_batched", (double)(batch * nd4_flops_qr(M, N)), (double)(8.0 * batch * (double)(M * N + (M + N) * (M < N ? M : N))));
  ND4_CHECK_ARG(batch >= 0 && M >= 0 && N >= 0, "nd4hip_dgeqrf_q_batched: negative extent");
  if (batch == 0 || M == 0 || N == 0) return 0;
  ND4_CHECK_ARG(A && Q && R, "nd4hip_dgeqrf_q_batched: NULL pointer");
  { const int64_t L = M < N ? M : N;
    ND4_FOR_CHUNKS(batch) ND4_TRY(nd4_geqrf_q(h, nb, M, N, A + b0 * M * N, Q + b0 * M * L, R + b0 * L * N)); }
  return 0;
}

// qr_decomp_full (qr.js:27-77) for every shape: Q [M, M], R [M, N]
extern "C" int nd4hip_dgeqrf_full_batched_dev(nd4hip_handle* h, int64_t batch, int64_t M, int64_t N, const double* A, double* Q, double* R) {
  ND4_CHECK_ARG(h != nullptr, "nd4hip_dgeqrf_full_batched: NULL handle");
  Nd4DeviceGuard guard(h);
  Nd4Prof prof(h, "dgeqrf_full_batched", (double)(batch * nd4_flops_qr(M, N)), (double)(8.0 * batch * (double)(2 * M * N + M * M)));
  ND4_CHECK_ARG(batch >= 0 && M >= 0 && N >= 0, "nd4hip_dgeqrf_full_batched: negative extent");
  if (batch == 0 || M == 0 || N == 0) return 0;
  ND4_CHECK_ARG(A && Q && R, "nd4hip_dgeqrf_full_batched: NULL pointer");
  ND4_FOR_CHUNKS(batch) ND4_TRY(nd4_geqrf_q_ex(h, nb, M, N, A + b0 * M * N, Q + b0 * M * M, R + b0 * M * N, true));
  return 0;
}

// _qr_decomp_inplace (qr.js:146-183): A [M, N] <- R, Y [M, L] <- Q^T Y with the full (M x M) Q
extern "C" int nd4hip_dgeqrf_qty_batched_dev(nd4hip_handle* h, int64_t batch, int64_t M, int64_t N, int64_t L, double* A, double* Y) {
  ND4_CHECK_ARG(h != nullptr, "nd4hip_dgeqrf_qty_batched: NULL handle");
  Nd4DeviceGuard guard(h);
  Nd4Prof prof(h, "dgeqrf_qty_batched", (double)(batch * nd4_flops_qr(M, N)), (double)(16.0 * batch * (double)(M * N + M * L)));
  ND4_CHECK_ARG(batch >= 0 && M >= 0 && N >= 0 && L >= 0, "nd4hip_dgeqrf_qty_batched: negative extent");
  if (batch == 0 || M == 0 || N == 0) return 0;
  ND4_CHECK_ARG(A && (Y || L == 0), "nd4hip_dgeqrf_qty_batched: NULL pointer");
  Nd4WsScope scope(h);
  void* p = nullptr;
  ND4_TRY(nd4_ws_alloc(h, D * (size_t)batch * (size_t)(M * M + M * N + M * L), &p));
  double* Q = static_cast<double*>(p);
  double* R = Q + (size_t)batch * M * M;
  double* Yt = R + (size_t)batch * M * N;
  ND4_TRY(nd4_geqrf_q_ex(h, batch, M, N, A, Q, R, true));
  ND4_HIP(hipMemcpyAsync(A, R, D * (size_t)(batch * M * N), hipMemcpyDeviceToDevice, h->stream));
  if (L > 0) {
    ND4_TRY(nd4_gemm(h, true, false, M, L, M, 1.0, Q, M, M * M, Y, L, M * L, 0.0, Yt, L, M * L, batch));
    ND4_HIP(hipMemcpyAsync(Y, Yt, D * (size_t)(batch * M * L), hipMemcpyDeviceToDevice, h->stream));
  }
  return 0;
}

// ------------------------------------------------------------------------------------ SVD
extern "C" int nd4hip_dgesvdj_batched_dev(nd4hip_handle* h, int64_t batch, int64_t M, int64_t N, const double* A,
                                          double* U, double* sv, double* V, int* sweeps_out, double* offnorm_out) {
  ND4_CHECK_ARG(h != nullptr, "nd4hip_dgesvdj_batched: NULL handle");
  Nd4DeviceGuard guard(h);
  Nd4Prof prof(h, "dgesvdj_batched", (double)(batch * nd4_flops_svd(M, N)), (double)(8.0 * batch * (double)(M * N + (M + N + 1) * (M < N ? M : N))));
  ND4_CHECK_ARG(batch >= 0 && M >= 0 && N >= 0, "nd4hip_dgesvdj_batched: negative extent");
  if (sweeps_out) *sweeps_out = 0;
  if (offnorm_out) *offnorm_out = 0.0;
  if (batch == 0 || M == 0 || N == 0) return 0;
  ND4_CHECK_ARG(A && U && sv && V, "nd4hip_dgesvdj_batched: NULL pointer");
  {
    const int64_t L = M < N ? M : N;
    int sweeps = 0; double off = 0.0; unsigned long long rot = 0;
    ND4_FOR_CHUNKS(batch) {
      int sw = 0; double of = 0.0;
      ND4_TRY(nd4_gesvdj(h, nb, M, N, A + b0 * M * N, U + b0 * M * L, sv + b0 * L, V + b0 * L * N, &sw, &of));
      if (sw > sweeps) sweeps = sw;
      if (of > off) off = of;
      rot += h->svd_rotations;
    }
    h->svd_sweeps = sweeps; h->svd_offnorm = off; h->svd_rotations = rot;
    if (sweeps_out) *sweeps_out = sweeps;
    if (offnorm_out) *offnorm_out = off;
  }
  return 0;
}

extern "C" int nd4hip_dgeqr2_panel_batched_dev(nd4hip_handle* h, int64_t batch, int64_t M, int64_t b, double* A, double* V, double* T) {
  ND4_CHECK_ARG(h != nullptr, "nd4hip_dgeqr2_panel_batched: NULL handle");
  Nd4DeviceGuard guard(h);
  Nd4Prof prof(h, "dgeqr2_panel_batched", (double)(2.0 * batch * M * b * b), (double)(16.0 * batch * M * b));
  ND4_CHECK_ARG(b == 16, "nd4hip_dgeqr2_panel_batched: the panel width is 16");
  ND4_CHECK_ARG(batch >= 0 && batch <= 65535 && M >= 1 && M <= 2048, "nd4hip_dgeqr2_panel_batched: 1 <= M <= 2048 rows, batch <= 65535");
  if (batch == 0) return 0;
  ND4_CHECK_ARG(A && V && T, "nd4hip_dgeqr2_panel_batched: NULL pointer");
  return nd4_geqr2_panel(h, (int)batch, (int)M, A, V, T);
}

extern "C" int nd4hip_dgesvdj_last_info(nd4hip_handle* h, int* sweeps, unsigned long long* rotations, double* offnorm) {
  ND4_CHECK_ARG(h != nullptr, "nd4hip_dgesvdj_last_info: NULL handle");
  if (sweeps) *sweeps = h->svd_sweeps;
  if (rotations) *rotations = h->svd_rotations;
  if (offnorm) *offnorm = h->svd_offnorm;
  return 0;
}
