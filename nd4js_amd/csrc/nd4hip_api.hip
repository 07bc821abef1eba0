// extern "C" entry points of include/nd4hip.h: argument validation, the *_dev (device pointer) forms
// and the host-pointer forms (H2D -> kernels -> D2H) that the N-API shim binds.
#include "nd4hip_internal.h"
#include <cmath>

namespace {

// Device staging for the host-pointer entry points. The blocks are cached in the handle between calls: every host-pointer
// call synchronises before it returns, so a released block is immediately reusable, and hipMalloc + hipFree (~100 us
// each, several per call) used to be most of the latency of a small call (16 x 16 svd_decomp: 1.05 ms). Best fit among the
// free blocks that are not more than 4x too large; the cache is trimmed when it exceeds ND4_STAGE_CAP bytes.
constexpr size_t ND4_STAGE_CAP = size_t(6) << 30;
struct DevBuf {
  void* p = nullptr;
  nd4hip_handle* h = nullptr;
  int slot = -1;
  ~DevBuf() {
    // error path of a host-pointer call: its async copies / kernels may still be using the block
    if (h && h->host_io_pending) { (void)hipStreamSynchronize(h->stream); h->host_io_pending = false; }
    if (slot >= 0) h->stage[(size_t)slot].in_use = false;
    else if (p) (void)hipFree(p);
  }
  int alloc(nd4hip_handle* hh, size_t bytes) {
    h = hh;
    if (bytes == 0) bytes = 8;
    int best = -1;
    for (size_t i = 0; i < h->stage.size(); i++) {
      const Nd4Stage& b = h->stage[i];
      if (!b.in_use && b.bytes >= bytes && b.bytes <= 4 * bytes + 4096 && (best < 0 || b.bytes < h->stage[(size_t)best].bytes)) best = (int)i;
    }
    if (best >= 0) { h->stage[(size_t)best].in_use = true; slot = best; p = h->stage[(size_t)best].p; return 0; }
    if (h->stage_bytes + bytes > ND4_STAGE_CAP || h->stage.size() >= 64) {
      bool any_used = false;
      for (const auto& b : h->stage) any_used = any_used || b.in_use;
      if (any_used) { ND4_HIP(hipMalloc(&p, bytes)); slot = -1; return 0; }     // over budget mid-call: an uncached block
      for (auto& b : h->stage) (void)hipFree(b.p);                              // between calls: start the cache afresh
      h->stage.clear(); h->stage_bytes = 0;
    }
    ND4_HIP(hipMalloc(&p, bytes));
    h->stage.push_back(Nd4Stage{p, bytes, true});
    h->stage_bytes += bytes;
    slot = (int)h->stage.size() - 1;
    return 0;
  }
};

int h2d(nd4hip_handle* h, void* d, const void* s, size_t bytes) {
  h->host_io_pending = true;
  if (bytes) ND4_HIP(hipMemcpyAsync(d, s, bytes, hipMemcpyHostToDevice, h->stream));
  return 0;
}
int d2h(nd4hip_handle* h, void* d, const void* s, size_t bytes) {
  h->host_io_pending = true;
  if (bytes) ND4_HIP(hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToHost, h->stream));
  return 0;
}
// end of a host-pointer call (or a mid-call read-back): everything queued on the stream has landed
int host_sync(nd4hip_handle* h) {
  ND4_HIP(hipStreamSynchronize(h->stream));
  h->host_io_pending = false;
  return 0;
}
constexpr size_t D = sizeof(double);

// The kernels carry the batch in gridDim.y/z (<= 65535): longer batches run in chunks of ND4_CHUNK matrices. `b0` is the
// first matrix of the chunk, `nb` its length; strides that are 0 (broadcast operand) stay 0.
constexpr int64_t ND4_CHUNK = 32768;
#define ND4_FOR_CHUNKS(batch) for (int64_t b0 = 0, nb = 0; (nb = ((batch) - b0 < ND4_CHUNK ? (batch) - b0 : ND4_CHUNK)) > 0; b0 += nb)

}  // namespace

// ------------------------------------------------------------------------------------ matmul
extern "C" int nd4hip_dgemm_batched_dev(nd4hip_handle* h, int64_t batch, int64_t I, int64_t K, int64_t J,
                                        const double* A, int64_t strideA, const double* B, int64_t strideB, double* C) {
  ND4_CHECK_ARG(h != nullptr, "nd4hip_dgemm_batched: NULL handle");
  Nd4DeviceGuard guard(h);
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

extern "C" int nd4hip_dgemm_batched(nd4hip_handle* h, int64_t batch, int64_t I, int64_t K, int64_t J,
                                    const double* A, int64_t strideA, const double* B, int64_t strideB, double* C) {
  ND4_CHECK_ARG(h != nullptr, "nd4hip_dgemm_batched: NULL handle");
  Nd4DeviceGuard guard(h);
  ND4_CHECK_ARG(batch >= 0 && I >= 0 && K >= 0 && J >= 0, "nd4hip_dgemm_batched: negative extent");
  if (batch == 0 || I == 0 || J == 0) return 0;
  const size_t nA = (size_t)(strideA ? (batch - 1) * strideA + I * K : I * K);
  const size_t nB = (size_t)(strideB ? (batch - 1) * strideB + K * J : K * J);
  const size_t nC = (size_t)(batch * I * J);
  DevBuf dA, dB, dC;
  ND4_TRY(dA.alloc(h, nA * D)); ND4_TRY(dB.alloc(h, nB * D)); ND4_TRY(dC.alloc(h, nC * D));
  ND4_TRY(h2d(h, dA.p, A, nA * D)); ND4_TRY(h2d(h, dB.p, B, nB * D));
  ND4_TRY(nd4hip_dgemm_batched_dev(h, batch, I, K, J, (const double*)dA.p, strideA, (const double*)dB.p, strideB, (double*)dC.p));
  ND4_TRY(d2h(h, C, dC.p, nC * D));
  ND4_TRY(host_sync(h));
  return 0;
}

extern "C" int nd4hip_dgemm_ex_dev(nd4hip_handle* h, int transA, int transB, int64_t M, int64_t N, int64_t K,
                                   double alpha, const double* A, int64_t lda, const double* B, int64_t ldb,
                                   double beta, double* C, int64_t ldc) {
  ND4_CHECK_ARG(h != nullptr, "nd4hip_dgemm_ex: NULL handle");
  Nd4DeviceGuard guard(h);
  ND4_CHECK_ARG(M >= 0 && N >= 0 && K >= 0, "nd4hip_dgemm_ex: negative extent");
  ND4_CHECK_ARG(lda >= (transA ? M : K) && ldb >= (transB ? K : N) && ldc >= N, "nd4hip_dgemm_ex: leading dimension too small");
  if (M == 0 || N == 0) return 0;
  return nd4_gemm(h, transA != 0, transB != 0, M, N, K, alpha, A, lda, 0, B, ldb, 0, beta, C, ldc, 0, 1);
}

// ------------------------------------------------------------------------------------ LU
extern "C" int nd4hip_dgetrf_batched_dev(nd4hip_handle* h, int64_t batch, int64_t N, const double* A, double* LU, int32_t* P) {
  ND4_CHECK_ARG(h != nullptr, "nd4hip_dgetrf_batched: NULL handle");
  Nd4DeviceGuard guard(h);
  ND4_CHECK_ARG(batch >= 0 && N >= 0, "nd4hip_dgetrf_batched: negative extent");
  if (batch == 0 || N == 0) return 0;
  ND4_CHECK_ARG(A && LU && P, "nd4hip_dgetrf_batched: NULL pointer");
  ND4_FOR_CHUNKS(batch) ND4_TRY(nd4_getrf(h, nb, N, A + b0 * N * N, LU + b0 * N * N, P + b0 * N));
  return 0;
}
extern "C" int nd4hip_dgetrf_batched(nd4hip_handle* h, int64_t batch, int64_t N, const double* A, double* LU, int32_t* P) {
  ND4_CHECK_ARG(h != nullptr, "nd4hip_dgetrf_batched: NULL handle");
  Nd4DeviceGuard guard(h);
  ND4_CHECK_ARG(batch >= 0 && N >= 0, "nd4hip_dgetrf_batched: negative extent");
  if (batch == 0 || N == 0) return 0;
  const size_t n = (size_t)(batch * N * N);
  DevBuf dA, dLU, dP;
  ND4_TRY(dA.alloc(h, n * D)); ND4_TRY(dLU.alloc(h, n * D)); ND4_TRY(dP.alloc(h, (size_t)(batch * N) * 4));
  ND4_TRY(h2d(h, dA.p, A, n * D));
  ND4_TRY(nd4hip_dgetrf_batched_dev(h, batch, N, (const double*)dA.p, (double*)dLU.p, (int32_t*)dP.p));
  ND4_TRY(d2h(h, LU, dLU.p, n * D)); ND4_TRY(d2h(h, P, dP.p, (size_t)(batch * N) * 4));
  ND4_TRY(host_sync(h));
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
  ND4_CHECK_ARG(batch >= 0 && N >= 0 && J >= 0, "nd4hip_dgetrs_batched: negative extent");
  ND4_CHECK_ARG((strideLU == 0 || strideLU >= N * N) && (strideP == 0 || strideP >= N) && (strideY == 0 || strideY >= N * J),
                "nd4hip_dgetrs_batched: a stride must be 0 or at least the size of one operand");
  if (batch == 0 || N == 0 || J == 0) return 0;
  ND4_CHECK_ARG(LU && P && Y && X, "nd4hip_dgetrs_batched: NULL pointer");
  ND4_FOR_CHUNKS(batch) ND4_TRY(nd4_getrs(h, nb, N, J, LU + b0 * strideLU, strideLU, P + b0 * strideP, strideP, Y + b0 * strideY, strideY, X + b0 * N * J));
  return 0;
}
extern "C" int nd4hip_dgetrs_batched(nd4hip_handle* h, int64_t batch, int64_t N, int64_t J, const double* LU, int64_t strideLU,
                                     const int32_t* P, int64_t strideP, const double* Y, int64_t strideY, double* X) {
  ND4_CHECK_ARG(h != nullptr, "nd4hip_dgetrs_batched: NULL handle");
  Nd4DeviceGuard guard(h);
  ND4_CHECK_ARG(batch >= 0 && N >= 0 && J >= 0, "nd4hip_dgetrs_batched: negative extent");
  if (batch == 0 || N == 0 || J == 0) return 0;
  const size_t nLU = (size_t)(strideLU ? (batch - 1) * strideLU + N * N : N * N);
  const size_t nP = (size_t)(strideP ? (batch - 1) * strideP + N : N);
  const size_t nY = (size_t)(strideY ? (batch - 1) * strideY + N * J : N * J);
  const size_t nX = (size_t)(batch * N * J);
  DevBuf dLU, dP, dY, dX;
  ND4_TRY(dLU.alloc(h, nLU * D)); ND4_TRY(dP.alloc(h, nP * 4)); ND4_TRY(dY.alloc(h, nY * D)); ND4_TRY(dX.alloc(h, nX * D));
  ND4_TRY(h2d(h, dLU.p, LU, nLU * D)); ND4_TRY(h2d(h, dP.p, P, nP * 4)); ND4_TRY(h2d(h, dY.p, Y, nY * D));
  ND4_TRY(nd4hip_dgetrs_batched_dev(h, batch, N, J, (const double*)dLU.p, strideLU, (const int32_t*)dP.p, strideP,
                                    (const double*)dY.p, strideY, (double*)dX.p));
  ND4_TRY(d2h(h, X, dX.p, nX * D));
  ND4_TRY(host_sync(h));
  return 0;
}

extern "C" int nd4hip_dtrsm_batched_dev(nd4hip_handle* h, int upper, int unit_diag, int64_t batch, int64_t M, int64_t J,
                                        const double* T, int64_t strideT, const double* Y, int64_t strideY, double* X) {
  ND4_CHECK_ARG(h != nullptr, "nd4hip_dtrsm_batched: NULL handle");
  Nd4DeviceGuard guard(h);
  ND4_CHECK_ARG(batch >= 0 && M >= 0 && J >= 0, "nd4hip_dtrsm_batched: negative extent");
  ND4_CHECK_ARG((strideT == 0 || strideT >= M * M) && (strideY == 0 || strideY >= M * J),
                "nd4hip_dtrsm_batched: a stride must be 0 or at least the size of one operand");
  if (batch == 0 || M == 0 || J == 0) return 0;
  ND4_CHECK_ARG(T && Y && X, "nd4hip_dtrsm_batched: NULL pointer");
  if (X != Y || strideY != M * J) ND4_TRY(copy_rhs(h, batch, M, J, Y, strideY, X));
  ND4_FOR_CHUNKS(batch) ND4_TRY(nd4_trsm(h, upper != 0, unit_diag != 0, nb, M, J, T + b0 * strideT, strideT, X + b0 * M * J));
  return 0;
}
extern "C" int nd4hip_dtrsm_batched(nd4hip_handle* h, int upper, int unit_diag, int64_t batch, int64_t M, int64_t J,
                                    const double* T, int64_t strideT, const double* Y, int64_t strideY, double* X) {
  ND4_CHECK_ARG(h != nullptr, "nd4hip_dtrsm_batched: NULL handle");
  Nd4DeviceGuard guard(h);
  ND4_CHECK_ARG(batch >= 0 && M >= 0 && J >= 0, "nd4hip_dtrsm_batched: negative extent");
  if (batch == 0 || M == 0 || J == 0) return 0;
  const size_t nT = (size_t)(strideT ? (batch - 1) * strideT + M * M : M * M);
  const size_t nY = (size_t)(strideY ? (batch - 1) * strideY + M * J : M * J);
  const size_t nX = (size_t)(batch * M * J);
  DevBuf dT, dY, dX;
  ND4_TRY(dT.alloc(h, nT * D)); ND4_TRY(dY.alloc(h, nY * D)); ND4_TRY(dX.alloc(h, nX * D));
  ND4_TRY(h2d(h, dT.p, T, nT * D)); ND4_TRY(h2d(h, dY.p, Y, nY * D));
  ND4_TRY(nd4hip_dtrsm_batched_dev(h, upper, unit_diag, batch, M, J, (const double*)dT.p, strideT, (const double*)dY.p, strideY, (double*)dX.p));
  ND4_TRY(d2h(h, X, dX.p, nX * D));
  ND4_TRY(host_sync(h));
  return 0;
}

// ---- least squares from a factorisation: qr_lstsq (qr.js:186-273), svd_lstsq / svd_solve (svd.js:66-228)
extern "C" int nd4hip_dqrls_batched_dev(nd4hip_handle* h, int64_t batch, int64_t N, int64_t M, int64_t I, int64_t J,
                                        const double* Q, int64_t strideQ, const double* R, int64_t strideR,
                                        const double* Y, int64_t strideY, double* X) {
  ND4_CHECK_ARG(h != nullptr, "nd4hip_dqrls_batched: NULL handle");
  Nd4DeviceGuard guard(h);
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
extern "C" int nd4hip_dqrls_batched(nd4hip_handle* h, int64_t batch, int64_t N, int64_t M, int64_t I, int64_t J,
                                    const double* Q, int64_t strideQ, const double* R, int64_t strideR,
                                    const double* Y, int64_t strideY, double* X) {
  ND4_CHECK_ARG(h != nullptr, "nd4hip_dqrls_batched: NULL handle");
  Nd4DeviceGuard guard(h);
  ND4_CHECK_ARG(batch >= 0 && N >= 0 && M >= 0 && I >= 0 && J >= 0, "nd4hip_dqrls_batched: negative extent");
  if (batch == 0 || I == 0 || J == 0) return 0;
  const size_t nQ = (size_t)(strideQ ? (batch - 1) * strideQ + N * M : N * M);
  const size_t nR = (size_t)(strideR ? (batch - 1) * strideR + M * I : M * I);
  const size_t nY = (size_t)(strideY ? (batch - 1) * strideY + N * J : N * J);
  const size_t nX = (size_t)(batch * I * J);
  DevBuf dQ, dR, dY, dX;
  ND4_TRY(dQ.alloc(h, nQ * D)); ND4_TRY(dR.alloc(h, nR * D)); ND4_TRY(dY.alloc(h, nY * D)); ND4_TRY(dX.alloc(h, nX * D));
  if (nQ) ND4_TRY(h2d(h, dQ.p, Q, nQ * D));
  if (nR) ND4_TRY(h2d(h, dR.p, R, nR * D));
  if (nY) ND4_TRY(h2d(h, dY.p, Y, nY * D));
  ND4_TRY(nd4hip_dqrls_batched_dev(h, batch, N, M, I, J, (const double*)dQ.p, strideQ, (const double*)dR.p, strideR,
                                   (const double*)dY.p, strideY, (double*)dX.p));
  ND4_TRY(d2h(h, X, dX.p, nX * D));
  ND4_TRY(host_sync(h));
  return 0;
}

extern "C" int nd4hip_dsvdls_batched_dev(nd4hip_handle* h, int64_t batch, int64_t N, int64_t M, int64_t I, int64_t J,
                                         const double* U, int64_t strideU, const double* sv, int64_t strideSv,
                                         const double* V, int64_t strideV, const double* Y, int64_t strideY, double* X) {
  ND4_CHECK_ARG(h != nullptr, "nd4hip_dsvdls_batched: NULL handle");
  Nd4DeviceGuard guard(h);
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
extern "C" int nd4hip_dsvdls_batched(nd4hip_handle* h, int64_t batch, int64_t N, int64_t M, int64_t I, int64_t J,
                                     const double* U, int64_t strideU, const double* sv, int64_t strideSv,
                                     const double* V, int64_t strideV, const double* Y, int64_t strideY, double* X) {
  ND4_CHECK_ARG(h != nullptr, "nd4hip_dsvdls_batched: NULL handle");
  Nd4DeviceGuard guard(h);
  ND4_CHECK_ARG(batch >= 0 && N >= 0 && M >= 0 && I >= 0 && J >= 0, "nd4hip_dsvdls_batched: negative extent");
  if (batch == 0 || I == 0 || J == 0) return 0;
  const size_t nU = (size_t)(strideU ? (batch - 1) * strideU + N * M : N * M);
  const size_t nS = (size_t)(strideSv ? (batch - 1) * strideSv + M : M);
  const size_t nV = (size_t)(strideV ? (batch - 1) * strideV + M * I : M * I);
  const size_t nY = (size_t)(strideY ? (batch - 1) * strideY + N * J : N * J);
  const size_t nX = (size_t)(batch * I * J);
  if (M > 0) {
    ND4_CHECK_ARG(sv != nullptr, "nd4hip_dsvdls_batched: NULL pointer");
    for (size_t i = 0; i < nS; i++) ND4_CHECK_ARG(std::isfinite(sv[i]), "svd_solve(): NaN or Infinity encountered.");   // svd.js:171-172
  }
  DevBuf dU, dS, dV, dY, dX;
  ND4_TRY(dU.alloc(h, nU * D)); ND4_TRY(dS.alloc(h, nS * D)); ND4_TRY(dV.alloc(h, nV * D)); ND4_TRY(dY.alloc(h, nY * D)); ND4_TRY(dX.alloc(h, nX * D));
  if (nU) ND4_TRY(h2d(h, dU.p, U, nU * D));
  if (nS) ND4_TRY(h2d(h, dS.p, sv, nS * D));
  if (nV) ND4_TRY(h2d(h, dV.p, V, nV * D));
  if (nY) ND4_TRY(h2d(h, dY.p, Y, nY * D));
  ND4_TRY(nd4hip_dsvdls_batched_dev(h, batch, N, M, I, J, (const double*)dU.p, strideU, (const double*)dS.p, strideSv,
                                    (const double*)dV.p, strideV, (const double*)dY.p, strideY, (double*)dX.p));
  ND4_TRY(d2h(h, X, dX.p, nX * D));
  ND4_TRY(host_sync(h));
  return 0;
}

// ---- Cholesky: cholesky_decomp (cholesky.js:51-71), cholesky_solve (:74-150)   
extern "C" int nd4hip_dpotrf_batched_dev(nd4hip_handle* h, int64_t batch, int64_t N, const double* S, double* L) {
  ND4_CHECK_ARG(h != nullptr, "nd4hip_dpotrf_batched: NULL handle");
  Nd4DeviceGuard guard(h);
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
  ND4_TRY(host_sync(h));
  for (int64_t b = 0; b < batch; b++)
    if (host[b]) { nd4_set_error("Matrix contains NaNs or is (near) singular."); return ND4HIP_ERR_SINGULAR; }
  return 0;
}
extern "C" int nd4hip_dpotrf_batched(nd4hip_handle* h, int64_t batch, int64_t N, const double* S, double* L) {
  ND4_CHECK_ARG(h != nullptr, "nd4hip_dpotrf_batched: NULL handle");
  Nd4DeviceGuard guard(h);
  ND4_CHECK_ARG(batch >= 0 && N >= 0, "nd4hip_dpotrf_batched: negative extent");
  if (batch == 0 || N == 0) return 0;
  const size_t n = (size_t)(batch * N * N);
  DevBuf dS, dL;
  ND4_TRY(dS.alloc(h, n * D)); ND4_TRY(dL.alloc(h, n * D));
  ND4_TRY(h2d(h, dS.p, S, n * D));
  ND4_TRY(nd4hip_dpotrf_batched_dev(h, batch, N, (const double*)dS.p, (double*)dL.p));
  ND4_TRY(d2h(h, L, dL.p, n * D));
  ND4_TRY(host_sync(h));
  return 0;
}
extern "C" int nd4hip_dpotrs_batched_dev(nd4hip_handle* h, int64_t batch, int64_t N, int64_t J, const double* L, int64_t strideL,
                                         const double* Y, int64_t strideY, double* X) {
  ND4_CHECK_ARG(h != nullptr, "nd4hip_dpotrs_batched: NULL handle");
  Nd4DeviceGuard guard(h);
  ND4_CHECK_ARG(batch >= 0 && N >= 0 && J >= 0, "nd4hip_dpotrs_batched: negative extent");
  ND4_CHECK_ARG((strideL == 0 || strideL >= N * N) && (strideY == 0 || strideY >= N * J),
                "nd4hip_dpotrs_batched: a stride must be 0 or at least the size of one operand");
  if (batch == 0 || N == 0 || J == 0) return 0;
  ND4_CHECK_ARG(L && Y && X, "nd4hip_dpotrs_batched: NULL pointer");
  ND4_FOR_CHUNKS(batch) ND4_TRY(nd4_potrs(h, nb, N, J, L + b0 * strideL, strideL, Y + b0 * strideY, strideY, X + b0 * N * J));
  return 0;
}
extern "C" int nd4hip_dpotrs_batched(nd4hip_handle* h, int64_t batch, int64_t N, int64_t J, const double* L, int64_t strideL,
                                     const double* Y, int64_t strideY, double* X) {
  ND4_CHECK_ARG(h != nullptr, "nd4hip_dpotrs_batched: NULL handle");
  Nd4DeviceGuard guard(h);
  ND4_CHECK_ARG(batch >= 0 && N >= 0 && J >= 0, "nd4hip_dpotrs_batched: negative extent");
  if (batch == 0 || N == 0 || J == 0) return 0;
  const size_t nL = (size_t)(strideL ? (batch - 1) * strideL + N * N : N * N);
  const size_t nY = (size_t)(strideY ? (batch - 1) * strideY + N * J : N * J);
  const size_t nX = (size_t)(batch * N * J);
  DevBuf dL, dY, dX;
  ND4_TRY(dL.alloc(h, nL * D)); ND4_TRY(dY.alloc(h, nY * D)); ND4_TRY(dX.alloc(h, nX * D));
  ND4_TRY(h2d(h, dL.p, L, nL * D)); ND4_TRY(h2d(h, dY.p, Y, nY * D));
  ND4_TRY(nd4hip_dpotrs_batched_dev(h, batch, N, J, (const double*)dL.p, strideL, (const double*)dY.p, strideY, (double*)dX.p));
  ND4_TRY(d2h(h, X, dX.p, nX * D));
  ND4_TRY(host_sync(h));
  return 0;
}

// ---- LDL^T: ldl_decomp (ldl.js:67-90), ldl_solve (:133-201)   
extern "C" int nd4hip_dldltrf_batched_dev(nd4hip_handle* h, int64_t batch, int64_t N, const double* S, double* LD) {
  ND4_CHECK_ARG(h != nullptr, "nd4hip_dldltrf_batched: NULL handle");
  Nd4DeviceGuard guard(h);
  ND4_CHECK_ARG(batch >= 0 && N >= 0, "nd4hip_dldltrf_batched: negative extent");
  if (batch == 0 || N == 0) return 0;
  ND4_CHECK_ARG(S && LD, "nd4hip_dldltrf_batched: NULL pointer");
  ND4_FOR_CHUNKS(batch) ND4_TRY(nd4_ldltrf(h, nb, N, S + b0 * N * N, LD + b0 * N * N));
  return 0;
}
extern "C" int nd4hip_dldltrf_batched(nd4hip_handle* h, int64_t batch, int64_t N, const double* S, double* LD) {
  ND4_CHECK_ARG(h != nullptr, "nd4hip_dldltrf_batched: NULL handle");
  Nd4DeviceGuard guard(h);
  ND4_CHECK_ARG(batch >= 0 && N >= 0, "nd4hip_dldltrf_batched: negative extent");
  if (batch == 0 || N == 0) return 0;
  const size_t n = (size_t)(batch * N * N);
  DevBuf dS, dL;
  ND4_TRY(dS.alloc(h, n * D)); ND4_TRY(dL.alloc(h, n * D));
  ND4_TRY(h2d(h, dS.p, S, n * D));
  ND4_TRY(nd4hip_dldltrf_batched_dev(h, batch, N, (const double*)dS.p, (double*)dL.p));
  ND4_TRY(d2h(h, LD, dL.p, n * D));
  ND4_TRY(host_sync(h));
  return 0;
}
extern "C" int nd4hip_dldltrs_batched_dev(nd4hip_handle* h, int64_t batch, int64_t N, int64_t J, const double* LD, int64_t strideLD,
                                          const double* Y, int64_t strideY, double* X) {
  ND4_CHECK_ARG(h != nullptr, "nd4hip_dldltrs_batched: NULL handle");
  Nd4DeviceGuard guard(h);
  ND4_CHECK_ARG(batch >= 0 && N >= 0 && J >= 0, "nd4hip_dldltrs_batched: negative extent");
  ND4_CHECK_ARG((strideLD == 0 || strideLD >= N * N) && (strideY == 0 || strideY >= N * J),
                "nd4hip_dldltrs_batched: a stride must be 0 or at least the size of one operand");
  if (batch == 0 || N == 0 || J == 0) return 0;
  ND4_CHECK_ARG(LD && Y && X, "nd4hip_dldltrs_batched: NULL pointer");
  ND4_FOR_CHUNKS(batch) ND4_TRY(nd4_ldltrs(h, nb, N, J, LD + b0 * strideLD, strideLD, Y + b0 * strideY, strideY, X + b0 * N * J));
  return 0;
}
extern "C" int nd4hip_dldltrs_batched(nd4hip_handle* h, int64_t batch, int64_t N, int64_t J, const double* LD, int64_t strideLD,
                                      const double* Y, int64_t strideY, double* X) {
  ND4_CHECK_ARG(h != nullptr, "nd4hip_dldltrs_batched: NULL handle");
  Nd4DeviceGuard guard(h);
  ND4_CHECK_ARG(batch >= 0 && N >= 0 && J >= 0, "nd4hip_dldltrs_batched: negative extent");
  if (batch == 0 || N == 0 || J == 0) return 0;
  const size_t nL = (size_t)(strideLD ? (batch - 1) * strideLD + N * N : N * N);
  const size_t nY = (size_t)(strideY ? (batch - 1) * strideY + N * J : N * J);
  const size_t nX = (size_t)(batch * N * J);
  DevBuf dL, dY, dX;
  ND4_TRY(dL.alloc(h, nL * D)); ND4_TRY(dY.alloc(h, nY * D)); ND4_TRY(dX.alloc(h, nX * D));
  ND4_TRY(h2d(h, dL.p, LD, nL * D)); ND4_TRY(h2d(h, dY.p, Y, nY * D));
  ND4_TRY(nd4hip_dldltrs_batched_dev(h, batch, N, J, (const double*)dL.p, strideLD, (const double*)dY.p, strideY, (double*)dX.p));
  ND4_TRY(d2h(h, X, dX.p, nX * D));
  ND4_TRY(host_sync(h));
  return 0;
}

// ---- bidiag_decomp (bidiag.js:245-319)   
extern "C" int nd4hip_dgebrd_batched_dev(nd4hip_handle* h, int64_t batch, int64_t M, int64_t N, const double* A, double* U, double* B, double* V) {
  ND4_CHECK_ARG(h != nullptr, "nd4hip_dgebrd_batched: NULL handle");
  Nd4DeviceGuard guard(h);
  ND4_CHECK_ARG(batch >= 0 && M >= 0 && N >= 0, "nd4hip_dgebrd_batched: negative extent");
  if (batch == 0 || M == 0 || N == 0) return 0;
  ND4_CHECK_ARG(A && U && B && V, "nd4hip_dgebrd_batched: NULL pointer");
  { const int64_t K = M < N ? M : N, Jb = M >= N ? K : K + 1;
    ND4_FOR_CHUNKS(batch) ND4_TRY(nd4_gebrd(h, nb, M, N, A + b0 * M * N, U + b0 * M * K, B + b0 * K * Jb, V + b0 * Jb * N)); }
  return 0;
}
extern "C" int nd4hip_dgebrd_batched(nd4hip_handle* h, int64_t batch, int64_t M, int64_t N, const double* A, double* U, double* B, double* V) {
  ND4_CHECK_ARG(h != nullptr, "nd4hip_dgebrd_batched: NULL handle");
  Nd4DeviceGuard guard(h);
  ND4_CHECK_ARG(batch >= 0 && M >= 0 && N >= 0, "nd4hip_dgebrd_batched: negative extent");
  if (batch == 0 || M == 0 || N == 0) return 0;
  const int64_t K = M < N ? M : N, J = M >= N ? K : K + 1;
  const size_t nA = (size_t)(batch * M * N), nU = (size_t)(batch * M * K), nB = (size_t)(batch * K * J), nV = (size_t)(batch * J * N);
  DevBuf dA, dU, dB, dV;
  ND4_TRY(dA.alloc(h, nA * D)); ND4_TRY(dU.alloc(h, nU * D)); ND4_TRY(dB.alloc(h, nB * D)); ND4_TRY(dV.alloc(h, nV * D));
  ND4_TRY(h2d(h, dA.p, A, nA * D));
  ND4_TRY(nd4hip_dgebrd_batched_dev(h, batch, M, N, (const double*)dA.p, (double*)dU.p, (double*)dB.p, (double*)dV.p));
  ND4_TRY(d2h(h, U, dU.p, nU * D)); ND4_TRY(d2h(h, B, dB.p, nB * D)); ND4_TRY(d2h(h, V, dV.p, nV * D));
  ND4_TRY(host_sync(h));
  return 0;
}

// ---- hessenberg_decomp (hessenberg.js:89-115)   
extern "C" int nd4hip_dgehrd_batched_dev(nd4hip_handle* h, int64_t batch, int64_t N, const double* A, double* U, double* H) {
  ND4_CHECK_ARG(h != nullptr, "nd4hip_dgehrd_batched: NULL handle");
  Nd4DeviceGuard guard(h);
  ND4_CHECK_ARG(batch >= 0 && N >= 0, "nd4hip_dgehrd_batched: negative extent");
  if (batch == 0 || N == 0) return 0;
  ND4_CHECK_ARG(A && U && H, "nd4hip_dgehrd_batched: NULL pointer");
  ND4_FOR_CHUNKS(batch) ND4_TRY(nd4_gehrd(h, nb, N, A + b0 * N * N, U + b0 * N * N, H + b0 * N * N));
  return 0;
}
extern "C" int nd4hip_dgehrd_batched(nd4hip_handle* h, int64_t batch, int64_t N, const double* A, double* U, double* H) {
  ND4_CHECK_ARG(h != nullptr, "nd4hip_dgehrd_batched: NULL handle");
  Nd4DeviceGuard guard(h);
  ND4_CHECK_ARG(batch >= 0 && N >= 0, "nd4hip_dgehrd_batched: negative extent");
  if (batch == 0 || N == 0) return 0;
  const size_t n = (size_t)(batch * N * N);
  DevBuf dU, dH;
  ND4_TRY(dU.alloc(h, n * D)); ND4_TRY(dH.alloc(h, n * D));
  ND4_TRY(h2d(h, dH.p, A, n * D));
  ND4_TRY(nd4hip_dgehrd_batched_dev(h, batch, N, (const double*)dH.p, (double*)dU.p, (double*)dH.p));      // in place on the copy
  ND4_TRY(d2h(h, U, dU.p, n * D)); ND4_TRY(d2h(h, H, dH.p, n * D));
  ND4_TRY(host_sync(h));
  return 0;
}

// ------------------------------------------------------------------------------------ QR
extern "C" int nd4hip_dgeqrf_q_batched_dev(nd4hip_handle* h, int64_t batch, int64_t M, int64_t N, const double* A, double* Q, double* R) {
  ND4_CHECK_ARG(h != nullptr, "nd4hip_dgeqrf_q_batched: NULL handle");
  Nd4DeviceGuard guard(h);
  ND4_CHECK_ARG(batch >= 0 && M >= 0 && N >= 0, "nd4hip_dgeqrf_q_batched: negative extent");
  if (batch == 0 || M == 0 || N == 0) return 0;
  ND4_CHECK_ARG(A && Q && R, "nd4hip_dgeqrf_q_batched: NULL pointer");
  { const int64_t L = M < N ? M : N;
    ND4_FOR_CHUNKS(batch) ND4_TRY(nd4_geqrf_q(h, nb, M, N, A + b0 * M * N, Q + b0 * M * L, R + b0 * L * N)); }
  return 0;
}
extern "C" int nd4hip_dgeqrf_q_batched(nd4hip_handle* h, int64_t batch, int64_t M, int64_t N, const double* A, double* Q, double* R) {
  ND4_CHECK_ARG(h != nullptr, "nd4hip_dgeqrf_q_batched: NULL handle");
  Nd4DeviceGuard guard(h);
  ND4_CHECK_ARG(batch >= 0 && M >= 0 && N >= 0, "nd4hip_dgeqrf_q_batched: negative extent");
  if (batch == 0 || M == 0 || N == 0) return 0;
  const int64_t L = M < N ? M : N;
  const size_t nA = (size_t)(batch * M * N), nQ = (size_t)(batch * M * L), nR = (size_t)(batch * L * N);
  DevBuf dA, dQ, dR;
  ND4_TRY(dA.alloc(h, nA * D)); ND4_TRY(dQ.alloc(h, nQ * D)); ND4_TRY(dR.alloc(h, nR * D));
  ND4_TRY(h2d(h, dA.p, A, nA * D));
  ND4_TRY(nd4hip_dgeqrf_q_batched_dev(h, batch, M, N, (const double*)dA.p, (double*)dQ.p, (double*)dR.p));
  ND4_TRY(d2h(h, Q, dQ.p, nQ * D)); ND4_TRY(d2h(h, R, dR.p, nR * D));
  ND4_TRY(host_sync(h));
  return 0;
}

// qr_decomp_full (qr.js:27-77) for every shape: Q [M, M], R [M, N]
extern "C" int nd4hip_dgeqrf_full_batched_dev(nd4hip_handle* h, int64_t batch, int64_t M, int64_t N, const double* A, double* Q, double* R) {
  ND4_CHECK_ARG(h != nullptr, "nd4hip_dgeqrf_full_batched: NULL handle");
  Nd4DeviceGuard guard(h);
  ND4_CHECK_ARG(batch >= 0 && M >= 0 && N >= 0, "nd4hip_dgeqrf_full_batched: negative extent");
  if (batch == 0 || M == 0 || N == 0) return 0;
  ND4_CHECK_ARG(A && Q && R, "nd4hip_dgeqrf_full_batched: NULL pointer");
  ND4_FOR_CHUNKS(batch) ND4_TRY(nd4_geqrf_q_ex(h, nb, M, N, A + b0 * M * N, Q + b0 * M * M, R + b0 * M * N, true));
  return 0;
}
extern "C" int nd4hip_dgeqrf_full_batched(nd4hip_handle* h, int64_t batch, int64_t M, int64_t N, const double* A, double* Q, double* R) {
  ND4_CHECK_ARG(h != nullptr, "nd4hip_dgeqrf_full_batched: NULL handle");
  Nd4DeviceGuard guard(h);
  ND4_CHECK_ARG(batch >= 0 && M >= 0 && N >= 0, "nd4hip_dgeqrf_full_batched: negative extent");
  if (batch == 0 || M == 0 || N == 0) return 0;
  const size_t nA = (size_t)(batch * M * N), nQ = (size_t)(batch * M * M);
  DevBuf dA, dQ, dR;
  ND4_TRY(dA.alloc(h, nA * D)); ND4_TRY(dQ.alloc(h, nQ * D)); ND4_TRY(dR.alloc(h, nA * D));
  ND4_TRY(h2d(h, dA.p, A, nA * D));
  ND4_TRY(nd4hip_dgeqrf_full_batched_dev(h, batch, M, N, (const double*)dA.p, (double*)dQ.p, (double*)dR.p));
  ND4_TRY(d2h(h, Q, dQ.p, nQ * D)); ND4_TRY(d2h(h, R, dR.p, nA * D));
  ND4_TRY(host_sync(h));
  return 0;
}

// _qr_decomp_inplace (qr.js:146-183): A [M, N] <- R, Y [M, L] <- Q^T Y with the full (M x M) Q
extern "C" int nd4hip_dgeqrf_qty_batched_dev(nd4hip_handle* h, int64_t batch, int64_t M, int64_t N, int64_t L, double* A, double* Y) {
  ND4_CHECK_ARG(h != nullptr, "nd4hip_dgeqrf_qty_batched: NULL handle");
  Nd4DeviceGuard guard(h);
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
extern "C" int nd4hip_dgeqrf_qty_batched(nd4hip_handle* h, int64_t batch, int64_t M, int64_t N, int64_t L, double* A, double* Y) {
  ND4_CHECK_ARG(h != nullptr, "nd4hip_dgeqrf_qty_batched: NULL handle");
  Nd4DeviceGuard guard(h);
  ND4_CHECK_ARG(batch >= 0 && M >= 0 && N >= 0 && L >= 0, "nd4hip_dgeqrf_qty_batched: negative extent");
  if (batch == 0 || M == 0 || N == 0) return 0;
  const size_t nA = (size_t)(batch * M * N), nY = (size_t)(batch * M * L);
  DevBuf dA, dY;
  ND4_TRY(dA.alloc(h, nA * D)); ND4_TRY(dY.alloc(h, nY * D));
  ND4_TRY(h2d(h, dA.p, A, nA * D));
  if (nY) ND4_TRY(h2d(h, dY.p, Y, nY * D));
  ND4_TRY(nd4hip_dgeqrf_qty_batched_dev(h, batch, M, N, L, (double*)dA.p, (double*)dY.p));
  ND4_TRY(d2h(h, A, dA.p, nA * D));
  if (nY) ND4_TRY(d2h(h, Y, dY.p, nY * D));
  ND4_TRY(host_sync(h));
  return 0;
}

// ------------------------------------------------------------------------------------ SVD
extern "C" int nd4hip_dgesvdj_batched_dev(nd4hip_handle* h, int64_t batch, int64_t M, int64_t N, const double* A,
                                          double* U, double* sv, double* V, int* sweeps_out, double* offnorm_out) {
  ND4_CHECK_ARG(h != nullptr, "nd4hip_dgesvdj_batched: NULL handle");
  Nd4DeviceGuard guard(h);
  ND4_CHECK_ARG(batch >= 0 && M >= 0 && N >= 0, "nd4hip_dgesvdj_batched: negative extent");
  if (sweeps_out) *sweeps_out = 0;
  if (offnorm_out) *offnorm_out = 0.0;
  if (batch == 0 || M == 0 || N == 0) return 0;
  ND4_CHECK_ARG(A && U && sv && V, "nd4hip_dgesvdj_batched: NULL pointer");
  {
    const int64_t L = M < N ? M : N;
    int sweeps = 0; double off = 0.0;
    ND4_FOR_CHUNKS(batch) {
      int sw = 0; double of = 0.0;
      ND4_TRY(nd4_gesvdj(h, nb, M, N, A + b0 * M * N, U + b0 * M * L, sv + b0 * L, V + b0 * L * N, &sw, &of));
      if (sw > sweeps) sweeps = sw;
      if (of > off) off = of;
    }
    if (sweeps_out) *sweeps_out = sweeps;
    if (offnorm_out) *offnorm_out = off;
  }
  return 0;
}
extern "C" int nd4hip_dgesvdj_batched(nd4hip_handle* h, int64_t batch, int64_t M, int64_t N, const double* A,
                                      double* U, double* sv, double* V, int* sweeps_out, double* offnorm_out) {
  ND4_CHECK_ARG(h != nullptr, "nd4hip_dgesvdj_batched: NULL handle");
  Nd4DeviceGuard guard(h);
  ND4_CHECK_ARG(batch >= 0 && M >= 0 && N >= 0, "nd4hip_dgesvdj_batched: negative extent");
  if (batch == 0 || M == 0 || N == 0) { if (sweeps_out) *sweeps_out = 0; if (offnorm_out) *offnorm_out = 0; return 0; }
  const int64_t L = M < N ? M : N;
  const size_t nA = (size_t)(batch * M * N), nU = (size_t)(batch * M * L), nS = (size_t)(batch * L), nV = (size_t)(batch * L * N);
  DevBuf dA, dU, dS, dV;
  ND4_TRY(dA.alloc(h, nA * D)); ND4_TRY(dU.alloc(h, nU * D)); ND4_TRY(dS.alloc(h, nS * D)); ND4_TRY(dV.alloc(h, nV * D));
  ND4_TRY(h2d(h, dA.p, A, nA * D));
  ND4_TRY(nd4hip_dgesvdj_batched_dev(h, batch, M, N, (const double*)dA.p, (double*)dU.p, (double*)dS.p, (double*)dV.p,
                                     sweeps_out, offnorm_out));
  ND4_TRY(d2h(h, U, dU.p, nU * D)); ND4_TRY(d2h(h, sv, dS.p, nS * D)); ND4_TRY(d2h(h, V, dV.p, nV * D));
  ND4_TRY(host_sync(h));
  return 0;
}

extern "C" int nd4hip_dgesvdj_last_info(nd4hip_handle* h, int* sweeps, unsigned long long* rotations, double* offnorm) {
  ND4_CHECK_ARG(h != nullptr, "nd4hip_dgesvdj_last_info: NULL handle");
  if (sweeps) *sweeps = h->svd_sweeps;
  if (rotations) *rotations = h->svd_rotations;
  if (offnorm) *offnorm = h->svd_offnorm;
  return 0;
}
