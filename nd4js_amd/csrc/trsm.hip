// Batched triangular solves and lu_solve on the device (the solve-side consumers of the path).
//
// Replaces src/la/tri.js:45-95 (_tril_solve / _triu_solve, behind tril_solve / triu_solve :155-290) and
// src/la/lu.js:84-177 (lu_solve: X = Y[P,:], forward substitution with the unit-lower part, _triu_solve).
// Blocked substitution, block = 32 rows:
//   tri_block_solve  the 32x32 diagonal block in LDS (broadcast reads), one thread per right-hand-side column
//                    with its 32 unknowns in registers, true division by the diagonal like the reference;
//   nd4_gemm         X[other rows] -= T[other rows, block] * X[block]   (fp64 MFMA, K = 32).
// Everything is in place on X (initialised with Y, or with the row-gathered Y[P,:] for lu_solve).
#include "nd4hip_internal.h"

namespace {

constexpr int TB = 32;

template <bool UPPER>
__global__ __launch_bounds__(256) void tri_block_solve(const double* __restrict__ Tm, int M, long sT, double* __restrict__ Xm, int J, long sX,
                                                        int r0, int nbt, int unit, int trans) {   // M = leading dimension of T
  // trans: the triangle is the TRANSPOSE of what is stored (UPPER + trans = L^T of a stored lower L, tri.js:100-125)
  __shared__ double s_t[TB][TB + 1];
  const double* T = Tm + blockIdx.y * sT;
  double* X = Xm + blockIdx.y * sX;
  const int t = threadIdx.x;
  for (int e = t; e < TB * TB; e += 256) {
    const int i = e / TB, j = e % TB;
    double v = (i == j) ? 1.0 : 0.0;                               // identity padding beyond nbt
    if (i < nbt && j < nbt) {
      const bool tri = UPPER ? (j >= i) : (j <= i);
      v = tri ? (trans ? T[(long)(r0 + j) * M + r0 + i] : T[(long)(r0 + i) * M + r0 + j]) : 0.0;
      if (unit && i == j) v = 1.0;
    }
    s_t[i][j] = v;
  }
  __syncthreads();
  const int col = blockIdx.x * 256 + t;
  if (col >= J) return;
  double x[TB];
#pragma unroll
  for (int i = 0; i < TB; i++) x[i] = (i < nbt) ? X[(long)(r0 + i) * J + col] : 0.0;
  if (UPPER) {
#pragma unroll
    for (int i = TB - 1; i >= 0; i--) {                            // tri.js:87-94 (k descending)
      double s = x[i];
#pragma unroll
      for (int k = TB - 1; k > i; k--) s -= s_t[i][k] * x[k];
      x[i] = s / s_t[i][i];
    }
  } else {
#pragma unroll
    for (int i = 0; i < TB; i++) {                                 // tri.js:61-70 (k ascending)
      double s = x[i];
#pragma unroll
      for (int k = 0; k < i; k++) s -= s_t[i][k] * x[k];
      x[i] = unit ? s : s / s_t[i][i];
    }
  }
#pragma unroll
  for (int i = 0; i < TB; i++)
    if (i < nbt) X[(long)(r0 + i) * J + col] = x[i];
}

// X[i,:] = Y[P[i],:]   (lu.js:131-136)
__global__ void gather_rows(const double* __restrict__ Ym, long sY, const int32_t* __restrict__ Pm, long sP, double* __restrict__ Xm,
                            int N, int J) {
  const double* Y = Ym + blockIdx.z * sY; const int32_t* P = Pm + blockIdx.z * sP; double* X = Xm + blockIdx.z * (long)N * J;
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= J) return;
  for (int i = blockIdx.y; i < N; i += gridDim.y) {
    const int src = P[i];
    X[(long)i * J + col] = (src >= 0 && src < N) ? Y[(long)src * J + col] : 0.0;
  }
}

}  // namespace

// in place: X <- T^-1 X ; T = leading M x M block of [batch (stride sT, 0 = broadcast)] matrices with leading
// dimension ldT, X = first M rows of [batch (stride sX)] row-major blocks with J columns
int nd4_trsm_ld(nd4hip_handle* h, bool upper, bool unit, int64_t batch, int64_t M64, int64_t J64, const double* T, int64_t ldT64, int64_t sT,
                double* X, int64_t sX64) {
  ND4_CHECK_ARG(M64 < (1ll << 30) && J64 < (1ll << 30) && batch < 65536, "nd4_trsm: extent out of range");
  const int M = (int)M64, J = (int)J64, ldT = (int)ldT64;
  if (M == 0 || J == 0 || batch == 0) return 0;
  const long sX = (long)sX64;
  const dim3 grid((unsigned)((J + 255) / 256), (unsigned)batch);
  const int nblocks = (M + TB - 1) / TB;
  for (int bi = 0; bi < nblocks; bi++) {
    const int b = upper ? nblocks - 1 - bi : bi;
    const int r0 = b * TB, nbt = M - r0 < TB ? M - r0 : TB;
    if (upper) hipLaunchKernelGGL(tri_block_solve<true>, grid, dim3(256), 0, h->stream, T, ldT, (long)sT, X, J, sX, r0, nbt, unit ? 1 : 0, 0);
    else       hipLaunchKernelGGL(tri_block_solve<false>, grid, dim3(256), 0, h->stream, T, ldT, (long)sT, X, J, sX, r0, nbt, unit ? 1 : 0, 0);
    ND4_HIP(hipGetLastError());
    if (upper) {
      if (r0 > 0)      // rows above the block
        ND4_TRY(nd4_gemm(h, false, false, r0, J, nbt, -1.0, T + r0, ldT, sT, X + (long)r0 * J, J, sX, 1.0, X, J, sX, batch));
    } else {
      const int below = M - r0 - nbt;
      if (below > 0)   // rows below the block
        ND4_TRY(nd4_gemm(h, false, false, below, J, nbt, -1.0, T + (long)(r0 + nbt) * ldT + r0, ldT, sT, X + (long)r0 * J, J, sX,
                         1.0, X + (long)(r0 + nbt) * J, J, sX, batch));
    }
  }
  return 0;
}

// in place: X <- L^-T X for a stored LOWER triangle L (non-unit diagonal): _tril_t_solve, tri.js:100-125
int nd4_trsm_t(nd4hip_handle* h, int64_t batch, int64_t M, int64_t J, const double* T, int64_t ldT, int64_t sT, double* X, int64_t sX) {
  return nd4_trsm_t_ex(h, false, batch, M, J, T, ldT, sT, X, sX);
}
// unit = true: the diagonal of L is taken as ones (packed LD of ldl_decomp, ldl.js:125-129)
int nd4_trsm_t_ex(nd4hip_handle* h, bool unit, int64_t batch, int64_t M64, int64_t J64, const double* T, int64_t ldT64, int64_t sT, double* X, int64_t sX64) {
  ND4_CHECK_ARG(M64 < (1ll << 30) && J64 < (1ll << 30) && batch < 65536, "nd4_trsm_t: extent out of range");
  const int M = (int)M64, J = (int)J64, ldT = (int)ldT64;
  if (M == 0 || J == 0 || batch == 0) return 0;
  const long sX = (long)sX64;
  const dim3 grid((unsigned)((J + 255) / 256), (unsigned)batch);
  const int nblocks = (M + TB - 1) / TB;
  for (int b = nblocks - 1; b >= 0; b--) {
    const int r0 = b * TB, nbt = M - r0 < TB ? M - r0 : TB;
    hipLaunchKernelGGL(tri_block_solve<true>, grid, dim3(256), 0, h->stream, T, ldT, (long)sT, X, J, sX, r0, nbt, unit ? 1 : 0, 1);
    ND4_HIP(hipGetLastError());
    if (r0 > 0)      // rows above: X[0:r0] -= L[block, 0:r0]^T X[block]
      ND4_TRY(nd4_gemm(h, true, false, r0, J, nbt, -1.0, T + (long)r0 * ldT, ldT, sT, X + (long)r0 * J, J, sX, 1.0, X, J, sX, batch));
  }
  return 0;
}

int nd4_trsm(nd4hip_handle* h, bool upper, bool unit, int64_t batch, int64_t M, int64_t J, const double* T, int64_t sT, double* X) {
  return nd4_trsm_ld(h, upper, unit, batch, M, J, T, M, sT, X, M * J);
}

// qr_lstsq (qr.js:186-273): X [batch, I, J] = R[0:L,0:L]^-1 (Q^T Y)[0:L], L = min(M, I), rows L..I-1 stay 0.
// Q [N, M], R [M, I], Y [N, J]; strides 0 = broadcast.
int nd4_qrls(nd4hip_handle* h, int64_t batch, int64_t N, int64_t M, int64_t I, int64_t J, const double* Q, int64_t sQ,
             const double* R, int64_t sR, const double* Y, int64_t sY, double* X) {
  const int64_t L = M < I ? M : I;
  if (L < I) ND4_HIP(hipMemsetAsync(X, 0, sizeof(double) * (size_t)(batch * I * J), h->stream));
  // (Q^T Y)[0:L]: A = Q stored N x M (operand = its transpose, first L columns), K = N    (qr.js:236-239)
  ND4_TRY(nd4_gemm(h, true, false, L, J, N, 1.0, Q, M, sQ, Y, J, sY, 0.0, X, J, I * J, batch));
  return nd4_trsm_ld(h, true, false, batch, L, J, R, I, sR, X, I * J);                      // qr.js:241
}

// per matrix: rank = first r with |sv_r| <= sqrt(eps) |sv_0| (svd.js:165-177); tmp[i,:] /= sv_i for i < rank, = 0 otherwise
namespace {
__global__ __launch_bounds__(256) void svdls_scale(double* __restrict__ tmp, int M, int J, const double* __restrict__ svm, long sSv) {
  const double* sv = svm + blockIdx.z * sSv;
  double* t = tmp + blockIdx.z * (long)M * J;
  __shared__ int s_rank;
  if (threadIdx.x == 0) {
    const double T = 1.4901161193847656e-08 * fabs(sv[0]);              // Math.sqrt(2^-52)
    int r = 0;
    while (r < M && fabs(sv[r]) > T) r++;
    s_rank = r;
  }
  __syncthreads();
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (col >= J) return;
  for (int i = blockIdx.y; i < M; i += gridDim.y)
    t[(long)i * J + col] = (i < s_rank) ? t[(long)i * J + col] / sv[i] : 0.0;
}
}  // namespace

// svd_lstsq (svd.js:100-228): X [batch, I, J] = V[0:r,:]^T diag(1/sv[0:r]) U[:,0:r]^T Y ; U [N, M], sv [M], V [M, I], Y [N, J]
int nd4_svdls(nd4hip_handle* h, int64_t batch, int64_t N, int64_t M, int64_t I, int64_t J, const double* U, int64_t sU,
              const double* sv, int64_t sSv, const double* V, int64_t sV, const double* Y, int64_t sY, double* X) {
  Nd4WsScope scope(h);
  void* p = nullptr;
  ND4_TRY(nd4_ws_alloc(h, sizeof(double) * (size_t)(batch * M * J), &p));
  double* tmp = static_cast<double*>(p);
  ND4_TRY(nd4_gemm(h, true, false, M, J, N, 1.0, U, M, sU, Y, J, sY, 0.0, tmp, J, M * J, batch));     // U^T Y  (svd.js:184-189)
  const unsigned gy = (unsigned)(M < 1024 ? M : 1024);
  hipLaunchKernelGGL(svdls_scale, dim3((unsigned)((J + 255) / 256), gy, (unsigned)batch), dim3(256), 0, h->stream, tmp, (int)M, (int)J, sv, (long)sSv);
  ND4_HIP(hipGetLastError());
  return nd4_gemm(h, true, false, I, J, M, 1.0, V, I, sV, tmp, J, M * J, 0.0, X, J, I * J, batch);     // V^T tmp (svd.js:197-201)
}

int nd4_getrs(nd4hip_handle* h, int64_t batch, int64_t N, int64_t J, const double* LU, int64_t sLU, const int32_t* P, int64_t sP,
              const double* Y, int64_t sY, double* X) {
  ND4_CHECK_ARG(N < (1ll << 30) && J < (1ll << 30) && batch < 65536, "nd4_getrs: extent out of range");
  if (N == 0 || J == 0 || batch == 0) return 0;
  const unsigned gy = (unsigned)(N < 1024 ? N : 1024);
  hipLaunchKernelGGL(gather_rows, dim3((unsigned)((J + 255) / 256), gy, (unsigned)batch), dim3(256), 0, h->stream,
                     Y, (long)sY, P, (long)sP, X, (int)N, (int)J);
  ND4_HIP(hipGetLastError());
  ND4_TRY(nd4_trsm(h, false, true, batch, N, J, LU, sLU, X));      // L (unit diagonal) : lu.js:139-142
  return nd4_trsm(h, true, false, batch, N, J, LU, sLU, X);        // U : lu.js:145
}
