// Batched Cholesky factorisation and solve on the device.
//
// Replaces src/la/cholesky.js:51-71 (cholesky_decomp; row-wise Cholesky-Banachiewicz with Kahan sums, :27-48) and
// :74-150 (cholesky_solve = _tril_solve + _tril_t_solve, tri.js:45-71 / :100-125). S = L L^T with a positive diagonal is
// unique, so the blocked right-looking factorisation below agrees with the reference to rounding:
//   chol_diag   one wave per matrix: the 32x32 diagonal block, one row per lane in registers, column by column
//               (sqrt and true division like cholesky.js:40-41), pivot/column broadcast through LDS;
//   chol_trsm   rows below the block: one thread per row, x L_kk^T = a with L_kk in LDS (broadcast reads);
//   nd4_syrk_lower  trailing update A22 -= L21 L21^T on fp64 MFMA (NT form, K = 32); workgroups whose 128x128 tile
//               lies strictly above the diagonal exit at once;
//   chol_finish zero the strict upper triangle (the reference returns exact zeros there, cholesky.js:63-68).
// Only the lower triangle of S is read (cholesky.js:65-67). A NaN pivot (negative or NaN radicand) raises the
// per-matrix flag that the entry point turns into the reference's 'Matrix contains NaNs or is (near) singular.'
#include "nd4hip_internal.h"

namespace {

constexpr int CB = 32;            // block size
typedef double d2 __attribute__((ext_vector_type(2)));

// L <- tril(S), strict upper <- 0
__global__ void chol_copy_lower(const double* __restrict__ Sm, double* __restrict__ Lm, int N) {
  const long base = (long)blockIdx.z * N * N;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= N) return;
  for (int i = blockIdx.y; i < N; i += gridDim.y) Lm[base + (long)i * N + j] = (j <= i) ? Sm[base + (long)i * N + j] : 0.0;
}

__global__ __launch_bounds__(64) void chol_diag(double* __restrict__ Lm, int N, int j0, int nb, int* __restrict__ flags) {
  double* A = Lm + (long)blockIdx.x * N * N + (long)j0 * N + j0;
  __shared__ double s_col[CB];
  const int i = threadIdx.x;                       // row of the block (lanes >= nb idle but keep the barriers)
  double a[CB];
#pragma unroll
  for (int k = 0; k < CB; k++) a[k] = (i < nb && k <= i && k < nb) ? A[(long)i * N + k] : 0.0;
  bool bad = false;
#pragma unroll
  for (int j = 0; j < CB; j++) {
    if (j < nb) {
      if (i == j) { const double d = sqrt(a[j]); bad = bad || (d != d); a[j] = d; s_col[j] = d; }
      __syncthreads();
      const double d = s_col[j];
      __syncthreads();
      if (i > j && i < nb) a[j] = a[j] / d;
      if (i < CB) s_col[i] = a[j];                 // column j of L (rows <= j hold their own earlier values: unused)
      __syncthreads();
      if (i > j && i < nb) {
#pragma unroll
        for (int k = j + 1; k < CB; k++)
          if (k <= i) a[k] -= a[j] * s_col[k];
      }
      __syncthreads();
    }
  }
  if (i < nb) {
#pragma unroll
    for (int k = 0; k < CB; k++)
      if (k < nb) A[(long)i * N + k] = (k <= i) ? a[k] : 0.0;
  }
  if (bad) atomicOr(&flags[blockIdx.x], 1);
}

// rows r in [j0+nb, N): L[r, j0:j0+nb] = A[r, j0:j0+nb] L_kk^-T
__global__ __launch_bounds__(256) void chol_trsm(double* __restrict__ Lm, int N, int j0, int nb) {
  double* Lb = Lm + (long)blockIdx.y * N * N;
  __shared__ double s_l[CB][CB + 1];
  const int t = threadIdx.x;
  for (int e = t; e < CB * CB; e += 256) {
    const int i = e / CB, j = e % CB;
    s_l[i][j] = (i < nb && j < nb && j <= i) ? Lb[(long)(j0 + i) * N + j0 + j] : (i == j ? 1.0 : 0.0);
  }
  __syncthreads();
  const int r = j0 + nb + blockIdx.x * 256 + t;
  if (r >= N) return;
  double* row = Lb + (long)r * N + j0;
  double x[CB];
  const bool vec = nb == CB && (N & 1) == 0;       // full block and 16-byte aligned rows
  if (vec) {
#pragma unroll
    for (int q = 0; q < CB / 2; q++) { const d2 v = *reinterpret_cast<const d2*>(row + 2 * q); x[2 * q] = v.x; x[2 * q + 1] = v.y; }
  } else {
#pragma unroll
    for (int k = 0; k < CB; k++) x[k] = (k < nb) ? row[k] : 0.0;
  }
#pragma unroll
  for (int j = 0; j < CB; j++) {
    double s = x[j];
#pragma unroll
    for (int k = 0; k < j; k++) s -= x[k] * s_l[j][k];
    x[j] = s / s_l[j][j];
    // x[j] passes through an empty asm that also "touches" the LDS tile: pins this column's arithmetic before the next
    // column's LDS reads (otherwise all 528 tile reads are issued up front and spill to scratch)
    asm volatile("" : "+v"(x[j]) : "v"(&s_l[0][0]) : "memory");
  }
  if (vec) {
#pragma unroll
    for (int q = 0; q < CB / 2; q++) { d2 v; v.x = x[2 * q]; v.y = x[2 * q + 1]; *reinterpret_cast<d2*>(row + 2 * q) = v; }
  } else {
#pragma unroll
    for (int k = 0; k < CB; k++)
      if (k < nb) row[k] = x[k];
  }
}

__global__ void chol_finish(double* __restrict__ Lm, int N) {
  const long base = (long)blockIdx.z * N * N;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= N) return;
  for (int i = blockIdx.y; i < N && i < j; i += gridDim.y) Lm[base + (long)i * N + j] = 0.0;
}

}  // namespace

// S [batch, N, N] -> L [batch, N, N]; flags [batch] int (device), set to 1 where a pivot was NaN
int nd4_potrf(nd4hip_handle* h, int64_t batch64, int64_t N64, const double* S, double* L, int* flags) {
  ND4_CHECK_ARG(N64 < (1ll << 30) && batch64 < 65536, "nd4_potrf: extent out of range");
  const int N = (int)N64, batch = (int)batch64;
  if (N == 0 || batch == 0) return 0;
  const long sL = (long)N * N;
  const unsigned gy = (unsigned)(N < 1024 ? N : 1024);
  ND4_HIP(hipMemsetAsync(flags, 0, sizeof(int) * (size_t)batch, h->stream));
  hipLaunchKernelGGL(chol_copy_lower, dim3((unsigned)((N + 255) / 256), gy, (unsigned)batch), dim3(256), 0, h->stream, S, L, N);
  for (int j0 = 0; j0 < N; j0 += CB) {
    const int nb = N - j0 < CB ? N - j0 : CB;
    hipLaunchKernelGGL(chol_diag, dim3((unsigned)batch), dim3(64), 0, h->stream, L, N, j0, nb, flags);
    const int m2 = N - j0 - nb;
    if (m2 <= 0) break;
    hipLaunchKernelGGL(chol_trsm, dim3((unsigned)((m2 + 255) / 256), (unsigned)batch), dim3(256), 0, h->stream, L, N, j0, nb);
    ND4_HIP(hipGetLastError());
    // A22 -= L21 L21^T: one launch, 128x128 tiles strictly above the diagonal are skipped
    const int r0 = j0 + nb;
    ND4_TRY(nd4_syrk_lower(h, N - r0, nb, -1.0, L + (long)r0 * N + j0, N, sL, 1.0, L + (long)r0 * N + r0, N, sL, batch));
  }
  hipLaunchKernelGGL(chol_finish, dim3((unsigned)((N + 255) / 256), gy, (unsigned)batch), dim3(256), 0, h->stream, L, N);
  ND4_HIP(hipGetLastError());
  return 0;
}

// X [batch, N, J] = L^-T L^-1 Y (cholesky.js:117-123); strides 0 = broadcast
int nd4_potrs(nd4hip_handle* h, int64_t batch, int64_t N, int64_t J, const double* L, int64_t sL, const double* Y, int64_t sY, double* X) {
  if (N == 0 || J == 0 || batch == 0) return 0;
  if (X != Y || sY != N * J) ND4_TRY(nd4_copy_matrix(h, N, J, Y, J, X, J, batch, sY, N * J));
  ND4_TRY(nd4_trsm_ld(h, false, false, batch, N, J, L, N, sL, X, N * J));          // _tril_solve   (tri.js:45-71)
  return nd4_trsm_t(h, batch, N, J, L, N, sL, X, N * J);                           // _tril_t_solve (tri.js:100-125)
}
