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
//
// LDL^T (src/la/ldl.js:47-201, no pivoting, D may be indefinite) shares the structure: ldl_diag / ldl_trsm keep the
// UNSCALED panel W = L21 D11 next to L21 so that the trailing update is A22 -= W L21^T on the same tile-skipping GEMM.
#include "nd4hip_internal.h"
#include "dpp.h"
#include <type_traits>

namespace {

constexpr int CB = 32;            // block size
typedef double d2 __attribute__((ext_vector_type(2)));
typedef double d4 __attribute__((ext_vector_type(4)));

// L <- tril(S), strict upper <- 0
__global__ void chol_copy_lower(const double* __restrict__ Sm, double* __restrict__ Lm, int N) {
  const long base = (long)blockIdx.z * N * N;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= N) return;
  for (int i = blockIdx.y; i < N; i += gridDim.y) Lm[base + (long)i * N + j] = (j <= i) ? Sm[base + (long)i * N + j] : 0.0;
}

// One wave: lane i keeps row i of the block in registers. Column j: the pivot and then column j of L reach the other lanes as
// wave-uniform v_readlane values (SGPRs) - no LDS, no barrier (the LDS form with four barriers per column took 15.4 us per block, all of
// it latency). Entries right of the diagonal are scratch (never stored).
__device__ __forceinline__ double chol_rl(double v, int lane) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
__device__ __forceinline__ void chol_diag_body(const int mat, double* __restrict__ Lm, int N, int j0, int nb, int* __restrict__ flags,
                                               double* __restrict__ invout = nullptr) {
  double* A = Lm + (long)mat * N * N + (long)j0 * N + j0;
  const int i = threadIdx.x;                       // row of the block (lanes >= nb idle)
  double a[CB];
#pragma unroll
  for (int k = 0; k < CB; k++) a[k] = (i < nb && k <= i && k < nb) ? A[(long)i * N + k] : 0.0;
  bool bad = false;
  auto column = [&](auto jc) {
    constexpr int j = decltype(jc)::value;
    if (j < nb) {                                  // uniform
      // every lane forms 1/sqrt of the pivot itself (rsq + two Newton steps, ~12 dependent instructions instead of sqrt ~30 +
      // division ~35 on the column's critical path). d = x * rsqrt(x) and l = a / d = a * rsqrt(x) agree with sqrt / division to a
      // few ulp (parity is to 1e-14 cond, the reference sums with Kahan anyway); a negative or NaN radicand still gives NaN (the flag).
      // The Newton steps turn rsq(0) = inf and rsq(inf) = 0 into NaN, so those two radicands take the reference's own formulas
      // (cholesky.js:40-41): d = sqrt(x), l = a / d, i.e. d = +-0 with +-Inf (NaN for 0 / 0) below it, and d = Inf with zeros below it.
      const double x = chol_rl(a[j], j);
      double ri = nd4dpp::fast_rsqrt(x), d = x * ri;
      if (__builtin_expect(x == 0.0 || __builtin_isinf(x), 0)) { d = sqrt(x); ri = 1.0 / d; }
      if (i == j) { bad = bad || (d != d); a[j] = d; }
      if (i > j) a[j] = a[j] * ri;
      const double aj = (i > j) ? a[j] : 0.0;      // rows <= j are finished
#pragma unroll
      for (int k = j + 1; k < CB; k++) a[k] -= aj * chol_rl(a[j], k);      // L[k][j] from lane k (lanes >= nb hold zeros)
    }
  };
#define ND4_CC(J) column(std::integral_constant<int, J>{});
  ND4_CC(0) ND4_CC(1) ND4_CC(2) ND4_CC(3) ND4_CC(4) ND4_CC(5) ND4_CC(6) ND4_CC(7) ND4_CC(8) ND4_CC(9) ND4_CC(10) ND4_CC(11) ND4_CC(12) ND4_CC(13) ND4_CC(14) ND4_CC(15)
  ND4_CC(16) ND4_CC(17) ND4_CC(18) ND4_CC(19) ND4_CC(20) ND4_CC(21) ND4_CC(22) ND4_CC(23) ND4_CC(24) ND4_CC(25) ND4_CC(26) ND4_CC(27) ND4_CC(28) ND4_CC(29) ND4_CC(30) ND4_CC(31)
#undef ND4_CC
  if (i < nb) {
#pragma unroll
    for (int k = 0; k < CB; k++)
      if (k < nb) A[(long)i * N + k] = (k <= i) ? a[k] : 0.0;
  }
  if (bad) atomicOr(&flags[mat], 1);
  if (invout != nullptr && i < CB) {
    // L_kk^-1 for chol_trsm_narrow (rows below the block by MFMA): lane j solves L y = e_j, the entries of L arrive as readlane
    // values. Identity beyond nb.
    double y[CB];
#pragma unroll
    for (int r = 0; r < CB; r++) {
      double acc = (r == i) ? 1.0 : 0.0;
#pragma unroll
      for (int k = 0; k < r; k++) acc -= chol_rl(a[k], r) * y[k];
      const double d = chol_rl(a[r], r);
      y[r] = (r < nb) ? acc / d : acc;
    }
#pragma unroll
    for (int r = 0; r < CB; r++) invout[(long)mat * CB * CB + r * CB + i] = y[r];
  }
}
// (the LDL^T diagonal block: defined here because the look-ahead kernels below are shared with it)
__device__ __forceinline__ void ldl_diag_body(const int mat, double* __restrict__ Lm, int N, int j0, int nb, double* __restrict__ invout = nullptr) {
  double* A = Lm + (long)mat * N * N + (long)j0 * N + j0;
  const int i = threadIdx.x;
  double a[CB];
#pragma unroll
  for (int k = 0; k < CB; k++) a[k] = (i < nb && k <= i && k < nb) ? A[(long)i * N + k] : 0.0;
  // as chol_diag_body: D_j and the unscaled column V = L[:,j] D_j (ldl.js:52) reach the other lanes as v_readlane values
  auto column = [&](auto jc) {
    constexpr int j = decltype(jc)::value;
    if (j < nb) {
      const double d = chol_rl(a[j], j);            // D_j (ldl.js:58-59 divides by LD[j,j])
      const bool below = i > j && i < nb;
      const double l = below ? a[j] / d : 0.0;
#pragma unroll
      for (int k = j + 1; k < CB; k++) a[k] -= l * chol_rl(a[j], k);      // V_k from lane k, still unscaled there
      if (below) a[j] = l;
    }
  };
#define ND4_CC(J) column(std::integral_constant<int, J>{});
  ND4_CC(0) ND4_CC(1) ND4_CC(2) ND4_CC(3) ND4_CC(4) ND4_CC(5) ND4_CC(6) ND4_CC(7) ND4_CC(8) ND4_CC(9) ND4_CC(10) ND4_CC(11) ND4_CC(12) ND4_CC(13) ND4_CC(14) ND4_CC(15)
  ND4_CC(16) ND4_CC(17) ND4_CC(18) ND4_CC(19) ND4_CC(20) ND4_CC(21) ND4_CC(22) ND4_CC(23) ND4_CC(24) ND4_CC(25) ND4_CC(26) ND4_CC(27) ND4_CC(28) ND4_CC(29) ND4_CC(30) ND4_CC(31)
#undef ND4_CC
  if (i < nb) {
#pragma unroll
    for (int k = 0; k < CB; k++)
      if (k < nb) A[(long)i * N + k] = (k <= i) ? a[k] : 0.0;
  }
  if (invout != nullptr && i < CB) {                   // inverse of the UNIT lower block (no divisions), as in chol_diag_body
    double y[CB];
#pragma unroll
    for (int r = 0; r < CB; r++) {
      double acc = (r == i) ? 1.0 : 0.0;
#pragma unroll
      for (int k = 0; k < r; k++) acc -= chol_rl(a[k], r) * y[k];
      y[r] = acc;
    }
#pragma unroll
    for (int r = 0; r < CB; r++) invout[(long)mat * CB * CB + r * CB + i] = y[r];
  }
}
__global__ __launch_bounds__(64) void ldl_diag(double* __restrict__ Lm, int N, int j0, int nb) { ldl_diag_body(blockIdx.x, Lm, N, j0, nb); }

// L21 of the 16 rows from row0 (block column j0) as the transposed accumulator image lo = columns 0..15, hi = 16..31: lane (fx, fk),
// register r holds L21[row0 + fx][fk + 4 r (+16)] - at once the A operand (own rows) and the B operand (rows of the next block) of
// A[r, next block] -= L21[r] L21[next block]^T. k-steps run over the columns in the order 8 q + 2 fk + e: 16 contiguous bytes per load.
__device__ __forceinline__ void chol_dprime(const double* __restrict__ Lb, int N, int j0, int row0, const double (*s_inv)[CB + 1],
                                            int fx, int fk, d4& lo, d4& hi) {
  double bk[8];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const d2 v = *reinterpret_cast<const d2*>(Lb + (long)(row0 + fx) * N + j0 + 8 * q + 2 * fk);
    bk[2 * q] = v.x; bk[2 * q + 1] = v.y;
  }
  lo = d4{0.0, 0.0, 0.0, 0.0}; hi = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int ks = 0; ks < 8; ks++) {
    const int kap = 8 * (ks >> 1) + 2 * fk + (ks & 1);
    lo = __builtin_amdgcn_mfma_f64_16x16x4f64(s_inv[fx][kap], bk[ks], lo, 0, 0, 0);
    hi = __builtin_amdgcn_mfma_f64_16x16x4f64(s_inv[16 + fx][kap], bk[ks], hi, 0, 0, 0);
  }
}
__global__ __launch_bounds__(64) void chol_diag(double* __restrict__ Lm, int N, int j0, int nb, int* __restrict__ flags) {
  chol_diag_body(blockIdx.x, Lm, N, j0, nb, flags);
}

// ---- look-ahead: the diagonal block of step k (one wave, 15 us of latency) in the same launch as the trailing update of step k-1 ----
// Workgroup 0 of a matrix factorises A[j0:j0+nb, j0:j0+nb]; the others apply A22 -= L21 L21^T of the PREVIOUS block column (at pj0) to the
// lower tiles from block column j0 + CB on: 64 x 32 tiles, a wave owns 16 x 32, operands straight from global memory in MFMA layout (NT:
// both are '4 consecutive doubles of a row'), accumulators start at C. Block column j0 itself got that update between the two launches
// (the narrow GEMM in nd4_potrf), so the trailing update is off the critical path: per block diag + trsm + narrow instead of diag +
// trsm + full update.
// one 64 x 32 tile of the trailing update A22 -= L21 L21^T of the block column at pj0, tiles counted from row / column c1.
// (Sharing the tiles between the two launches of a step was tried: 1.63 -> 1.86 ms at 2048^2 - chol_diag_la is bound by its
// one-wave chain of diagonal block + inverse, not by the tiles it carries.)
template <bool LDL>
__device__ __forceinline__ void chol_wide_tile(double* __restrict__ Lb, int N, int pj0, int c1, int tile, int ntc) {
  const int t = threadIdx.x;
  const int tr = tile / ntc, tc = tile % ntc;
  const int row0 = c1 + tr * 64, col0 = c1 + tc * 32;
  if (col0 > row0 + 63) return;                             // strictly above the diagonal
  const int lane = t & 63, w = t >> 6, fx = lane & 15, fk = lane >> 4;
  const int rb = row0 + 16 * w;
  if (rb >= N) return;
  d4 acc[2];
#pragma unroll
  for (int j = 0; j < 2; j++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int row = rb + fk + 4 * r, col = col0 + 16 * j + fx;
      acc[j][r] = (row < N && col < N) ? Lb[(long)row * N + col] : 0.0;
    }
  double a[8], b[2][8];
#pragma unroll
  for (int kk = 0; kk < 8; kk++) {
    const int ra = rb + fx;
    a[kk] = (ra < N) ? -Lb[(long)ra * N + pj0 + kk * 4 + fk] : 0.0;
    if (LDL) a[kk] *= Lb[(long)(pj0 + kk * 4 + fk) * N + pj0 + kk * 4 + fk];       // (L21 D11) L21^T: column c of L21 times d_c
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int cr = col0 + 16 * j + fx;                    // row of L that is column cr of L^T
      b[j][kk] = (cr < N) ? Lb[(long)cr * N + pj0 + kk * 4 + fk] : 0.0;
    }
  }
#pragma unroll
  for (int kk = 0; kk < 8; kk++)
#pragma unroll
    for (int j = 0; j < 2; j++) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kk], b[j][kk], acc[j], 0, 0, 0);
#pragma unroll
  for (int j = 0; j < 2; j++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int row = rb + fk + 4 * r, col = col0 + 16 * j + fx;
      if (row < N && col < N) Lb[(long)row * N + col] = acc[j][r];
    }
}

template <bool LDL>
__global__ __launch_bounds__(256) void chol_diag_la(double* __restrict__ Lm, int N, int j0, int nb, int* __restrict__ flags, int pj0, int ntc,
                                                     double* __restrict__ invout, double* __restrict__ ynext) {
  const int mat = blockIdx.y, t = threadIdx.x;
  if (blockIdx.x == 0) {
    if (t >= 64) return;
    if (LDL) ldl_diag_body(mat, Lm, N, j0, nb, invout); else chol_diag_body(mat, Lm, N, j0, nb, flags, invout);
    if (ynext != nullptr && j0 + 2 * CB <= N) {
      // L21 of the NEXT block's 32 rows, for every wave of chol_trsm_narrow (which must not read those rows of A21 itself: their owner
      // overwrites them in the same launch). One wave: the inverse goes through LDS into MFMA operand layout.
      __shared__ double s_inv[CB][CB + 1];
      __threadfence_block();
      for (int e = t; e < CB * CB; e += 64) s_inv[e / CB][e % CB] = invout[(long)mat * CB * CB + e];
      __syncthreads();
      const int fx = t & 15, fk = t >> 4;
      double* yo = ynext + (long)mat * 16 * 64;
#pragma unroll
      for (int jt = 0; jt < 2; jt++) {
        d4 lo, hi;
        chol_dprime(Lm + (long)mat * N * N, N, j0, j0 + CB + 16 * jt, s_inv, fx, fk, lo, hi);
        if (LDL) {                                       // image of W = L21 D11 -> image of L21: column fk + 4 r (+16) divided by its d
          const double* Ld = Lm + (long)mat * N * N;
#pragma unroll
          for (int r = 0; r < 4; r++) {
            lo[r] = lo[r] / Ld[(long)(j0 + fk + 4 * r) * N + j0 + fk + 4 * r];
            hi[r] = hi[r] / Ld[(long)(j0 + 16 + fk + 4 * r) * N + j0 + 16 + fk + 4 * r];
          }
        }
#pragma unroll
        for (int r = 0; r < 4; r++) { yo[((jt * 2 + 0) * 4 + r) * 64 + t] = lo[r]; yo[((jt * 2 + 1) * 4 + r) * 64 + t] = hi[r]; }
      }
    }
    return;
  }
  chol_wide_tile<LDL>(Lm + (long)mat * N * N, N, pj0, j0 + CB, (int)blockIdx.x - 1, ntc);
}

// rows below a full block (N a multiple of 32): L21 = A21 L_kk^-T on fp64 MFMA with the inverted diagonal block, AND the update of the
// next block column, A[r, j1:j1+32] -= L21[r] L21[j1:j1+32]^T, in the same launch. A wave owns 16 rows. It forms D' = inv A_rows^T: the
// accumulator of D' holds L21[row fx][fk + 4 r (+16)], which is both the A operand of the update (its own rows) and - computed
// once more for the 32 rows of the next block, which every wave needs - the B operand: no LDS, no second launch, no dependence on
// another workgroup. k-steps run over the columns in the order 8 q + 2 fk + e so that a lane reads 16 contiguous bytes per load.
template <bool LDL>
__global__ __launch_bounds__(256) void chol_trsm_narrow(double* __restrict__ Lm, int N, int j0, const double* __restrict__ invm,
                                                         const double* __restrict__ ynext) {
  __shared__ double s_inv[CB][CB + 1];
  const int mat = blockIdx.y, t = threadIdx.x, lane = t & 63, w = t >> 6, fx = lane & 15, fk = lane >> 4;
  double* Lb = Lm + (long)mat * N * N;
  for (int e = t; e < CB * CB; e += 256) s_inv[e / CB][e % CB] = invm[(long)mat * CB * CB + e];
  __syncthreads();
  const int j1 = j0 + CB;
  const int r0 = j1 + ((int)blockIdx.x * 4 + w) * 16;
  if (r0 >= N) return;
  d4 x0, x1, y[2][2];
  chol_dprime(Lb, N, j0, r0, s_inv, fx, fk, x0, x1);
  {                                                      // the next block's rows: from chol_diag_la (their owner overwrites them here)
    const double* yi = ynext + (long)mat * 16 * 64;
#pragma unroll
    for (int jt = 0; jt < 2; jt++)
#pragma unroll
      for (int r = 0; r < 4; r++) { y[jt][0][r] = yi[((jt * 2 + 0) * 4 + r) * 64 + lane]; y[jt][1][r] = yi[((jt * 2 + 1) * 4 + r) * 64 + lane]; }
  }
  // the next block column first (reads the old A21 columns of nobody: only columns j1..), then L21 in place of A21
#pragma unroll
  for (int jt = 0; jt < 2; jt++) {
    const int c0 = j1 + 16 * jt;
    if (c0 > r0 + 15) continue;                          // strictly above the diagonal
    d4 c;
#pragma unroll
    for (int r = 0; r < 4; r++) c[r] = Lb[(long)(r0 + fk + 4 * r) * N + c0 + fx];
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {
      c = __builtin_amdgcn_mfma_f64_16x16x4f64(-x0[kk], y[jt][0][kk], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f64_16x16x4f64(-x1[kk], y[jt][1][kk], c, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; r++) Lb[(long)(r0 + fk + 4 * r) * N + c0 + fx] = c[r];
  }
#pragma unroll
  for (int r = 0; r < 4; r++) {
    // LDL^T: x is the image of W = L21 D11 (what the update above used as its A operand); L21 = W / d, true division like ldl.js:58-59
    const double d0 = LDL ? Lb[(long)(j0 + fk + 4 * r) * N + j0 + fk + 4 * r] : 1.0;
    const double d1 = LDL ? Lb[(long)(j0 + 16 + fk + 4 * r) * N + j0 + 16 + fk + 4 * r] : 1.0;
    Lb[(long)(r0 + fx) * N + j0 + fk + 4 * r] = LDL ? x0[r] / d0 : x0[r];
    Lb[(long)(r0 + fx) * N + j0 + 16 + fk + 4 * r] = LDL ? x1[r] / d1 : x1[r];
  }
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

// ---- LDL^T ----
// rows r in [j0+nb, N): y = A[r, j0:j0+nb] L11^-T (unit lower), W[r - r0, :] = y, L[r, j0:j0+nb] = y / D
__global__ __launch_bounds__(256) void ldl_trsm(double* __restrict__ Lm, int N, int j0, int nb, double* __restrict__ Wm, long sW) {
  double* Lb = Lm + (long)blockIdx.y * N * N;
  double* W = Wm + (long)blockIdx.y * sW;
  __shared__ double s_l[CB][CB + 1];
  const int t = threadIdx.x;
  for (int e = t; e < CB * CB; e += 256) {
    const int i = e / CB, j = e % CB;
    s_l[i][j] = (i < nb && j < nb && j <= i) ? Lb[(long)(j0 + i) * N + j0 + j] : (i == j ? 1.0 : 0.0);   // diagonal = D
  }
  __syncthreads();
  const int r = j0 + nb + blockIdx.x * 256 + t;
  if (r >= N) return;
  double* row = Lb + (long)r * N + j0;
  double* wrow = W + (long)(r - j0 - nb) * CB;
  double x[CB];
#pragma unroll
  for (int k = 0; k < CB; k++) x[k] = (k < nb) ? row[k] : 0.0;
#pragma unroll
  for (int j = 0; j < CB; j++) {                   // y_j = a_j - sum_{k<j} y_k l_jk   (unit diagonal)
    double s = x[j];
#pragma unroll
    for (int k = 0; k < j; k++) s -= x[k] * s_l[j][k];
    x[j] = s;
    asm volatile("" : "+v"(x[j]) : "v"(&s_l[0][0]) : "memory");     // see chol_trsm
  }
#pragma unroll
  for (int k = 0; k < CB; k++) {
    wrow[k] = x[k];                                // columns >= nb are exact zeros: the K = 32 GEMM may read them
    if (k < nb) row[k] = x[k] / s_l[k][k];
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
  static const bool la_off = getenv("ND4HIP_CHOL_NO_LOOKAHEAD") != nullptr;          // A/B switch
  const bool lookahead = !la_off && N >= 4 * CB && (long)batch * N <= 65536;
  static const bool fused_off = getenv("ND4HIP_CHOL_NO_FUSED_TRSM") != nullptr;      // A/B switch
  const bool fused_trsm = lookahead && !fused_off && (N % CB) == 0;                  // chol_trsm_narrow
  Nd4WsScope scope(h);
  double* inv = nullptr;
  double* ynext = nullptr;
  if (fused_trsm) { void* p = nullptr; ND4_TRY(nd4_ws_alloc(h, sizeof(double) * (size_t)batch * 2 * CB * CB, &p)); inv = static_cast<double*>(p); ynext = inv + (size_t)batch * CB * CB; }
  for (int j0 = 0, pj0 = -1; j0 < N; j0 += CB) {
    const int nb = N - j0 < CB ? N - j0 : CB;
    if (lookahead) {
      // block column j0 is complete (narrow update below); the columns from j0 + CB on still lack the update of block column pj0
      const int rest = N - (j0 + CB);
      const int ntr = (pj0 >= 0 && rest > 0) ? (rest + 63) / 64 : 0, ntc = (pj0 >= 0 && rest > 0) ? (rest + 31) / 32 : 1;
      hipLaunchKernelGGL(chol_diag_la<false>, dim3((unsigned)(1 + ntr * ntc), (unsigned)batch), dim3(256), 0, h->stream, L, N, j0, nb, flags,
                         pj0 < 0 ? 0 : pj0, ntc, inv, ynext);
    } else {
      hipLaunchKernelGGL(chol_diag, dim3((unsigned)batch), dim3(64), 0, h->stream, L, N, j0, nb, flags);
    }
    const int m2 = N - j0 - nb;
    if (m2 <= 0) break;
    if (fused_trsm) {                                      // rows below the block and the next block column in one launch
      hipLaunchKernelGGL(chol_trsm_narrow<false>, dim3((unsigned)((m2 + 63) / 64), (unsigned)batch), dim3(256), 0, h->stream, L, N, j0, inv, ynext);
      ND4_HIP(hipGetLastError());
      pj0 = j0;
      continue;
    }
    hipLaunchKernelGGL(chol_trsm, dim3((unsigned)((m2 + 255) / 256), (unsigned)batch), dim3(256), 0, h->stream, L, N, j0, nb);
    ND4_HIP(hipGetLastError());
    const int r0 = j0 + nb;
    if (lookahead) {
      // only the next block column now: A[r0:, r0:r0+CB] -= L[r0:, j0:j0+CB] L[r0:r0+CB, j0:j0+CB]^T; the rest rides under the next diagonal block
      const int ncn = N - r0 < CB ? N - r0 : CB;
      ND4_TRY(nd4_gemm(h, false, true, N - r0, ncn, nb, -1.0, L + (long)r0 * N + j0, N, sL, L + (long)r0 * N + j0, N, sL,
                       1.0, L + (long)r0 * N + r0, N, sL, batch));
      pj0 = j0;
    } else {
      // A22 -= L21 L21^T: one launch, 128x128 tiles strictly above the diagonal are skipped
      ND4_TRY(nd4_syrk_lower(h, N - r0, nb, -1.0, L + (long)r0 * N + j0, N, sL, 1.0, L + (long)r0 * N + r0, N, sL, batch));
    }
  }
  hipLaunchKernelGGL(chol_finish, dim3((unsigned)((N + 255) / 256), gy, (unsigned)batch), dim3(256), 0, h->stream, L, N);
  ND4_HIP(hipGetLastError());
  return 0;
}

// S [batch, N, N] -> packed LD [batch, N, N] (ldl.js:67-90)
int nd4_ldltrf(nd4hip_handle* h, int64_t batch64, int64_t N64, const double* S, double* LD) {
  ND4_CHECK_ARG(N64 < (1ll << 30) && batch64 < 65536, "nd4_ldltrf: extent out of range");
  const int N = (int)N64, batch = (int)batch64;
  if (N == 0 || batch == 0) return 0;
  const long sL = (long)N * N, sW = (long)N * CB;
  Nd4WsScope scope(h);
  void* p = nullptr;
  ND4_TRY(nd4_ws_alloc(h, sizeof(double) * (size_t)batch * (size_t)sW, &p));
  double* W = static_cast<double*>(p);
  const unsigned gy = (unsigned)(N < 1024 ? N : 1024);
  hipLaunchKernelGGL(chol_copy_lower, dim3((unsigned)((N + 255) / 256), gy, (unsigned)batch), dim3(256), 0, h->stream, S, LD, N);
  static const bool la_off = getenv("ND4HIP_CHOL_NO_LOOKAHEAD") != nullptr || getenv("ND4HIP_CHOL_NO_FUSED_TRSM") != nullptr;   // A/B switches
  if (!la_off && N >= 4 * CB && (N % CB) == 0 && (long)batch * N <= 65536) {
    // the Cholesky look-ahead step with D in it (chol_diag_la<true> / chol_trsm_narrow<true>): see nd4_potrf
    void* q = nullptr;
    ND4_TRY(nd4_ws_alloc(h, sizeof(double) * (size_t)batch * 2 * CB * CB, &q));
    double* inv = static_cast<double*>(q);
    double* ynext = inv + (size_t)batch * CB * CB;
    for (int j0 = 0, pj0 = -1; j0 < N; j0 += CB) {
      const int rest = N - (j0 + CB);
      const int ntr = (pj0 >= 0 && rest > 0) ? (rest + 63) / 64 : 0, ntc = (pj0 >= 0 && rest > 0) ? (rest + 31) / 32 : 1;
      hipLaunchKernelGGL(chol_diag_la<true>, dim3((unsigned)(1 + ntr * ntc), (unsigned)batch), dim3(256), 0, h->stream, LD, N, j0, CB,
                         (int*)nullptr, pj0 < 0 ? 0 : pj0, ntc, inv, ynext);
      const int m2 = N - j0 - CB;
      if (m2 <= 0) break;
      hipLaunchKernelGGL(chol_trsm_narrow<true>, dim3((unsigned)((m2 + 63) / 64), (unsigned)batch), dim3(256), 0, h->stream, LD, N, j0, inv, ynext);
      pj0 = j0;
    }
    hipLaunchKernelGGL(chol_finish, dim3((unsigned)((N + 255) / 256), gy, (unsigned)batch), dim3(256), 0, h->stream, LD, N);
    ND4_HIP(hipGetLastError());
    return 0;
  }
  for (int j0 = 0; j0 < N; j0 += CB) {
    const int nb = N - j0 < CB ? N - j0 : CB;
    hipLaunchKernelGGL(ldl_diag, dim3((unsigned)batch), dim3(64), 0, h->stream, LD, N, j0, nb);
    const int m2 = N - j0 - nb;
    if (m2 <= 0) break;
    hipLaunchKernelGGL(ldl_trsm, dim3((unsigned)((m2 + 255) / 256), (unsigned)batch), dim3(256), 0, h->stream, LD, N, j0, nb, W, sW);
    ND4_HIP(hipGetLastError());
    const int r0 = j0 + nb;                        // A22 -= (L21 D11) L21^T, tiles above the diagonal skipped
    ND4_TRY(nd4_gemm_nt_lower(h, m2, nb, -1.0, W, CB, sW, LD + (long)r0 * N + j0, N, sL, 1.0, LD + (long)r0 * N + r0, N, sL, batch));
  }
  hipLaunchKernelGGL(chol_finish, dim3((unsigned)((N + 255) / 256), gy, (unsigned)batch), dim3(256), 0, h->stream, LD, N);
  ND4_HIP(hipGetLastError());
  return 0;
}

namespace {
// X[i, :] /= LD[i, i]   (ldl.js:120-122)
__global__ void ldl_scale_rows(double* __restrict__ Xm, int N, int J, const double* __restrict__ LDm, long sLD) {
  double* X = Xm + (long)blockIdx.z * N * J; const double* LD = LDm + blockIdx.z * sLD;
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= J) return;
  for (int i = blockIdx.y; i < N; i += gridDim.y) X[(long)i * J + col] /= LD[(long)i * N + i];
}
}  // namespace

// X [batch, N, J] = L^-T D^-1 L^-1 Y (ldl.js:114-129); strides 0 = broadcast
int nd4_ldltrs(nd4hip_handle* h, int64_t batch, int64_t N, int64_t J, const double* LD, int64_t sLD, const double* Y, int64_t sY, double* X) {
  if (N == 0 || J == 0 || batch == 0) return 0;
  if (X != Y || sY != N * J) ND4_TRY(nd4_copy_matrix(h, N, J, Y, J, X, J, batch, sY, N * J));
  ND4_TRY(nd4_trsm_ld(h, false, true, batch, N, J, LD, N, sLD, X, N * J));         // forward, unit diagonal
  const unsigned gy = (unsigned)(N < 1024 ? N : 1024);
  hipLaunchKernelGGL(ldl_scale_rows, dim3((unsigned)((J + 255) / 256), gy, (unsigned)batch), dim3(256), 0, h->stream, X, (int)N, (int)J, LD, (long)sLD);
  ND4_HIP(hipGetLastError());
  return nd4_trsm_t_ex(h, true, batch, N, J, LD, N, sLD, X, N * J);                // backward with L^T, unit diagonal
}

// X [batch, N, J] = L^-T L^-1 Y (cholesky.js:117-123); strides 0 = broadcast
int nd4_potrs(nd4hip_handle* h, int64_t batch, int64_t N, int64_t J, const double* L, int64_t sL, const double* Y, int64_t sY, double* X) {
  if (N == 0 || J == 0 || batch == 0) return 0;
  if (X != Y || sY != N * J) ND4_TRY(nd4_copy_matrix(h, N, J, Y, J, X, J, batch, sY, N * J));
  ND4_TRY(nd4_trsm_ld(h, false, false, batch, N, J, L, N, sL, X, N * J));          // _tril_solve   (tri.js:45-71)
  return nd4_trsm_t(h, batch, N, J, L, N, sL, X, N * J);                           // _tril_t_solve (tri.js:100-125)
}
