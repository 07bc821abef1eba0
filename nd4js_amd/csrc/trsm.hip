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

// ---- many right-hand sides: the whole solve in ONE launch, 16 columns per workgroup ----
// A triangular solve is independent per right-hand-side column, so a workgroup that owns 16 columns needs no other workgroup: its
// M x 16 slice of X lives in the MFMA accumulator registers of 8 waves for the whole solve (wave w owns the 32-row blocks w, w + 8, ...),
// and per block b: the owner wave forms X_b = E_bb^-1 B_b with the explicitly inverted 32 x 32 diagonal block (12 MFMAs; tri_inv_blocks
// computes the inverses by substitution, true divisions, once per call), publishes it through LDS in B-operand layout (the
// accumulator layout of a 16-row tile IS the B-operand layout of its four k-steps), one barrier, and every wave subtracts E[rows, b] X_b
// from the blocks it owns (A operands straight from global memory; T is read once per workgroup and stays in L2). 2 M^2 16 flop per
// workgroup on ONE CU's MFMA pipe: 2048 x 2048 rhs in ~0.3 ms against 64 blocks x (block solve + GEMM launch) = 1.4 ms.
typedef double d4 __attribute__((ext_vector_type(4)));

// barrier for data exchanged through LDS only: __syncthreads() also waits for every outstanding GLOBAL load (s_waitcnt vmcnt(0)), which
// would put the prefetched operands of trsm_cols on the critical path of every step
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// inverse of every 32 x 32 diagonal block of E = op(T): thread j solves E_bb y = e_j. grid (blocks, batch), 32 threads.
template <bool UPPER>
__global__ __launch_bounds__(32) void tri_inv_blocks(const double* __restrict__ Tm, int ldT, long sT, int M, int unit, int trans,
                                                      double* __restrict__ invm, long sInv) {
  __shared__ double s_t[TB][TB + 1];
  const double* T = Tm + blockIdx.y * sT;
  double* inv = invm + blockIdx.y * sInv + (long)blockIdx.x * TB * TB;
  const int r0 = blockIdx.x * TB, j = threadIdx.x, nbt = M - r0 < TB ? M - r0 : TB;
  for (int i = 0; i < TB; i++) {
    double v = (i == j) ? 1.0 : 0.0;                               // identity padding beyond M
    if (i < nbt && j < nbt) {
      const bool tri = UPPER ? (j >= i) : (j <= i);
      v = tri ? (trans ? T[(long)(r0 + j) * ldT + r0 + i] : T[(long)(r0 + i) * ldT + r0 + j]) : 0.0;
      if (unit && i == j) v = 1.0;
    }
    s_t[i][j] = v;
  }
  __syncthreads();
  double y[TB];
  if (UPPER) {
#pragma unroll
    for (int i = TB - 1; i >= 0; i--) {
      double acc = (i == j) ? 1.0 : 0.0;
#pragma unroll
      for (int k = TB - 1; k > i; k--) acc -= s_t[i][k] * y[k];
      y[i] = acc / s_t[i][i];
    }
  } else {
#pragma unroll
    for (int i = 0; i < TB; i++) {
      double acc = (i == j) ? 1.0 : 0.0;
#pragma unroll
      for (int k = 0; k < i; k++) acc -= s_t[i][k] * y[k];
      y[i] = acc / s_t[i][i];
    }
  }
#pragma unroll
  for (int i = 0; i < TB; i++) inv[i * TB + j] = y[i];
}

// One panel of <= 1024 rows (32 blocks): 16 waves, wave w owns blocks w and w + 16 (4 accumulator tiles). All A operands a wave needs
// in a step (<= 2 blocks x 16 values) are requested BEFORE the step's barriers - they do not depend on X_b - so that after the owner
// has published X_b only LDS reads and MFMAs remain.
template <bool UPPER, bool TRANS>
__global__ __launch_bounds__(512) void trsm_cols(const double* __restrict__ Tm, int ldT, long sT, int M,
                                                   const double* __restrict__ invm, long sInv, double* __restrict__ Xm, int J, long sX) {
  __shared__ double s_x[2][2][4][64];                   // [step parity][tile of the block][k-step][lane]: X_b in B-operand layout
  __shared__ double s_inv[2][TB][TB + 1];               // the inverted diagonal block of this step / being fetched for the next
  const double* T = Tm + blockIdx.y * sT;
  const double* inv = invm + blockIdx.y * sInv;
  double* X = Xm + blockIdx.y * sX;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, fx = lane & 15, fk = lane >> 4;
  const int c = blockIdx.x * 16 + fx;
  const bool cok = c < J;
  const int nblk = (M + TB - 1) / TB;                    // <= 32
  auto todo = [&](int q, int b) -> bool { const int bb = w + 8 * q; return UPPER ? (bb < b) : (bb > b && bb < nblk); };
  // E[32 rows of block bb, 32 columns of block b] as A operands [i][k-step], 8 k-steps over the 32 columns. Which column a lane's
  // k-step stands for is free as long as the B operand agrees: k-step 2 q' + e of lane (fx, fk) is column kap = 8 q' + 2 fk + e, so that
  // the four lanes of a row read 64 contiguous bytes with one 16-byte load each. M is a multiple of 32 here and the rows of T are
  // 16-byte aligned (the host checks both), so there is no bounds test and exactly one load form per instantiation: any more code in
  // this loop and the 208 registers of accumulators + operands in flight spill.
  auto loadA = [&](double (&a)[16], int bb, int b) {
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int r = bb * TB + 16 * i + fx;
#pragma unroll
      for (int qq = 0; qq < 4; qq++) {
        const int cc = b * TB + 8 * qq + 2 * fk;         // columns cc, cc + 1
        if (TRANS) {
          a[i * 8 + 2 * qq] = T[(long)cc * ldT + r];
          a[i * 8 + 2 * qq + 1] = T[(long)(cc + 1) * ldT + r];
        } else {
          const double2 v = *reinterpret_cast<const double2*>(T + (long)r * ldT + cc);
          a[i * 8 + 2 * qq] = v.x; a[i * 8 + 2 * qq + 1] = v.y;
        }
      }
    }
  };
  d4 acc[8];                                             // block slot q (block w + 8 q), tile i: acc[2 q + i]
#pragma unroll
  for (int q = 0; q < 4; q++)
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int row = (w + 8 * q) * TB + 16 * i + fk + 4 * r;
        acc[2 * q + i][r] = (row < M && cok) ? X[(long)row * J + c] : 0.0;
      }
  {                                                      // the first step's inverted block
    const int b0 = UPPER ? nblk - 1 : 0;
    s_inv[0][t / TB][t % TB] = inv[(long)b0 * TB * TB + t];
    s_inv[0][(512 + t) / TB][(512 + t) % TB] = inv[(long)b0 * TB * TB + 512 + t];
  }
  for (int step = 0; step < nblk; step++) {
    const int b = UPPER ? nblk - 1 - step : step;
    const int par = step & 1;
    const int bn = UPPER ? b - 1 : b + 1;
    const double inx = (step + 1 < nblk) ? inv[(long)bn * TB * TB + t] : 0.0;   // next step's inverted block: in flight during this step
    const double iny = (step + 1 < nblk) ? inv[(long)bn * TB * TB + 512 + t] : 0.0;
    double a0[16], a1[16], a2[16], a3[16];
    const bool t0 = todo(0, b), t1 = todo(1, b), t2 = todo(2, b), t3 = todo(3, b);
    if (t0) loadA(a0, w, b);
    if (t1) loadA(a1, w + 8, b);
    if (t2) loadA(a2, w + 16, b);
    if (t3) loadA(a3, w + 24, b);
    lds_barrier();                                       // s_inv[par] (written during the previous step) is visible
    if (w == (b & 7)) {                                  // ---- owner: X_b = inv(E_bb) B_b
#pragma unroll
      for (int q = 0; q < 4; q++)
        if (q == (b >> 3)) {
          const d4 b0 = acc[2 * q], b1 = acc[2 * q + 1];
          d4 x0 = d4{0.0, 0.0, 0.0, 0.0}, x1 = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int kk = 0; kk < 4; kk++) {
            x0 = __builtin_amdgcn_mfma_f64_16x16x4f64(s_inv[par][fx][kk * 4 + fk], b0[kk], x0, 0, 0, 0);
            x1 = __builtin_amdgcn_mfma_f64_16x16x4f64(s_inv[par][16 + fx][16 + kk * 4 + fk], b1[kk], x1, 0, 0, 0);
          }
#pragma unroll
          for (int kk = 0; kk < 4; kk++) {
            if (UPPER) x0 = __builtin_amdgcn_mfma_f64_16x16x4f64(s_inv[par][fx][16 + kk * 4 + fk], b1[kk], x0, 0, 0, 0);
            else       x1 = __builtin_amdgcn_mfma_f64_16x16x4f64(s_inv[par][16 + fx][kk * 4 + fk], b0[kk], x1, 0, 0, 0);
          }
          acc[2 * q] = x0; acc[2 * q + 1] = x1;
#pragma unroll
          for (int kk = 0; kk < 4; kk++) { s_x[par][0][kk][lane] = x0[kk]; s_x[par][1][kk][lane] = x1[kk]; }
        }
    }
    lds_barrier();
    double bx[8];                                        // X_b[kap][column fx] for this lane's 8 k-steps (see loadA)
#pragma unroll
    for (int ks = 0; ks < 8; ks++) {
      const int kap = 8 * (ks >> 1) + 2 * fk + (ks & 1);  // row of X_b: tile kap / 16, accumulator register (kap % 16) / 4 of lane fx + 16 (kap % 4)
      bx[ks] = -s_x[par][kap >> 4][(kap & 15) >> 2][fx + 16 * (kap & 3)];     // (negated: the update subtracts)
    }
    if (step + 1 < nblk) { s_inv[par ^ 1][t / TB][t % TB] = inx; s_inv[par ^ 1][(512 + t) / TB][(512 + t) % TB] = iny; }
    // ---- everybody: the blocks still to be solved lose E[rows, b] X_b
    if (t0) {
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int ks = 0; ks < 8; ks++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[i * 8 + ks], bx[ks], acc[i], 0, 0, 0);
    }
    if (t1) {
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int ks = 0; ks < 8; ks++) acc[2 + i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[i * 8 + ks], bx[ks], acc[2 + i], 0, 0, 0);
    }
    if (t2) {
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int ks = 0; ks < 8; ks++) acc[4 + i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2[i * 8 + ks], bx[ks], acc[4 + i], 0, 0, 0);
    }
    if (t3) {
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int ks = 0; ks < 8; ks++) acc[6 + i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a3[i * 8 + ks], bx[ks], acc[6 + i], 0, 0, 0);
    }
  }
#pragma unroll
  for (int q = 0; q < 4; q++)
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int row = (w + 8 * q) * TB + 16 * i + fk + 4 * r;
        if (row < M && cok) X[(long)row * J + c] = acc[2 * q + i][r];
      }
}

// the one-launch form when it pays (many right-hand sides); panels of TC_PANEL rows, the rows beyond a panel by one GEMM
constexpr int TC_PANEL = 32 * TB;
bool trsm_cols_ok(int M, int J, int64_t batch, const double* T, int ldT, int64_t sT) {
  return M >= 256 && (M % TB) == 0 && (int64_t)J * batch >= 32 && (ldT & 1) == 0 && (sT & 1) == 0 && (reinterpret_cast<uintptr_t>(T) & 15) == 0;
}
// E = op(T) (lower: forward over the panels; upper: backward); trans: E = T^T of a stored lower T
template <bool UPPER>
int launch_trsm_cols(nd4hip_handle* h, bool unit, bool trans, int64_t batch, int M, int J, const double* T, int ldT, int64_t sT, double* X, long sX) {
  const int nblk = (M + TB - 1) / TB;
  Nd4WsScope scope(h);
  void* p = nullptr;
  const long sInv = (long)nblk * TB * TB;
  ND4_TRY(nd4_ws_alloc(h, sizeof(double) * (size_t)sInv * (sT == 0 ? 1 : batch), &p));
  double* inv = static_cast<double*>(p);
  hipLaunchKernelGGL(tri_inv_blocks<UPPER>, dim3((unsigned)nblk, (unsigned)(sT == 0 ? 1 : batch)), dim3(32), 0, h->stream,
                     T, ldT, (long)sT, M, unit ? 1 : 0, trans ? 1 : 0, inv, sInv);
  const int npan = (M + TC_PANEL - 1) / TC_PANEL;
  for (int pi = 0; pi < npan; pi++) {
    const int pn = UPPER ? npan - 1 - pi : pi;
    const int p0 = pn * TC_PANEL, mp = M - p0 < TC_PANEL ? M - p0 : TC_PANEL;
    const dim3 grid((unsigned)((J + 15) / 16), (unsigned)batch);
    if (trans) hipLaunchKernelGGL((trsm_cols<UPPER, true>), grid, dim3(512), 0, h->stream, T + (long)p0 * ldT + p0, ldT, (long)sT, mp,
                                  inv + (long)(p0 / TB) * TB * TB, sT == 0 ? 0l : sInv, X + (long)p0 * J, J, sX);
    else       hipLaunchKernelGGL((trsm_cols<UPPER, false>), grid, dim3(512), 0, h->stream, T + (long)p0 * ldT + p0, ldT, (long)sT, mp,
                                  inv + (long)(p0 / TB) * TB * TB, sT == 0 ? 0l : sInv, X + (long)p0 * J, J, sX);
    ND4_HIP(hipGetLastError());
    // the rows still to be solved lose E[rows, panel] X[panel]
    if (!UPPER && p0 + mp < M) {
      const int below = M - p0 - mp;
      ND4_TRY(nd4_gemm(h, false, false, below, J, mp, -1.0, T + (long)(p0 + mp) * ldT + p0, ldT, sT, X + (long)p0 * J, J, sX,
                       1.0, X + (long)(p0 + mp) * J, J, sX, batch));
    }
    if (UPPER && p0 > 0) {
      if (trans) ND4_TRY(nd4_gemm(h, true, false, p0, J, mp, -1.0, T + (long)p0 * ldT, ldT, sT, X + (long)p0 * J, J, sX, 1.0, X, J, sX, batch));
      else       ND4_TRY(nd4_gemm(h, false, false, p0, J, mp, -1.0, T + p0, ldT, sT, X + (long)p0 * J, J, sX, 1.0, X, J, sX, batch));
    }
  }
  return 0;
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
  static const bool cols_off = getenv("ND4HIP_TRSM_BLOCKED") != nullptr;            // A/B switch: block solve + GEMM per 32 rows
  if (!cols_off && trsm_cols_ok(M, J, batch, T, ldT, sT))
    return upper ? launch_trsm_cols<true>(h, unit, false, batch, M, J, T, ldT, sT, X, sX) : launch_trsm_cols<false>(h, unit, false, batch, M, J, T, ldT, sT, X, sX);
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
  static const bool cols_off = getenv("ND4HIP_TRSM_BLOCKED") != nullptr;
  if (!cols_off && trsm_cols_ok(M, J, batch, T, ldT, sT)) return launch_trsm_cols<true>(h, unit, true, batch, M, J, T, ldT, sT, X, sX);
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
