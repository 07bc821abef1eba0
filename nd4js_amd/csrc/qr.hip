// Blocked Householder QR with explicit Q on row-major fp64 matrices, batched, with the reference's
// Givens sign convention restored afterwards.
//
// Replaces src/la/qr.js:27-77 (qr_decomp_full, Givens, cache-blocked loop order) / :80-145
// (qr_decomp). The GPU algorithm is different (compact-WY Householder); the contract that is kept
// is the OUTPUT: for M <= N the reference's plane rotations give R_jj = +norm
// for every column that had something eliminated below it and det(Q) = +1, so
//   qr_signfix flips column j of Q / row j of R where tau_j != 0 and R_jj < 0, and then, if the
//   parity of (#reflectors + #flips) is odd, flips the last column of Q / last row of R.
//
// Per block column of width NB = 16 (work matrix W, reflectors kept explicitly in Vall):
//   qr_panel   geqr2 + larft for rows [j0,M) x cols [j0,j0+nb) by ONE workgroup: 16 lanes own a
//              row (one lane per panel column, coalesced 128-B rows), the panel stays in registers
//              (<= 48 rows per lane group). Column norm = wave shuffle reduction + LDS across waves;
//              the 16 dot products v^T [panel] of a step are reduced together (shuffle over the 4 row
//              groups of a wave, LDS across the 16 waves) and give both the update of the columns to
//              the right and the new column of T (V^T v).
//   qr_vtc     Wp = V^T C for the trailing columns: fp64 MFMA 16x16x4 fed straight from global
//              memory (both operands are "4 rows x 16 contiguous doubles" fragments), 256-row chunks
//              combined through LDS, chunk partials summed in
//   qr_tw      W2 = op(T) * sum_chunks(Wp)  (one thread per column)
//   nd4_gemm   C -= V * W2 (gemm.hip, K = 16)
// Q is formed by applying the block reflectors backwards to the identity with the same kernels.
#include "nd4hip_internal.h"
#include <cstdlib>
#include "dpp.h"
#include "xchg.h"
#include <type_traits>
#include <cfloat>

typedef double d4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int NB = 16;
constexpr int RMAX = 48;   // (slots per lane group of the global-memory fallback panel)
[[maybe_unused]] constexpr int RMAX_USED = RMAX;
constexpr int VTC_ROWS = 256;     // rows per workgroup of qr_vtc
constexpr int VTC_COLS = 64;      // columns per workgroup of qr_vtc

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// ------------------------------------------------------------------------------------ panel
// Slot i of lane group g is row j0 + g + 64*i; lane c of the group is column j0 + c.
template <int R, bool REG>
struct Slots {
  double a[REG ? R : 1];
  double* base;        // &W[j0+g][j0+c]
  long step;           // 64 * ld
  int nslots;          // rows in range for this group (GLOBAL mode bound)
  bool col_ok;
  int g, M, j0;
  __device__ __forceinline__ int count() const { return REG ? R : nslots; }
  __device__ __forceinline__ bool valid(int i) const { return col_ok && (j0 + g + 64 * i) < M; }
  __device__ __forceinline__ double get(int i) const {
    if constexpr (REG) return a[i];
    else return valid(i) ? base[i * step] : 0.0;
  }
  __device__ __forceinline__ void set(int i, double v) {
    if constexpr (REG) a[i] = v;
    else if (valid(i)) base[i * step] = v;
  }
};

template <int R, bool REG>
__device__ __forceinline__ void qr_panel_body(double* __restrict__ Wm, int M, long ld, long strideW,
                                              double* __restrict__ Vall, long ldv, long strideV,
                                              double* __restrict__ Tall, long strideT,
                                              double* __restrict__ taus, long strideTau, int j0, int nb) {
  __shared__ double s_red[16];
  __shared__ double s_w[16][NB];
  __shared__ double s_T[NB][NB + 1];
  __shared__ double s_Z[NB][NB];
  __shared__ double s_tau[NB];
  __shared__ double s_alpha;
  double* A = Wm + blockIdx.x * strideW;
  double* V = Vall + blockIdx.x * strideV;
  const int t = threadIdx.x;
  const int c = t & (NB - 1), g = t >> 4, wave = t >> 6;
  const int klane_base = t & 48;

  Slots<R, REG> S;
  S.base = A + (long)(j0 + g) * ld + j0 + c; S.step = 64 * ld; S.col_ok = c < nb; S.g = g; S.M = M; S.j0 = j0;
  S.nslots = (M - j0 - g + 63) / 64; if (S.nslots < 0) S.nslots = 0;
  if constexpr (REG) {
#pragma unroll
    for (int i = 0; i < R; i++) S.a[i] = S.valid(i) ? S.base[i * S.step] : 0.0;
  }
  if (t < NB * (NB + 1)) (&s_T[0][0])[t] = 0.0;
  __syncthreads();

  for (int k = 0; k < nb; k++) {
    // ---- 1. alpha = W[jc][jc], sigma = sum_{r > jc} W[r][jc]^2 (rows beyond M hold zeros) ----
    double part = 0.0;
    if (c == k) {
      const double x0 = S.get(0);
      if (g > k) part = x0 * x0;
      if (g == k) s_alpha = x0;
#pragma unroll
      for (int i = 1; i < S.count(); i++) { const double x = S.get(i); part += x * x; }
    }
    part = wave_sum(part);
    if ((t & 63) == 0) s_red[wave] = part;
    __syncthreads();
    double sigma = 0.0;
#pragma unroll
    for (int w = 0; w < 16; w++) sigma += s_red[w];
    const double alpha = s_alpha;
    double beta = alpha, tau = 0.0, scale = 0.0;
    if (sigma != 0.0) {
      beta = -copysign(sqrt(alpha * alpha + sigma), alpha);
      tau = (beta - alpha) / beta;
      scale = 1.0 / (alpha - beta);
    }
    // ---- 2. d_c = sum_r v_r * W[r][c]  (c > k: columns still to update; c < k: V_c^T v_k for T) ----
    double d = 0.0;
    {
      const double x = S.get(0);
      const double xk = __shfl(x, klane_base | k, 64);
      const double vr = (g > k) ? xk * scale : ((g == k) ? 1.0 : 0.0);
      d = vr * x;
    }
#pragma unroll
    for (int i = 1; i < S.count(); i++) {
      const double x = S.get(i);
      const double vr = __shfl(x, klane_base | k, 64) * scale;
      d += vr * x;
    }
    d += __shfl_xor(d, 16);
    d += __shfl_xor(d, 32);
    if ((t & 63) < NB) s_w[wave][c] = d;
    __syncthreads();
    double wc = 0.0;
#pragma unroll
    for (int w = 0; w < 16; w++) wc += s_w[w][c];
    // ---- 3. apply H_k to the columns right of k, store v_k in column k ----
    {
      const double x = S.get(0);
      const double xk = __shfl(x, klane_base | k, 64);
      const double vr = (g > k) ? xk * scale : ((g == k) ? 1.0 : 0.0);
      double y = x;
      if (c > k) y = x - tau * vr * wc;
      else if (c == k) y = (g > k) ? vr : ((g == k) ? beta : x);
      S.set(0, y);
    }
#pragma unroll
    for (int i = 1; i < S.count(); i++) {
      const double x = S.get(i);
      const double vr = __shfl(x, klane_base | k, 64) * scale;
      double y = x;
      if (c > k) y = x - tau * vr * wc;
      else if (c == k) y = vr;
      S.set(i, y);
    }
    // ---- 4. keep z_k = V[:,0:k]^T v_k and tau_k for T (built after the loop) ----
    if (t < k) s_Z[k][t] = wc;          // lane t < 16 has c == t
    if (t == 0) { s_tau[k] = tau; taus[blockIdx.x * strideTau + j0 + k] = tau; }
    // no barrier needed here: the next iteration's LDS writes all sit behind its own barriers
  }
  __syncthreads();
  // ---- T (larft, forward columnwise): T[i][i] = tau_i, T[i][k] = -tau_k * sum_{j=i}^{k-1} T[i][j] * z_k[j].
  // Row i of T depends only on row i -> thread i builds its row alone.
  if (t < nb) {
    double row[NB];
#pragma unroll
    for (int k = 0; k < NB; k++) row[k] = 0.0;
#pragma unroll
    for (int k = 0; k < NB; k++) {
      if (k == t) row[k] = s_tau[k];
      else if (k > t && k < nb) {
        double sum = 0.0;
#pragma unroll
        for (int j = 0; j < NB; j++) if (j >= t && j < k) sum += row[j] * s_Z[k][j];
        row[k] = -s_tau[k] * sum;
      }
    }
#pragma unroll
    for (int k = 0; k < NB; k++) s_T[t][k] = row[k];
  }
  __syncthreads();

  // ---- write back: R part (upper triangle incl. diagonal) stays in W, V goes to Vall explicitly ----
#pragma unroll
  for (int i = 0; i < S.count(); i++) {
    const int lr = g + 64 * i, r = j0 + lr;
    if (r < M && c < nb) {
      const double x = S.get(i);
      double* w = A + (long)r * ld + j0 + c;
      double* v = V + (long)r * ldv + j0 + c;
      if (lr <= c) { *w = x; *v = (lr == c) ? 1.0 : 0.0; }
      else { *w = 0.0; *v = x; }
    }
  }
  if (t < NB * NB) {
    const int i = t / NB, j = t % NB;
    Tall[blockIdx.x * strideT + (long)(j0 / NB) * NB * NB + t] = (i <= j && j < nb) ? s_T[i][j] : 0.0;
  }
}

template <int R, bool REG>
__global__ __launch_bounds__(1024) void qr_panel(double* __restrict__ Wm, int M, long ld, long strideW,
                                                   double* __restrict__ Vall, long ldv, long strideV,
                                                   double* __restrict__ Tall, long strideT,
                                                   double* __restrict__ taus, long strideTau, int j0, int nb) {
  qr_panel_body<R, REG>(Wm, M, ld, strideW, Vall, ldv, strideV, Tall, strideT, taus, strideTau, j0, nb);
}
// the same panel, only for the matrices whose flag is set: the fall-back of a multi-workgroup panel taller than the register-resident
// kernel holds (m > 2048), as a launch of its own behind phase C (two short launches that do nothing in the common case)
__global__ __launch_bounds__(1024) void qr_panel_flagged(double* __restrict__ Wm, int M, long ld, long strideW,
                                                          double* __restrict__ Vall, long ldv, long strideV,
                                                          double* __restrict__ Tall, long strideT,
                                                          double* __restrict__ taus, long strideTau, int j0, int nb, const int* __restrict__ flag) {
  if (!flag[blockIdx.x]) return;
  qr_panel_body<1, false>(Wm, M, ld, strideW, Vall, ldv, strideV, Tall, strideT, taus, strideTau, j0, nb);
}

// ---- thread-per-row panel kernel (m <= 2048): each of 512 threads keeps R whole panel rows in registers ----
// Row r = j0 + t + 512*i. Per column k (fully unrolled -> static register indices, 2 barriers):
//   sigma = sum_{r>jc} a[r][k]^2        wave shuffle reduction + 8 LDS partials
//   d[c]  = sum_r v_r a[r][c], c=0..15  per-thread partials, then a HALVING butterfly: 8+4+2+1 shuffles leave
//                                       one column total per lane (+2 to finish the wave), 8 LDS partials per
//                                       column, and the 16 totals come back as wave-uniform readlane values
//   update a[r][c>k] -= tau v_r d[c];  a[r][k] = v_r;  z_k = d[c<k] feeds T.
template <int R, int NWV = 8>
__device__ __forceinline__ void qr_panel_row_body(const int mat, double* __restrict__ Wm, int M, long ld, long strideW,
                                                  double* __restrict__ Vall, long ldv, long strideV,
                                                  double* __restrict__ Tall, long strideT,
                                                  double* __restrict__ taus, long strideTau, int j0, int nb) {
  constexpr int TT = 64 * NWV;               // threads of the workgroup: row of (thread t, slot i) = j0 + t + TT * i
  __shared__ double s_w[2][NWV][NB];        // per-wave column sums, double-buffered by column parity (one barrier per column)
  __shared__ double s_top[2][NB];         // row jc of the tile as it stands before column k's reflector
  __shared__ double s_T[NB][NB + 1];
  __shared__ double s_Z[NB][NB];
  __shared__ double s_tau[NB];
  double* A = Wm + mat * strideW;
  double* V = Vall + mat * strideV;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4, b3 = lane & 8;
  const int mycol = (b0 ? 8 : 0) + (b1 ? 4 : 0) + (b2 ? 2 : 0) + (b3 ? 1 : 0);   // column this lane ends up with

  double a[R][NB];
#pragma unroll
  for (int i = 0; i < R; i++) {
    const int r = j0 + t + TT * i;
#pragma unroll
    for (int c = 0; c < NB; c++) a[i][c] = 0.0;
    if (r < M) {
      const double* src = A + (long)r * ld + j0;
      if (nb == NB && (ld & 1) == 0) {                   // 16-byte loads of the lane's own 128-B row segment
#pragma unroll
        for (int c = 0; c < NB; c += 2) { const double2 v = *reinterpret_cast<const double2*>(src + c); a[i][c] = v.x; a[i][c + 1] = v.y; }
      } else {
#pragma unroll
        for (int c = 0; c < NB; c++) if (c < nb) a[i][c] = src[c];
      }
    }
  }
  for (int e = t; e < NB * (NB + 1); e += TT) (&s_T[0][0])[e] = 0.0;

  // one column step per compile-time k (generic lambda, see lu.hip: convergent DPP ops block `#pragma unroll`)
  auto column_step = [&](auto kc) __attribute__((always_inline)) {
    constexpr int k = decltype(kc)::value;
    if (k < nb) {
      const int jc = j0 + k;
      // ONE reduction per column: p[c] = sum_{r > jc} a[r][k] a[r][c] for all 16 columns. p[k] is the sigma of the reflector, and
      // with v = (1, scale * a[r > jc][k]) the products v^T a_c are scale * p[c] + a[jc][c] (columns c < k hold earlier
      // reflectors: the same expression gives the z_k that T needs). The norm used to be a reduction and a barrier of its own.
      double d[NB];
#pragma unroll
      for (int c = 0; c < NB; c++) d[c] = 0.0;
#pragma unroll
      for (int i = 0; i < R; i++) {
        const int r = j0 + t + TT * i;
        const double ak = (i > 0 || r > jc) ? a[i][k] : 0.0;           // row slots >= 1 are always below row jc
#pragma unroll
        for (int c = 0; c < NB; c++) d[c] += ak * a[i][c];
      }
      if (t == k) {
#pragma unroll
        for (int c = 0; c < NB; c++) s_top[k & 1][c] = a[0][c];
      }
      // halving butterfly over the 16 lanes of a group, then across the 4 groups of the wave
      double e8[8], e4[4], e2[2], e1;
#pragma unroll
      for (int j = 0; j < 8; j++) { const double snd = b0 ? d[j] : d[j + 8], kp = b0 ? d[j + 8] : d[j]; e8[j] = kp + nd4dpp::xor1(snd); }
#pragma unroll
      for (int j = 0; j < 4; j++) { const double snd = b1 ? e8[j] : e8[j + 4], kp = b1 ? e8[j + 4] : e8[j]; e4[j] = kp + nd4dpp::xor2(snd); }
#pragma unroll
      for (int j = 0; j < 2; j++) { const double snd = b2 ? e4[j] : e4[j + 2], kp = b2 ? e4[j + 2] : e4[j]; e2[j] = kp + nd4dpp::xor4(snd); }
      { const double snd = b3 ? e2[0] : e2[1], kp = b3 ? e2[1] : e2[0]; e1 = kp + nd4dpp::xor8(snd); }
      e1 += __shfl_xor(e1, 16);
      e1 += __shfl_xor(e1, 32);
      if (lane < NB) s_w[k & 1][wave][mycol] = e1;
      __syncthreads();
      double tot = 0.0;                                   // lane -> column lane & 15
#pragma unroll
      for (int w = 0; w < NWV; w++) tot += s_w[k & 1][w][lane & 15];
      double wv[NB];                                      // wave-uniform totals
#define ND4_RL(C) wv[C] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(tot), C), __builtin_amdgcn_readlane(__double2loint(tot), C));
      ND4_RL(0) ND4_RL(1) ND4_RL(2) ND4_RL(3) ND4_RL(4) ND4_RL(5) ND4_RL(6) ND4_RL(7)
      ND4_RL(8) ND4_RL(9) ND4_RL(10) ND4_RL(11) ND4_RL(12) ND4_RL(13) ND4_RL(14) ND4_RL(15)
#undef ND4_RL
      const double sigma = wv[k], alpha = s_top[k & 1][k];
      double beta = alpha, tau = 0.0, scale = 0.0;
      if (sigma != 0.0) {
        // sqrt and the two divisions sit on the column's critical path (~100 dependent instructions): one rsqrt and one
        // rcp with two Newton steps each instead. The work copy is normalised to max|a| in [1,2), so nothing over/underflows;
        // a few ulp in (beta, tau, scale) perturb H by a few ulp, like the rounding of the update itself.
        const double nn = alpha * alpha + sigma, ri = nd4dpp::fast_rsqrt(nn);
        beta = -copysign(nn * ri, alpha);
        tau = (beta - alpha) * -copysign(ri, alpha);
        scale = nd4dpp::fast_rcp(alpha - beta);
      }
#pragma unroll
      for (int c = 0; c < NB; c++) wv[c] = fma(scale, wv[c], s_top[k & 1][c]);       // v^T a_c
      double vr[R];
#pragma unroll
      for (int i = 0; i < R; i++) {
        const int r = j0 + t + TT * i;
        vr[i] = (i > 0 || r > jc) ? a[i][k] * scale : ((r == jc) ? 1.0 : 0.0);
        const double tv = tau * vr[i];
#pragma unroll
        for (int c = k + 1; c < NB; c++) a[i][c] -= tv * wv[c];
        a[i][k] = (i > 0 || r > jc) ? vr[i] : ((r == jc) ? beta : a[i][k]);
      }
      if (t == 0) {
#pragma unroll
        for (int c = 0; c < NB; c++) if (c < k) s_Z[k][c] = wv[c];
        s_tau[k] = tau;
        taus[mat * strideTau + j0 + k] = tau;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };
#define ND4_STEP(K) column_step(std::integral_constant<int, K>{});
  ND4_STEP(0) ND4_STEP(1) ND4_STEP(2) ND4_STEP(3) ND4_STEP(4) ND4_STEP(5) ND4_STEP(6) ND4_STEP(7)
  ND4_STEP(8) ND4_STEP(9) ND4_STEP(10) ND4_STEP(11) ND4_STEP(12) ND4_STEP(13) ND4_STEP(14) ND4_STEP(15)
#undef ND4_STEP
  __syncthreads();
  if (t < nb) {                                          // larft: row t of T depends only on row t
    double row[NB];
#pragma unroll
    for (int k = 0; k < NB; k++) row[k] = 0.0;
#pragma unroll
    for (int k = 0; k < NB; k++) {
      if (k == t) row[k] = s_tau[k];
      else if (k > t && k < nb) {
        double sum = 0.0;
#pragma unroll
        for (int j = 0; j < NB; j++) if (j >= t && j < k) sum += row[j] * s_Z[k][j];
        row[k] = -s_tau[k] * sum;
      }
    }
#pragma unroll
    for (int k = 0; k < NB; k++) s_T[t][k] = row[k];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < R; i++) {
    const int lr = t + TT * i, r = j0 + lr;
    if (r < M) {
      double* w = A + (long)r * ld + j0;
      double* v = V + (long)r * ldv + j0;
      double wv[NB], vv[NB];
#pragma unroll
      for (int c = 0; c < NB; c++) {
        wv[c] = (lr <= c) ? a[i][c] : 0.0;                       // R part (upper triangle incl. diagonal)
        vv[c] = (lr < c) ? 0.0 : ((lr == c) ? 1.0 : a[i][c]);    // explicit reflector: zeros above, unit diagonal
      }
      if (nb == NB && (ld & 1) == 0) {
#pragma unroll
        for (int c = 0; c < NB; c += 2) {
          *reinterpret_cast<double2*>(w + c) = double2{wv[c], wv[c + 1]};
          *reinterpret_cast<double2*>(v + c) = double2{vv[c], vv[c + 1]};
        }
      } else {
#pragma unroll
        for (int c = 0; c < NB; c++) if (c < nb) { w[c] = wv[c]; v[c] = vv[c]; }
      }
    }
  }
  for (int e = t; e < NB * NB; e += TT) {
    const int i = e / NB, j = e % NB;
    Tall[mat * strideT + (long)(j0 / NB) * NB * NB + e] = (i <= j && j < nb) ? s_T[i][j] : 0.0;
  }
}

template <int R, int NWV = 8>
__global__ __launch_bounds__(64 * NWV) void qr_panel_row(double* __restrict__ Wm, int M, long ld, long strideW,
                                                     double* __restrict__ Vall, long ldv, long strideV,
                                                     double* __restrict__ Tall, long strideT,
                                                     double* __restrict__ taus, long strideTau, int j0, int nb) {
  qr_panel_row_body<R, NWV>(blockIdx.x, Wm, M, ld, strideW, Vall, ldv, strideV, Tall, strideT, taus, strideTau, j0, nb);
}

// ---- look-ahead: the block reflector applied to ONE block of <= 16 columns by ONE workgroup ----
// C (m rows x nc columns) <- (I - V op(T) V^T) C is local to a column: X = V^T C (16 x nc; the waves split the rows, fixed-order
// LDS reduction), W = op(T) X, C -= V W, all on fp64 MFMA 16x16x4 fed straight from global memory. Because no other workgroup is
// involved, the update of the columns BEHIND the next panel can run in the same launch as the next panel's factorisation
// (qr_panel_row_la: the panel kernel keeps one workgroup busy for 30-50 us and the rest of the chip used to idle), and only the 16
// columns of the next panel itself are updated on the critical path (qr_update_narrow, one workgroup of 1024 threads).
// V: rows relative to the panel's first row, explicit zeros above the unit diagonal. s_X: NW * 256 + 256 doubles, s_Tm: 16 x 17.
template <int NT>
__device__ __forceinline__ void qr_colblock_update(double* __restrict__ s_X, double (*s_Tm)[NB + 1], const double* __restrict__ Tg, int trans,
                                                   const double* __restrict__ V, long ldv, double* __restrict__ C, long ldc, int m, int nc) {
  constexpr int NW = NT / 64;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, fx = lane & 15, fk = lane >> 4;
  if (t < NB * NB) s_Tm[t / NB][t % NB] = Tg[t];
  const int rpw = ((m + NW * 16 - 1) / (NW * 16)) * 16;            // rows per wave, a multiple of 16
  const int r0 = wave * rpw, r1 = (r0 + rpw < m) ? r0 + rpw : m;
  const bool cok = fx < nc;
  d4 acc0 = d4{0.0, 0.0, 0.0, 0.0}, acc1 = d4{0.0, 0.0, 0.0, 0.0};
  for (int rb = r0; rb < r1; rb += 64) {                           // 16 k-steps per batch: 32 loads in flight per lane
    double a[16], b[16];
#pragma unroll
    for (int kk = 0; kk < 16; kk++) {
      const int r = rb + kk * 4 + fk;
      const bool ok = r < r1;
      a[kk] = ok ? V[(long)r * ldv + fx] : 0.0;
      b[kk] = (ok && cok) ? C[(long)r * ldc + fx] : 0.0;
    }
#pragma unroll
    for (int kk = 0; kk < 16; kk += 2) {
      acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kk], b[kk], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kk + 1], b[kk + 1], acc1, 0, 0, 0);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; r++) s_X[wave * 256 + (fk + 4 * r) * 16 + fx] = acc0[r] + acc1[r];     // X[i = fk + 4r][j = fx]
  __syncthreads();
  double* s_Xs = s_X + NW * 256;                                  // the summed X, then W in its place
  if (t < 256) {
    double x = 0.0;
#pragma unroll
    for (int w = 0; w < NW; w++) x += s_X[w * 256 + t];           // fixed order
    s_Xs[t] = x;
  }
  __syncthreads();
  double wv = 0.0;
  if (t < 256) {
    const int i = t / 16, j = t % 16;
    if (trans) {
#pragma unroll
      for (int l = 0; l < NB; l++) wv += s_Tm[l][i] * s_Xs[l * 16 + j];
    } else {
#pragma unroll
      for (int l = 0; l < NB; l++) wv += s_Tm[i][l] * s_Xs[l * 16 + j];
    }
  }
  __syncthreads();
  if (t < 256) s_Xs[t] = -wv;                                     // -W[i][j]
  __syncthreads();
  double bw[4];
#pragma unroll
  for (int kk = 0; kk < 4; kk++) bw[kk] = s_Xs[(kk * 4 + fk) * 16 + fx];
  for (int rt = r0; rt < r1; rt += 64) {                          // four 16-row tiles per batch
    double av[4][4]; d4 c[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int ra = rt + q * 16 + fx;
#pragma unroll
      for (int kk = 0; kk < 4; kk++) av[q][kk] = (ra < r1) ? V[(long)ra * ldv + kk * 4 + fk] : 0.0;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int rc = rt + q * 16 + fk + 4 * r;
        c[q][r] = (rc < r1 && cok) ? C[(long)rc * ldc + fx] : 0.0;
      }
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
#pragma unroll
      for (int kk = 0; kk < 4; kk++) c[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q][kk], bw[kk], c[q], 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int rc = rt + q * 16 + fk + 4 * r;
        if (rc < r1 && cok) C[(long)rc * ldc + fx] = c[q][r];
      }
    }
  }
}

// The 16 columns of the NEXT panel are on the critical path, and one workgroup is bound there by the MFMA rate of a single CU
// (2 x 1 MFLOP at 2048 rows = 2 x 3.9 us; the one-workgroup form took 20 us per panel). Two small launches instead, the rows split over
// workgroups of 256: (x) partial X = V^T C per workgroup, (apply) every workgroup adds the partials in the same fixed order, forms
// W = op(T) X itself and updates its own rows. No device-scope fence: the launch boundary is the synchronisation.
__global__ __launch_bounds__(256) void qr_narrow_x(const double* __restrict__ Wm, int M, int N, long ld, long strideW,
                                                    const double* __restrict__ Vall, long ldv, long strideV, int pj0, int c0,
                                                    double* __restrict__ Xp, long strideXp) {
  __shared__ double s_X[4 * 256];
  const int mat = blockIdx.y, t = threadIdx.x, lane = t & 63, wave = t >> 6, fx = lane & 15, fk = lane >> 4;
  const int m = M - pj0, nc = N - c0 < NB ? N - c0 : NB;
  const double* V = Vall + mat * strideV + (long)pj0 * ldv + pj0;
  const double* C = Wm + mat * strideW + (long)pj0 * ld + c0;
  const int rb = blockIdx.x * 256 + wave * 64;
  const bool cok = fx < nc;
  double a[16], b[16];
#pragma unroll
  for (int kk = 0; kk < 16; kk++) {
    const int r = rb + kk * 4 + fk;
    const bool ok = r < m;
    a[kk] = ok ? V[(long)r * ldv + fx] : 0.0;
    b[kk] = (ok && cok) ? C[(long)r * ld + fx] : 0.0;
  }
  d4 acc0 = d4{0.0, 0.0, 0.0, 0.0}, acc1 = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int kk = 0; kk < 16; kk += 2) {
    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kk], b[kk], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kk + 1], b[kk + 1], acc1, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 4; r++) s_X[wave * 256 + (fk + 4 * r) * 16 + fx] = acc0[r] + acc1[r];
  __syncthreads();
  Xp[mat * strideXp + (long)blockIdx.x * 256 + t] = ((s_X[t] + s_X[256 + t]) + s_X[512 + t]) + s_X[768 + t];
}
__global__ __launch_bounds__(256) void qr_narrow_apply(double* __restrict__ Wm, int M, int N, long ld, long strideW,
                                                        const double* __restrict__ Vall, long ldv, long strideV,
                                                        const double* __restrict__ Tall, long strideT, int pj0, int c0,
                                                        const double* __restrict__ Xp, long strideXp, int nparts) {
  __shared__ double s_Xs[256];
  __shared__ double s_Tm[NB][NB + 1];
  const int mat = blockIdx.y, t = threadIdx.x, lane = t & 63, wave = t >> 6, fx = lane & 15, fk = lane >> 4;
  const int m = M - pj0, nc = N - c0 < NB ? N - c0 : NB;
  const double* V = Vall + mat * strideV + (long)pj0 * ldv + pj0;
  double* C = Wm + mat * strideW + (long)pj0 * ld + c0;
  const int rt = blockIdx.x * 256 + wave * 64;
  const bool cok = fx < nc;
  // this workgroup's rows first: their latency overlaps the small W computation
  double av[4][4]; d4 c[4];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int ra = rt + q * 16 + fx;
#pragma unroll
    for (int kk = 0; kk < 4; kk++) av[q][kk] = (ra < m) ? V[(long)ra * ldv + kk * 4 + fk] : 0.0;
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int rc = rt + q * 16 + fk + 4 * r;
      c[q][r] = (rc < m && cok) ? C[(long)rc * ld + fx] : 0.0;
    }
  }
  s_Tm[t / NB][t % NB] = Tall[mat * strideT + (long)(pj0 / NB) * NB * NB + t];
  double x = 0.0;
  if (nparts <= 0) nparts = (int)gridDim.x;                                                    // qr_narrow_x: one partial per workgroup
  for (int g = 0; g < nparts; g++) x += Xp[mat * strideXp + (long)g * 256 + t];               // fixed order, the same in every workgroup
  s_Xs[t] = x;
  __syncthreads();
  double wv = 0.0;
  {
    const int i = t / 16, j = t % 16;
#pragma unroll
    for (int l = 0; l < NB; l++) wv += s_Tm[l][i] * s_Xs[l * 16 + j];            // T^T X (T: full 16 x 16, see qrh_reconstruct)
  }
  __syncthreads();
  s_Xs[t] = -wv;
  __syncthreads();
  double bw[4];
#pragma unroll
  for (int kk = 0; kk < 4; kk++) bw[kk] = s_Xs[(kk * 4 + fk) * 16 + fx];
#pragma unroll
  for (int q = 0; q < 4; q++) {
#pragma unroll
    for (int kk = 0; kk < 4; kk++) c[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q][kk], bw[kk], c[q], 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int rc = rt + q * 16 + fk + 4 * r;
      if (rc < m && cok) C[(long)rc * ld + fx] = c[q][r];
    }
  }
}

// panel `pnl` (first row/column j0, workgroup 0 of a matrix) together with the previous panel's block reflector applied to the columns
// from `wc0` on, one workgroup per block of 16 columns (workgroups 1..). trans = 1 throughout (H^T C during the factorisation).
template <int R>
__global__ __launch_bounds__(512) void qr_panel_row_la(double* __restrict__ Wm, int M, int N, long ld, long strideW,
                                                        double* __restrict__ Vall, long ldv, long strideV,
                                                        double* __restrict__ Tall, long strideT,
                                                        double* __restrict__ taus, long strideTau, int j0, int nb, int pj0, int wc0, int nwide,
                                                        double* __restrict__ QT, long strideQT) {
  const int mat = blockIdx.y;
  if (blockIdx.x == 0) {
    qr_panel_row_body<R>(mat, Wm, M, ld, strideW, Vall, ldv, strideV, Tall, strideT, taus, strideTau, j0, nb);
    return;
  }
  __shared__ double s_X[8 * 256 + 256];
  __shared__ double s_Tm[NB][NB + 1];
  const int bx = (int)blockIdx.x - 1;
  const double* Tg = Tall + mat * strideT + (long)(pj0 / NB) * NB * NB;
  const double* Vp = Vall + mat * strideV + (long)pj0 * ldv + pj0;
  if (bx < nwide) {
    const int c0 = wc0 + bx * NB;
    qr_colblock_update<512>(s_X, s_Tm, Tg, 1, Vp, ldv, Wm + mat * strideW + (long)pj0 * ld + c0, ld, M - pj0, N - c0 < NB ? N - c0 : NB);
  } else {
    // Q^T = H_p^T ... H_0^T I is the same column-local update as the trailing columns ([A | I] -> [R | Q^T]): the reflectors reach an
    // M x M accumulator in the shadow of the panels instead of being multiplied together after the factorisation
    const int c0 = (bx - nwide) * NB;
    qr_colblock_update<512>(s_X, s_Tm, Tg, 1, Vp, ldv, QT + mat * strideQT + (long)pj0 * M + c0, M, M - pj0, M - c0 < NB ? M - c0 : NB);
  }
}
// the block reflector of the panel at pj0 applied to the column blocks from wc0 on (grid.x blocks): wide form (512 threads per block
// of 16 columns) and narrow form (ONE block on 1024 threads: the next panel's columns, on the critical path)
template <int NT>
__global__ __launch_bounds__(NT) void qr_update_blocks(double* __restrict__ Cm, int M, int ncols, long ld, long strideC,
                                                        const double* __restrict__ Vall, long ldv, long strideV,
                                                        const double* __restrict__ Tall, long strideT, int pj0, int wc0) {
  __shared__ double s_X[(NT / 64) * 256 + 256];
  __shared__ double s_Tm[NB][NB + 1];
  const int mat = blockIdx.y;
  const int c0 = wc0 + (int)blockIdx.x * NB;
  const int nc = ncols - c0 < NB ? ncols - c0 : NB;
  qr_colblock_update<NT>(s_X, s_Tm, Tall + mat * strideT + (long)(pj0 / NB) * NB * NB, 1,
                         Vall + mat * strideV + (long)pj0 * ldv + pj0, ldv, Cm + mat * strideC + (long)pj0 * ld + c0, ld, M - pj0, nc);
}

// =====================================================================================================================
// Multi-workgroup panel (round 3): CholeskyQR2 + a Householder-type representation of its orthonormal factor.
//
// qr_panel_row<R> is ONE workgroup walking a chain of 16 dependent column steps (reduce over all rows, barrier, reflector
// scalars, update): 27-48 us per panel on one CU, 79 % of a 2048^2 factorisation. Here the panel's ROWS are split over
// workgroups of 512 rows and every step that touches all rows is an fp64-MFMA pass whose only cross-workgroup coupling is the
// sum of 16 x 16 partial matrices, taken across a kernel boundary (the cheap grid barrier of this chip, ~1.5 us):
//   A  C <- (I - V T^T V^T) C for the panel's own 16 columns (the previous reflector, X = V^T C summed from the partials phase C
//      left behind), then partial Gram matrices G = C^T C of the updated rows;
//   B  every workgroup: R1 = chol(G) and R1^-1 (one wave, Gaussian elimination on [G | I], columns across lanes, pivot column
//      broadcast by v_readlane), Q1 = C R1^-1 in place, partial Gram matrices of Q1;
//   C  every workgroup: R2 = chol(Q1^T Q1) = I + F from two fixed-point steps of F = triu(E - F^T F) (E = Q1^T Q1 - I is
//      O(cond^2 eps): no chain; the elimination chain only when max|E| > 1e-5), Q = Q1 R2^-1 (CholeskyQR2: orthonormal to
//      O(eps)), R = S R2 R1, and the orthogonal completion of Q in compact form (Yamamoto's representation with the sign
//      choice of Ballard, Demmel, Grigori, Jacquelin, Nguyen, Solomonik, "Reconstructing Householder vectors from tall-skinny
//      QR", 2014):   H = I - W K W^T,  W = Q - [S; 0],  K = -S (Q_top - S)^-T,  S = diag(+-1) = -sign of the pivots of the
//      Gauss-Jordan elimination of Q_top - S (every pivot has magnitude >= 1).  H is orthogonal (K^-1 + K^-T = W^T W), its
//      first 16 columns are Q S, so H^T [panel] = [S R; 0]: (W, K, S R) is a (V, T, R) triple for everything downstream — V's
//      top block is full instead of unit lower triangular and T is a full 16 x 16 matrix, which is why every consumer of T
//      uses all 256 entries. The same launch leaves the partials of X = V^T C for the NEXT panel's columns.
// Three short launches per panel instead of one long one. The previous reflector's work on the other column blocks (trailing
// columns of W, Q^T) rides along in two phases, split over 512-row chunks like the panel itself: partial X = V^T C in launch A,
// C -= V (T^T X) in launches B (W) and C (Q^T). Panels the Gram route must not take — (nearly) dependent columns (a Cholesky
// pivot below HR_PIVOT_THR of its diagonal entry), columns that are exactly zero below the top block (triangular / banded input,
// where the reference skips rotations: qr.js:60,63) or non-finite data — are flagged by phase B and factorised by
// qr_panel_row_body in workgroup 0 of phase C: same result as before, at the old speed.
constexpr double HR_PIVOT_THR = 1e-5;
constexpr double HR_SERIES_MAX = 1e-5;
constexpr int HR_MIN_ROWS = 64;

enum { SEG_NX = 0, SEG_NA, SEG_NX0, SEG_F };   // SEG_NX0: partial X of the next panel's block alone; SEG_F: partial X, exchange, apply in one go
struct QrhSeg { int kind, first, count; };

struct QrhP {
  double* Wm; int M, N; long ld, strideW;
  double* Vall; long ldv, strideV;
  double* Tall; long strideT;
  double* taus; long strideTau;
  double* Xp; long strideXp;        // partials of X = V^T C on the next panel's columns: [part][256]
  double* Gp; long strideGp;        // slot 0: Gram of the top 16 rows; slots 1..: partial Gram matrices of the rows below
  double* G2p; long strideG2;       // partial Gram matrices of Q1
  double* R1; int* flag;            // per matrix: R1 (16 x 16), fall-back flag
  double* QT; long strideQT;
  double* Xs; long strideXs;        // side work: partials of X per (column block, row chunk): [cb][rc][256]
  int j0;                           // first row / column of the panel
  int pj0;                          // previous panel (-1: none): its reflector is what the side blocks and phase A apply
  int nxp;                          // partials of X to sum in phase A
  int nrow;                         // row workgroups of this launch
  int ngp;                          // Gram partials to sum (phase B: phase A's row workgroups; phase C: phase B's)
  int wide0, nnw, nqb, nrc;         // side work (reflector pj0): nnw column blocks of W from wide0 on, then nqb blocks of Q^T; nrc row chunks each;
                                    // a side workgroup takes TWO adjacent blocks (pairs of W blocks first, then pairs of Q^T blocks)
  int nseg; QrhSeg seg[2];          // the side work of this launch: workgroups nrow.. walk these segments
  int skip_x;                       // phase C: no partial X for the next panel (last panel of an outer block: the block update covers it)
  int na_shift;                     // SEG_NA starts at this block of W (1: the fused launch B+C, whose row workgroups take the next panel's columns themselves)
  unsigned long long* Xch; long strideXch;   // fused launch B+C: the row workgroups' exchange slots (512 tagged words each)
  unsigned long long* Xsx; long strideXsx;   // fused side work: [column block][row chunk] slots of 512 tagged words (rcs_max chunks per block)
  int rcs_max;
  long long* stamps; int stamp_slot; // debug (ND4HIP_QR_STAMPS): 100 MHz wall-clock stamps of workgroup 0, 8 per launch
  int* status;                      // the handle's exchange status word (host-coherent): raised when a spin limit is hit
  int drop_tag;                     // tests only (ND4HIP_TEST_DROP_PUBLISH): the last row workgroup of the panel with this tag skips its first publication
};

#include "qr_chain16.h"

// (round 3's unblocked chains: kept behind -DND4HIP_QR_OLD_CHAINS for A/B builds)
template <int K, int I>
__device__ __forceinline__ void qrh_ge_row(double (&g)[16], double gk) {
  if constexpr (I > K) g[I] = fma(-nd4dpp::rl_d(g[I], K), gk, g[I]);
}
template <int K, int... I>
__device__ __forceinline__ void qrh_ge_rows(double (&g)[16], double gk, std::integer_sequence<int, I...>) { (qrh_ge_row<K, I>(g, gk), ...); }
template <int K>
__device__ __forceinline__ void qrh_ge_step(double (&g)[16], double (&d)[16]) {
  const double pk = nd4dpp::rl_d(g[K], K);
  d[K] = pk;
  qrh_ge_rows<K>(g, g[K] * nd4dpp::fast_rcp(pk), std::make_integer_sequence<int, 16>{});     // row i -= G[i][K] * (row K / pivot)
}
template <int... K>
__device__ __forceinline__ void qrh_ge_all(double (&g)[16], double (&d)[16], std::integer_sequence<int, K...>) { (qrh_ge_step<K>(g, d), ...); }

// One wave. s_G: symmetric positive definite 16 x 16. Out: s_R = chol(G)^T (upper, G = R^T R), s_Ri = R^-1 (upper).
// Lanes 0..15 hold the columns of G, lanes 16..31 the columns of I; elimination without pivoting leaves U = D L^T and L^-1,
// R = D^-1/2 U, R^-1 = (D^-1/2 L^-1)^T. Returns true when every pivot is positive and >= thr * its diagonal entry.
// Optional: s_Rt[j * 16 + i] = R[i][j] (column j contiguous) and s_rd[i] = 1 / R[i][i], what a forward substitution x R = c reads.
__device__ __forceinline__ bool qrh_chol16(const double* __restrict__ s_G, double* __restrict__ s_R, double* __restrict__ s_Ri, double thr,
                                           double* __restrict__ s_Rt = nullptr, double* __restrict__ s_rd = nullptr) {
  const int lane = threadIdx.x & 63;
  double g[16], d[16], d0[16];
#pragma unroll
  for (int i = 0; i < 16; i++) g[i] = lane < 16 ? s_G[i * 16 + lane] : ((lane - 16) == i ? 1.0 : 0.0);
#define ND4_D0(K) d0[K] = nd4dpp::rl_d(g[K], K);
  ND4_D0(0) ND4_D0(1) ND4_D0(2) ND4_D0(3) ND4_D0(4) ND4_D0(5) ND4_D0(6) ND4_D0(7)
  ND4_D0(8) ND4_D0(9) ND4_D0(10) ND4_D0(11) ND4_D0(12) ND4_D0(13) ND4_D0(14) ND4_D0(15)
#undef ND4_D0
  qrh_ge_all(g, d, std::make_integer_sequence<int, 16>{});
  bool ok = true;
  double rs[16];
#pragma unroll
  for (int k = 0; k < 16; k++) {
    ok = ok && (d[k] > 0.0) && (d[k] >= thr * d0[k]) && (d0[k] < DBL_MAX);
    rs[k] = nd4dpp::fast_rsqrt(d[k]);
  }
  if (lane < 16) {
#pragma unroll
    for (int i = 0; i < 16; i++) s_R[i * 16 + lane] = (i <= lane) ? g[i] * rs[i] : 0.0;
    if (s_Rt != nullptr) {
#pragma unroll
      for (int i = 0; i < 16; i++) s_Rt[lane * 16 + i] = (i <= lane) ? g[i] * rs[i] : 0.0;
#pragma unroll
      for (int i = 0; i < 16; i++) if (i == lane) s_rd[i] = rs[i];
    }
  } else if (lane < 32) {
    const int c = lane - 16;
#pragma unroll
    for (int r = 0; r < 16; r++) s_Ri[c * 16 + r] = (r >= c) ? g[r] * rs[r] : 0.0;
  }
  return ok;
}

template <int K, int I>
__device__ __forceinline__ void qrh_gj_row(double (&g)[16], double gk) {
  if constexpr (I != K) g[I] = fma(-nd4dpp::rl_d(g[I], K), gk, g[I]);
}
template <int K, int... I>
__device__ __forceinline__ void qrh_gj_rows(double (&g)[16], double gk, std::integer_sequence<int, I...>) { (qrh_gj_row<K, I>(g, gk), ...); }
template <int K>
__device__ __forceinline__ void qrh_gj_step(double (&g)[16], double (&sg)[16], int lane) {
  const double p = nd4dpp::rl_d(g[K], K);
  const double s = (p >= 0.0) ? -1.0 : 1.0;            // the pivot p - s has magnitude >= 1
  sg[K] = s;
  if (lane == K) g[K] -= s;
  const double gk = g[K] * nd4dpp::fast_rcp(p - s);    // row K of [B | I], scaled
  qrh_gj_rows<K>(g, gk, std::make_integer_sequence<int, 16>{});
  g[K] = gk;
}
template <int... K>
__device__ __forceinline__ void qrh_gj_all(double (&g)[16], double (&sg)[16], int lane, std::integer_sequence<int, K...>) { (qrh_gj_step<K>(g, sg, lane), ...); }

// One wave. s_Z: top 16 x 16 block of the orthonormal Q. Gauss-Jordan elimination without pivoting of [Z - S | I] with
// S_k = -sign(pivot) chosen on the way (lanes 0..15: columns of Z, lanes 16..31: columns of I -> B^-1, B = Z - S).
// Out: s_S (signs) and s_K = K = -S B^-T (K[c][r] = -S_c B^-1[r][c]: lane 16 + c writes row c).
__device__ __forceinline__ void qrh_gj16(const double* __restrict__ s_Z, double* __restrict__ s_K, double* __restrict__ s_S) {
  const int lane = threadIdx.x & 63;
  double g[16], sg[16];
#pragma unroll
  for (int i = 0; i < 16; i++) g[i] = lane < 16 ? s_Z[i * 16 + lane] : ((lane - 16) == i ? 1.0 : 0.0);
  qrh_gj_all(g, sg, lane, std::make_integer_sequence<int, 16>{});
  if (lane >= 16 && lane < 32) {
    const int c = lane - 16;
    double sc = 0.0;
#pragma unroll
    for (int k = 0; k < 16; k++) if (c == k) sc = sg[k];
#pragma unroll
    for (int r = 0; r < 16; r++) s_K[c * 16 + r] = -sc * g[r];
    s_S[c] = sc;
  }
}

#ifdef ND4HIP_QR_OLD_CHAINS
#define ND4_CHOL16 qrh_chol16
#define ND4_GJ16 qrh_gj16
#else
#define ND4_CHOL16 qrc_chol16_inv      // the eliminations on the matrix core (qr_chain16.h)
#define ND4_GJ16 qrc_gj16
#endif
constexpr int QRH_LDS = 8 * 256 + 256 + NB * (NB + 1) + 16;      // doubles: what the row phases and the side work need ...
constexpr int QRH_SMEM = QRH_LDS + 11 * 256 + 32;                // ... and the whole per-workgroup buffer: phase C carves its 16 x 16 matrices behind QRH_LDS

// sum of n partials (256 doubles apart) of one entry, fixed order; the loads are issued together (a rolled loop of dependent
// global loads costs a full memory round trip per partial: 1-1.5 us each behind a kernel boundary)
__device__ __forceinline__ double qrh_sum_parts(const double* __restrict__ src, int n) {
  double x = 0.0;
  for (int p0 = 0; p0 < n; p0 += 8) {
    double v[8];
#pragma unroll
    for (int p = 0; p < 8; p++) v[p] = (p0 + p < n) ? src[(long)(p0 + p) * 256] : 0.0;
#pragma unroll
    for (int p = 0; p < 8; p++) x += v[p];
  }
  return x;
}
// fixed-order sum of the eight waves' 16 x 16 accumulators -> dst[256] (global)
__device__ __forceinline__ void qrh_reduce_store(double* __restrict__ s_part, const d4& a0, const d4& a1, double* __restrict__ dst) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, fx = lane & 15, fk = lane >> 4;
#pragma unroll
  for (int r = 0; r < 4; r++) s_part[wave * 256 + (fk + 4 * r) * 16 + fx] = a0[r] + a1[r];
  __syncthreads();
  if (t < 256) {
    double x = 0.0;
#pragma unroll
    for (int w = 0; w < 8; w++) x += s_part[w * 256 + t];
    dst[t] = x;
  }
}

// The wave's 64 rows [rb, rb + 64) of NBLK adjacent 16-column blocks: C <- C - V (T^T X), X (per block) = the sum of nx partials
// (256 doubles apart, fixed order, the same in every workgroup; block b's partials start at Xsrc + b * xstride). Vcol / Ccol point at
// (row 0, first column) of the reflector / the first block; nc = columns in all (<= 16 NBLK). On return c[b] holds the updated tile of
// block b in accumulator layout (c[b][q][r]: row rb + 16 q + fk + 4 r, column 16 b + fx; rows outside [0, M) are zero).
// s_w: 256 (NBLK + 1) doubles of LDS. Every thread of the workgroup must call it. The reflector rows are loaded ONCE for all blocks.
template <int NBLK>
__device__ __forceinline__ void qrh_apply_rows(double* __restrict__ s_w, const double* __restrict__ Xsrc, long xstride, int nx, const double* __restrict__ Tg,
                                               const double* __restrict__ Vcol, long ldv, double* __restrict__ Ccol, long ldc, int nc,
                                               int rb, int M, d4 (&c)[NBLK][4]) {
  const int t = threadIdx.x, lane = t & 63, fx = lane & 15, fk = lane >> 4;
  double av[4][4];
#pragma unroll
  for (int q = 0; q < 4; q++) {
#pragma unroll
    for (int b = 0; b < NBLK; b++) {
      const bool cok = b * NB + fx < nc;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int rc = rb + q * 16 + fk + 4 * r;
        c[b][q][r] = (rc >= 0 && rc < M && cok) ? Ccol[(long)rc * ldc + b * NB + fx] : 0.0;
      }
    }
    const int ra = rb + q * 16 + fx;
    if (ra >= 0 && ra < M) {                                           // 32 contiguous bytes per lane: k-step kk <-> column 4 fk + kk
      const double2 v0 = *reinterpret_cast<const double2*>(Vcol + (long)ra * ldv + 4 * fk);
      const double2 v1 = *reinterpret_cast<const double2*>(Vcol + (long)ra * ldv + 4 * fk + 2);
      av[q][0] = v0.x; av[q][1] = v0.y; av[q][2] = v1.x; av[q][3] = v1.y;
    } else { av[q][0] = av[q][1] = av[q][2] = av[q][3] = 0.0; }
  }
  double* s_X = s_w;                                                   // NBLK x 256
  double* s_Tm = s_w + NBLK * 256;
  if (t < NBLK * 256) {
    const int b = t >> 8, e = t & 255;
    s_X[t] = (b * NB < nc) ? qrh_sum_parts(Xsrc + b * xstride + e, nx) : 0.0;
    if (t < 256) s_Tm[t] = Tg[t];
  }
  __syncthreads();
  double wv = 0.0;
  if (t < NBLK * 256) {
    const int b = t >> 8, i = (t & 255) / 16, j = t % 16;
#pragma unroll
    for (int l = 0; l < NB; l++) wv += s_Tm[l * 16 + i] * s_X[b * 256 + l * 16 + j];        // T^T X, T a full 16 x 16 matrix
  }
  __syncthreads();
  if (t < NBLK * 256) s_X[t] = -wv;
  __syncthreads();
#pragma unroll
  for (int b = 0; b < NBLK; b++) {
    const bool cok = b * NB + fx < nc;
    double bw[4];
#pragma unroll
    for (int kk = 0; kk < 4; kk++) bw[kk] = s_X[b * 256 + (4 * fk + kk) * 16 + fx];
#pragma unroll
    for (int q = 0; q < 4; q++) {
#pragma unroll
      for (int kk = 0; kk < 4; kk++) c[b][q] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q][kk], bw[kk], c[b][q], 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int rc = rb + q * 16 + fk + 4 * r;
        if (rc >= 0 && rc < M && cok) Ccol[(long)rc * ldc + b * NB + fx] = c[b][q][r];
      }
    }
  }
}

// ---- tagged words: the in-kernel exchange between co-resident workgroups (see qrh_bc) ----
// (qx_st / qx_sum: xchg.h)
// ---- side work of the panel launches: the previous reflector (panel pj0) on the other column blocks (trailing columns of W, then
// Q^T), in two phases split over 512-row chunks like the panel itself: partial X = V^T C (SEG_NX, launch A), C -= V (T^T X)
// (SEG_NA: the blocks of W in launch B, whose first one phase C reads; those of Q^T in launch C) ----
struct QrhSide { double* C; long ldc; int c0, nc, cb, rc; };   // cb: first block (index among W blocks, then Q^T blocks); nc: columns of the pair
__device__ __forceinline__ QrhSide qrh_near_of(const QrhP& P, int mat, int e) {
  QrhSide s;
  const int sh = P.na_shift;
  const int pr = e / P.nrc, nwp = (P.nnw - sh + 1) / 2;
  s.rc = e % P.nrc;
  if (pr < nwp) {
    s.cb = sh + 2 * pr; s.C = P.Wm + mat * P.strideW; s.ldc = P.ld; s.c0 = P.wide0 + s.cb * NB;
    const int end = P.wide0 + P.nnw * NB < P.N ? P.wide0 + P.nnw * NB : P.N;
    s.nc = end - s.c0 < 2 * NB ? end - s.c0 : 2 * NB;
  } else {
    const int qb = 2 * (pr - nwp);
    s.cb = P.nnw + qb; s.C = P.QT + mat * P.strideQT; s.ldc = P.M; s.c0 = qb * NB;
    s.nc = P.M - s.c0 < 2 * NB ? P.M - s.c0 : 2 * NB;
  }
  return s;
}
// partial X = V^T C over one 512-row chunk of a pair of column blocks (the reflector at vj0): the reflector slabs are loaded once
__device__ __forceinline__ void qrh_side_x(const QrhP& P, int mat, const QrhSide& s, int vj0, double* __restrict__ dst, double* __restrict__ s_buf) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, fx = lane & 15, fk = lane >> 4;
  const int rb = vj0 + s.rc * 512 + wave * 64;
  const double* V = P.Vall + mat * P.strideV + vj0;
  const bool ok0 = fx < s.nc, ok1 = NB + fx < s.nc;
  double c0[16], c1[16], v[16];
#pragma unroll
  for (int u = 0; u < 16; u++) {                                       // 16 slabs of 4 rows: all operands in the same (row fk, column fx) layout
    const int r = rb + 4 * u + fk;
    const bool rok = r < P.M;
    const double* cp = s.C + (long)r * s.ldc + s.c0 + fx;
    c0[u] = (rok && ok0) ? cp[0] : 0.0;
    c1[u] = (rok && ok1) ? cp[NB] : 0.0;
    v[u] = rok ? V[(long)r * P.ldv + fx] : 0.0;
  }
  d4 x0 = d4{0.0, 0.0, 0.0, 0.0}, x1 = x0, y0 = x0, y1 = x0;
#pragma unroll
  for (int u = 0; u < 16; u += 2) {
    x0 = __builtin_amdgcn_mfma_f64_16x16x4f64(v[u], c0[u], x0, 0, 0, 0);
    y0 = __builtin_amdgcn_mfma_f64_16x16x4f64(v[u], c1[u], y0, 0, 0, 0);
    x1 = __builtin_amdgcn_mfma_f64_16x16x4f64(v[u + 1], c0[u + 1], x1, 0, 0, 0);
    y1 = __builtin_amdgcn_mfma_f64_16x16x4f64(v[u + 1], c1[u + 1], y1, 0, 0, 0);
  }
  qrh_reduce_store(s_buf, x0, x1, dst);
  if (s.nc > NB) {                                                     // (uniform) the pair's second block: the next slot of Xs
    __syncthreads();
    qrh_reduce_store(s_buf, y0, y1, dst + (long)P.nrc * 256);
  }
}
// C -= V (T^T X) on one 512-row chunk of a pair of column blocks
__device__ __forceinline__ void qrh_near_apply(const QrhP& P, int mat, int e, double* __restrict__ s_buf) {
  const QrhSide s = qrh_near_of(P, mat, e);
  const int wave = threadIdx.x >> 6;
  d4 c[2][4];
  qrh_apply_rows<2>(s_buf, P.Xs + mat * P.strideXs + (long)s.cb * P.nrc * 256, (long)P.nrc * 256, P.nrc,
                    P.Tall + mat * P.strideT + (long)(P.pj0 / NB) * NB * NB,
                    P.Vall + mat * P.strideV + P.pj0, P.ldv, s.C + s.c0, s.ldc, s.nc, P.pj0 + s.rc * 512 + wave * 64, P.M, c);
}
// ---- the two side phases of one (pair of column blocks, 512-row chunk) in ONE workgroup (round 3, second half) ----
// Partial X = V^T C of the chunk, across to the nrc workgroups of the pair (consecutive workgroups of the launch: tagged words, see
// qrh_bc), C -= V (T^T X) on the tiles that are still in the registers: the side blocks are read once and written once per
// reflector instead of read twice and written once — the launches of the upper half of a 2048^2 factorisation are bound by exactly
// this traffic. (The accumulator image c[q][r] of a 16-row tile and the slab image of its rows 4 (4q + r) + fk are the same registers.)
__device__ __forceinline__ void qrh_side_fused(const QrhP& P, int mat, int e, double* __restrict__ s_buf) {
  const QrhSide s = qrh_near_of(P, mat, e);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, fx = lane & 15, fk = lane >> 4;
  const int vj0 = P.pj0, rb = vj0 + s.rc * 512 + wave * 64;
  const double* V = P.Vall + mat * P.strideV + vj0;
  const bool ok0 = fx < s.nc, ok1 = NB + fx < s.nc;
  const unsigned tag = (unsigned)(vj0 / NB + 1);
  double c0[16], c1[16];
  d4 x0 = d4{0.0, 0.0, 0.0, 0.0}, x1 = x0, y0 = x0, y1 = x0;
  {
    double v[16];
#pragma unroll
    for (int u = 0; u < 16; u++) {
      const int r = rb + 4 * u + fk;
      const bool rok = r < P.M;
      const double* cp = s.C + (long)r * s.ldc + s.c0 + fx;
      c0[u] = (rok && ok0) ? cp[0] : 0.0;
      c1[u] = (rok && ok1) ? cp[NB] : 0.0;
      v[u] = rok ? V[(long)r * P.ldv + fx] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 16; u += 2) {
      x0 = __builtin_amdgcn_mfma_f64_16x16x4f64(v[u], c0[u], x0, 0, 0, 0);
      y0 = __builtin_amdgcn_mfma_f64_16x16x4f64(v[u], c1[u], y0, 0, 0, 0);
      x1 = __builtin_amdgcn_mfma_f64_16x16x4f64(v[u + 1], c0[u + 1], x1, 0, 0, 0);
      y1 = __builtin_amdgcn_mfma_f64_16x16x4f64(v[u + 1], c1[u + 1], y1, 0, 0, 0);
    }
  }
  // the reflector rows as A operands of C -= V W (32 contiguous bytes per lane), requested before the exchange
  double av[4][4];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int ra = rb + q * 16 + fx;
    if (ra < P.M) {
      const double2 v0 = *reinterpret_cast<const double2*>(V + (long)ra * P.ldv + 4 * fk);
      const double2 v1 = *reinterpret_cast<const double2*>(V + (long)ra * P.ldv + 4 * fk + 2);
      av[q][0] = v0.x; av[q][1] = v0.y; av[q][2] = v1.x; av[q][3] = v1.y;
    } else { av[q][0] = av[q][1] = av[q][2] = av[q][3] = 0.0; }
  }
  // fixed-order sums over the eight waves, then across: slot (block, chunk), entry = position in the 16 x 16 matrix
  qx_u64* slots = P.Xsx + mat * P.strideXsx;
  double* s_Tm = s_buf + 8 * 256;
#pragma unroll
  for (int r = 0; r < 4; r++) s_buf[wave * 256 + (fk + 4 * r) * 16 + fx] = x0[r] + x1[r];
  if (t < 256) s_Tm[t] = P.Tall[mat * P.strideT + (long)(vj0 / NB) * NB * NB + t];
  __syncthreads();
  if (t < 256) {
    double xs = 0.0;
#pragma unroll
    for (int w = 0; w < 8; w++) xs += s_buf[w * 256 + t];
    qx_st(slots + ((long)s.cb * P.rcs_max + s.rc) * 512, t, xs, tag);
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; r++) s_buf[wave * 256 + (fk + 4 * r) * 16 + fx] = y0[r] + y1[r];
  __syncthreads();
  if (t >= 256 && s.nc > NB) {
    double xs = 0.0;
#pragma unroll
    for (int w = 0; w < 8; w++) xs += s_buf[w * 256 + (t - 256)];
    qx_st(slots + ((long)(s.cb + 1) * P.rcs_max + s.rc) * 512, t - 256, xs, tag);
  }
  __syncthreads();
  double* s_X = s_buf;                                                 // 2 x 256
  {
    const int b = t >> 8, en = t & 255;
    double x = 0.0;
    if (b == 0 || s.nc > NB) {
      const qx_u64* base = slots + (long)(s.cb + b) * P.rcs_max * 512;
      int spins = 0;
      for (;;) {
        bool ok = true;
        x = qx_sum(base, en, P.nrc, tag, ok);
        if (ok) break;
        if (++spins > QX_SPIN_LIMIT) { x = __builtin_nan(""); qx_raise(P.status); break; }    // a stuck exchange must not pass silently
        __builtin_amdgcn_s_sleep(2);
      }
    }
    s_X[t] = x;
  }
  __syncthreads();
  double wv = 0.0;
  {
    const int b = t >> 8, i = (t & 255) / 16, j = t % 16;
#pragma unroll
    for (int l = 0; l < NB; l++) wv += s_Tm[l * 16 + i] * s_X[b * 256 + l * 16 + j];          // T^T X, T a full 16 x 16 matrix
  }
  __syncthreads();
  s_X[t] = -wv;
  __syncthreads();
#pragma unroll
  for (int b = 0; b < 2; b++) {
    const bool cok = b == 0 ? ok0 : ok1;
    double bw[4];
#pragma unroll
    for (int kk = 0; kk < 4; kk++) bw[kk] = s_X[b * 256 + (4 * fk + kk) * 16 + fx];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      d4 c;
#pragma unroll
      for (int r = 0; r < 4; r++) c[r] = b == 0 ? c0[4 * q + r] : c1[4 * q + r];
#pragma unroll
      for (int kk = 0; kk < 4; kk++) c = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q][kk], bw[kk], c, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int rc = rb + q * 16 + fk + 4 * r;
        if (rc < P.M && cok) s.C[(long)rc * s.ldc + s.c0 + b * NB + fx] = c[r];
      }
    }
  }
}
// workgroup i of a launch's side work
__device__ __forceinline__ void qrh_side(const QrhP& P, int mat, int i, double* __restrict__ s_buf) {
  int kind = -1, e = 0;
  for (int sg = 0; sg < P.nseg; sg++) {
    if (i < P.seg[sg].count) { kind = P.seg[sg].kind; e = P.seg[sg].first + i; break; }
    i -= P.seg[sg].count;
  }
  if (kind == SEG_NX) {
    const QrhSide s = qrh_near_of(P, mat, e);
    qrh_side_x(P, mat, s, P.pj0, P.Xs + mat * P.strideXs + ((long)s.cb * P.nrc + s.rc) * 256, s_buf);
  } else if (kind == SEG_NA) {
    qrh_near_apply(P, mat, e, s_buf);
  } else if (kind == SEG_NX0) {                                        // the block right behind the panel alone: e = row chunk
    QrhSide s;
    s.rc = e; s.cb = 0; s.C = P.Wm + mat * P.strideW; s.ldc = P.ld; s.c0 = P.wide0;
    s.nc = P.N - s.c0 < NB ? P.N - s.c0 : NB;
    qrh_side_x(P, mat, s, P.pj0, P.Xs + mat * P.strideXs + (long)s.rc * 256, s_buf);
  } else if (kind == SEG_F) {
    qrh_side_fused(P, mat, e, s_buf);
  }
}
// side work on its own (the flush after the last such panel)
__global__ __launch_bounds__(512) void qrh_side_only(const QrhP P) {
  __shared__ double s_buf[QRH_SMEM];
  qrh_side(P, blockIdx.y, blockIdx.x, s_buf);
}

__device__ __forceinline__ void qrh_stamp(const QrhP& P, int k) {
  if (P.stamps != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) P.stamps[P.stamp_slot * 8 + k] = wall_clock64();
}
// X = V^T C over all m rows by the eight waves of one workgroup -> dst[256] (the fall-back panel's share of phase C)
__device__ __forceinline__ void qrh_x_full(double* __restrict__ s_part, const double* __restrict__ V, long ldv, const double* __restrict__ C, long ldc,
                                           int m, int nc, double* __restrict__ dst) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, fx = lane & 15, fk = lane >> 4;
  const int rpw = ((m + 8 * 16 - 1) / (8 * 16)) * 16;
  const int r0 = wave * rpw, r1 = (r0 + rpw < m) ? r0 + rpw : m;
  const bool cok = fx < nc;
  d4 a0 = d4{0.0, 0.0, 0.0, 0.0}, a1 = a0;
  for (int rr = r0; rr < r1; rr += 8) {
    const int ra = rr + fk, rb2 = rr + 4 + fk;
    const double va = ra < r1 ? V[(long)ra * ldv + fx] : 0.0, ca = (ra < r1 && cok) ? C[(long)ra * ldc + fx] : 0.0;
    const double vb = rb2 < r1 ? V[(long)rb2 * ldv + fx] : 0.0, cb = (rb2 < r1 && cok) ? C[(long)rb2 * ldc + fx] : 0.0;
    a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(va, ca, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(vb, cb, a1, 0, 0, 0);
  }
  qrh_reduce_store(s_part, a0, a1, dst);
}

// ---- phase A: the previous reflector on the panel's own columns (rows from j0 - 16), then the partial Gram matrices ----
__global__ __launch_bounds__(512) void qrh_gram(const QrhP P) {
  __shared__ double s_buf[QRH_SMEM];
  const int mat = blockIdx.y;
  if ((int)blockIdx.x >= P.nrow) { qrh_side(P, mat, (int)blockIdx.x - P.nrow, s_buf); return; }
  const int g = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6, fx = lane & 15, fk = lane >> 4;
  double* A = P.Wm + mat * P.strideW;
  const int j0 = P.j0, ub = j0 - NB, M = P.M;
  const int rb = ub + (g * 8 + wave) * 64;
  d4 cc[1][4];
  d4 (&c)[4] = cc[0];
  qrh_stamp(P, 0);
  if (P.pj0 >= 0) {
    const double* Xsrc = P.Xp + mat * P.strideXp;
    int nx = P.nxp;
    if (P.Xch != nullptr && P.flag[mat]) {
      // the previous panel took the fall-back inside the fused launch B+C, which leaves no partials of X behind: every workgroup
      // forms X = V^T C over all rows itself (rare path), in its own slot of Xp
      double* mine = P.Xp + mat * P.strideXp + (long)g * 256;
      qrh_x_full(s_buf, P.Vall + mat * P.strideV + (long)ub * P.ldv + ub, P.ldv, A + (long)ub * P.ld + j0, P.ld, M - ub, NB, mine);
      __syncthreads();
      Xsrc = mine; nx = 1;
    }
    qrh_apply_rows<1>(s_buf, Xsrc, 0, nx, P.Tall + mat * P.strideT + (long)(ub / NB) * NB * NB,
                      P.Vall + mat * P.strideV + ub, P.ldv, A + j0, P.ld, NB, rb, M, cc);
    __syncthreads();                                                  // s_buf is reused below
    qrh_stamp(P, 1);
  } else {
#pragma unroll
    for (int q = 0; q < 4; q++) {
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int rc = rb + q * 16 + fk + 4 * r;
        c[q][r] = (rc >= 0 && rc < M) ? A[(long)rc * P.ld + j0 + fx] : 0.0;
      }
    }
  }
  // Gram matrices: the accumulator image of a 16-row tile is four 4-row slabs in operand layout (A and B operand alike)
  d4 g0 = d4{0.0, 0.0, 0.0, 0.0}, g1 = g0, gt = g0;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int ti = (g * 8 + wave) * 4 + q;                            // tile 0: the 16 rows above the panel, tile 1: its top block
    if (ti == 1) {
#pragma unroll
      for (int r = 0; r < 4; r++) gt = __builtin_amdgcn_mfma_f64_16x16x4f64(c[q][r], c[q][r], gt, 0, 0, 0);
    } else if (ti >= 2) {
      g0 = __builtin_amdgcn_mfma_f64_16x16x4f64(c[q][0], c[q][0], g0, 0, 0, 0);
      g1 = __builtin_amdgcn_mfma_f64_16x16x4f64(c[q][1], c[q][1], g1, 0, 0, 0);
      g0 = __builtin_amdgcn_mfma_f64_16x16x4f64(c[q][2], c[q][2], g0, 0, 0, 0);
      g1 = __builtin_amdgcn_mfma_f64_16x16x4f64(c[q][3], c[q][3], g1, 0, 0, 0);
    }
  }
  if (g == 0 && wave == 0) {
#pragma unroll
    for (int r = 0; r < 4; r++) P.Gp[mat * P.strideGp + (fk + 4 * r) * 16 + fx] = gt[r];
  }
  qrh_reduce_store(s_buf, g0, g1, P.Gp + mat * P.strideGp + (long)(1 + g) * 256);
  qrh_stamp(P, 4);
}

// the wave's 64 rows of the panel's 16 columns as MFMA A operands (k-step kk <-> column 4 fk + kk: 32 contiguous bytes per lane)
__device__ __forceinline__ void qrh_load_rows(const double* __restrict__ A, long ld, int j0, int rb, int M, double (&a)[4][4], int lo = 0) {
  const int lane = threadIdx.x & 63, fx = lane & 15, fk = lane >> 4;
  const bool vec = (ld & 1) == 0;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int ra = rb + q * 16 + fx;
    const double* src = A + (long)ra * ld + j0 + 4 * fk;
    if (ra < M && ra >= lo) {
      if (vec) {
        const double2 v0 = *reinterpret_cast<const double2*>(src), v1 = *reinterpret_cast<const double2*>(src + 2);
        a[q][0] = v0.x; a[q][1] = v0.y; a[q][2] = v1.x; a[q][3] = v1.y;
      } else { a[q][0] = src[0]; a[q][1] = src[1]; a[q][2] = src[2]; a[q][3] = src[3]; }
    } else { a[q][0] = a[q][1] = a[q][2] = a[q][3] = 0.0; }
  }
}

// ---- phase B: R1 = chol(G), Q1 = C R1^-1 in place, partial Gram matrices of Q1; decides the fall-back ----
__global__ __launch_bounds__(512) void qrh_chol(const QrhP P) {
  __shared__ double s_buf[QRH_SMEM];
  __shared__ double s_G[256], s_R[256], s_Ri[256], s_db[16];
  __shared__ int s_flag;
  const int mat = blockIdx.y;
  if ((int)blockIdx.x >= P.nrow) { qrh_side(P, mat, (int)blockIdx.x - P.nrow, s_buf); return; }
  const int g = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6, fx = lane & 15, fk = lane >> 4;
  double* A = P.Wm + mat * P.strideW;
  const int j0 = P.j0, M = P.M;
  const long ld = P.ld;
  const int rb = j0 + (g * 8 + wave) * 64;
  double a[4][4];
  qrh_load_rows(A, ld, j0, rb, M, a);
  qrh_stamp(P, 0);
  if (t < 256) {
    const double gb = qrh_sum_parts(P.Gp + mat * P.strideGp + 256 + t, P.ngp);
    s_G[t] = gb + P.Gp[mat * P.strideGp + t];
    if (t % 17 == 0) s_db[t / 17] = gb;                               // column sums of squares below the top block
  }
  __syncthreads();
  qrh_stamp(P, 1);
  if (wave == 0) {
    __builtin_amdgcn_s_setprio(3);                                     // the chain wave shares its CU with riding side workgroups
    bool ok = ND4_CHOL16(s_G, s_R, s_Ri, HR_PIVOT_THR);
    __builtin_amdgcn_s_setprio(0);
#pragma unroll
    for (int k = 0; k < 16; k++) ok = ok && (s_db[k] > 0.0);
    if (lane == 0) s_flag = ok ? 0 : 1;
  }
  __syncthreads();
  const int flag = s_flag;
  qrh_stamp(P, 2);
  if (g == 0) {
    if (t == 0) P.flag[mat] = flag;
    if (t < 256) P.R1[(long)mat * 256 + t] = s_R[t];
  }
  if (flag) return;
  double bw[4];
#pragma unroll
  for (int kk = 0; kk < 4; kk++) bw[kk] = s_Ri[(4 * fk + kk) * 16 + fx];
  d4 acc[4];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    acc[q] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < 4; kk++) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q][kk], bw[kk], acc[q], 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int rc = rb + q * 16 + fk + 4 * r;
      if (rc < M) A[(long)rc * ld + j0 + fx] = acc[q][r];
    }
  }
  d4 g0 = d4{0.0, 0.0, 0.0, 0.0}, g1 = g0;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    g0 = __builtin_amdgcn_mfma_f64_16x16x4f64(acc[q][0], acc[q][0], g0, 0, 0, 0);
    g1 = __builtin_amdgcn_mfma_f64_16x16x4f64(acc[q][1], acc[q][1], g1, 0, 0, 0);
    g0 = __builtin_amdgcn_mfma_f64_16x16x4f64(acc[q][2], acc[q][2], g0, 0, 0, 0);
    g1 = __builtin_amdgcn_mfma_f64_16x16x4f64(acc[q][3], acc[q][3], g1, 0, 0, 0);
  }
  qrh_reduce_store(s_buf, g0, g1, P.G2p + mat * P.strideG2 + (long)g * 256);
  qrh_stamp(P, 4);
}

// ---- phase C: R2, the representation (W, K, S R), and the partials of X = V^T C for the next panel's columns ----
template <int R>
__global__ __launch_bounds__(512) void qrh_reconstruct(const QrhP P) {
  __shared__ double s_buf[QRH_SMEM];
  const int mat = blockIdx.y;
  if ((int)blockIdx.x >= P.nrow) { qrh_side(P, mat, (int)blockIdx.x - P.nrow, s_buf); return; }
  const int g = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6, fx = lane & 15, fk = lane >> 4;
  double* A = P.Wm + mat * P.strideW;
  const int j0 = P.j0, M = P.M, N = P.N;
  const long ld = P.ld;
  const int c0 = j0 + NB, nc = P.skip_x ? 0 : (N - c0 < NB ? N - c0 : NB);   // the next panel's columns (nc <= 0: none)
  double* Xdst = P.Xp + mat * P.strideXp + (long)g * 256;
  if (P.flag[mat]) {
    if constexpr (R == 0) return;                                      // tall panel: qr_panel_flagged + qrh_x_flagged follow
    else if (g == 0) {
      qr_panel_row_body<(R > 0 ? R : 1)>(mat, P.Wm, M, ld, P.strideW, P.Vall, P.ldv, P.strideV, P.Tall, P.strideT, P.taus, P.strideTau, j0, NB);
      __threadfence();
      __syncthreads();
      if (nc > 0) qrh_x_full(s_buf, P.Vall + mat * P.strideV + (long)j0 * P.ldv + j0, P.ldv, A + (long)j0 * ld + c0, ld, M - j0, nc, Xdst);
    } else if (nc > 0 && t < 256) Xdst[t] = 0.0;
    return;
  }
  double* s_E = s_buf + QRH_LDS; double* s_F = s_E + 256; double* s_P = s_F + 256; double* s_R1 = s_P + 256; double* s_Qt = s_R1 + 256;
  double* s_R2 = s_Qt + 256; double* s_R2i = s_R2 + 256; double* s_Z = s_R2i + 256; double* s_Rm = s_Z + 256; double* s_K = s_Rm + 256; double* s_S = s_K + 256;
  __shared__ int s_emax;
  const int rb = j0 + (g * 8 + wave) * 64;
  double a[4][4]; d4 cs[4];
  qrh_load_rows(A, ld, j0, rb, M, a);
  qrh_stamp(P, 0);
#pragma unroll
  for (int q = 0; q < 4; q++) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int rc = rb + q * 16 + fk + 4 * r;
      cs[q][r] = (rc < M && fx < nc) ? A[(long)rc * ld + c0 + fx] : 0.0;
    }
  }
  const int i = (t & 255) / 16, j = t % 16;
  double x = 0.0, r1v = 0.0, qtv = 0.0;
  if (t < 256) {
    x = qrh_sum_parts(P.G2p + mat * P.strideG2 + t, P.ngp);
    r1v = P.R1[(long)mat * 256 + t];
    qtv = A[(long)(j0 + i) * ld + j0 + j];
  }
  if (t == 0) s_emax = 0;
  __syncthreads();
  if (t < 256) {
    x -= (i == j) ? 1.0 : 0.0;                                         // E = Q1^T Q1 - I
    s_E[t] = x;
    s_F[t] = (i < j) ? x : ((i == j) ? 0.5 * x : 0.0);
    s_R1[t] = r1v;
    s_Qt[t] = qtv;
    const float ax = fabsf((float)x);
    atomicMax(&s_emax, (ax == ax) ? __float_as_int(ax) : 0x7f800000);
  }
  __syncthreads();
  const bool series = __int_as_float(s_emax) <= (float)HR_SERIES_MAX;
  qrh_stamp(P, 1);
  if (series) {
    // R2 = I + F with F = triu(E - F^T F) (diagonal halved), two fixed-point steps from F = triu(E): error O(|E|^3);
    // R2^-1 = (I - F)(I + F^2) = I - F + F^2 - F^3: error O(|F|^4)
    if (t < 256) {
      double pp = 0.0;
#pragma unroll
      for (int l = 0; l < 16; l++) pp += s_F[l * 16 + i] * s_F[l * 16 + j];
      s_P[t] = pp;
    }
    __syncthreads();
    if (t < 256) {
      const double x = s_E[t] - s_P[t];
      s_F[t] = (i < j) ? x : ((i == j) ? 0.5 * x : 0.0);
    }
    __syncthreads();
    if (t < 256) {
      double pp = 0.0;
#pragma unroll
      for (int l = 0; l < 16; l++) pp += s_F[i * 16 + l] * s_F[l * 16 + j];
      s_P[t] = pp + ((i == j) ? 1.0 : 0.0);                            // I + F^2
      s_R2[t] = s_F[t] + ((i == j) ? 1.0 : 0.0);
    }
    __syncthreads();
    if (t < 256) {
      double pp = 0.0;
#pragma unroll
      for (int l = 0; l < 16; l++) pp += (((i == l) ? 1.0 : 0.0) - s_F[i * 16 + l]) * s_P[l * 16 + j];
      s_R2i[t] = pp;
    }
  } else {
    if (t < 256) s_E[t] += (i == j) ? 1.0 : 0.0;
    __syncthreads();
    if (wave == 0) (void)ND4_CHOL16(s_E, s_R2, s_R2i, 0.0);
  }
  __syncthreads();
  if (t < 256) {
    double z = 0.0, rr = 0.0;
#pragma unroll
    for (int l = 0; l < 16; l++) {
      z += s_Qt[i * 16 + l] * s_R2i[l * 16 + j];                       // top block of Q = Q1 R2^-1
      rr += s_R2[i * 16 + l] * s_R1[l * 16 + j];                       // R = R2 R1
    }
    s_Z[t] = z; s_Rm[t] = rr;
  }
  __syncthreads();
  qrh_stamp(P, 3);
  if (wave == 0) { __builtin_amdgcn_s_setprio(3); ND4_GJ16(s_Z, s_K, s_S); __builtin_amdgcn_s_setprio(0); }
  double bw[4];
#pragma unroll
  for (int kk = 0; kk < 4; kk++) bw[kk] = s_R2i[(4 * fk + kk) * 16 + fx];
  d4 y[4];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    y[q] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < 4; kk++) y[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q][kk], bw[kk], y[q], 0, 0, 0);
  }
  __syncthreads();                                                     // s_K, s_S
  qrh_stamp(P, 4);
  double* V = P.Vall + mat * P.strideV + j0;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const bool top = (g == 0 && wave == 0 && q == 0);
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int ii = fk + 4 * r, rc = rb + q * 16 + ii;
      double wv = 0.0;
      if (top) {                                                       // W = Q - [S; 0]; R = S R2 R1 in place
        if (ii == fx) y[q][r] -= s_S[ii];
        wv = (ii <= fx) ? s_S[ii] * s_Rm[ii * 16 + fx] : 0.0;
      }
      if (rc < M) { V[(long)rc * P.ldv + fx] = y[q][r]; A[(long)rc * ld + j0 + fx] = wv; }
    }
  }
  if (g == 0) {
    if (t < 256) P.Tall[mat * P.strideT + (long)(j0 / NB) * NB * NB + t] = s_K[t];
    if (t < 16) P.taus[mat * P.strideTau + j0 + t] = 1.0;             // "a reflector was needed" (qr_flips)
  }
  if (nc > 0) {
    d4 x0 = d4{0.0, 0.0, 0.0, 0.0}, x1 = x0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      x0 = __builtin_amdgcn_mfma_f64_16x16x4f64(y[q][0], cs[q][0], x0, 0, 0, 0);
      x1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y[q][1], cs[q][1], x1, 0, 0, 0);
      x0 = __builtin_amdgcn_mfma_f64_16x16x4f64(y[q][2], cs[q][2], x0, 0, 0, 0);
      x1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y[q][3], cs[q][3], x1, 0, 0, 0);
    }
    qrh_reduce_store(s_buf, x0, x1, Xdst);
  }
  qrh_stamp(P, 5);
  qrh_stamp(P, 6);
}

// ---- phases B and C in ONE launch (round 3, second half) ----
// What phase C needs from the other row workgroups of phase B is one 16 x 16 matrix each (the partial Gram matrix of Q1). An exchange
// of that size inside a kernel costs ~1.4 us on this chip when it is made of agent-scope relaxed atomics only (sc1 stores are
// written through, sc1 loads miss the non-coherent caches: no write-back, no invalidate; tools/xwg_lat.hip, lu.hip: lu_panel_mw) —
// less than the kernel boundary it replaces, and the panel's rows need not be read a third time: Q1 stays in the registers (its
// accumulator image goes through LDS once to become the A operand of Q = Q1 R2^-1). Every 8-byte word that crosses carries 32 bits
// of payload and a 32-bit tag (panel number + 1): a double is valid as soon as both its words carry the tag, so there is no flag,
// no s_waitcnt and no ordering between words. The row partition is phase A's (rows from j0 - 16: tile 0 of workgroup 0 lies above
// the panel and only sees the previous reflector). The next panel's 16 columns get the previous reflector from the ROW workgroups
// here (X summed from the side partials of launch A), each for its own rows, which it then holds for the partial X of the next
// panel: the side work of this launch starts one block later (na_shift) and covers W and Q^T in one go.
// A flagged panel: workgroup 0 runs qr_panel_row_body (R > 0; tall panels: qr_panel_flagged in a launch of its own, as before) and
// nobody writes partials of X: the next launch A forms X itself (qrh_gram), the last panel's narrow update goes through qrh_x_flagged.
constexpr long QX_ROW_SLOT = 2560;   // words per row workgroup: values [0,256) G, [256,512) X of the next block, [512,768) top-tile Gram, 768 "stores are out", [1024,1280) Gram of Q1
// MERGED: phase A in the same launch as well — the previous reflector on the panel's own columns, the partial Gram matrices and the
// partial X of the NEXT panel's block cross in a first exchange, so a panel is ONE launch (all side work rides in it).
template <int R, bool MERGED>
__global__ __launch_bounds__(512) void qrh_bc(const QrhP P) {
  __shared__ double s_buf[QRH_SMEM];
  __shared__ double s_G[256], s_R[256], s_Ri[256], s_db[16];
  __shared__ int s_flag, s_emax;
  const int mat = blockIdx.y;
  if ((int)blockIdx.x >= P.nrow) { qrh_side(P, mat, (int)blockIdx.x - P.nrow, s_buf); return; }
  const int g = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6, fx = lane & 15, fk = lane >> 4;
  double* A = P.Wm + mat * P.strideW;
  const int j0 = P.j0, M = P.M, N = P.N, ub = j0 - NB;
  const long ld = P.ld;
  const int c0 = j0 + NB, nc = P.skip_x ? 0 : (N - c0 < NB ? N - c0 : NB);   // the next panel's columns (nc <= 0: none)
  const int rb = ub + (g * 8 + wave) * 64;                                  // phase A's row partition
  const unsigned tag = (unsigned)(j0 / NB + 1);
  qx_u64* slots = P.Xch + mat * P.strideXch;
  qx_u64* myslot = slots + (long)g * QX_ROW_SLOT;
  double* Xdst = P.Xp + mat * P.strideXp + (long)g * 256;
  qrh_stamp(P, 0);
  d4 csb[1][4];
  d4 (&cs)[4] = csb[0];
  double a[4][4];
  __shared__ double s_T2[256], s_Xs[256], s_Xn[256];
  d4 cnx[4]; double avp[4][4]; bool nextupd = false;       // MERGED: the next block's tile, the previous reflector's rows
  // the next block through the previous reflector for this workgroup's rows: C -= V (T^T X), -T^T X in s_Xn
  auto apply_next = [&]() {
    double bwn[4];
#pragma unroll
    for (int kk = 0; kk < 4; kk++) bwn[kk] = s_Xn[(4 * fk + kk) * 16 + fx];
#pragma unroll
    for (int q = 0; q < 4; q++) {
#pragma unroll
      for (int kk = 0; kk < 4; kk++) cnx[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(avp[q][kk], bwn[kk], cnx[q], 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int rc = rb + q * 16 + fk + 4 * r;
        if (rc >= 0 && rc < M && fx < nc) A[(long)rc * ld + c0 + fx] = cnx[q][r];
      }
    }
  };
  if constexpr (MERGED) {
    // ---- phase A: the previous reflector on the panel's own columns, partial Gram matrices, partial X of the next block ----
    // All loads first: the own tile and the next block's tile in the accumulator image (== slab image: slab u = 4 q + r), the
    // previous reflector's rows as A operands (avp) and as slabs (vs). The next block's tile and avp stay in registers until the
    // block is updated in the shadow of the second exchange.
    d4 c[4];
    const bool havep = P.pj0 >= 0;
    nextupd = nc > 0 && havep;
    const double* Vp = P.Vall + mat * P.strideV + ub;
    const bool prevflag = havep && P.flag[mat] != 0;                  // the previous panel was flagged: no partials of X (see qrh_gram)
    // first the loads that head the dependent chain of the own-column update (loads return in order): X partials and T
    double xsum0 = 0.0, tprev = 0.0, xp8[8];
#pragma unroll
    for (int p8 = 0; p8 < 8; p8++) xp8[p8] = 0.0;
    if (havep && t < 256) {
      if (!prevflag) {
#pragma unroll
        for (int p8 = 0; p8 < 8; p8++) if (p8 < P.nxp) xp8[p8] = P.Xp[mat * P.strideXp + (long)p8 * 256 + t];   // (summed below, after the tile loads are under way)
      }
      tprev = P.Tall[mat * P.strideT + (long)(ub / NB) * NB * NB + t];
    }
    double vs[16];
#pragma unroll
    for (int q = 0; q < 4; q++) {
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int rc = rb + q * 16 + fk + 4 * r;
        const bool rok = rc >= 0 && rc < M;
        c[q][r] = rok ? A[(long)rc * ld + j0 + fx] : 0.0;
        cnx[q][r] = (rok && fx < nc && (nextupd || rc >= j0)) ? A[(long)rc * ld + c0 + fx] : 0.0;
        vs[4 * q + r] = (rok && nextupd) ? Vp[(long)rc * P.ldv + fx] : 0.0;
      }
      const int ra = rb + q * 16 + fx;
      if (havep && ra >= 0 && ra < M) {
        const double2 v0 = *reinterpret_cast<const double2*>(Vp + (long)ra * P.ldv + 4 * fk);
        const double2 v1 = *reinterpret_cast<const double2*>(Vp + (long)ra * P.ldv + 4 * fk + 2);
        avp[q][0] = v0.x; avp[q][1] = v0.y; avp[q][2] = v1.x; avp[q][3] = v1.y;
      } else { avp[q][0] = avp[q][1] = avp[q][2] = avp[q][3] = 0.0; }
    }
    const int i = (t & 255) / 16, j = t % 16;
    if (havep) {
      if (prevflag) {
        double* mine = P.Xp + mat * P.strideXp + (long)g * 256;
        qrh_x_full(s_buf, P.Vall + mat * P.strideV + (long)ub * P.ldv + ub, P.ldv, A + (long)ub * ld + j0, ld, M - ub, NB, mine);
        __syncthreads();
        if (t < 256) xsum0 = mine[t];
        __syncthreads();
      }
      if (t < 256) {
        if (!prevflag) {
#pragma unroll
          for (int p8 = 0; p8 < 8; p8++) xsum0 += xp8[p8];
          if (P.nxp > 8) xsum0 += qrh_sum_parts(P.Xp + mat * P.strideXp + 8 * 256 + t, P.nxp - 8);
        }
        s_buf[t] = xsum0; s_T2[t] = tprev;
      }
      __syncthreads();
      double wv = 0.0;
      if (t < 256) {
#pragma unroll
        for (int l = 0; l < NB; l++) wv += s_T2[l * 16 + i] * s_buf[l * 16 + j];               // T^T X
      }
      __syncthreads();
      if (t < 256) s_buf[t] = -wv;
      __syncthreads();
      double bwp[4];
#pragma unroll
      for (int kk = 0; kk < 4; kk++) bwp[kk] = s_buf[(4 * fk + kk) * 16 + fx];
#pragma unroll
      for (int q = 0; q < 4; q++) {
#pragma unroll
        for (int kk = 0; kk < 4; kk++) c[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(avp[q][kk], bwp[kk], c[q], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int rc = rb + q * 16 + fk + 4 * r;
          if (rc >= 0 && rc < M) A[(long)rc * ld + j0 + fx] = c[q][r];
        }
      }
      __syncthreads();                                                  // s_buf is reused below
    }
    qrh_stamp(P, 1);
    d4 g0 = d4{0.0, 0.0, 0.0, 0.0}, g1 = g0, gt = g0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int ti = (g * 8 + wave) * 4 + q;                          // tile 0: the 16 rows above the panel, tile 1: its top block
      if (ti == 1) {
#pragma unroll
        for (int r = 0; r < 4; r++) gt = __builtin_amdgcn_mfma_f64_16x16x4f64(c[q][r], c[q][r], gt, 0, 0, 0);
      } else if (ti >= 2) {
        g0 = __builtin_amdgcn_mfma_f64_16x16x4f64(c[q][0], c[q][0], g0, 0, 0, 0);
        g1 = __builtin_amdgcn_mfma_f64_16x16x4f64(c[q][1], c[q][1], g1, 0, 0, 0);
        g0 = __builtin_amdgcn_mfma_f64_16x16x4f64(c[q][2], c[q][2], g0, 0, 0, 0);
        g1 = __builtin_amdgcn_mfma_f64_16x16x4f64(c[q][3], c[q][3], g1, 0, 0, 0);
      }
    }
    // partial X = V^T C of the next panel's block over this workgroup's rows (the previous reflector; slab images)
    d4 xn0 = d4{0.0, 0.0, 0.0, 0.0}, xn1 = xn0;
    if (nextupd) {
#pragma unroll
      for (int q = 0; q < 4; q++) {
        xn0 = __builtin_amdgcn_mfma_f64_16x16x4f64(vs[4 * q + 0], cnx[q][0], xn0, 0, 0, 0);
        xn1 = __builtin_amdgcn_mfma_f64_16x16x4f64(vs[4 * q + 1], cnx[q][1], xn1, 0, 0, 0);
        xn0 = __builtin_amdgcn_mfma_f64_16x16x4f64(vs[4 * q + 2], cnx[q][2], xn0, 0, 0, 0);
        xn1 = __builtin_amdgcn_mfma_f64_16x16x4f64(vs[4 * q + 3], cnx[q][3], xn1, 0, 0, 0);
      }
    }
    // the panel tile from its accumulator image to the A-operand image (tile 0 lies above the panel: zero), through the wave's own LDS
    {
      double* sw = s_buf + 8 * 256 + wave * 272;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int ti = (g * 8 + wave) * 4 + q;
#pragma unroll
        for (int r = 0; r < 4; r++) sw[(fk + 4 * r) * 17 + fx] = ti >= 1 ? c[q][r] : 0.0;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int kk = 0; kk < 4; kk++) a[q][kk] = sw[fx * 17 + 4 * fk + kk];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    }
    // first exchange: G partial, X partial, (workgroup 0, wave 0) the top tile's Gram matrix
#pragma unroll
    for (int r = 0; r < 4; r++) s_buf[wave * 256 + (fk + 4 * r) * 16 + fx] = g0[r] + g1[r];
    if (g == 0 && wave == 0) {
#pragma unroll
      for (int r = 0; r < 4; r++) qx_st(myslot, 512 + (fk + 4 * r) * 16 + fx, gt[r], tag);
    }
    __syncthreads();
    if (t < 256) {
      double xs = 0.0;
#pragma unroll
      for (int w = 0; w < 8; w++) xs += s_buf[w * 256 + t];
      if (!(P.drop_tag == (int)tag && g == P.nrow - 1)) qx_st(myslot, t, xs, tag);   // (tests: one dropped publication)
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; r++) s_buf[wave * 256 + (fk + 4 * r) * 16 + fx] = xn0[r] + xn1[r];
    __syncthreads();
    if (t >= 256) {
      double xs = 0.0;
#pragma unroll
      for (int w = 0; w < 8; w++) xs += s_buf[w * 256 + (t - 256)];
      qx_st(myslot, t, xs, tag);
    }
    qrh_stamp(P, 2);
    if (t < 256) {
      double x = 0.0, xn = 0.0, top = 0.0;
      int spins = 0;
      for (;;) {
        bool ok = true;
        x = qx_sum(slots, t, P.nrow, tag, ok, QX_ROW_SLOT);
        if (nextupd) xn = qx_sum(slots, 256 + t, P.nrow, tag, ok, QX_ROW_SLOT);
        const qx_u64 w0 = __hip_atomic_load(slots + 2 * (512 + t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const qx_u64 w1 = __hip_atomic_load(slots + 2 * (512 + t) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = ok && (unsigned)(w0 >> 32) == tag && (unsigned)(w1 >> 32) == tag;
        top = __longlong_as_double((long long)((w1 << 32) | (w0 & 0xffffffffull)));
        if (ok) break;
        if (++spins > QX_SPIN_LIMIT) { x = __builtin_nan(""); qx_raise(P.status); break; }    // a stuck exchange must not pass silently
        __builtin_amdgcn_s_sleep(2);
      }
      s_G[t] = x + top;
      if (t % 17 == 0) s_db[t / 17] = x;                               // column sums of squares below the top block
      s_Xs[t] = xn;
    }
    if (t == 0) s_emax = 0;
    __syncthreads();
    qrh_stamp(P, 3);
    // while wave 0 walks the elimination chain below, waves 4..7 form -T^T X for the next block
    if (nextupd && t >= 256) {
      double wv = 0.0;
#pragma unroll
      for (int l = 0; l < NB; l++) wv += s_T2[l * 16 + i] * s_Xs[l * 16 + j];
      s_Xn[t - 256] = -wv;
    }
  } else {
    // ---- the next panel's columns: the previous reflector for this workgroup's rows (kept in cs for the partial X below) ----
    if (nc > 0 && P.pj0 >= 0) {
      qrh_apply_rows<1>(s_buf, P.Xs + mat * P.strideXs, 0, P.nrc, P.Tall + mat * P.strideT + (long)(ub / NB) * NB * NB,
                        P.Vall + mat * P.strideV + ub, P.ldv, A + c0, ld, nc, rb, M, csb);
      __syncthreads();
    } else {
#pragma unroll
      for (int q = 0; q < 4; q++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int rc = rb + q * 16 + fk + 4 * r;
          cs[q][r] = (rc >= j0 && rc < M && fx < nc) ? A[(long)rc * ld + c0 + fx] : 0.0;
        }
      }
    }
    qrh_load_rows(A, ld, j0, rb, M, a, j0);
    if (t < 256) {
      const double gb = qrh_sum_parts(P.Gp + mat * P.strideGp + 256 + t, P.ngp);
      s_G[t] = gb + P.Gp[mat * P.strideGp + t];
      if (t % 17 == 0) s_db[t / 17] = gb;                             // column sums of squares below the top block
    }
    if (t == 0) s_emax = 0;
    __syncthreads();
    qrh_stamp(P, 1);
  }
  // ---- phase B: R1 = chol(G), the fall-back decision ----
  if (wave == 0) {
    __builtin_amdgcn_s_setprio(3);
    bool ok = ND4_CHOL16(s_G, s_R, s_Ri, HR_PIVOT_THR);
    __builtin_amdgcn_s_setprio(0);
#pragma unroll
    for (int k = 0; k < 16; k++) ok = ok && (s_db[k] > 0.0);
    if (lane == 0) s_flag = ok ? 0 : 1;
  }
  __syncthreads();
  const int flag = s_flag;                                            // (the same in every row workgroup: same sums in the same order)
  if (g == 0 && t == 0) P.flag[mat] = flag;
  if (flag) {
    if constexpr (MERGED) { if (nextupd) apply_next(); }          // (the next launch needs the block whatever happens to this panel)
    if constexpr (R > 0) {
      if constexpr (MERGED) {
        // the panel's columns were written by all row workgroups of THIS launch: every workgroup pushes its stores out (release:
        // write back the L2) and says so; workgroup 0 waits for all, invalidates (acquire) and runs the classic panel. Rare path.
        __threadfence();
        __syncthreads();
        if (t == 0) qx_st(myslot, 768, 1.0, tag);
        if (g == 0) {
          if (t < P.nrow) {
            int spins = 0;
            for (;;) {
              bool ok = true;
              (void)qx_sum(slots + (long)t * QX_ROW_SLOT, 768, 1, tag, ok, QX_ROW_SLOT);
              if (ok) break;
              if (++spins > QX_SPIN_LIMIT) { qx_raise(P.status); break; }
              __builtin_amdgcn_s_sleep(4);
            }
          }
          __syncthreads();
          __threadfence();
        }
      }
      if (g == 0) qr_panel_row_body<(R > 0 ? R : 1)>(mat, P.Wm, M, ld, P.strideW, P.Vall, P.ldv, P.strideV, P.Tall, P.strideT, P.taus, P.strideTau, j0, NB);
    }
    return;
  }
  // ---- Q1 = C R1^-1 (registers), its partial Gram matrix across to the other row workgroups ----
  double bw[4];
#pragma unroll
  for (int kk = 0; kk < 4; kk++) bw[kk] = s_Ri[(4 * fk + kk) * 16 + fx];
  d4 acc[4];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    acc[q] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < 4; kk++) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q][kk], bw[kk], acc[q], 0, 0, 0);
  }
  d4 g0 = d4{0.0, 0.0, 0.0, 0.0}, g1 = g0;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    g0 = __builtin_amdgcn_mfma_f64_16x16x4f64(acc[q][0], acc[q][0], g0, 0, 0, 0);
    g1 = __builtin_amdgcn_mfma_f64_16x16x4f64(acc[q][1], acc[q][1], g1, 0, 0, 0);
    g0 = __builtin_amdgcn_mfma_f64_16x16x4f64(acc[q][2], acc[q][2], g0, 0, 0, 0);
    g1 = __builtin_amdgcn_mfma_f64_16x16x4f64(acc[q][3], acc[q][3], g1, 0, 0, 0);
  }
  double* s_E = s_buf + QRH_LDS; double* s_F = s_E + 256; double* s_P = s_F + 256; double* s_Qt = s_P + 512;
  double* s_R2 = s_Qt + 256; double* s_R2i = s_R2 + 256; double* s_Z = s_R2i + 256; double* s_Rm = s_Z + 256; double* s_K = s_Rm + 256; double* s_S = s_K + 256;
#pragma unroll
  for (int r = 0; r < 4; r++) s_buf[wave * 256 + (fk + 4 * r) * 16 + fx] = g0[r] + g1[r];
  if (g == 0 && wave == 0) {                                           // top block of Q1 (tile 1)
#pragma unroll
    for (int r = 0; r < 4; r++) s_Qt[(fk + 4 * r) * 16 + fx] = acc[1][r];
  }
  __syncthreads();
  const int i = (t & 255) / 16, j = t % 16;
  if (t < 256) {
    double xs = 0.0;
#pragma unroll
    for (int w = 0; w < 8; w++) xs += s_buf[w * 256 + t];
    qx_st(myslot, 1024 + t, xs, tag);
  }
  if constexpr (MERGED) {                                              // in the shadow of the exchange: the next block
    if (nextupd) apply_next();
#pragma unroll
    for (int q = 0; q < 4; q++) cs[q] = cnx[q];
  }
  qrh_stamp(P, 3);
  double x = 0.0;
  if (t < 256) {
    int spins = 0;
    for (;;) {
      bool ok = true;
      x = qx_sum(slots, 1024 + t, P.nrow, tag, ok, QX_ROW_SLOT);
      if (ok) break;
      if (++spins > QX_SPIN_LIMIT) { x = __builtin_nan(""); qx_raise(P.status); break; }      // a stuck exchange must not pass silently
      __builtin_amdgcn_s_sleep(2);
    }
    x -= (i == j) ? 1.0 : 0.0;                                         // E = Q1^T Q1 - I
    s_E[t] = x;
    s_F[t] = (i < j) ? x : ((i == j) ? 0.5 * x : 0.0);
    const float ax = fabsf((float)x);
    atomicMax(&s_emax, (ax == ax) ? __float_as_int(ax) : 0x7f800000);
  }
  __syncthreads();
  const bool series = __int_as_float(s_emax) <= (float)HR_SERIES_MAX;
  qrh_stamp(P, 4);
  // Q1 from its accumulator image to the A-operand image, one tile at a time through the wave's own piece of LDS (rows padded to 17
  // doubles; the DS operations of one wave execute in order, the waits keep compiler and hardware from overlapping write and read;
  // the eight partial matrices that lay there were consumed before the barrier above)
  {
    double* sw = s_buf + wave * 272;                                   // 8 x 272 doubles: fits in front of QRH_LDS
#pragma unroll
    for (int q = 0; q < 4; q++) {
#pragma unroll
      for (int r = 0; r < 4; r++) sw[(fk + 4 * r) * 17 + fx] = acc[q][r];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int kk = 0; kk < 4; kk++) a[q][kk] = sw[fx * 17 + 4 * fk + kk];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
#ifndef ND4HIP_QR_OLD_CHAINS
  // R2 and R2^-1 on wave 0 (the products of the series on the matrix core, qr_chain16.h), then — workgroup 0 — the top block of Q,
  // R = R2 R1 and the elimination behind S and K, while the other waves form Q = Q1 R2^-1
  if (wave == 0) qrc_series16(s_E, series, s_R2, s_R2i, s_F);
  __syncthreads();
  if (g == 0 && wave == 0) {
    __builtin_amdgcn_s_setprio(3);
    qrc_zr16(s_Qt, s_R2i, s_R2, s_R, s_Z, s_Rm);
    qrh_stamp(P, 5);
    qrc_gj16(s_Z, s_K, s_S);
    __builtin_amdgcn_s_setprio(0);
  }
#else
  if (series) {
    if (t < 256) {
      double pp = 0.0;
#pragma unroll
      for (int l = 0; l < 16; l++) pp += s_F[l * 16 + i] * s_F[l * 16 + j];
      s_P[t] = pp;
    }
    __syncthreads();
    if (t < 256) {
      const double xx = s_E[t] - s_P[t];
      s_F[t] = (i < j) ? xx : ((i == j) ? 0.5 * xx : 0.0);
    }
    __syncthreads();
    if (t < 256) {
      double pp = 0.0;
#pragma unroll
      for (int l = 0; l < 16; l++) pp += s_F[i * 16 + l] * s_F[l * 16 + j];
      s_P[t] = pp + ((i == j) ? 1.0 : 0.0);                            // I + F^2
      s_R2[t] = s_F[t] + ((i == j) ? 1.0 : 0.0);
    }
    __syncthreads();
    if (t < 256) {
      double pp = 0.0;
#pragma unroll
      for (int l = 0; l < 16; l++) pp += (((i == l) ? 1.0 : 0.0) - s_F[i * 16 + l]) * s_P[l * 16 + j];
      s_R2i[t] = pp;
    }
  } else {
    if (t < 256) s_E[t] += (i == j) ? 1.0 : 0.0;
    __syncthreads();
    if (wave == 0) (void)ND4_CHOL16(s_E, s_R2, s_R2i, 0.0);
  }
  __syncthreads();
  if (g == 0) {                                                        // only workgroup 0 holds the top block: Z, R, K, S
    if (t < 256) {
      double z = 0.0, rr = 0.0;
#pragma unroll
      for (int l = 0; l < 16; l++) {
        z += s_Qt[i * 16 + l] * s_R2i[l * 16 + j];                     // top block of Q = Q1 R2^-1
        rr += s_R2[i * 16 + l] * s_R[l * 16 + j];                      // R = R2 R1
      }
      s_Z[t] = z; s_Rm[t] = rr;
    }
    __syncthreads();
    qrh_stamp(P, 5);
    if (wave == 0) { __builtin_amdgcn_s_setprio(3); ND4_GJ16(s_Z, s_K, s_S); __builtin_amdgcn_s_setprio(0); }
  }
#endif
#pragma unroll
  for (int kk = 0; kk < 4; kk++) bw[kk] = s_R2i[(4 * fk + kk) * 16 + fx];
  d4 y[4];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    y[q] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < 4; kk++) y[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q][kk], bw[kk], y[q], 0, 0, 0);
  }
  if (g == 0) __syncthreads();                                         // s_K, s_S (uniform per workgroup)
  qrh_stamp(P, 6);
  double* V = P.Vall + mat * P.strideV + j0;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const bool top = (g == 0 && wave == 0 && q == 1);
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int ii = fk + 4 * r, rc = rb + q * 16 + ii;
      double wv = 0.0;
      if (top) {                                                       // W = Q - [S; 0]; R = S R2 R1 in place
        if (ii == fx) y[q][r] -= s_S[ii];
        wv = (ii <= fx) ? s_S[ii] * s_Rm[ii * 16 + fx] : 0.0;
      }
      if (rc >= j0 && rc < M) { V[(long)rc * P.ldv + fx] = y[q][r]; A[(long)rc * ld + j0 + fx] = wv; }
    }
  }
  if (g == 0) {
    if (t < 256) P.Tall[mat * P.strideT + (long)(j0 / NB) * NB * NB + t] = s_K[t];
    if (t < 16) P.taus[mat * P.strideTau + j0 + t] = 1.0;             // "a reflector was needed" (qr_flips)
  }
  if (nc > 0) {
    d4 x0 = d4{0.0, 0.0, 0.0, 0.0}, x1 = x0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      x0 = __builtin_amdgcn_mfma_f64_16x16x4f64(y[q][0], cs[q][0], x0, 0, 0, 0);
      x1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y[q][1], cs[q][1], x1, 0, 0, 0);
      x0 = __builtin_amdgcn_mfma_f64_16x16x4f64(y[q][2], cs[q][2], x0, 0, 0, 0);
      x1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y[q][3], cs[q][3], x1, 0, 0, 0);
    }
    __syncthreads();                                                   // (the relayout pieces in s_buf are consumed)
    qrh_reduce_store(s_buf, x0, x1, Xdst);
  }
  qrh_stamp(P, 7);
}

// partials of X = V^T C for the next panel's columns after qr_panel_flagged (workgroup 0: all rows; the others: zero)
__global__ __launch_bounds__(512) void qrh_x_flagged(const QrhP P) {
  __shared__ double s_buf[QRH_LDS];
  const int mat = blockIdx.y, g = blockIdx.x, t = threadIdx.x;
  if (!P.flag[mat]) return;
  const int j0 = P.j0, c0 = j0 + NB, nc = P.N - c0 < NB ? P.N - c0 : NB;
  if (nc <= 0) return;
  double* Xdst = P.Xp + mat * P.strideXp + (long)g * 256;
  if (g == 0) qrh_x_full(s_buf, P.Vall + mat * P.strideV + (long)j0 * P.ldv + j0, P.ldv, P.Wm + mat * P.strideW + (long)j0 * P.ld + c0, P.ld, P.M - j0, nc, Xdst);
  else if (t < 256) Xdst[t] = 0.0;
}

// ---- the same panel kernel for taller panels: 1024 threads leave 128 VGPRs per lane = R rows of W columns with R * W = 32:
// 2048 < m <= 4096 rows: R = 4, W = 8; 4096 < m <= 8192 rows: R = 8, W = 4.
// A 16-column panel slot is then factorised in parts (two 8-column halves / four 4-column quarters).
// Every part writes ONLY its own triangular factor into the live T slot (zeros elsewhere), so that the ordinary
// block-reflector machinery applies just this part's reflectors to the remaining columns of the slot, and also into a
// side slot that collects the diagonal blocks; qr_t_assemble then builds the full 16 x 16 factor from them and V^T V.
template <int W> __device__ __forceinline__ double qr_xor_lanes(double v) {       // value of lane ^ W inside a row of 16 lanes
  if constexpr (W == 1) return nd4dpp::xor1(v);
  else if constexpr (W == 2) return nd4dpp::xor2(v);
  else if constexpr (W == 4) return nd4dpp::xor4(v);
  else return nd4dpp::xor8(v);
}
template <int R, int W8>
__global__ __launch_bounds__(1024) void qr_panel_part(double* __restrict__ Wm, int M, long ld, long strideW,
                                                     double* __restrict__ Vall, long ldv, long strideV,
                                                     double* __restrict__ Tall, double* __restrict__ Tside, long strideT,
                                                     double* __restrict__ taus, long strideTau, int c0, int nb) {
  constexpr int NW = 16;                               // waves per workgroup (W8 = columns of this kernel)
  static_assert(W8 == 8 || W8 == 4, "two halves or four quarters of a 16-column slot");
  __shared__ double s_red[NW];
  __shared__ double s_w[NW][W8];
  __shared__ double s_T[W8][W8 + 1];
  __shared__ double s_Z[W8][W8];
  __shared__ double s_tau[W8];
  const int j0 = c0;                                   // rows and columns of this half start at its own diagonal
  __shared__ double s_alpha;
  double* A = Wm + blockIdx.x * strideW;
  double* V = Vall + blockIdx.x * strideV;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4;
  const int mycol = W8 == 8 ? (b0 ? 4 : 0) + (b1 ? 2 : 0) + (b2 ? 1 : 0) : (b0 ? 2 : 0) + (b1 ? 1 : 0);   // column this lane ends up with

  double a[R][W8];
#pragma unroll
  for (int i = 0; i < R; i++) {
    const int r = j0 + t + 1024 * i;
#pragma unroll
    for (int c = 0; c < W8; c++) a[i][c] = 0.0;
    if (r < M) {
      const double* src = A + (long)r * ld + j0;
      if (nb == W8 && (ld & 1) == 0) {                   // 16-byte loads of the lane's own 128-B row segment
#pragma unroll
        for (int c = 0; c < W8; c += 2) { const double2 v = *reinterpret_cast<const double2*>(src + c); a[i][c] = v.x; a[i][c + 1] = v.y; }
      } else {
#pragma unroll
        for (int c = 0; c < W8; c++) if (c < nb) a[i][c] = src[c];
      }
    }
  }
  if (t < W8 * (W8 + 1)) (&s_T[0][0])[t] = 0.0;

  // one column step per compile-time k (generic lambda, see lu.hip: convergent DPP ops block `#pragma unroll`)
  auto column_step = [&](auto kc) __attribute__((always_inline)) {
    constexpr int k = decltype(kc)::value;
    if (k < nb) {
      const int jc = j0 + k;
      double part = 0.0;
#pragma unroll
      for (int i = 0; i < R; i++) {
        const int r = j0 + t + 1024 * i;
        if (i > 0 || r > jc) part += a[i][k] * a[i][k];              // row slots >= 1 are always below row jc
      }
      if (t == k) s_alpha = a[0][k];
      part = nd4dpp::wave_sum(part);
      if (lane == 0) s_red[wave] = part;
      __syncthreads();
      double sigma = 0.0;
#pragma unroll
      for (int w = 0; w < NW; w++) sigma += s_red[w];
      const double alpha = s_alpha;
      double beta = alpha, tau = 0.0, scale = 0.0;
      if (sigma != 0.0) {
        // sqrt and the two divisions sit on the column's critical path (~100 dependent instructions): one rsqrt and one
        // rcp with two Newton steps each instead. The work copy is normalised to max|a| in [1,2), so nothing over/underflows;
        // a few ulp in (beta, tau, scale) perturb H by a few ulp, like the rounding of the update itself.
        const double nn = alpha * alpha + sigma, ri = nd4dpp::fast_rsqrt(nn);
        beta = -copysign(nn * ri, alpha);
        tau = (beta - alpha) * -copysign(ri, alpha);
        scale = nd4dpp::fast_rcp(alpha - beta);
      }
      double vr[R], d[W8];
#pragma unroll
      for (int c = 0; c < W8; c++) d[c] = 0.0;
#pragma unroll
      for (int i = 0; i < R; i++) {
        const int r = j0 + t + 1024 * i;
        vr[i] = (i > 0 || r > jc) ? a[i][k] * scale : ((r == jc) ? 1.0 : 0.0);
#pragma unroll
        for (int c = 0; c < W8; c++) d[c] += vr[i] * a[i][c];
      }
      // halving butterfly over the W8 lanes of a group (W8/2 + ... + 1 exchanges), then across the groups of the wave
      double e4[4], e2[2], e1;
      if constexpr (W8 == 8) {
#pragma unroll
        for (int j = 0; j < 4; j++) { const double snd = b0 ? d[j] : d[j + 4], kp = b0 ? d[j + 4] : d[j]; e4[j] = kp + nd4dpp::xor1(snd); }
      } else {
#pragma unroll
        for (int j = 0; j < 4; j++) e4[j] = d[j];
      }
      {
        const bool bb = W8 == 8 ? b1 : b0;
#pragma unroll
        for (int j = 0; j < 2; j++) { const double snd = bb ? e4[j] : e4[j + 2], kp = bb ? e4[j + 2] : e4[j]; e2[j] = kp + qr_xor_lanes<W8 / 4>(snd); }
      }
      { const bool bb = W8 == 8 ? b2 : b1; const double snd = bb ? e2[0] : e2[1], kp = bb ? e2[1] : e2[0]; e1 = kp + qr_xor_lanes<W8 / 2>(snd); }
      if constexpr (W8 == 4) e1 += nd4dpp::xor4(e1);
      e1 += nd4dpp::xor8(e1);
      e1 += __shfl_xor(e1, 16);
      e1 += __shfl_xor(e1, 32);
      if (lane < W8) s_w[wave][mycol] = e1;
      __syncthreads();
      double tot = 0.0;                                   // lane -> column lane & (W8 - 1)
#pragma unroll
      for (int w = 0; w < NW; w++) tot += s_w[w][lane & (W8 - 1)];
      double wv[W8];                                      // wave-uniform totals
#define ND4_RL(C) wv[C] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(tot), C), __builtin_amdgcn_readlane(__double2loint(tot), C));
      ND4_RL(0) ND4_RL(1) ND4_RL(2) ND4_RL(3)
      if constexpr (W8 == 8) { ND4_RL(4) ND4_RL(5) ND4_RL(6) ND4_RL(7) }
#undef ND4_RL
#pragma unroll
      for (int i = 0; i < R; i++) {
        const int r = j0 + t + 1024 * i;
        const double tv = tau * vr[i];
#pragma unroll
        for (int c = k + 1; c < W8; c++) a[i][c] -= tv * wv[c];
        a[i][k] = (i > 0 || r > jc) ? vr[i] : ((r == jc) ? beta : a[i][k]);
      }
      if (t == 0) {
#pragma unroll
        for (int c = 0; c < W8; c++) if (c < k) s_Z[k][c] = wv[c];
        s_tau[k] = tau;
        taus[blockIdx.x * strideTau + j0 + k] = tau;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };
#define ND4_STEP(K) column_step(std::integral_constant<int, K>{});
  ND4_STEP(0) ND4_STEP(1) ND4_STEP(2) ND4_STEP(3)
  if constexpr (W8 == 8) { ND4_STEP(4) ND4_STEP(5) ND4_STEP(6) ND4_STEP(7) }
#undef ND4_STEP
  __syncthreads();
  if (t < nb) {                                          // larft: row t of T depends only on row t
    double row[W8];
#pragma unroll
    for (int k = 0; k < W8; k++) row[k] = 0.0;
#pragma unroll
    for (int k = 0; k < W8; k++) {
      if (k == t) row[k] = s_tau[k];
      else if (k > t && k < nb) {
        double sum = 0.0;
#pragma unroll
        for (int j = 0; j < W8; j++) if (j >= t && j < k) sum += row[j] * s_Z[k][j];
        row[k] = -s_tau[k] * sum;
      }
    }
#pragma unroll
    for (int k = 0; k < W8; k++) s_T[t][k] = row[k];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < R; i++) {
    const int lr = t + 1024 * i, r = j0 + lr;
    if (r < M) {
      double* w = A + (long)r * ld + j0;
      double* v = V + (long)r * ldv + j0;
      double wv[W8], vv[W8];
#pragma unroll
      for (int c = 0; c < W8; c++) {
        wv[c] = (lr <= c) ? a[i][c] : 0.0;                       // R part (upper triangle incl. diagonal)
        vv[c] = (lr < c) ? 0.0 : ((lr == c) ? 1.0 : a[i][c]);    // explicit reflector: zeros above, unit diagonal
      }
      if (nb == W8 && (ld & 1) == 0) {
#pragma unroll
        for (int c = 0; c < W8; c += 2) {
          *reinterpret_cast<double2*>(w + c) = double2{wv[c], wv[c + 1]};
          *reinterpret_cast<double2*>(v + c) = double2{vv[c], vv[c + 1]};
        }
      } else {
#pragma unroll
        for (int c = 0; c < W8; c++) if (c < nb) { w[c] = wv[c]; v[c] = vv[c]; }
      }
    }
  }
  if (t < 16 * 16) {                                     // the 16 x 16 slot of the enclosing panel
    const int slot0 = (c0 / 16) * 16, part = (c0 - slot0) / W8;
    const int i = t / 16, j = t % 16;
    const long so = blockIdx.x * strideT + (long)(slot0 / 16) * 256;
    const bool mine = (i / W8 == part) && (j / W8 == part);
    const double val = (mine && (i % W8) <= (j % W8) && (j % W8) < nb) ? s_T[i % W8][j % W8] : 0.0;
    Tall[so + t] = val;                                  // live slot: ONLY this part's block (the partial block reflector)
    if (mine) Tside[so + t] = val;                       // side slot: collects the diagonal blocks of all parts
  }
}


// ------------------------------------------------------------------------------------ V^T C
// Wp[chunk][i][j] = sum_{r in chunk} V[r][i] * C[r][j]; V: m x 16 (ldv), C: m x n (ldc).
// grid (ceil(n/64), ceil(m/256), batch), 256 threads: wave w takes rows chunk*256 + w*64 .. +64 and
// all 4 column strips of 16; the 4 waves' tiles are summed through LDS.
__global__ __launch_bounds__(256) void qr_vtc(const double* __restrict__ V, long ldv, long strideV,
                                               const double* __restrict__ C, long ldc, long strideC,
                                               int m, int n, double* __restrict__ Wp, long ldw, long strideChunk, long strideWb) {
  __shared__ double s_acc[4][4][4][64];      // [wave][strip][reg][lane]
  V += blockIdx.z * strideV; C += blockIdx.z * strideC; Wp += blockIdx.z * strideWb + blockIdx.y * strideChunk;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int fx = lane & 15, fk = lane >> 4;
  const int col0 = blockIdx.x * VTC_COLS;
  const int row0 = blockIdx.y * VTC_ROWS + wave * 64;
  d4 acc[4];
#pragma unroll
  for (int s = 0; s < 4; s++) acc[s] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
  for (int kk = 0; kk < 16; kk++) {
    const int r = row0 + kk * 4 + fk;
    const bool rok = r < m;
    const double a = rok ? V[(long)r * ldv + fx] : 0.0;
    double b[4];
#pragma unroll
    for (int s = 0; s < 4; s++) {
      const int col = col0 + s * 16 + fx;
      b[s] = (rok && col < n) ? C[(long)r * ldc + col] : 0.0;
    }
#pragma unroll
    for (int s = 0; s < 4; s++) acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[s], acc[s], 0, 0, 0);
  }
#pragma unroll
  for (int s = 0; s < 4; s++)
#pragma unroll
    for (int r = 0; r < 4; r++) s_acc[wave][s][r][lane] = acc[s][r];
  __syncthreads();
  // thread t -> (strip s = wave, lane): sums the four waves' copies, fixed order
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const double v = ((s_acc[0][wave][r][lane] + s_acc[1][wave][r][lane]) + s_acc[2][wave][r][lane]) + s_acc[3][wave][r][lane];
    const int i = fk + 4 * r, col = col0 + wave * 16 + fx;
    if (col < n) Wp[(long)i * ldw + col] = v;
  }
}

// W2[:, j] = op(T) * sum_chunks Wp[chunk][:, j]; T upper triangular 16x16; trans -> T^T
__global__ __launch_bounds__(256) void qr_tw(const double* __restrict__ Tall, long strideT, int panel, int trans,
                                              const double* __restrict__ Wp, long ldw, long strideChunk, long strideWb, int nchunks,
                                              int n, double* __restrict__ W2, long strideW2) {
  __shared__ double s_T[NB][NB + 1];
  const double* T = Tall + blockIdx.y * strideT + (long)panel * NB * NB;
  Wp += blockIdx.y * strideWb; W2 += blockIdx.y * strideW2;
  const int t = threadIdx.x;
  s_T[t / NB][t % NB] = T[t];
  __syncthreads();
  const int col = blockIdx.x * 256 + t;
  if (col >= n) return;
  double w[NB];
#pragma unroll
  for (int i = 0; i < NB; i++) w[i] = 0.0;
  for (int ch = 0; ch < nchunks; ch++)
#pragma unroll
    for (int i = 0; i < NB; i++) w[i] += Wp[ch * strideChunk + (long)i * ldw + col];
#pragma unroll
  for (int i = 0; i < NB; i++) {
    double s = 0.0;
    if (trans) {
#pragma unroll
      for (int j = 0; j < NB; j++) s += s_T[j][i] * w[j];
    } else {
#pragma unroll
      for (int j = 0; j < NB; j++) s += s_T[i][j] * w[j];
    }
    W2[(long)i * ldw + col] = s;
  }
}

// ------------------------------------------------------------------------------------ sign fix
// One thread per matrix: decides the flips (reference convention, see file header).
// Tall input (M > N) follows the reference's other branch (qr.js:97-139): every rotation is normalised to
// c >= 0, which PRESERVES the sign of the pivot R_jj, so sign(R_jj) is whatever it was when row j finished its
// own eliminations = sign(det A[0:j+1,0:j+1] / det A[0:j,0:j]). With A[0:k,0:k] = Q[0:k,0:k] R[0:k,0:k] the
// flip of column j is the sign of the j-th pivot of Gaussian elimination WITHOUT pivoting on Q[0:N,0:N].
__global__ void qr_flips_tall(const double* __restrict__ LUq, int N, int* __restrict__ flips, int batch) {
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= (long)batch * N) return;
  const long b = i / N, j = i % N;
  flips[i] = LUq[b * N * N + j * N + j] < 0.0 ? 1 : 0;
}
__global__ __launch_bounds__(256) void qr_flips(const double* __restrict__ Rm, long ldr, long strideR, const double* __restrict__ taus, long strideTau,
                                                 int M, int N, int L, int* __restrict__ flips, int batch) {
  // one workgroup per matrix: flags in parallel, parity by a block-wide XOR
  const int b = blockIdx.x;
  const double* R = Rm + b * strideR; const double* tau = taus + b * strideTau; int* f = flips + (long)b * L;
  __shared__ int s_par[4];
  int parity = 0;
  for (int j = threadIdx.x; j < L; j += 256) {
    int fl = 0;
    if (tau[j] != 0.0) { parity ^= 1; if (R[(long)j * ldr + j] < 0.0) { fl = 1; parity ^= 1; } }
    f[j] = fl;
  }
  parity = __popcll(__ballot(parity)) & 1;
  if ((threadIdx.x & 63) == 0) s_par[threadIdx.x >> 6] = parity;
  __syncthreads();
  // plane rotations never change the determinant: det(Q) = +1 whenever Q is square (M <= N)
  if (threadIdx.x == 0 && M <= N && ((s_par[0] ^ s_par[1] ^ s_par[2] ^ s_par[3]) & 1)) f[L - 1] ^= 1;
}
// scale rows of R (rows x cols, ld) by -1 where flips[row]
__global__ void qr_flip_rows(double* __restrict__ X, long ld, long strideX, int rows, int cols, const int* __restrict__ flips, int L) {
  X += blockIdx.z * strideX; flips += (long)blockIdx.z * L;
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= cols) return;
  for (int r = blockIdx.y; r < rows; r += gridDim.y)
    if (flips[r]) X[(long)r * ld + col] = -X[(long)r * ld + col];
}
// scale columns of Q (rows x cols, ld) by -1 where flips[col]
__global__ void qr_flip_cols(double* __restrict__ X, long ld, long strideX, int rows, int cols, const int* __restrict__ flips, int L) {
  X += blockIdx.z * strideX; flips += (long)blockIdx.z * L;
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= cols || !flips[col]) return;
  for (int r = blockIdx.y; r < rows; r += gridDim.y) X[(long)r * ld + col] = -X[(long)r * ld + col];
}

// Full 16 x 16 triangular factor of a slot that was factorised in parts of bs columns: the diagonal blocks come from the
// side slot, the rest from T[1,2] = -T1 (V1^T V2) T2 applied level by level (groups of bs, 2 bs, ... columns); G = V^T V of
// the slot's 16 reflector columns (row-major 16 x 16). One small workgroup per matrix.
__global__ __launch_bounds__(256) void qr_t_assemble(double* __restrict__ Tall, const double* __restrict__ Tside, long strideT, int panel,
                                                      const double* __restrict__ Gm, long sG, int bs) {
  __shared__ double s_t[16][17], s_g[16][17], s_x[16][17];
  const long so = blockIdx.x * strideT + (long)panel * 256;
  const int t = threadIdx.x, i = t / 16, j = t % 16;
  s_t[i][j] = Tside[so + t];
  s_g[i][j] = Gm[blockIdx.x * sG + t];
  __syncthreads();
  for (int b = bs; b < 16; b *= 2) {
    // thread (i, j) with i in the first group of its pair and j in the second: x = (G12 T2)[i][j], then T12 = -T1 x
    const int pair0 = (i / (2 * b)) * 2 * b;
    const bool on = (i - pair0) < b && j >= pair0 + b && j < pair0 + 2 * b;
    double x = 0.0;
    if (on) for (int q = pair0 + b; q < pair0 + 2 * b; q++) x += s_g[i][q] * s_t[q][j];
    s_x[i][j] = x;
    __syncthreads();
    double y = 0.0;
    if (on) for (int q = pair0; q < pair0 + b; q++) y += s_t[i][q] * s_x[q][j];
    __syncthreads();
    if (on) s_t[i][j] = -y;
    __syncthreads();
  }
  Tall[so + t] = s_t[i][j];
}

#include "qr_batched_panel.h"

template <int R, int NWV = 8>
void launch_panel_row(nd4hip_handle* h, int batch, double* W, int M, long ld, long sW, double* V, long ldv, long sV,
                      double* T, long sT, double* taus, long sTau, int j0, int nb) {
  hipLaunchKernelGGL((qr_panel_row<R, NWV>), dim3(batch), dim3(64 * NWV), 0, h->stream, W, M, ld, sW, V, ldv, sV, T, sT, taus, sTau, j0, nb);
}
// The panel of m <= 2048 rows for `batch` matrices. One matrix is a latency problem: many waves with few rows each (R = 1, 2, 4 on
// 512 threads). A batch that fills the chip is a throughput problem: the same body on FEW waves with four rows per thread
// (128 threads for m <= 512, 256 for m <= 1024: 194 VGPRs, so four / two workgroups share a CU, their barriers cost next to nothing
// and a column step is ~530 instructions of ONE wave) — 2048 panels of 512 rows: 220 -> 151 us, 4096 of 256 rows: 315 -> 154 us
// (bench ops.qr_panel). All three shapes then run at the same 1.75 TB/s: the launch is bound by the instruction issue of the
// 4096 wave-panels (8500 instructions each, two waves per SIMD); forcing three waves per SIMD spills (196 us).
// Batches of full panels on the matrix cores (qr_batched_panel.h). ZERO: the rows below the top block are zeroed in W.
template <int R, int NWV, bool ZERO>
static void launch_qrb(nd4hip_handle* h, int batch, double* W, int M, long ld, long sW, double* V, long ldv, long sV,
                       double* T, long sT, double* taus, long sTau, int j0, long long* stamps) {
  hipLaunchKernelGGL((qrb_panel<R, NWV, ZERO>), dim3(batch), dim3(64 * NWV), 0, h->stream, W, M, ld, sW, V, ldv, sV, T, sT, taus, sTau, j0, stamps);
}
template <bool ZERO>
static void launch_panel_mfma(nd4hip_handle* h, int batch, double* W, int M, int m, long ld, long sW, double* V, long ldv, long sV,
                              double* T, long sT, double* taus, long sTau, int j0, long long* stamps = nullptr) {
  static const int small_min = [] { const char* e = getenv("ND4HIP_QR_SMALL_WG_BATCH"); return e ? atoi(e) : 64; }();   // 0: never
  const bool many = small_min > 0 && batch >= small_min;
  if (many && m <= 256)       launch_qrb<4, 1, ZERO>(h, batch, W, M, ld, sW, V, ldv, sV, T, sT, taus, sTau, j0, stamps);
  else if (many && m <= 512)  launch_qrb<4, 2, ZERO>(h, batch, W, M, ld, sW, V, ldv, sV, T, sT, taus, sTau, j0, stamps);
  else if (many && m <= 1024) launch_qrb<4, 4, ZERO>(h, batch, W, M, ld, sW, V, ldv, sV, T, sT, taus, sTau, j0, stamps);
  else if (m <= 512)          launch_qrb<1, 8, ZERO>(h, batch, W, M, ld, sW, V, ldv, sV, T, sT, taus, sTau, j0, stamps);
  else if (m <= 1024)         launch_qrb<2, 8, ZERO>(h, batch, W, M, ld, sW, V, ldv, sV, T, sT, taus, sTau, j0, stamps);
  else                        launch_qrb<4, 8, ZERO>(h, batch, W, M, ld, sW, V, ldv, sV, T, sT, taus, sTau, j0, stamps);
}
static void launch_panel_rows(nd4hip_handle* h, int batch, double* W, int M, int m, long ld, long sW, double* V, long ldv, long sV,
                              double* T, long sT, double* taus, long sTau, int j0, int nb) {
  static const int small_min = [] { const char* e = getenv("ND4HIP_QR_SMALL_WG_BATCH"); return e ? atoi(e) : 64; }();   // 0: never
  const bool many = small_min > 0 && batch >= small_min;
  if (many && m <= 256)       launch_panel_row<4, 1>(h, batch, W, M, ld, sW, V, ldv, sV, T, sT, taus, sTau, j0, nb);
  else if (many && m <= 512)  launch_panel_row<4, 2>(h, batch, W, M, ld, sW, V, ldv, sV, T, sT, taus, sTau, j0, nb);
  else if (many && m <= 1024) launch_panel_row<4, 4>(h, batch, W, M, ld, sW, V, ldv, sV, T, sT, taus, sTau, j0, nb);
  else if (m <= 512)          launch_panel_row<1>(h, batch, W, M, ld, sW, V, ldv, sV, T, sT, taus, sTau, j0, nb);
  else if (m <= 1024)         launch_panel_row<2>(h, batch, W, M, ld, sW, V, ldv, sV, T, sT, taus, sTau, j0, nb);
  else                        launch_panel_row<4>(h, batch, W, M, ld, sW, V, ldv, sV, T, sT, taus, sTau, j0, nb);
}
// ---- exact power-of-two normalisation: the reference's Givens kernel (_giv_rot.js:22-37) scales by max(|a|,|b|)
// and never overflows; Householder squares the entries. Per matrix: e = 0 if 2^-400 <= max|a| <= 2^400, else the
// exponent that brings max|a| to [1,2); the work copy is multiplied by 2^-e and R by 2^e (both exact).
__global__ __launch_bounds__(256) void qr_amax(const double* __restrict__ Am, long n_per, unsigned long long* __restrict__ amax_bits) {
  const double* A = Am + blockIdx.y * n_per;
  double mx = 0.0;
  for (long i = blockIdx.x * 256l + threadIdx.x; i < n_per; i += (long)gridDim.x * 256) { const double v = fabs(A[i]); mx = (v > mx) ? v : mx; }   // NaN ignored
  mx = nd4dpp::wave_max(mx);
  // the bit pattern of a non-negative double is monotone: an integer max is order independent (deterministic)
  if ((threadIdx.x & 63) == 0 && mx > 0.0) atomicMax(amax_bits + blockIdx.y, (unsigned long long)__double_as_longlong(mx));
}
__device__ __forceinline__ int qr_scale_exponent(unsigned long long bits) {
  const double m = __longlong_as_double((long long)bits);
  int e = 0;
  if (m > 0.0 && m < DBL_MAX * 2.0) { int ex; (void)frexp(m, &ex); if (ex > 400 || ex < -400) e = ex - 1; }
  return e;
}
__global__ __launch_bounds__(256) void qr_scale_apply(const double* __restrict__ src, double* __restrict__ dst, long n_per,
                                                       const unsigned long long* __restrict__ amax_bits, int sign) {
  const int e = sign * qr_scale_exponent(amax_bits[blockIdx.y]);
  const long i = blockIdx.x * 256l + threadIdx.x;
  if (i >= n_per) return;
  const double v = src[blockIdx.y * n_per + i];
  if (e != 0 || src != dst) dst[blockIdx.y * n_per + i] = (e == 0) ? v : ldexp(v, e);
}

template <int R, bool REG>
void launch_panel(nd4hip_handle* h, int batch, double* W, int M, long ld, long sW, double* V, long ldv, long sV,
                  double* T, long sT, double* taus, long sTau, int j0, int nb) {
  hipLaunchKernelGGL((qr_panel<R, REG>), dim3(batch), dim3(1024), 0, h->stream, W, M, ld, sW, V, ldv, sV, T, sT, taus, sTau, j0, nb);
}

struct QrWs {
  double *V, *T, *Tside, *taus, *Wp, *W2, *work; int* flips;
  long ldv, sV, sT, sTau, ldw, sChunk, sWb, sW2;
  int nchunks_max;
};

// C (rows [j0,M) x n columns starting at C0) <- (I - V op(T) V^T) C, V = Vall[j0:, j0:j0+nb]
int apply_block_reflector(nd4hip_handle* h, const QrWs& ws, int batch, int M, int j0, int panel, int trans,
                          double* C0, long ldc, long strideC, int n) {
  if (n <= 0) return 0;
  const int m = M - j0;
  const int nchunks = (m + VTC_ROWS - 1) / VTC_ROWS;
  const double* Vp = ws.V + (long)j0 * ws.ldv + j0;
  hipLaunchKernelGGL(qr_vtc, dim3((unsigned)((n + VTC_COLS - 1) / VTC_COLS), (unsigned)nchunks, (unsigned)batch), dim3(256), 0, h->stream,
                     Vp, ws.ldv, ws.sV, C0, ldc, strideC, m, n, ws.Wp, ws.ldw, ws.sChunk, ws.sWb);
  hipLaunchKernelGGL(qr_tw, dim3((unsigned)((n + 255) / 256), (unsigned)batch), dim3(256), 0, h->stream,
                     ws.T, ws.sT, panel, trans, ws.Wp, ws.ldw, ws.sChunk, ws.sWb, nchunks, n, ws.W2, ws.sW2);
  ND4_HIP(hipGetLastError());
  return nd4_gemm(h, false, false, m, n, NB, -1.0, Vp, ws.ldv, ws.sV, ws.W2, ws.ldw, ws.sW2, 1.0, C0, ldc, strideC, batch);
}

// diagonal bs x bs blocks of the n x n compact-WY factor <- the given blocks (everything else zero); bs = 1: the taus
__global__ void wy_t_diag(const double* __restrict__ Tp, double* __restrict__ Tall, int n, int bs) {
  const int blk = blockIdx.x, t = threadIdx.x;                      // bs * bs threads
  Tall[(long)(blk * bs + t / bs) * n + blk * bs + t % bs] = Tp[(long)blk * bs * bs + t];
}

int form_q_compact_wy(nd4hip_handle* h, const QrWs& ws, int batch, int M, int Lq, int npanels, double* Q, long sQ) {
  const int n = npanels * NB;                                       // == ws.ldv
  for (int m = 0; m < batch; m++)
    ND4_TRY(nd4_wy_form(h, M, n, ws.V + (long)m * ws.sV, ws.T + (long)m * ws.sT, NB, Q + (long)m * sQ, Lq));
  return 0;
}


// host side of the multi-workgroup panels: one panel = three launches (+ two conditional ones for panels taller than the register
// kernel), the previous reflector's side work riding along on the column blocks of W up to near_end (then, optionally, Q^T)
struct QrhHost {
  nd4hip_handle* h; QrhP P; int batch; int nq;          // nq: column blocks of Q^T (0: no Q^T accumulation)
  double *V, *T; long ldv, sV, sT;                        // for qr_narrow_apply
  int pj0 = -1;                                           // the reflector whose narrow / side work is pending (-1: none)
  bool fused_last = false;                                // the last panel went through qrh_bc
  void add_seg(int kind, int first, int count) { if (count > 0) { P.seg[P.nseg].kind = kind; P.seg[P.nseg].first = first; P.seg[P.nseg].count = count; P.nseg++; } }
  int seg_total() const { int n = 0; for (int i = 0; i < P.nseg; i++) n += P.seg[i].count; return n; }
  void side_launch() { if (P.nseg > 0) hipLaunchKernelGGL(qrh_side_only, dim3((unsigned)seg_total(), (unsigned)batch), dim3(512), 0, h->stream, P); P.nseg = 0; }
  // the side work of the reflector at pj: workgroups for W (columns [pj + 32, near_end)), for W and Q^T
  void side_of(int pj, int near_end, bool with_qt, int& nw_e, int& all_e) {
    nw_e = 0; all_e = 0;
    if (pj < 0) return;
    const int wide0 = pj + 2 * NB;
    P.wide0 = wide0; P.nrc = (P.M - pj + 511) / 512;
    P.nnw = wide0 < near_end ? (near_end - wide0 + NB - 1) / NB : 0;
    P.nqb = with_qt ? nq : 0;
    nw_e = ((P.nnw + 1) / 2) * P.nrc; all_e = nw_e + ((P.nqb + 1) / 2) * P.nrc;      // a workgroup takes two adjacent column blocks
  }
  int panel(int j0, int near_end, bool with_qt, bool skip_x) {
    const int m = P.M - j0;
    int side_w = 0, side_all = 0;
    side_of(pj0, near_end, with_qt, side_w, side_all);
    P.j0 = j0; P.pj0 = pj0; P.skip_x = skip_x ? 1 : 0; P.na_shift = 0;
    const int nA = (m + NB + 511) / 512, nB = (m + 511) / 512;
    // phases B and C in one launch (qrh_bc): the row workgroups keep phase A's partition and exchange the Gram matrices of Q1 inside
    // the kernel; they also take the next panel's columns through the previous reflector, so the side work starts one block later.
    // The side work itself: both phases of a (block pair, row chunk) in one workgroup (qrh_side_fused): W pairs ride in launch A,
    // Q^T pairs in launch B+C; only the next panel's own block still needs its partial X ahead of launch B+C (SEG_NX0).
    static const bool bc_off = [] { const char* e = getenv("ND4HIP_QR_NO_FUSED_BC"); return e && *e && *e != '0'; }();
    static const bool sf_off = [] { const char* e = getenv("ND4HIP_QR_NO_FUSED_SIDE"); return e && *e && *e != '0'; }();
    const bool next_cols = !skip_x && j0 + NB < P.N;
    if (!bc_off && P.Xch != nullptr && !(pj0 >= 0 && next_cols && P.nnw < 1)) {
      const int sh = (pj0 >= 0 && next_cols) ? 1 : 0;
      const int nwp = pj0 >= 0 ? (P.nnw - sh + 1) / 2 : 0, nqp = pj0 >= 0 ? (P.nqb + 1) / 2 : 0;
      const bool sf = !sf_off && P.Xsx != nullptr;
      static const bool mg_off = [] { const char* e = getenv("ND4HIP_QR_NO_MERGED"); return e && *e && *e != '0'; }();
      if ((sf || side_all == 0) && !mg_off) {
        // ONE launch per panel: phase A rides in front of B and C (first exchange: Gram partials + partial X of the next block)
        P.na_shift = sh;
        P.nrow = nA; P.ngp = 0; P.nseg = 0; add_seg(SEG_F, 0, (nwp + nqp) * P.nrc);
        const dim3 gm((unsigned)(nA + seg_total()), (unsigned)batch);
        if (m <= 512)       hipLaunchKernelGGL((qrh_bc<1, true>), gm, dim3(512), 0, h->stream, P);
        else if (m <= 1024) hipLaunchKernelGGL((qrh_bc<2, true>), gm, dim3(512), 0, h->stream, P);
        else if (m <= 2048) hipLaunchKernelGGL((qrh_bc<4, true>), gm, dim3(512), 0, h->stream, P);
        else {
          hipLaunchKernelGGL((qrh_bc<0, true>), gm, dim3(512), 0, h->stream, P);
          hipLaunchKernelGGL(qr_panel_flagged, dim3((unsigned)batch), dim3(1024), 0, h->stream, P.Wm, P.M, P.ld, P.strideW, P.Vall, P.ldv, P.strideV,
                             P.Tall, P.strideT, P.taus, P.strideTau, j0, NB, P.flag);
        }
        P.na_shift = 0;
        pj0 = j0; P.nxp = nA; P.nseg = 0; P.stamp_slot++; fused_last = true;
        ND4_HIP(hipGetLastError());
        return 0;
      }
      P.nrow = nA; P.ngp = 0; P.nseg = 0;
      static const int sf_a = [] { const char* e = getenv("ND4HIP_QR_SIDE_IN_A"); return e ? atoi(e) : 0; }();   // percent of the W pairs that ride in launch A
      const int nwa = nwp * sf_a / 100;
      if (sf) { P.na_shift = sh; if (sh) add_seg(SEG_NX0, 0, P.nrc); add_seg(SEG_F, 0, nwa * P.nrc); }
      else add_seg(SEG_NX, 0, side_all);
      hipLaunchKernelGGL(qrh_gram, dim3((unsigned)(nA + seg_total()), (unsigned)batch), dim3(512), 0, h->stream, P); P.stamp_slot++;
      P.na_shift = sh;
      P.nrow = nA; P.ngp = nA; P.nseg = 0;
      if (sf) add_seg(SEG_F, nwa * P.nrc, (nwp - nwa + nqp) * P.nrc); else add_seg(SEG_NA, 0, (nwp + nqp) * P.nrc);
      const dim3 gc((unsigned)(nA + seg_total()), (unsigned)batch);
      if (m <= 512)       hipLaunchKernelGGL((qrh_bc<1, false>), gc, dim3(512), 0, h->stream, P);
      else if (m <= 1024) hipLaunchKernelGGL((qrh_bc<2, false>), gc, dim3(512), 0, h->stream, P);
      else if (m <= 2048) hipLaunchKernelGGL((qrh_bc<4, false>), gc, dim3(512), 0, h->stream, P);
      else {
        hipLaunchKernelGGL((qrh_bc<0, false>), gc, dim3(512), 0, h->stream, P);
        hipLaunchKernelGGL(qr_panel_flagged, dim3((unsigned)batch), dim3(1024), 0, h->stream, P.Wm, P.M, P.ld, P.strideW, P.Vall, P.ldv, P.strideV,
                           P.Tall, P.strideT, P.taus, P.strideTau, j0, NB, P.flag);
      }
      P.na_shift = 0;
      pj0 = j0; P.nxp = nA; P.nseg = 0; P.stamp_slot++; fused_last = true;
      ND4_HIP(hipGetLastError());
      return 0;
    }
    P.nrow = nA; P.ngp = 0; P.nseg = 0; add_seg(SEG_NX, 0, side_all);
    hipLaunchKernelGGL(qrh_gram, dim3((unsigned)(nA + seg_total()), (unsigned)batch), dim3(512), 0, h->stream, P); P.stamp_slot++;
    fused_last = false;
    P.nrow = nB; P.ngp = nA; P.nseg = 0; add_seg(SEG_NA, 0, side_w);
    hipLaunchKernelGGL(qrh_chol, dim3((unsigned)(nB + seg_total()), (unsigned)batch), dim3(512), 0, h->stream, P); P.stamp_slot++;
    P.ngp = nB; P.nseg = 0; add_seg(SEG_NA, side_w, side_all - side_w);
    const dim3 gc((unsigned)(nB + seg_total()), (unsigned)batch);
    if (m <= 512)       hipLaunchKernelGGL(qrh_reconstruct<1>, gc, dim3(512), 0, h->stream, P);
    else if (m <= 1024) hipLaunchKernelGGL(qrh_reconstruct<2>, gc, dim3(512), 0, h->stream, P);
    else if (m <= 2048) hipLaunchKernelGGL(qrh_reconstruct<4>, gc, dim3(512), 0, h->stream, P);
    else {
      // taller than the register-resident panel: a flagged panel is factorised by the global-memory panel kernel in a launch of its own
      hipLaunchKernelGGL(qrh_reconstruct<0>, gc, dim3(512), 0, h->stream, P);
      hipLaunchKernelGGL(qr_panel_flagged, dim3((unsigned)batch), dim3(1024), 0, h->stream, P.Wm, P.M, P.ld, P.strideW, P.Vall, P.ldv, P.strideV,
                         P.Tall, P.strideT, P.taus, P.strideTau, j0, NB, P.flag);
      if (!skip_x) hipLaunchKernelGGL(qrh_x_flagged, dim3((unsigned)nB, (unsigned)batch), dim3(512), 0, h->stream, P);
    }
    pj0 = j0; P.nxp = nB; P.nseg = 0; P.stamp_slot++;
    ND4_HIP(hipGetLastError());
    return 0;
  }
  // the last reflector: the 16 columns behind its panel (unless the caller's block update covers them), then its side work on its own
  int finish(int near_end, bool with_qt, bool narrow) {
    if (pj0 < 0) return 0;
    if (narrow && pj0 + NB < P.N) {
      const int m = P.M - pj0;
      if (fused_last) {   // a flagged panel of the fused launch left no partials of X behind
        P.j0 = pj0; P.na_shift = 0;
        hipLaunchKernelGGL(qrh_x_flagged, dim3((unsigned)P.nxp, (unsigned)batch), dim3(512), 0, h->stream, P);
      }
      hipLaunchKernelGGL(qr_narrow_apply, dim3((unsigned)((m + 255) / 256), (unsigned)batch), dim3(256), 0, h->stream,
                         P.Wm, P.M, P.N, P.ld, P.strideW, V, ldv, sV, T, sT, pj0, pj0 + NB, P.Xp, P.strideXp, P.nxp);
    }
    int side_w = 0, side_all = 0;
    side_of(pj0, near_end, with_qt, side_w, side_all);
    P.pj0 = pj0;
    P.nseg = 0; add_seg(SEG_NX, 0, side_all); side_launch();
    add_seg(SEG_NA, 0, side_all); side_launch();
    pj0 = -1;
    ND4_HIP(hipGetLastError());
    return 0;
  }
  int dump_stamps() {                                      // debug: per launch, the stamps of workgroup 0 relative to the first one, in us
    if (!P.stamps) return 0;
    std::vector<long long> st((size_t)8 * P.stamp_slot);
    ND4_HIP(hipStreamSynchronize(h->stream));
    ND4_HIP(hipMemcpy(st.data(), P.stamps, sizeof(long long) * st.size(), hipMemcpyDeviceToHost));
    (void)hipFree(P.stamps); P.stamps = nullptr;
    for (int i = 0; i < P.stamp_slot; i++) {
      fprintf(stderr, "qrh stamp launch %d:", i);
      for (int k = 0; k < 8; k++) fprintf(stderr, " %.2f", st[i * 8 + k] ? (st[i * 8 + k] - st[0]) * 0.01 : 0.0);
      fprintf(stderr, "\n");
    }
    return 0;
  }
};

}  // namespace

// Compact-WY factor of up to 64 reflector columns = up to four panels of a batch member: T (n x n) from the panels' 16 x 16 factors
// on its diagonal and G = V^T V, T12 = -T1 (V1^T V2) T2 level by level (16, then 32 columns; valid for the FULL factors of the
// matrix-core panels as well as for triangular ones). One workgroup per matrix, everything in LDS: each G12 block is used once, so
// G12 T2 overwrites it. Batches beyond the look-ahead form (qr_batch: the rank-16 updates were 56 % of the run).
__global__ __launch_bounds__(256) void wy_t_small(const double* __restrict__ Tdiag, long sTd, const double* __restrict__ G, int ldg, long sG,
                                                   double* __restrict__ Tout, int ldt, long sTo, int n) {
  __shared__ double s_t[64 * 64], s_g[64 * 64];
  const long m = blockIdx.x;
  const int t = threadIdx.x;
  for (int e = t; e < 4096; e += 256) {
    const int i = e >> 6, j = e & 63;
    const bool in = i < n && j < n;
    s_g[e] = in ? G[m * sG + (long)i * ldg + j] : 0.0;
    s_t[e] = (in && (i >> 4) == (j >> 4)) ? Tdiag[m * sTd + (long)(i >> 4) * 256 + (i & 15) * 16 + (j & 15)] : 0.0;
  }
  __syncthreads();
  for (int b = 16; b < n; b <<= 1) {
    double v[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {                                    // X = G12 T2 on the 12-blocks of this level
      const int e = t + 256 * k, i = e >> 6, j = e & 63, p0 = i & ~(2 * b - 1);
      const int qe = (p0 + 2 * b < n) ? p0 + 2 * b : n;
      double x = 0.0;
      if (i - p0 < b && j >= p0 + b && j < qe) for (int q = p0 + b; q < qe; q++) x += s_g[i * 64 + q] * s_t[q * 64 + j];
      v[k] = x;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const int e = t + 256 * k, i = e >> 6, j = e & 63, p0 = i & ~(2 * b - 1);
      const int qe = (p0 + 2 * b < n) ? p0 + 2 * b : n;
      if (i - p0 < b && j >= p0 + b && j < qe) s_g[e] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; k++) {                                    // T12 = -T1 X
      const int e = t + 256 * k, i = e >> 6, j = e & 63, p0 = i & ~(2 * b - 1);
      const int qe = (p0 + 2 * b < n) ? p0 + 2 * b : n;
      double y = 0.0;
      if (i - p0 < b && j >= p0 + b && j < qe) for (int q = p0; q < p0 + b; q++) y += s_t[i * 64 + q] * s_g[q * 64 + j];
      v[k] = y;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const int e = t + 256 * k, i = e >> 6, j = e & 63, p0 = i & ~(2 * b - 1);
      const int qe = (p0 + 2 * b < n) ? p0 + 2 * b : n;
      if (i - p0 < b && j >= p0 + b && j < qe) s_t[e] = -v[k];
    }
    __syncthreads();
  }
  for (int e = t; e < 4096; e += 256) {
    const int i = e >> 6, j = e & 63;
    if (i < n && j < n) Tout[m * sTo + (long)i * ldt + j] = s_t[e];
  }
}

static int wy_build_T(nd4hip_handle* h, int M, int n, const double* V, long ldv, const double* Tdiag, int bs, double* Tall, double* G, double* tmp);

int nd4_geqrf_q(nd4hip_handle* h, int64_t batch, int64_t M, int64_t N, const double* A, double* Q, double* R) {
  return nd4_geqrf_q_ex(h, batch, M, N, A, Q, R, false);
}

// full = false: qr_decomp (qr.js:80-145): Q [M, L], R [L, N], tall input with the c >= 0 convention of :97-139.
// full = true : qr_decomp_full (qr.js:27-77): Q [M, M], R [M, N] with the Givens-full convention for every shape
//               (R_jj >= 0 wherever something was eliminated; det Q = +1 fixes the last row when M <= N). For M > N the
//               trailing M-N columns of Q are an orthonormal completion (not unique; the reference's is the one its
//               rotation order happens to produce) and rows N.. of R are zero.
// Tall-skinny input (M > 2048 rows, M >= 4 N): TSQR. The rows are cut into blocks of <= 2048 (what the register-resident
// panel kernel holds), all blocks are factorised in ONE batched call, the stacked R factors (nblk N x N) are factorised
// again (recursively), and Q = blockwise Q1_b Q2_b is one strided-batched GEMM. A QR factorisation is unique up to the signs
// of R's rows, so the reference's convention is imposed on the final pair by nd4_givens_signs, whatever the levels did.
// Without it a 65536 x 32 panel would run on one workgroup through the global-memory fallback (31 ms instead of < 1 ms).
static int geqrf_tsqr(nd4hip_handle* h, int batch, int M, int N, const double* A, double* Q, double* R) {
  const int nblk = (M + 2047) / 2048;
  int mb = (M + nblk - 1) / nblk;
  mb = (mb + 1) & ~1;
  const long Mp = (long)nblk * mb;
  Nd4WsScope scope(h);
  void* p = nullptr;
  const size_t nAp = (size_t)batch * Mp * N, nR1 = (size_t)batch * nblk * N * N;
  ND4_TRY(nd4_ws_alloc(h, sizeof(double) * (2 * nAp + 2 * nR1 + (size_t)batch * Mp * N) + sizeof(int) * ((size_t)batch * N + 2) + 64, &p));
  double* Ap = static_cast<double*>(p);
  double* Q1 = Ap + nAp;
  double* R1 = Q1 + nAp;                                               // per matrix: the nblk R factors stacked = (nblk N) x N
  double* Q2 = R1 + nR1;
  double* Qp = Q2 + nR1;
  int* flips = reinterpret_cast<int*>(Qp + (size_t)batch * Mp * N);
  if (Mp != M) ND4_HIP(hipMemsetAsync(Ap, 0, sizeof(double) * nAp, h->stream));
  ND4_TRY(nd4_copy_matrix(h, M, N, A, N, Ap, N, batch, (long)M * N, Mp * N));
  ND4_TRY(nd4_geqrf_q_ex(h, (int64_t)batch * nblk, mb, N, Ap, Q1, R1, false));
  ND4_TRY(nd4_geqrf_q_ex(h, batch, (int64_t)nblk * N, N, R1, Q2, R, false));
  double* Qdst = (Mp == M) ? Q : Qp;
  ND4_TRY(nd4_gemm(h, false, false, mb, N, N, 1.0, Q1, N, (long)mb * N, Q2, N, (long)N * N, 0.0, Qdst, N, (long)mb * N, (int64_t)batch * nblk));
  if (Mp != M) ND4_TRY(nd4_copy_matrix(h, M, N, Qp, N, Q, N, batch, Mp * N, (long)M * N));
  return nd4_givens_signs(h, batch, M, N, N, true, Q, N, (long)M * N, R, N, (long)N * N, nullptr, 0, flips);
}

int nd4_geqrf_q_ex(nd4hip_handle* h, int64_t batch64, int64_t M64, int64_t N64, const double* A, double* Q, double* R, bool full) {
  ND4_CHECK_ARG(M64 < (1ll << 30) && N64 < (1ll << 30) && batch64 < 65536, "nd4_geqrf_q: extent out of range");
  const int M = (int)M64, N = (int)N64, batch = (int)batch64;
  {
    static const bool tsqr_off = [] { const char* e = getenv("ND4HIP_QR_NO_TSQR"); return e && *e && *e != '0'; }();
    const long nblk = (M + 2047) / 2048;
    // every block needs >= N rows, and the stacked R (nblk N rows) must either fit the fast panel kernel directly or be
    // at most half as tall as the input (so that the recursion terminates quickly)
    if (!full && !tsqr_off && M > 2048 && N <= 2048 && (long)batch * nblk <= 32768 && M / nblk >= N &&
        (nblk * N <= 2048 || 2 * nblk * N <= (long)M))
      return geqrf_tsqr(h, batch, M, N, A, Q, R);
  }
  const int L = M < N ? M : N;
  const int npanels = (L + NB - 1) / NB;
  const bool tall = M > N;
  const int Lq = (full && tall) ? M : L;                    // columns of Q
  const int Lr = (full && tall) ? M : L;                    // rows of R

  // ---- workspace carve-up (all per-matrix blocks are multiples of 2 doubles -> 16-B aligned) ----
  QrWs ws;
  ws.ldv = ((L + NB - 1) / NB) * NB;                        // Vall: M x ldv, zero above each panel
  ws.sV = (long)M * ws.ldv;
  ws.sT = (long)npanels * NB * NB;
  ws.sTau = ws.ldv;
  const int ncols = (N > Lq ? N : Lq);
  ws.ldw = ((ncols + 1) / 2) * 2;
  ws.nchunks_max = (M + VTC_ROWS - 1) / VTC_ROWS;
  ws.sChunk = (long)NB * ws.ldw;
  ws.sWb = ws.sChunk * ws.nchunks_max;
  ws.sW2 = ws.sChunk;
  const long sWork = tall ? (long)M * N : 0;
  static const bool la_off = [] { const char* e = getenv("ND4HIP_QR_NO_LOOKAHEAD"); return e && *e && *e != '0'; }();
  static const bool wy_off = [] { const char* e = getenv("ND4HIP_QR_NO_WY"); return e && *e && *e != '0'; }();
  static const bool qt_off = [] { const char* e = getenv("ND4HIP_QR_NO_QT"); return e && *e && *e != '0'; }();
  // (like LU: the look-ahead launches serve few matrices; a batch that fills the chip takes the plain sequence with the
  //  matrix-core panels of qr_batched_panel.h: 1024 x 512^2 48.6 -> 43.7 ms; 8 matrices 1.97 against 2.53 ms, 32: 2.91 against 2.87.
  //  ND4HIP_QR_LA_MAX_BATCH moves the switch.)
  static const int la_max_batch = [] { const char* e = getenv("ND4HIP_QR_LA_MAX_BATCH"); return e ? atoi(e) : 24; }();
  const bool lookahead = !la_off && M <= 2048 && M >= 64 && batch <= la_max_batch;
  const bool use_qt = lookahead && !wy_off && !qt_off && batch <= 4 && L >= 256;   // Q^T accumulated in the shadow of the panels
  const long sQT = use_qt ? (long)M * M : 0;
  static const bool hr_off = [] { const char* e = getenv("ND4HIP_QR_NO_HR"); return e && *e && *e != '0'; }();
  // panels taller than the register kernel: two-level driver below (a ragged last panel must fit the thread-per-row kernel)
  const bool hr_tall = !hr_off && !la_off && batch <= 8 && M > 2048 && M <= 16384 && L >= 256 && (L % NB == 0 || M - (L / NB) * NB <= 2048);
  const bool use_hr = (lookahead && !hr_off && batch <= 8) || hr_tall;    // multi-workgroup panels (CholeskyQR2 + compact orthogonal completion)
  const int hr_parts = (M + NB + 511) / 512 + 1;
  const long sGp = use_hr ? (long)(hr_parts + 1) * 256 : 0, sG2 = use_hr ? (long)hr_parts * QX_ROW_SLOT : 0, sR1 = use_hr ? 256 : 0;   // G2: partials (3 launches) or 512 tagged words per row workgroup (qrh_bc)
  const long hr_rcs = (M + 511) / 512;
  const long sXs = use_hr ? ((N + NB - 1) / NB + (use_qt ? (M + NB - 1) / NB : 0)) * hr_rcs * 256 : 0;   // side work: [column block][row chunk][256]
  const long sXsx = 2 * sXs;                                 // fused side work: 512 tagged words per (column block, row chunk)
  size_t doubles = (size_t)batch * (ws.sV + 2 * ws.sT + ws.sTau + ws.sWb + ws.sW2 + sWork + sQT + sGp + sG2 + sR1 + sXs + sXsx);
  size_t bytes = doubles * sizeof(double) + ((size_t)batch * L + 2) * sizeof(int) + (size_t)batch * 8 + (size_t)batch * sizeof(int) + 64;
  void* p = nullptr;
  Nd4WsScope scope(h);
  ND4_TRY(nd4_ws_alloc(h, bytes, &p));
  double* d = static_cast<double*>(p);
  ws.V = d; d += (size_t)batch * ws.sV;
  ws.T = d; d += (size_t)batch * ws.sT;
  ws.Tside = d; d += (size_t)batch * ws.sT;                 // diagonal blocks of slots factorised in parts (tall panels)
  ws.taus = d; d += (size_t)batch * ws.sTau;
  ws.Wp = d; d += (size_t)batch * ws.sWb;
  ws.W2 = d; d += (size_t)batch * ws.sW2;
  ws.work = tall ? d : nullptr; d += (size_t)batch * sWork;
  double* QT = use_qt ? d : nullptr; d += (size_t)batch * sQT;
  double* hrGp = d; d += (size_t)batch * sGp;
  double* hrG2 = d; d += (size_t)batch * sG2;
  double* hrR1 = d; d += (size_t)batch * sR1;
  double* hrXs = d; d += (size_t)batch * sXs;
  double* hrXsx = d; d += (size_t)batch * sXsx;
  ws.flips = reinterpret_cast<int*>(d);

  // working matrix: R's buffer when it has A's shape (M <= N), a workspace copy when tall
  double* W = tall ? ws.work : R;
  const long ld = N, sW = (long)M * N;
  unsigned long long* exps = reinterpret_cast<unsigned long long*>(ws.flips + (((size_t)batch * L + 1) & ~size_t(1)));   // max|a| bits per matrix
  int* hrFlag = reinterpret_cast<int*>(exps + batch);
  ND4_HIP(hipMemsetAsync(exps, 0, sizeof(unsigned long long) * batch, h->stream));
  {
    long nblk = (sW + 256 * 16 - 1) / (256 * 16); if (nblk > 2048) nblk = 2048;
    hipLaunchKernelGGL(qr_amax, dim3((unsigned)nblk, (unsigned)batch), dim3(256), 0, h->stream, A, sW, exps);
  }
  hipLaunchKernelGGL(qr_scale_apply, dim3((unsigned)((sW + 255) / 256), (unsigned)batch), dim3(256), 0, h->stream, A, W, sW, exps, -1);
  ND4_HIP(hipMemsetAsync(ws.V, 0, sizeof(double) * (size_t)batch * ws.sV, h->stream));
  if (M > 2048) ND4_HIP(hipMemsetAsync(ws.Tside, 0, sizeof(double) * (size_t)batch * ws.sT, h->stream));

  // ---- factorisation: panels left to right ----
  // Look-ahead form (every panel fits the thread-per-row kernel): panel p shares its launch with the update of the columns behind
  // it by reflector p-1; only the 16 columns of panel p+1 are updated between two panels. See qr_colblock_update.
  static const int hr_min_rows = [] { const char* e = getenv("ND4HIP_QR_HR_MIN_ROWS"); const int v = e ? atoi(e) : 0; return v >= HR_MIN_ROWS ? v : HR_MIN_ROWS; }();
  QrhHost hr;
  hr.h = h; hr.batch = batch; hr.nq = QT ? (M + NB - 1) / NB : 0; hr.V = ws.V; hr.T = ws.T; hr.ldv = ws.ldv; hr.sV = ws.sV; hr.sT = ws.sT;
  if (use_hr) {
    QrhP& P = hr.P;
    P.Wm = W; P.M = M; P.N = N; P.ld = ld; P.strideW = sW; P.Vall = ws.V; P.ldv = ws.ldv; P.strideV = ws.sV; P.Tall = ws.T; P.strideT = ws.sT;
    P.taus = ws.taus; P.strideTau = ws.sTau; P.Xp = ws.Wp; P.strideXp = ws.sWb; P.Gp = hrGp; P.strideGp = sGp; P.G2p = hrG2; P.strideG2 = sG2;
    P.R1 = hrR1; P.flag = hrFlag; P.QT = QT; P.strideQT = sQT; P.nxp = 0; P.Xs = hrXs; P.strideXs = sXs;
    P.Xch = reinterpret_cast<unsigned long long*>(hrG2); P.strideXch = sG2; P.na_shift = 0;
    P.Xsx = reinterpret_cast<unsigned long long*>(hrXsx); P.strideXsx = sXsx; P.rcs_max = (int)hr_rcs;
    ND4_HIP(hipMemsetAsync(hrXsx, 0, sizeof(double) * (size_t)batch * sXsx, h->stream));
    ND4_HIP(hipMemsetAsync(hrG2, 0, sizeof(double) * (size_t)batch * sG2, h->stream));
    ND4_HIP(hipMemsetAsync(hrFlag, 0, sizeof(int) * (size_t)batch, h->stream));
    P.nseg = 0; P.wide0 = 0; P.nrc = 1; P.nnw = 0; P.nqb = 0; P.skip_x = 0; P.j0 = 0; P.pj0 = -1; P.nrow = 0; P.ngp = 0;
    static const bool want_stamps = [] { const char* e = getenv("ND4HIP_QR_STAMPS"); return e && *e && *e != '0'; }();
    P.stamps = nullptr; P.stamp_slot = 0; P.status = h->xstat; { const int dp = nd4_test_drop_panel(); P.drop_tag = dp >= 0 ? dp + 1 : -1; }
    if (want_stamps) { ND4_HIP(hipMalloc(&P.stamps, sizeof(long long) * 8 * 3 * (npanels + 1))); ND4_HIP(hipMemset(P.stamps, 0, sizeof(long long) * 8 * 3 * (npanels + 1))); }
  }
  double *btT = nullptr, *btG = nullptr, *btX = nullptr, *btW = nullptr;              // batched two-level form (generic branch below)
  int bt_ppb = 0; long bt_sT = 0, bt_sX = 0;
  if (lookahead) {
    int pj0 = -1;                                            // first row/column of the previous panel
    const int nq = QT ? (M + NB - 1) / NB : 0;
    if (QT) ND4_TRY(nd4_set_identity(h, M, M, QT, M, batch, sQT));
    int pnl = 0;
    if (use_hr) {
      // multi-workgroup panels while they are full and tall enough: three launches per panel (Gram / Cholesky / representation). The
      // previous reflector's side work rides along: partial X of every column block (trailing columns of W, then Q^T) in launch A,
      // the update of the blocks of W in launch B (phase C reads the first of them), of Q^T in launch C.
      for (; pnl < npanels; pnl++) {
        const int j0 = pnl * NB, nb = L - j0 < NB ? L - j0 : NB, m = M - j0;
        if (nb < NB || m < hr_min_rows) break;
        ND4_TRY(hr.panel(j0, N, QT != nullptr, false));
      }
      ND4_TRY(hr.finish(N, QT != nullptr, true));             // nothing pending afterwards: the remaining panels start afresh
      ND4_TRY(hr.dump_stamps());
    }
    for (; pnl < npanels; pnl++) {
      const int j0 = pnl * NB, nb = L - j0 < NB ? L - j0 : NB, m = M - j0;
      // reflector p-1 has reached the 16 columns behind its own panel (the narrow launch: [pj0 + NB, pj0 + 2 NB), which contain this
      // panel); the columns from there on still lack it
      const int wide0 = pj0 + 2 * NB, nwide = (pj0 >= 0 && wide0 < N) ? (N - wide0 + NB - 1) / NB : 0;
      const int wc0 = j0 + nb;
      const dim3 grid((unsigned)(1 + nwide + (pj0 >= 0 ? nq : 0)), (unsigned)batch);
      if (m <= 512)       hipLaunchKernelGGL(qr_panel_row_la<1>, grid, dim3(512), 0, h->stream, W, M, N, ld, sW, ws.V, ws.ldv, ws.sV, ws.T, ws.sT, ws.taus, ws.sTau, j0, nb, pj0 < 0 ? 0 : pj0, wide0, nwide, QT, sQT);
      else if (m <= 1024) hipLaunchKernelGGL(qr_panel_row_la<2>, grid, dim3(512), 0, h->stream, W, M, N, ld, sW, ws.V, ws.ldv, ws.sV, ws.T, ws.sT, ws.taus, ws.sTau, j0, nb, pj0 < 0 ? 0 : pj0, wide0, nwide, QT, sQT);
      else                hipLaunchKernelGGL(qr_panel_row_la<4>, grid, dim3(512), 0, h->stream, W, M, N, ld, sW, ws.V, ws.ldv, ws.sV, ws.T, ws.sT, ws.taus, ws.sTau, j0, nb, pj0 < 0 ? 0 : pj0, wide0, nwide, QT, sQT);
      if (wc0 < N) {                                           // the next panel's columns (or the first block right of the last panel)
        const dim3 gn((unsigned)((m + 255) / 256), (unsigned)batch);
        hipLaunchKernelGGL(qr_narrow_x, gn, dim3(256), 0, h->stream, W, M, N, ld, sW, ws.V, ws.ldv, ws.sV, j0, wc0, ws.Wp, ws.sWb);
        hipLaunchKernelGGL(qr_narrow_apply, gn, dim3(256), 0, h->stream, W, M, N, ld, sW, ws.V, ws.ldv, ws.sV, ws.T, ws.sT, j0, wc0, ws.Wp, ws.sWb, 0);
      }
      pj0 = j0;
    }
    if (pj0 >= 0) {   // the last panel was a thread-per-row one; wide input: its reflector on the columns right of the block the narrow launch has done
      const int j0 = (npanels - 1) * NB, nb = L - j0 < NB ? L - j0 : NB, wc0 = j0 + nb + NB;
      if (wc0 < N)
        hipLaunchKernelGGL(qr_update_blocks<512>, dim3((unsigned)((N - wc0 + NB - 1) / NB), (unsigned)batch), dim3(512), 0, h->stream,
                           W, M, N, ld, sW, ws.V, ws.ldv, ws.sV, ws.T, ws.sT, j0, wc0);
      if (QT) hipLaunchKernelGGL(qr_update_blocks<512>, dim3((unsigned)nq, (unsigned)batch), dim3(512), 0, h->stream,
                                 QT, M, M, (long)M, sQT, ws.V, ws.ldv, ws.sV, ws.T, ws.sT, (npanels - 1) * NB, 0);
    }
    ND4_HIP(hipGetLastError());
  } else if (hr_tall) {
    // ---- M > 2048, two levels with multi-workgroup panels (round 3) ----
    // The split register panels (two 8-column halves / four 4-column quarters per 16-column slot, each followed by its own
    // block-reflector launches and a Gram product) took ~165 us per slot, and every slot read-modify-wrote the whole trailing
    // matrix. Now: outer blocks of 128 columns; inside a block the row-split panels of qrh_* (no height limit: the rows are cut into
    // 512-row workgroups) with the previous reflector riding along on the block's own columns only; after the block its compact-WY
    // factor T (128 x 128: the panels' 16 x 16 factors on the diagonal, T12 = -T1 (V1^T V2) T2 level by level from one Gram matrix)
    // takes all 128 reflectors to the columns right of it at once on the tiled MFMA kernel: X = V^T C, W = T^T X, C -= V W (K = 128).
    // Once the panels fit the register kernel (m <= 2048) the rest is factorised with one level, as a 2048-row problem. Q is formed
    // afterwards by applying the same block reflectors backwards (4/3 M^3 flop instead of the 6 M n^2 of the one-shot formation).
    const int ppb = 128 / NB;
    const size_t nbo = 128, nblocks = (size_t)(npanels + ppb - 1) / ppb;
    void* q = nullptr;
    const int ncq = N > Lq ? N : Lq;
    ND4_TRY(nd4_ws_alloc(h, sizeof(double) * (nblocks * nbo * nbo + nbo * nbo * 2 + nbo * nbo / 2 + 16 + 2 * nbo * (size_t)ncq + 64), &q));
    double* blkT = static_cast<double*>(q); double* farG = blkT + nblocks * nbo * nbo; double* farTmp = farG + nbo * nbo;
    double* farX = farTmp + nbo * nbo / 2 + 16; double* farW = farX + nbo * (size_t)ncq;
    int pnl = 0;
    int first_low = npanels;                                  // first panel of the one-level part
    for (int P0 = 0; P0 < npanels && M - P0 * NB > 2048; P0 += ppb) {
      const int pend = P0 + ppb < npanels ? P0 + ppb : npanels;
      const int bend = pend < npanels ? pend * NB : N;
      bool ok = true;
      for (pnl = P0; pnl < pend; pnl++) {
        const int j0 = pnl * NB, nb = L - j0 < NB ? L - j0 : NB;
        if (nb < NB) { ok = false; break; }                   // (a ragged last panel: only when L is not a multiple of 16; handled below)
        ND4_TRY(hr.panel(j0, bend, false, pnl == pend - 1));
      }
      if (!ok) break;
      ND4_TRY(hr.finish(bend, false, false));                 // the block's last reflector has no column of the block left to reach
      first_low = pend;
      if (bend < N) {
        const int J = P0 * NB, nblk = (pend - P0) * NB, mJ = M - J, far = N - bend;
        for (int mt = 0; mt < batch; mt++) {
          const double* Vb = ws.V + (long)mt * ws.sV + (long)J * ws.ldv + J;
          double* C = W + (long)mt * sW + (long)J * ld + bend;
          double* Tb = blkT + (size_t)(P0 / ppb) * nbo * nbo;
          ND4_TRY(wy_build_T(h, mJ, nblk, Vb, ws.ldv, ws.T + (long)mt * ws.sT + (long)P0 * NB * NB, NB, Tb, farG, farTmp));
          ND4_TRY(nd4_gemm(h, true, false, nblk, far, mJ, 1.0, Vb, ws.ldv, 0, C, ld, 0, 0.0, farX, far, 0, 1));
          ND4_TRY(nd4_gemm(h, true, false, nblk, far, nblk, 1.0, Tb, nblk, 0, farX, far, 0, 0.0, farW, far, 0, 1));
          ND4_TRY(nd4_gemm(h, false, false, mJ, far, nblk, -1.0, Vb, ws.ldv, 0, farW, far, 0, 1.0, C, ld, 0, 1));
        }
      }
    }
    // the rest as one level: multi-workgroup panels with all remaining columns as side work, then the short tail panels
    pnl = first_low;
    for (; pnl < npanels; pnl++) {
      const int j0 = pnl * NB, nb = L - j0 < NB ? L - j0 : NB, m = M - j0;
      if (nb < NB || m < hr_min_rows) break;
      ND4_TRY(hr.panel(j0, N, false, false));
    }
    ND4_TRY(hr.finish(N, false, true));
    {
      int pj0 = -1;
      for (; pnl < npanels; pnl++) {
        const int j0 = pnl * NB, nb = L - j0 < NB ? L - j0 : NB, m = M - j0;
        const int wide0 = pj0 + 2 * NB, nwide = (pj0 >= 0 && wide0 < N) ? (N - wide0 + NB - 1) / NB : 0;
        const int wc0 = j0 + nb;
        const dim3 grid((unsigned)(1 + nwide), (unsigned)batch);
        if (m <= 512)       hipLaunchKernelGGL(qr_panel_row_la<1>, grid, dim3(512), 0, h->stream, W, M, N, ld, sW, ws.V, ws.ldv, ws.sV, ws.T, ws.sT, ws.taus, ws.sTau, j0, nb, pj0 < 0 ? 0 : pj0, wide0, nwide, (double*)nullptr, 0l);
        else if (m <= 1024) hipLaunchKernelGGL(qr_panel_row_la<2>, grid, dim3(512), 0, h->stream, W, M, N, ld, sW, ws.V, ws.ldv, ws.sV, ws.T, ws.sT, ws.taus, ws.sTau, j0, nb, pj0 < 0 ? 0 : pj0, wide0, nwide, (double*)nullptr, 0l);
        else                hipLaunchKernelGGL(qr_panel_row_la<4>, grid, dim3(512), 0, h->stream, W, M, N, ld, sW, ws.V, ws.ldv, ws.sV, ws.T, ws.sT, ws.taus, ws.sTau, j0, nb, pj0 < 0 ? 0 : pj0, wide0, nwide, (double*)nullptr, 0l);
        if (wc0 < N) {
          const dim3 gn((unsigned)((m + 255) / 256), (unsigned)batch);
          hipLaunchKernelGGL(qr_narrow_x, gn, dim3(256), 0, h->stream, W, M, N, ld, sW, ws.V, ws.ldv, ws.sV, j0, wc0, ws.Wp, ws.sWb);
          hipLaunchKernelGGL(qr_narrow_apply, gn, dim3(256), 0, h->stream, W, M, N, ld, sW, ws.V, ws.ldv, ws.sV, ws.T, ws.sT, j0, wc0, ws.Wp, ws.sWb, 0);
        }
        pj0 = j0;
      }
      if (pj0 >= 0) {
        const int j0 = (npanels - 1) * NB, nb = L - j0 < NB ? L - j0 : NB, wc0 = j0 + nb + NB;
        if (wc0 < N)
          hipLaunchKernelGGL(qr_update_blocks<512>, dim3((unsigned)((N - wc0 + NB - 1) / NB), (unsigned)batch), dim3(512), 0, h->stream,
                             W, M, N, ld, sW, ws.V, ws.ldv, ws.sV, ws.T, ws.sT, j0, wc0);
      }
      ND4_HIP(hipGetLastError());
    }
    ND4_TRY(hr.dump_stamps());
    // ---- Q = H_0 H_1 ... [I; 0] by the block reflectors of 128 columns, applied backwards to E (the blocks of the one-level part
    // get their T now) ----
    {
      const long sQ = (long)M * Lq;
      ND4_TRY(nd4_set_identity(h, M, Lq, Q, Lq, batch, sQ));
      const int ncolsV = npanels * NB;
      for (int mt = 0; mt < batch; mt++) {
        for (int b = (int)nblocks - 1; b >= 0; b--) {
          const int J = b * (int)nbo, nblk = ncolsV - J < (int)nbo ? ncolsV - J : (int)nbo, mJ = M - J, nq2 = Lq - J;
          if (nq2 <= 0) continue;
          const double* Vb = ws.V + (long)mt * ws.sV + (long)J * ws.ldv + J;
          double* Tb = blkT + (size_t)b * nbo * nbo;
          if (b * ppb >= first_low || batch > 1)               // (blocks of the tall part of a single matrix still hold their T)
            ND4_TRY(wy_build_T(h, mJ, nblk, Vb, ws.ldv, ws.T + (long)mt * ws.sT + (long)b * ppb * NB * NB, NB, Tb, farG, farTmp));
          double* Qs = Q + (long)mt * sQ + (long)J * Lq + J;
          ND4_TRY(nd4_gemm(h, true, false, nblk, nq2, mJ, 1.0, Vb, ws.ldv, 0, Qs, Lq, 0, 0.0, farX, nq2, 0, 1));
          ND4_TRY(nd4_gemm(h, false, false, nblk, nq2, nblk, 1.0, Tb, nblk, 0, farX, nq2, 0, 0.0, farW, nq2, 0, 1));
          ND4_TRY(nd4_gemm(h, false, false, mJ, nq2, nblk, -1.0, Vb, ws.ldv, 0, farW, nq2, 0, 1.0, Qs, Lq, 0, 1));
        }
      }
    }
  } else {
  // Two-level blocking for M > 2048 (round 3): every panel used to read-modify-write the whole trailing matrix with a K = 16 update
  // (qr_vtc / qr_tw / rank-16 product). Now the panels of an outer block of 128 columns (ND4HIP_QR_OUTER) apply their reflectors to
  // the block's own columns only; then the block's compact-WY factor T (128 x 128: the panels' factors on the diagonal,
  // T12 = -T1 (V1^T V2) T2 level by level from ONE Gram matrix, the routine Q is formed with) takes all 128 reflectors to the rest at
  // once on the tiled MFMA kernel: X = V^T C, W = T^T X, C -= V W (K = 128). M <= 2048 without look-ahead: one level.
  static const int nbo_env = [] { const char* e = getenv("ND4HIP_QR_OUTER"); return e ? atoi(e) : 128; }();
  // Batches beyond the look-ahead form (M <= 2048), round 4: the same two levels with outer blocks of 128 columns (64 below 256 columns; ND4HIP_QR_BATCH_OUTER:
  // 32 | 48 | 64 | 128, 0 = off), everything batched — G = Vb^T Vb and the three products of the far update as strided GEMMs over the batch,
  // the block's T by one small workgroup per matrix (wy_t_small) — and Q formed backwards block by block from the stored T's: the
  // rank-16 updates of every panel over the whole trailing matrix (56 % of 1024 x 512^2) become rank-128 updates, an eighth of the traffic.
  static const int qbo_set = [] { const char* e = getenv("ND4HIP_QR_BATCH_OUTER"); return e ? atoi(e) : -1; }();
  const int qbo_env = qbo_set >= 0 ? qbo_set : (npanels >= 16 ? 128 : 64);          // 1024 x 512^2: 43.8 (off) -> 30.5 (64) -> 23.6 ms (128)
  const bool batch2 = M <= 2048 && batch > 1 && (qbo_env == 32 || qbo_env == 48 || qbo_env == 64 || qbo_env == 128) && L % NB == 0 && npanels >= 2 * (qbo_env / NB);
  const int ppb = (M > 2048 && nbo_env >= 32) ? nbo_env / NB : (batch2 ? qbo_env / NB : npanels);        // panels per outer block
  double *farT = nullptr, *farG = nullptr, *farTmp = nullptr, *farX = nullptr, *farW = nullptr;
  if (batch2) {
    const size_t nbo = (size_t)ppb * NB, nblocks = (size_t)(npanels + ppb - 1) / ppb, wcols = (size_t)(N > Lq ? N : Lq);
    void* q = nullptr;
    ND4_TRY(nd4_ws_alloc(h, sizeof(double) * (size_t)batch * (nblocks * nbo * nbo + nbo * nbo + 2 * nbo * wcols) + 64, &q));
    btT = static_cast<double*>(q); btG = btT + (size_t)batch * nblocks * nbo * nbo; btX = btG + (size_t)batch * nbo * nbo; btW = btX + (size_t)batch * nbo * wcols;
    bt_ppb = ppb; bt_sT = (long)(nblocks * nbo * nbo); bt_sX = (long)(nbo * wcols);
    ND4_HIP(hipMemsetAsync(btT, 0, sizeof(double) * (size_t)batch * nblocks * nbo * nbo, h->stream));   // (the blocks below the diagonal 64 x 64 blocks stay zero)
  } else
  if (ppb < npanels) {
    const size_t nbo = (size_t)ppb * NB;
    void* q = nullptr;
    ND4_TRY(nd4_ws_alloc(h, sizeof(double) * (nbo * nbo * 2 + nbo * nbo / 2 + 16 + 2 * nbo * (size_t)N + 64), &q));
    farT = static_cast<double*>(q); farG = farT + nbo * nbo; farTmp = farG + nbo * nbo; farX = farTmp + nbo * nbo / 2 + 16; farW = farX + nbo * (size_t)N;
  }
  for (int P0 = 0; P0 < npanels; P0 += ppb) {
  const int pend = P0 + ppb < npanels ? P0 + ppb : npanels;
  const int bend = pend < npanels ? pend * NB : N;                                // the block's reflectors reach the columns up to here at once
  for (int pnl = P0; pnl < pend; pnl++) {
    const int j0 = pnl * NB, nb = L - j0 < NB ? L - j0 : NB, m = M - j0;
    if (m > 2048 && m <= 8192) {
      // the 16-column slot in parts on 1024 threads: two 8-column halves (m <= 4096: 4 rows x 8 columns per lane) or four
      // 4-column quarters (m <= 8192: 8 rows x 4 columns per lane). After each part its reflectors are applied to the rest
      // of the slot (the live T holds only that part's block); at the end T is assembled from the side blocks and V^T V.
      const int w = m <= 4096 ? 8 : 4;
      for (int c = 0; c < nb; c += w) {
        const int nbp = nb - c < w ? nb - c : w;
        if (w == 8) hipLaunchKernelGGL((qr_panel_part<4, 8>), dim3(batch), dim3(1024), 0, h->stream, W, M, ld, sW, ws.V, ws.ldv, ws.sV, ws.T, ws.Tside, ws.sT,
                                       ws.taus, ws.sTau, j0 + c, nbp);
        else        hipLaunchKernelGGL((qr_panel_part<8, 4>), dim3(batch), dim3(1024), 0, h->stream, W, M, ld, sW, ws.V, ws.ldv, ws.sV, ws.T, ws.Tside, ws.sT,
                                       ws.taus, ws.sTau, j0 + c, nbp);
        ND4_HIP(hipGetLastError());
        if (c + w < nb)
          ND4_TRY(apply_block_reflector(h, ws, batch, M, j0, pnl, /*trans=*/1, W + (long)j0 * ld + j0 + c + w, ld, sW, nb - c - w));
      }
      if (nb > w) {
        const double* Vs = ws.V + (long)j0 * ws.ldv + j0;
        ND4_TRY(nd4_gemm(h, true, false, NB, NB, m, 1.0, Vs, ws.ldv, ws.sV, Vs, ws.ldv, ws.sV, 0.0, ws.W2, NB, ws.sW2, batch));
        hipLaunchKernelGGL(qr_t_assemble, dim3(batch), dim3(256), 0, h->stream, ws.T, ws.Tside, ws.sT, pnl, ws.W2, ws.sW2, w);
        ND4_HIP(hipGetLastError());
      }
    } else
    if (m <= 2048) {
      // batches of full panels: one workgroup per panel on the matrix cores (qr_batched_panel.h: V = Q - [S; 0], full T, like the
      // row-split panels of one matrix); short / narrow / odd-stride panels keep the thread-per-row Householder kernel
      static const bool qrb_off = [] { const char* e = getenv("ND4HIP_QR_NO_BATCHED_MFMA"); return e && *e && *e != '0'; }();
      if (!qrb_off && batch > 8 && nb == NB && m >= HR_MIN_ROWS && (ld & 1) == 0 && (ws.ldv & 1) == 0)
        launch_panel_mfma<true>(h, batch, W, M, m, ld, sW, ws.V, ws.ldv, ws.sV, ws.T, ws.sT, ws.taus, ws.sTau, j0);
      else
        launch_panel_rows(h, batch, W, M, m, ld, sW, ws.V, ws.ldv, ws.sV, ws.T, ws.sT, ws.taus, ws.sTau, j0, nb);
    } else                launch_panel<1, false>(h, batch, W, M, ld, sW, ws.V, ws.ldv, ws.sV, ws.T, ws.sT, ws.taus, ws.sTau, j0, nb);
    ND4_HIP(hipGetLastError());
    // trailing columns of the outer block: C <- H^T C = (I - V T^T V^T) C
    ND4_TRY(apply_block_reflector(h, ws, batch, M, j0, pnl, /*trans=*/1, W + (long)j0 * ld + j0 + nb, ld, sW, bend - j0 - nb));
  }
  if (batch2) {                                                                    // the block's T for every member; its reflectors on everything right of it
    const int J = P0 * NB, nblk = (pend - P0) * NB, mJ = M - J, far = N - bend;
    const long nbo2 = (long)bt_ppb * NB * bt_ppb * NB;
    const double* Vb = ws.V + (long)J * ws.ldv + J;
    double* Tb = btT + (long)(P0 / bt_ppb) * nbo2;
    ND4_TRY(nd4_gemm(h, true, false, nblk, nblk, mJ, 1.0, Vb, ws.ldv, ws.sV, Vb, ws.ldv, ws.sV, 0.0, btG, nblk, nbo2, batch));
    {
      // T of the block: 64-column halves in LDS (wy_t_small), the coupling of the two halves of a 128-column block by two products
      const int n1 = nblk < 64 ? nblk : 64, n2 = nblk - n1;
      hipLaunchKernelGGL(wy_t_small, dim3((unsigned)batch), dim3(256), 0, h->stream, ws.T + (long)P0 * NB * NB, ws.sT, btG, nblk, nbo2, Tb, nblk, bt_sT, n1);
      if (n2 > 0) {
        hipLaunchKernelGGL(wy_t_small, dim3((unsigned)batch), dim3(256), 0, h->stream, ws.T + (long)(P0 + 4) * NB * NB, ws.sT, btG + (long)n1 * nblk + n1, nblk, nbo2,
                           Tb + (long)n1 * nblk + n1, nblk, bt_sT, n2);
        ND4_TRY(nd4_gemm(h, false, false, n1, n2, n2, 1.0, btG + n1, nblk, nbo2, Tb + (long)n1 * nblk + n1, nblk, bt_sT, 0.0, btX, n2, bt_sX, batch));
        ND4_TRY(nd4_gemm(h, false, false, n1, n2, n1, -1.0, Tb, nblk, bt_sT, btX, n2, bt_sX, 0.0, Tb + n1, nblk, bt_sT, batch));
      }
      ND4_HIP(hipGetLastError());
    }
    if (far > 0) {
      double* C = W + (long)J * ld + bend;
      ND4_TRY(nd4_gemm(h, true, false, nblk, far, mJ, 1.0, Vb, ws.ldv, ws.sV, C, ld, sW, 0.0, btX, far, bt_sX, batch));
      ND4_TRY(nd4_gemm(h, true, false, nblk, far, nblk, 1.0, Tb, nblk, bt_sT, btX, far, bt_sX, 0.0, btW, far, bt_sX, batch));
      ND4_TRY(nd4_gemm(h, false, false, mJ, far, nblk, -1.0, Vb, ws.ldv, ws.sV, btW, far, bt_sX, 1.0, C, ld, sW, batch));
    }
  } else
  if (bend < N) {                                                                  // the block's 128 reflectors on everything right of it
    const int J = P0 * NB, nblk = (pend - P0) * NB, mJ = M - J, far = N - bend;
    for (int mt = 0; mt < batch; mt++) {
      const double* Vb = ws.V + (long)mt * ws.sV + (long)J * ws.ldv + J;
      double* C = W + (long)mt * sW + (long)J * ld + bend;
      ND4_TRY(wy_build_T(h, mJ, nblk, Vb, ws.ldv, ws.T + (long)mt * ws.sT + (long)P0 * NB * NB, NB, farT, farG, farTmp));
      ND4_TRY(nd4_gemm(h, true, false, nblk, far, mJ, 1.0, Vb, ws.ldv, 0, C, ld, 0, 0.0, farX, far, 0, 1));
      ND4_TRY(nd4_gemm(h, true, false, nblk, far, nblk, 1.0, farT, nblk, 0, farX, far, 0, 0.0, farW, far, 0, 1));
      ND4_TRY(nd4_gemm(h, false, false, mJ, far, nblk, -1.0, Vb, ws.ldv, 0, farW, far, 0, 1.0, C, ld, 0, 1));
    }
  }
  }
  }

  // ---- R out (tall: top N x N of the work matrix; else already in place, lower part zeroed by the panels) ----
  if (tall) {
    if (Lr > L) ND4_HIP(hipMemsetAsync(R, 0, sizeof(double) * (size_t)batch * Lr * N, h->stream));
    ND4_TRY(nd4_copy_matrix(h, L, N, W, ld, R, N, batch, sW, (long)Lr * N));
  }

  {   // undo the power-of-two normalisation on R (exact; a no-op when the exponent is 0)
    const long nR = (long)Lr * N;
    hipLaunchKernelGGL(qr_scale_apply, dim3((unsigned)((nR + 255) / 256), (unsigned)batch), dim3(256), 0, h->stream, R, R, nR, exps, +1);
  }
  // ---- Q = H_0 H_1 ... H_{p-1} [I; 0]: block reflectors applied backwards ----
  const long sQ = (long)M * Lq;
  if (hr_tall) {
    // formed above from the block reflectors
  } else if (use_qt) {
    ND4_TRY(nd4_transpose(h, Lq, M, QT, M, Q, Lq, batch, sQT, sQ));        // Q = (first Lq rows of Q^T)^T
  } else if (!wy_off && batch <= 4 && L >= 256) {
    ND4_TRY(form_q_compact_wy(h, ws, batch, M, Lq, npanels, Q, sQ));
  } else if (btT != nullptr) {
    // batches, two-level: Q <- (I - Vb Tb Vb^T) Q block by block, backwards, with the blocks' stored factors
    ND4_TRY(nd4_set_identity(h, M, Lq, Q, Lq, batch, sQ));
    const long nbo2 = (long)bt_ppb * NB * bt_ppb * NB;
    for (int P0 = ((npanels - 1) / bt_ppb) * bt_ppb; P0 >= 0; P0 -= bt_ppb) {
      const int pend = P0 + bt_ppb < npanels ? P0 + bt_ppb : npanels;
      const int J = P0 * NB, nblk = (pend - P0) * NB, mJ = M - J, nq2 = Lq - J;
      if (nq2 <= 0) continue;
      const double* Vb = ws.V + (long)J * ws.ldv + J;
      const double* Tb = btT + (long)(P0 / bt_ppb) * nbo2;
      double* Qs = Q + (long)J * Lq + J;
      ND4_TRY(nd4_gemm(h, true, false, nblk, nq2, mJ, 1.0, Vb, ws.ldv, ws.sV, Qs, Lq, sQ, 0.0, btX, nq2, bt_sX, batch));
      ND4_TRY(nd4_gemm(h, false, false, nblk, nq2, nblk, 1.0, Tb, nblk, bt_sT, btX, nq2, bt_sX, 0.0, btW, nq2, bt_sX, batch));
      ND4_TRY(nd4_gemm(h, false, false, mJ, nq2, nblk, -1.0, Vb, ws.ldv, ws.sV, btW, nq2, bt_sX, 1.0, Qs, Lq, sQ, batch));
    }
  } else {
    ND4_TRY(nd4_set_identity(h, M, Lq, Q, Lq, batch, sQ));
    for (int pnl = npanels - 1; pnl >= 0; pnl--) {
      const int j0 = pnl * NB;
      ND4_TRY(apply_block_reflector(h, ws, batch, M, j0, pnl, /*trans=*/0, Q + (long)j0 * Lq + j0, Lq, sQ, Lq - j0));
    }
  }

  // ---- reference sign convention ----
  return nd4_givens_signs(h, batch, M, L, N, tall && !full, Q, Lq, sQ, R, N, (long)Lr * N, ws.taus, ws.sTau, ws.flips);
}

// One panel factorisation on its own (the building block north_star's "HBM fraction on the QR panel" is quoted on):
// A [batch, M, 16] -> R in the top 16 x 16 of A (in place), the reflector block V [batch, M, 16] and its factor T [batch, 16, 16] with
// Q_panel = I - V T V^T. Panels of < 64 rows: the thread-per-row Householder kernel (V unit lower trapezoidal, T upper triangular);
// otherwise CholeskyQR2 + the compact orthogonal completion (V = Q - [S; 0], T = K full): up to 8 panels as the row-split launch
// (qrh_bc), more as one workgroup per panel on the matrix cores (qr_batched_panel.h).
int nd4_geqr2_panel(nd4hip_handle* h, int batch, int M, double* A, double* V, double* T) {
  Nd4WsScope scope(h);
  void* p = nullptr;
  const int parts = (M + NB + 511) / 512 + 1;
  ND4_TRY(nd4_ws_alloc(h, sizeof(double) * (size_t)batch * (NB + (parts + 1) * 256 + parts * QX_ROW_SLOT + 256 + 2) + 64, &p));
  double* taus = static_cast<double*>(p);
  const long sW = (long)M * NB;
  const int nb = M < NB ? M : NB;
  static const bool hr_off = [] { const char* e = getenv("ND4HIP_QR_NO_HR"); return e && *e && *e != '0'; }();
  if (!hr_off && batch <= 8 && M >= HR_MIN_ROWS) {
    // few panels: the row-split form (three launches, the rows over workgroups of 512): V = Q - [S; 0] and T = K of qrh_reconstruct
    QrhHost hr;
    hr.h = h; hr.batch = batch; hr.nq = 0; hr.V = V; hr.T = T; hr.ldv = NB; hr.sV = sW; hr.sT = NB * NB;
    QrhP& P = hr.P;
    double* d = taus + (size_t)batch * NB;
    P.Wm = A; P.M = M; P.N = NB; P.ld = NB; P.strideW = sW; P.Vall = V; P.ldv = NB; P.strideV = sW; P.Tall = T; P.strideT = NB * NB;
    P.taus = taus; P.strideTau = NB; P.Xp = nullptr; P.strideXp = 0;
    P.Gp = d; P.strideGp = (long)(parts + 1) * 256; d += (size_t)batch * P.strideGp;
    P.G2p = d; P.strideG2 = (long)parts * QX_ROW_SLOT; d += (size_t)batch * P.strideG2;
    P.Xch = reinterpret_cast<unsigned long long*>(P.G2p); P.strideXch = P.strideG2; P.na_shift = 0;
    P.Xsx = nullptr; P.strideXsx = 0; P.rcs_max = 1;
    ND4_HIP(hipMemsetAsync(P.G2p, 0, sizeof(double) * (size_t)batch * P.strideG2, h->stream));
    P.R1 = d; d += (size_t)batch * 256;
    P.flag = reinterpret_cast<int*>(d);
    ND4_HIP(hipMemsetAsync(P.flag, 0, sizeof(int) * (size_t)batch, h->stream));   // (as in the main path: never read before written today, but not by accident)
    P.QT = nullptr; P.strideQT = 0; P.nxp = 0; P.Xs = nullptr; P.strideXs = 0;
    P.nseg = 0; P.wide0 = 0; P.nrc = 1; P.nnw = 0; P.nqb = 0; P.skip_x = 1; P.j0 = 0; P.pj0 = -1; P.nrow = 0; P.ngp = 0;
    P.stamps = nullptr; P.stamp_slot = 0; P.status = h->xstat; { const int dp = nd4_test_drop_panel(); P.drop_tag = dp >= 0 ? dp + 1 : -1; }
    ND4_TRY(hr.panel(0, NB, false, true));
    return 0;
  }
  static const bool qrb_off = [] { const char* e = getenv("ND4HIP_QR_NO_BATCHED_MFMA"); return e && *e && *e != '0'; }();
  if (!qrb_off && nb == NB && M >= HR_MIN_ROWS) {
    // a batch of panels: one workgroup per panel on the matrix cores (qr_batched_panel.h); V = Q - [S; 0], T = K as above
    static const bool stamps_on = [] { const char* e = getenv("ND4HIP_QRB_STAMPS"); return e && *e && *e != '0'; }();
    long long* stamps = nullptr;
    if (stamps_on) { static long long* dbuf = nullptr; if (!dbuf) ND4_HIP(hipMalloc(&dbuf, 512)); stamps = dbuf; }
    static const bool copy_only = [] { const char* e = getenv("ND4HIP_QRB_COPY_ONLY"); return e && *e && *e != '0'; }();
    static const int copy_mode = [] { const char* e = getenv("ND4HIP_QRB_COPY_ONLY"); return e ? atoi(e) : 0; }();
    if (copy_only) {                                                   // debug: the kernel's data movement alone (see qrb_copy_only)
      if (M <= 256)       hipLaunchKernelGGL((qrb_copy_only<4, 1>), dim3(batch), dim3(64), 0, h->stream, A, M, (long)NB, sW, V, (long)NB, sW, 0, copy_mode);
      else if (M <= 512)  hipLaunchKernelGGL((qrb_copy_only<4, 2>), dim3(batch), dim3(128), 0, h->stream, A, M, (long)NB, sW, V, (long)NB, sW, 0, copy_mode);
      else if (M <= 1024) hipLaunchKernelGGL((qrb_copy_only<4, 4>), dim3(batch), dim3(256), 0, h->stream, A, M, (long)NB, sW, V, (long)NB, sW, 0, copy_mode);
      else                hipLaunchKernelGGL((qrb_copy_only<4, 8>), dim3(batch), dim3(512), 0, h->stream, A, M, (long)NB, sW, V, (long)NB, sW, 0, copy_mode);
      ND4_HIP(hipGetLastError());
      return 0;
    }
    launch_panel_mfma<false>(h, batch, A, M, M, NB, sW, V, NB, sW, T, NB * NB, taus, NB, 0, stamps);
    ND4_HIP(hipGetLastError());
    if (stamps) {
      long long hs[64];
      static const char* names[13] = {"load+gram", "wait all", "chol", "barrier", "q1", "gram2", "wait all", "series", "barrier", "q", "gj", "stores", "T"};
      ND4_HIP(hipStreamSynchronize(h->stream));
      ND4_HIP(hipMemcpy(hs, stamps, 512, hipMemcpyDeviceToHost));
      fprintf(stderr, "qrb stamps %d x %d rows (us | shader cycles; workgroup 0):", batch, M);
      for (int k = 0; k < 13; k++) fprintf(stderr, " %s %.2f|%lld", names[k], (hs[2 * k + 2] - hs[2 * k]) * 0.01, hs[2 * k + 3] - hs[2 * k + 1]);
      fprintf(stderr, " total %.2f|%lld; columns 4/8/12 of q1 at %lld %lld %lld of q at %lld %lld %lld\n", (hs[26] - hs[0]) * 0.01, hs[27] - hs[1],
              hs[29] - hs[9], hs[31] - hs[9], hs[33] - hs[9], hs[35] - hs[19], hs[37] - hs[19], hs[39] - hs[19]);
    }
    return 0;
  }
  launch_panel_rows(h, batch, A, M, M, NB, sW, V, NB, sW, T, NB * NB, taus, NB, 0, nb);
  ND4_HIP(hipGetLastError());
  return 0;
}

// Sign convention of the reference's Givens-built factors, applied to a Householder-built pair (Q [M, >= L] with leading
// dimension ldq, R [L, ncols] with leading dimension ldr; column j of Q and row j of R are flipped together):
//   lu_rule = false (qr_decomp_full, the square / wide branches): R_jj >= 0 wherever a reflector was needed (tau_j != 0),
//             and det Q = +1 (plane rotations) decides the last one when Q is square (M == L);
//   lu_rule = true  (the c >= 0 branches for tall input, qr.js:97-139 / bidiag.js:49-61): every leading principal minor of
//             Q's top L x L block is positive = positive pivots in its LU factorisation WITHOUT pivoting.
// flips: L ints per matrix of scratch.
int nd4_givens_signs(nd4hip_handle* h, int batch, int M, int L, int ncols, bool lu_rule, double* Q, long ldq, long sQ,
                     double* R, long ldr, long sR, const double* taus, long sTau, int* flips) {
  if (lu_rule) {
    Nd4WsScope scope2(h);
    void* q = nullptr;
    ND4_TRY(nd4_ws_alloc(h, sizeof(double) * (size_t)batch * L * L + sizeof(int32_t) * (size_t)batch * L, &q));
    double* LUq = static_cast<double*>(q);
    int32_t* Pq = reinterpret_cast<int32_t*>(LUq + (size_t)batch * L * L);
    ND4_TRY(nd4_copy_matrix(h, L, L, Q, ldq, LUq, L, batch, sQ, (long)L * L));
    ND4_TRY(nd4_getrf_nopivot(h, batch, L, LUq, LUq, Pq));
    hipLaunchKernelGGL(qr_flips_tall, dim3((unsigned)(((long)batch * L + 255) / 256)), dim3(256), 0, h->stream, LUq, L, flips, batch);
  } else {
    // qr_flips decides the parity flip by "M <= N": pass N = M when Q is square, N = M - 1 otherwise
    hipLaunchKernelGGL(qr_flips, dim3((unsigned)batch), dim3(256), 0, h->stream,
                       R, ldr, sR, taus, sTau, M, (M == L ? M : M - 1), L, flips, batch);
  }
  {
    const unsigned gy = (unsigned)(L < 512 ? L : 512);
    hipLaunchKernelGGL(qr_flip_rows, dim3((unsigned)((ncols + 255) / 256), gy, (unsigned)batch), dim3(256), 0, h->stream,
                       R, ldr, sR, L, ncols, flips, L);
    const unsigned gq = (unsigned)(M < 512 ? M : 512);
    hipLaunchKernelGGL(qr_flip_cols, dim3((unsigned)((L + 255) / 256), gq, (unsigned)batch), dim3(256), 0, h->stream,
                       Q, ldq, sQ, M, L, flips, L);
  }
  ND4_HIP(hipGetLastError());
  return 0;
}

// Q [M, Lq] (row-major, ld Lq) = (H_0 H_1 ... H_{n-1}) [I; 0] for reflectors H_j = I - tau_j v_j v_j^T given as the columns of
// V [M, n] (ld n), formed at once from their compact-WY representation H_0 ... H_{n-1} = I - V T V^T: T is upper triangular
// with the given bs x bs factors on its diagonal (bs = 1: the taus themselves) and T[1,2] = -T1 (V1^T V2) T2 for two adjacent
// groups, applied level by level (groups of bs, 2 bs, 4 bs, ... columns). One Gram matrix V^T V (TN GEMM), two
// strided-batched small GEMMs per level, W = T V[0:Lq,:]^T and Q = E - V W: ~20-30 launches and 6 M n^2 flop on the MFMA
// kernels instead of a backward loop of rank-bs updates. n must be a multiple of bs; zero columns of V are harmless.
// The n x n compact-WY factor T of n reflector columns V [M, n] (leading dimension ldv) from its bs x bs diagonal blocks Tdiag
// ([n / bs][bs][bs]; bs = 1: the taus) and the Gram matrix V^T V: T[1,2] = -T1 (V1^T V2) T2 for two adjacent groups, level by level
// (groups of bs, 2 bs, 4 bs, ... columns). G, tmp: workspaces of n * n and n * n / 2 + 16 doubles. n must be a multiple of bs.
static int wy_build_T(nd4hip_handle* h, int M, int n, const double* V, long ldv, const double* Tdiag, int bs, double* Tall, double* G, double* tmp) {
  ND4_TRY(nd4_gemm(h, true, false, n, n, M, 1.0, V, ldv, 0, V, ldv, 0, 0.0, G, n, 0, 1));
  ND4_HIP(hipMemsetAsync(Tall, 0, sizeof(double) * (size_t)n * n, h->stream));
  hipLaunchKernelGGL(wy_t_diag, dim3((unsigned)(n / bs)), dim3((unsigned)(bs * bs)), 0, h->stream, Tdiag, Tall, n, bs);
  ND4_HIP(hipGetLastError());
  for (int b = bs; b < n; b *= 2) {
    const long step = 2l * b * (n + 1);                             // from one pair of groups to the next, along the diagonal
    const int full = n / (2 * b);
    for (int f0 = 0; f0 < full; f0 += 32768) {                      // gridDim.y limit of the batched launch
      const int nf = full - f0 < 32768 ? full - f0 : 32768;
      const long o = f0 * step;
      ND4_TRY(nd4_gemm(h, false, false, b, b, b, 1.0, G + b + o, n, step, Tall + (long)b * (n + 1) + o, n, step, 0.0, tmp, b, (long)b * b, nf));
      ND4_TRY(nd4_gemm(h, false, false, b, b, b, -1.0, Tall + o, n, step, tmp, b, (long)b * b, 0.0, Tall + b + o, n, step, nf));
    }
    const int i0 = full * 2 * b, n2 = n - i0 - b;                    // ragged last pair: second group narrower
    if (n2 > 0) {
      ND4_TRY(nd4_gemm(h, false, false, b, n2, n2, 1.0, G + (long)i0 * n + i0 + b, n, 0, Tall + (long)(i0 + b) * (n + 1), n, 0, 0.0, tmp, n2, 0, 1));
      ND4_TRY(nd4_gemm(h, false, false, b, n2, b, -1.0, Tall + (long)i0 * (n + 1), n, 0, tmp, n2, 0, 0.0, Tall + (long)i0 * n + i0 + b, n, 0, 1));
    }
  }
  return 0;
}

int nd4_wy_form(nd4hip_handle* h, int M, int n, const double* V, const double* Tdiag, int bs, double* Q, int Lq) {
  Nd4WsScope scope(h);
  void* p = nullptr;
  ND4_TRY(nd4_ws_alloc(h, sizeof(double) * ((size_t)n * n * 2 + (size_t)n * n / 2 + (size_t)n * Lq + 64), &p));
  double* G = static_cast<double*>(p);
  double* Tall = G + (size_t)n * n;
  double* tmp = Tall + (size_t)n * n;
  double* W = tmp + (size_t)n * n / 2 + 16;
  ND4_TRY(wy_build_T(h, M, n, V, n, Tdiag, bs, Tall, G, tmp));
  ND4_TRY(nd4_gemm(h, false, true, n, Lq, n, 1.0, Tall, n, 0, V, n, 0, 0.0, W, Lq, 0, 1));                   // W = T V[0:Lq,:]^T
  ND4_TRY(nd4_set_identity(h, M, Lq, Q, Lq, 1, (long)M * Lq));
  return nd4_gemm(h, false, false, M, Lq, n, -1.0, V, n, 0, W, Lq, 0, 1.0, Q, Lq, 0, 1);                     // Q = E - V W
}
