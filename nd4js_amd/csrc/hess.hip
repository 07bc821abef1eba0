// Batched Hessenberg reduction on the device.
//
// Replaces src/la/hessenberg.js:89-115 (hessenberg_decomp; kernel _hessenberg_decomp :27-86): A = U H U^T with H upper
// Hessenberg. The reference finishes the rows from the bottom up: for i = N-1 .. 2 a Householder reflector
// P = I - v v^T (v^T v = 2, support 0..i-1, FrobeniusNorm scaling of norm.js:22-67, sign chosen against cancellation)
// is applied to H from the right (rows < i), from the left (rows < i, all columns) and to U from the right. The same
// reflectors are used here (so H and U agree with the reference to rounding); the three applications of one step are
// merged algebraically into one read pass and one read-modify-write pass over the active part:
//     y = H[0:i,0:i] v      x = H[0:i,:]^T v      yu = U[0:N-1,0:i] v                      (hess_pass_a)
//     H[j,k] -= y_j v_k + v_j (x_k - (v^T y) v_k)      U[j,k] -= yu_j v_k                   (hess_pass_b)
// (v_k = 0 for k >= i, so the formulas hold for every column). Bound: HBM — 2 reads + 1 write of the active part of H
// and of U per step; four launches per step (vector, read pass, partial reduction, update pass).
#include "nd4hip_internal.h"
#include "xchg16.h"
#include "dpp.h"
#include <cstdlib>

namespace {

constexpr int RW = 2;             // rows per wave in the read pass: small, so that even the late steps fill the chip
constexpr int HR = 4 * RW;        // rows per workgroup (and per partial of x)

struct HessWs {
  double* v; double* y; double* yu; double* w; double* xpart; int* skip;   // per matrix: v, y, yu, w [N]; xpart [P][N]; skip
  long sV, sX;                                                      // strides per matrix
  double* vstore; int nstore;                                       // all reflectors as columns (N x nstore per matrix), or null
};

__device__ __forceinline__ double block_max(double v, double* s_red) {
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off));
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
  __syncthreads();
  v = fmax(fmax(s_red[0], s_red[1]), fmax(s_red[2], s_red[3]));
  __syncthreads();
  return v;
}
__device__ __forceinline__ double block_sum(double v, double* s_red) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
  __syncthreads();
  v = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
  __syncthreads();
  return v;
}

// Householder vector of row i (hessenberg.js:43-56) and the finished row (:83-84). One workgroup per matrix.
__global__ __launch_bounds__(256) void hess_vec(double* __restrict__ Hm, int N, int i, HessWs ws) {
  __shared__ double s_red[4];
  double* row = Hm + (long)blockIdx.x * N * N + (long)i * N;
  double* v = ws.v + blockIdx.x * ws.sV;
  const int t = threadIdx.x, ii = i - 1;
  double m1 = 0.0;
  for (int j = t; j < ii; j += 256) m1 = fmax(m1, fabs(row[j]));
  m1 = block_max(m1, s_red);
  double* vcol = ws.vstore ? ws.vstore + (long)blockIdx.x * N * ws.nstore + (N - 1 - i) : nullptr;   // column N-1-i (processing order)
  if (m1 == 0.0) {                                                  // NORM.max === 0 -> continue (:46)
    for (int j = t; j < N; j += 256) v[j] = 0.0;
    if (t == 0) ws.skip[blockIdx.x] = 1;
    return;
  }
  const double hii0 = row[ii];
  const double mx = fmax(m1, fabs(hii0));
  double ss = 0.0;
  for (int j = t; j < i; j += 256) { const double x = row[j] / mx; ss += x * x; }
  ss = block_sum(ss, s_red);
  const double nrm = (isfinite(mx) ? sqrt(ss) * mx : mx) * (hii0 > 0 ? -1.0 : 1.0);        // :47
  const double hii = hii0 - nrm;                                                            // :48
  const double mx2 = fmax(m1, fabs(hii));
  double s2 = 0.0;
  for (int j = t; j < ii; j += 256) { const double x = row[j] / mx2; s2 += x * x; }
  s2 = block_sum(s2, s_red);
  { const double x = hii / mx2; s2 += x * x; }
  const double div = sqrt(s2);
  for (int j = t; j < N; j += 256) {
    double vj = 0.0;
    if (j < i) vj = (j == ii ? hii : row[j]) / mx2 * 1.4142135623730951 / div;              // :51-52
    v[j] = vj;
    if (vcol) vcol[(long)j * ws.nstore] = vj;
  }
  __syncthreads();
  for (int j = t; j < ii; j += 256) row[j] = 0.0;                                           // :83
  if (t == 0) { row[ii] = nrm; ws.skip[blockIdx.x] = 0; }                                   // :84
}

// blockIdx.y = 0: rows of H (y and the column partials of x); 1: rows of U (yu). One wave = RW rows.
__global__ __launch_bounds__(256) void hess_pass_a(const double* __restrict__ Hm, const double* __restrict__ Um, int N, int i, HessWs ws) {
  const long b = blockIdx.z;
  if (ws.skip[b]) return;
  const bool isU = blockIdx.y == 1;
  const int nrows = isU ? N - 1 : i;
  const int r0 = blockIdx.x * HR;
  if (r0 >= nrows) return;
  const double* M = (isU ? Um : Hm) + b * (long)N * N;
  const double* v = ws.v + b * ws.sV;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int rw = r0 + wave * RW;
  const int ncols = isU ? i : N;                                    // v_k = 0 for k >= i: U needs no column beyond i
  __shared__ double s_x[4][256];
  double yacc[RW];
#pragma unroll
  for (int r = 0; r < RW; r++) yacc[r] = 0.0;
  double vr[RW];
#pragma unroll
  for (int r = 0; r < RW; r++) vr[r] = (!isU && rw + r < nrows) ? v[rw + r] : 0.0;
  for (int c0 = 0; c0 < ncols; c0 += 256) {
    // 4 x 64-column slices per iteration: lane handles columns c0 + q*64 + lane
    double xs[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int k = c0 + q * 64 + lane;
      if (k < ncols) {
        const double vk = v[k];
#pragma unroll
        for (int r = 0; r < RW; r++) {
          if (rw + r < nrows) {
            const double h = M[(long)(rw + r) * N + k];
            yacc[r] += h * vk;
            xs[q] += vr[r] * h;
          }
        }
      }
    }
    if (!isU) {                                                     // combine the 4 waves, one partial per workgroup
#pragma unroll
      for (int q = 0; q < 4; q++) s_x[wave][q * 64 + lane] = xs[q];
      __syncthreads();
      const int k = c0 + t;
      if (k < N) ws.xpart[b * ws.sX + (long)blockIdx.x * N + k] = (s_x[0][t] + s_x[1][t]) + (s_x[2][t] + s_x[3][t]);
      __syncthreads();
    }
  }
  double* yout = (isU ? ws.yu : ws.y) + b * ws.sV;
#pragma unroll
  for (int r = 0; r < RW; r++) {
    double s = yacc[r];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0 && rw + r < nrows) yout[rw + r] = s;
  }
}

// w_k = x_k - (v^T y) v_k with x = sum of the row-group partials. One workgroup per 32 columns; 8 thread groups share
// the partials of a column (short dependent chains, fixed summation order: deterministic).
constexpr int RC = 32, RG = 8;
__global__ __launch_bounds__(256) void hess_reduce(int N, int i, HessWs ws) {
  const long b = blockIdx.y;
  if (ws.skip[b]) return;
  __shared__ double s_red[4];
  __shared__ double s_part[RG][RC];
  const double* v = ws.v + b * ws.sV;
  const double* y = ws.y + b * ws.sV;
  double part = 0.0;
  for (int j = threadIdx.x; j < i; j += 256) part += v[j] * y[j];
  const double vy = block_sum(part, s_red);                          // v^T y
  const int c = threadIdx.x % RC, g = threadIdx.x / RC;
  const int k = blockIdx.x * RC + c;
  const int P = (i + HR - 1) / HR;
  double xk = 0.0;
  if (k < N)
    for (int p = g; p < P; p += RG) xk += ws.xpart[b * ws.sX + (long)p * N + k];
  s_part[g][c] = xk;
  __syncthreads();
  if (g == 0 && k < N) {
    double x = 0.0;
#pragma unroll
    for (int q = 0; q < RG; q++) x += s_part[q][c];
    ws.w[b * ws.sV + k] = x - vy * v[k];
  }
}

// blockIdx.z = batch, blockIdx.y = row group, blockIdx.x = column chunk of 256; gridDim.x covers H's N columns first,
// then U's (flag in the upper half of blockIdx.x): see the launcher.
__global__ __launch_bounds__(256) void hess_pass_b(double* __restrict__ Hm, double* __restrict__ Um, int N, int i, int cchunks, HessWs ws) {
  const long b = blockIdx.z;
  if (ws.skip[b]) return;
  const bool isU = blockIdx.x >= (unsigned)cchunks;
  const int cx = isU ? blockIdx.x - cchunks : blockIdx.x;
  const int nrows = isU ? N - 1 : i;
  const int r0 = blockIdx.y * HR;
  const int k = cx * 256 + threadIdx.x;
  if (r0 >= nrows) return;
  if (isU && cx * 256 >= i) return;
  const double* v = ws.v + b * ws.sV;
  const double vk = (k < N) ? v[k] : 0.0;
  if (isU) {
    if (k >= i) return;
    double* U = Um + b * (long)N * N;
    const double* yu = ws.yu + b * ws.sV;
    for (int j = r0; j < r0 + HR && j < nrows; j++) U[(long)j * N + k] -= yu[j] * vk;
    return;
  }
  const double* y = ws.y + b * ws.sV;
  if (k >= N) return;
  const double wk = ws.w[b * ws.sV + k];
  double* H = Hm + b * (long)N * N;
  for (int j = r0; j < r0 + HR && j < nrows; j++) H[(long)j * N + k] -= y[j] * vk + v[j] * wk;
}

// ================================================================ blocked form (one large matrix) ==========================
// Within a block of NBH steps H is NOT modified: with the reflectors V, and W, Y collected as columns,
//     H_k = H_0 - V W^T - Y V^T          (each step: H' = H - v w^T - y v^T with y = H v, w = H^T v - (v^T y) v)
// so a step needs ONE read pass over H_0 (y = H_0 v, x = H_0^T v) plus corrections with the N x k arrays,
//     y -= V (W^T v) + Y (V^T v),   x -= W (V^T v) + V (Y^T v),   v^T y = v^T y_raw - sum_j (a_j b_j + b_j c_j)
// (a = W^T v, b = V^T v, c = Y^T v), and the row the next reflector is built from is H_0[i,:] - V[i,:] W^T - Y[i,:] V^T.
// The rank-2 read-modify-write pass of every step (hess_pass_b) becomes two GEMMs per block on the MFMA kernel.
// V, W, Y are kept TRANSPOSED ([NBH][N], one reflector per row): every access below is contiguous across threads.
constexpr int NBH = 32;
constexpr int BR = 16;            // rows per workgroup of the read pass (32: fewer, fatter workgroups, 12.7 us instead of 9.4)
constexpr int BC = 512;           // columns per workgroup of the read pass (256 threads x one 16-byte load per row)

struct HessBlk {
  double *Vt, *Wt, *Yt;          // [NBH][N], zero outside the steps done in this block
  double *dots;                  // a, b, c: [3][NBH]
  double *nrm;                   // [N]: the sub-diagonal entry of every finished row
  double *ypart;                 // [N / BC][N]: per column chunk partials of y
  double *vyp;                   // [row groups x column chunks]: partials of v^T y_raw
  int* skipv;                    // [N]: step i skipped (row already in Hessenberg form)
  double* vrows;                 // [nstore][N]: all reflectors as rows, in processing order (transposed into ws.vstore at the end)
  double* nextrow;               // [N]: the corrected row of the NEXT step, left behind by hessb_reduce
};

// step i = k-th of its block: the Householder vector of row i (hessenberg.js:43-56) from the corrected row x (columns < i), by EVERY
// workgroup of the read pass, into LDS (s_v[N]); the lead workgroup stores it. x = row i of H (k = 0: the block's updates have been
// applied by the GEMMs) or what hessb_reduce of the previous step left in bk.nextrow. As a launch of its own (one workgroup) this was
// 8.9 us of the 27 us per step. Returns false when the step is skipped (row already in Hessenberg form, :46).
__device__ __forceinline__ bool hessb_reflector(double* __restrict__ s_v, double* s_red, const double* __restrict__ H, int N, int i, int k,
                                                HessWs ws, HessBlk bk, bool lead) {
  constexpr int T = 256;
  const int t = threadIdx.x, ii = i - 1;
  const double* x = (k == 0) ? H + (long)i * N : bk.nextrow;
  double m1 = 0.0;                                     // max |row[j]|, j < i-1
  for (int c0 = t; c0 < i; c0 += 8 * T) {              // eight loads in flight at a time
    double xv[8];
#pragma unroll
    for (int q = 0; q < 8; q++) xv[q] = (c0 + q * T < i) ? x[c0 + q * T] : 0.0;
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const int c = c0 + q * T;
      if (c < i) { s_v[c] = xv[q]; if (c < ii) m1 = fmax(m1, fabs(xv[q])); }
    }
  }
  m1 = block_max(m1, s_red);                           // (its barriers publish s_v)
  double* vrow = bk.vrows + (long)(N - 1 - i) * N;    // reflector number N-1-i (processing order)
  if (m1 == 0.0) {                                    // NORM.max === 0 -> continue (:46)
    if (lead) {
      for (int j = t; j < N; j += T) ws.v[j] = 0.0;
      if (t == 0) { ws.skip[0] = 1; bk.skipv[i] = 1; }
    }
    return false;
  }
  // ONE scaled sum of squares over the entries left of (i, i-1), S1 = sum (x_j / m1)^2; the two FrobeniusNorm results of
  // hessenberg.js:47-50 follow from it: sum over j <= i-1 of (x_j / mx)^2 = S1 (m1/mx)^2 + (h_ii / mx)^2 (same scaling, so no
  // over/underflow either; rounding-level difference to accumulating them entry by entry)
  double S1 = 0.0;
  for (int j = t; j < ii; j += T) { const double q = s_v[j] / m1; S1 += q * q; }
  S1 = block_sum(S1, s_red);
  const double hii0 = s_v[ii];
  const double mx = fmax(m1, fabs(hii0));
  const double q1 = m1 / mx, q0 = hii0 / mx;
  const double ss = S1 * q1 * q1 + q0 * q0;
  const double nrm = (isfinite(mx) ? sqrt(ss) * mx : mx) * (hii0 > 0 ? -1.0 : 1.0);        // :47
  const double hii = hii0 - nrm;                                                            // :48
  const double mx2 = fmax(m1, fabs(hii));
  const double r1 = m1 / mx2, r0 = hii / mx2;
  const double div = sqrt(S1 * r1 * r1 + r0 * r0);
  const double scale = 1.4142135623730951 / div;
  for (int j = t; j < N; j += T) {
    double vj = 0.0;
    if (j < i) vj = (j == ii ? hii : s_v[j]) / mx2 * scale;                                // :51-52
    s_v[j] = vj;
    if (lead) { ws.v[j] = vj; vrow[j] = vj; bk.Vt[(long)k * N + j] = vj; }
  }
  if (lead && t == 0) { ws.skip[0] = 0; bk.skipv[i] = 0; bk.nrm[i] = nrm; }
  __syncthreads();
  return true;
}

// The read pass over H_0[0:i, :]: workgroup (row group g, column chunk c) reads BR rows x BC columns with ONE 16-byte load per
// thread and row, all in flight at once; x partial [g][columns] (a thread owns its two columns: no reduction), y partial
// [c][rows] (wave reduction per row). The workgroups past the grid of the pass compute a = W^T v, b = V^T v, c = Y^T v.
__global__ __launch_bounds__(256) void hessb_pass(const double* __restrict__ H, int N, int i, int k, int ngroups, int nchunks, HessWs ws, HessBlk bk) {
  extern __shared__ double s_v[];                                  // [N]: the reflector of this step
  __shared__ double s_redv[4];
  if (!hessb_reflector(s_v, s_redv, H, N, i, k, ws, bk, blockIdx.x == 0)) return;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const double* v = s_v;
  if ((int)blockIdx.x >= ngroups * nchunks) {                     // ---- the 3 k dot products, one wave each
    const int d = ((int)blockIdx.x - ngroups * nchunks) * 4 + wave;
    if (d >= 3 * k) return;
    const int which = d / k, j = d % k;
    const double* M = (which == 0 ? bk.Wt : (which == 1 ? bk.Vt : bk.Yt)) + (long)j * N;
    double s = 0.0;
    for (int r0 = lane; r0 < i; r0 += 8 * 64) {                     // eight loads in flight at a time (see hessb_reduce)
      double mv[8];
#pragma unroll
      for (int q = 0; q < 8; q++) mv[q] = (r0 + 64 * q < i) ? M[r0 + 64 * q] : 0.0;
#pragma unroll
      for (int q = 0; q < 8; q++) if (r0 + 64 * q < i) s += mv[q] * v[r0 + 64 * q];
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) bk.dots[which * NBH + j] = s;
    return;
  }
  __shared__ double s_y[4][BR];
  const int g = blockIdx.x / nchunks, ch = blockIdx.x % nchunks;
  const int r0 = g * BR, c = ch * BC + 2 * t;
  const bool incol = c < N;                                       // N is even here (N >= 512 and the loads are 16-byte)
  const double2 vc = (incol && c < i) ? double2{v[c], (c + 1 < i) ? v[c + 1] : 0.0} : double2{0.0, 0.0};   // y only sums columns < i
  double2 hrow[BR];
#pragma unroll
  for (int r = 0; r < BR; r++)
    hrow[r] = (incol && r0 + r < i) ? *reinterpret_cast<const double2*>(H + (long)(r0 + r) * N + c) : double2{0.0, 0.0};
  double2 x = double2{0.0, 0.0};
  double yp[BR];
#pragma unroll
  for (int r = 0; r < BR; r++) {
    const double vr = (r0 + r < i) ? v[r0 + r] : 0.0;
    x.x += vr * hrow[r].x; x.y += vr * hrow[r].y;
    yp[r] = hrow[r].x * vc.x + hrow[r].y * vc.y;
  }
  if (incol) *reinterpret_cast<double2*>(ws.xpart + (long)g * N + c) = x;
#pragma unroll
  for (int r = 0; r < BR; r++) {
    double s = yp[r];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) s_y[wave][r] = s;
  }
  __syncthreads();
  if (t < 64) {                                                   // wave 0: the y partials of the BR rows and their share of v^T y
    double yv = 0.0;
    if (t < BR && r0 + t < i) {
      const double ys = (s_y[0][t] + s_y[1][t]) + (s_y[2][t] + s_y[3][t]);
      bk.ypart[(long)ch * N + r0 + t] = ys;
      yv = ys * v[r0 + t];
    }
    for (int off = 32; off > 0; off >>= 1) yv += __shfl_xor(yv, off);
    if (t == 0) bk.vyp[blockIdx.x] = yv;
  }
}

// y and w of step (i, k) from the partials of the pass and the corrections; rows k of Yt and Wt. 32 indices per workgroup,
// 8 thread groups share the partial sums of an index (short chains, fixed order: deterministic).
__global__ __launch_bounds__(256) void hessb_reduce(const double* __restrict__ H, int N, int i, int k, int nchunks, int want_next, HessWs ws, HessBlk bk) {
  __shared__ double s_red[4];
  __shared__ double s_a[NBH], s_b[NBH], s_c[NBH];
  __shared__ double s_px[RG][RC], s_py[RG][RC];
  __shared__ double s_vn[NBH + 1], s_yn[NBH + 1], s_ynp[RG];
  const int t = threadIdx.x;
  const bool skip = ws.skip[0] != 0;
  const int cidx = t % RC, grp = t / RC;
  const int e = blockIdx.x * RC + cidx;
  const int inext = i - 1;                                             // the next step's row
  double wke = 0.0;                                                    // W_k[e] of this step (0 when skipped)
  if (t < NBH) { s_a[t] = t < k ? bk.dots[t] : 0.0; s_b[t] = t < k ? bk.dots[NBH + t] : 0.0; s_c[t] = t < k ? bk.dots[2 * NBH + t] : 0.0; }
  if (!skip) {
    // v^T y_raw from the per-workgroup partials of the pass (fixed order: deterministic)
    double part = 0.0;
    for (int j = t; j < ((i + BR - 1) / BR) * nchunks; j += 256) part += bk.vyp[j];
    double vy = block_sum(part, s_red);                                // (ends with a barrier: s_a, s_b, s_c are visible)
    for (int j = 0; j < k; j++) vy -= s_a[j] * s_b[j] + s_b[j] * s_c[j];
    const int P = (i + BR - 1) / BR;
    double x = 0.0, y = 0.0;
    if (e < N) {
      // (eight loads in flight at a time: a rolled loop of dependent load-add pairs costs a memory round trip per partial)
      for (int p0 = grp; p0 < P; p0 += 8 * RG) {
        double xv[8];
#pragma unroll
        for (int q = 0; q < 8; q++) xv[q] = (p0 + q * RG < P) ? ws.xpart[(long)(p0 + q * RG) * N + e] : 0.0;
#pragma unroll
        for (int q = 0; q < 8; q++) x += xv[q];
      }
      // corrections: thread group grp takes the steps j = grp, grp + RG, ...
      for (int j = grp; j < k; j += RG) {
        const double wj = bk.Wt[(long)j * N + e], vj = bk.Vt[(long)j * N + e], yj = bk.Yt[(long)j * N + e];
        x -= wj * s_b[j] + vj * s_c[j];
        y -= vj * s_a[j] + yj * s_b[j];
      }
      if (e < i) for (int ch = grp; ch < nchunks; ch += RG) y += bk.ypart[(long)ch * N + e];
    }
    s_px[grp][cidx] = x; s_py[grp][cidx] = y;
    // y_k at the next step's row index, by the same thread groups in the same order as its owner computes it (every workgroup needs it)
    if (want_next && cidx == 0) {
      double yn = 0.0;
      for (int j = grp; j < k; j += RG) yn -= bk.Vt[(long)j * N + inext] * s_a[j] + bk.Yt[(long)j * N + inext] * s_b[j];
      for (int ch = grp; ch < nchunks; ch += RG) yn += bk.ypart[(long)ch * N + inext];
      s_ynp[grp] = yn;
    }
    __syncthreads();
    if (grp == 0 && e < N) {
      double xs = 0.0, ys = 0.0;
#pragma unroll
      for (int q = 0; q < RG; q++) { xs += s_px[q][cidx]; ys += s_py[q][cidx]; }
      wke = xs - vy * ws.v[e];
      bk.Wt[(long)k * N + e] = wke;
      bk.Yt[(long)k * N + e] = (e < i) ? ys : 0.0;
    }
  }
  if (!want_next) return;
  // ---- the row the NEXT reflector is built from: H_0[i-1, :] - V[i-1, :] W^T - Y[i-1, :] V^T over the k + 1 steps so far ----
  if (t <= k && t <= NBH) {
    if (t < k) { s_vn[t] = bk.Vt[(long)t * N + inext]; s_yn[t] = bk.Yt[(long)t * N + inext]; }
    else if (skip) { s_vn[t] = 0.0; s_yn[t] = 0.0; }
    else {
      double yn = 0.0;
#pragma unroll
      for (int q = 0; q < RG; q++) yn += s_ynp[q];
      s_vn[t] = ws.v[inext]; s_yn[t] = yn;
    }
  }
  __syncthreads();
  if (grp == 0 && e < inext) {
    double xr = H[(long)inext * N + e];
    for (int j = 0; j < k; j++) xr -= s_vn[j] * bk.Wt[(long)j * N + e] + s_yn[j] * bk.Vt[(long)j * N + e];
    if (!skip) xr -= s_vn[k] * wke + s_yn[k] * ws.v[e];
    bk.nextrow[e] = xr;
  }
}

// rows finished in the block [ilo, ihi]: zeros left of the sub-diagonal, the norm on it (hessenberg.js:83-84)
__global__ __launch_bounds__(256) void hessb_fix(double* __restrict__ H, int N, int ilo, int ihi, HessBlk bk) {
  const int i = ilo + blockIdx.y;
  if (i > ihi || bk.skipv[i]) return;
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c < i - 1) H[(long)i * N + c] = 0.0;
  else if (c == i - 1) H[(long)i * N + c] = bk.nrm[i];
}

__global__ void hess_fill(double* __restrict__ x, int n, double val) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] = val;
}

// U <- [[I, 0], [0, 1]] pattern of hessenberg.js:33 / :88-89 (identity; the reflectors never touch the last row/column)
__global__ void hess_init_u(double* __restrict__ Um, int N) {
  const long base = (long)blockIdx.z * N * N;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= N) return;
  for (int r = blockIdx.y; r < N; r += gridDim.y) Um[base + (long)r * N + j] = (r == j) ? 1.0 : 0.0;
}


// ================================================================ one launch for the whole reduction (N <= 2048) ============
// VERDICT r3 #6. The blocked form above is two DEPENDENT launches per row (18 us at 2048^2, each at the latency floor of a small
// kernel), i.e. it is defined by its kernel boundaries. Here 16 x 16 co-resident workgroups own H as 16E x 16E tiles IN REGISTERS
// (2048^2: 128 KB per workgroup, 64 values per thread) for all N - 2 steps; nothing of H crosses HBM between the first load and
// the last store, so the rank-2 update of every step costs no traffic and the unblocked recurrence of hessenberg.js:27-86 is
// used as it stands (no V / W / Y history, no GEMMs). Thread (tr, tc) of workgroup (p, q) holds rows 16E p + tr + 16 a and
// columns 16E q + tc + 16 b. Per step two rounds of tagged words (xchg.h) cross between the workgroups, a few values per thread:
//   B  every workgroup publishes its partials of x = H^T v (its 16E columns), y = H v (its 16E rows) and v^T H v, and the
//      workgroups holding row i-1 publish it as it stands; workgroup (p, q) sums, in a fixed order, the 16 partials of E columns
//      of x and of E rows of y and all 256 partials of v^T y: w = x - (v^T y) v; from y at row i-1 it also knows its E columns of
//      row i-1 AFTER this step's update (the row the next reflector is built from) and their scaled-norm partial;
//   C  it publishes those 3E + 2 values; (p, q) picks up w over its columns, y over its rows, the next row over both, all 256
//      norm partials and the entry (i-1, i-2), and updates its tile. Every workgroup then builds the next v from the same scalars.
// Tags are the row index (never repeat within a launch; the area is cleared per call), areas alternate with the parity of i; each
// round ends with every workgroup having read something of every other, which is what keeps a fast one from overwriting an area
// a slow one has not read yet (a skipped step — row already in Hessenberg form — therefore still runs both rounds, with zeros).
// Every spin is bounded; a time-out raises the handle's status word and an abort flag that ends every workgroup
// (ND4HIP_ERR_XCHG at the next synchronising call). The scalars of the reflector use the few-ulp reciprocal / reciprocal square
// root of dpp.h: an IEEE division is ~35 dependent instructions and the chain is the step.
constexpr unsigned HP_REP_BV = 2 * 256 * 16, HP_REP_CN = 2 * 512 * 16;         // bytes per copy of the arrays everybody reads (HP_NREP copies, xchg16.h)
struct HessPx {
  qx_u64* base;          // one buffer: every value that crosses is a 16-byte pair of tagged words, stored and loaded as ONE access
  unsigned oB, oC;       // [2][256][3T | 3E] values: what a workgroup publishes for its row / column partners (byte offsets)
  unsigned oBv, oCn;     // [2][256][1 | 2] values: what every workgroup reads of every other (v^T H v partials; norm partials),
  unsigned bytes;        // contiguous so that a wave reads them as whole cache lines
  int* abort;
};
template <int E>
__global__ __launch_bounds__(256) void hessp(double* __restrict__ H, int N, double* __restrict__ vrows, HessPx X, int* status, int drop_row, long long* stamps, int delay) {
  using namespace nd4dpp;
  constexpr int T = 16 * E;
  constexpr int BV = 3 * T, CV = 3 * E;                              // values per slot
  constexpr int LE = E == 8 ? 3 : (E == 4 ? 2 : 1);                 // log2 E
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, tr = t >> 4, tc = t & 15;
  const int wg = blockIdx.x, p = wg >> 4, q = wg & 15;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(X.base, 0, (int)X.bytes, 0x00020000);
  __shared__ double s_rc[T], s_rr[T], s_v[T], s_w[T], s_y[T], s_xp[4][T], s_yp[T], s_q[32 * E], s_e[16 + E];
  __shared__ double s_wm[4], s_wS[4], s_vy[4], s_red[4], s_h0;
  __shared__ int s_dead;
  double h[E][E];
#pragma unroll
  for (int a = 0; a < E; a++)
#pragma unroll
    for (int b = 0; b < E; b++) {
      const int r = T * p + tr + 16 * a, c = T * q + tc + 16 * b;
      h[a][b] = (r < N && c < N) ? H[(long)r * N + c] : 0.0;
    }
  if (t == 0) s_dead = 0;
  bool dead = false;
  // ---- the first row (N-1) straight from memory: its slices, and the norm partials grouped exactly as the later steps group them
  {
    const double* row = H + (long)(N - 1) * N;
    if (t < T) s_rc[t] = (T * q + t < N) ? row[T * q + t] : 0.0;
    else if (t < 2 * T) s_rr[t - T] = (T * p + t - T < N) ? row[T * p + t - T] : 0.0;
    if (t == 0) s_h0 = row[N - 2];
    double rv[E], m = 0.0;
#pragma unroll
    for (int c = 0; c < E; c++) {
      const int e = T * (t & 15) + E * (t >> 4) + c;
      rv[c] = (e < N - 2) ? fabs(row[e]) : 0.0;
      m = fmax(m, rv[c]);
    }
    double ss = 0.0;
    const double im = m > 0.0 ? fast_rcp(m) : 0.0;
#pragma unroll
    for (int c = 0; c < E; c++) { const double x = rv[c] * im; ss += x * x; }
    const double wm = wave_max(m);
    const double f = wm > 0.0 ? m * fast_rcp(wm) : 0.0;
    const double wS = wave_sum(ss * f * f);
    if (lane == 0) { s_wm[wave] = wm; s_wS[wave] = wS; }
  }
  __syncthreads();
  // ---- the reflector's scalars (hessenberg.js:43-56), by every thread in the same order; branch-free (a zero row, :46, is selected
  // at the end) so that the chain can be scheduled between the FMAs of the tile update that precedes it
  struct Refl { bool skip; double nrm, vii, inv2, scale; };
  auto reflector = [&]() -> Refl {
    const double m1 = fmax(fmax(s_wm[0], s_wm[1]), fmax(s_wm[2], s_wm[3]));
    const bool skip = m1 == 0.0;                                  // NORM.max === 0 -> continue (:46)
    const double m1s = skip ? 1.0 : m1;
    const double i1 = fast_rcp(m1s);
    double S1 = 0.0;
#pragma unroll
    for (int g = 0; g < 4; g++) { const double f = s_wm[g] * i1; S1 += s_wS[g] * f * f; }
    const double hii0 = s_h0;
    const double mx = fmax(m1s, fabs(hii0));
    const double imx = fast_rcp(mx);
    const double q1 = m1s * imx, q0 = hii0 * imx;
    const double ss = S1 * q1 * q1 + q0 * q0;
    const double nrm = (isfinite(mx) ? ss * fast_rsqrt(ss) * mx : mx) * (hii0 > 0 ? -1.0 : 1.0);        // :47
    const double hii = hii0 - nrm;                                                                        // :48
    const double mx2 = fmax(m1s, fabs(hii));
    const double inv2 = fast_rcp(mx2);
    const double r1 = m1s * inv2, r0 = hii * inv2;
    const double scale = 1.4142135623730951 * fast_rsqrt(S1 * r1 * r1 + r0 * r0);
    Refl r;
    r.skip = skip; r.nrm = skip ? 0.0 : nrm; r.inv2 = skip ? 0.0 : inv2; r.scale = skip ? 0.0 : scale; r.vii = skip ? 0.0 : hii * inv2 * scale;
    return r;
  };
  Refl rf = reflector();
  long long tacc[6] = {0, 0, 0, 0, 0, 0}, tlast = stamps ? (long long)wall_clock64() : 0;   // ND4HIP_HESSP_STAMPS: time per phase (100 MHz ticks), thread 0
#define HP_STAMP(k) if (stamps) { const long long now = (long long)wall_clock64(); tacc[k] += now - tlast; tlast = now; }
  for (int i = N - 1; i > 1; i--) {
    const unsigned tag = (unsigned)i;
    const int par = i & 1, ii = i - 1;
    const int pi = i / T, pn = ii / T, ln = ii % T;                // row block of row i; row block and local index of row i-1
    const unsigned Bp = X.oB + (unsigned)par * 256 * BV * 16, Cp = X.oC + (unsigned)par * 256 * CV * 16;
    const unsigned bslot = Bp + (unsigned)wg * BV * 16;
    // row i-1 as it stands (before this step's update), by the 16 lanes that hold it
    if (p == pn && tr == (ln & 15) && i != drop_row) {
#pragma unroll
      for (int b = 0; b < E; b++) {
        double x = 0.0;
#pragma unroll
        for (int a = 0; a < E; a++) if (a == (ln >> 4)) x = h[a][b];
        hp_st(rs, bslot + (2 * T + tc + 16 * b) * 16, x, tag);
      }
    }
    // (the reflector's scalars of this step were computed at the end of the previous one, beside its tile update)
    const bool skip = rf.skip;
    const double nrm = rf.nrm, vii = rf.vii, inv2 = rf.inv2, scale = rf.scale;
    double vr[E], vc[E];
#pragma unroll
    for (int a = 0; a < E; a++) {
      const int j = T * p + tr + 16 * a;
      vr[a] = (!skip && j < i) ? (j == ii ? vii : s_rr[tr + 16 * a] * inv2 * scale) : 0.0;       // :51-52
    }
#pragma unroll
    for (int b = 0; b < E; b++) {
      const int j = T * q + tc + 16 * b;
      vc[b] = (!skip && j < i) ? (j == ii ? vii : s_rc[tc + 16 * b] * inv2 * scale) : 0.0;
    }
    if (t < T) {
      const int j = T * q + t;
      const double vj = (!skip && j < i) ? (j == ii ? vii : s_rc[t] * inv2 * scale) : 0.0;
      s_v[t] = vj;
      if (p == 0 && j < N && !skip) vrows[(long)(N - 1 - i) * N + j] = vj;          // reflector number N-1-i (processing order)
    }
    HP_STAMP(0)
    // ---- B: partials of x = H^T v (rows < i: v is zero beyond), y = H v (columns < i), v^T H v
    {
      double xs[E], ys[E];
#pragma unroll
      for (int b = 0; b < E; b++) xs[b] = 0.0;
#pragma unroll
      for (int a = 0; a < E; a++) {
        double y = 0.0;
#pragma unroll
        for (int b = 0; b < E; b++) { xs[b] = fma(vr[a], h[a][b], xs[b]); y = fma(h[a][b], vc[b], y); }
        ys[a] = y;
      }
      double vy = 0.0;
#pragma unroll
      for (int b = 0; b < E; b++) vy = fma(xs[b], vc[b], vy);
      vy = wave_sum(vy);
      if (lane == 0) s_vy[wave] = vy;
      // y over the 16 lanes of a lane row (each lane ends with the row tc >> (4 - LE)), x over the 4 lane rows of the wave (then over
      // the waves through LDS)
      const double ysum = hp_sum_over_tc<E>(ys, tc);
      if ((tc & ((16 >> LE) - 1)) == 0) s_yp[tr + 16 * (tc >> (4 - LE))] = ysum;
      hp_sum_over_tr<E>(xs, tr, tc, lane, s_xp[wave]);
    }
    __syncthreads();
    HP_STAMP(1)
    if (t < T) hp_st(rs, bslot + t * 16, skip ? 0.0 : (s_xp[0][t] + s_xp[1][t]) + (s_xp[2][t] + s_xp[3][t]), tag);
    else if (t < 2 * T) hp_st(rs, bslot + t * 16, skip ? 0.0 : s_yp[t - T], tag);
    if (t < HP_NREP) hp_st(rs, X.oBv + t * HP_REP_BV + (par * 256 + wg) * 16, skip ? 0.0 : (s_vy[0] + s_vy[1]) + (s_vy[2] + s_vy[3]), tag);
    // the 16 partials of this workgroup's E columns of x and E rows of y, all 256 partials of v^T y, the 16 partials of y at row
    // i-1 and this workgroup's E columns of row i-1
    {
      HpReq rq[3];
      double xv[3] = {0.0, 0.0, 0.0};
      if (t < 16 * E) rq[0] = HpReq{Bp + (unsigned)(((t / E) * 16 + q) * BV + E * p + t % E) * 16, true};
      else { const int u = t - 16 * E; rq[0] = HpReq{Bp + (unsigned)((p * 16 + u / E) * BV + T + E * q + u % E) * 16, t < 32 * E}; }
      rq[1] = HpReq{X.oBv + (unsigned)(wg % HP_NREP) * HP_REP_BV + (unsigned)(par * 256 + t) * 16, true};
      if (t < 16) rq[2] = HpReq{Bp + (unsigned)((pn * 16 + t) * BV + T + ln) * 16, true};
      else rq[2] = HpReq{Bp + (unsigned)((pn * 16 + q) * BV + 2 * T + E * p + (t - 16)) * 16, t < 16 + E};
      hp_wait<3>(rs, rq, xv, tag, dead, X.abort, status, delay >> 8);
      if (t < 32 * E) s_q[t] = xv[0];
      if (t < 16 + E) s_e[t] = xv[2];
      const double vys = wave_sum(xv[1]);
      if (lane == 0) s_red[wave] = vys;
    }
    if (dead) s_dead = 1;
    __syncthreads();
    if (s_dead) break;
    HP_STAMP(2)
    if (t < 2 * E) {
      const int c = t < E ? t : t - E;
      double s = 0.0;
#pragma unroll
      for (int pp = 0; pp < 16; pp++) s += s_q[(t < E ? 0 : 16 * E) + pp * E + c];
      const unsigned cslot = Cp + (unsigned)wg * CV * 16;
      if (t < E) {
        const double vy = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
        const double ve = s_v[E * p + c];
        const double w = s - vy * ve;                                       // w = x - (v^T y) v
        double yn = 0.0;                                                    // y at row i-1, summed as its owner sums it
#pragma unroll
        for (int pp = 0; pp < 16; pp++) yn += s_e[pp];
        const double rn = fma(-vii, w, fma(-yn, ve, s_e[16 + c]));         // row i-1 after this step's update, as the tile computes it
        const int e = T * q + E * p + c;
        double am = (e < ii - 1) ? fabs(rn) : 0.0, m = am;                  // the next step scans the columns < i-2
        if constexpr (E >= 2) m = fmax(m, xor1(m));
        if constexpr (E >= 4) m = fmax(m, xor2(m));
        if constexpr (E >= 8) m = fmax(m, xor4(m));
        const double x = m > 0.0 ? am * fast_rcp(m) : 0.0;
        double ss = x * x;
        if constexpr (E >= 2) ss += xor1(ss);
        if constexpr (E >= 4) ss += xor2(ss);
        if constexpr (E >= 8) ss += xor4(ss);
        hp_st(rs, cslot + c * 16, w, tag);
        hp_st(rs, cslot + (2 * E + c) * 16, rn, tag);
        // (the E lanes all hold the reduced m and ss: lane c writes the copies c, c + E, ...)
        for (int r = c; r < (int)HP_NREP; r += E) { hp_st(rs, X.oCn + r * HP_REP_CN + (par * 512 + 2 * wg) * 16, m, tag); hp_st(rs, X.oCn + r * HP_REP_CN + (par * 512 + 2 * wg + 1) * 16, ss, tag); }
      } else {
        hp_st(rs, cslot + (E + c) * 16, (T * p + E * q + c < i) ? s : 0.0, tag);      // y, rows < i only
      }
    }
    // ---- C: w over the columns q, y over the rows p, the next row over both, the norm partials, the entry (i-1, i-2)
    {
      HpReq rq[5];
      double xv[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
      if (t < T) {
        const unsigned s = Cp + (unsigned)(((t / E) * 16 + q) * CV) * 16;
        rq[0] = HpReq{s + (t % E) * 16, true}; rq[1] = HpReq{s + (2 * E + t % E) * 16, true};
      } else {
        const int j = t - T;
        rq[0] = HpReq{Cp + (unsigned)((p * 16 + j / E) * CV + E + j % E) * 16, t < 2 * T};
        rq[1] = HpReq{Cp + (unsigned)(((j / E) * 16 + p) * CV + 2 * E + j % E) * 16, t < 2 * T};
      }
      rq[2] = HpReq{X.oCn + (unsigned)(wg % HP_NREP) * HP_REP_CN + (unsigned)(par * 512 + 2 * t) * 16, true};
      rq[3] = HpReq{X.oCn + (unsigned)(wg % HP_NREP) * HP_REP_CN + (unsigned)(par * 512 + 2 * t + 1) * 16, true};
      { const int e = ii - 1 < 0 ? 0 : ii - 1, j = e % T; rq[4] = HpReq{Cp + (unsigned)(((j / E) * 16 + e / T) * CV + 2 * E + j % E) * 16, t == 0}; }
      hp_wait<5>(rs, rq, xv, tag, dead, X.abort, status, delay & 255);
      if (t < T) { s_w[t] = xv[0]; s_rc[t] = xv[1]; }
      else if (t < 2 * T) { s_y[t - T] = xv[0]; s_rr[t - T] = xv[1]; }
      if (t == 0) s_h0 = xv[4];
      const double wm = wave_max(xv[2]);
      const double f = wm > 0.0 ? xv[2] * fast_rcp(wm) : 0.0;
      const double wS = wave_sum(xv[3] * f * f);
      if (lane == 0) { s_wm[wave] = wm; s_wS[wave] = wS; }
    }
    if (dead) s_dead = 1;
    __syncthreads();
    if (s_dead) break;
    HP_STAMP(3)
    const Refl rfn = reflector();                                            // the next step's, from what round C brought
    {                                                                        // (a skipped step published zeros: the update is exact then)
#pragma unroll
      for (int a = 0; a < E; a++) {
        const double ya = s_y[tr + 16 * a];
#pragma unroll
        for (int b = 0; b < E; b++) h[a][b] = fma(-vr[a], s_w[tc + 16 * b], fma(-ya, vc[b], h[a][b]));
      }
      if (!skip && p == pi) {                                              // the finished row (:83-84)
#pragma unroll
        for (int a = 0; a < E; a++)
#pragma unroll
          for (int b = 0; b < E; b++) {
            const int r = T * p + tr + 16 * a, c = T * q + tc + 16 * b;
            if (r == i) { if (c < ii) h[a][b] = 0.0; else if (c == ii) h[a][b] = nrm; }
          }
      }
    }
    rf = rfn;
    HP_STAMP(4)
  }
#undef HP_STAMP
  if (stamps && t == 0) for (int k = 0; k < 5; k++) stamps[wg * 8 + k] = tacc[k];
#pragma unroll
  for (int a = 0; a < E; a++)
#pragma unroll
    for (int b = 0; b < E; b++) {
      const int r = T * p + tr + 16 * a, c = T * q + tc + 16 * b;
      if (r < N && c < N) H[(long)r * N + c] = h[a][b];
    }
}

}  // namespace

// A [batch, N, N] -> U, H [batch, N, N]
int nd4_gehrd(nd4hip_handle* h, int64_t batch64, int64_t N64, const double* A, double* U, double* H) {
  ND4_CHECK_ARG(N64 < 32768 && batch64 < 65536, "nd4_gehrd: extent out of range");
  const int N = (int)N64, batch = (int)batch64;
  if (N == 0 || batch == 0) return 0;
  const size_t nn = (size_t)N * N;
  if (H != A) ND4_HIP(hipMemcpyAsync(H, A, sizeof(double) * nn * batch, hipMemcpyDeviceToDevice, h->stream));
  const unsigned gy = (unsigned)(N < 1024 ? N : 1024);
  hipLaunchKernelGGL(hess_init_u, dim3((unsigned)((N + 255) / 256), gy, (unsigned)batch), dim3(256), 0, h->stream, U, N);
  ND4_HIP(hipGetLastError());
  if (N <= 2) return 0;
  const int Pmax = (N + HR - 1) / HR;
  Nd4WsScope scope(h);
  void* p = nullptr;
  // one big matrix: U is not dragged through every step (its two passes are half of the traffic) but formed at the end
  // from the stored reflectors, U = I - V T V^T (nd4_wy_form: Gram + level-wise T + two GEMMs; tau = 1 since v^T v = 2)
  const bool wy = batch <= 4 && N >= 256;
  const int nstore = wy ? ((N + 15) / 16) * 16 : 0;
  ND4_TRY(nd4_ws_alloc(h, sizeof(double) * ((size_t)batch * ((size_t)4 * N + (size_t)Pmax * N + (size_t)N * nstore) + (size_t)nstore + 16) +
                              sizeof(int) * (size_t)batch + 64, &p));
  HessWs ws;
  ws.v = static_cast<double*>(p); ws.y = ws.v + (size_t)batch * N; ws.yu = ws.y + (size_t)batch * N; ws.w = ws.yu + (size_t)batch * N;
  ws.xpart = ws.w + (size_t)batch * N;
  ws.vstore = wy ? ws.xpart + (size_t)batch * Pmax * N : nullptr;
  ws.nstore = nstore;
  double* ones = ws.xpart + (size_t)batch * Pmax * N + (size_t)batch * N * nstore;
  ws.skip = reinterpret_cast<int*>(ones + nstore + 16);
  ws.sV = N; ws.sX = (long)Pmax * N;
  if (wy) {
    ND4_HIP(hipMemsetAsync(ws.vstore, 0, sizeof(double) * (size_t)batch * N * nstore, h->stream));
    hipLaunchKernelGGL(hess_fill, dim3((unsigned)((nstore + 255) / 256)), dim3(256), 0, h->stream, ones, nstore, 1.0);
  }
  const int cchunks = (N + 255) / 256;
  const bool blocked_path = wy && batch == 1 && N >= 512 && (N & 1) == 0 && (size_t)N * sizeof(double) <= 48 * 1024 && !getenv("ND4HIP_HESS_UNBLOCKED");
  const bool no_persist = getenv("ND4HIP_HESS_NO_PERSIST") != nullptr;            // (read per call: the tests switch between the paths)
  // (the 256 workgroups of hessp must all be resident at once: one or two per CU, see xchg.h)
  bool persist = wy && batch == 1 && N >= 128 && N <= 2048 && !no_persist && !getenv("ND4HIP_HESS_UNBLOCKED");
  if (persist) {
    int per_cu = 0;
    const hipError_t oe = N <= 512 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, hessp<2>, 256, 0)
                        : N <= 1024 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, hessp<4>, 256, 0)
                                    : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, hessp<8>, 256, 0);
    persist = oe == hipSuccess && (long)per_cu * h->num_cu >= 256;
  }
  if (persist) {
    // ---- one launch: 16 x 16 workgroups keep H in registers for the whole reduction (see hessp) ----
    const int E = N <= 512 ? 2 : (N <= 1024 ? 4 : 8), T = 16 * E;
    const size_t BV = 3 * T, CV = 3 * E;                                                    // values (16 bytes each) per slot
    const size_t xwords = 2 * (2 * 256 * (BV + CV)) + (size_t)HP_NREP * (HP_REP_BV + HP_REP_CN) / 8;
    void* q = nullptr;
    ND4_TRY(nd4_ws_alloc(h, sizeof(double) * (size_t)nstore * N + sizeof(qx_u64) * xwords + 64, &q));
    double* vrows = static_cast<double*>(q);
    HessPx X;
    X.base = reinterpret_cast<qx_u64*>(vrows + (size_t)nstore * N);
    X.oB = 0; X.oC = (unsigned)(2 * 256 * BV * 16); X.oBv = X.oC + (unsigned)(2 * 256 * CV * 16); X.oCn = X.oBv + HP_NREP * HP_REP_BV;
    X.bytes = X.oCn + HP_NREP * HP_REP_CN;
    X.abort = reinterpret_cast<int*>(X.base + xwords);
    ND4_HIP(hipMemsetAsync(vrows, 0, sizeof(double) * (size_t)nstore * N + sizeof(qx_u64) * xwords + 64, h->stream));
    const int drop = nd4_test_drop_panel() >= 0 ? N - 1 - nd4_test_drop_panel() : -1;       // test hook: row N-1-k is never published
    static const bool want_stamps = getenv("ND4HIP_HESSP_STAMPS") != nullptr;
    // s_sleep(8) units (~0.22 us) before the first look of round B (high byte) and round C (low byte)
    const int delay = getenv("ND4HIP_HESSP_DELAY") ? atoi(getenv("ND4HIP_HESSP_DELAY")) : (2 << 8 | 4);
    long long* stamps = nullptr;
    if (want_stamps) { void* sp = nullptr; ND4_TRY(nd4_ws_alloc(h, sizeof(long long) * 256 * 8, &sp)); stamps = static_cast<long long*>(sp); }
    if (E == 2) hipLaunchKernelGGL(hessp<2>, dim3(256), dim3(256), 0, h->stream, H, N, vrows, X, h->xstat, drop, stamps, delay);
    else if (E == 4) hipLaunchKernelGGL(hessp<4>, dim3(256), dim3(256), 0, h->stream, H, N, vrows, X, h->xstat, drop, stamps, delay);
    else hipLaunchKernelGGL(hessp<8>, dim3(256), dim3(256), 0, h->stream, H, N, vrows, X, h->xstat, drop, stamps, delay);
    ND4_HIP(hipGetLastError());
    if (stamps) {                                     // per step, in us: reflector | products | B | sums + C | update
      long long hs[256 * 8];
      ND4_HIP(hipMemcpyAsync(hs, stamps, sizeof(hs), hipMemcpyDeviceToHost, h->stream));
      ND4_HIP(hipStreamSynchronize(h->stream));
      for (int g : {0, 17, 119, 255})
        fprintf(stderr, "hessp N=%d wg %3d: scalars+v %.2f  matvec %.2f  B %.2f  C %.2f  update %.2f us per step\n", N, g,
                hs[g * 8] * 0.01 / (N - 2), hs[g * 8 + 1] * 0.01 / (N - 2), hs[g * 8 + 2] * 0.01 / (N - 2), hs[g * 8 + 3] * 0.01 / (N - 2), hs[g * 8 + 4] * 0.01 / (N - 2));
    }
    ND4_TRY(nd4_transpose(h, nstore, N, vrows, N, ws.vstore, nstore, 1, 0, 0));            // reflectors as columns for nd4_wy_form
  } else if (blocked_path) {
    // ---- blocked: NBH steps per block, H untouched inside a block, two GEMMs per block ----
    const int nchunks = (N + BC - 1) / BC;
    void* q = nullptr;
    ND4_TRY(nd4_ws_alloc(h, sizeof(double) * ((size_t)3 * N * NBH + 3 * NBH + (size_t)N + (size_t)nchunks * N + (size_t)((N + BR - 1) / BR) * nchunks + (size_t)nstore * N + (size_t)N) +
                                sizeof(int) * (size_t)N + 64, &q));
    HessBlk bk;
    bk.Vt = static_cast<double*>(q); bk.Wt = bk.Vt + (size_t)N * NBH; bk.Yt = bk.Wt + (size_t)N * NBH;
    bk.dots = bk.Yt + (size_t)N * NBH; bk.nrm = bk.dots + 3 * NBH; bk.ypart = bk.nrm + N; bk.vyp = bk.ypart + (size_t)nchunks * N;
    bk.vrows = bk.vyp + (size_t)((N + BR - 1) / BR) * nchunks;
    bk.nextrow = bk.vrows + (size_t)nstore * N;
    bk.skipv = reinterpret_cast<int*>(bk.nextrow + N);
    ND4_HIP(hipMemsetAsync(bk.skipv, 0, sizeof(int) * (size_t)N, h->stream));
    ND4_HIP(hipMemsetAsync(bk.vrows, 0, sizeof(double) * (size_t)nstore * N, h->stream));
    for (int ihi = N - 1; ihi > 1; ihi -= NBH) {
      const int ilo = ihi - NBH + 1 > 2 ? ihi - NBH + 1 : 2;
      ND4_HIP(hipMemsetAsync(bk.Vt, 0, sizeof(double) * (size_t)3 * N * NBH, h->stream));
      for (int i = ihi, k = 0; i >= ilo; i--, k++) {
        const int ngroups = (i + BR - 1) / BR, ndot = (3 * k + 3) / 4;
        // two launches per step: the pass builds the reflector itself (every workgroup, from the row hessb_reduce left behind)
        hipLaunchKernelGGL(hessb_pass, dim3((unsigned)(ngroups * nchunks + ndot)), dim3(256), (size_t)N * sizeof(double), h->stream, H, N, i, k, ngroups, nchunks, ws, bk);
        hipLaunchKernelGGL(hessb_reduce, dim3((unsigned)((N + RC - 1) / RC)), dim3(256), 0, h->stream, H, N, i, k, nchunks, i > ilo ? 1 : 0, ws, bk);
      }
      ND4_HIP(hipGetLastError());
      const int nk = ihi - ilo + 1;
      // H[0:ihi, :] -= V W^T + Y V^T   (rows >= ihi: V and Y are zero there); V^T etc. are stored [NBH][N]: transA
      ND4_TRY(nd4_gemm(h, true, false, ihi, N, nk, -1.0, bk.Vt, N, 0, bk.Wt, N, 0, 1.0, H, N, 0, 1));
      ND4_TRY(nd4_gemm(h, true, false, ihi, N, nk, -1.0, bk.Yt, N, 0, bk.Vt, N, 0, 1.0, H, N, 0, 1));
      hipLaunchKernelGGL(hessb_fix, dim3((unsigned)((N + 255) / 256), (unsigned)nk), dim3(256), 0, h->stream, H, N, ilo, ihi, bk);
    }
    ND4_HIP(hipGetLastError());
    ND4_TRY(nd4_transpose(h, nstore, N, bk.vrows, N, ws.vstore, nstore, 1, 0, 0));     // reflectors as columns for nd4_wy_form
  } else
  for (int i = N - 1; i > 1; i--) {
    hipLaunchKernelGGL(hess_vec, dim3((unsigned)batch), dim3(256), 0, h->stream, H, N, i, ws);
    const unsigned rg = (unsigned)((N - 1 + HR - 1) / HR);           // row groups: enough for U's N-1 rows (H uses i <= N-1)
    hipLaunchKernelGGL(hess_pass_a, dim3(rg, wy ? 1 : 2, (unsigned)batch), dim3(256), 0, h->stream, H, U, N, i, ws);
    hipLaunchKernelGGL(hess_reduce, dim3((unsigned)((N + RC - 1) / RC), (unsigned)batch), dim3(256), 0, h->stream, N, i, ws);
    hipLaunchKernelGGL(hess_pass_b, dim3((unsigned)((wy ? 1 : 2) * cchunks), rg, (unsigned)batch), dim3(256), 0, h->stream, H, U, N, i, cchunks, ws);
    ND4_HIP(hipGetLastError());
  }
  if (wy)
    for (int m = 0; m < batch; m++)                                   // skipped steps stored v = 0: they drop out of V T V^T
      ND4_TRY(nd4_wy_form(h, N, nstore, ws.vstore + (size_t)m * N * nstore, ones, 1, U + (size_t)m * nn, N));
  return 0;
}
