// Batched bidiagonalisation on the device.
//
// Replaces src/la/bidiag.js:245-319 (bidiag_decomp; kernels _bidiag_decomp_vert :32-110, _square :113-161, _horiz
// :164-242): A [M,N] = U B V with B [I,J] upper bidiagonal, U [M,I] with orthonormal columns, V [J,N] with orthonormal
// rows, I = min(M,N), J = I for M >= N and I+1 for M < N.
//
// The reference eliminates column i with plane rotations and the tail of row i with one Householder reflector. The
// factorisation is unique up to the signs of the columns of U / rows of V, and those signs are pinned as follows:
//   * right side: the very same reflector convention as the reference (FrobeniusNorm scaling, B[i,i+1] = -sign(x) |x..|,
//     bidiag.js:66-76): the reflector does not depend on the sign of the row it is built from, so V agrees whatever the
//     left side did;
//   * left side: Householder reflectors (LAPACK-style), then the reference's Givens sign convention is imposed on
//     (U, rows of B) afterwards by nd4_givens_signs: B[i,i] >= 0 and det U = +1 for M <= N (rotations without
//     normalisation, bidiag.js:123-136 / :183-195), positive leading minors of U's top block for M > N (the c >= 0
//     normalisation of bidiag.js:49-61 preserves the sign of the pivot, exactly like qr.js:111-115).
// Unblocked (BLAS-2): every step is two matrix-vector products and two rank-1 updates on the trailing block, bound by HBM;
// U and V are formed afterwards from the stored reflectors: at once from their compact-WY form (nd4_wy_form: one big matrix)
// or by applying them backwards to the identity (batches of small matrices).
#include "nd4hip_internal.h"
#include "xchg16.h"
#include <cstdio>
#include <cstdlib>

namespace {

constexpr int GR = 8;             // rows per workgroup of the weighted column sum (2 per wave): many small workgroups

__device__ __forceinline__ double blk_max(double v, double* s_red) {
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off));
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
  __syncthreads();
  v = fmax(fmax(s_red[0], s_red[1]), fmax(s_red[2], s_red[3]));
  __syncthreads();
  return v;
}
__device__ __forceinline__ double blk_sum(double v, double* s_red) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
  __syncthreads();
  v = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
  __syncthreads();
  return v;
}

// ---- reflector generation ----
// left: column i of W, rows i..M-1 -> u (u_i = 1) into column i of UL (ld K), tau, W[i,i] = beta, zeros below.
// LAPACK dlarfg with a max-scaled norm: H = I - tau u u^T maps the column to beta e_1.
__global__ __launch_bounds__(256) void bd_vec_col(double* __restrict__ Wm, int M, int N, int i, double* __restrict__ ULm, int K,
                                                   double* __restrict__ tauLm) {
  __shared__ double s_red[4];
  double* W = Wm + (long)blockIdx.x * M * N;
  double* UL = ULm + (long)blockIdx.x * M * K;
  const int t = threadIdx.x;
  double mx = 0.0;
  for (int r = i + 1 + t; r < M; r += 256) mx = fmax(mx, fabs(W[(long)r * N + i]));
  mx = blk_max(mx, s_red);
  for (int r = t; r < i; r += 256) UL[(long)r * K + i] = 0.0;
  if (mx == 0.0) {                                                  // nothing below the diagonal: H = I
    for (int r = i + 1 + t; r < M; r += 256) UL[(long)r * K + i] = 0.0;
    if (t == 0) { UL[(long)i * K + i] = 1.0; tauLm[(long)blockIdx.x * K + i] = 0.0; }
    return;
  }
  const double alpha = W[(long)i * N + i];
  const double sc = fmax(mx, fabs(alpha));
  double ss = 0.0;
  for (int r = i + 1 + t; r < M; r += 256) { const double x = W[(long)r * N + i] / sc; ss += x * x; }
  ss = blk_sum(ss, s_red);
  const double a1 = alpha / sc;
  const double nrm = sqrt(ss + a1 * a1) * sc;
  const double beta = alpha > 0 ? -nrm : nrm;
  const double tau = (beta - alpha) / beta;
  const double inv = 1.0 / (alpha - beta);
  for (int r = i + 1 + t; r < M; r += 256) { UL[(long)r * K + i] = W[(long)r * N + i] * inv; W[(long)r * N + i] = 0.0; }
  if (t == 0) { UL[(long)i * K + i] = 1.0; W[(long)i * N + i] = beta; tauLm[(long)blockIdx.x * K + i] = tau; }
}

// right: row i of W, columns first = i+1 .. N-1 -> unit vector v into row i of VR (zeros elsewhere), flag, finished row.
// The reference's convention (bidiag.js:66-76): norm = -sign(x_first) |x|, v = (x - norm e) / max / sqrt(sum), H = I - 2 v v^T.
__global__ __launch_bounds__(256) void bd_vec_row(double* __restrict__ Wm, int M, int N, int i, double* __restrict__ VRm, int K,
                                                   int* __restrict__ flagRm) {
  __shared__ double s_red[4];
  double* row = Wm + (long)blockIdx.x * M * N + (long)i * N;
  double* v = VRm + (long)blockIdx.x * K * N + (long)i * N;
  const int t = threadIdx.x, first = i + 1;
  double m1 = 0.0;
  for (int j = first + 1 + t; j < N; j += 256) m1 = fmax(m1, fabs(row[j]));
  m1 = blk_max(m1, s_red);
  if (m1 == 0.0) {                                                  // NORM.max === 0 -> continue
    for (int j = t; j < N; j += 256) v[j] = 0.0;
    if (t == 0) flagRm[(long)blockIdx.x * K + i] = 0;
    return;
  }
  const double x0 = row[first];
  const double mx = fmax(m1, fabs(x0));
  double ss = 0.0;
  for (int j = first + t; j < N; j += 256) { const double x = row[j] / mx; ss += x * x; }
  ss = blk_sum(ss, s_red);
  const double nrm = (isfinite(mx) ? sqrt(ss) * mx : mx) * (x0 > 0 ? -1.0 : 1.0);
  const double head = x0 - nrm;
  const double mx2 = fmax(m1, fabs(head));
  double s2 = 0.0;
  for (int j = first + 1 + t; j < N; j += 256) { const double x = row[j] / mx2; s2 += x * x; }
  s2 = blk_sum(s2, s_red);
  { const double x = head / mx2; s2 += x * x; }
  const double div = sqrt(s2);
  for (int j = t; j < N; j += 256) {
    double vj = 0.0;
    if (j >= first) vj = (j == first ? head : row[j]) / mx2 / div;
    v[j] = vj;
  }
  __syncthreads();
  for (int j = first + 1 + t; j < N; j += 256) row[j] = 0.0;
  if (t == 0) { row[first] = nrm; flagRm[(long)blockIdx.x * K + i] = 1; }
}

// ---- the three BLAS-2 building blocks on a sub-block X[r0:r1, c0:c1] (row-major, ld) ----
struct Blk { double* X; long ld, sX; int r0, r1, c0, c1; };

// zpart[p][c] = sum over the GR rows of group p of a[r] * X[r,c]      (a = column `acol` of A2, ld lda)
__global__ __launch_bounds__(256) void bd_colsum(Blk b, const double* __restrict__ Am, long lda, long sA, int acol,
                                                  const double* __restrict__ gate, long sGate, int gidx, double* __restrict__ zpart, long sZ, int ncols_total) {
  const long m = blockIdx.z;
  if (gate && gate[m * sGate + gidx] == 0.0) return;
  const double* X = b.X + m * b.sX;
  const double* a = Am + m * sA + acol;
  const int rg = b.r0 + blockIdx.y * GR;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  __shared__ double s_x[4][64];
  const int c = b.c0 + blockIdx.x * 64 + lane;
  double acc = 0.0;
  if (c < b.c1) {
#pragma unroll
    for (int q = 0; q < GR / 4; q++) {
      const int r = rg + wave * (GR / 4) + q;
      if (r < b.r1) acc += a[(long)r * lda] * X[(long)r * b.ld + c];
    }
  }
  s_x[wave][lane] = acc;
  __syncthreads();
  if (wave == 0 && c < b.c1) zpart[m * sZ + (long)blockIdx.y * ncols_total + c] = (s_x[0][lane] + s_x[1][lane]) + (s_x[2][lane] + s_x[3][lane]);
}
// z[c] = scale * sum_p zpart[p][c], scale = tau[gidx] (or 1); 32 columns per workgroup, 8 thread groups per column
__global__ __launch_bounds__(256) void bd_colsum_reduce(int c0, int c1, int P, const double* __restrict__ zpart, long sZ, int ncols_total,
                                                         const double* __restrict__ gate, long sGate, int gidx, double* __restrict__ z, long sz) {
  const long m = blockIdx.y;
  double scale = 1.0;
  if (gate) { scale = gate[m * sGate + gidx]; if (scale == 0.0) return; }
  __shared__ double s_part[8][32];
  const int cc = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int c = c0 + blockIdx.x * 32 + cc;
  double x = 0.0;
  if (c < c1)
    for (int p = g; p < P; p += 8) x += zpart[m * sZ + (long)p * ncols_total + c];
  s_part[g][cc] = x;
  __syncthreads();
  if (g == 0 && c < c1) {
    double s = 0.0;
#pragma unroll
    for (int q = 0; q < 8; q++) s += s_part[q][cc];
    z[m * sz + c] = scale * s;
  }
}
// y[r] = scale * sum_c X[r,c] * bvec[c]; one wave per row, 4 rows per workgroup
__global__ __launch_bounds__(256) void bd_rowdot(Blk b, const double* __restrict__ bm, long sB, double scale, const int* __restrict__ flag, long sF, int fidx,
                                                  double* __restrict__ y, long sy) {
  const long m = blockIdx.y;
  if (flag && !flag[m * sF + fidx]) return;
  const double* X = b.X + m * b.sX;
  const double* bv = bm + m * sB;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = b.r0 + blockIdx.x * 4 + wave;
  if (r >= b.r1) return;
  double acc = 0.0;
  for (int c = b.c0 + lane; c < b.c1; c += 64) acc += X[(long)r * b.ld + c] * bv[c];
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
  if (lane == 0) y[m * sy + r] = scale * acc;
}
// X[r,c] -= p[r] * q[c]   (p: stride ldp, e.g. a column of UL; q contiguous)
__global__ __launch_bounds__(256) void bd_rank1(Blk b, const double* __restrict__ pm, long ldp, long sP, const double* __restrict__ qm, long sQ,
                                                 const double* __restrict__ gate, long sGate, const int* __restrict__ flag, long sF, int gidx) {
  const long m = blockIdx.z;
  if (gate && gate[m * sGate + gidx] == 0.0) return;
  if (flag && !flag[m * sF + gidx]) return;
  double* X = b.X + m * b.sX;
  const double* p = pm + m * sP;
  const double* q = qm + m * sQ;
  const int c = b.c0 + blockIdx.x * 256 + threadIdx.x;
  if (c >= b.c1) return;
  const double qc = q[c];
  const int rg = b.r0 + blockIdx.y * 16;
#pragma unroll 4
  for (int r = rg; r < rg + 16 && r < b.r1; r++) X[(long)r * b.ld + c] -= p[(long)r * ldp] * qc;
}

__global__ void bd_fill(double* __restrict__ x, int n, double val) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] = val;
}
__global__ void bd_set_identity(double* __restrict__ Xm, int rows, int cols) {
  const long base = (long)blockIdx.z * rows * cols;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= cols) return;
  for (int r = blockIdx.y; r < rows; r += gridDim.y) Xm[base + (long)r * cols + j] = (r == j) ? 1.0 : 0.0;
}
// B [K, J] <- diagonal and super-diagonal of W [M, N], zeros elsewhere
__global__ void bd_extract(const double* __restrict__ Wm, int M, int N, double* __restrict__ Bm, int K, int J) {
  const double* W = Wm + (long)blockIdx.z * M * N;
  double* B = Bm + (long)blockIdx.z * K * J;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= J) return;
  for (int i = blockIdx.y; i < K; i += gridDim.y) B[(long)i * J + j] = (j == i || j == i + 1) ? W[(long)i * N + j] : 0.0;
}

// ================================================================ fused form (one large matrix): 3 launches per step ==========
// With u, tau the left reflector of step i and v the right one, the step is
//     z = tau A^T u          (A = rows i.., columns i+1..)        column-local: bd2_colpass, one workgroup per 16 columns, z COMPLETE
//     row i:  r = A[i,:] - z  -> v (reference convention), B[i,i+1]                     every workgroup of bd2_rowpass, redundantly
//     y = 2 (A - u z^T) v = 2 (A v - u (z.v))                      row-local: bd2_rowpass, one wave per row
//     A -= u z^T + y v^T                                           applied LAZILY by the next step's bd2_colpass, in the same pass
//                                                                  that forms the next z (column-local: 16 columns x all rows)
// and the left reflector of step i+1 is built in the prologue of that bd2_colpass by every workgroup from column i+1 as it will be
// after the update, which the waves of bd2_rowpass leave behind in ucol (each has y_r). Neither z nor y needs a partial-sum launch,
// the two reflector kernels and the update launch are gone: 2 dependent launches per step instead of 7, and 1 read + 1
// read-modify-write of the trailing block instead of 2 + 2.
constexpr int BD2_MAXM = 4096, BD2_MAXN = 3072;      // what the LDS copies of (u, u', y') and of row i are sized for

constexpr int CPT = 1024;          // threads of bd2_colpass: 16 columns x 64 row lanes (with 256 the 128 workgroups were latency-bound: 8 loads per thread in flight)
template <int NT> __device__ __forceinline__ double blk_max_n(double v, double* s_red) {
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off));
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
  __syncthreads();
  v = s_red[0];
#pragma unroll
  for (int w = 1; w < NT / 64; w++) v = fmax(v, s_red[w]);
  __syncthreads();
  return v;
}
template <int NT> __device__ __forceinline__ double blk_sum_n(double v, double* s_red) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
  __syncthreads();
  v = 0.0;
#pragma unroll
  for (int w = 0; w < NT / 64; w++) v += s_red[w];
  __syncthreads();
  return v;
}

__global__ void bd2_init_ucol(const double* __restrict__ W, int M, int N, double* __restrict__ ucol) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < M) ucol[r] = W[(long)r * N];
}

// left reflector of step i (bd_vec_col's formulas) by every workgroup; workgroup 0 stores it. Then, for 16 columns and in ONE pass
// over the rows i..: the pending rank-2 update of step i-1 (A -= u' z'^T + y' v'^T, written back) and z of step i from the updated
// values. The update of a step thus never has a launch of its own: 2 launches per step.
__global__ __launch_bounds__(CPT) void bd2_colpass(double* __restrict__ W, int M, int N, int i, double* __restrict__ UL, int K,
                                                    double* __restrict__ tauL, const double* __restrict__ ucol,
                                                    const double* __restrict__ zprev, int Pprev, double* __restrict__ znew,
                                                    int prev, const double* __restrict__ uprev, double* __restrict__ unew, const double* __restrict__ y,
                                                    const double* __restrict__ VR, const double* __restrict__ rowfin) {
  // grid (column blocks of 16, row parts): with all rows in one workgroup only N/16 <= 128 workgroups stream the trailing block;
  // the parts' z go to znew[part][N] and are added (fixed order) by their readers. zprev / znew alternate by step.
  extern __shared__ double s_u[];                        // [3 rpp]: this part's rows of u (u_0 = 1), then of u' and y' (pending update)
  __shared__ double s_red[CPT / 64];
  __shared__ double s_zc[CPT / 16][17];
  const int t = threadIdx.x, m = M - i;
  const int P = gridDim.y, rpp = (((m + P - 1) / P) + 15) & ~15;
  const int ra = blockIdx.y * rpp, rb = (ra + rpp < m) ? ra + rpp : m;       // this part's rows [ra, rb) relative to row i
  double* s_up = s_u + rpp; double* s_yp = s_up + rpp;
  const bool pend = prev >= 0, lead = blockIdx.x == 0 && blockIdx.y == 0;
  if (pend) {
    for (int r = ra + t; r < rb; r += CPT) { s_up[r - ra] = uprev[i + r]; s_yp[r - ra] = y[i + r]; }
    if (lead)                                             // row i-1, finished by bd2_rowpass (rows >= i are all this launch reads)
      for (int c = i + t; c < N; c += CPT) W[(long)prev * N + c] = rowfin[c];
  }
  double mx = 0.0;
  for (int r = 1 + t; r < m; r += CPT) mx = fmax(mx, fabs(ucol[i + r]));
  mx = blk_max_n<CPT>(mx, s_red);
  double tau = 0.0;
  if (mx == 0.0) {                                       // nothing below the diagonal: H = I, z = 0
    for (int r = ra + t; r < rb; r += CPT) s_u[r - ra] = (r == 0) ? 1.0 : 0.0;
    if (lead) {
      for (int r = t; r < M; r += CPT) { UL[(long)r * K + i] = (r == i) ? 1.0 : 0.0; unew[r] = (r == i) ? 1.0 : 0.0; }
      // column i in memory still lacks the pending update: its true values are ucol (zeros below the diagonal here)
      for (int r = 1 + t; r < m; r += CPT) W[(long)(i + r) * N + i] = 0.0;
      if (t == 0) { tauL[i] = 0.0; W[(long)i * N + i] = ucol[i]; }
    }
  } else {
    const double alpha = ucol[i];
    const double sc = fmax(mx, fabs(alpha));
    double ss = 0.0;
    for (int r = 1 + t; r < m; r += CPT) { const double x = ucol[i + r] / sc; ss += x * x; }
    ss = blk_sum_n<CPT>(ss, s_red);
    const double a1 = alpha / sc;
    const double nrm = sqrt(ss + a1 * a1) * sc;
    const double beta = alpha > 0 ? -nrm : nrm;
    tau = (beta - alpha) / beta;
    const double inv = 1.0 / (alpha - beta);
    for (int r = ra + t; r < rb; r += CPT) s_u[r - ra] = (r == 0) ? 1.0 : ucol[i + r] * inv;
    if (lead) {                                           // (unew: contiguous copy of u for bd2_rowpass and the next pending update)
      for (int r = t; r < i; r += CPT) { UL[(long)r * K + i] = 0.0; unew[r] = 0.0; }
      for (int r = 1 + t; r < m; r += CPT) { const double ur = ucol[i + r] * inv; UL[(long)(i + r) * K + i] = ur; unew[i + r] = ur; W[(long)(i + r) * N + i] = 0.0; }
      if (t == 0) { UL[(long)i * K + i] = 1.0; unew[i] = 1.0; W[(long)i * N + i] = beta; tauL[i] = tau; }
    }
  }
  __syncthreads();
  const int cx = t & 15, ry = t >> 4;                    // 16 columns x CPT/16 row lanes: a wave reads 4 whole 128-byte row segments
  const int c = i + 1 + blockIdx.x * 16 + cx;
  double acc = 0.0;
  if (c < N) {
    double* col = W + (long)(i + ra) * N + c;
    double zc = 0.0;
    if (pend) for (int q = 0; q < Pprev; q++) zc += zprev[(long)q * N + c];
    const double vc = pend ? VR[(long)prev * N + c] : 0.0;
    const int m = rb - ra;                               // (rows of this part from here on)
    constexpr int RL = CPT / 16, UB = 16;
    int r = ry;
    for (; r + (UB - 1) * RL < m; r += UB * RL) {          // UB loads in flight per thread
      double x[UB];
#pragma unroll
      for (int q = 0; q < UB; q++) x[q] = col[(long)(r + q * RL) * N];
      if (pend) {
#pragma unroll
        for (int q = 0; q < UB; q++) { x[q] -= s_up[r + q * RL] * zc + s_yp[r + q * RL] * vc; col[(long)(r + q * RL) * N] = x[q]; }
      }
#pragma unroll
      for (int q = 0; q < UB; q++) acc += s_u[r + q * RL] * x[q];
    }
    for (; r < m; r += RL) {
      double x = col[(long)r * N];
      if (pend) { x -= s_up[r] * zc + s_yp[r] * vc; col[(long)r * N] = x; }
      acc += s_u[r] * x;
    }
  }
  s_zc[ry][cx] = acc;
  __syncthreads();
  if (t < 16) {
    double sum = 0.0;
#pragma unroll
    for (int q = 0; q < CPT / 16; q++) sum += s_zc[q][t];      // fixed order
    const int cc = i + 1 + blockIdx.x * 16 + t;
    if (cc < N) znew[(long)blockIdx.y * N + cc] = tau * sum;
  }
}

// right reflector of step i (bd_vec_row's formulas) by every workgroup from r = A[i,:] - z; workgroup 0 stores it and finishes
// row i. Then y for 4 rows (one wave each). has_right = 0: the step has no right reflector (i + 1 >= N - 1): row i only.
__global__ __launch_bounds__(256) void bd2_rowpass(double* __restrict__ W, int M, int N, int i, const double* __restrict__ UL, int K,
                                                    const double* __restrict__ zp, int P, double* __restrict__ VR, int* __restrict__ flagR,
                                                    double* __restrict__ y, int has_right, double* __restrict__ rowfin,
                                                    const double* __restrict__ ucur, double* __restrict__ ucol) {
  extern __shared__ double s_v[];                        // [N - first]: r, then v; [N - first]: z
  __shared__ double s_red[4];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, first = i + 1, n = N - first;
  double* s_z = s_v + n;
  const double* rowi = W + (long)i * N;                 // read by every workgroup: its final form goes to rowfin, bd2_update stores it
  const int r = first + blockIdx.x * 4 + wave;           // this wave's row
  double m1 = 0.0;
  for (int j0 = t; j0 < n; j0 += 8 * 256) {             // eight entries per thread with all their loads in flight (a rolled loop of
    double zj[8], xr[8];                                 // dependent global loads costs a memory round trip per entry)
#pragma unroll
    for (int e = 0; e < 8; e++) { zj[e] = 0.0; xr[e] = (j0 + 256 * e < n) ? rowi[first + j0 + 256 * e] : 0.0; }
    for (int q = 0; q < P; q++) {                        // the row parts of bd2_colpass, fixed order
#pragma unroll
      for (int e = 0; e < 8; e++) if (j0 + 256 * e < n) zj[e] += zp[(long)q * N + first + j0 + 256 * e];
    }
#pragma unroll
    for (int e = 0; e < 8; e++) {
      const int j = j0 + 256 * e;
      if (j < n) {
        s_z[j] = zj[e];
        const double x = xr[e] - zj[e];                  // u_i = 1
        s_v[j] = x;
        if (j > 0) m1 = fmax(m1, fabs(x));
      }
    }
  }
  m1 = blk_max(m1, s_red);                               // (its barriers publish s_v)
  if (!has_right || m1 == 0.0) {                         // NORM.max === 0 -> continue (bidiag.js:67): no right reflector
    if (blockIdx.x == 0) {
      for (int j = t; j < n; j += 256) rowfin[first + j] = s_v[j];
      if (has_right) { for (int j = t; j < N; j += 256) VR[(long)i * N + j] = 0.0; if (t == 0) flagR[i] = 0; }
    }
    if (r < M && lane == 0) { y[r] = 0.0; ucol[r] = W[(long)r * N + first] - ucur[r] * s_z[0]; }   // column i+1 as the next step sees it
    return;
  }
  const double x0 = s_v[0];
  const double mx = fmax(m1, fabs(x0));
  double ss = 0.0;
  for (int j = t; j < n; j += 256) { const double x = s_v[j] / mx; ss += x * x; }
  ss = blk_sum(ss, s_red);
  const double nrm = (isfinite(mx) ? sqrt(ss) * mx : mx) * (x0 > 0 ? -1.0 : 1.0);
  const double head = x0 - nrm;
  const double mx2 = fmax(m1, fabs(head));
  double s2 = 0.0;
  for (int j = 1 + t; j < n; j += 256) { const double x = s_v[j] / mx2; s2 += x * x; }
  s2 = blk_sum(s2, s_red);
  { const double x = head / mx2; s2 += x * x; }
  const double div = sqrt(s2);
  double zv = 0.0;
  for (int j = t; j < n; j += 256) {
    const double vj = (j == 0 ? head : s_v[j]) / mx2 / div;
    s_v[j] = vj;
    zv += s_z[j] * vj;
  }
  zv = blk_sum(zv, s_red);                               // (publishes v)
  if (blockIdx.x == 0) {
    for (int j = t; j < N; j += 256) VR[(long)i * N + j] = (j >= first) ? s_v[j - first] : 0.0;
    for (int j = 1 + t; j < n; j += 256) rowfin[first + j] = 0.0;
    if (t == 0) { rowfin[first] = nrm; flagR[i] = 1; }
  }
  if (r >= M) return;
  const double* row = W + (long)r * N + first;
  double acc = 0.0;
  int j = lane;
  for (; j + 7 * 64 < n; j += 8 * 64) {
    double x[8];
#pragma unroll
    for (int q = 0; q < 8; q++) x[q] = row[j + q * 64];
#pragma unroll
    for (int q = 0; q < 8; q++) acc += x[q] * s_v[j + q * 64];
  }
  for (; j < n; j += 64) acc += row[j] * s_v[j];
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
  if (lane == 0) {
    const double yr = 2.0 * (acc - ucur[r] * zv);
    y[r] = yr;
    ucol[r] = row[0] - ucur[r] * s_z[0] - yr * s_v[0];   // column i+1 with this step's update: the next left reflector is built from it
  }
}

// the last step's row (no later launch stores it)
__global__ void bd2_finish_row(double* __restrict__ W, int N, int i, const double* __restrict__ rowfin) {
  const int c = i + 1 + blockIdx.x * blockDim.x + threadIdx.x;
  if (c < N) W[(long)i * N + c] = rowfin[c];
}

// ================================================================ one launch for the whole reduction (M, N <= 2048) =========
// VERDICT r3 #6, second half (hess.hip: hessp has the first). The fused form above is two DEPENDENT launches per step (29 us at
// 2048^2). Here 16 x 16 co-resident workgroups keep the matrix as 16E x 16E tiles IN REGISTERS for all K steps (thread (tr, tc) of
// workgroup (p, q): rows 16E p + tr + 16 a, columns 16E q + tc + 16 b), the rank-2 update of a step costs no traffic, and a step is
// four rounds of 16-byte tagged pairs (xchg16.h) — the two matrix-vector products of a step depend on each other (v is built from
// row i AFTER the left reflector), so each needs its own reduce and broadcast:
//   1  every workgroup publishes its partial of u^T A over its 16E columns (and, where it holds them, row i and column i+1 as they
//      stand); workgroup (p, q) sums the 16 partials of E columns in a fixed order: z = tau u^T A, r = A[i,:] - z, and the scaled
//      norm partial and z.r partial of its columns > i+1;
//   2  it publishes z, r over its columns (+ the three partials, read by everybody): every workgroup builds the right reflector v
//      (reference convention, bidiag.js:66-76) from the same scalars, and knows z.v;
//   3  every workgroup publishes its partial of A v over its 16E rows; (p, q) sums the 16 partials of E rows: y = 2 (A v - u (z.v)),
//      column i+1 after this step's update at its rows (the next left reflector is built from it) and its scaled norm partial;
//   4  it publishes those; every workgroup picks up y and the next column over its rows and all norm partials, updates its tile
//      (A -= u z^T + y v^T) and builds the next u, tau from the same scalars.
// Only the two diagonals of B, the reflectors and tau leave the kernel. Tags are the step + 1, areas alternate with the parity of
// the step; rounds 2 and 4 end with every workgroup having read something of every other, so no area is overwritten before it
// has been read. Every spin is bounded (time-out: abort flag for all workgroups, ND4HIP_ERR_XCHG at the next synchronising call).
constexpr unsigned BP_NREP = 8;                                                    // copies of the arrays everybody reads (see HP_NREP in xchg16.h); 2048^2: 1 / 4 / 8 / 16 copies 27.8 / 27.7 / 27.4 / 28.2 ms
constexpr unsigned BP_REP_A2 = 2 * 256 * 3 * 16, BP_REP_A4 = 2 * 256 * 2 * 16;    // bytes per copy
struct BdPx {
  qx_u64* base;
  unsigned o1, o2, o3, o4;   // [2][256][3T | 2E | T | 2E] values (byte offsets)
  unsigned oA2, oA4;         // [2][256][3 | 2] values read by everybody
  unsigned bytes;
  int* abort;
};

template <int E>
__global__ __launch_bounds__(256) void bdp(double* __restrict__ W, int M, int N, int K, double* __restrict__ ULt, double* __restrict__ tauL,
                                            double* __restrict__ VR, int* __restrict__ flagR, BdPx X, int* status, int delay, long long* stamps, int drop_step) {
  using namespace nd4dpp;
  constexpr int T = 16 * E, V1 = 3 * T, V2 = 2 * E, V3 = T, V4 = 2 * E;
  constexpr int LE = E == 8 ? 3 : (E == 4 ? 2 : 1);                 // log2 E
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, tr = t >> 4, tc = t & 15;
  const int wg = blockIdx.x, p = wg >> 4, q = wg & 15;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(X.base, 0, (int)X.bytes, 0x00020000);
  __shared__ double s_cu[T], s_u[T], s_z[T], s_r[T], s_y[T], s_xp[4][T], s_yp[T], s_q[17 * E];
  __shared__ double s_wm[4], s_wS[4], s_rm[4], s_rS[4], s_rZ[4], s_alpha, s_x0, s_z0;
  __shared__ int s_dead;
  double a[E][E];
#pragma unroll
  for (int ai = 0; ai < E; ai++)
#pragma unroll
    for (int bi = 0; bi < E; bi++) {
      const int r = T * p + tr + 16 * ai, c = T * q + tc + 16 * bi;
      a[ai][bi] = (r < M && c < N) ? W[(long)r * N + c] : 0.0;
    }
  if (t == 0) s_dead = 0;
  bool dead = false;
  // ---- column 0 straight from memory: its slice over the rows p, and the norm partials grouped as round 3 groups them
  {
    if (t < T) s_cu[t] = (T * p + t < M) ? W[(long)(T * p + t) * N] : 0.0;
    if (t == 0) s_alpha = W[0];
    double rv[E], m = 0.0;
#pragma unroll
    for (int c = 0; c < E; c++) {
      const int e = T * (t >> 4) + E * (t & 15) + c;
      rv[c] = (e > 0 && e < M) ? fabs(W[(long)e * N]) : 0.0;
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
  long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = stamps ? (long long)wall_clock64() : 0;   // ND4HIP_BDP_STAMPS: time per phase (100 MHz ticks)
#define BP_STAMP(k) if (stamps) { const long long now = (long long)wall_clock64(); tacc[k] += now - tlast; tlast = now; }
  for (int i = 0; i < K; i++) {
    const unsigned tag = (unsigned)i + 1;
    const int par = i & 1, first = i + 1;
    const int pi = i / T, li = i % T, q1 = first / T, l1 = first % T;
    const bool more = first < K;
    const unsigned R1p = X.o1 + (unsigned)par * 256 * V1 * 16, R2p = X.o2 + (unsigned)par * 256 * V2 * 16;
    const unsigned R3p = X.o3 + (unsigned)par * 256 * V3 * 16, R4p = X.o4 + (unsigned)par * 256 * V4 * 16;
    const unsigned A2p = X.oA2 + (unsigned)par * 256 * 3 * 16, A4p = X.oA4 + (unsigned)par * 256 * 2 * 16;     // copy 0; the copies are BP_REP_A2 / _A4 bytes apart
    const unsigned A2r = A2p + (unsigned)(wg % BP_NREP) * BP_REP_A2, A4r = A4p + (unsigned)(wg % BP_NREP) * BP_REP_A4;   // the copy this workgroup reads
    const unsigned slot1 = R1p + (unsigned)wg * V1 * 16;
    // row i and column i+1 as they stand, by the lanes that hold them
    if (p == pi && tr == (li & 15) && i != drop_step) {                 // (drop_step: tests only, ND4HIP_TEST_DROP_PUBLISH: row never published)
#pragma unroll
      for (int bi = 0; bi < E; bi++) {
        double x = 0.0;
#pragma unroll
        for (int ai = 0; ai < E; ai++) if (ai == (li >> 4)) x = a[ai][bi];
        hp_st(rs, slot1 + (T + tc + 16 * bi) * 16, x, tag);
      }
    }
    if (more && q == q1 && tc == (l1 & 15)) {
#pragma unroll
      for (int ai = 0; ai < E; ai++) {
        double x = 0.0;
#pragma unroll
        for (int bi = 0; bi < E; bi++) if (bi == (l1 >> 4)) x = a[ai][bi];
        hp_st(rs, slot1 + (2 * T + tr + 16 * ai) * 16, x, tag);
      }
    }
    // ---- the left reflector's scalars (LAPACK dlarfg with a max-scaled norm, as bd_vec_col), by every thread in the same order
    const double cmx = fmax(fmax(s_wm[0], s_wm[1]), fmax(s_wm[2], s_wm[3]));
    const double alpha = s_alpha;
    double tau = 0.0, inv = 0.0, dI = alpha;
    if (cmx != 0.0) {
      const double i1 = fast_rcp(cmx);
      double S = 0.0;
#pragma unroll
      for (int g = 0; g < 4; g++) { const double f = s_wm[g] * i1; S += s_wS[g] * f * f; }
      const double sc = fmax(cmx, fabs(alpha));
      const double isc = fast_rcp(sc);
      const double a1 = alpha * isc, c1 = cmx * isc;
      const double t2 = S * c1 * c1 + a1 * a1;
      const double nrm = t2 * fast_rsqrt(t2) * sc;
      const double beta = alpha > 0 ? -nrm : nrm;
      tau = (beta - alpha) * fast_rcp(beta);
      inv = fast_rcp(alpha - beta);
      dI = beta;
    }
    double ur[E];
#pragma unroll
    for (int ai = 0; ai < E; ai++) {
      const int r = T * p + tr + 16 * ai;
      ur[ai] = r < i ? 0.0 : (r == i ? 1.0 : s_cu[tr + 16 * ai] * inv);
    }
    if (t < T) {
      const int r = T * p + t;
      const double u = r < i ? 0.0 : (r == i ? 1.0 : s_cu[t] * inv);
      s_u[t] = u;
      if (q == 0 && r < M) ULt[(long)i * M + r] = u;
    }
    BP_STAMP(0)
    // ---- round 1: partials of u^T A over the rows p
    {
      double zs[E];
#pragma unroll
      for (int bi = 0; bi < E; bi++) zs[bi] = 0.0;
#pragma unroll
      for (int ai = 0; ai < E; ai++)
#pragma unroll
        for (int bi = 0; bi < E; bi++) zs[bi] = fma(ur[ai], a[ai][bi], zs[bi]);
      hp_sum_over_tr<E>(zs, tr, tc, lane, s_xp[wave]);
    }
    __syncthreads();
    if (t < T) hp_st(rs, slot1 + t * 16, (s_xp[0][t] + s_xp[1][t]) + (s_xp[2][t] + s_xp[3][t]), tag);
    {
      HpReq rq[2];
      double xv[2] = {0.0, 0.0};
      rq[0] = HpReq{R1p + (unsigned)((((t / E) & 15) * 16 + q) * V1 + E * p + t % E) * 16, t < 16 * E};
      rq[1] = HpReq{R1p + (unsigned)((pi * 16 + q) * V1 + T + E * p + (t % E)) * 16, t < E};
      hp_wait<2>(rs, rq, xv, tag, dead, X.abort, status, delay >> 8);
      if (t < 16 * E) s_q[t] = xv[0];
      if (t < E) s_q[16 * E + t] = xv[1];
    }
    if (dead) s_dead = 1;
    __syncthreads();
    if (s_dead) break;
    BP_STAMP(1)
    if (t < E) {
      const int c = t, e = T * q + E * p + c;
      double s = 0.0;
#pragma unroll
      for (int pp = 0; pp < 16; pp++) s += s_q[pp * E + c];
      const double z = (e > i && e < N) ? tau * s : 0.0;
      const double r = (e > i && e < N) ? s_q[16 * E + c] - z : 0.0;                 // u_i = 1
      const double am = (e > first) ? fabs(r) : 0.0;
      double m = am;
      if constexpr (E >= 2) m = fmax(m, xor1(m));
      if constexpr (E >= 4) m = fmax(m, xor2(m));
      if constexpr (E >= 8) m = fmax(m, xor4(m));
      const double im = m > 0.0 ? fast_rcp(m) : 0.0;
      const double x = am * im;
      double ss = x * x, zr = (e > first) ? z * (r * im) : 0.0;
      if constexpr (E >= 2) { ss += xor1(ss); zr += xor1(zr); }
      if constexpr (E >= 4) { ss += xor2(ss); zr += xor2(zr); }
      if constexpr (E >= 8) { ss += xor4(ss); zr += xor4(zr); }
      const unsigned slot2 = R2p + (unsigned)wg * V2 * 16;
      hp_st(rs, slot2 + c * 16, z, tag);
      hp_st(rs, slot2 + (E + c) * 16, r, tag);
      for (int r = c; r < (int)BP_NREP; r += E) {                          // (the E lanes all hold the reduced values: lane c writes the copies c, c + E, ...)
        hp_st(rs, A2p + r * BP_REP_A2 + (wg * 3) * 16, m, tag); hp_st(rs, A2p + r * BP_REP_A2 + (wg * 3 + 1) * 16, ss, tag); hp_st(rs, A2p + r * BP_REP_A2 + (wg * 3 + 2) * 16, zr, tag);
      }
    }
    // ---- round 2: z and r over the columns q, the partials of everybody, the entries at column i+1
    {
      HpReq rq[7];
      double xv[7] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
      const unsigned src = R2p + (unsigned)((((t / E) & 15) * 16 + q) * V2) * 16;
      rq[0] = HpReq{src + (t % E) * 16, t < T};
      rq[1] = HpReq{src + (E + t % E) * 16, t < T};
      rq[2] = HpReq{A2r + (unsigned)(t * 3) * 16, true};
      rq[3] = HpReq{A2r + (unsigned)(t * 3 + 1) * 16, true};
      rq[4] = HpReq{A2r + (unsigned)(t * 3 + 2) * 16, true};
      const int ef = first < N ? first : i, jf = ef % T;
      const unsigned srcf = R2p + (unsigned)(((jf / E) * 16 + ef / T) * V2) * 16;
      rq[5] = HpReq{srcf + (E + jf % E) * 16, t == 0};
      rq[6] = HpReq{srcf + (jf % E) * 16, t == 0};
      hp_wait<7>(rs, rq, xv, tag, dead, X.abort, status, delay & 255);
      if (t < T) { s_z[t] = xv[0]; s_r[t] = xv[1]; }
      if (t == 0) { s_x0 = xv[5]; s_z0 = xv[6]; }
      const double wm = wave_max(xv[2]);
      const double f = wm > 0.0 ? xv[2] * fast_rcp(wm) : 0.0;
      const double wS = wave_sum(xv[3] * f * f), wZ = wave_sum(xv[4] * f);
      if (lane == 0) { s_rm[wave] = wm; s_rS[wave] = wS; s_rZ[wave] = wZ; }
    }
    if (dead) s_dead = 1;
    __syncthreads();
    if (s_dead) break;
    BP_STAMP(2)
    // ---- the right reflector's scalars (bidiag.js:66-76: norm = -sign(x_first) |x|, v = (x - norm e) / max / sqrt(sum), H = I - 2 v v^T)
    const double m1 = fmax(fmax(s_rm[0], s_rm[1]), fmax(s_rm[2], s_rm[3]));
    const bool skipR = !(first < N - 1) || m1 == 0.0;              // NORM.max === 0 -> continue (bidiag.js:67)
    const double x0 = s_x0, z0 = s_z0;
    double nrmR = 0.0, vfirst = 0.0, fvr = 0.0, zv = 0.0;
    if (!skipR) {
      const double i1 = fast_rcp(m1);
      double S1 = 0.0, Z1 = 0.0;
#pragma unroll
      for (int g = 0; g < 4; g++) { const double f = s_rm[g] * i1; S1 += s_rS[g] * f * f; Z1 += s_rZ[g] * f; }
      const double mx = fmax(m1, fabs(x0));
      const double imx = fast_rcp(mx);
      const double c1 = m1 * imx, c0 = x0 * imx;
      const double ss = S1 * c1 * c1 + c0 * c0;
      nrmR = (isfinite(mx) ? ss * fast_rsqrt(ss) * mx : mx) * (x0 > 0 ? -1.0 : 1.0);
      const double head = x0 - nrmR;
      const double mx2 = fmax(m1, fabs(head));
      const double inv2 = fast_rcp(mx2);
      const double r1 = m1 * inv2, r0 = head * inv2;
      const double rdiv = fast_rsqrt(S1 * r1 * r1 + r0 * r0);
      fvr = inv2 * rdiv;
      vfirst = r0 * rdiv;
      zv = (z0 * r0 + Z1 * r1) * rdiv;                               // sum over j >= i+1 of z_j v_j
    }
    double vc[E], zc[E];
#pragma unroll
    for (int bi = 0; bi < E; bi++) {
      const int c = T * q + tc + 16 * bi;
      vc[bi] = (!skipR && c >= first && c < N) ? (c == first ? vfirst : s_r[tc + 16 * bi] * fvr) : 0.0;
      zc[bi] = s_z[tc + 16 * bi];
    }
    if (t < T && p == 0 && !skipR) {
      const int c = T * q + t;
      if (c < N) VR[(long)i * N + c] = c >= first ? (c == first ? vfirst : s_r[t] * fvr) : 0.0;
    }
    if (wg == 0 && t == 0) {
      W[(long)i * N + i] = dI;
      if (first < N) W[(long)i * N + first] = skipR ? x0 : nrmR;
      tauL[i] = tau; flagR[i] = skipR ? 0 : 1;
    }
    if (!more) break;
    BP_STAMP(3)
    // ---- round 3: partials of A v over the columns q
    {
      double ys[E];
#pragma unroll
      for (int ai = 0; ai < E; ai++) {
        double y = 0.0;
#pragma unroll
        for (int bi = 0; bi < E; bi++) y = fma(a[ai][bi], vc[bi], y);
        ys[ai] = y;
      }
      const double ysum = hp_sum_over_tc<E>(ys, tc);
      if ((tc & ((16 >> LE) - 1)) == 0) s_yp[tr + 16 * (tc >> (4 - LE))] = ysum;
    }
    __syncthreads();
    if (t < T) hp_st(rs, R3p + (unsigned)(wg * V3 + t) * 16, skipR ? 0.0 : s_yp[t], tag);
    {
      HpReq rq[2];
      double xv[2] = {0.0, 0.0};
      rq[0] = HpReq{R3p + (unsigned)((p * 16 + ((t / E) & 15)) * V3 + E * q + t % E) * 16, t < 16 * E};
      rq[1] = HpReq{R1p + (unsigned)((p * 16 + q1) * V1 + 2 * T + E * q + (t % E)) * 16, t < E};
      hp_wait<2>(rs, rq, xv, tag, dead, X.abort, status, delay >> 8);
      if (t < 16 * E) s_q[t] = xv[0];
      if (t < E) s_q[16 * E + t] = xv[1];
    }
    if (dead) s_dead = 1;
    __syncthreads();
    if (s_dead) break;
    BP_STAMP(4)
    if (t < E) {
      const int c = t, e = T * p + E * q + c;
      double s = 0.0;
#pragma unroll
      for (int pp = 0; pp < 16; pp++) s += s_q[pp * E + c];
      const double ue = s_u[E * q + c];
      const double y = (e > i && e < M) ? 2.0 * (s - ue * zv) : 0.0;
      const double cn = (e > i && e < M) ? fma(-y, vfirst, fma(-ue, z0, s_q[16 * E + c])) : 0.0;   // column i+1 as the tile will hold it
      const double am = (e > first) ? fabs(cn) : 0.0;
      double m = am;
      if constexpr (E >= 2) m = fmax(m, xor1(m));
      if constexpr (E >= 4) m = fmax(m, xor2(m));
      if constexpr (E >= 8) m = fmax(m, xor4(m));
      const double x = m > 0.0 ? am * fast_rcp(m) : 0.0;
      double ss = x * x;
      if constexpr (E >= 2) ss += xor1(ss);
      if constexpr (E >= 4) ss += xor2(ss);
      if constexpr (E >= 8) ss += xor4(ss);
      const unsigned slot4 = R4p + (unsigned)wg * V4 * 16;
      hp_st(rs, slot4 + c * 16, y, tag);
      hp_st(rs, slot4 + (E + c) * 16, cn, tag);
      for (int r = c; r < (int)BP_NREP; r += E) { hp_st(rs, A4p + r * BP_REP_A4 + (wg * 2) * 16, m, tag); hp_st(rs, A4p + r * BP_REP_A4 + (wg * 2 + 1) * 16, ss, tag); }
    }
    // ---- round 4: y and the next column over the rows p, the norm partials of everybody, the next diagonal entry
    {
      HpReq rq[5];
      double xv[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
      const unsigned src = R4p + (unsigned)((p * 16 + ((t / E) & 15)) * V4) * 16;
      rq[0] = HpReq{src + (t % E) * 16, t < T};
      rq[1] = HpReq{src + (E + t % E) * 16, t < T};
      rq[2] = HpReq{A4r + (unsigned)(t * 2) * 16, true};
      rq[3] = HpReq{A4r + (unsigned)(t * 2 + 1) * 16, true};
      rq[4] = HpReq{R4p + (unsigned)((q1 * 16 + l1 / E) * V4 + E + l1 % E) * 16, t == 0};       // row i+1: block q1, local l1
      hp_wait<5>(rs, rq, xv, tag, dead, X.abort, status, delay & 255);
      if (t < T) { s_y[t] = xv[0]; s_cu[t] = xv[1]; }
      if (t == 0) s_alpha = xv[4];
      const double wm = wave_max(xv[2]);
      const double f = wm > 0.0 ? xv[2] * fast_rcp(wm) : 0.0;
      const double wS = wave_sum(xv[3] * f * f);
      if (lane == 0) { s_wm[wave] = wm; s_wS[wave] = wS; }
    }
    if (dead) s_dead = 1;
    __syncthreads();
    if (s_dead) break;
    BP_STAMP(5)
#pragma unroll
    for (int ai = 0; ai < E; ai++) {
      const double ya = s_y[tr + 16 * ai];
#pragma unroll
      for (int bi = 0; bi < E; bi++) a[ai][bi] = fma(-ya, vc[bi], fma(-ur[ai], zc[bi], a[ai][bi]));
    }
    BP_STAMP(6)
  }
#undef BP_STAMP
  if (stamps && t == 0) for (int k = 0; k < 7; k++) stamps[wg * 8 + k] = tacc[k];
}

struct BdWs { double* z; double* y; double* zpart; long sV, sZ; int ncols_total; };

// X[r0:r1, c0:c1] -= a * (scale * a^T X): a = column acol of A2 (ld lda); gate = tau array (scale and on/off switch)
int reflect_left(nd4hip_handle* h, int batch, const Blk& b, const double* A2, long lda, long sA, int acol, const double* tau, long sTau, int tidx, const BdWs& ws) {
  const int nr = b.r1 - b.r0, nc = b.c1 - b.c0;
  if (nr <= 0 || nc <= 0) return 0;
  const int P = (nr + GR - 1) / GR;
  hipLaunchKernelGGL(bd_colsum, dim3((unsigned)((nc + 63) / 64), (unsigned)P, (unsigned)batch), dim3(256), 0, h->stream,
                     b, A2, lda, sA, acol, tau, sTau, tidx, ws.zpart, ws.sZ, ws.ncols_total);
  hipLaunchKernelGGL(bd_colsum_reduce, dim3((unsigned)((nc + 31) / 32), (unsigned)batch), dim3(256), 0, h->stream,
                     b.c0, b.c1, P, ws.zpart, ws.sZ, ws.ncols_total, tau, sTau, tidx, ws.z, ws.sV);
  hipLaunchKernelGGL(bd_rank1, dim3((unsigned)((nc + 255) / 256), (unsigned)((nr + 15) / 16), (unsigned)batch), dim3(256), 0, h->stream,
                     b, A2 + acol, lda, sA, ws.z, ws.sV, tau, sTau, (const int*)nullptr, 0l, tidx);
  ND4_HIP(hipGetLastError());
  return 0;
}
// X[r0:r1, c0:c1] -= (2 X v) v^T with the unit vector v (row vidx of VR); flag switches the step off
int reflect_right(nd4hip_handle* h, int batch, const Blk& b, const double* v, long sVR, const int* flag, long sF, int fidx, const BdWs& ws) {
  const int nr = b.r1 - b.r0, nc = b.c1 - b.c0;
  if (nr <= 0 || nc <= 0) return 0;
  hipLaunchKernelGGL(bd_rowdot, dim3((unsigned)((nr + 3) / 4), (unsigned)batch), dim3(256), 0, h->stream, b, v, sVR, 2.0, flag, sF, fidx, ws.y, ws.sV);
  hipLaunchKernelGGL(bd_rank1, dim3((unsigned)((nc + 255) / 256), (unsigned)((nr + 15) / 16), (unsigned)batch), dim3(256), 0, h->stream,
                     b, ws.y, 1l, ws.sV, v, sVR, (const double*)nullptr, 0l, flag, sF, fidx);
  ND4_HIP(hipGetLastError());
  return 0;
}

}  // namespace

// A [batch, M, N] -> U [batch, M, K], B [batch, K, J], V [batch, J, N]; K = min(M, N), J = K (M >= N) or K + 1
int nd4_gebrd(nd4hip_handle* h, int64_t batch64, int64_t M64, int64_t N64, const double* A, double* U, double* B, double* V) {
  ND4_CHECK_ARG(M64 < 32768 && N64 < 32768 && batch64 < 65536, "nd4_gebrd: extent out of range");
  const int M = (int)M64, N = (int)N64, batch = (int)batch64;
  if (M == 0 || N == 0 || batch == 0) return 0;
  const int K = M < N ? M : N, J = M >= N ? K : K + 1;
  const int mx = M > N ? M : N;
  const int Pmax = (M + GR - 1) / GR;
  Nd4WsScope scope(h);
  void* p = nullptr;
  const size_t nd = (size_t)batch * ((size_t)M * N + (size_t)M * K + (size_t)K * N + (size_t)K + 2 * (size_t)mx + (size_t)(Pmax + 24) * mx);   // (+24 mx: scratch of the fused form)
  ND4_TRY(nd4_ws_alloc(h, sizeof(double) * nd + sizeof(int) * (size_t)batch * (2 * (size_t)K + 2) + 256, &p));
  double* W = static_cast<double*>(p);
  double* UL = W + (size_t)batch * M * N;
  double* VR = UL + (size_t)batch * M * K;
  double* tauL = VR + (size_t)batch * K * N;
  BdWs ws;
  ws.z = tauL + (size_t)batch * K; ws.y = ws.z + (size_t)batch * mx; ws.zpart = ws.y + (size_t)batch * mx;
  ws.sV = mx; ws.sZ = (long)Pmax * mx; ws.ncols_total = mx;
  int* flagR = reinterpret_cast<int*>(ws.zpart + (size_t)batch * (Pmax + 24) * mx);
  int* flips = flagR + (size_t)batch * K + 1;
  const long sW = (long)M * N, sUL = (long)M * K, sVRm = (long)K * N;
  ND4_HIP(hipMemcpyAsync(W, A, sizeof(double) * (size_t)batch * sW, hipMemcpyDeviceToDevice, h->stream));
  ND4_HIP(hipMemsetAsync(VR, 0, sizeof(double) * (size_t)batch * sVRm, h->stream));
  ND4_HIP(hipMemsetAsync(flagR, 0, sizeof(int) * (size_t)batch * K, h->stream));

  // ---- factorisation ----
  static const bool fused_off = getenv("ND4HIP_BIDIAG_UNFUSED") != nullptr;          // A/B switch
  const bool no_persist = getenv("ND4HIP_BIDIAG_NO_PERSIST") != nullptr;          // (read per call: the tests switch between the paths)
  bool persist = !fused_off && !no_persist && batch == 1 && M >= 128 && N >= 128 && M <= 2048 && N <= 2048;
  if (persist) {                                             // (the 256 workgroups of bdp must all be resident at once, see xchg.h)
    int per_cu = 0;
    const hipError_t oe = mx <= 512 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, bdp<2>, 256, 0)
                        : mx <= 1024 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, bdp<4>, 256, 0)
                                     : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, bdp<8>, 256, 0);
    persist = oe == hipSuccess && (long)per_cu * h->num_cu >= 256;
  }
  if (persist) {
    // ---- one launch: 16 x 16 workgroups keep the matrix in registers for the whole reduction (see bdp) ----
    const int E = mx <= 512 ? 2 : (mx <= 1024 ? 4 : 8), T = 16 * E;
    const size_t V1 = 3 * T, V2 = 2 * E, V3 = T, V4 = 2 * E;
    const size_t xbytes = 16 * 2 * 256 * (V1 + V2 + V3 + V4) + (size_t)BP_NREP * (BP_REP_A2 + BP_REP_A4);
    Nd4WsScope scope2(h);
    void* qp = nullptr;
    ND4_TRY(nd4_ws_alloc(h, sizeof(double) * (size_t)K * M + xbytes + 64, &qp));
    double* ULt = static_cast<double*>(qp);                  // [K][M]: reflector i as a row (transposed into UL at the end)
    BdPx X;
    X.base = reinterpret_cast<qx_u64*>(ULt + (size_t)K * M);
    X.o1 = 0; X.o2 = X.o1 + (unsigned)(2 * 256 * V1 * 16); X.o3 = X.o2 + (unsigned)(2 * 256 * V2 * 16); X.o4 = X.o3 + (unsigned)(2 * 256 * V3 * 16);
    X.oA2 = X.o4 + (unsigned)(2 * 256 * V4 * 16); X.oA4 = X.oA2 + BP_NREP * BP_REP_A2; X.bytes = X.oA4 + BP_NREP * BP_REP_A4;
    X.abort = reinterpret_cast<int*>(reinterpret_cast<char*>(X.base) + xbytes);
    ND4_HIP(hipMemsetAsync(X.base, 0, xbytes + 64, h->stream));
    static const bool want_stamps = getenv("ND4HIP_BDP_STAMPS") != nullptr;
    // s_sleep(8) units (~0.22 us) before the first look of rounds 1 / 3 (high byte) and rounds 2 / 4 (low byte)
    const int delay = getenv("ND4HIP_BDP_DELAY") ? atoi(getenv("ND4HIP_BDP_DELAY")) : (0 << 8 | 4);
    long long* stamps = nullptr;
    if (want_stamps) { void* sp = nullptr; ND4_TRY(nd4_ws_alloc(h, sizeof(long long) * 256 * 8, &sp)); stamps = static_cast<long long*>(sp); }
    if (E == 2) hipLaunchKernelGGL(bdp<2>, dim3(256), dim3(256), 0, h->stream, W, M, N, K, ULt, tauL, VR, flagR, X, h->xstat, delay, stamps, nd4_test_drop_panel());
    else if (E == 4) hipLaunchKernelGGL(bdp<4>, dim3(256), dim3(256), 0, h->stream, W, M, N, K, ULt, tauL, VR, flagR, X, h->xstat, delay, stamps, nd4_test_drop_panel());
    else hipLaunchKernelGGL(bdp<8>, dim3(256), dim3(256), 0, h->stream, W, M, N, K, ULt, tauL, VR, flagR, X, h->xstat, delay, stamps, nd4_test_drop_panel());
    ND4_HIP(hipGetLastError());
    if (stamps) {                                     // per step, in us: left reflector | round 1 | sums + round 2 | right reflector | round 3 | sums + round 4 | update
      long long hs[256 * 8];
      ND4_HIP(hipMemcpyAsync(hs, stamps, sizeof(hs), hipMemcpyDeviceToHost, h->stream));
      ND4_HIP(hipStreamSynchronize(h->stream));
      for (int g : {0, 119, 255})
        fprintf(stderr, "bdp %dx%d wg %3d: left %.2f  r1 %.2f  r2 %.2f  right %.2f  r3 %.2f  r4 %.2f  update %.2f us per step\n", M, N, g,
                hs[g * 8] * 0.01 / K, hs[g * 8 + 1] * 0.01 / K, hs[g * 8 + 2] * 0.01 / K, hs[g * 8 + 3] * 0.01 / K, hs[g * 8 + 4] * 0.01 / K,
                hs[g * 8 + 5] * 0.01 / K, hs[g * 8 + 6] * 0.01 / K);
    }
    ND4_TRY(nd4_transpose(h, K, M, ULt, M, UL, K, 1, 0, 0));
  } else if (!fused_off && batch == 1 && M >= 128 && N >= 128 && M <= BD2_MAXM && N <= BD2_MAXN) {
    double* ucol = ws.zpart;                                 // M doubles (the partial sums of the unfused form are not needed)
    double* rowfin = ucol + mx;                              // N doubles
    double* ubuf[2] = {rowfin + mx, rowfin + 2 * mx};        // contiguous u of the previous / the current step
    constexpr int PMAXZ = 8;
    double* zbuf[2] = {rowfin + 3 * mx, rowfin + (3 + PMAXZ) * (size_t)mx};     // z in up to PMAXZ row parts, alternating by step
    int Pprev = 1;
    hipLaunchKernelGGL(bd2_init_ucol, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, h->stream, W, M, N, ucol);
    int prev = -1;                                           // step whose rank-2 update and row are still pending
    for (int i = 0; i < K; i++) {
      const int nc = N - i - 1, nr = M - i - 1;
      const int has_right = (i + 1 < N - 1) ? 1 : 0;
      const int ncb = nc > 0 ? (nc + 15) / 16 : 1;
      // two row parts (z as two partial sums added by its readers) while one part per column block would leave half of the CUs idle;
      // more parts cost more than they give (every part repeats the reflector prologue, every reader adds P partials: 4-8 parts
      // on 256 threads 71 ms against 63 at 2048^2)
      static const int p_env = getenv("ND4HIP_BD_PARTS") ? atoi(getenv("ND4HIP_BD_PARTS")) : 0;
      const int P = p_env ? (p_env > PMAXZ ? PMAXZ : p_env) : ((ncb <= 128 && M - i >= 512) ? 2 : 1);
      const int rpp = ((((M - i) + P - 1) / P) + 15) & ~15;
      hipLaunchKernelGGL(bd2_colpass, dim3((unsigned)ncb, (unsigned)P), dim3(CPT), sizeof(double) * 3 * (size_t)rpp, h->stream,
                         W, M, N, i, UL, K, tauL, ucol, zbuf[(i + 1) & 1], Pprev, zbuf[i & 1], prev, ubuf[(i + 1) & 1], ubuf[i & 1], ws.y, VR, rowfin);
      prev = -1;
      if (nc <= 0) continue;
      hipLaunchKernelGGL(bd2_rowpass, dim3((unsigned)(nr > 0 ? (nr + 3) / 4 : 1)), dim3(256), sizeof(double) * 2 * (size_t)nc, h->stream,
                         W, M, N, i, UL, K, zbuf[i & 1], P, VR, flagR, ws.y, has_right, rowfin, ubuf[i & 1], ucol);
      prev = i; Pprev = P;
    }
    if (prev >= 0)                                           // only for M < N: the last step has no rows below, its row is all that is pending
      hipLaunchKernelGGL(bd2_finish_row, dim3((unsigned)((N - prev - 1 + 255) / 256)), dim3(256), 0, h->stream, W, N, prev, rowfin);
    ND4_HIP(hipGetLastError());
  } else
  for (int i = 0; i < K; i++) {
    hipLaunchKernelGGL(bd_vec_col, dim3((unsigned)batch), dim3(256), 0, h->stream, W, M, N, i, UL, K, tauL);
    {
      Blk b{W, N, sW, i, M, i + 1, N};                       // (I - tau u u^T) on the columns to the right
      ND4_TRY(reflect_left(h, batch, b, UL, K, sUL, i, tauL, K, i, ws));
    }
    if (i + 1 < N - 1) {                                     // something to the right of the super-diagonal
      hipLaunchKernelGGL(bd_vec_row, dim3((unsigned)batch), dim3(256), 0, h->stream, W, M, N, i, VR, K, flagR);
      Blk b{W, N, sW, i + 1, M, i + 1, N};                   // rows below, same reflector from the right
      ND4_TRY(reflect_right(h, batch, b, VR + (long)i * N, sVRm, flagR, K, i, ws));
    }
  }
  {
    const unsigned gy = (unsigned)(K < 512 ? K : 512);
    hipLaunchKernelGGL(bd_extract, dim3((unsigned)((J + 255) / 256), gy, (unsigned)batch), dim3(256), 0, h->stream, W, M, N, B, K, J);
  }
  if (batch <= 4 && K >= 256) {
    // one big matrix: U and V at once from the compact-WY form of the stored reflectors (nd4_wy_form, shared with QR):
    // U = [I;0] - UL (T_L UL[0:K,:]^T) with the taus on T_L's diagonal; V^T[:, 0:J] = E_J - VR^T (T_R VR[:,0:J]) with
    // tau = 2 for the unit vectors of the right reflectors (a skipped step stored v = 0 and drops out).
    Nd4WsScope scope2(h);
    void* q = nullptr;
    ND4_TRY(nd4_ws_alloc(h, sizeof(double) * ((size_t)N * K + (size_t)N * J + (size_t)K + 16), &q));
    double* VRt = static_cast<double*>(q);                   // N x K: column k = v_k
    double* Vt = VRt + (size_t)N * K;                        // N x J
    double* twos = Vt + (size_t)N * J;
    hipLaunchKernelGGL(bd_fill, dim3((unsigned)((K + 255) / 256)), dim3(256), 0, h->stream, twos, K, 2.0);
    for (int m = 0; m < batch; m++) {
      ND4_TRY(nd4_wy_form(h, M, K, UL + (size_t)m * sUL, tauL + (size_t)m * K, 1, U + (size_t)m * M * K, K));
      ND4_TRY(nd4_transpose(h, K, N, VR + (size_t)m * sVRm, N, VRt, K, 1, 0, 0));
      ND4_TRY(nd4_wy_form(h, N, K, VRt, twos, 1, Vt, J));
      ND4_TRY(nd4_transpose(h, N, J, Vt, J, V + (size_t)m * J * N, N, 1, 0, 0));
    }
  } else {
  // ---- U = H_0 ... H_{K-1} [I; 0]: reflectors applied backwards ----
  {
    const unsigned gy = (unsigned)(M < 512 ? M : 512);
    hipLaunchKernelGGL(bd_set_identity, dim3((unsigned)((K + 255) / 256), gy, (unsigned)batch), dim3(256), 0, h->stream, U, M, K);
  }
  for (int i = K - 1; i >= 0; i--) {
    Blk b{U, K, (long)M * K, i, M, i, K};
    ND4_TRY(reflect_left(h, batch, b, UL, K, sUL, i, tauL, K, i, ws));
  }
  // ---- V = first J rows of H^R_{r-1} ... H^R_0 ----
  {
    const unsigned gy = (unsigned)(J < 512 ? J : 512);
    hipLaunchKernelGGL(bd_set_identity, dim3((unsigned)((N + 255) / 256), gy, (unsigned)batch), dim3(256), 0, h->stream, V, J, N);
  }
  for (int k = K - 1; k >= 0; k--) {
    if (!(k + 1 < N - 1)) continue;
    Blk b{V, N, (long)J * N, k + 1, J, k + 1, N};
    ND4_TRY(reflect_right(h, batch, b, VR + (long)k * N, sVRm, flagR, K, k, ws));
  }
  }
  ND4_HIP(hipGetLastError());
  // ---- the reference's sign convention on (columns of U, rows of B) ----
  return nd4_givens_signs(h, batch, M, K, J, M > N, U, K, (long)M * K, B, J, (long)K * J, tauL, K, flips);
}
