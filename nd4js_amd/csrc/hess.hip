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
