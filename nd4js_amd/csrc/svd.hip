// One-sided (Hestenes) Jacobi SVD on the ROWS of row-major fp64 matrices, batched.
//
// Output contract of src/la/svd.js:25 (svd_decomp = svd_dc, svd_dc.js:883-932): A = U diag(sv) V,
// sv >= 0 sorted descending, V rows = right singular vectors. The reference's own Jacobi relatives
// (svd_jac_2sided.js:95-134, _svd_jac_utils.js:123-188) define the stopping rule and the
// post-processing that are mirrored here:
//   * rotate rows p,q iff |a_p.a_q| > N*eps*|a_p||a_q|  (one-sided form of svd_jac_2sided.js:112);
//     stop after a sweep without rotations (:95-97);
//   * epilogue = _svd_jac_post: sv from the work matrix, non-negative, sorted descending with the
//     same permutation applied to U^T rows and V rows, U transposed at the end.
// W = Ut * A is iterated (Ut = accumulated left rotations, starts as I): at convergence the rows of W
// are orthogonal, sv_i = |w_i|, V_i = w_i / sv_i, U = Ut^T. Rows are contiguous -> every access is
// coalesced. Pairs follow a round-robin tournament: N/2 disjoint pairs per step, N-1 steps per sweep,
// one workgroup per pair; the three inner products use wave shuffle reductions.
// Rectangular input is reduced to square by QR first (svd_jac_2sided.js:42-52 does the same).
#include "svd_internal.h"
#include "dpp.h"
#include <cmath>
#include <cfloat>
#include <cstring>
#include <cstdlib>

namespace {

constexpr int MAX_SWEEPS = 60;

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

__global__ __launch_bounds__(256) void jac_step(double* __restrict__ Wm, double* __restrict__ Utm, int N, long strideM,
                                                 int n2, int step, double tol2, JacState* __restrict__ st,
                                                 const double* __restrict__ floor2, unsigned long long* __restrict__ offmax) {
  const int mat = blockIdx.y;
  if (st[mat].done) return;
  int p, q;
  nd4_rr_pair(n2, step, blockIdx.x, p, q);
  if (q >= N) return;                                  // dummy player of an odd N
  double* wp = Wm + mat * strideM + (long)p * N;
  double* wq = Wm + mat * strideM + (long)q * N;
  const int t = threadIdx.x;
  double aa = 0.0, bb = 0.0, ab = 0.0;
  for (int j = t; j < N; j += 256) {
    const double a = wp[j], b = wq[j];
    aa += a * a; bb += b * b; ab += a * b;
  }
  __shared__ double s_red[4][3];
  aa = wave_sum(aa); bb = wave_sum(bb); ab = wave_sum(ab);
  if ((t & 63) == 0) { s_red[t >> 6][0] = aa; s_red[t >> 6][1] = bb; s_red[t >> 6][2] = ab; }
  __syncthreads();
  aa = (s_red[0][0] + s_red[1][0]) + (s_red[2][0] + s_red[3][0]);
  bb = (s_red[0][1] + s_red[1][1]) + (s_red[2][1] + s_red[3][1]);
  ab = (s_red[0][2] + s_red[1][2]) + (s_red[2][2] + s_red[3][2]);
  // rows at or below the noise floor N*eps*max|a_i| carry no information (a rotation only swaps
  // rounding noise): they are frozen here and get an orthonormal completion in the epilogue
  const double fl = floor2[mat];
  if (aa <= fl || bb <= fl) return;
  const double lim = tol2 * aa * bb;
  if (!(ab * ab > lim)) return;                        // orthogonal enough (also: NaN)
  // rotation that zeroes the inner product: t^2 + 2*zeta*t - 1 = 0, smaller root
  const double zeta = (bb - aa) / (2.0 * ab);
  const double tn = copysign(1.0, zeta) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
  const double c = 1.0 / sqrt(1.0 + tn * tn), s = c * tn;
  // Rutishauser form with tau = tan(theta/2): x_p' = x_p - s (x_q + tau x_p), x_q' = x_q + s (x_p - tau x_q).
  // 1 - c = s*tau is carried exactly, so the rotation stays orthogonal to O(eps * theta^2) even when c
  // rounds to 1 (plain c*x - s*y grows every row norm by theta^2 per tiny rotation: a systematic drift).
  const double tau = s / (1.0 + c);
  for (int j = t; j < N; j += 256) {
    const double a = wp[j], b = wq[j];
    wp[j] = a - s * (b + tau * a);
    wq[j] = b + s * (a - tau * b);
  }
  double* up = Utm + mat * strideM + (long)p * N;
  double* uq = Utm + mat * strideM + (long)q * N;
  for (int j = t; j < N; j += 256) {
    const double a = up[j], b = uq[j];
    up[j] = a - s * (b + tau * a);
    uq[j] = b + s * (a - tau * b);
  }
  if (t == 0) {
    atomicAdd(&st[mat].rotations, 1u);
    const double rel = (ab * ab) / (aa * bb);          // cos^2 of the angle before the rotation
    atomicMax(offmax, (unsigned long long)__double_as_longlong(rel));
  }
}

// floor2[mat] = (N*eps)^2 * max_i |w_i|^2 from the row norms (jac_norms); one workgroup per matrix
__global__ __launch_bounds__(256) void jac_floor(const double* __restrict__ svr, int N, double tol2, double* __restrict__ floor2) {
  const double* s = svr + (long)blockIdx.x * N;
  double mx = 0.0;
  for (int i = threadIdx.x; i < N; i += 256) mx = fmax(mx, s[i]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off));
  __shared__ double s_mx[4];
  if ((threadIdx.x & 63) == 0) s_mx[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) { const double m = fmax(fmax(s_mx[0], s_mx[1]), fmax(s_mx[2], s_mx[3])); floor2[blockIdx.x] = tol2 * m * m; }
}

__global__ void jac_sweep_end(JacState* __restrict__ st, int batch, unsigned* __restrict__ active, unsigned long long* __restrict__ rot_total) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= batch) return;
  if (!st[m].done) {
    if (st[m].rotations == 0) st[m].done = 1; else { atomicAdd(active, 1u); atomicAdd(rot_total, (unsigned long long)st[m].rotations); }
    st[m].rotations = 0;
  }
}

// sv_raw[i] = |w_i| : one wave per row
__global__ __launch_bounds__(256) void jac_norms(const double* __restrict__ Wm, int N, long strideM, double* __restrict__ svr) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= N) return;
  const double* w = Wm + blockIdx.y * strideM + (long)row * N;
  double s = 0.0;
  for (int j = lane; j < N; j += 64) s += w[j] * w[j];
  s = wave_sum(s);
  if (lane == 0) svr[(long)blockIdx.y * N + row] = sqrt(s);
}

// rank[i] = position of row i in the descending order (stable: ties keep ascending index)
__global__ void jac_rank(const double* __restrict__ svr, int N, int* __restrict__ rank) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const double* s = svr + (long)blockIdx.y * N;
  const double si = s[i];
  int r = 0;
  for (int j = 0; j < N; j++) { const double sj = s[j]; r += (sj > si || (sj == si && j < i)) ? 1 : 0; }
  rank[(long)blockIdx.y * N + i] = r;
}

// V[rank_i] = w_i / sv_i ; Utp[rank_i] = Ut[i] ; sv[rank_i] = sv_i
__global__ __launch_bounds__(256) void jac_emit(const double* __restrict__ Wm, const double* __restrict__ Utm, int N, long strideM,
                                                 const double* __restrict__ svr, const int* __restrict__ rank,
                                                 double* __restrict__ V, double* __restrict__ Utp, double* __restrict__ sv) {
  const int i = blockIdx.x, mat = blockIdx.y, t = threadIdx.x;
  const int r = rank[(long)mat * N + i];
  const double s = svr[(long)mat * N + i];
  const double inv = s > 0.0 ? 1.0 / s : 0.0;
  const double* w = Wm + mat * strideM + (long)i * N;
  const double* u = Utm + mat * strideM + (long)i * N;
  double* v = V + mat * strideM + (long)r * N;
  double* o = Utp + mat * strideM + (long)r * N;
  for (int j = t; j < N; j += 256) { v[j] = w[j] * inv; o[j] = u[j]; }
  if (t == 0) sv[(long)mat * N + r] = s;
}

// number of leading singular values above the noise floor (sv is sorted): rows z0.. of V need a completion
__global__ void jac_count_rank(const double* __restrict__ svm, int N, const double* __restrict__ floor2, int* __restrict__ z0, int batch) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= batch) return;
  const double* sv = svm + (long)m * N;
  const double fl = floor2[m];
  int z = N;
  while (z > 0 && sv[z - 1] * sv[z - 1] <= fl) z--;
  z0[m] = z;
}

// Rows of V whose singular value is at or below the noise floor (sorted to the end) are replaced by an orthonormal
// completion: pick the unit vector e_j with the largest residual against the rows above, orthogonalise
// twice (classical Gram-Schmidt with re-orthogonalisation), normalise. One workgroup per matrix.
__global__ __launch_bounds__(1024) void jac_complete(double* __restrict__ Vm, int N, long strideM, const double* __restrict__ svm,
                                                      const double* __restrict__ floor2, double* __restrict__ scratch) {
  double* V = Vm + blockIdx.x * strideM;
  const double* sv = svm + (long)blockIdx.x * N;
  double* d = scratch + (long)blockIdx.x * N;
  const int t = threadIdx.x, T = 1024, lane = t & 63, wave = t >> 6, nw = 16;
  int z0 = N;
  const double fl = floor2[blockIdx.x];
  while (z0 > 0 && sv[z0 - 1] * sv[z0 - 1] <= fl) z0--;
  if (z0 == N) return;
  __shared__ double s_val[16];
  __shared__ int s_idx[16];
  __shared__ int s_best;
  __shared__ double s_norm;
  for (int r = z0; r < N; r++) {
    // candidate = column with the smallest sum of squares over rows < r
    double best = DBL_MAX; int bidx = 0x7fffffff;
    for (int j = t; j < N; j += T) {
      double cn = 0.0;
      for (int k = 0; k < r; k++) { const double x = V[(long)k * N + j]; cn += x * x; }
      if (cn < best) { best = cn; bidx = j; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const double ob = __shfl_xor(best, off); const int oi = __shfl_xor(bidx, off);
      if (ob < best || (ob == best && oi < bidx)) { best = ob; bidx = oi; }
    }
    if (lane == 0) { s_val[wave] = best; s_idx[wave] = bidx; }
    __syncthreads();
    if (t == 0) {
      double b = s_val[0]; int bi = s_idx[0];
      for (int w = 1; w < nw; w++) if (s_val[w] < b || (s_val[w] == b && s_idx[w] < bi)) { b = s_val[w]; bi = s_idx[w]; }
      s_best = bi;
    }
    __syncthreads();
    const int cand = s_best;
    double* v = V + (long)r * N;
    for (int j = t; j < N; j += T) v[j] = (j == cand) ? 1.0 : 0.0;
    __syncthreads();
    for (int pass = 0; pass < 2; pass++) {
      for (int k = wave; k < r; k += nw) {
        double s = 0.0;
        for (int j = lane; j < N; j += 64) s += V[(long)k * N + j] * v[j];
        s = wave_sum(s);
        if (lane == 0) d[k] = s;
      }
      __syncthreads();
      for (int j = t; j < N; j += T) {
        double acc = v[j];
        for (int k = 0; k < r; k++) acc -= d[k] * V[(long)k * N + j];
        v[j] = acc;
      }
      __syncthreads();
    }
    double s = 0.0;
    for (int j = t; j < N; j += T) s += v[j] * v[j];
    s = wave_sum(s);
    if (lane == 0) s_val[wave] = s;
    __syncthreads();
    if (t == 0) { double n = 0.0; for (int w = 0; w < nw; w++) n += s_val[w]; s_norm = sqrt(n); }
    __syncthreads();
    const double inv = 1.0 / s_norm;
    for (int j = t; j < N; j += T) v[j] *= inv;
    __syncthreads();
  }
}

// ---- small matrices (N <= 64): ALL sweeps in one launch, one workgroup per matrix (round 3) ----
// The driver below costs a launch per round-robin step (or three per block step) and a stream synchronisation every few sweeps: a
// single 16^2 matrix took 1.0 ms, 64^2 1.5 ms, almost all of it launch and round-trip latency. Here W and Ut live in LDS (rows
// padded to an odd stride), 8 lanes own a row pair of the current round (32 pairs x 8 lanes = 256 threads; each lane 1/8 of the
// columns, the three inner products meet by three DPP-free shuffle steps inside the 8 lanes), one barrier per round, and the sweep loop
// runs on the device until a sweep rotates nothing — the same rule (svd_jac_2sided.js:95-97, :112 in its one-sided form), the same
// noise floor and the same Rutishauser rotation as jac_step. W and Ut go back to global memory for the common epilogue.
template <int NMAX>
__global__ __launch_bounds__(256) void jac_small(double* __restrict__ Wm, double* __restrict__ Utm, int N, long strideM, double tol2,
                                                  double* __restrict__ floor2_out, unsigned* __restrict__ sweeps_max, unsigned* __restrict__ not_converged,
                                                  unsigned long long* __restrict__ offmax, unsigned long long* __restrict__ rot_total, int max_sweeps) {
  constexpr int LD = NMAX + 1;
  constexpr int C8 = NMAX / 8;                                     // columns per lane
  __shared__ double s_w[NMAX * LD], s_u[NMAX * LD];
  __shared__ double s_red[4];
  __shared__ unsigned s_rot;
  __shared__ unsigned long long s_off;
  __shared__ unsigned short s_pair[(NMAX - 1) * 32];               // the tournament, once: p | q << 8 per (step, slot) (two integer divisions each)
  const int mat = blockIdx.x, t = threadIdx.x, pr = t >> 3, sub = t & 7;
  double* W = Wm + mat * strideM;
  double* Ut = Utm + mat * strideM;
  for (int e = t; e < NMAX * NMAX; e += 256) {
    const int r = e / NMAX, c = e % NMAX;
    s_w[r * LD + c] = (r < N && c < N) ? W[(long)r * N + c] : 0.0;
    s_u[r * LD + c] = (r == c) ? 1.0 : 0.0;
  }
  if (t == 0) { s_rot = 0; s_off = 0; }
  {
    const int n2s = (N + 1) & ~1;
    for (int e = t; e < (n2s - 1) * 32; e += 256) {
      int p = 0, q = 255;
      if ((e & 31) < n2s / 2) nd4_rr_pair(n2s, e >> 5, e & 31, p, q);
      s_pair[e] = (unsigned short)(p | (q << 8));
    }
  }
  __syncthreads();
  // noise floor: (N eps)^2 max_i |w_i|^2 of the input rows (jac_norms + jac_floor)
  double mx = 0.0;
  for (int r = t >> 3; r < N; r += 32) {
    double ss = 0.0;
#pragma unroll
    for (int i = 0; i < C8; i++) { const double x = s_w[r * LD + sub + 8 * i]; ss += x * x; }
    ss += nd4dpp::xor1(ss); ss += nd4dpp::xor2(ss); ss += nd4dpp::xor4(ss);
    mx = fmax(mx, ss);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off));
  if ((t & 63) == 0) s_red[t >> 6] = mx;
  __syncthreads();
  const double fl = tol2 * fmax(fmax(s_red[0], s_red[1]), fmax(s_red[2], s_red[3]));
  if (t == 0) floor2_out[mat] = fl;
  const int n2 = (N + 1) & ~1;
  int sweeps = 0;
  unsigned long long rotations = 0;
  bool converged = N <= 1;
  unsigned my_rot = 0; double my_off = 0.0;
  while (!converged && sweeps < max_sweeps) {
    if (t == 0) s_off = 0;                                          // only the LAST sweep's rotations count (the rounds' barriers order this)
    for (int step = 0; step < n2 - 1; step++) {
      const unsigned pq = s_pair[step * 32 + pr];
      const int p = pq & 255, q = pq >> 8;
      if (q < N) {                                                  // (uniform inside the 8 lanes of a pair)
        double a[C8], b[C8], ua[C8], ub[C8], aa = 0.0, bb = 0.0, ab = 0.0;
#pragma unroll
        for (int i = 0; i < C8; i++) {                              // (the Ut rows are requested with the W rows: one LDS round trip)
          a[i] = s_w[p * LD + sub + 8 * i]; b[i] = s_w[q * LD + sub + 8 * i];
          ua[i] = s_u[p * LD + sub + 8 * i]; ub[i] = s_u[q * LD + sub + 8 * i];
        }
#pragma unroll
        for (int i = 0; i < C8; i++) { aa += a[i] * a[i]; bb += b[i] * b[i]; ab += a[i] * b[i]; }
        // the 8 lanes of a pair sit in one DPP row: cross-lane adds at VALU speed (a ds_bpermute butterfly is ~100 cycles per step)
        aa += nd4dpp::xor1(aa); bb += nd4dpp::xor1(bb); ab += nd4dpp::xor1(ab);
        aa += nd4dpp::xor2(aa); bb += nd4dpp::xor2(bb); ab += nd4dpp::xor2(ab);
        aa += nd4dpp::xor4(aa); bb += nd4dpp::xor4(bb); ab += nd4dpp::xor4(ab);
        if (aa > fl && bb > fl && ab * ab > tol2 * aa * bb) {
          // few-ulp reciprocals / reciprocal square roots (as in svd_block.hip): s and tau = s / (1 + c) only have to be mutually
          // consistent to a few ulp; two sqrt and three divisions were ~175 dependent instructions per round
          const double zeta = (bb - aa) * 0.5 * nd4dpp::fast_rcp(ab);
          const double z1 = 1.0 + zeta * zeta;
          const double tn = copysign(nd4dpp::fast_rcp(fabs(zeta) + z1 * nd4dpp::fast_rsqrt(z1)), zeta);
          const double c = nd4dpp::fast_rsqrt(1.0 + tn * tn), sn = c * tn, tau = sn * nd4dpp::fast_rcp(1.0 + c);
#pragma unroll
          for (int i = 0; i < C8; i++) {
            s_w[p * LD + sub + 8 * i] = a[i] - sn * (b[i] + tau * a[i]);
            s_w[q * LD + sub + 8 * i] = b[i] + sn * (a[i] - tau * b[i]);
            s_u[p * LD + sub + 8 * i] = ua[i] - sn * (ub[i] + tau * ua[i]);
            s_u[q * LD + sub + 8 * i] = ub[i] + sn * (ua[i] - tau * ub[i]);
          }
          my_rot += 1u;                                              // (counted per pair leader at the sweep's end: no LDS atomics in the rounds)
          my_off = fmax(my_off, (ab * ab) * nd4dpp::fast_rcp(aa * bb));
        }
      }
      __syncthreads();
    }
    sweeps++;
    if (sub == 0 && my_rot) { atomicAdd(&s_rot, my_rot); atomicMax(&s_off, (unsigned long long)__double_as_longlong(my_off)); }
    my_rot = 0; my_off = 0.0;
    __syncthreads();
    const unsigned r = s_rot;
    __syncthreads();
    if (t == 0) s_rot = 0;
    rotations += r;
    converged = r == 0;
    __syncthreads();
  }
  for (int e = t; e < N * N; e += 256) {
    const int r = e / N, c = e % N;
    W[e] = s_w[r * LD + c];
    Ut[e] = s_u[r * LD + c];
  }
  if (t == 0) {
    atomicMax(sweeps_max, (unsigned)sweeps);
    if (!converged) atomicAdd(not_converged, 1u);
    atomicMax(offmax, s_off);
    atomicAdd(rot_total, rotations);
  }
}

// W (N x N, batch) holds the input and is destroyed. Outputs U, sv, V (dense, N x N / N).
int jacobi_square(nd4hip_handle* h, int batch, int N, double* W, double* U, double* sv, double* V,
                  int* sweeps_out, double* offnorm_out) {
  Nd4WsScope scope(h);
  const long sM = (long)N * N;
  void* p = nullptr;
  // Block Jacobi on the matrix cores (svd_block.hip) wants N = 0 mod 64. Other N >= 16 run it on the matrix padded with
  // zero rows and columns to Np = the next multiple of 64: a zero row has norm 0 < the noise floor and is never rotated, a
  // zero column stays zero under row rotations, so the leading N x N parts of W and Ut evolve exactly as they would alone
  // (Ut' = diag(Ut, I)) and are copied back before the epilogue. (The row-pair kernel it replaces there is 4-6x slower:
  // N = 1000 took 222 ms against 39 ms at 1024.)
  // environment switches are read once per process (never inside the sweep loop)
  static const bool noblock = getenv("ND4HIP_SVD_NOBLOCK") != nullptr;
  static const bool precheck_always = getenv("ND4HIP_JAC_PRECHECK_ALWAYS") != nullptr;
  static const bool debug = getenv("ND4HIP_SVD_DEBUG") != nullptr;
  static const bool small_off = getenv("ND4HIP_SVD_NO_SMALL") != nullptr;
  const bool small = !small_off && N > 1 && N <= 64;              // all sweeps in one launch (jac_small)
  const bool blocked = N >= 16 && !noblock && !small;             // (without jac_small: below 16 the row-pair kernel)
  const int Np = blocked ? ((N + 63) / 64) * 64 : N;
  const bool padded = Np != N;
  const long sMp = (long)Np * Np;
  const size_t nblock = blocked ? nd4_jacobi_block_scratch_doubles(batch, Np) : 0;
  const size_t npad = padded ? (size_t)batch * (2 * sMp + Np) : 0;
  const size_t nd = (size_t)batch * (2 * sM + 2 * N + 1) + nblock + npad;
  const size_t nrank = ((size_t)batch * N + 1) & ~size_t(1);          // keeps the 64-bit words behind it aligned
  ND4_TRY(nd4_ws_alloc(h, nd * sizeof(double) + nrank * sizeof(int) + (size_t)batch * sizeof(JacState) + 96, &p));
  double* Ut = static_cast<double*>(p);
  double* Utp = Ut + (size_t)batch * sM;
  double* svr = Utp + (size_t)batch * sM;
  double* scratch = svr + (size_t)batch * N;
  double* floor2 = scratch + (size_t)batch * N;
  double* bscratch = floor2 + batch;
  double* Wp = bscratch + nblock;                                       // padded copies (only when padded)
  double* Utpad = Wp + (padded ? (size_t)batch * sMp : 0);
  double* svrp = Utpad + (padded ? (size_t)batch * sMp : 0);
  int* rank = reinterpret_cast<int*>(bscratch + nblock + npad);
  JacState* st = reinterpret_cast<JacState*>(rank + nrank);
  unsigned* active = reinterpret_cast<unsigned*>(st + batch);           // [1] + padding
  unsigned long long* offmax = reinterpret_cast<unsigned long long*>(active + 2);
  unsigned long long* rot_total = offmax + 1;                           // rotations applied, summed over sweeps and matrices
  void* pin = nullptr;
  ND4_TRY(nd4_pinned(h, 64 + sizeof(int) * (size_t)batch, &pin));
  unsigned* h_active = static_cast<unsigned*>(pin);
  unsigned long long* h_off = reinterpret_cast<unsigned long long*>(h_active + 2);

  if (!small) ND4_TRY(nd4_set_identity(h, N, N, Ut, N, batch, sM));
  ND4_HIP(hipMemsetAsync(st, 0, sizeof(JacState) * batch + 32, h->stream));
  if (padded) {
    ND4_HIP(hipMemsetAsync(Wp, 0, sizeof(double) * (size_t)batch * sMp, h->stream));
    ND4_TRY(nd4_copy_matrix(h, N, N, W, N, Wp, Np, batch, sM, sMp));
    ND4_TRY(nd4_set_identity(h, Np, Np, Utpad, Np, batch, sMp));
  }
  double* Wb = padded ? Wp : W;                                         // what the block sweeps work on
  double* Utb = padded ? Utpad : Ut;

  const int n2 = (N + 1) & ~1;
  const double eps = 0x1p-52, tol = N * eps, tol2 = tol * tol;
  if (!small) {
    hipLaunchKernelGGL(jac_norms, dim3((unsigned)((N + 3) / 4), (unsigned)batch), dim3(256), 0, h->stream, W, N, sM, svr);
    hipLaunchKernelGGL(jac_floor, dim3((unsigned)batch), dim3(256), 0, h->stream, svr, N, tol2, floor2);
  }
  int sweeps = 0;
  unsigned long long last_off = 0, rot_seen = 0;
  // the first sweep of a large matrix rotates (nearly) every pair; afterwards the rotation count of the last sweep decides
  bool dense_phase = blocked && Np >= 512 && !precheck_always;
  if (small) {
    // all sweeps in ONE launch, one workgroup per matrix (jac_small); sweeps_max lives in the word behind `active`
    ND4_HIP(hipMemsetAsync(active, 0, 16, h->stream));
    if (N <= 32) hipLaunchKernelGGL(jac_small<32>, dim3((unsigned)batch), dim3(256), 0, h->stream, W, Ut, N, sM, tol2, floor2, active + 1, active, offmax, rot_total, MAX_SWEEPS);
    else         hipLaunchKernelGGL(jac_small<64>, dim3((unsigned)batch), dim3(256), 0, h->stream, W, Ut, N, sM, tol2, floor2, active + 1, active, offmax, rot_total, MAX_SWEEPS);
    ND4_HIP(hipGetLastError());
    ND4_HIP(hipMemcpyAsync(h_active, active, 24, hipMemcpyDeviceToHost, h->stream));
    ND4_HIP(hipStreamSynchronize(h->stream));
    sweeps = (int)h_active[1];
    last_off = *h_off;
    if (h_active[0] != 0) sweeps = MAX_SWEEPS;                        // some matrix did not converge
  } else
  if (N > 1) {
    for (;;) {
      ND4_HIP(hipMemsetAsync(active, 0, 16, h->stream));            // active + offmax (the rotation total runs on)
      if (blocked) {
        ND4_TRY(nd4_jacobi_block_sweep(h, batch, Np, Wb, Utb, st, floor2, tol2, offmax, bscratch, dense_phase));
      } else {
        for (int s = 0; s < n2 - 1; s++)
          hipLaunchKernelGGL(jac_step, dim3((unsigned)(n2 / 2), (unsigned)batch), dim3(256), 0, h->stream,
                             W, Ut, N, sM, n2, s, tol2, st, floor2, offmax);
      }
      hipLaunchKernelGGL(jac_sweep_end, dim3((unsigned)((batch + 63) / 64)), dim3(64), 0, h->stream, st, batch, active, rot_total);
      ND4_HIP(hipGetLastError());
      sweeps++;
      // The convergence flag is read back (one stream synchronisation) after every sweep for large matrices; a sweep of a
      // small matrix costs less than the round trip, so it is checked every 4th (N <= 64) / 2nd (N <= 256) sweep only:
      // converged matrices are skipped on the device anyway (JacState.done), an extra sweep over them rotates nothing.
      const int check_every = Np <= 64 ? 4 : (Np <= 256 ? 2 : 1);
      if (sweeps % check_every != 0 && sweeps < MAX_SWEEPS) continue;
      ND4_HIP(hipMemcpyAsync(h_active, active, 24, hipMemcpyDeviceToHost, h->stream));   // active, offmax, rotation total
      ND4_HIP(hipStreamSynchronize(h->stream));
      if (debug) {
        double r; unsigned long long bb = *h_off; memcpy(&r, &bb, 8);
        fprintf(stderr, "[nd4hip svd] sweep %d: active matrices %u, max |cos| found %.3e, rotations so far %llu\n", sweeps, h_active[0], sqrt(r),
                (unsigned long long)h_off[1]);
      }
      if (h_active[0] == 0) { last_off = *h_off; break; }
      last_off = *h_off;
      {
        const unsigned long long in_sweep = h_off[1] - rot_seen;                  // rotations of the sweeps since the last read-back
        rot_seen = h_off[1];
        dense_phase = dense_phase && check_every == 1 &&
                      (double)in_sweep > 0.25 * (double)h_active[0] * 0.5 * (double)N * (double)(N - 1);
      }
      if (sweeps >= MAX_SWEEPS) break;
    }
  }
  if (padded) {
    ND4_TRY(nd4_copy_matrix(h, N, N, Wp, Np, W, N, batch, sMp, sM));
    ND4_TRY(nd4_copy_matrix(h, N, N, Utpad, Np, Ut, N, batch, sMp, sM));
  }
  (void)svrp;
  // ---- epilogue (_svd_jac_post contract) ----
  hipLaunchKernelGGL(jac_norms, dim3((unsigned)((N + 3) / 4), (unsigned)batch), dim3(256), 0, h->stream, W, N, sM, svr);
  hipLaunchKernelGGL(jac_rank, dim3((unsigned)((N + 255) / 256), (unsigned)batch), dim3(256), 0, h->stream, svr, N, rank);
  hipLaunchKernelGGL(jac_emit, dim3((unsigned)N, (unsigned)batch), dim3(256), 0, h->stream, W, Ut, N, sM, svr, rank, V, Utp, sv);
  if (N < 128) {
    hipLaunchKernelGGL(jac_complete, dim3((unsigned)batch), dim3(1024), 0, h->stream, V, N, sM, sv, floor2, scratch);
    ND4_HIP(hipGetLastError());
  } else {
    // Rank-deficient input: the rows of V that belong to singular values at the noise floor are an arbitrary orthonormal
    // completion (the reference's are whatever its D&C leaves there). jac_complete builds them one by one on a single
    // workgroup (N^3 work: 5.5 s for a rank-1 2048^2 matrix); here they come from ONE full QR of the r valid right vectors:
    // qr_decomp_full(V[0:r,:]^T) = [+-V_r^T | Q2], and the rows of Q2^T are the completion.
    int* z0d = reinterpret_cast<int*>(scratch);                       // scratch: batch * N doubles, idle here
    int* z0h = reinterpret_cast<int*>(static_cast<char*>(pin) + 64);
    hipLaunchKernelGGL(jac_count_rank, dim3((unsigned)((batch + 63) / 64)), dim3(64), 0, h->stream, sv, N, floor2, z0d, batch);
    ND4_HIP(hipGetLastError());
    ND4_HIP(hipMemcpyAsync(z0h, z0d, sizeof(int) * (size_t)batch, hipMemcpyDeviceToHost, h->stream));
    ND4_HIP(hipStreamSynchronize(h->stream));
    for (int m = 0; m < batch; m++) {
      const int r = z0h[m];
      if (r >= N) continue;
      double* Vm = V + (size_t)m * sM;
      if (r == 0) { ND4_TRY(nd4_set_identity(h, N, N, Vm, N, 1, sM)); continue; }
      Nd4WsScope scope3(h);
      void* q = nullptr;
      ND4_TRY(nd4_ws_alloc(h, sizeof(double) * ((size_t)N * N + 2 * (size_t)N * r), &q));
      double* Qf = static_cast<double*>(q);
      double* VrT = Qf + (size_t)N * N;
      double* Rf = VrT + (size_t)N * r;
      ND4_TRY(nd4_transpose(h, r, N, Vm, N, VrT, r, 1, 0, 0));
      ND4_TRY(nd4_geqrf_q_ex(h, 1, N, r, VrT, Qf, Rf, true));
      ND4_TRY(nd4_transpose(h, N, N - r, Qf + r, N, Vm + (size_t)r * N, N, 1, 0, 0));
    }
  }
  ND4_TRY(nd4_transpose(h, N, N, Utp, N, U, N, batch, sM, sM));
  {
    double r; unsigned long long b = last_off; memcpy(&r, &b, 8);
    h->svd_sweeps = sweeps; h->svd_offnorm = sqrt(r); h->svd_rotations = N > 1 ? h_off[1] : 0;
    if (sweeps_out) *sweeps_out = sweeps;
    if (offnorm_out) *offnorm_out = h->svd_offnorm;
  }
  if (sweeps >= MAX_SWEEPS && N > 1) {
    ND4_HIP(hipStreamSynchronize(h->stream));
    nd4_set_error("nd4hip_dgesvdj: no convergence after %d sweeps", sweeps);
    return ND4HIP_ERR_NOCONV;
  }
  return 0;
}

}  // namespace

int nd4_gesvdj(nd4hip_handle* h, int64_t batch64, int64_t M64, int64_t N64, const double* A,
               double* U, double* sv, double* V, int* sweeps_out, double* offnorm_out) {
  ND4_CHECK_ARG((M64 < N64 ? M64 : N64) < 46000 && M64 < (1ll << 30) && N64 < (1ll << 30) && batch64 < 65536,
                "nd4_gesvdj: extent out of range");              // the Jacobi part works on min(M,N)^2; QR takes the long side
  const int M = (int)M64, N = (int)N64, batch = (int)batch64;
  const int L = M < N ? M : N;
  Nd4WsScope scope(h);
  const long sL = (long)L * L;
  void* p = nullptr;
  if (M == N) {
    ND4_TRY(nd4_ws_alloc(h, sizeof(double) * (size_t)batch * sL, &p));
    double* W = static_cast<double*>(p);
    ND4_HIP(hipMemcpyAsync(W, A, sizeof(double) * (size_t)batch * sL, hipMemcpyDeviceToDevice, h->stream));
    return jacobi_square(h, batch, N, W, U, sv, V, sweeps_out, offnorm_out);
  }
  const long sA = (long)M * N;                 // also the size of the thin Q
  if (M > N) {
    // A = Q R ; R = Ur S V  ->  U = Q Ur                         (svd_jac_2sided.js:44-47)
    ND4_TRY(nd4_ws_alloc(h, sizeof(double) * (size_t)batch * (sA + 2 * sL), &p));
    double* Q = static_cast<double*>(p); double* R = Q + (size_t)batch * sA; double* Ur = R + (size_t)batch * sL;
    ND4_TRY(nd4_geqrf_q(h, batch, M, N, A, Q, R));
    ND4_TRY(jacobi_square(h, batch, N, R, Ur, sv, V, sweeps_out, offnorm_out));
    return nd4_gemm(h, false, false, M, N, N, 1.0, Q, N, sA, Ur, N, sL, 0.0, U, N, sA, batch);
  }
  // M < N:  A^T = Q R  ->  A = R^T Q^T ; R^T = U S Vr  ->  V = Vr Q^T   (svd_jac_2sided.js:48-52)
  ND4_TRY(nd4_ws_alloc(h, sizeof(double) * (size_t)batch * (2 * sA + 3 * sL), &p));
  double* At = static_cast<double*>(p); double* Q = At + (size_t)batch * sA;
  double* R = Q + (size_t)batch * sA; double* Rt = R + (size_t)batch * sL; double* Vr = Rt + (size_t)batch * sL;
  ND4_TRY(nd4_transpose(h, M, N, A, N, At, M, batch, sA, sA));
  ND4_TRY(nd4_geqrf_q(h, batch, N, M, At, Q, R));
  ND4_TRY(nd4_transpose(h, M, M, R, M, Rt, M, batch, sL, sL));
  ND4_TRY(jacobi_square(h, batch, M, Rt, U, sv, Vr, sweeps_out, offnorm_out));
  return nd4_gemm(h, false, true, M, N, M, 1.0, Vr, M, sL, Q, M, sA, 0.0, V, N, sA, batch);
}
