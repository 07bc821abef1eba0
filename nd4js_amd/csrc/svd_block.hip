// Block one-sided Jacobi sweeps for the SVD (N % 64 == 0): the HBM-bound row-pair rotations of
// svd.hip become GEMM-shaped work on the fp64 matrix cores.
//
// Rows are grouped in blocks of 32; a sweep visits every block pair (I,J) once (round-robin tournament
// over blocks, nblk/2 disjoint pairs per step). For each pair, X = [W_I; W_J] (64 x N):
//   jacb_gram    Gp[chunk] = X[:,chunk] X[:,chunk]^T      fp64 MFMA fed straight from global memory
//                (the A and B fragments of a Gram product are the SAME "16 rows x 4 k" register image;
//                 one 16-byte load feeds two MFMA k-steps); wave w owns tile-row w -> no reduction.
//   jacb_eigen   G = sum_chunks Gp; cyclic two-sided Jacobi on the 64x64 Gram matrix in LDS with the
//                reference's relative criterion (svd_jac_2sided.js:112, one-sided form) and the noise
//                floor of svd.hip; 32 disjoint rotations per round, each wave owns 8 of them (angles
//                computed SIMD across lanes, broadcast by readlane), row phase / column phase separated
//                by a barrier; the accumulated left transform Qt (64x64) goes to global memory.
//   jacb_apply   X <- Qt X and Ut_pair <- Qt Ut_pair on fp64 MFMA, 16-column strips: the 64 source rows
//                of a strip are loaded as B fragments (coalesced 128-B rows) before any store, so the
//                update is in place.
// Per sweep: 12 N^3 flop on MFMA and 40 N^3 / 32 bytes of traffic instead of 32 N^3 bytes.
#include "svd_internal.h"
#include <cstdlib>

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int BB = 32;         // rows per block
constexpr int PB = 64;         // rows per block pair
constexpr int CH = 256;        // columns per workgroup (gram and apply)
constexpr int MAX_INNER_DEFAULT = 1;   // inner Jacobi sweeps per visit of a pair (measured: the outer sweep count does not
                                       // depend on it, 15-16 at N=2048 for 1, 2 and 4)

// wave-uniform broadcast of lane K's value through SGPRs (v_readlane_b32): no LDS traffic, unlike __shfl
template <int K> __device__ __forceinline__ int bcast_i(int v) { return __builtin_amdgcn_readlane(v, K); }
template <int K> __device__ __forceinline__ double bcast_d(double v) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), K), __builtin_amdgcn_readlane(__double2loint(v), K));
}

// few-ulp reciprocal / reciprocal square root: hardware estimate + two Newton steps. The Jacobi angle
// only needs s and tau = s/(1+c) to be mutually consistent to a few ulp (orthogonality defect ~ theta^2 * ulp),
// so IEEE-exact division (~35 dependent fp64 instructions each) is not worth its latency here.
__device__ __forceinline__ double fast_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}
__device__ __forceinline__ double fast_rsqrt(double x) {
  double r = __builtin_amdgcn_rsq(x);
  r = r * fma(-0.5 * x * r, r, 1.5);
  r = r * fma(-0.5 * x * r, r, 1.5);
  return r;
}

__device__ __forceinline__ long pair_row(int x, int I, int J) { return x < BB ? (long)I * BB + x : (long)J * BB + (x - BB); }

// acc (16 tiles of the 64x64 Gram of the block pair) over the columns [col0, col0 + ncols) of W, for one wave.
// The A and B fragments of a Gram product are the same register image; one 16-byte load feeds two MFMA k-steps.
__device__ __forceinline__ void gram_accumulate(d4 (&acc)[4][4], const double* const (&rp)[4], int col0, int ncols, int N, int fk) {
#pragma unroll 8
  for (int k8 = 0; k8 < ncols / 8; k8++) {
    const int c = col0 + k8 * 8 + 2 * fk;
    d2 f[4];
#pragma unroll
    for (int t = 0; t < 4; t++) f[t] = (c < N) ? *reinterpret_cast<const d2*>(rp[t] + c) : d2{0.0, 0.0};
    // the Gram matrix is symmetric: only the 10 tiles with j >= i are computed (mirrored in gram_reduce_lds)
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = i; j < 4; j++) {
        acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[i].x, f[j].x, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[i].y, f[j].y, acc[i][j], 0, 0, 0);
      }
  }
}
// fixed-order (deterministic) reduction of the four waves' tiles into an LDS image G[64][ldg]
template <int LDG>
__device__ __forceinline__ void gram_reduce_lds(double (*G)[LDG], const d4 (&acc)[4][4], int wave, int fk, int fx) {
#pragma unroll
  for (int w = 0; w < 4; w++) {
    if (wave == w) {
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = i; j < 4; j++)
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const int row = i * 16 + fk + 4 * r, col = j * 16 + fx;
            double* d = &G[row][col];
            const double v = (w == 0) ? acc[i][j][r] : *d + acc[i][j][r];
            *d = v;
            if (w == 3 && j > i) G[col][row] = v;                  // mirror the off-diagonal tiles once the sum is final
          }
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void jacb_gram(const double* __restrict__ Wm, int N, long sM, int nblk, int nblk2, int step,
                                                  const JacState* __restrict__ st, double* __restrict__ Gpart, int nchunks, long sG_mat) {
  __shared__ double s_g[PB][PB + 1];
  const int pairIdx = blockIdx.x, chunk = blockIdx.y, mat = blockIdx.z;
  if (st[mat].done) return;
  int I, J;
  nd4_rr_pair(nblk2, step, pairIdx, I, J);
  if (J >= nblk) return;
  const double* W = Wm + mat * sM;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fx = lane & 15, fk = lane >> 4;
  const double* rp[4];
#pragma unroll
  for (int t = 0; t < 4; t++) rp[t] = W + pair_row(t * 16 + fx, I, J) * N;
  // wave w owns columns [col0 + 64 w, +64) of the chunk and accumulates ALL 16 tiles over them: every
  // fragment is loaded by exactly one wave
  d4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};
  gram_accumulate(acc, rp, chunk * CH + wave * (CH / 4), CH / 4, N, fk);
  gram_reduce_lds<PB + 1>(s_g, acc, wave, fk, fx);
  double* G = Gpart + mat * sG_mat + ((long)pairIdx * nchunks + chunk) * (PB * PB);
  for (int e = threadIdx.x; e < PB * PB; e += 256) G[e] = s_g[e / PB][e % PB];
}

// FUSED: the workgroup computes the Gram matrix of its block pair itself (small N: the MFMA phase of one
// workgroup overlaps the LDS-bound rotation rounds of the other workgroup on the CU, and the partial-Gram
// round trip through HBM disappears); Gpart then carries W and nchunks carries N.
template <bool FUSED>
__global__ __launch_bounds__(256) void jacb_eigen(const double* __restrict__ Gpart, int nchunks, long sG_mat, int nblk, int nblk2, int step,
                                                   JacState* __restrict__ st, const double* __restrict__ floor2, double tol2,
                                                   double* __restrict__ Qt_all, long sQ_mat, int* __restrict__ flags, long sF_mat,
                                                   unsigned long long* __restrict__ offmax, int max_inner, int cross_only) {
  __shared__ double G[PB][PB + 1];
  __shared__ double Q[PB][PB + 1];
  __shared__ unsigned s_rot[4];
  const int pairIdx = blockIdx.x, mat = blockIdx.y;
  if (st[mat].done) return;
  int I, J;
  nd4_rr_pair(nblk2, step, pairIdx, I, J);
  if (J >= nblk) return;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  if (FUSED) {
    const int N = nchunks, fx = lane & 15, fk = lane >> 4;
    const double* W = Gpart + mat * sG_mat;
    const double* rp[4];
#pragma unroll
    for (int q = 0; q < 4; q++) rp[q] = W + pair_row(q * 16 + fx, I, J) * N;
    d4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = 0; j < 4; j++) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};
    const int per = ((N + 31) / 32) * 8;                      // columns per wave, multiple of 8
    gram_accumulate(acc, rp, wave * per, per, N, fk);
    gram_reduce_lds<PB + 1>(G, acc, wave, fk, fx);
    for (int e = t; e < PB * PB; e += 256) Q[e / PB][e % PB] = (e / PB == e % PB) ? 1.0 : 0.0;
  } else {
    const double* Gp = Gpart + mat * sG_mat + (long)pairIdx * nchunks * (PB * PB);
    for (int e = t; e < PB * PB; e += 256) {
      double s = 0.0;
      for (int ch = 0; ch < nchunks; ch++) s += Gp[(long)ch * (PB * PB) + e];
      G[e / PB][e % PB] = s;
      Q[e / PB][e % PB] = (e / PB == e % PB) ? 1.0 : 0.0;
    }
  }
  __syncthreads();
  const double fl = floor2[mat];
  unsigned total = 0;
  double relmax = 0.0;
  // converged pairs (the common case in the last sweeps) leave after one pass over the off-diagonal
  {
    int need = 0;
    for (int e = t; e < PB * PB; e += 256) {
      const int i = e / PB, j = e % PB;
      if (i < j) { const double a = G[i][i], b = G[j][j], g = G[i][j]; need |= (a > fl) && (b > fl) && (g * g > tol2 * a * b); }
    }
    need = __syncthreads_or(need);
    if (!need) { if (t == 0) flags[mat * sF_mat + pairIdx] = 0; return; }
  }
  // cross_only: the two blocks were orthogonalised internally on earlier visits of this sweep, so only the
  // 32 x 32 pairs (p in I, q in J) are visited: 32 rounds of 32 disjoint pairs (i, 32 + (i + r) % 32).
  const int nrounds = cross_only ? BB : PB - 1;
  for (int inner = 0; inner < max_inner; inner++) {
    unsigned rot = 0;
    for (int r = 0; r < nrounds; r++) {
      int p, q;
      if (cross_only) { p = wave * 8 + (lane & 7); q = BB + ((p + r) & (BB - 1)); }
      else nd4_rr_pair(PB, r, wave * 8 + (lane & 7), p, q);
      const double a = G[p][p], b = G[q][q], g = G[p][q];
      const bool go = (a > fl) && (b > fl) && (g * g > tol2 * a * b);
      double c = 1.0, s = 0.0;                              // kept as (s, tau = tan(theta/2)): see svd.hip jac_step
      if (go) {
        // t = sign(zeta) / (|zeta| + sqrt(1 + zeta^2)), zeta = (b-a)/(2g), rewritten without the division by g
        const double d = b - a, hh = 2.0 * g, rr = d * d + hh * hh;
        const double root = rr * fast_rsqrt(rr);
        const double tn = (((d < 0.0) != (g < 0.0)) ? -fabs(hh) : fabs(hh)) * fast_rcp(fabs(d) + root);
        const double cc = fast_rsqrt(1.0 + tn * tn);
        s = cc * tn;
        c = s * fast_rcp(1.0 + cc);                        // c now holds tau = tan(theta/2)
        relmax = fmax(relmax, (g * g) / (a * b));
      }
      rot += (unsigned)__popcll(__ballot(go && lane < 8));
      // wave-uniform rotation parameters of this wave's 8 pairs (SGPRs)
      double sk[8], ck[8]; int pk[8], qk[8];
#define ND4_BC(K) sk[K] = bcast_d<K>(s); ck[K] = bcast_d<K>(c); pk[K] = bcast_i<K>(p); qk[K] = bcast_i<K>(q);
      ND4_BC(0) ND4_BC(1) ND4_BC(2) ND4_BC(3) ND4_BC(4) ND4_BC(5) ND4_BC(6) ND4_BC(7)
#undef ND4_BC
      // ---- row phase: rows p,q of G and of Q (lane = column). The 8 pairs touch disjoint rows, so all
      // 32 reads are issued before the first write (the compiler cannot prove that by itself and would
      // serialise read->fma->write eight times: this loop is LDS-latency bound, not bandwidth bound).
      // An idle pair (s = 0) rewrites its rows unchanged.
      {
        double gp[8], gq[8], up[8], uq[8];
#pragma unroll
        for (int k = 0; k < 8; k++) { gp[k] = G[pk[k]][lane]; gq[k] = G[qk[k]][lane]; up[k] = Q[pk[k]][lane]; uq[k] = Q[qk[k]][lane]; }
#pragma unroll
        for (int k = 0; k < 8; k++) {
          G[pk[k]][lane] = gp[k] - sk[k] * (gq[k] + ck[k] * gp[k]);
          G[qk[k]][lane] = gq[k] + sk[k] * (gp[k] - ck[k] * gq[k]);
          Q[pk[k]][lane] = up[k] - sk[k] * (uq[k] + ck[k] * up[k]);
          Q[qk[k]][lane] = uq[k] + sk[k] * (up[k] - ck[k] * uq[k]);
        }
      }
      __syncthreads();
      // ---- column phase: columns p,q of G (lane = row) ----
      {
        double gp[8], gq[8];
#pragma unroll
        for (int k = 0; k < 8; k++) { gp[k] = G[lane][pk[k]]; gq[k] = G[lane][qk[k]]; }
#pragma unroll
        for (int k = 0; k < 8; k++) {
          G[lane][pk[k]] = gp[k] - sk[k] * (gq[k] + ck[k] * gp[k]);
          G[lane][qk[k]] = gq[k] + sk[k] * (gp[k] - ck[k] * gq[k]);
        }
      }
      __syncthreads();
    }
    if (lane == 0) s_rot[wave] = rot;
    __syncthreads();
    const unsigned tot = s_rot[0] + s_rot[1] + s_rot[2] + s_rot[3];
    __syncthreads();
    total += tot;
    if (tot == 0) break;
  }
  double* Qt = Qt_all + mat * sQ_mat + (long)pairIdx * (PB * PB);
  if (total) for (int e = t; e < PB * PB; e += 256) Qt[e] = Q[e / PB][e % PB];
  // max over the wave of the largest cos^2 that triggered a rotation
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) relmax = fmax(relmax, __shfl_xor(relmax, off));
  if (lane == 0 && relmax > 0.0) atomicMax(offmax, (unsigned long long)__double_as_longlong(relmax));
  if (t == 0) {
    flags[mat * sF_mat + pairIdx] = total ? 1 : 0;
    if (total) atomicAdd(&st[mat].rotations, total);
  }
}

// The same rotation rounds on 8 or 16 waves (512 / 1024 threads): each wave carries 4 or 2 of the 32 disjoint pairs of a
// round instead of 8 (2048^2: 115.7 ms with 4 waves, 98.1 with 8, 95.0 with 16).
// The rounds are LDS-latency bound (read 2 rows / 2 columns, rotate, write back, barrier), and with only nblk/2 workgroups
// in flight for a single large matrix the chip is mostly idle: twice the waves per workgroup hide that latency better.
template <int PW>                                      // pairs per wave: 4 -> 8 waves, 2 -> 16 waves
__global__ __launch_bounds__(2048 / PW) void jacb_eigen8(const double* __restrict__ Gpart, int nchunks, long sG_mat, int nblk, int nblk2, int step,
                                                   JacState* __restrict__ st, const double* __restrict__ floor2, double tol2,
                                                   double* __restrict__ Qt_all, long sQ_mat, int* __restrict__ flags, long sF_mat,
                                                   unsigned long long* __restrict__ offmax, int max_inner, int cross_only) {
  __shared__ double G[PB][PB + 1];
  __shared__ double Q[PB][PB + 1];
  __shared__ unsigned s_rot[32 / PW];
  const int pairIdx = blockIdx.x, mat = blockIdx.y;
  if (st[mat].done) return;
  int I, J;
  nd4_rr_pair(nblk2, step, pairIdx, I, J);
  if (J >= nblk) return;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  {
    const double* Gp = Gpart + mat * sG_mat + (long)pairIdx * nchunks * (PB * PB);
    for (int e = t; e < PB * PB; e += 2048 / PW) {
      double s = 0.0;
      for (int ch = 0; ch < nchunks; ch++) s += Gp[(long)ch * (PB * PB) + e];
      G[e / PB][e % PB] = s;
      Q[e / PB][e % PB] = (e / PB == e % PB) ? 1.0 : 0.0;
    }
  }
  __syncthreads();
  const double fl = floor2[mat];
  unsigned total = 0;
  double relmax = 0.0;
  // converged pairs (the common case in the last sweeps) leave after one pass over the off-diagonal
  {
    int need = 0;
    for (int e = t; e < PB * PB; e += 2048 / PW) {
      const int i = e / PB, j = e % PB;
      if (i < j) { const double a = G[i][i], b = G[j][j], g = G[i][j]; need |= (a > fl) && (b > fl) && (g * g > tol2 * a * b); }
    }
    need = __syncthreads_or(need);
    if (!need) { if (t == 0) flags[mat * sF_mat + pairIdx] = 0; return; }
  }
  // cross_only: the two blocks were orthogonalised internally on earlier visits of this sweep, so only the
  // 32 x 32 pairs (p in I, q in J) are visited: 32 rounds of 32 disjoint pairs (i, 32 + (i + r) % 32).
  const int nrounds = cross_only ? BB : PB - 1;
  for (int inner = 0; inner < max_inner; inner++) {
    unsigned rot = 0;
    for (int r = 0; r < nrounds; r++) {
      int p, q;
      if (cross_only) { p = wave * PW + (lane & (PW - 1)); q = BB + ((p + r) & (BB - 1)); }
      else nd4_rr_pair(PB, r, wave * PW + (lane & (PW - 1)), p, q);
      const double a = G[p][p], b = G[q][q], g = G[p][q];
      const bool go = (a > fl) && (b > fl) && (g * g > tol2 * a * b);
      double c = 1.0, s = 0.0;                              // kept as (s, tau = tan(theta/2)): see svd.hip jac_step
      if (go) {
        // t = sign(zeta) / (|zeta| + sqrt(1 + zeta^2)), zeta = (b-a)/(2g), rewritten without the division by g
        const double d = b - a, hh = 2.0 * g, rr = d * d + hh * hh;
        const double root = rr * fast_rsqrt(rr);
        const double tn = (((d < 0.0) != (g < 0.0)) ? -fabs(hh) : fabs(hh)) * fast_rcp(fabs(d) + root);
        const double cc = fast_rsqrt(1.0 + tn * tn);
        s = cc * tn;
        c = s * fast_rcp(1.0 + cc);                        // c now holds tau = tan(theta/2)
        relmax = fmax(relmax, (g * g) / (a * b));
      }
      rot += (unsigned)__popcll(__ballot(go && lane < PW));
      // wave-uniform rotation parameters of this wave's 8 pairs (SGPRs)
      double sk[PW], ck[PW]; int pk[PW], qk[PW];
#define ND4_BC(K) sk[K] = bcast_d<K>(s); ck[K] = bcast_d<K>(c); pk[K] = bcast_i<K>(p); qk[K] = bcast_i<K>(q);
      ND4_BC(0) ND4_BC(1)
      if constexpr (PW > 2) { ND4_BC(2) ND4_BC(3) }
#undef ND4_BC
      // ---- row phase: rows p,q of G and of Q (lane = column). The 8 pairs touch disjoint rows, so all
      // 32 reads are issued before the first write (the compiler cannot prove that by itself and would
      // serialise read->fma->write eight times: this loop is LDS-latency bound, not bandwidth bound).
      // An idle pair (s = 0) rewrites its rows unchanged.
      {
        double gp[PW], gq[PW], up[PW], uq[PW];
#pragma unroll
        for (int k = 0; k < PW; k++) { gp[k] = G[pk[k]][lane]; gq[k] = G[qk[k]][lane]; up[k] = Q[pk[k]][lane]; uq[k] = Q[qk[k]][lane]; }
#pragma unroll
        for (int k = 0; k < PW; k++) {
          G[pk[k]][lane] = gp[k] - sk[k] * (gq[k] + ck[k] * gp[k]);
          G[qk[k]][lane] = gq[k] + sk[k] * (gp[k] - ck[k] * gq[k]);
          Q[pk[k]][lane] = up[k] - sk[k] * (uq[k] + ck[k] * up[k]);
          Q[qk[k]][lane] = uq[k] + sk[k] * (up[k] - ck[k] * uq[k]);
        }
      }
      __syncthreads();
      // ---- column phase: columns p,q of G (lane = row) ----
      {
        double gp[PW], gq[PW];
#pragma unroll
        for (int k = 0; k < PW; k++) { gp[k] = G[lane][pk[k]]; gq[k] = G[lane][qk[k]]; }
#pragma unroll
        for (int k = 0; k < PW; k++) {
          G[lane][pk[k]] = gp[k] - sk[k] * (gq[k] + ck[k] * gp[k]);
          G[lane][qk[k]] = gq[k] + sk[k] * (gp[k] - ck[k] * gq[k]);
        }
      }
      __syncthreads();
    }
    if (lane == 0) s_rot[wave] = rot;
    __syncthreads();
    unsigned tot = 0;
    for (int w = 0; w < 32 / PW; w++) tot += s_rot[w];
    __syncthreads();
    total += tot;
    if (tot == 0) break;
  }
  double* Qt = Qt_all + mat * sQ_mat + (long)pairIdx * (PB * PB);
  if (total) for (int e = t; e < PB * PB; e += 2048 / PW) Qt[e] = Q[e / PB][e % PB];
  // max over the wave of the largest cos^2 that triggered a rotation
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) relmax = fmax(relmax, __shfl_xor(relmax, off));
  if (lane == 0 && relmax > 0.0) atomicMax(offmax, (unsigned long long)__double_as_longlong(relmax));
  if (t == 0) {
    flags[mat * sF_mat + pairIdx] = total ? 1 : 0;
    if (total) atomicAdd(&st[mat].rotations, total);
  }
}

__global__ __launch_bounds__(256) void jacb_apply(double* __restrict__ Wm, double* __restrict__ Utm, int N, long sM, int nblk, int nblk2, int step,
                                                   const JacState* __restrict__ st, const double* __restrict__ Qt_all, long sQ_mat,
                                                   const int* __restrict__ flags, long sF_mat, int nchunks) {
  __shared__ double sQ[PB][PB + 4];
  const int pairIdx = blockIdx.x, mat = blockIdx.z;
  int chunk = blockIdx.y;
  if (st[mat].done || !flags[mat * sF_mat + pairIdx]) return;
  int I, J;
  nd4_rr_pair(nblk2, step, pairIdx, I, J);
  if (J >= nblk) return;
  double* X = (chunk < nchunks ? Wm : Utm) + mat * sM;
  if (chunk >= nchunks) chunk -= nchunks;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int fx = lane & 15, fk = lane >> 4;
  const double* Qt = Qt_all + mat * sQ_mat + (long)pairIdx * (PB * PB);
  for (int e = t; e < PB * PB; e += 256) sQ[e / PB][e % PB] = Qt[e];
  __syncthreads();
  long rowoff[16];                                   // source row of k-step ks for this lane
#pragma unroll
  for (int ks = 0; ks < 16; ks++) rowoff[ks] = pair_row(ks * 4 + fk, I, J) * N;
  const int colw = chunk * CH + wave * 64;
  for (int strip = 0; strip < 4; strip++) {
    const int cbase = colw + strip * 16;
    if (cbase >= N) break;                           // wave-uniform (N % 16 == 0)
    const int c = cbase + fx;
    double b[16];
#pragma unroll
    for (int ks = 0; ks < 16; ks++) b[ks] = X[rowoff[ks] + c];
#pragma unroll
    for (int it = 0; it < 4; it++) {
      d4 acc = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int ks = 0; ks < 16; ks++)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sQ[it * 16 + fx][ks * 4 + fk], b[ks], acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; r++) X[pair_row(it * 16 + fk + 4 * r, I, J) * N + c] = acc[r];
    }
  }
}

}  // namespace

size_t nd4_jacobi_block_scratch_doubles(int batch, int N) {
  const int nblk = N / BB, nblk2 = (nblk + 1) & ~1, npairs = nblk2 / 2, nchunks = (N + CH - 1) / CH;
  const size_t per = (size_t)npairs * nchunks * PB * PB + (size_t)npairs * PB * PB + (size_t)((npairs + 1) / 2 + 1);
  return per * batch;
}

int nd4_jacobi_block_sweep(nd4hip_handle* h, int batch, int N, double* W, double* Ut, JacState* st,
                           const double* floor2, double tol2, unsigned long long* offmax, double* scratch) {
  const int nblk = N / BB, nblk2 = (nblk + 1) & ~1, npairs = nblk2 / 2, nchunks = (N + CH - 1) / CH;
  const long sM = (long)N * N;
  const long sG = (long)npairs * nchunks * PB * PB, sQ = (long)npairs * PB * PB;
  double* Gpart = scratch;
  double* Qt = Gpart + (size_t)batch * sG;
  int* flags = reinterpret_cast<int*>(Qt + (size_t)batch * sQ);
  const long sF = ((npairs + 1) / 2 + 1) * 2;          // ints per matrix
  static const int max_inner = getenv("ND4HIP_JAC_INNER") ? atoi(getenv("ND4HIP_JAC_INNER")) : MAX_INNER_DEFAULT;
  static const int cross = getenv("ND4HIP_JAC_CROSS") ? atoi(getenv("ND4HIP_JAC_CROSS")) : 1;
  for (int step = 0; step < nblk2 - 1; step++) {
    static const int fuse_env = getenv("ND4HIP_JAC_FUSE") ? atoi(getenv("ND4HIP_JAC_FUSE")) : -1;
    // fused Gram+eigen pays when the (pair, matrix) workgroups alone fill the chip about once or twice; with many
    // more of them the separate, fully parallel Gram launch hides its load latency better (measured at N = 512:
    // batch 64: 83 -> 52 ms fused; batch 128: 94 -> 102; batch 1024: 738 -> 790)
    const bool fused = fuse_env >= 0 ? fuse_env != 0 : (N <= 1024 && batch * npairs >= 128 && batch * npairs <= 768);
    if (fused) {
      hipLaunchKernelGGL(jacb_eigen<true>, dim3((unsigned)npairs, (unsigned)batch), dim3(256), 0, h->stream,
                         W, N, sM, nblk, nblk2, step, st, floor2, tol2, Qt, sQ, flags, sF, offmax, max_inner,
                         (cross && step > 0) ? 1 : 0);
    } else {
      hipLaunchKernelGGL(jacb_gram, dim3((unsigned)npairs, (unsigned)nchunks, (unsigned)batch), dim3(256), 0, h->stream,
                         W, N, sM, nblk, nblk2, step, st, Gpart, nchunks, sG);
      // ND4HIP_JAC_EIGEN8 = 0 (4 waves always) | 8 | 16; default: 16 waves when there are few workgroups
      static const int e8 = getenv("ND4HIP_JAC_EIGEN8") ? atoi(getenv("ND4HIP_JAC_EIGEN8")) : -1;
      const bool eight = e8 >= 0 ? e8 != 0 : ((long)batch * npairs <= 256);     // few workgroups: latency, not throughput, matters
      if (eight && e8 != 8)
        hipLaunchKernelGGL(jacb_eigen8<2>, dim3((unsigned)npairs, (unsigned)batch), dim3(1024), 0, h->stream,
                           Gpart, nchunks, sG, nblk, nblk2, step, st, floor2, tol2, Qt, sQ, flags, sF, offmax, max_inner,
                           (cross && step > 0) ? 1 : 0);
      else if (eight)
        hipLaunchKernelGGL(jacb_eigen8<4>, dim3((unsigned)npairs, (unsigned)batch), dim3(512), 0, h->stream,
                           Gpart, nchunks, sG, nblk, nblk2, step, st, floor2, tol2, Qt, sQ, flags, sF, offmax, max_inner,
                           (cross && step > 0) ? 1 : 0);
      else
      hipLaunchKernelGGL(jacb_eigen<false>, dim3((unsigned)npairs, (unsigned)batch), dim3(256), 0, h->stream,
                         Gpart, nchunks, sG, nblk, nblk2, step, st, floor2, tol2, Qt, sQ, flags, sF, offmax, max_inner,
                         (cross && step > 0) ? 1 : 0);
    }
    hipLaunchKernelGGL(jacb_apply, dim3((unsigned)npairs, (unsigned)(2 * nchunks), (unsigned)batch), dim3(256), 0, h->stream,
                       W, Ut, N, sM, nblk, nblk2, step, st, Qt, sQ, flags, sF, nchunks);
  }
  ND4_HIP(hipGetLastError());
  return 0;
}
