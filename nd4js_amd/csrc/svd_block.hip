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
// One inner Jacobi sweep per visit of a pair (measured: the outer sweep count does not depend on it, 15-16 at N=2048 for 1, 2, 4).

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

// Block pair (I < J) of slot `i` at step `s` of a sweep over n2 blocks. n2 > 0: the round-robin tournament of nd4_rr_pair.
// n2 < 0 (n = -n2, n % 4 == 0): the block-recursive tournament of round 3, in which the slots [0, n/4) and [n/4, n/2) of every step
// are two CLOSED groups for a whole phase (so that two chains of launches can run side by side, see nd4_jacobi_block_sweep):
//   phase 1 (n/2 - 1 steps): round robin inside each half;  phase 2a (n/4 steps): H1a x H2a | H1b x H2b;  phase 2b (n/4 steps):
//   H1a x H2b | H1b x H2a (cyclic shifts). Every pair once per sweep, n - 1 steps, as before.
__host__ __device__ __forceinline__ void jacb_pair(int n2, int s, int i, int& I, int& J) {
  if (n2 > 0) { nd4_rr_pair(n2, s, i, I, J); return; }
  const int n = -n2, h = n / 2, q = n / 4;
  const int grp = i >= q ? 1 : 0, ii = i - grp * q;
  if (s < h - 1) {
    nd4_rr_pair(h, s, ii, I, J);
    I += grp * h; J += grp * h;
  } else {
    const int c = s - (h - 1), second = c >= q ? 1 : 0, cc = c - second * q;
    I = grp * q + ii;
    J = h + ((grp ^ second) ? q : 0) + (ii + cc) % q;
  }
}
__host__ __device__ __forceinline__ int jacb_abs(int n2) { return n2 < 0 ? -n2 : n2; }

// acc (16 tiles of the 64x64 Gram of the block pair) over the columns [col0, col0 + ncols) of W, for one wave.
// The A and B fragments of a Gram product are the same register image; one 16-byte load feeds two MFMA k-steps.
// The order in which the columns are summed is free (A and B fragment are the same registers), so lane (fx, fk) takes the
// 4 consecutive columns 4 fk .. 4 fk + 3 of every group of 16: the four lanes of a row then cover one whole 128-byte line with
// their two 16-byte loads.
__device__ __forceinline__ void gram_load_group(d2 (&dst)[2][4], const double* const (&rp)[4], int c) {
#pragma unroll
  for (int t = 0; t < 4; t++) {
    dst[0][t] = *reinterpret_cast<const d2*>(rp[t] + c);
    dst[1][t] = *reinterpret_cast<const d2*>(rp[t] + c + 2);
  }
}
__device__ __forceinline__ void gram_mfma_group(d4 (&acc)[4][4], const d2 (&f)[2][4]) {
  // the Gram matrix is symmetric: only the 10 tiles with j >= i are computed (mirrored in gram_reduce_lds)
#pragma unroll
  for (int u = 0; u < 2; u++)
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = i; j < 4; j++) {
        acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[u][i].x, f[u][j].x, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[u][i].y, f[u][j].y, acc[i][j], 0, 0, 0);
      }
}
// NG groups of 16 columns starting at col0, all inside the matrix; the loads of group g+1 are in flight while the 40 MFMAs of
// group g run (straight-line code: the register indices of the two fragment buffers are static)
template <int NG>
__device__ __forceinline__ void gram_accumulate_fixed(d4 (&acc)[4][4], const double* const (&rp)[4], int col0, int fk) {
  d2 f[2][2][4];
  gram_load_group(f[0], rp, col0 + 4 * fk);
#pragma unroll
  for (int g = 0; g < NG; g++) {
    if (g + 1 < NG) gram_load_group(f[(g + 1) & 1], rp, col0 + (g + 1) * 16 + 4 * fk);
    gram_mfma_group(acc, f[g & 1]);
  }
}
// any number of columns (a multiple of 16), clipped at N (a multiple of 16)
__device__ __forceinline__ void gram_accumulate(d4 (&acc)[4][4], const double* const (&rp)[4], int col0, int ncols, int N, int fk) {
  for (int c0 = col0; c0 < col0 + ncols && c0 < N; c0 += 16) {
    d2 f[2][4];
    gram_load_group(f, rp, c0 + 4 * fk);
    gram_mfma_group(acc, f);
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
  jacb_pair(nblk2, step, pairIdx, I, J);
  if (J >= nblk) return;
  const double* W = Wm + mat * sM;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
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
  const int col0 = chunk * CH + wave * (CH / 4);
  if (col0 + CH / 4 <= N) gram_accumulate_fixed<CH / 64>(acc, rp, col0, fk);   // N % 64 == 0: a wave's strip is inside or outside
  gram_reduce_lds<PB + 1>(s_g, acc, wave, fk, fx);
  double* G = Gpart + mat * sG_mat + ((long)pairIdx * nchunks + chunk) * (PB * PB);
  for (int e = threadIdx.x; e < PB * PB; e += 256) G[e] = s_g[e / PB][e % PB];
}

// ---- Gram partials, second form: tiles per wave, fragments through LDS -------------------------------------------------------
// jacb_gram gives every wave its own columns and ALL ten tiles: each wave then loads four 16-row fragments straight from global
// memory in MFMA layout (16 rows x 64 B per instruction) and the four waves' accumulators meet in a 4-phase LDS reduction.
// Here the 64 x 256 chunk is staged through LDS in four sub-chunks of 64 columns (coalesced: 32 lanes read one 512-B row segment),
// double-buffered, and each wave owns 2-3 of the ten tiles over ALL columns: no cross-wave reduction, every wave stores its
// tiles itself. LDS image [64][66]: lane (fx, fk) of a fragment read hits bank (4 fx + 2 fk) mod 64 -> conflict-free b64 reads.
template <int NT>
__device__ __forceinline__ void gram2_tiles(d4 (&acc)[3], const double (*st)[PB + 2], const int (&ti)[3], const int (&tj)[3], int fx, int fk) {
#pragma unroll
  for (int ks = 0; ks < 16; ks++) {
    double f[4];
#pragma unroll
    for (int q = 0; q < 4; q++) f[q] = st[q * 16 + fx][ks * 4 + fk];
#pragma unroll
    for (int n = 0; n < NT; n++) acc[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[ti[n]], f[tj[n]], acc[n], 0, 0, 0);
  }
}
__global__ __launch_bounds__(256) void jacb_gram2(const double* __restrict__ Wm, int N, long sM, int nblk, int nblk2, int step,
                                                   const JacState* __restrict__ st, double* __restrict__ Gpart, int nchunks, long sG_mat, int pair0) {
  __shared__ double s_x[2][PB][PB + 2];
  const int pairIdx = blockIdx.x + pair0, chunk = blockIdx.y, mat = blockIdx.z;
  if (st[mat].done) return;
  int I, J;
  jacb_pair(nblk2, step, pairIdx, I, J);
  if (J >= nblk) return;
  const double* W = Wm + mat * sM;
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int fx = lane & 15, fk = lane >> 4;
  // staging: d2 number q = t + 256 k of a 64 x 64 sub-chunk: row q >> 5, columns 2 (q & 31)
  const double* src[8];
#pragma unroll
  for (int k = 0; k < 8; k++) src[k] = W + pair_row((t >> 5) + 8 * k, I, J) * N + chunk * CH + 2 * (t & 31);
  const int nsub = (N - chunk * CH >= CH) ? CH / 64 : (N - chunk * CH) / 64;      // N % 64 == 0, wave-uniform
  // Measured with parts of the kernel switched off: the load stream alone 7 us (4.8 TB/s), + MFMAs 11 us (10 tiles x 64 k-steps x
  // 64 cycles over 4 SIMDs = 4.8 us per CU, only partly in the shadow of the loads), + the stores 13 us. One sub-chunk ahead is the
  // best prefetch distance: with all four sub-chunks requested up front (4 LDS buffers) the kernel takes 14.2 us.
  d2 v[8];
  auto load = [&](int sub) {
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = *reinterpret_cast<const d2*>(src[k] + sub * 64);
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int k = 0; k < 8; k++) *reinterpret_cast<d2*>(&s_x[buf][(t >> 5) + 8 * k][2 * (t & 31)]) = v[k];
  };
  d4 acc[3] = {d4{0.0, 0.0, 0.0, 0.0}, d4{0.0, 0.0, 0.0, 0.0}, d4{0.0, 0.0, 0.0, 0.0}};
  if (nsub > 0) { load(0); stash(0); }
  __syncthreads();
  for (int sub = 0; sub < nsub; sub++) {
    if (sub + 1 < nsub) load(sub + 1);
    const double (*sx)[PB + 2] = s_x[sub & 1];
    // tiles (i, j), i <= j, of the 4 x 4 tile grid: 3, 3, 2, 2 per wave
    if (wave == 0)      { const int ti[3] = {0, 0, 0}, tj[3] = {0, 1, 2}; gram2_tiles<3>(acc, sx, ti, tj, fx, fk); }
    else if (wave == 1) { const int ti[3] = {0, 1, 1}, tj[3] = {3, 1, 2}; gram2_tiles<3>(acc, sx, ti, tj, fx, fk); }
    else if (wave == 2) { const int ti[3] = {1, 2, 2}, tj[3] = {3, 2, 2}; gram2_tiles<2>(acc, sx, ti, tj, fx, fk); }
    else                { const int ti[3] = {2, 3, 3}, tj[3] = {3, 3, 3}; gram2_tiles<2>(acc, sx, ti, tj, fx, fk); }
    if (sub + 1 < nsub) stash((sub + 1) & 1);
    __syncthreads();
  }
  // Only the ten tiles with i <= j are stored (the consumers mirror after summing the chunks): the mirrored stores were
  // 8-byte writes 512 B apart, 2 us of the kernel.
  const int TI[4][3] = {{0, 0, 0}, {0, 1, 1}, {1, 2, 2}, {2, 3, 3}};
  const int TJ[4][3] = {{0, 1, 2}, {3, 1, 2}, {3, 2, 2}, {3, 3, 3}};
  double* G = Gpart + mat * sG_mat + ((long)pairIdx * nchunks + chunk) * (PB * PB);
  const int ntiles = wave < 2 ? 3 : 2;
#pragma unroll
  for (int n = 0; n < 3; n++) {
    if (n < ntiles) {
      const int i = TI[wave][n], j = TJ[wave][n];
#pragma unroll
      for (int r = 0; r < 4; r++) G[(i * 16 + fk + 4 * r) * PB + j * 16 + fx] = acc[n][r];
    }
  }
}

// Two-phase rotation rounds (row phase, barrier, column phase, barrier) on the 64 x 64 Gram matrix in LDS: the full sweep of
// step 0 and the mid-size batches. PW = pairs of a round per wave: 8 -> 4 waves (throughput: many workgroups per CU), 2 -> 16
// waves (2048^2: 115.7 ms with 4 waves, 98.1 with 8, 95.0 with 16 when this kernel ran every step).
// FUSED: the workgroup computes the Gram matrix of its block pair itself (small N: the MFMA phase of one workgroup overlaps the
// LDS-bound rotation rounds of the other workgroup on the CU, and the partial-Gram round trip through HBM disappears); Gpart
// then carries W and nchunks carries N.
// The rounds are LDS-latency bound (read 2 rows / 2 columns, rotate, write back, barrier), and with only nblk/2 workgroups
// in flight for a single large matrix the chip is mostly idle: twice the waves per workgroup hide that latency better.
template <bool FUSED, int PW>
__global__ __launch_bounds__(2048 / PW) void jacb_eigen(const double* __restrict__ Gpart, int nchunks, long sG_mat, int nblk, int nblk2, int step,
                                                   JacState* __restrict__ st, const double* __restrict__ floor2, double tol2,
                                                   double* __restrict__ Qt_all, long sQ_mat, int* __restrict__ flags, long sF_mat,
                                                   unsigned long long* __restrict__ offmax, int max_inner, int cross_only) {
  __shared__ double G[PB][PB + 1];
  __shared__ double Q[PB][PB + 1];
  __shared__ unsigned s_rot[32 / PW];
  const int pairIdx = blockIdx.x, mat = blockIdx.y;
  if (st[mat].done) return;
  int I, J;
  jacb_pair(nblk2, step, pairIdx, I, J);
  if (J >= nblk) return;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  if constexpr (FUSED) {
    static_assert(!FUSED || PW == 8, "the fused Gram phase is written for 4 waves");
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
    const int per = ((N + 63) / 64) * 16;                     // columns per wave, multiple of 16
    gram_accumulate(acc, rp, wave * per, per, N, fk);
    gram_reduce_lds<PB + 1>(G, acc, wave, fk, fx);
    for (int e = t; e < PB * PB; e += 2048 / PW) Q[e / PB][e % PB] = (e / PB == e % PB) ? 1.0 : 0.0;
  } else {
    const double* Gp = Gpart + mat * sG_mat + (long)pairIdx * nchunks * (PB * PB);
    for (int e = t; e < PB * PB; e += 2048 / PW) {
      const int r = e / PB, c = e % PB;
      Q[r][c] = (r == c) ? 1.0 : 0.0;
      if ((r >> 4) > (c >> 4)) continue;                     // the partials hold the ten 16 x 16 tiles on and above the diagonal
      double s = 0.0;
      for (int ch = 0; ch < nchunks; ch++) s += Gp[(long)ch * (PB * PB) + e];
      G[r][c] = s;
      if ((r >> 4) != (c >> 4)) G[c][r] = s;
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
      if constexpr (PW > 4) { ND4_BC(4) ND4_BC(5) ND4_BC(6) ND4_BC(7) }
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

// ---- cross-only rotation rounds: one barrier per round, a dedicated pivot wave ----------------------------------------------
// The 32 rounds of a cross-only visit are a dependent chain (the angles of round r+1 need the Gram matrix after round r), so the
// visit costs 32 x the latency of one round. jacb_eigen(8) below spend ~3600 cycles per round: a row phase and a column phase
// through LDS with a barrier after each, every wave recomputing angles it then broadcasts. Three observations remove most of it:
//  * With ALL 32 angles of a round known, G' = R G R^T can be formed block-wise: the 2x2 block (row pair i, column pair j) of G'
//    depends only on the same block of G and on the angles of i and j. No separate row and column phase.
//  * Thread (row pair i, column pairs j) always owns the same entries of the A quadrant (rows and columns of block I keep their
//    place): they stay in registers. C = G[I][J] and B = G[J][J] are double-buffered in LDS (read one buffer, write the other), so
//    the reads and writes of a round need no barrier between them. The C^T quadrant is never stored (symmetry). Rows of Q that
//    belong to block I stay in registers too, rows of block J pass through LDS.
//  * Measured (tools/lat_f64.hip): one wave issues one fp64 instruction per ~8 cycles whether or not it depends on the previous
//    one, so what a round costs is the instruction count of its busiest wave. When every wave computes the angle chain (~50
//    instructions) before its blocks, a round is ~1800 cycles (that version, jacb_eigen_x, took 36 us per visit against 54 us).
// Here the chain runs on ONE wave while eight others rotate: in interval k (between two barriers)
//   pivot wave : pivots of S_k from the near-diagonal entries of S_{k-1} and the angles of round k-1 (diagonal blocks in closed
//                form, the off-diagonal pivot = one entry of block (i, i+1)), then the angle chain of round k -> LDS
//   bulk waves : S_k = R_{k-1} S_{k-1} R_{k-1}^T and Q_k = R_{k-1} Q_{k-1}, angles of round k-1 read from LDS
// Both read state k-1 and write state k (double-buffered), so one barrier per round remains and its critical path is
// max(pivot, bulk) instead of their sum (~1300 cycles, 28 us per visit). The pivot wave keeps its own a, b, g in registers; they
// differ from the bulk's copies by rounding only, and only the pivot wave's decide the angles that G and Q are BOTH rotated
// with, so G stays consistent with the accumulated Q.
// Angle formulas: c^2 = (1 + |d|/root)/2, s = g/(root*c), t = s/c, tau = s/(1+c) with d = b - a, root = sqrt(d^2 + 4 g^2): two
// rsqrt chains and one rcp chain, no IEEE division.
template <int CG>   // chunks of partial Gram matrices in flight at once: 8 for one large matrix (one memory round trip at 2048^2), 4 for batches (86
                    // instead of 150 VGPRs: two workgroups per CU, which is what the throughput-bound batches need: 1024 x 512^2 0.69 -> 0.59 s)
__device__ __forceinline__ void eigen_p_body(double* __restrict__ stage_raw, const int pairIdx, const int mat,
                                             const double* __restrict__ Gpart, int nchunks, long sG_mat, int nblk, int nblk2, int step,
                                             JacState* __restrict__ st, const double* __restrict__ floor2, double tol2,
                                             double* __restrict__ Qt_all, long sQ_mat, int* __restrict__ flags, long sF_mat,
                                             unsigned long long* __restrict__ offmax, int precheck) {
  constexpr int T = 576, TB = 512, NB = 2, NQ = 4;
  constexpr int LD = BB + 1;
  __shared__ double sC[2][BB][LD];        // C(x, y) = G[x][32 + y]
  __shared__ double sB[2][BB][LD];        // B(x, y) = G[32 + x][32 + y] (both triangles)
  __shared__ double sA1[2][BB];           // A(i, i+1): the one entry of the A quadrant the pivot wave needs
  __shared__ double sAng[2][2][BB];       // [round parity][s | tau][pair]
  double (*stage)[PB + 1] = reinterpret_cast<double (*)[PB + 1]>(stage_raw);   // the summed Gram matrix on entry; afterwards its first 32 rows hold QJ (rows of Q of block J)
  __shared__ double sRd[PB];              // 1 / diagonal (pre-check)
  __shared__ unsigned sTotal;
  double (*QJ)[PB + 1] = stage;
  if (st[mat].done) return;
  int I, J;
  jacb_pair(nblk2, step, pairIdx, I, J);
  if (J >= nblk) return;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  if (t < TB) {
    // sum of the column chunks' partial Gram matrices, chunk order fixed; 16-byte loads, all of a chunk group in flight at once
    constexpr int KL = (PB * PB / 2) / TB;
    const d2* Gp = reinterpret_cast<const d2*>(Gpart + mat * sG_mat + (long)pairIdx * nchunks * (PB * PB));
    d2 acc[KL];
#pragma unroll
    for (int k = 0; k < KL; k++) acc[k] = d2{0.0, 0.0};
    for (int ch0 = 0; ch0 < nchunks; ch0 += CG) {
      d2 v[CG][KL];
#pragma unroll
      for (int c = 0; c < CG; c++)
#pragma unroll
        for (int k = 0; k < KL; k++) {
          const int e = 2 * (t + TB * k);                    // the partials hold the ten 16 x 16 tiles on and above the diagonal
          v[c][k] = (ch0 + c < nchunks && ((e / PB) >> 4) <= ((e % PB) >> 4)) ? Gp[(long)(ch0 + c) * (PB * PB / 2) + t + TB * k] : d2{0.0, 0.0};
        }
#pragma unroll
      for (int c = 0; c < CG; c++)
#pragma unroll
        for (int k = 0; k < KL; k++) acc[k] += v[c][k];
    }
#pragma unroll
    for (int k = 0; k < KL; k++) {
      const int e = 2 * (t + TB * k), r = e / PB, c = e % PB;
      if ((r >> 4) <= (c >> 4)) {
        stage[r][c] = acc[k].x;
        stage[r][c + 1] = acc[k].y;
        if ((r >> 4) != (c >> 4)) { stage[c][r] = acc[k].x; stage[c + 1][r] = acc[k].y; }
      }
    }
  }
  __syncthreads();
  const double fl = floor2[mat];
  // The pass over the off-diagonal (largest cos^2 of the pair as it is now = the off-norm the driver reports; converged pairs
  // leave here) costs 3.7 us of the visit. In the dense phase of the iteration every pair rotates anyway: the driver switches it
  // off while the previous sweep applied more than a quarter of all possible rotations.
  if (precheck) {
    if (t < PB) { const double dd = stage[t][t]; sRd[t] = dd > fl ? __builtin_amdgcn_rcp(dd) : 0.0; }   // 0: row at the noise floor
    __syncthreads();
    double rel = 0.0;
    for (int e = t; e < PB * PB; e += T) {
      const int x = e / PB, y = e % PB;
      const double g = stage[x][y];
      if (x < y) rel = fmax(rel, g * g * sRd[x] * sRd[y]);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) rel = fmax(rel, __shfl_xor(rel, off));
    if (lane == 0 && rel > 0.0) atomicMax(offmax, (unsigned long long)__double_as_longlong(rel));
    const int need = __syncthreads_or(rel > tol2 * (1.0 - 0x1p-20));
    if (!need) { if (t == 0) flags[mat * sF_mat + pairIdx] = 0; return; }
  }
  const int i = lane & 31, h = lane >> 5;
  const bool pivot = wave == 8;
  const int jb = NB * (2 * wave + h);     // bulk: this thread's column pairs jb, jb+1
  const int cb = NQ * (2 * wave + h);     // and its 4 columns of Q
  double Areg[NB] = {0.0, 0.0}, Qp[NQ] = {0.0, 0.0, 0.0, 0.0};
  double pa = 0.0, pb = 0.0, pg = 0.0;    // pivot wave: A(i,i), B(ip,ip), C(i,ip) of the current state
  if (pivot) {
    pa = stage[i][i]; pb = stage[BB + i][BB + i]; pg = stage[i][BB + i];
  } else {
#pragma unroll
    for (int m = 0; m < NB; m++) Areg[m] = stage[i][jb + m];
#pragma unroll
    for (int m = 0; m < NQ; m++) Qp[m] = (cb + m == i) ? 1.0 : 0.0;
  }
  for (int e = t; e < BB * BB; e += T) {
    const int x = e / BB, y = e % BB;
    sC[0][x][y] = stage[x][BB + y];
    sB[0][x][y] = stage[BB + x][BB + y];
  }
  if (t < BB) sA1[0][t] = stage[t][(t + 1) & (BB - 1)];
  __syncthreads();                        // everybody is done with `stage`
  for (int e = t; e < BB * PB; e += T) QJ[e / PB][e % PB] = (e % PB == BB + e / PB) ? 1.0 : 0.0;
  // interval k = 0 .. 32: the pivot wave produces the angles of round k (k < 32), the bulk applies round k-1 (k > 0)
  unsigned rot = 0;
  double ps = 0.0, ptau = 0.0, sn = 0.0, tn = 0.0;   // pivot wave: angle of round k-1 for pair i and for pair i+1
  for (int k = 0; k <= BB; k++) {
    const int prv = (k + 1) & 1, cur = k & 1;                     // state k-1 lives in buffer prv, state k goes to cur
    if (pivot) {
      if (k > 0) {
        // pg <- entry z_pq of block (i, i+1) of round k-1 (the neighbour's angle was fetched at the end of the last interval)
        const int ipp = (i + k - 1) & (BB - 1), y = (i + k) & (BB - 1), i1 = (i + 1) & (BB - 1);
        const double xpp = sA1[prv][i], xpq = sC[prv][i][y], xqp = sC[prv][i1][ipp], xqq = sB[prv][ipp][y];
        const double ypq = xpq - ps * (xqq + ptau * xpq), ypp = xpp - ps * (xqp + ptau * xpp);
        pg = ypq + sn * (ypp - tn * ypq);
      }
      if (k < BB) {
        const double a = pa, b = pb, g = pg;
        const bool go = (a > fl) && (b > fl) && (g * g > tol2 * a * b);
        const double d = b - a, hh = g + g, rr = fma(d, d, hh * hh);
        const double rs = fast_rsqrt(rr);                       // 1 / root
        const double x = fma(0.5 * fabs(d), rs, 0.5);           // c^2 in [1/2, 1]
        const double ric = fast_rsqrt(x);                       // 1 / c
        const double c = x * ric;
        const double sabs = fabs(g) * rs * ric;                 // |s| = |g| / (root c)
        double s = ((d < 0.0) != (g < 0.0)) ? -sabs : sabs;
        double tau = s * fast_rcp(1.0 + c);
        double tt = s * ric;                                    // tan(theta)
        if (!go) { s = 0.0; tau = 0.0; tt = 0.0; }
        rot += (unsigned)__popc((unsigned)__ballot(go));
        ps = s; ptau = tau;
        if (h == 0) { sAng[cur][0][i] = s; sAng[cur][1][i] = tau; }
        // diagonal blocks in closed form: A(i,i) and B(ip,ip) after this round; the latter is the next pivot b of pair i-1,
        // and pair i+1's angle turns the columns of block (i, i+1): neighbour exchange now, hidden behind the barrier
        const int i1 = (i + 1) & (BB - 1);
        pa = a - tt * g;
        pb = __shfl(b + tt * g, i1);
        sn = __shfl(s, i1); tn = __shfl(tau, i1);
      }
    } else if (k > 0) {
      const int r = k - 1;
      const int ip = (i + r) & (BB - 1);
      const double s = sAng[prv][0][i], tau = sAng[prv][1][i];
      double xpq[NB], xqp[NB], xqq[NB], sjv[NB], tjv[NB], uq[NQ];
      int jp[NB];
#pragma unroll
      for (int m = 0; m < NB; m++) {
        jp[m] = (jb + m + r) & (BB - 1);
        xpq[m] = sC[prv][i][jp[m]];
        xqp[m] = sC[prv][jb + m][ip];
        xqq[m] = sB[prv][ip][jp[m]];
        sjv[m] = sAng[prv][0][jb + m];
        tjv[m] = sAng[prv][1][jb + m];
      }
#pragma unroll
      for (int m = 0; m < NQ; m++) uq[m] = QJ[ip][cb + m];
#pragma unroll
      for (int m = 0; m < NB; m++) {
        const double sj = sjv[m], tj = tjv[m];
        const double xpp = Areg[m];
        const double ypp = xpp - s * (xqp[m] + tau * xpp), yqp = xqp[m] + s * (xpp - tau * xqp[m]);
        const double ypq = xpq[m] - s * (xqq[m] + tau * xpq[m]), yqq = xqq[m] + s * (xpq[m] - tau * xqq[m]);
        const double zpp = ypp - sj * (ypq + tj * ypp);
        const double zpq = ypq + sj * (ypp - tj * ypq);
        const double zqq = yqq + sj * (yqp - tj * yqq);
        Areg[m] = zpp;
        if (jb + m == ((i + 1) & (BB - 1))) sA1[cur][i] = zpp;
        sC[cur][i][jp[m]] = zpq;
        sB[cur][ip][jp[m]] = zqq;
      }
#pragma unroll
      for (int m = 0; m < NQ; m++) {
        const double up = Qp[m];
        Qp[m] = up - s * (uq[m] + tau * up);
        QJ[ip][cb + m] = uq[m] + s * (up - tau * uq[m]);
      }
    }
    __syncthreads();
  }
  if (t == 8 * 64) sTotal = rot;
  __syncthreads();
  const unsigned total = sTotal;
  if (total) {
    double* Qt = Qt_all + mat * sQ_mat + (long)pairIdx * (PB * PB);
    if (!pivot) {
#pragma unroll
      for (int m = 0; m < NQ; m++) Qt[i * PB + cb + m] = Qp[m];
    }
    for (int e = t; e < BB * PB; e += T) Qt[BB * PB + e] = QJ[e / PB][e % PB];
  }
  if (t == 0) {
    flags[mat * sF_mat + pairIdx] = total ? 1 : 0;
    if (total) atomicAdd(&st[mat].rotations, total);
  }
}

// chunk in [0, nchunks): columns of W, [nchunks, 2 nchunks): columns of Ut. T threads load Qt, the first 256 do the products.
constexpr int SQ_LD = PB + 4;
constexpr int STAGE_DOUBLES = PB * SQ_LD;            // one LDS array serves as Qt here and as the Gram stage of the rotation kernel
template <int T, int STRIPS = 4>
__device__ __forceinline__ void apply_body(double* __restrict__ sq_raw, const int pairIdx, int chunk, const int mat,
                                           double* __restrict__ Wm, double* __restrict__ Utm, int N, long sM, int nblk, int nblk2, int step,
                                           const JacState* __restrict__ st, const double* __restrict__ Qt_all, long sQ_mat,
                                           const int* __restrict__ flags, long sF_mat, int nchunks) {
  double (*sQ)[SQ_LD] = reinterpret_cast<double (*)[SQ_LD]>(sq_raw);
  if (st[mat].done || !flags[mat * sF_mat + pairIdx]) return;
  int I, J;
  jacb_pair(nblk2, step, pairIdx, I, J);
  if (J >= nblk) return;
  double* X = (chunk < nchunks ? Wm : Utm) + mat * sM;
  if (chunk >= nchunks) chunk -= nchunks;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int fx = lane & 15, fk = lane >> 4;
  const double* Qt = Qt_all + mat * sQ_mat + (long)pairIdx * (PB * PB);
  for (int e = t; e < PB * PB; e += T) sQ[e / PB][e % PB] = Qt[e];
  __syncthreads();
  if (T > 256 && t >= 256) return;
  long rowoff[16];                                   // source row of k-step ks for this lane
#pragma unroll
  for (int ks = 0; ks < 16; ks++) rowoff[ks] = pair_row(ks * 4 + fk, I, J) * N;
  const int colw = chunk * (64 * STRIPS) + wave * (16 * STRIPS);     // nchunks counts chunks of 64 * STRIPS columns
  for (int strip = 0; strip < STRIPS; strip++) {
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

__global__ __launch_bounds__(576) void jacb_eigen_p(const double* __restrict__ Gpart, int nchunks, long sG_mat, int nblk, int nblk2, int step,
                                                     JacState* __restrict__ st, const double* __restrict__ floor2, double tol2,
                                                     double* __restrict__ Qt_all, long sQ_mat, int* __restrict__ flags, long sF_mat,
                                                     unsigned long long* __restrict__ offmax, int precheck, int pair0) {
  __shared__ double stage_raw[STAGE_DOUBLES];
  eigen_p_body<4>(stage_raw, blockIdx.x + pair0, blockIdx.y, Gpart, nchunks, sG_mat, nblk, nblk2, step, st, floor2, tol2, Qt_all, sQ_mat, flags, sF_mat,
               offmax, precheck);
}

__global__ __launch_bounds__(256) void jacb_apply(double* __restrict__ Wm, double* __restrict__ Utm, int N, long sM, int nblk, int nblk2, int step,
                                                   const JacState* __restrict__ st, const double* __restrict__ Qt_all, long sQ_mat,
                                                   const int* __restrict__ flags, long sF_mat, int nchunks, int chunk0, int pair0) {
  __shared__ double sq_raw[STAGE_DOUBLES];
  apply_body<256>(sq_raw, blockIdx.x + pair0, blockIdx.y + chunk0, blockIdx.z, Wm, Utm, N, sM, nblk, nblk2, step, st, Qt_all, sQ_mat, flags, sF_mat, nchunks);
}

// The W half alone (the deferred form below): narrower column chunks, or the launch has one workgroup per CU and is latency-bound
template <int STRIPS>
__global__ __launch_bounds__(256) void jacb_apply_w(double* __restrict__ Wm, int N, long sM, int nblk, int nblk2, int step,
                                                     const JacState* __restrict__ st, const double* __restrict__ Qt_all, long sQ_mat,
                                                     const int* __restrict__ flags, long sF_mat, int pair0) {
  __shared__ double sq_raw[STAGE_DOUBLES];
  apply_body<256, STRIPS>(sq_raw, blockIdx.x + pair0, blockIdx.y, blockIdx.z, Wm, nullptr, N, sM, nblk, nblk2, step, st, Qt_all, sQ_mat, flags, sF_mat,
                          (int)gridDim.y);
}

// Rotation kernel of step `step` + the deferred U update of step - 1 in ONE launch. Only W is on the critical path of a step (the next
// Gram needs it); Ut <- Qt Ut is independent of everything until the sweep ends. With one large matrix the rotation kernel keeps only
// npairs workgroups (of 256 CUs) busy for its whole latency-bound run, so the Ut half of the previous step's apply (half of its bytes)
// rides along as npairs * nchunks extra workgroups of this launch. Qt and flags alternate between two sets by step parity: the
// rotation workgroups write set step & 1 while the U workgroups read set (step - 1) & 1. (The same overlap through a side stream and
// events was slower than no overlap: 76 against 64 ms at 2048^2 - the cross-stream dependencies cost more than the overlap gains.)
__global__ __launch_bounds__(576) void jacb_eigen_pu(const double* __restrict__ Gpart, int nchunks, long sG_mat, int nblk, int nblk2, int step,
                                                      JacState* __restrict__ st, const double* __restrict__ floor2, double tol2,
                                                      double* __restrict__ Qt_all, long sQ_mat, int* __restrict__ flags, long sF_mat,
                                                      unsigned long long* __restrict__ offmax, int precheck, int npairs,
                                                      double* __restrict__ Utm, int N, long sM, const double* __restrict__ Qt_prev,
                                                      const int* __restrict__ flags_prev, int pair0) {
  __shared__ double stage_raw[STAGE_DOUBLES];
  if ((int)blockIdx.x < npairs) {
    eigen_p_body<8>(stage_raw, blockIdx.x + pair0, blockIdx.y, Gpart, nchunks, sG_mat, nblk, nblk2, step, st, floor2, tol2, Qt_all, sQ_mat, flags, sF_mat,
                 offmax, precheck);
  } else {
    const int idx = (int)blockIdx.x - npairs;
    apply_body<576>(stage_raw, idx % npairs + pair0, nchunks + idx / npairs, blockIdx.y, nullptr, Utm, N, sM, nblk, nblk2, step - 1, st, Qt_prev, sQ_mat,
                    flags_prev, sF_mat, nchunks);
  }
}

}  // namespace

size_t nd4_jacobi_block_scratch_doubles(int batch, int N) {
  const int nblk = N / BB, nblk2 = (nblk + 1) & ~1, npairs = nblk2 / 2, nchunks = (N + CH - 1) / CH;
  const size_t per = (size_t)npairs * nchunks * PB * PB + 2 * ((size_t)npairs * PB * PB + (size_t)((npairs + 1) / 2 + 1));
  return per * batch;                                // Gram partials + two sets of (Qt, flags): see jacb_eigen_pu
}

int nd4_jacobi_block_sweep(nd4hip_handle* h, int batch, int N, double* W, double* Ut, JacState* st,
                           const double* floor2, double tol2, unsigned long long* offmax, double* scratch, bool dense_phase) {
  const int nblk = N / BB, nblk2 = (nblk + 1) & ~1, npairs = nblk2 / 2, nchunks = (N + CH - 1) / CH;
  const long sM = (long)N * N;
  const long sG = (long)npairs * nchunks * PB * PB, sQ = (long)npairs * PB * PB;
  double* Gpart = scratch;
  const long sF = ((npairs + 1) / 2 + 1) * 2;          // ints per matrix
  double* Qt2[2]; int* flags2[2];
  Qt2[0] = Gpart + (size_t)batch * sG;
  flags2[0] = reinterpret_cast<int*>(Qt2[0] + (size_t)batch * sQ);
  Qt2[1] = Qt2[0] + (size_t)batch * sQ + (size_t)batch * (sF / 2);
  flags2[1] = reinterpret_cast<int*>(Qt2[1] + (size_t)batch * sQ);
  static const bool no_defer = getenv("ND4HIP_JAC_NO_DEFER") != nullptr;           // A/B switch
  const bool defer = !no_defer && (long)batch * npairs <= 64 && N >= 512;            // see jacb_eigen_pu
  // Round 3: two chains of launches side by side. The rotation kernel of a step keeps only npairs workgroups busy for ~26 us (issue-
  // bound on their CUs) while Gram and the W update are throughput kernels; in ONE stream they cannot overlap (a launch that combines
  // the rotation kernel of one pair group with the Gram / update of another still has the rotation kernel in two of every three
  // launches). With the block-recursive tournament of jacb_pair the slots [0, npairs/2) and [npairs/2, npairs) of every step are two
  // closed groups for a whole phase, so each group's Gram -> rotation -> update chain runs on its own stream (chain A on the handle's
  // stream, chain B on aux_stream) and the rotation kernel of one chain runs beside the throughput kernels of the other. The chains
  // meet at the three phase ends of a sweep (one event pair each); the step before a meeting applies W and U together (no deferred
  // U update across a meeting). Single large matrix only.
  // Measured: 4096^2 317 -> 268 ms; 2048^2 57.5 -> 57.3 and 1024^2 22.6 -> 23.4 ms: there each of the three kernels of a step is ONE wave of
  // workgroups whose duration does not shrink with half the pairs (Gram 12.2 -> 11.5 us, W update 16.6 -> 18.7, and the rotation kernel
  // slows from 26 to 35 us when it shares the chip), so a chain's step takes as long as the whole step did. Hence N >= 4096 only
  // (ND4HIP_JAC_TWO_CHAINS=<min N> moves the threshold, 0 switches it off) — and, because the deferred U update needs
  // batch * npairs <= 64 (`defer`: N <= 4096 with 32-row blocks), in effect for padded N == 4096 alone: 8192^2 keeps one chain.
  const char* two_e = getenv("ND4HIP_JAC_TWO_CHAINS");      // (read per sweep, not cached: the tests move the threshold)
  const int two_env = two_e ? atoi(two_e) : 4096;
  const bool two_off = two_env <= 0;
  const int two_min_n = two_env > 0 ? two_env : 1 << 30;
  if (!two_off && defer && batch == 1 && nblk % 4 == 0 && N >= two_min_n) {
    const int n2 = -nblk, hq = nblk / 2, q = nblk / 4, nsteps = nblk - 1;
    static const int wstrips_env2 = getenv("ND4HIP_JAC_WSTRIPS") ? atoi(getenv("ND4HIP_JAC_WSTRIPS")) : 0;
    const int wstrips = wstrips_env2 ? wstrips_env2 : (N <= 2048 ? 1 : 2);
    hipStream_t chain[2] = {h->stream, h->aux_stream};
    auto meet = [&]() -> int {                               // both chains wait for each other
      ND4_HIP(hipEventRecord(h->ev_aux_a, chain[0]));
      ND4_HIP(hipEventRecord(h->ev_aux_b, chain[1]));
      ND4_HIP(hipStreamWaitEvent(chain[1], h->ev_aux_a, 0));
      ND4_HIP(hipStreamWaitEvent(chain[0], h->ev_aux_b, 0));
      return 0;
    };
    // step 0 (all pairs of the 64 rows, intra-block pairs included) on the handle's stream, as in the one-chain form
    {
      hipLaunchKernelGGL(jacb_gram2, dim3((unsigned)npairs, (unsigned)nchunks, 1u), dim3(256), 0, chain[0], W, N, sM, nblk, n2, 0, st, Gpart, nchunks, sG, 0);
      hipLaunchKernelGGL((jacb_eigen<false, 2>), dim3((unsigned)npairs, 1u), dim3(1024), 0, chain[0],
                         Gpart, nchunks, sG, nblk, n2, 0, st, floor2, tol2, Qt2[0], sQ, flags2[0], sF, offmax, 1, 0);
      hipLaunchKernelGGL(jacb_apply, dim3((unsigned)npairs, (unsigned)(2 * nchunks), 1u), dim3(256), 0, chain[0],
                         W, Ut, N, sM, nblk, n2, 0, st, Qt2[0], sQ, flags2[0], sF, nchunks, 0, 0);
      ND4_HIP(hipEventRecord(h->ev_aux_a, chain[0]));
      ND4_HIP(hipStreamWaitEvent(chain[1], h->ev_aux_a, 0));
    }
    bool prev_full = true;                                   // the previous step applied U itself (nothing deferred)
    for (int step = 1; step < nsteps; step++) {
      const bool phase_end = step == hq - 2 || step == hq - 2 + q || step == nsteps - 1;
      double* Qt = Qt2[step & 1];
      int* flags = flags2[step & 1];
      for (int c = 0; c < 2; c++) {
        const int p0 = c * q;
        hipLaunchKernelGGL(jacb_gram2, dim3((unsigned)q, (unsigned)nchunks, 1u), dim3(256), 0, chain[c], W, N, sM, nblk, n2, step, st, Gpart, nchunks, sG, p0);
        static const int pad2_kb = getenv("ND4HIP_JAC_EIGEN_PAD_KB") ? atoi(getenv("ND4HIP_JAC_EIGEN_PAD_KB")) : 0;
        if (prev_full)
          hipLaunchKernelGGL(jacb_eigen_p, dim3((unsigned)q, 1u), dim3(576), (size_t)pad2_kb * 1024, chain[c],
                             Gpart, nchunks, sG, nblk, n2, step, st, floor2, tol2, Qt, sQ, flags, sF, offmax, dense_phase ? 0 : 1, p0);
        else
          hipLaunchKernelGGL(jacb_eigen_pu, dim3((unsigned)(q + q * nchunks), 1u), dim3(576), (size_t)pad2_kb * 1024, chain[c],
                             Gpart, nchunks, sG, nblk, n2, step, st, floor2, tol2, Qt, sQ, flags, sF, offmax, dense_phase ? 0 : 1, q,
                             Ut, N, sM, Qt2[(step - 1) & 1], flags2[(step - 1) & 1], p0);
        if (phase_end)
          hipLaunchKernelGGL(jacb_apply, dim3((unsigned)q, (unsigned)(2 * nchunks), 1u), dim3(256), 0, chain[c],
                             W, Ut, N, sM, nblk, n2, step, st, Qt, sQ, flags, sF, nchunks, 0, p0);
        else if (wstrips == 1)
          hipLaunchKernelGGL(jacb_apply_w<1>, dim3((unsigned)q, (unsigned)(N / 64), 1u), dim3(256), 0, chain[c],
                             W, N, sM, nblk, n2, step, st, Qt, sQ, flags, sF, p0);
        else
          hipLaunchKernelGGL(jacb_apply_w<2>, dim3((unsigned)q, (unsigned)((N + 127) / 128), 1u), dim3(256), 0, chain[c],
                             W, N, sM, nblk, n2, step, st, Qt, sQ, flags, sF, p0);
      }
      prev_full = phase_end;
      if (phase_end) ND4_TRY(meet());
    }
    ND4_HIP(hipGetLastError());
    return 0;
  }
  for (int step = 0; step < nblk2 - 1; step++) {
    // step 0 of a sweep rotates all pairs of the 64 rows (the pairs inside a block are visited there, once per sweep); the later
    // steps only the 32 x 32 pairs across the two blocks
    const bool cross_only = step > 0;
    // fused Gram+eigen pays when the (pair, matrix) workgroups alone fill the chip about once or twice; with many
    // more of them the separate, fully parallel Gram launch hides its load latency better (measured at N = 512:
    // batch 64: 83 -> 52 ms fused; batch 128: 94 -> 102; batch 1024: 738 -> 790)
    const bool fused = N <= 1024 && batch * npairs >= 128 && batch * npairs <= 768;
    double* Qt = Qt2[defer ? (step & 1) : 0];
    int* flags = flags2[defer ? (step & 1) : 0];
    const bool last = step == nblk2 - 2;
    if (fused) {
      hipLaunchKernelGGL((jacb_eigen<true, 8>), dim3((unsigned)npairs, (unsigned)batch), dim3(256), 0, h->stream,
                         W, N, sM, nblk, nblk2, step, st, floor2, tol2, Qt, sQ, flags, sF, offmax, 1, cross_only ? 1 : 0);
    } else {
      static const bool gram1 = getenv("ND4HIP_JAC_GRAM1") != nullptr;             // A/B switch: the first form of the Gram kernel
      if (gram1)
        hipLaunchKernelGGL(jacb_gram, dim3((unsigned)npairs, (unsigned)nchunks, (unsigned)batch), dim3(256), 0, h->stream,
                           W, N, sM, nblk, nblk2, step, st, Gpart, nchunks, sG);
      else
        hipLaunchKernelGGL(jacb_gram2, dim3((unsigned)npairs, (unsigned)nchunks, (unsigned)batch), dim3(256), 0, h->stream,
                           W, N, sM, nblk, nblk2, step, st, Gpart, nchunks, sG, 0);
      if (cross_only && defer) {
        // dynamic LDS on top of the kernel's ~70 KB: above 80 KB in all a rotation workgroup has its CU to itself (the riding U
        // workgroups go elsewhere): its rounds are bound by the instruction issue of that one CU
        static const int pad_kb = getenv("ND4HIP_JAC_EIGEN_PAD_KB") ? atoi(getenv("ND4HIP_JAC_EIGEN_PAD_KB")) : 0;
        hipLaunchKernelGGL(jacb_eigen_pu, dim3((unsigned)(npairs + npairs * nchunks), (unsigned)batch), dim3(576), (size_t)pad_kb * 1024, h->stream,
                           Gpart, nchunks, sG, nblk, nblk2, step, st, floor2, tol2, Qt, sQ, flags, sF, offmax, dense_phase ? 0 : 1, npairs,
                           Ut, N, sM, Qt2[(step - 1) & 1], flags2[(step - 1) & 1], 0);
      } else if (cross_only) {
        hipLaunchKernelGGL(jacb_eigen_p, dim3((unsigned)npairs, (unsigned)batch), dim3(576), 0, h->stream,
                           Gpart, nchunks, sG, nblk, nblk2, step, st, floor2, tol2, Qt, sQ, flags, sF, offmax, dense_phase ? 0 : 1, 0);
      } else if ((long)batch * npairs <= 256) {            // few workgroups: latency matters, 16 waves hide more of it
        hipLaunchKernelGGL((jacb_eigen<false, 2>), dim3((unsigned)npairs, (unsigned)batch), dim3(1024), 0, h->stream,
                           Gpart, nchunks, sG, nblk, nblk2, step, st, floor2, tol2, Qt, sQ, flags, sF, offmax, 1, 0);
      } else {
        hipLaunchKernelGGL((jacb_eigen<false, 8>), dim3((unsigned)npairs, (unsigned)batch), dim3(256), 0, h->stream,
                           Gpart, nchunks, sG, nblk, nblk2, step, st, floor2, tol2, Qt, sQ, flags, sF, offmax, 1, 0);
      }
    }
    // deferred: W now, Ut with the next step's rotation kernel; the last step of a sweep has no successor and does both
    // 2048^2: 58.8 ms with 256-column chunks, 57.5 with 128, 57.2 with 64; 4096^2: 322 / 318 / 328
    static const int wstrips_env = getenv("ND4HIP_JAC_WSTRIPS") ? atoi(getenv("ND4HIP_JAC_WSTRIPS")) : 0;
    const int wstrips = wstrips_env ? wstrips_env : (N <= 2048 ? 1 : 2);
    if (defer && !last && wstrips == 1)
      hipLaunchKernelGGL(jacb_apply_w<1>, dim3((unsigned)npairs, (unsigned)(N / 64), (unsigned)batch), dim3(256), 0, h->stream,
                         W, N, sM, nblk, nblk2, step, st, Qt, sQ, flags, sF, 0);
    else if (defer && !last && wstrips == 2)
      hipLaunchKernelGGL(jacb_apply_w<2>, dim3((unsigned)npairs, (unsigned)((N + 127) / 128), (unsigned)batch), dim3(256), 0, h->stream,
                         W, N, sM, nblk, nblk2, step, st, Qt, sQ, flags, sF, 0);
    else
      hipLaunchKernelGGL(jacb_apply, dim3((unsigned)npairs, (unsigned)((defer && !last ? 1 : 2) * nchunks), (unsigned)batch), dim3(256), 0, h->stream,
                         W, Ut, N, sM, nblk, nblk2, step, st, Qt, sQ, flags, sF, nchunks, 0, 0);
  }
  ND4_HIP(hipGetLastError());
  return 0;
}
