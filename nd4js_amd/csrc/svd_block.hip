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

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int BB = 32;         // rows per block
constexpr int PB = 64;         // rows per block pair
constexpr int CH = 256;        // columns per workgroup (gram and apply)
constexpr int MAX_INNER = 3;   // inner Jacobi sweeps per visit of a pair

__device__ __forceinline__ long pair_row(int x, int I, int J) { return x < BB ? (long)I * BB + x : (long)J * BB + (x - BB); }

__global__ __launch_bounds__(256) void jacb_gram(const double* __restrict__ Wm, int N, long sM, int nblk, int nblk2, int step,
                                                  const JacState* __restrict__ st, double* __restrict__ Gpart, int nchunks, long sG_mat) {
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
  d4 acc[4];
#pragma unroll
  for (int j = 0; j < 4; j++) acc[j] = d4{0.0, 0.0, 0.0, 0.0};
  const int col0 = chunk * CH;
#pragma unroll 2
  for (int k8 = 0; k8 < CH / 8; k8++) {
    const int c = col0 + k8 * 8 + 2 * fk;
    d2 f[4];
#pragma unroll
    for (int t = 0; t < 4; t++) f[t] = (c < N) ? *reinterpret_cast<const d2*>(rp[t] + c) : d2{0.0, 0.0};
    const d2 fa = wave == 0 ? f[0] : (wave == 1 ? f[1] : (wave == 2 ? f[2] : f[3]));
#pragma unroll
    for (int j = 0; j < 4; j++) {
      acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa.x, f[j].x, acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa.y, f[j].y, acc[j], 0, 0, 0);
    }
  }
  double* G = Gpart + mat * sG_mat + ((long)pairIdx * nchunks + chunk) * (PB * PB);
#pragma unroll
  for (int j = 0; j < 4; j++)
#pragma unroll
    for (int r = 0; r < 4; r++) G[(wave * 16 + fk + 4 * r) * PB + j * 16 + fx] = acc[j][r];
}

__global__ __launch_bounds__(256) void jacb_eigen(const double* __restrict__ Gpart, int nchunks, long sG_mat, int nblk, int nblk2, int step,
                                                   JacState* __restrict__ st, const double* __restrict__ floor2, double tol2,
                                                   double* __restrict__ Qt_all, long sQ_mat, int* __restrict__ flags, long sF_mat,
                                                   unsigned long long* __restrict__ offmax) {
  __shared__ double G[PB][PB + 1];
  __shared__ double Q[PB][PB + 1];
  __shared__ unsigned s_rot[4];
  const int pairIdx = blockIdx.x, mat = blockIdx.y;
  if (st[mat].done) return;
  int I, J;
  nd4_rr_pair(nblk2, step, pairIdx, I, J);
  if (J >= nblk) return;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const double* Gp = Gpart + mat * sG_mat + (long)pairIdx * nchunks * (PB * PB);
  for (int e = t; e < PB * PB; e += 256) {
    double s = 0.0;
    for (int ch = 0; ch < nchunks; ch++) s += Gp[(long)ch * (PB * PB) + e];
    G[e / PB][e % PB] = s;
    Q[e / PB][e % PB] = (e / PB == e % PB) ? 1.0 : 0.0;
  }
  __syncthreads();
  const double fl = floor2[mat];
  unsigned total = 0;
  double relmax = 0.0;
  for (int inner = 0; inner < MAX_INNER; inner++) {
    unsigned rot = 0;
    for (int r = 0; r < PB - 1; r++) {
      int p, q;
      nd4_rr_pair(PB, r, wave * 8 + (lane & 7), p, q);
      const double a = G[p][p], b = G[q][q], g = G[p][q];
      const bool go = (a > fl) && (b > fl) && (g * g > tol2 * a * b);
      double c = 1.0, s = 0.0;                              // kept as (s, tau = tan(theta/2)): see svd.hip jac_step
      if (go) {
        const double zeta = (b - a) / (2.0 * g);
        const double tn = copysign(1.0, zeta) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        const double cc = 1.0 / sqrt(1.0 + tn * tn);
        s = cc * tn;
        c = s / (1.0 + cc);                                // c now holds tau
        relmax = fmax(relmax, (g * g) / (a * b));
      }
      rot += (unsigned)__popcll(__ballot(go && lane < 8));
      // ---- row phase: rows p,q of G and of Q (lane = column) ----
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const double sk = __shfl(s, k, 64);
        if (sk != 0.0) {                                   // wave-uniform
          const double ck = __shfl(c, k, 64);
          const int pk = __shfl(p, k, 64), qk = __shfl(q, k, 64);
          const double gp = G[pk][lane], gq = G[qk][lane];
          G[pk][lane] = gp - sk * (gq + ck * gp);
          G[qk][lane] = gq + sk * (gp - ck * gq);
          const double up = Q[pk][lane], uq = Q[qk][lane];
          Q[pk][lane] = up - sk * (uq + ck * up);
          Q[qk][lane] = uq + sk * (up - ck * uq);
        }
      }
      __syncthreads();
      // ---- column phase: columns p,q of G (lane = row) ----
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const double sk = __shfl(s, k, 64);
        if (sk != 0.0) {
          const double ck = __shfl(c, k, 64);
          const int pk = __shfl(p, k, 64), qk = __shfl(q, k, 64);
          const double gp = G[lane][pk], gq = G[lane][qk];
          G[lane][pk] = gp - sk * (gq + ck * gp);
          G[lane][qk] = gq + sk * (gp - ck * gq);
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
  for (int step = 0; step < nblk2 - 1; step++) {
    hipLaunchKernelGGL(jacb_gram, dim3((unsigned)npairs, (unsigned)nchunks, (unsigned)batch), dim3(256), 0, h->stream,
                       W, N, sM, nblk, nblk2, step, st, Gpart, nchunks, sG);
    hipLaunchKernelGGL(jacb_eigen, dim3((unsigned)npairs, (unsigned)batch), dim3(256), 0, h->stream,
                       Gpart, nchunks, sG, nblk, nblk2, step, st, floor2, tol2, Qt, sQ, flags, sF, offmax);
    hipLaunchKernelGGL(jacb_apply, dim3((unsigned)npairs, (unsigned)(2 * nchunks), (unsigned)batch), dim3(256), 0, h->stream,
                       W, Ut, N, sM, nblk, nblk2, step, st, Qt, sQ, flags, sF, nchunks);
  }
  ND4_HIP(hipGetLastError());
  return 0;
}
