// fp64 GEMM for gfx950 (MI355X): C = alpha * op(A) * op(B) + beta * C, row-major, batched.
//
// Replaces the reference's scalar i-k-j triple loop (src/la/matmul.js:49-53) and is the
// trailing-update engine of the blocked LU / QR / block-Jacobi kernels.
//
// Design:
//  * 256-thread workgroup = 4 wave64 in a 2x2 grid; macro tile 128x128, K-step 16;
//    each wave owns a 64x64 sub-tile = 4x4 accumulators of v_mfma_f64_16x16x4_f64
//    (16 independent 64-cycle MFMA chains per wave -> the matrix pipe never waits on a dependency).
//  * operand tiles are staged global -> registers -> LDS, double-buffered, ONE barrier per K-step:
//    the global loads of step t+1 are issued before the 64 MFMAs of step t and written to the other
//    LDS buffer after them. Two workgroups per CU (<= 256 VGPR, 72 KiB LDS each) cover each other's
//    barrier / write phases.
//  * LDS images are chosen so that every ds_read_b64 fragment read is bank-conflict free:
//      "row" image  [128][17]  (operand stored x-major, k contiguous): lane (x=l&15,k=l>>4) ->
//                   dword bank (34*x + 2*k) mod 64: 32 lanes x 2 dwords cover all 64 banks once;
//      "kmaj" image [16][144]  (operand stored k-major, x contiguous): row stride 288 dwords
//                   = 32 mod 64, so lanes 16-31 (next k) take the other half of the banks.
//  * workgroup -> tile map is XCD-aware (blocks b, b+8, ... share an XCD's L2): each XCD gets a
//    contiguous chunk of tiles, rasterised in groups of 8 tile-rows so the A/B panels it streams
//    stay in its 4 MiB L2.
//  * edges: rows/cols/k beyond the matrix are loaded as zeros (predicated loads) and never stored;
//    the 16-byte vector path needs even leading dimensions / extents / offsets, otherwise the
//    scalar path (8-byte loads) runs.
#include "nd4hip_internal.h"
#include <cstdlib>

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int BM = 128, BN = 128, BK = 16;
constexpr int LDR = BK + 1;       // row image leading dim (doubles)
constexpr int LDK = BN + 16;      // kmaj image leading dim (doubles)
constexpr int TILE = BK * LDK;    // 2304 doubles >= BM*LDR = 2176: one operand buffer
constexpr int NXCD = 8, GROUP_M = 8;

struct GemmArgs {
  const double* A; const double* B; double* C;
  int M, N, K;
  long lda, ldb, ldc, sA, sB, sC;
  double alpha, beta;
  int tiles_m, tiles_n;
  int lower;          // 1: skip tiles that lie entirely above the diagonal (symmetric rank-k updates)
  int kchunk;         // > 0: split-K, blockIdx.z owns k in [z*kchunk, (z+1)*kchunk) and writes its raw partial to Cpart
  double* Cpart;      //      [z][batch][M][N] (ld = N)
};

// ---- global -> register staging -------------------------------------------------------------
// ROW operand: stored [x][k] (k contiguous). thread t covers x = q*32 + t/8, k = 2*(t%8)+{0,1}
template <bool VEC, bool FULL>
__device__ __forceinline__ void load_row(d2 (&r)[4], const double* __restrict__ P, long ld,
                                         int x0, int X, int k0, int K, int t) {
  const int kc = k0 + ((t & 7) << 1);
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int x = x0 + q * 32 + (t >> 3);
    const double* p = P + (long)x * ld + kc;
    if (FULL) {
      r[q] = *reinterpret_cast<const d2*>(p);
    } else if (VEC) {
      d2 v = {0.0, 0.0};
      if (x < X && kc < K) v = *reinterpret_cast<const d2*>(p);
      r[q] = v;
    } else {
      double a = 0.0, b = 0.0;
      if (x < X && kc < K) a = p[0];
      if (x < X && kc + 1 < K) b = p[1];
      r[q].x = a; r[q].y = b;
    }
  }
}
__device__ __forceinline__ void store_row(double* S, const d2 (&r)[4], int t) {
  const int kc = (t & 7) << 1;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    double* s = S + (q * 32 + (t >> 3)) * LDR + kc;
    s[0] = r[q].x; s[1] = r[q].y;
  }
}
// KMAJ operand: stored [k][x] (x contiguous). thread t covers k = t/16, x = q*32 + 2*(t%16)+{0,1}
template <bool VEC, bool FULL>
__device__ __forceinline__ void load_kmaj(d2 (&r)[4], const double* __restrict__ P, long ld,
                                          int x0, int X, int k0, int K, int t) {
  const int k = k0 + (t >> 4);
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int x = x0 + q * 32 + ((t & 15) << 1);
    const double* p = P + (long)k * ld + x;
    if (FULL) {
      r[q] = *reinterpret_cast<const d2*>(p);
    } else if (VEC) {
      d2 v = {0.0, 0.0};
      if (k < K && x < X) v = *reinterpret_cast<const d2*>(p);
      r[q] = v;
    } else {
      double a = 0.0, b = 0.0;
      if (k < K && x < X) a = p[0];
      if (k < K && x + 1 < X) b = p[1];
      r[q].x = a; r[q].y = b;
    }
  }
}
__device__ __forceinline__ void store_kmaj(double* S, const d2 (&r)[4], int t) {
#pragma unroll
  for (int q = 0; q < 4; q++)
    *reinterpret_cast<d2*>(S + (t >> 4) * LDK + q * 32 + ((t & 15) << 1)) = r[q];
}

// TA: A is stored K x M (operand = transpose of the stored matrix); TB: B is stored N x K.
// FULL: M, N multiples of 128 and K a multiple of 16 (and VEC): no bounds predicate anywhere.
template <bool TA, bool TB, bool VEC, bool FULL>
__global__ __launch_bounds__(256, 2) void dgemm_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) double lds[4 * TILE];   // [buf][A|B][TILE]

  // ---- XCD-aware tile assignment (bijective for any tile count) ----
  const int nwg = g.tiles_m * g.tiles_n;
  int wg;
  {
    const int bid = blockIdx.x, xcd = bid % NXCD, within = bid / NXCD;
    const int q = nwg / NXCD, r = nwg % NXCD;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
  }
  const int per_group = GROUP_M * g.tiles_n;
  const int first_m = (wg / per_group) * GROUP_M;
  const int gsz = min(g.tiles_m - first_m, GROUP_M);
  const int tm = first_m + (wg % per_group) % gsz;
  const int tn = (wg % per_group) / gsz;
  const int m0 = tm * BM, n0 = tn * BN;
  if (g.lower && n0 > m0 + BM - 1) return;          // whole workgroup, before any barrier

  const long bz = blockIdx.y;
  const double* __restrict__ A = g.A + bz * g.sA;
  const double* __restrict__ B = g.B + bz * g.sB;
  double* __restrict__ C = g.C + bz * g.sC;
  int kbeg = 0, kend = g.K;
  long ldc = g.ldc;
  double alpha = g.alpha, beta = g.beta;
  if (g.kchunk > 0) {                                 // split-K: raw partial product of this k range
    kbeg = blockIdx.z * g.kchunk; kend = min(g.K, kbeg + g.kchunk);
    C = g.Cpart + ((long)blockIdx.z * gridDim.y + bz) * g.M * g.N;
    ldc = g.N; alpha = 1.0; beta = 0.0;
  }

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  const int fx = lane & 15, fk = lane >> 4;

  d4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};

  d2 ra[4], rb[4];
  auto gload = [&](int k0) {
    if (TA) load_kmaj<VEC, FULL>(ra, A, g.lda, m0, g.M, k0, kend, t); else load_row<VEC, FULL>(ra, A, g.lda, m0, g.M, k0, kend, t);
    if (TB) load_row<VEC, FULL>(rb, B, g.ldb, n0, g.N, k0, kend, t);  else load_kmaj<VEC, FULL>(rb, B, g.ldb, n0, g.N, k0, kend, t);
  };
  auto sstore = [&](int buf) {
    double* sa = lds + buf * 2 * TILE; double* sb = sa + TILE;
    if (TA) store_kmaj(sa, ra, t); else store_row(sa, ra, t);
    if (TB) store_row(sb, rb, t);  else store_kmaj(sb, rb, t);
  };

  const int nk = (kend - kbeg + BK - 1) / BK;
  gload(kbeg);
  sstore(0);
  __syncthreads();

  for (int kt = 0; kt < nk; kt++) {
    const int cur = kt & 1;
    if (kt + 1 < nk) gload(kbeg + (kt + 1) * BK);   // in flight during the MFMAs below
    const double* sa = lds + cur * 2 * TILE;
    const double* sb = sa + TILE;
#pragma unroll
    for (int kk = 0; kk < BK / 4; kk++) {
      double a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; i++)
        a[i] = TA ? sa[(kk * 4 + fk) * LDK + wm + i * 16 + fx] : sa[(wm + i * 16 + fx) * LDR + kk * 4 + fk];
#pragma unroll
      for (int j = 0; j < 4; j++)
        b[j] = TB ? sb[(wn + j * 16 + fx) * LDR + kk * 4 + fk] : sb[(kk * 4 + fk) * LDK + wn + j * 16 + fx];
      if (kk == BK / 4 - 1 && kt + 1 < nk) {
        // stage tile kt+1 into the other LDS buffer BEFORE the last 16 MFMAs: the LDS writes drain behind
        // them, so at the barrier nobody waits for vmcnt / the write pass (the other buffer is idle: its
        // last reader passed the previous barrier)
        sstore(cur ^ 1);
        __builtin_amdgcn_sched_barrier(0);
      }
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    }
    __syncthreads();
  }

  // ---- epilogue: lane holds C[row = (lane>>4) + 4r][col = lane&15] of each 16x16 tile ----
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int col = n0 + wn + j * 16 + fx;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int row = m0 + wm + i * 16 + fk + 4 * r;
        if (FULL || (row < g.M && col < g.N)) {
          double* c = C + (long)row * ldc + col;
          double v = alpha * acc[i][j][r];
          if (beta != 0.0) v += beta * *c;
          *c = v;
        }
      }
    }
}

// ---- rank-k update kernel: K <= 32, A not transposed --------------------------------------------------------------
// The trailing updates of LU / QR / Cholesky / LDL and of the blocked triangular solves are C (+)= alpha A op(B) with
// K = 16 or 32: two K-steps of the tiled kernel above, whose fixed cost (LDS staging, barriers, one workgroup per CU
// at 2048^2) is then most of the launch. Here nothing is staged: every wave owns 16 x 32 of C (SK_RT x SK_CT = 1 x 2 accumulators),
// reads its A rows and the B panel straight from global memory in MFMA operand layout (cache-line-complete, the four
// waves of a workgroup share the B panel through L1/L2), starts the accumulators at (beta/alpha) C (exact for the
// alpha = +-1, beta in {0, +-1} these callers use; other combinations take the tiled kernel), and writes alpha * acc.
// No LDS, no barrier, workgroups of 64 x 32 (4 waves): the launch is bound by the read-modify-write of C.
constexpr int SK_RT = 1;                                   // 16-row MFMA tiles per wave
constexpr int SK_CT = 2;                                   // 16-column MFMA tiles per wave
constexpr int SK_BM = 4 * 16 * SK_RT, SK_BN = 16 * SK_CT, SK_KSTEPS = 8;

template <bool TB>
__global__ __launch_bounds__(256, 4) void dgemm_smallk_kernel(GemmArgs g) {
  const int tm = blockIdx.x / g.tiles_n, tn = blockIdx.x % g.tiles_n;
  const int m0 = tm * SK_BM, n0 = tn * SK_BN;
  if (g.lower && n0 > m0 + SK_BM - 1) return;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int fx = lane & 15, fk = lane >> 4;
  const int rbase = m0 + w * 16 * SK_RT;
  if (rbase >= g.M) return;                                  // no barrier anywhere: a wave may leave alone
  const long bz = blockIdx.y;
  const double* __restrict__ A = g.A + bz * g.sA;
  const double* __restrict__ B = g.B + bz * g.sB;
  double* __restrict__ C = g.C + bz * g.sC;
  const int nk = (g.K + 3) >> 2;

  d4 acc[SK_RT][SK_CT];
  const double scale = (g.beta == 0.0) ? 0.0 : g.beta / g.alpha;
#pragma unroll
  for (int i = 0; i < SK_RT; i++)
#pragma unroll
    for (int j = 0; j < SK_CT; j++) {
      const int col = n0 + j * 16 + fx;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int row = rbase + i * 16 + fk + 4 * r;
        double c = 0.0;
        if (scale != 0.0 && row < g.M && col < g.N) c = scale * C[(long)row * g.ldc + col];
        acc[i][j][r] = c;
      }
    }
  double a[SK_RT][SK_KSTEPS], b[SK_CT][SK_KSTEPS];
#pragma unroll
  for (int i = 0; i < SK_RT; i++) {
    const int row = rbase + i * 16 + fx;
    const double* ap = A + (long)row * g.lda + fk;
#pragma unroll
    for (int kk = 0; kk < SK_KSTEPS; kk++) a[i][kk] = (row < g.M && kk * 4 + fk < g.K) ? ap[kk * 4] : 0.0;
  }
#pragma unroll
  for (int j = 0; j < SK_CT; j++) {
    const int col = n0 + j * 16 + fx;
#pragma unroll
    for (int kk = 0; kk < SK_KSTEPS; kk++) {
      const int k = kk * 4 + fk;
      double v = 0.0;
      if (col < g.N && k < g.K) v = TB ? B[(long)col * g.ldb + k] : B[(long)k * g.ldb + col];
      b[j][kk] = v;
    }
  }
#pragma unroll
  for (int kk = 0; kk < SK_KSTEPS; kk++)
    if (kk < nk) {
#pragma unroll
      for (int i = 0; i < SK_RT; i++)
#pragma unroll
        for (int j = 0; j < SK_CT; j++) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i][kk], b[j][kk], acc[i][j], 0, 0, 0);
    }
  const double alpha = g.alpha;
#pragma unroll
  for (int i = 0; i < SK_RT; i++)
#pragma unroll
    for (int j = 0; j < SK_CT; j++) {
      const int col = n0 + j * 16 + fx;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int row = rbase + i * 16 + fk + 4 * r;
        if (row < g.M && col < g.N) C[(long)row * g.ldc + col] = alpha * acc[i][j][r];
      }
    }
}

// the callers' (alpha, beta) for which starting the accumulators at (beta/alpha) C is exact
bool smallk_ok(const GemmArgs& g, bool transA) {
  return !transA && g.K >= 1 && g.K <= 4 * SK_KSTEPS && (g.alpha == 1.0 || g.alpha == -1.0) &&
         (g.beta == 0.0 || g.beta == 1.0 || g.beta == -1.0);
}
template <bool TB>
int launch_smallk(nd4hip_handle* h, GemmArgs g, int64_t batch) {
  g.tiles_m = (g.M + SK_BM - 1) / SK_BM; g.tiles_n = (g.N + SK_BN - 1) / SK_BN;
  ND4_CHECK_ARG((int64_t)g.tiles_m * g.tiles_n < (1ll << 31), "nd4_gemm: too many tiles");
  hipLaunchKernelGGL((dgemm_smallk_kernel<TB>), dim3((unsigned)(g.tiles_m * g.tiles_n), (unsigned)batch, 1), dim3(256, 1, 1), 0, h->stream, g);
  ND4_HIP(hipGetLastError());
  return 0;
}

// C = alpha * sum_z part[z] + beta * C   (fixed order: deterministic)
__global__ __launch_bounds__(256) void dgemm_splitk_reduce(const double* __restrict__ part, int nsplit, long per_split, int M, int N,
                                                            double alpha, double beta, double* __restrict__ Cm, long ldc, long sC) {
  const long b = blockIdx.y;
  const long e = blockIdx.x * 256l + threadIdx.x;
  if (e >= (long)M * N) return;
  double s = 0.0;
  for (int z = 0; z < nsplit; z++) s += part[z * per_split + b * (long)M * N + e];
  double* c = Cm + b * sC + (e / N) * ldc + e % N;
  double v = alpha * s;
  if (beta != 0.0) v += beta * *c;
  *c = v;
}

// ND4HIP_GEMM_TILED=1 sends every product through the tiled kernel (A/B measurements of the rank-k kernel)
bool nd4_gemm_force_tiled() {
  static const bool v = [] { const char* e = getenv("ND4HIP_GEMM_TILED"); return e && *e && *e != '0'; }();
  return v;
}

template <bool TA, bool TB>
int launch(nd4hip_handle* h, const GemmArgs& g, bool vec, int64_t batch) {
  const unsigned nsplit = g.kchunk > 0 ? (unsigned)((g.K + g.kchunk - 1) / g.kchunk) : 1u;
  dim3 grid((unsigned)(g.tiles_m * g.tiles_n), (unsigned)batch, nsplit), block(256, 1, 1);
  const bool full = vec && g.M % BM == 0 && g.N % BN == 0 && g.K % BK == 0 && g.K > 0;
  if (full)     hipLaunchKernelGGL((dgemm_kernel<TA, TB, true, true>), grid, block, 0, h->stream, g);
  else if (vec) hipLaunchKernelGGL((dgemm_kernel<TA, TB, true, false>), grid, block, 0, h->stream, g);
  else          hipLaunchKernelGGL((dgemm_kernel<TA, TB, false, false>), grid, block, 0, h->stream, g);
  ND4_HIP(hipGetLastError());
  return 0;
}

}  // namespace

int nd4_gemm(nd4hip_handle* h, bool transA, bool transB, int64_t M, int64_t N, int64_t K,
             double alpha, const double* A, int64_t lda, int64_t sA,
             const double* B, int64_t ldb, int64_t sB,
             double beta, double* C, int64_t ldc, int64_t sC, int64_t batch) {
  if (M <= 0 || N <= 0 || batch <= 0) return 0;
  ND4_CHECK_ARG(K >= 0 && M < (1 << 30) && N < (1 << 30) && K < (1 << 30), "nd4_gemm: extent out of range");
  ND4_CHECK_ARG(batch <= 65535, "nd4_gemm: batch %lld exceeds 65535 per launch", (long long)batch);
  GemmArgs g;
  g.A = A; g.B = B; g.C = C; g.M = (int)M; g.N = (int)N; g.K = (int)K;
  g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.sA = sA; g.sB = sB; g.sC = sC;
  g.alpha = alpha; g.beta = beta; g.lower = 0; g.kchunk = 0; g.Cpart = nullptr;
  g.tiles_m = (int)((M + BM - 1) / BM); g.tiles_n = (int)((N + BN - 1) / BN);
  ND4_CHECK_ARG((int64_t)g.tiles_m * g.tiles_n < (1ll << 31), "nd4_gemm: too many tiles");
  if (smallk_ok(g, transA) && !nd4_gemm_force_tiled()) return transB ? launch_smallk<true>(h, g, batch) : launch_smallk<false>(h, g, batch);
  // few output tiles and a long K (tall-skinny products, Gram matrices, Q^T y): split K over blockIdx.z so that the chip is
  // filled, then add the partials in a fixed order. (64 x 4096) x (4096 x 4096): 0.62 -> see DESIGN.md 4.1.
  Nd4WsScope scope(h);
  const int64_t tiles = (int64_t)g.tiles_m * g.tiles_n * batch;
  int nsplit = 1;
  if (tiles <= 160 && K >= 512) {                       // (<= 160: a launch of 128-160 tiles leaves a third of the 256 CUs idle)
    int64_t want = 384 / tiles;
    if (want > K / 256) want = K / 256;
    if (want > 32) want = 32;
    if (want >= 2) {
      int64_t kc = ((K + want - 1) / want + BK - 1) / BK * BK;
      nsplit = (int)((K + kc - 1) / kc);
      void* p = nullptr;
      ND4_TRY(nd4_ws_alloc(h, sizeof(double) * (size_t)nsplit * (size_t)batch * (size_t)M * (size_t)N, &p));
      g.kchunk = (int)kc; g.Cpart = static_cast<double*>(p);
    }
  }
  auto even = [](int64_t v) { return (v & 1) == 0; };
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  // 16-byte loads need: aligned bases, even row strides / batch strides, and an even extent along
  // the contiguous axis of each operand (so a 2-element chunk is never half out of range).
  const bool vec = al16(A) && al16(B) && even(lda) && even(ldb) && even(sA) && even(sB) &&
                   even(transA ? M : K) && even(transB ? K : N);
  if (transA) ND4_TRY((transB ? launch<true, true>(h, g, vec, batch) : launch<true, false>(h, g, vec, batch)));
  else        ND4_TRY((transB ? launch<false, true>(h, g, vec, batch) : launch<false, false>(h, g, vec, batch)));
  if (g.kchunk > 0) {
    hipLaunchKernelGGL(dgemm_splitk_reduce, dim3((unsigned)((M * N + 255) / 256), (unsigned)batch), dim3(256), 0, h->stream,
                       g.Cpart, nsplit, (long)batch * M * N, (int)M, (int)N, alpha, beta, C, (long)ldc, (long)sC);
    ND4_HIP(hipGetLastError());
  }
  return 0;
}

// C[lower tiles] = alpha * A A^T + beta * C for A [N, K] (row-major, lda): every 128x128 tile that touches the diagonal
// or lies below it is computed in full, tiles strictly above are skipped (their C entries are left untouched).
int nd4_syrk_lower(nd4hip_handle* h, int64_t N, int64_t K, double alpha, const double* A, int64_t lda, int64_t sA,
                   double beta, double* C, int64_t ldc, int64_t sC, int64_t batch) {
  return nd4_gemm_nt_lower(h, N, K, alpha, A, lda, sA, A, lda, sA, beta, C, ldc, sC, batch);
}
// the same with two different N x K factors: C[lower tiles] = alpha * A B^T + beta * C   (LDL^T: A = L21 D11, B = L21)
int nd4_gemm_nt_lower(nd4hip_handle* h, int64_t N, int64_t K, double alpha, const double* A, int64_t lda, int64_t sA,
                      const double* B, int64_t ldb, int64_t sB, double beta, double* C, int64_t ldc, int64_t sC, int64_t batch) {
  if (N <= 0 || batch <= 0) return 0;
  ND4_CHECK_ARG(K >= 0 && N < (1 << 30) && K < (1 << 30), "nd4_gemm_nt_lower: extent out of range");
  ND4_CHECK_ARG(batch <= 65535, "nd4_gemm_nt_lower: batch %lld exceeds 65535 per launch", (long long)batch);
  GemmArgs g;
  g.A = A; g.B = B; g.C = C; g.M = (int)N; g.N = (int)N; g.K = (int)K;
  g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.sA = sA; g.sB = sB; g.sC = sC;
  g.alpha = alpha; g.beta = beta; g.lower = 1; g.kchunk = 0; g.Cpart = nullptr;
  if (smallk_ok(g, false) && !nd4_gemm_force_tiled()) return launch_smallk<true>(h, g, batch);
  g.tiles_m = (int)((N + BM - 1) / BM); g.tiles_n = g.tiles_m;
  auto ok = [](const double* p, int64_t ld, int64_t st) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0 && (ld & 1) == 0 && (st & 1) == 0; };
  const bool vec = ok(A, lda, sA) && ok(B, ldb, sB) && (K & 1) == 0;
  return launch<false, true>(h, g, vec, batch);
}
