// Blocked right-looking LU with partial (row) pivoting on row-major fp64 matrices, batched.
//
// Replaces src/la/lu.js:24-81 (unblocked Doolittle, first-maximum row pivoting, full-row swaps,
// permutation VECTOR output). Per block column of width NB = 16:
//   lu_panel_* one workgroup per matrix factors rows [j0,N) x cols [j0,j0+nb): per column a
//              wave-shuffle + LDS arg-max (ties -> lowest row, exactly the reference's strict '>' scan,
//              lu.js:48-52), in-panel row swap, multipliers and rank-1 update. lu_panel_row<R>: one thread per
//              row, the R x 16 tile of each lane in registers (m <= 2048); lu_panel_row<R,W,1024>: the same on
//              1024 threads with 8- / 4-column panels (m <= 4096 / 8192: batches too large for the next form);
//              lu_panel_mw<R,PQ>: m > 2048, the rows over P <= 16 co-resident workgroups with ONE in-kernel exchange per column
//              (tagged words, see the comment at lu_panel_mw_body) — 16 columns wide at any height up to 32768 rows;
//              lu_panel_global: 16 lanes per row / panel in global memory (short tails).
//   look-ahead lu_panel_row_la / lu_panel_mw_la: the panel shares its launch with everything the previous panel still owes the other
//              columns (lu_colblock_update); between two panels only the next panel's 16 columns are updated (lu_narrow_fused, or
//              folded into the multi-workgroup panel's prologue). N > 2048: outer blocks of 512 columns, TRSM + one MFMA product each.
//   lu_laswp   applies the panel's nb row swaps to the columns left and right of it (lu.js:59-61 swaps full
//              rows) and, in the same pass, U12 = L11^-1 * A12 for the columns to the right (unit lower
//              16x16 in LDS), one thread per column -> coalesced.
//   nd4_gemm   A22 -= L21 * U12 on the fp64 MFMA GEMM (gemm.hip, rank-k kernel).
//   lu_build_perm  P from the recorded interchanges, once per factorisation.
#include "nd4hip_internal.h"
#include "dpp.h"
#include "xchg.h"
#include <type_traits>
#include <cfloat>

namespace {

constexpr int NB = 16;          // panel width == lanes per row
typedef double d4 __attribute__((ext_vector_type(4)));

struct PivCand { double mag; int idx; };

__device__ __forceinline__ PivCand better(PivCand a, PivCand b) {
  // larger magnitude wins; equal magnitude -> lower row index (first maximum, lu.js:50-52)
  return (b.mag > a.mag || (b.mag == a.mag && b.idx < a.idx)) ? b : a;
}
__device__ __forceinline__ double pivot_mag(double x, int r, int jc) {
  double m = fabs(x);
  // NaN never wins the reference's `abs(x) > abs(cur)` scan, unless it is the start row itself
  if (m != m) m = (r == jc) ? DBL_MAX * 2.0 /* +inf */ : -1.0;
  return m;
}

// block-wide arg-max over candidates; result broadcast through *s_piv
__device__ __forceinline__ int block_argmax(PivCand c, PivCand* s_red, int* s_piv, int t, int T) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    PivCand o;
    o.mag = __shfl_xor(c.mag, off);
    o.idx = __shfl_xor(c.idx, off);
    c = better(c, o);
  }
  const int wave = t >> 6, nw = (T + 63) >> 6;
  if ((t & 63) == 0) s_red[wave] = c;
  __syncthreads();
  if (t == 0) {
    PivCand b = s_red[0];
    for (int w = 1; w < nw; w++) b = better(b, s_red[w]);
    *s_piv = b.idx;
  }
  __syncthreads();
  return *s_piv;
}

// ---- general panel kernel: panel lives in global memory (any m) ------------------------------
__global__ void lu_panel_global(double* __restrict__ LU, int N, long strideM, int j0, int nb,
                                int32_t* __restrict__ P, int32_t* __restrict__ ipiv, int nopivot) {
  __shared__ PivCand s_red[16];
  __shared__ int s_piv;
  __shared__ double s_u[NB];
  double* A = LU + blockIdx.x * strideM;
  int32_t* ip = ipiv + (long)blockIdx.x * N;
  const int t = threadIdx.x, T = blockDim.x;
  const int c = t & (NB - 1), g = t >> 4, G = T >> 4;

  for (int k = 0; k < nb; k++) {
    const int jc = j0 + k;
    PivCand cand{-2.0, 0x7fffffff};
    for (int r = jc + t; r < N; r += T) {
      PivCand o{pivot_mag(A[(long)r * N + jc], r, jc), r};
      cand = better(cand, o);
    }
    int piv = block_argmax(cand, s_red, &s_piv, t, T);
    if (nopivot) piv = jc;
    if (t == 0) {
      ip[jc] = piv;
    }
    if (piv != jc && t < nb) {
      double* a = A + (long)jc * N + j0 + t; double* b = A + (long)piv * N + j0 + t;
      const double x = *a; *a = *b; *b = x;
    }
    __syncthreads();
    if (t < nb) s_u[t] = A[(long)jc * N + j0 + t];
    __syncthreads();
    const double pv = s_u[k], uc = (c < nb) ? s_u[c] : 0.0;
    for (int r = jc + 1 + g; r < N; r += G) {
      double* row = A + (long)r * N + j0;
      const double l = row[k] / pv;                  // lu.js:68
      if (c == k) row[k] = l;
      else if (c > k && c < nb) row[c] -= l * uc;    // lu.js:71-72
    }
    __syncthreads();
  }
}

// ---- thread-per-row panel kernel: each of T threads keeps R whole panel rows (W doubles each) in registers ----
// Row r = j0 + t + T*i. <R, 16, 512>: m <= 2048 (R = 1, 2, 4). Taller panels keep the layout on 1024 threads (128 VGPRs per lane) by
// narrowing the panel: <4, 8, 1024> up to 4096 rows (8-column panels), <8, 4, 1024> up to 8192 rows (4-column panels).
// Per column: one reciprocal + a residual correction and W-1-k FMAs per row, no cross-lane traffic in the update; the arg-max is a DPP
// wave reduction + T/64 LDS partials that every thread finishes itself (no second barrier); the pivot row and the displaced
// row travel through LDS. The column loop is expanded at compile time, so every register index is static. 2 barriers per column.
// stage != nullptr (look-ahead form, round 3): the panel's columns come from the contiguous block lu_narrow_fused has left behind
// ([N - j0 rows][16], already interchanged and updated by the previous panel), and the 16 x 16 block of U above the panel
// (stage_top) is copied into place on the way.
template <int R, int W, int T>
__device__ __forceinline__ void lu_panel_row_body(const int mat, double* __restrict__ LU, int N, long strideM, int j0, int nb,
                                                  int32_t* __restrict__ ipiv, int nopivot, const double* __restrict__ stage = nullptr,
                                                  const double* __restrict__ stage_top = nullptr) {
  constexpr int NWV = T / 64;
  static_assert(NWV == 8 || NWV == 16, "8 or 16 waves");
  __shared__ PivCand s_red[NWV];
  __shared__ double s_u[W], s_j[W];
  double* A = LU + mat * strideM;
  int32_t* ip = ipiv + (long)mat * N;
  const int t = threadIdx.x, wave = t >> 6;
  double a[R][W];
#pragma unroll
  for (int i = 0; i < R; i++) {
    const int r = j0 + t + T * i;
#pragma unroll
    for (int c = 0; c < W; c++) a[i][c] = 0.0;
    if (r < N) {
      const double* src = stage != nullptr ? stage + (long)(r - j0) * W : A + (long)r * N + j0;
      if (nb == W && ((N & 1) == 0 || stage != nullptr)) {   // 16-byte loads: each lane reads its own 128-B row segment
#pragma unroll
        for (int c = 0; c < W; c += 2) { const double2 v = *reinterpret_cast<const double2*>(src + c); a[i][c] = v.x; a[i][c + 1] = v.y; }
      } else {
#pragma unroll
        for (int c = 0; c < W; c++) if (c < nb) a[i][c] = src[c];
      }
    }
  }
  if (stage_top != nullptr && t < NB * NB) A[(long)(j0 - NB + t / NB) * N + j0 + t % NB] = stage_top[t];   // U12 of the previous panel for these columns
  // The column loop is expanded at compile time (a generic lambda called with integral constants), not by the loop unroller:
  // the DPP cross-lane moves are convergent operations, a loop that contains them is only unrolled late, after the pass that
  // splits the register tile into scalars has run, and the tile would then live in scratch memory.
  auto column = [&](auto kc) {
    constexpr int k = decltype(kc)::value;
    if (k < nb) {                                        // uniform
      const int jc = j0 + k;
      // ---- arg-max of |column k| over rows >= jc ----
      PivCand cand{-2.0, 0x7fffffff};
#pragma unroll
      for (int i = 0; i < R; i++) {
        const int r = j0 + t + T * i;
        PivCand o{pivot_mag(a[i][k], r, jc), r};
        if ((i == 0 && t < k) || r >= N) o.mag = -2.0;   // r < jc is only possible in row slot 0 of the first k threads
        cand = better(cand, o);
      }
      // wave arg-max on DPP (VALU speed; a ds_bpermute butterfly is six dependent LDS-crossbar round trips): the largest
      // magnitude first, then the lowest row among the lanes that hold it (first maximum, lu.js:50-52)
      const double wm = nd4dpp::wave_max(cand.mag);
      const int wi = nd4dpp::wave_min(cand.mag == wm ? cand.idx : 0x7fffffff);
      if ((t & 63) == 0) { s_red[wave].mag = wm; s_red[wave].idx = wi; }
      __syncthreads();
      // the wave candidates: lane l reads candidate l & (NWV - 1), log2(NWV) DPP steps leave the block result in every lane
      const PivCand c8 = s_red[t & (NWV - 1)];
      double bm = fmax(c8.mag, nd4dpp::xor1(c8.mag)); bm = fmax(bm, nd4dpp::xor2(bm)); bm = fmax(bm, nd4dpp::xor4(bm));
      if constexpr (NWV == 16) bm = fmax(bm, nd4dpp::xor8(bm));
      int bi = c8.mag == bm ? c8.idx : 0x7fffffff;
      bi = min(bi, nd4dpp::xor1(bi)); bi = min(bi, nd4dpp::xor2(bi)); bi = min(bi, nd4dpp::xor4(bi));
      if constexpr (NWV == 16) bi = min(bi, nd4dpp::xor8(bi));
      const int piv = nopivot ? jc : bi;               // (kept in a VGPR: a scalar row index would turn the `i == pi` selects below
                                                       //  into a dynamically indexed register tile, i.e. scratch memory)
      if (t == 0) {
        ip[jc] = piv;
        }
      // ---- publish the pivot row and the displaced row jc (owner: thread k, slot 0) ----
      // (each row slot tests its own row number: a slot index derived from piv would let the optimiser turn the tile into a
      //  dynamically indexed array, i.e. scratch memory)
#pragma unroll
      for (int i = 0; i < R; i++)
        if (j0 + t + T * i == piv) {
#pragma unroll
          for (int c = 0; c < W; c++) s_u[c] = a[i][c];
        }
      if (t == k) {
#pragma unroll
        for (int c = 0; c < W; c++) s_j[c] = a[0][c];
      }
      __syncthreads();
      if (piv != jc) {
#pragma unroll
        for (int i = 0; i < R; i++)
          if (j0 + t + T * i == piv) {
#pragma unroll
            for (int c = 0; c < W; c++) a[i][c] = s_j[c];
          }
        if (t == k) {
#pragma unroll
          for (int c = 0; c < W; c++) a[0][c] = s_u[c];
        }
      }
      // ---- eliminate below the pivot ----
      // The pivot row once into registers (wave-uniform LDS broadcast reads): every row slot uses the same values. Only row slot 0
      // of the first k+1 threads can hold a row that is already finished (r <= jc <=> i == 0 && t <= k): slots 1.. need no test
      // at all (rows >= N hold zeros: harmless, never stored), so the update is straight-line code for them — with a test per slot
      // the compiler re-read the pivot row from LDS inside each of the four divergent branches (32 ds_read_b128 per column).
      double u[W];
#pragma unroll
      for (int c = 0; c < W; c++) u[c] = (c >= k) ? s_u[c] : 0.0;
      const double pk = u[k];
      // the reciprocal route is used while 1/pivot and the products with it stay far from over/underflow (2^-1000 <= |pivot| <= 2^1000);
      // zero, denormal, huge, Inf and NaN pivots take the IEEE division of lu.js:68 itself (wave-uniform branch: pk is)
      const double apk = fabs(pk);
      const bool fast = apk >= 0x1p-1000 && apk <= 0x1p1000;
      const double rk = fast ? nd4dpp::fast_rcp(pk) : 0.0;
#pragma unroll
      for (int i = 0; i < R; i++) {
        if (i > 0 || t > k) {
          // a / pivot (lu.js:68) as a * (1/pivot) with one residual correction: the quotient of a division that comes out
          // exact (integer-valued and structured inputs, where later pivot TIES depend on it) is reproduced exactly, any other
          // to an ulp; 4 instructions per row instead of the ~15 of an IEEE division. L therefore matches the reference to 1 ulp,
          // not bit for bit: P is bit-identical on every golden and oracle comparison, but on a NEAR-tie of two candidates whose
          // values differ in the last bit that identity rests on those comparisons, not on the arithmetic.
          const double q0 = a[i][k] * rk;
          const double l = fast ? fma(fma(-q0, pk, a[i][k]), rk, q0) : a[i][k] / pk;
          a[i][k] = l;
#pragma unroll
          for (int c = k + 1; c < W; c++) a[i][c] -= l * u[c];   // lu.js:71-72
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);                   // keep the unrolled columns from interleaving (VGPR pressure)
  };
#define ND4_COL(K) column(std::integral_constant<int, K>{});
  ND4_COL(0) ND4_COL(1) ND4_COL(2) ND4_COL(3)
  if constexpr (W > 4) { ND4_COL(4) ND4_COL(5) ND4_COL(6) ND4_COL(7) }
  if constexpr (W > 8) { ND4_COL(8) ND4_COL(9) ND4_COL(10) ND4_COL(11) ND4_COL(12) ND4_COL(13) ND4_COL(14) ND4_COL(15) }
#undef ND4_COL
#pragma unroll
  for (int i = 0; i < R; i++) {
    const int r = j0 + t + T * i;
    if (r < N) {
      double* dst = A + (long)r * N + j0;
      if (nb == W && (N & 1) == 0) {
#pragma unroll
        for (int c = 0; c < W; c += 2) *reinterpret_cast<double2*>(dst + c) = double2{a[i][c], a[i][c + 1]};
      } else {
#pragma unroll
        for (int c = 0; c < W; c++) if (c < nb) dst[c] = a[i][c];
      }
    }
  }
}


template <int R, int W, int T>
__global__ __launch_bounds__(T) void lu_panel_row(double* __restrict__ LU, int N, long strideM, int j0, int nb,
                                                       int32_t* __restrict__ P, int32_t* __restrict__ ipiv, int nopivot) {
  lu_panel_row_body<R, W, T>(blockIdx.x, LU, N, strideM, j0, nb, ipiv, nopivot);
}

// ---- look-ahead: everything panel pj0 does to ONE block of <= 16 columns outside it, by ONE workgroup ----
// The 16 row interchanges, U12 = L11^-1 A12 and A22 -= L21 U12 are all local to a column, so one workgroup can take a block of columns
// through all three without any other workgroup: the update of the columns BEHIND the next panel then runs in the same launch as the
// next panel's factorisation (lu_panel_row_la: the panel keeps one workgroup busy for 20-37 us while the chip idled), and only the next
// panel's own 16 columns are updated between two panels (lu_narrow_top + lu_narrow_gemm).
// The interchanges are not replayed one after the other (a chain of 16 dependent load/store pairs): the rows they touch are the 16 top
// rows and the <= 16 pivot rows, and what ends up in each of them is found by walking the transpositions backwards (16 compare steps
// on registers), so it is one gather, a barrier, one scatter. right = false (columns left of the panel): the interchanges only.
// Pm != nullptr: this workgroup also carries the permutation vector through the same gather (as one more column, lu.js:59-61).
template <int NT, bool GEMM>
__device__ __forceinline__ void lu_colblock_update(double* __restrict__ A, int N, int pj0, const int32_t* __restrict__ ip, int do_swap,
                                                   int c0, int nc, bool right, int32_t* __restrict__ Pm) {
  __shared__ int s_piv[NB], s_src[2 * NB], s_dst[2 * NB];
  __shared__ double s_top[NB][NB + 1], s_l[NB][NB + 1];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, fx = lane & 15, fk = lane >> 4;
  if (t < NB) s_piv[t] = do_swap ? ip[pj0 + t] : pj0 + t;
  if (right && t >= 256 && t < 512) {                             // L11 (strictly lower, unit diagonal implied)
    const int i = (t - 256) / NB, j = (t - 256) % NB;
    s_l[i][j] = (j < i) ? A[(long)(pj0 + i) * N + pj0 + j] : 0.0;
  }
  __syncthreads();
  if (t < 2 * NB) {
    const int k = t & (NB - 1);
    const int dst = t < NB ? pj0 + k : s_piv[k];
    int pos = dst;
#pragma unroll
    for (int q = NB - 1; q >= 0; q--) {                           // where the content of row dst comes from: the transpositions backwards
      const int ps = s_piv[q];
      pos = (pos == pj0 + q) ? ps : ((pos == ps) ? pj0 + q : pos);
    }
    s_src[t] = pos;
    s_dst[t] = (t < NB || dst >= pj0 + NB) ? dst : -1;             // a pivot row inside the top block is already there as a top row
  }
  __syncthreads();
  double v[(2 * NB * NB + NT - 1) / NT];
  int32_t pval = 0;
#pragma unroll
  for (int e = 0; e < (2 * NB * NB + NT - 1) / NT; e++) {
    const int idx = t + NT * e, k = idx / NB, c = idx % NB;
    v[e] = 0.0;
    if (idx < 2 * NB * NB && c < nc && s_dst[k] >= 0) v[e] = A[(long)s_src[k] * N + c0 + c];
  }
  if (Pm != nullptr && t < 2 * NB && s_dst[t] >= 0) pval = Pm[s_src[t]];
  __syncthreads();
#pragma unroll
  for (int e = 0; e < (2 * NB * NB + NT - 1) / NT; e++) {
    const int idx = t + NT * e, k = idx / NB, c = idx % NB;
    if (idx < 2 * NB * NB && c < nc && s_dst[k] >= 0) {
      if (right && k < NB) s_top[k][c] = v[e];
      else A[(long)s_dst[k] * N + c0 + c] = v[e];
    }
    if (idx < 2 * NB * NB && c >= nc && k < NB) s_top[k][c] = 0.0;
  }
  if (Pm != nullptr && t < 2 * NB && s_dst[t] >= 0) Pm[s_dst[t]] = pval;
  if (!right) return;
  __syncthreads();
  if (t < NB) {                                                   // U12 = L11^-1 A12, one thread per column, same order as lu.js:71-72
    double x[NB];
#pragma unroll
    for (int i = 0; i < NB; i++) x[i] = s_top[i][t];
#pragma unroll
    for (int i = 1; i < NB; i++) {
      double acc = x[i];
#pragma unroll
      for (int j = 0; j < i; j++) acc -= s_l[i][j] * x[j];
      x[i] = acc;
    }
#pragma unroll
    for (int i = 0; i < NB; i++) {
      s_top[i][t] = x[i];
      if (t < nc) A[(long)(pj0 + i) * N + c0 + t] = x[i];
    }
  }
  if constexpr (!GEMM) return;
  __syncthreads();
  // A22 -= L21 U12 on fp64 MFMA: the waves split the rows below the top block
  constexpr int NW = NT / 64;
  const int m2 = N - pj0 - NB;
  const int rpw = ((m2 + NW * 16 - 1) / (NW * 16)) * 16;
  const int r0 = wave * rpw, r1 = (r0 + rpw < m2) ? r0 + rpw : m2;
  const bool cok = fx < nc;
  double bw[4];
#pragma unroll
  for (int kk = 0; kk < 4; kk++) bw[kk] = -s_top[kk * 4 + fk][fx];
  const double* Lp = A + (long)(pj0 + NB) * N + pj0;
  double* Cp = A + (long)(pj0 + NB) * N + c0;
  for (int rt = r0; rt < r1; rt += 64) {                          // four 16-row tiles per batch
    double av[4][4]; d4 c[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int ra = rt + q * 16 + fx;
#pragma unroll
      for (int kk = 0; kk < 4; kk++) av[q][kk] = (ra < r1) ? Lp[(long)ra * N + kk * 4 + fk] : 0.0;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int rc = rt + q * 16 + fk + 4 * r;
        c[q][r] = (rc < r1 && cok) ? Cp[(long)rc * N + fx] : 0.0;
      }
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
#pragma unroll
      for (int kk = 0; kk < 4; kk++) c[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q][kk], bw[kk], c[q], 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int rc = rt + q * 16 + fk + 4 * r;
        if (rc < r1 && cok) Cp[(long)rc * N + fx] = c[q][r];
      }
    }
  }
}

// which block of columns an update workgroup owns: block 0 = the permutation vector alone, then the blocks left of the panel at pj0
// (interchanges only), then the blocks from column wide0 on
// (two-level blocking: the blocks from column full_end on — beyond the outer block — get the interchanges only, like the left ones)
template <int NT, bool GEMM>
__device__ __forceinline__ void lu_update_block(int b, int mat, double* __restrict__ LU, int N, long strideM, int pj0, int wide0,
                                                const int32_t* __restrict__ ipiv, int do_swap, int32_t* __restrict__ Pm, int full_end) {
  double* A = LU + mat * strideM;
  const int32_t* ip = ipiv + (long)mat * N;
  const int nleft = pj0 / NB;
  if (b == 0) { if (do_swap) lu_colblock_update<NT, false>(A, N, pj0, ip, do_swap, 0, 0, false, Pm + (long)mat * N); return; }
  b -= 1;
  if (b < nleft) { if (do_swap) lu_colblock_update<NT, false>(A, N, pj0, ip, do_swap, b * NB, NB, false, nullptr); return; }
  const int c0 = wide0 + (b - nleft) * NB;
  if (c0 >= full_end) { if (do_swap) lu_colblock_update<NT, false>(A, N, pj0, ip, do_swap, c0, N - c0 < NB ? N - c0 : NB, false, nullptr); return; }
  lu_colblock_update<NT, GEMM>(A, N, pj0, ip, do_swap, c0, N - c0 < NB ? N - c0 : NB, true, nullptr);
}

// panel at j0 (workgroup 0 of a matrix) together with everything the previous panel (at pj0) still owes the other columns
template <int R>
__global__ __launch_bounds__(512) void lu_panel_row_la(double* __restrict__ LU, int N, long strideM, int j0, int32_t* __restrict__ P,
                                                        int32_t* __restrict__ ipiv, int nopivot, int pj0, int wide0,
                                                        const double* __restrict__ stage, long strideStage, int full_end) {
  if (blockIdx.x == 0) {
    const double* sg = stage != nullptr ? stage + blockIdx.y * strideStage : nullptr;
    lu_panel_row_body<R, NB, 512>(blockIdx.y, LU, N, strideM, j0, NB, ipiv, nopivot, sg != nullptr ? sg + NB * NB : nullptr, sg);
    return;
  }
  lu_update_block<512, true>((int)blockIdx.x - 1, blockIdx.y, LU, N, strideM, pj0, wide0, ipiv, nopivot ? 0 : 1, P, full_end);
}
// the same update work on its own (the last look-ahead panel's debt)
__global__ __launch_bounds__(512) void lu_update_blocks(double* __restrict__ LU, int N, long strideM, int32_t* __restrict__ P,
                                                         const int32_t* __restrict__ ipiv, int nopivot, int pj0, int wide0, int full_end) {
  lu_update_block<512, true>((int)blockIdx.x, blockIdx.y, LU, N, strideM, pj0, wide0, ipiv, nopivot ? 0 : 1, P, full_end);
}
// ---- panels taller than one workgroup's registers: the rows over P co-resident workgroups, ONE exchange per column (round 3) ----
// Partial pivoting needs the arg-max over ALL rows below the diagonal for every column (lu.js:48-52), so a panel whose rows are
// spread over workgroups needs one all-to-all per column. Done with fences (release: write back the L2, acquire: invalidate it) such
// an exchange costs 2-18 us on this chip; done with agent-scope relaxed atomics only (sc1 stores are written through, sc1 loads
// miss the non-coherent caches: no write-back, no invalidate) it costs 1.3-1.7 us for P <= 16, also with the rest of the chip
// streaming through the same L2s (tools/xwg_lat.hip) — less than one kernel boundary. Each memory hop of such an exchange is
// 0.4-1 us, so the protocol has two (store -> load), not four (store, wait, flag -> poll -> payload): every 8-byte word that
// crosses carries 32 bits of payload and a 32-bit tag (the global column number + 1), i.e. a double travels as two single-copy-atomic
// words and is valid as soon as both tags match: no flag, no s_waitcnt, no ordering between words needed. Per column k:
//   every wave: arg-max of its rows (DPP), its candidate and the candidate's 16-value row segment into LDS; barrier;
//   wave 0:     the workgroup's candidate -> its slot (row segment, magnitude, row number; workgroup 0 also the current row jc,
//               which the pivot row displaces); then it polls all P slots (ONE polling wave per workgroup: eight waves polling
//               the same lines doubled every hop), finds the winner (largest magnitude, ties -> lowest row: the first maximum of
//               lu.js:50-52, exactly as inside one workgroup) and leaves pivot row, displaced row and pivot row number in LDS; barrier;
//   every wave: eliminates its rows; the thread that owns the pivot row takes the displaced row jc, thread k of workgroup 0 (the
//               owner of row jc) the pivot row.
// The slots alternate by column parity: a workgroup publishes column k + 2 only after it has read every other workgroup's column
// k + 1, which those publish after reading column k. The slots are cleared once per call (a tag never repeats inside one
// factorisation). Every spin is bounded: a stuck exchange (cannot happen while the P workgroups are co-resident: P * batch <= 64
// here) raises *stuck, every workgroup leaves, and lu_mw_poison marks the permutation vector invalid.
constexpr int MW_SLOT = 128;                      // 8-byte words per slot; value v = words 2v, 2v+1: [0,16) candidate row, 16 magnitude, 17 row number, [18,34) row jc
constexpr int MW_MAXP = 16;
constexpr long MW_STRIDE = 2l * MW_MAXP * MW_SLOT;   // per matrix: [parity][workgroup] slots (in doubles == words)
constexpr int MW_SPIN_LIMIT = 1 << 21;
typedef unsigned long long mw_u64;

__device__ __forceinline__ void mw_st(mw_u64* slot, int v, double x, unsigned tag) {
  const mw_u64 bits = (mw_u64)__double_as_longlong(x), tg = (mw_u64)tag << 32;
  __hip_atomic_store(slot + 2 * v, (bits & 0xffffffffull) | tg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(slot + 2 * v + 1, (bits >> 32) | tg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double mw_ld(const mw_u64* slot, int v, unsigned tag, bool& ok) {
  const mw_u64 w0 = __hip_atomic_load(slot + 2 * v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const mw_u64 w1 = __hip_atomic_load(slot + 2 * v + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  ok = ok && (unsigned)(w0 >> 32) == tag && (unsigned)(w1 >> 32) == tag;
  return __longlong_as_double((long long)((w1 << 32) | (w0 & 0xffffffffull)));
}

// stage != nullptr (look-ahead form): the panel's columns come from the contiguous block lu_narrow_fused has left behind, and the
// 16 x 16 block of U above the panel (stage_top) is copied into place on the way (as in lu_panel_row_body).
template <int R, int PQ, bool STAMPS = false>
__device__ __forceinline__ void lu_panel_mw_body(const int w, const int mat, double* __restrict__ LU, int N, long strideM, int j0,
                                                 int32_t* __restrict__ ipiv, int nopivot, double* __restrict__ xbuf, int* __restrict__ stuck,
                                                 int P, unsigned long long* __restrict__ stamps, const double* __restrict__ stage,
                                                 const double* __restrict__ stage_top, const int fold_pj0 = -1) {
  // fold_pj0 >= 0 (one row per thread only): the previous panel's work on THESE 16 columns happens here instead of in a launch of
  // its own (lu_narrow_fused): every workgroup works out the interchanges, gathers the 16 x 16 block above the panel, solves
  // U12 = L11^-1 A12 (redundantly) and updates its own rows, read from where the interchanges take them from, with 256 FMAs per row
  // in the order of lu.js:71-72. Nothing is written before the first exchange: a row is only ever read from itself or from one of
  // the 16 top rows, the top rows' sources are rows of some workgroup's tile, and no workgroup stores its tile (or U12: workgroup 0,
  // at the end) before every workgroup has published its first column, i.e. has left this prologue.
  constexpr int T = 512, W = NB, RT = R * T, NWV = 8;
  const bool fold = R == 1 && fold_pj0 >= 0;
  const bool prio = stage_top != nullptr || stage != nullptr || fold;
  __shared__ PivCand s_red[NWV];
  __shared__ double s_rows[NWV][W], s_j[W], s_u[W], s_dj[W];
  __shared__ int s_piv;
  double* A = LU + mat * strideM;
  int32_t* ip = ipiv + (long)mat * N;
  mw_u64* slots = reinterpret_cast<mw_u64*>(xbuf + mat * MW_STRIDE);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int rbase = j0 + w * RT;                  // first row of this workgroup; row of (thread t, slot i) = rbase + t + T * i
  const bool vec = (N & 1) == 0;
  double a[R][W];
  __shared__ int s_piv2[NB], s_src2[2 * NB], s_dst2[2 * NB];
  __shared__ double s_top2[NB][NB + 1], s_l2[NB][NB + 1];
  if (fold) {
    const int pj = fold_pj0;
    if (t < NB) s_piv2[t] = nopivot ? pj + t : ip[pj + t];
    if (t < NB * NB) { const int i = t / NB, j = t % NB; s_l2[i][j] = (j < i) ? A[(long)(pj + i) * N + pj + j] : 0.0; }
    __syncthreads();
    if (t < 2 * NB) {                                             // where the content of the 16 top rows / the 16 pivot rows comes from
      const int k = t & (NB - 1);
      const int dst = t < NB ? pj + k : s_piv2[k];
      int pos = dst;
#pragma unroll
      for (int q = NB - 1; q >= 0; q--) {
        const int ps = s_piv2[q];
        pos = (pos == pj + q) ? ps : ((pos == ps) ? pj + q : pos);
      }
      s_src2[t] = pos;
      s_dst2[t] = (t < NB || dst >= pj + NB) ? dst : -1;
    }
    __syncthreads();
    if (t < NB * NB) { const int k = t / NB, c = t % NB; s_top2[k][c] = A[(long)s_src2[k] * N + j0 + c]; }
    __syncthreads();
    if (t < NB) {                                                 // U12 = L11^-1 A12, one thread per column
      double x[NB];
#pragma unroll
      for (int i = 0; i < NB; i++) x[i] = s_top2[i][t];
#pragma unroll
      for (int i = 1; i < NB; i++) {
        double acc = x[i];
#pragma unroll
        for (int j = 0; j < i; j++) acc -= s_l2[i][j] * x[j];
        x[i] = acc;
      }
#pragma unroll
      for (int i = 0; i < NB; i++) s_top2[i][t] = x[i];
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < R; i++) {
    const int r = rbase + t + T * i;
#pragma unroll
    for (int c = 0; c < W; c++) a[i][c] = 0.0;
    if (r < N && fold) {
      int srow = r;
#pragma unroll
      for (int k = 0; k < NB; k++) if (s_dst2[NB + k] == r) srow = s_src2[NB + k];
      const double* lrow = A + (long)r * N + fold_pj0;
      const double* vrow = A + (long)srow * N + j0;
      double l[NB];
      if (vec) {
#pragma unroll
        for (int c = 0; c < W; c += 2) {
          const double2 lv = *reinterpret_cast<const double2*>(lrow + c), vv = *reinterpret_cast<const double2*>(vrow + c);
          l[c] = lv.x; l[c + 1] = lv.y; a[i][c] = vv.x; a[i][c + 1] = vv.y;
        }
      } else {
#pragma unroll
        for (int c = 0; c < W; c++) { l[c] = lrow[c]; a[i][c] = vrow[c]; }
      }
#pragma unroll
      for (int j = 0; j < NB; j++) {
#pragma unroll
        for (int c = 0; c < W; c++) a[i][c] -= l[j] * s_top2[j][c];       // lu.js:71-72, column j of L after column j - 1
      }
    } else if (r < N) {
      const double* src = stage != nullptr ? stage + (long)(r - j0) * W : A + (long)r * N + j0;
      if (vec || stage != nullptr) {
#pragma unroll
        for (int c = 0; c < W; c += 2) { const double2 v = *reinterpret_cast<const double2*>(src + c); a[i][c] = v.x; a[i][c + 1] = v.y; }
      } else {
#pragma unroll
        for (int c = 0; c < W; c++) a[i][c] = src[c];
      }
    }
  }
  if (stage_top != nullptr && w == 0 && t < NB * NB) A[(long)(j0 - NB + t / NB) * N + j0 + t % NB] = stage_top[t];   // U12 of the previous panel for these columns
  if (prio) __builtin_amdgcn_s_setprio(3);          // the chain before the update workgroups that share the chip (and maybe the CU)
  const int drop_tag = w == P - 1 ? stuck[1] : 0;    // tests only (ND4HIP_TEST_DROP_PUBLISH): the column whose publication this workgroup skips; 0 = none
  auto column = [&](auto kc) {
    constexpr int k = decltype(kc)::value;
    const int jc = j0 + k, tag = jc + 1;
    // debug stamps (ND4HIP_LU_STAMPS): [workgroup][wave][column][phase] shader clocks of lane 0
    auto stamp = [&](int ph) { if constexpr (STAMPS) { if (lane == 0 && mat == 0) stamps[(((long)w * NWV + wave) * W + k) * 8 + ph] = __builtin_amdgcn_s_memtime(); } };
    stamp(0);
    mw_u64* myslot = slots + ((long)(k & 1) * MW_MAXP + w) * MW_SLOT;
    const mw_u64* sl = slots + (long)(k & 1) * MW_MAXP * MW_SLOT;
    // ---- this wave's candidate ----
    PivCand cand{-2.0, 0x7fffffff};
#pragma unroll
    for (int i = 0; i < R; i++) {
      const int r = rbase + t + T * i;
      PivCand o{pivot_mag(a[i][k], r, jc), r};
      if (nopivot) o.mag = (r == jc) ? DBL_MAX * 2.0 : -2.0;
      if ((w == 0 && i == 0 && t < k) || r >= N) o.mag = -2.0;
      cand = better(cand, o);
    }
    const double wm = nd4dpp::wave_max(cand.mag);
    // the lowest row among the lanes that hold the maximum: with one row per thread (or a single such lane) that is the lowest such lane
    const unsigned long long eqm = __builtin_amdgcn_ballot_w64(cand.mag == wm);
    int wi;
    if (R == 1 || __builtin_popcountll(eqm) == 1) wi = __builtin_amdgcn_readlane(cand.idx, (int)__builtin_ctzll(eqm));
    else wi = nd4dpp::wave_min(cand.mag == wm ? cand.idx : 0x7fffffff);
    if (lane == 0) { s_red[wave].mag = wm; s_red[wave].idx = wi; }
#pragma unroll
    for (int i = 0; i < R; i++)
      if (rbase + t + T * i == wi) {
#pragma unroll
        for (int c = 0; c < W; c++) s_rows[wave][c] = a[i][c];
      }
    if (w == 0 && t == k) {
#pragma unroll
      for (int c = 0; c < W; c++) s_j[c] = a[0][c];
    }
    stamp(1);
    __syncthreads();
    stamp(2);
    if (wave == 0) {
      // ---- publish the workgroup's candidate ----
      const PivCand c8 = s_red[lane & (NWV - 1)];
      double bm = fmax(c8.mag, nd4dpp::xor1(c8.mag)); bm = fmax(bm, nd4dpp::xor2(bm)); bm = fmax(bm, nd4dpp::xor4(bm));
      int bi = c8.mag == bm ? c8.idx : 0x7fffffff;
      bi = min(bi, nd4dpp::xor1(bi)); bi = min(bi, nd4dpp::xor2(bi)); bi = min(bi, nd4dpp::xor4(bi));
      const int ww = (((bi - rbase) & (T - 1)) >> 6) & (NWV - 1);
      if (drop_tag == tag) {}                                          // (tests: one dropped publication)
      else if (lane < W) mw_st(myslot, lane, s_rows[ww][lane], (unsigned)tag);
      else if (lane == 16) mw_st(myslot, 16, bm, (unsigned)tag);
      else if (lane == 17) mw_st(myslot, 17, __longlong_as_double((long long)bi), (unsigned)tag);
      else if (w == 0 && lane >= 32 && lane < 32 + W) mw_st(myslot, 18 + (lane - 32), s_j[lane - 32], (unsigned)tag);
      stamp(3);
      // ---- all candidates at once, until every word carries this column's tag: lanes 0..15 magnitude + row number of workgroup
      //      `lane`, lanes 32..47 the displaced row jc, and the P row segments 16 lanes each ----
      // Three polls in flight, a few hundred cycles apart (loads return in order, so the first is examined while the others are
      // still under way): the exchange is seen ~ one stagger after it lands instead of up to one memory round trip (~1000 cycles) later.
      double m1 = -3.0, m2 = 0.0, rowsv[PQ];
      // (branch-free: every lane loads from a valid address — lanes without a job of their own repeat a neighbour's — so that all
      //  loads of a poll are in flight together; with the loads under divergent branches the compiler waits for each in turn)
      const bool l_dj = lane >= 32 && lane < 32 + W;
      const int gl = (lane & 15) < P ? (lane & 15) : P - 1;
      const mw_u64* p1 = l_dj ? sl + 2 * (18 + (lane - 32)) : sl + (long)gl * MW_SLOT + 2 * 16;
      const mw_u64* p2 = sl + (long)gl * MW_SLOT + 2 * 17;
      const mw_u64* pr[PQ];
#pragma unroll
      for (int q = 0; q < PQ; q++) {
        const int g = q * 4 + (lane >> 4);
        pr[q] = sl + (long)(g < P ? g : P - 1) * MW_SLOT + 2 * (lane & 15);
      }
      constexpr int NWD = 2 * (2 + PQ);
      auto issue = [&](mw_u64 (&wd)[NWD]) {
        wd[0] = __hip_atomic_load(p1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); wd[1] = __hip_atomic_load(p1 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        wd[2] = __hip_atomic_load(p2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); wd[3] = __hip_atomic_load(p2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int q = 0; q < PQ; q++) {
          wd[4 + 2 * q] = __hip_atomic_load(pr[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          wd[5 + 2 * q] = __hip_atomic_load(pr[q] + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      };
      // (the empty asm pins the examination of a poll behind the branch on the previous one: otherwise all three are examined —
      //  i.e. waited for — before the first branch)
      auto good = [&](mw_u64 (&wd)[NWD]) -> bool {
        bool ok = true;
#pragma unroll
        for (int i = 0; i < NWD; i++) { asm volatile("" : "+v"(wd[i])); ok = ok && (unsigned)(wd[i] >> 32) == (unsigned)tag; }
        return __builtin_amdgcn_ballot_w64(!ok) == 0;
      };
      auto val = [&](const mw_u64 (&wd)[NWD], int v) -> double {
        return __longlong_as_double((long long)((wd[2 * v + 1] << 32) | (wd[2 * v] & 0xffffffffull)));
      };
      auto take = [&](const mw_u64 (&wd)[NWD]) {
        m1 = val(wd, 0); m2 = val(wd, 1);
#pragma unroll
        for (int q = 0; q < PQ; q++) rowsv[q] = val(wd, 2 + q);
      };
      {
        mw_u64 wa[NWD], wb[NWD], wc[NWD];
        issue(wa); __builtin_amdgcn_s_sleep(3); issue(wb); __builtin_amdgcn_s_sleep(3); issue(wc);
        int spins = 0;
        for (;;) {
          if (good(wa)) { take(wa); break; }
          issue(wa);
          if (good(wb)) { take(wb); break; }
          issue(wb);
          if (good(wc)) { take(wc); break; }
          issue(wc);
          if (++spins > MW_SPIN_LIMIT) { if (lane == 0) __hip_atomic_store(stuck, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); take(wa); break; }
          if ((spins & 255) == 0 && __hip_atomic_load(stuck, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { take(wa); break; }
        }
      }
      stamp(4);
      double mm = lane < W ? m1 : -3.0;
      mm = fmax(mm, nd4dpp::xor1(mm)); mm = fmax(mm, nd4dpp::xor2(mm)); mm = fmax(mm, nd4dpp::xor4(mm)); mm = fmax(mm, nd4dpp::xor8(mm));
      int ii = (lane < P && m1 == mm) ? (int)__double_as_longlong(m2) : 0x7fffffff;
      ii = min(ii, nd4dpp::xor1(ii)); ii = min(ii, nd4dpp::xor2(ii)); ii = min(ii, nd4dpp::xor4(ii)); ii = min(ii, nd4dpp::xor8(ii));
      int pv = __builtin_amdgcn_readfirstlane(ii);
      pv = pv < jc ? jc : (pv > N - 1 ? N - 1 : pv);           // (a stuck exchange must not send later kernels out of bounds)
      int gw = (pv - j0) / RT; gw = gw > P - 1 ? P - 1 : gw;
      double sel = rowsv[0];
#pragma unroll
      for (int q = 1; q < PQ; q++) sel = (gw >> 2) == q ? rowsv[q] : sel;
      if ((lane >> 4) == (gw & 3)) s_u[lane & 15] = sel;
      if (lane >= 32 && lane < 32 + W) s_dj[lane - 32] = m1;
      if (lane == 0) { s_piv = pv; if (w == 0) ip[jc] = pv; }
    }
    __syncthreads();
    const int piv = s_piv;
    double u[W];
#pragma unroll
    for (int c = 0; c < W; c++) u[c] = s_u[c];
    stamp(5);
    if (piv != jc) {
#pragma unroll
      for (int i = 0; i < R; i++)
        if (rbase + t + T * i == piv) {
#pragma unroll
          for (int c = 0; c < W; c++) a[i][c] = s_dj[c];
        }
      if (w == 0 && t == k) {
#pragma unroll
        for (int c = 0; c < W; c++) a[0][c] = u[c];
      }
    }
    // ---- eliminate below the pivot (same arithmetic as lu_panel_row_body) ----
    const double pk = u[k];
    const double apk = fabs(pk);
    const bool fast = apk >= 0x1p-1000 && apk <= 0x1p1000;
    const double rk = fast ? nd4dpp::fast_rcp(pk) : 0.0;
#pragma unroll
    for (int i = 0; i < R; i++) {
      if (w > 0 || i > 0 || t > k) {
        const double q0 = a[i][k] * rk;
        const double l = fast ? fma(fma(-q0, pk, a[i][k]), rk, q0) : a[i][k] / pk;
        a[i][k] = l;
#pragma unroll
        for (int c = k + 1; c < W; c++) a[i][c] -= l * u[c];
      }
    }
    stamp(6);
    __builtin_amdgcn_sched_barrier(0);
  };
#define ND4_COL(K) column(std::integral_constant<int, K>{});
  ND4_COL(0) ND4_COL(1) ND4_COL(2) ND4_COL(3) ND4_COL(4) ND4_COL(5) ND4_COL(6) ND4_COL(7)
  ND4_COL(8) ND4_COL(9) ND4_COL(10) ND4_COL(11) ND4_COL(12) ND4_COL(13) ND4_COL(14) ND4_COL(15)
#undef ND4_COL
  if (fold && w == 0 && t < NB * NB) A[(long)(fold_pj0 + t / NB) * N + j0 + t % NB] = s_top2[t / NB][t % NB];   // U12 (see the note at the top)
#pragma unroll
  for (int i = 0; i < R; i++) {
    const int r = rbase + t + T * i;
    if (r < N) {
      double* dst = A + (long)r * N + j0;
      if (vec) {
#pragma unroll
        for (int c = 0; c < W; c += 2) *reinterpret_cast<double2*>(dst + c) = double2{a[i][c], a[i][c + 1]};
      } else {
#pragma unroll
        for (int c = 0; c < W; c++) dst[c] = a[i][c];
      }
    }
  }
}
template <int R, int PQ, bool STAMPS = false>
__global__ __launch_bounds__(512) void lu_panel_mw(double* __restrict__ LU, int N, long strideM, int j0, int32_t* __restrict__ ipiv,
                                                   int nopivot, double* __restrict__ xbuf, int* __restrict__ stuck, int P,
                                                   unsigned long long* __restrict__ stamps = nullptr) {
  lu_panel_mw_body<R, PQ, STAMPS>(blockIdx.x, blockIdx.y, LU, N, strideM, j0, ipiv, nopivot, xbuf, stuck, P, stamps, nullptr, nullptr);
}
// the same panel (workgroups 0 .. P-1 of a matrix) together with everything the previous panel (at pj0) still owes the other columns
template <int R, int PQ>
__global__ __launch_bounds__(512) void lu_panel_mw_la(double* __restrict__ LU, int N, long strideM, int j0, int32_t* __restrict__ Pm,
                                                      int32_t* __restrict__ ipiv, int nopivot, double* __restrict__ xbuf,
                                                      int* __restrict__ stuck, int P, int pj0, int wide0,
                                                      const double* __restrict__ stage, long strideStage, int full_end, int fold) {
  if ((int)blockIdx.x < P) {
    const double* sg = stage != nullptr ? stage + blockIdx.y * strideStage : nullptr;
    lu_panel_mw_body<R, PQ, false>(blockIdx.x, blockIdx.y, LU, N, strideM, j0, ipiv, nopivot, xbuf, stuck, P, nullptr,
                                   sg != nullptr ? sg + NB * NB : nullptr, sg, fold ? pj0 : -1);
    return;
  }
  lu_update_block<512, true>((int)blockIdx.x - P, blockIdx.y, LU, N, strideM, pj0, wide0, ipiv, nopivot ? 0 : 1, Pm, full_end);
}
__global__ void lu_mw_poison(int32_t* __restrict__ P, long total, const int* __restrict__ stuck, int* __restrict__ status) {
  if (*stuck == 0) return;
  if (blockIdx.x == 0 && threadIdx.x == 0) qx_raise(status);          // -> ND4HIP_ERR_XCHG at the next synchronising entry point
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i < total) P[i] = -1;
}

// the next panel's own columns [c0, c0 + 16): interchanges + U12 by one workgroup, then the rank-16 update with the rows split over
// workgroups of 256 (one workgroup is bound by the MFMA rate of its CU)
__global__ __launch_bounds__(512) void lu_narrow_top(double* __restrict__ LU, int N, long strideM, const int32_t* __restrict__ ipiv,
                                                      int nopivot, int pj0, int c0) {
  lu_colblock_update<512, false>(LU + blockIdx.x * strideM, N, pj0, ipiv + (long)blockIdx.x * N, nopivot ? 0 : 1, c0, N - c0 < NB ? N - c0 : NB,
                                 true, nullptr);
}
__global__ __launch_bounds__(256) void lu_narrow_gemm(double* __restrict__ LU, int N, long strideM, int pj0, int c0) {
  double* A = LU + blockIdx.y * strideM;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, fx = lane & 15, fk = lane >> 4;
  const int m2 = N - pj0 - NB, nc = N - c0 < NB ? N - c0 : NB;
  const int rt = blockIdx.x * 256 + wave * 64;
  const bool cok = fx < nc;
  const double* Lp = A + (long)(pj0 + NB) * N + pj0;
  double* Cp = A + (long)(pj0 + NB) * N + c0;
  double bw[4], av[4][4]; d4 c[4];
#pragma unroll
  for (int kk = 0; kk < 4; kk++) bw[kk] = cok ? -A[(long)(pj0 + kk * 4 + fk) * N + c0 + fx] : 0.0;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int ra = rt + q * 16 + fx;
#pragma unroll
    for (int kk = 0; kk < 4; kk++) av[q][kk] = (ra < m2) ? Lp[(long)ra * N + kk * 4 + fk] : 0.0;
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int rc = rt + q * 16 + fk + 4 * r;
      c[q][r] = (rc < m2 && cok) ? Cp[(long)rc * N + fx] : 0.0;
    }
  }
#pragma unroll
  for (int q = 0; q < 4; q++) {
#pragma unroll
    for (int kk = 0; kk < 4; kk++) c[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q][kk], bw[kk], c[q], 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int rc = rt + q * 16 + fk + 4 * r;
      if (rc < m2 && cok) Cp[(long)rc * N + fx] = c[q][r];
    }
  }
}

// Round 3: lu_narrow_top + lu_narrow_gemm in ONE launch, out of place. Every workgroup (256 rows of the part below the panel's top
// block) works out the interchanges for itself (the gather trick of lu_colblock_update on 32 rows), gathers the 16 x 16 block above
// the next panel as it stands after them, solves U12 = L11^-1 A12 (16 threads, the order of lu.js:71-72), and updates its own rows,
// read from where the interchanges take them from: a row below the top block only ever receives the content of one of the 16 top rows
// (a pivot row trades places with a top row; a top row with another top row), so nobody reads a row that another workgroup writes —
// as long as NOTHING is written in place: the results go to a contiguous stage ([0, 256): U12; then [N - c0 rows][16]: the next
// panel's columns), from which the next panel loads its tile and puts U12 into place. One dependent launch per panel less.
__global__ __launch_bounds__(256) void lu_narrow_fused(const double* __restrict__ LU, int N, long strideM, const int32_t* __restrict__ ipiv,
                                                        int nopivot, int pj0, double* __restrict__ stage, long strideStage) {
  __shared__ int s_piv[NB], s_src[2 * NB], s_dst[2 * NB], s_map[256];
  __shared__ double s_top[NB][NB + 1], s_l[NB][NB + 1];
  const double* A = LU + blockIdx.y * strideM;
  const int32_t* ip = ipiv + (long)blockIdx.y * N;
  double* out = stage + blockIdx.y * strideStage;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, fx = lane & 15, fk = lane >> 4;
  const int c0 = pj0 + NB;                                      // the next panel's first row and column
  const int m2 = N - c0, r0 = (int)blockIdx.x * 256;           // this workgroup: rows c0 + r0 .. + 255
  if (t < NB) s_piv[t] = nopivot ? pj0 + t : ip[pj0 + t];
  {
    const int i = t / NB, j = t % NB;
    s_l[i][j] = (j < i) ? A[(long)(pj0 + i) * N + pj0 + j] : 0.0;
    s_map[t] = c0 + r0 + t;
  }
  __syncthreads();
  if (t < 2 * NB) {
    const int k = t & (NB - 1);
    const int dst = t < NB ? pj0 + k : s_piv[k];
    int pos = dst;
#pragma unroll
    for (int q = NB - 1; q >= 0; q--) {
      const int ps = s_piv[q];
      pos = (pos == pj0 + q) ? ps : ((pos == ps) ? pj0 + q : pos);
    }
    s_src[t] = pos;
    s_dst[t] = (t < NB || dst >= pj0 + NB) ? dst : -1;
  }
  __syncthreads();
  if (t >= NB && t < 2 * NB) {                                 // pivot rows below the top block that fall into this workgroup's range
    const int d = s_dst[t] - (c0 + r0);
    if (s_dst[t] >= 0 && d >= 0 && d < 256) s_map[d] = s_src[t];
  }
  {
    const int k = t / NB, c = t % NB;
    s_top[k][c] = A[(long)s_src[k] * N + c0 + c];
  }
  __syncthreads();
  if (t < NB) {                                                 // U12 = L11^-1 A12, one thread per column
    double x[NB];
#pragma unroll
    for (int i = 0; i < NB; i++) x[i] = s_top[i][t];
#pragma unroll
    for (int i = 1; i < NB; i++) {
      double acc = x[i];
#pragma unroll
      for (int j = 0; j < i; j++) acc -= s_l[i][j] * x[j];
      x[i] = acc;
    }
#pragma unroll
    for (int i = 0; i < NB; i++) { s_top[i][t] = x[i]; if (blockIdx.x == 0) out[i * NB + t] = x[i]; }
  }
  __syncthreads();
  const int rt = r0 + wave * 64;
  const double* Lp = A + (long)c0 * N + pj0;
  double bw[4], av[4][4]; d4 c[4];
#pragma unroll
  for (int kk = 0; kk < 4; kk++) bw[kk] = -s_top[kk * 4 + fk][fx];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int ra = rt + q * 16 + fx;
#pragma unroll
    for (int kk = 0; kk < 4; kk++) av[q][kk] = (ra < m2) ? Lp[(long)ra * N + kk * 4 + fk] : 0.0;
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int rc = rt + q * 16 + fk + 4 * r;
      c[q][r] = (rc < m2) ? A[(long)s_map[rc - r0] * N + c0 + fx] : 0.0;
    }
  }
#pragma unroll
  for (int q = 0; q < 4; q++) {
#pragma unroll
    for (int kk = 0; kk < 4; kk++) c[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q][kk], bw[kk], c[q], 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int rc = rt + q * 16 + fk + 4 * r;
      if (rc < m2) out[NB * NB + (long)rc * NB + fx] = c[q][r];
    }
  }
}

// ---- apply the panel's row swaps to the columns outside the panel, and (fused) U12 = L11^-1 A12 for the columns to
// its right: both are one-thread-per-column jobs over the same columns, and the 16 swapped-in pivot rows are exactly the
// rows the triangular solve works on, so they never leave the registers in between. One launch per panel instead of two.
__global__ void lu_iota(int32_t* __restrict__ P, long total, int N) {
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i < total) P[i] = (int32_t)(i % N);
}
// Pm != NULL: thread 0 of the first workgroup also carries the permutation vector through the panel's interchanges, as one more
// "column" (lu.js:59-61 swaps P together with the rows). Rebuilding P from all interchanges at the end was one thread walking a
// 2048-step dependent chain: 155 us, 2 % of the whole factorisation. (Composing the nb interchanges into one gather/scatter so
// that a column's loads are independent was tried: 11.0 us per launch against 9.7 us — the kernel is bound by its launch and
// 8 workgroups of latency, not by the chain.)
__global__ __launch_bounds__(256) void lu_laswp(double* __restrict__ LU, int N, long strideM, int j0, int nb, const int32_t* __restrict__ ipiv,
                                                int do_swap, int32_t* __restrict__ Pm, int u12_end) {
  __shared__ double s_l[NB][NB + 1];
  __shared__ int s_piv[NB];
  double* A = LU + blockIdx.y * strideM;
  const int32_t* ip = ipiv + (long)blockIdx.y * N;
  const int t = threadIdx.x;
  {
    const int i = t / NB, j = t % NB;                        // 256 threads = NB * NB
    s_l[i][j] = (i < nb && j < i) ? A[(long)(j0 + i) * N + j0 + j] : 0.0;
    if (t < NB) s_piv[t] = (t < nb) ? ip[j0 + t] : j0 + t;
  }
  __syncthreads();
  if (Pm != nullptr && do_swap && blockIdx.x == 0 && t == 255) {     // the permutation vector rides along as one more column
    int32_t* P = Pm + (long)blockIdx.y * N;
    for (int k = 0; k < nb; k++) {
      const int r = j0 + k, pv = s_piv[k];
      if (pv != r) { const int32_t x = P[r]; P[r] = P[pv]; P[pv] = x; }
    }
  }
  int col = blockIdx.x * blockDim.x + t;                   // index among the N - nb outside columns
  if (col >= N - nb) return;
  bool right = col >= j0;
  if (right) col += nb;
  right = right && col < u12_end;                          // two-level blocking: columns beyond the outer block get the interchanges only
  double x[NB];
#pragma unroll
  for (int k = 0; k < NB; k++) {                           // swap k: rows j0 + k <-> piv_k; x[k] ends up with row j0 + k
    x[k] = 0.0;
    if (k < nb) {
      const int r = j0 + k, pv = s_piv[k];
      double top = A[(long)r * N + col];
      if (do_swap && pv != r) {
        // the pivot row may itself be one of the panel's later top rows (pv < j0 + nb): then it is still in memory,
        // not yet in x[], because x[] only holds rows that were already finalised (k' < k)
        const double other = A[(long)pv * N + col];
        A[(long)pv * N + col] = top;
        top = other;
      }
      x[k] = top;
      if (!right && do_swap && pv != r) A[(long)r * N + col] = top;   // (columns without the U12 solve: the row is final as swapped)
    }
  }
  if (!right) return;
#pragma unroll
  for (int i = 1; i < NB; i++) {
    double s = x[i];
#pragma unroll
    for (int j = 0; j < i; j++) s -= s_l[i][j] * x[j];      // same order as lu.js:71-72 applied column-wise
    x[i] = s;
  }
#pragma unroll
  for (int i = 0; i < NB; i++)
    if (i < nb) A[(long)(j0 + i) * N + col] = x[i];
}

// P = the identity with the recorded row interchanges applied in order (lu.js:59-61 swaps P together with the rows).
// The panel kernels only record ipiv: the dependent global load/store chain of the swap used to sit on the critical
// path of every column (thread 0 arrived ~1000 cycles late at the barrier). One workgroup per matrix, P in LDS.
__global__ __launch_bounds__(256) void lu_build_perm(int32_t* __restrict__ Pm, const int32_t* __restrict__ ipiv, int N, int nopivot) {
  extern __shared__ int32_t s_p[];
  int32_t* P = Pm + (long)blockIdx.x * N;
  const int32_t* ip = ipiv + (long)blockIdx.x * N;
  for (int i = threadIdx.x; i < N; i += 256) s_p[i] = i;
  __syncthreads();
  if (threadIdx.x == 0 && !nopivot)
    for (int j = 0; j < N; j++) {
      const int pv = ip[j];
      if (pv != j) { const int32_t tmp = s_p[j]; s_p[j] = s_p[pv]; s_p[pv] = tmp; }
    }
  __syncthreads();
  for (int i = threadIdx.x; i < N; i += 256) P[i] = s_p[i];
}
// the same for N beyond the LDS budget: straight in global memory
__global__ void lu_build_perm_global(int32_t* __restrict__ Pm, const int32_t* __restrict__ ipiv, int N, int nopivot) {
  int32_t* P = Pm + (long)blockIdx.x * N;
  const int32_t* ip = ipiv + (long)blockIdx.x * N;
  for (int i = threadIdx.x; i < N; i += blockDim.x) P[i] = i;
  __syncthreads();
  if (threadIdx.x == 0 && !nopivot)
    for (int j = 0; j < N; j++) {
      const int pv = ip[j];
      if (pv != j) { const int32_t tmp = P[j]; P[j] = P[pv]; P[pv] = tmp; }
    }
}

template <int R>
void launch_panel_row(nd4hip_handle* h, double* LU, int N, long strideM, int j0, int nb, int32_t* P, int32_t* ipiv, int batch, int nopivot) {
  hipLaunchKernelGGL((lu_panel_row<R, NB, 512>), dim3(batch), dim3(512), 0, h->stream, LU, N, strideM, j0, nb, P, ipiv, nopivot);
}

}  // namespace

template <int R, int W, int T>
void launch_panel_row_wt(nd4hip_handle* h, double* LU, int N, long strideM, int j0, int nb, int32_t* P, int32_t* ipiv, int batch, int nopivot) {
  hipLaunchKernelGGL((lu_panel_row<R, W, T>), dim3(batch), dim3(T), 0, h->stream, LU, N, strideM, j0, nb, P, ipiv, nopivot);
}

static int getrf_impl(nd4hip_handle* h, int64_t batch, int64_t N64, const double* A, double* LU, int32_t* P, int nopivot) {
  ND4_CHECK_ARG(N64 < (1ll << 30) && batch < 65536, "nd4_getrf: extent out of range");
  const int N = (int)N64;
  const long strideM = (long)N * N;
  if (LU != A) ND4_HIP(hipMemcpyAsync(LU, A, sizeof(double) * batch * strideM, hipMemcpyDeviceToDevice, h->stream));
  const long total = (long)batch * N;
  void* ws = nullptr;
  Nd4WsScope scope(h);
  ND4_TRY(nd4_ws_alloc(h, sizeof(int32_t) * total, &ws));
  int32_t* ipiv = static_cast<int32_t*>(ws);

  // every panel of a matrix wider than one panel is followed by lu_laswp: P then rides along with it
  const bool p_in_laswp = N > NB;
  if (p_in_laswp) hipLaunchKernelGGL(lu_iota, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, h->stream, P, total, (int)N);
  int j_start = 0;
  // Two-level blocking for N > 2048 (round 3): with panels of 8 / 4 columns every step used to read-modify-write the whole trailing
  // matrix (N^3 / (3 NB) * 16 B: 180 GB at 8192^2 for 366 GFLOP). Now an outer block of 512 columns (ND4HIP_LU_OUTER; with the look-ahead panels 256: 3 % slower, 1024: 1 % faster) is factorised by the same
  // panel kernels with the rank-NB updates restricted to the block (the interchanges still go to every column at once: two rows per
  // swap), then U12 = L11^-1 A12 for the whole block row (unit lower, the one-launch solver of trsm.hip on a contiguous
  // copy) and ONE K = 256 product A22 -= L21 U12 on the tiled MFMA kernel. N <= 2048: one level (nbo = N), as before.
  static const int nbo_env = [] { const char* e = getenv("ND4HIP_LU_OUTER"); return e ? atoi(e) : 512; }();
  // Batches that fill the chip (no look-ahead form, see la_on below) are bound by the read-modify-write of every trailing matrix per
  // 16-column panel (1024 x 512^2: 45 GB for 4.3 GB of matrices): two levels for them too, outer blocks of ND4HIP_LU_BATCH_OUTER
  // columns (1024 x 512^2: one level 19.9 ms, outer 32 / 64 / 128: 16.6 / 15.8 / 14.7 ms).
  static const int la_max_batch = [] { const char* e = getenv("ND4HIP_LU_LA_MAX_BATCH"); return e ? atoi(e) : 12; }();
  static const int nbo_batch = [] { const char* e = getenv("ND4HIP_LU_BATCH_OUTER"); return e ? atoi(e) : 128; }();
  const bool batch_two_level = batch > la_max_batch && nbo_batch >= 32 && N <= 2048 && N >= 4 * nbo_batch;
  const int NBO = (N > 2048 && nbo_env >= 32) ? nbo_env : (batch_two_level ? nbo_batch : N);
  void* u12buf = nullptr;
  if (NBO < N) ND4_TRY(nd4_ws_alloc(h, sizeof(double) * (size_t)batch * NBO * N, &u12buf));
  // panels taller than 2048 rows: the rows over P co-resident workgroups of 512 threads x R rows (lu_panel_mw), 16 columns wide
  // (rows per thread R: the fewest that keep P <= 16 — N <= 8192: 1, <= 16384: 2, <= 32768: 4; measured at 8192^2: 42.3 / 43.1 / 48.4 ms
  //  for R = 1 / 2 / 4. ND4HIP_LU_MW_R = 1, 2, 4 forces one, 0 switches back to round 2's split panels.)
  const int mw_env = [] { const char* e = getenv("ND4HIP_LU_MW_R"); return e ? atoi(e) : -1; }();   // (read per call: the tests switch it)
  const int mw_r = mw_env >= 0 ? mw_env : (N <= MW_MAXP * 512 ? 1 : (N <= MW_MAXP * 1024 ? 2 : 4));
  const int mw_rt = (mw_r == 1 ? 1 : mw_r == 2 ? 2 : 4) * 512;
  const bool mw_on = mw_r != 0 && N > 2048 && (long)batch * ((N + mw_rt - 1) / mw_rt) <= 64 && N <= MW_MAXP * mw_rt;
  double* xbuf = nullptr; int* stuck = nullptr;
  if (mw_on) {
    void* xb = nullptr;
    const size_t xbytes = sizeof(double) * (size_t)batch * MW_STRIDE + 256;
    ND4_TRY(nd4_ws_alloc(h, xbytes, &xb));
    ND4_HIP(hipMemsetAsync(xb, 0, xbytes, h->stream));
    stuck = static_cast<int*>(xb);
    if (const int dp = nd4_test_drop_panel(); dp >= 0) {               // tests only: word 1 = the column tag whose publication one workgroup drops
      const int dtag = dp * NB + 1;
      ND4_HIP(hipMemcpyAsync(stuck + 1, &dtag, sizeof(int), hipMemcpyHostToDevice, h->stream));
      ND4_HIP(hipStreamSynchronize(h->stream));
    }
    xbuf = reinterpret_cast<double*>(static_cast<char*>(xb) + 256);
  }
  // the part of an outer block's work beyond it: U12 = L11^-1 A12 for the whole block row, then ONE K = nbo product on the tiled MFMA kernel
  auto outer_far = [&](const int J, const int bend) -> int {
    const int far = N - bend, nbo = bend - J;
    if (far > 0) {
      ND4_HIP(hipGetLastError());
      double* U12 = static_cast<double*>(u12buf);
      const long sU = (long)nbo * far;
      ND4_TRY(nd4_copy_matrix(h, nbo, far, LU + (long)J * N + bend, N, U12, far, batch, strideM, sU));
      ND4_TRY(nd4_trsm_ld(h, false, true, batch, nbo, far, LU + (long)J * N + J, N, strideM, U12, sU));
      ND4_TRY(nd4_copy_matrix(h, nbo, far, U12, far, LU + (long)J * N + bend, N, batch, sU, strideM));
      ND4_TRY(nd4_gemm(h, false, false, far, far, nbo, -1.0, LU + (long)bend * N + J, N, strideM, U12, far, sU,
                       1.0, LU + (long)bend * N + bend, N, strideM, batch));
    }
    return 0;
  };
  auto outer_block = [&](const int J, const int bend, const bool two_level) -> int {
    for (int j0 = J, step = NB; j0 < bend; j0 += step) {
      const int m = N - j0;
      // taller panels keep the thread-per-row layout on 1024 threads (128 VGPRs per lane) by narrowing the panel:
      // 4 rows x 8 columns up to 4096 rows, 8 rows x 4 columns up to 8192 rows
      const bool tall8 = m > 2048 && m <= 4096;
      const bool tall4 = m > 4096 && m <= 8192;
      const bool mw = mw_on && m > 2048 && bend - j0 >= NB;
      step = mw ? NB : tall8 ? 8 : tall4 ? 4 : NB;
      const int nb = bend - j0 < step ? bend - j0 : step;
      if (mw) {
        const int Pw = (m + mw_rt - 1) / mw_rt;
        const dim3 grid((unsigned)Pw, (unsigned)batch);
#define ND4_MW(RR, PQ) hipLaunchKernelGGL((lu_panel_mw<RR, PQ>), grid, dim3(512), 0, h->stream, LU, N, strideM, j0, ipiv, nopivot, xbuf, stuck, Pw)
        static const bool stamps_on = [] { const char* e = getenv("ND4HIP_LU_STAMPS"); return e && *e && *e != '0'; }();
        if (stamps_on && mw_rt == 1024 && Pw <= 4 && j0 == 0) {
          unsigned long long* st = nullptr; const size_t nst = (size_t)4 * 8 * 16 * 8;
          ND4_HIP(hipMalloc(&st, nst * 8)); ND4_HIP(hipMemset(st, 0, nst * 8));
          hipLaunchKernelGGL((lu_panel_mw<2, 1, true>), grid, dim3(512), 0, h->stream, LU, N, strideM, j0, ipiv, nopivot, xbuf, stuck, Pw, st);
          std::vector<unsigned long long> hs(nst);
          ND4_HIP(hipStreamSynchronize(h->stream)); ND4_HIP(hipMemcpy(hs.data(), st, nst * 8, hipMemcpyDeviceToHost)); ND4_HIP(hipFree(st));
          const unsigned long long t0 = hs[0];
          for (int ww = 0; ww < Pw; ww++) for (int wv = 0; wv < 8; wv += 7) for (int kk = 6; kk < 9; kk++) {
            fprintf(stderr, "mw stamps wg %d wave %d col %2d:", ww, wv, kk);
            for (int ph = 0; ph < 7; ph++) fprintf(stderr, " %8lld", (long long)(hs[(((size_t)ww * 8 + wv) * 16 + kk) * 8 + ph] - t0));
            fprintf(stderr, "\n");
          }
        } else
        if (mw_rt == 2048) { if (Pw <= 4) ND4_MW(4, 1); else if (Pw <= 8) ND4_MW(4, 2); else ND4_MW(4, 4); }
        else if (mw_rt == 1024) { if (Pw <= 4) ND4_MW(2, 1); else if (Pw <= 8) ND4_MW(2, 2); else ND4_MW(2, 4); }
        else               { if (Pw <= 4) ND4_MW(1, 1); else if (Pw <= 8) ND4_MW(1, 2); else ND4_MW(1, 4); }
#undef ND4_MW
      } else if (tall8) {
        launch_panel_row_wt<4, 8, 1024>(h, LU, N, strideM, j0, nb, P, ipiv, (int)batch, nopivot);
      } else if (tall4) {
        launch_panel_row_wt<8, 4, 1024>(h, LU, N, strideM, j0, nb, P, ipiv, (int)batch, nopivot);
      } else if (m >= 64 && m <= 2048) {
        if (m <= 512)       launch_panel_row<1>(h, LU, N, strideM, j0, nb, P, ipiv, (int)batch, nopivot);
        else if (m <= 1024) launch_panel_row<2>(h, LU, N, strideM, j0, nb, P, ipiv, (int)batch, nopivot);
        else                launch_panel_row<4>(h, LU, N, strideM, j0, nb, P, ipiv, (int)batch, nopivot);
      } else {
        int T = ((m * NB + 63) / 64) * 64; if (T > 1024) T = 1024; if (T < 64) T = 64;
        hipLaunchKernelGGL(lu_panel_global, dim3((unsigned)batch), dim3(T), 0, h->stream, LU, N, strideM, j0, nb, P, ipiv, nopivot);
      }
      const int rest = N - j0 - nb;
      if (N > nb && (!nopivot || rest > 0))
        hipLaunchKernelGGL(lu_laswp, dim3((unsigned)((N - nb + 255) / 256), (unsigned)batch), dim3(256), 0, h->stream,
                           LU, N, strideM, j0, nb, ipiv, nopivot ? 0 : 1, p_in_laswp ? P : (int32_t*)nullptr, bend);
      const int inner = bend - j0 - nb;                        // columns of the block right of the panel
      if (rest > 0 && inner > 0) {
        ND4_HIP(hipGetLastError());
        ND4_TRY(nd4_gemm(h, false, false, rest, inner, nb, -1.0,
                         LU + (long)(j0 + nb) * N + j0, N, strideM,
                         LU + (long)j0 * N + j0 + nb, N, strideM,
                         1.0, LU + (long)(j0 + nb) * N + j0 + nb, N, strideM, batch));
      }
    }
    if (two_level) ND4_TRY(outer_far(J, bend));
      return 0;
  };
  // ---- look-ahead form (see lu_colblock_update): the panel at j0 shares its launch with everything the previous panel still owes the
  //      other columns; between two panels only the next panel's 16 columns are updated (staged out of place by lu_narrow_fused).
  //      Panels of <= 2048 rows: lu_panel_row_la; taller ones (two-level blocking, the rows over P workgroups): lu_panel_mw_la.
  //      [j_from, j_to): the panels of the range; full_end: columns from there on get the interchanges only (two-level: the outer
  //      block's end; their U12 and update come from outer_far). Returns the first column it did not factorise. ----
  static const bool la_off = [] { const char* e = getenv("ND4HIP_LU_NO_LOOKAHEAD"); return e && *e && *e != '0'; }();
  static const bool fuse_off = [] { const char* e = getenv("ND4HIP_LU_NO_FUSED_NARROW"); return e && *e && *e != '0'; }();
  const bool fused = !fuse_off && (N & 1) == 0;
  const long sStage = (long)NB * NB + (long)N * NB;
  double* stage = nullptr;
  // The look-ahead form (panel p in one workgroup, panel p - 1's column-block updates riding in the same launch) hides a latency
  // chain behind otherwise idle CUs: right for one or a few matrices. A batch that fills the chip by itself is a throughput
  // problem, for which the plain sequence panel -> interchanges + U12 -> rank-16 product on the MFMA kernel moves fewer bytes per
  // launch (1024 x 512^2: 73.9 ms with the look-ahead form, 19.8 ms without; 8 matrices: 1.05 against 1.37 ms, 16: 1.64 against
  // 1.48). ND4HIP_LU_LA_MAX_BATCH moves the switch.
  const bool la_on = !la_off && N >= 64 + NB && p_in_laswp && batch <= la_max_batch;
  if (la_on) {
    void* stg = nullptr;
    ND4_TRY(nd4_ws_alloc(h, sizeof(double) * (size_t)batch * sStage, &stg));
    stage = static_cast<double*>(stg);
  }
  // (ND4HIP_LU_NO_FOLD=1: the narrow update of a multi-workgroup panel's columns as a launch of its own, as for the short panels)
  static const bool fold_off = [] { const char* e = getenv("ND4HIP_LU_NO_FOLD"); return e && *e && *e != '0'; }();
  auto la_range = [&](const int j_from, const int j_to, const int full_end, int* j_next) -> int {
    int pj0 = -1, j0 = j_from;
    bool narrow_done = true, folded = false;                  // folded: this panel's kernel does the previous panel's narrow update itself
    for (; j0 < j_to && N - j0 >= 64; j0 += NB) {
      const int m = N - j0;
      // the panel at pj0 has reached its own columns and the 16 behind them (the narrow launch: [pj0 + NB, pj0 + 2 NB) = this panel,
      // staged out of place by lu_narrow_fused); it still owes the columns from pj0 + 2 NB on, the columns left of it, and P
      const int wide0 = pj0 + 2 * NB;
      const int nupd = pj0 < 0 ? 0 : 1 + pj0 / NB + (wide0 < N ? (N - wide0 + NB - 1) / NB : 0);
      const double* sg = (fused && pj0 >= 0 && !folded) ? stage : nullptr;
      const int pp = pj0 < 0 ? 0 : pj0;
      const int fold_arg = folded ? 1 : 0;
      if (m > 2048) {
        const int Pw = (m + mw_rt - 1) / mw_rt;
        const dim3 grid((unsigned)(Pw + nupd), (unsigned)batch);
#define ND4_MWLA(RR, PQ) hipLaunchKernelGGL((lu_panel_mw_la<RR, PQ>), grid, dim3(512), 0, h->stream, LU, N, strideM, j0, P, ipiv, nopivot, xbuf, stuck, Pw, pp, wide0, sg, sStage, full_end, fold_arg)
        if (mw_rt == 2048)      { if (Pw <= 4) ND4_MWLA(4, 1); else if (Pw <= 8) ND4_MWLA(4, 2); else ND4_MWLA(4, 4); }
        else if (mw_rt == 1024) { if (Pw <= 4) ND4_MWLA(2, 1); else if (Pw <= 8) ND4_MWLA(2, 2); else ND4_MWLA(2, 4); }
        else                    { if (Pw <= 4) ND4_MWLA(1, 1); else if (Pw <= 8) ND4_MWLA(1, 2); else ND4_MWLA(1, 4); }
#undef ND4_MWLA
      } else {
        const dim3 grid((unsigned)(1 + nupd), (unsigned)batch);
        if (m <= 512)       hipLaunchKernelGGL(lu_panel_row_la<1>, grid, dim3(512), 0, h->stream, LU, N, strideM, j0, P, ipiv, nopivot, pp, wide0, sg, sStage, full_end);
        else if (m <= 1024) hipLaunchKernelGGL(lu_panel_row_la<2>, grid, dim3(512), 0, h->stream, LU, N, strideM, j0, P, ipiv, nopivot, pp, wide0, sg, sStage, full_end);
        else                hipLaunchKernelGGL(lu_panel_row_la<4>, grid, dim3(512), 0, h->stream, LU, N, strideM, j0, P, ipiv, nopivot, pp, wide0, sg, sStage, full_end);
      }
      const int c0 = j0 + NB;                                // the next panel's columns
      const bool more = c0 < j_to && N - c0 >= 64;           // another look-ahead panel follows: stage its columns out of place
      narrow_done = true;
      folded = false;
      if (more && !fold_off && mw_rt == 512 && m - NB > 2048) {
        folded = true;                                       // the next panel is a multi-workgroup one with one row per thread: it folds this in
      } else if (fused && more) {
        hipLaunchKernelGGL(lu_narrow_fused, dim3((unsigned)((m - NB + 255) / 256), (unsigned)batch), dim3(256), 0, h->stream,
                           LU, N, strideM, ipiv, nopivot, j0, stage, sStage);
      } else if (c0 < full_end && c0 < N) {
        hipLaunchKernelGGL(lu_narrow_top, dim3((unsigned)batch), dim3(512), 0, h->stream, LU, N, strideM, ipiv, nopivot, j0, c0);
        hipLaunchKernelGGL(lu_narrow_gemm, dim3((unsigned)((m - NB + 255) / 256), (unsigned)batch), dim3(256), 0, h->stream, LU, N, strideM, j0, c0);
      } else {
        narrow_done = false;                                 // the range's last panel with nothing of the outer block right of it
      }
      pj0 = j0;
    }
    if (pj0 >= 0) {   // the last panel's debt
      const int wide0 = narrow_done ? pj0 + 2 * NB : pj0 + NB;
      const int nupd = 1 + pj0 / NB + (wide0 < N ? (N - wide0 + NB - 1) / NB : 0);
      hipLaunchKernelGGL(lu_update_blocks, dim3((unsigned)nupd, (unsigned)batch), dim3(512), 0, h->stream, LU, N, strideM, P, ipiv, nopivot, pj0, wide0, full_end);
    }
    ND4_HIP(hipGetLastError());
    *j_next = j0;
    return 0;
  };
  // phase 1 (N > 2048): outer blocks while the panels are taller than 2048 rows
  if (NBO < N) {
    int J = 0;
    for (; N - J > 2048; J += NBO) {
      const int bend = J + NBO < N ? J + NBO : N;
      if (la_on && mw_on && (bend - J) % NB == 0) {
        int jn = 0;
        ND4_TRY(la_range(J, bend, bend, &jn));
        ND4_TRY(outer_far(J, bend));
      } else {
        ND4_TRY(outer_block(J, bend, true));
      }
    }
    j_start = J;
  }
  // phase 2: the look-ahead form on the whole matrix (N <= 2048) or on the trailing <= 2048 rows of a larger one
  if (la_on && N - j_start <= 2048 && N - j_start >= 64 + NB) ND4_TRY(la_range(j_start, N, N, &j_start));
  // phase 3: what is left (short panels after the look-ahead form; everything when it is switched off), one level
  if (j_start < N) {
    if (batch_two_level && !la_on) {
      for (int J = j_start; J < N; J += NBO) {
        const int bend = J + NBO < N ? J + NBO : N;
        ND4_TRY(outer_block(J, bend, bend < N));
      }
    } else {
      ND4_TRY(outer_block(j_start, N, false));
    }
  }
  if (mw_on && p_in_laswp) hipLaunchKernelGGL(lu_mw_poison, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, h->stream, P, total, stuck, h->xstat);
  if (p_in_laswp) { ND4_HIP(hipGetLastError()); return 0; }
  if ((size_t)N * sizeof(int32_t) <= 60 * 1024)
    hipLaunchKernelGGL(lu_build_perm, dim3((unsigned)batch), dim3(256), (size_t)N * sizeof(int32_t), h->stream, P, ipiv, N, nopivot);
  else
    hipLaunchKernelGGL(lu_build_perm_global, dim3((unsigned)batch), dim3(256), 0, h->stream, P, ipiv, N, nopivot);
  ND4_HIP(hipGetLastError());
  return 0;
}

int nd4_getrf(nd4hip_handle* h, int64_t batch, int64_t N, const double* A, double* LU, int32_t* P) {
  return getrf_impl(h, batch, N, A, LU, P, 0);
}
// Gaussian elimination WITHOUT pivoting (only the signs of the pivots are used: tall-QR sign convention, qr.hip)
int nd4_getrf_nopivot(nd4hip_handle* h, int64_t batch, int64_t N, const double* A, double* LU, int32_t* P) {
  return getrf_impl(h, batch, N, A, LU, P, 1);
}
