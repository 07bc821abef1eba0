// Batched QR panels on the matrix cores: ONE workgroup per panel, no exchange between workgroups (round 4).
// Included by qr.hip inside its anonymous namespace, after qr_panel_row_body, qrh_chol16 and qrh_gj16.
//
// Replaces, for batches of panels, the thread-per-row Householder kernel qr_panel_row (the inner step of qr_decomp,
// src/la/qr.js:54-68): that kernel walks 16 dependent column steps of ~530 instructions per wave, and a batch that fills the chip
// is bound by their issue (21.6 % of the HBM peak for 256 x 2048 rows). Here a panel is factorised the way the row-split panels
// of one matrix are (qrh_*: CholeskyQR2 + the compact orthogonal completion, see the comment above HR_PIVOT_THR), with the
// whole panel in the registers of ONE workgroup:
//   rows            thread-per-row: thread t keeps R rows of 16 doubles (row j0 + t + 64 NWV i), 16-byte loads / stores of its own
//                   128-byte row segment (ld, ldv, j0 even); every element is read once and (as V) written once
//   G = C^T C       on fp64 MFMA 16x16x4: a wave stages 64 rows in its own 8 KB of LDS (the 16-byte pieces of a row XOR-swizzled
//                   with the row number: the row-wise writes and the tile-wise reads are both conflict free) and reads them back
//                   as 16-row tiles in the accumulator image, which is the A and the B operand of a Gram product at once
//   R1 = chol(G)    one wave, blocked by 4 on the matrix core (qr_chain16.h), which also decides the fall-back
//                   (qr_panel_row_body in this same workgroup)
//   Q1 = C R1^-1    forward substitution in the rows' registers, R1 as wave-uniform LDS broadcasts: 136 FMAs per row (the
//                   triangle; the matrix-core form multiplies by the full inverse), no re-layout
//   R2, Q = Q1 R2^-1  E = Q1^T Q1 - I by the same Gram routine, R2 = I + F from the fixed-point series on one wave (chol when
//                   max|E| > HR_SERIES_MAX), the same substitution again: Q orthonormal to O(eps)
//   (V, T, R)       V = Q - [S; 0], T = K = -S (Q_top - S)^-T (qrh_gj16, in the shadow of the row stores), R = S R2 R1
// ZERO: rows below the top block of the panel's columns in W are zeroed (the in-matrix panels of the batched QR driver); the
// panel entry point leaves them alone, so a panel moves its algorithmic 16 m b bytes and nothing else.

// A staged slot = 64 rows of 128 bytes in a wave's own 8 KB of LDS; the 16-byte piece p of row l sits at piece p ^ qrb_sw(l).
// sw(l) = (l1, l2, l0 ^ l3) (bits of l) makes every access pattern of the kernel conflict free: row-wise 16-byte writes and reads
// (lane = row: 8 consecutive rows give 8 different pieces; the 16 lanes of a ds_read_b128 group — {0-3, 12-15, 20-27}, ... — give
// 16 different (row parity, piece) pairs), line-wise 16-byte writes and reads (lane = (row 8k + l / 8, piece l % 8): the global
// side of the kernel) and the 8-byte tile reads of the Gram product (two rows of opposite parity per group).
__device__ __forceinline__ int qrb_sw(int l) { return ((l >> 1) & 3) | (((l ^ (l >> 3)) & 1) << 2); }
__device__ __forceinline__ int qrb_off(int l, int c) { return l * 16 + ((((c >> 1) ^ qrb_sw(l)) << 1) | (c & 1)); }

// Gram matrix of the wave's R x 64 rows -> g (accumulator image: g[r] of lane (fx, fk) = G[fk + 4 r][fx]); with s_top != nullptr
// the first 16-row tile of slot 0 is left out of g and its own Gram matrix goes to s_top (row major)
// LINES: the slot arrives from global memory as whole lines (ln[i][k] of lane l = piece l % 8 of row 8 k + l / 8 of slot i) and is
// staged from there; the rows' registers a are then READ from the staged slot (thread-per-row), after the tile reads of the Gram product.
template <int R, bool LINES = false>
__device__ __forceinline__ void qrb_gram(double* __restrict__ s_st, double (&a)[R][NB], double* __restrict__ s_top, d4& g,
                                         const double2 (*ln)[8] = nullptr) {
  int lane = threadIdx.x & 63;
  asm volatile("" : "+v"(lane));                                       // (see qrb_park)
  const int fx = lane & 15, fk = lane >> 4;
  d4 acc0 = d4{0.0, 0.0, 0.0, 0.0}, acc1 = acc0;
  // The DS operations of one wave execute in order, so slot i + 1 is written right behind the READS of slot i and lands while the
  // 16 MFMAs of slot i run: the matrix pipe no longer idles through a write -> wait -> read round trip per slot.
  auto stage = [&](auto ic) {
    constexpr int I = decltype(ic)::value;
    if constexpr (I < R) {
      if constexpr (LINES) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
          const int row = 8 * k + (lane >> 3);
          *reinterpret_cast<double2*>(s_st + row * 16 + (((lane & 7) ^ qrb_sw(row)) << 1)) = ln[I][k];
        }
      } else {
#pragma unroll
        for (int p = 0; p < 8; p++)
          *reinterpret_cast<double2*>(s_st + lane * 16 + ((p ^ qrb_sw(lane)) << 1)) = double2{a[I][2 * p], a[I][2 * p + 1]};
      }
    }
  };
  auto slot = [&](auto ic) {
    constexpr int I = decltype(ic)::value;
    if constexpr (I < R) {
      if constexpr (LINES) {
#pragma unroll
        for (int p = 0; p < 8; p++) {
          const double2 v = *reinterpret_cast<const double2*>(s_st + lane * 16 + ((p ^ qrb_sw(lane)) << 1));
          a[I][2 * p] = v.x; a[I][2 * p + 1] = v.y;
        }
      }
      double c[16];
#pragma unroll
      for (int u = 0; u < 16; u++) c[u] = s_st[qrb_off(16 * (u >> 2) + 4 * (u & 3) + fk, fx)];
      asm volatile("" ::: "memory");                                   // (the next slot's writes stay behind these reads)
      stage(std::integral_constant<int, I + 1>{});
#pragma unroll
      for (int q = 0; q < 4; q++) {
        if (q == 0 && I == 0 && s_top != nullptr) {                    // (uniform) the panel's top block: a Gram matrix of its own
          d4 gt = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int r = 0; r < 4; r++) gt = __builtin_amdgcn_mfma_f64_16x16x4f64(c[r], c[r], gt, 0, 0, 0);
#pragma unroll
          for (int r = 0; r < 4; r++) s_top[(fk + 4 * r) * 16 + fx] = gt[r];
        } else {
          acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(c[4 * q], c[4 * q], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(c[4 * q + 1], c[4 * q + 1], acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(c[4 * q + 2], c[4 * q + 2], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(c[4 * q + 3], c[4 * q + 3], acc1, 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  stage(std::integral_constant<int, 0>{});
  slot(std::integral_constant<int, 0>{});
  slot(std::integral_constant<int, 1>{});
  slot(std::integral_constant<int, 2>{});
  slot(std::integral_constant<int, 3>{});
#pragma unroll
  for (int r = 0; r < 4; r++) g[r] = acc0[r] + acc1[r];
}

// x R = c for every row, in place (R upper triangular: s_Rt[j * 16 + l] = R[l][j], s_rd[j] = 1 / R[j][j]; wave-uniform reads)
template <int R, typename F>
__device__ __forceinline__ void qrb_trsolve(double (&a)[R][NB], const double* s_Rt, const double* s_rd, F&& probe) {
#pragma unroll
  for (int j = 0; j < NB; j++) {
    if (j == 4 || j == 8 || j == 12) probe(j / 4 - 1);
    double acc[R];
#pragma unroll
    for (int i = 0; i < R; i++) acc[i] = a[i][j];
#pragma unroll
    for (int l = 0; l < j; l++) {
      const double rv = s_Rt[j * 16 + l];
#pragma unroll
      for (int i = 0; i < R; i++) acc[i] = fma(-a[i][l], rv, acc[i]);
    }
    const double rd = s_rd[j];
#pragma unroll
    for (int i = 0; i < R; i++) a[i][j] = acc[i] * rd;
    // the column's results pinned here: left alone, the compiler requests all 136 entries of R up front and spills them
#pragma unroll
    for (int i = 0; i < R; i++) asm volatile("" : "+v"(a[i][j]) : : "memory");
    __builtin_amdgcn_sched_barrier(0);
  }
}

// The chains below run on wave 0 alone, in the registers of a kernel whose every wave holds a tile of 32 R registers: left alone,
// the compiler spills part of the tile to scratch memory around them (for all waves: a round trip to HBM of ~10000 cycles in the
// hot path). Wave 0 parks its last NP row slots in the staging slots instead (its own and wave 1's, idle behind the barrier), which
// frees their registers for the length of the chain.
template <int R, int NP>
__device__ __forceinline__ void qrb_park(const double (&a)[R][NB], double* __restrict__ s_st, int lane) {
  asm volatile("" : "+v"(lane));                                       // (addresses recomputed here, not kept or spilled across phases)
#pragma unroll
  for (int u = 0; u < NP; u++) {
#pragma unroll
    for (int p = 0; p < 8; p++)
      *reinterpret_cast<double2*>(s_st + u * 1024 + lane * 16 + ((p ^ qrb_sw(lane)) << 1)) = double2{a[R - 1 - u][2 * p], a[R - 1 - u][2 * p + 1]};
  }
}
template <int R, int NP>
__device__ __forceinline__ void qrb_unpark(double (&a)[R][NB], const double* __restrict__ s_st, int lane) {
  asm volatile("" : "+v"(lane));
#pragma unroll
  for (int u = 0; u < NP; u++) {
#pragma unroll
    for (int p = 0; p < 8; p++) {
      const double2 v = *reinterpret_cast<const double2*>(s_st + u * 1024 + lane * 16 + ((p ^ qrb_sw(lane)) << 1));
      a[R - 1 - u][2 * p] = v.x; a[R - 1 - u][2 * p + 1] = v.y;
    }
  }
}

// the fall-back as a function of its own: inlined, its 16 unrolled column steps would share the register allocation of the hot path
template <int R, int NWV>
__device__ __attribute__((noinline)) void qrb_fallback(int mat, double* Wm, int M, long ld, long strideW, double* Vall, long ldv, long strideV,
                                                       double* Tall, long strideT, double* taus, long strideTau, int j0) {
  qr_panel_row_body<R, NWV>(mat, Wm, M, ld, strideW, Vall, ldv, strideV, Tall, strideT, taus, strideTau, j0, NB);
}

constexpr int QRB_SMALL = 7 * 256 + 64;                                // doubles of LDS besides the per-wave staging and partial slots
template <int NWV> constexpr int qrb_lds_doubles() { return NWV * (1024 + 256) + QRB_SMALL; }

// The data movement of qrb_panel alone (debug, ND4HIP_QRB_COPY_ONLY=1: the entry point then copies A to V and computes nothing): the
// ceiling the panel kernel's own access pattern sets — thread-per-row 16-byte loads and stores, one workgroup of 64 NWV threads per
// panel, every workgroup streaming its own contiguous panel. Measured (round 4, bench protocol of ops.qr_panel): 256 x 2048 rows
// 40.0 us = 3.35 TB/s = 42 % of the HBM peak, 2048 x 512 rows 71.7 us = 47 % (mode 3 = each row slot stored as soon as it has
// arrived: 38.9 / 70.2 us); the same bytes through a plain device copy (tools/copy_bw.py): 24.6 us = 5.5 TB/s. A variant with
// full-line accesses (a wave instruction = 8 whole 128-byte rows) was SLOWER: 50.8 / 95.5 us.
template <int R, int NWV>
__global__ __launch_bounds__(64 * NWV, 2) void qrb_copy_only(const double* __restrict__ Wm, int M, long ld, long strideW,
                                                             double* __restrict__ Vall, long ldv, long strideV, int j0, int mode) {
  constexpr int TT = 64 * NWV;
  const int mat = blockIdx.x, t = threadIdx.x;
  if (mode == 2) {                                                     // full-line accesses: a wave instruction = 8 whole 128-byte rows
    const double* A3 = Wm + mat * strideW;
    double* V3 = Vall + mat * strideV;
    const int lane = t & 63, wave = t >> 6;
    double2 v[R][8];
#pragma unroll
    for (int i = 0; i < R; i++) {
#pragma unroll
      for (int k = 0; k < 8; k++) {
        int r = j0 + wave * 64 + TT * i + 8 * k + (lane >> 3);
        r = r < M ? r : M - 1;                                           // (unconditional loads: all in flight)
        v[i][k] = *reinterpret_cast<const double2*>(A3 + (long)r * ld + j0 + 2 * (lane & 7));
      }
    }
#pragma unroll
    for (int i = 0; i < R; i++) {
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const int r = j0 + wave * 64 + TT * i + 8 * k + (lane >> 3);
        if (r < M) *reinterpret_cast<double2*>(V3 + (long)r * ldv + j0 + 2 * (lane & 7)) = v[i][k];
      }
    }
    return;
  }
  if (mode == 4) {                                                     // non-temporal loads and stores
    const double* A3 = Wm + mat * strideW;
    double* V3 = Vall + mat * strideV;
    typedef double dd2 __attribute__((ext_vector_type(2)));
    dd2 v[R][8];
#pragma unroll
    for (int i = 0; i < R; i++) {
      const int r = j0 + t + TT * i;
      if (r < M) {
#pragma unroll
        for (int c = 0; c < 8; c++) v[i][c] = __builtin_nontemporal_load(reinterpret_cast<const dd2*>(A3 + (long)r * ld + j0 + 2 * c));
      }
    }
#pragma unroll
    for (int i = 0; i < R; i++) {
      const int r = j0 + t + TT * i;
      if (r < M) {
#pragma unroll
        for (int c = 0; c < 8; c++) __builtin_nontemporal_store(v[i][c], reinterpret_cast<dd2*>(V3 + (long)r * ldv + j0 + 2 * c));
      }
    }
    return;
  }
  if (mode == 3) {                                                     // every row slot stored as soon as it has arrived
    const double* A3 = Wm + mat * strideW;
    double* V3 = Vall + mat * strideV;
#pragma unroll
    for (int i = 0; i < R; i++) {
      const int r = j0 + t + TT * i;
      if (r < M) {
        double2 v[8];
#pragma unroll
        for (int c = 0; c < 8; c++) v[c] = *reinterpret_cast<const double2*>(A3 + (long)r * ld + j0 + 2 * c);
#pragma unroll
        for (int c = 0; c < 8; c++) *reinterpret_cast<double2*>(V3 + (long)r * ldv + j0 + 2 * c) = v[c];
      }
      asm volatile("" ::: "memory");
    }
    return;
  }
  const double* A = Wm + mat * strideW;
  double* V = Vall + mat * strideV;
  double a[R][NB];
#pragma unroll
  for (int i = 0; i < R; i++) {
    const int r = j0 + t + TT * i;
    if (r < M) {
      const double* src = A + (long)r * ld + j0;
#pragma unroll
      for (int c = 0; c < NB; c += 2) { const double2 v = *reinterpret_cast<const double2*>(src + c); a[i][c] = v.x; a[i][c + 1] = v.y; }
    }
  }
#pragma unroll
  for (int i = 0; i < R; i++) {
    const int r = j0 + t + TT * i;
    if (r < M) {
      double* v = V + (long)r * ldv + j0;
#pragma unroll
      for (int c = 0; c < NB; c += 2) *reinterpret_cast<double2*>(v + c) = double2{a[i][c], a[i][c + 1]};
    }
  }
}

template <int R, int NWV, bool ZERO>
__global__ __launch_bounds__(64 * NWV, 2) void qrb_panel(double* __restrict__ Wm, int M, long ld, long strideW,
                                                      double* __restrict__ Vall, long ldv, long strideV,
                                                      double* __restrict__ Tall, long strideT,
                                                      double* __restrict__ taus, long strideTau, int j0, long long* __restrict__ stamps) {
  constexpr int TT = 64 * NWV;
  __shared__ __attribute__((aligned(16))) double s_mem[qrb_lds_doubles<NWV>()];
  __shared__ int s_flag;
  double* s_st = s_mem;                                   // NWV x 1024: the waves' staging slots
  double* s_part = s_mem + NWV * 1024;                    // NWV x 256: the waves' partial Gram matrices
  double* s_X = s_part + NWV * 256;                       // 3 x 256, wave 0 only: {-, Gtop, -}, later {Z, K, -}
  double* s_R = s_X + 3 * 256;                            // R1 (row major)
  double* s_Rt = s_R + 256;                               // R1, column j contiguous
  double* s_R2t = s_Rt + 256;                             // R2, column j contiguous
  double* s_Rm = s_R2t + 256;                             // R2 R1
  double* s_rd = s_Rm + 256; double* s_rd2 = s_rd + 16; double* s_db = s_rd2 + 16; double* s_S = s_db + 16;
  const int mat = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6, fx = lane & 15, fk = lane >> 4;
  double* A = Wm + mat * strideW;
  double* V = Vall + mat * strideV;
  auto stamp = [&](int k) { if (stamps != nullptr && mat == 0 && t == 0) { stamps[2 * k] = wall_clock64(); stamps[2 * k + 1] = clock64(); } };   // 100 MHz wall clock, shader clock
  stamp(0);

  // The panel comes in as whole 128-byte lines (a wave instruction = 8 rows: lane l takes piece l % 8 of row 8 k + l / 8 of its 64-row
  // slot): the thread-per-row form of the same loads reached 3.4 TB/s on its own, this one 4.3 (qrb_copy_only, modes 1 / 2). The
  // lines are staged in the wave's LDS slot, from which both the tiles of the Gram product and the rows' registers are read.
  double a[R][NB];
  {
    double2 ln[R][8];
#pragma unroll
    for (int i = 0; i < R; i++) {
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const int r = j0 + 64 * wave + TT * i + 8 * k + (lane >> 3);
        const int rc = r < M ? r : M - 1;                              // (unconditional loads: all of them in flight at once)
        const double2 v = *reinterpret_cast<const double2*>(A + (long)rc * ld + j0 + 2 * (lane & 7));
        ln[i][k] = r < M ? v : double2{0.0, 0.0};
      }
    }
    // ---- G = C^T C (the top 16 rows apart: a column that is zero below them must take the fall-back) ----
    d4 g;
    qrb_gram<R, true>(s_st + wave * 1024, a, wave == 0 ? s_X + 256 : nullptr, g, ln);
#pragma unroll
    for (int r = 0; r < 4; r++) s_part[wave * 256 + (fk + 4 * r) * 16 + fx] = g[r];
  }
  stamp(1);
  __syncthreads();
  stamp(2);
  constexpr int NP = (NWV >= 2 && R >= 2) ? 2 : 1;                     // row slots wave 0 parks in LDS around its chains
  if (wave == 0) {
    qrb_park<R, NP>(a, s_st, lane);
    // G in the accumulator image (register r of lane (fx, fk): G[4 r + fk][fx]); R1 = chol(G) blocked by 4 on the matrix core
    d4 g;
    bool zero_below = false;
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int e = (fk + 4 * r) * 16 + fx;
      double x = 0.0;
#pragma unroll
      for (int w = 0; w < NWV; w++) x += s_part[w * 256 + e];
      g[r] = x + s_X[256 + e];
      zero_below = zero_below || (fx == fk + 4 * r && !(x > 0.0));     // a column with nothing below the top block
    }
    double rr[4];
    const bool ok = qrc_chol16(g, HR_PIVOT_THR, rr, s_rd) && __ballot(zero_below) == 0ull;
#pragma unroll
    for (int r = 0; r < 4; r++) {
      s_R[(fk + 4 * r) * 16 + fx] = rr[r];
      s_Rt[fx * 16 + fk + 4 * r] = rr[r];
    }
    if (lane == 0) s_flag = ok ? 0 : 1;
    qrb_unpark<R, NP>(a, s_st, lane);
    stamp(3);
  }
  __syncthreads();
  stamp(4);
  if (__builtin_amdgcn_readfirstlane(s_flag)) {                        // (scalar branch) the classic panel, from memory: nothing was written yet
    qrb_fallback<R, NWV>(mat, Wm, M, ld, strideW, Vall, ldv, strideV, Tall, strideT, taus, strideTau, j0);
    return;
  }
  // ---- Q1 = C R1^-1, E = Q1^T Q1 - I ----
  qrb_trsolve<R>(a, s_Rt, s_rd, [&](int k) { stamp(14 + k); });
  stamp(5);
  {
    d4 g;
    qrb_gram<R>(s_st + wave * 1024, a, nullptr, g);
#pragma unroll
    for (int r = 0; r < 4; r++) s_part[wave * 256 + (fk + 4 * r) * 16 + fx] = g[r];
  }
  stamp(6);
  __syncthreads();
  stamp(7);
  if (wave == 0) {
    qrb_park<R, NP>(a, s_st, lane);
    // R2 = chol(I + E) = I + F with F = triu(E - F^T F) (diagonal halved), two fixed-point steps from F = triu(E): error O(|E|^3).
    // All in the accumulator image: F^T F is a Gram product of the lane's own registers, R = R2 R1 reads R2 back transposed.
    d4 ev;
    double emax = 0.0;
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int e = (fk + 4 * r) * 16 + fx;
      double x = 0.0;
#pragma unroll
      for (int w = 0; w < NWV; w++) x += s_part[w * 256 + e];
      x -= (fx == fk + 4 * r) ? 1.0 : 0.0;
      ev[r] = x;
      const double ax = fabs(x);
      emax = (ax <= emax) ? emax : ax;                                 // a NaN wins
    }
    emax = nd4dpp::wave_max(emax);
    double r2[4];
    if (emax <= HR_SERIES_MAX) {
      d4 pp = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int i = fk + 4 * r;
        const double f0 = (i < fx) ? ev[r] : ((i == fx) ? 0.5 * ev[r] : 0.0);
        pp = __builtin_amdgcn_mfma_f64_16x16x4f64(f0, f0, pp, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int i = fk + 4 * r;
        const double x = ev[r] - pp[r];
        r2[r] = ((i < fx) ? x : ((i == fx) ? 0.5 * x : 0.0)) + ((i == fx) ? 1.0 : 0.0);
        if (i == fx) s_rd2[i] = 1.0 / r2[r];
      }
    } else {
      d4 g2;
#pragma unroll
      for (int r = 0; r < 4; r++) g2[r] = ev[r] + ((fx == fk + 4 * r) ? 1.0 : 0.0);
      (void)qrc_chol16(g2, 0.0, r2, s_rd2);
    }
#pragma unroll
    for (int r = 0; r < 4; r++) s_R2t[fx * 16 + fk + 4 * r] = r2[r];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    d4 rm = d4{0.0, 0.0, 0.0, 0.0};                                    // R = R2 R1 (the sign S comes with the top block)
#pragma unroll
    for (int s4 = 0; s4 < 4; s4++)
      rm = __builtin_amdgcn_mfma_f64_16x16x4f64(s_R2t[(4 * s4 + fk) * 16 + fx], s_R[(4 * s4 + fk) * 16 + fx], rm, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; r++) s_Rm[(fk + 4 * r) * 16 + fx] = rm[r];
    qrb_unpark<R, NP>(a, s_st, lane);
    stamp(8);
  }
  __syncthreads();
  stamp(9);
  // ---- Q = Q1 R2^-1; V = Q - [S; 0] ----
  qrb_trsolve<R>(a, s_R2t, s_rd2, [&](int k) { stamp(17 + k); });
  stamp(10);
  // Wave 0 first completes the top block (the 16-step elimination behind S and K: qrh_gj16), while the other waves' row stores
  // fill the memory pipeline, and stores its own rows last: behind its own row stores the top block's few stores waited in the
  // queue for up to 10 us.
  double* s_Z = s_X; double* s_K = s_X + 256;
  double sg = 0.0;
  if (wave == 0) {
    if (t < NB) {
#pragma unroll
      for (int c = 0; c < NB; c++) s_Z[t * 16 + c] = a[0][c];
    }
    qrb_park<R, 1>(a, s_st, lane);                                     // (its own slot only: the other waves are staging their stores in theirs)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    qrc_gj16(s_Z, s_K, s_S);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    qrb_unpark<R, 1>(a, s_st, lane);
    if (t < NB) sg = s_S[t];
  }
  stamp(11);
  // V goes out as whole lines too: each slot from the rows' registers through the wave's LDS slot (row-wise writes, line-wise
  // reads), one slot behind the next one's writes. The top block's rows carry V = Q - S; R = S R2 R1 is written by its 16 lanes.
  if (t < NB) {
#pragma unroll
    for (int c = 0; c < NB; c++) a[0][c] -= (c == t) ? sg : 0.0;
    double* w = A + (long)(j0 + t) * ld + j0;
#pragma unroll
    for (int c = 0; c < NB; c += 2)
      *reinterpret_cast<double2*>(w + c) = double2{(t <= c) ? sg * s_Rm[t * 16 + c] : 0.0, (t <= c + 1) ? sg * s_Rm[t * 16 + c + 1] : 0.0};
  }
  {
    double* s_w = s_st + wave * 1024;
    int ln_ = lane;
    asm volatile("" : "+v"(ln_));                                      // (see qrb_park)
#pragma unroll
    for (int i = 0; i < R; i++) {
#pragma unroll
      for (int p = 0; p < 8; p++)
        *reinterpret_cast<double2*>(s_w + ln_ * 16 + ((p ^ qrb_sw(ln_)) << 1)) = double2{a[i][2 * p], a[i][2 * p + 1]};
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      double2 out[8];
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const int row = 8 * k + (ln_ >> 3);
        out[k] = *reinterpret_cast<const double2*>(s_w + row * 16 + (((ln_ & 7) ^ qrb_sw(row)) << 1));
      }
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const int r = j0 + 64 * wave + TT * i + 8 * k + (ln_ >> 3);
        if (r < M) {
          *reinterpret_cast<double2*>(V + (long)r * ldv + j0 + 2 * (ln_ & 7)) = out[k];
          if constexpr (ZERO) {
            if (r >= j0 + NB) *reinterpret_cast<double2*>(A + (long)r * ld + j0 + 2 * (ln_ & 7)) = double2{0.0, 0.0};
          }
        }
      }
    }
  }
  stamp(12);
  if (wave == 0) {
    if (t < NB) taus[mat * strideTau + j0 + t] = 1.0;                  // "a reflector was needed" (qr_flips)
#pragma unroll
    for (int k = 0; k < 4; k++) Tall[mat * strideT + (long)(j0 / NB) * NB * NB + lane + 64 * k] = s_K[lane + 64 * k];
  }
  stamp(13);
}
