// The 16 x 16 chains of a QR panel on ONE wave, with the matrix core doing the eliminations (round 4).
// Included by qr.hip (anonymous namespace).
//
// A 16 x 16 matrix lives in the accumulator image of an fp64 MFMA 16x16x4: register r of lane (fx, fk) (fx = lane & 15,
// fk = lane >> 4) is M[4 r + fk][fx]. Row k of M is register k / 4 at the 16 lanes fk == k % 4 — exactly where k-slot k % 4 of a B
// operand (lane (n, kk): B[kk][n]) and of an A operand (lane (m, kk): A[m][kk]) is read. So a rank-1 update  M -= x y^T  with x
// and y derived from ROW k of M (or of M^T, kept alongside) is one MFMA whose operands are "that row where fk == k % 4, zero
// elsewhere": no readlane broadcast of 15 multipliers, no 15 row updates. The unblocked chains this replaces (qrh_chol16 / qrh_gj16:
// per pivot 2 readlanes + 1 FMA for each remaining row, ~60 instructions of ONE wave per step at ~8 cycles each) took 3.5 and 4.4 us
// of a 30-40 us panel; a step is now the pivot's readlane, its reciprocal (square root), a few selects and one to three MFMAs.
// (A 4 x 4-blocked Cholesky with the diagonal block factorised redundantly by all lanes was measured first: 2.5 us, bound by the
// ~110 wave-uniform instructions per block.)

__device__ __forceinline__ double qrc_rl(double v, int lane) { return nd4dpp::rl_d(v, lane); }

// ---- Cholesky: G = R^T R ----
template <int K>
__device__ __forceinline__ void qrc_chol_step(d4& g, double (&rrow)[4], int fx, int fk) {
  constexpr int RG = K / 4, SL = K % 4;
  const double p = qrc_rl(g[RG], 16 * SL + K);                         // the pivot G[K][K]
  const double rs = nd4dpp::fast_rsqrt(p);
  const bool slot = fk == SL;
  const double u = (slot && fx >= K) ? g[RG] * rs : 0.0;               // row K of R: G[K][fx] / sqrt(p)
  rrow[RG] = slot ? u : rrow[RG];
  if constexpr (K < 15) g = __builtin_amdgcn_mfma_f64_16x16x4f64(-u, u, g, 0, 0, 0);   // G -= u u^T (symmetric: u is column K too)
}
template <int... K>
__device__ __forceinline__ void qrc_chol_all(d4& g, double (&rrow)[4], int fx, int fk, std::integer_sequence<int, K...>) {
  (qrc_chol_step<K>(g, rrow, fx, fk), ...);
}
// One wave. g: symmetric positive definite 16 x 16 in the accumulator image. Out: rrow = R = chol(G)^T in the accumulator image
// (G = R^T R), s_rd[i] = 1 / R[i][i] to a few ulp (may be nullptr). Returns true when every pivot is positive and >= thr * its
// diagonal entry.
__device__ __forceinline__ bool qrc_chol16(d4 g, double thr, double (&rrow)[4], double* __restrict__ s_rd) {
  const int lane = threadIdx.x & 63, fx = lane & 15, fk = lane >> 4;
  const d4 g0 = g;
#pragma unroll
  for (int r = 0; r < 4; r++) rrow[r] = 0.0;
  qrc_chol_all(g, rrow, fx, fk, std::make_integer_sequence<int, 16>{});
  // the pivots are the squares of R's diagonal (a negative or NaN pivot leaves a NaN there); the lane that holds R[i][i] also holds
  // the diagonal entry G[i][i] it is measured against
  bool bad = false;
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const double d = rrow[r], o = g0[r];
    const bool diag = fx == 4 * r + fk;
    bad = bad || (diag && !(d > 0.0 && d * d >= thr * o && o < DBL_MAX));
    if (diag && s_rd != nullptr) s_rd[fx] = nd4dpp::fast_rcp(d);
  }
  return __ballot(bad) == 0ull;
}

// ---- Gauss-Jordan inversion of B = Z - S in place, S_k = -sign(pivot) chosen on the way (every pivot p - s has magnitude >= 1) ----
// In-place step k on T (with uf = column k, r = row k, p = T[k][k], inv = 1 / (p - s), c = (p + 1) inv):
//   T' = T - uf w1^T + e_k w2^T,   w1 = r inv (entry k: inv + 1),   w2 = r (c - 1) (entry k: c)
// i.e. T'[i][j] = T[i][j] - T[i][k] T[k][j] inv, T'[i][k] = -T[i][k] inv, T'[k][j] = T[k][j] inv, T'[k][k] = inv. Row k of T is
// rewritten exactly after the MFMA; the transpose is carried along (column k of T is row k of T^T: the A operand), its column k
// fixed by the second rank-1 term.
template <int K>
__device__ __forceinline__ void qrc_gj_step(d4& t, d4& tt, double (&sreg)[4], int fx, int fk) {
  constexpr int RG = K / 4, SL = K % 4;
  const double p = qrc_rl(t[RG], 16 * SL + K);
  const double s = (p >= 0.0) ? -1.0 : 1.0;
  const double inv = nd4dpp::fast_rcp(p - s);
  const double c = (p + 1.0) * inv;
  const bool slot = fk == SL, col = fx == K;
  const double r = t[RG], uf = tt[RG];                                 // at the slot lanes: T[K][fx], T[fx][K]
  const double v = col ? inv : r * inv;                                // the new row K
  const double w1 = col ? inv + 1.0 : v;
  const double w2 = col ? c : r * (c - 1.0);
  t = __builtin_amdgcn_mfma_f64_16x16x4f64(slot ? -uf : 0.0, slot ? w1 : 0.0, t, 0, 0, 0);
  t[RG] = slot ? v : t[RG];
  tt = __builtin_amdgcn_mfma_f64_16x16x4f64(slot ? -w1 : 0.0, slot ? uf : 0.0, tt, 0, 0, 0);
  tt = __builtin_amdgcn_mfma_f64_16x16x4f64(slot ? w2 : 0.0, (slot && col) ? 1.0 : 0.0, tt, 0, 0, 0);
  sreg[RG] = slot ? s : sreg[RG];
}
template <int... K>
__device__ __forceinline__ void qrc_gj_all(d4& t, d4& tt, double (&sreg)[4], int fx, int fk, std::integer_sequence<int, K...>) {
  (qrc_gj_step<K>(t, tt, sreg, fx, fk), ...);
}
// One wave. s_Z: top 16 x 16 block of the orthonormal Q, row major. Out: s_S[16] (signs) and s_K = K = -S B^-T, B = Z - S (row major).
__device__ __forceinline__ void qrc_gj16(const double* __restrict__ s_Z, double* __restrict__ s_K, double* __restrict__ s_S) {
  const int lane = threadIdx.x & 63, fx = lane & 15, fk = lane >> 4;
  d4 t, tt;
  double sreg[4];
#pragma unroll
  for (int r = 0; r < 4; r++) { t[r] = s_Z[(4 * r + fk) * 16 + fx]; tt[r] = s_Z[fx * 16 + 4 * r + fk]; sreg[r] = 0.0; }
  qrc_gj_all(t, tt, sreg, fx, fk, std::make_integer_sequence<int, 16>{});
  // K[c][r] = -S_c B^-1[r][c]: the transposed image, row c scaled by -S_c
#pragma unroll
  for (int r = 0; r < 4; r++) {
    s_K[(4 * r + fk) * 16 + fx] = -sreg[r] * tt[r];
    if (fx == 0) s_S[4 * r + fk] = sreg[r];
  }
}
