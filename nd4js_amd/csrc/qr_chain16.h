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

// ---- Cholesky with the inverse factor: the row operations of the elimination applied to the identity as well (one more MFMA per
//      pivot, independent of the first) leave Y = (unit lower L)^-1; R^-1[r][c] = Y[c][r] / R[c][c] ----
template <int K>
__device__ __forceinline__ void qrc_chol_inv_step(d4& g, d4& y, double (&rrow)[4], double (&rsreg)[4], int fx, int fk) {
  constexpr int RG = K / 4, SL = K % 4;
  const double p = qrc_rl(g[RG], 16 * SL + K);
  const double rs = nd4dpp::fast_rsqrt(p);
  const bool slot = fk == SL;
  const double u = (slot && fx >= K) ? g[RG] * rs : 0.0;
  rrow[RG] = slot ? u : rrow[RG];
  rsreg[RG] = slot ? rs : rsreg[RG];
  if constexpr (K < 15) {
    const double mneg = (fx > K) ? -(u * rs) : 0.0;                    // -G[fx][K] / p for the rows below the pivot (zero outside the k-slot: u is)
    y = __builtin_amdgcn_mfma_f64_16x16x4f64(mneg, slot ? y[RG] : 0.0, y, 0, 0, 0);   // rows i > K of Y -= m_i * row K of Y
    g = __builtin_amdgcn_mfma_f64_16x16x4f64(-u, u, g, 0, 0, 0);
  }
}
template <int... K>
__device__ __forceinline__ void qrc_chol_inv_all(d4& g, d4& y, double (&rrow)[4], double (&rsreg)[4], int fx, int fk, std::integer_sequence<int, K...>) {
  (qrc_chol_inv_step<K>(g, y, rrow, rsreg, fx, fk), ...);
}
// One wave; drop-in for qrh_chol16: s_G symmetric positive definite 16 x 16 (row major, all 256 entries). Out: s_R = chol(G)^T
// (upper, G = R^T R) and s_Ri = R^-1 (upper), row major. Returns true when every pivot is positive and >= thr * its diagonal entry.
__device__ __forceinline__ bool qrc_chol16_inv(const double* __restrict__ s_G, double* __restrict__ s_R, double* __restrict__ s_Ri, double thr) {
  const int lane = threadIdx.x & 63, fx = lane & 15, fk = lane >> 4;
  d4 g, y;
  double rrow[4], rsreg[4];
#pragma unroll
  for (int r = 0; r < 4; r++) { g[r] = s_G[(4 * r + fk) * 16 + fx]; y[r] = (fx == 4 * r + fk) ? 1.0 : 0.0; rrow[r] = 0.0; rsreg[r] = 0.0; }
  const d4 g0 = g;
  qrc_chol_inv_all(g, y, rrow, rsreg, fx, fk, std::make_integer_sequence<int, 16>{});
  bool bad = false;
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const double d = rrow[r], o = g0[r];
    bad = bad || (fx == 4 * r + fk && !(d > 0.0 && d * d >= thr * o && o < DBL_MAX));
    s_R[(4 * r + fk) * 16 + fx] = d;
    s_Ri[fx * 16 + 4 * r + fk] = (fx <= 4 * r + fk) ? y[r] * rsreg[r] : 0.0;   // R^-1[fx][c] = Y[c][fx] / R[c][c], c = 4 r + fk
  }
  return __ballot(bad) == 0ull;
}

// ---- Gauss-Jordan inversion of B = Z - S in place, S_k = -sign(pivot) chosen on the way (every pivot p - s has magnitude >= 1) ----
// In-place step k on T (with uf = column k, r = row k, p = T[k][k], inv = 1 / (p - s), c = (p + 1) inv):
//   T' = T - uf w1^T + e_k w2^T,   w1 = r inv (entry k: inv + 1),   w2 = r (c - 1) (entry k: c)
// i.e. T'[i][j] = T[i][j] - T[i][k] T[k][j] inv, T'[i][k] = -T[i][k] inv, T'[k][j] = T[k][j] inv, T'[k][k] = inv. Row k of T is
// rewritten exactly after the MFMA; the transpose is carried along (column k of T is row k of T^T: the A operand), its column k
// fixed by the second rank-1 term, which rides in another k-slot of the same MFMA (w2 crosses 32 lanes by v_permlane32_swap).
// value of lane ^ 32 (the same fx, k-slot fk ^ 2), valid in the lanes of k-slot SL2: v_permlane32_swap exchanges lanes [32, 64) of
// its first operand with lanes [0, 32) of its second
template <int SL2>
__device__ __forceinline__ double qrc_from_other_half(double v) {
  const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(v), __double2loint(v), false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(v), __double2hiint(v), false, false);
  return SL2 >= 2 ? __hiloint2double(hi[0], lo[0]) : __hiloint2double(hi[1], lo[1]);
}
template <int K>
__device__ __forceinline__ void qrc_gj_step(d4& t, d4& tt, double (&sreg)[4], int fx, int fk) {
  constexpr int RG = K / 4, SL = K % 4, SL2 = SL ^ 2;
  const double p = qrc_rl(t[RG], 16 * SL + K);
  const double s = (p >= 0.0) ? -1.0 : 1.0;
  const double inv = nd4dpp::fast_rcp(p - s);
  const double cm1 = fma(p + 1.0, inv, -1.0);                          // c - 1
  const bool slot = fk == SL, col = fx == K;
  const double ecol = col ? 1.0 : 0.0, e2 = (fk == SL2 && col) ? 1.0 : 0.0;   // (constants of the step)
  const double r = t[RG], uf = tt[RG];                                 // at the slot lanes: T[K][fx], T[fx][K]
  const double vk = col ? inv : r * inv;                               // the new row K
  const double w1n = fma(vk, -1.0, -ecol);                             // -w1
  const double w2 = col ? cm1 + 1.0 : r * cm1;
  const double w2s = qrc_from_other_half<SL2>(w2);                     // w2 in the lanes of k-slot SL2
  // T -= uf w1^T: only the B operand is masked to its k-slot (the other slots of A meet zeros there)
  t = __builtin_amdgcn_mfma_f64_16x16x4f64(uf, slot ? w1n : 0.0, t, 0, 0, 0);
  t[RG] = slot ? vk : t[RG];
  // T^T += (-w1) uf^T in k-slot SL and w2 e_K^T in k-slot SL2: one MFMA
  tt = __builtin_amdgcn_mfma_f64_16x16x4f64(fk == SL2 ? w2s : w1n, slot ? uf : e2, tt, 0, 0, 0);
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

// ---- the small products of CholeskyQR2's second pass on one wave (round 3: 256 threads, four barriers, ~2.7 us) ----
// images of a row-major 16 x 16 matrix in LDS: the accumulator image (B operand "rows 4r..4r+3") and the image of its transpose
// (A operand "columns 4r..4r+3")
__device__ __forceinline__ d4 qrc_img(const double* __restrict__ s, int fx, int fk) {
  d4 x;
#pragma unroll
  for (int r = 0; r < 4; r++) x[r] = s[(4 * r + fk) * 16 + fx];
  return x;
}
__device__ __forceinline__ d4 qrc_img_t(const double* __restrict__ s, int fx, int fk) {
  d4 x;
#pragma unroll
  for (int r = 0; r < 4; r++) x[r] = s[fx * 16 + 4 * r + fk];
  return x;
}
__device__ __forceinline__ void qrc_put(double* __restrict__ s, const d4& x, int fx, int fk) {
#pragma unroll
  for (int r = 0; r < 4; r++) s[(4 * r + fk) * 16 + fx] = x[r];
}
// A B for A given as the image of A^T and B as its accumulator image
__device__ __forceinline__ d4 qrc_mul(const d4& at, const d4& b) {
  d4 c = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int r = 0; r < 4; r++) c = __builtin_amdgcn_mfma_f64_16x16x4f64(at[r], b[r], c, 0, 0, 0);
  return c;
}
#define QRC_LDS_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
// One wave. s_E = Q1^T Q1 - I (row major). Out (row major): s_R2 = chol(I + E)^T and s_R2i = R2^-1.
// series: R2 = I + F, F = triu(E - F0^T F0) with the diagonal halved, F0 = triu(E) likewise (two fixed-point steps: error O(|E|^3));
// R2^-1 = (I - F)(I + F^2) (error O(|F|^4)). Otherwise (max|E| > HR_SERIES_MAX) the elimination chain on I + E. s_tmp: 256 doubles.
__device__ __forceinline__ void qrc_series16(double* __restrict__ s_E, bool series, double* __restrict__ s_R2, double* __restrict__ s_R2i,
                                             double* __restrict__ s_tmp) {
  const int lane = threadIdx.x & 63, fx = lane & 15, fk = lane >> 4;
  if (!series) {
#pragma unroll
    for (int r = 0; r < 4; r++) if (fx == 4 * r + fk) s_E[fx * 17] += 1.0;
    QRC_LDS_SYNC();
    (void)qrc_chol16_inv(s_E, s_R2, s_R2i, 0.0);
    QRC_LDS_SYNC();
    return;
  }
  const d4 ev = qrc_img(s_E, fx, fk);
  d4 f0, f1, eye;
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int i = 4 * r + fk;
    eye[r] = (i == fx) ? 1.0 : 0.0;
    f0[r] = (i < fx) ? ev[r] : ((i == fx) ? 0.5 * ev[r] : 0.0);
  }
  d4 pp = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int r = 0; r < 4; r++) pp = __builtin_amdgcn_mfma_f64_16x16x4f64(f0[r], f0[r], pp, 0, 0, 0);      // F0^T F0: a Gram product of the image with itself
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int i = 4 * r + fk;
    const double x = ev[r] - pp[r];
    f1[r] = (i < fx) ? x : ((i == fx) ? 0.5 * x : 0.0);
  }
  qrc_put(s_tmp, f1, fx, fk);
  qrc_put(s_R2, f1 + eye, fx, fk);
  QRC_LDS_SYNC();
  const d4 f1t = qrc_img_t(s_tmp, fx, fk);
  const d4 f2 = qrc_mul(f1t, f1);                                      // F F
  const d4 r2i = qrc_mul(eye - f1t, f2 + eye);                         // (I - F)(I + F^2)   (the identity is symmetric: one image)
  qrc_put(s_R2i, r2i, fx, fk);
  QRC_LDS_SYNC();
}
// One wave. s_Z = Qt R2^-1 (the top block of Q) and s_Rm = R2 R1, row major in and out.
__device__ __forceinline__ void qrc_zr16(const double* __restrict__ s_Qt, const double* __restrict__ s_R2i, const double* __restrict__ s_R2,
                                         const double* __restrict__ s_R1, double* __restrict__ s_Z, double* __restrict__ s_Rm) {
  const int lane = threadIdx.x & 63, fx = lane & 15, fk = lane >> 4;
  const d4 z = qrc_mul(qrc_img_t(s_Qt, fx, fk), qrc_img(s_R2i, fx, fk));
  const d4 rm = qrc_mul(qrc_img_t(s_R2, fx, fk), qrc_img(s_R1, fx, fk));
  qrc_put(s_Z, z, fx, fk);
  qrc_put(s_Rm, rm, fx, fk);
  QRC_LDS_SYNC();
}
