/* oracle/nd4_oracle_svd_dc.c — CPU restatement of the reference's svd_decomp (= svd_dc: bidiagonalisation + divide & conquer).
 * TEST INFRASTRUCTURE ONLY (see nd4_oracle.h): included by nd4_oracle.c, never linked into the product.
 *
 * Restates, statement by statement and in the reference's operation order (-ffp-contract=off):
 *   /root/reference/src/la/svd_dc.js:37-61    _svd_dc_1x2
 *   /root/reference/src/la/svd_dc.js:72-143   _svd_dc_2x3
 *   /root/reference/src/la/svd_dc.js:169-657  _svd_dc_neves (deflation, duplicate diagonals, secular equations by bisection,
 *                                             the Gu-Eisenstat recomputation of z, the updates of U and V)
 *   /root/reference/src/la/svd_dc.js:666-824  _svd_dc_bidiag (divide, the middle row, merge order, conquer)
 *   /root/reference/src/la/svd_dc.js:827-880  _svd_dc (bidiagonalisation by _bidiag_decomp_horiz, bidiag.js:164-242, restated
 *                                             as bidiag_horiz1 in nd4_oracle.c; back-multiplication :864-879)
 *   /root/reference/src/la/svd_dc.js:883-932  svd_dc (the M > N branch through the transpose)
 * Pinned by tests/test_oracle_golden.py against the fixtures oracle/gen_golden.js generated from the reference's
 * nd.la.svd_decomp (c1_svd32, mid_svd96, wide / tall / batched / edge families). */

#define SVD_DC_FAIL(code) do { return (code); } while (0)
#define JS_MIN_VALUE 4.9406564584124654e-324                    /* Number.MIN_VALUE */
#define JS_EPSILON   2.220446049250313e-16                      /* Number.EPSILON = eps('float64'), dt/index.js:33-43 */

static double fro_result(const fro_t* f) { return isfinite(f->max) ? sqrt(f->sum) * f->max : f->max; }   /* norm.js:59-62 */
static double js_sign(double x) { return x > 0 ? 1.0 : (x < 0 ? -1.0 : x); }                            /* Math.sign: +-0 and NaN pass */

/* svd_dc.js:37-61 */
static void svd_dc_1x2(int64_t N, double* U, int64_t U_off, double* F, int64_t B_off, int64_t V_off) {
  double c, s, norm;
  nd4o_giv_rot_qr(F[B_off], F[B_off + 1], &c, &s, &norm);
  if (0 != norm) {
    F[B_off] = norm;
    F[B_off + 1] = NAN;
    F[V_off] = c;
    F[V_off + 1] = -s;
    F[V_off + N * 1] = s;
    F[V_off + N * 1 + 1] = c;
  } else {
    F[V_off] = 1;
    F[V_off + N * 1 + 1] = 1;
  }
  U[U_off] = 1;
}

/* svd_dc.js:72-143 */
static void svd_dc_2x3(int64_t N, double* U, int64_t U_off, double* F, int64_t B_off, int64_t V_off) {
  const int64_t M = N - 1;
  double b1 = F[B_off], b2 = F[B_off + 1], b3 = F[B_off + 2], b4 = F[B_off + 3];
  if (0 != b4) {
    double ca, sa, norm;
    nd4o_giv_rot_qr(b3, b4, &ca, &sa, &norm);
    F[V_off + N * 1 + 1] = ca;
    F[V_off + N * 2 + 1] = sa;
    b3 = norm;
    b4 = -sa * b2;
    b2 = ca * b2;
    if (0 != b4) {
      double cb, sb, norm2;
      nd4o_giv_rot_qr(b1, b4, &cb, &sb, &norm2);
      b1 = norm2;
      F[V_off] = cb;
      F[V_off + N * 1] = sb * -sa;
      F[V_off + N * 2] = sb * ca;
      F[V_off + 2] = -sb;
      F[V_off + N * 1 + 2] = cb * -sa;
      F[V_off + N * 2 + 2] = cb * ca;
    } else {
      F[V_off] = 1;
      F[V_off + N * 1 + 2] = -sa;
      F[V_off + N * 2 + 2] = ca;
    }
  } else {
    F[V_off] = 1;
    F[V_off + N * 1 + 1] = 1;
    F[V_off + N * 2 + 2] = 1;
  }
  double ca, sa, cb, sb;
  svd_jac_angles(b1, b2, 0, b3, &ca, &sa, &cb, &sb);
  F[B_off + 1] = NAN;
  F[B_off + 3] = NAN;
  F[B_off] = ca * b1 * cb - (ca * b2 + sa * b3) * sb;
  b3 = (-sa * b2 + ca * b3) * cb - sa * b1 * sb;
  const double s = b3 < 0 ? -1 : +1;
  F[B_off + 2] = s * b3;
  U[U_off] = ca;
  U[U_off + 1] = -sa * s;
  U[U_off + M * 1] = sa;
  U[U_off + M * 1 + 1] = ca * s;
  F[V_off + 1] = sb * F[V_off];
  F[V_off] *= cb;
  const double V1 = F[V_off + N * 1] * cb - F[V_off + N * 1 + 1] * sb;
  F[V_off + N * 1 + 1] = F[V_off + N * 1] * sb + F[V_off + N * 1 + 1] * cb;
  F[V_off + N * 1] = V1;
  const double V2 = F[V_off + N * 2] * cb - F[V_off + N * 2 + 1] * sb;
  F[V_off + N * 2 + 1] = F[V_off + N * 2] * sb + F[V_off + N * 2 + 1] * cb;
  F[V_off + N * 2] = V2;
}

/* svd_dc.js:169-657. Returns 0, or the line of the reference assertion that failed. */
static int svd_dc_neves(int64_t N, int64_t n, double* U, int64_t U_off, double* F, int64_t B_off, int64_t V_off, int32_t* I) {
  const int64_t M = N - 1, m = n - 1;
  const int64_t s_off = M * (M + 2) - m,          /* σ_off */
                mm_off = M * (M + 2) - m * 2,
                W_off = M * (M + 2) - m * (m + 2);
  const int64_t rot_off = m * 2, inn_off = m, out_off = 0;
  if (n < 2) SVD_DC_FAIL(195);
  if (F[B_off + 2 * m - 2] != 0) SVD_DC_FAIL(202);
  for (int64_t i = 1; i < m; i++)
    if (F[B_off + 2 * (i - 1)] < F[B_off + 2 * i]) SVD_DC_FAIL(206);

  fro_t NORM = {0.0, 0.0};
  double zNorm, scale;
  {                                                                          /* :211-225 */
    for (int64_t i = 0; i < m; i++) fro_include(&NORM, F[B_off + 2 * i + 1]);
    zNorm = fro_result(&NORM);
    for (int64_t i = 0; i < m; i++) fro_include(&NORM, F[B_off + 2 * i]);
    scale = fro_result(&NORM);
    if (scale == 0) scale = 1;
    zNorm = zNorm / scale;
  }
  for (int64_t i = 0; i < 2 * m; i++) F[B_off + i] /= scale;                 /* :228-229 */
  const double TOL = JS_EPSILON;

  /* STEP 1: deflation (:261-281) */
  int64_t n0 = 0;
  {
    for (int64_t j = m - 1, i = m - 1; i-- > 0;) {
      const double di = F[B_off + 2 * i], zi = F[B_off + 2 * i + 1];
      const int32_t oi = I[out_off + i];
      if (fabs(zi) / TOL <= di) {
        F[s_off + n0] = di;
        I[inn_off + n0] = oi;
        ++n0;
      } else {
        --j;
        F[B_off + 2 * j] = di;
        F[B_off + 2 * j + 1] = F[B_off + 2 * i + 1];
        I[inn_off + j] = oi;
      }
    }
  }
  for (int64_t i = 0; i < n0; i++) I[out_off + i] = I[inn_off + i];          /* :284-285 */

  /* STEP 2: duplicate values on the diagonal (:346-378) */
  int64_t n1 = n0;
  {
    for (int64_t j = m - 1, i = m - 1; i-- > n0;) {
      const double di = F[B_off + 2 * i], dj = F[B_off + 2 * j];
      const int32_t oi = I[inn_off + i];
      const double zi = F[B_off + 2 * i + 1], zj = F[B_off + 2 * j + 1];
      double c, s, z;
      nd4o_giv_rot_qr(zj, zi, &c, &s, &z);
      if ((di - dj) / TOL <= di || !isfinite(sqrt((double)m) / (di - dj))) {
        F[W_off + 2 * n1] = c;
        F[W_off + 2 * n1 + 1] = s;
        F[B_off + 2 * j + 1] = z;
        F[B_off + 2 * j] = F[s_off + n1] = di;
        I[out_off + n1] = oi;
        I[rot_off + n1] = (int32_t)j;
        ++n1;
      } else {
        --j;
        F[B_off + 2 * j] = di;
        F[B_off + 2 * j + 1] = F[B_off + 2 * i + 1];
        I[out_off + j] = oi;
      }
    }
  }
  for (int64_t i = 2 * n0; i < 2 * n1; i++) {                                /* :374-378 */
    F[B_off + i] = F[W_off + i];
    F[W_off + i] = 0;
  }

  /* STEP 3: the secular equations by bisection (:388-436) */
  for (int64_t i = n1; i < m; i++) {
    double sLo = F[B_off + 2 * i], sHi = n1 < i ? F[B_off + 2 * i - 2] : (sLo + zNorm);
    if (sLo > sHi) SVD_DC_FAIL(393);
    double shift;
    {
      int taken = 0;
      shift = 0;
      if (i > n1) {
        const double mid = (sLo + sHi) / 2;
        double sum = 1;
        for (int64_t k = n1; k < m; k++) {
          const double di = F[B_off + 2 * k], zi = F[B_off + 2 * k + 1];
          sum += zi / (di - mid) * (zi / (di + mid));
        }
        if (!isfinite(sum)) SVD_DC_FAIL(405);
        if (sum < 0) { const double s = sHi; sLo = sLo - sHi; sHi = -JS_MIN_VALUE; shift = s; taken = 1; }
      }
      if (!taken) { const double s = sLo; sHi = sHi - sLo; sLo = +JS_MIN_VALUE; shift = s; }
    }
    for (;;) {
      const double s = (sLo + sHi) / 2;
      if (s == sLo || s == sHi) { F[s_off + i] = s; break; }
      double sum = 1;
      for (int64_t k = n1; k < m; k++) {
        const double di = F[B_off + 2 * k], zi = F[B_off + 2 * k + 1];
        sum += zi / (di - shift - s) * (zi / (di + shift + s));
      }
      if (!isfinite(sum)) SVD_DC_FAIL(429);
      if (sum <= 0) sLo = s;
      if (sum >= 0) sHi = s;
    }
  }
  if (fabs(F[B_off + 2 * m - 1]) == 0) {                                     /* :437-440 */
    F[B_off + 2 * m - 2] = 0;
    F[B_off + 2 * m - 1] = 0; F[s_off + m - 1] = 0;
  }

  /* STEP 4: recompute z (:446-472) */
  {
    const double sn_ = F[s_off + m - 1],                                      /* σn */
                 sn = F[B_off + 2 * (m - 1 - (sn_ < 0))];
    for (int64_t i = n1; i < m; i++) {
      const double di = F[B_off + 2 * i];
      double zi = (sn - di + sn_) * (sn + di + sn_);
      for (int64_t j = n1; j < i; j++) {
        const double sj_ = F[s_off + j], sj = F[B_off + 2 * (j - (sj_ < 0))], dj = F[B_off + 2 * j];
        zi *= ((sj - di + sj_) / (dj - di)) * ((sj + di + sj_) / (dj + di));
      }
      for (int64_t j = i; j < m - 1; j++) {
        const double sj_ = F[s_off + j], sj = F[B_off + 2 * (j - (sj_ < 0))], dj = F[B_off + 2 * j + 2];
        zi *= ((sj - di + sj_) / (dj - di)) * ((sj + di + sj_) / (dj + di));
      }
      F[B_off + 2 * i + 1] = js_sign(F[B_off + 2 * i + 1]) * sqrt(zi);
    }
  }

  /* triple merge of the singular values from the deflations and from the secular equations (:476-488) */
  for (int64_t h = n0 - 1, i = n1 - 1, j = n1, k = 0; k < m; k++) {
    double val = -INFINITY;
    int best = 3;
    if (j < m) { const double sj_ = F[s_off + j]; best = 2; val = sj_ + F[B_off + 2 * (j - (sj_ < 0))]; }
    if (i >= n0) { const double si_ = F[s_off + i]; if (!(si_ < val)) { best = 1; val = si_; } }
    if (h >= 0) { const double sh_ = F[s_off + h]; if (!(sh_ < val)) { best = 0; val = sh_; } }
    switch (best) {
      case 0: I[inn_off + h--] = (int32_t)k; continue;
      case 1: I[inn_off + i--] = (int32_t)k; continue;
      case 2: I[inn_off + j++] = (int32_t)k; continue;
      default: SVD_DC_FAIL(486);
    }
  }

  /* STEP 5: update U (:493-560) */
  for (int64_t i = n1; i < m; i++) {
    const double si_ = F[s_off + i], si = F[B_off + 2 * (i - (si_ < 0))];
    NORM.sum = NORM.max = 0;
    for (int64_t j = n1; j < m - 1; j++) {
      const double dj = F[B_off + 2 * j], zj = F[B_off + 2 * j + 1],
                   W_ij = (zj / (dj - si - si_)) * (dj / (dj + si + si_));
      fro_include(&NORM, F[W_off + m * i + j] = W_ij);
    }
    fro_include(&NORM, F[W_off + m * i + m - 1] = -1);
    const double norm = fro_result(&NORM);
    if (!(0 < norm)) SVD_DC_FAIL(507);
    for (int64_t j = n1; j < m; j++) F[W_off + m * i + j] /= norm;
  }
  for (int64_t i = n1; i < m; i++)                                           /* :514-519 */
    for (int64_t j = i; ++j < m;) {
      const double W_ij = F[W_off + m * i + j];
      F[W_off + m * i + j] = F[W_off + m * j + i];
      F[W_off + m * j + i] = W_ij;
    }
  if (n0 < n1) {                                                             /* :521-541 */
    for (int64_t i = n0; i < n1; i++) {
      for (int64_t e = W_off + m * i + n0; e < W_off + m * i + m; e++) F[e] = 0.0;
      F[W_off + m * i + i] = 1;
    }
    for (int64_t i = n1; i < m; i++)
      for (int64_t e = W_off + m * i + n0; e < W_off + m * i + n1; e++) F[e] = 0.0;
    for (int64_t i = n1; i-- > n0;) {
      const int64_t j = I[rot_off + i];
      if (j < m - 1) {
        const double c = F[B_off + 2 * i], s = F[B_off + 2 * i + 1];
        giv_rot_rows(F, m - i, W_off + m * i + i, W_off + m * j + i, c, s);
      }
    }
  }
  for (int64_t r = 0; r < m; r++) {                                          /* U = U W^T (:544-566) */
    for (int64_t e = mm_off + n0; e < mm_off + m; e++) F[e] = 0.0;
    for (int64_t i = n0; i < m; i++) {
      const double U_ri = U[U_off + M * r + I[out_off + i]];
      if (0 != U_ri) for (int64_t j = n0; j < m; j++) F[mm_off + j] += U_ri * F[W_off + m * i + j];
    }
    for (int64_t i = 0; i < n0; i++) {
      const int64_t c = I[out_off + i];
      F[mm_off + i] = U[U_off + M * r + c];
    }
    for (int64_t i = 0; i < m; i++) {
      const int64_t c = I[inn_off + i];
      U[U_off + M * r + c] = F[mm_off + i];
    }
  }

  /* STEP 6: update V (:571-645) */
  for (int64_t i = n1; i < m; i++) {
    const double si_ = F[s_off + i], si = F[B_off + 2 * (i - (si_ < 0))];
    NORM.sum = NORM.max = 0;
    for (int64_t j = n1; j < m; j++) {
      const double dj = F[B_off + 2 * j], zj = F[B_off + 2 * j + 1],
                   W_ij = zj / (dj - si - si_) / (dj + si + si_);
      fro_include(&NORM, F[W_off + m * i + j] = W_ij);
    }
    const double norm = fro_result(&NORM);
    if (!(0 < norm || i == m - 1)) SVD_DC_FAIL(583);
    for (int64_t j = n1; j < m; j++) F[W_off + m * i + j] /= norm;
  }
  if (0 == F[B_off + 2 * m - 1]) {                                           /* :589-593 */
    for (int64_t i = n1; i < m - 1; i++) F[W_off + m * (m - 1) + i] = 0;
    F[W_off + m * (m - 1) + (m - 1)] = 1;
  }
  for (int64_t i = n1; i < m; i++)                                           /* :596-601 */
    for (int64_t j = i; ++j < m;) {
      const double W_ij = F[W_off + m * i + j];
      F[W_off + m * i + j] = F[W_off + m * j + i];
      F[W_off + m * j + i] = W_ij;
    }
  if (n0 < n1) {                                                             /* :603-621 */
    for (int64_t i = n0; i < n1; i++) {
      for (int64_t e = W_off + m * i + n0; e < W_off + m * i + m; e++) F[e] = 0.0;
      F[W_off + m * i + i] = 1;
    }
    for (int64_t i = n1; i < m; i++)
      for (int64_t e = W_off + m * i + n0; e < W_off + m * i + n1; e++) F[e] = 0.0;
    for (int64_t i = n1; i-- > n0;) {
      const int64_t j = I[rot_off + i];
      const double c = F[B_off + 2 * i], s = F[B_off + 2 * i + 1];
      giv_rot_rows(F, m - i, W_off + m * i + i, W_off + m * j + i, c, s);
    }
  }
  for (int64_t r = 0; r < n; r++) {                                          /* V = V W^T (:625-646) */
    for (int64_t e = mm_off + n0; e < mm_off + m; e++) F[e] = 0.0;
    for (int64_t i = n0; i < m; i++) {
      const double V_ri = F[V_off + N * r + I[out_off + i]];
      if (0 != V_ri) for (int64_t j = n0; j < m; j++) F[mm_off + j] += V_ri * F[W_off + m * i + j];
    }
    for (int64_t i = 0; i < n0; i++) {
      const int64_t c = I[out_off + i];
      F[mm_off + i] = F[V_off + N * r + c];
    }
    for (int64_t i = 0; i < m; i++) {
      const int64_t c = I[inn_off + i];
      F[V_off + N * r + c] = F[mm_off + i];
    }
  }

  /* STEP 7 (:651-658) */
  for (int64_t i = n1; i < m; i++) F[s_off + i] += F[B_off + 2 * (i - (F[s_off + i] < 0))];
  for (int64_t i = 0; i < m; i++) {
    const int64_t j = I[inn_off + i];
    F[B_off + 2 * j] = F[s_off + i] * scale;
    F[B_off + 2 * j + 1] = NAN;
  }
  return 0;
}

/* svd_dc.js:666-824 */
static int svd_dc_bidiag(int64_t N, int64_t n, double* U, int64_t U_off, double* F, int64_t B_off, int64_t V_off, int32_t* I) {
  if (n > N) SVD_DC_FAIL(682);
  if (1 >= n) SVD_DC_FAIL(683);
  if (2 == n) { svd_dc_1x2(N, U, U_off, F, B_off, V_off); return 0; }
  if (3 == n) { svd_dc_2x3(N, U, U_off, F, B_off, V_off); return 0; }
  const int64_t M = N - 1, m = n - 1, n0 = n >> 1, m0 = n0 - 1;
  int rc = svd_dc_bidiag(N, n0, U, U_off, F, B_off, V_off, I);
  if (rc) return rc;
  rc = svd_dc_bidiag(N, n - n0, U, U_off + M * n0 + n0, F, B_off + 2 * n0, V_off + N * n0 + n0, I);
  if (rc) return rc;
  U[U_off + M * m0 + m0] = 1;
  const double b1 = F[B_off + 2 * m0], b2 = F[B_off + 2 * m0 + 1];
  for (int64_t i = 0; i < m0; i++) F[B_off + 2 * i + 1] = b1 * F[V_off + N * m0 + i];          /* :741-742 */
  for (int64_t i = n0; i < m; i++) F[B_off + 2 * i + 1] = b2 * F[V_off + N * n0 + i];
  double c, s, h;
  nd4o_giv_rot_qr(b1 * F[V_off + N * m0 + m0], b2 * F[V_off + N * n0 + m], &c, &s, &h);         /* :779-782 */
  F[B_off + 2 * m0] = 0;
  F[B_off + 2 * m0 + 1] = h;
  if (0 != h) {                                                                                  /* :789-794 */
    for (int64_t i = 0; i < n0; i++) { F[V_off + N * i + m] = F[V_off + N * i + m0] * -s; F[V_off + N * i + m0] *= c; }
    for (int64_t i = n0; i < n; i++) { F[V_off + N * i + m0] = F[V_off + N * i + m] * s; F[V_off + N * i + m] *= c; }
  }
  I[m - 1] = (int32_t)m0;                                                                        /* :803-809 */
  for (int64_t i = 0, j = n0, k = 0; k < m - 1; k++)
    I[k] = (int32_t)((j >= m || (i < m0 && F[B_off + 2 * i] >= F[B_off + 2 * j])) ? i++ : j++);
  for (int64_t i = 0; i < m; i++) {                                                              /* :813-822 */
    const int64_t j = I[i];
    F[2 * i] = F[B_off + 2 * j];
    F[2 * i + 1] = F[B_off + 2 * j + 1];
  }
  for (int64_t i = 0; i < m; i++) {
    F[B_off + 2 * i] = F[2 * i];
    F[B_off + 2 * i + 1] = F[2 * i + 1];
  }
  return svd_dc_neves(N, n, U, U_off, F, B_off, V_off, I);
}

/* svd_dc.js:827-880: one M x N matrix with M <= N. V holds A on entry. U is zero on entry. */
static int svd_dc1(int64_t M, int64_t N, double* U, double* sv, double* V, int32_t* I, double* F) {
  if (M > N) SVD_DC_FAIL(829);
  const int64_t B_off = M * (M + 2), V1_off = B_off + M * 2, V2_off = V1_off + (M + 1) * (M + 1);
  for (int64_t e = V1_off; e < V2_off + N; e++) F[e] = 0.0;
  for (int64_t i = 0; i < M; i++)
    for (int64_t j = 0; j < N; j++) F[V2_off + N + N * i + j] = V[N * i + j];
  for (int64_t e = 0; e < M * N; e++) V[e] = 0.0;
  bidiag_horiz1(M, N, U, F, F + V2_off);                                      /* _bidiag_decomp_horiz(M,N, U,U_off, F,0, F,V2_off) */
  for (int64_t i = 0; i < M; i++) {
    F[B_off + 2 * i] = F[(M + 1) * i + i];
    F[B_off + 2 * i + 1] = F[(M + 1) * i + i + 1];
  }
  const int rc = svd_dc_bidiag(M + 1, M + 1, V, 0, F, B_off, V1_off, I);
  if (rc) return rc;
  for (int64_t i = 0; i < M; i++) sv[i] = F[B_off + 2 * i];
  for (int64_t i = 0; i < M; i++) {                                           /* U = U U2 (U2 sits in V) (:864-870) */
    for (int64_t j = 0; j < M; j++) F[j] = 0.0;
    for (int64_t k = 0; k < M; k++)
      for (int64_t j = 0; j < M; j++) F[j] += U[M * i + k] * V[M * k + j];
    for (int64_t j = 0; j < M; j++) U[M * i + j] = F[j];
  }
  for (int64_t e = 0; e < M * M; e++) V[e] = 0.0;                            /* V = V1 V2 (:873-879; V.fill(0.0, V_off, V_off + M*M)) */
  for (int64_t k = 0; k < M + 1; k++)
    for (int64_t i = 0; i < M; i++)
      for (int64_t j = 0; j < N; j++) V[N * i + j] += F[V1_off + (M + 1) * k + i] * F[V2_off + N * k + j];
  return 0;
}

/* svd_dc.js:883-932 svd_dc (= nd.la.svd_decomp, svd.js:25): A [batch, M, N] -> U [batch, M, L], sv [batch, L], V [batch, L, N],
 * L = min(M, N). M > N goes through the transpose like the reference (:892-896). Returns 0, or the line number of the
 * reference assertion that failed ("Assertion failed." thrown by the reference), or -1 when out of memory. */
int nd4o_svd_dc(int64_t batch, int64_t M, int64_t N, const double* A, double* U, double* sv, double* V) {
  if (M > N) {
    /* [U', sv, V'] = svd_dc(A^T): U' [N,N], V' [N,M]; result U = V'^T [M,N], V = U'^T [N,N] */
    double* At = (double*)malloc(sizeof(double) * (size_t)(batch * M * N + batch * N * N + batch * N * M));
    if (!At) return -1;
    double* Up = At + batch * M * N; double* Vp = Up + batch * N * N;
    for (int64_t b = 0; b < batch; b++)
      for (int64_t i = 0; i < M; i++)
        for (int64_t j = 0; j < N; j++) At[b * M * N + j * M + i] = A[b * M * N + i * N + j];
    const int rc = nd4o_svd_dc(batch, N, M, At, Up, sv, Vp);
    if (rc == 0)
      for (int64_t b = 0; b < batch; b++) {
        for (int64_t i = 0; i < M; i++)
          for (int64_t j = 0; j < N; j++) U[b * M * N + i * N + j] = Vp[b * N * M + j * M + i];
        for (int64_t i = 0; i < N; i++)
          for (int64_t j = 0; j < N; j++) V[b * N * N + i * N + j] = Up[b * N * N + j * N + i];
      }
    free(At);
    return rc;
  }
  const int64_t w = (1 + M > N) ? 1 + M : N;
  const size_t nF = (size_t)(M * (M + 2) + M * 2 + (M + 1) * (M + 1) + (M + 1) * w);
  double* F = (double*)calloc(nF, sizeof(double));
  int32_t* I = (int32_t*)calloc((size_t)(M * 3 + 3), sizeof(int32_t));
  if (!F || !I) { free(F); free(I); return -1; }
  int rc = 0;
  for (int64_t b = 0; b < batch && rc == 0; b++) {
    double* u = U + b * M * M; double* v = V + b * M * N;
    for (int64_t e = 0; e < M * M; e++) u[e] = 0.0;
    memcpy(v, A + b * M * N, sizeof(double) * (size_t)(M * N));
    rc = svd_dc1(M, N, u, sv + b * M, v, I, F);
  }
  free(F); free(I);
  return rc;
}
