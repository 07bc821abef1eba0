/* N-API shim: the thin Node.js <-> C-ABI binding of include/nd4hip.h (north_star: "the Node.js host
 * keeps the nd.la API surface and calls into a thin N-API C-ABI addon").
 *
 * Plain C, built without node-gyp:  gcc -shared -fPIC -I/usr/include/node napi_shim.c -ldl
 * libnd4hip.so is dlopen'ed from the directory above this addon, so the addon itself has no HIP
 * link-time dependency. Every export takes TypedArrays (host memory, exactly what NDArray.data is in
 * the reference, src/nd_array.js:135-147) and calls the HOST-pointer entry points, which do
 * H2D -> kernels -> D2H synchronously: same blocking semantics as the reference's JS functions.
 * Errors become JS exceptions carrying nd4hip_last_error(). There is no CPU fallback here.
 */
#define _GNU_SOURCE
#define NAPI_VERSION 6
#include <node_api.h>
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "nd4hip.h"

static void* g_lib = NULL;
static nd4hip_handle* g_handle = NULL;

static int  (*p_device_count)(void);
static int  (*p_create)(nd4hip_handle**, int);
static void (*p_destroy)(nd4hip_handle*);
static const char* (*p_last_error)(void);
static const char* (*p_version)(void);
static int (*p_dgemm)(nd4hip_handle*, int64_t, int64_t, int64_t, int64_t, const double*, int64_t, const double*, int64_t, double*);
static int (*p_dgetrf)(nd4hip_handle*, int64_t, int64_t, const double*, double*, int32_t*);
static int (*p_dgeqrf)(nd4hip_handle*, int64_t, int64_t, int64_t, const double*, double*, double*);
static int (*p_dgesvdj)(nd4hip_handle*, int64_t, int64_t, int64_t, const double*, double*, double*, double*, int*, double*);
static int (*p_dqrls)(nd4hip_handle*, int64_t, int64_t, int64_t, int64_t, int64_t, const double*, int64_t, const double*, int64_t, const double*, int64_t, double*);
static int (*p_dsvdls)(nd4hip_handle*, int64_t, int64_t, int64_t, int64_t, int64_t, const double*, int64_t, const double*, int64_t, const double*, int64_t,
                       const double*, int64_t, double*);
static int (*p_dgetrs)(nd4hip_handle*, int64_t, int64_t, int64_t, const double*, int64_t, const int32_t*, int64_t, const double*, int64_t, double*);
static int (*p_dtrsm)(nd4hip_handle*, int, int, int64_t, int64_t, int64_t, const double*, int64_t, const double*, int64_t, double*);

static char g_load_error[512] = "";

static int load_library(void) {
  if (g_lib) return 0;
  Dl_info info;
  char path[4096];
  const char* env = getenv("ND4HIP_LIBRARY");
  if (env && *env) {
    snprintf(path, sizeof path, "%s", env);
  } else if (dladdr((void*)&load_library, &info) && info.dli_fname) {
    snprintf(path, sizeof path, "%s", info.dli_fname);
    char* slash = strrchr(path, '/');                   /* .../nd4js_amd/js/nd4hip_napi.node */
    if (slash) *slash = 0;
    slash = strrchr(path, '/');                         /* .../nd4js_amd/js */
    if (slash) *slash = 0;
    strncat(path, "/libnd4hip.so", sizeof path - strlen(path) - 1);
  } else {
    snprintf(path, sizeof path, "libnd4hip.so");
  }
  g_lib = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
  if (!g_lib) { snprintf(g_load_error, sizeof g_load_error, "cannot load %s: %s (no CPU fallback)", path, dlerror()); return -1; }
#define SYM(var, name) do { *(void**)(&var) = dlsym(g_lib, name); \
    if (!var) { snprintf(g_load_error, sizeof g_load_error, "symbol %s missing in %s", name, path); return -1; } } while (0)
  SYM(p_device_count, "nd4hip_device_count");
  SYM(p_create, "nd4hip_create");
  SYM(p_destroy, "nd4hip_destroy");
  SYM(p_last_error, "nd4hip_last_error");
  SYM(p_version, "nd4hip_version");
  SYM(p_dgemm, "nd4hip_dgemm_batched");
  SYM(p_dgetrf, "nd4hip_dgetrf_batched");
  SYM(p_dgeqrf, "nd4hip_dgeqrf_q_batched");
  SYM(p_dgesvdj, "nd4hip_dgesvdj_batched");
  SYM(p_dgetrs, "nd4hip_dgetrs_batched");
  SYM(p_dqrls, "nd4hip_dqrls_batched");
  SYM(p_dsvdls, "nd4hip_dsvdls_batched");
  SYM(p_dtrsm, "nd4hip_dtrsm_batched");
#undef SYM
  return 0;
}

#define THROW(env, msg) do { napi_throw_error((env), "ND4HIP", (msg)); return NULL; } while (0)

/* lazily created, idempotent (the reference has no init step: SURVEY.md §3.5) */
static int ensure_handle(napi_env env) {
  if (load_library() != 0) { napi_throw_error(env, "ND4HIP", g_load_error); return -1; }
  if (g_handle) return 0;
  int dev = 0;
  const char* e = getenv("ND4HIP_DEVICE");
  if (e && *e) dev = atoi(e);
  if (p_create(&g_handle, dev) != 0) { napi_throw_error(env, "ND4HIP", p_last_error()); g_handle = NULL; return -1; }
  return 0;
}

static int get_i64(napi_env env, napi_value v, int64_t* out) {
  double d;
  if (napi_get_value_double(env, v, &d) != napi_ok) { napi_throw_type_error(env, "ND4HIP", "expected a number"); return -1; }
  *out = (int64_t)d;
  return 0;
}
/* typed array -> pointer + element count; `want` = napi_float64_array or napi_int32_array */
static int get_ta(napi_env env, napi_value v, napi_typedarray_type want, void** data, size_t* len) {
  bool is_ta = false;
  napi_is_typedarray(env, v, &is_ta);
  if (!is_ta) { napi_throw_type_error(env, "ND4HIP", "expected a TypedArray"); return -1; }
  napi_typedarray_type type; napi_value buf; size_t off;
  if (napi_get_typedarray_info(env, v, &type, len, data, &buf, &off) != napi_ok || type != want) {
    napi_throw_type_error(env, "ND4HIP", want == napi_float64_array ? "expected a Float64Array" : "expected an Int32Array");
    return -1;
  }
  return 0;
}
#define NEED(cond, msg) do { if (!(cond)) { napi_throw_range_error(env, "ND4HIP", msg); return NULL; } } while (0)

static napi_value js_device_count(napi_env env, napi_callback_info info) {
  (void)info;
  napi_value r;
  if (load_library() != 0) THROW(env, g_load_error);
  napi_create_int32(env, p_device_count(), &r);
  return r;
}
static napi_value js_version(napi_env env, napi_callback_info info) {
  (void)info;
  napi_value r;
  if (load_library() != 0) THROW(env, g_load_error);
  napi_create_string_utf8(env, p_version(), NAPI_AUTO_LENGTH, &r);
  return r;
}

/* dgemm_batched(batch, I, K, J, A, strideA, B, strideB, C) */
static napi_value js_dgemm(napi_env env, napi_callback_info info) {
  size_t argc = 9; napi_value a[9];
  napi_get_cb_info(env, info, &argc, a, NULL, NULL);
  NEED(argc == 9, "dgemm_batched: 9 arguments expected");
  int64_t batch, I, K, J, sA, sB; void *A, *B, *C; size_t nA, nB, nC;
  if (get_i64(env, a[0], &batch) || get_i64(env, a[1], &I) || get_i64(env, a[2], &K) || get_i64(env, a[3], &J) ||
      get_ta(env, a[4], napi_float64_array, &A, &nA) || get_i64(env, a[5], &sA) ||
      get_ta(env, a[6], napi_float64_array, &B, &nB) || get_i64(env, a[7], &sB) ||
      get_ta(env, a[8], napi_float64_array, &C, &nC)) return NULL;
  NEED(batch >= 0 && I >= 0 && K >= 0 && J >= 0 && sA >= 0 && sB >= 0, "dgemm_batched: negative extent");
  NEED(batch == 0 || ((size_t)((batch - 1) * sA + I * K) <= nA && (size_t)((batch - 1) * sB + K * J) <= nB && (size_t)(batch * I * J) <= nC),
       "dgemm_batched: buffer too small");
  if (ensure_handle(env)) return NULL;
  if (p_dgemm(g_handle, batch, I, K, J, (const double*)A, sA, (const double*)B, sB, (double*)C) != 0) THROW(env, p_last_error());
  return NULL;
}
/* dgetrf_batched(batch, N, A, LU, P) */
static napi_value js_dgetrf(napi_env env, napi_callback_info info) {
  size_t argc = 5; napi_value a[5];
  napi_get_cb_info(env, info, &argc, a, NULL, NULL);
  NEED(argc == 5, "dgetrf_batched: 5 arguments expected");
  int64_t batch, N; void *A, *LU, *P; size_t nA, nLU, nP;
  if (get_i64(env, a[0], &batch) || get_i64(env, a[1], &N) || get_ta(env, a[2], napi_float64_array, &A, &nA) ||
      get_ta(env, a[3], napi_float64_array, &LU, &nLU) || get_ta(env, a[4], napi_int32_array, &P, &nP)) return NULL;
  NEED(batch >= 0 && N >= 0 && (size_t)(batch * N * N) <= nA && (size_t)(batch * N * N) <= nLU && (size_t)(batch * N) <= nP,
       "dgetrf_batched: buffer too small");
  if (ensure_handle(env)) return NULL;
  if (p_dgetrf(g_handle, batch, N, (const double*)A, (double*)LU, (int32_t*)P) != 0) THROW(env, p_last_error());
  return NULL;
}
/* dgeqrf_q_batched(batch, M, N, A, Q, R) */
static napi_value js_dgeqrf(napi_env env, napi_callback_info info) {
  size_t argc = 6; napi_value a[6];
  napi_get_cb_info(env, info, &argc, a, NULL, NULL);
  NEED(argc == 6, "dgeqrf_q_batched: 6 arguments expected");
  int64_t batch, M, N; void *A, *Q, *R; size_t nA, nQ, nR;
  if (get_i64(env, a[0], &batch) || get_i64(env, a[1], &M) || get_i64(env, a[2], &N) ||
      get_ta(env, a[3], napi_float64_array, &A, &nA) || get_ta(env, a[4], napi_float64_array, &Q, &nQ) ||
      get_ta(env, a[5], napi_float64_array, &R, &nR)) return NULL;
  int64_t L = M < N ? M : N;
  NEED(batch >= 0 && M >= 0 && N >= 0 && (size_t)(batch * M * N) <= nA && (size_t)(batch * M * L) <= nQ && (size_t)(batch * L * N) <= nR,
       "dgeqrf_q_batched: buffer too small");
  if (ensure_handle(env)) return NULL;
  if (p_dgeqrf(g_handle, batch, M, N, (const double*)A, (double*)Q, (double*)R) != 0) THROW(env, p_last_error());
  return NULL;
}
/* dgesvdj_batched(batch, M, N, A, U, sv, V) -> {sweeps, offnorm} */
static napi_value js_dgesvdj(napi_env env, napi_callback_info info) {
  size_t argc = 7; napi_value a[7];
  napi_get_cb_info(env, info, &argc, a, NULL, NULL);
  NEED(argc == 7, "dgesvdj_batched: 7 arguments expected");
  int64_t batch, M, N; void *A, *U, *S, *V; size_t nA, nU, nS, nV;
  if (get_i64(env, a[0], &batch) || get_i64(env, a[1], &M) || get_i64(env, a[2], &N) ||
      get_ta(env, a[3], napi_float64_array, &A, &nA) || get_ta(env, a[4], napi_float64_array, &U, &nU) ||
      get_ta(env, a[5], napi_float64_array, &S, &nS) || get_ta(env, a[6], napi_float64_array, &V, &nV)) return NULL;
  int64_t L = M < N ? M : N;
  NEED(batch >= 0 && M >= 0 && N >= 0 && (size_t)(batch * M * N) <= nA && (size_t)(batch * M * L) <= nU &&
       (size_t)(batch * L) <= nS && (size_t)(batch * L * N) <= nV, "dgesvdj_batched: buffer too small");
  if (ensure_handle(env)) return NULL;
  int sweeps = 0; double off = 0.0;
  if (p_dgesvdj(g_handle, batch, M, N, (const double*)A, (double*)U, (double*)S, (double*)V, &sweeps, &off) != 0) THROW(env, p_last_error());
  napi_value r, v;
  napi_create_object(env, &r);
  napi_create_int32(env, sweeps, &v); napi_set_named_property(env, r, "sweeps", v);
  napi_create_double(env, off, &v); napi_set_named_property(env, r, "offnorm", v);
  return r;
}

/* dgetrs_batched(batch, N, J, LU, strideLU, P, strideP, Y, strideY, X)   (lu_solve, lu.js:84-177) */
static napi_value js_dgetrs(napi_env env, napi_callback_info info) {
  size_t argc = 10; napi_value a[10];
  napi_get_cb_info(env, info, &argc, a, NULL, NULL);
  NEED(argc == 10, "dgetrs_batched: 10 arguments expected");
  int64_t batch, N, J, sLU, sP, sY; void *LU, *P, *Y, *X; size_t nLU, nP, nY, nX;
  if (get_i64(env, a[0], &batch) || get_i64(env, a[1], &N) || get_i64(env, a[2], &J) ||
      get_ta(env, a[3], napi_float64_array, &LU, &nLU) || get_i64(env, a[4], &sLU) ||
      get_ta(env, a[5], napi_int32_array, &P, &nP) || get_i64(env, a[6], &sP) ||
      get_ta(env, a[7], napi_float64_array, &Y, &nY) || get_i64(env, a[8], &sY) ||
      get_ta(env, a[9], napi_float64_array, &X, &nX)) return NULL;
  NEED(batch >= 0 && N >= 0 && J >= 0 && sLU >= 0 && sP >= 0 && sY >= 0, "dgetrs_batched: negative extent");
  NEED(batch == 0 || ((size_t)((batch - 1) * sLU + N * N) <= nLU && (size_t)((batch - 1) * sP + N) <= nP &&
                      (size_t)((batch - 1) * sY + N * J) <= nY && (size_t)(batch * N * J) <= nX), "dgetrs_batched: buffer too small");
  if (ensure_handle(env)) return NULL;
  if (p_dgetrs(g_handle, batch, N, J, (const double*)LU, sLU, (const int32_t*)P, sP, (const double*)Y, sY, (double*)X) != 0) THROW(env, p_last_error());
  return NULL;
}
/* dtrsm_batched(upper, unit_diag, batch, M, J, T, strideT, Y, strideY, X)   (tril_solve / triu_solve, tri.js:155-290) */
static napi_value js_dtrsm(napi_env env, napi_callback_info info) {
  size_t argc = 10; napi_value a[10];
  napi_get_cb_info(env, info, &argc, a, NULL, NULL);
  NEED(argc == 10, "dtrsm_batched: 10 arguments expected");
  int64_t upper, unit, batch, M, J, sT, sY; void *T, *Y, *X; size_t nT, nY, nX;
  if (get_i64(env, a[0], &upper) || get_i64(env, a[1], &unit) || get_i64(env, a[2], &batch) || get_i64(env, a[3], &M) || get_i64(env, a[4], &J) ||
      get_ta(env, a[5], napi_float64_array, &T, &nT) || get_i64(env, a[6], &sT) ||
      get_ta(env, a[7], napi_float64_array, &Y, &nY) || get_i64(env, a[8], &sY) ||
      get_ta(env, a[9], napi_float64_array, &X, &nX)) return NULL;
  NEED(batch >= 0 && M >= 0 && J >= 0 && sT >= 0 && sY >= 0, "dtrsm_batched: negative extent");
  NEED(batch == 0 || ((size_t)((batch - 1) * sT + M * M) <= nT && (size_t)((batch - 1) * sY + M * J) <= nY && (size_t)(batch * M * J) <= nX),
       "dtrsm_batched: buffer too small");
  if (ensure_handle(env)) return NULL;
  if (p_dtrsm(g_handle, (int)upper, (int)unit, batch, M, J, (const double*)T, sT, (const double*)Y, sY, (double*)X) != 0) THROW(env, p_last_error());
  return NULL;
}

/* dqrls_batched(batch, N, M, I, J, Q, strideQ, R, strideR, Y, strideY, X)   (qr_lstsq, qr.js:186-273) */
static napi_value js_dqrls(napi_env env, napi_callback_info info) {
  size_t argc = 12; napi_value a[12];
  napi_get_cb_info(env, info, &argc, a, NULL, NULL);
  NEED(argc == 12, "dqrls_batched: 12 arguments expected");
  int64_t batch, N, M, I, J, sQ, sR, sY; void *Q, *R, *Y, *X; size_t nQ, nR, nY, nX;
  if (get_i64(env, a[0], &batch) || get_i64(env, a[1], &N) || get_i64(env, a[2], &M) || get_i64(env, a[3], &I) || get_i64(env, a[4], &J) ||
      get_ta(env, a[5], napi_float64_array, &Q, &nQ) || get_i64(env, a[6], &sQ) ||
      get_ta(env, a[7], napi_float64_array, &R, &nR) || get_i64(env, a[8], &sR) ||
      get_ta(env, a[9], napi_float64_array, &Y, &nY) || get_i64(env, a[10], &sY) ||
      get_ta(env, a[11], napi_float64_array, &X, &nX)) return NULL;
  NEED(batch >= 0 && N >= 0 && M >= 0 && I >= 0 && J >= 0 && sQ >= 0 && sR >= 0 && sY >= 0, "dqrls_batched: negative extent");
  NEED(batch == 0 || ((size_t)((batch - 1) * sQ + N * M) <= nQ && (size_t)((batch - 1) * sR + M * I) <= nR &&
                      (size_t)((batch - 1) * sY + N * J) <= nY && (size_t)(batch * I * J) <= nX), "dqrls_batched: buffer too small");
  if (ensure_handle(env)) return NULL;
  if (p_dqrls(g_handle, batch, N, M, I, J, (const double*)Q, sQ, (const double*)R, sR, (const double*)Y, sY, (double*)X) != 0) THROW(env, p_last_error());
  return NULL;
}
/* dsvdls_batched(batch, N, M, I, J, U, strideU, sv, strideSv, V, strideV, Y, strideY, X)   (svd_lstsq, svd.js:100-228) */
static napi_value js_dsvdls(napi_env env, napi_callback_info info) {
  size_t argc = 14; napi_value a[14];
  napi_get_cb_info(env, info, &argc, a, NULL, NULL);
  NEED(argc == 14, "dsvdls_batched: 14 arguments expected");
  int64_t batch, N, M, I, J, sU, sS, sV, sY; void *U, *S, *V, *Y, *X; size_t nU, nS, nV, nY, nX;
  if (get_i64(env, a[0], &batch) || get_i64(env, a[1], &N) || get_i64(env, a[2], &M) || get_i64(env, a[3], &I) || get_i64(env, a[4], &J) ||
      get_ta(env, a[5], napi_float64_array, &U, &nU) || get_i64(env, a[6], &sU) ||
      get_ta(env, a[7], napi_float64_array, &S, &nS) || get_i64(env, a[8], &sS) ||
      get_ta(env, a[9], napi_float64_array, &V, &nV) || get_i64(env, a[10], &sV) ||
      get_ta(env, a[11], napi_float64_array, &Y, &nY) || get_i64(env, a[12], &sY) ||
      get_ta(env, a[13], napi_float64_array, &X, &nX)) return NULL;
  NEED(batch >= 0 && N >= 0 && M >= 0 && I >= 0 && J >= 0 && sU >= 0 && sS >= 0 && sV >= 0 && sY >= 0, "dsvdls_batched: negative extent");
  NEED(batch == 0 || ((size_t)((batch - 1) * sU + N * M) <= nU && (size_t)((batch - 1) * sS + M) <= nS &&
                      (size_t)((batch - 1) * sV + M * I) <= nV && (size_t)((batch - 1) * sY + N * J) <= nY && (size_t)(batch * I * J) <= nX),
       "dsvdls_batched: buffer too small");
  if (ensure_handle(env)) return NULL;
  if (p_dsvdls(g_handle, batch, N, M, I, J, (const double*)U, sU, (const double*)S, sS, (const double*)V, sV, (const double*)Y, sY, (double*)X) != 0)
    THROW(env, p_last_error());
  return NULL;
}

static void cleanup(void* arg) {
  (void)arg;
  if (g_handle && p_destroy) { p_destroy(g_handle); g_handle = NULL; }
}

static napi_value init(napi_env env, napi_value exports) {
  napi_property_descriptor props[] = {
    {"device_count", NULL, js_device_count, NULL, NULL, NULL, napi_default, NULL},
    {"version", NULL, js_version, NULL, NULL, NULL, napi_default, NULL},
    {"dgemm_batched", NULL, js_dgemm, NULL, NULL, NULL, napi_default, NULL},
    {"dgetrf_batched", NULL, js_dgetrf, NULL, NULL, NULL, napi_default, NULL},
    {"dgeqrf_q_batched", NULL, js_dgeqrf, NULL, NULL, NULL, napi_default, NULL},
    {"dgesvdj_batched", NULL, js_dgesvdj, NULL, NULL, NULL, napi_default, NULL},
    {"dgetrs_batched", NULL, js_dgetrs, NULL, NULL, NULL, napi_default, NULL},
    {"dqrls_batched", NULL, js_dqrls, NULL, NULL, NULL, napi_default, NULL},
    {"dsvdls_batched", NULL, js_dsvdls, NULL, NULL, NULL, napi_default, NULL},
    {"dtrsm_batched", NULL, js_dtrsm, NULL, NULL, NULL, napi_default, NULL},
  };
  napi_define_properties(env, exports, sizeof props / sizeof props[0], props);
  napi_add_env_cleanup_hook(env, cleanup, NULL);
  return exports;
}
NAPI_MODULE(NODE_GYP_MODULE_NAME, init)
