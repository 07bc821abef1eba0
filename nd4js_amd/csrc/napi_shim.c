/* N-API shim: the thin Node.js <-> C-ABI binding of include/nd4hip.h (north_star: "the Node.js host
 * keeps the nd.la API surface and calls into a thin N-API C-ABI addon").
 *
 * Plain C, built without node-gyp:  gcc -shared -fPIC -I/usr/include/node napi_shim.c -ldl
 * libnd4hip.so is dlopen'ed from the directory above this addon, so the addon itself has no HIP
 * link-time dependency. Every array operand of a compute export is EITHER
 *   - a TypedArray (host memory, exactly what NDArray.data is in the reference, src/nd_array.js:135-147):
 *     the HOST-pointer entry points run H2D -> kernels -> D2H synchronously, the blocking semantics of
 *     the reference's JS functions; OR
 *   - a device view {b: <buffer from dev_alloc>, o: <element offset>} (device-resident
 *     NDArray): the `_dev` entry points run on the handle's stream and nothing crosses PCIe.
 * All operands of one call must live on the same side. Errors become JS exceptions carrying
 * nd4hip_last_error(). There is no CPU fallback here.
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

typedef int (*gemm_fn)(nd4hip_handle*, int64_t, int64_t, int64_t, int64_t, const double*, int64_t, const double*, int64_t, double*);
typedef int (*getrf_fn)(nd4hip_handle*, int64_t, int64_t, const double*, double*, int32_t*);
typedef int (*geqrf_fn)(nd4hip_handle*, int64_t, int64_t, int64_t, const double*, double*, double*);
typedef int (*gesvdj_fn)(nd4hip_handle*, int64_t, int64_t, int64_t, const double*, double*, double*, double*, int*, double*);
typedef int (*qrls_fn)(nd4hip_handle*, int64_t, int64_t, int64_t, int64_t, int64_t, const double*, int64_t, const double*, int64_t, const double*, int64_t, double*);
typedef int (*svdls_fn)(nd4hip_handle*, int64_t, int64_t, int64_t, int64_t, int64_t, const double*, int64_t, const double*, int64_t, const double*, int64_t,
                        const double*, int64_t, double*);
typedef int (*getrs_fn)(nd4hip_handle*, int64_t, int64_t, int64_t, const double*, int64_t, const int32_t*, int64_t, const double*, int64_t, double*);
typedef int (*trsm_fn)(nd4hip_handle*, int, int, int64_t, int64_t, int64_t, const double*, int64_t, const double*, int64_t, double*);

static int  (*p_device_count)(void);
static int  (*p_create_multi)(nd4hip_handle**, const int*, int);
static int  (*p_device_list)(nd4hip_handle*, int*, int);
static int  (*p_create)(nd4hip_handle**, int);
static void (*p_destroy)(nd4hip_handle*);
static const char* (*p_last_error)(void);
static const char* (*p_version)(void);
/* [0] = host-pointer entry point, [1] = its _dev twin */
static gemm_fn p_dgemm[2];
static getrf_fn p_dgetrf[2];
static geqrf_fn p_dgeqrf[2];
static geqrf_fn p_dgeqrf_full[2];
static int (*p_dgebrd[2])(nd4hip_handle*, int64_t, int64_t, int64_t, const double*, double*, double*, double*);
static int (*p_dgehrd[2])(nd4hip_handle*, int64_t, int64_t, const double*, double*, double*);
static int (*p_dgeqrf_qty[2])(nd4hip_handle*, int64_t, int64_t, int64_t, int64_t, double*, double*);
static gesvdj_fn p_dgesvdj[2];
static int (*p_svd_info)(nd4hip_handle*, int*, unsigned long long*, double*);
static qrls_fn p_dqrls[2];
static svdls_fn p_dsvdls[2];
static getrs_fn p_dgetrs[2];
static int (*p_dpotrf[2])(nd4hip_handle*, int64_t, int64_t, const double*, double*);
static int (*p_dldltrf[2])(nd4hip_handle*, int64_t, int64_t, const double*, double*);
static int (*p_dldltrs[2])(nd4hip_handle*, int64_t, int64_t, int64_t, const double*, int64_t, const double*, int64_t, double*);
static int (*p_dpotrs[2])(nd4hip_handle*, int64_t, int64_t, int64_t, const double*, int64_t, const double*, int64_t, double*);
static trsm_fn p_dtrsm[2];
static int (*p_malloc)(nd4hip_handle*, size_t, void**);
static int (*p_free)(nd4hip_handle*, void*);
static int (*p_h2d)(nd4hip_handle*, void*, const void*, size_t);
static int (*p_d2h)(nd4hip_handle*, void*, const void*, size_t);
static int (*p_sync)(nd4hip_handle*);
static int (*p_prof_enable)(nd4hip_handle*, int);
static int (*p_prof_last)(nd4hip_handle*, nd4hip_prof*, int);

static char g_load_error[4608] = "";

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
  SYM(p_create_multi, "nd4hip_create_multi");
  SYM(p_device_list, "nd4hip_device_list");
  SYM(p_create, "nd4hip_create");
  SYM(p_destroy, "nd4hip_destroy");
  SYM(p_last_error, "nd4hip_last_error");
  SYM(p_version, "nd4hip_version");
#define SYM2(var, name) SYM(var[0], name); SYM(var[1], name "_dev")
  SYM2(p_dgemm, "nd4hip_dgemm_batched");
  SYM2(p_dgetrf, "nd4hip_dgetrf_batched");
  SYM2(p_dgeqrf, "nd4hip_dgeqrf_q_batched");
  SYM2(p_dgeqrf_full, "nd4hip_dgeqrf_full_batched");
  SYM2(p_dgeqrf_qty, "nd4hip_dgeqrf_qty_batched");
  SYM2(p_dgehrd, "nd4hip_dgehrd_batched");
  SYM2(p_dgebrd, "nd4hip_dgebrd_batched");
  SYM2(p_dgesvdj, "nd4hip_dgesvdj_batched");
  SYM2(p_dgetrs, "nd4hip_dgetrs_batched");
  SYM2(p_dpotrf, "nd4hip_dpotrf_batched");
  SYM2(p_dpotrs, "nd4hip_dpotrs_batched");
  SYM2(p_dldltrf, "nd4hip_dldltrf_batched");
  SYM2(p_dldltrs, "nd4hip_dldltrs_batched");
  SYM2(p_dqrls, "nd4hip_dqrls_batched");
  SYM2(p_dsvdls, "nd4hip_dsvdls_batched");
  SYM2(p_dtrsm, "nd4hip_dtrsm_batched");
#undef SYM2
  SYM(p_svd_info, "nd4hip_dgesvdj_last_info");
  SYM(p_malloc, "nd4hip_malloc");
  SYM(p_free, "nd4hip_free");
  SYM(p_h2d, "nd4hip_memcpy_h2d");
  SYM(p_d2h, "nd4hip_memcpy_d2h");
  SYM(p_sync, "nd4hip_synchronize");
  SYM(p_prof_enable, "nd4hip_profile_enable");
  SYM(p_prof_last, "nd4hip_profile_last");
#undef SYM
  return 0;
}

#define THROW(env, msg) do { napi_throw_error((env), "ND4HIP", (msg)); return NULL; } while (0)

/* lazily created, idempotent (the reference has no init step: SURVEY.md §3.5) */
static int ensure_handle(napi_env env) {
  if (load_library() != 0) { napi_throw_error(env, "ND4HIP", g_load_error); return -1; }
  if (g_handle) return 0;
  /* ND4HIP_DEVICES = "all" | "0,1,2,...": one handle over several GPUs; batched calls on host arrays are then sharded along the
     leading batch axis (include/nd4hip.h, nd4hip_create_multi). Default: the single device ND4HIP_DEVICE (0). */
  const char* list = getenv("ND4HIP_DEVICES");
  if (list && *list) {
    int ids[64], n = 0;
    if (strcmp(list, "all") == 0) { const int cnt = p_device_count(); for (; n < cnt && n < 64; n++) ids[n] = n; }
    else for (const char* q = list; *q && n < 64;) { ids[n++] = atoi(q); while (*q && *q != ',') q++; if (*q == ',') q++; }
    if (n == 0) { napi_throw_error(env, "ND4HIP", "nd4hip: no HIP device available (ND4HIP_DEVICES)"); return -1; }
    if (p_create_multi(&g_handle, ids, n) != 0) { napi_throw_error(env, "ND4HIP", p_last_error()); g_handle = NULL; return -1; }
    return 0;
  }
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

/* ---- device buffers: an external with a finalizer; JS passes views {b: buffer, o: element offset} ---- */
#define DEVBUF_MAGIC 0x6e64346869706466ull
typedef struct { uint64_t magic; void* p; size_t bytes; } devbuf;

static void devbuf_release(devbuf* b) {
  if (b->p && g_handle && p_free) p_free(g_handle, b->p);     /* after the env cleanup hook the handle (and its memory) is gone */
  b->p = NULL;
}
static void devbuf_finalize(napi_env env, void* data, void* hint) {
  (void)hint;
  devbuf* b = (devbuf*)data;
  if (b->p) { int64_t adj; napi_adjust_external_memory(env, -(int64_t)b->bytes, &adj); }
  devbuf_release(b);
  free(b);
}
static devbuf* get_devbuf(napi_env env, napi_value v) {
  napi_valuetype t;
  void* data = NULL;
  if (napi_typeof(env, v, &t) != napi_ok || t != napi_external || napi_get_value_external(env, v, &data) != napi_ok ||
      !data || ((devbuf*)data)->magic != DEVBUF_MAGIC) {
    napi_throw_type_error(env, "ND4HIP", "expected a device buffer from dev_alloc()");
    return NULL;
  }
  return (devbuf*)data;
}

/* one array operand: host TypedArray, or device view {b, o}; len = elements available from p */
typedef struct { void* p; size_t len; int dev; } opnd;
static int get_op(napi_env env, napi_value v, napi_typedarray_type want, opnd* out) {
  bool is_ta = false;
  napi_is_typedarray(env, v, &is_ta);
  if (is_ta) { out->dev = 0; return get_ta(env, v, want, &out->p, &out->len); }
  napi_valuetype t;
  napi_value vb, vo;
  if (napi_typeof(env, v, &t) != napi_ok || t != napi_object || napi_get_named_property(env, v, "b", &vb) != napi_ok ||
      napi_get_named_property(env, v, "o", &vo) != napi_ok) {
    napi_throw_type_error(env, "ND4HIP", "expected a TypedArray or a device view {b, o}");
    return -1;
  }
  devbuf* b = get_devbuf(env, vb);
  if (!b) return -1;
  if (!b->p) { napi_throw_error(env, "ND4HIP", "device buffer was already freed"); return -1; }
  int64_t off;
  if (get_i64(env, vo, &off)) return -1;
  const size_t esz = want == napi_float64_array ? 8 : 4, total = b->bytes / esz;
  if (off < 0 || (size_t)off > total) { napi_throw_range_error(env, "ND4HIP", "device view offset out of range"); return -1; }
  out->p = (char*)b->p + (size_t)off * esz;
  out->len = total - (size_t)off;
  out->dev = 1;
  return 0;
}
#define SAME_SIDE(cond, name) do { if (!(cond)) { napi_throw_type_error(env, "ND4HIP", name ": operands must be all host TypedArrays or all device views"); return NULL; } } while (0)

#define NEED(cond, msg) do { if (!(cond)) { napi_throw_range_error(env, "ND4HIP", msg); return NULL; } } while (0)

static napi_value js_device_count(napi_env env, napi_callback_info info) {
  (void)info;
  napi_value r;
  if (load_library() != 0) THROW(env, g_load_error);
  napi_create_int32(env, p_device_count(), &r);
  return r;
}
/* devices() -> [ids] behind the (lazily created) handle */
static napi_value js_devices(napi_env env, napi_callback_info info) {
  (void)info;
  if (ensure_handle(env)) return NULL;
  int ids[64];
  const int n = p_device_list(g_handle, ids, 64);
  napi_value arr, v;
  napi_create_array_with_length(env, (size_t)n, &arr);
  for (int i = 0; i < n && i < 64; i++) { napi_create_int32(env, ids[i], &v); napi_set_element(env, arr, (uint32_t)i, v); }
  return arr;
}
static napi_value js_version(napi_env env, napi_callback_info info) {
  (void)info;
  napi_value r;
  if (load_library() != 0) THROW(env, g_load_error);
  napi_create_string_utf8(env, p_version(), NAPI_AUTO_LENGTH, &r);
  return r;
}

#define ARGS(n, name) size_t argc = (n); napi_value a[(n)]; napi_get_cb_info(env, info, &argc, a, NULL, NULL); \
  NEED(argc == (n), name ": " #n " arguments expected")
#define F64(i, o) get_op(env, a[i], napi_float64_array, &(o))
#define I32(i, o) get_op(env, a[i], napi_int32_array, &(o))
#define FAIL_IF(rc) do { if ((rc) != 0) THROW(env, p_last_error()); } while (0)

/* dgemm_batched(batch, I, K, J, A, strideA, B, strideB, C) */
static napi_value js_dgemm(napi_env env, napi_callback_info info) {
  ARGS(9, "dgemm_batched");
  int64_t batch, I, K, J, sA, sB; opnd A, B, C;
  if (get_i64(env, a[0], &batch) || get_i64(env, a[1], &I) || get_i64(env, a[2], &K) || get_i64(env, a[3], &J) ||
      F64(4, A) || get_i64(env, a[5], &sA) || F64(6, B) || get_i64(env, a[7], &sB) || F64(8, C)) return NULL;
  NEED(batch >= 0 && I >= 0 && K >= 0 && J >= 0 && sA >= 0 && sB >= 0, "dgemm_batched: negative extent");
  NEED(batch == 0 || ((size_t)((batch - 1) * sA + I * K) <= A.len && (size_t)((batch - 1) * sB + K * J) <= B.len && (size_t)(batch * I * J) <= C.len),
       "dgemm_batched: buffer too small");
  SAME_SIDE(A.dev == B.dev && B.dev == C.dev, "dgemm_batched");
  if (ensure_handle(env)) return NULL;
  FAIL_IF(p_dgemm[A.dev](g_handle, batch, I, K, J, (const double*)A.p, sA, (const double*)B.p, sB, (double*)C.p));
  return NULL;
}
/* dgetrf_batched(batch, N, A, LU, P) */
static napi_value js_dgetrf(napi_env env, napi_callback_info info) {
  ARGS(5, "dgetrf_batched");
  int64_t batch, N; opnd A, LU, P;
  if (get_i64(env, a[0], &batch) || get_i64(env, a[1], &N) || F64(2, A) || F64(3, LU) || I32(4, P)) return NULL;
  NEED(batch >= 0 && N >= 0 && (size_t)(batch * N * N) <= A.len && (size_t)(batch * N * N) <= LU.len && (size_t)(batch * N) <= P.len,
       "dgetrf_batched: buffer too small");
  SAME_SIDE(A.dev == LU.dev && LU.dev == P.dev, "dgetrf_batched");
  if (ensure_handle(env)) return NULL;
  FAIL_IF(p_dgetrf[A.dev](g_handle, batch, N, (const double*)A.p, (double*)LU.p, (int32_t*)P.p));
  return NULL;
}
/* dgeqrf_q_batched(batch, M, N, A, Q, R) */
static napi_value js_dgeqrf(napi_env env, napi_callback_info info) {
  ARGS(6, "dgeqrf_q_batched");
  int64_t batch, M, N; opnd A, Q, R;
  if (get_i64(env, a[0], &batch) || get_i64(env, a[1], &M) || get_i64(env, a[2], &N) || F64(3, A) || F64(4, Q) || F64(5, R)) return NULL;
  int64_t L = M < N ? M : N;
  NEED(batch >= 0 && M >= 0 && N >= 0 && (size_t)(batch * M * N) <= A.len && (size_t)(batch * M * L) <= Q.len && (size_t)(batch * L * N) <= R.len,
       "dgeqrf_q_batched: buffer too small");
  SAME_SIDE(A.dev == Q.dev && Q.dev == R.dev, "dgeqrf_q_batched");
  if (ensure_handle(env)) return NULL;
  FAIL_IF(p_dgeqrf[A.dev](g_handle, batch, M, N, (const double*)A.p, (double*)Q.p, (double*)R.p));
  return NULL;
}
/* dgeqrf_full_batched(batch, M, N, A, Q, R)   (qr_decomp_full, qr.js:27-77: Q is M x M, R is M x N) */
static napi_value js_dgeqrf_full(napi_env env, napi_callback_info info) {
  ARGS(6, "dgeqrf_full_batched");
  int64_t batch, M, N; opnd A, Q, R;
  if (get_i64(env, a[0], &batch) || get_i64(env, a[1], &M) || get_i64(env, a[2], &N) || F64(3, A) || F64(4, Q) || F64(5, R)) return NULL;
  NEED(batch >= 0 && M >= 0 && N >= 0 && (size_t)(batch * M * N) <= A.len && (size_t)(batch * M * M) <= Q.len && (size_t)(batch * M * N) <= R.len,
       "dgeqrf_full_batched: buffer too small");
  SAME_SIDE(A.dev == Q.dev && Q.dev == R.dev, "dgeqrf_full_batched");
  if (ensure_handle(env)) return NULL;
  FAIL_IF(p_dgeqrf_full[A.dev](g_handle, batch, M, N, (const double*)A.p, (double*)Q.p, (double*)R.p));
  return NULL;
}
/* dgeqrf_qty_batched(batch, M, N, L, A, Y)   (_qr_decomp_inplace, qr.js:146-183: A <- R, Y <- Q^T Y in place) */
static napi_value js_dgeqrf_qty(napi_env env, napi_callback_info info) {
  ARGS(6, "dgeqrf_qty_batched");
  int64_t batch, M, N, L; opnd A, Y;
  if (get_i64(env, a[0], &batch) || get_i64(env, a[1], &M) || get_i64(env, a[2], &N) || get_i64(env, a[3], &L) || F64(4, A) || F64(5, Y)) return NULL;
  NEED(batch >= 0 && M >= 0 && N >= 0 && L >= 0 && (size_t)(batch * M * N) <= A.len && (size_t)(batch * M * L) <= Y.len,
       "dgeqrf_qty_batched: buffer too small");
  SAME_SIDE(A.dev == Y.dev, "dgeqrf_qty_batched");
  if (ensure_handle(env)) return NULL;
  FAIL_IF(p_dgeqrf_qty[A.dev](g_handle, batch, M, N, L, (double*)A.p, (double*)Y.p));
  return NULL;
}
/* dgebrd_batched(batch, M, N, A, U, B, V)   (bidiag_decomp, bidiag.js:245-319) */
static napi_value js_dgebrd(napi_env env, napi_callback_info info) {
  ARGS(7, "dgebrd_batched");
  int64_t batch, M, N; opnd A, U, B, V;
  if (get_i64(env, a[0], &batch) || get_i64(env, a[1], &M) || get_i64(env, a[2], &N) || F64(3, A) || F64(4, U) || F64(5, B) || F64(6, V)) return NULL;
  NEED(batch >= 0 && M >= 0 && N >= 0, "dgebrd_batched: negative extent");
  const int64_t K = M < N ? M : N, J = M >= N ? K : K + 1;
  NEED((size_t)(batch * M * N) <= A.len && (size_t)(batch * M * K) <= U.len && (size_t)(batch * K * J) <= B.len && (size_t)(batch * J * N) <= V.len,
       "dgebrd_batched: buffer too small");
  SAME_SIDE(A.dev == U.dev && U.dev == B.dev && B.dev == V.dev, "dgebrd_batched");
  if (ensure_handle(env)) return NULL;
  FAIL_IF(p_dgebrd[A.dev](g_handle, batch, M, N, (const double*)A.p, (double*)U.p, (double*)B.p, (double*)V.p));
  return NULL;
}
/* dgehrd_batched(batch, N, A, U, H)   (hessenberg_decomp, hessenberg.js:89-115) */
static napi_value js_dgehrd(napi_env env, napi_callback_info info) {
  ARGS(5, "dgehrd_batched");
  int64_t batch, N; opnd A, U, H;
  if (get_i64(env, a[0], &batch) || get_i64(env, a[1], &N) || F64(2, A) || F64(3, U) || F64(4, H)) return NULL;
  NEED(batch >= 0 && N >= 0 && (size_t)(batch * N * N) <= A.len && (size_t)(batch * N * N) <= U.len && (size_t)(batch * N * N) <= H.len,
       "dgehrd_batched: buffer too small");
  SAME_SIDE(A.dev == U.dev && U.dev == H.dev, "dgehrd_batched");
  if (ensure_handle(env)) return NULL;
  FAIL_IF(p_dgehrd[A.dev](g_handle, batch, N, (const double*)A.p, (double*)U.p, (double*)H.p));
  return NULL;
}
/* dgesvdj_batched(batch, M, N, A, U, sv, V) -> {sweeps, offnorm, rotations} */
static napi_value js_dgesvdj(napi_env env, napi_callback_info info) {
  ARGS(7, "dgesvdj_batched");
  int64_t batch, M, N; opnd A, U, S, V;
  if (get_i64(env, a[0], &batch) || get_i64(env, a[1], &M) || get_i64(env, a[2], &N) || F64(3, A) || F64(4, U) || F64(5, S) || F64(6, V)) return NULL;
  int64_t L = M < N ? M : N;
  NEED(batch >= 0 && M >= 0 && N >= 0 && (size_t)(batch * M * N) <= A.len && (size_t)(batch * M * L) <= U.len &&
       (size_t)(batch * L) <= S.len && (size_t)(batch * L * N) <= V.len, "dgesvdj_batched: buffer too small");
  SAME_SIDE(A.dev == U.dev && U.dev == S.dev && S.dev == V.dev, "dgesvdj_batched");
  if (ensure_handle(env)) return NULL;
  int sweeps = 0; double off = 0.0;
  FAIL_IF(p_dgesvdj[A.dev](g_handle, batch, M, N, (const double*)A.p, (double*)U.p, (double*)S.p, (double*)V.p, &sweeps, &off));
  napi_value r, v;
  napi_create_object(env, &r);
  napi_create_int32(env, sweeps, &v); napi_set_named_property(env, r, "sweeps", v);
  napi_create_double(env, off, &v); napi_set_named_property(env, r, "offnorm", v);
  unsigned long long rot = 0;
  FAIL_IF(p_svd_info(g_handle, NULL, &rot, NULL));
  napi_create_double(env, (double)rot, &v); napi_set_named_property(env, r, "rotations", v);     /* exact up to 2^53 */
  return r;
}

/* dgetrs_batched(batch, N, J, LU, strideLU, P, strideP, Y, strideY, X)   (lu_solve, lu.js:84-177) */
static napi_value js_dgetrs(napi_env env, napi_callback_info info) {
  ARGS(10, "dgetrs_batched");
  int64_t batch, N, J, sLU, sP, sY; opnd LU, P, Y, X;
  if (get_i64(env, a[0], &batch) || get_i64(env, a[1], &N) || get_i64(env, a[2], &J) || F64(3, LU) || get_i64(env, a[4], &sLU) ||
      I32(5, P) || get_i64(env, a[6], &sP) || F64(7, Y) || get_i64(env, a[8], &sY) || F64(9, X)) return NULL;
  NEED(batch >= 0 && N >= 0 && J >= 0 && sLU >= 0 && sP >= 0 && sY >= 0, "dgetrs_batched: negative extent");
  NEED(batch == 0 || ((size_t)((batch - 1) * sLU + N * N) <= LU.len && (size_t)((batch - 1) * sP + N) <= P.len &&
                      (size_t)((batch - 1) * sY + N * J) <= Y.len && (size_t)(batch * N * J) <= X.len), "dgetrs_batched: buffer too small");
  SAME_SIDE(LU.dev == P.dev && P.dev == Y.dev && Y.dev == X.dev, "dgetrs_batched");
  if (ensure_handle(env)) return NULL;
  FAIL_IF(p_dgetrs[X.dev](g_handle, batch, N, J, (const double*)LU.p, sLU, (const int32_t*)P.p, sP, (const double*)Y.p, sY, (double*)X.p));
  return NULL;
}
/* dtrsm_batched(upper, unit_diag, batch, M, J, T, strideT, Y, strideY, X)   (tril_solve / triu_solve, tri.js:155-290) */
static napi_value js_dtrsm(napi_env env, napi_callback_info info) {
  ARGS(10, "dtrsm_batched");
  int64_t upper, unit, batch, M, J, sT, sY; opnd T, Y, X;
  if (get_i64(env, a[0], &upper) || get_i64(env, a[1], &unit) || get_i64(env, a[2], &batch) || get_i64(env, a[3], &M) || get_i64(env, a[4], &J) ||
      F64(5, T) || get_i64(env, a[6], &sT) || F64(7, Y) || get_i64(env, a[8], &sY) || F64(9, X)) return NULL;
  NEED(batch >= 0 && M >= 0 && J >= 0 && sT >= 0 && sY >= 0, "dtrsm_batched: negative extent");
  NEED(batch == 0 || ((size_t)((batch - 1) * sT + M * M) <= T.len && (size_t)((batch - 1) * sY + M * J) <= Y.len && (size_t)(batch * M * J) <= X.len),
       "dtrsm_batched: buffer too small");
  SAME_SIDE(T.dev == Y.dev && Y.dev == X.dev, "dtrsm_batched");
  if (ensure_handle(env)) return NULL;
  FAIL_IF(p_dtrsm[X.dev](g_handle, (int)upper, (int)unit, batch, M, J, (const double*)T.p, sT, (const double*)Y.p, sY, (double*)X.p));
  return NULL;
}
/* dqrls_batched(batch, N, M, I, J, Q, strideQ, R, strideR, Y, strideY, X)   (qr_lstsq, qr.js:186-273) */
static napi_value js_dqrls(napi_env env, napi_callback_info info) {
  ARGS(12, "dqrls_batched");
  int64_t batch, N, M, I, J, sQ, sR, sY; opnd Q, R, Y, X;
  if (get_i64(env, a[0], &batch) || get_i64(env, a[1], &N) || get_i64(env, a[2], &M) || get_i64(env, a[3], &I) || get_i64(env, a[4], &J) ||
      F64(5, Q) || get_i64(env, a[6], &sQ) || F64(7, R) || get_i64(env, a[8], &sR) || F64(9, Y) || get_i64(env, a[10], &sY) || F64(11, X)) return NULL;
  NEED(batch >= 0 && N >= 0 && M >= 0 && I >= 0 && J >= 0 && sQ >= 0 && sR >= 0 && sY >= 0, "dqrls_batched: negative extent");
  NEED(batch == 0 || ((size_t)((batch - 1) * sQ + N * M) <= Q.len && (size_t)((batch - 1) * sR + M * I) <= R.len &&
                      (size_t)((batch - 1) * sY + N * J) <= Y.len && (size_t)(batch * I * J) <= X.len), "dqrls_batched: buffer too small");
  SAME_SIDE(Q.dev == R.dev && R.dev == Y.dev && Y.dev == X.dev, "dqrls_batched");
  if (ensure_handle(env)) return NULL;
  FAIL_IF(p_dqrls[X.dev](g_handle, batch, N, M, I, J, (const double*)Q.p, sQ, (const double*)R.p, sR, (const double*)Y.p, sY, (double*)X.p));
  return NULL;
}
/* dsvdls_batched(batch, N, M, I, J, U, strideU, sv, strideSv, V, strideV, Y, strideY, X)   (svd_lstsq, svd.js:100-228) */
static napi_value js_dsvdls(napi_env env, napi_callback_info info) {
  ARGS(14, "dsvdls_batched");
  int64_t batch, N, M, I, J, sU, sS, sV, sY; opnd U, S, V, Y, X;
  if (get_i64(env, a[0], &batch) || get_i64(env, a[1], &N) || get_i64(env, a[2], &M) || get_i64(env, a[3], &I) || get_i64(env, a[4], &J) ||
      F64(5, U) || get_i64(env, a[6], &sU) || F64(7, S) || get_i64(env, a[8], &sS) || F64(9, V) || get_i64(env, a[10], &sV) ||
      F64(11, Y) || get_i64(env, a[12], &sY) || F64(13, X)) return NULL;
  NEED(batch >= 0 && N >= 0 && M >= 0 && I >= 0 && J >= 0 && sU >= 0 && sS >= 0 && sV >= 0 && sY >= 0, "dsvdls_batched: negative extent");
  NEED(batch == 0 || ((size_t)((batch - 1) * sU + N * M) <= U.len && (size_t)((batch - 1) * sS + M) <= S.len &&
                      (size_t)((batch - 1) * sV + M * I) <= V.len && (size_t)((batch - 1) * sY + N * J) <= Y.len && (size_t)(batch * I * J) <= X.len),
       "dsvdls_batched: buffer too small");
  SAME_SIDE(U.dev == S.dev && S.dev == V.dev && V.dev == Y.dev && Y.dev == X.dev, "dsvdls_batched");
  if (ensure_handle(env)) return NULL;
  FAIL_IF(p_dsvdls[X.dev](g_handle, batch, N, M, I, J, (const double*)U.p, sU, (const double*)S.p, sS, (const double*)V.p, sV,
                          (const double*)Y.p, sY, (double*)X.p));
  return NULL;
}

/* dpotrf_batched(batch, N, S, L)   (cholesky_decomp, cholesky.js:51-71) */
static napi_value js_dpotrf(napi_env env, napi_callback_info info) {
  ARGS(4, "dpotrf_batched");
  int64_t batch, N; opnd S, L;
  if (get_i64(env, a[0], &batch) || get_i64(env, a[1], &N) || F64(2, S) || F64(3, L)) return NULL;
  NEED(batch >= 0 && N >= 0 && (size_t)(batch * N * N) <= S.len && (size_t)(batch * N * N) <= L.len, "dpotrf_batched: buffer too small");
  SAME_SIDE(S.dev == L.dev, "dpotrf_batched");
  if (ensure_handle(env)) return NULL;
  FAIL_IF(p_dpotrf[S.dev](g_handle, batch, N, (const double*)S.p, (double*)L.p));
  return NULL;
}
/* dpotrs_batched(batch, N, J, L, strideL, Y, strideY, X)   (cholesky_solve, cholesky.js:74-150) */
static napi_value js_dpotrs(napi_env env, napi_callback_info info) {
  ARGS(8, "dpotrs_batched");
  int64_t batch, N, J, sL, sY; opnd L, Y, X;
  if (get_i64(env, a[0], &batch) || get_i64(env, a[1], &N) || get_i64(env, a[2], &J) || F64(3, L) || get_i64(env, a[4], &sL) ||
      F64(5, Y) || get_i64(env, a[6], &sY) || F64(7, X)) return NULL;
  NEED(batch >= 0 && N >= 0 && J >= 0 && sL >= 0 && sY >= 0, "dpotrs_batched: negative extent");
  NEED(batch == 0 || ((size_t)((batch - 1) * sL + N * N) <= L.len && (size_t)((batch - 1) * sY + N * J) <= Y.len && (size_t)(batch * N * J) <= X.len),
       "dpotrs_batched: buffer too small");
  SAME_SIDE(L.dev == Y.dev && Y.dev == X.dev, "dpotrs_batched");
  if (ensure_handle(env)) return NULL;
  FAIL_IF(p_dpotrs[X.dev](g_handle, batch, N, J, (const double*)L.p, sL, (const double*)Y.p, sY, (double*)X.p));
  return NULL;
}

/* dldltrf_batched(batch, N, S, LD)   (ldl_decomp, ldl.js:67-90) */
static napi_value js_dldltrf(napi_env env, napi_callback_info info) {
  ARGS(4, "dldltrf_batched");
  int64_t batch, N; opnd S, L;
  if (get_i64(env, a[0], &batch) || get_i64(env, a[1], &N) || F64(2, S) || F64(3, L)) return NULL;
  NEED(batch >= 0 && N >= 0 && (size_t)(batch * N * N) <= S.len && (size_t)(batch * N * N) <= L.len, "dldltrf_batched: buffer too small");
  SAME_SIDE(S.dev == L.dev, "dldltrf_batched");
  if (ensure_handle(env)) return NULL;
  FAIL_IF(p_dldltrf[S.dev](g_handle, batch, N, (const double*)S.p, (double*)L.p));
  return NULL;
}
/* dldltrs_batched(batch, N, J, LD, strideLD, Y, strideY, X)   (ldl_solve, ldl.js:133-201) */
static napi_value js_dldltrs(napi_env env, napi_callback_info info) {
  ARGS(8, "dldltrs_batched");
  int64_t batch, N, J, sL, sY; opnd L, Y, X;
  if (get_i64(env, a[0], &batch) || get_i64(env, a[1], &N) || get_i64(env, a[2], &J) || F64(3, L) || get_i64(env, a[4], &sL) ||
      F64(5, Y) || get_i64(env, a[6], &sY) || F64(7, X)) return NULL;
  NEED(batch >= 0 && N >= 0 && J >= 0 && sL >= 0 && sY >= 0, "dldltrs_batched: negative extent");
  NEED(batch == 0 || ((size_t)((batch - 1) * sL + N * N) <= L.len && (size_t)((batch - 1) * sY + N * J) <= Y.len && (size_t)(batch * N * J) <= X.len),
       "dldltrs_batched: buffer too small");
  SAME_SIDE(L.dev == Y.dev && Y.dev == X.dev, "dldltrs_batched");
  if (ensure_handle(env)) return NULL;
  FAIL_IF(p_dldltrs[X.dev](g_handle, batch, N, J, (const double*)L.p, sL, (const double*)Y.p, sY, (double*)X.p));
  return NULL;
}

/* ---- device memory ---- */
/* dev_alloc(bytes) -> buffer */
static napi_value js_dev_alloc(napi_env env, napi_callback_info info) {
  ARGS(1, "dev_alloc");
  int64_t bytes;
  if (get_i64(env, a[0], &bytes)) return NULL;
  NEED(bytes >= 0, "dev_alloc: negative size");
  if (ensure_handle(env)) return NULL;
  devbuf* b = (devbuf*)calloc(1, sizeof(devbuf));
  if (!b) THROW(env, "dev_alloc: out of host memory");
  b->magic = DEVBUF_MAGIC; b->bytes = (size_t)bytes;
  if (p_malloc(g_handle, b->bytes, &b->p) != 0) { free(b); THROW(env, p_last_error()); }
  napi_value r;
  if (napi_create_external(env, b, devbuf_finalize, NULL, &r) != napi_ok) { devbuf_release(b); free(b); THROW(env, "dev_alloc: napi_create_external failed"); }
  int64_t adj; napi_adjust_external_memory(env, (int64_t)b->bytes, &adj);     /* let the GC see the device bytes it keeps alive */
  return r;
}
/* dev_free(buffer): idempotent; the finalizer covers buffers that are simply dropped */
static napi_value js_dev_free(napi_env env, napi_callback_info info) {
  ARGS(1, "dev_free");
  devbuf* b = get_devbuf(env, a[0]);
  if (!b) return NULL;
  if (b->p) { int64_t adj; napi_adjust_external_memory(env, -(int64_t)b->bytes, &adj); devbuf_release(b); b->bytes = 0; }
  return NULL;
}
static int get_any_ta(napi_env env, napi_value v, void** data, size_t* bytes) {
  bool is_ta = false;
  napi_is_typedarray(env, v, &is_ta);
  napi_typedarray_type type; napi_value buf; size_t off, len;
  if (!is_ta || napi_get_typedarray_info(env, v, &type, &len, data, &buf, &off) != napi_ok ||
      (type != napi_float64_array && type != napi_int32_array)) {
    napi_throw_type_error(env, "ND4HIP", "expected a Float64Array or an Int32Array");
    return -1;
  }
  *bytes = len * (type == napi_float64_array ? 8 : 4);
  return 0;
}
/* dev_upload(buffer, byteOffset, typedArray): blocking H2D */
static napi_value js_dev_upload(napi_env env, napi_callback_info info) {
  ARGS(3, "dev_upload");
  devbuf* b = get_devbuf(env, a[0]);
  int64_t off; void* src; size_t bytes;
  if (!b || get_i64(env, a[1], &off) || get_any_ta(env, a[2], &src, &bytes)) return NULL;
  NEED(b->p && off >= 0 && (size_t)off + bytes <= b->bytes, "dev_upload: out of range (or freed buffer)");
  if (ensure_handle(env)) return NULL;
  FAIL_IF(p_h2d(g_handle, (char*)b->p + off, src, bytes));
  return NULL;
}
/* dev_download(typedArray, buffer, byteOffset): blocking D2H after everything queued on the handle's stream */
static napi_value js_dev_download(napi_env env, napi_callback_info info) {
  ARGS(3, "dev_download");
  devbuf* b = get_devbuf(env, a[1]);
  int64_t off; void* dst; size_t bytes;
  if (!b || get_i64(env, a[2], &off) || get_any_ta(env, a[0], &dst, &bytes)) return NULL;
  NEED(b->p && off >= 0 && (size_t)off + bytes <= b->bytes, "dev_download: out of range (or freed buffer)");
  if (ensure_handle(env)) return NULL;
  FAIL_IF(p_d2h(g_handle, dst, (const char*)b->p + off, bytes));
  return NULL;
}
static napi_value js_synchronize(napi_env env, napi_callback_info info) {
  (void)info;
  if (ensure_handle(env)) return NULL;
  FAIL_IF(p_sync(g_handle));
  return NULL;
}

static void cleanup(void* arg) {
  (void)arg;
  if (g_handle && p_destroy) { p_destroy(g_handle); g_handle = NULL; }
}

/* profile_enable(on) */
static napi_value js_profile_enable(napi_env env, napi_callback_info info) {
  size_t argc = 1; napi_value a[1]; bool on = true;
  napi_get_cb_info(env, info, &argc, a, NULL, NULL);
  if (argc >= 1) napi_get_value_bool(env, a[0], &on);
  if (ensure_handle(env)) return NULL;
  if (p_prof_enable(g_handle, on ? 1 : 0) != 0) THROW(env, p_last_error());
  return NULL;
}
/* profile_last() -> [{device, valid, op, kernel_ms, flops, bytes}] : one record per device of the handle (nd4hip_profile_last) */
static napi_value js_profile_last(napi_env env, napi_callback_info info) {
  (void)info;
  if (ensure_handle(env)) return NULL;
  nd4hip_prof rec[64];
  const int n = p_prof_last(g_handle, rec, 64);
  if (n < 0) THROW(env, p_last_error());
  napi_value arr, o, v;
  napi_create_array_with_length(env, (size_t)n, &arr);
  for (int i = 0; i < n && i < 64; i++) {
    napi_create_object(env, &o);
    napi_create_int32(env, rec[i].device, &v); napi_set_named_property(env, o, "device", v);
    napi_get_boolean(env, rec[i].valid != 0, &v); napi_set_named_property(env, o, "valid", v);
    napi_create_string_utf8(env, rec[i].op, NAPI_AUTO_LENGTH, &v); napi_set_named_property(env, o, "op", v);
    napi_create_double(env, rec[i].kernel_ms, &v); napi_set_named_property(env, o, "kernel_ms", v);
    napi_create_double(env, rec[i].flops, &v); napi_set_named_property(env, o, "flops", v);
    napi_create_double(env, rec[i].bytes, &v); napi_set_named_property(env, o, "bytes", v);
    napi_set_element(env, arr, (uint32_t)i, o);
  }
  return arr;
}

static napi_value init(napi_env env, napi_value exports) {
  napi_property_descriptor props[] = {
    {"device_count", NULL, js_device_count, NULL, NULL, NULL, napi_default, NULL},
    {"version", NULL, js_version, NULL, NULL, NULL, napi_default, NULL},
    {"devices", NULL, js_devices, NULL, NULL, NULL, napi_default, NULL},
    {"dgemm_batched", NULL, js_dgemm, NULL, NULL, NULL, napi_default, NULL},
    {"dgetrf_batched", NULL, js_dgetrf, NULL, NULL, NULL, napi_default, NULL},
    {"dgeqrf_q_batched", NULL, js_dgeqrf, NULL, NULL, NULL, napi_default, NULL},
    {"dgeqrf_full_batched", NULL, js_dgeqrf_full, NULL, NULL, NULL, napi_default, NULL},
    {"dgeqrf_qty_batched", NULL, js_dgeqrf_qty, NULL, NULL, NULL, napi_default, NULL},
    {"dgebrd_batched", NULL, js_dgebrd, NULL, NULL, NULL, napi_default, NULL},
    {"dgehrd_batched", NULL, js_dgehrd, NULL, NULL, NULL, napi_default, NULL},
    {"dgesvdj_batched", NULL, js_dgesvdj, NULL, NULL, NULL, napi_default, NULL},
    {"dgetrs_batched", NULL, js_dgetrs, NULL, NULL, NULL, napi_default, NULL},
    {"dqrls_batched", NULL, js_dqrls, NULL, NULL, NULL, napi_default, NULL},
    {"dsvdls_batched", NULL, js_dsvdls, NULL, NULL, NULL, napi_default, NULL},
    {"dtrsm_batched", NULL, js_dtrsm, NULL, NULL, NULL, napi_default, NULL},
    {"dpotrf_batched", NULL, js_dpotrf, NULL, NULL, NULL, napi_default, NULL},
    {"dpotrs_batched", NULL, js_dpotrs, NULL, NULL, NULL, napi_default, NULL},
    {"dldltrf_batched", NULL, js_dldltrf, NULL, NULL, NULL, napi_default, NULL},
    {"dldltrs_batched", NULL, js_dldltrs, NULL, NULL, NULL, napi_default, NULL},
    {"dev_alloc", NULL, js_dev_alloc, NULL, NULL, NULL, napi_default, NULL},
    {"dev_free", NULL, js_dev_free, NULL, NULL, NULL, napi_default, NULL},
    {"dev_upload", NULL, js_dev_upload, NULL, NULL, NULL, napi_default, NULL},
    {"dev_download", NULL, js_dev_download, NULL, NULL, NULL, napi_default, NULL},
    {"synchronize", NULL, js_synchronize, NULL, NULL, NULL, napi_default, NULL},
    {"profile_enable", NULL, js_profile_enable, NULL, NULL, NULL, napi_default, NULL},
    {"profile_last", NULL, js_profile_last, NULL, NULL, NULL, napi_default, NULL},
  };
  napi_define_properties(env, exports, sizeof props / sizeof props[0], props);
  napi_add_env_cleanup_hook(env, cleanup, NULL);
  return exports;
}
NAPI_MODULE(NODE_GYP_MODULE_NAME, init)
