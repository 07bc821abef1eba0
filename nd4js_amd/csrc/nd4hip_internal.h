// Internal declarations shared by the libnd4hip.so translation units (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdarg>
#include <vector>
#include "../../include/nd4hip.h"

struct Nd4WsBlock { char* p; size_t size, used; };

struct Nd4Stage { void* p; size_t bytes; bool in_use; };   // cached device staging block of the host-pointer entry points

struct nd4hip_handle {
  int device = 0;
  hipStream_t own_stream = nullptr;   // created by nd4hip_create
  hipStream_t stream = nullptr;       // the stream work is enqueued on (own or caller's)
  std::vector<Nd4WsBlock> ws;         // device workspace arena (bump allocation, LIFO release)
  std::vector<Nd4Stage> stage;        // staging blocks kept between host-pointer calls (hipMalloc/hipFree cost ~100 us each)
  size_t stage_bytes = 0;
  void* pinned = nullptr;             // small pinned host buffer for scalar read-backs
  size_t pinned_bytes = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  hipEvent_t ev_order = nullptr;      // orders the workspace arena across a change of stream (nd4hip_set_stream)
  hipStream_t copy_stream = nullptr;  // H2D / D2H of the host-pointer entry points (overlaps the kernels on `stream`)
  hipStream_t aux_stream = nullptr;   // second chain of the block-Jacobi sweeps (svd_block.hip), beside the first on `stream`
  hipEvent_t ev_aux_a = nullptr, ev_aux_b = nullptr;
  hipEvent_t ev_in[2] = {nullptr, nullptr}, ev_chunk[2] = {nullptr, nullptr};   // per staging set: inputs landed / kernels done
  std::vector<nd4hip_handle*> peers;  // further devices of a multi-device handle (nd4hip_create_multi); owned
  int* xstat = nullptr;               // host-coherent status word the kernels raise when an in-kernel exchange times out (ND4HIP_ERR_XCHG)
  int num_cu = 256;
  unsigned ws_generation = 0;         // bumped whenever the idle arena is dropped and rebuilt (Nd4WsScope)
  int svd_sweeps = 0; unsigned long long svd_rotations = 0; double svd_offnorm = 0.0;   // audit of the last SVD call
  bool prof_on = false, prof_valid = false; int prof_depth = 0;                          // nd4hip_profile_enable / _last
  hipEvent_t ev_p0 = nullptr, ev_p1 = nullptr; double prof_flops = 0.0, prof_bytes = 0.0; char prof_op[32] = "";
};

// Makes h->device the calling thread's current HIP device for the duration of an entry point and restores the previous
// one: workspace hipMalloc, kernel launches and event records all act on the current device, and a caller such as torch may
// have another device current (tensor on cuda:1 while cuda:0 is current).
struct Nd4DeviceGuard {
  int prev = -1; bool switched = false;
  explicit Nd4DeviceGuard(const nd4hip_handle* h) {
    if (hipGetDevice(&prev) == hipSuccess && prev != h->device) switched = hipSetDevice(h->device) == hipSuccess;
  }
  ~Nd4DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
  Nd4DeviceGuard(const Nd4DeviceGuard&) = delete;
  Nd4DeviceGuard& operator=(const Nd4DeviceGuard&) = delete;
};

// brackets the kernels of one entry point with the handle's profile events (outermost entry point only: QR inside SVD does not count)
struct Nd4Prof {
  nd4hip_handle* h; bool active;
  Nd4Prof(nd4hip_handle* hh, const char* op, double flops, double bytes);
  ~Nd4Prof();
  Nd4Prof(const Nd4Prof&) = delete;
  Nd4Prof& operator=(const Nd4Prof&) = delete;
};

void nd4_set_error(const char* fmt, ...);
// after a synchronisation: ND4HIP_ERR_XCHG (and the word cleared) if a kernel raised the handle's exchange status word
int  nd4_xchg_check(nd4hip_handle* h, const char* where);
// tests only: ND4HIP_TEST_DROP_PUBLISH=<panel> (read per call) makes one workgroup of that panel skip one publication
int  nd4_test_drop_panel();
int  nd4_hip_fail(hipError_t e, const char* what, const char* file, int line);

#define ND4_HIP(expr)                                                           \
  do { hipError_t _e = (expr);                                                  \
       if (_e != hipSuccess) return nd4_hip_fail(_e, #expr, __FILE__, __LINE__); } while (0)
#define ND4_CHECK_ARG(cond, ...)                                                \
  do { if (!(cond)) { nd4_set_error(__VA_ARGS__); return ND4HIP_ERR_ARG; } } while (0)
#define ND4_TRY(expr) do { int _rc = (expr); if (_rc != 0) return _rc; } while (0)

// Workspace arena. Kernels are stream-ordered, so a region released at the end of one call can be
// handed out to the next call on the same stream. Nested users (SVD -> QR -> GEMM) each open a
// Nd4WsScope; allocations never move or free blocks that are in use.
int nd4_ws_alloc(nd4hip_handle* h, size_t bytes, void** out);      // 256-byte aligned
struct Nd4WsScope {
  nd4hip_handle* h; size_t nblocks; size_t used_last; unsigned generation;
  explicit Nd4WsScope(nd4hip_handle* hh);
  ~Nd4WsScope();
};
int nd4_pinned(nd4hip_handle* h, size_t bytes, void** out);

// ---- internal launchers (device pointers, enqueue on h->stream) --------------------------------
// C = alpha*op(A)*op(B) + beta*C, batched over gridDim.y with element strides sA/sB/sC
int nd4_gemm(nd4hip_handle* h, bool transA, bool transB, int64_t M, int64_t N, int64_t K,
             double alpha, const double* A, int64_t lda, int64_t sA,
             const double* B, int64_t ldb, int64_t sB,
             double beta, double* C, int64_t ldc, int64_t sC, int64_t batch);

int nd4_getrf(nd4hip_handle* h, int64_t batch, int64_t N, const double* A, double* LU, int32_t* P);
int nd4_getrf_nopivot(nd4hip_handle* h, int64_t batch, int64_t N, const double* A, double* LU, int32_t* P);
int nd4_trsm(nd4hip_handle* h, bool upper, bool unit, int64_t batch, int64_t M, int64_t J, const double* T, int64_t sT, double* X);
int nd4_syrk_lower(nd4hip_handle* h, int64_t N, int64_t K, double alpha, const double* A, int64_t lda, int64_t sA,
                   double beta, double* C, int64_t ldc, int64_t sC, int64_t batch);
int nd4_gemm_nt_lower(nd4hip_handle* h, int64_t N, int64_t K, double alpha, const double* A, int64_t lda, int64_t sA,
                      const double* B, int64_t ldb, int64_t sB, double beta, double* C, int64_t ldc, int64_t sC, int64_t batch);
int nd4_trsm_ld(nd4hip_handle* h, bool upper, bool unit, int64_t batch, int64_t M, int64_t J, const double* T, int64_t ldT, int64_t sT,
                double* X, int64_t sX);
int nd4_trsm_t(nd4hip_handle* h, int64_t batch, int64_t M, int64_t J, const double* T, int64_t ldT, int64_t sT, double* X, int64_t sX);
int nd4_trsm_t_ex(nd4hip_handle* h, bool unit, int64_t batch, int64_t M, int64_t J, const double* T, int64_t ldT, int64_t sT, double* X, int64_t sX);
int nd4_ldltrf(nd4hip_handle* h, int64_t batch, int64_t N, const double* S, double* LD);
int nd4_ldltrs(nd4hip_handle* h, int64_t batch, int64_t N, int64_t J, const double* LD, int64_t sLD, const double* Y, int64_t sY, double* X);
int nd4_gebrd(nd4hip_handle* h, int64_t batch, int64_t M, int64_t N, const double* A, double* U, double* B, double* V);
int nd4_gehrd(nd4hip_handle* h, int64_t batch, int64_t N, const double* A, double* U, double* H);
int nd4_potrf(nd4hip_handle* h, int64_t batch, int64_t N, const double* S, double* L, int* flags);
int nd4_potrs(nd4hip_handle* h, int64_t batch, int64_t N, int64_t J, const double* L, int64_t sL, const double* Y, int64_t sY, double* X);
int nd4_qrls(nd4hip_handle* h, int64_t batch, int64_t N, int64_t M, int64_t I, int64_t J, const double* Q, int64_t sQ,
             const double* R, int64_t sR, const double* Y, int64_t sY, double* X);
int nd4_svdls(nd4hip_handle* h, int64_t batch, int64_t N, int64_t M, int64_t I, int64_t J, const double* U, int64_t sU,
              const double* sv, int64_t sSv, const double* V, int64_t sV, const double* Y, int64_t sY, double* X);
int nd4_getrs(nd4hip_handle* h, int64_t batch, int64_t N, int64_t J, const double* LU, int64_t sLU, const int32_t* P, int64_t sP,
              const double* Y, int64_t sY, double* X);
int nd4_geqrf_q(nd4hip_handle* h, int64_t batch, int64_t M, int64_t N, const double* A, double* Q, double* R);
int nd4_geqr2_panel(nd4hip_handle* h, int batch, int M, double* A, double* V, double* T);
int nd4_givens_signs(nd4hip_handle* h, int batch, int M, int L, int ncols, bool lu_rule, double* Q, long ldq, long sQ,
                     double* R, long ldr, long sR, const double* taus, long sTau, int* flips);
int nd4_wy_form(nd4hip_handle* h, int M, int n, const double* V, const double* Tdiag, int bs, double* Q, int Lq);
int nd4_geqrf_q_ex(nd4hip_handle* h, int64_t batch, int64_t M, int64_t N, const double* A, double* Q, double* R, bool full);
int nd4_gesvdj(nd4hip_handle* h, int64_t batch, int64_t M, int64_t N, const double* A,
               double* U, double* sv, double* V, int* sweeps_out, double* offnorm_out);

// small utility kernels (nd4hip_core.hip)
int nd4_copy_matrix(nd4hip_handle* h, int64_t rows, int64_t cols, const double* src, int64_t lds,
                    double* dst, int64_t ldd, int64_t batch, int64_t ssrc, int64_t sdst);
int nd4_transpose(nd4hip_handle* h, int64_t rows, int64_t cols, const double* src, int64_t lds,
                  double* dst, int64_t ldd, int64_t batch, int64_t ssrc, int64_t sdst);
int nd4_set_identity(nd4hip_handle* h, int64_t rows, int64_t cols, double* dst, int64_t ldd,
                     int64_t batch, int64_t sdst);
