// Host-pointer entry points of include/nd4hip.h (what the N-API shim binds: JS TypedArrays are host memory) on top of the
// *_dev forms, through ONE driver:
//   * the batch axis — the reference's loops over independent matrices (qr.js:43-49, lu.js:34-40, svd_dc.js:918-925,
//     matmul odometer matmul.js:44-70) — is cut into contiguous blocks, one per device of the handle (nd4hip_create_multi),
//     each block driven by its own host thread on its own device, streams and staging: no data-path collective, the only
//     "reductions" (max sweeps, off-norm, rotation count, error code) happen on the host;
//   * inside a block the batch is cut again into chunks that are pipelined: H2D of chunk k+1 and D2H of chunk k-1 run on a copy
//     stream while chunk k computes (two staging sets). Measured on this pool (tools/pcie_bench.hip): pageable host memory
//     moves at the pinned rate (56 GB/s) through hipMemcpyAsync, so the user's buffers are used as they are; the copies block
//     the calling host thread, the kernels do not, which is all the overlap needs;
//   * a single large matmul is pipelined over the ROWS of A and C (B is a broadcast operand).
// Error path: a failing chunk stops the block, the block's streams are synchronised BEFORE its staging returns to the cache.
#include "nd4hip_internal.h"
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <thread>
#include <vector>

namespace {

constexpr size_t D = sizeof(double);
constexpr size_t ND4_STAGE_CAP = size_t(12) << 30;     // cached staging per device before the cache is trimmed

// Device staging blocks are cached in the handle between calls: every host-pointer call synchronises before it returns, so a
// released block is immediately reusable, and hipMalloc + hipFree (~100 us each, several per call) used to be most of the
// latency of a small call. Best fit among the free blocks that are not more than 4x too large.
struct DevBuf {
  void* p = nullptr;
  nd4hip_handle* h = nullptr;
  int slot = -1;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { release(); }
  void release() {
    if (slot >= 0) h->stage[(size_t)slot].in_use = false;
    else if (p) (void)hipFree(p);
    p = nullptr; slot = -1;
  }
  int alloc(nd4hip_handle* hh, size_t bytes) {
    h = hh;
    if (bytes == 0) bytes = 8;
    int best = -1;
    for (size_t i = 0; i < h->stage.size(); i++) {
      const Nd4Stage& b = h->stage[i];
      if (!b.in_use && b.bytes >= bytes && b.bytes <= 4 * bytes + 4096 && (best < 0 || b.bytes < h->stage[(size_t)best].bytes)) best = (int)i;
    }
    if (best >= 0) { h->stage[(size_t)best].in_use = true; slot = best; p = h->stage[(size_t)best].p; return 0; }
    if (h->stage_bytes + bytes > ND4_STAGE_CAP || h->stage.size() >= 64) {
      bool any_used = false;
      for (const auto& b : h->stage) any_used = any_used || b.in_use;
      if (any_used) { ND4_HIP(hipMalloc(&p, bytes)); slot = -1; return 0; }     // over budget mid-call: an uncached block
      for (auto& b : h->stage) (void)hipFree(b.p);                              // between calls: start the cache afresh
      h->stage.clear(); h->stage_bytes = 0;
    }
    ND4_HIP(hipMalloc(&p, bytes));
    h->stage.push_back(Nd4Stage{p, bytes, true});
    h->stage_bytes += bytes;
    slot = (int)h->stage.size() - 1;
    return 0;
  }
};

// One operand of a batched host-pointer call: `item` elements per batch member, members `stride` elements apart on the host
// (0 = one block shared by all members: uploaded once per device). The device image keeps the same stride.
struct Operand {
  const void* src;      // host source, NULL = not an input
  void* dst;            // host destination, NULL = not an output (src == dst: in place)
  int64_t item;         // elements per batch member
  int64_t stride;       // elements between members (0 = broadcast)
  size_t es;            // bytes per element
};
inline Operand in_op(const void* p, int64_t item, int64_t stride, size_t es = D) { return Operand{p, nullptr, item, stride, es}; }
inline Operand out_op(void* p, int64_t item, size_t es = D) { return Operand{nullptr, p, item, item, es}; }
inline Operand inout_op(void* p, int64_t item, size_t es = D) { return Operand{p, p, item, item, es}; }

// run(hd, dev_index, nb, dptr): nb batch members whose operands start at dptr[i] on device handle hd
using ChunkFn = std::function<int(nd4hip_handle* hd, int dev, int64_t nb, void* const* dptr)>;

struct Plan {
  int64_t min_chunk = 1;                 // never cut a block into chunks smaller than this many members (efficiency of the kernels)
  size_t chunk_bytes = size_t(64) << 20; // target bytes moved per chunk
  int max_chunks = 8;
};

inline size_t span_bytes(const Operand& o, int64_t nb) { return (size_t)((o.stride ? (nb - 1) * o.stride : 0) + o.item) * o.es; }

// One device's contiguous block [lo, hi) of the batch: chunked, double-buffered.
int run_block(nd4hip_handle* h, int dev, int64_t lo, int64_t hi, const std::vector<Operand>& ops, const ChunkFn& fn, const Plan& plan) {
  Nd4DeviceGuard guard(h);
  const int64_t n = hi - lo;
  if (n <= 0) return 0;
  const size_t nops = ops.size();
  size_t per_item = 0;
  for (const auto& o : ops) if (o.stride) per_item += (size_t)o.item * o.es;
  int64_t nchunks = (int64_t)((per_item * (size_t)n + plan.chunk_bytes - 1) / plan.chunk_bytes);
  if (nchunks > plan.max_chunks) nchunks = plan.max_chunks;
  if (nchunks > n / plan.min_chunk) nchunks = n / plan.min_chunk;
  if (nchunks < 1) nchunks = 1;
  const int64_t per = (n + nchunks - 1) / nchunks;
  nchunks = (n + per - 1) / per;
  const int nsets = nchunks > 1 ? 2 : 1;

  std::vector<DevBuf> bufs(nops * 2);
  auto buf = [&](size_t i, int set) -> DevBuf& { return bufs[i * 2 + (size_t)(ops[i].stride ? set : 0)]; };
  int rc = 0;
  static const bool one_stream = [] { const char* e = getenv("ND4HIP_HOST_ONE_STREAM"); return e && *e && *e != '0'; }();   // A/B switch
  hipStream_t cs = (h->copy_stream && !one_stream) ? h->copy_stream : h->stream;       // copies; compute goes to h->stream
  auto fail = [&](int code) {
    // the block's copies / kernels may still be using the staging blocks: drain before the DevBufs go back to the cache
    (void)hipStreamSynchronize(cs); (void)hipStreamSynchronize(h->stream);
    return code;
  };
#define BLK_TRY(expr) do { int _rc = (expr); if (_rc != 0) return fail(_rc); } while (0)
#define BLK_HIP(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return fail(nd4_hip_fail(_e, #expr, __FILE__, __LINE__)); } while (0)
  for (size_t i = 0; i < nops; i++) {
    const Operand& o = ops[i];
    if (o.stride == 0) {
      BLK_TRY(buf(i, 0).alloc(h, span_bytes(o, 1)));
      if (o.src) BLK_HIP(hipMemcpyAsync(buf(i, 0).p, o.src, span_bytes(o, 1), hipMemcpyHostToDevice, cs));
    } else {
      for (int s = 0; s < nsets; s++) BLK_TRY(buf(i, s).alloc(h, span_bytes(o, per)));
    }
  }
  std::vector<void*> dptr(nops);
  auto upload = [&](int64_t c) -> int {
    const int64_t b0 = lo + c * per, nb = (b0 + per <= hi) ? per : hi - b0;
    const int set = (int)(c & 1) % nsets;
    // the set was last used by chunk c-2: its kernels must be done before the inputs are overwritten
    if (c >= 2) ND4_HIP(hipStreamWaitEvent(cs, h->ev_chunk[set], 0));
    for (size_t i = 0; i < nops; i++) {
      const Operand& o = ops[i];
      if (o.stride && o.src)
        ND4_HIP(hipMemcpyAsync(buf(i, set).p, static_cast<const char*>(o.src) + (size_t)(b0 * o.stride) * o.es, span_bytes(o, nb), hipMemcpyHostToDevice, cs));
    }
    ND4_HIP(hipEventRecord(h->ev_in[set], cs));
    return 0;
  };
  auto compute = [&](int64_t c) -> int {
    const int64_t b0 = lo + c * per, nb = (b0 + per <= hi) ? per : hi - b0;
    const int set = (int)(c & 1) % nsets;
    if (cs != h->stream) ND4_HIP(hipStreamWaitEvent(h->stream, h->ev_in[set], 0));
    for (size_t i = 0; i < nops; i++) dptr[i] = buf(i, set).p;
    ND4_TRY(fn(h, dev, nb, dptr.data()));
    ND4_HIP(hipEventRecord(h->ev_chunk[set], h->stream));
    return 0;
  };
  auto download = [&](int64_t c) -> int {
    const int64_t b0 = lo + c * per, nb = (b0 + per <= hi) ? per : hi - b0;
    const int set = (int)(c & 1) % nsets;
    if (cs != h->stream) ND4_HIP(hipStreamWaitEvent(cs, h->ev_chunk[set], 0));
    for (size_t i = 0; i < nops; i++) {
      const Operand& o = ops[i];
      if (o.dst)
        ND4_HIP(hipMemcpyAsync(static_cast<char*>(o.dst) + (size_t)(b0 * o.stride) * o.es, buf(i, set).p, span_bytes(o, nb), hipMemcpyDeviceToHost, cs));
    }
    return 0;
  };
  // software pipeline: while chunk c computes, the results of chunk c-1 come down and the inputs of chunk c+1 go up (in this
  // order on the copy stream: an in-place operand of set (c+1)&1 must be read out before it is overwritten)
  rc = upload(0);
  for (int64_t c = 0; c < nchunks && rc == 0; c++) {
    rc = compute(c);
    if (rc == 0 && c >= 1) rc = download(c - 1);
    if (rc == 0 && c + 1 < nchunks) rc = upload(c + 1);
  }
  if (rc == 0) rc = download(nchunks - 1);
  if (rc != 0) return fail(rc);
  BLK_HIP(hipStreamSynchronize(cs));
  BLK_HIP(hipStreamSynchronize(h->stream));
  BLK_TRY(nd4_xchg_check(h, "host-pointer entry point"));
#undef BLK_TRY
#undef BLK_HIP
  return 0;
}

// All devices of the handle: contiguous blocks of the batch (remainder to the low devices), one host thread per extra device.
int run_host(nd4hip_handle* h, int64_t batch, const std::vector<Operand>& ops, const ChunkFn& fn, const Plan& plan = Plan()) {
  if (batch <= 0) return 0;
  const int ndev = 1 + (int)h->peers.size();
  const int used = (int)(batch < ndev ? batch : ndev);
  if (used <= 1) return run_block(h, 0, 0, batch, ops, fn, plan);
  std::vector<int> rcs((size_t)used, 0);
  std::vector<std::string> errs((size_t)used);
  std::vector<std::thread> workers;
  const int64_t per = batch / used, rem = batch % used;
  auto edge = [&](int d) { return (int64_t)d * per + (d < rem ? d : rem); };
  for (int d = 1; d < used; d++) {
    workers.emplace_back([&, d] {
      rcs[(size_t)d] = run_block(h->peers[(size_t)d - 1], d, edge(d), edge(d + 1), ops, fn, plan);
      if (rcs[(size_t)d] != 0) errs[(size_t)d] = nd4hip_last_error();         // the error text is thread-local
    });
  }
  rcs[0] = run_block(h, 0, edge(0), edge(1), ops, fn, plan);
  for (auto& w : workers) w.join();
  for (int d = 0; d < used; d++)
    if (rcs[(size_t)d] != 0) { if (d > 0) nd4_set_error("%s (device %d of the handle)", errs[(size_t)d].c_str(), d); return rcs[(size_t)d]; }
  return 0;
}

inline double* P(void* const* d, int i) { return static_cast<double*>(d[i]); }

}  // namespace

// ------------------------------------------------------------------------------------ lifecycle: several devices behind one handle
extern "C" int nd4hip_create_multi(nd4hip_handle** out, const int* device_ids, int n_dev) {
  ND4_CHECK_ARG(out != nullptr, "nd4hip_create_multi: out is NULL");
  *out = nullptr;
  ND4_CHECK_ARG(n_dev >= 1 && n_dev <= 64 && device_ids != nullptr, "nd4hip_create_multi: need 1..64 device ids");
  // ND4HIP_TEST_ALLOW_DUP_DEVICES=1 (tests only): the same device may be listed several times, so that the per-device host threads,
  // the block partition, the merged SVD audit and the first-error-with-its-device hand-off run on a one-GPU box
  const char* dup = getenv("ND4HIP_TEST_ALLOW_DUP_DEVICES");
  if (!(dup && *dup && *dup != '0'))
    for (int i = 0; i < n_dev; i++)
      for (int j = 0; j < i; j++) ND4_CHECK_ARG(device_ids[i] != device_ids[j], "nd4hip_create_multi: device %d listed twice", device_ids[i]);
  nd4hip_handle* h = nullptr;
  ND4_TRY(nd4hip_create(&h, device_ids[0]));
  for (int i = 1; i < n_dev; i++) {
    nd4hip_handle* p = nullptr;
    const int rc = nd4hip_create(&p, device_ids[i]);
    if (rc != 0) { nd4hip_destroy(h); return rc; }
    h->peers.push_back(p);
  }
  *out = h;
  return 0;
}
extern "C" int nd4hip_device_list(nd4hip_handle* h, int* device_ids, int capacity) {
  ND4_CHECK_ARG(h != nullptr, "nd4hip_device_list: NULL handle");
  const int n = 1 + (int)h->peers.size();
  for (int i = 0; i < n && i < capacity && device_ids; i++) device_ids[i] = i == 0 ? h->device : h->peers[(size_t)i - 1]->device;
  return n;
}
// [lo, hi) of a batch of `batch` members that device `index` of an n_dev-device handle processes (exposed for tests / hosts)
extern "C" int nd4hip_partition(int64_t batch, int n_dev, int index, int64_t* lo, int64_t* hi) {
  ND4_CHECK_ARG(batch >= 0 && n_dev >= 1 && index >= 0 && index < n_dev && lo && hi, "nd4hip_partition: bad argument");
  const int used = (int)(batch < n_dev ? batch : n_dev);
  if (index >= used) { *lo = *hi = batch; return 0; }
  const int64_t per = batch / used, rem = batch % used;
  *lo = (int64_t)index * per + (index < rem ? index : rem);
  *hi = (int64_t)(index + 1) * per + (index + 1 < rem ? index + 1 : rem);
  return 0;
}

// ------------------------------------------------------------------------------------ matmul
extern "C" int nd4hip_dgemm_batched(nd4hip_handle* h, int64_t batch, int64_t I, int64_t K, int64_t J,
                                    const double* A, int64_t strideA, const double* B, int64_t strideB, double* C) {
  ND4_CHECK_ARG(h != nullptr, "nd4hip_dgemm_batched: NULL handle");
  ND4_CHECK_ARG(batch >= 0 && I >= 0 && K >= 0 && J >= 0, "nd4hip_dgemm_batched: negative extent");
  ND4_CHECK_ARG(strideA == 0 || strideA >= I * K, "nd4hip_dgemm_batched: strideA must be 0 or >= I*K");
  ND4_CHECK_ARG(strideB == 0 || strideB >= K * J, "nd4hip_dgemm_batched: strideB must be 0 or >= K*J");
  if (batch == 0 || I == 0 || J == 0) return 0;
  ND4_CHECK_ARG(C && (K == 0 || (A && B)), "nd4hip_dgemm_batched: NULL matrix pointer");
  if (batch == 1 && K > 0) {
    // one product: the "batch" of the driver is the rows of A and C (a row of C needs its row of A and all of B), so the
    // upload of the next row block and the download of the previous one overlap the MFMA kernel; one device (replicas only)
    Plan plan; plan.min_chunk = 512;
    std::vector<Operand> ops{in_op(A, K, K), in_op(B, K * J, 0), out_op(C, J)};
    ChunkFn fn = [=](nd4hip_handle* hd, int, int64_t rows, void* const* d) {
      return nd4hip_dgemm_batched_dev(hd, 1, rows, K, J, P(d, 0), 0, P(d, 1), 0, P(d, 2));
    };
    return run_block(h, 0, 0, I, ops, fn, plan);
  }
  std::vector<Operand> ops{in_op(A, I * K, strideA), in_op(B, K * J, strideB), out_op(C, I * J)};
  ChunkFn fn = [=](nd4hip_handle* hd, int, int64_t nb, void* const* d) {
    return nd4hip_dgemm_batched_dev(hd, nb, I, K, J, P(d, 0), nb > 1 ? strideA : 0, P(d, 1), nb > 1 ? strideB : 0, P(d, 2));
  };
  return run_host(h, batch, ops, fn);
}

// ------------------------------------------------------------------------------------ LU
extern "C" int nd4hip_dgetrf_batched(nd4hip_handle* h, int64_t batch, int64_t N, const double* A, double* LU, int32_t* Pv) {
  ND4_CHECK_ARG(h != nullptr, "nd4hip_dgetrf_batched: NULL handle");
  ND4_CHECK_ARG(batch >= 0 && N >= 0, "nd4hip_dgetrf_batched: negative extent");
  if (batch == 0 || N == 0) return 0;
  ND4_CHECK_ARG(A && LU && Pv, "nd4hip_dgetrf_batched: NULL pointer");
  std::vector<Operand> ops{in_op(A, N * N, N * N), out_op(LU, N * N), out_op(Pv, N, 4)};
  ChunkFn fn = [=](nd4hip_handle* hd, int, int64_t nb, void* const* d) {
    return nd4hip_dgetrf_batched_dev(hd, nb, N, P(d, 0), P(d, 1), static_cast<int32_t*>(d[2]));
  };
  return run_host(h, batch, ops, fn);
}

// ------------------------------------------------------------------------------------ solves
extern "C" int nd4hip_dgetrs_batched(nd4hip_handle* h, int64_t batch, int64_t N, int64_t J, const double* LU, int64_t strideLU,
                                     const int32_t* Pv, int64_t strideP, const double* Y, int64_t strideY, double* X) {
  ND4_CHECK_ARG(h != nullptr, "nd4hip_dgetrs_batched: NULL handle");
  ND4_CHECK_ARG(batch >= 0 && N >= 0 && J >= 0, "nd4hip_dgetrs_batched: negative extent");
  ND4_CHECK_ARG((strideLU == 0 || strideLU >= N * N) && (strideP == 0 || strideP >= N) && (strideY == 0 || strideY >= N * J),
                "nd4hip_dgetrs_batched: a stride must be 0 or at least the size of one operand");
  if (batch == 0 || N == 0 || J == 0) return 0;
  ND4_CHECK_ARG(LU && Pv && Y && X, "nd4hip_dgetrs_batched: NULL pointer");
  std::vector<Operand> ops{in_op(LU, N * N, strideLU), in_op(Pv, N, strideP, 4), in_op(Y, N * J, strideY), out_op(X, N * J)};
  ChunkFn fn = [=](nd4hip_handle* hd, int, int64_t nb, void* const* d) {
    return nd4hip_dgetrs_batched_dev(hd, nb, N, J, P(d, 0), strideLU, static_cast<const int32_t*>(d[1]), strideP, P(d, 2), strideY, P(d, 3));
  };
  return run_host(h, batch, ops, fn);
}
extern "C" int nd4hip_dtrsm_batched(nd4hip_handle* h, int upper, int unit_diag, int64_t batch, int64_t M, int64_t J,
                                    const double* T, int64_t strideT, const double* Y, int64_t strideY, double* X) {
  ND4_CHECK_ARG(h != nullptr, "nd4hip_dtrsm_batched: NULL handle");
  ND4_CHECK_ARG(batch >= 0 && M >= 0 && J >= 0, "nd4hip_dtrsm_batched: negative extent");
  ND4_CHECK_ARG((strideT == 0 || strideT >= M * M) && (strideY == 0 || strideY >= M * J),
                "nd4hip_dtrsm_batched: a stride must be 0 or at least the size of one operand");
  if (batch == 0 || M == 0 || J == 0) return 0;
  ND4_CHECK_ARG(T && Y && X, "nd4hip_dtrsm_batched: NULL pointer");
  std::vector<Operand> ops{in_op(T, M * M, strideT), in_op(Y, M * J, strideY), out_op(X, M * J)};
  ChunkFn fn = [=](nd4hip_handle* hd, int, int64_t nb, void* const* d) {
    return nd4hip_dtrsm_batched_dev(hd, upper, unit_diag, nb, M, J, P(d, 0), strideT, P(d, 1), strideY, P(d, 2));
  };
  return run_host(h, batch, ops, fn);
}

// ---- least squares from a factorisation: qr_lstsq (qr.js:186-273), svd_lstsq / svd_solve (svd.js:66-228)
extern "C" int nd4hip_dqrls_batched(nd4hip_handle* h, int64_t batch, int64_t N, int64_t M, int64_t I, int64_t J,
                                    const double* Q, int64_t strideQ, const double* R, int64_t strideR,
                                    const double* Y, int64_t strideY, double* X) {
  ND4_CHECK_ARG(h != nullptr, "nd4hip_dqrls_batched: NULL handle");
  ND4_CHECK_ARG(batch >= 0 && N >= 0 && M >= 0 && I >= 0 && J >= 0, "nd4hip_dqrls_batched: negative extent");
  ND4_CHECK_ARG(I <= N, "qr_lstsq(Q,R,y): Under-determined systems not supported. Use rrqr instead.");      // qr.js:209
  ND4_CHECK_ARG((strideQ == 0 || strideQ >= N * M) && (strideR == 0 || strideR >= M * I) && (strideY == 0 || strideY >= N * J),
                "nd4hip_dqrls_batched: a stride must be 0 or at least the size of one operand");
  if (batch == 0 || I == 0 || J == 0) return 0;
  ND4_CHECK_ARG(X != nullptr, "nd4hip_dqrls_batched: NULL pointer");
  if (N == 0 || M == 0) { memset(X, 0, D * (size_t)(batch * I * J)); return 0; }
  ND4_CHECK_ARG(Q && R && Y, "nd4hip_dqrls_batched: NULL pointer");
  std::vector<Operand> ops{in_op(Q, N * M, strideQ), in_op(R, M * I, strideR), in_op(Y, N * J, strideY), out_op(X, I * J)};
  ChunkFn fn = [=](nd4hip_handle* hd, int, int64_t nb, void* const* d) {
    return nd4hip_dqrls_batched_dev(hd, nb, N, M, I, J, P(d, 0), strideQ, P(d, 1), strideR, P(d, 2), strideY, P(d, 3));
  };
  return run_host(h, batch, ops, fn);
}
extern "C" int nd4hip_dsvdls_batched(nd4hip_handle* h, int64_t batch, int64_t N, int64_t M, int64_t I, int64_t J,
                                     const double* U, int64_t strideU, const double* sv, int64_t strideSv,
                                     const double* V, int64_t strideV, const double* Y, int64_t strideY, double* X) {
  ND4_CHECK_ARG(h != nullptr, "nd4hip_dsvdls_batched: NULL handle");
  ND4_CHECK_ARG(batch >= 0 && N >= 0 && M >= 0 && I >= 0 && J >= 0, "nd4hip_dsvdls_batched: negative extent");
  ND4_CHECK_ARG((strideU == 0 || strideU >= N * M) && (strideSv == 0 || strideSv >= M) && (strideV == 0 || strideV >= M * I) &&
                (strideY == 0 || strideY >= N * J), "nd4hip_dsvdls_batched: a stride must be 0 or at least the size of one operand");
  if (batch == 0 || I == 0 || J == 0) return 0;
  ND4_CHECK_ARG(X != nullptr, "nd4hip_dsvdls_batched: NULL pointer");
  if (N == 0 || M == 0) { memset(X, 0, D * (size_t)(batch * I * J)); return 0; }
  ND4_CHECK_ARG(U && sv && V && Y, "nd4hip_dsvdls_batched: NULL pointer");
  const size_t nS = (size_t)(strideSv ? (batch - 1) * strideSv + M : M);
  for (size_t i = 0; i < nS; i++) ND4_CHECK_ARG(std::isfinite(sv[i]), "svd_solve(): NaN or Infinity encountered.");   // svd.js:171-172
  std::vector<Operand> ops{in_op(U, N * M, strideU), in_op(sv, M, strideSv), in_op(V, M * I, strideV), in_op(Y, N * J, strideY), out_op(X, I * J)};
  ChunkFn fn = [=](nd4hip_handle* hd, int, int64_t nb, void* const* d) {
    return nd4hip_dsvdls_batched_dev(hd, nb, N, M, I, J, P(d, 0), strideU, P(d, 1), strideSv, P(d, 2), strideV, P(d, 3), strideY, P(d, 4));
  };
  return run_host(h, batch, ops, fn);
}

// ---- Cholesky / LDL^T
extern "C" int nd4hip_dpotrf_batched(nd4hip_handle* h, int64_t batch, int64_t N, const double* S, double* L) {
  ND4_CHECK_ARG(h != nullptr, "nd4hip_dpotrf_batched: NULL handle");
  ND4_CHECK_ARG(batch >= 0 && N >= 0, "nd4hip_dpotrf_batched: negative extent");
  if (batch == 0 || N == 0) return 0;
  ND4_CHECK_ARG(S && L, "nd4hip_dpotrf_batched: NULL pointer");
  std::vector<Operand> ops{in_op(S, N * N, N * N), out_op(L, N * N)};
  ChunkFn fn = [=](nd4hip_handle* hd, int, int64_t nb, void* const* d) { return nd4hip_dpotrf_batched_dev(hd, nb, N, P(d, 0), P(d, 1)); };
  return run_host(h, batch, ops, fn);
}
extern "C" int nd4hip_dpotrs_batched(nd4hip_handle* h, int64_t batch, int64_t N, int64_t J, const double* L, int64_t strideL,
                                     const double* Y, int64_t strideY, double* X) {
  ND4_CHECK_ARG(h != nullptr, "nd4hip_dpotrs_batched: NULL handle");
  ND4_CHECK_ARG(batch >= 0 && N >= 0 && J >= 0, "nd4hip_dpotrs_batched: negative extent");
  ND4_CHECK_ARG((strideL == 0 || strideL >= N * N) && (strideY == 0 || strideY >= N * J),
                "nd4hip_dpotrs_batched: a stride must be 0 or at least the size of one operand");
  if (batch == 0 || N == 0 || J == 0) return 0;
  ND4_CHECK_ARG(L && Y && X, "nd4hip_dpotrs_batched: NULL pointer");
  std::vector<Operand> ops{in_op(L, N * N, strideL), in_op(Y, N * J, strideY), out_op(X, N * J)};
  ChunkFn fn = [=](nd4hip_handle* hd, int, int64_t nb, void* const* d) {
    return nd4hip_dpotrs_batched_dev(hd, nb, N, J, P(d, 0), strideL, P(d, 1), strideY, P(d, 2));
  };
  return run_host(h, batch, ops, fn);
}
extern "C" int nd4hip_dldltrf_batched(nd4hip_handle* h, int64_t batch, int64_t N, const double* S, double* LD) {
  ND4_CHECK_ARG(h != nullptr, "nd4hip_dldltrf_batched: NULL handle");
  ND4_CHECK_ARG(batch >= 0 && N >= 0, "nd4hip_dldltrf_batched: negative extent");
  if (batch == 0 || N == 0) return 0;
  ND4_CHECK_ARG(S && LD, "nd4hip_dldltrf_batched: NULL pointer");
  std::vector<Operand> ops{in_op(S, N * N, N * N), out_op(LD, N * N)};
  ChunkFn fn = [=](nd4hip_handle* hd, int, int64_t nb, void* const* d) { return nd4hip_dldltrf_batched_dev(hd, nb, N, P(d, 0), P(d, 1)); };
  return run_host(h, batch, ops, fn);
}
extern "C" int nd4hip_dldltrs_batched(nd4hip_handle* h, int64_t batch, int64_t N, int64_t J, const double* LD, int64_t strideLD,
                                      const double* Y, int64_t strideY, double* X) {
  ND4_CHECK_ARG(h != nullptr, "nd4hip_dldltrs_batched: NULL handle");
  ND4_CHECK_ARG(batch >= 0 && N >= 0 && J >= 0, "nd4hip_dldltrs_batched: negative extent");
  ND4_CHECK_ARG((strideLD == 0 || strideLD >= N * N) && (strideY == 0 || strideY >= N * J),
                "nd4hip_dldltrs_batched: a stride must be 0 or at least the size of one operand");
  if (batch == 0 || N == 0 || J == 0) return 0;
  ND4_CHECK_ARG(LD && Y && X, "nd4hip_dldltrs_batched: NULL pointer");
  std::vector<Operand> ops{in_op(LD, N * N, strideLD), in_op(Y, N * J, strideY), out_op(X, N * J)};
  ChunkFn fn = [=](nd4hip_handle* hd, int, int64_t nb, void* const* d) {
    return nd4hip_dldltrs_batched_dev(hd, nb, N, J, P(d, 0), strideLD, P(d, 1), strideY, P(d, 2));
  };
  return run_host(h, batch, ops, fn);
}

// ---- bidiag_decomp (bidiag.js:245-319), hessenberg_decomp (hessenberg.js:89-115)
extern "C" int nd4hip_dgebrd_batched(nd4hip_handle* h, int64_t batch, int64_t M, int64_t N, const double* A, double* U, double* B, double* V) {
  ND4_CHECK_ARG(h != nullptr, "nd4hip_dgebrd_batched: NULL handle");
  ND4_CHECK_ARG(batch >= 0 && M >= 0 && N >= 0, "nd4hip_dgebrd_batched: negative extent");
  if (batch == 0 || M == 0 || N == 0) return 0;
  ND4_CHECK_ARG(A && U && B && V, "nd4hip_dgebrd_batched: NULL pointer");
  const int64_t K = M < N ? M : N, J = M >= N ? K : K + 1;
  std::vector<Operand> ops{in_op(A, M * N, M * N), out_op(U, M * K), out_op(B, K * J), out_op(V, J * N)};
  ChunkFn fn = [=](nd4hip_handle* hd, int, int64_t nb, void* const* d) { return nd4hip_dgebrd_batched_dev(hd, nb, M, N, P(d, 0), P(d, 1), P(d, 2), P(d, 3)); };
  return run_host(h, batch, ops, fn);
}
extern "C" int nd4hip_dgehrd_batched(nd4hip_handle* h, int64_t batch, int64_t N, const double* A, double* U, double* H) {
  ND4_CHECK_ARG(h != nullptr, "nd4hip_dgehrd_batched: NULL handle");
  ND4_CHECK_ARG(batch >= 0 && N >= 0, "nd4hip_dgehrd_batched: negative extent");
  if (batch == 0 || N == 0) return 0;
  ND4_CHECK_ARG(A && U && H, "nd4hip_dgehrd_batched: NULL pointer");
  std::vector<Operand> ops{in_op(A, N * N, N * N), out_op(U, N * N), out_op(H, N * N)};
  ChunkFn fn = [=](nd4hip_handle* hd, int, int64_t nb, void* const* d) { return nd4hip_dgehrd_batched_dev(hd, nb, N, P(d, 0), P(d, 1), P(d, 2)); };
  return run_host(h, batch, ops, fn);
}

// ------------------------------------------------------------------------------------ QR
extern "C" int nd4hip_dgeqrf_q_batched(nd4hip_handle* h, int64_t batch, int64_t M, int64_t N, const double* A, double* Q, double* R) {
  ND4_CHECK_ARG(h != nullptr, "nd4hip_dgeqrf_q_batched: NULL handle");
  ND4_CHECK_ARG(batch >= 0 && M >= 0 && N >= 0, "nd4hip_dgeqrf_q_batched: negative extent");
  if (batch == 0 || M == 0 || N == 0) return 0;
  ND4_CHECK_ARG(A && Q && R, "nd4hip_dgeqrf_q_batched: NULL pointer");
  const int64_t L = M < N ? M : N;
  std::vector<Operand> ops{in_op(A, M * N, M * N), out_op(Q, M * L), out_op(R, L * N)};
  ChunkFn fn = [=](nd4hip_handle* hd, int, int64_t nb, void* const* d) { return nd4hip_dgeqrf_q_batched_dev(hd, nb, M, N, P(d, 0), P(d, 1), P(d, 2)); };
  return run_host(h, batch, ops, fn);
}
extern "C" int nd4hip_dgeqrf_full_batched(nd4hip_handle* h, int64_t batch, int64_t M, int64_t N, const double* A, double* Q, double* R) {
  ND4_CHECK_ARG(h != nullptr, "nd4hip_dgeqrf_full_batched: NULL handle");
  ND4_CHECK_ARG(batch >= 0 && M >= 0 && N >= 0, "nd4hip_dgeqrf_full_batched: negative extent");
  if (batch == 0 || M == 0 || N == 0) return 0;
  ND4_CHECK_ARG(A && Q && R, "nd4hip_dgeqrf_full_batched: NULL pointer");
  std::vector<Operand> ops{in_op(A, M * N, M * N), out_op(Q, M * M), out_op(R, M * N)};
  ChunkFn fn = [=](nd4hip_handle* hd, int, int64_t nb, void* const* d) { return nd4hip_dgeqrf_full_batched_dev(hd, nb, M, N, P(d, 0), P(d, 1), P(d, 2)); };
  return run_host(h, batch, ops, fn);
}
extern "C" int nd4hip_dgeqrf_qty_batched(nd4hip_handle* h, int64_t batch, int64_t M, int64_t N, int64_t L, double* A, double* Y) {
  ND4_CHECK_ARG(h != nullptr, "nd4hip_dgeqrf_qty_batched: NULL handle");
  ND4_CHECK_ARG(batch >= 0 && M >= 0 && N >= 0 && L >= 0, "nd4hip_dgeqrf_qty_batched: negative extent");
  if (batch == 0 || M == 0 || N == 0) return 0;
  ND4_CHECK_ARG(A && (Y || L == 0), "nd4hip_dgeqrf_qty_batched: NULL pointer");
  std::vector<Operand> ops{inout_op(A, M * N)};
  if (L > 0) ops.push_back(inout_op(Y, M * L));
  ChunkFn fn = [=](nd4hip_handle* hd, int, int64_t nb, void* const* d) {
    return nd4hip_dgeqrf_qty_batched_dev(hd, nb, M, N, L, P(d, 0), L > 0 ? P(d, 1) : nullptr);
  };
  return run_host(h, batch, ops, fn);
}

// ------------------------------------------------------------------------------------ SVD
extern "C" int nd4hip_dgesvdj_batched(nd4hip_handle* h, int64_t batch, int64_t M, int64_t N, const double* A,
                                      double* U, double* sv, double* V, int* sweeps_out, double* offnorm_out) {
  ND4_CHECK_ARG(h != nullptr, "nd4hip_dgesvdj_batched: NULL handle");
  ND4_CHECK_ARG(batch >= 0 && M >= 0 && N >= 0, "nd4hip_dgesvdj_batched: negative extent");
  if (sweeps_out) *sweeps_out = 0;
  if (offnorm_out) *offnorm_out = 0.0;
  if (batch == 0 || M == 0 || N == 0) return 0;
  ND4_CHECK_ARG(A && U && sv && V, "nd4hip_dgesvdj_batched: NULL pointer");
  const int64_t L = M < N ? M : N;
  const int ndev = 1 + (int)h->peers.size();
  // per-device audit (each device's host thread writes its own slot); reduced on the host afterwards — this is the R1
  // "health" reduction of SURVEY.md §8(e), and with host-side results it needs no collective
  std::vector<int> sweeps((size_t)ndev, 0);
  std::vector<double> off((size_t)ndev, 0.0);
  std::vector<unsigned long long> rot((size_t)ndev, 0);
  std::vector<Operand> ops{in_op(A, M * N, M * N), out_op(U, M * L), out_op(sv, L), out_op(V, L * N)};
  ChunkFn fn = [&, M, N](nd4hip_handle* hd, int dev, int64_t nb, void* const* d) {
    int sw = 0; double of = 0.0;
    ND4_TRY(nd4hip_dgesvdj_batched_dev(hd, nb, M, N, P(d, 0), P(d, 1), P(d, 2), P(d, 3), &sw, &of));
    if (sw > sweeps[(size_t)dev]) sweeps[(size_t)dev] = sw;
    if (of > off[(size_t)dev]) off[(size_t)dev] = of;
    rot[(size_t)dev] += hd->svd_rotations;
    return 0;
  };
  // a Jacobi batch below ~100 matrices leaves most of the chip idle: chunks stay large, small batches are not cut at all
  Plan plan; plan.min_chunk = 128; plan.chunk_bytes = size_t(256) << 20;
  ND4_TRY(run_host(h, batch, ops, fn, plan));
  int sw = 0; double of = 0.0; unsigned long long r = 0;
  for (int d = 0; d < ndev; d++) { if (sweeps[(size_t)d] > sw) sw = sweeps[(size_t)d]; if (off[(size_t)d] > of) of = off[(size_t)d]; r += rot[(size_t)d]; }
  h->svd_sweeps = sw; h->svd_offnorm = of; h->svd_rotations = r;
  if (sweeps_out) *sweeps_out = sw;
  if (offnorm_out) *offnorm_out = of;
  return 0;
}
