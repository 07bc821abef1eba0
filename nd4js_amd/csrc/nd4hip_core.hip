// libnd4hip.so core: handle lifecycle, error reporting, memory, timing, synthetic-input fill and the
// small copy / transpose / identity kernels shared by the decompositions.
#include "nd4hip_internal.h"
#include <cstdlib>
#include <cstring>
#include <new>

static thread_local char g_err[512] = "";

void nd4_set_error(const char* fmt, ...) {
  va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
}
int nd4_hip_fail(hipError_t e, const char* what, const char* file, int line) {
  nd4_set_error("HIP error %d (%s) in `%s` at %s:%d", (int)e, hipGetErrorString(e), what, file, line);
  return ND4HIP_ERR_HIP;
}

extern "C" const char* nd4hip_last_error(void) { return g_err; }
extern "C" const char* nd4hip_version(void) { return "nd4hip 0.1.0 (gfx950)"; }

extern "C" int nd4hip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
  return n;
}

extern "C" int nd4hip_create(nd4hip_handle** out, int device) {
  ND4_CHECK_ARG(out != nullptr, "nd4hip_create: out is NULL");
  *out = nullptr;
  int n = nd4hip_device_count();
  if (n <= 0) { nd4_set_error("nd4hip_create: no HIP device available"); return ND4HIP_ERR_NODEV; }
  if (device < 0) ND4_HIP(hipGetDevice(&device));
  ND4_CHECK_ARG(device < n, "nd4hip_create: device %d out of range (%d devices)", device, n);
  ND4_HIP(hipSetDevice(device));
  nd4hip_handle* h = new (std::nothrow) nd4hip_handle();
  ND4_CHECK_ARG(h != nullptr, "nd4hip_create: out of host memory");
  h->device = device;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess) h->num_cu = prop.multiProcessorCount;
  hipError_t e = hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreate(&h->ev0);
  if (e == hipSuccess) e = hipEventCreate(&h->ev1);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_order, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreate(&h->ev_p0);
  if (e == hipSuccess) e = hipEventCreate(&h->ev_p1);
  if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&h->xstat), 64, hipHostMallocMapped | hipHostMallocCoherent);
  if (e == hipSuccess) *h->xstat = 0;
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->aux_stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_aux_a, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_aux_b, hipEventDisableTiming);
  for (int i = 0; i < 2 && e == hipSuccess; i++) {
    e = hipEventCreateWithFlags(&h->ev_in[i], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_chunk[i], hipEventDisableTiming);
  }
  if (e != hipSuccess) { nd4hip_destroy(h); return nd4_hip_fail(e, "stream/event create", __FILE__, __LINE__); }
  h->stream = h->own_stream;
  *out = h;
  return 0;
}

extern "C" void nd4hip_destroy(nd4hip_handle* h) {
  if (!h) return;
  for (nd4hip_handle* p : h->peers) nd4hip_destroy(p);
  h->peers.clear();
  int prev = -1;
  (void)hipGetDevice(&prev);
  (void)hipSetDevice(h->device);
  if (h->copy_stream) { (void)hipStreamSynchronize(h->copy_stream); (void)hipStreamDestroy(h->copy_stream); }
  if (h->aux_stream) { (void)hipStreamSynchronize(h->aux_stream); (void)hipStreamDestroy(h->aux_stream); }
  if (h->ev_aux_a) (void)hipEventDestroy(h->ev_aux_a);
  if (h->ev_aux_b) (void)hipEventDestroy(h->ev_aux_b);
  for (int i = 0; i < 2; i++) { if (h->ev_in[i]) (void)hipEventDestroy(h->ev_in[i]); if (h->ev_chunk[i]) (void)hipEventDestroy(h->ev_chunk[i]); }
  if (h->own_stream) (void)hipStreamSynchronize(h->own_stream);
  for (auto& b : h->ws) (void)hipFree(b.p);
  for (auto& b : h->stage) (void)hipFree(b.p);
  if (h->pinned) (void)hipHostFree(h->pinned);
  if (h->xstat) (void)hipHostFree(h->xstat);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  if (h->ev_order) (void)hipEventDestroy(h->ev_order);
  if (h->ev_p0) (void)hipEventDestroy(h->ev_p0);
  if (h->ev_p1) (void)hipEventDestroy(h->ev_p1);
  if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
  delete h;
  if (prev >= 0) (void)hipSetDevice(prev);
}

extern "C" int nd4hip_set_stream(nd4hip_handle* h, void* s) {
  ND4_CHECK_ARG(h != nullptr, "nd4hip_set_stream: NULL handle");
  hipStream_t next = reinterpret_cast<hipStream_t>(s);     // NULL = HIP's default stream
  if (next != h->stream) {
    // the workspace arena (and cached staging) is reused in stream order: work queued on the new stream must not start
    // before what the old stream still has in flight on those blocks
    Nd4DeviceGuard guard(h);
    // (the previous stream must still be alive here; if the caller has already destroyed it the record fails: order by a device
    //  synchronisation instead and switch anyway, so that the handle does not stay on a dead stream)
    if (hipEventRecord(h->ev_order, h->stream) == hipSuccess) {
      ND4_HIP(hipStreamWaitEvent(next, h->ev_order, 0));
    } else {
      (void)hipGetLastError();
      (void)hipDeviceSynchronize();
      (void)hipGetLastError();
    }
    h->stream = next;
  }
  return 0;
}
extern "C" int nd4hip_reset_stream(nd4hip_handle* h) {
  ND4_CHECK_ARG(h != nullptr, "nd4hip_reset_stream: NULL handle");
  return nd4hip_set_stream(h, h->own_stream);
}
extern "C" int nd4hip_synchronize(nd4hip_handle* h) {
  ND4_CHECK_ARG(h != nullptr, "nd4hip_synchronize: NULL handle");
  Nd4DeviceGuard guard(h);
  ND4_HIP(hipStreamSynchronize(h->stream));
  return nd4_xchg_check(h, "nd4hip_synchronize");
}

int nd4_xchg_check(nd4hip_handle* h, const char* where) {
  if (h->xstat == nullptr || *static_cast<volatile int*>(h->xstat) == 0) return 0;
  *static_cast<volatile int*>(h->xstat) = 0;
  nd4_set_error("%s: an exchange between workgroups inside a kernel timed out; the results of the calls since the last "
                "synchronisation are invalid (NaN in Q / R, -1 in the permutation vector of an LU)", where);
  return ND4HIP_ERR_XCHG;
}
int nd4_test_drop_panel() {
  const char* e = getenv("ND4HIP_TEST_DROP_PUBLISH");
  return (e && *e) ? atoi(e) : -1;
}

int nd4_ws_alloc(nd4hip_handle* h, size_t bytes, void** out) {
  bytes = (bytes + 255) & ~size_t(255);
  if (bytes == 0) bytes = 256;
  // first fit in the LAST block only (LIFO discipline keeps release trivial)
  if (!h->ws.empty()) {
    Nd4WsBlock& b = h->ws.back();
    if (b.size - b.used >= bytes) { *out = b.p + b.used; b.used += bytes; return 0; }
  }
  // nothing in use anywhere and nothing fits: drop the cached blocks instead of piling up new ones
  bool idle = true;
  for (auto& b : h->ws) idle = idle && b.used == 0;
  if (idle && !h->ws.empty()) {
    for (auto& b : h->ws) if (b.size >= bytes) { std::swap(b, h->ws.back()); h->ws.back().used = bytes; *out = h->ws.back().p; return 0; }
    ND4_HIP(hipStreamSynchronize(h->stream));
    for (auto& b : h->ws) (void)hipFree(b.p);
    h->ws.clear();
    h->ws_generation++;
  }
  size_t want = bytes < (size_t(64) << 20) ? (size_t(64) << 20) : bytes;
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, want);
  if (e != hipSuccess && want != bytes) { (void)hipGetLastError(); want = bytes; e = hipMalloc(&p, want); }
  if (e != hipSuccess) return nd4_hip_fail(e, "hipMalloc(workspace)", __FILE__, __LINE__);
  h->ws.push_back(Nd4WsBlock{static_cast<char*>(p), want, bytes});
  *out = p;
  return 0;
}
Nd4WsScope::Nd4WsScope(nd4hip_handle* hh)
    : h(hh), nblocks(hh->ws.size()), used_last(hh->ws.empty() ? 0 : hh->ws.back().used), generation(hh->ws_generation) {}
Nd4WsScope::~Nd4WsScope() {
  if (h->ws_generation != generation) {
    // the arena was dropped and rebuilt inside this scope: that only happens while NOTHING is in use, so this scope (and any
    // enclosing one) held no allocation at that point and everything allocated since belongs to it -> all blocks are free again
    for (auto& b : h->ws) b.used = 0;
    return;
  }
  // blocks opened inside the scope become empty (kept for reuse); the block that was last at entry
  // returns to its old fill level
  for (size_t i = nblocks; i < h->ws.size(); i++) h->ws[i].used = 0;
  if (nblocks > 0) h->ws[nblocks - 1].used = used_last;
}
int nd4_pinned(nd4hip_handle* h, size_t bytes, void** out) {
  if (bytes > h->pinned_bytes) {
    if (h->pinned) { ND4_HIP(hipHostFree(h->pinned)); h->pinned = nullptr; h->pinned_bytes = 0; }
    size_t want = bytes < 4096 ? 4096 : bytes;
    ND4_HIP(hipHostMalloc(&h->pinned, want, hipHostMallocDefault));
    h->pinned_bytes = want;
  }
  *out = h->pinned;
  return 0;
}

extern "C" int nd4hip_malloc(nd4hip_handle* h, size_t bytes, void** p) {
  ND4_CHECK_ARG(h && p, "nd4hip_malloc: NULL argument");
  Nd4DeviceGuard guard(h);
  ND4_HIP(hipMalloc(p, bytes ? bytes : 8));
  return 0;
}
extern "C" int nd4hip_free(nd4hip_handle* h, void* p) {
  ND4_CHECK_ARG(h != nullptr, "nd4hip_free: NULL handle");
  Nd4DeviceGuard guard(h);
  if (p) { ND4_HIP(hipStreamSynchronize(h->stream)); ND4_HIP(hipFree(p)); }
  return 0;
}
extern "C" int nd4hip_memcpy_h2d(nd4hip_handle* h, void* d, const void* s, size_t bytes) {
  ND4_CHECK_ARG(h != nullptr, "nd4hip_memcpy_h2d: NULL handle");
  Nd4DeviceGuard guard(h);
  if (bytes) ND4_HIP(hipMemcpyAsync(d, s, bytes, hipMemcpyHostToDevice, h->stream));
  ND4_HIP(hipStreamSynchronize(h->stream));
  return 0;
}
extern "C" int nd4hip_memcpy_d2h(nd4hip_handle* h, void* d, const void* s, size_t bytes) {
  ND4_CHECK_ARG(h != nullptr, "nd4hip_memcpy_d2h: NULL handle");
  Nd4DeviceGuard guard(h);
  if (bytes) ND4_HIP(hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToHost, h->stream));
  ND4_HIP(hipStreamSynchronize(h->stream));
  return 0;
}

Nd4Prof::Nd4Prof(nd4hip_handle* hh, const char* op, double flops, double bytes) : h(hh), active(false) {
  if (!h || !h->prof_on) return;
  if (h->prof_depth++ > 0) return;                       // an entry point called from another one
  active = true;
  h->prof_flops = flops; h->prof_bytes = bytes; h->prof_valid = false;
  snprintf(h->prof_op, sizeof h->prof_op, "%s", op);
  (void)hipEventRecord(h->ev_p0, h->stream);
}
Nd4Prof::~Nd4Prof() {
  if (!h || !h->prof_on) return;
  if (h->prof_depth > 0) h->prof_depth--;
  if (!active) return;
  h->prof_valid = hipEventRecord(h->ev_p1, h->stream) == hipSuccess;
}
extern "C" int nd4hip_profile_enable(nd4hip_handle* h, int on) {
  ND4_CHECK_ARG(h != nullptr, "nd4hip_profile_enable: NULL handle");
  h->prof_on = on != 0; h->prof_depth = 0; if (!on) h->prof_valid = false;
  for (nd4hip_handle* p : h->peers) { p->prof_on = on != 0; p->prof_depth = 0; if (!on) p->prof_valid = false; }
  return 0;
}
extern "C" int nd4hip_profile_last(nd4hip_handle* h, nd4hip_prof* out, int capacity) {
  ND4_CHECK_ARG(h != nullptr && (out != nullptr || capacity == 0), "nd4hip_profile_last: NULL argument");
  const int n = 1 + (int)h->peers.size();
  for (int i = 0; i < n && i < capacity; i++) {
    nd4hip_handle* d = i == 0 ? h : h->peers[(size_t)i - 1];
    nd4hip_prof& r = out[i];
    memset(&r, 0, sizeof r);
    r.device = d->device;
    if (!d->prof_valid) continue;
    Nd4DeviceGuard guard(d);
    float ms = 0.0f;
    ND4_HIP(hipEventSynchronize(d->ev_p1));
    ND4_HIP(hipEventElapsedTime(&ms, d->ev_p0, d->ev_p1));
    r.kernel_ms = ms; r.flops = d->prof_flops; r.bytes = d->prof_bytes; r.valid = 1;
    snprintf(r.op, sizeof r.op, "%s", d->prof_op);
  }
  return n;
}

extern "C" int nd4hip_timer_start(nd4hip_handle* h) {
  ND4_CHECK_ARG(h != nullptr, "nd4hip_timer_start: NULL handle");
  Nd4DeviceGuard guard(h);
  ND4_HIP(hipEventRecord(h->ev0, h->stream));
  return 0;
}
extern "C" int nd4hip_timer_stop(nd4hip_handle* h, float* ms) {
  ND4_CHECK_ARG(h && ms, "nd4hip_timer_stop: NULL argument");
  Nd4DeviceGuard guard(h);
  ND4_HIP(hipEventRecord(h->ev1, h->stream));
  ND4_HIP(hipEventSynchronize(h->ev1));
  ND4_HIP(hipEventElapsedTime(ms, h->ev0, h->ev1));
  return nd4_xchg_check(h, "nd4hip_timer_stop");
}

// ---------------------------------------------------------------- synthetic inputs
__device__ __forceinline__ uint32_t fmix32(uint32_t x) {
  x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16; return x;
}
__global__ void fill_uniform_kernel(uint32_t seed, uint32_t offset, int64_t n, double* __restrict__ out) {
  const uint32_t s = fmix32(seed);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t idx = offset + (uint32_t)i;
    const uint32_t hi = fmix32(idx ^ s);
    const uint32_t lo = fmix32(hi + 0x9E3779B9u + idx);
    const double m = (double)(hi >> 5) * 67108864.0 + (double)(lo >> 6);
    out[i] = m * 0x1p-52 - 1.0;
  }
}
extern "C" int nd4hip_fill_uniform_dev(nd4hip_handle* h, uint32_t seed, uint32_t offset, int64_t n, double* out) {
  ND4_CHECK_ARG(h && (out || n == 0) && n >= 0, "nd4hip_fill_uniform_dev: bad argument");
  Nd4DeviceGuard guard(h);
  if (n == 0) return 0;
  int64_t blocks = (n + 255) / 256; if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(fill_uniform_kernel, dim3((unsigned)blocks), dim3(256), 0, h->stream, seed, offset, n, out);
  ND4_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------- small matrix utilities
__global__ void copy_matrix_kernel(int64_t rows, int64_t cols, const double* __restrict__ src, int64_t lds,
                                   double* __restrict__ dst, int64_t ldd, int64_t ssrc, int64_t sdst) {
  src += blockIdx.z * ssrc; dst += blockIdx.z * sdst;
  const int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (c >= cols) return;
  for (int64_t r = blockIdx.y; r < rows; r += gridDim.y) dst[r * ldd + c] = src[r * lds + c];
}
int nd4_copy_matrix(nd4hip_handle* h, int64_t rows, int64_t cols, const double* src, int64_t lds,
                    double* dst, int64_t ldd, int64_t batch, int64_t ssrc, int64_t sdst) {
  if (rows <= 0 || cols <= 0 || batch <= 0) return 0;
  for (int64_t b0 = 0; b0 < batch; b0 += 32768) {                  // gridDim.z limit
    const int64_t nb = batch - b0 < 32768 ? batch - b0 : 32768;
    dim3 grid((unsigned)((cols + 255) / 256), (unsigned)(rows < 1024 ? rows : 1024), (unsigned)nb);
    hipLaunchKernelGGL(copy_matrix_kernel, grid, dim3(256), 0, h->stream, rows, cols, src + b0 * ssrc, lds, dst + b0 * sdst, ldd, ssrc, sdst);
  }
  ND4_HIP(hipGetLastError());
  return 0;
}

// dst[c][r] = src[r][c], 32x32 tiles through LDS (padded: conflict-free both ways)
__global__ void transpose_kernel(int64_t rows, int64_t cols, const double* __restrict__ src, int64_t lds,
                                 double* __restrict__ dst, int64_t ldd, int64_t ssrc, int64_t sdst) {
  __shared__ double tile[32][33];
  src += blockIdx.z * ssrc; dst += blockIdx.z * sdst;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 256 threads: 32 x 8
  const int64_t c0 = blockIdx.x * 32ll, r0 = blockIdx.y * 32ll;
  for (int i = ty; i < 32; i += 8)
    if (r0 + i < rows && c0 + tx < cols) tile[i][tx] = src[(r0 + i) * lds + c0 + tx];
  __syncthreads();
  for (int i = ty; i < 32; i += 8)
    if (c0 + i < cols && r0 + tx < rows) dst[(c0 + i) * ldd + r0 + tx] = tile[tx][i];
}
int nd4_transpose(nd4hip_handle* h, int64_t rows, int64_t cols, const double* src, int64_t lds,
                  double* dst, int64_t ldd, int64_t batch, int64_t ssrc, int64_t sdst) {
  if (rows <= 0 || cols <= 0 || batch <= 0) return 0;
  dim3 grid((unsigned)((cols + 31) / 32), (unsigned)((rows + 31) / 32), (unsigned)batch);
  hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, h->stream, rows, cols, src, lds, dst, ldd, ssrc, sdst);
  ND4_HIP(hipGetLastError());
  return 0;
}

__global__ void identity_kernel(int64_t rows, int64_t cols, double* __restrict__ dst, int64_t ldd, int64_t sdst) {
  dst += blockIdx.z * sdst;
  const int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (c >= cols) return;
  for (int64_t r = blockIdx.y; r < rows; r += gridDim.y) dst[r * ldd + c] = (r == c) ? 1.0 : 0.0;
}
int nd4_set_identity(nd4hip_handle* h, int64_t rows, int64_t cols, double* dst, int64_t ldd,
                     int64_t batch, int64_t sdst) {
  if (rows <= 0 || cols <= 0 || batch <= 0) return 0;
  dim3 grid((unsigned)((cols + 255) / 256), (unsigned)(rows < 1024 ? rows : 1024), (unsigned)batch);
  hipLaunchKernelGGL(identity_kernel, grid, dim3(256), 0, h->stream, rows, cols, dst, ldd, sdst);
  ND4_HIP(hipGetLastError());
  return 0;
}
