// Microbenchmark: what does ONE exchange between co-resident workgroups cost inside a kernel on MI355X?
// P workgroups (every `stride`-th block of the launch: stride 8 keeps them on one XCD, stride 1 spreads them over all eight)
// publish 32 doubles each, raise a flag, wait for the flags of all P and read all P payloads: the pattern of a pivot search split
// over workgroups (lu.hip) or of a reduction finished by every workgroup. Modes:
//   0  payload by plain stores, release fence (agent), flag; reader: flag poll, acquire fence (agent), plain loads
//   1  payload by relaxed agent-scope atomic stores (sc1: written through), s_waitcnt, flag; reader: flag poll, payload by relaxed
//      agent-scope atomic loads (no cache write-back / invalidate at all)
// `bg` > 0: the other workgroups of the launch stream (read-modify-write) over a large buffer meanwhile, bg passes each.
// Every spin is bounded (a stuck exchange sets err and every workgroup leaves).
// hipcc --offload-arch=gfx950 -O3 tools/xwg_lat.hip -o gpurun_out/xwg_lat && ./gpurun_out/xwg_lat
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef unsigned long long u64;

template <int MODE>
__global__ void __launch_bounds__(512) xchg(u64* payload, int* flags, int* err, double* out, double* bgbuf, size_t bgn,
                                            int P, int stride, int rounds, int bg) {
  const int t = threadIdx.x;
  const bool part = (blockIdx.x % stride) == 0 && (int)(blockIdx.x / stride) < P;
  if (!part) {
    if (bg > 0) {
      const size_t nb = gridDim.x, per = bgn / nb;
      double* p = bgbuf + (size_t)blockIdx.x * per;
      for (int it = 0; it < bg; it++)
        for (size_t i = t; i < per; i += blockDim.x) p[i] = p[i] * 1.0000001 + 1.0;
    }
    return;
  }
  const int rank = blockIdx.x / stride;
  __shared__ int bad;
  if (t == 0) bad = 0;
  __syncthreads();
  double acc = 0.0;
  const u64 w0 = wall_clock64();
  for (int r = 0; r < rounds; r++) {
    u64* mine = payload + ((size_t)(r & 1) * P + rank) * 32;
    if (t < 32) {
      const double v = (double)(rank + 1) * (r + 1) + t;
      if (MODE == 0) ((double*)mine)[t] = v;
      else __hip_atomic_store(mine + t, (u64)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (t < 64) {   // wave 0: all its stores have left before the flag goes up
      if (MODE == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (t == 0) __hip_atomic_store(flags + rank, r + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (t < P) {
      int spins = 0;
      while (__hip_atomic_load(flags + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < r + 1) {
        if (++spins > (1 << 22) || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { bad = 1; break; }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    __syncthreads();
    if (bad) { if (t == 0) __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
    if (MODE == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    const u64* all = payload + (size_t)(r & 1) * P * 32;
    for (int i = t; i < P * 32; i += blockDim.x) {
      u64 w;
      if (MODE == 0) w = all[i];
      else w = __hip_atomic_load(all + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const double v = __longlong_as_double((long long)w);
      const double want = (double)(i / 32 + 1) * (r + 1) + (i & 31);
      if (v != want) bad = 2;
      acc += v;
    }
    __syncthreads();
    if (bad) { if (t == 0) __hip_atomic_store(err, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
  }
  out[(size_t)rank * 512 + t] = acc;
  if (rank == 0 && t == 0) out[0] = (double)(wall_clock64() - w0);   // 100 MHz ticks
}

template <int MODE>
static void run(int P, int stride, int rounds, int bg, double* bgbuf, size_t bgn) {
  u64* payload; int *flags, *err; double* out;
  CK(hipMalloc(&payload, sizeof(u64) * 2 * P * 32));
  CK(hipMalloc(&flags, sizeof(int) * P));
  CK(hipMalloc(&err, sizeof(int)));
  CK(hipMalloc(&out, sizeof(double) * P * 512));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int grid = bg > 0 ? 256 : P * stride;
  float best = 1e30f; int herr = 0;
  for (int rep = 0; rep < 3; rep++) {
    CK(hipMemset(flags, 0, sizeof(int) * P)); CK(hipMemset(err, 0, sizeof(int)));
    CK(hipEventRecord(e0));
    xchg<MODE><<<grid, 512>>>(payload, flags, err, out, bgbuf, bgn, P, stride, rounds, bg);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(&herr, err, sizeof(int), hipMemcpyDeviceToHost));
    if (herr) break;
    double ticks; CK(hipMemcpy(&ticks, out, sizeof(double), hipMemcpyDeviceToHost));
    ms = (float)(ticks * 1e-5);   // wall_clock64: 100 MHz
    if (ms < best) best = ms;
  }
  printf("mode %d  P %3d  stride %d  bg %d: %8.3f us per exchange%s\n", MODE, P, stride, bg, best * 1e3 / rounds,
         herr == 1 ? "  [STUCK]" : herr == 2 ? "  [WRONG DATA]" : "");
  fflush(stdout);
  CK(hipFree(payload)); CK(hipFree(flags)); CK(hipFree(err)); CK(hipFree(out));
}

int main() {
  const size_t bgn = (size_t)64 << 20;   // 512 MiB of doubles
  double* bgbuf; CK(hipMalloc(&bgbuf, bgn * sizeof(double))); CK(hipMemset(bgbuf, 0, bgn * sizeof(double)));
  const int rounds = 2000;
  for (int stride : {8, 1})
    for (int P : {2, 4, 8, 16, 32}) {
      if (P * stride > 256) continue;
      run<0>(P, stride, rounds, 0, bgbuf, bgn);
      run<1>(P, stride, rounds, 0, bgbuf, bgn);
    }
  // with the rest of the chip streaming through the same L2s (2 passes over 512 MiB: ~ 1 ms of background work)
  for (int stride : {8, 1})
    for (int P : {4, 16}) {
      run<0>(P, stride, 400, 8, bgbuf, bgn);
      run<1>(P, stride, 400, 8, bgbuf, bgn);
    }
  CK(hipFree(bgbuf));
  return 0;
}
