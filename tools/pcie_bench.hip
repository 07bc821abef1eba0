// Microbenchmark for the host-pointer path: pageable vs pinned vs hipHostRegister'ed H2D / D2H rates and the price of registering.
// hipcc --offload-arch=gfx950 -O2 tools/pcie_bench.hip -o build/pcie_bench && ./build/pcie_bench
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  for (size_t mb : {8, 128, 1024}) {
    const size_t n = mb << 20;
    void *d, *pinned;
    hipMalloc(&d, n); hipHostMalloc(&pinned, n, hipHostMallocDefault);
    char* pageable = (char*)aligned_alloc(4096, n); memset(pageable, 1, n); memset(pinned, 1, n);
    hipStream_t s; hipStreamCreate(&s);
    auto rate = [&](const char* what, void* dst, const void* src, hipMemcpyKind k) {
      double best = 1e9;
      for (int r = 0; r < 3; r++) { double t = now(); hipMemcpyAsync(dst, src, n, k, s); hipStreamSynchronize(s); t = now() - t; if (t < best) best = t; }
      printf("%5zu MiB %-28s %7.2f ms  %6.1f GB/s\n", mb, what, best * 1e3, n / best / 1e9);
    };
    rate("H2D pageable", d, pageable, hipMemcpyHostToDevice);
    rate("D2H pageable", pageable, d, hipMemcpyDeviceToHost);
    rate("H2D pinned", d, pinned, hipMemcpyHostToDevice);
    rate("D2H pinned", pinned, d, hipMemcpyDeviceToHost);
    double t = now(); hipError_t e = hipHostRegister(pageable, n, hipHostRegisterDefault); double treg = now() - t;
    printf("%5zu MiB hipHostRegister: %s %.2f ms\n", mb, e == hipSuccess ? "ok" : hipGetErrorString(e), treg * 1e3);
    if (e == hipSuccess) {
      rate("H2D registered", d, pageable, hipMemcpyHostToDevice);
      rate("D2H registered", pageable, d, hipMemcpyDeviceToHost);
      t = now(); hipHostUnregister(pageable); printf("%5zu MiB hipHostUnregister %.2f ms\n", mb, (now() - t) * 1e3);
      t = now(); hipHostRegister(pageable, n, hipHostRegisterDefault); printf("%5zu MiB re-register %.2f ms\n", mb, (now() - t) * 1e3); hipHostUnregister(pageable);
    }
    // host memcpy into pinned (the bounce-buffer alternative), 1 thread
    t = now(); memcpy(pinned, pageable, n); printf("%5zu MiB host memcpy pageable->pinned (1 thread) %.2f ms %.1f GB/s\n", mb, (now() - t) * 1e3, n / (now() - t) / 1e9);
    hipFree(d); hipHostFree(pinned); free(pageable); hipStreamDestroy(s);
  }
  return 0;
}
