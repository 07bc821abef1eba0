// What a FRESH host buffer costs (JS / numpy allocate a new result array per call): D2H / H2D with never-touched pages,
// after a pre-fault, and through a persistent pinned bounce buffer + memcpy.
// hipcc --offload-arch=gfx950 -O2 tools/pcie_fresh.hip -o build/pcie_fresh && ./build/pcie_fresh
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sys/mman.h>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const size_t n = size_t(128) << 20;
  void *d, *pinned; hipMalloc(&d, n); hipHostMalloc(&pinned, n, hipHostMallocDefault); memset(pinned, 1, n);
  hipStream_t s; hipStreamCreate(&s);
  hipMemset(d, 1, n); hipDeviceSynchronize();
  for (int rep = 0; rep < 3; rep++) {
    char* f = (char*)malloc(n);                       // fresh: pages not touched yet
    double t = now(); hipMemcpyAsync(f, d, n, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); double t1 = now() - t;
    t = now(); hipMemcpyAsync(f, d, n, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); double t2 = now() - t;
    printf("D2H 128 MiB into fresh malloc: first %.2f ms, again %.2f ms\n", t1 * 1e3, t2 * 1e3);
    free(f);
    f = (char*)malloc(n);
    t = now(); for (size_t i = 0; i < n; i += 4096) f[i] = 0; double tp = now() - t;
    t = now(); hipMemcpyAsync(f, d, n, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); t1 = now() - t;
    printf("  prefault (1 thread) %.2f ms, then D2H %.2f ms\n", tp * 1e3, t1 * 1e3);
    free(f);
    f = (char*)malloc(n);
    t = now();
    { std::vector<std::thread> th; const int T = 8; for (int k = 0; k < T; k++) th.emplace_back([=] { for (size_t i = n / T * k; i < n / T * (k + 1); i += 4096) f[i] = 0; }); for (auto& x : th) x.join(); }
    tp = now() - t;
    t = now(); hipMemcpyAsync(f, d, n, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); t1 = now() - t;
    printf("  prefault (8 threads) %.2f ms, then D2H %.2f ms\n", tp * 1e3, t1 * 1e3);
    free(f);
    f = (char*)malloc(n);
    t = now(); hipMemcpyAsync(pinned, d, n, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); double tc = now() - t;
    t = now(); memcpy(f, pinned, n); double tm = now() - t;
    printf("  D2H into pinned %.2f ms + memcpy to fresh %.2f ms\n", tc * 1e3, tm * 1e3);
    free(f);
    f = (char*)malloc(n); memset(f, 1, n);            // input-like: touched once by the producer
    t = now(); hipMemcpyAsync(d, f, n, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); t1 = now() - t;
    t = now(); hipMemcpyAsync(d, f, n, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); t2 = now() - t;
    printf("  H2D from a touched, never-copied buffer: first %.2f ms, again %.2f ms\n", t1 * 1e3, t2 * 1e3);
    free(f);
  }
  return 0;
}
