// Microbenchmark: sustained v_mfma_f64_16x16x4_f64 rate with operands in registers (no memory traffic),
// 1 or 2 waves per SIMD, 16 independent accumulators per wave (same as dgemm_kernel). Random-ish data.
// hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o gpurun_out/mfma_peak && ./gpurun_out/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256, 2) void k(double* out, int iters, double seed) {
  d4 acc[16];
  const int l = threadIdx.x;
  double a[4], b[4];
  for (int i = 0; i < 4; i++) { a[i] = seed * (l * 0.37 + i) - 0.5; b[i] = 0.25 - seed * (l * 0.11 + i); }
  for (int i = 0; i < 16; i++) acc[i] = d4{0, 0, 0, 0};
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = 0; j < 4; j++) acc[i * 4 + j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i * 4 + j], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < 16; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  double* out; hipMalloc(&out, sizeof(double) * 256 * 2048);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wgs_per_cu = 1; wgs_per_cu <= 2; wgs_per_cu++) {
    const int grid = 256 * wgs_per_cu, iters = 4000;
    for (int rep = 0; rep < 3; rep++) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, out, iters, 1e-3);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double flops = (double)grid * 4 * iters * 16 * 2048.0;
      printf("waves/SIMD %d: %.3f ms  %.2f TFLOP/s  (%.1f%% of 78.6)\n", wgs_per_cu, ms, flops / ms / 1e9, flops / ms / 1e9 / 78.6 * 100);
    }
  }
  return 0;
}
