// Which SIMD does each wave of a 576-thread workgroup (9 waves: svd_block.hip's rotation kernel) run on? HW_ID bits [5:4] = SIMD_ID.
// hipcc --offload-arch=gfx950 -O2 tools/simd_probe.hip -o /tmp/simd_probe && /tmp/simd_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(576) void probe(int* out) {
  unsigned hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = (int)hw;
}
int main() {
  int* d; hipMalloc(&d, 64 * 16 * sizeof(int));
  hipLaunchKernelGGL(probe, dim3(64), dim3(576), 0, 0, d);
  int h[64 * 16]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int b = 0; b < 6; b++) {
    printf("wg %d:", b);
    for (int w = 0; w < 9; w++) printf(" w%d:simd%d(cu%d)", w, (h[b * 16 + w] >> 4) & 3, (h[b * 16 + w] >> 8) & 15);
    printf("\n");
  }
  return 0;
}
