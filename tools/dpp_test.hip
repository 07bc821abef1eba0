// self-test of nd4js_amd/csrc/dpp.h on the GPU: hipcc --offload-arch=gfx950 tools/dpp_test.hip -o /tmp/dpp_test && /tmp/dpp_test
#include "../nd4js_amd/csrc/dpp.h"
#include <cstdio>
#include <cmath>
__global__ void k(double* out, int* iout) {
  const int l = threadIdx.x;
  const double v = 100.0 + l;
  out[0 * 64 + l] = nd4dpp::xor1(v); out[1 * 64 + l] = nd4dpp::xor2(v); out[2 * 64 + l] = nd4dpp::xor4(v); out[3 * 64 + l] = nd4dpp::xor8(v);
  out[4 * 64 + l] = nd4dpp::wave_sum(v); out[5 * 64 + l] = nd4dpp::wave_max((l * 37) % 64 + 0.5);
  iout[l] = nd4dpp::wave_min((l * 29 + 7) % 64 + 3);
}
int main() {
  double* d; int* di; hipMalloc(&d, 6 * 64 * 8); hipMalloc(&di, 64 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, di); hipDeviceSynchronize();
  double h[6 * 64]; int hi[64]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost); hipMemcpy(hi, di, sizeof hi, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; l++) {
    if (h[l] != 100.0 + (l ^ 1)) bad++; if (h[64 + l] != 100.0 + (l ^ 2)) bad++;
    if (h[128 + l] != 100.0 + (l ^ 4)) bad++; if (h[192 + l] != 100.0 + (l ^ 8)) bad++;
    if (h[256 + l] != 6400.0 + 2016.0) bad++; if (h[320 + l] != 63.5) bad++; if (hi[l] != 3) bad++;
  }
  printf("dpp self-test: %s (%d mismatches) xor4 lane0..7:", bad ? "FAIL" : "ok", bad);
  for (int l = 0; l < 8; l++) printf(" %g", h[128 + l]);
  printf("\n");
  return bad != 0;
}
