#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(const double* x, double* o, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { o[2*i] = __builtin_amdgcn_rcp(x[i]); o[2*i+1] = __builtin_amdgcn_rsq(x[i]); }
}
int main() {
  const int n = 1 << 16; double *hx = new double[n], *ho = new double[2*n], *dx, *dout;
  for (int i = 0; i < n; i++) hx[i] = 0.5 + 3.5 * (i + 0.37) / n;
  hipMalloc(&dx, n * 8); hipMalloc(&dout, 2 * n * 8); hipMemcpy(dx, hx, n * 8, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(dx, dout, n); hipMemcpy(ho, dout, 2 * n * 8, hipMemcpyDeviceToHost);
  double e1 = 0, e2 = 0;
  for (int i = 0; i < n; i++) { e1 = fmax(e1, fabs(ho[2*i] * hx[i] - 1.0)); e2 = fmax(e2, fabs(ho[2*i+1] * sqrt(hx[i]) - 1.0)); }
  printf("max rel err: v_rcp_f64 %.3e (2^%.1f)  v_rsq_f64 %.3e (2^%.1f)\n", e1, log2(e1), e2, log2(e2));
  return 0;
}
