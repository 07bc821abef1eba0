// Microbenchmark: latencies that bound the rotation rounds of jacb_eigen_x (one workgroup on one CU):
// dependent v_fma_f64 / v_rsq_f64 / v_rcp_f64 chains, independent v_fma_f64 issue (1 and 2 waves per SIMD), dependent LDS
// read, ds_bpermute, LDS write+barrier round trip with 4 / 8 / 16 waves. Cycles from s_memtime (shader clock).
// hipcc --offload-arch=gfx950 -O3 tools/lat_f64.hip -o gpurun_out/lat_f64 && ./gpurun_out/lat_f64
#include <hip/hip_runtime.h>
#include <cstdio>
#define STAMP() __builtin_amdgcn_s_memtime()
__global__ void k(double* out, unsigned long long* cyc, double seed, int n) {
  __shared__ double lds[4096];
  __shared__ int idx[1024];
  const int t = threadIdx.x;
  for (int i = t; i < 4096; i += blockDim.x) lds[i] = seed * i;
  for (int i = t; i < 1024; i += blockDim.x) idx[i] = (i * 17 + 1) & 1023;
  __syncthreads();
  double x = seed * (t + 1), y = 1.0 + seed, z = 0.5;
  unsigned long long t0, t1;
  // 1: dependent fma chain
  t0 = STAMP();
  for (int i = 0; i < n; i++) { x = fma(x, y, z); x = fma(x, y, z); x = fma(x, y, z); x = fma(x, y, z); }
  t1 = STAMP(); if (t == 0) cyc[0] = (t1 - t0);
  // 2: 4 independent fma chains
  double a0 = x, a1 = x + 1, a2 = x + 2, a3 = x + 3;
  t0 = STAMP();
  for (int i = 0; i < n; i++) { a0 = fma(a0, y, z); a1 = fma(a1, y, z); a2 = fma(a2, y, z); a3 = fma(a3, y, z); }
  t1 = STAMP(); if (t == 0) cyc[1] = (t1 - t0);
  x = a0 + a1 + a2 + a3;
  // 3: dependent rsq chain
  x = fabs(x) + 1.0;
  t0 = STAMP();
  for (int i = 0; i < n; i++) { x = __builtin_amdgcn_rsq(x); x = __builtin_amdgcn_rsq(x); x = __builtin_amdgcn_rsq(x); x = __builtin_amdgcn_rsq(x); }
  t1 = STAMP(); if (t == 0) cyc[2] = (t1 - t0);
  // 4: dependent rcp chain
  t0 = STAMP();
  for (int i = 0; i < n; i++) { x = __builtin_amdgcn_rcp(x); x = __builtin_amdgcn_rcp(x); x = __builtin_amdgcn_rcp(x); x = __builtin_amdgcn_rcp(x); }
  t1 = STAMP(); if (t == 0) cyc[3] = (t1 - t0);
  // 5: dependent LDS read (pointer chase)
  int p = t & 1023;
  t0 = STAMP();
  for (int i = 0; i < n; i++) { p = idx[p]; p = idx[p]; p = idx[p]; p = idx[p]; }
  t1 = STAMP(); if (t == 0) cyc[4] = (t1 - t0);
  // 6: dependent bpermute
  int q = p;
  t0 = STAMP();
  for (int i = 0; i < n; i++) { q = __shfl(q, (q + 1) & 63); q = __shfl(q, (q + 1) & 63); q = __shfl(q, (q + 1) & 63); q = __shfl(q, (q + 1) & 63); }
  t1 = STAMP(); if (t == 0) cyc[5] = (t1 - t0);
  // 7: LDS write -> barrier -> LDS read of another thread's value (the per-round hand-off)
  double w = x;
  t0 = STAMP();
  for (int i = 0; i < n; i++) {
    lds[t] = w; __syncthreads(); w = lds[(t + 65) % blockDim.x] + 1.0;
    lds[2048 + t] = w; __syncthreads(); w = lds[2048 + (t + 65) % blockDim.x] + 1.0;
    lds[t] = w; __syncthreads(); w = lds[(t + 65) % blockDim.x] + 1.0;
    lds[2048 + t] = w; __syncthreads(); w = lds[2048 + (t + 65) % blockDim.x] + 1.0;
  }
  t1 = STAMP(); if (t == 0) cyc[6] = (t1 - t0);
  // 8: fast_rsqrt-like chain (rsq + 2 newton), dependent
  x = fabs(w) + 1.5;
  t0 = STAMP();
  for (int i = 0; i < n; i++) {
#pragma unroll
    for (int u = 0; u < 4; u++) { double r = __builtin_amdgcn_rsq(x); r = r * fma(-0.5 * x * r, r, 1.5); r = r * fma(-0.5 * x * r, r, 1.5); x = r + 1.25; }
  }
  t1 = STAMP(); if (t == 0) cyc[7] = (t1 - t0);
  out[blockIdx.x * blockDim.x + t] = x + p + q + w;
}
int main() {
  double* out; unsigned long long* cyc, h[8];
  hipMalloc(&out, sizeof(double) * 4096); hipMalloc(&cyc, 64);
  const int n = 256;
  const char* names[8] = {"dependent v_fma_f64", "4 independent v_fma_f64 (per instr)", "dependent v_rsq_f64", "dependent v_rcp_f64", "dependent ds_read_b32",
                          "dependent ds_bpermute_b32", "ds_write_b64 + barrier + ds_read_b64 + add", "rsq + 2 Newton + add (9-10 ops)"};
  for (int threads : {64, 256, 512, 1024}) {
    for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(k, dim3(1), dim3(threads), 0, 0, out, cyc, 1e-3, n);
    hipDeviceSynchronize();
    hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    printf("---- %d threads (%d waves on one CU)\n", threads, threads / 64);
    for (int i = 0; i < 8; i++) printf("%-48s %.1f cycles\n", names[i], (double)h[i] / (4.0 * n));
  }
  return 0;
}
