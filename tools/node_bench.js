'use strict';
/* End-to-end timing through the JS host: host-resident NDArrays (H2D + kernel + D2H per call) against
 * device-resident DeviceNDArrays (SURVEY.md §8f N3). Usage: node tools/node_bench.js [N] [svd batch] */
const path = require('path');
const la = require(path.join(__dirname, '..', 'nd4js_amd', 'js'));
const N = parseInt(process.argv[2] || '4096');
function fill(seed, shape) {                     // same counter-based generator as nd4js_amd/rng.py
  const n = shape.reduce((a, b) => a * b, 1), out = new Float64Array(n);
  const fmix = h => { h ^= h >>> 16; h = Math.imul(h, 0x85ebca6b); h ^= h >>> 13; h = Math.imul(h, 0xc2b2ae35); h ^= h >>> 16; return h >>> 0; };
  for (let i = 0; i < n; i++) {
    const hi = fmix((seed * 0x9e3779b1 + 2 * i) >>> 0), lo = fmix((seed * 0x9e3779b1 + 2 * i + 1) >>> 0);
    out[i] = ((hi >>> 5) * 67108864 + (lo >>> 6)) / 9007199254740992 * 2 - 1;
  }
  return new la.NDArray(Int32Array.from(shape), out);
}
const ms = f => { const t = process.hrtime.bigint(); f(); return Number(process.hrtime.bigint() - t) / 1e6; };
const best = (f, n) => { let b = Infinity; for (let i = 0; i < n; i++) b = Math.min(b, ms(f)); return b; };
const A = fill(5, [N, N]), B = fill(6, [N, N]);
la.matmul2(A, B);                                                                  // warm-up (library load, handle)
const host = best(() => la.matmul2(A, B), 3);
const dA = la.to_device(A), dB = la.to_device(B);
la.matmul2(dA, dB); la.synchronize();
const dev = best(() => { la.matmul2(dA, dB); la.synchronize(); }, 5);
const chainHost = best(() => la.lu_solve(la.lu_decomp(A), B), 2);
const chainDev = best(() => { la.lu_solve(la.lu_decomp(dA), dB); la.synchronize(); }, 3);
const flops = 2 * N * N * N;
const out = {N, matmul2_host_ms: host, matmul2_device_ms: dev, matmul2_host_tflops: flops / host / 1e9, matmul2_device_tflops: flops / dev / 1e9,
             lu_decomp_solve_host_ms: chainHost, lu_decomp_solve_device_ms: chainDev, node: process.version};
// BASELINE configs[4] through the production boundary: a batch of 512 x 512 SVDs from host Float64Arrays (H2D, kernels, D2H of U, sv, V)
// and from a DeviceNDArray (results stay on the device; only sv comes back)
const SB = parseInt(process.argv[3] || '0');
if (SB > 0) {
  const n = 512, X = fill(1000, [SB, n, n]);
  la.svd_decomp(fill(999, [8, n, n]));                                             // warm-up
  const hostMs = best(() => la.svd_decomp(X), 2);
  const dX = la.to_device(X);
  let sv = null;
  const devMs = best(() => { const r = la.svd_decomp(dX); sv = la.to_host(r[1]); la.synchronize(); }, 2);
  const f = 21 * n * n * n * SB;
  out.svd_batch = {batch: SB, n, host_arrays_ms: hostMs, device_arrays_ms: devMs, host_arrays_gflops_nominal: f / hostMs / 1e6,
                   device_arrays_gflops_nominal: f / devMs / 1e6, sv0: sv.data[0]};
}
console.log(JSON.stringify(out));
