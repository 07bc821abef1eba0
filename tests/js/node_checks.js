'use strict';
/* Node-side checks of the JS host wrapper + N-API addon. Driven by tests/test_node_addon.py.
 *   node node_checks.js cpu              (no GPU: validation, error text, loud failure)
 *   node node_checks.js gpu <golden dir> (GPU: results vs the reference-generated golden vectors)
 *   node node_checks.js install <path to reference dist/nd.js>   (drop-in patching, build container only)
 */
const fs = require('fs'), path = require('path');
const la = require(path.join(__dirname, '..', '..', 'nd4js_amd', 'js'));
const assert = require('assert');
const mode = process.argv[2];

function fmix32(h) { h ^= h >>> 16; h = Math.imul(h, 0x85ebca6b); h ^= h >>> 13; h = Math.imul(h, 0xc2b2ae35); h ^= h >>> 16; return h >>> 0; }
function uniform(seed, idx) {
  const hi = fmix32((idx ^ fmix32(seed >>> 0)) >>> 0), lo = fmix32((hi + 0x9E3779B9 + idx) >>> 0);
  return ((hi >>> 5) * 67108864 + (lo >>> 6)) * 2.220446049250313e-16 - 1.0;
}
function fill(seed, shape) {
  const n = shape.reduce((a, b) => a * b, 1), d = new Float64Array(n);
  for (let i = 0; i < n; i++) d[i] = uniform(seed, i);
  return new la.NDArray(Int32Array.from(shape), d);
}
function loadNpy(file) {
  const buf = fs.readFileSync(file), hlen = buf.readUInt16LE(8), hdr = buf.toString('latin1', 10, 10 + hlen);
  const descr = /'descr': '([^']+)'/.exec(hdr)[1];
  const shape = /'shape': \(([^)]*)\)/.exec(hdr)[1].split(',').map(s => s.trim()).filter(s => s).map(Number);
  const body = buf.slice(10 + hlen), ab = body.buffer.slice(body.byteOffset, body.byteOffset + body.byteLength);
  return {shape, data: descr === '<f8' ? new Float64Array(ab) : new Int32Array(ab)};
}
function relerr(x, ref) { let n = 0, d = 0; for (let i = 0; i < ref.length; i++) { d += (x[i] - ref[i]) ** 2; n += ref[i] ** 2; } return Math.sqrt(d / Math.max(n, 1e-300)); }

if (mode === 'cpu') {
  assert.throws(() => la.matmul2([1, 2, 3], [[1], [2], [3]]), /A must be at least 2D\./);
  assert.throws(() => la.matmul2([[1, 2, 3]], [1, 2, 3]), /B must be at least 2D\./);
  assert.throws(() => la.matmul2([[1, 2, 3]], [[1, 2], [3, 4]]), /do not match/);
  assert.throws(() => la.matmul2(fill(1, [2, 2, 3]), fill(2, [3, 3, 2])), /broadcast-compatible/);
  assert.throws(() => la.qr_decomp([1, 2, 3]), /at least 2/);
  assert.throws(() => la.lu_decomp([[1, 2, 3], [4, 5, 6]]), /quadratic/);
  assert.throws(() => la.triu_solve([[1, 2, 3], [4, 5, 6]], [[1], [2]]), /must be quadratic/);
  assert.throws(() => la.lu_solve([[1, 0], [0, 1]], new la.NDArray(Int32Array.of(2), Int32Array.of(0, 1)), [[1], [2], [3]]), /LU and y don't match/);
  assert.throws(() => la.qr_lstsq(fill(1, [4, 3]), fill(2, [3, 3]), fill(3, [5, 1])), /Q and y don't match/);
  assert.throws(() => la.qr_lstsq(fill(1, [3, 3]), fill(2, [3, 5]), fill(3, [3, 1])), /Under-determined/);
  assert.throws(() => la.svd_lstsq(fill(1, [4, 3]), fill(2, [2]), fill(3, [3, 3]), fill(4, [4, 1])), /U and sv don't match/);
  assert.throws(() => la.svd_lstsq(fill(1, [2, 2]), new la.NDArray(Int32Array.of(2), Float64Array.of(1, NaN)), fill(3, [2, 2]), fill(4, [2, 1])), /NaN or Infinity/);
  assert.throws(() => la.svd_solve(fill(1, [4, 3]), fill(2, [3]), fill(3, [3, 3]), fill(4, [4, 1])), /System not square/);
  assert.throws(() => la.cholesky_decomp(fill(1, [2, 3])), /must be quadratic/);
  assert.throws(() => la.hessenberg_decomp(fill(1, [2, 3])), /A must be square/);
  assert.throws(() => la.ldl_solve(fill(1, [3, 3]), fill(2, [4, 1])), /ldl_solve\(LD,y\): LD and y don't match/);
  assert.throws(() => la.cholesky_solve(fill(1, [3, 3]), fill(2, [4, 1])), /L and y don't match/);
  const g = la.bcastGroups([3, 4], [3, 1], [4], 35, 42);
  assert.deepStrictEqual(g.map(x => x.slice(0, 5)), [[4, 0, 0, 0, 42], [4, 35, 0, 0, 42], [4, 70, 0, 0, 42]]);
  assert.ok(/gfx950/.test(la.version()));
  if (la.device_count() === 0) assert.throws(() => la.to_device(fill(1, [4, 4])), /no HIP device/);
  assert.throws(() => require(path.join(path.dirname(require.resolve(process.argv[1])), '..', '..', 'nd4js_amd', 'js', 'nd4hip_napi.node')).dgemm_batched(
    1, 2, 2, 2, new Float64Array(4), 0, {b: {}, o: 0}, 0, new Float64Array(4)), /device buffer from dev_alloc/);
  if (la.device_count() === 0) {
    for (const f of [() => la.matmul2(fill(1, [4, 4]), fill(2, [4, 4])), () => la.qr_decomp(fill(1, [4, 4])),
                     () => la.lu_decomp(fill(1, [4, 4])), () => la.svd_decomp(fill(1, [4, 4]))])
      assert.throws(f, /no HIP device/);          // loud failure, no CPU fallback
  }
  assert.throws(() => la.matmul(fill(1, [2, 3]), fill(2, [4, 2]), fill(3, [2, 2])), /Shape mismatch\./);
  assert.throws(() => la.matmul(fill(1, [2, 2, 3]), fill(2, [3, 3, 2]), fill(3, [2, 2])), /broadcast-compatible/);
  { const one = fill(1, [3, 4]); assert.strictEqual(la.matmul(one), one); }
  { const r = la.svd_rank(new la.NDArray(Int32Array.of(3, 4), Float64Array.of(5, 3, 1e-9, 0,  2, 1, 0.5, 0.25,  1, 1e-20, NaN, 7)));      // svd.js:31-63
    assert.deepStrictEqual(Array.from(r.data), [2, 4, 1]); assert.deepStrictEqual(Array.from(r.shape), [3]);
    assert.throws(() => la.svd_rank(new la.NDArray(Int32Array.of(3), Float64Array.of(1, NaN, 0))), /NaN or Infinity/); }
  if (process.argv[3]) {            // chain plans for the golden chain shapes, compared by the Python test with la.chain_plan
    const man = JSON.parse(fs.readFileSync(path.join(process.argv[3], 'manifest.json'))).cases, plans = {};
    for (const [name, m] of Object.entries(man)) if (m.op === 'matmul' && m.shapes.length > 2) plans[name] = la._chain_plan(m.shapes);
    console.log('PLANS ' + JSON.stringify(plans));
  }
  console.log('node cpu checks ok');
}

if (mode === 'install') {
  const nd = require(process.argv[3]);
  const nd2 = la.install(nd);
  const a = new nd.NDArray(Int32Array.of(2, 2), Int32Array.of(1, 2, 3, 4));
  const c = nd2.la.matmul2(a, a);                  // int32 x int32 keeps going to the reference's own code
  assert.ok(c.data instanceof Int32Array && Array.from(c.data).join() === '7,10,15,22');
  assert.strictEqual(typeof nd2.la.det, 'function'); // everything else is still there
  if (la.device_count() === 0)
    assert.throws(() => nd2.la.matmul2(new nd.NDArray(Int32Array.of(2, 2), Float64Array.of(1, 2, 3, 4)), a), /no HIP device/);
  { // float32 operands go to the reference's matmul2 through the fallback: the chain ORDER is then the only thing that
    // can make la.matmul differ from the reference's matmul (float32 rounding after every product is order-sensitive)
    const f32 = (seed, shape) => { const x = fill(seed, shape); return new nd.NDArray(Int32Array.from(shape), Float32Array.from(x.data)); };
    for (const shapes of [[[30, 4], [4, 50], [50, 6]], [[5, 60], [60, 7], [7, 80]], [[40, 10], [10, 33], [33, 5], [5, 64]],
                          [[3, 1, 8, 20], [2, 20, 6], [6, 11]], [[2, 9, 14], [14, 14], [2, 14, 3], [3, 22], [1, 22, 5]], [[16, 16], [16, 16], [16, 16], [16, 16]]]) {
      const ms = shapes.map((sh, k) => f32(900 + k, sh)), mine = nd2.la.matmul(...ms), ref = nd.la.matmul(...ms);
      assert.ok(mine.data instanceof Float32Array); assert.deepStrictEqual(Array.from(mine.shape), Array.from(ref.shape));
      for (let i = 0; i < ref.data.length; i++) assert.ok(mine.data[i] === ref.data[i], 'chain order differs from the reference');
    }
  }
  { const svs = new nd.NDArray(Int32Array.of(2, 3), Float64Array.of(4, 2, 1e-12, 3, 3, 3));       // same ranks as the reference's svd_rank
    assert.deepStrictEqual(Array.from(nd2.la.svd_rank(svs).data), Array.from(nd2.la.__nd4hip_original__.svd_rank(svs).data)); }
  { // install(nd, {minWork}): tiny float64 calls go to the HOST MODULE's own functions (bit-identical to them, no GPU needed)
    delete require.cache[require.resolve(process.argv[3])];
    const ndB = require(process.argv[3]), ndC = la.install(ndB, {minWork: 1e5}), orig = ndC.la.__nd4hip_original__;
    const f64 = (seed, shape) => new ndB.NDArray(Int32Array.from(shape), Float64Array.from(fill(seed, shape).data));
    const x = f64(5, [3, 8, 8]);                                       // 3 * 8 * 8 * 8 = 1536 < minWork
    const [LU, P] = ndC.la.lu_decomp(x), [LU0, P0] = orig.lu_decomp(x);
    assert.deepStrictEqual(Array.from(LU.data), Array.from(LU0.data)); assert.deepStrictEqual(Array.from(P.data), Array.from(P0.data));
    const [U, sv, V] = ndC.la.svd_decomp(x), [U0, sv0, V0] = orig.svd_decomp(x);
    assert.deepStrictEqual(Array.from(sv.data), Array.from(sv0.data)); assert.deepStrictEqual(Array.from(U.data), Array.from(U0.data));
    assert.deepStrictEqual(Array.from(ndC.la.matmul2(x, x).data), Array.from(orig.matmul2(x, x).data));
    if (la.device_count() === 0)                                       // 64^3 = 262144 >= minWork: the accelerated path, which fails loudly here
      assert.throws(() => ndC.la.lu_decomp(f64(6, [64, 64])), /no HIP device/);
  }
  console.log('node install checks ok');
}

if (mode === 'gpu') {
  const dir = process.argv[3], man = JSON.parse(fs.readFileSync(path.join(dir, 'manifest.json'))).cases;
  const npy = (c, k) => loadNpy(path.join(dir, man[c].files[k]));
  { // nd4hip_profile_enable / nd4hip_profile_last through the N-API table
    la.profile_enable(true);
    la.matmul2(fill(1, [64, 64]), fill(2, [64, 64]));
    const pr = la.profile_last();
    assert.ok(pr.length >= 1 && pr[0].valid && pr[0].op === 'dgemm_batched' && pr[0].flops === 2 * 64 * 64 * 64 && pr[0].bytes === 8 * 3 * 64 * 64 && pr[0].kernel_ms > 0);
    la.lu_decomp(fill(3, [4, 96, 96]));
    const pl = la.profile_last();
    assert.ok(pl[0].op === 'dgetrf_batched' && Math.abs(pl[0].flops - 4 * 2 / 3 * 96 ** 3) < 1 && pl[0].kernel_ms > 0);
    la.profile_enable(false);
    assert.ok(!la.profile_last()[0].valid);
  }
  { const m = man.c1_matmul64, C = la.matmul2(fill(m.seedA, m.shapeA), fill(m.seedB, m.shapeB));
    assert.ok(relerr(C.data, npy('c1_matmul64', 'C').data) <= 1e-13); }
  { const m = man.bc_matmul_a, C = la.matmul2(fill(m.seedA, m.shapeA), fill(m.seedB, m.shapeB));
    assert.deepStrictEqual(Array.from(C.shape), m.shapeC); assert.ok(relerr(C.data, npy('bc_matmul_a', 'C').data) <= 1e-13); }
  { const m = man.c1_qr32, [Q, R] = la.qr_decomp(fill(m.seed, m.shape));
    assert.ok(relerr(Q.data, npy('c1_qr32', 'Q').data) <= 1e-12 && relerr(R.data, npy('c1_qr32', 'R').data) <= 1e-12); }
  { const m = man.mid_lu96, [LU, P] = la.lu_decomp(fill(m.seed, m.shape));
    assert.deepStrictEqual(Array.from(P.data), Array.from(npy('mid_lu96', 'P').data));
    assert.ok(P.data instanceof Int32Array && relerr(LU.data, npy('mid_lu96', 'LU').data) <= 1e-12); }
  { const m = man.mid_svd96, [U, sv, V] = la.svd_decomp(fill(m.seed, m.shape)), ref = npy('mid_svd96', 'sv').data;
    let d = 0; for (let i = 0; i < ref.length; i++) d = Math.max(d, Math.abs(sv.data[i] - ref[i]));
    assert.ok(d <= 1e-12 * ref[0]); assert.deepStrictEqual(Array.from(U.shape), [96, 96]); assert.deepStrictEqual(Array.from(V.shape), [96, 96]);
    assert.ok(la.last_svd_info.sweeps > 0 && la.last_svd_info.rotations > 0 && la.last_svd_info.offnorm <= 96 * 2.3e-16);
    assert.deepStrictEqual(Array.from(la.svd_rank(sv).data), [96]); assert.deepStrictEqual(la.devices(), [0]); }
  { const m = man.solve_lu_bcast, A = fill(m.seedA, m.shapeA), Y = fill(m.seedY, m.shapeY);
    const X = la.lu_solve(la.lu_decomp(A), Y), ref = npy('solve_lu_bcast', 'X');
    assert.deepStrictEqual(Array.from(X.shape), ref.shape); assert.ok(relerr(X.data, ref.data) <= 1e-11); }
  { const m = man.solve_tril_bcast, T = fill(m.seedT, m.shapeT), M = m.shapeT[m.shapeT.length - 1];
    for (let i = 0; i < M; i++) for (let j = 0; j < M; j++) { const k = i * M + j;
      if (i === j) T.data[k] += T.data[k] >= 0 ? 2 : -2; else if (j > i) T.data[k] = 0; else T.data[k] *= 0.25; }
    const X = la.tril_solve(T, fill(m.seedY, m.shapeY)), ref = npy('solve_tril_bcast', 'X');
    assert.deepStrictEqual(Array.from(X.shape), ref.shape); assert.ok(relerr(X.data, ref.data) <= 1e-12); }
  { const m = man.lstsq_qr_bcast, X = la.qr_lstsq(la.qr_decomp(fill(m.seedA, m.shapeA)), fill(m.seedY, m.shapeY)), ref = npy('lstsq_qr_bcast', 'X');
    assert.deepStrictEqual(Array.from(X.shape), ref.shape); assert.ok(relerr(X.data, ref.data) <= 1e-11); }
  { const m = man.lstsq_svd_rankdef, n = k => { const a = npy('lstsq_svd_rankdef', k); return new la.NDArray(Int32Array.from(a.shape), a.data); };
    const X = la.svd_lstsq(n('U'), n('sv'), n('V'), fill(m.seedY, m.shapeY)), ref = npy('lstsq_svd_rankdef', 'X');
    assert.deepStrictEqual(Array.from(X.shape), ref.shape); assert.ok(relerr(X.data, ref.data) <= 1e-12); }
  /* ---- device-resident arrays (SURVEY §8f N3): same kernels, nothing crosses PCIe between calls ---- */
  { const same = (x, y) => { assert.strictEqual(x.length, y.length); for (let i = 0; i < x.length; i++) assert.ok(Object.is(x[i], y[i]) || x[i] === y[i], `differs at ${i}`); };
    const A = fill(11, [96, 96]), B = fill(12, [96, 40]), dA = la.to_device(A), dB = la.to_device(B);
    const dC = la.matmul2(dA, dB);
    assert.ok(dC instanceof la.DeviceNDArray && dC.onDevice && dC.dtype === 'float64'); assert.deepStrictEqual(Array.from(dC.shape), [96, 40]);
    same(dC.data, la.matmul2(A, B).data);                                 // lazy D2H; identical kernels -> identical bits
    assert.strictEqual(dC.data, dC.data);                                  // cached host copy
    same(la.matmul2(dA, B).data, dC.data); same(la.matmul2(A, dB).data, dC.data);     // mixed residency: host operand uploaded on the way in
    const [dLU, dP] = la.lu_decomp(dA), [LU, P] = la.lu_decomp(A);
    assert.ok(dLU.onDevice && dP.onDevice && dP.dtype === 'int32'); same(dLU.data, LU.data); same(dP.data, P.data);
    const dX = la.lu_solve(dLU, dP, dB); assert.ok(dX.onDevice); same(dX.data, la.lu_solve(LU, P, B).data);
    same(la.lu_solve([dLU, dP], B).data, dX.data);
    const [dQ, dR] = la.qr_decomp(dA), [Q, R] = la.qr_decomp(A); same(dQ.data, Q.data); same(dR.data, R.data);
    same(la.qr_lstsq(dQ, dR, dB).data, la.qr_lstsq(Q, R, B).data);
    same(la.triu_solve(dR, dB).data, la.triu_solve(R, B).data);
    const dS = la.svd_decomp(dA), S = la.svd_decomp(A); for (let k = 0; k < 3; k++) same(dS[k].data, S[k].data);
    same(la.svd_lstsq(dS, dB).data, la.svd_lstsq(S, B).data); same(la.svd_solve(dS, dB).data, la.svd_solve(S, B).data);
    const chainD = la.matmul(dA, dB, la.to_device(fill(13, [40, 3]))), chainH = la.matmul(A, B, fill(13, [40, 3]));
    assert.ok(chainD.onDevice); same(chainD.data, chainH.data);
    const bA = la.to_device(fill(14, [3, 1, 8, 5])), bB = fill(15, [2, 5, 4]);           // broadcast groups with device views at offsets
    same(la.matmul2(bA, bB).data, la.matmul2(la.to_host(bA), bB).data);
    const h = la.to_host(dC); assert.ok(h instanceof la.NDArray && h.data === dC.data);
    dC.dispose(); same(dC.data, h.data);                                   // the cached host copy survives
    const dD = la.matmul2(dA, dB); dD.dispose(); assert.throws(() => dD.data, /disposed/); assert.throws(() => la.matmul2(dA, dD), /disposed|freed/);
    la.synchronize(); }
  { const m = man.inplace_qr_20x45, A = fill(m.seedA, m.shapeA), Y = fill(m.seedY, [m.shapeA[0], m.L]), refR = npy('inplace_qr_20x45', 'R'), refY = npy('inplace_qr_20x45', 'QtY');
    const [Qf, Rf] = la.qr_decomp_full(A); assert.deepStrictEqual(Array.from(Qf.shape), [20, 20]); assert.ok(relerr(Rf.data, refR.data) <= 1e-12);
    const a = Float64Array.from(A.data), y = new Float64Array(3 + Y.data.length); y.set(Y.data, 3);       // with an offset, like the TODO at qr_test.js:221
    la._qr_decomp_inplace(20, 45, m.L, a, 0, y, 3);
    assert.ok(relerr(a, refR.data) <= 1e-12 && relerr(y.subarray(3), refY.data) <= 1e-11 && y[0] === 0 && y[2] === 0);
    assert.throws(() => la._qr_decomp_inplace(20, 45, m.L, a, 1, y, 3), /Assertion failed/); }
  { const m = man.chol_bcast_y, N = m.shape[0], B = fill(m.seed, m.shape), S = la.matmul2(B, B.T);
    for (let i = 0; i < N; i++) S.data[i * N + i] += N;
    const L = la.cholesky_decomp(S), X = la.cholesky_solve(L, fill(m.seedY, m.shapeY)), refL = npy('chol_bcast_y', 'L'), refX = npy('chol_bcast_y', 'X');
    assert.ok(relerr(L.data, refL.data) <= 1e-13); assert.deepStrictEqual(Array.from(X.shape), refX.shape); assert.ok(relerr(X.data, refX.data) <= 1e-12);
    const dL = la.cholesky_decomp(la.to_device(S)); assert.ok(dL.onDevice); for (let i = 0; i < L.data.length; i++) assert.ok(L.data[i] === dL.data[i]);
    assert.throws(() => la.cholesky_decomp([[1, 2], [2, 1]]), /Matrix contains NaNs or is \(near\) singular\./); }
  { const m = man.ldl_bcast_y, refLD = npy('ldl_bcast_y', 'LD'), refX = npy('ldl_bcast_y', 'X'), N = m.shape[0];
    const L = new la.NDArray(Int32Array.of(N, N), Float64Array.from(refLD.data, (v, k) => (k % N === (k / N | 0)) ? 1 : v));
    const Dm = new la.NDArray(Int32Array.of(N, N), Float64Array.from(refLD.data, (v, k) => (k % N === (k / N | 0)) ? v : 0));
    const S = la.matmul(L, Dm, L.T), LD = la.ldl_decomp(S);                   // S rebuilt from the reference's factors
    assert.ok(relerr(LD.data, refLD.data) <= 1e-12);
    const X = la.ldl_solve(new la.NDArray(Int32Array.of(N, N), refLD.data), fill(m.seedY, m.shapeY));
    assert.deepStrictEqual(Array.from(X.shape), refX.shape); assert.ok(relerr(X.data, refX.data) <= 1e-12); }
  { const m = man.hess_17, [U, H] = la.hessenberg_decomp(fill(m.seed, m.shape));
    assert.ok(relerr(U.data, npy('hess_17', 'U').data) <= 1e-12 && relerr(H.data, npy('hess_17', 'H').data) <= 1e-12); }
  { for (const c of ['bidiag_sq_17', 'bidiag_vert_20x7', 'bidiag_horiz_7x20']) { const m = man[c], [U, B, V] = la.bidiag_decomp(fill(m.seed, m.shape));
      for (const [x, k] of [[U, 'U'], [B, 'B'], [V, 'V']]) { const ref = npy(c, k); assert.deepStrictEqual(Array.from(x.shape), ref.shape); assert.ok(relerr(x.data, ref.data) <= 1e-11, c + ' ' + k); } } }
  for (const c of ['chain_4_bcast', 'chain_5', 'chain_hand_1x4_4x3_3x2']) {    // matmul(...ms) against the reference's own results
    const m = man[c], ref = npy(c, 'C');
    const ms = m.hand ? m.shapes.map((sh, k) => { const x = npy(c, 'M' + k); return new la.NDArray(Int32Array.from(x.shape), x.data); }) : m.shapes.map((sh, k) => fill(m.seed0 + k, sh));
    const C = la.matmul(...ms); assert.deepStrictEqual(Array.from(C.shape), ref.shape); assert.ok(relerr(C.data, ref.data) <= 1e-13, c); }
  console.log('node gpu checks ok');
}
