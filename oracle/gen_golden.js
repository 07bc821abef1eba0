#!/usr/bin/env node
/* Golden-vector generator (TEST INFRASTRUCTURE — runs only in the build container).
 *
 * Loads the real reference bundle (/root/reference/dist/nd.js, webpack build of
 * src/help.js -> src/la/*.js), feeds it inputs from the repo's counter-based
 * generator `nd4_uniform(seed, idx)` (bit-identical twins: oracle/nd4_oracle.c,
 * nd4js_amd/rng.py, nd4js_amd/csrc/fill.hip) and stores ONLY numbers (outputs,
 * sampled entries, norms) as .npy files + manifest.json under tests/golden/.
 * Nothing of the reference's source is written anywhere.
 *
 *   node oracle/gen_golden.js small            # C1 + mid + edge families (seconds)
 *   node oracle/gen_golden.js c2               # 4096^2 matmul samples   (~2 min)
 *   node oracle/gen_golden.js c3               # 2048^2 QR + LU samples  (~20 s)
 *   node oracle/gen_golden.js c4               # 2048^2 SVD sv + samples (~70 s)
 *   node oracle/gen_golden.js c5 [stride]      # 512^2 SVDs of every `stride`-th batch member
 *   node oracle/gen_golden.js chain            # matmul(...ms) chains: hand cases of matmul_test.js:32-79 + seeded 3-5 operand chains
 *   node oracle/gen_golden.js c3b              # 4096^2 QR + LU (rows beyond the 2048-row register panels) (~3 min)
 */
'use strict';
const fs = require('fs'), path = require('path');
const REF = process.env.ND4_REFERENCE || '/root/reference/dist/nd.js';
const nd = require(REF);
const OUT = path.join(__dirname, '..', 'tests', 'golden');
fs.mkdirSync(OUT, {recursive: true});

/* ---------- counter-based generator (32-bit ops only) ---------- */
function fmix32(h) {
  h ^= h >>> 16; h = Math.imul(h, 0x85ebca6b);
  h ^= h >>> 13; h = Math.imul(h, 0xc2b2ae35);
  h ^= h >>> 16; return h >>> 0;
}
function nd4_uniform(seed, idx) {            // -> [-1, 1), multiple of 2^-52
  const hi = fmix32((idx ^ fmix32(seed >>> 0)) >>> 0);
  const lo = fmix32((hi + 0x9E3779B9 + idx) >>> 0);
  const m = (hi >>> 5) * 67108864 + (lo >>> 6);      // 27 + 26 = 53 bits
  return m * 2.220446049250313e-16 - 1.0;            // 2*m*2^-53 - 1
}
function fill(seed, n, off = 0) {
  const a = new Float64Array(n);
  for (let i = 0; i < n; i++) a[i] = nd4_uniform(seed, off + i);
  return a;
}
function hashIdx(seed, i, mod) { return fmix32((fmix32(seed) + Math.imul(i, 0x9E3779B1)) >>> 0) % mod; }

/* ---------- npy writer ---------- */
function npy(name, typed, shape) {
  const descr = typed instanceof Float64Array ? '<f8' : typed instanceof Int32Array ? '<i4' : null;
  if (!descr) throw new Error('dtype');
  let hdr = `{'descr': '${descr}', 'fortran_order': False, 'shape': (${shape.join(', ')}${shape.length === 1 ? ',' : ''}), }`;
  const pad = 64 - ((10 + hdr.length + 1) % 64);
  hdr += ' '.repeat(pad % 64) + '\n';
  const head = Buffer.alloc(10);
  head.write('\x93NUMPY', 0, 'latin1'); head[6] = 1; head[7] = 0; head.writeUInt16LE(hdr.length, 8);
  const body = Buffer.from(typed.buffer, typed.byteOffset, typed.byteLength);
  fs.writeFileSync(path.join(OUT, name + '.npy'), Buffer.concat([head, Buffer.from(hdr, 'latin1'), body]));
}
const NDA = (shape, data) => new nd.NDArray(Int32Array.from(shape), data);

const manifestPath = path.join(OUT, 'manifest.json');
const manifest = fs.existsSync(manifestPath) ? JSON.parse(fs.readFileSync(manifestPath)) : {rng: 'fmix32-v1', cases: {}};
function record(name, meta, tensors) {
  const files = {};
  for (const [k, [typed, shape]] of Object.entries(tensors)) { npy(`${name}.${k}`, typed, shape); files[k] = `${name}.${k}.npy`; }
  manifest.cases[name] = Object.assign({}, meta, {files});
  fs.writeFileSync(manifestPath, JSON.stringify(manifest, null, 1));
  console.log('wrote', name);
}
function fro(a) { let s = 0; for (let i = 0; i < a.length; i++) s += a[i] * a[i]; return Math.sqrt(s); }

/* ---------- input families (mirroring the reference tests' families) ---------- */
function applyFamily(fam, a, M, N, seed) {
  switch (fam) {
    case 'dense': break;
    case 'sparse10':                       // ~10 % exact zeros (qr_test.js:93-104 family)
      for (let i = 0; i < a.length; i++) if (hashIdx(seed + 77, i, 10) === 0) a[i] = 0; break;
    case 'zerorow': { const r = hashIdx(seed + 78, 0, M); for (let j = 0; j < N; j++) a[r * N + j] = 0; break; }
    case 'zerocol': { const c = hashIdx(seed + 79, 0, N); for (let i = 0; i < M; i++) a[i * N + c] = 0; break; }
    case 'rankdef': {                      // rows r >= rank are copies/combinations of the first rows
      const rank = Math.max(1, Math.min(M, N) >> 1);
      for (let i = rank; i < M; i++) for (let j = 0; j < N; j++)
        a[i * N + j] = 0.5 * a[((i - rank) % rank) * N + j] - 0.25 * a[((i + 1) % rank) * N + j];
      break; }
    case 'diag': for (let i = 0; i < M; i++) for (let j = 0; j < N; j++) if (i !== j) a[i * N + j] = 0; break;
    case 'triu': for (let i = 0; i < M; i++) for (let j = 0; j < i && j < N; j++) a[i * N + j] = 0; break;
    default: throw new Error(fam);
  }
  return a;
}

function caseMatmul(name, seedA, seedB, shapeA, shapeB) {
  const nA = shapeA.reduce((a, b) => a * b, 1), nB = shapeB.reduce((a, b) => a * b, 1);
  const C = nd.la.matmul2(NDA(shapeA, fill(seedA, nA)), NDA(shapeB, fill(seedB, nB)));
  record(name, {op: 'matmul2', seedA, seedB, shapeA, shapeB, shapeC: Array.from(C.shape)}, {C: [C.data, Array.from(C.shape)]});
}
function caseQR(name, seed, shape, fam = 'dense', fn = 'qr_decomp') {
  const M = shape[shape.length - 2], N = shape[shape.length - 1], n = shape.reduce((a, b) => a * b, 1);
  const a = fill(seed, n);
  for (let o = 0, b = 0; o < n; o += M * N, b++) applyFamily(fam, a.subarray(o, o + M * N), M, N, seed + b);
  const [Q, R] = nd.la[fn](NDA(shape, a));
  record(name, {op: fn, seed, shape, family: fam}, {Q: [Q.data, Array.from(Q.shape)], R: [R.data, Array.from(R.shape)]});
}
function caseLU(name, seed, shape, fam = 'dense') {
  const N = shape[shape.length - 1], n = shape.reduce((a, b) => a * b, 1);
  const a = fill(seed, n);
  for (let o = 0, b = 0; o < n; o += N * N, b++) applyFamily(fam, a.subarray(o, o + N * N), N, N, seed + b);
  const [LU, P] = nd.la.lu_decomp(NDA(shape, a));
  record(name, {op: 'lu_decomp', seed, shape, family: fam}, {LU: [LU.data, Array.from(LU.shape)], P: [P.data, Array.from(P.shape)]});
}
function caseSVD(name, seed, shape, fam = 'dense', fn = 'svd_decomp') {
  const M = shape[shape.length - 2], N = shape[shape.length - 1], n = shape.reduce((a, b) => a * b, 1);
  const a = fill(seed, n);
  for (let o = 0, b = 0; o < n; o += M * N, b++) applyFamily(fam, a.subarray(o, o + M * N), M, N, seed + b);
  const [U, sv, V] = nd.la[fn](NDA(shape, a));
  record(name, {op: fn, seed, shape, family: fam},
    {U: [U.data, Array.from(U.shape)], sv: [sv.data, Array.from(sv.shape)], V: [V.data, Array.from(V.shape)]});
}
function sample(typed, n, seed) {
  const idx = new Int32Array(n), val = new Float64Array(n);
  for (let i = 0; i < n; i++) { idx[i] = hashIdx(seed, i, typed.length); val[i] = typed[idx[i]]; }
  return [idx, val];
}

const what = process.argv[2] || 'small';
const t0 = Date.now();

if (what === 'small') {
  /* generator self-check vectors */
  record('rng', {op: 'rng', seed: 12345, n: 64, offset: 1000}, {u: [fill(12345, 64, 1000), [64]]});
  /* C1: 64^2 matmul + 32^2 QR (BASELINE.json configs[0]) and 32^2 of the other ops */
  caseMatmul('c1_matmul64', 1, 2, [64, 64], [64, 64]);
  caseQR('c1_qr32', 3, [32, 32]);
  caseQR('c1_qrfull32', 3, [32, 32], 'dense', 'qr_decomp_full');
  caseLU('c1_lu32', 4, [32, 32]);
  caseSVD('c1_svd32', 8, [32, 32]);
  caseSVD('c1_svdjac32', 8, [32, 32], 'dense', 'svd_jac_2sided');
  /* mid sizes, deliberately not tile multiples */
  caseMatmul('mid_matmul', 11, 12, [96, 80], [80, 112]);
  caseMatmul('mid_matmul_sq200', 13, 14, [200, 200], [200, 200]);
  caseQR('mid_qr96', 15, [96, 96]);
  caseQR('mid_qr_wide', 16, [40, 72]);
  caseQR('mid_qr_tall', 17, [72, 40]);
  caseQR('mid_qrfull_tall', 17, [24, 10], 'dense', 'qr_decomp_full');
  caseQR('mid_qr130', 18, [130, 130]);
  caseLU('mid_lu96', 19, [96, 96]);
  caseLU('mid_lu130', 20, [130, 130]);
  caseSVD('mid_svd96', 21, [96, 96]);
  caseSVD('mid_svdjac48', 22, [48, 48], 'dense', 'svd_jac_2sided');
  caseSVD('mid_svd_wide', 23, [24, 40]);
  caseSVD('mid_svd_tall', 24, [40, 24]);
  /* batched + broadcast matmul shapes (matmul_test.js:86-118 style) */
  caseMatmul('bc_matmul_a', 31, 32, [3, 1, 5, 7], [4, 7, 6]);
  caseMatmul('bc_matmul_b', 33, 34, [2, 3, 4, 5], [5, 2]);
  caseMatmul('bc_matmul_c', 35, 36, [6, 9], [2, 1, 3, 9, 4]);
  caseMatmul('bc_matmul_vec', 37, 38, [1, 13], [13, 1]);
  caseMatmul('bc_matmul_batch', 39, 40, [5, 33, 17], [5, 17, 29]);
  /* batched decompositions */
  caseQR('b_qr', 41, [3, 2, 12, 12]);
  caseLU('b_lu', 42, [4, 17, 17]);
  caseSVD('b_svd', 43, [3, 20, 20]);
  /* edge families (qr_test.js:67-146, lu_test.js:82-94, _generic_test_svd_decomp.js:180-336) */
  for (const fam of ['sparse10', 'zerorow', 'zerocol', 'rankdef', 'diag', 'triu']) {
    caseQR('edge_qr_' + fam, 50, [2, 16, 16], fam);
    caseSVD('edge_svdjac_' + fam, 52, [2, 16, 16], fam, 'svd_jac_2sided');
  }
  for (const fam of ['sparse10', 'diag', 'triu']) caseLU('edge_lu_' + fam, 51, [2, 16, 16], fam);
  for (const fam of ['sparse10', 'diag']) caseSVD('edge_svd_' + fam, 52, [2, 16, 16], fam);
  caseQR('edge_qr_1x1', 53, [1, 1]); caseLU('edge_lu_1x1', 54, [1, 1]); caseSVD('edge_svd_1x1', 55, [3, 1, 1]);
  caseMatmul('edge_matmul_1x1', 56, 57, [1, 1], [1, 1]);
}

if (what === 'solve') {
  /* SURVEY §8f N1: the solve-side consumers of the path (lu.js:84-177, tri.js:155-290) */
  const triangle = (seed, shape, upper) => {      // well-conditioned triangle: off-diagonal / 4, |diag| in [2, 3)
    const M = shape[shape.length - 1], a = fill(seed, shape.reduce((x, y) => x * y, 1));
    for (let o = 0; o < a.length; o += M * M)
      for (let i = 0; i < M; i++) for (let j = 0; j < M; j++) {
        const k = o + i * M + j;
        if (i === j) a[k] += a[k] >= 0 ? 2 : -2;
        else if (upper ? j < i : j > i) a[k] = 0; else a[k] *= 0.25;
      }
    return a;
  };
  const caseLuSolve = (name, seedA, shapeA, seedY, shapeY) => {
    const [LU, P] = nd.la.lu_decomp(NDA(shapeA, fill(seedA, shapeA.reduce((x, y) => x * y, 1))));
    const X = nd.la.lu_solve(LU, P, NDA(shapeY, fill(seedY, shapeY.reduce((x, y) => x * y, 1))));
    record(name, {op: 'lu_solve', seedA, shapeA, seedY, shapeY}, {X: [X.data, Array.from(X.shape)]});
  };
  const caseTri = (name, fn, seedT, shapeT, seedY, shapeY) => {
    const X = nd.la[fn](NDA(shapeT, triangle(seedT, shapeT, fn === 'triu_solve')), NDA(shapeY, fill(seedY, shapeY.reduce((x, y) => x * y, 1))));
    record(name, {op: fn, seedT, shapeT, seedY, shapeY}, {X: [X.data, Array.from(X.shape)]});
  };
  caseLuSolve('solve_lu_40', 61, [40, 40], 62, [40, 7]);
  caseLuSolve('solve_lu_bcast', 63, [3, 12, 12], 64, [12, 5]);
  caseLuSolve('solve_lu_batch', 65, [2, 33, 33], 66, [2, 33, 1]);
  caseLuSolve('solve_lu_1x1', 67, [1, 1], 68, [1, 3]);
  caseLuSolve('solve_lu_130', 69, [130, 130], 70, [130, 130]);
  caseTri('solve_triu_30', 'triu_solve', 71, [30, 30], 72, [30, 4]);
  caseTri('solve_tril_30', 'tril_solve', 73, [30, 30], 74, [30, 4]);
  caseTri('solve_triu_bcast', 'triu_solve', 75, [2, 10, 10], 76, [10, 3]);
  caseTri('solve_tril_bcast', 'tril_solve', 77, [17, 17], 78, [3, 17, 2]);
  caseTri('solve_triu_100', 'triu_solve', 79, [100, 100], 80, [100, 65]);
  caseTri('solve_tril_100', 'tril_solve', 81, [100, 100], 82, [100, 65]);
}

if (what === 'lstsq') {
  /* SURVEY §8f N1: qr_lstsq (qr.js:186-273) and svd_lstsq / svd_solve (svd.js:66-228). qr cases are keyed by seeds
     (the oracle's qr_decomp is bit-identical); svd cases carry the reference's own U, sv, V as inputs. */
  const numel = sh => sh.reduce((x, y) => x * y, 1);
  const caseQrLs = (name, seedA, shapeA, seedY, shapeY, full) => {
    const [Q, R] = nd.la[full ? 'qr_decomp_full' : 'qr_decomp'](NDA(shapeA, fill(seedA, numel(shapeA))));
    const X = nd.la.qr_lstsq(Q, R, NDA(shapeY, fill(seedY, numel(shapeY))));
    record(name, {op: 'qr_lstsq', seedA, shapeA, seedY, shapeY, full: !!full}, {X: [X.data, Array.from(X.shape)]});
  };
  const caseSvdLs = (name, fn, seedA, shapeA, seedY, shapeY, rankDef) => {
    const a = fill(seedA, numel(shapeA)), [M, N] = shapeA.slice(-2);
    if (rankDef) for (let o = 0; o < a.length; o += M * N) for (let i = 0; i < M; i++) a[o + i * N + N - 1] = 2 * a[o + i * N];  // last column = 2 x first
    const [U, sv, V] = nd.la.svd_decomp(NDA(shapeA, a));
    const X = nd.la[fn](U, sv, V, NDA(shapeY, fill(seedY, numel(shapeY))));
    record(name, {op: fn, seedA, shapeA, seedY, shapeY, rankDef: !!rankDef},
           {U: [U.data, Array.from(U.shape)], sv: [sv.data, Array.from(sv.shape)], V: [V.data, Array.from(V.shape)], X: [X.data, Array.from(X.shape)]});
  };
  caseQrLs('lstsq_qr_40x40', 91, [40, 40], 92, [40, 3]);
  caseQrLs('lstsq_qr_70x30', 93, [70, 30], 94, [70, 5]);
  caseQrLs('lstsq_qr_full_50x20', 95, [50, 20], 96, [50, 2], true);
  caseQrLs('lstsq_qr_bcast', 97, [3, 24, 10], 98, [24, 4]);
  caseQrLs('lstsq_qr_130x100', 99, [130, 100], 100, [130, 70]);
  caseQrLs('lstsq_qr_1x1', 101, [1, 1], 102, [1, 2]);
  caseSvdLs('lstsq_svd_40x40', 'svd_lstsq', 103, [40, 40], 104, [40, 3]);
  caseSvdLs('lstsq_svd_60x25', 'svd_lstsq', 105, [60, 25], 106, [60, 4]);
  caseSvdLs('lstsq_svd_25x60', 'svd_lstsq', 107, [25, 60], 108, [25, 4]);
  caseSvdLs('lstsq_svd_rankdef', 'svd_lstsq', 109, [30, 12], 110, [30, 2], true);
  caseSvdLs('lstsq_svd_bcast', 'svd_lstsq', 111, [2, 20, 8], 112, [20, 3]);
  caseSvdLs('lstsq_svd_100x70', 'svd_lstsq', 113, [100, 70], 114, [100, 66]);
  caseSvdLs('solve_svd_32', 'svd_solve', 115, [32, 32], 116, [32, 5]);
  caseSvdLs('solve_svd_rankdef', 'svd_solve', 117, [16, 16], 118, [16, 2], true);   // the reference returns (its rank loop never runs)
}

if (what === 'inplace') {
  /* SURVEY §8f N2: qr_decomp_full for every shape (qr.js:27-77) and _qr_decomp_inplace (qr.js:146-183). The bundle does
     not export _qr_decomp_inplace; the expected values are the ones the reference's own test compares it with
     (qr_test.js:213-225): R of qr_decomp_full and matmul2(Q.T, Y). */
  const numel = sh => sh.reduce((x, y) => x * y, 1);
  const zeros10 = (a, seed) => { const m = fill(seed, a.length); for (let i = 0; i < a.length; i++) if (m[i] > 0.8) a[i] = 0; return a; };
  const caseInplace = (name, seedA, shapeA, seedY, L, sparse) => {
    const [M, N] = shapeA, a = fill(seedA, M * N);
    if (sparse) zeros10(a, seedA + 1000);
    const Y = NDA([M, L], fill(seedY, M * L));
    const [Q, R] = nd.la.qr_decomp_full(NDA(shapeA, a));
    const QtY = nd.la.matmul2(Q.T, Y);
    record(name, {op: 'qr_decomp_inplace', seedA, shapeA, seedY, L, sparse: !!sparse},
           {Q: [Q.data, Array.from(Q.shape)], R: [R.data, Array.from(R.shape)], QtY: [QtY.data, Array.from(QtY.shape)]});
  };
  caseInplace('inplace_qr_24x24', 121, [24, 24], 122, 3);
  caseInplace('inplace_qr_20x45', 123, [20, 45], 124, 7);
  caseInplace('inplace_qr_45x20', 125, [45, 20], 126, 5);
  caseInplace('inplace_qr_sparse_33x33', 127, [33, 33], 128, 1, true);
  caseInplace('inplace_qr_1x1', 129, [1, 1], 130, 2);
  caseInplace('inplace_qr_70x70', 131, [70, 70], 132, 40);
  caseInplace('inplace_qr_100x37', 133, [100, 37], 134, 9);
}

if (what === 'chol') {
  /* SURVEY §8f N4: cholesky_decomp / cholesky_solve (cholesky.js:51-150). SPD input S = B B^T + N I from the seeded B,
     formed here with the reference's own matmul2 and committed only through the seed (the oracle's matmul2 is bit-identical). */
  const numel = sh => sh.reduce((x, y) => x * y, 1);
  const spd = (seed, shape) => {
    const N = shape[shape.length - 1], B = NDA(shape, fill(seed, numel(shape))), S = nd.la.matmul2(B, B.T);
    for (let o = 0; o < S.data.length; o += N * N) for (let i = 0; i < N; i++) S.data[o + i * N + i] += N;
    return S;
  };
  const caseChol = (name, seed, shape, seedY, shapeY) => {
    const S = spd(seed, shape), L = nd.la.cholesky_decomp(S);
    const t = {L: [L.data, Array.from(L.shape)]}, meta = {op: 'cholesky_decomp', seed, shape};
    if (seedY !== undefined) {
      const X = nd.la.cholesky_solve(L, NDA(shapeY, fill(seedY, numel(shapeY))));
      t.X = [X.data, Array.from(X.shape)]; meta.seedY = seedY; meta.shapeY = shapeY;
    }
    record(name, meta, t);
  };
  caseChol('chol_1x1', 141, [1, 1], 142, [1, 3]);
  caseChol('chol_32', 143, [32, 32], 144, [32, 5]);
  caseChol('chol_33', 145, [33, 33], 146, [33, 1]);
  caseChol('chol_batch', 147, [3, 2, 20, 20], 148, [2, 20, 4]);
  caseChol('chol_bcast_y', 149, [40, 40], 150, [3, 40, 2]);
  caseChol('chol_100', 151, [100, 100], 152, [100, 70]);
  caseChol('chol_257', 153, [257, 257]);
}

if (what === 'ldl') {
  /* SURVEY §8f N4: ldl_decomp / ldl_solve (ldl.js:47-201). Symmetric INDEFINITE input with safe pivots, built like the
     reference's own test (ldl_test.js:44-50): S = L D L^T from a seeded unit-lower L (entries / 4) and D_i = +-(1 + |u|). */
  const numel = sh => sh.reduce((x, y) => x * y, 1);
  const sym = (seed, shape) => {
    const N = shape[shape.length - 1], r = fill(seed, numel(shape)), Ld = new Float64Array(r.length), Dd = new Float64Array(r.length);
    for (let o = 0; o < r.length; o += N * N)
      for (let i = 0; i < N; i++) for (let j = 0; j < N; j++) {
        const k = o + i * N + j;
        Ld[k] = i === j ? 1 : j < i ? r[k] * 0.25 : 0;
        Dd[k] = i === j ? (r[k] >= 0 ? 1 + r[k] : -1 + r[k]) : 0;
      }
    const L = NDA(shape, Ld);
    return nd.la.matmul2(nd.la.matmul2(L, NDA(shape, Dd)), L.T);
  };
  const caseLdl = (name, seed, shape, seedY, shapeY) => {
    const LD = nd.la.ldl_decomp(sym(seed, shape)), t = {LD: [LD.data, Array.from(LD.shape)]}, meta = {op: 'ldl_decomp', seed, shape};
    if (seedY !== undefined) {
      const X = nd.la.ldl_solve(LD, NDA(shapeY, fill(seedY, numel(shapeY))));
      t.X = [X.data, Array.from(X.shape)]; meta.seedY = seedY; meta.shapeY = shapeY;
    }
    record(name, meta, t);
  };
  caseLdl('ldl_1x1', 161, [1, 1], 162, [1, 2]);
  caseLdl('ldl_32', 163, [32, 32], 164, [32, 5]);
  caseLdl('ldl_33', 165, [33, 33], 166, [33, 1]);
  caseLdl('ldl_batch', 167, [2, 3, 20, 20], 168, [3, 20, 4]);
  caseLdl('ldl_bcast_y', 169, [40, 40], 170, [2, 40, 3]);
  caseLdl('ldl_100', 171, [100, 100], 172, [100, 70]);
  caseLdl('ldl_200', 173, [200, 200]);
}

if (what === 'hess') {
  /* SURVEY §8f N4: hessenberg_decomp (hessenberg.js:27-115) */
  const numel = sh => sh.reduce((x, y) => x * y, 1);
  const caseHess = (name, seed, shape, family) => {
    const a = fill(seed, numel(shape)), N = shape[shape.length - 1];
    if (family === 'sparse') { const m = fill(seed + 1000, a.length); for (let i = 0; i < a.length; i++) if (m[i] > 0.6) a[i] = 0; }
    if (family === 'hess') for (let o = 0; o < a.length; o += N * N) for (let i = 0; i < N; i++) for (let j = 0; j + 1 < i; j++) a[o + i * N + j] = 0;   // already Hessenberg: every step is skipped
    const [U, H] = nd.la.hessenberg_decomp(NDA(shape, a));
    record(name, {op: 'hessenberg_decomp', seed, shape, family: family || 'dense'}, {U: [U.data, Array.from(U.shape)], H: [H.data, Array.from(H.shape)]});
  };
  caseHess('hess_1x1', 181, [1, 1]);
  caseHess('hess_2x2', 182, [2, 2]);
  caseHess('hess_3x3', 183, [3, 3]);
  caseHess('hess_17', 184, [17, 17]);
  caseHess('hess_batch', 185, [2, 3, 12, 12]);
  caseHess('hess_sparse_40', 186, [40, 40], 'sparse');
  caseHess('hess_already_20', 187, [20, 20], 'hess');
  caseHess('hess_100', 188, [100, 100]);
  caseHess('hess_257', 189, [257, 257]);
}

if (what === 'bidiag') {
  /* SURVEY §8f N4: bidiag_decomp (bidiag.js:32-319), all three shape branches */
  const numel = sh => sh.reduce((x, y) => x * y, 1);
  const caseBd = (name, seed, shape, sparse) => {
    const a = fill(seed, numel(shape));
    if (sparse) { const m = fill(seed + 1000, a.length); for (let i = 0; i < a.length; i++) if (m[i] > 0.6) a[i] = 0; }
    const [U, B, V] = nd.la.bidiag_decomp(NDA(shape, a));
    record(name, {op: 'bidiag_decomp', seed, shape, sparse: !!sparse},
           {U: [U.data, Array.from(U.shape)], B: [B.data, Array.from(B.shape)], V: [V.data, Array.from(V.shape)]});
  };
  caseBd('bidiag_1x1', 191, [1, 1]);
  caseBd('bidiag_2x2', 192, [2, 2]);
  caseBd('bidiag_sq_17', 193, [17, 17]);
  caseBd('bidiag_sq_64', 194, [64, 64]);
  caseBd('bidiag_vert_20x7', 195, [20, 7]);
  caseBd('bidiag_vert_65x64', 196, [65, 64]);
  caseBd('bidiag_vert_5x1', 197, [5, 1]);
  caseBd('bidiag_horiz_7x20', 198, [7, 20]);
  caseBd('bidiag_horiz_1x5', 199, [1, 5]);
  caseBd('bidiag_horiz_40x41', 200, [40, 41]);
  caseBd('bidiag_batch', 201, [2, 3, 9, 12]);
  caseBd('bidiag_sparse_sq_30', 202, [30, 30], true);
  caseBd('bidiag_sparse_vert', 203, [33, 12], true);
  caseBd('bidiag_sq_130', 204, [130, 130]);
}

if (what === 'chain') {
  /* SURVEY §8a A2: matmul(...ms) (matmul.js:150-236). Hand cases = the three value-pinned cases of the reference's own
     test (matmul_test.js:32-79), stored as float64 inputs + the reference's result; seeded chains exercise the
     FLOP-optimal ordering with NumPy-broadcast leading axes. */
  const numel = sh => sh.reduce((x, y) => x * y, 1);
  const hand = (name, mats) => {
    const ms = mats.map(([shape, vals]) => NDA(shape, Float64Array.from(vals)));
    const C = ms.length === 2 ? nd.la.matmul2(...ms) : nd.la.matmul(...ms);
    const t = {C: [Float64Array.from(C.data), Array.from(C.shape)]};
    ms.forEach((m, k) => { t['M' + k] = [m.data, Array.from(m.shape)]; });
    record(name, {op: 'matmul', hand: true, n: ms.length, shapes: mats.map(m => m[0])}, t);
  };
  hand('chain_hand_2x1_1x3', [[[2, 1], [1, 2]], [[1, 3], [30, 40, 50]]]);
  hand('chain_hand_2x3_3x2', [[[2, 3], [1, 2, 3, 4, 5, 6]], [[3, 2], [70, 80, 90, 100, 110, 120]]]);
  hand('chain_hand_1x4_4x3_3x2', [[[1, 4], [1, 2, 3, 4]], [[4, 3], [11, 12, 13, 21, 22, 23, 31, 32, 33, 41, 42, 43]], [[3, 2], [5, 6, 7, 8, 9, 10]]]);
  const seeded = (name, seed0, shapes) => {
    const ms = shapes.map((sh, k) => NDA(sh, fill(seed0 + k, numel(sh))));
    const C = nd.la.matmul(...ms);
    record(name, {op: 'matmul', seed0, shapes, shapeC: Array.from(C.shape)}, {C: [C.data, Array.from(C.shape)]});
  };
  seeded('chain_3_small_mid', 301, [[30, 4], [4, 50], [50, 6]]);                 // (AB)C vs A(BC): right first
  seeded('chain_3_left_first', 311, [[5, 60], [60, 7], [7, 80]]);
  seeded('chain_4', 321, [[40, 10], [10, 33], [33, 5], [5, 64]]);
  seeded('chain_5', 331, [[12, 31], [31, 9], [9, 45], [45, 3], [3, 27]]);
  seeded('chain_3_bcast', 341, [[3, 1, 8, 20], [2, 20, 6], [6, 11]]);            // leading axes change the costs
  seeded('chain_4_bcast', 351, [[7, 15], [4, 1, 15, 2], [3, 2, 40], [1, 3, 40, 9]]);
  seeded('chain_5_bcast', 361, [[2, 9, 14], [14, 14], [2, 14, 3], [3, 22], [1, 22, 5]]);
  seeded('chain_ties', 371, [[16, 16], [16, 16], [16, 16], [16, 16]]);           // every order costs the same: first split wins
  seeded('chain_1', 381, [[6, 7]]);
  seeded('chain_vec', 391, [[1, 33], [33, 33], [33, 1]]);
}

if (what === 'c3b') {
  /* 4096^2 QR and LU: sizes whose panels take the >2048-row register-panel variants on the GPU */
  const N = 4096, a = fill(17, N * N);
  let t = Date.now();
  const [Q, R] = nd.la.qr_decomp(NDA([N, N], a)); const tqr = (Date.now() - t) / 1e3;
  console.log('qr', tqr);
  t = Date.now();
  const [LU, P] = nd.la.lu_decomp(NDA([N, N], a)); const tlu = (Date.now() - t) / 1e3;
  const dR = new Float64Array(N); for (let i = 0; i < N; i++) dR[i] = R.data[i * N + i];
  const [qi, qv] = sample(Q.data, 4096, 1701), [ri, rv] = sample(R.data, 4096, 1702), [li, lv] = sample(LU.data, 4096, 1703);
  record('c3b_qr4096', {op: 'qr_decomp', seed: 17, shape: [N, N], froQ: fro(Q.data), froR: fro(R.data), ref_seconds: tqr},
    {diagR: [dR, [N]], Qidx: [qi, [4096]], Qval: [qv, [4096]], Ridx: [ri, [4096]], Rval: [rv, [4096]]});
  record('c3b_lu4096', {op: 'lu_decomp', seed: 17, shape: [N, N], froLU: fro(LU.data), ref_seconds: tlu},
    {P: [P.data, [N]], LUidx: [li, [4096]], LUval: [lv, [4096]]});
}

if (what === 'c2') {
  const N = 4096, A = fill(5, N * N), B = fill(6, N * N);
  const t = Date.now();
  const C = nd.la.matmul2(NDA([N, N], A), NDA([N, N], B)).data;
  const secs = (Date.now() - t) / 1e3;
  const [idx, val] = sample(C, 4096, 605);
  const rows = new Int32Array(64), rowsum = new Float64Array(64);
  for (let r = 0; r < 64; r++) { rows[r] = hashIdx(606, r, N); let s = 0; for (let j = 0; j < N; j++) s += C[rows[r] * N + j]; rowsum[r] = s; }
  record('c2_matmul4096', {op: 'matmul2', seedA: 5, seedB: 6, shapeA: [N, N], shapeB: [N, N], fro: fro(C), ref_seconds: secs},
    {idx: [idx, [4096]], val: [val, [4096]], rows: [rows, [64]], rowsum: [rowsum, [64]]});
}

if (what === 'c3') {
  const N = 2048, a = fill(7, N * N);
  let t = Date.now();
  const [Q, R] = nd.la.qr_decomp(NDA([N, N], a)); const tqr = (Date.now() - t) / 1e3;
  t = Date.now();
  const [LU, P] = nd.la.lu_decomp(NDA([N, N], a)); const tlu = (Date.now() - t) / 1e3;
  const dR = new Float64Array(N); for (let i = 0; i < N; i++) dR[i] = R.data[i * N + i];
  const [qi, qv] = sample(Q.data, 4096, 701), [ri, rv] = sample(R.data, 4096, 702), [li, lv] = sample(LU.data, 4096, 703);
  record('c3_qr2048', {op: 'qr_decomp', seed: 7, shape: [N, N], froQ: fro(Q.data), froR: fro(R.data), ref_seconds: tqr},
    {diagR: [dR, [N]], Qidx: [qi, [4096]], Qval: [qv, [4096]], Ridx: [ri, [4096]], Rval: [rv, [4096]]});
  record('c3_lu2048', {op: 'lu_decomp', seed: 7, shape: [N, N], froLU: fro(LU.data), ref_seconds: tlu},
    {P: [P.data, [N]], LUidx: [li, [4096]], LUval: [lv, [4096]]});
}

if (what === 'c4') {
  const N = 2048, a = fill(9, N * N);
  const t = Date.now();
  const [U, sv, V] = nd.la.svd_decomp(NDA([N, N], a)); const secs = (Date.now() - t) / 1e3;
  const [ui, uv] = sample(U.data, 4096, 901), [vi, vv] = sample(V.data, 4096, 902);
  record('c4_svd2048', {op: 'svd_decomp', seed: 9, shape: [N, N], ref_seconds: secs},
    {sv: [sv.data, [N]], Uidx: [ui, [4096]], Uval: [uv, [4096]], Vidx: [vi, [4096]], Vval: [vv, [4096]]});
}

if (what === 'c5') {
  const N = 512, stride = parseInt(process.argv[3] || '16'), B = 1024, members = [];
  for (let b = 0; b < B; b += stride) members.push(b);
  const svs = new Float64Array(members.length * N); let secs = 0;
  members.forEach((b, k) => {
    const a = fill(1000 + b, N * N); const t = Date.now();
    const sv = nd.la.svd_decomp(NDA([N, N], a))[1].data; secs += (Date.now() - t) / 1e3;
    svs.set(sv, k * N);
    if (k % 8 === 0) console.log('c5 member', b);
  });
  record('c5_svd512', {op: 'svd_decomp', seed_base: 1000, batch: B, stride, shape: [N, N], ref_seconds_per_matrix: secs / members.length},
    {members: [Int32Array.from(members), [members.length]], sv: [svs, [members.length, N]]});
}
console.log('done', what, ((Date.now() - t0) / 1e3).toFixed(1) + ' s');
