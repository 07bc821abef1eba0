'use strict';
/* nd4hip — Node.js host side of the MI355X backend for nd4js's `nd.la` hot path.
 *
 * Same function names, argument meaning, result layout and error messages as the reference:
 *   matmul2     src/la/matmul.js:91-147        qr_decomp   src/la/qr.js:80-145
 *   matmul      src/la/matmul.js:150-236       lu_decomp   src/la/lu.js:24-81
 *   svd_decomp  src/la/svd.js:25 (= svd_dc, src/la/svd_dc.js:883-932)
 * Every float64 call goes through the N-API addon (nd4hip_napi.node -> libnd4hip.so -> HIP kernels).
 * There is NO CPU implementation in this package. Two ways to use it:
 *
 *   const la = require('nd4js_amd/js');                 // standalone: minimal NDArray, float64 only
 *   const nd = require('nd4js'); require('nd4js_amd/js').install(nd);
 *        // drop-in: nd.la.{matmul2,matmul,qr_decomp,lu_decomp,svd_decomp,svd_dc} run on the GPU for
 *        // float64 / int32(promoted) input; other dtypes (float32, complex128, object, int32 x int32
 *        // matmul) keep going to nd4js's own functions, exactly as before.
 */
const path = require('path');
let addon = null;
function native() {
  if (addon === null) {
    try { addon = require(process.env.ND4HIP_NAPI_ADDON || path.join(__dirname, 'nd4hip_napi.node')); }   // (override: the sanitizer build of the shim, tools/check_sanitize.sh)
    catch (e) { throw new Error('nd4hip: cannot load the N-API addon (' + e.message + '). Build it with `python -m nd4js_amd.build`; there is no CPU fallback.'); }
  }
  return addon;
}

/* ---- minimal NDArray (src/nd_array.js:128-158): dense, row-major, shape Int32Array ---- */
class NDArray {
  constructor(shape, data) {
    if (!(shape instanceof Int32Array)) throw new Error('Shape must be Int32Array.');
    if (shape.some(s => s < 1)) throw new Error(`Invalid shape: ${shape}.`);
    if (data.length !== shape.reduce((a, b) => a * b, 1)) throw new Error(`Shape [${shape}] does not match array length of ${data.length}.`);
    this.shape = shape; this.data = data; Object.freeze(this.shape.buffer);
  }
  get ndim() { return this.shape.length; }
  get dtype() {
    if (this.data instanceof Float64Array) return 'float64';
    if (this.data instanceof Float32Array) return 'float32';
    if (this.data instanceof Int32Array) return 'int32';
    return 'object';
  }
  get T() {   // transposes the last two axes (copy), like nd_array.js:362-366
    const sh = Int32Array.from(this.shape), n = sh.length;
    if (n < 2) return this;
    const [M, N] = [sh[n - 2], sh[n - 1]]; sh[n - 2] = N; sh[n - 1] = M;
    const out = new this.data.constructor(this.data.length);
    for (let o = 0; o < out.length; o += M * N)
      for (let i = 0; i < M; i++) for (let j = 0; j < N; j++) out[o + j * M + i] = this.data[o + i * N + j];
    return new NDArray(sh, out);
  }
}
function fromNested(a) {           // nested JS arrays -> NDArray(float64)  (asarray, nd_array.js:102-126)
  const shape = [];
  for (let x = a; Array.isArray(x); x = x[0]) shape.push(x.length);
  const data = new Float64Array(shape.reduce((p, q) => p * q, 1));
  let k = 0;
  (function walk(x, d) {
    if (d === shape.length) { data[k++] = +x; return; }
    if (!Array.isArray(x) || x.length !== shape[d]) throw new Error('asarray(): ragged nested array.');
    for (const y of x) walk(y, d + 1);
  })(a, 0);
  return new NDArray(Int32Array.from(shape.length ? shape : [1]), data);
}
/* ---- device-resident arrays (SURVEY.md §8f N3): `data` stays in HBM across calls, lazy D2H on `.data` ----
 * An extension of the NDArray contract (nd_array.js:128-158), not part of the reference: any nd.la hot-path function
 * that receives at least one DeviceNDArray runs the `_dev` entry points (nothing crosses PCIe except host operands
 * uploaded on the way in) and returns DeviceNDArrays. Results are immutable, so the host copy is cached. */
class DevBuf {
  constructor(length, Ctor) {
    this.length = length; this.Ctor = Ctor;
    this.ext = native().dev_alloc(length * Ctor.BYTES_PER_ELEMENT);
  }
  static from(typed) { const b = new DevBuf(typed.length, typed.constructor); native().dev_upload(b.ext, 0, typed); return b; }
  view(off) { if (!this.ext) throw new Error('nd4hip: device array was disposed.'); return {b: this.ext, o: off}; }
  free() { if (this.ext) { native().dev_free(this.ext); this.ext = null; } }
}
class DeviceNDArray {
  constructor(shape, buf) {
    if (!(shape instanceof Int32Array)) throw new Error('Shape must be Int32Array.');
    if (!(buf instanceof DevBuf) || buf.length !== shape.reduce((a, b) => a * b, 1)) throw new Error(`Shape [${shape}] does not match the device buffer.`);
    this.shape = shape; this._buf = buf; this._host = null;
  }
  get ndim() { return this.shape.length; }
  get dtype() { return this._buf.Ctor === Int32Array ? 'int32' : 'float64'; }
  get onDevice() { return true; }
  get data() {                                   // lazy D2H (blocks until everything queued before it has finished)
    if (this._host === null) {
      if (!this._buf.ext) throw new Error('nd4hip: device array was disposed.');
      const h = new this._buf.Ctor(this._buf.length);
      native().dev_download(h, this._buf.ext, 0);
      this._host = h;
    }
    return this._host;
  }
  toHost(NDA) { return new (NDA || NDArray)(Int32Array.from(this.shape), this.data); }
  dispose() { this._buf.free(); }                // optional: dropped arrays are freed by the GC finalizer
}
const isDev = a => a instanceof DeviceNDArray;

function makeAsarray(NDA) {
  return function asarray(a) {
    if (isDev(a)) return a;                                                       // never touches .data (no D2H)
    if (a && a.shape instanceof Int32Array && a.data !== undefined) return a;     // NDArray of either package
    if (Array.isArray(a)) { const r = fromNested(a); return NDA === NDArray ? r : new NDA(r.shape, r.data); }
    if (typeof a === 'number') return new NDA(Int32Array.of(1), Float64Array.of(a));
    throw new Error('asarray(): unsupported argument.');
  };
}
const dtypeOf = a => isDev(a) ? a.dtype : a.data instanceof Float64Array ? 'float64' : a.data instanceof Int32Array ? 'int32' :
  a.data instanceof Float32Array ? 'float32' : (a.dtype || 'object');
const f64 = a => a.data instanceof Float64Array ? a.data : Float64Array.from(a.data);    // int32 -> float64 promotion
const prod = (s, from, to) => { let p = 1; for (let i = from; i < to; i++) p *= s[i]; return p; };

/* Flatten NumPy-style broadcasting of the leading axes into (count, offA, strideA, offB, strideB, offC)
 * groups with batch strides in {0, dense}: the job of the odometer at matmul.js:44-70. */
function bcastGroups(lead, la, lb, IK, KJ) {
  const nb = lead.length, pad = (s) => Array(nb - s.length).fill(1).concat(Array.from(s));
  la = pad(la); lb = pad(lb);
  const strides = (sh, unit) => { const st = Array(nb).fill(0); let s = unit; for (let d = nb - 1; d >= 0; d--) { st[d] = sh[d] > 1 ? s : 0; s *= sh[d]; } return st; };
  const stA = strides(la, IK), stB = strides(lb, KJ), total = lead.reduce((a, b) => a * b, 1);
  const offA = new Float64Array(total), offB = new Float64Array(total), idx = Array(nb).fill(0);
  for (let b = 0; b < total; b++) {
    let a = 0, c = 0; for (let d = 0; d < nb; d++) { a += idx[d] * stA[d]; c += idx[d] * stB[d]; }
    offA[b] = a; offB[b] = c;
    for (let d = nb - 1; d >= 0; d--) { if (++idx[d] < lead[d]) break; idx[d] = 0; }
  }
  const groups = [];
  for (let b0 = 0; b0 < total;) {
    let b1 = b0 + 1, sA = 0, sB = 0;
    if (b1 < total) {
      sA = offA[b1] - offA[b0]; sB = offB[b1] - offB[b0];
      if ((sA === 0 || sA === IK) && (sB === 0 || sB === KJ))
        while (b1 < total && offA[b1] - offA[b1 - 1] === sA && offB[b1] - offB[b1 - 1] === sB) b1++;
      else sA = sB = 0;
    }
    groups.push([b1 - b0, offA[b0], sA, offB[b0], sB, b0]);
    b0 = b1;
  }
  return groups;
}

/* n-operand form of bcastGroups: [count, [offsets], [strides], outIndex] with stride_k in {0, units[k]} */
function bcastGroupsN(lead, shapes, units) {
  const nb = lead.length, total = lead.reduce((a, b) => a * b, 1);
  const offs = shapes.map((sh, k) => {
    const s = Array(nb - sh.length).fill(1).concat(Array.from(sh)), st = Array(nb).fill(0);
    let u = units[k]; for (let d = nb - 1; d >= 0; d--) { st[d] = s[d] > 1 ? u : 0; u *= s[d]; }
    const o = new Float64Array(total), idx = Array(nb).fill(0);
    for (let b = 0; b < total; b++) {
      let a = 0; for (let d = 0; d < nb; d++) a += idx[d] * st[d];
      o[b] = a;
      for (let d = nb - 1; d >= 0; d--) { if (++idx[d] < lead[d]) break; idx[d] = 0; }
    }
    return o;
  });
  const groups = [];
  for (let b0 = 0; b0 < total;) {
    let b1 = b0 + 1, strides = units.map(() => 0);
    if (b1 < total) {
      const cand = offs.map(o => o[b1] - o[b0]);
      if (cand.every((c, k) => c === 0 || c === units[k])) {
        strides = cand;
        while (b1 < total && offs.every((o, k) => o[b1] - o[b1 - 1] === cand[k])) b1++;
      }
    }
    groups.push([b1 - b0, offs.map(o => o[b0]), strides, b0]);
    b0 = b1;
  }
  return groups;
}
/* Common shape of right-aligned leading-axis lists under NumPy broadcasting (extent 1 stretches). */
function bcastLead(shapes, err) {
  const rank = shapes.reduce((r, sh) => Math.max(r, sh.length), 0), lead = new Array(rank).fill(1);
  shapes.forEach(sh => {
    const shift = rank - sh.length;
    for (let d = 0; d < sh.length; d++) {
      const have = lead[shift + d], want = sh[d];
      if (want === 1 || want === have) continue;
      if (have !== 1) throw new Error(err);
      lead[shift + d] = want;
    }
  });
  return lead;
}
/* Shape of matmul2 for operand shapes sA, sB (plain arrays / Int32Arrays): [result shape, I, K, J]. */
function productShape(sA, sB, mismatch) {
  const I = sA[sA.length - 2], K = sA[sA.length - 1], J = sB[sB.length - 1];
  if (sB[sB.length - 2] != K) throw new Error(mismatch);
  const lead = bcastLead([Array.from(sA).slice(0, -2), Array.from(sB).slice(0, -2)], 'Shapes are not broadcast-compatible.');
  return [lead.concat([I, J]), I, K, J];
}

function makeLa(NDA, fallback) {
  const asarray = makeAsarray(NDA);
  const gpuOk = a => { const d = dtypeOf(a); return d === 'float64' || d === 'int32'; };
  const la = {};
  /* residency plumbing: one code path per function, host TypedArrays or device buffers */
  const alloc = (dev, n, Ctor) => dev ? new DevBuf(n, Ctor || Float64Array) : new (Ctor || Float64Array)(n);
  const view = (x, off) => x instanceof DevBuf ? x.view(off) : x.subarray(off);
  const wrap = (dev, shape, x) => dev ? new DeviceNDArray(Int32Array.from(shape), x) : new NDA(Int32Array.from(shape), x);
  const opF64 = (a, dev, temps) => {             // operand storage as float64 on the side the call runs on
    if (!dev) return f64(a);
    if (isDev(a) && a.dtype === 'float64') return a._buf;
    const b = DevBuf.from(f64(a)); temps.push(b); return b;      // host operand (or device int32): upload a float64 copy
  };
  const opI32 = (a, dev, temps) => {
    if (!dev) return a.data;
    if (isDev(a)) return a._buf;
    const b = DevBuf.from(a.data); temps.push(b); return b;
  };
  const release = temps => { for (const t of temps) t.free(); };
  la.DeviceNDArray = DeviceNDArray;
  la.to_device = a => { a = asarray(a); if (isDev(a)) return a; if (!gpuOk(a)) throw new Error('nd4hip.to_device: dtype ' + dtypeOf(a) + ' is not accelerated.');
                        return new DeviceNDArray(Int32Array.from(a.shape), DevBuf.from(a.data)); };
  la.to_host = a => isDev(a) ? a.toHost(NDA) : asarray(a);
  la.synchronize = () => native().synchronize();
  // per-call profile of the C ABI (nd4hip_profile_enable / nd4hip_profile_last): kernel ms, algorithmic flops and bytes per device
  la.profile_enable = on => native().profile_enable(on !== false);
  la.profile_last = () => native().profile_last();

  la.matmul2 = function matmul2(a, b) {
    a = asarray(a); b = asarray(b);
    if (a.ndim < 2) throw new Error('A must be at least 2D.');
    if (b.ndim < 2) throw new Error('B must be at least 2D.');
    const [shape, I, K, J] = productShape(a.shape, b.shape, 'The last dimension of A and the 2nd to last dimension of B do not match.');
    const da = dtypeOf(a), db = dtypeOf(b);
    // GPU path: at least one float64 operand and the other float64/int32 (result dtype float64, matmul.js:119);
    // int32 x int32 (wrapping Int32Array result), float32, complex128, object -> the host's own function
    if (!((da === 'float64' && gpuOk(b)) || (db === 'float64' && gpuOk(a)))) {
      if (fallback && fallback.matmul2) return fallback.matmul2(a, b);
      throw new Error(`nd4hip.matmul2: dtype pair (${da}, ${db}) is not accelerated and no host nd4js was installed.`);
    }
    const lead = shape.slice(0, -2);
    const dev = isDev(a) || isDev(b), temps = [];
    const A = opF64(a, dev, temps), B = opF64(b, dev, temps), C = alloc(dev, shape.reduce((m, n) => m * n, 1));
    for (const [cnt, offA, sA, offB, sB, offC] of bcastGroups(lead, a.shape.subarray(0, a.ndim - 2), b.shape.subarray(0, b.ndim - 2), I * K, K * J))
      native().dgemm_batched(cnt, I, K, J, view(A, offA), sA, view(B, offB), sB, view(C, offC * I * J));
    release(temps);
    return wrap(dev, shape, C);
  };

  /* Matrix-chain ordering (CLRS 15.2) with broadcast leading axes; one product costs numel(result)*K, which is how
     matmul.js:150-236 counts. cut[lo][hi] = last operand of the left factor of the cheapest split of lo..hi, the first
     among equals. A sub-chain's shape does not depend on its split, so one shape per span is kept. */
  la._chain_plan = function chainPlan(shapes) {
    const n = shapes.length, table = init => shapes.map(() => new Array(n).fill(init));
    const span = table(null), cost = table(0), cut = table(-1);
    shapes.forEach((sh, k) => { span[k][k] = Array.from(sh); });
    for (let width = 1; width < n; width++)
      for (let lo = 0, hi = width; hi < n; lo++, hi++) {
        let best = Infinity;
        for (let mid = lo; mid < hi; mid++) {
          const [shp, , K] = productShape(span[lo][mid], span[mid + 1][hi], 'Shape mismatch.');
          const c = shp.reduce((v, e) => v * e, 1) * K + (cost[lo][mid] + cost[mid + 1][hi]);
          if (c < best) { best = c; cut[lo][hi] = mid; span[lo][hi] = shp; }
        }
        if (cut[lo][hi] < 0) throw new Error('Integer overflow (too many FLOPs).');
        cost[lo][hi] = best;
      }
    return cut;
  };
  la.matmul = function matmul(...matrices) {            // contract of matmul.js:150-236; the ordering stays on the host
    const ms = matrices.map(asarray);
    if (ms.length === 1) return ms[0];
    if (ms.length === 2) return la.matmul2(ms[0], ms[1]);
    const cut = la._chain_plan(ms.map(m => m.shape));
    const evaluate = (lo, hi) => lo === hi ? ms[lo] : la.matmul2(evaluate(lo, cut[lo][hi]), evaluate(cut[lo][hi] + 1, hi));
    return evaluate(0, ms.length - 1);
  };

  la.qr_decomp = function qr_decomp(A) {
    A = asarray(A);
    if (A.ndim < 2) throw new Error('qr_decomp(A): A.ndim must be at least 2.');
    if (!gpuOk(A)) { if (fallback && fallback.qr_decomp) return fallback.qr_decomp(A); throw new Error('nd4hip.qr_decomp: dtype ' + dtypeOf(A) + ' is not accelerated.'); }
    const nd_ = A.ndim, M = A.shape[nd_ - 2], N = A.shape[nd_ - 1], L = Math.min(M, N), batch = prod(A.shape, 0, nd_ - 2);
    const Qs = Int32Array.from(A.shape), Rs = Int32Array.from(A.shape); Qs[nd_ - 1] = L; Rs[nd_ - 2] = L;
    const dev = isDev(A), temps = [];
    const Q = alloc(dev, batch * M * L), R = alloc(dev, batch * L * N);
    native().dgeqrf_q_batched(batch, M, N, view(opF64(A, dev, temps), 0), view(Q, 0), view(R, 0));
    release(temps);
    return [wrap(dev, Qs, Q), wrap(dev, Rs, R)];
  };

  la.qr_decomp_full = function qr_decomp_full(A) {         // qr.js:27-77: Q [..., M, M], R [..., M, N]
    A = asarray(A);
    if (A.ndim < 2) throw new Error('A must be at least 2D.');
    if (!gpuOk(A)) { if (fallback && fallback.qr_decomp_full) return fallback.qr_decomp_full(A); throw new Error('nd4hip.qr_decomp_full: dtype ' + dtypeOf(A) + ' is not accelerated.'); }
    const nd_ = A.ndim, M = A.shape[nd_ - 2], N = A.shape[nd_ - 1], batch = prod(A.shape, 0, nd_ - 2);
    const Qs = Int32Array.from(A.shape); Qs[nd_ - 1] = M;
    const dev = isDev(A), temps = [];
    const Q = alloc(dev, batch * M * M), R = alloc(dev, batch * M * N);
    native().dgeqrf_full_batched(batch, M, N, view(opF64(A, dev, temps), 0), view(Q, 0), view(R, 0));
    release(temps);
    return [wrap(dev, Qs, Q), wrap(dev, A.shape, R)];
  };

  /* qr.js:146-183, same signature and in-place semantics: A (M x N at A_off) <- R, Y (M x L at Y_off) <- Q^T Y.
     A and Y are Float64Arrays; the reference's assertions are kept (its only message is 'Assertion failed.'). */
  la._qr_decomp_inplace = function _qr_decomp_inplace(M, N, L, A, A_off, Y, Y_off) {
    for (const v of [M, N, L, A_off, Y_off]) if (v % 1 !== 0 || !(0 <= v)) throw new Error('Assertion failed.');
    if (!(A instanceof Float64Array) || !(Y instanceof Float64Array)) {
      if (fallback && fallback._qr_decomp_inplace) return fallback._qr_decomp_inplace(M, N, L, A, A_off, Y, Y_off);
      throw new Error('nd4hip._qr_decomp_inplace: only Float64Array storage is accelerated.');
    }
    if (!(M * N <= A.length - A_off)) throw new Error('Assertion failed.');
    if (!(M * L <= Y.length - Y_off)) throw new Error('Assertion failed.');
    if (M === 0 || N === 0) return;
    native().dgeqrf_qty_batched(1, M, N, L, A.subarray(A_off, A_off + M * N), Y.subarray(Y_off, Y_off + M * L));
  };

  la.lu_decomp = function lu_decomp(A) {
    A = asarray(A);
    const nd_ = A.ndim;
    if (nd_ < 2 || A.shape[nd_ - 2] != A.shape[nd_ - 1]) throw new Error('Last two dimensions must be quadratic.');
    if (!gpuOk(A)) { if (fallback && fallback.lu_decomp) return fallback.lu_decomp(A); throw new Error('nd4hip.lu_decomp: dtype ' + dtypeOf(A) + ' is not accelerated.'); }
    const N = A.shape[nd_ - 1], batch = prod(A.shape, 0, nd_ - 2);
    const dev = isDev(A), temps = [];
    const LU = alloc(dev, batch * N * N), P = alloc(dev, batch * N, Int32Array);
    native().dgetrf_batched(batch, N, view(opF64(A, dev, temps), 0), view(LU, 0), view(P, 0));
    release(temps);
    return [wrap(dev, A.shape, LU), wrap(dev, A.shape.subarray(0, nd_ - 1), P)];
  };

  la.svd_decomp = function svd_decomp(A) {
    A = asarray(A);
    if (String(dtypeOf(A)).startsWith('complex')) throw new Error('svd_dc(A): A.dtype must be float.');
    if (A.ndim < 2) throw new Error('svd_decomp(A): A.ndim must be at least 2.');
    if (!gpuOk(A)) { if (fallback && fallback.svd_decomp) return fallback.svd_decomp(A); throw new Error('nd4hip.svd_decomp: dtype ' + dtypeOf(A) + ' is not accelerated.'); }
    const nd_ = A.ndim, M = A.shape[nd_ - 2], N = A.shape[nd_ - 1], L = Math.min(M, N), batch = prod(A.shape, 0, nd_ - 2);
    const Us = Int32Array.from(A.shape), Vs = Int32Array.from(A.shape); Us[nd_ - 1] = L; Vs[nd_ - 2] = L;
    const dev = isDev(A), temps = [];
    const U = alloc(dev, batch * M * L), sv = alloc(dev, batch * L), V = alloc(dev, batch * L * N);
    la.last_svd_info = native().dgesvdj_batched(batch, M, N, view(opF64(A, dev, temps), 0), view(U, 0), view(sv, 0), view(V, 0));
    release(temps);
    return [wrap(dev, Us, U), wrap(dev, Us.subarray(0, nd_ - 1), sv), wrap(dev, Vs, V)];
  };
  la.svd_dc = la.svd_decomp;

  /* ---- SURVEY §8f N1: solve-side consumers (csrc/trsm.hip) ---- */
  const triSolve = (upper, name, T, Y) => {
    T = asarray(T); Y = asarray(Y);
    const tn = upper ? 'U' : 'L';
    if (T.ndim < 2) throw new Error(`${name}(${tn},Y): ${tn}.ndim must be at least 2.`);
    if (Y.ndim < 2) throw new Error(`${name}(${tn},Y): Y.ndim must be at least 2.`);
    const M = Y.shape[Y.ndim - 2], J = Y.shape[Y.ndim - 1];
    if (T.shape[T.ndim - 2] !== M) throw new Error(`${name}(${tn},Y): ${tn} and Y don't match.`);
    if (T.shape[T.ndim - 1] !== M) throw new Error(`${name}(${tn},Y): Last two dimensions of ${tn} must be quadratic.`);
    if (!gpuOk(T) || !gpuOk(Y)) { if (fallback && fallback[name]) return fallback[name](T, Y); throw new Error(`nd4hip.${name}: dtype is not accelerated.`); }
    const lead = bcastLead([Array.from(T.shape.subarray(0, T.ndim - 2)), Array.from(Y.shape.subarray(0, Y.ndim - 2))], `${name}(${tn},Y): ${tn} and Y not broadcast-compatible.`);
    const dev = isDev(T) || isDev(Y), temps = [];
    const Td = opF64(T, dev, temps), Yd = opF64(Y, dev, temps), X = alloc(dev, lead.reduce((a, b) => a * b, 1) * M * J);
    for (const [cnt, [oT, oY], [sT, sY], b0] of bcastGroupsN(lead, [T.shape.subarray(0, T.ndim - 2), Y.shape.subarray(0, Y.ndim - 2)], [M * M, M * J]))
      native().dtrsm_batched(upper ? 1 : 0, 0, cnt, M, J, view(Td, oT), sT, view(Yd, oY), sY, view(X, b0 * M * J));
    release(temps);
    return wrap(dev, [...lead, M, J], X);
  };
  la.tril_solve = (L, Y) => triSolve(false, 'tril_solve', L, Y);
  la.triu_solve = (U, Y) => triSolve(true, 'triu_solve', U, Y);

  la.lu_solve = function lu_solve(LU, P, y) {
    if (undefined == y) { y = P; [LU, P] = LU; }
    LU = asarray(LU); if (LU.ndim < 2) throw new Error('LU must be at least 2D.');
    P = asarray(P); if (P.ndim < 1) throw new Error('P must be at least 1D.');
    y = asarray(y); if (y.ndim < 2) throw new Error('y must be at least 2D.');
    const N = LU.shape[LU.ndim - 2], I = y.shape[y.ndim - 2], J = y.shape[y.ndim - 1];
    if (LU.shape[LU.ndim - 1] != N) throw new Error('Last two dimensions of LU must be quadratic.');
    if (N != I) throw new Error("LU and y don't match.");
    if (N != P.shape[P.ndim - 1]) throw new Error("LU and P don't match.");
    if (!gpuOk(LU) || !gpuOk(y) || dtypeOf(P) !== 'int32') {
      if (fallback && fallback.lu_solve) return fallback.lu_solve(LU, P, y);
      throw new Error('nd4hip.lu_solve: dtype is not accelerated.');
    }
    const lLU = Array.from(LU.shape.subarray(0, LU.ndim - 2)), lP = Array.from(P.shape.subarray(0, P.ndim - 1)), lY = Array.from(y.shape.subarray(0, y.ndim - 2));
    let lead = bcastLead([lLU, lY], 'LU and y are not broadcast-compatible.');
    lead = bcastLead([lead, lP], 'P is not broadcast-compatible.');
    const dev = isDev(LU) || isDev(P) || isDev(y), temps = [];
    const X = alloc(dev, lead.reduce((a, b) => a * b, 1) * N * J), LUd = opF64(LU, dev, temps), Pd = opI32(P, dev, temps), yd = opF64(y, dev, temps);
    for (const [cnt, [oLU, oP, oY], [sLU, sP, sY], b0] of bcastGroupsN(lead, [lLU, lP, lY], [N * N, N, N * J]))
      native().dgetrs_batched(cnt, N, J, view(LUd, oLU), sLU, view(Pd, oP), sP, view(yd, oY), sY, view(X, b0 * N * J));
    release(temps);
    return wrap(dev, [...lead, N, J], X);
  };

  /* ---- SURVEY §8f N4: Cholesky (csrc/chol.hip) ---- */
  la.cholesky_decomp = function cholesky_decomp(S) {       // cholesky.js:51-71
    S = asarray(S);
    const nd_ = S.ndim;
    if (nd_ < 2 || S.shape[nd_ - 2] != S.shape[nd_ - 1]) throw new Error('Last two dimensions must be quadratic.');
    if (!gpuOk(S)) { if (fallback && fallback.cholesky_decomp) return fallback.cholesky_decomp(S); throw new Error('nd4hip.cholesky_decomp: dtype ' + dtypeOf(S) + ' is not accelerated.'); }
    const N = S.shape[nd_ - 1], batch = prod(S.shape, 0, nd_ - 2), dev = isDev(S), temps = [];
    const L = alloc(dev, batch * N * N);
    try { native().dpotrf_batched(batch, N, view(opF64(S, dev, temps), 0), view(L, 0)); }
    catch (e) { if (dev) L.free(); if (/near\) singular/.test(e.message)) throw new Error('Matrix contains NaNs or is (near) singular.'); throw e; }
    finally { release(temps); }
    return wrap(dev, S.shape, L);
  };

  la.cholesky_solve = function cholesky_solve(L, y) {      // cholesky.js:74-150
    L = asarray(L); y = asarray(y);
    if (L.ndim < 2) throw new Error('L must be at least 2D.');
    if (y.ndim < 2) throw new Error('y must be at least 2D.');
    const N = L.shape[L.ndim - 2], M = L.shape[L.ndim - 1], I = y.shape[y.ndim - 2], J = y.shape[y.ndim - 1];
    if (N != M) throw new Error('Last two dimensions of L must be quadratic.');
    if (I != M) throw new Error("L and y don't match.");
    if (!gpuOk(L) || !gpuOk(y)) { if (fallback && fallback.cholesky_solve) return fallback.cholesky_solve(L, y); throw new Error('nd4hip.cholesky_solve: dtype is not accelerated.'); }
    const lL = Array.from(L.shape.subarray(0, L.ndim - 2)), lY = Array.from(y.shape.subarray(0, y.ndim - 2));
    const lead = bcastLead([lL, lY], 'Shapes are not broadcast-compatible.');
    const dev = isDev(L) || isDev(y), temps = [];
    const X = alloc(dev, lead.reduce((a, b) => a * b, 1) * N * J), Ld = opF64(L, dev, temps), yd = opF64(y, dev, temps);
    for (const [cnt, [oL, oY], [sL, sY], b0] of bcastGroupsN(lead, [lL, lY], [N * N, N * J]))
      native().dpotrs_batched(cnt, N, J, view(Ld, oL), sL, view(yd, oY), sY, view(X, b0 * N * J));
    release(temps);
    return wrap(dev, [...lead, N, J], X);
  };

  la.ldl_decomp = function ldl_decomp(S) {                 // ldl.js:67-90
    S = asarray(S);
    const nd_ = S.ndim;
    if (nd_ < 2 || S.shape[nd_ - 2] !== S.shape[nd_ - 1]) throw new Error('Last two dimensions must be quadratic.');
    if (!gpuOk(S)) { if (fallback && fallback.ldl_decomp) return fallback.ldl_decomp(S); throw new Error('nd4hip.ldl_decomp: dtype ' + dtypeOf(S) + ' is not accelerated.'); }
    const N = S.shape[nd_ - 1], batch = prod(S.shape, 0, nd_ - 2), dev = isDev(S), temps = [];
    const LD = alloc(dev, batch * N * N);
    native().dldltrf_batched(batch, N, view(opF64(S, dev, temps), 0), view(LD, 0));
    release(temps);
    return wrap(dev, S.shape, LD);
  };

  la.ldl_solve = function ldl_solve(LD, y) {               // ldl.js:133-201
    LD = asarray(LD); y = asarray(y);
    if (LD.ndim < 2) throw new Error('ldl_solve(LD,y): LD must be at least 2D.');
    if (y.ndim < 2) throw new Error('ldl_solve(LD,y): y must be at least 2D.');
    const N = LD.shape[LD.ndim - 2], M = LD.shape[LD.ndim - 1], I = y.shape[y.ndim - 2], J = y.shape[y.ndim - 1];
    if (N != M) throw new Error('ldl_solve(LD,y): Last two dimensions of LD must be quadratic.');
    if (I != M) throw new Error("ldl_solve(LD,y): LD and y don't match.");
    if (!gpuOk(LD) || !gpuOk(y)) { if (fallback && fallback.ldl_solve) return fallback.ldl_solve(LD, y); throw new Error('nd4hip.ldl_solve: dtype is not accelerated.'); }
    const lL = Array.from(LD.shape.subarray(0, LD.ndim - 2)), lY = Array.from(y.shape.subarray(0, y.ndim - 2));
    const lead = bcastLead([lL, lY], 'Shapes are not broadcast-compatible.');
    const dev = isDev(LD) || isDev(y), temps = [];
    const X = alloc(dev, lead.reduce((a, b) => a * b, 1) * N * J), Ld = opF64(LD, dev, temps), yd = opF64(y, dev, temps);
    for (const [cnt, [oL, oY], [sL, sY], b0] of bcastGroupsN(lead, [lL, lY], [N * N, N * J]))
      native().dldltrs_batched(cnt, N, J, view(Ld, oL), sL, view(yd, oY), sY, view(X, b0 * N * J));
    release(temps);
    return wrap(dev, [...lead, N, J], X);
  };

  la.bidiag_decomp = function bidiag_decomp(A) {           // bidiag.js:245-319
    A = asarray(A);
    if (A.ndim < 2) throw new Error('bidiag_decomp(A): A must be at least 2D.');
    if (String(dtypeOf(A)).startsWith('complex')) throw new Error('bidiag_decomp(A): complex A not yet supported.');
    if (!gpuOk(A)) { if (fallback && fallback.bidiag_decomp) return fallback.bidiag_decomp(A); throw new Error('nd4hip.bidiag_decomp: dtype ' + dtypeOf(A) + ' is not accelerated.'); }
    const nd_ = A.ndim, M = A.shape[nd_ - 2], N = A.shape[nd_ - 1], I = Math.min(M, N), J = M >= N ? I : I + 1, batch = prod(A.shape, 0, nd_ - 2);
    const Us = Int32Array.from(A.shape), Bs = Int32Array.from(A.shape), Vs = Int32Array.from(A.shape);
    Us[nd_ - 1] = I; Bs[nd_ - 2] = I; Bs[nd_ - 1] = J; Vs[nd_ - 2] = J;
    const dev = isDev(A), temps = [];
    const U = alloc(dev, batch * M * I), B = alloc(dev, batch * I * J), V = alloc(dev, batch * J * N);
    native().dgebrd_batched(batch, M, N, view(opF64(A, dev, temps), 0), view(U, 0), view(B, 0), view(V, 0));
    release(temps);
    return [wrap(dev, Us, U), wrap(dev, Bs, B), wrap(dev, Vs, V)];
  };

  la.hessenberg_decomp = function hessenberg_decomp(A) {   // hessenberg.js:89-115
    A = asarray(A);
    if (A.ndim < 2) throw new Error('hessenberg_decomp(A): A must at least be 2D.');
    const nd_ = A.ndim, N = A.shape[nd_ - 1];
    if (N != A.shape[nd_ - 2]) throw new Error('hessenberg_decomp(A): A must be square.');
    if (!gpuOk(A)) { if (fallback && fallback.hessenberg_decomp) return fallback.hessenberg_decomp(A); throw new Error('nd4hip.hessenberg_decomp: dtype ' + dtypeOf(A) + ' is not accelerated.'); }
    const batch = prod(A.shape, 0, nd_ - 2), dev = isDev(A), temps = [];
    const U = alloc(dev, batch * N * N), H = alloc(dev, batch * N * N);
    native().dgehrd_batched(batch, N, view(opF64(A, dev, temps), 0), view(U, 0), view(H, 0));
    release(temps);
    return [wrap(dev, A.shape, U), wrap(dev, A.shape, H)];
  };

  /* ---- least squares from a factorisation: qr_lstsq (qr.js:186-273), svd_lstsq / svd_solve (svd.js:66-228) ---- */
  la.qr_lstsq = function qr_lstsq(Q, R, y) {
    if (undefined == y) { y = R; [Q, R] = Q; }
    Q = asarray(Q); if (Q.ndim < 2) throw new Error('qr_lstsq(Q,R,y): Q.ndim must be at least 2.');
    R = asarray(R); if (R.ndim < 2) throw new Error('qr_lstsq(Q,R,y): R.ndim must be at least 2.');
    y = asarray(y); if (y.ndim < 2) throw new Error('qr_lstsq(Q,R,y): y.ndim must be at least 2.');
    const N = Q.shape[Q.ndim - 2], M = Q.shape[Q.ndim - 1], I = R.shape[R.ndim - 1], J = y.shape[y.ndim - 1];
    if (N != y.shape[y.ndim - 2]) throw new Error("qr_lstsq(Q,R,y): Q and y don't match.");
    if (M != R.shape[R.ndim - 2]) throw new Error("qr_lstsq(Q,R,y): Q and R don't match.");
    if (I > N) throw new Error('qr_lstsq(Q,R,y): Under-determined systems not supported. Use rrqr instead.');
    if (!gpuOk(Q) || !gpuOk(R) || !gpuOk(y)) {
      if (fallback && fallback.qr_lstsq) return fallback.qr_lstsq(Q, R, y);
      throw new Error('nd4hip.qr_lstsq: dtype is not accelerated.');
    }
    const lQ = Array.from(Q.shape.subarray(0, Q.ndim - 2)), lR = Array.from(R.shape.subarray(0, R.ndim - 2)), lY = Array.from(y.shape.subarray(0, y.ndim - 2));
    const lead = bcastLead([lQ, lR, lY], 'Q, R, y are not broadcast-compatible.');
    const dev = isDev(Q) || isDev(R) || isDev(y), temps = [];
    const X = alloc(dev, lead.reduce((a, b) => a * b, 1) * I * J), Qd = opF64(Q, dev, temps), Rd = opF64(R, dev, temps), yd = opF64(y, dev, temps);
    for (const [cnt, [oQ, oR, oY], [sQ, sR, sY], b0] of bcastGroupsN(lead, [lQ, lR, lY], [N * M, M * I, N * J]))
      native().dqrls_batched(cnt, N, M, I, J, view(Qd, oQ), sQ, view(Rd, oR), sR, view(yd, oY), sY, view(X, b0 * I * J));
    release(temps);
    return wrap(dev, [...lead, I, J], X);
  };

  la.svd_lstsq = function svd_lstsq(U, sv, V, y) {
    if (y == undefined) {
      if (V != undefined) throw new Error('svd_lstsq(Q,R,P, y): Either 2 ([Q,R,P], y) or 4 arguments (Q,R,P, y) expected.');
      y = sv; [U, sv, V] = U;
    }
    U = asarray(U); if (U.ndim < 2) throw new Error('svd_lstsq(U,sv,V, y): U.ndim must be at least 2.');
    sv = asarray(sv); if (sv.ndim < 1) throw new Error('svd_lstsq(U,sv,V, y): sv.ndim must be at least 1.');
    V = asarray(V); if (V.ndim < 2) throw new Error('svd_lstsq(U,sv,V, y): V.ndim must be at least 2.');
    y = asarray(y); if (y.ndim < 2) throw new Error('svd_lstsq(U,sv,V, y): y.ndim must be at least 2.');
    const N = U.shape[U.ndim - 2], M = U.shape[U.ndim - 1], I = V.shape[V.ndim - 1], J = y.shape[y.ndim - 1];
    if (N !== y.shape[y.ndim - 2]) throw new Error("svd_lstsq(U,sv,V, y): U and y don't match.");
    if (M !== sv.shape[sv.ndim - 1]) throw new Error("svd_lstsq(U,sv,V, y): U and sv don't match.");
    if (M !== V.shape[V.ndim - 2]) throw new Error("svd_lstsq(U,sv,V, y): V and sv don't match.");
    if (!gpuOk(U) || !gpuOk(sv) || !gpuOk(V) || !gpuOk(y)) {
      if (fallback && fallback.svd_lstsq) return fallback.svd_lstsq(U, sv, V, y);
      throw new Error('nd4hip.svd_lstsq: dtype is not accelerated.');
    }
    const lU = Array.from(U.shape.subarray(0, U.ndim - 2)), lS = Array.from(sv.shape.subarray(0, sv.ndim - 1)),
          lV = Array.from(V.shape.subarray(0, V.ndim - 2)), lY = Array.from(y.shape.subarray(0, y.ndim - 2));
    const lead = bcastLead([lU, lV, lY, lS], 'svd_lstsq(U,sv,V, y): U,sv,V,y not broadcast-compatible.');
    const svh = sv.data;                                     // tiny (batch x M): a device sv is read back once and cached
    for (let i = 0; i < svh.length; i++) if (!isFinite(svh[i])) throw new Error('svd_solve(): NaN or Infinity encountered.');   // svd.js:171-172
    const dev = isDev(U) || isDev(sv) || isDev(V) || isDev(y), temps = [];
    const X = alloc(dev, lead.reduce((a, b) => a * b, 1) * I * J), Ud = opF64(U, dev, temps), svd = opF64(sv, dev, temps), Vd = opF64(V, dev, temps), yd = opF64(y, dev, temps);
    for (const [cnt, [oU, oS, oV, oY], [sU, sS, sV, sY], b0] of bcastGroupsN(lead, [lU, lS, lV, lY], [N * M, M, M * I, N * J]))
      native().dsvdls_batched(cnt, N, M, I, J, view(Ud, oU), sU, view(svd, oS), sS, view(Vd, oV), sV, view(yd, oY), sY, view(X, b0 * I * J));
    release(temps);
    return wrap(dev, [...lead, I, J], X);
  };

  /* svd.js:66-97: the reference's singularity loop (`for( let r; r < N; r++ )`, :85) never executes, so apart from the
     squareness check svd_solve IS svd_lstsq; mirrored as such. */
  la.svd_solve = function svd_solve(U, sv, V, y) {
    if (y == undefined) {
      if (V != undefined) throw new Error('svd_lstsq(Q,R,P, y): Either 2 ([Q,R,P], y) or 4 arguments (Q,R,P, y) expected.');
      y = sv; [U, sv, V] = U;
    }
    U = asarray(U); sv = asarray(sv); V = asarray(V);
    if (U.shape[U.ndim - 2] !== V.shape[V.ndim - 1]) throw new Error('rrqr_solve(Q,R,P, y): System not square.');
    return la.svd_lstsq(U, sv, V, y);
  };
  /* svd.js:31-63: per matrix the first r with |sv_r| <= sqrt(eps) |sv_0| (entries are only examined up to that point, so a
     NaN behind the cut does not raise, exactly like the reference's loop). Host side: sv is tiny; a device array is read back. */
  la.svd_rank = function svd_rank(sv) {
    sv = la.to_host(asarray(sv));
    const N = sv.shape[sv.ndim - 1], data = sv.data, count = N > 0 ? data.length / N : 0, r = new Int32Array(count);
    const EPS = Math.sqrt(dtypeOf(sv) === 'float32' ? 2 ** -23 : 2 ** -52);
    for (let off = 0; off < count; off++) {
      const T = EPS * Math.abs(data[N * off]);
      for (; r[off] < N; r[off]++) {
        const x = Math.abs(data[N * off + r[off]]);
        if (!isFinite(x)) throw new Error('svd_rank(): NaN or Infinity encountered.');
        if (x <= T) break;
      }
    }
    return new NDA(Int32Array.from(sv.shape.subarray(0, sv.ndim - 1)), r);
  };
  return la;
}

const standalone = makeLa(NDArray, null);

// estimated work of a call (flops, SURVEY.md 8d conventions up to a constant): what install(nd, {minWork}) compares with
function estimatedWork(name, args) {
  const dims = a => { const s = a && a.shape ? Array.from(a.shape) : null; if (!s || s.length < 2) return null;
                      let b = 1; for (let i = 0; i < s.length - 2; i++) b *= s[i]; return [b, s[s.length - 2], s[s.length - 1]]; };
  const a = dims(args[0]);
  if (!a) return Infinity;                                  // nested JS arrays etc.: let the accelerated path coerce them
  if (name === 'matmul2') { const b = dims(args[1]); return b ? 2 * Math.max(a[0], b[0]) * a[1] * a[2] * b[2] : Infinity; }
  if (name === 'matmul') { let w = 0; for (let i = 0; i + 1 < args.length; i++) { const x = dims(args[i]), y = dims(args[i + 1]); if (!x || !y) return Infinity; w += 2 * Math.max(x[0], y[0]) * x[1] * x[2] * y[2]; } return w; }
  return a[0] * a[1] * a[2] * Math.min(a[1], a[2]);
}

/** Patch a loaded nd4js instance in place: the hot-path functions run on the GPU. Returns nd.
 *  opts.minWork (default 0 = off): a call whose estimated work (flops: 2 I K J for products, M N min(M, N) per matrix for
 *  decompositions and solves) is below it — and whose operands all live on the host — is forwarded to the HOST MODULE'S OWN
 *  function (`nd.la.*` as it was before install; never to this repo's oracle): one tiny matrix costs 50 us - 1 ms through any GPU
 *  path (the reference's own suites live at N <= 161: _generic_test_svd_decomp.js:308-336, lu_test.js:82-94). */
function install(nd, opts) {
  if (!nd || !nd.la || !nd.NDArray) throw new Error('nd4hip.install(nd): pass the nd4js module.');
  const minWork = opts && opts.minWork > 0 ? +opts.minWork : 0;
  const original = {matmul2: nd.la.matmul2, matmul: nd.la.matmul, qr_decomp: nd.la.qr_decomp, qr_decomp_full: nd.la.qr_decomp_full,
                    lu_decomp: nd.la.lu_decomp, svd_decomp: nd.la.svd_decomp, svd_dc: nd.la.svd_dc,
                    lu_solve: nd.la.lu_solve, tril_solve: nd.la.tril_solve, triu_solve: nd.la.triu_solve,
                    qr_lstsq: nd.la.qr_lstsq, svd_lstsq: nd.la.svd_lstsq, svd_solve: nd.la.svd_solve, svd_rank: nd.la.svd_rank,
                    cholesky_decomp: nd.la.cholesky_decomp, cholesky_solve: nd.la.cholesky_solve,
                    ldl_decomp: nd.la.ldl_decomp, ldl_solve: nd.la.ldl_solve, hessenberg_decomp: nd.la.hessenberg_decomp, bidiag_decomp: nd.la.bidiag_decomp};
  const acc = makeLa(nd.NDArray, original);
  const target = Object.isFrozen(nd.la) || !Object.getOwnPropertyDescriptor(nd.la, 'matmul2').writable ? null : nd.la;
  const patched = target || Object.create(nd.la);
  const route = k => {
    if (!minWork || typeof original[k] !== 'function') return acc[k];
    const f = function (...args) {
      const onHost = args.every(x => !(x instanceof acc.DeviceNDArray));
      return onHost && estimatedWork(k, args) < minWork ? original[k].apply(nd.la, args) : acc[k](...args);
    };
    Object.defineProperty(f, 'name', {value: k});
    return f;
  };
  for (const k of Object.keys(original)) Object.defineProperty(patched, k, {value: route(k), writable: true, enumerable: true, configurable: true});
  for (const k of ['to_device', 'to_host', 'synchronize', 'DeviceNDArray', 'profile_enable', 'profile_last'])   // §8f N3 / §8b extensions, not in the reference
    Object.defineProperty(patched, k, {value: acc[k], writable: true, enumerable: false, configurable: true});
  if (!target) { try { nd.la = patched; } catch (e) { /* exported getter: caller uses the returned object */ } }
  patched.__nd4hip_original__ = original;
  if (target) return nd;
  const out = Object.create(nd);
  Object.defineProperty(out, 'la', {value: patched, enumerable: true, writable: true, configurable: true});
  return out;
}

module.exports = Object.assign(standalone, {
  NDArray, DeviceNDArray, install, bcastGroups,
  accelerated: a => !!a && (isDev(a) || a.data instanceof Float64Array || a.data instanceof Int32Array),
  device_count: () => native().device_count(),
  devices: () => native().devices(),            // ids behind the handle (ND4HIP_DEVICES=all|0,1,...: batched host calls are sharded)
  version: () => native().version(),
});
