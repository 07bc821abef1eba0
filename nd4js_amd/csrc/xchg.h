// Tagged words: the exchange between co-resident workgroups INSIDE a kernel (tools/xwg_lat.hip).
// Agent-scope relaxed atomics only: sc1 stores are written through, sc1 loads miss the non-coherent caches, so nothing has to be
// written back or invalidated (a release / acquire pair costs 2-18 us on this chip, this ~1.4 us). Every 8-byte word that crosses
// carries 32 bits of payload and a 32-bit tag: a double travels as two single-copy-atomic words and is valid as soon as both carry
// the tag the reader expects — no flag, no s_waitcnt, no ordering between words. The slots are cleared before a tag can repeat.
#pragma once
#include <hip/hip_runtime.h>

typedef unsigned long long qx_u64;
// The partners of an exchange are consecutive workgroups of ONE launch and the host only uses it while they all fit on the chip at
// once (QR: the row workgroups of a panel launch come first in the grid, at most 8 x 33 of 512 threads = one per CU; LU:
// P x batch <= 64 workgroups; the one-launch Hessenberg / bidiagonalisation, xchg16.h: exactly 256 workgroups of 256 threads, taken
// only when the occupancy query says that 256 fit the device's CUs at once), so each partner is resident or next in the dispatch order. HIP does not promise that order, and a GPU
// shared with other processes may delay a workgroup: every spin is therefore bounded. A stuck exchange ends after about a second
// (QX_SPIN_LIMIT polls of ~1 us), marks its results (NaN / P = -1) and raises the handle's status word, which the next synchronising
// entry point turns into ND4HIP_ERR_XCHG.
constexpr int QX_SPIN_LIMIT = 1 << 20;
__device__ __forceinline__ void qx_raise(int* status) {
  if (status != nullptr) __hip_atomic_store(status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__device__ __forceinline__ void qx_st(qx_u64* slot, int v, double x, unsigned tag) {
  const qx_u64 bits = (qx_u64)__double_as_longlong(x), tg = (qx_u64)tag << 32;
  __hip_atomic_store(slot + 2 * v, (bits & 0xffffffffull) | tg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(slot + 2 * v + 1, (bits >> 32) | tg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// one value; ok stays true only if both words carried the tag
__device__ __forceinline__ double qx_ld(const qx_u64* slot, int v, unsigned tag, bool& ok) {
  const qx_u64 w0 = __hip_atomic_load(slot + 2 * v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const qx_u64 w1 = __hip_atomic_load(slot + 2 * v + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  ok = ok && (unsigned)(w0 >> 32) == tag && (unsigned)(w1 >> 32) == tag;
  return __longlong_as_double((long long)((w1 << 32) | (w0 & 0xffffffffull)));
}
// fixed-order sum of entry v over the n slots (`stride` words apart); ok stays true only if every word carried the tag
__device__ __forceinline__ double qx_sum(const qx_u64* __restrict__ slots, int v, int n, unsigned tag, bool& ok, long stride = 512) {
  double x = 0.0;
  for (int p0 = 0; p0 < n; p0 += 8) {
    qx_u64 w0[8], w1[8];
#pragma unroll
    for (int p = 0; p < 8; p++) {
      const qx_u64* sp = slots + (long)(p0 + p < n ? p0 + p : 0) * stride + 2 * v;
      w0[p] = __hip_atomic_load(sp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      w1[p] = __hip_atomic_load(sp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for (int p = 0; p < 8; p++)
      if (p0 + p < n) {
        ok = ok && (unsigned)(w0[p] >> 32) == tag && (unsigned)(w1[p] >> 32) == tag;
        x += __longlong_as_double((long long)((w1[p] << 32) | (w0[p] & 0xffffffffull)));
      }
  }
  return x;
}
