// Tiles of one matrix in the registers of 16 x 16 co-resident workgroups for a whole reduction (hess.hip: hessp, bidiag.hip: bdp):
// what crosses between the workgroups per step are 16-byte pairs of the tagged words of xchg.h, written through and read back by
// ONE buffer access each (aux 16 = sc1; each 8-byte half is valid on its own, the reader checks both tags), plus the lane
// exchanges of the partial sums inside a wave.
#pragma once
#include <hip/hip_runtime.h>
#include "xchg.h"
#include "dpp.h"

// The arrays that EVERY workgroup reads (one or a few values per workgroup: partial dot products, norm partials) are kept in HP_NREP
// copies, written by their owners' lanes and read as copy (workgroup % HP_NREP): 256 readers of the same few lines meet in the
// same memory channels (Hessenberg 2048^2: 1 / 8 / 16 / 32 copies 19.8 / 18.9 / 18.3 / 18.3 ms).
constexpr unsigned HP_NREP = 16;
struct HpReq { unsigned off; bool on; };
typedef unsigned int hp_u4 __attribute__((ext_vector_type(4)));

// the two tagged words of xchg.h (payload low / tag, payload high / tag) written through by one 16-byte store (aux 16 = sc1); each
// 8-byte half is valid on its own, the reader checks both tags
__device__ __forceinline__ void hp_st(__amdgpu_buffer_rsrc_t rs, unsigned off, double x, unsigned tag) {
  hp_u4 w;
  w.x = (unsigned)__double2loint(x); w.y = tag; w.z = (unsigned)__double2hiint(x); w.w = tag;
  __builtin_amdgcn_raw_buffer_store_b128(w, rs, (int)off, 0, 16);
}

template <int K>
__device__ __forceinline__ void hp_wait(__amdgpu_buffer_rsrc_t rs, const HpReq (&r)[K], double (&x)[K], unsigned tag, bool& dead, int* abort,
                                        int* status, int delay) {
  bool need[K];
#pragma unroll
  for (int k = 0; k < K; k++) need[k] = r[k].on;
  // look only when the values can have arrived: a poll that fails is traffic the publishers' stores queue behind (two polls in flight
  // at a time: 24 -> 33 ms at 2048^2; a first look ~0.9 us after this workgroup's own publication: 24 -> 20.6 ms)
  for (int d = delay; d > 0; d--) __builtin_amdgcn_s_sleep(8);
  for (int n = 0; n < QX_SPIN_LIMIT; n++) {
    bool ok = true;                                      // (only what has not arrived yet is asked for again)
#pragma unroll
    for (int k = 0; k < K; k++)
      if (need[k]) {
        const hp_u4 w = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)r[k].off, 0, 16);
        if (w.y == tag && w.w == tag) { x[k] = __hiloint2double((int)w.z, (int)w.x); need[k] = false; } else ok = false;
      }
    if (ok) return;
    if ((n & 255) == 255 && __hip_atomic_load(abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { dead = true; return; }
    __builtin_amdgcn_s_sleep(1);
  }
  dead = true;
  __hip_atomic_store(abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  qx_raise(status);
}

// a + b of the partner 16 / 32 lanes away where each keeps one of two values: lanes with the bit clear end with lo(own) + lo(partner),
// lanes with the bit set with hi(own) + hi(partner) (v_permlane{16,32}_swap exchange the odd rows / upper half of the first operand
// with the even rows / lower half of the second)
__device__ __forceinline__ double hp_fold16(double lo, double hi) {
  const auto l = __builtin_amdgcn_permlane16_swap(__double2loint(lo), __double2loint(hi), false, false);
  const auto h = __builtin_amdgcn_permlane16_swap(__double2hiint(lo), __double2hiint(hi), false, false);
  return __hiloint2double(h[0], l[0]) + __hiloint2double(h[1], l[1]);
}
__device__ __forceinline__ double hp_fold32(double lo, double hi) {
  const auto l = __builtin_amdgcn_permlane32_swap(__double2loint(lo), __double2loint(hi), false, false);
  const auto h = __builtin_amdgcn_permlane32_swap(__double2hiint(lo), __double2hiint(hi), false, false);
  return __hiloint2double(h[0], l[0]) + __hiloint2double(h[1], l[1]);
}


// Thread (tr, tc) = (t >> 4, t & 15) of a 256-thread workgroup holds E x E entries of a 16E x 16E tile: rows tr + 16 a, columns tc + 16 b.
// Sum of E per-thread values (one per a) over the 16 lanes of a lane row: halving exchanges first, then plain ones; every lane ends
// with the sum for a = tc >> (4 - log2 E). Fixed order: deterministic.
template <int E>
__device__ __forceinline__ double hp_sum_over_tc(double (&z)[E], int tc) {
  using namespace nd4dpp;
  if constexpr (E >= 2) {
    const bool up = (tc & 8) != 0;
#pragma unroll
    for (int k = 0; k < E / 2; k++) { const double keep = up ? z[k + E / 2] : z[k], send = up ? z[k] : z[k + E / 2]; z[k] = keep + xor8(send); }
  } else z[0] += xor8(z[0]);
  if constexpr (E >= 4) {
    const bool up = (tc & 4) != 0;
#pragma unroll
    for (int k = 0; k < E / 4; k++) { const double keep = up ? z[k + E / 4] : z[k], send = up ? z[k] : z[k + E / 4]; z[k] = keep + xor4(send); }
  } else z[0] += xor4(z[0]);
  if constexpr (E >= 8) {
    const bool up = (tc & 2) != 0;
    const double keep = up ? z[1] : z[0], send = up ? z[0] : z[1];
    z[0] = keep + xor2(send);
  } else z[0] += xor2(z[0]);
  z[0] += xor1(z[0]);
  return z[0];
}
// Sum of E per-thread values (one per b) over the 4 lane rows of a wave, into s_wave[tc + 16 b] (the workgroup adds its 4 waves)
template <int E>
__device__ __forceinline__ void hp_sum_over_tr(const double (&xs)[E], int tr, int tc, int lane, double* s_wave) {
  static_assert(E == 2 || E == 4 || E == 8, "tile edge 32, 64 or 128");
  if constexpr (E == 8) {
    double z[4];
#pragma unroll
    for (int k = 0; k < 4; k++) z[k] = hp_fold16(xs[k], xs[k + 4]);
    const double z0 = hp_fold32(z[0], z[2]), z1 = hp_fold32(z[1], z[3]);
    const int b0 = 4 * (tr & 1) + 2 * ((tr >> 1) & 1);
    s_wave[tc + 16 * b0] = z0; s_wave[tc + 16 * (b0 + 1)] = z1;
  } else if constexpr (E == 4) {
    const double z0 = hp_fold16(xs[0], xs[2]), z1 = hp_fold16(xs[1], xs[3]);
    const double zz = hp_fold32(z0, z1);
    s_wave[tc + 16 * (2 * (tr & 1) + ((tr >> 1) & 1))] = zz;
  } else {
    const double z0 = hp_fold16(xs[0], xs[1]);
    const double zz = hp_fold32(z0, z0);
    if (lane < 32) s_wave[tc + 16 * (tr & 1)] = zz;
  }
}
