// Wave64 cross-lane helpers built on DPP (data-parallel primitives: v_mov_b32_dpp modifiers, VALU speed)
// instead of ds_bpermute (__shfl*, LDS crossbar, ~100+ cycles of dependent latency per step). The panel
// factorisations are latency chains of small reductions, so this is where their time goes.
#pragma once
#include <hip/hip_runtime.h>

namespace nd4dpp {

constexpr int QP_XOR1 = 0xB1;        // quad_perm [1,0,3,2]
constexpr int QP_XOR2 = 0x4E;        // quad_perm [2,3,0,1]
constexpr int ROW_SHL4 = 0x104;      // lane i <- lane i+4 (within its row of 16)
constexpr int ROW_SHR4 = 0x114;      // lane i <- lane i-4
constexpr int ROW_ROR8 = 0x128;      // lane i <- lane (i+8) mod 16  == xor 8
constexpr int ROW_MIRROR = 0x140;
constexpr int ROW_HALF_MIRROR = 0x141;

template <int CTRL, int BANK = 0xf>
__device__ __forceinline__ int mov_i(int old, int v) { return __builtin_amdgcn_update_dpp(old, v, CTRL, 0xf, BANK, false); }
template <int CTRL, int BANK = 0xf>
__device__ __forceinline__ double mov_d(double old, double v) {
  return __hiloint2double(mov_i<CTRL, BANK>(__double2hiint(old), __double2hiint(v)),
                          mov_i<CTRL, BANK>(__double2loint(old), __double2loint(v)));
}
// value of lane (i ^ 1 | 2 | 4 | 8) inside each row of 16 lanes
__device__ __forceinline__ double xor1(double v) { return mov_d<QP_XOR1>(v, v); }
__device__ __forceinline__ double xor2(double v) { return mov_d<QP_XOR2>(v, v); }
__device__ __forceinline__ double xor4(double v) {                  // banks 0,2 read +4, banks 1,3 read -4
  double r = mov_d<ROW_SHL4, 0x5>(v, v);
  return mov_d<ROW_SHR4, 0xa>(r, v);
}
__device__ __forceinline__ double xor8(double v) { return mov_d<ROW_ROR8>(v, v); }
__device__ __forceinline__ int xor1(int v) { return mov_i<QP_XOR1>(v, v); }
__device__ __forceinline__ int xor2(int v) { return mov_i<QP_XOR2>(v, v); }
__device__ __forceinline__ int xor4(int v) { int r = mov_i<ROW_SHL4, 0x5>(v, v); return mov_i<ROW_SHR4, 0xa>(r, v); }
__device__ __forceinline__ int xor8(int v) { return mov_i<ROW_ROR8>(v, v); }

__device__ __forceinline__ double rl_d(double v, int lane) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
// all-lanes (wave-uniform) reductions over the 64 lanes
__device__ __forceinline__ double wave_sum(double v) {
  v += xor1(v); v += xor2(v); v += xor4(v); v += xor8(v);            // every lane holds its row's sum
  return (rl_d(v, 0) + rl_d(v, 16)) + (rl_d(v, 32) + rl_d(v, 48));
}
__device__ __forceinline__ double wave_max(double v) {
  v = fmax(v, xor1(v)); v = fmax(v, xor2(v)); v = fmax(v, xor4(v)); v = fmax(v, xor8(v));
  return fmax(fmax(rl_d(v, 0), rl_d(v, 16)), fmax(rl_d(v, 32), rl_d(v, 48)));
}
__device__ __forceinline__ int wave_min(int v) {
  v = min(v, xor1(v)); v = min(v, xor2(v)); v = min(v, xor4(v)); v = min(v, xor8(v));
  return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
             min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}

// few-ulp reciprocal / reciprocal square root: hardware estimate + two Newton steps (~10 dependent instructions instead of the
// ~35 of an IEEE division / ~30 of sqrt). For scalars on a latency chain whose consumers tolerate a few ulp.
__device__ __forceinline__ double fast_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}
__device__ __forceinline__ double fast_rsqrt(double x) {
  double r = __builtin_amdgcn_rsq(x);
  r = r * fma(-0.5 * x * r, r, 1.5);
  r = r * fma(-0.5 * x * r, r, 1.5);
  return r;
}

}  // namespace nd4dpp
