// wave.h — 64-lane wavefront primitives (DPP / permlane-swap based; no LDS traffic).
#pragma once
#include "common.h"

namespace vdetr {

// DPP control words (LLVM AMDGPU DppCtrl encoding)
constexpr int kDppQuadXor1 = 0xB1;     // quad_perm:[1,0,3,2]
constexpr int kDppQuadXor2 = 0x4E;     // quad_perm:[2,3,0,1]
constexpr int kDppRowHalfMirror = 0x141;
constexpr int kDppRowMirror = 0x140;

template <int CTRL>
__device__ __forceinline__ unsigned dpp_u32(unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xF, 0xF, false);
}
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
  return __uint_as_float(dpp_u32<CTRL>(__float_as_uint(v)));
}
template <int CTRL>
__device__ __forceinline__ unsigned long long dpp_u64(unsigned long long v) {
  const unsigned lo = dpp_u32<CTRL>((unsigned)v), hi = dpp_u32<CTRL>((unsigned)(v >> 32));
  return ((unsigned long long)hi << 32) | lo;
}

// The same exchange with bound_ctrl:0 and a zero `old` operand: hipcc then folds the permutation into the consuming
// VALU op (one `v_max_u32_dpp` per stage instead of v_mov / s_nop / v_mov_dpp / v_max).  A source lane that is
// switched off reads as 0 instead of the lane's own value, so these are for code that runs with every lane of the
// wave active (the patterns used here have no out-of-row sources).
template <int CTRL>
__device__ __forceinline__ unsigned dppz_u32(unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}

// Cross-row / cross-half exchange.  permlane16_swap(vdst, src) swaps the odd 16-lane rows of vdst with
// the even rows of src; permlane32_swap swaps lanes 32-63 of vdst with lanes 0-31 of src.  Called with
// vdst = src = v the two results are A' = [r0,r0,r2,r2] / [lo,lo] and B' = [r1,r1,r3,r3] / [hi,hi], so
// op(A', B') is the all-reduce over the row pair / the two halves for any commutative op.
struct pair_u32 { unsigned a, b; };
__device__ __forceinline__ pair_u32 xrow16(unsigned v) {
  auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
  return {r[0], r[1]};
}
__device__ __forceinline__ pair_u32 xhalf32(unsigned v) {
  auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
  return {r[0], r[1]};
}

__device__ __forceinline__ unsigned long long umax64(unsigned long long a, unsigned long long b) {
  return a > b ? a : b;
}

// all-reduce max over the 16 lanes of a DPP row
__device__ __forceinline__ unsigned long long row_allmax_u64(unsigned long long v) {
  v = umax64(v, dpp_u64<kDppQuadXor1>(v));
  v = umax64(v, dpp_u64<kDppQuadXor2>(v));
  v = umax64(v, dpp_u64<kDppRowHalfMirror>(v));
  v = umax64(v, dpp_u64<kDppRowMirror>(v));
  return v;
}
// all-reduce max over the whole wave
__device__ __forceinline__ unsigned long long wave_allmax_u64(unsigned long long v) {
  v = row_allmax_u64(v);
  {
    const pair_u32 lo = xrow16((unsigned)v), hi = xrow16((unsigned)(v >> 32));
    v = umax64(((unsigned long long)hi.a << 32) | lo.a, ((unsigned long long)hi.b << 32) | lo.b);
  }
  {
    const pair_u32 lo = xhalf32((unsigned)v), hi = xhalf32((unsigned)(v >> 32));
    v = umax64(((unsigned long long)hi.a << 32) | lo.a, ((unsigned long long)hi.b << 32) | lo.b);
  }
  return v;
}

// all-reduce min of a double over the whole wave (operands must not be NaN)
__device__ __forceinline__ double wave_allmin_f64(double v) {
  auto mn = [](double a, unsigned long long b) { return fmin(a, __longlong_as_double((long long)b)); };
  unsigned long long u = (unsigned long long)__double_as_longlong(v);
  v = mn(v, dpp_u64<kDppQuadXor1>(u));
  u = (unsigned long long)__double_as_longlong(v);
  v = mn(v, dpp_u64<kDppQuadXor2>(u));
  u = (unsigned long long)__double_as_longlong(v);
  v = mn(v, dpp_u64<kDppRowHalfMirror>(u));
  u = (unsigned long long)__double_as_longlong(v);
  v = mn(v, dpp_u64<kDppRowMirror>(u));
  u = (unsigned long long)__double_as_longlong(v);
  {
    const pair_u32 lo = xrow16((unsigned)u), hi = xrow16((unsigned)(u >> 32));
    v = fmin(__longlong_as_double((long long)(((unsigned long long)hi.a << 32) | lo.a)),
             __longlong_as_double((long long)(((unsigned long long)hi.b << 32) | lo.b)));
    u = (unsigned long long)__double_as_longlong(v);
  }
  {
    const pair_u32 lo = xhalf32((unsigned)u), hi = xhalf32((unsigned)(u >> 32));
    v = fmin(__longlong_as_double((long long)(((unsigned long long)hi.a << 32) | lo.a)),
             __longlong_as_double((long long)(((unsigned long long)hi.b << 32) | lo.b)));
  }
  return v;
}

__device__ __forceinline__ float row_allmax_f32(float v) {
  v = fmaxf(v, dpp_f32<kDppQuadXor1>(v));
  v = fmaxf(v, dpp_f32<kDppQuadXor2>(v));
  v = fmaxf(v, dpp_f32<kDppRowHalfMirror>(v));
  v = fmaxf(v, dpp_f32<kDppRowMirror>(v));
  return v;
}
__device__ __forceinline__ float row_allsum_f32(float v) {
  v += dpp_f32<kDppQuadXor1>(v);
  v += dpp_f32<kDppQuadXor2>(v);
  v += dpp_f32<kDppRowHalfMirror>(v);
  v += dpp_f32<kDppRowMirror>(v);
  return v;
}
// full-exec float forms for hot loops (see dppz_u32).  The sum folds into v_add_f32_dpp by itself; for the max hipcc
// would canonicalise the permuted operand first (v_mov_dpp + v_max v,v + v_max), so its four stages are written out:
// v_max_f32 of two ordinary (non signalling-NaN) floats needs no canonicalisation.  s_nop 1: the two wait states a DPP
// read of a VALU result needs, which nobody inserts inside inline asm.
__device__ __forceinline__ float row_allsum_f32_fx(float v) {
  v += __uint_as_float(dppz_u32<kDppQuadXor1>(__float_as_uint(v)));
  v += __uint_as_float(dppz_u32<kDppQuadXor2>(__float_as_uint(v)));
  v += __uint_as_float(dppz_u32<kDppRowHalfMirror>(__float_as_uint(v)));
  v += __uint_as_float(dppz_u32<kDppRowMirror>(__float_as_uint(v)));
  return v;
}
__device__ __forceinline__ float row_allmax_f32_fx(float v) {
  asm(
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf"
      : "+v"(v));
  return v;
}
__device__ __forceinline__ float wave_allmax_f32(float v) {
  v = row_allmax_f32(v);
  pair_u32 p = xrow16(__float_as_uint(v));
  v = fmaxf(__uint_as_float(p.a), __uint_as_float(p.b));
  p = xhalf32(__float_as_uint(v));
  return fmaxf(__uint_as_float(p.a), __uint_as_float(p.b));
}
__device__ __forceinline__ float wave_allmin_f32(float v) { return -wave_allmax_f32(-v); }
__device__ __forceinline__ float wave_allsum_f32(float v) {
  v = row_allsum_f32(v);
  pair_u32 p = xrow16(__float_as_uint(v));
  v = __uint_as_float(p.a) + __uint_as_float(p.b);
  p = xhalf32(__float_as_uint(v));
  return __uint_as_float(p.a) + __uint_as_float(p.b);
}

// 32-bit unsigned all-reduces: one DPP-fused VALU op per stage
__device__ __forceinline__ unsigned row_allmax_u32(unsigned v) {
  v = max(v, dpp_u32<kDppQuadXor1>(v));
  v = max(v, dpp_u32<kDppQuadXor2>(v));
  v = max(v, dpp_u32<kDppRowHalfMirror>(v));
  v = max(v, dpp_u32<kDppRowMirror>(v));
  return v;
}
__device__ __forceinline__ unsigned row_allmin_u32(unsigned v) {
  v = min(v, dpp_u32<kDppQuadXor1>(v));
  v = min(v, dpp_u32<kDppQuadXor2>(v));
  v = min(v, dpp_u32<kDppRowHalfMirror>(v));
  v = min(v, dpp_u32<kDppRowMirror>(v));
  return v;
}
// full-exec forms (see dppz_u32)
__device__ __forceinline__ unsigned row_allmax_u32_fx(unsigned v) {
  v = max(v, dppz_u32<kDppQuadXor1>(v));
  v = max(v, dppz_u32<kDppQuadXor2>(v));
  v = max(v, dppz_u32<kDppRowHalfMirror>(v));
  v = max(v, dppz_u32<kDppRowMirror>(v));
  return v;
}
__device__ __forceinline__ unsigned row_allmin_u32_fx(unsigned v) {
  v = min(v, dppz_u32<kDppQuadXor1>(v));
  v = min(v, dppz_u32<kDppQuadXor2>(v));
  v = min(v, dppz_u32<kDppRowHalfMirror>(v));
  v = min(v, dppz_u32<kDppRowMirror>(v));
  return v;
}
// Wave-wide max / min as a SCALAR (full exec): the row butterflies, then row_bcast:15 into rows 1 and 3 and
// row_bcast:31 into rows 2 and 3 leave the result in row 3; six DPP-fused ops and one v_readlane instead of the
// twelve VALU ops of the permlane-swap all-reduce.  (`old` = the op's identity: rows outside the row mask keep v.)
constexpr int kDppRowBcast15 = 0x142;
constexpr int kDppRowBcast31 = 0x143;
__device__ __forceinline__ unsigned wave_max_u32_s(unsigned v) {
  v = row_allmax_u32_fx(v);
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, kDppRowBcast15, 0xA, 0xF, false));
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, kDppRowBcast31, 0xC, 0xF, false));
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ unsigned wave_min_u32_s(unsigned v) {
  v = row_allmin_u32_fx(v);
  v = min(v, (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, kDppRowBcast15, 0xA, 0xF, false));
  v = min(v, (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, kDppRowBcast31, 0xC, 0xF, false));
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ unsigned wave_allmax_u32(unsigned v) {
  v = row_allmax_u32(v);
  pair_u32 p = xrow16(v);
  v = max(p.a, p.b);
  p = xhalf32(v);
  return max(p.a, p.b);
}
__device__ __forceinline__ unsigned wave_allmin_u32(unsigned v) {
  v = row_allmin_u32(v);
  pair_u32 p = xrow16(v);
  v = min(p.a, p.b);
  p = xhalf32(v);
  return min(p.a, p.b);
}

__device__ __forceinline__ float readlane_f32(float v, int lane) {
  return __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(v), lane));
}
// v with lane `lane` replaced by the wave-uniform value x: one v_writelane_b32.  (This clang has no writelane
// builtin.  Two different SGPR sources would break the one-constant-bus-read rule, so the lane select goes through
// M0; the s_nop covers the wait states a VALU-written data SGPR may need, inline asm is not scanned for hazards.)
__device__ __forceinline__ unsigned writelane_u32(unsigned v, unsigned x, int lane) {
  x = (unsigned)__builtin_amdgcn_readfirstlane((int)x);  // no-ops for SGPR values; an "s" operand left in a VGPR is not legalised
  lane = __builtin_amdgcn_readfirstlane(lane);
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 3\n\tv_writelane_b32 %0, %1, m0" : "+v"(v) : "s"(x), "s"(lane) : "m0");
  return v;
}
__device__ __forceinline__ unsigned readlane_u32(unsigned v, int lane) {
  return (unsigned)__builtin_amdgcn_readlane((int)v, lane);
}

}  // namespace vdetr
