// kv_pack.h — the operand images attn_bwd_kv_kernel reads (attn_bwd_kv.hip), shared by the launch that packs them on its own
// (attn_bwd_kv_pack_kernel) and by the row-block backward kernels that leave the dO part behind as they produce dO (rowblock.hip, round 6).
//
// Per problem (shared K/V: one per scene, rows r = 4 q + h; per-head K/V: one per (scene, head), rows = queries) and 32-row tile:
//   [3 kinds][4 subs][hi, lo][64 lanes] x 16 B.  lane = (l31 = lane & 31, g = lane >> 5), e = 0 .. 7:
//   kind 0 (sub = s):         A of dP~ = dO V^T:   dO[r0 + l31][16 s + 8 g + e]
//   kind 1 (sub = 2 mt + t):  A of dV^T = dO^T P~: dO[r0 + kv_row(8 t + e, g)][32 mt + l31]
//   kind 2:                   A of dK^T = q^T dS:  as kind 1 with q
// (kv_row: the tile row that accumulator register 8 t + e of lane group g holds.)
#pragma once
#include "attn_common.h"

namespace vdetr {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kKvOperandUnits = 2 * kWave;      // one packed operand: (hi, lo) x 64 lanes, 16 B each
constexpr int kKvTileUnits = 3 * 4 * kKvOperandUnits;

// row (inside a 32 x 32 tile) of accumulator register v in lane group g = lane >> 5
__device__ __forceinline__ constexpr int kv_row(int v, int g) { return (v & 3) + 8 * (v >> 2) + 4 * g; }

__device__ __forceinline__ void kv_split8(const float (&x)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const __bf16 h = (__bf16)x[e];
    hi[e] = h;
    lo[e] = (__bf16)(x[e] - (float)h);
  }
}

__device__ __forceinline__ void kv_store_unit(uint4* pack, long tile, int kind, int sub, int lane, const float (&x)[8]) {
  bf16x8 hi, lo;
  kv_split8(x, hi, lo);
  uint4* dst = pack + ((tile * 3 + kind) * 4 + sub) * kKvOperandUnits;
  dst[lane] = __builtin_bit_cast(uint4, hi);
  dst[kWave + lane] = __builtin_bit_cast(uint4, lo);
}

// What a launch that has just produced 16 rows of dO (one scene: 16 consecutive queries q0 .., all 4 heads x 64 channels, in an LDS
// tile `xs` of `stride` floats per row) leaves for the key-side pass of the attention that consumes it: the kind-0 / kind-1 images of
// those rows, delta = rowsum(dO * O), and max |dO row|^2 (aux word 0).  256 threads; nQ a multiple of 32, q0 of 16.
struct KvEmit {
  uint4* pack;       // the attention call's image buffer
  float* delta;      // shared K/V: [nQ][4]; per head: [4][nQ]
  const float* out;  // O [nQ][256], the attention's saved output
  unsigned* aux;     // or nullptr
  int per_head, nQ;
};
__device__ __forceinline__ void kv_emit_rows16(const KvEmit& E, const float* xs, int stride, int q0, int tid, float* wmax) {
  const int NT = E.per_head ? E.nQ / 32 : E.nQ / 8;
  // ---- delta and |dO row|^2: thread = ((q, h) pair, quarter of its 64 channels) ----
  {
    const int pair = tid >> 2, quarter = tid & 3, ql = pair >> 2, h = pair & 3;
    const float* orow = E.out + ((size_t)(q0 + ql) * 4 + h) * kDh + 16 * quarter;
    const float* drow = xs + ql * stride + h * kDh + 16 * quarter;
    float s = 0.f, n2 = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x4 o = *reinterpret_cast<const f32x4*>(orow + 4 * i);
      const f32x4 g = *reinterpret_cast<const f32x4*>(drow + 4 * i);
#pragma unroll
      for (int e = 0; e < 4; ++e) { s = fmaf(g[e], o[e], s); n2 = fmaf(g[e], g[e], n2); }
    }
    s += dpp_f32<kDppQuadXor1>(s); s += dpp_f32<kDppQuadXor2>(s);
    n2 += dpp_f32<kDppQuadXor1>(n2); n2 += dpp_f32<kDppQuadXor2>(n2);
    if (quarter == 0) E.delta[E.per_head ? (size_t)h * E.nQ + q0 + ql : (size_t)(q0 + ql) * 4 + h] = s;
    if (E.aux) {
      for (int m = 32; m >= 1; m >>= 1) n2 = fmaxf(n2, __shfl_xor(n2, m, 64));
      if ((tid & 63) == 0) wmax[tid >> 6] = n2;
    }
  }
  // ---- images: 1024 (unit, lane) pairs, four per thread ----
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int id = tid + 256 * u;
    float x[8];
    if (!E.per_head) {  // two tiles of 8 queries x 4 heads
      const int tl = id >> 9, unit = (id >> 6) & 7, lane = id & 63, kind = unit >> 2, sub = unit & 3, l31 = lane & 31, g = lane >> 5;
      if (kind == 0) {
        const float* src = xs + (tl * 8 + (l31 >> 2)) * stride + (l31 & 3) * kDh + 16 * sub + 8 * g;
        const f32x4 a = *reinterpret_cast<const f32x4*>(src), b = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { x[e] = a[e]; x[4 + e] = b[e]; }
      } else {
        const int d = 32 * (sub >> 1) + l31, t = sub & 1;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int row = kv_row(8 * t + e, g);
          x[e] = xs[(tl * 8 + (row >> 2)) * stride + (row & 3) * kDh + d];
        }
      }
      kv_store_unit(E.pack, (long)(q0 >> 3) + tl, kind, sub, lane, x);
    } else {  // per head: half a tile (16 queries) of each of the 4 problems
      const int head = id >> 8, rem = id & 255, half = (q0 >> 4) & 1;
      const long tile = (long)head * NT + (q0 >> 5);
      if (rem < 128) {  // kind 0: the 32 lanes of each sub whose row lies in this half
        const int sub = rem >> 5, j = rem & 31, rl = j & 15, g = j >> 4, lane = 16 * half + rl + 32 * g;
        const float* src = xs + rl * stride + head * kDh + 16 * sub + 8 * g;
        const f32x4 a = *reinterpret_cast<const f32x4*>(src), b = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { x[e] = a[e]; x[4 + e] = b[e]; }
        kv_store_unit(E.pack, tile, 0, sub, lane, x);
      } else {  // kind 1: the two subs (t = half) whose contraction slots are this half's rows
        const int mt = (rem - 128) >> 6, lane = rem & 63, l31 = lane & 31, g = lane >> 5, d = 32 * mt + l31;
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = xs[(kv_row(8 * half + e, g) - 16 * half) * stride + head * kDh + d];
        kv_store_unit(E.pack, tile, 1, 2 * mt + half, lane, x);
      }
    }
  }
  if (E.aux) {
    __syncthreads();  // one atomic per workgroup
    if (tid == 0) atomicMax(E.aux, __float_as_uint(fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]))));
  }
}

}  // namespace vdetr
