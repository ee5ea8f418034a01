// rowblock.h — the row-block kernels' shared device helpers (rowblock.hip, rowblock_pos.hip): a 16 x 256 activation tile in LDS,
// the weight ring of a [16 x 256] x [256 x 256] product on v_mfma_f32_16x16x4_f32, dropout streams, LayerNorm row statistics.
#pragma once
#include "attn_common.h"

namespace vdetr {

constexpr int kRbRows = 16;
constexpr int kRbC = 256;
constexpr int kRbThreads = 256;
constexpr int kRbDepth = 6;  // weight tiles in flight per wave (steps of 16 matrix instructions): one wave per SIMD, registers to spare
constexpr int kRbStride = kRbC + 4;  // floats per LDS row: 16 rows x 1040 B land on 16 different 16-byte slots

struct RbDrop {  // a dropout stream (add_ln.hip: LnRng / bn_act.hip: BnRng): keep iff 16-bit draw >= thresh
  unsigned seed_lo, seed_hi, off_lo, off_hi, thresh;
  float scale;
};
__device__ __forceinline__ RbDrop rb_drop(float p, unsigned long long seed, unsigned long long offset, const uint64_t* rng) {
  RbDrop r;
  unsigned long long s = seed, o = offset;
  if (rng) { s ^= rng[0]; o += rng[1]; }
  r.seed_lo = (unsigned)s; r.seed_hi = (unsigned)(s >> 32); r.off_lo = (unsigned)o; r.off_hi = (unsigned)(o >> 32);
  r.thresh = 0; r.scale = 1.f;
  if (p > 0.f) {
    int t = (int)((double)p * 65536.0 + 0.5);
    t = t < 1 ? 1 : (t > 65535 ? 65535 : t);
    r.thresh = (unsigned)t;
    r.scale = 65536.f / (float)(65536 - t);
  }
  return r;
}
// the residual blocks' mask (add_ln.hip: ln_rowkey / ln_keep4): channels 4 j .. 4 j + 3 of `row`
__device__ __forceinline__ void rb_keep4_ln(const RbDrop& g, int row, int j, bool (&keep)[4]) {
  if (!g.thresh) { keep[0] = keep[1] = keep[2] = keep[3] = true; return; }
  unsigned k = fmix32(((unsigned)row * 0x9E3779B1u + g.off_lo) ^ g.seed_lo);
  k = fmix32(k ^ (0x27D4EB2Fu + g.off_hi) ^ g.seed_hi);
  const unsigned x = fmix32(k ^ ((unsigned)j * 0x165667B1u));
  const unsigned y = fmix32(x + 0x9E3779B9u);
  keep[0] = (x & 0xFFFFu) >= g.thresh; keep[1] = (x >> 16) >= g.thresh;
  keep[2] = (y & 0xFFFFu) >= g.thresh; keep[3] = (y >> 16) >= g.thresh;
}
// the FFN activation's mask (bn_act.hip: relu_dropout_fwd_kernel): float4 number i of the flat tensor
__device__ __forceinline__ void rb_keep4_act(const RbDrop& g, long i, bool (&keep)[4]) {
  if (!g.thresh) { keep[0] = keep[1] = keep[2] = keep[3] = true; return; }
  unsigned r0 = fmix32(((unsigned)i * 0x9E3779B1u + g.off_lo) ^ g.seed_lo ^ ((unsigned)(i >> 32) * 0x27D4EB2Fu));
  r0 = fmix32(r0 ^ g.seed_hi ^ g.off_hi);
  const unsigned r1 = fmix32(r0 + 0x9E3779B9u);
  keep[0] = (r0 & 0xFFFFu) >= g.thresh; keep[1] = (r0 >> 16) >= g.thresh;
  keep[2] = (r1 & 0xFFFFu) >= g.thresh; keep[3] = (r1 >> 16) >= g.thresh;
}

// ---- the 16 x 256 activation tile: global -> LDS (row-major, padded), LDS -> the 64 A-operand registers of a lane ----------
// Rows are numbered as the decoder's sequence-first tensors lay them out: row = q * B + b.  The attention cores read and write
// batch-first tensors [B, nQ, C]: `rb_bmajor` is the row of (q, b) there (identity for one scene).
__device__ __forceinline__ int rb_bmajor(int row, int B, int nQ) { return B == 1 ? row : (row % B) * nQ + row / B; }
__device__ __forceinline__ void rb_stage_rows(const float* __restrict__ src, int row0, int rows, int B, bool bmajor, float* xs, int tid,
                                              const float* __restrict__ add = nullptr, float* sum_out = nullptr, float* copy_out = nullptr) {
  constexpr int kPer = kRbRows * kRbC / 4 / kRbThreads;  // 4 float4 per thread: all requested before the first is used
  f32x4 v[kPer], p[kPer];
  int rowv[kPer];
#pragma unroll
  for (int u = 0; u < kPer; ++u) {
    const int e = tid + u * kRbThreads, r = e >> 6, c4 = e & 63;
    const int row = min(row0 + r, rows - 1);  // rows past the end are computed on a copy of the last row and not stored
    rowv[u] = row;
    const int srow = bmajor ? rb_bmajor(row, B, rows / B) : row;
    v[u] = reinterpret_cast<const f32x4*>(src + (size_t)srow * kRbC)[c4];
    p[u] = add ? reinterpret_cast<const f32x4*>(add + (size_t)row * kRbC)[c4] : f32x4{0.f, 0.f, 0.f, 0.f};  // (sequence-first, like the rows)
  }
#pragma unroll
  for (int u = 0; u < kPer; ++u) {
    const int e = tid + u * kRbThreads, r = e >> 6, c4 = e & 63;
    const f32x4 t = v[u] + p[u];
    const bool live = row0 + r < rows;
    if (sum_out && live) reinterpret_cast<f32x4*>(sum_out + (size_t)rowv[u] * kRbC)[c4] = t;    // t + pos
    if (copy_out && live) reinterpret_cast<f32x4*>(copy_out + (size_t)rowv[u] * kRbC)[c4] = t;  // the rows in sequence-first order
    *reinterpret_cast<f32x4*>(xs + r * kRbStride + 4 * c4) = t;
  }
}
__device__ __forceinline__ void rb_load_a(const float* xs, int lane, float (&a)[64]) {
  const int i = lane & 15, kg = lane >> 4;
#pragma unroll
  for (int m = 0; m < 16; ++m) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(xs + i * kRbStride + 16 * m + 4 * kg);
#pragma unroll
    for (int e = 0; e < 4; ++e) a[4 * m + e] = v[e];
  }
}
// acc[nt][r] += sum_k X[4 g + r][k] M[k][col0 + 4 c + nt]   (M [256][256] row-major: a W^T image for y = x W^T, W itself for
// dX = dY W); lane = (g = lane >> 4, c = lane & 15).  The weights of steps m + 1 .. m + kRbDepth - 1 are in flight while step m's
// 16 matrix instructions run; the scheduling barriers pin that order (left alone, the compiler sinks every load to just above its
// first use and the loop pays a memory round trip per step).  rb_w_begin issues the first kRbDepth - 1 steps and returns: a kernel
// calls it BEFORE the epilogue of the product in front, so that a product starts on weights that have arrived (one wave per SIMD:
// nothing else hides that round trip).
struct RbRing {
  f32x4 b[kRbDepth][4];
};
__device__ __forceinline__ const float* rb_w_ptr(const float* __restrict__ M, int col0, int lane) {
  return M + (size_t)(4 * (lane >> 4)) * kRbC + col0 + 4 * (lane & 15);  // + (16 m + e) rows
}
__device__ __forceinline__ void rb_w_begin(const float* __restrict__ M, int col0, int lane, RbRing& R) {
  const float* wp = rb_w_ptr(M, col0, lane);
#pragma unroll
  for (int d = 0; d < kRbDepth - 1; ++d)
#pragma unroll
    for (int e = 0; e < 4; ++e) R.b[d][e] = *reinterpret_cast<const f32x4*>(wp + (size_t)(16 * d + e) * kRbC);
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void rb_w_run(const float (&a)[64], const float* __restrict__ M, int col0, int lane, RbRing& R, f32x4 (&acc)[4]) {
  const float* wp = rb_w_ptr(M, col0, lane);
#pragma unroll
  for (int m = 0; m < 16; ++m) {
    if (m + kRbDepth - 1 < 16) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        R.b[(m + kRbDepth - 1) % kRbDepth][e] = *reinterpret_cast<const f32x4*>(wp + (size_t)(16 * (m + kRbDepth - 1) + e) * kRbC);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * m + e], R.b[m % kRbDepth][e][nt], acc[nt], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}
__device__ __forceinline__ void rb_zero(f32x4 (&acc)[4]) {
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
}
__device__ __forceinline__ f32x4 rb_ld4(const float* p, int row, int colq) { return *reinterpret_cast<const f32x4*>(p + (size_t)row * kRbC + colq); }
__device__ __forceinline__ void rb_st4(float* p, int row, int colq, const f32x4& v) { *reinterpret_cast<f32x4*>(p + (size_t)row * kRbC + colq) = v; }
__device__ __forceinline__ f32x4 rb_ldv(const float* p, int colq) { return p ? *reinterpret_cast<const f32x4*>(p + colq) : f32x4{0.f, 0.f, 0.f, 0.f}; }

// accumulators -> per-row float4 of the four adjacent columns (row 4 g + r, columns col0 + 4 c ..)
__device__ __forceinline__ f32x4 rb_row(const f32x4 (&acc)[4], int r) { return f32x4{acc[0][r], acc[1][r], acc[2][r], acc[3][r]}; }

// LayerNorm statistics of the workgroup's 16 rows from each lane's 16 values (4 rows x 4 columns): DPP row sums, then the 4
// waves through LDS.  red: [2][4 waves][16 rows] floats.  Two-pass (mean, then centred squares), as add_ln.hip does.
__device__ __forceinline__ void rb_row_stats(const f32x4 (&y)[4], float* red, int w, int lane, float eps, float (&mean)[4], float (&rstd)[4]) {
  const int g = lane >> 4, c = lane & 15;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float s = row_allsum_f32((y[r][0] + y[r][1]) + (y[r][2] + y[r][3]));
    if (c == 0) red[w * 16 + 4 * g + r] = s;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r)
    mean[r] = ((red[4 * g + r] + red[16 + 4 * g + r]) + (red[32 + 4 * g + r] + red[48 + 4 * g + r])) * (1.f / kRbC);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) { const float t = y[r][e] - mean[r]; q += t * t; }
    q = row_allsum_f32(q);
    if (c == 0) red[64 + w * 16 + 4 * g + r] = q;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r)
    rstd[r] = rsqrtf(((red[64 + 4 * g + r] + red[80 + 4 * g + r]) + (red[96 + 4 * g + r] + red[112 + 4 * g + r])) * (1.f / kRbC) + eps);
}

// ---- the same tile out of the key-split partials of the attention forward (vdetr_attn_fwd_parts_f32) -------------------------------
// out = sum_s exp(lse_s - M) o_s / sum_s exp(lse_s - M): attn_fwd_combine_kernel's arithmetic in its order (bit-identical), done by the
// launch that reads the rows anyway.  Row (q, b) of the tile is attention row b * nQ + q; its 256 columns are 4 heads x 64.  The
// merged rows and their log-sum-exp are written where the merge launch would have left them (the backward reads both).
struct RbParts {
  const float* part_o;    // [ks][rows4][64]
  const float* part_lse;  // [ks][rows4]
  float* out;             // [B, nQ, 256]
  float* lse;             // [B, nQ, 4]
  long rows4;
  int ks;
};
template <int KS>  // > 0: that many partials, unrolled (every load in flight before the first use); 0: P.ks of them, one after the other
__device__ __forceinline__ void rb_stage_parts(const RbParts& P, int row0, int rows, int B, float* xs, int tid) {
  constexpr int kPer = kRbRows * kRbC / 4 / kRbThreads;
  constexpr int kS = KS > 0 ? KS : 1;
  f32x4 o[kPer][kS];
  float ls[kPer][kS];
  size_t prow[kPer];
#pragma unroll
  for (int u = 0; u < kPer; ++u) {
    const int e = tid + u * kRbThreads, r = e >> 6, c4 = e & 63;
    const int row = min(row0 + r, rows - 1);
    prow[u] = (size_t)rb_bmajor(row, B, rows / B) * 4 + (c4 >> 4);
    if constexpr (KS > 0) {
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        ls[u][s] = P.part_lse[(size_t)s * P.rows4 + prow[u]];
        o[u][s] = reinterpret_cast<const f32x4*>(P.part_o + ((size_t)s * P.rows4 + prow[u]) * kDh)[c4 & 15];
      }
    }
  }
#pragma unroll
  for (int u = 0; u < kPer; ++u) {
    const int e = tid + u * kRbThreads, r = e >> 6, c4 = e & 63;
    float M = kNegBig, L = 0.f;
    f32x4 val = {0.f, 0.f, 0.f, 0.f};
    if constexpr (KS > 0) {
#pragma unroll
      for (int s = 0; s < KS; ++s) M = fmaxf(M, ls[u][s]);
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const float f = __expf(ls[u][s] - M);
        L += f;
#pragma unroll
        for (int i = 0; i < 4; ++i) val[i] += f * o[u][s][i];
      }
    } else {
      for (int s = 0; s < P.ks; ++s) M = fmaxf(M, P.part_lse[(size_t)s * P.rows4 + prow[u]]);
      for (int s = 0; s < P.ks; ++s) {
        const float f = __expf(P.part_lse[(size_t)s * P.rows4 + prow[u]] - M);
        const f32x4 ov = reinterpret_cast<const f32x4*>(P.part_o + ((size_t)s * P.rows4 + prow[u]) * kDh)[c4 & 15];
        L += f;
#pragma unroll
        for (int i = 0; i < 4; ++i) val[i] += f * ov[i];
      }
    }
    f32x4 t;
#pragma unroll
    for (int i = 0; i < 4; ++i) t[i] = L > 0.f ? val[i] / L : 0.f;
    if (row0 + r < rows) {
      reinterpret_cast<f32x4*>(P.out + prow[u] * kDh)[c4 & 15] = t;  // (b, q, h) rows of 64 = [B, nQ, 256]
      if ((c4 & 15) == 0) P.lse[prow[u]] = L > 0.f ? M + __logf(L) : kNegBig;
    }
    *reinterpret_cast<f32x4*>(xs + r * kRbStride + 4 * c4) = t;
  }
}

typedef vdetr_rb_qkv_desc RbQkvArgs;

}  // namespace vdetr

#define RB_ALIGNED(p) ((((uintptr_t)(p)) & 15) == 0)
