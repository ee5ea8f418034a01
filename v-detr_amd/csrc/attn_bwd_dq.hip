// attn_bwd_dq.hip — dQ = scale * dS K of the attention backward as a ROW-OWNER kernel, exact fp32 products, gfx950.
//
// Reference: the autograd of `attn @ v` / `q @ k^T` in GlobalShareCrossAttention.forward (models/vdetr_transformer.py:733-757)
// and nn.MultiheadAttention (:468) — there a batched matmul of the [.., nQ, nK] gradient with K.  The library's GEMM for
// this shape (a 4096 x 4096 or 4 x 1024 x 1024 operand against a 64-wide K: N = 64) takes 28-33 us / 25-29 us at the model's
// size, 2.2 TB/s resp. 19 TFLOP/s: K is tiny (1 MB, L2-resident), the cost is streaming dS once, and at N = 64 the
// library's tiles leave most of the chip's matrix time unused.  Here a workgroup OWNS 16 rows of dS (one per row of a
// v_mfma_f32_16x16x4_f32 tile) and all 64 output columns; its four waves split the KEYS (64-key chunks w, w + 4, ...), so
// nothing is loaded twice inside a workgroup, and their partial tiles meet once in LDS at the end.
//   A operand (dS): lane (i = lane & 15, g = lane >> 4) reads 16 consecutive floats dS[row0 + i, kc + 16 g ..] as four
//                   16-byte loads; instruction s of the chunk contracts over keys kc + 16 g + s (a permuted contraction
//                   order: any order works as long as both operands use it).
//   B operand (K) : the same lane reads K[kc + 16 g + s, 4 n .. 4 n + 3] as ONE 16-byte load and feeds its four components
//                   to four instructions — tile c holds the output columns 4 n + c, so a lane's four accumulators of a
//                   row are four CONSECUTIVE floats of dQ: the epilogue stores 16 bytes per lane and row.
// Matrix time: R/16 x 4 x nK/4 instructions of 32 cycles over 1024 SIMDs: 13.7 us at 4096 x 4096, 3.4 us at 4 x 1024 x 1024.
// Measured (HIP events, alone): per-head 4 x 1024 x 1024: 11.7 us against the library's 19.2 — used (the query self-attention).
// Shared K/V 4096 x 4096: 31.2 us against the library's 26.0 (four scenes: 147 against 72): every 16-row workgroup reads ALL of
// K from L2 (256 x 1 MB per launch, 8 TB/s), where the library shares a K tile among many more rows through LDS; with
// R = 4096 rows there are only 256 row tiles, so larger tiles would leave CUs idle — the host keeps the library there.
#include "attn_common.h"

namespace vdetr {

constexpr int kDqWaves = 4;
constexpr int kDqThreads = kDqWaves * kWave;
constexpr int kDqChunk = 64;  // keys per chunk: 16 instructions x 4 key groups

struct DqParams {
  const float* ds;   // [P][R][nK]
  const float* k;    // problem p, key j, column d: k[p_off(p) + j * k_row + d]
  float* dq;         // problem p, row r, column d: dq[q_off(p) + r * q_row + d]
  int R, nK, H, perhead;
  int k_row, q_row;
  long k_batch, q_batch;  // floats between scenes
  float alpha;
};

__global__ __launch_bounds__(kDqThreads) void attn_bwd_dq_kernel(DqParams P) {
  __shared__ __attribute__((aligned(16))) float red[kDqWaves - 1][16][64 + 4];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int p = blockIdx.y;
  const int row0 = blockIdx.x * 16;
  // problem p: scene b (shared K/V) or (scene b, head h) (per head: K columns h*64.., dQ columns h*64..)
  const int b = P.perhead ? p / P.H : p, h = P.perhead ? p % P.H : 0;
  const float* kbase = P.k + (size_t)b * P.k_batch + (P.perhead ? h * kDh : 0);
  float* qbase = P.dq + (size_t)b * P.q_batch + (P.perhead ? h * kDh : 0);
  const float* dsrow = P.ds + ((size_t)p * P.R + min(row0 + i, P.R - 1)) * (size_t)P.nK;
  const int nchunks = (P.nK + kDqChunk - 1) / kDqChunk;
  const bool aligned = (P.nK & 3) == 0;  // 16-byte loads of a dS row need nK % 4 == 0 (rows start at multiples of nK floats)

  f32x4 acc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};

  struct Ops {
    f32x4 a[4];
    f32x4 kb[16];
  };
  auto fetch = [&](int chunk, Ops& o) {
    const int k0 = chunk * kDqChunk + 16 * g;
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const int kk = k0 + 4 * s4;
      if (aligned && kk + 3 < P.nK) {
        o.a[s4] = *reinterpret_cast<const f32x4*>(dsrow + kk);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) o.a[s4][e] = kk + e < P.nK ? dsrow[kk + e] : 0.f;  // keys past the end contribute nothing
      }
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int kk = min(k0 + s, P.nK - 1);  // (clamped rows meet a zero A operand)
      o.kb[s] = *reinterpret_cast<const f32x4*>(kbase + (size_t)kk * P.k_row + 4 * i);
    }
  };
  auto multiply = [&](const Ops& o) {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const float x = o.a[s >> 2][s & 3];
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, o.kb[s][c], acc[c], 0, 0, 0);
    }
  };
  // two operand sets in turn: the next chunk's 20 loads are in flight under the current chunk's 64 instructions, no copies.
  // (Four chunks of dS in flight per wave instead of one: 31 -> 38 us at 4096 x 4096 — the launch is not waiting for HBM.)
  Ops o0, o1;
  int chunk = wv;
  if (chunk < nchunks) fetch(chunk, o0);
  while (chunk < nchunks) {
    if (chunk + kDqWaves < nchunks) fetch(chunk + kDqWaves, o1);
    multiply(o0);
    chunk += kDqWaves;
    if (chunk >= nchunks) break;
    if (chunk + kDqWaves < nchunks) fetch(chunk + kDqWaves, o0);
    multiply(o1);
    chunk += kDqWaves;
  }
  // accumulator layout: register r of lane (n = lane & 15, g) is row 4 g + r, column (tile c) 4 n + c
  if (wv > 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      *reinterpret_cast<f32x4*>(&red[wv - 1][4 * g + r][4 * i]) = f32x4{acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
  }
  __syncthreads();
  if (wv == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      f32x4 v = {acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
#pragma unroll
      for (int w2 = 0; w2 < kDqWaves - 1; ++w2) v += *reinterpret_cast<const f32x4*>(&red[w2][4 * g + r][4 * i]);  // fixed order
      const int row = row0 + 4 * g + r;
      if (row < P.R) *reinterpret_cast<f32x4*>(qbase + (size_t)row * P.q_row + 4 * i) = v * P.alpha;
    }
  }
}

}  // namespace vdetr

using namespace vdetr;

// dq [B, nQ, H*64] = scale * dS K.  ds: [B, nQ, H, nK] (shared K/V: row (q, h) against the scene's one K) or [B, H, nQ, nK]
// (per head: head h against columns h*64.. of K [B, nK, H*64]); the layouts vdetr_attn_bwd_kv_f32 writes.
extern "C" int vdetr_attn_bwd_dq_f32(const vdetr_attn_desc* d, const float* ds, const float* k, float* dq, vdetr_stream_t stream) {
  VDETR_REQUIRE(d && ds && k && dq, "attn_bwd_dq: null pointer");
  VDETR_REQUIRE(d->B > 0 && d->H > 0 && d->nQ > 0 && d->nK > 0, "attn_bwd_dq: empty problem B=%d H=%d nQ=%d nK=%d", d->B, d->H, d->nQ, d->nK);
  const bool perhead = d->kind == VDETR_ATTN_PER_HEAD;
  VDETR_REQUIRE(perhead || d->kind == VDETR_ATTN_SHARED_KV, "attn_bwd_dq: kind %d", d->kind);
  const int dense = perhead ? d->H * kDh : kDh;
  const int k_row = d->k_row_stride ? d->k_row_stride : dense;
  VDETR_REQUIRE(k_row >= dense && k_row % 4 == 0, "attn_bwd_dq: k_row_stride %d (a multiple of 4, at least %d)", k_row, dense);
  VDETR_REQUIRE(((uintptr_t)k & 15) == 0 && ((uintptr_t)dq & 15) == 0 && ((uintptr_t)ds & 15) == 0, "attn_bwd_dq: 16-byte aligned operands");
  DqParams P;
  P.ds = ds; P.k = k; P.dq = dq;
  P.H = d->H; P.perhead = perhead ? 1 : 0;
  P.R = perhead ? d->nQ : d->nQ * d->H;
  P.nK = d->nK;
  P.k_row = k_row; P.k_batch = (long)d->nK * k_row;
  P.q_row = perhead ? d->H * kDh : kDh;
  P.q_batch = (long)d->nQ * d->H * kDh;
  P.alpha = d->scale;
  const long probs = perhead ? (long)d->B * d->H : d->B;
  VDETR_REQUIRE(probs <= 65535, "attn_bwd_dq: %ld problems (at most 65535)", probs);
  hipLaunchKernelGGL(attn_bwd_dq_kernel, dim3((P.R + 15) / 16, (unsigned)probs), dim3(kDqThreads), 0, (hipStream_t)stream, P);
  return check_launch("attn_bwd_dq");
}
