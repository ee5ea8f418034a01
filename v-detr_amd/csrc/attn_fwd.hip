// attn_fwd.hip — fused attention forward for gfx950:  out = dropout(softmax(scale*q k^T + rpe + mask)) v
//
// Replaces the ~30 ATen kernels of GlobalShareCrossAttention.forward (vdetr_transformer.py:701-758: eight
// grid_sample passes over a [B,nQ,nK,3] tensor, a [B,H,nQ,nK] bias tensor, softmax, two matmuls) and the
// matmul/softmax/matmul of nn.MultiheadAttention / ShareSelfAttention (:468, :633-653) with ONE kernel.
//
// Tiling.  One MFMA tile is v_mfma_f32_16x16x4_f32: 16 score rows x 16 keys, exact fp32.
//   shared-KV kinds (the 3DV-RPE cross attention): K and V are shared by the 4 heads, so the 4 heads of a
//   query are 4 ROWS of the same tile: rows = 4 queries x 4 heads.  In the accumulator layout
//   (col = lane&15, row = 4*(lane>>4)+reg) lane (g,c) then owns exactly ONE (query g, key c) pair and its
//   4 registers are the 4 heads — which is the shape of the RPE lookup: the per-pair geometry (24 log2,
//   floor/frac, trilinear weights) is computed once per lane and the table cell read from LDS is one
//   16-byte float4 = the 4 heads.  No shuffles between the MFMA result and the bias.
//   per-head kind (nn.MultiheadAttention): rows = 16 queries of one head.
// A workgroup = 8 waves owns one row tile and a key range; wave w takes key tiles w, w+8, ... with a
// private online-softmax state, the 8 partial states are merged through LDS at the end.  256 workgroups
// for nQ=1024 (B=1): one per CU, which is also what the 128 KB LDS table image allows.
//   LDS: [8][T^3] float4 RPE table (128,000 B for T=10) + 8 x 1,280 B P-transpose pads.
// Contraction index order inside a tile is permuted (d = 16*(lane>>4)+s for QK^T, key = 4*(lane>>4)+s and
// d = 4*(lane&15)+t for PV) so that every operand fetch is a contiguous float4 per lane.
#include "attn_common.h"

namespace vdetr {

constexpr int kFwdThreads = 512;
constexpr int kFwdWaves = kFwdThreads / kWave;
constexpr int kPPad = 20;  // floats per row of the P transpose pad (16 + 4: keeps float4 alignment)

// BOX (RPE only): this instantiation handles the workgroups whose four queries are axis-aligned boxes (6 axis taps per
// pair, attn_common.h), the BOX = false one all the others; both are launched over the same grid and a workgroup exits
// at once when it belongs to the other kind (the two paths in one kernel needed > 256 VGPRs).
// BF16 (shared-KV kinds): q, k, v are bf16 in memory and QK^T / PV run on the bf16 matrix instructions
// (v_mfma_f32_16x16x32_bf16: one instruction per 32 of the 64 head dims; v_mfma_f32_16x16x16_bf16 for the 16 keys of a
// tile); scores, RPE bias, softmax, accumulators and the output stay fp32 (BASELINE config 4).  Same tiling, same lane
// layouts: the operand of lane (row/col = lane & 15, k-group = lane >> 4) is 8 (resp. 4) consecutive k instead of one.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short short4v __attribute__((ext_vector_type(4)));

// -DVDETR_FWD_SPLIT=1 (build option, OFF by default): fp32 operands on the bf16 matrix unit, x = hi + lo (bf16, round-to-nearest),
// products as the three leading cross terms (2^-16 per product, fp32 accumulate): QK^T is 6 instructions of 16 cycles instead of
// 16 of 32, PV 12 of 8 instead of 16 of 32 — and matrix time ADDS to VALU time on this chip.  Measured: forward
// 176 -> 161 us, captured C2 step 9.55 -> 9.44 ms; every output stays within 1e-3 of the oracle, but scores that are off by 1e-5
// instead of 1e-7 flip near-tie proposal selections downstream: the whole-model parity case with three ragged scenes then has 39
// token rows of the feature gradient outside its tolerance (tests/test_gpu_model.py).  The forward keeps exact fp32 products.
#ifndef VDETR_FWD_SPLIT
#define VDETR_FWD_SPLIT 0
#endif
constexpr bool kFwdSplit = VDETR_FWD_SPLIT != 0;
__device__ __forceinline__ void fwd_split8(const f32x4& x0, const f32x4& x1, bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const __bf16 h0 = (__bf16)x0[e], h1 = (__bf16)x1[e];
    hi[e] = h0; hi[4 + e] = h1;
    lo[e] = (__bf16)(x0[e] - (float)h0); lo[4 + e] = (__bf16)(x1[e] - (float)h1);
  }
}
__device__ __forceinline__ void fwd_split4(const f32x4& x, bf16x4& hi, bf16x4& lo) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const __bf16 h = (__bf16)x[e];
    hi[e] = h;
    lo[e] = (__bf16)(x[e] - (float)h);
  }
}

template <bool PERHEAD, bool RPE, bool BOX = false, bool BF16 = false, int WAVES = kFwdWaves>
__device__ __forceinline__ void attn_fwd_body(AttnParams P) {
  static_assert(WAVES == 8 || WAVES == 4, "the merge of the wave states is written for 8 or 4 waves");
  static_assert(!(BF16 && PERHEAD), "the bf16 path is built for the shared-KV kinds");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  attn_load_rng(P);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int g = lane >> 4, c = lane & 15;
  const int b = blockIdx.z;
  const int head = PERHEAD ? (int)blockIdx.y % P.H : 0;
  const int split = PERHEAD ? (int)blockIdx.y / P.H : (int)blockIdx.y;  // key split: grid.y = split (shared) or head + H*split
  const int H = P.H, nQ = P.nQ, nK = P.nK;
  const int rows_per_tile_q = PERHEAD ? 16 : 4;  // queries per workgroup
  const int q0 = blockIdx.x * rows_per_tile_q;
  const int qstride = H * kDh;                 // floats per query row of q / out
  const int kvoff = PERHEAD ? head * kDh : 0;

  const int table_floats = RPE ? kRpeVerts * P.T * P.T * P.T * 4 : 0;
  f32x4* tab = reinterpret_cast<f32x4*>(smem);
  float* ppad = smem + table_floats + w * (16 * kPPad);
  // ---- per-lane pair geometry (RPE): query g of the tile; decides which instantiation owns this workgroup ---------
  float vx[8], vy[8], vz[8], rc = 1.f, rs = 0.f, eX = 0.f, eY = 0.f, eZ = 0.f;
  const bool rot = RPE && P.cos_sin != nullptr;
  const int q_pair = min(q0 + g, nQ - 1);
  if (RPE) {
    const float* vp = P.vertices + ((size_t)b * nQ + q_pair) * 24;
#pragma unroll
    for (int i = 0; i < 8; ++i) { vx[i] = vp[i * 3]; vy[i] = vp[i * 3 + 1]; vz[i] = vp[i * 3 + 2]; }
    if (rot) { rc = P.cos_sin[((size_t)b * nQ + q_pair) * 2]; rs = P.cos_sin[((size_t)b * nQ + q_pair) * 2 + 1]; }
    // same for the 8 waves: same 4 queries.  With the rotation operand: the corners of rotated boxes (attn_common.h)
    const bool box = P.box_path && (rot ? __all(rpe_box_pattern_rot(vx, vy, vz, rc, rs, eX, eY, eZ)) : __all(rpe_box_pattern(vx, vy, vz)));
    if (box != BOX) return;
  }
  const float bX[2] = {vx[0], vx[2]}, bY[2] = {vy[0], vy[1]}, bZ[2] = {vz[0], vz[4]};
  if (RPE) rpe_stage_table(P, tab, tid, WAVES * kWave);

  // ---- A operand of QK^T: row i = c ---------------------------------------------------------------
  float qa[BF16 ? 1 : 16];
  bf16x8 qa8[2];
  {
    const int qi = PERHEAD ? min(q0 + c, nQ - 1) : min(q0 + (c >> 2), nQ - 1);
    const int hoff = PERHEAD ? head * kDh : (c & 3) * kDh;
    if (BF16) {  // 16 bf16 of the row: d = 16 g + 8 m + e for instruction m; the scale goes onto the fp32 scores
      const bf16x8* src = reinterpret_cast<const bf16x8*>(reinterpret_cast<const __bf16*>(P.q) + ((size_t)b * nQ + qi) * qstride + hoff + 16 * g);
      qa8[0] = src[0]; qa8[1] = src[1];
    } else {
      const f32x4* src = reinterpret_cast<const f32x4*>(P.q + ((size_t)b * nQ + qi) * qstride + hoff + 16 * g);
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const f32x4 v = src[s4];
#pragma unroll
        for (int e = 0; e < 4; ++e) qa[s4 * 4 + e] = v[e] * P.scale;
      }
    }
  }
  bf16x8 qh[2], ql[2];  // split form of the scaled q operand: instruction m covers d = 16 g + 8 m + e
  if constexpr (!BF16 && kFwdSplit) {
#pragma unroll
    for (int mm = 0; mm < 2; ++mm) {
      const f32x4 x0 = {qa[8 * mm], qa[8 * mm + 1], qa[8 * mm + 2], qa[8 * mm + 3]};
      const f32x4 x1 = {qa[8 * mm + 4], qa[8 * mm + 5], qa[8 * mm + 6], qa[8 * mm + 7]};
      fwd_split8(x0, x1, qh[mm], ql[mm]);
    }
  }
  // query index of accumulator register r
  int qrow[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) qrow[r] = PERHEAD ? (q0 + 4 * g + r) : (q0 + g);

  f32x4 o[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) o[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m[4] = {kNegBig, kNegBig, kNegBig, kNegBig}, l[4] = {0.f, 0.f, 0.f, 0.f};

  if (RPE) __syncthreads();  // table staged

  const int ntiles = (nK + 15) >> 4;
  const int tile_begin = split * P.tiles_per_split;
  const int tile_end = min(ntiles, tile_begin + P.tiles_per_split);

  // Operands of a key tile: fetched one tile AHEAD (software pipelining).  With loads issued right before their
  // use every tile step exposed 2-3 full memory latencies (vmcnt is in-order on gfx950, so a V load also waits
  // for the score stores issued before it).
  struct TileOps {
    f32x4 kb[BF16 ? 1 : 4], vb[BF16 ? 1 : 4];
    bf16x8 kb8[2];
    bf16x4 vb4[4];
    float kx, ky, kz;
  };
  auto fetch = [&](int tile, TileOps& t) {
    const int key0 = tile << 4;
    const int keyc = min(key0 + c, nK - 1);
    if (BF16) {
      const bf16x8* kp = reinterpret_cast<const bf16x8*>(reinterpret_cast<const __bf16*>(P.k) + ((size_t)b * nK + keyc) * P.k_stride + kvoff + 16 * g);
      t.kb8[0] = kp[0]; t.kb8[1] = kp[1];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int kk = min(key0 + 4 * g + s, nK - 1);
        t.vb4[s] = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const __bf16*>(P.v) + ((size_t)b * nK + kk) * P.v_stride + kvoff + 4 * c);
      }
    } else {
      const f32x4* kp = reinterpret_cast<const f32x4*>(P.k + ((size_t)b * nK + keyc) * P.k_stride + kvoff + 16 * g);
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) t.kb[s4] = kp[s4];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int kk = min(key0 + 4 * g + s, nK - 1);
        t.vb[s] = *reinterpret_cast<const f32x4*>(P.v + ((size_t)b * nK + kk) * P.v_stride + kvoff + 4 * c);
      }
    }
    if (RPE) {
      const float* xp = P.xyz + ((size_t)b * nK + keyc) * 3;
      t.kx = xp[0]; t.ky = xp[1]; t.kz = xp[2];
    }
  };
  TileOps ops, nxt;
  if (tile_begin + w < tile_end) fetch(tile_begin + w, ops);

  for (int tile = tile_begin + w; tile < tile_end; tile += WAVES) {
    const int key0 = tile << 4;
    const int key = key0 + c;
    const bool kvalid = key < nK;
    const int keyc = min(key, nK - 1);
    if (tile + WAVES < tile_end) fetch(tile + WAVES, nxt);
    // ---- S = Q K^T --------------------------------------------------------------------------------
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if constexpr (BF16) {
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa8[0], ops.kb8[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa8[1], ops.kb8[1], acc, 0, 0, 0);
      acc *= P.scale;
    } else if constexpr (kFwdSplit) {
#pragma unroll
      for (int mm = 0; mm < 2; ++mm) {
        bf16x8 kh, kl;
        fwd_split8(ops.kb[2 * mm], ops.kb[2 * mm + 1], kh, kl);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ql[mm], kh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qh[mm], kl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qh[mm], kh, acc, 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int s = 0; s < 16; ++s)
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[s], ops.kb[s >> 2][s & 3], acc, 0, 0, 0);
    }
    float sc[4] = {acc[0], acc[1], acc[2], acc[3]};
    // ---- + RPE bias -------------------------------------------------------------------------------
    if (RPE) {
      if (BOX) {
        if (rot) {  // one rotation per pair: R (P_0 - X), then the box's edges along the turned axes
          float d0x = vx[0] - ops.kx, d0y = vy[0] - ops.ky;
          const float d0z = vz[0] - ops.kz;
          rpe_rotate(d0x, d0y, rc, rs);
          const float dx[2] = {d0x, d0x + eX}, dy[2] = {d0y, d0y + eY}, dz[2] = {d0z, d0z + eZ};
          rpe_pair_bias_box_d(P, tab, dx, dy, dz, sc);
        } else {
          rpe_pair_bias_box(P, tab, bX, bY, bZ, ops.kx, ops.ky, ops.kz, sc);
        }
      } else {
        rpe_pair_bias(P, tab, vx, vy, vz, ops.kx, ops.ky, ops.kz, rot, rc, rs, sc);
      }
    }
    // ---- mask, tail ------------------------------------------------------------------------------
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (P.mask_kind != VDETR_MASK_NONE) {
        const size_t mi = ((size_t)b * nQ + min(qrow[r], nQ - 1)) * nK + keyc;
        if (P.mask_kind == VDETR_MASK_BOOL) {
          if (reinterpret_cast<const unsigned char*>(P.mask)[mi]) sc[r] = -100.f;
        } else {
          sc[r] += reinterpret_cast<const float*>(P.mask)[mi];
        }
      }
      if (!kvalid) sc[r] = kNegBig;
    }
    if (P.scores && kvalid) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (qrow[r] < nQ) {
          const size_t row = PERHEAD ? (((size_t)b * H + head) * nQ + qrow[r]) : (((size_t)b * nQ + qrow[r]) * H + r);
          if (VDETR_STREAM_NT) __builtin_nontemporal_store(sc[r], &P.scores[row * nK + key]);  // 67 MB per layer: past L2
          else P.scores[row * nK + key] = sc[r];
        }
      }
    }
    // ---- online softmax (rows live across the 16 lanes of a DPP row) --------------------------------
    float p[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float tmax = row_allmax_f32_fx(sc[r]);
      const float mn = fmaxf(m[r], tmax);
      const float alpha = __expf(m[r] - mn);
      const float e = kvalid ? __expf(sc[r] - mn) : 0.f;
      l[r] = l[r] * alpha + row_allsum_f32_fx(e);
      m[r] = mn;
      p[r] = e;
#pragma unroll
      for (int t = 0; t < 4; ++t) o[t][r] *= alpha;
    }
    // ---- dropout on the probabilities (normaliser l uses the undropped p) ---------------------------
    if (P.drop_thresh) {
      if (PERHEAD) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const uint4 rnd = attn_rand4(P, b, qrow[r], key, head >> 2);
          p[r] = pick4(rnd, head & 3) >= P.drop_thresh ? p[r] * P.drop_scale : 0.f;
        }
      } else {
        const uint4 rnd = attn_rand4(P, b, qrow[0], key, 0);
        p[0] = rnd.x >= P.drop_thresh ? p[0] * P.drop_scale : 0.f;
        p[1] = rnd.y >= P.drop_thresh ? p[1] * P.drop_scale : 0.f;
        p[2] = rnd.z >= P.drop_thresh ? p[2] * P.drop_scale : 0.f;
        p[3] = rnd.w >= P.drop_thresh ? p[3] * P.drop_scale : 0.f;
      }
    }
    // ---- P: accumulator layout -> A-operand layout through the wave-private LDS pad -------------------
#pragma unroll
    for (int r = 0; r < 4; ++r) ppad[(4 * g + r) * kPPad + c] = p[r];
    __builtin_amdgcn_wave_barrier();
    const f32x4 pa = *reinterpret_cast<const f32x4*>(ppad + c * kPPad + 4 * g);
    __builtin_amdgcn_wave_barrier();
    // ---- O += P V ------------------------------------------------------------------------------------
    if constexpr (BF16) {
      const bf16x4 pb = {(__bf16)pa[0], (__bf16)pa[1], (__bf16)pa[2], (__bf16)pa[3]};  // P[row c][keys 4g..4g+3]
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const bf16x4 vt = {ops.vb4[0][t], ops.vb4[1][t], ops.vb4[2][t], ops.vb4[3][t]};  // V[keys 4g..4g+3][d = 4c + t]
        o[t] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(short4v, pb), __builtin_bit_cast(short4v, vt), o[t], 0, 0, 0);
      }
    } else if constexpr (kFwdSplit) {
      bf16x4 ph, pl;
      fwd_split4(pa, ph, pl);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const f32x4 vt = {ops.vb[0][t], ops.vb[1][t], ops.vb[2][t], ops.vb[3][t]};  // V[keys 4g..4g+3][d = 4c + t]
        bf16x4 vh, vl;
        fwd_split4(vt, vh, vl);
        o[t] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(short4v, pl), __builtin_bit_cast(short4v, vh), o[t], 0, 0, 0);
        o[t] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(short4v, ph), __builtin_bit_cast(short4v, vl), o[t], 0, 0, 0);
        o[t] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(short4v, ph), __builtin_bit_cast(short4v, vh), o[t], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int s = 0; s < 4; ++s) o[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[s], ops.vb[s][t], o[t], 0, 0, 0);
    }
    ops = nxt;
  }

  // ---- merge the 8 wave states -------------------------------------------------------------------------
  __syncthreads();  // every wave is done with the table image: reuse it
  float* red = smem;  // [w][lane][24]: 16 o + 4 m + 4 l
  {
    float* mine = red + ((size_t)w * kWave + lane) * 24;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) mine[t * 4 + r] = o[t][r];
#pragma unroll
    for (int r = 0; r < 4; ++r) { mine[16 + r] = m[r]; mine[20 + r] = l[r]; }
  }
  __syncthreads();
  // 8 waves: wave w finishes d-tile t = w>>1 for registers r in {2*(w&1), 2*(w&1)+1}; 4 waves: d-tile t = w, all four registers
  {
    const int t = WAVES == 8 ? (w >> 1) : w;
#pragma unroll
    for (int rr = 0; rr < 16 / WAVES; ++rr) {
      const int r = WAVES == 8 ? 2 * (w & 1) + rr : rr;
      float M = kNegBig;
#pragma unroll
      for (int ww = 0; ww < WAVES; ++ww) M = fmaxf(M, red[((size_t)ww * kWave + lane) * 24 + 16 + r]);
      float L = 0.f, val = 0.f;
#pragma unroll
      for (int ww = 0; ww < WAVES; ++ww) {
        const float* src = red + ((size_t)ww * kWave + lane) * 24;
        const float f = __expf(src[16 + r] - M);
        L += src[20 + r] * f;
        val += src[t * 4 + r] * f;
      }
      const int qi = qrow[r];
      if (qi < nQ) {
        const float inv = L > 0.f ? 1.f / L : 0.f;
        const float lse = L > 0.f ? M + __logf(L) : kNegBig;
        const int hh = PERHEAD ? head : r;
        const int d = 4 * c + t;
        if (P.ksplit == 1) {
          P.out[((size_t)b * nQ + qi) * qstride + hh * kDh + d] = val * inv;
          if (t == 0 && c == 0) {
            const size_t row = PERHEAD ? (((size_t)b * H + hh) * nQ + qi) : (((size_t)b * nQ + qi) * H + hh);
            P.lse[row] = lse;
          }
        } else {
          const size_t row = ((size_t)b * nQ + qi) * H + hh;
          const size_t rows = (size_t)P.B * nQ * H;
          P.part_o[((size_t)split * rows + row) * kDh + d] = val * inv;
          if (t == 0 && c == 0) P.part_lse[(size_t)split * rows + row] = lse;
        }
      }
    }
  }
}

template <bool PERHEAD, bool RPE, bool BOX = false, bool BF16 = false>
__global__ __launch_bounds__(kFwdThreads) void attn_fwd_kernel(AttnParams P) {
  attn_fwd_body<PERHEAD, RPE, BOX, BF16>(P);
}
// The per-head kind (the query self-attention) in workgroups of FOUR waves, one per SIMD: 152 registers each, so two or three of
// them share a CU.  At the model's size the launch is H * nQ / 16 = 256 workgroups = the chip's CU count, and in the training
// step one CU is always held by the next scene's sampling kernel: the eight-wave form (one workgroup per CU) then runs two rounds
// (27 -> 47 us per layer); these all start at once.  (Round 5 tried the other ways out: key halves + a combine launch, and the
// eight-wave form squeezed into 128 registers — no gain, 27 spills.)
__global__ __launch_bounds__(4 * kWave) void attn_fwd_perhead4_kernel(AttnParams P) {
  attn_fwd_body<true, false, false, false, 4>(P);
}

// The RPE attention with the instantiation chosen per workgroup IN the kernel (its 4 queries are axis-aligned boxes or not:
// the test attn_fwd_body repeats).  Launching the two instantiations side by side cost a 1024-workgroup launch of
// immediate exits per layer (6 us with the 133 KB LDS reservation); both bodies use the same register budget.
template <bool BF16>
__global__ __launch_bounds__(kFwdThreads) void attn_fwd_rpe_auto_kernel(AttnParams P) {
  const int g = (threadIdx.x & 63) >> 4;
  const int q_pair = min((int)blockIdx.x * 4 + g, P.nQ - 1);
  const float* vp = P.vertices + ((size_t)blockIdx.z * P.nQ + q_pair) * 24;
  float vx[8], vy[8], vz[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { vx[i] = vp[i * 3]; vy[i] = vp[i * 3 + 1]; vz[i] = vp[i * 3 + 2]; }
  bool box;
  if (P.cos_sin == nullptr) {
    box = P.box_path && __all(rpe_box_pattern(vx, vy, vz));
  } else {  // rotated boxes (the body repeats the test and keeps the edges)
    const float rc = P.cos_sin[((size_t)blockIdx.z * P.nQ + q_pair) * 2], rs = P.cos_sin[((size_t)blockIdx.z * P.nQ + q_pair) * 2 + 1];
    float ex, ey, ez;
    box = P.box_path && __all(rpe_box_pattern_rot(vx, vy, vz, rc, rs, ex, ey, ez));
  }
  if (box) attn_fwd_body<false, true, true, BF16>(P);
  else attn_fwd_body<false, true, false, BF16>(P);
}


// merge of key-split partials: out = sum_s exp(lse_s - LSE) * o_s
__global__ __launch_bounds__(256) void attn_fwd_combine_kernel(AttnParams P) {
  const size_t rows = (size_t)P.B * P.nQ * P.H;
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;  // element of [rows][64]
  if (e >= rows * kDh) return;
  const size_t row = e / kDh;
  const int d = (int)(e % kDh);
  float M = kNegBig;
  for (int s = 0; s < P.ksplit; ++s) M = fmaxf(M, P.part_lse[(size_t)s * rows + row]);
  float L = 0.f, val = 0.f;
  for (int s = 0; s < P.ksplit; ++s) {
    const float f = __expf(P.part_lse[(size_t)s * rows + row] - M);
    L += f;
    val += f * P.part_o[((size_t)s * rows + row) * kDh + d];
  }
  // row = (b*nQ + q)*H + h  ->  out[(b*nQ+q)*H*64 + h*64 + d] is the same flat offset
  P.out[row * kDh + d] = L > 0.f ? val / L : 0.f;
  if (d == 0) {
    size_t li = row;  // shared kinds: lse rows are (b, q, h); per-head kind: (b, h, q)
    if (P.kind == VDETR_ATTN_PER_HEAD) {
      const size_t bq = row / P.H, h = row - bq * P.H, b = bq / P.nQ, q = bq - b * P.nQ;
      li = (b * P.H + h) * P.nQ + q;
    }
    P.lse[li] = L > 0.f ? M + __logf(L) : kNegBig;
  }
}

// ---- stand-alone RPE bias (parity hook for vdetr_transformer.py:710-731) ------------------------------
__global__ __launch_bounds__(256) void rpe_bias_kernel(AttnParams P, float* rpe) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  f32x4* tab = reinterpret_cast<f32x4*>(smem);
  rpe_stage_table(P, tab, threadIdx.x, 256);
  __syncthreads();
  const int b = blockIdx.z, q = blockIdx.y;
  const float* vp = P.vertices + ((size_t)b * P.nQ + q) * 24;
  float vx[8], vy[8], vz[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { vx[i] = vp[i * 3]; vy[i] = vp[i * 3 + 1]; vz[i] = vp[i * 3 + 2]; }
  const bool rot = P.cos_sin != nullptr;
  const float rc = rot ? P.cos_sin[((size_t)b * P.nQ + q) * 2] : 1.f;
  const float rs = rot ? P.cos_sin[((size_t)b * P.nQ + q) * 2 + 1] : 0.f;
  for (int key = blockIdx.x * 256 + threadIdx.x; key < P.nK; key += gridDim.x * 256) {
    const float* xp = P.xyz + ((size_t)b * P.nK + key) * 3;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    rpe_pair_bias(P, tab, vx, vy, vz, xp[0], xp[1], xp[2], rot, rc, rs, acc);
#pragma unroll
    for (int h = 0; h < 4; ++h) rpe[(((size_t)b * P.nQ + q) * 4 + h) * P.nK + key] = acc[h];
  }
}

// ---- MFMA layout self test: C[16][16] = A[16][64] * B[16][64]^T with the operand mapping used above ----
__global__ void mfma_selftest_kernel(const float* a, const float* b, float* cmat) {
  const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < 16; ++s)
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c * 64 + 16 * g + s], b[c * 64 + 16 * g + s], acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) cmat[(4 * g + r) * 16 + c] = acc[r];
}

}  // namespace vdetr

using namespace vdetr;

namespace vdetr {
bool attn_fwd_self_eligible(const vdetr_attn_desc* d, int ksplit);  // attn_fwd_self.hip
int attn_fwd_self_launch(const AttnParams& P, hipStream_t st);
int attn_fill_params(const vdetr_attn_desc* d, AttnParams* P, const char* op) {
  VDETR_REQUIRE(d != nullptr, "%s: null descriptor", op);
  VDETR_REQUIRE(d->kind == VDETR_ATTN_SHARED_KV || d->kind == VDETR_ATTN_PER_HEAD, "%s: bad kind %d", op, d->kind);
  VDETR_REQUIRE(d->B > 0 && d->H > 0 && d->nQ > 0 && d->nK > 0, "%s: empty dimension B=%d H=%d nQ=%d nK=%d", op,
                d->B, d->H, d->nQ, d->nK);
  VDETR_REQUIRE(d->B <= 65535, "%s: B=%d > 65535", op, d->B);
  VDETR_REQUIRE(d->kind == VDETR_ATTN_PER_HEAD || d->H == kRpeHeads,
                "%s: shared-KV attention is built for %d heads (got %d)", op, kRpeHeads, d->H);
  VDETR_REQUIRE(d->dropout_p >= 0.f && d->dropout_p < 1.f, "%s: dropout_p %f outside [0,1)", op, d->dropout_p);
  VDETR_REQUIRE(d->mask_kind == VDETR_MASK_NONE || d->mask != nullptr, "%s: mask_kind set but mask is null", op);
  *P = AttnParams{};
  P->kind = d->kind; P->B = d->B; P->H = d->H; P->nQ = d->nQ; P->nK = d->nK; P->scale = d->scale;
  const int dense = d->kind == VDETR_ATTN_PER_HEAD ? d->H * 64 : 64;
  P->k_stride = d->k_row_stride ? d->k_row_stride : dense;
  P->v_stride = d->v_row_stride ? d->v_row_stride : dense;
  VDETR_REQUIRE(P->k_stride >= dense && P->v_stride >= dense && P->k_stride % 4 == 0 && P->v_stride % 4 == 0,
                "%s: K/V row strides %d / %d must be multiples of 4 and >= %d", op, P->k_stride, P->v_stride, dense);
  P->bwd_aux = d->bwd_aux;
  P->table = d->table;
  if (d->table) {
    VDETR_REQUIRE(d->kind == VDETR_ATTN_SHARED_KV, "%s: RPE needs the shared-KV kind", op);
    VDETR_REQUIRE(d->vertices && d->xyz, "%s: RPE table given without vertices/xyz", op);
    VDETR_REQUIRE(d->table_size >= 2, "%s: table_size %d < 2", op, d->table_size);
    const size_t bytes = (size_t)kRpeVerts * d->table_size * d->table_size * d->table_size * 16 + kFwdWaves * 16 * kPPad * 4;
    VDETR_REQUIRE(bytes <= 160 * 1024, "%s: RPE table of edge %d does not fit the 160 KB LDS", op, d->table_size);
    P->T = d->table_size;
    P->log_scale = d->log_scale;
    P->pix_mul = d->inv_log_norm * 0.5f * (float)d->table_size;
    P->pix_add = 0.5f * (float)(d->table_size - 1);
    P->vertices = d->vertices; P->xyz = d->xyz; P->cos_sin = d->cos_sin;
  }
  P->mask = d->mask; P->mask_kind = d->mask ? d->mask_kind : VDETR_MASK_NONE;
  if (d->dropout_p > 0.f) {
    int t = (int)((double)d->dropout_p * 65536.0 + 0.5);
    t = t < 1 ? 1 : (t > 65535 ? 65535 : t);
    P->drop_thresh = (unsigned)t;
    P->drop_scale = 65536.f / (float)(65536 - t);
  } else {
    P->drop_thresh = 0; P->drop_scale = 1.f;
  }
  P->seed_lo = (unsigned)d->seed; P->seed_hi = (unsigned)(d->seed >> 32);
  P->off_lo = (unsigned)d->offset; P->off_hi = (unsigned)(d->offset >> 32);
  P->rng = reinterpret_cast<const unsigned long long*>(d->rng_state);
  P->ksplit = 1; P->tiles_per_split = (d->nK + 15) / 16;
  return VDETR_OK;
}

int attn_fwd_pipe_launch(const AttnParams& P, unsigned* counter, int workgroups, char* kv_img, int split, bool packed, bool src_f32, hipStream_t st);  // attn_fwd_pipe.hip
int attn_fwd_pack_launch(const void* k, const void* v, int B, int nK, int k_stride, int v_stride, int nlayers, long layer_stride, char* img,
                         int split, bool src_f32, hipStream_t st);
size_t attn_fwd_pipe_img_bytes(int B, int nK, int split);

// the persistent forward (attn_fwd_pipe.hip) takes the 3DV-RPE attention as the model runs it: fp32, table edge 10, no mask
static bool pipe_eligible(const vdetr_attn_desc* d) {
  return d->kind == VDETR_ATTN_SHARED_KV && d->table && d->table_size == 10 && !d->mask && d->fwd_kernel != 1;
}
// fwd_kernel 0: the persistent forward with its products on the bf16 matrix unit (three-way / two-way split operands, fp32
// accuracy: attn_fwd_pipe.hip); 2: the same with fp32 matrix instructions; 1: the grid kernel; 3: the persistent forward with q / k / v
// rounded to ONE bf16 part each (bf16 products as BASELINE config 4 names them, on f32 tensors).  Returns the parts of K (0: none).
static int pipe_split(const vdetr_attn_desc* d) { return !pipe_eligible(d) ? 0 : d->fwd_kernel == 0 ? 3 : d->fwd_kernel == 3 ? 1 : 0; }

// key split so that small query counts still fill the chip (shared kinds only)
static int choose_ksplit(const vdetr_attn_desc* d) {
  if (d->kind != VDETR_ATTN_SHARED_KV) {
    // per-head kind: the query self-attention launches H * nQ/16 = 256 workgroups at the model's size, i.e. exactly one
    // per CU — two rounds whenever a CU is busy elsewhere (see below).  Two key halves per (head, query tile) instead.
    // (measured: no gain at nQ = nK = 1024 — the 25 us workgroups are prologue / merge dominated and the combine launch
    // costs what the second round did; kept behind VDETR_FWD_KSPLIT_PERHEAD for larger self-attentions.  Also tried: the
    // per-head instantiation compiled for 128 VGPRs (two workgroups per CU, so that the 256 workgroups fit next to a busy
    // CU): 27 spilled registers, 48.7 instead of 41.4 us inside the step)
    const int ph = VDETR_AB("VDETR_FWD_KSPLIT_PERHEAD", 1);
    const int ntiles = (d->nK + 15) / 16;
    return (ph > 1 && ntiles >= 2 * ph * kFwdWaves) ? ph : 1;
  }
  const long wgs = (long)d->B * ((d->nQ + 3) / 4);
  const int ntiles = (d->nK + 15) / 16;
  int ks = 1;
  while (wgs * ks < 256 && ks * 2 * kFwdWaves <= ntiles && ks < 16) ks *= 2;
  // A grid that exactly fills the chip (256 workgroups, one per CU) takes TWO rounds as soon as one CU is busy with
  // something else — and in the training step one always is: the next scene's furthest-point sampling runs on a side
  // stream (measured: 307 us instead of 181 us per launch).  Finer workgroups let the hardware dispatcher
  // balance the load over whatever CUs are free (4.02 -> 5 rounds of 1/4 size instead of 2 of full size); the price is the
  // per-workgroup prologue + merge (~6 us) and the combine kernel.
  const int forced = VDETR_AB("VDETR_FWD_KSPLIT", 0);
  const int fine = forced > 0 ? forced : 4;  // step time at 1/2/4/8: 18.13 / 17.91 / 17.74 / 18.06 ms
  if (d->table && ks < fine && wgs >= 64) {
    while (ks < fine && ks * 2 * kFwdWaves <= ntiles) ks *= 2;
  }
  return ks;
}
}  // namespace vdetr

extern "C" size_t vdetr_attn_fwd_workspace_bytes(const vdetr_attn_desc* d) {
  if (!d) return 0;
  const int ks = choose_ksplit(d);
  const size_t sched = pipe_eligible(d) && !d->fwd_sched ? 256 : 0;  // the item counter, where the caller brings none
  const size_t img = pipe_split(d) && !d->kv_img ? attn_fwd_pipe_img_bytes(d->B, d->nK, 3) + 256 : 0;  // (the bf16 forward's image is smaller: same bound)
  if (ks == 1) return sched + img;
  const size_t rows = (size_t)d->B * d->nQ * d->H;
  return (size_t)ks * rows * (kDh + 1) * sizeof(float) + 256 + sched + img;
}

extern "C" size_t vdetr_attn_kv_image_bytes(int B, int nK) { return B > 0 && nK > 0 ? attn_fwd_pipe_img_bytes(B, nK, 3) : 0; }
extern "C" size_t vdetr_attn_kv_image_parts_bytes(int B, int nK, int parts) {
  return B > 0 && nK > 0 && (parts == 1 || parts == 3) ? attn_fwd_pipe_img_bytes(B, nK, parts) : 0;
}

extern "C" int vdetr_attn_pack_kv_f32(const float* k, const float* v, int B, int nK, int k_row_stride, int v_row_stride, int nlayers,
                                      int64_t layer_stride, void* img, vdetr_stream_t stream) {
  VDETR_REQUIRE(k && v && img, "attn_pack_kv: null pointer");
  VDETR_REQUIRE(B > 0 && nK > 0 && nlayers > 0 && nlayers <= 65535 && B <= 65535, "attn_pack_kv: B=%d nK=%d nlayers=%d", B, nK, nlayers);
  VDETR_REQUIRE(k_row_stride >= kDh && v_row_stride >= kDh && k_row_stride % 4 == 0 && v_row_stride % 4 == 0 && layer_stride % 4 == 0 &&
                (((uintptr_t)k | (uintptr_t)v | (uintptr_t)img) & 15) == 0, "attn_pack_kv: rows of >= 64 floats, strides multiples of 4, 16-B aligned");
  return attn_fwd_pack_launch(k, v, B, nK, k_row_stride, v_row_stride, nlayers, (long)layer_stride, (char*)img, 3, true, (hipStream_t)stream);
}

extern "C" int vdetr_attn_pack_kv_parts_f32(const float* k, const float* v, int B, int nK, int k_row_stride, int v_row_stride, int nlayers,
                                            int64_t layer_stride, int parts, void* img, vdetr_stream_t stream) {
  VDETR_REQUIRE(k && v && img, "attn_pack_kv: null pointer");
  VDETR_REQUIRE(parts == 1 || parts == 3, "attn_pack_kv: parts=%d (3 = f32 accuracy, 1 = operands rounded to bf16)", parts);
  VDETR_REQUIRE(B > 0 && nK > 0 && nlayers > 0 && nlayers <= 65535 && B <= 65535, "attn_pack_kv: B=%d nK=%d nlayers=%d", B, nK, nlayers);
  VDETR_REQUIRE(k_row_stride >= kDh && v_row_stride >= kDh && k_row_stride % 4 == 0 && v_row_stride % 4 == 0 && layer_stride % 4 == 0 &&
                (((uintptr_t)k | (uintptr_t)v | (uintptr_t)img) & 15) == 0, "attn_pack_kv: rows of >= 64 floats, strides multiples of 4, 16-B aligned");
  return attn_fwd_pack_launch(k, v, B, nK, k_row_stride, v_row_stride, nlayers, (long)layer_stride, (char*)img, parts, true, (hipStream_t)stream);
}

// parts != nullptr: the key-split merge is left to the consumer (vdetr_attn_fwd_parts_f32)
static int attn_fwd_run(const vdetr_attn_desc* d, const float* q, const float* k, const float* v, float* out, float* lse, float* scores,
                        void* workspace, size_t workspace_bytes, vdetr_attn_parts* parts, vdetr_stream_t stream) {
  AttnParams P;
  if (int e = attn_fill_params(d, &P, "attn_fwd")) return e;
  VDETR_REQUIRE(q && k && v && out && lse, "attn_fwd: null pointer");
  P.q = q; P.k = k; P.v = v; P.out = out; P.lse = lse; P.scores = scores;
  const bool perhead = d->kind == VDETR_ATTN_PER_HEAD;
  const bool rpe = d->table != nullptr;
  const int ks = choose_ksplit(d);
  const bool pipe = pipe_eligible(d);
  const size_t need = vdetr_attn_fwd_workspace_bytes(d);
  if (need && (!workspace || workspace_bytes < need)) {
    set_error("attn_fwd: workspace %zu B < required %zu B", workspace_bytes, need);
    return VDETR_ERR_WORKSPACE;
  }
  unsigned* sched = d->fwd_sched;
  size_t sched_bytes = 0;
  if (pipe && !sched) {  // head of the workspace, cleared in front of the launch (a memset node in a captured graph)
    sched = (unsigned*)(((uintptr_t)workspace + 15) & ~(uintptr_t)15);
    sched_bytes = 256;
    if (hipMemsetAsync(sched, 0, 16, (hipStream_t)stream) != hipSuccess) {
      set_error("attn_fwd: cannot clear the item counter");
      return VDETR_ERR_LAUNCH;
    }
  }
  uintptr_t ws_top = (uintptr_t)workspace + sched_bytes;
  if (ks > 1) {
    const size_t rows = (size_t)d->B * d->nQ * d->H;
    uintptr_t base = (ws_top + 255) & ~(uintptr_t)255;
    P.part_o = (float*)base;
    P.part_lse = P.part_o + (size_t)ks * rows * kDh;
    P.ksplit = ks;
    const int ntiles = (d->nK + 15) / 16;
    P.tiles_per_split = (ntiles + ks - 1) / ks;
    ws_top = base + (size_t)ks * rows * (kDh + 1) * sizeof(float);
  }
  char* kv_img = pipe_split(d) ? (d->kv_img ? (char*)d->kv_img : (char*)((ws_top + 255) & ~(uintptr_t)255)) : nullptr;
  VDETR_REQUIRE(!d->kv_img || (((uintptr_t)d->kv_img) & 15) == 0, "attn_fwd: kv_img must be 16-B aligned");
  const size_t lds_table = rpe ? (size_t)kRpeVerts * P.T * P.T * P.T * 16 : 0;
  const size_t lds = lds_table + (size_t)kFwdWaves * 16 * kPPad * 4 > (size_t)kFwdWaves * kWave * 24 * 4
                         ? lds_table + (size_t)kFwdWaves * 16 * kPPad * 4
                         : (size_t)kFwdWaves * kWave * 24 * 4;
  hipStream_t st = (hipStream_t)stream;
  if (pipe) {
    VDETR_REQUIRE((size_t)d->nK * P.k_stride < (1u << 30) && (size_t)d->nK * P.v_stride < (1u << 30) && (size_t)4 * d->nK < (1u << 30),
                  "attn_fwd: nK=%d too large for the persistent forward's 32-bit tile offsets", d->nK);
    if (int e = attn_fwd_pipe_launch(P, sched, device_cu_count(), kv_img, pipe_split(d), d->kv_img != nullptr, true, st)) return e;
  } else if (perhead && VDETR_AB("VDETR_FWD_SELF", 1) && attn_fwd_self_eligible(d, ks)) {
    if (int e = attn_fwd_self_launch(P, st)) return e;  // the lean kernel of the decoder's own case (attn_fwd_self.hip)
  } else if (perhead) {
    dim3 grid((d->nQ + 15) / 16, d->H * ks, d->B);
    const long wgs = (long)grid.x * grid.y * grid.z;
    const int four = VDETR_AB("VDETR_FWD_PERHEAD4", -1);  // -1: where the eight-wave workgroups would (almost) fill the chip or more
    if (d->fwd_kernel != 1 && (four > 0 || (four < 0 && 10 * wgs > 9 * (long)device_cu_count()))) {  // (fwd_kernel 1: the eight-wave form, A/B)
      const size_t lds4 = (size_t)4 * kWave * 24 * 4 > (size_t)4 * 16 * kPPad * 4 ? (size_t)4 * kWave * 24 * 4 : (size_t)4 * 16 * kPPad * 4;
      if (int e = set_lds(attn_fwd_perhead4_kernel, lds4, "attn_fwd")) return e;
      hipLaunchKernelGGL(attn_fwd_perhead4_kernel, grid, dim3(4 * kWave), lds4, st, P);
    } else {
      if (int e = set_lds(attn_fwd_kernel<true, false>, lds, "attn_fwd")) return e;
      hipLaunchKernelGGL((attn_fwd_kernel<true, false>), grid, dim3(kFwdThreads), lds, st, P);
    }
  } else {
    dim3 grid((d->nQ + 3) / 4, ks, d->B);
    if (rpe) {
      const int box_env = VDETR_AB("VDETR_FWD_BOX", 1);
      const int box_rot = VDETR_AB("VDETR_FWD_BOX_ROT", 1);
      P.box_path = box_env && (!d->cos_sin || box_rot);  // (VDETR_FWD_BOX_ROT=0: rotated boxes take the general body)
      const int auto_env = VDETR_AB("VDETR_FWD_AUTO", 1);
      if (P.box_path && auto_env) {  // one launch, the box / general body chosen per workgroup on the device
        if (int e = set_lds(attn_fwd_rpe_auto_kernel<false>, lds, "attn_fwd")) return e;
        hipLaunchKernelGGL((attn_fwd_rpe_auto_kernel<false>), grid, dim3(kFwdThreads), lds, st, P);
      } else {
        if (int e = set_lds(attn_fwd_kernel<false, true, false>, lds, "attn_fwd")) return e;
        hipLaunchKernelGGL((attn_fwd_kernel<false, true, false>), grid, dim3(kFwdThreads), lds, st, P);
        if (P.box_path) {
          if (int e = set_lds(attn_fwd_kernel<false, true, true>, lds, "attn_fwd")) return e;
          hipLaunchKernelGGL((attn_fwd_kernel<false, true, true>), grid, dim3(kFwdThreads), lds, st, P);
        }
      }
    } else {
      if (int e = set_lds(attn_fwd_kernel<false, false>, lds, "attn_fwd")) return e;
      hipLaunchKernelGGL((attn_fwd_kernel<false, false>), grid, dim3(kFwdThreads), lds, st, P);
    }
  }
  if (int e = check_launch("attn_fwd")) return e;
  if (parts) {
    parts->part_o = ks > 1 ? P.part_o : nullptr;
    parts->part_lse = ks > 1 ? P.part_lse : nullptr;
    parts->ksplit = ks;
    parts->reserved = 0;
    parts->rows = (int64_t)d->B * d->nQ * d->H;
    return VDETR_OK;
  }
  if (ks > 1) {
    const size_t elems = (size_t)d->B * d->nQ * d->H * kDh;
    hipLaunchKernelGGL(attn_fwd_combine_kernel, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, st, P);
    return check_launch("attn_fwd_combine");
  }
  return VDETR_OK;
}

extern "C" int vdetr_attn_fwd_f32(const vdetr_attn_desc* d, const float* q, const float* k, const float* v,
                                  float* out, float* lse, float* scores, void* workspace, size_t workspace_bytes,
                                  vdetr_stream_t stream) {
  return attn_fwd_run(d, q, k, v, out, lse, scores, workspace, workspace_bytes, nullptr, stream);
}

extern "C" int vdetr_attn_fwd_parts_f32(const vdetr_attn_desc* d, const float* q, const float* k, const float* v,
                                        float* out, float* lse, float* scores, void* workspace, size_t workspace_bytes,
                                        vdetr_attn_parts* parts, vdetr_stream_t stream) {
  VDETR_REQUIRE(d && parts, "attn_fwd_parts: null pointer");
  VDETR_REQUIRE(d->kind == VDETR_ATTN_SHARED_KV, "attn_fwd_parts: built for the shared-KV kinds (rows in (b, q, h) order)");
  return attn_fwd_run(d, q, k, v, out, lse, scores, workspace, workspace_bytes, parts, stream);
}

// q, k, v bf16 (k_row_stride / v_row_stride in ELEMENTS, multiples of 8: 16-B aligned operand loads); out, lse, scores fp32
extern "C" int vdetr_attn_fwd_bf16(const vdetr_attn_desc* d, const void* q, const void* k, const void* v, float* out, float* lse,
                                   float* scores, void* workspace, size_t workspace_bytes, vdetr_stream_t stream) {
  AttnParams P;
  if (int e = attn_fill_params(d, &P, "attn_fwd_bf16")) return e;
  VDETR_REQUIRE(d->kind == VDETR_ATTN_SHARED_KV, "attn_fwd_bf16: built for the shared-KV kinds");
  VDETR_REQUIRE(q && k && v && out && lse, "attn_fwd_bf16: null pointer");
  VDETR_REQUIRE(P.k_stride % 8 == 0 && P.v_stride % 8 == 0, "attn_fwd_bf16: K / V row strides %d / %d must be multiples of 8", P.k_stride, P.v_stride);
  VDETR_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15) == 0, "attn_fwd_bf16: operands must be 16-B aligned");
  P.q = (const float*)q; P.k = (const float*)k; P.v = (const float*)v; P.out = out; P.lse = lse; P.scores = scores;
  const bool rpe = d->table != nullptr;
  const int ks = choose_ksplit(d);
  const bool pipe = pipe_split(d) == 3;  // (fwd_kernel 0) the persistent forward, bf16 operands re-laid into its images (attn_fwd_pipe.hip, SPLIT = 1)
  const size_t need = vdetr_attn_fwd_workspace_bytes(d);
  if (need && (!workspace || workspace_bytes < need)) {
    set_error("attn_fwd_bf16: workspace %zu B < required %zu B", workspace_bytes, need);
    return VDETR_ERR_WORKSPACE;
  }
  unsigned* sched = d->fwd_sched;
  size_t sched_bytes = 0;
  if (pipe && !sched) {
    sched = (unsigned*)(((uintptr_t)workspace + 15) & ~(uintptr_t)15);
    sched_bytes = 256;
    if (hipMemsetAsync(sched, 0, 16, (hipStream_t)stream) != hipSuccess) {
      set_error("attn_fwd_bf16: cannot clear the item counter");
      return VDETR_ERR_LAUNCH;
    }
  }
  uintptr_t ws_top = (uintptr_t)workspace + sched_bytes;
  if (ks > 1) {
    const size_t rows = (size_t)d->B * d->nQ * d->H;
    uintptr_t base = (ws_top + 255) & ~(uintptr_t)255;
    P.part_o = (float*)base;
    P.part_lse = P.part_o + (size_t)ks * rows * kDh;
    P.ksplit = ks;
    P.tiles_per_split = ((d->nK + 15) / 16 + ks - 1) / ks;
    ws_top = base + (size_t)ks * rows * (kDh + 1) * sizeof(float);
  }
  if (pipe) {
    VDETR_REQUIRE((size_t)4 * d->nK < (1u << 30), "attn_fwd_bf16: nK=%d too large for the persistent forward's 32-bit tile offsets", d->nK);
    char* kv_img = (char*)((ws_top + 255) & ~(uintptr_t)255);
    if (int e = attn_fwd_pipe_launch(P, sched, device_cu_count(), kv_img, 1, false, false, (hipStream_t)stream)) return e;
    if (ks > 1) {
      const size_t elems = (size_t)d->B * d->nQ * d->H * kDh;
      hipLaunchKernelGGL(attn_fwd_combine_kernel, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, (hipStream_t)stream, P);
      return check_launch("attn_fwd_combine");
    }
    return VDETR_OK;
  }
  const size_t lds_table = rpe ? (size_t)kRpeVerts * P.T * P.T * P.T * 16 : 0;
  const size_t lds = lds_table + (size_t)kFwdWaves * 16 * kPPad * 4 > (size_t)kFwdWaves * kWave * 24 * 4
                         ? lds_table + (size_t)kFwdWaves * 16 * kPPad * 4
                         : (size_t)kFwdWaves * kWave * 24 * 4;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((d->nQ + 3) / 4, ks, d->B);
  if (rpe) {
    const int box_env = VDETR_AB("VDETR_FWD_BOX", 1);
    const int box_rot = VDETR_AB("VDETR_FWD_BOX_ROT", 1);
    P.box_path = box_env && (!d->cos_sin || box_rot);
    // (the merged kernel of the fp32 path spills 11 registers when built for bf16 operands: opt-in only)
    const int auto_env = VDETR_AB("VDETR_FWD_AUTO", 0);
    if (P.box_path && auto_env == 2) {
      if (int e = set_lds(attn_fwd_rpe_auto_kernel<true>, lds, "attn_fwd_bf16")) return e;
      hipLaunchKernelGGL((attn_fwd_rpe_auto_kernel<true>), grid, dim3(kFwdThreads), lds, st, P);
    } else {
      if (int e = set_lds(attn_fwd_kernel<false, true, false, true>, lds, "attn_fwd_bf16")) return e;
      hipLaunchKernelGGL((attn_fwd_kernel<false, true, false, true>), grid, dim3(kFwdThreads), lds, st, P);
      if (P.box_path) {
        if (int e = set_lds(attn_fwd_kernel<false, true, true, true>, lds, "attn_fwd_bf16")) return e;
        hipLaunchKernelGGL((attn_fwd_kernel<false, true, true, true>), grid, dim3(kFwdThreads), lds, st, P);
      }
    }
  } else {
    if (int e = set_lds(attn_fwd_kernel<false, false, false, true>, lds, "attn_fwd_bf16")) return e;
    hipLaunchKernelGGL((attn_fwd_kernel<false, false, false, true>), grid, dim3(kFwdThreads), lds, st, P);
  }
  if (int e = check_launch("attn_fwd_bf16")) return e;
  if (ks > 1) {
    const size_t elems = (size_t)d->B * d->nQ * d->H * kDh;
    hipLaunchKernelGGL(attn_fwd_combine_kernel, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, st, P);
    return check_launch("attn_fwd_combine");
  }
  return VDETR_OK;
}

extern "C" int vdetr_rpe_bias_f32(const vdetr_attn_desc* d, float* rpe, vdetr_stream_t stream) {
  AttnParams P;
  if (int e = attn_fill_params(d, &P, "rpe_bias")) return e;
  VDETR_REQUIRE(d->table && rpe, "rpe_bias: null pointer");
  VDETR_REQUIRE(d->nQ <= 65535, "rpe_bias: nQ=%d > 65535", d->nQ);
  const size_t lds = (size_t)kRpeVerts * P.T * P.T * P.T * 16;
  if (int e = set_lds(rpe_bias_kernel, lds, "rpe_bias")) return e;
  dim3 grid(min(ceil_div(d->nK, 256), 64), d->nQ, d->B);
  hipLaunchKernelGGL(rpe_bias_kernel, grid, dim3(256), lds, (hipStream_t)stream, P, rpe);
  return check_launch("rpe_bias");
}

extern "C" int vdetr_selftest_mfma_f32(const float* a, const float* b, float* c, vdetr_stream_t stream) {
  VDETR_REQUIRE(a && b && c, "selftest_mfma: null pointer");
  hipLaunchKernelGGL(mfma_selftest_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a, b, c);
  return check_launch("selftest_mfma");
}
