// attn_bwd_kv.hip — the key side of the shared-KV attention backward as ONE pass over the stored scores (gfx950).
//
// Autograd of vdetr_transformer.py:739-753 (attn = softmax(q k^T * scale + rpe); x = dropout(attn) v) needs, per layer,
//     dP~ = dO V^T,   P~ / dS = softmax + dropout backward,   dV = P~^T dO,   dK = scale dS^T q,   dQ = scale dS K
// over a [4 nQ x nK] score matrix (4 heads share K and V: rows are (query, head)).  As library GEMMs around an
// element-wise kernel that is 67 MB written for dP~, 67 + 67 MB read and 67 + 67 MB written element-wise, and 67 MB read
// by each of the three contractions (4096 x 4096 at C2): ~540 MB of HBM traffic, and fp32 matrix instructions that run at
// 1/16 of the bf16 rate.  Here a wave owns 32 keys and walks 32-row tiles of the scores:
//     S tile (the only big read) -> dP~ tile on the matrix unit (dO tile x the wave's V operand, resident in registers)
//     -> P~, dS in registers -> dS stored once (the table gradient and the dQ GEMM read it) -> dV^T += dO^T P~,
//     dK^T += q^T dS with the tile as the B operand exactly as the matrix unit left it (the accumulator layout of
//     v_mfma_f32_32x32x16_bf16 IS its B-operand layout up to a permutation of the 16 contraction slots, which the
//     pre-packed A operands follow).
// fp32 operands go through the bf16 unit as hi + lo (round-to-nearest) with the three leading cross terms: relative
// error of a product <= 2^-16, far below the 1e-3 parity tolerance and summed over thousands of terms of mixed sign.
// Determinism: the 16 row slots of a key tile (2 workgroups x 8 waves) are reduced by a fixed tree in LDS, and the two
// workgroups add their sums onto a zeroed output — two commutative float adds, the same bits in either order.
#include "kv_pack.h"

#include <stdlib.h>

#include <atomic>

namespace vdetr {

int attn_fill_params(const vdetr_attn_desc* d, AttnParams* P, const char* op);

typedef float f32x16 __attribute__((ext_vector_type(16)));

// waves per workgroup: 8 (2 per SIMD) when the kernel has the chip to itself; 4 (1 per SIMD, <= 256 registers) fit NEXT to
// the table-gradient kernel's 4 x 64 registers per SIMD when that runs on a side stream (attention.py)

struct KvParams {
  AttnParams A;
  const float* dout;  // [B][nQ][H * 64]
  float* dk;          // [B][nK][64] (shared K/V) or [B][nK][H * 64] (per head)
  float* dv;
  uint4* pack;        // [problems][NT][3 kinds][4][hi, lo][64 lanes] operand images of dO and q
  int R, NT;          // rows of a problem, 32-row tiles
};

// One problem = one score matrix [R x nK] with its own 64-wide K / V:
//   shared K/V (4 heads):  one per scene, rows r = (query, head) = 4 q + h, R = 4 nQ
//   per-head K/V:          one per (scene, head), rows = queries, R = nQ
struct KvProblem {
  const float* q;      // row r at q + r * rstride
  const float* dout;
  const float* v;      // key k at v + k * A.v_stride
  float* dk;           // key k at dk + k * ostride
  float* dv;
  size_t score_off;    // first element of the problem in scores / ds_out
  size_t row_off;      // first element in lse / delta
  int rstride, ostride, b, head;
};
template <bool PERHEAD>
__device__ __forceinline__ KvProblem kv_problem(const KvParams& K, int prob) {
  const AttnParams& P = K.A;
  KvProblem p;
  if (PERHEAD) {
    const int b = prob / P.H, hh = prob - b * P.H, C = P.H * kDh;
    p.b = b; p.head = hh;
    p.q = P.q + (size_t)b * P.nQ * C + hh * kDh;
    p.dout = K.dout + (size_t)b * P.nQ * C + hh * kDh;
    p.rstride = C;
    p.v = P.v + (size_t)b * P.nK * P.v_stride + hh * kDh;
    p.dk = K.dk + (size_t)b * P.nK * C + hh * kDh;
    p.dv = K.dv + (size_t)b * P.nK * C + hh * kDh;
    p.ostride = C;
  } else {
    p.b = prob; p.head = 0;
    p.q = P.q + (size_t)prob * K.R * kDh;
    p.dout = K.dout + (size_t)prob * K.R * kDh;
    p.rstride = kDh;
    p.v = P.v + (size_t)prob * P.nK * P.v_stride;
    p.dk = K.dk + (size_t)prob * P.nK * kDh;
    p.dv = K.dv + (size_t)prob * P.nK * kDh;
    p.ostride = kDh;
  }
  p.score_off = (size_t)prob * K.R * P.nK;
  p.row_off = (size_t)prob * K.R;
  return p;
}


__device__ __forceinline__ f32x16 kv_mfma(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// a b with a = ah + al, b = bh + bl: the three leading terms
__device__ __forceinline__ f32x16 kv_mfma3(bf16x8 ah, bf16x8 al, bf16x8 bh, bf16x8 bl, f32x16 c) {
  c = kv_mfma(al, bh, c);
  c = kv_mfma(ah, bl, c);
  return kv_mfma(ah, bh, c);
}

// ---- operand images ----------------------------------------------------------------------------------------------------
// kind 0 (sub = s):         A of dP~ = dO V^T:   lane (row = l & 31, g)  e -> dO[r0 + row][16 s + 8 g + e]
// kind 1 (sub = 2 mt + t):  A of dV^T = dO^T P~: lane (d = 32 mt + (l & 31), g)  e -> dO[r0 + kv_row(8 t + e, g)][d]
// kind 2:                   A of dK^T = q^T dS:  as kind 1 with q
// (the contraction slot (g, e) of instruction t is the tile row kv_row(8 t + e, g): what accumulator register 8 t + e of
// lane group g holds).  Also zero-fills dK / dV, which the main kernel accumulates into.
// The launch's first `ndelta` workgroups compute delta = rowsum(dO * O) and the words of bwd_aux (attn_delta_body) when the
// caller asks for it (vdetr_attn_bwd_kv_delta_f32): the two preparations are independent and 8 + 6 us apart.
template <bool PERHEAD>
__global__ __launch_bounds__(256) void attn_bwd_kv_pack_kernel(KvParams K, DeltaArgs D, int ndelta) {
  if ((int)blockIdx.x < ndelta) {
    __shared__ float wmax[4];
    attn_delta_body(D, (int)blockIdx.x, wmax);
    return;
  }
  const AttnParams& P = K.A;
  const int R = K.R, NT = K.NT;
  const int nprob = PERHEAD ? P.B * P.H : P.B;
  const long gid = (long)((int)blockIdx.x - ndelta) * 256 + threadIdx.x, stride = (long)((int)gridDim.x - ndelta) * 256;
  const long nunits = (long)nprob * NT * 12 * kWave;
  for (long u = gid; u < nunits; u += stride) {
    const int lane = (int)(u & 63);
    long r = u >> 6;
    const int sub = (int)(r & 3);
    r >>= 2;
    const int kind = (int)(r % 3);
    const long bt = r / 3;
    const int prob = (int)(bt / NT), tile = (int)(bt - (long)prob * NT), r0 = tile * 32;
    const int l31 = lane & 31, g = lane >> 5;
    const KvProblem pb = kv_problem<PERHEAD>(K, prob);
    const float* src = kind == 2 ? pb.q : pb.dout;
    float x[8];
    if (kind == 0) {
      const int row = r0 + l31;
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = row < R ? src[(size_t)row * pb.rstride + 16 * sub + 8 * g + e] : 0.f;
    } else {
      const int d = 32 * (sub >> 1) + l31, t = sub & 1;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int row = r0 + kv_row(8 * t + e, g);
        x[e] = row < R ? src[(size_t)row * pb.rstride + d] : 0.f;
      }
    }
    bf16x8 hi, lo;
    kv_split8(x, hi, lo);
    uint4* dst = K.pack + ((bt * 3 + kind) * 4 + sub) * kKvOperandUnits;
    dst[lane] = __builtin_bit_cast(uint4, hi);
    dst[kWave + lane] = __builtin_bit_cast(uint4, lo);
  }
  const long nz = (long)P.B * P.nK * (PERHEAD ? P.H : 1) * kDh / 4;
  f32x4* zk = reinterpret_cast<f32x4*>(K.dk);
  f32x4* zv = reinterpret_cast<f32x4*>(K.dv);
  for (long i = gid; i < nz; i += stride) {
    zk[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    zv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
}

// grid (2 x key tiles, B); workgroup (key tile kt, half hf): wave w walks the row tiles hf * 8 + w, + 16, ...
//
// One tile step, written as phases so that every load is in flight long before its use (with 2 waves per SIMD nothing
// else hides a round trip; a first version that let the compiler place the loads ran 8 + 8 + 8 dependent L2 round trips
// per tile: 70 us instead of 40):
//   0  issue: dO operands of dP~ (8 x 16 B), dO^T / q^T operands of contraction step t = 0 (8), lse / delta of queries 0-1
//   1  dP~ = 12 matrix instructions (V operand from the wave's LDS strip)
//   2  issue: operands of step t = 1 into the registers phase 1 freed, lse / delta of queries 2-3, then the NEXT tile's
//      scores (16 dwords; last, because vmcnt retires in order: nothing this tile waits for sits behind them)
//   3  softmax backward of queries 0-1, dS stores, step t = 0: 12 matrix instructions
//   4  the same for queries 2-3 and t = 1
// Scores and dS go through buffer instructions whose range check does the masking: a lane whose key is past nK carries
// offset 2^31, a row past the end is past num_records, so there is no branch around any load or store.
// HALVES workgroups per key tile (2: the launch alone on the chip; 1: half the workgroups, each twice as long — the per-head pass
// of the query self-attention, 256 one-per-CU workgroups at the model's size, ran four rounds on the 64 CUs a live table-gradient
// kernel leaves: 79 us instead of 24)
template <bool PERHEAD, int WAVES, int HALVES = 2>
__global__ __launch_bounds__(WAVES * kWave) void attn_bwd_kv_kernel(KvParams K) {
  constexpr int kKvWaves = WAVES, kKvThreads = WAVES * kWave, kKvSlots = HALVES * WAVES;  // row slots per key tile
  AttnParams P = K.A;
  attn_load_rng(P);
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, g = lane >> 5;
  const int kt = blockIdx.x / HALVES, hf = blockIdx.x % HALVES, prob = blockIdx.y;
  const KvProblem pb = kv_problem<PERHEAD>(K, prob);
  const int b = pb.b;
  const int R = K.R, NT = K.NT, nK = P.nK;
  const int key = kt * 32 + l31;
  const bool kvalid = key < nK;

  // B operand of dP~: V[key][16 s + 8 g + e] as (hi, lo), in the wave's LDS strip: [s][hi, lo][lane] 16-B units
  uint4* vstrip = reinterpret_cast<uint4*>(smem) + w * (8 * kWave) + lane;
  {
    const float* vrow = pb.v + (size_t)(kvalid ? key : 0) * P.v_stride + 8 * g;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(vrow + 16 * s), c = *reinterpret_cast<const f32x4*>(vrow + 16 * s + 4);
      float x[8] = {a[0], a[1], a[2], a[3], c[0], c[1], c[2], c[3]};
      if (!kvalid) {
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = 0.f;
      }
      bf16x8 hi, lo;
      kv_split8(x, hi, lo);
      vstrip[(2 * s) * kWave] = __builtin_bit_cast(uint4, hi);
      vstrip[(2 * s + 1) * kWave] = __builtin_bit_cast(uint4, lo);
    }
  }
  f32x16 accV[2], accK[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { accV[0][i] = 0.f; accV[1][i] = 0.f; accK[0][i] = 0.f; accK[1][i] = 0.f; }

  using rsrc_t = __amdgpu_buffer_rsrc_t;
  const unsigned recs = (unsigned)((size_t)R * nK * 4);  // the whole offset goes into the VGPR: it is what the range check sees
  const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(P.scores + pb.score_off, 0, (int)recs, 0x00020000);
  const rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(P.ds_out + pb.score_off, 0, (int)recs, 0x00020000);
  const int rowbytes = nK * 4;
  const unsigned lane_off = kvalid ? (unsigned)(4 * g * rowbytes + key * 4) : 0x80000000u;
  const float* lse_b = P.lse + pb.row_off;
  const float* delta_b = P.delta + pb.row_off;
  const bool has_mask = P.mask_kind == VDETR_MASK_BOOL;
  const unsigned char* mask_b = reinterpret_cast<const unsigned char*>(P.mask) + (size_t)b * P.nQ * nK + (kvalid ? key : 0);

  auto load_scores = [&](int tile, float (&sv)[16]) {
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const unsigned voff = lane_off + (unsigned)((tile * 32 + 8 * a) * rowbytes);
#pragma unroll
      for (int h = 0; h < 4; ++h)
        sv[4 * a + h] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)(voff + (unsigned)(h * rowbytes)), 0, kStreamAux));
    }
  };
  const int first = hf * kKvWaves + w;
  float sv[16], sn[16];
  if (first < NT) load_scores(first, sn);

  for (int tile = first; tile < NT; tile += kKvSlots) {
    const int r0 = tile * 32;
#pragma unroll
    for (int i = 0; i < 16; ++i) sv[i] = sn[i];
    const uint4* pk = K.pack + ((size_t)prob * NT + tile) * kKvTileUnits + lane;
    auto operand = [&](int kind, int sub, int hl) { return pk[((kind * 4 + sub) * 2 + hl) * kWave]; };
    // lse / delta of the 8 rows of contraction step t (accumulator registers 8 t .. 8 t + 7 of this lane group)
    auto row_consts = [&](int t, float (&lv)[8], float (&dv)[8]) {
#pragma unroll
      for (int a2 = 0; a2 < 2; ++a2) {
        const int rb = r0 + 8 * (2 * t + a2) + 4 * g;
        if (PERHEAD) {  // 4 consecutive queries, any nQ
#pragma unroll
          for (int h = 0; h < 4; ++h) {
            const int row = min(rb + h, R - 1);
            lv[4 * a2 + h] = lse_b[row];
            dv[4 * a2 + h] = delta_b[row];
          }
        } else {        // the 4 heads of one query: one aligned 16-B load (R is a multiple of 4)
          const f32x4 l4 = *reinterpret_cast<const f32x4*>(lse_b + min(rb, R - 4));
          const f32x4 d4 = *reinterpret_cast<const f32x4*>(delta_b + min(rb, R - 4));
#pragma unroll
          for (int h = 0; h < 4; ++h) { lv[4 * a2 + h] = l4[h]; dv[4 * a2 + h] = d4[h]; }
        }
      }
    };
    // ---- phase 0 ------------------------------------------------------------------------------------------------------
    uint4 opA[8], opB[8];  // [2 sub + hl]
#pragma unroll
    for (int s = 0; s < 4; ++s) { opA[2 * s] = operand(0, s, 0); opA[2 * s + 1] = operand(0, s, 1); }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      opB[2 * mt] = operand(1, 2 * mt, 0); opB[2 * mt + 1] = operand(1, 2 * mt, 1);
      opB[4 + 2 * mt] = operand(2, 2 * mt, 0); opB[4 + 2 * mt + 1] = operand(2, 2 * mt, 1);
    }
    float lv0[8], dv0[8];
    row_consts(0, lv0, dv0);
    __builtin_amdgcn_sched_barrier(0);
    // ---- phase 1: dP~ tile = dO tile x V^T --------------------------------------------------------------------------------
    f32x16 dp;
#pragma unroll
    for (int i = 0; i < 16; ++i) dp[i] = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const bf16x8 vh = __builtin_bit_cast(bf16x8, vstrip[(2 * s) * kWave]), vl = __builtin_bit_cast(bf16x8, vstrip[(2 * s + 1) * kWave]);
      dp = kv_mfma3(__builtin_bit_cast(bf16x8, opA[2 * s]), __builtin_bit_cast(bf16x8, opA[2 * s + 1]), vh, vl, dp);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- phase 2 ------------------------------------------------------------------------------------------------------
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      opA[2 * mt] = operand(1, 2 * mt + 1, 0); opA[2 * mt + 1] = operand(1, 2 * mt + 1, 1);
      opA[4 + 2 * mt] = operand(2, 2 * mt + 1, 0); opA[4 + 2 * mt + 1] = operand(2, 2 * mt + 1, 1);
    }
    float lv1[8], dv1[8];
    row_consts(1, lv1, dv1);
    if (tile + kKvSlots < NT) load_scores(tile + kKvSlots, sn);
    __builtin_amdgcn_sched_barrier(0);
    // ---- phases 3, 4: softmax + dropout backward of 2 queries x 4 heads, then their contraction step -----------------------
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      float xp[8], xs[8];
#pragma unroll
      for (int a2 = 0; a2 < 2; ++a2) {
        const int a = 2 * t + a2;
        const int rb = r0 + 8 * a + 4 * g;  // shared K/V: rows rb .. rb + 3 are the 4 heads of query rb / 4
        const unsigned voff = lane_off + (unsigned)((r0 + 8 * a) * rowbytes);
        uint4 rnd = make_uint4(0xFFFFu, 0xFFFFu, 0xFFFFu, 0xFFFFu);
        bool masked = false;
        if (!PERHEAD) {
          const int qi = min(rb, R - 4) >> 2;
          if (P.drop_thresh) rnd = attn_rand4(P, b, qi, key, 0);
          if (has_mask) masked = mask_b[(size_t)qi * nK] != 0;
        }
#pragma unroll
        for (int h = 0; h < 4; ++h) {
          const int v = 4 * a + h, e = 4 * a2 + h;
          const bool ok = rb + h < R && kvalid;
          int pick = h;
          if (PERHEAD) {
            const int qi = min(rb + h, R - 1);
            if (P.drop_thresh) rnd = attn_rand4(P, b, qi, key, pb.head >> 2);
            if (has_mask) masked = mask_b[(size_t)qi * nK] != 0;
            pick = pb.head & 3;
          }
          const bool keep = pick4(rnd, pick) >= P.drop_thresh;
          const ScoreGrad gr = score_grad(sv[v], t ? lv1[e] : lv0[e], keep, P.drop_scale, true, dp[v], t ? dv1[e] : dv0[e], masked);
          xp[e] = ok ? gr.p_drop : 0.f;
          xs[e] = ok ? gr.ds : 0.f;
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(gr.ds), rd, (int)(voff + (unsigned)(h * rowbytes)), 0, kStreamAux);
        }
      }
      bf16x8 ph, pl, sh, sl;
      kv_split8(xp, ph, pl);
      kv_split8(xs, sh, sl);
      const uint4* op = t ? opA : opB;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        accV[mt] = kv_mfma3(__builtin_bit_cast(bf16x8, op[2 * mt]), __builtin_bit_cast(bf16x8, op[2 * mt + 1]), ph, pl, accV[mt]);
        accK[mt] = kv_mfma3(__builtin_bit_cast(bf16x8, op[4 + 2 * mt]), __builtin_bit_cast(bf16x8, op[4 + 2 * mt + 1]), sh, sl, accK[mt]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  __syncthreads();  // the V strips are dead: the reduction reuses the memory

  // ---- the 8 waves' sums: fixed tree through LDS ([slot][register][lane]), then key-major rows for the global adds ---------
  auto put = [&](int slot) {
    float* dst = smem + (size_t)slot * 64 * kWave + lane;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      dst[i * kWave] = accV[0][i]; dst[(16 + i) * kWave] = accV[1][i];
      dst[(32 + i) * kWave] = accK[0][i]; dst[(48 + i) * kWave] = accK[1][i];
    }
  };
  auto add = [&](int slot) {
    const float* src = smem + (size_t)slot * 64 * kWave + lane;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      accV[0][i] += src[i * kWave]; accV[1][i] += src[(16 + i) * kWave];
      accK[0][i] += src[(32 + i) * kWave]; accK[1][i] += src[(48 + i) * kWave];
    }
  };
#pragma unroll
  for (int half = kKvWaves / 2; half >= 1; half >>= 1) {  // waves [half, 2 half) hand their sums to waves [0, half)
    if (w >= half && w < 2 * half) put(w - half);
    __syncthreads();
    if (w < half) add(w);
    __syncthreads();
  }
  constexpr int kFinStride = kDh + 1;
  if (w == 0) {  // fin[which][key 32][d 64 (+1)]: accumulator (d = 32 mt + kv_row(i, g), key = l31)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int d = 32 * mt + kv_row(i, g);
        smem[l31 * kFinStride + d] = accV[mt][i];
        smem[(32 + l31) * kFinStride + d] = accK[mt][i] * P.scale;
      }
  }
  __syncthreads();
  for (int idx = tid; idx < 2 * 32 * kDh; idx += kKvThreads) {
    const int which = idx >> 11, kl = (idx >> 6) & 31, d = idx & 63;
    const int kg = kt * 32 + kl;
    if (kg < nK) unsafeAtomicAdd((which ? pb.dk : pb.dv) + (size_t)kg * pb.ostride + d, smem[(which * 32 + kl) * kFinStride + d]);
  }
}

// ---- everything the key-side passes of a step need that does NOT depend on dO, for up to 16 attention calls in ONE launch -------------
// (round 6) The operand-packing launch in front of every key-side pass sat on the backward's critical chain: 16 launches of 8-12 us
// per step.  Its dO half (images of dO, delta, max |dO row|^2) is now left behind by the row-block kernel that produces dO
// (kv_pack.h: kv_emit_rows16); the rest — the q images, the zeroed dK / dV, max |V row|^2 and the box test of the RPE vertices
// (bwd_aux words 1, 4, 5) — depends on the forward only and is done here for all layers at once, at the head of the backward.
constexpr int kPrepMax = 16;
struct PrepItem {
  const float* q; const float* v; const float* vertices; const float* cos_sin;
  uint4* pack; float* dk; float* dv; unsigned* aux;
  int B, H, nQ, nK, v_stride, perhead;
};
struct PrepBatch { PrepItem it[kPrepMax]; };

__global__ __launch_bounds__(256) void attn_bwd_kv_prep_kernel(PrepBatch Bt) {
  const PrepItem& I = Bt.it[blockIdx.y];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int nvb = I.aux ? (int)(((long)I.B * I.nK + 4 * kDeltaKeys - 1) / (4 * kDeltaKeys)) : 0;
  const int nqb = (I.aux && I.vertices) ? (int)(((long)I.B * I.nQ + 3) / 4) : 0;
  int blk = blockIdx.x;
  if (blk < nvb) {  // max |V row|^2 -> aux[1] (attn_delta_body's second half)
    __shared__ float wmax[4];
    float n2max = 0.f;
    const int key0 = (blk * 4 + wv) * kDeltaKeys;
#pragma unroll 4
    for (int i = 0; i < kDeltaKeys; ++i) {
      const int key = key0 + i;
      const float x = key < I.B * I.nK ? I.v[(size_t)key * I.v_stride + lane] : 0.f;
      n2max = fmaxf(n2max, wave_allsum_f32(x * x));
    }
    if (lane == 0) wmax[wv] = n2max;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(I.aux + 1, __float_as_uint(fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]))));
    return;
  }
  blk -= nvb;
  if (blk < nqb) {  // aux[4] += 1 per query whose 8 RPE vertices are not a box, aux[5] = 1 (attn_delta_body's test, word for word)
    const int row = blk * 4 + wv;
    if (row >= I.B * I.nQ) return;
    if (row == 0 && lane == 0) I.aux[5] = 1u;
    const float* vp = I.vertices + (size_t)row * 24;
    const int i = lane & 7;
    bool ok;
    if (!I.cos_sin) {
      ok = vp[i * 3] == vp[rpe_box_xi(i) ? 6 : 0] && vp[i * 3 + 1] == vp[rpe_box_yi(i) ? 4 : 1] && vp[i * 3 + 2] == vp[rpe_box_zi(i) ? 14 : 2];
    } else {
      float ex = vp[i * 3] - vp[0], ey = vp[i * 3 + 1] - vp[1];
      const float ez = vp[i * 3 + 2] - vp[2];
      rpe_rotate(ex, ey, I.cos_sin[(size_t)row * 2], I.cos_sin[(size_t)row * 2 + 1]);
      const float EX = readlane_f32(ex, 3), EY = readlane_f32(ey, 1), EZ = readlane_f32(ez, 4);
      const float tol = 1e-5f * (1.f + fabsf(vp[0]) + fabsf(vp[1]) + fabsf(vp[2]));
      ok = fabsf(ex - (rpe_box_xi(i) ? EX : 0.f)) <= tol && fabsf(ey - (rpe_box_yi(i) ? EY : 0.f)) <= tol &&
           fabsf(ez - (rpe_box_zi(i) ? EZ : 0.f)) <= tol;
    }
    if (!__all(ok) && lane == 0) atomicAdd(I.aux + 4, 1u);
    return;
  }
  blk -= nqb;
  const int nwork = (int)gridDim.x - nvb - nqb;
  if (nwork <= 0) return;
  // ---- the q images (kind 2) and the zeroed outputs ----
  const int C = I.H * kDh;
  const int R = I.perhead ? I.nQ : I.H * I.nQ, NT = (R + 31) / 32, nprob = I.perhead ? I.B * I.H : I.B;
  const long gid = (long)blk * 256 + threadIdx.x, stride = (long)nwork * 256;
  const long nunits = (long)nprob * NT * 4 * kWave;
  for (long u = gid; u < nunits; u += stride) {
    const int ln = (int)(u & 63), sub = (int)((u >> 6) & 3);
    const long bt = u >> 8;
    const int prob = (int)(bt / NT), tile = (int)(bt - (long)prob * NT), r0 = tile * 32;
    const int l31 = ln & 31, g = ln >> 5, d = 32 * (sub >> 1) + l31, t = sub & 1;
    const float* src;
    int rstride;
    if (I.perhead) {
      const int b = prob / I.H, hh = prob - b * I.H;
      src = I.q + (size_t)b * I.nQ * C + hh * kDh;
      rstride = C;
    } else {
      src = I.q + (size_t)prob * R * kDh;
      rstride = kDh;
    }
    float x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int row = r0 + kv_row(8 * t + e, g);
      x[e] = row < R ? src[(size_t)row * rstride + d] : 0.f;
    }
    kv_store_unit(I.pack, bt, 2, sub, ln, x);
  }
  const long nz = (long)I.B * I.nK * (I.perhead ? I.H : 1) * kDh / 4;
  f32x4* zk = reinterpret_cast<f32x4*>(I.dk);
  f32x4* zv = reinterpret_cast<f32x4*>(I.dv);
  for (long i = gid; i < nz; i += stride) {
    zk[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    zv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
}

}  // namespace vdetr

using namespace vdetr;

static bool kv_supported(const vdetr_attn_desc* d) {
  return d && ((d->kind == VDETR_ATTN_SHARED_KV && d->H == 4) || d->kind == VDETR_ATTN_PER_HEAD);
}
static size_t kv_rows(const vdetr_attn_desc* d) { return d->kind == VDETR_ATTN_PER_HEAD ? (size_t)d->nQ : (size_t)d->nQ * 4; }
static size_t kv_problems(const vdetr_attn_desc* d) { return d->kind == VDETR_ATTN_PER_HEAD ? (size_t)d->B * d->H : (size_t)d->B; }

extern "C" size_t vdetr_attn_bwd_kv_workspace_bytes(const vdetr_attn_desc* d) {
  if (!kv_supported(d) || d->B <= 0 || d->nQ <= 0 || d->H <= 0) return 0;
  const size_t nt = (kv_rows(d) + 31) / 32;
  return kv_problems(d) * nt * kKvTileUnits * sizeof(uint4) + 256;
}

template <bool PERHEAD, int WAVES, int HALVES = 2>
static int kv_launch(const KvParams& K, int nkt, int nprob, long pack_work, const DeltaArgs& D, int ndelta, bool packed, hipStream_t st) {
  const unsigned pack_blocks = (unsigned)((pack_work + 255) / 256 < 2048 ? (pack_work + 255) / 256 : 2048);
  if (!packed) {  // (packed: the images, delta and the zeroed outputs are there already — vdetr_attn_bwd_kv_packed_f32)
    hipLaunchKernelGGL(attn_bwd_kv_pack_kernel<PERHEAD>, dim3(pack_blocks + (unsigned)ndelta), dim3(256), 0, st, K, D, ndelta);
    if (int e = check_launch("attn_bwd_kv_pack")) return e;
  }
  const size_t strips = (size_t)WAVES * 8 * kWave * sizeof(uint4), tree = (size_t)(WAVES / 2) * 64 * kWave * sizeof(float);
  const size_t fin = (size_t)2 * 32 * (kDh + 1) * sizeof(float);
  const size_t lds = strips > tree ? (strips > fin ? strips : fin) : (tree > fin ? tree : fin);
  if (int e = set_lds(attn_bwd_kv_kernel<PERHEAD, WAVES, HALVES>, lds, "attn_bwd_kv")) return e;
  hipLaunchKernelGGL((attn_bwd_kv_kernel<PERHEAD, WAVES, HALVES>), dim3(HALVES * nkt, nprob), dim3(WAVES * kWave), lds, st, K);
  return check_launch("attn_bwd_kv");
}

static int kv_run(const vdetr_attn_desc* d, const float* q, const float* v, const float* dout, const float* out,
                  const float* scores, const float* lse, float* delta, float* ds_out, float* dk, float* dv, void* workspace,
                  size_t workspace_bytes, vdetr_stream_t stream, bool packed = false) {
  KvParams K;
  if (int e = attn_fill_params(d, &K.A, "attn_bwd_kv")) return e;
  VDETR_REQUIRE(kv_supported(d), "attn_bwd_kv: built for shared K/V with 4 heads and for per-head K/V (kind %d, H %d)", d->kind, d->H);
  VDETR_REQUIRE(q && v && dout && scores && lse && delta && ds_out && dk && dv, "attn_bwd_kv: null pointer");
  VDETR_REQUIRE(K.A.v_stride % 4 == 0 && (((uintptr_t)v | (uintptr_t)lse | (uintptr_t)delta | (uintptr_t)dk | (uintptr_t)dv) & 15) == 0,
                "attn_bwd_kv: v rows, lse, delta, dk and dv must be 16-B aligned");
  const size_t rows = kv_rows(d), nprob = kv_problems(d);
  VDETR_REQUIRE(rows * d->nK * 4 < ((size_t)1 << 31), "attn_bwd_kv: a score matrix (%zu x %d) must stay below 2 GB", rows, d->nK);
  VDETR_REQUIRE(nprob <= 65535, "attn_bwd_kv: %zu score matrices > 65535", nprob);
  const size_t need = vdetr_attn_bwd_kv_workspace_bytes(d);
  if (!workspace || workspace_bytes < need) {
    set_error("attn_bwd_kv: workspace %zu B < required %zu B", workspace_bytes, need);
    return VDETR_ERR_WORKSPACE;
  }
  K.A.q = q; K.A.v = v;
  K.A.scores = const_cast<float*>(scores); K.A.lse = const_cast<float*>(lse); K.A.delta = delta; K.A.ds_out = ds_out;
  K.dout = dout; K.dk = dk; K.dv = dv;
  K.pack = reinterpret_cast<uint4*>(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  K.R = (int)rows;
  K.NT = (K.R + 31) / 32;
  const long units = (long)nprob * K.NT * 12 * kWave;
  const long zero4 = (long)d->B * d->nK * (d->kind == VDETR_ATTN_PER_HEAD ? d->H : 1) * kDh / 4;
  const long work = units > zero4 ? units : zero4;
  const int nkt = (d->nK + 31) / 32;
  VDETR_REQUIRE(d->kv_waves == 0 || d->kv_waves == 4 || d->kv_waves == 8, "attn_bwd_kv: kv_waves %d (0, 4 or 8)", d->kv_waves);
  VDETR_REQUIRE(d->kv_halves == 0 || d->kv_halves == 1 || d->kv_halves == 2, "attn_bwd_kv: kv_halves %d (0, 1 or 2)", d->kv_halves);
  const bool four = d->kv_waves == 4;  // (see the note at the kernel)
  const bool one = d->kv_halves == 1 && !four;
  hipStream_t st = (hipStream_t)stream;
  DeltaArgs D{};
  int ndelta = 0;
  if (out) {  // delta (and bwd_aux) produced by the first workgroups of the packing launch
    VDETR_REQUIRE(!d->bwd_aux || d->kind == VDETR_ATTN_SHARED_KV, "attn_bwd_kv: bwd_aux needs the shared-KV kind");
    attn_delta_args(d, dout, out, v, delta, &D);
    ndelta = D.qblocks + D.vblocks;
  }
  if (d->kind == VDETR_ATTN_PER_HEAD) {
    if (one) return kv_launch<true, 8, 1>(K, nkt, (int)nprob, work, D, ndelta, packed, st);
    return four ? kv_launch<true, 4>(K, nkt, (int)nprob, work, D, ndelta, packed, st) : kv_launch<true, 8>(K, nkt, (int)nprob, work, D, ndelta, packed, st);
  }
  if (one) return kv_launch<false, 8, 1>(K, nkt, (int)nprob, work, D, ndelta, packed, st);
  return four ? kv_launch<false, 4>(K, nkt, (int)nprob, work, D, ndelta, packed, st) : kv_launch<false, 8>(K, nkt, (int)nprob, work, D, ndelta, packed, st);
}

extern "C" int vdetr_attn_bwd_kv_f32(const vdetr_attn_desc* d, const float* q, const float* v, const float* dout,
                                     const float* scores, const float* lse, const float* delta, float* ds_out, float* dk,
                                     float* dv, void* workspace, size_t workspace_bytes, vdetr_stream_t stream) {
  return kv_run(d, q, v, dout, nullptr, scores, lse, const_cast<float*>(delta), ds_out, dk, dv, workspace, workspace_bytes, stream);
}

extern "C" int vdetr_attn_bwd_kv_delta_f32(const vdetr_attn_desc* d, const float* q, const float* v, const float* dout,
                                           const float* out, const float* scores, const float* lse, float* delta, float* ds_out,
                                           float* dk, float* dv, void* workspace, size_t workspace_bytes, vdetr_stream_t stream) {
  VDETR_REQUIRE(out, "attn_bwd_kv_delta: null pointer (out)");
  return kv_run(d, q, v, dout, out, scores, lse, delta, ds_out, dk, dv, workspace, workspace_bytes, stream);
}

// The key-side pass on operands that are ALREADY packed: `workspace` holds the images (dO's from vdetr_rb_ffn_bwd_emit_f32 /
// vdetr_rb_proj_q_bwd_emit_f32, q's from vdetr_attn_bwd_kv_prep_f32), `delta` is read, dk / dv are zero, bwd_aux is filled.
extern "C" int vdetr_attn_bwd_kv_packed_f32(const vdetr_attn_desc* d, const float* q, const float* v, const float* dout, const float* scores,
                                            const float* lse, const float* delta, float* ds_out, float* dk, float* dv, void* workspace,
                                            size_t workspace_bytes, vdetr_stream_t stream) {
  VDETR_REQUIRE(workspace && (((uintptr_t)workspace) & 255) == 0, "attn_bwd_kv_packed: the workspace (the operand images) must be 256-B aligned");
  return kv_run(d, q, v, dout, nullptr, scores, lse, const_cast<float*>(delta), ds_out, dk, dv, workspace, workspace_bytes, stream, true);
}

extern "C" int vdetr_attn_bwd_kv_prep_f32(const vdetr_attn_kv_prep* items, int n, vdetr_stream_t stream) {
  VDETR_REQUIRE(items && n >= 1 && n <= kPrepMax, "attn_bwd_kv_prep: %d items (1..%d)", n, kPrepMax);
  PrepBatch Bt;
  long blocks = 1;
  for (int i = 0; i < n; ++i) {
    const vdetr_attn_kv_prep& s = items[i];
    VDETR_REQUIRE(s.q && s.v && s.workspace && s.dk && s.dv, "attn_bwd_kv_prep: null pointer in item %d", i);
    VDETR_REQUIRE(s.kind == VDETR_ATTN_SHARED_KV || s.kind == VDETR_ATTN_PER_HEAD, "attn_bwd_kv_prep: kind %d", s.kind);
    VDETR_REQUIRE(s.B > 0 && s.nQ > 0 && s.nK > 0 && s.H == 4, "attn_bwd_kv_prep: item %d: B=%d nQ=%d nK=%d H=%d (4 heads)", i, s.B, s.nQ, s.nK, s.H);
    VDETR_REQUIRE((((uintptr_t)s.workspace) & 255) == 0 && (((uintptr_t)s.dk | (uintptr_t)s.dv) & 15) == 0, "attn_bwd_kv_prep: item %d: alignment", i);
    VDETR_REQUIRE(s.kind == VDETR_ATTN_SHARED_KV || !s.bwd_aux, "attn_bwd_kv_prep: bwd_aux needs the shared-KV kind");
    PrepItem& I = Bt.it[i];
    I.q = s.q; I.v = s.v; I.vertices = s.vertices; I.cos_sin = s.vertices ? s.cos_sin : nullptr;
    I.pack = reinterpret_cast<uint4*>(s.workspace); I.dk = s.dk; I.dv = s.dv; I.aux = s.bwd_aux;
    I.B = s.B; I.H = s.H; I.nQ = s.nQ; I.nK = s.nK;
    I.perhead = s.kind == VDETR_ATTN_PER_HEAD ? 1 : 0;
    I.v_stride = s.v_row_stride ? s.v_row_stride : (I.perhead ? s.H * 64 : 64);
    const long nvb = s.bwd_aux ? ((long)s.B * s.nK + 4 * kDeltaKeys - 1) / (4 * kDeltaKeys) : 0;
    const long nqb = (s.bwd_aux && s.vertices) ? ((long)s.B * s.nQ + 3) / 4 : 0;
    const long R = I.perhead ? s.nQ : (long)s.H * s.nQ, NT = (R + 31) / 32, nprob = I.perhead ? (long)s.B * s.H : s.B;
    const long units = nprob * NT * 4 * kWave, zero4 = (long)s.B * s.nK * (I.perhead ? s.H : 1) * kDh / 4;
    long work = ((units > zero4 ? units : zero4) + 255) / 256;
    work = work < 1 ? 1 : (work > 512 ? 512 : work);
    blocks = blocks > nvb + nqb + work ? blocks : nvb + nqb + work;
  }
  for (int i = n; i < kPrepMax; ++i) Bt.it[i] = Bt.it[0];
  hipLaunchKernelGGL(attn_bwd_kv_prep_kernel, dim3((unsigned)blocks, n), dim3(256), 0, (hipStream_t)stream, Bt);
  return check_launch("attn_bwd_kv_prep");
}
