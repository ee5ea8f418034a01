// attn_bwd.hip — backward "score stage" of the attention + the 3DV-RPE table gradient, for gfx950.
//
// Backward of out = dropout(softmax(S)) V with S = scale*q k^T + rpe + mask is split as
//     dP~ = dO V^T                      (plain GEMM, library)
//     P~  = dropout(softmax(S)),  dS = softmax(S) * (dropout'(dP~) - rowsum(dO*O))     <- THIS FILE
//     dV = P~^T dO,  dK = dS^T q,  dQ = dS K      (plain GEMMs, library)
//     dTable[i,cell,h] += trilinear_weight(pair, cell) * dS[h, pair]                  <- THIS FILE
// The forward saved S (the biased scores) and the row log-sum-exp, so nothing of the QK^T / table lookup
// is recomputed here except the per-pair lookup GEOMETRY (cell + 3 fractions per vertex), which is cheaper
// to recompute than to store (16 B x 8 vertices x nQ x nK).
// The table gradient is a 32,000-bin weighted histogram with 8*8*4*nQ*nK contributions
// (vdetr_transformer.py:725-731 backward = grid_sampler_3d_backward, 47 % of the reference layer time on
// CPU).  Each workgroup keeps a private copy of the histogram in LDS (128 KB) and adds to it with
// ds_add_f32; the copies are written to a workspace and summed by a second, coalesced kernel —
// no global atomics.
#include "attn_common.h"

#include <stdlib.h>

namespace vdetr {

int attn_fill_params(const vdetr_attn_desc* d, AttnParams* P, const char* op);

constexpr int kBwdThreads = 1024;

// element-wise part shared by both kernels: returns P~ and dS for one head of one (q,key) pair
struct ScoreGrad {
  float p_drop, ds;
};
__device__ __forceinline__ ScoreGrad score_grad(float s, float lse, bool keep, float drop_scale, bool have_grad,
                                                float dprob, float delta, bool masked) {
  const float p = __expf(s - lse);
  ScoreGrad r;
  r.p_drop = keep ? p * drop_scale : 0.f;
  float ds = 0.f;
  if (have_grad) {
    const float dp = keep ? dprob * drop_scale : 0.f;
    ds = p * (dp - delta);
    if (masked) ds = 0.f;  // masked_fill_ overwrote the score: no gradient reaches q, k or the table
  }
  r.ds = ds;
  return r;
}

// ---- generic kernel (no RPE): one thread per score element ----------------------------------------------
__global__ __launch_bounds__(256) void attn_bwd_scores_kernel(AttnParams P) {
  attn_load_rng(P);
  const size_t total = (size_t)P.B * P.H * P.nQ * P.nK;
  const bool perhead = P.kind == VDETR_ATTN_PER_HEAD;
  const bool have_grad = P.dprob != nullptr;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const size_t row = e / P.nK;
    const int key = (int)(e - row * P.nK);
    int b, q, h;
    if (perhead) {  // [B,H,nQ,nK]
      q = (int)(row % P.nQ);
      h = (int)((row / P.nQ) % P.H);
      b = (int)(row / ((size_t)P.nQ * P.H));
    } else {  // [B,nQ,H,nK]
      h = (int)(row % P.H);
      q = (int)((row / P.H) % P.nQ);
      b = (int)(row / ((size_t)P.nQ * P.H));
    }
    bool keep = true;
    if (P.drop_thresh) keep = pick4(attn_rand4(P, b, q, key, h >> 2), h & 3) >= P.drop_thresh;
    const bool masked = P.mask_kind == VDETR_MASK_BOOL &&
                        reinterpret_cast<const unsigned char*>(P.mask)[((size_t)b * P.nQ + q) * P.nK + key];
    const ScoreGrad g = score_grad(P.scores[e], P.lse[row], keep, P.drop_scale, have_grad,
                                   have_grad ? P.dprob[e] : 0.f, have_grad ? P.delta[row] : 0.f, masked);
    P.scores[e] = g.p_drop;
    if (have_grad) P.dprob[e] = g.ds;
  }
}

// ---- RPE kernel ------------------------------------------------------------------------------------------
// One (batch, query) item at a time per workgroup; a wave takes 64 consecutive keys, one (query,key) pair per lane
// with the 4 heads in registers.
//
// Table gradient.  A pair contributes w_corner * dS[h] to 8 corners x 4 heads of each of the 8 vertex tables.  The
// log-spaced table makes far cells huge, so most of a wave's 64 (spatially neighbouring, see the Morton ordering in
// the host module) keys hit the SAME cell — the worst case for LDS atomics (same address = serialised).  Variant 1
// therefore aggregates inside the wave first: for each distinct base cell among the 64 lanes (usually 1-3) the 32
// products are summed over the member lanes with a 6-stage reduce-scatter (permlane32/16 swaps + DPP), after which
// lane 2j holds the wave total of value j and ONE ds_add_f32 instruction with 32 distinct addresses updates the
// histogram.  Variant 0 (plain per-lane atomics) is kept for A/B measurements.
template <int CTRL>
__device__ __forceinline__ float dpp_recv(float v) {
  return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), CTRL, 0xF, 0xF, false));
}

// v[32] per lane -> returns, in lane l, the sum over all 64 lanes of v[l >> 1]
__device__ __forceinline__ float wave_reduce_scatter32(float (&v)[32], int lane) {
  float a[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) {  // halves: value bit 4 <- lane bit 5
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[j]), __float_as_uint(v[j + 16]), false, false);
    a[j] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
  float b[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {  // row pairs: value bit 3 <- lane bit 4
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a[j]), __float_as_uint(a[j + 8]), false, false);
    b[j] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
  float c[4];
  const bool b3 = lane & 8, b2 = lane & 4, b1 = lane & 2;
#pragma unroll
  for (int j = 0; j < 4; ++j) {  // lane ^ 8 (row_ror:8): value bit 2 <- lane bit 3
    const float keep = b3 ? b[j + 4] : b[j], send = b3 ? b[j] : b[j + 4];
    c[j] = keep + dpp_recv<0x128>(send);
  }
  float d[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {  // row_half_mirror (c <-> 7-c): value bit 1 <- lane bit 2
    const float keep = b2 ? c[j + 2] : c[j], send = b2 ? c[j] : c[j + 2];
    d[j] = keep + dpp_recv<kDppRowHalfMirror>(send);
  }
  const float keep = b1 ? d[1] : d[0], send = b1 ? d[0] : d[1];  // quad xor 2: value bit 0 <- lane bit 1
  float e = keep + dpp_recv<kDppQuadXor2>(send);
  e += dpp_recv<kDppQuadXor1>(e);  // quad xor 1: both lanes of a pair hold the total
  return e;
}

template <int VARIANT>
__global__ __launch_bounds__(kBwdThreads) void attn_bwd_scores_rpe_kernel(AttnParams P) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // dTable copy [8][T^3][4]
  attn_load_rng(P);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int T = P.T, TT = T * T, T3 = TT * T;
  const int table_floats = kRpeVerts * T3 * 4;
  const bool want_table = P.dtable_part != nullptr;
  if (want_table) {
    for (int i = tid; i < table_floats; i += kBwdThreads) smem[i] = 0.f;
    __syncthreads();
  }
  const bool have_grad = P.dprob != nullptr;
  const bool rot = P.cos_sin != nullptr;
  const int items = P.B * P.nQ;
  const int nchunks = (P.nK + kWave - 1) / kWave;
  // lane l adds value j = l>>1 = corner*4 + h (corner = cz*4 + cy*2 + cx) when l is even
  const int jv = lane >> 1;
  const int my_off = (((jv >> 4) & 1) * TT + ((jv >> 3) & 1) * T + ((jv >> 2) & 1)) * 4 + (jv & 3);
  for (int item = blockIdx.x; item < items; item += gridDim.x) {
    const int b = item / P.nQ, q = item - b * P.nQ;
    const size_t row0 = ((size_t)b * P.nQ + q) * 4;  // rows (b,q,h) for h = 0..3
    float lse[4], delta[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) { lse[h] = P.lse[row0 + h]; delta[h] = have_grad ? P.delta[row0 + h] : 0.f; }
    float vx[8], vy[8], vz[8];
    const float* vp = P.vertices + ((size_t)b * P.nQ + q) * 24;
#pragma unroll
    for (int i = 0; i < 8; ++i) { vx[i] = vp[i * 3]; vy[i] = vp[i * 3 + 1]; vz[i] = vp[i * 3 + 2]; }
    const float rc = rot ? P.cos_sin[((size_t)b * P.nQ + q) * 2] : 1.f;
    const float rs = rot ? P.cos_sin[((size_t)b * P.nQ + q) * 2 + 1] : 0.f;

    for (int chunk = w; chunk < nchunks; chunk += kBwdThreads / kWave) {
      const int key = chunk * kWave + lane;
      const bool valid = key < P.nK;
      const int keyc = valid ? key : P.nK - 1;
      float ds[4] = {0.f, 0.f, 0.f, 0.f};
      if (valid) {
        uint4 rnd = make_uint4(0xFFFFu, 0xFFFFu, 0xFFFFu, 0xFFFFu);
        if (P.drop_thresh) rnd = attn_rand4(P, b, q, key, 0);
        const bool masked = P.mask_kind == VDETR_MASK_BOOL &&
                            reinterpret_cast<const unsigned char*>(P.mask)[((size_t)b * P.nQ + q) * P.nK + key];
#pragma unroll
        for (int h = 0; h < 4; ++h) {
          const size_t e = (row0 + h) * P.nK + key;
          const bool keep = pick4(rnd, h) >= P.drop_thresh;
          const ScoreGrad g = score_grad(P.scores[e], lse[h], keep, P.drop_scale, have_grad,
                                         have_grad ? P.dprob[e] : 0.f, delta[h], masked);
          P.scores[e] = g.p_drop;
          if (have_grad) P.dprob[e] = g.ds;
          ds[h] = g.ds;
        }
      }
      if (!want_table) continue;
      const float* xp = P.xyz + ((size_t)b * P.nK + keyc) * 3;
      const float kx = xp[0], ky = xp[1], kz = xp[2];
#pragma unroll
      for (int i = 0; i < kRpeVerts; ++i) {
        float dx = vx[i] - kx, dy = vy[i] - ky, dz = vz[i] - kz;
        if (rot) rpe_rotate(dx, dy, rc, rs);
        const AxisTap ax = rpe_axis(dx, P), ay = rpe_axis(dy, P), az = rpe_axis(dz, P);
        const int cell = i * T3 + rpe_cell(ax, ay, az, T);
        const float wz[2] = {az.wa, az.wb}, wy[2] = {ay.wa, ay.wb}, wx[2] = {ax.wa, ax.wb};
        if (VARIANT == 0) {
          float* t = smem + (size_t)cell * 4;
#pragma unroll
          for (int c = 0; c < 8; ++c) {
            const float wgt = wz[c >> 2] * wy[(c >> 1) & 1] * wx[c & 1];
            float* cp = t + ((c >> 2) * TT + ((c >> 1) & 1) * T + (c & 1)) * 4;
#pragma unroll
            for (int h = 0; h < 4; ++h) atomicAdd(cp + h, wgt * ds[h]);
          }
        } else {
          float wgt[8];
#pragma unroll
          for (int c = 0; c < 8; ++c) wgt[c] = wz[c >> 2] * wy[(c >> 1) & 1] * wx[c & 1];
          unsigned long long todo = ~0ull;  // every lane takes part (out-of-range keys carry ds = 0)
          while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const int c0 = __builtin_amdgcn_readlane(cell, leader);
            const bool member = cell == c0;
            todo &= ~__ballot(member);
            float v[32];
#pragma unroll
            for (int c = 0; c < 8; ++c)
#pragma unroll
              for (int h = 0; h < 4; ++h) v[c * 4 + h] = member ? wgt[c] * ds[h] : 0.f;
            const float total = wave_reduce_scatter32(v, lane);
            if (!(lane & 1)) atomicAdd(smem + (size_t)c0 * 4 + my_off, total);
          }
        }
      }
    }
  }
  if (want_table) {
    __syncthreads();
    float* dst = P.dtable_part + (size_t)blockIdx.x * table_floats;
    for (int i = tid; i < table_floats; i += kBwdThreads) dst[i] = smem[i];
  }
}

// ---- RPE kernel, matrix-core variant (default) --------------------------------------------------------------------
// The wave-level aggregation of the table gradient IS a small matrix product:
//     G[group][value] = sum over the 64 lanes  M[group][lane] * V[lane][value]
// with M the 0/1 membership of a lane (pair) in a group (= distinct lookup cell among the wave's lanes, <= 16 per
// round) and V the lane's 32 products weight(corner) * dS(head).  Variant 1 evaluates it group by group with a
// masked 6-stage shuffle reduction (~125 VALU ops per group, 7.9 groups per wave step measured); here it is 16
// v_mfma_f32_16x16x4_f32 per 16 values for ALL groups at once — exact fp32 (products with 0/1, fp32 accumulation
// in a fixed order, so the result is also run-to-run deterministic up to the final LDS adds).
//   lane l as A operand: row = group l&15, k-slot l>>4 -> membership of pair 4s + (l>>4)   (cells via an LDS strip)
//   lane l as B operand: k-slot l>>4, column = value l&15        -> V[pair 4s + (l>>4)][l&15] (values via an LDS strip)
//   result: lane l holds value l&15 of groups 4*(l>>4)+r -> 4 ds_add_f32 with distinct addresses.
constexpr int kMmThreads = 512;
constexpr int kMmWaves = kMmThreads / kWave;
constexpr int kMmStripFloats = kWave + kWave * 16;  // cell ids + 64 x 16 values

__global__ __launch_bounds__(kMmThreads) void attn_bwd_scores_rpe_mm_kernel(AttnParams P) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [dTable copy 8*T^3*4][8 strips]
  attn_load_rng(P);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int T = P.T, TT = T * T, T3 = TT * T;
  const int table_floats = kRpeVerts * T3 * 4;
  for (int i = tid; i < table_floats; i += kMmThreads) smem[i] = 0.f;
  __syncthreads();
  int* cellbuf = reinterpret_cast<int*>(smem + table_floats + w * kMmStripFloats);
  float* vbuf = smem + table_floats + w * kMmStripFloats + kWave;
  const bool rot = P.cos_sin != nullptr;
  const int items = P.B * P.nQ;
  const int nchunks = (P.nK + kWave - 1) / kWave;
  const int kk = lane >> 4, c15 = lane & 15;
  // bin offsets of this lane's output column for the two 16-value tiles (tile jt = corners 4jt..4jt+3)
  int off[2];
#pragma unroll
  for (int jt = 0; jt < 2; ++jt) {
    const int corner = jt * 4 + (c15 >> 2);
    off[jt] = (((corner >> 2) & 1) * TT + ((corner >> 1) & 1) * T + (corner & 1)) * 4 + (c15 & 3);
  }
  // swizzled strip columns (bank-conflict-free 16-B writes AND 4-B reads)
  const int wr_sw = lane >> 1;

  for (int item = blockIdx.x; item < items; item += gridDim.x) {
    const int b = item / P.nQ, q = item - b * P.nQ;
    const size_t row0 = ((size_t)b * P.nQ + q) * 4;
    float lse[4], delta[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) { lse[h] = P.lse[row0 + h]; delta[h] = P.delta[row0 + h]; }
    float vx[8], vy[8], vz[8];
    const float* vp = P.vertices + ((size_t)b * P.nQ + q) * 24;
#pragma unroll
    for (int i = 0; i < 8; ++i) { vx[i] = vp[i * 3]; vy[i] = vp[i * 3 + 1]; vz[i] = vp[i * 3 + 2]; }
    const float rc = rot ? P.cos_sin[((size_t)b * P.nQ + q) * 2] : 1.f;
    const float rs = rot ? P.cos_sin[((size_t)b * P.nQ + q) * 2 + 1] : 0.f;

    for (int chunk = w; chunk < nchunks; chunk += kMmWaves) {
      const int key = chunk * kWave + lane;
      const bool valid = key < P.nK;
      const int keyc = valid ? key : P.nK - 1;
      float ds[4] = {0.f, 0.f, 0.f, 0.f};
      if (valid) {
        uint4 rnd = make_uint4(0xFFFFu, 0xFFFFu, 0xFFFFu, 0xFFFFu);
        if (P.drop_thresh) rnd = attn_rand4(P, b, q, key, 0);
        const bool masked = P.mask_kind == VDETR_MASK_BOOL &&
                            reinterpret_cast<const unsigned char*>(P.mask)[((size_t)b * P.nQ + q) * P.nK + key];
#pragma unroll
        for (int h = 0; h < 4; ++h) {
          const size_t e = (row0 + h) * P.nK + key;
          const bool keep = pick4(rnd, h) >= P.drop_thresh;
          const ScoreGrad g = score_grad(P.scores[e], lse[h], keep, P.drop_scale, true, P.dprob[e], delta[h], masked);
          P.scores[e] = g.p_drop;
          P.dprob[e] = g.ds;
          ds[h] = g.ds;
        }
      }
      const float* xp = P.xyz + ((size_t)b * P.nK + keyc) * 3;
      const float kx = xp[0], ky = xp[1], kz = xp[2];
#pragma unroll 1
      for (int i = 0; i < kRpeVerts; ++i) {
        float dx = vx[i] - kx, dy = vy[i] - ky, dz = vz[i] - kz;
        if (rot) rpe_rotate(dx, dy, rc, rs);
        const AxisTap ax = rpe_axis(dx, P), ay = rpe_axis(dy, P), az = rpe_axis(dz, P);
        const int cell = i * T3 + rpe_cell(ax, ay, az, T);
        const float w00 = az.wa * ay.wa, w01 = az.wa * ay.wb, w10 = az.wb * ay.wa, w11 = az.wb * ay.wb;
        const float wgt[8] = {w00 * ax.wa, w00 * ax.wb, w01 * ax.wa, w01 * ax.wb,
                              w10 * ax.wa, w10 * ax.wb, w11 * ax.wa, w11 * ax.wb};
        // cells of all 64 pairs, laid out so that k-slot kk reads pairs kk, 4+kk, 8+kk, ... as 4 float4
        __builtin_amdgcn_wave_barrier();
        cellbuf[(lane & 3) * 16 + (lane >> 2)] = cell;
        __builtin_amdgcn_wave_barrier();
        int cr[16];
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) {
          const int4 v4 = *reinterpret_cast<const int4*>(cellbuf + kk * 16 + t4 * 4);
          cr[t4 * 4] = v4.x; cr[t4 * 4 + 1] = v4.y; cr[t4 * 4 + 2] = v4.z; cr[t4 * 4 + 3] = v4.w;
        }
        unsigned long long todo = ~0ull;
        while (todo) {
          // up to 16 distinct cells of this round: lane g keeps the cell of group g
          int mygcell = -1, ng = 0;
          while (todo && ng < 16) {
            const int leader = __ffsll((long long)todo) - 1;
            const int c0 = __builtin_amdgcn_readlane(cell, leader);
            todo &= ~__ballot(cell == c0);
            if (lane == ng) mygcell = c0;
            ++ng;
          }
          const int rowcell = __shfl(mygcell, c15);
          float am[16];
#pragma unroll
          for (int s2 = 0; s2 < 16; ++s2) am[s2] = cr[s2] == rowcell ? 1.f : 0.f;
          int gc[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) gc[r] = __shfl(mygcell, 4 * kk + r);
#pragma unroll
          for (int jt = 0; jt < 2; ++jt) {
            // V strip: my 16 values (4 corners x 4 heads), 16-B blocks rotated by lane>>1
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
              const float wc = wgt[jt * 4 + cc];
              *reinterpret_cast<f32x4*>(vbuf + lane * 16 + ((cc + wr_sw) & 3) * 4) =
                  f32x4{wc * ds[0], wc * ds[1], wc * ds[2], wc * ds[3]};
            }
            __builtin_amdgcn_wave_barrier();
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s2 = 0; s2 < 16; ++s2) {
              const int p = 4 * s2 + kk;
              const float bv = vbuf[p * 16 + (((c15 >> 2) + (p >> 1)) & 3) * 4 + (c15 & 3)];
              acc = __builtin_amdgcn_mfma_f32_16x16x4f32(am[s2], bv, acc, 0, 0, 0);
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (gc[r] >= 0) atomicAdd(smem + (size_t)gc[r] * 4 + off[jt], acc[r]);
          }
        }
      }
    }
  }
  __syncthreads();
  float* dst = P.dtable_part + (size_t)blockIdx.x * table_floats;
  for (int i = tid; i < table_floats; i += kMmThreads) dst[i] = smem[i];
}

// ---- RPE kernel, producer/consumer variant (default) -------------------------------------------------------------
// The table gradient needs, per (pair, vertex), 8 corners x 4 heads = 32 products added into 32 histogram bins.
// Aggregating them ACROSS lanes (variant 1) costs a masked 6-stage reduce-scatter per distinct cell (measured: 7.9
// distinct cells per 64 Morton-consecutive keys).  This variant turns the problem by 90 degrees:
//   producer  lane = pair, as before: per vertex it computes only the lookup RECORD (cell id + the 3 sub-cell
//             coordinates, 16 B) and parks it, with the pair's dS[4], in a wave-private LDS strip;
//   consumer  lane = BIN: lane (half, j) owns corner j>>2 / head j&3 of one vertex per half-wave and walks the 64
//             records of the strip in order.  All 32 lanes of a half read the same record (LDS broadcast), build
//             their own corner weight with 6 VALU ops, and add weight*dS[h] into ONE register.  Because every
//             lane of the half sees the same cell sequence, the run-length accumulation is uniform: when the cell
//             changes (34 % of the steps on Morton-ordered keys) the half flushes with a single ds_add_f32 whose 32
//             addresses are distinct — no same-address serialisation, no cross-lane reduction at all.
// ~13 VALU + 2 LDS reads per (pair, vertex) for 32 bins, against ~150 VALU per distinct cell before.
constexpr int kPcThreads = 512;
constexpr int kPcWaves = kPcThreads / kWave;
constexpr int kPcStripFloats = 3 * kWave * 4;  // 2 vertex records + dS, float4 each, per lane

__global__ __launch_bounds__(kPcThreads) void attn_bwd_scores_rpe_pc_kernel(AttnParams P) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [dTable copy 8*T^3*4][8 strips]
  attn_load_rng(P);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int T = P.T, TT = T * T, T3 = TT * T;
  const int table_floats = kRpeVerts * T3 * 4;
  for (int i = tid; i < table_floats; i += kPcThreads) smem[i] = 0.f;
  __syncthreads();
  f32x4* strip = reinterpret_cast<f32x4*>(smem + table_floats + w * kPcStripFloats);  // [3][64]
  const bool rot = P.cos_sin != nullptr;
  const int items = P.B * P.nQ;
  const int nchunks = (P.nK + kWave - 1) / kWave;
  // consumer role of this lane
  const int half = lane >> 5, jv = lane & 31;
  const float czf = (float)((jv >> 4) & 1), cyf = (float)((jv >> 3) & 1), cxf = (float)((jv >> 2) & 1);
  const int hsel = jv & 3;
  const int my_off = (((jv >> 4) & 1) * TT + ((jv >> 3) & 1) * T + ((jv >> 2) & 1)) * 4 + hsel;

  for (int item = blockIdx.x; item < items; item += gridDim.x) {
    const int b = item / P.nQ, q = item - b * P.nQ;
    const size_t row0 = ((size_t)b * P.nQ + q) * 4;
    float lse[4], delta[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) { lse[h] = P.lse[row0 + h]; delta[h] = P.delta[row0 + h]; }
    float vx[8], vy[8], vz[8];
    const float* vp = P.vertices + ((size_t)b * P.nQ + q) * 24;
#pragma unroll
    for (int i = 0; i < 8; ++i) { vx[i] = vp[i * 3]; vy[i] = vp[i * 3 + 1]; vz[i] = vp[i * 3 + 2]; }
    const float rc = rot ? P.cos_sin[((size_t)b * P.nQ + q) * 2] : 1.f;
    const float rs = rot ? P.cos_sin[((size_t)b * P.nQ + q) * 2 + 1] : 0.f;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};  // one running bin per vertex group (this half's vertex 2*g + half)
    int cur[4] = {-1, -1, -1, -1};

    for (int chunk = w; chunk < nchunks; chunk += kPcWaves) {
      // ---- producer: element-wise softmax backward for this lane's pair ------------------------------------
      const int key = chunk * kWave + lane;
      const bool valid = key < P.nK;
      const int keyc = valid ? key : P.nK - 1;
      f32x4 ds = {0.f, 0.f, 0.f, 0.f};
      if (valid) {
        uint4 rnd = make_uint4(0xFFFFu, 0xFFFFu, 0xFFFFu, 0xFFFFu);
        if (P.drop_thresh) rnd = attn_rand4(P, b, q, key, 0);
        const bool masked = P.mask_kind == VDETR_MASK_BOOL &&
                            reinterpret_cast<const unsigned char*>(P.mask)[((size_t)b * P.nQ + q) * P.nK + key];
#pragma unroll
        for (int h = 0; h < 4; ++h) {
          const size_t e = (row0 + h) * P.nK + key;
          const bool keep = pick4(rnd, h) >= P.drop_thresh;
          const ScoreGrad g = score_grad(P.scores[e], lse[h], keep, P.drop_scale, true, P.dprob[e], delta[h], masked);
          P.scores[e] = g.p_drop;
          P.dprob[e] = g.ds;
          ds[h] = g.ds;
        }
      }
      const float* xp = P.xyz + ((size_t)b * P.nK + keyc) * 3;
      const float kx = xp[0], ky = xp[1], kz = xp[2];
      __builtin_amdgcn_wave_barrier();
      strip[2 * kWave + lane] = ds;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        // ---- producer: lookup records of vertices 2g, 2g+1 ------------------------------------------------
#pragma unroll
        for (int hv = 0; hv < 2; ++hv) {
          const int i = 2 * g + hv;
          float dx = vx[i] - kx, dy = vy[i] - ky, dz = vz[i] - kz;
          if (rot) rpe_rotate(dx, dy, rc, rs);
          float tx, ty, tz;
          const int bx = rpe_axis_base(dx, P, tx), by = rpe_axis_base(dy, P, ty), bz = rpe_axis_base(dz, P, tz);
          const int cell = i * T3 + (bz * T + by) * T + bx;
          strip[hv * kWave + lane] = f32x4{__int_as_float(cell), tz, ty, tx};
        }
        __builtin_amdgcn_wave_barrier();
        // ---- consumer: walk the 64 records of this half's vertex ---------------------------------------------
        const f32x4* rec = strip + half * kWave;
        const float* dsr = reinterpret_cast<const float*>(strip + 2 * kWave) + hsel;
        float a = acc[g];
        int c = cur[g];
#pragma unroll 4
        for (int p = 0; p < kWave; ++p) {
          const f32x4 r = rec[p];
          const float d = dsr[p * 4];
          const int cell = __float_as_int(r[0]);
          const float wz = __saturatef(1.f - fabsf(r[1] - czf));
          const float wy = __saturatef(1.f - fabsf(r[2] - cyf));
          const float wx = __saturatef(1.f - fabsf(r[3] - cxf));
          if (cell != c) {
            if (c >= 0) atomicAdd(smem + (size_t)c * 4 + my_off, a);
            a = 0.f;
            c = cell;
          }
          a = __builtin_fmaf(wz * wy * wx, d, a);
        }
        acc[g] = a;
        cur[g] = c;
        __builtin_amdgcn_wave_barrier();
      }
    }
#pragma unroll
    for (int g = 0; g < 4; ++g)
      if (cur[g] >= 0) atomicAdd(smem + (size_t)cur[g] * 4 + my_off, acc[g]);
  }
  __syncthreads();
  float* dst = P.dtable_part + (size_t)blockIdx.x * table_floats;
  for (int i = tid; i < table_floats; i += kPcThreads) dst[i] = smem[i];
}

// ---- RPE kernel, run-length variant (default) ---------------------------------------------------------------
// Lanes own QUERIES and walk the (Morton-ordered) keys one by one: lane l = (query l>>1 of a 32-query group,
// vertex half l&1 -> 4 of the 8 vertices).  A (query, vertex) sees its lookup cell change only when the key
// stream crosses one of ITS cell boundaries, so the 32 corner x head partial sums of the current cell live in
// registers and are only flushed to the LDS histogram (32 ds_add_f32, lanes that flush hit different cells) when
// the cell changes or the key range ends.  Per key step a wave spends ~40 VALU ops of geometry + 44 of
// weights/FMAs per vertex, independent of how many distinct cells its lanes touch — the leader-loop variant
// above pays ~150 ops per distinct cell.  S / dP~ are read as float4 along the key axis (each lane streams its
// own 4 rows); the two lanes of a query read the same addresses, lane 0 of the pair writes P~ / dS back.
constexpr int kRlThreads = 512;
constexpr int kRlQueries = 32;  // queries per wave (2 lanes each)

__device__ __forceinline__ f32x4 load4(const float* p, int n_valid, bool aligned) {
  if (aligned && n_valid >= 4) return *reinterpret_cast<const f32x4*>(p);
  f32x4 r = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (j < n_valid) r[j] = p[j];
  return r;
}
__device__ __forceinline__ void store4(float* p, const f32x4& v, int n_valid, bool aligned) {
  if (aligned && n_valid >= 4) {
    *reinterpret_cast<f32x4*>(p) = v;
    return;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (j < n_valid) p[j] = v[j];
}

__global__ __launch_bounds__(kRlThreads) void attn_bwd_scores_rpe_rl_kernel(AttnParams P, int keys_per_wg) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // dTable copy [8][T^3][4]
  attn_load_rng(P);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int T = P.T, TT = T * T, T3 = TT * T;
  const int table_floats = kRpeVerts * T3 * 4;
  for (int i = tid; i < table_floats; i += kRlThreads) smem[i] = 0.f;
  __syncthreads();
  const bool rot = P.cos_sin != nullptr;
  const int b = blockIdx.z;
  const int q = blockIdx.x * kRlQueries + (lane >> 1);
  const bool qvalid = q < P.nQ;
  const int qc = qvalid ? q : P.nQ - 1;
  const int vh = lane & 1;
  const size_t row0 = ((size_t)b * P.nQ + qc) * 4;
  float lse[4], delta[4];
#pragma unroll
  for (int h = 0; h < 4; ++h) { lse[h] = P.lse[row0 + h]; delta[h] = P.delta[row0 + h]; }
  float vx[4], vy[4], vz[4];
  {
    const float* vp = P.vertices + ((size_t)b * P.nQ + qc) * 24 + vh * 12;
#pragma unroll
    for (int v = 0; v < 4; ++v) { vx[v] = vp[v * 3]; vy[v] = vp[v * 3 + 1]; vz[v] = vp[v * 3 + 2]; }
  }
  const float rc = rot ? P.cos_sin[((size_t)b * P.nQ + qc) * 2] : 1.f;
  const float rs = rot ? P.cos_sin[((size_t)b * P.nQ + qc) * 2 + 1] : 0.f;

  // this wave's key range: the workgroup's range cut into 8 pieces of a multiple of 4 keys
  const int wg_beg = blockIdx.y * keys_per_wg;
  const int wg_end = min(P.nK, wg_beg + keys_per_wg);
  const int per_wave = ((keys_per_wg + (kRlThreads / kWave) - 1) / (kRlThreads / kWave) + 3) & ~3;
  const int kbeg = wg_beg + w * per_wave;
  const int kend = min(wg_end, kbeg + per_wave);
  const bool aligned = (P.nK & 3) == 0;  // rows start 16-B aligned and kbeg is a multiple of 4

  float acc[4][32];
  int cur[4];
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    cur[v] = -1;
#pragma unroll
    for (int e = 0; e < 32; ++e) acc[v][e] = 0.f;
  }
  auto flush = [&](int v) {
    float* base = smem + (size_t)cur[v] * 4;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      float* cp = base + ((c >> 2) * TT + ((c >> 1) & 1) * T + (c & 1)) * 4;
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        atomicAdd(cp + h, acc[v][c * 4 + h]);
        acc[v][c * 4 + h] = 0.f;
      }
    }
  };

  for (int kk = kbeg; kk < kend; kk += 4) {
    const int nv = min(4, kend - kk);
    f32x4 ds4[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const size_t e = (row0 + h) * P.nK + kk;
      const f32x4 s4 = load4(P.scores + e, nv, aligned);
      const f32x4 d4 = load4(P.dprob + e, nv, aligned);
      f32x4 p4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int key = kk + j;
        bool keep = true;
        if (P.drop_thresh) keep = pick4(attn_rand4(P, b, qc, key, 0), h) >= P.drop_thresh;
        const bool masked = P.mask_kind == VDETR_MASK_BOOL && j < nv &&
                            reinterpret_cast<const unsigned char*>(P.mask)[((size_t)b * P.nQ + qc) * P.nK + key];
        const ScoreGrad g = score_grad(s4[j], lse[h], keep, P.drop_scale, true, d4[j], delta[h], masked);
        p4[j] = g.p_drop;
        ds4[h][j] = qvalid ? g.ds : 0.f;
      }
      if (vh == 0 && qvalid) {
        store4(P.scores + e, p4, nv, aligned);
        store4(P.dprob + e, ds4[h], nv, aligned);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (j >= nv) break;  // wave-uniform
      const float* xp = P.xyz + ((size_t)b * P.nK + kk + j) * 3;
      const float kx = xp[0], ky = xp[1], kz = xp[2];
      const float ds[4] = {ds4[0][j], ds4[1][j], ds4[2][j], ds4[3][j]};
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        float dx = vx[v] - kx, dy = vy[v] - ky, dz = vz[v] - kz;
        if (rot) rpe_rotate(dx, dy, rc, rs);
        const AxisTap ax = rpe_axis(dx, P), ay = rpe_axis(dy, P), az = rpe_axis(dz, P);
        const int cell = (vh * 4 + v) * T3 + rpe_cell(ax, ay, az, T);
        if (cell != cur[v]) {
          if (cur[v] >= 0) flush(v);
          cur[v] = cell;
        }
        const float w00 = az.wa * ay.wa, w01 = az.wa * ay.wb, w10 = az.wb * ay.wa, w11 = az.wb * ay.wb;
        const float wgt[8] = {w00 * ax.wa, w00 * ax.wb, w01 * ax.wa, w01 * ax.wb,
                              w10 * ax.wa, w10 * ax.wb, w11 * ax.wa, w11 * ax.wb};
#pragma unroll
        for (int c = 0; c < 8; ++c)
#pragma unroll
          for (int h = 0; h < 4; ++h) acc[v][c * 4 + h] = __builtin_fmaf(wgt[c], ds[h], acc[v][c * 4 + h]);
      }
    }
  }
#pragma unroll
  for (int v = 0; v < 4; ++v)
    if (cur[v] >= 0) flush(v);
  __syncthreads();
  const int wg = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
  float* dst = P.dtable_part + (size_t)wg * table_floats;
  for (int i = tid; i < table_floats; i += kRlThreads) dst[i] = smem[i];
}

// dtable[e] += sum over workgroup copies; blockIdx.y takes a slice of the copies so that the 32 MB of partials are
// read by ~2000 blocks instead of 125
constexpr int kRedSlice = 16;
__global__ __launch_bounds__(256) void attn_bwd_table_reduce_kernel(const float* part, int nparts, int n, float* dtable) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  const int p0 = blockIdx.y * kRedSlice, p1 = min(nparts, p0 + kRedSlice);
  float s = 0.f;
  for (int p = p0; p < p1; ++p) s += part[(size_t)p * n + e];
  unsafeAtomicAdd(dtable + e, s);
}

// keep-mask dump (test hook)
__global__ __launch_bounds__(256) void attn_dropout_mask_kernel(AttnParams P, uint8_t* keep) {
  attn_load_rng(P);
  const size_t total = (size_t)P.B * P.H * P.nQ * P.nK;
  const bool perhead = P.kind == VDETR_ATTN_PER_HEAD;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const size_t row = e / P.nK;
    const int key = (int)(e - row * P.nK);
    int b, q, h;
    if (perhead) {
      q = (int)(row % P.nQ); h = (int)((row / P.nQ) % P.H); b = (int)(row / ((size_t)P.nQ * P.H));
    } else {
      h = (int)(row % P.H); q = (int)((row / P.H) % P.nQ); b = (int)(row / ((size_t)P.nQ * P.H));
    }
    keep[e] = (!P.drop_thresh || pick4(attn_rand4(P, b, q, key, h >> 2), h & 3) >= P.drop_thresh) ? 1 : 0;
  }
}

}  // namespace vdetr

using namespace vdetr;

static int bwd_variant() {
  static const int variant = [] { const char* v = getenv("VDETR_BWD_VARIANT"); return v ? atoi(v) : 4; }();
  return variant;
}
// run-length variant: grid (query groups, key splits, B); splits chosen to give >= ~256 workgroups
static void rl_geometry(const vdetr_attn_desc* d, int* qgroups, int* splits, int* keys_per_wg) {
  *qgroups = (d->nQ + kRlQueries - 1) / kRlQueries;
  int sp = 1;
  while ((long)*qgroups * d->B * sp < 256 && sp < 64 && d->nK / (sp * 2) >= 64) sp *= 2;
  int per = ((d->nK + sp - 1) / sp + 31) & ~31;  // multiple of 32: 8 waves x multiple of 4 keys
  *splits = (d->nK + per - 1) / per;
  *keys_per_wg = per;
}
static int bwd_grid(const vdetr_attn_desc* d) {
  if (bwd_variant() == 2) {
    int qg, sp, per;
    rl_geometry(d, &qg, &sp, &per);
    return qg * sp * d->B;
  }
  const long items = (long)d->B * d->nQ;
  return (int)(items < 256 ? items : 256);
}

extern "C" size_t vdetr_attn_bwd_workspace_bytes(const vdetr_attn_desc* d) {
  if (!d || !d->table) return 0;
  const size_t table_floats = (size_t)kRpeVerts * d->table_size * d->table_size * d->table_size * 4;
  return (size_t)bwd_grid(d) * table_floats * sizeof(float) + 256;
}

extern "C" int vdetr_attn_bwd_scores_f32(const vdetr_attn_desc* d, float* scores, float* dprob, const float* lse,
                                         const float* delta, float* dtable, void* workspace, size_t workspace_bytes,
                                         vdetr_stream_t stream) {
  AttnParams P;
  if (int e = attn_fill_params(d, &P, "attn_bwd_scores")) return e;
  VDETR_REQUIRE(scores && lse, "attn_bwd_scores: null pointer");
  VDETR_REQUIRE((dprob == nullptr) == (delta == nullptr), "attn_bwd_scores: dprob and delta go together");
  VDETR_REQUIRE(!dtable || (d->table && dprob), "attn_bwd_scores: dtable needs an RPE descriptor and dprob");
  P.scores = scores; P.dprob = dprob; P.lse = const_cast<float*>(lse); P.delta = delta;
  hipStream_t st = (hipStream_t)stream;
  if (!d->table) {
    const size_t total = (size_t)d->B * d->H * d->nQ * d->nK;
    const unsigned grid = (unsigned)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(attn_bwd_scores_kernel, dim3(grid), dim3(256), 0, st, P);
    return check_launch("attn_bwd_scores");
  }
  const int grid = bwd_grid(d);
  const int table_floats = kRpeVerts * P.T * P.T * P.T * 4;
  size_t lds = 16;
  if (dtable) {
    const size_t need = vdetr_attn_bwd_workspace_bytes(d);
    if (!workspace || workspace_bytes < need) {
      set_error("attn_bwd_scores: workspace %zu B < required %zu B", workspace_bytes, need);
      return VDETR_ERR_WORKSPACE;
    }
    P.dtable_part = (float*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    lds = (size_t)table_floats * sizeof(float);
  }
  const int variant = bwd_variant();
  if (variant == 4 && dtable) {
    lds += (size_t)kMmWaves * kMmStripFloats * sizeof(float);
    if (int e = set_lds(attn_bwd_scores_rpe_mm_kernel, lds, "attn_bwd_scores")) return e;
    hipLaunchKernelGGL(attn_bwd_scores_rpe_mm_kernel, dim3(grid), dim3(kMmThreads), lds, st, P);
  } else if (variant == 3 && dtable) {
    lds += (size_t)kPcWaves * kPcStripFloats * sizeof(float);
    if (int e = set_lds(attn_bwd_scores_rpe_pc_kernel, lds, "attn_bwd_scores")) return e;
    hipLaunchKernelGGL(attn_bwd_scores_rpe_pc_kernel, dim3(grid), dim3(kPcThreads), lds, st, P);
  } else if (variant == 2 && dtable) {
    int qg, sp, per;
    rl_geometry(d, &qg, &sp, &per);
    if (int e = set_lds(attn_bwd_scores_rpe_rl_kernel, lds, "attn_bwd_scores")) return e;
    hipLaunchKernelGGL(attn_bwd_scores_rpe_rl_kernel, dim3(qg, sp, d->B), dim3(kRlThreads), lds, st, P, per);
  } else if (variant == 0) {
    if (int e = set_lds(attn_bwd_scores_rpe_kernel<0>, lds, "attn_bwd_scores")) return e;
    hipLaunchKernelGGL((attn_bwd_scores_rpe_kernel<0>), dim3(grid), dim3(kBwdThreads), lds, st, P);
  } else {
    if (int e = set_lds(attn_bwd_scores_rpe_kernel<1>, lds, "attn_bwd_scores")) return e;
    hipLaunchKernelGGL((attn_bwd_scores_rpe_kernel<1>), dim3(grid), dim3(kBwdThreads), lds, st, P);
  }
  if (int e = check_launch("attn_bwd_scores_rpe")) return e;
  if (dtable) {
    hipLaunchKernelGGL(attn_bwd_table_reduce_kernel, dim3((table_floats + 255) / 256, (grid + kRedSlice - 1) / kRedSlice),
                       dim3(256), 0, st, P.dtable_part, grid, table_floats, dtable);
    return check_launch("attn_bwd_table_reduce");
  }
  return VDETR_OK;
}

extern "C" int vdetr_attn_dropout_mask_u8(const vdetr_attn_desc* d, uint8_t* keep, vdetr_stream_t stream) {
  AttnParams P;
  if (int e = attn_fill_params(d, &P, "attn_dropout_mask")) return e;
  VDETR_REQUIRE(keep, "attn_dropout_mask: null pointer");
  const size_t total = (size_t)d->B * d->H * d->nQ * d->nK;
  const unsigned grid = (unsigned)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
  hipLaunchKernelGGL(attn_dropout_mask_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, P, keep);
  return check_launch("attn_dropout_mask");
}
