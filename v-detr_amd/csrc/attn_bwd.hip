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
    P.probs_out[e] = g.p_drop;
    if (have_grad) P.ds_out[e] = g.ds;
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
          P.probs_out[e] = g.p_drop;
          if (have_grad) P.ds_out[e] = g.ds;
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
// Measured on MI355X (nQ=1024, nK=4096, B=1): every variant that adds into the LDS histogram with
// ds_add_f32 takes time proportional to its number of atomic LANE operations (~4 cycles each, CU-wide serial):
// plain per-lane atomics 6.5 ms, wave-aggregated (variant 1) 1.29 ms — no matter how cheap the aggregation itself is.
// This variant therefore removes the atomics altogether:
//   * wave w of the workgroup owns VERTEX w: it is the only writer of table i = w in the workgroup's LDS histogram,
//     (plain read/add/write instead of atomics is NOT possible: neighbouring cells' 8-corner footprints overlap, so two
//     groups of one update can hit the same bin);
//   * the wave-level aggregation is a small matrix product on the matrix cores,
//         G[group][value] = sum over the 64 lanes  M[group][lane] * V[lane][value]
//     with M the 0/1 membership of a lane (pair) in a group (= distinct lookup cell among the wave's 64 pairs, <= 16
//     per round, 7.9 on average) and V the lane's products weight(corner) * dS(head): 16 v_mfma_f32_16x16x4_f32 per
//     16 values for ALL groups at once — exact fp32 (0/1 factors, fixed summation order: run-to-run deterministic).
//       lane l as A operand: row = group l&15, k-slot l>>4 -> membership of pair 4s + (l>>4)   (cells via an LDS strip)
//       lane l as B operand: k-slot l>>4, column = value l&15 -> V[pair 4s + (l>>4)][l&15]      (values via an LDS strip)
//       result: lane l holds value l&15 of groups 4*(l>>4)+r.
//   * every wave recomputes the cheap element-wise softmax backward of the 64 pairs it looks at (8x redundant, ~100
//     VALU ops against ~600 + 32 MFMA of vertex work); wave 0 writes P~ / dS.  Because the waves of a workgroup
//     drift apart, P~ / dS go to SEPARATE output tensors (in-place would let a slow wave read overwritten scores).
constexpr int kMmThreads = 512;
constexpr int kMmWaves = kMmThreads / kWave;
constexpr int kMmStripFloats = kWave + kWave * 16;  // cell ids + 64 x 16 values
static_assert(kMmWaves == kRpeVerts, "one wave per vertex table");

// |dP~| maximum of the launch (bit pattern of a non-negative float, atomicMax on unsigned)
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, size_t n, unsigned* __restrict__ out) {
  float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f;
  const size_t n4 = n >> 2;
  const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  auto amax4 = [](const f32x4& v) { return fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))); };
  for (; i + 3 * stride < n4; i += 4 * stride) {  // four independent 16-B loads in flight per lane
    const f32x4 a = x4[i], b = x4[i + stride], c = x4[i + 2 * stride], d = x4[i + 3 * stride];
    m0 = fmaxf(m0, amax4(a)); m1 = fmaxf(m1, amax4(b)); m2 = fmaxf(m2, amax4(c)); m3 = fmaxf(m3, amax4(d));
  }
  for (; i < n4; i += stride) m0 = fmaxf(m0, amax4(x4[i]));
  for (size_t t = (n4 << 2) + (size_t)blockIdx.x * 256 + threadIdx.x; t < n; t += stride) m0 = fmaxf(m0, fabsf(x[t]));
  const float m = wave_allmax_f32(fmaxf(fmaxf(m0, m1), fmaxf(m2, m3)));
  if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));
}

// FIXED: the LDS histogram is int32 fixed point.  Measured on MI355X (tools/kernel_bench.py --lds): ds_add_f32 runs
// at 0.37 lane-updates/clk/CU regardless of address conflicts, ds_add_u32 at 2.3 — the float atomics alone were
// ~0.5 ms of this kernel.  The scale is exact-safe: per bin, sum |contribution| <= sum over the workgroup's queries of
// sum_k P(q,k) * |dP - delta| <= queries_per_wg * 2 * drop_scale * max|dP~|  (weights <= 1, softmax rows sum to 1),
// so with S = 2^floor(log2(2^30 / bound)) no partial sum can overflow, and the resolution (bound * 2^-30) is ~1e-5
// of a typical group sum.  Integer adds also make the histogram independent of the order of the updates.
template <bool FIXED>
__global__ __launch_bounds__(kMmThreads) void attn_bwd_scores_rpe_mm_kernel(AttnParams P) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [dTable copy 8*T^3*4][8 strips]
  attn_load_rng(P);
  float fix_scale = 1.f, fix_inv = 1.f;
  if (FIXED) {
    const float dmax = __uint_as_float(*P.absmax);
    const int per_wg = (P.B * P.nQ + (int)gridDim.x - 1) / (int)gridDim.x;
    const float bound = 2.f * P.drop_scale * dmax * (float)per_wg;
    if (bound > 0.f && bound < INFINITY) {
      const int e = 30 - (int)ceilf(__log2f(bound) + 1e-3f);  // 2^e * bound <= 2^30
      fix_scale = ldexpf(1.f, e);
      fix_inv = ldexpf(1.f, -e);
    }
  }
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);  // = vertex index
  const int T = P.T, TT = T * T, T3 = TT * T;
  const int table_floats = kRpeVerts * T3 * 4;
  for (int i = tid; i < table_floats; i += kMmThreads) smem[i] = 0.f;
  __syncthreads();
  // wave-private strip, accessed as int throughout (values are bit-cast) so that the scoreboard and the value
  // tile, which share memory, are never seen through two different types by the compiler's alias analysis
  int* cellbuf = reinterpret_cast<int*>(smem + table_floats + w * kMmStripFloats);  // 64 ints
  int* vbuf = cellbuf + kWave;                                                       // 64 x 16 values
  int* scoreboard = vbuf;                                                            // T^3 <= 1024 ints, aliases vbuf
  float* mytab = smem + (size_t)w * T3 * 4;
  const bool rot = P.cos_sin != nullptr;
  const int items = P.B * P.nQ;
  const int nchunks = (P.nK + kWave - 1) / kWave;
  const int kk = lane >> 4, c15 = lane & 15;
  int off[2];  // bin offsets of this lane's output column for the two 16-value tiles (tile jt = corners 4jt..4jt+3)
#pragma unroll
  for (int jt = 0; jt < 2; ++jt) {
    const int corner = jt * 4 + (c15 >> 2);
    off[jt] = (((corner >> 2) & 1) * TT + ((corner >> 1) & 1) * T + (corner & 1)) * 4 + (c15 & 3);
  }
  const int wr_sw = lane >> 1;  // 16-B blocks of a strip row are rotated by lane>>1: conflict-free b128 writes AND b32 reads

  // operands of one chunk (64 consecutive keys, one per lane); fetched one chunk ahead
  struct ChunkOps {
    float s[4], d[4], kx, ky, kz;
    unsigned char masked;
  };
  auto fetch = [&](int b, size_t row0, int q, int chunk, ChunkOps& o) {
    const int key = min(chunk * kWave + lane, P.nK - 1);
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const size_t e = (row0 + h) * P.nK + key;
      o.s[h] = P.scores[e];
      o.d[h] = P.dprob[e];
    }
    const float* xp = P.xyz + ((size_t)b * P.nK + key) * 3;
    o.kx = xp[0]; o.ky = xp[1]; o.kz = xp[2];
    o.masked = P.mask_kind == VDETR_MASK_BOOL
                   ? reinterpret_cast<const unsigned char*>(P.mask)[((size_t)b * P.nQ + q) * P.nK + key] : 0;
  };

  for (int item = blockIdx.x; item < items; item += gridDim.x) {
    const int b = item / P.nQ, q = item - b * P.nQ;
    const size_t row0 = ((size_t)b * P.nQ + q) * 4;
    float lse[4], delta[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) { lse[h] = P.lse[row0 + h]; delta[h] = P.delta[row0 + h]; }
    const float* vp = P.vertices + ((size_t)b * P.nQ + q) * 24 + w * 3;
    const float vx = vp[0], vy = vp[1], vz = vp[2];
    const float rc = rot ? P.cos_sin[((size_t)b * P.nQ + q) * 2] : 1.f;
    const float rs = rot ? P.cos_sin[((size_t)b * P.nQ + q) * 2 + 1] : 0.f;
    ChunkOps ops, nxt;
    fetch(b, row0, q, 0, ops);

    for (int chunk = 0; chunk < nchunks; ++chunk) {
      if (chunk + 1 < nchunks) fetch(b, row0, q, chunk + 1, nxt);
      // ---- element-wise softmax backward of this lane's pair (recomputed by every wave; wave 0 stores) ----------
      const int key = chunk * kWave + lane;
      const bool valid = key < P.nK;
      float ds[4];
      {
        uint4 rnd = make_uint4(0xFFFFu, 0xFFFFu, 0xFFFFu, 0xFFFFu);
        if (P.drop_thresh) rnd = attn_rand4(P, b, q, key, 0);
#pragma unroll
        for (int h = 0; h < 4; ++h) {
          const bool keep = pick4(rnd, h) >= P.drop_thresh;
          const ScoreGrad g = score_grad(ops.s[h], lse[h], keep, P.drop_scale, true, ops.d[h], delta[h], ops.masked != 0);
          if (w == 0 && valid) {
            const size_t e = (row0 + h) * P.nK + key;
            P.probs_out[e] = g.p_drop;
            P.ds_out[e] = g.ds;
          }
          ds[h] = valid ? g.ds : 0.f;
        }
      }
      // ---- lookup geometry of (pair, vertex w) ---------------------------------------------------------------------
      float dx = vx - ops.kx, dy = vy - ops.ky, dz = vz - ops.kz;
      if (rot) rpe_rotate(dx, dy, rc, rs);
      const AxisTap ax = rpe_axis(dx, P), ay = rpe_axis(dy, P), az = rpe_axis(dz, P);
      const int cell = rpe_cell(ax, ay, az, T);
      const float w00 = az.wa * ay.wa, w01 = az.wa * ay.wb, w10 = az.wb * ay.wa, w11 = az.wb * ay.wb;
      const float wgt[8] = {w00 * ax.wa, w00 * ax.wb, w01 * ax.wa, w01 * ax.wb,
                            w10 * ax.wa, w10 * ax.wb, w11 * ax.wa, w11 * ax.wb};
      // ---- groups = distinct cells among the 64 pairs, without a serial loop: every lane posts its id on a
      // scoreboard slot indexed by its cell; whoever is read back is the group's leader, and a group's index is the
      // rank of its leader among the leaders.
      __builtin_amdgcn_wave_barrier();
      scoreboard[cell] = lane;
      __builtin_amdgcn_wave_barrier();
      const int leader = scoreboard[cell];
      const unsigned long long lmask = __ballot(leader == lane);
      const int ngroups = __popcll(lmask);
      const int gidx = __popcll(lmask & ((1ull << leader) - 1ull));
      // group index of all 64 pairs, laid out so that k-slot kk reads pairs kk, 4+kk, 8+kk, ... as 4 x 16 bytes
      cellbuf[(lane & 3) * 16 + (lane >> 2)] = gidx;
      __builtin_amdgcn_wave_barrier();
      int cr[16];
#pragma unroll
      for (int t4 = 0; t4 < 4; ++t4) {
        const int4 v4 = *reinterpret_cast<const int4*>(cellbuf + kk * 16 + t4 * 4);
        cr[t4 * 4] = v4.x; cr[t4 * 4 + 1] = v4.y; cr[t4 * 4 + 2] = v4.z; cr[t4 * 4 + 3] = v4.w;
      }
      __builtin_amdgcn_wave_barrier();
      if (leader == lane) cellbuf[gidx] = cell;  // the strip now holds cell-of-group
      __builtin_amdgcn_wave_barrier();

      for (int g0 = 0; g0 < ngroups; g0 += 16) {  // 16 groups per round (one round in 95 % of the chunks)
        float am[16];
#pragma unroll
        for (int s2 = 0; s2 < 16; ++s2) am[s2] = cr[s2] == g0 + c15 ? 1.f : 0.f;
        int gc[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int g = g0 + 4 * kk + r;
          gc[r] = g < ngroups ? cellbuf[g] : -1;
        }
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int cc = 0; cc < 4; ++cc) {
            const float wc = wgt[jt * 4 + cc];
            *reinterpret_cast<int4*>(vbuf + lane * 16 + ((cc + wr_sw) & 3) * 4) =
                make_int4(__float_as_int(wc * ds[0]), __float_as_int(wc * ds[1]), __float_as_int(wc * ds[2]),
                          __float_as_int(wc * ds[3]));
          }
          __builtin_amdgcn_wave_barrier();
          float bv[16];
#pragma unroll
          for (int s2 = 0; s2 < 16; ++s2) {
            const int p = 4 * s2 + kk;
            bv[s2] = __int_as_float(vbuf[p * 16 + (((c15 >> 2) + (p >> 1)) & 3) * 4 + (c15 & 3)]);
          }
          f32x4 acc[4];  // four independent chains: the MFMAs are issue-, not latency-bound
#pragma unroll
          for (int ch = 0; ch < 4; ++ch) acc[ch] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int s2 = 0; s2 < 16; ++s2)
            acc[s2 & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(am[s2], bv[s2], acc[s2 & 3], 0, 0, 0);
          // wave w is the only writer of table w in this workgroup; the (group, value) bins of one update are distinct
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (gc[r] >= 0) {
              float* bin = mytab + (size_t)gc[r] * 4 + off[jt];
              const float tot = (acc[0][r] + acc[1][r]) + (acc[2][r] + acc[3][r]);
              if (FIXED) atomicAdd(reinterpret_cast<int*>(bin), __float2int_rn(tot * fix_scale));
              else atomicAdd(bin, tot);
            }
        }
      }
      ops = nxt;
    }
  }
  __syncthreads();
  float* dst = P.dtable_part + (size_t)blockIdx.x * table_floats;
  for (int i = tid; i < table_floats; i += kMmThreads)
    dst[i] = FIXED ? (float)reinterpret_cast<const int*>(smem)[i] * fix_inv : smem[i];
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

// LDS atomic-rate probe (tools/kernel_bench.py --lds): mode 0 ds_add_f32, 1 ds_add_u32, 2 plain read-add-write,
// 3 ds_add_f32 with all lanes on 8 addresses; 512-thread workgroups, `iters` updates per lane to scattered bins
__global__ __launch_bounds__(512) void lds_atomic_probe_kernel(int mode, int iters, float* sink) {
  __shared__ float tab[32768];
  for (int i = threadIdx.x; i < 32768; i += 512) tab[i] = 0.f;
  __syncthreads();
  unsigned a = threadIdx.x * 2654435761u;
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    a = a * 1664525u + 1013904223u;
    const unsigned slot = mode == 3 ? ((a >> 10) & 7u) * 4u : ((a >> 10) & 8191u) * 4u + (threadIdx.x & 3);
    if (mode == 0 || mode == 3) atomicAdd(&tab[slot], 1.0f);
    else if (mode == 1) atomicAdd(reinterpret_cast<unsigned*>(&tab[slot]), 1u);
    else tab[slot] += 1.0f;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 32768; i += 512) acc += tab[i];
  if (acc == -1.f) sink[0] = acc;
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
  // 0: per-lane ds_add_f32, 1: wave-aggregated + ds_add_f32, 4: matrix-core aggregation, no atomics (default)
  static const int variant = [] { const char* v = getenv("VDETR_BWD_VARIANT"); return v ? atoi(v) : 4; }();
  return variant;
}
static int bwd_grid(const vdetr_attn_desc* d) {
  const long items = (long)d->B * d->nQ;
  return (int)(items < 256 ? items : 256);
}

extern "C" size_t vdetr_attn_bwd_workspace_bytes(const vdetr_attn_desc* d) {
  if (!d || !d->table) return 0;
  const size_t table_floats = (size_t)kRpeVerts * d->table_size * d->table_size * d->table_size * 4;
  return (size_t)bwd_grid(d) * table_floats * sizeof(float) + 512;  // partial tables + |dP~| max scalar + alignment
}

extern "C" int vdetr_attn_bwd_scores_f32(const vdetr_attn_desc* d, const float* scores, const float* dprob,
                                         const float* lse, const float* delta, float* probs_out, float* ds_out,
                                         float* dtable, void* workspace, size_t workspace_bytes,
                                         vdetr_stream_t stream) {
  AttnParams P;
  if (int e = attn_fill_params(d, &P, "attn_bwd_scores")) return e;
  VDETR_REQUIRE(scores && lse && probs_out, "attn_bwd_scores: null pointer");
  VDETR_REQUIRE((dprob == nullptr) == (delta == nullptr) && (dprob == nullptr) == (ds_out == nullptr),
                "attn_bwd_scores: dprob, delta and ds_out go together");
  VDETR_REQUIRE(!dtable || (d->table && dprob), "attn_bwd_scores: dtable needs an RPE descriptor and dprob");
  VDETR_REQUIRE(!dtable || (probs_out != scores && ds_out != dprob),
                "attn_bwd_scores: with a table gradient the outputs must not alias the inputs "
                "(several waves re-read the scores)");
  P.scores = const_cast<float*>(scores); P.dprob = const_cast<float*>(dprob);
  P.lse = const_cast<float*>(lse); P.delta = delta;
  P.probs_out = probs_out; P.ds_out = ds_out;
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
  if (dtable && variant != 0 && variant != 1 && P.T * P.T * P.T <= kWave * 16) {
    lds += (size_t)kMmWaves * kMmStripFloats * sizeof(float);
    if (variant == 5) {  // float LDS atomics (A/B reference for the fixed-point histogram)
      if (int e = set_lds(attn_bwd_scores_rpe_mm_kernel<false>, lds, "attn_bwd_scores")) return e;
      hipLaunchKernelGGL(attn_bwd_scores_rpe_mm_kernel<false>, dim3(grid), dim3(kMmThreads), lds, st, P);
    } else {
      // max |dP~| of the launch -> scale of the fixed-point histogram (scalar lives at the end of the workspace)
      unsigned* absmax = reinterpret_cast<unsigned*>(P.dtable_part + (size_t)grid * table_floats);
      P.absmax = absmax;
      if (hipMemsetAsync(absmax, 0, sizeof(unsigned), st) != hipSuccess) {
        set_error("attn_bwd_scores: memset failed");
        return VDETR_ERR_LAUNCH;
      }
      const size_t total = (size_t)d->B * d->H * d->nQ * d->nK;
      hipLaunchKernelGGL(absmax_kernel, dim3(2048), dim3(256), 0, st, dprob, total, absmax);
      if (int e = set_lds(attn_bwd_scores_rpe_mm_kernel<true>, lds, "attn_bwd_scores")) return e;
      hipLaunchKernelGGL(attn_bwd_scores_rpe_mm_kernel<true>, dim3(grid), dim3(kMmThreads), lds, st, P);
    }
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

extern "C" int vdetr_selftest_lds_atomics(int mode, int iters, float* sink, vdetr_stream_t stream) {
  hipLaunchKernelGGL(lds_atomic_probe_kernel, dim3(256), dim3(512), 0, (hipStream_t)stream, mode, iters, sink);
  return check_launch("lds_atomic_probe");
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
