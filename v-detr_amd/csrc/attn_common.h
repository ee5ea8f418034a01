// attn_common.h — device helpers shared by the attention forward / backward kernels.
#pragma once
#include "common.h"
#include "wave.h"

namespace vdetr {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kDh = 64;          // head dim (reference: dec_dim 256 / dec_nhead 4, main.py:73,76)
constexpr int kRpeHeads = 4;     // the RPE lane layout carries the heads of one (query,key) pair in 4 registers
constexpr int kRpeVerts = 8;     // 8 box vertices (vdetr_transformer.py:710)
constexpr float kNegBig = -1e30f;
// Cache policy of the score-sized streams (S, dS: 67 MB per layer and direction at C2, twice the 32 MB of L2): `nt` keeps them
// from evicting what the small kernels between the attention kernels re-use (weights, activations).  -DVDETR_STREAM_NT=0: off.
#ifndef VDETR_STREAM_NT
#define VDETR_STREAM_NT 1
#endif
constexpr int kStreamAux = VDETR_STREAM_NT ? 2 : 0;  // raw buffer aux bits: 2 = nt (slc)

// Table-gradient kernels, dynamic query distribution: the most queries one persistent workgroup may take, given the even share
// `per_wg`.  The int32 histogram's fixed-point scale is sized for this many queries (a power of two below 2^30 / bound), so the
// cap trades resolution against slack for uneven progress: 1.5x the share (it was 2x until round 4: 16 queries at 1024 queries
// on 128 workgroups per z-half; 1.5x keeps those 16 at the 96 workgroups per half a side-stream launch uses and gives the
// default grid one more bit).  Workgroups x cap >= queries, so every query is taken whatever the workgroups' pace.
__host__ __device__ __forceinline__ constexpr int bwd_query_cap(int per_wg) { return per_wg + (per_wg + 1) / 2; }
constexpr float kLog2e = 1.4426950408889634f;

struct AttnParams {
  int kind, B, H, nQ, nK;
  float scale;
  const float* q;
  const float* k;
  const float* v;
  int k_stride, v_stride;  // floats between consecutive key rows
  int box_path;            // forward: the axis-aligned-box instantiation is launched too (see attn_fwd.hip)
  const unsigned* bwd_aux; // {max |dO row|^2, max |V row|^2, query counter half 0, half 1} or NULL
  float* out;
  float* lse;
  float* scores;
  // rpe
  const float* table;
  int T;             // table edge
  float log_scale;   // 512
  float pix_mul;     // inv_log_norm * T / 2
  float pix_add;     // (T - 1) / 2
  const float* vertices;
  const float* xyz;
  const float* cos_sin;
  const void* mask;
  int mask_kind;
  // dropout
  unsigned drop_thresh;  // keep iff rand16 >= thresh (thresh = p * 65536)
  float drop_scale;      // 65536 / (65536 - thresh): unbiased for the quantised rate
  unsigned seed_lo, seed_hi, off_lo, off_hi;
  const unsigned long long* rng;  // optional device {seed, offset}
  // key split (forward)
  int ksplit, tiles_per_split;
  float* part_o;    // [ksplit][rows][64]
  float* part_lse;  // [ksplit][rows]
  // backward
  float* dprob;
  const float* delta;
  float* probs_out;
  float* ds_out;
  float* dtable_part;  // [gridDim.x][8*T^3*H]
  int ds_given;        // table-gradient kernels: `dprob` already holds dS (attn_bwd_kv.hip wrote it); nothing else is read or stored
};

// ---- dropout random numbers: counter-based (stateless), so forward and backward regenerate the same keep-mask
// from (seed, offset, b, head group, q, key).  A chain of murmur3 finalisers (fmix32: a bijection of 32 bits with
// full avalanche) keyed by the counter words; the (seed, offset, b, q) prefix is lane-invariant in the kernels and
// is hoisted out of the key loop, leaving two fmix32 (~16 VALU ops) per (query, key) pair for its 4 heads —
// a Philox4x32-10 here cost ~100 ops, a seventh of the whole forward tile step.
__device__ __forceinline__ unsigned fmix32(unsigned x) {
  x ^= x >> 16;
  x *= 0x85EBCA6Bu;
  x ^= x >> 13;
  x *= 0xC2B2AE35u;
  x ^= x >> 16;
  return x;
}
// kernels call this first: fold the device-resident {seed, offset} pair into the by-value one
__device__ __forceinline__ void attn_load_rng(AttnParams& P) {
  if (P.rng) {
    // device state is COMBINED with the by-value pair: seed ^= state.seed, offset += state.offset, so that
    // many attention modules can share one per-step device state and still draw independent masks
    const unsigned long long s = P.rng[0] ^ (((unsigned long long)P.seed_hi << 32) | P.seed_lo);
    const unsigned long long o = P.rng[1] + (((unsigned long long)P.off_hi << 32) | P.off_lo);
    P.seed_lo = (unsigned)s; P.seed_hi = (unsigned)(s >> 32);
    P.off_lo = (unsigned)o; P.off_hi = (unsigned)(o >> 32);
  }
}
// Four 16-bit uniform values of attention element (b, q, key) for heads 4*hgroup .. 4*hgroup+3.
__device__ __forceinline__ uint4 attn_rand4(const AttnParams& P, int b, int q, int key, int hgroup) {
  unsigned x = fmix32(((unsigned)q * 0x9E3779B1u + P.off_lo) ^ P.seed_lo);
  x = fmix32(x ^ (((unsigned)b * 64u + (unsigned)hgroup) * 0x27D4EB2Fu + P.off_hi) ^ P.seed_hi);
  x = fmix32(x ^ ((unsigned)key * 0x165667B1u));
  const unsigned y = fmix32(x + 0x9E3779B9u);
  return make_uint4(x & 0xFFFFu, x >> 16, y & 0xFFFFu, y >> 16);
}
__device__ __forceinline__ unsigned pick4(const uint4& r, int i) {
  return i == 0 ? r.x : (i == 1 ? r.y : (i == 2 ? r.z : r.w));
}

// ---- element-wise softmax backward: P~ and dS for one head of one (query, key) pair (attn_bwd*.hip) ------------------
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

// ---- 3DV-RPE lookup geometry ----------------------------------------------------------------------
// Reference (vdetr_transformer.py:711-731 + F.grid_sample, bilinear/zeros/align_corners=False):
//   g   = sign(d) * log2(|d|*log_scale + 1) / log2(8) / max_value
//   pix = ((g + 1) * T - 1) / 2
//   trilinear over floor(pix), floor(pix)+1 with out-of-range corners contributing 0.
// Per axis this is the hat function sum_c T[c] * max(0, 1 - |pix - c|): with base = clamp(floor(pix),0,T-2)
// the two cells base, base+1 carry wa = sat(1-|pix-base|), wb = sat(1-|pix-base-1|) and every
// zero-padding case (pix<0, pix>T-1, far outside) falls out of the saturation — no bounds branches.
struct AxisTap {
  int base;
  float wa, wb;
};
// v_med3_f32 clamps: __saturatef / fminf(fmaxf()) expand to compare+select pairs (4 instructions per weight, measured
// 192 of the forward kernel's ~950 VALU instructions per 64 pairs)
__device__ __forceinline__ float sat01(float x) { return __builtin_amdgcn_fmed3f(x, 0.f, 1.f); }
__device__ __forceinline__ AxisTap rpe_axis(float d, const AttnParams& P) {
  const float L = __log2f(__builtin_fmaf(fabsf(d), P.log_scale, 1.0f));
  const float pix = __builtin_fmaf(copysignf(L, d), P.pix_mul, P.pix_add);
  const float bf = __builtin_amdgcn_fmed3f(floorf(pix), 0.f, (float)(P.T - 2));
  const float t = pix - bf;
  AxisTap a;
  a.base = (int)bf;
  a.wa = sat01(1.f - fabsf(t));
  a.wb = sat01(1.f - fabsf(t - 1.f));
  return a;
}

// x -> LAST table axis, y -> middle, z -> FIRST (grid_sample: x=W, y=H, z=D)
__device__ __forceinline__ int rpe_cell(const AxisTap& ax, const AxisTap& ay, const AxisTap& az, int T) {
  return (az.base * T + ay.base) * T + ax.base;
}

// rotation used by angle_type == "object_coords" (vdetr_transformer.py:712-720 reduces to a yaw
// rotation of (dx,dy): x' = dx*c - dy*s, y' = dx*s + dy*c, z' = dz)
__device__ __forceinline__ void rpe_rotate(float& dx, float& dy, float c, float s) {
  const float nx = dx * c - dy * s;
  const float ny = dx * s + dy * c;
  dx = nx;
  dy = ny;
}

// bias of one (query, key) pair for the 4 heads; tab = LDS image [8][T^3] of float4 (heads)
__device__ __forceinline__ void rpe_pair_bias(const AttnParams& P, const f32x4* tab, const float (&vx)[8],
                                              const float (&vy)[8], const float (&vz)[8], float kx, float ky,
                                              float kz, bool rot, float rc, float rs, float (&acc)[4]) {
  const int T = P.T, TT = T * T, T3 = TT * T;
#pragma unroll
  for (int i = 0; i < kRpeVerts; ++i) {
    float dx = vx[i] - kx, dy = vy[i] - ky, dz = vz[i] - kz;
    if (rot) rpe_rotate(dx, dy, rc, rs);
    const AxisTap ax = rpe_axis(dx, P), ay = rpe_axis(dy, P), az = rpe_axis(dz, P);
    const f32x4* t = tab + i * T3 + rpe_cell(ax, ay, az, T);
    const float w00 = az.wa * ay.wa, w01 = az.wa * ay.wb, w10 = az.wb * ay.wa, w11 = az.wb * ay.wb;
    const f32x4 c000 = t[0], c001 = t[1], c010 = t[T], c011 = t[T + 1];
    const f32x4 c100 = t[TT], c101 = t[TT + 1], c110 = t[TT + T], c111 = t[TT + T + 1];
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2 axw = {ax.wa, ax.wb};  // packed fp32: two corner weights per v_pk_mul_f32
    const f32x2 p0 = axw * w00, p1 = axw * w01, p2 = axw * w10, p3 = axw * w11;
    const float w000 = p0[0], w001 = p0[1], w010 = p1[0], w011 = p1[1], w100 = p2[0], w101 = p2[1], w110 = p3[0], w111 = p3[1];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      float s = acc[h];
      s = __builtin_fmaf(w000, c000[h], s);
      s = __builtin_fmaf(w001, c001[h], s);
      s = __builtin_fmaf(w010, c010[h], s);
      s = __builtin_fmaf(w011, c011[h], s);
      s = __builtin_fmaf(w100, c100[h], s);
      s = __builtin_fmaf(w101, c101[h], s);
      s = __builtin_fmaf(w110, c110[h], s);
      s = __builtin_fmaf(w111, c111[h], s);
      acc[h] = s;
    }
  }
}

// ---- axis-aligned boxes: 6 axis taps per pair instead of 24 ---------------------------------------------------------
// The eight RPE vertices are the corners of a box (box_util.py:338-346 sign pattern, converted to the lidar frame by
// :98-102).  Without rotation (ScanNet: one angle bin) their coordinates take only two values per axis, so the per-axis
// tap (log2, floor, hat weights: ~11 VALU instructions) is the same for four vertices at a time.  Vertex i uses
//   x: value (i >> 1) & 1,   y: value 1 for i & 3 in {1, 2},   z: value i >> 2
// (checked bit-wise against the actual vertices by rpe_box_pattern; anything else takes the general path).  Taps, weight
// products and the accumulation order are those of rpe_pair_bias: the result is bit-identical.
__device__ __forceinline__ constexpr int rpe_box_xi(int i) { return (i >> 1) & 1; }
__device__ __forceinline__ constexpr int rpe_box_yi(int i) { return ((i & 3) == 1 || (i & 3) == 2) ? 1 : 0; }
__device__ __forceinline__ constexpr int rpe_box_zi(int i) { return i >> 2; }
__device__ __forceinline__ bool rpe_box_pattern(const float (&vx)[8], const float (&vy)[8], const float (&vz)[8]) {
  bool ok = true;
#pragma unroll
  for (int i = 0; i < 8; ++i)
    ok = ok && vx[i] == vx[rpe_box_xi(i) ? 2 : 0] && vy[i] == vy[rpe_box_yi(i) ? 1 : 0] && vz[i] == vz[rpe_box_zi(i) ? 4 : 0];
  return ok;
}
// the two offsets per axis given directly (dx[a] = x value a of the box minus the key, in the frame the table is looked up in)
__device__ __forceinline__ void rpe_pair_bias_box_d(const AttnParams& P, const f32x4* tab, const float (&dx)[2],
                                                    const float (&dy)[2], const float (&dz)[2], float (&acc)[4]) {
  const int T = P.T, TT = T * T, T3 = TT * T;
  const AxisTap ax[2] = {rpe_axis(dx[0], P), rpe_axis(dx[1], P)};
  const AxisTap ay[2] = {rpe_axis(dy[0], P), rpe_axis(dy[1], P)};
  const AxisTap az[2] = {rpe_axis(dz[0], P), rpe_axis(dz[1], P)};
  float w00[2][2], w01[2][2], w10[2][2], w11[2][2];
  int zy[2][2];
#pragma unroll
  for (int zi = 0; zi < 2; ++zi)
#pragma unroll
    for (int yi = 0; yi < 2; ++yi) {
      w00[zi][yi] = az[zi].wa * ay[yi].wa; w01[zi][yi] = az[zi].wa * ay[yi].wb;
      w10[zi][yi] = az[zi].wb * ay[yi].wa; w11[zi][yi] = az[zi].wb * ay[yi].wb;
      zy[zi][yi] = (az[zi].base * T + ay[yi].base) * T;
    }
#pragma unroll
  for (int i = 0; i < kRpeVerts; ++i) {
    const int xi = rpe_box_xi(i), yi = rpe_box_yi(i), zi = rpe_box_zi(i);
    const f32x4* t = tab + i * T3 + zy[zi][yi] + ax[xi].base;
    const f32x4 c000 = t[0], c001 = t[1], c010 = t[T], c011 = t[T + 1];
    const f32x4 c100 = t[TT], c101 = t[TT + 1], c110 = t[TT + T], c111 = t[TT + T + 1];
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2 axw = {ax[xi].wa, ax[xi].wb};  // packed fp32: two corner weights per v_pk_mul_f32
    // (the four scalar factors pass through an empty asm: the compiler then cannot know that one of them is the HIGH half of a
    // register pair it built, which is what made it emit the second-source high-broadcast form of v_pk_mul_f32 that the build
    // check refuses, build.py; as opaque 32-bit registers they are read with op_sel_hi, the low-half broadcast)
    float s00 = w00[zi][yi], s01 = w01[zi][yi], s10 = w10[zi][yi], s11 = w11[zi][yi];
    asm volatile("" : "+v"(s00), "+v"(s01), "+v"(s10), "+v"(s11));
    const f32x2 p0 = axw * s00, p1 = axw * s01, p2 = axw * s10, p3 = axw * s11;
    const float w000 = p0[0], w001 = p0[1], w010 = p1[0], w011 = p1[1], w100 = p2[0], w101 = p2[1], w110 = p3[0], w111 = p3[1];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      float s = acc[h];
      s = __builtin_fmaf(w000, c000[h], s);
      s = __builtin_fmaf(w001, c001[h], s);
      s = __builtin_fmaf(w010, c010[h], s);
      s = __builtin_fmaf(w011, c011[h], s);
      s = __builtin_fmaf(w100, c100[h], s);
      s = __builtin_fmaf(w101, c101[h], s);
      s = __builtin_fmaf(w110, c110[h], s);
      s = __builtin_fmaf(w111, c111[h], s);
      acc[h] = s;
    }
  }
}

__device__ __forceinline__ void rpe_pair_bias_box(const AttnParams& P, const f32x4* tab, const float (&X)[2],
                                                  const float (&Y)[2], const float (&Z)[2], float kx, float ky, float kz,
                                                  float (&acc)[4]) {
  const float dx[2] = {X[0] - kx, X[1] - kx}, dy[2] = {Y[0] - ky, Y[1] - ky}, dz[2] = {Z[0] - kz, Z[1] - kz};
  rpe_pair_bias_box_d(P, tab, dx, dy, dz, acc);
}
// angle_type "object_coords": in the frame the offsets are turned into (rpe_rotate by the query's angle) the corners of a ROTATED
// box are an axis-aligned box: R (P_i - P_0) = (xi EX, yi EY, zi EZ) with the edges of vertices 3, 1 and 4 — the test of
// attn_delta_body (a few ulps of the coordinates: the corners come out of fp32 arithmetic).  The box body then looks up at
// R (P_0 - X) + (xi EX, yi EY, zi EZ): one rotation per pair instead of eight, 6 axis taps instead of 24.
__device__ __forceinline__ bool rpe_box_pattern_rot(const float (&vx)[8], const float (&vy)[8], const float (&vz)[8], float rc,
                                                    float rs, float& EX, float& EY, float& EZ) {
  float ex[8], ey[8], ez[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    ex[i] = vx[i] - vx[0]; ey[i] = vy[i] - vy[0]; ez[i] = vz[i] - vz[0];
    rpe_rotate(ex[i], ey[i], rc, rs);
  }
  EX = ex[3]; EY = ey[1]; EZ = ez[4];
  const float tol = 1e-5f * (1.f + fabsf(vx[0]) + fabsf(vy[0]) + fabsf(vz[0]));
  bool ok = true;
#pragma unroll
  for (int i = 0; i < 8; ++i)
    ok = ok && fabsf(ex[i] - (rpe_box_xi(i) ? EX : 0.f)) <= tol && fabsf(ey[i] - (rpe_box_yi(i) ? EY : 0.f)) <= tol &&
         fabsf(ez[i] - (rpe_box_zi(i) ? EZ : 0.f)) <= tol;
  return ok;
}

// cooperative copy of the [8][T^3][4] table into LDS: all of a thread's loads are issued before its first store (a plain
// copy loop waits for every load: 16 dependent round trips ~ 7 us per workgroup, measured through the key-split sweep)
__device__ __forceinline__ void rpe_stage_table(const AttnParams& P, f32x4* tab, int tid, int nthreads) {
  const int cells = kRpeVerts * P.T * P.T * P.T;
  const f32x4* src = reinterpret_cast<const f32x4*>(P.table);
  constexpr int kBatch = 16;
  for (int c0 = tid; c0 < cells; c0 += nthreads * kBatch) {
    f32x4 v[kBatch];
#pragma unroll
    for (int u = 0; u < kBatch; ++u) {
      const int c = c0 + u * nthreads;
      v[u] = c < cells ? src[c] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < kBatch; ++u) {
      const int c = c0 + u * nthreads;
      if (c < cells) tab[c] = v[u];
    }
  }
}

// ---- delta[row] = sum_d dO[b,q,h,d] * O[b,q,h,d] (the softmax-backward term): one wave per (b, q), lane l holds element
// h*64 + l of the heads.  With `aux` (zeroed by the caller) the launch also leaves max_row |dO row|^2 in aux[0] and, from extra
// workgroups that walk the key rows, max_key |V row|^2 in aux[1] (bit patterns of non-negative floats under atomicMax),
// counts the queries whose RPE vertices are not an axis-aligned box in aux[4] and sets aux[5].  256-thread workgroups; `block`
// in [0, qblocks + vblocks).
constexpr int kDeltaKeys = 16;  // keys per wave in the |V row| pass
struct DeltaArgs {
  const float* dout;
  const float* out;
  float* delta;
  int B, nQ, H, perhead;
  unsigned* aux;
  const float* v;
  int nK, v_stride, qblocks, vblocks;
  const float* vertices;
  const float* cos_sin;
};
inline void attn_delta_args(const vdetr_attn_desc* d, const float* dout, const float* out, const float* v, float* delta, DeltaArgs* A) {
  A->dout = dout; A->out = out; A->delta = delta;
  A->B = d->B; A->nQ = d->nQ; A->H = d->H; A->perhead = d->kind == VDETR_ATTN_PER_HEAD ? 1 : 0;
  A->aux = d->bwd_aux; A->v = v; A->nK = d->nK;
  A->v_stride = d->v_row_stride ? d->v_row_stride : 64;
  A->qblocks = (int)(((long)d->B * d->nQ + 3) / 4);
  A->vblocks = d->bwd_aux ? (int)(((long)d->B * d->nK + 4 * kDeltaKeys - 1) / (4 * kDeltaKeys)) : 0;
  A->vertices = d->table ? d->vertices : nullptr;
  A->cos_sin = d->table ? d->cos_sin : nullptr;
}
__device__ __forceinline__ void attn_delta_body(const DeltaArgs& A, int block, float* wmax) {
  const float* __restrict__ dout = A.dout;
  const float* __restrict__ out = A.out;
  float* __restrict__ delta = A.delta;
  unsigned* aux = A.aux;
  const int B = A.B, nQ = A.nQ, H = A.H, nK = A.nK;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float n2max = 0.f;
  int slot = 0;
  if (block >= A.qblocks) {  // |V row|^2 (shared-KV: 64 floats per key), kDeltaKeys keys per wave
    slot = 1;
    const int key0 = ((block - A.qblocks) * 4 + wv) * kDeltaKeys;
#pragma unroll 4
    for (int i = 0; i < kDeltaKeys; ++i) {
      const int key = key0 + i;
      const float x = key < B * nK ? A.v[(size_t)key * A.v_stride + lane] : 0.f;
      n2max = fmaxf(n2max, wave_allsum_f32(x * x));
    }
  } else {
    const int row = block * 4 + wv;
    if (row < B * nQ) {
      const int b = row / nQ, q = row - b * nQ;
      for (int h = 0; h < H; ++h) {
        const size_t e = ((size_t)row * H + h) * 64 + lane;
        const float g = dout[e];
        const float s = wave_allsum_f32(g * out[e]);
        if (aux) n2max = fmaxf(n2max, wave_allsum_f32(g * g));
        if (lane == 0) delta[A.perhead ? ((size_t)b * H + h) * nQ + q : (size_t)row * H + h] = s;
      }
      if (aux && A.vertices && row == 0 && lane == 0) aux[5] = 1u;  // "word 4 is meaningful": without it the box kernels stay off
      if (aux && A.vertices) {  // aux[4] += 1 for a query whose 8 RPE vertices are not a box (in the frame the kernels look up in)
        const float* vp = A.vertices + (size_t)row * 24;
        const int i = lane & 7;
        bool ok;
        if (!A.cos_sin) {  // no rotation operand: an axis-aligned box, bit for bit
          ok = vp[i * 3] == vp[rpe_box_xi(i) ? 6 : 0] && vp[i * 3 + 1] == vp[rpe_box_yi(i) ? 4 : 1] &&
               vp[i * 3 + 2] == vp[rpe_box_zi(i) ? 14 : 2];
        } else {
          // angle_type "object_coords": the offsets are turned by the query's angle before the look-up (rpe_rotate), and the
          // corners of a ROTATED box are an axis-aligned box in that frame: R (P_i - P_0) = (xi EX, yi EY, zi EZ) with the edge
          // vectors of vertices 3, 1 and 4.  The corners come out of fp32 arithmetic (box_decode), so the test has a tolerance
          // of a few ulps of the coordinates; the box kernel then looks up at R (P_0 - X) + (xi EX, yi EY, zi EZ), which
          // differs from the forward's R (P_i - X) by that much: a shift of the trilinear weights far below the 1e-3 budget.
          float ex = vp[i * 3] - vp[0], ey = vp[i * 3 + 1] - vp[1];
          const float ez = vp[i * 3 + 2] - vp[2];
          rpe_rotate(ex, ey, A.cos_sin[(size_t)row * 2], A.cos_sin[(size_t)row * 2 + 1]);
          const float EX = readlane_f32(ex, 3), EY = readlane_f32(ey, 1), EZ = readlane_f32(ez, 4);
          const float tol = 1e-5f * (1.f + fabsf(vp[0]) + fabsf(vp[1]) + fabsf(vp[2]));
          ok = fabsf(ex - (rpe_box_xi(i) ? EX : 0.f)) <= tol && fabsf(ey - (rpe_box_yi(i) ? EY : 0.f)) <= tol &&
               fabsf(ez - (rpe_box_zi(i) ? EZ : 0.f)) <= tol;
        }
        if (!__all(ok) && lane == 0) atomicAdd(aux + 4, 1u);
      }
    }
  }
  if (!aux) return;
  if (lane == 0) wmax[wv] = n2max;
  __syncthreads();  // one atomic per workgroup: thousands of atomics on one word serialise (measured 32 us per launch)
  if (threadIdx.x == 0)
    atomicMax(aux + slot, __float_as_uint(fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]))));
}

}  // namespace vdetr
