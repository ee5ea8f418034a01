// attn_bwd_box.hip — softmax backward + 3DV-RPE table gradient for AXIS-ALIGNED boxes (ScanNet: one angle bin), gfx950.
//
// Same contract as attn_bwd_scores_rpe_mm_kernel (attn_bwd.hip), which stays the path for rotated boxes / arbitrary vertices.
// What the box structure buys (vertices = corners of a box: two coordinate values per axis, attn_common.h:rpe_box_*):
//   * a workgroup owns one z-half of the box (4 vertices that share their z coordinate): 5 axis taps per (query, key) pair
//     instead of 12, computed once per pair;
//   * ONE grouping per 64-key chunk for all 4 vertices: pairs are grouped by the joint signature
//     J = (zbase, ybase[0..1], xbase[0..1]); pairs with equal J share their lookup cell in every vertex table.  The groups
//     come from a scalar ballot loop (no LDS scoreboard), lane g keeps group g's membership mask and signature;
//   * the 128 products  w_z w_y w_x dS  of a pair (4 vertices x 8 corners x 4 heads) are never transposed through LDS:
//     a pair leaves a 14-float RECORD (its 10 axis weights + 4 dS) in the wave's LDS strip, and the lane that owns column
//     n = (cy, cx, head) of the MFMA B operand rebuilds the products of its 16 pairs from the records (broadcast reads).
//     LDS traffic per chunk: 4 KB written + 64 small reads per lane, against 16 KB written + 128 reads before;
//   * G[group][value] = sum_pairs M[group][pair] * V[pair][value] on the bf16 matrix pipe with V = hi + lo (rel. error
//     2^-16), M exact, fp32 accumulation; the 16 x 128 group sums go to the workgroup's int32 fixed-point histogram with
//     ds_add_u32 (order independent, run-to-run deterministic).
// The launch is gated on the device: vdetr_attn_delta_f32 counts the queries whose vertices are NOT an axis-aligned box in
// bwd_aux[4]; this kernel runs when the count is 0 and the general kernel when it is not (each returns at once otherwise).
#include "attn_common.h"

#include <stdlib.h>

namespace vdetr {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// the pair records are written as 16-B blocks and read back as 8- / 4-B pieces: tell the alias analysis so
typedef f32x4 __attribute__((may_alias)) rec4_t;
typedef f32x2 __attribute__((may_alias)) rec2_t;
typedef float __attribute__((may_alias)) rec1_t;

constexpr int kBoxRecWords = 16;                     // floats per pair record (14 used)
constexpr int kBoxStripWords = kWave * kBoxRecWords;  // one 4 KB strip per wave
constexpr int kBoxT = 10;                            // table edge this kernel is compiled for ("bilinear_4_10")

template <int kBoxWaves>
__global__ __launch_bounds__(kBoxWaves * kWave) void attn_bwd_box_kernel(AttnParams P) {
  constexpr int kBoxThreads = kBoxWaves * kWave;
  constexpr int T = kBoxT, TT = T * T, T3 = TT * T;
  constexpr int table_words = 4 * T3 * 4;  // 4 vertex tables x cells x heads
  if (P.bwd_aux[4] != 0 || P.bwd_aux[5] == 0) return;            // a query is not an axis-aligned box: the general kernel runs instead
  extern __shared__ __attribute__((aligned(16))) float smem[];
  attn_load_rng(P);
  int* tab = reinterpret_cast<int*>(smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int part = blockIdx.x & 1, wg = blockIdx.x >> 1, nwg = gridDim.x >> 1;
  const int items = P.B * P.nQ;
  for (int i = tid; i < table_words; i += kBoxThreads) tab[i] = 0;
  // fixed-point scale of the histogram: |dP~| <= |dO row| |V row| (maxima from the delta launch), at most `cap` queries
  const int per_wg = (items + nwg - 1) / nwg;
  const int cap = bwd_query_cap(per_wg);
  float fix_scale = 1.f, fix_inv = 1.f;
  {
    const float dmax = sqrtf(__uint_as_float(P.bwd_aux[0]) * __uint_as_float(P.bwd_aux[1]));
    const float bound = 2.f * P.drop_scale * dmax * (float)cap;
    if (bound > 0.f && bound < INFINITY) {
      const int e = 30 - (int)ceilf(__log2f(bound) + 1e-3f);  // 2^e * bound <= 2^30
      fix_scale = ldexpf(1.f, e);
      fix_inv = ldexpf(1.f, -e);
    }
  }
  __syncthreads();
  int* rec = tab + table_words + wv * kBoxStripWords;
  const int kk = lane >> 4, c15 = lane & 15;
  const int cy = (c15 >> 3) & 1, cx = (c15 >> 2) & 1, hh = c15 & 3;
  // bins of this lane's output column (cy, cx, head) for the corner planes cz = 0 / 1, in bytes
  const int off0 = ((cy * T + cx) * 4 + hh) * 4;
  // record addressing (bytes): pair p's 16-B block j sits at position (j + (p >> 1)) & 3 (conflict-free b128 stores);
  // this lane reads pairs p = 16 t + 4 m + kk: rotation (2 m + (kk >> 1)) & 3 -> even m: r0, odd m: r0 ^ 2
  const int r0 = kk >> 1;
  const int rd_z = kk * 64 + (((0 + r0) & 3) * 4) * 4;                          // words 0..1  (block 0)
  const int rd_y = kk * 64 + ((((cy ? 1 : 0) + r0) & 3) * 4 + (cy ? 0 : 2)) * 4;  // words 2+2cy.. (block cy)
  const int rd_x = kk * 64 + ((((cx ? 2 : 1) + r0) & 3) * 4 + (cx ? 0 : 2)) * 4;  // words 6+2cx.. (block 1 + cx)
  const int rd_d = kk * 64 + ((((hh < 2 ? 2 : 3) + r0) & 3) * 4 + ((2 + hh) & 3)) * 4;  // word 10+h
  const int wr_rot = (lane >> 1) & 3;

  struct ChunkOps {
    float s[4], d[4], kx, ky, kz;
    unsigned char masked;
  };
  using rsrc_t = __amdgpu_buffer_rsrc_t;
  auto make_rsrc = [](const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
  };
  auto ldf = [](rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
  };
  const int rowbytes = P.nK * 4;
  auto fetch = [&](rsrc_t rs, rsrc_t rd, rsrc_t rx, rsrc_t rm, bool has_mask, int chunk, ChunkOps& o) {
    const int key = chunk * kWave + lane;
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      o.s[h] = ldf(rs, key * 4, h * rowbytes);
      o.d[h] = ldf(rd, key * 4, h * rowbytes);
    }
    o.kx = ldf(rx, key * 12, 0); o.ky = ldf(rx, key * 12 + 4, 0); o.kz = ldf(rx, key * 12 + 8, 0);
    o.masked = has_mask ? __builtin_amdgcn_raw_buffer_load_b8(rm, key, 0, 0) : 0;
  };
  const bool writer = part == 0;  // the z-half 0 workgroups store P~ / dS of the chunks they visit
  const int nchunks = (P.nK + kWave - 1) / kWave;

  __shared__ int next_item;
  for (int it = 0; it < cap; ++it) {
    if (tid == 0) next_item = (int)atomicAdd(const_cast<unsigned*>(P.bwd_aux) + 2 + part, 1u);
    __syncthreads();
    const int item = next_item;
    __syncthreads();
    if (item >= items) break;
    const int b = item / P.nQ, q = item - b * P.nQ;
    const size_t row0 = ((size_t)b * P.nQ + q) * 4;
    auto uni = [](float x) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x))); };
    float lse[4], delta[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) { lse[h] = uni(P.lse[row0 + h]); delta[h] = uni(P.delta[row0 + h]); }
    const float* vp = P.vertices + ((size_t)b * P.nQ + q) * 24;
    // box coordinates: x of vertices 0 / 2, y of vertices 0 / 1, z of vertex 4 * part   (attn_common.h:rpe_box_*)
    const float X0 = uni(vp[0]), X1 = uni(vp[6]), Y0 = uni(vp[1]), Y1 = uni(vp[4]), Zp = uni(vp[part * 12 + 2]);
    const rsrc_t rsc = make_rsrc(P.scores + row0 * P.nK, 4u * rowbytes), rd = make_rsrc(P.dprob + row0 * P.nK, 4u * rowbytes);
    const rsrc_t rp = make_rsrc(P.probs_out + row0 * P.nK, 4u * rowbytes), rg = make_rsrc(P.ds_out + row0 * P.nK, 4u * rowbytes);
    const rsrc_t rx = make_rsrc(P.xyz + (size_t)b * P.nK * 3, 3u * rowbytes);
    const bool has_mask = P.mask_kind == VDETR_MASK_BOOL;
    const rsrc_t rm = make_rsrc(has_mask ? reinterpret_cast<const unsigned char*>(P.mask) + ((size_t)b * P.nQ + q) * P.nK
                                         : reinterpret_cast<const unsigned char*>(P.xyz), has_mask ? (unsigned)P.nK : 0u);
    ChunkOps ops, nxt;
    fetch(rsc, rd, rx, rm, has_mask, wv, ops);

    for (int chunk = wv; chunk < nchunks; chunk += kBoxWaves) {
      if (chunk + kBoxWaves < nchunks) fetch(rsc, rd, rx, rm, has_mask, chunk + kBoxWaves, nxt);
      // ---- element-wise softmax backward of this lane's pair -----------------------------------------------------------
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
          if (writer && valid) {
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(g.p_drop), rp, key * 4, h * rowbytes, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(g.ds * P.scale), rg, key * 4, h * rowbytes, 0);
          }
          ds[h] = valid ? g.ds * fix_scale : 0.f;  // power-of-two scale: exact
        }
      }
      // ---- 5 axis taps, joint signature, pair record -----------------------------------------------------------------
      const AxisTap az = rpe_axis(Zp - ops.kz, P);
      const AxisTap ay0 = rpe_axis(Y0 - ops.ky, P), ay1 = rpe_axis(Y1 - ops.ky, P);
      const AxisTap ax0 = rpe_axis(X0 - ops.kx, P), ax1 = rpe_axis(X1 - ops.kx, P);
      const int J = az.base | (ay0.base << 4) | (ay1.base << 8) | (ax0.base << 12) | (ax1.base << 16);
      __builtin_amdgcn_wave_barrier();
      {
        int* mine = rec + lane * kBoxRecWords;
        *reinterpret_cast<rec4_t*>(mine + ((0 + wr_rot) & 3) * 4) = f32x4{az.wa, az.wb, ay0.wa, ay1.wa};
        *reinterpret_cast<rec4_t*>(mine + ((1 + wr_rot) & 3) * 4) = f32x4{ay0.wb, ay1.wb, ax0.wa, ax1.wa};
        *reinterpret_cast<rec4_t*>(mine + ((2 + wr_rot) & 3) * 4) = f32x4{ax0.wb, ax1.wb, ds[0], ds[1]};
        *reinterpret_cast<rec4_t*>(mine + ((3 + wr_rot) & 3) * 4) = f32x4{ds[2], ds[3], 0.f, 0.f};
      }
      // ---- groups = distinct signatures among the 64 pairs: lane g keeps group g's membership mask and signature ------
      int ngroups = 0;
      int g_lo = 0, g_hi = 0, g_J = 0;
      {
        unsigned long long todo = ~0ull;
        while (todo) {
          const int leader = __builtin_ctzll(todo);
          const int jl = __builtin_amdgcn_readlane(J, leader);
          const unsigned long long m = __ballot(J == jl);
          todo &= ~m;
          const bool me = lane == ngroups;  // lane g keeps group g
          g_lo = me ? (int)(unsigned)m : g_lo;
          g_hi = me ? (int)(unsigned)(m >> 32) : g_hi;
          g_J = me ? jl : g_J;
          ++ngroups;
        }
      }
      __builtin_amdgcn_wave_barrier();  // records visible to the wave's readers (LDS is in-order within a wave)

      for (int g0 = 0; g0 < ngroups; g0 += 16) {  // 16 groups per round
        // membership of this lane's row (group g0 + c15) for its k-slots: pair p = 16 t + 4 m + kk
        const int src = (g0 + c15) << 2;
        const unsigned mlo = (unsigned)__builtin_amdgcn_ds_bpermute(src, g_lo);
        const unsigned mhi = (unsigned)__builtin_amdgcn_ds_bpermute(src, g_hi);
        const bool rowok = g0 + c15 < ngroups;
        const unsigned long long msk = rowok ? (((unsigned long long)mhi << 32) | mlo) >> kk : 0ull;
        const unsigned mw[2] = {(unsigned)msk, (unsigned)(msk >> 32)};
        f32x4 acc[4][2];
#pragma unroll
        for (int vl = 0; vl < 4; ++vl)
#pragma unroll
          for (int cz = 0; cz < 2; ++cz) acc[vl][cz] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          i32x4 am;
#pragma unroll
          for (int m = 0; m < 4; ++m)  // bit 16 (t & 1) + 4 m of word t >> 1, sign-extended -> both bf16 halves = 1.0
            am[m] = __builtin_amdgcn_sbfe((int)mw[t >> 1], 16 * (t & 1) + 4 * m, 1) & 0x3F803F80;
          i32x4 bw[4][2];
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            const int pb = (16 * t + 4 * m) * kBoxRecWords * 4;  // byte offset of pair 16 t + 4 m (+ kk: in rd_*)
            const int fl = (m & 1) ? 32 : 0;                       // odd m: rotation ^ 2 = byte offset ^ 32
            const char* base = reinterpret_cast<const char*>(rec) + pb;
            const f32x2 z2 = *reinterpret_cast<const rec2_t*>(base + (rd_z ^ fl));
            const f32x2 y2 = *reinterpret_cast<const rec2_t*>(base + (rd_y ^ fl));
            const f32x2 x2 = *reinterpret_cast<const rec2_t*>(base + (rd_x ^ fl));
            const float dd = *reinterpret_cast<const rec1_t*>(base + (rd_d ^ fl));
            // SCALAR products on purpose.  Written as float2 arithmetic (u1 = z2 * y2[1], v = u * t1[1]) the compiler emits
            // v_pk_mul_f32 with op_sel:[0,1] (both result lanes read the HIGH register of the source pair), and with that
            // form the kernel produced wrong sums on MI355X (ROCm 7.2): every vertex whose products pass through a
            // hi-broadcast operand was off, deterministically in the 8-wave build and intermittently under load in the
            // 16-wave one, while the low-broadcast form (op_sel_hi:[1,0], used by the forward kernel) is fine.
            const float t10 = x2[0] * dd, t11 = x2[1] * dd;  // (x index 0, 1) * dS
            const float u00 = z2[0] * y2[0], u01 = z2[1] * y2[0], u10 = z2[0] * y2[1], u11 = z2[1] * y2[1];  // u[y index][cz]
            // local vertex vl: x index (vl >> 1), y index 1 for vl in {1, 2}   (attn_common.h:rpe_box_xi / _yi)
            const f32x2 v[4] = {f32x2{u00 * t10, u01 * t10}, f32x2{u10 * t10, u11 * t10}, f32x2{u10 * t11, u11 * t11},
                                f32x2{u00 * t11, u01 * t11}};
#pragma unroll
            for (int vl = 0; vl < 4; ++vl) {
              const f32x2 hi = {__int_as_float(__float_as_int(v[vl][0]) & 0xFFFF0000), __int_as_float(__float_as_int(v[vl][1]) & 0xFFFF0000)};
              const f32x2 lo = v[vl] - hi;  // exact
#pragma unroll
              for (int cz = 0; cz < 2; ++cz)
                bw[vl][cz][m] = (int)__builtin_amdgcn_perm((unsigned)__float_as_int(v[vl][cz]), (unsigned)__float_as_int(lo[cz]), 0x07060302u);
            }
          }
#pragma unroll
          for (int vl = 0; vl < 4; ++vl)
#pragma unroll
            for (int cz = 0; cz < 2; ++cz)
              acc[vl][cz] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, am), __builtin_bit_cast(bf16x8, bw[vl][cz]),
                                                                    acc[vl][cz], 0, 0, 0);
        }
        // ---- group sums -> histogram: lane holds column (cy, cx, head) of groups g0 + 4 kk + r ---------------------------
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int g = g0 + 4 * kk + r;
          const int Jg = __builtin_amdgcn_ds_bpermute(g << 2, g_J);
          if (g < ngroups) {
            const int zb = Jg & 15, yb0 = (Jg >> 4) & 15, yb1 = (Jg >> 8) & 15, xb0 = (Jg >> 12) & 15, xb1 = (Jg >> 16) & 15;
            const int zy0 = (zb * T + yb0) * T, zy1 = (zb * T + yb1) * T;
            const int cell[4] = {zy0 + xb0, zy1 + xb0, zy1 + xb1, zy0 + xb1};
#pragma unroll
            for (int vl = 0; vl < 4; ++vl) {
              char* bin = reinterpret_cast<char*>(tab) + (vl * T3 + cell[vl]) * 16 + off0;
              atomicAdd(reinterpret_cast<int*>(bin), __float2int_rn(acc[vl][0][r]));
              atomicAdd(reinterpret_cast<int*>(bin + TT * 16), __float2int_rn(acc[vl][1][r]));
            }
          }
        }
      }
      ops = nxt;
    }
  }
  __syncthreads();
  float* dst = P.dtable_part + (size_t)blockIdx.x * table_words;
  for (int i = tid; i < table_words; i += kBoxThreads) dst[i] = (float)tab[i] * fix_inv;
}

template <int WAVES>
static int launch_box(const AttnParams& P, int grid, hipStream_t st) {
  const size_t lds = (size_t)4 * kBoxT * kBoxT * kBoxT * 4 * 4 + (size_t)WAVES * kBoxStripWords * 4;
  if (int e = set_lds(attn_bwd_box_kernel<WAVES>, lds, "attn_bwd_box")) return e;
  hipLaunchKernelGGL((attn_bwd_box_kernel<WAVES>), dim3(grid), dim3(WAVES * kWave), lds, st, P);
  return check_launch("attn_bwd_box");
}

int launch_attn_bwd_box(const AttnParams& P, int grid, hipStream_t st) {
  static const int waves = [] { const char* v = getenv("VDETR_BOX_WAVES"); return v ? atoi(v) : 16; }();
  if (waves == 8) return launch_box<8>(P, grid, st);
  if (waves == 12) return launch_box<12>(P, grid, st);
  return launch_box<16>(P, grid, st);
}

}  // namespace vdetr
