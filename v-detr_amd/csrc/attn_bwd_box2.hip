// attn_bwd_box2.hip — softmax backward + 3DV-RPE table gradient for axis-aligned boxes, second design (gfx950).
//
// Same contract and gating as attn_bwd_box.hip (z-half workgroups, device gate on bwd_aux[4], int32 fixed-point histogram).
// What changed: the 128 products  w_z w_y * w_x dS  of a pair are never formed on the VALU.  They are an OUTER PRODUCT
//     U[c] (8 values: c = (cz, cy, y-index) -> w_z[cz] w_y[yi][cy])   x   T[n] (16 values: n = (x-index, cx, head) -> w_x dS)
// and a sum of outer products over the pairs of a group is what a matrix unit computes:  C[c][n] = sum_p U_p[c] T_p[n].
// So the pairs of a 64-key chunk are SORTED by group (joint signature J as in attn_bwd_box.hip) into k-slots, two groups
// share one v_mfma_f32_16x16x32_bf16 (A rows = 2 groups x 8 c, B columns = 16 n, K = 8 pairs x 4 split terms), and a
// group's 128 sums are complete when its last pair has gone through the matrix unit — then they are flushed to the
// histogram.  Per pair and lane the VALU work is one U or T product and its bf16 split (hi + lo, all four cross terms
// go through the MFMA: relative error 2^-15) instead of 32 products and 32 splits; matrix instructions per chunk drop
// from ~44 to ~10 because every k-slot carries a real pair (groups are padded to multiples of 4 pairs only).
//   k-slot layout of MFMA i: slot = 8 i + 4 st + w  (st = stream 0 / 1: the two groups in flight, w = 0..3), lane group
//   kk = slot >> 1 supplies slots 2 kk, 2 kk + 1; A rows of stream st are zero for the other stream's slots.
#include "attn_common.h"

#include <stdlib.h>

namespace vdetr {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef f32x4 __attribute__((may_alias)) rec4_t;
typedef float __attribute__((may_alias)) rec1_t;
typedef int __attribute__((may_alias)) reci_t;
typedef unsigned short __attribute__((may_alias)) recs_t;
typedef unsigned char __attribute__((may_alias)) recb_t;

constexpr int kB2Waves = 16;
constexpr int kB2Threads = kB2Waves * kWave;
constexpr int kB2T = 10;                 // table edge ("bilinear_4_10")
constexpr int kB2RecStride = 20;         // words per pair record (14 used): 80-B rows make the 16-B stores conflict-free
constexpr int kB2MaxHalf = 48;           // half-slots per stream (worst case 40: (64 + 16) / 2)
constexpr int kB2SlotWords = kB2MaxHalf * 8 / 4;   // slot table: one byte per k-slot
constexpr int kB2StripWords = kWave * kB2RecStride + kB2SlotWords + 2 * kB2MaxHalf;

size_t attn_bwd_box2_lds_bytes() { return (size_t)4 * kB2T * kB2T * kB2T * 4 * 4 + (size_t)kB2Waves * kB2StripWords * 4; }

// F32: the products go through v_mfma_f32_16x16x4_f32 as they are (k-slot = one pair, 4 per instruction, 2 per stream)
// instead of as four bf16 cross terms (8 pairs per v_mfma_f32_16x16x32_bf16).  The matrix unit then runs ~4x longer —
// it was 3 % busy — and the VALU no longer splits: one multiply per operand element instead of multiply + mask +
// subtract + two byte permutes, and groups are padded to pairs, not quads.  The kernel is VALU-bound.
template <bool F32>
__global__ __launch_bounds__(kB2Threads) void attn_bwd_box2_kernel(AttnParams P) {
  constexpr int HS = F32 ? 2 : 4;  // pairs per half-slot (a stream's share of one matrix instruction)
  constexpr int T = kB2T, TT = T * T, T3 = TT * T;
  constexpr int table_words = 4 * T3 * 4;
  if (P.bwd_aux[4] != 0 || P.bwd_aux[5] == 0) return;  // a query is not an axis-aligned box: the general kernel runs instead
  extern __shared__ __attribute__((aligned(16))) float smem[];
  attn_load_rng(P);
  int* tab = reinterpret_cast<int*>(smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int part = blockIdx.x & 1, wg = blockIdx.x >> 1, nwg = gridDim.x >> 1;
  const int items = P.B * P.nQ;
  for (int i = tid; i < table_words; i += kB2Threads) tab[i] = 0;
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
  char* strip = reinterpret_cast<char*>(tab + table_words + wv * kB2StripWords);
  char* rec = strip;                                         // [64][20] floats
  char* slot_tab = strip + kWave * kB2RecStride * 4;         // [kB2MaxHalf * 8] bytes: pair of k-slot s, 0xFF = empty
  char* meta = slot_tab + kB2SlotWords * 4;                  // [2][kB2MaxHalf] ints: signature J of (stream, half-slot), -1 = empty
  const int kk = lane >> 4, c15 = lane & 15;
  // A role: row c15 = (gl, c) with c = (cz, cy, yi); active for the slots of stream gl only
  const int a_gl = c15 >> 3, a_cz = (c15 >> 2) & 1, a_cy = (c15 >> 1) & 1, a_yi = c15 & 1;
  const bool a_active = a_gl == (kk >> 1);
  const int a_offz = a_cz * 4, a_offy = (2 + 2 * a_cy + a_yi) * 4;      // record words wz[cz], wy[yi][cy] (bytes)
  // B role: column c15 = n = (xi, cx, h)
  const int b_xi = c15 >> 3, b_cx = (c15 >> 2) & 1, b_h = c15 & 3;
  const int b_offx = (6 + 2 * b_cx + b_xi) * 4, b_offd = (10 + b_h) * 4;
  // output role: rows 4 kk + r = (gl = kk >> 1, cz = kk & 1, cy = r >> 1, yi = r & 1), column n = (xi, cx, h)
  const int o_gl = kk >> 1, o_cz = kk & 1;
  const int o_const = ((o_cz * TT + b_cx) * 4 + b_h) * 4;               // byte offset of (cz, cy = 0, cx, h) inside a cell block
  const int o_vl[2] = {b_xi ? 3 : 0, b_xi ? 2 : 1};                     // local vertex of (xi, yi): attn_common.h:rpe_box_*
  const int o_xshift = 12 + 4 * b_xi;

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
  const bool dsg = P.ds_given != 0;  // dS comes from attn_bwd_kv.hip: one stream to read, no exp, no hash, no stores
  auto fetch = [&](rsrc_t rs, rsrc_t rd, rsrc_t rx, rsrc_t rm, bool has_mask, int chunk, ChunkOps& o) {
    const int key = chunk * kWave + lane;
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      if (!dsg) o.s[h] = ldf(rs, key * 4, h * rowbytes);
      o.d[h] = ldf(rd, key * 4, h * rowbytes);
    }
    o.kx = ldf(rx, key * 12, 0); o.ky = ldf(rx, key * 12 + 4, 0); o.kz = ldf(rx, key * 12 + 8, 0);
    o.masked = has_mask ? __builtin_amdgcn_raw_buffer_load_b8(rm, key, 0, 0) : 0;
  };
  const bool writer = part == 0;
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
    for (int h = 0; h < 4; ++h) { lse[h] = P.ds_given ? 0.f : uni(P.lse[row0 + h]); delta[h] = P.ds_given ? 0.f : uni(P.delta[row0 + h]); }
    const float* vp = P.vertices + ((size_t)b * P.nQ + q) * 24;
    const float X0 = uni(vp[0]), X1 = uni(vp[6]), Y0 = uni(vp[1]), Y1 = uni(vp[4]), Zp = uni(vp[part * 12 + 2]);
    const rsrc_t rsc = make_rsrc(P.scores + row0 * P.nK, 4u * rowbytes), rd = make_rsrc(P.dprob + row0 * P.nK, 4u * rowbytes);
    const rsrc_t rp = make_rsrc(P.probs_out + row0 * P.nK, 4u * rowbytes), rg = make_rsrc(P.ds_out + row0 * P.nK, 4u * rowbytes);
    const rsrc_t rx = make_rsrc(P.xyz + (size_t)b * P.nK * 3, 3u * rowbytes);
    const bool has_mask = P.mask_kind == VDETR_MASK_BOOL && !dsg;
    const rsrc_t rm = make_rsrc(has_mask ? reinterpret_cast<const unsigned char*>(P.mask) + ((size_t)b * P.nQ + q) * P.nK
                                         : reinterpret_cast<const unsigned char*>(P.xyz), has_mask ? (unsigned)P.nK : 0u);
    ChunkOps ops, nxt;
    fetch(rsc, rd, rx, rm, has_mask, wv, ops);

    for (int chunk = wv; chunk < nchunks; chunk += kB2Waves) {
      if (chunk + kB2Waves < nchunks) fetch(rsc, rd, rx, rm, has_mask, chunk + kB2Waves, nxt);
      // ---- element-wise softmax backward of this lane's pair -----------------------------------------------------------
      const int key = chunk * kWave + lane;
      const bool valid = key < P.nK;
      float ds[4];
      if (dsg) {
#pragma unroll
        for (int h = 0; h < 4; ++h) ds[h] = valid ? ops.d[h] * fix_scale : 0.f;
      } else {
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
        char* mine = rec + lane * (kB2RecStride * 4);
        *reinterpret_cast<rec4_t*>(mine) = f32x4{az.wa, az.wb, ay0.wa, ay1.wa};          // wz[cz], wy[yi][cy=0]
        *reinterpret_cast<rec4_t*>(mine + 16) = f32x4{ay0.wb, ay1.wb, ax0.wa, ax1.wa};   // wy[yi][cy=1], wx[xi][cx=0]
        *reinterpret_cast<rec4_t*>(mine + 32) = f32x4{ax0.wb, ax1.wb, ds[0], ds[1]};     // wx[xi][cx=1], dS
        *reinterpret_cast<rec4_t*>(mine + 48) = f32x4{ds[2], ds[3], 0.f, 0.f};
        // clear the slot table (0xFF = empty) and the stream metadata (-1)
        reinterpret_cast<reci_t*>(slot_tab)[lane] = -1;
        if (lane < kB2SlotWords + 2 * kB2MaxHalf - kWave) reinterpret_cast<reci_t*>(slot_tab)[kWave + lane] = -1;
        if (lane < kB2SlotWords + 2 * kB2MaxHalf - 2 * kWave) reinterpret_cast<reci_t*>(slot_tab)[2 * kWave + lane] = -1;
      }
      __builtin_amdgcn_wave_barrier();
      // ---- sort the pairs by group into k-slots: group -> stream with the fewer half-slots so far -----------------------
      int n0 = 0, n1 = 0, myslot = 0;
      {
        unsigned long long todo = ~0ull;
        while (todo) {
          const int leader = __builtin_ctzll(todo);
          const int jl = __builtin_amdgcn_readlane(J, leader);
          const unsigned long long m = __ballot(J == jl);
          todo &= ~m;
          const int hs = (__builtin_popcountll(m) + HS - 1) / HS;  // half-slots of this group
          const int st = n1 < n0 ? 1 : 0;
          const int base = st ? n1 : n0;
          const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
          if (J == jl) myslot = (base + rank / HS) * (2 * HS) + st * HS + rank % HS;
          if (lane < hs) reinterpret_cast<reci_t*>(meta)[st * kB2MaxHalf + base + lane] = jl;
          if (st) n1 += hs; else n0 += hs;
        }
      }
      reinterpret_cast<recb_t*>(slot_tab)[myslot] = (unsigned char)lane;
      const int nmfma = n0 > n1 ? n0 : n1;
      __builtin_amdgcn_wave_barrier();

      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      int Jcur = reinterpret_cast<const reci_t*>(meta)[o_gl * kB2MaxHalf];
      for (int i = 0; i < nmfma; ++i) {
        const int Jnext = reinterpret_cast<const reci_t*>(meta)[o_gl * kB2MaxHalf + i + 1];
        if (F32) {  // slot 4 i + kk: one pair per lane group
          const unsigned p = reinterpret_cast<const recb_t*>(slot_tab)[i * 4 + kk];
          const bool occupied = p != 0xFFu;
          const char* r = rec + (p & 63u) * (kB2RecStride * 4);
          const float u = *reinterpret_cast<const rec1_t*>(r + a_offz) * *reinterpret_cast<const rec1_t*>(r + a_offy);
          const float t = *reinterpret_cast<const rec1_t*>(r + b_offx) * *reinterpret_cast<const rec1_t*>(r + b_offd);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32((a_active && occupied) ? u : 0.f, occupied ? t : 0.f, acc, 0, 0, 0);
        } else {
        const unsigned sp = reinterpret_cast<const recs_t*>(slot_tab)[i * 4 + kk];  // slots 8 i + 2 kk, + 1
        i32x4 aw, bw;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const unsigned p = (sp >> (8 * e)) & 0xFFu;
          const bool occupied = p != 0xFFu;
          const char* r = rec + (p & 63u) * (kB2RecStride * 4);
          // A: U = w_z[cz] w_y[yi][cy] of this row, split into (hi, lo) bf16; zero for the other stream / an empty slot
          const float u = *reinterpret_cast<const rec1_t*>(r + a_offz) * *reinterpret_cast<const rec1_t*>(r + a_offy);
          const float uh = __int_as_float(__float_as_int(u) & 0xFFFF0000);
          const int uw = (int)__builtin_amdgcn_perm((unsigned)__float_as_int(u), (unsigned)__float_as_int(u - uh), 0x07060302u);
          const int a = (a_active && occupied) ? uw : 0;   // (elem 1, elem 0) = (hi, lo)
          aw[2 * e] = a; aw[2 * e + 1] = a;
          // B: T = w_x[xi][cx] dS[h] of this column: (hi, hi) and (lo, lo) -> all four cross terms
          const float t = *reinterpret_cast<const rec1_t*>(r + b_offx) * *reinterpret_cast<const rec1_t*>(r + b_offd);
          const float th = __int_as_float(__float_as_int(t) & 0xFFFF0000);
          const float tl = t - th;
          bw[2 * e] = (int)__builtin_amdgcn_perm((unsigned)__float_as_int(t), (unsigned)__float_as_int(t), 0x07060706u);
          bw[2 * e + 1] = (int)__builtin_amdgcn_perm((unsigned)__float_as_int(tl), (unsigned)__float_as_int(tl), 0x07060706u);
        }
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, aw), __builtin_bit_cast(bf16x8, bw), acc, 0, 0, 0);
        }
        // ---- a stream's group is complete when the next half-slot belongs to another group: flush its 128 sums ----------
        if (Jcur != Jnext) {
          if (Jcur >= 0) {
            const int zb = Jcur & 15, xb = (Jcur >> o_xshift) & 15;
#pragma unroll
            for (int yi = 0; yi < 2; ++yi) {
              const int yb = (Jcur >> (4 + 4 * yi)) & 15;
              const int cell = (zb * T + yb) * T + xb;
              char* bin = reinterpret_cast<char*>(tab) + (o_vl[yi] * T3 + cell) * 16 + o_const;
              atomicAdd(reinterpret_cast<int*>(bin), __float2int_rn(acc[yi]));                 // r = yi     : cy = 0
              atomicAdd(reinterpret_cast<int*>(bin + T * 16), __float2int_rn(acc[2 + yi]));    // r = 2 + yi : cy = 1
            }
          }
          acc = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        Jcur = Jnext;
      }
      ops = nxt;
    }
  }
  __syncthreads();
  float* dst = P.dtable_part + (size_t)blockIdx.x * table_words;
  for (int i = tid; i < table_words; i += kB2Threads) dst[i] = (float)tab[i] * fix_inv;
}

int launch_attn_bwd_box2(const AttnParams& P, int grid, hipStream_t st, bool f32_products) {
  const size_t lds = attn_bwd_box2_lds_bytes();
  if (f32_products) {
    if (int e = set_lds(attn_bwd_box2_kernel<true>, lds, "attn_bwd_box2")) return e;
    hipLaunchKernelGGL(attn_bwd_box2_kernel<true>, dim3(grid), dim3(kB2Threads), lds, st, P);
  } else {
    if (int e = set_lds(attn_bwd_box2_kernel<false>, lds, "attn_bwd_box2")) return e;
    hipLaunchKernelGGL(attn_bwd_box2_kernel<false>, dim3(grid), dim3(kB2Threads), lds, st, P);
  }
  return check_launch("attn_bwd_box2");
}

}  // namespace vdetr
