// attn_bwd_box3.hip — 3DV-RPE table gradient for axis-aligned boxes from a given dS, third design (gfx950; round 4).
//
// Reference math: vdetr_transformer.py:710-731 backward (grid_sampler_3d_backward of the eight per-vertex tables).
// Same contract as attn_bwd_box2.hip in its dS-given mode (z-half workgroups that pull queries from a device counter, device
// gate on bwd_aux[4] / [5], int32 fixed-point histogram in LDS, one partial table per workgroup).  What changed:
//
// box2 grouped the pairs of ONE 64-key chunk per wave: 12.7 groups of ~5 pairs, every group padded to a multiple of 4 k-slots,
// sorted by a ballot loop, its 128 sums flushed to the histogram, and the outer products fed to the bf16 matrix instructions as
// four split terms formed on the VALU: 1,100 VALU instructions per 64 pairs and z-half, 143 M per launch — the kernel was
// VALU-issue bound.  Here the WORKGROUP sorts 1024 keys at a time:
//   1  one key per thread: 5 axis taps, the joint signature J of their cells, a dense group id (the two taps of an axis move
//      along a monotone staircase as the key coordinate grows, so base0 + base1 names the pair: 9 x 17 x 17 ids), and a STABLE
//      rank inside the group: the counter of a group is a u64 of eight bytes, one per pair of waves; a lane adds 1 to its
//      pair's byte with a returning ds_add_u64 — even waves first, odd waves after a barrier — so the byte it reads back is
//      the number of earlier keys of its wave pair in the group, and the bytes of the lower pairs (read again after the
//      barrier) are the keys before those: rank = key order, whatever order the waves arrive in.  (The sums below are float
//      sums: an arrival-order rank made the result differ in the last bit from run to run.)
//   2  exclusive scan of the 2,601 group sizes (three per thread, DPP wave scan)
//   3  the pair's record (8 products w_z w_y, 4 weights w_x, 4 dS; 16-byte parts XOR-swizzled by the slot number so that the
//      scattered 16-byte stores spread over all banks) goes to its sorted slot: groups are contiguous runs, ~97 per 1024 keys
//      instead of 16 x 12.7, nothing is padded
//   4  waves draw chunks of 32 sorted slots from an LDS counter (groups differ in length, so equal slot counts are not equal
//      work): the operands of the chunk's 8 quads are read in one burst (3 LDS reads and 1 multiply per lane and quad), group ends
//      come from comparing neighbouring J (a signature that shares an id with another one costs a flush, never a wrong cell),
//      a quad of slots is ONE v_mfma_f32_16x16x4_f32 (rows: the 8 (cy, cz, yi) products w_z w_y, columns: the 16 (xi, cx, h)
//      products w_x dS, K = 4 pairs; exact fp32 products, no bf16 split), a quad that holds a group end is issued once per
//      segment with the other slots' rows zeroed, and a finished group's 128 sums are flushed with four ds_add_u32 per lane
//      (32 lanes = the (cy, cx, h) corner block of one vertex and z-cell: 32 different banks for any cell).
// Barriers inside the loop wait for LDS only (s_waitcnt lgkmcnt(0); s_barrier): __syncthreads() would also wait for the
// NEXT tile's global loads that are in flight across the whole tile (measured: 1 us per barrier).
// Products are exact fp32 now (box2: 2^-15 split-bf16); sums stay int32 fixed point.
#include "attn_common.h"

#include <stdlib.h>

namespace vdetr {

typedef f32x4 __attribute__((may_alias)) b3_rec4_t;
typedef float __attribute__((may_alias)) b3_rec1_t;
typedef int __attribute__((may_alias)) b3_reci_t;

constexpr int kB3Waves = 16;
constexpr int kB3Threads = kB3Waves * kWave;
constexpr int kB3Tile = kB3Threads;            // keys per sort: one per thread
constexpr int kB3T = 10;                       // table edge ("bilinear_4_10")
constexpr int kB3RecBytes = 64;                // 16 words: U[8] = wz wy, wx[4], dS[4] (parts of 16 B, XOR-swizzled by the slot)
constexpr int kB3Gids = 9 * 17 * 17;           // (z base, y base0 + base1, x base0 + base1)
constexpr int kB3GidSlots = 2604;              // counters: 3 per scanning thread, 868 threads scan
constexpr int kB3Chunk = 32;                   // sorted slots a wave draws at a time
constexpr int kB3TableWords = 4 * kB3T * kB3T * kB3T * 4;
constexpr int kB3Misc = 64;                    // words: wave totals [16], next item [16], chunk counter [20]
static_assert(kB3Gids <= kB3GidSlots && kB3GidSlots % 3 == 0 && kB3GidSlots / 3 <= kB3Threads, "scan covers every group id");

// LDS: histogram 64,000 | records 65,536 | group counters (u64: a byte per wave pair) 20,832 | first slots (u16) 5,208 (+ pad) |
// signatures of the sorted slots 4,096 | misc 256  =  159,936 B of 163,840
constexpr size_t kB3OffCnt = (size_t)kB3TableWords * 4 + (size_t)kB3Tile * kB3RecBytes;
constexpr size_t kB3OffFirst = kB3OffCnt + (size_t)kB3GidSlots * 8;
constexpr size_t kB3OffJ = (kB3OffFirst + (size_t)kB3GidSlots * 2 + 15) & ~(size_t)15;
constexpr size_t kB3OffMisc = kB3OffJ + (size_t)kB3Tile * 4;
size_t attn_bwd_box3_lds_bytes() { return kB3OffMisc + (size_t)kB3Misc * 4; }

// workgroup barrier that waits for this wave's LDS traffic only (global loads stay in flight)
__device__ __forceinline__ void b3_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// inclusive scan over the wave: Hillis-Steele inside each 16-lane row (row_shr 1, 2, 4, 8), then the row totals carried over
// by row_bcast:15 (rows 1, 3) and row_bcast:31 (rows 2, 3); lanes without a source add `old` = 0
__device__ __forceinline__ unsigned wave_incl_scan_u32(unsigned v) {
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, kDppRowBcast15, 0xA, 0xF, false);
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, kDppRowBcast31, 0xC, 0xF, false);
  return v;
}

__global__ __launch_bounds__(kB3Threads) void attn_bwd_box3_kernel(AttnParams P) {
  constexpr int T = kB3T, T3 = T * T * T;
  if (P.bwd_aux[4] != 0 || P.bwd_aux[5] == 0) return;  // a query is not an axis-aligned box: the general kernel runs instead
  extern __shared__ __attribute__((aligned(16))) float smem[];
  char* lds = reinterpret_cast<char*>(smem);
  int* tab = reinterpret_cast<int*>(lds);
  char* rec = lds + (size_t)kB3TableWords * 4;
  unsigned long long* cnt = reinterpret_cast<unsigned long long*>(lds + kB3OffCnt);
  unsigned short* first = reinterpret_cast<unsigned short*>(lds + kB3OffFirst);
  int* sigs = reinterpret_cast<int*>(lds + kB3OffJ);
  unsigned* misc = reinterpret_cast<unsigned*>(lds + kB3OffMisc);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int part = blockIdx.x & 1, nwg = gridDim.x >> 1;
  const int items = P.B * P.nQ;
  for (int i = tid; i < kB3TableWords; i += kB3Threads) tab[i] = 0;
  for (int i = tid; i < kB3GidSlots; i += kB3Threads) cnt[i] = 0ull;
  for (int i = tid; i < kB3Tile * kB3RecBytes / 4; i += kB3Threads) reinterpret_cast<b3_rec1_t*>(rec)[i] = 0.f;  // a partly filled quad multiplies 0 x (whatever the slot held): must be finite
  const int per_wg = (items + nwg - 1) / nwg;
  const int cap = bwd_query_cap(per_wg);
  float fix_scale = 1.f, fix_inv = 1.f;
  {  // the bound of attn_bwd_box2.hip: |bin sum| <= queries of this workgroup x 2 drop_scale max|dO row| max|V row|
    const float dmax = sqrtf(__uint_as_float(P.bwd_aux[0]) * __uint_as_float(P.bwd_aux[1]));
    const float bound = 2.f * P.drop_scale * dmax * (float)cap;
    if (bound > 0.f && bound < INFINITY) {
      const int e = 30 - (int)ceilf(__log2f(bound) + 1e-3f);  // 2^e * bound <= 2^30
      fix_scale = ldexpf(1.f, e);
      fix_inv = ldexpf(1.f, -e);
    }
  }
  unsigned* counter = const_cast<unsigned*>(P.bwd_aux) + 2 + part;
  if (tid == 0) { misc[16] = atomicAdd(counter, 1u); misc[20] = 0u; }
  __syncthreads();
  int item = (int)misc[16];
  int taken = 1;
  __syncthreads();

  const int kk = lane >> 4, c15 = lane & 15;
  // A role (row m = c15 of the matrix instruction): U[m], m = (cy, cz, yi), for m < 8; rows 8..15 stay zero.  Part p of slot s sits at 16 (p ^ (s & 3)); a lane's slots are 4 i + kk, so s & 3 = kk.
  const int a_off = (((c15 >> 2) & 1) ^ kk) * 16 + (c15 & 3) * 4;
  // B role (column n = c15 = (xi, cx, h)): wx[2 cx + xi] in part 2, dS[h] in part 3
  const int b_offx = (2 ^ kk) * 16 + (2 * ((c15 >> 2) & 1) + (c15 >> 3)) * 4, b_offd = (3 ^ kk) * 16 + (c15 & 3) * 4;
  // output role: lane (kk, n) register r holds row 4 kk + r = (cy = kk, cz = r >> 1, yi = r & 1), column n
  const int o_xi = c15 >> 3;
  const int o_lane = (((kk & 1) * T + ((c15 >> 2) & 1)) * 4 + (c15 & 3)) * 4;  // bytes of (cy, cx, h) inside a cell block

  using rsrc_t = __amdgpu_buffer_rsrc_t;
  auto make_rsrc = [](const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
  };
  auto ldf = [](rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
  };
  auto uni = [](float x) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x))); };
  const int rowbytes = P.nK * 4;
  const int ntiles = (P.nK + kB3Tile - 1) / kB3Tile;
  struct Ops { float d[4], kx, ky, kz; };
  struct Box { float x0, x1, y0, y1, zp; };
  auto fetch = [&](int it, int tile, Ops& o) {  // dS of the 4 heads + the key's position (out-of-range keys read 0)
    const int b = it / P.nQ;
    const rsrc_t rd = make_rsrc(P.dprob + (size_t)it * 4 * P.nK, 4u * rowbytes);
    const rsrc_t rx = make_rsrc(P.xyz + (size_t)b * P.nK * 3, 3u * rowbytes);
    const int key = tile * kB3Tile + tid;
#pragma unroll
    for (int h = 0; h < 4; ++h) o.d[h] = ldf(rd, key * 4, h * rowbytes);
    o.kx = ldf(rx, key * 12, 0); o.ky = ldf(rx, key * 12 + 4, 0); o.kz = ldf(rx, key * 12 + 8, 0);
  };
  auto fetch_box = [&](int it, Box& bx) {  // the two coordinate values per axis of the box (attn_common.h: rpe_box_*)
    const float* vp = P.vertices + (size_t)it * 24;
    bx.x0 = vp[0]; bx.x1 = vp[6]; bx.y0 = vp[1]; bx.y1 = vp[4]; bx.zp = vp[part * 12 + 2];
  };

#ifdef VDETR_B3_PROF
  unsigned long long prof[6] = {0, 0, 0, 0, 0, 0}, pt = __builtin_amdgcn_s_memtime();
#define B3_MARK(i) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); prof[i] += now_ - pt; pt = now_; } while (0)
#else
#define B3_MARK(i) do { } while (0)
#endif
  Ops ops;
  Box box;
  fetch(min(item, items - 1), 0, ops);
  fetch_box(min(item, items - 1), box);
  int drawn = items;
  const int pair = wv >> 1;                                         // this wave's byte of a group counter
  const unsigned long long pair_one = 1ull << (8 * pair);
  const unsigned long long below = pair_one - 1ull;                 // the bytes of the lower wave pairs

  while (item < items) {
    const float X0 = uni(box.x0), X1 = uni(box.x1), Y0 = uni(box.y0), Y1 = uni(box.y1), Zp = uni(box.zp);
    int nitem = items;
    for (int tile = 0; tile < ntiles; ++tile) {
      // ---- 1: taps, signature, group id, stable rank --------------------------------------------------------------------
      const int key = tile * kB3Tile + tid;
      const bool valid = key < P.nK;
      const int nvalid = min(kB3Tile, P.nK - tile * kB3Tile);
      const AxisTap az = rpe_axis(Zp - ops.kz, P);
      const AxisTap ay0 = rpe_axis(Y0 - ops.ky, P), ay1 = rpe_axis(Y1 - ops.ky, P);
      const AxisTap ax0 = rpe_axis(X0 - ops.kx, P), ax1 = rpe_axis(X1 - ops.kx, P);
      const int J = az.base | (ay0.base << 4) | (ay1.base << 8) | (ax0.base << 12) | (ax1.base << 16);
      const int gid = az.base * 289 + (ay0.base + ay1.base) * 17 + ax0.base + ax1.base;
      unsigned long long seen = 0;
      if (valid && !(wv & 1)) seen = __hip_atomic_fetch_add(&cnt[gid], pair_one, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      // U[m] = w_z[cz] w_y[yi][cy], m = 4 cy + 2 cz + yi  (scalar multiplies: see DESIGN.md 4.4b on packed forms)
      const float u0 = az.wa * ay0.wa, u1 = az.wa * ay1.wa, u2 = az.wb * ay0.wa, u3 = az.wb * ay1.wa;
      const float u4 = az.wa * ay0.wb, u5 = az.wa * ay1.wb, u6 = az.wb * ay0.wb, u7 = az.wb * ay1.wb;
      const float d0 = ops.d[0] * fix_scale, d1 = ops.d[1] * fix_scale, d2 = ops.d[2] * fix_scale, d3 = ops.d[3] * fix_scale;
      // The query after this one is drawn while this one's first tile is sorted: one lane, a returning global atomic written
      // as asm so that nothing waits for it here (the compiler's atomic optimiser turns the builtin forms into a wave reduction
      // that reads the result back at once: ~1 us in front of the first barrier of every query).  It is handed to the workgroup in
      // the query's LAST tile, behind a wait that then has nothing left to wait for.
      if (tile == 0 && tid == 0) {
        drawn = items;
        if (taken < cap) asm volatile("global_atomic_add %0, %1, %2, off sc0\n\ts_nop 3" : "=&v"(drawn) : "v"(counter), "v"(1) : "memory");  // (the nop: nobody may overwrite the address / data registers before the instruction has read them)
      }
      b3_barrier();
      if (valid && (wv & 1)) seen = __hip_atomic_fetch_add(&cnt[gid], pair_one, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (tile == ntiles - 1 && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        misc[16] = (unsigned)drawn;
      }
      B3_MARK(0);
      b3_barrier();
      B3_MARK(1);
      // The next unit's operands (the next tile of this query, or the first tile of the next one) are loaded NOW, straight
      // into the registers phase 1 has just finished with: no branch around the loads and no copy behind them (a phi copy made
      // the compiler wait for the loads right here: 1 us per tile).  Past the last unit the loads re-read the last query.
      if (tile == ntiles - 1) { nitem = (int)misc[16]; if (taken < cap) ++taken; }
      {
        const bool more = tile + 1 < ntiles;
        const int uitem = more ? item : min(nitem, items - 1);
        fetch(uitem, more ? tile + 1 : 0, ops);
        fetch_box(uitem, box);
      }
      // ---- 2: group sizes -> first slots ----------------------------------------------------------------------------------
      {
        const bool scans = tid < kB3GidSlots / 3;
        unsigned c0 = 0, c1 = 0, c2 = 0;
        if (scans) {
          const unsigned long long w0 = cnt[3 * tid], w1 = cnt[3 * tid + 1], w2 = cnt[3 * tid + 2];
          c0 = __builtin_amdgcn_sad_u8((unsigned)w0, 0u, __builtin_amdgcn_sad_u8((unsigned)(w0 >> 32), 0u, 0u));
          c1 = __builtin_amdgcn_sad_u8((unsigned)w1, 0u, __builtin_amdgcn_sad_u8((unsigned)(w1 >> 32), 0u, 0u));
          c2 = __builtin_amdgcn_sad_u8((unsigned)w2, 0u, __builtin_amdgcn_sad_u8((unsigned)(w2 >> 32), 0u, 0u));
        }
        const unsigned s = c0 + c1 + c2;
        const unsigned inc = wave_incl_scan_u32(s);
        if (lane == 63) misc[wv] = inc;
        b3_barrier();
        unsigned x = lane < kB3Waves ? misc[lane] : 0u;
        x = lane < wv ? x : 0u;
        x += dpp_u32<kDppQuadXor1>(x);
        x += dpp_u32<kDppQuadXor2>(x);
        x += dpp_u32<kDppRowHalfMirror>(x);
        x += dpp_u32<kDppRowMirror>(x);
        const unsigned base = (unsigned)__builtin_amdgcn_readfirstlane((int)x) + inc - s;
        if (scans) {
          first[3 * tid] = (unsigned short)base; first[3 * tid + 1] = (unsigned short)(base + c0);
          first[3 * tid + 2] = (unsigned short)(base + c0 + c1);
        }
        if (tid == 0) misc[20] = 0u;  // chunk counter of phase 4
      }
      b3_barrier();
      B3_MARK(2);
      // ---- 3: the record goes to its sorted slot ------------------------------------------------------------------------
      if (valid) {
        const unsigned long long fin = cnt[gid] & below;  // keys of the lower wave pairs in this group
        const unsigned before = __builtin_amdgcn_sad_u8((unsigned)fin, 0u, __builtin_amdgcn_sad_u8((unsigned)(fin >> 32), 0u, 0u));
        const unsigned slot = (unsigned)first[gid] + before + ((unsigned)(seen >> (8 * pair)) & 0xFFu);
        char* mine = rec + slot * kB3RecBytes;
        const unsigned sw = (slot & 3u) << 4;
        *reinterpret_cast<b3_rec4_t*>(mine + sw) = f32x4{u0, u1, u2, u3};
        *reinterpret_cast<b3_rec4_t*>(mine + (sw ^ 16u)) = f32x4{u4, u5, u6, u7};
        *reinterpret_cast<b3_rec4_t*>(mine + (sw ^ 32u)) = f32x4{ax0.wa, ax1.wa, ax0.wb, ax1.wb};   // wx[2 cx + xi]
        *reinterpret_cast<b3_rec4_t*>(mine + (sw ^ 48u)) = f32x4{d0, d1, d2, d3};
        sigs[slot] = J;
      }
      b3_barrier();
      B3_MARK(3);
      // ---- 4: waves draw chunks of 32 sorted slots ------------------------------------------------------------------------
      if (tid < kB3GidSlots / 3) { cnt[3 * tid] = 0ull; cnt[3 * tid + 1] = 0ull; cnt[3 * tid + 2] = 0ull; }  // (read for the last time in 3)
      const int nchunks = (nvalid + kB3Chunk - 1) / kB3Chunk;
      for (;;) {
        unsigned cw = 0;
        if (lane == 0) cw = __hip_atomic_fetch_add(&misc[20], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const int c = __builtin_amdgcn_readfirstlane((int)cw);
        if (c >= nchunks) break;
        const int s0 = c * kB3Chunk;
        const int ns = min(kB3Chunk, nvalid - s0);
        const int Jm = lane < ns ? sigs[s0 + lane] : -1;
        const int Jn = lane + 1 < ns ? sigs[s0 + lane + 1] : -2;
        const char* rq = rec + (s0 + kk) * kB3RecBytes;
        float u[kB3Chunk / 4], t[kB3Chunk / 4];
#pragma unroll
        for (int i = 0; i < kB3Chunk / 4; ++i) {  // (slots past the tile's last one hold older records: finite, and never in a segment)
          const float uu = *reinterpret_cast<const b3_rec1_t*>(rq + i * 4 * kB3RecBytes + a_off);
          u[i] = c15 < 8 ? uu : 0.f;  // rows 8..15 of the tile stay zero
          t[i] = *reinterpret_cast<const b3_rec1_t*>(rq + i * 4 * kB3RecBytes + b_offx) *
                 *reinterpret_cast<const b3_rec1_t*>(rq + i * 4 * kB3RecBytes + b_offd);
        }
        const unsigned emask = (unsigned)__ballot(lane < ns && Jm != Jn);  // slot is the last of its group (in this chunk)
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        auto flush = [&](int slot) {  // the group that ends in `slot` of this chunk: 128 sums -> histogram
          const int Jg = __builtin_amdgcn_readlane(Jm, slot);
          const int zrow = (Jg & 15) * (T * T);
          const int yb0 = ((Jg >> 4) & 15) * T, yb1 = ((Jg >> 8) & 15) * T, xb0 = (Jg >> 12) & 15, xb1 = (Jg >> 16) & 15;
          // local vertex of (xi, yi): (0,0) -> 0, (0,1) -> 1, (1,1) -> 2, (1,0) -> 3 (attn_common.h: rpe_box_xi / rpe_box_yi)
          const int b00 = (0 * T3 + zrow + yb0 + xb0) * 16, b01 = (3 * T3 + zrow + yb0 + xb1) * 16;
          const int b10 = (1 * T3 + zrow + yb1 + xb0) * 16, b11 = (2 * T3 + zrow + yb1 + xb1) * 16;
          if (lane < 32) {  // rows 0..7 of the tile; the other lanes hold the zero rows
            char* bin0 = reinterpret_cast<char*>(tab) + (o_xi ? b01 : b00) + o_lane;  // yi = 0
            char* bin1 = reinterpret_cast<char*>(tab) + (o_xi ? b11 : b10) + o_lane;  // yi = 1
            atomicAdd(reinterpret_cast<int*>(bin0), __float2int_rn(acc[0]));                    // r = 0: cz = 0, yi = 0
            atomicAdd(reinterpret_cast<int*>(bin1), __float2int_rn(acc[1]));                    // r = 1: cz = 0, yi = 1
            atomicAdd(reinterpret_cast<int*>(bin0 + T * T * 16), __float2int_rn(acc[2]));       // r = 2: cz = 1, yi = 0
            atomicAdd(reinterpret_cast<int*>(bin1 + T * T * 16), __float2int_rn(acc[3]));       // r = 3: cz = 1, yi = 1
          }
          acc = f32x4{0.f, 0.f, 0.f, 0.f};
        };
#pragma unroll
        for (int i = 0; i < kB3Chunk / 4; ++i) {
          if (4 * i < ns) {
            unsigned eb = (emask >> (4 * i)) & 0xFu;
            if (eb == 0u) {  // the quad lies inside one group (groups average ~10 slots): one matrix instruction, nothing else
              acc = __builtin_amdgcn_mfma_f32_16x16x4f32(u[i], t[i], acc, 0, 0, 0);
            } else if (eb == 8u) {  // ... or ends one with its last slot
              acc = __builtin_amdgcn_mfma_f32_16x16x4f32(u[i], t[i], acc, 0, 0, 0);
              flush(4 * i + 3);
            } else {  // a group ends inside: one instruction per segment, the other slots' rows zeroed
              const int nq = min(4, ns - 4 * i);
              int s = 0;
              do {
                const int e = eb ? __builtin_ctz(eb) : 3;
                const bool on = kk >= s && kk <= e;
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(on ? u[i] : 0.f, t[i], acc, 0, 0, 0);
                if (eb) {
                  flush(4 * i + e);
                  eb &= eb - 1;
                }
                s = e + 1;
              } while (s < nq);
            }
          }
        }
      }
      B3_MARK(4);
      b3_barrier();  // records and counters are free for the next tile
      B3_MARK(5);
    }
    item = nitem;
  }
  __syncthreads();
  float* dst = P.dtable_part + (size_t)blockIdx.x * kB3TableWords;
  for (int i = tid; i < kB3TableWords; i += kB3Threads) dst[i] = (float)tab[i] * fix_inv;
#ifdef VDETR_B3_PROF
  if (blockIdx.x == 2 && lane == 0 && (wv == 0 || wv == 7 || wv == 15))
    printf("box3 prof wg %d wave %d: phase1 %llu  wait1 %llu  scan %llu  scatter %llu  walk %llu  wait5 %llu cycles\n", (int)blockIdx.x, wv,
           prof[0], prof[1], prof[2], prof[3], prof[4], prof[5]);
#endif
}

int launch_attn_bwd_box3(const AttnParams& P, int grid, hipStream_t st) {
  const size_t lds = attn_bwd_box3_lds_bytes();
  if (int e = set_lds(attn_bwd_box3_kernel, lds, "attn_bwd_box3")) return e;
  hipLaunchKernelGGL(attn_bwd_box3_kernel, dim3(grid), dim3(kB3Threads), lds, st, P);
  return check_launch("attn_bwd_box3");
}

}  // namespace vdetr
