// attn_bwd_box4.hip — 3DV-RPE table gradient for axis-aligned boxes from a given dS, fourth design (gfx950; round 4).
//
// Reference math: vdetr_transformer.py:710-731 backward (grid_sampler_3d_backward of the eight per-vertex tables).
// Contract of the round-2 box kernel (attn_bwd_box2.hip, deleted in round 5: `git log`) in its dS-given mode (z-half workgroups that pull queries from a device counter, device gate
// on bwd_aux[4] / [5], int32 fixed-point histogram in LDS, one partial table per workgroup).
//
// What the two earlier designs showed: with the 64 KB histogram a CU holds ONE 16-wave workgroup, a
// SIMD therefore 4 waves, and such a kernel runs at the speed its waves' instruction streams allow.  box2 (wave-private
// 64-key chunks, ballot-loop grouping, padded k-slots, four split-bf16 terms formed on the VALU) spends 1,100 VALU
// instructions per chunk and z-half.  box3 (workgroup-wide counting sort of 1024 keys: 97 groups instead of 16 x 12.7,
// exact fp32 products) halves the VALU work but stands at six workgroup barriers per tile: its waves wait 48 % of the
// time and it ends 5 % ahead of box2.  This kernel keeps box2's independence (a wave owns a chunk of 64 keys, nothing in
// the loop waits for another wave) and takes from box3 what made its instruction streams short:
//   * grouping = ONE returning LDS add per pair: the pair's signature is hashed into 64 wave-private counters, the returned
//     value is its rank in the bucket, a DPP scan of the 64 counts gives the buckets' first slots.  Pairs of one group are
//     then consecutive slots (a bucket may hold two groups: that costs a flush, never a wrong cell, because group ends are
//     found by comparing the signatures of neighbouring slots).  box2's ballot / mbcnt loop was ~150 instructions per chunk.
//   * nothing is padded: a quad of slots is one v_mfma_f32_16x16x4_f32 (rows: the 8 products w_z w_y, columns: the 16
//     products w_x dS, K = 4 pairs), a quad that holds a group end is issued once per segment with the other slots' rows zeroed
//   * exact fp32 products (no bf16 split: one multiply per operand and quad), operands of the chunk's 16 quads read from the
//     wave's record strip in one burst (3 LDS reads per lane and quad, 16-byte parts XOR-swizzled by the slot)
//   * a group's 128 sums leave with four ds_add_u32 per lane whose addresses come from a signature that is stored as the
//     bins' cell numbers (two extracts, two adds), rounded by v_cvt_rpi_i32_f32
// The rank a lane reads back from its LDS add follows the lane order inside the instruction, so the slot order — and with
// it the order of the float additions inside the matrix unit — is the same in every run: results are bit-reproducible.
#include "attn_common.h"

#include <stdlib.h>

namespace vdetr {

typedef f32x4 __attribute__((may_alias)) b4_rec4_t;
typedef float __attribute__((may_alias)) b4_rec1_t;
typedef int __attribute__((may_alias)) b4_reci_t;
typedef unsigned __attribute__((may_alias)) b4_recu_t;

#ifndef VDETR_B4_WAVES
#define VDETR_B4_WAVES 16
#endif
constexpr int kB4Waves = VDETR_B4_WAVES;  // (-DVDETR_B4_WAVES=8: half a CU's registers per workgroup - an experiment)
constexpr int kB4Threads = kB4Waves * kWave;
constexpr int kB4T = 10;                                     // table edge ("bilinear_4_10")
constexpr int kB4RecBytes = 64;                              // U[8] = wz wy | wx[4] | dS[4]; 16-byte parts at 16 (p ^ (slot & 3))
constexpr int kB4StripBytes = kWave * kB4RecBytes + (kWave + 4) * 4 + 2 * kWave * 4;  // records | signatures (+ pad) | counters | first slots
constexpr int kB4TableWords = 4 * kB4T * kB4T * kB4T * 4;

size_t attn_bwd_box4_lds_bytes() { return (size_t)kB4TableWords * 4 + (size_t)kB4Waves * kB4StripBytes + 64; }

__device__ __forceinline__ unsigned b4_wave_incl_scan(unsigned v) {  // Hillis-Steele in the rows, row totals by row_bcast
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, kDppRowBcast15, 0xA, 0xF, false);
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, kDppRowBcast31, 0xC, 0xF, false);
  return v;
}
__device__ __forceinline__ int b4_round(float x) {  // floor(x + 0.5): one instruction (rndne + cvt are two)
  int r;
  asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(x));
  return r;
}

__global__ __launch_bounds__(kB4Threads)
#if VDETR_B4_WAVES < 16
__attribute__((amdgpu_waves_per_eu(4, 4)))  // <= 128 registers per lane also with fewer waves: other kernels fit next to it
#endif
void attn_bwd_box4_kernel(AttnParams P) {
  constexpr int T = kB4T, T3 = T * T * T;
  if (P.bwd_aux[4] != 0 || P.bwd_aux[5] == 0) return;  // a query is not an axis-aligned box: the general kernel runs instead
  extern __shared__ __attribute__((aligned(16))) float smem[];
  char* lds = reinterpret_cast<char*>(smem);
  int* tab = reinterpret_cast<int*>(lds);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  char* rec = lds + (size_t)kB4TableWords * 4 + (size_t)wv * kB4StripBytes;     // [64] records
  char* sigs = rec + kWave * kB4RecBytes;                                       // [64 + 4] signatures of the sorted slots
  char* wcnt = sigs + (kWave + 4) * 4;                                          // [64] bucket counters
  char* wfirst = wcnt + kWave * 4;                                              // [64] first slot of a bucket
  unsigned* misc = reinterpret_cast<unsigned*>(lds + (size_t)kB4TableWords * 4 + (size_t)kB4Waves * kB4StripBytes);
  const int part = blockIdx.x & 1, nwg = gridDim.x >> 1;
  const int items = P.B * P.nQ;
  for (int i = tid; i < kB4TableWords; i += kB4Threads) tab[i] = 0;
  for (int i = lane; i < kB4StripBytes / 4; i += kWave) reinterpret_cast<b4_recu_t*>(rec)[i] = 0u;  // finite records, zero counters
  const int per_wg = (items + nwg - 1) / nwg;
  const int cap = bwd_query_cap(per_wg);
  float fix_scale = 1.f, fix_inv = 1.f;
  {  // the bound of the round-2 box kernel: |bin sum| <= queries of this workgroup x 2 drop_scale max|dO row| max|V row|
    const float dmax = sqrtf(__uint_as_float(P.bwd_aux[0]) * __uint_as_float(P.bwd_aux[1]));
    const float bound = 2.f * P.drop_scale * dmax * (float)cap;
    if (bound > 0.f && bound < INFINITY) {
      const int e = 30 - (int)ceilf(__log2f(bound) + 1e-3f);  // 2^e * bound <= 2^30
      fix_scale = ldexpf(1.f, e);
      fix_inv = ldexpf(1.f, -e);
    }
  }
  unsigned* counter = const_cast<unsigned*>(P.bwd_aux) + 2 + part;
  if (tid == 0) misc[0] = atomicAdd(counter, 1u);
  __syncthreads();
  int item = (int)misc[0];
  int taken = 1;

  const int kk = lane >> 4, c15 = lane & 15;
  // A role (row m = c15 of the matrix instruction): U[m], m = (cy, cz, yi), for m < 8; rows 8..15 stay zero.  Part p of slot s
  // sits at 16 (p ^ (s & 3)); a lane's slots are 4 i + kk, so s & 3 = kk.
  const int a_off = (((c15 >> 2) & 1) ^ kk) * 16 + (c15 & 3) * 4;
  // B role (column n = c15 = (xi, cx, h)): wx[2 cx + xi] in part 2, dS[h] in part 3
  const int b_offx = (2 ^ kk) * 16 + (2 * ((c15 >> 2) & 1) + (c15 >> 3)) * 4, b_offd = (3 ^ kk) * 16 + (c15 & 3) * 4;
  // output role: lane (kk, n) register r holds row 4 kk + r = (cy = kk, cz = r >> 1, yi = r & 1), column n = (xi, cx, h).
  // Local vertex of (xi, yi): (0,0) -> 0, (0,1) -> 1, (1,1) -> 2, (1,0) -> 3 (attn_common.h: rpe_box_xi / rpe_box_yi).
  const int o_xi = c15 >> 3;
  const int o_lane = (((kk & 1) * T + ((c15 >> 2) & 1)) * 4 + (c15 & 3)) * 4;  // bytes of (cy, cx, h) inside a cell block
  const int o_c0 = (o_xi ? 3 : 0) * T3 * 16 + o_lane, o_c1 = (o_xi ? 2 : 1) * T3 * 16 + o_lane;  // yi = 0 / 1
  const int o_xshift = 20 + 4 * o_xi;

  using rsrc_t = __amdgpu_buffer_rsrc_t;
  auto make_rsrc = [](const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
  };
  auto ldf = [](rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
  };
  auto uni = [](float x) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x))); };
  const int rowbytes = P.nK * 4;
  const bool rot = P.cos_sin != nullptr;
  const int nchunks = (P.nK + kWave - 1) / kWave;
  struct Ops { float d[4], kx, ky, kz; };
  // dS of the 4 heads + the key's position.  Of a key past nK only the POSITION reads 0 (its resource ends with the keys); the dS
  // resource spans the 4 head rows, so d[h] of such a lane holds the next head's row for h < 3: lanes past the end never write a
  // record or bump a counter below, and nothing may use their d[]
  auto fetch = [&](rsrc_t rd, rsrc_t rx, int chunk, Ops& o) {
    const int key = chunk * kWave + lane;
#pragma unroll
    for (int h = 0; h < 4; ++h)
      o.d[h] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rd, key * 4, h * rowbytes, kStreamAux));  // dS: read once
    o.kx = ldf(rx, key * 12, 0); o.ky = ldf(rx, key * 12 + 4, 0); o.kz = ldf(rx, key * 12 + 8, 0);
  };

  int slot_of_next = 1;  // misc[1] / misc[2] alternate as the hand-over word of the next query
  while (item < items) {
    int drawn = items;  // the query after this one: drawn now by one lane (asm: nothing waits for it), handed over below
    if (tid == 0 && taken < cap) asm volatile("global_atomic_add %0, %1, %2, off sc0\n\ts_nop 3" : "=&v"(drawn) : "v"(counter), "v"(1) : "memory");  // (the nop: nobody may overwrite the address / data registers before the instruction has read them)
    const int b = item / P.nQ;
    const rsrc_t rd = make_rsrc(P.dprob + (size_t)item * 4 * P.nK, 4u * rowbytes);
    const rsrc_t rx = make_rsrc(P.xyz + (size_t)b * P.nK * 3, 3u * rowbytes);
    Ops ops;
    fetch(rd, rx, wv, ops);  // (first: the box's coordinates below are one more round trip, not two, behind the hand-over)
    const float* vp = P.vertices + (size_t)item * 24;
    const float bx0 = vp[0], bx1 = vp[6], by0 = vp[1], by1 = vp[4], bz = vp[part * 12 + 2];
    float X0 = uni(bx0), X1 = uni(bx1), Y0 = uni(by0), Y1 = uni(by1), Zp = uni(bz);
    // rotated boxes (angle_type "object_coords", the cos / sin operand): the look-up frame is turned by the query's angle, where
    // the corners are an axis-aligned box again (attn_common.h: attn_delta_body checks that).  A pair's offset is then
    // R (P_0 - X) + (xi EX, yi EY, zi EZ): X0 / Y0 hold P_0, X1 / Y1 the edge lengths EX / EY in the turned frame.
    float rc = 1.f, rs = 0.f;
    if (rot) {
      rc = uni(P.cos_sin[(size_t)item * 2]); rs = uni(P.cos_sin[(size_t)item * 2 + 1]);
      float ex = uni(vp[9]) - X0, ey = uni(vp[10]) - Y0;     // vertex 3: xi = 1, yi = 0
      rpe_rotate(ex, ey, rc, rs);
      float fx = uni(vp[3]) - X0, fy = uni(vp[4]) - Y0;      // vertex 1: xi = 0, yi = 1
      rpe_rotate(fx, fy, rc, rs);
      X1 = ex; Y1 = fy;  // (z is not turned: Zp stays the z of this half's vertices)
    }
    for (int chunk = wv; chunk < nchunks; chunk += kB4Waves) {
      // ---- taps, signature, products of this lane's pair ---------------------------------------------------------------------
      const int ns = min(kWave, P.nK - chunk * kWave);   // valid pairs of this chunk: the low lanes
      const bool valid = lane < ns;
      float dx0 = X0 - ops.kx, dy0 = Y0 - ops.ky, dx1 = X1 - ops.kx, dy1 = Y1 - ops.ky;
      if (rot) {
        rpe_rotate(dx0, dy0, rc, rs);
        dx1 = dx0 + X1; dy1 = dy0 + Y1;
      }
      const AxisTap az = rpe_axis(Zp - ops.kz, P);
      const AxisTap ay0 = rpe_axis(dy0, P), ay1 = rpe_axis(dy1, P);
      const AxisTap ax0 = rpe_axis(dx0, P), ax1 = rpe_axis(dx1, P);
      // signature = the cell numbers the flush needs: (z T + y0) T | (z T + y1) T << 10 | x0 << 20 | x1 << 24
      const int zrow = az.base * (T * T);
      const int J = (zrow + ay0.base * T) | ((zrow + ay1.base * T) << 10) | (ax0.base << 20) | (ax1.base << 24);
      const unsigned hsh = ((unsigned)J * 0x9E3779B1u) >> 26;
      // U[m] = w_z[cz] w_y[yi][cy], m = 4 cy + 2 cz + yi  (scalar multiplies: see DESIGN.md 4.4b on packed forms)
      const float u0 = az.wa * ay0.wa, u1 = az.wa * ay1.wa, u2 = az.wb * ay0.wa, u3 = az.wb * ay1.wa;
      const float u4 = az.wa * ay0.wb, u5 = az.wa * ay1.wb, u6 = az.wb * ay0.wb, u7 = az.wb * ay1.wb;
      const float d0 = ops.d[0] * fix_scale, d1 = ops.d[1] * fix_scale, d2 = ops.d[2] * fix_scale, d3 = ops.d[3] * fix_scale;
      if (chunk + kB4Waves < nchunks) fetch(rd, rx, chunk + kB4Waves, ops);  // (phase order: the registers are free now)
      // ---- sort the chunk's pairs by signature bucket ---------------------------------------------------------------------------
      __builtin_amdgcn_wave_barrier();
      unsigned rank = 0;
      if (valid) rank = __hip_atomic_fetch_add(reinterpret_cast<b4_recu_t*>(wcnt) + hsh, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      __builtin_amdgcn_wave_barrier();
      const unsigned cnt = reinterpret_cast<const b4_recu_t*>(wcnt)[lane];
      const unsigned incl = b4_wave_incl_scan(cnt);
      reinterpret_cast<b4_recu_t*>(wfirst)[lane] = incl - cnt;
      reinterpret_cast<b4_recu_t*>(wcnt)[lane] = 0u;
      __builtin_amdgcn_wave_barrier();
      if (valid) {
        const unsigned slot = reinterpret_cast<const b4_recu_t*>(wfirst)[hsh] + rank;
        char* mine = rec + slot * kB4RecBytes;
        const unsigned sw = (slot & 3u) << 4;
        *reinterpret_cast<b4_rec4_t*>(mine + sw) = f32x4{u0, u1, u2, u3};
        *reinterpret_cast<b4_rec4_t*>(mine + (sw ^ 16u)) = f32x4{u4, u5, u6, u7};
        *reinterpret_cast<b4_rec4_t*>(mine + (sw ^ 32u)) = f32x4{ax0.wa, ax1.wa, ax0.wb, ax1.wb};   // wx[2 cx + xi]
        *reinterpret_cast<b4_rec4_t*>(mine + (sw ^ 48u)) = f32x4{d0, d1, d2, d3};
        reinterpret_cast<b4_reci_t*>(sigs)[slot] = J;
      }
      __builtin_amdgcn_wave_barrier();
      // ---- walk the sorted slots -------------------------------------------------------------------------------------------------
      const int Jm = valid ? reinterpret_cast<const b4_reci_t*>(sigs)[lane] : -1;
      const int Jn = lane + 1 < ns ? reinterpret_cast<const b4_reci_t*>(sigs)[lane + 1] : -2;
      const char* rq = rec + kk * kB4RecBytes;
      float u[kWave / 4], t[kWave / 4];
#pragma unroll
      for (int i = 0; i < kWave / 4; ++i) {
        // (Slots past the chunk's last pair hold older records: finite, and never inside a segment.  Rows 8..15 of the tile read
        // U[m & 7] like rows 0..7: a row of the result depends on its own row of A only, and rows 8..15 are never flushed.)
        u[i] = *reinterpret_cast<const b4_rec1_t*>(rq + i * 4 * kB4RecBytes + a_off);
        t[i] = *reinterpret_cast<const b4_rec1_t*>(rq + i * 4 * kB4RecBytes + b_offx) *
               *reinterpret_cast<const b4_rec1_t*>(rq + i * 4 * kB4RecBytes + b_offd);
      }
      const unsigned long long emask = __ballot(valid && Jm != Jn);  // slot is the last of its group
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      auto flush = [&](int slot) {  // the group that ends in `slot`: 128 sums -> histogram
        const int Jg = __builtin_amdgcn_readlane(Jm, slot);
        const int cell0 = Jg & 1023, cell1 = (Jg >> 10) & 1023;  // (z T + y) T of yi = 0 / 1
        if (lane < 32) {  // rows 0..7 of the tile
          const int xb = (int)__builtin_amdgcn_ubfe((unsigned)Jg, (unsigned)o_xshift, 4u);
          char* bin0 = reinterpret_cast<char*>(tab) + ((cell0 + xb) << 4) + o_c0;
          char* bin1 = reinterpret_cast<char*>(tab) + ((cell1 + xb) << 4) + o_c1;
          atomicAdd(reinterpret_cast<int*>(bin0), b4_round(acc[0]));                    // r = 0: cz = 0, yi = 0
          atomicAdd(reinterpret_cast<int*>(bin1), b4_round(acc[1]));                    // r = 1: cz = 0, yi = 1
          atomicAdd(reinterpret_cast<int*>(bin0 + T * T * 16), b4_round(acc[2]));       // r = 2: cz = 1, yi = 0
          atomicAdd(reinterpret_cast<int*>(bin1 + T * T * 16), b4_round(acc[3]));       // r = 3: cz = 1, yi = 1
        }
        acc = f32x4{0.f, 0.f, 0.f, 0.f};
      };
      // (Measured alternatives of this walk, all bit-identical: a rolled loop over the quads with the operands fetched one quad
      // ahead, 330 us; the same conditions handed to the compiler as exec-masked regions — which keeps the accumulator in one
      // register quad, where scalar branches make it copy the four registers behind the wait states of a matrix result at every
      // merge —, 319 us; the matrix instruction as asm with the accumulator tied in place: the compiler still copies it right
      // behind the asm, where nothing pads the hazard.  The form below is the one with the fewest instructions: 299 us.)
      if (ns == kWave) {  // a full chunk (every chunk when nK is a multiple of 64): no bounds in the walk
#pragma unroll
        for (int i = 0; i < kWave / 4; ++i) {
          const unsigned eb = (unsigned)(emask >> (4 * i)) & 0xFu;
          if (eb == 0u) {  // the quad lies inside one group: one matrix instruction, nothing else
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(u[i], t[i], acc, 0, 0, 0);
          } else if (eb == 8u) {  // ... or ends one with its last slot
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(u[i], t[i], acc, 0, 0, 0);
            flush(4 * i + 3);
          } else {
            // a group ends inside the quad (slot e0 < 3): its slots' rows first, flush, then the rows of the slots behind it —
            // which belong to the next group(s).  One more end at slot 3 or none: straight-line; two ends inside: the loop.
            const int e0 = __builtin_ctz(eb);
            const bool head = kk <= e0;
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(head ? u[i] : 0.f, t[i], acc, 0, 0, 0);
            flush(4 * i + e0);
            unsigned rest = eb & (eb - 1u);
            if ((rest & 7u) == 0u) {
              acc = __builtin_amdgcn_mfma_f32_16x16x4f32(head ? 0.f : u[i], t[i], acc, 0, 0, 0);
              if (rest) flush(4 * i + 3);
            } else {
              int s = e0 + 1;
              do {
                const int e = rest ? __builtin_ctz(rest) : 3;
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kk >= s && kk <= e ? u[i] : 0.f, t[i], acc, 0, 0, 0);
                if (rest) {
                  flush(4 * i + e);
                  rest &= rest - 1u;
                }
                s = e + 1;
              } while (s < 4);
            }
          }
        }
      } else {  // the last, partly filled chunk of a key count that is not a multiple of 64
        for (int i = 0; 4 * i < ns; ++i) {
          unsigned eb = (unsigned)(emask >> (4 * i)) & 0xFu;
          const int nq = min(4, ns - 4 * i);
          // (u / t of quad i: re-read, this loop is not unrolled)
          const float ui = *reinterpret_cast<const b4_rec1_t*>(rq + i * 4 * kB4RecBytes + a_off);
          const float ti = *reinterpret_cast<const b4_rec1_t*>(rq + i * 4 * kB4RecBytes + b_offx) *
                           *reinterpret_cast<const b4_rec1_t*>(rq + i * 4 * kB4RecBytes + b_offd);
          int s = 0;
          do {
            const int e = eb ? __builtin_ctz(eb) : 3;
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kk >= s && kk <= e ? ui : 0.f, ti, acc, 0, 0, 0);
            if (eb) {
              flush(4 * i + e);
              eb &= eb - 1u;
            }
            s = e + 1;
          } while (s < nq);
        }
      }
    }
    // ---- hand the next query over (one workgroup barrier per query; it waits for LDS only) -----------------------------------------
    if (tid == 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      misc[slot_of_next] = (unsigned)drawn;
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    item = (int)misc[slot_of_next];
    slot_of_next ^= 3;  // 1 <-> 2
    if (taken < cap) ++taken;
  }
  __syncthreads();
  float* dst = P.dtable_part + (size_t)blockIdx.x * kB4TableWords;
  for (int i = tid; i < kB4TableWords; i += kB4Threads) dst[i] = (float)tab[i] * fix_inv;
}

int launch_attn_bwd_box4(const AttnParams& P, int grid, hipStream_t st) {
  const size_t lds = attn_bwd_box4_lds_bytes();
  if (int e = set_lds(attn_bwd_box4_kernel, lds, "attn_bwd_box4")) return e;
  hipLaunchKernelGGL(attn_bwd_box4_kernel, dim3(grid), dim3(kB4Threads), lds, st, P);
  return check_launch("attn_bwd_box4");
}

}  // namespace vdetr
