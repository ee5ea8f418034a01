// fps_rows.hip — furthest point sampling, the fast path of vdetr_furthest_point_sampling(_varlen)_f32.
// Bit-exact with the reference's result (third_party/pointnet2/_ext_src/src/sampling_gpu.cu:73-176), like fps.hip,
// and built on the same exact box-skip argument (see fps.hip's header); what differs is how a round's dependent
// chain is laid out.  What the measurements behind it say (tools/probes/lat_probe.hip, tools/fps_variants.py, one CU):
//   * a wave retires one instruction of this kind of code every ~8-10 cycles, dependent or not, and every TAKEN
//     branch costs a refetch: the round is an instruction-count and control-flow problem, not a bandwidth one
//     (an L2-resident 1 KB bucket fetch is ~300 cycles, an LDS round trip ~60, s_barrier ~25);
//   * more waves help as long as they do different buckets (16 waves beat 8 and 4 by 1.5x / 2x), but whatever all
//     waves repeat after the barrier is slowed down by the SIMD they share;
//   * variants that lost: 16/32-point buckets with several buckets per wave pass (more box tests than they save),
//     a shared survivor list that hands every wave one bucket per round (3 more barriers: 6.2 ms against 4.8),
//     the last wave to arrive decoding alone (+8 %), runtime-selected code paths in the loop (+13 %).
//
// One workgroup of W waves per scene.  The cloud is counting-sorted into Z-order over 2^15 near-cubic cells (the
// split sequence follows the cloud's aspect ratio) and cut into BUCKETS of 64 consecutive points.  Bucket g belongs
// to wave g % W for good: its bounding box, current max running distance (as an order-preserving rank) and the tie
// key of that max live in the registers of an owner lane of that wave, the coordinates of the max in LDS (cand[g]).
//
// Round j, per wave, ONE workgroup barrier:
//   test    every owner lane: is the new sample closer to my box than my bucket's max?  (14 VALU per slot)
//   batch   the wave's surviving buckets, 1, 2 or 4 at a time (the count is known from the ballots, the common case
//           of one bucket is the fall-through path): 16-B load per lane, distance, min, store, wave max of the rank as
//           a scalar (6 DPP stages + v_readlane); the tie keys are reduced only if the maximum is not unique; the
//           winner lane refreshes cand[g], the owner lane gets (rank, key) by v_writelane.
//   reduce  arg-max over the owner lanes, ONLY if the wave's best bucket was one of those processed: running
//           distances never grow, so nothing else can invalidate it
//   barrier LDS only: the running-distance stores are re-read by the same wave, nobody else needs them
//   decode  every 16-lane row all-reduces the W entries, v_readlane of the winner's coordinates; the sample is
//           recorded as its tie key and translated to the point index after the last round.
#include "fps.h"

#include <stdlib.h>
#include <type_traits>

namespace vdetr {

constexpr int kRowsCellBits = 14;
constexpr int kRowsCells = 1 << kRowsCellBits;
constexpr int kRowsHistWords = kRowsCells + (kRowsCells >> 5) + 4;  // one pad word per 32 cells + the end sentinel
__device__ __forceinline__ int hidx(int c) { return c + (c >> 5); }

template <int BP>
__device__ __forceinline__ unsigned grp_allmax_u32(unsigned v) {
  v = row_allmax_u32_fx(v);
  if (BP >= 32) { const pair_u32 p = xrow16(v); v = max(p.a, p.b); }
  if (BP >= 64) { const pair_u32 p = xhalf32(v); v = max(p.a, p.b); }
  return v;
}
template <int BP>
__device__ __forceinline__ unsigned grp_allmin_u32(unsigned v) {
  v = row_allmin_u32_fx(v);
  if (BP >= 32) { const pair_u32 p = xrow16(v); v = min(p.a, p.b); }
  if (BP >= 64) { const pair_u32 p = xhalf32(v); v = min(p.a, p.b); }
  return v;
}
template <int BP>
__device__ __forceinline__ float grp_allmax_f32(float v) {
  v = row_allmax_f32(v);
  if (BP >= 32) { const pair_u32 p = xrow16(__float_as_uint(v)); v = fmaxf(__uint_as_float(p.a), __uint_as_float(p.b)); }
  if (BP >= 64) { const pair_u32 p = xhalf32(__float_as_uint(v)); v = fmaxf(__uint_as_float(p.a), __uint_as_float(p.b)); }
  return v;
}
template <int BP>
__device__ __forceinline__ float grp_allmin_f32(float v) { return -grp_allmax_f32<BP>(-v); }

// deposits the low bits of c at the set bits of m, lowest first (m is wave-uniform: a scalar loop)
__device__ __forceinline__ unsigned deposit_bits(unsigned c, unsigned m) {
  unsigned r = 0;
  while (m) {
    const unsigned low = m & (0u - m);
    if (c & 1u) r |= low;
    c >>= 1;
    m ^= low;
  }
  return r;
}

__device__ unsigned long long g_rows_cyc[16][8];
__device__ __forceinline__ unsigned long long rows_clock() {
  __builtin_amdgcn_sched_barrier(0);
  const unsigned long long t = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}

constexpr int kBP = 64;  // points per bucket: one wave-wide load
constexpr int kKB = 4;   // buckets in flight per wave

// W waves per scene, at most NS buckets per owner lane (compile-time bound of the slot loops)
template <int W, int NS, bool DEBUG>
__global__ __launch_bounds__(W * kWave) void fps_rows_kernel(RowsParams Pin) {
  constexpr int T = W * kWave;
  extern __shared__ __align__(16) unsigned char smem[];
  int* const s_hist = reinterpret_cast<int*>(smem);        // prologue: cell starts, then per-leaf fill counters
  int* const s_leaf = s_hist + kRowsHistWords;             // prologue: slot offset and first bucket of the cell's tree leaf
  float4* const s_cand = reinterpret_cast<float4*>(smem);  // rounds: (x,y,z,key) of every bucket's max
  __shared__ int s_wsum[W];
  __shared__ float s_red[W][6];
  __shared__ unsigned s_xr[2][16];
  __shared__ float4 s_xc[2][16];

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const RowsScene S = Pin.scenes[blockIdx.x];
  const float* __restrict__ xyz = S.xyz;
  int32_t* __restrict__ out = S.idx;
  float4* __restrict__ pts = Pin.pts + S.ws_off;
  uint32_t* __restrict__ keys = Pin.keys + S.ws_off;
  const int n = S.n, m = Pin.m;
  int nb = 0;  // buckets of this scene: decided in prologue 2
  const unsigned rb = (unsigned)S.ref_block;

  // ---- prologue 1: bounding box of the cloud, the split sequence of the cell grid ------------------------------
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int k = tid; k < n; k += T) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float v = xyz[k * 3 + a];
      lo[a] = fminf(lo[a], v);
      hi[a] = fmaxf(hi[a], v);
    }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    lo[a] = wave_allmin_f32(lo[a]);
    hi[a] = wave_allmax_f32(hi[a]);
  }
  if (lane == 0) {
#pragma unroll
    for (int a = 0; a < 3; ++a) { s_red[w][a] = lo[a]; s_red[w][3 + a] = hi[a]; }
  }
  for (int c = tid; c < kRowsHistWords; c += T) s_hist[c] = 0;
  __syncthreads();
  float scale[3], cmaxf[3];
  unsigned dep[3] = {0u, 0u, 0u};
  {
    float ext[3];
    int bits[3] = {0, 0, 0};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      float l = s_red[0][a], h = s_red[0][3 + a];
      for (int ww = 1; ww < W; ++ww) { l = fminf(l, s_red[ww][a]); h = fmaxf(h, s_red[ww][3 + a]); }
      lo[a] = l;
      const float e = h - l;
      ext[a] = (e > 0.f && e < INFINITY) ? e : 0.f;
    }
    float cell[3] = {ext[0], ext[1], ext[2]};
    for (int i = 0; i < kRowsCellBits; ++i) {  // halve the longest cell edge: near-cubic cells whatever the aspect ratio
      const int a = (cell[0] >= cell[1] && cell[0] >= cell[2]) ? 0 : (cell[1] >= cell[2] ? 1 : 2);
#pragma unroll
      for (int q = 0; q < 3; ++q)
        if (q == a) { dep[q] |= 1u << (kRowsCellBits - 1 - i); ++bits[q]; cell[q] *= 0.5f; }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      scale[a] = ext[a] > 0.f ? (float)(1 << bits[a]) / ext[a] : 0.f;
      cmaxf[a] = (float)((1 << bits[a]) - 1);
    }
  }
  auto cell_of = [&](float x, float y, float z) -> int {
    // NaN / inf coordinates fall into cell 0 of their axis (the cast of NaN is made harmless by the integer clamp)
    int cx = (int)fminf(fmaxf((x - lo[0]) * scale[0], 0.f), cmaxf[0]);
    int cy = (int)fminf(fmaxf((y - lo[1]) * scale[1], 0.f), cmaxf[1]);
    int cz = (int)fminf(fmaxf((z - lo[2]) * scale[2], 0.f), cmaxf[2]);
    cx = min(max(cx, 0), (int)cmaxf[0]); cy = min(max(cy, 0), (int)cmaxf[1]); cz = min(max(cz, 0), (int)cmaxf[2]);
    return (int)(deposit_bits((unsigned)cx, dep[0]) | deposit_bits((unsigned)cy, dep[1]) | deposit_bits((unsigned)cz, dep[2]));
  };

  // ---- prologue 2: histogram, exclusive scan, scatter ------------------------------------------------------------
  for (int k = tid; k < n; k += T) atomicAdd(&s_hist[hidx(cell_of(xyz[k * 3], xyz[k * 3 + 1], xyz[k * 3 + 2]))], 1);
  __syncthreads();
  constexpr int kPer = kRowsCells / T;  // consecutive cells per thread
  __shared__ int s_total;
  {
    int sum = 0;
    for (int i = 0; i < kPer; ++i) sum += s_hist[hidx(tid * kPer + i)];
    int incl = sum;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
      const int o = __shfl_up(incl, off);
      if (lane >= off) incl += o;
    }
    if (lane == kWave - 1) s_wsum[w] = incl;
    __syncthreads();
    int base = 0;
    for (int ww = 0; ww < w; ++ww) base += s_wsum[ww];
    int run = base + incl - sum;
    for (int i = 0; i < kPer; ++i) {
      const int v = s_hist[hidx(tid * kPer + i)];
      s_hist[hidx(tid * kPer + i)] = run;
      run += v;
    }
    if (tid == T - 1) s_hist[hidx(kRowsCells)] = run;  // = n: the end sentinel of the prefix
  }
  __syncthreads();
  // ---- buckets = leaves of the Z-curve's binary tree ------------------------------------------------------------------
  // A run of 64 consecutive sorted points straddles octant boundaries of the curve at random, and its box is then far
  // larger than its points' share of space; a tree node is an aligned box.  On the host model (tools/fps_model.py, 40k
  // points): 8.5 instead of 14.0 surviving buckets per round, 1.5 instead of 2.3 on the busiest wave.
  // The leaf of cell c is the shallowest node [c & ~(size-1), +size), size = 2^(bits-d), that holds <= 64 K points; it
  // takes ceil(count / 64) buckets, filled in cell order.  K = 1 costs ~1.6x the buckets of plain runs; the first of
  // K = 1, 2, 4, 8 whose buckets fit the owner lanes' slots (cap_buckets) is used, plain runs (K = inf) otherwise.
  // s_leaf[c] = (first slot of the leaf - first sorted position of the leaf) << 13 | first bucket of the leaf (at its head)
  auto head_of = [&](int c, int limit, int& cnt) -> int {
    int dlo = 0, dhi = kRowsCellBits;
    while (dlo < dhi) {
      const int mid = (dlo + dhi) >> 1, size = 1 << (kRowsCellBits - mid), st = c & ~(size - 1);
      if (s_hist[hidx(st + size)] - s_hist[hidx(st)] <= limit) dhi = mid; else dlo = mid + 1;
    }
    const int size = 1 << (kRowsCellBits - dlo), head = c & ~(size - 1);
    cnt = s_hist[hidx(head + size)] - s_hist[hidx(head)];
    return head;
  };
  bool tree = false;
  int limit = kBP;
  for (int ks = 0; ks < 4 && S.cap_buckets > 0; ++ks, limit *= 2) {
    int sum = 0;
    for (int i = 0; i < kPer; ++i) {
      const int c = tid * kPer + i;
      int cnt;
      const int head = head_of(c, limit, cnt);
      const int nbk = c == head ? (cnt + kBP - 1) / kBP : 0;
      s_leaf[c] = nbk;
      sum += nbk;
    }
    int incl = sum;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
      const int o = __shfl_up(incl, off);
      if (lane >= off) incl += o;
    }
    __syncthreads();  // s_wsum and s_total of the pass before have been read by everybody
    if (lane == kWave - 1) s_wsum[w] = incl;
    __syncthreads();
    int base = 0;
    for (int ww = 0; ww < w; ++ww) base += s_wsum[ww];
    int run = base + incl - sum;
    for (int i = 0; i < kPer; ++i) {
      const int c = tid * kPer + i;
      const int v = s_leaf[c];
      s_leaf[c] = min(run, 8191);
      run += v;
    }
    if (tid == T - 1) s_total = run;
    __syncthreads();
    if (s_total <= S.cap_buckets) { tree = true; break; }
  }
  const float p0x = xyz[0], p0y = xyz[1], p0z = xyz[2];
  nb = tree ? s_total : (n + kBP - 1) / kBP;
  const int npad = nb * kBP;
  if (tree) {
    for (int i = 0; i < kPer; ++i) {
      const int c = tid * kPer + i;
      int cnt;
      const int head = head_of(c, limit, cnt);
      const int delta = (s_leaf[head] & 8191) * kBP - s_hist[hidx(head)];  // >= 0: the leaves before hold all the points before
      s_leaf[c] |= (int)((unsigned)delta << 13);  // (readers of this word want its low 13 bits only: unchanged)
    }
    // every slot starts as padding (rank 0: never a candidate)
    for (int k = tid; k < npad; k += T) {
      pts[k] = make_float4(p0x, p0y, p0z, -INFINITY);
      keys[k] = 0xFFFFFFFFu;
    }
  } else {
    for (int c = tid; c < kRowsCells; c += T) s_leaf[c] = 0;
    for (int k = n + tid; k < npad; k += T) {
      pts[k] = make_float4(p0x, p0y, p0z, -INFINITY);
      keys[k] = 0xFFFFFFFFu;
    }
  }
  __syncthreads();
  for (int k = tid; k < n; k += T) {
    const float x = xyz[k * 3], y = xyz[k * 3 + 1], z = xyz[k * 3 + 2];
    const int cell = cell_of(x, y, z);
    const int pos = atomicAdd(&s_hist[hidx(cell)], 1) + (int)((unsigned)s_leaf[cell] >> 13);
    // origin-skip rule: `if (mag <= 1e-3) continue;` compares the float mag against a DOUBLE literal
    // (sampling_gpu.cu:103-104); mag in the same contraction order as the distance.
    const float mag = sqdist3(x, y, z);
    const bool skip = (double)mag <= 1e-3;
    pts[pos] = make_float4(x, y, z, skip ? -INFINITY : 1e10f);
    keys[pos] = fps_tie_key((unsigned)k, rb, S.ref_log2);
  }
  __syncthreads();  // the sorted cloud is visible to every wave; s_hist is dead, s_cand may be written

  // ---- prologue 3: bucket boxes into owner-lane registers, cand[] ----------------------------------------------
  // bucket g: wave g % W, owner lane (g / W) % 64, slot g / (64 W)
  const int nslots = (nb + W * kWave - 1) / (W * kWave);
  float blo[NS][3], bhi[NS][3];
  unsigned brank[NS], bkey[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    brank[s] = 0u; bkey[s] = 0xFFFFFFFFu;
#pragma unroll
    for (int a = 0; a < 3; ++a) { blo[s][a] = 0.f; bhi[s][a] = 0.f; }
    if (s < nslots) {
      for (int li = 0; li < kWave; ++li) {
        const int g = w + W * (li + kWave * s);
        if (g >= nb) break;
        const float4 p = pts[g * kBP + lane];
        const unsigned key = keys[g * kBP + lane];
        const bool cnd = p.w >= 0.f;
        float l3[3], h3[3];
        l3[0] = wave_allmin_f32(cnd ? p.x : INFINITY); h3[0] = wave_allmax_f32(cnd ? p.x : -INFINITY);
        l3[1] = wave_allmin_f32(cnd ? p.y : INFINITY); h3[1] = wave_allmax_f32(cnd ? p.y : -INFINITY);
        l3[2] = wave_allmin_f32(cnd ? p.z : INFINITY); h3[2] = wave_allmax_f32(cnd ? p.z : -INFINITY);
        const float anyv = wave_allmax_f32(cnd ? p.w : -INFINITY);  // 1e10 if the bucket holds a candidate
        const unsigned kmin = grp_allmin_u32<kWave>(cnd ? key : 0xFFFFFFFFu);
        if (cnd && key == kmin) s_cand[g] = p;
        if (lane == li) {
#pragma unroll
          for (int a = 0; a < 3; ++a) { blo[s][a] = l3[a]; bhi[s][a] = h3[a]; }
          brank[s] = fps_rank_of(anyv); bkey[s] = kmin;
        }
      }
    }
  }

  // ---- rounds ------------------------------------------------------------------------------------------------
  float cx = p0x, cy = p0y, cz = p0z;  // the reference starts from index 0 unconditionally (:89-90)
  if (tid == 0) out[0] = 0;
  unsigned long long acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  bool dirty = true;  // wave-uniform: the wave's best bucket (owner lane best_l, slot best_s) was processed
  int best_l = -1, best_s = -1;
  unsigned wrank = 0u, wkey = 0xFFFFFFFFu;
  float4 cw = make_float4(0.f, 0.f, 0.f, 0.f);  // lane 0: coordinates of this wave's best
  unsigned long long todo[NS];

  // next surviving bucket of this wave, across slots (scalar code)
  auto pick = [&](int& l, int& sl) __attribute__((always_inline)) {  // the caller knows one exists
    if (NS == 1) {
      l = __ffsll((long long)todo[0]) - 1;
      sl = 0;
      todo[0] &= todo[0] - 1ull;
      return;
    }
    bool found = false;
#pragma unroll
    for (int q = 0; q < NS; ++q) {  // no early exit: every todo[] index stays a compile-time constant (registers, not scratch)
      const bool hit = !found && todo[q] != 0ull;
      if (hit) { l = __ffsll((long long)todo[q]) - 1; sl = q; }
      todo[q] = hit ? (todo[q] & (todo[q] - 1ull)) : todo[q];
      found |= hit;
    }
  };
  // NC buckets (owner lane bl[u], slot bs[u]) as NC interleaved chains
  int bl[kKB] = {0, 0, 0, 0}, bs[kKB] = {0, 0, 0, 0};
  auto process = [&](auto nc_tag) __attribute__((always_inline)) {
    constexpr int NC = decltype(nc_tag)::value;
    unsigned g[NC], pos[NC], ky[NC], rk[NC], v[NC];
    float4 p[NC];
#pragma unroll
    for (int u = 0; u < NC; ++u) {
      g[u] = (unsigned)(w + W * (bl[u] + kWave * bs[u]));
      pos[u] = g[u] * kBP + (unsigned)lane;
      p[u] = pts[pos[u]];
      ky[u] = keys[pos[u]];
    }
#pragma unroll
    for (int u = 0; u < NC; ++u) {
      const float d = sqdist3(p[u].x - cx, p[u].y - cy, p[u].z - cz);
      const float t = fminf(d, p[u].w);  // -inf (non-candidate) stays -inf
      pts[pos[u]].w = t;                 // unconditional: a predicated store would split the chains into blocks
      rk[u] = fps_rank_of(t);
      v[u] = rk[u];
    }
    // wave max of the ranks as a scalar: six DPP stages + v_readlane
#pragma unroll
    for (int u = 0; u < NC; ++u) v[u] = wave_max_u32_s(v[u]);
    // an active bucket holds a candidate, so its max rank is >= 1 and padding lanes (rank 0) never match
    unsigned long long tm[NC];
    bool multi = false;
#pragma unroll
    for (int u = 0; u < NC; ++u) {
      tm[u] = __ballot(rk[u] == v[u]);
      multi |= (tm[u] & (tm[u] - 1ull)) != 0ull;
    }
    if (__builtin_expect(multi, 0)) {  // wave-uniform, rare: equal maxima inside a bucket, the smallest tie key wins
#pragma unroll
      for (int u = 0; u < NC; ++u) {
        const bool top = rk[u] == v[u];
        const unsigned gk = wave_min_u32_s(top ? ky[u] : 0xFFFFFFFFu);
        tm[u] = __ballot(top && ky[u] == gk);
      }
    }
#pragma unroll
    for (int u = 0; u < NC; ++u) {
      const int wl = __ffsll((long long)tm[u]) - 1;
      const unsigned qr = readlane_u32(rk[u], wl), qk = readlane_u32(ky[u], wl);
      if (lane == wl) s_cand[g[u]] = p[u];  // (x,y,z) of the bucket's new max; .w is not read
      // ranks only ever decrease, so the wave's arg-max stays valid unless it was THIS bucket
      dirty |= (bl[u] == best_l) & (bs[u] == best_s);
#pragma unroll
      for (int q = 0; q < NS; ++q) {
        if (bs[u] == q) {  // wave-uniform
          brank[q] = writelane_u32(brank[q], qr, bl[u]);
          bkey[q] = writelane_u32(bkey[q], qk, bl[u]);
        }
      }
    }
  };

  for (int j = 1; j < m; ++j) {
    unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
    if (DEBUG) t0 = rows_clock();
    int left = 0;  // surviving buckets of this wave
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      todo[s] = 0;
      if (NS == 1 || s < nslots) {
        // distance of the sample to the bucket box, same arithmetic as a point distance; d >= +0, so
        // d < max  <=>  bits(d) + 1 < rank(max)   (rank 0, no candidate: never)
        const float dx = fmaxf(fmaxf(blo[s][0] - cx, cx - bhi[s][0]), 0.f);
        const float dy = fmaxf(fmaxf(blo[s][1] - cy, cy - bhi[s][1]), 0.f);
        const float dz = fmaxf(fmaxf(blo[s][2] - cz, cz - bhi[s][2]), 0.f);
        todo[s] = __ballot(__float_as_uint(sqdist3(dx, dy, dz)) + 1u < brank[s]);
        left += __popcll(todo[s]);
      }
    }
    if (DEBUG) t1 = rows_clock();
    while (__builtin_expect(left > 0, 1)) {
      const int cnt = left < kKB ? left : kKB;
      left -= cnt;
      pick(bl[0], bs[0]);
      if (__builtin_expect(cnt == 1, 1)) {
        process(std::integral_constant<int, 1>());
      } else {
        pick(bl[1], bs[1]);
        if (cnt == 2) {
          process(std::integral_constant<int, 2>());
        } else {
          pick(bl[2], bs[2]);
          bl[3] = bl[0]; bs[3] = bs[0];  // a missing fourth repeats the first bucket (harmless: min and max are idempotent)
          if (cnt == 4) pick(bl[3], bs[3]);
          process(std::integral_constant<int, kKB>());
        }
      }
      if (DEBUG) { ++acc[5]; acc[6] += cnt; }
    }
    if (DEBUG) t2 = rows_clock();
    // arg-max over this wave's buckets (only if one of them changed), its coordinates from cand[]
    if (__builtin_expect(dirty, 0)) {
      dirty = false;
      unsigned mrank = brank[0], mkey = bkey[0];
      int mslot = 0;
#pragma unroll
      for (int s = 1; s < NS; ++s) {
        if (s < nslots) {
          const bool b = (brank[s] > mrank) | ((brank[s] == mrank) & (bkey[s] < mkey));
          mrank = b ? brank[s] : mrank;
          mkey = b ? bkey[s] : mkey;
          mslot = b ? s : mslot;
        }
      }
      wrank = wave_max_u32_s(mrank);
      const bool top = mrank == wrank;
      unsigned long long tm = __ballot(top);
      if (__builtin_expect((tm & (tm - 1ull)) != 0ull, 0)) {
        const unsigned k2 = wave_min_u32_s(top ? mkey : 0xFFFFFFFFu);
        tm = __ballot(top && mkey == k2);
      }
      const int wl = __ffsll((long long)tm) - 1;
      wkey = readlane_u32(mkey, wl);
      const int wslot = NS > 1 ? (int)readlane_u32((unsigned)mslot, wl) : 0;
      best_l = wl; best_s = wslot;
      if (lane == 0) {
        const int gw = min(w + W * (wl + kWave * wslot), nb - 1);  // wrank == 0: any bucket, the entry is ignored
        cw = s_cand[gw];
      }
    }
    const int par = j & 1;
    if (lane == 0) {
      s_xr[par][w] = wrank;
      s_xc[par][w] = make_float4(cw.x, cw.y, cw.z, __uint_as_float(wkey));
    }
    if (DEBUG) t3 = rows_clock();
    // LDS-only barrier: a full __syncthreads() would also drain vmcnt, i.e. wait for the L2 acknowledgement of this
    // round's running-distance stores, which only this wave reads back.
    // (Tried: the last wave to arrive, found with a returning LDS add, decodes alone and publishes the sample while the
    // others are parked at the barrier: +8 %, the add's round trip and the extra read cost more than the contention.)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (DEBUG) t4 = rows_clock();
    {
      const int sl = lane & (W - 1);  // lanes 0..W-1 hold the W entries (the other rows repeat them)
      const unsigned xr = s_xr[par][sl];
      float4 xc = s_xc[par][sl];
      asm volatile("" : "+v"(xc.x), "+v"(xc.y), "+v"(xc.z), "+v"(xc.w));  // one ds_read_b128 here, not a dependent read later
      const unsigned xk = __float_as_uint(xc.w);
      const unsigned grank = row_allmax_u32_fx(xr);
      const bool top = xr == grank;
      unsigned long long tm = __ballot(top) & ((1ull << W) - 1ull);
      if (__builtin_expect((tm & (tm - 1ull)) != 0ull, 0)) {
        const unsigned k2 = row_allmin_u32_fx(top ? xk : 0xFFFFFFFFu);
        tm = __ballot(top && xk == k2) & ((1ull << W) - 1ull);
      }
      const int ws = __ffsll((long long)tm) - 1;
      const unsigned gkey = readlane_u32(xk, ws);
      const float nx = readlane_f32(xc.x, ws), ny = readlane_f32(xc.y, ws), nz = readlane_f32(xc.z, ws);
      // no candidate at all: the reference's reduction returns besti = 0 (:93-94), whose key is 0
      cx = grank ? nx : p0x; cy = grank ? ny : p0y; cz = grank ? nz : p0z;
      if (tid == 0) out[j] = (int32_t)(grank ? gkey : 0u);  // translated to the point index below
    }
    if (DEBUG) {
      const unsigned long long t5 = rows_clock();
      acc[0] += t1 - t0; acc[1] += t2 - t1; acc[2] += t3 - t2; acc[3] += t4 - t3; acc[4] += t5 - t4;
    }
  }
  // ---- epilogue: tie keys -> point indices (wave 0 wrote the keys; they are read past this CU's L1) ---------------
  __syncthreads();
  for (int k = 1 + tid; k < m; k += T) {
    const unsigned key = (unsigned)__hip_atomic_load(&out[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    out[k] = fps_decode_key(key, rb, S.ref_log2);
  }
  if (DEBUG && lane == 0 && blockIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) g_rows_cyc[w][i] = acc[i];
  }
}

// ---- host --------------------------------------------------------------------------------------------------------

bool fps_rows_plan(int nmax, RowsPlan* plan) {
  const int env_impl = VDETR_AB("VDETR_FPS_IMPL", 0);  // 2: fps.hip's kernel always
  const int env_waves = VDETR_AB("VDETR_FPS_WAVES", 0);
  if (env_impl == 2 || nmax <= 0) return false;
  const long nb = ((long)nmax + kBP - 1) / kBP;
  int waves = (env_waves == 4 || env_waves == 8 || env_waves == 16) ? env_waves : 16;
  while (waves < 16 && nb > (long)kRowsSlots * kWave * waves) waves *= 2;
  if (nb > (long)kRowsSlots * kWave * waves) return false;
  plan->waves = waves;
  plan->bucket_pts = kBP;
  return true;
}

template <int W, int NS>
static int launch_rows(RowsParams& P, int b, size_t lds, bool debug, hipStream_t stream) {
  if (debug) {
    int rc = set_lds(fps_rows_kernel<W, NS, true>, lds, "furthest_point_sampling");
    if (rc != VDETR_OK) return rc;
    hipLaunchKernelGGL((fps_rows_kernel<W, NS, true>), dim3(b), dim3(W * kWave), lds, stream, P);
    (void)hipDeviceSynchronize();
    unsigned long long z[16][8];
    (void)hipMemcpyFromSymbol(z, HIP_SYMBOL(g_rows_cyc), sizeof(z));
    const unsigned long long r = P.m > 1 ? P.m - 1 : 1;
    for (int i = 0; i < W; ++i)
      fprintf(stderr, "[fps rows debug] W=%d NS=%d wave %2d cycles/round: test %llu batches %llu reduce %llu barrier %llu decode %llu | batches/round %.2f buckets/round %.2f\n",
              W, NS, i, z[i][0] / r, z[i][1] / r, z[i][2] / r, z[i][3] / r, z[i][4] / r, (double)z[i][5] / (double)r, (double)z[i][6] / (double)r);
    return check_launch("furthest_point_sampling");
  }
  int rc = set_lds(fps_rows_kernel<W, NS, false>, lds, "furthest_point_sampling");
  if (rc != VDETR_OK) return rc;
  hipLaunchKernelGGL((fps_rows_kernel<W, NS, false>), dim3(b), dim3(W * kWave), lds, stream, P);
  return check_launch("furthest_point_sampling");
}

int fps_rows_launch(RowsParams& P, int b, const RowsPlan& pl, hipStream_t stream) {
  const bool debug = VDETR_AB("VDETR_FPS_DEBUG", 0) != 0;
  // VDETR_FPS_TREE: 0 = runs of 64 sorted points always; 1 (default) = tree leaves in the slots the runs need (a
  // second slot per owner lane costs more in the box test than the tighter boxes win: 4.92 vs 4.28 ms at 40k points);
  // 2 = room for 2 n / 64 + 64 leaves
  const int env_tree = VDETR_AB("VDETR_FPS_TREE", 1);
  long capmax = 0, runmax = 0;
  for (int i = 0; i < b; ++i) {
    const long runs = ((long)P.scenes[i].n + kBP - 1) / kBP;
    runmax = runmax > runs ? runmax : runs;
  }
  const long per_slot = (long)pl.waves * kWave;
  const long run_slots = (runmax + per_slot - 1) / per_slot * per_slot;
  for (int i = 0; i < b; ++i) {
    const long runs = ((long)P.scenes[i].n + kBP - 1) / kBP;
    long cap = fps_rows_cap(P.scenes[i].n, pl.waves);  // what the workspace holds
    if (env_tree == 0) cap = 0;
    else if (env_tree == 1) cap = cap < run_slots ? cap : run_slots;
    P.scenes[i].cap_buckets = (int)cap;
    const long most = cap > runs ? cap : runs;
    capmax = capmax > most ? capmax : most;
  }
  size_t lds = (size_t)(kRowsHistWords + kRowsCells) * sizeof(int);
  if ((size_t)capmax * sizeof(float4) > lds) lds = (size_t)capmax * sizeof(float4);
  const int ns = (int)((capmax + per_slot - 1) / per_slot);
  switch (pl.waves) {
    case 16:
      if (ns <= 1) return launch_rows<16, 1>(P, b, lds, debug, stream);
      if (ns <= 2) return launch_rows<16, 2>(P, b, lds, debug, stream);
      return launch_rows<16, 4>(P, b, lds, debug, stream);
    case 8:
      if (ns <= 2) return launch_rows<8, 2>(P, b, lds, debug, stream);
      return launch_rows<8, 4>(P, b, lds, debug, stream);
    case 4:
      if (ns <= 2) return launch_rows<4, 2>(P, b, lds, debug, stream);
      return launch_rows<4, 4>(P, b, lds, debug, stream);
  }
  set_error("furthest_point_sampling: no kernel for %d waves", pl.waves);
  return VDETR_ERR_ARG;
}

}  // namespace vdetr
