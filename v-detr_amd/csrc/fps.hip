// fps.hip — furthest point sampling for gfx950, bit-exact with the reference kernel's result
// (third_party/pointnet2/_ext_src/src/sampling_gpu.cu:73-176) but a different algorithm.
//
// What the reference does: ONE 512-thread block per batch element; each of the m-1 rounds re-reads all
// n points + n running distances from global memory (20 B/point/round = 3.3 GB for n=40k, m=4096) and
// does a 9-level __syncthreads tree reduction.  At batch 1 that is one SM/CU and it is bound by the
// L2->CU stream, not by arithmetic.
//
// What this kernel does (one 1024-thread workgroup = 16 waves per batch element, one launch):
//   prologue  the cloud is counting-sorted into Morton order of a 16^3 grid (LDS histogram + scan),
//             written as float4 (x,y,z,t) to the caller's workspace and cut into 64-point BUCKETS;
//             each bucket's bounding box lives in the registers of one owner lane.
//   round j   every lane tests its buckets' boxes against the newly sampled point p: the running
//             distance t_k = min(t_k, |p_k - p|²) can only change inside buckets whose box is closer
//             to p than the bucket's current max t.  The skip test is EXACT, not approximate:
//             fl(a-b), fl(x*x) and fma are monotone, so the distance computed for any point of a box
//             is >= the distance computed (same formula) for the box's nearest corner.
//             Only the surviving buckets (a handful once sampling has spread) are touched: one wave
//             per bucket, one 16-B load per lane, a DPP/permlane arg-max, then a 16-slot LDS exchange
//             and ONE barrier per round.
//   result    the arg-max with the reference's tie order.  The reference's strided scan + tree
//             reduction picks, among equal maxima, the point whose scanning thread (k mod bs) has the
//             smallest BIT-REVERSED id, then the smallest k (strict '>' keeps the lower slot at every
//             tree level, and the last level compares bit 0).  That order is encoded in a 32-bit key so
//             any reduction shape reproduces it.
// Work drops from n*(m-1) distance evaluations to roughly 4 n ln m; the per-round cost is a few L2
// round trips instead of a 640 KB sweep.
#include "common.h"
#include "wave.h"

namespace vdetr {

constexpr int kFpsThreads = 1024;
constexpr int kFpsWaves = kFpsThreads / kWave;  // 16
constexpr int kFpsSlots = 4;                    // buckets per owner lane
constexpr int kFpsMaxBuckets = kFpsWaves * kWave * kFpsSlots;  // 4096
constexpr int kGridBits = 4;
constexpr int kCells = 1 << (3 * kGridBits);  // 4096

struct FpsParams {
  const float* xyz;  // (b,n,3)
  int32_t* idx;      // (b,m)
  float4* pts;       // workspace: (b, npad) sorted (x,y,z,t)
  uint32_t* keys;    // workspace: (b, npad) tie-order key of each sorted point
  int n, m, npad;
  int bucket_pts;    // 64 * R
  int nbuckets;
  int ref_block;     // opt_n_threads(n) of the reference (cuda_utils.h:17-21)
  int ref_log2;
};

__device__ __forceinline__ unsigned bitrev(unsigned v, int bits) { return bits ? (__brev(v) >> (32 - bits)) : 0u; }
__device__ __forceinline__ unsigned spread3(unsigned v) {  // 4 bits -> every third bit
  return (v & 1u) | ((v & 2u) << 2) | ((v & 4u) << 4) | ((v & 8u) << 6);
}
// (t, key) -> one 64-bit word whose unsigned max is "largest t, then smallest key".  t is >= 0 for every
// candidate point; non-candidates (origin-skip, padding) carry t = -inf and map to 0.
__device__ __forceinline__ unsigned long long pack_cand(float t, unsigned key) {
  return t >= 0.f ? (((unsigned long long)(__float_as_uint(t) + 1u) << 32) | (unsigned)(~key)) : 0ull;
}

__global__ __launch_bounds__(kFpsThreads) void fps_kernel(FpsParams P) {
  __shared__ int s_hist[kCells];
  __shared__ int s_wsum[kFpsWaves];
  __shared__ float s_red[kFpsWaves][6];
  __shared__ unsigned long long s_best[2][kFpsWaves];
  __shared__ float s_bxyz[2][kFpsWaves][3];

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int bi = blockIdx.x;
  const float* __restrict__ xyz = P.xyz + (size_t)bi * P.n * 3;
  int32_t* __restrict__ out = P.idx + (size_t)bi * P.m;
  float4* __restrict__ pts = P.pts + (size_t)bi * P.npad;
  uint32_t* __restrict__ keys = P.keys + (size_t)bi * P.npad;
  const int n = P.n;

  // ---- prologue 1: bounding box of the cloud ------------------------------------------------------
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int k = tid; k < n; k += kFpsThreads) {
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
  for (int c = tid; c < kCells; c += kFpsThreads) s_hist[c] = 0;
  __syncthreads();
  float scale[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float l = s_red[0][a], h = s_red[0][3 + a];
    for (int ww = 1; ww < kFpsWaves; ++ww) { l = fminf(l, s_red[ww][a]); h = fmaxf(h, s_red[ww][3 + a]); }
    lo[a] = l;
    const float ext = h - l;
    scale[a] = (ext > 0.f && ext < INFINITY) ? (float)(1 << kGridBits) / ext : 0.f;
  }
  auto cell_of = [&](float x, float y, float z) -> unsigned {
    const int gmax = (1 << kGridBits) - 1;
    // NaN / inf coordinates fall into cell 0 (the cast of NaN is made harmless by the clamp on an int)
    int cx = (int)fminf(fmaxf((x - lo[0]) * scale[0], 0.f), (float)gmax);
    int cy = (int)fminf(fmaxf((y - lo[1]) * scale[1], 0.f), (float)gmax);
    int cz = (int)fminf(fmaxf((z - lo[2]) * scale[2], 0.f), (float)gmax);
    cx = min(max(cx, 0), gmax); cy = min(max(cy, 0), gmax); cz = min(max(cz, 0), gmax);
    return spread3((unsigned)cx) | (spread3((unsigned)cy) << 1) | (spread3((unsigned)cz) << 2);
  };

  // ---- prologue 2: histogram, exclusive scan, scatter ---------------------------------------------
  for (int k = tid; k < n; k += kFpsThreads)
    atomicAdd(&s_hist[cell_of(xyz[k * 3], xyz[k * 3 + 1], xyz[k * 3 + 2])], 1);
  __syncthreads();
  {
    constexpr int kPer = kCells / kFpsThreads;  // 4 consecutive cells per thread
    int v[kPer], sum = 0;
#pragma unroll
    for (int i = 0; i < kPer; ++i) { v[i] = s_hist[tid * kPer + i]; sum += v[i]; }
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
#pragma unroll
    for (int i = 0; i < kPer; ++i) { s_hist[tid * kPer + i] = run; run += v[i]; }
  }
  __syncthreads();
  const unsigned rb = (unsigned)P.ref_block;
  for (int k = tid; k < n; k += kFpsThreads) {
    const float x = xyz[k * 3], y = xyz[k * 3 + 1], z = xyz[k * 3 + 2];
    const int pos = atomicAdd(&s_hist[cell_of(x, y, z)], 1);
    // origin-skip rule: `if (mag <= 1e-3) continue;` compares the float mag against a DOUBLE literal
    // (sampling_gpu.cu:103-104); mag in the same contraction order as the distance.
    const float mag = __fmaf_rn(z, z, __fmaf_rn(x, x, __fmul_rn(y, y)));
    const bool skip = (double)mag <= 1e-3;
    pts[pos] = make_float4(x, y, z, skip ? -INFINITY : 1e10f);
    keys[pos] = (bitrev((unsigned)k % rb, P.ref_log2) << 22) | ((unsigned)k / rb);
  }
  const float p0x = xyz[0], p0y = xyz[1], p0z = xyz[2];
  for (int k = n + tid; k < P.npad; k += kFpsThreads) {
    pts[k] = make_float4(p0x, p0y, p0z, -INFINITY);
    keys[k] = 0xFFFFFFFFu;
  }
  __syncthreads();  // workgroup-scope release/acquire: the sorted cloud is visible to every wave

  // ---- prologue 3: bucket boxes into owner-lane registers ------------------------------------------
  // bucket g is owned by wave g%16, lane (g/16)%64, slot g/1024
  const int R = P.bucket_pts / kWave;
  float blo[kFpsSlots][3], bhi[kFpsSlots][3], bmax[kFpsSlots], bpx[kFpsSlots], bpy[kFpsSlots], bpz[kFpsSlots];
  unsigned bkey[kFpsSlots];
#pragma unroll
  for (int s = 0; s < kFpsSlots; ++s) {
    bmax[s] = -INFINITY; bkey[s] = 0xFFFFFFFFu; bpx[s] = bpy[s] = bpz[s] = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) { blo[s][a] = 0.f; bhi[s][a] = 0.f; }
    for (int li = 0; li < kWave; ++li) {
      const int g = w + kFpsWaves * (li + kWave * s);
      if (g >= P.nbuckets) break;
      float l3[3] = {INFINITY, INFINITY, INFINITY}, h3[3] = {-INFINITY, -INFINITY, -INFINITY};
      float anyv = -INFINITY;
      for (int r = 0; r < R; ++r) {
        const float4 p = pts[(size_t)g * P.bucket_pts + r * kWave + lane];
        if (p.w >= 0.f) {
          l3[0] = fminf(l3[0], p.x); h3[0] = fmaxf(h3[0], p.x);
          l3[1] = fminf(l3[1], p.y); h3[1] = fmaxf(h3[1], p.y);
          l3[2] = fminf(l3[2], p.z); h3[2] = fmaxf(h3[2], p.z);
          anyv = p.w;
        }
      }
#pragma unroll
      for (int a = 0; a < 3; ++a) { l3[a] = wave_allmin_f32(l3[a]); h3[a] = wave_allmax_f32(h3[a]); }
      anyv = wave_allmax_f32(anyv);
      if (lane == li) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { blo[s][a] = l3[a]; bhi[s][a] = h3[a]; }
        bmax[s] = anyv;  // 1e10 if the bucket holds a candidate, -inf otherwise
      }
    }
  }

  // ---- rounds ---------------------------------------------------------------------------------------
  float cx = p0x, cy = p0y, cz = p0z;  // the reference starts from index 0 unconditionally (:89-90)
  if (tid == 0) out[0] = 0;
  for (int j = 1; j < P.m; ++j) {
#pragma unroll
    for (int s = 0; s < kFpsSlots; ++s) {
      // distance of p to the bucket box, same arithmetic as a point distance
      const float dx = fmaxf(fmaxf(blo[s][0] - cx, cx - bhi[s][0]), 0.f);
      const float dy = fmaxf(fmaxf(blo[s][1] - cy, cy - bhi[s][1]), 0.f);
      const float dz = fmaxf(fmaxf(blo[s][2] - cz, cz - bhi[s][2]), 0.f);
      const bool active = sqdist3(dx, dy, dz) < bmax[s];
      unsigned long long todo = __ballot(active);
      while (todo) {
        const int li = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const int g = w + kFpsWaves * (li + kWave * s);
        float4* bp = pts + (size_t)g * P.bucket_pts;
        const uint32_t* bk = keys + (size_t)g * P.bucket_pts;
        unsigned long long cand = 0ull;
        float qx = 0.f, qy = 0.f, qz = 0.f;
        for (int r = 0; r < R; ++r) {
          const float4 p = bp[r * kWave + lane];
          const unsigned key = bk[r * kWave + lane];
          const float d = sqdist3(p.x - cx, p.y - cy, p.z - cz);
          const float t = fminf(d, p.w);  // p.w = -inf for non-candidates: stays -inf
          if (t < p.w) bp[r * kWave + lane].w = t;
          const unsigned long long c = pack_cand(t, key);
          if (c > cand) { cand = c; qx = p.x; qy = p.y; qz = p.z; }
        }
        const unsigned long long best = wave_allmax_u64(cand);
        const int src = __ffsll((long long)__ballot(cand == best)) - 1;
        const float wx = readlane_f32(qx, src), wy = readlane_f32(qy, src), wz = readlane_f32(qz, src);
        if (lane == li) {
          bmax[s] = best ? __uint_as_float((unsigned)(best >> 32) - 1u) : -INFINITY;
          bkey[s] = ~(unsigned)best;
          bpx[s] = wx; bpy[s] = wy; bpz[s] = wz;
        }
      }
    }
    // arg-max over this wave's buckets, then over the 16 waves through LDS (double-buffered: 1 barrier)
    unsigned long long mine = 0ull;
    float mx = 0.f, my = 0.f, mz = 0.f;
#pragma unroll
    for (int s = 0; s < kFpsSlots; ++s) {
      const unsigned long long c = pack_cand(bmax[s], bkey[s]);
      if (c > mine) { mine = c; mx = bpx[s]; my = bpy[s]; mz = bpz[s]; }
    }
    const unsigned long long wbest = wave_allmax_u64(mine);
    const int src = __ffsll((long long)__ballot(mine == wbest)) - 1;
    const float wx = readlane_f32(mx, src), wy = readlane_f32(my, src), wz = readlane_f32(mz, src);
    const int par = j & 1;
    if (lane == 0) {
      s_best[par][w] = wbest;
      s_bxyz[par][w][0] = wx; s_bxyz[par][w][1] = wy; s_bxyz[par][w][2] = wz;
    }
    __syncthreads();
    const unsigned long long slot = s_best[par][lane & (kFpsWaves - 1)];
    const unsigned long long gbest = row_allmax_u64(slot);  // every 16-lane row holds all 16 slots
    const int ws = (__ffsll((long long)__ballot(slot == gbest)) - 1) & (kFpsWaves - 1);
    int winner = 0;
    if (gbest) {
      const unsigned key = ~(unsigned)gbest;
      winner = (int)((key & 0x3FFFFFu) * rb + bitrev(key >> 22, P.ref_log2));
      cx = s_bxyz[par][ws][0]; cy = s_bxyz[par][ws][1]; cz = s_bxyz[par][ws][2];
    } else {  // no candidate at all: the reference's reduction returns besti = 0 (:93-94)
      cx = p0x; cy = p0y; cz = p0z;
    }
    if (tid == 0) out[j] = winner;
  }
}

}  // namespace vdetr

using namespace vdetr;

static int fps_geometry(int n, int* npad, int* bucket_pts, int* nbuckets) {
  if (n <= 0) { *npad = 0; *bucket_pts = kWave; *nbuckets = 0; return 0; }
  long r = ((long)n + (long)kFpsMaxBuckets * kWave - 1) / ((long)kFpsMaxBuckets * kWave);
  *bucket_pts = (int)(r * kWave);
  *nbuckets = (int)(((long)n + *bucket_pts - 1) / *bucket_pts);
  *npad = *nbuckets * *bucket_pts;
  return 0;
}

extern "C" size_t vdetr_fps_workspace_bytes(int b, int n) {
  int npad, bp, nb;
  fps_geometry(n, &npad, &bp, &nb);
  if (b <= 0) return 0;
  return (size_t)b * (size_t)npad * (sizeof(float4) + sizeof(uint32_t)) + 256;
}

extern "C" int vdetr_furthest_point_sampling_f32(const float* xyz, int b, int n, int m, int32_t* idx,
                                                 void* workspace, size_t workspace_bytes,
                                                 vdetr_stream_t stream) {
  VDETR_REQUIRE(b >= 0 && n >= 0, "furthest_point_sampling: negative dimension");
  if (b == 0 || m <= 0) return VDETR_OK;  // `if (m <= 0) return;` sampling_gpu.cu:77
  VDETR_REQUIRE(n > 0, "furthest_point_sampling: empty cloud with nsamples=%d", m);
  VDETR_REQUIRE(xyz && idx, "furthest_point_sampling: null pointer");
  VDETR_REQUIRE((long)n < (1L << 30), "furthest_point_sampling: n=%d too large", n);
  FpsParams P;
  fps_geometry(n, &P.npad, &P.bucket_pts, &P.nbuckets);
  const size_t need = vdetr_fps_workspace_bytes(b, n);
  if (!workspace || workspace_bytes < need) {
    set_error("furthest_point_sampling: workspace %zu B < required %zu B", workspace_bytes, need);
    return VDETR_ERR_WORKSPACE;
  }
  uintptr_t base = ((uintptr_t)workspace + 255) & ~(uintptr_t)255;
  P.pts = (float4*)base;
  P.keys = (uint32_t*)(base + (size_t)b * P.npad * sizeof(float4));
  P.xyz = xyz; P.idx = idx; P.n = n; P.m = m;
  // opt_n_threads(n): 2^floor(log2 n) clamped to [1,512] (cuda_utils.h:17-21)
  int lg = 0;
  while ((2L << lg) <= (long)n) ++lg;
  if (lg > 9) lg = 9;
  P.ref_log2 = lg; P.ref_block = 1 << lg;
  hipLaunchKernelGGL(fps_kernel, dim3(b), dim3(kFpsThreads), 0, (hipStream_t)stream, P);
  return check_launch("furthest_point_sampling");
}
