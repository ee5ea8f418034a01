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
#include "fps.h"

#include <stdlib.h>

namespace vdetr {

constexpr int kFpsThreads = 1024;
constexpr int kFpsWaves = kFpsThreads / kWave;  // 16
constexpr int kFpsSlots = 4;                    // buckets per owner lane
constexpr int kFpsMaxBuckets = kFpsWaves * kWave * kFpsSlots;  // 4096
constexpr int kGridBits = 4;
constexpr int kCells = 1 << (3 * kGridBits);  // 4096

struct FpsScene {    // one scene of a variable-length batch: its own cloud, size and sorting geometry
  const float* xyz;  // (n,3)
  long ws_off;       // first workspace element of this scene
  int n, npad, bucket_pts, nbuckets, ref_block, ref_log2;
};

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
  int nscenes;       // > 0: variable-length batch, workgroup bi takes its geometry from scenes[bi]
  FpsScene scenes[kFpsMaxScenes];
};

__device__ __forceinline__ unsigned spread3(unsigned v) {  // 4 bits -> every third bit
  return (v & 1u) | ((v & 2u) << 2) | ((v & 4u) << 4) | ((v & 8u) << 6);
}
__device__ __forceinline__ unsigned rank_of(float t) { return fps_rank_of(t); }
__device__ __forceinline__ float t_of_rank(unsigned r) { return fps_t_of_rank(r); }

// wave arg-max of (rank desc, key asc): two 32-bit all-reduces; returns the winning lane
struct WaveBest {
  unsigned rank, key;
  int lane;
};
__device__ __forceinline__ WaveBest wave_argbest(unsigned rank, unsigned key) {
  WaveBest r;
  r.rank = wave_allmax_u32(rank);
  r.key = wave_allmin_u32(rank == r.rank ? key : 0xFFFFFFFFu);
  r.lane = __ffsll((long long)__ballot(rank == r.rank && key == r.key)) - 1;
  return r;
}

__device__ unsigned long long g_fps_cyc[8];
__device__ __forceinline__ unsigned long long fps_clock() {
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  const unsigned long long t = clock64();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}

template <bool DEBUG>
__global__ __launch_bounds__(kFpsThreads) void fps_kernel(FpsParams Pin) {
  FpsParams P = Pin;
  __shared__ int s_hist[kCells];
  __shared__ int s_wsum[kFpsWaves];
  __shared__ float s_red[kFpsWaves][6];
  __shared__ unsigned s_rank[2][kFpsWaves], s_key[2][kFpsWaves];
  __shared__ float s_bxyz[2][kFpsWaves][3];

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int bi = blockIdx.x;
  size_t ws_off = (size_t)bi * P.npad;
  const float* xyz0 = P.xyz + (size_t)bi * P.n * 3;
  if (P.nscenes > 0) {  // every scene of the batch has its own size: one workgroup each, all in this launch
    const FpsScene S = Pin.scenes[bi];
    P.n = S.n, P.npad = S.npad, P.bucket_pts = S.bucket_pts, P.nbuckets = S.nbuckets;
    P.ref_block = S.ref_block, P.ref_log2 = S.ref_log2;
    xyz0 = S.xyz, ws_off = (size_t)S.ws_off;
  }
  const float* __restrict__ xyz = xyz0;
  int32_t* __restrict__ out = P.idx + (size_t)bi * P.m;
  float4* __restrict__ pts = P.pts + ws_off;
  uint32_t* __restrict__ keys = P.keys + ws_off;
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
    const float mag = sqdist3(x, y, z);
    const bool skip = (double)mag <= 1e-3;
    pts[pos] = make_float4(x, y, z, skip ? -INFINITY : 1e10f);
    keys[pos] = fps_tie_key((unsigned)k, rb, P.ref_log2);
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
  constexpr int kBatch = 4;  // buckets in flight per wave: their loads are issued together, their reductions interleave
  float cx = p0x, cy = p0y, cz = p0z;  // the reference starts from index 0 unconditionally (:89-90)
  if (tid == 0) out[0] = 0;
  for (int j = 1; j < P.m; ++j) {
    unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0;
    if (DEBUG) t0 = fps_clock();
#pragma unroll
    for (int s = 0; s < kFpsSlots; ++s) {
      // distance of p to the bucket box, same arithmetic as a point distance
      const float dx = fmaxf(fmaxf(blo[s][0] - cx, cx - bhi[s][0]), 0.f);
      const float dy = fmaxf(fmaxf(blo[s][1] - cy, cy - bhi[s][1]), 0.f);
      const float dz = fmaxf(fmaxf(blo[s][2] - cz, cz - bhi[s][2]), 0.f);
      const bool active = sqdist3(dx, dy, dz) < bmax[s];
      unsigned long long todo = __ballot(active);
      while (todo) {
        int li[kBatch];
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
          li[u] = todo ? __ffsll((long long)todo) - 1 : -1;
          todo &= todo - 1;  // 0 stays 0
        }
        unsigned rk[kBatch], ky[kBatch];
        float qx[kBatch], qy[kBatch], qz[kBatch];
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
          rk[u] = 0u; ky[u] = 0xFFFFFFFFu; qx[u] = qy[u] = qz[u] = 0.f;
        }
        if (R == 1) {
          float4 pt[kBatch];
          unsigned kk[kBatch];
#pragma unroll
          for (int u = 0; u < kBatch; ++u)
            if (li[u] >= 0) {  // wave-uniform
              const size_t base = (size_t)(w + kFpsWaves * (li[u] + kWave * s)) * kWave + lane;
              pt[u] = pts[base];
              kk[u] = keys[base];
            }
#pragma unroll
          for (int u = 0; u < kBatch; ++u)
            if (li[u] >= 0) {
              const size_t base = (size_t)(w + kFpsWaves * (li[u] + kWave * s)) * kWave + lane;
              const float d = sqdist3(pt[u].x - cx, pt[u].y - cy, pt[u].z - cz);
              const float t = fminf(d, pt[u].w);  // -inf (non-candidate) stays -inf
              if (t < pt[u].w) pts[base].w = t;
              rk[u] = rank_of(t); ky[u] = kk[u]; qx[u] = pt[u].x; qy[u] = pt[u].y; qz[u] = pt[u].z;
            }
        } else {
#pragma unroll
          for (int u = 0; u < kBatch; ++u)
            if (li[u] >= 0) {
              const size_t base = (size_t)(w + kFpsWaves * (li[u] + kWave * s)) * P.bucket_pts + lane;
              for (int r = 0; r < R; ++r) {
                const float4 p = pts[base + r * kWave];
                const unsigned key = keys[base + r * kWave];
                const float d = sqdist3(p.x - cx, p.y - cy, p.z - cz);
                const float t = fminf(d, p.w);
                if (t < p.w) pts[base + r * kWave].w = t;
                const unsigned rr = rank_of(t);
                if (rr > rk[u] || (rr == rk[u] && key < ky[u])) { rk[u] = rr; ky[u] = key; qx[u] = p.x; qy[u] = p.y; qz[u] = p.z; }
              }
            }
        }
#pragma unroll
        for (int u = 0; u < kBatch; ++u)
          if (li[u] >= 0) {
            const WaveBest wb = wave_argbest(rk[u], ky[u]);
            const float wx = readlane_f32(qx[u], wb.lane), wy = readlane_f32(qy[u], wb.lane), wz = readlane_f32(qz[u], wb.lane);
            if (lane == li[u]) {
              bmax[s] = t_of_rank(wb.rank);
              bkey[s] = wb.key;
              bpx[s] = wx; bpy[s] = wy; bpz[s] = wz;
            }
          }
      }
    }
    if (DEBUG) t1 = fps_clock();
    // arg-max over this wave's buckets, then over the 16 waves through LDS (double-buffered: 1 barrier)
    unsigned mrank = 0u, mkey = 0xFFFFFFFFu;
    float mx = 0.f, my = 0.f, mz = 0.f;
#pragma unroll
    for (int s = 0; s < kFpsSlots; ++s) {
      const unsigned rr = rank_of(bmax[s]);
      if (rr > mrank || (rr == mrank && bkey[s] < mkey)) { mrank = rr; mkey = bkey[s]; mx = bpx[s]; my = bpy[s]; mz = bpz[s]; }
    }
    const WaveBest wb = wave_argbest(mrank, mkey);
    const float wx = readlane_f32(mx, wb.lane), wy = readlane_f32(my, wb.lane), wz = readlane_f32(mz, wb.lane);
    const int par = j & 1;
    if (lane == 0) {
      s_rank[par][w] = wb.rank; s_key[par][w] = wb.key;
      s_bxyz[par][w][0] = wx; s_bxyz[par][w][1] = wy; s_bxyz[par][w][2] = wz;
    }
    // LDS-only barrier.  __syncthreads() would also drain vmcnt, i.e. make every wave wait for the L2
    // acknowledgement of the running-distance stores it issued this round — stores that only the SAME lane
    // reads back in a later round, so nobody needs them visible here.  That wait was most of the round time.
    if (DEBUG) t2 = fps_clock();
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (DEBUG) t3 = fps_clock();
    const int sl = lane & (kFpsWaves - 1);  // every 16-lane row reads all 16 slots
    const unsigned srank = s_rank[par][sl], skey = s_key[par][sl];
    const unsigned grank = row_allmax_u32(srank);
    const unsigned gkey = row_allmin_u32(srank == grank ? skey : 0xFFFFFFFFu);
    const int ws = (__ffsll((long long)__ballot(srank == grank && skey == gkey)) - 1) & (kFpsWaves - 1);
    int winner = 0;
    if (grank) {
      winner = fps_decode_key(gkey, rb, P.ref_log2);
      cx = s_bxyz[par][ws][0]; cy = s_bxyz[par][ws][1]; cz = s_bxyz[par][ws][2];
    } else {  // no candidate at all: the reference's reduction returns besti = 0 (:93-94)
      cx = p0x; cy = p0y; cz = p0z;
    }
    if (tid == 0) out[j] = winner;
    if (DEBUG && lane == 0) {
      const unsigned long long t4 = fps_clock();
      if (w == 0 || w == 7) {
        unsigned long long* c = g_fps_cyc + (w ? 4 : 0);
        c[0] += t1 - t0; c[1] += t2 - t1; c[2] += t3 - t2; c[3] += t4 - t3;
      }
    }
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

static int ref_log2_of(int n) { return fps_ref_log2_of(n); }

// points of workspace one scene of n points needs: whichever kernel takes it (their paddings differ)
static size_t ws_points(int n) {
  int npad, bp, nb;
  fps_geometry(n, &npad, &bp, &nb);
  const size_t rows = n > 0 ? (size_t)fps_rows_cap(n) * 64 : 0;  // fps_rows.hip: 64-slot segments of its tree-leaf buckets
  return rows > (size_t)npad ? rows : (size_t)npad;
}

extern "C" size_t vdetr_fps_workspace_bytes(int b, int n) {
  if (b <= 0) return 0;
  return (size_t)b * ws_points(n) * (sizeof(float4) + sizeof(uint32_t)) + 256;
}

extern "C" int vdetr_furthest_point_sampling_f32(const float* xyz, int b, int n, int m, int32_t* idx,
                                                 void* workspace, size_t workspace_bytes,
                                                 vdetr_stream_t stream) {
  VDETR_REQUIRE(b >= 0 && n >= 0, "furthest_point_sampling: negative dimension");
  if (b == 0 || m <= 0) return VDETR_OK;  // `if (m <= 0) return;` sampling_gpu.cu:77
  VDETR_REQUIRE(n > 0, "furthest_point_sampling: empty cloud with nsamples=%d", m);
  VDETR_REQUIRE(xyz && idx, "furthest_point_sampling: null pointer");
  VDETR_REQUIRE((long)n < (1L << 30), "furthest_point_sampling: n=%d too large", n);
  const size_t need = vdetr_fps_workspace_bytes(b, n);
  if (!workspace || workspace_bytes < need) {
    set_error("furthest_point_sampling: workspace %zu B < required %zu B", workspace_bytes, need);
    return VDETR_ERR_WORKSPACE;
  }
  uintptr_t base = ((uintptr_t)workspace + 255) & ~(uintptr_t)255;
  RowsPlan plan;
  if (b <= kFpsMaxScenes && fps_rows_plan(n, &plan)) {  // the row-per-bucket kernel (fps_rows.hip)
    RowsParams R{};
    const long npad = fps_rows_npad(n, plan);
    R.pts = (float4*)base;
    R.keys = (uint32_t*)(base + (size_t)b * npad * sizeof(float4));
    R.m = m;
    for (int i = 0; i < b; ++i) {
      RowsScene& S = R.scenes[i];
      S.xyz = xyz + (size_t)i * n * 3;
      S.idx = idx + (size_t)i * m;
      S.ws_off = (long)i * npad;
      S.n = n;
      S.ref_log2 = ref_log2_of(n);
      S.ref_block = 1 << S.ref_log2;
    }
    return fps_rows_launch(R, b, plan, (hipStream_t)stream);
  }
  FpsParams P;
  fps_geometry(n, &P.npad, &P.bucket_pts, &P.nbuckets);
  P.pts = (float4*)base;
  P.keys = (uint32_t*)(base + (size_t)b * P.npad * sizeof(float4));
  P.xyz = xyz; P.idx = idx; P.n = n; P.m = m;
  P.nscenes = 0;
  P.ref_log2 = ref_log2_of(n); P.ref_block = 1 << P.ref_log2;
  const bool debug = VDETR_AB("VDETR_FPS_DEBUG", 0) != 0;
  if (debug) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_fps_cyc), z, sizeof(z));
    hipLaunchKernelGGL(fps_kernel<true>, dim3(b), dim3(kFpsThreads), 0, (hipStream_t)stream, P);
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(z, HIP_SYMBOL(g_fps_cyc), sizeof(z));
    for (int i = 0; i < 2; ++i)
      fprintf(stderr, "[fps debug] wave %d cycles/round: buckets %llu, wave-argmax+lds %llu, barrier wait %llu, decode %llu\n",
              i ? 7 : 0, z[i * 4] / (m - 1), z[i * 4 + 1] / (m - 1), z[i * 4 + 2] / (m - 1), z[i * 4 + 3] / (m - 1));
    return check_launch("furthest_point_sampling");
  }
  hipLaunchKernelGGL(fps_kernel<false>, dim3(b), dim3(kFpsThreads), 0, (hipStream_t)stream, P);
  return check_launch("furthest_point_sampling");
}

extern "C" size_t vdetr_fps_varlen_workspace_bytes(const int32_t* counts, int b) {
  size_t total = 0;
  for (int i = 0; i < b; ++i) {
    total += ws_points(counts[i]);
  }
  return b > 0 ? total * (sizeof(float4) + sizeof(uint32_t)) + 256 : 0;
}

extern "C" int vdetr_furthest_point_sampling_varlen_f32(const float* const* xyz, const int32_t* counts, int b, int m,
                                                        int32_t* idx, void* workspace, size_t workspace_bytes,
                                                        vdetr_stream_t stream) {
  if (b == 0 || m <= 0) return VDETR_OK;
  VDETR_REQUIRE(b > 0 && b <= kFpsMaxScenes, "furthest_point_sampling_varlen: 1..%d scenes per launch, got %d", kFpsMaxScenes, b);
  VDETR_REQUIRE(xyz && counts && idx, "furthest_point_sampling_varlen: null pointer");
  const size_t need = vdetr_fps_varlen_workspace_bytes(counts, b);
  if (!workspace || workspace_bytes < need) {
    set_error("furthest_point_sampling_varlen: workspace %zu B < required %zu B", workspace_bytes, need);
    return VDETR_ERR_WORKSPACE;
  }
  int nmax = 0;
  for (int i = 0; i < b; ++i) {
    VDETR_REQUIRE(counts[i] > 0 && (long)counts[i] < (1L << 30), "furthest_point_sampling_varlen: scene %d has %d points", i, counts[i]);
    VDETR_REQUIRE(xyz[i] != nullptr, "furthest_point_sampling_varlen: scene %d: null pointer", i);
    nmax = counts[i] > nmax ? counts[i] : nmax;
  }
  RowsPlan plan;
  if (fps_rows_plan(nmax, &plan)) {  // one bucket size for the launch, chosen for the largest scene
    RowsParams R{};
    long total = 0;
    for (int i = 0; i < b; ++i) {
      RowsScene& S = R.scenes[i];
      S.xyz = xyz[i];
      S.idx = idx + (size_t)i * m;
      S.ws_off = total;
      S.n = counts[i];
      S.ref_log2 = ref_log2_of(S.n);
      S.ref_block = 1 << S.ref_log2;
      total += fps_rows_npad(S.n, plan);
    }
    uintptr_t rbase = ((uintptr_t)workspace + 255) & ~(uintptr_t)255;
    R.pts = (float4*)rbase;
    R.keys = (uint32_t*)(rbase + (size_t)total * sizeof(float4));
    R.m = m;
    return fps_rows_launch(R, b, plan, (hipStream_t)stream);
  }
  FpsParams P{};
  long total = 0;
  for (int i = 0; i < b; ++i) {
    FpsScene& S = P.scenes[i];
    S.xyz = xyz[i];
    S.n = counts[i];
    fps_geometry(S.n, &S.npad, &S.bucket_pts, &S.nbuckets);
    S.ws_off = total;
    total += S.npad;
    S.ref_log2 = ref_log2_of(S.n);
    S.ref_block = 1 << S.ref_log2;
  }
  uintptr_t base = ((uintptr_t)workspace + 255) & ~(uintptr_t)255;
  P.pts = (float4*)base;
  P.keys = (uint32_t*)(base + (size_t)total * sizeof(float4));
  P.idx = idx; P.m = m; P.nscenes = b;
  hipLaunchKernelGGL(fps_kernel<false>, dim3(b), dim3(kFpsThreads), 0, (hipStream_t)stream, P);
  return check_launch("furthest_point_sampling_varlen");
}
