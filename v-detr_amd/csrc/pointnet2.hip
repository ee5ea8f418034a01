// pointnet2.hip — gather / group / three_nn / three_interpolate / ball_query for gfx950.
//
// MI355X-native re-design of the indexing ops behind the reference's `pointnet2._ext` module
// (third_party/pointnet2/_ext_src/src/bindings.cpp:9-22).  The reference launches ONE block per batch
// element for most of these (group_points_gpu.cu:37, interpolate_gpu.cu:69,110, ball_query_gpu.cu:53),
// which leaves 255 of 256 CUs idle at batch 1.  Here every op is a flat, coalesced, chip-filling grid:
// consecutive lanes own consecutive OUTPUT elements (64 x 4 B = one 256-B line per wave store), the
// index / weight rows are read once per thread and reused across a channel strip, and the scans
// (ball query, three_nn) are wave-cooperative.  All of them are HBM/L2-bound integer/byte work.  Round 6: the ball query's
// cloud goes through LDS tiles shared by eight queries, and the two scattering gradients accumulate a channel row in LDS.
#include "common.h"
#include "wave.h"

namespace vdetr {

constexpr int kThreads = 256;
constexpr int kChanStrip = 8;  // channels handled per thread: amortises the idx/weight loads

// ---------------------------------------------------------------------------------------------
// gather_points: out[b,c,j] = points[b,c,idx[b,j]]                      (sampling_gpu.cu:11-23)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void gather_points_kernel(const float* __restrict__ points,
                                                                 const int32_t* __restrict__ idx,
                                                                 float* __restrict__ out, int c, int n,
                                                                 int m) {
  const int j = blockIdx.x * kThreads + threadIdx.x;
  const int bi = blockIdx.z;
  if (j >= m) return;
  const int a = idx[(size_t)bi * m + j];
  const int l0 = blockIdx.y * kChanStrip;
#pragma unroll
  for (int dl = 0; dl < kChanStrip; ++dl) {
    const int l = l0 + dl;
    if (l < c) out[((size_t)bi * c + l) * m + j] = points[((size_t)bi * c + l) * n + a];
  }
}

// The same with the channel row streamed through LDS: one workgroup per (channel, batch element) reads its row of n floats ONCE,
// coalesced, in chunks of 80 KB (one workgroup each), and serves the m sampled positions that fall into the chunk out of LDS.  The form above reads one
// 128-byte line per sampled element: 134 MB of lines for 4 MB of values at c = 256, n = 40k, m = 4096 (17.6 us); the rows themselves
// are 41 MB.  Worth it when the samples are a fair share of the row (m >= n / 32) and there are workgroups enough to fill the chip.
constexpr int kGpThreads = 512;
constexpr int kGpChunk = 20480;  // floats per LDS chunk (80 KB: two workgroups per CU)
__global__ __launch_bounds__(kGpThreads) void gather_points_lds_kernel(const float* __restrict__ points, const int32_t* __restrict__ idx,
                                                                       float* __restrict__ out, int c, int n, int m) {
  extern __shared__ __attribute__((aligned(16))) float chunk[];
  const int l = blockIdx.x, bi = blockIdx.z;
  const float* row = points + ((size_t)bi * c + l) * n;
  const int32_t* ix = idx + (size_t)bi * m;
  float* orow = out + ((size_t)bi * c + l) * m;
  const bool vec = ((((uintptr_t)row) & 15) == 0);
  const int off = blockIdx.y * kGpChunk;  // one workgroup per chunk of the row: two of them share a CU, one stages while the other serves
  const int len = min(kGpChunk, n - off);
  if (vec) {
    const float4* s4 = reinterpret_cast<const float4*>(row + off);
    float4* c4 = reinterpret_cast<float4*>(chunk);
    for (int e = threadIdx.x; e < (len >> 2); e += kGpThreads) c4[e] = s4[e];
    for (int e = (len & ~3) + threadIdx.x; e < len; e += kGpThreads) chunk[e] = row[off + e];
  } else {
    for (int e = threadIdx.x; e < len; e += kGpThreads) chunk[e] = row[off + e];
  }
  __syncthreads();
  for (int j = threadIdx.x; j < m; j += kGpThreads) {
    const int a = ix[j] - off;
    if (a >= 0 && a < len) orow[j] = chunk[a];
  }
}

// gather_points_grad: grad_points[b,c,idx[b,j]] += grad_out[b,c,j]       (sampling_gpu.cu:37-50)
__global__ __launch_bounds__(kThreads) void gather_points_grad_kernel(
    const float* __restrict__ grad_out, const int32_t* __restrict__ idx, float* __restrict__ grad_points,
    int c, int n, int m) {
  const int j = blockIdx.x * kThreads + threadIdx.x;
  const int bi = blockIdx.z;
  if (j >= m) return;
  const int a = idx[(size_t)bi * m + j];
  const int l0 = blockIdx.y * kChanStrip;
#pragma unroll
  for (int dl = 0; dl < kChanStrip; ++dl) {
    const int l = l0 + dl;
    if (l < c)
      unsafeAtomicAdd(grad_points + ((size_t)bi * c + l) * n + a, grad_out[((size_t)bi * c + l) * m + j]);
  }
}

// ---------------------------------------------------------------------------------------------
// gather_rows: out[b,j,:] = rows_b[idx[b,j], :]  for point-major tables rows_b [n_b, c], one table per scene.
// The backbone hands over voxel coordinates [n,3] and features [n,256] point-major (out.C / out.F,
// model_vdetr.py:279-280); the reference transposes the features to (1,256,n) just to gather 4096 columns of it with a
// stride of n floats (model_vdetr.py:22-34, sampling_gpu.cu:11-23) and transposes the result again for the decoder.
// Gathering ROWS moves 1 KB contiguous per sampled point, needs no transposed copy of the 41 MB table in either
// direction, and takes scenes of different sizes in one launch (per-scene base pointers travel as kernel arguments).
// ---------------------------------------------------------------------------------------------
constexpr int kRowScenes = 32;
struct RowTables { const float* src[kRowScenes]; };
struct RowGradTables { float* dst[kRowScenes]; };

template <int VEC>
__global__ __launch_bounds__(kThreads) void gather_rows_kernel(RowTables T, const int32_t* __restrict__ idx,
                                                               float* __restrict__ out, int c, int m) {
  const int bi = blockIdx.y;
  const int cv = c / VEC;
  const long e = (long)blockIdx.x * kThreads + threadIdx.x;
  if (e >= (long)m * cv) return;
  const int j = (int)(e / cv), q = (int)(e - (long)j * cv);
  const long a = idx[(size_t)bi * m + j];
  const float* __restrict__ src = T.src[bi] + a * c + q * VEC;
  float* __restrict__ dst = out + ((size_t)bi * m + j) * c + q * VEC;
  if (VEC == 4) {
    *reinterpret_cast<float4*>(dst) = *reinterpret_cast<const float4*>(src);
  } else {
    *dst = *src;
  }
}

// grad_rows_b[idx[b,j], :] += grad_out[b,j,:]   (atomics: a sampling may repeat an index)
template <int VEC>
__global__ __launch_bounds__(kThreads) void gather_rows_grad_kernel(RowGradTables T, const int32_t* __restrict__ idx,
                                                                    const float* __restrict__ grad_out, int c, int m) {
  const int bi = blockIdx.y;
  const int cv = c / VEC;
  const long e = (long)blockIdx.x * kThreads + threadIdx.x;
  if (e >= (long)m * cv) return;
  const int j = (int)(e / cv), q = (int)(e - (long)j * cv);
  const long a = idx[(size_t)bi * m + j];
  float* __restrict__ dst = T.dst[bi] + a * c + q * VEC;
  const float* __restrict__ src = grad_out + ((size_t)bi * m + j) * c + q * VEC;
#pragma unroll
  for (int t = 0; t < VEC; ++t) unsafeAtomicAdd(dst + t, src[t]);
}

// ---------------------------------------------------------------------------------------------
// group_points: out[b,c,j,k] = points[b,c,idx[b,j,k]]                 (group_points_gpu.cu:11-31)
// flat element e = j*nsample + k, so the (b,npoints,nsample) index tensor is read coalesced.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void group_points_kernel(const float* __restrict__ points,
                                                                const int32_t* __restrict__ idx,
                                                                float* __restrict__ out, int c, int n,
                                                                long ne) {
  const long e = (long)blockIdx.x * kThreads + threadIdx.x;
  const int bi = blockIdx.z;
  if (e >= ne) return;
  const int a = idx[(size_t)bi * ne + e];
  const int l0 = blockIdx.y * kChanStrip;
#pragma unroll
  for (int dl = 0; dl < kChanStrip; ++dl) {
    const int l = l0 + dl;
    if (l < c) out[((size_t)bi * c + l) * ne + e] = points[((size_t)bi * c + l) * n + a];
  }
}

// The same with the channel row staged in LDS (round 6): the form above reads one 64-byte sector per output element (67 MB of
// output, 128 x 2048 x 64, out of a 20 MB table: 30.7 us).  Workgroup (channel l, slice s of the index tensor, batch element) reads
// its row of n <= 40,000 floats once, coalesced, and serves its slice of the (point, sample) pairs out of LDS — four per lane, 16-byte
// index loads and output stores; equal indices inside a load (ball_query's padding) are a broadcast.  The slices exist to fill the
// chip when c < 256: each re-reads the row (158 KB, out of L2 after the first).
constexpr int kGrThreads = 1024;
constexpr int kGrRow = 40000;  // floats of a row in LDS
__global__ __launch_bounds__(kGrThreads) void group_points_lds_kernel(const float* __restrict__ points, const int32_t* __restrict__ idx,
                                                                      float* __restrict__ out, int c, int n, long ne, long per) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int l = blockIdx.x, bi = blockIdx.z;
  const float* src = points + ((size_t)bi * c + l) * n;
  const int sh = (int)(((uintptr_t)src >> 2) & 3);  // the row at its offset inside its 16-byte line: global quads == LDS quads
  float* row = lds + sh;
  const int head = min((4 - sh) & 3, n), quads = (n - head) >> 2;
  if (threadIdx.x < head) row[threadIdx.x] = src[threadIdx.x];
  for (int e = threadIdx.x; e < quads; e += kGrThreads)
    *reinterpret_cast<float4*>(row + head + 4 * e) = *reinterpret_cast<const float4*>(src + head + 4 * e);
  for (int e = head + 4 * quads + threadIdx.x; e < n; e += kGrThreads) row[e] = src[e];
  __syncthreads();
  const int32_t* ix = idx + (size_t)bi * ne;
  float* dst = out + ((size_t)bi * c + l) * ne;
  const long e0 = (long)blockIdx.y * per, e1 = min(ne, e0 + per);  // `per` is a multiple of 4
  if ((ne & 3) == 0 && (((uintptr_t)idx | (uintptr_t)out) & 15) == 0) {
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    for (long q = (e0 >> 2) + threadIdx.x; q < (e1 >> 2); q += kGrThreads) {
      const i32x4 a = reinterpret_cast<const i32x4*>(ix)[q];
      f32x4 v;
      v[0] = row[a[0]]; v[1] = row[a[1]]; v[2] = row[a[2]]; v[3] = row[a[3]];
      __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(dst) + q);
    }
  } else {
    for (long e = e0 + threadIdx.x; e < e1; e += kGrThreads) dst[e] = row[ix[e]];
  }
}

// group_points_grad: grad_points[b,c,idx[b,j,k]] += grad_out[b,c,j,k] (group_points_gpu.cu:46-67)
// One WAVE per (point j, strip of 16 channels); lane k owns sample k of the row, so a channel's 64 gradients are ONE coalesced
// 256-byte load (round 1's thread-per-row walk read them at a 256-byte lane stride: 67 MB at 0.25 TB/s, 273 us).  Duplicates are
// the norm here — ball_query pads a row with its first hit, and equal addresses inside one atomic instruction serialise — so the
// lanes whose index equals the row's first one (the first hit and all the padding) are summed across the wave and leave as a
// single atomic of lane 0; every other lane adds its own value.  Correct for any index row (other duplicates meet in the atomics).
constexpr int kGgStrip = 16;
__global__ __launch_bounds__(kThreads) void group_points_grad_kernel(
    const float* __restrict__ grad_out, const int32_t* __restrict__ idx, float* __restrict__ grad_points,
    int c, int n, int npoints, int nsample) {
  const int lane = threadIdx.x & (kWave - 1);
  const int j = blockIdx.x * (kThreads / kWave) + (threadIdx.x >> 6);
  const int bi = blockIdx.z;
  if (j >= npoints) return;
  const int32_t* row = idx + ((size_t)bi * npoints + j) * nsample;
  const int a0 = row[0];
  const int l0 = blockIdx.y * kGgStrip;
  for (int k0 = 0; k0 < nsample; k0 += kWave) {
    const int k = k0 + lane;
    const bool live = k < nsample;
    const int a = live ? row[k] : a0;
    const bool first = a == a0;  // the first hit or padding
    float g[kGgStrip];
#pragma unroll
    for (int dl = 0; dl < kGgStrip; ++dl) {
      const int l = l0 + dl;
      g[dl] = (live && l < c) ? grad_out[(((size_t)bi * c + l) * npoints + j) * nsample + k] : 0.f;
    }
#pragma unroll
    for (int dl = 0; dl < kGgStrip; ++dl) {
      const int l = l0 + dl;
      if (l >= c) break;
      float* gp = grad_points + ((size_t)bi * c + l) * n;
      const float s = wave_allsum_f32(first ? g[dl] : 0.f);
      if (lane == 0) unsafeAtomicAdd(gp + a0, s);
      if (live && !first) unsafeAtomicAdd(gp + a, g[dl]);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// The two scattering gradients without global atomics and without the caller's zero-fill (round 6): a channel's row of
// grad_points (n floats: 158 KB at n = 40k) is accumulated in LDS by ONE workgroup and written once, coalesced.
//   gather_points_grad, c = 256, n = 40k, m = 4096: 41 MB of zeros written by a memset, then 1 M global atomics (51 us + the memset)
//   group_points_grad,  c = 128, n = 40k, 2048 x 64: 41 MB memset (b*c*n) + ~9 M global float atomics (126 us)
// Here workgroup (channel l, range r, batch element) zeroes its `len` floats of LDS, walks ALL ne (index, gradient) pairs of the
// channel in 64-pair steps — `ds_add_f32` for the pairs whose index falls into its range — and stores the range.  With DEDUP the
// lanes of a step whose index equals lane 0's (ball_query pads a row with its first hit: most of a 64-sample row) are summed across
// the wave first and leave as one add: equal addresses inside one LDS atomic instruction serialise.  Any index row is handled
// correctly (other duplicates meet in the LDS atomics).  The sum's order differs from the atomics' (which had none to speak of).
// ---------------------------------------------------------------------------------------------
constexpr int kScThreads = 1024;
constexpr int kScRange = 40000;  // floats of a row per workgroup (160,000 of the CU's 163,840 bytes)
template <bool DEDUP>
__global__ __launch_bounds__(kScThreads) void scatter_rows_lds_kernel(const float* __restrict__ grad_out, const int32_t* __restrict__ idx,
                                                                      float* __restrict__ grad_points, int c, int n, long ne, int range) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int l = blockIdx.x, bi = blockIdx.z, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int off = blockIdx.y * range, len = min(range, n - off);
  float* row = grad_points + ((size_t)bi * c + l) * n + off;
  // element i of the range lives at lds[sh + i], sh = the row's offset inside its 16-byte line: LDS quads and global quads line up
  const int sh = (int)(((uintptr_t)row >> 2) & 3);
  float* acc = lds + sh;
  for (int e = threadIdx.x; e < ((sh + len + 3) >> 2); e += kScThreads) reinterpret_cast<float4*>(lds)[e] = make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();
  const float* g = grad_out + ((size_t)bi * c + l) * ne;
  const int32_t* ix = idx + (size_t)bi * ne;
  constexpr int kWaves = kScThreads / kWave;
  if ((ne & 3) == 0 && (((uintptr_t)grad_out | (uintptr_t)idx) & 15) == 0) {
    // four consecutive pairs per lane (16-byte loads), two such steps in flight.  The walk is instruction-bound — at c = 128 half the
    // CUs carry 131k pairs each — so the duplicates are folded where it is cheapest: inside the lane (pairs 1-3 that repeat pair 0's
    // index join its value), then across the 16 lanes of a DPP row (64 pairs: one ball-query row) for the lanes that start with the
    // row's first index; what is left goes to LDS pair by pair.
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const i32x4* ix4 = reinterpret_cast<const i32x4*>(ix);
    const f32x4* g4 = reinterpret_cast<const f32x4*>(g);
    const long nq = ne >> 2;
    constexpr int kV = 2;
    for (long q0 = (long)wave * kWave; q0 < nq; q0 += (long)kWaves * kWave * kV) {
      i32x4 a[kV];
      f32x4 v[kV];
#pragma unroll
      for (int u = 0; u < kV; ++u) {
        const long q = q0 + (long)u * kWaves * kWave + lane;
        a[u] = q < nq ? ix4[q] : i32x4{-1, -1, -1, -1};
        v[u] = q < nq ? g4[q] : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int u = 0; u < kV; ++u) {
        const int a0 = a[u][0];
        float v0 = v[u][0];
        if (DEDUP) {
#pragma unroll
          for (int t = 1; t < 4; ++t) v0 += a[u][t] == a0 ? v[u][t] : 0.f;
          const int r0 = __shfl(a0, lane & ~15, 64);
          const bool same = a0 == r0;
          const float s = row_allsum_f32(same ? v0 : 0.f);
          if ((lane & 15) == 0) v0 = s;  // the row's leader carries the row's sum (it is `same` by construction)
          else if (same) v0 = 0.f, a[u][0] = -1;
        }
        if ((unsigned)(a[u][0] - off) < (unsigned)len) unsafeAtomicAdd(acc + (a[u][0] - off), v0);
#pragma unroll
        for (int t = 1; t < 4; ++t)
          if ((!DEDUP || a[u][t] != a0) && (unsigned)(a[u][t] - off) < (unsigned)len) unsafeAtomicAdd(acc + (a[u][t] - off), v[u][t]);
      }
    }
  } else {
    constexpr int kU = 4;  // 64-pair steps in flight per wave
    for (long e0 = (long)wave * kWave; e0 < ne; e0 += (long)kWaves * kWave * kU) {
      int a[kU];
      float v[kU];
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const long e = e0 + (long)u * kWaves * kWave + lane;
        a[u] = e < ne ? ix[e] - off : -1;
        v[u] = e < ne ? g[e] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const bool in = (unsigned)a[u] < (unsigned)len;
        if (DEDUP) {
          const int a0 = __builtin_amdgcn_readfirstlane(a[u]);
          const bool same = a[u] == a0;
          const float s = wave_allsum_f32(same ? v[u] : 0.f);
          if (lane == 0 && in) unsafeAtomicAdd(acc + a0, s);
          if (in && !same) unsafeAtomicAdd(acc + a[u], v[u]);
        } else if (in) {
          unsafeAtomicAdd(acc + a[u], v[u]);
        }
      }
    }
  }
  __syncthreads();
  const int head = min((4 - sh) & 3, len), quads = (len - head) >> 2;
  if (threadIdx.x < head) row[threadIdx.x] = acc[threadIdx.x];
  for (int e = threadIdx.x; e < quads; e += kScThreads)
    *reinterpret_cast<float4*>(row + head + 4 * e) = *reinterpret_cast<const float4*>(acc + head + 4 * e);
  for (int e = head + 4 * quads + threadIdx.x; e < len; e += kScThreads) row[e] = acc[e];
}

// ---------------------------------------------------------------------------------------------
// three_interpolate: out[b,c,j] = p[i1]*w1 + p[i2]*w2 + p[i3]*w3      (interpolate_gpu.cu:75-104)
// Contraction order pinned as t = p2*w2; t = fma(p1,w1,t); t = fma(p3,w3,t) (see common.h).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void three_interpolate_kernel(
    const float* __restrict__ points, const int32_t* __restrict__ idx, const float* __restrict__ weight,
    float* __restrict__ out, int c, int m, int n) {
  const int j = blockIdx.x * kThreads + threadIdx.x;
  const int bi = blockIdx.z;
  if (j >= n) return;
  const size_t r = ((size_t)bi * n + j) * 3;
  const int i1 = idx[r], i2 = idx[r + 1], i3 = idx[r + 2];
  const float w1 = weight[r], w2 = weight[r + 1], w3 = weight[r + 2];
  const int l0 = blockIdx.y * kChanStrip;
#pragma unroll
  for (int dl = 0; dl < kChanStrip; ++dl) {
    const int l = l0 + dl;
    if (l < c) {
      const float* p = points + ((size_t)bi * c + l) * m;
      float t = __fmul_rn(p[i2], w2);
      t = __fmaf_rn(p[i1], w1, t);
      t = __fmaf_rn(p[i3], w3, t);
      out[((size_t)bi * c + l) * n + j] = t;
    }
  }
}

// The same through LDS, for known-point sets whose channel rows fit: a workgroup stages kTiStrip channel rows of `points` (m floats
// each, read coalesced ONCE) and every thread then interpolates its point j in all of them out of LDS.  The global form above reads
// 4 bytes per (channel, neighbour) at a stride of m floats (one 128-B line per element: 14 us at c = 256, n = 2048, m = 1024 for
// 3 MB of tensors).  Same contraction order.
constexpr int kTiStrip = 16;
__global__ __launch_bounds__(kThreads) void three_interpolate_lds_kernel(
    const float* __restrict__ points, const int32_t* __restrict__ idx, const float* __restrict__ weight,
    float* __restrict__ out, int c, int m, int n) {
  extern __shared__ __attribute__((aligned(16))) float rows[];  // [kTiStrip][m]
  const int bi = blockIdx.z, l0 = blockIdx.y * kTiStrip;
  const int nl = min(kTiStrip, c - l0);
  const float* src = points + ((size_t)bi * c + l0) * m;  // the strip's rows are contiguous
  if ((((uintptr_t)src) & 15) == 0 && ((nl * m) & 3) == 0) {  // 16 bytes per lane: a strip is 64 KB
    const float4* s4 = reinterpret_cast<const float4*>(src);
    float4* r4 = reinterpret_cast<float4*>(rows);
    for (int e = threadIdx.x; e < (nl * m) >> 2; e += kThreads) r4[e] = s4[e];
  } else {
    for (int e = threadIdx.x; e < nl * m; e += kThreads) rows[e] = src[e];
  }
  __syncthreads();
  for (int j = blockIdx.x * kThreads + threadIdx.x; j < n; j += gridDim.x * kThreads) {
    const size_t r = ((size_t)bi * n + j) * 3;
    const int i1 = idx[r], i2 = idx[r + 1], i3 = idx[r + 2];
    const float w1 = weight[r], w2 = weight[r + 1], w3 = weight[r + 2];
#pragma unroll 4
    for (int dl = 0; dl < nl; ++dl) {
      const float* p = rows + dl * m;
      float t = __fmul_rn(p[i2], w2);
      t = __fmaf_rn(p[i1], w1, t);
      t = __fmaf_rn(p[i3], w3, t);
      out[((size_t)bi * c + l0 + dl) * n + j] = t;
    }
  }
}

// three_interpolate_grad: grad_points[b,c,i_t] += grad_out[b,c,j]*w_t (interpolate_gpu.cu:119-146)
__global__ __launch_bounds__(kThreads) void three_interpolate_grad_kernel(
    const float* __restrict__ grad_out, const int32_t* __restrict__ idx, const float* __restrict__ weight,
    float* __restrict__ grad_points, int c, int n, int m) {
  const int j = blockIdx.x * kThreads + threadIdx.x;
  const int bi = blockIdx.z;
  if (j >= n) return;
  const size_t r = ((size_t)bi * n + j) * 3;
  const int i1 = idx[r], i2 = idx[r + 1], i3 = idx[r + 2];
  const float w1 = weight[r], w2 = weight[r + 1], w3 = weight[r + 2];
  const int l0 = blockIdx.y * kChanStrip;
#pragma unroll
  for (int dl = 0; dl < kChanStrip; ++dl) {
    const int l = l0 + dl;
    if (l < c) {
      const float g = grad_out[((size_t)bi * c + l) * n + j];
      float* gp = grad_points + ((size_t)bi * c + l) * m;
      unsafeAtomicAdd(gp + i1, __fmul_rn(g, w1));
      unsafeAtomicAdd(gp + i2, __fmul_rn(g, w2));
      unsafeAtomicAdd(gp + i3, __fmul_rn(g, w3));
    }
  }
}

// ---------------------------------------------------------------------------------------------
// three_nn: 3 nearest `known` points of every `unknown` point         (interpolate_gpu.cu:12-62)
// The reference's sequential insertion with strict '<' returns the three smallest entries under the order
// (distance, index) — the earlier index wins ties — so the scan can be split: one WAVE per unknown point, lane l scans
// known points l, l+64, ... keeping its own top-3, then three rounds of "wave arg-min, winner pops its head" merge the
// 64 lists.  64x the parallelism of the reference's one-thread-per-point scan at identical results.
// `double bestN = 1e40` sentinels (-> +inf when stored to float, index 0) are float +inf here: a real distance of
// +inf or NaN is never inserted by the reference ('<' against 1e40 / NaN is false) and is skipped here too.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void three_nn_kernel(const float* __restrict__ unknown,
                                                            const float* __restrict__ known,
                                                            float* __restrict__ dist2,
                                                            int32_t* __restrict__ idx, int n, int m) {
  const int lane = threadIdx.x & 63;
  const int j = blockIdx.x * (kThreads / kWave) + (threadIdx.x >> 6);
  const int bi = blockIdx.z;
  if (j >= n) return;  // wave-uniform
  const float* u = unknown + ((size_t)bi * n + j) * 3;
  const float ux = u[0], uy = u[1], uz = u[2];
  const float* kb = known + (size_t)bi * m * 3;
  float b1 = INFINITY, b2 = INFINITY, b3 = INFINITY;
  int i1 = 0x7FFFFFFF, i2 = 0x7FFFFFFF, i3 = 0x7FFFFFFF;
  for (int k = lane; k < m; k += kWave) {
    const float d = sqdist3(ux - kb[k * 3], uy - kb[k * 3 + 1], uz - kb[k * 3 + 2]);
    if (d < b1) {
      b3 = b2; i3 = i2; b2 = b1; i2 = i1; b1 = d; i1 = k;
    } else if (d < b2) {
      b3 = b2; i3 = i2; b2 = d; i2 = k;
    } else if (d < b3) {
      b3 = d; i3 = k;
    }
  }
  float od[3];
  int oi[3];
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    // wave arg-min of (b1, i1): distance bits are monotone for d >= 0; +inf heads (empty lists) never win over a finite one
    const unsigned db = __float_as_uint(b1);
    const unsigned dmin = ~wave_allmax_u32(~db);
    const unsigned imin = ~wave_allmax_u32(db == dmin ? ~(unsigned)i1 : 0u);
    const bool none = dmin >= 0x7F800000u;  // every list is exhausted (m < 3)
    od[t] = none ? INFINITY : __uint_as_float(dmin);
    oi[t] = none ? 0 : (int)imin;
    if (!none && db == dmin && (unsigned)i1 == imin) {  // the winner pops its head
      b1 = b2; i1 = i2; b2 = b3; i2 = i3; b3 = INFINITY; i3 = 0x7FFFFFFF;
    }
  }
  if (lane == 0) {
    const size_t r = ((size_t)bi * n + j) * 3;
    dist2[r] = od[0]; dist2[r + 1] = od[1]; dist2[r + 2] = od[2];
    idx[r] = oi[0]; idx[r + 1] = oi[1]; idx[r + 2] = oi[2];
  }
}

// ---------------------------------------------------------------------------------------------
// ball_query: first `nsample` indices (ascending) with d² < r²          (ball_query_gpu.cu:12-47)
// One WAVE per query, eight queries per workgroup, the cloud streamed through LDS in tiles of 2048 points that all eight share
// (round 6: one wave per query reading the cloud itself moved 12nm bytes through L2 — 983 MB for 2048 queries x 40k points,
// 135 us; the tile cuts that 8x and the next tile's loads are in flight while this one is scanned).  Each step tests 64
// consecutive points, a 64-bit ballot + prefix popcount compacts the hits in ascending index order, and a wave stops scanning as
// soon as it has nsample hits (it keeps loading tiles for the others; the workgroup stops when all eight have); four such steps are issued
// together (hits past the nsample-th are dropped by position, as the reference's loop exit does).  Unfilled slots are
// set to the first hit afterwards, which is what the reference's "first hit pre-fills the row" produces; rows with no hit are left
// untouched (zero from the caller).
// ---------------------------------------------------------------------------------------------
constexpr int kBqThreads = 512, kBqQueries = kBqThreads / kWave, kBqTile = 4096;
constexpr int kBqPer = kBqTile * 3 / kBqThreads;  // floats of a tile per thread
constexpr int kBqUnroll = 4;                      // 64-point steps in flight per wave (two waves per SIMD: the loads need the overlap)

__global__ __launch_bounds__(kBqThreads) void ball_query_kernel(const float* __restrict__ new_xyz,
                                                                const float* __restrict__ xyz,
                                                                int32_t* __restrict__ idx, int n, int m,
                                                                float radius2, int nsample) {
  __shared__ float tile[(kBqTile + kBqUnroll * kWave) * 3];  // + one step of pad, read ahead and never used
  __shared__ int full;  // waves whose row is complete
  const int lane = threadIdx.x & 63;
  const int j = blockIdx.x * kBqQueries + (threadIdx.x >> 6);
  const int bi = blockIdx.z;
  const bool live = j < m;  // wave-uniform
  const float* q = new_xyz + ((size_t)bi * m + (live ? j : 0)) * 3;
  const float qx = q[0], qy = q[1], qz = q[2];
  const float* pts = xyz + (size_t)bi * n * 3;
  int32_t* row = idx + ((size_t)bi * m + (live ? j : 0)) * nsample;
  const long total = (long)n * 3;
  int cnt = live ? 0 : nsample, first = -1;
  bool told = false;
  if (threadIdx.x == 0) full = 0;
  float nxt[kBqPer];
  auto fetch = [&](int t0) {
#pragma unroll
    for (int u = 0; u < kBqPer; ++u) {
      const long e = (long)t0 * 3 + u * kBqThreads + threadIdx.x;
      nxt[u] = e < total ? pts[e] : 3e38f;  // past the cloud's end: a point no radius reaches (d2 = inf), so the scan needs no index test
    }
  };
  fetch(0);
  for (int t0 = 0; t0 < n; t0 += kBqTile) {
#pragma unroll
    for (int u = 0; u < kBqPer; ++u) tile[u * kBqThreads + threadIdx.x] = nxt[u];
    __syncthreads();
    if (t0 + kBqTile < n) fetch(t0 + kBqTile);
    const int tn = min(kBqTile, n - t0);
    // the tile is a multiple of the step; the next step's points are read from LDS (into the other register set) before this
    // step's are tested: two waves per SIMD do not cover the read latency by themselves
    constexpr int kStep = kBqUnroll * kWave;
    float pa[3][kBqUnroll], pb[3][kBqUnroll];
    auto read = [&](float (&p)[3][kBqUnroll], int k0) {
#pragma unroll
      for (int u = 0; u < kBqUnroll; ++u) {
        const int k = k0 + u * kWave + lane;
        p[0][u] = tile[k * 3]; p[1][u] = tile[k * 3 + 1]; p[2][u] = tile[k * 3 + 2];
      }
    };
    auto step = [&](const float (&p)[3][kBqUnroll], float (&q)[3][kBqUnroll], int k0) {
      float d2[kBqUnroll];
#pragma unroll
      for (int u = 0; u < kBqUnroll; ++u) d2[u] = sqdist3(qx - p[0][u], qy - p[1][u], qz - p[2][u]);
      read(q, k0 + kStep);  // past the tile's end on its last step: the pad (never tested: the loop ends)
      unsigned long long mask[kBqUnroll], any = 0;
#pragma unroll
      for (int u = 0; u < kBqUnroll; ++u) {
        mask[u] = __ballot(d2[u] < radius2);
        any |= mask[u];
      }
      if (any == 0) return;
#pragma unroll
      for (int u = 0; u < kBqUnroll; ++u) {
        if (mask[u] == 0) continue;
        const int base = t0 + k0 + u * kWave;
        if (first < 0) first = base + __ffsll((long long)mask[u]) - 1;
        const int pos = cnt + __popcll(mask[u] & ((1ull << lane) - 1ull));
        if (((mask[u] >> lane) & 1ull) && pos < nsample) row[pos] = base + lane;
        cnt += __popcll(mask[u]);
      }
    };
    if (cnt < nsample) read(pa, 0);
    for (int k0 = 0; k0 < tn && cnt < nsample; k0 += 2 * kStep) {
      step(pa, pb, k0);
      if (k0 + kStep >= tn || cnt >= nsample) break;
      step(pb, pa, k0 + kStep);
    }
    if (cnt >= nsample && !told) {
      told = true;
      if (lane == 0) atomicAdd(&full, 1);
    }
    __syncthreads();  // the tile may be overwritten; `full` is final for this tile
    if (full == kBqQueries) break;
  }
  if (live && first >= 0)
    for (int l = min(cnt, nsample) + lane; l < nsample; l += kWave) row[l] = first;
}

}  // namespace vdetr

using namespace vdetr;

static int check_bcnm(const char* op, int b, int c, int n, int m) {
  VDETR_REQUIRE(b >= 0 && c >= 0 && n >= 0 && m >= 0, "%s: negative dimension", op);
  VDETR_REQUIRE(b <= 65535, "%s: batch %d > 65535", op, b);
  return VDETR_OK;
}

extern "C" int vdetr_gather_points_f32(const float* points, const int32_t* idx, float* out, int b, int c,
                                       int n, int m, vdetr_stream_t stream) {
  if (int e = check_bcnm("gather_points", b, c, n, m)) return e;
  if (b == 0 || c == 0 || m == 0) return VDETR_OK;
  VDETR_REQUIRE(points && idx && out, "gather_points: null pointer");
  if ((long)m * 32 >= n && (long)c * b >= 64 && b <= 65535) {  // the rows through LDS (see the kernel)
    const size_t lds = (size_t)(n < kGpChunk ? n : kGpChunk) * sizeof(float);
    if (int e = set_lds(gather_points_lds_kernel, lds, "gather_points")) return e;
    hipLaunchKernelGGL(gather_points_lds_kernel, dim3(c, ceil_div(n, kGpChunk), b), dim3(kGpThreads), lds, (hipStream_t)stream, points, idx, out, c, n, m);
    return check_launch("gather_points");
  }
  dim3 grid(ceil_div(m, kThreads), ceil_div(c, kChanStrip), b);
  hipLaunchKernelGGL(gather_points_kernel, grid, dim3(kThreads), 0, (hipStream_t)stream, points, idx, out,
                     c, n, m);
  return check_launch("gather_points");
}

extern "C" int vdetr_gather_points_grad_f32(const float* grad_out, const int32_t* idx, float* grad_points,
                                            int b, int c, int n, int m, vdetr_stream_t stream) {
  if (int e = check_bcnm("gather_points_grad", b, c, n, m)) return e;
  if (b == 0 || c == 0 || m == 0) return VDETR_OK;
  VDETR_REQUIRE(grad_out && idx && grad_points, "gather_points_grad: null pointer");
  dim3 grid(ceil_div(m, kThreads), ceil_div(c, kChanStrip), b);
  hipLaunchKernelGGL(gather_points_grad_kernel, grid, dim3(kThreads), 0, (hipStream_t)stream, grad_out,
                     idx, grad_points, c, n, m);
  return check_launch("gather_points_grad");
}

extern "C" int vdetr_group_points_f32(const float* points, const int32_t* idx, float* out, int b, int c,
                                      int n, int npoints, int nsample, vdetr_stream_t stream) {
  if (int e = check_bcnm("group_points", b, c, n, npoints)) return e;
  VDETR_REQUIRE(nsample >= 0, "group_points: negative nsample");
  const long ne = (long)npoints * nsample;
  if (b == 0 || c == 0 || ne == 0) return VDETR_OK;
  VDETR_REQUIRE(points && idx && out, "group_points: null pointer");
  if (n <= kGrRow && ne * 8 >= n && (long)b * c >= 32) {  // the row through LDS (see the kernel): pairs enough to pay for staging it
    long slices = ceil_div(256, (long)b * c);
    slices = slices < 1 ? 1 : (slices > 8 ? 8 : slices);
    const long per = ((ne + slices - 1) / slices + 3) & ~3L;
    const size_t lds = (size_t)(n + 4) * sizeof(float);
    if (int e = set_lds(group_points_lds_kernel, lds, "group_points")) return e;
    hipLaunchKernelGGL(group_points_lds_kernel, dim3(c, ceil_div(ne, per), b), dim3(kGrThreads), lds, (hipStream_t)stream, points, idx, out, c, n, ne, per);
    return check_launch("group_points");
  }
  dim3 grid(ceil_div(ne, kThreads), ceil_div(c, kChanStrip), b);
  hipLaunchKernelGGL(group_points_kernel, grid, dim3(kThreads), 0, (hipStream_t)stream, points, idx, out, c,
                     n, ne);
  return check_launch("group_points");
}

extern "C" int vdetr_group_points_grad_f32(const float* grad_out, const int32_t* idx, float* grad_points,
                                           int b, int c, int n, int npoints, int nsample,
                                           vdetr_stream_t stream) {
  if (int e = check_bcnm("group_points_grad", b, c, n, npoints)) return e;
  VDETR_REQUIRE(nsample >= 0, "group_points_grad: negative nsample");
  const long ne = (long)npoints * nsample;
  if (b == 0 || c == 0 || ne == 0) return VDETR_OK;
  VDETR_REQUIRE(grad_out && idx && grad_points, "group_points_grad: null pointer");
  dim3 grid(ceil_div(npoints, kThreads / kWave), ceil_div(c, kGgStrip), b);
  hipLaunchKernelGGL(group_points_grad_kernel, grid, dim3(kThreads), 0, (hipStream_t)stream, grad_out, idx,
                     grad_points, c, n, npoints, nsample);
  return check_launch("group_points_grad");
}

// the LDS form fills the chip when there are channel rows enough; below that the memset + atomics pair is the better one
static bool scatter_lds_pays(int b, int c) { return (long)b * c >= 64; }

template <bool DEDUP>
static int scatter_rows_lds(const char* op, const float* grad_out, const int32_t* idx, float* grad_points, int b, int c, int n, long ne,
                            hipStream_t stream) {
  const int ranges = ceil_div(n, kScRange);
  const int range = (ceil_div(n, ranges) + 3) & ~3;  // equal ranges, whole float4s
  auto kernel = scatter_rows_lds_kernel<DEDUP>;
  const size_t lds = (size_t)(range + 4) * sizeof(float);  // + the row's offset inside its 16-byte line
  if (int e = set_lds(kernel, lds, op)) return e;
  hipLaunchKernelGGL(kernel, dim3(c, ranges, b), dim3(kScThreads), lds, stream, grad_out, idx, grad_points, c, n, ne, range);
  return check_launch(op);
}

extern "C" int vdetr_gather_points_grad_set_f32(const float* grad_out, const int32_t* idx, float* grad_points,
                                                int b, int c, int n, int m, vdetr_stream_t stream) {
  if (int e = check_bcnm("gather_points_grad_set", b, c, n, m)) return e;
  if (b == 0 || c == 0 || n == 0) return VDETR_OK;
  VDETR_REQUIRE(grad_points && (m == 0 || (grad_out && idx)), "gather_points_grad_set: null pointer");
  if (m > 0 && scatter_lds_pays(b, c))
    return scatter_rows_lds<false>("gather_points_grad_set", grad_out, idx, grad_points, b, c, n, m, (hipStream_t)stream);
  if (hipMemsetAsync(grad_points, 0, (size_t)b * c * n * sizeof(float), (hipStream_t)stream) != hipSuccess) {
    set_error("gather_points_grad_set: cannot clear grad_points");
    return VDETR_ERR_LAUNCH;
  }
  return vdetr_gather_points_grad_f32(grad_out, idx, grad_points, b, c, n, m, stream);
}

extern "C" int vdetr_group_points_grad_set_f32(const float* grad_out, const int32_t* idx, float* grad_points,
                                               int b, int c, int n, int npoints, int nsample, vdetr_stream_t stream) {
  if (int e = check_bcnm("group_points_grad_set", b, c, n, npoints)) return e;
  VDETR_REQUIRE(nsample >= 0, "group_points_grad_set: negative nsample");
  const long ne = (long)npoints * nsample;
  if (b == 0 || c == 0 || n == 0) return VDETR_OK;
  VDETR_REQUIRE(grad_points && (ne == 0 || (grad_out && idx)), "group_points_grad_set: null pointer");
  if (ne > 0 && scatter_lds_pays(b, c))
    return scatter_rows_lds<true>("group_points_grad_set", grad_out, idx, grad_points, b, c, n, ne, (hipStream_t)stream);
  if (hipMemsetAsync(grad_points, 0, (size_t)b * c * n * sizeof(float), (hipStream_t)stream) != hipSuccess) {
    set_error("group_points_grad_set: cannot clear grad_points");
    return VDETR_ERR_LAUNCH;
  }
  return vdetr_group_points_grad_f32(grad_out, idx, grad_points, b, c, n, npoints, nsample, stream);
}

extern "C" int vdetr_three_nn_f32(const float* unknown, const float* known, float* dist2, int32_t* idx,
                                  int b, int n, int m, vdetr_stream_t stream) {
  if (int e = check_bcnm("three_nn", b, 0, n, m)) return e;
  if (b == 0 || n == 0) return VDETR_OK;
  VDETR_REQUIRE(unknown && dist2 && idx && (known || m == 0), "three_nn: null pointer");
  dim3 grid(ceil_div(n, kThreads / kWave), 1, b);
  hipLaunchKernelGGL(three_nn_kernel, grid, dim3(kThreads), 0, (hipStream_t)stream, unknown, known, dist2,
                     idx, n, m);
  return check_launch("three_nn");
}

extern "C" int vdetr_three_interpolate_f32(const float* points, const int32_t* idx, const float* weight,
                                           float* out, int b, int c, int m, int n, vdetr_stream_t stream) {
  if (int e = check_bcnm("three_interpolate", b, c, n, m)) return e;
  if (b == 0 || c == 0 || n == 0) return VDETR_OK;
  VDETR_REQUIRE(points && idx && weight && out, "three_interpolate: null pointer");
  const size_t lds = (size_t)kTiStrip * m * sizeof(float);
  if (lds <= 64 * 1024 && n >= 4 * kThreads) {  // the rows of a strip fit in LDS and there are points enough to pay for staging them
    const int jblocks = ceil_div(n, kThreads) < 8 ? ceil_div(n, kThreads) : 8;  // (each workgroup stages 64 KB: few, fat workgroups per strip)
    dim3 grid(jblocks, ceil_div(c, kTiStrip), b);
    if (int e = set_lds(three_interpolate_lds_kernel, lds, "three_interpolate")) return e;
    hipLaunchKernelGGL(three_interpolate_lds_kernel, grid, dim3(kThreads), lds, (hipStream_t)stream, points, idx, weight, out, c, m, n);
    return check_launch("three_interpolate");
  }
  dim3 grid(ceil_div(n, kThreads), ceil_div(c, kChanStrip), b);
  hipLaunchKernelGGL(three_interpolate_kernel, grid, dim3(kThreads), 0, (hipStream_t)stream, points, idx,
                     weight, out, c, m, n);
  return check_launch("three_interpolate");
}

extern "C" int vdetr_three_interpolate_grad_f32(const float* grad_out, const int32_t* idx,
                                                const float* weight, float* grad_points, int b, int c,
                                                int n, int m, vdetr_stream_t stream) {
  if (int e = check_bcnm("three_interpolate_grad", b, c, n, m)) return e;
  if (b == 0 || c == 0 || n == 0) return VDETR_OK;
  VDETR_REQUIRE(grad_out && idx && weight && grad_points, "three_interpolate_grad: null pointer");
  dim3 grid(ceil_div(n, kThreads), ceil_div(c, kChanStrip), b);
  hipLaunchKernelGGL(three_interpolate_grad_kernel, grid, dim3(kThreads), 0, (hipStream_t)stream, grad_out,
                     idx, weight, grad_points, c, n, m);
  return check_launch("three_interpolate_grad");
}

extern "C" int vdetr_ball_query_f32(const float* new_xyz, const float* xyz, int32_t* idx, int b, int n,
                                    int m, float radius, int nsample, vdetr_stream_t stream) {
  if (int e = check_bcnm("ball_query", b, 0, n, m)) return e;
  VDETR_REQUIRE(nsample >= 0, "ball_query: negative nsample");
  if (b == 0 || m == 0 || nsample == 0 || n == 0) return VDETR_OK;
  VDETR_REQUIRE(new_xyz && xyz && idx, "ball_query: null pointer");
  dim3 grid(ceil_div(m, kBqQueries), 1, b);
  // radius2 is formed in float like the reference (ball_query_gpu.cu:25)
  const float radius2 = radius * radius;
  hipLaunchKernelGGL(ball_query_kernel, grid, dim3(kBqThreads), 0, (hipStream_t)stream, new_xyz, xyz, idx, n,
                     m, radius2, nsample);
  return check_launch("ball_query");
}

extern "C" int vdetr_gather_rows_f32(const float* const* rows, const int32_t* idx, float* out, int b, int c, int m,
                                     vdetr_stream_t stream) {
  VDETR_REQUIRE(b >= 0 && c >= 0 && m >= 0, "gather_rows: negative dimension");
  if (b == 0 || c == 0 || m == 0) return VDETR_OK;
  VDETR_REQUIRE(b <= kRowScenes, "gather_rows: %d scenes > %d per launch", b, kRowScenes);
  VDETR_REQUIRE(rows && idx && out, "gather_rows: null pointer");
  RowTables T{};
  bool vec = c % 4 == 0 && ((uintptr_t)out & 15) == 0;
  for (int i = 0; i < b; ++i) {
    VDETR_REQUIRE(rows[i] != nullptr, "gather_rows: scene %d: null pointer", i);
    T.src[i] = rows[i];
    vec = vec && ((uintptr_t)rows[i] & 15) == 0;
  }
  if (vec) {
    dim3 grid(ceil_div((long)m * (c / 4), kThreads), b);
    hipLaunchKernelGGL(gather_rows_kernel<4>, grid, dim3(kThreads), 0, (hipStream_t)stream, T, idx, out, c, m);
  } else {
    dim3 grid(ceil_div((long)m * c, kThreads), b);
    hipLaunchKernelGGL(gather_rows_kernel<1>, grid, dim3(kThreads), 0, (hipStream_t)stream, T, idx, out, c, m);
  }
  return check_launch("gather_rows");
}

extern "C" int vdetr_gather_rows_grad_f32(const float* grad_out, const int32_t* idx, float* const* grad_rows, int b, int c,
                                          int m, vdetr_stream_t stream) {
  VDETR_REQUIRE(b >= 0 && c >= 0 && m >= 0, "gather_rows_grad: negative dimension");
  if (b == 0 || c == 0 || m == 0) return VDETR_OK;
  VDETR_REQUIRE(b <= kRowScenes, "gather_rows_grad: %d scenes > %d per launch", b, kRowScenes);
  VDETR_REQUIRE(grad_out && idx && grad_rows, "gather_rows_grad: null pointer");
  RowGradTables T{};
  for (int i = 0; i < b; ++i) {
    VDETR_REQUIRE(grad_rows[i] != nullptr, "gather_rows_grad: scene %d: null pointer", i);
    T.dst[i] = grad_rows[i];
  }
  if (c % 4 == 0) {
    dim3 grid(ceil_div((long)m * (c / 4), kThreads), b);
    hipLaunchKernelGGL(gather_rows_grad_kernel<4>, grid, dim3(kThreads), 0, (hipStream_t)stream, T, idx, grad_out, c, m);
  } else {
    dim3 grid(ceil_div((long)m * c, kThreads), b);
    hipLaunchKernelGGL(gather_rows_grad_kernel<1>, grid, dim3(kThreads), 0, (hipStream_t)stream, T, idx, grad_out, c, m);
  }
  return check_launch("gather_rows_grad");
}
