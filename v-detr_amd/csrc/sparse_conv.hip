// sparse_conv.hip — index kernels of the sparse-convolution backbone, gfx950.
//
// Reference: the backbone is MinkowskiEngine's generalized sparse convolution (models/mink_resnet.py:38-84,
// models/model_vdetr.py:141-176,248-280); MinkowskiEngine itself is not under /root/reference (un-vendored, no pinned
// commit: README.md:47-53), so what is restated here is its PUBLISHED operator (Choy et al., "4D Spatio-Temporal ConvNets",
// CVPR 2019, eq. 3):   out[u] = sum_{i in N(u)} in[u + i] W_i   over the occupied sites u + i only.
// Parity is unpinned against the MinkowskiEngine binary; the oracle (oracle/sparse_oracle.py) pins these kernels against
// torch's dense conv3d / conv_transpose3d on densified grids.
//
// Layout: a sparse tensor = sorted int64 voxel KEYS [N] + a point-major feature table [N, C] (exactly what the hot path's
// FPS / row gathers consume downstream).  A key packs (batch, x, y, z) as 16-bit biased fields, so ascending key order =
// lexicographic (batch, x, y, z) order and a neighbour lookup is a binary search in an L2-resident array (40 k keys =
// 320 KB) instead of a hash table: no atomics, no collisions, and the resulting maps are deterministic.
//   vdetr_sp_kernel_map_i32   nbr[k][u]  = row of the input site at out_key[u] + offset[k], or -1         (geometry only:
//   vdetr_sp_inverse_map_i32  inv[k][i]  = the output u that reads input i through offset k, or -1         once per scene)
//   vdetr_sp_gather_cols_f32  col[u][k][:] = in[nbr[k][u]][:] or 0     (the im2col operand of ONE library GEMM per layer)
//   vdetr_sp_gather_sum_f32   din[i][:] = sum_k dcol[inv[k][i]][k][:]  (its adjoint as a GATHER: no float atomics; the source
//                             may also be offset-major [K][M][C]: the compacted per-offset row lists of sparse_ops.ConvPlan)
// The two feature kernels are pure HBM streams of C-float rows (256 B - 2 KB each): float4 per lane, rows x offsets over
// the whole chip.
#include <stdlib.h>
#include <atomic>
#include <mutex>
#include "common.h"

namespace vdetr {

typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef VDETR_SP_PROBE
#define VDETR_SP_PROBE 0  // measurements only.  1: operands loaded once (MFMA loop alone), 2: loads alone (folded with adds)
#endif

constexpr long long kKeyBias = 32768;

__device__ __forceinline__ long long sp_key_add(long long key, int dx, int dy, int dz, bool& ok) {
  const int b = (int)((unsigned long long)key >> 48);
  const int x = (int)((key >> 32) & 0xFFFF) + dx, y = (int)((key >> 16) & 0xFFFF) + dy, z = (int)(key & 0xFFFF) + dz;
  ok = ((unsigned)x | (unsigned)y | (unsigned)z) < 65536u;  // a neighbour outside the 16-bit box cannot be occupied
  return ((long long)b << 48) | ((long long)x << 32) | ((long long)y << 16) | (long long)z;
}

// lower bound in a sorted array; -1 if `key` is absent
__device__ __forceinline__ int sp_find(const long long* __restrict__ keys, int n, long long key) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (keys[mid] < key) lo = mid + 1;
    else hi = mid;
  }
  return (lo < n && keys[lo] == key) ? lo : -1;
}

__global__ __launch_bounds__(256) void sp_kernel_map_kernel(const long long* __restrict__ in_keys, int nin,
                                                           const long long* __restrict__ out_keys, int nout,
                                                           const int* __restrict__ offsets, int K, int* __restrict__ nbr) {
  const int u = blockIdx.x * 256 + threadIdx.x, k = blockIdx.y;
  if (u >= nout) return;
  bool ok;
  const long long q = sp_key_add(out_keys[u], offsets[k * 3], offsets[k * 3 + 1], offsets[k * 3 + 2], ok);
  nbr[(size_t)k * nout + u] = ok ? sp_find(in_keys, nin, q) : -1;
}

__global__ __launch_bounds__(256) void sp_inverse_map_kernel(const int* __restrict__ nbr, int K, int nout, int nin,
                                                            int* __restrict__ inv) {
  const int u = blockIdx.x * 256 + threadIdx.x, k = blockIdx.y;
  if (u >= nout) return;
  const int i = nbr[(size_t)k * nout + u];
  if (i >= 0) inv[(size_t)k * nin + i] = u;  // (i, k) has at most one reader: u = i - offset[k] on the output lattice
}

// ---- pair lists of a kernel map, built on the device (no host round trip until the counts are needed) ------------------
// pairs sorted by (offset k, output row u): pass 1 counts the occupied neighbours per (k, block of 256 rows), pass 2 scans the
// block counts, pass 3 writes pin / pout / slot / islot at block offset + rank in the block.  Deterministic (no atomics).
__global__ __launch_bounds__(256) void sp_plan_count_kernel(const int* __restrict__ nbr, int nout, int nb, int* __restrict__ bc) {
  __shared__ int w[4];
  const int b = blockIdx.x, k = blockIdx.y, u = b * 256 + threadIdx.x;
  const bool v = u < nout && nbr[(size_t)k * nout + u] >= 0;
  const unsigned long long m = __ballot(v);
  if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = __popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) bc[k * nb + b] = w[0] + w[1] + w[2] + w[3];
}

__global__ __launch_bounds__(1024) void sp_plan_scan_kernel(int* __restrict__ bc, int n, int nb, int K, int* __restrict__ counts) {
  __shared__ int part[1024];
  const int t = threadIdx.x, per = (n + 1023) / 1024;
  const int lo = min(t * per, n), hi = min(lo + per, n);
  int s = 0;
  for (int i = lo; i < hi; ++i) s += bc[i];
  part[t] = s;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    const int v = t >= d ? part[t - d] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  int run = part[t] - s;
  for (int i = lo; i < hi; ++i) {
    const int c = bc[i];
    bc[i] = run;  // exclusive offset of block i in (k, b) order
    run += c;
  }
  __syncthreads();
  const int total = part[1023];
  if (t < K) counts[t] = (t + 1 < K ? bc[(t + 1) * nb] : total) - bc[t * nb];
  if (t == 0) counts[K] = total;
}

__global__ __launch_bounds__(256) void sp_plan_fill_kernel(const int* __restrict__ nbr, int nout, int nin, int nb,
                                                          const int* __restrict__ boff, int* __restrict__ pin,
                                                          int* __restrict__ pout, int* __restrict__ slot, int* __restrict__ islot) {
  __shared__ int w[4];
  const int b = blockIdx.x, k = blockIdx.y, u = b * 256 + threadIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int i = u < nout ? nbr[(size_t)k * nout + u] : -1;
  const bool v = i >= 0;
  const unsigned long long m = __ballot(v);
  if (lane == 0) w[wv] = __popcll(m);
  __syncthreads();
  int p = boff[k * nb + b] + __popcll(m & ((1ull << lane) - 1ull));
  for (int j = 0; j < wv; ++j) p += w[j];
  if (v) {
    pin[p] = i;
    pout[p] = u;
    islot[(size_t)k * nin + i] = p;
  }
  if (u < nout) slot[(size_t)k * nout + u] = v ? p : -1;
}

// col[u][k][c4] <- in[nbr[k][u]][c4]: one float4 per thread, threads of a workgroup walk consecutive (k, c4) of a row u
template <bool SUM>
__global__ __launch_bounds__(256) void sp_gather_kernel(const float* __restrict__ src, const int* __restrict__ map, int K,
                                                       int nrows, int nmap, int C4, float* __restrict__ dst) {
  // SUM = false: dst [nrows][K][C] <- src [nmap_rows][C] through map [K][nrows]
  // SUM = true : dst [nrows][C]    <- sum_k src [.][K][C] rows map[k][row], slice k
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (!SUM) {
    const long long total = (long long)nrows * K * C4;
    if (t >= total) return;
    const int c4 = (int)(t % C4);
    const int k = (int)((t / C4) % K);
    const int u = (int)(t / ((long long)C4 * K));
    const int i = map[(size_t)k * nrows + u];
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (i >= 0) v = reinterpret_cast<const f32x4*>(src)[(size_t)i * C4 + c4];
    reinterpret_cast<f32x4*>(dst)[t] = v;
  } else {
    const long long total = (long long)nrows * C4;
    if (t >= total) return;
    const int c4 = (int)(t % C4);
    const int i = (int)(t / C4);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    // nmap = 0: src is [rows][K][C] (the im2col layout); nmap = M > 0: src is offset-major [K][M][C]; nmap < 0: the map
    // holds absolute rows of a flat [P][C] source (the pair lists).
    // Eight offsets at a time: their map entries are loaded together, then their rows — an index -> row chain per offset
    // (27 dependent round trips per thread) left the kernel latency-bound at 3.4 TB/s.  The additions keep the offset order:
    // deterministic sums.
#ifndef VDETR_SP_GATHER_G
#define VDETR_SP_GATHER_G 8
#endif
    constexpr int G = VDETR_SP_GATHER_G;
    for (int k0 = 0; k0 < K; k0 += G) {
      int u[G];
#pragma unroll
      for (int j = 0; j < G; ++j) u[j] = k0 + j < K ? map[(size_t)(k0 + j) * nrows + i] : -1;
      f32x4 v[G];
#pragma unroll
      for (int j = 0; j < G; ++j) {
        const int k = k0 + j;
        v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (u[j] >= 0)
          v[j] = reinterpret_cast<const f32x4*>(src)[(nmap > 0 ? (size_t)k * nmap + u[j] : nmap < 0 ? (size_t)u[j] : (size_t)u[j] * K + k) * C4 + c4];
      }
#pragma unroll
      for (int j = 0; j < G; ++j)
        if (u[j] >= 0) acc += v[j];
    }
    reinterpret_cast<f32x4*>(dst)[t] = acc;
  }
}

}  // namespace vdetr

using namespace vdetr;

extern "C" int vdetr_sp_kernel_map_i32(const int64_t* in_keys, int nin, const int64_t* out_keys, int nout,
                                       const int32_t* offsets, int K, int32_t* nbr, vdetr_stream_t stream) {
  VDETR_REQUIRE(nin >= 0 && nout >= 0 && K > 0, "sp_kernel_map: negative size (nin=%d nout=%d K=%d)", nin, nout, K);
  if (nout == 0) return VDETR_OK;
  VDETR_REQUIRE(out_keys && offsets && nbr && (in_keys || nin == 0), "sp_kernel_map: null pointer");
  VDETR_REQUIRE(K <= 65535, "sp_kernel_map: K=%d > 65535", K);
  hipLaunchKernelGGL(sp_kernel_map_kernel, dim3(ceil_div(nout, 256), K), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const long long*>(in_keys), nin, reinterpret_cast<const long long*>(out_keys), nout,
                     offsets, K, nbr);
  return check_launch("sp_kernel_map");
}

extern "C" int vdetr_sp_inverse_map_i32(const int32_t* nbr, int K, int nout, int nin, int32_t* inv, vdetr_stream_t stream) {
  VDETR_REQUIRE(nin >= 0 && nout >= 0 && K > 0, "sp_inverse_map: negative size");
  if (nout == 0 || nin == 0) return VDETR_OK;
  VDETR_REQUIRE(nbr && inv, "sp_inverse_map: null pointer");
  hipLaunchKernelGGL(sp_inverse_map_kernel, dim3(ceil_div(nout, 256), K), dim3(256), 0, (hipStream_t)stream, nbr, K, nout,
                     nin, inv);
  return check_launch("sp_inverse_map");
}

extern "C" int vdetr_sp_gather_cols_f32(const float* in, const int32_t* nbr, int K, int nout, int C, float* col,
                                        vdetr_stream_t stream) {
  VDETR_REQUIRE(K > 0 && nout >= 0 && C > 0, "sp_gather_cols: bad size (K=%d nout=%d C=%d)", K, nout, C);
  VDETR_REQUIRE(C % 4 == 0, "sp_gather_cols: C=%d must be a multiple of 4 (float4 rows)", C);
  if (nout == 0) return VDETR_OK;
  VDETR_REQUIRE(in && nbr && col, "sp_gather_cols: null pointer");
  const long long total = (long long)nout * K * (C / 4);
  hipLaunchKernelGGL((sp_gather_kernel<false>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in,
                     nbr, K, nout, 0, C / 4, col);
  return check_launch("sp_gather_cols");
}

extern "C" int vdetr_sp_gather_sum_f32(const float* dcol, const int32_t* inv, int K, int nin, int C, int offset_major_rows,
                                       float* din, vdetr_stream_t stream) {
  VDETR_REQUIRE(K > 0 && nin >= 0 && C > 0, "sp_gather_sum: bad size (K=%d nin=%d C=%d)", K, nin, C);
  VDETR_REQUIRE(C % 4 == 0, "sp_gather_sum: C=%d must be a multiple of 4 (float4 rows)", C);
  if (nin == 0) return VDETR_OK;
  VDETR_REQUIRE(dcol && inv && din, "sp_gather_sum: null pointer");
  const long long total = (long long)nin * (C / 4);
  hipLaunchKernelGGL((sp_gather_kernel<true>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dcol,
                     inv, K, nin, offset_major_rows, C / 4, din);
  return check_launch("sp_gather_sum");
}

// ====================================================================================================================
// Fused pair-list convolution (fp32 matrix cores).  The geometry of a layer is a list of (input row, output row) PAIRS
// sorted by kernel offset (segment k = the pairs of offset k); the three products of a layer run over that list directly:
//   vdetr_sp_pairs_gemm_f32    Y[p][:] = X[arow[p]][:] * W[k(p)]        (forward: X = features, W[k] as stored [Cin, Cout];
//                                                                      input gradient: X = dout, W[k] transposed)
//   vdetr_sp_pairs_wgrad_f32   dW[k]   = sum_{p in segment k} X[pin[p]]^T dY[pout[p]]
// followed by the fixed-order gather-sum over the (<= K) pairs of a row (vdetr_sp_gather_sum_f32, flat mode).  Nothing is
// padded to a common length and no gathered operand is ever written to memory: rows are gathered straight into the MFMA
// A operand.  Tile = 64 pairs x 64 channels per 256-thread workgroup, v_mfma_f32_16x16x4_f32 (exact fp32), contraction
// index permuted (k = 4 g + s) so that a lane's four A values are one float4.
// ====================================================================================================================
namespace vdetr {

// Workgroup tile = (WR x RT x 16) pairs x (WC x CT x 16) channels: the 4 waves form a WR x WC grid, a wave owns RT x CT
// MFMA tiles.  <2,2,4,4>: 128 x 128 (16 accumulators per wave; per contraction step of 16 a wave loads 4 float4 of A and
// 4 float4 / 16 dwords of B for 64 MFMAs), <4,1,2,4>: 128 x 64 for layers with <= 64 output channels.
// tiles [ntiles][3] = (offset k, first pair, pair count <= 128)
constexpr int kSpTileM = 128;

#ifndef VDETR_SP_WAVES
#define VDETR_SP_WAVES 2  // waves per SIMD of the matrix-core kernels (more than 2 halves the MFMA rate)
#endif
template <bool TRANS, int WR, int WC, int RT, int CT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, VDETR_SP_WAVES))) void sp_pairs_gemm_kernel(const float* __restrict__ X, const int* __restrict__ arow,
                                                           const float* __restrict__ W, const int* __restrict__ tiles,
                                                           int CA, int CB, int wk_stride, float* __restrict__ Y) {
  // CA = contraction width (row length of X), CB = output width.  W[k] is [Cin][Cout] row-major; !TRANS: CA = Cin, CB = Cout,
  // B[c][n] = W[k][c][n];  TRANS: CA = Cout, CB = Cin, B[c][n] = W[k][n][c].
  static_assert(WR * WC == 4 && WR * RT * 16 == kSpTileM, "tile shape");
  constexpr int TN = WC * CT * 16;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, g = lane >> 4, c = lane & 15;
  const int wr = w / WC, wc = w % WC;
  const int k = tiles[blockIdx.x * 3], p0 = tiles[blockIdx.x * 3 + 1], cnt = tiles[blockIdx.x * 3 + 2];
  const int n0 = blockIdx.y * TN + wc * CT * 16;
  const int r0 = wr * RT * 16;
  // No bounds tests inside the loop: rows past the tile's count read the tile's first pair and columns past CB read column
  // CB - 1 (valid memory); what they produce is never stored.
  const float* xrow[RT];
#pragma unroll
  for (int i = 0; i < RT; ++i) {
    const int rl = r0 + 16 * i + c;
    xrow[i] = X + (size_t)arow[p0 + (rl < cnt ? rl : 0)] * CA + 4 * g;
  }
  const float* wk = W + (size_t)k * wk_stride;
  const float* wcol[CT];
#pragma unroll
  for (int t = 0; t < CT; ++t) {
    const int n = min(n0 + 16 * t + c, CB - 1);
    wcol[t] = TRANS ? wk + (size_t)n * CA + 4 * g : wk + (size_t)(4 * g) * CB + n;
  }
  f32x4 acc[RT][CT];
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int t = 0; t < CT; ++t) acc[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int kc = 0; kc < CA; kc += 16) {
    f32x4 a[RT];
    float b[CT][4];
    const int kl = (VDETR_SP_PROBE == 1 || VDETR_SP_PROBE == 3) ? 0 : kc;
    if (VDETR_SP_PROBE == 3) asm volatile("" ::: "memory");  // 3: the loads stay in the loop but always hit the same lines
#pragma unroll
    for (int i = 0; i < RT; ++i) a[i] = *reinterpret_cast<const f32x4*>(xrow[i] + kl);
#pragma unroll
    for (int t = 0; t < CT; ++t) {
      if (TRANS) {  // W[k][n][kc + 4g .. +3]: one float4
        const f32x4 v = *reinterpret_cast<const f32x4*>(wcol[t] + kl);
        b[t][0] = v[0]; b[t][1] = v[1]; b[t][2] = v[2]; b[t][3] = v[3];
      } else {      // W[k][kc + 4g + s][n]
#pragma unroll
        for (int s = 0; s < 4; ++s) b[t][s] = wcol[t][(size_t)(kl + s) * CB];
      }
    }
#if VDETR_SP_PROBE == 2
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int t = 0; t < CT; ++t) acc[i][t][s] += a[i][s] + b[t][s];
#else
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int t = 0; t < CT; ++t) acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][s], b[t][s], acc[i][t], 0, 0, 0);
#endif
  }
  // accumulator: lane (g, c) holds rows 4 g + r, column c of every 16 x 16 tile
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = r0 + 16 * i + 4 * g + r;
      if (row >= cnt) continue;
      float* y = Y + (size_t)(p0 + row) * CB + n0 + c;
#pragma unroll
      for (int t = 0; t < CT; ++t)
        if (n0 + 16 * t + c < CB) y[16 * t] = acc[i][t][r];
    }
}

// Persistent form of the 128 x 128 kernel (layers with > 64 output channels and a contraction width that is a multiple of 32):
// 2 workgroups per CU walk the (pair tile, channel tile) work items with a stride of the grid, and everything a work item has
// to wait for is fetched under the previous one's MFMAs: the operands of K-step s+1 are loaded into a second register set
// before the 64 MFMAs of step s are issued (the plain kernel issues its loads, waits, then multiplies), the next item's tile
// descriptor and row indices are fetched when the current item starts, and its first operands under the current item's
// last K-step.  The 14 us of MFMA of a 256-channel item were sitting behind an index -> row -> operand chain of three dependent
// memory round trips and 64 KB of stores per workgroup; a persistent workgroup pays that chain once.
// B operand addresses are a uniform row pointer (scalar registers) + a per-lane column offset: no 64-bit vector address
// arithmetic in the loop (the plain kernel: 16 v_lshl_add_u64 per K-step).
// fp32 operands on the bf16 matrix unit: x = hi + lo (both round-to-nearest bf16), products as the three leading cross terms
// hi hi + hi lo + lo hi: relative error of a product <= 2^-16, fp32 accumulation.  v_mfma_f32_16x16x4_f32 runs at 1/16 of
// the bf16 rate (MI355X_MICROARCH.md): the fp32 form of these kernels sat at 45-55 % of THAT peak.
typedef __bf16 sp_bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void sp_split8(const f32x4& x0, const f32x4& x1, sp_bf16x8& hi, sp_bf16x8& lo) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const __bf16 h0 = (__bf16)x0[e], h1 = (__bf16)x1[e];
    hi[e] = h0; hi[4 + e] = h1;
    lo[e] = (__bf16)(x0[e] - (float)h0); lo[4 + e] = (__bf16)(x1[e] - (float)h1);
  }
}
__device__ __forceinline__ f32x4 sp_mfma3(const sp_bf16x8& ah, const sp_bf16x8& al, const sp_bf16x8& bh, const sp_bf16x8& bl, f32x4 c) {
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, c, 0, 0, 0);
}

template <bool TRANS, int KSUB, bool SPLIT = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, (KSUB == 1 || SPLIT) ? 2 : 1))) void sp_pairs_gemm_persistent_kernel(
    const float* __restrict__ X, const int* __restrict__ arow, const float* __restrict__ W, const int* __restrict__ tiles,
    int ntiles, int CA, int CB, int wk_stride, float* __restrict__ Y, int* __restrict__ ticket) {
  constexpr int RT = 4, CT = 4;
  static_assert(!SPLIT || KSUB == 2, "the split-bf16 form contracts 32 channels per step");
  // ticket != NULL (default; VDETR_SP_TICKET=0 for the static stride): work items are handed out by a device counter (the first
  // gridDim.x statically), so that a CU busy with another stream's long kernel — the next scene's 9 ms sampling — takes no
  // items instead of making its workgroup start a round late: worth 1.1-2 ms of 26.7 in the training step.  (Tried: every
  // WAVE drawing 64 x 64 quadrants on its own, no LDS / barrier — the four quadrants of an item then run on four CUs and their
  // shared rows / columns miss the L1: 153 instead of 143 us at 256 channels.)
  __shared__ int next_item[2];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, g = lane >> 4, c = lane & 15;
  const int wr = w >> 1, wc = w & 1;
  const int r0 = wr * 64;
  const int NT = (CB + 127) >> 7, nwork = ntiles * NT, NS = CA / (16 * KSUB);  // K-steps of 16 * KSUB, taken in pairs
  int item = blockIdx.x;
  if (item >= nwork) return;

  struct Work {
    int k, p0, cnt, n0;
  };
  auto describe = [&](int it) {
    const int t = it / NT, nt = it - t * NT;
    Work d;
    d.k = tiles[t * 3];
    d.p0 = tiles[t * 3 + 1];
    d.cnt = tiles[t * 3 + 2];
    d.n0 = nt * 128 + wc * 64;
    return d;
  };
  auto rows_of = [&](const Work& d, int (&ri)[RT]) {
#pragma unroll
    for (int i = 0; i < RT; ++i) {
      const int rl = r0 + 16 * i + c;
      ri[i] = arow[d.p0 + (rl < d.cnt ? rl : 0)];  // rows past the tile's count read its first pair (never stored)
    }
  };
  // operand addresses of a work item as buffer loads: a descriptor in scalar registers (X once, W[k] per item), a per-lane byte
  // offset that does not change inside an item and a scalar offset per K-step — no 64-bit vector address arithmetic in the loop
  // (the plain kernel: 16 v_lshl_add_u64 per K-step) and no pointer pairs held in vector registers
  using rsrc_t = __amdgpu_buffer_rsrc_t;
  // (num_records = 2^32 - 1: the per-lane byte offsets are 32-bit, so the feature table must stay below 4 GB — checked by the
  // caller, v-detr_amd/sparse_ops.py:pairs_gemm)
  auto make_rsrc = [](const void* p) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, -1, 0x00020000); };
  const rsrc_t rX = make_rsrc(X);
  auto a_offsets = [&](const int (&ri)[RT], unsigned (&xo)[RT]) {
#pragma unroll
    for (int i = 0; i < RT; ++i) xo[i] = ((unsigned)ri[i] * (unsigned)CA + 4u * g) * 4u;
  };
  unsigned bo[CT];  // per-lane part of the B address (bytes)
  auto b_offsets = [&](const Work& d) {
#pragma unroll
    for (int t = 0; t < CT; ++t) {
      const int n = min(d.n0 + 16 * t + c, CB - 1);  // columns past CB read column CB - 1 (never stored)
      bo[t] = (TRANS ? (unsigned)n * (unsigned)CA + 4u * g : (unsigned)(4 * g) * (unsigned)CB + (unsigned)n) * 4u;
    }
  };
  auto load_ops = [&](const unsigned (&xo)[RT], rsrc_t rW, int kc0, f32x4 (&a)[KSUB][RT], f32x4 (&b)[KSUB][CT]) {
#if defined(VDETR_SP_PROBE) && (VDETR_SP_PROBE == 1 || VDETR_SP_PROBE == 3)
    kc0 = 0;  // measurements only: every K-step reads the same (cache-resident) operands
#endif
#pragma unroll
    for (int u = 0; u < KSUB; ++u) {
      const int kc = kc0 + 16 * u;
#pragma unroll
      for (int i = 0; i < RT; ++i)
        a[u][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rX, (int)xo[i], kc * 4, 0));
#pragma unroll
      for (int t = 0; t < CT; ++t) {
        if (TRANS) {
          b[u][t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rW, (int)bo[t], kc * 4, 0));
        } else {
#pragma unroll
          for (int s2 = 0; s2 < 4; ++s2)
            b[u][t][s2] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rW, (int)bo[t], (kc + s2) * CB * 4, 0));
        }
      }
    }
  };

  Work cur = describe(item);
  int ri[RT];
  rows_of(cur, ri);
  unsigned xo[RT];
  a_offsets(ri, xo);
  b_offsets(cur);
  rsrc_t rW = make_rsrc(W + (size_t)cur.k * wk_stride);
  f32x4 a0[KSUB][RT], b0[KSUB][CT], a1[KSUB][RT], b1[KSUB][CT];
  load_ops(xo, rW, 0, a0, b0);

  for (int round = 0;; ++round) {
    int nitem = item + (int)gridDim.x;
    if (ticket) {
      if (threadIdx.x == 0) next_item[round & 1] = (int)gridDim.x + atomicAdd(ticket, 1);
      __syncthreads();
      nitem = next_item[round & 1];
    }
    const bool more = nitem < nwork;
    Work nxt = cur;
    int rin[RT];
    if (more) {  // descriptor and row indices of the next item: in flight during this item's K-loop
      nxt = describe(nitem);
      rows_of(nxt, rin);
    }
    f32x4 acc[RT][CT];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
      for (int t = 0; t < CT; ++t) acc[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto mfma_step = [&](const f32x4 (&a)[KSUB][RT], const f32x4 (&b)[KSUB][CT]) {
      if (SPLIT) {  // (KSUB == 2) the lane's 2 x 4 contraction values of a K-step of 32 are one bf16 operand: 3 instead of 8 MFMAs
        sp_bf16x8 ah[RT], al[RT], bh[CT], bl[CT];
#pragma unroll
        for (int i = 0; i < RT; ++i) sp_split8(a[0][i], a[KSUB - 1][i], ah[i], al[i]);
#pragma unroll
        for (int t = 0; t < CT; ++t) sp_split8(b[0][t], b[KSUB - 1][t], bh[t], bl[t]);
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
          for (int t = 0; t < CT; ++t) acc[i][t] = sp_mfma3(ah[i], al[i], bh[t], bl[t], acc[i][t]);
        return;
      }
#pragma unroll
      for (int u = 0; u < KSUB; ++u)
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
          for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int t = 0; t < CT; ++t)
              acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][i][s2], b[u][t][s2], acc[i][t], 0, 0, 0);
    };
    // K-steps in pairs (NS is even): set 0 -> set 1 -> set 0; the last step of the item fetches the next item's first operands
    for (int s = 0; s < NS; s += 2) {
      load_ops(xo, rW, (s + 1) * 16 * KSUB, a1, b1);
      mfma_step(a0, b0);
      if (s + 2 < NS) {
        load_ops(xo, rW, (s + 2) * 16 * KSUB, a0, b0);
      } else if (more) {  // the operand addresses switch to the next item (the epilogue below only needs cur.p0 / cnt / n0)
        a_offsets(rin, xo);
        b_offsets(nxt);
        rW = make_rsrc(W + (size_t)nxt.k * wk_stride);
        load_ops(xo, rW, 0, a0, b0);
      }
      mfma_step(a1, b1);
    }
    // accumulator: lane (g, c) holds rows 4 g + r, column c of every 16 x 16 tile
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = r0 + 16 * i + 4 * g + r;
        if (row >= cur.cnt) continue;
        float* y = Y + (size_t)(cur.p0 + row) * CB + cur.n0 + c;
#pragma unroll
        for (int t = 0; t < CT; ++t)
          if (cur.n0 + 16 * t + c < CB) y[16 * t] = acc[i][t][r];
      }
    if (!more) break;
    item = nitem;
    cur = nxt;
  }
  // the last workgroup to leave resets the launch's counter pair, so that the ring slot is zero again when it comes round:
  // no memset in front of every launch (a 5-10 us node of its own on the stream)
  if (ticket) {
    __syncthreads();
    if (threadIdx.x == 0 && atomicAdd(ticket + 1, 1) == (int)gridDim.x - 1) {
      ticket[0] = 0;
      ticket[1] = 0;
    }
  }
}

// dW[k][ci][co] (+ split partials): workgroup = (segment chunk, ci tile, co tile); the 4 waves form a WR x WC grid over the
// (ci, co) tile, a wave owns RT x CT MFMA tiles: <4,1,1,4> = 64 x 64, <2,2,4,4> = 128 x 128 (wide layers: half the
// re-gathering of rows per channel tile, 8 LDS reads per 16 MFMAs instead of 5 per 4).
// The chunk's pairs are walked SB at a time: the 256 threads stage the gathered X rows (TCI channels) and dY rows (TCO
// channels) in LDS (float4 per thread, rows of >= 256 B: coalesced), double-buffered, and the waves read their MFMA operands
// from there (a pair is one k-slot: A[m = ci][k] = X[pair][ci], B[k][n = co] = dY[pair][co]).
template <int WR, int WC, int RT, int CT, int SB, bool SPLIT = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, VDETR_SP_WAVES))) void sp_pairs_wgrad_kernel(const float* __restrict__ X, const float* __restrict__ dY,
                                                            const int* __restrict__ pin, const int* __restrict__ pout,
                                                            const int* __restrict__ chunks, int Cin, int Cout,
                                                            float* __restrict__ part) {
  constexpr int TCI = WR * RT * 16, TCO = WC * CT * 16;
  constexpr int LX = TCI + 16, LY = TCO + 16;  // row strides = 16 mod 32 floats: the 32 lanes of a read hit 32 different banks
  constexpr int XV = SB * TCI / 4 / 256, YV = SB * TCO / 4 / 256;  // float4 per thread and stage
  static_assert(WR * WC == 4 && XV >= 1 && YV >= 1, "tile shape");
  __shared__ __attribute__((aligned(16))) float lx[2][SB][LX], ly[2][SB][LY];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, c = lane & 15;
  const int wr = w / WC, wc = w % WC;
  const int p0 = chunks[blockIdx.x * 4 + 1], cnt = chunks[blockIdx.x * 4 + 2], slot = chunks[blockIdx.x * 4 + 3];
  const int ci0 = blockIdx.y * TCI, co0 = blockIdx.z * TCO;
  f32x4 acc[RT][CT];
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int t = 0; t < CT; ++t) acc[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  // staging roles: float4 e of the X stage = (pair e / (TCI/4), channels 4 (e % (TCI/4)) ..), e = tid + 256 v
  // row indices are fetched TWO stages ahead of their rows (index -> row is a dependent chain of two memory latencies)
  int ri[XV], ro[YV], ri_n[XV], ro_n[YV];
  auto index_load = [&](int j, int (&a)[XV], int (&b)[YV]) {
#pragma unroll
    for (int v = 0; v < XV; ++v) {
      const int p = j + (tid + 256 * v) / (TCI / 4);
      a[v] = p < cnt ? pin[p0 + p] : -1;
    }
#pragma unroll
    for (int v = 0; v < YV; ++v) {
      const int p = j + (tid + 256 * v) / (TCO / 4);
      b[v] = p < cnt ? pout[p0 + p] : -1;
    }
  };
  f32x4 vx[XV], vy[YV];
  auto row_load = [&](const int (&a)[XV], const int (&b)[YV]) {
#pragma unroll
    for (int v = 0; v < XV; ++v) {
      const int col = ci0 + 4 * ((tid + 256 * v) % (TCI / 4));
      vx[v] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (a[v] >= 0 && col < Cin) vx[v] = *reinterpret_cast<const f32x4*>(X + (size_t)a[v] * Cin + col);
    }
#pragma unroll
    for (int v = 0; v < YV; ++v) {
      const int col = co0 + 4 * ((tid + 256 * v) % (TCO / 4));
      vy[v] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (b[v] >= 0 && col < Cout) vy[v] = *reinterpret_cast<const f32x4*>(dY + (size_t)b[v] * Cout + col);
    }
  };
  index_load(0, ri, ro);
  index_load(SB, ri_n, ro_n);
  row_load(ri, ro);
  int buf = 0;
  for (int j = 0; j < cnt; j += SB) {
#pragma unroll
    for (int v = 0; v < XV; ++v) {
      const int e = tid + 256 * v;
      *reinterpret_cast<f32x4*>(&lx[buf][e / (TCI / 4)][4 * (e % (TCI / 4))]) = vx[v];
    }
#pragma unroll
    for (int v = 0; v < YV; ++v) {
      const int e = tid + 256 * v;
      *reinterpret_cast<f32x4*>(&ly[buf][e / (TCO / 4)][4 * (e % (TCO / 4))]) = vy[v];
    }
    __syncthreads();
    if (j + SB < cnt) {  // next stage's rows in flight while this one is multiplied; the stage after next: its indices
#pragma unroll
      for (int v = 0; v < XV; ++v) ri[v] = ri_n[v];
#pragma unroll
      for (int v = 0; v < YV; ++v) ro[v] = ro_n[v];
      row_load(ri, ro);
      index_load(j + 2 * SB, ri_n, ro_n);
    }
    if (SPLIT) {  // 32 pairs per matrix instruction: lane (c, g) holds pairs 8 g .. 8 g + 7 of a group for its channel
      static_assert(!SPLIT || SB % 32 == 0, "the split-bf16 form walks 32 pairs per step");
#pragma unroll
      for (int s = 0; s < SB / 32; ++s) {
        sp_bf16x8 ah[RT], al[RT], bh[CT], bl[CT];
#pragma unroll
        for (int i = 0; i < RT; ++i) {
          f32x4 x0, x1;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            x0[e] = lx[buf][32 * s + 8 * g + e][(wr * RT + i) * 16 + c];
            x1[e] = lx[buf][32 * s + 8 * g + 4 + e][(wr * RT + i) * 16 + c];
          }
          sp_split8(x0, x1, ah[i], al[i]);
        }
#pragma unroll
        for (int t = 0; t < CT; ++t) {
          f32x4 y0, y1;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            y0[e] = ly[buf][32 * s + 8 * g + e][(wc * CT + t) * 16 + c];
            y1[e] = ly[buf][32 * s + 8 * g + 4 + e][(wc * CT + t) * 16 + c];
          }
          sp_split8(y0, y1, bh[t], bl[t]);
        }
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
          for (int t = 0; t < CT; ++t) acc[i][t] = sp_mfma3(ah[i], al[i], bh[t], bl[t], acc[i][t]);
      }
    } else
#pragma unroll
    for (int s = 0; s < SB / 4; ++s) {
      float a[RT], b[CT];
#pragma unroll
      for (int i = 0; i < RT; ++i) a[i] = lx[buf][4 * s + g][(wr * RT + i) * 16 + c];
#pragma unroll
      for (int t = 0; t < CT; ++t) b[t] = ly[buf][4 * s + g][(wc * CT + t) * 16 + c];
#pragma unroll
      for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int t = 0; t < CT; ++t) acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[t], acc[i][t], 0, 0, 0);
    }
    buf ^= 1;  // the other buffer was last read before the previous barrier
  }
  float* dst = part + (size_t)slot * Cin * Cout;
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = ci0 + (wr * RT + i) * 16 + 4 * g + r;
      if (row >= Cin) continue;
#pragma unroll
      for (int t = 0; t < CT; ++t) {
        const int co = co0 + (wc * CT + t) * 16 + c;
        if (co < Cout) dst[(size_t)row * Cout + co] = acc[i][t][r];
      }
    }
}

// dW[k] = sum of the partial products of offset k's chunks (partials [seg[k], seg[k+1]) of `part`), fixed order: deterministic.
__global__ __launch_bounds__(256) void sp_wgrad_reduce_kernel(const f32x4* __restrict__ part, const int* __restrict__ seg, long elems4,
                                                             f32x4* __restrict__ dw) {
  const int k = blockIdx.y;
  const int c0 = seg[k], c1 = seg[k + 1];
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < elems4; e += (long)gridDim.x * 256) {
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    int c = c0;
    for (; c + 4 <= c1; c += 4) {  // four loads in flight; the additions keep the chunk order
      const f32x4 v0 = part[(size_t)c * elems4 + e], v1 = part[(size_t)(c + 1) * elems4 + e];
      const f32x4 v2 = part[(size_t)(c + 2) * elems4 + e], v3 = part[(size_t)(c + 3) * elems4 + e];
      acc += v0;
      acc += v1;
      acc += v2;
      acc += v3;
    }
    for (; c < c1; ++c) acc += part[(size_t)c * elems4 + e];
    dw[(size_t)k * elems4 + e] = acc;
  }
}

}  // namespace vdetr

extern "C" int vdetr_sp_pairs_gemm_f32(const float* x, const int32_t* arow, const float* w, const int32_t* tiles, int ntiles,
                                       int cin, int cout, int transposed, float* y, vdetr_stream_t stream) {
  VDETR_REQUIRE(ntiles >= 0 && cin > 0 && cout > 0, "sp_pairs_gemm: bad size");
  if (ntiles == 0) return VDETR_OK;
  VDETR_REQUIRE(x && arow && w && tiles && y, "sp_pairs_gemm: null pointer");
  const int CA = transposed ? cout : cin, CB = transposed ? cin : cout;
  VDETR_REQUIRE(CA % 16 == 0, "sp_pairs_gemm: contraction width %d must be a multiple of 16", CA);
  hipStream_t st = (hipStream_t)stream;
#ifdef VDETR_SP_PADTEST
  const int pad = VDETR_AB("VDETR_SP_LDS_PAD", 0);  // unused LDS: caps the workgroups per CU
  if (pad > 0) {
    dim3 grid(ntiles, ceil_div(CB, 128));
    auto kern = transposed ? sp_pairs_gemm_kernel<true, 2, 2, 4, 4> : sp_pairs_gemm_kernel<false, 2, 2, 4, 4>;
    if (set_lds(kern, pad, "sp_pairs_gemm") != VDETR_OK) return VDETR_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, grid, dim3(256), pad, st, x, arow, w, tiles, CA, CB, cin * cout, y);
    return check_launch("sp_pairs_gemm");
  }
#endif
  if (CB <= 64) {  // narrow output: 128 x 64 tiles
    dim3 grid(ntiles, ceil_div(CB, 64));
    if (transposed)
      hipLaunchKernelGGL((sp_pairs_gemm_kernel<true, 4, 1, 2, 4>), grid, dim3(256), 0, st, x, arow, w, tiles, CA, CB, cin * cout, y);
    else
      hipLaunchKernelGGL((sp_pairs_gemm_kernel<false, 4, 1, 2, 4>), grid, dim3(256), 0, st, x, arow, w, tiles, CA, CB, cin * cout, y);
  } else {
    // persistent = 2 (default): one workgroup per CU, K-steps of 32 (the next 32 x (A, B) operands in flight under 128 MFMAs =
    // 1.7 us: covers a row fetched from HBM); 1: two workgroups per CU, K-steps of 16; 0: the plain kernel.  A/B switch.
    const int persistent = VDETR_AB("VDETR_SP_PERSISTENT", 2);
    const int ksub = persistent == 2 && CA % 64 == 0 ? 2 : 1;
    if (persistent && CA % 32 == 0) {
      // per-device state (CU count, work-counter ring), created under a lock on the device's first launch: a process may
      // drive several GPUs from several threads
      struct DevState { int cus = 0; int* ring = nullptr; std::atomic<unsigned> next{0}; };
      static DevState devs[64];
      static std::mutex dev_mu;
      int dev = 0;
      if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
      DevState& D = devs[dev];
      if (!D.cus) {
        std::lock_guard<std::mutex> lock(dev_mu);
        if (!D.cus) {
          hipDeviceProp_t prop;
          D.cus = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 256;
        }
      }
      const int cus = D.cus;
      // workgroups per CU: the fp32 K-step of 32 is 1.7 us of MFMA (covers a gathered row's latency with one workgroup);
      // the split-bf16 one is 0.4 us: two workgroups keep twice the loads in flight (VDETR_SP_PER_CU overrides)
      const int split_pc = VDETR_AB("VDETR_SP_SPLIT", 1);
      const int per_cu_env = VDETR_AB("VDETR_SP_PER_CU", 0);
      const int per_cu = per_cu_env > 0 ? per_cu_env : (ksub == 2 && !split_pc ? 1 : 2);
      const int nwork = ntiles * ceil_div(CB, 128);
      // `spare` CUs are left to whatever else is running (A/B switch): with a grid of exactly one workgroup per CU, a CU that is
      // busy with another stream's long kernel (the next scene's 9 ms sampling) makes its workgroup start a round late
      const int spare = VDETR_AB("VDETR_SP_SPARE_CUS", 0);
      const int slots = per_cu * (cus - spare) > 0 ? per_cu * (cus - spare) : 1;
      dim3 grid(nwork < slots ? nwork : slots);
      // VDETR_SP_SPLIT=0: exact fp32 products (v_mfma_f32_16x16x4_f32) everywhere; default: split-bf16 products where the
      // contraction is a multiple of 64 channels (the wide layers, which hold the time)
      const int split = VDETR_AB("VDETR_SP_SPLIT", 1);
      auto kern = ksub == 2 ? (split ? (transposed ? sp_pairs_gemm_persistent_kernel<true, 2, true> : sp_pairs_gemm_persistent_kernel<false, 2, true>)
                                     : (transposed ? sp_pairs_gemm_persistent_kernel<true, 2> : sp_pairs_gemm_persistent_kernel<false, 2>))
                            : (transposed ? sp_pairs_gemm_persistent_kernel<true, 1> : sp_pairs_gemm_persistent_kernel<false, 1>);
      int* ticket = nullptr;
      const int use_ticket = VDETR_AB("VDETR_SP_TICKET", 1);  // A/B switch
      if (use_ticket) {  // a (work counter, exit counter) pair per launch out of a ring: launches of different streams may overlap
        constexpr unsigned kRing = 4096;
        if (!D.ring) {  // zeroed once; every launch leaves its pair zeroed again (the kernel's last workgroup resets it)
          std::lock_guard<std::mutex> lock(dev_mu);
          if (!D.ring) {
            hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing(st, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) {
              set_error("sp_pairs_gemm: the first launch on a device allocates its work counters and cannot be captured into a "
                        "hipGraph: run the layer once eagerly first");
              return VDETR_ERR_LAUNCH;
            }
            int* r = nullptr;
            if (hipMalloc(&r, 2 * kRing * sizeof(int)) != hipSuccess || hipMemset(r, 0, 2 * kRing * sizeof(int)) != hipSuccess) {
              set_error("sp_pairs_gemm: cannot allocate the work counters");
              return VDETR_ERR_LAUNCH;
            }
            D.ring = r;
          }
        }
        ticket = D.ring + 2 * (D.next.fetch_add(1) % kRing);
      }
      hipLaunchKernelGGL(kern, grid, dim3(256), 0, st, x, arow, w, tiles, ntiles, CA, CB, cin * cout, y, ticket);
      return check_launch("sp_pairs_gemm");
    }
    dim3 grid(ntiles, ceil_div(CB, 128));
    if (transposed)
      hipLaunchKernelGGL((sp_pairs_gemm_kernel<true, 2, 2, 4, 4>), grid, dim3(256), 0, st, x, arow, w, tiles, CA, CB, cin * cout, y);
    else
      hipLaunchKernelGGL((sp_pairs_gemm_kernel<false, 2, 2, 4, 4>), grid, dim3(256), 0, st, x, arow, w, tiles, CA, CB, cin * cout, y);
  }
  return check_launch("sp_pairs_gemm");
}

extern "C" int vdetr_sp_pairs_wgrad_f32(const float* x, const float* dy, const int32_t* pin, const int32_t* pout,
                                        const int32_t* chunks, int nchunks, int cin, int cout, float* partials,
                                        vdetr_stream_t stream) {
  VDETR_REQUIRE(nchunks >= 0 && cin > 0 && cout > 0, "sp_pairs_wgrad: bad size");
  if (nchunks == 0) return VDETR_OK;
  VDETR_REQUIRE(x && dy && pin && pout && chunks && partials, "sp_pairs_wgrad: null pointer");
  VDETR_REQUIRE(ceil_div(cout, 64) <= 65535 && ceil_div(cin, 64) <= 65535, "sp_pairs_wgrad: too many channel tiles");
  VDETR_REQUIRE(cin % 4 == 0 && cout % 4 == 0, "sp_pairs_wgrad: channel counts must be multiples of 4 (float4 rows)");
  if (cin >= 128 && cout >= 128) {
    dim3 grid(nchunks, ceil_div(cin, 128), ceil_div(cout, 128));
#ifndef VDETR_SP_WGRAD_SB
#define VDETR_SP_WGRAD_SB 16
#endif
    const int split = VDETR_AB("VDETR_SP_SPLIT", 1);
    if (split)
      hipLaunchKernelGGL((sp_pairs_wgrad_kernel<2, 2, 4, 4, 32, true>), grid, dim3(256), 0, (hipStream_t)stream, x, dy, pin, pout,
                         chunks, cin, cout, partials);
    else
      hipLaunchKernelGGL((sp_pairs_wgrad_kernel<2, 2, 4, 4, VDETR_SP_WGRAD_SB>), grid, dim3(256), 0, (hipStream_t)stream, x, dy, pin,
                         pout, chunks, cin, cout, partials);
  } else {
    dim3 grid(nchunks, ceil_div(cin, 64), ceil_div(cout, 64));
    hipLaunchKernelGGL((sp_pairs_wgrad_kernel<4, 1, 1, 4, 32>), grid, dim3(256), 0, (hipStream_t)stream, x, dy, pin, pout, chunks,
                       cin, cout, partials);
  }
  return check_launch("sp_pairs_wgrad");
}

extern "C" int vdetr_sp_wgrad_reduce_f32(const float* partials, const int32_t* seg, int K, long elems, float* dw,
                                         vdetr_stream_t stream) {
  VDETR_REQUIRE(K > 0 && elems > 0 && elems % 4 == 0, "sp_wgrad_reduce: bad size K=%d elems=%ld", K, elems);
  VDETR_REQUIRE(partials && seg && dw, "sp_wgrad_reduce: null pointer");
  const long e4 = elems / 4;
  dim3 grid((unsigned)((e4 + 255) / 256 < 1024 ? (e4 + 255) / 256 : 1024), K);
  hipLaunchKernelGGL(sp_wgrad_reduce_kernel, grid, dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const f32x4*>(partials), seg,
                     e4, reinterpret_cast<f32x4*>(dw));
  return check_launch("sp_wgrad_reduce");
}

extern "C" int vdetr_sp_pair_plan_workspace_ints(int K, int nout) { return K * ceil_div(nout > 0 ? nout : 1, 256); }

extern "C" int vdetr_sp_pair_plan_i32(const int32_t* nbr, int K, int nout, int nin, int32_t* pin, int32_t* pout, int32_t* slot,
                                      int32_t* islot, int32_t* counts, int32_t* workspace, vdetr_stream_t stream) {
  VDETR_REQUIRE(K > 0 && K <= 1024 && nout >= 0 && nin >= 0, "sp_pair_plan: bad size (K=%d nout=%d nin=%d)", K, nout, nin);
  VDETR_REQUIRE(counts, "sp_pair_plan: null pointer");
  hipStream_t st = (hipStream_t)stream;
  if (nout == 0) return hipMemsetAsync(counts, 0, sizeof(int32_t) * (K + 1), st) == hipSuccess ? VDETR_OK : VDETR_ERR_LAUNCH;
  VDETR_REQUIRE(nbr && pin && pout && slot && (islot || nin == 0) && workspace, "sp_pair_plan: null pointer");
  const int nb = ceil_div(nout, 256);
  if (nin > 0 && hipMemsetAsync(islot, 0xFF, sizeof(int32_t) * (size_t)K * nin, st) != hipSuccess) return VDETR_ERR_LAUNCH;
  hipLaunchKernelGGL(sp_plan_count_kernel, dim3(nb, K), dim3(256), 0, st, nbr, nout, nb, workspace);
  hipLaunchKernelGGL(sp_plan_scan_kernel, dim3(1), dim3(1024), 0, st, workspace, K * nb, nb, K, counts);
  hipLaunchKernelGGL(sp_plan_fill_kernel, dim3(nb, K), dim3(256), 0, st, nbr, nout, nin, nb, workspace, pin, pout, slot, islot);
  return check_launch("sp_pair_plan");
}
