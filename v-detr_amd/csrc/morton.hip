// morton.hip — Z-order permutation of the key points of a scene, and the proposals' stable descending order, one launch each.
//
// The decoder sorts the encoder tokens along a 30-bit Morton curve of their bounding box before the cross attention
// (v-detr_amd/vdetr_transformer.py: attention does not depend on the order of its keys, the RPE kernels' LDS broadcast
// and table-gradient grouping do).  As tensor expressions that is ~50 launches on [B,4096] tensors (min, max, quantise,
// 3 x 4 shift/or/and rounds, argsort); here every workgroup does the bounding box and the codes of the whole scene and — up to 16384
// points — ranks its 64 points by counting (below).  Arithmetic follows pc_util.morton_argsort operation by operation ((x - lo) / ext *
// 1023 in fp32, truncation, clamp), ties are ordered by index (a stable sort).
#include "common.h"
#include "wave.h"

namespace vdetr {


__device__ __forceinline__ unsigned spread10(unsigned v) {
  v = (v | (v << 16)) & 0x030000FFu;
  v = (v | (v << 8)) & 0x0300F00Fu;
  v = (v | (v << 4)) & 0x030C30C3u;
  return (v | (v << 2)) & 0x09249249u;
}

// ---- orders by COUNTING, on many workgroups (round 6) -------------------------------------------------------------------------------
// One workgroup sorting 4096 keys is bound by its one CU: the bitonic sort with the keys in registers (wgsort.h of this round's
// first half; git log keeps it) took 40 us for the Morton order and 34 us for the proposals' order inside the step — 16 waves share
// one LDS pipe for the lane exchanges —, both on the forward's serial chain.  The rank of a key among n DISTINCT keys is
// the number of keys below it: n^2 = 16.7 M comparisons, nothing for the chip.  Workgroup g owns 64 elements; each of its 16 waves
// counts, for the wave's lane = element, the keys of one sixteenth of the row that are smaller (the keys sit in LDS, every lane reads
// the same one: a broadcast), the 16 counts are added through LDS, and the element's index is stored at its rank.  Every workgroup
// builds all n keys itself (the bounding box and the codes are 48 KB of reads): no second launch, no exchange.  The keys carry the
// element's index in their low half: distinct, and ties come out in index order (a stable sort).
constexpr int kRankMax = 16384;  // 128 KB of LDS keys
constexpr int kRankElems = 64;

__device__ __forceinline__ void rank_and_store(const unsigned long long* keys, int n, int nout, unsigned* cnt, long long* __restrict__ order_row) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int e = blockIdx.x * kRankElems + lane;
  const unsigned long long mine = e < n ? keys[e] : ~0ull;
  const int per = (n + 15) >> 4, i0 = wv * per, i1 = min(n, i0 + per);
  unsigned c0 = 0, c1 = 0, c2 = 0, c3 = 0;
  int i = i0;
  for (; i + 3 < i1; i += 4) {
    c0 += keys[i] < mine;
    c1 += keys[i + 1] < mine;
    c2 += keys[i + 2] < mine;
    c3 += keys[i + 3] < mine;
  }
  for (; i < i1; ++i) c0 += keys[i] < mine;
  cnt[wv * kRankElems + lane] = (c0 + c1) + (c2 + c3);
  __syncthreads();
  if (wv == 0 && e < n) {
    unsigned r = 0;
#pragma unroll
    for (int w2 = 0; w2 < 16; ++w2) r += cnt[w2 * kRankElems + lane];
    if ((int)r < nout) order_row[r] = (long long)e;
  }
}

__global__ __launch_bounds__(1024) void morton_rank_kernel(const float* __restrict__ xyz, int n, int* __restrict__ codes, long long* __restrict__ order) {
  extern __shared__ unsigned long long keys[];
  __shared__ float red[16][6];
  __shared__ float box[6];
  __shared__ unsigned cnt[16 * kRankElems];
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const float* p = xyz + (size_t)b * n * 3;
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int i = tid; i < n; i += 1024) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float v = p[3 * i + a];
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
    for (int a = 0; a < 3; ++a) {
      red[wv][a] = lo[a];
      red[wv][3 + a] = hi[a];
    }
  }
  __syncthreads();
  if (tid < 6) {
    float v = red[0][tid];
    for (int w = 1; w < 16; ++w) v = tid < 3 ? fminf(v, red[w][tid]) : fmaxf(v, red[w][tid]);
    box[tid] = v;
  }
  __syncthreads();
  float ext[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    lo[a] = box[a];
    ext[a] = fmaxf(__fsub_rn(box[3 + a], box[a]), 1e-6f);
  }
  for (int i = tid; i < n; i += 1024) {  // (pc_util.morton_codes, operation by operation)
    unsigned q[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float t = __fmul_rn(__fdiv_rn(__fsub_rn(p[3 * i + a], lo[a]), ext[a]), 1023.0f);
      long long v = (long long)t;
      v = v < 0 ? 0 : (v > 1023 ? 1023 : v);
      q[a] = (unsigned)v;
    }
    const unsigned code = spread10(q[0]) | (spread10(q[1]) << 1) | (spread10(q[2]) << 2);
    if (codes && blockIdx.x == 0) codes[(size_t)b * n + i] = (int)code;
    if (order) keys[i] = ((unsigned long long)code << 32) | (unsigned)i;
  }
  if (!order) return;  // the codes only (more points than the keys' LDS takes: the caller sorts them)
  __syncthreads();
  rank_and_store(keys, n, n, cnt, order + (size_t)b * n);
}

__global__ __launch_bounds__(1024) void topk_rank_kernel(const float* __restrict__ values, int n, int nq, long long* __restrict__ order) {
  extern __shared__ unsigned long long keys[];
  __shared__ unsigned cnt[16 * kRankElems];
  const int b = blockIdx.y, tid = threadIdx.x;
  const float* v = values + (size_t)b * n;
  for (int i = tid; i < n; i += 1024) {
    unsigned u = __float_as_uint(v[i]);
    u = u == 0x80000000u ? 0u : u;                   // (-0.0 == +0.0: they tie, as in a comparison sort)
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // order-preserving map of the floats to unsigned
    keys[i] = ((unsigned long long)(~u) << 32) | (unsigned)i;  // descending values, ascending indices among equals
  }
  __syncthreads();
  rank_and_store(keys, n, nq, cnt, order + (size_t)b * nq);
}

}  // namespace vdetr

using namespace vdetr;

extern "C" int vdetr_morton_sort_max(void) { return kRankMax; }

extern "C" int vdetr_morton_order_f32(const float* xyz, int B, int n, int* codes, long long* order, vdetr_stream_t stream) {
  VDETR_REQUIRE(xyz && (codes || order), "morton_order: null pointer");
  VDETR_REQUIRE(B > 0 && B <= 65535 && n > 0, "morton_order: bad shape B=%d n=%d", B, n);
  VDETR_REQUIRE(!order || n <= kRankMax, "morton_order: n=%d exceeds the in-LDS keys (%d): pass order=NULL and sort the codes", n, kRankMax);
  // order: ranks by counting, n / 64 workgroups per scene; codes only: one workgroup per scene, no keys kept
  const size_t lds = order ? (size_t)n * sizeof(unsigned long long) : 0;
  if (int rc = set_lds(morton_rank_kernel, lds, "morton_order")) return rc;
  hipLaunchKernelGGL(morton_rank_kernel, dim3(order ? ceil_div(n, kRankElems) : 1, B), dim3(1024), lds, (hipStream_t)stream, xyz, n, codes, order);
  return check_launch("morton_order");
}

extern "C" int vdetr_topk_order_f32(const float* values, int B, int n, int nq, long long* order, vdetr_stream_t stream) {
  VDETR_REQUIRE(values && order, "topk_order: null pointer");
  VDETR_REQUIRE(B > 0 && B <= 65535 && n > 0 && nq > 0 && nq <= n, "topk_order: bad shape B=%d n=%d nq=%d", B, n, nq);
  VDETR_REQUIRE(n <= kRankMax, "topk_order: n=%d exceeds the in-LDS keys (%d)", n, kRankMax);
  const size_t lds = (size_t)n * sizeof(unsigned long long);
  if (int rc = set_lds(topk_rank_kernel, lds, "topk_order")) return rc;
  hipLaunchKernelGGL(topk_rank_kernel, dim3(ceil_div(n, kRankElems), B), dim3(1024), lds, (hipStream_t)stream, values, n, nq, order);
  return check_launch("topk_order");
}
