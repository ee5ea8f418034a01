// morton.hip — Z-order permutation of the key points of a scene in ONE launch.
//
// The decoder sorts the encoder tokens along a 30-bit Morton curve of their bounding box before the cross attention
// (v-detr_amd/vdetr_transformer.py: attention does not depend on the order of its keys, the RPE kernels' LDS broadcast
// and table-gradient grouping do).  As tensor expressions that is ~50 launches on [B,4096] tensors (min, max, quantise,
// 3 x 4 shift/or/and rounds, argsort); here one workgroup per scene does the bounding box, the codes and — up to 8192
// points — a bitonic sort of (code, index) with the keys in registers (wgsort.h, round 6: lane exchanges inside a wave, LDS only across
// waves: 43 -> ~9 us for 4096 points).  Arithmetic follows pc_util.morton_argsort operation by operation ((x - lo) / ext * 1023 in fp32,
// truncation, clamp), ties are ordered by index (a stable sort).  The same sort orders the decoder's proposals (topk_order_kernel).
#include "common.h"
#include "wave.h"
#include "wgsort.h"

namespace vdetr {

constexpr int kMortonSortMax = 8192;  // 64 KB of LDS keys

__device__ __forceinline__ unsigned spread10(unsigned v) {
  v = (v | (v << 16)) & 0x030000FFu;
  v = (v | (v << 8)) & 0x0300F00Fu;
  v = (v | (v << 4)) & 0x030C30C3u;
  return (v | (v << 2)) & 0x09249249u;
}

template <int E>
__global__ __launch_bounds__(1024) void morton_order_kernel(const float* __restrict__ xyz, int n, int* __restrict__ codes, long long* __restrict__ order) {
  extern __shared__ unsigned long long keys[];
  __shared__ float red[16][6];
  __shared__ float box[6];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
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
  auto key_of = [&](int i) {
    unsigned q[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float t = __fmul_rn(__fdiv_rn(__fsub_rn(p[3 * i + a], lo[a]), ext[a]), 1023.0f);
      long long v = (long long)t;  // truncation, as Tensor.long()
      v = v < 0 ? 0 : (v > 1023 ? 1023 : v);
      q[a] = (unsigned)v;
    }
    const unsigned code = spread10(q[0]) | (spread10(q[1]) << 1) | (spread10(q[2]) << 2);
    if (codes) codes[(size_t)b * n + i] = (int)code;
    return ((unsigned long long)code << 32) | (unsigned)i;
  };
  if (order == nullptr) {  // the codes only (more points than the sort takes: the caller sorts them)
    for (int i = tid; i < n; i += 1024) key_of(i);
    return;
  }
  unsigned long long key[E];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int i = tid * E + e;
    key[e] = i < n ? key_of(i) : ~0ull;
  }
  wg_bitonic_sort<E>(key, keys, tid);
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int i = tid * E + e;
    if (i < n) order[(size_t)b * n + i] = (long long)(key[e] & 0xFFFFFFFFull);
  }
}

// indices of the nq largest of n values per row, largest first, equal values by ascending index — torch.sort(descending=True,
// stable=True)[1][:, :nq] (the decoder's proposal order, models/vdetr_transformer.py:364-366) — as one launch of one workgroup per row
template <int E>
__global__ __launch_bounds__(1024) void topk_order_kernel(const float* __restrict__ values, int n, int nq, long long* __restrict__ order) {
  extern __shared__ unsigned long long keys[];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* v = values + (size_t)b * n;
  unsigned long long key[E];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int i = tid * E + e;
    key[e] = ~0ull;
    if (i < n) {
      unsigned u = __float_as_uint(v[i]);
      u = u == 0x80000000u ? 0u : u;                   // (-0.0 == +0.0: they tie, as in a comparison sort)
      u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // order-preserving map of the floats to unsigned
      key[e] = ((unsigned long long)(~u) << 32) | (unsigned)i;  // descending values, ascending indices among equals
    }
  }
  wg_bitonic_sort<E>(key, keys, tid);
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int i = tid * E + e;
    if (i < nq) order[(size_t)b * nq + i] = (long long)(key[e] & 0xFFFFFFFFull);
  }
}

}  // namespace vdetr

using namespace vdetr;

extern "C" int vdetr_morton_sort_max(void) { return kMortonSortMax; }

#define VDETR_SORT_DISPATCH(KERNEL, GRID, ...)                                                                     \
  do {                                                                                                             \
    const size_t lds_ = (size_t)1024 * e_ * sizeof(unsigned long long);                                            \
    switch (e_) {                                                                                                  \
      case 1: hipLaunchKernelGGL(KERNEL<1>, GRID, dim3(1024), lds_, st_, __VA_ARGS__); break;                      \
      case 2: hipLaunchKernelGGL(KERNEL<2>, GRID, dim3(1024), lds_, st_, __VA_ARGS__); break;                      \
      case 4: hipLaunchKernelGGL(KERNEL<4>, GRID, dim3(1024), lds_, st_, __VA_ARGS__); break;                      \
      default:                                                                                                     \
        if (int rc_ = set_lds(KERNEL<8>, lds_, "sort")) return rc_;                                                \
        hipLaunchKernelGGL(KERNEL<8>, GRID, dim3(1024), lds_, st_, __VA_ARGS__);                                   \
        break;                                                                                                     \
    }                                                                                                              \
  } while (0)

static int sort_slots(int n) {  // keys per thread of the 1024-thread sort: 1, 2, 4 or 8
  int e = 1;
  while (1024 * e < n) e <<= 1;
  return e;
}

extern "C" int vdetr_morton_order_f32(const float* xyz, int B, int n, int* codes, long long* order, vdetr_stream_t stream) {
  VDETR_REQUIRE(xyz && (codes || order), "morton_order: null pointer");
  VDETR_REQUIRE(B > 0 && n > 0, "morton_order: bad shape B=%d n=%d", B, n);
  VDETR_REQUIRE(!order || n <= kMortonSortMax, "morton_order: n=%d exceeds the in-LDS sort (%d): pass order=NULL and sort the codes",
                n, kMortonSortMax);
  hipStream_t st_ = (hipStream_t)stream;
  const int e_ = order ? sort_slots(n) : 1;
  VDETR_SORT_DISPATCH(morton_order_kernel, dim3(B), xyz, n, codes, order);
  return check_launch("morton_order");
}

extern "C" int vdetr_topk_order_f32(const float* values, int B, int n, int nq, long long* order, vdetr_stream_t stream) {
  VDETR_REQUIRE(values && order, "topk_order: null pointer");
  VDETR_REQUIRE(B > 0 && n > 0 && nq > 0 && nq <= n, "topk_order: bad shape B=%d n=%d nq=%d", B, n, nq);
  VDETR_REQUIRE(n <= kMortonSortMax, "topk_order: n=%d exceeds the workgroup sort (%d)", n, kMortonSortMax);
  hipStream_t st_ = (hipStream_t)stream;
  const int e_ = sort_slots(n);
  VDETR_SORT_DISPATCH(topk_order_kernel, dim3(B), values, n, nq, order);
  return check_launch("topk_order");
}
