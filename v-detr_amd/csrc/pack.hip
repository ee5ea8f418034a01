// pack.hip — many small tensors -> one flat buffer in ONE launch.
//
// The gradients of the 796 parameter tensors come out of autograd as separate allocations; the flat optimizer /
// all-reduce buffers (v-detr_amd/dist.py:FlatParams) want them contiguous.  torch._foreach_copy_ needs ~49 launches
// for that (kernel-argument space bounds the tensors per launch); here the (source, offset, length) table lives in
// device memory, so the launch count is 1 and the copy runs at HBM speed.
// Replaces the bucket copy of DistributedDataParallel's reducer (reference main.py:515-517 wraps the model in DDP).
#include "common.h"

namespace vdetr {

constexpr int kPackChunk = 8192;  // floats per workgroup

template <bool SUMSQ>
__global__ __launch_bounds__(256) void pack_kernel(const vdetr_pack_entry* __restrict__ entries,
                                                  const uint32_t* __restrict__ block_entry,
                                                  const uint32_t* __restrict__ block_chunk, float* __restrict__ dst,
                                                  float* __restrict__ sumsq) {
  float acc = 0.f;  // SUMSQ: sum of squares of this thread's elements (the gradient norm's first half, optim.hip)
  const vdetr_pack_entry e = entries[block_entry[blockIdx.x]];
  const uint64_t begin = (uint64_t)block_chunk[blockIdx.x] * kPackChunk;
  const uint64_t end = begin + kPackChunk < e.numel ? begin + kPackChunk : e.numel;
  const float* src = static_cast<const float*>(e.src);
  float* out = dst + e.dst_offset;
  const bool vec = ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
  if (vec) {
    const uint64_t b4 = begin >> 2, e4 = end >> 2;  // begin is a multiple of 4
    for (uint64_t i = b4 + threadIdx.x; i < e4; i += 256) {
      float4 v = src ? reinterpret_cast<const float4*>(src)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      reinterpret_cast<float4*>(out)[i] = v;
      if (SUMSQ) acc += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
    for (uint64_t i = (e4 << 2) + threadIdx.x; i < end; i += 256) {
      const float x = src ? src[i] : 0.f;
      out[i] = x;
      if (SUMSQ) acc += x * x;
    }
  } else {
    for (uint64_t i = begin + threadIdx.x; i < end; i += 256) {
      const float x = src ? src[i] : 0.f;
      out[i] = x;
      if (SUMSQ) acc += x * x;
    }
  }
  if constexpr (SUMSQ) {  // fixed order: lanes by xor-shuffle, the four waves through LDS
    __shared__ float red[4];
    for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) sumsq[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
  }
}

// out[c] = sum over rows of x[row, c] — the bias gradient of a Linear / 1x1 convolution.  ATen's generic reduction needs
// two launches (14 us) for these tall-skinny [1024..4096, 64..1280] matrices; here one workgroup per 64-column strip,
// 16 waves striding the rows with 4 independent accumulators, combined through LDS in a fixed order (deterministic).
__global__ __launch_bounds__(1024) void colsum_kernel(const float* __restrict__ x, float* __restrict__ out, int rows, int cols,
                                                      long row_stride) {
  __shared__ float part[16][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (c < cols) {
    const float* p = x + c;
    int r = wv;
    for (; r + 48 < rows; r += 64) {
      a0 += p[(size_t)r * row_stride];
      a1 += p[(size_t)(r + 16) * row_stride];
      a2 += p[(size_t)(r + 32) * row_stride];
      a3 += p[(size_t)(r + 48) * row_stride];
    }
    for (; r < rows; r += 16) a0 += p[(size_t)r * row_stride];
  }
  part[wv][lane] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (wv == 0 && c < cols) {
    float s = 0.f;
#pragma unroll
    for (int w2 = 0; w2 < 16; ++w2) s += part[w2][lane];
    out[c] = s;
  }
}

// The same for n matrices, with the rows of a tall one split over S workgroups per strip: out[i][c] = sum_r x[i][r][c].  A [4096 x 1024]
// matrix has 16 strips of 64 columns — 16 workgroups walked 16 MB in 31 us on the tail of the backward (a latency chain of 64 dependent
// load batches per wave), and ATen's reduction of a [2, 4096, 256] stack took 36.  Here a lane owns four columns (one 16-byte load per
// row), a wave has eight rows in flight, a workgroup (4 waves) takes rows / S rows of a 256-column strip; with S > 1 a second launch of
// the same kernel adds the [S x cols] partial sums (two launches of ~3 us; fixed summation order).  Not one launch with the last
// workgroup adding up: the device-scope fence that hands the partials from one XCD's L2 to another's costs more than the launch
// (measured: 43-68 us for the shapes below).
typedef float f32x4c __attribute__((ext_vector_type(4)));
struct ColsumB {
  const float* x; float* out;
  const float* const* items;  // or nullptr: item i starts at x + i * item_stride
  int rows, cols, S;
  long row_stride, item_stride;
};
__global__ __launch_bounds__(256) void colsum4_kernel(ColsumB A) {
  __shared__ f32x4c part[4][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int strip = blockIdx.x, sp = blockIdx.y, item = blockIdx.z;
  const int c = strip * 256 + 4 * lane;
  const bool live = c < A.cols;
  const int per = (A.rows + A.S - 1) / A.S, r0 = sp * per, r1 = min(A.rows, r0 + per);
  const f32x4c zero = {0.f, 0.f, 0.f, 0.f};
  f32x4c acc[2] = {zero, zero};
  if (live) {
    const float* p = (A.items ? A.items[item] : A.x + (size_t)item * A.item_stride) + c;
    int r = r0 + wv;
    for (; r + 28 < r1; r += 32) {
      f32x4c v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4c*>(p + (size_t)(r + 4 * u) * A.row_stride);
#pragma unroll
      for (int u = 0; u < 8; ++u) acc[u & 1] += v[u];
    }
    for (; r < r1; r += 4) acc[0] += *reinterpret_cast<const f32x4c*>(p + (size_t)r * A.row_stride);
  }
  part[wv][lane] = acc[0] + acc[1];
  __syncthreads();
  f32x4c v = zero;
  if (wv == 0) v = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
  // S == 1: the sums themselves; S > 1: this workgroup's partial sums, row sp of a [n, S, cols] table that a second launch adds up
  if (wv == 0 && live) *reinterpret_cast<f32x4c*>(A.out + ((size_t)item * A.S + sp) * A.cols + c) = v;
}

// (columns or strides that are not multiples of four: one lane per column, no split)
__global__ __launch_bounds__(1024) void colsum_batched_kernel(ColsumB A) {
  __shared__ float part[16][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int item = blockIdx.z;
  const int c = blockIdx.x * 64 + lane;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (c < A.cols) {
    const float* p = (A.items ? A.items[item] : A.x + (size_t)item * A.item_stride) + c;
    int r = wv;
    for (; r + 48 < A.rows; r += 64) {
      a0 += p[(size_t)r * A.row_stride];
      a1 += p[(size_t)(r + 16) * A.row_stride];
      a2 += p[(size_t)(r + 32) * A.row_stride];
      a3 += p[(size_t)(r + 48) * A.row_stride];
    }
    for (; r < A.rows; r += 16) a0 += p[(size_t)r * A.row_stride];
  }
  part[wv][lane] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (wv == 0 && c < A.cols) {
    float v = 0.f;
#pragma unroll
    for (int w2 = 0; w2 < 16; ++w2) v += part[w2][lane];
    A.out[(size_t)item * A.cols + c] = v;
  }
}

}  // namespace vdetr

using namespace vdetr;

static bool colsum_vec(const float* x, int cols, long row_stride, long item_stride) {
  return cols % 4 == 0 && row_stride % 4 == 0 && item_stride % 4 == 0 && (((uintptr_t)x) & 15) == 0;
}
// row splits of the four-columns-per-lane kernel: towards ~512 workgroups, at least 32 rows each
static int colsum_splits(int n, int rows, int cols) {
  if (cols % 4) return 1;
  const long wgs = (long)n * ceil_div(cols, 256);
  int S = 1;
  while (wgs * S < 512 && S < 256 && rows / (2 * S) >= 32) S *= 2;
  return S;
}

extern "C" size_t vdetr_colsum_workspace_bytes(int n, int rows, int cols) {
  if (n <= 0 || rows <= 0 || cols <= 0) return 0;
  const int S = colsum_splits(n, rows, cols);
  return S > 1 ? (size_t)n * S * cols * sizeof(float) : 0;
}

// items != nullptr: a DEVICE array of n pointers to 16-byte aligned matrices (vdetr_colsum_ptrs_f32)
static int colsum_run(const float* x, const float* const* items, float* out, int n, int rows, int cols, long row_stride, long item_stride,
                      void* workspace, size_t workspace_bytes, vdetr_stream_t stream) {
  VDETR_REQUIRE((x || items) && out, "colsum_batched: null pointer");
  VDETR_REQUIRE(n > 0 && n <= 65535 && rows > 0 && cols > 0 && row_stride >= cols, "colsum_batched: bad shape n=%d rows=%d cols=%d stride=%ld", n, rows, cols, row_stride);
  const bool vec = (items ? (cols % 4 == 0 && row_stride % 4 == 0) : colsum_vec(x, cols, row_stride, item_stride)) && (((uintptr_t)out) & 15) == 0;
  const size_t need = vdetr_colsum_workspace_bytes(n, rows, cols);
  const int S = (vec && need && workspace && workspace_bytes >= need && (((uintptr_t)workspace) & 15) == 0) ? colsum_splits(n, rows, cols) : 1;
  ColsumB A;
  A.x = x; A.items = items; A.rows = rows; A.cols = cols; A.S = S; A.row_stride = row_stride; A.item_stride = item_stride;
  A.out = S > 1 ? reinterpret_cast<float*>(workspace) : out;
  if (vec) {
    hipLaunchKernelGGL(colsum4_kernel, dim3(ceil_div(cols, 256), S, n), dim3(256), 0, (hipStream_t)stream, A);
    if (S > 1) {  // the [n, S, cols] partial sums -> out
      ColsumB B;
      B.x = A.out; B.items = nullptr; B.out = out; B.rows = S; B.cols = cols; B.S = 1; B.row_stride = cols; B.item_stride = (long)S * cols;
      hipLaunchKernelGGL(colsum4_kernel, dim3(ceil_div(cols, 256), 1, n), dim3(256), 0, (hipStream_t)stream, B);
    }
  } else {
    hipLaunchKernelGGL(colsum_batched_kernel, dim3(ceil_div(cols, 64), 1, n), dim3(1024), 0, (hipStream_t)stream, A);
  }
  return check_launch("colsum_batched");
}

extern "C" int vdetr_colsum_batched_f32(const float* x, float* out, int n, int rows, int cols, long row_stride, long item_stride,
                                        void* workspace, size_t workspace_bytes, vdetr_stream_t stream) {
  return colsum_run(x, nullptr, out, n, rows, cols, row_stride, item_stride, workspace, workspace_bytes, stream);
}

extern "C" int vdetr_colsum_ptrs_f32(const float* const* items, float* out, int n, int rows, int cols, long row_stride, void* workspace,
                                     size_t workspace_bytes, vdetr_stream_t stream) {
  VDETR_REQUIRE(cols % 4 == 0 && row_stride % 4 == 0, "colsum_ptrs: cols %d and row_stride %ld must be multiples of 4 (16-byte aligned items)", cols, row_stride);
  return colsum_run(nullptr, items, out, n, rows, cols, row_stride, 0, workspace, workspace_bytes, stream);
}

extern "C" int vdetr_colsum_f32(const float* x, float* out, int rows, int cols, long row_stride, vdetr_stream_t stream) {
  VDETR_REQUIRE(x && out, "colsum: null pointer");
  VDETR_REQUIRE(rows > 0 && cols > 0 && row_stride >= cols, "colsum: bad shape rows=%d cols=%d stride=%ld", rows, cols, row_stride);
  hipLaunchKernelGGL(colsum_kernel, dim3(ceil_div(cols, 64)), dim3(1024), 0, (hipStream_t)stream, x, out, rows, cols, row_stride);
  return check_launch("colsum");
}

extern "C" int vdetr_pack_chunk_floats(void) { return kPackChunk; }

extern "C" int vdetr_pack_f32(const vdetr_pack_entry* entries, const uint32_t* block_entry, const uint32_t* block_chunk,
                              int nblocks, float* dst, vdetr_stream_t stream) {
  VDETR_REQUIRE(nblocks >= 0, "pack: negative block count");
  if (nblocks == 0) return VDETR_OK;
  VDETR_REQUIRE(entries && block_entry && block_chunk && dst, "pack: null pointer");
  hipLaunchKernelGGL(pack_kernel<false>, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, entries, block_entry, block_chunk, dst, (float*)nullptr);
  return check_launch("pack");
}

extern "C" int vdetr_pack_sumsq_f32(const vdetr_pack_entry* entries, const uint32_t* block_entry, const uint32_t* block_chunk,
                                    int nblocks, float* dst, float* sumsq, vdetr_stream_t stream) {
  VDETR_REQUIRE(nblocks > 0, "pack_sumsq: no blocks");
  VDETR_REQUIRE(entries && block_entry && block_chunk && dst && sumsq, "pack_sumsq: null pointer");
  hipLaunchKernelGGL(pack_kernel<true>, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, entries, block_entry, block_chunk, dst, sumsq);
  return check_launch("pack_sumsq");
}
