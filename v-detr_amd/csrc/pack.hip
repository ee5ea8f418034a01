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

}  // namespace vdetr

using namespace vdetr;

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
