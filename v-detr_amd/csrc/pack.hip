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

__global__ __launch_bounds__(256) void pack_kernel(const vdetr_pack_entry* __restrict__ entries,
                                                  const uint32_t* __restrict__ block_entry,
                                                  const uint32_t* __restrict__ block_chunk, float* __restrict__ dst) {
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
    }
    for (uint64_t i = (e4 << 2) + threadIdx.x; i < end; i += 256) out[i] = src ? src[i] : 0.f;
  } else {
    for (uint64_t i = begin + threadIdx.x; i < end; i += 256) out[i] = src ? src[i] : 0.f;
  }
}

}  // namespace vdetr

using namespace vdetr;

extern "C" int vdetr_pack_chunk_floats(void) { return kPackChunk; }

extern "C" int vdetr_pack_f32(const vdetr_pack_entry* entries, const uint32_t* block_entry, const uint32_t* block_chunk,
                              int nblocks, float* dst, vdetr_stream_t stream) {
  VDETR_REQUIRE(nblocks >= 0, "pack: negative block count");
  if (nblocks == 0) return VDETR_OK;
  VDETR_REQUIRE(entries && block_entry && block_chunk && dst, "pack: null pointer");
  hipLaunchKernelGGL(pack_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, entries, block_entry, block_chunk, dst);
  return check_launch("pack");
}
