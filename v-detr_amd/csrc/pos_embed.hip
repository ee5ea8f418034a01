// pos_embed.hip — the fixed coordinate embeddings of models/position_embedding.py:51-127 as one launch each.
//
// The reference builds them from ~10 ATen ops (clone, shift_scale_points, mul, mm, sin, cos, cat, permute); the
// permute leaves a (B, d_pos, N) VIEW of a channel-last tensor that the first consumer copies.  Here one kernel
// normalises, projects and writes the channel-major result directly: a thread owns one point, walks the channels,
// and a wave's 64 stores of a channel are 256 contiguous bytes.  HBM-bound on the output (4 * d_pos * N bytes per
// scene); the input is 12 N bytes.  Arithmetic in the reference's order: ((x - min) * 1) / (max - min) + 0, * 2pi,
// the projection accumulated over the axes in order.
#include "common.h"

namespace vdetr {

constexpr int kPeThreads = 256;
constexpr int kPeChunk = 32;  // channels (sin/cos pairs for the fourier kind) per workgroup

__device__ __forceinline__ float pe_normalise(float v, const float* rmin, const float* rmax, int b, int a) {
  if (!rmin) return v;
  const float lo = rmin[b * 3 + a], hi = rmax[b * 3 + a];
  return (v - lo) / (hi - lo);  // shift_scale_points to [0,1]: the * 1 and + 0 of the general form are exact no-ops
}

// out (b, 2*d_out, n): channel c < d_out: sin(2pi x_n . B[:,c]), channel d_out + c: cos of the same
__global__ __launch_bounds__(kPeThreads) void pos_embed_fourier_kernel(const float* __restrict__ xyz, int n,
                                                                       const float* __restrict__ rmin, const float* __restrict__ rmax,
                                                                       const float* __restrict__ gauss_b, int ldb, int d_out,
                                                                       float* __restrict__ out) {
  const int i = blockIdx.x * kPeThreads + threadIdx.x, b = blockIdx.z;
  const int c0 = blockIdx.y * kPeChunk, c1 = min(c0 + kPeChunk, d_out);
  if (i >= n) return;
  float x[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) x[a] = pe_normalise(xyz[((size_t)b * n + i) * 3 + a], rmin, rmax, b, a) * 6.2831855f;  // fl32(2 pi)
  float* o = out + (size_t)b * 2 * d_out * n + i;
  for (int c = c0; c < c1; ++c) {
    const float proj = __fmaf_rn(x[2], gauss_b[2 * ldb + c], __fmaf_rn(x[1], gauss_b[ldb + c], x[0] * gauss_b[c]));
    float s, co;
    sincosf(proj, &s, &co);
    o[(size_t)c * n] = s;
    o[(size_t)(d_out + c) * n] = co;
  }
}

// out (b, num_channels, n): per axis d a block of cdim channels, channel i of the block =
//   (i even ? sin : cos)(x_d * scale / temperature^(2 floor(i/2) / cdim));  cdim = ndim (+2 for the first axes while
//   channels remain), ndim = even part of num_channels / 3   (position_embedding.py:60-94)
__global__ __launch_bounds__(kPeThreads) void pos_embed_sine_kernel(const float* __restrict__ xyz, int n, const float* __restrict__ rmin,
                                                                    const float* __restrict__ rmax, int num_channels, float temperature,
                                                                    float scale, float* __restrict__ out) {
  const int i = blockIdx.x * kPeThreads + threadIdx.x, b = blockIdx.z;
  const int c0 = blockIdx.y * kPeChunk, c1 = min(c0 + kPeChunk, num_channels);
  if (i >= n) return;
  int ndim = num_channels / 3;
  ndim -= ndim % 2;
  int rems = num_channels - ndim * 3;
  float* o = out + (size_t)b * num_channels * n + i;
  int start = 0;
  for (int d = 0; d < 3; ++d) {
    int cdim = ndim;
    if (rems > 0) { cdim += 2; rems -= 2; }
    const int lo = max(c0, start), hi = min(c1, start + cdim);
    if (lo < hi) {
      float raw = pe_normalise(xyz[((size_t)b * n + i) * 3 + d], rmin, rmax, b, d);
      if (scale != 0.f) raw *= scale;  // `if self.scale:` (:81-82)
      for (int c = lo; c < hi; ++c) {
        const int k = c - start;
        const float dim_t = powf(temperature, (float)(2 * (k / 2)) / (float)cdim);
        const float pos = raw / dim_t;
        o[(size_t)c * n] = (k & 1) ? cosf(pos) : sinf(pos);
      }
    }
    start += cdim;
  }
}

}  // namespace vdetr

using namespace vdetr;

extern "C" int vdetr_pos_embed_fourier_f32(const float* xyz, int b, int n, const float* range_min, const float* range_max,
                                           const float* gauss_b, int ldb, int d_out, float* out, vdetr_stream_t stream) {
  VDETR_REQUIRE(b >= 0 && n >= 0 && d_out >= 0 && ldb >= d_out, "pos_embed_fourier: bad dimensions b=%d n=%d d_out=%d ldb=%d", b, n, d_out, ldb);
  if (b == 0 || n == 0 || d_out == 0) return VDETR_OK;
  VDETR_REQUIRE(xyz && gauss_b && out, "pos_embed_fourier: null pointer");
  VDETR_REQUIRE((range_min == nullptr) == (range_max == nullptr), "pos_embed_fourier: give both range ends or neither");
  const dim3 grid(ceil_div(n, kPeThreads), ceil_div(d_out, kPeChunk), b);
  hipLaunchKernelGGL(pos_embed_fourier_kernel, grid, dim3(kPeThreads), 0, (hipStream_t)stream, xyz, n, range_min, range_max, gauss_b,
                     ldb, d_out, out);
  return check_launch("pos_embed_fourier");
}

extern "C" int vdetr_pos_embed_sine_f32(const float* xyz, int b, int n, const float* range_min, const float* range_max,
                                        int num_channels, float temperature, float scale, float* out, vdetr_stream_t stream) {
  VDETR_REQUIRE(b >= 0 && n >= 0 && num_channels >= 0, "pos_embed_sine: bad dimensions b=%d n=%d channels=%d", b, n, num_channels);
  VDETR_REQUIRE(num_channels % 2 == 0, "pos_embed_sine: num_channels=%d must be even", num_channels);
  if (b == 0 || n == 0 || num_channels == 0) return VDETR_OK;
  VDETR_REQUIRE(xyz && out, "pos_embed_sine: null pointer");
  VDETR_REQUIRE((range_min == nullptr) == (range_max == nullptr), "pos_embed_sine: give both range ends or neither");
  const dim3 grid(ceil_div(n, kPeThreads), ceil_div(num_channels, kPeChunk), b);
  hipLaunchKernelGGL(pos_embed_sine_kernel, grid, dim3(kPeThreads), 0, (hipStream_t)stream, xyz, n, range_min, range_max, num_channels,
                     temperature, scale, out);
  return check_launch("pos_embed_sine");
}
