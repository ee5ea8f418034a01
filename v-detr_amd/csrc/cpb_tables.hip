// cpb_tables.hip — the RPE tables of all decoder layers in ONE launch.
//
// Reference: models/vdetr_transformer.py:725 (`self.cpb_mlps[i](self.relative_coords_table)`: Linear(3, hidden) -> ReLU -> Linear(hidden, H,
// bias=False) on the T^3 grid points, eight MLPs per layer).  As tensor expressions for the 64 MLPs of 8 layers that was a cat (the bias
// as a fourth weight column), a batched GEMM with K = 4, a ReLU pass over a 33 MB tensor and a batched GEMM with N = 4: 40 us of the
// forward's serial prologue for 0.13 GFLOP.  Here workgroup (chunk, m) evaluates MLP m on 64 grid points: lane = point, the four waves
// split the hidden units; the hidden activations (which the tables' backward reads, attention.DeferredTableGrads) leave through an LDS
// tile as whole rows, the H outputs are reduced over the four waves in a fixed order.  fp32 fmaf chains (the GEMMs' numerics class; the
// summation ORDER differs from the library's: rounding apart).
#include "common.h"

namespace vdetr {

constexpr int kCpbPts = 64;
constexpr int kCpbMaxHid = 256;
typedef float f32x4t __attribute__((ext_vector_type(4)));

struct CpbArgs {
  const float* coords;  // [P, 3]
  const float* w1;      // [n, hid, 3]
  const float* b1;      // [n, hid]
  const float* w2;      // [n, 4, hid]
  float* hid_out;       // [n, P, hid]
  float* tables;        // [n, P, 4]
  int P, hid;
};

__global__ __launch_bounds__(256) void cpb_tables_kernel(CpbArgs A) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int hid = A.hid, stride = hid + 4;      // tile rows of hid + 4 floats: 16-byte aligned, rows on different banks
  float* wts = smem;                            // [hid][8]: w1 (3), b1, w2 (4 heads)
  float* tile = smem + hid * 8;                 // [64 points][stride]
  float* red = tile + kCpbPts * stride;         // [4 waves][4 heads][64]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int m = blockIdx.y, p0 = blockIdx.x * kCpbPts, p = p0 + lane;
  for (int j = tid; j < hid; j += 256) {
    const float* w = A.w1 + ((size_t)m * hid + j) * 3;
    float* d = wts + j * 8;
    d[0] = w[0]; d[1] = w[1]; d[2] = w[2]; d[3] = A.b1[(size_t)m * hid + j];
#pragma unroll
    for (int h = 0; h < 4; ++h) d[4 + h] = A.w2[((size_t)m * 4 + h) * hid + j];
  }
  const bool live = p < A.P;
  const float x = live ? A.coords[(size_t)p * 3] : 0.f, y = live ? A.coords[(size_t)p * 3 + 1] : 0.f, z = live ? A.coords[(size_t)p * 3 + 2] : 0.f;
  __syncthreads();
  const int per = hid >> 2, j0 = wv * per;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int j = j0; j < j0 + per; ++j) {
    const f32x4t a = *reinterpret_cast<const f32x4t*>(wts + j * 8), b = *reinterpret_cast<const f32x4t*>(wts + j * 8 + 4);  // (broadcast reads)
    const float h = fmaxf(fmaf(z, a[2], fmaf(y, a[1], fmaf(x, a[0], a[3]))), 0.f);
    tile[lane * stride + j] = h;
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] = fmaf(h, b[e], acc[e]);
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) red[(wv * 4 + e) * kCpbPts + lane] = acc[e];
  __syncthreads();
  if (wv == 0 && live) {
    f32x4t o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (red[e * kCpbPts + lane] + red[(4 + e) * kCpbPts + lane]) + (red[(8 + e) * kCpbPts + lane] + red[(12 + e) * kCpbPts + lane]);
    *reinterpret_cast<f32x4t*>(A.tables + ((size_t)m * A.P + p) * 4) = o;
  }
  // the hidden activations of the 64 points: rows of hid floats, written 16 bytes per thread
  const int q4 = hid >> 2;  // float4 per row
  for (int i = tid; i < kCpbPts * q4; i += 256) {
    const int r = i / q4, c4 = i - r * q4;
    if (p0 + r < A.P)
      *reinterpret_cast<f32x4t*>(A.hid_out + ((size_t)m * A.P + p0 + r) * hid + 4 * c4) = *reinterpret_cast<const f32x4t*>(tile + r * stride + 4 * c4);
  }
}

}  // namespace vdetr

using namespace vdetr;

extern "C" int vdetr_cpb_tables_f32(const float* coords, const float* w1, const float* b1, const float* w2, int n, int P, int hid, int H,
                                    float* hid_out, float* tables, vdetr_stream_t stream) {
  VDETR_REQUIRE(coords && w1 && b1 && w2 && hid_out && tables, "cpb_tables: null pointer");
  VDETR_REQUIRE(n > 0 && n <= 65535 && P > 0, "cpb_tables: n=%d P=%d", n, P);
  VDETR_REQUIRE(H == 4 && hid >= 16 && hid <= kCpbMaxHid && hid % 16 == 0, "cpb_tables: built for 4 heads and a hidden width of 16..%d in steps of 16 (H=%d, hidden=%d)", kCpbMaxHid, H, hid);
  VDETR_REQUIRE((((uintptr_t)hid_out | (uintptr_t)tables) & 15) == 0, "cpb_tables: outputs must be 16-B aligned");
  CpbArgs A{coords, w1, b1, w2, hid_out, tables, P, hid};
  const size_t lds = ((size_t)hid * 8 + (size_t)kCpbPts * (hid + 4) + 16 * kCpbPts) * sizeof(float);
  if (int e = set_lds(cpb_tables_kernel, lds, "cpb_tables")) return e;
  hipLaunchKernelGGL(cpb_tables_kernel, dim3(ceil_div(P, kCpbPts), n), dim3(256), lds, (hipStream_t)stream, A);
  return check_launch("cpb_tables");
}
