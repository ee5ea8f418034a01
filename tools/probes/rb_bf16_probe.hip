// rb_bf16_probe.hip — would rowblock.hip's 16 x 256 x 256 product pay on the bf16 matrix unit?  64 workgroups of 4 waves, `reps` products
// each on warm weight images, v_mfma_f32_16x16x32_bf16 with the operands as bf16 parts:
//   mode 0: weights 3 parts, activations 3 parts, 6 terms (f32-equivalent)     mode 1: weights 2 parts, activations 2 parts, 3 terms (2^-16)
//   mode 2: the weight loads of mode 0 alone     mode 3: the matrix instructions of mode 0 alone     mode 4: f32 v_mfma_f32_16x16x4_f32 reference
// build + run:  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/rb_bf16_probe.hip -o /tmp/rbb && /tmp/rbb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kC = 256;
constexpr size_t kPartBytes = (size_t)kC * kC * 2;  // one bf16 image of a [256 x 256] weight: [8 k-steps][4 kg][256 n][8 k]

template <int WP, int AP, int TERMS, bool LOADS, bool MFMA>
__device__ __forceinline__ void product(const bf16x8 (&a)[3][8], const char* __restrict__ img, int col0, int lane, f32x4 (&acc)[4]) {
  const int c = lane & 15, kg = lane >> 4;
  constexpr int D = 3;
  bf16x8 b[D][WP][4];
  auto load = [&](int ks, int slot) {
#pragma unroll
    for (int p = 0; p < WP; ++p)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
        b[slot][p][nt] = *reinterpret_cast<const bf16x8*>(img + p * kPartBytes + ((size_t)(ks * 4 + kg) * kC + col0 + 4 * c + nt) * 16);
  };
  if (LOADS) {
#pragma unroll
    for (int d = 0; d < D - 1; ++d) load(d, d);
  }
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    if (LOADS && ks + D - 1 < 8) load(ks + D - 1, (ks + D - 1) % D);
    __builtin_amdgcn_sched_barrier(0);
    if (MFMA) {
      // terms in ascending magnitude: (lo, hi), (mid, mid), (hi, lo), (mid, hi), (hi, mid), (hi, hi)
      constexpr int ta[6] = {2, 1, 0, 1, 0, 0}, tb[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
      for (int t = 6 - TERMS; t < 6; ++t) {
        const int pa = ta[t] < AP ? ta[t] : AP - 1, pb = tb[t] < WP ? tb[t] : WP - 1;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          const bf16x8 bb = LOADS ? b[ks % D][pb][nt] : a[pb][(ks + nt) & 7];
          acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[pa][ks], bb, acc[nt], 0, 0, 0);
        }
      }
    } else {
#pragma unroll
      for (int p = 0; p < WP; ++p)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt][0] += (float)b[ks % D][p][nt][0];
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

__global__ __launch_bounds__(256) void probe(const char* img, float* out, int reps, int mode) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  bf16x8 a[3][8];
#pragma unroll
  for (int p = 0; p < 3; ++p)
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
#pragma unroll
      for (int e = 0; e < 8; ++e) a[p][ks][e] = (__bf16)(0.001f * (float)((lane * 7 + ks * 3 + e + p) & 31));
  f32x4 acc[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int r = 0; r < reps; ++r) {
    const char* W = img + (size_t)(r & 7) * 3 * kPartBytes;
    if (mode == 0) product<3, 3, 6, true, true>(a, W, 64 * wv, lane, acc);
    else if (mode == 1) product<2, 2, 3, true, true>(a, W, 64 * wv, lane, acc);
    else if (mode == 2) product<3, 3, 6, true, false>(a, W, 64 * wv, lane, acc);
    else if (mode == 3) product<3, 3, 6, false, true>(a, W, 64 * wv, lane, acc);
    else if (mode == 5) product<2, 3, 5, true, true>(a, W, 64 * wv, lane, acc);
    else {
      float af[64];
#pragma unroll
      for (int s = 0; s < 64; ++s) af[s] = (float)a[s & 1][s >> 3][s & 7];
#pragma unroll
      for (int s = 0; s < 64; ++s)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s], af[(s + nt) & 63], acc[nt], 0, 0, 0);
    }
  }
  const int g = lane >> 4, c = lane & 15;
#pragma unroll
  for (int r = 0; r < 4; ++r)
    out[(size_t)(blockIdx.x * 16 + 4 * g + r) * kC + 64 * wv + 4 * c] = acc[0][r] + acc[1][r] + acc[2][r] + acc[3][r];
}

int main() {
  char* img;
  float* out;
  hipMalloc(&img, 8 * 3 * kPartBytes);
  hipMemset(img, 0x3c, 8 * 3 * kPartBytes);
  hipMalloc(&out, (size_t)256 * 16 * kC * 4);
  const char* names[] = {"bf16 3 x 3 parts, 6 terms", "bf16 2 x 2 parts, 3 terms", "weight loads of 3 parts only", "6-term matrix instructions only",
                         "f32 16x16x4, registers only", "bf16 w 2 / a 3 parts, 5 terms"};
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int grid : {64, 256})
    for (int mode = 0; mode < 6; ++mode) {
      float t[2];
      int k = 0;
      for (int reps : {1, 33}) {
        for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(probe, dim3(grid), dim3(256), 0, 0, img, out, reps, mode);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int it = 0; it < 20; ++it) hipLaunchKernelGGL(probe, dim3(grid), dim3(256), 0, 0, img, out, reps, mode);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&t[k], e0, e1);
        t[k] = t[k] / 20 * 1000;
        ++k;
      }
      printf("grid %3d  %-34s per product %6.2f us   (1 rep %6.2f us, 33 reps %7.2f us)\n", grid, names[mode], (t[1] - t[0]) / 32, t[0], t[1]);
    }
  return 0;
}
