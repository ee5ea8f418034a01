// Issue rate of v_mfma_f32_16x16x4_f32 / v_mfma_f32_32x32x2_f32 under the shapes the sparse-convolution kernels use.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/mfma_f32_rate.hip -o /tmp/mfma_f32_rate && /tmp/mfma_f32_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, int MODE>
__global__ __launch_bounds__(256) void k16(float* out, int iters, float x) {
  f32x4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a[4] = {x, x + 1, x + 2, x + 3}, b[4] = {x * 2, x * 3, x * 4, x * 5};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 64 / NACC; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) {
        // MODE 0: A shared by 4 consecutive MFMAs (the conv kernels' order), MODE 1: distinct A/B register per MFMA
        const float av = MODE == 0 ? a[(i / 4) & 3] : a[i & 3];
        const float bv = MODE == 0 ? b[i & 3] : b[(i + r) & 3];
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[i], 0, 0, 0);
      }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k32(float* out, int iters, float x) {
  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  float a[2] = {x, x + 1}, b[2] = {x * 2, x * 3};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i >> 1], b[i & 1], acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) s += acc[i][j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// random operands from memory, short loops, many workgroups: the shape of one sparse-convolution launch
template <int NACC, int SH = 0, int NREG = 16>
__global__ __launch_bounds__(256) void k16_data(const float* __restrict__ in, float* out, int iters) {
  f32x4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a[16], b[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    a[i] = in[(threadIdx.x * 16 + i) & 65535];
    b[i] = in[(threadIdx.x * 16 + i + 4096) & 65535];
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int t = 0; t < 4; ++t)
          acc[4 * i + t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(4 * i + s) & (NREG - 1)], b[(4 * t + ((s + SH) & 3)) & (NREG - 1)],
                                                                acc[4 * i + t], 0, 0, 0);
  }
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) out[((size_t)blockIdx.x * 64 + i * 4 + r) * 256 + threadIdx.x] = acc[i][r];
}

template <typename K>
static void run_data(const char* name, K kern, int grid, int iters, float scale) {
  float *in, *out;
  hipMalloc(&in, 65536 * 4);
  hipMalloc(&out, (size_t)grid * 64 * 256 * 4);
  float* h = (float*)malloc(65536 * 4);
  unsigned st = 12345;
  for (int i = 0; i < 65536; ++i) {
    st = st * 1664525u + 1013904223u;
    h[i] = scale * ((int)(st >> 8) - (1 << 23)) / (float)(1 << 23);
    if (scale == 0.0f) h[i] = 0.0f;    // +0.0 everywhere
    if (scale == 7.0f) h[i] = 1.0f;    // one constant everywhere
    if (scale == 8.0f) h[i] = (float)((st >> 20) & 7);  // small integers: few mantissa bits
  }
  hipMemcpy(in, h, 65536 * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0);
    for (int j = 0; j < 10; ++j) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, in, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  const double flops = 64.0 * 2048.0 * iters * 4.0 * grid * 10;
  printf("%-44s grid=%d iters=%d  %8.3f ms/launch  %7.1f TFLOP/s\n", name, grid, iters, ms / 10, flops / ms / 1e9);
  hipFree(in);
  hipFree(out);
  free(h);
}

template <typename K>
static void run(const char* name, K kern, int wgs_per_cu, double flop_per_wave_iter, int lds_bytes) {
  float* out;
  const int grid = 256 * wgs_per_cu, iters = 2000;
  hipMalloc(&out, (size_t)grid * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds_bytes, 0, out, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
  }
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = flop_per_wave_iter * iters * 4.0 * grid;
  printf("%-44s WG/CU=%d  %8.3f ms  %7.1f TFLOP/s\n", name, wgs_per_cu, ms, flops / ms / 1e9);
  hipFree(out);
}

int main() {
  const double f16 = 64.0 * 2048.0, f32 = 32.0 * 4096.0;
  for (int w = 1; w <= 4; ++w) {
    run("16x16x4, 16 accumulators, shared A", k16<16, 0>, w, f16, 0);
    run("16x16x4, 16 accumulators, distinct A/B", k16<16, 1>, w, f16, 0);
    run("16x16x4,  4 accumulators", k16<4, 1>, w, f16, 0);
    run("32x32x2,  4 accumulators", k32, w, f32, 0);
  }
  // one workgroup per CU is forced by a large dynamic LDS request; more are spread by the dispatcher
  run("16x16x4, 16 acc, 1 WG/CU by LDS (grid 256)", k16<16, 0>, 1, f16, 100 * 1024);
  run_data("16+16 regs, A/B same bank, random", k16_data<16, 0, 16>, 768, 2000, 1.0f);
  run_data("16+16 regs, A/B same bank, zeros", k16_data<16, 0, 16>, 768, 2000, 0.0f);
  run_data("16+16 regs, B bank +1", k16_data<16, 1, 16>, 768, 2000, 1.0f);
  run_data("16+16 regs, B bank +2", k16_data<16, 2, 16>, 768, 2000, 1.0f);
  run_data("16+16 regs, all 1.0", k16_data<16, 0, 16>, 768, 2000, 7.0f);
  run_data("16+16 regs, small integers", k16_data<16, 0, 16>, 768, 2000, 8.0f);
  run_data("16+16 regs, random, 1 WG/CU", k16_data<16, 0, 16>, 256, 2000, 1.0f);
  run_data("16+16 regs, random, 2 WG/CU", k16_data<16, 0, 16>, 512, 2000, 1.0f);
  run_data("16+16 regs, random, 4 WG/CU", k16_data<16, 0, 16>, 1024, 2000, 1.0f);
  run_data("4+4 regs, zeros", k16_data<16, 0, 4>, 768, 2000, 0.0f);
  run_data("8+8 regs, same bank", k16_data<16, 0, 8>, 768, 2000, 1.0f);
  run_data("4+4 regs, same bank", k16_data<16, 0, 4>, 768, 2000, 1.0f);
  run_data("4+4 regs, B bank +1", k16_data<16, 1, 4>, 768, 2000, 1.0f);
  run_data("conv-launch shape (1404 x 16 iters)", k16_data<16, 0, 16>, 1404, 16, 1.0f);
  return 0;
}
