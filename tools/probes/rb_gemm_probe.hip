// rb_gemm_probe.hip — what bounds rowblock.hip's 16 x 256 x 256 product?  64 workgroups of 4 waves (as the launches),
// `reps` products each on warm weights, in four modes:
//   0: rb_gemm (y = x W^T)    1: rb_gemm_t (dx = dy W)    2: the 256 matrix instructions alone (operands in registers)
//   3: the weight loads of rb_gemm alone (no matrix instructions)
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include tools/probes/rb_gemm_probe.hip -o /tmp/rbp && /tmp/rbp
#include "../../v-detr_amd/csrc/rowblock.hip"
#include "../../v-detr_amd/csrc/core.hip"

using namespace vdetr;

__global__ __launch_bounds__(kRbThreads) void probe_kernel(const float* x, const float* w, float* out, int reps, int mode) {
  __shared__ __attribute__((aligned(16))) float xs[kRbRows * kRbStride];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  rb_stage_rows(x, blockIdx.x * kRbRows, 4096, 1, false, xs, tid);
  __syncthreads();
  float a[64];
  rb_load_a(xs, lane, a);
  f32x4 acc[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int r = 0; r < reps; ++r) {
    const float* W = w + (size_t)(r & 7) * kRbC * kRbC;
    if (mode == 0) rb_gemm(a, W, 64 * wv, lane, acc);
    else if (mode == 1) rb_gemm_t(a, W, 64 * wv, lane, acc);
    else if (mode == 2) {
#pragma unroll
      for (int s = 0; s < 64; ++s)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], a[(s + nt) & 63], acc[nt], 0, 0, 0);
    } else {
      const int j = lane & 15, kg = lane >> 4;
      const f32x4* wp = reinterpret_cast<const f32x4*>(W + (size_t)(64 * wv + 4 * j) * kRbC + 4 * kg);
      f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int m = 0; m < 16; ++m)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) t += wp[nt * 64 + 4 * m];
      acc[0] += t;
    }
  }
  const int g = lane >> 4, c = lane & 15;
#pragma unroll
  for (int r = 0; r < 4; ++r)
    *reinterpret_cast<f32x4*>(out + (size_t)(blockIdx.x * kRbRows + 4 * g + r) * kRbC + 64 * wv + 4 * c) = rb_row(acc, r);
}

int main() {
  float *x, *w, *out;
  hipMalloc(&x, 4096 * 256 * 4);
  hipMalloc(&w, 8 * 256 * 256 * 4);
  hipMalloc(&out, 4096 * 256 * 4);
  hipMemset(x, 0, 4096 * 256 * 4);
  hipMemset(w, 0, 8 * 256 * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const char* names[] = {"rb_gemm", "rb_gemm_t", "256 mfma, registers only", "weight loads only"};
  for (int grid : {64, 256})
    for (int mode = 0; mode < 4; ++mode)
      for (int reps : {1, 9, 33}) {
        float best = 1e9f;
        for (int it = 0; it < 5; ++it) {
          hipEventRecord(e0);
          hipLaunchKernelGGL(probe_kernel, dim3(grid), dim3(kRbThreads), 0, 0, x, w, out, reps, mode);
          hipEventRecord(e1);
          hipEventSynchronize(e1);
          float ms;
          hipEventElapsedTime(&ms, e0, e1);
          best = ms < best ? ms : best;
        }
        printf("grid %3d  %-26s reps %2d: %8.2f us\n", grid, names[mode], reps, best * 1e3f);
      }
  return 0;
}
