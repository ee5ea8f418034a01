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


// ---- variants (round 5): a weight row's 128-byte line read by two back-to-back instructions (k = 32 m2 + 8 kg + 4 h + e), and
// the contraction split over two waves (8 waves per workgroup, wave w and w + 4 own the same 64 columns, k halves) -------------------
template <int STEPS>  // STEPS = 16-wide k steps this wave contracts (16 = all of k, 8 = half)
__device__ __forceinline__ void gemm_paired(const float* a, const float* __restrict__ W, int col0, int k0, int lane, f32x4 (&acc)[4], bool mfma) {
  const int j = lane & 15, kg = lane >> 4;
  const f32x4* wp = reinterpret_cast<const f32x4*>(W + (size_t)(col0 + 4 * j) * kRbC + k0 + 8 * kg);  // + nt * 64 (next row), + 8 m2 + h
  constexpr int D = 3;  // pairs in flight
  f32x4 b[D][4][2];
#pragma unroll
  for (int d = 0; d < D - 1; ++d)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) { b[d][nt][0] = wp[nt * 64 + 8 * d]; b[d][nt][1] = wp[nt * 64 + 8 * d + 1]; }
#pragma unroll
  for (int m2 = 0; m2 < STEPS / 2; ++m2) {
    if (m2 + D - 1 < STEPS / 2) {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        b[(m2 + D - 1) % D][nt][0] = wp[nt * 64 + 8 * (m2 + D - 1)];
        b[(m2 + D - 1) % D][nt][1] = wp[nt * 64 + 8 * (m2 + D - 1) + 1];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (mfma) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int nt = 0; nt < 4; ++nt)
            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[8 * m2 + 4 * h + e], b[m2 % D][nt][h][e], acc[nt], 0, 0, 0);
    } else {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[nt] += b[m2 % D][nt][0] + b[m2 % D][nt][1];
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}
// rot: the workgroup walks the contraction index from step `rot` on (the workgroups of an XCD would otherwise ask its L2 for the same
// line at the same moment)
template <int STEPS>
__device__ __forceinline__ void gemm_t_part(const float* a, const float* __restrict__ W, int col0, int k0, int lane, f32x4 (&acc)[4], int rot = 0) {
  const int j = lane & 15, kg = lane >> 4;
  const float* wp = W + (size_t)(k0 + 4 * kg) * kRbC + col0 + 4 * j;
#define ROTM(m) (((m) + rot) & (STEPS - 1))
  constexpr int D = 6;
  f32x4 b[D][4];
#pragma unroll
  for (int d = 0; d < D - 1; ++d)
#pragma unroll
    for (int e = 0; e < 4; ++e) b[d][e] = *reinterpret_cast<const f32x4*>(wp + (size_t)(16 * ROTM(d) + e) * kRbC);
#pragma unroll
  for (int m = 0; m < STEPS; ++m) {
    if (m + D - 1 < STEPS) {
#pragma unroll
      for (int e = 0; e < 4; ++e) b[(m + D - 1) % D][e] = *reinterpret_cast<const f32x4*>(wp + (size_t)(16 * ROTM(m + D - 1) + e) * kRbC);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * m + e], b[m % D][e][nt], acc[nt], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// mode 0: paired forward product, 4 waves; 1: its loads alone; 2: paired forward, 8 waves (k halves); 3: dX = dY W, 8 waves (k halves)
template <int THREADS>
__global__ __launch_bounds__(THREADS) void probe2_kernel(const float* x, const float* w, float* out, int reps, int mode) {
  __shared__ __attribute__((aligned(16))) float xs[kRbRows * kRbStride];
  __shared__ __attribute__((aligned(16))) float xch[4 * 64 * 8];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wv & 3, hp = wv >> 2;
  for (int e = tid; e < kRbRows * kRbC / 4; e += THREADS)
    *reinterpret_cast<f32x4*>(xs + (e >> 6) * kRbStride + 4 * (e & 63)) = reinterpret_cast<const f32x4*>(x + (size_t)blockIdx.x * kRbRows * kRbC)[e];
  __syncthreads();
  constexpr int STEPS = THREADS == 512 ? 8 : 16;
  float a[4 * STEPS];
  {
    const int i = lane & 15, kg = lane >> 4;
#pragma unroll
    for (int m = 0; m < STEPS; ++m) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(xs + i * kRbStride + 128 * hp + 16 * m + 4 * kg);
#pragma unroll
      for (int e = 0; e < 4; ++e) a[4 * m + e] = v[e];
    }
  }
  f32x4 acc[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int r = 0; r < reps; ++r) {
    const float* W = w + (size_t)(r & 7) * kRbC * kRbC;
    if (mode == 3) gemm_t_part<STEPS>(a, W, 64 * wc, 128 * hp, lane, acc);
    else if (mode == 4) gemm_t_part<STEPS>(a, W, 64 * wc, 128 * hp, lane, acc, 2 * ((blockIdx.x >> 3) & 7));
    else if (mode == 5) gemm_t_part<STEPS>(a, W, 64 * ((wc + (blockIdx.x >> 3)) & 3), 128 * hp, lane, acc, 4 * ((blockIdx.x >> 5) & 3));
    else if (mode == 6) gemm_paired<STEPS>(a, W, 64 * ((wc + (blockIdx.x >> 3)) & 3), 128 * hp, lane, acc, false);
    else gemm_paired<STEPS>(a, W, 64 * wc, 128 * hp, lane, acc, mode != 1);
    if (THREADS == 512) {  // the two k halves meet: each wave hands the other the two rows it will not finish itself
      __syncthreads();
      float* mine = xch + (wc * 64 + lane) * 8;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) { mine[2 * nt] = hp ? acc[nt][0] : acc[nt][2]; mine[2 * nt + 1] = hp ? acc[nt][1] : acc[nt][3]; }
      __syncthreads();
      // (both halves write the same slot in this probe: the real kernels use two slots; the traffic is what is timed)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) { acc[nt][0] += mine[2 * nt]; acc[nt][1] += mine[2 * nt + 1]; }
    }
  }
  const int g = lane >> 4, c = lane & 15;
  if (hp == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      *reinterpret_cast<f32x4*>(out + (size_t)(blockIdx.x * kRbRows + 4 * g + r) * kRbC + 64 * wc + 4 * c) = rb_row(acc, r);
  }
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
  for (int grid : {1, 8, 64})
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
  const char* names2[] = {"paired rows, 4 waves", "paired rows, loads only", "paired rows, 8 waves (k halves)", "dX = dY W, 8 waves (k halves)",
                          "dX = dY W, k rotated per wg", "dX = dY W, k + columns rotated", "paired loads only, cols rotated"};
  for (int mode = 0; mode < 7; ++mode)
    for (int reps : {1, 9, 33}) {
      float best = 1e9f;
      for (int it = 0; it < 5; ++it) {
        hipEventRecord(e0);
        if (mode < 2 || mode > 3) hipLaunchKernelGGL(probe2_kernel<256>, dim3(64), dim3(256), 0, 0, x, w, out, reps, mode);
        else hipLaunchKernelGGL(probe2_kernel<512>, dim3(64), dim3(512), 0, 0, x, w, out, reps, mode);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
      }
      printf("grid  64  %-32s reps %2d: %8.2f us\n", names2[mode], reps, best * 1e3f);
    }
  return 0;
}
