// lds_atomic_rate.hip — what an LDS histogram update costs on gfx950, by instruction and address pattern.
// The table-gradient kernels (csrc/attn_bwd_box*.hip) flush matrix-unit sums into an int32 LDS histogram; this probe
// measures the rate of the candidate instructions with 4 / 8 / 16 waves per CU, one workgroup per CU on every CU.
//   hipcc -O3 --offload-arch=gfx950 -o tools/probes/bin/lds_atomic_rate tools/probes/lds_atomic_rate.hip
//   tools/probes/bin/lds_atomic_rate
// Output: wave-instructions per microsecond per CU and shader cycles per wave-instruction (s_memtime inside the kernel).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int kWords = 32768;  // 128 KB of bins

// address patterns (word index of lane l in iteration it)
//  0 consecutive words (conflict-free)            1 random words                     2 the box2 flush pattern
//  3 stride 2 words (2-way on b32)                4 all lanes one word               5 consecutive 8-byte slots (for u64)
//  6 random 8-byte slots
__device__ __forceinline__ unsigned addr_of(int pat, int lane, int it, unsigned& rnd) {
  rnd = rnd * 1664525u + 1013904223u;
  switch (pat) {
    case 0: return (unsigned)(lane + it * 64) & (kWords - 1);
    case 1: return (rnd >> 9) & (kWords - 1);
    case 2: {  // lane = (kk, c15): c15 = (xi, cx, h), kk = (gl, cz); cell from rnd (wave-uniform per gl in the kernel; here per kk>>1)
      const int kk = lane >> 4, c15 = lane & 15, xi = c15 >> 3, cx = (c15 >> 2) & 1, h = c15 & 3, cz = kk & 1;
      const unsigned cell = (__builtin_amdgcn_readfirstlane(rnd) >> (9 + (kk >> 1))) % 800u;
      return ((xi * 3 * 1000 + cell + cz * 100 + cx) * 4 + h + (it & 1) * 40) & (kWords - 1);
    }
    case 3: return (unsigned)(2 * lane + it * 128) & (kWords - 1);
    case 4: return (unsigned)(it * 7) & (kWords - 1);
    case 5: return (unsigned)(2 * lane + it * 128) & (kWords - 2);
    default: return ((rnd >> 9) & (kWords - 1)) & ~1u;
  }
}

// op: 0 ds_add_u32 (no return)   1 ds_add_rtn_u32 (value used)   2 ds_add_u64 (no return)   3 ds_write_b32   4 ds_add_f32
//     5 ds_read_b32 + v_add + ds_write_b32 (not atomic; for scale)
template <int op, int pat>
__global__ void probe(int iters, unsigned long long* cycles, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned tab[];
  for (int i = threadIdx.x; i < kWords; i += blockDim.x) tab[i] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  unsigned rnd = threadIdx.x * 2654435761u + blockIdx.x;
  unsigned acc = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    const unsigned a = addr_of(pat, lane, it, rnd);
    if (op == 0) __hip_atomic_fetch_add(&tab[a], 1u + (unsigned)lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else if (op == 1) acc += __hip_atomic_fetch_add(&tab[a], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else if (op == 2) __hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(__builtin_assume_aligned(&tab[a & ~1u], 8)), 0x100000001ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else if (op == 3) tab[a] = rnd;
    else if (op == 4) __hip_atomic_fetch_add(reinterpret_cast<float*>(&tab[a]), 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else { tab[a] += 1u; }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  __syncthreads();
  for (int i = threadIdx.x; i < kWords; i += blockDim.x) acc += tab[i];
  if (acc == 0xFFFFFFFFu) sink[0] = acc;
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int op, int pat>
static void launch1(int cus, int waves, int iters, unsigned long long* cyc, unsigned* sink) {
  static bool once = false;
  if (!once) { CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe<op, pat>), hipFuncAttributeMaxDynamicSharedMemorySize, kWords * 4)); once = true; }
  hipLaunchKernelGGL((probe<op, pat>), dim3(cus), dim3(waves * 64), kWords * 4, 0, iters, cyc, sink);
}
template <int op>
static void launch_pat(int pat, int cus, int waves, int iters, unsigned long long* cyc, unsigned* sink) {
  switch (pat) {
    case 0: launch1<op, 0>(cus, waves, iters, cyc, sink); break;
    case 1: launch1<op, 1>(cus, waves, iters, cyc, sink); break;
    case 2: launch1<op, 2>(cus, waves, iters, cyc, sink); break;
    case 3: launch1<op, 3>(cus, waves, iters, cyc, sink); break;
    case 4: launch1<op, 4>(cus, waves, iters, cyc, sink); break;
    case 5: launch1<op, 5>(cus, waves, iters, cyc, sink); break;
    default: launch1<op, 6>(cus, waves, iters, cyc, sink); break;
  }
}
static void launch(int op, int pat, int cus, int waves, int iters, unsigned long long* cyc, unsigned* sink) {
  switch (op) {
    case 0: launch_pat<0>(pat, cus, waves, iters, cyc, sink); break;
    case 1: launch_pat<1>(pat, cus, waves, iters, cyc, sink); break;
    case 2: launch_pat<2>(pat, cus, waves, iters, cyc, sink); break;
    case 3: launch_pat<3>(pat, cus, waves, iters, cyc, sink); break;
    case 4: launch_pat<4>(pat, cus, waves, iters, cyc, sink); break;
    default: launch_pat<5>(pat, cus, waves, iters, cyc, sink); break;
  }
}

int main() {
  int dev = 0;
  CHECK(hipSetDevice(dev));
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, dev));
  const int cus = prop.multiProcessorCount;
  unsigned long long* cyc;
  unsigned* sink;
  CHECK(hipMalloc(&cyc, cus * sizeof(unsigned long long)));
  CHECK(hipMalloc(&sink, 64));
  const char* ops[] = {"ds_add_u32", "ds_add_rtn_u32", "ds_add_u64", "ds_write_b32", "ds_add_f32", "read-add-write"};
  const char* pats[] = {"consecutive", "random", "box2-flush", "stride-2", "one-word", "consecutive-8B", "random-8B"};
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const int iters = 4096;
  printf("%d CUs, %d MHz; %d iterations per lane\n", cus, prop.clockRate / 1000, iters);
  for (int waves : {4, 8, 16}) {
    for (int op = 0; op < 6; ++op) {
      for (int pat = 0; pat < 7; ++pat) {
        if ((op == 2) != (pat >= 5)) continue;
        if (op >= 3 && pat > 1) continue;
        launch(op, pat, cus, waves, iters, cyc, sink);  // warm
        CHECK(hipEventRecord(e0));
        launch(op, pat, cus, waves, iters, cyc, sink);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long h[512];
        CHECK(hipMemcpy(h, cyc, cus * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        double avg = 0;
        for (int i = 0; i < cus; ++i) avg += (double)h[i];
        avg /= cus;
        const double winst = (double)waves * iters;  // wave-instructions per CU
        printf("waves/CU %2d  %-15s %-15s  %8.1f us  %6.2f wave-inst/us/CU  %6.1f memtime ticks per wave-inst (CU)  %5.2f lanes/tick\n", waves, ops[op],
               pats[pat], ms * 1e3, winst / (ms * 1e3), avg / winst, 64.0 * winst / avg);
      }
    }
  }
  return 0;
}
