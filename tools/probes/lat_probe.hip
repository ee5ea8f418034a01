// lat_probe.hip — per-instruction latencies that decide the shape of a serial, one-CU kernel (furthest point
// sampling): dependent VALU / DPP / permlane / readlane chains, LDS and L1/L2 load round trips, s_barrier, each with
// 4, 8 and 16 waves of ONE workgroup running the same chain (1, 2, 4 waves per SIMD).
//   hipcc -O3 --offload-arch=gfx950 -o tools/probes/bin/lat_probe tools/probes/lat_probe.hip && tools/probes/bin/lat_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ unsigned long long clk() {
  __builtin_amdgcn_sched_barrier(0);
  unsigned long long t = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}

template <int CTRL>
__device__ __forceinline__ unsigned dppz(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true); }

constexpr int kReps = 64;

// mode: 0 v_fma chain, 1 v_max_u32_dpp chain, 2 permlane32_swap+max chain, 3 ballot->ff1->readlane->valu chain,
//       4 LDS pointer chase, 5 global pointer chase (buffer given), 6 s_barrier, 7 two independent dpp chains,
//       8 empty (clock overhead), 9 16-B global load + dependent index (like a bucket fetch)
__global__ void probe(int mode, const unsigned* __restrict__ chase, unsigned nchase, unsigned long long* out, unsigned* sink) {
  __shared__ unsigned lds[4096];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = (i * 193u + 64u) & 4095u;
  __syncthreads();
  unsigned v = sink[threadIdx.x & 63] + lane;
  float f = (float)v * 1e-9f;
  unsigned long long t0 = 0, t1 = 0;
  __syncthreads();
  t0 = clk();
  if (mode == 0) {
    const float fa = 1.0000001f + f * 1e-30f, fb = 1e-7f + f * 1e-30f;
#pragma unroll 8
    for (int i = 0; i < kReps; ++i) f = __fmaf_rn(f, fa, fb);
  } else if (mode == 1) {
#pragma unroll
    for (int i = 0; i < kReps / 4; ++i) {
      v = max(v, dppz<0xB1>(v)) + 1u; v = max(v, dppz<0x4E>(v)) + 1u; v = max(v, dppz<0x141>(v)) + 1u; v = max(v, dppz<0x140>(v)) + 1u;
    }
  } else if (mode == 2) {
#pragma unroll
    for (int i = 0; i < kReps; ++i) {
      auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
      v = max(r[0], r[1]) + 1u;
    }
  } else if (mode == 3) {
#pragma unroll
    for (int i = 0; i < kReps; ++i) {
      const unsigned long long m = __ballot((v & 7u) == (unsigned)(lane & 7));
      const int l = m ? __ffsll((long long)m) - 1 : 0;
      v += (unsigned)__builtin_amdgcn_readlane((int)v, l) + 1u;
    }
  } else if (mode == 4) {
    unsigned p = (unsigned)lane;
#pragma unroll
    for (int i = 0; i < kReps; ++i) p = lds[p];
    v += p;
  } else if (mode == 5) {
    unsigned p = (unsigned)(lane * 16) % nchase;
#pragma unroll
    for (int i = 0; i < kReps; ++i) p = chase[p];
    v += p;
  } else if (mode == 6) {
#pragma unroll
    for (int i = 0; i < kReps; ++i) asm volatile("s_barrier" ::: "memory");
  } else if (mode == 7) {
    unsigned u = v ^ 0x55u;
#pragma unroll
    for (int i = 0; i < kReps / 4; ++i) {
      v = max(v, dppz<0xB1>(v)) + 1u; u = max(u, dppz<0xB1>(u)) + 1u;
      v = max(v, dppz<0x4E>(v)) + 1u; u = max(u, dppz<0x4E>(u)) + 1u;
      v = max(v, dppz<0x141>(v)) + 1u; u = max(u, dppz<0x141>(u)) + 1u;
      v = max(v, dppz<0x140>(v)) + 1u; u = max(u, dppz<0x140>(u)) + 1u;
    }
    v += u;
  } else if (mode == 10 || mode == 11) {
    // a bucket fetch: wave-uniform dependent index, 64 contiguous float4 (1 KB per wave-load); mode 11 adds the 256-B key load
    const uint4* c4 = reinterpret_cast<const uint4*>(chase);
    const unsigned nb = nchase / 4 / 64;
    unsigned g = (unsigned)(w * 7 + 1) % nb;
#pragma unroll 4
    for (int i = 0; i < kReps; ++i) {
      const uint4 q = c4[g * 64 + lane];
      unsigned k = 0;
      if (mode == 11) k = chase[g * 64 + lane];
      g = (unsigned)__builtin_amdgcn_readfirstlane((int)((q.x >> 6) + k)) % nb;
    }
    v += g;
  } else if (mode == 12) {
    // 4 independent bucket fetches in flight per wave
    const uint4* c4 = reinterpret_cast<const uint4*>(chase);
    const unsigned nb = nchase / 4 / 64;
    unsigned g = (unsigned)(w * 7 + 1) % nb;
#pragma unroll 2
    for (int i = 0; i < kReps / 4; ++i) {
      const uint4 q0 = c4[g * 64 + lane], q1 = c4[((g + 17) % nb) * 64 + lane], q2 = c4[((g + 101) % nb) * 64 + lane], q3 = c4[((g + 203) % nb) * 64 + lane];
      g = (unsigned)__builtin_amdgcn_readfirstlane((int)((q0.x + q1.x + q2.x + q3.x) >> 6)) % nb;
    }
    v += g;
  } else if (mode == 9) {
    const uint4* c4 = reinterpret_cast<const uint4*>(chase);
    unsigned p = (unsigned)(lane) % (nchase / 4);
#pragma unroll
    for (int i = 0; i < kReps; ++i) { const uint4 q = c4[p]; p = (q.x + q.w) % (nchase / 4); }
    v += p;
  }
  t1 = clk();
  sink[threadIdx.x & 63] = v + (unsigned)f;
  if (lane == 0) out[w] = t1 - t0;
}

int main() {
  unsigned long long* out;
  unsigned* sink;
  CHECK(hipMalloc(&out, 16 * sizeof(unsigned long long)));
  CHECK(hipMalloc(&sink, 64 * sizeof(unsigned)));
  CHECK(hipMemset(sink, 0, 64 * sizeof(unsigned)));
  const char* names[] = {"v_fma_f32 dependent", "v_max_u32_dpp(+add) dependent", "permlane32_swap+max+add dependent",
                         "ballot>ff1>readlane>add dependent", "LDS pointer chase (ds_read_b32)", "global pointer chase", "s_barrier",
                         "2 interleaved dpp chains (per pair)", "empty (clock overhead)", "global 16-B load + dependent index",
                         "bucket fetch 1 KB coalesced, dependent", "bucket fetch 1 KB + 256 B keys, dependent", "4 bucket fetches in flight (per fetch)"};
  // chase buffers: 8 KB (L1), 640 KB (L2), 64 MB (beyond L2: MALL), random cyclic permutation with 64-B stride granularity
  const size_t sizes[] = {2048, 160 * 1024, 16 * 1024 * 1024};
  const char* szn[] = {"8 KB", "640 KB", "64 MB"};
  unsigned* bufs[3];
  for (int b = 0; b < 3; ++b) {
    const size_t n = sizes[b];
    unsigned* h = (unsigned*)malloc(n * 4);
    // permutation over 16-word blocks
    const size_t nblk = n / 16;
    unsigned* perm = (unsigned*)malloc(nblk * 4);
    for (size_t i = 0; i < nblk; ++i) perm[i] = (unsigned)i;
    srand(1);
    for (size_t i = nblk - 1; i > 0; --i) { size_t j = (size_t)rand() % (i + 1); unsigned t = perm[i]; perm[i] = perm[j]; perm[j] = t; }
    for (size_t i = 0; i < n; ++i) h[i] = 0;
    for (size_t i = 0; i < nblk; ++i) {
      const unsigned nxt = perm[(i + 1) % nblk] * 16;
      for (int k = 0; k < 16; ++k) h[(size_t)perm[i] * 16 + k] = nxt + (unsigned)k;  // lane offset preserved within the block
    }
    CHECK(hipMalloc(&bufs[b], n * 4));
    CHECK(hipMemcpy(bufs[b], h, n * 4, hipMemcpyHostToDevice));
    free(h); free(perm);
  }
  for (int waves : {1, 4, 8, 16}) {
    printf("---- %d waves in one workgroup (%d per SIMD): cycles per step on wave 0 / last wave\n", waves, waves / 4);
    for (int mode = 0; mode < 13; ++mode) {
      for (int b = 0; b < 3; ++b) {
        if (mode != 5 && mode < 9 && b > 0) break;
        unsigned long long h[16];
        for (int rep = 0; rep < 3; ++rep) {  // the last repetition is reported (warm caches)
          hipLaunchKernelGGL(probe, dim3(1), dim3(waves * 64), 0, 0, mode, bufs[b], (unsigned)sizes[b], out, sink);
          CHECK(hipDeviceSynchronize());
        }
        CHECK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
        printf("%-40s %-7s %8.1f %8.1f\n", names[mode], (mode == 5 || mode >= 9) ? szn[b] : "", (double)h[0] / kReps, (double)h[waves - 1] / kReps);
      }
    }
  }
  return 0;
}
