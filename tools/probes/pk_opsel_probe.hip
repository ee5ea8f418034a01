// pk_opsel_probe.hip — does packed fp32 math with a broadcast source compute what its encoding says, on this GPU, under
// load?  DESIGN.md 4.4b records a kernel that went wrong with `v_pk_mul_f32 ... op_sel:[0,1]` (source 1 read from the
// HIGH register for both result lanes) and right with scalar multiplies; whether the hardware or the surrounding
// code generation was at fault was left open.  Every op_sel / op_sel_hi combination of v_pk_mul_f32, v_pk_add_f32 and
// v_pk_fma_f32 is issued here from inline asm (so the encoding is exactly what is written), by 16-wave workgroups on
// every CU, with LDS traffic and transcendental ops around it, and compared bit for bit with scalar arithmetic.
//   hipcc -O3 --offload-arch=gfx950 -o tools/probes/bin/pk_opsel_probe tools/probes/pk_opsel_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float float2v __attribute__((ext_vector_type(2)));

#define PK2(op, sel, selhi, d, a, b) \
  asm volatile(op " %0, %1, %2 op_sel:[" sel "] op_sel_hi:[" selhi "]" : "=v"(d) : "v"(a), "v"(b))
#define PK3(op, sel, selhi, d, a, b, c) \
  asm volatile(op " %0, %1, %2, %3 op_sel:[" sel ",0] op_sel_hi:[" selhi ",1]" : "=v"(d) : "v"(a), "v"(b), "v"(c))

__device__ __forceinline__ float pick(float2v v, int hi) { return hi ? v.y : v.x; }

// results: [form 0..15][mul, add, fma] mismatch counters
__global__ __launch_bounds__(1024) void probe(const float* __restrict__ in, unsigned* __restrict__ bad, int iters) {
  __shared__ float lds[4096];
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  float2v a = {in[(t * 4) & 0xFFFF], in[(t * 4 + 1) & 0xFFFF]};
  float2v b = {in[(t * 4 + 2) & 0xFFFF], in[(t * 4 + 3) & 0xFFFF]};
  float2v c = {a.y * 0.5f, b.x * 0.25f};
  for (int it = 0; it < iters; ++it) {
    lds[(threadIdx.x * 7 + it) & 4095] = a.x;  // LDS traffic + a transcendental between the packed ops
    a.x = __expf(-fabsf(a.x) * 1e-3f) + a.y;
    __syncthreads();
    b.y += lds[(threadIdx.x * 13 + it) & 4095] * 1e-3f;
#define FORM(idx, s0, s1, h0, h1)                                                                                   \
    {                                                                                                               \
      float2v d;                                                                                                    \
      PK2("v_pk_mul_f32", #s0 "," #s1, #h0 "," #h1, d, a, b);                                                       \
      const float e0 = pick(a, s0) * pick(b, s1), e1 = pick(a, h0) * pick(b, h1);                                   \
      if (__float_as_uint(d.x) != __float_as_uint(e0) || __float_as_uint(d.y) != __float_as_uint(e1)) atomicAdd(&bad[idx * 3 + 0], 1u); \
      PK2("v_pk_add_f32", #s0 "," #s1, #h0 "," #h1, d, a, b);                                                       \
      const float f0 = pick(a, s0) + pick(b, s1), f1 = pick(a, h0) + pick(b, h1);                                   \
      if (__float_as_uint(d.x) != __float_as_uint(f0) || __float_as_uint(d.y) != __float_as_uint(f1)) atomicAdd(&bad[idx * 3 + 1], 1u); \
      PK3("v_pk_fma_f32", #s0 "," #s1, #h0 "," #h1, d, a, b, c);                                                    \
      const float g0 = __fmaf_rn(pick(a, s0), pick(b, s1), c.x), g1 = __fmaf_rn(pick(a, h0), pick(b, h1), c.y);     \
      if (__float_as_uint(d.x) != __float_as_uint(g0) || __float_as_uint(d.y) != __float_as_uint(g1)) atomicAdd(&bad[idx * 3 + 2], 1u); \
    }
    FORM(0, 0, 0, 0, 0) FORM(1, 0, 0, 0, 1) FORM(2, 0, 0, 1, 0) FORM(3, 0, 0, 1, 1)
    FORM(4, 0, 1, 0, 0) FORM(5, 0, 1, 0, 1) FORM(6, 0, 1, 1, 0) FORM(7, 0, 1, 1, 1)
    FORM(8, 1, 0, 0, 0) FORM(9, 1, 0, 0, 1) FORM(10, 1, 0, 1, 0) FORM(11, 1, 0, 1, 1)
    FORM(12, 1, 1, 0, 0) FORM(13, 1, 1, 0, 1) FORM(14, 1, 1, 1, 0) FORM(15, 1, 1, 1, 1)
    a.y = a.y * 0.999f + b.x * 1e-3f;
    b.x = b.x * 1.001f - a.x * 1e-3f;
  }
  if (a.x == 123.456f) bad[63] = 1;  // keep the chain alive
}

int main() {
  const int n = 1 << 16;
  float* h = (float*)malloc(n * sizeof(float));
  srand(7);
  for (int i = 0; i < n; ++i) h[i] = ((float)rand() / RAND_MAX - 0.5f) * 8.f;
  float* d;
  unsigned* bad;
  CHECK(hipMalloc(&d, n * sizeof(float)));
  CHECK(hipMalloc(&bad, 64 * sizeof(unsigned)));
  CHECK(hipMemcpy(d, h, n * sizeof(float), hipMemcpyHostToDevice));
  for (int waves : {8, 16}) {
    CHECK(hipMemset(bad, 0, 64 * sizeof(unsigned)));
    hipLaunchKernelGGL(probe, dim3(512), dim3(waves * 64), 0, 0, d, bad, 200);
    CHECK(hipDeviceSynchronize());
    unsigned hb[64];
    CHECK(hipMemcpy(hb, bad, sizeof(hb), hipMemcpyDeviceToHost));
    unsigned long total = 0;
    printf("%d-wave workgroups x 512, 200 iterations: mismatches per form [op_sel lo=(s0,s1) hi=(h0,h1)] mul/add/fma\n", waves);
    for (int f = 0; f < 16; ++f) {
      printf("  op_sel:[%d,%d] op_sel_hi:[%d,%d]  %u %u %u%s\n", (f >> 3) & 1, (f >> 2) & 1, (f >> 1) & 1, f & 1, hb[f * 3], hb[f * 3 + 1],
             hb[f * 3 + 2], ((f >> 2) & 1) && (f & 1) ? "   <- source 1 high-broadcast (the form of DESIGN.md 4.4b)" : "");
      total += hb[f * 3] + hb[f * 3 + 1] + hb[f * 3 + 2];
    }
    printf("  total mismatches: %lu\n", total);
  }
  return 0;
}
