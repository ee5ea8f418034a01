// optim.hip — gradient-norm clipping + AdamW on the flat parameter buffer as ONE launch (two with the norm's partial sums).
//
// Reference: engine.py:105-107 (`clip_grad_norm_(model.parameters(), args.clip_gradient)` then `optimizer.step()`), optimizer.py:6-26
// (torch.optim.AdamW).  The step ended with: a norm over the 47 MB flat gradient (two reduction launches), three scalar launches
// for the clip coefficient, and torch's multi-tensor AdamW with the coefficient as `grad_scale` (93 us for 331 MB: it also writes the
// scaled gradient back) — 127 us behind the last gradient.  Here:
//   * the sum of squares comes out of the launch that writes the flat gradient anyway (pack.hip: one partial per workgroup), or out
//     of `sumsq_kernel` where the flat gradient was all-reduced after the pack (N > 1);
//   * `adamw_clip_kernel`: every workgroup adds the partials up in the same fixed order (a few thousand floats out of L2; in double),
//     forms the clip coefficient max((norm + 1e-6) / max_norm, 1) and applies torch's AdamW update to its slice: 16 bytes read, 12
//     written per element, nothing else.  The step count lives on the device (a captured graph replays the launch): every workgroup
//     reads it, the LAST one to finish writes it back incremented (a ticket counter that it also resets).
// Arithmetic as torch/aten/src/ATen/native/cuda/fused_adam_utils.cuh (ADAMW mode, amsgrad off, maximize off): bias corrections and
// step size from double lr / betas rounded once to float, `p -= lr wd p`, `m = m + (1 - b1)(g - m)`, `v = b2 v + (1 - b2) g g`,
// `p -= step_size m / (sqrt(v) / sqrt(bias2) + eps)`.
#include "common.h"

namespace vdetr {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kOptThreads = 256;

// fixed-order sum of a workgroup's 256 per-thread values (double): lanes by xor-shuffle, then the four waves through LDS
__device__ __forceinline__ double opt_block_sum(double s, double* red) {
  for (int m = 32; m >= 1; m >>= 1) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, s);
    const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)u, m, 64), hi = (unsigned)__shfl_xor((int)(unsigned)(u >> 32), m, 64);
    s += __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
  }
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// partial[b] = sum of squares of workgroup b's slice of g (N > 1: the flat gradient after its all-reduce)
__global__ __launch_bounds__(kOptThreads) void sumsq_kernel(const float* __restrict__ g, long n, float* __restrict__ partial) {
  __shared__ double red[4];
  const long n4 = n >> 2;
  const f32x4* g4 = reinterpret_cast<const f32x4*>(g);
  float a0 = 0.f, a1 = 0.f;
  long i = (long)blockIdx.x * kOptThreads + threadIdx.x;
  const long stride = (long)gridDim.x * kOptThreads;
  for (; i + stride < n4; i += 2 * stride) {
    const f32x4 x = g4[i], y = g4[i + stride];
    a0 += (x[0] * x[0] + x[1] * x[1]) + (x[2] * x[2] + x[3] * x[3]);
    a1 += (y[0] * y[0] + y[1] * y[1]) + (y[2] * y[2] + y[3] * y[3]);
  }
  if (i < n4) {
    const f32x4 x = g4[i];
    a0 += (x[0] * x[0] + x[1] * x[1]) + (x[2] * x[2] + x[3] * x[3]);
  }
  if (blockIdx.x == 0 && threadIdx.x < (int)(n & 3)) {
    const float x = g[(n4 << 2) + threadIdx.x];
    a1 += x * x;
  }
  const double s = opt_block_sum((double)a0 + (double)a1, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = (float)s;
}

struct AdamArgs {
  float* p; const float* g; float* m; float* v;
  long n;
  float* step;          // device: steps taken so far (float, as torch's capturable optimizers keep it)
  unsigned* ticket;     // device: zero before the launch, zero after it
  const float* partial; // sums of squares of the gradient's pieces (nullptr: no clipping)
  int npartial;
  float max_norm, norm_eps;
  float* norm_out;      // optional: the gradient's norm (what clip_grad_norm_ returns)
  double lr, beta1, beta2, eps, weight_decay;
};

__global__ __launch_bounds__(kOptThreads) void adamw_clip_kernel(AdamArgs A) {
  __shared__ double red[4];
  const float step = *A.step + 1.f;
  float scale = 1.f;  // 1 / clip coefficient (FlatParams.clip_scale)
  if (A.partial) {
    double s = 0.0;
    for (int i = threadIdx.x; i < A.npartial; i += kOptThreads) s += (double)A.partial[i];
    const float norm = (float)sqrt(opt_block_sum(s, red));
    scale = fmaxf((norm + A.norm_eps) / A.max_norm, 1.f);
    if (A.norm_out && blockIdx.x == 0 && threadIdx.x == 0) *A.norm_out = norm;
  }
  const float bias1 = (float)(1.0 - pow(A.beta1, (double)step));
  const float bias2 = (float)(1.0 - pow(A.beta2, (double)step));
  const float step_size = (float)(A.lr / (double)bias1);
  const float bias2_sqrt = sqrtf(bias2);
  const float decay = (float)(A.lr * A.weight_decay);
  const float w1 = (float)(1.0 - A.beta1), b2 = (float)A.beta2, w2 = (float)(1.0 - A.beta2), eps = (float)A.eps;
  const bool clip = A.partial != nullptr;
  const long n4 = A.n >> 2;
  f32x4* p4 = reinterpret_cast<f32x4*>(A.p);
  f32x4* m4 = reinterpret_cast<f32x4*>(A.m);
  f32x4* v4 = reinterpret_cast<f32x4*>(A.v);
  const f32x4* g4 = reinterpret_cast<const f32x4*>(A.g);
  const long stride = (long)gridDim.x * kOptThreads;
  auto update = [&](float& p, float g, float& m, float& v) {
    if (clip) g = g / scale;
    p -= decay * p;
    m = fmaf(w1, g - m, m);
    v = b2 * v + w2 * g * g;
    const float denom = sqrtf(v) / bias2_sqrt + eps;
    p -= step_size * m / denom;
  };
  for (long i = (long)blockIdx.x * kOptThreads + threadIdx.x; i < n4; i += stride) {
    f32x4 p = p4[i], m = m4[i], v = v4[i];
    const f32x4 g = g4[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float pe = p[e], me = m[e], ve = v[e];
      update(pe, g[e], me, ve);
      p[e] = pe; m[e] = me; v[e] = ve;
    }
    p4[i] = p; m4[i] = m; v4[i] = v;
  }
  if (blockIdx.x == 0 && threadIdx.x < (int)(A.n & 3)) {
    const long i = (n4 << 2) + threadIdx.x;
    float p = A.p[i], m = A.m[i], v = A.v[i];
    update(p, A.g[i], m, v);
    A.p[i] = p; A.m[i] = m; A.v[i] = v;
  }
  // every workgroup has read the step count by now only if it has STARTED: the last ticket is drawn after all the others were, i.e.
  // after every other workgroup has finished
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned t = atomicAdd(A.ticket, 1u);
    if (t == gridDim.x - 1) {
      *A.step = step;
      atomicExch(A.ticket, 0u);
    }
  }
}

}  // namespace vdetr

using namespace vdetr;

extern "C" int vdetr_sumsq_blocks(long n) {
  const long want = (n / 4 + kOptThreads * 8 - 1) / (kOptThreads * 8);
  return (int)(want < 1 ? 1 : (want > 2048 ? 2048 : want));
}

extern "C" int vdetr_sumsq_f32(const float* g, long n, float* partial, int npartial, vdetr_stream_t stream) {
  VDETR_REQUIRE(g && partial && n > 0, "sumsq: null pointer or empty buffer");
  VDETR_REQUIRE(npartial == vdetr_sumsq_blocks(n), "sumsq: %d partials, vdetr_sumsq_blocks(n) = %d", npartial, vdetr_sumsq_blocks(n));
  VDETR_REQUIRE(((uintptr_t)g & 15) == 0, "sumsq: the buffer must be 16-B aligned");
  hipLaunchKernelGGL(sumsq_kernel, dim3(npartial), dim3(kOptThreads), 0, (hipStream_t)stream, g, n, partial);
  return check_launch("sumsq");
}

extern "C" int vdetr_adamw_clip_f32(const vdetr_adamw_desc* d, vdetr_stream_t stream) {
  VDETR_REQUIRE(d != nullptr, "adamw_clip: null descriptor");
  VDETR_REQUIRE(d->param && d->grad && d->exp_avg && d->exp_avg_sq && d->step && d->ticket && d->n > 0, "adamw_clip: null pointer or empty buffer");
  VDETR_REQUIRE((((uintptr_t)d->param | (uintptr_t)d->grad | (uintptr_t)d->exp_avg | (uintptr_t)d->exp_avg_sq) & 15) == 0,
                "adamw_clip: the four buffers must be 16-B aligned");
  VDETR_REQUIRE(d->lr >= 0.0 && d->beta1 >= 0.0 && d->beta1 < 1.0 && d->beta2 >= 0.0 && d->beta2 < 1.0 && d->eps >= 0.0 && d->weight_decay >= 0.0,
                "adamw_clip: lr %g betas (%g, %g) eps %g weight_decay %g", d->lr, d->beta1, d->beta2, d->eps, d->weight_decay);
  VDETR_REQUIRE(!d->sumsq || (d->nsumsq > 0 && d->max_norm > 0.f), "adamw_clip: %d partial sums, max_norm %g", d->nsumsq, d->max_norm);
  AdamArgs A;
  A.p = d->param; A.g = d->grad; A.m = d->exp_avg; A.v = d->exp_avg_sq;
  A.n = (long)d->n;
  A.step = d->step; A.ticket = d->ticket;
  A.partial = d->sumsq; A.npartial = d->nsumsq;
  A.max_norm = d->max_norm; A.norm_eps = d->norm_eps;
  A.norm_out = d->norm_out;
  A.lr = d->lr; A.beta1 = d->beta1; A.beta2 = d->beta2; A.eps = d->eps; A.weight_decay = d->weight_decay;
  const long n4 = A.n / 4;
  long blocks = (n4 + kOptThreads * 4 - 1) / (kOptThreads * 4);  // ~4 float4 per thread
  const long cap = (long)device_cu_count() * 16;
  blocks = blocks < 1 ? 1 : (blocks > cap ? cap : blocks);
  hipLaunchKernelGGL(adamw_clip_kernel, dim3((unsigned)blocks), dim3(kOptThreads), 0, (hipStream_t)stream, A);
  return check_launch("adamw_clip");
}
