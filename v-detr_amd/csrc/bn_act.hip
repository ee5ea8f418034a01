// bn_act.hip — y = dropout(relu(BatchNorm1d(x))) over a [B, C, N] tensor, forward and backward, ONE launch each.
//
// Reference: the hidden blocks of GenericMLP (models/helpers.py:74-141, `Conv1d -> BatchNorm1d -> ReLU -> Dropout`) as
// used by the five box heads of every decoder stage (models/vdetr_transformer.py:193-242) and by
// PositionEmbeddingLearned (helpers.py:17-33, without dropout).  ATen / MIOpen spend three launches forward (batch-norm,
// clamp, dropout) and three backward on tensors of ~5 MB; the step pays ~4.5 us per launch.
// A channel's statistics only involve its own B*N elements, so one wave owns one channel: three sweeps over the row
// (mean, variance, normalise) forward and two backward (dgamma / dbeta, then dx) — the row is 4-16 KB and stays in L2 —
// with no cross-workgroup reduction, no atomics, no scratch.  Training mode updates running_mean / running_var in place
// (momentum, unbiased variance) as nn.BatchNorm1d does; eval mode normalises with the running statistics.
// The dropout mask is the counter-based hash of attn_common.h keyed by (channel, element): backward regenerates it.
#include "attn_common.h"
#include "bn_common.h"

namespace vdetr {

constexpr int kBnThreads = 256;  // 4 channels per workgroup

// bookkeeping nn.BatchNorm1d does next to the normalisation: num_batches_tracked += 1 (one counter per module of a
// channel group), and the running mean of conv(x) + bias when the convolution's bias was left out of x (a bias in front
// of a batch-statistics BatchNorm cancels exactly: only the running mean sees it)
__device__ __forceinline__ void bn_bookkeeping(const vdetr_bnact_desc& d) {
  if (blockIdx.x == 0 && threadIdx.x < (unsigned)d.ncounters && d.training) d.counters[threadIdx.x][0] += 1;
}

__global__ __launch_bounds__(kBnThreads) void bn_act_fwd_kernel(vdetr_bnact_desc d) {
  bn_bookkeeping(d);
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= d.C) return;
  const int N = d.N, n = d.B * N;
  const size_t bstride = (size_t)d.C * N;
  const float* xc = d.x + (size_t)c * N;
  float mean, invstd;
  if (d.training && d.stats_given) {  // statistics of the batch over all ranks, from the caller
    mean = d.save_mean[c];
    invstd = d.save_invstd[c];
  } else if (d.training) {
    float s = 0.f;
    for (int b = 0; b < d.B; ++b)
      for (int i = lane; i < N; i += 64) s += xc[b * bstride + i];
    mean = wave_allsum_f32(s) / (float)n;
    float q = 0.f;
    for (int b = 0; b < d.B; ++b)
      for (int i = lane; i < N; i += 64) { const float t = xc[b * bstride + i] - mean; q += t * t; }
    const float var = wave_allsum_f32(q) / (float)n;  // biased: what the batch is normalised with
    invstd = rsqrtf(var + d.eps);
    if (lane == 0) {
      d.save_mean[c] = mean;
      d.save_invstd[c] = invstd;
      if (d.running_mean) {
        const float m = d.momentum;
        d.running_mean[c] = (1.f - m) * d.running_mean[c] + m * (mean + (d.pre_bias ? d.pre_bias[c] : 0.f));
        d.running_var[c] = (1.f - m) * d.running_var[c] + m * var * ((float)n / (float)(n > 1 ? n - 1 : 1));
      }
    }
  } else {
    mean = d.running_mean[c] - (d.pre_bias ? d.pre_bias[c] : 0.f);
    invstd = rsqrtf(d.running_var[c] + d.eps);
  }
  const float ga = d.gamma ? d.gamma[c] : 1.f, be = d.beta ? d.beta[c] : 0.f;
  const float a = ga * invstd, sh = be - mean * a;
  const BnRng rg = bn_rng(d);
  const unsigned ck = bn_chankey(rg, c);
  float* yc = d.y + (size_t)c * N;
  for (int b = 0; b < d.B; ++b)
    for (int i = lane; i < N; i += 64) {
      float v = xc[b * bstride + i] * a + sh;
      if (d.relu) v = fmaxf(v, 0.f);
      if (rg.thresh) v = bn_keep(rg, ck, b * N + i) ? v * rg.scale : 0.f;
      yc[b * bstride + i] = v;
    }
}

// this rank's share of a cross-replica BatchNorm: per channel the local mean and M2 = sum (x - mean)^2
__global__ __launch_bounds__(kBnThreads) void bn_stats_kernel(vdetr_bnact_desc d, float* __restrict__ mean_out, float* __restrict__ m2_out) {
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= d.C) return;
  const int N = d.N, n = d.B * N;
  const size_t bstride = (size_t)d.C * N;
  const float* xc = d.x + (size_t)c * N;
  float s = 0.f;
  for (int b = 0; b < d.B; ++b)
    for (int i = lane; i < N; i += 64) s += xc[b * bstride + i];
  const float mean = wave_allsum_f32(s) / (float)n;
  float q = 0.f;
  for (int b = 0; b < d.B; ++b)
    for (int i = lane; i < N; i += 64) { const float t = xc[b * bstride + i] - mean; q += t * t; }
  q = wave_allsum_f32(q);
  if (lane == 0) { mean_out[c] = mean; m2_out[c] = q; }
}

// training-mode backward: g = dy through dropout and relu; dbeta = sum g, dgamma = sum g*xhat,
// dx = gamma * invstd * (g - dbeta/n - xhat * dgamma/n)
__device__ __forceinline__ void bn_act_bwd_body(const vdetr_bnact_desc& d, const vdetr_bnact_grads& g) {
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= d.C) return;
  const int N = d.N, n = d.B * N;
  const size_t bstride = (size_t)d.C * N;
  const float* xc = d.x + (size_t)c * N;
  const float* gc = g.dy + (size_t)c * N;
  const float mean = d.save_mean[c], invstd = d.save_invstd[c];
  const float ga = d.gamma ? d.gamma[c] : 1.f, be = d.beta ? d.beta[c] : 0.f;
  const BnRng rg = bn_rng(d);
  const unsigned ck = bn_chankey(rg, c);
  const float a = ga * invstd, sh = be - mean * a;  // the forward's expression: identical relu decisions
  auto grad_at = [&](int b, int i, float xh) {
    float v = gc[b * bstride + i];
    if (rg.thresh) v = bn_keep(rg, ck, b * N + i) ? v * rg.scale : 0.f;
    if (d.relu && !(xc[b * bstride + i] * a + sh > 0.f)) v = 0.f;
    return v;
  };
  float s1 = 0.f, s2 = 0.f;
  for (int b = 0; b < d.B; ++b)
    for (int i = lane; i < N; i += 64) {
      const float xh = (xc[b * bstride + i] - mean) * invstd;
      const float v = grad_at(b, i, xh);
      s1 += v; s2 += v * xh;
    }
  const float dbeta = wave_allsum_f32(s1), dgamma = wave_allsum_f32(s2);
  if (lane == 0) {
    if (g.d_gamma) g.d_gamma[c] = dgamma;
    if (g.d_beta) g.d_beta[c] = dbeta;
  }
  if (!g.dx) return;
  const float inv_n = g.inv_count ? g.inv_count[0] : 1.f / (float)n;
  const float k = ga * invstd, m1 = (g.sum_dy ? g.sum_dy[c] : dbeta) * inv_n, m2 = (g.sum_dy_xhat ? g.sum_dy_xhat[c] : dgamma) * inv_n;
  float* dxc = g.dx + (size_t)c * N;
  for (int b = 0; b < d.B; ++b)
    for (int i = lane; i < N; i += 64) {
      const float xh = (xc[b * bstride + i] - mean) * invstd;
      dxc[b * bstride + i] = k * (grad_at(b, i, xh) - m1 - xh * m2);
    }
}


// ---- register-resident variants: B * N == NCH * 256 (N % 4 == 0).  A lane holds NCH float4 of its channel: every load
// of the launch is issued before the first reduction (one memory latency), the row is read once.  (The sweep kernels
// above took 17 / 20 us on the heads' [1, 1280, 1024] tensors: three resp. two dependent passes of scalar loads.)
template <int NCH>
__global__ __launch_bounds__(kBnThreads) void bn_act_fwd_reg_kernel(vdetr_bnact_desc d) {
  bn_bookkeeping(d);
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= d.C) return;
  constexpr int N = NCH * 256;  // elements of the channel over the whole batch
  const int n4 = d.N >> 2;      // float4 per (scene, channel) row
  // float4 #idx of the channel -> scene idx / n4, offset idx % n4 (one scene: the identity)
  auto at = [&](const float* base, int idx) {
    const int b = idx / n4, i4 = idx - b * n4;
    return reinterpret_cast<const f32x4*>(base + ((size_t)b * d.C + c) * d.N) + i4;
  };
  f32x4 xr[NCH];
#pragma unroll
  for (int j = 0; j < NCH; ++j) xr[j] = *at(d.x, j * 64 + lane);
  const float ga = d.gamma ? d.gamma[c] : 1.f, be = d.beta ? d.beta[c] : 0.f;
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NCH; ++j) s += (xr[j][0] + xr[j][1]) + (xr[j][2] + xr[j][3]);
  float mean = wave_allsum_f32(s) * (1.f / (float)N);
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < NCH; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) { const float t = xr[j][e] - mean; q += t * t; }
  const float var = wave_allsum_f32(q) * (1.f / (float)N);
  float invstd = rsqrtf(var + d.eps);
  if (d.stats_given) {  // statistics of the batch over all ranks, from the caller
    mean = d.save_mean[c];
    invstd = d.save_invstd[c];
  } else if (lane == 0) {
    d.save_mean[c] = mean;
    d.save_invstd[c] = invstd;
    if (d.running_mean) {
      const float m = d.momentum;
      d.running_mean[c] = (1.f - m) * d.running_mean[c] + m * (mean + (d.pre_bias ? d.pre_bias[c] : 0.f));
      d.running_var[c] = (1.f - m) * d.running_var[c] + m * var * ((float)N / (float)(N - 1));
    }
  }
  const float a = ga * invstd, sh = be - mean * a;
  const BnRng rg = bn_rng(d);
  const unsigned ck = bn_chankey(rg, c);
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float t = xr[j][e] * a + sh;
      if (d.relu) t = fmaxf(t, 0.f);
      if (rg.thresh) t = bn_keep(rg, ck, (j * 64 + lane) * 4 + e) ? t * rg.scale : 0.f;
      v[e] = t;
    }
    *const_cast<f32x4*>(at(d.y, j * 64 + lane)) = v;
  }
}

template <int NCH>
__device__ __forceinline__ void bn_act_bwd_reg_body(const vdetr_bnact_desc& d, const vdetr_bnact_grads& g) {
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= d.C) return;
  constexpr int N = NCH * 256;
  const int n4 = d.N >> 2;
  auto at = [&](const float* base, int idx) {
    const int b = idx / n4, i4 = idx - b * n4;
    return reinterpret_cast<const f32x4*>(base + ((size_t)b * d.C + c) * d.N) + i4;
  };
  f32x4 xr[NCH], gr[NCH];
#pragma unroll
  for (int j = 0; j < NCH; ++j) { xr[j] = *at(d.x, j * 64 + lane); gr[j] = *at(g.dy, j * 64 + lane); }
  const float mean = d.save_mean[c], invstd = d.save_invstd[c];
  const float ga = d.gamma ? d.gamma[c] : 1.f, be = d.beta ? d.beta[c] : 0.f;
  const BnRng rg = bn_rng(d);
  const unsigned ck = bn_chankey(rg, c);
  const float a = ga * invstd, sh = be - mean * a;
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int j = 0; j < NCH; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v = gr[j][e];
      if (rg.thresh) v = bn_keep(rg, ck, (j * 64 + lane) * 4 + e) ? v * rg.scale : 0.f;
      if (d.relu && !(xr[j][e] * a + sh > 0.f)) v = 0.f;
      const float xh = (xr[j][e] - mean) * invstd;
      gr[j][e] = v;
      xr[j][e] = xh;
      s1 += v; s2 += v * xh;
    }
  const float dbeta = wave_allsum_f32(s1), dgamma = wave_allsum_f32(s2);
  if (lane == 0) {
    if (g.d_gamma) g.d_gamma[c] = dgamma;
    if (g.d_beta) g.d_beta[c] = dbeta;
  }
  if (!g.dx) return;
  const float inv_n = g.inv_count ? g.inv_count[0] : 1.f / (float)N;
  const float k = ga * invstd, m1 = (g.sum_dy ? g.sum_dy[c] : dbeta) * inv_n, m2 = (g.sum_dy_xhat ? g.sum_dy_xhat[c] : dgamma) * inv_n;
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = k * (gr[j][e] - m1 - xr[j][e] * m2);
    *const_cast<f32x4*>(at(g.dx, j * 64 + lane)) = v;
  }
}

// kernels: one problem, or a batch of independent problems of the same B*N (blockIdx.y picks the problem; their
// descriptors travel by value) -- the hidden blocks of several decoder stages' box heads in ONE backward launch
constexpr int kBnBatch = 12;
struct BnBwdBatch {
  vdetr_bnact_desc d[kBnBatch];
  vdetr_bnact_grads g[kBnBatch];
};
__global__ __launch_bounds__(kBnThreads) void bn_act_bwd_kernel(vdetr_bnact_desc d, vdetr_bnact_grads g) { bn_act_bwd_body(d, g); }
template <int NCH>
__global__ __launch_bounds__(kBnThreads) void bn_act_bwd_reg_kernel(vdetr_bnact_desc d, vdetr_bnact_grads g) {
  bn_act_bwd_reg_body<NCH>(d, g);
}
__global__ __launch_bounds__(kBnThreads) void bn_act_bwd_batch_kernel(BnBwdBatch batch) {
  bn_act_bwd_body(batch.d[blockIdx.y], batch.g[blockIdx.y]);
}
template <int NCH>
__global__ __launch_bounds__(kBnThreads) void bn_act_bwd_reg_batch_kernel(BnBwdBatch batch) {
  bn_act_bwd_reg_body<NCH>(batch.d[blockIdx.y], batch.g[blockIdx.y]);
}

// ---- y = dropout(relu(x)) element-wise (the FFN's `self.dropout(self.activation(self.linear1(.)))`,
// models/vdetr_transformer.py:566,604): one launch instead of clamp + dropout; the backward needs only y
// (y > 0 <=> the element passed the relu AND was kept): dx = y > 0 ? dy * scale : 0.
__global__ __launch_bounds__(256) void relu_dropout_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long n4,
                                                              vdetr_bnact_desc d) {
  const BnRng rg = bn_rng(d);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
    unsigned r0 = 0xFFFFFFFFu, r1 = 0xFFFFFFFFu;
    if (rg.thresh) {
      r0 = fmix32(((unsigned)i * 0x9E3779B1u + rg.off_lo) ^ rg.seed_lo ^ ((unsigned)(i >> 32) * 0x27D4EB2Fu));
      r0 = fmix32(r0 ^ rg.seed_hi ^ rg.off_hi);
      r1 = fmix32(r0 + 0x9E3779B9u);
    }
    const unsigned k[4] = {r0 & 0xFFFFu, r0 >> 16, r1 & 0xFFFFu, r1 >> 16};
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (v[e] > 0.f && k[e] >= rg.thresh) ? v[e] * rg.scale : 0.f;
    reinterpret_cast<f32x4*>(y)[i] = v;
  }
}
__global__ __launch_bounds__(256) void relu_dropout_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy,
                                                              float* __restrict__ dx, long n4, float scale) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const f32x4 yv = reinterpret_cast<const f32x4*>(y)[i];
    f32x4 g = reinterpret_cast<const f32x4*>(dy)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) g[e] = yv[e] > 0.f ? g[e] * scale : 0.f;
    reinterpret_cast<f32x4*>(dx)[i] = g;
  }
}

}  // namespace vdetr

using namespace vdetr;

static int bnact_check(const vdetr_bnact_desc* d, const char* op) {
  VDETR_REQUIRE(d != nullptr, "%s: null descriptor", op);
  VDETR_REQUIRE(d->B > 0 && d->C > 0 && d->N > 0, "%s: empty tensor B=%d C=%d N=%d", op, d->B, d->C, d->N);
  VDETR_REQUIRE(d->x, "%s: null input", op);
  VDETR_REQUIRE((d->gamma == nullptr) == (d->beta == nullptr), "%s: gamma and beta go together", op);
  VDETR_REQUIRE((d->running_mean == nullptr) == (d->running_var == nullptr), "%s: running_mean and running_var go together", op);
  VDETR_REQUIRE(d->dropout_p >= 0.f && d->dropout_p < 1.f, "%s: dropout_p %f outside [0,1)", op, d->dropout_p);
  VDETR_REQUIRE(d->ncounters >= 0 && d->ncounters <= 8, "%s: ncounters %d outside [0,8]", op, d->ncounters);
  return VDETR_OK;
}

extern "C" int vdetr_bn_act_fwd_f32(const vdetr_bnact_desc* d, vdetr_stream_t stream) {
  if (int e = bnact_check(d, "bn_act_fwd")) return e;
  VDETR_REQUIRE(d->y, "bn_act_fwd: null output");
  VDETR_REQUIRE(!d->stats_given || d->training, "bn_act_fwd: stats_given is a training-mode option");
  VDETR_REQUIRE(d->training ? (d->save_mean && d->save_invstd) : (d->running_mean != nullptr),
                "bn_act_fwd: training needs save_mean / save_invstd, eval needs the running statistics");
  VDETR_REQUIRE(d->training || d->dropout_p == 0.f, "bn_act_fwd: dropout in eval mode");
  const dim3 grid(ceil_div(d->C, 4)), block(kBnThreads);
  hipStream_t st = (hipStream_t)stream;
  const long tot = (long)d->B * d->N;
  const bool reg = d->training && d->N % 4 == 0 && tot % 256 == 0 && tot <= 4096 && ((uintptr_t)d->x & 15) == 0 &&
                   ((uintptr_t)d->y & 15) == 0;
  switch (reg ? (int)(tot / 256) : 0) {
    case 1: hipLaunchKernelGGL(bn_act_fwd_reg_kernel<1>, grid, block, 0, st, *d); break;
    case 2: hipLaunchKernelGGL(bn_act_fwd_reg_kernel<2>, grid, block, 0, st, *d); break;
    case 4: hipLaunchKernelGGL(bn_act_fwd_reg_kernel<4>, grid, block, 0, st, *d); break;
    case 8: hipLaunchKernelGGL(bn_act_fwd_reg_kernel<8>, grid, block, 0, st, *d); break;
    case 16: hipLaunchKernelGGL(bn_act_fwd_reg_kernel<16>, grid, block, 0, st, *d); break;
    default: hipLaunchKernelGGL(bn_act_fwd_kernel, grid, block, 0, st, *d); break;
  }
  return check_launch("bn_act_fwd");
}

extern "C" int vdetr_bn_stats_f32(const vdetr_bnact_desc* d, float* mean, float* m2, vdetr_stream_t stream) {
  VDETR_REQUIRE(d != nullptr && d->x && mean && m2, "bn_stats: null pointer");
  VDETR_REQUIRE(d->B > 0 && d->C > 0 && d->N > 0, "bn_stats: empty tensor B=%d C=%d N=%d", d->B, d->C, d->N);
  hipLaunchKernelGGL(bn_stats_kernel, dim3(ceil_div(d->C, 4)), dim3(kBnThreads), 0, (hipStream_t)stream, *d, mean, m2);
  return check_launch("bn_stats");
}

extern "C" int vdetr_bn_act_bwd_f32(const vdetr_bnact_desc* d, const vdetr_bnact_grads* g, vdetr_stream_t stream) {
  if (int e = bnact_check(d, "bn_act_bwd")) return e;
  VDETR_REQUIRE(d->training, "bn_act_bwd: only the training-mode backward is built (batch statistics)");
  VDETR_REQUIRE(g && g->dy && d->save_mean && d->save_invstd, "bn_act_bwd: null pointer");
  VDETR_REQUIRE(g->dx || g->d_gamma || g->d_beta, "bn_act_bwd: nothing to compute");
  VDETR_REQUIRE((g->sum_dy == nullptr) == (g->sum_dy_xhat == nullptr) && (g->sum_dy == nullptr) == (g->inv_count == nullptr),
                "bn_act_bwd: the cross-replica sums and the inverse count go together");
  const dim3 grid(ceil_div(d->C, 4)), block(kBnThreads);
  hipStream_t st = (hipStream_t)stream;
  const long tot = (long)d->B * d->N;
  const bool reg = d->N % 4 == 0 && tot % 256 == 0 && tot <= 4096 && ((uintptr_t)d->x & 15) == 0 &&
                   ((uintptr_t)g->dy & 15) == 0 && (!g->dx || ((uintptr_t)g->dx & 15) == 0);
  switch (reg ? (int)(tot / 256) : 0) {
    case 1: hipLaunchKernelGGL(bn_act_bwd_reg_kernel<1>, grid, block, 0, st, *d, *g); break;
    case 2: hipLaunchKernelGGL(bn_act_bwd_reg_kernel<2>, grid, block, 0, st, *d, *g); break;
    case 4: hipLaunchKernelGGL(bn_act_bwd_reg_kernel<4>, grid, block, 0, st, *d, *g); break;
    case 8: hipLaunchKernelGGL(bn_act_bwd_reg_kernel<8>, grid, block, 0, st, *d, *g); break;
    case 16: hipLaunchKernelGGL(bn_act_bwd_reg_kernel<16>, grid, block, 0, st, *d, *g); break;
    default: hipLaunchKernelGGL(bn_act_bwd_kernel, grid, block, 0, st, *d, *g); break;
  }
  return check_launch("bn_act_bwd");
}

extern "C" int vdetr_relu_dropout_fwd_f32(const float* x, float* y, long n, float dropout_p, uint64_t seed, uint64_t offset,
                                          const uint64_t* rng_state, vdetr_stream_t stream) {
  VDETR_REQUIRE(x && y && n > 0 && n % 4 == 0, "relu_dropout_fwd: needs non-null tensors with a multiple of 4 elements");
  VDETR_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "relu_dropout_fwd: dropout_p %f outside [0,1)", dropout_p);
  VDETR_REQUIRE((((uintptr_t)x | (uintptr_t)y) & 15) == 0, "relu_dropout_fwd: tensors must be 16-B aligned");
  vdetr_bnact_desc d = {};
  d.dropout_p = dropout_p; d.seed = seed; d.offset = offset; d.rng_state = rng_state;
  const long n4 = n / 4;
  const int grid = (int)(n4 / 256 + 1 < 2048 ? n4 / 256 + 1 : 2048);
  hipLaunchKernelGGL(relu_dropout_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, y, n4, d);
  return check_launch("relu_dropout_fwd");
}

extern "C" int vdetr_relu_dropout_bwd_f32(const float* y, const float* dy, float* dx, long n, float dropout_p,
                                          vdetr_stream_t stream) {
  VDETR_REQUIRE(y && dy && dx && n > 0 && n % 4 == 0, "relu_dropout_bwd: needs non-null tensors with a multiple of 4 elements");
  VDETR_REQUIRE((((uintptr_t)y | (uintptr_t)dy | (uintptr_t)dx) & 15) == 0, "relu_dropout_bwd: tensors must be 16-B aligned");
  vdetr_bnact_desc d = {};
  d.dropout_p = dropout_p;
  const BnRng rg = [&] { BnRng r; r.thresh = 0; r.scale = 1.f; if (dropout_p > 0.f) { int t = (int)((double)dropout_p * 65536.0 + 0.5); t = t < 1 ? 1 : (t > 65535 ? 65535 : t); r.scale = 65536.f / (float)(65536 - t); } return r; }();
  const long n4 = n / 4;
  const int grid = (int)(n4 / 256 + 1 < 2048 ? n4 / 256 + 1 : 2048);
  hipLaunchKernelGGL(relu_dropout_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, y, dy, dx, n4, rg.scale);
  return check_launch("relu_dropout_bwd");
}

extern "C" int vdetr_bn_act_bwd_batch_f32(const vdetr_bnact_desc* descs, const vdetr_bnact_grads* grads, int n,
                                          vdetr_stream_t stream) {
  VDETR_REQUIRE(descs && grads && n >= 1, "bn_act_bwd_batch: null descriptors");
  hipStream_t st = (hipStream_t)stream;
  for (int s0 = 0; s0 < n; s0 += kBnBatch) {
    const int cnt = n - s0 < kBnBatch ? n - s0 : kBnBatch;
    BnBwdBatch batch{};
    bool reg = true;
    int maxc = 1;
    const long tot = (long)descs[s0].B * descs[s0].N;
    for (int k = 0; k < cnt; ++k) {
      const vdetr_bnact_desc* d = descs + s0 + k;
      const vdetr_bnact_grads* g = grads + s0 + k;
      if (int e = bnact_check(d, "bn_act_bwd_batch")) return e;
      VDETR_REQUIRE(d->training, "bn_act_bwd_batch: only the training-mode backward is built (batch statistics)");
      VDETR_REQUIRE(g->dy && d->save_mean && d->save_invstd, "bn_act_bwd_batch: null pointer");
      VDETR_REQUIRE((long)d->B * d->N == tot, "bn_act_bwd_batch: problem %d has B*N = %ld, the first %ld", s0 + k, (long)d->B * d->N, tot);
      reg = reg && d->N % 4 == 0 && ((uintptr_t)d->x & 15) == 0 && ((uintptr_t)g->dy & 15) == 0 && (!g->dx || ((uintptr_t)g->dx & 15) == 0);
      maxc = d->C > maxc ? d->C : maxc;
      batch.d[k] = *d;
      batch.g[k] = *g;
    }
    reg = reg && tot % 256 == 0 && tot <= 4096;
    const dim3 grid(ceil_div(maxc, 4), cnt), block(kBnThreads);
    switch (reg ? (int)(tot / 256) : 0) {
      case 1: hipLaunchKernelGGL(bn_act_bwd_reg_batch_kernel<1>, grid, block, 0, st, batch); break;
      case 2: hipLaunchKernelGGL(bn_act_bwd_reg_batch_kernel<2>, grid, block, 0, st, batch); break;
      case 4: hipLaunchKernelGGL(bn_act_bwd_reg_batch_kernel<4>, grid, block, 0, st, batch); break;
      case 8: hipLaunchKernelGGL(bn_act_bwd_reg_batch_kernel<8>, grid, block, 0, st, batch); break;
      case 16: hipLaunchKernelGGL(bn_act_bwd_reg_batch_kernel<16>, grid, block, 0, st, batch); break;
      default: hipLaunchKernelGGL(bn_act_bwd_batch_kernel, grid, block, 0, st, batch); break;
    }
    if (int e = check_launch("bn_act_bwd_batch")) return e;
  }
  return VDETR_OK;
}
