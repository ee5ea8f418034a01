// sp_bn.hip — BatchNorm (+ residual) (+ ReLU / ELU) over the point-major feature table [N, C] of a sparse tensor, gfx950.
//
// Reference: ME.MinkowskiBatchNorm = nn.BatchNorm1d over all sites of the batch, followed by MinkowskiReLU / MinkowskiELU,
// and in the residual blocks by `out += residual; relu` (models/mink_resnet.py:38-84 via MinkowskiEngine's BasicBlock;
// models/model_vdetr.py:141-176).  ATen's channels-last statistics kernels take ~41 us per call on a [36k, 64] table (9 MB:
// 2 us of HBM time) and the block needs 4-6 launches; here
//   forward : stats partials (a workgroup per 128 rows: mean and sum of squared deviations from ITS mean, two sweeps over
//             rows that stay in cache) -> apply (every workgroup merges the partials with Chan's formula in a fixed order —
//             deterministic — and writes y = act(xhat * gamma + beta + residual); workgroup 0 also stores mean / invstd
//             and updates the running statistics)
//   backward: partials of sum(g), sum(g * xhat) with g = dy * act'(y) -> apply dx = gamma * invstd * (g - mean(g) -
//             xhat * mean(g xhat)), dresidual = g; workgroup 0 stores dgamma / dbeta.
// Pure HBM streams: x (and residual) read twice / once, y written once.
#include "common.h"

namespace vdetr {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kBnRowsMin = 16, kBnRowsMax = 128;  // rows per workgroup: chosen so that a table gives >= ~1000 workgroups
static inline int bn_rows(int N) { int r = N / 1024; r = r < kBnRowsMin ? kBnRowsMin : r > kBnRowsMax ? kBnRowsMax : r; return (r + 15) / 16 * 16; }

__device__ __forceinline__ f32x4 bn_act(f32x4 v, int act) {
  if (act == 1) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
  } else if (act == 2) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : expm1f(v[e]);
  }
  return v;
}
// d act / d pre-activation from the OUTPUT y
__device__ __forceinline__ f32x4 bn_act_grad(f32x4 dy, f32x4 y, int act) {
  if (act == 1) {
#pragma unroll
    for (int e = 0; e < 4; ++e) dy[e] = y[e] > 0.f ? dy[e] : 0.f;
  } else if (act == 2) {
#pragma unroll
    for (int e = 0; e < 4; ++e) dy[e] = y[e] > 0.f ? dy[e] : dy[e] * (y[e] + 1.f);
  }
  return dy;
}

// thread layout: c4 = tid % C4 owns 4 channels, rs = tid / C4 walks rows rs, rs + RPI, ... of the workgroup's 128 rows
// partial [nblk][3][C]: count, mean, M2   (forward)   /   [nblk][2][C]: sum g, sum g xhat   (backward)
__global__ __launch_bounds__(256) void sp_bn_stats_kernel(const float* __restrict__ x, int N, int C, int kBnRows, float* __restrict__ part) {
  __shared__ f32x4 red[256];
  const int C4 = C >> 2, tid = threadIdx.x, c4 = tid % C4, rs = tid / C4, rpi = 256 / C4;
  const int r0 = blockIdx.x * kBnRows, r1 = min(N, r0 + kBnRows);
  const bool act = rs < rpi;  // 256 % C4 leftover threads idle
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (act)
    for (int r = r0 + rs; r < r1; r += rpi) s += reinterpret_cast<const f32x4*>(x)[(size_t)r * C4 + c4];
  red[tid] = s;
  __syncthreads();
  f32x4 tot = {0.f, 0.f, 0.f, 0.f};
  for (int j = 0; j < rpi; ++j) tot += red[j * C4 + c4];  // fixed order
  const float cnt = (float)(r1 - r0);
  const f32x4 mean = tot / cnt;
  __syncthreads();
  f32x4 m2 = {0.f, 0.f, 0.f, 0.f};
  if (act)
    for (int r = r0 + rs; r < r1; r += rpi) {
      const f32x4 d = reinterpret_cast<const f32x4*>(x)[(size_t)r * C4 + c4] - mean;
      m2 += d * d;
    }
  red[tid] = m2;
  __syncthreads();
  if (rs == 0) {
    f32x4 t2 = {0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < rpi; ++j) t2 += red[j * C4 + c4];
    float* p = part + (size_t)blockIdx.x * 3 * C;
    reinterpret_cast<f32x4*>(p)[c4] = f32x4{cnt, cnt, cnt, cnt};
    reinterpret_cast<f32x4*>(p + C)[c4] = mean;
    reinterpret_cast<f32x4*>(p + 2 * C)[c4] = t2;
  }
}

// Chan et al.: merge of (count, mean, M2) aggregates
struct BnAgg {
  float n;
  f32x4 mu, m2;
};
__device__ __forceinline__ void bn_agg_add(BnAgg& a, float nb, const f32x4& mb, const f32x4& sb) {
  if (nb <= 0.f) return;
  const float nn = a.n + nb;
  const f32x4 d = mb - a.mu;
  a.mu += d * (nb / nn);
  a.m2 += sb + d * d * (a.n * nb / nn);
  a.n = nn;
}

struct SpBnParams {
  int N, C, act, training, nblk, rows;
  float eps, momentum;
  const float *x, *gamma, *beta, *residual;
  float *running_mean, *running_var, *y, *save_mean, *save_invstd;
  float* part;
  float* sums;  // backward: [2][C] = sum g, sum g xhat (behind the partials in the workspace)
  long long* num_batches_tracked;
};

// one workgroup per 4 channels: thread t merges the partials t, t + 256, ... ; the 256 aggregates are merged in index order by
// a fixed tree through LDS (deterministic); thread 0 stores mean / invstd and updates the running statistics
__global__ __launch_bounds__(256) void sp_bn_finalize_kernel(SpBnParams P) {
  __shared__ float lds[4][9];
  const int c4 = blockIdx.x, tid = threadIdx.x;
  BnAgg a{0.f, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  for (int b = tid; b < P.nblk; b += 256) {
    const float* p = P.part + (size_t)b * 3 * P.C;
    bn_agg_add(a, p[c4 * 4], reinterpret_cast<const f32x4*>(p + P.C)[c4], reinterpret_cast<const f32x4*>(p + 2 * P.C)[c4]);
  }
  // fixed merge tree (deterministic): inside a wave through lane shuffles (no barrier), then the 4 wave aggregates through LDS.
  // (The first version merged all 256 aggregates through LDS: 8 levels, 16 barriers, 7.8 us for a few KB.)
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const float on = __shfl_down(a.n, off);
    f32x4 omu, om2;
#pragma unroll
    for (int e = 0; e < 4; ++e) { omu[e] = __shfl_down(a.mu[e], off); om2[e] = __shfl_down(a.m2[e], off); }
    if ((tid & 63) < off) bn_agg_add(a, on, omu, om2);
  }
  if ((tid & 63) == 0) {
    float* o = lds[tid >> 6];
    o[0] = a.n;
#pragma unroll
    for (int e = 0; e < 4; ++e) { o[1 + e] = a.mu[e]; o[5 + e] = a.m2[e]; }
  }
  __syncthreads();
  if (tid == 0) {
    for (int w2 = 1; w2 < 4; ++w2) {
      const float* o = lds[w2];
      bn_agg_add(a, o[0], f32x4{o[1], o[2], o[3], o[4]}, f32x4{o[5], o[6], o[7], o[8]});
    }
  }
  if (tid == 0) {
    const f32x4 mean = a.mu, var = a.m2 / a.n;
    f32x4 invstd;
#pragma unroll
    for (int e = 0; e < 4; ++e) invstd[e] = rsqrtf(var[e] + P.eps);
    reinterpret_cast<f32x4*>(P.save_mean)[c4] = mean;
    reinterpret_cast<f32x4*>(P.save_invstd)[c4] = invstd;
    if (P.running_mean) {  // nn.BatchNorm1d: running = (1 - m) running + m stat, unbiased variance
      const float unb = a.n > 1.f ? a.n / (a.n - 1.f) : 1.f;
      f32x4* rm = reinterpret_cast<f32x4*>(P.running_mean) + c4;
      f32x4* rv = reinterpret_cast<f32x4*>(P.running_var) + c4;
      *rm = *rm * (1.f - P.momentum) + mean * P.momentum;
      *rv = *rv * (1.f - P.momentum) + var * (unb * P.momentum);
      if (c4 == 0 && P.num_batches_tracked) *P.num_batches_tracked += 1;
    }
  }
}

__global__ __launch_bounds__(256) void sp_bn_apply_kernel(SpBnParams P) {
  const int C4 = P.C >> 2, tid = threadIdx.x, c4 = tid % C4, rs = tid / C4, rpi = 256 / C4;
  if (rs >= rpi) return;
  f32x4 mean, invstd;
  if (P.training) {  // left by sp_bn_finalize_kernel
    mean = reinterpret_cast<const f32x4*>(P.save_mean)[c4];
    invstd = reinterpret_cast<const f32x4*>(P.save_invstd)[c4];
  } else {
    mean = reinterpret_cast<const f32x4*>(P.running_mean)[c4];
    const f32x4 var = reinterpret_cast<const f32x4*>(P.running_var)[c4];
#pragma unroll
    for (int e = 0; e < 4; ++e) invstd[e] = rsqrtf(var[e] + P.eps);
    if (blockIdx.x == 0 && rs == 0 && P.save_mean) {  // what a backward pass in eval mode normalises with
      reinterpret_cast<f32x4*>(P.save_mean)[c4] = mean;
      reinterpret_cast<f32x4*>(P.save_invstd)[c4] = invstd;
    }
  }
  const f32x4 g = P.gamma ? reinterpret_cast<const f32x4*>(P.gamma)[c4] : f32x4{1.f, 1.f, 1.f, 1.f};
  const f32x4 b = P.beta ? reinterpret_cast<const f32x4*>(P.beta)[c4] : f32x4{0.f, 0.f, 0.f, 0.f};
  const f32x4 scale = invstd * g, shift = b - mean * scale;
  const int r0 = blockIdx.x * P.rows, r1 = min(P.N, r0 + P.rows);
  for (int r = r0 + rs; r < r1; r += rpi) {
    const size_t i = (size_t)r * C4 + c4;
    f32x4 v = reinterpret_cast<const f32x4*>(P.x)[i] * scale + shift;
    if (P.residual) v += reinterpret_cast<const f32x4*>(P.residual)[i];
    reinterpret_cast<f32x4*>(P.y)[i] = bn_act(v, P.act);
  }
}

struct SpBnGrads {
  const float* dy;
  float *dx, *dres, *dgamma, *dbeta;
};

// partial [nblk][2][C]: sum g, sum g * xhat
__global__ __launch_bounds__(256) void sp_bn_bwd_stats_kernel(SpBnParams P, SpBnGrads G) {
  __shared__ f32x4 red[2][256];
  const int C4 = P.C >> 2, tid = threadIdx.x, c4 = tid % C4, rs = tid / C4, rpi = 256 / C4;
  const int r0 = blockIdx.x * P.rows, r1 = min(P.N, r0 + P.rows);
  f32x4 sg = {0.f, 0.f, 0.f, 0.f}, sgx = {0.f, 0.f, 0.f, 0.f};
  if (rs < rpi) {
    const f32x4 mean = reinterpret_cast<const f32x4*>(P.save_mean)[c4], invstd = reinterpret_cast<const f32x4*>(P.save_invstd)[c4];
    for (int r = r0 + rs; r < r1; r += rpi) {
      const size_t i = (size_t)r * C4 + c4;
      const f32x4 g = bn_act_grad(reinterpret_cast<const f32x4*>(G.dy)[i], reinterpret_cast<const f32x4*>(P.y)[i], P.act);
      const f32x4 xh = (reinterpret_cast<const f32x4*>(P.x)[i] - mean) * invstd;
      sg += g;
      sgx += g * xh;
    }
  }
  red[0][tid] = sg;
  red[1][tid] = sgx;
  __syncthreads();
  if (rs == 0) {
    f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < rpi; ++j) { a += red[0][j * C4 + c4]; b += red[1][j * C4 + c4]; }
    float* p = P.part + (size_t)blockIdx.x * 2 * P.C;
    reinterpret_cast<f32x4*>(p)[c4] = a;
    reinterpret_cast<f32x4*>(p + P.C)[c4] = b;
  }
}

// one workgroup per 4 channels: sums of the partials (thread t: partials t, t + 256, ...; fixed tree) -> sums, dgamma, dbeta
__global__ __launch_bounds__(256) void sp_bn_bwd_finalize_kernel(SpBnParams P, SpBnGrads G) {
  __shared__ f32x4 red[2][4];
  const int c4 = blockIdx.x, tid = threadIdx.x;
  f32x4 sg = {0.f, 0.f, 0.f, 0.f}, sgx = {0.f, 0.f, 0.f, 0.f};
  for (int b = tid; b < P.nblk; b += 256) {
    const float* p = P.part + (size_t)b * 2 * P.C;
    sg += reinterpret_cast<const f32x4*>(p)[c4];
    sgx += reinterpret_cast<const f32x4*>(p + P.C)[c4];
  }
  // fixed tree: lane shuffles inside a wave, the 4 wave sums through LDS (one barrier instead of 16)
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float a = __shfl_down(sg[e], off), b = __shfl_down(sgx[e], off);
      if ((tid & 63) < off) { sg[e] += a; sgx[e] += b; }
    }
  }
  if ((tid & 63) == 0) { red[0][tid >> 6] = sg; red[1][tid >> 6] = sgx; }
  __syncthreads();
  if (tid == 0) {
    for (int w2 = 1; w2 < 4; ++w2) { sg += red[0][w2]; sgx += red[1][w2]; }
    reinterpret_cast<f32x4*>(P.sums)[c4] = sg;
    reinterpret_cast<f32x4*>(P.sums + P.C)[c4] = sgx;
    if (G.dgamma) reinterpret_cast<f32x4*>(G.dgamma)[c4] = sgx;
    if (G.dbeta) reinterpret_cast<f32x4*>(G.dbeta)[c4] = sg;
  }
}

__global__ __launch_bounds__(256) void sp_bn_bwd_apply_kernel(SpBnParams P, SpBnGrads G) {
  const int C4 = P.C >> 2, tid = threadIdx.x, c4 = tid % C4, rs = tid / C4, rpi = 256 / C4;
  if (rs >= rpi) return;
  const f32x4 sg = reinterpret_cast<const f32x4*>(P.sums)[c4], sgx = reinterpret_cast<const f32x4*>(P.sums + P.C)[c4];
  const f32x4 mean = reinterpret_cast<const f32x4*>(P.save_mean)[c4], invstd = reinterpret_cast<const f32x4*>(P.save_invstd)[c4];
  const f32x4 gam = P.gamma ? reinterpret_cast<const f32x4*>(P.gamma)[c4] : f32x4{1.f, 1.f, 1.f, 1.f};
  const float invn = 1.f / (float)P.N;
  const f32x4 mg = sg * invn, mgx = sgx * invn, k = gam * invstd;
  const int r0 = blockIdx.x * P.rows, r1 = min(P.N, r0 + P.rows);
  for (int r = r0 + rs; r < r1; r += rpi) {
    const size_t i = (size_t)r * C4 + c4;
    const f32x4 g = bn_act_grad(reinterpret_cast<const f32x4*>(G.dy)[i], reinterpret_cast<const f32x4*>(P.y)[i], P.act);
    if (G.dres) reinterpret_cast<f32x4*>(G.dres)[i] = g;
    if (G.dx) {
      const f32x4 xh = (reinterpret_cast<const f32x4*>(P.x)[i] - mean) * invstd;
      reinterpret_cast<f32x4*>(G.dx)[i] = P.training ? k * (g - mg - xh * mgx) : k * g;
    }
  }
}

static int fill(const vdetr_spbn_desc* d, SpBnParams* P, const char* op) {
  VDETR_REQUIRE(d && d->N >= 0 && d->C > 0, "%s: bad descriptor", op);
  VDETR_REQUIRE(d->C % 4 == 0 && d->C <= 1024, "%s: C=%d must be a multiple of 4 and <= 1024", op, d->C);
  VDETR_REQUIRE(d->act >= 0 && d->act <= 2, "%s: act %d", op, d->act);
  P->N = d->N; P->C = d->C; P->act = d->act; P->training = d->training; P->rows = bn_rows(d->N); P->nblk = ceil_div(d->N, P->rows);
  P->eps = d->eps; P->momentum = d->momentum;
  P->x = d->x; P->gamma = d->gamma; P->beta = d->beta; P->residual = d->residual;
  P->running_mean = d->running_mean; P->running_var = d->running_var; P->y = d->y;
  P->save_mean = d->save_mean; P->save_invstd = d->save_invstd; P->part = (float*)d->workspace;
  P->sums = P->part ? P->part + (size_t)P->nblk * 3 * d->C : nullptr;
  P->num_batches_tracked = (long long*)d->num_batches_tracked;
  return VDETR_OK;
}

}  // namespace vdetr

using namespace vdetr;

extern "C" size_t vdetr_sp_bn_workspace_bytes(int N, int C) {
  return ((size_t)ceil_div(N > 0 ? N : 1, bn_rows(N)) * 3 + 2) * C * sizeof(float);  // partials + the backward's two sums
}

extern "C" int vdetr_sp_bn_act_fwd_f32(const vdetr_spbn_desc* d, vdetr_stream_t stream) {
  SpBnParams P;
  if (int e = fill(d, &P, "sp_bn_act_fwd")) return e;
  if (d->N == 0) return VDETR_OK;
  VDETR_REQUIRE(d->x && d->y, "sp_bn_act_fwd: null pointer");
  VDETR_REQUIRE(d->training ? (d->workspace && d->save_mean && d->save_invstd) : (d->running_mean && d->running_var),
                "sp_bn_act_fwd: %s", d->training ? "training needs workspace, save_mean, save_invstd" : "eval needs running statistics");
  if (d->training) {
    hipLaunchKernelGGL(sp_bn_stats_kernel, dim3(P.nblk), dim3(256), 0, (hipStream_t)stream, d->x, d->N, d->C, P.rows, P.part);
    if (int e = check_launch("sp_bn_stats")) return e;
    hipLaunchKernelGGL(sp_bn_finalize_kernel, dim3(d->C / 4), dim3(256), 0, (hipStream_t)stream, P);
    if (int e = check_launch("sp_bn_finalize")) return e;
  }
  hipLaunchKernelGGL(sp_bn_apply_kernel, dim3(P.nblk), dim3(256), 0, (hipStream_t)stream, P);
  return check_launch("sp_bn_apply");
}

extern "C" int vdetr_sp_bn_act_bwd_f32(const vdetr_spbn_desc* d, const float* dy, float* dx, float* dresidual, float* dgamma,
                                       float* dbeta, vdetr_stream_t stream) {
  SpBnParams P;
  if (int e = fill(d, &P, "sp_bn_act_bwd")) return e;
  if (d->N == 0) return VDETR_OK;
  VDETR_REQUIRE(d->x && d->y && dy && d->workspace && d->save_mean && d->save_invstd, "sp_bn_act_bwd: null pointer");
  SpBnGrads G{dy, dx, dresidual, dgamma, dbeta};
  hipLaunchKernelGGL(sp_bn_bwd_stats_kernel, dim3(P.nblk), dim3(256), 0, (hipStream_t)stream, P, G);
  if (int e = check_launch("sp_bn_bwd_stats")) return e;
  hipLaunchKernelGGL(sp_bn_bwd_finalize_kernel, dim3(d->C / 4), dim3(256), 0, (hipStream_t)stream, P, G);
  if (int e = check_launch("sp_bn_bwd_finalize")) return e;
  hipLaunchKernelGGL(sp_bn_bwd_apply_kernel, dim3(P.nblk), dim3(256), 0, (hipStream_t)stream, P, G);
  return check_launch("sp_bn_bwd_apply");
}
