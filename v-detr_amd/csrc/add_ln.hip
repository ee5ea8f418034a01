// add_ln.hip — y = x + dropout(r);  out = LayerNorm(y; gamma, beta)  [out2 = LayerNorm(y; gamma2, beta2)], forward and
// backward, ONE launch each.
//
// Reference: the pre-norm residual blocks of GlobalDecoderLayer.forward_pre (models/vdetr_transformer.py:531-568):
//     tgt = tgt + self.dropoutN(tgt2);  tgt2 = self.norm{N+1}(tgt)
// and TransformerDecoder's `self.norm(output)` next to the following layer's `norm1(output)` (:401,:433: two affine
// maps of the same normalised tensor).  In ATen this is dropout + add + 2 LayerNorm launches forward and ~6 backward per
// block; the step pays ~4.5 us per launch whatever its size, and these tensors are 1 MB.
// One wave per row (C <= 1024 channels, 4 per lane and pass); statistics through DPP all-reduces; the dropout mask is
// the counter-based hash of attn_common.h keyed by (row, channel), so backward regenerates it.
// Backward: per-workgroup partial dgamma / dbeta in registers -> LDS -> global partials, summed in a fixed order by a
// second, tiny launch (no atomics, no pre-zeroed buffers: run-to-run deterministic).
#include "attn_common.h"

namespace vdetr {

constexpr int kLnThreads = 256;                 // 4 waves = 4 rows in flight per workgroup
constexpr int kLnRowsPerWg = 8;                 // forward: rows a workgroup walks (2 per wave; 128 workgroups at 1024 rows)
constexpr int kLnThreadsBwd = 256;              // backward: 4 waves x 4 rows, all loads of a wave issued at once
constexpr int kLnRowsPerWgBwd = 16;
constexpr int kLnMaxC = 1024;

struct LnRng {
  unsigned seed_lo, seed_hi, off_lo, off_hi, thresh;
  float scale;
};
__device__ __forceinline__ LnRng ln_rng(const vdetr_addln_desc& d) {
  LnRng r;
  unsigned long long s = d.seed, o = d.offset;
  if (d.rng_state) { s ^= d.rng_state[0]; o += d.rng_state[1]; }
  r.seed_lo = (unsigned)s; r.seed_hi = (unsigned)(s >> 32); r.off_lo = (unsigned)o; r.off_hi = (unsigned)(o >> 32);
  r.thresh = 0; r.scale = 1.f;
  if (d.dropout_p > 0.f && d.r) {
    int t = (int)((double)d.dropout_p * 65536.0 + 0.5);
    t = t < 1 ? 1 : (t > 65535 ? 65535 : t);
    r.thresh = (unsigned)t;
    r.scale = 65536.f / (float)(65536 - t);
  }
  return r;
}
// keep flags of channels 4*j .. 4*j+3 of `row` (one hash per 4 channels: 16-bit lanes of two chained mixes)
__device__ __forceinline__ void ln_keep4(const LnRng& g, unsigned rowkey, int j, bool (&keep)[4]) {
  if (!g.thresh) { keep[0] = keep[1] = keep[2] = keep[3] = true; return; }
  const unsigned x = fmix32(rowkey ^ ((unsigned)j * 0x165667B1u));
  const unsigned y = fmix32(x + 0x9E3779B9u);
  keep[0] = (x & 0xFFFFu) >= g.thresh; keep[1] = (x >> 16) >= g.thresh;
  keep[2] = (y & 0xFFFFu) >= g.thresh; keep[3] = (y >> 16) >= g.thresh;
}
__device__ __forceinline__ unsigned ln_rowkey(const LnRng& g, int row) {
  const unsigned x = fmix32(((unsigned)row * 0x9E3779B1u + g.off_lo) ^ g.seed_lo);
  return fmix32(x ^ (0x27D4EB2Fu + g.off_hi) ^ g.seed_hi);
}

template <int NPASS>  // 256 channels (64 lanes x 4) per pass
__global__ __launch_bounds__(kLnThreads) void add_ln_fwd_kernel(vdetr_addln_desc d) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  constexpr int C = NPASS * 256, npass = NPASS;
  const LnRng g = ln_rng(d);
  const float invC = 1.f / (float)C;
  for (int i = wv; i < kLnRowsPerWg; i += 4) {
    const int row = blockIdx.x * kLnRowsPerWg + i;
    if (row >= d.rows) return;
    const size_t base = (size_t)row * C;
    const unsigned rowkey = ln_rowkey(g, row);
    f32x4 y[NPASS];
    float s = 0.f;
#pragma unroll
    for (int p = 0; p < npass; ++p) {
      const int j = p * 64 + lane;
      f32x4 v = reinterpret_cast<const f32x4*>(d.x + base)[j];
      if (d.r) {
        const f32x4 rr = reinterpret_cast<const f32x4*>(d.r + base)[j];
        bool keep[4];
        ln_keep4(g, rowkey, j, keep);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += keep[e] ? rr[e] * g.scale : 0.f;
        reinterpret_cast<f32x4*>(d.y + base)[j] = v;
      }
      y[p] = v;
      s += (v[0] + v[1]) + (v[2] + v[3]);
    }
    const float mean = wave_allsum_f32(s) * invC;
    float q = 0.f;
#pragma unroll
    for (int p = 0; p < npass; ++p)
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float t = y[p][e] - mean; q += t * t; }
    const float rstd = rsqrtf(wave_allsum_f32(q) * invC + d.eps);
    if (lane == 0) { d.mean[row] = mean; d.rstd[row] = rstd; }
#pragma unroll
    for (int p = 0; p < npass; ++p) {
      const int j = p * 64 + lane;
      const f32x4 ga = reinterpret_cast<const f32x4*>(d.gamma)[j], be = reinterpret_cast<const f32x4*>(d.beta)[j];
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (y[p][e] - mean) * rstd * ga[e] + be[e];
      reinterpret_cast<f32x4*>(d.out + base)[j] = o;
      if (d.out2) {
        const f32x4 g2 = reinterpret_cast<const f32x4*>(d.gamma2)[j], b2 = reinterpret_cast<const f32x4*>(d.beta2)[j];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (y[p][e] - mean) * rstd * g2[e] + b2[e];
        reinterpret_cast<f32x4*>(d.out2 + base)[j] = o;
      }
    }
  }
}

// dy_total = d_y + LN-backward(d_out [, d_out2]);  d_x = dy_total;  d_r = dy_total * mask * scale
// A wave owns RPW consecutive rows and issues ALL their loads before the first reduction (one memory latency per
// workgroup instead of one per row: the kernel is latency-, not bandwidth-bound at these sizes).
template <int NPASS>
__global__ __launch_bounds__(kLnThreadsBwd) void add_ln_bwd_kernel(vdetr_addln_desc d, vdetr_addln_grads g) {
  constexpr int kWaves = kLnThreadsBwd / 64, RPW = kLnRowsPerWgBwd / kWaves;
  __shared__ float red[kWaves][NPASS * 256];  // per-wave partial of one of the four parameter gradients at a time
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  constexpr int C = NPASS * 256;
  const LnRng rg = ln_rng(d);
  const float invC = 1.f / (float)C;
  const float* ysrc = d.r ? d.y : d.x;  // the tensor that was normalised
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  f32x4 dga[NPASS], dbe[NPASS], dga2[NPASS], dbe2[NPASS], ga[NPASS], ga2[NPASS];
#pragma unroll
  for (int p = 0; p < NPASS; ++p) {
    dga[p] = dbe[p] = dga2[p] = dbe2[p] = zero;
    ga[p] = reinterpret_cast<const f32x4*>(d.gamma)[p * 64 + lane];
    ga2[p] = g.d_out2 ? reinterpret_cast<const f32x4*>(d.gamma2)[p * 64 + lane] : zero;
  }
  const int row0 = blockIdx.x * kLnRowsPerWgBwd + wv * RPW;
  f32x4 yv[RPW][NPASS], go[RPW][NPASS], go2[RPW][NPASS], dy[RPW][NPASS];
  float mean[RPW], rstd[RPW];
#pragma unroll
  for (int rr = 0; rr < RPW; ++rr) {
    const int row = min(row0 + rr, d.rows - 1);  // clamped rows are computed and discarded
    const size_t base = (size_t)row * C;
    mean[rr] = d.mean[row]; rstd[rr] = d.rstd[row];
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
      const int j = p * 64 + lane;
      yv[rr][p] = reinterpret_cast<const f32x4*>(ysrc + base)[j];
      go[rr][p] = g.d_out ? reinterpret_cast<const f32x4*>(g.d_out + base)[j] : zero;
      go2[rr][p] = g.d_out2 ? reinterpret_cast<const f32x4*>(g.d_out2 + base)[j] : zero;
      dy[rr][p] = g.d_y ? reinterpret_cast<const f32x4*>(g.d_y + base)[j] : zero;
    }
  }
#pragma unroll
  for (int rr = 0; rr < RPW; ++rr) {
    const int row = row0 + rr;
    const bool live = row < d.rows;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int p = 0; p < NPASS; ++p)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float h = (yv[rr][p][e] - mean[rr]) * rstd[rr];
        const float o1 = live ? go[rr][p][e] : 0.f, o2 = live ? go2[rr][p][e] : 0.f;
        dga[p][e] += o1 * h; dbe[p][e] += o1;
        dga2[p][e] += o2 * h; dbe2[p][e] += o2;
        const float t = o1 * ga[p][e] + o2 * ga2[p][e];
        yv[rr][p][e] = h;   // x_hat
        go[rr][p][e] = t;   // d(x_hat)
        s1 += t; s2 += t * h;
      }
    const float m1 = wave_allsum_f32(s1) * invC, m2 = wave_allsum_f32(s2) * invC;
    if (!live) continue;
    const size_t base = (size_t)row * C;
    const unsigned rowkey = ln_rowkey(rg, row);
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
      const int j = p * 64 + lane;
      f32x4 t = dy[rr][p];
#pragma unroll
      for (int e = 0; e < 4; ++e) t[e] += rstd[rr] * (go[rr][p][e] - m1 - yv[rr][p][e] * m2);
      reinterpret_cast<f32x4*>(g.d_x + base)[j] = t;
      if (g.d_r) {
        bool keep[4];
        ln_keep4(rg, rowkey, j, keep);
#pragma unroll
        for (int e = 0; e < 4; ++e) t[e] = keep[e] ? t[e] * rg.scale : 0.f;
        reinterpret_cast<f32x4*>(g.d_r + base)[j] = t;
      }
    }
  }
  // ---- parameter gradients: waves -> LDS -> this workgroup's partial row (summed by add_ln_param_reduce_kernel) ----
  const int nparts = gridDim.x;
  float* part = g.partials + (size_t)blockIdx.x * 4 * C;
  const int nwhich = g.d_out2 ? 4 : 2;
  for (int which = 0; which < nwhich; ++which) {
    __syncthreads();
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
      const f32x4 v = which == 0 ? dga[p] : (which == 1 ? dbe[p] : (which == 2 ? dga2[p] : dbe2[p]));
      reinterpret_cast<f32x4*>(red[wv])[p * 64 + lane] = v;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += kLnThreadsBwd) {
      float acc = 0.f;
#pragma unroll
      for (int w2 = 0; w2 < kWaves; ++w2) acc += red[w2][c];
      part[which * C + c] = acc;
    }
  }
}

// Second (tiny) launch: fixed-order sum of the workgroups' partial rows.  A "last workgroup reduces" single-launch
// variant was measured slower (17-30 us): its device-scope release/acquire fences write back and invalidate whole L2s
// on a multi-XCD part, a kernel boundary is cheaper.
__device__ __forceinline__ void ln_param_reduce_body(const float* __restrict__ partials, int nparts, int C, int which, int cblock,
                                                     float* dst) {
  // block = 64 channels x 4 groups of partial rows; each thread keeps up to 16 independent loads in flight (the sum is
  // latency-bound: 64 partial rows of 4 KB)
  __shared__ float comb[4][64];
  const int c = cblock * 64 + (threadIdx.x & 63), pg = threadIdx.x >> 6;
  const float* src = partials + (size_t)which * C + c;
  const size_t pitch = (size_t)4 * C;
  float acc[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) acc[u] = 0.f;
  for (int p0 = pg; p0 < nparts; p0 += 64) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int p = p0 + 4 * u;
      if (p < nparts) acc[u] += src[(size_t)p * pitch];
    }
  }
  float t = 0.f;
#pragma unroll
  for (int u = 0; u < 16; ++u) t += acc[u];
  comb[pg][threadIdx.x & 63] = t;
  __syncthreads();
  if (pg == 0) dst[c] = (comb[0][threadIdx.x] + comb[1][threadIdx.x]) + (comb[2][threadIdx.x] + comb[3][threadIdx.x]);
}

__global__ __launch_bounds__(256) void add_ln_param_reduce_kernel(const float* __restrict__ partials, int nparts, int C,
                                                                 float* d_gamma, float* d_beta, float* d_gamma2, float* d_beta2) {
  const int which = blockIdx.y;
  ln_param_reduce_body(partials, nparts, C, which, blockIdx.x,
                       which == 0 ? d_gamma : (which == 1 ? d_beta : (which == 2 ? d_gamma2 : d_beta2)));
}

// the same sums for SEVERAL LayerNorm backward passes in one launch (descriptors by value, blockIdx.z = item): the per-layer
// reductions of a decoder are 27 launches of ~7 us that nothing in the backward waits for
constexpr int kLnReduceBatch = 32;
struct LnReduceBatch {
  vdetr_addln_reduce item[kLnReduceBatch];
};
__global__ __launch_bounds__(256) void add_ln_param_reduce_batch_kernel(LnReduceBatch b) {
  const vdetr_addln_reduce& it = b.item[blockIdx.z];
  const int which = blockIdx.y;
  if ((int)blockIdx.x * 64 >= it.C) return;
  float* dst = which == 0 ? it.d_gamma : (which == 1 ? it.d_beta : (which == 2 ? it.d_gamma2 : it.d_beta2));
  if (dst == nullptr) return;
  ln_param_reduce_body(it.partials, it.nparts, it.C, which, blockIdx.x, dst);
}

}  // namespace vdetr

using namespace vdetr;

static int addln_check(const vdetr_addln_desc* d, const char* op) {
  VDETR_REQUIRE(d != nullptr, "%s: null descriptor", op);
  VDETR_REQUIRE(d->rows > 0 && d->C > 0, "%s: empty tensor rows=%d C=%d", op, d->rows, d->C);
  VDETR_REQUIRE(d->C % 256 == 0 && d->C <= kLnMaxC, "%s: C=%d must be a multiple of 256, at most %d", op, d->C, kLnMaxC);
  VDETR_REQUIRE(d->x && d->gamma && d->beta && d->mean && d->rstd, "%s: null pointer", op);
  VDETR_REQUIRE(!d->r || d->y, "%s: y is required when a residual branch r is given", op);
  VDETR_REQUIRE((d->gamma2 == nullptr) == (d->beta2 == nullptr), "%s: gamma2 and beta2 go together", op);
  VDETR_REQUIRE(d->dropout_p >= 0.f && d->dropout_p < 1.f, "%s: dropout_p %f outside [0,1)", op, d->dropout_p);
  return VDETR_OK;
}

extern "C" size_t vdetr_add_ln_bwd_workspace_bytes(const vdetr_addln_desc* d) {
  if (!d || d->rows <= 0 || d->C <= 0) return 0;
  return (size_t)ceil_div(d->rows, kLnRowsPerWgBwd) * 4 * d->C * sizeof(float);
}

extern "C" int vdetr_add_ln_fwd_f32(const vdetr_addln_desc* d, vdetr_stream_t stream) {
  if (int e = addln_check(d, "add_ln_fwd")) return e;
  VDETR_REQUIRE(d->out, "add_ln_fwd: null output");
  VDETR_REQUIRE(!d->out2 || d->gamma2, "add_ln_fwd: out2 needs gamma2 / beta2");
  const dim3 grid(ceil_div(d->rows, kLnRowsPerWg)), block(kLnThreads);
  hipStream_t st = (hipStream_t)stream;
  switch (d->C / 256) {
    case 1: hipLaunchKernelGGL(add_ln_fwd_kernel<1>, grid, block, 0, st, *d); break;
    case 2: hipLaunchKernelGGL(add_ln_fwd_kernel<2>, grid, block, 0, st, *d); break;
    case 3: hipLaunchKernelGGL(add_ln_fwd_kernel<3>, grid, block, 0, st, *d); break;
    default: hipLaunchKernelGGL(add_ln_fwd_kernel<4>, grid, block, 0, st, *d); break;
  }
  return check_launch("add_ln_fwd");
}

extern "C" int vdetr_add_ln_bwd_f32(const vdetr_addln_desc* d, const vdetr_addln_grads* g, vdetr_stream_t stream) {
  if (int e = addln_check(d, "add_ln_bwd")) return e;
  VDETR_REQUIRE(g && g->d_x && g->partials && (g->d_gamma == nullptr) == (g->d_beta == nullptr), "add_ln_bwd: null pointer");
  VDETR_REQUIRE(g->d_out || g->d_out2 || g->d_y, "add_ln_bwd: no incoming gradient");
  VDETR_REQUIRE(!g->d_out2 || (d->gamma2 && (g->d_gamma == nullptr || (g->d_gamma2 && g->d_beta2))),
                "add_ln_bwd: d_out2 needs gamma2 and its gradient buffers");
  VDETR_REQUIRE(!g->d_r || d->r, "add_ln_bwd: d_r without a residual branch");
  const dim3 grid(ceil_div(d->rows, kLnRowsPerWgBwd)), block(kLnThreadsBwd);
  hipStream_t st = (hipStream_t)stream;
  switch (d->C / 256) {
    case 1: hipLaunchKernelGGL(add_ln_bwd_kernel<1>, grid, block, 0, st, *d, *g); break;
    case 2: hipLaunchKernelGGL(add_ln_bwd_kernel<2>, grid, block, 0, st, *d, *g); break;
    case 3: hipLaunchKernelGGL(add_ln_bwd_kernel<3>, grid, block, 0, st, *d, *g); break;
    default: hipLaunchKernelGGL(add_ln_bwd_kernel<4>, grid, block, 0, st, *d, *g); break;
  }
  if (int e = check_launch("add_ln_bwd")) return e;
  if (g->d_gamma == nullptr) return VDETR_OK;  // parameter sums left to vdetr_add_ln_param_reduce_batch_f32
  hipLaunchKernelGGL(add_ln_param_reduce_kernel, dim3(d->C / 64, g->d_out2 ? 4 : 2), dim3(256), 0, st, g->partials, (int)grid.x, d->C,
                     g->d_gamma, g->d_beta, g->d_gamma2, g->d_beta2);
  return check_launch("add_ln_param_reduce");
}

extern "C" int vdetr_add_ln_param_reduce_batch_f32(const vdetr_addln_reduce* items, int n, vdetr_stream_t stream) {
  VDETR_REQUIRE(items != nullptr && n >= 1, "add_ln_param_reduce_batch: null items");
  for (int i0 = 0; i0 < n; i0 += kLnReduceBatch) {
    const int cnt = n - i0 < kLnReduceBatch ? n - i0 : kLnReduceBatch;
    LnReduceBatch b{};
    int maxc = 0;
    for (int k = 0; k < cnt; ++k) {
      const vdetr_addln_reduce& it = items[i0 + k];
      VDETR_REQUIRE(it.partials && it.d_gamma && it.d_beta && it.nparts >= 1 && it.C % 256 == 0 && it.C > 0 && it.C <= kLnMaxC,
                    "add_ln_param_reduce_batch: bad item %d", i0 + k);
      VDETR_REQUIRE((it.d_gamma2 == nullptr) == (it.d_beta2 == nullptr), "add_ln_param_reduce_batch: item %d: one of the *2 outputs", i0 + k);
      b.item[k] = it;
      maxc = it.C > maxc ? it.C : maxc;
    }
    hipLaunchKernelGGL(add_ln_param_reduce_batch_kernel, dim3(maxc / 64, 4, cnt), dim3(256), 0, (hipStream_t)stream, b);
    if (int e = check_launch("add_ln_param_reduce_batch")) return e;
  }
  return VDETR_OK;
}
