// attn_bwd.hip — backward "score stage" of the attention + the 3DV-RPE table gradient, for gfx950.
//
// Backward of out = dropout(softmax(S)) V with S = scale*q k^T + rpe + mask is split as
//     dP~ = dO V^T                      (plain GEMM, library)
//     P~  = dropout(softmax(S)),  dS = softmax(S) * (dropout'(dP~) - rowsum(dO*O))     <- THIS FILE
//     dV = P~^T dO,  dK = dS^T q,  dQ = dS K      (plain GEMMs, library)
//     dTable[i,cell,h] += trilinear_weight(pair, cell) * dS[h, pair]                  <- THIS FILE
// The forward saved S (the biased scores) and the row log-sum-exp, so nothing of the QK^T / table lookup
// is recomputed here except the per-pair lookup GEOMETRY (cell + 3 fractions per vertex), which is cheaper
// to recompute than to store (16 B x 8 vertices x nQ x nK).
// The table gradient is a 32,000-bin weighted histogram with 8*8*4*nQ*nK contributions
// (vdetr_transformer.py:725-731 backward = grid_sampler_3d_backward, 47 % of the reference layer time on
// CPU).  Each workgroup keeps a private copy of the histogram in LDS (128 KB) and adds to it with
// ds_add_f32; the copies are written to a workspace and summed by a second, coalesced kernel —
// no global atomics.
#include "attn_common.h"

namespace vdetr {

int attn_fill_params(const vdetr_attn_desc* d, AttnParams* P, const char* op);

constexpr int kBwdThreads = 1024;

// element-wise part shared by both kernels: returns P~ and dS for one head of one (q,key) pair
struct ScoreGrad {
  float p_drop, ds;
};
__device__ __forceinline__ ScoreGrad score_grad(float s, float lse, bool keep, float drop_scale, bool have_grad,
                                                float dprob, float delta, bool masked) {
  const float p = __expf(s - lse);
  ScoreGrad r;
  r.p_drop = keep ? p * drop_scale : 0.f;
  float ds = 0.f;
  if (have_grad) {
    const float dp = keep ? dprob * drop_scale : 0.f;
    ds = p * (dp - delta);
    if (masked) ds = 0.f;  // masked_fill_ overwrote the score: no gradient reaches q, k or the table
  }
  r.ds = ds;
  return r;
}

// ---- generic kernel (no RPE): one thread per score element ----------------------------------------------
__global__ __launch_bounds__(256) void attn_bwd_scores_kernel(AttnParams P) {
  attn_load_rng(P);
  const size_t total = (size_t)P.B * P.H * P.nQ * P.nK;
  const bool perhead = P.kind == VDETR_ATTN_PER_HEAD;
  const bool have_grad = P.dprob != nullptr;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const size_t row = e / P.nK;
    const int key = (int)(e - row * P.nK);
    int b, q, h;
    if (perhead) {  // [B,H,nQ,nK]
      q = (int)(row % P.nQ);
      h = (int)((row / P.nQ) % P.H);
      b = (int)(row / ((size_t)P.nQ * P.H));
    } else {  // [B,nQ,H,nK]
      h = (int)(row % P.H);
      q = (int)((row / P.H) % P.nQ);
      b = (int)(row / ((size_t)P.nQ * P.H));
    }
    bool keep = true;
    if (P.drop_thresh) keep = pick4(attn_rand4(P, b, q, key, h >> 2), h & 3) >= P.drop_thresh;
    const bool masked = P.mask_kind == VDETR_MASK_BOOL &&
                        reinterpret_cast<const unsigned char*>(P.mask)[((size_t)b * P.nQ + q) * P.nK + key];
    const ScoreGrad g = score_grad(P.scores[e], P.lse[row], keep, P.drop_scale, have_grad,
                                   have_grad ? P.dprob[e] : 0.f, have_grad ? P.delta[row] : 0.f, masked);
    P.scores[e] = g.p_drop;
    if (have_grad) P.dprob[e] = g.ds;
  }
}

// ---- RPE kernel: one thread per (query, key) pair, 4 heads in registers -----------------------------------
template <int VARIANT>
__global__ __launch_bounds__(kBwdThreads) void attn_bwd_scores_rpe_kernel(AttnParams P) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // dTable copy [8][T^3][4]
  attn_load_rng(P);
  const int tid = threadIdx.x;
  const int T = P.T, TT = T * T, T3 = TT * T;
  const int table_floats = kRpeVerts * T3 * 4;
  const bool want_table = P.dtable_part != nullptr;
  if (want_table) {
    for (int i = tid; i < table_floats; i += kBwdThreads) smem[i] = 0.f;
    __syncthreads();
  }
  const bool have_grad = P.dprob != nullptr;
  const bool rot = P.cos_sin != nullptr;
  const int items = P.B * P.nQ;
  for (int item = blockIdx.x; item < items; item += gridDim.x) {
    const int b = item / P.nQ, q = item - b * P.nQ;
    const size_t row0 = ((size_t)b * P.nQ + q) * 4;  // rows (b,q,h) for h = 0..3
    float lse[4], delta[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) { lse[h] = P.lse[row0 + h]; delta[h] = have_grad ? P.delta[row0 + h] : 0.f; }
    float vx[8], vy[8], vz[8];
    const float* vp = P.vertices + ((size_t)b * P.nQ + q) * 24;
#pragma unroll
    for (int i = 0; i < 8; ++i) { vx[i] = vp[i * 3]; vy[i] = vp[i * 3 + 1]; vz[i] = vp[i * 3 + 2]; }
    const float rc = rot ? P.cos_sin[((size_t)b * P.nQ + q) * 2] : 1.f;
    const float rs = rot ? P.cos_sin[((size_t)b * P.nQ + q) * 2 + 1] : 0.f;

    for (int key = tid; key < P.nK; key += kBwdThreads) {
      uint4 rnd = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
      if (P.drop_thresh) rnd = attn_rand4(P, b, q, key, 0);
      const bool masked = P.mask_kind == VDETR_MASK_BOOL &&
                          reinterpret_cast<const unsigned char*>(P.mask)[((size_t)b * P.nQ + q) * P.nK + key];
      float ds[4];
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        const size_t e = (row0 + h) * P.nK + key;
        const bool keep = pick4(rnd, h) >= P.drop_thresh;
        const ScoreGrad g = score_grad(P.scores[e], lse[h], keep, P.drop_scale, have_grad,
                                       have_grad ? P.dprob[e] : 0.f, delta[h], masked);
        P.scores[e] = g.p_drop;
        if (have_grad) P.dprob[e] = g.ds;
        ds[h] = g.ds;
      }
      if (!want_table) continue;
      const float* xp = P.xyz + ((size_t)b * P.nK + key) * 3;
      const float kx = xp[0], ky = xp[1], kz = xp[2];
#pragma unroll
      for (int i = 0; i < kRpeVerts; ++i) {
        float dx = vx[i] - kx, dy = vy[i] - ky, dz = vz[i] - kz;
        if (rot) rpe_rotate(dx, dy, rc, rs);
        const AxisTap ax = rpe_axis(dx, P), ay = rpe_axis(dy, P), az = rpe_axis(dz, P);
        float* t = smem + (size_t)(i * T3 + rpe_cell(ax, ay, az, T)) * 4;
        const float wz[2] = {az.wa, az.wb}, wy[2] = {ay.wa, ay.wb}, wx[2] = {ax.wa, ax.wb};
#pragma unroll
        for (int cz = 0; cz < 2; ++cz)
#pragma unroll
          for (int cy = 0; cy < 2; ++cy)
#pragma unroll
            for (int cx = 0; cx < 2; ++cx) {
              const float wgt = wz[cz] * wy[cy] * wx[cx];
              float* cell = t + (cz * TT + cy * T + cx) * 4;
#pragma unroll
              for (int h = 0; h < 4; ++h) atomicAdd(cell + h, wgt * ds[h]);
            }
      }
    }
  }
  if (want_table) {
    __syncthreads();
    float* dst = P.dtable_part + (size_t)blockIdx.x * table_floats;
    for (int i = tid; i < table_floats; i += kBwdThreads) dst[i] = smem[i];
  }
}

// dtable[e] += sum over workgroup copies
__global__ __launch_bounds__(256) void attn_bwd_table_reduce_kernel(const float* part, int nparts, int n, float* dtable) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  float s = 0.f;
  for (int p = 0; p < nparts; ++p) s += part[(size_t)p * n + e];
  dtable[e] += s;
}

// keep-mask dump (test hook)
__global__ __launch_bounds__(256) void attn_dropout_mask_kernel(AttnParams P, uint8_t* keep) {
  attn_load_rng(P);
  const size_t total = (size_t)P.B * P.H * P.nQ * P.nK;
  const bool perhead = P.kind == VDETR_ATTN_PER_HEAD;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const size_t row = e / P.nK;
    const int key = (int)(e - row * P.nK);
    int b, q, h;
    if (perhead) {
      q = (int)(row % P.nQ); h = (int)((row / P.nQ) % P.H); b = (int)(row / ((size_t)P.nQ * P.H));
    } else {
      h = (int)(row % P.H); q = (int)((row / P.H) % P.nQ); b = (int)(row / ((size_t)P.nQ * P.H));
    }
    keep[e] = (!P.drop_thresh || pick4(attn_rand4(P, b, q, key, h >> 2), h & 3) >= P.drop_thresh) ? 1 : 0;
  }
}

}  // namespace vdetr

using namespace vdetr;

static int bwd_grid(const vdetr_attn_desc* d) {
  const long items = (long)d->B * d->nQ;
  return (int)(items < 256 ? items : 256);
}

extern "C" size_t vdetr_attn_bwd_workspace_bytes(const vdetr_attn_desc* d) {
  if (!d || !d->table) return 0;
  const size_t table_floats = (size_t)kRpeVerts * d->table_size * d->table_size * d->table_size * 4;
  return (size_t)bwd_grid(d) * table_floats * sizeof(float) + 256;
}

extern "C" int vdetr_attn_bwd_scores_f32(const vdetr_attn_desc* d, float* scores, float* dprob, const float* lse,
                                         const float* delta, float* dtable, void* workspace, size_t workspace_bytes,
                                         vdetr_stream_t stream) {
  AttnParams P;
  if (int e = attn_fill_params(d, &P, "attn_bwd_scores")) return e;
  VDETR_REQUIRE(scores && lse, "attn_bwd_scores: null pointer");
  VDETR_REQUIRE((dprob == nullptr) == (delta == nullptr), "attn_bwd_scores: dprob and delta go together");
  VDETR_REQUIRE(!dtable || (d->table && dprob), "attn_bwd_scores: dtable needs an RPE descriptor and dprob");
  P.scores = scores; P.dprob = dprob; P.lse = const_cast<float*>(lse); P.delta = delta;
  hipStream_t st = (hipStream_t)stream;
  if (!d->table) {
    const size_t total = (size_t)d->B * d->H * d->nQ * d->nK;
    const unsigned grid = (unsigned)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(attn_bwd_scores_kernel, dim3(grid), dim3(256), 0, st, P);
    return check_launch("attn_bwd_scores");
  }
  const int grid = bwd_grid(d);
  const int table_floats = kRpeVerts * P.T * P.T * P.T * 4;
  size_t lds = 16;
  if (dtable) {
    const size_t need = vdetr_attn_bwd_workspace_bytes(d);
    if (!workspace || workspace_bytes < need) {
      set_error("attn_bwd_scores: workspace %zu B < required %zu B", workspace_bytes, need);
      return VDETR_ERR_WORKSPACE;
    }
    P.dtable_part = (float*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    lds = (size_t)table_floats * sizeof(float);
    if (int e = set_lds(attn_bwd_scores_rpe_kernel<0>, lds, "attn_bwd_scores")) return e;
  }
  hipLaunchKernelGGL((attn_bwd_scores_rpe_kernel<0>), dim3(grid), dim3(kBwdThreads), lds, st, P);
  if (int e = check_launch("attn_bwd_scores_rpe")) return e;
  if (dtable) {
    hipLaunchKernelGGL(attn_bwd_table_reduce_kernel, dim3((table_floats + 255) / 256), dim3(256), 0, st,
                       P.dtable_part, grid, table_floats, dtable);
    return check_launch("attn_bwd_table_reduce");
  }
  return VDETR_OK;
}

extern "C" int vdetr_attn_dropout_mask_u8(const vdetr_attn_desc* d, uint8_t* keep, vdetr_stream_t stream) {
  AttnParams P;
  if (int e = attn_fill_params(d, &P, "attn_dropout_mask")) return e;
  VDETR_REQUIRE(keep, "attn_dropout_mask: null pointer");
  const size_t total = (size_t)d->B * d->H * d->nQ * d->nK;
  const unsigned grid = (unsigned)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
  hipLaunchKernelGGL(attn_dropout_mask_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, P, keep);
  return check_launch("attn_dropout_mask");
}
