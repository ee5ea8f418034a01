// rowblock_pos.hip — rb_qkv with the learned query position computed on the way in (vdetr_rb_qkv_pos_f32).
// Its own translation unit: the position MLP's scalar first convolution must be compiled without SLP vectorisation (build.py NO_SLP:
// the packed-fma op_sel form the vectoriser makes of it), which rowblock.hip's epilogues want to keep.
#include "rowblock.h"

namespace vdetr {

// ---- rb_qkv_pos_kernel: rb_qkv_kernel whose `pos` rows are computed on the way in (round 6) -------------------------------------
// The learned query position (PositionEmbeddingLearned, helpers.py:17-33; heads.hip: pos_mlp_kernel) was a launch of its own in front
// of every decoder layer: 15 us on the forward chain for a [16 x 256] x [256 x 256] product per workgroup behind a BatchNorm whose batch
// statistics every workgroup derives itself from the coordinates' mean and covariance (heads.hip).  Its tiles are the row blocks of
// rb_qkv: the q and k workgroups (blockIdx.y = 0, 1) now form their own 16 rows of it — same arithmetic, same order as pos_mlp_kernel —,
// add them to the rows of norm1(tgt) in LDS and go on as before; the q workgroups write everything the position MLP's launch wrote
// (pos, the hidden pre-/activations, the saved statistics; workgroup (0, 0) the running statistics).  The v workgroups are rb_qkv's.
template <int CIN>
__global__ __launch_bounds__(kRbThreads) void rb_qkv_pos_kernel(RbQkvArgs A, vdetr_posmlp_desc M) {
  constexpr int kMom = CIN + CIN * (CIN + 1) / 2;  // sums + upper triangle of the second moments
  __shared__ __attribute__((aligned(16))) float xs[kRbRows * kRbStride];
  __shared__ float momq[kMom][64];
  __shared__ float mom[kMom];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, c = lane & 15;
  const int row0 = blockIdx.x * kRbRows, which = blockIdx.y;
  const int col0 = 64 * w, colq = col0 + 4 * c;
  const float* Wt = A.wt + (size_t)which * kRbC * kRbC;
  const f32x4 bias = rb_ldv(A.b ? A.b + which * kRbC : nullptr, colq);
  float a[64];
  f32x4 acc[4];
  RbRing R;
  if (which == 2) {  // v = norm1(tgt) Wv^T + bv: no position
    rb_w_begin(Wt, col0, lane, R);
    __builtin_amdgcn_sched_barrier(0);
    rb_stage_rows(A.t, row0, A.rows, A.B, false, xs, tid);
    __syncthreads();
  } else {
    rb_w_begin(M.w2t, col0, lane, R);
    const int T = M.B * M.N, B = M.B;
    // ---- this thread's channel of the first convolution and BatchNorm; the workgroup's 16 tokens (wave-uniform: scalar loads) ----
    const int ch = tid;
    float wv[CIN], xt[kRbRows][CIN], ref[CIN];
#pragma unroll
    for (int i = 0; i < CIN; ++i) wv[i] = M.w1[(size_t)ch * CIN + i];
    const float gam = M.gamma[ch], bet = M.beta[ch], bias1 = M.b1 ? M.b1[ch] : 0.f;
    const f32x4 bias2 = rb_ldv(M.b2, colq);
#pragma unroll
    for (int i = 0; i < CIN; ++i) ref[i] = M.x[i];
#pragma unroll
    for (int t = 0; t < kRbRows; ++t) {
      const int row = row0 + t, q = row / B, b = row - q * B;
#pragma unroll
      for (int i = 0; i < CIN; ++i) xt[t][i] = M.x[((size_t)b * M.N + q) * CIN + i];
    }
    // ---- the coordinates' first and second moments over all B N tokens (heads.hip: pos_mlp_kernel, phase 1) ----
    double sm[kMom];
#pragma unroll
    for (int i = 0; i < kMom; ++i) sm[i] = 0.0;
    for (int t = tid; t < T; t += kRbThreads) {
      float xv[CIN];
#pragma unroll
      for (int i = 0; i < CIN; ++i) xv[i] = M.x[(size_t)t * CIN + i];
      double dv[CIN];
#pragma unroll
      for (int i = 0; i < CIN; ++i) dv[i] = (double)(xv[i] - ref[i]);
      int at = CIN;
#pragma unroll
      for (int i = 0; i < CIN; ++i) {
        sm[i] += dv[i];
#pragma unroll
        for (int j = i; j < CIN; ++j) sm[at++] += dv[i] * dv[j];
      }
    }
#pragma unroll
    for (int i = 0; i < kMom; ++i) {
      float r = (float)sm[i];
      r += dpp_f32<kDppQuadXor1>(r);
      r += dpp_f32<kDppQuadXor2>(r);
      if ((lane & 3) == 0) momq[i][tid >> 2] = r;
    }
    __syncthreads();
    for (int i = w; i < kMom; i += 4) {
      const float r = wave_allsum_f32(momq[i][lane]);
      if (lane == 0) mom[i] = r;
    }
    __syncthreads();
    // ---- BatchNorm of this thread's channel, relu(bn(W1 x)) of the 16 tokens into the tile (phase 2) ----
    {
      const double inv = 1.0 / (double)T;
      double mu[CIN];
#pragma unroll
      for (int i = 0; i < CIN; ++i) mu[i] = (double)mom[i] * inv;
      double mean = 0.0, var = 0.0;
      int at = CIN;
#pragma unroll
      for (int i = 0; i < CIN; ++i) {
        mean += (double)wv[i] * (mu[i] + (double)ref[i]);
#pragma unroll
        for (int j = i; j < CIN; ++j) {
          const double cov = (double)mom[at] * inv - mu[i] * mu[j];
          var += (i == j ? 1.0 : 2.0) * (double)wv[i] * (double)wv[j] * cov;
          ++at;
        }
      }
      var = var > 0.0 ? var : 0.0;
      const float meanf = (float)mean, varf = (float)var;
      const float invstd = rsqrtf(varf + M.eps);
      const float sc = gam * invstd, sh = bet - meanf * sc;
      if (blockIdx.x == 0 && which == 0) {
        M.save_mean[ch] = meanf;
        M.save_invstd[ch] = invstd;
        if (M.running_mean) {
          const float m = M.momentum, n = (float)T;
          M.running_mean[ch] = (1.f - m) * M.running_mean[ch] + m * (meanf + bias1);
          M.running_var[ch] = (1.f - m) * M.running_var[ch] + m * varf * (n / (n > 1.f ? n - 1.f : 1.f));
        }
        if (tid == 0 && M.counter) M.counter[0] += 1;
      }
      float hp[kRbRows], ha[kRbRows];
#pragma unroll
      for (int i = 0; i < kRbRows; ++i) {
        float h = 0.f;
#pragma unroll
        for (int k = 0; k < CIN; ++k) h = fmaf(xt[i][k], wv[k], h);
        const float act = fmaxf(h * sc + sh, 0.f);
        hp[i] = h;
        ha[i] = act;
        xs[i * kRbStride + ch] = act;
      }
      if (which == 0) {
        if (B == 1) {  // the block's tokens are 16 consecutive ones of the channel's row
          const size_t o = (size_t)ch * M.N + row0;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<f32x4*>(M.hpre + o + 4 * i) = f32x4{hp[4 * i], hp[4 * i + 1], hp[4 * i + 2], hp[4 * i + 3]};
            *reinterpret_cast<f32x4*>(M.hact + o + 4 * i) = f32x4{ha[4 * i], ha[4 * i + 1], ha[4 * i + 2], ha[4 * i + 3]};
          }
        } else {
#pragma unroll
          for (int i = 0; i < kRbRows; ++i) {
            const int row = row0 + i, q = row / B, b = row - q * B;
            const size_t o = ((size_t)b * kRbC + ch) * M.N + q;
            M.hpre[o] = hp[i];
            M.hact[o] = ha[i];
          }
        }
      }
    }
    __syncthreads();
    // ---- pos = act W2^T + b2 (phase 3), then x = norm1(tgt) + pos in the tile ----
    rb_load_a(xs, lane, a);
    __syncthreads();  // every wave has its operand: the tile is free for the rows of norm1(tgt)
    rb_stage_rows(A.t, row0, A.rows, A.B, false, xs, tid);
    rb_zero(acc);
    rb_w_run(a, M.w2t, col0, lane, R, acc);
    rb_w_begin(Wt, col0, lane, R);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = row0 + 4 * g + r;
      const f32x4 pv = rb_row(acc, r) + bias2;
      float* cell = xs + (4 * g + r) * kRbStride + colq;
      const f32x4 xv = *reinterpret_cast<const f32x4*>(cell) + pv;
      *reinterpret_cast<f32x4*>(cell) = xv;
      if (which == 0) {
        rb_st4(M.out, row, colq, pv);
        rb_st4(A.x, row, colq, xv);
      }
    }
    __syncthreads();
  }
  rb_load_a(xs, lane, a);
  rb_zero(acc);
  rb_w_run(a, Wt, col0, lane, R, acc);
  float* out = A.out + (size_t)which * A.rows * kRbC;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = row0 + 4 * g + r;
    rb_st4(out, rb_bmajor(row, A.B, A.rows / A.B), colq, rb_row(acc, r) + bias);
  }
}

}  // namespace vdetr

using namespace vdetr;

static int rb_common(int rows, int B, const char* op) {
  VDETR_REQUIRE(rows > 0 && B > 0 && rows % B == 0, "%s: rows=%d must be a positive multiple of B=%d", op, rows, B);
  return VDETR_OK;
}

extern "C" int vdetr_rb_qkv_pos_f32(const vdetr_rb_qkv_desc* d, const vdetr_posmlp_desc* m, vdetr_stream_t stream) {
  VDETR_REQUIRE(d != nullptr && m != nullptr, "rb_qkv_pos: null descriptor");
  if (int e = rb_common(d->rows, d->B, "rb_qkv_pos")) return e;
  VDETR_REQUIRE(d->t && d->wt && d->out && d->x, "rb_qkv_pos: null pointer (wt: the W^T images, vdetr_rb_transpose_f32; x: t + pos, written)");
  VDETR_REQUIRE(m->B == d->B && m->N > 0 && (long)m->B * m->N == d->rows && d->rows % kRbRows == 0,
                "rb_qkv_pos: %d x %d tokens for %d rows (a multiple of %d)", m->B, m->N, d->rows, kRbRows);
  VDETR_REQUIRE(m->B == 1 || m->N % 4 == 0, "rb_qkv_pos: N=%d", m->N);
  VDETR_REQUIRE(m->cin >= 1 && m->cin <= 8, "rb_qkv_pos: cin=%d outside [1, 8]", m->cin);
  VDETR_REQUIRE(m->x && m->w1 && m->gamma && m->beta && m->w2t && m->hpre && m->hact && m->save_mean && m->save_invstd && m->out,
                "rb_qkv_pos: null operand of the position MLP");
  VDETR_REQUIRE((m->running_mean != nullptr) == (m->running_var != nullptr), "rb_qkv_pos: running_mean and running_var go together");
  VDETR_REQUIRE(d->pos == nullptr || d->pos == m->out, "rb_qkv_pos: pos is computed by the launch (pass NULL or the position MLP's `out`)");
  VDETR_REQUIRE(RB_ALIGNED(d->t) && RB_ALIGNED(d->wt) && RB_ALIGNED(d->b) && RB_ALIGNED(d->x) && RB_ALIGNED(d->out) && RB_ALIGNED(m->w2t) &&
                RB_ALIGNED(m->b2) && RB_ALIGNED(m->hpre) && RB_ALIGNED(m->hact) && RB_ALIGNED(m->out), "rb_qkv_pos: operands must be 16-B aligned");
  const dim3 grid(d->rows / kRbRows, 3), block(kRbThreads);
  hipStream_t st = (hipStream_t)stream;
  switch (m->cin) {
    case 1: hipLaunchKernelGGL(rb_qkv_pos_kernel<1>, grid, block, 0, st, *d, *m); break;
    case 2: hipLaunchKernelGGL(rb_qkv_pos_kernel<2>, grid, block, 0, st, *d, *m); break;
    case 3: hipLaunchKernelGGL(rb_qkv_pos_kernel<3>, grid, block, 0, st, *d, *m); break;
    case 4: hipLaunchKernelGGL(rb_qkv_pos_kernel<4>, grid, block, 0, st, *d, *m); break;
    case 5: hipLaunchKernelGGL(rb_qkv_pos_kernel<5>, grid, block, 0, st, *d, *m); break;
    case 6: hipLaunchKernelGGL(rb_qkv_pos_kernel<6>, grid, block, 0, st, *d, *m); break;  // (box centre + size: the decoder's query position)
    case 7: hipLaunchKernelGGL(rb_qkv_pos_kernel<7>, grid, block, 0, st, *d, *m); break;
    default: hipLaunchKernelGGL(rb_qkv_pos_kernel<8>, grid, block, 0, st, *d, *m); break;
  }
  return check_launch("rb_qkv_pos");
}

