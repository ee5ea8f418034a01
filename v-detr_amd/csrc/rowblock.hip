// rowblock.hip — the decoder layer's glue between its attention kernels as THREE launches (round 5).
//
// Reference: GlobalDecoderLayer.forward_pre (models/vdetr_transformer.py:531-568).  Between the self-attention core, the
// cross-attention core and the next layer the reference (and rounds 1-4 of this library) run ~13 launches per layer on
// [nQ, 256] tensors of 1 MB: position adds, 8 projections of [nQ x 256] x [256 x 256], three dropout + residual + LayerNorm
// blocks, the FFN's relu + dropout.  Every one of them pays the ~4.5 us a dispatch costs whatever its size, the
// library GEMMs 7-9 us each.  A 16-row block of the activations is 16 KB: it fits a workgroup's LDS / registers for the whole
// chain, and each element of the chain is row-local (LayerNorm included).  So:
//   rb_qkv_kernel   x = tgt2 + pos;  q = x Wq^T + bq,  k = x Wk^T + bk,  v = tgt2 Wv^T + bv           (the self-attention's operands)
//   rb_proj_q_kernel   y = tgt + drop(a Wo^T + bo);  t2 = LN(y);  x = t2 + pos;  q = x Wq^T + bq         (behind the self-attention)
//   rb_ffn_kernel   y = tgt + drop(a Wp^T + bp);  t2 = LN(y);  h = drop(relu(t2 W1^T + b1));
//                   z = y + drop(h W2^T + b2);  o1 = LN(z; g1, b1') [, o2 = LN(z; g2, b2')]           (behind the cross-attention)
// One workgroup = 16 rows x all 256 columns, 4 waves; wave w owns columns 64 w .. 64 w + 63 as four 16-column MFMA tiles
// (v_mfma_f32_16x16x4_f32: exact fp32 products, an fmaf chain per output element — the library GEMMs' numerics class).
// MFMA column j of tile nt is output column 64 w + 4 j + nt, so that a lane's four accumulators of one row are four ADJACENT
// columns: bias, dropout quad hash (add_ln.hip / bn_act.hip key their masks by groups of 4 channels), residual, LayerNorm and
// the stores all work on float4.  The contraction index of MFMA step s in lane group kg is 16 (s >> 2) + 4 kg + (s & 3): each
// lane's share of the activation row comes out of LDS as float4s.
// The weights of a FORWARD product y = x W^T are read from a transposed image Wt[k][n] (vdetr_rb_transpose_f32, one launch per
// step for all layers): the four tiles' B operands of one contraction index are then ONE float4 of an image row, and the four
// lanes the texture unit serves per cycle read 64 contiguous bytes.  Out of nn.Linear's own [n][k] layout every lane of such a quad
// reads a different row: 64 cycles per load instruction instead of 16, 6.9 us per 256 x 256 product on one CU alone instead of
// the 4.2 us of its 256 matrix instructions (tools/probes/rb_gemm_probe.hip, profiles/r05_rb_gemm_probe.txt).  The backward's
// dX = dY W reads W as stored.
// Everything the existing backward kernels read (y, mean, rstd, t2, h, ...) is written exactly as the separate launches wrote
// it, with the same dropout streams: the autograd side (v-detr_amd/rowblock.py) reuses vdetr_add_ln_bwd_f32 /
// vdetr_relu_dropout_bwd_f32 and the parked weight gradients unchanged.
#include "rowblock.h"
#include "kv_pack.h"

namespace vdetr {

typedef vdetr_rb_linear RbLinear;
typedef vdetr_rb_norm RbNorm;
typedef vdetr_rb_drop RbDropArgs;

// ---- rb_ffn_kernel -------------------------------------------------------------------------------------------------------
// Order inside every kernel below: the first product's weights and everything the epilogues read from global memory (residual
// rows, biases, LayerNorm parameters) are REQUESTED first; the activation tile is staged behind them; each product's successor has
// its first weight steps requested before the epilogue in between runs.
typedef vdetr_rb_ffn_desc RbFfnArgs;

// PARTS: -1 = the rows of A.a; >= 0: the merge of key-split partials (rb_stage_parts<PARTS>);
// -2 = the FFN layer in front of the decoder (reference :585-606, FFNLayer.forward_pre): no attention branch — t2 = norm3(tgt) is BOTH the
//      FFN's input and its residual: z = t2 + drop3(lin2(drop(relu(lin1 t2)))), o1 = post1(z); A.a, A.proj, A.y are not used
//      (the step's first stage ran these 4096 tokens through five launches of 5-12 us forward and seven backward)
template <int PARTS>
__global__ __launch_bounds__(kRbThreads) void rb_ffn_kernel(RbFfnArgs A, RbParts Pp) {
  __shared__ __attribute__((aligned(16))) float xs[kRbRows * kRbStride];
  __shared__ float red[128];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, c = lane & 15;
  const int row0 = blockIdx.x * kRbRows;
  const int col0 = 64 * w, colq = col0 + 4 * c;  // this lane's four columns
  const RbDrop d2 = rb_drop(A.drop2.p, A.drop2.seed, 0, A.rng_state), da = rb_drop(A.drop_act.p, A.drop_act.seed, 0, A.rng_state),
               d3 = rb_drop(A.drop3.p, A.drop3.seed, 0, A.rng_state);
  float a[64];
  f32x4 acc[4];
  f32x4 y[4];  // the residual stream of this lane's rows 4 g + r, columns colq ..
  float mean[4], rstd[4];
  RbRing R;
  constexpr bool kFfn0 = PARTS == -2;
  rb_w_begin(kFfn0 ? A.lin1.wt : A.proj.wt, col0, lane, R);
  const bool two = A.post2.gamma != nullptr;
  const f32x4 bias_p = rb_ldv(kFfn0 ? nullptr : A.proj.b, colq), bias_1 = rb_ldv(A.lin1.b, colq), bias_2 = rb_ldv(A.lin2.b, colq);
  const f32x4 ga3 = rb_ldv(A.norm3.gamma, colq), be3 = rb_ldv(A.norm3.beta, colq);
  const f32x4 gap = rb_ldv(A.post1.gamma, colq), bep = rb_ldv(A.post1.beta, colq);
  const f32x4 gap2 = rb_ldv(A.post2.gamma, colq), bep2 = rb_ldv(A.post2.beta, colq);
  f32x4 tg[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) tg[r] = rb_ld4(A.tgt, min(row0 + 4 * g + r, A.rows - 1), colq);
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (kFfn0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) y[r] = tg[r];
  } else {
    if constexpr (PARTS < 0) rb_stage_rows(A.a, row0, A.rows, A.B, true, xs, tid);
    else rb_stage_parts<PARTS>(Pp, row0, A.rows, A.B, xs, tid);
    __syncthreads();
    rb_load_a(xs, lane, a);
    rb_zero(acc);
    rb_w_run(a, A.proj.wt, col0, lane, R, acc);
    rb_w_begin(A.lin1.wt, col0, lane, R);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = row0 + 4 * g + r, rowc = min(row, A.rows - 1);
      f32x4 v = rb_row(acc, r) + bias_p;
      bool keep[4];
      rb_keep4_ln(d2, rowc, colq >> 2, keep);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = tg[r][e] + (keep[e] ? v[e] * d2.scale : 0.f);
      y[r] = v;
      if (row < A.rows) rb_st4(A.y, row, colq, v);
    }
  }
  rb_row_stats(y, red, w, lane, A.norm3.eps, mean, rstd);  // (its barriers also fence the reads of xs above)
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = row0 + 4 * g + r;
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (y[r][e] - mean[r]) * rstd[r] * ga3[e] + be3[e];
    *reinterpret_cast<f32x4*>(xs + (4 * g + r) * kRbStride + colq) = o;
    if (row < A.rows) {
      rb_st4(A.t2, row, colq, o);
      if (w == 0 && c == 0) { A.mean_y[row] = mean[r]; A.rstd_y[row] = rstd[r]; }
    }
    if constexpr (kFfn0) y[r] = o;  // (the residual of the FFN is the NORMED input)
  }
  __syncthreads();
  rb_load_a(xs, lane, a);
  rb_zero(acc);
  rb_w_run(a, A.lin1.wt, col0, lane, R, acc);
  rb_w_begin(A.lin2.wt, col0, lane, R);
  __syncthreads();  // every wave has its A operand: the tile can be overwritten
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = row0 + 4 * g + r, rowc = min(row, A.rows - 1);
    f32x4 v = rb_row(acc, r) + bias_1;
    bool keep[4];
    rb_keep4_act(da, ((long)rowc * kRbC + colq) >> 2, keep);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (v[e] > 0.f && keep[e]) ? v[e] * da.scale : 0.f;
    *reinterpret_cast<f32x4*>(xs + (4 * g + r) * kRbStride + colq) = v;
    if (row < A.rows) rb_st4(A.h, row, colq, v);
  }
  __syncthreads();
  rb_load_a(xs, lane, a);
  rb_zero(acc);
  rb_w_run(a, A.lin2.wt, col0, lane, R, acc);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = row0 + 4 * g + r, rowc = min(row, A.rows - 1);
    f32x4 v = rb_row(acc, r) + bias_2;
    bool keep[4];
    rb_keep4_ln(d3, rowc, colq >> 2, keep);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = y[r][e] + (keep[e] ? v[e] * d3.scale : 0.f);
    y[r] = v;
    if (row < A.rows) rb_st4(A.z, row, colq, v);
  }
  rb_row_stats(y, red, w, lane, A.post1.eps, mean, rstd);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = row0 + 4 * g + r;
    if (row >= A.rows) continue;
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (y[r][e] - mean[r]) * rstd[r] * gap[e] + bep[e];
    rb_st4(A.o1, row, colq, o);
    if (two) {
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (y[r][e] - mean[r]) * rstd[r] * gap2[e] + bep2[e];
      rb_st4(A.o2, row, colq, o);
    }
    if (w == 0 && c == 0) { A.mean_z[row] = mean[r]; A.rstd_z[row] = rstd[r]; }
  }
}

// ---- rb_proj_q_kernel: out-projection of the self-attention + residual block 1 + the cross-attention's query projection ------
typedef vdetr_rb_projq_desc RbProjQArgs;

__global__ __launch_bounds__(kRbThreads) void rb_proj_q_kernel(RbProjQArgs A) {
  __shared__ __attribute__((aligned(16))) float xs[kRbRows * kRbStride];
  __shared__ float red[128];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, c = lane & 15;
  const int row0 = blockIdx.x * kRbRows;
  const int col0 = 64 * w, colq = col0 + 4 * c;
  const RbDrop d1 = rb_drop(A.drop1.p, A.drop1.seed, 0, A.rng_state);
  float a[64];
  f32x4 acc[4];
  f32x4 y[4];
  float mean[4], rstd[4];
  RbRing R;
  rb_w_begin(A.proj.wt, col0, lane, R);
  const f32x4 bias_p = rb_ldv(A.proj.b, colq), bias_q = rb_ldv(A.q.b, colq);
  const f32x4 ga = rb_ldv(A.norm2.gamma, colq), be = rb_ldv(A.norm2.beta, colq);
  f32x4 tg[4], ps[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int rowc = min(row0 + 4 * g + r, A.rows - 1);
    tg[r] = rb_ld4(A.tgt, rowc, colq);
    ps[r] = A.pos ? rb_ld4(A.pos, rowc, colq) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  __builtin_amdgcn_sched_barrier(0);
  rb_stage_rows(A.a, row0, A.rows, A.B, true, xs, tid);
  __syncthreads();
  rb_load_a(xs, lane, a);
  rb_zero(acc);
  rb_w_run(a, A.proj.wt, col0, lane, R, acc);
  rb_w_begin(A.q.wt, col0, lane, R);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = row0 + 4 * g + r, rowc = min(row, A.rows - 1);
    f32x4 v = rb_row(acc, r) + bias_p;
    bool keep[4];
    rb_keep4_ln(d1, rowc, colq >> 2, keep);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = tg[r][e] + (keep[e] ? v[e] * d1.scale : 0.f);
    y[r] = v;
    if (row < A.rows) rb_st4(A.y, row, colq, v);
  }
  rb_row_stats(y, red, w, lane, A.norm2.eps, mean, rstd);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = row0 + 4 * g + r;
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (y[r][e] - mean[r]) * rstd[r] * ga[e] + be[e];
    if (row < A.rows) {
      rb_st4(A.t2, row, colq, o);
      if (w == 0 && c == 0) { A.mean_y[row] = mean[r]; A.rstd_y[row] = rstd[r]; }
    }
    if (A.pos) {
      o += ps[r];
      if (row < A.rows) rb_st4(A.xq, row, colq, o);
    }
    *reinterpret_cast<f32x4*>(xs + (4 * g + r) * kRbStride + colq) = o;
  }
  __syncthreads();
  rb_load_a(xs, lane, a);
  rb_zero(acc);
  rb_w_run(a, A.q.wt, col0, lane, R, acc);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = row0 + 4 * g + r;
    if (row >= A.rows) continue;
    rb_st4(A.qout, rb_bmajor(row, A.B, A.rows / A.B), colq, rb_row(acc, r) + bias_q);
  }
}

// ---- rb_qkv_kernel: the self-attention's three projections; blockIdx.y = 0 / 1 / 2 = q / k / v ---------------------------------

__global__ __launch_bounds__(kRbThreads) void rb_qkv_kernel(RbQkvArgs A) {
  __shared__ __attribute__((aligned(16))) float xs[kRbRows * kRbStride];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, c = lane & 15;
  const int row0 = blockIdx.x * kRbRows, which = blockIdx.y;
  const int col0 = 64 * w, colq = col0 + 4 * c;
  const bool with_pos = which < 2 && A.pos != nullptr;
  const float* Wt = A.wt + (size_t)which * kRbC * kRbC;
  RbRing R;
  rb_w_begin(Wt, col0, lane, R);
  const f32x4 bias = rb_ldv(A.b ? A.b + which * kRbC : nullptr, colq);
  __builtin_amdgcn_sched_barrier(0);
  rb_stage_rows(A.t, row0, A.rows, A.B, false, xs, tid, with_pos ? A.pos : nullptr, which == 0 ? A.x : nullptr);
  __syncthreads();
  float a[64];
  rb_load_a(xs, lane, a);
  f32x4 acc[4];
  rb_zero(acc);
  rb_w_run(a, Wt, col0, lane, R, acc);
  float* out = A.out + (size_t)which * A.rows * kRbC;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = row0 + 4 * g + r;
    if (row >= A.rows) continue;
    rb_st4(out, rb_bmajor(row, A.B, A.rows / A.B), colq, rb_row(acc, r) + bias);
  }
}

// ---- vdetr_rb_transpose_f32: dst[i][k][n] = src[i][n][k] for n matrices of 256 x 256 (the forward launches' weight images) ----
__global__ __launch_bounds__(256) void rb_transpose_kernel(const float* const* __restrict__ src, float* __restrict__ dst) {
  __shared__ float tile[32][33];
  const float* S = src[blockIdx.y];
  float* D = dst + (size_t)blockIdx.y * kRbC * kRbC;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int n0 = (blockIdx.x >> 3) * 32, k0 = (blockIdx.x & 7) * 32;
#pragma unroll
  for (int i = 0; i < 4; ++i) tile[ty + 8 * i][tx] = S[(size_t)(n0 + ty + 8 * i) * kRbC + k0 + tx];
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) D[(size_t)(k0 + ty + 8 * i) * kRbC + n0 + tx] = tile[tx][ty + 8 * i];
}

// =====================================================================================================================================
// Backward of the three launches: the input-gradient chains, again one launch each.  The weight / bias gradients stay what they
// were — batched GEMMs after the backward pass (helpers.DeferredParamGrads) on the (dY, X) pairs these kernels write —, the
// LayerNorm parameter sums leave as per-workgroup partial rows in the layout of add_ln.hip (add_ln_param_reduce_batch_kernel
// sums them at the flush).  dX = dY W needs W "column-wise": with MFMA column j of tile nt = output column 64 w + 4 j + nt the
// four tiles' B operands of one contraction index are ONE float4 of a weight row.
// =====================================================================================================================================

// sum over the four row groups g of a wave (lanes c, c + 16, c + 32, c + 48), result in every lane
__device__ __forceinline__ float rb_sum_g(float v) {
  pair_u32 p = xrow16(__float_as_uint(v));
  v = __uint_as_float(p.a) + __uint_as_float(p.b);
  p = xhalf32(__float_as_uint(v));
  return __uint_as_float(p.a) + __uint_as_float(p.b);
}

// LayerNorm backward of the workgroup's 16 rows (add_ln.hip: add_ln_bwd_kernel): yv = the normalised tensor, go / go2 = the
// gradients of its one or two affine outputs, dy_in = the gradient reaching it from elsewhere.  Returns in dx the total gradient;
// adds the block's parameter sums to part [4][256] (dgamma, dbeta, dgamma2, dbeta2; rows past the end contribute nothing).
// red: [2][4 waves][16 rows].
__device__ __forceinline__ void rb_ln_bwd(const f32x4 (&yv)[4], const f32x4 (&go)[4], const f32x4 (&go2)[4], bool two, const f32x4 (&dy_in)[4],
                                          const float (&mean)[4], const float (&rstd)[4], const bool (&live)[4], const float* gamma,
                                          const float* gamma2, int colq, float* red, float* part, int w, int lane, f32x4 (&dx)[4]) {
  const int g = lane >> 4, c = lane & 15;
  const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + colq);
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  const f32x4 ga2 = two ? *reinterpret_cast<const f32x4*>(gamma2 + colq) : zero;
  f32x4 xh[4], t[4];
  f32x4 dga = zero, dbe = zero, dga2 = zero, dbe2 = zero;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float h = (yv[r][e] - mean[r]) * rstd[r];
      const float o1 = live[r] ? go[r][e] : 0.f, o2 = (live[r] && two) ? go2[r][e] : 0.f;
      dga[e] += o1 * h; dbe[e] += o1; dga2[e] += o2 * h; dbe2[e] += o2;
      const float tt = o1 * ga[e] + o2 * ga2[e];
      xh[r][e] = h; t[r][e] = tt;
      s1 += tt; s2 += tt * h;
    }
    s1 = row_allsum_f32(s1); s2 = row_allsum_f32(s2);
    if (c == 0) { red[w * 16 + 4 * g + r] = s1; red[64 + w * 16 + 4 * g + r] = s2; }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = 4 * g + r;
    const float m1 = ((red[i] + red[16 + i]) + (red[32 + i] + red[48 + i])) * (1.f / kRbC);
    const float m2 = ((red[64 + i] + red[80 + i]) + (red[96 + i] + red[112 + i])) * (1.f / kRbC);
#pragma unroll
    for (int e = 0; e < 4; ++e) dx[r][e] = dy_in[r][e] + rstd[r] * (t[r][e] - m1 - xh[r][e] * m2);
  }
  __syncthreads();  // red may be reused
  // parameter sums of this block: over r in the lane (done), over g across the wave; lanes g == 0 write their four columns
#pragma unroll
  for (int e = 0; e < 4; ++e) { dga[e] = rb_sum_g(dga[e]); dbe[e] = rb_sum_g(dbe[e]); }
  if (two) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { dga2[e] = rb_sum_g(dga2[e]); dbe2[e] = rb_sum_g(dbe2[e]); }
  }
  if (g == 0) {
    *reinterpret_cast<f32x4*>(part + colq) = dga;
    *reinterpret_cast<f32x4*>(part + kRbC + colq) = dbe;
    if (two) {
      *reinterpret_cast<f32x4*>(part + 2 * kRbC + colq) = dga2;
      *reinterpret_cast<f32x4*>(part + 3 * kRbC + colq) = dbe2;
    }
  }
}


// ---- rb_ffn_bwd_kernel ---------------------------------------------------------------------------------------------------------------
// EMIT: the packed form of d a for the key-side pass of the attention that produced a (kv_pack.h)
// FFN0: the backward of rb_ffn_kernel<-2> (A.y = the layer's INPUT rows with mean_y / rstd_y their statistics; d_tgt = its gradient)
template <bool EMIT, bool FFN0 = false>
__global__ __launch_bounds__(kRbThreads) void rb_ffn_bwd_kernel(vdetr_rb_ffn_desc A, vdetr_rb_ffn_grads G, KvEmit E) {
  __shared__ __attribute__((aligned(16))) float xs[kRbRows * kRbStride];
  __shared__ float red[128];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, c = lane & 15;
  const int row0 = blockIdx.x * kRbRows;
  const int col0 = 64 * w, colq = col0 + 4 * c;
  const RbDrop d2 = rb_drop(A.drop2.p, A.drop2.seed, 0, A.rng_state), da = rb_drop(A.drop_act.p, A.drop_act.seed, 0, A.rng_state),
               d3 = rb_drop(A.drop3.p, A.drop3.seed, 0, A.rng_state);
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  const bool two = G.d_o2 != nullptr;
  int rowc[4];
  bool live[4];
  float mean[4], rstd[4], mean2[4], rstd2[4];
  f32x4 yv[4], go[4], go2[4], din[4], dy[4], hv[4], yv2[4];
  RbRing R;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = row0 + 4 * g + r;
    live[r] = row < A.rows;
    rowc[r] = min(row, A.rows - 1);
    mean[r] = A.mean_z[rowc[r]]; rstd[r] = A.rstd_z[rowc[r]];
    yv[r] = rb_ld4(A.z, rowc[r], colq);
    go[r] = G.d_o1 ? rb_ld4(G.d_o1, rowc[r], colq) : zero;
    go2[r] = two ? rb_ld4(G.d_o2, rowc[r], colq) : zero;
    din[r] = G.d_z ? rb_ld4(G.d_z, rowc[r], colq) : zero;
  }
  rb_w_begin(A.lin2.w, col0, lane, R);
#pragma unroll
  for (int r = 0; r < 4; ++r) {  // what the later blocks read: requested now
    hv[r] = rb_ld4(A.h, rowc[r], colq);
    yv2[r] = rb_ld4(A.y, rowc[r], colq);
    mean2[r] = A.mean_y[rowc[r]]; rstd2[r] = A.rstd_y[rowc[r]];
  }
  __builtin_amdgcn_sched_barrier(0);
  // block 3 backward: z = y + drop3(lin2 h); o1 = post1(z), o2 = post2(z)
  rb_ln_bwd(yv, go, go2, two, din, mean, rstd, live, A.post1.gamma, A.post2.gamma, colq, red, G.part_post + (size_t)blockIdx.x * 4 * kRbC, w, lane, dy);
#pragma unroll
  for (int r = 0; r < 4; ++r) {  // d (lin2 output) = dy through the mask of dropout3
    bool keep[4];
    rb_keep4_ln(d3, rowc[r], colq >> 2, keep);
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = keep[e] ? dy[r][e] * d3.scale : 0.f;
    if (live[r]) rb_st4(G.d_lin2, rowc[r], colq, v);
    *reinterpret_cast<f32x4*>(xs + (4 * g + r) * kRbStride + colq) = v;
  }
  __syncthreads();
  float a[64];
  f32x4 acc[4];
  rb_load_a(xs, lane, a);
  rb_zero(acc);
  rb_w_run(a, A.lin2.w, col0, lane, R, acc);  // d h
  rb_w_begin(A.lin1.w, col0, lane, R);
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r) {  // through relu + dropout: h > 0 <=> passed both
    f32x4 v = rb_row(acc, r);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = hv[r][e] > 0.f ? v[e] * da.scale : 0.f;
    if (live[r]) rb_st4(G.d_lin1, rowc[r], colq, v);
    *reinterpret_cast<f32x4*>(xs + (4 * g + r) * kRbStride + colq) = v;
  }
  __syncthreads();
  rb_load_a(xs, lane, a);
  rb_zero(acc);
  rb_w_run(a, A.lin1.w, col0, lane, R, acc);  // d t2
  if constexpr (!FFN0) rb_w_begin(A.proj.w, col0, lane, R);
  // block 2 backward: y = tgt + drop2(proj a); t2 = norm3(y)   (FFN0: t2 = norm3(x) feeds lin1 AND the residual: d t2 = both)
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    mean[r] = mean2[r]; rstd[r] = rstd2[r];
    yv[r] = yv2[r];
    go[r] = FFN0 ? rb_row(acc, r) + dy[r] : rb_row(acc, r);
    go2[r] = zero;
    din[r] = FFN0 ? zero : dy[r];
  }
  rb_ln_bwd(yv, go, go2, false, din, mean, rstd, live, A.norm3.gamma, nullptr, colq, red, G.part_n3 + (size_t)blockIdx.x * 4 * kRbC, w, lane, dy);
  // (rb_ln_bwd's barriers: every wave is past its reads of xs)
  if constexpr (FFN0) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (live[r]) rb_st4(G.d_tgt, rowc[r], colq, dy[r]);
    return;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if (live[r]) rb_st4(G.d_tgt, rowc[r], colq, dy[r]);
    bool keep[4];
    rb_keep4_ln(d2, rowc[r], colq >> 2, keep);
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = keep[e] ? dy[r][e] * d2.scale : 0.f;
    if (live[r]) rb_st4(G.d_proj, rowc[r], colq, v);
    *reinterpret_cast<f32x4*>(xs + (4 * g + r) * kRbStride + colq) = v;
  }
  __syncthreads();
  rb_load_a(xs, lane, a);
  rb_zero(acc);
  rb_w_run(a, A.proj.w, col0, lane, R, acc);  // d a
  if (G.d_a) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (live[r]) rb_st4(G.d_a, rb_bmajor(rowc[r], A.B, A.rows / A.B), colq, rb_row(acc, r));
  }
  if constexpr (EMIT) {
    __syncthreads();  // every wave has read its operand out of the tile
#pragma unroll
    for (int r = 0; r < 4; ++r) *reinterpret_cast<f32x4*>(xs + (4 * g + r) * kRbStride + colq) = rb_row(acc, r);
    __syncthreads();
    kv_emit_rows16(E, xs, kRbStride, row0, tid, red);
  }
}

// ---- rb_proj_q_bwd_kernel ------------------------------------------------------------------------------------------------------------
template <bool EMIT>
__global__ __launch_bounds__(kRbThreads) void rb_proj_q_bwd_kernel(vdetr_rb_projq_desc A, vdetr_rb_projq_grads G, KvEmit E) {
  __shared__ __attribute__((aligned(16))) float xs[kRbRows * kRbStride];
  __shared__ float red[128];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, c = lane & 15;
  const int row0 = blockIdx.x * kRbRows;
  const int col0 = 64 * w, colq = col0 + 4 * c;
  const RbDrop d1 = rb_drop(A.drop1.p, A.drop1.seed, 0, A.rng_state);
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  int rowc[4];
  bool live[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) { live[r] = row0 + 4 * g + r < A.rows; rowc[r] = min(row0 + 4 * g + r, A.rows - 1); }
  float a[64];
  f32x4 acc[4];
  rb_zero(acc);
  RbRing R;
  float mean[4], rstd[4];
  f32x4 yv[4], go[4], go2[4], din[4], dy[4];
  rb_w_begin(G.d_qout ? A.q.w : A.proj.w, col0, lane, R);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    mean[r] = A.mean_y[rowc[r]]; rstd[r] = A.rstd_y[rowc[r]];
    yv[r] = rb_ld4(A.y, rowc[r], colq);
    din[r] = G.d_y ? rb_ld4(G.d_y, rowc[r], colq) : zero;
  }
  __builtin_amdgcn_sched_barrier(0);
  if (G.d_qout) {  // d (t2 + pos) = d q Wq
    rb_stage_rows(G.d_qout, row0, A.rows, A.B, true, xs, tid, nullptr, nullptr, G.dq_rows);
    __syncthreads();
    rb_load_a(xs, lane, a);
    rb_w_run(a, A.q.w, col0, lane, R, acc);
    rb_w_begin(A.proj.w, col0, lane, R);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    go[r] = rb_row(acc, r);
    go2[r] = zero;
    if (live[r] && G.d_t2) rb_st4(G.d_t2, rowc[r], colq, go[r]);
  }
  rb_ln_bwd(yv, go, go2, false, din, mean, rstd, live, A.norm2.gamma, nullptr, colq, red, G.part_n2 + (size_t)blockIdx.x * 4 * kRbC, w, lane, dy);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if (live[r]) rb_st4(G.d_tgt, rowc[r], colq, dy[r]);
    bool keep[4];
    rb_keep4_ln(d1, rowc[r], colq >> 2, keep);
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = keep[e] ? dy[r][e] * d1.scale : 0.f;
    if (live[r]) rb_st4(G.d_proj, rowc[r], colq, v);
    *reinterpret_cast<f32x4*>(xs + (4 * g + r) * kRbStride + colq) = v;
  }
  __syncthreads();
  rb_load_a(xs, lane, a);
  rb_zero(acc);
  rb_w_run(a, A.proj.w, col0, lane, R, acc);
  if (G.d_a) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (live[r]) rb_st4(G.d_a, rb_bmajor(rowc[r], A.B, A.rows / A.B), colq, rb_row(acc, r));
  }
  if constexpr (EMIT) {
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) *reinterpret_cast<f32x4*>(xs + (4 * g + r) * kRbStride + colq) = rb_row(acc, r);
    __syncthreads();
    kv_emit_rows16(E, xs, kRbStride, row0, tid, red);
  }
}

// ---- rb_qkv_bwd_kernel: d (t + pos) = dq Wq + dk Wk;  d t = d (t + pos) + dv Wv ------------------------------------------------------
__global__ __launch_bounds__(kRbThreads) void rb_qkv_bwd_kernel(vdetr_rb_qkv_desc A, vdetr_rb_qkv_grads G) {
  __shared__ __attribute__((aligned(16))) float xs[kRbRows * kRbStride];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, c = lane & 15;
  const int row0 = blockIdx.x * kRbRows;
  const int col0 = 64 * w, colq = col0 + 4 * c;
  float a[64];
  f32x4 acc[4];
  rb_zero(acc);
  RbRing R;
  rb_w_begin(A.w, col0, lane, R);
  const float* dsrc[3] = {G.dq, G.dk, G.dv};
  float* drows[3] = {G.dq_rows, G.dk_rows, G.dv_rows};
#pragma unroll
  for (int which = 0; which < 3; ++which) {
    if (which) __syncthreads();  // the previous operand has been read out of the tile
    rb_stage_rows(dsrc[which], row0, A.rows, A.B, true, xs, tid, nullptr, nullptr, drows[which]);
    __syncthreads();
    rb_load_a(xs, lane, a);
    rb_w_run(a, A.w + (size_t)which * kRbC * kRbC, col0, lane, R, acc);
    if (which < 2) rb_w_begin(A.w + (size_t)(which + 1) * kRbC * kRbC, col0, lane, R);
    if (which == 1 && G.d_x) {  // the gradient of pos: + what reached pos through its other consumer (d_x_add; not part of d t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = row0 + 4 * g + r;
        if (row < A.rows) rb_st4(G.d_x, row, colq, G.d_x_add ? rb_row(acc, r) + rb_ld4(G.d_x_add, row, colq) : rb_row(acc, r));
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r)
    if (row0 + 4 * g + r < A.rows) rb_st4(G.d_t, row0 + 4 * g + r, colq, rb_row(acc, r));
}

}  // namespace vdetr

using namespace vdetr;

static int rb_common(int rows, int B, const char* op) {
  VDETR_REQUIRE(rows > 0 && B > 0 && rows % B == 0, "%s: rows=%d must be a positive multiple of B=%d", op, rows, B);
  return VDETR_OK;
}

extern "C" int vdetr_rb_transpose_f32(const float* const* src, float* dst, int n, vdetr_stream_t stream) {
  VDETR_REQUIRE(src && dst && n > 0 && n <= 65535, "rb_transpose: null pointer or n=%d outside [1, 65535]", n);
  VDETR_REQUIRE(RB_ALIGNED(dst), "rb_transpose: dst must be 16-B aligned");
  hipLaunchKernelGGL(rb_transpose_kernel, dim3(64, n), dim3(256), 0, (hipStream_t)stream, src, dst);
  return check_launch("rb_transpose");
}

// e != nullptr: the emitting form (vdetr_rb_ffn_bwd_emit_f32 / vdetr_rb_proj_q_bwd_emit_f32)
static int rb_emit_args(const vdetr_rb_attn_emit* e, int rows, int B, const float* d_a, KvEmit* E, const char* op) {
  VDETR_REQUIRE(e->workspace && e->delta && e->out && d_a, "%s: null pointer (workspace, delta, out, d_a)", op);
  VDETR_REQUIRE(B == 1 && e->nQ == rows && rows % 32 == 0, "%s: one scene, nQ = rows = %d a multiple of 32 (B %d, nQ %d)", op, rows, B, e->nQ);
  VDETR_REQUIRE((((uintptr_t)e->workspace) & 255) == 0 && RB_ALIGNED(e->out), "%s: workspace 256-B, out 16-B aligned", op);
  E->pack = reinterpret_cast<uint4*>(e->workspace);
  E->delta = e->delta; E->out = e->out; E->aux = e->bwd_aux;
  E->per_head = e->per_head ? 1 : 0; E->nQ = e->nQ;
  return VDETR_OK;
}

extern "C" int vdetr_rb_qkv_f32(const vdetr_rb_qkv_desc* d, vdetr_stream_t stream) {
  VDETR_REQUIRE(d != nullptr, "rb_qkv: null descriptor");
  if (int e = rb_common(d->rows, d->B, "rb_qkv")) return e;
  VDETR_REQUIRE(d->t && d->wt && d->out && (!d->pos || d->x), "rb_qkv: null pointer (wt: the W^T images, vdetr_rb_transpose_f32)");
  VDETR_REQUIRE(RB_ALIGNED(d->t) && RB_ALIGNED(d->pos) && RB_ALIGNED(d->wt) && RB_ALIGNED(d->b) && RB_ALIGNED(d->x) && RB_ALIGNED(d->out),
                "rb_qkv: operands must be 16-B aligned");
  hipLaunchKernelGGL(rb_qkv_kernel, dim3(ceil_div(d->rows, kRbRows), 3), dim3(kRbThreads), 0, (hipStream_t)stream, *d);
  return check_launch("rb_qkv");
}

extern "C" int vdetr_rb_proj_q_f32(const vdetr_rb_projq_desc* d, vdetr_stream_t stream) {
  VDETR_REQUIRE(d != nullptr, "rb_proj_q: null descriptor");
  if (int e = rb_common(d->rows, d->B, "rb_proj_q")) return e;
  VDETR_REQUIRE(d->a && d->tgt && d->proj.wt && d->q.wt && d->norm2.gamma && d->norm2.beta && d->y && d->mean_y && d->rstd_y && d->t2 &&
                d->qout && (!d->pos || d->xq), "rb_proj_q: null pointer (proj.wt / q.wt: the W^T images, vdetr_rb_transpose_f32)");
  VDETR_REQUIRE(d->drop1.p >= 0.f && d->drop1.p < 1.f, "rb_proj_q: dropout_p %f outside [0,1)", d->drop1.p);
  VDETR_REQUIRE(RB_ALIGNED(d->a) && RB_ALIGNED(d->tgt) && RB_ALIGNED(d->pos) && RB_ALIGNED(d->proj.wt) && RB_ALIGNED(d->proj.b) &&
                RB_ALIGNED(d->q.wt) && RB_ALIGNED(d->q.b) && RB_ALIGNED(d->norm2.gamma) && RB_ALIGNED(d->norm2.beta) && RB_ALIGNED(d->y) &&
                RB_ALIGNED(d->t2) && RB_ALIGNED(d->xq) && RB_ALIGNED(d->qout), "rb_proj_q: operands must be 16-B aligned");
  hipLaunchKernelGGL(rb_proj_q_kernel, dim3(ceil_div(d->rows, kRbRows)), dim3(kRbThreads), 0, (hipStream_t)stream, *d);
  return check_launch("rb_proj_q");
}

extern "C" int vdetr_rb_ffn_f32(const vdetr_rb_ffn_desc* d, vdetr_stream_t stream) {
  VDETR_REQUIRE(d != nullptr, "rb_ffn: null descriptor");
  if (int e = rb_common(d->rows, d->B, "rb_ffn")) return e;
  VDETR_REQUIRE(d->a && d->tgt && d->proj.wt && d->lin1.wt && d->lin2.wt && d->norm3.gamma && d->norm3.beta && d->post1.gamma && d->post1.beta &&
                d->y && d->mean_y && d->rstd_y && d->t2 && d->h && d->z && d->mean_z && d->rstd_z && d->o1,
                "rb_ffn: null pointer (proj.wt / lin1.wt / lin2.wt: the W^T images, vdetr_rb_transpose_f32)");
  VDETR_REQUIRE((d->post2.gamma == nullptr) == (d->post2.beta == nullptr) && (d->post2.gamma == nullptr) == (d->o2 == nullptr),
                "rb_ffn: post2.gamma, post2.beta and o2 go together");
  for (const vdetr_rb_drop* dr : {&d->drop2, &d->drop_act, &d->drop3})
    VDETR_REQUIRE(dr->p >= 0.f && dr->p < 1.f, "rb_ffn: dropout_p %f outside [0,1)", dr->p);
  VDETR_REQUIRE(RB_ALIGNED(d->a) && RB_ALIGNED(d->tgt) && RB_ALIGNED(d->proj.wt) && RB_ALIGNED(d->proj.b) && RB_ALIGNED(d->lin1.wt) &&
                RB_ALIGNED(d->lin1.b) && RB_ALIGNED(d->lin2.wt) && RB_ALIGNED(d->lin2.b) && RB_ALIGNED(d->norm3.gamma) && RB_ALIGNED(d->norm3.beta) &&
                RB_ALIGNED(d->post1.gamma) && RB_ALIGNED(d->post1.beta) && RB_ALIGNED(d->post2.gamma) && RB_ALIGNED(d->post2.beta) &&
                RB_ALIGNED(d->y) && RB_ALIGNED(d->t2) && RB_ALIGNED(d->h) && RB_ALIGNED(d->z) && RB_ALIGNED(d->o1) && RB_ALIGNED(d->o2),
                "rb_ffn: operands must be 16-B aligned");
  hipLaunchKernelGGL(rb_ffn_kernel<-1>, dim3(ceil_div(d->rows, kRbRows)), dim3(kRbThreads), 0, (hipStream_t)stream, *d, RbParts{});
  return check_launch("rb_ffn");
}

// the FFN layer in front of the decoder (rb_ffn_kernel<-2>): d->tgt = its input rows, norm3 = its norm, post1 (post2) = the norms applied
// to its output; d->a, d->proj, d->y, d->drop2 are not used
static int rb_ffn0_check(const vdetr_rb_ffn_desc* d, const char* op) {
  VDETR_REQUIRE(d != nullptr, "%s: null descriptor", op);
  if (int e = rb_common(d->rows, d->B, op)) return e;
  VDETR_REQUIRE(d->norm3.gamma && d->post1.gamma && d->mean_y && d->rstd_y && d->h && d->z && d->mean_z && d->rstd_z, "%s: null pointer", op);
  for (const vdetr_rb_drop* dr : {&d->drop_act, &d->drop3})
    VDETR_REQUIRE(dr->p >= 0.f && dr->p < 1.f, "%s: dropout_p %f outside [0,1)", op, dr->p);
  return VDETR_OK;
}

extern "C" int vdetr_rb_ffn0_f32(const vdetr_rb_ffn_desc* d, vdetr_stream_t stream) {
  if (int e = rb_ffn0_check(d, "rb_ffn0")) return e;
  VDETR_REQUIRE(d->tgt && d->lin1.wt && d->lin2.wt && d->norm3.beta && d->post1.beta && d->t2 && d->o1,
                "rb_ffn0: null pointer (lin1.wt / lin2.wt: the W^T images, vdetr_rb_transpose_f32)");
  VDETR_REQUIRE((d->post2.gamma == nullptr) == (d->post2.beta == nullptr) && (d->post2.gamma == nullptr) == (d->o2 == nullptr),
                "rb_ffn0: post2.gamma, post2.beta and o2 go together");
  VDETR_REQUIRE(RB_ALIGNED(d->tgt) && RB_ALIGNED(d->lin1.wt) && RB_ALIGNED(d->lin1.b) && RB_ALIGNED(d->lin2.wt) && RB_ALIGNED(d->lin2.b) &&
                RB_ALIGNED(d->norm3.gamma) && RB_ALIGNED(d->norm3.beta) && RB_ALIGNED(d->post1.gamma) && RB_ALIGNED(d->post1.beta) &&
                RB_ALIGNED(d->post2.gamma) && RB_ALIGNED(d->post2.beta) && RB_ALIGNED(d->t2) && RB_ALIGNED(d->h) && RB_ALIGNED(d->z) &&
                RB_ALIGNED(d->o1) && RB_ALIGNED(d->o2), "rb_ffn0: operands must be 16-B aligned");
  hipLaunchKernelGGL(rb_ffn_kernel<-2>, dim3(ceil_div(d->rows, kRbRows)), dim3(kRbThreads), 0, (hipStream_t)stream, *d, RbParts{});
  return check_launch("rb_ffn0");
}

extern "C" int vdetr_rb_ffn0_bwd_f32(const vdetr_rb_ffn_desc* d, const vdetr_rb_ffn_grads* g, vdetr_stream_t stream) {
  if (int e = rb_ffn0_check(d, "rb_ffn0_bwd")) return e;
  VDETR_REQUIRE(g != nullptr && d->y && d->lin1.w && d->lin2.w && g->d_tgt && g->d_lin2 && g->d_lin1 && g->part_post && g->part_n3 &&
                (g->d_z || g->d_o1 || g->d_o2), "rb_ffn0_bwd: null pointer (d->y = the layer's input rows)");
  VDETR_REQUIRE(!g->d_o2 || d->post2.gamma, "rb_ffn0_bwd: d_o2 without a second output norm");
  VDETR_REQUIRE(RB_ALIGNED(d->y) && RB_ALIGNED(d->lin1.w) && RB_ALIGNED(d->lin2.w) && RB_ALIGNED(d->norm3.gamma) && RB_ALIGNED(d->post1.gamma) &&
                RB_ALIGNED(d->post2.gamma) && RB_ALIGNED(d->h) && RB_ALIGNED(d->z) && RB_ALIGNED(g->d_z) && RB_ALIGNED(g->d_o1) && RB_ALIGNED(g->d_o2) &&
                RB_ALIGNED(g->d_tgt) && RB_ALIGNED(g->d_lin2) && RB_ALIGNED(g->d_lin1) && RB_ALIGNED(g->part_post) && RB_ALIGNED(g->part_n3),
                "rb_ffn0_bwd: operands must be 16-B aligned");
  hipLaunchKernelGGL((rb_ffn_bwd_kernel<false, true>), dim3(ceil_div(d->rows, kRbRows)), dim3(kRbThreads), 0, (hipStream_t)stream, *d, *g, KvEmit{});
  return check_launch("rb_ffn0_bwd");
}

extern "C" int vdetr_rb_ffn_parts_f32(const vdetr_rb_ffn_desc* d, const vdetr_attn_parts* parts, float* attn_out, float* attn_lse,
                                      vdetr_stream_t stream) {
  VDETR_REQUIRE(d != nullptr && parts != nullptr, "rb_ffn_parts: null descriptor");
  if (int e = rb_common(d->rows, d->B, "rb_ffn_parts")) return e;
  VDETR_REQUIRE(attn_out && attn_lse && d->tgt && d->proj.wt && d->lin1.wt && d->lin2.wt && d->norm3.gamma && d->norm3.beta && d->post1.gamma &&
                d->post1.beta && d->y && d->mean_y && d->rstd_y && d->t2 && d->h && d->z && d->mean_z && d->rstd_z && d->o1,
                "rb_ffn_parts: null pointer (proj.wt / lin1.wt / lin2.wt: the W^T images, vdetr_rb_transpose_f32)");
  VDETR_REQUIRE(parts->ksplit >= 2 && parts->ksplit <= 16 && parts->part_o && parts->part_lse,
                "rb_ffn_parts: ksplit %d outside 2..16 or null partials (ksplit 1: the forward's out / lse are final, take vdetr_rb_ffn_f32)", parts->ksplit);
  VDETR_REQUIRE(parts->rows == (int64_t)d->rows * 4, "rb_ffn_parts: %lld partial rows for %d rows of 4 heads", (long long)parts->rows, d->rows);
  VDETR_REQUIRE((d->post2.gamma == nullptr) == (d->post2.beta == nullptr) && (d->post2.gamma == nullptr) == (d->o2 == nullptr),
                "rb_ffn_parts: post2.gamma, post2.beta and o2 go together");
  for (const vdetr_rb_drop* dr : {&d->drop2, &d->drop_act, &d->drop3})
    VDETR_REQUIRE(dr->p >= 0.f && dr->p < 1.f, "rb_ffn_parts: dropout_p %f outside [0,1)", dr->p);
  VDETR_REQUIRE(RB_ALIGNED(attn_out) && RB_ALIGNED(parts->part_o) && RB_ALIGNED(d->tgt) && RB_ALIGNED(d->proj.wt) && RB_ALIGNED(d->proj.b) &&
                RB_ALIGNED(d->lin1.wt) && RB_ALIGNED(d->lin1.b) && RB_ALIGNED(d->lin2.wt) && RB_ALIGNED(d->lin2.b) && RB_ALIGNED(d->norm3.gamma) &&
                RB_ALIGNED(d->norm3.beta) && RB_ALIGNED(d->post1.gamma) && RB_ALIGNED(d->post1.beta) && RB_ALIGNED(d->post2.gamma) &&
                RB_ALIGNED(d->post2.beta) && RB_ALIGNED(d->y) && RB_ALIGNED(d->t2) && RB_ALIGNED(d->h) && RB_ALIGNED(d->z) && RB_ALIGNED(d->o1) &&
                RB_ALIGNED(d->o2), "rb_ffn_parts: operands must be 16-B aligned");
  RbParts P{parts->part_o, parts->part_lse, attn_out, attn_lse, (long)parts->rows, parts->ksplit};
  const dim3 grid(ceil_div(d->rows, kRbRows));
  if (parts->ksplit == 4) hipLaunchKernelGGL(rb_ffn_kernel<4>, grid, dim3(kRbThreads), 0, (hipStream_t)stream, *d, P);
  else if (parts->ksplit == 2) hipLaunchKernelGGL(rb_ffn_kernel<2>, grid, dim3(kRbThreads), 0, (hipStream_t)stream, *d, P);
  else hipLaunchKernelGGL(rb_ffn_kernel<0>, grid, dim3(kRbThreads), 0, (hipStream_t)stream, *d, P);
  return check_launch("rb_ffn_parts");
}

extern "C" int vdetr_rb_qkv_bwd_f32(const vdetr_rb_qkv_desc* d, const vdetr_rb_qkv_grads* g, vdetr_stream_t stream) {
  VDETR_REQUIRE(d != nullptr && g != nullptr, "rb_qkv_bwd: null descriptor");
  if (int e = rb_common(d->rows, d->B, "rb_qkv_bwd")) return e;
  VDETR_REQUIRE(d->w && g->dq && g->dk && g->dv && g->d_t, "rb_qkv_bwd: null pointer");
  VDETR_REQUIRE(RB_ALIGNED(d->w) && RB_ALIGNED(g->dq) && RB_ALIGNED(g->dk) && RB_ALIGNED(g->dv) && RB_ALIGNED(g->dq_rows) && RB_ALIGNED(g->dk_rows) &&
                RB_ALIGNED(g->dv_rows) && RB_ALIGNED(g->d_x) && RB_ALIGNED(g->d_x_add) && RB_ALIGNED(g->d_t), "rb_qkv_bwd: operands must be 16-B aligned");
  VDETR_REQUIRE(!g->d_x_add || g->d_x, "rb_qkv_bwd: d_x_add without d_x");
  hipLaunchKernelGGL(rb_qkv_bwd_kernel, dim3(ceil_div(d->rows, kRbRows)), dim3(kRbThreads), 0, (hipStream_t)stream, *d, *g);
  return check_launch("rb_qkv_bwd");
}

static int rb_proj_q_bwd_run(const vdetr_rb_projq_desc* d, const vdetr_rb_projq_grads* g, const vdetr_rb_attn_emit* e, vdetr_stream_t stream) {
  VDETR_REQUIRE(d != nullptr && g != nullptr, "rb_proj_q_bwd: null descriptor");
  if (int e = rb_common(d->rows, d->B, "rb_proj_q_bwd")) return e;
  VDETR_REQUIRE(d->proj.w && d->q.w && d->norm2.gamma && d->y && d->mean_y && d->rstd_y && g->d_tgt && g->d_proj && g->part_n2 &&
                (g->d_y || g->d_qout), "rb_proj_q_bwd: null pointer");
  VDETR_REQUIRE(RB_ALIGNED(d->proj.w) && RB_ALIGNED(d->q.w) && RB_ALIGNED(d->norm2.gamma) && RB_ALIGNED(d->y) && RB_ALIGNED(g->d_y) &&
                RB_ALIGNED(g->d_qout) && RB_ALIGNED(g->d_tgt) && RB_ALIGNED(g->d_a) && RB_ALIGNED(g->d_t2) && RB_ALIGNED(g->dq_rows) &&
                RB_ALIGNED(g->d_proj) && RB_ALIGNED(g->part_n2), "rb_proj_q_bwd: operands must be 16-B aligned");
  if (e) {
    KvEmit E;
    if (int err = rb_emit_args(e, d->rows, d->B, g->d_a, &E, "rb_proj_q_bwd_emit")) return err;
    hipLaunchKernelGGL(rb_proj_q_bwd_kernel<true>, dim3(ceil_div(d->rows, kRbRows)), dim3(kRbThreads), 0, (hipStream_t)stream, *d, *g, E);
  } else {
    hipLaunchKernelGGL(rb_proj_q_bwd_kernel<false>, dim3(ceil_div(d->rows, kRbRows)), dim3(kRbThreads), 0, (hipStream_t)stream, *d, *g, KvEmit{});
  }
  return check_launch("rb_proj_q_bwd");
}

extern "C" int vdetr_rb_proj_q_bwd_f32(const vdetr_rb_projq_desc* d, const vdetr_rb_projq_grads* g, vdetr_stream_t stream) {
  return rb_proj_q_bwd_run(d, g, nullptr, stream);
}
extern "C" int vdetr_rb_proj_q_bwd_emit_f32(const vdetr_rb_projq_desc* d, const vdetr_rb_projq_grads* g, const vdetr_rb_attn_emit* e,
                                            vdetr_stream_t stream) {
  VDETR_REQUIRE(e != nullptr, "rb_proj_q_bwd_emit: null descriptor");
  return rb_proj_q_bwd_run(d, g, e, stream);
}

static int rb_ffn_bwd_run(const vdetr_rb_ffn_desc* d, const vdetr_rb_ffn_grads* g, const vdetr_rb_attn_emit* e, vdetr_stream_t stream) {
  VDETR_REQUIRE(d != nullptr && g != nullptr, "rb_ffn_bwd: null descriptor");
  if (int e = rb_common(d->rows, d->B, "rb_ffn_bwd")) return e;
  VDETR_REQUIRE(d->proj.w && d->lin1.w && d->lin2.w && d->norm3.gamma && d->post1.gamma && d->y && d->mean_y && d->rstd_y && d->h && d->z &&
                d->mean_z && d->rstd_z && g->d_tgt && g->d_lin2 && g->d_lin1 && g->d_proj && g->part_post && g->part_n3 &&
                (g->d_z || g->d_o1 || g->d_o2), "rb_ffn_bwd: null pointer");
  VDETR_REQUIRE(!g->d_o2 || d->post2.gamma, "rb_ffn_bwd: d_o2 without a second output norm");
  VDETR_REQUIRE(RB_ALIGNED(d->proj.w) && RB_ALIGNED(d->lin1.w) && RB_ALIGNED(d->lin2.w) && RB_ALIGNED(d->norm3.gamma) && RB_ALIGNED(d->post1.gamma) &&
                RB_ALIGNED(d->post2.gamma) && RB_ALIGNED(d->y) && RB_ALIGNED(d->h) && RB_ALIGNED(d->z) && RB_ALIGNED(g->d_z) && RB_ALIGNED(g->d_o1) &&
                RB_ALIGNED(g->d_o2) && RB_ALIGNED(g->d_tgt) && RB_ALIGNED(g->d_a) && RB_ALIGNED(g->d_lin2) && RB_ALIGNED(g->d_lin1) &&
                RB_ALIGNED(g->d_proj) && RB_ALIGNED(g->part_post) && RB_ALIGNED(g->part_n3), "rb_ffn_bwd: operands must be 16-B aligned");
  if (e) {
    KvEmit E;
    if (int err = rb_emit_args(e, d->rows, d->B, g->d_a, &E, "rb_ffn_bwd_emit")) return err;
    hipLaunchKernelGGL(rb_ffn_bwd_kernel<true>, dim3(ceil_div(d->rows, kRbRows)), dim3(kRbThreads), 0, (hipStream_t)stream, *d, *g, E);
  } else {
    hipLaunchKernelGGL(rb_ffn_bwd_kernel<false>, dim3(ceil_div(d->rows, kRbRows)), dim3(kRbThreads), 0, (hipStream_t)stream, *d, *g, KvEmit{});
  }
  return check_launch("rb_ffn_bwd");
}

extern "C" int vdetr_rb_ffn_bwd_f32(const vdetr_rb_ffn_desc* d, const vdetr_rb_ffn_grads* g, vdetr_stream_t stream) {
  return rb_ffn_bwd_run(d, g, nullptr, stream);
}
extern "C" int vdetr_rb_ffn_bwd_emit_f32(const vdetr_rb_ffn_desc* d, const vdetr_rb_ffn_grads* g, const vdetr_rb_attn_emit* e,
                                         vdetr_stream_t stream) {
  VDETR_REQUIRE(e != nullptr, "rb_ffn_bwd_emit: null descriptor");
  return rb_ffn_bwd_run(d, g, e, stream);
}
