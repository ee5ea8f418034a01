// heads.hip — the five box heads of a decoder stage as THREE launches, and the learned query-position MLP as ONE (round 6).
//
// Reference: get_proposal_box_predictions_refine (models/vdetr_transformer.py:244-285) runs five GenericMLPs
// (models/helpers.py:74-141: Conv1d -> BatchNorm1d -> ReLU -> Dropout, twice, -> Conv1d) on the same [B, 256, N] features of every
// one of the 9 stages; PositionEmbeddingLearned (helpers.py:17-33) turns the decoded box of a query into the next layer's
// query position.  Rounds 1-5 ran a stage's heads as three library GEMMs, two BatchNorm launches and a bias add (~58 us for
// 1024 tokens, the GEMMs 13 / 17 / 10 us each at 5-20 % of the chip) and the position MLP as three more (~19 us).
//
// BatchNorm's batch statistics are the only thing that is not local to a tile of tokens.  They are handled WITHOUT a grid
// barrier and without atomics: the launch that produces a layer's pre-activations also writes, per 32-token tile and channel,
// the tile's (mean, M2); the next launch's workgroups merge the tiles of the 256 channels they consume (Chan et al.; 32
// partials for 1024 tokens) while their weights are in flight.  A device-scope barrier costs 4-7 us on this chip
// (MI355X_MICROARCH.md: barrier-xcd) against ~1.5 us for a launch boundary, the partials make every run bit-identical, and the
// workgroups need not be co-resident (the next scene's sampling kernel and, in the backward, the table kernels hold CUs).
//
//   heads_l1_kernel   pre1[b][g*256 + c][n] = sum_k x[n][b][k] W1[g][c][k]                        grid (B N / 32, G)
//   heads_l2_kernel   h1 = drop(relu(bn1(pre1)));  pre2[b][g*256 + c][n] = sum_k W2[g][c][k] h1[b][g*256 + k][n]
//   heads_l3_kernel   h2 = drop(relu(bn2(pre2)));  y[b][g][r][n] = sum_k W3[g][r][k] h2[b][g*256 + k][n] + b3[g][r]   grid (B N / 16, G)
//   pos_mlp_kernel    out = W2 relu(bn(W1 x)) + b2 on [B, N, cin <= 8] coordinates                 grid (B N / 16)
//
// Matrix products: v_mfma_f32_16x16x4_f32 (exact fp32 products, the library GEMMs' numerics class), MFMA rows = tokens,
// MFMA column j of tile u = channel 64 w + 4 j + u (rowblock.hip's interleave): a lane's four accumulators of a token are four
// ADJACENT channels, the four tiles' B operands of one contraction index are one float4 of a W^T image row (quads of lanes read
// 64 contiguous bytes: the fast path of the load unit), and a lane's four accumulators of a CHANNEL are four
// consecutive tokens — one float4 of the channel-major tensors the backward reads.
#include "attn_common.h"
#include "bn_common.h"

namespace vdetr {

constexpr int kHdC = 256;          // channels of a head's hidden layers = contraction length of every product
constexpr int kHdThreads = 256;
constexpr int kHdTok = 32;         // tokens per workgroup of the hidden layers (= tokens per statistics partial)
constexpr int kHdTok3 = 16;        // tokens per workgroup of the output layer / the position MLP
constexpr int kHdStride = kHdC + 4;  // floats per LDS row: 16 rows x 1040 B land on 16 different 16-byte slots
constexpr int kHdDepth = 4;        // weight steps (of 16 contraction indices) in flight per wave
constexpr int kHdMergeAbove = 48;  // tiles above which the statistics partials are merged by a launch of their own

struct HdRing {
  f32x4 b[kHdDepth][4];
};
__device__ __forceinline__ f32x4 hd_ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void hd_st4(float* p, const f32x4& v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ const float* hd_w_ptr(const float* __restrict__ Wt, int col0, int lane) {
  return Wt + (size_t)(4 * (lane >> 4)) * kHdC + col0 + 4 * (lane & 15);  // + (16 m + e) rows
}
__device__ __forceinline__ void hd_w_begin(const float* __restrict__ Wt, int col0, int lane, HdRing& R) {
  const float* wp = hd_w_ptr(Wt, col0, lane);
#pragma unroll
  for (int d = 0; d < kHdDepth - 1; ++d)
#pragma unroll
    for (int e = 0; e < 4; ++e) R.b[d][e] = hd_ld4(wp + (size_t)(16 * d + e) * kHdC);
  __builtin_amdgcn_sched_barrier(0);
}
// acc[t][u][r] += sum_k xs[16 t + 4 kg + r][k] Wt[k][col0 + 4 c + u]      (lane = (kg = lane >> 4, c = lane & 15); xs: LDS rows of
// kHdStride floats).  The contraction index of step (m, e) in lane group kg is 16 m + 4 kg + e for both operands.
template <int NT>
__device__ __forceinline__ void hd_w_run(const float* xs, const float* __restrict__ Wt, int col0, int lane, HdRing& R, f32x4 (&acc)[NT][4]) {
  const float* wp = hd_w_ptr(Wt, col0, lane);
  const float* ap = xs + (lane & 15) * kHdStride + 4 * (lane >> 4);
  f32x4 a[2][NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) a[0][t] = hd_ld4(ap + 16 * t * kHdStride);
#pragma unroll
  for (int m = 0; m < 16; ++m) {
    if (m + kHdDepth - 1 < 16) {
#pragma unroll
      for (int e = 0; e < 4; ++e) R.b[(m + kHdDepth - 1) % kHdDepth][e] = hd_ld4(wp + (size_t)(16 * (m + kHdDepth - 1) + e) * kHdC);
    }
    if (m + 1 < 16) {
#pragma unroll
      for (int t = 0; t < NT; ++t) a[(m + 1) & 1][t] = hd_ld4(ap + 16 * t * kHdStride + 16 * (m + 1));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int u = 0; u < 4; ++u)
          acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m & 1][t][e], R.b[m % kHdDepth][e][u], acc[t][u], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}
// sum over the four lane groups kg of a wave (lanes c, c + 16, c + 32, c + 48), result in every lane
__device__ __forceinline__ float hd_xsum(float v) {
  const pair_u32 p = xrow16(__float_as_uint(v));
  v = __uint_as_float(p.a) + __uint_as_float(p.b);
  const pair_u32 q = xhalf32(__float_as_uint(v));
  return __uint_as_float(q.a) + __uint_as_float(q.b);
}

// The hidden layers' epilogue: the tile's pre-activations to the channel-major tensor (a float4 = four consecutive tokens of a
// channel) and the tile's per-channel (mean, M2) to the partial table.  out: scene b's [G*256][N]; ch: this lane's first channel.
template <int NT>
__device__ __forceinline__ void hd_store_stats(const f32x4 (&acc)[NT][4], float* __restrict__ out, float* __restrict__ part, int ch, int q0,
                                               int N, int lane) {
  const int kg = lane >> 4;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    float* row = out + (size_t)(ch + u) * N + q0 + 4 * kg;
#pragma unroll
    for (int t = 0; t < NT; ++t) hd_st4(row + 16 * t, acc[t][u]);
  }
  float mean[4], m2[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) s += (acc[t][u][0] + acc[t][u][1]) + (acc[t][u][2] + acc[t][u][3]);
    mean[u] = hd_xsum(s) * (1.f / (16 * NT));
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    float q = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) { const float d = acc[t][u][r] - mean[u]; q += d * d; }
    m2[u] = hd_xsum(q);
  }
  if (kg == 0) {  // [tile][channel][2]: this lane's four channels are 32 contiguous bytes
    hd_st4(part + 2 * ch, f32x4{mean[0], m2[0], mean[1], m2[1]});
    hd_st4(part + 2 * ch + 4, f32x4{mean[2], m2[2], mean[3], m2[3]});
  }
}

// (mean, M2) of channel `ch` over all P tiles of `cnt` elements each (Chan's pairwise update, tile after tile: the same order in
// every workgroup and every run)
__device__ __forceinline__ void hd_merge(const float* __restrict__ part, int P, int GC, int ch, float cnt, float& mean, float& m2) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  mean = 0.f; m2 = 0.f;
  float n = 0.f;
  const f32x2* p = reinterpret_cast<const f32x2*>(part) + ch;
  int i = 0;
  for (; i + 8 <= P; i += 8) {
    f32x2 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = p[(size_t)(i + j) * GC];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float nn = n + cnt, d = v[j][0] - mean;
      mean += d * (cnt / nn);
      m2 += v[j][1] + d * d * (n * cnt / nn);
      n = nn;
    }
  }
  for (; i < P; ++i) {
    const f32x2 v = p[(size_t)i * GC];
    const float nn = n + cnt, d = v[0] - mean;
    mean += d * (cnt / nn);
    m2 += v[1] + d * d * (n * cnt / nn);
    n = nn;
  }
}

// What a consumer workgroup needs of the BatchNorm in front of it: thread tid merges channel g*256 + tid, leaves the affine map
// (a, sh) and the channel's dropout key in LDS; the first token tile's workgroups also write what the backward and nn.BatchNorm1d
// expect (save_mean / save_invstd, running statistics, num_batches_tracked).
struct HdBn {
  const float* part;
  const float *gamma, *beta;
  float *running_mean, *running_var, *save_mean, *save_invstd;
  int64_t* counter;
};
__device__ __forceinline__ void hd_bn_prepare(const HdBn& S, int P, int GC, int g, int tid, float eps, float momentum, const BnRng& rg,
                                              bool owner, float* ab, unsigned* ck, int merged) {
  const int ch = g * kHdC + tid;
  float mean, m2;
  const float n = (float)P * kHdTok;
  if (merged) {  // heads_merge_kernel has folded the P tiles into entry 0 of the table
    const float* p = S.part + 2 * (size_t)ch;
    mean = p[0]; m2 = p[1];
  } else {
    hd_merge(S.part, P, GC, ch, (float)kHdTok, mean, m2);
  }
  const float var = m2 / n;  // biased: what the batch is normalised with
  const float invstd = rsqrtf(var + eps);
  const float a = S.gamma[ch] * invstd, sh = S.beta[ch] - mean * a;
  ab[2 * tid] = a;
  ab[2 * tid + 1] = sh;
  ck[tid] = bn_chankey(rg, ch);
  if (owner) {
    S.save_mean[ch] = mean;
    S.save_invstd[ch] = invstd;
    if (S.running_mean) {
      S.running_mean[ch] = (1.f - momentum) * S.running_mean[ch] + momentum * mean;
      S.running_var[ch] = (1.f - momentum) * S.running_var[ch] + momentum * var * (n / (n > 1.f ? n - 1.f : 1.f));
    }
    if (tid == 0 && S.counter) S.counter[0] += 1;
  }
}
// dropout(relu(v * a + sh)) for four consecutive elements e0 .. e0 + 3 of a channel (e0 even)
__device__ __forceinline__ f32x4 hd_act4(const f32x4& v, float a, float sh, const BnRng& rg, unsigned key, int e0) {
  f32x4 h;
#pragma unroll
  for (int e = 0; e < 4; ++e) h[e] = fmaxf(v[e] * a + sh, 0.f);
  if (rg.thresh) {
    const unsigned x0 = bn_draw2(key, e0 >> 1), x1 = bn_draw2(key, (e0 >> 1) + 1);
    h[0] = (x0 & 0xFFFFu) >= rg.thresh ? h[0] * rg.scale : 0.f;
    h[1] = (x0 >> 16) >= rg.thresh ? h[1] * rg.scale : 0.f;
    h[2] = (x1 & 0xFFFFu) >= rg.thresh ? h[2] * rg.scale : 0.f;
    h[3] = (x1 >> 16) >= rg.thresh ? h[3] * rg.scale : 0.f;
  }
  return h;
}

typedef vdetr_heads_desc HdArgs;
__device__ __forceinline__ float* hd_part(const HdArgs& A, int which) {  // the two partial tables: [B N / 32][G*256][2] floats each
  return reinterpret_cast<float*>(A.workspace) + (size_t)which * (A.B * A.N / kHdTok) * A.G * kHdC * 2;
}

// ---- launch 1 -----------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kHdThreads) void heads_l1_kernel(HdArgs A) {
  __shared__ __attribute__((aligned(16))) float xs[kHdTok * kHdStride];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tps = A.N / kHdTok, tile = blockIdx.x, b = tile / tps, q0 = (tile - b * tps) * kHdTok;
  const int g = blockIdx.y, col0 = 64 * w, GC = A.G * kHdC;
  const float* Wt = A.w1t + (size_t)g * kHdC * kHdC;
  HdRing R;
  hd_w_begin(Wt, col0, lane, R);
  {  // the tile's 32 feature rows (token q0 + r of scene b = row (q0 + r) B + b of the sequence-first tensor) -> LDS
    constexpr int kPer = kHdTok * kHdC / 4 / kHdThreads;  // 8 float4 per thread
    f32x4 v[kPer];
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
      const int e = tid + u * kHdThreads, r = e >> 6, c4 = e & 63;
      v[u] = hd_ld4(A.x + ((size_t)(q0 + r) * A.B + b) * kHdC + 4 * c4);
    }
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
      const int e = tid + u * kHdThreads, r = e >> 6, c4 = e & 63;
      hd_st4(xs + r * kHdStride + 4 * c4, v[u]);
    }
  }
  __syncthreads();
  f32x4 acc[2][4];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[t][u] = f32x4{0.f, 0.f, 0.f, 0.f};
  hd_w_run<2>(xs, Wt, col0, lane, R, acc);
  hd_store_stats<2>(acc, A.pre1 + (size_t)b * GC * A.N, hd_part(A, 0) + (size_t)tile * GC * 2, g * kHdC + col0 + 4 * (lane & 15), q0, A.N, lane);
}

// ---- launch 2 -----------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kHdThreads) void heads_l2_kernel(HdArgs A, int merged) {
  __shared__ __attribute__((aligned(16))) float xs[kHdTok * kHdStride];
  __shared__ float ab[2 * kHdC];
  __shared__ unsigned ck[kHdC];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tps = A.N / kHdTok, tile = blockIdx.x, b = tile / tps, q0 = (tile - b * tps) * kHdTok;
  const int g = blockIdx.y, col0 = 64 * w, GC = A.G * kHdC;
  const float* Wt = A.w2t + (size_t)g * kHdC * kHdC;
  HdRing R;
  hd_w_begin(Wt, col0, lane, R);
  // the tile of pre1 this workgroup turns into h1: channels g*256 + 64 w + 8 it + (lane >> 3), tokens q0 + 4 (lane & 7) ..
  const int tq = lane & 7, kk = lane >> 3;
  const size_t base = ((size_t)b * GC + g * kHdC) * A.N + q0 + 4 * tq;
  f32x4 v[8];
#pragma unroll
  for (int it = 0; it < 8; ++it) v[it] = hd_ld4(A.pre1 + base + (size_t)(64 * w + 8 * it + kk) * A.N);
  const BnRng rg = bn_rng_of(A.p1, A.salt1, 0, A.rng_state);
  const HdBn S = {hd_part(A, 0), A.gamma1, A.beta1, A.running_mean1, A.running_var1, A.save_mean1, A.save_invstd1, A.counters1[g]};
  hd_bn_prepare(S, A.B * tps, GC, g, tid, A.eps, A.momentum, rg, blockIdx.x == 0, ab, ck, merged);
  __syncthreads();
  const int e0 = b * A.N + q0 + 4 * tq;
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int k = 64 * w + 8 * it + kk;
    const f32x4 h = hd_act4(v[it], ab[2 * k], ab[2 * k + 1], rg, ck[k], e0);
    hd_st4(A.h1 + base + (size_t)k * A.N, h);
#pragma unroll
    for (int e = 0; e < 4; ++e) xs[(4 * tq + e) * kHdStride + k] = h[e];
  }
  __syncthreads();
  f32x4 acc[2][4];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[t][u] = f32x4{0.f, 0.f, 0.f, 0.f};
  hd_w_run<2>(xs, Wt, col0, lane, R, acc);
  hd_store_stats<2>(acc, A.pre2 + (size_t)b * GC * A.N, hd_part(A, 1) + (size_t)tile * GC * 2, g * kHdC + col0 + 4 * (lane & 15), q0, A.N, lane);
}

// ---- between the launches, for long sequences only: the P tile partials of every channel folded into entry 0 of the table -------
// (every consumer workgroup merging P partials itself is P^2 work: fine at 32 tiles (1024 tokens), 335 MB of L2 reads at 128)
// Eight lanes per channel: each folds an eighth of the tiles (Chan's update, tile after tile), the eight partial (n, mean, M2) are
// combined pairwise by three lane exchanges, lower lane = left operand: a fixed tree, the same bits in every run.  (One thread per
// channel walking all 128 tiles was a 14 us chain of dependent divisions on 5 workgroups, twice in front of the first stage.)
constexpr int kHdMergeLanes = 8;
__global__ __launch_bounds__(kHdThreads) void heads_merge_kernel(float* part, int P, int GC) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const int t = blockIdx.x * kHdThreads + threadIdx.x;
  const int ch = min(t / kHdMergeLanes, GC - 1), sub = t % kHdMergeLanes;  // (lanes past the end fold channel GC - 1 again, and do not store)
  const int per = (P + kHdMergeLanes - 1) / kHdMergeLanes, i0 = sub * per, i1 = min(P, i0 + per);
  const float cnt = (float)kHdTok;
  float n = 0.f, mean = 0.f, m2 = 0.f;
  const f32x2* p = reinterpret_cast<const f32x2*>(part) + ch;
  for (int i = i0; i < i1; i += 4) {
    f32x2 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = i + j < i1 ? p[(size_t)(i + j) * GC] : f32x2{0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (i + j < i1) {
        const float nn = n + cnt, d = v[j][0] - mean;
        mean += d * (cnt / nn);
        m2 += v[j][1] + d * d * (n * cnt / nn);
        n = nn;
      }
    }
  }
#pragma unroll
  for (int m = 1; m < kHdMergeLanes; m <<= 1) {
    const float no = __shfl_xor(n, m, 64), mo = __shfl_xor(mean, m, 64), qo = __shfl_xor(m2, m, 64);
    const bool left = (sub & m) == 0;  // this lane's state is the left operand of the pair
    const float na = left ? n : no, ma = left ? mean : mo, qa = left ? m2 : qo;
    const float nb = left ? no : n, mb = left ? mo : mean, qb = left ? qo : m2;
    const float nn = na + nb, d = mb - ma;
    mean = nn > 0.f ? ma + d * (nb / nn) : 0.f;
    m2 = nn > 0.f ? qa + qb + d * d * (na * nb / nn) : 0.f;
    n = nn;
  }
  __syncthreads();  // every lane of the workgroup has read entry 0 of its channels (sub 0 reads tile 0) before anybody overwrites it
  if (sub == 0 && t / kHdMergeLanes < GC) {
    part[2 * (size_t)ch] = mean;
    part[2 * (size_t)ch + 1] = m2;
  }
}

// ---- launch 3 -----------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kHdThreads) void heads_l3_kernel(HdArgs A, int merged) {
  __shared__ __attribute__((aligned(16))) float xs[kHdTok3 * kHdStride];
  __shared__ float ab[2 * kHdC];
  __shared__ unsigned ck[kHdC];
  __shared__ float red[4 * 2 * 256];  // [wave][row tile][row j][token i]
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tps = A.N / kHdTok3, tile = blockIdx.x, b = tile / tps, q0 = (tile - b * tps) * kHdTok3;
  const int g = blockIdx.y, GC = A.G * kHdC;
  const int c = lane & 15, kg = lane >> 4;
  const int njt = A.rows > 16 ? 2 : 1;
  // this wave's share of the output layer: contraction indices 64 w .. 64 w + 63; B operand = rows of W3 as stored
  f32x4 wb[2][4];
#pragma unroll
  for (int jt = 0; jt < 2; ++jt) {
    const int row = 16 * jt + c;
#pragma unroll
    for (int m = 0; m < 4; ++m)
      wb[jt][m] = row < A.rows ? hd_ld4(A.w3 + ((size_t)g * A.rows + row) * kHdC + 64 * w + 16 * m + 4 * kg) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int tq = lane & 3, kk = lane >> 2;  // pre2 tile: channels g*256 + 64 w + 16 it + kk, tokens q0 + 4 tq ..
  const size_t base = ((size_t)b * GC + g * kHdC) * A.N + q0 + 4 * tq;
  f32x4 v[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) v[it] = hd_ld4(A.pre2 + base + (size_t)(64 * w + 16 * it + kk) * A.N);
  const BnRng rg = bn_rng_of(A.p2, A.salt2, 0, A.rng_state);
  const HdBn S = {hd_part(A, 1), A.gamma2, A.beta2, A.running_mean2, A.running_var2, A.save_mean2, A.save_invstd2, A.counters2[g]};
  hd_bn_prepare(S, A.B * (A.N / kHdTok), GC, g, tid, A.eps, A.momentum, rg, blockIdx.x == 0, ab, ck, merged);
  __syncthreads();
  const int e0 = b * A.N + q0 + 4 * tq;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int k = 64 * w + 16 * it + kk;
    const f32x4 h = hd_act4(v[it], ab[2 * k], ab[2 * k + 1], rg, ck[k], e0);
    hd_st4(A.h2 + base + (size_t)k * A.N, h);
#pragma unroll
    for (int e = 0; e < 4; ++e) xs[(4 * tq + e) * kHdStride + k] = h[e];
  }
  __syncthreads();
  f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const f32x4 a = hd_ld4(xs + c * kHdStride + 64 * w + 16 * m + 4 * kg);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], wb[0][m][e], acc[0], 0, 0, 0);
      if (njt > 1) acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], wb[1][m][e], acc[1], 0, 0, 0);
    }
  }
  // acc[jt][r] = this wave's part of y[token 4 kg + r][row 16 jt + c]
#pragma unroll
  for (int jt = 0; jt < 2; ++jt)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[((w * 2 + jt) * 16 + c) * 16 + 4 * kg + r] = acc[jt][r];
  __syncthreads();
  for (int o = tid; o < njt * 256; o += kHdThreads) {
    const int jt = o >> 8, j = (o >> 4) & 15, i = o & 15, row = 16 * jt + j;
    if (row >= A.rows) continue;
    const int at = (jt * 16 + j) * 16 + i;
    const float s = (red[at] + red[512 + at]) + (red[1024 + at] + red[1536 + at]);
    A.y[(((size_t)b * A.G + g) * A.rows + row) * A.N + q0 + i] = s + A.b3[g * A.rows + row];
  }
}

// ---- the learned position embedding of a box (PositionEmbeddingLearned) -----------------------------------------------------------
// Phases of a workgroup (16 tokens): (1) the coordinates' first and second moments over ALL B*N tokens, every workgroup for
// itself — sums of the offsets from token 0 (the same reference everywhere: what is left to cancel in E[dd] - E[d]E[d] is the
// spread of the boxes, not their distance from the origin), per thread in fp64, across the workgroup in fp32 (quads by DPP, one
// lane per quad through LDS, the wave that owns a moment sums its 64 values); (2) thread = hidden channel: mean and variance of
// w . x from the moments (fp64: 6 + 21 terms), the affine map, the channel's 16 hidden values; (3) the second convolution on the
// matrix unit.  Everything phase 2 reads from global memory is requested before phase 1 starts.
typedef vdetr_posmlp_desc PmArgs;
constexpr int kPmMaxIn = 8;

template <int CIN>
__global__ __launch_bounds__(kHdThreads) void pos_mlp_kernel(PmArgs A) {
  constexpr int kMom = CIN + CIN * (CIN + 1) / 2;  // sums + upper triangle of the second moments
  __shared__ __attribute__((aligned(16))) float xs[kHdTok3 * kHdStride];
  __shared__ float momq[kMom][64];
  __shared__ float mom[kMom];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tps = A.N / kHdTok3, tile = blockIdx.x, b = tile / tps, q0 = (tile - b * tps) * kHdTok3;
  const int col0 = 64 * w, T = A.B * A.N;
  HdRing R;
  hd_w_begin(A.w2t, col0, lane, R);
  // what phase 2 needs: this thread's channel of the first convolution and BatchNorm, the workgroup's 16 tokens
  const int ch = tid;
  float wv[CIN], xt[kHdTok3][CIN], ref[CIN];
#pragma unroll
  for (int i = 0; i < CIN; ++i) wv[i] = A.w1[(size_t)ch * CIN + i];
  const float gam = A.gamma[ch], bet = A.beta[ch], bias1 = A.b1 ? A.b1[ch] : 0.f;
#pragma unroll
  for (int i = 0; i < CIN; ++i) ref[i] = A.x[i];
#pragma unroll
  for (int t = 0; t < kHdTok3; ++t)
#pragma unroll
    for (int i = 0; i < CIN; ++i) xt[t][i] = A.x[((size_t)b * A.N + q0 + t) * CIN + i];  // (wave-uniform addresses: scalar loads)
  // ---- phase 1 ----
  double s[kMom];
#pragma unroll
  for (int i = 0; i < kMom; ++i) s[i] = 0.0;
  for (int t = tid; t < T; t += kHdThreads) {
    float xv[CIN];
#pragma unroll
    for (int i = 0; i < CIN; ++i) xv[i] = A.x[(size_t)t * CIN + i];
    double dv[CIN];
#pragma unroll
    for (int i = 0; i < CIN; ++i) dv[i] = (double)(xv[i] - ref[i]);  // (the difference of two nearby floats: exact or nearly so)
    int at = CIN;
#pragma unroll
    for (int i = 0; i < CIN; ++i) {
      s[i] += dv[i];
#pragma unroll
      for (int j = i; j < CIN; ++j) s[at++] += dv[i] * dv[j];
    }
  }
#pragma unroll
  for (int i = 0; i < kMom; ++i) {
    float r = (float)s[i];
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
  // ---- phase 2 ----
  {
    const double inv = 1.0 / (double)T;
    double mu[CIN];  // mean offset from the reference token
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
    const float invstd = rsqrtf(varf + A.eps);
    const float a = gam * invstd, sh = bet - meanf * a;
    if (blockIdx.x == 0) {
      A.save_mean[ch] = meanf;
      A.save_invstd[ch] = invstd;
      if (A.running_mean) {
        const float m = A.momentum, n = (float)T;
        A.running_mean[ch] = (1.f - m) * A.running_mean[ch] + m * (meanf + bias1);
        A.running_var[ch] = (1.f - m) * A.running_var[ch] + m * varf * (n / (n > 1.f ? n - 1.f : 1.f));
      }
      if (tid == 0 && A.counter) A.counter[0] += 1;
    }
    f32x4 hp[4], ha[4];
#pragma unroll
    for (int i = 0; i < kHdTok3; ++i) {
      float h = 0.f;
#pragma unroll
      for (int k = 0; k < CIN; ++k) h = fmaf(xt[i][k], wv[k], h);
      const float act = fmaxf(h * a + sh, 0.f);
      hp[i >> 2][i & 3] = h;
      ha[i >> 2][i & 3] = act;
      xs[i * kHdStride + ch] = act;
    }
    const size_t o = ((size_t)b * kHdC + ch) * A.N + q0;
#pragma unroll
    for (int i = 0; i < 4; ++i) { hd_st4(A.hpre + o + 4 * i, hp[i]); hd_st4(A.hact + o + 4 * i, ha[i]); }
  }
  __syncthreads();
  // ---- phase 3 ----
  f32x4 acc[1][4];
#pragma unroll
  for (int u = 0; u < 4; ++u) acc[0][u] = f32x4{0.f, 0.f, 0.f, 0.f};
  hd_w_run<1>(xs, A.w2t, col0, lane, R, acc);
  const int c = lane & 15, kg = lane >> 4, colq = col0 + 4 * c;
  const f32x4 bias = A.b2 ? hd_ld4(A.b2 + colq) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = 4 * kg + r;
    hd_st4(A.out + ((size_t)(q0 + i) * A.B + b) * kHdC + colq, f32x4{acc[0][0][r], acc[0][1][r], acc[0][2][r], acc[0][3][r]} + bias);
  }
}

}  // namespace vdetr

using namespace vdetr;

#define HD_ALIGNED(p) ((((uintptr_t)(p)) & 15) == 0)

extern "C" size_t vdetr_heads_workspace_bytes(int B, int N, int G) {
  if (B <= 0 || N <= 0 || G <= 0) return 0;
  return (size_t)2 * ((size_t)B * N / kHdTok) * G * kHdC * 2 * sizeof(float);
}

extern "C" int vdetr_heads_fwd_f32(const vdetr_heads_desc* d, vdetr_stream_t stream) {
  VDETR_REQUIRE(d != nullptr, "heads_fwd: null descriptor");
  VDETR_REQUIRE(d->B > 0 && d->N > 0 && d->N % kHdTok == 0, "heads_fwd: B=%d, N=%d: N must be a positive multiple of %d", d->B, d->N, kHdTok);
  VDETR_REQUIRE(d->G >= 1 && d->G <= 8 && d->rows >= 1 && d->rows <= 32, "heads_fwd: G=%d outside [1, 8] or rows=%d outside [1, 32]", d->G, d->rows);
  VDETR_REQUIRE((long)d->B * d->N / kHdTok3 <= 2147483647L, "heads_fwd: too many tokens");
  VDETR_REQUIRE(d->x && d->w1t && d->w2t && d->w3 && d->b3 && d->gamma1 && d->beta1 && d->gamma2 && d->beta2, "heads_fwd: null operand");
  VDETR_REQUIRE(d->pre1 && d->h1 && d->pre2 && d->h2 && d->y && d->save_mean1 && d->save_invstd1 && d->save_mean2 && d->save_invstd2 && d->workspace,
                "heads_fwd: null output or workspace");
  VDETR_REQUIRE((d->running_mean1 != nullptr) == (d->running_var1 != nullptr) && (d->running_mean1 != nullptr) == (d->running_mean2 != nullptr) &&
                    (d->running_mean2 != nullptr) == (d->running_var2 != nullptr), "heads_fwd: the four running statistics go together");
  VDETR_REQUIRE(d->p1 >= 0.f && d->p1 < 1.f && d->p2 >= 0.f && d->p2 < 1.f, "heads_fwd: dropout rates %f, %f outside [0,1)", d->p1, d->p2);
  VDETR_REQUIRE(HD_ALIGNED(d->x) && HD_ALIGNED(d->w1t) && HD_ALIGNED(d->w2t) && HD_ALIGNED(d->w3) && HD_ALIGNED(d->pre1) && HD_ALIGNED(d->h1) &&
                    HD_ALIGNED(d->pre2) && HD_ALIGNED(d->h2) && HD_ALIGNED(d->workspace), "heads_fwd: operands must be 16-B aligned");
  hipStream_t st = (hipStream_t)stream;
  const int tiles = d->B * d->N / kHdTok;
  const int GC = d->G * kHdC;
  const int merged = tiles > kHdMergeAbove ? 1 : 0;
  float* part = reinterpret_cast<float*>(d->workspace);
  hipLaunchKernelGGL(heads_l1_kernel, dim3(tiles, d->G), dim3(kHdThreads), 0, st, *d);
  if (merged) hipLaunchKernelGGL(heads_merge_kernel, dim3(ceil_div(GC * kHdMergeLanes, kHdThreads)), dim3(kHdThreads), 0, st, part, tiles, GC);
  hipLaunchKernelGGL(heads_l2_kernel, dim3(tiles, d->G), dim3(kHdThreads), 0, st, *d, merged);
  if (merged) hipLaunchKernelGGL(heads_merge_kernel, dim3(ceil_div(GC * kHdMergeLanes, kHdThreads)), dim3(kHdThreads), 0, st, part + (size_t)tiles * GC * 2, tiles, GC);
  hipLaunchKernelGGL(heads_l3_kernel, dim3(d->B * d->N / kHdTok3, d->G), dim3(kHdThreads), 0, st, *d, merged);
  return check_launch("heads_fwd");
}

extern "C" int vdetr_pos_mlp_fwd_f32(const vdetr_posmlp_desc* d, vdetr_stream_t stream) {
  VDETR_REQUIRE(d != nullptr, "pos_mlp_fwd: null descriptor");
  VDETR_REQUIRE(d->B > 0 && d->N > 0 && d->N % kHdTok3 == 0, "pos_mlp_fwd: B=%d, N=%d: N must be a positive multiple of %d", d->B, d->N, kHdTok3);
  VDETR_REQUIRE(d->cin >= 1 && d->cin <= kPmMaxIn, "pos_mlp_fwd: cin=%d outside [1, %d]", d->cin, kPmMaxIn);
  VDETR_REQUIRE(d->x && d->w1 && d->gamma && d->beta && d->w2t && d->hpre && d->hact && d->save_mean && d->save_invstd && d->out, "pos_mlp_fwd: null pointer");
  VDETR_REQUIRE((d->running_mean != nullptr) == (d->running_var != nullptr), "pos_mlp_fwd: running_mean and running_var go together");
  VDETR_REQUIRE(HD_ALIGNED(d->w2t) && HD_ALIGNED(d->b2) && HD_ALIGNED(d->hpre) && HD_ALIGNED(d->hact) && HD_ALIGNED(d->out), "pos_mlp_fwd: operands must be 16-B aligned");
  const dim3 grid(d->B * d->N / kHdTok3), block(kHdThreads);
  hipStream_t st = (hipStream_t)stream;
  switch (d->cin) {
    case 1: hipLaunchKernelGGL(pos_mlp_kernel<1>, grid, block, 0, st, *d); break;
    case 2: hipLaunchKernelGGL(pos_mlp_kernel<2>, grid, block, 0, st, *d); break;
    case 3: hipLaunchKernelGGL(pos_mlp_kernel<3>, grid, block, 0, st, *d); break;  // (key positions: pos_for_key)
    case 4: hipLaunchKernelGGL(pos_mlp_kernel<4>, grid, block, 0, st, *d); break;
    case 5: hipLaunchKernelGGL(pos_mlp_kernel<5>, grid, block, 0, st, *d); break;
    case 6: hipLaunchKernelGGL(pos_mlp_kernel<6>, grid, block, 0, st, *d); break;  // (box centre + size: the decoder's query position)
    case 7: hipLaunchKernelGGL(pos_mlp_kernel<7>, grid, block, 0, st, *d); break;
    default: hipLaunchKernelGGL(pos_mlp_kernel<8>, grid, block, 0, st, *d); break;
  }
  return check_launch("pos_mlp_fwd");
}
