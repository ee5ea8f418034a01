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
int launch_attn_bwd_box4(const AttnParams& P, int grid, hipStream_t st);  // attn_bwd_box4.hip (dS given)


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
    P.probs_out[e] = g.p_drop;
    if (have_grad) P.ds_out[e] = g.ds * P.scale;
  }
}

// ---- RPE kernel, matrix-core variant (default) --------------------------------------------------------------------
// Measured on MI355X (nQ=1024, nK=4096, B=1): every variant that adds into the LDS histogram with
// ds_add_f32 takes time proportional to its number of atomic LANE operations (~4 cycles each, CU-wide serial):
// plain per-lane atomics 6.5 ms, wave-aggregated (variant 1) 1.29 ms — no matter how cheap the aggregation itself is.
// This variant therefore removes the atomics altogether:
//   * wave w of the workgroup owns VERTEX w: it is the only writer of table i = w in the workgroup's LDS histogram,
//     (plain read/add/write instead of atomics is NOT possible: neighbouring cells' 8-corner footprints overlap, so two
//     groups of one update can hit the same bin);
//   * the wave-level aggregation is a small matrix product on the matrix cores,
//         G[group][value] = sum over the 64 lanes  M[group][lane] * V[lane][value]
//     with M the 0/1 membership of a lane (pair) in a group (= distinct lookup cell among the wave's 64 pairs, <= 16
//     per round, 7.9 on average) and V the lane's products weight(corner) * dS(head): 16 v_mfma_f32_16x16x4_f32 per
//     16 values for ALL groups at once — exact fp32 (0/1 factors, fixed summation order: run-to-run deterministic).
//       lane l as A operand: row = group l&15, k-slot l>>4 -> membership of pair 4s + (l>>4)   (cells via an LDS strip)
//       lane l as B operand: k-slot l>>4, column = value l&15 -> V[pair 4s + (l>>4)][l&15]      (values via an LDS strip)
//       result: lane l holds value l&15 of groups 4*(l>>4)+r.
//   * every wave recomputes the cheap element-wise softmax backward of the 64 pairs it looks at (8x redundant, ~100
//     VALU ops against ~600 + 32 MFMA of vertex work); wave 0 writes P~ / dS.  Because the waves of a workgroup
//     drift apart, P~ / dS go to SEPARATE output tensors (in-place would let a slow wave read overwritten scores).
constexpr int kMmStripFloats = kWave + kWave * 16;  // cell ids + 64 x 16 values

// Workgroup shapes (VERTS vertex tables in the LDS histogram, WPV waves per vertex):
//   <8,1>  512 threads, 128 KB histogram + 8 strips: one wave per vertex, 2 waves per SIMD;
//   <4,4> 1024 threads,  64 KB histogram + 16 strips: a query is covered by TWO workgroups (vertices 0-3 / 4-7), the four
//         waves of a vertex take every fourth 64-key chunk and share the table through the (integer) atomics.  Same LDS
//         budget, twice the waves per SIMD to hide the LDS / MFMA latency chain of a chunk, half the partial-table bytes.
// FIXED: the LDS histogram is int32 fixed point.  Measured on MI355X (tools/kernel_bench.py --lds): ds_add_f32 runs
// at 0.37 lane-updates/clk/CU regardless of address conflicts, ds_add_u32 at 2.3 — the float atomics alone were
// ~0.5 ms of this kernel.  The scale is exact-safe: per bin, sum |contribution| <= sum over the workgroup's queries of
// sum_k P(q,k) * |dP - delta| <= queries_per_wg * 2 * drop_scale * max|dP~|  (weights <= 1, softmax rows sum to 1,
// max taken over the workgroup's own rows in a prologue pass), so with S = 2^floor(log2(2^30 / bound)) no partial sum
// can overflow, and the resolution (bound * 2^-30) is ~1e-5 of a typical group sum.  Integer adds also make the
// histogram independent of the order of the updates.
// SPLIT16: the product runs on the bf16 matrix pipe instead of v_mfma_f32_16x16x4_f32.  Measured on MI355X
// (tools/kernel_bench.py --issue): an fp32 MFMA occupies its SIMD for 32 cycles and VALU instructions of other waves do
// NOT issue underneath it, so the 32 fp32 MFMAs of a chunk cost as much as ~290 VALU instructions.  The 0/1 membership
// is exact in bf16; a value v is sent as hi = v & 0xFFFF0000 and lo = bf16(v - hi) packed in one 32-bit word (relative
// error <= 2^-16, far inside the 1e-3 gradient budget), and G = M*hi + M*lo takes 8 v_mfma_f32_16x16x32_bf16 (16 cycles
// each) with fp32 accumulation; hi and lo ride in the same MFMA as two k-entries of one pair.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
// VLOOP: a wave takes every (VERTS*WPV)-th chunk and walks ALL the workgroup's vertices for it, so the element-wise
// softmax backward (loads, exp, dropout hash, stores: ~85 of ~385 VALU instructions per (chunk, vertex)) is done once
// per chunk and workgroup instead of once per vertex.
template <bool FIXED, int VERTS, int WPV, bool SPLIT16, bool VLOOP>
__global__ __launch_bounds__(VERTS * WPV * kWave) void attn_bwd_scores_rpe_mm_kernel(AttnParams P) {
  constexpr int kThreads = VERTS * WPV * kWave;
  constexpr int kChunkStride = VLOOP ? VERTS * WPV : WPV;
  constexpr int kSplit = kRpeVerts / VERTS;  // workgroups per query
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [dTable copy VERTS*T^3*4][strips]
  // every query is an axis-aligned box (count of the others left by the delta launch): attn_bwd_box_kernel does this launch
  if (P.box_path && P.bwd_aux[4] == 0 && P.bwd_aux[5] != 0) return;  // (word 5: the delta launch did look at the vertices)
  attn_load_rng(P);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int vloc = VLOOP ? 0 : wv % VERTS, cslot = VLOOP ? wv : wv / VERTS;
  const int part = blockIdx.x % kSplit, wg = blockIdx.x / kSplit, nwg = gridDim.x / kSplit;
  const int w = part * VERTS + vloc;  // (first) vertex index
  const int T = P.T, TT = T * T, T3 = TT * T;
  const int table_floats = VERTS * T3 * 4;
  const int items = P.B * P.nQ;
  for (int i = tid; i < table_floats; i += kThreads) smem[i] = 0.f;
  float fix_scale = 1.f, fix_inv = 1.f;
  // Dynamic query distribution (P.bwd_aux != NULL): the workgroups of a vertex half pull queries from a device counter,
  // so a CU that is busy elsewhere (the side-stream FPS) costs 1/8 of a round instead of a whole one.  The histogram
  // scale must then hold for ANY set of queries: |dP~| <= |dO row| * |V row| (Cauchy-Schwarz) with the two maxima left
  // in aux[0..1] by vdetr_attn_delta_f32, and a workgroup stops after `cap` queries.
  const bool dynamic = P.bwd_aux != nullptr;
  const int per_wg = (items + nwg - 1) / nwg;
  const int cap = dynamic ? bwd_query_cap(per_wg) : per_wg;
  if (FIXED && dynamic) {
    const float dmax = sqrtf(__uint_as_float(P.bwd_aux[0]) * __uint_as_float(P.bwd_aux[1]));
    const float bound = 2.f * P.drop_scale * dmax * (float)cap;
    if (bound > 0.f && bound < INFINITY) {
      const int e = 30 - (int)ceilf(__log2f(bound) + 1e-3f);  // 2^e * bound <= 2^30
      fix_scale = ldexpf(1.f, e);
      fix_inv = ldexpf(1.f, -e);
    }
  } else if (FIXED) {
    // max |dP~| over this workgroup's rows (the 4 head rows of a query are contiguous: one 16-B aligned run)
    float m = 0.f;
    for (int item = wg; item < items; item += nwg) {
      const float* base = P.dprob + (size_t)item * 4 * P.nK;
      if ((reinterpret_cast<uintptr_t>(base) & 15) == 0) {
        const f32x4* b4 = reinterpret_cast<const f32x4*>(base);
        for (int i = tid; i < P.nK; i += kThreads) {
          const f32x4 v = b4[i];
          m = fmaxf(m, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
        }
      } else {
        for (int i = tid; i < 4 * P.nK; i += kThreads) m = fmaxf(m, fabsf(base[i]));
      }
    }
    m = wave_allmax_f32(m);
    float* red = smem + table_floats;  // strip memory, not yet in use
    if (lane == 0) red[wv] = m;
    __syncthreads();
    float dmax = 0.f;
#pragma unroll
    for (int i = 0; i < VERTS * WPV; ++i) dmax = fmaxf(dmax, red[i]);
    const float bound = 2.f * P.drop_scale * dmax * (float)per_wg;
    if (bound > 0.f && bound < INFINITY) {
      const int e = 30 - (int)ceilf(__log2f(bound) + 1e-3f);  // 2^e * bound <= 2^30
      fix_scale = ldexpf(1.f, e);
      fix_inv = ldexpf(1.f, -e);
    }
  }
  __syncthreads();
  // wave-private strip, accessed as int throughout (values are bit-cast) so that the scoreboard and the value
  // tile, which share memory, are never seen through two different types by the compiler's alias analysis
  int* cellbuf = reinterpret_cast<int*>(smem + table_floats + wv * kMmStripFloats);  // 64 ints
  int* vbuf = cellbuf + kWave;                                                        // 64 x 16 values
  int* scoreboard = vbuf;                                                             // T^3 <= 1024 ints, aliases vbuf
  const bool rot = P.cos_sin != nullptr;
  const bool writer = w == 0;  // the waves that look at vertex 0 store P~ / dS of the chunks they visit
  const int nchunks = (P.nK + kWave - 1) / kWave;
  const int kk = lane >> 4, c15 = lane & 15;
  int off[2];  // bin offsets of this lane's output column for the two 16-value tiles (tile jt = corners 4jt..4jt+3)
#pragma unroll
  for (int jt = 0; jt < 2; ++jt) {
    const int corner = jt * 4 + (c15 >> 2);
    off[jt] = (((corner >> 2) & 1) * TT + ((corner >> 1) & 1) * T + (corner & 1)) * 4 + (c15 & 3);
  }
  const int wr_sw = lane >> 1;  // 16-B blocks of a strip row are rotated by lane>>1: conflict-free b128 writes AND b32 reads

  // operands of one chunk (64 consecutive keys, one per lane); fetched one visit ahead
  struct ChunkOps {
    float s[4], d[4], kx, ky, kz;
    unsigned char masked;
  };
  // Buffer addressing: a wave-uniform resource descriptor per row block (SGPRs) + one 32-bit lane offset.  With flat
  // pointers the compiler keeps a 64-bit per-lane address for each of the 16 streams (32 VGPRs, spilled under the
  // 128-VGPR budget of the 16-wave shape); reads past the end of a block return 0.
  using rsrc_t = __amdgpu_buffer_rsrc_t;
  auto make_rsrc = [](const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
  };
  auto ldf = [](rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
  };
  const int rowbytes = P.nK * 4;
  const bool dsg = P.ds_given != 0;  // `dprob` holds dS already (attn_bwd_kv.hip): one stream to read, nothing to store
  auto fetch = [&](rsrc_t rs, rsrc_t rd, rsrc_t rx, rsrc_t rm, bool has_mask, int chunk, ChunkOps& o) {
    const int key = chunk * kWave + lane;
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      if (!dsg) o.s[h] = ldf(rs, key * 4, h * rowbytes);
      o.d[h] = ldf(rd, key * 4, h * rowbytes);
    }
    o.kx = ldf(rx, key * 12, 0); o.ky = ldf(rx, key * 12 + 4, 0); o.kz = ldf(rx, key * 12 + 8, 0);
    o.masked = has_mask ? __builtin_amdgcn_raw_buffer_load_b8(rm, key, 0, 0) : 0;
  };

  __shared__ int next_item;
  for (int it = 0; it < cap; ++it) {
    int item = wg + it * nwg;
    if (dynamic) {
      if (tid == 0) next_item = (int)atomicAdd(const_cast<unsigned*>(P.bwd_aux) + 2 + part, 1u);
      __syncthreads();
      item = next_item;
      __syncthreads();  // everyone holds the item before thread 0 fetches the next one
    }
    if (item >= items) break;
    const int b = item / P.nQ, q = item - b * P.nQ;
    const size_t row0 = ((size_t)b * P.nQ + q) * 4;
    // per-item constants are wave-uniform: pin them to SGPRs (the loads are vector loads because the kernel also stores)
    auto uni = [](float x) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x))); };
    float lse[4], delta[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) { lse[h] = P.ds_given ? 0.f : uni(P.lse[row0 + h]); delta[h] = P.ds_given ? 0.f : uni(P.delta[row0 + h]); }
    constexpr int kVL = VLOOP ? VERTS : 1;
    const float* vp = P.vertices + ((size_t)b * P.nQ + q) * 24 + w * 3;
    float vxs[kVL], vys[kVL], vzs[kVL];
#pragma unroll
    for (int vl = 0; vl < kVL; ++vl) { vxs[vl] = uni(vp[vl * 3]); vys[vl] = uni(vp[vl * 3 + 1]); vzs[vl] = uni(vp[vl * 3 + 2]); }
    const float rc = rot ? uni(P.cos_sin[((size_t)b * P.nQ + q) * 2]) : 1.f;
    const float rs = rot ? uni(P.cos_sin[((size_t)b * P.nQ + q) * 2 + 1]) : 0.f;
    const rsrc_t rsc = make_rsrc(P.scores + row0 * P.nK, 4u * rowbytes), rd = make_rsrc(P.dprob + row0 * P.nK, 4u * rowbytes);
    const rsrc_t rp = make_rsrc(P.probs_out + row0 * P.nK, 4u * rowbytes), rg = make_rsrc(P.ds_out + row0 * P.nK, 4u * rowbytes);
    const rsrc_t rx = make_rsrc(P.xyz + (size_t)b * P.nK * 3, 3u * rowbytes);
    const bool has_mask = P.mask_kind == VDETR_MASK_BOOL && !dsg;
    const rsrc_t rm = make_rsrc(has_mask ? reinterpret_cast<const unsigned char*>(P.mask) + ((size_t)b * P.nQ + q) * P.nK
                                         : reinterpret_cast<const unsigned char*>(P.xyz), has_mask ? (unsigned)P.nK : 0u);
    ChunkOps ops, nxt;
    fetch(rsc, rd, rx, rm, has_mask, cslot, ops);

    for (int chunk = cslot; chunk < nchunks; chunk += kChunkStride) {
      if (chunk + kChunkStride < nchunks) fetch(rsc, rd, rx, rm, has_mask, chunk + kChunkStride, nxt);
      // ---- element-wise softmax backward of this lane's pair (recomputed by every wave; vertex 0's waves store) ---
      const int key = chunk * kWave + lane;
      const bool valid = key < P.nK;
      float ds[4];
      if (dsg) {
#pragma unroll
        for (int h = 0; h < 4; ++h) ds[h] = valid ? ops.d[h] * fix_scale : 0.f;
      } else {
        uint4 rnd = make_uint4(0xFFFFu, 0xFFFFu, 0xFFFFu, 0xFFFFu);
        if (P.drop_thresh) rnd = attn_rand4(P, b, q, key, 0);
#pragma unroll
        for (int h = 0; h < 4; ++h) {
          const bool keep = pick4(rnd, h) >= P.drop_thresh;
          const ScoreGrad g = score_grad(ops.s[h], lse[h], keep, P.drop_scale, true, ops.d[h], delta[h], ops.masked != 0);
          if (writer && valid) {
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(g.p_drop), rp, key * 4, h * rowbytes, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(g.ds * P.scale), rg, key * 4, h * rowbytes, 0);
          }
          ds[h] = valid ? g.ds * fix_scale : 0.f;  // fix_scale is a power of two (1 for the float histogram): exact
        }
      }
#pragma unroll
      for (int vl = 0; vl < kVL; ++vl) {
      if (VLOOP) __builtin_amdgcn_sched_barrier(0);  // keep the vertices' geometry from being hoisted together (VGPRs)
      float* mytab = smem + (size_t)(VLOOP ? vl : vloc) * T3 * 4;
      // ---- lookup geometry of (pair, vertex) ------------------------------------------------------------------------
      float dx = vxs[vl] - ops.kx, dy = vys[vl] - ops.ky, dz = vzs[vl] - ops.kz;
      if (rot) rpe_rotate(dx, dy, rc, rs);
      const AxisTap ax = rpe_axis(dx, P), ay = rpe_axis(dy, P), az = rpe_axis(dz, P);
      const int cell = rpe_cell(ax, ay, az, T);
      const float w00 = az.wa * ay.wa, w01 = az.wa * ay.wb, w10 = az.wb * ay.wa, w11 = az.wb * ay.wb;
      const float wgt[8] = {w00 * ax.wa, w00 * ax.wb, w01 * ax.wa, w01 * ax.wb,
                            w10 * ax.wa, w10 * ax.wb, w11 * ax.wa, w11 * ax.wb};
      // ---- groups = distinct cells among the 64 pairs, without a serial loop: every lane posts its id on a
      // scoreboard slot indexed by its cell; whoever is read back is the group's leader, and a group's index is the
      // rank of its leader among the leaders.
      __builtin_amdgcn_wave_barrier();
      scoreboard[cell] = lane;
      __builtin_amdgcn_wave_barrier();
      const int leader = scoreboard[cell];
      const unsigned long long lmask = __ballot(leader == lane);
      const int ngroups = __popcll(lmask);
      const int gidx = __popcll(lmask & ((1ull << leader) - 1ull));
      __builtin_amdgcn_wave_barrier();
      if (leader == lane) cellbuf[gidx] = cell;  // cell of every group

      for (int g0 = 0; g0 < ngroups; g0 += 16) {  // 16 groups per round (one round in 95 % of the chunks)
        // group index of all 64 pairs through the (currently free) value tile, laid out so that k-slot kk reads pairs
        // kk, 4+kk, 8+kk, ... as 4 x 16 bytes; re-read per round so that the 16 registers are not live across the round
        __builtin_amdgcn_wave_barrier();
        vbuf[(lane & 3) * 16 + (lane >> 2)] = gidx;
        __builtin_amdgcn_wave_barrier();
        int cr[16];
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) {
          const int4 v4 = *reinterpret_cast<const int4*>(vbuf + kk * 16 + t4 * 4);
          cr[t4 * 4] = v4.x; cr[t4 * 4 + 1] = v4.y; cr[t4 * 4 + 2] = v4.z; cr[t4 * 4 + 3] = v4.w;
        }
        int gc[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int g = g0 + 4 * kk + r;
          gc[r] = g < ngroups ? cellbuf[g] : -1;
        }
        // membership operand, built once per round.  k-slot (kk, s) of every MFMA = pair 4s + kk.
        float am[16];
        i32x4 am16[4];
        if (SPLIT16) {
          // a strip word carries (hi, lo) of one value = two consecutive bf16 k-entries, so the words feed the B operand
          // as they are and the membership of a pair is simply duplicated into both halves of an A register
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int m = 0; m < 4; ++m) am16[t][m] = cr[4 * t + m] == g0 + c15 ? 0x3F803F80 : 0;  // bf16 (1.0, 1.0)
        } else {
#pragma unroll
          for (int s2 = 0; s2 < 16; ++s2) am[s2] = cr[s2] == g0 + c15 ? 1.f : 0.f;
        }
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int cc = 0; cc < 4; ++cc) {
            const float wc = wgt[jt * 4 + cc];
            int wd[4];
            if (SPLIT16) {
              typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
              for (int hp = 0; hp < 2; ++hp) {  // packed fp32: two heads per v_pk_mul / v_pk_add
                const f32x2 v = f32x2{wc, wc} * f32x2{ds[2 * hp], ds[2 * hp + 1]};
                const f32x2 hi = {__int_as_float(__float_as_int(v[0]) & 0xFFFF0000), __int_as_float(__float_as_int(v[1]) & 0xFFFF0000)};
                const f32x2 lo = v - hi;  // exact
#pragma unroll
                for (int e = 0; e < 2; ++e)
                  wd[2 * hp + e] = (int)__builtin_amdgcn_perm((unsigned)__float_as_int(v[e]), (unsigned)__float_as_int(lo[e]), 0x07060302u);
              }
            } else {
#pragma unroll
              for (int h = 0; h < 4; ++h) wd[h] = __float_as_int(wc * ds[h]);
            }
            *reinterpret_cast<int4*>(vbuf + lane * 16 + ((cc + wr_sw) & 3) * 4) = make_int4(wd[0], wd[1], wd[2], wd[3]);
          }
          __builtin_amdgcn_wave_barrier();
          int bw[16];
#pragma unroll
          for (int s2 = 0; s2 < 16; ++s2) {
            const int p = 4 * s2 + kk;
            bw[s2] = vbuf[p * 16 + (((c15 >> 2) + (p >> 1)) & 3) * 4 + (c15 & 3)];
          }
          float tot[4];
          if (SPLIT16) {
            f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int t = 0; t < 4; ++t) {  // 16 pairs x (hi, lo) per MFMA
              const i32x4 b = {bw[4 * t], bw[4 * t + 1], bw[4 * t + 2], bw[4 * t + 3]};
              acc[t & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, am16[t]),
                                                                   __builtin_bit_cast(bf16x8, b), acc[t & 1], 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) tot[r] = acc[0][r] + acc[1][r];
          } else {
            f32x4 acc[4];  // four independent chains
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) acc[ch] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s2 = 0; s2 < 16; ++s2)
              acc[s2 & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(am[s2], __int_as_float(bw[s2]), acc[s2 & 3], 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) tot[r] = (acc[0][r] + acc[1][r]) + (acc[2][r] + acc[3][r]);
          }
          // the (group, value) bins of one update are distinct, but neighbouring cells' footprints overlap: atomics
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (gc[r] >= 0) {
              float* bin = mytab + gc[r] * 4 + off[jt];
              if (FIXED) atomicAdd(reinterpret_cast<int*>(bin), __float2int_rn(tot[r]));
              else atomicAdd(bin, tot[r]);
            }
        }
      }
      }  // vertex loop
      ops = nxt;
    }
  }
  __syncthreads();
  float* dst = P.dtable_part + (size_t)blockIdx.x * table_floats;
  for (int i = tid; i < table_floats; i += kThreads)
    dst[i] = FIXED ? (float)reinterpret_cast<const int*>(smem)[i] * fix_inv : smem[i];
}

// dtable[e] += sum over workgroup copies.  Workgroup p of the scores kernel wrote the `n / nsplit` floats of table
// slice p % nsplit; blockIdx.y takes kRedSlice of the copies so that the partials are read by ~1000 blocks
constexpr int kRedSlice = 16;
// gate != NULL (bwd_kernel 2: the caller vouched for boxes and the general kernel was not launched): where the device found a query
// that is not a box, nothing filled the partial tables — the gradient is poisoned with NaN instead of being left wrong.
__global__ __launch_bounds__(256) void attn_bwd_table_reduce_kernel(const float* part, int nparts, int nsplit, int n,
                                                                    float* dtable, const unsigned* gate) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  if (gate && (gate[4] != 0 || gate[5] == 0)) {
    dtable[e] = __builtin_nanf("");
    return;
  }
  const int nsub = n / nsplit, sl = e / nsub, esub = e - sl * nsub;
  const int copies = nparts / nsplit;
  const int p0 = blockIdx.y * kRedSlice, p1 = min(copies, p0 + kRedSlice);
  float s = 0.f;
  for (int p = p0; p < p1; ++p) s += part[((size_t)p * nsplit + sl) * nsub + esub];
  unsafeAtomicAdd(dtable + e, s);
}

// LDS atomic-rate probe (tools/kernel_bench.py --lds): mode 0 ds_add_f32, 1 ds_add_u32, 2 plain read-add-write,
// 3 ds_add_f32 with all lanes on 8 addresses; 512-thread workgroups, `iters` updates per lane to scattered bins
__global__ __launch_bounds__(512) void lds_atomic_probe_kernel(int mode, int iters, float* sink) {
  __shared__ float tab[32768];
  for (int i = threadIdx.x; i < 32768; i += 512) tab[i] = 0.f;
  __syncthreads();
  unsigned a = threadIdx.x * 2654435761u;
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    a = a * 1664525u + 1013904223u;
    const unsigned slot = mode == 3 ? ((a >> 10) & 7u) * 4u : ((a >> 10) & 8191u) * 4u + (threadIdx.x & 3);
    if (mode == 0 || mode == 3) atomicAdd(&tab[slot], 1.0f);
    else if (mode == 1) atomicAdd(reinterpret_cast<unsigned*>(&tab[slot]), 1u);
    else tab[slot] += 1.0f;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 32768; i += 512) acc += tab[i];
  if (acc == -1.f) sink[0] = acc;
}

// Issue-overlap probe (tools/kernel_bench.py --issue): 512-thread workgroups, one per CU, `iters` rounds of
//   mode 10: 16 independent v_mfma_f32_16x16x4_f32        mode 11: 128 independent v_fma_f32
//   mode 12: both, interleaved in every wave              mode 13: waves 0-3 MFMA only, waves 4-7 VALU only
//   mode 14: 128 v_fma_f32 + 16 ds_read_b128
//   mode 15: 16 v_mfma_f32_16x16x32_bf16                  mode 16: waves 0-3 bf16 MFMA, waves 4-7 VALU
//   mode 17: bf16 MFMA + VALU interleaved in every wave
__global__ __launch_bounds__(512) void issue_probe_kernel(int mode, int iters, float* sink) {
  __shared__ f32x4 buf[1024];
  for (int i = threadIdx.x; i < 1024; i += 512) buf[i] = f32x4{1.f, 2.f, 3.f, 4.f};
  __syncthreads();
  const int wv = threadIdx.x >> 6;
  const bool do_mfma = mode == 10 || mode == 12 || (mode == 13 && wv < 4);
  const bool do_valu = mode == 11 || mode == 12 || mode == 14 || mode == 17 || ((mode == 13 || mode == 16) && wv >= 4);
  const bool do_bf16 = mode == 15 || mode == 17 || (mode == 16 && wv < 4);
  typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
  bf16x8 pa, pb;
#pragma unroll
  for (int i = 0; i < 8; ++i) { pa[i] = (__bf16)(1.0f + i); pb[i] = (__bf16)(0.5f * (threadIdx.x & 3)); }
  f32x4 acc[4] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
  float v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = (float)(threadIdx.x + i);
  const float a = 1.0001f, b = 0.5f;
  f32x4 l = f32x4{0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    if (do_mfma) {
#pragma unroll
      for (int s = 0; s < 16; ++s) acc[s & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[s & 3], 0, 0, 0);
    }
    if (do_bf16) {
#pragma unroll
      for (int s = 0; s < 16; ++s) acc[s & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa, pb, acc[s & 3], 0, 0, 0);
    }
    if (do_valu) {
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = __builtin_fmaf(v[i], a, b);
    }
    if (mode == 14) {
#pragma unroll
      for (int r = 0; r < 16; ++r) l += buf[(threadIdx.x * 7 + r * 64 + it) & 1023];
    }
  }
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) t += v[i];
  t += acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + l[0] + l[3];
  if (t == -1.f) sink[0] = t;
}

// delta launch: see attn_delta_body (attn_common.h); also run as the first workgroups of attn_bwd_kv.hip's preparation launch
__global__ __launch_bounds__(256) void attn_delta_kernel(DeltaArgs A) {
  __shared__ float wmax[4];
  attn_delta_body(A, (int)blockIdx.x, wmax);
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

// Workgroups of the table-gradient launches: one per CU by default; vdetr_attn_desc.table_grid lowers it so that a caller who
// runs the table gradient on a side stream leaves whole CUs to the main chain (the persistent workgroups hold every register
// and 150 KB of LDS of their CU: nothing else fits next to one).  The int32 histogram's fixed-point scale is sized for the most
// queries one workgroup may take (bwd_query_cap): a grid so small that this resolution would exceed 1e-3 of the bound is refused.
constexpr int kBwdSplit = 2;  // workgroups per query (the two z halves)
static int bwd_table_grid(const vdetr_attn_desc* d, int* grid) {
  const int cus = device_cu_count();
  int want = d->table_grid ? d->table_grid : (cus & ~1);
  VDETR_REQUIRE(want >= 2 && want <= cus && want % 2 == 0, "attn_bwd_table: table_grid %d (0 = one workgroup per CU, or an even count in 2..%d)",
                d->table_grid, cus);
  const long wgs = (long)d->B * d->nQ * kBwdSplit;
  if (wgs < want) want = (int)wgs;
  // resolution of the histogram: bound / 2^30 with bound = cap x (2 drop_scale max|dO| max|V|), against entries of the order
  // of ONE query's bound: cap / 2^30 must stay below 1e-3 (cap = 1.5 x queries per workgroup)
  const long per_wg = ((long)d->B * d->nQ + want / kBwdSplit - 1) / (want / kBwdSplit);
  VDETR_REQUIRE(bwd_query_cap((int)per_wg) <= (1 << 20), "attn_bwd_table: %d workgroups for %ld queries leave the fixed-point "
                "histogram a resolution above 1e-3 (at most ~700k queries per workgroup)", want, (long)d->B * d->nQ);
  *grid = want;
  return VDETR_OK;
}

extern "C" size_t vdetr_attn_bwd_workspace_bytes(const vdetr_attn_desc* d) {
  if (!d || !d->table) return 0;
  const size_t table_floats = (size_t)kRpeVerts * d->table_size * d->table_size * d->table_size * 4;
  const long wgs = (long)d->B * d->nQ;  // (sized for the default grid: independent of d->table_grid)
  const long cus = device_cu_count();
  return (size_t)(wgs < cus ? wgs : cus) * table_floats * sizeof(float) + 256;  // partial tables + alignment
}

template <bool FIXED, int VERTS, int WPV, bool SPLIT16, bool VLOOP>
static int launch_mm(const AttnParams& P, int grid, size_t lds, hipStream_t st) {
  if (int e = set_lds(attn_bwd_scores_rpe_mm_kernel<FIXED, VERTS, WPV, SPLIT16, VLOOP>, lds, "attn_bwd_scores")) return e;
  hipLaunchKernelGGL((attn_bwd_scores_rpe_mm_kernel<FIXED, VERTS, WPV, SPLIT16, VLOOP>), dim3(grid),
                     dim3(VERTS * WPV * kWave), lds, st, P);
  return VDETR_OK;
}

static int attn_bwd_scores_impl(const vdetr_attn_desc* d, const float* scores, const float* dprob, const float* lse,
                                const float* delta, float* probs_out, float* ds_out, float* dtable, void* workspace,
                                size_t workspace_bytes, vdetr_stream_t stream, bool ds_given) {
  AttnParams P;
  if (int e = attn_fill_params(d, &P, "attn_bwd_scores")) return e;
  if (ds_given) {
    VDETR_REQUIRE(dprob && dtable && d->table && d->bwd_aux, "attn_bwd_table: needs dS, dtable, the RPE operands and bwd_aux");
  } else {
    VDETR_REQUIRE(scores && lse && probs_out, "attn_bwd_scores: null pointer");
    VDETR_REQUIRE((dprob == nullptr) == (delta == nullptr) && (dprob == nullptr) == (ds_out == nullptr),
                  "attn_bwd_scores: dprob, delta and ds_out go together");
    VDETR_REQUIRE(!dtable || (d->table && dprob), "attn_bwd_scores: dtable needs an RPE descriptor and dprob");
    VDETR_REQUIRE(!dtable || (probs_out != scores && ds_out != dprob),
                  "attn_bwd_scores: with a table gradient the outputs must not alias the inputs "
                  "(several waves re-read the scores)");
  }
  P.scores = const_cast<float*>(scores); P.dprob = const_cast<float*>(dprob);
  P.lse = const_cast<float*>(lse); P.delta = delta;
  P.probs_out = probs_out; P.ds_out = ds_out;
  P.ds_given = ds_given ? 1 : 0;
  hipStream_t st = (hipStream_t)stream;
  if (!d->table || !dtable) {  // element-wise only: P~ and dS do not depend on the look-up geometry
    const size_t total = (size_t)d->B * d->H * d->nQ * d->nK;
    const unsigned grid = (unsigned)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(attn_bwd_scores_kernel, dim3(grid), dim3(256), 0, st, P);
    return check_launch("attn_bwd_scores");
  }
  const int table_floats = kRpeVerts * P.T * P.T * P.T * 4;
  VDETR_REQUIRE(P.T * P.T * P.T <= kWave * 16, "attn_bwd_scores: table edge %d too large for the matrix-unit kernel", P.T);
  int grid = 0;
  if (int e = bwd_table_grid(d, &grid)) return e;
  if (dtable) {
    const size_t need = vdetr_attn_bwd_workspace_bytes(d);
    if (!workspace || workspace_bytes < need) {
      set_error("attn_bwd_scores: workspace %zu B < required %zu B", workspace_bytes, need);
      return VDETR_ERR_WORKSPACE;
    }
    P.dtable_part = (float*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  }
  // Two kernels over the same grid and partial-table layout; the DEVICE decides which of them works (bwd_aux words 4 / 5):
  //   attn_bwd_scores_rpe_mm_kernel: any eight vertices per query;
  //   attn_bwd_box4_kernel (dS given, table edge 10, dynamic distribution): every query's vertices are an axis-aligned box, or,
  //   with the rotation operand, a box in the frame the offsets are turned into.
  // d->bwd_kernel = 1 keeps the general kernel alone (the parity tests compare the two).
  VDETR_REQUIRE(d->bwd_kernel >= 0 && d->bwd_kernel <= 2, "attn_bwd_table: bwd_kernel %d (0, 1 or 2)", d->bwd_kernel);
  const bool box = ds_given && dtable && d->bwd_kernel != 1 && d->bwd_aux && P.T == 10;
  // bwd_kernel = 2: the caller vouches that every query's vertices are a box (they come out of a box decode): the general kernel —
  // a launch whose workgroups would all leave at once — is not put in front of the box kernel
  const bool box_only = box && d->bwd_kernel == 2;
  P.box_path = box ? 1 : 0;
  {
    const size_t lds = (size_t)table_floats / kBwdSplit * sizeof(float) + (size_t)8 * kBwdSplit * kMmStripFloats * sizeof(float);
    if (!box_only)
      if (int e = launch_mm<true, 4, 4, true, true>(P, grid, lds, st)) return e;
    if (box)
      if (int e2 = launch_attn_bwd_box4(P, grid, st)) return e2;
  }
  if (int e = check_launch("attn_bwd_scores_rpe")) return e;
  if (dtable) {
    hipLaunchKernelGGL(attn_bwd_table_reduce_kernel,
                       dim3((table_floats + 255) / 256, (grid / kBwdSplit + kRedSlice - 1) / kRedSlice), dim3(256), 0, st,
                       P.dtable_part, grid, kBwdSplit, table_floats, dtable, box_only ? d->bwd_aux : (const unsigned*)nullptr);
    return check_launch("attn_bwd_table_reduce");
  }
  return VDETR_OK;
}

extern "C" int vdetr_attn_bwd_scores_f32(const vdetr_attn_desc* d, const float* scores, const float* dprob,
                                         const float* lse, const float* delta, float* probs_out, float* ds_out,
                                         float* dtable, void* workspace, size_t workspace_bytes,
                                         vdetr_stream_t stream) {
  return attn_bwd_scores_impl(d, scores, dprob, lse, delta, probs_out, ds_out, dtable, workspace, workspace_bytes, stream, false);
}

extern "C" int vdetr_attn_bwd_table_f32(const vdetr_attn_desc* d, const float* ds, float* dtable, void* workspace,
                                        size_t workspace_bytes, vdetr_stream_t stream) {
  return attn_bwd_scores_impl(d, nullptr, ds, nullptr, nullptr, nullptr, nullptr, dtable, workspace, workspace_bytes, stream, true);
}

extern "C" int vdetr_attn_bwd_table_kernel_names(const vdetr_attn_desc* d, const char** box_kernel, const char** general_kernel) {
  VDETR_REQUIRE(d && box_kernel && general_kernel, "attn_bwd_table_kernel_names: null pointer");
  VDETR_REQUIRE(d->table, "attn_bwd_table_kernel_names: no RPE table in the descriptor");
  const int T = d->table_size;
  const bool box = d->bwd_kernel != 1 && d->bwd_aux && T == 10;
  *box_kernel = box ? "attn_bwd_box4_kernel" : nullptr;
  *general_kernel = "attn_bwd_scores_rpe_mm_kernel";
  return VDETR_OK;
}

extern "C" int vdetr_attn_delta_f32(const vdetr_attn_desc* d, const float* dout, const float* out, const float* v,
                                    float* delta, vdetr_stream_t stream) {
  VDETR_REQUIRE(d && dout && out && delta, "attn_delta: null pointer");
  VDETR_REQUIRE(d->B > 0 && d->nQ > 0 && d->H > 0, "attn_delta: empty dimension");
  const bool norms = d->bwd_aux != nullptr;
  VDETR_REQUIRE(!norms || (v && d->kind == VDETR_ATTN_SHARED_KV), "attn_delta: bwd_aux needs v and the shared-KV kind");
  const int qblocks = ceil_div((long)d->B * d->nQ, 4);
  const int vblocks = norms ? ceil_div((long)d->B * d->nK, 4 * kDeltaKeys) : 0;
  DeltaArgs A;
  attn_delta_args(d, dout, out, v, delta, &A);
  hipLaunchKernelGGL(attn_delta_kernel, dim3(qblocks + vblocks), dim3(256), 0, (hipStream_t)stream, A);
  return check_launch("attn_delta");
}

extern "C" int vdetr_selftest_lds_atomics(int mode, int iters, float* sink, vdetr_stream_t stream) {
  if (mode >= 10) hipLaunchKernelGGL(issue_probe_kernel, dim3(256), dim3(512), 0, (hipStream_t)stream, mode, iters, sink);
  else hipLaunchKernelGGL(lds_atomic_probe_kernel, dim3(256), dim3(512), 0, (hipStream_t)stream, mode, iters, sink);
  return check_launch("lds_atomic_probe");
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
