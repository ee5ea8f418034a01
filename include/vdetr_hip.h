/*
 * vdetr_hip.h — C-ABI of libvdetr_hip.so, the MI355X (gfx950) hot-path library.
 *
 * This is the drop-in boundary for the three native surfaces of the V-DETR hot path:
 *   (iii) the pointnet2 extension  (reference: third_party/pointnet2/_ext_src/src/bindings.cpp:9-22),
 *   (i)   the 3D-vertex-RPE cross attention (reference: models/vdetr_transformer.py:701-758),
 *   (ii)  the query self attention  (reference: models/vdetr_transformer.py:468,541-542 and :609-653).
 *
 * Conventions (all entry points):
 *   - plain device pointers + explicit sizes; no torch / ATen types;
 *   - all tensors are contiguous, float32 or int32, laid out exactly as the reference op lays them out;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); every launch goes on it;
 *   - the library never allocates, frees or synchronises; scratch comes from the caller
 *     (see the *_workspace_bytes queries);
 *   - return value: 0 = ok, non-zero = error; vdetr_last_error() returns a thread-local message.
 *     (the reference prints and calls exit(-1) on a launch failure: include/cuda_utils.h:32-41;
 *      argument errors there are AT_ASSERT exceptions: include/utils.h:8-28);
 *   - outputs that kernels ACCUMULATE into (all *_grad outputs) and the ball_query index tensor
 *     must be zero-filled by the caller, as the reference's torch::zeros allocations do
 *     (sampling.cpp:27-29, ball_query.cpp:23-25, group_points.cpp:50-52, interpolate.cpp:87-89).
 */
#ifndef VDETR_HIP_H
#define VDETR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VDETR_OK 0
#define VDETR_ERR_ARG 1
#define VDETR_ERR_LAUNCH 2
#define VDETR_ERR_WORKSPACE 3

typedef void* vdetr_stream_t; /* hipStream_t */

/* library / diagnostics */
int vdetr_abi_version(void); /* 3: vdetr_attn_desc carries the launch shape (table_grid, kv_waves, fwd_kernel, fwd_sched); the
                                process-wide setters vdetr_attn_bwd_table_set_grid / vdetr_attn_bwd_kv_set_waves are gone */
const char* vdetr_last_error(void);
int vdetr_ab_switches(void); /* 1: a probe build that reads VDETR_* environment switches (once per process); 0: the shipped build, which reads none */

/* ------------------------------------------------------------------------------------------------
 * (iii) pointnet2 ops.  One symbol per function of bindings.cpp:9-22.
 * ---------------------------------------------------------------------------------------------- */

/* furthest_point_sampling(points, nsamples)  — sampling.cpp:67-88, sampling_gpu.cu:73-232.
 *   xyz (b,n,3) f32 -> idx (b,m) i32.  Bit-exact with the reference kernel's result, including
 *   its origin-skip rule (x²+y²+z² <= 1e-3 points are never candidates) and its tie order
 *   (block size opt_n_threads(n), strided scan, tree reduction).
 *   The reference's `temp` (b,n) buffer (sampling.cpp:75-77) lives inside `workspace` here. */
size_t vdetr_fps_workspace_bytes(int b, int n);
int vdetr_furthest_point_sampling_f32(const float* xyz, int b, int n, int m, int32_t* idx,
                                      void* workspace, size_t workspace_bytes, vdetr_stream_t stream);

/* Variable-length batch: scene i is its own cloud xyz[i] (counts[i],3); `xyz` and `counts` are HOST arrays of b entries
 * (b <= 32), read at call time.  idx (b,m) holds indices relative to each scene.  One launch, one workgroup per scene;
 * results per scene are those of vdetr_furthest_point_sampling_f32 on that scene alone.  Replaces the per-scene Python
 * loop of the reference (models/model_vdetr.py:285-316: one FPS + gather launch pair per scene). */
size_t vdetr_fps_varlen_workspace_bytes(const int32_t* counts, int b);
int vdetr_furthest_point_sampling_varlen_f32(const float* const* xyz, const int32_t* counts, int b, int m, int32_t* idx,
                                             void* workspace, size_t workspace_bytes, vdetr_stream_t stream);

/* gather_rows: out[i,j,:] = rows[i][idx[i,j], :] for point-major tables rows[i] (n_i, c) (HOST array of b <= 32 device
 * pointers).  The backbone's out.C / out.F are point-major (model_vdetr.py:279-280); FPSModule.forward
 * (model_vdetr.py:22-34) transposes the features to (b,c,n) to gather columns of them and transposes the result back.
 * Row gathering needs neither copy and takes scenes of different sizes in one launch.  The gradient accumulates into
 * zero-filled tables with atomics (as gather_points_grad does). */
int vdetr_gather_rows_f32(const float* const* rows, const int32_t* idx, float* out, int b, int c, int m, vdetr_stream_t stream);
int vdetr_gather_rows_grad_f32(const float* grad_out, const int32_t* idx, float* const* grad_rows, int b, int c, int m,
                               vdetr_stream_t stream);

/* gather_points(points, idx) — sampling.cpp:17-42, sampling_gpu.cu:11-33.
 *   points (b,c,n) f32, idx (b,m) i32 -> out (b,c,m) f32 */
int vdetr_gather_points_f32(const float* points, const int32_t* idx, float* out, int b, int c, int n,
                            int m, vdetr_stream_t stream);
/* gather_points_grad(grad_out, idx, n) — sampling.cpp:44-66, sampling_gpu.cu:37-60.
 *   grad_out (b,c,m), idx (b,m) -> grad_points (b,c,n) (+=, caller zero-fills) */
int vdetr_gather_points_grad_f32(const float* grad_out, const int32_t* idx, float* grad_points, int b,
                                 int c, int n, int m, vdetr_stream_t stream);
/* The same as the reference's binding returns it — grad_points WRITTEN, every element, no zero-fill by the caller (sampling.cpp:53-55
 * creates the zeros inside the op).  For b*c >= 64 channel rows: one workgroup per row accumulates it in LDS (no global atomics, no
 * memset); below that a memset node + the launch above. */
int vdetr_gather_points_grad_set_f32(const float* grad_out, const int32_t* idx, float* grad_points, int b,
                                     int c, int n, int m, vdetr_stream_t stream);

/* ball_query(new_xyz, xyz, radius, nsample) — ball_query.cpp:11-35, ball_query_gpu.cu:12-57.
 *   new_xyz (b,m,3), xyz (b,n,3) -> idx (b,m,nsample) i32 (caller zero-fills; rows with no
 *   neighbour stay 0; first hit pre-fills the row; ascending point index; strict d² < r²). */
int vdetr_ball_query_f32(const float* new_xyz, const float* xyz, int32_t* idx, int b, int n, int m,
                         float radius, int nsample, vdetr_stream_t stream);

/* group_points(points, idx) — group_points.cpp:14-38, group_points_gpu.cu:11-42.
 *   points (b,c,n), idx (b,npoints,nsample) -> out (b,c,npoints,nsample) */
int vdetr_group_points_f32(const float* points, const int32_t* idx, float* out, int b, int c, int n,
                           int npoints, int nsample, vdetr_stream_t stream);
/* group_points_grad(grad_out, idx, n) — group_points.cpp:40-63, group_points_gpu.cu:46-78. */
int vdetr_group_points_grad_f32(const float* grad_out, const int32_t* idx, float* grad_points, int b,
                                int c, int n, int npoints, int nsample, vdetr_stream_t stream);
/* grad_points WRITTEN (group_points.cpp:50-52 creates the zeros inside the op); as vdetr_gather_points_grad_set_f32, with the padding
 * of ball_query's rows (repeats of the first hit) summed across the wave before it reaches LDS. */
int vdetr_group_points_grad_set_f32(const float* grad_out, const int32_t* idx, float* grad_points, int b,
                                    int c, int n, int npoints, int nsample, vdetr_stream_t stream);

/* three_nn(unknowns, knows) — interpolate.cpp:17-44, interpolate_gpu.cu:12-73.
 *   unknown (b,n,3), known (b,m,3) -> dist2 (b,n,3) f32 (SQUARED), idx (b,n,3) i32.
 *   m<3: trailing idx stay 0 and dist2 = +inf (double 1e40 stored to float). */
int vdetr_three_nn_f32(const float* unknown, const float* known, float* dist2, int32_t* idx, int b,
                       int n, int m, vdetr_stream_t stream);
/* three_interpolate(points, idx, weight) — interpolate.cpp:46-74, interpolate_gpu.cu:75-114.
 *   points (b,c,m), idx (b,n,3), weight (b,n,3) -> out (b,c,n) */
int vdetr_three_interpolate_f32(const float* points, const int32_t* idx, const float* weight,
                                float* out, int b, int c, int m, int n, vdetr_stream_t stream);
/* three_interpolate_grad(grad_out, idx, weight, m) — interpolate.cpp:75-101, interpolate_gpu.cu:119-157. */
int vdetr_three_interpolate_grad_f32(const float* grad_out, const int32_t* idx, const float* weight,
                                     float* grad_points, int b, int c, int n, int m,
                                     vdetr_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * (i)+(ii) attention.  Row-major, batch-first device tensors; head_dim is fixed at 64.
 *
 * Kinds:
 *   VDETR_ATTN_SHARED_KV  K,V are [B,nK,64] shared by all heads (GlobalShareCrossAttention
 *                         vdetr_transformer.py:733-735, ShareSelfAttention :638-639)
 *   VDETR_ATTN_PER_HEAD   K,V are [B,nK,H*64] (nn.MultiheadAttention, :468)
 * ---------------------------------------------------------------------------------------------- */
#define VDETR_ATTN_SHARED_KV 0
#define VDETR_ATTN_PER_HEAD 1

#define VDETR_MASK_NONE 0
#define VDETR_MASK_BOOL 1  /* uint8 [B,nQ,nK]; non-zero -> score := -100 (vdetr_transformer.py:746-747) */
#define VDETR_MASK_FLOAT 2 /* f32   [B,nQ,nK]; added to the score        (vdetr_transformer.py:748-749) */

typedef struct vdetr_attn_desc {
  int32_t kind;        /* VDETR_ATTN_* */
  int32_t B, H, nQ, nK;
  float scale;         /* applied to q (head_dim^-0.5, vdetr_transformer.py:670,738) */
  /* --- 3DV-RPE (NULL table = no positional bias).  vdetr_transformer.py:710-731,741 --- */
  const float* table;    /* [8, T, T, T, H] cpb_mlps[i](relative_coords_table), T = table_size */
  int32_t table_size;    /* 10 for rpe_quant "bilinear_4_10" */
  float log_scale;       /* 512 */
  float inv_log_norm;    /* 1 / (log2(8) * max_value) = 1/12 */
  const float* vertices; /* [B, nQ, 8, 3] reference_point */
  const float* xyz;      /* [B, nK, 3] */
  const float* cos_sin;  /* [B, nQ, 2] (cos, sin of reference_angle) or NULL (angle_type != object_coords) */
  /* --- additive / boolean mask --- */
  const void* mask;
  int32_t mask_kind;     /* VDETR_MASK_* */
  /* --- dropout on the attention probabilities (attn_drop, :752 / MHA dropout) --- */
  float dropout_p;       /* 0 = off */
  uint64_t seed, offset; /* key / counter offset of the counter-based dropout generator */
  const uint64_t* rng_state; /* optional DEVICE pointer to {seed, offset}, folded into the two fields above
                                (seed ^= state[0], offset += state[1]): a captured hipGraph can advance the
                                device offset between replays, and modules sharing one state stay independent
                                through their by-value seed */
  /* --- strided K / V (forward only): floats between consecutive key rows, 0 = dense (64 shared-KV, H*64 per head).
         Element (b, key, d) lives at k[(b*nK + key) * k_row_stride + d]; multiples of 4, base 16-B aligned.  Lets the
         K/V of all decoder layers come out of ONE projection GEMM over the layer-invariant encoder features
         (vdetr_transformer.py:733-735 runs self.k / self.v per layer on the same `key`). --- */
  int32_t k_row_stride, v_row_stride;
  /* --- backward only: EIGHT zero-initialised device words {max |dO row|^2, max |V row|^2, query counters of the two
         vertex halves, number of queries whose RPE vertices are not an axis-aligned box, "word 4 was filled" flag, 2 spare}.
         vdetr_attn_delta_f32 fills words 0, 1, 4 and 5 (4 and 5 only when the descriptor carries the RPE operands), vdetr_attn_bwd_scores_f32 then distributes the queries dynamically
         over its workgroups (a CU busy with other work costs 1/8 of a round, not a whole one), takes the fixed-point
         scale from the Cauchy-Schwarz bound and, when word 4 is 0 and word 5 is set (and there is no rotation operand), runs the
         axis-aligned-box kernel (attn_bwd_box4.hip) instead of the general one.  NULL: static distribution. --- */
  uint32_t* bwd_aux;
  /* --- launch shape (ABI 3).  0 = the library's default in every field.  Per call and re-entrant: these fields replace the
         process-wide setters of ABI 2 (vdetr_attn_bwd_table_set_grid, vdetr_attn_bwd_kv_set_waves). --- */
  int32_t table_grid; /* workgroups of the table-gradient launches (vdetr_attn_bwd_table_f32 / _scores_f32 with a table): 0 = one
                         per CU; an even count in 2..CUs otherwise.  The persistent workgroups take every register and ~150 KB
                         of LDS of their CU, so a caller that runs the table gradient on a side stream (it feeds parameters only)
                         lowers the count to leave WHOLE CUs to the kernels of its main chain.  The fixed-point scale of the
                         histogram follows the queries per workgroup: results are reproducible per count; counts whose
                         resolution would fall below 1e-3 of the largest entry are refused (VDETR_ERR_ARG). */
  int32_t kv_waves;   /* workgroup shape of vdetr_attn_bwd_kv_f32: 0 or 8 = 8 waves (the kernel alone on the chip), 4 = one wave
                         per SIMD with <= 256 registers (fits next to a table-gradient kernel on another stream); either shape
                         computes the same values */
  int32_t fwd_kernel; /* forward of the RPE kind: 0 = persistent workgroups, QK^T / PV on the bf16 matrix unit from split f32
                         operands (attn_fwd_pipe.hip: scores at f32 accuracy, output 8e-6 relative); 2 = the same with f32
                         matrix instructions; 1 = one workgroup per (query quad, key chunk) (attn_fwd.hip; the round-4 kernel,
                         kept for A/B runs and parity tests); 3 = the persistent kernel with q / k / v each ROUNDED to one bf16
                         part (round to nearest even) on the way into the matrix unit: the bf16 products of BASELINE config 4 on
                         f32 tensors — scores, softmax, RPE bias, accumulators and every stored tensor stay f32, no cast launches.
                         Per-head kind: 1 = the general body in eight-wave workgroups, 4 = the general body in the shape the
                         library would pick, also where the lean self-attention kernel (attn_fwd_self.hip: no mask, nQ % 16 == 0,
                         nK % 128 == 0) would be taken — same values either way up to the order of the key sums */
  int32_t bwd_kernel; /* table gradient: 0 = the box kernel where every query's vertices are a box (attn_bwd_box4.hip), the general
                         kernel otherwise — both are launched, the device decides; 1 = the general kernel only (parity tests compare
                         the two); 2 = the box kernel only: the caller vouches for boxes (vertices out of a box decode) and saves the
                         general kernel's launch; a query that is not a box then poisons dtable with NaN */
  int32_t kv_halves;  /* vdetr_attn_bwd_kv_f32: workgroups per key tile.  0 or 2 = two (each walks every other group of row tiles:
                         twice the workgroups, the shape for a launch alone on the chip), 1 = one (half the workgroups, each twice
                         as long: next to a table-gradient kernel that holds most CUs, 256 one-per-CU workgroups would run four
                         rounds on the CUs left).  Same values either way up to the order of the row-tile sums. */
  const void* kv_img; /* vdetr_attn_fwd_f32 with fwd_kernel 0 / 3: the K / V operand images of this call, packed ahead by
                         vdetr_attn_pack_kv_f32 / vdetr_attn_pack_kv_parts_f32 with the matching part count (one launch for the K / V of all decoder layers); NULL: the call packs its own into
                         `workspace` (one more launch) */
  uint32_t* fwd_sched; /* persistent forward: ONE zero device word (the item counter), left zero by the call; a word must not be
                          shared by launches that may run concurrently.  NULL: the library clears a word at the head of
                          `workspace` with a memset node in front of the launch. */
} vdetr_attn_desc;

/* Scratch needed by fwd (key-split partials, the item counter, the operand images where the caller brings none). */
size_t vdetr_attn_fwd_workspace_bytes(const vdetr_attn_desc* d);
/* K / V [B, nK, 64] f32 (row strides in floats) of `nlayers` attention calls, `layer_stride` floats apart (the layers' column blocks
 * of one joint projection) -> their operand images for vdetr_attn_desc.kv_img, vdetr_attn_kv_image_bytes(B, nK) bytes each, one
 * after the other in `img`.  Reference: the k / v projections of models/vdetr_transformer.py:735-739. */
size_t vdetr_attn_kv_image_bytes(int B, int nK);
int vdetr_attn_pack_kv_f32(const float* k, const float* v, int B, int nK, int k_row_stride, int v_row_stride, int nlayers,
                           int64_t layer_stride, void* img, vdetr_stream_t stream);
/* The same with the number of bf16 parts per operand chosen by the caller: 3 = vdetr_attn_pack_kv_f32 (fwd_kernel 0), 1 = operands
 * rounded to bf16 (fwd_kernel 3; images of vdetr_attn_kv_image_parts_bytes(B, nK, 1) bytes each). */
size_t vdetr_attn_kv_image_parts_bytes(int B, int nK, int parts);
int vdetr_attn_pack_kv_parts_f32(const float* k, const float* v, int B, int nK, int k_row_stride, int v_row_stride, int nlayers,
                                 int64_t layer_stride, int parts, void* img, vdetr_stream_t stream);

/* Fused forward: out = dropout(softmax(scale*q k^T + rpe + mask)) v.
 *   q      [B,nQ,H*64]
 *   k, v   [B,nK,64] (shared) or [B,nK,H*64] (per head)
 *   out    [B,nQ,H*64]                 (heads concatenated, vdetr_transformer.py:755)
 *   lse    row log-sum-exp of the biased scores (saved for backward)
 *   scores biased, pre-softmax scores (saved for backward), or NULL
 * Row order of lse / scores / delta / dprob: shared-KV kind [B,nQ,H](,nK); per-head kind [B,H,nQ](,nK)
 * (so that the library GEMMs of the backward are plain batched GEMMs in both cases). */
int vdetr_attn_fwd_f32(const vdetr_attn_desc* d, const float* q, const float* k, const float* v,
                       float* out, float* lse, float* scores, void* workspace, size_t workspace_bytes,
                       vdetr_stream_t stream);

/* The same forward with the merge of its key-split partial results LEFT TO THE CONSUMER (shared-KV kind): where the library splits the
 * keys of a query over several workgroups (the persistent forward does, for load balance), vdetr_attn_fwd_f32 ends with a launch that
 * merges the partial outputs — 10 us on the decoder layer's serial chain for 4 MB that the next launch reads anyway.  This entry point
 * skips it and reports where the partials are; vdetr_rb_ffn_parts_f32 (the only consumer) merges them on its way in and writes
 * `out` / `lse` as the merge launch would have (same arithmetic, same order: bit-identical).  `workspace` must stay untouched until
 * that consumer has run.  parts->ksplit == 1: nothing was split, out / lse are final. */
typedef struct vdetr_attn_parts {
  const float* part_o;   /* [ksplit][rows][64] normalised partial outputs */
  const float* part_lse; /* [ksplit][rows]     their log-sum-exp */
  int32_t ksplit;
  int32_t reserved;
  int64_t rows;          /* B * nQ * H, in (b, q, h) order */
} vdetr_attn_parts;
int vdetr_attn_fwd_parts_f32(const vdetr_attn_desc* d, const float* q, const float* k, const float* v,
                             float* out, float* lse, float* scores, void* workspace, size_t workspace_bytes,
                             vdetr_attn_parts* parts, vdetr_stream_t stream);

/* Backward, score stage (row order as above):
 *   scores  in : saved scores                         probs_out : P_drop = dropout(softmax)  (feeds dV = P_drop^T dO)
 *   dprob   in : dO V^T                               ds_out    : scale * dS                 (feeds dQ = ds_out K, dK = ds_out^T Q;
 *                                                                 the q-scale is folded in here, the table gradient uses dS)
 *   lse, delta = rowsum(dO * O)
 *   dtable [8,T,T,T,H] or NULL: += gradient of the RPE table (caller zero-fills)
 * probs_out / ds_out may alias scores / dprob (element-wise in place) EXCEPT when dtable is requested: the table
 * gradient kernel lets several waves re-read the inputs, so its outputs must be separate tensors.
 * With dprob == delta == ds_out == NULL only P_drop is produced (the `attn` return value, vdetr_transformer.py:758). */
/* The forward with q, k, v stored as bf16 (BASELINE config 4: "bf16" activations; shared-KV kinds only).  QK^T and PV run
 * on the bf16 matrix instructions; the scores, the RPE bias, the softmax, the accumulators and every output (out, lse,
 * scores) are fp32, so the backward entry points above apply unchanged (their GEMM operands are the caller's).  Row
 * strides of the descriptor count ELEMENTS and must be multiples of 8; q, k, v 16-byte aligned. */
int vdetr_attn_fwd_bf16(const vdetr_attn_desc* d, const void* q, const void* k, const void* v, float* out, float* lse,
                        float* scores, void* workspace, size_t workspace_bytes, vdetr_stream_t stream);
size_t vdetr_attn_bwd_workspace_bytes(const vdetr_attn_desc* d);
int vdetr_attn_bwd_scores_f32(const vdetr_attn_desc* d, const float* scores, const float* dprob, const float* lse,
                              const float* delta, float* probs_out, float* ds_out, float* dtable, void* workspace,
                              size_t workspace_bytes, vdetr_stream_t stream);

/* The key side of the shared-KV backward in one pass over the stored scores (autograd of vdetr_transformer.py:739-753;
 * replaces torch.bmm(dO, V^T), the element-wise stage above and the two GEMMs dV = P_drop^T dO, dK = scale dS^T q):
 *   q, dout [B,nQ,4*64];  v [B,nK,64] (row stride d->v_row_stride);  scores, lse, delta as saved / produced above
 *   ds_out [B,nQ,4,nK]  UNSCALED dS (input of vdetr_attn_bwd_table_f32 and of dQ = scale * ds_out K)
 *   dk, dv [B,nK,64]    written (not accumulated); 16-B aligned
 * Shared-KV kind with 4 heads — and the PER-HEAD kind (nn.MultiheadAttention's core): then q, dout [B,nQ,H*64], v, dk, dv
 * [B,nK,H*64] (row stride d->v_row_stride), scores / ds_out [B,H,nQ,nK], lse / delta [B,H,nQ], no bwd_aux.  Limits (checked,
 * VDETR_ERR_ARG): a score matrix (4 nQ x nK resp. nQ x nK floats) below 2 GB, at most 65535 of them (B resp. B*H), v rows, lse,
 * delta, dk, dv 16-B aligned.  fp32 operands pass through the bf16 matrix unit as hi + lo halves (three cross terms: relative
 * error of a product <= 2^-16); results are bit-reproducible from run to run. */
size_t vdetr_attn_bwd_kv_workspace_bytes(const vdetr_attn_desc* d);
int vdetr_attn_bwd_kv_f32(const vdetr_attn_desc* d, const float* q, const float* v, const float* dout, const float* scores,
                          const float* lse, const float* delta, float* ds_out, float* dk, float* dv, void* workspace,
                          size_t workspace_bytes, vdetr_stream_t stream);
/* vdetr_attn_delta_f32 + vdetr_attn_bwd_kv_f32 with the delta / bwd_aux work done by the first workgroups of the pass's
 * operand-packing launch (one launch less per attention backward): `out` is the forward's output, `delta` is WRITTEN. */
int vdetr_attn_bwd_kv_delta_f32(const vdetr_attn_desc* d, const float* q, const float* v, const float* dout, const float* out,
                                const float* scores, const float* lse, float* delta, float* ds_out, float* dk, float* dv,
                                void* workspace, size_t workspace_bytes, vdetr_stream_t stream);
/* ---- The key-side pass WITHOUT its operand-packing launch (round 6).  That launch — 8-12 us in front of every pass, 16 per step on the
 * backward's critical chain — does two independent things.  What depends on dO (the images of dO, delta, max |dO row|^2) is left
 * behind by the row-block kernel that PRODUCES dO (vdetr_rb_ffn_bwd_emit_f32 / vdetr_rb_proj_q_bwd_emit_f32 with a vdetr_rb_attn_emit);
 * what depends on the forward only (the images of q, the zeroed dk / dv, bwd_aux words 1, 4, 5) is done for up to 16 attention calls
 * by ONE launch at the head of the backward (vdetr_attn_bwd_kv_prep_f32).  vdetr_attn_bwd_kv_packed_f32 is then the pass alone:
 * `workspace` (vdetr_attn_bwd_kv_workspace_bytes, 256-B aligned) holds the images, `delta` is read, dk / dv are accumulated into.
 * One scene (B = 1), nQ a multiple of 32 for the emitting kernels; the prep launch has no such limits. */
typedef struct vdetr_attn_kv_prep {
  int32_t kind, B, H, nQ, nK, v_row_stride;   /* as in the call's vdetr_attn_desc (H = 4) */
  const float* q;          /* [B,nQ,H*64] */
  const float* v;          /* as the pass reads it */
  const float* vertices;   /* [B,nQ,8,3] or NULL (no RPE table: no box test) */
  const float* cos_sin;    /* [B,nQ,2] or NULL */
  void* workspace;         /* the pass's workspace: the q images are written */
  float *dk, *dv;          /* zeroed */
  uint32_t* bwd_aux;       /* the call's 8 zero words or NULL: words 1, 4, 5 are filled */
} vdetr_attn_kv_prep;
int vdetr_attn_bwd_kv_prep_f32(const vdetr_attn_kv_prep* items, int n, vdetr_stream_t stream);
int vdetr_attn_bwd_kv_packed_f32(const vdetr_attn_desc* d, const float* q, const float* v, const float* dout, const float* scores,
                                 const float* lse, const float* delta, float* ds_out, float* dk, float* dv, void* workspace,
                                 size_t workspace_bytes, vdetr_stream_t stream);
/* what a row-block backward kernel leaves for the key-side pass of the attention whose output gradient it has just produced */
typedef struct vdetr_rb_attn_emit {
  void* workspace;     /* that pass's workspace (256-B aligned): the dO images are written */
  float* delta;        /* [nQ,4] (shared K/V) or [4,nQ] (per head): written */
  const float* out;    /* [nQ,256] the attention's forward output */
  uint32_t* bwd_aux;   /* or NULL: max |dO row|^2 -> word 0 */
  int32_t per_head, nQ;
} vdetr_rb_attn_emit;
/* dq [B,nQ,H*64] = scale * dS K from the UNSCALED dS that vdetr_attn_bwd_kv_f32 wrote ([B,nQ,H,nK] for shared K/V, [B,H,nQ,nK]
 * per head), k as in the forward (k_row_stride honoured): a row-owner kernel with exact fp32 products (attn_bwd_dq.hip) in place
 * of the batched library GEMM the host composition used (`torch.baddbmm(..., alpha=scale)`: N = 64 is a poor shape for it).
 * Replaces the matmul autograd of vdetr_transformer.py:733-757 / nn.MultiheadAttention (:468) for dQ.  16-byte aligned operands.
 * Measured at the model's size (alone): per head 4 x 1024 x 1024 11.7 us (library 19.2), shared K/V 4096 x 4096 31 us (library
 * 26); the host module keeps the library by default (attention.py: VDETR_BWD_DQ). */
int vdetr_attn_bwd_dq_f32(const vdetr_attn_desc* d, const float* ds, const float* k, float* dq, vdetr_stream_t stream);
/* The RPE table gradient alone, from the dS that vdetr_attn_bwd_kv_f32 wrote (same kernels, workspace and bwd_aux contract
 * as vdetr_attn_bwd_scores_f32 with a dtable; d->bwd_aux is required).  dtable [8,T,T,T,4]: caller zero-fills. */
int vdetr_attn_bwd_table_f32(const vdetr_attn_desc* d, const float* ds, float* dtable, void* workspace,
                             size_t workspace_bytes, vdetr_stream_t stream);
/* Names of the kernels vdetr_attn_bwd_table_f32 launches for this descriptor (static strings; for profiles and bench labels).
 * The two table kernels share one grid and the DEVICE decides which of them works: `box_kernel` (NULL when none is launched:
 * a rotation operand, another table edge, no bwd_aux) runs iff every query's vertices are an axis-aligned box, i.e. iff
 * bwd_aux[4] == 0 && bwd_aux[5] != 0 after the launch; otherwise `general_kernel` does the work. */
int vdetr_attn_bwd_table_kernel_names(const vdetr_attn_desc* d, const char** box_kernel, const char** general_kernel);

/* delta = rowsum(dO * O) over the 64 channels of a head, in the row order of the kind (see vdetr_attn_fwd_f32);
 * dout / out [B,nQ,H*64].  The softmax-backward term the score stage subtracts.  With d->bwd_aux (shared-KV kind, `v` as
 * passed to the forward) the launch also produces the two norm maxima described at vdetr_attn_desc.bwd_aux. */
int vdetr_attn_delta_f32(const vdetr_attn_desc* d, const float* dout, const float* out, const float* v, float* delta,
                         vdetr_stream_t stream);

/* Test hook: writes the dropout keep-mask (1/0 as uint8) [B,nQ,H,nK] the kernels above use. */
int vdetr_attn_dropout_mask_u8(const vdetr_attn_desc* d, uint8_t* keep, vdetr_stream_t stream);

/* Stand-alone RPE bias (no attention): rpe [B,nQ,H,nK].  Debug / parity hook for
 * vdetr_transformer.py:710-731. */
int vdetr_rpe_bias_f32(const vdetr_attn_desc* d, float* rpe, vdetr_stream_t stream);

/* ----------------------------------------------------------------------------------------------
 * Box decode of one decoder stage: head outputs -> box parameters, corners, class probabilities.
 * Replaces the ATen composition of TransformerDecoder.get_proposal_box_predictions_refine
 * (models/vdetr_transformer.py:244-333: lines :286-318 after the five mlp heads), BoxProcessor (:20-90),
 * datasets/scannet.py:168-171 and utils/box_util.py:294-352 (flip_axis_to_camera_tensor, roty_batch_tensor,
 * get_3d_box_batch_tensor).  All tensors fp32 contiguous unless noted; nullable pointers say so.
 * ---------------------------------------------------------------------------------------------- */
#define VDETR_CLS_SOFTMAX 0 /* celoss: sem_cls_prob = softmax[..., :-1], objectness = 1 - softmax[..., -1]  (:80-86) */
#define VDETR_CLS_SIGMOID 1 /* focalloss: objectness = max sigmoid(logits); sem_cls_prob is the logits tensor (:74-79) */

typedef struct vdetr_box_decode_desc {
  int32_t B, N;          /* scenes, queries */
  int32_t A;             /* channels of the angle heads (= number of angle bins) */
  int32_t C1;            /* channels of the class head */
  int32_t num_angle_bin; /* dataset_config.num_angle_bin (angle_per_cls = 2 pi / num_angle_bin, :63) */
  int32_t cls_kind;      /* VDETR_CLS_* */
  /* inputs: head outputs in the Conv1d layout [B, channels, N] (:286-301 transposes views of these) */
  const float *center, *size;         /* [B,3,N] */
  const float *angle_cls, *angle_res; /* [B,A,N] */
  const float* cls;                   /* [B,C1,N] */
  const float *pre_center_norm, *pre_size_norm; /* [B,N,3] prior the stage refines (:278-284), no gradient */
  const float *dims_min, *dims_max;   /* [B,3] point_cloud_dims */
  /* outputs [B,N,3] */
  float *center_reg, *size_reg, *center_unnorm, *center_norm, *size_unnorm, *size_norm, *pre_center_unnorm, *pre_size_unnorm;
  float* angle_residual;            /* [B,N,A] = angle_res^T * pi / A (:303) */
  float *angle_cont, *angle_prob;   /* [B,N] */
  int32_t* angle_class;             /* [B,N] arg-max bin (saved for backward) */
  float* corners;                   /* [B,N,8,3] camera frame */
  float* corners_aa;                /* [B,N,8,3] zero-angle corners, or NULL (with A == 1 they equal `corners`) */
  float* cls_prob;                  /* [B,N,C1-1] (VDETR_CLS_SOFTMAX) or NULL */
  float* objectness;                /* [B,N] */
  /* --- the five head outputs as slabs of ONE tensor [B, 5, slab_rows, N] (they come out of one batched GEMM):
         floats between scenes for every input pointer, 0 = each input dense [B,ch,N] --- */
  int32_t in_batch_stride;
  /* transposed copies [B,N,ch] of the logits the reference hands out as transposed views (:286,:300-301), or NULL */
  float *cls_logits_t, *angle_logits_t, *angle_res_norm_t;
  /* what the NEXT decoder layer consumes (no gradient; :408-415): the corners in the lidar frame
     (convert_corners_camera2lidar, :98-102) [B,N,8,3] and cat(center_unnorm, size_unnorm) [B,N,6]; each may be NULL */
  float *corners_lidar, *center_size;
} vdetr_box_decode_desc;

/* Gradients w.r.t. the forward's outputs (each may be NULL = unused) and w.r.t. the head outputs (required). */
typedef struct vdetr_box_decode_grads {
  const float *center_reg, *size_reg, *center_unnorm, *center_norm, *size_unnorm, *size_norm; /* [B,N,3] */
  const float* angle_residual;          /* [B,N,A] */
  const float *angle_cont, *angle_prob; /* [B,N] */
  const float *corners, *corners_aa;    /* [B,N,8,3] */
  float *d_center, *d_size;             /* [B,3,N] */
  float *d_angle_cls, *d_angle_res;     /* [B,A,N] */
  /* gradients of the transposed logit outputs [B,N,ch] (each may be NULL) and, if cls_logits_t was produced, the class
     head's gradient buffer [B,C1,N] (NULL otherwise) */
  const float *cls_logits_t, *angle_logits_t, *angle_res_norm_t;
  float* d_cls;
  /* slab mode: floats between scenes for every d_* pointer (0 = dense) and rows per slab (rows >= ch are zero-filled) */
  int32_t out_batch_stride, slab_rows;
} vdetr_box_decode_grads;

int vdetr_box_decode_fwd_f32(const vdetr_box_decode_desc* d, vdetr_stream_t stream);
/* `d` as passed to the forward (inputs + the saved outputs size_unnorm, pre_size_unnorm, angle_cont, angle_class). */
int vdetr_box_decode_bwd_f32(const vdetr_box_decode_desc* d, const vdetr_box_decode_grads* g, vdetr_stream_t stream);
/* The encoder proposals' boxes (models/model_vdetr.py:348-362, :383-390): per token the class = arg max of sigmoid(point-class logit)
 * (first maximum), size = that class's anchor [ncls, 3], centre = the token's xyz; normalised centre / size over the scene extent;
 * the 8 corners at yaw 0 in the camera frame (box_util.py:319-352).  logits [B, N, ncls], xyz [B, N, 3], dims [B, 3];
 * outputs [B, N, 3] x 3 and corners [B, N, 8, 3].  One launch for ~10 tensor expressions; nothing here is differentiated. */
int vdetr_anchor_boxes_f32(const float* logits, const float* xyz, const float* dims_min, const float* dims_max, const float* anchors,
                           int B, int N, int ncls, float* size_unnorm, float* center_norm, float* size_norm, float* corners,
                           vdetr_stream_t stream);
/* The first decoder layer's box inputs (models/vdetr_transformer.py:364-398): rows topk[b, q] of a stage's corners (camera frame ->
 * lidar (x, z, -y) when corners_are_camera, :98-102), centre, size, angle, normalised centre / size, and [centre | size] for the
 * position MLP.  topk [B, nq] int64 indices into N; one launch for ~12 gathers / stacks / cats. */
int vdetr_gather_proposals_f32(const long long* topk, int B, int N, int nq, int corners_are_camera, const float* corners,
                               const float* center, const float* size, const float* angle, const float* center_norm,
                               const float* size_norm, float* o_corners_lidar, float* o_center, float* o_size, float* o_angle,
                               float* o_center_norm, float* o_size_norm, float* o_query_ref, vdetr_stream_t stream);
/* n independent backward problems (HOST arrays of descriptors / gradient blocks) in one launch per 8: the stages of a decoder are
 * differentiated together (models/vdetr_transformer.py:417-436 run their heads stage by stage; the backward of all of them is due at once). */
int vdetr_box_decode_bwd_batch_f32(const vdetr_box_decode_desc* d, const vdetr_box_decode_grads* g, int n, vdetr_stream_t stream);

/* ----------------------------------------------------------------------------------------------
 * Residual add + dropout + LayerNorm:  y = x + dropout(r);  out = LN(y; gamma, beta);  out2 = LN(y; gamma2, beta2).
 * The pre-norm residual blocks of GlobalDecoderLayer.forward_pre (models/vdetr_transformer.py:531-568:
 * `tgt = tgt + self.dropoutN(tgt2); tgt2 = self.norm(tgt)`) and the decoder's `self.norm(output)` next to the next
 * layer's `norm1(output)` (:401,:433) as one launch forward, one backward.  r == NULL: plain LayerNorm of x.
 * C must be a multiple of 256 (<= 1024); all tensors fp32 contiguous.
 * ---------------------------------------------------------------------------------------------- */
typedef struct vdetr_addln_desc {
  int32_t rows, C;
  float eps;
  float dropout_p;           /* on r; 0 = off */
  uint64_t seed, offset;     /* counter-based generator, as in vdetr_attn_desc */
  const uint64_t* rng_state; /* optional device {seed, offset}, folded in */
  const float* x;            /* [rows, C] */
  const float* r;            /* [rows, C] or NULL.  Backward only tests it for NULL (the data is not read). */
  const float *gamma, *beta; /* [C] */
  const float *gamma2, *beta2; /* [C] second affine map of the same statistics, or NULL */
  float* y;                  /* [rows, C] = x + dropout(r); required iff r != NULL (backward reads it) */
  float *out, *out2;         /* [rows, C]; out2 may be NULL */
  float *mean, *rstd;        /* [rows] statistics (written by forward, read by backward) */
} vdetr_addln_desc;

typedef struct vdetr_addln_grads {
  const float *d_out, *d_out2; /* [rows, C] gradients of out / out2 (each may be NULL) */
  const float* d_y;            /* [rows, C] gradient reaching y (or x when r == NULL) from its other uses, or NULL */
  float* d_x;                  /* [rows, C] total gradient of y = gradient of x */
  float* d_r;                  /* [rows, C] gradient of r (d_x through the dropout mask), or NULL */
  float *d_gamma, *d_beta, *d_gamma2, *d_beta2; /* [C]; the *2 pair only with d_out2 */
  float* partials;             /* vdetr_add_ln_bwd_workspace_bytes() of scratch */
} vdetr_addln_grads;

int vdetr_add_ln_fwd_f32(const vdetr_addln_desc* d, vdetr_stream_t stream);
size_t vdetr_add_ln_bwd_workspace_bytes(const vdetr_addln_desc* d);
int vdetr_add_ln_bwd_f32(const vdetr_addln_desc* d, const vdetr_addln_grads* g, vdetr_stream_t stream);
/* With d_gamma == d_beta == NULL vdetr_add_ln_bwd_f32 only leaves the per-workgroup partial sums in `partials`
 * (nparts = workspace bytes / (4 * C * 4) rows of [4][C]); this call then reduces the partials of SEVERAL backward passes
 * (HOST array of n items) in one launch per 32 items.  d_gamma2 / d_beta2 NULL: the pass had no second affine map. */
typedef struct vdetr_addln_reduce {
  const float* partials;
  int32_t nparts, C;
  float *d_gamma, *d_beta, *d_gamma2, *d_beta2;
} vdetr_addln_reduce;
int vdetr_add_ln_param_reduce_batch_f32(const vdetr_addln_reduce* items, int n, vdetr_stream_t stream);

/* ----------------------------------------------------------------------------------------------
 * The decoder layer's glue between its attention cores, three launches per layer (rowblock.hip).
 * Replaces the ~13 launches of GlobalDecoderLayer.forward_pre (models/vdetr_transformer.py:531-568) between
 * nn.MultiheadAttention's core, GlobalShareCrossAttention's core and the next layer: the q / k / v projections (:541-542,
 * in_proj of :468), out_proj + dropout1 + residual + norm2 + the cross attention's q projection (:543-545, :733), and
 * proj + proj_drop + dropout2 + residual + norm3 + linear1 + relu + dropout + linear2 + dropout3 + residual (+ the norms the
 * decoder applies to the layer output, :401 / :433) (:556-567, :755-757).  d_model = dim_feedforward = 256.
 * Rows are numbered as the decoder's sequence-first tensors lay them out (row = q * B + b); the attention cores' operands
 * (`a`, `qout`, `out`) are batch-first [B, nQ, 256].  Exact fp32 products (v_mfma_f32_16x16x4_f32).  The dropout masks are
 * those of vdetr_add_ln_fwd_f32 / vdetr_relu_dropout_fwd_f32 for the same (p, seed, rng_state), and every tensor their
 * backward entry points read is written: the backward of a fused launch is the backward of the launches it replaces.
 * All pointers 16-B aligned.
 * ---------------------------------------------------------------------------------------------- */
typedef struct vdetr_rb_linear {
  const float* w;  /* [256, 256] (out, in): nn.Linear.weight — read by the backward launches (dX = dY W) */
  const float* b;  /* [256] or NULL */
  const float* wt; /* [256, 256] (in, out): the same matrix transposed (vdetr_rb_transpose_f32) — read by the forward launches */
} vdetr_rb_linear;
typedef struct vdetr_rb_norm {
  const float *gamma, *beta; /* [256] */
  float eps;
} vdetr_rb_norm;
typedef struct vdetr_rb_drop {
  float p;       /* 0 = off */
  uint64_t seed; /* the launch's salt; offset 0; folded with rng_state as in vdetr_addln_desc */
} vdetr_rb_drop;

typedef struct vdetr_rb_qkv_desc {
  int32_t rows, B;
  const float* t;   /* [rows,256] norm1(tgt) */
  const float* pos; /* [rows,256] or NULL: q and k project t + pos, v projects t (:540-542) */
  const float* w;   /* [768,256] in_proj_weight: q | k | v blocks (backward) */
  const float* b;   /* [768] or NULL */
  const float* wt;  /* [3][256,256] the three blocks transposed (vdetr_rb_transpose_f32; forward) */
  float* x;         /* [rows,256] t + pos, written (operand of the q / k weight gradients); required with pos */
  float* out;       /* [3][B,nQ,256] batch-first q | k | v */
} vdetr_rb_qkv_desc;
int vdetr_rb_qkv_f32(const vdetr_rb_qkv_desc* d, vdetr_stream_t stream);
/* dst[i][k][n] = src[i][n][k], i < n: the forward launches read a weight through its transposed image (four lanes of a load then
 * share 64 contiguous bytes; out of the [out][in] layout each reads its own row and the load unit serialises them).  `src` is a
 * DEVICE array of n device pointers to [256,256] matrices; one launch per optimiser step covers every layer. */
int vdetr_rb_transpose_f32(const float* const* src, float* dst, int n, vdetr_stream_t stream);

typedef struct vdetr_rb_projq_desc {
  int32_t rows, B;
  const uint64_t* rng_state; /* device {seed, offset} of the step, or NULL */
  const float* a;   /* [B,nQ,256] self-attention core output */
  const float* tgt; /* [rows,256] residual stream */
  const float* pos; /* [rows,256] or NULL */
  vdetr_rb_linear proj, q;
  vdetr_rb_drop drop1;
  vdetr_rb_norm norm2;
  float *y, *mean_y, *rstd_y, *t2; /* y = tgt + drop1(proj a) [rows,256]; statistics of y [rows]; t2 = norm2(y) */
  float* xq;   /* [rows,256] t2 + pos (operand of the q weight gradient); required with pos */
  float* qout; /* [B,nQ,256] (t2 + pos) Wq^T + bq */
} vdetr_rb_projq_desc;
int vdetr_rb_proj_q_f32(const vdetr_rb_projq_desc* d, vdetr_stream_t stream);

typedef struct vdetr_rb_ffn_desc {
  int32_t rows, B;
  const uint64_t* rng_state;
  const float* a;   /* [B,nQ,256] cross-attention core output */
  const float* tgt; /* [rows,256] residual stream */
  vdetr_rb_linear proj, lin1, lin2;
  vdetr_rb_drop drop2, drop_act, drop3; /* drop2: proj_drop and dropout2 as one mask (keep probability the product) */
  vdetr_rb_norm norm3, post1, post2;    /* post2.gamma == NULL: one output norm */
  float *y, *mean_y, *rstd_y, *t2;      /* y = tgt + drop2(proj a); statistics; t2 = norm3(y) */
  float* h;                             /* drop_act(relu(lin1 t2)) */
  float *z, *mean_z, *rstd_z, *o1, *o2; /* z = y + drop3(lin2 h); statistics; o1 = post1(z), o2 = post2(z) */
} vdetr_rb_ffn_desc;
int vdetr_rb_ffn_f32(const vdetr_rb_ffn_desc* d, vdetr_stream_t stream);
/* vdetr_rb_ffn_f32 whose input rows are the merge of a forward's key-split partials (vdetr_attn_fwd_parts_f32, H = 4, ksplit <= 16):
 * d->a is not read; the merged rows are WRITTEN to attn_out [B, nQ, 256] and their log-sum-exp to attn_lse [B, nQ, 4] (what
 * vdetr_attn_fwd_f32 would have left there), everything else as vdetr_rb_ffn_f32. */
int vdetr_rb_ffn_parts_f32(const vdetr_rb_ffn_desc* d, const vdetr_attn_parts* parts, float* attn_out, float* attn_lse,
                           vdetr_stream_t stream);

/* Backward of the three launches: the input gradients as one launch each (same descriptors as the forward, whose saved
 * outputs they read); they also write the dY operand of every linear map's weight gradient (the caller computes those, e.g.
 * batched after the backward pass) and per-workgroup partial sums of the LayerNorm parameter gradients, [ceil(rows / 16)][4][256]
 * = (dgamma, dbeta, dgamma2, dbeta2) rows in the layout vdetr_add_ln_param_reduce_batch_f32 reduces.  Gradient inputs may be
 * NULL (= zero) where noted. */
typedef struct vdetr_rb_qkv_grads {
  const float *dq, *dk, *dv;           /* [B,nQ,256] gradients of the three outputs */
  float *dq_rows, *dk_rows, *dv_rows;  /* [rows,256] the same in sequence-first row order (weight-gradient operands), or NULL */
  float* d_x;                          /* [rows,256] gradient of t + pos (= gradient of pos), or NULL */
  const float* d_x_add;                /* [rows,256] or NULL: added into d_x only (a gradient that reached pos through another consumer:
                                          one launch less than summing the two afterwards) */
  float* d_t;                          /* [rows,256] gradient of t */
} vdetr_rb_qkv_grads;
int vdetr_rb_qkv_bwd_f32(const vdetr_rb_qkv_desc* d, const vdetr_rb_qkv_grads* g, vdetr_stream_t stream);

typedef struct vdetr_rb_projq_grads {
  const float* d_y;     /* [rows,256] gradient of y from its later uses, or NULL */
  const float* d_qout;  /* [B,nQ,256] gradient of qout, or NULL */
  float* d_tgt;         /* [rows,256] gradient of tgt */
  float* d_a;           /* [B,nQ,256] gradient of a, or NULL */
  float* d_t2;          /* [rows,256] gradient of t2 + pos (= gradient of pos), or NULL */
  float* dq_rows;       /* [rows,256] d_qout in sequence-first row order (operand of the q weight gradient), or NULL */
  float* d_proj;        /* [rows,256] gradient of the out-projection's output (after the dropout mask) */
  float* part_n2;       /* [ceil(rows/16)][4][256] */
} vdetr_rb_projq_grads;
int vdetr_rb_proj_q_bwd_f32(const vdetr_rb_projq_desc* d, const vdetr_rb_projq_grads* g, vdetr_stream_t stream);

typedef struct vdetr_rb_ffn_grads {
  const float *d_z, *d_o1, *d_o2;   /* [rows,256] each may be NULL */
  float* d_tgt;                     /* [rows,256] */
  float* d_a;                       /* [B,nQ,256] or NULL */
  float *d_lin2, *d_lin1, *d_proj;  /* [rows,256] dY of lin2, lin1, proj */
  float *part_post, *part_n3;       /* [ceil(rows/16)][4][256]: post1 / post2; norm3 (rows 0-1) */
} vdetr_rb_ffn_grads;
int vdetr_rb_ffn_bwd_f32(const vdetr_rb_ffn_desc* d, const vdetr_rb_ffn_grads* g, vdetr_stream_t stream);
/* The FFN layer in front of the decoder (models/vdetr_transformer.py:585-606, FFNLayer.forward_pre) as one launch forward, one backward:
 *   t2 = norm3(tgt);  h = drop_act(relu(lin1 t2));  z = t2 + drop3(lin2 h);  o1 = post1(z) [, o2 = post2(z)]
 * in the descriptor of vdetr_rb_ffn_f32 (a, proj, y, drop2 unused forward; backward: y = the layer's input rows, mean_y / rstd_y the
 * statistics the forward left; g->d_tgt = the input's gradient, g->d_a / d_proj unused). */
int vdetr_rb_ffn0_f32(const vdetr_rb_ffn_desc* d, vdetr_stream_t stream);
int vdetr_rb_ffn0_bwd_f32(const vdetr_rb_ffn_desc* d, const vdetr_rb_ffn_grads* g, vdetr_stream_t stream);
/* the two backward launches that produce an attention's output gradient, leaving its packed form behind as well (see
 * vdetr_attn_bwd_kv_packed_f32): d->B = 1, d->rows = e->nQ a multiple of 32, g->d_a required */
int vdetr_rb_ffn_bwd_emit_f32(const vdetr_rb_ffn_desc* d, const vdetr_rb_ffn_grads* g, const vdetr_rb_attn_emit* e, vdetr_stream_t stream);
int vdetr_rb_proj_q_bwd_emit_f32(const vdetr_rb_projq_desc* d, const vdetr_rb_projq_grads* g, const vdetr_rb_attn_emit* e,
                                 vdetr_stream_t stream);

/* ----------------------------------------------------------------------------------------------
 * y = dropout(relu(BatchNorm1d(x))) on [B, C, N]: the hidden blocks of GenericMLP (models/helpers.py:74-141,
 * Conv1d -> BatchNorm1d -> ReLU -> Dropout; the box heads of models/vdetr_transformer.py:193-242 and
 * PositionEmbeddingLearned, helpers.py:17-33).  One launch forward, one backward.
 * ---------------------------------------------------------------------------------------------- */
typedef struct vdetr_bnact_desc {
  int32_t B, C, N;
  int32_t training;          /* 1: batch statistics (+ running-statistics update), 0: running statistics */
  int32_t relu;              /* apply max(., 0) after the affine map */
  float eps, momentum;
  float dropout_p;           /* after the relu; training only */
  uint64_t seed, offset;     /* counter-based generator, as in vdetr_attn_desc */
  const uint64_t* rng_state;
  const float* x;            /* [B, C, N] */
  const float *gamma, *beta; /* [C] or both NULL */
  float *running_mean, *running_var; /* [C], updated in place in training mode (may be NULL there) */
  float* y;                  /* [B, C, N] */
  float *save_mean, *save_invstd;    /* [C], written in training mode, read by backward */
  const float* pre_bias;     /* [C] or NULL: bias of the convolution in front, left OUT of x (it cancels under batch
                                statistics; the running mean is kept as that of conv(x) + bias) */
  int64_t* counters[8];      /* num_batches_tracked of the modules that make up the channel group: += 1 (training) */
  int32_t ncounters;
  int32_t stats_given;       /* training only.  1: save_mean / save_invstd are INPUTS (statistics of the batch over all
                                data-parallel ranks, SyncBatchNorm: reference main.py:512-514); the kernel normalises with them
                                and leaves the running statistics alone (the caller keeps them, it knows the global count) */
} vdetr_bnact_desc;

typedef struct vdetr_bnact_grads {
  const float* dy;            /* [B, C, N] */
  float* dx;                  /* [B, C, N] or NULL */
  float *d_gamma, *d_beta;    /* [C] or NULL: THIS rank's sums (sum g * xhat, sum g) */
  /* cross-replica statistics: the two sums over ALL ranks ([C] each) and 1 / (global element count) as a device scalar;
   * all NULL: dx uses the launch's own sums and B * N */
  const float *sum_dy_xhat, *sum_dy, *inv_count;
} vdetr_bnact_grads;

/* y = dropout(relu(x)), element-wise over n floats (n % 4 == 0, 16-B aligned): the FFN's
 * `self.dropout(self.activation(self.linear1(.)))` (models/vdetr_transformer.py:566,604).  The backward needs y only. */
int vdetr_relu_dropout_fwd_f32(const float* x, float* y, long n, float dropout_p, uint64_t seed, uint64_t offset,
                               const uint64_t* rng_state, vdetr_stream_t stream);
int vdetr_relu_dropout_bwd_f32(const float* y, const float* dy, float* dx, long n, float dropout_p, vdetr_stream_t stream);
int vdetr_bn_act_fwd_f32(const vdetr_bnact_desc* d, vdetr_stream_t stream);
/* this rank's per-channel mean and M2 = sum (x - mean)^2 of x [B, C, N] (what ranks exchange for SyncBatchNorm; merged with
 * the counts by Chan's formula on the caller's side) */
int vdetr_bn_stats_f32(const vdetr_bnact_desc* d, float* mean, float* m2, vdetr_stream_t stream);
int vdetr_bn_act_bwd_f32(const vdetr_bnact_desc* d, const vdetr_bnact_grads* g, vdetr_stream_t stream);
/* n independent problems (HOST arrays) of the same B*N in one launch per 12: the hidden blocks of several decoder stages'
 * box heads, whose backward v-detr_amd/vdetr_transformer.py:_DeferredHeads runs as one batched pass */
int vdetr_bn_act_bwd_batch_f32(const vdetr_bnact_desc* descs, const vdetr_bnact_grads* grads, int n, vdetr_stream_t stream);

/* ----------------------------------------------------------------------------------------------
 * Gradient packing: n separate fp32 tensors -> slices of one flat buffer, one launch.  The role of the bucket copy in
 * DistributedDataParallel's reducer (reference main.py:515-517).  All three tables are DEVICE arrays:
 *   entries[e]      = {src (NULL = zero-fill), dst_offset (floats), numel}
 *   block_entry[b], block_chunk[b]: workgroup b copies floats [chunk * C, (chunk+1) * C) of entry block_entry[b],
 *   C = vdetr_pack_chunk_floats().
 * ---------------------------------------------------------------------------------------------- */
typedef struct vdetr_pack_entry {
  const void* src;
  uint64_t dst_offset;
  uint64_t numel;
} vdetr_pack_entry;
/* out[c] = sum_r x[r * row_stride + c]: the bias gradient of nn.Linear / 1x1 Conv1d (the `sum` inside AddmmBackward of
 * every projection in models/vdetr_transformer.py), one launch. */
int vdetr_colsum_f32(const float* x, float* out, int rows, int cols, long row_stride, vdetr_stream_t stream);
/* out[i][c] = sum_r x[i * item_stride + r * row_stride + c] for n matrices: one launch, or two for tall matrices with few columns
 * (partial sums of row ranges into `workspace`, vdetr_colsum_workspace_bytes(n, rows, cols) bytes, then their sum; without a workspace:
 * one launch, unsplit).  Fixed summation order: bit-reproducible. */
size_t vdetr_colsum_workspace_bytes(int n, int rows, int cols);
int vdetr_colsum_batched_f32(const float* x, float* out, int n, int rows, int cols, long row_stride, long item_stride,
                             void* workspace, size_t workspace_bytes, vdetr_stream_t stream);
/* the same for n separately allocated matrices: `items` is a DEVICE array of n pointers (16-byte aligned; cols, row_stride % 4 == 0) */
int vdetr_colsum_ptrs_f32(const float* const* items, float* out, int n, int rows, int cols, long row_stride, void* workspace,
                          size_t workspace_bytes, vdetr_stream_t stream);
int vdetr_pack_chunk_floats(void);
int vdetr_pack_f32(const vdetr_pack_entry* entries, const uint32_t* block_entry, const uint32_t* block_chunk, int nblocks,
                   float* dst, vdetr_stream_t stream);
/* The same launch, which also leaves sumsq[b] = the sum of squares of what workgroup b copied (nblocks floats): the gradient norm of
 * clip_grad_norm_ (engine.py:105-106) costs no pass of its own (vdetr_adamw_clip_f32 adds the partials up). */
int vdetr_pack_sumsq_f32(const vdetr_pack_entry* entries, const uint32_t* block_entry, const uint32_t* block_chunk, int nblocks,
                         float* dst, float* sumsq, vdetr_stream_t stream);

/* ----------------------------------------------------------------------------------------------
 * The RPE tables of n cpb MLPs (Linear(3, hidden) -> ReLU -> Linear(hidden, 4, bias=False); models/vdetr_transformer.py:725 evaluates
 * eight of them per cross-attention layer on the T^3 grid) in one launch (csrc/cpb_tables.hip).
 *   coords [P,3]; w1 [n,hidden,3]; b1 [n,hidden]; w2 [n,4,hidden]  ->  hid_out [n,P,hidden] = relu(coords w1^T + b1) (what the tables'
 *   backward reads), tables [n,P,4] = hid_out w2^T.  hidden a multiple of 16, <= 256.
 * ---------------------------------------------------------------------------------------------- */
int vdetr_cpb_tables_f32(const float* coords, const float* w1, const float* b1, const float* w2, int n, int P, int hidden, int H,
                         float* hid_out, float* tables, vdetr_stream_t stream);

/* ----------------------------------------------------------------------------------------------
 * Gradient-norm clipping + AdamW on a flat parameter buffer, one launch (csrc/optim.hip).
 * Reference: engine.py:105-107 (clip_grad_norm_ then optimizer.step()), optimizer.py:6-26 (torch.optim.AdamW; amsgrad off).
 *   p -= lr wd p;  m += (1 - b1)(g' - m);  v = b2 v + (1 - b2) g'^2;  p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
 *   g' = g / max((||g|| + norm_eps) / max_norm, 1), ||g||^2 = the sum of `sumsq` (vdetr_pack_sumsq_f32's or vdetr_sumsq_f32's
 *   partials; NULL: no clipping).  t = *step + 1; the launch leaves *step = t (device-resident: a captured graph replays it).
 *   `ticket`: one zero device word, left zero.  norm_out (optional): ||g||.
 * ---------------------------------------------------------------------------------------------- */
typedef struct vdetr_adamw_desc {
  float* param;         /* [n] */
  const float* grad;    /* [n] */
  float* exp_avg;       /* [n] */
  float* exp_avg_sq;    /* [n] */
  int64_t n;
  float* step;
  uint32_t* ticket;
  const float* sumsq;
  int32_t nsumsq;
  float max_norm;
  float norm_eps;       /* 1e-6 in clip_grad_norm_ */
  float* norm_out;
  double lr, beta1, beta2, eps, weight_decay;
} vdetr_adamw_desc;
int vdetr_adamw_clip_f32(const vdetr_adamw_desc* d, vdetr_stream_t stream);
/* partial[b] = sum of squares of slice b of g [n] (npartial = vdetr_sumsq_blocks(n)): the norm's first half where the flat gradient
 * changed after the pack (all-reduce, N > 1). */
int vdetr_sumsq_blocks(long n);
int vdetr_sumsq_f32(const float* g, long n, float* partial, int npartial, vdetr_stream_t stream);

/* ----------------------------------------------------------------------------------------------
 * Z-order permutation of a scene's key points, one launch (ranks by counting, n / 64 workgroups per scene).  Replaces the tensor expression the
 * decoder would otherwise spend ~50 launches on in front of the cross attention (v-detr_amd/pc_util.py:morton_argsort; the
 * reference attends in FPS order, models/vdetr_transformer.py:400-436 — attention does not depend on the key order).
 *   xyz [B,n,3] fp32 -> codes [B,n] int32 (30-bit Morton code of the point in the scene's bounding box; may be NULL)
 *                       order [B,n] int64 (indices sorted by (code, index); may be NULL; n <= vdetr_morton_sort_max())
 * ---------------------------------------------------------------------------------------------- */
int vdetr_morton_sort_max(void);
int vdetr_morton_order_f32(const float* xyz, int B, int n, int* codes, long long* order, vdetr_stream_t stream);
/* order [B, nq] int64 = indices of the nq largest of values [B, n] per row, largest first, equal values by ascending index — what
 * torch.sort(descending=True, stable=True) returns (one of the orders torch.topk may return, models/vdetr_transformer.py:364-366:
 * the decoder's proposals); n / 64 workgroups per row, n <= vdetr_morton_sort_max() = 16384. */
int vdetr_topk_order_f32(const float* values, int B, int n, int nq, long long* order, vdetr_stream_t stream);

/* ----------------------------------------------------------------------------------------------
 * Set criterion on the device (reference criterion.py).  Replaces, without a host round trip:
 *   repeat_ground_truth (criterion.py:511-600), the pairwise GIoU / centre / size matrices and the matcher's cost
 *   (criterion.py:618-631, 122-196; utils/box_util.py:441-600 with rotated_boxes=False), the nine
 *   scipy.optimize.linear_sum_assignment calls on the host (criterion.py:198-221), points_in_boxes_all
 *   (criterion.py:270-289) and the matched losses with their gradients (criterion.py:77-98, 329-509).
 * Ground truth travels as packed records of VDETR_GT_FLOATS floats per box slot.
 * ---------------------------------------------------------------------------------------------- */
#define VDETR_GT_CORNERS 0    /* 24: gt_box_corners[8][3], camera frame */
#define VDETR_GT_CENTER 24    /* 3: gt_box_centers */
#define VDETR_GT_SIZE 27      /* 3: gt_box_sizes */
#define VDETR_GT_ANGLE 30     /* gt_box_angles */
#define VDETR_GT_LABEL 31     /* gt_box_sem_cls_label (as float) */
#define VDETR_GT_ANGLE_CLS 32 /* gt_angle_class_label (as float) */
#define VDETR_GT_ANGLE_RES 33 /* gt_angle_residual_label */
#define VDETR_GT_PRESENT 34   /* gt_box_present */
#define VDETR_GT_FLOATS 36

/* gt [B,G,F] -> gt_rep [B,G*repeat,F]: the list tiled `repeat` times, present boxes first in stable order, the rest
 * zero (criterion.py:511-590); nactual[B] / nactual_rep[B] = present counts (:592, :660); sums[0] = sum nactual,
 * sums[1] = sum nactual_rep as floats (the caller averages them over ranks and clamps at 1: :593, :661);
 * sums[2] = 1 if any slot has a positive angle (torch.any(gt_box_angles > 0), criterion.py:616: selects the rotated GIoU),
 * else 0.  `sums` holds 3 floats. */
int vdetr_gt_prepare_f32(const float* gt, int B, int G, int repeat, float* gt_rep, int64_t* nactual, int64_t* nactual_rep,
                         float* sums, vdetr_stream_t stream);

typedef struct vdetr_match_desc {
  int32_t B, P, G;   /* scenes, proposals, ground-truth slots */
  int32_t C, A;      /* class channels of `cls`, angle bins */
  int32_t cls_kind;  /* VDETR_CLS_SIGMOID: focal cost of sigmoid(cls) (criterion.py:123-138); _SOFTMAX: -cls[label] (:139-146) */
  int32_t label_override; /* >= 0: every box carries this class (the binary first stage, criterion.py:679-681); -1: records */
  float w_cls, w_objectness, w_center, w_giou, w_size, w_angle_cls, w_angle_reg; /* matcher_*_cost (main.py:118-124) */
  const float* cls;         /* [B,P,C] outputs["sem_cls_prob"] */
  const float* objectness;  /* [B,P] */
  const float *center_reg, *size_reg, *pre_center, *pre_size; /* [B,P,3] center_reg, size_reg, pre_box_*_unnormalized */
  const float* corners;     /* [B,P,8,3] box_corners */
  const float *angle_logits, *angle_res_norm; /* [B,P,A] */
  const float* gt;          /* [B,G,VDETR_GT_FLOATS] */
  const int64_t* nactual;   /* [B] */
  float* cost_t;            /* [B,G,P]: final_cost[b,p,g] stored box-major (columns >= nactual[b] are not read by the solver) */
  float* giou_t;            /* optional [B,G,P] pairwise GIoU (outputs["gious"], box-major), or NULL */
  const float* rotated;     /* device scalar (sums[2] of vdetr_gt_prepare_f32) != 0: footprint overlap by polygon clipping
                               (rotated_boxes=True, criterion.py:616, box_util.py:566-589); NULL = axis-aligned */
} vdetr_match_desc;
int vdetr_match_cost_f32(const vdetr_match_desc* d, vdetr_stream_t stream);
/* the same for n stage descriptors (HOST array; shapes may differ) in one launch per 12 descriptors */
int vdetr_match_cost_batch_f32(const vdetr_match_desc* descs, int n, vdetr_stream_t stream);

/* Rectangular linear sum assignment, one workgroup per (problem, scene): scipy's shortest-augmenting-path solver
 * (rectangular_lsap.cpp of scipy 1.5.1, requirements.txt:9) restated in fp64 with its traversal order and tie rules, so
 * the result is the one linear_sum_assignment(final_cost[b, :, :nactual[b]]) returns.  Writes per_prop_gt_inds and
 * proposal_matched_mask (criterion.py:200-221), zero where unmatched.  Limits: max(P, G) <= 8192, min(P, G) <= 2048.
 * status (optional, 2 ints per (problem, scene) pair w in launch order, zero-filled by the caller): status[2w] is set to 1
 * if a cost is NaN/-inf or the matrix is infeasible (scipy raises ValueError there; that scene's outputs stay zero);
 * status[2w+1] receives the number of row scans the solve took (diagnostic). */
#define VDETR_LSA_MAX_PROBLEMS 16
typedef struct vdetr_lsa_problem {
  const float* cost_t;    /* [B,G,P] */
  const int64_t* nactual; /* [B] */
  int64_t* inds;          /* [B,P] */
  float* mask;            /* [B,P] */
  int32_t B, P, G;
  int32_t row_repeat;     /* > 1: the first nactual box rows are `row_repeat` identical tiles (repeated ground truth,
                             criterion.py:511-600): row r == row r % (nactual / row_repeat) bit for bit; lets the solver keep
                             the distinct rows in LDS.  0 or 1: no structure assumed. */
} vdetr_lsa_problem;
typedef struct vdetr_lsa_batch {
  int32_t nproblems;
  int32_t reserved;
  vdetr_lsa_problem p[VDETR_LSA_MAX_PROBLEMS];
} vdetr_lsa_batch;
int vdetr_lsa_f64(const vdetr_lsa_batch* batch, int32_t* status, vdetr_stream_t stream);

/* Seed-point labels of loss_point_cls (criterion.py:270-301): label[b,n] = class of the smallest-volume box containing
 * seed n (mmcv points_in_boxes_all semantics on (centre, size, angle) with the bottom face at z - dz/2), C = none. */
int vdetr_point_labels_f32(const float* seed_xyz, const float* gt, const int64_t* nactual, int B, int N, int G, int C,
                           int64_t* labels, float* matched_count, vdetr_stream_t stream);
/* matched_count (optional, zero-filled by the caller): += number of seeds inside a box */

typedef struct vdetr_setloss_desc {
  int32_t B, P, G, C, A;
  int32_t label_override;  /* as in vdetr_match_desc */
  float focal_alpha;       /* cls_loss "focalloss_<alpha>" (criterion.py:238-239) */
  float w_cls, w_angle_cls, w_angle_reg, w_center, w_size, w_giou; /* loss_*_weight (main.py:128-136); folded into values and gradients */
  int32_t cls_kind;        /* VDETR_CLS_SIGMOID: focal loss; VDETR_CLS_SOFTMAX: cross entropy over C classes whose LAST one is "no
                              object" (cls_loss="celoss", criterion.py:360-371), weighted mean with w_no_object on that class */
  float w_no_object;       /* loss_no_object_weight (main.py:131) */
  const float* cls_logits; /* [B,P,C] */
  /* box terms; all NULL = classification only (the seed-point loss) */
  const float *center_reg, *size_reg, *pre_center, *pre_size; /* [B,P,3] */
  const float* corners;    /* [B,P,8,3] */
  const float *angle_logits, *angle_res_norm; /* [B,P,A] */
  const float* gt;         /* [B,G,VDETR_GT_FLOATS] */
  const int64_t* nactual;  /* [B]: gates everything on sum > 0 (num_boxes_replica, criterion.py:331,...) */
  const int64_t* inds;     /* [B,P] per_prop_gt_inds, or NULL with `labels` */
  const float* mask;       /* [B,P] proposal_matched_mask */
  const int64_t* labels;   /* [B,P] class per row directly (C = none); used when inds == NULL */
  const float* num_boxes;  /* device scalar */
  /* outputs.  losses[0..7] += {sem_cls, angle_cls, angle_reg, center, size, giou, cardinality, weighted total}
     (the six losses already multiplied by their weights, as loss_dict holds them: criterion.py:648-651). */
  float* losses;
  unsigned long long* card_ws; /* [B] zero-initialised scratch of the cardinality count (one 64-bit atomic per workgroup) */
  float *d_cls_logits, *d_center_reg, *d_size_reg, *d_corners, *d_angle_logits, *d_angle_res_norm; /* d total / d input; written in full */
  const float* rotated;    /* as in vdetr_match_desc */
  const float* ce_rows_matched; /* cross entropy with `labels`: device scalar = rows whose label is not C-1 (the weighted mean's
                                   normaliser); NULL with (inds, mask): sum_b min(nactual_b, P) */
} vdetr_setloss_desc;
int vdetr_set_loss_f32(const vdetr_setloss_desc* d, vdetr_stream_t stream);
/* the same for n descriptors (HOST array: all stages of a step + the seed-point loss) in one launch per 12 descriptors */
int vdetr_set_loss_batch_f32(const vdetr_setloss_desc* descs, int n, vdetr_stream_t stream);

/* ----------------------------------------------------------------------------------------------
 * Greedy 3-D NMS of a scene's predictions (reference utils/nms.py:78-162 nms_3d_faster /
 * nms_3d_faster_samecls as called from utils/ap_calculator.py:165-220 on the min / max extents of the 8 box corners).
 * corners (B,K,8,3) f32, score (B,K) f32, cls (B,K) i32 or NULL (class-agnostic nms_3d_faster), valid (B,K) u8 or NULL
 * (the nonempty_box_mask: invalid boxes are neither kept nor suppress), order (B,K) i64 = ascending STABLE arg-sort of
 * score (the reference's np.argsort; boxes are visited from its end) -> keep (B,K) u8.  IoU and the threshold test run in
 * fp64 as in numpy (np.zeros((K, 8)) holds the float32 values as float64).  K <= 4096.
 * ---------------------------------------------------------------------------------------------- */
size_t vdetr_nms3d_workspace_bytes(int B, int K);
int vdetr_nms3d_f32(const float* corners, const float* score, const int32_t* cls, const uint8_t* valid, const int64_t* order,
                    int B, int K, double iou_threshold, int old_type, uint8_t* keep, void* workspace, size_t workspace_bytes,
                    vdetr_stream_t stream);

/* Points inside every predicted box: counts (B,K) i32 += #{n : points[b,n] in boxes[b,k]} (ZERO-FILLED by the caller).
 * Replaces mmcv points_in_boxes_all + sum as used by parse_predictions' remove_empty_box (utils/ap_calculator.py:78-93)
 * without the (B,N,K) flag tensor.  points (B,N,3) f32; boxes (B,K,7) f32 = centre xyz, sizes dx dy dz, yaw rz (the
 * caller's bottom-centre shift and mmcv's shift back are applied inside, in the reference's operation order). */
int vdetr_box_point_count_f32(const float* points, const float* boxes, int B, int N, int K, int32_t* counts,
                              vdetr_stream_t stream);

/* Best ground-truth box of every detection for the AP computation (utils/eval_det.py:160-172 with get_iou_obb =
 * utils/box_util.py:122-147 box3d_iou).  pred_corners (P,8,3) f32 with image index pred_img (P) and class pred_cls (P);
 * gt_corners (G,8,3) f32 grouped by image: the boxes of image i are rows img_gt_begin[i] .. img_gt_begin[i+1]-1, classes
 * gt_cls (G).  -> ovmax (P) f64 = largest IoU with a ground-truth box of the same image and class (-inf if none),
 * jmax (P) i32 = its row (the first of equal maxima; -1 if none).  float64 arithmetic on the float32 corners. */
int vdetr_box3d_iou_max_f64(const float* pred_corners, const int32_t* pred_img, const int32_t* pred_cls, int P,
                            const float* gt_corners, const int32_t* gt_cls, const int32_t* img_gt_begin, double* ovmax,
                            int32_t* jmax, vdetr_stream_t stream);

/* ----------------------------------------------------------------------------------------------
 * Sparse-convolution backbone: index kernels behind the MinkowskiEngine call sites of the
 * reference (models/mink_resnet.py:38-84 ME.MinkowskiConvolution / BasicBlock, models/model_vdetr.py:141-176
 * MinkowskiConvolution / MinkowskiConvolutionTranspose / MinkowskiGenerativeConvolutionTranspose, :250-280 run_encoder).
 * MinkowskiEngine is an un-vendored dependency (README.md:47-53, no pinned commit): the operator restated is the published
 * generalized sparse convolution  out[u] = sum_k in[u + offset_k] W_k  over occupied sites.
 * A voxel KEY packs (batch, x, y, z) into an int64 as (batch << 48) | (x + 32768) << 32 | (y + 32768) << 16 | (z + 32768):
 * ascending keys = lexicographic (batch, x, y, z).  Coordinates are in units of the finest voxel (tensor stride 1).
 * ---------------------------------------------------------------------------------------------- */
/* nbr[k][u] (K x nout, i32) = index into the SORTED in_keys of the site out_keys[u] + offsets[k] (offsets [K,3] i32 as
 * dx, dy, dz), or -1 if it is not occupied.  Replaces the coordinate manager's kernel-map construction. */
int vdetr_sp_kernel_map_i32(const int64_t* in_keys, int nin, const int64_t* out_keys, int nout, const int32_t* offsets, int K,
                            int32_t* nbr, vdetr_stream_t stream);
/* inv[k][i] (K x nin, i32, filled with -1 by the caller) = the output row u with nbr[k][u] == i (unique on a lattice). */
int vdetr_sp_inverse_map_i32(const int32_t* nbr, int K, int nout, int nin, int32_t* inv, vdetr_stream_t stream);
/* col[u][k][:] = in[nbr[k][u]][:] or zeros: in [nin,C] f32 -> col [nout,K,C] f32, C % 4 == 0.  The forward pass of a layer
 * is col [nout, K*C] x W [K*C, Cout] (one library GEMM), its weight gradient col^T x dout. */
int vdetr_sp_gather_cols_f32(const float* in, const int32_t* nbr, int K, int nout, int C, float* col, vdetr_stream_t stream);
/* din[i][:] = sum_k dcol[inv[k][i]][k][:]: the adjoint of vdetr_sp_gather_cols_f32 written as a gather (fixed summation
 * order, no atomics).  dcol [nout,K,C] f32, inv [K,nin] -> din [nin,C], written in full.  offset_major_rows = M > 0: the
 * source is laid out [K,M,C] instead (row inv[k][i] of slice k): the per-offset compacted row lists, where slice k holds only
 * the sites that HAVE a neighbour through offset k. */
int vdetr_sp_gather_sum_f32(const float* dcol, const int32_t* inv, int K, int nin, int C, int offset_major_rows, float* din,
                            vdetr_stream_t stream);
/* offset_major_rows < 0: `inv` holds ABSOLUTE rows of a flat source [P,C] (the pair lists below). */

/* Fused pair-list products (fp32 matrix cores, nothing padded, no gathered operand in memory).  A layer's geometry is the
 * list of its (input row, output row) pairs sorted by kernel offset; `tiles` [ntiles,3] i32 = (offset k, first pair, pair
 * count <= 128) cuts the list into 128-pair tiles that never straddle two offsets.
 *   y[p][:] = x[arow[p]][:] * W[k(p)]              transposed = 0: x [.,cin], y [P,cout]  (forward: arow = input rows)
 *   y[p][:] = x[arow[p]][:] * W[k(p)]^T            transposed = 1: x [.,cout], y [P,cin]  (input gradient: arow = output rows)
 * w [K,cin,cout] f32 as MinkowskiConvolution.kernel stores it; the contraction width must be a multiple of 16. */
int vdetr_sp_pairs_gemm_f32(const float* x, const int32_t* arow, const float* w, const int32_t* tiles, int ntiles, int cin,
                            int cout, int transposed, float* y, vdetr_stream_t stream);
/* Weight gradient over the pair list: partials[slot] [cin,cout] = sum over the chunk's pairs of x[pin[p]]^T dy[pout[p]];
 * `chunks` [nchunks,4] i32 = (offset k, first pair, pair count, partial slot).  A segment may be cut into several chunks
 * (split-K for narrow layers); the caller sums the partial slots of an offset in a fixed order. */
int vdetr_sp_pairs_wgrad_f32(const float* x, const float* dy, const int32_t* pin, const int32_t* pout, const int32_t* chunks,
                             int nchunks, int cin, int cout, float* partials, vdetr_stream_t stream);
/* dw[k][:] = sum of partials[c][:] over c in [seg[k], seg[k+1]) (seg: K+1 ints on the device; elems = Cin*Cout, a multiple of 4):
 * the chunk partials of vdetr_sp_pairs_wgrad_f32 -> the weight gradient, in chunk order (deterministic). */
int vdetr_sp_wgrad_reduce_f32(const float* partials, const int32_t* seg, int K, long elems, float* dw, vdetr_stream_t stream);
/* The pair lists of a kernel map, built on the device: pairs sorted by (offset k, output row u).
 *   nbr [K][nout] (vdetr_sp_kernel_map_i32)  ->  pin, pout [capacity K*nout; the first P entries are written]
 *                                                slot [K][nout], islot [K][nin]  (pair index or -1)
 *                                                counts [K+1] (device): pairs per offset, counts[K] = P
 * workspace: vdetr_sp_pair_plan_workspace_ints(K, nout) ints.  No host synchronisation: the caller copies `counts` back
 * when it needs the sizes (v-detr_amd/sparse_ops.py:PairPlan).  Deterministic (ranks, no atomics). */
int vdetr_sp_pair_plan_workspace_ints(int K, int nout);
int vdetr_sp_pair_plan_i32(const int32_t* nbr, int K, int nout, int nin, int32_t* pin, int32_t* pout, int32_t* slot,
                           int32_t* islot, int32_t* counts, int32_t* workspace, vdetr_stream_t stream);

/* BatchNorm (+ residual) (+ activation) over the point-major feature table [N,C] of a sparse tensor: ME.MinkowskiBatchNorm
 * followed by MinkowskiReLU / MinkowskiELU and, in the residual blocks, `out += residual` in front of the ReLU
 * (models/mink_resnet.py:38-84 via MinkowskiEngine's BasicBlock; models/model_vdetr.py:141-176).
 *   y = act((x - mean) * invstd * gamma + beta + residual),  act: 0 none, 1 ReLU, 2 ELU(alpha = 1)
 * training != 0: batch statistics (biased variance for the normalisation; running statistics updated with `momentum` and the
 * unbiased variance, num_batches_tracked += 1, as nn.BatchNorm1d); 0: running statistics.  save_mean / save_invstd [C] are
 * written for the backward.  workspace: vdetr_sp_bn_workspace_bytes(N, C), shared by forward and backward.  C % 4 == 0. */
typedef struct vdetr_spbn_desc {
  int32_t N, C, act, training;
  float eps, momentum;
  const float *x, *gamma, *beta, *residual; /* gamma / beta / residual may be NULL */
  float *running_mean, *running_var;        /* may be NULL in training mode */
  int64_t* num_batches_tracked;             /* optional */
  float *y, *save_mean, *save_invstd;
  void* workspace;
} vdetr_spbn_desc;
size_t vdetr_sp_bn_workspace_bytes(int N, int C);
int vdetr_sp_bn_act_fwd_f32(const vdetr_spbn_desc* d, vdetr_stream_t stream);
/* dx [N,C] (or NULL), dresidual [N,C] (or NULL: = dy * act'), dgamma / dbeta [C] (or NULL) from dy [N,C], the forward's x, y,
 * save_mean, save_invstd */
int vdetr_sp_bn_act_bwd_f32(const vdetr_spbn_desc* d, const float* dy, float* dx, float* dresidual, float* dgamma, float* dbeta,
                            vdetr_stream_t stream);

/* ----------------------------------------------------------------------------------------------
 * Fixed coordinate embeddings — PositionEmbeddingCoordsSine (models/position_embedding.py:21-148); one launch each.
 *   xyz (b,n,3); range_min / range_max (b,3) = the scene extent for `normalize=True` (shift_scale_points,
 *   utils/pc_util.py:38-66), both NULL for normalize=False.
 * fourier (:98-127): out (b, 2*d_out, n); channel c = sin(2 pi x . gauss_b[:,c]), channel d_out + c = its cos;
 *   gauss_b (3, ldb) row-major is the module's checkpointed buffer, d_out <= ldb columns of it are used.
 * sine (:51-96): out (b, num_channels, n), num_channels even; per axis a block of channels
 *   (even i: sin, odd i: cos)(x * scale / temperature^(2 floor(i/2) / cdim)); scale == 0 leaves x unscaled. */
int vdetr_pos_embed_fourier_f32(const float* xyz, int b, int n, const float* range_min, const float* range_max,
                                const float* gauss_b, int ldb, int d_out, float* out, vdetr_stream_t stream);
int vdetr_pos_embed_sine_f32(const float* xyz, int b, int n, const float* range_min, const float* range_max,
                             int num_channels, float temperature, float scale, float* out, vdetr_stream_t stream);

/* ----------------------------------------------------------------------------------------------
 * The five box heads of a decoder stage as three launches (heads.hip, round 6) — reference
 * models/vdetr_transformer.py:244-285 (five GenericMLPs on the same features) + models/helpers.py:74-141
 * (Conv1d -> BatchNorm1d -> ReLU -> Dropout, twice, -> Conv1d); training mode (batch statistics).
 *   launch 1   pre1 = W1 x                                   [G*256 channels] + per-32-token partial statistics
 *   launch 2   h1 = drop(relu(bn1(pre1)));  pre2[g] = W2[g] h1[g]         + partial statistics
 *   launch 3   h2 = drop(relu(bn2(pre2)));  y[g] = W3[g] h2[g] + b3[g]
 * The BatchNorm statistics of a channel are merged (Chan) from the partials by every workgroup that needs them: no atomics,
 * no grid barrier, bit-reproducible.  Everything the existing backward reads is written as the separate launches wrote it
 * (pre1, h1, pre2, h2 as [B, G*256, N], save_mean / save_invstd, the same dropout streams: bn_common.h).
 * Needs N % 32 == 0, G <= 8, rows <= 32, 16-B aligned operands; `workspace` of vdetr_heads_workspace_bytes(B, N, G).
 * ---------------------------------------------------------------------------------------------- */
typedef struct vdetr_heads_desc {
  int32_t B, N;            /* scenes, tokens per scene */
  int32_t G, rows;         /* heads of the stage (5), rows of a head's zero-padded output slab */
  const float* x;          /* [N, B, 256] the stage's features, sequence-first (the decoder layer's normed output) */
  const float* w1t;        /* [G][256 in][256 out]: transposed images of the first layers (vdetr_rb_transpose_f32) */
  const float* w2t;        /* [G][256 in][256 out]: of the second layers */
  const float* w3;         /* [G][rows][256] */
  const float* b3;         /* [G][rows] */
  const float *gamma1, *beta1, *gamma2, *beta2;                 /* [G*256] */
  float *running_mean1, *running_var1, *running_mean2, *running_var2; /* [G*256], updated in place (all four or none) */
  int64_t* counters1[8];   /* num_batches_tracked of the G BatchNorm modules of block 1 (NULL entries are skipped): += 1 */
  int64_t* counters2[8];
  float eps, momentum;
  float p1, p2;            /* dropout rates behind the two hidden blocks */
  uint64_t salt1, salt2;   /* their streams (vdetr_bnact_desc.seed of the one-launch-per-block form) */
  const uint64_t* rng_state;
  float *pre1, *h1, *pre2, *h2;                                 /* [B, G*256, N] */
  float *save_mean1, *save_invstd1, *save_mean2, *save_invstd2; /* [G*256] */
  float* y;                /* [B, G, rows, N] */
  void* workspace;
} vdetr_heads_desc;
size_t vdetr_heads_workspace_bytes(int B, int N, int G);
int vdetr_heads_fwd_f32(const vdetr_heads_desc* d, vdetr_stream_t stream);

/* ----------------------------------------------------------------------------------------------
 * PositionEmbeddingLearned (models/helpers.py:17-33: Conv1d(cin, 256) -> BatchNorm1d -> ReLU -> Conv1d(256, 256)) on DETACHED
 * box coordinates, training mode, ONE launch (heads.hip).  The first convolution is linear in `cin` (<= 8) coordinates, so the
 * batch statistics of its 256 outputs follow from the coordinates' mean and covariance (accumulated in fp64 by every workgroup
 * over all B*N tokens): no second pass, no cross-workgroup exchange.
 *   x [B, N, cin] -> hpre [B, 256, N] (first convolution WITHOUT its bias: it cancels under batch statistics and only enters
 *   the running mean), hact [B, 256, N] = relu(bn(hpre)), out [N, B, 256] = hact^T W2^T + b2 (sequence-first, dense).
 * ---------------------------------------------------------------------------------------------- */
typedef struct vdetr_posmlp_desc {
  int32_t B, N, cin;
  const float* x;          /* [B, N, cin] */
  const float* w1;         /* [256][cin] */
  const float* b1;         /* [256] or NULL */
  const float *gamma, *beta;
  float *running_mean, *running_var; /* both or none */
  int64_t* counter;        /* num_batches_tracked or NULL */
  float eps, momentum;
  const float* w2t;        /* [256 in][256 out] transposed image of the second convolution */
  const float* b2;         /* [256] or NULL */
  float *hpre, *hact;      /* [B, 256, N] */
  float *save_mean, *save_invstd; /* [256] */
  float* out;              /* [N, B, 256] */
} vdetr_posmlp_desc;
int vdetr_pos_mlp_fwd_f32(const vdetr_posmlp_desc* d, vdetr_stream_t stream);
/* vdetr_rb_qkv_f32 whose `pos` rows are the position MLP of `m`, computed by the same launch (rowblock.hip: the q and k workgroups
 * form their 16 rows of it on the way in; one launch less in front of every decoder layer).  Writes everything vdetr_pos_mlp_fwd_f32
 * writes (m->out = the pos rows [N, B, 256], hpre, hact, the statistics) and everything vdetr_rb_qkv_f32 writes; d->pos is not read
 * (NULL or m->out).  B * N = d->rows, a multiple of 16. */
int vdetr_rb_qkv_pos_f32(const vdetr_rb_qkv_desc* d, const vdetr_posmlp_desc* m, vdetr_stream_t stream);

/* LDS update-rate probe (mode 0 ds_add_f32, 1 ds_add_u32, 2 plain read-add-write, 3 ds_add_f32 on 8 hot bins):
 * 256 workgroups x 512 threads x `iters` updates.  Measurement hook used by tools/kernel_bench.py --lds. */
int vdetr_selftest_lds_atomics(int mode, int iters, float* sink, vdetr_stream_t stream);

/* Timeline probe: one wave stores the device's constant-rate clock (100 MHz ticks) to *slot, in stream order.  A captured step
 * with a few of these between its phases shows where its streams really are (tools/probes/step_timeline.py). */
int vdetr_probe_timestamp(uint64_t* slot, vdetr_stream_t stream);

/* MFMA layout self-test: C[16,16] = A[16,64] * B[16,64]^T through v_mfma_f32_16x16x4_f32. */
int vdetr_selftest_mfma_f32(const float* a, const float* b, float* c, vdetr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* VDETR_HIP_H */
