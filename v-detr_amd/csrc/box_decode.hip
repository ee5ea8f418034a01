// box_decode.hip — head outputs -> box parameters, corners and class probabilities of one decoder stage, forward and
// backward, as ONE launch each.
//
// Reference: TransformerDecoder.get_proposal_box_predictions_refine (models/vdetr_transformer.py:244-333) with
// BoxProcessor (:20-90), ScannetDatasetConfig.box_parametrization_to_corners (datasets/scannet.py:168-171) and
// get_3d_box_batch_tensor / flip_axis_to_camera_tensor / roty_batch_tensor (utils/box_util.py:294-352).  The reference
// spends ~45 ATen kernels per stage on [B, nQ, <=24] tensors here (and as many again in backward); with 9 stages per
// step that was ~800 of the step's 2,650 dispatches, each of which costs ~4.5 us on MI355X however little it does.
// One thread per (scene, query): channel-major head outputs [B, ch, N] are read coalesced across the queries.
#include "common.h"

namespace vdetr {

constexpr int kMaxAngleBins = 32;
constexpr float kPi = 3.14159265358979323846f;
// corner sign pattern of get_3d_box_batch_tensor (box_util.py:338-346): x = +-l/2, y = +-h/2, z = +-w/2
__device__ __constant__ float kSX[8] = {0.5f, 0.5f, -0.5f, -0.5f, 0.5f, 0.5f, -0.5f, -0.5f};
__device__ __constant__ float kSY[8] = {0.5f, 0.5f, 0.5f, 0.5f, -0.5f, -0.5f, -0.5f, -0.5f};
__device__ __constant__ float kSZ[8] = {0.5f, -0.5f, -0.5f, 0.5f, 0.5f, -0.5f, -0.5f, 0.5f};

// corners of a box: size (l, w, h), yaw (cos c, sin s), centre in the CAMERA frame (box_util.py:319-352)
__device__ __forceinline__ void box_corners(float l, float w, float h, float c, float s, float cx, float cy, float cz,
                                            float* out /* 8 x 3 */) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float lx = l * kSX[i], ly = h * kSY[i], lz = w * kSZ[i];
    out[i * 3 + 0] = (lx * c + lz * s) + cx;  // local @ roty(angle)^T
    out[i * 3 + 1] = ly + cy;
    out[i * 3 + 2] = (lz * c - lx * s) + cz;
  }
}

// One wave per workgroup: a query's ~130 outputs are scattered stores (stride 96 B for the corners), i.e. ~64 memory transactions
// per wave instruction, and the four waves of a 256-thread workgroup queued them all on ONE CU's address unit (15 us for 1024
// queries on 4 CUs); 64-thread workgroups spread the same waves over 16 CUs.
constexpr int kBoxThreads = 64;
__global__ __launch_bounds__(kBoxThreads) void box_decode_fwd_kernel(vdetr_box_decode_desc d) {
  const int t = blockIdx.x * kBoxThreads + threadIdx.x;
  if (t >= d.B * d.N) return;
  const int b = t / d.N, n = t - b * d.N;
  const size_t o3 = (size_t)t * 3;
  // element (b, channel a, n) of a head output with `ch` channels: dense [B,ch,N] or a slab of a joint tensor
  auto in = [&](const float* p, int ch, int a) {
    return p[(size_t)b * (d.in_batch_stride ? (size_t)d.in_batch_stride : (size_t)ch * d.N) + (size_t)a * d.N + n];
  };
  // The class logits of a query are C1 loads at a stride of N floats.  Walked in loops of run-time length (the class block at the
  // end) they were 2-4 x C1 DEPENDENT round trips to L2 in a one-wave workgroup (12.4 us per stage for 1,024 queries); up to
  // kBoxMaxC of them are requested here, at once, in front of everything else.
  const int C1 = d.C1;
  constexpr int kBoxMaxC = 32;
  float cl[kBoxMaxC];
  if (C1 <= kBoxMaxC) {
#pragma unroll
    for (int c = 0; c < kBoxMaxC; ++c) cl[c] = c < C1 ? in(d.cls, C1, c) : -INFINITY;
  }
  float dmin[3], scene[3], pcu[3], psu[3], cu[3], su[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    dmin[a] = d.dims_min[b * 3 + a];
    scene[a] = d.dims_max[b * 3 + a] - dmin[a];
    pcu[a] = d.pre_center_norm[o3 + a] * scene[a] + dmin[a];
    psu[a] = d.pre_size_norm[o3 + a] * scene[a];
    const float creg = in(d.center, 3, a);
    const float sreg = in(d.size, 3, a);
    cu[a] = creg * psu[a] + pcu[a];
    su[a] = expf(sreg) * psu[a];
    d.center_reg[o3 + a] = creg;
    d.size_reg[o3 + a] = sreg;
    d.pre_center_unnorm[o3 + a] = pcu[a];
    d.pre_size_unnorm[o3 + a] = psu[a];
    d.center_unnorm[o3 + a] = cu[a];
    d.center_norm[o3 + a] = (cu[a] - dmin[a]) / scene[a];
    d.size_unnorm[o3 + a] = su[a];
    d.size_norm[o3 + a] = su[a] / scene[a];
  }
  // ---- angle (BoxProcessor.compute_predicted_angle, :48-71) --------------------------------------------------------
  const int A = d.A;
  float angle = 0.f, prob = 0.f;
  int cls = 0;
  const float res_scale = kPi / (float)A;
  if (A == 1) {
    const float al = in(d.angle_cls, 1, 0), arn = in(d.angle_res, 1, 0), ar = arn * res_scale;
    d.angle_residual[t] = ar;
    if (d.angle_logits_t) d.angle_logits_t[t] = al;
    if (d.angle_res_norm_t) d.angle_res_norm_t[t] = arn;
    angle = fmaxf(al * 0.f + ar * 0.f, 0.f);  // (:53-55) the head outputs stay in the graph, multiplied by zero
    prob = angle;
  } else {
    float mx = -INFINITY;
    for (int a = 0; a < A; ++a) {
      const float al = in(d.angle_cls, A, a);
      if (al > mx) { mx = al; cls = a; }  // first maximum
    }
    float den = 0.f, rsel = 0.f;
    for (int a = 0; a < A; ++a) {
      const float al = in(d.angle_cls, A, a), arn = in(d.angle_res, A, a);
      den += expf(al - mx);
      const float ar = arn * res_scale;
      d.angle_residual[(size_t)t * A + a] = ar;
      if (d.angle_logits_t) d.angle_logits_t[(size_t)t * A + a] = al;
      if (d.angle_res_norm_t) d.angle_res_norm_t[(size_t)t * A + a] = arn;
      if (a == cls) rsel = ar;
    }
    prob = 1.f / den;
    angle = (2.f * kPi / (float)d.num_angle_bin) * (float)cls + rsel;
    if (angle > kPi) angle -= 2.f * kPi;
  }
  d.angle_cont[t] = angle;
  d.angle_prob[t] = prob;
  d.angle_class[t] = cls;
  // ---- corners (camera frame: (x, -z, y), box_util.py:294-301) -----------------------------------------------------
  float cor[24];
  box_corners(su[0], su[1], su[2], cosf(angle), sinf(angle), cu[0], -cu[2], cu[1], cor);
#pragma unroll
  for (int i = 0; i < 24; ++i) d.corners[(size_t)t * 24 + i] = cor[i];
  if (d.corners_lidar) {  // convert_corners_camera2lidar (:98-102): (x, y, z) -> (x, z, -y); what the next layer's RPE reads
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      d.corners_lidar[(size_t)t * 24 + i * 3] = cor[i * 3];
      d.corners_lidar[(size_t)t * 24 + i * 3 + 1] = cor[i * 3 + 2];
      d.corners_lidar[(size_t)t * 24 + i * 3 + 2] = -cor[i * 3 + 1];
    }
  }
  if (d.center_size) {  // torch.cat([center_unnormalized, size_unnormalized], -1) (:415): input of the query-pos MLP
#pragma unroll
    for (int a = 0; a < 3; ++a) { d.center_size[(size_t)t * 6 + a] = cu[a]; d.center_size[(size_t)t * 6 + 3 + a] = su[a]; }
  }
  if (d.corners_aa) {  // zero-angle corners (:311-316); with one angle bin the caller reuses `corners`
    box_corners(su[0], su[1], su[2], 1.f, 0.f, cu[0], -cu[2], cu[1], cor);
#pragma unroll
    for (int i = 0; i < 24; ++i) d.corners_aa[(size_t)t * 24 + i] = cor[i];
  }
  // ---- class probabilities (BoxProcessor.compute_objectness_and_cls_prob, :73-86; no gradient) ---------------------
  if (C1 <= kBoxMaxC) {  // (the logits were requested at the top)
    if (d.cls_logits_t) {
#pragma unroll
      for (int c = 0; c < kBoxMaxC; ++c)
        if (c < C1) d.cls_logits_t[(size_t)t * C1 + c] = cl[c];
    }
    float mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < kBoxMaxC; ++c) mx = fmaxf(mx, cl[c]);  // (the padding is -inf)
    if (d.cls_kind == VDETR_CLS_SOFTMAX) {
      float den = 0.f;
#pragma unroll
      for (int c = 0; c < kBoxMaxC; ++c)
        if (c < C1) den += expf(cl[c] - mx);  // same order of the sum as the loop form below
      const float inv = 1.f / den;
      float last = 0.f;
#pragma unroll
      for (int c = 0; c < kBoxMaxC; ++c) {
        if (c < C1) {
          const float p = expf(cl[c] - mx) * inv;
          if (c < C1 - 1) d.cls_prob[(size_t)t * (C1 - 1) + c] = p;
          else last = p;
        }
      }
      d.objectness[t] = 1.f - last;
    } else {  // focal loss: sem_cls_prob IS the logits (a view on the caller's side); objectness = max sigmoid
      d.objectness[t] = 1.f / (1.f + expf(-mx));
    }
    return;
  }
  if (d.cls_logits_t)
    for (int c = 0; c < C1; ++c) d.cls_logits_t[(size_t)t * C1 + c] = in(d.cls, C1, c);
  if (d.cls_kind == VDETR_CLS_SOFTMAX) {
    float mx = -INFINITY;
    for (int c = 0; c < C1; ++c) mx = fmaxf(mx, in(d.cls, C1, c));
    float den = 0.f;
    for (int c = 0; c < C1; ++c) den += expf(in(d.cls, C1, c) - mx);
    const float inv = 1.f / den;
    float last = 0.f;
    for (int c = 0; c < C1; ++c) {
      const float p = expf(in(d.cls, C1, c) - mx) * inv;
      if (c < C1 - 1) d.cls_prob[(size_t)t * (C1 - 1) + c] = p;
      else last = p;
    }
    d.objectness[t] = 1.f - last;
  } else {  // focal loss: sem_cls_prob IS the logits (a view on the caller's side); objectness = max sigmoid
    float mx = -INFINITY;
    for (int c = 0; c < C1; ++c) mx = fmaxf(mx, in(d.cls, C1, c));
    d.objectness[t] = 1.f / (1.f + expf(-mx));
  }
}

// Backward: every incoming gradient pointer may be NULL (that output was not used).
__device__ __forceinline__ void box_decode_bwd_body(const vdetr_box_decode_desc& d, const vdetr_box_decode_grads& g, int t) {
  if (t >= d.B * d.N) return;
  const int b = t / d.N, n = t - b * d.N;
  const size_t o3 = (size_t)t * 3;
  auto ld = [](const float* p, size_t i) { return p ? p[i] : 0.f; };
  auto in = [&](const float* p, int ch, int a) {
    return p[(size_t)b * (d.in_batch_stride ? (size_t)d.in_batch_stride : (size_t)ch * d.N) + (size_t)a * d.N + n];
  };
  // gradient element (b, channel a, n) of a head output; with slabs (g.slab_rows > 0) the rows [ch, slab_rows) are padding
  auto out = [&](float* p, int ch, int a) -> float& {
    return p[(size_t)b * (g.out_batch_stride ? (size_t)g.out_batch_stride : (size_t)ch * d.N) + (size_t)a * d.N + n];
  };
  auto pad = [&](float* p, int ch) {
    for (int a = ch; a < g.slab_rows; ++a) out(p, ch, a) = 0.f;
  };
  float scene[3], psu[3], su[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    scene[a] = d.dims_max[b * 3 + a] - d.dims_min[b * 3 + a];
    psu[a] = d.pre_size_unnorm[o3 + a];
    su[a] = d.size_unnorm[o3 + a];
  }
  // gradients of the unnormalised centre / size / angle, starting with the direct uses
  float gcu[3], gsu[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    gcu[a] = ld(g.center_unnorm, o3 + a) + ld(g.center_norm, o3 + a) / scene[a];
    gsu[a] = ld(g.size_unnorm, o3 + a) + ld(g.size_norm, o3 + a) / scene[a];
  }
  float gang = ld(g.angle_cont, t);
  const float angle = d.angle_cont[t];
  // corners: x' = lx c + lz s + cx, y' = ly + cy, z' = lz c - lx s + cz, (lx, ly, lz) = (l sx, h sy, w sz)
  auto corner_grads = [&](const float* gc, float c, float s, bool with_angle) {
    if (!gc) return;
    float gl = 0.f, gw = 0.f, gh = 0.f, gx = 0.f, gy = 0.f, gz = 0.f, gth = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float gxi = gc[(size_t)t * 24 + i * 3], gyi = gc[(size_t)t * 24 + i * 3 + 1], gzi = gc[(size_t)t * 24 + i * 3 + 2];
      gx += gxi; gy += gyi; gz += gzi;
      gl += kSX[i] * (c * gxi - s * gzi);
      gh += kSY[i] * gyi;
      gw += kSZ[i] * (s * gxi + c * gzi);
      const float lx = su[0] * kSX[i], lz = su[1] * kSZ[i];
      gth += gxi * (lz * c - lx * s) + gzi * (-lz * s - lx * c);
    }
    gsu[0] += gl; gsu[1] += gw; gsu[2] += gh;
    gcu[0] += gx; gcu[2] -= gy; gcu[1] += gz;  // camera (x, -z, y)
    if (with_angle) gang += gth;
  };
  corner_grads(g.corners, cosf(angle), sinf(angle), true);
  corner_grads(g.corners_aa, 1.f, 0.f, false);
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    out(g.d_center, 3, a) = ld(g.center_reg, o3 + a) + gcu[a] * psu[a];
    out(g.d_size, 3, a) = ld(g.size_reg, o3 + a) + gsu[a] * su[a];
  }
  pad(g.d_center, 3);
  pad(g.d_size, 3);
  if (g.d_cls) {  // the class logits are handed out transposed ([B,N,C1]); their gradient comes back the same way
    for (int c = 0; c < d.C1; ++c) out(g.d_cls, d.C1, c) = ld(g.cls_logits_t, (size_t)t * d.C1 + c);
    pad(g.d_cls, d.C1);
  }
  const int A = d.A;
  const float res_scale = kPi / (float)A;
  if (A == 1) {
    // angle = clamp(0*logit + 0*residual, 0): the head outputs only receive the direct residual gradient (x0 paths
    // contribute exact zeros unless the incoming gradient is not finite, which the reference would also spread)
    const float z = (gang + ld(g.angle_prob, t)) * 0.f;
    out(g.d_angle_cls, 1, 0) = z + ld(g.angle_logits_t, t);
    out(g.d_angle_res, 1, 0) = ld(g.angle_residual, t) * res_scale + z * res_scale + ld(g.angle_res_norm_t, t);
  } else {
    const int cls = d.angle_class[t];
    const float gp = ld(g.angle_prob, t);
    float mx = -INFINITY;
    for (int a = 0; a < A; ++a) mx = fmaxf(mx, in(d.angle_cls, A, a));
    float den = 0.f;
    for (int a = 0; a < A; ++a) den += expf(in(d.angle_cls, A, a) - mx);
    const float pc = 1.f / den;  // = softmax at the arg-max
    for (int a = 0; a < A; ++a) {
      const float pa = expf(in(d.angle_cls, A, a) - mx) / den;
      out(g.d_angle_cls, A, a) = gp * pc * ((a == cls ? 1.f : 0.f) - pa) + ld(g.angle_logits_t, (size_t)t * A + a);
      out(g.d_angle_res, A, a) = (ld(g.angle_residual, (size_t)t * A + a) + (a == cls ? gang : 0.f)) * res_scale +
                                 ld(g.angle_res_norm_t, (size_t)t * A + a);
    }
  }
  pad(g.d_angle_cls, A);
  pad(g.d_angle_res, A);
}

__global__ __launch_bounds__(kBoxThreads) void box_decode_bwd_kernel(vdetr_box_decode_desc d, vdetr_box_decode_grads g) {
  box_decode_bwd_body(d, g, blockIdx.x * kBoxThreads + threadIdx.x);
}

// Several stages' backward in one launch (vdetr_box_decode_bwd_batch_f32): blockIdx.y = stage.  The decoder's nine stages are
// differentiated together at the head of the backward pass (vdetr_transformer._DeferredHeads): nine dependent 8-us launches there
// were 0.1 ms of the step.
constexpr int kBoxBatch = 8;  // (8 x 432 B of descriptors: inside the 4 KB of kernel arguments)
struct BoxBwdBatch {
  vdetr_box_decode_desc d[kBoxBatch];
  vdetr_box_decode_grads g[kBoxBatch];
};
static_assert(sizeof(BoxBwdBatch) <= 3584, "box_decode_bwd_batch: descriptors exceed the kernel-argument budget");
__global__ __launch_bounds__(kBoxThreads) void box_decode_bwd_batch_kernel(BoxBwdBatch Bt) {
  const int s = blockIdx.y;
  box_decode_bwd_body(Bt.d[s], Bt.g[s], blockIdx.x * kBoxThreads + threadIdx.x);
}

}  // namespace vdetr

using namespace vdetr;

static int box_check(const vdetr_box_decode_desc* d, const char* op) {
  VDETR_REQUIRE(d != nullptr, "%s: null descriptor", op);
  VDETR_REQUIRE(d->B > 0 && d->N > 0 && d->A > 0 && d->C1 > 0, "%s: empty dimension B=%d N=%d A=%d C1=%d", op, d->B,
                d->N, d->A, d->C1);
  VDETR_REQUIRE(d->A <= kMaxAngleBins, "%s: %d angle bins > %d", op, d->A, kMaxAngleBins);
  VDETR_REQUIRE(d->num_angle_bin > 0, "%s: num_angle_bin must be positive", op);
  VDETR_REQUIRE(d->cls_kind == VDETR_CLS_SOFTMAX || d->cls_kind == VDETR_CLS_SIGMOID, "%s: bad cls_kind %d", op, d->cls_kind);
  VDETR_REQUIRE(d->center && d->size && d->angle_cls && d->angle_res && d->cls && d->pre_center_norm && d->pre_size_norm &&
                    d->dims_min && d->dims_max, "%s: null input pointer", op);
  return VDETR_OK;
}

// ---- the encoder proposals' anchor boxes and the gather of the top proposals: two launches for ~25 tensor expressions --------------
namespace vdetr {
// models/model_vdetr.py:348-362 — class = arg max sigmoid(point-class logit) (first maximum), size = that class's anchor, centre = the
// token; convert_unnorm2norm (:383-390) of both; corners at yaw 0 (box_util.py:319-352 on the camera-frame centre (x, -z, y)).
__global__ __launch_bounds__(kBoxThreads) void anchor_boxes_kernel(const float* __restrict__ logits, const float* __restrict__ xyz,
                                                                   const float* __restrict__ dims_min, const float* __restrict__ dims_max,
                                                                   const float* __restrict__ anchors, int B, int N, int ncls,
                                                                   float* __restrict__ size_unnorm, float* __restrict__ center_norm,
                                                                   float* __restrict__ size_norm, float* __restrict__ corners) {
  const int t = blockIdx.x * kBoxThreads + threadIdx.x;
  if (t >= B * N) return;
  const int b = t / N;
  const float* lg = logits + (size_t)t * ncls;
  int cls = 0;
  float best = -INFINITY;
  for (int c = 0; c < ncls; ++c) {
    const float p = 1.f / (1.f + expf(-lg[c]));  // (the probabilities, not the logits: saturated values tie as they do in the reference)
    if (p > best) { best = p; cls = c; }
  }
  float su[3], cu[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float lo = dims_min[b * 3 + a], scene = dims_max[b * 3 + a] - lo;
    su[a] = anchors[cls * 3 + a];
    cu[a] = xyz[(size_t)t * 3 + a];
    size_unnorm[(size_t)t * 3 + a] = su[a];
    center_norm[(size_t)t * 3 + a] = (cu[a] - lo) / scene;
    size_norm[(size_t)t * 3 + a] = su[a] / scene;
  }
  float cor[24];
  box_corners(su[0], su[1], su[2], 1.f, 0.f, cu[0], -cu[2], cu[1], cor);
#pragma unroll
  for (int i = 0; i < 24; ++i) corners[(size_t)t * 24 + i] = cor[i];
}

// models/vdetr_transformer.py:364-398 — the boxes of the nq highest-objectness tokens (indices given), as the first decoder layer reads them
struct ProposalArgs {
  const long long* topk;
  int B, N, nq, camera;
  const float *corners, *center, *size, *angle, *center_norm, *size_norm;
  float *o_corners, *o_center, *o_size, *o_angle, *o_center_norm, *o_size_norm, *o_query_ref;
};
__global__ __launch_bounds__(kBoxThreads) void gather_proposals_kernel(ProposalArgs A) {
  const int t = blockIdx.x * kBoxThreads + threadIdx.x;
  if (t >= A.B * A.nq) return;
  const int b = t / A.nq;
  long long k = A.topk[t];
  k = k < 0 ? 0 : (k >= A.N ? A.N - 1 : k);
  const size_t src = (size_t)b * A.N + (size_t)k;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float x = A.corners[src * 24 + i * 3], y = A.corners[src * 24 + i * 3 + 1], z = A.corners[src * 24 + i * 3 + 2];
    // convert_corners_camera2lidar (:98-102): (x, y, z) -> (x, z, -y)
    A.o_corners[(size_t)t * 24 + i * 3] = x;
    A.o_corners[(size_t)t * 24 + i * 3 + 1] = A.camera ? z : y;
    A.o_corners[(size_t)t * 24 + i * 3 + 2] = A.camera ? -y : z;
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float c = A.center[src * 3 + a], s = A.size[src * 3 + a];
    A.o_center[(size_t)t * 3 + a] = c;
    A.o_size[(size_t)t * 3 + a] = s;
    A.o_query_ref[(size_t)t * 6 + a] = c;
    A.o_query_ref[(size_t)t * 6 + 3 + a] = s;
    A.o_center_norm[(size_t)t * 3 + a] = A.center_norm[src * 3 + a];
    A.o_size_norm[(size_t)t * 3 + a] = A.size_norm[src * 3 + a];
  }
  A.o_angle[t] = A.angle[src];
}
}  // namespace vdetr

extern "C" int vdetr_anchor_boxes_f32(const float* logits, const float* xyz, const float* dims_min, const float* dims_max, const float* anchors,
                                      int B, int N, int ncls, float* size_unnorm, float* center_norm, float* size_norm, float* corners,
                                      vdetr_stream_t stream) {
  VDETR_REQUIRE(logits && xyz && dims_min && dims_max && anchors && size_unnorm && center_norm && size_norm && corners, "anchor_boxes: null pointer");
  VDETR_REQUIRE(B > 0 && N > 0 && ncls > 0, "anchor_boxes: bad shape B=%d N=%d ncls=%d", B, N, ncls);
  hipLaunchKernelGGL(vdetr::anchor_boxes_kernel, dim3(vdetr::ceil_div((long)B * N, vdetr::kBoxThreads)), dim3(vdetr::kBoxThreads), 0, (hipStream_t)stream,
                     logits, xyz, dims_min, dims_max, anchors, B, N, ncls, size_unnorm, center_norm, size_norm, corners);
  return vdetr::check_launch("anchor_boxes");
}

extern "C" int vdetr_gather_proposals_f32(const long long* topk, int B, int N, int nq, int corners_are_camera, const float* corners,
                                          const float* center, const float* size, const float* angle, const float* center_norm,
                                          const float* size_norm, float* o_corners_lidar, float* o_center, float* o_size, float* o_angle,
                                          float* o_center_norm, float* o_size_norm, float* o_query_ref, vdetr_stream_t stream) {
  VDETR_REQUIRE(topk && corners && center && size && angle && center_norm && size_norm && o_corners_lidar && o_center && o_size && o_angle &&
                    o_center_norm && o_size_norm && o_query_ref, "gather_proposals: null pointer");
  VDETR_REQUIRE(B > 0 && N > 0 && nq > 0, "gather_proposals: bad shape B=%d N=%d nq=%d", B, N, nq);
  vdetr::ProposalArgs A = {topk, B, N, nq, corners_are_camera ? 1 : 0, corners, center, size, angle, center_norm, size_norm,
                           o_corners_lidar, o_center, o_size, o_angle, o_center_norm, o_size_norm, o_query_ref};
  hipLaunchKernelGGL(vdetr::gather_proposals_kernel, dim3(vdetr::ceil_div((long)B * nq, vdetr::kBoxThreads)), dim3(vdetr::kBoxThreads), 0,
                     (hipStream_t)stream, A);
  return vdetr::check_launch("gather_proposals");
}

extern "C" int vdetr_box_decode_fwd_f32(const vdetr_box_decode_desc* d, vdetr_stream_t stream) {
  if (int e = box_check(d, "box_decode_fwd")) return e;
  VDETR_REQUIRE(d->center_reg && d->size_reg && d->center_unnorm && d->center_norm && d->size_unnorm && d->size_norm &&
                    d->pre_center_unnorm && d->pre_size_unnorm && d->angle_residual && d->angle_cont && d->angle_prob &&
                    d->angle_class && d->corners && d->objectness,
                "box_decode_fwd: null output pointer");
  VDETR_REQUIRE(d->cls_kind != VDETR_CLS_SOFTMAX || d->cls_prob, "box_decode_fwd: cls_prob is required for the softmax kind");
  hipLaunchKernelGGL(box_decode_fwd_kernel, dim3(ceil_div((long)d->B * d->N, kBoxThreads)), dim3(kBoxThreads), 0, (hipStream_t)stream, *d);
  return check_launch("box_decode_fwd");
}

extern "C" int vdetr_box_decode_bwd_f32(const vdetr_box_decode_desc* d, const vdetr_box_decode_grads* g,
                                        vdetr_stream_t stream) {
  if (int e = box_check(d, "box_decode_bwd")) return e;
  VDETR_REQUIRE(g != nullptr, "box_decode_bwd: null gradient block");
  VDETR_REQUIRE(d->size_unnorm && d->pre_size_unnorm && d->angle_cont && d->angle_class,
                "box_decode_bwd: the forward's size_unnorm / pre_size_unnorm / angle_cont / angle_class are required");
  VDETR_REQUIRE(g->d_center && g->d_size && g->d_angle_cls && g->d_angle_res, "box_decode_bwd: null output pointer");
  hipLaunchKernelGGL(box_decode_bwd_kernel, dim3(ceil_div((long)d->B * d->N, kBoxThreads)), dim3(kBoxThreads), 0, (hipStream_t)stream, *d, *g);
  return check_launch("box_decode_bwd");
}

static int box_bwd_check(const vdetr_box_decode_desc* d, const vdetr_box_decode_grads* g) {
  if (int e = box_check(d, "box_decode_bwd")) return e;
  VDETR_REQUIRE(g != nullptr, "box_decode_bwd: null gradient block");
  VDETR_REQUIRE(d->size_unnorm && d->pre_size_unnorm && d->angle_cont && d->angle_class,
                "box_decode_bwd: the forward's size_unnorm / pre_size_unnorm / angle_cont / angle_class are required");
  VDETR_REQUIRE(g->d_center && g->d_size && g->d_angle_cls && g->d_angle_res, "box_decode_bwd: null output pointer");
  return VDETR_OK;
}

extern "C" int vdetr_box_decode_bwd_batch_f32(const vdetr_box_decode_desc* d, const vdetr_box_decode_grads* g, int n,
                                              vdetr_stream_t stream) {
  VDETR_REQUIRE(d && g && n > 0, "box_decode_bwd_batch: null pointer or n=%d", n);
  for (int i0 = 0; i0 < n; i0 += kBoxBatch) {
    const int m = n - i0 < kBoxBatch ? n - i0 : kBoxBatch;
    BoxBwdBatch Bt{};
    long rows = 0;
    for (int i = 0; i < m; ++i) {
      if (int e = box_bwd_check(d + i0 + i, g + i0 + i)) return e;
      Bt.d[i] = d[i0 + i];
      Bt.g[i] = g[i0 + i];
      const long r = (long)d[i0 + i].B * d[i0 + i].N;
      rows = r > rows ? r : rows;
    }
    hipLaunchKernelGGL(box_decode_bwd_batch_kernel, dim3(ceil_div(rows, kBoxThreads), m), dim3(kBoxThreads), 0, (hipStream_t)stream, Bt);
    if (int e = check_launch("box_decode_bwd_batch")) return e;
  }
  return VDETR_OK;
}
