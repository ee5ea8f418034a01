// criterion.hip — the set criterion on the device (reference criterion.py).
//
// The reference builds three [B,P,G] matrices with ~60 torch launches per stage, copies the cost to the host, runs
// scipy's Hungarian solver there (9 x per step, each behind a device->host sync), copies the assignment back and
// gathers the matched losses from the matrices.  Here, per stage:
//   match_cost_kernel   one launch: GIoU + centre + size + class terms -> cost, stored box-major so the solver reads rows
//   lsa_kernel          ONE launch for all stages: a workgroup per (stage, scene) runs the same shortest-augmenting-path
//                       algorithm as scipy in fp64, columns spread over the lanes, state in registers / LDS
//   set_loss_kernel     one launch: matched losses AND their gradients (the pairwise matrices are never needed again)
// Everything is latency-bound integer/fp32 work of a few hundred KB; the design goals are launch count, no host round
// trip (the whole step stays capturable in a hipGraph) and bit-identical assignments.
#include "wave.h"

#include <stdlib.h>

namespace vdetr {
namespace {

constexpr int F = VDETR_GT_FLOATS;

// ------------------------------------------------------------------------------------------------ ground truth
// One workgroup; scenes in sequence (B is the per-GPU batch, 1..8).
__global__ __launch_bounds__(256) void gt_prepare_kernel(const float* __restrict__ gt, int B, int G, int repeat,
                                                         float* __restrict__ gt_rep, int64_t* __restrict__ nactual,
                                                         int64_t* __restrict__ nactual_rep, float* __restrict__ sums) {
  extern __shared__ int prefix[];  // [G + 1] exclusive count of present boxes; [G + 1] = rotated-box flag
  int total = 0;
  bool rotated = false;
  for (int b = 0; b < B; ++b) {
    const float* src = gt + (size_t)b * G * F;
    if (threadIdx.x == 0) {
      int n = 0;
      for (int s = 0; s < G; ++s) {
        prefix[s] = n;
        const bool present = src[s * F + VDETR_GT_PRESENT] > 0.f;
        n += present;
        // the reference tests every slot, and absent slots are zero (criterion.py:616 on the padded tensor)
        rotated |= src[s * F + VDETR_GT_ANGLE] > 0.f;
      }
      prefix[G] = n;
    }
    __syncthreads();
    const int n = prefix[G];
    total += n;
    if (gt_rep != nullptr) {
      float* dst = gt_rep + (size_t)b * G * repeat * F;
      const int slots = G * repeat;
      // present boxes of every tile move to the front, tile after tile (stable)
      for (int e = threadIdx.x; e < slots * F; e += blockDim.x) {
        const int t = e / F, f = e - t * F;
        const int s = t % G, tile = t / G;
        if (src[s * F + VDETR_GT_PRESENT] > 0.f) dst[(tile * n + prefix[s]) * F + f] = src[s * F + f];
      }
      for (int e = n * repeat * F + threadIdx.x; e < slots * F; e += blockDim.x) dst[e] = 0.f;
    }
    if (threadIdx.x == 0) {
      nactual[b] = n;
      if (nactual_rep != nullptr) nactual_rep[b] = (int64_t)n * repeat;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    sums[0] = (float)total;
    sums[1] = (float)(total * repeat);
    sums[2] = rotated ? 1.f : 0.f;  // criterion.py:616: torch.any(gt_box_angles > 0) selects the polygon-clip overlap
  }
}

// ------------------------------------------------------------------------------------------------ box geometry
struct BoxGeo {           // what the GIoU needs of one box given as 8 corners (camera frame, y down)
  float mn[3], mx[3];     // extent over the corners
  float c0x, c0y, c0z;    // corner 0 = (+l/2, +h/2, +w/2) side
  float c2x, c2z;         // corner 2 = (-l/2, ., -w/2)
  float c4y;              // corner 4: the other y face
  float vol;              // clamped volume from three edges (box_util.py:441-463)
};

__device__ __forceinline__ float edge_len(const float* c, int i, int j) {
  const float dx = c[i * 3] - c[j * 3], dy = c[i * 3 + 1] - c[j * 3 + 1], dz = c[i * 3 + 2] - c[j * 3 + 2];
  return sqrtf(fmaxf((dx * dx + dy * dy) + dz * dz, 1e-6f));
}

template <typename Ptr>
__device__ __forceinline__ BoxGeo box_geo(Ptr c) {
  BoxGeo g;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float lo = c[a], hi = c[a];
#pragma unroll
    for (int k = 1; k < 8; ++k) {
      lo = fminf(lo, c[k * 3 + a]);
      hi = fmaxf(hi, c[k * 3 + a]);
    }
    g.mn[a] = lo;
    g.mx[a] = hi;
  }
  g.c0x = c[0], g.c0y = c[1], g.c0z = c[2];
  g.c2x = c[6], g.c2z = c[8];
  g.c4y = c[13];
  float cc[24];
#pragma unroll
  for (int k = 0; k < 24; ++k) cc[k] = c[k];
  g.vol = fmaxf((edge_len(cc, 0, 1) * edge_len(cc, 1, 2)) * edge_len(cc, 0, 4), 1e-8f);
  return g;
}

// generalized_box3d_iou_tensor with rotated_boxes=False (box_util.py:523-600) for one pair
// `area` = footprint overlap: wx * wz for axis-aligned boxes, the polygon clip (where wx * wz != 0) for rotated ones
__device__ __forceinline__ float giou_pair(const BoxGeo& p, const BoxGeo& g, float area) {
  const float height = fmaxf(fminf(p.c0y, g.c0y) - fmaxf(p.c4y, g.c4y), 0.f);
  const float ex = fabsf(fmaxf(p.mx[0], g.mx[0]) - fminf(p.mn[0], g.mn[0]));
  const float ey = fabsf(fminf(-p.mx[1], -g.mx[1]) - fmaxf(-p.mn[1], -g.mn[1]));
  const float ez = fabsf(fmaxf(p.mx[2], g.mx[2]) - fminf(p.mn[2], g.mn[2]));
  const float enclosing = (ex * ey) * ez;
  const float total = p.vol + g.vol;
  const float inter = area * height;
  const float uni = fmaxf(total - inter, 1e-8f);
  const float giou = inter / uni + (-(1.f - uni / enclosing));
  return (enclosing > 2e-8f && total > 4e-8f) ? giou : giou * 0.f;
}

__device__ __forceinline__ float aligned_overlap(const BoxGeo& p, const BoxGeo& g) {
  const float wx = fmaxf(fminf(p.c0x, g.c0x) - fmaxf(p.c2x, g.c2x), 0.f);
  const float wz = fmaxf(fminf(p.c0z, g.c0z) - fmaxf(p.c2z, g.c2z), 0.f);
  return wx * wz;
}

// ---- rotated boxes: footprint overlap by Sutherland-Hodgman clipping (box_util.py:393-439, 566-589) ----------------
// A footprint is the quadrilateral of corners 3,2,1,0 in (x,z) (counter-clockwise, box_util.py:539-544).  The subject is
// the prediction, the clip polygon the ground-truth box.  NT > 0 carries NT forward-mode tangents per coordinate (the 8
// footprint coordinates of the prediction), which yields d area / d corners without recording the clip sequence.
template <int NT>
struct ClipPoly {
  float x[8], y[8];
  float tx[NT > 0 ? 8 : 1][NT > 0 ? NT : 1], ty[NT > 0 ? 8 : 1][NT > 0 ? NT : 1];
};

__device__ __forceinline__ bool clip_inside(float c1x, float c1y, float c2x, float c2y, float px, float py) {
  return (c2x - c1x) * (py - c1y) > (c2y - c1y) * (px - c1x);
}

// twice the area of subject clipped by clip (abs of the shoelace sum); grad[t] = d(that)/d(input t) when NT > 0
template <int NT>
__device__ float clip_area2(const float* sx, const float* sy, const float* cx, const float* cy, float* grad) {
  ClipPoly<NT> a, b;
  ClipPoly<NT>* in = &a;
  ClipPoly<NT>* out = &b;
  int n = 4;
  for (int i = 0; i < 4; ++i) {
    out->x[i] = sx[i], out->y[i] = sy[i];
    if (NT > 0)
      for (int t = 0; t < NT; ++t) out->tx[i][t] = t == 2 * i ? 1.f : 0.f, out->ty[i][t] = t == 2 * i + 1 ? 1.f : 0.f;
  }
  float c1x = cx[3], c1y = cy[3];
  for (int ce = 0; ce < 4; ++ce) {
    const float c2x = cx[ce], c2y = cy[ce];
    ClipPoly<NT>* tmp = in;
    in = out;
    out = tmp;
    const int nin = n;
    n = 0;
    int si = nin - 1;
    for (int ei = 0; ei < nin; ++ei) {
      const float ex = in->x[ei], ey = in->y[ei], px = in->x[si], py = in->y[si];
      const bool e_in = clip_inside(c1x, c1y, c2x, c2y, ex, ey);
      const bool s_in = clip_inside(c1x, c1y, c2x, c2y, px, py);
      if (e_in != s_in) {  // the edge s -> e crosses the clip line: helper_computeIntersection
        const float dcx = c1x - c2x, dcy = c1y - c2y;
        const float dpx = px - ex, dpy = py - ey;
        const float n1 = c1x * c2y - c1y * c2x;
        const float n2 = px * ey - py * ex;
        const float den = dcx * dpy - dcy * dpx;
        const float n3 = 1.f / den;
        const float ax = n1 * dpx - n2 * dcx, ay = n1 * dpy - n2 * dcy;
        out->x[n] = ax * n3, out->y[n] = ay * n3;
        if (NT > 0)
          for (int t = 0; t < NT; ++t) {
            const float dpx_t = in->tx[si][t] - in->tx[ei][t], dpy_t = in->ty[si][t] - in->ty[ei][t];
            const float n2_t = in->tx[si][t] * ey + px * in->ty[ei][t] - in->ty[si][t] * ex - py * in->tx[ei][t];
            const float n3_t = -(dcx * dpy_t - dcy * dpx_t) * n3 * n3;
            out->tx[n][t] = (n1 * dpx_t - n2_t * dcx) * n3 + ax * n3_t;
            out->ty[n][t] = (n1 * dpy_t - n2_t * dcy) * n3 + ay * n3_t;
          }
        ++n;
      }
      if (e_in) {
        out->x[n] = ex, out->y[n] = ey;
        if (NT > 0)
          for (int t = 0; t < NT; ++t) out->tx[n][t] = in->tx[ei][t], out->ty[n][t] = in->ty[ei][t];
        ++n;
      }
      si = ei;
    }
    c1x = c2x, c1y = c2y;
    if (n == 0) {
      if (NT > 0)
        for (int t = 0; t < NT; ++t) grad[t] = 0.f;
      return 0.f;
    }
  }
  // xs . roll(ys, 1) - ys . roll(xs, 1)
  float sum = 0.f;
  for (int i = 0; i < n; ++i) {
    const int j = i == 0 ? n - 1 : i - 1;
    sum += out->x[i] * out->y[j];
  }
  float sum2 = 0.f;
  for (int i = 0; i < n; ++i) {
    const int j = i == 0 ? n - 1 : i - 1;
    sum2 += out->y[i] * out->x[j];
  }
  const float raw = sum - sum2;
  if (NT > 0) {
    const float sg = raw > 0.f ? 1.f : (raw < 0.f ? -1.f : 0.f);
    for (int t = 0; t < NT; ++t) {
      float g = 0.f;
      for (int i = 0; i < n; ++i) {
        const int j = i == 0 ? n - 1 : i - 1;
        g += out->tx[i][t] * out->y[j] + out->x[i] * out->ty[j][t] - out->ty[i][t] * out->x[j] - out->y[i] * out->tx[j][t];
      }
      grad[t] = sg * g;
    }
  }
  return fabsf(raw);
}

// footprint (x,z) of corners 3,2,1,0
template <typename Ptr>
__device__ __forceinline__ void footprint(Ptr c, float* fx, float* fz) {
#pragma unroll
  for (int i = 0; i < 4; ++i) fx[i] = c[(3 - i) * 3], fz[i] = c[(3 - i) * 3 + 2];
}

__device__ __forceinline__ float huber1(float e) {
  const float a = fabsf(e), q = fminf(a, 1.f);
  return (0.5f * q) * q + (a - q);
}

// ------------------------------------------------------------------------------------------------ matcher cost
constexpr int kMatchBoxes = 16;  // ground-truth boxes per workgroup

struct GtDerived {
  BoxGeo geo;
  float fx[4], fz[4];  // footprint (rotated boxes)
  float center[3], size[3];
  float ares_norm;
  int label, alabel;
};

constexpr int kCritBatch = 12;  // stage descriptors per launch (kernel arguments: 12 x 232 B < 4 KB)
struct MatchBatch {
  vdetr_match_desc d[kCritBatch];
};
struct LossBatch {
  vdetr_setloss_desc d[kCritBatch];
};

// All stages of a step in ONE launch: blockIdx.z enumerates (stage, scene); the grid is sized for the largest stage and
// the workgroups outside a smaller stage's extent leave at once.
__global__ __launch_bounds__(256) void match_cost_kernel(MatchBatch batch) {
  __shared__ GtDerived sh[kMatchBoxes];
  int z = blockIdx.z, stage = 0;
  while (z >= batch.d[stage].B) z -= batch.d[stage++].B;
  const vdetr_match_desc d = batch.d[stage];
  const int b = z;
  const int g0 = blockIdx.y * kMatchBoxes;
  if (g0 >= d.G || (int)(blockIdx.x * blockDim.x) >= d.P) return;
  const int ng = min(kMatchBoxes, d.G - g0);
  if ((int)threadIdx.x < ng) {
    const float* r = d.gt + ((size_t)b * d.G + g0 + threadIdx.x) * F;
    GtDerived& s = sh[threadIdx.x];
    s.geo = box_geo(r + VDETR_GT_CORNERS);
    footprint(r + VDETR_GT_CORNERS, s.fx, s.fz);
#pragma unroll
    for (int a = 0; a < 3; ++a) s.center[a] = r[VDETR_GT_CENTER + a], s.size[a] = r[VDETR_GT_SIZE + a];
    s.label = d.label_override >= 0 ? d.label_override : (int)r[VDETR_GT_LABEL];
    s.alabel = (int)r[VDETR_GT_ANGLE_CLS];
    s.ares_norm = r[VDETR_GT_ANGLE_RES] / (3.14159265358979323846f / (float)d.A);
  }
  __syncthreads();
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= d.P) return;
  const size_t row = (size_t)b * d.P + p;
  const BoxGeo pg = box_geo(d.corners + row * 24);
  const bool rot = d.rotated != nullptr && d.rotated[0] != 0.f;
  float pfx[4], pfz[4];
  footprint(d.corners + row * 24, pfx, pfz);
  float creg[3], sreg[3], pc[3], ps[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    creg[a] = d.center_reg[row * 3 + a], sreg[a] = d.size_reg[row * 3 + a];
    pc[a] = d.pre_center[row * 3 + a], ps[a] = d.pre_size[row * 3 + a] + 1e-5f;
  }
  const float obj = -d.objectness[row];
  const int nact = (int)d.nactual[b];
  const float* __restrict__ cls_row = d.cls + row * d.C;
  const float* __restrict__ al_row = d.angle_logits + row * d.A;
  const float* __restrict__ ar_row = d.angle_res_norm + row * d.A;
  float* __restrict__ cost_out = d.cost_t;
  float* __restrict__ giou_out = d.giou_t;
#pragma unroll 4
  for (int gi = 0; gi < ng; ++gi) {
    const GtDerived& s = sh[gi];
    float area = aligned_overlap(pg, s.geo);
    if (rot && area != 0.f && g0 + gi < nact) {  // box_util.py:571-589
      float gfx[4], gfz[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) gfx[i] = s.fx[i], gfz[i] = s.fz[i];
      area = 0.5f * clip_area2<0>(pfx, pfz, gfx, gfz, nullptr);
    }
    const float giou = giou_pair(pg, s.geo, area);
    float center = 0.f, size = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      center += fabsf(creg[a] - (s.center[a] - pc[a]) / ps[a]);
      size += fabsf(sreg[a] - logf((s.size[a] + 1e-5f) / ps[a]));
    }
    float cls;
    const float x = cls_row[s.label];
    if (d.cls_kind == VDETR_CLS_SIGMOID) {
      const float pr = 1.f / (1.f + expf(-x));
      const float neg = (0.75f * (pr * pr)) * (-logf((1.f - pr) + 1e-8f));
      const float pos = (0.25f * ((1.f - pr) * (1.f - pr))) * (-logf(pr + 1e-8f));
      cls = pos - neg;
    } else {
      cls = -x;
    }
    const float acls = -al_row[s.alabel];
    const float areg = huber1(ar_row[s.alabel] - s.ares_norm);
    float cost = d.w_cls * cls + d.w_objectness * obj;
    cost += d.w_center * center;
    cost += d.w_giou * (-giou);
    cost += d.w_size * size;
    cost += d.w_angle_cls * acls;
    cost += d.w_angle_reg * areg;
    const size_t o = ((size_t)b * d.G + g0 + gi) * d.P + p;
    cost_out[o] = cost;
    if (giou_out != nullptr) giou_out[o] = (g0 + gi) < nact ? giou : 0.f;
  }
}

// ------------------------------------------------------------------------------------------------ assignment
// fp64 -> u64 whose unsigned order is the numeric order
__device__ __forceinline__ unsigned long long f64_key(double v) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double key_f64(unsigned long long k) {
  const unsigned long long u = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
  return __longlong_as_double((long long)u);
}

// One workgroup per problem.  Lane t owns columns t, t+T, ... (<= CPT of them) with their dual variable, shortest-path
// cost and position in scipy's `remaining` array in registers.  T is the smallest of {64, 256, MAXT} whose T*CPT covers
// the problem's columns: ONE wave for <= 1024 columns (CPT = 16: no barrier and no LDS exchange inside a scan at all),
// 4 waves (one per SIMD) up to 4096, the whole workgroup beyond; the surplus waves exit before the first barrier.
// A scan step = one row of the cost matrix (coalesced, L2-resident after the warm-up pass), one lexicographic arg-min
// (DPP inside the wave; across waves one LDS slot per wave + one barrier), one LDS lookup of the row assigned to the
// winning column.
template <int CPT, int MAXT>
__global__ __launch_bounds__(MAXT) void lsa_kernel(vdetr_lsa_batch batch, int32_t* __restrict__ status, int nr_cap,
                                                   int nc_cap, int cols_per_lane, int cache_bytes) {
  extern __shared__ unsigned char smem[];
  int wg = blockIdx.x, k = 0;
  while (wg >= batch.p[k].B) wg -= batch.p[k++].B;
  const vdetr_lsa_problem pr = batch.p[k];
  const int b = wg, P = pr.P, G = pr.G;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = max(0, min((int)pr.nactual[b], G));
  // scipy solves a tall matrix transposed (rectangular_lsap.cpp: `transpose = nc < nr`); final_cost[b,:, :n] is P x n
  const bool transpose = n < P;
  const int nr = transpose ? n : P, nc = transpose ? P : n;
  // `cols_per_lane` columns per lane where the workgroup is wide enough, more (<= CPT) where it is not
  const int nthr = min(((nc + cols_per_lane - 1) / cols_per_lane + 63) & ~63, (int)blockDim.x);
  if (tid >= nthr) return;  // before any barrier: s_barrier only counts the waves still alive
  const int nwaves = nthr >> 6;
  int64_t* inds = pr.inds + (size_t)b * P;
  float* mask = pr.mask + (size_t)b * P;
  for (int p = tid; p < P; p += nthr) inds[p] = 0, mask[p] = 0.f;
  if (n == 0) return;
  const float* __restrict__ cost = pr.cost_t + (size_t)b * G * P;
  const int rs = transpose ? P : 1, cs = transpose ? 1 : P;  // max(P,G)^2 <= 2^26: int offsets

  double* u = reinterpret_cast<double*>(smem);                                   // [nr_cap]
  unsigned long long* wslot = reinterpret_cast<unsigned long long*>(u + nr_cap);  // [2][16] x {key, payload}
  int* col4row = reinterpret_cast<int*>(wslot + 64);                              // [nr_cap]
  int* row4col = col4row + nr_cap;                                               // [nc_cap]
  int* path = row4col + nc_cap;                                                  // [nc_cap]
  float* rcache = reinterpret_cast<float*>(path + nc_cap);                       // [cache_rows][nc] cost rows
  // Repeated ground truth (criterion.py:511-600) makes the box rows periodic: row r == row r % period, bit for bit (the
  // same records through the same arithmetic).  Only `period` distinct rows exist; as many as fit stay in LDS.
  int period = nr, cache_rows = 0;
  if (transpose && pr.row_repeat > 1 && n % pr.row_repeat == 0) {
    period = n / pr.row_repeat;
    cache_rows = min(period, cache_bytes / (nc * (int)sizeof(float)));
  }
  int* rbase = reinterpret_cast<int*>(rcache);  // [nr_cap] distinct-row index of every row (in front of the row cache)
  rcache += nr_cap;
  for (int i = tid; i < nr; i += nthr) u[i] = 0.0, col4row[i] = -1, rbase[i] = i % period;
  for (int j = tid; j < nc; j += nthr) row4col[j] = -1, path[j] = -1;
  const double kInf = __longlong_as_double(0x7FF0000000000000ll);
  double v[CPT], spc[CPT];
  int pos[CPT], pth[CPT];
  int off[CPT];        // element offset of this lane's columns inside a global row (0 for columns that do not exist)
  bool valid[CPT];     // the column exists
  bool bad = false;    // this thread saw a NaN / -inf entry
#pragma unroll
  for (int c = 0; c < CPT; ++c) {
    const int j = tid + c * nthr;
    v[c] = 0.0;
    valid[c] = j < nc;
    off[c] = valid[c] ? j * cs : 0;
  }
  // warm-up: every row is scanned several times; pull the distinct rows into LDS / this XCD's L2 once, and validate
  // them on the way
  for (int r = 0; r < period; ++r) {
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
      const float cij = cost[r * rs + off[c]];
      if (valid[c]) {
        if (!(cij == cij) || cij == -INFINITY) bad = true;
        if (r < cache_rows) rcache[r * nc + tid + c * nthr] = cij;
      }
    }
  }
  if (__syncthreads_or(bad)) {  // invalid numeric entries: scipy raises ValueError
    if (tid == 0 && status != nullptr) status[2 * blockIdx.x] = 1;
    return;
  }

  bool stop = false;  // uniform: no finite candidate left (infeasible matrix)
  int step = 0;
  for (int cur = 0; cur < nr; ++cur) {
    bool live[CPT], unassigned[CPT];  // live: not yet scanned (not in SC)
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
      const int j = tid + c * nthr;
      spc[c] = kInf;
      pos[c] = nc - 1 - j;
      live[c] = valid[c];
      unassigned[c] = valid[c] && row4col[valid[c] ? j : 0] == -1;
    }
    double min_val = 0.0;
    int i = cur, remaining = nc, sink = -1;
    float cij[CPT];
    auto fetch_row = [&](int base) {  // base: distinct-row index (uniform)
      if (base < cache_rows) {
        const float* __restrict__ src = rcache + base * nc + tid;
#pragma unroll
        for (int c = 0; c < CPT; ++c) cij[c] = src[valid[c] ? c * nthr : 0];
      } else {
        const float* __restrict__ src = cost + base * rs;
#pragma unroll
        for (int c = 0; c < CPT; ++c) cij[c] = src[off[c]];
      }
    };
    fetch_row(rbase[i]);
    while (true) {
      const double ui = u[i];
      // pass 1 (branch-free): relax this lane's columns with row i, keep the smallest shortest-path cost
      double bval = kInf;
#pragma unroll
      for (int c = 0; c < CPT; ++c) {
        const double r = ((min_val + (double)cij[c]) - ui) - v[c];
        const bool better = live[c] && r < spc[c];
        spc[c] = better ? r : spc[c];
        pth[c] = better ? i : pth[c];
        bval = fmin(bval, live[c] ? spc[c] : kInf);
      }
      const double wm = wave_allmin_f64(bval);
      // pass 2: only the lanes holding the wave minimum rank their tied columns (scipy's tie rule as a sort key:
      // an unassigned column beats an assigned one; among unassigned the LAST in `remaining` order, else the FIRST)
      unsigned long long bpay = 0ull;
      if (bval == wm) {
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
          if (live[c] && spc[c] == wm) {
            const unsigned sec = unassigned[c] ? (0x80000000u | (unsigned)pos[c]) : (0x7FFFFFFFu - (unsigned)pos[c]);
            const unsigned long long pay = ((unsigned long long)sec << 32) | (unsigned)(tid + c * nthr);
            bpay = pay > bpay ? pay : bpay;
          }
        }
      }
      const unsigned long long owners = __ballot(bpay != 0ull);
      unsigned long long wp;
      if (__popcll(owners) <= 1) {  // the common case: no tie inside the wave (0 owners: nothing live, wm = inf)
        const int src = owners ? __ffsll((long long)owners) - 1 : 0;
        wp = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(bpay >> 32), src) << 32) |
             (unsigned)__builtin_amdgcn_readlane((int)bpay, src);
      } else {
        wp = wave_allmax_u64(bpay);
      }
      ++step;
      double gm = wm;
      unsigned long long gp = wp;
      if (nwaves > 1) {  // one slot per wave, one barrier, independent reads
        const int buf = (step & 1) * 32;
        if (lane == 0) wslot[buf + 2 * wave] = (unsigned long long)__double_as_longlong(wm), wslot[buf + 2 * wave + 1] = wp;
        __syncthreads();
        gm = kInf, gp = 0ull;
        if (nwaves == 4) {
          double mm[4];
          unsigned long long pp[4];
#pragma unroll
          for (int w2 = 0; w2 < 4; ++w2)
            mm[w2] = __longlong_as_double((long long)wslot[buf + 2 * w2]), pp[w2] = wslot[buf + 2 * w2 + 1];
#pragma unroll
          for (int w2 = 0; w2 < 4; ++w2)
            if (mm[w2] < gm || (mm[w2] == gm && pp[w2] > gp)) gm = mm[w2], gp = pp[w2];
        } else {
          for (int w2 = 0; w2 < nwaves; ++w2) {
            const double m2 = __longlong_as_double((long long)wslot[buf + 2 * w2]);
            const unsigned long long p2 = wslot[buf + 2 * w2 + 1];
            if (m2 < gm || (m2 == gm && p2 > gp)) gm = m2, gp = p2;
          }
        }
      }
      min_val = gm;
      if (gm == kInf) {  // infeasible: scipy raises ValueError
        stop = true;
        break;
      }
      const int jstar = (int)(unsigned)gp;
      const unsigned sec = (unsigned)(gp >> 32);
      const bool is_free = sec >> 31;
      const int pstar = is_free ? (int)(sec & 0x7FFFFFFFu) : (int)(0x7FFFFFFFu - sec);
      --remaining;
#pragma unroll
      for (int c = 0; c < CPT; ++c) live[c] = live[c] && (tid + c * nthr != jstar);
      if (is_free) {
        sink = jstar;
        break;
      }
      i = row4col[jstar];
      fetch_row(rbase[i]);  // the next row's entries are requested before the bookkeeping below
#pragma unroll
      for (int c = 0; c < CPT; ++c) pos[c] = (live[c] && pos[c] == remaining) ? pstar : pos[c];  // remaining[index] = remaining[--num_remaining]
    }
    if (stop) break;
    // dual variables (rows of SR other than cur are the rows assigned to the scanned columns); the path entries of the
    // scanned columns (the only ones the augmentation can visit) go to LDS once per row
    if (tid == 0) u[cur] += min_val;
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
      if (valid[c] && !live[c]) {
        const int j = tid + c * nthr;
        const double dlt = min_val - spc[c];
        if (j != sink) u[row4col[j]] += dlt;
        v[c] -= dlt;
        path[j] = pth[c];
      }
    }
    __syncthreads();
    if (tid == 0) {  // augment along the path
      int j = sink;
      while (true) {
        const int r = path[j];
        row4col[j] = r;
        const int t = col4row[r];
        col4row[r] = j;
        j = t;
        if (r == cur) break;
      }
    }
    __syncthreads();
  }
  if (stop) {
    if (tid == 0 && status != nullptr) status[2 * blockIdx.x] = 1;
    return;
  }
  for (int r = tid; r < nr; r += nthr) {
    const int c = col4row[r];
    const int prop = transpose ? c : r, box = transpose ? r : c;
    inds[prop] = box;
    mask[prop] = 1.f;
  }
  if (tid == 0 && status != nullptr) status[2 * blockIdx.x + 1] = step;
}

// ------------------------------------------------------------------------------------------------ seed labels
__global__ __launch_bounds__(256) void point_labels_kernel(const float* __restrict__ xyz, const float* __restrict__ gt,
                                                           const int64_t* __restrict__ nactual, int N, int G, int C,
                                                           int64_t* __restrict__ labels, float* __restrict__ matched_count) {
  const int b = blockIdx.y, nidx = blockIdx.x * blockDim.x + threadIdx.x;
  if (nidx >= N) return;
  const float* p = xyz + ((size_t)b * N + nidx) * 3;
  const float x = p[0], y = p[1], z = p[2];
  const int n = max(0, min((int)nactual[b], G));
  float best = 100.f;  // the appended "no box" column (criterion.py:288)
  int pick = -1;
  for (int g = 0; g < n; ++g) {
    const float* r = gt + ((size_t)b * G + g) * F;
    const float dx = r[VDETR_GT_SIZE], dy = r[VDETR_GT_SIZE + 1], dz = r[VDETR_GT_SIZE + 2];
    const float zb = r[VDETR_GT_CENTER + 2] - dz / 2.f;  // bottom centre (criterion.py:277)
    const float zc = zb + dz / 2.f;
    if (fabsf(z - zc) > dz / 2.f) continue;
    const float sx = x - r[VDETR_GT_CENTER], sy = y - r[VDETR_GT_CENTER + 1];
    const float ang = -r[VDETR_GT_ANGLE];
    const float ca = cosf(ang), sa = sinf(ang);
    const float lx = sx * ca - sy * sa, ly = sx * sa + sy * ca;
    if (!(lx > -dx / 2.f && lx < dx / 2.f && ly > -dy / 2.f && ly < dy / 2.f)) continue;
    float vol = (dx * dy) * dz;
    if (vol == 0.f) vol = 1000.f;
    if (vol < best) best = vol, pick = g;
  }
  labels[(size_t)b * N + nidx] = pick >= 0 ? (int64_t)gt[((size_t)b * G + pick) * F + VDETR_GT_LABEL] : (int64_t)C;
  if (matched_count != nullptr) {  // rows that carry a real class: the normaliser of the weighted cross-entropy mean
    const unsigned long long hit = __ballot(pick >= 0);
    if ((threadIdx.x & 63) == 0 && hit) atomicAdd(matched_count, (float)__popcll(hit));
  }
}

// ------------------------------------------------------------------------------------------------ losses + gradients
__device__ __forceinline__ float wave_sum(float v) {
  v = row_allsum_f32(v);
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}

// weights of the two arguments' gradients in torch.minimum / maximum(a, b): the smaller (larger) takes it, a tie splits
__device__ __forceinline__ float take_min(float a, float b) { return a < b ? 1.f : (a == b ? 0.5f : 0.f); }
__device__ __forceinline__ float take_max(float a, float b) { return a > b ? 1.f : (a == b ? 0.5f : 0.f); }
__device__ __forceinline__ float sgn(float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); }

// d (1 - giou) * scale / d corners of the prediction; returns giou.  dc[24] is overwritten.
__device__ __forceinline__ float giou_grad(const float* c, const BoxGeo& p, const BoxGeo& g, float scale, float* dc, bool rot,
                                           const float* gt_corners) {
#pragma unroll
  for (int k = 0; k < 24; ++k) dc[k] = 0.f;
  const float top = fminf(p.c0y, g.c0y), bot = fmaxf(p.c4y, g.c4y);
  const float hraw = top - bot, height = fmaxf(hraw, 0.f);
  const float wxr = fminf(p.c0x, g.c0x) - fmaxf(p.c2x, g.c2x), wx = fmaxf(wxr, 0.f);
  const float wzr = fminf(p.c0z, g.c0z) - fmaxf(p.c2z, g.c2z), wz = fmaxf(wzr, 0.f);
  const float exr = fmaxf(p.mx[0], g.mx[0]) - fminf(p.mn[0], g.mn[0]);
  const float eyr = fminf(-p.mx[1], -g.mx[1]) - fmaxf(-p.mn[1], -g.mn[1]);
  const float ezr = fmaxf(p.mx[2], g.mx[2]) - fminf(p.mn[2], g.mn[2]);
  const float ex = fabsf(exr), ey = fabsf(eyr), ez = fabsf(ezr);
  const float enclosing = (ex * ey) * ez;
  const float e01 = edge_len(c, 0, 1), e12 = edge_len(c, 1, 2), e04 = edge_len(c, 0, 4);
  const float vraw = (e01 * e12) * e04;
  const float total = p.vol + g.vol;
  float area = wx * wz;
  float clip_grad[8];
  bool clipped = false;
  if (rot) {  // footprint overlap by polygon clipping where the axis-aligned test sees one (box_util.py:571-589)
    if (area != 0.f) {
      float pfx[4], pfz[4], gfx[4], gfz[4];
      footprint(c, pfx, pfz);
      footprint(gt_corners, gfx, gfz);
      area = 0.5f * clip_area2<8>(pfx, pfz, gfx, gfz, clip_grad);
      clipped = true;
    }
  }
  const float inter = area * height;
  const float uraw = total - inter, uni = fmaxf(uraw, 1e-8f);
  const float giou = inter / uni + (-(1.f - uni / enclosing));
  const bool good = enclosing > 2e-8f && total > 4e-8f;
  if (!good) return giou * 0.f;
  const float s = -scale;                                   // d loss / d giou
  float d_inter = s / uni;
  const float d_uni = s * (-inter / (uni * uni) + 1.f / enclosing);
  const float d_encl = s * (-uni / (enclosing * enclosing));
  const float d_uraw = uraw >= 1e-8f ? d_uni : 0.f;
  d_inter -= d_uraw;
  const float d_vol = vraw >= 1e-8f ? d_uraw : 0.f;
  // volume: three edges
  {
    const float de[3] = {d_vol * e12 * e04, d_vol * e01 * e04, d_vol * e01 * e12};
    const int ei[3] = {0, 1, 0}, ej[3] = {1, 2, 4};
    const float el[3] = {e01, e12, e04};
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const int i = ei[t], j = ej[t];
      const float dx = c[i * 3] - c[j * 3], dy = c[i * 3 + 1] - c[j * 3 + 1], dz = c[i * 3 + 2] - c[j * 3 + 2];
      const float sq = (dx * dx + dy * dy) + dz * dz;
      const float dsq = sq >= 1e-6f ? de[t] * (0.5f / el[t]) : 0.f;
      dc[i * 3] += 2.f * dx * dsq, dc[i * 3 + 1] += 2.f * dy * dsq, dc[i * 3 + 2] += 2.f * dz * dsq;
      dc[j * 3] -= 2.f * dx * dsq, dc[j * 3 + 1] -= 2.f * dy * dsq, dc[j * 3 + 2] -= 2.f * dz * dsq;
    }
  }
  // intersection
  const float d_area = d_inter * height, d_height = d_inter * area;
  const float d_hraw = hraw >= 0.f ? d_height : 0.f;
  dc[1] += d_hraw * take_min(p.c0y, g.c0y);        // corner 0, y
  dc[13] -= d_hraw * take_max(p.c4y, g.c4y);       // corner 4, y
  if (!rot) {
    const float d_wxr = wxr >= 0.f ? d_area * wz : 0.f;
    const float d_wzr = wzr >= 0.f ? d_area * wx : 0.f;
    dc[0] += d_wxr * take_min(p.c0x, g.c0x);         // corner 0, x
    dc[6] -= d_wxr * take_max(p.c2x, g.c2x);         // corner 2, x
    dc[2] += d_wzr * take_min(p.c0z, g.c0z);         // corner 0, z
    dc[8] -= d_wzr * take_max(p.c2z, g.c2z);         // corner 2, z
  } else if (clipped) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {  // footprint vertex i = corner 3 - i, (x, z)
      dc[(3 - i) * 3] += d_area * (0.5f * clip_grad[2 * i]);
      dc[(3 - i) * 3 + 2] += d_area * (0.5f * clip_grad[2 * i + 1]);
    }
  }
  // enclosing box: extremes over the 8 corners (the gradient of max/min(dim) goes to the first extreme index)
  float d_mx[3], d_mn[3];
  {
    const float dex = d_encl * ey * ez * sgn(exr), dey = d_encl * ex * ez * sgn(eyr), dez = d_encl * ex * ey * sgn(ezr);
    d_mx[0] = dex * take_max(p.mx[0], g.mx[0]);
    d_mn[0] = -dex * take_min(p.mn[0], g.mn[0]);
    d_mx[1] = -(dey * take_min(-p.mx[1], -g.mx[1]));
    d_mn[1] = dey * take_max(-p.mn[1], -g.mn[1]);
    d_mx[2] = dez * take_max(p.mx[2], g.mx[2]);
    d_mn[2] = -dez * take_min(p.mn[2], g.mn[2]);
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    bool mx_done = false, mn_done = false;
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2) {
      const float val = c[k2 * 3 + a];
      if (!mx_done && val == p.mx[a]) dc[k2 * 3 + a] += d_mx[a], mx_done = true;
      if (!mn_done && val == p.mn[a]) dc[k2 * 3 + a] += d_mn[a], mn_done = true;
    }
  }
  return giou;
}

constexpr int kLossRows = 256;  // rows (proposals / seed points) per workgroup

// Phase A, one thread per row: the row's label, the box terms of its matched pair with their gradients, the arg-max for
// the cardinality count.  Phase B, all threads over the chunk's rows x C logits (coalesced): focal loss + gradient.
__global__ __launch_bounds__(kLossRows) void set_loss_kernel(LossBatch batch) {
  __shared__ float red[kLossRows / 64][8];
  __shared__ int lab[kLossRows];
  int z = blockIdx.y, stage = 0;
  while (z >= batch.d[stage].B) z -= batch.d[stage++].B;
  const vdetr_setloss_desc d = batch.d[stage];
  const int b = z, tid = threadIdx.x;
  const int p0 = blockIdx.x * kLossRows;
  if (p0 >= d.P) return;
  const unsigned row_chunks = (unsigned)((d.P + kLossRows - 1) / kLossRows);  // this stage's workgroups per scene
  const int rows = min(kLossRows, d.P - p0);
  long total_boxes = 0;
  for (int i = 0; i < d.B; ++i) total_boxes += d.nactual[i];
  const bool gate = total_boxes > 0;
  const float inv_nb = 1.f / d.num_boxes[0];
  const bool boxes = d.center_reg != nullptr;
  // class loss: focal (sum / num_boxes) or the weighted cross-entropy MEAN of criterion.py:360-371 with the last class = "no
  // object" at weight w_no_object: its normaliser is sum of the row weights = M + (rows - M) * w_no_object, M = rows that
  // carry a real class = sum_b min(nactual_b, P) after a Hungarian match (or the caller's count for the seed-point loss)
  const bool ce = d.cls_kind == VDETR_CLS_SOFTMAX;
  float cls_scale = inv_nb;
  if (ce) {
    float m = 0.f;
    if (d.ce_rows_matched != nullptr) m = d.ce_rows_matched[0];
    else
      for (int i = 0; i < d.B; ++i) m += (float)min((long)d.nactual[i], (long)d.P);
    cls_scale = 1.f / (m + ((float)d.B * (float)d.P - m) * d.w_no_object);
  }
  float acc[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // sem_cls, angle_cls, angle_reg, center, size, giou, object count
  if (tid < rows) {
    const size_t row = (size_t)b * d.P + p0 + tid;
    bool matched = false;
    int gi = 0, label = ce ? d.C - 1 : d.C;
    if (d.inds != nullptr) {
      matched = d.mask[row] != 0.f;
      gi = (int)d.inds[row];
      if (matched) label = d.label_override >= 0 ? d.label_override : (int)d.gt[((size_t)b * d.G + gi) * F + VDETR_GT_LABEL];
    } else {
      label = (int)d.labels[row];
    }
    lab[tid] = label;
    {  // cardinality: arg-max over the logits, first maximum (criterion.py:262-268)
      const float* __restrict__ x = d.cls_logits + row * d.C;
      float best = x[0];
      int arg = 0;
      for (int c = 1; c < d.C; ++c) {
        const float xv = x[c];
        if (xv > best) best = xv, arg = c;
      }
      acc[6] = arg != d.C - 1 ? 1.f : 0.f;
      if (ce) {  // weighted cross entropy of this row + its gradient
        float se = 0.f;
        for (int c = 0; c < d.C; ++c) se += expf(x[c] - best);
        const float lse = best + logf(se);
        const float wy = label == d.C - 1 ? d.w_no_object : 1.f;
        acc[0] = wy * (lse - x[label]);
        const float gs = gate ? wy * (d.w_cls * cls_scale) : 0.f;
        float* __restrict__ dx = d.d_cls_logits + row * d.C;
        for (int c = 0; c < d.C; ++c) dx[c] = (expf(x[c] - lse) - (c == label ? 1.f : 0.f)) * gs;
      }
    }
    if (boxes) {
      float d_creg[3] = {0.f, 0.f, 0.f}, d_sreg[3] = {0.f, 0.f, 0.f};
      float dc[24];
#pragma unroll
      for (int k = 0; k < 24; ++k) dc[k] = 0.f;
      int alabel = 0;
      float g_areg = 0.f;
      const bool active = matched && gate;
      if (active) {
        const float* __restrict__ r = d.gt + ((size_t)b * d.G + gi) * F;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          const float ps = d.pre_size[row * 3 + a] + 1e-5f;
          const float ec = d.center_reg[row * 3 + a] - (r[VDETR_GT_CENTER + a] - d.pre_center[row * 3 + a]) / ps;
          acc[3] += fabsf(ec);
          d_creg[a] = sgn(ec) * (d.w_center * inv_nb);
          const float es = d.size_reg[row * 3 + a] - logf((r[VDETR_GT_SIZE + a] + 1e-5f) / ps);
          acc[4] += fabsf(es);
          d_sreg[a] = sgn(es) * (d.w_size * inv_nb);
        }
        float c[24];
#pragma unroll
        for (int k = 0; k < 24; ++k) c[k] = d.corners[row * 24 + k];
        const BoxGeo pg = box_geo(c), gg = box_geo(r + VDETR_GT_CORNERS);
        const bool rot = d.rotated != nullptr && d.rotated[0] != 0.f;
        const float giou = gi < (int)d.nactual[b] ? giou_grad(c, pg, gg, d.w_giou * inv_nb, dc, rot, r + VDETR_GT_CORNERS) : 0.f;
        acc[5] = 1.f - giou;
        alabel = (int)r[VDETR_GT_ANGLE_CLS];
        const float e = d.angle_res_norm[row * d.A + alabel] - r[VDETR_GT_ANGLE_RES] / (3.14159265358979323846f / (float)d.A);
        acc[2] = huber1(e);
        g_areg = fmaxf(-1.f, fminf(e, 1.f)) * (d.w_angle_reg * inv_nb);
        // cross entropy over the angle bins
        const float* __restrict__ al = d.angle_logits + row * d.A;
        float m = al[0];
        for (int a = 1; a < d.A; ++a) m = fmaxf(m, al[a]);
        float se = 0.f;
        for (int a = 0; a < d.A; ++a) se += expf(al[a] - m);
        const float lse = m + logf(se);
        acc[1] = lse - al[alabel];
        const float g_acls = d.w_angle_cls * inv_nb;
        for (int a = 0; a < d.A; ++a)
          d.d_angle_logits[row * d.A + a] = (expf(al[a] - lse) - (a == alabel ? 1.f : 0.f)) * g_acls;
      } else {
        for (int a = 0; a < d.A; ++a) d.d_angle_logits[row * d.A + a] = 0.f;
      }
      for (int a = 0; a < d.A; ++a) d.d_angle_res_norm[row * d.A + a] = (active && a == alabel) ? g_areg : 0.f;
#pragma unroll
      for (int a = 0; a < 3; ++a) d.d_center_reg[row * 3 + a] = d_creg[a], d.d_size_reg[row * 3 + a] = d_sreg[a];
#pragma unroll
      for (int k = 0; k < 24; ++k) d.d_corners[row * 24 + k] = dc[k];
    }
  }
  __syncthreads();
  if (!ce) {  // focal loss (criterion.py:77-98) over this chunk's logits
    const size_t base = ((size_t)b * d.P + p0) * d.C;
    const float* __restrict__ x = d.cls_logits + base;
    float* __restrict__ dx = d.d_cls_logits + base;
    const float gscale = gate ? d.w_cls * inv_nb : 0.f;
    const int nel = rows * d.C;
    for (int e = tid; e < nel; e += kLossRows) {
      const int r = e / d.C, c = e - r * d.C;
      const float xv = x[e];
      const bool t = c == lab[r];
      const float pr = 1.f / (1.f + expf(-xv));
      const float ce = fmaxf(xv, 0.f) - (t ? xv : 0.f) + log1pf(expf(-fabsf(xv)));
      const float one_m_pt = t ? 1.f - pr : pr;
      const float at = t ? d.focal_alpha : 1.f - d.focal_alpha;
      const float mod = one_m_pt * one_m_pt;
      acc[0] += at * (ce * mod);
      const float dpt = t ? pr * (1.f - pr) : -(pr * (1.f - pr));
      dx[e] = at * ((pr - (t ? 1.f : 0.f)) * mod - ce * (2.f * one_m_pt) * dpt) * gscale;
    }
  }
  // ---- workgroup sums -> one atomic per component
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int t = 0; t < 7; ++t) {
    const float sm = wave_sum(acc[t]);
    if (lane == 0) red[wave][t] = sm;
  }
  __syncthreads();
  if (tid == 0) {
    float sm[7];
    for (int t = 0; t < 7; ++t) {
      sm[t] = 0.f;
      for (int w2 = 0; w2 < kLossRows / 64; ++w2) sm[t] += red[w2][t];
    }
    const float w[6] = {d.w_cls, d.w_angle_cls, d.w_angle_reg, d.w_center, d.w_size, d.w_giou};
    float tot = 0.f;
    for (int t = 0; t < 6; ++t) {
      const float val = gate ? sm[t] * (t == 0 ? cls_scale : inv_nb) * w[t] : 0.f;
      if (w[t] > 0.f) tot += val;
      atomicAdd(d.losses + t, val);
    }
    atomicAdd(d.losses + 7, tot);
    // cardinality needs the scene's complete count: count and arrival ticket travel in ONE 64-bit atomic, so the
    // workgroup that draws the last ticket holds the total without any fence
    const unsigned long long add = (1ull << 32) | (unsigned long long)(unsigned)sm[6];
    const unsigned long long old = atomicAdd(d.card_ws + b, add);
    if ((unsigned)(old >> 32) == row_chunks - 1) {
      const float count = (float)((unsigned)old + (unsigned)sm[6]);
      atomicAdd(d.losses + 6, fabsf(count - (float)d.nactual[b]) / (float)d.B);
    }
  }
}

}  // namespace
}  // namespace vdetr

using namespace vdetr;

extern "C" int vdetr_gt_prepare_f32(const float* gt, int B, int G, int repeat, float* gt_rep, int64_t* nactual,
                                    int64_t* nactual_rep, float* sums, vdetr_stream_t stream) {
  VDETR_REQUIRE(gt && nactual && sums, "gt_prepare: null pointer");
  VDETR_REQUIRE(B >= 1 && G >= 1 && G <= 8192 && repeat >= 1, "gt_prepare: bad sizes B=%d G=%d repeat=%d", B, G, repeat);
  hipLaunchKernelGGL(gt_prepare_kernel, dim3(1), dim3(256), (G + 1) * sizeof(int), (hipStream_t)stream, gt, B, G, repeat, gt_rep,
                     nactual, nactual_rep, sums);
  return check_launch("gt_prepare");
}

static int check_match_desc(const vdetr_match_desc* d) {
  VDETR_REQUIRE(d->B >= 1 && d->P >= 1 && d->G >= 1 && d->C >= 1 && d->A >= 1, "match_cost: bad sizes");
  VDETR_REQUIRE(d->cls && d->objectness && d->center_reg && d->size_reg && d->pre_center && d->pre_size && d->corners &&
                    d->angle_logits && d->angle_res_norm && d->gt && d->nactual && d->cost_t,
                "match_cost: null pointer");
  VDETR_REQUIRE(d->label_override < d->C, "match_cost: label_override %d >= C %d", d->label_override, d->C);
  return VDETR_OK;
}

extern "C" int vdetr_match_cost_batch_f32(const vdetr_match_desc* descs, int n, vdetr_stream_t stream) {
  VDETR_REQUIRE(descs != nullptr && n >= 1, "match_cost: null descriptors");
  for (int s0 = 0; s0 < n; s0 += kCritBatch) {
    const int cnt = n - s0 < kCritBatch ? n - s0 : kCritBatch;
    MatchBatch batch{};
    int maxp = 1, maxg = 1, zs = 0;
    for (int k = 0; k < kCritBatch; ++k) {
      if (k >= cnt) {
        batch.d[k].B = 1 << 30;  // terminates the (stage, scene) search
        continue;
      }
      if (int e = check_match_desc(descs + s0 + k)) return e;
      batch.d[k] = descs[s0 + k];
      maxp = batch.d[k].P > maxp ? batch.d[k].P : maxp;
      maxg = batch.d[k].G > maxg ? batch.d[k].G : maxg;
      zs += batch.d[k].B;
    }
    VDETR_REQUIRE(zs <= 65535, "match_cost: %d (stage, scene) pairs > 65535", zs);
    const dim3 grid(ceil_div(maxp, 256), ceil_div(maxg, kMatchBoxes), zs);
    hipLaunchKernelGGL(match_cost_kernel, grid, dim3(256), 0, (hipStream_t)stream, batch);
    if (int e = check_launch("match_cost")) return e;
  }
  return VDETR_OK;
}

extern "C" int vdetr_match_cost_f32(const vdetr_match_desc* d, vdetr_stream_t stream) {
  VDETR_REQUIRE(d != nullptr, "match_cost: null descriptor");
  return vdetr_match_cost_batch_f32(d, 1, stream);
}

extern "C" int vdetr_lsa_f64(const vdetr_lsa_batch* batch, int32_t* status, vdetr_stream_t stream) {
  VDETR_REQUIRE(batch != nullptr && batch->nproblems >= 1 && batch->nproblems <= VDETR_LSA_MAX_PROBLEMS,
                "lsa: 1..%d problems per launch", VDETR_LSA_MAX_PROBLEMS);
  int nr_cap = 1, nc_cap = 1, wgs = 0;
  size_t cache_want = 0;
  vdetr_lsa_batch padded = *batch;
  for (int k = 0; k < VDETR_LSA_MAX_PROBLEMS; ++k) {
    if (k >= batch->nproblems) {
      padded.p[k] = vdetr_lsa_problem{};
      padded.p[k].B = 1 << 30;  // terminates the workgroup -> problem search
      continue;
    }
    const vdetr_lsa_problem& p = batch->p[k];
    VDETR_REQUIRE(p.cost_t && p.nactual && p.inds && p.mask, "lsa: null pointer in problem %d", k);
    VDETR_REQUIRE(p.B >= 1 && p.P >= 1 && p.G >= 1, "lsa: bad sizes in problem %d", k);
    const int lo = p.P < p.G ? p.P : p.G, hi = p.P < p.G ? p.G : p.P;
    VDETR_REQUIRE(hi <= 8192 && lo <= 2048, "lsa: problem %d is %d x %d; limits are 8192 columns, 2048 rows", k, p.P, p.G);
    nr_cap = lo > nr_cap ? lo : nr_cap;
    nc_cap = hi > nc_cap ? hi : nc_cap;
    VDETR_REQUIRE(p.row_repeat >= 0, "lsa: negative row_repeat in problem %d", k);
    if (p.row_repeat > 1) {  // distinct box rows of a repeated list: at most ceil(G / repeat), each P floats
      const size_t want = (size_t)((p.G + p.row_repeat - 1) / p.row_repeat) * p.P * sizeof(float);
      cache_want = want > cache_want ? want : cache_want;
    }
    wgs += p.B;
  }
  // up to 4096 columns: <= 4 columns per lane in a 1024-thread workgroup; beyond: <= 8
  const bool small = nc_cap <= 4096;
  const int cols_per_lane = VDETR_AB("VDETR_LSA_COLS", 4);
  const int cpl = cols_per_lane < 1 ? 1 : (cols_per_lane > 8 ? 8 : cols_per_lane);
  int threads = ((nc_cap + cpl - 1) / cpl + 63) & ~63;
  threads = threads > 1024 ? 1024 : threads;
  const size_t lds_base = (size_t)nr_cap * 8 + 64 * 8 + (size_t)nr_cap * 8 + (size_t)nc_cap * 8;  // + rbase
  const size_t lds_room = lds_base < 150 * 1024 ? 150 * 1024 - lds_base : 0;  // 160 KB per CU on gfx950
  const int cache_bytes = (int)(cache_want < lds_room ? cache_want : lds_room);
  const size_t lds = lds_base + cache_bytes;
  int rc;
#define VDETR_LSA_LAUNCH(CPT, MAXT)                                                                                  \
  rc = set_lds(lsa_kernel<CPT, MAXT>, lds, "lsa");                                                                   \
  if (rc != VDETR_OK) return rc;                                                                                     \
  hipLaunchKernelGGL((lsa_kernel<CPT, MAXT>), dim3(wgs), dim3(threads), lds, (hipStream_t)stream, padded, status, nr_cap, nc_cap, cpl, cache_bytes)
  if (small && cpl <= 4) {
    VDETR_LSA_LAUNCH(4, 1024);
  } else {
    VDETR_LSA_LAUNCH(8, 1024);
  }
#undef VDETR_LSA_LAUNCH
  return check_launch("lsa");
}

extern "C" int vdetr_point_labels_f32(const float* seed_xyz, const float* gt, const int64_t* nactual, int B, int N, int G,
                                      int C, int64_t* labels, float* matched_count, vdetr_stream_t stream) {
  VDETR_REQUIRE(seed_xyz && gt && nactual && labels, "point_labels: null pointer");
  VDETR_REQUIRE(B >= 1 && N >= 1 && G >= 1 && C >= 1, "point_labels: bad sizes");
  hipLaunchKernelGGL(point_labels_kernel, dim3(ceil_div(N, 256), B), dim3(256), 0, (hipStream_t)stream, seed_xyz, gt, nactual, N,
                     G, C, labels, matched_count);
  return check_launch("point_labels");
}

static int check_loss_desc(const vdetr_setloss_desc* d) {
  VDETR_REQUIRE(d->B >= 1 && d->P >= 1 && d->C >= 1, "set_loss: bad sizes");
  VDETR_REQUIRE(d->cls_logits && d->d_cls_logits && d->nactual && d->num_boxes && d->losses, "set_loss: null pointer");
  VDETR_REQUIRE((d->inds != nullptr && d->mask != nullptr && d->gt != nullptr) || d->labels != nullptr,
                "set_loss: need (inds, mask, gt) or labels");
  if (d->center_reg != nullptr) {
    VDETR_REQUIRE(d->size_reg && d->pre_center && d->pre_size && d->corners && d->angle_logits && d->angle_res_norm && d->gt &&
                      d->inds && d->mask && d->d_center_reg && d->d_size_reg && d->d_corners && d->d_angle_logits &&
                      d->d_angle_res_norm && d->A >= 1 && d->G >= 1,
                  "set_loss: box terms need all box pointers");
  }
  VDETR_REQUIRE(d->card_ws != nullptr, "set_loss: card_ws (B zero-initialised uint64) is required");
  return VDETR_OK;
}

extern "C" int vdetr_set_loss_batch_f32(const vdetr_setloss_desc* descs, int n, vdetr_stream_t stream) {
  VDETR_REQUIRE(descs != nullptr && n >= 1, "set_loss: null descriptors");
  for (int s0 = 0; s0 < n; s0 += kCritBatch) {
    const int cnt = n - s0 < kCritBatch ? n - s0 : kCritBatch;
    LossBatch batch{};
    int maxp = 1, ys = 0;
    for (int k = 0; k < kCritBatch; ++k) {
      if (k >= cnt) {
        batch.d[k].B = 1 << 30;
        continue;
      }
      if (int e = check_loss_desc(descs + s0 + k)) return e;
      batch.d[k] = descs[s0 + k];
      maxp = batch.d[k].P > maxp ? batch.d[k].P : maxp;
      ys += batch.d[k].B;
    }
    VDETR_REQUIRE(ys <= 65535, "set_loss: %d (stage, scene) pairs > 65535", ys);
    hipLaunchKernelGGL(set_loss_kernel, dim3(ceil_div(maxp, kLossRows), ys), dim3(kLossRows), 0, (hipStream_t)stream, batch);
    if (int e = check_launch("set_loss")) return e;
  }
  return VDETR_OK;
}

extern "C" int vdetr_set_loss_f32(const vdetr_setloss_desc* d, vdetr_stream_t stream) {
  VDETR_REQUIRE(d != nullptr, "set_loss: null descriptor");
  return vdetr_set_loss_batch_f32(d, 1, stream);
}
