// nms.hip — greedy 3-D non-maximum suppression of a scene's box predictions on the device (the
// evaluation's post-processing; reference utils/nms.py:78-162 called from utils/ap_calculator.py:165-220).
//
// The reference copies corners / scores to the host and runs a numpy loop per scene: pick the best remaining box, delete
// every remaining box of the same class whose IoU with it exceeds the threshold, repeat (~K numpy round trips of K-long
// vectors; 50-100 ms for 1024 boxes).  Here one workgroup per scene:
//   1. axis-aligned extents = min / max over the 8 corners (ap_calculator.py:168-214), volumes in fp64 as numpy has them
//      (np.zeros((K, 8)) is a float64 array filled with float32 values);
//   2. the K x K "i suppresses j" relation as a bit matrix in the caller's workspace, all threads, same fp64 expression
//      `inter / (area_i + area_j - inter) > thr` (or inter / area_j for old_type) times the class equality;
//   3. one wave walks the boxes in score order: a box still alive is kept and ORs its row into the dead set.
// Greedy suppression is inherently sequential in the kept boxes; the walk costs ~100 cycles per box.
#include "wave.h"

namespace vdetr {
namespace {

struct NmsParams {
  const float* corners;       // [B,K,8,3]
  const int32_t* cls;         // [B,K] or NULL (class-agnostic)
  const uint8_t* valid;       // [B,K] or NULL
  const int64_t* order;       // [B,K] ascending stable arg-sort of score
  uint8_t* keep;              // [B,K]
  // workspace
  float* ext;                 // [B,K,6] extents in rank order (rank r = r-th best box: order[K-1-r])
  int* rcls;                  // [B,K] class in rank order; < 0: invalid box (matches nothing)
  unsigned long long* rel;    // [B,K,W] "rank r suppresses rank 64w+t" bits, W = ceil(K/64)
  int K, W, old_type;
  double thr;
};

// 1. extents + classes in rank order
__global__ __launch_bounds__(256) void nms3d_prepare_kernel(NmsParams P) {
  const int b = blockIdx.y, r = blockIdx.x * 256 + threadIdx.x, K = P.K;
  if (r >= K) return;
  const int i = (int)P.order[(size_t)b * K + K - 1 - r];
  const float* c = P.corners + ((size_t)b * K + i) * 24;
  float lo[3] = {c[0], c[1], c[2]}, hi[3] = {c[0], c[1], c[2]};
  for (int k = 1; k < 8; ++k)
    for (int a = 0; a < 3; ++a) {
      lo[a] = fminf(lo[a], c[k * 3 + a]);
      hi[a] = fmaxf(hi[a], c[k * 3 + a]);
    }
  float* e = P.ext + ((size_t)b * K + r) * 6;
  for (int a = 0; a < 3; ++a) e[a] = lo[a], e[3 + a] = hi[a];
  const bool ok = P.valid == nullptr || P.valid[(size_t)b * K + i] != 0;
  P.rcls[(size_t)b * K + r] = ok ? (P.cls ? P.cls[(size_t)b * K + i] : 0) : -1 - r;
}

// 2. the K x K relation over the whole GPU: grid (K/256, 4W, B).  A thread owns rank r and builds a quarter (16 bits) of
//    word blockIdx.y/4 of its row; the 16 partner boxes are staged in LDS as float64 extents + volume (broadcast
//    reads), and only ranks worse than r are evaluated.  The division is skipped where no lane has an intersection.
constexpr int kNmsQuarter = 16;
__global__ __launch_bounds__(256) void nms3d_relation_kernel(NmsParams P) {
  __shared__ double sbox[kNmsQuarter][8];
  __shared__ int scls[kNmsQuarter];
  const int b = blockIdx.z, piece = blockIdx.y, K = P.K, W = P.W;
  const int r = blockIdx.x * 256 + threadIdx.x, s0 = piece * kNmsQuarter;
  unsigned short* out = reinterpret_cast<unsigned short*>(P.rel + (size_t)b * K * W);
  if (s0 + kNmsQuarter - 1 <= (int)(blockIdx.x * 256) || s0 >= K) {  // every pair of this block has s <= r
    if (r < K) out[(size_t)r * W * 4 + piece] = 0;
    return;
  }
  const float* __restrict__ ext = P.ext + (size_t)b * K * 6;
  const int* __restrict__ cls = P.rcls + (size_t)b * K;
  if (threadIdx.x < kNmsQuarter) {
    const int s = s0 + threadIdx.x;
    double c[6];
    for (int q = 0; q < 6; ++q) c[q] = s < K ? (double)ext[(size_t)s * 6 + q] : 0.0;
    for (int q = 0; q < 6; ++q) sbox[threadIdx.x][q] = c[q];
    sbox[threadIdx.x][6] = ((c[3] - c[0]) * (c[4] - c[1])) * (c[5] - c[2]);  // float64 volumes, as numpy computes them
    scls[threadIdx.x] = s < K ? cls[s] : -1;  // a negative class (invalid box, see prepare) matches nothing
  }
  __syncthreads();
  if (r >= K) return;
  unsigned bits = 0u;
  const int cr = cls[r];
  double a[6];
  for (int q = 0; q < 6; ++q) a[q] = (double)ext[(size_t)r * 6 + q];
  const double va = ((a[3] - a[0]) * (a[4] - a[1])) * (a[5] - a[2]);
  const bool always_divide = !(P.thr >= 0.0);
#pragma unroll
  for (int t = 0; t < kNmsQuarter; ++t) {
    const double* c = sbox[t];
    const bool pair = s0 + t > r && scls[t] == cr;
    const double l = fmax(0.0, fmin(a[3], c[3]) - fmax(a[0], c[0]));
    const double wd = fmax(0.0, fmin(a[4], c[4]) - fmax(a[1], c[1]));
    const double h = fmax(0.0, fmin(a[5], c[5]) - fmax(a[2], c[2]));
    const double inter = (l * wd) * h;
    if (pair && (inter > 0.0 || always_divide)) {  // inter == 0: the ratio is 0 or NaN, never above a threshold >= 0
      const double o = P.old_type ? inter / c[6] : inter / ((va + c[6]) - inter);
      if (o > P.thr) bits |= 1u << t;
    }
  }
  out[(size_t)r * W * 4 + piece] = (unsigned short)(cr >= 0 ? bits : 0u);
}

__device__ __forceinline__ unsigned long long readlane_u64(unsigned long long v, int lane) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, lane);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), lane);
  return ((unsigned long long)hi << 32) | lo;
}

// 3. the greedy walk, one workgroup per scene.  The relation is pulled into LDS when it fits (K <= 1024: 128 KB); then
//    ONE wave walks the ranks 64 at a time: lane l owns word l of the dead set (K <= 4096: <= 64 words); inside a block
//    of 64 ranks only the diagonal word decides, which is resolved with scalar instructions (lane t holds row t's
//    diagonal word); the rows of the block's survivors are then OR-ed into every lane's word with 64 unconditional,
//    masked loads (no dependent latency chain).  A box is kept iff its bit is still clear at the end: a row only has
//    bits of worse ranks, so nothing can kill a rank after its own turn.
constexpr int kNmsThreads = 1024;
template <bool IN_LDS>
__global__ __launch_bounds__(kNmsThreads) void nms3d_walk_kernel(NmsParams P) {
  extern __shared__ unsigned long long srel[];
  __shared__ unsigned long long dead0[64];
  const int b = blockIdx.x, tid = threadIdx.x, K = P.K, W = P.W;
  const unsigned long long* __restrict__ grel = P.rel + (size_t)b * K * W;
  const int* cls = P.rcls + (size_t)b * K;
  for (int r0 = 0; r0 < 64 * W; r0 += kNmsThreads) {  // invalid boxes and the tail of the last word start out dead
    const int r = r0 + tid;
    const unsigned long long m = __ballot(r >= K || cls[min(r, K - 1)] < 0);
    if ((tid & 63) == 0 && (r >> 6) < W) dead0[r >> 6] = m;
  }
  if (IN_LDS) {  // 8 loads in flight per thread: the copy is one memory round trip, not sixteen
    const int n = K * W;
    for (int e0 = 0; e0 < n; e0 += 8 * kNmsThreads) {
      unsigned long long v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int e = e0 + q * kNmsThreads + tid;
        v[q] = e < n ? grel[e] : 0ull;
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int e = e0 + q * kNmsThreads + tid;
        if (e < n) srel[e] = v[q];
      }
    }
  }
  __syncthreads();
  // one address space per instantiation: a pointer that may be either would compile to flat loads
  auto rel_at = [&](int e) -> unsigned long long { return IN_LDS ? srel[e] : grel[e]; };
  if (tid < 64) {
    // lane layout of the apply step: W2 = W rounded up to a power of two; lane = q * W2 + word, so one load instruction
    // fetches 64 / W2 rows of the block at once
    int W2 = 1;
    while (W2 < W) W2 <<= 1;
    const int rows_per_load = 64 / W2, q = tid / W2, wd = tid & (W2 - 1);
    const bool word_ok = wd < W;
    const int lw = min(wd, W - 1);
    unsigned long long mine = tid < W ? dead0[tid] : ~0ull;
    for (int w = 0; w < W; ++w) {
      const unsigned long long diag = rel_at(min(64 * w + tid, K - 1) * W + w);
      unsigned long long dw = readlane_u64(mine, w);
#pragma unroll
      for (int t = 0; t < 64; ++t) {
        const unsigned long long d = readlane_u64(diag, t);
        if (!((dw >> t) & 1ull)) dw |= d;
      }
      const unsigned long long alive = ~dw;  // survivors of this block (wave-uniform)
      const int last = K - 1 - 64 * w;       // rows beyond the last box are clamped; their alive bit is clear
      unsigned long long acc = 0ull;
#pragma unroll 4
      for (int i = 0; i < W2; ++i) {
        const int t = i * rows_per_load + q;
        const unsigned long long row = rel_at((64 * w + min(t, last)) * W + lw);
        acc |= ((alive >> t) & 1ull) && word_ok ? row : 0ull;
      }
      for (int off = W2; off < 64; off <<= 1) acc |= __shfl_xor(acc, off);  // OR over the q groups
      mine |= acc;                                                          // lane l < W: word l
    }
    dead0[tid] = mine;
  }
  __syncthreads();
  for (int r = tid; r < K; r += kNmsThreads)
    P.keep[(size_t)b * K + (int)P.order[(size_t)b * K + K - 1 - r]] = (uint8_t)(((~dead0[r >> 6]) >> (r & 63)) & 1ull);
}

// ---------------------------------------------------------------------------------------------------------------------
// points inside every predicted box (parse_predictions' remove_empty_box, utils/ap_calculator.py:78-93: mmcv
// points_in_boxes_all -> [B,N,K] flags -> sum over the points).  The flags are never materialised: a thread owns a box, a
// workgroup walks a slice of the scene's points through LDS (broadcast reads) and adds its count with one integer atomic.
// The test is mmcv's check_pt_in_box3d (bottom-centre z convention, strict x / y bounds in the box frame, yaw by -rz).
constexpr int kCountChunk = 1024;  // points staged per round
__global__ __launch_bounds__(256) void box_point_count_kernel(const float* __restrict__ points, const float* __restrict__ boxes,
                                                              int N, int K, int slice, int* __restrict__ counts) {
  __shared__ float4 pts[kCountChunk];
  const int b = blockIdx.z, k = blockIdx.y * 256 + threadIdx.x;
  const int n0 = blockIdx.x * slice, n1 = min(N, n0 + slice);
  float cx = 0.f, cy = 0.f, zc = 0.f, hx = -1.f, hy = -1.f, hz = -1.f, ca = 1.f, sa = 0.f;
  if (k < K) {
    const float* r = boxes + ((size_t)b * K + k) * 7;  // centre xyz, sizes, yaw
    const float zb = r[2] - r[5] / 2.f;                // the caller's bottom-centre shift (ap_calculator.py:80) ...
    zc = zb + r[5] / 2.f;                              // ... and mmcv's shift back to the centre
    cx = r[0]; cy = r[1];
    hx = r[3] / 2.f; hy = r[4] / 2.f; hz = r[5] / 2.f;
    ca = cosf(-r[6]); sa = sinf(-r[6]);
  }
  int cnt = 0;
  for (int c0 = n0; c0 < n1; c0 += kCountChunk) {
    const int m = min(kCountChunk, n1 - c0);
    __syncthreads();
    for (int i = threadIdx.x; i < m; i += 256) {
      const float* p = points + ((size_t)b * N + c0 + i) * 3;
      pts[i] = make_float4(p[0], p[1], p[2], 0.f);
    }
    __syncthreads();
#pragma unroll 4
    for (int i = 0; i < m; ++i) {
      const float4 p = pts[i];
      const float sx = p.x - cx, sy = p.y - cy;
      const float lx = sx * ca - sy * sa, ly = sx * sa + sy * ca;
      const bool in = !(fabsf(p.z - zc) > hz) && lx > -hx && lx < hx && ly > -hy && ly < hy;
      cnt += in ? 1 : 0;
    }
  }
  if (k < K && cnt) atomicAdd(counts + (size_t)b * K + k, cnt);
}

// ---------------------------------------------------------------------------------------------------------------------
// best ground-truth box of every detection (utils/eval_det.py:160-172 -> get_iou_obb -> utils/box_util.py:122-147
// box3d_iou): float64 on the float32 corners, the reference's Sutherland-Hodgman clip of the two x-z footprints
// (box_util.py:37-84: strict `inside`, the same intersection formula and operation order), heights from corners 0 / 4,
// volumes from three edge lengths.  The overlap area is the area of the convex hull of the clipped points, as the reference
// takes it from qhull (see the end of footprint_intersection).
struct P2 {
  double x, y;
};
__device__ __forceinline__ double box3d_vol64(const float* c) {
  auto len = [&](int i, int j) {
    const double dx = (double)c[i * 3] - (double)c[j * 3], dy = (double)c[i * 3 + 1] - (double)c[j * 3 + 1],
                 dz = (double)c[i * 3 + 2] - (double)c[j * 3 + 2];
    return sqrt((dx * dx + dy * dy) + dz * dz);
  };
  return (len(0, 1) * len(1, 2)) * len(0, 4);
}
__device__ double footprint_intersection(const float* c1, const float* c2) {
  P2 out[10], in[10];
  int n = 4;
  P2 clip[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {  // rect = corners 3, 2, 1, 0 (x, z): counter-clockwise (box_util.py:135-136)
    out[i] = P2{(double)c1[(3 - i) * 3], (double)c1[(3 - i) * 3 + 2]};
    clip[i] = P2{(double)c2[(3 - i) * 3], (double)c2[(3 - i) * 3 + 2]};
  }
  P2 cp1 = clip[3];
  for (int cv = 0; cv < 4; ++cv) {
    const P2 cp2 = clip[cv];
    const int m = n;
    for (int i = 0; i < m; ++i) in[i] = out[i];
    n = 0;
    auto inside = [&](const P2& p) { return (cp2.x - cp1.x) * (p.y - cp1.y) > (cp2.y - cp1.y) * (p.x - cp1.x); };
    auto cross_point = [&](const P2& s, const P2& e) {
      const double dc0 = cp1.x - cp2.x, dc1 = cp1.y - cp2.y, dp0 = s.x - e.x, dp1 = s.y - e.y;
      const double n1 = cp1.x * cp2.y - cp1.y * cp2.x, n2 = s.x * e.y - s.y * e.x;
      const double n3 = 1.0 / (dc0 * dp1 - dc1 * dp0);
      return P2{(n1 * dp0 - n2 * dc0) * n3, (n1 * dp1 - n2 * dc1) * n3};
    };
    P2 sv = in[m - 1];
    for (int i = 0; i < m; ++i) {
      const P2 e = in[i];
      if (inside(e)) {
        if (!inside(sv) && n < 10) out[n++] = cross_point(sv, e);
        if (n < 10) out[n++] = e;
      } else if (inside(sv)) {
        if (n < 10) out[n++] = cross_point(sv, e);
      }
      sv = e;
    }
    cp1 = cp2;
    if (n == 0) return 0.0;
  }
  // The reference hands the clipped points to qhull and takes the hull's area (box_util.py:92-105).  For a well-conditioned
  // clip that is the polygon itself; for nearly coincident rotated boxes the intersection formula divides by ~0 and the
  // "polygon" is garbage whose HULL the reference still measures (IoU > 1 happens) -- so the hull it is: monotone chain
  // over <= 10 points.  qhull raises on non-finite or degenerate input and the reference then counts no overlap.
  if (n < 3) return 0.0;
  for (int i = 0; i < n; ++i)
    if (!(fabs(out[i].x) < INFINITY) || !(fabs(out[i].y) < INFINITY)) return 0.0;
  for (int i = 1; i < n; ++i) {  // insertion sort by (x, y)
    const P2 v = out[i];
    int j = i - 1;
    while (j >= 0 && (out[j].x > v.x || (out[j].x == v.x && out[j].y > v.y))) {
      out[j + 1] = out[j];
      --j;
    }
    out[j + 1] = v;
  }
  auto turn = [](const P2& o, const P2& a, const P2& b) { return (a.x - o.x) * (b.y - o.y) - (a.y - o.y) * (b.x - o.x); };
  P2 chain[20];
  int h = 0;
  for (int i = 0; i < n; ++i) {  // lower chain
    while (h >= 2 && turn(chain[h - 2], chain[h - 1], out[i]) <= 0.0) --h;
    chain[h++] = out[i];
  }
  const int lower = h + 1;
  for (int i = n - 2; i >= 0; --i) {  // upper chain
    while (h >= lower && turn(chain[h - 2], chain[h - 1], out[i]) <= 0.0) --h;
    chain[h++] = out[i];
  }
  --h;  // the last point repeats the first
  if (h < 3) return 0.0;
  double acc = 0.0;
  for (int i = 0; i < h; ++i) {
    const P2 a = chain[i], b = chain[(i + 1) % h];
    acc += a.x * b.y - a.y * b.x;
  }
  return 0.5 * fabs(acc);
}
__global__ __launch_bounds__(128) void box3d_iou_max_kernel(const float* __restrict__ pred, const int* __restrict__ pred_img,
                                                            const int* __restrict__ pred_cls, int P, const float* __restrict__ gt,
                                                            const int* __restrict__ gt_cls, const int* __restrict__ img_gt_begin,
                                                            double* __restrict__ ovmax, int* __restrict__ jmax) {
  const int d = blockIdx.x * 128 + threadIdx.x;
  if (d >= P) return;
  float c1[24];
  for (int k = 0; k < 24; ++k) c1[k] = pred[(size_t)d * 24 + k];
  const int img = pred_img[d], cls = pred_cls[d];
  const double vol1 = box3d_vol64(c1);
  double best = -INFINITY;
  int arg = -1;
  for (int j = img_gt_begin[img]; j < img_gt_begin[img + 1]; ++j) {
    if (gt_cls[j] != cls) continue;
    float c2[24];
    for (int k = 0; k < 24; ++k) c2[k] = gt[(size_t)j * 24 + k];
    const double area = footprint_intersection(c1, c2);
    const double ymax = fmin((double)c1[1], (double)c2[1]), ymin = fmax((double)c1[13], (double)c2[13]);
    const double inter_vol = area * fmax(0.0, ymax - ymin);
    const double iou = inter_vol / ((vol1 + box3d_vol64(c2)) - inter_vol);
    if (iou > best) best = iou, arg = j;  // the first of equal maxima stays (eval_det.py:169-171)
  }
  ovmax[d] = best;
  jmax[d] = arg;
}

}  // namespace
}  // namespace vdetr

using namespace vdetr;

static size_t nms_align(size_t v) { return (v + 255) & ~(size_t)255; }

extern "C" size_t vdetr_nms3d_workspace_bytes(int B, int K) {
  if (B <= 0 || K <= 0) return 0;
  const size_t W = (K + 63) / 64;
  return nms_align((size_t)B * K * W * 8) + nms_align((size_t)B * K * 6 * 4) + nms_align((size_t)B * K * 4) + 256;
}

extern "C" int vdetr_nms3d_f32(const float* corners, const float* score, const int32_t* cls, const uint8_t* valid,
                               const int64_t* order, int B, int K, double iou_threshold, int old_type, uint8_t* keep,
                               void* workspace, size_t workspace_bytes, vdetr_stream_t stream) {
  VDETR_REQUIRE(B >= 0 && K >= 0, "nms3d: negative dimension");
  if (B == 0 || K == 0) return VDETR_OK;
  VDETR_REQUIRE(corners && score && order && keep, "nms3d: null pointer");
  VDETR_REQUIRE(K <= 4096 && B <= 65535, "nms3d: %d scenes x %d boxes: limits are 65535 x 4096", B, K);
  const size_t need = vdetr_nms3d_workspace_bytes(B, K);
  if (!workspace || workspace_bytes < need) {
    set_error("nms3d: workspace %zu B < required %zu B", workspace_bytes, need);
    return VDETR_ERR_WORKSPACE;
  }
  NmsParams P;
  P.corners = corners; P.cls = cls; P.valid = valid; P.order = order; P.keep = keep;
  P.K = K; P.W = (K + 63) / 64; P.old_type = old_type; P.thr = iou_threshold;
  uintptr_t base = ((uintptr_t)workspace + 255) & ~(uintptr_t)255;
  P.rel = reinterpret_cast<unsigned long long*>(base);
  base += nms_align((size_t)B * K * P.W * 8);
  P.ext = reinterpret_cast<float*>(base);
  base += nms_align((size_t)B * K * 6 * 4);
  P.rcls = reinterpret_cast<int*>(base);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(nms3d_prepare_kernel, dim3(ceil_div(K, 256), B), dim3(256), 0, st, P);
  if (int e = check_launch("nms3d_prepare")) return e;
  hipLaunchKernelGGL(nms3d_relation_kernel, dim3(ceil_div(K, 256), P.W * 4, B), dim3(256), 0, st, P);
  if (int e = check_launch("nms3d_relation")) return e;
  const size_t relb = (size_t)K * P.W * sizeof(unsigned long long);
  const int in_lds = relb <= 150 * 1024;
  const size_t lds = in_lds ? relb : 0;
  if (in_lds) {
    if (int rc = set_lds(nms3d_walk_kernel<true>, lds, "nms3d")) return rc;
    hipLaunchKernelGGL(nms3d_walk_kernel<true>, dim3(B), dim3(kNmsThreads), lds, st, P);
  } else {
    hipLaunchKernelGGL(nms3d_walk_kernel<false>, dim3(B), dim3(kNmsThreads), 0, st, P);
  }
  return check_launch("nms3d_walk");
}

extern "C" int vdetr_box_point_count_f32(const float* points, const float* boxes, int B, int N, int K, int32_t* counts,
                                         vdetr_stream_t stream) {
  VDETR_REQUIRE(B >= 0 && N >= 0 && K >= 0, "box_point_count: negative dimension");
  if (B == 0 || N == 0 || K == 0) return VDETR_OK;
  VDETR_REQUIRE(points && boxes && counts, "box_point_count: null pointer");
  VDETR_REQUIRE(B <= 65535, "box_point_count: %d scenes > 65535", B);
  // enough point slices to fill the chip (~1024 workgroups = 4 per CU)
  const int kb = ceil_div(K, 256);
  int splits = 1024 / (kb * B > 0 ? kb * B : 1);
  splits = splits < 1 ? 1 : splits;
  int slice = ceil_div(ceil_div(N, splits), 64) * 64;
  hipLaunchKernelGGL(box_point_count_kernel, dim3(ceil_div(N, slice), kb, B), dim3(256), 0, (hipStream_t)stream, points, boxes, N, K,
                     slice, counts);
  return check_launch("box_point_count");
}

extern "C" int vdetr_box3d_iou_max_f64(const float* pred_corners, const int32_t* pred_img, const int32_t* pred_cls, int P,
                                       const float* gt_corners, const int32_t* gt_cls, const int32_t* img_gt_begin,
                                       double* ovmax, int32_t* jmax, vdetr_stream_t stream) {
  VDETR_REQUIRE(P >= 0, "box3d_iou_max: negative count");
  if (P == 0) return VDETR_OK;
  VDETR_REQUIRE(pred_corners && pred_img && pred_cls && img_gt_begin && ovmax && jmax, "box3d_iou_max: null pointer");
  hipLaunchKernelGGL(box3d_iou_max_kernel, dim3(ceil_div(P, 128)), dim3(128), 0, (hipStream_t)stream, pred_corners, pred_img, pred_cls,
                     P, gt_corners, gt_cls, img_gt_begin, ovmax, jmax);
  return check_launch("box3d_iou_max");
}
