/*
 * pointnet2_oracle.c — CPU restatement of the reference's pointnet2 CUDA kernels.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under v-detr_amd/ may import, link or call this file; it is
 * the checker for tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 *
 * The reference kernels (third_party/pointnet2/_ext_src/src/*.cu) cannot be built here (CUDA only,
 * every op asserts "CPU not supported": sampling.cpp:36,62,84, ball_query.cpp:31,
 * group_points.cpp:34,59, interpolate.cpp:38,68,97) and the reference ships ONE test for them
 * (pointnet2_test.py:15-27, a CUDA-only gradcheck of three_interpolate).  PARITY UNPINNED for FPS,
 * gather, ball_query, group and three_nn: this file restates the kernels' arithmetic and ordering and
 * is pinned by hand-computable known-answer tests (tests/test_oracle_pointnet2.py), not by vectors
 * produced by the CUDA binary.
 *
 * Floating-point contraction.  nvcc's default (-fmad=true) contracts a*a + b*b + c*c.  The order is
 * pinned to the one LLVM's DAG combiner emits (checked on hipcc for this exact expression):
 *     t = b*b;  t = fma(a,a,t);  t = fma(c,c,t)
 * and is written with explicit fmaf() so that no compiler flag changes it (build with
 * -ffp-contract=off).  The HIP kernels use the same order (v-detr_amd/csrc/common.h sqdist3).
 * ORACLE_SQ3_ORDER selects the other orders a CUDA build could have produced — 1: the first product rounded
 * (t = a*a; t = fma(b,b,t); t = fma(c,c,t)), 2: no contraction ((a*a + b*b) + c*c, nvcc -fmad=false) — for
 * oracle/fps_order_exposure.py, which counts the sampled indices that depend on the choice;
 * common.h's VDETR_SQDIST_ORDER is the kernels' matching build constant.
 */
#ifndef ORACLE_SQ3_ORDER
#define ORACLE_SQ3_ORDER 0
#endif
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#if ORACLE_SQ3_ORDER == 0
static inline float sq3(float a, float b, float c) { return fmaf(c, c, fmaf(a, a, b * b)); }
#elif ORACLE_SQ3_ORDER == 1
static inline float sq3(float a, float b, float c) { return fmaf(c, c, fmaf(b, b, a * a)); }
#else
static inline float sq3(float a, float b, float c) { const float t = a * a + b * b; return t + c * c; }
#endif

/* cuda_utils.h:17-21  opt_n_threads */
static int ref_block(int work) {
  int p = 0;
  if (work < 1) return 1;
  while ((2L << p) <= (long)work) ++p;
  if (p > 9) p = 9;
  return 1 << p;
}

/* ------------------------------------------------------------------------------------------------
 * furthest_point_sampling — sampling_gpu.cu:73-176 + sampling.cpp:67-88.
 * Literal simulation: `bs` virtual threads, strided scan keeping the first strict maximum
 * (:98-113), then the tree reduction where slot t absorbs slot t+h only if strictly greater
 * (:62-68, :119-171).  temp is initialised to 1e10 (sampling.cpp:75-77).
 * ---------------------------------------------------------------------------------------------- */
void oracle_fps(const float* xyz, int b, int n, int m, int32_t* idx) {
  if (m <= 0 || n <= 0) return;
  const int bs = ref_block(n);
  float* temp = (float*)malloc(sizeof(float) * (size_t)n);
  float* dists = (float*)malloc(sizeof(float) * (size_t)bs);
  int* dists_i = (int*)malloc(sizeof(int) * (size_t)bs);
  for (int bi = 0; bi < b; ++bi) {
    const float* d = xyz + (size_t)bi * n * 3;
    int32_t* out = idx + (size_t)bi * m;
    for (int k = 0; k < n; ++k) temp[k] = 1e10f;
    int old = 0;
    out[0] = 0;
    for (int j = 1; j < m; ++j) {
      const float x1 = d[old * 3], y1 = d[old * 3 + 1], z1 = d[old * 3 + 2];
      for (int t = 0; t < bs; ++t) {
        int besti = 0;
        float best = -1.f;
        for (int k = t; k < n; k += bs) {
          const float x2 = d[k * 3], y2 = d[k * 3 + 1], z2 = d[k * 3 + 2];
          const float mag = sq3(x2, y2, z2);
          if ((double)mag <= 1e-3) continue; /* float vs double literal, :103-104 */
          const float dd = sq3(x2 - x1, y2 - y1, z2 - z1);
          const float d2 = fminf(dd, temp[k]);
          temp[k] = d2;
          besti = d2 > best ? k : besti;
          best = d2 > best ? d2 : best;
        }
        dists[t] = best;
        dists_i[t] = besti;
      }
      for (int h = bs / 2; h >= 1; h >>= 1)
        for (int t = 0; t < h; ++t) {
          const float v1 = dists[t], v2 = dists[t + h];
          const int i1 = dists_i[t], i2 = dists_i[t + h];
          dists[t] = v1 > v2 ? v1 : v2; /* max(v1, v2) */
          dists_i[t] = v2 > v1 ? i2 : i1;
        }
      old = dists_i[0];
      out[j] = old;
    }
  }
  free(temp);
  free(dists);
  free(dists_i);
}

/* Same result through the closed form the HIP kernel relies on: the winner of a round is the point
 * with the largest running distance; among equals, the smallest (bit-reversed (k mod bs), k / bs).
 * tests/ checks oracle_fps == oracle_fps_keyed on tie-heavy (voxel grid) clouds. */
static unsigned bitrev_u(unsigned v, int bits) {
  unsigned r = 0;
  for (int i = 0; i < bits; ++i) r |= ((v >> i) & 1u) << (bits - 1 - i);
  return r;
}
static int ilog2(int v) { int p = 0; while ((1 << (p + 1)) <= v) ++p; return p; }

void oracle_fps_keyed(const float* xyz, int b, int n, int m, int32_t* idx) {
  if (m <= 0 || n <= 0) return;
  const int bs = ref_block(n), lg = ilog2(bs);
  float* temp = (float*)malloc(sizeof(float) * (size_t)n);
  for (int bi = 0; bi < b; ++bi) {
    const float* d = xyz + (size_t)bi * n * 3;
    int32_t* out = idx + (size_t)bi * m;
    for (int k = 0; k < n; ++k) {
      const float mag = sq3(d[k * 3], d[k * 3 + 1], d[k * 3 + 2]);
      temp[k] = ((double)mag <= 1e-3) ? -INFINITY : 1e10f;
    }
    int old = 0;
    out[0] = 0;
    for (int j = 1; j < m; ++j) {
      const float x1 = d[old * 3], y1 = d[old * 3 + 1], z1 = d[old * 3 + 2];
      float best = -INFINITY;
      uint32_t bkey = 0xFFFFFFFFu;
      int besti = 0;
      for (int k = 0; k < n; ++k) {
        const float dd = sq3(d[k * 3] - x1, d[k * 3 + 1] - y1, d[k * 3 + 2] - z1);
        const float t = fminf(dd, temp[k]);
        temp[k] = t;
        if (!(t >= 0.f)) continue;
        const uint32_t key = (bitrev_u((unsigned)k % (unsigned)bs, lg) << 22) | ((unsigned)k / (unsigned)bs);
        if (t > best || (t == best && key < bkey)) { best = t; bkey = key; besti = k; }
      }
      old = besti;
      out[j] = old;
    }
  }
  free(temp);
}

/* CPU model of the HIP kernel's bucketed algorithm (v-detr_amd/csrc/fps.hip): points in `order`
 * (any permutation) are cut into `bucket` consecutive points; a bucket is only touched in a round when
 * the sampled point is closer to its bounding box than the bucket's current maximum.  Checks on the CPU
 * that the skip rule is exact (tests/ compares it with oracle_fps). Returns the number of point
 * distance evaluations so the work saving can be asserted too. */
long oracle_fps_bucketed(const float* xyz, int n, int m, const int32_t* order, int bucket, int32_t* idx) {
  if (m <= 0 || n <= 0) return 0;
  const int bs = ref_block(n), lg = ilog2(bs);
  const int nb = (n + bucket - 1) / bucket;
  float* temp = (float*)malloc(sizeof(float) * (size_t)n);
  float* lo = (float*)malloc(sizeof(float) * 3 * (size_t)nb);
  float* hi = (float*)malloc(sizeof(float) * 3 * (size_t)nb);
  float* bmax = (float*)malloc(sizeof(float) * (size_t)nb);
  uint32_t* bkey = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)nb);
  int* bidx = (int*)malloc(sizeof(int) * (size_t)nb);
  long evals = 0;
  for (int g = 0; g < nb; ++g) {
    for (int a = 0; a < 3; ++a) { lo[g * 3 + a] = INFINITY; hi[g * 3 + a] = -INFINITY; }
    bmax[g] = -INFINITY; bkey[g] = 0xFFFFFFFFu; bidx[g] = 0;
    for (int s = g * bucket; s < n && s < (g + 1) * bucket; ++s) {
      const int k = order[s];
      const float mag = sq3(xyz[k * 3], xyz[k * 3 + 1], xyz[k * 3 + 2]);
      if ((double)mag <= 1e-3) { temp[k] = -INFINITY; continue; }
      temp[k] = 1e10f;
      bmax[g] = 1e10f;
      for (int a = 0; a < 3; ++a) {
        lo[g * 3 + a] = fminf(lo[g * 3 + a], xyz[k * 3 + a]);
        hi[g * 3 + a] = fmaxf(hi[g * 3 + a], xyz[k * 3 + a]);
      }
    }
  }
  int old = 0;
  idx[0] = 0;
  for (int j = 1; j < m; ++j) {
    const float c[3] = {xyz[old * 3], xyz[old * 3 + 1], xyz[old * 3 + 2]};
    float best = -INFINITY;
    uint32_t gkey = 0xFFFFFFFFu;
    int besti = 0;
    for (int g = 0; g < nb; ++g) {
      float dd[3];
      for (int a = 0; a < 3; ++a) dd[a] = fmaxf(fmaxf(lo[g * 3 + a] - c[a], c[a] - hi[g * 3 + a]), 0.f);
      if (sq3(dd[0], dd[1], dd[2]) < bmax[g]) {
        float m2 = -INFINITY;
        uint32_t k2 = 0xFFFFFFFFu;
        int i2 = 0;
        for (int s = g * bucket; s < n && s < (g + 1) * bucket; ++s) {
          const int k = order[s];
          const float d = sq3(xyz[k * 3] - c[0], xyz[k * 3 + 1] - c[1], xyz[k * 3 + 2] - c[2]);
          ++evals;
          const float t = fminf(d, temp[k]);
          temp[k] = t;
          if (!(t >= 0.f)) continue;
          const uint32_t key = (bitrev_u((unsigned)k % (unsigned)bs, lg) << 22) | ((unsigned)k / (unsigned)bs);
          if (t > m2 || (t == m2 && key < k2)) { m2 = t; k2 = key; i2 = k; }
        }
        bmax[g] = m2; bkey[g] = k2; bidx[g] = i2;
      }
      if (bmax[g] >= 0.f && (bmax[g] > best || (bmax[g] == best && bkey[g] < gkey))) {
        best = bmax[g]; gkey = bkey[g]; besti = bidx[g];
      }
    }
    old = best >= 0.f ? besti : 0;
    idx[j] = old;
  }
  free(temp); free(lo); free(hi); free(bmax); free(bkey); free(bidx);
  return evals;
}

/* gather_points — sampling_gpu.cu:11-23 */
void oracle_gather_points(const float* points, const int32_t* idx, float* out, int b, int c, int n, int m) {
  for (int i = 0; i < b; ++i)
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < m; ++j)
        out[((size_t)i * c + l) * m + j] = points[((size_t)i * c + l) * n + idx[(size_t)i * m + j]];
}

/* gather_points_grad — sampling_gpu.cu:37-50 (atomicAdd; here in ascending j order), output zeroed
 * by the host (sampling.cpp:55-57) */
void oracle_gather_points_grad(const float* grad_out, const int32_t* idx, float* grad_points, int b, int c,
                               int n, int m) {
  memset(grad_points, 0, sizeof(float) * (size_t)b * c * n);
  for (int i = 0; i < b; ++i)
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < m; ++j)
        grad_points[((size_t)i * c + l) * n + idx[(size_t)i * m + j]] += grad_out[((size_t)i * c + l) * m + j];
}

/* ball_query — ball_query_gpu.cu:12-47; idx zero-filled by the host (ball_query.cpp:23-25) */
void oracle_ball_query(const float* new_xyz, const float* xyz, int32_t* idx, int b, int n, int m,
                       float radius, int nsample) {
  memset(idx, 0, sizeof(int32_t) * (size_t)b * m * nsample);
  const float radius2 = radius * radius;
  for (int i = 0; i < b; ++i)
    for (int j = 0; j < m; ++j) {
      const float* q = new_xyz + ((size_t)i * m + j) * 3;
      int32_t* row = idx + ((size_t)i * m + j) * nsample;
      for (int k = 0, cnt = 0; k < n && cnt < nsample; ++k) {
        const float* p = xyz + ((size_t)i * n + k) * 3;
        const float d2 = sq3(q[0] - p[0], q[1] - p[1], q[2] - p[2]);
        if (d2 < radius2) {
          if (cnt == 0)
            for (int l = 0; l < nsample; ++l) row[l] = k;
          row[cnt] = k;
          ++cnt;
        }
      }
    }
}

/* group_points — group_points_gpu.cu:11-31 */
void oracle_group_points(const float* points, const int32_t* idx, float* out, int b, int c, int n,
                         int npoints, int nsample) {
  for (int i = 0; i < b; ++i)
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < npoints; ++j)
        for (int k = 0; k < nsample; ++k)
          out[(((size_t)i * c + l) * npoints + j) * nsample + k] =
              points[((size_t)i * c + l) * n + idx[((size_t)i * npoints + j) * nsample + k]];
}

/* group_points_grad — group_points_gpu.cu:46-67 */
void oracle_group_points_grad(const float* grad_out, const int32_t* idx, float* grad_points, int b, int c,
                              int n, int npoints, int nsample) {
  memset(grad_points, 0, sizeof(float) * (size_t)b * c * n);
  for (int i = 0; i < b; ++i)
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < npoints; ++j)
        for (int k = 0; k < nsample; ++k)
          grad_points[((size_t)i * c + l) * n + idx[((size_t)i * npoints + j) * nsample + k]] +=
              grad_out[(((size_t)i * c + l) * npoints + j) * nsample + k];
}

/* three_nn — interpolate_gpu.cu:12-62: double sentinels 1e40, strict '<', stored to float;
 * outputs zero-filled by the host (interpolate.cpp:27-32) */
void oracle_three_nn(const float* unknown, const float* known, float* dist2, int32_t* idx, int b, int n,
                     int m) {
  for (int i = 0; i < b; ++i)
    for (int j = 0; j < n; ++j) {
      const float* u = unknown + ((size_t)i * n + j) * 3;
      double best1 = 1e40, best2 = 1e40, best3 = 1e40;
      int besti1 = 0, besti2 = 0, besti3 = 0;
      for (int k = 0; k < m; ++k) {
        const float* p = known + ((size_t)i * m + k) * 3;
        const float d = sq3(u[0] - p[0], u[1] - p[1], u[2] - p[2]);
        if (d < best1) {
          best3 = best2; besti3 = besti2; best2 = best1; besti2 = besti1; best1 = d; besti1 = k;
        } else if (d < best2) {
          best3 = best2; besti3 = besti2; best2 = d; besti2 = k;
        } else if (d < best3) {
          best3 = d; besti3 = k;
        }
      }
      float* dr = dist2 + ((size_t)i * n + j) * 3;
      int32_t* ir = idx + ((size_t)i * n + j) * 3;
      dr[0] = (float)best1; dr[1] = (float)best2; dr[2] = (float)best3;
      ir[0] = besti1; ir[1] = besti2; ir[2] = besti3;
    }
}

/* three_interpolate — interpolate_gpu.cu:75-104.  p1*w1 + p2*w2 + p3*w3 contracted in the pinned
 * order: t = p2*w2; t = fma(p1,w1,t); t = fma(p3,w3,t). */
void oracle_three_interpolate(const float* points, const int32_t* idx, const float* weight, float* out,
                              int b, int c, int m, int n) {
  for (int i = 0; i < b; ++i)
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < n; ++j) {
        const int32_t* ir = idx + ((size_t)i * n + j) * 3;
        const float* w = weight + ((size_t)i * n + j) * 3;
        const float* p = points + ((size_t)i * c + l) * m;
        float t = p[ir[1]] * w[1];
        t = fmaf(p[ir[0]], w[0], t);
        t = fmaf(p[ir[2]], w[2], t);
        out[((size_t)i * c + l) * n + j] = t;
      }
}

/* three_interpolate_grad — interpolate_gpu.cu:119-146 */
void oracle_three_interpolate_grad(const float* grad_out, const int32_t* idx, const float* weight,
                                   float* grad_points, int b, int c, int n, int m) {
  memset(grad_points, 0, sizeof(float) * (size_t)b * c * m);
  for (int i = 0; i < b; ++i)
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < n; ++j) {
        const int32_t* ir = idx + ((size_t)i * n + j) * 3;
        const float* w = weight + ((size_t)i * n + j) * 3;
        const float g = grad_out[((size_t)i * c + l) * n + j];
        float* gp = grad_points + ((size_t)i * c + l) * m;
        gp[ir[0]] += g * w[0];
        gp[ir[1]] += g * w[1];
        gp[ir[2]] += g * w[2];
      }
}
