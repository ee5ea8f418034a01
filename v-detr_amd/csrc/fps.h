// fps.h — shared between the two furthest-point-sampling kernels (fps.hip: any n; fps_rows.hip: the fast path for
// clouds of up to 64 * 64 * 16 * kRowsSlots = 262,144 points).
#pragma once
#include "common.h"
#include "wave.h"

namespace vdetr {

constexpr int kFpsMaxScenes = 32;  // scenes of one variable-length launch

__device__ __forceinline__ unsigned fps_bitrev(unsigned v, int bits) { return bits ? (__brev(v) >> (32 - bits)) : 0u; }

// The reference's strided scan + tree reduction (sampling_gpu.cu:98-165) picks, among equal maxima, the point
// whose scanning thread (k mod bs) has the smallest BIT-REVERSED id, then the smallest k.  That order as one
// 32-bit key (smaller wins); decode_key is its inverse.
__device__ __forceinline__ unsigned fps_tie_key(unsigned k, unsigned ref_block, int ref_log2) {
  return (fps_bitrev(k % ref_block, ref_log2) << 22) | (k / ref_block);
}
__device__ __forceinline__ int fps_decode_key(unsigned key, unsigned ref_block, int ref_log2) {
  return (int)((key & 0x3FFFFFu) * ref_block + fps_bitrev(key >> 22, ref_log2));
}

// Candidate order: largest running distance t, then smallest tie key.  t >= 0 for every candidate, so its bit
// pattern is monotone as an unsigned; non-candidates (origin-skip, padding: t = -inf) rank as 0.
__device__ __forceinline__ unsigned fps_rank_of(float t) { return t >= 0.f ? __float_as_uint(t) + 1u : 0u; }
__device__ __forceinline__ float fps_t_of_rank(unsigned r) { return r ? __uint_as_float(r - 1u) : -INFINITY; }

// opt_n_threads(n): 2^floor(log2 n) clamped to [1,512] (cuda_utils.h:17-21)
inline int fps_ref_log2_of(int n) {
  int lg = 0;
  while ((2L << lg) <= (long)n) ++lg;
  return lg > 9 ? 9 : lg;
}

// ---- fps_rows.hip ------------------------------------------------------------------
constexpr int kRowsSlots = 4;  // buckets per owner lane, at most (a power of two)

struct RowsScene {
  const float* xyz;  // (n,3)
  int32_t* idx;      // (m)
  long ws_off;       // first workspace element (point) of this scene
  int n, cap_buckets, ref_block, ref_log2;  // cap_buckets: 64-slot segments this scene owns in the workspace
};

struct RowsParams {
  float4* pts;     // workspace: sorted (x,y,z,t), all scenes back to back
  uint32_t* keys;  // workspace: tie-order key of each sorted point
  int m;
  RowsScene scenes[kFpsMaxScenes];
};

struct RowsPlan {  // geometry chosen on the host for one launch
  int waves;       // 4, 8 or 16 waves per workgroup
  int bucket_pts;  // 64 points per bucket
};

// false: the cloud is too large for the row kernel (the caller uses fps.hip's kernel)
bool fps_rows_plan(int nmax, RowsPlan* plan);
// Buckets of a scene: leaves of the Z-curve's binary tree with <= 64 points take up to ~1.65 n / 64 segments of 64
// slots on uniform clouds; the workspace holds 2 n / 64 + 64 (capped by what the owner lanes can hold), and a cloud that
// needs more falls back to plain runs of 64 sorted points (n / 64 segments) inside the kernel.
inline long fps_rows_cap(int n, int waves = 16) {
  const long runs = ((long)n + 63) / 64, most = (long)kRowsSlots * 64 * waves;
  const long want = 2 * runs + 64;
  return want < most ? want : (runs > most ? runs : most);
}
inline long fps_rows_npad(int n, const RowsPlan& pl) { return fps_rows_cap(n, pl.waves) * pl.bucket_pts; }
// launches one workgroup per scene; P.scenes[0..b) filled by the caller except cap_buckets
int fps_rows_launch(RowsParams& P, int b, const RowsPlan& plan, hipStream_t stream);

}  // namespace vdetr
