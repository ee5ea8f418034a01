// bn_common.h — the dropout stream of the BatchNorm + ReLU + Dropout blocks (bn_act.hip, heads.hip): one counter-based hash keyed by
// (channel, element of the channel's B*N values), so that a fused launch and the one-launch-per-block form draw the same masks
// for the same (salt, generator state) and the backward regenerates them.
#pragma once
#include "attn_common.h"

namespace vdetr {

struct BnRng {
  unsigned seed_lo, seed_hi, off_lo, off_hi, thresh;
  float scale;
};
__device__ __forceinline__ BnRng bn_rng_of(float dropout_p, unsigned long long seed, unsigned long long offset, const uint64_t* rng_state) {
  BnRng r;
  unsigned long long s = seed, o = offset;
  if (rng_state) { s ^= rng_state[0]; o += rng_state[1]; }
  r.seed_lo = (unsigned)s; r.seed_hi = (unsigned)(s >> 32); r.off_lo = (unsigned)o; r.off_hi = (unsigned)(o >> 32);
  r.thresh = 0; r.scale = 1.f;
  if (dropout_p > 0.f) {
    int t = (int)((double)dropout_p * 65536.0 + 0.5);
    t = t < 1 ? 1 : (t > 65535 ? 65535 : t);
    r.thresh = (unsigned)t;
    r.scale = 65536.f / (float)(65536 - t);
  }
  return r;
}
__device__ __forceinline__ BnRng bn_rng(const vdetr_bnact_desc& d) { return bn_rng_of(d.dropout_p, d.seed, d.offset, d.rng_state); }
__device__ __forceinline__ unsigned bn_chankey(const BnRng& g, int c) {
  const unsigned x = fmix32(((unsigned)c * 0x9E3779B1u + g.off_lo) ^ g.seed_lo);
  return fmix32(x ^ (0x27D4EB2Fu + g.off_hi) ^ g.seed_hi);
}
// keep flag of element e (index within the channel's B*N elements)
__device__ __forceinline__ bool bn_keep(const BnRng& g, unsigned chankey, int e) {
  if (!g.thresh) return true;
  const unsigned x = fmix32(chankey ^ ((unsigned)(e >> 1) * 0x165667B1u));
  return ((e & 1) ? (x >> 16) : (x & 0xFFFFu)) >= g.thresh;
}
// the two 16-bit draws of elements 2 p and 2 p + 1
__device__ __forceinline__ unsigned bn_draw2(unsigned chankey, int pair) { return fmix32(chankey ^ ((unsigned)pair * 0x165667B1u)); }

}  // namespace vdetr
