// wgsort.h — ascending bitonic sort of 1024 * E 64-bit keys by one 1024-thread workgroup, keys in registers.
//
// Element i = tid * E + e of the network is slot e of thread tid.  A compare-exchange stage (k, j) pairs i with i ^ j:
//   j < E            both elements in one thread: registers only
//   E <= j < 64 E    partner in another lane of the same wave (lane ^ (j / E)): two 32-bit lane exchanges per key
//   j >= 64 E        partner in another wave: through LDS (one round of writes and reads between two barriers)
// For 4096 keys (E = 4) that is 10 LDS stages and 45 lane-exchange stages instead of the 78 barrier-separated LDS stages of the plain
// form (morton_order_kernel: 43 -> 9 us).  Keys must be distinct where the order of equal keys matters (pack an index into the low bits).
#pragma once
#include "common.h"

namespace vdetr {

__device__ __forceinline__ unsigned long long wg_shfl_xor_u64(unsigned long long v, int mask) {
  const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)v, mask, 64), hi = (unsigned)__shfl_xor((int)(unsigned)(v >> 32), mask, 64);
  return ((unsigned long long)hi << 32) | lo;
}

// one compare-exchange stage inside the thread (j < E: both indices compile-time constants after unrolling)
template <int E>
__device__ __forceinline__ void wg_sort_in_thread(unsigned long long (&key)[E], int tid, int k, int j) {
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int p = e ^ j;
    if (p > e && p < E) {
      const bool up = ((tid * E + e) & k) == 0;
      const unsigned long long a = key[e], b = key[p];
      const bool swap = (a > b) == up;
      key[e] = swap ? b : a;
      key[p] = swap ? a : b;
    }
  }
}

// lds: 1024 * E keys.  Every thread of the 1024 must call.
template <int E>
__device__ __forceinline__ void wg_bitonic_sort(unsigned long long (&key)[E], unsigned long long* lds, int tid) {
  constexpr int n = 1024 * E;
  const int lane = tid & 63;
#pragma unroll
  for (int k = 2; k <= E; k <<= 1)  // the phases that fit a thread
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) wg_sort_in_thread<E>(key, tid, k, j);
#pragma unroll 1
  for (int k = 2 * E; k <= n; k <<= 1) {
#pragma unroll 1
    for (int j = k >> 1; j >= E; j >>= 1) {
      if (j >= 64 * E) {  // ---- another wave: through LDS
        __syncthreads();  // (the previous LDS stage's reads are done)
#pragma unroll
        for (int e = 0; e < E; ++e) lds[tid * E + e] = key[e];
        __syncthreads();
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const int i = tid * E + e;
          const unsigned long long other = lds[i ^ j];
          const bool take_min = ((i & j) == 0) == ((i & k) == 0);
          key[e] = take_min ? (other < key[e] ? other : key[e]) : (other > key[e] ? other : key[e]);
        }
      } else {  // ---- another lane of this wave
        const int m = j / E;
        const bool low = (lane & m) == 0;
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const unsigned long long other = wg_shfl_xor_u64(key[e], m);
          const bool take_min = low == (((tid * E + e) & k) == 0);
          key[e] = take_min ? (other < key[e] ? other : key[e]) : (other > key[e] ? other : key[e]);
        }
      }
    }
#pragma unroll
    for (int j = E >> 1; j > 0; j >>= 1) wg_sort_in_thread<E>(key, tid, k, j);
  }
}

}  // namespace vdetr
