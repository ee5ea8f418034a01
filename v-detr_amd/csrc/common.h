// common.h — shared host/device helpers for libvdetr_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <string.h>

#include "../../include/vdetr_hip.h"

namespace vdetr {

constexpr int kWave = 64;  // CDNA wavefront

void set_error(const char* fmt, ...);

// Records the launch status instead of exiting (reference: cuda_utils.h:32-41 exits the process).
inline int check_launch(const char* what) {
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) {
    set_error("%s: launch failed: %s", what, hipGetErrorString(err));
    return VDETR_ERR_LAUNCH;
  }
  return VDETR_OK;
}

#define VDETR_REQUIRE(cond, ...)      \
  do {                                \
    if (!(cond)) {                    \
      vdetr::set_error(__VA_ARGS__);  \
      return VDETR_ERR_ARG;           \
    }                                 \
  } while (0)

inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

int device_cu_count();  // core.hip

// A/B switches.  The shipped library reads no environment variable: VDETR_AB(name, default) IS the default.  A probe build
// (VDETR_EXTRA_HIPCC_FLAGS=-DVDETR_AB_SWITCHES, v-detr_amd/build.py) reads each switch once per process instead, which is how
// the ladders in DESIGN.md were measured.  vdetr_ab_switches() tells a caller which build it loaded.
#ifdef VDETR_AB_SWITCHES
int ab_env(const char* name, int dflt);  // core.hip
#define VDETR_AB(name, dflt) ([] { static const int v_ = vdetr::ab_env(name, dflt); return v_; }())
#else
#define VDETR_AB(name, dflt) (dflt)
#endif

// Raises a kernel's dynamic-LDS limit (needed above 64 KB) once per high-water mark, so that after the first
// launches no runtime API call is left on the launch path (safe inside hipGraph capture).  core.hip.
int reserve_lds(const void* kernel, size_t bytes, const char* op);
template <typename K>
inline int set_lds(K kernel, size_t bytes, const char* op) {
  return reserve_lds(reinterpret_cast<const void*>(kernel), bytes, op);
}

// Squared distance in the contraction order LLVM emits for
//   (x2-x1)*(x2-x1) + (y2-y1)*(y2-y1) + (z2-z1)*(z2-z1)
// (sampling_gpu.cu:106-107, ball_query_gpu.cu:34-35, interpolate_gpu.cu:36): t = dy*dy; t = fma(dx,dx,t);
// t = fma(dz,dz,t).  The oracle (oracle/pointnet2_oracle.c) pins the same order.
// VDETR_SQDIST_ORDER is the documented build constant for the day somebody holds this library against the CUDA binary and finds
// nvcc chose differently: 0 (default) as above; 1: t = dx*dx; t = fma(dy,dy,t); t = fma(dz,dz,t); 2: no contraction.  How many
// sampled indices depend on it: oracle/fps_order_exposure.py, profiles/r06_fps_order_exposure.txt.
#ifndef VDETR_SQDIST_ORDER
#define VDETR_SQDIST_ORDER 0
#endif
__device__ __forceinline__ float sqdist3(float dx, float dy, float dz) {
#if VDETR_SQDIST_ORDER == 0
  return __fmaf_rn(dz, dz, __fmaf_rn(dx, dx, __fmul_rn(dy, dy)));
#elif VDETR_SQDIST_ORDER == 1
  return __fmaf_rn(dz, dz, __fmaf_rn(dy, dy, __fmul_rn(dx, dx)));
#else
  return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
#endif
}

}  // namespace vdetr
