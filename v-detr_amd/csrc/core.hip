// core.hip — error reporting + ABI version for libvdetr_hip.so.
#include "common.h"

#include <stdlib.h>

#include <mutex>
#include <unordered_map>

namespace vdetr {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int reserve_lds(const void* kernel, size_t bytes, const char* op) {
  static std::mutex mu;
  static std::unordered_map<const void*, size_t> granted;
  if (bytes <= 48 * 1024) return VDETR_OK;
  std::lock_guard<std::mutex> lock(mu);
  size_t& g = granted[kernel];
  if (bytes > g) {
    hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) {
      set_error("%s: cannot reserve %zu B of LDS: %s", op, bytes, hipGetErrorString(e));
      return VDETR_ERR_LAUNCH;
    }
    g = bytes;
  }
  return VDETR_OK;
}

// compute units of the current device (cached per device: no runtime query is left on the launch path after the first call)
int device_cu_count() {
  static int cached[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  int n = __atomic_load_n(&cached[dev], __ATOMIC_RELAXED);
  if (n == 0) {
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    __atomic_store_n(&cached[dev], n, __ATOMIC_RELAXED);
  }
  return n;
}
#ifdef VDETR_AB_SWITCHES
int ab_env(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}
#endif
}  // namespace vdetr

// One wave writes the device's constant-rate wall clock (100 MHz) to *slot: a timeline of a CAPTURED step without a tracer
// (rocprofv3 delays the cross-queue start of the side branch; tools/probes/step_timeline.py)
__global__ void vdetr_timestamp_kernel(unsigned long long* slot) {
  if (threadIdx.x == 0) *slot = wall_clock64();
}
extern "C" int vdetr_probe_timestamp(uint64_t* slot, vdetr_stream_t stream) {
  VDETR_REQUIRE(slot != nullptr, "probe_timestamp: null slot");
  hipLaunchKernelGGL(vdetr_timestamp_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, reinterpret_cast<unsigned long long*>(slot));
  return vdetr::check_launch("probe_timestamp");
}

extern "C" int vdetr_abi_version(void) { return 3; }
extern "C" int vdetr_ab_switches(void) {
#ifdef VDETR_AB_SWITCHES
  return 1;
#else
  return 0;
#endif
}
extern "C" const char* vdetr_last_error(void) { return vdetr::g_err; }
