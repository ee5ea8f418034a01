// core.hip — error reporting + ABI version for libvdetr_hip.so.
#include "common.h"

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
}  // namespace vdetr

extern "C" int vdetr_abi_version(void) { return 2; }
extern "C" const char* vdetr_last_error(void) { return vdetr::g_err; }
