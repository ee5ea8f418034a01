// core.hip — error reporting + ABI version for libvdetr_hip.so.
#include "common.h"

namespace vdetr {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace vdetr

extern "C" int vdetr_abi_version(void) { return 1; }
extern "C" const char* vdetr_last_error(void) { return vdetr::g_err; }
