// Error plumbing and identification entry points of libyv4_hip.so.
#include <string.h>

#include "yv4_common.h"

namespace yv4 {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace yv4

extern "C" int yv4_abi_version(void) { return YV4_ABI_VERSION; }
extern "C" const char* yv4_last_error(void) { return yv4::g_err; }
extern "C" const char* yv4_arch(void) { return "gfx950"; }
