// Error plumbing and identification entry points of libyv4_hip.so.
#include <string.h>

#include <atomic>

#include "yv4_common.h"

namespace yv4 {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

static std::atomic<int> g_nms_iou_form{YV4_NMS_IOU_DIV};
int nms_iou_form() { return g_nms_iou_form.load(std::memory_order_relaxed); }
}  // namespace yv4

extern "C" int yv4_abi_version(void) { return YV4_ABI_VERSION; }
extern "C" const char* yv4_last_error(void) { return yv4::g_err; }
extern "C" const char* yv4_arch(void) { return "gfx950"; }

extern "C" int yv4_nms_set_iou_form(int form) {
  YV4_REQUIRE(form == YV4_NMS_IOU_DIV || form == YV4_NMS_IOU_MUL, "nms_set_iou_form: form must be 0 (division) or 1 (product)");
  yv4::g_nms_iou_form.store(form, std::memory_order_relaxed);
  return YV4_OK;
}
extern "C" int yv4_nms_get_iou_form(void) { return yv4::nms_iou_form(); }
