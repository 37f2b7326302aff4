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

static std::atomic<int> g_deterministic{0};
bool deterministic() { return g_deterministic.load(std::memory_order_relaxed) != 0; }
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

// Deterministic mode (yv4_common.h, "deterministic mode"): every floating-point accumulation that meets in atomics runs
// on fixed-point integer words, so a training step is bit-reproducible run to run.  Process-wide; switch it between
// steps, not while a statistics buffer filled in the other mode is still waiting for its yv4_bn_finalize.
extern "C" int yv4_set_deterministic(int on) {
  yv4::g_deterministic.store(on ? 1 : 0, std::memory_order_relaxed);
  return YV4_OK;
}
extern "C" int yv4_get_deterministic(void) { return yv4::deterministic() ? 1 : 0; }
