// Library-level entry points: version and the thread-local error string.
#include <stdarg.h>

#include "common.h"

static thread_local char g_err[512] = "";

void rovit_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// 100: round 1.  200: round 2 changed rovit_vit_backward(_notify) (leading `images`) and rovit_joint_loss (float severity targets).
// 300: round 3 adds the fused MLP entry points and the prepared-weight stream they read (rovit_vit_prep_bytes grew).
// 400: round 4 -- rovit_vit_forward / _backward(_notify) take `mlp_path`; every rovit_set_* knob and the experiments that lost left the ABI.
// 410: round 4 -- rovit_joint_loss takes int64 severity labels (severity_is_int64); rovit_head_phase_*, rovit_sq_norm_clip, rovit_adamw_flat_multi added.
extern "C" int rovit_version(void) { return 410; }
extern "C" const char* rovit_last_error_string(void) { return g_err; }

#include <mutex>
#include <utility>
#include <vector>

bool rovit_set_max_lds(const void* fn, size_t bytes) {
  static std::mutex mu;
  static std::vector<std::pair<const void*, int>> done;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  std::lock_guard<std::mutex> lock(mu);
  for (const auto& d : done)
    if (d.first == fn && d.second == dev) return true;
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return false;
  done.emplace_back(fn, dev);
  return true;
}

#ifdef ROVIT_DEV
#include <stdlib.h>
#include <string.h>
int g_rovit_knob[ROVIT_KNOB_COUNT] = {0};
bool g_rovit_knob_set[ROVIT_KNOB_COUNT] = {false};
// ROVIT_DEV_KNOBS="id=value,id=value" (the developer library's ONE environment variable), read when the library is loaded
static int g_rovit_knob_env = [] {
  const char* e = getenv("ROVIT_DEV_KNOBS");
  while (e && *e) {
    int id = -1, v = 0, n = 0;
    if (sscanf(e, "%d=%d%n", &id, &v, &n) == 2 && id >= 0 && id < ROVIT_KNOB_COUNT) { g_rovit_knob[id] = v; g_rovit_knob_set[id] = true; }
    e = strchr(e, ',');
    if (e) ++e;
  }
  return 0;
}();
// developer library only (make dev): override / clear (value < 0 with clear != 0) a knob of common.h's RovitKnob list
extern "C" int rovit_dev_set_knob(int id, int value, int clear) {
  ROVIT_CHECK_ARG(id >= 0 && id < ROVIT_KNOB_COUNT, ROVIT_ERR_SHAPE, "dev knob %d out of range", id);
  g_rovit_knob[id] = value;
  g_rovit_knob_set[id] = !clear;
  return ROVIT_OK;
}
#endif
