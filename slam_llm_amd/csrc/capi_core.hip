// Error plumbing + library identity for libslamhip.so (C ABI; see include/slam_hip.h).
// Reference convention being replaced: Python exceptions raised by the loaders
// (src/slam_llm/utils/model_utils.py:17-23); the ctypes binding turns a non-zero return code into
// RuntimeError(slam_last_error()).
#include "common.h"
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

void slam_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* slam_last_error() { return g_err; }

extern "C" int slam_abi_version() { return 1; }

extern "C" const char* slam_target_arch() { return "gfx950"; }
