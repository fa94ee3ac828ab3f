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

// 2 (round 6): slam_lora_a_fwd gained `Rpad` mid-list, slam_logmel_workspace_bytes returns int64_t, a negative slam_attn_fwd scale means
// "Q arrives pre-scaled" (all round 5, shipped under version 1: ADVICE r5), + slam_reset_tuning, slam_label_rows.  slam_llm_amd/lib.py
// refuses a library whose version is not the one its signature table was written for.
extern "C" int slam_abi_version() { return 2; }

const unsigned long long* g_slam_drop_salt = nullptr;
extern "C" int slam_set_dropout_salt(const unsigned long long* device_word) {   // see common.h; null detaches
  g_slam_drop_salt = device_word;
  return 0;
}

void slam_gemm_reset_tuning_();
void slam_attn_reset_tuning_();
extern "C" int slam_reset_tuning() {
  slam_gemm_reset_tuning_();
  slam_attn_reset_tuning_();
  return 0;
}

extern "C" const char* slam_target_arch() { return "gfx950"; }
