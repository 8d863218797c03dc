// Error plumbing and ABI version of libtad_hip.so.
#include <stdarg.h>
#include <stdio.h>

#include "../../include/tad_hip.h"

static thread_local char g_err[512] = "";

void tad_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// Name of the device kernel the current C-ABI call launched (for profiling tools that aggregate by kernel).
static thread_local char g_kernel[160] = "";
static thread_local char g_kernel_out[160] = "";

void tad_note_kernel(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_kernel, sizeof(g_kernel), fmt, ap);
  va_end(ap);
}

extern "C" const char* tad_last_kernel(void) {
  snprintf(g_kernel_out, sizeof(g_kernel_out), "%s", g_kernel);
  g_kernel[0] = 0;
  return g_kernel_out;
}

extern "C" int tad_abi_version(void) { return 16; }
extern "C" const char* tad_last_error(void) { return g_err; }
