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

extern "C" int tad_abi_version(void) { return 1; }
extern "C" const char* tad_last_error(void) { return g_err; }
