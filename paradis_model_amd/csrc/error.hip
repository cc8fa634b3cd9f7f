// Thread-local error string + ABI version for libparadis_hip.
#include <stdarg.h>
#include "common.h"

static thread_local char g_err[512] = "";

void paradis_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* paradis_last_error(void) { return g_err; }
extern "C" int paradis_abi_version(void) { return 3; }   // 3: flags argument of sl_advect_*, no debug setters
