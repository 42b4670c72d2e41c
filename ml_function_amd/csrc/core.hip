// Error reporting + version for libfil_hip.so.
#include "common.h"

namespace fil {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

}  // namespace fil

extern "C" int fil_version(void) { return 100; }  // 0.1.0
extern "C" const char* fil_last_error(void) { return fil::g_err; }
