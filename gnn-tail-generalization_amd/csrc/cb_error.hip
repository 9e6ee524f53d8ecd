#include "cb_common.h"

namespace cb {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace cb

extern "C" int cb_version(void) { return 1; }
extern "C" const char* cb_last_error(void) { return cb::g_err; }
