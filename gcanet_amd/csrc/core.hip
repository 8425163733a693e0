// core.hip -- error reporting and version of libgcanet_hip.so
#include <stdarg.h>
#include <string.h>

#include "common.h"

namespace gcn {
static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace gcn

GCN_EXPORT const char *gcn_last_error(void) { return gcn::g_err; }
GCN_EXPORT int gcn_version(void) { return 100; }
