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
char *g_zero_lo = nullptr, *g_zero_hi = nullptr;
}  // namespace gcn

GCN_EXPORT int gcn_zero_arena_register(void *base, long bytes) {
  GCN_REQUIRE(bytes >= 0 && (base || bytes == 0), "gcn_zero_arena_register: bad arena");
  gcn::g_zero_lo = (char *)base;
  gcn::g_zero_hi = (char *)base + (base ? bytes : 0);
  return GCN_OK;
}

GCN_EXPORT const char *gcn_last_error(void) { return gcn::g_err; }
GCN_EXPORT int gcn_version(void) { return 100; }
