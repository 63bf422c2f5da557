// Error plumbing + version of the C ABI (include/coarse3d_hip.h).
#include <string.h>
#include "../../include/coarse3d_hip.h"

static thread_local char g_err[512] = "";

extern "C" void c3d_set_error(const char* msg) {
  strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
  g_err[sizeof(g_err) - 1] = 0;
}
extern "C" const char* c3d_last_error(void) { return g_err; }
extern "C" int c3d_version(void) { return 100; }
