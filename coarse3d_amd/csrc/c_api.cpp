// Error plumbing + version of the C ABI (include/coarse3d_hip.h).
#include <string.h>
#include "../../include/coarse3d_hip.h"

static thread_local char g_err[512] = "";

extern "C" void c3d_set_error(const char* msg) {
  strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
  g_err[sizeof(g_err) - 1] = 0;
}
extern "C" const char* c3d_last_error(void) { return g_err; }
extern "C" int c3d_version(void) { return 101; }
// sizeof of the descriptor structs, so that a binding can verify its mirror of the layouts
extern "C" int c3d_abi_sizes(int32_t* out5) {
  out5[0] = (int32_t)sizeof(c3d_src);
  out5[1] = (int32_t)sizeof(c3d_conv_desc);
  out5[2] = (int32_t)sizeof(c3d_wgrad_desc);
  out5[3] = (int32_t)sizeof(c3d_pack_entry);
  out5[4] = (int32_t)sizeof(c3d_wgrad_fold);
  return 0;
}
