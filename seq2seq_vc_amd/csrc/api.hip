// Error plumbing + ABI version of libs2svc_hip.so.
#include <string.h>
#include "common.h"
#include "../../include/s2svc_hip.h"

static thread_local char g_err[512] = "";

extern "C" void s2svc_set_error(const char* msg) {
  size_t n = strlen(g_err);
  if (n && n + 2 < sizeof(g_err)) { g_err[n++] = ':'; g_err[n++] = ' '; g_err[n] = 0; }
  strncat(g_err, msg, sizeof(g_err) - n - 1);
}
extern "C" const char* s2svc_last_error(void) {
  static thread_local char out[512];
  strncpy(out, g_err, sizeof(out));
  g_err[0] = 0;
  return out;
}
extern "C" int s2svc_abi_version(void) { return 1; }
