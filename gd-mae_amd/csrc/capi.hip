// C-ABI plumbing: error reporting and build identification (see include/gdmae_hip.h).
#include "common.h"
#include <string.h>

static thread_local char g_err[512] = "";

extern "C" void gd_set_error(int code, const char* file, int line, const char* msg) {
  snprintf(g_err, sizeof(g_err), "gdmae_hip error %d at %s:%d: %s", code, file, line, msg ? msg : "");
}

extern "C" const char* gdmae_last_error(void) { return g_err; }

extern "C" int gdmae_abi_version(void) { return 1; }

extern "C" const char* gdmae_target_arch(void) { return "gfx950"; }
