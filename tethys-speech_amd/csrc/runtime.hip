// Host-side runtime glue of libtethys_mi.so: error slot and launch check.
#include "tmi_common.h"
#include <string.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

void tmi_set_error(const char* msg) {
  strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
  g_err[sizeof(g_err) - 1] = 0;
}

int tmi_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return TMI_ERR_LAUNCH;
  }
  return TMI_OK;
}

// Reproducible reductions (tmi_set_deterministic): kernels whose fp32 atomics make a result depend on arrival order take
// their fixed-order form while it is on (tmi_colsum: one workgroup per column group).  Process-wide, read at launch time.
static int g_deterministic = 0;
int tmi_deterministic() { return g_deterministic; }
extern "C" int tmi_set_deterministic(int on) {
  const int was = g_deterministic;
  g_deterministic = on ? 1 : 0;
  return was;
}

extern "C" int tmi_abi_version(void) { return 26; }
extern "C" const char* tmi_last_error(void) { return g_err; }
