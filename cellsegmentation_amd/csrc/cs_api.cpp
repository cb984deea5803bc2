// ABI bookkeeping for libcellseg_hip.so: version + thread-local last-error string.
#include <string.h>
#include "../../include/cellseg_hip.h"

static thread_local char g_err[512] = "";

extern "C" void cs_set_error_(const char* msg) {
    strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
    g_err[sizeof(g_err) - 1] = '\0';
}
extern "C" int cs_abi_version(void) { return 2; }   // 2: workspaces on the BN reductions, deferred column sums, multi-layer staging, stage helpers
extern "C" const char* cs_last_error(void) { return g_err; }
