// ABI bookkeeping for libcellseg_hip.so: version + thread-local last-error string.
#include <string.h>
#include "../../include/cellseg_hip.h"

static thread_local char g_err[512] = "";

extern "C" void cs_set_error_(const char* msg) {
    strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
    g_err[sizeof(g_err) - 1] = '\0';
}
extern "C" int cs_abi_version(void) { return 3; }   // 3: packed-operand convolutions (conv_v2), packed layouts in CsStageDesc
extern "C" const char* cs_last_error(void) { return g_err; }

// name of the conv-family kernel instantiation the calling thread launched last (set by the launchers)
static thread_local char g_variant[160] = "";
extern "C" void cs_set_variant_(const char* name) {
    strncpy(g_variant, name ? name : "", sizeof(g_variant) - 1);
    g_variant[sizeof(g_variant) - 1] = '\0';
}
extern "C" const char* cs_last_conv_variant(void) { return g_variant; }
