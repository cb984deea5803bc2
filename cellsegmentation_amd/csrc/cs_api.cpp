// ABI bookkeeping for libcellseg_hip.so: version + thread-local last-error string.
#include <string.h>
#include <mutex>
#include <hip/hip_runtime_api.h>
#include "../../include/cellseg_hip.h"

static thread_local char g_err[512] = "";

extern "C" void cs_set_error_(const char* msg) {
    strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
    g_err[sizeof(g_err) - 1] = '\0';
}
extern "C" int cs_abi_version(void) { return 6; }   // 3: packed-operand convolutions (conv_v2), packed layouts in CsStageDesc; 4: round-3 entry points (cs_adam_step, cs_sample_sum, ...); 5: cs_set_igemm_path left the production library (A/B flavour only), CS_BN_BWD_* flags; 6: cs_adam_step_dev (device step counts: capturable)
extern "C" const char* cs_last_error(void) { return g_err; }

// name of the conv-family kernel instantiation the calling thread launched last (set by the launchers)
static thread_local char g_variant[160] = "";
extern "C" void cs_set_variant_(const char* name) {
    strncpy(g_variant, name ? name : "", sizeof(g_variant) - 1);
    g_variant[sizeof(g_variant) - 1] = '\0';
}
extern "C" const char* cs_last_conv_variant(void) { return g_variant; }

// (device, kernel) pairs whose dynamic-LDS limit has been raised (the attribute is per device; the table is shared by
// every launching thread)
extern "C" void cs_set_error_(const char* msg);
extern "C" int cs_allow_dynamic_lds_(const void* fn, size_t bytes, size_t limit) {
    if (bytes <= 65536) return 1;
    static std::mutex mu;
    static struct { const void* fn; int dev; } done[512];
    static int n_done = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { cs_set_error_("cannot query the current device"); return 0; }
    std::lock_guard<std::mutex> lock(mu);
    for (int i = 0; i < n_done; ++i) if (done[i].fn == fn && done[i].dev == dev) return 1;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)limit) != hipSuccess) {
        cs_set_error_("cannot raise the dynamic LDS limit of a kernel");
        return 0;
    }
    if (n_done < 512) { done[n_done].fn = fn; done[n_done].dev = dev; ++n_done; }
    return 1;
}

// compute units of the CURRENT device (cached per device; 256 -- a whole MI355X -- when the query fails, e.g. no GPU in a build
// container): the launch rules that ask "are all tiles resident at once?" must not assume an unpartitioned chip
extern "C" int cs_device_cus_(void) {
    static std::mutex mu;
    static int cus[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    std::lock_guard<std::mutex> lock(mu);
    if (cus[dev] == 0) {
        int n = 0;
        cus[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
    }
    return cus[dev];
}
