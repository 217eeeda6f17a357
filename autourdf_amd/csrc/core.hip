// core.hip -- version, error string, device check.
#include <cstring>
#include "creg_common.h"

namespace creg {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace creg

extern "C" int creg_version(void) { return 600; }      // round 6: creg_train_plan_resume / creg_train_state, creg_train_plan_info_t.chain_probe_us, creg_icp_nn_counters (400: y_unchanged, named info fields)
extern "C" const char* creg_last_error(void) { return creg::g_err; }
extern "C" int creg_device_check(void) {
    int dev = 0;
    CREG_HIP(hipGetDevice(&dev));
    hipDeviceProp_t p;
    CREG_HIP(hipGetDeviceProperties(&p, dev));
    if (strncmp(p.gcnArchName, "gfx950", 6) != 0) {
        creg::set_error("libcreg is built for gfx950 only, current device is %s", p.gcnArchName);
        return CREG_EARCH;
    }
    return CREG_OK;
}
