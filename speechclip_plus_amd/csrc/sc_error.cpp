// Thread-local last-error string of the C ABI and the immutable device-property cache.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

#include "speechclip_hip.h"

static thread_local char g_err[512] = "";

void sc_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int sc_num_cus() {
    static int cus = 0;                    // every GPU of a node is the same part: one query serves all devices
    if (cus == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
            cus = n;
        else
            cus = 256;
    }
    return cus;
}

extern "C" const char* sc_last_error(void) { return g_err; }
extern "C" int sc_abi_version(void) { return 1; }
