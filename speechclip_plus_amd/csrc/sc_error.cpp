// Thread-local last-error string of the C ABI and the immutable device-property cache.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
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

// host twin of the device-side dropout hash (csrc/sc_common.h): lets a caller or a test reconstruct a mask exactly
extern "C" uint32_t sc_hash32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352dU;
    x ^= x >> 15;
    x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}

extern "C" const char* sc_last_error(void) { return g_err; }
extern "C" int sc_abi_version(void) { return 4; }
// 1 = the diagnostics build (timing-only / stamped kernels + the LayerNorm-folded GEMMs), 0 = the product library
extern "C" int sc_is_diag_build(void) {
#ifdef SC_DIAG_BUILD
    return 1;
#else
    return 0;
#endif
}

// sizeof of the argument structs, for bindings that mirror them by hand (ctypes): a layout mismatch is caught before the first call
extern "C" int64_t sc_sizeof(int32_t what) {
    switch (what) {
        case 0: return (int64_t)sizeof(sc_gemm_args);
        case 1: return (int64_t)sizeof(sc_hubert_layer_args);
        case 2: return (int64_t)sizeof(sc_rt_gemm_args);
        case 3: return (int64_t)sizeof(sc_rt_ln_args);
        case 4: return (int64_t)sizeof(sc_rt_ln_bwd_args);
        case 5: return (int64_t)sizeof(sc_segments);
        default: return -1;
    }
}

// tuning switches for same-process A/B measurements (tools/): not part of the computation's contract, results never depend on them
static int g_options[8] = {1, 0, 0, 0, 0, 0, 0, 0};
int sc_option(int key) { return (key >= 0 && key < 8) ? g_options[key] : 0; }
extern "C" int sc_set_option(int32_t key, int32_t value) {
    if (key < 0 || key >= 8) {
        sc_set_error("sc_set_option: key %d out of range", key);
        return -1;
    }
    g_options[key] = value;
    return 0;
}
