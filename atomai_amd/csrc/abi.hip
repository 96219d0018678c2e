// abi.hip — error text of the C ABI (include/atomai_amd.h: amx_last_error).
//
// Every entry point returns an int (0 = ok, >0 = hipError_t, <0 = -(offending argument group)); the macros in
// amx_device.h additionally record WHICH function failed and why in a thread-local buffer, so that a caller that
// only sees "-3" can ask for "amx_conv2d_fwd: bad argument group 3".  No allocation, no synchronisation.
#include "amx_device.h"
#include <cstring>

static thread_local char amx_err_buf[256] = "";

extern "C" void amx_set_error(const char* fn, int code, const char* detail) {
    snprintf(amx_err_buf, sizeof(amx_err_buf), "%s: %s (code %d)", fn ? fn : "?", detail ? detail : "error", code);
}

extern "C" const char* amx_last_error(void) { return amx_err_buf; }

extern "C" int amx_clear_error(void) {
    amx_err_buf[0] = 0;
    return 0;
}
