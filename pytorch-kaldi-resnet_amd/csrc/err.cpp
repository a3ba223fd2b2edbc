// Error string storage + version for libspkhip.
#include <stdarg.h>
#include <stdio.h>
#include "spk_common.h"

static thread_local char g_err[512] = "";

void spk_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* spk_last_error(void) { return g_err; }
extern "C" int spk_version(void) { return 300; }
// bit 0: built with SPK_EXPERIMENTAL - the measured, not-faster kernel forms are present (producer / consumer convolution and
// weight gradient, in-wave pipelined weight gradient, in-wave pipelined fused-BatchNorm-backward data gradient)
extern "C" int spk_build_flags(void) {
#ifdef SPK_EXPERIMENTAL
    return 1;
#else
    return 0;
#endif
}
