// Library-level entry points of libdistillclip_hip.so (include/dclip.h).
#include <stdarg.h>
#include <stdio.h>
#include "common.h"

static thread_local char g_err[512] = "";

void dclip_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int dclip_version(void) { return 1; }
extern "C" const char* dclip_arch(void) { return "gfx950"; }
extern "C" const char* dclip_last_error_string(void) { return g_err; }
