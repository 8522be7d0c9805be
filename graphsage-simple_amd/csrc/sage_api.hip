// libsage355: error reporting + version entry points (include/sage355.h).
#include <stdarg.h>
#include <string.h>

#include "sage_common.h"

static thread_local char g_err[512] = "";

void sage_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int sage_abi_version(void) { return SAGE_ABI_VERSION; }
extern "C" const char* sage_last_error(void) { return g_err; }
extern "C" const char* sage_build_arch(void) { return "gfx950"; }
