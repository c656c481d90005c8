// Error reporting + version of libcapmi.so.
#include <stdarg.h>
#include "common.h"

static thread_local char g_err[512] = "";

extern "C" void capmi_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* capmi_last_error(void) { return g_err; }
extern "C" int capmi_version(void) { return CAPMI_ABI_VERSION; }
