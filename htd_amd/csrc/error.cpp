#include <stdarg.h>
#include <stdio.h>

#include "../../include/htd_amd.h"

namespace htd {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace htd

extern "C" const char *htd_last_error(void) { return htd::g_err; }
extern "C" int htd_abi_version(void) { return HTD_ABI_VERSION; }
