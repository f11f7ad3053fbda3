// gf_api.hip -- error reporting and ABI version of libgf_hip.so.
#include <stdarg.h>

#include "gf_common.hpp"

namespace gf {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace gf

extern "C" int gf_abi_version(void) { return GF_ABI_VERSION; }
extern "C" const char *gf_last_error(void) { return gf::g_err; }
