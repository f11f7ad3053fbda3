// gf_api.hip -- error reporting and ABI version of libgf_hip.so.
#include <stdarg.h>

#include <vector>

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

namespace gf {
static std::vector<hipEvent_t> g_events;  // pairs: [2i] before, [2i+1] after
static int g_used = 0;
static int g_stride = 1, g_calls = 0;  // every g_stride-th launch is timed
bool profile_slot(hipEvent_t *before, hipEvent_t *after)
{
    if (g_events.empty()) return false;
    if (g_calls++ % g_stride != 0) return false;
    if ((size_t)(2 * g_used + 2) > g_events.size()) return false;
    *before = g_events[2 * g_used];
    *after = g_events[2 * g_used + 1];
    ++g_used;
    return true;
}
}  // namespace gf

extern "C" int gf_profile_enable(int max_records)
{
    for (hipEvent_t e : gf::g_events) (void)hipEventDestroy(e);
    gf::g_events.clear();
    gf::g_used = 0;
    gf::g_calls = 0;
    for (int i = 0; i < 2 * max_records; ++i) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) {
            gf::set_error("gf_profile_enable: hipEventCreate failed");
            return GF_ELAUNCH;
        }
        gf::g_events.push_back(e);
    }
    return GF_OK;
}

extern "C" int gf_profile_stride(int every)
{
    if (every < 1) {
        gf::set_error("gf_profile_stride: stride must be >= 1");
        return GF_EINVAL;
    }
    gf::g_stride = every;
    gf::g_calls = 0;
    return GF_OK;
}

extern "C" int gf_profile_read(float *ms_out, int capacity)
{
    int n = 0;
    for (int i = 0; i < gf::g_used && n < capacity; ++i) {
        float ms = 0.f;
        if (hipEventSynchronize(gf::g_events[2 * i + 1]) != hipSuccess) break;
        if (hipEventElapsedTime(&ms, gf::g_events[2 * i], gf::g_events[2 * i + 1]) != hipSuccess) break;
        ms_out[n++] = ms;
    }
    gf::g_used = 0;
    return n;
}

extern "C" int gf_abi_version(void) { return GF_ABI_VERSION; }
extern "C" const char *gf_last_error(void) { return gf::g_err; }
