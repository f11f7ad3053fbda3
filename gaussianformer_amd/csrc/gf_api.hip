// gf_api.hip -- error reporting and ABI version of libgf_hip.so.
#include <stdarg.h>
#include <string.h>

#include <atomic>
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

namespace gf {
static std::atomic<int> g_options[kOptCount];
int option(int which) { return which >= 0 && which < kOptCount ? g_options[which].load(std::memory_order_relaxed) : 0; }
static const struct { const char *name; int which; } kOptionNames[] = {
    {"splat.mfma_tile_kernel", kOptSplatTileKernel}, {"daf.backward_tiles", kOptDafBackwardTiles}, {"subm.f32_mfma", kOptSubmF32Mfma}, {"subm.tile_gemm", kOptSubmTileGemm}, {"subm.bf16x3", kOptSubmBf16x3},
#if GF_DEV
    {"dev.splat_pair", kOptSplatPair}, {"dev.splat_solo", kOptSplatSolo}, {"dev.splat_solo_waves", kOptSplatSoloWaves},
    {"dev.splat_fused", kOptSplatFused}, {"dev.splat_fused_why", kOptSplatFusedWhy}, {"dev.units_bands", kOptUnitsBands},
    {"dev.prep_waves", kOptPrepWaves}, {"dev.bwd_no_lists", kOptBwdNoLists}, {"dev.bwd_no_big", kOptBwdNoBig},
    {"dev.daf_vec4", kOptDafVec4}, {"dev.daf_plain", kOptDafPlain},
#endif
};
static int option_index(const char *name)
{
    if (name)
        for (const auto &o : kOptionNames)
            if (strcmp(o.name, name) == 0) return o.which;
    return -1;
}
}  // namespace gf

extern "C" int gf_set_option(const char *name, int value)
{
    const int i = gf::option_index(name);
    if (i < 0) {
        gf::set_error("gf_set_option: unknown option '%s'", name ? name : "(null)");
        return GF_EINVAL;
    }
    gf::g_options[i].store(value, std::memory_order_relaxed);
    return GF_OK;
}

extern "C" int gf_get_option(const char *name, int *value)
{
    const int i = gf::option_index(name);
    if (i < 0 || !value) {
        gf::set_error("gf_get_option: unknown option '%s'", name ? name : "(null)");
        return GF_EINVAL;
    }
    *value = gf::option(i);
    return GF_OK;
}

extern "C" int gf_is_development_build(void) { return GF_DEV ? 1 : 0; }

extern "C" int gf_abi_version(void) { return GF_ABI_VERSION; }
extern "C" const char *gf_last_error(void) { return gf::g_err; }
