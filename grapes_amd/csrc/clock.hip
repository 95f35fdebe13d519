// Kernel clock table: host-side bookkeeping of the in-kernel begin / end stamps (measurement infrastructure, see common.h).
// bench.py enables a table, captures a probe copy of the step (every instrumented launch reserves its stamp range once, at
// enqueue / capture time), replays it and reads the stamps back: per launch, duration = (latest end - earliest begin) / rate —
// the time the kernel's wavefronts occupied the chip inside the REPLAYED hipGraph, which HIP events cannot bracket.
#include "common.h"
#include <string.h>

namespace {
struct ClockEntry { char kernel[64]; int64_t offset; int32_t pairs; };
struct ClockState {
    unsigned long long* table = nullptr;
    int64_t words = 0, used = 0;
    int32_t n = 0;
    ClockEntry e[512];
} g_clk;
}  // namespace

unsigned long long* grapes_clock_reserve(const char* kernel, int grid, int waves_per_block) {
    if (!g_clk.table || g_clk.n >= 512 || grid <= 0 || waves_per_block <= 0) return nullptr;
    const int64_t pairs = (int64_t)grid * waves_per_block;
    if (g_clk.used + 2 * pairs > g_clk.words) return nullptr;
    ClockEntry& en = g_clk.e[g_clk.n++];
    strncpy(en.kernel, kernel, sizeof(en.kernel) - 1); en.kernel[sizeof(en.kernel) - 1] = 0;
    en.offset = g_clk.used; en.pairs = (int32_t)pairs;
    g_clk.used += 2 * pairs;
    return g_clk.table + en.offset;
}

bool grapes_clock_enabled() { return g_clk.table != nullptr; }

extern "C" int grapes_kernel_clock_enable(uint64_t* table, int64_t words) {
    if (table && words < 2) return GRAPES_EINVAL;
    g_clk.table = (unsigned long long*)table; g_clk.words = table ? words : 0; g_clk.used = 0; g_clk.n = 0;
    return 0;
}
extern "C" int32_t grapes_kernel_clock_launches(void) { return g_clk.n; }
extern "C" int grapes_kernel_clock_entry(int32_t i, char* kernel64, int64_t* offset_words, int32_t* pairs) {
    if (i < 0 || i >= g_clk.n || !kernel64 || !offset_words || !pairs) return GRAPES_EINVAL;
    memcpy(kernel64, g_clk.e[i].kernel, 64);
    *offset_words = g_clk.e[i].offset; *pairs = g_clk.e[i].pairs;
    return 0;
}
extern "C" int32_t grapes_kernel_clock_rate_khz(void) {
    int dev = 0, khz = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess) return 0;
    return khz;
}
