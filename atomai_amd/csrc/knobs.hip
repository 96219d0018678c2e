// knobs.hip — resolution of the AMX_* switches (knobs.h) and their C-ABI view.
#include "amx_device.h"
#include "knobs.h"
#include <atomic>
#include <climits>
#include <cstdlib>
#include <cstring>
#include <mutex>

namespace {
struct KnobRow { const char* env; int AmxKnobs::*field; int def; };
// name, field, default.  DESIGN.md §4 "Switches" documents what each one is for and where it was measured.
const KnobRow kRows[] = {
    {"AMX_CONV_LATTICE", &AmxKnobs::conv_lattice, 1},
    {"AMX_CONV_NT", &AmxKnobs::conv_nt, 0},
    {"AMX_CONV_TH", &AmxKnobs::conv_th, 0},
    {"AMX_CONV_REM", &AmxKnobs::conv_rem, 1},
    {"AMX_CONV_XCD", &AmxKnobs::conv_xcd, 1},
    {"AMX_CONV_XPACK", &AmxKnobs::conv_xpack, 1},
    {"AMX_BWD_FUSE", &AmxKnobs::bwd_fuse, 1},
    {"AMX_BWD_SUMS", &AmxKnobs::bwd_sums, 1},
    {"AMX_CONV_WS", &AmxKnobs::conv_ws, 1},
    {"AMX_CONV_WS_DGRAD", &AmxKnobs::conv_ws_dgrad, 7},
    {"AMX_WGRAD_TH", &AmxKnobs::wgrad_th, 0},
    {"AMX_WGRAD_WGS", &AmxKnobs::wgrad_wgs, 0},
    {"AMX_WGRAD_WS", &AmxKnobs::wgrad_ws, 1},
    {"AMX_WGRAD_WS_MASK", &AmxKnobs::wgrad_ws_mask, 3},
    {"AMX_WGRAD_WS_WM4", &AmxKnobs::wgrad_ws_wm4, 128},
    {"AMX_GEMM_TILE", &AmxKnobs::gemm_tile, 0},
    {"AMX_RDEC_FWD_MT", &AmxKnobs::rdec_fwd_mt, 64},
    {"AMX_RDEC_BWD_MT", &AmxKnobs::rdec_bwd_mt, 64},
};
constexpr int kNumRows = (int)(sizeof(kRows) / sizeof(kRows[0]));

std::mutex g_mu;
std::atomic<const AmxKnobs*> g_knobs{nullptr};
std::atomic<long> g_generation{0};       // bumped by every (re-)resolution: host-side caches of amx_knob values key on it

const AmxKnobs* resolve_locked() {
    AmxKnobs* k = new AmxKnobs();          // immutable once published; a reload leaks the previous few dozen bytes on
    for (const KnobRow& r : kRows) {       // purpose (a launch on another thread may still be reading them)
        const char* e = getenv(r.env);
        k->*(r.field) = (e && *e) ? atoi(e) : r.def;
    }
    g_knobs.store(k, std::memory_order_release);
    g_generation.fetch_add(1, std::memory_order_release);
    return k;
}
}  // namespace

const AmxKnobs& amx_knobs() {
    const AmxKnobs* k = g_knobs.load(std::memory_order_acquire);
    if (!k) {
        std::lock_guard<std::mutex> lock(g_mu);
        k = g_knobs.load(std::memory_order_acquire);
        if (!k) k = resolve_locked();
    }
    return *k;
}

// Re-reads the environment (A/B scripts and plan-comparison tests; not used by the product path).
extern "C" int amx_knobs_reload(void) {
    std::lock_guard<std::mutex> lock(g_mu);
    resolve_locked();
    return 0;
}

// Number of times the table has been resolved (1 after the first launch; +1 per amx_knobs_reload, whoever called it): a
// host-side cache of amx_knob values is valid for one generation (atomai_amd/_lib.py kept a cache that a
// reload through another handle of this library left stale).
extern "C" long amx_knobs_generation(void) { amx_knobs(); return g_generation.load(std::memory_order_acquire); }

// Value of one resolved switch by its environment name; INT_MIN for a name the library does not know.
extern "C" int amx_knob(const char* name) {
    if (!name) return INT_MIN;
    const AmxKnobs& k = amx_knobs();
    for (const KnobRow& r : kRows)
        if (!strcmp(name, r.env)) return k.*(r.field);
    return INT_MIN;
}

// Number of switches / name of the i-th one (documentation and the ABI test enumerate the table through these).
extern "C" int amx_knob_count(void) { return kNumRows; }
extern "C" const char* amx_knob_name(int i) { return (i >= 0 && i < kNumRows) ? kRows[i].env : nullptr; }
