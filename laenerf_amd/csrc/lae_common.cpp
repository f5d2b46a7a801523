// lae_common.cpp -- library identification and per-thread error string.
#include <stdio.h>
#include <mutex>
#include "lae_common.h"

namespace lae {
static thread_local char g_err[256] = "";
void set_last_error(const char* what, hipError_t e) {
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
}

void set_last_error_str(const char* what) { snprintf(g_err, sizeof(g_err), "%s", what); }

static std::mutex g_ws_mutex;
static void* g_ws[WS_SLOTS] = {};
static size_t g_ws_bytes[WS_SLOTS] = {};

void* workspace(WsSlot slot, size_t bytes) {
    std::lock_guard<std::mutex> lk(g_ws_mutex);
    if (g_ws_bytes[slot] >= bytes && g_ws[slot]) return g_ws[slot];
    if (g_ws[slot]) { (void)hipDeviceSynchronize(); (void)hipFree(g_ws[slot]); g_ws[slot] = nullptr; g_ws_bytes[slot] = 0; }
    size_t want = bytes + bytes / 4;                 // headroom: sample counts drift from step to step
    if (want < (16u << 20)) want = 16u << 20;
    hipError_t e = hipMalloc(&g_ws[slot], want);
    if (e != hipSuccess) { set_last_error("workspace hipMalloc", e); g_ws[slot] = nullptr; return nullptr; }
    g_ws_bytes[slot] = want;
    return g_ws[slot];
}

void free_workspaces() {
    std::lock_guard<std::mutex> lk(g_ws_mutex);
    for (int i = 0; i < WS_SLOTS; i++)
        if (g_ws[i]) { (void)hipFree(g_ws[i]); g_ws[i] = nullptr; g_ws_bytes[i] = 0; }
}

int num_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0; hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) n = p.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}
}  // namespace lae

extern "C" {
const char* lae_version(void) { return "laenerf-hip gfx950 abi1"; }
const char* lae_last_error(void) { return lae::g_err; }
}
