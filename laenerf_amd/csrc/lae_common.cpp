// lae_common.cpp -- library identification and per-thread error string.
#include <stdio.h>
#include <map>
#include <mutex>
#include <utility>
#include <vector>
#include "lae_common.h"

namespace lae {
static thread_local char g_err[256] = "";
void set_last_error(const char* what, hipError_t e) {
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
}

void set_last_error_str(const char* what) { snprintf(g_err, sizeof(g_err), "%s", what); }

// ---- library-owned scratch.  Keyed by (device, slot).  A buffer that is outgrown is RETIRED, not freed: kernels in
// flight and captured HIP graphs may still hold its address (a replayed graph writes through the pointer it was
// captured with), so it stays allocated until the caller says no such user is left (lae_free_workspaces).  Growth never
// synchronises; it is refused while `stream` is being captured (hipMalloc is illegal there, and the captured graph
// would bake in a buffer sized for this call only): warm up eagerly at the largest size first.
static std::mutex g_ws_mutex;
struct WsBuf { void* p = nullptr; size_t bytes = 0; };
static std::map<std::pair<int, int>, WsBuf> g_ws;
static std::vector<std::pair<int, WsBuf>> g_retired;
static uint64_t g_ws_epoch = 1;                     // bumped by free_workspaces(): addresses seen before may be handed out again

void* workspace(WsSlot slot, size_t bytes, hipStream_t stream) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
    std::lock_guard<std::mutex> lk(g_ws_mutex);
    WsBuf& b = g_ws[std::make_pair(dev, (int)slot)];
    if (b.p && b.bytes >= bytes) return b.p;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &st) != hipSuccess) { (void)hipGetLastError(); st = hipStreamCaptureStatusNone; }
    if (st != hipStreamCaptureStatusNone) {
        set_last_error_str("library workspace must grow inside a stream capture: run the same call once eagerly (warm-up) "
                           "at the largest size before capturing");
        return nullptr;
    }
    size_t want = bytes + bytes / 2;                 // headroom: sample counts drift from step to step
    if (want < (16u << 20)) want = 16u << 20;
    void* np = nullptr;
    hipError_t e = hipMalloc(&np, want);
    if (e != hipSuccess) { set_last_error("workspace hipMalloc", e); return nullptr; }
    if (b.p) g_retired.emplace_back(dev, b);          // still referenced by queued kernels / captured graphs
    b.p = np; b.bytes = want;
    return np;
}

uint64_t workspace_epoch() { std::lock_guard<std::mutex> lk(g_ws_mutex); return g_ws_epoch; }

size_t workspace_bytes(bool retired) {
    std::lock_guard<std::mutex> lk(g_ws_mutex);
    size_t n = 0;
    if (retired) for (const auto& r : g_retired) n += r.second.bytes;
    else for (const auto& kv : g_ws) n += kv.second.bytes;
    return n;
}

// frees live and retired buffers of every device.  The caller guarantees that no kernel is in flight on them and that
// no captured graph that used the library will be replayed again.
void free_workspaces() {
    std::lock_guard<std::mutex> lk(g_ws_mutex);
    int cur = 0;
    (void)hipGetDevice(&cur);
    auto drop = [&](int dev, void* p) {
        if (!p) return;
        (void)hipSetDevice(dev);
        (void)hipDeviceSynchronize();
        (void)hipFree(p);
    };
    for (auto& kv : g_ws) drop(kv.first.first, kv.second.p);
    for (auto& r : g_retired) drop(r.first, r.second.p);
    g_ws.clear();
    g_retired.clear();
    g_ws_epoch++;
    (void)hipSetDevice(cur);
}

int num_cus() {
    static std::mutex m;
    static std::map<int, int> per_dev;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
    std::lock_guard<std::mutex> lk(m);
    auto it = per_dev.find(dev);
    if (it != per_dev.end()) return it->second;
    int n = 0;
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, dev) == hipSuccess) n = p.multiProcessorCount; else (void)hipGetLastError();
    if (n <= 0) n = 256;
    per_dev[dev] = n;
    return n;
}
}  // namespace lae

extern "C" {
const char* lae_version(void) { return "laenerf-hip gfx950 " LAE_ABI_TAG; }
const char* lae_last_error(void) { return lae::g_err; }
int lae_free_workspaces(void) { lae::free_workspaces(); return LAE_OK; }
uint64_t lae_workspace_bytes(int retired) { return (uint64_t)lae::workspace_bytes(retired != 0); }
}
