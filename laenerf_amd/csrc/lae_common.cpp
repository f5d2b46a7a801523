// lae_common.cpp -- library identification and per-thread error string.
#include <stdio.h>
#include "lae_common.h"

namespace lae {
static thread_local char g_err[256] = "";
void set_last_error(const char* what, hipError_t e) {
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
}
}  // namespace lae

extern "C" {
const char* lae_version(void) { return "laenerf-hip gfx950 abi1"; }
const char* lae_last_error(void) { return lae::g_err; }
}
