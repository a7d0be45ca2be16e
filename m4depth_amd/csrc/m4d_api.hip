// Library introspection entry points (no GPU work).
#include <hip/hip_runtime.h>
#include "../../include/m4depth_hip.h"

#include <atomic>

extern "C" int m4d_abi_version(void) { return M4D_ABI_VERSION; }

static std::atomic<long long> g_launches{0};
extern "C" __attribute__((visibility("hidden"))) void m4d_count_launch(void) { g_launches.fetch_add(1, std::memory_order_relaxed); }
extern "C" long long m4d_launch_count(void) { return g_launches.load(std::memory_order_relaxed); }

#define M4D_STR2(x) #x
#define M4D_STR(x) M4D_STR2(x)
#if defined(M4D_EXPERIMENTS) && M4D_EXPERIMENTS
#define M4D_FLAVOUR " +experiments"
#else
#define M4D_FLAVOUR ""
#endif
extern "C" const char* m4d_build_info(void) {
  return "libm4depth_hip gfx950 (CDNA4) -ffp-contract=off abi=" M4D_STR(M4D_ABI_VERSION) M4D_FLAVOUR " built " __DATE__ " " __TIME__;
}
