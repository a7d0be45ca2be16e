// Library introspection entry points (no GPU work).
#include <hip/hip_runtime.h>
#include "../../include/m4depth_hip.h"

extern "C" int m4d_abi_version(void) { return M4D_ABI_VERSION; }

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
