// Library-level entry points of the C ABI.
#include "common.h"

extern "C" int pe_abi_version(void) { return 1; }

extern "C" int pe_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) return -(int)e;
  return n;
}
