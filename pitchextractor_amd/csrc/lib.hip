// Library-level entry points of the C ABI.
#include "common.h"

extern "C" int pe_abi_version(void) { return 3; }     // bumped with every round that changes a signature

extern "C" int pe_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) return -(int)e;
  return n;
}

// A non-blocking stream of the current device at its LOWEST priority (torch offers normal / high only): the model
// puts its weight-gradient kernels there so that they fill what the critical chain leaves free.  Created by THIS
// library's HIP runtime, the one every other call of the ABI uses; the caller owns the handle for the life of the
// process.
extern "C" int pe_stream_create_low_priority(void** stream_out) {
  if (!stream_out) return PE_E_ARG;
  int least = 0, greatest = 0;
  PE_CHECK_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
  hipStream_t st = nullptr;
  PE_CHECK_HIP(hipStreamCreateWithPriority(&st, hipStreamNonBlocking, least));
  *stream_out = reinterpret_cast<void*>(st);
  return PE_OK;
}
