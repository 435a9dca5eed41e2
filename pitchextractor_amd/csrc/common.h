// Shared helpers for the gfx950 kernels (wave = 64 lanes, 256-thread workgroups).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/pitchextractor_hip.h"

#define PE_CHECK_HIP(expr)                              \
  do {                                                  \
    hipError_t _e = (expr);                             \
    if (_e != hipSuccess) return (int)_e;               \
  } while (0)

// Launch-error check that never synchronises.
#define PE_LAUNCH_CHECK()                               \
  do {                                                  \
    hipError_t _e = hipGetLastError();                  \
    if (_e != hipSuccess) return (int)_e;               \
  } while (0)

static inline hipStream_t pe_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

static inline int pe_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define PE_WAVE 64
