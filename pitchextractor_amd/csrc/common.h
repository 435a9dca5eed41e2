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

// Sum `n` strided doubles with one wave: lane l takes elements l, l+64, ...; result valid in lane 0.
__device__ __forceinline__ double pe_wave_strided_sum(const double* base, long stride, int n) {
  const int lane = threadIdx.x & 63;
  double s = 0.0;
  for (int z = lane; z < n; z += 64) s += base[(long)z * stride];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  return s;
}

// Split-K planning: choose the number of k-splits so that tiles * splits fills whole "waves" of
// resident workgroups (wave = CUs x workgroups per CU); a 1.5-wave grid idles a quarter of the chip.
static inline int pe_pick_splits(int tiles, long K, int min_k_per_split, int resident) {
  long max_s = K / min_k_per_split;
  if (max_s < 1) max_s = 1;
  if (max_s > 1024) max_s = 1024;
  int best = 1;
  double best_score = -1.0;
  for (int sp = 1; sp <= (int)max_s; ++sp) {
    const long wgs = (long)tiles * sp;
    const long waves = (wgs + resident - 1) / resident;
    const double eff = (double)wgs / (double)(waves * resident);
    // prefer full waves, then fewer splits (less slab traffic) once at least ~2 waves are queued
    const double score = eff - 0.0001 * sp + (wgs >= 2L * resident ? 0.05 : 0.0);
    if (score > best_score) { best_score = score; best = sp; }
  }
  return best;
}

// Ordered sum of `splits` slabs at float4 index i4 (element 4*i4 .. 4*i4+3), accumulated in DOUBLE in slab order
// (every fp32 slab value is exact in double and so is their sum up to 2^29 terms' worth of exponent spread: the
// result is the correctly rounded sum of the slabs for any realistic input, independent of how the reduction
// dimension was cut -- the second level of the weight-gradient sums adds no error of its own).  Loads are issued
// eight at a time so the reduce kernels run at memory speed instead of one dependent load latency per slab.
__device__ __forceinline__ float4 pe_ordered_slab_sum4(const float* __restrict__ ws, long slab_stride, int splits,
                                                       long i4) {
  const float4* p = reinterpret_cast<const float4*>(ws) + i4;
  const long st4 = slab_stride >> 2;
  double sx = 0.0, sy = 0.0, sz = 0.0, sw = 0.0;
  int z = 0;
  for (; z + 8 <= splits; z += 8) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = p[(long)(z + u) * st4];
#pragma unroll
    for (int u = 0; u < 8; ++u) { sx += (double)v[u].x; sy += (double)v[u].y; sz += (double)v[u].z; sw += (double)v[u].w; }
  }
  for (; z < splits; ++z) {
    const float4 v = p[(long)z * st4];
    sx += (double)v.x; sy += (double)v.y; sz += (double)v.z; sw += (double)v.w;
  }
  return make_float4((float)sx, (float)sy, (float)sz, (float)sw);
}

// Philox4x32-10 (dropout masks): 4 random words for counter `ctr`; element quad i of a dropout call uses
// ctr = offset + i, and keeps element k when (float)(word[k] >> 8) * 2^-24 >= p.
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t (&k)[2]) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
  const uint64_t p0 = (uint64_t)M0 * c[0], p1 = (uint64_t)M1 * c[2];
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k[0];
  const uint32_t n1 = (uint32_t)p1;
  const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k[1];
  const uint32_t n3 = (uint32_t)p0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
  k[0] += 0x9E3779B9u; k[1] += 0xBB67AE85u;
}

__device__ __forceinline__ void philox4(uint64_t seed, uint64_t ctr, uint32_t (&out)[4]) {
  uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u};
  uint32_t k[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
#pragma unroll
  for (int i = 0; i < 10; ++i) philox_round(c, k);
  out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
}

__device__ __forceinline__ bool pe_dropout_keep(uint32_t word, float p) {
  return ((float)(word >> 8) * (1.0f / 16777216.0f)) >= p;
}
