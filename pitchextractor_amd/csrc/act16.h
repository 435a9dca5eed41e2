// Activation element access for the two storage formats of the conv stack's tensors: fp32 (parity path) and bf16
// (mixed precision: what the reference's autocast keeps in memory for conv / linear outputs, trainer.py:226-235 and
// README.md:36 -- half the HBM bytes of every BatchNorm / pooling / staging pass and half the saved-for-backward
// footprint).  Arithmetic is always fp32: ld4 widens (a 16-bit shift), st4 rounds to nearest even.
#pragma once
#include "common.h"

namespace pe {

typedef __bf16 act16_t;

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, const float4& v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float ld1(const float* p) { return *p; }
__device__ __forceinline__ void st1(float* p, float v) { *p = v; }

__device__ __forceinline__ float4 ld4(const act16_t* p) {            // 4 consecutive bf16 (8-byte aligned)
  const uint2 u = *reinterpret_cast<const uint2*>(p);
  return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                     __uint_as_float(u.y & 0xffff0000u));
}
__device__ __forceinline__ unsigned pack_bf16_rne(float a, float b) {
  typedef __bf16 b2 __attribute__((ext_vector_type(2)));
  typedef float f2 __attribute__((ext_vector_type(2)));
  const f2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, b2));
}
__device__ __forceinline__ void st4(act16_t* p, const float4& v) {
  *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf16_rne(v.x, v.y), pack_bf16_rne(v.z, v.w));
}
__device__ __forceinline__ float ld1(const act16_t* p) {
  return __uint_as_float((unsigned)*reinterpret_cast<const unsigned short*>(p) << 16);
}
__device__ __forceinline__ void st1(act16_t* p, float v) { *p = (act16_t)v; }

// A quad as it comes out of memory: kernels keep the loaded bits in registers until they are staged (converting right
// after the load would put the conversion -- and with it the wait for the load -- in front of the MFMA phase the load
// is meant to overlap: the bf16-tensor weight-gradient kernel ran 2x slower that way).
template <class T> struct RawQuad;
template <> struct RawQuad<float> { typedef float4 type; };
template <> struct RawQuad<act16_t> { typedef uint2 type; };
__device__ __forceinline__ float4 ldraw(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ uint2 ldraw(const act16_t* p) { return *reinterpret_cast<const uint2*>(p); }
__device__ __forceinline__ float4 widen(const float4& v) { return v; }
__device__ __forceinline__ float4 widen(const uint2& u) {
  return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                     __uint_as_float(u.y & 0xffff0000u));
}
// the same quad through a buffer descriptor: voffset / soffset in bytes, out-of-range requests read as zero
template <class T> __device__ __forceinline__ typename RawQuad<T>::type ldraw_buffer(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff);
template <> __device__ __forceinline__ float4 ldraw_buffer<float>(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
  typedef unsigned int u4 __attribute__((ext_vector_type(4)));
  const u4 d = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
  return make_float4(__uint_as_float(d.x), __uint_as_float(d.y), __uint_as_float(d.z), __uint_as_float(d.w));
}
template <> __device__ __forceinline__ uint2 ldraw_buffer<act16_t>(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
  typedef unsigned int u2 __attribute__((ext_vector_type(2)));
  const u2 d = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 0);
  return make_uint2(d.x, d.y);
}
template <class R> __device__ __forceinline__ R zero_raw();
template <> __device__ __forceinline__ float4 zero_raw<float4>() { return make_float4(0.f, 0.f, 0.f, 0.f); }
template <> __device__ __forceinline__ uint2 zero_raw<uint2>() { return make_uint2(0u, 0u); }

}  // namespace pe
