// Micro-benchmark: issue cost of v_mfma_f32_32x32x16_bf16 on gfx950 under the conditions of the x3 kernels.
// One workgroup per CU (grid = 256), WAVES waves per SIMD; s_memtime around a loop of MFMAs.
//   mode 0: 4 independent accumulators round-robin        mode 1: one accumulator (dependent chain)
//   mode 2: groups of 6 dependent MFMAs on 4 accumulators (the mfma_split order)
//   mode 3: mode 0 + 2 ds_read_b128 per 4 MFMAs           mode 4: mode 0 + 22 VALU (a split3) per 4 MFMAs
//   mode 5: waves with odd SIMD-slot run VALU only (176 per iteration) while the others run mode 0
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(512) void k(const float* in, float* out, long long* cyc, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[8192];
  const int tid = threadIdx.x;
  for (int i = tid; i < 8192; i += blockDim.x) lds[i] = in[i];
  __syncthreads();
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)in[tid * 8 + i]; b[i] = (__bf16)in[4096 + tid * 8 + i]; }
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j) for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;
  float4 v = *reinterpret_cast<const float4*>(in + tid * 4);
  float vs = 0.f;
  const bool valu_only = MODE == 5 && ((tid >> 8) & 1);       // second half of a 512-thread block
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (valu_only) {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float r1 = x[e] - __uint_as_float(__float_as_uint(x[e]) & 0xffff0000u);
          float r2 = r1 - __uint_as_float(__float_as_uint(r1) & 0xffff0000u);
          vs += r2; x[e] = r1 + 1.0f;
        }
        v = make_float4(x[0], x[1], x[2], x[3]);
      }
      continue;
    }
    if (MODE == 1) {
#pragma unroll
      for (int m = 0; m < 24; ++m) acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[0], 0, 0, 0);
    } else if (MODE == 2) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int m = 0; m < 6; ++m) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
    } else {
#pragma unroll
      for (int m = 0; m < 6; ++m) {
        if (MODE == 3) {
          const float4 l0 = *reinterpret_cast<const float4*>(lds + ((tid & 63) * 36 + m * 8) % 8000);
          const float4 l1 = *reinterpret_cast<const float4*>(lds + ((tid & 63) * 36 + m * 8 + 4) % 8000);
          a[0] = (__bf16)l0.x; b[0] = (__bf16)l1.y;
        }
        if (MODE == 4) {
          float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float r1 = x[e] - __uint_as_float(__float_as_uint(x[e]) & 0xffff0000u);
            float r2 = r1 - __uint_as_float(__float_as_uint(r1) & 0xffff0000u);
            vs += r2; x[e] = r1 + 1.0f;
          }
          v = make_float4(x[0], x[1], x[2], x[3]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
      }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = vs + v.x;
  for (int j = 0; j < 4; ++j) for (int q = 0; q < 16; ++q) s += acc[j][q];
  out[blockIdx.x * blockDim.x + tid] = s;
  if ((tid & 63) == 0) cyc[blockIdx.x * 8 + (tid >> 6)] = t1 - t0;
}

template <int MODE>
void run(int threads, const float* in, float* out, long long* cyc, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, in, out, cyc, iters);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, in, out, cyc, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h[8 * 256];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  const double mf = 24.0 * iters;
  printf("mode %d threads %d: wall %.3f ms; wave0 %.1f cyc/MFMA (ticks), wave%d %.1f; wall-derived %.1f ns/MFMA\n", MODE,
         threads, ms, h[0] / mf, threads / 64 - 1, h[threads / 64 - 1] / mf, ms * 1e6 / mf);
}

int main() {
  float *in, *out; long long* cyc;
  hipMalloc(&in, 8192 * 4); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 8 * 256 * 8);
  float h[8192];
  srand(1);
  for (int i = 0; i < 8192; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  const int iters = 2000;
  for (int threads : {256, 512}) {
    run<0>(threads, in, out, cyc, iters);
    run<1>(threads, in, out, cyc, iters);
    run<2>(threads, in, out, cyc, iters);
    run<3>(threads, in, out, cyc, iters);
    run<4>(threads, in, out, cyc, iters);
  }
  run<5>(512, in, out, cyc, iters);
  return 0;
}
