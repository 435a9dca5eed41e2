// Persistent LSTM recurrence: ONE launch walks all T time steps of a layer for up to 4 cells.
//
// Why: with one launch per step every workgroup re-reads its W_hh slice (196 KB) from L2/MALL each
// step -- 56 MB per step chip-wide -- and pays a kernel boundary; the step ran at 32 us against 10 us
// of MFMA time.  Here the W_hh slice lives in VGPRs for the whole sequence (192 registers per lane),
// only h_{t-1} (98 KB per workgroup) moves per step, and the step boundary is a group barrier among
// the 12 workgroups that share (cell, batch tile) -- different batch rows never interact.
//
// Workgroup = (cell, 64 batch rows, 32 hidden units x 4 gates); its 4 waves split K = H four ways
// (wave w: k in [w*H/4, (w+1)*H/4)), each lane half taking a contiguous H/8 run so A operands load as
// float4.  Partial 64x128 tiles are summed through LDS, the cell update is fused, c stays in registers.
//
// Cross-workgroup visibility follows the agent-scope release/acquire recipe of the CDNA guide
// (Guideline 16): every storing wave drains vmcnt, workgroup barrier, lane 0 release fence + drain +
// relaxed agent atomic add; consumer polls relaxed, then one acquire fence + drain + barrier, then
// plain loads.  Every written 128-byte line is written whole by one workgroup.  Spins are bounded;
// on timeout a sticky error word is set and all waits fall through.  Counters are zeroed by a
// memset node in front of every launch.  The grid (<= 192 workgroups, 1 per CU) must be fully
// resident: the host refuses the launch when the device has fewer CUs than workgroups.
#include "gemm_engine.h"

namespace {
using namespace pe;

constexpr int kMaxCells = 4;
constexpr int kRs = 132;                 // padded row stride of the 64 x 128 partial tiles
constexpr unsigned kSpinLimit = 4000000; // ~ seconds; a healthy wait is tens of microseconds
constexpr int kCtrStride = 32;           // one counter per 128-byte line

struct PFwdCells {
  const float* whh[kMaxCells];
  float* gates[kMaxCells];
  float* y[kMaxCells];
  float* c[kMaxCells];
  int reverse[kMaxCells];
};

struct PBwdCells {
  const float* whh_t[kMaxCells];
  float* gates[kMaxCells];
  const float* c[kMaxCells];
  const float* dy[kMaxCells];
  int reverse[kMaxCells];
};

__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }

// Group barrier: `members` workgroups add 1 each; wait until the counter reaches `target`.
__device__ __forceinline__ void group_barrier(unsigned* ctr, unsigned target, unsigned* err) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(2);
      if ((++spins & 63u) == 0u) {
        if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
        if (spins > kSpinLimit) {
          __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
}

template <int H>
__global__ __launch_bounds__(256, 1) void lstm_fwd_persistent_kernel(const PFwdCells cells, int B, int T, long ldy,
                                                                     unsigned* sync) {
  constexpr int KQ = H / 4, KH = KQ / 2, NV = KH / 4, NJ = H / 32;
  static_assert(H % 32 == 0, "hidden size is a multiple of 32");
  extern __shared__ __attribute__((aligned(16))) float red[];          // [4][64][kRs]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const int nbt = (B + 63) / 64;
  const int jt = blockIdx.x % NJ, bt = (blockIdx.x / NJ) % nbt, cell = blockIdx.x / (NJ * nbt);
  const int j0 = jt * 32, b0 = bt * 64;
  const int rev = cells.reverse[cell];
  float* y = cells.y[cell];
  float* gates = cells.gates[cell];
  float* cb = cells.c[cell];
  unsigned* err = sync;
  unsigned* ctr = sync + kCtrStride * (1 + cell * nbt + bt);

  // W_hh slice of this wave / lane: rows {g*H + j0 + r}, k = wv*KQ + hh*KH + s, kept in registers
  float bw[4][KH];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float* src = cells.whh[cell] + (long)(g * H + j0 + r) * H + wv * KQ + hh * KH;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const float4 t4 = *reinterpret_cast<const float4*>(src + 4 * v);
      bw[g][4 * v] = t4.x; bw[g][4 * v + 1] = t4.y; bw[g][4 * v + 2] = t4.z; bw[g][4 * v + 3] = t4.w;
    }
  }
  const int jj = tid & 31, bgrp = tid >> 5;
  float creg[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) creg[i] = 0.f;

  for (int step = 0; step < T; ++step) {
    const int t = rev ? T - 1 - step : step;
    const int tp = rev ? t + 1 : t - 1;
    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[i][g][q] = 0.f;
    if (step > 0) {
      float av[2][KH];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = b0 + i * 32 + r;
        const float* src = y + ((long)row * T + tp) * ldy + wv * KQ + hh * KH;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          float4 t4 = make_float4(0.f, 0.f, 0.f, 0.f);
          if (row < B) t4 = *reinterpret_cast<const float4*>(src + 4 * v);
          av[i][4 * v] = t4.x; av[i][4 * v + 1] = t4.y; av[i][4 * v + 2] = t4.z; av[i][4 * v + 3] = t4.w;
        }
      }
#pragma unroll
      for (int s = 0; s < KH; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int g = 0; g < 4; ++g) acc[i][g] = mfma32(av[i][s], bw[g][s], acc[i][g]);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int row = i * 32 + (q & 3) + 8 * (q >> 2) + 4 * hh;
          red[(wv * 64 + row) * kRs + g * 32 + r] = acc[i][g][q];
        }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int bl_ = bgrp + 8 * i;
      const int b = b0 + bl_;
      if (b < B) {
        const long rowi = (long)b * T + t;
        float* gp = gates + rowi * 4 * H + j0 + jj;
        float pre[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float* rp = red + bl_ * kRs + g * 32 + jj;
          pre[g] = ((rp[0] + rp[64 * kRs]) + (rp[2 * 64 * kRs] + rp[3 * 64 * kRs])) + gp[g * H];
        }
        const float gi = sigm(pre[0]), gf = sigm(pre[1]), gg = tanhf(pre[2]), go = sigm(pre[3]);
        const float cn = gf * creg[i] + gi * gg;
        creg[i] = cn;
        gp[0] = gi; gp[H] = gf; gp[2 * H] = gg; gp[3 * H] = go;
        cb[rowi * H + j0 + jj] = cn;
        y[rowi * ldy + j0 + jj] = go * tanhf(cn);
      }
    }
    if (step + 1 < T) group_barrier(ctr, (unsigned)(NJ * (step + 1)), err);
  }
}

// Backward: dh_t = dY_t + dgates_{t+1} . W_hh  (K = 4H: wave w owns gate block w, lane halves take
// H/2 contiguous k each, streamed in chunks of CH).  W_hh^T slice (rows j0 + r) stays in registers.
template <int H>
__global__ __launch_bounds__(256, 1) void lstm_bwd_persistent_kernel(const PBwdCells cells, int B, int T, long lddy,
                                                                     unsigned* sync) {
  constexpr int KH = H / 2, NJ = H / 32;
  constexpr int CH = (KH % 48 == 0) ? 48 : 16;           // A-operand chunk (values per lane per row tile)
  constexpr int NCH = KH / CH;
  static_assert(KH % CH == 0, "chunking");
  extern __shared__ __attribute__((aligned(16))) float red[];          // [4][64][33]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const int nbt = (B + 63) / 64;
  const int jt = blockIdx.x % NJ, bt = (blockIdx.x / NJ) % nbt, cell = blockIdx.x / (NJ * nbt);
  const int j0 = jt * 32, b0 = bt * 64;
  const int rev = cells.reverse[cell];
  const int K = 4 * H;
  float* gates = cells.gates[cell];
  const float* cb = cells.c[cell];
  const float* dy = cells.dy[cell];
  unsigned* err = sync;
  unsigned* ctr = sync + kCtrStride * (1 + cell * nbt + bt);

  float bw[KH];
  {
    const float* src = cells.whh_t[cell] + (long)(j0 + r) * K + wv * H + hh * KH;
#pragma unroll
    for (int v = 0; v < KH / 4; ++v) {
      const float4 t4 = *reinterpret_cast<const float4*>(src + 4 * v);
      bw[4 * v] = t4.x; bw[4 * v + 1] = t4.y; bw[4 * v + 2] = t4.z; bw[4 * v + 3] = t4.w;
    }
  }
  const int jj = tid & 31, bgrp = tid >> 5;
  float dcar[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) dcar[i] = 0.f;

  for (int step = 0; step < T; ++step) {
    const int t = rev ? step : T - 1 - step;
    const int tn = rev ? t - 1 : t + 1;
    const int tp = rev ? t + 1 : t - 1;
    f32x16 acc[2];
#pragma unroll
    for (int q = 0; q < 16; ++q) { acc[0][q] = 0.f; acc[1][q] = 0.f; }
    if (step > 0) {
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        float av[2][CH];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int row = b0 + i * 32 + r;
          const float* src = gates + ((long)row * T + tn) * K + wv * H + hh * KH + c * CH;
#pragma unroll
          for (int v = 0; v < CH / 4; ++v) {
            float4 t4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < B) t4 = *reinterpret_cast<const float4*>(src + 4 * v);
            av[i][4 * v] = t4.x; av[i][4 * v + 1] = t4.y; av[i][4 * v + 2] = t4.z; av[i][4 * v + 3] = t4.w;
          }
        }
#pragma unroll
        for (int s = 0; s < CH; ++s) {
          acc[0] = mfma32(av[0][s], bw[c * CH + s], acc[0]);
          acc[1] = mfma32(av[1][s], bw[c * CH + s], acc[1]);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = i * 32 + (q & 3) + 8 * (q >> 2) + 4 * hh;
        red[(wv * 64 + row) * 33 + r] = acc[i][q];
      }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int bl_ = bgrp + 8 * i;
      const int b = b0 + bl_;
      if (b < B) {
        const long rowi = (long)b * T + t;
        const int j = j0 + jj;
        const float* rp = red + bl_ * 33 + jj;
        const float dh = dy[rowi * lddy + j] + ((rp[0] + rp[64 * 33]) + (rp[2 * 64 * 33] + rp[3 * 64 * 33]));
        float* gp = gates + rowi * K + j;
        const float gi = gp[0], gf = gp[H], gg = gp[2 * H], go = gp[3 * H];
        const float cn = cb[rowi * H + j];
        const bool has_prev = rev ? (tp < T) : (tp >= 0);
        const float cprev = has_prev ? cb[((long)b * T + tp) * H + j] : 0.f;
        const float tc = tanhf(cn);
        const float dc = dh * go * (1.f - tc * tc) + dcar[i];
        gp[0] = dc * gg * gi * (1.f - gi);
        gp[H] = dc * cprev * gf * (1.f - gf);
        gp[2 * H] = dc * gi * (1.f - gg * gg);
        gp[3 * H] = dh * tc * go * (1.f - go);
        dcar[i] = dc * gf;
      }
    }
    if (step + 1 < T) group_barrier(ctr, (unsigned)(NJ * (step + 1)), err);
  }
}

int device_cus() {
  static int cus = -1;
  if (cus < 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
    cus = prop.multiProcessorCount;
  }
  return cus;
}

constexpr size_t kFwdLds = (size_t)4 * 64 * kRs * sizeof(float);
constexpr size_t kBwdLds = (size_t)4 * 64 * 33 * sizeof(float);

template <int H>
int launch_fwd(const PFwdCells& cells, int grid, int B, int T, long ldy, unsigned* sync, hipStream_t st) {
  static bool attr = false;
  if (!attr) {
    PE_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&lstm_fwd_persistent_kernel<H>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFwdLds));
    attr = true;
  }
  hipLaunchKernelGGL(lstm_fwd_persistent_kernel<H>, dim3(grid), dim3(256), kFwdLds, st, cells, B, T, ldy, sync);
  PE_LAUNCH_CHECK();
  return PE_OK;
}

template <int H>
int launch_bwd(const PBwdCells& cells, int grid, int B, int T, long lddy, unsigned* sync, hipStream_t st) {
  hipLaunchKernelGGL(lstm_bwd_persistent_kernel<H>, dim3(grid), dim3(256), kBwdLds, st, cells, B, T, lddy, sync);
  PE_LAUNCH_CHECK();
  return PE_OK;
}

int sync_words(int ncells, int B) { return kCtrStride * (1 + ncells * ((B + 63) / 64)); }

}  // namespace

extern "C" size_t pe_lstm_persistent_sync_bytes(int ncells, int B) {
  return (size_t)sync_words(ncells, B) * sizeof(unsigned);
}

// 1 if the persistent kernels can run this shape on the current device (hidden size instantiated and
// the whole grid co-resident at one workgroup per CU), else 0.
extern "C" int pe_lstm_persistent_supported(int ncells, int B, int H) {
  if (ncells < 1 || ncells > kMaxCells || B <= 0) return 0;
  if (!(H == 32 || H == 64 || H == 96 || H == 384)) return 0;
  const int grid = ncells * ((B + 63) / 64) * (H / 32);
  return grid <= device_cus() ? 1 : 0;
}

extern "C" int pe_lstm_fwd_persistent(int ncells, const float* const* whh, float* const* gates, float* const* y,
                                      float* const* cbuf, const int* reverse, long ldy, int B, int T, int H,
                                      unsigned* sync, void* stream) {
  if (!whh || !gates || !y || !cbuf || !reverse || !sync || T <= 0) return PE_E_ARG;
  if (!pe_lstm_persistent_supported(ncells, B, H) || (ldy & 3)) return PE_E_UNSUPPORTED;
  PFwdCells cells{};
  for (int i = 0; i < ncells; ++i) {
    if (!whh[i] || !gates[i] || !y[i] || !cbuf[i]) return PE_E_ARG;
    cells.whh[i] = whh[i]; cells.gates[i] = gates[i]; cells.y[i] = y[i]; cells.c[i] = cbuf[i];
    cells.reverse[i] = reverse[i];
  }
  hipStream_t st = pe_stream(stream);
  // word 0 is the sticky error flag (cleared only by the owner of the buffer); counters start at line 1
  PE_CHECK_HIP(hipMemsetAsync(sync + kCtrStride, 0, pe_lstm_persistent_sync_bytes(ncells, B) - kCtrStride * 4, st));
  const int grid = ncells * ((B + 63) / 64) * (H / 32);
  switch (H) {
    case 32: return launch_fwd<32>(cells, grid, B, T, ldy, sync, st);
    case 64: return launch_fwd<64>(cells, grid, B, T, ldy, sync, st);
    case 96: return launch_fwd<96>(cells, grid, B, T, ldy, sync, st);
    case 384: return launch_fwd<384>(cells, grid, B, T, ldy, sync, st);
  }
  return PE_E_UNSUPPORTED;
}

extern "C" int pe_lstm_bwd_persistent(int ncells, const float* const* whh_t, float* const* gates,
                                      const float* const* cbuf, const float* const* dy, const int* reverse,
                                      long lddy, int B, int T, int H, unsigned* sync, void* stream) {
  if (!whh_t || !gates || !cbuf || !dy || !reverse || !sync || T <= 0) return PE_E_ARG;
  if (!pe_lstm_persistent_supported(ncells, B, H) || (lddy & 3)) return PE_E_UNSUPPORTED;
  PBwdCells cells{};
  for (int i = 0; i < ncells; ++i) {
    if (!whh_t[i] || !gates[i] || !cbuf[i] || !dy[i]) return PE_E_ARG;
    cells.whh_t[i] = whh_t[i]; cells.gates[i] = gates[i]; cells.c[i] = cbuf[i]; cells.dy[i] = dy[i];
    cells.reverse[i] = reverse[i];
  }
  hipStream_t st = pe_stream(stream);
  PE_CHECK_HIP(hipMemsetAsync(sync + kCtrStride, 0, pe_lstm_persistent_sync_bytes(ncells, B) - kCtrStride * 4, st));
  const int grid = ncells * ((B + 63) / 64) * (H / 32);
  switch (H) {
    case 32: return launch_bwd<32>(cells, grid, B, T, lddy, sync, st);
    case 64: return launch_bwd<64>(cells, grid, B, T, lddy, sync, st);
    case 96: return launch_bwd<96>(cells, grid, B, T, lddy, sync, st);
    case 384: return launch_bwd<384>(cells, grid, B, T, lddy, sync, st);
  }
  return PE_E_UNSUPPORTED;
}
