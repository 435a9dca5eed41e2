// Persistent LSTM recurrence: ONE launch walks all T time steps of a layer for up to 4 cells.
//
// Measured on MI355X (tools/bench_lstm.py, B=256, H=384, 4 cells): with one launch per step the
// step takes 32 us against 12 us of MFMA time; an ablation showed the rest is latency that a
// launch-per-step structure cannot hide -- 5.5 us of uncoalesced operand loads, ~10 us of
// epilogue (x-projection loads, gate stores, store drain) and ~6 us of step boundary.  This kernel
// keeps the W_hh slice in VGPRs for the whole sequence and software-pipelines two independent
// 32-row half tiles per workgroup so every wait overlaps the other half's MFMA phase:
//
//   workgroup = (cell, 64 batch rows, 32 hidden units x 4 gates), 4 waves splitting K four ways;
//   per step and half:  wait(h_{t-1} of this half published by the 12 workgroups of the group)
//                       -> coalesced h_{t-1} rows -> LDS -> per-lane MFMA operands
//                       -> 192 MFMAs per wave -> partial tiles summed through LDS
//                       -> fused cell update, 16-byte stores -> arrive(group counter).
//
// Different batch rows never interact, so a "group" is the NJ = H/32 workgroups sharing
// (cell, batch tile, half).  Visibility follows the CDNA guide's Guideline 16, form R1: the
// handed-off payload (h_t forward, dgates_t backward) is stored write-through (sc1, 16 B per
// lane, every 128-byte line written whole by one workgroup), every storing wave drains vmcnt,
// workgroup barrier, ONE lane does a relaxed agent-scope atomic add; the consumer polls that
// counter relaxed, then a workgroup barrier; every load of handed-off bytes is itself an sc1 load
// (L1-bypassing), which replaces the acquire fence (guide: Valid forms, first table row).  Spins are bounded (sticky error word, all later waits fall through); counters are
// zeroed by a memset node in front of every launch; the grid (<= 192 workgroups at one per CU)
// must be co-resident, so the host refuses shapes that exceed the device's CU count.
//
// Template flag X3 (the *_x3 entry points, H % 64 == 0): the recurrent products run on the bf16 MFMA pipe
// as the exact three-term split of gemm_engine.h -- W hi/mid terms in registers, lo terms in LDS, operand
// rows split per use -- 144 v_mfma_f32_32x32x16_bf16 per item and wave instead of 192 fp32 MFMAs of
// twice the cycles.  Everything about the hand-off protocol is identical in both forms.
#include <stdlib.h>
#include <type_traits>
#include "gemm_engine.h"

namespace {
using namespace pe;

constexpr int kMaxCells = 4;
constexpr int kEPieces = 44;             // pieces the forward kernel's cell update is cut into (one per MFMA gap)
constexpr int kRs = 132;                 // row stride of the 32 x 128 partial tiles (forward)
constexpr int kRb = 36;                  // row stride of the 32 x 32 partial tiles (backward)
constexpr unsigned kSpinLimit = 4000000; // a healthy wait is microseconds
constexpr int kCtrStride = 32;           // one counter per 128-byte line

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

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
  float* dbias[kMaxCells];               // optional [batch tiles][4H] column sums of the gate gradients (k-split kernel)
  unsigned* amax[kMaxCells];             // optional zeroed words: largest gate-gradient magnitude per cell (k-split kernel)
  int reverse[kMaxCells];
};

// Gate non-linearities on the hardware exp2 / rcp units (v_exp_f32, v_rcp_f32: ~1 ulp each).  The libm expf /
// tanhf cost ~650 VALU instructions per thread and item (s_memtime: 4 260 of 16 900 cycles of a forward
// item); these cost ~150 and agree with them to ~3e-7 (tests/test_ops_gpu.py holds the layer to 1e-5).
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }
__device__ __forceinline__ float sigm(float x) { return __builtin_amdgcn_rcpf(1.0f + fast_exp(-x)); }
__device__ __forceinline__ float tanh_fast(float x) {
  const float ax = fabsf(x), x2 = x * x;
  const float e = fast_exp(-2.0f * ax);
  const float big = (1.0f - e) * __builtin_amdgcn_rcpf(1.0f + e);          // |x| >= 0.25: no harmful cancellation
  const float small = ax * fmaf(x2, fmaf(x2, fmaf(x2, fmaf(x2, 62.0f / 2835.0f, -17.0f / 315.0f), 2.0f / 15.0f),
                                         -1.0f / 3.0f), 1.0f);             // odd series, next term < 2e-9 at 0.25
  return copysignf(ax < 0.25f ? small : big, x);
}

// The same two functions for N independent values, stage by stage (all exponentials, then all reciprocals, ...):
// at one wave per SIMD nothing else hides the latency of a dependent v_exp -> v_add -> v_rcp chain, so the
// independent chains have to be interleaved in program order.  Results are bit-identical to sigm / tanh_fast.
template <int N>
__device__ __forceinline__ void sigm_n(const float (&x)[N], float (&y)[N]) {
  float e[N];
#pragma unroll
  for (int k = 0; k < N; ++k) e[k] = __builtin_amdgcn_exp2f(-x[k] * 1.44269504088896340736f);
#pragma unroll
  for (int k = 0; k < N; ++k) e[k] = 1.0f + e[k];
#pragma unroll
  for (int k = 0; k < N; ++k) y[k] = __builtin_amdgcn_rcpf(e[k]);
}
template <int N>
__device__ __forceinline__ void tanh_n(const float (&x)[N], float (&y)[N]) {
  float ax[N], e[N], rc[N], sm[N];
#pragma unroll
  for (int k = 0; k < N; ++k) ax[k] = fabsf(x[k]);
#pragma unroll
  for (int k = 0; k < N; ++k) e[k] = __builtin_amdgcn_exp2f((-2.0f * ax[k]) * 1.44269504088896340736f);
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const float x2 = x[k] * x[k];
    sm[k] = ax[k] * fmaf(x2, fmaf(x2, fmaf(x2, fmaf(x2, 62.0f / 2835.0f, -17.0f / 315.0f), 2.0f / 15.0f),
                              -1.0f / 3.0f), 1.0f);
  }
#pragma unroll
  for (int k = 0; k < N; ++k) rc[k] = __builtin_amdgcn_rcpf(1.0f + e[k]);
#pragma unroll
  for (int k = 0; k < N; ++k) y[k] = copysignf(ax[k] < 0.25f ? sm[k] : (1.0f - e[k]) * rc[k], x[k]);
}

// c_t = f * c_{t-1} + i * g with the contraction spelled out: the forward kernels evaluate it in several code
// instances (halves, peeled steps, the piecewise epilogue) and hipcc is free to fuse either product into the add --
// two instances that choose differently make a sample's result depend on its position in the batch.
__device__ __forceinline__ float cell_c(float f, float c, float i, float g) { return fmaf(f, c, i * g); }

// The stages of tanh_n / sigm_n one value at a time, for epilogues that are cut into pieces and issued between MFMAs
// (same operations in the same order: bit-identical to the functions above).
__device__ __forceinline__ float sig_exp(float x) { return __builtin_amdgcn_exp2f(-x * 1.44269504088896340736f); }
__device__ __forceinline__ void tanh_s1(float x, float& ax, float& e) {
  ax = fabsf(x);
  e = __builtin_amdgcn_exp2f((-2.0f * ax) * 1.44269504088896340736f);
}
__device__ __forceinline__ float tanh_s2(float x, float ax) {
  const float x2 = x * x;
  return ax * fmaf(x2, fmaf(x2, fmaf(x2, fmaf(x2, 62.0f / 2835.0f, -17.0f / 315.0f), 2.0f / 15.0f), -1.0f / 3.0f), 1.0f);
}
__device__ __forceinline__ float tanh_s4(float x, float ax, float e, float sm, float rc) {
  return copysignf(ax < 0.25f ? sm : (1.0f - e) * rc, x);
}

// X3 variants: the recurrent products run as the exact three-term bf16 split (gemm_engine.h).  The W_hh
// slice is split once; its hi and mid terms stay in registers for the whole sequence, the lo terms of the
// last blocks live in LDS (they feed one product in six and the register file is full).  The h / dgates
// rows are split per use, one block ahead of the MFMAs that consume them.  One MFMA k-block = the 8 k of
// each lane half (the halves own disjoint k ranges; any pairing is a valid k order).
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split8(const float4& p, const float4& q, bf16x8 (&out)[3]) {
  const f32x2 v[4] = {{p.x, p.y}, {p.z, p.w}, {q.x, q.y}, {q.z, q.w}};
  u32x4 a, b, c;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const u32x2 u0 = __builtin_bit_cast(u32x2, v[i]);
    const f32x2 r1 = v[i] - __builtin_bit_cast(f32x2, u0 & 0xffff0000u);
    const u32x2 u1 = __builtin_bit_cast(u32x2, r1);
    const f32x2 r2 = r1 - __builtin_bit_cast(f32x2, u1 & 0xffff0000u);
    const u32x2 u2 = __builtin_bit_cast(u32x2, r2);
    a[i] = __builtin_amdgcn_perm(u0[1], u0[0], 0x07060302u);
    b[i] = __builtin_amdgcn_perm(u1[1], u1[0], 0x07060302u);
    c[i] = __builtin_amdgcn_perm(u2[1], u2[0], 0x07060302u);
  }
  out[0] = __builtin_bit_cast(bf16x8, a);
  out[1] = __builtin_bit_cast(bf16x8, b);
  out[2] = __builtin_bit_cast(bf16x8, c);
}
// four values -> three terms of 4 bf16 each (same truncating split as split8)
__device__ __forceinline__ void split4(const float4& p, u32x2 (&out)[3]) {
  const f32x2 v[2] = {{p.x, p.y}, {p.z, p.w}};
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const u32x2 u0 = __builtin_bit_cast(u32x2, v[i]);
    const f32x2 r1 = v[i] - __builtin_bit_cast(f32x2, u0 & 0xffff0000u);
    const u32x2 u1 = __builtin_bit_cast(u32x2, r1);
    const f32x2 r2 = r1 - __builtin_bit_cast(f32x2, u1 & 0xffff0000u);
    const u32x2 u2 = __builtin_bit_cast(u32x2, r2);
    out[0][i] = __builtin_amdgcn_perm(u0[1], u0[0], 0x07060302u);
    out[1][i] = __builtin_amdgcn_perm(u1[1], u1[0], 0x07060302u);
    out[2][i] = __builtin_amdgcn_perm(u2[1], u2[0], 0x07060302u);
  }
}
typedef pe_half_t half_x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x2 round4(const float4& p) {
  half_x4 o;
  o[0] = (pe_half_t)p.x; o[1] = (pe_half_t)p.y; o[2] = (pe_half_t)p.z; o[3] = (pe_half_t)p.w;
  return __builtin_bit_cast(u32x2, o);
}
// (a term, b term) of the six products.  a_hi x W_lo goes last: W_lo may come from LDS, and its read then
// completes under the five products that do not need it.
// mixed precision: 8 consecutive k rounded to bf16 (RNE)
__device__ __forceinline__ bf16x8 round8(const float4& p, const float4& q) {
  bf16x8 o;
  o[0] = (pe_half_t)p.x; o[1] = (pe_half_t)p.y; o[2] = (pe_half_t)p.z; o[3] = (pe_half_t)p.w;
  o[4] = (pe_half_t)q.x; o[5] = (pe_half_t)q.y; o[6] = (pe_half_t)q.z; o[7] = (pe_half_t)q.w;
  return o;
}
constexpr int kTa[6] = {2, 1, 1, 0, 0, 0}, kTb[6] = {0, 1, 0, 1, 0, 2};
// blocks whose W lo term stays in registers; the rest (at most 32 KB per workgroup) goes to LDS
template <int H> constexpr int fwd_lo_reg_blocks() { return (H / 64) <= 1 ? (H / 64) : 1; }
template <int H> constexpr int bwd_lo_reg_blocks() { return (H / 16) <= 16 ? (H / 16) : 16; }

// 16-byte write-through (sc1) store: the data leaves for memory, no release fence needed later.
__device__ __forceinline__ void store_sc1(__amdgpu_buffer_rsrc_t rs, unsigned byte_off, float4 v) {
  u32x4 d = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
  __builtin_amdgcn_raw_buffer_store_b128(d, rs, byte_off, 0, 16 /* sc1 */);
}

// arrive: all of this workgroup's payload stores are complete, then one relaxed agent-scope add
__device__ __forceinline__ void group_arrive(unsigned* ctr) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// 16-byte sc1 load of handed-off data: served by L2 / memory, never by this CU's (possibly stale) L1
__device__ __forceinline__ float4 load_sc1(__amdgpu_buffer_rsrc_t rs, unsigned byte_off) {
  const u32x4 d = __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, 16 /* sc1 */);
  return make_float4(__uint_as_float(d.x), __uint_as_float(d.y), __uint_as_float(d.z), __uint_as_float(d.w));
}

// the same two for handed-off data kept in 16 bits (bf16, round to nearest even): 8 bytes per lane.  The mixed-
// precision backward kernel exchanges its partial dh tiles this way -- it is bound by that exchange, not by its MFMAs
// (the three-term and the one-term build took the same time), and the operands of these products are bf16 already.
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void store_sc1_h(__amdgpu_buffer_rsrc_t rs, unsigned byte_off, float4 v) {
  typedef __bf16 b2 __attribute__((ext_vector_type(2)));
  typedef float f2 __attribute__((ext_vector_type(2)));
  const f2 lo = {v.x, v.y}, hi = {v.z, v.w};
  const u32x2_t d = {__builtin_bit_cast(unsigned, __builtin_convertvector(lo, b2)),
                     __builtin_bit_cast(unsigned, __builtin_convertvector(hi, b2))};
  __builtin_amdgcn_raw_buffer_store_b64(d, rs, byte_off, 0, 16 /* sc1 */);
}
__device__ __forceinline__ float4 load_sc1_h(__amdgpu_buffer_rsrc_t rs, unsigned byte_off) {
  const u32x2_t d = __builtin_amdgcn_raw_buffer_load_b64(rs, byte_off, 0, 16 /* sc1 */);
  return make_float4(__uint_as_float(d.x << 16), __uint_as_float(d.x & 0xffff0000u), __uint_as_float(d.y << 16),
                     __uint_as_float(d.y & 0xffff0000u));
}

// wait: one lane polls the counter (an sc1 load) until the group has arrived, then the workgroup
// barrier releases the other waves.  Every load of handed-off bytes in these kernels is an sc1 load and
// every such byte was stored sc1 and drained before the producer's atomic add (guide, Valid forms,
// first table row), so no L1 invalidate is needed; the wavefront-scope fence only pins compiler order.
__device__ __forceinline__ void group_wait(unsigned* ctr, unsigned target, unsigned* err) {
  if (threadIdx.x == 0) {
    unsigned spins = 0;
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if ((++spins & 63u) == 0u) {
        if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
        if (spins > kSpinLimit) {
          __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  __syncthreads();
}

// The same wait without the workgroup barrier: lane 0 of EVERY wave polls (four sc1 loads of one line instead of
// one), the waves stay decoupled.  For kernels whose LDS hazards are ordered by other barriers.
__device__ __forceinline__ void wave_wait(unsigned* ctr, unsigned target, unsigned* err) {
  if ((threadIdx.x & 63) == 0) {
    unsigned spins = 0;
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if ((++spins & 63u) == 0u) {
        if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
        if (spins > kSpinLimit) {
          __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Hand-off by FLAGS instead of a counter: every producer wave owns one word of the group's flag block and stores the
// step it has completed (plain sc1 store behind its drained payload); a consumer wave reads all n words with one
// load and waits until none is behind.  Same-address atomics serialise in the L2 (48 arrivals on one counter cost
// more than the payload); stores to distinct words of a line do not.
__device__ __forceinline__ unsigned flags_peek(const unsigned* flags, int n) {
  const int l = threadIdx.x & 63;
  return l < n ? __hip_atomic_load(flags + l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0xffffffffu;
}
__device__ __forceinline__ void flags_wait(const unsigned* flags, int n, unsigned target, unsigned seen, unsigned* err) {
  unsigned spins = 0;
  while (__ballot(seen < target) != 0ull) {
    __builtin_amdgcn_s_sleep(1);
    if ((++spins & 63u) == 0u) {
      if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
      if (spins > kSpinLimit) {
        __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
    seen = flags_peek(flags, n);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Workgroup -> (group, member).  A group's NJ workgroups all read the same operand rows (h_{t-1} forward,
// dgates_{t+1} backward) and hand each other their results, so they are placed on ONE XCD (hardware
// dispatches workgroup b to XCD b % 8): the 12-fold re-read then hits that XCD's L2 instead of crossing
// the fabric (PMC: 13 GB per backward launch with the round-robin order) and the sc1 hand-off stays local.
__device__ __forceinline__ void group_of_block(int NJ, int& jt, int& gidx) {
  const int ngroups = gridDim.x / NJ;
  if ((ngroups & 7) == 0) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    gidx = xcd * (ngroups >> 3) + slot / NJ;
    jt = slot % NJ;
  } else {
    gidx = blockIdx.x / NJ;
    jt = blockIdx.x % NJ;
  }
}

// --------------------------------------------------------------------------------------- forward, overlapped
// Same tiling, same hand-off protocol, same arithmetic (bit for bit) as lstm_fwd_persistent_kernel, but the part of
// an item that is not MFMA work -- partial-tile exchange, gate functions, the stores of h / gates / c, the store
// drain, the arrival, the wait for the next item's h rows and their fetch -- no longer runs between two MFMA
// phases: item i-1's epilogue executes UNDER item i's MFMAs.  One iteration (item i = (step, half), step >= 1):
//
//     barrier #1   As(i) [h_{t-1} rows] and red(i-1) [partial tiles of the previous item] are complete
//     MFMA blocks 0 .. NB/2-1  interleaved with  E(i-1): red + x-projection -> gates -> c, h;  h stored sc1 FIRST,
//                                                 then gates, c;  s_waitcnt vmcnt(5) = the h store has completed
//     barrier #2, ONE lane: arrive(i-1)
//     MFMA blocks NB/2 .. NB-2;  ONE lane polls the counter of item i+1 (= the arrivals of item i-1)
//     barrier #3, every lane issues the sc1 loads of item i+1's h rows (in flight under the last block)
//     MFMA block NB-1;  accumulators -> red(i)
//     barrier #0   every wave has finished reading As(i);  h rows of item i+1 -> As
//
// What overlaps is the MEMORY side of the epilogue (store drain, hand-off propagation, counter poll, h-row fetch):
// all of it is in flight under MFMAs.  The gate arithmetic issues in order with the wave's MFMAs, and hipcc keeps
// the two instruction groups apart even under sched_group_barrier (s_memtime stamps, tools/stamp_lstm.py: 5.2 k of
// an item's 12.4 k cycles were that region, 2.3 k of them MFMA issue), so for H = 384 the update is cut BY HAND
// into kEPieces pieces of a few instructions, one per MFMA gap of the first three blocks, and the last block runs
// gate by gate with the next item's row requests and the finished gates' partial-tile stores in its gaps
// (static_for + sched_barrier(0) after every MFMA; item 12.3 -> 10.0 k cycles).  The region is branch-free: step 0
// (no recurrent input) is peeled, rows beyond B are handled by the buffer range check (loads return 0, stores
// are dropped) instead of exec-masked branches.
// The consumer of item i-1's h is item i+1 (same half, next step), so the hand-off has half an MFMA phase to
// become visible and the fetch the other half to land.  LDS: As (50 KB) and red (68 KB) can no longer share
// space; the W_hh lo terms that do not fit the register file next to them keep (NB - NBR) blocks in LDS.
// diagnostic stamps (PE_LSTM_STAMP=1, tools/stamp_lstm.py): s_memtime around the regions of an iteration, summed per
// workgroup by wave 0 into words [2048 + 16 * block, +16) of the sync buffer.  Never active in the product build.
__device__ __forceinline__ unsigned long long stamp_now() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#define PE_STAMP(k)                                         \
  if constexpr (STAMP) {                                    \
    const unsigned long long _n = stamp_now();              \
    st_acc[k] += _n - st_last;                              \
    st_last = _n;                                           \
  }

template <int H, int TERMS, int NBR_, bool STAMP = false>
__global__ __launch_bounds__(256, 1) void lstm_fwd_persistent_v2_kernel(const PFwdCells cells, int B, int T, long ldy,
                                                                        unsigned y_bytes, unsigned g_bytes,
                                                                        unsigned c_bytes, unsigned* sync) {
  static_assert(TERMS == 3 || TERMS == 1, "bf16-term pipelines only");
  constexpr int KQ = H / 4, KH = KQ / 2, NJ = H / 32;
  constexpr int ASTR = H + 4, ROW4 = H / 4, NST = ROW4 / 8;
  constexpr int NB = KH / 8;                          // 8-k blocks per lane
  constexpr int NBR = TERMS == 3 ? NBR_ : NB;         // blocks whose lo term lives in registers
  constexpr int NB1 = NB / 2;                         // blocks of the first half (epilogue underneath)
  static_assert(H % 64 == 0, "H % 64 == 0");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                                   // [32][ASTR]
  float* red = smem + 32 * ASTR;                      // [4][32][kRs]
  uint4* wlo_lds = reinterpret_cast<uint4*>(smem + 32 * ASTR + 4 * 32 * kRs);   // [4][NB - NBR][256]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const int nbt = (B + 63) / 64;
  int jt, gidx;
  group_of_block(NJ, jt, gidx);
  const int bt = gidx % nbt, cell = gidx / nbt;
  const int j0 = jt * 32, b0 = bt * 64;
  const int rev = cells.reverse[cell];
  unsigned* err = sync;
  unsigned* ctr = sync + kCtrStride * (1 + (cell * nbt + bt) * 2);
  const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(cells.y[cell], 0, y_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc(cells.gates[cell], 0, g_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc(cells.c[cell], 0, c_bytes, 0x00020000);

  bf16x8 bwhm[4][NB][TERMS == 3 ? 2 : 1], bwlo[4][NBR > 0 ? NBR : 1];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float* src = cells.whh[cell] + (long)(g * H + j0 + r) * H + wv * KQ + hh * KH;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const float4 w0 = *reinterpret_cast<const float4*>(src + 8 * b);
      const float4 w1 = *reinterpret_cast<const float4*>(src + 8 * b + 4);
      if constexpr (TERMS == 3) {
        bf16x8 t3[3];
        split8(w0, w1, t3);
        bwhm[g][b][0] = t3[0];
        bwhm[g][b][1] = t3[1];
        if (b < NBR) bwlo[g][b < NBR ? b : 0] = t3[2];
        else wlo_lds[(g * (NB - NBR) + (b - NBR)) * 256 + tid] = __builtin_bit_cast(uint4, t3[2]);
      } else {
        bwhm[g][b][0] = round8(w0, w1);
      }
    }
  }
  const int prow = tid >> 3, pq = tid & 7;            // cell-update item: row prow, hidden units j0 + 4 pq .. +3
  float4 creg[2];
  creg[0] = creg[1] = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 stage[NST];
  float4 xp_prev[4], xp_cur[4];

  // element (batch row of this thread's update item, time) of item (step, hf); 0x7fffffff past the batch: every
  // byte offset formed from it lies outside its buffer (loads give 0, stores are dropped; 32-bit wrap is harmless
  // because the descriptors' sizes are far below 2^31 elements -- checked by the host)
  auto item_elem = [&](int step, int hf) -> unsigned {
    const int pb = b0 + 32 * hf + prow;
    const int t = rev ? T - 1 - step : step;
    return pb < B ? (unsigned)(pb * T + t) : 0x7fffffffu;
  };
  auto oob = [&](unsigned e, unsigned off) -> unsigned { return e == 0x7fffffffu ? 0xfffffff0u : off; };
  auto load_xp = [&](float4 (&xp)[4], int step, int hf) {
    const unsigned e = item_elem(step, hf);
    const unsigned base = oob(e, (e * (unsigned)(4 * H) + (unsigned)(j0 + 4 * pq)) * 4u);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const u32x4 d = __builtin_amdgcn_raw_buffer_load_b128(grs, base, (unsigned)(g * H * 4), 0);
      xp[g] = make_float4(__uint_as_float(d.x), __uint_as_float(d.y), __uint_as_float(d.z), __uint_as_float(d.w));
    }
  };
  auto fetch_rows = [&](int step, int hf) {           // sc1 loads of h_{t-1} rows of item (step, hf): its group arrived
    const int t = rev ? T - 1 - step : step, tp = rev ? t + 1 : t - 1;
    const int brow = b0 + 32 * hf + (tid >> 3);
    const unsigned base = brow < B ? ((unsigned)(brow * T + tp) * (unsigned)ldy + (unsigned)(tid & 7) * 4u) * 4u
                                   : 0xfffffff0u;
#pragma unroll
    for (int v = 0; v < NST; ++v) stage[v] = load_sc1(yrs, base + (brow < B ? (unsigned)v * 128u : 0u));
  };
  auto store4 = [&](__amdgpu_buffer_rsrc_t rs, unsigned off, unsigned soff, const float4& v) {
    const u32x4 d = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
    __builtin_amdgcn_raw_buffer_store_b128(d, rs, off, soff, 0);
  };
  // gate functions + state update of item (step, hf) from its partial tiles in red and its x-projection; branch-free
  auto cell_update = [&](int step, int hf, const float4 (&xp)[4]) {
    const unsigned e = item_elem(step, hf);
    float4 pre[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float* rp = red + prow * kRs + g * 32 + 4 * pq;
      const float4 p0 = *reinterpret_cast<const float4*>(rp);
      const float4 p1 = *reinterpret_cast<const float4*>(rp + 32 * kRs);
      const float4 p2 = *reinterpret_cast<const float4*>(rp + 2 * 32 * kRs);
      const float4 p3 = *reinterpret_cast<const float4*>(rp + 3 * 32 * kRs);
      pre[g].x = ((p0.x + p1.x) + (p2.x + p3.x)) + xp[g].x;
      pre[g].y = ((p0.y + p1.y) + (p2.y + p3.y)) + xp[g].y;
      pre[g].z = ((p0.z + p1.z) + (p2.z + p3.z)) + xp[g].z;
      pre[g].w = ((p0.w + p1.w) + (p2.w + p3.w)) + xp[g].w;
    }
    const float sx[12] = {pre[0].x, pre[0].y, pre[0].z, pre[0].w, pre[1].x, pre[1].y, pre[1].z, pre[1].w,
                          pre[3].x, pre[3].y, pre[3].z, pre[3].w};
    const float tx[4] = {pre[2].x, pre[2].y, pre[2].z, pre[2].w};
    float sy[12], ty[4];
    sigm_n<12>(sx, sy);
    tanh_n<4>(tx, ty);
    const float4 gi = make_float4(sy[0], sy[1], sy[2], sy[3]);
    const float4 gf = make_float4(sy[4], sy[5], sy[6], sy[7]);
    const float4 go = make_float4(sy[8], sy[9], sy[10], sy[11]);
    const float4 gg = make_float4(ty[0], ty[1], ty[2], ty[3]);
    float4 cn;
    cn.x = cell_c(gf.x, creg[hf].x, gi.x, gg.x);
    cn.y = cell_c(gf.y, creg[hf].y, gi.y, gg.y);
    cn.z = cell_c(gf.z, creg[hf].z, gi.z, gg.z);
    cn.w = cell_c(gf.w, creg[hf].w, gi.w, gg.w);
    creg[hf] = cn;
    const float cx[4] = {cn.x, cn.y, cn.z, cn.w};
    float cy[4];
    tanh_n<4>(cx, cy);
    const float4 hv = make_float4(go.x * cy[0], go.y * cy[1], go.z * cy[2], go.w * cy[3]);
    // h first and write-through: it is what the group waits for; the five stores behind it need not have
    // completed when this workgroup arrives (vmcnt counts in order)
    store_sc1(yrs, oob(e, (e * (unsigned)ldy + (unsigned)(j0 + 4 * pq)) * 4u), hv);
    const unsigned gb = oob(e, (e * (unsigned)(4 * H) + (unsigned)(j0 + 4 * pq)) * 4u);
    store4(grs, gb, 0u, gi);
    store4(grs, gb, (unsigned)(H * 4), gf);
    store4(grs, gb, (unsigned)(2 * H * 4), gg);
    store4(grs, gb, (unsigned)(3 * H * 4), go);
    store4(crs, oob(e, (e * (unsigned)H + (unsigned)(j0 + 4 * pq)) * 4u), 0u, cn);
  };
  auto poll = [&](int half, unsigned target) {         // ONE lane; relaxed sc1 polls, bounded
    if (threadIdx.x == 0) {
      unsigned* c = ctr + kCtrStride * half;
      unsigned spins = 0;
      while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(1);
        if ((++spins & 63u) == 0u) {
          if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
          if (spins > kSpinLimit) {
            __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
          }
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  auto arrive = [&](int half) {
    if (threadIdx.x == 0)
      __hip_atomic_fetch_add(ctr + kCtrStride * half, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  auto commit_rows = [&]() {
    float* adst = As + (tid >> 3) * ASTR + (tid & 7) * 4;
#pragma unroll
    for (int v = 0; v < NST; ++v) *reinterpret_cast<float4*>(adst + 32 * v) = stage[v];
  };

  // ---- step 0 (h_{-1} = 0: no recurrent product).  Item (0,0) completes here; item (0,1) is left pending with
  // all-zero partial tiles so that the steady-state loop needs no special case.
  for (int z = tid; z < 4 * 32 * kRs / 4; z += 256) reinterpret_cast<float4*>(red)[z] = make_float4(0.f, 0.f, 0.f, 0.f);
  load_xp(xp_prev, 0, 0);
  __syncthreads();
  cell_update(0, 0, xp_prev);
  asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
  __syncthreads();
  arrive(0);
  load_xp(xp_prev, 0, 1);
  if (T > 1) {
    poll(0, (unsigned)NJ);
    __syncthreads();
    fetch_rows(1, 0);
    commit_rows();
    __syncthreads();
  }

  f32x16 acc[4];
  const float* asrc = As + r * ASTR + wv * KQ + hh * KH;
  auto mfma_block = [&](int b) {
    const float4 a0 = *reinterpret_cast<const float4*>(asrc + 8 * b);
    const float4 a1 = *reinterpret_cast<const float4*>(asrc + 8 * b + 4);
    if constexpr (TERMS == 3) {
      bf16x8 fa[3];
      split8(a0, a1, fa);
      bf16x8 wl[4];
#pragma unroll
      for (int g = 0; g < 4; ++g)
        wl[g] = b < NBR ? bwlo[g][b < NBR ? b : 0]
                        : __builtin_bit_cast(bf16x8, wlo_lds[(g * (NB - NBR) + (b - NBR)) * 256 + tid]);
#pragma unroll
      for (int t6 = 0; t6 < 6; ++t6)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          acc[g] = mfma_bf16(kTb[t6] == 2 ? wl[g] : bwhm[g][b][kTb[t6]], fa[kTa[t6]], acc[g]);
    } else {
      const bf16x8 fa = round8(a0, a1);
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[g] = mfma_bf16(bwhm[g][b][0], fa, acc[g]);
    }
  };
  unsigned long long st_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_last = 0;
  if constexpr (STAMP) st_last = stamp_now();

  for (int step = 1; step < T; ++step) {
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {                   // unrolled: creg[hf] and the counter halves are static
      // item i = (step, hf); pending epilogue: item i-1 = (step - 1 + hf, hf ^ 1)
      const int pstep = hf ? step : step - 1, phf = hf ^ 1;
      // ---- region 1 (straight-line): first MFMA blocks with the pending epilogue spread between them
      load_xp(xp_cur, step, hf);
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[g][q] = 0.f;
      if constexpr (TERMS == 3 && NB1 * 24 >= kEPieces + NB1) {
        // The pending epilogue E(i-1) in kEPieces pieces of a few instructions, one per MFMA gap (an MFMA holds the
        // vector issue port for 8 of its 32 cycles): sched_barrier(0) after every MFMA pins the interleave, which
        // hipcc does not produce by itself.  Operations and their order are those of cell_update.
        float4 pp[2][4], pre[4], cnv, hv;
        float se[12], sy[12], tax[4], te[4], tsm[4], trc[4], ty[4], cax[4], ce[4], csm[4], crc[4], cy[4];
        unsigned e_el = 0u, gbo = 0u;
        auto rd_red = [&](int g, float4 (&d)[4]) {
          const float* rp = red + prow * kRs + g * 32 + 4 * pq;
          d[0] = *reinterpret_cast<const float4*>(rp);
          d[1] = *reinterpret_cast<const float4*>(rp + 32 * kRs);
          d[2] = *reinterpret_cast<const float4*>(rp + 2 * 32 * kRs);
          d[3] = *reinterpret_cast<const float4*>(rp + 3 * 32 * kRs);
        };
        auto sxv = [&](int k) -> float {               // sigmoid inputs: gates i, f, o
          const float4& v = pre[k < 4 ? 0 : k < 8 ? 1 : 3];
          return (k & 3) == 0 ? v.x : (k & 3) == 1 ? v.y : (k & 3) == 2 ? v.z : v.w;
        };
        auto comp = [&](const float4& v, int k) -> float { return k == 0 ? v.x : k == 1 ? v.y : k == 2 ? v.z : v.w; };
        auto e_piece = [&](auto P) {
          constexpr int p = decltype(P)::value;
          if constexpr (p == 0) { e_el = item_elem(pstep, phf); rd_red(0, pp[0]); }
          else if constexpr (p == 1) rd_red(1, pp[1]);
          else if constexpr (p >= 2 && p <= 9) {       // pre[g]: halves xy / zw; the next gate's tiles two pieces ahead
            constexpr int g = (p - 2) / 2, half = (p - 2) % 2;
            const float4(&q)[4] = pp[g & 1];
            if constexpr (half == 0) {
              pre[g].x = ((q[0].x + q[1].x) + (q[2].x + q[3].x)) + xp_prev[g].x;
              pre[g].y = ((q[0].y + q[1].y) + (q[2].y + q[3].y)) + xp_prev[g].y;
            } else {
              pre[g].z = ((q[0].z + q[1].z) + (q[2].z + q[3].z)) + xp_prev[g].z;
              pre[g].w = ((q[0].w + q[1].w) + (q[2].w + q[3].w)) + xp_prev[g].w;
              if constexpr (g + 2 < 4) rd_red(g + 2, pp[g & 1]);
            }
          } else if constexpr (p >= 10 && p <= 12) {   // sigmoid: exponentials
#pragma unroll
            for (int k = 4 * (p - 10); k < 4 * (p - 10) + 4; ++k) se[k] = sig_exp(sxv(k));
          } else if constexpr (p == 13 || p == 14) {   // tanh(g gate): |x|, exponential
#pragma unroll
            for (int k = 2 * (p - 13); k < 2 * (p - 13) + 2; ++k) tanh_s1(comp(pre[2], k), tax[k], te[k]);
          } else if constexpr (p == 15 || p == 16) {
#pragma unroll
            for (int k = 6 * (p - 15); k < 6 * (p - 15) + 6; ++k) se[k] = 1.0f + se[k];
          } else if constexpr (p >= 17 && p <= 20) {
#pragma unroll
            for (int k = 3 * (p - 17); k < 3 * (p - 17) + 3; ++k) sy[k] = __builtin_amdgcn_rcpf(se[k]);
          } else if constexpr (p >= 21 && p <= 24) {
            tsm[p - 21] = tanh_s2(comp(pre[2], p - 21), tax[p - 21]);
          } else if constexpr (p == 25 || p == 26) {
#pragma unroll
            for (int k = 2 * (p - 25); k < 2 * (p - 25) + 2; ++k) trc[k] = __builtin_amdgcn_rcpf(1.0f + te[k]);
          } else if constexpr (p == 27 || p == 28) {
#pragma unroll
            for (int k = 2 * (p - 27); k < 2 * (p - 27) + 2; ++k)
              ty[k] = tanh_s4(comp(pre[2], k), tax[k], te[k], tsm[k], trc[k]);
          } else if constexpr (p == 29) {              // c_t = f * c_{t-1} + i * g
            cnv.x = cell_c(sy[4], creg[phf].x, sy[0], ty[0]);
            cnv.y = cell_c(sy[5], creg[phf].y, sy[1], ty[1]);
            cnv.z = cell_c(sy[6], creg[phf].z, sy[2], ty[2]);
            cnv.w = cell_c(sy[7], creg[phf].w, sy[3], ty[3]);
            creg[phf] = cnv;
          } else if constexpr (p == 30 || p == 31) {
#pragma unroll
            for (int k = 2 * (p - 30); k < 2 * (p - 30) + 2; ++k) tanh_s1(comp(cnv, k), cax[k], ce[k]);
          } else if constexpr (p >= 32 && p <= 35) {
            csm[p - 32] = tanh_s2(comp(cnv, p - 32), cax[p - 32]);
          } else if constexpr (p == 36 || p == 37) {
#pragma unroll
            for (int k = 2 * (p - 36); k < 2 * (p - 36) + 2; ++k) crc[k] = __builtin_amdgcn_rcpf(1.0f + ce[k]);
          } else if constexpr (p == 38 || p == 39) {
#pragma unroll
            for (int k = 2 * (p - 38); k < 2 * (p - 38) + 2; ++k) cy[k] = tanh_s4(comp(cnv, k), cax[k], ce[k], csm[k], crc[k]);
          } else if constexpr (p == 40) {              // h first and write-through: it is what the group waits for
            hv = make_float4(sy[8] * cy[0], sy[9] * cy[1], sy[10] * cy[2], sy[11] * cy[3]);
            store_sc1(yrs, oob(e_el, (e_el * (unsigned)ldy + (unsigned)(j0 + 4 * pq)) * 4u), hv);
          } else if constexpr (p == 41) {
            gbo = oob(e_el, (e_el * (unsigned)(4 * H) + (unsigned)(j0 + 4 * pq)) * 4u);
            store4(grs, gbo, 0u, make_float4(sy[0], sy[1], sy[2], sy[3]));
            store4(grs, gbo, (unsigned)(H * 4), make_float4(sy[4], sy[5], sy[6], sy[7]));
          } else if constexpr (p == 42) {
            store4(grs, gbo, (unsigned)(2 * H * 4), make_float4(ty[0], ty[1], ty[2], ty[3]));
            store4(grs, gbo, (unsigned)(3 * H * 4), make_float4(sy[8], sy[9], sy[10], sy[11]));
          } else if constexpr (p == 43) {
            store4(crs, oob(e_el, (e_el * (unsigned)H + (unsigned)(j0 + 4 * pq)) * 4u), 0u, cnv);
          }
        };
        bf16x8 fa1[TERMS == 3 ? 3 : 1], wl1[4];
        static_for<NB1 * 24>([&](auto Q) {
          constexpr int q = decltype(Q)::value, b = q / 24, t6 = (q % 24) / 4, g = q % 4;
          if constexpr (q % 24 == 0) {                 // operands of block b
            const float4 a0 = *reinterpret_cast<const float4*>(asrc + 8 * b);
            const float4 a1 = *reinterpret_cast<const float4*>(asrc + 8 * b + 4);
            if constexpr (TERMS == 3) {
              split8(a0, a1, fa1);
#pragma unroll
              for (int gg = 0; gg < 4; ++gg)
                wl1[gg] = b < NBR ? bwlo[gg][b < NBR ? b : 0]
                                  : __builtin_bit_cast(bf16x8, wlo_lds[(gg * (NB - NBR) + (b - NBR)) * 256 + tid]);
            } else {
              fa1[0] = round8(a0, a1);
            }
          }
          if constexpr (TERMS == 3) {
            acc[g] = mfma_bf16(kTb[t6] == 2 ? wl1[g] : bwhm[g][b][kTb[t6] == 2 ? 0 : kTb[t6]], fa1[kTa[t6]], acc[g]);
          } else {
            if constexpr (t6 == 0) acc[g] = mfma_bf16(bwhm[g][b][0], fa1[0], acc[g]);
          }
          constexpr int gap = q - (q / 24 + 1);        // gaps that carry no operand preparation, numbered from 0
          if constexpr (q % 24 != 0 && gap < kEPieces) e_piece(std::integral_constant<int, gap>{});
          __builtin_amdgcn_sched_barrier(0);
        });
      } else {
#pragma unroll
        for (int b = 0; b < NB1; ++b) mfma_block(b);
        cell_update(pstep, phf, xp_prev);
      }
      PE_STAMP(0)                                                 // region 1: MFMAs + pending epilogue issued
      asm volatile("s_waitcnt vmcnt(5)" ::: "memory");            // the h store (issued first) has completed
      PE_STAMP(1)                                                 // ... its store drain
      __syncthreads();                                            // #2
      PE_STAMP(2)
      arrive(phf);
      // ---- region 2: second part of the MFMAs; one lane waits for the next item's group
#pragma unroll
      for (int b = NB1; b < NB - 1; ++b) mfma_block(b);
      const int nstep = hf ? step + 1 : step, nhf = hf ^ 1;       // item i+1
      const bool want_fetch = nstep < T;
      PE_STAMP(3)                                                 // region 2 MFMAs issued
      if (want_fetch) poll(nhf, (unsigned)(NJ * nstep));
      PE_STAMP(4)                                                 // poll (wave 0)
      __syncthreads();                                            // #3: the poll result reaches every wave
      PE_STAMP(5)
      auto red_store = [&](int g, int q4) {
        *reinterpret_cast<float4*>(red + (wv * 32 + r) * kRs + g * 32 + 8 * q4 + 4 * hh) =
            make_float4(acc[g][4 * q4], acc[g][4 * q4 + 1], acc[g][4 * q4 + 2], acc[g][4 * q4 + 3]);
      };
      if constexpr (TERMS == 3 && NST <= 12) {
        // last block gate by gate (each accumulator's six products keep their order): the h-row requests of the
        // next item ride in the odd MFMA gaps, a finished gate's partial tile goes to LDS in the even gaps of the
        // gates after it -- instead of a burst of requests before and a burst of stores after the 24 MFMAs
        const int t_n = rev ? T - 1 - nstep : nstep, tp_n = rev ? t_n + 1 : t_n - 1;
        const int brow_n = b0 + 32 * nhf + (tid >> 3);
        const bool ok_n = want_fetch && brow_n < B;
        const unsigned base_n = ok_n ? ((unsigned)(brow_n * T + tp_n) * (unsigned)ldy + (unsigned)(tid & 7) * 4u) * 4u
                                     : 0xfffffff0u;
        bf16x8 fa3[3], wl3[4];
        {
          constexpr int b = NB - 1;
          const float4 a0 = *reinterpret_cast<const float4*>(asrc + 8 * b);
          const float4 a1 = *reinterpret_cast<const float4*>(asrc + 8 * b + 4);
          split8(a0, a1, fa3);
#pragma unroll
          for (int gg = 0; gg < 4; ++gg)
            wl3[gg] = b < NBR ? bwlo[gg][b < NBR ? b : 0]
                              : __builtin_bit_cast(bf16x8, wlo_lds[(gg * (NB - NBR) + (b - NBR)) * 256 + tid]);
        }
        static_for<24>([&](auto Q) {
          constexpr int q = decltype(Q)::value, g = q / 6, t6 = q % 6, b = NB - 1;
          acc[g] = mfma_bf16(kTb[t6] == 2 ? wl3[g] : bwhm[g][b][kTb[t6] == 2 ? 0 : kTb[t6]], fa3[kTa[t6]], acc[g]);
          if constexpr ((q & 1) == 1 && q / 2 < NST)
            stage[q / 2] = load_sc1(yrs, base_n + (ok_n ? (unsigned)(q / 2) * 128u : 0u));
          if constexpr ((q & 1) == 0 && q >= 6) {    // even gaps: gate (q - 6) / 8's tile, one float4 per gap
            constexpr int e = (q - 6) / 2;           // 0 .. 8
            if constexpr (e / 4 < g) red_store(e / 4, e % 4);
          }
          __builtin_amdgcn_sched_barrier(0);
        });
        // what the gaps did not cover: gate 2's last float4s and gate 3
#pragma unroll
        for (int e = 9; e < 16; ++e) red_store(e / 4, e % 4);
      } else {
        if (want_fetch) fetch_rows(nstep, nhf);
        mfma_block(NB - 1);
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) red_store(g, q4);
      }
      PE_STAMP(6)                                                 // last block + accumulators -> red
      __syncthreads();                                            // #0: As(i) is free, red(i) complete
      PE_STAMP(7)
      if (want_fetch) commit_rows();
#pragma unroll
      for (int g = 0; g < 4; ++g) xp_prev[g] = xp_cur[g];
      PE_STAMP(8)                                                 // wait for the fetched rows + LDS writes
      __syncthreads();                                            // #1
      PE_STAMP(9)
    }
  }
  cell_update(T - 1, 1, xp_prev);
  if constexpr (STAMP) {
    if (tid == 0)
      for (int k = 0; k < 10; ++k) sync[2048 + 16 * blockIdx.x + k] = (unsigned)(st_acc[k] >> 4);
  }
}

// --------------------------------------------------------------------------------------- backward, k-split
// dh_t = dY_t + dgates_{t+1} . W_hh.  The kernel above gives workgroup jt the 32 output columns j of dh and has it
// read ALL 4H gate gradients of its 32 batch rows: 192 KB per item and workgroup through one CU's 64 B/clk L1 path,
// twelve times over per group -- s_memtime stamps put ~16 k of an item's 20 k cycles on those loads (tools/
// stamp_lstm.py), the 144 MFMAs need 4.6 k.  This kernel splits the product over k instead: workgroup jt PRODUCES
// the 128 gate gradients of its own hidden slice, so it multiplies exactly those -- they never leave the CU (LDS) --
// with the matching 128 rows of W_hh for ALL H output columns, and hands each of the NJ consumers a 32 x 32 fp32
// partial tile.  Per item a workgroup stores 4 KB to and loads 4 KB from each peer: 48 KB each way at H = 384, a
// quarter of the read volume, and every transfer is a full-line 1 KB instruction.  Same MFMA count, same registers
// for W.  dh sums the NJ partial tiles in producer order (fixed, so runs are reproducible).
//   exchange region: [cell][batch tile][half][slot = step & 1][consumer][producer][i = col / 8][32 rows][8 cols]
//   (two slots: a producer can only write step s + 2 after every peer has arrived at s + 1, i.e. has consumed s).
// The MFMA operands are swapped (A = W_hh^T block, B = gate gradients) so that a lane ends up with 4 consecutive
// COLUMNS of one batch row -- the float4 the consumer's gate-gradient update wants -- instead of 4 rows.
// Schedule per workgroup (M = product + hand-off of a half, E = gate-gradient update of a half):
//   E(0,0) E(0,1) M(1,0) | M(s,1) E(s,0) M(s+1,0) E(s,1) | ...   the wait for the peers' tiles of one half sits in
// the middle of the other half's M, the tile loads land under the rest of it.
constexpr int kXchgWord = 8192;          // flag blocks start this many words into the sync buffer ...
constexpr int kFlagWords = 16384;        // ... [cell][batch tile][half][64 words]; the partial tiles follow

template <int H, int TERMS, int NBR_, bool STAMP = false>
__global__ __launch_bounds__(256, 1) void lstm_bwd_persistent_ks_kernel(const PBwdCells cells, int B, int T, long lddy,
                                                                        unsigned g_bytes, unsigned* sync) {
  static_assert(TERMS == 3 || TERMS == 1, "bf16-term pipelines only");
  static_assert(H % 128 == 0, "each wave owns whole 32-column blocks of dh");
  constexpr int NJ = H / 32, K = 4 * H;
  constexpr int NBW = H / 128;                      // 32-column blocks of dh per wave
  constexpr int NKB = 8;                            // 16-k MFMA blocks over the 128 local k
  constexpr int NBK = NBW * NKB;                    // W blocks per lane
  constexpr int NBR = TERMS == 3 ? NBR_ : NBK;
  constexpr unsigned TILE = 1024;                   // elements per 32 x 32 partial tile
  constexpr unsigned SLOT = NJ * NJ * TILE;
  constexpr unsigned XB = TERMS == 1 ? 2u : 4u;      // bytes per exchanged tile element: bf16 in the one-term build
  auto xstore = [&](__amdgpu_buffer_rsrc_t rs, unsigned off, float4 v) {
    if constexpr (TERMS == 1) store_sc1_h(rs, off, v);
    else store_sc1(rs, off, v);
  };
  auto xload = [&](__amdgpu_buffer_rsrc_t rs, unsigned off) -> float4 {
    if constexpr (TERMS == 1) return load_sc1_h(rs, off);
    else return load_sc1(rs, off);
  };
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NT = TERMS == 3 ? 3 : 1;
  constexpr int DGH = NT * NKB * 2 * 32;            // uint4 per half
  uint4* dg = reinterpret_cast<uint4*>(smem);       // [2 halves][term][NKB][2][32 rows] x 8 bf16: this slice's
                                                    // gate gradients, already split, in B-fragment order
  uint4* wlo_lds = dg + 2 * DGH;                    // [NBK - NBR][256]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const int nbt = (B + 63) / 64;
  int jt, gidx;
  group_of_block(NJ, jt, gidx);
  const int bt = gidx % nbt, cell = gidx / nbt;
  const int j0 = jt * 32, b0 = bt * 64;
  const int rev = cells.reverse[cell];
  float* gates = cells.gates[cell];
  const float* cb = cells.c[cell];
  const float* dy = cells.dy[cell];
  unsigned* err = sync;
  static_assert(4 * NJ <= 64, "one flag word per producer wave, one load per consumer wave");
  unsigned* flags = sync + kXchgWord + (cell * nbt + bt) * 128;        // [half][64]
  float* xbase = reinterpret_cast<float*>(sync + kXchgWord + kFlagWords) + (size_t)(cell * nbt + bt) * 4 * SLOT;
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(xbase, 0, 4u * SLOT * 4u, 0x00020000);
  const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc(gates, 0, g_bytes, 0x00020000);

  // W block (nb, kb): column n = wv * H/4 + nb * 32 + r of dh, k = gate (kb >> 1), hidden j0 + (kb & 1) * 16 + hh * 8 ..
  bf16x8 bwhm[NBK][TERMS == 3 ? 2 : 1], bwlo[NBR > 0 ? NBR : 1];
#pragma unroll
  for (int nb = 0; nb < NBW; ++nb) {
    const float* src = cells.whh_t[cell] + (long)(wv * (H / 4) + nb * 32 + r) * K + j0 + hh * 8;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      const int b = nb * NKB + kb;
      const float* sp = src + (kb >> 1) * H + (kb & 1) * 16;
      const float4 w0 = *reinterpret_cast<const float4*>(sp);
      const float4 w1 = *reinterpret_cast<const float4*>(sp + 4);
      if constexpr (TERMS == 3) {
        bf16x8 t3[3];
        split8(w0, w1, t3);
        bwhm[b][0] = t3[0];
        bwhm[b][1] = t3[1];
        if (b < NBR) bwlo[b < NBR ? b : 0] = t3[2];
        else wlo_lds[(b - NBR) * 256 + tid] = __builtin_bit_cast(uint4, t3[2]);
      } else {
        bwhm[b][0] = round8(w0, w1);
      }
    }
  }
  const int prow = tid >> 3, pq = tid & 7;
  const int j = j0 + 4 * pq;
  float4 dcar[2];
  // bias gradient = column sums of the gate gradients: every thread keeps the running sums of its 4 gates x 4
  // columns over all steps and both halves; folded over the 32 rows of a half at the end of the kernel
  float4 bsum[4];
  float gmax = 0.f;                 // largest gate-gradient magnitude this thread produced: the "h2" scale source of
                                    // the dX / dW_ih / dW_hh products that read the gradient tensor
#pragma unroll
  for (int g = 0; g < 4; ++g) bsum[g] = make_float4(0.f, 0.f, 0.f, 0.f);
  dcar[0] = dcar[1] = make_float4(0.f, 0.f, 0.f, 0.f);
  unsigned long long st_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_last = 0;

  // Inputs of one gate-gradient update (single set: requested after the previous update has consumed its own).
  // Every request is ONE buffer instruction with a per-thread offset computed once and a scalar offset per step: at
  // one wave per SIMD a burst of loads with 64-bit address arithmetic in front stalls the wave on the texture
  // addresser's queue (s_memtime: ~2 000 cycles per item), so the product below also spreads these requests
  // between its MFMA groups.  Rows past B point out of the descriptors' range and read as zero.
  constexpr unsigned kOob = 0x80000000u;            // host: every tensor of a cell is smaller than 2 GiB
  const __amdgpu_buffer_rsrc_t crs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(cb), 0, (unsigned)B * T * H * 4u, 0x00020000);
  const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(dy), 0, (unsigned)(((long)B * T - 1) * lddy + H) * 4u, 0x00020000);
  unsigned vg[2], vc[2], vd[2];
#pragma unroll
  for (int hf = 0; hf < 2; ++hf) {
    const int pb = b0 + 32 * hf + prow;
    const bool ok = pb < B;
    vg[hf] = ok ? ((unsigned)(pb * T) * K + j) * 4u : kOob;
    vc[hf] = ok ? ((unsigned)(pb * T) * H + j) * 4u : kOob;
    vd[hf] = ok ? ((unsigned)(pb * T) * (unsigned)lddy + j) * 4u : kOob;
  }
  float4 in_dy, in_c, in_cp, in_g[4], part[NJ], ccar[2];
  auto ld4 = [&](__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
    const u32x4 d = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
    return make_float4(__uint_as_float(d.x), __uint_as_float(d.y), __uint_as_float(d.z), __uint_as_float(d.w));
  };
  constexpr int kInputs = 6;                        // dY, four gates, c of the previous time step
  auto issue_input = [&](int which, int step, int hf) {
    const int t = rev ? step : T - 1 - step;
    const int tp = rev ? t + 1 : t - 1;
    const bool has_prev = rev ? (tp < T) : (tp >= 0);
    if (which == 0) in_dy = ld4(drs, vd[hf], (unsigned)t * (unsigned)lddy * 4u);
    else if (which <= 4) in_g[which - 1] = ld4(grs, vg[hf], ((unsigned)t * K + (unsigned)(which - 1) * H) * 4u);
    else in_cp = ld4(crs, has_prev ? vc[hf] : kOob, (unsigned)(has_prev ? tp : 0) * H * 4u);
  };
  // c_t of a step is c_{t-1} of the step before it: only step 0 fetches it, later steps inherit (ccar)
  auto issue_inputs_all = [&](int step, int hf) {
#pragma unroll
    for (int w = 0; w < kInputs; ++w) issue_input(w, step, hf);
    if (step == 0) in_c = ld4(crs, vc[hf], (unsigned)(rev ? 0 : T - 1) * H * 4u);
  };
  // the NJ tiles addressed to this workgroup: tile p of (step, hf); group_wait(step, hf) must have returned
  auto tile_base = [&](int step, int hf) {
    return (((unsigned)(hf * 2 + (step & 1)) * NJ * NJ + (unsigned)jt * NJ) * TILE + (unsigned)(pq >> 1) * 256u +
            (unsigned)prow * 8u + (unsigned)(pq & 1) * 4u) * XB;
  };
  auto wait_peers = [&](int step, int hf, unsigned seen) {
    // No workgroup barrier: this slice's dg[hf] is protected by the barrier that ends the gate update between the
    // product that read it and this wait.
    flags_wait(flags + 64 * hf, 4 * NJ, (unsigned)step, seen, err);
  };

  // E: gate gradients of (step, hf) from dh = dY + sum of the peers' partial tiles
  auto gate_update = [&](int step, auto HF, bool with_part, auto&& handoff) {
    constexpr int hf = decltype(HF)::value;
    const int t = rev ? step : T - 1 - step;
    float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (with_part) {
#pragma unroll
      for (int p = 0; p < NJ; ++p) { s4.x += part[p].x; s4.y += part[p].y; s4.z += part[p].z; s4.w += part[p].w; }
    }
    const float dhv[4] = {in_dy.x + s4.x, in_dy.y + s4.y, in_dy.z + s4.z, in_dy.w + s4.w};
    const float gi[4] = {in_g[0].x, in_g[0].y, in_g[0].z, in_g[0].w};
    const float gf[4] = {in_g[1].x, in_g[1].y, in_g[1].z, in_g[1].w};
    const float gg[4] = {in_g[2].x, in_g[2].y, in_g[2].z, in_g[2].w};
    const float go[4] = {in_g[3].x, in_g[3].y, in_g[3].z, in_g[3].w};
    const float4 c_now = step == 0 ? in_c : ccar[hf];
    ccar[hf] = in_cp;
    const float cn[4] = {c_now.x, c_now.y, c_now.z, c_now.w};
    const float cp[4] = {in_cp.x, in_cp.y, in_cp.z, in_cp.w};
    float dcv[4] = {dcar[hf].x, dcar[hf].y, dcar[hf].z, dcar[hf].w};
    float tc[4], oi[4], of[4], og[4], oo[4];
    tanh_n<4>(cn, tc);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float dc = dhv[e] * go[e] * (1.f - tc[e] * tc[e]) + dcv[e];
      oi[e] = dc * gg[e] * gi[e] * (1.f - gi[e]);
      of[e] = dc * cp[e] * gf[e] * (1.f - gf[e]);
      og[e] = dc * gi[e] * (1.f - gg[e] * gg[e]);
      oo[e] = dhv[e] * tc[e] * go[e] * (1.f - go[e]);
      dcv[e] = dc * gf[e];
    }
    dcar[hf] = make_float4(dcv[0], dcv[1], dcv[2], dcv[3]);
    const float4 og4[4] = {make_float4(oi[0], oi[1], oi[2], oi[3]), make_float4(of[0], of[1], of[2], of[3]),
                           make_float4(og[0], og[1], og[2], og[3]), make_float4(oo[0], oo[1], oo[2], oo[3])};
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      bsum[g].x += og4[g].x; bsum[g].y += og4[g].y; bsum[g].z += og4[g].z; bsum[g].w += og4[g].w;
      gmax = fmaxf(gmax, fmaxf(fmaxf(fabsf(og4[g].x), fabsf(og4[g].y)), fmaxf(fabsf(og4[g].z), fabsf(og4[g].w))));
    }
    // local k = g * 32 + 4 pq + e  ->  block kb = 2 g + (pq >> 2), lane half (pq >> 1) & 1, position (pq & 1) * 4 + e;
    // split here, once, instead of in each of the four product waves
    unsigned char* dl = reinterpret_cast<unsigned char*>(dg + hf * DGH + ((pq >> 2) * 2 + ((pq >> 1) & 1)) * 32 + prow) +
                        (pq & 1) * 8;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if constexpr (TERMS == 3) {
        u32x2 t3[3];
        split4(og4[g], t3);
#pragma unroll
        for (int t = 0; t < 3; ++t)
          *reinterpret_cast<u32x2*>(dl + (size_t)((t * NKB + 2 * g) * 2 * 32) * 16) = t3[t];
      } else {
        *reinterpret_cast<u32x2*>(dl + (size_t)(2 * g * 2 * 32) * 16) = round4(og4[g]);
      }
    }
    // the gradient tensor itself (input of the dW / dX GEMMs): nobody in this launch waits for these stores
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const u32x4 d = {__float_as_uint(og4[g].x), __float_as_uint(og4[g].y), __float_as_uint(og4[g].z),
                       __float_as_uint(og4[g].w)};
      __builtin_amdgcn_raw_buffer_store_b128(d, grs, vg[hf], ((unsigned)t * K + (unsigned)g * H) * 4u, 0);
    }
    handoff();                                        // the preceding product's tiles (four younger stores above)
    __syncthreads();                                  // dg[hf] complete before the next product reads it
  };

  // M: partial tiles of (step, hf) = this slice's gate gradients of step - 1 times its 128 rows of W_hh, handed off.
  // The requests of the pending update (pre_step, pre_hf) ride between the MFMA groups, one or two per group: its
  // inputs from the start, the wait for the peers in the middle, their tiles right behind it -- early enough that
  // they have landed when the last tile store is drained.  A column block's tile stores likewise go out one per
  // group of the next block.
  // MFMA group in front of which the wait sits.  One-term build: a group is ONE MFMA, the whole product is shorter than
  // the hand-off it should cover (stamps: 2.6 k of an item's 7.6 k cycles in the poll), so the wait goes as late as the
  // tile requests behind it allow (16 -> 22 of 24: backward 1.73 -> 1.64 ms per layer, interleaved on two boxes)
  constexpr int LPOLL = TERMS == 1 ? NBK - 2 : NBW >= 3 ? (2 * NBK) / 3 : (NBW / 2) * NKB + NKB / 2;
  constexpr int LPEEK = LPOLL >= 4 ? LPOLL - 4 : 0;
  constexpr int LSPAN = (NBK - LPOLL) / 2 > 0 ? (NBK - LPOLL) / 2 : 1;  // groups that carry tile requests
  constexpr int LPER = (NJ + LSPAN - 1) / LSPAN;                        // tile requests per group
  auto product = [&](int step, auto HF, int pre_step, int pre_hf, bool pre, bool pre_part) {
    constexpr int hf = decltype(HF)::value;
    const uint4* ap = dg + hf * DGH + hh * 32 + r;
    // lane (r, hh) ends up with, of batch row r, columns 8 i + 4 hh .. + 3 of consumer block wv * NBW + nb
    const unsigned xo = (((unsigned)(hf * 2 + (step & 1)) * NJ * NJ + (unsigned)jt) * TILE + (unsigned)r * 8u +
                         (unsigned)hh * 4u) * XB;
    const unsigned tb = tile_base(pre_step, pre_hf);
    // two accumulator pairs (even / odd k groups), alternating per column block: the finished block's tile is summed
    // and stored piecewise in the next block's MFMA gaps instead of in one burst between the blocks
    f32x16 accs[2][2];
    auto tile_piece = [&](const f32x16& a, const f32x16& b, int i) {
      return make_float4(a[4 * i] + b[4 * i], a[4 * i + 1] + b[4 * i + 1], a[4 * i + 2] + b[4 * i + 2],
                         a[4 * i + 3] + b[4 * i + 3]);
    };
    unsigned seen = 0xffffffffu;
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) {
      f32x16& acc = accs[nb & 1][0];
      f32x16& acc2 = accs[nb & 1][1];
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[q] = acc2[q] = 0.f;
      // operands of group kb + 1 are requested before group kb multiplies; the scheduling barriers keep hipcc from
      // sinking those reads down to their use (it does, and every product then waits out an LDS round trip)
      uint4 fn[NT], wn = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
      for (int t = 0; t < NT; ++t) fn[t] = ap[t * NKB * 64];
      if constexpr (TERMS == 3) {
        if (nb * NKB >= NBR) wn = wlo_lds[(nb * NKB - NBR) * 256 + tid];
      }
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) {
        const int b = nb * NKB + kb;
        bf16x8 fa[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) fa[t] = __builtin_bit_cast(bf16x8, fn[t]);
        const bf16x8 wl = b < NBR ? bwlo[b < NBR ? b : 0] : __builtin_bit_cast(bf16x8, wn);
        // a poll is an L2 round trip (~700 cycles) even when the peers have long arrived: the first one is sent
        // four groups early and only looked at here
        if (b == LPEEK && pre && pre_part) seen = flags_peek(flags + 64 * pre_hf, 4 * NJ);
        if (b == LPOLL) {
          PE_STAMP(0)                                           // first half of the product
          if constexpr (STAMP) {
            if (pre_step == 100 && pre_hf == 0 && tid == 0) sync[2048 + 16 * blockIdx.x + 11] = (unsigned)__builtin_amdgcn_s_memrealtime();
          }
          if (pre && pre_part) wait_peers(pre_step, pre_hf, seen);
          if constexpr (STAMP) {
            if (pre_step == 100 && pre_hf == 0 && tid == 0) sync[2048 + 16 * blockIdx.x + 12] = (unsigned)__builtin_amdgcn_s_memrealtime();
          }
          PE_STAMP(1)                                           // poll
        }
        if (kb + 1 < NKB) {
#pragma unroll
          for (int t = 0; t < NT; ++t) fn[t] = ap[(t * NKB + kb + 1) * 64];
          if constexpr (TERMS == 3) {
            if (b + 1 >= NBR) wn = wlo_lds[(b + 1 - NBR) * 256 + tid];
          }
        }
        if (pre && b < kInputs) issue_input(b, pre_step, pre_hf);
        if (pre && pre_part && b >= LPOLL) {
#pragma unroll
          for (int u = 0; u < LPER; ++u) {
            const int p = (b - LPOLL) * LPER + u;
            if (p < NJ) part[p < NJ ? p : 0] = xload(xrs, tb + (unsigned)p * TILE * XB);
          }
        }
        if (nb > 0 && kb < 4)
          xstore(xrs, xo + ((unsigned)((wv * NBW + nb - 1) * NJ) * TILE + (unsigned)kb * 256u) * XB,
                 tile_piece(accs[(nb - 1) & 1][0], accs[(nb - 1) & 1][1], kb));
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (TERMS == 3) {
#pragma unroll
          for (int t6 = 0; t6 < 6; ++t6) {
            const bf16x8 wt = kTb[t6] == 2 ? wl : bwhm[b][kTb[t6]];
            if (t6 & 1) acc2 = mfma_bf16(wt, fa[kTa[t6]], acc2);
            else acc = mfma_bf16(wt, fa[kTa[t6]], acc);
          }
        } else {
          if (kb & 1) acc2 = mfma_bf16(bwhm[b][0], fa[0], acc2);
          else acc = mfma_bf16(bwhm[b][0], fa[0], acc);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
      xstore(xrs, xo + ((unsigned)((wv * NBW + NBW - 1) * NJ) * TILE + (unsigned)i * 256u) * XB,
             tile_piece(accs[(NBW - 1) & 1][0], accs[(NBW - 1) & 1][1], i));
    PE_STAMP(2)                                                   // second half of the product, tile stores issued
  };
  // Hand-off of a product's tiles: each wave drains its own stores and arrives (no workgroup barrier: the next writer
  // of dg[hf] sits behind the barrier that ends this very gate update).  Called at the END of the gate
  // update that follows the product -- the peers need these tiles a whole product later, and by then the stores
  // have long completed, so the wave does not idle on the drain; `younger` = VMEM instructions issued since.
  auto arrive = [&](int step, int hf, auto YOUNGER) {
    constexpr int younger = decltype(YOUNGER)::value;
    if constexpr (younger == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PE_STAMP(3)                                                   // tile store drain
    if (lane == 0)
      __hip_atomic_store(flags + 64 * hf + 4 * jt + wv, (unsigned)step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if constexpr (STAMP) {
      if (step == 100 && hf == 0 && tid == 0) sync[2048 + 16 * blockIdx.x + 10] = (unsigned)__builtin_readcyclecounter();
      if (step == 100 && hf == 0 && tid == 0) sync[2048 + 16 * blockIdx.x + 13] = (unsigned)__builtin_amdgcn_s_memrealtime();
    }
    PE_STAMP(4)                                                   // arrive
  };

  using H0 = std::integral_constant<int, 0>;
  using H1 = std::integral_constant<int, 1>;
  using Y0 = std::integral_constant<int, 0>;
  using Y4 = std::integral_constant<int, 4>;
  auto none = [] {};
  issue_inputs_all(0, 0);
  gate_update(0, H0{}, false, none);
  issue_inputs_all(0, 1);
  gate_update(0, H1{}, false, none);
  if constexpr (STAMP) st_last = stamp_now();
  if (T > 1) {
    product(1, H0{}, 0, 0, false, false);
    arrive(1, 0, Y0{});
  }
  for (int step = 1; step < T; ++step) {
    product(step, H1{}, step, 0, true, true);
    gate_update(step, H0{}, true, [&] { arrive(step, 1, Y4{}); });
    PE_STAMP(5)                                                   // gate-gradient update + barrier
    if (step + 1 < T) {
      product(step + 1, H0{}, step, 1, true, true);
      gate_update(step, H1{}, true, [&] { arrive(step + 1, 0, Y4{}); });
    } else {
      issue_inputs_all(step, 1);
      wait_peers(step, 1, flags_peek(flags + 64, 4 * NJ));
#pragma unroll
      for (int p = 0; p < NJ; ++p) part[p] = xload(xrs, tile_base(step, 1) + (unsigned)p * TILE * XB);
      gate_update(step, H1{}, true, none);
    }
    PE_STAMP(5)
  }
  if (cells.dbias[cell] != nullptr) {                // (every gate update ended with a barrier: dg is free)
    float4* fold = reinterpret_cast<float4*>(smem);  // [32 rows][8 column quads][4 gates]
#pragma unroll
    for (int g = 0; g < 4; ++g) fold[(prow * 8 + pq) * 4 + g] = bsum[g];
    __syncthreads();
    if (tid < 128) {                                 // column quad tid >> 4, gate (tid >> 2) & 3, element tid & 3
      const float* f = reinterpret_cast<const float*>(fold) + tid;
      float sum = 0.f;
#pragma unroll 8
      for (int row = 0; row < 32; ++row) sum += f[row * 128];
      const int g = (tid >> 2) & 3;
      cells.dbias[cell][(long)bt * K + g * H + j0 + 4 * (tid >> 4) + (tid & 3)] = sum;
    }
  }
  if (cells.amax[cell] != nullptr) {                 // one no-return atomicMax per wave on the cell's word (bit patterns
#pragma unroll                                         // of non-negative floats order like unsigned integers)
    for (int off = 32; off > 0; off >>= 1) gmax = fmaxf(gmax, __shfl_xor(gmax, off, 64));
    if (lane == 0) atomicMax(cells.amax[cell], __float_as_uint(gmax));
  }
  if constexpr (STAMP) {
    if (tid == 0)
      for (int k = 0; k < 10; ++k) sync[2048 + 16 * blockIdx.x + k] = (unsigned)(st_acc[k] >> 4);
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

template <int H, int TERMS, int NBR>
constexpr size_t fwd_v2_lds() {
  constexpr int NB = H / 64, NBL = TERMS == 3 ? NB - NBR : 0;
  return (size_t)(32 * (H + 4) + 4 * 32 * kRs) * sizeof(float) + (size_t)4 * NBL * 256 * 16;
}

template <int H, int TERMS, int NBR, bool STAMP = false>
int launch_fwd_v2(const PFwdCells& cells, int grid, int B, int T, long ldy, unsigned* sync, hipStream_t st) {
  static_assert(fwd_v2_lds<H, TERMS, NBR>() <= 160 * 1024, "LDS budget");
  static bool attr = false;
  if (!attr) {
    PE_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&lstm_fwd_persistent_v2_kernel<H, TERMS, NBR, STAMP>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)fwd_v2_lds<H, TERMS, NBR>()));
    attr = true;
  }
  const unsigned y_bytes = (unsigned)((size_t)B * T * ldy * sizeof(float));
  const unsigned g_bytes = (unsigned)((size_t)B * T * 4 * H * sizeof(float));
  const unsigned c_bytes = (unsigned)((size_t)B * T * H * sizeof(float));
  hipLaunchKernelGGL((lstm_fwd_persistent_v2_kernel<H, TERMS, NBR, STAMP>), dim3(grid), dim3(256),
                     (fwd_v2_lds<H, TERMS, NBR>()), st, cells, B, T, ldy, y_bytes, g_bytes, c_bytes, sync);
  PE_LAUNCH_CHECK();
  return PE_OK;
}

template <int H, int TERMS, int NBR>
constexpr size_t bwd_v2_lds() {
  constexpr int NBK = (H / 128) * 8, NBL = TERMS == 3 ? NBK - NBR : 0;
  return (size_t)(2 * (TERMS == 3 ? 3 : 1) * 8 * 2 * 32) * 16 + (size_t)NBL * 256 * 16;
}

template <int H, int TERMS, int NBR, bool STAMP = false>
int launch_bwd_v2(const PBwdCells& cells, int grid, int B, int T, long lddy, unsigned* sync, hipStream_t st) {
  static_assert(bwd_v2_lds<H, TERMS, NBR>() <= 160 * 1024, "LDS budget");
  static bool attr = false;
  if (!attr) {
    PE_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&lstm_bwd_persistent_ks_kernel<H, TERMS, NBR, STAMP>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)bwd_v2_lds<H, TERMS, NBR>()));
    attr = true;
  }
  const unsigned g_bytes = (unsigned)((size_t)B * T * 4 * H * sizeof(float));
  hipLaunchKernelGGL((lstm_bwd_persistent_ks_kernel<H, TERMS, NBR, STAMP>), dim3(grid), dim3(256),
                     (bwd_v2_lds<H, TERMS, NBR>()), st, cells, B, T, lddy, g_bytes, sync);
  PE_LAUNCH_CHECK();
  return PE_OK;
}

// diagnostic instantiation with s_memtime stamps (tools/stamp_lstm.py): selected by pe_lstm_configure_stamps, never
// by the environment
static bool g_lstm_stamps = false;

int sync_words(int ncells, int B) { return kCtrStride * (1 + 2 * ncells * ((B + 63) / 64)); }

// exchange region of the k-split backward kernel (partial dh tiles), sized for H <= 384
size_t xchg_bytes(int ncells, int B) { return (size_t)ncells * ((B + 63) / 64) * 4 * (12 * 12 * 1024) * sizeof(float); }

}  // namespace

#ifndef PE_F16_BUILD
// Bytes of the zero-initialised buffer every persistent launch takes as `sync`: [error word + group counters | pad to
// kXchgWord words | the k-split backward kernel's flag blocks | its partial tiles].  Counters and flags are reset per
// launch.
extern "C" size_t pe_lstm_persistent_sync_bytes(int ncells, int B) {
  if (sync_words(ncells, B) > kXchgWord) return 0;
  if (ncells * ((B + 63) / 64) * 128 > kFlagWords) return 0;
  return (size_t)(kXchgWord + kFlagWords) * sizeof(unsigned) + xchg_bytes(ncells, B);
}
#endif

// 1 if the persistent kernels can run this shape on the current device: the hidden size they are instantiated for
// (H = 384, the reference's default, model.py:198), every tensor of a cell addressable with 32-bit offsets below 2 GiB,
// the flag blocks within the sync buffer, and the whole grid co-resident at one workgroup per CU; else 0 (run the
// one-launch-per-time-step kernels of lstm.hip).  The recurrent products are 16-bit-term MFMAs (three exact bf16 terms
// or one rounded term): there is no native-fp32 persistent form.
#ifndef PE_F16_BUILD
extern "C" int pe_lstm_persistent_supported(int ncells, int B, int H) {
  if (ncells < 1 || ncells > kMaxCells || B <= 0 || H != 384) return 0;
  if (sync_words(ncells, B) > kXchgWord || ncells * ((B + 63) / 64) * 128 > kFlagWords) return 0;
  const int grid = ncells * ((B + 63) / 64) * (H / 32);
  return grid <= device_cus() ? 1 : 0;
}

// Diagnostic: 1 = run the stamped instantiations (s_memtime per region, tools/stamp_lstm.py) where they exist (x3,
// grid <= 128 workgroups).  Returns the previous setting.  Not used by the product path.
extern "C" int pe_lstm_configure_stamps(int enable) {
  const int old = g_lstm_stamps ? 1 : 0;
  g_lstm_stamps = enable != 0;
  return old;
}
#else
extern "C" int pe_lstm_persistent_supported(int ncells, int B, int H);
#endif

static bool small_enough(int B, int T, int H, long ld) {
  return (size_t)B * T * 4 * H * sizeof(float) < (1ull << 31) && (size_t)B * T * (size_t)ld * sizeof(float) < (1ull << 31);
}

static int lstm_fwd_persistent_impl(int terms, int ncells, const float* const* whh, float* const* gates,
                                    float* const* y, float* const* cbuf, const int* reverse, long ldy, int B, int T,
                                    int H, unsigned* sync, void* stream) {
  if (!whh || !gates || !y || !cbuf || !reverse || !sync || T <= 0) return PE_E_ARG;
  if (!pe_lstm_persistent_supported(ncells, B, H) || (ldy & 3) || !small_enough(B, T, H, ldy)) return PE_E_UNSUPPORTED;
  PFwdCells cells{};
  for (int i = 0; i < ncells; ++i) {
    if (!whh[i] || !gates[i] || !y[i] || !cbuf[i]) return PE_E_ARG;
    cells.whh[i] = whh[i]; cells.gates[i] = gates[i]; cells.y[i] = y[i]; cells.c[i] = cbuf[i];
    cells.reverse[i] = reverse[i];
  }
  hipStream_t st = pe_stream(stream);
  // word 0 is the sticky error flag (cleared only by the owner of the buffer); counters start at line 1
  PE_CHECK_HIP(hipMemsetAsync(sync + kCtrStride, 0, (size_t)(sync_words(ncells, B) - kCtrStride) * 4, st));
  const int grid = ncells * ((B + 63) / 64) * (H / 32);
  if (terms == 3 && g_lstm_stamps && grid <= 128) return launch_fwd_v2<384, 3, 4, true>(cells, grid, B, T, ldy, sync, st);
  if (terms == 1 && g_lstm_stamps && grid <= 128) return launch_fwd_v2<384, 1, 6, true>(cells, grid, B, T, ldy, sync, st);
  return terms == 3 ? launch_fwd_v2<384, 3, 4>(cells, grid, B, T, ldy, sync, st)
                    : launch_fwd_v2<384, 1, 6>(cells, grid, B, T, ldy, sync, st);
}

#ifndef PE_F16_BUILD
extern "C" int pe_lstm_fwd_persistent_x3(int ncells, const float* const* whh, float* const* gates, float* const* y,
                                         float* const* cbuf, const int* reverse, long ldy, int B, int T, int H,
                                         unsigned* sync, void* stream) {
  return lstm_fwd_persistent_impl(3, ncells, whh, gates, y, cbuf, reverse, ldy, B, T, H, sync, stream);
}
#endif

static int lstm_bwd_persistent_impl(int terms, int ncells, const float* const* whh_t, float* const* gates,
                                    const float* const* cbuf, const float* const* dy, const int* reverse, long lddy,
                                    int B, int T, int H, float* const* dbias_rows, unsigned* const* dgates_amax,
                                    unsigned* sync, void* stream) {
  if (!whh_t || !gates || !cbuf || !dy || !reverse || !sync || T <= 0) return PE_E_ARG;
  if (!pe_lstm_persistent_supported(ncells, B, H) || (lddy & 3) || !small_enough(B, T, H, lddy)) return PE_E_UNSUPPORTED;
  PBwdCells cells{};
  for (int i = 0; i < ncells; ++i) {
    if (!whh_t[i] || !gates[i] || !cbuf[i] || !dy[i]) return PE_E_ARG;
    cells.whh_t[i] = whh_t[i]; cells.gates[i] = gates[i]; cells.c[i] = cbuf[i]; cells.dy[i] = dy[i];
    cells.reverse[i] = reverse[i];
    cells.dbias[i] = dbias_rows ? dbias_rows[i] : nullptr;
    cells.amax[i] = dgates_amax ? dgates_amax[i] : nullptr;
  }
  hipStream_t st = pe_stream(stream);
  PE_CHECK_HIP(hipMemsetAsync(sync + kCtrStride, 0, (size_t)(sync_words(ncells, B) - kCtrStride) * 4, st));
  PE_CHECK_HIP(hipMemsetAsync(sync + kXchgWord, 0, (size_t)ncells * ((B + 63) / 64) * 128 * sizeof(unsigned), st));
  const int grid = ncells * ((B + 63) / 64) * (H / 32);
  if (terms == 3 && g_lstm_stamps && grid <= 128) return launch_bwd_v2<384, 3, 8, true>(cells, grid, B, T, lddy, sync, st);
  if (terms == 1 && g_lstm_stamps && grid <= 128) return launch_bwd_v2<384, 1, 24, true>(cells, grid, B, T, lddy, sync, st);
  return terms == 3 ? launch_bwd_v2<384, 3, 8>(cells, grid, B, T, lddy, sync, st)
                    : launch_bwd_v2<384, 1, 24>(cells, grid, B, T, lddy, sync, st);
}

#ifndef PE_F16_BUILD
// Rows ([ceil(B / 64)][4H] per cell) that pe_lstm_bwd_persistent_* writes into a non-null dbias_rows for this
// configuration; 0 = the persistent kernel does not serve it (run pe_lstm_bwd and pe_colsum).
extern "C" int pe_lstm_bwd_persistent_dbias_rows(int ncells, int B, int T, int H, long lddy) {
  if (!pe_lstm_persistent_supported(ncells, B, H) || (lddy & 3) || T <= 0 || !small_enough(B, T, H, lddy)) return 0;
  return (B + 63) / 64;
}

extern "C" int pe_lstm_bwd_persistent_x3(int ncells, const float* const* whh_t, float* const* gates,
                                         const float* const* cbuf, const float* const* dy, const int* reverse,
                                         long lddy, int B, int T, int H, float* const* dbias_rows,
                                         unsigned* const* dgates_amax, unsigned* sync, void* stream) {
  return lstm_bwd_persistent_impl(3, ncells, whh_t, gates, cbuf, dy, reverse, lddy, B, T, H, dbias_rows, dgates_amax, sync,
                                  stream);
}
#endif

extern "C" int PE_HALF(pe_lstm_fwd_persistent)(int ncells, const float* const* whh, float* const* gates, float* const* y,
                                           float* const* cbuf, const int* reverse, long ldy, int B, int T, int H,
                                           unsigned* sync, void* stream) {
  return lstm_fwd_persistent_impl(1, ncells, whh, gates, y, cbuf, reverse, ldy, B, T, H, sync, stream);
}

extern "C" int PE_HALF(pe_lstm_bwd_persistent)(int ncells, const float* const* whh_t, float* const* gates,
                                           const float* const* cbuf, const float* const* dy, const int* reverse,
                                           long lddy, int B, int T, int H, float* const* dbias_rows,
                                           unsigned* const* dgates_amax, unsigned* sync, void* stream) {
  return lstm_bwd_persistent_impl(1, ncells, whh_t, gates, cbuf, dy, reverse, lddy, B, T, H, dbias_rows, dgates_amax, sync,
                                  stream);
}
