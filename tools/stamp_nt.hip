// Where does a k-tile of the h2 NT GEMM main loop spend its time?  A copy of nt_mainloop_split (gemm_engine.h) with
// ablation switches -- drop the global loads, the LDS stores (and the split in front of them), the LDS fragment reads or
// the MFMAs -- timed with HIP events on the 128 x 192 tile, M = 49152, N = 768, K = 1536 (the step's largest NT shape).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o gpurun_bin/stamp_nt tools/stamp_nt.hip && gpurun_bin/stamp_nt
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../pitchextractor_amd/csrc/common.h"
#include "../pitchextractor_amd/csrc/gemm_engine.h"
using namespace pe;

#ifndef STAMP_NT
#define STAMP_NT 2          // operand terms: 2 = h2 (two fp16 terms, 3 MFMAs per block), 1 = one rounded bf16 term
#endif
enum { kFull = 0, kNoLoads = 1, kNoStage = 2, kNoFrag = 3, kNoMfma = 4, kNoSplit = 5, kSameTile = 6, kShape16 = 7 };

template <class TL, int AB, int OCC, int PAD = 0>
__global__ __launch_bounds__(256, OCC) void nt_kernel(RowLoader al, RowLoader bl, float* out, int ldc, int K, int tiles_m,
                                                      int tiles_n, float sa, float sb) {
  constexpr int NT = STAMP_NT;
  __shared__ __attribute__((aligned(16))) float As_f[TL::BM * (STAMP_NT * kBK / 2)];
  __shared__ __attribute__((aligned(16))) float Bs_f[TL::BN * (STAMP_NT * kBK / 2)];
  __shared__ float pad_lds[PAD + 1];
  if (PAD > 0 && K < 0) pad_lds[threadIdx.x] = 1.f;
  const int tile_id = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int m0 = (tile_id / tiles_n) * TL::BM, n0 = (tile_id % tiles_n) * TL::BN;
  al.init(m0);
  bl.init(n0);
  f32x16 acc[TL::TM][TL::TN];
  zero_acc<TL>(acc);
  constexpr int A_IMG = TL::BM * kBK, B_IMG = TL::BN * kBK;
  __bf16* As = reinterpret_cast<__bf16*>(As_f);
  __bf16* Bs = reinterpret_cast<__bf16*>(Bs_f);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wm = wv / TL::WAVES_N, wn = wv % TL::WAVES_N;
  const int r = lane & 31, h = lane >> 5;
  const int nk = (K + kBK - 1) / kBK;
  float4 ra[TL::A_LOADS], rb[TL::B_LOADS];
#pragma unroll
  for (int i = 0; i < TL::A_LOADS; ++i) ra[i] = al.load(i, 0);
#pragma unroll
  for (int i = 0; i < TL::B_LOADS; ++i) rb[i] = bl.load(i, 0);
  const int srow = tid >> 3, piece = tid & 7;
  bf16x8 fa[TL::TM][NT], fb[TL::TN][NT];
  if (AB == kNoFrag) {
#pragma unroll
    for (int i = 0; i < TL::TM; ++i)
#pragma unroll
      for (int c = 0; c < NT; ++c) fa[i][c] = __builtin_bit_cast(bf16x8, ra[0]);
#pragma unroll
    for (int j = 0; j < TL::TN; ++j)
#pragma unroll
      for (int c = 0; c < NT; ++c) fb[j][c] = __builtin_bit_cast(bf16x8, rb[0]);
  }
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();
    if (AB != kNoStage) {
#pragma unroll
      for (int i = 0; i < TL::A_LOADS; ++i) {
        if (AB == kNoSplit && NT == 2) {
          const int off = swz_off(srow + 32 * i, piece >> 1) + (piece & 1) * 4;
          *reinterpret_cast<uint2*>(As + off) = make_uint2(__float_as_uint(ra[i].x), __float_as_uint(ra[i].y));
          *reinterpret_cast<uint2*>(As + off + A_IMG) = make_uint2(__float_as_uint(ra[i].z), __float_as_uint(ra[i].w));
        } else halo_store<NT>(As, A_IMG, srow + 32 * i, piece, ra[i], sa);
      }
#pragma unroll
      for (int i = 0; i < TL::B_LOADS; ++i) {
        if (AB == kNoSplit && NT == 2) {
          const int off = swz_off(srow + 32 * i, piece >> 1) + (piece & 1) * 4;
          *reinterpret_cast<uint2*>(Bs + off) = make_uint2(__float_as_uint(rb[i].x), __float_as_uint(rb[i].y));
          *reinterpret_cast<uint2*>(Bs + off + B_IMG) = make_uint2(__float_as_uint(rb[i].z), __float_as_uint(rb[i].w));
        } else halo_store<NT>(Bs, B_IMG, srow + 32 * i, piece, rb[i], sb);
      }
    }
    __syncthreads();
    if (AB != kNoLoads && kt + 1 < nk) {
#pragma unroll
      for (int i = 0; i < TL::A_LOADS; ++i) ra[i] = al.load(i, AB == kSameTile ? (kt & 1) : kt + 1);
#pragma unroll
      for (int i = 0; i < TL::B_LOADS; ++i) rb[i] = bl.load(i, AB == kSameTile ? (kt & 1) : kt + 1);
    }
#pragma unroll
    for (int kk = 0; kk < kBK / 16; ++kk) {
      if (AB != kNoFrag) {
#pragma unroll
        for (int i = 0; i < TL::TM; ++i)
#pragma unroll
          for (int c = 0; c < NT; ++c)
            fa[i][c] = *reinterpret_cast<const bf16x8*>(As + c * A_IMG + swz_off(wm * TL::WM + i * 32 + r, kk * 2 + h));
#pragma unroll
        for (int j = 0; j < TL::TN; ++j)
#pragma unroll
          for (int c = 0; c < NT; ++c)
            fb[j][c] = *reinterpret_cast<const bf16x8*>(Bs + c * B_IMG + swz_off(wn * TL::WN + j * 32 + r, kk * 2 + h));
      }
      if constexpr (AB == kShape16 && NT == 2) {
        // the same FLOPs on v_mfma_f32_16x16x32_f16 (two per 32x32x16; fragment registers reused as they are, so the
        // VALUES are meaningless: a timing / power probe of the instruction shape only)
        typedef float f4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int i = 0; i < TL::TM; ++i)
#pragma unroll
          for (int j = 0; j < TL::TN; ++j)
#pragma unroll
            for (int t = 0; t < 3; ++t) {
              const f16x8 av = __builtin_bit_cast(f16x8, fa[i][t == 0 ? 1 : 0]), bv = __builtin_bit_cast(f16x8, fb[j][t == 1 ? 1 : 0]);
              f4 c0 = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
              f4 c1 = {acc[i][j][4], acc[i][j][5], acc[i][j][6], acc[i][j][7]};
              c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, c0, 0, 0, 0);
              c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, c1, 0, 0, 0);
              acc[i][j][0] = c0[0]; acc[i][j][1] = c0[1]; acc[i][j][2] = c0[2]; acc[i][j][3] = c0[3];
              acc[i][j][4] = c1[0]; acc[i][j][5] = c1[1]; acc[i][j][6] = c1[2]; acc[i][j][7] = c1[3];
            }
      } else if (AB != kNoMfma) {
#pragma unroll
        for (int i = 0; i < TL::TM; ++i)
#pragma unroll
          for (int j = 0; j < TL::TN; ++j) acc[i][j] = mfma_terms<NT, false>(fa[i], fb[j], acc[i][j]);
      } else {
#pragma unroll
        for (int i = 0; i < TL::TM; ++i)
#pragma unroll
          for (int j = 0; j < TL::TN; ++j) acc[i][j][0] += __builtin_bit_cast(float4, fa[i][0]).x + __builtin_bit_cast(float4, fb[j][1]).y;
      }
    }
  }
  for_each_acc<TL>(acc, [&](int rr, int cc, float v) { out[(long)(m0 + rr) * ldc + n0 + cc] = v; });
}


// One barrier per k-tile: two LDS stages; the split + LDS stores of tile kt + 1 and the requests for tile kt + 2 sit in
// the same straight-line block as the MFMAs of tile kt.
template <class TL, int OCC, int IL>
__global__ __launch_bounds__(256, OCC) void nt_ov_kernel(RowLoader al, RowLoader bl, float* out, int ldc, int K,
                                                         int tiles_m, int tiles_n, float sa, float sb) {
  constexpr int NT = STAMP_NT;
  constexpr int A_IMG = TL::BM * kBK, B_IMG = TL::BN * kBK;
  __shared__ __attribute__((aligned(16))) __bf16 As[2][NT * A_IMG];
  __shared__ __attribute__((aligned(16))) __bf16 Bs[2][NT * B_IMG];
  const int tile_id = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int m0 = (tile_id / tiles_n) * TL::BM, n0 = (tile_id % tiles_n) * TL::BN;
  al.init(m0);
  bl.init(n0);
  f32x16 acc[TL::TM][TL::TN];
  zero_acc<TL>(acc);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wm = wv / TL::WAVES_N, wn = wv % TL::WAVES_N;
  const int r = lane & 31, h = lane >> 5;
  const int nk = (K + kBK - 1) / kBK;
  float4 ra[TL::A_LOADS], rb[TL::B_LOADS];
  const int srow = tid >> 3, piece = tid & 7;
  auto fetch = [&](int kt) {
#pragma unroll
    for (int i = 0; i < TL::A_LOADS; ++i) ra[i] = al.load(i, kt);
#pragma unroll
    for (int i = 0; i < TL::B_LOADS; ++i) rb[i] = bl.load(i, kt);
  };
  auto stage = [&](int buf) {
#pragma unroll
    for (int i = 0; i < TL::A_LOADS; ++i) halo_store<NT>(As[buf], A_IMG, srow + 32 * i, piece, ra[i], sa);
#pragma unroll
    for (int i = 0; i < TL::B_LOADS; ++i) halo_store<NT>(Bs[buf], B_IMG, srow + 32 * i, piece, rb[i], sb);
  };
  auto products = [&](int buf) {
#pragma unroll
    for (int kk = 0; kk < kBK / 16; ++kk) {
      bf16x8 fa[TL::TM][NT], fb[TL::TN][NT];
#pragma unroll
      for (int i = 0; i < TL::TM; ++i)
#pragma unroll
        for (int c = 0; c < NT; ++c)
          fa[i][c] = *reinterpret_cast<const bf16x8*>(As[buf] + c * A_IMG + swz_off(wm * TL::WM + i * 32 + r, kk * 2 + h));
#pragma unroll
      for (int j = 0; j < TL::TN; ++j)
#pragma unroll
        for (int c = 0; c < NT; ++c)
          fb[j][c] = *reinterpret_cast<const bf16x8*>(Bs[buf] + c * B_IMG + swz_off(wn * TL::WN + j * 32 + r, kk * 2 + h));
#pragma unroll
      for (int i = 0; i < TL::TM; ++i)
#pragma unroll
        for (int j = 0; j < TL::TN; ++j) acc[i][j] = mfma_terms<NT, false>(fa[i], fb[j], acc[i][j]);
    }
  };
  fetch(0);
  stage(0);
  fetch(1);
  __syncthreads();
  auto iter = [&](auto bc, int kt) {
    constexpr int buf = decltype(bc)::value;
    stage(buf ^ 1);                  // tile kt + 1 (zeros past K: nobody reads them)
    fetch(kt + 2);
    products(buf);
    if constexpr (IL > 0) {
      constexpr int NF = (TL::TM + TL::TN) * NT, NM = TL::TM * TL::TN * 3;
      __builtin_amdgcn_sched_group_barrier(0x100, NF, 0);      // fragments of the first 16 k
      for (int g = 0; g < NM; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // 1 MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, IL, 0);    // IL VALU (the split)
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);     // 1 DS write
      }
      __builtin_amdgcn_sched_group_barrier(0x100, NF, 0);      // fragments of the second 16 k
      for (int g = 0; g < NM; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);     // 1 VMEM read
      }
    }
    __syncthreads();
  };
  int kt = 0;
  for (; kt + 1 < nk; kt += 2) {
    iter(std::integral_constant<int, 0>{}, kt);
    iter(std::integral_constant<int, 1>{}, kt + 1);
  }
  if (kt < nk) iter(std::integral_constant<int, 0>{}, kt);
  for_each_acc<TL>(acc, [&](int rr, int cc, float v) { out[(long)(m0 + rr) * ldc + n0 + cc] = v; });
}

template <class TL, int OCC, int IL>
void run_ov(const char* name, const float* A, const float* B, float* C, int M, int N, int K, int reps) {
  RowLoader al{A, (long)K, M, K, 0}, bl{B, (long)K, N, K, 0};
  const int tm = M / TL::BM, tn = N / TL::BN;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) nt_ov_kernel<TL, OCC, IL><<<tm * tn, 256>>>(al, bl, C, N, K, tm, tn, 1024.f, 1024.f);
  hipEventRecord(e0);
  for (int w = 0; w < reps; ++w) nt_ov_kernel<TL, OCC, IL><<<tm * tn, 256>>>(al, bl, C, N, K, tm, tn, 1024.f, 1024.f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  printf("%-44s %.3f ms (%.0f TF)  %s\n", name, ms, 2.0 * M * N * K / ms * 1e-9, hipGetErrorString(hipGetLastError()));
}


#if STAMP_NT == 2
// Hand-ordered version of the same loop: one MFMA per slot, the slot's share of the next tile's staging behind it, a
// scheduling fence after every slot.
template <class TL, int OCC, int PAD = 0>
__global__ __launch_bounds__(256, OCC) void nt_slot_kernel(RowLoader al, RowLoader bl, float* out, int ldc, int K,
                                                           int tiles_m, int tiles_n, float sa, float sb) {
  constexpr int NT = STAMP_NT;
  constexpr int A_IMG = TL::BM * kBK, B_IMG = TL::BN * kBK;
  constexpr int NQ = TL::A_LOADS + TL::B_LOADS;            // operand quads per thread and k-tile
  constexpr int NB = TL::TM * TL::TN;                      // accumulator blocks
  constexpr int SLOTS = 2 * NB * 3;                        // MFMAs per k-tile
  __shared__ __attribute__((aligned(16))) __bf16 As[2][NT * A_IMG];
  __shared__ __attribute__((aligned(16))) __bf16 Bs[2][NT * B_IMG];
  __shared__ float pad_lds[PAD + 1];
  if (PAD > 0 && K < 0) pad_lds[threadIdx.x] = 1.f;         // keeps the padding allocated
  const int tile_id = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int m0 = (tile_id / tiles_n) * TL::BM, n0 = (tile_id % tiles_n) * TL::BN;
  al.init(m0);
  bl.init(n0);
  f32x16 acc[TL::TM][TL::TN];
  zero_acc<TL>(acc);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wm = wv / TL::WAVES_N, wn = wv % TL::WAVES_N;
  const int r = lane & 31, h = lane >> 5;
  const int nk = (K + kBK - 1) / kBK;
  float4 raw[NQ];
  const int srow = tid >> 3, piece = tid & 7;
  auto load_q = [&](auto qc, int kt) {
    constexpr int q = decltype(qc)::value;
    if constexpr (q < TL::A_LOADS) raw[q] = al.load(q, kt);
    else raw[q] = bl.load(q - TL::A_LOADS, kt);
  };
  // staging offsets (bf16 elements) of this thread's quads, and fragment read offsets
  int st_off[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int row = srow + 32 * (q < TL::A_LOADS ? q : q - TL::A_LOADS);
    st_off[q] = swz_off(row, piece >> 1) + (piece & 1) * 4;
  }
  // micro-ops of one quad: 4 x (two mixed-precision FMAs -> one packed fp16 pair), stores, request
  uint2 hi[2], lo[2];                                      // of the quad in flight (two alternate)
  auto pair_hi = [&](float x, float y, float sc) {
    unsigned d;
    asm("v_fma_mixlo_f16 %0, %1, %3, 0\n\tv_fma_mixhi_f16 %0, %2, %3, 0" : "=&v"(d) : "v"(x), "v"(y), "v"(sc));
    return d;
  };
  auto micro = [&](auto qc, auto pc, int buf, int kt) {
    constexpr int q = decltype(qc)::value, ph = decltype(pc)::value;
    const float sc = q < TL::A_LOADS ? sa : sb;
    if constexpr (ph == 0) hi[q & 1].x = pair_hi(raw[q].x, raw[q].y, sc);
    else if constexpr (ph == 1) hi[q & 1].y = pair_hi(raw[q].z, raw[q].w, sc);
    else if constexpr (ph == 2) lo[q & 1].x = split2_lo_pair(raw[q].x, raw[q].y, sc, hi[q & 1].x);
    else if constexpr (ph == 3) lo[q & 1].y = split2_lo_pair(raw[q].z, raw[q].w, sc, hi[q & 1].y);
    else if constexpr (ph == 4) {
      __bf16* img = q < TL::A_LOADS ? As[buf ^ 1] : Bs[buf ^ 1];
      constexpr int img_elems = q < TL::A_LOADS ? A_IMG : B_IMG;
      *reinterpret_cast<uint2*>(img + st_off[q]) = hi[q & 1];
      *reinterpret_cast<uint2*>(img + st_off[q] + img_elems) = lo[q & 1];
    } else load_q(qc, kt + 2);
  };
  bf16x8 fa[2][TL::TM][NT], fb[2][TL::TN][NT];
  auto read_frags = [&](auto kc, int buf) {
    constexpr int kk = decltype(kc)::value;
#pragma unroll
    for (int i = 0; i < TL::TM; ++i)
#pragma unroll
      for (int c = 0; c < NT; ++c)
        fa[kk][i][c] = *reinterpret_cast<const bf16x8*>(As[buf] + c * A_IMG + swz_off(wm * TL::WM + i * 32 + r, kk * 2 + h));
#pragma unroll
    for (int j = 0; j < TL::TN; ++j)
#pragma unroll
      for (int c = 0; c < NT; ++c)
        fb[kk][j][c] = *reinterpret_cast<const bf16x8*>(Bs[buf] + c * B_IMG + swz_off(wn * TL::WN + j * 32 + r, kk * 2 + h));
  };
  // slot s: kk = s / (3 NB), term t = (s / NB) % 3 in the order a1b0, a0b1, a0b0, block = s % NB
  auto mfma_slot = [&](auto sc) {
    constexpr int s = decltype(sc)::value;
    constexpr int kk = s / (3 * NB), t = (s / NB) % 3, b = s % NB, i = b / TL::TN, j = b % TL::TN;
    constexpr int ca = t == 0 ? 1 : 0, cb = t == 1 ? 1 : 0;
    acc[i][j] = mfma_f16(fa[kk][i][ca], fb[kk][j][cb], acc[i][j]);
  };

  // prologue: tile 0 staged, tile 1 in registers
  static_for<NQ>([&](auto qc) { load_q(qc, 0); });
  static_for<NQ>([&](auto qc) { static_for<5>([&](auto pc) { micro(qc, pc, 1, 0); }); });
  static_for<NQ>([&](auto qc) { load_q(qc, 1); });
  __syncthreads();

  auto iter = [&](auto bc, int kt) {
    constexpr int buf = decltype(bc)::value;
    read_frags(std::integral_constant<int, 0>{}, buf);
    __builtin_amdgcn_sched_barrier(0);
    static_for<SLOTS>([&](auto sc) {
      constexpr int s = decltype(sc)::value;
      mfma_slot(sc);
      // staging of tile kt + 1: 6 micro-ops per quad, spread evenly over the slots
      constexpr int NM = 6 * NQ, m0 = s * NM / SLOTS, m1 = (s + 1) * NM / SLOTS;
      static_for<m1 - m0>([&](auto uc) {
        constexpr int m = m0 + decltype(uc)::value;
        micro(std::integral_constant<int, m / 6>{}, std::integral_constant<int, m % 6>{}, buf, kt);
      });
      if constexpr (s == 3) read_frags(std::integral_constant<int, 1>{}, buf);
      __builtin_amdgcn_sched_barrier(0);
    });
    __syncthreads();
  };
  int kt = 0;
  for (; kt + 1 < nk; kt += 2) {
    iter(std::integral_constant<int, 0>{}, kt);
    iter(std::integral_constant<int, 1>{}, kt + 1);
  }
  if (kt < nk) iter(std::integral_constant<int, 0>{}, kt);
  for_each_acc<TL>(acc, [&](int rr, int cc, float v) { out[(long)(m0 + rr) * ldc + n0 + cc] = v + (PAD > 0 && K < 0 ? pad_lds[0] : 0.f); });
}

template <class TL, int OCC, int PAD = 0>
void run_slot(const char* name, const float* A, const float* B, float* C, int M, int N, int K, int reps) {
  RowLoader al{A, (long)K, M, K, 0}, bl{B, (long)K, N, K, 0};
  const int tm = M / TL::BM, tn = N / TL::BN;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) nt_slot_kernel<TL, OCC, PAD><<<tm * tn, 256>>>(al, bl, C, N, K, tm, tn, 1024.f, 1024.f);
  hipEventRecord(e0);
  for (int w = 0; w < reps; ++w) nt_slot_kernel<TL, OCC, PAD><<<tm * tn, 256>>>(al, bl, C, N, K, tm, tn, 1024.f, 1024.f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  int occ = 0;
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, nt_slot_kernel<TL, OCC, PAD>, 256, 0);
  printf("%-44s %.3f ms (%.0f TF)  %d WG/CU  %s\n", name, ms, 2.0 * M * N * K / ms * 1e-9, occ, hipGetErrorString(hipGetLastError()));
}


// What pre-split operands would buy: both operands already stored as two fp16 planes per 32-k tile ([row][k-tile][hi 64 B |
// lo 64 B]), staged by LDS-DMA (no VGPR round trip, no split, no ds_write), two LDS stages, one barrier per k-tile.
// Values are whatever the buffers hold: a timing probe.
template <class TL, int OCC>
__global__ __launch_bounds__(256, OCC) void nt_dma_kernel(const uint4* __restrict__ Asp, const uint4* __restrict__ Bsp,
                                                          float* out, int ldc, int M, int N, int K, int tiles_m,
                                                          int tiles_n) {
  constexpr int NT = STAMP_NT;
  constexpr int A_IMG = TL::BM * kBK, B_IMG = TL::BN * kBK;           // bf16 elements per term image
  __shared__ __attribute__((aligned(16))) __bf16 As[2][NT * A_IMG];
  __shared__ __attribute__((aligned(16))) __bf16 Bs[2][NT * B_IMG];
  const int tile_id = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int m0 = (tile_id / tiles_n) * TL::BM, n0 = (tile_id % tiles_n) * TL::BN;
  f32x16 acc[TL::TM][TL::TN];
  zero_acc<TL>(acc);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wm = wv / TL::WAVES_N, wn = wv % TL::WAVES_N;
  const int r = lane & 31, h = lane >> 5;
  const int nk = K / kBK;
  // a piece = 16 rows x 64 B of ONE term image = 1 KB per wave instruction; lane l -> row l / 4, slot l % 4, and it
  // fetches the chunk that the swizzle puts into that slot
  const long row_bytes = (long)nk * 128;                             // bytes per operand row in the split layout
  const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(Asp) + (long)m0 * row_bytes / 16, 0, (unsigned)(TL::BM * row_bytes), 0x00020000);
  const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(Bsp) + (long)n0 * row_bytes / 16, 0, (unsigned)(TL::BN * row_bytes), 0x00020000);
  const int prow = lane >> 2, slot = lane & 3;
  constexpr int A_PIECES = TL::BM / 16 * NT, B_PIECES = TL::BN / 16 * NT;   // per k-tile
  auto dma = [&](int buf, int kt) {
    // pieces are dealt round-robin to the four waves
#pragma unroll
    for (int p = 0; p < (A_PIECES + 3) / 4; ++p) {
      const int pc = p * 4 + wv;
      if (pc < A_PIECES) {
        const int c = pc % NT, rb = pc / NT, row = rb * 16 + prow;
        const unsigned vo = (unsigned)(row * row_bytes + c * 64 + ((slot ^ ((row >> 2) & 3)) * 16));
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ars, (__attribute__((address_space(3))) void*)(As[buf] + c * A_IMG + rb * 16 * 32), 16, vo, (unsigned)(kt * 128), 0, 0);
      }
    }
#pragma unroll
    for (int p = 0; p < (B_PIECES + 3) / 4; ++p) {
      const int pc = p * 4 + wv;
      if (pc < B_PIECES) {
        const int c = pc % NT, rb = pc / NT, row = rb * 16 + prow;
        const unsigned vo = (unsigned)(row * row_bytes + c * 64 + ((slot ^ ((row >> 2) & 3)) * 16));
        __builtin_amdgcn_raw_ptr_buffer_load_lds(brs, (__attribute__((address_space(3))) void*)(Bs[buf] + c * B_IMG + rb * 16 * 32), 16, vo, (unsigned)(kt * 128), 0, 0);
      }
    }
  };
  dma(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                  // tile kt has landed everywhere; everyone is done with tile kt - 1
    if (kt + 1 < nk) dma(buf ^ 1, kt + 1);
#pragma unroll
    for (int kk = 0; kk < kBK / 16; ++kk) {
      bf16x8 fa[TL::TM][NT], fb[TL::TN][NT];
#pragma unroll
      for (int i = 0; i < TL::TM; ++i)
#pragma unroll
        for (int c = 0; c < NT; ++c)
          fa[i][c] = *reinterpret_cast<const bf16x8*>(As[buf] + c * A_IMG + swz_off(wm * TL::WM + i * 32 + r, kk * 2 + h));
#pragma unroll
      for (int j = 0; j < TL::TN; ++j)
#pragma unroll
        for (int c = 0; c < NT; ++c)
          fb[j][c] = *reinterpret_cast<const bf16x8*>(Bs[buf] + c * B_IMG + swz_off(wn * TL::WN + j * 32 + r, kk * 2 + h));
#pragma unroll
      for (int i = 0; i < TL::TM; ++i)
#pragma unroll
        for (int j = 0; j < TL::TN; ++j) acc[i][j] = mfma_terms<NT, false>(fa[i], fb[j], acc[i][j]);
    }
  }
  for_each_acc<TL>(acc, [&](int rr, int cc, float v) { out[(long)(m0 + rr) * ldc + n0 + cc] = v; });
}

template <class TL, int OCC>
void run_dma(const char* name, const float* A, const float* B, float* C, int M, int N, int K, int reps) {
  // the fp32 buffers are simply reinterpreted: same bytes per element as two fp16 planes
  const int tm = M / TL::BM, tn = N / TL::BN;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  auto go = [&]() { nt_dma_kernel<TL, OCC><<<tm * tn, 256>>>((const uint4*)A, (const uint4*)B, C, N, M, N, K, tm, tn); };
  for (int w = 0; w < 2; ++w) go();
  hipEventRecord(e0);
  for (int w = 0; w < reps; ++w) go();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  int occ = 0;
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, nt_dma_kernel<TL, OCC>, 256, 0);
  printf("%-44s %.3f ms (%.0f TF)  %d WG/CU  %s\n", name, ms, 2.0 * M * N * K / ms * 1e-9, occ, hipGetErrorString(hipGetLastError()));
}

#endif

template <class TL, int AB, int OCC, int PAD = 0>
float run(const float* A, const float* B, float* C, int M, int N, int K, int reps) {
  RowLoader al{A, (long)K, M, K, 0}, bl{B, (long)K, N, K, 0};
  const int tm = M / TL::BM, tn = N / TL::BN;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) nt_kernel<TL, AB, OCC, PAD><<<tm * tn, 256>>>(al, bl, C, N, K, tm, tn, 1024.f, 1024.f);
  hipEventRecord(e0);
  for (int w = 0; w < reps; ++w) nt_kernel<TL, AB, OCC, PAD><<<tm * tn, 256>>>(al, bl, C, N, K, tm, tn, 1024.f, 1024.f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

template <class TL, int OCC>
void sweep(const char* name, const float* A, const float* B, float* C, int M, int N, int K) {
  const double fl = 2.0 * M * N * K;
  const float tsame = run<TL, kSameTile, OCC>(A, B, C, M, N, K, 20);
  const float t16 = run<TL, kShape16, OCC>(A, B, C, M, N, K, 20);
  const float t[6] = {run<TL, kFull, OCC>(A, B, C, M, N, K, 20), run<TL, kNoLoads, OCC>(A, B, C, M, N, K, 20),
                      run<TL, kNoStage, OCC>(A, B, C, M, N, K, 20), run<TL, kNoFrag, OCC>(A, B, C, M, N, K, 20),
                      run<TL, kNoMfma, OCC>(A, B, C, M, N, K, 20), run<TL, kNoSplit, OCC>(A, B, C, M, N, K, 20)};
  int occ = 0;
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, nt_kernel<TL, kFull, OCC>, 256, 0);
  printf("[%d WG/CU] ", occ);
  printf("%-28s full %.3f ms (%.0f TF) | no global loads %.3f | no split+LDS stores %.3f | no fragment reads %.3f | no MFMA %.3f | stores without split %.3f | loads that always hit %.3f | 16x16x32 MFMA shape (same FLOPs) %.3f\n",
         name, t[0], fl / t[0] * 1e-9, t[1], t[2], t[3], t[4], t[5], tsame, t16);
}

template <class TL>
void one_wg(const float* A, const float* B, float* C, int M, int N, int K) {
  const float t = run<TL, kFull, 1, 28000>(A, B, C, M, N, K, 20);
  int occ = 0;
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, nt_kernel<TL, kFull, 1, 28000>, 256, 0);
  printf("[%d WG/CU] two-barrier loop, padded LDS: %.3f ms\n", occ, t);
}

int main() {
  const int M = 49152, N = 768, K = 1536;
  float *A, *B, *C;
  hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&B, (size_t)N * K * 4); hipMalloc(&C, (size_t)M * N * 4);
  std::vector<float> h((size_t)M * K);
  unsigned s = 12345u;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) * (1.0f / 16777216.0f) - 0.5f) * 1e-3f; }
  hipMemcpy(A, h.data(), (size_t)M * K * 4, hipMemcpyHostToDevice);
  hipMemcpy(B, h.data(), (size_t)N * K * 4, hipMemcpyHostToDevice);
#if STAMP_NT == 1
  // one rounded bf16 term per operand (the mixed-precision GEMMs): a third of the matrix work, same staging
  sweep<Tile<128, 192, 2, 2>, 2>("1 term, 128x192, 2 WG/CU", A, B, C, M, N, K);
  run_ov<Tile<128, 192, 2, 2>, 1, 0>("1 term, one barrier / two stages 128x192", A, B, C, M, N, K, 20);
  run_ov<Tile<128, 192, 2, 2>, 1, 3>("1 term, one barrier, pipelined 128x192", A, B, C, M, N, K, 20);
  sweep<Tile<128, 384, 2, 2>, 1>("1 term, 128x384", A, B, C, M, N, K);
  sweep<Tile<256, 192, 4, 1>, 1>("1 term, 256x192 (4x1 waves)", A, B, C, M, N, K);
  sweep<Tile<128, 128, 2, 2>, 3>("1 term, 128x128, 3 WG/CU", A, B, C, M, N, K);
  run_ov<Tile<128, 128, 2, 2>, 1, 0>("1 term, one barrier / two stages 128x128", A, B, C, M, N, K, 20);
  return 0;
#else
  run_dma<Tile<128, 192, 2, 2>, 1>("pre-split operands by LDS-DMA, 128x192", A, B, C, M, N, K, 20);
  run_dma<Tile<128, 128, 2, 2>, 1>("pre-split operands by LDS-DMA, 128x128", A, B, C, M, N, K, 20);
  run_slot<Tile<128, 192, 2, 2>, 1>("slotted 128x192", A, B, C, M, N, K, 20);
  run_slot<Tile<128, 192, 2, 2>, 1, 4096>("slotted 128x192, padded LDS", A, B, C, M, N, K, 20);
  one_wg<Tile<128, 192, 2, 2>>(A, B, C, M, N, K);
  sweep<Tile<128, 192, 2, 2>, 2>("128x192, 2 WG/CU", A, B, C, M, N, K);
  hipMemset(A, 0, (size_t)M * K * 4); hipMemset(B, 0, (size_t)N * K * 4);
  sweep<Tile<128, 192, 2, 2>, 2>("128x192, zero data", A, B, C, M, N, K);
  hipMemcpy(A, h.data(), (size_t)M * K * 4, hipMemcpyHostToDevice);
  hipMemcpy(B, h.data(), (size_t)N * K * 4, hipMemcpyHostToDevice);
  sweep<Tile<128, 192, 2, 2>, 1>("128x192, launch bound 1", A, B, C, M, N, K);
  sweep<Tile<128, 128, 2, 2>, 3>("128x128, 3 WG/CU", A, B, C, M, N, K);
  sweep<Tile<256, 64, 4, 1>, 2>("256x64", A, B, C, M, N, K);
#endif
  return 0;
}
