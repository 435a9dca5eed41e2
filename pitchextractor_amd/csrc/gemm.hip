// Dense fp32 GEMMs on the MFMA tile engines (gemm_engine.h).
//
//   pe_gemm_nt:  C[M][N] = A[M][K] . B[N][K]^T (+ bias0[n] + bias1[n]) (+ C)
//                -> nn.Linear forward, LSTM input projections (model.py:220-227), 1x1 convs in
//                   channels-last (model.py:53,167), transposed-weight dgrad products.
//   pe_gemm_tn:  C[M][N] = sum_k A[k][m] . B[k][n]  (k = rows), split over k across workgroups
//                -> weight gradients (dW = dY^T X), deterministic slab + ordered reduce.
#include <stdlib.h>
#include "gemm_engine.h"

namespace {
using namespace pe;

struct StoreEpi {
  float* C;
  long ldc;
  const float* bias0;
  const float* bias1;
  int M, N, accumulate;
  __device__ __forceinline__ void operator()(int row, int col, float v) const {
    if (row < M && col < N) {
      if (bias0) v += bias0[col];
      if (bias1) v += bias1[col];
      float* dst = C + (long)row * ldc + col;
      if (accumulate) v += *dst;
      *dst = v;
    }
  }
};

template <class TL, int MODE>
__global__ __launch_bounds__(256) void gemm_nt_kernel(RowLoader al, RowLoader bl, StoreEpi ep, int K,
                                                      int tiles_m, int tiles_n) {
  __shared__ __attribute__((aligned(16))) float As[TL::BM * nt_row_floats<MODE>()];
  __shared__ __attribute__((aligned(16))) float Bs[TL::BN * nt_row_floats<MODE>()];
  const int tile = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int m0 = (tile / tiles_n) * TL::BM, n0 = (tile % tiles_n) * TL::BN;
  al.init(m0);
  bl.init(n0);
  f32x16 acc[TL::TM][TL::TN];
  zero_acc<TL>(acc);
  nt_mainloop_mode<TL, MODE>(al, bl, K, As, Bs, acc);
  for_each_acc<TL>(acc, [&](int r, int c, float v) { ep(m0 + r, n0 + c, v); });
}

// The same product with the operand roles swapped inside the tile engine (the NT main loop is symmetric in its two
// operands): the accumulators then hold the TRANSPOSED 32 x 32 blocks, i.e. a lane owns four consecutive output
// columns of one row instead of four rows of one column, and the epilogue writes 16-byte pieces (a quarter of the
// store instructions, bias fetched once per column quad).  Bit-identical sums.  Needs N % 4 == 0 and a 16-byte
// aligned C with ldc % 4 == 0 (checked by the host).
template <class TL, int MODE>
__global__ __launch_bounds__(256) void gemm_nt_t_kernel(RowLoader al, RowLoader bl, StoreEpi ep, int K, int tiles_m,
                                                        int tiles_n) {
  using TT = Tile<TL::BN, TL::BM, TL::WAVES_N, TL::WAVES_M>;
  __shared__ __attribute__((aligned(16))) float As[TL::BM * nt_row_floats<MODE>()];
  __shared__ __attribute__((aligned(16))) float Bs[TL::BN * nt_row_floats<MODE>()];
  const int tile = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int m0 = (tile / tiles_n) * TL::BM, n0 = (tile % tiles_n) * TL::BN;
  al.init(m0);
  bl.init(n0);
  f32x16 acc[TT::TM][TT::TN];
  zero_acc<TT>(acc);
  nt_mainloop_mode<TT, MODE, true>(bl, al, K, Bs, As, acc);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int wn = wv / TT::WAVES_N, wm = wv % TT::WAVES_N;          // TT's "rows" are output columns
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int i = 0; i < TT::TM; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int col = n0 + wn * TT::WM + i * 32 + 8 * q + 4 * h;
      if (col >= ep.N) continue;
      float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0;        // (acc + bias0) + bias1, as the scalar epilogue
      if (ep.bias0) b0 = make_float4(ep.bias0[col], ep.bias0[col + 1], ep.bias0[col + 2], ep.bias0[col + 3]);
      if (ep.bias1) b1 = make_float4(ep.bias1[col], ep.bias1[col + 1], ep.bias1[col + 2], ep.bias1[col + 3]);
#pragma unroll
      for (int j = 0; j < TT::TN; ++j) {
        const int row = m0 + wm * TT::WN + j * 32 + r;
        if (row >= ep.M) continue;
        float4* dst = reinterpret_cast<float4*>(ep.C + (long)row * ep.ldc + col);
        float4 v = make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
        if (ep.bias0) { v.x += b0.x; v.y += b0.y; v.z += b0.z; v.w += b0.w; }
        if (ep.bias1) { v.x += b1.x; v.y += b1.y; v.z += b1.z; v.w += b1.w; }
        if (ep.accumulate) { const float4 o = *dst; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
        *dst = v;
      }
    }
}

template <class TL, int MODE>
int launch_nt(const RowLoader& al, const RowLoader& bl, const StoreEpi& ep, int M, int N, int K,
              hipStream_t st) {
  const int tm = pe_cdiv(M, TL::BM), tn = pe_cdiv(N, TL::BN);
  static const bool off = getenv("PE_GEMM_NT_SCALAR_EPILOGUE") != nullptr;     // A/B switch (tools/ab_gemm.py)
  const bool vec = !off && MODE != kNative && (N & 3) == 0 && (ep.ldc & 3) == 0 &&
                   (reinterpret_cast<uintptr_t>(ep.C) & 15) == 0;
  if (vec) hipLaunchKernelGGL((gemm_nt_t_kernel<TL, MODE>), dim3(tm * tn), dim3(256), 0, st, al, bl, ep, K, tm, tn);
  else hipLaunchKernelGGL((gemm_nt_kernel<TL, MODE>), dim3(tm * tn), dim3(256), 0, st, al, bl, ep, K, tm, tn);
  PE_LAUNCH_CHECK();
  return PE_OK;
}

// ---- NT with the B operand (weights) pre-packed as MFMA fragments (pe_wfrag_pack, conv.hip)
// Same idea as conv3x3_halo_wf_kernel: the weight matrix is packed once per call into B-fragment order and every
// wave pulls its fragments straight from L2 into a register ring, D steps ahead of the MFMAs; only the activation
// tile goes through LDS (split into bf16 terms on the way in), double-buffered: ONE barrier per 32-k tile and no
// weight split / weight LDS traffic in the loop.  Tile 128 x BN, waves 2 x 2, step = (kk, j) = 2 TM MFMA groups.
// ABL: timing-only ablation mask of the diagnostic entry pe_gemm_nt_wf_ablate (results are wrong when non-zero):
//   1 no weight-fragment loads in the loop, 2 no split + LDS store of the next A tile, 4 no global loads of A,
//   8 no epilogue stores, 16 no per-k-tile barrier, 32 no A-fragment LDS reads in the loop
template <int BN, int MODE, int D, bool FA2, int ABL = 0>
__global__ __launch_bounds__(256, 2) void gemm_nt_wf_kernel(RowLoader al, const uint4* __restrict__ wf, StoreEpi ep,
                                                            int N, int K, int tiles_m, int tiles_n) {
  constexpr int NT = MODE == kSplit ? 3 : 1;
  constexpr int TM = 2, TN = BN / 64, WN = BN / 2;
  constexpr int AIMG = 128 * 32;                                   // bf16 elements per term image
  constexpr int S = 2 * TN;                                        // steps (kk, j) per k-tile
  static_assert(S % D == 0 && D >= 2, "the fragment ring wraps at k-tile boundaries");
  __shared__ __attribute__((aligned(16))) __bf16 As[2 * NT * AIMG];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wm = wv >> 1, wn = wv & 1, r = lane & 31, h = lane >> 5;
  const int tile = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int m0 = (tile / tiles_n) * 128, n0 = (tile % tiles_n) * BN;
  const int srow = tid >> 3, piece = tid & 7;
  const int nk = (K + kBK - 1) / kBK, KB = K >> 4;
  al.init(m0);

  const int NB32 = (N + 31) >> 5;
  int nbo[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int nb = (n0 + wn * WN + j * 32) >> 5;
    nbo[j] = (nb < NB32 ? nb : NB32 - 1) * NT * 64 + lane;         // a tile hanging over N re-reads the last block
  }
  const int kb_stride = NB32 * NT * 64;
  bf16x8 ring[D][NT];
  auto issue = [&](int slot, int kt, int s) {                      // s = kk * TN + j
    const int j = s % TN, kk = s / TN;
    int kb = kt * 2 + kk;
    kb = kb < KB ? kb : KB - 1;                                    // K % 32 == 16: the activation half-tile is zero
    const uint4* pw = wf + (long)kb * kb_stride + nbo[j];
#pragma unroll
    for (int c = 0; c < NT; ++c) ring[slot][c] = __builtin_bit_cast(bf16x8, pw[c * 64]);
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int g = 0; g < 16; ++g) acc[i][j][g] = 0.f;

  float4 ra[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) ra[i] = al.load(i, 0);
#pragma unroll
  for (int s = 0; s < D - 1; ++s) issue(s, 0, s);
#pragma unroll
  for (int i = 0; i < 4; ++i) halo_store<NT>(As, AIMG, i * 32 + srow, piece, ra[i]);
  __syncthreads();

  auto load_fa = [&](bf16x8 (&fa)[TM][NT], const __bf16* buf, int kk) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int c = 0; c < NT; ++c)
        fa[i][c] = *reinterpret_cast<const bf16x8*>(buf + c * AIMG + swz_off(wm * 64 + i * 32 + r, kk * 2 + h));
  };

  for (int kt = 0; kt < nk; ++kt) {
    const __bf16* cur = As + (kt & 1) * NT * AIMG;
    __bf16* nxt = As + ((kt + 1) & 1) * NT * AIMG;
    if (kt + 1 < nk && !(ABL & 4)) {
#pragma unroll
      for (int i = 0; i < 4; ++i) ra[i] = al.load(i, kt + 1);
    }
    bf16x8 fa[FA2 ? 2 : 1][TM][NT];
    if (FA2 && (!(ABL & 32) || kt == 0)) load_fa(fa[0], cur, 0);
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const int j = s % TN, kk = s / TN;
      if (!(ABL & 1)) {
        const int sn = s + D - 1;
        if (sn < S) issue(sn % D, kt, sn);
        else if (kt + 1 < nk) issue(sn % D, kt + 1, sn - S);
      }
      if (!(ABL & 32) || kt == 0) {
        if (FA2) {
          if (j == 0 && kk == 0) load_fa(fa[1], cur, 1);
        } else if (j == 0) {
          load_fa(fa[0], cur, kk);
        }
      }
      const int fs = FA2 ? kk : 0;
      if constexpr (NT == 3) {
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[i][j] = mfma_split(fa[fs][i], ring[s % D], acc[i][j]);
      } else {
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[i][j] = mfma_bf16(fa[fs][i][0], ring[s % D][0], acc[i][j]);
      }
      __builtin_amdgcn_sched_barrier(0);                           // keep the prefetches where they are (see conv.hip)
    }
    if (kt + 1 < nk && !(ABL & 2)) {
#pragma unroll
      for (int i = 0; i < 4; ++i) halo_store<NT>(nxt, AIMG, i * 32 + srow, piece, ra[i]);
    }
    if (!(ABL & 16)) __syncthreads();                              // tile kt + 1 is complete; tile kt is free
  }
  if ((ABL & 8) && tiles_m > 0) return;                            // (never false: keeps the accumulators live)
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int g = 0; g < 16; ++g)
        ep(m0 + wm * 64 + i * 32 + (g & 3) + 8 * (g >> 2) + 4 * h, n0 + wn * WN + j * 32 + r, acc[i][j][g]);
}

template <int BN, int MODE, int D, bool FA2, int ABL = 0>
int launch_nt_wf(const RowLoader& al, const void* wf, const StoreEpi& ep, int M, int N, int K, hipStream_t st) {
  const int tm = pe_cdiv(M, 128), tn = pe_cdiv(N, BN);
  hipLaunchKernelGGL((gemm_nt_wf_kernel<BN, MODE, D, FA2, ABL>), dim3(tm * tn), dim3(256), 0, st, al,
                     reinterpret_cast<const uint4*>(wf), ep, N, K, tm, tn);
  PE_LAUNCH_CHECK();
  return PE_OK;
}

// ---- TN with split-K
template <int BM, int BN, int MODE>
__global__ __launch_bounds__(256) void gemm_tn_kernel(KRowLoader<BM> al, KRowLoader<BN> bl, float* out,
                                                      long ldo, long split_stride, int M, int N, int K,
                                                      int k_per_split, int tiles_n, int accumulate) {
  __shared__ __attribute__((aligned(16))) float As[tn_lds_floats<MODE, BM>()];
  __shared__ __attribute__((aligned(16))) float Bs[tn_lds_floats<MODE, BN>()];
  const int m0 = (blockIdx.x / tiles_n) * BM, n0 = (blockIdx.x % tiles_n) * BN;
  const int kb = blockIdx.y * k_per_split;
  const int ke = min(K, kb + k_per_split);
  al.init(m0, kb);
  bl.init(n0, kb);
  f32x16 acc[BM / 64][BN / 64];
#pragma unroll
  for (int i = 0; i < BM / 64; ++i)
#pragma unroll
    for (int j = 0; j < BN / 64; ++j)
#pragma unroll
      for (int g = 0; g < 16; ++g) acc[i][j][g] = 0.f;
  tn_mainloop_mode<MODE, BM, BN>(al, bl, kb, ke, As, Bs, acc);
  float* dst = out + (long)blockIdx.y * split_stride;
  tn_for_each_acc<BM, BN>(acc, [&](int r, int c, float v) {
    const int row = m0 + r, col = n0 + c;
    if (row < M && col < N) {
      float* d = dst + (long)row * ldo + col;
      if (accumulate) v += *d;
      *d = v;
    }
  });
}

__global__ void splitk_reduce_kernel(const float* ws, long split_stride, int splits, float* C, long ldc,
                                     int M, int N, int accumulate) {   // N % 4 == 0: a float4 stays inside one row
  const long i4 = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i4 * 4 >= (long)M * N) return;
  const float4 s = pe_ordered_slab_sum4(ws, split_stride, splits, i4);
  const long idx = i4 * 4;
  const int row = (int)(idx / N), col = (int)(idx - (long)row * N);
  float* d = C + (long)row * ldc + col;                          // C may be an unaligned view: scalar stores
  if (accumulate) { d[0] += s.x; d[1] += s.y; d[2] += s.z; d[3] += s.w; }
  else { d[0] = s.x; d[1] = s.y; d[2] = s.z; d[3] = s.w; }
}

void tn_plan(int M, int N, int K, int bm, int bn, int mode, int* splits, int* k_per_split) {
  const int tiles = pe_cdiv(M, bm) * pe_cdiv(N, bn);
  // resident workgroups: 3 per CU (native), 2 per CU when the three-term images fill the LDS
  const int s = pe_pick_splits(tiles, K, 512, mode == kSplit ? 512 : 768);   // (bf16: LDS is small, 768 too)
  int kps = pe_cdiv(K, s);
  kps = (kps + kBK - 1) / kBK * kBK;
  *splits = pe_cdiv(K, kps);
  *k_per_split = kps;
}

template <int BM, int BN, int MODE>
int launch_tn(const float* A, long lda, const float* B, long ldb, float* C, long ldc, int M, int N, int K,
              int accumulate, float* ws, size_t ws_bytes, hipStream_t st) {
  int splits, kps;
  tn_plan(M, N, K, BM, BN, MODE, &splits, &kps);
  KRowLoader<BM> al{A, lda, M, 0};
  KRowLoader<BN> bl{B, ldb, N, 0};
  const int tm = pe_cdiv(M, BM), tn = pe_cdiv(N, BN);
  if (splits == 1) {
    hipLaunchKernelGGL((gemm_tn_kernel<BM, BN, MODE>), dim3(tm * tn, 1), dim3(256), 0, st, al, bl, C, ldc, 0L, M, N,
                       K, kps, tn, accumulate);
    PE_LAUNCH_CHECK();
    return PE_OK;
  }
  const size_t need = (size_t)splits * M * N * sizeof(float);
  if (!ws || ws_bytes < need) return PE_E_WORKSPACE;
  hipLaunchKernelGGL((gemm_tn_kernel<BM, BN, MODE>), dim3(tm * tn, splits), dim3(256), 0, st, al, bl, ws, (long)N,
                     (long)M * N, M, N, K, kps, tn, 0);
  PE_LAUNCH_CHECK();
  const long total = (long)M * N;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(pe_cdiv(total / 4, 256)), dim3(256), 0, st, ws, (long)M * N, splits,
                     C, ldc, M, N, accumulate);
  PE_LAUNCH_CHECK();
  return PE_OK;
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

template <int MODE>
static int gemm_nt_impl(const float* A, long lda, const float* B, long ldb, float* C, long ldc, int M, int N,
                        int K, const float* bias0, const float* bias1, int accumulate, void* stream) {
  if (!A || !B || !C || M < 0 || N < 0 || K <= 0) return PE_E_ARG;
  if (M == 0 || N == 0) return PE_OK;
  if ((K & 3) || (lda & 3) || (ldb & 3) || !aligned16(A) || !aligned16(B)) return PE_E_UNSUPPORTED;
  RowLoader al{A, lda, M, K, 0};
  RowLoader bl{B, ldb, N, K, 0};
  StoreEpi ep{C, ldc, bias0, bias1, M, N, accumulate};
  hipStream_t st = pe_stream(stream);
  if (N <= 32) return launch_nt<Tile<128, 32, 4, 1>, MODE>(al, bl, ep, M, N, K, st);
  if (N <= 64) return launch_nt<Tile<256, 64, 4, 1>, MODE>(al, bl, ep, M, N, K, st);
  if (N % 192 == 0 && (N % 128 != 0 || MODE != kNative))     // bf16-term modes: the wider tile stages 17 % fewer rows per MFMA
    return launch_nt<Tile<128, 192, 2, 2>, MODE>(al, bl, ep, M, N, K, st);
  return launch_nt<Tile<128, 128, 2, 2>, MODE>(al, bl, ep, M, N, K, st);
}

#ifndef PE_F16_BUILD
extern "C" int pe_gemm_nt(const float* A, long lda, const float* B, long ldb, float* C, long ldc, int M, int N,
                          int K, const float* bias0, const float* bias1, int accumulate, void* stream) {
  return gemm_nt_impl<kNative>(A, lda, B, ldb, C, ldc, M, N, K, bias0, bias1, accumulate, stream);
}
#endif

extern "C" int PE_HALF(pe_gemm_nt)(const float* A, long lda, const float* B, long ldb, float* C, long ldc, int M,
                               int N, int K, const float* bias0, const float* bias1, int accumulate,
                               void* stream) {
  return gemm_nt_impl<kBf16>(A, lda, B, ldb, C, ldc, M, N, K, bias0, bias1, accumulate, stream);
}

#ifndef PE_F16_BUILD
extern "C" int pe_gemm_nt_x3(const float* A, long lda, const float* B, long ldb, float* C, long ldc, int M, int N,
                             int K, const float* bias0, const float* bias1, int accumulate, void* stream) {
  return gemm_nt_impl<kSplit>(A, lda, B, ldb, C, ldc, M, N, K, bias0, bias1, accumulate, stream);
}
#endif

template <int MODE>
static int gemm_nt_wf_impl(const float* A, long lda, const void* wfrag, float* C, long ldc, int M, int N, int K,
                           const float* bias0, const float* bias1, int accumulate, void* stream) {
  if (!A || !wfrag || !C || M < 0 || N < 0 || K <= 0) return PE_E_ARG;
  if (M == 0 || N == 0) return PE_OK;
  if ((K & 15) || (lda & 3) || !aligned16(A)) return PE_E_UNSUPPORTED;
  RowLoader al{A, lda, M, K, 0};
  StoreEpi ep{C, ldc, bias0, bias1, M, N, accumulate};
  hipStream_t st = pe_stream(stream);
  if (N <= 64) return launch_nt_wf<64, MODE, 2, true>(al, wfrag, ep, M, N, K, st);
  if (N % 192 == 0) return launch_nt_wf<192, MODE, 3, true>(al, wfrag, ep, M, N, K, st);
  return launch_nt_wf<128, MODE, 4, true>(al, wfrag, ep, M, N, K, st);
}

// Diagnostic (tools/ablate_gemm.py): the 128 x 192 x3 kernel with parts of its loop removed.  Timing only.
#ifndef PE_F16_BUILD
extern "C" int pe_gemm_nt_wf_ablate(int mask, const float* A, long lda, const void* wfrag, float* C, long ldc, int M,
                                    int N, int K, void* stream) {
  if (!A || !wfrag || !C || M <= 0 || N <= 0 || K <= 0 || (K & 31) || (N % 192)) return PE_E_ARG;
  RowLoader al{A, lda, M, K, 0};
  StoreEpi ep{C, ldc, nullptr, nullptr, M, N, 0};
  hipStream_t st = pe_stream(stream);
  switch (mask) {
    case 0: return launch_nt_wf<192, kSplit, 3, true, 0>(al, wfrag, ep, M, N, K, st);
    case 1: return launch_nt_wf<192, kSplit, 3, true, 1>(al, wfrag, ep, M, N, K, st);
    case 2: return launch_nt_wf<192, kSplit, 3, true, 2>(al, wfrag, ep, M, N, K, st);
    case 6: return launch_nt_wf<192, kSplit, 3, true, 6>(al, wfrag, ep, M, N, K, st);
    case 8: return launch_nt_wf<192, kSplit, 3, true, 8>(al, wfrag, ep, M, N, K, st);
    case 16: return launch_nt_wf<192, kSplit, 3, true, 16>(al, wfrag, ep, M, N, K, st);
    case 22: return launch_nt_wf<192, kSplit, 3, true, 22>(al, wfrag, ep, M, N, K, st);
    case 32: return launch_nt_wf<192, kSplit, 3, true, 32>(al, wfrag, ep, M, N, K, st);
    case 63: return launch_nt_wf<192, kSplit, 3, true, 63>(al, wfrag, ep, M, N, K, st);
    case 55: return launch_nt_wf<192, kSplit, 3, true, 55>(al, wfrag, ep, M, N, K, st);
  }
  return PE_E_UNSUPPORTED;
}

extern "C" int pe_gemm_nt_wf_x3(const float* A, long lda, const void* wfrag, float* C, long ldc, int M, int N,
                                int K, const float* bias0, const float* bias1, int accumulate, void* stream) {
  return gemm_nt_wf_impl<kSplit>(A, lda, wfrag, C, ldc, M, N, K, bias0, bias1, accumulate, stream);
}
#endif

extern "C" int PE_HALF(pe_gemm_nt_wf)(const float* A, long lda, const void* wfrag, float* C, long ldc, int M, int N,
                                  int K, const float* bias0, const float* bias1, int accumulate, void* stream) {
  return gemm_nt_wf_impl<kBf16>(A, lda, wfrag, C, ldc, M, N, K, bias0, bias1, accumulate, stream);
}

#ifndef PE_F16_BUILD
extern "C" size_t pe_gemm_tn_workspace_bytes(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  size_t need = 0;
  for (int mode : {kNative, kSplit}) {                   // one size serves every pe_gemm_tn* (bf16 plans like native)
    int splits, kps;
    tn_plan(M, N, K, M <= 64 ? 64 : 128, N <= 64 ? 64 : 128, mode, &splits, &kps);
    const size_t b = splits > 1 ? (size_t)splits * M * N * sizeof(float) : 0;
    need = b > need ? b : need;
  }
  return need;
}
#endif

template <int MODE>
static int gemm_tn_impl(const float* A, long lda, const float* B, long ldb, float* C, long ldc, int M, int N,
                        int K, int accumulate, float* workspace, size_t workspace_bytes, void* stream) {
  if (!A || !B || !C || M < 0 || N < 0 || K <= 0) return PE_E_ARG;
  if (M == 0 || N == 0) return PE_OK;
  if ((M & 3) || (N & 3) || (lda & 3) || (ldb & 3) || !aligned16(A) || !aligned16(B)) return PE_E_UNSUPPORTED;
  hipStream_t st = pe_stream(stream);
  if (M <= 64 && N <= 64)
    return launch_tn<64, 64, MODE>(A, lda, B, ldb, C, ldc, M, N, K, accumulate, workspace, workspace_bytes, st);
  if (M <= 64)
    return launch_tn<64, 128, MODE>(A, lda, B, ldb, C, ldc, M, N, K, accumulate, workspace, workspace_bytes, st);
  if (N <= 64)
    return launch_tn<128, 64, MODE>(A, lda, B, ldb, C, ldc, M, N, K, accumulate, workspace, workspace_bytes, st);
  return launch_tn<128, 128, MODE>(A, lda, B, ldb, C, ldc, M, N, K, accumulate, workspace, workspace_bytes, st);
}

#ifndef PE_F16_BUILD
extern "C" int pe_gemm_tn(const float* A, long lda, const float* B, long ldb, float* C, long ldc, int M, int N,
                          int K, int accumulate, float* workspace, size_t workspace_bytes, void* stream) {
  return gemm_tn_impl<kNative>(A, lda, B, ldb, C, ldc, M, N, K, accumulate, workspace, workspace_bytes, stream);
}

extern "C" int pe_gemm_tn_x3(const float* A, long lda, const float* B, long ldb, float* C, long ldc, int M, int N,
                             int K, int accumulate, float* workspace, size_t workspace_bytes, void* stream) {
  return gemm_tn_impl<kSplit>(A, lda, B, ldb, C, ldc, M, N, K, accumulate, workspace, workspace_bytes, stream);
}
#endif

extern "C" int PE_HALF(pe_gemm_tn)(const float* A, long lda, const float* B, long ldb, float* C, long ldc, int M, int N,
                               int K, int accumulate, float* workspace, size_t workspace_bytes, void* stream) {
  return gemm_tn_impl<kBf16>(A, lda, B, ldb, C, ldc, M, N, K, accumulate, workspace, workspace_bytes, stream);
}
