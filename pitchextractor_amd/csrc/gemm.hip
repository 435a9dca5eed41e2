// Dense fp32 GEMMs on the MFMA tile engines (gemm_engine.h).
//
//   pe_gemm_nt:  C[M][N] = A[M][K] . B[N][K]^T (+ bias0[n] + bias1[n]) (+ C)
//                -> nn.Linear forward, LSTM input projections (model.py:220-227), 1x1 convs in
//                   channels-last (model.py:53,167), transposed-weight dgrad products.
//   pe_gemm_tn:  C[M][N] = sum_k A[k][m] . B[k][n]  (k = rows), split over k across workgroups
//                -> weight gradients (dW = dY^T X), deterministic slab + ordered reduce.
#include <stdlib.h>
#include <type_traits>
#include <utility>
#include "gemm_engine.h"


namespace {
using namespace pe;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <class TO>
struct StoreEpiT {
  TO* C;
  long ldc;
  const float* bias0;
  const float* bias1;
  int M, N, accumulate;
  __device__ __forceinline__ void operator()(int row, int col, float v) const {
    if (row < M && col < N) {
      if (bias0) v += bias0[col];
      if (bias1) v += bias1[col];
      TO* dst = C + (long)row * ldc + col;
      if (accumulate) v += ld1(dst);
      st1(dst, v);
    }
  }
};
typedef StoreEpiT<float> StoreEpi;

template <class TL, int MODE, class TA = float>
__global__ __launch_bounds__(256) void gemm_nt_kernel(RowLoaderT<TA> al, RowLoader bl, StoreEpiT<TA> ep, int K,
                                                      int tiles_m, int tiles_n, const unsigned* amax_a,
                                                      const unsigned* amax_b) {
  __shared__ __attribute__((aligned(16))) float As[TL::BM * nt_row_floats<MODE>()];
  __shared__ __attribute__((aligned(16))) float Bs[TL::BN * nt_row_floats<MODE>()];
  const int tile = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int m0 = (tile / tiles_n) * TL::BM, n0 = (tile % tiles_n) * TL::BN;
  al.init(m0);
  bl.init(n0);
  f32x16 acc[TL::TM][TL::TN];
  zero_acc<TL>(acc);
  H2Scales hs{1.f, 1.f, 1.f};
  if constexpr (MODE == kSplit2) hs.load(amax_a, amax_b);
  nt_mainloop_mode<TL, MODE>(al, bl, K, As, Bs, acc, hs.sa, hs.sb);
  for_each_acc<TL>(acc, [&](int r, int c, float v) { ep(m0 + r, n0 + c, MODE == kSplit2 ? v * hs.inv : v); });
}

// The same product with the operand roles swapped inside the tile engine (the NT main loop is symmetric in its two
// operands): the accumulators then hold the TRANSPOSED 32 x 32 blocks, i.e. a lane owns four consecutive output
// columns of one row instead of four rows of one column, and the epilogue writes 16-byte pieces (a quarter of the
// store instructions, bias fetched once per column quad).  Bit-identical sums.  Needs N % 4 == 0 and a 16-byte
// aligned C with ldc % 4 == 0 (checked by the host).
template <class TL, int MODE, class TA = float>
__global__ __launch_bounds__(256, ((MODE == kSplit || MODE == kSplit2) && TL::BM == 128 && TL::BN == 128) ? 3 : 1)
void gemm_nt_t_kernel(RowLoaderT<TA> al, RowLoader bl, StoreEpiT<TA> ep, int K, int tiles_m, int tiles_n,
                      const unsigned* amax_a, const unsigned* amax_b) {
  using TT = Tile<TL::BN, TL::BM, TL::WAVES_N, TL::WAVES_M>;
  __shared__ __attribute__((aligned(16))) float As[TL::BM * nt_row_floats<MODE>()];
  __shared__ __attribute__((aligned(16))) float Bs[TL::BN * nt_row_floats<MODE>()];
  const int tile = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int m0 = (tile / tiles_n) * TL::BM, n0 = (tile % tiles_n) * TL::BN;
  al.init(m0);
  bl.init(n0);
  f32x16 acc[TT::TM][TT::TN];
  zero_acc<TT>(acc);
  H2Scales hs{1.f, 1.f, 1.f};
  if constexpr (MODE == kSplit2) hs.load(amax_a, amax_b);
  nt_mainloop_mode<TT, MODE, true>(bl, al, K, Bs, As, acc, hs.sb, hs.sa);
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
        TA* dst = ep.C + (long)row * ep.ldc + col;
        float4 v = make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
        if constexpr (MODE == kSplit2) { v.x *= hs.inv; v.y *= hs.inv; v.z *= hs.inv; v.w *= hs.inv; }
        if (ep.bias0) { v.x += b0.x; v.y += b0.y; v.z += b0.z; v.w += b0.w; }
        if (ep.bias1) { v.x += b1.x; v.y += b1.y; v.z += b1.z; v.w += b1.w; }
        if (ep.accumulate) { const float4 o = ld4(dst); v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
        st4(dst, v);
      }
    }
}

template <class TL, int MODE, class TA = float>
int launch_nt(const RowLoaderT<TA>& al, const RowLoader& bl, const StoreEpiT<TA>& ep, int M, int N, int K,
              hipStream_t st, const unsigned* amax_a = nullptr, const unsigned* amax_b = nullptr) {
  const int tm = pe_cdiv(M, TL::BM), tn = pe_cdiv(N, TL::BN);
  const bool vec = MODE != kNative && (N & 3) == 0 && (ep.ldc & 3) == 0 &&
                   (reinterpret_cast<uintptr_t>(ep.C) & (4 * sizeof(TA) - 1)) == 0;
  if (vec)
    hipLaunchKernelGGL((gemm_nt_t_kernel<TL, MODE, TA>), dim3(tm * tn), dim3(256), 0, st, al, bl, ep, K, tm, tn,
                       amax_a, amax_b);
  else
    hipLaunchKernelGGL((gemm_nt_kernel<TL, MODE, TA>), dim3(tm * tn), dim3(256), 0, st, al, bl, ep, K, tm, tn, amax_a,
                       amax_b);
  PE_LAUNCH_CHECK();
  return PE_OK;
}

// ---- TN with split-K
template <int BM, int BN, int MODE, class TA = float>
__global__ __launch_bounds__(256) void gemm_tn_kernel(KRowLoader<BM, TA> al, KRowLoader<BN, TA> bl, float* out,
                                                      long ldo, long split_stride, int M, int N, int K,
                                                      int k_per_split, int tiles_n, int accumulate,
                                                      const unsigned* amax_a, const unsigned* amax_b) {
  __shared__ __attribute__((aligned(16))) float As[tn_lds_floats<MODE, BM>()];
  __shared__ __attribute__((aligned(16))) float Bs[tn_lds_floats<MODE, BN>()];
  // 1-D grid over (split, tile) with every XCD taking a CONTIGUOUS run of it: the tiles of one k-split then share
  // an XCD's L2 for the operand rows they all read (PMC: 2.3 GB of fabric reads per dW_ih launch, 5x the operands,
  // with the (tile, split) grid whose consecutive workgroups go round-robin over the eight XCDs)
  const int tiles_mn = ((M + BM - 1) / BM) * tiles_n;
  const int lin = xcd_remap(blockIdx.x, gridDim.x);                // grid = tiles_mn * splits workgroups
  const int tile_id = lin % tiles_mn, split_id = lin / tiles_mn;
  const int m0 = (tile_id / tiles_n) * BM, n0 = (tile_id % tiles_n) * BN;
  const int kb = split_id * k_per_split;
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
  H2Scales hs{1.f, 1.f, 1.f};
  if constexpr (MODE == kSplit2) hs.load(amax_a, amax_b);
  tn_mainloop_mode<MODE, BM, BN>(al, bl, kb, ke, As, Bs, acc, hs.sa, hs.sb);
  float* dst = out + (long)split_id * split_stride;
  tn_for_each_acc<BM, BN>(acc, [&](int r, int c, float v) {
    const int row = m0 + r, col = n0 + c;
    if constexpr (MODE == kSplit2) v *= hs.inv;
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
  const int s = pe_pick_splits(tiles, K, 512, (mode == kSplit || mode == kSplit2) ? 512 : 768);   // (bf16: LDS is small, 768 too)
  int kps = pe_cdiv(K, s);
  kps = (kps + kBK - 1) / kBK * kBK;
  *splits = pe_cdiv(K, kps);
  *k_per_split = kps;
}

template <int BM, int BN, int MODE, class TA = float>
int launch_tn(const TA* A, long lda, const TA* B, long ldb, float* C, long ldc, int M, int N, int K,
              int accumulate, float* ws, size_t ws_bytes, hipStream_t st, const unsigned* amax_a = nullptr,
              const unsigned* amax_b = nullptr) {
  int splits, kps;
  tn_plan(M, N, K, BM, BN, MODE, &splits, &kps);
  KRowLoader<BM, TA> al{A, lda, M, 0};
  KRowLoader<BN, TA> bl{B, ldb, N, 0};
  const int tm = pe_cdiv(M, BM), tn = pe_cdiv(N, BN);
  if (splits == 1) {
    hipLaunchKernelGGL((gemm_tn_kernel<BM, BN, MODE, TA>), dim3(tm * tn, 1), dim3(256), 0, st, al, bl, C, ldc, 0L, M, N,
                       K, kps, tn, accumulate, amax_a, amax_b);
    PE_LAUNCH_CHECK();
    return PE_OK;
  }
  const size_t need = (size_t)splits * M * N * sizeof(float);
  if (!ws || ws_bytes < need) return PE_E_WORKSPACE;
  hipLaunchKernelGGL((gemm_tn_kernel<BM, BN, MODE, TA>), dim3(tm * tn * splits), dim3(256), 0, st, al, bl, ws, (long)N,
                     (long)M * N, M, N, K, kps, tn, 0, amax_a, amax_b);
  PE_LAUNCH_CHECK();
  const long total = (long)M * N;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(pe_cdiv(total / 4, 256)), dim3(256), 0, st, ws, (long)M * N, splits,
                     C, ldc, M, N, accumulate);
  PE_LAUNCH_CHECK();
  return PE_OK;
}

}  // namespace

template <int MODE, class TA = float>
static int gemm_nt_impl(const TA* A, long lda, const float* B, long ldb, TA* C, long ldc, int M, int N,
                        int K, const float* bias0, const float* bias1, int accumulate, void* stream,
                        const unsigned* amax_a = nullptr, const unsigned* amax_b = nullptr) {
  if (!A || !B || !C || M < 0 || N < 0 || K <= 0) return PE_E_ARG;
  if (MODE == kSplit2 && (!amax_a || !amax_b)) return PE_E_ARG;
  if (M == 0 || N == 0) return PE_OK;
  if ((K & 3) || (lda & 3) || (ldb & 3) || (reinterpret_cast<uintptr_t>(A) & (4 * sizeof(TA) - 1)) || !aligned16(B))
    return PE_E_UNSUPPORTED;
  // the operand loaders address a tile's rows through 32-bit buffer offsets (gemm_engine.h, RowLoaderT)
  if (lda < 0 || ldb < 0 || lda >= (1L << 21) || ldb >= (1L << 21)) return PE_E_UNSUPPORTED;
  RowLoaderT<TA> al{A, lda, M, K, 0};
  RowLoader bl{B, ldb, N, K, 0};
  StoreEpiT<TA> ep{C, ldc, bias0, bias1, M, N, accumulate};
  hipStream_t st = pe_stream(stream);
  if (N <= 32) return launch_nt<Tile<128, 32, 4, 1>, MODE, TA>(al, bl, ep, M, N, K, st, amax_a, amax_b);
  if (N <= 64) return launch_nt<Tile<256, 64, 4, 1>, MODE, TA>(al, bl, ep, M, N, K, st, amax_a, amax_b);
  if (N % 192 == 0 && (N % 128 != 0 || MODE != kNative))   // 16-bit-term modes: the wider tile stages 17 % fewer rows per MFMA
    return launch_nt<Tile<128, 192, 2, 2>, MODE, TA>(al, bl, ep, M, N, K, st, amax_a, amax_b);
  return launch_nt<Tile<128, 128, 2, 2>, MODE, TA>(al, bl, ep, M, N, K, st, amax_a, amax_b);
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
// mixed precision with bf16 ACTIVATION STORAGE: A and C are bf16 tensors in HBM (weights and biases stay fp32)
extern "C" int pe_gemm_nt_bf16_a16(const void* A, long lda, const float* B, long ldb, void* C, long ldc, int M, int N,
                                   int K, const float* bias0, const float* bias1, int accumulate, void* stream) {
  return gemm_nt_impl<kBf16, act16_t>(static_cast<const act16_t*>(A), lda, B, ldb, static_cast<act16_t*>(C), ldc, M, N,
                                      K, bias0, bias1, accumulate, stream);
}
#endif

#ifndef PE_F16_BUILD
extern "C" int pe_gemm_nt_x3(const float* A, long lda, const float* B, long ldb, float* C, long ldc, int M, int N,
                             int K, const float* bias0, const float* bias1, int accumulate, void* stream) {
  return gemm_nt_impl<kSplit>(A, lda, B, ldb, C, ldc, M, N, K, bias0, bias1, accumulate, stream);
}

extern "C" int pe_gemm_nt_h2(const float* A, long lda, const float* B, long ldb, float* C, long ldc, int M, int N,
                             int K, const float* bias0, const float* bias1, int accumulate, const unsigned* amax_a,
                             const unsigned* amax_b, void* stream) {
  return gemm_nt_impl<kSplit2>(A, lda, B, ldb, C, ldc, M, N, K, bias0, bias1, accumulate, stream, amax_a, amax_b);
}
#endif

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

template <int MODE, class TA = float>
static int gemm_tn_impl(const TA* A, long lda, const TA* B, long ldb, float* C, long ldc, int M, int N,
                        int K, int accumulate, float* workspace, size_t workspace_bytes, void* stream,
                        const unsigned* amax_a = nullptr, const unsigned* amax_b = nullptr) {
  if (!A || !B || !C || M < 0 || N < 0 || K <= 0) return PE_E_ARG;
  if (MODE == kSplit2 && (!amax_a || !amax_b)) return PE_E_ARG;
  if (M == 0 || N == 0) return PE_OK;
  if ((M & 3) || (N & 3) || (lda & 3) || (ldb & 3) || (reinterpret_cast<uintptr_t>(A) & (4 * sizeof(TA) - 1)) ||
      (reinterpret_cast<uintptr_t>(B) & (4 * sizeof(TA) - 1)))
    return PE_E_UNSUPPORTED;
  if (lda < 0 || ldb < 0 || lda >= (1L << 24) || ldb >= (1L << 24)) return PE_E_UNSUPPORTED;   // 32-bit offsets per k-tile
  hipStream_t st = pe_stream(stream);
  if (M <= 64 && N <= 64)
    return launch_tn<64, 64, MODE, TA>(A, lda, B, ldb, C, ldc, M, N, K, accumulate, workspace, workspace_bytes, st, amax_a,
                                   amax_b);
  if (M <= 64)
    return launch_tn<64, 128, MODE, TA>(A, lda, B, ldb, C, ldc, M, N, K, accumulate, workspace, workspace_bytes, st, amax_a,
                                    amax_b);
  if (N <= 64)
    return launch_tn<128, 64, MODE, TA>(A, lda, B, ldb, C, ldc, M, N, K, accumulate, workspace, workspace_bytes, st, amax_a,
                                    amax_b);
  return launch_tn<128, 128, MODE, TA>(A, lda, B, ldb, C, ldc, M, N, K, accumulate, workspace, workspace_bytes, st, amax_a,
                                   amax_b);
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

extern "C" int pe_gemm_tn_h2(const float* A, long lda, const float* B, long ldb, float* C, long ldc, int M, int N,
                             int K, int accumulate, float* workspace, size_t workspace_bytes, const unsigned* amax_a,
                             const unsigned* amax_b, void* stream) {
  return gemm_tn_impl<kSplit2>(A, lda, B, ldb, C, ldc, M, N, K, accumulate, workspace, workspace_bytes, stream, amax_a,
                               amax_b);
}
#endif

extern "C" int PE_HALF(pe_gemm_tn)(const float* A, long lda, const float* B, long ldb, float* C, long ldc, int M, int N,
                               int K, int accumulate, float* workspace, size_t workspace_bytes, void* stream) {
  return gemm_tn_impl<kBf16>(A, lda, B, ldb, C, ldc, M, N, K, accumulate, workspace, workspace_bytes, stream);
}

#ifndef PE_F16_BUILD
extern "C" int pe_gemm_tn_bf16_a16(const void* A, long lda, const void* B, long ldb, float* C, long ldc, int M, int N,
                                   int K, int accumulate, float* workspace, size_t workspace_bytes, void* stream) {
  return gemm_tn_impl<kBf16, act16_t>(static_cast<const act16_t*>(A), lda, static_cast<const act16_t*>(B), ldb, C, ldc,
                                      M, N, K, accumulate, workspace, workspace_bytes, stream);
}
#endif
