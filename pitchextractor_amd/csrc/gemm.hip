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

// ---- NT, three-term split, software-pipelined: 256 x 128 tile, ONE workgroup per CU
// The kernels above alternate a staging phase (split + LDS stores, matrix pipe idle) with an MFMA phase between two
// barriers per 32-k tile and rely on a second resident workgroup to fill the gaps; measured, the pair gets ~10 k cycles
// per k-tile where the MFMAs alone need 4.6 k.  Here one wave per SIMD owns the whole register file and the whole
// schedule: tile kt + 1 is split and stored into the other LDS stage, and tile kt + 2 is requested from memory, in
// small pieces BETWEEN the 96 MFMAs of tile kt (one piece per two MFMAs, fragment reads on the odd ones; an MFMA
// holds the vector issue port for 8 of its 32 cycles, so ~5 single-issue instructions ride in its shadow -- MI355X
// microarchitecture guide, constants table).  __builtin_amdgcn_sched_barrier(0) after every MFMA pins that order.
// One barrier per k-tile.  Wave tile 128 x 64 (4 x 2 blocks): 3.6 filler instructions per MFMA instead of ~5.
// Accumulators are transposed (operands swapped) for the 16-byte epilogue; products and their order are those of
// nt_mainloop_split, so results are bit-identical to gemm_nt_t_kernel.
// MEASURED (M 49152, N 1536, K 768; 9 rounds of 256 tiles): 0.73 ms = the two-workgroup kernel's time.  Per tile:
// main loop 2.7 us per k-tile (~52 cycles per MFMA at the ~1.85 GHz the chip holds under this load; with every
// filler removed the loop still needs ~39: one barrier + an exposed fragment read per k-tile), of which the split
// pieces are 20 %, fragment reads 5 %, operand requests 5 %; epilogue 16 us (33 MB written by all CUs at once, nobody
// computing meanwhile -- with one workgroup per CU nothing overlaps it); prologue 6 us.  Kept as an opt-in
// (PE_GEMM_NT_PIPE=1): what it needs next is a persistent tile loop whose epilogue stores ride in the next tile's
// MFMA shadows (a second accumulator set fits: 293 of 512 registers used).
__global__ __launch_bounds__(256, 1) void gemm_nt_pipe_kernel(RowLoader al, RowLoader bl, StoreEpi ep, int K,
                                                              int tiles_m, int tiles_n) {
  constexpr int BM = 256, BN = 128, AIMG = BM * 32, BIMG = BN * 32;   // bf16 elements per term image
  constexpr int STAGE = 3 * (AIMG + BIMG);
  constexpr int UA = BM / 32, UB = BN / 32, U = UA + UB;               // staging units (float4 per thread) per k-tile
  constexpr int NM = 4, NN = 2;                                        // 32 x 32 blocks per wave: rows, columns
  constexpr int NQ = 2 * NM * NN * 6;                                  // MFMAs per k-tile and wave
  static_assert(NQ / 2 == 4 * U && U == 12, "filler schedule below");
  extern __shared__ __attribute__((aligned(16))) __bf16 lds[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wm = wv >> 1, wn = wv & 1, r = lane & 31, h = lane >> 5;
  const int tile = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
  const int srow = tid >> 3, piece = tid & 7;
  const int nk = K / kBK;                                              // host: K % 32 == 0
  // Branch-free operand loads: one buffer instruction each, rows past the matrix fall outside the descriptor and
  // read as zero (host: each operand spans less than 4 GiB).  No control flow in the loop body, so hipcc counts
  // its vmcnt waits exactly.
  const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(al.p), 0, (unsigned)(((long)al.rows - 1) * al.ld + K) * 4u, 0x00020000);
  const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(bl.p), 0, (unsigned)(((long)bl.rows - 1) * bl.ld + K) * 4u, 0x00020000);
  const unsigned voa = (unsigned)((long)(m0 + srow) * al.ld + piece * 4) * 4u;
  const unsigned vob = (unsigned)((long)(n0 + srow) * bl.ld + piece * 4) * 4u;
  const unsigned ustep_a = (unsigned)(32 * al.ld) * 4u, ustep_b = (unsigned)(32 * bl.ld) * 4u;
  auto ldu = [&](int u, int kt) {
    const u32x4 d = u < UA ? __builtin_amdgcn_raw_buffer_load_b128(ars, voa + (unsigned)u * ustep_a, (unsigned)kt * 128u, 0)
                           : __builtin_amdgcn_raw_buffer_load_b128(brs, vob + (unsigned)(u - UA) * ustep_b,
                                                                   (unsigned)kt * 128u, 0);
    return make_float4(__uint_as_float(d.x), __uint_as_float(d.y), __uint_as_float(d.z), __uint_as_float(d.w));
  };
  auto unit_off = [&](int u) {                                         // element offset of the unit's hi-term store
    return u < UA ? swz_off(srow + 32 * u, piece >> 1) + (piece & 1) * 4
                  : 3 * AIMG + swz_off(srow + 32 * (u - UA), piece >> 1) + (piece & 1) * 4;
  };
  float4 ra[U], rb[U];                               // tile kt + 1 (being staged), tile kt + 2 (in flight)
#pragma unroll
  for (int u = 0; u < U; ++u) ra[u] = ldu(u, 0);
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const Split3 sp = split3(ra[u]);
    __bf16* d = lds + unit_off(u);
    const int img = u < UA ? AIMG : BIMG;
    *reinterpret_cast<uint2*>(d) = sp.hi;
    *reinterpret_cast<uint2*>(d + img) = sp.mid;
    *reinterpret_cast<uint2*>(d + 2 * img) = sp.lo;
  }
#pragma unroll
  for (int u = 0; u < U; ++u) ra[u] = ldu(u, 1);
  __syncthreads();

  f32x16 acc[NN][NM];
#pragma unroll
  for (int j = 0; j < NN; ++j)
#pragma unroll
    for (int i = 0; i < NM; ++i)
#pragma unroll
      for (int g = 0; g < 16; ++g) acc[j][i][g] = 0.f;

  const int a_row = wm * 128 + r, b_row = wn * 64 + r;
  float x[4], r1[4], r2[4];
  for (int kt = 0; kt < nk; ++kt) {
    const __bf16* cur = lds + (kt & 1) * STAGE;
    __bf16* nxt = lds + ((kt + 1) & 1) * STAGE;
    auto rd_a = [&](int kk, int i, int c) {
      return *reinterpret_cast<const bf16x8*>(cur + c * AIMG + swz_off(a_row + 32 * i, kk * 2 + h));
    };
    auto rd_b = [&](int kk, int j, int c) {
      return *reinterpret_cast<const bf16x8*>(cur + 3 * AIMG + c * BIMG + swz_off(b_row + 32 * j, kk * 2 + h));
    };
    // one staging piece: unit u, part 0 (first residual), 1 (second residual), 2 (pack + three LDS stores)
    auto stage_piece = [&](auto P) {
      constexpr int p = decltype(P)::value, u = p / 3, part = p % 3;
      if constexpr (part == 0) {
        x[0] = ra[u].x; x[1] = ra[u].y; x[2] = ra[u].z; x[3] = ra[u].w;
#pragma unroll
        for (int e = 0; e < 4; ++e) r1[e] = x[e] - trunc_bf16(x[e]);
      } else if constexpr (part == 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) r2[e] = r1[e] - trunc_bf16(r1[e]);
      } else {
        __bf16* d = nxt + unit_off(u);
        const int img = u < UA ? AIMG : BIMG;
        *reinterpret_cast<uint2*>(d) = make_uint2(pack_hi16(x[0], x[1]), pack_hi16(x[2], x[3]));
        *reinterpret_cast<uint2*>(d + img) = make_uint2(pack_hi16(r1[0], r1[1]), pack_hi16(r1[2], r1[3]));
        *reinterpret_cast<uint2*>(d + 2 * img) = make_uint2(pack_hi16(r2[0], r2[1]), pack_hi16(r2[2], r2[3]));
      }
    };
    bf16x8 fa[2][3], fb[2][NN][3];                   // [step parity], [kk parity]
#pragma unroll
    for (int c = 0; c < 3; ++c) fa[0][c] = rd_a(0, 0, c);
#pragma unroll
    for (int j = 0; j < NN; ++j)
#pragma unroll
      for (int c = 0; c < 3; ++c) fb[0][j][c] = rd_b(0, j, c);
    static_for<NQ>([&](auto Q) {                     // (a plain unrolled loop of this size is only partly unrolled)
      constexpr int q = decltype(Q)::value;
      constexpr int step = q / (NN * 6), kk = step / NM, i = step % NM, j = (q % (NN * 6)) / 6, t6 = q % 6;
      constexpr int pos = q % (NN * 6);
      {
        constexpr int ta[6] = {0, 2, 1, 0, 1, 0}, tb[6] = {2, 0, 1, 1, 0, 0};     // (weight term, activation term)
        acc[j][i] = mfma_bf16(fb[kk & 1][j][ta[t6]], fa[step & 1][tb[t6]], acc[j][i]);
      }
      // ---- fillers.  Even MFMAs: first half of the tile alternates the 12 requests for tile kt + 2 with staging
      // pieces 0 .. 11, second half carries pieces 12 .. 35.  Odd MFMAs: fragment reads of the next step, and, where
      // those leave room in the second half, the hand-over rb -> ra of a unit whose part 0 has consumed ra.
      if constexpr ((q & 1) == 0) {
        constexpr int slot = q / 2;
        if constexpr (slot < 24) {
          if constexpr ((slot & 1) == 0) rb[slot / 2] = ldu(slot / 2, kt + 2);
          else stage_piece(std::integral_constant<int, slot / 2>{});
        } else {
          stage_piece(std::integral_constant<int, slot - 12>{});
        }
      } else {
        if constexpr (step + 1 < 2 * NM) {
          constexpr int ns = step + 1, nkk = ns / NM, ni = ns % NM;
          if constexpr (pos == 1 || pos == 3 || pos == 5) fa[ns & 1][pos / 2] = rd_a(nkk, ni, pos / 2);
          if constexpr (ni == 0 && pos >= 7) {       // next step opens a new kk: its weight fragments too
            constexpr int f = (pos - 7) / 2 * 2;     // 0, 2, 4
            fb[nkk & 1][f / 3][f % 3] = rd_b(nkk, f / 3, f % 3);
            fb[nkk & 1][(f + 1) / 3][(f + 1) % 3] = rd_b(nkk, (f + 1) / 3, (f + 1) % 3);
          }
        }
        constexpr int mv = q == 55 ? 0 : q == 57 ? 1 : q == 59 ? 2 : q == 67 ? 3 : q == 69 ? 4 : q == 71 ? 5
                         : q == 79 ? 6 : q == 81 ? 7 : q == 83 ? 8 : q == 85 ? 9 : q == 87 ? 10 : q == 91 ? 11 : -1;
        if constexpr (mv >= 0) ra[mv] = rb[mv];
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    __syncthreads();                                 // stage nxt is complete, stage cur is free
  }

#pragma unroll
  for (int j = 0; j < NN; ++j)
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      const int col = n0 + wn * 64 + j * 32 + 8 * qd + 4 * h;
      if (col >= ep.N) continue;
      float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0;
      if (ep.bias0) b0 = make_float4(ep.bias0[col], ep.bias0[col + 1], ep.bias0[col + 2], ep.bias0[col + 3]);
      if (ep.bias1) b1 = make_float4(ep.bias1[col], ep.bias1[col + 1], ep.bias1[col + 2], ep.bias1[col + 3]);
#pragma unroll
      for (int i = 0; i < NM; ++i) {
        const int row = m0 + wm * 128 + i * 32 + r;
        if (row >= ep.M) continue;
        float4* dst = reinterpret_cast<float4*>(ep.C + (long)row * ep.ldc + col);
        float4 v = make_float4(acc[j][i][4 * qd], acc[j][i][4 * qd + 1], acc[j][i][4 * qd + 2], acc[j][i][4 * qd + 3]);
        if (ep.bias0) { v.x += b0.x; v.y += b0.y; v.z += b0.z; v.w += b0.w; }
        if (ep.bias1) { v.x += b1.x; v.y += b1.y; v.z += b1.z; v.w += b1.w; }
        if (ep.accumulate) { const float4 o = *dst; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
        *dst = v;
      }
    }
}

int launch_nt_pipe(const RowLoader& al, const RowLoader& bl, const StoreEpi& ep, int M, int N, int K, hipStream_t st) {
  constexpr int kLds = 2 * 3 * (256 + 128) * 32 * 2;
  static bool attr = false;
  if (!attr) {
    PE_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_pipe_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, kLds));
    attr = true;
  }
  const int tm = pe_cdiv(M, 256), tn = pe_cdiv(N, 128);
  hipLaunchKernelGGL(gemm_nt_pipe_kernel, dim3(tm * tn), dim3(256), kLds, st, al, bl, ep, K, tm, tn);
  PE_LAUNCH_CHECK();
  return PE_OK;
}

template <class TL, int MODE, class TA = float>
int launch_nt(const RowLoaderT<TA>& al, const RowLoader& bl, const StoreEpiT<TA>& ep, int M, int N, int K,
              hipStream_t st, const unsigned* amax_a = nullptr, const unsigned* amax_b = nullptr) {
  const int tm = pe_cdiv(M, TL::BM), tn = pe_cdiv(N, TL::BN);
  static const bool off = getenv("PE_GEMM_NT_SCALAR_EPILOGUE") != nullptr;     // A/B switch (tools/ab_gemm.py)
  const bool vec = !off && MODE != kNative && (N & 3) == 0 && (ep.ldc & 3) == 0 &&
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

static bool g_nt_pipeline = getenv("PE_GEMM_NT_PIPE") != nullptr && getenv("PE_GEMM_NT_PIPE")[0] == '1';

#ifndef PE_F16_BUILD
extern "C" int pe_gemm_nt_pipeline(int enable) {
  const int old = g_nt_pipeline ? 1 : 0;
  g_nt_pipeline = enable != 0;
  return old;
}
#endif

template <int MODE, class TA = float>
static int gemm_nt_impl(const TA* A, long lda, const float* B, long ldb, TA* C, long ldc, int M, int N,
                        int K, const float* bias0, const float* bias1, int accumulate, void* stream,
                        const unsigned* amax_a = nullptr, const unsigned* amax_b = nullptr) {
  if (!A || !B || !C || M < 0 || N < 0 || K <= 0) return PE_E_ARG;
  if (MODE == kSplit2 && (!amax_a || !amax_b)) return PE_E_ARG;
  if (M == 0 || N == 0) return PE_OK;
  if ((K & 3) || (lda & 3) || (ldb & 3) || (reinterpret_cast<uintptr_t>(A) & (4 * sizeof(TA) - 1)) || !aligned16(B))
    return PE_E_UNSUPPORTED;
  RowLoaderT<TA> al{A, lda, M, K, 0};
  RowLoader bl{B, ldb, N, K, 0};
  StoreEpiT<TA> ep{C, ldc, bias0, bias1, M, N, accumulate};
  hipStream_t st = pe_stream(stream);
  if constexpr (MODE == kSplit && std::is_same<TA, float>::value) {
    // opt-in (PE_GEMM_NT_PIPE=1): measured at par with the two-workgroup kernels below on the step's shapes --
    // see the kernel's comment and DESIGN.md for the breakdown
    const bool pipe_off = !g_nt_pipeline;
    const bool fits32 = ((size_t)M * lda + K) * 4 < (1ull << 32) && ((size_t)N * ldb + K) * 4 < (1ull << 32);
    if (!pipe_off && N % 128 == 0 && K >= 256 && K % 32 == 0 && M >= 256 && (ldc & 3) == 0 && aligned16(C) && fits32)
      return launch_nt_pipe(al, bl, ep, M, N, K, st);
  }
  if (N <= 32) return launch_nt<Tile<128, 32, 4, 1>, MODE, TA>(al, bl, ep, M, N, K, st, amax_a, amax_b);
  if (N <= 64) return launch_nt<Tile<256, 64, 4, 1>, MODE, TA>(al, bl, ep, M, N, K, st, amax_a, amax_b);
  static const bool narrow = getenv("PE_GEMM_NT_TILE128") != nullptr;                 // A/B switch
  if (N % 192 == 0 && (N % 128 != 0 || MODE != kNative) && !(narrow && N % 128 == 0))   // bf16-term modes: the wider tile stages 17 % fewer rows per MFMA
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
