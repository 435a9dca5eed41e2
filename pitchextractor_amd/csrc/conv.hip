// 3x3 / pad-1 convolutions of the JDCNet stack (model.py:23-28,157-161) as implicit GEMMs on the
// fp32 MFMA engine, activations channels-last [B][T][F][C].
//
//   forward / data-gradient:  Y[p][n] = sum_{tap,c} X[p + tap][c] * Wp[n][tap][c]   (NT engine,
//       im2col gathered on the fly by ConvLoader; dgrad is the same kernel on flipped, transposed
//       weights);
//   weight gradient:          dW[n][tap][c] = sum_p dY[p][n] * X[p + tap][c]         (TN engine,
//       pixels split across workgroups, per-split slabs reduced in a fixed order -> deterministic);
//   first layer (Cin = 1, model.py:24): 9 FMAs per output, HBM-bound on the 64-channel write.
#include <type_traits>
#include "gemm_engine.h"

namespace {
using namespace pe;
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------- weight repack (OIHW -> packed)
__global__ void repack3x3_kernel(const float* __restrict__ w, float* __restrict__ w_fwd,
                                 float* __restrict__ w_dgrad, int Cout, int Cin) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= Cout * Cin * 9) return;
  const int tap = idx % 9, ci = (idx / 9) % Cin, co = idx / (9 * Cin);
  const float v = w[idx];
  if (w_fwd) w_fwd[((long)co * 9 + tap) * Cin + ci] = v;
  if (w_dgrad) w_dgrad[((long)ci * 9 + (8 - tap)) * Cout + co] = v;   // (2-kh)*3 + (2-kw) = 8 - tap
}

__global__ void transpose2d_kernel(const float* __restrict__ in, float* __restrict__ out, int rows, int cols) {
  __shared__ float tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  for (int i = threadIdx.y; i < 32; i += 8) {
    const int r = r0 + i, c = c0 + threadIdx.x;
    tile[i][threadIdx.x] = (r < rows && c < cols) ? in[(long)r * cols + c] : 0.f;
  }
  __syncthreads();
  for (int i = threadIdx.y; i < 32; i += 8) {
    const int c = c0 + i, r = r0 + threadIdx.x;
    if (r < rows && c < cols) out[(long)c * rows + r] = tile[threadIdx.x][i];
  }
}

// ---------------------------------------------------------------- forward / dgrad
template <class TO>
struct ConvEpiT {
  TO* Y;
  int rows, N, accumulate;
  double* stats;          // optional [tiles_m][2][N]: per-tile column sums / sums of squares of the final outputs
  __device__ __forceinline__ void operator()(int row, int col, float v) const {
    if (row < rows && col < N) {
      TO* d = Y + (long)row * N + col;
      if (accumulate) v += ld1(d);
      st1(d, v);
    }
  }
};
typedef ConvEpiT<float> ConvEpi;

// the value a TO tensor holds after storing v (bf16: round to nearest even), so that BatchNorm statistics taken in
// the epilogue describe the stored tensor
template <class TO> __device__ __forceinline__ float stored_value(float v) {
  if constexpr (std::is_same<TO, float>::value) return v;
  else return __uint_as_float(pack_bf16_rne(v, 0.f) << 16);
}

template <class TL, int MODE, class TA = float>
__global__ __launch_bounds__(256) void conv3x3_kernel(ConvLoader<TL::A_LOADS, TA> al, RowLoader bl, ConvEpiT<TA> ep,
                                                      int K, int tiles_m, int tiles_n, const unsigned* amax_x,
                                                      const unsigned* amax_w) {
  __shared__ __attribute__((aligned(16))) float As[TL::BM * nt_row_floats<MODE>()];
  __shared__ __attribute__((aligned(16))) float Bs[TL::BN * nt_row_floats<MODE>()];
  const int tile = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int m0 = (tile / tiles_n) * TL::BM, n0 = (tile % tiles_n) * TL::BN;
  al.init(m0);
  bl.init(n0);
  f32x16 acc[TL::TM][TL::TN];
  zero_acc<TL>(acc);
  H2Scales hs{1.f, 1.f, 1.f};
  if constexpr (MODE == kSplit2) hs.load(amax_x, amax_w);
  nt_mainloop_mode<TL, MODE>(al, bl, K, As, Bs, acc, hs.sa, hs.sb);
  for_each_acc<TL>(acc, [&](int r, int c, float v) { ep(m0 + r, n0 + c, MODE == kSplit2 ? v * hs.inv : v); });
}

template <class TL, int MODE, class TA = float>
int launch_conv(const TA* x, const float* wp, TA* y, int B, int T, int F, int C, int N, int accumulate,
                hipStream_t st, const unsigned* amax_x = nullptr, const unsigned* amax_w = nullptr) {
  const int rows = B * T * F, K = 9 * C;
  ConvLoader<TL::A_LOADS, TA> al;
  al.p = x; al.T = T; al.F = F; al.C = C; al.rows = rows;
  RowLoader bl{wp, (long)K, N, K, 0};
  ConvEpiT<TA> ep{y, rows, N, accumulate, nullptr};
  const int tm = pe_cdiv(rows, TL::BM), tn = pe_cdiv(N, TL::BN);
  hipLaunchKernelGGL((conv3x3_kernel<TL, MODE, TA>), dim3(tm * tn), dim3(256), 0, st, al, bl, ep, K, tm, tn, amax_x,
                     amax_w);
  PE_LAUNCH_CHECK();
  return PE_OK;
}

// ---------------------------------------------------------------- forward / dgrad without im2col re-reads
// The implicit-GEMM kernel above fetches every input pixel nine times (once per tap); with ~100 KB of
// distinct rows in flight per tile the XCD's 4 MB L2 does not absorb that (PMC: 2.5-4x the algorithmic
// bytes reach the fabric).  This kernel stages, per 32-channel chunk, the whole input window of its 128
// output pixels once -- the 128 + 2F + 2 consecutive pixels [p0 - F - 1, p0 + 128 + F] -- and lets the nine
// taps read it at nine row offsets; only the weights change per tap.  A neighbour that the flat offset
// takes across an image border (other time row / other utterance) is redirected, per output pixel and
// tap, to an all-zero row of the image.
//
// bf16-term images in LDS are [row][32 k] with NO padding: the 16-byte chunk c of a row sits at chunk
// c ^ ((row >> 2) & 3), which keeps ds_read_b128 fragment reads (32 consecutive rows at any offset) and the
// ds_write_b64 staging stores conflict-free (MI355X_MICROARCH.md, LDS lane groups).  NT = 3 images per
// operand for the exact three-term split, 1 for mixed precision.
// PASSES = staged window rows / 32: 7 (F <= 47) for the wide tiles, 10 (F <= 95) for the 64-channel layers.

// ---------------------------------------------------------------- weights as MFMA fragments straight from L2
// Staging a 32-k weight slab per tap through LDS as well (the first form of this kernel) made every workgroup repeat
// the split of the same few hundred KB of weights, and each of the nine stages of a channel chunk cost two workgroup
// barriers although the activation window does not change.  Here the weights are packed ONCE per call
// (wfrag_pack_kernel) into the B-operand fragment order of v_mfma_f32_32x32x16_bf16 -- for every
// (16-k block, 32-row block, term) the 64 lanes' 16-byte pieces are contiguous (1 KB) -- and a wave loads
// its fragments directly into registers with one coalesced global_load_dwordx4 per fragment, a ring of
// D steps ahead of the MFMAs that consume them.  LDS holds only the activation window: two barriers per
// channel chunk (9 taps = 18 k-blocks) instead of per tap, no weight split, no weight LDS traffic.
//   fragment (kb, nb, c), lane l = 32 h + r  <->  B[n = 32 nb + r][k = 16 kb + 8 h .. + 7] of term c.
template <int NT>
__global__ void wfrag_pack_kernel(const float* __restrict__ w, long ld, int N, int K, uint4* __restrict__ out,
                                  const unsigned* __restrict__ amax = nullptr) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int NB32 = (N + 31) >> 5, KB = K >> 4;
  const int lane = (int)(idx & 63);
  const long blk = idx >> 6;
  if (blk >= (long)NB32 * KB) return;
  const int nb = (int)(blk % NB32), kb = (int)(blk / NB32);
  const int n = nb * 32 + (lane & 31), k = kb * 16 + (lane >> 5) * 8;
  float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
  if (n < N) {
    v0 = *reinterpret_cast<const float4*>(w + (long)n * ld + k);
    v1 = *reinterpret_cast<const float4*>(w + (long)n * ld + k + 4);
  }
  uint4* dst = out + ((long)(kb * NB32 + nb) * NT) * 64 + lane;
  if constexpr (NT == 3) {
    const Split3 a = split3(v0), b = split3(v1);
    dst[0] = make_uint4(a.hi.x, a.hi.y, b.hi.x, b.hi.y);
    dst[64] = make_uint4(a.mid.x, a.mid.y, b.mid.x, b.mid.y);
    dst[128] = make_uint4(a.lo.x, a.lo.y, b.lo.x, b.lo.y);
  } else if constexpr (NT == 2) {
    const float sc = h2_scale(*amax);
    const Split2 a = split2(v0, sc), b = split2(v1, sc);
    dst[0] = make_uint4(a.hi.x, a.hi.y, b.hi.x, b.hi.y);
    dst[64] = make_uint4(a.lo.x, a.lo.y, b.lo.x, b.lo.y);
  } else {
    const bf16x4 a = to_bf16x4(v0), b = to_bf16x4(v1);
    const uint2 ua = __builtin_bit_cast(uint2, a), ub = __builtin_bit_cast(uint2, b);
    dst[0] = make_uint4(ua.x, ua.y, ub.x, ub.y);
  }
}

template <int BN, int MODE, int PASSES, int D, bool FA2, class TA = float>
__global__ __launch_bounds__(256, 2) void conv3x3_halo_wf_kernel(const TA* __restrict__ x,
                                                                 const uint4* __restrict__ wf, ConvEpiT<TA> ep, int T,
                                                                 int F, int C, int N, int P, int tiles_m,
                                                                 int tiles_n, const unsigned* amax_x,
                                                                 const unsigned* amax_w) {
  constexpr int NT = mode_terms<MODE>();
  H2Scales hs{1.f, 1.f, 1.f};
  if constexpr (MODE == kSplit2) hs.load(amax_x, amax_w);
  constexpr int TM = 2, TN = BN / 64, WN = BN / 2;
  constexpr int ZR = PASSES * 32;                                  // an all-zero row behind the window
  constexpr int AIMG = (ZR + 4) * 32;                              // bf16 elements per image
  constexpr int S = 18 * TN;                                       // steps (tap, kk, j) per channel chunk
  static_assert(S % D == 0 && D >= 2, "the fragment ring wraps at chunk boundaries");
  __shared__ __attribute__((aligned(16))) __bf16 As[NT * AIMG];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wm = wv >> 1, wn = wv & 1, r = lane & 31, h = lane >> 5;
  const int tile = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int p0 = (tile / tiles_n) * 128, n0 = (tile % tiles_n) * BN;
  const int WR = 128 + 2 * F + 2;
  const int srow = tid >> 3, piece = tid & 7;
  const int nchunks = C / 32;

  unsigned vbits[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int p = p0 + wm * 64 + i * 32 + r;
    unsigned b = 0;
    if (p < P) {
      const int f = p % F, t = (p / F) % T;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int tt = t + tap / 3 - 1, ff = f + tap % 3 - 1;
        if (tt >= 0 && tt < T && ff >= 0 && ff < F) b |= 1u << tap;
      }
    }
    vbits[i] = b;
  }

  typedef typename RawQuad<TA>::type Raw;      // the staged window keeps its storage form until the LDS stores
  Raw ra[PASSES];
  // Branch-free loads: ONE buffer instruction per request, rows outside the window / the tensor point past the
  // descriptor's range and read as zero (host: every tensor below 2 GiB).  With predicated global loads hipcc branches
  // around each of them and can no longer count what is in flight: it waited `vmcnt(0)` in front of the first MFMA of
  // every channel chunk, i.e. for the window of the NEXT chunk it had just requested from HBM (seen in the .s).
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<TA*>(x), 0, (unsigned)((long)P * C * (long)sizeof(TA)), 0x00020000);
  unsigned voa[PASSES];
#pragma unroll
  for (int ps = 0; ps < PASSES; ++ps) {
    const int wr = ps * 32 + srow;
    const long q = (long)p0 - F - 1 + wr;
    voa[ps] = (wr < WR && q >= 0 && q < P) ? (unsigned)((q * C + piece * 4) * (long)sizeof(TA)) : 0x80000000u;
  }
  auto fetch_a = [&](int cc) {
    const unsigned so = (unsigned)(cc * 32 * (int)sizeof(TA));
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
      if constexpr (std::is_same<TA, float>::value) {
        const u32x4_t d = __builtin_amdgcn_raw_buffer_load_b128(xrs, voa[ps], so, 0);
        ra[ps] = make_float4(__uint_as_float(d.x), __uint_as_float(d.y), __uint_as_float(d.z), __uint_as_float(d.w));
      } else {
        const u32x2_t d = __builtin_amdgcn_raw_buffer_load_b64(xrs, voa[ps], so, 0);
        ra[ps] = make_uint2(d.x, d.y);
      }
    }
  };

  // weight fragments: byte offsets of this wave's n-blocks inside a 16-k block; a tile hanging over N re-reads the
  // last block (its columns are dropped by the epilogue)
  const int NB32 = (N + 31) >> 5, kb_tap = C >> 4;
  const int kb_stride = NB32 * NT * 64;                            // uint4 per 16-k block
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint4*>(wf), 0, (unsigned)((long)9 * kb_tap * kb_stride * 16), 0x00020000);
  unsigned vow[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int nb = (n0 + wn * WN + j * 32) >> 5;
    vow[j] = (unsigned)(((nb < NB32 ? nb : NB32 - 1) * NT * 64 + lane) * 16);
  }
  bf16x8 ring[D][NT];
  auto issue = [&](int slot, int cc, int s) {                      // s = (tap * 2 + kk) * TN + j
    const int j = s % TN, kk = (s / TN) & 1, tap = s / (2 * TN);
    const unsigned so = (unsigned)((tap * kb_tap + cc * 2 + kk) * kb_stride) * 16u;      // wave-uniform
#pragma unroll
    for (int c = 0; c < NT; ++c)
      ring[slot][c] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, vow[j], so + (unsigned)c * 1024u, 0));
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int g = 0; g < 16; ++g) acc[i][j][g] = 0.f;

  if (tid < 8) {
#pragma unroll
    for (int c = 0; c < NT; ++c) *reinterpret_cast<uint2*>(As + c * AIMG + ZR * 32 + tid * 4) = make_uint2(0u, 0u);
  }
  fetch_a(0);
#pragma unroll
  for (int s = 0; s < D - 1; ++s) issue(s, 0, s);

  auto load_fa = [&](bf16x8 (&fa)[TM][NT], int tap, int kk) {
    const int shift = (tap / 3) * F + tap % 3;                     // (F + 1) + (dt * F + df)
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int row = (vbits[i] >> tap) & 1u ? wm * 64 + i * 32 + r + shift : ZR;
#pragma unroll
      for (int c = 0; c < NT; ++c)
        fa[i][c] = *reinterpret_cast<const bf16x8*>(As + c * AIMG + swz_off(row, kk * 2 + h));
    }
  };

  for (int cc = 0; cc < nchunks; ++cc) {
    __syncthreads();                                               // every wave is done with the previous window
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) halo_store<NT>(As, AIMG, ps * 32 + srow, piece, ra[ps], hs.sa);
    __syncthreads();
    if (cc + 1 < nchunks) fetch_a(cc + 1);
    // FA2: the activation fragments of the next k-block are read under this block's MFMAs (24 more registers)
    bf16x8 fa[FA2 ? 2 : 1][TM][NT];
    if (FA2) load_fa(fa[0], 0, 0);
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const int j = s % TN, blk = s / TN;                          // blk = tap * 2 + kk
      {                                                            // keep the ring D - 1 steps ahead
        const int sn = s + D - 1;
        if (sn < S) issue(sn % D, cc, sn);
        else if (cc + 1 < nchunks) issue(sn % D, cc + 1, sn - S);
      }
      if (FA2) {
        if (j == 0 && blk + 1 < 18) load_fa(fa[(blk + 1) & 1], (blk + 1) >> 1, (blk + 1) & 1);
      } else if (j == 0) {
        load_fa(fa[0], blk >> 1, blk & 1);
      }
      const int fs = FA2 ? (blk & 1) : 0;
#pragma unroll
      for (int i = 0; i < TM; ++i) acc[i][j] = mfma_terms<NT>(fa[fs][i], ring[s % D], acc[i][j]);
      // hipcc otherwise sinks every prefetch down to its first use (seen in the .s: one step of lookahead
      // whatever D says); pinning the step boundaries keeps the loads D - 1 steps ahead of their MFMAs
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // ---- epilogue.  With ep.stats the BatchNorm statistics of the layer that consumes this output (model.py:25,37,
  // 150,159) are a by-product: per-column sum and sum of squares of the FINAL values (after the residual add) over
  // this tile's 128 pixels, in double, written as one partial per (pixel tile, column) -- the separate 1 GB
  // statistics pass over the activation disappears (pe_bn_finalize_stats sums the partials in a fixed order).
  double s1[TN], s2[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) s1[j] = s2[j] = 0.0;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const int row = p0 + wm * 64 + i * 32 + (g & 3) + 8 * (g >> 2) + 4 * h, col = n0 + wn * WN + j * 32 + r;
        if (row < ep.rows && col < ep.N) {
          TA* d = ep.Y + (long)row * ep.N + col;
          float v = MODE == kSplit2 ? acc[i][j][g] * hs.inv : acc[i][j][g];
          if (ep.accumulate) v += ld1(d);
          st1(d, v);
          v = stored_value<TA>(v);
          s1[j] += (double)v;
          s2[j] += (double)v * (double)v;
        }
      }
  if (ep.stats) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      s1[j] += __shfl_xor(s1[j], 32, 64);                          // the two lane halves hold different rows
      s2[j] += __shfl_xor(s2[j], 32, 64);
    }
    __syncthreads();                                               // the activation window is dead: reuse its space
    double* red = reinterpret_cast<double*>(As);                   // [2 (wm)][BN][2]
    if (h == 0) {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        red[(wm * BN + wn * WN + j * 32 + r) * 2] = s1[j];
        red[(wm * BN + wn * WN + j * 32 + r) * 2 + 1] = s2[j];
      }
    }
    __syncthreads();
    const long tile_m = tile / tiles_n;
    for (int c = tid; c < BN; c += 256) {
      if (n0 + c < ep.N) {
        ep.stats[(tile_m * 2) * ep.N + n0 + c] = red[c * 2] + red[(BN + c) * 2];
        ep.stats[(tile_m * 2 + 1) * ep.N + n0 + c] = red[c * 2 + 1] + red[(BN + c) * 2 + 1];
      }
    }
  }
}

template <int BN, int MODE, int PASSES, int D, bool FA2, class TA = float>
int launch_conv_halo_wf(const TA* x, const void* wf, TA* y, int B, int T, int F, int C, int N, int accumulate,
                        double* stats, hipStream_t st, const unsigned* amax_x = nullptr,
                        const unsigned* amax_w = nullptr) {
  const int P = B * T * F;
  ConvEpiT<TA> ep{y, P, N, accumulate, stats};
  const int tm = pe_cdiv(P, 128), tn = pe_cdiv(N, BN);
  hipLaunchKernelGGL((conv3x3_halo_wf_kernel<BN, MODE, PASSES, D, FA2, TA>), dim3(tm * tn), dim3(256), 0, st, x,
                     reinterpret_cast<const uint4*>(wf), ep, T, F, C, N, P, tm, tn, amax_x, amax_w);
  PE_LAUNCH_CHECK();
  return PE_OK;
}

static inline size_t pe_wfrag_bytes_host(int N, int K, int terms) { return (size_t)((N + 31) / 32) * (K / 16) * terms * 1024; }

// which staged-window variant serves (F, N): 7 passes + a 128/192-wide tile, 10 passes + a 64-wide tile
// (two workgroups per CU must fit the LDS), or 0 = use the implicit-GEMM kernel
int conv_halo_passes(int F, int N) {
  const int wr = 128 + 2 * F + 2;
  if (wr <= 7 * 32 && N >= 96) return 7;
  if (wr <= 10 * 32 && N <= 64) return 10;
  return 0;
}

// ---------------------------------------------------------------- weight gradient
template <int BM, int BN>
__global__ __launch_bounds__(256) void conv3x3_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            float* __restrict__ ws, int T, int F, int Cin,
                                                            int Cout, int P, int k_per_split, int tiles_n,
                                                            int n_tiles, int n_chunks) {
  __shared__ __attribute__((aligned(16))) float As[kBK * BM];
  __shared__ __attribute__((aligned(16))) float Bs[kBK * BN];
  // Block order: workgroups b and b+8 share an XCD (its L2).  The 9 taps of one (tile, pixel chunk)
  // are made consecutive workgroups of ONE XCD, so they stream the same dY / X lines at the same time
  // and HBM sees them once instead of 9 times (PMC: 17 GB -> per launch before this ordering).
  const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
  const int tap = q % 9;
  const int chunk = (q / 9) * 8 + xcd;
  if (chunk >= n_chunks) return;
  const int tile = chunk % n_tiles, split = chunk / n_tiles;
  const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
  const int kb = split * k_per_split;
  const int ke = min(P, kb + k_per_split);
  KRowLoader<BM> al{dy, (long)Cout, Cout, 0};
  ShiftedPixelLoader<BN> bl;
  bl.p = x; bl.T = T; bl.F = F; bl.C = Cin; bl.dt = tap / 3 - 1; bl.df = tap % 3 - 1;
  al.init(m0, kb);
  bl.init(n0, kb);
  f32x16 acc[BM / 64][BN / 64];
#pragma unroll
  for (int i = 0; i < BM / 64; ++i)
#pragma unroll
    for (int j = 0; j < BN / 64; ++j)
#pragma unroll
      for (int g = 0; g < 16; ++g) acc[i][j][g] = 0.f;
  tn_mainloop<BM, BN>(al, bl, kb, ke, As, Bs, acc);
  // slab layout: [split][tap][Cout][Cin]
  float* dst = ws + ((long)split * 9 + tap) * Cout * Cin;
  tn_for_each_acc<BM, BN>(acc, [&](int r, int c, float v) {
    const int co = m0 + r, ci = n0 + c;
    if (co < Cout && ci < Cin) dst[(long)co * Cin + ci] = v;
  });
}

// ---------------------------------------------------------------- weight gradient, all 9 taps per workgroup
// 64 (Cout) x 64 (Cin) tile, the 4 waves own its 32x32 quadrants and keep one accumulator per tap
// (9 x 16 registers).  Per 32-pixel k-tile the workgroup stages dY rows once and, for each row
// offset dt in {-1,0,1}, the 34 consecutive X rows [k0 + dt*F - 1, k0 + dt*F + 32]; the three
// column shifts df read the same window at row k + df + 1.  That is 134 staged rows for 9 x 32
// tap-rows of MFMA work (68 FLOP per staged byte instead of 16), and 64-wide tiles fit every layer
// (64/128/192/256 channels) without padding.  A source row is invalid for a tap when the shift crossed
// an image border: that only depends on the source pixel's own (t, f) and on (dt, df), so a small
// mask table Mk[window][row][df] is rebuilt per k-tile and multiplied into the X operand.
template <int DUMMY>
__global__ __launch_bounds__(256, 2) void conv3x3_wgrad9_kernel(const float* __restrict__ dy,
                                                                const float* __restrict__ x,
                                                                float* __restrict__ ws, int T, int F, int Cin,
                                                                int Cout, int P, int k_per_split, int tiles_n) {
  __shared__ __attribute__((aligned(16))) float Ys[kBK * 64];
  __shared__ __attribute__((aligned(16))) float Xs[3 * 34 * 64];
  __shared__ float Mk[3 * 34 * 4];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wm = wv >> 1, wn = wv & 1, r = lane & 31, h = lane >> 5;
  // 1-D grid over (split, tile), XCD-contiguous (as gemm_tn_kernel): the tiles of a pixel range share an XCD's L2
  const int tiles_mn = (Cout / 64) * tiles_n;
  const int lin = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_id = lin % tiles_mn, split_id = lin / tiles_mn;
  const int m0 = (tile_id / tiles_n) * 64, n0 = (tile_id % tiles_n) * 64;
  const int kb = split_id * k_per_split;
  const int ke = min(P, kb + k_per_split);

  f32x16 acc[9];
#pragma unroll
  for (int tp = 0; tp < 9; ++tp)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[tp][q] = 0.f;

  // staging assignments: dY 32 rows x 16 float4 (2 per thread); X 102 rows x 16 float4 (up to 7 per thread)
  const int c4 = (tid & 15) * 4;
  float4 ry[2], rx[7];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int k = k0 + (tid >> 4) + 16 * i;
      ry[i] = k < ke ? *reinterpret_cast<const float4*>(dy + (long)k * Cout + m0 + c4)
                     : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const int wr = (tid >> 4) + 16 * i;                 // window-row id 0..101 (w = wr / 34, row = wr % 34)
      rx[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (wr < 102) {
        const int w = wr / 34, row = wr - w * 34;
        const long q = (long)k0 + (long)(w - 1) * F + row - 1;
        if (q >= 0 && q < P) rx[i] = *reinterpret_cast<const float4*>(x + q * Cin + n0 + c4);
      }
    }
  };
  fetch(kb);
  for (int k0 = kb; k0 < ke; k0 += kBK) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) *reinterpret_cast<float4*>(Ys + ((tid >> 4) + 16 * i) * 64 + c4) = ry[i];
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const int wr = (tid >> 4) + 16 * i;
      if (wr < 102) *reinterpret_cast<float4*>(Xs + wr * 64 + c4) = rx[i];
    }
    if (tid < 102) {                                       // border masks of this k-tile's source rows
      const int w = tid / 34, row = tid - w * 34;
      const long q = (long)k0 + (long)(w - 1) * F + row - 1;
      float m0v = 0.f, m1v = 0.f, m2v = 0.f;
      if (q >= 0 && q < P) {
        const int fq = (int)(q % F), tq = (int)((q / F) % T);
        const bool trow = !((w == 2 && tq == 0) || (w == 0 && tq == T - 1));   // dt = w - 1
        m0v = (trow && fq != F - 1) ? 1.f : 0.f;          // df = -1: source column F-1 means the shift wrapped
        m1v = trow ? 1.f : 0.f;                            // df =  0
        m2v = (trow && fq != 0) ? 1.f : 0.f;              // df = +1
      }
      Mk[tid * 4 + 0] = m0v; Mk[tid * 4 + 1] = m1v; Mk[tid * 4 + 2] = m2v;
    }
    __syncthreads();
    if (k0 + kBK < ke) fetch(k0 + kBK);
    const float* a_rd = Ys + wm * 32 + r;
    const float* b_rd = Xs + wn * 32 + r;
#pragma unroll 4
    for (int s = 0; s < kBK / 2; ++s) {
      const int k = 2 * s + h;
      const float a = a_rd[k * 64];
#pragma unroll
      for (int w = 0; w < 3; ++w)
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          const int row = k + d;                           // k + df + 1 with df = d - 1
          const float b = b_rd[(w * 34 + row) * 64] * Mk[(w * 34 + row) * 4 + d];
          acc[w * 3 + d] = mfma32(a, b, acc[w * 3 + d]);
        }
    }
  }
  // slab layout: [split][tap][Cout][Cin]
#pragma unroll
  for (int tp = 0; tp < 9; ++tp) {
    float* dst = ws + ((long)split_id * 9 + tp) * Cout * Cin;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int co = m0 + wm * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
      const int ci = n0 + wn * 32 + r;
      dst[(long)co * Cin + ci] = acc[tp][q];
    }
  }
}

// The same 9-tap tile on the bf16 MFMA pipe with the exact three-term split (gemm_engine.h): dY and the
// X windows are staged as three [row][64 + 32 pad] bf16 images each, MFMA fragments (8 consecutive pixels
// of one channel) come from ds_read_b64_tr_b16, and the border mask becomes a 16-bit AND mask per pixel:
// Mk16[d][w][k] covers source row k + d of window w for column shift df = d - 1, so the 8 masks of a
// fragment are one aligned 16-byte read.
template <int NT, class TA = float>   // NT: 3 = exact three-term split, 2 = two scaled fp16 terms, 1 = operands rounded to bf16
__global__ __launch_bounds__(256, 2) void conv3x3_wgrad9_x3_kernel(const TA* __restrict__ dy,
                                                                   const TA* __restrict__ x,
                                                                   float* __restrict__ ws, int T, int F, int Cin,
                                                                   int Cout, int P, int k_per_split, int tiles_n,
                                                                   const unsigned* amax_dy, const unsigned* amax_x) {
  H2Scales hs{1.f, 1.f, 1.f};
  if constexpr (NT == 2) hs.load(amax_dy, amax_x);
  constexpr int ST = 96;                                   // bf16 elements per staged row
  constexpr int YIMG = kBK * ST, XIMG = 102 * ST;
  __shared__ __attribute__((aligned(16))) __bf16 Ys[NT * YIMG];
  __shared__ __attribute__((aligned(16))) __bf16 Xs[NT * XIMG];
  __shared__ __attribute__((aligned(16))) unsigned short Mk16[9 * kBK];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wm = wv >> 1, wn = wv & 1, r = lane & 31, h = lane >> 5;
  const int g1 = (lane >> 4) & 1, q4 = (lane & 15) >> 2, p4 = (lane & 3) * 4;
  // 1-D grid over (split, tile), XCD-contiguous (as gemm_tn_kernel): the tiles of a pixel range share an XCD's L2
  const int tiles_mn = (Cout / 64) * tiles_n;
  const int lin = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_id = lin % tiles_mn, split_id = lin / tiles_mn;
  const int m0 = (tile_id / tiles_n) * 64, n0 = (tile_id % tiles_n) * 64;
  const int kb = split_id * k_per_split;
  const int ke = min(P, kb + k_per_split);

  f32x16 acc[9];
#pragma unroll
  for (int tp = 0; tp < 9; ++tp)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[tp][q] = 0.f;

  const int c4 = (tid & 15) * 4;
  typedef typename RawQuad<TA>::type Raw;
  Raw ry[2], rx[7];
  // Branch-free requests (see RowLoaderT): the descriptors are re-based per k-tile in scalar registers -- dY at pixel
  // k0 with the rows left in this split as its range, X at the first pixel of the three-row window (clamped to the
  // tensor; offsets of pixels in front of it wrap to huge unsigned values) -- so every out-of-range pixel reads as zero.
  constexpr unsigned kEsz = (unsigned)sizeof(TA);
  unsigned voy[2], vox[7];
#pragma unroll
  for (int i = 0; i < 2; ++i) voy[i] = (unsigned)(((tid >> 4) + 16 * i) * Cout + m0 + c4) * kEsz;
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const int wr = (tid >> 4) + 16 * i;
    const int w = wr / 34, row = wr - w * 34;
    vox[i] = wr < 102 ? (unsigned)((w * F + row) * Cin + n0 + c4) * kEsz : 0x80000000u;
  }
  auto fetch = [&](int k0) {
    const int rows_y = min(ke - k0, kBK);
    const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<TA*>(dy + (long)k0 * Cout), 0, (unsigned)(rows_y > 0 ? rows_y * Cout : 0) * kEsz, 0x00020000);
#pragma unroll
    for (int i = 0; i < 2; ++i) ry[i] = ldraw_buffer<TA>(rsy, voy[i], 0u);
    const int first = k0 - F - 1, base_row = first > 0 ? first : 0;
    const int nrows = min(P - base_row, 2 * F + 36);
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<TA*>(x + (long)base_row * Cin), 0, (unsigned)(nrows > 0 ? nrows * Cin : 0) * kEsz, 0x00020000);
    const unsigned dlt = (unsigned)((first - base_row) * Cin) * kEsz;        // <= 0: wraps
#pragma unroll
    for (int i = 0; i < 7; ++i) rx[i] = ldraw_buffer<TA>(rsx, vox[i] + dlt, 0u);
  };
  auto store3 = [&](__bf16* img, int img_elems, int off, const Raw& raw, float scale) {
    if constexpr (NT == 1 && !std::is_same<TA, float>::value) {
      *reinterpret_cast<Raw*>(img + off) = raw;              // bf16 tensor, one rounded term: the bits as they are
      return;
    }
    const float4 v = widen(raw);
    if constexpr (NT == 3) {
      const Split3 sp = split3(v);
      *reinterpret_cast<uint2*>(img + off) = sp.hi;
      *reinterpret_cast<uint2*>(img + off + img_elems) = sp.mid;
      *reinterpret_cast<uint2*>(img + off + 2 * img_elems) = sp.lo;
    } else if constexpr (NT == 2) {
      const Split2 sp = split2(v, scale);
      *reinterpret_cast<uint2*>(img + off) = sp.hi;
      *reinterpret_cast<uint2*>(img + off + img_elems) = sp.lo;
    } else {
      *reinterpret_cast<bf16x4*>(img + off) = to_bf16x4(v);
    }
  };
  fetch(kb);
  const __bf16* a_rd = Ys + (8 * h + q4) * ST + wm * 32 + 16 * g1 + p4;
  const __bf16* b_rd = Xs + (8 * h + q4) * ST + wn * 32 + 16 * g1 + p4;
  for (int k0 = kb; k0 < ke; k0 += kBK) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) store3(Ys, YIMG, ((tid >> 4) + 16 * i) * ST + c4, ry[i], hs.sa);
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const int wr = (tid >> 4) + 16 * i;
      if (wr < 102) store3(Xs, XIMG, wr * ST + c4, rx[i], hs.sb);
    }
    if (tid < 102) {                                       // border masks of this k-tile's source rows
      const int w = tid / 34, row = tid - w * 34;
      const long q = (long)k0 + (long)(w - 1) * F + row - 1;
      bool mv[3] = {false, false, false};
      if (q >= 0 && q < P) {
        const int fq = (int)(q % F), tq = (int)((q / F) % T);
        const bool trow = !((w == 2 && tq == 0) || (w == 0 && tq == T - 1));   // dt = w - 1
        mv[0] = trow && fq != F - 1;                       // df = -1: source column F-1 means the shift wrapped
        mv[1] = trow;
        mv[2] = trow && fq != 0;                           // df = +1
      }
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        const int k = row - d;
        if (k >= 0 && k < kBK) Mk16[(d * 3 + w) * kBK + k] = mv[d] ? 0xffffu : 0u;
      }
    }
    __syncthreads();
    if (k0 + kBK < ke) fetch(k0 + kBK);
#pragma unroll
    for (int kk = 0; kk < kBK / 16; ++kk) {
      bf16x8 fa[NT];
#pragma unroll
      for (int c = 0; c < NT; ++c) fa[c] = tr_fragment(a_rd + c * YIMG + kk * 16 * ST, ST);
#pragma unroll
      for (int w = 0; w < 3; ++w)
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          const uint4 mk = *reinterpret_cast<const uint4*>(Mk16 + (d * 3 + w) * kBK + kk * 16 + 8 * h);
          bf16x8 fb[NT];
#pragma unroll
          for (int c = 0; c < NT; ++c) {
            uint4 v = __builtin_bit_cast(uint4, tr_fragment(b_rd + c * XIMG + (w * 34 + kk * 16 + d) * ST, ST));
            v.x &= mk.x; v.y &= mk.y; v.z &= mk.z; v.w &= mk.w;
            fb[c] = __builtin_bit_cast(bf16x8, v);
          }
          acc[w * 3 + d] = mfma_terms<NT>(fa, fb, acc[w * 3 + d]);
        }
    }
  }
#pragma unroll
  for (int tp = 0; tp < 9; ++tp) {
    float* dst = ws + ((long)split_id * 9 + tp) * Cout * Cin;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int co = m0 + wm * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
      const int ci = n0 + wn * 32 + r;
      dst[(long)co * Cin + ci] = NT == 2 ? acc[tp][q] * hs.inv : acc[tp][q];
    }
  }
}

// sum slabs in split order and scatter to OIHW: dw[(co*Cin + ci)*9 + tap]
__global__ void conv3x3_wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int splits,
                                            int Cout, int Cin) {
  const int i4 = blockIdx.x * blockDim.x + threadIdx.x;        // float4 index over [tap][co][ci] (Cin % 4 == 0)
  const int n = 9 * Cout * Cin;
  if (i4 * 4 >= n) return;
  const float4 s = pe_ordered_slab_sum4(ws, n, splits, i4);
  const int idx = i4 * 4;
  const int ci = idx % Cin, co = (idx / Cin) % Cout, tap = idx / (Cin * Cout);
  float* d = dw + ((long)co * Cin + ci) * 9 + tap;
  d[0] = s.x; d[9] = s.y; d[18] = s.z; d[27] = s.w;
}

bool wgrad9_ok(int Cout, int Cin) { return (Cout % 64) == 0 && (Cin % 64) == 0; }

void wgrad_plan(int P, int Cout, int Cin, int* bm, int* bn, int* splits, int* kps) {
  const bool nine = wgrad9_ok(Cout, Cin);
  *bm = (nine || Cout <= 64) ? 64 : 128;
  *bn = (nine || Cin <= 64) ? 64 : 128;
  const int tiles = pe_cdiv(Cout, *bm) * pe_cdiv(Cin, *bn) * (nine ? 1 : 9);
  const int s = pe_pick_splits(tiles, P, 1024, nine ? 512 : 768);   // resident: 2 (9-tap) or 3 workgroups per CU
  int k = pe_cdiv(P, s);
  k = (k + kBK - 1) / kBK * kBK;
  *kps = k;
  *splits = pe_cdiv(P, k);
}

template <int BM, int BN>
int launch_wgrad(const float* x, const float* dy, float* dw, float* ws, int B, int T, int F, int Cin, int Cout,
                 int splits, int kps, hipStream_t st) {
  const int P = B * T * F;
  const int tm = pe_cdiv(Cout, BM), tn = pe_cdiv(Cin, BN);
  const int n_tiles = tm * tn, n_chunks = n_tiles * splits;
  const int grid = 8 * 9 * pe_cdiv(n_chunks, 8);
  hipLaunchKernelGGL((conv3x3_wgrad_kernel<BM, BN>), dim3(grid), dim3(256), 0, st, dy, x, ws, T, F, Cin, Cout, P, kps,
                     tn, n_tiles, n_chunks);
  PE_LAUNCH_CHECK();
  const int n = 9 * Cout * Cin;
  hipLaunchKernelGGL(conv3x3_wgrad_reduce_kernel, dim3(pe_cdiv(n / 4, 256)), dim3(256), 0, st, ws, dw, splits, Cout,
                     Cin);
  PE_LAUNCH_CHECK();
  return PE_OK;
}

// ---------------------------------------------------------------- first layer, Cin = 1 -> Cout = 64
// x element (b, t, f) at x[b*sb + t*st + f*sf]; y channels-last [B][T][F][64].  16 threads per pixel,
// 4 output channels each (one float4 store; 16 threads write one 256-B pixel row).
// STATS: also reduce the BatchNorm batch statistics of the output (double sum / sum of squares per channel over this
// workgroup's pixels -> one [2][64] row of bn_partials per workgroup), saving the 1 GB statistics pass over y
template <bool STATS, class TA = float>
__global__ __launch_bounds__(256) void conv3x3_c1_fwd_kernel(const float* __restrict__ x, long sb, long st_, long sf,
                                                             const float* __restrict__ w, TA* __restrict__ y,
                                                             double* __restrict__ bn_partials, int B, int T, int F) {
  __shared__ double sred[STATS ? 2 * 16 * 64 : 1];
  double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
  const int q = threadIdx.x & 15;                 // channel quad
  float wr[4][9];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int k = 0; k < 9; ++k) wr[c][k] = w[(q * 4 + c) * 9 + k];
  // A 16-thread group takes RUNS of four consecutive f positions of one (b, t) row (F % 4 == 0, host): 18 input
  // loads for four output pixels instead of 36 -- the kernel is bound by vector-memory instruction issue (ten per KB
  // written before), not by HBM.
  const int F4 = F >> 2, R = B * T * F4;          // < 2^31 (checked by the host): 32-bit index arithmetic
  for (int run = blockIdx.x * 16 + (threadIdx.x >> 4); run < R; run += gridDim.x * 16) {
    const int bt = run / F4, f0 = (run - bt * F4) * 4;
    const long b = bt / T;
    const int t = bt - (int)b * T;
    float in[3][6];
#pragma unroll
    for (int dt = 0; dt < 3; ++dt)
#pragma unroll
      for (int df = 0; df < 6; ++df) {
        const int tt = t + dt - 1, ff = f0 + df - 1;
        in[dt][df] = (tt >= 0 && tt < T && ff >= 0 && ff < F) ? x[b * sb + tt * st_ + ff * sf] : 0.f;
      }
#pragma unroll
    for (int px = 0; px < 4; ++px) {
      float4 o;
      float* op = reinterpret_cast<float*>(&o);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) s = fmaf(in[k / 3][px + k % 3], wr[c][k], s);
        op[c] = s;
        if constexpr (STATS) {
          const double dv = (double)stored_value<TA>(s);
          s1[c] += dv;
          s2[c] += dv * dv;
        }
      }
      st4(y + ((long)bt * F + f0 + px) * 64 + q * 4, o);
    }
  }
  if constexpr (STATS) {                          // fold the 16 pixel groups of the workgroup (fixed order)
    const int grp = threadIdx.x >> 4;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      sred[grp * 64 + q * 4 + c] = s1[c];
      sred[16 * 64 + grp * 64 + q * 4 + c] = s2[c];
    }
    __syncthreads();
    if (threadIdx.x < 128) {
      const int which = threadIdx.x >> 6, ch = threadIdx.x & 63;
      double t = 0.0;
#pragma unroll
      for (int g = 0; g < 16; ++g) t += sred[which * 16 * 64 + g * 64 + ch];
      bn_partials[((long)blockIdx.x * 2 + which) * 64 + ch] = t;
    }
  }
}

// dW[co][tap] = sum_p dY[p][co] * x[p + tap]; partials per workgroup, then an ordered reduce.
template <class TA = float>
__global__ __launch_bounds__(256) void conv3x3_c1_wgrad_kernel(const float* __restrict__ x, long sb, long st_,
                                                               long sf, const TA* __restrict__ dy,
                                                               float* __restrict__ partial, int B, int T, int F) {
  // 16 threads per pixel, 4 output channels each (one float4 of dY per thread); 16 pixels per block pass;
  // the nine taps of a pixel are read once per 16-thread group (same address: one broadcast load).
  __shared__ float red[16][64][9];
  const int q = threadIdx.x & 15, grp = threadIdx.x >> 4;
  float acc[4][9];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[c][k] = 0.f;
  // runs of four consecutive f positions per group, as in the forward kernel: 18 + 4 loads for four pixels
  const int F4 = F >> 2, R = B * T * F4;                       // F % 4 == 0, B*T*F < 2^31 (checked by the host)
  for (int run = blockIdx.x * 16 + grp; run < R; run += gridDim.x * 16) {
    const int bt = run / F4, f0 = (run - bt * F4) * 4, b = bt / T, t = bt - b * T;
    const float* xb = x + b * sb;
    float in[3][6];
#pragma unroll
    for (int dt = 0; dt < 3; ++dt)
#pragma unroll
      for (int df = 0; df < 6; ++df) {
        const int tt = t + dt - 1, ff = f0 + df - 1;
        in[dt][df] = (tt >= 0 && tt < T && ff >= 0 && ff < F) ? xb[tt * st_ + ff * sf] : 0.f;
      }
#pragma unroll
    for (int px = 0; px < 4; ++px) {
      const float4 g = ld4(dy + ((long)bt * F + f0 + px) * 64 + q * 4);
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        const float xv = in[k / 3][px + k % 3];
        acc[0][k] = fmaf(g.x, xv, acc[0][k]);
        acc[1][k] = fmaf(g.y, xv, acc[1][k]);
        acc[2][k] = fmaf(g.z, xv, acc[2][k]);
        acc[3][k] = fmaf(g.w, xv, acc[3][k]);
      }
    }
  }
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int k = 0; k < 9; ++k) red[grp][q * 4 + c][k] = acc[c][k];
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 9; i += 256) {
    const int c = i / 9, k = i % 9;
    float s = 0.f;
#pragma unroll
    for (int gq = 0; gq < 16; ++gq) s += red[gq][c][k];
    partial[(long)blockIdx.x * 576 + i] = s;
  }
}

__global__ void c1_wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw, int nblocks) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);            // one wave per output element
  if (i >= 576) return;
  const int lane = threadIdx.x & 63;
  double s = 0.0;
  for (int z = lane; z < nblocks; z += 64) s += (double)partial[(long)z * 576 + i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if (lane == 0) dw[i] = (float)s;
}

constexpr int kC1WgradBlocks = 2048;

}  // namespace

#ifndef PE_F16_BUILD
extern "C" int pe_conv3x3_repack(const float* w_oihw, float* w_fwd, float* w_dgrad, int Cout, int Cin,
                                 void* stream) {
  if (!w_oihw || Cout <= 0 || Cin <= 0) return PE_E_ARG;
  const int n = Cout * Cin * 9;
  hipLaunchKernelGGL(repack3x3_kernel, dim3(pe_cdiv(n, 256)), dim3(256), 0, pe_stream(stream), w_oihw, w_fwd,
                     w_dgrad, Cout, Cin);
  PE_LAUNCH_CHECK();
  return PE_OK;
}

extern "C" int pe_transpose2d(const float* in, float* out, int rows, int cols, void* stream) {
  if (!in || !out || rows <= 0 || cols <= 0) return PE_E_ARG;
  hipLaunchKernelGGL(transpose2d_kernel, dim3(pe_cdiv(cols, 32), pe_cdiv(rows, 32)), dim3(32, 8), 0,
                     pe_stream(stream), in, out, rows, cols);
  PE_LAUNCH_CHECK();
  return PE_OK;
}
#endif

template <int MODE, class TA = float>
static int conv3x3_fwd_impl(const TA* x, const float* w_packed, TA* y, int B, int T, int F, int C, int N,
                            int accumulate, void* stream, const unsigned* amax_x = nullptr,
                            const unsigned* amax_w = nullptr) {
  if (!x || !w_packed || !y || B <= 0 || T <= 0 || F <= 0 || C <= 0 || N <= 0) return PE_E_ARG;
  if ((C % 32) != 0 || (long)B * T * F * (C > N ? C : N) >= (1L << 31)) return PE_E_UNSUPPORTED;
  if (MODE == kSplit2 && (!amax_x || !amax_w)) return PE_E_ARG;
  hipStream_t st = pe_stream(stream);
  if (N <= 64)
    return launch_conv<Tile<256, 64, 4, 1>, MODE, TA>(x, w_packed, y, B, T, F, C, N, accumulate, st, amax_x, amax_w);
  if (N % 192 == 0 && N % 128 != 0)
    return launch_conv<Tile<128, 192, 2, 2>, MODE, TA>(x, w_packed, y, B, T, F, C, N, accumulate, st, amax_x, amax_w);
  return launch_conv<Tile<128, 128, 2, 2>, MODE, TA>(x, w_packed, y, B, T, F, C, N, accumulate, st, amax_x, amax_w);
}

#ifndef PE_F16_BUILD
extern "C" int pe_conv3x3_fwd(const float* x, const float* w_packed, float* y, int B, int T, int F, int C, int N,
                              int accumulate, void* stream) {
  return conv3x3_fwd_impl<kNative>(x, w_packed, y, B, T, F, C, N, accumulate, stream);
}
#endif

extern "C" int PE_HALF(pe_conv3x3_fwd)(const float* x, const float* w_packed, float* y, int B, int T, int F, int C,
                                   int N, int accumulate, void* stream) {
  return conv3x3_fwd_impl<kBf16>(x, w_packed, y, B, T, F, C, N, accumulate, stream);
}

#ifndef PE_F16_BUILD
extern "C" int pe_conv3x3_fwd_x3(const float* x, const float* w_packed, float* y, int B, int T, int F, int C,
                                 int N, int accumulate, void* stream) {
  return conv3x3_fwd_impl<kSplit>(x, w_packed, y, B, T, F, C, N, accumulate, stream);
}

extern "C" int pe_conv3x3_fwd_h2(const float* x, const float* w_packed, float* y, int B, int T, int F, int C,
                                 int N, int accumulate, const unsigned* amax_x, const unsigned* amax_w, void* stream) {
  return conv3x3_fwd_impl<kSplit2>(x, w_packed, y, B, T, F, C, N, accumulate, stream, amax_x, amax_w);
}
#endif

// ---- weights pre-packed as MFMA fragments (x3: three bf16 terms; bf16: one rounded term)
#ifndef PE_F16_BUILD
extern "C" size_t pe_wfrag_bytes(int N, int K, int terms) {
  if (N <= 0 || K <= 0 || (K & 15) || terms < 1 || terms > 3) return 0;
  return (size_t)((N + 31) / 32) * (K / 16) * terms * 1024;
}

extern "C" int pe_wfrag_pack(const float* w, long ld, int N, int K, int terms, void* out, void* stream) {
  if (!w || !out || N <= 0 || K <= 0 || ld < K) return PE_E_ARG;
  if ((K & 15) || (ld & 3) || (terms != 1 && terms != 3)) return PE_E_UNSUPPORTED;
  const long threads = (long)((N + 31) / 32) * (K / 16) * 64;
  if (terms == 3)
    hipLaunchKernelGGL(wfrag_pack_kernel<3>, dim3(pe_cdiv(threads, 256)), dim3(256), 0, pe_stream(stream), w, ld, N, K,
                       reinterpret_cast<uint4*>(out));
  else
    hipLaunchKernelGGL(wfrag_pack_kernel<1>, dim3(pe_cdiv(threads, 256)), dim3(256), 0, pe_stream(stream), w, ld, N, K,
                       reinterpret_cast<uint4*>(out));
  PE_LAUNCH_CHECK();
  return PE_OK;
}

// two scaled fp16 terms per weight ("h2"): the scale comes from *amax (pe_absmax of w)
extern "C" int pe_wfrag_pack_h2(const float* w, long ld, int N, int K, const unsigned* amax, void* out, void* stream) {
  if (!w || !out || !amax || N <= 0 || K <= 0 || ld < K) return PE_E_ARG;
  if ((K & 15) || (ld & 3)) return PE_E_UNSUPPORTED;
  const long threads = (long)((N + 31) / 32) * (K / 16) * 64;
  hipLaunchKernelGGL(wfrag_pack_kernel<2>, dim3(pe_cdiv(threads, 256)), dim3(256), 0, pe_stream(stream), w, ld, N, K,
                     reinterpret_cast<uint4*>(out), amax);
  PE_LAUNCH_CHECK();
  return PE_OK;
}

extern "C" int pe_conv3x3_wf_supported(int F, int C, int N) {
  return (C % 32) == 0 && conv_halo_passes(F, N) != 0 ? 1 : 0;
}

// number of per-tile BatchNorm partials pe_conv3x3_fwd_wf_* writes: bn_partials is [parts][2][N] doubles
extern "C" int pe_conv3x3_wf_stat_parts(int B, int T, int F) { return pe_cdiv((long)B * T * F, 128); }
#endif

#ifdef PE_F16_BUILD
// fp16 build: one RNE-rounded fp16 term per weight, same fragment order as pe_wfrag_pack(..., terms = 1, ...)
extern "C" int pe_wfrag_pack_f16(const float* w, long ld, int N, int K, void* out, void* stream) {
  if (!w || !out || N <= 0 || K <= 0 || ld < K) return PE_E_ARG;
  if ((K & 15) || (ld & 3)) return PE_E_UNSUPPORTED;
  const long threads = (long)((N + 31) / 32) * (K / 16) * 64;
  hipLaunchKernelGGL(wfrag_pack_kernel<1>, dim3(pe_cdiv(threads, 256)), dim3(256), 0, pe_stream(stream), w, ld, N, K,
                     reinterpret_cast<uint4*>(out));
  PE_LAUNCH_CHECK();
  return PE_OK;
}
#endif
template <int MODE, class TA = float>
static int conv3x3_fwd_wf_impl(const TA* x, const void* wfrag, TA* y, int B, int T, int F, int C, int N,
                               int accumulate, double* stats, void* stream, const unsigned* amax_x = nullptr,
                               const unsigned* amax_w = nullptr) {
  if (!x || !wfrag || !y || B <= 0 || T <= 0 || F <= 0 || C <= 0 || N <= 0) return PE_E_ARG;
  if ((C % 32) != 0 || (long)B * T * F * (C > N ? C : N) >= (1L << 31)) return PE_E_UNSUPPORTED;
  // the kernel addresses x and the fragment buffer through 32-bit buffer offsets, out-of-range = past 2 GiB
  if ((long)B * T * F * C * (long)sizeof(TA) >= (1L << 31) || (long)pe_wfrag_bytes_host(N, 9 * C, mode_terms<MODE>()) >= (1L << 31))
    return PE_E_UNSUPPORTED;
  if (MODE == kSplit2 && (!amax_x || !amax_w)) return PE_E_ARG;
  hipStream_t st = pe_stream(stream);
  const int passes = conv_halo_passes(F, N);
  if (passes == 10)
    return launch_conv_halo_wf<64, MODE, 10, 6, true, TA>(x, wfrag, y, B, T, F, C, N, accumulate, stats, st, amax_x,
                                                          amax_w);
  if (passes == 7) {
    if (N % 192 == 0 && N % 128 != 0)
      return launch_conv_halo_wf<192, MODE, 7, 3, false, TA>(x, wfrag, y, B, T, F, C, N, accumulate, stats, st, amax_x,
                                                             amax_w);
    return launch_conv_halo_wf<128, MODE, 7, 3, true, TA>(x, wfrag, y, B, T, F, C, N, accumulate, stats, st, amax_x,
                                                          amax_w);
  }
  return PE_E_UNSUPPORTED;
}

#ifndef PE_F16_BUILD
extern "C" int pe_conv3x3_fwd_wf_x3(const float* x, const void* wfrag, float* y, int B, int T, int F, int C, int N,
                                    int accumulate, double* bn_partials, void* stream) {
  return conv3x3_fwd_wf_impl<kSplit>(x, wfrag, y, B, T, F, C, N, accumulate, bn_partials, stream);
}

extern "C" int pe_conv3x3_fwd_wf_h2(const float* x, const void* wfrag, float* y, int B, int T, int F, int C, int N,
                                    int accumulate, double* bn_partials, const unsigned* amax_x,
                                    const unsigned* amax_w, void* stream) {
  return conv3x3_fwd_wf_impl<kSplit2>(x, wfrag, y, B, T, F, C, N, accumulate, bn_partials, stream, amax_x, amax_w);
}
#endif

extern "C" int PE_HALF(pe_conv3x3_fwd_wf)(const float* x, const void* wfrag, float* y, int B, int T, int F, int C, int N,
                                      int accumulate, double* bn_partials, void* stream) {
  return conv3x3_fwd_wf_impl<kBf16>(x, wfrag, y, B, T, F, C, N, accumulate, bn_partials, stream);
}

#ifndef PE_F16_BUILD
extern "C" size_t pe_conv3x3_wgrad_workspace_bytes(int B, int T, int F, int Cin, int Cout) {
  if (Cin == 1) return (size_t)kC1WgradBlocks * 576 * sizeof(float);
  int bm, bn, splits, kps;
  wgrad_plan(B * T * F, Cout, Cin, &bm, &bn, &splits, &kps);
  return (size_t)splits * 9 * Cout * Cin * sizeof(float);
}
#endif

template <int MODE, class TA = float>
static int conv3x3_wgrad_impl(const TA* x, const TA* dy, float* dw_oihw, int B, int T, int F, int Cin,
                              int Cout, float* workspace, size_t workspace_bytes, void* stream,
                              const unsigned* amax_x = nullptr, const unsigned* amax_dy = nullptr) {
  if (!x || !dy || !dw_oihw || B <= 0 || T <= 0 || F <= 0 || Cin <= 0 || Cout <= 0) return PE_E_ARG;
  if (MODE == kSplit2 && (!amax_x || !amax_dy)) return PE_E_ARG;
  if ((Cin & 3) || (Cout & 3)) return PE_E_UNSUPPORTED;
  if (!workspace || workspace_bytes < pe_conv3x3_wgrad_workspace_bytes(B, T, F, Cin, Cout)) return PE_E_WORKSPACE;
  int bm, bn, splits, kps;
  wgrad_plan(B * T * F, Cout, Cin, &bm, &bn, &splits, &kps);
  hipStream_t st = pe_stream(stream);
  if (wgrad9_ok(Cout, Cin)) {
    const int P = B * T * F, tn = Cin / 64;
    if (MODE == kSplit)
      hipLaunchKernelGGL((conv3x3_wgrad9_x3_kernel<3, TA>), dim3((Cout / 64) * tn * splits), dim3(256), 0, st, dy, x,
                         workspace, T, F, Cin, Cout, P, kps, tn, nullptr, nullptr);
    else if (MODE == kSplit2)
      hipLaunchKernelGGL((conv3x3_wgrad9_x3_kernel<2, TA>), dim3((Cout / 64) * tn * splits), dim3(256), 0, st, dy, x,
                         workspace, T, F, Cin, Cout, P, kps, tn, amax_dy, amax_x);
    else if (MODE == kBf16)
      hipLaunchKernelGGL((conv3x3_wgrad9_x3_kernel<1, TA>), dim3((Cout / 64) * tn * splits), dim3(256), 0, st, dy, x,
                         workspace, T, F, Cin, Cout, P, kps, tn, nullptr, nullptr);
    else if constexpr (std::is_same<TA, float>::value)
      hipLaunchKernelGGL(conv3x3_wgrad9_kernel<0>, dim3((Cout / 64) * tn * splits), dim3(256), 0, st, dy, x,
                         workspace, T, F, Cin, Cout, P, kps, tn);
    PE_LAUNCH_CHECK();
    const int n = 9 * Cout * Cin;
    hipLaunchKernelGGL(conv3x3_wgrad_reduce_kernel, dim3(pe_cdiv(n / 4, 256)), dim3(256), 0, st, workspace, dw_oihw,
                       splits, Cout, Cin);
    PE_LAUNCH_CHECK();
    return PE_OK;
  }
  // channel counts that are not multiples of 64: per-tap kernels on the native fp32 MFMA in every mode (fp32 tensors)
  if constexpr (std::is_same<TA, float>::value) {
    if (bm == 64 && bn == 64) return launch_wgrad<64, 64>(x, dy, dw_oihw, workspace, B, T, F, Cin, Cout, splits, kps, st);
    if (bm == 64) return launch_wgrad<64, 128>(x, dy, dw_oihw, workspace, B, T, F, Cin, Cout, splits, kps, st);
    if (bn == 64) return launch_wgrad<128, 64>(x, dy, dw_oihw, workspace, B, T, F, Cin, Cout, splits, kps, st);
    return launch_wgrad<128, 128>(x, dy, dw_oihw, workspace, B, T, F, Cin, Cout, splits, kps, st);
  }
  return PE_E_UNSUPPORTED;
}

#ifndef PE_F16_BUILD
extern "C" int pe_conv3x3_wgrad(const float* x, const float* dy, float* dw_oihw, int B, int T, int F, int Cin,
                                int Cout, float* workspace, size_t workspace_bytes, void* stream) {
  return conv3x3_wgrad_impl<kNative>(x, dy, dw_oihw, B, T, F, Cin, Cout, workspace, workspace_bytes, stream);
}
#endif

extern "C" int PE_HALF(pe_conv3x3_wgrad)(const float* x, const float* dy, float* dw_oihw, int B, int T, int F, int Cin,
                                     int Cout, float* workspace, size_t workspace_bytes, void* stream) {
  return conv3x3_wgrad_impl<kBf16>(x, dy, dw_oihw, B, T, F, Cin, Cout, workspace, workspace_bytes, stream);
}

#ifndef PE_F16_BUILD
// ---- mixed precision with bf16 ACTIVATION STORAGE (x, y, dy are bf16 tensors in HBM; weights / gradients fp32)
extern "C" int pe_conv3x3_fwd_bf16_a16(const void* x, const float* w_packed, void* y, int B, int T, int F, int C, int N,
                                       int accumulate, void* stream) {
  return conv3x3_fwd_impl<kBf16, act16_t>(static_cast<const act16_t*>(x), w_packed, static_cast<act16_t*>(y), B, T, F,
                                          C, N, accumulate, stream);
}

extern "C" int pe_conv3x3_fwd_wf_bf16_a16(const void* x, const void* wfrag, void* y, int B, int T, int F, int C, int N,
                                          int accumulate, double* bn_partials, void* stream) {
  return conv3x3_fwd_wf_impl<kBf16, act16_t>(static_cast<const act16_t*>(x), wfrag, static_cast<act16_t*>(y), B, T, F,
                                             C, N, accumulate, bn_partials, stream);
}

extern "C" int pe_conv3x3_wgrad_bf16_a16(const void* x, const void* dy, float* dw_oihw, int B, int T, int F, int Cin,
                                         int Cout, float* workspace, size_t workspace_bytes, void* stream) {
  return conv3x3_wgrad_impl<kBf16, act16_t>(static_cast<const act16_t*>(x), static_cast<const act16_t*>(dy), dw_oihw,
                                            B, T, F, Cin, Cout, workspace, workspace_bytes, stream);
}
#endif

#ifndef PE_F16_BUILD
extern "C" int pe_conv3x3_wgrad_x3(const float* x, const float* dy, float* dw_oihw, int B, int T, int F, int Cin,
                                   int Cout, float* workspace, size_t workspace_bytes, void* stream) {
  return conv3x3_wgrad_impl<kSplit>(x, dy, dw_oihw, B, T, F, Cin, Cout, workspace, workspace_bytes, stream);
}

extern "C" int pe_conv3x3_wgrad_h2(const float* x, const float* dy, float* dw_oihw, int B, int T, int F, int Cin,
                                   int Cout, float* workspace, size_t workspace_bytes, const unsigned* amax_x,
                                   const unsigned* amax_dy, void* stream) {
  return conv3x3_wgrad_impl<kSplit2>(x, dy, dw_oihw, B, T, F, Cin, Cout, workspace, workspace_bytes, stream, amax_x,
                                     amax_dy);
}

static int c1_grid(int B, int T, int F) {
  const long R = (long)B * T * (F / 4);           // runs of four pixels, 16 per workgroup pass
  return (int)((R + 15) / 16 < 8192 ? (R + 15) / 16 : 8192);
}

// rows of the [rows][2][64] double partials pe_conv3x3_c1_fwd writes when bn_partials is given
extern "C" int pe_conv3x3_c1_stat_parts(int B, int T, int F) { return (B > 0 && T > 0 && F > 0) ? c1_grid(B, T, F) : 0; }

template <class TA>
static int conv3x3_c1_fwd_impl(const float* x, long sb, long st, long sf, const float* w_oihw, TA* y, int B, int T,
                               int F, double* bn_partials, void* stream) {
  if (!x || !w_oihw || !y || B <= 0 || T <= 0 || F <= 0) return PE_E_ARG;
  if ((F & 3) || (long)B * T * F >= (1L << 31) - 64L * 8192) return PE_E_UNSUPPORTED;   // runs of 4, 32-bit indices
  const int grid = c1_grid(B, T, F);
  if (bn_partials)
    hipLaunchKernelGGL((conv3x3_c1_fwd_kernel<true, TA>), dim3(grid), dim3(256), 0, pe_stream(stream), x, sb, st, sf,
                       w_oihw, y, bn_partials, B, T, F);
  else
    hipLaunchKernelGGL((conv3x3_c1_fwd_kernel<false, TA>), dim3(grid), dim3(256), 0, pe_stream(stream), x, sb, st, sf,
                       w_oihw, y, nullptr, B, T, F);
  PE_LAUNCH_CHECK();
  return PE_OK;
}

extern "C" int pe_conv3x3_c1_fwd(const float* x, long sb, long st, long sf, const float* w_oihw, float* y, int B,
                                 int T, int F, double* bn_partials, void* stream) {
  return conv3x3_c1_fwd_impl<float>(x, sb, st, sf, w_oihw, y, B, T, F, bn_partials, stream);
}

extern "C" int pe_conv3x3_c1_fwd_a16(const float* x, long sb, long st, long sf, const float* w_oihw, void* y, int B,
                                     int T, int F, double* bn_partials, void* stream) {
  return conv3x3_c1_fwd_impl<act16_t>(x, sb, st, sf, w_oihw, static_cast<act16_t*>(y), B, T, F, bn_partials, stream);
}

template <class TA>
static int conv3x3_c1_wgrad_impl(const float* x, long sb, long st, long sf, const TA* dy, float* dw_oihw, int B, int T,
                                 int F, float* workspace, size_t workspace_bytes, void* stream) {
  if (!x || !dy || !dw_oihw || B <= 0 || T <= 0 || F <= 0) return PE_E_ARG;
  if ((F & 3) || (long)B * T * F >= (1L << 31) - 65536) return PE_E_UNSUPPORTED;   // runs of 4, 32-bit pixel index
  if (!workspace || workspace_bytes < (size_t)kC1WgradBlocks * 576 * sizeof(float)) return PE_E_WORKSPACE;
  hipLaunchKernelGGL(conv3x3_c1_wgrad_kernel<TA>, dim3(kC1WgradBlocks), dim3(256), 0, pe_stream(stream), x, sb, st, sf,
                     dy, workspace, B, T, F);
  PE_LAUNCH_CHECK();
  hipLaunchKernelGGL(c1_wgrad_reduce_kernel, dim3(144), dim3(256), 0, pe_stream(stream), workspace, dw_oihw,
                     kC1WgradBlocks);
  PE_LAUNCH_CHECK();
  return PE_OK;
}

extern "C" int pe_conv3x3_c1_wgrad(const float* x, long sb, long st, long sf, const float* dy, float* dw_oihw,
                                   int B, int T, int F, float* workspace, size_t workspace_bytes, void* stream) {
  return conv3x3_c1_wgrad_impl<float>(x, sb, st, sf, dy, dw_oihw, B, T, F, workspace, workspace_bytes, stream);
}

extern "C" int pe_conv3x3_c1_wgrad_a16(const float* x, long sb, long st, long sf, const void* dy, float* dw_oihw,
                                       int B, int T, int F, float* workspace, size_t workspace_bytes, void* stream) {
  return conv3x3_c1_wgrad_impl<act16_t>(x, sb, st, sf, static_cast<const act16_t*>(dy), dw_oihw, B, T, F, workspace,
                                        workspace_bytes, stream);
}
#endif
