// fp32 MFMA tile engines for gfx950 (v_mfma_f32_32x32x2_f32: exact-f32 fmaf chains).
//
// Two main loops, both 256 threads = 4 waves, BK = 32, register-prefetched single LDS stage:
//
//   NT:  C[m][n] = sum_k A[m][k] * B[n][k]   both operands k-contiguous in memory.
//        LDS images are [row][36] floats (32 + one 16-byte pad: conflict-free ds_read_b128).
//        Lane (r = lane & 31, h = lane >> 5) reads 4 consecutive k with one ds_read_b128 and
//        feeds 4 MFMAs; MFMA j of an 8-k block covers k = {j, 4 + j} (the k order inside a
//        block is a free permutation as long as A and B agree).
//
//   TN:  C[m][n] = sum_k A[k][m] * B[k][n]   both operands k-major (weight-gradient GEMMs,
//        k = pixels or batch*time).  LDS images are [k][cols]; operands come from
//        conflict-free ds_read_b32 (32 consecutive lanes -> 32 consecutive columns).
//
// MFMA 32x32x2 maps (cdna_hip_programming.md section 3): A operand lane l holds A[i = l & 31][k = l >> 5],
// B operand lane l holds B[k = l >> 5][j = l & 31]; accumulator register g of lane l is
// C[row = (g & 3) + 8 * (g >> 2) + 4 * (l >> 5)][col = l & 31].
#pragma once
#include <utility>
#include "act16.h"

namespace pe {

constexpr int kBK = 32;
// precision modes of the tile engines
constexpr int kNative = 0;    // v_mfma_f32_32x32x2_f32
constexpr int kBf16 = 1;      // operands rounded to bf16 (mixed precision)
constexpr int kSplit = 2;     // fp32 as three bf16 terms, six bf16 MFMAs per product block (fp32-accurate)
constexpr int kSplit2 = 3;    // fp32 as two scaled fp16 terms, three fp16 MFMAs per product block ("h2", below)
template <int MODE> constexpr int mode_terms() { return MODE == kSplit ? 3 : MODE == kSplit2 ? 2 : 1; }
constexpr int kLdsStride = kBK + 4;

template <int BM_, int BN_, int WAVES_M_, int WAVES_N_>
struct Tile {
  static constexpr int BM = BM_, BN = BN_;
  static constexpr int WAVES_M = WAVES_M_, WAVES_N = WAVES_N_;
  static constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  static constexpr int TM = WM / 32, TN = WN / 32;
  static constexpr int A_LOADS = BM / 32, B_LOADS = BN / 32;   // float4 per thread per k-tile
  static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");
  static_assert(WM % 32 == 0 && WN % 32 == 0, "wave tile is a multiple of the 32x32 MFMA");
};

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// XCD-aware tile order: workgroups b and b+8 share an XCD (and its L2); give every XCD a
// contiguous run of tiles so neighbouring tiles (shared halo rows / operand panels) hit the
// same L2.  Bijective for any tile count.
__device__ __forceinline__ int xcd_remap(int bid, int ntiles) {
  const int q = ntiles >> 3, r = ntiles & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// ------------------------------------------------------------------ operand loaders (NT)
// A loader hands out float4 = 4 consecutive k of one row.  `slot` i in [0, LOADS) addresses
// row (tid >> 3) + 32 * i of the tile; k4 = (tid & 7) * 4 within the 32-wide k-tile.

// Loads go through a buffer descriptor based at the tile's first row: ONE instruction per request, rows past the
// matrix (and the k tail) read as zero by the descriptor's range check.  Predicated global loads cost a branch, an exec
// save/restore, a 64-bit address add and four zero-fill moves each in the k-loop, and hipcc, unable to count loads it
// branches around, waits for ALL of them (vmcnt(0)) wherever one is needed.
template <class T>
struct RowLoaderT {           // plain row-major matrix of T (float, or bf16 activations), rows x K, leading dimension ld
  const T* p;
  long ld;
  int rows, K;
  int row0;
  const T* base_ = nullptr;   // device state, set by init()
  unsigned bytes_ = 0, vo_ = 0, rowstep_ = 0;
  typedef typename RawQuad<T>::type Raw;
  __device__ __forceinline__ void init(int first_row) {
    row0 = first_row + (threadIdx.x >> 3);
    const long left = (long)rows - first_row;                       // a tile spans at most 256 rows: offsets fit 32 bits
    long bytes = left > 0 ? ((left - 1) * ld + K) * (long)sizeof(T) : 0;
    bytes_ = (unsigned)(bytes < 0x7fffffffL ? bytes : 0x7fffffffL);
    base_ = p + (long)first_row * ld;
    vo_ = (unsigned)(((long)(threadIdx.x >> 3) * ld + (threadIdx.x & 7) * 4) * (long)sizeof(T));
    rowstep_ = (unsigned)(32 * ld * (long)sizeof(T));
  }
  __device__ __forceinline__ Raw load(int slot, int kt) const {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(base_), 0, bytes_, 0x00020000);
    const unsigned ktail = (kt * kBK + (int)(threadIdx.x & 7) * 4 < K) ? 0u : 0x80000000u;
    return ldraw_buffer<T>(rs, (vo_ + (unsigned)slot * rowstep_) | ktail, (unsigned)(kt * kBK * (int)sizeof(T)));
  }
};
typedef RowLoaderT<float> RowLoader;

template <int SLOTS, class TA = float>
struct ConvLoader {           // implicit im2col of a channels-last [B][T][F][C] tensor, 3x3, pad 1
  const TA* p;
  int T, F, C, rows;          // rows = B*T*F output pixels; K = 9*C ordered (kh, kw, c)
  int t_[SLOTS], f_[SLOTS];
  long base_[SLOTS];
  bool ok_[SLOTS];
  typedef typename RawQuad<TA>::type Raw;
  __device__ __forceinline__ void init(int first_row) {
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
      const int row = first_row + (threadIdx.x >> 3) + 32 * i;
      ok_[i] = row < rows;
      const int r = ok_[i] ? row : 0;
      f_[i] = r % F;
      t_[i] = (r / F) % T;
      base_[i] = (long)r * C;
    }
  }
  __device__ __forceinline__ Raw load(int slot, int kt) const {
    const int kbase = kt * kBK;             // wave-uniform
    const int tap = kbase / C;              // C % 32 == 0: a k-tile never straddles taps
    const int c = kbase - tap * C + (threadIdx.x & 7) * 4;
    const int dt = tap / 3 - 1, df = tap % 3 - 1;
    const int tt = t_[slot] + dt, ff = f_[slot] + df;
    if (ok_[slot] && tt >= 0 && tt < T && ff >= 0 && ff < F)
      return ldraw(p + base_[slot] + (long)(dt * F + df) * C + c);
    return zero_raw<Raw>();
  }
};

// ------------------------------------------------------------------ NT main loop
template <class TL, class AL, class BL>
__device__ __forceinline__ void nt_mainloop(AL& al, BL& bl, int K, float* As, float* Bs,
                                            f32x16 (&acc)[TL::TM][TL::TN]) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wm = wv / TL::WAVES_N, wn = wv % TL::WAVES_N;
  const int r = lane & 31, h = lane >> 5;
  const int nk = (K + kBK - 1) / kBK;
  float4 ra[TL::A_LOADS], rb[TL::B_LOADS];
#pragma unroll
  for (int i = 0; i < TL::A_LOADS; ++i) ra[i] = al.load(i, 0);
#pragma unroll
  for (int i = 0; i < TL::B_LOADS; ++i) rb[i] = bl.load(i, 0);

  const int st_off = (tid >> 3) * kLdsStride + (tid & 7) * 4;
  const float* a_rd = As + (wm * TL::WM + r) * kLdsStride + h * 4;
  const float* b_rd = Bs + (wn * TL::WN + r) * kLdsStride + h * 4;

  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < TL::A_LOADS; ++i)
      *reinterpret_cast<float4*>(As + st_off + i * 32 * kLdsStride) = ra[i];
#pragma unroll
    for (int i = 0; i < TL::B_LOADS; ++i)
      *reinterpret_cast<float4*>(Bs + st_off + i * 32 * kLdsStride) = rb[i];
    __syncthreads();
    if (kt + 1 < nk) {
#pragma unroll
      for (int i = 0; i < TL::A_LOADS; ++i) ra[i] = al.load(i, kt + 1);
#pragma unroll
      for (int i = 0; i < TL::B_LOADS; ++i) rb[i] = bl.load(i, kt + 1);
    }
#pragma unroll
    for (int k8 = 0; k8 < kBK / 8; ++k8) {
      float4 fa[TL::TM], fb[TL::TN];
#pragma unroll
      for (int i = 0; i < TL::TM; ++i)
        fa[i] = *reinterpret_cast<const float4*>(a_rd + i * 32 * kLdsStride + k8 * 8);
#pragma unroll
      for (int j = 0; j < TL::TN; ++j)
        fb[j] = *reinterpret_cast<const float4*>(b_rd + j * 32 * kLdsStride + k8 * 8);
#pragma unroll
      for (int i = 0; i < TL::TM; ++i)
#pragma unroll
        for (int j = 0; j < TL::TN; ++j) {
          acc[i][j] = mfma32(fa[i].x, fb[j].x, acc[i][j]);
          acc[i][j] = mfma32(fa[i].y, fb[j].y, acc[i][j]);
          acc[i][j] = mfma32(fa[i].z, fb[j].z, acc[i][j]);
          acc[i][j] = mfma32(fa[i].w, fb[j].w, acc[i][j]);
        }
    }
  }
}

// ------------------------------------------------------------------ NT main loop, bf16 operands
// Opt-in "mixed precision" variant (reference trainer.py:103 autocast): the fp32 operands are rounded
// to bf16 (RNE, v_cvt_pk_bf16_f32) on their way into LDS and multiplied with
// v_mfma_f32_32x32x16_bf16; accumulation, epilogue and every tensor in HBM stay fp32.
// LDS images are [row][40] bf16 (32 + one 16-byte pad): lane (r, h) reads k = 16*kk + 8*h .. +7 of
// row r with one ds_read_b128, which is exactly the 32x32x16 A/B operand layout.
// The single-term ("mixed precision") pipelines exist for two 16-bit operand types: bf16 (the default build) and
// fp16 = the reference's literal autocast dtype (trainer.py:64-102).  The fp16 form is the SAME source compiled a
// second time with -DPE_F16_BUILD (pitchextractor_amd/build.py): only the conversion instruction and the MFMA
// opcode differ, and that build exports only the pe_*_f16 entry points (PE_HALF names).  The three-term split
// paths are bf16 by construction and are not exported from the fp16 build.
#ifdef PE_F16_BUILD
typedef _Float16 pe_half_t;
#define PE_HALF(name) name##_f16
#else
typedef __bf16 pe_half_t;
#define PE_HALF(name) name##_bf16
#endif
typedef pe_half_t bf16x8 __attribute__((ext_vector_type(8)));
typedef pe_half_t bf16x4 __attribute__((ext_vector_type(4)));
constexpr int kLdsStrideH = kBK + 8;   // in bf16 elements

__device__ __forceinline__ bf16x4 to_bf16x4(const float4& v) {     // RNE to the build's 16-bit operand type
  bf16x4 o;
  o[0] = (pe_half_t)v.x; o[1] = (pe_half_t)v.y; o[2] = (pe_half_t)v.z; o[3] = (pe_half_t)v.w;
  return o;
}

// the LDS image's four 16-bit operands of a raw quad: fp32 data is rounded (RNE); bf16 data is already there
__device__ __forceinline__ bf16x4 to_half4(const float4& v) { return to_bf16x4(v); }
__device__ __forceinline__ bf16x4 to_half4(const uint2& raw) { return __builtin_bit_cast(bf16x4, raw); }

__device__ __forceinline__ f32x16 mfma_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
#ifdef PE_F16_BUILD
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
#else
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
#endif
}

template <class TL, class AL, class BL>
__device__ __forceinline__ void nt_mainloop_bf16(AL& al, BL& bl, int K, float* As_f, float* Bs_f,
                                                 f32x16 (&acc)[TL::TM][TL::TN]) {
  __bf16* As = reinterpret_cast<__bf16*>(As_f);
  __bf16* Bs = reinterpret_cast<__bf16*>(Bs_f);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wm = wv / TL::WAVES_N, wn = wv % TL::WAVES_N;
  const int r = lane & 31, h = lane >> 5;
  const int nk = (K + kBK - 1) / kBK;
  decltype(al.load(0, 0)) ra[TL::A_LOADS];
  decltype(bl.load(0, 0)) rb[TL::B_LOADS];
#pragma unroll
  for (int i = 0; i < TL::A_LOADS; ++i) ra[i] = al.load(i, 0);
#pragma unroll
  for (int i = 0; i < TL::B_LOADS; ++i) rb[i] = bl.load(i, 0);

  const int st_off = (tid >> 3) * kLdsStrideH + (tid & 7) * 4;
  const __bf16* a_rd = As + (wm * TL::WM + r) * kLdsStrideH + h * 8;
  const __bf16* b_rd = Bs + (wn * TL::WN + r) * kLdsStrideH + h * 8;

  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < TL::A_LOADS; ++i)
      *reinterpret_cast<bf16x4*>(As + st_off + i * 32 * kLdsStrideH) = to_half4(ra[i]);
#pragma unroll
    for (int i = 0; i < TL::B_LOADS; ++i)
      *reinterpret_cast<bf16x4*>(Bs + st_off + i * 32 * kLdsStrideH) = to_half4(rb[i]);
    __syncthreads();
    if (kt + 1 < nk) {
#pragma unroll
      for (int i = 0; i < TL::A_LOADS; ++i) ra[i] = al.load(i, kt + 1);
#pragma unroll
      for (int i = 0; i < TL::B_LOADS; ++i) rb[i] = bl.load(i, kt + 1);
    }
#pragma unroll
    for (int kk = 0; kk < kBK / 16; ++kk) {
      bf16x8 fa[TL::TM], fb[TL::TN];
#pragma unroll
      for (int i = 0; i < TL::TM; ++i)
        fa[i] = *reinterpret_cast<const bf16x8*>(a_rd + i * 32 * kLdsStrideH + kk * 16);
#pragma unroll
      for (int j = 0; j < TL::TN; ++j)
        fb[j] = *reinterpret_cast<const bf16x8*>(b_rd + j * 32 * kLdsStrideH + kk * 16);
#pragma unroll
      for (int i = 0; i < TL::TM; ++i)
#pragma unroll
        for (int j = 0; j < TL::TN; ++j)
          acc[i][j] = mfma_bf16(fa[i], fb[j], acc[i][j]);
    }
  }
}

// ------------------------------------------------------------------ NT main loop, fp32 as three bf16 terms
// fp32-accurate products on the bf16 MFMA pipe (16x the fp32 MFMA rate on gfx950).  Every fp32 operand
// is split EXACTLY into three bf16 terms by truncation, x = hi + mid + lo (8 + 8 + 8 significand bits;
// both subtractions are exact), on its way into LDS.  a*b = sum of 9 exact cross products; the three
// smallest (mid*lo, lo*mid, lo*lo <= 2^-23 |a*b|, below the rounding of an fp32 product) are dropped and
// the other six are accumulated in fp32 by v_mfma_f32_32x32x16_bf16, small terms first.
// LDS: three [row][40] bf16 images per operand.
constexpr int kSplitRowFloats = 3 * kBK / 2;             // floats of LDS per tile row (three 64-byte term rows)

struct Split3 { uint2 hi, mid, lo; };                    // 4 consecutive k of one row, packed bf16 pairs

__device__ __forceinline__ unsigned pack_hi16(float a, float b) {   // {bf16 bits of a, bf16 bits of b} by truncation
  return __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
}
__device__ __forceinline__ float trunc_bf16(float x) { return __uint_as_float(__float_as_uint(x) & 0xffff0000u); }

__device__ __forceinline__ Split3 split3(const float4& v) {
  Split3 o;
  const float x[4] = {v.x, v.y, v.z, v.w};
  float r1[4], r2[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) r1[i] = x[i] - trunc_bf16(x[i]);
#pragma unroll
  for (int i = 0; i < 4; ++i) r2[i] = r1[i] - trunc_bf16(r1[i]);
  o.hi = make_uint2(pack_hi16(x[0], x[1]), pack_hi16(x[2], x[3]));
  o.mid = make_uint2(pack_hi16(r1[0], r1[1]), pack_hi16(r1[2], r1[3]));
  o.lo = make_uint2(pack_hi16(r2[0], r2[1]), pack_hi16(r2[2], r2[3]));
  return o;
}

// acc += a * b to fp32 accuracy, a and b given as their three bf16 terms
__device__ __forceinline__ f32x16 mfma_split(const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x16 c) {
  c = mfma_bf16(a[2], b[0], c);
  c = mfma_bf16(a[0], b[2], c);
  c = mfma_bf16(a[1], b[1], c);
  c = mfma_bf16(a[1], b[0], c);
  c = mfma_bf16(a[0], b[1], c);
  c = mfma_bf16(a[0], b[0], c);
  return c;
}

// the same six products in the same order when the caller has swapped the roles of its two operands
__device__ __forceinline__ f32x16 mfma_split_swapped(const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x16 c) {
  c = mfma_bf16(a[0], b[2], c);
  c = mfma_bf16(a[2], b[0], c);
  c = mfma_bf16(a[1], b[1], c);
  c = mfma_bf16(a[0], b[1], c);
  c = mfma_bf16(a[1], b[0], c);
  c = mfma_bf16(a[0], b[0], c);
  return c;
}

// ------------------------------------------------------------------ fp32 as two scaled fp16 terms ("h2")
// fp16 carries 11 significand bits, so TWO terms hold 22: x * s = hi + lo (+ a residual <= 2^-22 |x s|), hi =
// RN_f16(x s), lo = RN_f16(x s - hi) (the subtraction is exact in fp32).  a * b is then hi_a hi_b + hi_a lo_b +
// lo_a hi_b (every product exact in the fp32 accumulator of v_mfma_f32_32x32x16_f16) with lo_a lo_b <= 2^-24 |a b|
// dropped: THREE MFMAs per product block instead of the six of the bf16 split, for a per-product error <= 2^-21
// |a b|, unbiased (round to nearest) and so averaging out over a sum; the accumulation rounding of a K-long fp32
// sum, common to every fp32 path, is ~2^-22 sqrt(K) |a b|.  What fp16 lacks is exponent range (5 bits), so every
// operand TENSOR carries a power-of-two scale s = 2^(140 - E), E = the biased exponent of its largest magnitude
// (pe_absmax, or the producing kernel's epilogue): max |x| s lies in [2^13, 2^14), hi stays a normal fp16 down to
// 2^-28 of the tensor's maximum and a subnormal with absolute resolution 2^-38 max below that.  The epilogue
// multiplies the accumulator by 1 / (s_a s_b), again a power of two: nothing but exponents change.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x16 mfma_f16(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// scale of a tensor whose largest magnitude has the IEEE bits `amax_bits` (sign bit clear)
__device__ __host__ __forceinline__ unsigned h2_scale_exp(unsigned amax_bits) {
  int es = 267 - (int)((amax_bits >> 23) & 0xffu);                  // biased exponent of 2^(140 - E)
  es = es < 1 ? 1 : es > 253 ? 253 : es;                            // all-zero / subnormal tensors: 2^126
  return (unsigned)es;
}
__device__ __forceinline__ float h2_scale(unsigned amax_bits) { return __uint_as_float(h2_scale_exp(amax_bits) << 23); }
__device__ __forceinline__ float h2_inv_scale(unsigned amax_bits) {
  return __uint_as_float((254u - h2_scale_exp(amax_bits)) << 23);
}

struct H2Scales {                                                   // s_a, s_b and 1 / (s_a s_b) of one product
  float sa, sb, inv;
  __device__ __forceinline__ void load(const unsigned* amax_a, const unsigned* amax_b) {
    const unsigned ba = __builtin_amdgcn_readfirstlane(*amax_a), bb = __builtin_amdgcn_readfirstlane(*amax_b);
    sa = h2_scale(ba); sb = h2_scale(bb);
    inv = h2_inv_scale(ba) * h2_inv_scale(bb);
  }
};

struct Split2 { uint2 hi, lo; };                                    // 4 consecutive k of one row, packed fp16 pairs

// lo pair of two elements: RN_f16(x * s - hi) with ONE mixed-precision FMA per element (fp32 x and s, fp16 hi read
// straight from its packed half, fp16 result written into its half of the destination).  x * s - hi is exact in fp32
// (s is a power of two and hi is within half an fp16 ulp of x * s), so this equals convert(fma) of the three-step form
// and costs 2 VALU per pair instead of 4 (unpack, packed FMA, pack): the staging passes of the h2 kernels are VALU-bound.
__device__ __forceinline__ unsigned split2_lo_pair(float x, float y, float s, unsigned hi) {
  unsigned lo;
  asm("v_fma_mixlo_f16 %0, %1, %3, -%4 op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixhi_f16 %0, %2, %3, -%4 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
      : "=&v"(lo) : "v"(x), "v"(y), "v"(s), "v"(hi));
  return lo;
}

__device__ __forceinline__ Split2 split2(const float4& v, float s) {
  typedef _Float16 h2v __attribute__((ext_vector_type(2)));
  typedef float f2v __attribute__((ext_vector_type(2)));
  const f2v t0 = {v.x * s, v.y * s}, t1 = {v.z * s, v.w * s};
  const h2v h0 = __builtin_convertvector(t0, h2v), h1 = __builtin_convertvector(t1, h2v);
  Split2 o;
  o.hi = make_uint2(__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1));
  o.lo = make_uint2(split2_lo_pair(v.x, v.y, s, o.hi.x), split2_lo_pair(v.z, v.w, s, o.hi.y));
  return o;
}

// acc += a * b from the two fp16 terms of each operand, small products first
__device__ __forceinline__ f32x16 mfma_split2(const bf16x8 (&a)[2], const bf16x8 (&b)[2], f32x16 c) {
  c = mfma_f16(a[1], b[0], c);
  c = mfma_f16(a[0], b[1], c);
  c = mfma_f16(a[0], b[0], c);
  return c;
}
// the same three products in the same order when the caller has swapped the roles of its two operands
__device__ __forceinline__ f32x16 mfma_split2_swapped(const bf16x8 (&a)[2], const bf16x8 (&b)[2], f32x16 c) {
  c = mfma_f16(a[0], b[1], c);
  c = mfma_f16(a[1], b[0], c);
  c = mfma_f16(a[0], b[0], c);
  return c;
}

// product block of an NT-term operand pair: NT = 3 bf16 split, 2 fp16 split, 1 rounded 16-bit operands
template <int NT, bool SW = false>
__device__ __forceinline__ f32x16 mfma_terms(const bf16x8 (&a)[NT], const bf16x8 (&b)[NT], f32x16 c) {
  if constexpr (NT == 3) return SW ? mfma_split_swapped(a, b, c) : mfma_split(a, b, c);
  else if constexpr (NT == 2) return SW ? mfma_split2_swapped(a, b, c) : mfma_split2(a, b, c);
  else return mfma_bf16(a[0], b[0], c);
}

// bf16-term images [row][32 k] with NO padding: 16-byte chunk c of a row sits at chunk c ^ ((row >> 2) & 3), which
// keeps ds_read_b128 fragment reads (32 consecutive rows at any row offset) and the ds_write_b64 staging stores
// conflict-free (MI355X_MICROARCH.md, LDS lane groups).
__device__ __forceinline__ int swz_off(int row, int chunk) {        // in bf16 elements
  return row * 32 + ((chunk ^ ((row >> 2) & 3)) << 3);
}

// 4 consecutive k (piece = float4 index 0..7 within the 32-k row) of one row -> NT term images
template <int NT>
__device__ __forceinline__ void halo_store(__bf16* img, int img_elems, int row, int piece, const float4& v,
                                           float scale = 1.0f) {
  const int off = swz_off(row, piece >> 1) + (piece & 1) * 4;
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
}

// the same for a quad that is still in its bf16 storage form: one rounded term needs no arithmetic at all
template <int NT>
__device__ __forceinline__ void halo_store(__bf16* img, int img_elems, int row, int piece, const uint2& raw,
                                           float scale = 1.0f) {
  if constexpr (NT == 1) *reinterpret_cast<uint2*>(img + swz_off(row, piece >> 1) + (piece & 1) * 4) = raw;
  else halo_store<NT>(img, img_elems, row, piece, widen(raw), scale);
}

template <class TL, bool SW = false, int NT = 3, int PF = 1, class AL, class BL>
__device__ __forceinline__ void nt_mainloop_split(AL& al, BL& bl, int K, float* As_f, float* Bs_f,
                                                  f32x16 (&acc)[TL::TM][TL::TN], float sa = 1.0f, float sb = 1.0f) {
  // unpadded, XOR-swizzled term images (swz_off): 64 B per row and term, so a 128 x 128 tile takes 48 KB (three
  // terms) and three workgroups share a CU (the padded 80-byte rows allowed two)
  constexpr int A_IMG = TL::BM * kBK, B_IMG = TL::BN * kBK;   // 16-bit elements per image
  __bf16* As = reinterpret_cast<__bf16*>(As_f);
  __bf16* Bs = reinterpret_cast<__bf16*>(Bs_f);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wm = wv / TL::WAVES_N, wn = wv % TL::WAVES_N;
  const int r = lane & 31, h = lane >> 5;
  const int nk = (K + kBK - 1) / kBK;
  const int srow = tid >> 3, piece = tid & 7;
  // PF k-tiles of operand quads in flight in registers.  PF = 2: the loads of tile kt + 2 are issued while tile kt's
  // products run and are first needed two stage phases later (one MFMA phase of a k-tile is ~0.7 us, an HBM round trip
  // under load 1.5-2 us: with one tile in flight every stage phase began by waiting for memory, WAIT_INST_ANY 44 % in
  // the PMC run).  Needs loaders whose requests are unconditional and read as zero past K (the buffer loaders), so the
  // compiler counts vmcnt exactly and the stage of tile kt waits for its own quads only; 40 more registers.
  decltype(al.load(0, 0)) ra[PF][TL::A_LOADS];
  decltype(bl.load(0, 0)) rb[PF][TL::B_LOADS];
#pragma unroll
  for (int q = 0; q < PF; ++q) {
#pragma unroll
    for (int i = 0; i < TL::A_LOADS; ++i) ra[q][i] = al.load(i, q);
#pragma unroll
    for (int i = 0; i < TL::B_LOADS; ++i) rb[q][i] = bl.load(i, q);
  }

  auto tile = [&](auto qc, int kt) {
    constexpr int q = decltype(qc)::value;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < TL::A_LOADS; ++i) halo_store<NT>(As, A_IMG, srow + 32 * i, piece, ra[q][i], sa);
#pragma unroll
    for (int i = 0; i < TL::B_LOADS; ++i) halo_store<NT>(Bs, B_IMG, srow + 32 * i, piece, rb[q][i], sb);
    __syncthreads();
    if (PF == 2 || kt + 1 < nk) {
#pragma unroll
      for (int i = 0; i < TL::A_LOADS; ++i) ra[q][i] = al.load(i, kt + PF);
#pragma unroll
      for (int i = 0; i < TL::B_LOADS; ++i) rb[q][i] = bl.load(i, kt + PF);
    }
    if constexpr (PF == 2) __builtin_amdgcn_sched_barrier(0);   // requests first, then the products: nothing of the next stage up here
#pragma unroll
    for (int kk = 0; kk < kBK / 16; ++kk) {
      bf16x8 fa[TL::TM][NT], fb[TL::TN][NT];
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
#pragma unroll
      for (int i = 0; i < TL::TM; ++i)
#pragma unroll
        for (int j = 0; j < TL::TN; ++j) acc[i][j] = mfma_terms<NT, SW>(fa[i], fb[j], acc[i][j]);
    }
    if constexpr (PF == 2) __builtin_amdgcn_sched_barrier(0);   // (the other set's split, hoisted, would wait for ITS loads here)
  };
  int kt = 0;
  if constexpr (PF == 2) {
    for (; kt + 1 < nk; kt += 2) {           // straight-line pairs: a conditional second tile costs the exact vmcnt
      tile(std::integral_constant<int, 0>{}, kt);
      tile(std::integral_constant<int, 1>{}, kt + 1);
    }
  }
  for (; kt < nk; ++kt) tile(std::integral_constant<int, 0>{}, kt);
}

// compile-time loop: body(std::integral_constant<int, I>) for I = 0 .. N-1 (a plain `#pragma unroll` loop of a few
// hundred instructions is only partly unrolled, and its indices then stop being constants)
template <class F, int... Q>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Q...>) {
  (f(std::integral_constant<int, Q>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

template <int MODE> constexpr int nt_row_floats() {
  return MODE == kSplit ? kSplitRowFloats : MODE == kSplit2 ? 2 * kBK / 2 : kLdsStride;
}

template <class TL, int MODE, bool SW = false, int PF = 1, class AL, class BL>
__device__ __forceinline__ void nt_mainloop_mode(AL& al, BL& bl, int K, float* As, float* Bs,
                                                 f32x16 (&acc)[TL::TM][TL::TN], float sa = 1.0f, float sb = 1.0f) {
  if constexpr (MODE == kBf16) nt_mainloop_bf16<TL>(al, bl, K, As, Bs, acc);
  else if constexpr (MODE == kSplit) nt_mainloop_split<TL, SW, 3, PF>(al, bl, K, As, Bs, acc);
  else if constexpr (MODE == kSplit2) nt_mainloop_split<TL, SW, 2, PF>(al, bl, K, As, Bs, acc, sa, sb);
  else nt_mainloop<TL>(al, bl, K, As, Bs, acc);
}

// Visit every accumulator element of this lane: fn(row_in_tile, col_in_tile, value).
template <class TL, class FN>
__device__ __forceinline__ void for_each_acc(const f32x16 (&acc)[TL::TM][TL::TN], FN&& fn) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int wm = wv / TL::WAVES_N, wn = wv % TL::WAVES_N;
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int i = 0; i < TL::TM; ++i)
#pragma unroll
    for (int j = 0; j < TL::TN; ++j)
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const int row = wm * TL::WM + i * 32 + (g & 3) + 8 * (g >> 2) + 4 * h;
        const int col = wn * TL::WN + j * 32 + r;
        fn(row, col, acc[i][j][g]);
      }
}

template <class TL>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[TL::TM][TL::TN]) {
#pragma unroll
  for (int i = 0; i < TL::TM; ++i)
#pragma unroll
    for (int j = 0; j < TL::TN; ++j)
#pragma unroll
      for (int g = 0; g < 16; ++g) acc[i][j][g] = 0.0f;
}

// ------------------------------------------------------------------ operand loaders (TN)
// A TN loader hands out float4 = 4 consecutive columns of one k-row of a [32][COLS] k-tile.
// COLS/4 threads cover a row, 1024/COLS rows per pass, COLS/32 passes ("slots") per k-tile.
template <int COLS>
struct TnGeom {
  static constexpr int TPR = COLS / 4;           // threads per k-row
  static constexpr int ROWS = 256 / TPR;         // k-rows per pass
  static constexpr int SLOTS = kBK / ROWS;
  static_assert(COLS == 64 || COLS == 128 || COLS == 256, "TN tile widths");
  __device__ static __forceinline__ int col4() { return (threadIdx.x % TPR) * 4; }
  __device__ static __forceinline__ int krow(int slot) { return threadIdx.x / TPR + ROWS * slot; }
};

template <int COLS, class TA = float>
struct KRowLoader {           // plain [K][cols] matrix; buffer loads based at the k-tile's first row (see RowLoaderT)
  const TA* p;
  long ld;
  int cols;
  int col0;
  unsigned vo_ = 0, rowstep_ = 0;
  typedef typename RawQuad<TA>::type Raw;
  __device__ __forceinline__ void init(int first_col, int /*k_begin*/) {
    col0 = first_col + TnGeom<COLS>::col4();
    vo_ = (unsigned)(((long)(threadIdx.x / TnGeom<COLS>::TPR) * ld + col0) * (long)sizeof(TA)) | (col0 < cols ? 0u : 0x80000000u);
    rowstep_ = (unsigned)(TnGeom<COLS>::ROWS * ld * (long)sizeof(TA));
  }
  __device__ __forceinline__ Raw load(int slot, int k0, int k_end) const {
    const long left = (long)k_end - k0;                               // wave-uniform: SALU
    long bytes = left > 0 ? ((left - 1) * ld + cols) * (long)sizeof(TA) : 0;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<TA*>(p + (long)k0 * ld), 0, (unsigned)(bytes < 0x7fffffffL ? bytes : 0x7fffffffL), 0x00020000);
    return ldraw_buffer<TA>(rs, vo_ + (unsigned)slot * rowstep_, 0u);
  }
};

template <int COLS>
struct ShiftedPixelLoader {   // [B*T*F][C] channels-last tensor read at pixel + (dt, df), zero outside
  const float* p;
  int T, F, C, dt, df;
  int col0;
  // (t, f) of each slot's current pixel row, advanced by 32 pixels per k-tile: no div/mod in the loop
  int t_[TnGeom<COLS>::SLOTS], f_[TnGeom<COLS>::SLOTS];
  __device__ __forceinline__ void init(int first_col, int k_begin) {
    col0 = first_col + TnGeom<COLS>::col4();
#pragma unroll
    for (int i = 0; i < TnGeom<COLS>::SLOTS; ++i) {
      const int k = k_begin + TnGeom<COLS>::krow(i);
      f_[i] = k % F;
      t_[i] = (k / F) % T;
    }
  }
  // must be called once per slot per k-tile, k-tiles in increasing order starting at k_begin
  __device__ __forceinline__ float4 load(int slot, int k0, int k_end) {
    const int k = k0 + TnGeom<COLS>::krow(slot);
    const int tt = t_[slot] + dt, ff = f_[slot] + df;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (k < k_end && col0 < C && tt >= 0 && tt < T && ff >= 0 && ff < F)
      v = *reinterpret_cast<const float4*>(p + ((long)k + dt * F + df) * C + col0);
    int f = f_[slot] + kBK, t = t_[slot];
    while (f >= F) { f -= F; ++t; }
    while (t >= T) t -= T;
    f_[slot] = f; t_[slot] = t;
    return v;
  }
};

template <int COLS>
struct ShiftedTimeLoader {    // [B][T][ld] sequence read at time t + dt (zero outside): h_{t-1} for dW_hh
  const float* p;
  long ld;
  int T, dt, cols;
  int col0;
  int t_[TnGeom<COLS>::SLOTS];
  __device__ __forceinline__ void init(int first_col, int k_begin) {
    col0 = first_col + TnGeom<COLS>::col4();
#pragma unroll
    for (int i = 0; i < TnGeom<COLS>::SLOTS; ++i) t_[i] = (k_begin + TnGeom<COLS>::krow(i)) % T;
  }
  __device__ __forceinline__ float4 load(int slot, int k0, int k_end) {
    const int k = k0 + TnGeom<COLS>::krow(slot);
    const int t = t_[slot] + dt;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (k < k_end && col0 < cols && t >= 0 && t < T)
      v = *reinterpret_cast<const float4*>(p + ((long)k + dt) * ld + col0);
    int tn = t_[slot] + kBK;
    while (tn >= T) tn -= T;
    t_[slot] = tn;
    return v;
  }
};

// ------------------------------------------------------------------ TN main loop
// Workgroup tile BM x BN (each 64 or 128), waves 2 x 2.  As: [32][BM], Bs: [32][BN] floats.
// k range [k_begin, k_end), k_begin a multiple of 32.
template <int BM, int BN, class AL, class BL>
__device__ __forceinline__ void tn_mainloop(AL& al, BL& bl, int k_begin, int k_end, float* As, float* Bs,
                                            f32x16 (&acc)[BM / 64][BN / 64]) {
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int SA = TnGeom<BM>::SLOTS, SB = TnGeom<BN>::SLOTS;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wm = wv >> 1, wn = wv & 1;
  const int r = lane & 31, h = lane >> 5;
  float4 ra[SA], rb[SB];
#pragma unroll
  for (int i = 0; i < SA; ++i) ra[i] = al.load(i, k_begin, k_end);
#pragma unroll
  for (int i = 0; i < SB; ++i) rb[i] = bl.load(i, k_begin, k_end);
  const int sta = (tid / TnGeom<BM>::TPR) * BM + TnGeom<BM>::col4();
  const int stb = (tid / TnGeom<BN>::TPR) * BN + TnGeom<BN>::col4();
  const float* a_rd = As + h * BM + wm * (BM / 2) + r;
  const float* b_rd = Bs + h * BN + wn * (BN / 2) + r;
  for (int k0 = k_begin; k0 < k_end; k0 += kBK) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < SA; ++i) *reinterpret_cast<float4*>(As + sta + i * TnGeom<BM>::ROWS * BM) = ra[i];
#pragma unroll
    for (int i = 0; i < SB; ++i) *reinterpret_cast<float4*>(Bs + stb + i * TnGeom<BN>::ROWS * BN) = rb[i];
    __syncthreads();
    if (k0 + kBK < k_end) {
#pragma unroll
      for (int i = 0; i < SA; ++i) ra[i] = al.load(i, k0 + kBK, k_end);
#pragma unroll
      for (int i = 0; i < SB; ++i) rb[i] = bl.load(i, k0 + kBK, k_end);
    }
#pragma unroll
    for (int s = 0; s < kBK / 2; ++s) {
      float fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[i] = a_rd[s * 2 * BM + i * 32];
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[j] = b_rd[s * 2 * BN + j * 32];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = mfma32(fa[i], fb[j], acc[i][j]);
    }
  }
}

// ------------------------------------------------------------------ TN main loop, fp32 as three bf16 terms
// Same exact three-term split as nt_mainloop_split.  The operands are k-major, so the LDS images stay
// [k][cols (+32 pad)] bf16 and the MFMA fragments (8 consecutive k of one column) come from the gfx950
// transposed read ds_read_b64_tr_b16: per 16-lane group, lane 4q+p supplies the address of row q,
// columns 4p..4p+3 of a 4 x 16 block and lane i receives column i of the 4 rows.  With a row stride of
// cols/2 + 16 dwords the four rows of a half-wave's two blocks fall on disjoint banks.
typedef short s16x4 __attribute__((ext_vector_type(4)));
template <int COLS> constexpr int tn_split_stride() { return COLS + 32; }                 // bf16 elements per k-row
template <int COLS, int NT = 3> constexpr int tn_split_floats() { return NT * kBK * tn_split_stride<COLS>() / 2; }

__device__ __forceinline__ s16x4 lds_read_tr(const __bf16* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4*)(reinterpret_cast<const short*>(p)));
}

// fragment of 16 k-rows starting at `p` (this lane's block-row address for rows +0..3; rows +4..7 one
// 4-row block further): elements j = 0..7 <-> k = 8h + j
__device__ __forceinline__ bf16x8 tr_fragment(const __bf16* p, int stride) {
  const s16x4 lo = lds_read_tr(p), hi = lds_read_tr(p + 4 * stride);
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

// NT = 3: the exact three-term split; 2: two scaled fp16 terms; 1: operands rounded to bf16 (mixed precision)
template <int COLS, int NT>
__device__ __forceinline__ void tn_split_store(__bf16* img, int off, const float4& v, float scale = 1.0f) {
  constexpr int IMG = kBK * tn_split_stride<COLS>();
  if constexpr (NT == 3) {
    const Split3 sp = split3(v);
    *reinterpret_cast<uint2*>(img + off) = sp.hi;
    *reinterpret_cast<uint2*>(img + off + IMG) = sp.mid;
    *reinterpret_cast<uint2*>(img + off + 2 * IMG) = sp.lo;
  } else if constexpr (NT == 2) {
    const Split2 sp = split2(v, scale);
    *reinterpret_cast<uint2*>(img + off) = sp.hi;
    *reinterpret_cast<uint2*>(img + off + IMG) = sp.lo;
  } else {
    *reinterpret_cast<bf16x4*>(img + off) = to_bf16x4(v);
  }
}

template <int COLS, int NT>
__device__ __forceinline__ void tn_split_store(__bf16* img, int off, const uint2& raw, float scale = 1.0f) {
  if constexpr (NT == 1) *reinterpret_cast<uint2*>(img + off) = raw;
  else tn_split_store<COLS, NT>(img, off, widen(raw), scale);
}

template <int BM, int BN, int NT, int PF = 1, class AL, class BL>
__device__ __forceinline__ void tn_mainloop_split(AL& al, BL& bl, int k_begin, int k_end, float* As_f, float* Bs_f,
                                                  f32x16 (&acc)[BM / 64][BN / 64], float sa = 1.0f, float sb = 1.0f) {
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int SA = TnGeom<BM>::SLOTS, SB = TnGeom<BN>::SLOTS;
  constexpr int STA = tn_split_stride<BM>(), STB = tn_split_stride<BN>();
  constexpr int IMGA = kBK * STA, IMGB = kBK * STB;
  __bf16* As = reinterpret_cast<__bf16*>(As_f);
  __bf16* Bs = reinterpret_cast<__bf16*>(Bs_f);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wm = wv >> 1, wn = wv & 1;
  const int h = lane >> 5, g1 = (lane >> 4) & 1, q = (lane & 15) >> 2, p4 = (lane & 3) * 4;
  // PF k-tiles of operand quads in flight (see nt_mainloop_split; PF = 2 needs the buffer loaders)
  decltype(al.load(0, 0, 0)) ra[PF][SA];
  decltype(bl.load(0, 0, 0)) rb[PF][SB];
#pragma unroll
  for (int u = 0; u < PF; ++u) {
#pragma unroll
    for (int i = 0; i < SA; ++i) ra[u][i] = al.load(i, k_begin + u * kBK, k_end);
#pragma unroll
    for (int i = 0; i < SB; ++i) rb[u][i] = bl.load(i, k_begin + u * kBK, k_end);
  }
  const int sta = (tid / TnGeom<BM>::TPR) * STA + TnGeom<BM>::col4();
  const int stb = (tid / TnGeom<BN>::TPR) * STB + TnGeom<BN>::col4();
  const __bf16* a_rd = As + (8 * h + q) * STA + wm * (BM / 2) + 16 * g1 + p4;
  const __bf16* b_rd = Bs + (8 * h + q) * STB + wn * (BN / 2) + 16 * g1 + p4;
  auto tile = [&](auto uc, int k0) {
    constexpr int u = decltype(uc)::value;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < SA; ++i) tn_split_store<BM, NT>(As, sta + i * TnGeom<BM>::ROWS * STA, ra[u][i], sa);
#pragma unroll
    for (int i = 0; i < SB; ++i) tn_split_store<BN, NT>(Bs, stb + i * TnGeom<BN>::ROWS * STB, rb[u][i], sb);
    __syncthreads();
    if (PF == 2 || k0 + kBK < k_end) {
#pragma unroll
      for (int i = 0; i < SA; ++i) ra[u][i] = al.load(i, k0 + PF * kBK, k_end);
#pragma unroll
      for (int i = 0; i < SB; ++i) rb[u][i] = bl.load(i, k0 + PF * kBK, k_end);
    }
    if constexpr (PF == 2) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kk = 0; kk < kBK / 16; ++kk) {
      bf16x8 fa[TM][NT], fb[TN][NT];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int c = 0; c < NT; ++c) fa[i][c] = tr_fragment(a_rd + c * IMGA + kk * 16 * STA + i * 32, STA);
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int c = 0; c < NT; ++c) fb[j][c] = tr_fragment(b_rd + c * IMGB + kk * 16 * STB + j * 32, STB);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = mfma_terms<NT>(fa[i], fb[j], acc[i][j]);
    }
    if constexpr (PF == 2) __builtin_amdgcn_sched_barrier(0);
  };
  int k0 = k_begin;
  if constexpr (PF == 2) {
    for (; k0 + kBK < k_end; k0 += 2 * kBK) {
      tile(std::integral_constant<int, 0>{}, k0);
      tile(std::integral_constant<int, 1>{}, k0 + kBK);
    }
  }
  for (; k0 < k_end; k0 += kBK) tile(std::integral_constant<int, 0>{}, k0);
}

template <int MODE, int COLS> constexpr int tn_lds_floats() {
  return MODE == kSplit ? tn_split_floats<COLS, 3>() : MODE == kSplit2 ? tn_split_floats<COLS, 2>()
       : MODE == kBf16 ? tn_split_floats<COLS, 1>() : kBK * COLS;
}

template <int MODE, int BM, int BN, int PF = 1, class AL, class BL>
__device__ __forceinline__ void tn_mainloop_mode(AL& al, BL& bl, int k_begin, int k_end, float* As, float* Bs,
                                                 f32x16 (&acc)[BM / 64][BN / 64], float sa = 1.0f, float sb = 1.0f) {
  if constexpr (MODE == kSplit) tn_mainloop_split<BM, BN, 3, PF>(al, bl, k_begin, k_end, As, Bs, acc);
  else if constexpr (MODE == kSplit2) tn_mainloop_split<BM, BN, 2, PF>(al, bl, k_begin, k_end, As, Bs, acc, sa, sb);
  else if constexpr (MODE == kBf16) tn_mainloop_split<BM, BN, 1, PF>(al, bl, k_begin, k_end, As, Bs, acc);
  else tn_mainloop<BM, BN>(al, bl, k_begin, k_end, As, Bs, acc);
}

// ------------------------------------------------------------------ NN main loop (64 x 64 tile)
// C[m][n] = sum_k A[m][k] * B[k][n]: A k-contiguous (NT-style image [64][36], ds_read_b128),
// B k-major (TN-style image [32][64], ds_read_b32).  Waves 2 x 2, one 32x32 MFMA tile each.
// Within an 8-wide k block MFMA j takes k = {j, 4 + j} for the two lane halves, on both operands.
template <class AL, class BL>
__device__ __forceinline__ void nn_mainloop_64(AL& al, BL& bl, int K, float* As, float* Bs, f32x16& acc) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wm = wv >> 1, wn = wv & 1;
  const int r = lane & 31, h = lane >> 5;
  const int nk = (K + kBK - 1) / kBK;
  float4 ra[2], rb[2];
  ra[0] = al.load(0, 0); ra[1] = al.load(1, 0);
  rb[0] = bl.load(0, 0, K); rb[1] = bl.load(1, 0, K);
  const int sta = (tid >> 3) * kLdsStride + (tid & 7) * 4;
  const int stb = (tid / TnGeom<64>::TPR) * 64 + TnGeom<64>::col4();
  const float* a_rd = As + (wm * 32 + r) * kLdsStride + h * 4;
  const float* b_rd = Bs + (h * 4) * 64 + wn * 32 + r;
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();
    *reinterpret_cast<float4*>(As + sta) = ra[0];
    *reinterpret_cast<float4*>(As + sta + 32 * kLdsStride) = ra[1];
    *reinterpret_cast<float4*>(Bs + stb) = rb[0];
    *reinterpret_cast<float4*>(Bs + stb + TnGeom<64>::ROWS * 64) = rb[1];
    __syncthreads();
    if (kt + 1 < nk) {
      ra[0] = al.load(0, kt + 1); ra[1] = al.load(1, kt + 1);
      rb[0] = bl.load(0, (kt + 1) * kBK, K); rb[1] = bl.load(1, (kt + 1) * kBK, K);
    }
#pragma unroll
    for (int k8 = 0; k8 < kBK / 8; ++k8) {
      const float4 fa = *reinterpret_cast<const float4*>(a_rd + k8 * 8);
      acc = mfma32(fa.x, b_rd[(k8 * 8 + 0) * 64], acc);
      acc = mfma32(fa.y, b_rd[(k8 * 8 + 1) * 64], acc);
      acc = mfma32(fa.z, b_rd[(k8 * 8 + 2) * 64], acc);
      acc = mfma32(fa.w, b_rd[(k8 * 8 + 3) * 64], acc);
    }
  }
}

// Visit the accumulators of a TN tile: fn(row_in_tile, col_in_tile, value).
template <int BM, int BN, class FN>
__device__ __forceinline__ void tn_for_each_acc(const f32x16 (&acc)[BM / 64][BN / 64], FN&& fn) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int wm = wv >> 1, wn = wv & 1;
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int i = 0; i < BM / 64; ++i)
#pragma unroll
    for (int j = 0; j < BN / 64; ++j)
#pragma unroll
      for (int g = 0; g < 16; ++g)
        fn(wm * (BM / 2) + i * 32 + (g & 3) + 8 * (g >> 2) + 4 * h, wn * (BN / 2) + j * 32 + r, acc[i][j][g]);
}

}  // namespace pe
