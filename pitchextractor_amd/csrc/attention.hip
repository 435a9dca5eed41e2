// Fused self-attention of nn.TransformerEncoderLayer (reference model.py:231-239; torch's MultiheadAttention:
// softmax(Q K^T / sqrt(dh)) -> dropout -> . V per head) for the geometry of this model: T = 192 frames, dh = 64.
//
// The unfused path materialises the (B*H*T) x T probability matrix in HBM (302 MB per layer and branch at B = 256)
// and walks it five times forward and six times backward.  Here three workgroups share one (batch, head): Q / K / V
// (48 KB each, fp32) pass through LDS or registers, scores never leave the register file, and what reaches HBM is O, the
// per-row log-sum-exp (768 B per head) and, when dropout is live, the 1-byte keep mask the parity tests export.
//
// Arithmetic: exact fp32 on v_mfma_f32_16x16x4_f32 (attention is 3 % of the step's FLOPs; the operands change every
// step, so a three-term bf16 split would cost more LDS than it saves MFMA time).  MFMA maps (CDNA guide section 3):
// A lane l = A[i = l & 15][k = l >> 4], B lane l = B[k = l >> 4][j = l & 15], C/D lane l reg r = C[4 (l >> 4) + r][l & 15].
//
// "Accumulator as the next operand": a 16 x 16 result X has its column on the lane and rows 4g .. 4g+3 in the four
// registers of lane group g, which is exactly the B-operand layout of a following MFMA that sums over X's ROW index:
// register q of X feeds an MFMA whose k-slot g means row 4g + q, and the A operand reads the matching rows.  So the
// products that sum over the score tile's rows take it straight from the registers:
//     forward:   S^T = K Q^T  (keys x queries)  ->  O^T  = V^T P^T        (sum over keys)
//     backward:  S   = Q K^T  (queries x keys)  ->  dV^T = dO^T Pd, dK^T = Q^T dS   (sum over queries)
//                S^T again (keys x queries)     ->  dQ^T = K^T dS^T       (sum over keys)
// and no score tile is ever transposed through LDS.  The backward recomputes S twice (two kernels) instead:
// deterministic, no atomics, no cross-workgroup reduction.
#include "common.h"
#include "act16.h"

// XCD-aware block order (workgroups b and b + 8 share an XCD and its L2): give every XCD a contiguous run of
// logical blocks, so the workgroups that read the same K / V rows hit the same L2.  Bijective for any grid size.
__device__ __forceinline__ int xcd_run(int bid, int n) {
  const int q = n >> 3, r = n & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

namespace {

constexpr int kDh = 64;
constexpr int kStr = 68;                 // LDS row stride in floats (64 + one 16-byte pad)

typedef float f32x4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4v mfma16(float a, float b, f32x4v c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// Four k-slots at once: c += sum_n a_n * b_n, the four exact-fp32 MFMAs above in the order n = 0..3, or -- mixed
// precision, as autocast runs these matmuls -- ONE v_mfma_f32_16x16x16_bf16 on the operands rounded to bf16.  The maps
// agree: the fp32 MFMA n of a group reads A[i][k = g] = a_n of lane (i, g), the bf16 MFMA reads A[i][k = 4 g + n] from
// the n-th element of lane (i, g)'s quad, likewise B; the accumulator layouts are the same, so "accumulator as the next
// operand" carries over (the four registers of a score tile, rounded, ARE the bf16 B quad).
typedef short bf16x4s __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x4s pack4(float a, float b, float c, float d) {
  typedef unsigned u2 __attribute__((ext_vector_type(2)));
  const u2 v = {pe::pack_bf16_rne(a, b), pe::pack_bf16_rne(c, d)};
  return __builtin_bit_cast(bf16x4s, v);
}
template <bool BF>
__device__ __forceinline__ f32x4v mm4(float a0, float a1, float a2, float a3, float b0, float b1, float b2, float b3,
                                      f32x4v c) {
  if constexpr (BF) {
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(pack4(a0, a1, a2, a3), pack4(b0, b1, b2, b3), c, 0, 0, 0);
  } else {
    c = mfma16(a0, b0, c); c = mfma16(a1, b1, c); c = mfma16(a2, b2, c); c = mfma16(a3, b3, c);
    return c;
  }
}
template <bool BF>
__device__ __forceinline__ f32x4v mm4(const float4& a, const float4& b, f32x4v c) {
  return mm4<BF>(a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c);
}

struct AttnArgs {
  const float* qkv;          // [B*T][ld_qkv]: Q at col h*64, K at D + h*64, V at 2D + h*64
  long ld_qkv;
  int D;                     // H * 64
  float* o;                  // forward out / backward in: [B*T][ld_o], head h at col h*64
  long ld_o;
  float* lse;                // [B*H*T] log-sum-exp of the scaled scores
  const uint8_t* mask_in;    // optional [B*H*T][T] keep mask to replay
  uint8_t* mask_out;         // optional, written when dropout is live
  const float* d_o;          // backward: gradient of o
  float* dqkv;               // backward out, same layout as qkv
  const uint8_t* mask;       // backward: the forward's keep mask (NULL: no dropout)
  int B, H;
  float scale, p_drop, keep_scale;
  unsigned long long seed, offset;
};

// stage a [T][64] fp32 matrix (global row stride ld) into LDS rows of kStr floats
template <int T>
__device__ __forceinline__ void stage64(const float* __restrict__ src, long ld, float* dst) {
  for (int idx = threadIdx.x; idx < T * 16; idx += 256) {
    const int t = idx >> 4, c4 = (idx & 15) * 4;
    *reinterpret_cast<float4*>(dst + t * kStr + c4) = *reinterpret_cast<const float4*>(src + (long)t * ld + c4);
  }
}

// delta[q] = sum_d dO[q][d] * O[q][d]  (16 lanes per row, one float4 each)
template <int T>
__device__ __forceinline__ void row_dots(const float* __restrict__ a, long lda, const float* __restrict__ b, long ldb,
                                         float* out) {
  for (int idx = threadIdx.x; idx < T * 16; idx += 256) {
    const int t = idx >> 4, c4 = (idx & 15) * 4;
    const float4 x = *reinterpret_cast<const float4*>(a + (long)t * lda + c4);
    const float4 y = *reinterpret_cast<const float4*>(b + (long)t * ldb + c4);
    float s = x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
    s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 8, 64);
    if ((idx & 15) == 0) out[t] = s;
  }
}

// Work decomposition: a workgroup = (batch, head, third of the tiles): 4 waves, one 16-row tile each.  The rows it
// sweeps (keys in the forward / dQ kernels, queries in the dK / dV kernel) pass through LDS in two halves of 96
// rows, so a workgroup needs 52 KB and three of them share a CU: their MFMA, VALU (softmax, Philox) and load
// phases overlap.  (A first version kept whole K / V per workgroup -- 157 KB, one workgroup and one wave per SIMD --
// and ran at a third of this rate: nothing hid the LDS and exp latencies.)

// ------------------------------------------------------------------------------------------------ forward
template <int T, bool BF>
__global__ __launch_bounds__(256, 3) void attn_fwd_kernel(const AttnArgs a) {
  constexpr int NT = T / 16, HR = T / 2, NH = NT / 2;       // HR rows / NH tiles per half
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ks = smem;                      // [HR][kStr]
  float* Vs = Ks + HR * kStr;            // [HR][kStr]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int blk = xcd_run(blockIdx.x, gridDim.x);              // the three parts of a head share an XCD's L2
  const int bh = blk / 3, part = blk - bh * 3, b = bh / a.H, h = bh - b * a.H;
  const float* base = a.qkv + (long)b * T * a.ld_qkv + h * kDh;
  const int q0 = (part * 4 + wv) * 16;
  const long row = (long)bh * T + q0 + j;
  const float c2 = a.scale * 1.44269504088896340736f;
  const bool drop = a.p_drop > 0.f;

  stage64<HR>(base + a.D, a.ld_qkv, Ks);
  stage64<HR>(base + 2 * a.D, a.ld_qkv, Vs);
  float4 qf[4];
  {
    const float* qrow = base + (long)(q0 + j) * a.ld_qkv + 4 * g;
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const float4*>(qrow + 16 * s);
  }
  __syncthreads();
  // ---- S^T tiles: acc[kt][r] = score(key 16 kt + 4 g + r, query q0 + j); two tiles at a time (independent chains)
  f32x4v acc[NT];
#pragma unroll
  for (int kt = 0; kt < NT; ++kt) acc[kt] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    if (half == 1) {
      __syncthreads();                                      // every wave is done with keys 0 .. HR-1
      stage64<HR>(base + a.D + (long)HR * a.ld_qkv, a.ld_qkv, Ks);
      __syncthreads();
    }
#pragma unroll
    for (int kl = 0; kl < NH; kl += 2) {
      const int kt = half * NH + kl;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float4 k0 = *reinterpret_cast<const float4*>(Ks + (16 * kl + j) * kStr + 16 * s + 4 * g);
        const float4 k1 = *reinterpret_cast<const float4*>(Ks + (16 * kl + 16 + j) * kStr + 16 * s + 4 * g);
        acc[kt] = mm4<BF>(k0, qf[s], acc[kt]);
        acc[kt + 1] = mm4<BF>(k1, qf[s], acc[kt + 1]);
      }
    }
  }
  // ---- softmax over the query's 192 keys: 48 in this lane, the rest in lanes j + 16, j + 32, j + 48
  float m = acc[0][0];
#pragma unroll
  for (int kt = 0; kt < NT; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) m = fmaxf(m, acc[kt][r]);
  m = fmaxf(m, __shfl_xor(m, 16, 64));
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int kt = 0; kt < NT; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float p = __builtin_amdgcn_exp2f((acc[kt][r] - m) * c2);
      acc[kt][r] = p;
      sum += p;
    }
  sum += __shfl_xor(sum, 16, 64);
  sum += __shfl_xor(sum, 32, 64);
  const float inv = 1.0f / sum;
  if (g == 0) a.lse[row] = m * a.scale + logf(sum);
  // ---- normalise (+ dropout: quad = 4 consecutive keys of one row, counter offset + row * T/4 + key/4)
#pragma unroll
  for (int kt = 0; kt < NT; ++kt) {
    if (drop) {
      bool keep[4];
      if (a.mask_in) {
        const uchar4 mk = *reinterpret_cast<const uchar4*>(a.mask_in + row * T + 16 * kt + 4 * g);
        keep[0] = mk.x; keep[1] = mk.y; keep[2] = mk.z; keep[3] = mk.w;
      } else {
        uint32_t rnd[4];
        philox4(a.seed, a.offset + (unsigned long long)(row * (T / 4) + 4 * kt + g), rnd);
#pragma unroll
        for (int r = 0; r < 4; ++r) keep[r] = pe_dropout_keep(rnd[r], a.p_drop);
      }
      if (a.mask_out)
        *reinterpret_cast<uchar4*>(a.mask_out + row * T + 16 * kt + 4 * g) =
            make_uchar4(keep[0], keep[1], keep[2], keep[3]);
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[kt][r] = keep[r] ? (acc[kt][r] * inv) * a.keep_scale : 0.f;
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[kt][r] *= inv;
    }
  }
  // ---- O^T[d][query] = sum_key V[key][d] Pd[key][query]; d-tile dt holds d = 4 i + dt (one float4 of V per lane)
  f32x4v o[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    if (half == 1) {
      __syncthreads();
      stage64<HR>(base + 2 * a.D + (long)HR * a.ld_qkv, a.ld_qkv, Vs);
      __syncthreads();
    }
#pragma unroll
    for (int kl = 0; kl < NH; ++kl) {
      const int kt = half * NH + kl;
      float4 vf[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) vf[r] = *reinterpret_cast<const float4*>(Vs + (16 * kl + 4 * g + r) * kStr + 4 * j);
      const f32x4v pk = acc[kt];
      o[0] = mm4<BF>(vf[0].x, vf[1].x, vf[2].x, vf[3].x, pk[0], pk[1], pk[2], pk[3], o[0]);
      o[1] = mm4<BF>(vf[0].y, vf[1].y, vf[2].y, vf[3].y, pk[0], pk[1], pk[2], pk[3], o[1]);
      o[2] = mm4<BF>(vf[0].z, vf[1].z, vf[2].z, vf[3].z, pk[0], pk[1], pk[2], pk[3], o[2]);
      o[3] = mm4<BF>(vf[0].w, vf[1].w, vf[2].w, vf[3].w, pk[0], pk[1], pk[2], pk[3], o[3]);
    }
  }
  // lane (j, g) holds d = 16 g + 4 r + dt of query q0 + j: 64 contiguous bytes
  float* orow = a.o + ((long)b * T + q0 + j) * a.ld_o + h * kDh + 16 * g;
#pragma unroll
  for (int r = 0; r < 4; ++r) *reinterpret_cast<float4*>(orow + 4 * r) = make_float4(o[0][r], o[1][r], o[2][r], o[3][r]);
}

// ------------------------------------------------------------------------------------------------ backward: dK, dV
// Wave owns one key tile; scores with QUERIES on the rows, keys on the lane.  The K / V fragments of the key tile are
// loaded from global into registers once; Q and dO pass through LDS in two halves; lse / delta of all queries in LDS.
template <int T, bool BF>
__global__ __launch_bounds__(256, 3) void attn_bwd_kv_kernel(const AttnArgs a) {
  constexpr int NT = T / 16, HR = T / 2, NH = NT / 2;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Qs = smem;                      // [HR][kStr]
  float* Gs = Qs + HR * kStr;            // [HR][kStr] dO
  float* lse_s = Gs + HR * kStr;         // [T]
  float* dl_s = lse_s + T;               // [T] delta
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int blk = xcd_run(blockIdx.x, gridDim.x);              // the three parts of a head share an XCD's L2
  const int bh = blk / 3, part = blk - bh * 3, b = bh / a.H, h = bh - b * a.H;
  const float* base = a.qkv + (long)b * T * a.ld_qkv + h * kDh;
  const float* dob = a.d_o + (long)b * T * a.ld_o + h * kDh;
  const int k0 = (part * 4 + wv) * 16;
  stage64<HR>(base, a.ld_qkv, Qs);
  stage64<HR>(dob, a.ld_o, Gs);
  row_dots<T>(dob, a.ld_o, a.o + (long)b * T * a.ld_o + h * kDh, a.ld_o, dl_s);
  for (int t = tid; t < T; t += 256) lse_s[t] = a.lse[(long)bh * T + t];
  // B operands of this key tile: K[k0 + j][16 s + 4 g ..] and V[k0 + j][16 s + 4 g ..]
  float4 kf[4], vf[4];
  {
    const float* krow = base + a.D + (long)(k0 + j) * a.ld_qkv + 4 * g;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      kf[s] = *reinterpret_cast<const float4*>(krow + 16 * s);
      vf[s] = *reinterpret_cast<const float4*>(krow + a.D + 16 * s);
    }
  }
  __syncthreads();
  const bool drop = a.mask != nullptr;
  f32x4v dk[4], dv[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) dk[dt] = dv[dt] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int half = 0; half < 2; ++half) {
    if (half == 1) {
      __syncthreads();
      stage64<HR>(base + (long)HR * a.ld_qkv, a.ld_qkv, Qs);
      stage64<HR>(dob + (long)HR * a.ld_o, a.ld_o, Gs);
      __syncthreads();
    }
#pragma unroll 2
    for (int ql = 0; ql < NH; ++ql) {
      const int qb = ql * 16, q0 = half * HR + qb;
      // S[query 4 g + r][key j] and dP = dO V^T in the same layout
      f32x4v sc = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float4 qa = *reinterpret_cast<const float4*>(Qs + (qb + j) * kStr + 16 * s + 4 * g);
        const float4 ga = *reinterpret_cast<const float4*>(Gs + (qb + j) * kStr + 16 * s + 4 * g);
        sc = mm4<BF>(qa, kf[s], sc);
        dp = mm4<BF>(ga, vf[s], dp);
      }
      f32x4v pd, ds;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int q = q0 + 4 * g + r;
        const float p = __builtin_amdgcn_exp2f((sc[r] * a.scale - lse_s[q]) * 1.44269504088896340736f);
        float keep = 1.0f;
        if (drop) keep = a.mask[((long)bh * T + q) * T + k0 + j] ? a.keep_scale : 0.f;
        pd[r] = p * keep;                                           // dropped-out, rescaled probability
        ds[r] = p * (dp[r] * keep - dl_s[q]) * a.scale;             // gradient of the raw score
      }
      // dV^T[d][key] += dO[query][d] Pd[query][key];  dK^T[d][key] += Q[query][d] dS[query][key]
      float4 gq[4], qq[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        gq[r] = *reinterpret_cast<const float4*>(Gs + (qb + 4 * g + r) * kStr + 4 * j);
        qq[r] = *reinterpret_cast<const float4*>(Qs + (qb + 4 * g + r) * kStr + 4 * j);
      }
      dv[0] = mm4<BF>(gq[0].x, gq[1].x, gq[2].x, gq[3].x, pd[0], pd[1], pd[2], pd[3], dv[0]);
      dk[0] = mm4<BF>(qq[0].x, qq[1].x, qq[2].x, qq[3].x, ds[0], ds[1], ds[2], ds[3], dk[0]);
      dv[1] = mm4<BF>(gq[0].y, gq[1].y, gq[2].y, gq[3].y, pd[0], pd[1], pd[2], pd[3], dv[1]);
      dk[1] = mm4<BF>(qq[0].y, qq[1].y, qq[2].y, qq[3].y, ds[0], ds[1], ds[2], ds[3], dk[1]);
      dv[2] = mm4<BF>(gq[0].z, gq[1].z, gq[2].z, gq[3].z, pd[0], pd[1], pd[2], pd[3], dv[2]);
      dk[2] = mm4<BF>(qq[0].z, qq[1].z, qq[2].z, qq[3].z, ds[0], ds[1], ds[2], ds[3], dk[2]);
      dv[3] = mm4<BF>(gq[0].w, gq[1].w, gq[2].w, gq[3].w, pd[0], pd[1], pd[2], pd[3], dv[3]);
      dk[3] = mm4<BF>(qq[0].w, qq[1].w, qq[2].w, qq[3].w, ds[0], ds[1], ds[2], ds[3], dk[3]);
    }
  }
  float* drow = a.dqkv + ((long)b * T + k0 + j) * a.ld_qkv + a.D + h * kDh + 16 * g;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    *reinterpret_cast<float4*>(drow + 4 * r) = make_float4(dk[0][r], dk[1][r], dk[2][r], dk[3][r]);
    *reinterpret_cast<float4*>(drow + a.D + 4 * r) = make_float4(dv[0][r], dv[1][r], dv[2][r], dv[3][r]);
  }
}

// ------------------------------------------------------------------------------------------------ backward: dQ
// Wave owns one query tile; scores with KEYS on the rows, queries on the lane (the forward's orientation).
template <int T, bool BF>
__global__ __launch_bounds__(256, 3) void attn_bwd_q_kernel(const AttnArgs a) {
  constexpr int NT = T / 16, HR = T / 2, NH = NT / 2;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ks = smem;                      // [HR][kStr]
  float* Vs = Ks + HR * kStr;            // [HR][kStr]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int blk = xcd_run(blockIdx.x, gridDim.x);              // the three parts of a head share an XCD's L2
  const int bh = blk / 3, part = blk - bh * 3, b = bh / a.H, h = bh - b * a.H;
  const float* base = a.qkv + (long)b * T * a.ld_qkv + h * kDh;
  const float* dob = a.d_o + (long)b * T * a.ld_o + h * kDh;
  const int q0 = (part * 4 + wv) * 16;
  const long row = (long)bh * T + q0 + j;
  stage64<HR>(base + a.D, a.ld_qkv, Ks);
  stage64<HR>(base + 2 * a.D, a.ld_qkv, Vs);
  float4 qf[4], gf[4];
  float delta = 0.f;
  {
    const float* qrow = base + (long)(q0 + j) * a.ld_qkv + 4 * g;
    const float* grow = dob + (long)(q0 + j) * a.ld_o + 4 * g;
    const float* orow = a.o + ((long)b * T + q0 + j) * a.ld_o + h * kDh + 4 * g;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      qf[s] = *reinterpret_cast<const float4*>(qrow + 16 * s);
      gf[s] = *reinterpret_cast<const float4*>(grow + 16 * s);
      const float4 ov = *reinterpret_cast<const float4*>(orow + 16 * s);
      delta += gf[s].x * ov.x + gf[s].y * ov.y + gf[s].z * ov.z + gf[s].w * ov.w;
    }
    delta += __shfl_xor(delta, 16, 64);                    // the row's 64 d are spread over the 4 lane groups
    delta += __shfl_xor(delta, 32, 64);
  }
  const float lse = a.lse[row];
  __syncthreads();
  const bool drop = a.mask != nullptr;
  f32x4v dq[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) dq[dt] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int half = 0; half < 2; ++half) {
    if (half == 1) {
      __syncthreads();
      stage64<HR>(base + a.D + (long)HR * a.ld_qkv, a.ld_qkv, Ks);
      stage64<HR>(base + 2 * a.D + (long)HR * a.ld_qkv, a.ld_qkv, Vs);
      __syncthreads();
    }
#pragma unroll 2
    for (int kl = 0; kl < NH; ++kl) {
      const int kb = kl * 16, k0 = half * HR + kb;
      f32x4v sc = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float4 ka = *reinterpret_cast<const float4*>(Ks + (kb + j) * kStr + 16 * s + 4 * g);
        const float4 va = *reinterpret_cast<const float4*>(Vs + (kb + j) * kStr + 16 * s + 4 * g);
        sc = mm4<BF>(ka, qf[s], sc);
        dp = mm4<BF>(va, gf[s], dp);
      }
      float keep[4] = {1.f, 1.f, 1.f, 1.f};
      if (drop) {
        const uchar4 mk = *reinterpret_cast<const uchar4*>(a.mask + row * T + k0 + 4 * g);
        keep[0] = mk.x ? a.keep_scale : 0.f; keep[1] = mk.y ? a.keep_scale : 0.f;
        keep[2] = mk.z ? a.keep_scale : 0.f; keep[3] = mk.w ? a.keep_scale : 0.f;
      }
      f32x4v ds;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = __builtin_amdgcn_exp2f((sc[r] * a.scale - lse) * 1.44269504088896340736f);   // key k0+4g+r, query q0+j
        ds[r] = p * (dp[r] * keep[r] - delta) * a.scale;
      }
      // dQ^T[d][query] += K[key][d] dS^T[key][query]
      float4 kq[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) kq[r] = *reinterpret_cast<const float4*>(Ks + (kb + 4 * g + r) * kStr + 4 * j);
      dq[0] = mm4<BF>(kq[0].x, kq[1].x, kq[2].x, kq[3].x, ds[0], ds[1], ds[2], ds[3], dq[0]);
      dq[1] = mm4<BF>(kq[0].y, kq[1].y, kq[2].y, kq[3].y, ds[0], ds[1], ds[2], ds[3], dq[1]);
      dq[2] = mm4<BF>(kq[0].z, kq[1].z, kq[2].z, kq[3].z, ds[0], ds[1], ds[2], ds[3], dq[2]);
      dq[3] = mm4<BF>(kq[0].w, kq[1].w, kq[2].w, kq[3].w, ds[0], ds[1], ds[2], ds[3], dq[3]);
    }
  }
  float* drow = a.dqkv + ((long)b * T + q0 + j) * a.ld_qkv + h * kDh + 16 * g;
#pragma unroll
  for (int r = 0; r < 4; ++r) *reinterpret_cast<float4*>(drow + 4 * r) = make_float4(dq[0][r], dq[1][r], dq[2][r], dq[3][r]);
}

bool attn_shape_ok(int T, int dh) { return T == 192 && dh == kDh; }

template <class K>
int set_lds(K kernel, size_t bytes) {
  return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)bytes);
}

}  // namespace

extern "C" int pe_attn_supported(int T, int dh) { return attn_shape_ok(T, dh) ? 1 : 0; }

template <bool BF>
static int attn_fwd_impl(const float* qkv, long ld_qkv, float* o, long ld_o, float* lse, const unsigned char* mask_in,
                         unsigned char* mask_out, int B, int T, int H, int dh, float scale, float p_drop,
                         unsigned long long seed, unsigned long long offset, void* stream) {
  if (!qkv || !o || !lse || B <= 0 || H <= 0 || p_drop < 0.f || p_drop >= 1.f) return PE_E_ARG;
  if (!attn_shape_ok(T, dh) || (ld_qkv & 3) || (ld_o & 3) || ld_qkv < 3L * H * dh || ld_o < (long)H * dh)
    return PE_E_UNSUPPORTED;
  AttnArgs a{};
  a.qkv = qkv; a.ld_qkv = ld_qkv; a.D = H * dh; a.o = o; a.ld_o = ld_o; a.lse = lse;
  a.mask_in = mask_in; a.mask_out = mask_out; a.B = B; a.H = H; a.scale = scale; a.p_drop = p_drop;
  a.keep_scale = 1.0f / (1.0f - p_drop); a.seed = seed; a.offset = offset;
  const size_t lds = (size_t)2 * 96 * kStr * sizeof(float);
  static bool attr = false;
  if (!attr) { PE_CHECK_HIP((hipError_t)set_lds(&attn_fwd_kernel<192, BF>, lds)); attr = true; }
  hipLaunchKernelGGL((attn_fwd_kernel<192, BF>), dim3(B * H * 3), dim3(256), lds, pe_stream(stream), a);
  PE_LAUNCH_CHECK();
  return PE_OK;
}

extern "C" int pe_attn_fwd(const float* qkv, long ld_qkv, float* o, long ld_o, float* lse, const unsigned char* mask_in,
                           unsigned char* mask_out, int B, int T, int H, int dh, float scale, float p_drop,
                           unsigned long long seed, unsigned long long offset, void* stream) {
  return attn_fwd_impl<false>(qkv, ld_qkv, o, ld_o, lse, mask_in, mask_out, B, T, H, dh, scale, p_drop, seed, offset,
                              stream);
}
extern "C" int pe_attn_fwd_bf16(const float* qkv, long ld_qkv, float* o, long ld_o, float* lse,
                                const unsigned char* mask_in, unsigned char* mask_out, int B, int T, int H, int dh,
                                float scale, float p_drop, unsigned long long seed, unsigned long long offset,
                                void* stream) {
  return attn_fwd_impl<true>(qkv, ld_qkv, o, ld_o, lse, mask_in, mask_out, B, T, H, dh, scale, p_drop, seed, offset,
                             stream);
}

template <bool BF>
static int attn_bwd_impl(const float* qkv, long ld_qkv, const float* o, const float* d_o, long ld_o, const float* lse,
                         const unsigned char* mask, float* dqkv, int B, int T, int H, int dh, float scale,
                         float p_drop, void* stream) {
  if (!qkv || !o || !d_o || !lse || !dqkv || B <= 0 || H <= 0 || p_drop < 0.f || p_drop >= 1.f) return PE_E_ARG;
  if (p_drop > 0.f && !mask) return PE_E_ARG;
  if (!attn_shape_ok(T, dh) || (ld_qkv & 3) || (ld_o & 3) || ld_qkv < 3L * H * dh || ld_o < (long)H * dh)
    return PE_E_UNSUPPORTED;
  AttnArgs a{};
  a.qkv = qkv; a.ld_qkv = ld_qkv; a.D = H * dh; a.o = const_cast<float*>(o); a.ld_o = ld_o;
  a.lse = const_cast<float*>(lse); a.d_o = d_o; a.dqkv = dqkv; a.mask = p_drop > 0.f ? mask : nullptr;
  a.B = B; a.H = H; a.scale = scale; a.p_drop = p_drop; a.keep_scale = 1.0f / (1.0f - p_drop);
  const size_t lds_kv = (size_t)(2 * 96 * kStr + 2 * 192) * sizeof(float), lds_q = (size_t)2 * 96 * kStr * sizeof(float);
  static bool attr = false;
  if (!attr) {
    PE_CHECK_HIP((hipError_t)set_lds(&attn_bwd_kv_kernel<192, BF>, lds_kv));
    PE_CHECK_HIP((hipError_t)set_lds(&attn_bwd_q_kernel<192, BF>, lds_q));
    attr = true;
  }
  hipStream_t st = pe_stream(stream);
  hipLaunchKernelGGL((attn_bwd_kv_kernel<192, BF>), dim3(B * H * 3), dim3(256), lds_kv, st, a);
  PE_LAUNCH_CHECK();
  hipLaunchKernelGGL((attn_bwd_q_kernel<192, BF>), dim3(B * H * 3), dim3(256), lds_q, st, a);
  PE_LAUNCH_CHECK();
  return PE_OK;
}

extern "C" int pe_attn_bwd(const float* qkv, long ld_qkv, const float* o, const float* d_o, long ld_o, const float* lse,
                           const unsigned char* mask, float* dqkv, int B, int T, int H, int dh, float scale,
                           float p_drop, void* stream) {
  return attn_bwd_impl<false>(qkv, ld_qkv, o, d_o, ld_o, lse, mask, dqkv, B, T, H, dh, scale, p_drop, stream);
}
extern "C" int pe_attn_bwd_bf16(const float* qkv, long ld_qkv, const float* o, const float* d_o, long ld_o,
                                const float* lse, const unsigned char* mask, float* dqkv, int B, int T, int H, int dh,
                                float scale, float p_drop, void* stream) {
  return attn_bwd_impl<true>(qkv, ld_qkv, o, d_o, ld_o, lse, mask, dqkv, B, T, H, dh, scale, p_drop, stream);
}
