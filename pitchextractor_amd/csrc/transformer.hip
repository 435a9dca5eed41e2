// Transformer temporal head (model.py:178-193,229-241,253-255: sinusoidal PE + LayerNorm, then
// post-norm nn.TransformerEncoderLayer x N with exact-erf GELU) -- the pieces that are not plain
// GEMMs: batched per-(batch, head) attention products on the fp32 MFMA engine, row softmax,
// LayerNorm (+ residual / positional-encoding add) and GELU, each with its backward.
//
// Attention at T = 192, head_dim = 64 is done unfused: S = Q K^T (NT), P = softmax(S / 8),
// O = P V (NN); backward dV = P^T dO, dK = dS^T Q (TN), dP = dO V^T (NT), dQ = dS K (NN).  Heads are
// addressed in place inside the packed [B*T][3*512] projection (row stride 1536, head offset 64 h)
// and O lands directly in merged-head layout, so no head split/merge copies exist.
#include "gemm_engine.h"

namespace {
using namespace pe;

struct BatchView {          // matrix b of a two-level batch: base + (b / inner) * s_outer + (b % inner) * s_inner
  float* base;
  long ld, s_outer, s_inner;
  __device__ __forceinline__ float* at(int b, int inner) const {
    return base + (long)(b / inner) * s_outer + (long)(b % inner) * s_inner;
  }
};

struct BgemmArgs {
  BatchView A, B, C;
  int inner, M, N, K, accumulate;
  float alpha;
};

__device__ __forceinline__ void bgemm_store(const BgemmArgs& g, float* C, int m0, int n0, const f32x16& acc) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int wm = wv >> 1, wn = wv & 1, r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int row = m0 + wm * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
    const int col = n0 + wn * 32 + r;
    if (row < g.M && col < g.N) {
      float v = acc[q] * g.alpha;
      float* d = C + (long)row * g.C.ld + col;
      if (g.accumulate) v += *d;
      *d = v;
    }
  }
}

// mode 0: C = A B^T (A [M][K], B [N][K]);  1: C = A B (B [K][N]);  2: C = A^T B (A [K][M], B [K][N])
template <int MODE>
__global__ __launch_bounds__(256) void bgemm64_kernel(const BgemmArgs g, int tiles_n) {
  __shared__ __attribute__((aligned(16))) float As[64 * kLdsStride];
  __shared__ __attribute__((aligned(16))) float Bs[64 * kLdsStride];
  const int b = blockIdx.y;
  const int m0 = (blockIdx.x / tiles_n) * 64, n0 = (blockIdx.x % tiles_n) * 64;
  const float* A = g.A.at(b, g.inner);
  const float* B = g.B.at(b, g.inner);
  float* C = g.C.at(b, g.inner);
  f32x16 acc;
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.f;
  if (MODE == 0) {
    RowLoader al{A, g.A.ld, g.M, g.K, 0};
    RowLoader bl{B, g.B.ld, g.N, g.K, 0};
    al.init(m0);
    bl.init(n0);
    f32x16 a2[1][1];
    a2[0][0] = acc;
    nt_mainloop<Tile<64, 64, 2, 2>>(al, bl, g.K, As, Bs, a2);
    acc = a2[0][0];
  } else if (MODE == 1) {
    RowLoader al{A, g.A.ld, g.M, g.K, 0};
    KRowLoader<64> bl{B, g.B.ld, g.N, 0};
    al.init(m0);
    bl.init(n0, 0);
    nn_mainloop_64(al, bl, g.K, As, Bs, acc);
  } else {
    KRowLoader<64> al{A, g.A.ld, g.M, 0};
    KRowLoader<64> bl{B, g.B.ld, g.N, 0};
    al.init(m0, 0);
    bl.init(n0, 0);
    f32x16 a2[1][1];
    a2[0][0] = acc;
    tn_mainloop<64, 64>(al, bl, 0, g.K, As, Bs, a2);
    acc = a2[0][0];
  }
  bgemm_store(g, C, m0, n0, acc);
}

// ------------------------------------------------------------------ row softmax (one wave per row)
constexpr int kMaxPerLane = 16;      // rows up to 1024 wide

__global__ __launch_bounds__(256) void softmax_fwd_kernel(float* __restrict__ s, long rows, int L, float scale) {
  const int lane = threadIdx.x & 63;
  const long row = ((long)blockIdx.x * 256 + threadIdx.x) >> 6;
  if (row >= rows) return;
  float* p = s + row * L;
  float v[kMaxPerLane];
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < kMaxPerLane; ++i) {
    const int c = lane + 64 * i;
    v[i] = c < L ? p[c] * scale : -INFINITY;
    mx = fmaxf(mx, v[i]);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < kMaxPerLane; ++i) {
    v[i] = (lane + 64 * i) < L ? expf(v[i] - mx) : 0.f;
    sum += v[i];
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
  const float inv = 1.0f / sum;
#pragma unroll
  for (int i = 0; i < kMaxPerLane; ++i) {
    const int c = lane + 64 * i;
    if (c < L) p[c] = v[i] * inv;
  }
}

// ds = scale * p * (dp - sum_j dp_j p_j), written over dp
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* __restrict__ p, float* __restrict__ dp,
                                                          long rows, int L, float scale) {
  const int lane = threadIdx.x & 63;
  const long row = ((long)blockIdx.x * 256 + threadIdx.x) >> 6;
  if (row >= rows) return;
  const float* pr = p + row * L;
  float* dr = dp + row * L;
  float pv[kMaxPerLane], dv[kMaxPerLane];
  float dot = 0.f;
#pragma unroll
  for (int i = 0; i < kMaxPerLane; ++i) {
    const int c = lane + 64 * i;
    pv[i] = c < L ? pr[c] : 0.f;
    dv[i] = c < L ? dr[c] : 0.f;
    dot = fmaf(pv[i], dv[i], dot);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) dot += __shfl_xor(dot, off, 64);
#pragma unroll
  for (int i = 0; i < kMaxPerLane; ++i) {
    const int c = lane + 64 * i;
    if (c < L) dr[c] = scale * pv[i] * (dv[i] - dot);
  }
}

// ------------------------------------------------------------------ dropout folded into a neighbouring pass
// Same element <-> Philox counter mapping, keep rule and scaling as dropout_fwd_kernel (elementwise.hip) on a dense
// [rows][cols] tensor: quad q = (row * cols + col) / 4 draws philox4(seed, offset + q); keep = u >= p; kept values
// are multiplied by 1 / (1 - p).  mask_in replays given bytes; mask_out (optional) records them for the backward.
struct DropArgs {
  const uint8_t* mask_in;
  uint8_t* mask_out;
  float p, scale;
  uint64_t seed, offset;
};

__device__ __forceinline__ float4 drop_quad(const DropArgs& d, long quad, const float4& v) {
  // no mul+add contraction: the fused and the plain instances must round alike (results are compared bit for bit)
#pragma clang fp contract(off)
  uint8_t keep[4];
  if (d.mask_in) {
    const uchar4 m = *reinterpret_cast<const uchar4*>(d.mask_in + quad * 4);
    keep[0] = m.x; keep[1] = m.y; keep[2] = m.z; keep[3] = m.w;
  } else {
    uint32_t rnd[4];
    philox4(d.seed, d.offset + (uint64_t)quad, rnd);
#pragma unroll
    for (int k = 0; k < 4; ++k) keep[k] = ((float)(rnd[k] >> 8) * (1.0f / 16777216.0f)) >= d.p ? 1 : 0;
  }
  if (d.mask_out) *reinterpret_cast<uchar4*>(d.mask_out + quad * 4) = make_uchar4(keep[0], keep[1], keep[2], keep[3]);
  return make_float4(keep[0] ? v.x * d.scale : 0.f, keep[1] ? v.y * d.scale : 0.f, keep[2] ? v.z * d.scale : 0.f,
                     keep[3] ? v.w * d.scale : 0.f);
}

// ------------------------------------------------------------------ LayerNorm (one wave per row, D = 256 * NV)
template <int NV, bool DROP>      // DROP: z = a + dropout(b) (+ pe), the post-norm residual of an encoder layer
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                            const float* __restrict__ pe, int period,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float eps,
                                                            float* __restrict__ z_out, float* __restrict__ y,
                                                            float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                            long rows, DropArgs drop) {
#pragma clang fp contract(off)
  constexpr int D = 256 * NV;
  const int lane = threadIdx.x & 63;
  const long row = ((long)blockIdx.x * 256 + threadIdx.x) >> 6;
  if (row >= rows) return;
  float4 v[NV];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane * 4 + 256 * i;
    v[i] = *reinterpret_cast<const float4*>(a + row * D + c);
    if (b) {
      float4 t = *reinterpret_cast<const float4*>(b + row * D + c);
      if constexpr (DROP) t = drop_quad(drop, (row * D + c) >> 2, t);
      v[i].x += t.x; v[i].y += t.y; v[i].z += t.z; v[i].w += t.w;
    }
    if (pe) {
      const float4 t = *reinterpret_cast<const float4*>(pe + (row % period) * D + c);
      v[i].x += t.x; v[i].y += t.y; v[i].z += t.z; v[i].w += t.w;
    }
    if (z_out) *reinterpret_cast<float4*>(z_out + row * D + c) = v[i];
    sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
  const float mean = sum / (float)D;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const float dx = v[i].x - mean, dy = v[i].y - mean, dz = v[i].z - mean, dw = v[i].w - mean;
    sq += (dx * dx + dy * dy) + (dz * dz + dw * dw);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) sq += __shfl_xor(sq, off, 64);
  const float rstd = 1.0f / sqrtf(sq / (float)D + eps);
  if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane * 4 + 256 * i;
    const float4 g = *reinterpret_cast<const float4*>(gamma + c);
    const float4 bb = *reinterpret_cast<const float4*>(beta + c);
    float4 o;
    o.x = (v[i].x - mean) * rstd * g.x + bb.x;
    o.y = (v[i].y - mean) * rstd * g.y + bb.y;
    o.z = (v[i].z - mean) * rstd * g.z + bb.z;
    o.w = (v[i].w - mean) * rstd * g.w + bb.w;
    *reinterpret_cast<float4*>(y + row * D + c) = o;
  }
}

// dz = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma;  per-wave partial sums of
// dy * xhat (dgamma) and dy (dbeta) go to partial[wave][2][D]
// FUSED: dy = dy + dy2 (the residual branch's gradient, when dy2 != NULL) and a second output dz_drop = dropout_bwd(dz)
// (when drop_mask != NULL): the two passes an encoder layer's backward otherwise runs around this one
template <int NV, bool FUSED>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ dy2,
                                                            const float* __restrict__ z,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma, float* __restrict__ dz,
                                                            const uint8_t* __restrict__ drop_mask, float drop_scale,
                                                            float* __restrict__ dz_drop,
                                                            float* __restrict__ partial, long rows) {
#pragma clang fp contract(off)
  constexpr int D = 256 * NV;
  const int lane = threadIdx.x & 63;
  const long wave = ((long)blockIdx.x * 256 + threadIdx.x) >> 6;
  const long nwaves = ((long)gridDim.x * 256) >> 6;
  float4 ag[NV], ab[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) { ag[i] = make_float4(0, 0, 0, 0); ab[i] = make_float4(0, 0, 0, 0); }
  for (long row = wave; row < rows; row += nwaves) {
    const float mu = mean[row], rs = rstd[row];
    float4 xh[NV], g[NV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane * 4 + 256 * i;
      float4 d = *reinterpret_cast<const float4*>(dy + row * D + c);
      if constexpr (FUSED) {
        if (dy2) {
          const float4 e = *reinterpret_cast<const float4*>(dy2 + row * D + c);
          d.x += e.x; d.y += e.y; d.z += e.z; d.w += e.w;
        }
      }
      const float4 zz = *reinterpret_cast<const float4*>(z + row * D + c);
      const float4 gm = *reinterpret_cast<const float4*>(gamma + c);
      xh[i] = make_float4((zz.x - mu) * rs, (zz.y - mu) * rs, (zz.z - mu) * rs, (zz.w - mu) * rs);
      g[i] = make_float4(d.x * gm.x, d.y * gm.y, d.z * gm.z, d.w * gm.w);
      s1 += (g[i].x + g[i].y) + (g[i].z + g[i].w);
      s2 += (g[i].x * xh[i].x + g[i].y * xh[i].y) + (g[i].z * xh[i].z + g[i].w * xh[i].w);
      ag[i].x += d.x * xh[i].x; ag[i].y += d.y * xh[i].y; ag[i].z += d.z * xh[i].z; ag[i].w += d.w * xh[i].w;
      ab[i].x += d.x; ab[i].y += d.y; ab[i].z += d.z; ab[i].w += d.w;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      s1 += __shfl_xor(s1, off, 64);
      s2 += __shfl_xor(s2, off, 64);
    }
    const float m1 = s1 / (float)D, m2 = s2 / (float)D;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane * 4 + 256 * i;
      float4 o;
      o.x = rs * (g[i].x - m1 - xh[i].x * m2);
      o.y = rs * (g[i].y - m1 - xh[i].y * m2);
      o.z = rs * (g[i].z - m1 - xh[i].z * m2);
      o.w = rs * (g[i].w - m1 - xh[i].w * m2);
      *reinterpret_cast<float4*>(dz + row * D + c) = o;
      if constexpr (FUSED) {
        if (drop_mask) {
          const uchar4 m = *reinterpret_cast<const uchar4*>(drop_mask + row * D + c);
          *reinterpret_cast<float4*>(dz_drop + row * D + c) =
              make_float4(m.x ? o.x * drop_scale : 0.f, m.y ? o.y * drop_scale : 0.f, m.z ? o.z * drop_scale : 0.f,
                          m.w ? o.w * drop_scale : 0.f);
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane * 4 + 256 * i;
    *reinterpret_cast<float4*>(partial + (wave * 2 + 0) * D + c) = ag[i];
    *reinterpret_cast<float4*>(partial + (wave * 2 + 1) * D + c) = ab[i];
  }
}

__global__ void layernorm_bwd_final_kernel(const float* __restrict__ partial, int nparts, int D,
                                           float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= 2 * D) return;
  const int which = c / D, col = c % D;
  const int lane = threadIdx.x & 63;
  double s = 0.0;
  for (int zz = lane; zz < nparts; zz += 64) s += (double)partial[((long)zz * 2 + which) * D + col];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if (lane == 0) (which == 0 ? dgamma : dbeta)[col] = (float)s;
}

// two workgroups per CU: the fused pass (five streams in, two out) ran at 3.0 TB/s with one, 4.3 with two (176 -> 123
// us at 49152 x 512); the plain pass is unchanged (68 us)
constexpr int kLnBwdBlocks = 512;     // -> 2048 waves of partials

// ------------------------------------------------------------------ exact (erf) GELU
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long n4) {
#pragma clang fp contract(off)
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    float4 o;
    o.x = 0.5f * v.x * (1.0f + erff(v.x * 0.70710678118654752f));
    o.y = 0.5f * v.y * (1.0f + erff(v.y * 0.70710678118654752f));
    o.z = 0.5f * v.z * (1.0f + erff(v.z * 0.70710678118654752f));
    o.w = 0.5f * v.w * (1.0f + erff(v.w * 0.70710678118654752f));
    reinterpret_cast<float4*>(y)[i] = o;
  }
}

__device__ __forceinline__ float gelu_grad(float x) {
#pragma clang fp contract(off)
  return 0.5f * (1.0f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * expf(-0.5f * x * x);
}

__global__ __launch_bounds__(256) void gelu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                       float* __restrict__ dx, long n4) {
#pragma clang fp contract(off)
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    const float4 d = reinterpret_cast<const float4*>(dy)[i];
    reinterpret_cast<float4*>(dx)[i] =
        make_float4(d.x * gelu_grad(v.x), d.y * gelu_grad(v.y), d.z * gelu_grad(v.z), d.w * gelu_grad(v.w));
  }
}

// a = dropout(gelu(h)) in one pass (linear1 -> activation -> dropout of an encoder layer), and its backward
// dh = dropout_bwd(da) * gelu'(h); dense [rows][cols], the quad index is the flat float4 index
__global__ __launch_bounds__(256) void gelu_dropout_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                               long n4, DropArgs drop) {
#pragma clang fp contract(off)
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    float4 o;
    o.x = 0.5f * v.x * (1.0f + erff(v.x * 0.70710678118654752f));
    o.y = 0.5f * v.y * (1.0f + erff(v.y * 0.70710678118654752f));
    o.z = 0.5f * v.z * (1.0f + erff(v.z * 0.70710678118654752f));
    o.w = 0.5f * v.w * (1.0f + erff(v.w * 0.70710678118654752f));
    reinterpret_cast<float4*>(y)[i] = drop_quad(drop, i, o);
  }
}

__global__ __launch_bounds__(256) void gelu_dropout_bwd_kernel(const float* __restrict__ x,
                                                               const float* __restrict__ dy,
                                                               const uint8_t* __restrict__ mask, float scale,
                                                               float* __restrict__ dx, long n4) {
#pragma clang fp contract(off)
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    float4 d = reinterpret_cast<const float4*>(dy)[i];
    const uchar4 m = reinterpret_cast<const uchar4*>(mask)[i];
    d = make_float4(m.x ? d.x * scale : 0.f, m.y ? d.y * scale : 0.f, m.z ? d.z * scale : 0.f,
                    m.w ? d.w * scale : 0.f);
    reinterpret_cast<float4*>(dx)[i] =
        make_float4(d.x * gelu_grad(v.x), d.y * gelu_grad(v.y), d.z * gelu_grad(v.z), d.w * gelu_grad(v.w));
  }
}

int ew_blocks(long n) {
  long g = (n + 255) / 256;
  return (int)(g > 16384 ? 16384 : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" int pe_bgemm(int mode, const float* A, long lda, long a_outer, long a_inner, const float* B, long ldb,
                        long b_outer, long b_inner, float* C, long ldc, long c_outer, long c_inner, int inner,
                        int batch, int M, int N, int K, float alpha, int accumulate, void* stream) {
  if (!A || !B || !C || mode < 0 || mode > 2 || inner <= 0 || batch < 0 || M <= 0 || N <= 0 || K <= 0)
    return PE_E_ARG;
  if (batch == 0) return PE_OK;
  if (batch > 65535) return PE_E_UNSUPPORTED;
  // every float4 the loaders issue must be 16-byte aligned and in range
  const bool a_kc = (mode != 2), b_kc = (mode == 0);
  if ((lda & 3) || (ldb & 3) || (a_outer & 3) || (a_inner & 3) || (b_outer & 3) || (b_inner & 3)) return PE_E_UNSUPPORTED;
  if (((uintptr_t)A & 15) || ((uintptr_t)B & 15)) return PE_E_UNSUPPORTED;
  if ((a_kc && (K & 3)) || (!a_kc && (M & 3)) || (b_kc && (K & 3)) || (!b_kc && (N & 3))) return PE_E_UNSUPPORTED;
  BgemmArgs g;
  g.A = BatchView{const_cast<float*>(A), lda, a_outer, a_inner};
  g.B = BatchView{const_cast<float*>(B), ldb, b_outer, b_inner};
  g.C = BatchView{C, ldc, c_outer, c_inner};
  g.inner = inner; g.M = M; g.N = N; g.K = K; g.accumulate = accumulate; g.alpha = alpha;
  const int tm = pe_cdiv(M, 64), tn = pe_cdiv(N, 64);
  dim3 grid(tm * tn, batch);
  hipStream_t st = pe_stream(stream);
  if (mode == 0) hipLaunchKernelGGL(bgemm64_kernel<0>, grid, dim3(256), 0, st, g, tn);
  else if (mode == 1) hipLaunchKernelGGL(bgemm64_kernel<1>, grid, dim3(256), 0, st, g, tn);
  else hipLaunchKernelGGL(bgemm64_kernel<2>, grid, dim3(256), 0, st, g, tn);
  PE_LAUNCH_CHECK();
  return PE_OK;
}

extern "C" int pe_softmax_fwd(float* s, long rows, int L, float scale, void* stream) {
  if (!s || rows <= 0 || L <= 0) return PE_E_ARG;
  if (L > 64 * kMaxPerLane) return PE_E_UNSUPPORTED;
  hipLaunchKernelGGL(softmax_fwd_kernel, dim3(pe_cdiv(rows, 4)), dim3(256), 0, pe_stream(stream), s, rows, L, scale);
  PE_LAUNCH_CHECK();
  return PE_OK;
}

extern "C" int pe_softmax_bwd(const float* p, float* dp, long rows, int L, float scale, void* stream) {
  if (!p || !dp || rows <= 0 || L <= 0) return PE_E_ARG;
  if (L > 64 * kMaxPerLane) return PE_E_UNSUPPORTED;
  hipLaunchKernelGGL(softmax_bwd_kernel, dim3(pe_cdiv(rows, 4)), dim3(256), 0, pe_stream(stream), p, dp, rows, L,
                     scale);
  PE_LAUNCH_CHECK();
  return PE_OK;
}

namespace {
template <bool DROP>
int launch_layernorm_fwd(const float* a, const float* b, const float* pe, int period, const float* gamma,
                         const float* beta, float eps, float* z_out, float* y, float* mean, float* rstd, long rows,
                         int D, const DropArgs& drop, hipStream_t st) {
  dim3 grid(pe_cdiv(rows, 4));
  if (D == 256) hipLaunchKernelGGL((layernorm_fwd_kernel<1, DROP>), grid, dim3(256), 0, st, a, b, pe, period, gamma, beta, eps, z_out, y, mean, rstd, rows, drop);
  else if (D == 512) hipLaunchKernelGGL((layernorm_fwd_kernel<2, DROP>), grid, dim3(256), 0, st, a, b, pe, period, gamma, beta, eps, z_out, y, mean, rstd, rows, drop);
  else if (D == 768) hipLaunchKernelGGL((layernorm_fwd_kernel<3, DROP>), grid, dim3(256), 0, st, a, b, pe, period, gamma, beta, eps, z_out, y, mean, rstd, rows, drop);
  else if (D == 1024) hipLaunchKernelGGL((layernorm_fwd_kernel<4, DROP>), grid, dim3(256), 0, st, a, b, pe, period, gamma, beta, eps, z_out, y, mean, rstd, rows, drop);
  else return PE_E_UNSUPPORTED;
  PE_LAUNCH_CHECK();
  return PE_OK;
}

template <bool FUSED>
int launch_layernorm_bwd(const float* dy, const float* dy2, const float* z, const float* mean, const float* rstd,
                         const float* gamma, float* dz, const uint8_t* mask, float scale, float* dz_drop,
                         float* dgamma, float* dbeta, long rows, int D, float* partial, hipStream_t st) {
  dim3 grid(kLnBwdBlocks);
  if (D == 256) hipLaunchKernelGGL((layernorm_bwd_kernel<1, FUSED>), grid, dim3(256), 0, st, dy, dy2, z, mean, rstd, gamma, dz, mask, scale, dz_drop, partial, rows);
  else if (D == 512) hipLaunchKernelGGL((layernorm_bwd_kernel<2, FUSED>), grid, dim3(256), 0, st, dy, dy2, z, mean, rstd, gamma, dz, mask, scale, dz_drop, partial, rows);
  else if (D == 768) hipLaunchKernelGGL((layernorm_bwd_kernel<3, FUSED>), grid, dim3(256), 0, st, dy, dy2, z, mean, rstd, gamma, dz, mask, scale, dz_drop, partial, rows);
  else if (D == 1024) hipLaunchKernelGGL((layernorm_bwd_kernel<4, FUSED>), grid, dim3(256), 0, st, dy, dy2, z, mean, rstd, gamma, dz, mask, scale, dz_drop, partial, rows);
  else return PE_E_UNSUPPORTED;
  PE_LAUNCH_CHECK();
  hipLaunchKernelGGL(layernorm_bwd_final_kernel, dim3(pe_cdiv(2 * D, 4)), dim3(256), 0, st, partial, kLnBwdBlocks * 4, D,
                     dgamma, dbeta);
  PE_LAUNCH_CHECK();
  return PE_OK;
}
}  // namespace

extern "C" int pe_layernorm_fwd(const float* a, const float* b, const float* pe, int period, const float* gamma,
                                const float* beta, float eps, float* z_out, float* y, float* mean, float* rstd,
                                long rows, int D, void* stream) {
  if (!a || !gamma || !beta || !y || !mean || !rstd || rows <= 0) return PE_E_ARG;
  if (pe && period <= 0) return PE_E_ARG;
  return launch_layernorm_fwd<false>(a, b, pe, period, gamma, beta, eps, z_out, y, mean, rstd, rows, D, DropArgs{},
                                     pe_stream(stream));
}

extern "C" int pe_layernorm_dropout_fwd(const float* a, const float* b, const float* pe, int period,
                                        const float* gamma, const float* beta, float eps, float* z_out, float* y,
                                        float* mean, float* rstd, long rows, int D, const unsigned char* mask_in,
                                        unsigned char* mask_out, float p, unsigned long long seed,
                                        unsigned long long offset, void* stream) {
  if (!a || !b || !gamma || !beta || !y || !mean || !rstd || rows <= 0 || p <= 0.f || p >= 1.f) return PE_E_ARG;
  if (pe && period <= 0) return PE_E_ARG;
  const DropArgs drop{mask_in, mask_out, p, 1.0f / (1.0f - p), (uint64_t)seed, (uint64_t)offset};
  return launch_layernorm_fwd<true>(a, b, pe, period, gamma, beta, eps, z_out, y, mean, rstd, rows, D, drop,
                                    pe_stream(stream));
}

extern "C" size_t pe_layernorm_bwd_workspace_bytes(int D) { return (size_t)kLnBwdBlocks * 4 * 2 * D * sizeof(float); }

extern "C" int pe_layernorm_bwd(const float* dy, const float* z, const float* mean, const float* rstd,
                                const float* gamma, float* dz, float* dgamma, float* dbeta, long rows, int D,
                                void* workspace, size_t workspace_bytes, void* stream) {
  if (!dy || !z || !mean || !rstd || !gamma || !dz || !dgamma || !dbeta || rows <= 0) return PE_E_ARG;
  if (!workspace || workspace_bytes < pe_layernorm_bwd_workspace_bytes(D)) return PE_E_WORKSPACE;
  return launch_layernorm_bwd<false>(dy, nullptr, z, mean, rstd, gamma, dz, nullptr, 1.f, nullptr, dgamma, dbeta, rows,
                                     D, reinterpret_cast<float*>(workspace), pe_stream(stream));
}

extern "C" int pe_layernorm_bwd_fused(const float* dy, const float* dy2, const float* z, const float* mean,
                                      const float* rstd, const float* gamma, float* dz,
                                      const unsigned char* drop_mask, float p, float* dz_drop, float* dgamma,
                                      float* dbeta, long rows, int D, void* workspace, size_t workspace_bytes,
                                      void* stream) {
  if (!dy || !z || !mean || !rstd || !gamma || !dz || !dgamma || !dbeta || rows <= 0) return PE_E_ARG;
  if ((drop_mask != nullptr) != (dz_drop != nullptr)) return PE_E_ARG;
  if (drop_mask && (p <= 0.f || p >= 1.f)) return PE_E_ARG;
  if (!workspace || workspace_bytes < pe_layernorm_bwd_workspace_bytes(D)) return PE_E_WORKSPACE;
  return launch_layernorm_bwd<true>(dy, dy2, z, mean, rstd, gamma, dz, drop_mask, drop_mask ? 1.0f / (1.0f - p) : 1.f,
                                    dz_drop, dgamma, dbeta, rows, D, reinterpret_cast<float*>(workspace),
                                    pe_stream(stream));
}

extern "C" int pe_gelu_fwd(const float* x, float* y, long n, void* stream) {
  if (!x || !y || n <= 0) return PE_E_ARG;
  if (n & 3) return PE_E_UNSUPPORTED;
  hipLaunchKernelGGL(gelu_fwd_kernel, dim3(ew_blocks(n / 4)), dim3(256), 0, pe_stream(stream), x, y, n / 4);
  PE_LAUNCH_CHECK();
  return PE_OK;
}

extern "C" int pe_gelu_bwd(const float* x, const float* dy, float* dx, long n, void* stream) {
  if (!x || !dy || !dx || n <= 0) return PE_E_ARG;
  if (n & 3) return PE_E_UNSUPPORTED;
  hipLaunchKernelGGL(gelu_bwd_kernel, dim3(ew_blocks(n / 4)), dim3(256), 0, pe_stream(stream), x, dy, dx, n / 4);
  PE_LAUNCH_CHECK();
  return PE_OK;
}

extern "C" int pe_gelu_dropout_fwd(const float* x, float* y, long n, const unsigned char* mask_in,
                                   unsigned char* mask_out, float p, unsigned long long seed,
                                   unsigned long long offset, void* stream) {
  if (!x || !y || n <= 0 || p <= 0.f || p >= 1.f) return PE_E_ARG;
  if (n & 3) return PE_E_UNSUPPORTED;
  const DropArgs drop{mask_in, mask_out, p, 1.0f / (1.0f - p), (uint64_t)seed, (uint64_t)offset};
  hipLaunchKernelGGL(gelu_dropout_fwd_kernel, dim3(ew_blocks(n / 4)), dim3(256), 0, pe_stream(stream), x, y, n / 4,
                     drop);
  PE_LAUNCH_CHECK();
  return PE_OK;
}

extern "C" int pe_gelu_dropout_bwd(const float* x, const float* dy, const unsigned char* mask, float p, float* dx,
                                   long n, void* stream) {
  if (!x || !dy || !mask || !dx || n <= 0 || p <= 0.f || p >= 1.f) return PE_E_ARG;
  if (n & 3) return PE_E_UNSUPPORTED;
  hipLaunchKernelGGL(gelu_dropout_bwd_kernel, dim3(ew_blocks(n / 4)), dim3(256), 0, pe_stream(stream), x, dy, mask,
                     1.0f / (1.0f - p), dx, n / 4);
  PE_LAUNCH_CHECK();
  return PE_OK;
}
