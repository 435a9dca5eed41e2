// Bidirectional multi-layer LSTM recurrence (nn.LSTM as built at model.py:218-227: gate order
// i, f, g, o; zero initial state; batch_first) on the fp32 MFMA engine.
//
// The input projections X.W_ih^T + b_ih + b_hh for all T steps are one big GEMM done by the
// caller (pe_gemm_nt); what is strictly sequential is  gates_t += h_{t-1}.W_hh^T  followed by the
// cell update, T times per layer.  One launch per time step covers up to 4 independent
// "cells" (2 directions x classifier/detector models) so the chip sees 192 workgroups per step:
//
//   forward step:  workgroup = 64 batch rows x 32 hidden units x all 4 gates (a 64x128 MFMA tile
//                  whose 128 columns are rows {g*H + j0 .. j0+31} of W_hh), gates exchanged through
//                  LDS so the cell update runs in the same launch; activated gates overwrite the
//                  projected ones in place (they are what backward needs).
//   backward step: dh_t = dY_t + dgates_{t+1}.W_hh  (K = 4H, split 4 ways over the waves of the
//                  workgroup and reduced in LDS), then the gate derivatives overwrite the
//                  activations in place, leaving dgates for the batched weight-gradient GEMMs.
#include "gemm_engine.h"

namespace {
using namespace pe;

constexpr int kMaxCells = 4;

struct FwdCells {
  const float* whh[kMaxCells];     // [4H][H]
  float* gates[kMaxCells];         // [B][T][4H]  in: x-projection (+biases), out: activated i,f,g,o
  float* y[kMaxCells];             // element (b,t,j) at y[(b*T+t)*ldy + j]  (already offset by dir*H)
  float* c[kMaxCells];             // [B][T][H]
  int reverse[kMaxCells];
};

struct GateRowLoader {             // B operand: rows {g*H + j0 + (0..31)}, g = slot
  const float* p;
  int H, j0;
  __device__ __forceinline__ void init(int) {}
  __device__ __forceinline__ float4 load(int slot, int kt) const {
    const int row = slot * H + j0 + (threadIdx.x >> 3);
    const int k = kt * kBK + (threadIdx.x & 7) * 4;
    if (k < H) return *reinterpret_cast<const float4*>(p + (long)row * H + k);
    return make_float4(0.f, 0.f, 0.f, 0.f);
  }
};

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

using FwdTile = Tile<64, 128, 2, 2>;
constexpr int kGs = 132;           // padded row stride of the gate exchange tile

__global__ __launch_bounds__(256) void lstm_fwd_step_kernel(const FwdCells cells, int step, int B, int T, int H,
                                                            long ldy) {
  __shared__ __attribute__((aligned(16))) float As[64 * kLdsStride];
  __shared__ __attribute__((aligned(16))) float Bs[128 * kLdsStride];
  __shared__ __attribute__((aligned(16))) float Gs[64 * kGs];
  const int cell = blockIdx.z;
  const int b0 = blockIdx.x * 64, j0 = blockIdx.y * 32;
  const int rev = cells.reverse[cell];
  const int t = rev ? T - 1 - step : step;
  const int tp = rev ? t + 1 : t - 1;
  float* y = cells.y[cell];

  f32x16 acc[FwdTile::TM][FwdTile::TN];
  zero_acc<FwdTile>(acc);
  if (step > 0) {
    RowLoader al{y + (long)tp * ldy, (long)T * ldy, B, H, 0};
    GateRowLoader bl{cells.whh[cell], H, j0};
    al.init(b0);
    nt_mainloop<FwdTile>(al, bl, H, As, Bs, acc);
  }
  for_each_acc<FwdTile>(acc, [&](int r, int c, float v) { Gs[r * kGs + c] = v; });
  __syncthreads();

  float* gates = cells.gates[cell];
  float* cb = cells.c[cell];
  const int jj = threadIdx.x & 31;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int bl_ = (threadIdx.x >> 5) + 8 * i;
    const int b = b0 + bl_;
    if (b < B) {
      const long row = (long)b * T + t;
      float* gp = gates + row * 4 * H + j0 + jj;
      const float gi = sigmoidf_(Gs[bl_ * kGs + jj] + gp[0]);
      const float gf = sigmoidf_(Gs[bl_ * kGs + 32 + jj] + gp[H]);
      const float gg = tanhf(Gs[bl_ * kGs + 64 + jj] + gp[2 * H]);
      const float go = sigmoidf_(Gs[bl_ * kGs + 96 + jj] + gp[3 * H]);
      const float cprev = step > 0 ? cb[((long)b * T + tp) * H + j0 + jj] : 0.f;
      const float cn = gf * cprev + gi * gg;
      gp[0] = gi; gp[H] = gf; gp[2 * H] = gg; gp[3 * H] = go;
      cb[row * H + j0 + jj] = cn;
      y[row * ldy + j0 + jj] = go * tanhf(cn);
    }
  }
}

// ------------------------------------------------------------------ backward
struct BwdCells {
  const float* whh_t[kMaxCells];   // W_hh^T: [H][4H]
  float* gates[kMaxCells];         // in: activated gates, out: d(pre-activation gates)
  const float* c[kMaxCells];       // [B][T][H]
  const float* dy[kMaxCells];      // element (b,t,j) at dy[(b*T+t)*lddy + j]
  float* dcarry[kMaxCells];        // [B][H] running dc, zero before the first step
  int reverse[kMaxCells];
};

// 64 x 32 output tile, K split four ways across the waves (wave w takes the w-th 8-wide k block of
// every 32-wide stage); partial tiles are summed through LDS.
__global__ __launch_bounds__(256) void lstm_bwd_step_kernel(const BwdCells cells, int step, int B, int T, int H,
                                                            long lddy) {
  __shared__ __attribute__((aligned(16))) float As[64 * kLdsStride];
  __shared__ __attribute__((aligned(16))) float Bs[32 * kLdsStride];
  __shared__ __attribute__((aligned(16))) float Rs[4 * 64 * 33];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int cell = blockIdx.z;
  const int b0 = blockIdx.x * 64, j0 = blockIdx.y * 32;
  const int rev = cells.reverse[cell];
  const int t = rev ? step : T - 1 - step;          // walk the forward order backwards
  const int tn = rev ? t - 1 : t + 1;               // the step that consumed h_t
  const int tp = rev ? t + 1 : t - 1;               // the step that produced c_{prev}
  const int K = 4 * H;
  float* gates = cells.gates[cell];

  f32x16 acc[2];
#pragma unroll
  for (int g = 0; g < 16; ++g) { acc[0][g] = 0.f; acc[1][g] = 0.f; }
  if (step > 0) {
    RowLoader al{gates + (long)tn * K, (long)T * K, B, K, 0};
    RowLoader bl{cells.whh_t[cell], (long)K, H, K, 0};
    al.init(b0);
    bl.init(j0);
    const int r = lane & 31, h = lane >> 5;
    const int nk = K / kBK;
    float4 ra[2], rb;
    ra[0] = al.load(0, 0); ra[1] = al.load(1, 0); rb = bl.load(0, 0);
    const int st_off = (tid >> 3) * kLdsStride + (tid & 7) * 4;
    const float* a_rd = As + r * kLdsStride + wv * 8 + h * 4;
    const float* b_rd = Bs + r * kLdsStride + wv * 8 + h * 4;
    for (int kt = 0; kt < nk; ++kt) {
      __syncthreads();
      *reinterpret_cast<float4*>(As + st_off) = ra[0];
      *reinterpret_cast<float4*>(As + st_off + 32 * kLdsStride) = ra[1];
      *reinterpret_cast<float4*>(Bs + st_off) = rb;
      __syncthreads();
      if (kt + 1 < nk) { ra[0] = al.load(0, kt + 1); ra[1] = al.load(1, kt + 1); rb = bl.load(0, kt + 1); }
      const float4 fa0 = *reinterpret_cast<const float4*>(a_rd);
      const float4 fa1 = *reinterpret_cast<const float4*>(a_rd + 32 * kLdsStride);
      const float4 fb = *reinterpret_cast<const float4*>(b_rd);
      acc[0] = mfma32(fa0.x, fb.x, acc[0]); acc[1] = mfma32(fa1.x, fb.x, acc[1]);
      acc[0] = mfma32(fa0.y, fb.y, acc[0]); acc[1] = mfma32(fa1.y, fb.y, acc[1]);
      acc[0] = mfma32(fa0.z, fb.z, acc[0]); acc[1] = mfma32(fa1.z, fb.z, acc[1]);
      acc[0] = mfma32(fa0.w, fb.w, acc[0]); acc[1] = mfma32(fa1.w, fb.w, acc[1]);
    }
  }
  {
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const int row = i * 32 + (g & 3) + 8 * (g >> 2) + 4 * h;
        Rs[(wv * 64 + row) * 33 + r] = acc[i][g];
      }
  }
  __syncthreads();

  const float* cb = cells.c[cell];
  const float* dy = cells.dy[cell];
  float* dcar = cells.dcarry[cell];
  const int jj = tid & 31;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int bl_ = (tid >> 5) + 8 * i;
    const int b = b0 + bl_;
    if (b < B) {
      const long row = (long)b * T + t;
      const int j = j0 + jj;
      float dh = dy[row * lddy + j];
      dh += (Rs[(0 * 64 + bl_) * 33 + jj] + Rs[(1 * 64 + bl_) * 33 + jj]) +
            (Rs[(2 * 64 + bl_) * 33 + jj] + Rs[(3 * 64 + bl_) * 33 + jj]);
      float* gp = gates + row * K + j;
      const float gi = gp[0], gf = gp[H], gg = gp[2 * H], go = gp[3 * H];
      const float cn = cb[row * H + j];
      const bool has_prev = rev ? (tp < T) : (tp >= 0);
      const float cprev = has_prev ? cb[((long)b * T + tp) * H + j] : 0.f;
      const float tc = tanhf(cn);
      const float dcar_in = step > 0 ? dcar[(long)b * H + j] : 0.f;
      const float dc = dh * go * (1.f - tc * tc) + dcar_in;
      gp[0] = dc * gg * gi * (1.f - gi);
      gp[H] = dc * cprev * gf * (1.f - gf);
      gp[2 * H] = dc * gi * (1.f - gg * gg);
      gp[3 * H] = dh * tc * go * (1.f - go);
      dcar[(long)b * H + j] = dc * gf;
    }
  }
}

// ------------------------------------------------------------------ dW_hh = sum_t dgates_t^T h_{t-1}
template <int BM, int BN, int MODE>
__global__ __launch_bounds__(256) void lstm_whh_grad_kernel(KRowLoader<BM> al, ShiftedTimeLoader<BN> bl, float* out,
                                                            long ldo, long split_stride, int M, int N, int K,
                                                            int k_per_split, int tiles_n, const unsigned* amax_a,
                                                            const unsigned* amax_b) {
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
    if (row < M && col < N) dst[(long)row * ldo + col] = MODE == kSplit2 ? v * hs.inv : v;
  });
}

__global__ void slab_reduce_kernel(const float* ws, long n, int splits, float* out) {   // n % 4 == 0
  const long i4 = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i4 * 4 >= n) return;
  const float4 s = pe_ordered_slab_sum4(ws, n, splits, i4);
  float* d = out + 4 * i4;                             // `out` may be an unaligned view of the flat gradient buffer
  d[0] = s.x; d[1] = s.y; d[2] = s.z; d[3] = s.w;
}

void whh_plan(int M, int N, int K, int mode, int* splits, int* kps) {
  const int tiles = pe_cdiv(M, 128) * pe_cdiv(N, 128);
  const int s = pe_pick_splits(tiles, K, 512, (mode == kSplit || mode == kSplit2) ? 512 : 768);
  int k = pe_cdiv(K, s);
  k = (k + kBK - 1) / kBK * kBK;
  *kps = k;
  *splits = pe_cdiv(K, k);
}

// ------------------------------------------------------------------ column sums (bias gradients)
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ x, long rows, int cols,
                                                             long ld, double* __restrict__ partial) {
  // grid.x over column blocks of 256 (64 float4 quads), grid.y over row chunks; the block's 4 row lanes take
  // every 4th row of the chunk with float4 loads, four rows in flight per lane (cols % 4 == 0, ld % 4 == 0)
  __shared__ double red[4][256];
  const int q = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int c = blockIdx.x * 256 + q * 4;
  const long chunk = (rows + gridDim.y - 1) / gridDim.y;
  const long r0 = (long)blockIdx.y * chunk;
  const long r1 = r0 + chunk < rows ? r0 + chunk : rows;
  double s[4] = {0, 0, 0, 0};
  if (c < cols) {
    long r = r0 + ry;
    for (; r + 12 < r1; r += 16) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(x + (r + 4 * u) * ld + c);
#pragma unroll
      for (int u = 0; u < 4; ++u) { s[0] += v[u].x; s[1] += v[u].y; s[2] += v[u].z; s[3] += v[u].w; }
    }
    for (; r < r1; r += 4) {
      const float4 v = *reinterpret_cast<const float4*>(x + r * ld + c);
      s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) red[ry][q * 4 + e] = s[e];
  __syncthreads();
  const int cc = blockIdx.x * 256 + threadIdx.x;
  if (cc < cols)
    partial[(long)blockIdx.y * cols + cc] = (red[0][threadIdx.x] + red[1][threadIdx.x]) +
                                            (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

__global__ void colsum_final_kernel(const double* __restrict__ partial, int nparts, int cols, float* out0,
                                    float* out1) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);            // one wave per column
  if (c >= cols) return;
  const double s = pe_wave_strided_sum(partial + c, (long)cols, nparts);
  if ((threadIdx.x & 63) != 0) return;
  out0[c] = (float)s;
  if (out1) out1[c] = (float)s;
}

constexpr int kColsumParts = 256;

}  // namespace

#ifndef PE_F16_BUILD
extern "C" int pe_lstm_fwd(int ncells, const float* const* whh, float* const* gates, float* const* y,
                           float* const* cbuf, const int* reverse, long ldy, int B, int T, int H, void* stream) {
  if (ncells < 1 || ncells > kMaxCells || !whh || !gates || !y || !cbuf || !reverse) return PE_E_ARG;
  if (B <= 0 || T <= 0 || H <= 0) return PE_E_ARG;
  if ((H % 32) != 0 || (ldy & 3)) return PE_E_UNSUPPORTED;
  FwdCells cells{};
  for (int i = 0; i < ncells; ++i) {
    if (!whh[i] || !gates[i] || !y[i] || !cbuf[i]) return PE_E_ARG;
    cells.whh[i] = whh[i]; cells.gates[i] = gates[i]; cells.y[i] = y[i]; cells.c[i] = cbuf[i];
    cells.reverse[i] = reverse[i];
  }
  dim3 grid(pe_cdiv(B, 64), H / 32, ncells);
  hipStream_t st = pe_stream(stream);
  for (int s = 0; s < T; ++s) {
    hipLaunchKernelGGL(lstm_fwd_step_kernel, grid, dim3(256), 0, st, cells, s, B, T, H, ldy);
  }
  PE_LAUNCH_CHECK();
  return PE_OK;
}

extern "C" int pe_lstm_bwd(int ncells, const float* const* whh_t, float* const* gates, const float* const* cbuf,
                           const float* const* dy, float* const* dcarry, const int* reverse, long lddy, int B,
                           int T, int H, void* stream) {
  if (ncells < 1 || ncells > kMaxCells || !whh_t || !gates || !cbuf || !dy || !dcarry || !reverse) return PE_E_ARG;
  if (B <= 0 || T <= 0 || H <= 0) return PE_E_ARG;
  if ((H % 32) != 0 || (lddy & 3)) return PE_E_UNSUPPORTED;
  BwdCells cells{};
  for (int i = 0; i < ncells; ++i) {
    if (!whh_t[i] || !gates[i] || !cbuf[i] || !dy[i] || !dcarry[i]) return PE_E_ARG;
    cells.whh_t[i] = whh_t[i]; cells.gates[i] = gates[i]; cells.c[i] = cbuf[i]; cells.dy[i] = dy[i];
    cells.dcarry[i] = dcarry[i]; cells.reverse[i] = reverse[i];
  }
  dim3 grid(pe_cdiv(B, 64), H / 32, ncells);
  hipStream_t st = pe_stream(stream);
  for (int s = 0; s < T; ++s) {
    hipLaunchKernelGGL(lstm_bwd_step_kernel, grid, dim3(256), 0, st, cells, s, B, T, H, lddy);
  }
  PE_LAUNCH_CHECK();
  return PE_OK;
}

extern "C" size_t pe_lstm_whh_grad_workspace_bytes(int B, int T, int H) {
  size_t need = 0;
  for (int mode : {kNative, kSplit}) {
    int splits, kps;
    whh_plan(4 * H, H, B * T, mode, &splits, &kps);
    const size_t b = (size_t)splits * 4 * H * H * sizeof(float);
    need = b > need ? b : need;
  }
  return need;
}
#endif

// dW_hh[4H][H] = sum_{b,t} dgates[b][t][:]^T . y[b][t -/+ 1][:]   (y = this direction's output slice)
template <int MODE>
static int whh_grad_impl(const float* dgates, const float* y, long ldy, float* dwhh, int B, int T, int H,
                         int reverse, float* workspace, size_t workspace_bytes, void* stream,
                         const unsigned* amax_dg = nullptr, const unsigned* amax_y = nullptr) {
  if (!dgates || !y || !dwhh || B <= 0 || T <= 0 || H <= 0) return PE_E_ARG;
  if (MODE == kSplit2 && (!amax_dg || !amax_y)) return PE_E_ARG;
  if ((H & 3) || (ldy & 3)) return PE_E_UNSUPPORTED;
  const int M = 4 * H, N = H, K = B * T;
  int splits, kps;
  whh_plan(M, N, K, MODE, &splits, &kps);
  const size_t need = (size_t)splits * M * N * sizeof(float);
  if (!workspace || workspace_bytes < need) return PE_E_WORKSPACE;
  KRowLoader<128> al{dgates, (long)M, M, 0};
  ShiftedTimeLoader<128> bl;
  bl.p = y; bl.ld = ldy; bl.T = T; bl.dt = reverse ? 1 : -1; bl.cols = N; bl.col0 = 0;
  const int tm = pe_cdiv(M, 128), tn = pe_cdiv(N, 128);
  hipStream_t st = pe_stream(stream);
  hipLaunchKernelGGL((lstm_whh_grad_kernel<128, 128, MODE>), dim3(tm * tn * splits), dim3(256), 0, st, al, bl,
                     workspace, (long)N, (long)M * N, M, N, K, kps, tn, amax_dg, amax_y);
  PE_LAUNCH_CHECK();
  const long n = (long)M * N;
  hipLaunchKernelGGL(slab_reduce_kernel, dim3(pe_cdiv(n / 4, 256)), dim3(256), 0, st, workspace, n, splits, dwhh);
  PE_LAUNCH_CHECK();
  return PE_OK;
}

#ifndef PE_F16_BUILD
extern "C" int pe_lstm_whh_grad(const float* dgates, const float* y, long ldy, float* dwhh, int B, int T, int H,
                                int reverse, float* workspace, size_t workspace_bytes, void* stream) {
  return whh_grad_impl<kNative>(dgates, y, ldy, dwhh, B, T, H, reverse, workspace, workspace_bytes, stream);
}

extern "C" int pe_lstm_whh_grad_x3(const float* dgates, const float* y, long ldy, float* dwhh, int B, int T, int H,
                                   int reverse, float* workspace, size_t workspace_bytes, void* stream) {
  return whh_grad_impl<kSplit>(dgates, y, ldy, dwhh, B, T, H, reverse, workspace, workspace_bytes, stream);
}

extern "C" int pe_lstm_whh_grad_h2(const float* dgates, const float* y, long ldy, float* dwhh, int B, int T, int H,
                                   int reverse, float* workspace, size_t workspace_bytes, const unsigned* amax_dgates,
                                   const unsigned* amax_y, void* stream) {
  return whh_grad_impl<kSplit2>(dgates, y, ldy, dwhh, B, T, H, reverse, workspace, workspace_bytes, stream,
                                amax_dgates, amax_y);
}
#endif

extern "C" int PE_HALF(pe_lstm_whh_grad)(const float* dgates, const float* y, long ldy, float* dwhh, int B, int T, int H,
                                   int reverse, float* workspace, size_t workspace_bytes, void* stream) {
  return whh_grad_impl<kBf16>(dgates, y, ldy, dwhh, B, T, H, reverse, workspace, workspace_bytes, stream);
}

#ifndef PE_F16_BUILD
extern "C" size_t pe_colsum_workspace_bytes(int cols) { return (size_t)kColsumParts * cols * sizeof(double); }
#endif

// out0[c] = out1[c] = sum_r x[r*ld + c]   (out1 optional: b_ih and b_hh share one gradient)
#ifndef PE_F16_BUILD
extern "C" int pe_colsum(const float* x, long rows, int cols, long ld, float* out0, float* out1, void* workspace,
                         size_t workspace_bytes, void* stream) {
  if (!x || !out0 || rows <= 0 || cols <= 0) return PE_E_ARG;
  if ((cols & 3) || (ld & 3) || (reinterpret_cast<uintptr_t>(x) & 15)) return PE_E_UNSUPPORTED;   // float4 loads
  if (!workspace || workspace_bytes < pe_colsum_workspace_bytes(cols)) return PE_E_WORKSPACE;
  double* partial = reinterpret_cast<double*>(workspace);
  const int parts = rows < kColsumParts ? (int)rows : kColsumParts;
  hipStream_t st = pe_stream(stream);
  hipLaunchKernelGGL(colsum_partial_kernel, dim3(pe_cdiv(cols, 256), parts), dim3(256), 0, st, x, rows, cols, ld,
                     partial);
  PE_LAUNCH_CHECK();
  hipLaunchKernelGGL(colsum_final_kernel, dim3(pe_cdiv(cols, 4)), dim3(256), 0, st, partial, parts, cols, out0,
                     out1);
  PE_LAUNCH_CHECK();
  return PE_OK;
}
#endif
