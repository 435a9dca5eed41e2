// HBM-bound stages of the JDCNet stack on channels-last activations [rows = B*T][F][C]:
// train-mode BatchNorm2d statistics (model.py:25,37,54,150,159), the fused
// BN -> LeakyReLU(0.01) -> MaxPool2d((1, k)) blocks (model.py:36-41,148-153) forward and backward,
// the detector-branch max-pools (model.py:45-49), dropout (model.py:40,56) and the
// (B,256,T,2) -> (B,T,512) re-layout of model.py:93,112.  Everything moves float4 (4 channels).
#include <type_traits>
#include "act16.h"

namespace {
using namespace pe;

constexpr int kMaxPartials = 4096;

// ---------------------------------------------------------------- per-channel two-value reduction
// Each thread owns one channel quad and every G-th pixel; sums are kept in double so the biased
// variance E[x^2] - E[x]^2 loses nothing at 4M samples per channel.
template <class F>
__device__ __forceinline__ void column_reduce2(F&& f, long n_pix, int C, double* partial /*[grid][2][C]*/) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* red = reinterpret_cast<double*>(smem_raw);            // [G][2][C]
  const int quads = C >> 2;
  const int G = 256 / quads;
  const int q = threadIdx.x % quads, g = threadIdx.x / quads;
  double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
  if (g < G) {
    for (long p = (long)blockIdx.x * G + g; p < n_pix; p += (long)gridDim.x * G) {
      float4 a, b;
      f(p, q * 4, a, b);
      s1[0] += a.x; s1[1] += a.y; s1[2] += a.z; s1[3] += a.w;
      s2[0] += b.x; s2[1] += b.y; s2[2] += b.z; s2[3] += b.w;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      red[(g * 2 + 0) * C + q * 4 + i] = s1[i];
      red[(g * 2 + 1) * C + q * 4 + i] = s2[i];
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 256) {
    double s = 0;
    for (int gg = 0; gg < G; ++gg) s += red[gg * 2 * C + i];
    partial[(long)blockIdx.x * 2 * C + i] = s;
  }
}

__device__ __forceinline__ float lrelu(float v, float slope) { return v > 0.f ? v : v * slope; }

// ---------------------------------------------------------------- BN statistics
template <class TA>
__global__ __launch_bounds__(256) void bn_stats_partial_kernel(const TA* __restrict__ x, long n_pix, int C,
                                                               double* __restrict__ partial) {
  column_reduce2(
      [&](long p, int c, float4& a, float4& b) {
        a = ld4(x + p * C + c);
        b = make_float4(a.x * a.x, a.y * a.y, a.z * a.z, a.w * a.w);
      },
      n_pix, C, partial);
}

// [nparts][2][C] -> [gridDim.x][2][C]: a thread owns one of the 2C columns, so every read is coalesced across the
// block; part p goes to block p % gridDim.x and is added in increasing p (fixed order: deterministic)
__global__ __launch_bounds__(256) void bn_partials_fold_kernel(const double* __restrict__ partial, int nparts, int C2,
                                                               double* __restrict__ out) {
  for (int col = threadIdx.x; col < C2; col += 256) {
    double s = 0.0;
    for (int p = blockIdx.x; p < nparts; p += gridDim.x) s += partial[(long)p * C2 + col];
    out[(long)blockIdx.x * C2 + col] = s;
  }
}

__global__ void bn_stats_finalize_kernel(const double* __restrict__ partial, int nparts, long n_pix, int C,
                                         const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                         float momentum, float* __restrict__ running_mean,
                                         float* __restrict__ running_var, float* __restrict__ mean_out,
                                         float* __restrict__ invstd_out, float* __restrict__ scale,
                                         float* __restrict__ shift) {
  // one wave per channel (4 channels per 256-thread block)
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= C) return;
  const double s1 = pe_wave_strided_sum(partial + c, 2L * C, nparts);
  const double s2 = pe_wave_strided_sum(partial + C + c, 2L * C, nparts);
  if ((threadIdx.x & 63) != 0) return;
  const double n = (double)n_pix;
  const double mean = s1 / n;
  double var = s2 / n - mean * mean;
  if (var < 0) var = 0;
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  mean_out[c] = (float)mean;
  invstd_out[c] = invstd;
  const float sc = gamma[c] * invstd;
  scale[c] = sc;
  shift[c] = beta[c] - (float)mean * sc;
  if (running_mean) {
    const double unbiased = n > 1 ? var * n / (n - 1.0) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
  }
}

__global__ void bn_eval_affine_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                      const float* __restrict__ rm, const float* __restrict__ rv, float eps, int C,
                                      float* __restrict__ scale, float* __restrict__ shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float invstd = 1.0f / sqrtf(rv[c] + eps);
  const float sc = gamma[c] * invstd;
  scale[c] = sc;
  shift[c] = beta[c] - rm[c] * sc;
}

// ---------------------------------------------------------------- "h2" scale source from the producing pass
// A pass that writes a tensor a later fp32 product reads can leave the tensor's largest magnitude behind: every
// thread keeps max |v| of what it stores (v_max_f32 with |.| source modifiers), the workgroup folds it and one
// no-return atomicMax per workgroup merges the IEEE bit patterns (ordered like unsigned integers) into *amax, a
// word the caller zeroed.  Exact and order-independent; the separate pe_absmax pass over the tensor disappears.
__device__ __forceinline__ float amax4(float m, const float4& v) {
  return fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
}

__device__ __forceinline__ void amax_commit(float m, unsigned* __restrict__ amax) {
  if (amax == nullptr) return;                                   // (uniform across the grid)
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  __shared__ float amax_red[4];
  if ((threadIdx.x & 63) == 0) amax_red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float w = fmaxf(fmaxf(amax_red[0], amax_red[1]), fmaxf(amax_red[2], amax_red[3]));
    atomicMax(amax, __float_as_uint(w));
  }
}

// ---------------------------------------------------------------- BN -> LReLU -> MaxPool(1,k) forward
// x: [rows][Fin][C]; y: pixel (row, fo) at y[(row*Fout + fo)*ldy + coff + c]
template <class TA>
__global__ __launch_bounds__(256) void bn_act_pool_fwd_kernel(const TA* __restrict__ x,
                                                              const float* __restrict__ scale,
                                                              const float* __restrict__ shift, float slope,
                                                              TA* __restrict__ y, long n_out_pix, int Fin, int C,
                                                              int pool, long ldy, int coff,
                                                              unsigned* __restrict__ amax) {
  const int quads = C >> 2;
  const int Fout = Fin / pool;
  float am = 0.f;
  // thread = (channel quad q, pixel lane g): a thread's quad never changes, so its per-channel vectors are loaded once,
  // not per pixel (with bf16 tensors they were most of the bytes a thread requested); 256 / quads pixels per block and
  // trip (C = 192: 5 pixels, 16 idle threads)
  const int G = 256 / quads, q = threadIdx.x % quads, g = threadIdx.x / quads;
  const float4 sc = *reinterpret_cast<const float4*>(scale + q * 4), sh = *reinterpret_cast<const float4*>(shift + q * 4);
  for (long op = (long)blockIdx.x * G + g; g < G && op < n_out_pix; op += (long)gridDim.x * G) {
    const long row = op / Fout;
    const int fo = (int)(op % Fout);
    const TA* xp = x + ((row * Fin + (long)fo * pool) * C + q * 4);
    float4 m;
    for (int j = 0; j < pool; ++j) {
      const float4 v = ld4(xp + (long)j * C);
      float4 a;
      a.x = lrelu(fmaf(v.x, sc.x, sh.x), slope);
      a.y = lrelu(fmaf(v.y, sc.y, sh.y), slope);
      a.z = lrelu(fmaf(v.z, sc.z, sh.z), slope);
      a.w = lrelu(fmaf(v.w, sc.w, sh.w), slope);
      if (j == 0) m = a;
      else { m.x = fmaxf(m.x, a.x); m.y = fmaxf(m.y, a.y); m.z = fmaxf(m.z, a.z); m.w = fmaxf(m.w, a.w); }
    }
    st4(y + op * ldy + coff + q * 4, m);
    am = amax4(am, m);
  }
  amax_commit(am, amax);
}

// dz for the input pixel (row, f) of a BN->LReLU->MaxPool block, recomputed from x and dy.
// The max-pool routes dy to the FIRST maximum of each window (torch max_pool2d backward).
__device__ __forceinline__ float dz_one(const float* __restrict__ xwin, long cstride, int j, int pool, float sc,
                                        float sh, float slope, float dyv) {
  // xwin points at window element 0 of this channel
  float best = lrelu(fmaf(xwin[0], sc, sh), slope);
  int arg = 0;
  for (int k = 1; k < pool; ++k) {
    const float a = lrelu(fmaf(xwin[(long)k * cstride], sc, sh), slope);
    if (a > best) { best = a; arg = k; }
  }
  if (arg != j) return 0.f;
  const float z = fmaf(xwin[(long)j * cstride], sc, sh);
  return z > 0.f ? dyv : dyv * slope;
}

template <class TA>
struct BnBwdArgsT {
  const TA* x;          // [rows][Fin][C]
  const TA* dy;         // pixel (row, fo) at dy[(row*Fout + fo)*lddy + coff + c]
  const float* scale;
  const float* shift;
  const float* mean;
  const float* invstd;
  float slope;
  long n_in_pix;        // rows * Fin
  int Fin, C, pool;
  long lddy;
  int coff;
};
typedef BnBwdArgsT<float> BnBwdArgs;

__device__ __forceinline__ float4 dz_quad(const BnBwdArgs& a, long p, int c) {
  const int Fout = a.Fin / a.pool;
  const long row = p / a.Fin;
  const int f = (int)(p % a.Fin);
  const int fo = f / a.pool, j = f - fo * a.pool;
  float4 out = make_float4(0.f, 0.f, 0.f, 0.f);
  if (fo >= Fout) return out;                                   // floor-mode remainder gets no gradient
  const float4 dyv = *reinterpret_cast<const float4*>(a.dy + (row * Fout + fo) * a.lddy + a.coff + c);
  const float* xw = a.x + (row * a.Fin + (long)fo * a.pool) * a.C + c;
  const float4 sc = *reinterpret_cast<const float4*>(a.scale + c);
  const float4 sh = *reinterpret_cast<const float4*>(a.shift + c);
  if (a.pool == 1) {
    const float4 v = *reinterpret_cast<const float4*>(xw);
    out.x = fmaf(v.x, sc.x, sh.x) > 0.f ? dyv.x : dyv.x * a.slope;
    out.y = fmaf(v.y, sc.y, sh.y) > 0.f ? dyv.y : dyv.y * a.slope;
    out.z = fmaf(v.z, sc.z, sh.z) > 0.f ? dyv.z : dyv.z * a.slope;
    out.w = fmaf(v.w, sc.w, sh.w) > 0.f ? dyv.w : dyv.w * a.slope;
    return out;
  }
  out.x = dz_one(xw + 0, a.C, j, a.pool, sc.x, sh.x, a.slope, dyv.x);
  out.y = dz_one(xw + 1, a.C, j, a.pool, sc.y, sh.y, a.slope, dyv.y);
  out.z = dz_one(xw + 2, a.C, j, a.pool, sc.z, sh.z, a.slope, dyv.z);
  out.w = dz_one(xw + 3, a.C, j, a.pool, sc.w, sh.w, a.slope, dyv.w);
  return out;
}

__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const BnBwdArgs a, double* __restrict__ partial) {
  column_reduce2(
      [&](long p, int c, float4& s, float4& t) {
        s = dz_quad(a, p, c);
        const float4 v = *reinterpret_cast<const float4*>(a.x + p * a.C + c);
        const float4 mu = *reinterpret_cast<const float4*>(a.mean + c);
        const float4 is = *reinterpret_cast<const float4*>(a.invstd + c);
        t = make_float4(s.x * ((v.x - mu.x) * is.x), s.y * ((v.y - mu.y) * is.y), s.z * ((v.z - mu.z) * is.z),
                        s.w * ((v.w - mu.w) * is.w));
      },
      a.n_in_pix, a.C, partial);
}

__global__ void bn_bwd_finalize_kernel(const double* __restrict__ partial, int nparts, long n_pix, int C,
                                       float* __restrict__ dgamma, float* __restrict__ dbeta,
                                       float* __restrict__ c1, float* __restrict__ c2) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= C) return;
  const double s1 = pe_wave_strided_sum(partial + c, 2L * C, nparts);
  const double s2 = pe_wave_strided_sum(partial + C + c, 2L * C, nparts);
  if ((threadIdx.x & 63) != 0) return;
  dbeta[c] = (float)s1;
  dgamma[c] = (float)s2;
  c1[c] = (float)(s1 / (double)n_pix);
  c2[c] = (float)(s2 / (double)n_pix);
}

// dx = gamma*invstd * (dz - mean(dz) - xhat * mean(dz*xhat));  scale == gamma*invstd
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const BnBwdArgs a, const float* __restrict__ c1,
                                                           const float* __restrict__ c2, float* __restrict__ dx,
                                                           unsigned* __restrict__ amax) {
  const int quads = a.C >> 2;
  const long total = a.n_in_pix * quads;
  float am = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % quads) * 4;
    const long p = i / quads;
    const float4 dz = dz_quad(a, p, c);
    const float4 v = *reinterpret_cast<const float4*>(a.x + p * a.C + c);
    const float4 mu = *reinterpret_cast<const float4*>(a.mean + c);
    const float4 is = *reinterpret_cast<const float4*>(a.invstd + c);
    const float4 sc = *reinterpret_cast<const float4*>(a.scale + c);
    const float4 k1 = *reinterpret_cast<const float4*>(c1 + c);
    const float4 k2 = *reinterpret_cast<const float4*>(c2 + c);
    float4 o;
    o.x = sc.x * (dz.x - k1.x - (v.x - mu.x) * is.x * k2.x);
    o.y = sc.y * (dz.y - k1.y - (v.y - mu.y) * is.y * k2.y);
    o.z = sc.z * (dz.z - k1.z - (v.z - mu.z) * is.z * k2.z);
    o.w = sc.w * (dz.w - k1.w - (v.w - mu.w) * is.w * k2.w);
    *reinterpret_cast<float4*>(dx + p * a.C + c) = o;
    am = amax4(am, o);
  }
  amax_commit(am, amax);
}

// Window form of the two backward passes for POOL in {1, 2, 4}: one thread owns a whole pool window of a
// channel quad, so x is read exactly once per pass (float4, coalesced along C), the arg-max is found once
// and only one 64-bit division is paid per window.  Item w of a row is window w for w < Fout, plus one
// trailing item for the floor-mode remainder pixels (dz = 0, but they count in the means and get a dx).
template <int POOL>
struct BnWindow {
  float4 v[POOL], dz[POOL];
  long p0;       // first input pixel of the item
  int n;         // pixels in it
};

template <int POOL, class TA>
__device__ __forceinline__ void bn_window(const BnBwdArgsT<TA>& a, long item, int nwin, int c, BnWindow<POOL>& w,
                                          const float4& sc, const float4& sh) {
  const int Fout = a.Fin / POOL;
  const long row = item / nwin;
  const int wi = (int)(item - row * nwin);
  const bool real = wi < Fout;
  w.n = real ? POOL : a.Fin - Fout * POOL;
  w.p0 = row * a.Fin + (long)wi * POOL;
  const TA* xp = a.x + w.p0 * a.C + c;
#pragma unroll
  for (int j = 0; j < POOL; ++j) {
    w.v[j] = j < w.n ? ld4(xp + (long)j * a.C) : make_float4(0.f, 0.f, 0.f, 0.f);
    w.dz[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if (!real) return;
  const float4 dyv = ld4(a.dy + (row * Fout + wi) * a.lddy + a.coff + c);
  const float scv[4] = {sc.x, sc.y, sc.z, sc.w}, shv[4] = {sh.x, sh.y, sh.z, sh.w};
  const float dys[4] = {dyv.x, dyv.y, dyv.z, dyv.w};
#pragma unroll
  for (int ch = 0; ch < 4; ++ch) {
    float best = lrelu(fmaf(reinterpret_cast<const float*>(&w.v[0])[ch], scv[ch], shv[ch]), a.slope);
    int arg = 0;
#pragma unroll
    for (int j = 1; j < POOL; ++j) {
      const float act = lrelu(fmaf(reinterpret_cast<const float*>(&w.v[j])[ch], scv[ch], shv[ch]), a.slope);
      if (act > best) { best = act; arg = j; }               // first maximum wins (torch max_pool2d backward)
    }
    const float g = best > 0.f ? dys[ch] : dys[ch] * a.slope;  // sign(lrelu(z)) == sign(z)
#pragma unroll
    for (int j = 0; j < POOL; ++j) reinterpret_cast<float*>(&w.dz[j])[ch] = j == arg ? g : 0.f;
  }
}

template <int POOL, class TA>
__global__ __launch_bounds__(256) void bn_bwd_partial_win_kernel(const BnBwdArgsT<TA> a, long n_items, int nwin,
                                                                 double* __restrict__ partial) {
  const int c0 = (threadIdx.x % (a.C >> 2)) * 4;               // column_reduce2 hands this thread no other quad
  const float4 mu = *reinterpret_cast<const float4*>(a.mean + c0), is = *reinterpret_cast<const float4*>(a.invstd + c0);
  const float4 sc = *reinterpret_cast<const float4*>(a.scale + c0), sh = *reinterpret_cast<const float4*>(a.shift + c0);
  column_reduce2(
      [&](long item, int c, float4& s, float4& t) {
        BnWindow<POOL> w;
        bn_window<POOL, TA>(a, item, nwin, c, w, sc, sh);
        s = make_float4(0.f, 0.f, 0.f, 0.f);
        t = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < POOL; ++j) {                      // one dz per channel is non-zero: plain fp32 sums are exact
          s.x += w.dz[j].x; s.y += w.dz[j].y; s.z += w.dz[j].z; s.w += w.dz[j].w;
          t.x += w.dz[j].x * ((w.v[j].x - mu.x) * is.x); t.y += w.dz[j].y * ((w.v[j].y - mu.y) * is.y);
          t.z += w.dz[j].z * ((w.v[j].z - mu.z) * is.z); t.w += w.dz[j].w * ((w.v[j].w - mu.w) * is.w);
        }
      },
      n_items, a.C, partial);
}

template <int POOL, class TA>
__global__ __launch_bounds__(256) void bn_bwd_apply_win_kernel(const BnBwdArgsT<TA> a, long n_items, int nwin,
                                                               const float* __restrict__ c1,
                                                               const float* __restrict__ c2, TA* __restrict__ dx,
                                                               unsigned* __restrict__ amax) {
  const int quads = a.C >> 2;
  float am = 0.f;
  const int G = 256 / quads, c = (threadIdx.x % quads) * 4, g = threadIdx.x / quads;     // see bn_act_pool_fwd_kernel
  const float4 mu = *reinterpret_cast<const float4*>(a.mean + c), is = *reinterpret_cast<const float4*>(a.invstd + c);
  const float4 sc = *reinterpret_cast<const float4*>(a.scale + c), sh = *reinterpret_cast<const float4*>(a.shift + c);
  const float4 k1 = *reinterpret_cast<const float4*>(c1 + c), k2 = *reinterpret_cast<const float4*>(c2 + c);
  for (long item = (long)blockIdx.x * G + g; g < G && item < n_items; item += (long)gridDim.x * G) {
    BnWindow<POOL> w;
    bn_window<POOL, TA>(a, item, nwin, c, w, sc, sh);
#pragma unroll
    for (int j = 0; j < POOL; ++j) {
      if (j >= w.n) break;
      float4 o;
      o.x = sc.x * (w.dz[j].x - k1.x - (w.v[j].x - mu.x) * is.x * k2.x);
      o.y = sc.y * (w.dz[j].y - k1.y - (w.v[j].y - mu.y) * is.y * k2.y);
      o.z = sc.z * (w.dz[j].z - k1.z - (w.v[j].z - mu.z) * is.z * k2.z);
      o.w = sc.w * (w.dz[j].w - k1.w - (w.v[j].w - mu.w) * is.w * k2.w);
      st4(dx + (w.p0 + j) * a.C + c, o);
      am = amax4(am, o);
    }
  }
  amax_commit(am, amax);
}

// ---------------------------------------------------------------- plain MaxPool(1,k) (detector taps)
template <class TA>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const TA* __restrict__ x, TA* __restrict__ y,
                                                          long n_out_pix, int Fin, int C, int pool, long ldy,
                                                          int coff, unsigned char* __restrict__ arg) {
  // arg (optional, [n_out_pix][C] bytes): the window position of each maximum (first one wins, as torch's
  // max_pool2d backward routes it), so that the backward pass does not have to read x again
  const int quads = C >> 2, Fout = Fin / pool;
  const long total = n_out_pix * quads;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int q = (int)(i % quads);
    const long op = i / quads, row = op / Fout;
    const int fo = (int)(op % Fout);
    const TA* xp = x + ((row * Fin + (long)fo * pool) * C + q * 4);
    float4 m = ld4(xp);
    int a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    int j = 1;
    for (; j + 4 <= pool; j += 4) {                               // four window rows in flight
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = ld4(xp + (long)(j + u) * C);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (v[u].x > m.x) { m.x = v[u].x; a0 = j + u; }
        if (v[u].y > m.y) { m.y = v[u].y; a1 = j + u; }
        if (v[u].z > m.z) { m.z = v[u].z; a2 = j + u; }
        if (v[u].w > m.w) { m.w = v[u].w; a3 = j + u; }
      }
    }
    for (; j < pool; ++j) {
      const float4 v = ld4(xp + (long)j * C);
      if (v.x > m.x) { m.x = v.x; a0 = j; }
      if (v.y > m.y) { m.y = v.y; a1 = j; }
      if (v.z > m.z) { m.z = v.z; a2 = j; }
      if (v.w > m.w) { m.w = v.w; a3 = j; }
    }
    st4(y + op * ldy + coff + q * 4, m);
    if (arg != nullptr)
      *reinterpret_cast<uchar4*>(arg + op * C + q * 4) = make_uchar4((unsigned char)a0, (unsigned char)a1,
                                                                     (unsigned char)a2, (unsigned char)a3);
  }
}

// dx[first argmax of each window] += dy   (one thread owns a whole window: no atomics)
template <class TA>
__global__ __launch_bounds__(256) void maxpool_bwd_add_kernel(const TA* __restrict__ x,
                                                              const TA* __restrict__ dy, TA* __restrict__ dx,
                                                              long n_out_pix, int Fin, int C, int pool, long lddy,
                                                              int coff, unsigned* __restrict__ amax,
                                                              const unsigned char* __restrict__ arg) {
  // one thread per (window, channel quad): float4 loads, four window rows in flight, first maximum wins; with `arg`
  // (the positions maxpool_fwd_kernel recorded) x is not read at all
  const int Fout = Fin / pool, quads = C >> 2;
  const long total = n_out_pix * quads;
  float am = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long op = i / quads;
    const int c = (int)(i - op * quads) * 4;
    const long row = op / Fout;
    const int fo = (int)(op - row * Fout);
    int a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    if (arg != nullptr) {
      const uchar4 a = *reinterpret_cast<const uchar4*>(arg + op * C + c);
      a0 = a.x; a1 = a.y; a2 = a.z; a3 = a.w;
    } else {
    const TA* xp = x + (row * Fin + (long)fo * pool) * C + c;
    float4 best = ld4(xp);
    int jn = 1;
    for (; jn + 4 <= pool; jn += 4) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = ld4(xp + (long)(jn + u) * C);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (v[u].x > best.x) { best.x = v[u].x; a0 = jn + u; }
        if (v[u].y > best.y) { best.y = v[u].y; a1 = jn + u; }
        if (v[u].z > best.z) { best.z = v[u].z; a2 = jn + u; }
        if (v[u].w > best.w) { best.w = v[u].w; a3 = jn + u; }
      }
    }
    for (; jn < pool; ++jn) {
      const float4 v = ld4(xp + (long)jn * C);
      if (v.x > best.x) { best.x = v.x; a0 = jn; }
      if (v.y > best.y) { best.y = v.y; a1 = jn; }
      if (v.z > best.z) { best.z = v.z; a2 = jn; }
      if (v.w > best.w) { best.w = v.w; a3 = jn; }
    }
    }
    const float4 g = ld4(dy + op * lddy + coff + c);
    TA* dp = dx + (row * Fin + (long)fo * pool) * C + c;
    const float4 nv = make_float4(ld1(dp + (long)a0 * C) + g.x, ld1(dp + (long)a1 * C + 1) + g.y,
                                  ld1(dp + (long)a2 * C + 2) + g.z, ld1(dp + (long)a3 * C + 3) + g.w);
    st1(dp + (long)a0 * C, nv.x);
    st1(dp + (long)a1 * C + 1, nv.y);
    st1(dp + (long)a2 * C + 2, nv.z);
    st1(dp + (long)a3 * C + 3, nv.w);
    am = amax4(am, nv);          // merged into dx's word: an upper bound of max |dx| (elements that shrank keep their old share)
  }
  amax_commit(am, amax);
}

// ---------------------------------------------------------------- dropout (Philox4x32-10)
// rows x cols, x row stride ldx, y row stride ldy; mask is dense [rows*cols] bytes (1 = kept).
// mask_in != NULL replays a given mask (parity tests); otherwise keep = (u >= p).
template <class TA>
__global__ __launch_bounds__(256) void dropout_fwd_kernel(const TA* __restrict__ x, long ldx,
                                                          TA* __restrict__ y, long ldy,
                                                          const uint8_t* __restrict__ mask_in,
                                                          uint8_t* __restrict__ mask_out, long rows, int cols,
                                                          float p, float scale, uint64_t seed, uint64_t offset) {
  const int quads = cols >> 2;
  const long total = rows * quads;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % quads) * 4;
    const long r = i / quads;
    uint8_t keep[4];
    if (mask_in) {
      const uchar4 m = *reinterpret_cast<const uchar4*>(mask_in + r * cols + c);
      keep[0] = m.x; keep[1] = m.y; keep[2] = m.z; keep[3] = m.w;
    } else {
      uint32_t rnd[4];
      philox4(seed, offset + (uint64_t)i, rnd);
#pragma unroll
      for (int k = 0; k < 4; ++k) keep[k] = ((float)(rnd[k] >> 8) * (1.0f / 16777216.0f)) >= p ? 1 : 0;
    }
    const float4 v = ld4(x + r * ldx + c);
    float4 o;
    o.x = keep[0] ? v.x * scale : 0.f;
    o.y = keep[1] ? v.y * scale : 0.f;
    o.z = keep[2] ? v.z * scale : 0.f;
    o.w = keep[3] ? v.w * scale : 0.f;
    st4(y + r * ldy + c, o);
    if (mask_out) *reinterpret_cast<uchar4*>(mask_out + r * cols + c) = make_uchar4(keep[0], keep[1], keep[2], keep[3]);
  }
}

// ---------------------------------------------------------------- (B,256,T,2) <-> (B,T,512) re-layout
// channels-last pixel pair (row, w in {0,1}) at x[(row*2 + w)*ldx + coff + c]  <->  seq[row][c*2 + w]
template <class TA>
__global__ __launch_bounds__(256) void nhwc_to_seq_kernel(const TA* __restrict__ x, long ldx, int coff,
                                                          float* __restrict__ seq, long rows, int C) {
  const long total = rows * C;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C);
    const long r = i / C;
    const float a = ld1(x + (r * 2 + 0) * ldx + coff + c);
    const float b = ld1(x + (r * 2 + 1) * ldx + coff + c);
    *reinterpret_cast<float2*>(seq + r * 2 * C + 2 * c) = make_float2(a, b);
  }
}

template <class TA>
__global__ __launch_bounds__(256) void seq_to_nhwc_kernel(const float* __restrict__ seq, TA* __restrict__ x,
                                                          long ldx, int coff, long rows, int C, int accumulate) {
  const long total = rows * C;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C);
    const long r = i / C;
    const float2 v = *reinterpret_cast<const float2*>(seq + r * 2 * C + 2 * c);
    TA* d0 = x + (r * 2 + 0) * ldx + coff + c;
    TA* d1 = x + (r * 2 + 1) * ldx + coff + c;
    if (accumulate) { st1(d0, ld1(d0) + v.x); st1(d1, ld1(d1) + v.y); }
    else { st1(d0, v.x); st1(d1, v.y); }
  }
}

// strided 2-D copy / add: dst[r*ldd + c] (+)= src[r*lds + c]
template <class TA>
__global__ __launch_bounds__(256) void copy2d_kernel(const TA* __restrict__ src, long lds, TA* __restrict__ dst,
                                                     long ldd, long rows, int cols, int accumulate) {
  const int quads = cols >> 2;
  const long total = rows * quads;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % quads) * 4;
    const long r = i / quads;
    float4 v = ld4(src + r * lds + c);
    TA* d = dst + r * ldd + c;
    if (accumulate) { const float4 o = ld4(d); v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
    st4(d, v);
  }
}

int ew_grid(long total_threads) {
  long g = (total_threads + 255) / 256;
  if (g > 16384) g = 16384;
  if (g < 1) g = 1;
  return (int)g;
}

int reduce_grid(long n_pix, int C, int cap = 1024) {
  const int G = 256 / (C / 4);
  long g = (n_pix + G - 1) / G;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

size_t reduce_lds(int C) { return (size_t)(256 / (C / 4)) * 2 * C * sizeof(double); }

bool bn_channels_ok(int C) { return C >= 4 && (C % 4) == 0 && C <= 1024; }

// ---------------------------------------------------------------- largest magnitude of a tensor (h2 operand scale)
// out[0] = max(out[0], IEEE bits of max |x|) over a row-strided [rows][cols] matrix (cols % 4 == 0).  The bit
// patterns of non-negative floats order like unsigned integers, so the cross-workgroup combine is an integer
// atomicMax: exact and independent of the order of arrival (NaN / inf operands give the largest patterns and poison
// the product they feed, as they would any other path).  The caller zeroes out[0] (pe_absmax does).
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, long rows, int cols4, long ld,
                                                     unsigned* __restrict__ out) {
  const long n4 = rows * cols4;
  unsigned m = 0u;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const long r = i / cols4;
    const int c = (int)(i - r * cols4);
    const float4 v = *reinterpret_cast<const float4*>(x + r * ld + c * 4);
    m = max(m, __float_as_uint(v.x) & 0x7fffffffu);
    m = max(m, __float_as_uint(v.y) & 0x7fffffffu);
    m = max(m, __float_as_uint(v.z) & 0x7fffffffu);
    m = max(m, __float_as_uint(v.w) & 0x7fffffffu);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, off, 64));
  __shared__ unsigned red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) atomicMax(out, max(max(red[0], red[1]), max(red[2], red[3])));
}

// absmax of many segments of one buffer in a single launch (the model's parameters inside the flat buffer): block
// (x, y) folds chunk x of segment y and max-merges into out[y]
__global__ __launch_bounds__(256) void absmax_segments_kernel(const float* __restrict__ base,
                                                              const long* __restrict__ seg_off,
                                                              const long* __restrict__ seg_len,
                                                              unsigned* __restrict__ out) {
  const float* x = base + seg_off[blockIdx.y];
  const long n = seg_len[blockIdx.y];
  unsigned m = 0u;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
    m = max(m, __float_as_uint(x[i]) & 0x7fffffffu);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, off, 64));
  __shared__ unsigned red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned w = max(max(red[0], red[1]), max(red[2], red[3]));
    if (w) atomicMax(out + blockIdx.y, w);
  }
}

}  // namespace

extern "C" size_t pe_bn_workspace_bytes(int C) { return (size_t)kMaxPartials * 2 * C * sizeof(double); }

template <class TA>
static int bn_train_stats_impl(const TA* x, long n_pix, int C, const float* gamma, const float* beta, float eps,
                               float momentum, float* running_mean, float* running_var, float* mean, float* invstd,
                               float* scale, float* shift, void* workspace, size_t workspace_bytes, void* stream) {
  if (!x || !gamma || !beta || !mean || !invstd || !scale || !shift || n_pix <= 0) return PE_E_ARG;
  if (!bn_channels_ok(C)) return PE_E_UNSUPPORTED;
  if (!workspace || workspace_bytes < pe_bn_workspace_bytes(C)) return PE_E_WORKSPACE;
  hipStream_t st = pe_stream(stream);
  const int grid = reduce_grid(n_pix, C);
  double* partial = reinterpret_cast<double*>(workspace);
  hipLaunchKernelGGL(bn_stats_partial_kernel<TA>, dim3(grid), dim3(256), reduce_lds(C), st, x, n_pix, C, partial);
  PE_LAUNCH_CHECK();
  hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3(pe_cdiv(C, 4)), dim3(256), 0, st, partial, grid, n_pix, C, gamma,
                     beta, eps, momentum, running_mean, running_var, mean, invstd, scale, shift);
  PE_LAUNCH_CHECK();
  return PE_OK;
}

extern "C" int pe_bn_train_stats(const float* x, long n_pix, int C, const float* gamma, const float* beta, float eps,
                                 float momentum, float* running_mean, float* running_var, float* mean,
                                 float* invstd, float* scale, float* shift, void* workspace, size_t workspace_bytes,
                                 void* stream) {
  return bn_train_stats_impl<float>(x, n_pix, C, gamma, beta, eps, momentum, running_mean, running_var, mean, invstd,
                                    scale, shift, workspace, workspace_bytes, stream);
}

extern "C" int pe_bn_train_stats_a16(const void* x, long n_pix, int C, const float* gamma, const float* beta, float eps,
                                     float momentum, float* running_mean, float* running_var, float* mean,
                                     float* invstd, float* scale, float* shift, void* workspace,
                                     size_t workspace_bytes, void* stream) {
  return bn_train_stats_impl<act16_t>(static_cast<const act16_t*>(x), n_pix, C, gamma, beta, eps, momentum,
                                      running_mean, running_var, mean, invstd, scale, shift, workspace,
                                      workspace_bytes, stream);
}

// BatchNorm training statistics from partials a producer kernel left behind ([nparts][2][C] doubles: column sums
// and sums of squares, e.g. pe_conv3x3_fwd_wf_*'s bn_partials): the finalize half of pe_bn_train_stats.
extern "C" int pe_bn_finalize_stats(const double* partials, int nparts, long n_pix, int C, const float* gamma,
                                    const float* beta, float eps, float momentum, float* running_mean,
                                    float* running_var, float* mean, float* invstd, float* scale, float* shift,
                                    void* workspace, size_t workspace_bytes, void* stream) {
  if (!partials || !gamma || !beta || !mean || !invstd || !scale || !shift || n_pix <= 0 || nparts <= 0) return PE_E_ARG;
  if (!bn_channels_ok(C)) return PE_E_UNSUPPORTED;
  hipStream_t st = pe_stream(stream);
  const double* src = partials;
  int n = nparts;
  if (nparts > 512) {                    // fold tens of thousands of tile partials with coalesced reads first; the
    // per-thread chain is a serial sum, so the fold goes as wide as the finalize below still reads cheaply (30 k parts
    // over 128 blocks = 240 dependent adds per thread took 50 us per BatchNorm)
    constexpr int kFold = 512;
    if (!workspace || workspace_bytes < (size_t)kFold * 2 * C * sizeof(double)) return PE_E_WORKSPACE;
    double* folded = reinterpret_cast<double*>(workspace);
    hipLaunchKernelGGL(bn_partials_fold_kernel, dim3(kFold), dim3(256), 0, st, partials, nparts, 2 * C, folded);
    PE_LAUNCH_CHECK();
    src = folded;
    n = kFold;
  }
  hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3(pe_cdiv(C, 4)), dim3(256), 0, st, src, n, n_pix, C, gamma, beta,
                     eps, momentum, running_mean, running_var, mean, invstd, scale, shift);
  PE_LAUNCH_CHECK();
  return PE_OK;
}

extern "C" int pe_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean,
                                 const float* running_var, float eps, int C, float* scale, float* shift,
                                 void* stream) {
  if (!gamma || !beta || !running_mean || !running_var || !scale || !shift || C <= 0) return PE_E_ARG;
  hipLaunchKernelGGL(bn_eval_affine_kernel, dim3(pe_cdiv(C, 64)), dim3(64), 0, pe_stream(stream), gamma, beta,
                     running_mean, running_var, eps, C, scale, shift);
  PE_LAUNCH_CHECK();
  return PE_OK;
}

template <class TA>
static int bn_act_pool_fwd_impl(const TA* x, const float* scale, const float* shift, float slope, TA* y, long rows,
                                int Fin, int C, int pool, long ldy, int coff, unsigned* amax_out, void* stream) {
  if (!x || !scale || !shift || !y || rows <= 0 || Fin <= 0 || pool <= 0) return PE_E_ARG;
  if (!bn_channels_ok(C) || (ldy & 3) || (coff & 3)) return PE_E_UNSUPPORTED;
  const long n_out = rows * (Fin / pool);
  hipLaunchKernelGGL(bn_act_pool_fwd_kernel<TA>, dim3(ew_grid(pe_cdiv(n_out, 256 / (C / 4)) * 256L)), dim3(256), 0, pe_stream(stream), x,
                     scale, shift, slope, y, n_out, Fin, C, pool, ldy, coff, amax_out);
  PE_LAUNCH_CHECK();
  return PE_OK;
}

extern "C" int pe_bn_act_pool_fwd(const float* x, const float* scale, const float* shift, float slope, float* y,
                                  long rows, int Fin, int C, int pool, long ldy, int coff, unsigned* amax_out,
                                  void* stream) {
  return bn_act_pool_fwd_impl<float>(x, scale, shift, slope, y, rows, Fin, C, pool, ldy, coff, amax_out, stream);
}

extern "C" int pe_bn_act_pool_fwd_a16(const void* x, const float* scale, const float* shift, float slope, void* y,
                                      long rows, int Fin, int C, int pool, long ldy, int coff, void* stream) {
  return bn_act_pool_fwd_impl<act16_t>(static_cast<const act16_t*>(x), scale, shift, slope, static_cast<act16_t*>(y),
                                       rows, Fin, C, pool, ldy, coff, nullptr, stream);
}

template <class TA>
static int bn_act_pool_bwd_impl(const TA* x, const TA* dy, const float* scale, const float* shift, const float* mean,
                                const float* invstd, float slope, TA* dx, float* dgamma, float* dbeta, long rows,
                                int Fin, int C, int pool, long lddy, int coff, void* workspace, size_t workspace_bytes,
                                unsigned* amax_out, void* stream) {
  if (!x || !dy || !scale || !shift || !mean || !invstd || !dx || !dgamma || !dbeta || rows <= 0) return PE_E_ARG;
  if (!bn_channels_ok(C) || (lddy & 3) || (coff & 3) || pool <= 0) return PE_E_UNSUPPORTED;
  const size_t need = pe_bn_workspace_bytes(C) + 2 * (size_t)C * sizeof(float);
  if (!workspace || workspace_bytes < need) return PE_E_WORKSPACE;
  hipStream_t st = pe_stream(stream);
  BnBwdArgsT<TA> a{x, dy, scale, shift, mean, invstd, slope, rows * Fin, Fin, C, pool, lddy, coff};
  double* partial = reinterpret_cast<double*>(workspace);
  float* c1 = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + pe_bn_workspace_bytes(C));
  float* c2 = c1 + C;
  const int nwin = Fin / pool + (Fin % pool ? 1 : 0);
  const long n_items = rows * nwin;
  const bool win = pool == 1 || pool == 2 || pool == 4;
  if (!win && !std::is_same<TA, float>::value) return PE_E_UNSUPPORTED;      // other pool widths: fp32 tensors only
  // bf16 tensors without pooling: 8-byte requests, so four times the workgroups keep the same bytes in flight
  // (partial<1>: 348 -> 257 us; the pooled forms and fp32 tensors, at HBM speed with 1024, lose 5-10 % with more)
  const int grid = reduce_grid(win ? n_items : a.n_in_pix, C,
                               (!std::is_same<TA, float>::value && pool == 1) ? kMaxPartials : 1024);
  if (pool == 1)
    hipLaunchKernelGGL((bn_bwd_partial_win_kernel<1, TA>), dim3(grid), dim3(256), reduce_lds(C), st, a, n_items, nwin, partial);
  else if (pool == 2)
    hipLaunchKernelGGL((bn_bwd_partial_win_kernel<2, TA>), dim3(grid), dim3(256), reduce_lds(C), st, a, n_items, nwin, partial);
  else if (pool == 4)
    hipLaunchKernelGGL((bn_bwd_partial_win_kernel<4, TA>), dim3(grid), dim3(256), reduce_lds(C), st, a, n_items, nwin, partial);
  else if constexpr (std::is_same<TA, float>::value)
    hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3(grid), dim3(256), reduce_lds(C), st, a, partial);
  PE_LAUNCH_CHECK();
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(pe_cdiv(C, 4)), dim3(256), 0, st, partial, grid, a.n_in_pix, C,
                     dgamma, dbeta, c1, c2);
  PE_LAUNCH_CHECK();
  const int agrid = win ? ew_grid(pe_cdiv(n_items, 256 / (C / 4)) * 256L) : ew_grid(a.n_in_pix * (C / 4));
  if (pool == 1)
    hipLaunchKernelGGL((bn_bwd_apply_win_kernel<1, TA>), dim3(agrid), dim3(256), 0, st, a, n_items, nwin, c1, c2, dx,
                       amax_out);
  else if (pool == 2)
    hipLaunchKernelGGL((bn_bwd_apply_win_kernel<2, TA>), dim3(agrid), dim3(256), 0, st, a, n_items, nwin, c1, c2, dx,
                       amax_out);
  else if (pool == 4)
    hipLaunchKernelGGL((bn_bwd_apply_win_kernel<4, TA>), dim3(agrid), dim3(256), 0, st, a, n_items, nwin, c1, c2, dx,
                       amax_out);
  else if constexpr (std::is_same<TA, float>::value)
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(agrid), dim3(256), 0, st, a, c1, c2, dx, amax_out);
  PE_LAUNCH_CHECK();
  return PE_OK;
}

extern "C" int pe_bn_act_pool_bwd(const float* x, const float* dy, const float* scale, const float* shift,
                                  const float* mean, const float* invstd, float slope, float* dx, float* dgamma,
                                  float* dbeta, long rows, int Fin, int C, int pool, long lddy, int coff,
                                  void* workspace, size_t workspace_bytes, unsigned* amax_out, void* stream) {
  return bn_act_pool_bwd_impl<float>(x, dy, scale, shift, mean, invstd, slope, dx, dgamma, dbeta, rows, Fin, C, pool,
                                     lddy, coff, workspace, workspace_bytes, amax_out, stream);
}

extern "C" int pe_bn_act_pool_bwd_a16(const void* x, const void* dy, const float* scale, const float* shift,
                                      const float* mean, const float* invstd, float slope, void* dx, float* dgamma,
                                      float* dbeta, long rows, int Fin, int C, int pool, long lddy, int coff,
                                      void* workspace, size_t workspace_bytes, void* stream) {
  return bn_act_pool_bwd_impl<act16_t>(static_cast<const act16_t*>(x), static_cast<const act16_t*>(dy), scale, shift,
                                       mean, invstd, slope, static_cast<act16_t*>(dx), dgamma, dbeta, rows, Fin, C,
                                       pool, lddy, coff, workspace, workspace_bytes, nullptr, stream);
}

template <class TA>
static int maxpool_fwd_impl(const TA* x, TA* y, long rows, int Fin, int C, int pool, long ldy, int coff,
                            unsigned char* argmax_out, void* stream) {
  if (!x || !y || rows <= 0 || Fin <= 0 || pool <= 0) return PE_E_ARG;
  if ((C & 3) || (ldy & 3) || (coff & 3) || (argmax_out && pool > 255)) return PE_E_UNSUPPORTED;
  const long n_out = rows * (Fin / pool);
  hipLaunchKernelGGL(maxpool_fwd_kernel<TA>, dim3(ew_grid(n_out * (C / 4))), dim3(256), 0, pe_stream(stream), x, y,
                     n_out, Fin, C, pool, ldy, coff, argmax_out);
  PE_LAUNCH_CHECK();
  return PE_OK;
}

extern "C" int pe_maxpool_fwd(const float* x, float* y, long rows, int Fin, int C, int pool, long ldy, int coff,
                              unsigned char* argmax_out, void* stream) {
  return maxpool_fwd_impl<float>(x, y, rows, Fin, C, pool, ldy, coff, argmax_out, stream);
}

extern "C" int pe_maxpool_fwd_a16(const void* x, void* y, long rows, int Fin, int C, int pool, long ldy, int coff,
                                  unsigned char* argmax_out, void* stream) {
  return maxpool_fwd_impl<act16_t>(static_cast<const act16_t*>(x), static_cast<act16_t*>(y), rows, Fin, C, pool, ldy,
                                   coff, argmax_out, stream);
}

template <class TA>
static int maxpool_bwd_add_impl(const TA* x, const unsigned char* argmax, const TA* dy, TA* dx, long rows, int Fin,
                                int C, int pool, long lddy, int coff, unsigned* amax_out, void* stream) {
  if ((!x && !argmax) || !dy || !dx || rows <= 0 || Fin <= 0 || pool <= 0 || C <= 0) return PE_E_ARG;
  if ((C & 3) || (lddy & 3) || (coff & 3)) return PE_E_UNSUPPORTED;
  const long n_out = rows * (Fin / pool);
  hipLaunchKernelGGL(maxpool_bwd_add_kernel<TA>, dim3(ew_grid(n_out * (C / 4))), dim3(256), 0, pe_stream(stream), x,
                     dy, dx, n_out, Fin, C, pool, lddy, coff, amax_out, argmax);
  PE_LAUNCH_CHECK();
  return PE_OK;
}

extern "C" int pe_maxpool_bwd_add(const float* x, const unsigned char* argmax, const float* dy, float* dx, long rows,
                                  int Fin, int C, int pool, long lddy, int coff, unsigned* amax_out, void* stream) {
  return maxpool_bwd_add_impl<float>(x, argmax, dy, dx, rows, Fin, C, pool, lddy, coff, amax_out, stream);
}

extern "C" int pe_maxpool_bwd_add_a16(const void* x, const unsigned char* argmax, const void* dy, void* dx, long rows,
                                      int Fin, int C, int pool, long lddy, int coff, void* stream) {
  return maxpool_bwd_add_impl<act16_t>(static_cast<const act16_t*>(x), argmax, static_cast<const act16_t*>(dy),
                                       static_cast<act16_t*>(dx), rows, Fin, C, pool, lddy, coff, nullptr, stream);
}

template <class TA>
static int dropout_fwd_impl(const TA* x, long ldx, TA* y, long ldy, const unsigned char* mask_in,
                            unsigned char* mask_out, long rows, int cols, float p, unsigned long long seed,
                            unsigned long long offset, void* stream) {
  if (!x || !y || rows <= 0 || cols <= 0 || p < 0.f || p >= 1.f) return PE_E_ARG;
  if ((cols & 3) || (ldx & 3) || (ldy & 3)) return PE_E_UNSUPPORTED;
  hipLaunchKernelGGL(dropout_fwd_kernel<TA>, dim3(ew_grid(rows * (cols / 4))), dim3(256), 0, pe_stream(stream), x, ldx,
                     y, ldy, mask_in, mask_out, rows, cols, p, 1.0f / (1.0f - p), (uint64_t)seed, (uint64_t)offset);
  PE_LAUNCH_CHECK();
  return PE_OK;
}

extern "C" int pe_dropout_fwd(const float* x, long ldx, float* y, long ldy, const unsigned char* mask_in,
                              unsigned char* mask_out, long rows, int cols, float p, unsigned long long seed,
                              unsigned long long offset, void* stream) {
  return dropout_fwd_impl<float>(x, ldx, y, ldy, mask_in, mask_out, rows, cols, p, seed, offset, stream);
}

extern "C" int pe_dropout_fwd_a16(const void* x, long ldx, void* y, long ldy, const unsigned char* mask_in,
                                  unsigned char* mask_out, long rows, int cols, float p, unsigned long long seed,
                                  unsigned long long offset, void* stream) {
  return dropout_fwd_impl<act16_t>(static_cast<const act16_t*>(x), ldx, static_cast<act16_t*>(y), ldy, mask_in,
                                   mask_out, rows, cols, p, seed, offset, stream);
}

extern "C" int pe_nhwc_to_seq(const float* x, long ldx, int coff, float* seq, long rows, int C, void* stream) {
  if (!x || !seq || rows <= 0 || C <= 0) return PE_E_ARG;
  hipLaunchKernelGGL(nhwc_to_seq_kernel<float>, dim3(ew_grid(rows * C)), dim3(256), 0, pe_stream(stream), x, ldx, coff,
                     seq, rows, C);
  PE_LAUNCH_CHECK();
  return PE_OK;
}

extern "C" int pe_nhwc_to_seq_a16(const void* x, long ldx, int coff, float* seq, long rows, int C, void* stream) {
  if (!x || !seq || rows <= 0 || C <= 0) return PE_E_ARG;
  hipLaunchKernelGGL(nhwc_to_seq_kernel<act16_t>, dim3(ew_grid(rows * C)), dim3(256), 0, pe_stream(stream),
                     static_cast<const act16_t*>(x), ldx, coff, seq, rows, C);
  PE_LAUNCH_CHECK();
  return PE_OK;
}

extern "C" int pe_seq_to_nhwc(const float* seq, float* x, long ldx, int coff, long rows, int C, int accumulate,
                              void* stream) {
  if (!x || !seq || rows <= 0 || C <= 0) return PE_E_ARG;
  hipLaunchKernelGGL(seq_to_nhwc_kernel<float>, dim3(ew_grid(rows * C)), dim3(256), 0, pe_stream(stream), seq, x, ldx,
                     coff, rows, C, accumulate);
  PE_LAUNCH_CHECK();
  return PE_OK;
}

extern "C" int pe_seq_to_nhwc_a16(const float* seq, void* x, long ldx, int coff, long rows, int C, int accumulate,
                                  void* stream) {
  if (!x || !seq || rows <= 0 || C <= 0) return PE_E_ARG;
  hipLaunchKernelGGL(seq_to_nhwc_kernel<act16_t>, dim3(ew_grid(rows * C)), dim3(256), 0, pe_stream(stream), seq,
                     static_cast<act16_t*>(x), ldx, coff, rows, C, accumulate);
  PE_LAUNCH_CHECK();
  return PE_OK;
}

template <class TA>
static int copy2d_impl(const TA* src, long lds, TA* dst, long ldd, long rows, int cols, int accumulate, void* stream) {
  if (!src || !dst || rows <= 0 || cols <= 0) return PE_E_ARG;
  if ((cols & 3) || (lds & 3) || (ldd & 3)) return PE_E_UNSUPPORTED;
  hipLaunchKernelGGL(copy2d_kernel<TA>, dim3(ew_grid(rows * (cols / 4))), dim3(256), 0, pe_stream(stream), src, lds, dst,
                     ldd, rows, cols, accumulate);
  PE_LAUNCH_CHECK();
  return PE_OK;
}

extern "C" int pe_copy2d(const float* src, long lds, float* dst, long ldd, long rows, int cols, int accumulate,
                         void* stream) {
  return copy2d_impl<float>(src, lds, dst, ldd, rows, cols, accumulate, stream);
}

extern "C" int pe_copy2d_a16(const void* src, long lds, void* dst, long ldd, long rows, int cols, int accumulate,
                             void* stream) {
  return copy2d_impl<act16_t>(static_cast<const act16_t*>(src), lds, static_cast<act16_t*>(dst), ldd, rows, cols,
                              accumulate, stream);
}

extern "C" int pe_absmax(const float* x, long rows, int cols, long ld, unsigned* out, void* stream) {
  if (!x || !out || rows < 0 || cols <= 0 || ld < cols) return PE_E_ARG;
  if ((cols & 3) || (ld & 3) || (reinterpret_cast<uintptr_t>(x) & 15)) return PE_E_UNSUPPORTED;
  hipStream_t st = pe_stream(stream);
  PE_CHECK_HIP(hipMemsetAsync(out, 0, sizeof(unsigned), st));
  if (rows == 0) return PE_OK;
  const long n4 = rows * (cols / 4);
  const int grid = (int)(n4 / 2048 < 1 ? 1 : n4 / 2048 > 2048 ? 2048 : n4 / 2048);     // >= 8 float4 per thread
  hipLaunchKernelGGL(absmax_kernel, dim3(grid), dim3(256), 0, st, x, rows, cols / 4, ld, out);
  PE_LAUNCH_CHECK();
  return PE_OK;
}

extern "C" int pe_absmax_segments(const float* base, const long* seg_off, const long* seg_len, int nseg, unsigned* out,
                                  void* stream) {
  if (!base || !seg_off || !seg_len || !out || nseg <= 0 || nseg > 65535) return PE_E_ARG;
  hipStream_t st = pe_stream(stream);
  PE_CHECK_HIP(hipMemsetAsync(out, 0, (size_t)nseg * sizeof(unsigned), st));
  hipLaunchKernelGGL(absmax_segments_kernel, dim3(16, nseg), dim3(256), 0, st, base, seg_off, seg_len, out);
  PE_LAUNCH_CHECK();
  return PE_OK;
}
