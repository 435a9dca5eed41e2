// Output heads, losses and the fused AdamW step.
//   heads:  classifier Linear(D, 1) and detector Linear(D, 2) followed by the sum over its two
//           logits (model.py:67-70,96-98,115-117) as one-wave-per-row dot products;
//   loss:   lambda * SmoothL1(f0_pred, f0) + BCEWithLogits(sil_pred, sil), both mean-reduced over
//           every frame (train.py:104-106, trainer.py:237-239), with their gradients;
//   AdamW:  torch.optim.AdamW update (optimizers.py:55-62) over one flat parameter buffer.
#include "common.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// y[r] = sum over o < n_out of (x[r] . w[o] + b[o])          (n_out = 1: plain Linear(D,1))
__global__ __launch_bounds__(256) void head_fwd_kernel(const float* __restrict__ x, long ldx,
                                                       const float* __restrict__ w, const float* __restrict__ b,
                                                       int n_out, float* __restrict__ y, long R, int D) {
  const int lane = threadIdx.x & 63;
  const long wave = ((long)blockIdx.x * 256 + threadIdx.x) >> 6;
  const long nwaves = ((long)gridDim.x * 256) >> 6;
  for (long r = wave; r < R; r += nwaves) {
    const float* xr = x + r * ldx;
    float tot = 0.f;
    for (int o = 0; o < n_out; ++o) {
      float s = 0.f;
      for (int k = lane * 4; k < D; k += 256) {
        const float4 xv = *reinterpret_cast<const float4*>(xr + k);
        const float4 wv = *reinterpret_cast<const float4*>(w + (long)o * D + k);
        s = fmaf(xv.x, wv.x, s); s = fmaf(xv.y, wv.y, s); s = fmaf(xv.z, wv.z, s); s = fmaf(xv.w, wv.w, s);
      }
      s = wave_sum(s);
      tot += s + b[o];
    }
    if (lane == 0) y[r] = tot;
  }
}

// dx[r][:] = dy[r] * sum_o w[o][:]
__global__ __launch_bounds__(256) void head_bwd_dx_kernel(const float* __restrict__ w, int n_out,
                                                          const float* __restrict__ dy, float* __restrict__ dx,
                                                          long lddx, long R, int D) {
  const int quads = D >> 2;
  const long total = R * quads;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % quads) * 4;
    const long r = i / quads;
    float4 ws = *reinterpret_cast<const float4*>(w + c);
    for (int o = 1; o < n_out; ++o) {
      const float4 t = *reinterpret_cast<const float4*>(w + (long)o * D + c);
      ws.x += t.x; ws.y += t.y; ws.z += t.z; ws.w += t.w;
    }
    const float g = dy[r];
    *reinterpret_cast<float4*>(dx + r * lddx + c) = make_float4(g * ws.x, g * ws.y, g * ws.z, g * ws.w);
  }
}

// partial[z][c] = sum over this chunk's rows of dy[r] * x[r][c];  column D holds sum dy[r]
__global__ __launch_bounds__(256) void head_bwd_dw_partial_kernel(const float* __restrict__ x, long ldx,
                                                                  const float* __restrict__ dy, long R, int D,
                                                                  double* __restrict__ partial) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c > D) return;
  const long chunk = (R + gridDim.y - 1) / gridDim.y;
  const long r0 = (long)blockIdx.y * chunk;
  const long r1 = r0 + chunk < R ? r0 + chunk : R;
  double s = 0;
  if (c < D) for (long r = r0; r < r1; ++r) s += (double)(dy[r] * x[r * ldx + c]);
  else for (long r = r0; r < r1; ++r) s += (double)dy[r];
  partial[(long)blockIdx.y * (D + 1) + c] = s;
}

__global__ void head_bwd_dw_final_kernel(const double* __restrict__ partial, int nparts, int D, int n_out,
                                         float* __restrict__ dw, float* __restrict__ db) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);            // one wave per column
  if (c > D) return;
  const double s = pe_wave_strided_sum(partial + c, (long)(D + 1), nparts);
  if ((threadIdx.x & 63) != 0) return;
  for (int o = 0; o < n_out; ++o) {
    if (c < D) dw[(long)o * D + c] = (float)s;
    else db[o] = (float)s;
  }
}

// ------------------------------------------------------------------ 360-bin F0 classification loss
// Build-defined (the reference has no classification loss): CREPE-style targets
//   cents = 1200 log2(f0 / 10 Hz),  bin = clamp(rint((cents - 1997.3794084376191) / 20), 0, C - 1)
// for voiced frames (f0 > 0); unvoiced frames are ignored by the cross-entropy, which is averaged over
// the voiced frames (0 when there are none).  total = lambda * CE + BCEWithLogits(sil), as the
// regression loss above.  One wave per row; index arithmetic in double so it matches the float64 oracle.
constexpr double kCrepeCents0 = 1997.3794084376191;

__device__ __forceinline__ int f0_bin(float f0, int C) {
  const double cents = 1200.0 * log2((double)f0 / 10.0);
  double b = rint((cents - kCrepeCents0) / 20.0);
  b = b < 0.0 ? 0.0 : b;
  b = b > (double)(C - 1) ? (double)(C - 1) : b;
  return (int)b;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}
__device__ __forceinline__ float wave_sum_all(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// row_ce[r] = -log softmax(logits[r])[bin] (0 for unvoiced), row_lse[r] = log-sum-exp of the row
__global__ __launch_bounds__(256) void bins_ce_rows_kernel(const float* __restrict__ logits, long ldl, int C,
                                                           const float* __restrict__ f0, long R,
                                                           float* __restrict__ row_ce, float* __restrict__ row_lse) {
  const int lane = threadIdx.x & 63;
  const long wave = ((long)blockIdx.x * 256 + threadIdx.x) >> 6;
  const long nwaves = ((long)gridDim.x * 256) >> 6;
  for (long r = wave; r < R; r += nwaves) {
    const float* lr = logits + r * ldl;
    float m = -INFINITY;
    for (int c = lane; c < C; c += 64) m = fmaxf(m, lr[c]);
    m = wave_max(m);
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += expf(lr[c] - m);
    s = wave_sum_all(s);
    const float lse = m + logf(s);
    if (lane == 0) {
      const float f = f0[r];
      row_lse[r] = lse;
      row_ce[r] = f > 0.f ? lse - lr[f0_bin(f, C)] : 0.f;
    }
  }
}

// out[0] = total, out[1] = lambda * ce, out[2] = bce, out[3] = voiced frame count; d_sil as in f0_sil_loss
__global__ __launch_bounds__(1024) void bins_ce_reduce_kernel(const float* __restrict__ row_ce,
                                                              const float* __restrict__ f0,
                                                              const float* __restrict__ sil_pred,
                                                              const float* __restrict__ sil, float lambda_f0, long R,
                                                              float grad_scale, float* __restrict__ out,
                                                              float* __restrict__ d_sil) {
  __shared__ double red[3][1024];
  double s1 = 0, s2 = 0, cnt = 0;
  const float inv_n = 1.0f / (float)R;
  for (long r = threadIdx.x; r < R; r += 1024) {
    s1 += row_ce[r];
    cnt += f0[r] > 0.f ? 1.0 : 0.0;
    const float z = sil_pred[r], y = sil[r];
    s2 += fmaxf(z, 0.f) - z * y + log1pf(expf(-fabsf(z)));
    if (d_sil) d_sil[r] = grad_scale * inv_n * (1.0f / (1.0f + expf(-z)) - y);
  }
  red[0][threadIdx.x] = s1; red[1][threadIdx.x] = s2; red[2][threadIdx.x] = cnt;
  __syncthreads();
  for (int s = 512; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s)
      for (int k = 0; k < 3; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double n_voiced = red[2][0];
    const float ce = n_voiced > 0 ? (float)(red[0][0] / n_voiced) : 0.f;
    const float bce = (float)(red[1][0] / (double)R);
    out[1] = lambda_f0 * ce;
    out[2] = bce;
    out[0] = lambda_f0 * ce + bce;
    out[3] = (float)n_voiced;
  }
}

// d_logits[r][c] = grad_scale * lambda / n_voiced * (softmax - onehot) on voiced rows, 0 elsewhere
__global__ __launch_bounds__(256) void bins_ce_grad_kernel(const float* __restrict__ logits, long ldl, int C,
                                                           const float* __restrict__ f0,
                                                           const float* __restrict__ row_lse,
                                                           const float* __restrict__ out, float lambda_f0,
                                                           float grad_scale, long R, float* __restrict__ d_logits,
                                                           long ldd) {
  const float nv = out[3];
  const float k = nv > 0.f ? grad_scale * lambda_f0 / nv : 0.f;
  const long total = R * C;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / C;
    const int c = (int)(i - r * C);
    const float f = f0[r];
    float g = 0.f;
    if (f > 0.f) g = k * (expf(logits[r * ldl + c] - row_lse[r]) - (c == f0_bin(f, C) ? 1.f : 0.f));
    d_logits[r * ldd + c] = g;
  }
}

constexpr int kHeadParts = 256;

// ------------------------------------------------------------------ losses
// out[0] = total, out[1] = lambda * smooth_l1, out[2] = bce;  gradients pre-multiplied by grad_scale.
__global__ __launch_bounds__(1024) void f0_sil_loss_kernel(const float* __restrict__ f0_pred,
                                                           const float* __restrict__ f0,
                                                           const float* __restrict__ sil_pred,
                                                           const float* __restrict__ sil, float lambda_f0, long R,
                                                           float grad_scale, float* __restrict__ out,
                                                           float* __restrict__ d_f0, float* __restrict__ d_sil) {
  __shared__ double red[2][1024];
  double s1 = 0, s2 = 0;
  const float inv_n = 1.0f / (float)R;
  for (long r = threadIdx.x; r < R; r += 1024) {
    const float d = f0_pred[r] - f0[r];
    const float ad = fabsf(d);
    s1 += ad < 1.f ? 0.5f * d * d : ad - 0.5f;
    const float z = sil_pred[r], y = sil[r];
    s2 += fmaxf(z, 0.f) - z * y + log1pf(expf(-fabsf(z)));
    if (d_f0) d_f0[r] = grad_scale * lambda_f0 * inv_n * (ad < 1.f ? d : (d > 0.f ? 1.f : -1.f));
    if (d_sil) d_sil[r] = grad_scale * inv_n * (1.0f / (1.0f + expf(-z)) - y);
  }
  red[0][threadIdx.x] = s1;
  red[1][threadIdx.x] = s2;
  __syncthreads();
  for (int s = 512; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      red[0][threadIdx.x] += red[0][threadIdx.x + s];
      red[1][threadIdx.x] += red[1][threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float l1 = (float)(red[0][0] / (double)R);
    const float bce = (float)(red[1][0] / (double)R);
    const float lf0 = lambda_f0 * l1;
    out[1] = lf0;
    out[2] = bce;
    out[0] = lf0 + bce;
  }
}

// ------------------------------------------------------------------ AdamW
struct AdamArgs {
  float lr, beta1, beta2, eps, weight_decay, step_size, bc2_sqrt, grad_scale;
};

// GradScaler's inf / nan test (reference trainer.py:241-244 -> torch.amp.GradScaler.step): flag = 1 if any element
// of the flat gradient buffer is not finite.  Every thread that sees one stores the same value: no atomics needed.
__global__ __launch_bounds__(256) void nonfinite_kernel(const float* __restrict__ g, long n, int* __restrict__ flag) {
  const long n4 = n >> 2;
  bool bad = false;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const float4 v = reinterpret_cast<const float4*>(g)[i];
    // exponent all ones <=> inf or nan
    bad |= ((__float_as_uint(v.x) & 0x7f800000u) == 0x7f800000u) | ((__float_as_uint(v.y) & 0x7f800000u) == 0x7f800000u) |
           ((__float_as_uint(v.z) & 0x7f800000u) == 0x7f800000u) | ((__float_as_uint(v.w) & 0x7f800000u) == 0x7f800000u);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3))
    bad |= (__float_as_uint(g[(n4 << 2) + threadIdx.x]) & 0x7f800000u) == 0x7f800000u;
  if (bad) *flag = 1;
}

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v, long n,
                                                    const AdamArgs a, const float* __restrict__ skip) {
  // device-side predicate: a step whose status word is non-zero (a persistent-LSTM hand-off timed out on some rank)
  // leaves parameters and moments untouched; the host learns of it with the loss scalars and redoes the step
  if (skip != nullptr && *skip != 0.0f) return;
  const long n4 = n >> 2;
  const float decay = 1.0f - a.lr * a.weight_decay;
  const float w1 = 1.0f - a.beta1, w2 = 1.0f - a.beta2;
  auto upd = [&](float& pp, float gg, float& mm, float& vv) {
    gg *= a.grad_scale;
    pp *= decay;
    mm = mm + w1 * (gg - mm);                       // exp_avg.lerp_(grad, 1 - beta1)
    vv = vv * a.beta2 + w2 * (gg * gg);             // exp_avg_sq.mul_(beta2).addcmul_(g, g, 1 - beta2)
    const float denom = sqrtf(vv) / a.bc2_sqrt + a.eps;
    pp = pp - a.step_size * (mm / denom);
  };
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    float4 pv = reinterpret_cast<float4*>(p)[i];
    const float4 gv = reinterpret_cast<const float4*>(g)[i];
    float4 mv = reinterpret_cast<float4*>(m)[i];
    float4 vv = reinterpret_cast<float4*>(v)[i];
    upd(pv.x, gv.x, mv.x, vv.x); upd(pv.y, gv.y, mv.y, vv.y);
    upd(pv.z, gv.z, mv.z, vv.z); upd(pv.w, gv.w, mv.w, vv.w);
    reinterpret_cast<float4*>(p)[i] = pv;
    reinterpret_cast<float4*>(m)[i] = mv;
    reinterpret_cast<float4*>(v)[i] = vv;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const long i = (n4 << 2) + threadIdx.x;
    upd(p[i], g[i], m[i], v[i]);
  }
}

}  // namespace

extern "C" int pe_head_fwd(const float* x, long ldx, const float* w, const float* bias, int n_out, float* y,
                           long R, int D, void* stream) {
  if (!x || !w || !bias || !y || R <= 0 || D <= 0 || n_out < 1 || n_out > 8) return PE_E_ARG;
  if ((D & 3) || (ldx & 3)) return PE_E_UNSUPPORTED;
  long grid = (R + 3) / 4;
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(head_fwd_kernel, dim3((int)grid), dim3(256), 0, pe_stream(stream), x, ldx, w, bias, n_out, y, R,
                     D);
  PE_LAUNCH_CHECK();
  return PE_OK;
}

extern "C" size_t pe_head_bwd_workspace_bytes(int D) { return (size_t)kHeadParts * (D + 1) * sizeof(double); }

extern "C" int pe_head_bwd(const float* x, long ldx, const float* w, const float* dy, int n_out, float* dx,
                           long lddx, float* dw, float* db, long R, int D, void* workspace, size_t workspace_bytes,
                           void* stream) {
  if (!x || !w || !dy || !dx || !dw || !db || R <= 0 || D <= 0 || n_out < 1 || n_out > 8) return PE_E_ARG;
  if ((D & 3) || (ldx & 3) || (lddx & 3)) return PE_E_UNSUPPORTED;
  if (!workspace || workspace_bytes < pe_head_bwd_workspace_bytes(D)) return PE_E_WORKSPACE;
  hipStream_t st = pe_stream(stream);
  long g = (R * (D / 4) + 255) / 256;
  if (g > 16384) g = 16384;
  hipLaunchKernelGGL(head_bwd_dx_kernel, dim3((int)g), dim3(256), 0, st, w, n_out, dy, dx, lddx, R, D);
  PE_LAUNCH_CHECK();
  double* partial = reinterpret_cast<double*>(workspace);
  const int parts = R < kHeadParts ? (int)R : kHeadParts;
  hipLaunchKernelGGL(head_bwd_dw_partial_kernel, dim3(pe_cdiv(D + 1, 256), parts), dim3(256), 0, st, x, ldx, dy, R, D,
                     partial);
  PE_LAUNCH_CHECK();
  hipLaunchKernelGGL(head_bwd_dw_final_kernel, dim3(pe_cdiv(D + 1, 4)), dim3(256), 0, st, partial, parts, D, n_out,
                     dw, db);
  PE_LAUNCH_CHECK();
  return PE_OK;
}

extern "C" int pe_f0_sil_loss(const float* f0_pred, const float* f0, const float* sil_pred, const float* sil,
                              float lambda_f0, long R, float grad_scale, float* out3, float* d_f0_pred,
                              float* d_sil_pred, void* stream) {
  if (!f0_pred || !f0 || !sil_pred || !sil || !out3 || R <= 0) return PE_E_ARG;
  hipLaunchKernelGGL(f0_sil_loss_kernel, dim3(1), dim3(1024), 0, pe_stream(stream), f0_pred, f0, sil_pred, sil,
                     lambda_f0, R, grad_scale, out3, d_f0_pred, d_sil_pred);
  PE_LAUNCH_CHECK();
  return PE_OK;
}

extern "C" size_t pe_f0_bins_ce_workspace_bytes(long R) { return (size_t)(2 * R) * sizeof(float); }

extern "C" int pe_f0_bins_ce_loss(const float* logits, long ldl, int C, const float* f0, const float* sil_pred,
                                  const float* sil, float lambda_f0, long R, float grad_scale, float* out4,
                                  float* d_logits, long ldd, float* d_sil_pred, float* workspace,
                                  size_t workspace_bytes, void* stream) {
  if (!logits || !f0 || !sil_pred || !sil || !out4 || R <= 0 || C < 2 || ldl < C) return PE_E_ARG;
  if (d_logits && ldd < C) return PE_E_ARG;
  if (!workspace || workspace_bytes < pe_f0_bins_ce_workspace_bytes(R)) return PE_E_WORKSPACE;
  hipStream_t st = pe_stream(stream);
  float* row_ce = workspace;
  float* row_lse = workspace + R;
  long g = (R + 3) / 4;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(bins_ce_rows_kernel, dim3((int)g), dim3(256), 0, st, logits, ldl, C, f0, R, row_ce, row_lse);
  PE_LAUNCH_CHECK();
  hipLaunchKernelGGL(bins_ce_reduce_kernel, dim3(1), dim3(1024), 0, st, row_ce, f0, sil_pred, sil, lambda_f0, R,
                     grad_scale, out4, d_sil_pred);
  PE_LAUNCH_CHECK();
  if (d_logits) {
    long gg = (R * C + 255) / 256;
    if (gg > 8192) gg = 8192;
    hipLaunchKernelGGL(bins_ce_grad_kernel, dim3((int)gg), dim3(256), 0, st, logits, ldl, C, f0, row_lse, out4,
                       lambda_f0, grad_scale, R, d_logits, ldd);
    PE_LAUNCH_CHECK();
  }
  return PE_OK;
}

extern "C" int pe_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long n, float lr,
                             float beta1, float beta2, float eps, float weight_decay, double bias_correction1,
                             double bias_correction2, float grad_scale, const float* skip_if_nonzero, void* stream) {
  if (!param || !grad || !exp_avg || !exp_avg_sq || n <= 0) return PE_E_ARG;
  if (((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) return PE_E_UNSUPPORTED;
  AdamArgs a;
  a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.weight_decay = weight_decay;
  a.step_size = (float)((double)lr / bias_correction1);
  a.bc2_sqrt = (float)sqrt(bias_correction2);
  a.grad_scale = grad_scale;
  long g = ((n >> 2) + 255) / 256;
  if (g > 8192) g = 8192;
  if (g < 1) g = 1;
  hipLaunchKernelGGL(adamw_kernel, dim3((int)g), dim3(256), 0, pe_stream(stream), param, grad, exp_avg, exp_avg_sq, n,
                     a, skip_if_nonzero);
  PE_LAUNCH_CHECK();
  return PE_OK;
}

extern "C" int pe_nonfinite_flag(const float* x, long n, int* flag, void* stream) {
  if (!x || !flag || n < 0) return PE_E_ARG;
  if (reinterpret_cast<uintptr_t>(x) & 15) return PE_E_UNSUPPORTED;
  hipStream_t st = pe_stream(stream);
  PE_CHECK_HIP(hipMemsetAsync(flag, 0, sizeof(int), st));
  if (n == 0) return PE_OK;
  long g = ((n >> 2) + 255) / 256;
  if (g > 8192) g = 8192;
  if (g < 1) g = 1;
  hipLaunchKernelGGL(nonfinite_kernel, dim3((int)g), dim3(256), 0, st, x, n, flag);
  PE_LAUNCH_CHECK();
  return PE_OK;
}
