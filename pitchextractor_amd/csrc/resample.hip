// Polyphase sinc resampler on the GPU: the arithmetic of
// torchaudio.functional.resample(x, orig, new) with its defaults -- sinc_interp_hann,
// lowpass_filter_width = 6, rolloff = 0.99 -- as the reference calls it at meldataset.py:621-627
// (44 100 -> 24 000 Hz reduces to 147 -> 80: 80 phases x 171 taps, output length ceil(80 L / 147)).
// One thread per output sample; the 54 KB tap table stays in L1/L2.
#include <math.h>
#include <vector>
#include "common.h"

struct pe_resample_plan {
  int orig, neu, width, taps;
  float* d_kernel;          // [neu][taps]
};

namespace {

__global__ __launch_bounds__(256) void resample_kernel(const float* __restrict__ x, long x_stride, int n_in,
                                                       float* __restrict__ y, long y_stride, int n_out,
                                                       const float* __restrict__ k, int orig, int neu, int width,
                                                       int taps) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= n_out) return;
  const float* xb = x + (long)blockIdx.y * x_stride;
  const int j = n / neu, p = n - j * neu;
  const float* kp = k + (long)p * taps;
  const int base = j * orig - width;
  float acc = 0.f;
  for (int i = 0; i < taps; ++i) {
    const int q = base + i;
    const float v = (q >= 0 && q < n_in) ? xb[q] : 0.f;
    acc = fmaf(kp[i], v, acc);
  }
  y[(long)blockIdx.y * y_stride + n] = acc;
}

int gcd_i(int a, int b) { while (b) { const int t = a % b; a = b; b = t; } return a; }

}  // namespace

extern "C" int pe_resample_plan_create(pe_resample_plan** plan_out, int orig_freq, int new_freq,
                                       int lowpass_filter_width, float rolloff) {
  if (!plan_out || orig_freq <= 0 || new_freq <= 0 || lowpass_filter_width <= 0 || !(rolloff > 0.f)) return PE_E_ARG;
  const int g = gcd_i(orig_freq, new_freq);
  const int orig = orig_freq / g, neu = new_freq / g;
  const double base_freq = (double)(orig < neu ? orig : neu) * (double)rolloff;
  const int width = (int)ceil((double)lowpass_filter_width * orig / base_freq);
  const int taps = 2 * width + orig;
  std::vector<float> k((size_t)neu * taps);
  const double scale = base_freq / orig;
  for (int p = 0; p < neu; ++p)
    for (int i = 0; i < taps; ++i) {
      double t = ((double)(-p) / neu + (double)(i - width) / orig) * base_freq;
      if (t < -lowpass_filter_width) t = -lowpass_filter_width;
      if (t > lowpass_filter_width) t = lowpass_filter_width;
      const double c = cos(t * M_PI / lowpass_filter_width / 2.0);
      const double window = c * c;
      const double tp = t * M_PI;
      const double sinc = (tp == 0.0) ? 1.0 : sin(tp) / tp;
      k[(size_t)p * taps + i] = (float)(sinc * window * scale);
    }
  float* d = nullptr;
  PE_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&d), k.size() * sizeof(float)));
  hipError_t e = hipMemcpy(d, k.data(), k.size() * sizeof(float), hipMemcpyHostToDevice);
  if (e != hipSuccess) { (void)hipFree(d); return (int)e; }
  pe_resample_plan* pl = new pe_resample_plan{orig, neu, width, taps, d};
  *plan_out = pl;
  return PE_OK;
}

extern "C" int pe_resample_plan_destroy(pe_resample_plan* plan) {
  if (!plan) return PE_E_ARG;
  hipError_t e = hipFree(plan->d_kernel);
  delete plan;
  return (int)e;
}

/* ceil(new * n_in / orig) with the reduced ratio */
extern "C" long pe_resample_out_len(const pe_resample_plan* plan, long n_in) {
  if (!plan || n_in < 0) return PE_E_ARG;
  return (n_in * plan->neu + plan->orig - 1) / plan->orig;
}

extern "C" int pe_resample_forward(const pe_resample_plan* plan, const float* x, int batch, int n_in, long x_stride,
                                   float* y, long y_stride, int n_out, void* stream) {
  if (!plan || !x || !y || batch < 0 || n_in < 0 || n_out < 0 || x_stride < n_in || y_stride < n_out) return PE_E_ARG;
  if (batch == 0 || n_out == 0) return PE_OK;
  if (batch > 65535) return PE_E_UNSUPPORTED;
  if ((long)n_out > pe_resample_out_len(plan, n_in)) return PE_E_ARG;
  hipLaunchKernelGGL(resample_kernel, dim3(pe_cdiv(n_out, 256), batch), dim3(256), 0, pe_stream(stream), x, x_stride,
                     n_in, y, y_stride, n_out, plan->d_kernel, plan->orig, plan->neu, plan->width, plan->taps);
  PE_LAUNCH_CHECK();
  return PE_OK;
}
