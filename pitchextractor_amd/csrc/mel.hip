// Fused mel front end for gfx950: reflect-pad framing + periodic Hann + 1024-pt
// real FFT + |.|^2 + sparse HTK mel filterbank + log/normalise, batched.
//
// Replaces torchaudio.transforms.MelSpectrogram as used by the reference at
// meldataset.py:34-40,58-77,644 and the log/affine of meldataset.py:650.
//
// Work decomposition: a 256-thread workgroup walks items of FB consecutive frames
// of one utterance.  Each of the 4 waves transforms one frame at a time, read
// straight from global memory in 512-byte runs (reflect indices resolved per
// sample on edge frames only): the 1024 real samples are packed as 512 complex
// points, 8 per lane, and run through three in-register radix-8 passes
// (512 = 8*8*8) with two LDS exchanges between them; the real-FFT split, the
// power and the (<= 2 non-zero weights per bin) mel filterbank follow from LDS.
// Window and twiddles are per-lane constants held in registers for the whole launch.
#include <math.h>
#include <string.h>
#include <vector>
#include "common.h"

namespace {

constexpr int kNfft = 1024;
constexpr int kHalf = 512;     // complex FFT length
constexpr int kXchg = 576;     // float2 slots per exchange region (8*72 = 64*9)
constexpr int kMaxParts = 6;   // 8-tap chunks per mel filter (filters up to 48 bins wide)

struct MelArgs {
  const float* wave;
  long wave_stride;
  int n_samples;
  int hop;
  int n_mels;
  int n_valid;                 // 1 + n_samples / hop
  int out_frames;
  float* out;
  long out_sb, out_sm, out_st;
  int log_mode;
  float log_eps, mean, inv_std, pad_value;
  const float* win;            // [1024]
  const float2* tw512;         // [512]  exp(-2 pi i k / 512)
  const float2* tw1024;        // [257]  exp(-2 pi i k / 1024)
  const int* fb_idx;           // [n_pairs] first FFT bin of each 8-tap chunk | [n_mels] first chunk | [n_mels] #chunks
  const float* fb_w;           // [n_pairs][8] chunk weights, zero padded
  int n_pairs;
  int audio_len;               // (FB-1)*hop + 1024, padded to a multiple of 4
  const int* n_samples_arr;    // optional [batch]: per-utterance length (ragged batches)
  const int* frame_start;      // optional [batch]: output frame t shows source frame t + frame_start[b]
};

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
// multiply by -i
__device__ __forceinline__ float2 mul_mi(float2 a) { return make_float2(a.y, -a.x); }

__device__ __forceinline__ void fft4(const float2 u0, const float2 u1, const float2 u2, const float2 u3,
                                     float2& o0, float2& o1, float2& o2, float2& o3) {
  const float2 t0 = cadd(u0, u2), t1 = csub(u0, u2), t2 = cadd(u1, u3), t3 = mul_mi(csub(u1, u3));
  o0 = cadd(t0, t2); o1 = cadd(t1, t3); o2 = csub(t0, t2); o3 = csub(t1, t3);
}

// In-place 8-point DFT, natural order in and out: v[k] = sum_a v[a] exp(-2 pi i a k / 8).
__device__ __forceinline__ void fft8(float2 (&v)[8]) {
  constexpr float c = 0.70710678118654752440f;
  const float2 s0 = cadd(v[0], v[4]), s1 = cadd(v[1], v[5]), s2 = cadd(v[2], v[6]), s3 = cadd(v[3], v[7]);
  const float2 e0 = csub(v[0], v[4]), e1 = csub(v[1], v[5]), e2 = csub(v[2], v[6]), e3 = csub(v[3], v[7]);
  const float2 d0 = e0;
  const float2 d1 = make_float2(c * (e1.x + e1.y), c * (e1.y - e1.x));
  const float2 d2 = mul_mi(e2);
  const float2 d3 = make_float2(c * (e3.y - e3.x), -c * (e3.x + e3.y));
  fft4(s0, s1, s2, s3, v[0], v[2], v[4], v[6]);
  fft4(d0, d1, d2, d3, v[1], v[3], v[5], v[7]);
}

// The 4 waves of a workgroup share only read-only LDS (audio chunk, filterbank); the exchange
// regions A/B/P are private to a wave, and one wave's DS instructions execute in program order, so
// the passes below need no s_barrier -- only a compiler fence that keeps LDS writes ahead of the
// dependent LDS reads (waves then drift freely instead of marching in lockstep).
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Persistent form: the grid is sized to the chip (4 workgroups per CU fit: 29 KB of LDS, 126 VGPRs) and every
// workgroup walks (utterance, block of FB frames) items, so the per-lane window / twiddle constants are loaded
// once per workgroup, not once per 16 frames.  Frames are read straight from global memory: lane l takes the
// float2 at complex index 64 r + l, so every load instruction is one contiguous 512-byte run; the 3.4x overlap
// of consecutive frames is absorbed by L2 (HBM sees each sample once), and no staging barrier is needed.
// The three radix-8 passes exchange through ONE LDS region per wave: within a phase every lane first reads
// its 8 values, then writes 8, and a wave's LDS instructions execute in program order, so the region can be
// overwritten in place.
// Work split: every utterance is cut into `parts` contiguous frame ranges of `span` frames (a multiple of 4, chosen
// by the host so that parts * batch fills the resident grid and the VALID frames divide evenly: 161 frames over 4
// workgroups = 41 each, where a fixed 16-frame block grid would give one workgroup 48 and another 32); a workgroup
// walks its range in chunks of FB frames.  Frames >= the utterance's frame count are written as padding.
template <int FB>
__global__ __launch_bounds__(256, 4) void mel_fwd_kernel(const MelArgs a, int parts, int span, int n_items) {
  static_assert(FB % 4 == 0, "FB frames are dealt to 4 waves");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float2* s_X = reinterpret_cast<float2*>(smem);                     // [4][kXchg] exchange / Z / P
  float* s_part = reinterpret_cast<float*>(s_X + 4 * kXchg);         // [4][256] chunk sums
  float* s_out = s_part + 4 * 256;                                   // [n_mels][FB+1]
  float* s_fbw = s_out + ((a.n_mels * (FB + 1) + 3) & ~3);           // [n_pairs][8], 16-byte aligned
  int* s_fbi = reinterpret_cast<int*>(s_fbw + a.n_pairs * 8);        // k0[n_pairs] | first[n_mels] | cnt[n_mels]

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int hop = a.hop, n_mels = a.n_mels;
  for (int j = tid; j < a.n_pairs * 8; j += 256) s_fbw[j] = a.fb_w[j];
  for (int j = tid; j < a.n_pairs + 2 * n_mels; j += 256) s_fbi[j] = a.fb_idx[j];

  // ---- per-lane constants, kept in registers across all items of this workgroup
  float2 win[8], tw1[8], tw2[8];
  const int c_ = lane & 7;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int n = 64 * r + lane;                        // complex sample index, stage 1
    win[r] = make_float2(a.win[2 * n], a.win[2 * n + 1]);
    tw1[r] = a.tw512[(lane * r) & 511];                 // W512^(m*k0)
    tw2[r] = a.tw512[(8 * c_ * r) & 511];               // W64^(c*k1)
  }
  const float2* twp = a.tw1024 + lane;                  // W1024^k of the real-FFT split: 4 cached loads per frame
  float2* X = s_X + wv * kXchg;
  float* P = reinterpret_cast<float*>(X);               // 513 (+7 zero) floats, after the real-FFT split
  float* part = s_part + wv * 256;
  __syncthreads();

  for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
   const int b = item / parts, prt = item - b * parts;
   // ranges [prt * span, +span) cover the first parts * span output frames; the last one also takes the rest
   const int r_lo = prt * span, r_hi = prt + 1 == parts ? a.out_frames : min(a.out_frames, r_lo + span);
   for (int f_out0 = r_lo; f_out0 < r_hi; f_out0 += FB) {      // first output frame of this chunk
    const int n_out = min(FB, r_hi - f_out0);
    const int n_samples = a.n_samples_arr ? a.n_samples_arr[b] : a.n_samples;
    // reflect padding needs more than n_fft/2 samples; shorter items come out as padding only
    const int n_valid = n_samples > kHalf ? 1 + n_samples / hop : 0;
    const int f0 = f_out0 + (a.frame_start ? max(a.frame_start[b], 0) : 0);   // first source frame
    const float* wsrc = a.wave + (long)b * a.wave_stride;
    const long N = n_samples;

    if (f0 < n_valid) {
      for (int it = 0; it < FB / 4; ++it) {
        const int fi = it * 4 + wv;
        const bool valid = fi < n_out && (f0 + fi) < n_valid;   // wave-uniform
        float2 v[8];
        if (valid) {
          // pass 1: radix-8 over a, lane = m = 8b+c, x[n = 64a + m]
          const long base = (long)(f0 + fi) * hop - kHalf;      // first sample of the frame (may be < 0)
          if (base >= 0 && base + kNfft <= N && ((base | a.wave_stride) & 1) == 0) {
            const float2* fr = reinterpret_cast<const float2*>(wsrc + base);
#pragma unroll
            for (int r = 0; r < 8; ++r) {
              const float2 x = fr[64 * r + lane];
              v[r] = make_float2(x.x * win[r].x, x.y * win[r].y);
            }
          } else {                                      // edge frame / odd alignment: reflect indices per sample
#pragma unroll
            for (int r = 0; r < 8; ++r) {
              float x2[2];
#pragma unroll
              for (int e = 0; e < 2; ++e) {
                long g = base + 2 * (64 * r + lane) + e;
                if (g < 0) g = -g;
                if (g >= N) g = 2 * (N - 1) - g;
                x2[e] = (g >= 0 && g < N) ? wsrc[g] : 0.0f;
              }
              v[r] = make_float2(x2[0] * win[r].x, x2[1] * win[r].y);
            }
          }
          fft8(v);
#pragma unroll
          for (int r = 0; r < 8; ++r) X[r * 72 + lane] = (r == 0) ? v[0] : cmul(v[r], tw1[r]);
        }
        wave_lds_sync();
        if (valid) {
          // pass 2: lane = 8*k0 + c, radix-8 over b (all 8 reads precede the writes: in place)
          const int k0 = lane >> 3;
#pragma unroll
          for (int r = 0; r < 8; ++r) v[r] = X[k0 * 72 + 8 * r + c_];
          fft8(v);
          wave_lds_sync();
#pragma unroll
          for (int r = 0; r < 8; ++r) X[(k0 + 8 * r) * 9 + c_] = (r == 0) ? v[0] : cmul(v[r], tw2[r]);
        }
        wave_lds_sync();
        if (valid) {
          // pass 3: lane = k0 + 8*k1, radix-8 over c -> Z[lane + 64*k2]
#pragma unroll
          for (int r = 0; r < 8; ++r) v[r] = X[lane * 9 + r];
          fft8(v);
          wave_lds_sync();
#pragma unroll
          for (int r = 0; r < 8; ++r) X[lane + 64 * r] = v[r];
        }
        wave_lds_sync();
        if (valid) {
          // real-FFT split: X[k] = E + W^k O, X[512-k] = conj(E - W^k O); keep powers (P overwrites Z in place)
          float2 zk[4], zn[4], tp[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int k = lane + 64 * j;
            tp[j] = twp[64 * j];
            zk[j] = X[k];
            zn[j] = X[(kHalf - k) & 511];
          }
          const float2 z256 = X[256];
          wave_lds_sync();
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int k = lane + 64 * j;
            const float2 E = make_float2(0.5f * (zk[j].x + zn[j].x), 0.5f * (zk[j].y - zn[j].y));
            const float2 O = make_float2(0.5f * (zk[j].y + zn[j].y), 0.5f * (zn[j].x - zk[j].x));
            const float2 wo = cmul(tp[j], O);
            const float2 xp = cadd(E, wo), xm = csub(E, wo);
            P[k] = xp.x * xp.x + xp.y * xp.y;
            P[kNfft / 2 - k] = xm.x * xm.x + xm.y * xm.y;
          }
          if (lane == 0) {
            P[256] = z256.x * z256.x + z256.y * z256.y;     // k = 256: W = -i, |X|^2 = |Z|^2
          } else if (lane < 8) {
            P[kNfft / 2 + lane] = 0.0f;                     // zero tail read by the padded 8-tap chunks
          }
        }
        wave_lds_sync();
        // mel filterbank, two balanced rounds: every filter is cut into chunks of 8 taps (zero padded);
        // a lane sums one chunk with a fixed, fully unrolled trip count, then each mel bin adds its
        // <= kMaxParts chunk sums.
        if (valid) {
          for (int id = lane; id < a.n_pairs; id += 64) {
            const int k0 = s_fbi[id];
            const float4 w0 = *reinterpret_cast<const float4*>(s_fbw + id * 8);
            const float4 w1 = *reinterpret_cast<const float4*>(s_fbw + id * 8 + 4);
            float acc = w0.x * P[k0];
            acc = fmaf(w0.y, P[k0 + 1], acc); acc = fmaf(w0.z, P[k0 + 2], acc); acc = fmaf(w0.w, P[k0 + 3], acc);
            acc = fmaf(w1.x, P[k0 + 4], acc); acc = fmaf(w1.y, P[k0 + 5], acc); acc = fmaf(w1.z, P[k0 + 6], acc);
            acc = fmaf(w1.w, P[k0 + 7], acc);
            part[id] = acc;
          }
        }
        wave_lds_sync();
        if (valid) {
          for (int m = lane; m < n_mels; m += 64) {
            const int first = s_fbi[a.n_pairs + m], cnt = s_fbi[a.n_pairs + n_mels + m];
            float acc = 0.0f;
#pragma unroll
            for (int j = 0; j < kMaxParts; ++j) acc += (j < cnt) ? part[first + j] : 0.0f;
            s_out[m * (FB + 1) + fi] = a.log_mode ? (logf(a.log_eps + acc) - a.mean) * a.inv_std : acc;
          }
        }
        wave_lds_sync();                                // P / part are rewritten by the next frame
      }
    }
    __syncthreads();

    // ---- store (frames past the utterance's last frame are padding)
    float* dst = a.out + (long)b * a.out_sb;
    const bool mel_fastest = (a.out_sm == 1);
    for (int idx = tid; idx < n_mels * FB; idx += 256) {
      const int m = mel_fastest ? idx % n_mels : idx / FB;
      const int fo = mel_fastest ? idx / n_mels : idx % FB;
      const int frame = f_out0 + fo;
      if (fo < n_out) {
        const float val = (f0 + fo < n_valid) ? s_out[m * (FB + 1) + fo] : a.pad_value;
        dst[(long)m * a.out_sm + (long)frame * a.out_st] = val;
      }
    }
    __syncthreads();                                    // s_out is rewritten by the next chunk
   }
  }
}

constexpr int kFB = 16;

int mel_cus() {
  static int cus = 0;
  if (cus <= 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
              ? prop.multiProcessorCount : 256;
  }
  return cus;
}

size_t mel_lds_bytes(int n_mels, int n_pairs, int fb) {
  size_t bytes = (size_t)4 * kXchg * 8 + 4 * 256 * 4 + (size_t)((n_mels * (fb + 1) + 3) & ~3) * 4 +
                 (size_t)n_pairs * 8 * 4 + (size_t)(n_pairs + 2 * n_mels) * 4;
  return (bytes + 15) & ~(size_t)15;
}

}  // namespace

struct pe_mel_plan {
  int sample_rate, n_fft, hop, n_mels, n_freq, n_pairs;
  void* d_base;
  float* d_win;
  float2* d_tw512;
  float2* d_tw1024;
  int* d_fb_idx;
  float* d_fb_w;
};

extern "C" int pe_mel_plan_create(pe_mel_plan** plan_out, int sample_rate, int n_fft, int win_length,
                                  int hop_length, int n_mels, float f_min, float f_max) {
  if (!plan_out || sample_rate <= 0 || hop_length <= 0 || n_mels <= 0) return PE_E_ARG;
  if (n_fft != kNfft || win_length != kNfft) return PE_E_UNSUPPORTED;
  if (!(f_max > f_min) || f_min < 0.0f) return PE_E_ARG;
  const int n_freq = n_fft / 2 + 1;

  // torchaudio.functional.melscale_fbanks(n_freqs, f_min, f_max, n_mels, sr, norm=None, "htk")
  std::vector<double> f_pts(n_mels + 2);
  const double m_min = 2595.0 * log10(1.0 + (double)f_min / 700.0);
  const double m_max = 2595.0 * log10(1.0 + (double)f_max / 700.0);
  for (int i = 0; i < n_mels + 2; ++i) {
    const double m = m_min + (m_max - m_min) * (double)i / (double)(n_mels + 1);
    f_pts[i] = 700.0 * (pow(10.0, m / 2595.0) - 1.0);
  }
  // sparse filterbank as 8-tap chunks: pair p covers bins [k0[p], k0[p] + 8) of one filter
  std::vector<int> pair_k0, mel_first(n_mels), mel_cnt(n_mels);
  std::vector<float> w;
  for (int m = 0; m < n_mels; ++m) {
    int first = -1, last = -1;
    std::vector<float> col(n_freq);
    for (int k = 0; k < n_freq; ++k) {
      const double f = (double)(sample_rate / 2) * (double)k / (double)(n_freq - 1);
      const double down = (f - f_pts[m]) / (f_pts[m + 1] - f_pts[m]);
      const double up = (f_pts[m + 2] - f) / (f_pts[m + 2] - f_pts[m + 1]);
      const double v = fmax(0.0, fmin(down, up));
      col[k] = (float)v;
      if (col[k] != 0.0f) { if (first < 0) first = k; last = k; }
    }
    mel_first[m] = (int)pair_k0.size();
    const int len = first < 0 ? 0 : last - first + 1;
    mel_cnt[m] = (len + 7) / 8;
    for (int c = 0; c < mel_cnt[m]; ++c) {
      pair_k0.push_back(first + 8 * c);
      for (int i = 0; i < 8; ++i) {
        const int k = first + 8 * c + i;
        w.push_back((k <= last) ? col[k] : 0.0f);
      }
    }
    if (mel_cnt[m] > kMaxParts) return PE_E_UNSUPPORTED;
  }
  const int n_pairs = (int)pair_k0.size();
  if (n_pairs > 256) return PE_E_UNSUPPORTED;        // chunk sums live in a 512-float2 exchange region
  std::vector<int> idx(pair_k0);
  idx.insert(idx.end(), mel_first.begin(), mel_first.end());
  idx.insert(idx.end(), mel_cnt.begin(), mel_cnt.end());

  std::vector<float> win(kNfft);
  std::vector<float2> tw512(512), tw1024(257);
  for (int n = 0; n < kNfft; ++n) win[n] = (float)(0.5 - 0.5 * cos(2.0 * M_PI * (double)n / (double)kNfft));
  for (int k = 0; k < 512; ++k) {
    const double ang = -2.0 * M_PI * (double)k / 512.0;
    tw512[k] = make_float2((float)cos(ang), (float)sin(ang));
  }
  for (int k = 0; k <= 256; ++k) {
    const double ang = -2.0 * M_PI * (double)k / 1024.0;
    tw1024[k] = make_float2((float)cos(ang), (float)sin(ang));
  }

  // one device allocation, 256-B aligned sub-tables
  auto up256 = [](size_t x) { return (x + 255) & ~(size_t)255; };
  size_t o_win = 0;
  size_t o_t5 = o_win + up256(sizeof(float) * kNfft);
  size_t o_t10 = o_t5 + up256(sizeof(float2) * 512);
  size_t o_ix = o_t10 + up256(sizeof(float2) * 257);
  size_t o_w = o_ix + up256(sizeof(int) * idx.size());
  size_t total = o_w + up256(sizeof(float) * (w.empty() ? 1 : w.size()));
  std::vector<char> host(total, 0);
  memcpy(host.data() + o_win, win.data(), sizeof(float) * kNfft);
  memcpy(host.data() + o_t5, tw512.data(), sizeof(float2) * 512);
  memcpy(host.data() + o_t10, tw1024.data(), sizeof(float2) * 257);
  memcpy(host.data() + o_ix, idx.data(), sizeof(int) * idx.size());
  if (!w.empty()) memcpy(host.data() + o_w, w.data(), sizeof(float) * w.size());

  void* d = nullptr;
  PE_CHECK_HIP(hipMalloc(&d, total));
  hipError_t e = hipMemcpy(d, host.data(), total, hipMemcpyHostToDevice);
  if (e != hipSuccess) { (void)hipFree(d); return (int)e; }
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mel_fwd_kernel<kFB>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) { (void)hipFree(d); return (int)e; }

  pe_mel_plan* p = new pe_mel_plan;
  p->sample_rate = sample_rate; p->n_fft = n_fft; p->hop = hop_length; p->n_mels = n_mels;
  p->n_freq = n_freq; p->n_pairs = n_pairs; p->d_base = d;
  char* c = reinterpret_cast<char*>(d);
  p->d_win = reinterpret_cast<float*>(c + o_win);
  p->d_tw512 = reinterpret_cast<float2*>(c + o_t5);
  p->d_tw1024 = reinterpret_cast<float2*>(c + o_t10);
  p->d_fb_idx = reinterpret_cast<int*>(c + o_ix);
  p->d_fb_w = reinterpret_cast<float*>(c + o_w);
  *plan_out = p;
  return PE_OK;
}

extern "C" int pe_mel_plan_destroy(pe_mel_plan* plan) {
  if (!plan) return PE_E_ARG;
  hipError_t e = hipFree(plan->d_base);
  delete plan;
  return (int)e;
}

extern "C" int pe_mel_num_frames(const pe_mel_plan* plan, int n_samples) {
  if (!plan || n_samples < 0) return PE_E_ARG;
  return 1 + n_samples / plan->hop;
}

static int mel_launch(const pe_mel_plan* plan, const float* wave, int batch, int n_samples, long wave_stride,
                      const int* n_samples_arr, const int* frame_start, float* out, long out_sb, long out_sm,
                      long out_st, int out_frames, int log_mode, float log_eps, float mean, float std,
                      float pad_value, void* stream) {
  if (!plan || !wave || !out || batch < 0 || out_frames < 0) return PE_E_ARG;
  // reflect padding needs n_fft/2 < n_samples (torch.stft raises otherwise)
  if ((!n_samples_arr && n_samples <= kHalf) || wave_stride < n_samples || std == 0.0f) return PE_E_ARG;
  if (batch == 0 || out_frames == 0) return PE_OK;
  if (batch > 65535) return PE_E_UNSUPPORTED;

  MelArgs a;
  a.n_samples_arr = n_samples_arr; a.frame_start = frame_start;
  a.wave = wave; a.wave_stride = wave_stride; a.n_samples = n_samples; a.hop = plan->hop;
  a.n_mels = plan->n_mels; a.n_valid = 1 + n_samples / plan->hop; a.out_frames = out_frames;
  a.out = out; a.out_sb = out_sb; a.out_sm = out_sm; a.out_st = out_st;
  a.log_mode = log_mode; a.log_eps = log_eps; a.mean = mean; a.inv_std = 1.0f / std;
  a.pad_value = pad_value;
  a.win = plan->d_win; a.tw512 = plan->d_tw512; a.tw1024 = plan->d_tw1024;
  a.fb_idx = plan->d_fb_idx; a.fb_w = plan->d_fb_w; a.n_pairs = plan->n_pairs;
  a.audio_len = 0;

  const size_t lds = mel_lds_bytes(plan->n_mels, plan->n_pairs, kFB);
  if (lds > 160 * 1024) return PE_E_UNSUPPORTED;
  // parts per utterance: fill the resident grid; the frames that carry work (the valid ones when the batch is
  // not ragged) are divided evenly, in multiples of 4 frames (one per wave)
  const long resident = (long)mel_cus() * ((160 * 1024) / (long)lds < 4 ? (160 * 1024) / (long)lds : 4);
  int work_frames = out_frames;
  if (!n_samples_arr && !frame_start && a.n_valid < work_frames) work_frames = a.n_valid;
  int parts = (int)((resident + batch - 1) / batch);
  const int max_parts = pe_cdiv(work_frames, 4);
  if (parts > max_parts) parts = max_parts;
  if (parts < 1) parts = 1;
  const int span = pe_cdiv(pe_cdiv(work_frames, parts), 4) * 4;
  parts = pe_cdiv(work_frames, span);
  const long n_items = (long)parts * batch;
  const int grid = (int)(n_items < resident ? n_items : resident);
  hipLaunchKernelGGL(mel_fwd_kernel<kFB>, dim3(grid), dim3(256), lds, pe_stream(stream), a, parts, span,
                     (int)n_items);
  PE_LAUNCH_CHECK();
  return PE_OK;
}

extern "C" int pe_mel_forward(const pe_mel_plan* plan, const float* wave, int batch, int n_samples,
                              long wave_stride, float* out, long out_sb, long out_sm, long out_st,
                              int out_frames, int log_mode, float log_eps, float mean, float std,
                              float pad_value, void* stream) {
  return mel_launch(plan, wave, batch, n_samples, wave_stride, nullptr, nullptr, out, out_sb, out_sm, out_st,
                    out_frames, log_mode, log_eps, mean, std, pad_value, stream);
}

extern "C" int pe_mel_forward_ragged(const pe_mel_plan* plan, const float* wave, int batch, int max_samples,
                                     long wave_stride, const int* n_samples, const int* frame_start, float* out,
                                     long out_sb, long out_sm, long out_st, int out_frames, int log_mode,
                                     float log_eps, float mean, float std, float pad_value, void* stream) {
  if (!n_samples) return PE_E_ARG;
  return mel_launch(plan, wave, batch, max_samples, wave_stride, n_samples, frame_start, out, out_sb, out_sm,
                    out_st, out_frames, log_mode, log_eps, mean, std, pad_value, stream);
}
