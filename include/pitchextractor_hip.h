/*
 * pitchextractor_hip.h -- C ABI of the MI355X (gfx950) JDC pitch-extractor
 * training hot path.  Built into pitchextractor_amd/libpitchextractor_hip.so.
 *
 * The reference (martinambrus/PitchExtractor) is pure Python on stock PyTorch
 * ops and has no FFI of its own; each entry point below names the reference
 * call site (file:line in the reference tree) whose work it replaces.
 *
 * Conventions (all entry points):
 *   - extern "C", plain C types only; every pointer is a DEVICE pointer owned
 *     by the caller unless the parameter name ends in _host;
 *   - `stream` is the caller's hipStream_t passed as void*; nothing here
 *     synchronises the device or allocates in steady state (plans/tables are
 *     created once by an explicit *_create);
 *   - activations are channels-last: [B][T][F][C] float32 ("NHWC", T = frames,
 *     F = mel/frequency axis, C = channels); conv weights are handed over in
 *     the reference's OIHW order and repacked on the device;
 *   - return value: 0 = ok, negative = invalid argument (PE_E_*), positive =
 *     hipError_t of a failed runtime call;
 *   - MFMA-bound entry points come in up to three forms with one contract:
 *       name       fp32 products on v_mfma_f32_32x32x2_f32;
 *       name_x3    fp32-accurate products as an exact three-term bf16 split on
 *                  v_mfma_f32_32x32x16_bf16 (the host side's default);
 *       name_bf16  operands rounded to bf16, fp32 accumulate (mixed precision).
 */
#ifndef PITCHEXTRACTOR_HIP_H
#define PITCHEXTRACTOR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PE_OK 0
#define PE_E_ARG (-1)       /* bad size / null pointer            */
#define PE_E_UNSUPPORTED (-2) /* shape outside what the kernels take */
#define PE_E_WORKSPACE (-3) /* workspace too small                 */

/* library / device ------------------------------------------------------- */
int pe_abi_version(void);
/* number of HIP devices visible, or negative hipError_t */
int pe_device_count(void);
/* *stream_out = a new non-blocking stream of the current device at its lowest priority (side work that must not
 * compete with the critical chain); created in this library's HIP runtime, owned by the caller. */
int pe_stream_create_low_priority(void** stream_out);

/* ---- mel front end ---------------------------------------------------------
 * Replaces MelDataset.to_melspec = torchaudio.transforms.MelSpectrogram(
 *   sample_rate=24000, n_fft=1024, win_length=1024, hop_length=300, n_mels=80)
 * (meldataset.py:34-40,58-77,644) and the log/normalise of meldataset.py:650,
 * batched over utterances.  center=True, reflect pad n_fft/2, periodic Hann,
 * power 2, HTK mel scale, f_min 0, f_max sr/2, norm None.
 */
typedef struct pe_mel_plan pe_mel_plan;
/* Only n_fft == win_length == 1024 is implemented (PE_E_UNSUPPORTED otherwise). */
int pe_mel_plan_create(pe_mel_plan** plan, int sample_rate, int n_fft, int win_length,
                       int hop_length, int n_mels, float f_min, float f_max);
int pe_mel_plan_destroy(pe_mel_plan* plan);
/* frames produced for an n_samples utterance: 1 + n_samples / hop */
int pe_mel_num_frames(const pe_mel_plan* plan, int n_samples);
/* wave: [batch] rows of n_samples float32, row stride wave_stride (elements).
 * out:  element (b, m, t) at out[b*out_sb + m*out_sm + t*out_st], for
 *       t < out_frames.  Frames t >= 1 + n_samples/hop are written as pad_value
 *       (Collater zero padding, meldataset.py:806-816).
 * log_mode 0: mel power; 1: (log(log_eps + mel) - mean) / std. */
int pe_mel_forward(const pe_mel_plan* plan, const float* wave, int batch, int n_samples,
                   long wave_stride, float* out, long out_sb, long out_sm, long out_st,
                   int out_frames, int log_mode, float log_eps, float mean, float std,
                   float pad_value, void* stream);
/* Ragged batch: utterance b has n_samples[b] <= max_samples samples (device int array) and, when
 * frame_start != NULL, output frame t shows source frame t + frame_start[b] -- the random crop of
 * meldataset.py:668-672 applied on the frame axis without recomputing anything.  Items with
 * n_samples[b] <= n_fft/2 produce padding only. */
int pe_mel_forward_ragged(const pe_mel_plan* plan, const float* wave, int batch, int max_samples,
                          long wave_stride, const int* n_samples, const int* frame_start, float* out,
                          long out_sb, long out_sm, long out_st, int out_frames, int log_mode,
                          float log_eps, float mean, float std, float pad_value, void* stream);


/* ---- dense fp32 GEMMs on the MFMA engine -----------------------------------
 * pe_gemm_nt: C[M][N] = A[M][K] . B[N][K]^T + bias0[n] + bias1[n] (+ C when accumulate).
 *   nn.Linear / LSTM input projection (model.py:220-227) / 1x1 convs (model.py:53,167).
 * pe_gemm_tn: C[M][N] = sum_k A[k][m] * B[k][n] (+ C): weight gradients, k split across
 *   workgroups into workspace slabs that are reduced in a fixed order (deterministic). */
int pe_gemm_nt(const float* A, long lda, const float* B, long ldb, float* C, long ldc, int M, int N,
               int K, const float* bias0, const float* bias1, int accumulate, void* stream);
/* Opt-in mixed-precision variant (reference trainer.py:103 autocast): same contract, operands rounded to
 * bf16 on the way into LDS (v_mfma_f32_32x32x16_bf16), fp32 accumulate, fp32 tensors in HBM. */
int pe_gemm_nt_bf16(const float* A, long lda, const float* B, long ldb, float* C, long ldc, int M, int N,
                    int K, const float* bias0, const float* bias1, int accumulate, void* stream);
/* fp32-accurate variant on the bf16 MFMA pipe: each fp32 operand is split exactly into three bf16 terms
 * (x = hi + mid + lo) and six of the nine cross products are accumulated in fp32; the dropped terms are
 * below 2^-23 of each product, i.e. under the rounding of an fp32 product.  Same contract as pe_gemm_nt. */
int pe_gemm_nt_x3(const float* A, long lda, const float* B, long ldb, float* C, long ldc, int M, int N,
                  int K, const float* bias0, const float* bias1, int accumulate, void* stream);
/* "h2": fp32 products from TWO fp16 terms per operand and three fp16 MFMAs per product block (half the matrix work of
 * the x3 split).  x * s = hi + lo with hi = RN_f16(x s), lo = RN_f16(x s - hi); hi_a lo_b + lo_a hi_b + hi_a hi_b is
 * accumulated in fp32 and lo_a lo_b (<= 2^-24 |a b|) dropped: per-product error <= 2^-21 |a b|, unbiased.  Each
 * operand tensor carries a power-of-two scale s = 2^(140 - E), E = biased exponent of its largest magnitude, which
 * the kernel derives from *amax_a / *amax_b (device words holding the IEEE bits of max |x|: pe_absmax, or the
 * epilogue of the kernel that produced the tensor); a value smaller than the tensor's true maximum makes fp16
 * overflow possible, a larger one only costs resolution (2^-38 of the stated maximum, absolute).  Same contract as
 * pe_gemm_nt otherwise. */
int pe_gemm_nt_h2(const float* A, long lda, const float* B, long ldb, float* C, long ldc, int M, int N,
                  int K, const float* bias0, const float* bias1, int accumulate, const unsigned* amax_a,
                  const unsigned* amax_b, void* stream);
/* out[0] = IEEE bits of max |x| over a [rows][cols] matrix with leading dimension ld (cols, ld % 4 == 0, x 16-byte
 * aligned); zeroes out[0] first.  Exact and order-independent (integer max of the magnitudes' bit patterns). */
int pe_absmax(const float* x, long rows, int cols, long ld, unsigned* out, void* stream);
/* out[s] = IEEE bits of max |base[seg_off[s] .. + seg_len[s])| for s < nseg, one launch (seg_off / seg_len: device
 * arrays; every parameter of a model that lives in one flat buffer). */
int pe_absmax_segments(const float* base, const long* seg_off, const long* seg_len, int nseg, unsigned* out,
                       void* stream);
size_t pe_gemm_tn_workspace_bytes(int M, int N, int K);
int pe_gemm_tn(const float* A, long lda, const float* B, long ldb, float* C, long ldc, int M, int N,
               int K, int accumulate, float* workspace, size_t workspace_bytes, void* stream);
int pe_gemm_tn_x3(const float* A, long lda, const float* B, long ldb, float* C, long ldc, int M, int N,
                  int K, int accumulate, float* workspace, size_t workspace_bytes, void* stream);
int pe_gemm_tn_bf16(const float* A, long lda, const float* B, long ldb, float* C, long ldc, int M, int N,
                    int K, int accumulate, float* workspace, size_t workspace_bytes, void* stream);  /* mixed precision */
int pe_gemm_tn_h2(const float* A, long lda, const float* B, long ldb, float* C, long ldc, int M, int N,
                  int K, int accumulate, float* workspace, size_t workspace_bytes, const unsigned* amax_a,
                  const unsigned* amax_b, void* stream);   /* two-term fp16 split, see pe_gemm_nt_h2 */
int pe_transpose2d(const float* in, float* out, int rows, int cols, void* stream);

/* ---- 3x3 / pad 1 convolutions (model.py:23-28,157-161), channels-last -------
 * pe_conv3x3_repack: OIHW weights -> w_fwd [Cout][kh][kw][Cin] and/or the flipped, transposed
 *   w_dgrad [Cin][2-kh][2-kw][Cout] (either may be NULL).
 * pe_conv3x3_fwd:  y[B][T][F][N] (+)= conv(x[B][T][F][C], w_packed[N][9*C]); the data gradient
 *   is the same call with (dy, w_dgrad, C = Cout, N = Cin).
 * pe_conv3x3_wgrad: dw (OIHW) = sum_pixels dy (x) shifted x.
 * pe_conv3x3_c1_*: the Cin = 1 first layer; x element (b,t,f) at x[b*sb + t*st + f*sf]. */
int pe_conv3x3_repack(const float* w_oihw, float* w_fwd, float* w_dgrad, int Cout, int Cin, void* stream);
int pe_conv3x3_fwd(const float* x, const float* w_packed, float* y, int B, int T, int F, int C, int N,
                   int accumulate, void* stream);
int pe_conv3x3_fwd_bf16(const float* x, const float* w_packed, float* y, int B, int T, int F, int C, int N,
                        int accumulate, void* stream);   /* bf16 operands, fp32 accumulate (see pe_gemm_nt_bf16) */
int pe_conv3x3_fwd_x3(const float* x, const float* w_packed, float* y, int B, int T, int F, int C, int N,
                      int accumulate, void* stream);     /* fp32-accurate, three-term bf16 split (see pe_gemm_nt_x3) */
int pe_conv3x3_fwd_h2(const float* x, const float* w_packed, float* y, int B, int T, int F, int C, int N,
                      int accumulate, const unsigned* amax_x, const unsigned* amax_w,
                      void* stream);                       /* two scaled fp16 terms (see pe_gemm_nt_h2) */
/* Weights pre-packed as MFMA B-operand fragments (x3: three exact bf16 terms, terms = 3; mixed precision: one
 * RNE-rounded term, terms = 1).  w is [N][K] row-major fp32 (K % 16 == 0); fragment (kb, nb, term) holds, for
 * lane 32 h + r, w[32 nb + r][16 kb + 8 h .. + 7] in 16 bytes, the 64 lanes contiguous (1 KB).  The halo
 * convolution then loads its weight operands straight from L2 into registers: LDS carries only activations.
 * pe_conv3x3_wf_supported: 1 if (F, C, N) is served by the fragment-fed kernel (else use pe_conv3x3_fwd_*). */
size_t pe_wfrag_bytes(int N, int K, int terms);
int pe_wfrag_pack(const float* w, long ld, int N, int K, int terms, void* wfrag, void* stream);
/* "h2" form: two fp16 terms of w * 2^(140 - E) per weight, E from *amax (pe_absmax of w); pe_wfrag_bytes(N, K, 2). */
int pe_wfrag_pack_h2(const float* w, long ld, int N, int K, const unsigned* amax, void* out, void* stream);
int pe_conv3x3_wf_supported(int F, int C, int N);
int pe_conv3x3_fwd_wf_x3(const float* x, const void* wfrag, float* y, int B, int T, int F, int C, int N,
                         int accumulate, double* bn_partials, void* stream);
int pe_conv3x3_fwd_wf_h2(const float* x, const void* wfrag, float* y, int B, int T, int F, int C, int N,
                         int accumulate, double* bn_partials, const unsigned* amax_x, const unsigned* amax_w,
                         void* stream);                    /* wfrag from pe_wfrag_pack_h2 with the same *amax_w */
int pe_conv3x3_fwd_wf_bf16(const float* x, const void* wfrag, float* y, int B, int T, int F, int C, int N,
                           int accumulate, double* bn_partials, void* stream);
/* bn_partials (optional): [pe_conv3x3_wf_stat_parts(B,T,F)][2][N] doubles -- per pixel tile, the column sums and sums
 * of squares of the FINAL outputs; pe_bn_finalize_stats turns them into the statistics of the BatchNorm that follows
 * (no separate pass over the activation). */
int pe_conv3x3_wf_stat_parts(int B, int T, int F);
size_t pe_conv3x3_wgrad_workspace_bytes(int B, int T, int F, int Cin, int Cout);
int pe_conv3x3_wgrad(const float* x, const float* dy, float* dw_oihw, int B, int T, int F, int Cin,
                     int Cout, float* workspace, size_t workspace_bytes, void* stream);
int pe_conv3x3_wgrad_x3(const float* x, const float* dy, float* dw_oihw, int B, int T, int F, int Cin,
                        int Cout, float* workspace, size_t workspace_bytes, void* stream);
int pe_conv3x3_wgrad_h2(const float* x, const float* dy, float* dw_oihw, int B, int T, int F, int Cin, int Cout,
                        float* workspace, size_t workspace_bytes, const unsigned* amax_x, const unsigned* amax_dy,
                        void* stream);
int pe_conv3x3_wgrad_bf16(const float* x, const float* dy, float* dw_oihw, int B, int T, int F, int Cin,
                          int Cout, float* workspace, size_t workspace_bytes, void* stream);   /* mixed precision */
/* first convolution (1 -> 64 channels).  bn_partials (nullable): [pe_conv3x3_c1_stat_parts()][2][64] doubles that
 * receive per-workgroup sums / sums of squares of the output channels (input of pe_bn_finalize_stats). */
int pe_conv3x3_c1_stat_parts(int B, int T, int F);
int pe_conv3x3_c1_fwd(const float* x, long sb, long st, long sf, const float* w_oihw, float* y, int B,
                      int T, int F, double* bn_partials, void* stream);
int pe_conv3x3_c1_wgrad(const float* x, long sb, long st, long sf, const float* dy, float* dw_oihw,
                        int B, int T, int F, float* workspace, size_t workspace_bytes, void* stream);

/* ---- fused self-attention (model.py:231-239: nn.MultiheadAttention inside TransformerEncoderLayer) -------------
 * qkv [B*T][ld_qkv] packed projections (Q | K | V, head h at columns h*dh of each part); o [B*T][ld_o] merged
 * heads; lse [B*H*T]; masks [B*H*T][T] bytes (1 = kept), element quad i = row * T/4 + key/4 draws Philox counter
 * offset + i exactly like pe_dropout_fwd on the (B*H*T) x T probability matrix.  One workgroup per (batch, head),
 * scores stay in registers; the backward recomputes them from lse.  pe_attn_supported: 1 for T = 192, dh = 64
 * (other shapes: pe_bgemm + pe_softmax_*). */
int pe_attn_supported(int T, int dh);
int pe_attn_fwd(const float* qkv, long ld_qkv, float* o, long ld_o, float* lse, const unsigned char* mask_in,
                unsigned char* mask_out, int B, int T, int H, int dh, float scale, float p_drop,
                unsigned long long seed, unsigned long long offset, void* stream);
int pe_attn_bwd(const float* qkv, long ld_qkv, const float* o, const float* d_o, long ld_o, const float* lse,
                const unsigned char* mask, float* dqkv, int B, int T, int H, int dh, float scale, float p_drop,
                void* stream);
/* The same two passes with the matmul operands (Q, K, V, dO, the probabilities and the score gradients) rounded to
 * bf16 and multiplied on v_mfma_f32_16x16x16_bf16, as torch.autocast runs the attention of trainer.py:226-235; softmax,
 * log-sum-exp, accumulation and every tensor in memory stay fp32, masks and Philox counters as above. */
int pe_attn_fwd_bf16(const float* qkv, long ld_qkv, float* o, long ld_o, float* lse, const unsigned char* mask_in,
                     unsigned char* mask_out, int B, int T, int H, int dh, float scale, float p_drop,
                     unsigned long long seed, unsigned long long offset, void* stream);
int pe_attn_bwd_bf16(const float* qkv, long ld_qkv, const float* o, const float* d_o, long ld_o, const float* lse,
                     const unsigned char* mask, float* dqkv, int B, int T, int H, int dh, float scale, float p_drop,
                     void* stream);

/* ---- fp16 operands (the reference's autocast default dtype, trainer.py:64-102) -------------------------------
 * Same contracts as the *_bf16 entry points above with operands rounded (RNE) to IEEE half instead of bf16 and
 * multiplied by v_mfma_f32_32x32x16_f16; accumulation and every tensor in memory stay fp32.  fp16 has 5 exponent
 * bits: the caller scales the loss (Trainer's GradScaler) so that gradients stay above 2^-24.
 * pe_wfrag_pack_f16: fragment order of pe_wfrag_pack with one fp16 term per weight. */
int pe_gemm_nt_f16(const float* A, long lda, const float* B, long ldb, float* C, long ldc, int M, int N,
                    int K, const float* bias0, const float* bias1, int accumulate, void* stream);
int pe_gemm_tn_f16(const float* A, long lda, const float* B, long ldb, float* C, long ldc, int M, int N,
                    int K, int accumulate, float* workspace, size_t workspace_bytes, void* stream);
int pe_conv3x3_fwd_f16(const float* x, const float* w_packed, float* y, int B, int T, int F, int C, int N,
                        int accumulate, void* stream);
int pe_conv3x3_fwd_wf_f16(const float* x, const void* wfrag, float* y, int B, int T, int F, int C, int N,
                          int accumulate, double* bn_partials, void* stream);
int pe_conv3x3_wgrad_f16(const float* x, const float* dy, float* dw_oihw, int B, int T, int F, int Cin,
                          int Cout, float* workspace, size_t workspace_bytes, void* stream);
int pe_lstm_fwd_persistent_f16(int ncells, const float* const* whh, float* const* gates, float* const* y,
                           float* const* cbuf, const int* reverse, long ldy, int B, int T, int H,
                           unsigned* sync, void* stream);
int pe_lstm_bwd_persistent_f16(int ncells, const float* const* whh_t, float* const* gates,
                           const float* const* cbuf, const float* const* dy, const int* reverse, long lddy,
                           int B, int T, int H, float* const* dbias_rows, unsigned* const* dgates_amax,
                           unsigned* sync, void* stream);
int pe_lstm_whh_grad_f16(const float* dgates, const float* y, long ldy, float* dwhh, int B, int T, int H,
                     int reverse, float* workspace, size_t workspace_bytes, void* stream);
int pe_wfrag_pack_f16(const float* w, long ld, int N, int K, void* wfrag, void* stream);

/* ---- BatchNorm2d (train statistics) / LeakyReLU / MaxPool2d((1,k)) / dropout -
 * Activations are [rows = B*T][F][C].  pe_bn_train_stats: batch mean / biased variance over all
 * n_pix = rows*F pixels (model.py:25,37,54,150,159), running-stat update with `momentum` and the
 * unbiased variance, plus the fused affine scale = gamma*invstd, shift = beta - mean*scale.
 * pe_bn_act_pool_fwd: y = maxpool_k(lrelu(x*scale + shift)) written at
 *   y[(row*Fout + fo)*ldy + coff + c] (model.py:36-41,148-153).
 * pe_bn_act_pool_bwd: gradient of that block w.r.t. x, gamma, beta (train-mode BN).
 * amax_out (optional, these two and pe_maxpool_bwd_add): a device word the caller zeroed; the pass max-merges the
 *   IEEE bits of the largest magnitude it stored into it (atomicMax), which is the scale source an "h2" product
 *   reading the output needs (pe_gemm_nt_h2) -- no separate pe_absmax pass.  pe_maxpool_bwd_add merges the values it
 *   rewrote into dx's word: the result bounds max |dx| from above. */
size_t pe_bn_workspace_bytes(int C);
int pe_bn_train_stats(const float* x, long n_pix, int C, const float* gamma, const float* beta, float eps,
                      float momentum, float* running_mean, float* running_var, float* mean, float* invstd,
                      float* scale, float* shift, void* workspace, size_t workspace_bytes, void* stream);
int pe_bn_finalize_stats(const double* partials, int nparts, long n_pix, int C, const float* gamma, const float* beta,
                         float eps, float momentum, float* running_mean, float* running_var, float* mean,
                         float* invstd, float* scale, float* shift, void* workspace, size_t workspace_bytes,
                         void* stream);   /* workspace: pe_bn_workspace_bytes(C) */
int pe_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean,
                      const float* running_var, float eps, int C, float* scale, float* shift, void* stream);
int pe_bn_act_pool_fwd(const float* x, const float* scale, const float* shift, float slope, float* y,
                       long rows, int Fin, int C, int pool, long ldy, int coff, unsigned* amax_out, void* stream);
int pe_bn_act_pool_bwd(const float* x, const float* dy, const float* scale, const float* shift,
                       const float* mean, const float* invstd, float slope, float* dx, float* dgamma,
                       float* dbeta, long rows, int Fin, int C, int pool, long lddy, int coff,
                       void* workspace, size_t workspace_bytes, unsigned* amax_out, void* stream);
/* detector-branch MaxPool2d((1,40|20|10)) (model.py:45-49,103-105) into a channel slice.  argmax_out (optional,
 * [rows * (Fin / pool)][C] bytes): the window position of each maximum (the first one, as torch's backward routes it);
 * pe_maxpool_bwd_add given that array as `argmax` scatters dy without reading x (x may then be NULL). */
int pe_maxpool_fwd(const float* x, float* y, long rows, int Fin, int C, int pool, long ldy, int coff,
                   unsigned char* argmax_out, void* stream);
int pe_maxpool_bwd_add(const float* x, const unsigned char* argmax, const float* dy, float* dx, long rows, int Fin,
                       int C, int pool, long lddy, int coff, unsigned* amax_out, void* stream);
/* nn.Dropout (model.py:40,56; LSTM inter-layer): Philox4x32-10 keyed by (seed, offset + quad index);
 * mask bytes (1 = kept) can be exported (mask_out) or replayed (mask_in). */
int pe_dropout_fwd(const float* x, long ldx, float* y, long ldy, const unsigned char* mask_in,
                   unsigned char* mask_out, long rows, int cols, float p, unsigned long long seed,
                   unsigned long long offset, void* stream);
/* (B,256,T,2) -> permute(0,2,1,3) -> (B,T,512) of model.py:93,112 and its transpose */
int pe_nhwc_to_seq(const float* x, long ldx, int coff, float* seq, long rows, int C, void* stream);
int pe_seq_to_nhwc(const float* seq, float* x, long ldx, int coff, long rows, int C, int accumulate,
                   void* stream);
int pe_copy2d(const float* src, long lds, float* dst, long ldd, long rows, int cols, int accumulate,
              void* stream);

/* ---- LSTM recurrence (nn.LSTM of model.py:218-227; gates i,f,g,o; zero initial state) ----
 * Up to 4 cells (directions x models) advance together, one launch per time step.
 * pe_lstm_fwd:  gates[c] holds X.W_ih^T + b_ih + b_hh on entry ([B][T][4H]) and the activated gates
 *   on exit; y[c] (already offset to this direction's H-wide slice, row stride ldy) receives h_t,
 *   cbuf[c] ([B][T][H]) the cell states.
 * pe_lstm_bwd:  gates[c] holds activated gates on entry and d(pre-activation gates) on exit;
 *   whh_t[c] = W_hh^T [H][4H]; dy[c] = dL/dh (same addressing as y); dcarry[c] = [B][H] scratch. */
int pe_lstm_fwd(int ncells, const float* const* whh, float* const* gates, float* const* y,
                float* const* cbuf, const int* reverse, long ldy, int B, int T, int H, void* stream);
int pe_lstm_bwd(int ncells, const float* const* whh_t, float* const* gates, const float* const* cbuf,
                const float* const* dy, float* const* dcarry, const int* reverse, long lddy, int B, int T,
                int H, void* stream);
/* Persistent variants: one launch for all T steps, W_hh slice resident in registers, group barriers
 * between steps (agent-scope release/acquire).  `sync` = pe_lstm_persistent_sync_bytes() of device
 * memory, zero-initialised once by the caller: the group counters, then the backward kernel's exchange
 * region (the step's gate gradients in MFMA fragment order, two slots per batch tile).  Word 0 is a
 * sticky error flag (non-zero = a bounded spin timed out: results invalid).  Only when pe_lstm_persistent_supported() returns 1. */
size_t pe_lstm_persistent_sync_bytes(int ncells, int B);
int pe_lstm_persistent_supported(int ncells, int B, int H);
/* Persistent recurrences (H = 384): ONE launch per layer for all cells, W_hh resident on chip, workgroups hand h /
 * partial dh tiles to each other through memory.  The recurrent products are 16-bit-term MFMAs: `_x3` = the exact
 * three-term bf16 split (fp32-accurate), `_bf16` / `_f16` = operands rounded to 16 bits (mixed precision).  There is
 * no native-fp32 persistent form: pe_lstm_fwd / pe_lstm_bwd (one launch per time step) serve that mode and every
 * shape pe_lstm_persistent_supported() declines.
 * dbias_rows (nullable): per cell a [pe_lstm_bwd_persistent_dbias_rows()][4H] buffer that receives the per-batch-tile
 * column sums of the gate gradients (their row sum is dL/db_ih = dL/db_hh), replacing a pe_colsum pass over the
 * [B*T][4H] gradient tensor.
 * dgates_amax (nullable): per cell a device word the caller zeroed; the kernel max-merges the IEEE bits of the largest
 * gate-gradient magnitude into it (the "h2" scale source of the dX / dW products). */
int pe_lstm_bwd_persistent_dbias_rows(int ncells, int B, int T, int H, long lddy);
int pe_lstm_fwd_persistent_x3(int ncells, const float* const* whh, float* const* gates, float* const* y,
                           float* const* cbuf, const int* reverse, long ldy, int B, int T, int H,
                           unsigned* sync, void* stream);
int pe_lstm_bwd_persistent_x3(int ncells, const float* const* whh_t, float* const* gates,
                           const float* const* cbuf, const float* const* dy, const int* reverse, long lddy,
                           int B, int T, int H, float* const* dbias_rows, unsigned* const* dgates_amax,
                           unsigned* sync, void* stream);
/* mixed precision: W_hh and the h / dgates rows rounded to bf16, fp32 accumulate and cell state */
int pe_lstm_fwd_persistent_bf16(int ncells, const float* const* whh, float* const* gates, float* const* y,
                           float* const* cbuf, const int* reverse, long ldy, int B, int T, int H,
                           unsigned* sync, void* stream);
int pe_lstm_bwd_persistent_bf16(int ncells, const float* const* whh_t, float* const* gates,
                           const float* const* cbuf, const float* const* dy, const int* reverse, long lddy,
                           int B, int T, int H, float* const* dbias_rows, unsigned* const* dgates_amax,
                           unsigned* sync, void* stream);
/* Diagnostic (tools/stamp_lstm.py): 1 = run the stamped instantiations of the x3 kernels (s_memtime per region of an
 * iteration; grid <= 128 workgroups).  Returns the previous setting.  The product path never calls it. */
int pe_lstm_configure_stamps(int enable);
size_t pe_lstm_whh_grad_workspace_bytes(int B, int T, int H);
int pe_lstm_whh_grad(const float* dgates, const float* y, long ldy, float* dwhh, int B, int T, int H,
                     int reverse, float* workspace, size_t workspace_bytes, void* stream);
int pe_lstm_whh_grad_x3(const float* dgates, const float* y, long ldy, float* dwhh, int B, int T, int H,
                     int reverse, float* workspace, size_t workspace_bytes, void* stream);   /* three-term bf16 split */
int pe_lstm_whh_grad_h2(const float* dgates, const float* y, long ldy, float* dwhh, int B, int T, int H, int reverse,
                        float* workspace, size_t workspace_bytes, const unsigned* amax_dgates, const unsigned* amax_y,
                        void* stream);
int pe_lstm_whh_grad_bf16(const float* dgates, const float* y, long ldy, float* dwhh, int B, int T, int H,
                     int reverse, float* workspace, size_t workspace_bytes, void* stream);   /* mixed precision */
size_t pe_colsum_workspace_bytes(int cols);
int pe_colsum(const float* x, long rows, int cols, long ld, float* out0, float* out1, void* workspace,
              size_t workspace_bytes, void* stream);

/* ---- heads, losses, optimiser ------------------------------------------------
 * pe_head_fwd: y[r] = sum_{o<n_out} (x[r].w[o] + bias[o]) -- Linear(D,1) (n_out 1) and
 *   Linear(D,2).sum(-1) (n_out 2) of model.py:67-70,96-98,115-117.
 * pe_f0_sil_loss: out3 = {lambda*SmoothL1 + BCE, lambda*SmoothL1, BCE} (train.py:104-106,
 *   trainer.py:237-239) and the gradients w.r.t. both prediction vectors (times grad_scale).
 * pe_adamw_step: torch.optim.AdamW (optimizers.py:55-62) on a flat buffer; the caller passes the
 *   current lr / beta1 (OneCycleLR rewrites both every step) and the bias corrections
 *   1 - beta^step computed in double. Gradients are multiplied by grad_scale first. */
int pe_head_fwd(const float* x, long ldx, const float* w, const float* bias, int n_out, float* y, long R,
                int D, void* stream);
size_t pe_head_bwd_workspace_bytes(int D);
int pe_head_bwd(const float* x, long ldx, const float* w, const float* dy, int n_out, float* dx, long lddx,
                float* dw, float* db, long R, int D, void* workspace, size_t workspace_bytes, void* stream);
int pe_f0_sil_loss(const float* f0_pred, const float* f0, const float* sil_pred, const float* sil,
                   float lambda_f0, long R, float grad_scale, float* out3, float* d_f0_pred,
                   float* d_sil_pred, void* stream);
/* 360-bin F0 classification loss (build-defined, the reference has none): CREPE bins
 * bin = clamp(rint((1200 log2(f0/10) - 1997.3794084376191) / 20), 0, C-1) on voiced frames (f0 > 0), CE averaged
 * over voiced frames, total = lambda * CE + BCEWithLogits(sil).  out4 = {total, lambda*CE, BCE, voiced count}. */
size_t pe_f0_bins_ce_workspace_bytes(long R);
int pe_f0_bins_ce_loss(const float* logits, long ldl, int C, const float* f0, const float* sil_pred,
                       const float* sil, float lambda_f0, long R, float grad_scale, float* out4,
                       float* d_logits, long ldd, float* d_sil_pred, float* workspace, size_t workspace_bytes,
                       void* stream);
/* skip_if_nonzero (nullable): a device float; when it is non-zero at execution time the launch updates nothing (the
 * trainer's fault word, agreed across ranks inside the gradient all-reduce: no host round trip before the update). */
int pe_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long n, float lr,
                  float beta1, float beta2, float eps, float weight_decay, double bias_correction1,
                  double bias_correction2, float grad_scale, const float* skip_if_nonzero, void* stream);
/* GradScaler support (reference trainer.py:241-244): *flag = 1 if any of x[0..n) is inf or nan, else 0. */
int pe_nonfinite_flag(const float* x, long n, int* flag, void* stream);

/* ---- bf16 ACTIVATION STORAGE for mixed precision (`*_a16`) ------------------------------------------------------
 * The reference's autocast keeps conv / linear outputs and what backward saves of them in 16 bits (trainer.py:226-235,
 * README.md:36).  These entry points are the fp32-tensor functions of the same name with the conv stack's activation
 * and activation-gradient tensors (x, y, dy, dx, the GEMM operand A and result C) stored as bf16 in HBM: half the bytes
 * of every BatchNorm / pooling / staging pass and of the saved-for-backward footprint.  Arithmetic, BatchNorm
 * statistics, weights, weight gradients and biases stay fp32; a store rounds to nearest even.  Statistics a kernel
 * leaves behind (bn_partials) are those of the ROUNDED values, i.e. of the tensor BatchNorm then reads.  Layouts,
 * strides (in elements) and argument meaning are unchanged; activation pointers need 8-byte alignment. */
int pe_conv3x3_c1_fwd_a16(const float* x, long sb, long st, long sf, const float* w_oihw, void* y, int B, int T, int F,
                          double* bn_partials, void* stream);
int pe_conv3x3_c1_wgrad_a16(const float* x, long sb, long st, long sf, const void* dy, float* dw_oihw, int B, int T,
                            int F, float* workspace, size_t workspace_bytes, void* stream);
int pe_conv3x3_fwd_bf16_a16(const void* x, const float* w_packed, void* y, int B, int T, int F, int C, int N,
                            int accumulate, void* stream);
int pe_conv3x3_fwd_wf_bf16_a16(const void* x, const void* wfrag, void* y, int B, int T, int F, int C, int N,
                               int accumulate, double* bn_partials, void* stream);
int pe_conv3x3_wgrad_bf16_a16(const void* x, const void* dy, float* dw_oihw, int B, int T, int F, int Cin, int Cout,
                              float* workspace, size_t workspace_bytes, void* stream);
int pe_gemm_nt_bf16_a16(const void* A, long lda, const float* B, long ldb, void* C, long ldc, int M, int N, int K,
                        const float* bias0, const float* bias1, int accumulate, void* stream);
int pe_gemm_tn_bf16_a16(const void* A, long lda, const void* B, long ldb, float* C, long ldc, int M, int N, int K,
                        int accumulate, float* workspace, size_t workspace_bytes, void* stream);
int pe_bn_train_stats_a16(const void* x, long n_pix, int C, const float* gamma, const float* beta, float eps,
                          float momentum, float* running_mean, float* running_var, float* mean, float* invstd,
                          float* scale, float* shift, void* workspace, size_t workspace_bytes, void* stream);
int pe_bn_act_pool_fwd_a16(const void* x, const float* scale, const float* shift, float slope, void* y, long rows,
                           int Fin, int C, int pool, long ldy, int coff, void* stream);
int pe_bn_act_pool_bwd_a16(const void* x, const void* dy, const float* scale, const float* shift, const float* mean,
                           const float* invstd, float slope, void* dx, float* dgamma, float* dbeta, long rows, int Fin,
                           int C, int pool, long lddy, int coff, void* workspace, size_t workspace_bytes,
                           void* stream);          /* pool in {1, 2, 4} */
int pe_maxpool_fwd_a16(const void* x, void* y, long rows, int Fin, int C, int pool, long ldy, int coff,
                       unsigned char* argmax_out, void* stream);
int pe_maxpool_bwd_add_a16(const void* x, const unsigned char* argmax, const void* dy, void* dx, long rows, int Fin,
                           int C, int pool, long lddy, int coff, void* stream);
int pe_dropout_fwd_a16(const void* x, long ldx, void* y, long ldy, const unsigned char* mask_in,
                       unsigned char* mask_out, long rows, int cols, float p, unsigned long long seed,
                       unsigned long long offset, void* stream);
int pe_nhwc_to_seq_a16(const void* x, long ldx, int coff, float* seq, long rows, int C, void* stream);   /* bf16 -> fp32 */
int pe_seq_to_nhwc_a16(const float* seq, void* x, long ldx, int coff, long rows, int C, int accumulate,
                       void* stream);                                                                    /* fp32 -> bf16 */
int pe_copy2d_a16(const void* src, long lds, void* dst, long ldd, long rows, int cols, int accumulate, void* stream);

/* ---- Transformer temporal head (model.py:178-193,229-241,253-255) ---------------------------
 * pe_bgemm: batched 64x64-tiled fp32 MFMA GEMM over `batch` matrices; matrix b of operand X lives
 *   at X + (b / inner) * x_outer + (b % inner) * x_inner (e.g. batch item / head inside the packed
 *   QKV projection).  mode 0: C = alpha A B^T (A [M][K], B [N][K]); 1: C = alpha A B (B [K][N]);
 *   2: C = alpha A^T B (A [K][M], B [K][N]).
 * pe_softmax_fwd: s[row][:L] = softmax(scale * s[row][:L]) in place; pe_softmax_bwd: dp <- ds.
 * pe_layernorm_fwd: z = a (+ b) (+ pe[row % period]); y = LN(z) * gamma + beta (eps inside the
 *   sqrt); z_out (optional) keeps z for the backward; D in {256, 512, 768, 1024}.
 * pe_gelu_*: exact erf GELU (activation="gelu") and its derivative. */
int pe_bgemm(int mode, const float* A, long lda, long a_outer, long a_inner, const float* B, long ldb,
             long b_outer, long b_inner, float* C, long ldc, long c_outer, long c_inner, int inner, int batch,
             int M, int N, int K, float alpha, int accumulate, void* stream);
int pe_softmax_fwd(float* s, long rows, int L, float scale, void* stream);
int pe_softmax_bwd(const float* p, float* dp, long rows, int L, float scale, void* stream);
int pe_layernorm_fwd(const float* a, const float* b, const float* pe, int period, const float* gamma,
                     const float* beta, float eps, float* z_out, float* y, float* mean, float* rstd, long rows,
                     int D, void* stream);
size_t pe_layernorm_bwd_workspace_bytes(int D);
int pe_layernorm_bwd(const float* dy, const float* z, const float* mean, const float* rstd, const float* gamma,
                     float* dz, float* dgamma, float* dbeta, long rows, int D, void* workspace,
                     size_t workspace_bytes, void* stream);
int pe_gelu_fwd(const float* x, float* y, long n, void* stream);
int pe_gelu_bwd(const float* x, const float* dy, float* dx, long n, void* stream);
/* Dropout folded into the neighbouring pass of an encoder layer (nn.TransformerEncoderLayer, model.py:231-239:
 * x = norm1(x + dropout1(sa)); x = norm2(x + dropout2(linear2(dropout(gelu(linear1(x))))))).  Element <-> Philox
 * counter mapping, keep rule and 1/(1-p) scaling are pe_dropout_fwd's on a dense tensor, so the results equal the
 * separate passes bit for bit; mask_in replays given bytes, mask_out (optional) records them; 0 < p < 1.
 * pe_layernorm_dropout_fwd: z = a + dropout(b) (+ pe); y = LN(z).
 * pe_layernorm_bwd_fused: dy2 (optional) is added to dy first (the residual branch's gradient); dz_drop (with
 *   drop_mask, both or neither) additionally receives dropout_bwd(dz).
 * pe_gelu_dropout_fwd: y = dropout(gelu(x)); pe_gelu_dropout_bwd: dx = dropout_bwd(dy) * gelu'(x). */
int pe_layernorm_dropout_fwd(const float* a, const float* b, const float* pe, int period, const float* gamma,
                             const float* beta, float eps, float* z_out, float* y, float* mean, float* rstd,
                             long rows, int D, const unsigned char* mask_in, unsigned char* mask_out, float p,
                             unsigned long long seed, unsigned long long offset, void* stream);
int pe_layernorm_bwd_fused(const float* dy, const float* dy2, const float* z, const float* mean, const float* rstd,
                           const float* gamma, float* dz, const unsigned char* drop_mask, float p, float* dz_drop,
                           float* dgamma, float* dbeta, long rows, int D, void* workspace, size_t workspace_bytes,
                           void* stream);
int pe_gelu_dropout_fwd(const float* x, float* y, long n, const unsigned char* mask_in, unsigned char* mask_out,
                        float p, unsigned long long seed, unsigned long long offset, void* stream);
int pe_gelu_dropout_bwd(const float* x, const float* dy, const unsigned char* mask, float p, float* dx, long n,
                        void* stream);

/* ---- resampler (meldataset.py:621-627 -> torchaudio.functional.resample defaults) ----
 * sinc_interp_hann, lowpass_filter_width 6, rolloff 0.99; y has ceil(new * n_in / orig) samples. */
typedef struct pe_resample_plan pe_resample_plan;
int pe_resample_plan_create(pe_resample_plan** plan, int orig_freq, int new_freq, int lowpass_filter_width,
                            float rolloff);
int pe_resample_plan_destroy(pe_resample_plan* plan);
long pe_resample_out_len(const pe_resample_plan* plan, long n_in);
int pe_resample_forward(const pe_resample_plan* plan, const float* x, int batch, int n_in, long x_stride, float* y,
                        long y_stride, int n_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PITCHEXTRACTOR_HIP_H */
