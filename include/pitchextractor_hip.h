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
 *     hipError_t of a failed runtime call.
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

#ifdef __cplusplus
}
#endif
#endif /* PITCHEXTRACTOR_HIP_H */
