"""CPU checks of the mel oracle: the two independent restatements must agree."""
import numpy as np

from oracle import mel_ref
from pitchextractor_amd import synthetic


def test_frame_count_and_reflect():
    assert mel_ref.num_frames(48000) == 161
    assert mel_ref.num_frames(58624) == 196
    idx = mel_ref.reflect_index(np.array([-3, -1, 0, 9, 10, 12]), 10)
    assert idx.tolist() == [3, 1, 0, 9, 8, 6]
    fr = mel_ref.frame_signal(np.arange(2000.0), 1024, 300)
    assert fr.shape == (7, 1024)
    assert fr[0, 0] == 512 and fr[0, 512] == 0 and fr[0, 511] == 1
    assert fr[6, -1] == 2 * 1999 - (6 * 300 + 1023 - 512)


def test_filterbank_shape_and_sparsity():
    fb = mel_ref.mel_filterbank()
    assert fb.shape == (513, 80)
    assert (fb >= 0).all() and fb.max() <= 1.0
    # HTK triangles overlap by half: at most two filters are non-zero per bin
    assert ((fb > 0).sum(axis=1) <= 2).all()


def test_direct_dft_matches_rfft():
    wave, _, _ = synthetic.utterance(0, duration=0.2)
    a = mel_ref.mel_spectrogram(wave, direct_dft=True)
    b = mel_ref.mel_spectrogram(wave, direct_dft=False)
    assert a.shape == (80, 17)
    assert np.allclose(a, b, rtol=1e-9, atol=1e-12)


def test_float64_oracle_matches_torch_stft():
    wave, _, _ = synthetic.utterance(3, duration=1.0)
    ref64 = mel_ref.mel_spectrogram(wave)
    ref32 = mel_ref.mel_spectrogram_torch_stft(wave)
    assert ref32.shape == ref64.shape == (80, 81)
    assert np.abs(ref32 - ref64).max() <= 1e-4 * ref64.max()
    # bins near the 1e-5 log floor sit on the fp32 FFT rounding floor (|dX| ~ 1e-8 |X|max):
    # two float32-vs-float64 implementations differ by a few 1e-4 there after the log
    strong = ref64 >= 1e-2
    assert np.abs(ref32 - ref64)[strong].max() <= 1e-4 * ref64[strong].min() or \
        (np.abs(ref32 - ref64)[strong] <= 1e-4 * ref64[strong]).all()
    a, b = mel_ref.log_normalise(ref64), mel_ref.log_normalise(ref32.astype(np.float64))
    assert np.abs(a - b).max() < 1e-3


def test_log_normalise_constants():
    # meldataset.py:650 with mean, std = -4, 4
    assert np.isclose(mel_ref.log_normalise(np.array([0.0]))[0], (np.log(1e-5) + 4) / 4)
