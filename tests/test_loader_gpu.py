"""End-to-end data path on the GPU: wav files -> build_dataloader -> device mel batch, against the
float64 mel oracle with the same crops; plus the ragged / frame-offset mel entry point."""
import random

import numpy as np
import pytest
import torch

from oracle import mel_ref, train_ref
from pitchextractor_amd import meldataset as md
from pitchextractor_amd import synthetic
from pitchextractor_amd.mel import MelSpectrogram
from tests.test_data_layer import write_wav

pytestmark = pytest.mark.gpu


def test_ragged_mel_with_frame_offsets(hip_device):
    rng = np.random.default_rng(0)
    lens = [58624, 48000, 21600, 600, 300]
    waves = np.zeros((5, 58624), np.float32)
    for i, n in enumerate(lens):
        waves[i, :n] = 0.2 * rng.standard_normal(n)
    starts = [3, 0, 0, 0, 0]
    tf = MelSpectrogram(**mel_ref.DEFAULT_MEL_PARAMS)
    out = tf.log_mel_ragged(torch.from_numpy(waves).to(hip_device),
                            torch.tensor(lens, dtype=torch.int32, device=hip_device),
                            torch.tensor(starts, dtype=torch.int32, device=hip_device)).cpu().numpy()
    assert out.shape == (5, 1, 80, 192)
    for i, n in enumerate(lens):
        if n <= 512:
            assert (out[i] == 0).all()                      # too short for reflect padding: padding only
            continue
        ref = mel_ref.log_mel(waves[i, :n])[:, starts[i]:starts[i] + 192]
        L = ref.shape[1]
        assert np.abs(out[i, 0, :, :L] - ref).max() <= 1e-3, i
        assert (out[i, 0, :, L:] == 0).all()


def test_dataloader_batches_match_oracle(tmp_path, hip_device):
    lines = []
    for i, dur in enumerate((2.0, 3.0, 0.9, 2.6, 2.0, 4.0)):
        wave, f0, _ = synthetic.utterance(i, duration=dur)
        p = tmp_path / f"u{i}.wav"
        write_wav(p, wave, 24000, "float32")
        np.save(str(p) + "_f0.npy", f0)
        lines.append(f"{p}|0\n")
    cfg = {"mel_params": {"sample_rate": 24000, "win_len": 1024, "n_fft": 1024, "n_mels": 80, "hop_length": 300},
           "dataloader": {"start_method": None}, "verbose": False}
    loader = md.build_dataloader(lines, validation=True, batch_size=3, num_workers=0, device="cuda:0",
                                 dataset_config=cfg)
    assert len(loader) == 2
    np.random.seed(5); random.seed(5)
    got = [(m.cpu().numpy(), f.cpu().numpy(), s.cpu().numpy()) for m, f, s in loader]
    # replay the same draws on the host and push each item through the float64 oracle
    np.random.seed(5); random.seed(5)
    ds = loader.dataset
    for bi in range(2):
        items = []
        for i in range(3 * bi, 3 * bi + 3):
            wave, f0, sil, crop = ds.path_to_wave_and_label(ds.data_list[i])
            mel = mel_ref.log_mel(wave)[:, crop:crop + 192].astype(np.float32)
            items.append((mel, f0, sil))
        rm, rf, rs = train_ref.collate(items)
        m, f, s = got[bi]
        assert m.shape == (3, 1, 80, 192)
        assert np.abs(m - rm).max() <= 1e-3
        np.testing.assert_array_equal(f, rf)
        np.testing.assert_array_equal(s, rs)


def test_resampler_matches_oracle(hip_device):
    from oracle import resample_ref as rr
    from pitchextractor_amd.resample import Resampler
    rng = np.random.default_rng(0)
    x = (0.3 * rng.standard_normal((3, 9001))).astype(np.float32)
    for orig, new in ((44100, 24000), (16000, 24000), (48000, 24000), (22050, 24000)):
        rs = Resampler(orig, new)
        y = rs(torch.from_numpy(x).to(hip_device)).cpu().numpy()
        for i in range(3):
            ref = rr.resample(x[i], orig, new)
            assert y[i].shape == ref.shape == (rs.out_len(9001),)
            assert np.abs(y[i] - ref).max() <= 2e-6 * max(1.0, np.abs(ref).max())
    assert Resampler(24000, 24000)(torch.ones(5, device=hip_device)).shape == (5,)


def test_dataloader_resamples_44k_files(tmp_path, hip_device):
    """Config 5's data path: 44.1 kHz files -> pre-crop at the source rate -> GPU resample -> mel."""
    from oracle import resample_ref as rr
    lines = []
    for i, dur in enumerate((2.0, 3.1)):
        n = int(dur * 44100)
        t = np.arange(n) / 44100.0
        wave = (0.5 * np.sin(2 * np.pi * (200 + 50 * i) * t)).astype(np.float32)
        p = tmp_path / f"h{i}.wav"
        write_wav(p, wave, 44100, "float32")
        np.save(str(p) + "_f0.npy", np.full(1 + int(dur * 24000) // 300, 200.0 + 50 * i, np.float32))
        lines.append(f"{p}|0\n")
    cfg = {"mel_params": {"sample_rate": 24000, "win_len": 1024, "n_fft": 1024, "n_mels": 80, "hop_length": 300},
           "verbose": False}
    loader = md.build_dataloader(lines, validation=True, batch_size=2, num_workers=0, device="cuda:0",
                                 dataset_config=cfg)
    np.random.seed(7); random.seed(7)
    (mels, f0s, sils), = list(loader)
    np.random.seed(7); random.seed(7)
    ds = loader.dataset
    for i in range(2):
        wave, f0, sil, crop = ds.path_to_wave_and_label(ds.data_list[i])
        assert ds._last_sr == 44100
        res = rr.resample(wave, 44100, 24000)
        ref = mel_ref.log_mel(res)[:, crop:crop + 192]
        L = ref.shape[1]
        assert np.abs(mels[i, 0, :, :L].cpu().numpy() - ref).max() <= 2e-3
        np.testing.assert_array_equal(f0s[i, :len(f0)].cpu().numpy(), f0)
    assert mels.shape == (2, 1, 80, 192)


def test_dataloader_takes_cached_spectrograms(tmp_path, hip_device):
    """meldataset.py:640-650: a whole-file item with a valid <wav>_mel.npy cache trains on the cached spectrogram
    (bit-exact: normalised with the reference's float32 expression), its neighbours on the device mel."""
    from tests.test_data_layer import _mel_cache_fixture
    lines, paths, cached, _ = _mel_cache_fixture(tmp_path)
    cfg = {"mel_params": {"sample_rate": 24000, "win_len": 1024, "n_fft": 1024, "n_mels": 80, "hop_length": 300},
           "verbose": False}
    loader = md.build_dataloader(lines, validation=True, batch_size=3, num_workers=2, device="cuda:0",
                                 dataset_config=cfg)           # worker processes + pinned memory carry the extra tensors
    (mels, f0s, sils), = list(loader)
    mels = mels.cpu()
    want = (torch.log(1e-5 + torch.from_numpy(cached)) + 4.0) / 4.0
    assert torch.equal(mels[1, 0, :, :161], want) and not mels[1, 0, :, 161:].any()
    for i in (0, 2):
        wave, _ = md.read_wav(paths[i])
        ref = mel_ref.log_mel(wave)
        assert np.abs(mels[i, 0, :, :161].numpy() - ref).max() <= 1e-3
    assert f0s.shape == (3, 192) and sils.shape == (3, 192)


def test_full_size_resample_and_mel_config4_shape(hip_device):
    """BASELINE config[4]'s front end at its real size: 256 utterances of 4 s at 44.1 kHz (176 400 samples) ->
    GPU resample (147 -> 80) -> ragged mel with a random 192-frame crop per item.  Size-independent properties
    (power-of-two scaling is exact, tiled replicas agree bit for bit) plus three rows against the float64 oracles."""
    from oracle import resample_ref as rr
    from pitchextractor_amd.resample import Resampler
    rng = np.random.default_rng(3)
    n_src = 4 * 44100
    t = np.arange(n_src) / 44100.0
    base = np.stack([(0.4 * np.sin(2 * np.pi * (110.0 + 37.0 * i) * t * (1.0 + 0.1 * t))
                      + 0.01 * rng.standard_normal(n_src)).astype(np.float32) for i in range(8)])
    x = torch.from_numpy(np.tile(base, (32, 1))).to(hip_device)                  # (256, 176400)
    rs = Resampler(44100, 24000)
    y = rs(x)
    assert y.shape == (256, rs.out_len(n_src)) == (256, 96000)
    assert torch.equal(rs(x * 2.0), y * 2.0)                                      # linear, power-of-two exact
    assert torch.equal(y[:8], y[248:])                                            # no cross-row leakage at full size
    for i in (0, 3, 7):
        ref = rr.resample(base[i], 44100, 24000)
        assert np.abs(y[i].cpu().numpy() - ref).max() <= 2e-6 * max(1.0, np.abs(ref).max())
    tf = MelSpectrogram(**mel_ref.DEFAULT_MEL_PARAMS)
    lengths = torch.full((256,), 96000, dtype=torch.int32, device=hip_device)
    crops_host = np.tile(rng.integers(0, 321 - 192, size=8), 32).astype(np.int32)
    crops = torch.from_numpy(crops_host).to(hip_device)
    mels = tf.log_mel_ragged(y, lengths, crops, max_frames=192)
    assert mels.shape == (256, 1, 80, 192) and torch.equal(mels[:8], mels[248:])
    for i in (0, 3, 7):
        ref = mel_ref.log_mel(rr.resample(base[i], 44100, 24000))[:, crops_host[i]:crops_host[i] + 192]
        assert np.abs(mels[i, 0].cpu().numpy() - ref).max() <= 2e-3
