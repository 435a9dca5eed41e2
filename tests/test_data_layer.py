"""Host-side data-layer arithmetic (bit-exact: indices, interpolation, padding) -- no GPU needed."""
import os
import random
import struct

import numpy as np
import pytest
import torch

from oracle import train_ref
from pitchextractor_amd import meldataset as md
from pitchextractor_amd import synthetic


def write_wav(path, data, sr, fmt="pcm16"):
    data = np.asarray(data)
    ch = 1 if data.ndim == 1 else data.shape[1]
    if fmt == "pcm16":
        raw = (np.clip(data, -1, 1) * 32767).astype("<i2").tobytes(); tag, bits = 1, 16
    elif fmt == "float32":
        raw = data.astype("<f4").tobytes(); tag, bits = 3, 32
    else:
        raise ValueError(fmt)
    align = ch * bits // 8
    with open(path, "wb") as fh:
        fh.write(b"RIFF" + struct.pack("<I", 36 + len(raw)) + b"WAVE")
        fh.write(b"fmt " + struct.pack("<IHHIIHH", 16, tag, ch, sr, sr * align, align, bits))
        fh.write(b"data" + struct.pack("<I", len(raw)) + raw)


def test_align_length_matches_reference_golden(golden_dir):
    D = np.load(golden_dir / "data_golden.npz")
    for n in (159, 161, 163, 200, 7, 1):
        np.testing.assert_array_equal(md.align_length(D[f"align_in_{n}"], 161), D[f"align_out_{n}"])
    np.testing.assert_array_equal(md.align_length([0, 0, 100, 110, 120, 0, 0, 130, 140, 150], 7), D["align_probe"])
    assert md.align_length(np.zeros(0), 5).tolist() == [0] * 5 and md.align_length([1.0, 2.0], 0).shape == (0,)


def test_segment_plan_arithmetic():
    # 58 624 samples @24 kHz, 107 722 @44.1 kHz
    rng = random.Random(3)
    s, seg, full = md.segment_plan(200000, 24000, 24000, 300, 1024, 192, rng)
    assert seg == 58624 and not full and 0 <= s <= 200000 - 58624
    _, seg44, _ = md.segment_plan(500000, 44100, 24000, 300, 1024, 192, rng)
    assert seg44 == 107722
    assert md.segment_plan(48000, 24000, 24000, 300, 1024, 192, rng) == (0, 58624, True)   # short file: whole
    assert md.segment_plan(0, 24000, 24000, 300, 1024, 192, rng) == (0, None, True)
    # same draws as the reference's random.randint(0, total - segment)
    r1, r2 = random.Random(1), random.Random(1)
    assert md.segment_plan(100000, 24000, 24000, 300, 1024, 192, r1)[0] == r2.randint(0, 100000 - 58624)


@pytest.mark.parametrize("fmt", ["pcm16", "float32"])
def test_wav_reader_roundtrip(tmp_path, fmt):
    wave, _, _ = synthetic.utterance(0, duration=0.5)
    p = tmp_path / "a.wav"
    write_wav(p, wave, 24000, fmt)
    assert md.wav_info(p) == (12000, 24000, 1)
    data, sr = md.read_wav(p)
    assert sr == 24000 and data.shape == (12000,)
    tol = 1e-7 if fmt == "float32" else 1e-4        # 16-bit quantisation
    assert np.abs(data - wave).max() <= tol
    part, _ = md.read_wav(p, start=1000, frames=500)
    np.testing.assert_array_equal(part, data[1000:1500])
    stereo = np.stack([wave, -wave], axis=1)
    write_wav(p, stereo, 24000, fmt)
    d2, _ = md.read_wav(p)
    assert d2.shape == (12000, 2) and md.wav_info(p) == (12000, 24000, 2)


def test_dataset_item_and_collater(tmp_path):
    rng = np.random.default_rng(0)
    lines = []
    for i, dur in enumerate((2.0, 3.0, 0.9)):
        wave, f0, _ = synthetic.utterance(i, duration=dur)
        p = tmp_path / f"u{i}.wav"
        write_wav(p, wave, 24000, "float32")
        np.save(str(p) + "_f0.npy", f0[: len(f0) - (i % 2)])          # legacy cache name, one frame short
        lines.append(f"{p}|0\n")
    ds = md.MelDataset(lines, verbose=False)     # the HIP transform is built lazily: everything but to_melspec runs here
    items = [ds[i] for i in range(3)]
    w0, f0_0, s0, c0, sr0 = items[0]
    assert sr0 == 24000
    assert w0.shape == (48000,) and f0_0.shape == (161,) and c0 == 0
    w1, f0_1, s1, c1, _ = items[1]
    assert w1.shape == (58624,)                                       # pre-cropped segment
    assert f0_1.shape == (192,) and 0 <= c1 < 196 - 192 + 1
    assert torch.equal(s1, (f0_1 == 0).float())
    w2, f0_2, _, _, _ = items[2]
    assert f0_2.shape == (1 + 21600 // 300,)
    waves, lengths, crops, f0s, sils, src_sr = md.Collater()(items)
    assert src_sr == 24000
    assert waves.shape == (3, 58624) and lengths.tolist() == [48000, 58624, 21600]
    assert f0s.shape == (3, 192) and (f0s[0, 161:] == 0).all() and (sils[0, 161:] == 0).all()
    # reference-shaped items go through the same padding as the oracle's collate
    mel_items = [(torch.ones(80, L) * (i + 1), torch.full((L,), 100.0), torch.zeros(L)) for i, L in enumerate((161, 192, 100))]
    mels, f0b, silb = md.Collater()(mel_items)
    rm, rf, rs = train_ref.collate([(m.numpy(), f.numpy(), s.numpy()) for m, f, s in mel_items])
    np.testing.assert_array_equal(mels.numpy(), rm)
    np.testing.assert_array_equal(f0b.numpy(), rf)
    np.testing.assert_array_equal(silb.numpy(), rs)


def test_missing_labels_fail_loudly(tmp_path):
    wave, _, _ = synthetic.utterance(0, duration=1.0)
    p = tmp_path / "x.wav"
    write_wav(p, wave, 24000)
    ds = md.MelDataset([f"{p}|0\n"], verbose=False)
    with pytest.raises(RuntimeError):
        ds._f0_for(str(p), wave, 0, None)


def test_cache_identifier_matches_reference_golden(golden_dir):
    """``F0Extractor.cache_identifier`` as the imported reference computes it (tests/golden/make_golden.py
    cache_id): the shipped config.yml names its caches ``<wav>_f0-swiftf0.npy``."""
    import json
    cases = json.loads((golden_dir / "cache_id_golden.json").read_text())
    assert cases["shipped_yaml"]["cache_identifier"] == "-swiftf0"
    for tag, case in cases.items():
        assert md.f0_cache_identifier(case["f0_params"]) == case["cache_identifier"], tag
    assert md.f0_cache_identifier({"cache_identifier": "-custom"}) == "-custom"


def test_f0_cache_contract(tmp_path):
    """meldataset.py:566-604: identifier-named cache + matching .json first, legacy ``_f0.npy`` second; a cache
    computed for another hop / rate / backend chain is not used (and, unlike the reference, not deleted)."""
    import json
    wave, f0, _ = synthetic.utterance(0, duration=1.0)
    p = str(tmp_path / "x.wav")
    write_wav(p, wave, 24000)
    params = {"backends": {"swiftf0": {"type": "swiftf0", "enabled": True}}}
    ds = md.MelDataset([f"{p}|0\n"], f0_params=params, verbose=False)
    assert ds.f0_cache_suffix == "_f0-swiftf0.npy" and ds.f0_meta_suffix == "_f0-swiftf0.json"
    meta = {"cache_identifier": "-swiftf0", "backend": "swiftf0", "sample_rate": 24000, "hop_length": 300}
    legacy, named = f0 + 1.0, f0 + 2.0
    np.save(p + "_f0.npy", legacy)
    np.save(p + "_f0-swiftf0.npy", named)
    # (1) no metadata next to the named cache -> not trusted; the legacy file is the label source
    np.testing.assert_array_equal(ds._load_cached_f0(p), legacy)
    # (2) matching metadata -> the named cache wins over the legacy file (which sorts first in a glob)
    json.dump(meta, open(p + "_f0-swiftf0.json", "w"))
    np.testing.assert_array_equal(ds._load_cached_f0(p), named)
    # (3) metadata from another hop length / sample rate / backend chain -> skipped, files stay on disk
    for key, val in (("hop_length", 256), ("sample_rate", 22050), ("cache_identifier", "-crepe")):
        json.dump(dict(meta, **{key: val}), open(p + "_f0-swiftf0.json", "w"))
        np.testing.assert_array_equal(ds._load_cached_f0(p), legacy)
        assert os.path.isfile(p + "_f0-swiftf0.npy") and os.path.isfile(p + "_f0-swiftf0.json")
    # (4) a cache of some other backend chain is never picked up by accident
    os.remove(p + "_f0.npy"); os.remove(p + "_f0-swiftf0.npy"); os.remove(p + "_f0-swiftf0.json")
    np.save(p + "_f0-crepe.npy", f0)
    assert ds._load_cached_f0(p) is None
    with pytest.raises(RuntimeError):
        ds._f0_for(p, wave, 0, None)
    # (5) unreadable metadata -> named cache skipped
    np.save(p + "_f0-swiftf0.npy", named)
    open(p + "_f0-swiftf0.json", "w").write("{not json")
    assert ds._load_cached_f0(p) is None


def test_resampler_oracle_properties():
    """torchaudio is absent (parity unpinned): pin the float64 restatement by construction."""
    from oracle import resample_ref as rr
    k, width, orig, new = rr.sinc_resample_kernel(44100, 24000)
    assert (k.shape, width, orig, new) == ((80, 171), 12, 147, 80)
    assert np.abs(k.sum(axis=1) - 1.0).max() < 1e-3                                # unit DC gain per phase
    t = np.arange(44100) / 44100.0
    y = rr.resample(np.sin(2 * np.pi * 440 * t), 44100, 24000)
    assert y.shape == (24000,)
    ref = np.sin(2 * np.pi * 440 * np.arange(24000) / 24000.0)
    assert np.abs(y - ref)[200:-200].max() < 1e-3
    assert rr.resample(np.ones(1000), 24000, 24000).shape == (1000,)
    assert rr.resample(np.zeros(107722), 44100, 24000).shape == (58625,)            # ceil(80 * 107722 / 147)


def _mel_cache_fixture(tmp_path, n_files=3):
    """Whole-file items (2 s < the 58 624-sample segment) with legacy F0 caches; file 1 gets a spectrogram cache
    written exactly as meldataset.py:780-786 writes it (np.save + json.dump(sort_keys=True))."""
    import json
    lines, paths = [], []
    for i in range(n_files):
        wave, f0, _ = synthetic.utterance(10 + i, duration=2.0)
        p = str(tmp_path / f"c{i}.wav")
        write_wav(p, wave, 24000, "float32")
        np.save(p + "_f0.npy", f0)
        lines.append(f"{p}|0\n")
        paths.append(p)
    rng = np.random.default_rng(5)
    cached = rng.uniform(1e-4, 3.0, size=(80, 161)).astype(np.float32)     # recognisably NOT what the device computes
    np.save(paths[1] + "_mel.npy", cached)
    meta = {"audio_sample_rate": 24000, "audio_num_samples": 48000, "audio_num_channels": 1,
            "dataset_sample_rate": 24000,
            "mel_params": {"sample_rate": 24000, "n_mels": 80, "n_fft": 1024, "win_length": 1024, "hop_length": 300}}
    with open(paths[1] + "_mel_meta.json", "w", encoding="utf-8") as fh:
        json.dump(meta, fh, sort_keys=True)
    return lines, paths, cached, meta


def test_mel_cache_contract_read_only(tmp_path):
    """meldataset.py:679-741: a whole-file item takes its spectrogram from <wav>_mel.npy only when
    <wav>_mel_meta.json equals the expected metadata; normalisation is the reference's float32 expression."""
    import json
    lines, paths, cached, meta = _mel_cache_fixture(tmp_path)
    ds = md.MelDataset(lines, verbose=False)
    assert ds._build_mel_metadata(48000, 24000) == meta
    plain, hit = ds[0], ds[1]
    assert len(plain) == 5 and len(hit) == 6
    want = (torch.log(1e-5 + torch.from_numpy(cached)) - (-4.0)) / 4.0            # meldataset.py:650
    assert torch.equal(hit[5], want) and hit[1].shape == (161,)
    out = md.Collater()([plain, hit, ds[2]])
    assert len(out) == 8 and out[6].tolist() == [1] and out[7].shape == (1, 80, 192)
    assert torch.equal(out[7][0, :, :161], want) and not out[7][0, :, 161:].any()  # zero padding after normalisation
    assert len(md.Collater()([plain, ds[2]])) == 6                                 # no cache rows: the usual tuple

    # a cached spectrogram of another length drives the label length, as mel_tensor.size(1) does (:651-656)
    np.save(paths[2] + "_mel.npy", cached[:, :150])
    with open(paths[2] + "_mel_meta.json", "w", encoding="utf-8") as fh:
        json.dump(meta, fh, sort_keys=True)
    assert ds[2][5].shape == (80, 150) and ds[2][1].shape == (150,)

    # mismatching metadata (other hop), missing metadata, augmentation: not used -- and nothing is deleted
    bad = dict(meta, mel_params=dict(meta["mel_params"], hop_length=256))
    with open(paths[1] + "_mel_meta.json", "w", encoding="utf-8") as fh:
        json.dump(bad, fh, sort_keys=True)
    assert len(ds[1]) == 5 and os.path.isfile(paths[1] + "_mel.npy") and os.path.isfile(paths[1] + "_mel_meta.json")
    os.remove(paths[2] + "_mel_meta.json")
    assert len(ds[2]) == 5 and os.path.isfile(paths[2] + "_mel.npy")
    with open(paths[1] + "_mel_meta.json", "w", encoding="utf-8") as fh:
        json.dump(meta, fh, sort_keys=True)
    assert len(ds[1]) == 6
    assert len(md.MelDataset(lines, verbose=False, data_augmentation=True)[1]) == 5


def test_mel_cache_not_used_for_pre_cropped_segments(tmp_path):
    """cache_key is the path only for whole files (meldataset.py:236): a 3 s file is pre-cropped, so its cache is ignored."""
    import json
    wave, f0, _ = synthetic.utterance(3, duration=3.0)
    p = str(tmp_path / "long.wav")
    write_wav(p, wave, 24000, "float32")
    np.save(p + "_f0.npy", f0)
    np.save(p + "_mel.npy", np.ones((80, 241), np.float32))
    ds = md.MelDataset([f"{p}|0\n"], verbose=False)
    with open(p + "_mel_meta.json", "w", encoding="utf-8") as fh:
        json.dump(ds._build_mel_metadata(72000, 24000), fh, sort_keys=True)
    assert len(ds[0]) == 5
