import sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from oracle import mel_ref
from pitchextractor_amd.mel import MelSpectrogram
dev = torch.device("cuda:0")
tf = MelSpectrogram(**mel_ref.DEFAULT_MEL_PARAMS)
for n in [513, 600, 899, 1000, 1500, 4800, 9000]:
    rng = np.random.default_rng(n)
    wave = (0.3 * rng.standard_normal(n)).astype(np.float32)
    got = tf(torch.from_numpy(wave).to(dev)).cpu().numpy().astype(np.float64)
    ref = mel_ref.mel_spectrogram(wave)
    print(n, got.shape, ref.shape, flush=True)
    if got.shape != ref.shape: continue
    rel = np.abs(got - ref).max(axis=0) / ref.max(axis=0)
    print(n, got.shape, " ".join(f"{x:.1e}" for x in rel))
# impulse probes: which sample does each frame position see?
n = 2000
for pos in [0, 1, 511, 512, 1998, 1999, 1700]:
    wave = np.zeros(n, np.float32); wave[pos] = 1.0
    got = tf(torch.from_numpy(wave).to(dev)).cpu().numpy().astype(np.float64)
    ref = mel_ref.mel_spectrogram(wave)
    rel = np.abs(got - ref).max(axis=0) / (ref.max(axis=0) + 1e-30)
    print("impulse", pos, " ".join(f"{x:.1e}" for x in rel))
