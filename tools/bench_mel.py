"""Micro-benchmark of the mel kernel alone (B=256, 2 s @ 24 kHz): HBM roofline fraction, warm (one batch re-run:
62.6 MB sits in the 256 MiB Infinity Cache) and cold (8 distinct batches + outputs = 0.6 GB cycled, so every
launch reads its audio from HBM)."""
import json
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pitchextractor_amd import synthetic
from pitchextractor_amd.mel import MelSpectrogram

B, reps, NB = 256, 200, 8
waves, _, _ = synthetic.batch(0, 8)
dev = torch.device("cuda:0")
base = torch.from_numpy(np.tile(waves, (B // 8, 1))).to(dev)
xs = [base.roll(i, 0).clone() for i in range(NB)]
outs = [torch.empty((B, 1, 80, 192), device=dev) for _ in range(NB)]
tf = MelSpectrogram(sample_rate=24000, n_fft=1024, win_length=1024, hop_length=300, n_mels=80)


def run(cold):
    for i in range(10):
        tf.log_mel_batch(xs[i % NB if cold else 0], out=outs[i % NB if cold else 0])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        tf.log_mel_batch(xs[i % NB if cold else 0], out=outs[i % NB if cold else 0])
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


frames = B * 161
alg_bytes = 1520 * frames
for cold in (False, True):
    ms = run(cold)
    print(json.dumps({"kernel": "mel_fwd", "cache": "cold (8 batches cycled)" if cold else "warm (one batch)",
                      "ms": ms, "frames_per_s": frames / ms * 1e3, "alg_GBps": alg_bytes / ms / 1e6,
                      "frac_of_8TBps": alg_bytes / ms / 1e6 / 8000}))
