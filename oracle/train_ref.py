"""CPU oracle of the optimiser, schedule, data-layer arithmetic and one training step
(TEST INFRASTRUCTURE ONLY).

* :func:`one_cycle` / :class:`AdamWRef` restate ``build_optimizer`` (reference
  optimizers.py:50-76 as called from train.py:93-102): AdamW(lr 1e-4, wd 5e-4, betas (0.9, 0.98),
  eps 1e-9) under OneCycleLR(max_lr, pct_start 0, div 25, final_div 5, cos anneal, cycling beta1
  0.85 -> 0.95).  Pinned by tests/golden/optimizer_golden.npz.
* :func:`align_length` restates ``F0Extractor.align_length`` (f0_backends.py:788-806) and
  :func:`collate` the Collater of meldataset.py:790-826.
* :class:`CpuTrainer` runs the reference-equivalent fp32 step (trainer.py:219-252) on the
  functional model oracle with torch autograd; it is what ``bench.py`` times as ``cpu_baseline``.
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np
import torch

from . import model_ref


# --------------------------------------------------------------------------- schedule / optimiser
def _cos_anneal(start, end, pct):
    return end + (start - end) / 2.0 * (math.cos(math.pi * pct) + 1.0)


def one_cycle(step: int, total_steps: int, max_lr: float, pct_start: float = 0.0, div_factor: float = 25.0,
              final_div_factor: float = 5.0, base_momentum: float = 0.85, max_momentum: float = 0.95):
    """(lr, beta1) in force for optimiser step ``step`` (0-based) -- torch OneCycleLR, two phases."""
    initial_lr = max_lr / div_factor
    min_lr = initial_lr / final_div_factor
    phases = [(float(pct_start * total_steps) - 1.0, initial_lr, max_lr, max_momentum, base_momentum),
              (float(total_steps - 1), max_lr, min_lr, base_momentum, max_momentum)]
    start = 0.0
    for i, (end, lr_a, lr_b, mom_a, mom_b) in enumerate(phases):
        if step <= end or i == len(phases) - 1:
            pct = (step - start) / (end - start)
            return _cos_anneal(lr_a, lr_b, pct), _cos_anneal(mom_a, mom_b, pct)
        start = end
    raise AssertionError


class AdamWRef:
    """torch.optim.AdamW update in float32 numpy, one flat vector."""

    def __init__(self, n, beta2=0.98, eps=1e-9, weight_decay=5e-4):
        self.m = np.zeros(n, dtype=np.float32)
        self.v = np.zeros(n, dtype=np.float32)
        self.t = 0
        self.beta2, self.eps, self.wd = beta2, eps, weight_decay

    def step(self, p, g, lr, beta1):
        self.t += 1
        f = np.float32
        p = p * f(1.0 - lr * self.wd)
        self.m = self.m + f(1.0 - beta1) * (g - self.m)
        self.v = self.v * f(self.beta2) + f(1.0 - self.beta2) * (g * g)
        bc1 = 1.0 - beta1 ** self.t
        bc2 = 1.0 - self.beta2 ** self.t
        denom = np.sqrt(self.v) / f(math.sqrt(bc2)) + f(self.eps)
        return (p - f(lr / bc1) * (self.m / denom)).astype(np.float32)


# --------------------------------------------------------------------------- data-layer arithmetic
def align_length(values, target_frames: int) -> np.ndarray:
    """f0_backends.py:788-806: float64 linear interpolation, then zero every target frame whose
    round()-nearest source frame is unvoiced (interpolation happens BEFORE masking)."""
    values = np.asarray(values, dtype=np.float64)
    if target_frames <= 0:
        return np.zeros((0,), dtype=np.float32)
    n = values.size
    if n == target_frames:
        return values.astype(np.float32)
    if n == 0:
        return np.zeros((target_frames,), dtype=np.float32)
    src = np.linspace(0.0, n - 1, num=n)
    dst = np.linspace(0.0, n - 1, num=target_frames)
    out = np.interp(dst, src, values)
    unvoiced = values == 0.0
    if unvoiced.any():
        nearest = np.clip(np.round(dst).astype(int), 0, n - 1)
        out[unvoiced[nearest]] = 0.0
    return out.astype(np.float32)


def collate(items, max_len: int = 192):
    """meldataset.py:804-826: zero-pad (mel (80,L), f0 (L,), sil (L,)) items to ``max_len`` frames."""
    B = len(items)
    n_mels = items[0][0].shape[0]
    mels = np.zeros((B, n_mels, max_len), dtype=np.float32)
    f0s = np.zeros((B, max_len), dtype=np.float32)
    sils = np.zeros((B, max_len), dtype=np.float32)
    for i, (mel, f0, sil) in enumerate(items):
        L = mel.shape[1]
        mels[i, :, :L] = mel
        f0s[i, :L] = f0
        sils[i, :L] = sil
    return mels[:, None], f0s, sils


# --------------------------------------------------------------------------- reference-equivalent CPU step
class CpuTrainer:
    """fp32 CPU restatement of ``Trainer.run`` (trainer.py:219-252), dropout disabled."""

    def __init__(self, state: Dict[str, torch.Tensor], seq_cfg: dict, max_lr=3e-4, total_steps=800,
                 lambda_f0=0.1, fused_lstm=False):
        self.fused_lstm = fused_lstm
        self.state = {k: v.clone() for k, v in state.items()}
        self.names = [k for k, v in self.state.items() if v.dtype.is_floating_point and
                      not k.endswith(("running_mean", "running_var"))]
        self.seq_cfg, self.max_lr, self.total, self.lam = dict(seq_cfg), max_lr, total_steps, lambda_f0
        self.opt = {k: AdamWRef(self.state[k].numel()) for k in self.names}
        self.k = 0

    def run(self, batch):
        x, f0, sil = batch
        params = {k: self.state[k].detach().clone().requires_grad_(True) for k in self.names}
        live = dict(self.state)
        live.update(params)
        new_stats = {}
        cls, det = model_ref.jdcnet_forward(live, x.transpose(-1, -2), self.seq_cfg, train=True,
                                            new_stats=new_stats, fused_lstm=self.fused_lstm)
        if cls.shape[-1] == 1:
            loss, lf0, lsil = model_ref.jdc_loss(cls, det, f0, sil, self.lam)
        else:                                    # build-defined 360-bin classification target
            loss, lf0, lsil = model_ref.jdc_bins_loss(cls, det, f0, sil, self.lam)
        loss.backward()
        lr, beta1 = one_cycle(self.k, self.total, self.max_lr)
        for k in self.names:
            p = self.state[k].numpy().reshape(-1)
            g = params[k].grad.numpy().reshape(-1)
            self.state[k] = torch.from_numpy(self.opt[k].step(p, g, lr, beta1)).view_as(self.state[k])
        self.state.update({k: v.detach() for k, v in new_stats.items()})
        self.k += 1
        return {"loss": loss.item(), "f0": lf0.item(), "sil": lsil.item()}
