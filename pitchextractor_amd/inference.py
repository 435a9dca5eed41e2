"""Inference recipe of the reference's evaluation notebooks (Utils/dynamic_pitch_behavior.ipynb, code
cell 5: ``load_model`` / ``waveform_to_mel`` / ``predict_f0``) on the HIP path.

``predict_f0`` keeps the notebook's semantics exactly -- mel -> chunks of ``chunk_size`` frames every
``chunk_size - overlap`` frames, the last chunk zero-padded, every chunk contributing its first
``end - start`` predictions, overlaps NOT blended (so the result is longer than the frame count) -- but
runs all chunks of an utterance as ONE batch through the network instead of one forward per chunk
(eval-mode BatchNorm makes samples independent, bit for bit).
"""
from __future__ import annotations

import re
from pathlib import Path

import numpy as np
import torch

from . import ops
from .mel import DEFAULT_MEL_PARAMS, LOG_EPS, MEL_MEAN, MEL_STD, MelSpectrogram
from .model import JDCNet


def infer_model_config(model_state: dict) -> tuple[int, dict]:
    """(num_class, sequence_model_config) read off a reference-format ``state_dict``."""
    num_class = int(model_state["classifier.weight"].shape[0]) if "classifier.weight" in model_state else 722
    keys = list(model_state)
    if any(k.startswith("sequence_classifier.model.weight_ih_l") for k in keys):
        layers = 1 + max(int(re.search(r"weight_ih_l(\d+)", k).group(1)) for k in keys
                         if k.startswith("sequence_classifier.model.weight_ih_l"))
        hidden = int(model_state["sequence_classifier.model.weight_hh_l0"].shape[1])
        bidir = "sequence_classifier.model.weight_ih_l0_reverse" in model_state
        cfg = {"model_type": "bilstm", "hidden_size": hidden, "num_layers": layers, "bidirectional": bidir}
    elif any(k.startswith("sequence_classifier.model.layers.") for k in keys):
        layers = 1 + max(int(re.search(r"layers\.(\d+)\.", k).group(1)) for k in keys
                         if k.startswith("sequence_classifier.model.layers."))
        ff = int(model_state["sequence_classifier.model.layers.0.linear1.weight"].shape[0])
        max_len = int(model_state["sequence_classifier.pos_encoding.pe"].shape[1])
        cfg = {"model_type": "transformer", "num_layers": layers, "dim_feedforward": ff, "max_len": max_len,
               "nhead": 8}
    else:
        raise RuntimeError("checkpoint has neither a BiLSTM nor a Transformer temporal head")
    return num_class, cfg


def load_model(checkpoint_path, device="cuda", sequence_model_config: dict | None = None) -> JDCNet:
    """Build a ``JDCNet`` for a reference-format checkpoint ({"model": state_dict, ...} or a bare state dict),
    load it non-strictly and put it in eval mode on ``device``."""
    path = Path(checkpoint_path)
    if not path.is_file():
        raise FileNotFoundError(f"Checkpoint not found: {path}")
    blob = torch.load(path, map_location="cpu", weights_only=True)
    if not isinstance(blob, dict):
        raise RuntimeError("Unexpected checkpoint format")
    state = blob.get("model", blob.get("state_dict", blob))
    if not isinstance(state, dict):
        raise RuntimeError("Checkpoint is missing a valid model state")
    num_class, cfg = infer_model_config(state)
    cfg.update(sequence_model_config or {})
    model = JDCNet(num_class=num_class, sequence_model_config=cfg)
    model.load_state_dict(state, strict=False)
    return model.to(device).eval()


def waveform_to_mel(audio, mel_transform: MelSpectrogram | None = None, device="cuda") -> torch.Tensor:
    """(N,) float audio at the model rate -> (n_mels, L) normalised log-mel on the device."""
    tf = mel_transform or MelSpectrogram(**DEFAULT_MEL_PARAMS)
    wave = torch.as_tensor(np.asarray(audio, dtype=np.float32)).to(device)
    mel = tf(wave)
    return (torch.log(mel + LOG_EPS) - MEL_MEAN) / MEL_STD


@torch.no_grad()
def predict_f0(model: JDCNet, audio, chunk_size: int = 192, overlap: int = 48,
               mel_transform: MelSpectrogram | None = None) -> np.ndarray:
    device = model.flat_parameters.device
    mel = waveform_to_mel(audio, mel_transform, device)
    total = mel.shape[-1]
    step = max(chunk_size - overlap, 1)
    starts = list(range(0, total, step))
    if not starts:
        return np.zeros((0,), dtype=np.float32)
    batch = torch.zeros((len(starts), 1, mel.shape[0], chunk_size), dtype=torch.float32, device=device)
    for i, s in enumerate(starts):
        e = min(s + chunk_size, total)
        batch[i, 0, :, :e - s] = mel[:, s:e]
    was_training = model.training
    model.eval()
    f0, _ = model(batch.transpose(-1, -2))
    if ops.persistent_lstm_error(device):          # a timed-out group barrier voids the outputs: redo on the safe kernels
        ops.clear_persistent_lstm_error(device)
        ops.USE_PERSISTENT_LSTM = False
        f0, _ = model(batch.transpose(-1, -2))
    if was_training:
        model.train()
    f0 = f0[..., 0].cpu().numpy() if f0.shape[-1] == 1 else f0.cpu().numpy()
    return np.concatenate([f0[i][:min(s + chunk_size, total) - s] for i, s in enumerate(starts)])
