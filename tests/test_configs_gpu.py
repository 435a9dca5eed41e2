"""BASELINE.json configs that no other GPU test covers end to end.

config[4] on one GPU: 44.1 kHz 4 s files on disk -> pre-crop at the source rate -> on-GPU resample (147 -> 80
polyphase) -> fused mel -> JDCNet with the Transformer head and ``num_class = 360`` -> CREPE-bin cross-entropy +
BCE -> AdamW, with ``gradient_checkpointing = True`` -- five optimiser steps against the CPU oracle trainer
(oracle/train_ref.CpuTrainer) fed by the float64 resampler + mel oracles on the same crops.
Also: whole-model gradient checkpointing (reference trainer.py:226-233) recomputes the forward and changes
nothing but memory and the BatchNorm running statistics' second momentum update.
"""
import logging
import random

import numpy as np
import pytest
import torch

from oracle import mel_ref, model_ref, resample_ref, train_ref
from pitchextractor_amd import meldataset as md
from pitchextractor_amd import ops
from pitchextractor_amd.model import JDCNet
from pitchextractor_amd.optimizers import build_optimizer
from pitchextractor_amd.trainer import Trainer
from tests.golden.make_golden import SEQ_CFG, TF_CFG, golden_input, golden_targets
from tests.test_data_layer import write_wav

pytestmark = pytest.mark.gpu
CRIT = {"l1": torch.nn.SmoothL1Loss(), "ce": torch.nn.BCEWithLogitsLoss()}


def _trainer(net, **kw):
    opt, sched = build_optimizer({"params": net.parameters(), "optimizer_params": {},
                                  "scheduler_params": {"max_lr": 3e-4, "pct_start": 0.0, "epochs": 100,
                                                       "steps_per_epoch": 8}})
    return Trainer(model=net, criterion=CRIT, optimizer=opt, scheduler=sched, device="cuda:0",
                   loss_config={"lambda_f0": 0.1}, logger=logging.getLogger("t"), **kw)


def _glide_44k(i, dur=4.0, sr=44100, hop_24k=300):
    """4 s sine glide at 44.1 kHz + its analytic F0 at the 24 kHz / hop-300 frame rate with an unvoiced gap."""
    rng = np.random.default_rng(900 + i)
    n = int(dur * sr)
    f = np.linspace(rng.uniform(80, 200), rng.uniform(200, 450), n)
    wave = (0.6 * np.sin(np.cumsum(2 * np.pi * f / sr))).astype(np.float32)
    n_frames = 1 + int(np.ceil(n * 80 / 147)) // hop_24k
    f0 = np.interp(np.linspace(0, n - 1, n_frames), np.arange(n), f).astype(np.float32)
    gap = int(rng.integers(20, 200))
    f0[gap:gap + 25] = 0.0
    wave[int(gap * hop_24k * 147 / 80):int((gap + 25) * hop_24k * 147 / 80)] = 0.0
    return wave, f0


def test_config4_resample_mel_transformer_bins_checkpointing_vs_cpu_oracle(tmp_path, hip_device):
    lines = []
    for i in range(4):
        wave, f0 = _glide_44k(i)
        p = tmp_path / f"s{i}.wav"
        write_wav(p, wave, 44100, "float32")
        np.save(str(p) + "_f0.npy", f0)
        lines.append(f"{p}|0\n")
    cfg = {"mel_params": {"sample_rate": 24000, "win_len": 1024, "n_fft": 1024, "n_mels": 80, "hop_length": 300},
           "verbose": False}
    loader = md.build_dataloader(lines, validation=True, batch_size=4, num_workers=0, device="cuda:0",
                                 dataset_config=cfg)
    state = model_ref.seeded_state(41, num_class=360, model_type="transformer")
    net = JDCNet(num_class=360, sequence_model_config=dict(TF_CFG))
    net.load_state_dict(state, strict=True)
    net = net.to(hip_device).train()
    net.block_dropout = 0.0
    tr = _trainer(net, gradient_checkpointing=True)
    torch.set_num_threads(16)
    cpu = train_ref.CpuTrainer(state, dict(TF_CFG), max_lr=3e-4, total_steps=800, lambda_f0=0.1)
    ds = loader.dataset
    for step in range(5):
        np.random.seed(100 + step); random.seed(100 + step)
        (batch,) = list(loader)                         # one batch per pass: pre-crop + crop draws from the seeds
        got = tr.run(batch)
        # the same draws replayed on the host, every item through the float64 oracles
        np.random.seed(100 + step); random.seed(100 + step)
        items = []
        for path in ds.data_list:
            wave, f0, sil, crop = ds.path_to_wave_and_label(path)
            assert ds._last_sr == 44100 and wave.shape == (107722,)           # pre-crop at 44.1 kHz
            res = resample_ref.resample(wave, 44100, 24000)
            mel = mel_ref.log_mel(res)[:, crop:crop + 192].astype(np.float32)
            assert mel.shape[1] == 192                                        # 4 s -> 196 frames -> random 192 crop
            items.append((mel, f0, sil))
        rm, rf, rs = train_ref.collate(items)
        assert np.abs(batch[0].cpu().numpy() - rm).max() <= 2e-3              # resample + mel vs float64 oracles
        np.testing.assert_array_equal(batch[1].cpu().numpy(), rf)
        ref = cpu.run((torch.from_numpy(rm), torch.from_numpy(rf), torch.from_numpy(rs)))
        # the inputs already differ by up to 2e-3 (fp32 resampler + fp32 FFT vs the float64 oracles) and the CPU
        # trainer is itself fp32: total loss to 2e-3, its two terms to 5e-3 over the five updates
        for key, tol in (("loss", 2e-3), ("f0", 5e-3), ("sil", 5e-3)):
            assert abs(got[key] - ref[key]) <= tol * abs(ref[key]) + 1e-5, (step, key, got, ref)
    assert ref["f0"] > 0.1                              # the 360-bin CE term is live


@pytest.mark.parametrize("head", ["bilstm", "transformer"])
def test_gradient_checkpointing_recomputes_and_matches_plain_step(hip_device, head):
    """trainer.py:226-233: the forward keeps no activations, backward re-runs it.  Same kernels, same dropout
    stream => loss and every gradient are bit-identical to the plain step; nothing is held between the two passes; BatchNorm
    running statistics take two momentum updates with the same batch statistics (what torch.utils.checkpoint
    does to the reference's BatchNorm layers)."""
    cfg = dict(SEQ_CFG if head == "bilstm" else TF_CFG, dropout=0.1)
    if head == "bilstm":
        cfg["hidden_size"] = 128
    state = model_ref.seeded_state(13, model_type=head, hidden_size=cfg.get("hidden_size", 384))
    x = golden_input(4, B=8)
    f0, sil = golden_targets(4, B=8)
    batch = (x.transpose(-1, -2).contiguous(), f0, sil)         # Trainer batches are (B,1,80,192)
    res = {}
    for ckpt in (False, True):
        net = JDCNet(num_class=1, sequence_model_config=dict(cfg))
        net.load_state_dict(state, strict=True)
        net = net.to(hip_device).train()                         # all dropouts live (block 0.5, head 0.1)
        net.dropout_cfg.seed = 99
        tr = _trainer(net, gradient_checkpointing=ckpt)
        out = tr.run(batch)
        stats = {k: v.clone() for k, v in net.state_dict().items() if "running" in k or "tracked" in k}
        grads, params, drop_pos = net.flat_gradients().clone(), net.flat_parameters.detach().clone(), net.dropout_cfg.offset
        # memory held between forward and backward (a single whole-model segment saves nothing at the backward
        # peak, where the recomputed activations are all live again -- in the reference as here)
        torch.cuda.synchronize()
        base = torch.cuda.memory_allocated()
        net.checkpoint_forward = ckpt
        outs = net(x.to(hip_device))
        torch.cuda.synchronize()
        held = torch.cuda.memory_allocated() - base
        del outs
        res[ckpt] = (out, grads, params, stats, held, drop_pos)
    (o0, g0, p0, s0, m0, d0), (o1, g1, p1, s1, m1, d1) = res[False], res[True]
    assert o0 == o1 and torch.equal(g0, g1) and torch.equal(p0, p1) and d0 == d1
    assert m1 < 0.1 * m0, (m0, m1)                               # no activations kept across the forward
    for k in s0:
        if k.endswith("num_batches_tracked"):
            assert int(s1[k]) == 2 and int(s0[k]) == 1
        else:
            init = state[k].to(hip_device)
            twice = init + (1 - 0.9 ** 2) / 0.1 * (s0[k] - init)           # r <- 0.9 r + 0.1 s, applied twice
            assert torch.allclose(s1[k], twice, rtol=1e-5, atol=1e-7), k


def test_fp16_gradscaler_mode(hip_device):
    """training.precision: fp16 = the reference's literal AMP (trainer.py:64-102,241-244): fp16 operands + GradScaler.
    12 optimiser steps track the fp32 CPU oracle trainer within 2 %; a step whose gradients overflow is skipped
    (parameters and AdamW moments untouched, scheduler still advances) and halves the scale; 3 clean steps at
    growth_interval = 3 double it."""
    from tests.golden.make_golden import training_batches
    state = model_ref.seeded_state(21)
    net = JDCNet(num_class=1, sequence_model_config=dict(SEQ_CFG))
    net.load_state_dict(state, strict=True)
    net = net.to(hip_device).train()
    net.block_dropout = 0.0
    tr = _trainer(net, use_mixed_precision=True, amp_dtype="fp16")
    assert tr.scaler is not None and tr.scaler.scale == 65536.0
    torch.set_num_threads(16)
    cpu = train_ref.CpuTrainer(state, dict(SEQ_CFG), max_lr=3e-4, total_steps=800, lambda_f0=0.1)
    batches = list(training_batches(12))
    for i, batch in enumerate(batches):
        got, ref = tr.run(batch), cpu.run(batch)
        assert abs(got["loss"] - ref["loss"]) <= 2e-2 * abs(ref["loss"]), (i, got, ref)
    assert tr.scaler.skipped_steps == 0 and tr.scaler.scale == 65536.0
    # overflow: a scale that pushes the loss gradients past fp32 range -> inf gradients -> skipped step
    before = net.flat_parameters.detach().clone()
    lr_before = tr._get_lr()
    tr.scaler.scale = 3.0e38
    tr.run(batches[0])
    assert tr.scaler.skipped_steps == 1 and tr.scaler.scale == 1.5e38
    assert torch.equal(net.flat_parameters.detach(), before)            # update skipped ...
    assert tr._get_lr() != lr_before                                    # ... the scheduler stepped (trainer.py:248)
    tr.scaler.scale, tr.scaler.growth_interval, tr.scaler._good_steps = 1024.0, 3, 0
    for batch in batches[:3]:
        tr.run(batch)
    assert tr.scaler.scale == 2048.0 and not torch.equal(net.flat_parameters.detach(), before)
    # bf16 mode has no scaler; unknown dtypes are refused
    assert _trainer(net, use_mixed_precision=True).scaler is None
    with pytest.raises(ValueError):
        _trainer(net, use_mixed_precision=True, amp_dtype="fp8")


def test_gradient_checkpointing_at_bench_size(hip_device):
    """The same bit-identity at B = 256 with the default BiLSTM and every dropout live (config[4] names the flag):
    the recomputed forward replays the Philox stream and the 192-workgroup persistent recurrences exactly."""
    cfg = dict(SEQ_CFG, dropout=0.1)
    state = model_ref.seeded_state(13)
    x = golden_input(4, B=8).repeat(32, 1, 1, 1)
    f0, sil = (t.repeat(32, 1) for t in golden_targets(4, B=8))
    batch = (x.transpose(-1, -2).contiguous(), f0, sil)
    res = []
    for ckpt in (False, True):
        net = JDCNet(num_class=1, sequence_model_config=dict(cfg))
        net.load_state_dict(state, strict=True)
        net = net.to(hip_device).train()
        net.dropout_cfg.seed = 7
        out = _trainer(net, gradient_checkpointing=ckpt).run(batch)
        res.append((out, net.flat_gradients().clone(), net.flat_parameters.detach().clone()))
    assert not ops.persistent_lstm_error(hip_device)
    (o0, g0, p0), (o1, g1, p1) = res
    assert o0 == o1 and torch.equal(g0, g1) and torch.equal(p0, p1)
    assert np.isfinite(o0["loss"]) and g0.abs().max().item() > 0
