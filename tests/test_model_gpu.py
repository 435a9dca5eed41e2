"""JDCNet on the HIP path vs the functional CPU oracle / the reference's golden vectors.

Tolerances: the oracle side is float64 (exact to ~1e-9 against the reference, see
tests/test_oracle_golden.py), the HIP side is fp32.  Outputs are held to 1e-4 of their scale
(north_star), per-parameter gradient norms to 2e-3 and the 100-step loss curve to 1e-3 relative.
The golden comparisons run in both fp32 product modes -- "x3" (default: exact three-term bf16 split on
the bf16 MFMA pipe) and "native" (v_mfma_f32_32x32x2_f32) -- at the same tolerances.
"""
import logging

import numpy as np
import pytest
import torch

from oracle import model_ref
from pitchextractor_amd import ops
from pitchextractor_amd.model import JDCNet
from pitchextractor_amd.optimizers import build_optimizer
from pitchextractor_amd.trainer import Trainer
from tests.golden.make_golden import SEQ_CFG, golden_input, golden_targets, tap_summary, training_batches

pytestmark = pytest.mark.gpu


def build(state, num_class, hidden, device, dropout=0.0):
    cfg = dict(SEQ_CFG, hidden_size=hidden, dropout=dropout)
    net = JDCNet(num_class=num_class, sequence_model_config=cfg)
    net.load_state_dict(state, strict=True)
    return net.to(device)


def close(got, ref, tol):
    got = np.asarray(got.detach().cpu().double() if torch.is_tensor(got) else got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape
    scale = np.abs(ref).max() + 1e-30
    assert np.abs(got - ref).max() <= tol * scale, (np.abs(got - ref).max(), scale)


@pytest.fixture(scope="module")
def G(golden_dir):
    return np.load(golden_dir / "model_golden.npz")


def test_state_dict_layout_matches_reference(G, hip_device):
    net = build(model_ref.seeded_state(11), 1, 384, hip_device)
    sd = net.state_dict()
    assert len(sd) == 125 and sum(p.numel() for p in net.parameters()) == 28882755
    assert [str(n) for n in G["nc1_grad_names"]] == [n for n, _ in net.named_parameters()]
    assert sd["conv_block.1.num_batches_tracked"].dtype == torch.long
    assert sd["detector_conv.0.weight"].shape == (256, 640, 1, 1)
    assert all(p.data_ptr() >= net.flat_parameters.data_ptr() for p in net.parameters())


@pytest.mark.parametrize("fp32_mode", ["h2", "x3", "native"])
@pytest.mark.parametrize("tag,nc,hidden", [("nc1", 1, 384), ("nc360", 360, 64)])
def test_eval_forward_matches_reference_golden(G, hip_device, tag, nc, hidden, fp32_mode, monkeypatch):
    monkeypatch.setattr(ops, "FP32_MATMUL", fp32_mode)
    net = build(model_ref.seeded_state(11, num_class=nc, hidden_size=hidden), nc, hidden, hip_device).eval()
    with torch.no_grad():
        cls, det = net(golden_input(3).to(hip_device))
    assert cls.shape == (2, 192, nc) and det.shape == (2, 192)
    close(cls, G[f"{tag}_eval_cls"], 1e-4)
    close(det, G[f"{tag}_eval_det"], 1e-4)


@pytest.fixture(scope="module", params=["h2", "x3", "native"])
def train_pass(hip_device, request):
    prev, ops.FP32_MATMUL = ops.FP32_MATMUL, request.param
    try:
        net = build(model_ref.seeded_state(11), 1, 384, hip_device).train()
        net.block_dropout = 0.0
        net.keep_last_context = True
        x = golden_input(3).to(hip_device)
        f0, sil = (t.to(hip_device) for t in golden_targets(3))
        cls, det = net(x)
        out3, d_f0, d_sil = ops.f0_sil_loss(cls.detach().reshape(-1), f0.reshape(-1), det.detach().reshape(-1),
                                            sil.reshape(-1), 0.1)
        torch.autograd.backward([cls, det], [d_f0.view_as(cls), d_sil.view_as(det)])
    finally:
        ops.FP32_MATMUL = prev
    return net, cls, det, out3


def test_train_forward_loss_and_taps(G, train_pass):
    net, cls, det, out3 = train_pass
    close(cls, G["nc1_f64_train_cls"], 1e-4)
    close(det, G["nc1_f64_train_det"], 1e-4)
    np.testing.assert_allclose(out3.cpu().numpy(), G["nc1_f64_loss"], rtol=1e-5)
    s = net.last_context
    nchw = lambda t: t.permute(0, 3, 1, 2)
    for name, ours in (("conv_block", nchw(s.cb)), ("res_block1", nchw(s.rb1)), ("res_block2", nchw(s.rb2)),
                       ("res_block3", nchw(s.rb3)), ("sequence_classifier", s.yc), ("sequence_detector", s.yd)):
        ref = G[f"nc1_f64_tap_{name}"]
        got = tap_summary(ours.cpu())
        assert np.abs(got - ref).max() <= 1e-4 * np.abs(ref).max(), name


def test_gradients_match_reference_float64(G, train_pass):
    net = train_pass[0]
    names = [str(n) for n in G["nc1_f64_grad_names"]]
    norms = dict(zip(names, G["nc1_f64_grad_norms"]))
    bad = []
    for n, p in net.named_parameters():
        assert p.grad is not None and p.grad.data_ptr() >= net.flat_gradients().data_ptr()
        got = p.grad.double().norm().item()
        if abs(got - norms[n]) > 2e-3 * norms[n] + 1e-9:
            bad.append((n, got, norms[n]))
    assert not bad, bad
    for key in G.files:
        if key.startswith("nc1_f64_grad_") and not key.endswith(("_grad_names", "_grad_norms")):
            n = key[len("nc1_f64_grad_"):]
            g = dict(net.named_parameters())[n].grad.flatten().cpu()
            idx = torch.linspace(0, g.numel() - 1, min(32, g.numel())).long()
            ref = G[key]
            assert np.abs(g[idx].numpy() - ref).max() <= 5e-3 * np.abs(ref).max() + 1e-9, n


def test_running_stats_updated_like_reference(G, train_pass):
    sd = train_pass[0].state_dict()
    for key in G.files:
        if key.startswith("nc1_f64_stat_"):
            n = key[len("nc1_f64_stat_"):]
            close(sd[n], G[key], 1e-5)
    assert int(sd["conv_block.1.num_batches_tracked"]) == 1


def test_dropout_masks_replayed_in_oracle(hip_device):
    """Train mode with every dropout live: export the HIP masks, replay them in the float64 oracle."""
    state = model_ref.seeded_state(11, hidden_size=64)
    net = build(state, 1, 64, hip_device, dropout=0.1).train()
    net.keep_last_context = True
    net.dropout_cfg.seed = 77
    x = golden_input(5)
    cls, det = net(x.to(hip_device))
    s = net.last_context
    masks = [s.mask_pool.cpu().view(2, 192, 2, 256), s.mask_det.cpu().view(2, 192, 2, 256)]
    for lay in s.lstm[:-1]:
        masks += [m.cpu().view(2, 192, -1) for (_, m) in lay.mask]
    assert abs(masks[0].float().mean().item() - 0.5) < 0.02 and abs(masks[2].float().mean().item() - 0.9) < 0.02
    st64 = {k: (v.double() if v.dtype.is_floating_point else v) for k, v in state.items()}
    cfg = dict(SEQ_CFG, hidden_size=64, dropout=0.1)
    ref_cls, ref_det = model_ref.jdcnet_forward(st64, x.double(), cfg, train=True, masks=iter(masks))
    close(cls, ref_cls.detach().numpy(), 1e-4)
    close(det, ref_det.detach().numpy(), 1e-4)


def test_hundred_step_loss_curve_matches_reference_trainer(golden_dir, hip_device):
    """BASELINE config 1 (B=4, default JDCNet+BiLSTM, dropout off): reference Trainer.run x100 on CPU
    (tests/golden/step_golden.npz) vs the HIP trainer on identical batches."""
    curve = np.load(golden_dir / "step_golden.npz")["curve"]
    net = build(model_ref.seeded_state(21), 1, 384, hip_device).train()
    net.block_dropout = 0.0
    opt, sched = build_optimizer({"params": net.parameters(), "optimizer_params": {},
                                  "scheduler_params": {"max_lr": 3e-4, "pct_start": 0.0, "epochs": 100,
                                                       "steps_per_epoch": 8}})
    crit = {"l1": torch.nn.SmoothL1Loss(), "ce": torch.nn.BCEWithLogitsLoss()}
    tr = Trainer(model=net, criterion=crit, optimizer=opt, scheduler=sched, device="cuda:0",
                 loss_config={"lambda_f0": 0.1}, logger=logging.getLogger("t"))
    got = []
    for batch in training_batches(len(curve)):
        r = tr.run(batch)
        got.append([r["loss"], r["f0"], r["sil"]])
    got = np.array(got)
    rel = np.abs(got[:, 0] - curve[:, 0]) / curve[:, 0]
    assert rel.max() <= 1e-3, (rel.max(), int(rel.argmax()))
    assert abs(got[-1, 0] - curve[-1, 0]) <= 1e-3 * curve[-1, 0]
    # the silence BCE term decays to ~1e-3 within 60 steps, where single frames flipping side dominate it:
    # held to 1e-3 of its initial scale while it is large, to 2e-3 absolute once it is that small
    assert np.abs(got[:20, 2] - curve[:20, 2]).max() <= 1e-3 * curve[:, 2].max()
    assert np.abs(got[:, 2] - curve[:, 2]).max() <= 2e-3


def test_classification_head_training_steps_match_oracle(hip_device):
    """N4 (build-defined, no reference semantics): num_class = 360 with the CREPE-bin cross-entropy.  Five
    optimiser steps of the HIP trainer vs the fp32 CPU oracle trainer on identical batches (dropout off)."""
    from oracle import train_ref
    cfg = dict(SEQ_CFG, hidden_size=64, dropout=0.0)
    state = model_ref.seeded_state(31, num_class=360, hidden_size=64)
    net = build(state, 360, 64, hip_device).train()
    net.block_dropout = 0.0
    opt, sched = build_optimizer({"params": net.parameters(), "optimizer_params": {},
                                  "scheduler_params": {"max_lr": 3e-4, "pct_start": 0.0, "epochs": 100,
                                                       "steps_per_epoch": 8}})
    tr = Trainer(model=net, criterion={"l1": torch.nn.SmoothL1Loss(), "ce": torch.nn.BCEWithLogitsLoss()},
                 optimizer=opt, scheduler=sched, device="cuda:0", loss_config={"lambda_f0": 0.1},
                 logger=logging.getLogger("t"))
    cpu = train_ref.CpuTrainer(state, cfg, max_lr=3e-4, total_steps=800, lambda_f0=0.1)
    for batch in training_batches(5):
        got, ref = tr.run(batch), cpu.run(batch)
        for key in ("loss", "f0", "sil"):
            assert abs(got[key] - ref[key]) <= 1e-3 * abs(ref[key]) + 1e-5, (key, got, ref)
    assert ref["f0"] > 0.1                       # the CE term is live (ln 360 * 0.1 ~ 0.59 at init)


def test_mixed_precision_training_tracks_fp32(hip_device):
    """training.mixed_precision (bf16 operands for conv / linear products and their weight gradients, fp32
    accumulate and state): 12 optimiser steps stay within 2 % of the fp32 loss curve on identical batches."""
    curves = {}
    for amp in (False, True):
        net = build(model_ref.seeded_state(21), 1, 384, hip_device).train()
        net.block_dropout = 0.0
        opt, sched = build_optimizer({"params": net.parameters(), "optimizer_params": {},
                                      "scheduler_params": {"max_lr": 3e-4, "pct_start": 0.0, "epochs": 100,
                                                           "steps_per_epoch": 8}})
        tr = Trainer(model=net, criterion={"l1": torch.nn.SmoothL1Loss(), "ce": torch.nn.BCEWithLogitsLoss()},
                     optimizer=opt, scheduler=sched, device="cuda:0", loss_config={"lambda_f0": 0.1},
                     logger=logging.getLogger("t"), use_mixed_precision=amp)
        curves[amp] = np.array([tr.run(b)["loss"] for b in training_batches(12)])
        assert ops.MATMUL_BF16 is False                         # the scope is per step
    rel = np.abs(curves[True] - curves[False]) / curves[False]
    assert rel.max() <= 2e-2, rel
    assert (curves[True] != curves[False]).any()                # the mode really changes the arithmetic


def test_full_batch_is_sample_independent_and_deterministic(hip_device):
    """BASELINE size B=256 in eval mode: every sample's logits equal the same sample run in a batch of 8,
    bit for bit (tile position must not change a row's summation order), and reruns are identical."""
    net = build(model_ref.seeded_state(11), 1, 384, hip_device).eval()
    x8 = golden_input(9, B=8).to(hip_device)
    big = x8.repeat(32, 1, 1, 1)
    with torch.no_grad():
        c8, d8 = net(x8)
        cb, db = net(big)
        cb2, db2 = net(big)
    assert cb.shape == (256, 192, 1)
    assert torch.equal(cb, cb2) and torch.equal(db, db2)
    assert torch.equal(cb[:8], c8) and torch.equal(cb[248:], c8) and torch.equal(db[96:104], d8)


def _f64_oracle_grads(state, cfg, x, f0, sil):
    """float64 CPU oracle: forward + backward of the reference-equivalent step (train mode, dropout off)."""
    st64 = {k: (v.double().requires_grad_(not k.endswith(("running_mean", "running_var")))
                if v.dtype.is_floating_point else v) for k, v in state.items()}
    cls, det = model_ref.jdcnet_forward(st64, x.double(), cfg, train=True)
    loss, _, _ = model_ref.jdc_loss(cls, det, f0.double(), sil.double(), 0.1)
    loss.backward()
    return cls.detach(), det.detach(), loss.item(), {k: v.grad for k, v in st64.items()
                                                      if torch.is_tensor(v) and v.dtype.is_floating_point
                                                      and v.grad is not None}


@pytest.fixture(scope="module")
def oracle_b8():
    torch.set_num_threads(16)
    state = model_ref.seeded_state(11)
    x8 = golden_input(9, B=8)
    f0, sil = golden_targets(9, B=8)
    return state, x8, f0, sil, _f64_oracle_grads(state, dict(SEQ_CFG), x8, f0, sil)


def _hip_step_grads(state, x, f0, sil, device):
    net = build(state, 1, 384, device).train()
    net.block_dropout = 0.0
    cls, det = net(x.to(device))
    out3, d_f0, d_sil = ops.f0_sil_loss(cls.detach().reshape(-1), f0.to(device).reshape(-1), det.detach().reshape(-1),
                                        sil.to(device).reshape(-1), 0.1)
    torch.autograd.backward([cls, det], [d_f0.view_as(cls), d_sil.view_as(det)])
    return cls, det, out3[0].item(), {n: p.grad.detach().clone() for n, p in net.named_parameters()}


@pytest.mark.parametrize("fp32_mode", ["h2", "x3", "native"])
def test_full_size_training_step_matches_oracle_by_tiling(hip_device, oracle_b8, fp32_mode, monkeypatch):
    """BASELINE config[1] size (B = 256, train mode, default BiLSTM) against the float64 oracle.

    Eight samples tiled x32: BatchNorm batch statistics, the mean loss and every parameter gradient of
    the tiled batch equal those of the eight-sample batch exactly in real arithmetic, so the float64
    oracle at B = 8 is the oracle for B = 256 too.  This is where the split-K planner, the XCD remap, the
    9-tap weight-gradient slab reduction over 3.9 M pixels and the full persistent-LSTM grid run at their
    bench shapes.  Tolerances: logits 1e-4 of scale, loss 1e-5, per-parameter gradient norms 2e-3 (the golden
    test's), every gradient ELEMENT within 1.5e-2 of its tensor's largest element against float64 -- the
    reference's own fp32 CPU path is off by up to 5.8e-3 there and this path by 1.07e-2 (h2) / 1.08e-2 (native) /
    1.37e-2 (x3).  That deviation is NOT the weight-gradient kernels' accumulation: capping a slab at 2 048 pixels
    instead of 7 680, summing the slabs in double, running the recurrences on the per-time-step kernels or on the
    native fp32 MFMA all leave it unchanged to three digits (tools/fs_tune.py) -- it is fp32 rounding anywhere upstream
    (forward activations, the 4 x 192-step recurrences) seen through train-mode BatchNorm's cancelling backward --
    and, the actual point of the test, within 1e-3 of the SAME path run at B = 8 (measured: identical to 3 digits)."""
    monkeypatch.setattr(ops, "FP32_MATMUL", fp32_mode)
    state, x8, f0, sil, (ref_cls, ref_det, ref_loss, ref_g) = oracle_b8
    reps = 32
    _, _, loss8, g8 = _hip_step_grads(state, x8, f0, sil, hip_device)
    cls, det, loss, g = _hip_step_grads(state, x8.repeat(reps, 1, 1, 1), f0.repeat(reps, 1), sil.repeat(reps, 1),
                                        hip_device)
    assert not ops.persistent_lstm_error(hip_device)
    assert cls.shape == (256, 192, 1)
    for r in (0, 13, 31):                                           # every replica of the tile, same numbers
        close(cls[8 * r:8 * r + 8], ref_cls.numpy(), 1e-4)
        close(det[8 * r:8 * r + 8], ref_det.numpy(), 1e-4)
    assert abs(loss - ref_loss) <= 1e-5 * abs(ref_loss) and abs(loss - loss8) <= 1e-6 * abs(ref_loss)
    bad, worst = [], (0.0, "")
    for n, got in g.items():
        ref = ref_g[n]
        got64, top = got.cpu().double(), ref.abs().max().item()
        rn = ref.norm().item()
        worst = max(worst, ((got64 - ref).abs().max().item() / (top + 1e-30), n))
        if abs(got64.norm().item() - rn) > 2e-3 * rn + 1e-9:
            bad.append((n, "norm", got64.norm().item(), rn))
        if (got64 - ref).abs().max().item() > 1.5e-2 * top + 1e-9:
            bad.append((n, "elem vs float64", (got64 - ref).abs().max().item(), top))
        if (got - g8[n]).abs().max().item() > 1e-3 * top + 1e-9:
            bad.append((n, "elem vs B=8", (got - g8[n]).abs().max().item(), top))
    print(f"{fp32_mode}: worst gradient element error vs float64 / tensor max = {worst[0]:.2e} ({worst[1]})")
    assert not bad, bad


@pytest.mark.parametrize("act16", [False, True])
def test_full_size_mixed_precision_step_by_tiling(hip_device, oracle_b8, act16, dtype="bf16"):
    """BASELINE config[3]'s per-GPU shape (B = 256, default BiLSTM, 16-bit operands: the shipped config.yml's
    `mixed_precision: true`, `precision: bf16`): the same tiling argument with the mixed-precision kernels -- bf16 conv,
    GEMM and weight-gradient products and the 16-bit persistent recurrences on the full 192-workgroup grid.  Both runs
    round their operands alike, so forward results agree to summation order: logits within 1e-4, loss 1e-6.  The
    gradients agree to bf16 resolution only (2e-2 of each tensor's largest element; measured 5.5e-3, worst in front of
    the first BatchNorms): the mixed-precision backward recurrence hands its partial dh tiles over in bf16, and a
    1e-7 difference in an input (BatchNorm statistics summed over 256 instead of 8 samples) that lands on the other
    side of a bf16 rounding boundary moves that tile element by 2^-8.  The recurrence kernels themselves are exactly
    batch-size-, replica- and scale-invariant (tests/test_ops_gpu.py::test_mixed_precision_recurrences_are_exactly_
    batch_and_scale_invariant).
    (fp16 operands are not comparable this way without the GradScaler: the per-element
    loss gradient is 32x smaller at B = 256 and underflows differently; that mode has its own test.)"""
    state, x8, f0, sil, (_, _, ref_loss, _) = oracle_b8
    tol_out, tol_g = 1e-4, 2e-2
    with ops.matmul_bf16(True, dtype, act16=act16):
        cls8, det8, loss8, g8 = _hip_step_grads(state, x8, f0, sil, hip_device)
        cls, det, loss, g = _hip_step_grads(state, x8.repeat(32, 1, 1, 1), f0.repeat(32, 1), sil.repeat(32, 1),
                                            hip_device)
    assert not ops.persistent_lstm_error(hip_device)
    assert abs(loss8 - ref_loss) <= 2e-2 * abs(ref_loss)            # 16-bit operands vs the float64 oracle
    assert abs(loss - loss8) <= 1e-6 * abs(loss8)
    for r in (0, 13, 31):
        assert (cls[8 * r:8 * r + 8] - cls8).abs().max().item() <= tol_out * cls8.abs().max().item()
        assert (det[8 * r:8 * r + 8] - det8).abs().max().item() <= tol_out * det8.abs().max().item()
    ratios = sorted(((g[n] - g8[n]).abs().max().item() / (g8[n].abs().max().item() + 1e-30), n,
                     g8[n].abs().max().item()) for n in g)
    print("worst five:", [(n, f"{r:.2e}", f"top {t:.2e}") for r, n, t in ratios[-5:]])
    worst = ratios[-1][0]
    print(f"{dtype} (bf16 activation storage: {act16}): loss {loss:.6f} (B=8 {loss8:.6f}, float64 fp32-model oracle "
          f"{ref_loss:.6f}); worst gradient element error / tensor max = {worst:.2e}")
    assert worst <= tol_g


def test_side_stream_weight_gradients_change_nothing(hip_device, oracle_b8, monkeypatch):
    """model._SideWork / the LSTM side stream only move WHERE the weight-gradient kernels run: at B = 256 (full
    persistent-LSTM grid, every overlap live) the loss and every gradient are bit-identical to the serialised
    backward, twice in a row (a cross-stream lifetime or ordering bug shows up as a differing element)."""
    from pitchextractor_amd import model as pe_model
    state, x8, f0, sil, _ = oracle_b8
    x, f0, sil = x8.repeat(32, 1, 1, 1), f0.repeat(32, 1), sil.repeat(32, 1)
    runs = {}
    for on in (False, True, True):
        for flag in ("OVERLAP_LSTM_WGRAD", "OVERLAP_CONV_WGRAD", "OVERLAP_TF_WGRAD"):
            monkeypatch.setattr(pe_model, flag, on)
        _, _, loss, g = _hip_step_grads(state, x, f0, sil, hip_device)
        runs.setdefault(on, []).append((loss, g))
    assert not ops.persistent_lstm_error(hip_device)
    (loss0, g0), = runs[False]
    for loss1, g1 in runs[True]:
        assert loss1 == loss0
        for name in g0:
            assert torch.equal(g0[name], g1[name]), name


def test_cpu_input_fails_loudly():
    net = JDCNet(num_class=1, sequence_model_config=dict(SEQ_CFG))
    with pytest.raises(RuntimeError):
        net(torch.zeros(1, 1, 192, 80))


@pytest.mark.parametrize("B,T", [(1, 192), (3, 48), (2, 208)])
def test_odd_batch_and_sequence_lengths(hip_device, B, T):
    """Inference chunks need not be 192 frames (notebook predict_f0 pads the tail) and B may be 1
    (the reference's ``.squeeze()`` then also drops the batch axis, which the flattened loss ignores)."""
    state = model_ref.seeded_state(11, hidden_size=64, num_layers=2)
    cfg = dict(SEQ_CFG, hidden_size=64, num_layers=2)
    net = JDCNet(num_class=1, sequence_model_config=dict(cfg))
    net.load_state_dict(state)
    net = net.to(hip_device).eval()
    x = golden_input(17, B=B, T=T)
    with torch.no_grad():
        cls, det = net(x.to(hip_device))
        ref_cls, ref_det = model_ref.jdcnet_forward(
            {k: (v.double() if v.dtype.is_floating_point else v) for k, v in state.items()}, x.double(), cfg)
    assert cls.shape == (B, T, 1) and det.shape == (B, T)
    close(cls, ref_cls.numpy(), 1e-4)
    close(det, ref_det.numpy(), 1e-4)


def test_checkpoint_roundtrip_and_partial_load(tmp_path, hip_device):
    """save_checkpoint / load_checkpoint keep the reference's dict layout (trainer.py:138-195) and the
    shape-tolerant copy (a num_class=360 checkpoint loads into a num_class=1 model by overlap)."""
    net = build(model_ref.seeded_state(11, hidden_size=64), 1, 64, hip_device)
    opt, sched = build_optimizer({"params": net.parameters(), "optimizer_params": {},
                                  "scheduler_params": {"max_lr": 3e-4, "pct_start": 0.0, "epochs": 2,
                                                       "steps_per_epoch": 4}})
    crit = {"l1": torch.nn.SmoothL1Loss(), "ce": torch.nn.BCEWithLogitsLoss()}
    tr = Trainer(model=net, criterion=crit, optimizer=opt, scheduler=sched, device="cuda:0",
                 loss_config={"lambda_f0": 0.1}, logger=logging.getLogger("t"))
    net.train()
    net.block_dropout = 0.0
    batch = next(iter(training_batches(1)))
    tr.run(batch)
    path = tmp_path / "ckpt" / "epoch_00001.pth"
    tr.save_checkpoint(str(path))
    blob = torch.load(path, map_location="cpu", weights_only=True)
    assert sorted(blob) == ["epochs", "model", "optimizer", "scheduler", "steps"]
    assert len(blob["model"]) == 125 and len(blob["optimizer"]["state"]) == 98
    st0 = blob["optimizer"]["state"][0]
    assert set(st0) >= {"step", "exp_avg", "exp_avg_sq"} and float(st0["step"]) == 1.0
    # resume into a fresh trainer: identical next step
    net2 = build(model_ref.seeded_state(12, hidden_size=64), 1, 64, hip_device).train()
    net2.block_dropout = 0.0
    opt2, sched2 = build_optimizer({"params": net2.parameters(), "optimizer_params": {},
                                    "scheduler_params": {"max_lr": 3e-4, "pct_start": 0.0, "epochs": 2,
                                                         "steps_per_epoch": 4}})
    tr2 = Trainer(model=net2, criterion=crit, optimizer=opt2, scheduler=sched2, device="cuda:0",
                  loss_config={"lambda_f0": 0.1}, logger=logging.getLogger("t"))
    tr2.load_checkpoint(str(path), load_only_params=False)
    a, b = tr.run(batch), tr2.run(batch)
    assert abs(a["loss"] - b["loss"]) <= 1e-6 * abs(a["loss"])
    assert torch.equal(net.flat_parameters, net2.flat_parameters)
    # shape-tolerant load (classifier 360 x D -> 1 x D keeps row 0)
    big = model_ref.seeded_state(5, num_class=360, hidden_size=64)
    torch.save({"model": big, "optimizer": {}, "scheduler": {}, "steps": 0, "epochs": 0}, tmp_path / "big.pth")
    tr2.load_checkpoint(str(tmp_path / "big.pth"), load_only_params=True)
    assert torch.equal(net2.classifier.weight.cpu(), big["classifier.weight"][:1])
    assert torch.equal(net2.conv_block[0].weight.cpu(), big["conv_block.0.weight"])


def test_optimizer_state_loaded_after_a_step_is_adopted(tmp_path, hip_device):
    """Loading a checkpoint into a trainer that has ALREADY stepped must continue from the loaded AdamW moments
    (the fused optimizer caches flat moment buffers; ``load_state_dict`` has to drop them)."""
    crit = {"l1": torch.nn.SmoothL1Loss(), "ce": torch.nn.BCEWithLogitsLoss()}
    sp = {"max_lr": 3e-4, "pct_start": 0.0, "epochs": 2, "steps_per_epoch": 8}
    batches = list(training_batches(3))

    def fresh(seed):
        net = build(model_ref.seeded_state(seed, hidden_size=64), 1, 64, hip_device).train()
        net.block_dropout = 0.0
        opt, sched = build_optimizer({"params": net.parameters(), "optimizer_params": {}, "scheduler_params": dict(sp)})
        return net, Trainer(model=net, criterion=crit, optimizer=opt, scheduler=sched, device="cuda:0",
                            loss_config={"lambda_f0": 0.1}, logger=logging.getLogger("t"))
    net_a, tr_a = fresh(11)
    tr_a.run(batches[0]); tr_a.run(batches[1])
    path = tmp_path / "two_steps.pth"
    tr_a.save_checkpoint(str(path))
    ref = tr_a.run(batches[2])                                   # the continuation every resume must reproduce
    net_b, tr_b = fresh(12)                                      # other weights, and it has its own history:
    tr_b.run(batches[2]); tr_b.run(batches[0])
    tr_b.load_checkpoint(str(path), load_only_params=False)
    got = tr_b.run(batches[2])
    assert abs(got["loss"] - ref["loss"]) <= 1e-6 * abs(ref["loss"])
    assert torch.equal(net_a.flat_parameters, net_b.flat_parameters)
    sd = tr_b.optimizer.state_dict()["state"]
    assert float(sd[0]["step"]) == 3.0
    m = tr_b.optimizer._flat_plans and next(iter(tr_b.optimizer._flat_plans.values()))["m"]
    assert sd[0]["exp_avg"].data_ptr() == m.data_ptr()            # the saved tensors are the ones being updated


def test_notebook_inference_recipe(tmp_path, hip_device):
    """load_model + predict_f0 (Utils/dynamic_pitch_behavior.ipynb, cell 5): chunking 192/48, zero-padded
    tail, un-blended overlaps -- against the oracle run chunk by chunk on the oracle's own mel."""
    from oracle import mel_ref
    from pitchextractor_amd import inference, synthetic
    state = model_ref.seeded_state(31, hidden_size=64, num_layers=2)
    torch.save({"model": state, "steps": 0, "epochs": 3}, tmp_path / "m.pth")
    net = inference.load_model(tmp_path / "m.pth", device=hip_device)
    assert net.sequence_classifier.hidden_size == 64 and net.sequence_classifier.num_layers == 2 and net.num_class == 1
    wave, _, _ = synthetic.utterance(2, duration=4.2)            # 337 frames -> chunks at 0, 144, 288
    got = inference.predict_f0(net, wave)
    mel = mel_ref.log_mel(wave)
    st64 = {k: (v.double() if v.dtype.is_floating_point else v) for k, v in state.items()}
    cfg = dict(SEQ_CFG, hidden_size=64, num_layers=2)
    parts = []
    for s in range(0, mel.shape[1], 144):
        e = min(s + 192, mel.shape[1])
        chunk = np.zeros((80, 192))
        chunk[:, :e - s] = mel[:, s:e]
        x = torch.from_numpy(chunk)[None, None].transpose(-1, -2)
        with torch.no_grad():
            f0, _ = model_ref.jdcnet_forward(st64, x, cfg)
        parts.append(f0[0, :e - s, 0].numpy())
    ref = np.concatenate(parts)
    assert got.shape == ref.shape == (192 + 192 + 49,)
    assert np.abs(got - ref).max() <= 2e-3 * np.abs(ref).max()   # fp32 mel + fp32 net vs float64 end to end
