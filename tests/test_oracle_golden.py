"""The CPU oracle reproduces the golden vectors captured from the reference's own modules
(tests/golden/make_golden.py): model forward/backward, optimiser + schedule, data arithmetic."""
import numpy as np
import pytest
import torch

from oracle import model_ref, train_ref
from pitchextractor_amd import synthetic
from tests.golden.make_golden import SEQ_CFG, golden_input, golden_targets, tap_summary


@pytest.fixture(scope="module")
def G(golden_dir):
    return np.load(golden_dir / "model_golden.npz")


def _train_pass(dtype):
    state = {k: (v.to(dtype) if v.dtype.is_floating_point else v)
             for k, v in model_ref.seeded_state(11, num_class=1, hidden_size=384).items()}
    params = {k: v.clone().requires_grad_(True) for k, v in state.items() if v.dtype.is_floating_point
              and not k.endswith(("running_mean", "running_var"))}
    live = dict(state)
    live.update(params)
    taps, stats = {}, {}
    cls, det = model_ref.jdcnet_forward(live, golden_input(3).to(dtype), dict(SEQ_CFG), train=True,
                                        new_stats=stats, taps=taps)
    f0, sil = (t.to(dtype) for t in golden_targets(3))
    loss, lf0, lsil = model_ref.jdc_loss(cls, det, f0, sil, 0.1)
    loss.backward()
    return dict(cls=cls, det=det, loss=(loss.item(), lf0.item(), lsil.item()), params=params, taps=taps,
                stats=stats)


@pytest.fixture(scope="module")
def nc1_train():
    return _train_pass(torch.float32)


@pytest.fixture(scope="module")
def nc1_train64():
    """float64 on both sides removes summation-order noise: the restatement must match to ~1e-9."""
    return _train_pass(torch.float64)


def test_eval_forward_matches_reference(G):
    for tag, nc, hidden in (("nc1", 1, 384), ("nc360", 360, 64)):
        state = model_ref.seeded_state(11, num_class=nc, hidden_size=hidden)
        with torch.no_grad():
            cls, det = model_ref.jdcnet_forward(state, golden_input(3), dict(SEQ_CFG, hidden_size=hidden))
        np.testing.assert_allclose(cls.numpy(), G[f"{tag}_eval_cls"], rtol=2e-4, atol=2e-5)
        np.testing.assert_allclose(det.numpy(), G[f"{tag}_eval_det"], rtol=2e-4, atol=2e-5)


def test_train_forward_and_loss_match_reference(G, nc1_train):
    np.testing.assert_allclose(nc1_train["cls"].detach().numpy(), G["nc1_train_cls"], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(nc1_train["det"].detach().numpy(), G["nc1_train_det"], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(nc1_train["loss"], G["nc1_loss"], rtol=1e-5)


def test_intermediate_taps_match_reference(G, nc1_train):
    t = nc1_train["taps"]
    pairs = {"conv_block": t["convblock_out"], "res_block1": t["resblock1_out"], "res_block2": t["resblock2_out"],
             "res_block3": t["resblock3_out"], "pool_block": t["poolblock_out"], "detector_conv": t["detector_feat"],
             "sequence_classifier": t["seq_classifier_out"], "sequence_detector": t["seq_detector_out"]}
    for name, ours in pairs.items():
        np.testing.assert_allclose(tap_summary(ours), G[f"nc1_tap_{name}"], rtol=5e-4, atol=5e-5, err_msg=name)


def _check_grads(G, tag, run, rtol_norm, rtol_el, atol_frac):
    names = [str(n) for n in G[f"{tag}_grad_names"]]
    norms = dict(zip(names, G[f"{tag}_grad_norms"]))
    assert sorted(names) == sorted(run["params"])
    for n, p in run["params"].items():
        got = p.grad.double().norm().item()
        assert abs(got - norms[n]) <= rtol_norm * norms[n] + 1e-12, (n, got, norms[n])
    for key in G.files:
        if key.startswith(f"{tag}_grad_") and not key.endswith(("_grad_names", "_grad_norms")):
            n = key[len(f"{tag}_grad_"):]
            g = run["params"][n].grad.flatten()
            idx = torch.linspace(0, g.numel() - 1, min(32, g.numel())).long()
            ref = G[key]
            np.testing.assert_allclose(g[idx].numpy(), ref, rtol=rtol_el, atol=atol_frac * np.abs(ref).max() + 1e-14,
                                       err_msg=n)


def test_gradients_match_reference_float64(G, nc1_train64):
    np.testing.assert_allclose(nc1_train64["cls"].detach().numpy(), G["nc1_f64_train_cls"], rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(nc1_train64["det"].detach().numpy(), G["nc1_f64_train_det"], rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(nc1_train64["loss"], G["nc1_f64_loss"], rtol=1e-12)
    _check_grads(G, "nc1_f64", nc1_train64, 1e-9, 1e-7, 1e-9)
    for key in G.files:
        if key.startswith("nc1_f64_stat_"):
            n = key[len("nc1_f64_stat_"):]
            np.testing.assert_allclose(nc1_train64["stats"][n].numpy(), G[key], rtol=1e-10, err_msg=n)


def test_gradients_match_reference_float32(G, nc1_train):
    # fp32 on both sides: summation order (oneDNN vs explicit loops) through 9 train-mode BNs and a
    # 4-layer BiLSTM leaves ~1e-3 relative noise on individual gradient entries
    _check_grads(G, "nc1", nc1_train, 2e-3, 5e-2, 5e-3)


def test_running_stats_match_reference(G, nc1_train):
    for key in G.files:
        if key.startswith("nc1_stat_") :
            n = key[len("nc1_stat_"):]
            np.testing.assert_allclose(nc1_train["stats"][n].numpy(), G[key], rtol=1e-4, atol=1e-6, err_msg=n)


def test_schedule_and_adamw_match_reference(golden_dir):
    O = np.load(golden_dir / "optimizer_golden.npz")
    p = np.linspace(-1.0, 1.0, 24, dtype=np.float32)
    opt = train_ref.AdamWRef(24)
    assert abs(O["lr0"][0] - train_ref.one_cycle(0, 800, 3e-4)[0]) < 1e-12
    assert abs(O["beta1_0"][0] - train_ref.one_cycle(0, 800, 3e-4)[1]) < 1e-12
    for s in range(12):
        lr, b1 = train_ref.one_cycle(s, 800, 3e-4)
        assert abs(lr - O["lr"][s]) < 1e-12 and abs(b1 - O["beta1"][s]) < 1e-12
        p = opt.step(p, O["grads"][s], lr, b1)
        np.testing.assert_allclose(p, O["param"][s], rtol=2e-6, atol=2e-8)  # |p| ~ 1, one f32 ulp = 6e-8
        np.testing.assert_allclose(opt.m, O["exp_avg"][s], rtol=2e-6, atol=2e-8)
        np.testing.assert_allclose(opt.v, O["exp_avg_sq"][s], rtol=2e-6, atol=1e-12)
    for i, (lr_ref, b_ref) in enumerate(O["far"]):
        lr, b1 = train_ref.one_cycle(12 + 37 * i, 800, 3e-4)
        assert abs(lr - lr_ref) < 1e-12 and abs(b1 - b_ref) < 1e-12


def test_align_length_and_glide_match_reference(golden_dir):
    D = np.load(golden_dir / "data_golden.npz")
    for n in (159, 161, 163, 200, 7, 1):
        np.testing.assert_array_equal(train_ref.align_length(D[f"align_in_{n}"], 161), D[f"align_out_{n}"])
    np.testing.assert_array_equal(
        train_ref.align_length(np.array([0, 0, 100, 110, 120, 0, 0, 130, 140, 150.0]), 7), D["align_probe"])
    np.testing.assert_array_equal(D["align_probe"], np.array([0, 50, 110, 60, 0, 135, 150], dtype=np.float32))
    np.testing.assert_array_equal(train_ref.align_length(np.zeros(0), 5), D["align_empty"])
    audio, t, f0 = synthetic.glide(2.0, 60.0, 500.0, 24000)
    np.testing.assert_array_equal(audio[:64], D["glide_head"])
    np.testing.assert_array_equal(audio[-64:], D["glide_tail"])
    np.testing.assert_array_equal(audio[24000:24064], D["glide_mid"])
    np.testing.assert_array_equal(f0[::4800], D["glide_f0"])
    np.testing.assert_array_equal(synthetic.frame_rate_f0(t, f0, 161), D["glide_ref_f0"])


def test_collate_contract():
    items = [(np.ones((80, L), np.float32) * (i + 1), np.full(L, 100.0 + i, np.float32), np.zeros(L, np.float32))
             for i, L in enumerate((161, 192, 100))]
    mels, f0s, sils = train_ref.collate(items)
    assert mels.shape == (3, 1, 80, 192) and f0s.shape == (3, 192) and sils.shape == (3, 192)
    assert (mels[0, 0, :, 161:] == 0).all() and (mels[2, 0, :, 100:] == 0).all() and (mels[1] == 2).all()
    assert (f0s[2, 100:] == 0).all() and (sils == 0).all()      # padded frames: f0 = 0, is_silence = 0


def test_fused_lstm_baseline_path_equals_explicit_loop():
    """bench.py's cpu_baseline uses torch's stock fused LSTM op; it must be the same function."""
    state = model_ref.seeded_state(4, hidden_size=32, num_layers=2)
    x = golden_input(8, B=1, T=24)
    cfg = dict(SEQ_CFG, hidden_size=32, num_layers=2)
    with torch.no_grad():
        a = model_ref.jdcnet_forward(state, x, cfg)
        b = model_ref.jdcnet_forward(state, x, cfg, fused_lstm=True)
    for u, v in zip(a, b):
        np.testing.assert_allclose(u.numpy(), v.numpy(), rtol=1e-4, atol=1e-5)


# ------------------------------------------------------------------ Transformer head
def _tf_pass(dtype):
    from tests.golden.make_golden import TF_CFG
    state = {k: (v.to(dtype) if v.dtype.is_floating_point else v)
             for k, v in model_ref.seeded_state(11, model_type="transformer").items()}
    params = {k: v.clone().requires_grad_(True) for k, v in state.items() if v.dtype.is_floating_point
              and not k.endswith(("running_mean", "running_var", "pos_encoding.pe"))}
    live = dict(state)
    live.update(params)
    cls, det = model_ref.jdcnet_forward(live, golden_input(3).to(dtype), dict(TF_CFG), train=True)
    f0, sil = (t.to(dtype) for t in golden_targets(3))
    loss, lf0, lsil = model_ref.jdc_loss(cls, det, f0, sil, 0.1)
    loss.backward()
    return dict(cls=cls, det=det, loss=(loss.item(), lf0.item(), lsil.item()), params=params)


def test_transformer_head_matches_reference_float64(G):
    run = _tf_pass(torch.float64)
    np.testing.assert_allclose(run["cls"].detach().numpy(), G["tf_f64_train_cls"], rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(run["det"].detach().numpy(), G["tf_f64_train_det"], rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(run["loss"], G["tf_f64_loss"], rtol=1e-12)
    _check_grads(G, "tf_f64", run, 1e-9, 1e-7, 1e-9)


def test_transformer_eval_forward_matches_reference(G):
    from tests.golden.make_golden import TF_CFG
    state = model_ref.seeded_state(11, model_type="transformer")
    with torch.no_grad():
        cls, det = model_ref.jdcnet_forward(state, golden_input(3), dict(TF_CFG))
    np.testing.assert_allclose(cls.numpy(), G["tf_eval_cls"], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(det.numpy(), G["tf_eval_det"], rtol=2e-4, atol=2e-5)


def test_crepe_bin_mapping_known_values():
    """N4 (build-defined): known anchors of the CREPE 360-bin grid -- bin b is centred on
    10 * 2^((20 b + 1997.379...)/1200) Hz, so bin 0 = 31.70 Hz (C1 - ~0.2 semitones) and bin 359 = 2005.5 Hz."""
    import numpy as np
    from oracle import model_ref
    centres = 10.0 * 2.0 ** ((20.0 * np.arange(360) + model_ref.CREPE_CENTS0) / 1200.0)
    assert abs(centres[0] - 31.70) < 0.01 and abs(centres[359] - 2005.5) < 0.5
    bins = model_ref.f0_to_bins(centres)
    assert (bins == np.arange(360)).all()
    assert list(model_ref.f0_to_bins([0.0, 5.0, 440.0, 1e5])) == [-1, 0, 228, 359]
    edge = 10.0 * 2.0 ** ((20.0 * 100.5 + model_ref.CREPE_CENTS0) / 1200.0)        # exactly between bins 100 and 101
    assert model_ref.f0_to_bins([edge * (1 - 1e-9), edge * (1 + 1e-9)]).tolist() == [100, 101]
