"""Transformer temporal head on the HIP path: each new op vs a PyTorch fp64 reference, then the whole
JDCNet(model_type="transformer") vs the reference's golden vectors (float64) and the oracle with
replayed dropout masks."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import model_ref
from pitchextractor_amd import ops
from pitchextractor_amd.model import JDCNet
from tests.golden.make_golden import TF_CFG, golden_input, golden_targets
from tests.test_ops_gpu import close, rnd

pytestmark = pytest.mark.gpu


def test_batched_attention_products(hip_device):
    B, H, T, dh = 3, 8, 192, 64
    D = H * dh
    qkv = rnd(B * T, 3 * D, seed=1).to(hip_device)
    q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
    hview, pview, oview = (3 * D, T * 3 * D, dh), (T, H * T * T, T * T), (D, T * D, dh)
    heads = lambda t: t.cpu().double().view(B, T, H, dh).transpose(1, 2)          # (B,H,T,dh)
    S = torch.empty(B * H * T, T, device=hip_device)
    ops.bgemm(0, q, hview, k, hview, S, pview, H, B * H, T, T, dh, alpha=0.5)
    close(S.view(B, H, T, T), 0.5 * heads(q) @ heads(k).transpose(-1, -2))
    P = torch.softmax(rnd(B * H * T, T, seed=2), dim=-1).to(hip_device)
    O = torch.empty(B * T, D, device=hip_device)
    ops.bgemm(1, P, pview, v, hview, O, oview, H, B * H, T, dh, T)
    refO = (P.cpu().double().view(B, H, T, T) @ heads(v)).transpose(1, 2).reshape(B * T, D)
    close(O, refO)
    dO = rnd(B * T, D, seed=3).to(hip_device)
    dqkv = torch.zeros_like(qkv)
    ops.bgemm(2, P, pview, dO, oview, dqkv[:, 2 * D:], hview, H, B * H, T, dh, T)
    ref_dv = (P.cpu().double().view(B, H, T, T).transpose(-1, -2) @ heads(dO)).transpose(1, 2).reshape(B * T, D)
    close(dqkv[:, 2 * D:], ref_dv)
    assert (dqkv[:, :2 * D] == 0).all()
    acc = rnd(B * T, D, seed=4).to(hip_device)
    ref_acc = acc.cpu().double() + refO
    ops.bgemm(1, P, pview, v, hview, acc, oview, H, B * H, T, dh, T, accumulate=True)
    close(acc, ref_acc)


def test_softmax_fwd_bwd(hip_device):
    s = (rnd(500, 192, seed=1) * 4).double().requires_grad_(True)
    p = torch.softmax(s * 0.125, dim=-1)
    dp = rnd(500, 192, seed=2).double()
    p.backward(dp)
    sd = s.detach().float().to(hip_device)
    ops.softmax_fwd_(sd, 0.125)
    close(sd, p)
    dpd = dp.float().to(hip_device)
    ops.softmax_bwd_(sd, dpd, 0.125)
    close(dpd, s.grad)
    odd = (rnd(7, 100, seed=3)).to(hip_device)                 # row length not a multiple of 64
    ref = torch.softmax(odd.cpu().double(), dim=-1)
    close(ops.softmax_fwd_(odd, 1.0), ref)


@pytest.mark.parametrize("D", [256, 512, 768])
def test_layernorm_fwd_bwd(hip_device, D):
    R, T = 333, 37
    a = rnd(R, D, seed=1).double().requires_grad_(True)
    b = rnd(R, D, seed=2).double().requires_grad_(True)
    pe = rnd(T, D, seed=3).double()
    g = (rnd(D, seed=4).abs() + 0.5).double().requires_grad_(True)
    be = rnd(D, seed=5).double().requires_grad_(True)
    z = a + b + pe[torch.arange(R) % T]
    y = F.layer_norm(z, (D,), g, be, 1e-5)
    dy = rnd(R, D, seed=6).double()
    y.backward(dy)
    dev = hip_device
    f = lambda t: t.detach().float().to(dev)
    yd, st = ops.layernorm_fwd(f(a), f(g), f(be), b2d=f(b), pe=f(pe))
    close(yd, y)
    close(st.z, z.detach())
    dg, db = torch.empty(D, device=dev), torch.empty(D, device=dev)
    dz = ops.layernorm_bwd(f(dy), st, f(g), dg, db)
    close(dz, a.grad, 2e-5)
    close(dg, g.grad, 2e-5)
    close(db, be.grad, 2e-5)
    y2, st2 = ops.layernorm_fwd(f(a), f(g), f(be))              # no residual: z is the input itself
    close(y2, F.layer_norm(a.detach(), (D,), g.detach(), be.detach(), 1e-5))
    assert st2.z.data_ptr() != 0


def test_gelu_exact(hip_device):
    x = (rnd(1000, 1536, seed=1) * 2).double().requires_grad_(True)
    y = F.gelu(x)                                                  # erf form
    dy = rnd(1000, 1536, seed=2).double()
    y.backward(dy)
    xd = x.detach().float().to(hip_device)
    close(ops.gelu_fwd(xd), y)
    close(ops.gelu_bwd(xd, dy.float().to(hip_device)), x.grad)


# ------------------------------------------------------------------ whole model
def build(state, device, dropout=0.0):
    net = JDCNet(num_class=1, sequence_model_config=dict(TF_CFG, dropout=dropout))
    net.load_state_dict(state, strict=True)
    return net.to(device)


@pytest.fixture(scope="module")
def G(golden_dir):
    return np.load(golden_dir / "model_golden.npz")


def _rel(got, ref):
    got = np.asarray(got.detach().cpu().double()); ref = np.asarray(ref, dtype=np.float64)
    return np.abs(got - ref).max() / (np.abs(ref).max() + 1e-30)


def test_transformer_eval_matches_reference(G, hip_device):
    net = build(model_ref.seeded_state(11, model_type="transformer"), hip_device).eval()
    with torch.no_grad():
        cls, det = net(golden_input(3).to(hip_device))
    assert _rel(cls, G["tf_eval_cls"]) <= 1e-4 and _rel(det, G["tf_eval_det"]) <= 1e-4


def test_transformer_train_step_matches_reference_float64(G, hip_device):
    net = build(model_ref.seeded_state(11, model_type="transformer"), hip_device).train()
    net.block_dropout = 0.0
    f0, sil = (t.to(hip_device) for t in golden_targets(3))
    cls, det = net(golden_input(3).to(hip_device))
    out3, d_f0, d_sil = ops.f0_sil_loss(cls.detach().reshape(-1), f0.reshape(-1), det.detach().reshape(-1),
                                        sil.reshape(-1), 0.1)
    torch.autograd.backward([cls, det], [d_f0.view_as(cls), d_sil.view_as(det)])
    assert _rel(cls, G["tf_f64_train_cls"]) <= 1e-4 and _rel(det, G["tf_f64_train_det"]) <= 1e-4
    np.testing.assert_allclose(out3.cpu().numpy(), G["tf_f64_loss"], rtol=1e-5)
    names = [str(n) for n in G["tf_f64_grad_names"]]
    assert names == [n for n, _ in net.named_parameters()]
    norms = dict(zip(names, G["tf_f64_grad_norms"]))
    bad = [(n, p.grad.double().norm().item(), norms[n]) for n, p in net.named_parameters()
           if abs(p.grad.double().norm().item() - norms[n]) > 2e-3 * norms[n] + 1e-9]
    assert not bad, bad
    for key in G.files:
        if key.startswith("tf_f64_grad_") and not key.endswith(("_grad_names", "_grad_norms")):
            n = key[len("tf_f64_grad_"):]
            g = dict(net.named_parameters())[n].grad.flatten().cpu()
            idx = torch.linspace(0, g.numel() - 1, min(32, g.numel())).long()
            assert np.abs(g[idx].numpy() - G[key]).max() <= 5e-3 * np.abs(G[key]).max() + 1e-9, n


def test_transformer_dropout_masks_replayed_in_oracle(hip_device):
    state = model_ref.seeded_state(11, model_type="transformer", num_layers=2)
    cfg = dict(TF_CFG, num_layers=2, dropout=0.1)
    net = JDCNet(num_class=1, sequence_model_config=dict(cfg))
    net.load_state_dict(state)
    net = net.to(hip_device).train()
    net.keep_last_context = True
    net.dropout_cfg.seed = 5
    x = golden_input(6)
    cls, det = net(x.to(hip_device))
    s = net.last_context
    masks = [s.mask_pool.cpu(), s.mask_det.cpu()]
    for tf in (s.tf_c, s.tf_d):
        for c in tf.layers:
            masks += [c.mask_p.cpu(), c.mask1.cpu(), c.mask_f.cpu(), c.mask2.cpu()]
    st64 = {k: (v.double() if v.dtype.is_floating_point else v) for k, v in state.items()}
    rc, rd = model_ref.jdcnet_forward(st64, x.double(), cfg, train=True, masks=iter(masks))
    assert _rel(cls, rc.detach().numpy()) <= 1e-4 and _rel(det, rd.detach().numpy()) <= 1e-4


# ------------------------------------------------------------------ BASELINE config[2]: Transformer head + mixed precision
# bf16 operands carry 8 significant bits (relative rounding 2^-9 = 2e-3 per operand); through ~12 conv and
# ~26 linear products the logits drift by a few 1e-3 of their scale and gradient norms by ~1 %.  Stated
# tolerances: logits 2e-2 of scale, loss 1e-2 relative, per-parameter gradient norms 8e-2 relative (measured:
# logits 1.1e-2 / 1.5e-2, loss 5e-5, worst gradient norm 5.1e-2 on a BatchNorm gamma, where the sum cancels).
BF16_LOGIT_TOL, BF16_LOSS_TOL, BF16_GRAD_TOL = 2e-2, 1e-2, 8e-2


def test_transformer_mixed_precision_step_matches_reference_float64(G, hip_device):
    net = build(model_ref.seeded_state(11, model_type="transformer"), hip_device).train()
    net.block_dropout = 0.0
    f0, sil = (t.to(hip_device) for t in golden_targets(3))
    with ops.matmul_bf16(True):
        cls, det = net(golden_input(3).to(hip_device))
        out3, d_f0, d_sil = ops.f0_sil_loss(cls.detach().reshape(-1), f0.reshape(-1), det.detach().reshape(-1),
                                            sil.reshape(-1), 0.1)
        torch.autograd.backward([cls, det], [d_f0.view_as(cls), d_sil.view_as(det)])
    r_cls, r_det = _rel(cls, G["tf_f64_train_cls"]), _rel(det, G["tf_f64_train_det"])
    assert r_cls <= BF16_LOGIT_TOL and r_det <= BF16_LOGIT_TOL, (r_cls, r_det)
    assert r_cls > 1e-6                                  # the mode really rounds operands
    np.testing.assert_allclose(out3.cpu().numpy()[0], G["tf_f64_loss"][0], rtol=BF16_LOSS_TOL)
    norms = dict(zip([str(n) for n in G["tf_f64_grad_names"]], G["tf_f64_grad_norms"]))
    bad = [(n, p.grad.double().norm().item(), norms[n]) for n, p in net.named_parameters()
           if abs(p.grad.double().norm().item() - norms[n]) > BF16_GRAD_TOL * norms[n] + 1e-9]
    assert not bad, bad


def test_transformer_mixed_precision_curve_tracks_cpu_oracle_trainer(hip_device):
    """12 optimiser steps of the HIP trainer with use_mixed_precision=True (Transformer head, dropout off)
    against the fp32 CPU oracle trainer (oracle/train_ref.CpuTrainer = reference Trainer.run restated) on
    identical batches: loss within 2 %."""
    import logging
    from oracle import train_ref
    from pitchextractor_amd.optimizers import build_optimizer
    from pitchextractor_amd.trainer import Trainer
    from tests.golden.make_golden import training_batches
    state = model_ref.seeded_state(21, model_type="transformer")
    net = build(state, hip_device).train()
    net.block_dropout = 0.0
    opt, sched = build_optimizer({"params": net.parameters(), "optimizer_params": {},
                                  "scheduler_params": {"max_lr": 3e-4, "pct_start": 0.0, "epochs": 100,
                                                       "steps_per_epoch": 8}})
    tr = Trainer(model=net, criterion={"l1": torch.nn.SmoothL1Loss(), "ce": torch.nn.BCEWithLogitsLoss()},
                 optimizer=opt, scheduler=sched, device="cuda:0", loss_config={"lambda_f0": 0.1},
                 logger=logging.getLogger("t"), use_mixed_precision=True)
    torch.set_num_threads(16)
    cpu = train_ref.CpuTrainer(state, dict(TF_CFG), max_lr=3e-4, total_steps=800, lambda_f0=0.1)
    for i, batch in enumerate(training_batches(12)):
        got, ref = tr.run(batch), cpu.run(batch)
        assert abs(got["loss"] - ref["loss"]) <= 2e-2 * abs(ref["loss"]), (i, got, ref)
        assert abs(got["sil"] - ref["sil"]) <= 2e-2 * abs(ref["sil"]) + 2e-3, (i, got, ref)


# ------------------------------------------------------------------ fused attention (T = 192, dh = 64)
@pytest.mark.parametrize("p", [0.0, 0.1])
def test_fused_attention_fwd_bwd(hip_device, p):
    """pe_attn_fwd / pe_attn_bwd against float64 softmax(QK^T/8) -> dropout(mask) -> V and its autograd gradient;
    the dropout keep mask is the kernel's own (exported), as in the whole-model mask-replay tests; and the Philox
    stream equals pe_dropout_fwd's on the (B*H*T) x T matrix, so fused and unfused paths draw identical masks."""
    B, H, T, dh = 3, 8, 192, 64
    D = H * dh
    assert ops.attn_supported(T, dh)
    qkv = (rnd(B * T, 3 * D, seed=1) * 1.5).to(hip_device)
    o, lse, mask = ops.attn_fwd(qkv, B, T, H, 0.125, p, seed=11, offset=1000)
    ref_in = qkv.cpu().double().requires_grad_(True)
    q, k, v = (ref_in[:, i * D:(i + 1) * D].view(B, T, H, dh).transpose(1, 2) for i in range(3))    # (B,H,T,dh)
    s = (q @ k.transpose(-1, -2)) * 0.125
    P = torch.softmax(s, dim=-1)
    if p > 0:
        keep = mask.cpu().view(B, H, T, T).double()
        assert abs(keep.mean().item() - (1 - p)) < 5e-3
        probe = torch.ones(B * H * T, T, device=hip_device)
        _, m2 = ops.dropout(probe, p, seed=11, offset=1000)
        assert torch.equal(m2, mask)                                   # same Philox stream as the unfused dropout
        P = P * keep / (1 - p)
    else:
        assert mask is None
    ref_o = (P @ v).transpose(1, 2).reshape(B * T, D)
    close(o, ref_o.detach(), 1e-5)
    close(lse.view(B, H, T), torch.logsumexp(s, dim=-1).detach(), 1e-5)
    d_o = rnd(B * T, D, seed=2)
    ref_o.backward(d_o.double())
    dqkv = ops.attn_bwd(qkv, o, d_o.to(hip_device), lse, mask, B, T, H, 0.125, p)
    for i, name in enumerate("qkv"):
        close(dqkv[:, i * D:(i + 1) * D], ref_in.grad[:, i * D:(i + 1) * D], 2e-5)
    # replaying the exported mask reproduces the forward bit for bit
    if p > 0:
        o2, _, _ = ops.attn_fwd(qkv, B, T, H, 0.125, p, mask_in=mask)
        assert torch.equal(o, o2)


@pytest.mark.parametrize("p", [0.0, 0.1])
def test_fused_attention_bf16_operands(hip_device, p):
    """Mixed precision (bf16): pe_attn_fwd_bf16 / pe_attn_bwd_bf16 round the matmul operands to bf16 and keep softmax,
    log-sum-exp and accumulation in fp32.  Forward against a float64 restatement that rounds at the same points (Q, K, V
    and the dropped-out probabilities): 1e-3 of the tensor maximum (a probability that sits near a bf16 rounding boundary
    rounds differently from fp32 than from float64: measured 3.5e-4); log-sum-exp likewise; gradients against the EXACT
    float64 gradient within 2e-2 of each tensor's maximum (bf16 operands: 2^-9 per product term); same masks and the
    same Philox stream as the fp32 kernels."""
    B, H, T, dh = 3, 8, 192, 64
    D = H * dh
    qkv = (rnd(B * T, 3 * D, seed=1) * 1.5).to(hip_device)
    d_o = rnd(B * T, D, seed=2)
    o32, lse32, mask32 = ops.attn_fwd(qkv, B, T, H, 0.125, p, seed=11, offset=1000)
    with ops.matmul_bf16(True, "bf16"):
        o, lse, mask = ops.attn_fwd(qkv, B, T, H, 0.125, p, seed=11, offset=1000)
        dqkv = ops.attn_bwd(qkv, o, d_o.to(hip_device), lse, mask, B, T, H, 0.125, p)
    if p > 0:
        assert torch.equal(mask, mask32)
    r16 = lambda t: t.to(torch.bfloat16).double()
    x = qkv.cpu()
    q, k, v = (r16(x[:, i * D:(i + 1) * D]).view(B, T, H, dh).transpose(1, 2) for i in range(3))
    s = (q @ k.transpose(-1, -2)) * 0.125
    P = torch.softmax(s, dim=-1)
    if p > 0:
        P = P * mask.cpu().view(B, H, T, T).double() / (1 - p)
    ref_o = (r16(P.float()) @ v).transpose(1, 2).reshape(B * T, D)
    assert (o.cpu().double() - ref_o).abs().max().item() <= 1e-3 * ref_o.abs().max().item()
    assert (lse.cpu().double().view(B, H, T) - torch.logsumexp(s, dim=-1)).abs().max().item() <= 1e-5 * s.abs().max().item()
    assert (o - o32).abs().max().item() <= 2e-2 * o32.abs().max().item()
    # exact gradient of the exact forward
    ref_in = x.double().requires_grad_(True)
    q, k, v = (ref_in[:, i * D:(i + 1) * D].view(B, T, H, dh).transpose(1, 2) for i in range(3))
    Pe = torch.softmax((q @ k.transpose(-1, -2)) * 0.125, dim=-1)
    if p > 0:
        Pe = Pe * mask.cpu().view(B, H, T, T).double() / (1 - p)
    (Pe @ v).transpose(1, 2).reshape(B * T, D).backward(d_o.double())
    for i in range(3):
        ref = ref_in.grad[:, i * D:(i + 1) * D]
        err = (dqkv[:, i * D:(i + 1) * D].cpu().double() - ref).abs().max().item()
        assert err <= 2e-2 * ref.abs().max().item(), ("qkv"[i], err, ref.abs().max().item())


def test_fused_and_unfused_attention_agree_in_the_model(hip_device, monkeypatch):
    """Whole Transformer-head JDCNet, train mode with live dropout: the fused attention path and the
    pe_bgemm + pe_softmax path give the same logits and gradients (same masks: same Philox offsets)."""
    state = model_ref.seeded_state(11, model_type="transformer", num_layers=2)
    cfg = dict(TF_CFG, num_layers=2, dropout=0.1)
    outs = {}
    for fused in (True, False):
        monkeypatch.setattr(ops, "FUSED_ATTENTION", fused)
        net = JDCNet(num_class=1, sequence_model_config=dict(cfg))
        net.load_state_dict(state)
        net = net.to(hip_device).train()
        net.dropout_cfg.seed = 5
        f0, sil = (t.to(hip_device) for t in golden_targets(6))
        cls, det = net(golden_input(6).to(hip_device))
        out3, d_f0, d_sil = ops.f0_sil_loss(cls.detach().reshape(-1), f0.reshape(-1), det.detach().reshape(-1),
                                            sil.reshape(-1), 0.1)
        torch.autograd.backward([cls, det], [d_f0.view_as(cls), d_sil.view_as(det)])
        outs[fused] = (cls.detach(), det.detach(), net.flat_gradients().clone(), net.dropout_cfg.offset)
    assert outs[True][3] == outs[False][3]
    assert _rel(outs[True][0], outs[False][0].cpu().numpy()) <= 2e-5 and _rel(outs[True][1], outs[False][1].cpu().numpy()) <= 2e-5
    ga, gb = outs[True][2].double(), outs[False][2].double()
    assert ((ga - gb).norm() / gb.norm()).item() <= 1e-4


def test_side_stream_weight_gradients_change_nothing_transformer(hip_device, monkeypatch):
    """Transformer-head JDCNet, train mode with live dropout, B = 16: with the weight-gradient kernels on the side
    stream the logits and the flat gradient buffer are bit-identical to the serialised backward."""
    from pitchextractor_amd import model as pe_model
    state = model_ref.seeded_state(11, model_type="transformer", num_layers=2)
    cfg = dict(TF_CFG, num_layers=2, dropout=0.1)
    x = golden_input(6).repeat(8, 1, 1, 1)
    f0, sil = (t.repeat(8, 1).to(hip_device) for t in golden_targets(6))
    outs = []
    for on in (False, True, True):
        for flag in ("OVERLAP_LSTM_WGRAD", "OVERLAP_CONV_WGRAD", "OVERLAP_TF_WGRAD"):
            monkeypatch.setattr(pe_model, flag, on)
        net = JDCNet(num_class=1, sequence_model_config=dict(cfg))
        net.load_state_dict(state)
        net = net.to(hip_device).train()
        net.dropout_cfg.seed = 5
        cls, det = net(x.to(hip_device))
        out3, d_f0, d_sil = ops.f0_sil_loss(cls.detach().reshape(-1), f0.reshape(-1), det.detach().reshape(-1),
                                            sil.reshape(-1), 0.1)
        torch.autograd.backward([cls, det], [d_f0.view_as(cls), d_sil.view_as(det)])
        outs.append((cls.detach().clone(), net.flat_gradients().clone()))
    for cls1, g1 in outs[1:]:
        assert torch.equal(cls1, outs[0][0]) and torch.equal(g1, outs[0][1])


@pytest.mark.parametrize("inject", [False, True])
def test_dropout_folded_into_layernorm_and_gelu_ops(hip_device, inject):
    """The fused passes equal the separate ones bit for bit: same Philox counters / replayed bytes, same arithmetic."""
    R, D, FF, p = 96, 512, 1024, 0.1
    a, b = rnd(R, D, seed=1).to(hip_device), rnd(R, D, seed=2).to(hip_device)
    gam, bet = rnd(D, seed=3).to(hip_device), rnd(D, seed=4).to(hip_device)
    m_in = (torch.rand(R, D, generator=torch.Generator().manual_seed(5)) >= p).to(torch.uint8).to(hip_device) \
        if inject else None
    bd, m_ref = ops.dropout(b, p, mask_in=m_in, seed=9, offset=1234)
    y_ref, st_ref = ops.layernorm_fwd(a, gam, bet, b2d=bd)
    y, st, m = ops.layernorm_dropout_fwd(a, b, gam, bet, p, mask_in=m_in, seed=9, offset=1234)
    assert torch.equal(m, m_ref) and torch.equal(y, y_ref) and torch.equal(st.z, st_ref.z)
    assert torch.equal(st.mean, st_ref.mean) and torch.equal(st.rstd, st_ref.rstd)
    assert 0.05 < 1.0 - m.float().mean().item() < 0.15
    # backward: dz = dLN(dy + dy2), second output dropout_bwd(dz)
    dy, dy2 = rnd(R, D, seed=6).to(hip_device), rnd(R, D, seed=7).to(hip_device)
    dg_ref, db_ref = torch.empty(D, device=hip_device), torch.empty(D, device=hip_device)
    dz_ref = ops.layernorm_bwd(dy + dy2, st_ref, gam, dg_ref, db_ref)
    dzd_ref, _ = ops.dropout(dz_ref, p, mask_in=m_ref)
    dg, db = torch.empty(D, device=hip_device), torch.empty(D, device=hip_device)
    dz, dzd = ops.layernorm_bwd(dy, st, gam, dg, db, dy_add=dy2, drop_mask=m, p=p)
    assert torch.equal(dz, dz_ref) and torch.equal(dzd, dzd_ref) and torch.equal(dg, dg_ref) and torch.equal(db, db_ref)
    dz_only = ops.layernorm_bwd(dy, st, gam, dg, db, dy_add=dy2)
    assert torch.equal(dz_only, dz_ref)
    # GELU
    h, da = rnd(R, FF, seed=8).to(hip_device), rnd(R, FF, seed=10).to(hip_device)
    mf_in = (torch.rand(R, FF, generator=torch.Generator().manual_seed(11)) >= p).to(torch.uint8).to(hip_device) \
        if inject else None
    act_ref, mf_ref = ops.dropout(ops.gelu_fwd(h), p, mask_in=mf_in, seed=9, offset=77)
    act, mf = ops.gelu_dropout_fwd(h, p, mask_in=mf_in, seed=9, offset=77)
    assert torch.equal(mf, mf_ref) and torch.equal(act, act_ref)
    dh_ref = ops.gelu_bwd(h, ops.dropout(da, p, mask_in=mf_ref)[0])
    assert torch.equal(ops.gelu_dropout_bwd(h, da, mf, p), dh_ref)


@pytest.mark.parametrize("dropout", [0.1, 0.0])
def test_folded_dropout_changes_nothing_in_the_model(hip_device, monkeypatch, dropout):
    """Transformer-head JDCNet, train mode: logits and the flat gradient buffer with the dropouts / residual adds folded
    into the LayerNorm and GELU passes are bit-identical to the separate passes (live Philox masks)."""
    from pitchextractor_amd import model as pe_model
    state = model_ref.seeded_state(11, model_type="transformer", num_layers=2)
    cfg = dict(TF_CFG, num_layers=2, dropout=dropout)
    x = golden_input(6).repeat(4, 1, 1, 1)
    f0, sil = (t.repeat(4, 1).to(hip_device) for t in golden_targets(6))
    outs = []
    for fused in (False, True):
        monkeypatch.setattr(pe_model, "TF_FUSE_DROPOUT", fused)
        net = JDCNet(num_class=1, sequence_model_config=dict(cfg))
        net.load_state_dict(state)
        net = net.to(hip_device).train()
        net.dropout_cfg.seed = 5
        cls, det = net(x.to(hip_device))
        out3, d_f0, d_sil = ops.f0_sil_loss(cls.detach().reshape(-1), f0.reshape(-1), det.detach().reshape(-1),
                                            sil.reshape(-1), 0.1)
        torch.autograd.backward([cls, det], [d_f0.view_as(cls), d_sil.view_as(det)])
        outs.append((cls.detach().clone(), net.flat_gradients().clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("amp", [False, True])
def test_full_size_transformer_step_by_tiling(hip_device, amp):
    """BASELINE config[2]'s shape (B = 256, Transformer head, fp32 and bf16 operands): eight samples tiled x32.
    BatchNorm batch statistics, the mean loss and every parameter gradient of the tiled batch equal those of the
    eight-sample batch in real arithmetic, and the B <= 8 path is pinned to the reference's float64 golden above --
    so this is where the fused attention grid (256 x 8 heads x 3 parts), the projection GEMMs at M = 49 152 and the
    LayerNorm / GELU passes run at bench shape.  Both runs round their operands alike, so only summation orders
    differ: logits per replica within 1e-4 of scale of the B = 8 run, loss 1e-6, every gradient element within 1e-4
    of its tensor's largest, in both operand modes."""
    state = model_ref.seeded_state(11, model_type="transformer")
    x8 = golden_input(5, B=8)
    f0, sil = golden_targets(5, B=8)

    def step(x, f0, sil):
        net = build(state, hip_device).train()
        net.block_dropout = 0.0
        with ops.matmul_bf16(amp, "bf16"):
            cls, det = net(x.to(hip_device))
            out3, d_f0, d_sil = ops.f0_sil_loss(cls.detach().reshape(-1), f0.to(hip_device).reshape(-1),
                                                det.detach().reshape(-1), sil.to(hip_device).reshape(-1), 0.1)
            torch.autograd.backward([cls, det], [d_f0.view_as(cls), d_sil.view_as(det)])
        return cls.detach(), det.detach(), out3[0].item(), {n: p.grad.detach().clone() for n, p in net.named_parameters()}

    cls8, det8, loss8, g8 = step(x8, f0, sil)
    cls, det, loss, g = step(x8.repeat(32, 1, 1, 1), f0.repeat(32, 1), sil.repeat(32, 1))
    assert cls.shape == (256, 192, 1) and np.isfinite(loss)
    tol_out, tol_loss, tol_g = 1e-4, 1e-6, 1e-4          # measured: gradients 3.4e-6 (fp32) / 2.1e-6 (bf16 operands)
    for r in (0, 13, 31):
        close(cls[8 * r:8 * r + 8], cls8.cpu(), tol_out)
        close(det[8 * r:8 * r + 8], det8.cpu(), tol_out)
    assert abs(loss - loss8) <= tol_loss * abs(loss8)
    bad, worst = [], 0.0
    for n, got in g.items():
        top = g8[n].abs().max().item()
        err = (got - g8[n]).abs().max().item()
        worst = max(worst, err / (top + 1e-30))
        if err > tol_g * top + 1e-9:
            bad.append((n, err, top))
    print(f"amp={amp}: loss {loss:.6f} vs {loss8:.6f}; worst gradient element error / tensor max = {worst:.2e}")
    assert not bad, bad
