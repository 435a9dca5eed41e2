"""Functional CPU oracle of JDCNet (TEST INFRASTRUCTURE ONLY -- never imported by the product).

A stateless restatement on stock ``torch.nn.functional`` ops: ``jdcnet_forward(state, x, cfg)``
takes the reference's ``state_dict`` (same keys) and reproduces ``JDCNet.forward``
(reference model.py:75-122) in NCHW, including the train-mode BatchNorm statistics
(model.py:25,37,54,150,159), the residual blocks (model.py:143-175), the detector concat
(model.py:103-109) and an explicit-loop LSTM with torch's gate order i,f,g,o
(model.py:218-227,249-252).  Pinned against the reference itself by
``tests/golden/make_golden.py`` (golden vectors in ``tests/golden/model_*.npz``).

Works in float32 or float64 (dtype follows ``x``/weights); backward comes from torch autograd.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn.functional as F

SLOPE = 0.01
BN_EPS = 1e-5
BN_MOMENTUM = 0.1


def _batchnorm(x, state, prefix, train, new_stats):
    w, b = state[prefix + ".weight"], state[prefix + ".bias"]
    rm, rv = state[prefix + ".running_mean"], state[prefix + ".running_var"]
    if train:
        mean = x.mean(dim=(0, 2, 3))
        var = x.var(dim=(0, 2, 3), unbiased=False)
        n = x.numel() // x.shape[1]
        if new_stats is not None:
            with torch.no_grad():
                new_stats[prefix + ".running_mean"] = (1 - BN_MOMENTUM) * rm + BN_MOMENTUM * mean
                new_stats[prefix + ".running_var"] = (1 - BN_MOMENTUM) * rv + BN_MOMENTUM * var * n / max(n - 1, 1)
    else:
        mean, var = rm, rv
    xhat = (x - mean[None, :, None, None]) / torch.sqrt(var[None, :, None, None] + BN_EPS)
    return xhat * w[None, :, None, None] + b[None, :, None, None]


def _conv(x, w):
    return F.conv2d(x, w, bias=None, padding=w.shape[-1] // 2)


def _drop(x, p, train, masks):
    """Inverted dropout with an explicit keep-mask (taken from ``masks``) or identity."""
    if not train or p <= 0:
        return x
    if masks is None:
        raise ValueError("train-mode oracle with dropout > 0 needs explicit masks")
    m = next(masks).to(x.dtype).reshape(x.shape)
    return x * m / (1.0 - p)


def _res_block(x, state, name, train, new_stats):
    h = _batchnorm(x, state, f"{name}.pre_conv.0", train, new_stats)
    h = F.leaky_relu(h, SLOPE)
    h = F.max_pool2d(h, kernel_size=(1, 2))
    y = _conv(h, state[f"{name}.conv.0.weight"])
    y = _batchnorm(y, state, f"{name}.conv.1", train, new_stats)
    y = F.leaky_relu(y, SLOPE)
    y = _conv(y, state[f"{name}.conv.3.weight"])
    return y + _conv(h, state[f"{name}.conv1by1.weight"])


def lstm_layer_direction(x, w_ih, w_hh, b_ih, b_hh, reverse):
    """x (B,T,in) -> h (B,T,H); zero initial state; gates ordered i,f,g,o."""
    B, T, _ = x.shape
    H = w_hh.shape[1]
    proj = x @ w_ih.T + b_ih + b_hh
    h = x.new_zeros(B, H)
    c = x.new_zeros(B, H)
    outs = [None] * T
    order = range(T - 1, -1, -1) if reverse else range(T)
    for t in order:
        g = proj[:, t] + h @ w_hh.T
        i, f, gg, o = g[:, :H], g[:, H:2 * H], g[:, 2 * H:3 * H], g[:, 3 * H:]
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        h = torch.sigmoid(o) * torch.tanh(c)
        outs[t] = h
    return torch.stack(outs, dim=1)


def bilstm_fused(x, state, prefix, num_layers, bidirectional):
    """Same network through torch's stock fused LSTM op (what nn.LSTM dispatches to; oneDNN on CPU).
    Used for the timed CPU baseline, where the explicit loop would understate the reference."""
    flat = []
    for layer in range(num_layers):
        for d in range(2 if bidirectional else 1):
            sfx = f"_l{layer}" + ("_reverse" if d else "")
            flat += [state[f"{prefix}.weight_ih{sfx}"], state[f"{prefix}.weight_hh{sfx}"],
                     state[f"{prefix}.bias_ih{sfx}"], state[f"{prefix}.bias_hh{sfx}"]]
    H = flat[1].shape[1]
    nd = 2 if bidirectional else 1
    h0 = x.new_zeros(num_layers * nd, x.shape[0], H)
    out, _, _ = torch._VF.lstm(x, (h0, h0.clone()), flat, True, num_layers, 0.0, False, bidirectional, True)
    return out


def bilstm(x, state, prefix, num_layers, bidirectional, p_drop, train, masks):
    for layer in range(num_layers):
        outs = []
        for d in range(2 if bidirectional else 1):
            sfx = f"_l{layer}" + ("_reverse" if d else "")
            outs.append(lstm_layer_direction(
                x, state[f"{prefix}.weight_ih{sfx}"], state[f"{prefix}.weight_hh{sfx}"],
                state[f"{prefix}.bias_ih{sfx}"], state[f"{prefix}.bias_hh{sfx}"], reverse=bool(d)))
        x = torch.cat(outs, dim=-1)
        if layer < num_layers - 1:
            x = _drop(x, p_drop, train, masks)
    return x


def transformer_head(x, state, prefix, seq_cfg, train, masks):
    """SequenceModel(model_type="transformer") (reference model.py:229-241,253-255): x + pe -> LayerNorm ->
    N post-norm encoder layers (MHA, residual, LN, Linear-GELU-Linear, residual, LN).  Dropout masks are
    consumed per layer in the order attention-probs, dropout1, feed-forward, dropout2."""
    B, T, D = x.shape
    H = seq_cfg.get("nhead", 8)
    dh = D // H
    p = seq_cfg.get("dropout", 0.3) if (train and masks is not None) else 0.0
    pe = state[f"{prefix}.pos_encoding.pe"][0, :T].to(x.dtype)
    x = F.layer_norm(x + pe, (D,), state[f"{prefix}.layer_norm.weight"], state[f"{prefix}.layer_norm.bias"], 1e-5)
    for i in range(seq_cfg.get("num_layers", 2)):
        L = f"{prefix}.model.layers.{i}"
        qkv = x @ state[f"{L}.self_attn.in_proj_weight"].T + state[f"{L}.self_attn.in_proj_bias"]
        q, k, v = (t.reshape(B, T, H, dh).transpose(1, 2) for t in qkv.split(D, dim=-1))     # (B,H,T,dh)
        probs = torch.softmax((q @ k.transpose(-1, -2)) / (dh ** 0.5), dim=-1)
        probs = _drop(probs.reshape(B * H * T, T), p, train, masks).reshape(B, H, T, T)
        o = (probs @ v).transpose(1, 2).reshape(B, T, D)
        sa = o @ state[f"{L}.self_attn.out_proj.weight"].T + state[f"{L}.self_attn.out_proj.bias"]
        sa = _drop(sa.reshape(B * T, D), p, train, masks).reshape(B, T, D)
        x = F.layer_norm(x + sa, (D,), state[f"{L}.norm1.weight"], state[f"{L}.norm1.bias"], 1e-5)
        hdn = x @ state[f"{L}.linear1.weight"].T + state[f"{L}.linear1.bias"]
        act = 0.5 * hdn * (1.0 + torch.erf(hdn / (2.0 ** 0.5)))
        act = _drop(act.reshape(B * T, -1), p, train, masks).reshape(B, T, -1)
        ff = act @ state[f"{L}.linear2.weight"].T + state[f"{L}.linear2.bias"]
        ff = _drop(ff.reshape(B * T, D), p, train, masks).reshape(B, T, D)
        x = F.layer_norm(x + ff, (D,), state[f"{L}.norm2.weight"], state[f"{L}.norm2.bias"], 1e-5)
    return x


def jdcnet_forward(state: Dict[str, torch.Tensor], x: torch.Tensor, seq_cfg: dict, train: bool = False,
                   masks=None, new_stats: Optional[dict] = None, taps: Optional[dict] = None,
                   fused_lstm: bool = False):
    """x (B,1,T,80) -> (classifier (B,T,num_class), detector (B,T)).

    ``masks``: iterator of keep-masks consumed in the order pool_block, detector_conv, then per
    LSTM layer (classifier, detector) -- the order the HIP path draws them; None with train=True
    requires all dropout rates to be 0.  ``new_stats`` collects updated BN running statistics;
    ``taps`` collects the intermediate tensors of SURVEY 3.3.
    """
    T = x.shape[-2]
    p_blk = 0.5 if (train and masks is not None) else 0.0
    num_layers = seq_cfg.get("num_layers", 2)
    p_seq = seq_cfg.get("dropout", 0.3) if num_layers > 1 else 0.0
    if masks is None:
        p_seq = 0.0
    bidir = seq_cfg.get("bidirectional", True)

    h = _conv(x, state["conv_block.0.weight"])
    h = F.leaky_relu(_batchnorm(h, state, "conv_block.1", train, new_stats), SLOPE)
    convblock = _conv(h, state["conv_block.3.weight"])
    rb1 = _res_block(convblock, state, "res_block1", train, new_stats)
    rb2 = _res_block(rb1, state, "res_block2", train, new_stats)
    rb3 = _res_block(rb2, state, "res_block3", train, new_stats)
    pb = F.leaky_relu(_batchnorm(rb3, state, "pool_block.0", train, new_stats), SLOPE)
    pb = F.max_pool2d(pb, kernel_size=(1, 4))
    pb = _drop(pb.permute(0, 2, 3, 1), p_blk, train, masks).permute(0, 3, 1, 2)   # masks are channels-last

    cat = torch.cat([F.max_pool2d(convblock, (1, 40)), F.max_pool2d(rb1, (1, 20)),
                     F.max_pool2d(rb2, (1, 10)), pb], dim=1)
    det = _conv(cat, state["detector_conv.0.weight"])
    det = F.leaky_relu(_batchnorm(det, state, "detector_conv.1", train, new_stats), SLOPE)
    det = _drop(det.permute(0, 2, 3, 1), p_blk, train, masks).permute(0, 3, 1, 2)

    seq_c = pb.permute(0, 2, 1, 3).reshape(-1, T, 512)
    seq_d = det.permute(0, 2, 1, 3).reshape(-1, T, 512)
    if seq_cfg.get("model_type", "bilstm").lower() == "transformer":
        # the HIP path runs the classifier branch to the end, then the detector branch
        yc = transformer_head(seq_c, state, "sequence_classifier", seq_cfg, train, masks)
        yd = transformer_head(seq_d, state, "sequence_detector", seq_cfg, train, masks)
        cls = yc @ state["classifier.weight"].T + state["classifier.bias"]
        dlog = yd @ state["detector.weight"].T + state["detector.bias"]
        if taps is not None:
            taps.update(convblock_out=convblock, resblock1_out=rb1, resblock2_out=rb2, resblock3_out=rb3,
                        poolblock_out=pb, detector_feat=det, seq_classifier_out=yc, seq_detector_out=yd)
        return cls, dlog.sum(dim=-1)
    # the HIP path draws LSTM masks layer by layer for (classifier, detector); mirror that order
    if masks is not None and train and p_seq > 0:
        yc, yd = seq_c, seq_d
        for layer in range(num_layers):
            one = {k: v for k, v in state.items()}
            yc = _one_layer(yc, one, "sequence_classifier.model", layer, bidir)
            yd = _one_layer(yd, one, "sequence_detector.model", layer, bidir)
            if layer < num_layers - 1:
                yc = _drop(yc, p_seq, train, masks)
                yd = _drop(yd, p_seq, train, masks)
    elif fused_lstm:
        yc = bilstm_fused(seq_c, state, "sequence_classifier.model", num_layers, bidir)
        yd = bilstm_fused(seq_d, state, "sequence_detector.model", num_layers, bidir)
    else:
        yc = bilstm(seq_c, state, "sequence_classifier.model", num_layers, bidir, 0.0, train, None)
        yd = bilstm(seq_d, state, "sequence_detector.model", num_layers, bidir, 0.0, train, None)

    cls = yc @ state["classifier.weight"].T + state["classifier.bias"]
    dlog = yd @ state["detector.weight"].T + state["detector.bias"]
    if taps is not None:
        taps.update(convblock_out=convblock, resblock1_out=rb1, resblock2_out=rb2, resblock3_out=rb3,
                    poolblock_out=pb, detector_feat=det, seq_classifier_out=yc, seq_detector_out=yd)
    return cls, dlog.sum(dim=-1)


def _one_layer(x, state, prefix, layer, bidir):
    outs = []
    for d in range(2 if bidir else 1):
        sfx = f"_l{layer}" + ("_reverse" if d else "")
        outs.append(lstm_layer_direction(
            x, state[f"{prefix}.weight_ih{sfx}"], state[f"{prefix}.weight_hh{sfx}"],
            state[f"{prefix}.bias_ih{sfx}"], state[f"{prefix}.bias_hh{sfx}"], reverse=bool(d)))
    return torch.cat(outs, dim=-1)


def jdc_loss(f0_pred, sil_pred, f0, sil, lambda_f0=0.1):
    """trainer.py:237-239 with train.py:104-106: lambda*SmoothL1(f0_pred.squeeze(), f0) + BCEWithLogits."""
    d = f0_pred.squeeze(-1) - f0
    ad = d.abs()
    l1 = torch.where(ad < 1.0, 0.5 * d * d, ad - 0.5).mean()
    z = sil_pred
    bce = (z.clamp(min=0) - z * sil + torch.log1p(torch.exp(-z.abs()))).mean()
    loss_f0 = lambda_f0 * l1
    return loss_f0 + bce, loss_f0, bce


CREPE_CENTS0 = 1997.3794084376191


def f0_to_bins(f0_hz, n_bins: int = 360):
    """Build-defined 360-bin target (the reference has no classification loss): CREPE bins,
    ``cents = 1200 log2(f0/10)``, ``bin = clip(rint((cents - 1997.3794084376191)/20), 0, n_bins-1)`` for voiced
    frames (f0 > 0) and -1 (ignored) for unvoiced ones.  float64 / int64 numpy."""
    import numpy as np
    f = np.asarray(f0_hz, dtype=np.float64)
    bins = np.full(f.shape, -1, dtype=np.int64)
    v = f > 0
    cents = 1200.0 * np.log2(f[v] / 10.0)
    bins[v] = np.clip(np.rint((cents - CREPE_CENTS0) / 20.0), 0, n_bins - 1).astype(np.int64)
    return bins


def jdc_bins_loss(logits, sil_pred, f0, sil, lambda_f0=0.1):
    """lambda * CE(logits, f0_to_bins(f0)) averaged over voiced frames (0 if none) + BCEWithLogits(sil)."""
    bins = torch.from_numpy(f0_to_bins(f0.detach().cpu().numpy(), logits.shape[-1])).to(logits.device)
    flat, tgt = logits.reshape(-1, logits.shape[-1]), bins.reshape(-1)
    voiced = tgt >= 0
    if bool(voiced.any()):
        lse = torch.logsumexp(flat[voiced], dim=-1)
        ce = (lse - flat[voiced].gather(1, tgt[voiced, None]).squeeze(1)).mean()
    else:
        ce = flat.sum() * 0.0
    z = sil_pred
    bce = (z.clamp(min=0) - z * sil + torch.log1p(torch.exp(-z.abs()))).mean()
    loss_f0 = lambda_f0 * ce
    return loss_f0 + bce, loss_f0, bce


# --------------------------------------------------------------------------- deterministic weights
def seeded_state(seed: int, num_class: int = 1, hidden_size: int = 384, num_layers: int = 4,
                 bidirectional: bool = True, dtype=torch.float32, model_type: str = "bilstm", nhead: int = 8,
                 dim_feedforward: int = 1536, max_len: int = 2048) -> Dict[str, torch.Tensor]:
    """A full JDCNet(+BiLSTM) state_dict drawn from ``numpy.random.default_rng(seed)``: the same
    numbers whatever the torch version, so fixtures only have to store outputs."""
    import numpy as np
    rng = np.random.default_rng(seed)
    st: Dict[str, torch.Tensor] = {}

    def normal(shape, std):
        return torch.from_numpy((rng.standard_normal(shape) * std).astype(np.float32)).to(dtype)

    def conv(name, co, ci, k):
        st[name + ".weight"] = normal((co, ci, k, k), (2.0 / ((ci + co) * k * k)) ** 0.5)

    def bn(name, c):
        st[name + ".weight"] = torch.from_numpy(rng.uniform(0.5, 1.5, c).astype(np.float32)).to(dtype)
        st[name + ".bias"] = normal((c,), 0.1)
        st[name + ".running_mean"] = normal((c,), 0.1)
        st[name + ".running_var"] = torch.from_numpy(rng.uniform(0.5, 1.5, c).astype(np.float32)).to(dtype)
        st[name + ".num_batches_tracked"] = torch.tensor(0, dtype=torch.long)

    conv("conv_block.0", 64, 1, 3); bn("conv_block.1", 64); conv("conv_block.3", 64, 64, 3)
    for i, (ci, co) in enumerate([(64, 128), (128, 192), (192, 256)], start=1):
        bn(f"res_block{i}.pre_conv.0", ci)
        conv(f"res_block{i}.conv.0", co, ci, 3); bn(f"res_block{i}.conv.1", co); conv(f"res_block{i}.conv.3", co, co, 3)
        conv(f"res_block{i}.conv1by1", co, ci, 1)
    bn("pool_block.0", 256)
    conv("detector_conv.0", 256, 640, 1); bn("detector_conv.1", 256)
    nd = 2 if bidirectional else 1
    if model_type == "transformer":
        import math
        D = 512
        pe = torch.zeros(max_len, D)
        pos = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
        div = torch.exp(torch.arange(0, D, 2).float() * (-math.log(10000.0) / D))
        pe[:, 0::2] = torch.sin(pos * div)
        pe[:, 1::2] = torch.cos(pos * div)
        for branch in ("sequence_classifier", "sequence_detector"):
            st[f"{branch}.pos_encoding.pe"] = pe.unsqueeze(0).to(dtype)
            for i in range(num_layers):
                L = f"{branch}.model.layers.{i}"
                st[f"{L}.self_attn.in_proj_weight"] = normal((3 * D, D), D ** -0.5)
                st[f"{L}.self_attn.in_proj_bias"] = normal((3 * D,), 0.1)
                st[f"{L}.self_attn.out_proj.weight"] = normal((D, D), D ** -0.5)
                st[f"{L}.self_attn.out_proj.bias"] = normal((D,), 0.1)
                st[f"{L}.linear1.weight"] = normal((dim_feedforward, D), D ** -0.5)
                st[f"{L}.linear1.bias"] = normal((dim_feedforward,), 0.1)
                st[f"{L}.linear2.weight"] = normal((D, dim_feedforward), dim_feedforward ** -0.5)
                st[f"{L}.linear2.bias"] = normal((D,), 0.1)
                for nm in ("norm1", "norm2"):
                    st[f"{L}.{nm}.weight"] = torch.from_numpy(rng.uniform(0.5, 1.5, D).astype(np.float32)).to(dtype)
                    st[f"{L}.{nm}.bias"] = normal((D,), 0.1)
            st[f"{branch}.layer_norm.weight"] = torch.from_numpy(rng.uniform(0.5, 1.5, D).astype(np.float32)).to(dtype)
            st[f"{branch}.layer_norm.bias"] = normal((D,), 0.1)
        st["classifier.weight"] = normal((num_class, D), D ** -0.5)
        st["classifier.bias"] = normal((num_class,), 0.1)
        st["detector.weight"] = normal((2, D), D ** -0.5)
        st["detector.bias"] = normal((2,), 0.1)
        return st
    for branch in ("sequence_classifier", "sequence_detector"):
        for layer in range(num_layers):
            in_sz = 512 if layer == 0 else hidden_size * nd
            for d in range(nd):
                sfx = f"_l{layer}" + ("_reverse" if d else "")
                st[f"{branch}.model.weight_ih{sfx}"] = normal((4 * hidden_size, in_sz), in_sz ** -0.5)
                st[f"{branch}.model.weight_hh{sfx}"] = normal((4 * hidden_size, hidden_size), hidden_size ** -0.5)
                st[f"{branch}.model.bias_ih{sfx}"] = normal((4 * hidden_size,), 0.1)
                st[f"{branch}.model.bias_hh{sfx}"] = normal((4 * hidden_size,), 0.1)
    D = hidden_size * nd
    st["classifier.weight"] = normal((num_class, D), D ** -0.5)
    st["classifier.bias"] = normal((num_class,), 0.1)
    st["detector.weight"] = normal((2, D), D ** -0.5)
    st["detector.bias"] = normal((2,), 0.1)
    return st
