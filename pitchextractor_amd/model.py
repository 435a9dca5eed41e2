"""JDCNet on hand-written HIP kernels: drop-in for the reference ``model.JDCNet``.

Same constructor, same ``forward(x: (B,1,T,80)) -> ((B,T,num_class), (B,T))`` and the same
``state_dict`` keys/shapes as the reference (model.py:13-122, 143-175, 196-256), so
checkpoints move both ways.  What differs is everything underneath:

* all parameters live in ONE flat fp32 buffer (fused AdamW, one gradient buffer for the
  data-parallel all-reduce); the ``nn.Parameter`` objects are views into it;
* activations are channels-last ``[B, T, F, C]``; the 3x3 convolutions are halo-staged implicit GEMMs
  whose fp32 products run as an exact three-term bf16 split on the bf16 MFMA pipe (``ops.FP32_MATMUL``:
  "x3", default) or on ``v_mfma_f32_32x32x2_f32`` ("native"); BatchNorm statistics / LeakyReLU /
  max-pool / dropout are fused HBM passes; the BiLSTM recurrence is ONE persistent launch per layer for
  all 4 (direction x branch) cells with W_hh resident on chip (one launch per time step only as the
  fallback for shapes / devices the persistent kernels do not cover);
* the whole network is a single ``autograd.Function`` with a hand-written backward that
  writes parameter gradients straight into the flat gradient buffer.

There is no eager/CPU path: running the module needs the HIP library and device tensors.
"""
from __future__ import annotations

import math
import os

import torch
from torch import nn

from . import ops

_ALIGN = 4  # parameters start on 16-byte boundaries inside the flat buffer


# --------------------------------------------------------------------------- parameter holders
class _Conv(nn.Module):
    """Bias-free Conv2d weight in the reference's OIHW layout."""

    def __init__(self, cin, cout, k):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k))


class _BatchNorm(nn.Module):
    def __init__(self, c, eps=1e-5, momentum=0.1):
        super().__init__()
        self.eps, self.momentum = eps, momentum
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))


class _Slots(nn.Module):
    """Children addressed by position, so keys read ``block.0.weight`` like the reference's Sequential."""

    def __init__(self, **children):
        super().__init__()
        for name, mod in children.items():
            self.add_module(name.lstrip("_"), mod)

    def __getitem__(self, idx):
        return self._modules[str(idx)]


class _ResBlock(nn.Module):
    """Parameters of model.py:143-175 (pre_conv BN; conv, BN, conv; 1x1 projection)."""

    def __init__(self, cin, cout):
        super().__init__()
        self.cin, self.cout = cin, cout
        self.pre_conv = _Slots(_0=_BatchNorm(cin))
        self.conv = _Slots(_0=_Conv(cin, cout, 3), _1=_BatchNorm(cout), _3=_Conv(cout, cout, 3))
        self.conv1by1 = _Conv(cin, cout, 1)


class _LSTMWeights(nn.Module):
    """Parameter set of ``nn.LSTM(input, hidden, num_layers, bidirectional)`` with torch's names and order."""

    def __init__(self, input_size, hidden_size, num_layers, bidirectional):
        super().__init__()
        self.input_size, self.hidden_size = input_size, hidden_size
        self.num_layers, self.num_dirs = num_layers, 2 if bidirectional else 1
        for layer in range(num_layers):
            in_sz = input_size if layer == 0 else hidden_size * self.num_dirs
            for d in range(self.num_dirs):
                sfx = f"_l{layer}" + ("_reverse" if d else "")
                self.register_parameter("weight_ih" + sfx, nn.Parameter(torch.empty(4 * hidden_size, in_sz)))
                self.register_parameter("weight_hh" + sfx, nn.Parameter(torch.empty(4 * hidden_size, hidden_size)))
                self.register_parameter("bias_ih" + sfx, nn.Parameter(torch.empty(4 * hidden_size)))
                self.register_parameter("bias_hh" + sfx, nn.Parameter(torch.empty(4 * hidden_size)))

    def cell(self, layer, d):
        sfx = f"_l{layer}" + ("_reverse" if d else "")
        return (getattr(self, "weight_ih" + sfx), getattr(self, "weight_hh" + sfx),
                getattr(self, "bias_ih" + sfx), getattr(self, "bias_hh" + sfx))


class SinusoidalPositionalEncoding(nn.Module):
    """Buffer-only module: ``pe`` (1, max_len, d_model) exactly as model.py:181-190 builds it (it is
    part of the checkpoint).  The add is fused into the first LayerNorm kernel."""

    def __init__(self, d_model, max_len=2000):
        super().__init__()
        pe = torch.zeros(max_len, d_model)
        position = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model))
        pe[:, 0::2] = torch.sin(position * div_term)
        pe[:, 1::2] = torch.cos(position * div_term)
        self.register_buffer("pe", pe.unsqueeze(0))


class _LayerNorm(nn.Module):
    def __init__(self, d, eps=1e-5):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(d))
        self.bias = nn.Parameter(torch.zeros(d))


class _SelfAttention(nn.Module):
    def __init__(self, d, nhead):
        super().__init__()
        self.embed_dim, self.num_heads = d, nhead
        self.in_proj_weight = nn.Parameter(torch.empty(3 * d, d))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * d))
        self.out_proj = _Linear(d, d)


class _EncoderLayer(nn.Module):
    """Parameters of nn.TransformerEncoderLayer(d, nhead, ff, dropout, batch_first, activation='gelu'),
    post-norm (norm_first=False), in torch's registration order."""

    def __init__(self, d, nhead, ff, dropout):
        super().__init__()
        self.p = dropout
        self.self_attn = _SelfAttention(d, nhead)
        self.linear1 = _Linear(d, ff)
        self.linear2 = _Linear(ff, d)
        self.norm1 = _LayerNorm(d)
        self.norm2 = _LayerNorm(d)


class _Encoder(nn.Module):
    def __init__(self, d, nhead, ff, dropout, num_layers):
        super().__init__()
        self.layers = nn.ModuleList([_EncoderLayer(d, nhead, ff, dropout) for _ in range(num_layers)])


class SequenceModel(nn.Module):
    """Temporal block of model.py:196-256.  ``bilstm`` runs on the HIP recurrence kernels."""

    def __init__(self, input_size, model_type="bilstm", hidden_size=384, num_layers=2, dropout=0.3,
                 bidirectional=True, nhead=8, dim_feedforward=1024, max_len=2000):
        super().__init__()
        self.model_type = model_type.lower()
        self.input_size, self.hidden_size = input_size, hidden_size
        self.bidirectional, self.num_layers = bidirectional, num_layers
        if self.model_type == "bilstm":
            self.dropout = dropout if num_layers > 1 else 0.0
            self.model = _LSTMWeights(input_size, hidden_size, num_layers, bidirectional)
            self._output_dim = hidden_size * (2 if bidirectional else 1)
        elif self.model_type == "transformer":
            if input_size not in (256, 512, 768, 1024) or input_size % nhead or (input_size // nhead) % 4:
                raise NotImplementedError("transformer head: d_model must be 256/512/768/1024 and divide by nhead")
            self.dropout = dropout
            self.nhead, self.dim_feedforward = nhead, dim_feedforward
            self.pos_encoding = SinusoidalPositionalEncoding(input_size, max_len=max_len)
            self.model = _Encoder(input_size, nhead, dim_feedforward, dropout, num_layers)
            self.layer_norm = _LayerNorm(input_size)
            self._output_dim = input_size
        else:
            raise ValueError(f"Unsupported sequence model type: {model_type}")

    @property
    def output_dim(self):
        return self._output_dim


class _Linear(nn.Module):
    def __init__(self, d_in, d_out):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(d_out, d_in))
        self.bias = nn.Parameter(torch.zeros(d_out))


# --------------------------------------------------------------------------- the network function
class _Ctx:
    """Tensors one forward pass leaves behind for its backward."""


def _bn(mod: _BatchNorm, x, train: bool, partials=None):
    """``partials``: column sums the convolution that produced x left behind (no pass over x for the statistics)."""
    if train:
        st = ops.bn_train_stats(x, mod.weight, mod.bias, mod.running_mean, mod.running_var, mod.eps, mod.momentum,
                                partials=partials)
        mod.num_batches_tracked += 1
        return st
    return ops.bn_eval_affine(mod.weight, mod.bias, mod.running_mean, mod.running_var, mod.eps)


def _flat2(t):
    return t.view(-1, t.shape[-1])


def _no_amax(_param):
    return None


def _res_forward(blk: _ResBlock, x, pool, train, need_grad, slope, x_stats=None, wa=_no_amax):
    """x [B,T,F,Cin] -> [B,T,F/pool,Cout]; returns (out, saved, BatchNorm partials of out or None)."""
    s = _Ctx()
    s.x = x
    s.bn_pre = _bn(blk.pre_conv[0], x, train, x_stats)
    s.am_p = ops.amax_word()            # "h2" products: one absmax word per operand tensor, left by its producer
    s.p = ops.bn_act_pool_fwd(x, s.bn_pre, pool=pool, slope=slope, amax_out=s.am_p)
    out = ops.gemm_nt(_flat2(s.p), blk.conv1by1.weight.view(blk.cout, blk.cin), amax_a=s.am_p,
                      amax_b=wa(blk.conv1by1.weight))
    out = out.view(*s.p.shape[:3], blk.cout)
    wf0, s.wd0 = ops.conv3x3_repack(blk.conv[0].weight, True, need_grad, amax=wa(blk.conv[0].weight))
    s.c, c_stats = ops.conv3x3_fwd(s.p, wf0, bn_stats=train, amax=s.am_p)
    s.bn_mid = _bn(blk.conv[1], s.c, train, c_stats)
    s.am_a = ops.amax_word()
    s.a = ops.bn_act_pool_fwd(s.c, s.bn_mid, pool=1, slope=slope, amax_out=s.am_a)
    wf3, s.wd3 = ops.conv3x3_repack(blk.conv[3].weight, True, need_grad, amax=wa(blk.conv[3].weight))
    _, out_stats = ops.conv3x3_fwd(s.a, wf3, out=out, accumulate=True, bn_stats=train,   # conv(x) + conv1by1(x), model.py:171-172
                                   amax=s.am_a)
    return out, s, out_stats


def _res_backward(blk: _ResBlock, s, d_out, pool, slope, grads, side, am_do=None, wa=_no_amax):
    """d_out: grad of the block output (am_do: its absmax word if the producer left one).  Returns (grad wrt the block
    input (dense, overwritten), its absmax word).  The three weight gradients go through `side` (_SideWork)."""
    if am_do is None:
        am_do = ops.amax_for(d_out)
    side.run(lambda: (ops.conv3x3_wgrad(s.a, d_out, grads[blk.conv[3].weight], amax_x=s.am_a, amax_dy=am_do),
                      ops.gemm_tn(_flat2(d_out), _flat2(s.p), out=grads[blk.conv1by1.weight].view(blk.cout, blk.cin),
                                  amax_a=am_do, amax_b=s.am_p)),
             d_out, am_do, on=OVERLAP_CONV_WGRAD)
    with ops.timer_tag("dgrad"):
        d_a = ops.conv3x3_fwd(d_out, s.wd3, amax=am_do)
    am_dc = ops.amax_word()
    d_c = ops.bn_act_pool_bwd(s.c, d_a, s.bn_mid, grads[blk.conv[1].weight], grads[blk.conv[1].bias], pool=1,
                              slope=slope, dx=d_a, amax_out=am_dc)
    side.run(lambda: ops.conv3x3_wgrad(s.p, d_c, grads[blk.conv[0].weight], amax_x=s.am_p, amax_dy=am_dc), d_c, am_dc,
             on=OVERLAP_CONV_WGRAD)
    with ops.timer_tag("dgrad"):
        d_p = ops.conv3x3_fwd(d_c, s.wd0, amax=am_dc)
    w1t = ops.transpose2d(blk.conv1by1.weight.view(blk.cout, blk.cin))
    ops.gemm_nt(_flat2(d_out), w1t, out=_flat2(d_p), accumulate=True, amax_a=am_do, amax_b=wa(blk.conv1by1.weight))
    am_dx = ops.amax_word()
    return ops.bn_act_pool_bwd(s.x, d_p, s.bn_pre, grads[blk.pre_conv[0].weight], grads[blk.pre_conv[0].bias],
                               pool=pool, slope=slope, amax_out=am_dx), am_dx


class _DropoutCfg:
    def __init__(self, seed=0):
        self.seed = int(seed)
        self.offset = 0
        self.inject = None          # optional iterator of uint8 masks replayed instead of Philox (parity tests)

    def next_offset(self, n_quads):
        off = self.offset
        self.offset += int(n_quads)
        return off


def _dropout(cfg: _DropoutCfg, x2d, p, out2d=None):
    """Returns (y, mask).  p == 0 -> identity (mask None)."""
    if p <= 0.0:
        if out2d is not None:
            ops.copy2d(x2d, out2d)
            return out2d, None
        return x2d, None
    mask_in = next(cfg.inject) if cfg.inject is not None else None
    quads = x2d.shape[0] * (x2d.shape[1] // 4)
    return ops.dropout(x2d, p, out2d=out2d, mask_in=mask_in, seed=cfg.seed, offset=cfg.next_offset(quads))


def _dropout_bwd(dy2d, p, mask, out2d=None):
    if mask is None:
        if out2d is not None:
            ops.copy2d(dy2d, out2d)
            return out2d
        return dy2d
    out, _ = ops.dropout(dy2d, p, out2d=out2d, mask_in=mask)
    return out


def _lstm_forward(models, xs, train, need_grad, drop: _DropoutCfg, wa=_no_amax):
    """models: list of SequenceModel (identical shapes); xs: list of [B,T,in].  One launch per time step
    advances every (model, direction) cell of a layer together."""
    m0 = models[0].model
    H, L, ND = m0.hidden_size, m0.num_layers, m0.num_dirs
    B, T = xs[0].shape[:2]
    saved = []
    cur = list(xs)
    for layer in range(L):
        lay = _Ctx()
        lay.x = cur
        lay.gates, lay.cbuf, lay.y, lay.mask = [], [], [], []
        ys = [torch.empty((B, T, ND * H), dtype=torch.float32, device=cur[0].device) for _ in models]
        whh, gts, ysl, cbs, rev = [], [], [], [], []
        # "h2" products: layer 0 reads the conv stack's features (one absmax pass each); deeper layers read LSTM
        # outputs, |h| < 1, scaled by the inter-layer dropout's 1 / (1 - p): a bound serves as the scale source
        lay.am_x = [ops.amax_for(cur[mi]) if layer == 0 else
                    ops.amax_bound(1.0 / (1.0 - (sm.dropout if train else 0.0)), cur[mi].device)
                    for mi, sm in enumerate(models)]
        for mi, sm in enumerate(models):
            for d in range(ND):
                w_ih, w_hh, b_ih, b_hh = sm.model.cell(layer, d)
                g = ops.gemm_nt(_flat2(cur[mi]), w_ih, bias0=b_ih, bias1=b_hh, amax_a=lay.am_x[mi],
                                amax_b=wa(w_ih)).view(B, T, 4 * H)
                cb = torch.empty((B, T, H), dtype=torch.float32, device=g.device)
                whh.append(w_hh); gts.append(g); cbs.append(cb); rev.append(d)
                ysl.append(ys[mi][:, :, d * H:(d + 1) * H])
        ops.lstm_fwd(whh, gts, ysl, cbs, rev, B, T, H)
        lay.gates, lay.cbuf, lay.y = gts, cbs, ys
        nxt = []
        for mi, sm in enumerate(models):
            p = sm.dropout if (train and layer < L - 1) else 0.0
            yd, mask = _dropout(drop, _flat2(ys[mi]), p)
            lay.mask.append((p, mask))
            nxt.append(yd.view(B, T, ND * H))
        cur = nxt
        saved.append(lay if need_grad else None)
    return cur, saved


OVERLAP_LSTM_WGRAD = os.environ.get("PE_OVERLAP_LSTM_WGRAD", "1") != "0"
_SIDE_STREAMS: dict = {}


def _side_stream(dev):
    """The side stream of `dev`, created at the LOWEST HIP priority (torch can only make streams of normal or high
    priority, and the main work usually runs on the default stream = normal): its weight-gradient kernels then fill
    what the critical data-gradient chain leaves free instead of competing with it (-1 % step time, measured)."""
    key = torch.device(dev)
    if key not in _SIDE_STREAMS:
        stream = None
        if os.environ.get("PE_SIDE_STREAM_PRIORITY", "low") == "low":
            import ctypes
            from . import _lib
            handle = ctypes.c_void_p()
            with torch.cuda.device(key):         # created by the HIP library's own runtime (include/pitchextractor_hip.h)
                _lib.check(_lib.load().pe_stream_create_low_priority(ctypes.byref(handle)), "pe_stream_create_low_priority")
            stream = torch.cuda.ExternalStream(handle.value, device=key)
        _SIDE_STREAMS[key] = stream if stream is not None else torch.cuda.Stream(device=key)
    return _SIDE_STREAMS[key]


class _SideWork:
    """Weight-gradient kernels need nothing downstream in the same backward pass, and they are MFMA-bound where the
    data-gradient chain between them is largely HBM-bound (BatchNorm / pooling backward): launched on a side stream
    they fill what the main chain leaves idle.  `run(fn, *keep)` orders fn behind everything already queued on the
    main stream and keeps the tensors it reads alive (the allocator recycles a freed block for the stream it was
    allocated on without waiting for other streams); `join()` makes the main stream wait for all of it."""

    def __init__(self, dev, enabled=True):
        self.dev, self.enabled, self.keep, self.used = dev, enabled, [], False

    def pending_stream(self):
        """The side stream if anything is queued on it since the last join (a reducer must wait for it), else None."""
        return _side_stream(self.dev) if self.used else None

    def run(self, fn, *keep, on=True):
        if not (self.enabled and on):
            fn()
            return
        side = _side_stream(self.dev)
        side.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(side):
            fn()
        self.keep.append(keep)
        self.used = True

    def join(self):
        if self.used:
            torch.cuda.current_stream(self.dev).wait_stream(_side_stream(self.dev))
        self.keep.clear()
        self.used = False


OVERLAP_CONV_WGRAD = os.environ.get("PE_OVERLAP_CONV_WGRAD", "1") != "0"
OVERLAP_TF_WGRAD = os.environ.get("PE_OVERLAP_TF_WGRAD", "1") != "0"
# the LSTM weight gradients are joined right after the LSTM backward: letting them float under the conv backward
# ("late") only delays the conv weight gradients queued behind them on the same side stream (+1.1 ms, measured)
LSTM_WGRAD_JOIN_EARLY = os.environ.get("PE_LSTM_WGRAD_JOIN", "early") == "early"


def _lstm_backward(models, saved, dys, grads, side, wa=_no_amax):
    """dys: list of [B,T,ND*H] grads of the top layer outputs.  Returns grads of the inputs.  Weight / bias gradients
    go through `side` (_SideWork) and are NOT joined here."""
    m0 = models[0].model
    H, L, ND = m0.hidden_size, m0.num_layers, m0.num_dirs
    B, T = dys[0].shape[:2]
    dev = dys[0].device
    for layer in reversed(range(L)):
        lay = saved[layer]
        dys = [_dropout_bwd(_flat2(dys[mi]), lay.mask[mi][0], lay.mask[mi][1]).view(B, T, ND * H)
               for mi in range(len(models))]
        whh_t, dsl, dcs, rev = [], [], [], []
        for mi, sm in enumerate(models):
            for d in range(ND):
                _, w_hh, _, _ = sm.model.cell(layer, d)
                whh_t.append(ops.transpose2d(w_hh))
                dsl.append(dys[mi][:, :, d * H:(d + 1) * H])
                dcs.append(torch.empty((B, H), dtype=torch.float32, device=dev))
                rev.append(d)
        # bias gradients come out of the recurrence kernel as per-batch-tile rows where it supports that
        nrows = ops.lstm_bwd_dbias_rows(len(whh_t), B, T, H, dsl[0].stride(1), dev)
        brows = [torch.empty((nrows, 4 * H), dtype=torch.float32, device=dev) for _ in whh_t] if nrows else None
        # "h2": the scale source of each cell's gate gradients comes out of the recurrence kernel where it can emit it
        am_dg = [ops.amax_word() for _ in whh_t] if (nrows and ops.h2_active()) else None
        have_db = ops.lstm_bwd(whh_t, lay.gates, lay.cbuf, dsl, dcs, rev, B, T, H,     # gates now hold d(pre-activations)
                               dbias_rows=brows, amax_out=am_dg)
        if am_dg is None:                                               # (else one pass per tensor; None outside "h2")
            am_dg = [ops.amax_for(gt) for gt in lay.gates]
        # the data gradients first (the next layer's recurrence waits for them) ...
        dxs = []
        for mi, sm in enumerate(models):
            dx = torch.empty_like(lay.x[mi])
            for d in range(ND):
                w_ih = sm.model.cell(layer, d)[0]
                ops.gemm_nt(_flat2(lay.gates[mi * ND + d]), ops.transpose2d(w_ih), out=_flat2(dx), accumulate=(d > 0),
                            amax_a=am_dg[mi * ND + d], amax_b=wa(w_ih))
            dxs.append(dx)

        # ... the weight / bias gradients need nothing downstream: on the (low-priority) side stream they fill the
        # CUs the next layer's persistent recurrence leaves idle (192 of 256) and then the gaps of the conv backward
        def weight_grads(layer=layer, lay=lay, brows=brows, have_db=have_db, am_dg=am_dg):
            for mi, sm in enumerate(models):
                x2 = _flat2(lay.x[mi])
                for d in range(ND):
                    w_ih, w_hh, b_ih, b_hh = sm.model.cell(layer, d)
                    dg = lay.gates[mi * ND + d]
                    dg2 = _flat2(dg)
                    ops.gemm_tn(dg2, x2, out=grads[w_ih], amax_a=am_dg[mi * ND + d], amax_b=lay.am_x[mi])
                    ops.lstm_whh_grad(dg, lay.y[mi][:, :, d * H:(d + 1) * H], grads[w_hh], d, B, T, H,
                                      amax_dg=am_dg[mi * ND + d])
                    ops.colsum(brows[mi * ND + d] if have_db else dg2, grads[b_ih], grads[b_hh])

        side.run(weight_grads, brows, am_dg, on=OVERLAP_LSTM_WGRAD)   # (kept alive until the side stream has read them)
        dys = dxs
    return dys


# Encoder-layer dropouts ride in the pass next to them (LayerNorm forward / backward, GELU forward / backward; the
# residual-branch gradient is added inside the LayerNorm backward too): same masks, same values bit for bit as the
# separate passes (tested), 1.8 GB less HBM traffic per layer and branch at B = 256.  False = the separate passes.
TF_FUSE_DROPOUT = os.environ.get("PE_TF_FUSE_DROPOUT", "1") != "0"


def _mask_or_offset(drop: _DropoutCfg, like):
    """The (injected mask, Philox offset) pair `_dropout` would use for a dense tensor shaped like `like`."""
    mask_in = next(drop.inject) if drop.inject is not None else None
    return mask_in, drop.next_offset(like.numel() // 4)


def _residual_norm(drop: _DropoutCfg, x2d, sub2d, norm, p):
    """LN(x + dropout(sub)) -> (y, LnState, mask)."""
    if p > 0.0 and TF_FUSE_DROPOUT:
        mask_in, off = _mask_or_offset(drop, sub2d)
        return ops.layernorm_dropout_fwd(x2d, sub2d, norm.weight, norm.bias, p, mask_in=mask_in, seed=drop.seed,
                                         offset=off, eps=norm.eps)
    sub2d, mask = _dropout(drop, sub2d, p)
    y, st = ops.layernorm_fwd(x2d, norm.weight, norm.bias, b2d=sub2d, eps=norm.eps)
    return y, st, mask


def _residual_norm_bwd(dy, dy_add, st, norm, g, p, mask):
    """Backward of `_residual_norm` for dy (+ dy_add): (gradient of the sum, gradient of the sub-layer output)."""
    if TF_FUSE_DROPOUT:
        if mask is None:
            dsum = ops.layernorm_bwd(dy, st, norm.weight, g[norm.weight], g[norm.bias], dy_add=dy_add)
            return dsum, dsum
        return ops.layernorm_bwd(dy, st, norm.weight, g[norm.weight], g[norm.bias], dy_add=dy_add, drop_mask=mask, p=p)
    if dy_add is not None:
        ops.copy2d(dy_add, dy, accumulate=True)
    dsum = ops.layernorm_bwd(dy, st, norm.weight, g[norm.weight], g[norm.bias])
    return dsum, _dropout_bwd(dsum, p, mask)


def _tf_forward(sm, x, train, need_grad, drop: _DropoutCfg):
    """SequenceModel(transformer).forward (model.py:253-255) for one branch.  x [B,T,D] -> [B,T,D]."""
    B, T, D = x.shape
    H = sm.nhead
    dh = D // H
    R = B * T
    p = sm.dropout if train else 0.0
    scale = 1.0 / math.sqrt(dh)
    saved = _Ctx()
    saved.layers = []
    pe = sm.pos_encoding.pe[0, :T]
    y, saved.ln0 = ops.layernorm_fwd(x.reshape(R, D), sm.layer_norm.weight, sm.layer_norm.bias, pe=pe.contiguous(),
                                     eps=sm.layer_norm.eps)
    for lyr in sm.model.layers:
        c = _Ctx()
        att = lyr.self_attn
        c.x = y
        c.qkv = ops.gemm_nt(y, att.in_proj_weight, bias0=att.in_proj_bias)               # [R, 3D]
        c.fused = ops.attn_supported(T, dh)
        if c.fused:                      # one workgroup per (batch, head); scores never reach HBM
            mask_in = next(drop.inject) if (p > 0.0 and drop.inject is not None) else None
            off = drop.next_offset(B * H * T * (T // 4)) if (p > 0.0 and mask_in is None) else 0
            c.o, c.lse, c.mask_p = ops.attn_fwd(c.qkv, B, T, H, scale, p, mask_in=mask_in, seed=drop.seed, offset=off)
        else:
            qv, kv, vv = c.qkv[:, 0:D], c.qkv[:, D:2 * D], c.qkv[:, 2 * D:3 * D]
            hview = (3 * D, T * 3 * D, dh)                                                 # (ld, per batch, per head)
            c.P = torch.empty((B * H * T, T), dtype=torch.float32, device=x.device)
            pview = (T, H * T * T, T * T)
            ops.bgemm(0, qv, hview, kv, hview, c.P, pview, H, B * H, T, T, dh)
            ops.softmax_fwd_(c.P, scale)
            c.Pd, c.mask_p = _dropout(drop, c.P, p)
            c.o = torch.empty((R, D), dtype=torch.float32, device=x.device)
            ops.bgemm(1, c.Pd, pview, vv, hview, c.o, (D, T * D, dh), H, B * H, T, dh, T)
        sa = ops.gemm_nt(c.o, att.out_proj.weight, bias0=att.out_proj.bias)
        x1, c.ln1, c.mask1 = _residual_norm(drop, y, sa, lyr.norm1, p)
        c.x1 = x1
        c.h = ops.gemm_nt(x1, lyr.linear1.weight, bias0=lyr.linear1.bias)                  # [R, FF]
        if p > 0.0 and TF_FUSE_DROPOUT:
            mask_in, off = _mask_or_offset(drop, c.h)
            c.a, c.mask_f = ops.gelu_dropout_fwd(c.h, p, mask_in=mask_in, seed=drop.seed, offset=off)
        else:
            c.a, c.mask_f = _dropout(drop, ops.gelu_fwd(c.h), p)
        ff = ops.gemm_nt(c.a, lyr.linear2.weight, bias0=lyr.linear2.bias)
        y, c.ln2, c.mask2 = _residual_norm(drop, x1, ff, lyr.norm2, p)
        saved.layers.append(c)
    saved.p, saved.shape = p, (B, T, D)
    return y.view(B, T, D), (saved if need_grad else None)


def _tf_backward(sm, saved, dy, g, side):
    """Backward of one Transformer branch; the weight / bias gradients go through `side` (_SideWork)."""
    B, T, D = saved.shape
    H = sm.nhead
    dh = D // H
    R = B * T
    p = saved.p
    scale = 1.0 / math.sqrt(dh)
    dy = dy.reshape(R, D)
    dres = None                          # residual-branch gradient still to be added to dy
    hview = (3 * D, T * 3 * D, dh)
    pview = (T, H * T * T, T * T)
    for lyr, c in zip(reversed(list(sm.model.layers)), reversed(saved.layers)):
        att = lyr.self_attn
        # y = LN2(x1 + dropout2(ff))
        dsum, dff = _residual_norm_bwd(dy, dres, c.ln2, lyr.norm2, g, p, c.mask2)
        side.run(lambda: (ops.gemm_tn(dff, c.a, out=g[lyr.linear2.weight]), ops.colsum(dff, g[lyr.linear2.bias])),
                 dff, on=OVERLAP_TF_WGRAD)
        da = ops.gemm_nt(dff, ops.transpose2d(lyr.linear2.weight))
        if c.mask_f is not None and TF_FUSE_DROPOUT:
            dh_ = ops.gelu_dropout_bwd(c.h, da, c.mask_f, p, out=da)
        else:
            da = _dropout_bwd(da, p, c.mask_f)
            dh_ = ops.gelu_bwd(c.h, da, out=da)
        side.run(lambda: (ops.gemm_tn(dh_, c.x1, out=g[lyr.linear1.weight]), ops.colsum(dh_, g[lyr.linear1.bias])),
                 dh_, on=OVERLAP_TF_WGRAD)
        dx1 = ops.gemm_nt(dh_, ops.transpose2d(lyr.linear1.weight))
        # x1 = LN1(x + dropout1(sa)); dsum is the residual branch's share of d(x1)
        dsum1, dsa = _residual_norm_bwd(dx1, dsum, c.ln1, lyr.norm1, g, p, c.mask1)
        side.run(lambda: (ops.gemm_tn(dsa, c.o, out=g[att.out_proj.weight]), ops.colsum(dsa, g[att.out_proj.bias])),
                 dsa, on=OVERLAP_TF_WGRAD)
        do = ops.gemm_nt(dsa, ops.transpose2d(att.out_proj.weight))                        # [R, D] merged heads
        if c.fused:
            dqkv = ops.attn_bwd(c.qkv, c.o, do, c.lse, c.mask_p, B, T, H, scale, p if c.mask_p is not None else 0.0)
        else:
            qv, kv, vv = c.qkv[:, 0:D], c.qkv[:, D:2 * D], c.qkv[:, 2 * D:3 * D]
            dqkv = torch.empty_like(c.qkv)
            dq, dk, dv = dqkv[:, 0:D], dqkv[:, D:2 * D], dqkv[:, 2 * D:3 * D]
            oview = (D, T * D, dh)
            ops.bgemm(2, c.Pd, pview, do, oview, dv, hview, H, B * H, T, dh, T)            # dV = Pd^T dO
            dP = torch.empty_like(c.P)
            ops.bgemm(0, do, oview, vv, hview, dP, pview, H, B * H, T, T, dh)              # dPd = dO V^T
            dP = _dropout_bwd(dP, p, c.mask_p)
            ops.softmax_bwd_(c.P, dP, scale)                                               # dP <- dS (incl. 1/sqrt(dh))
            ops.bgemm(1, dP, pview, kv, hview, dq, hview, H, B * H, T, dh, T)              # dQ = dS K
            ops.bgemm(2, dP, pview, qv, hview, dk, hview, H, B * H, T, dh, T)              # dK = dS^T Q
        side.run(lambda: (ops.gemm_tn(dqkv, c.x, out=g[att.in_proj_weight]), ops.colsum(dqkv, g[att.in_proj_bias])),
                 dqkv, on=OVERLAP_TF_WGRAD)
        dy = ops.gemm_nt(dqkv, ops.transpose2d(att.in_proj_weight))
        dres = dsum1
    if TF_FUSE_DROPOUT:
        dx = ops.layernorm_bwd(dy, saved.ln0, sm.layer_norm.weight, g[sm.layer_norm.weight], g[sm.layer_norm.bias],
                               dy_add=dres)
    else:
        if dres is not None:
            ops.copy2d(dres, dy, accumulate=True)
        dx = ops.layernorm_bwd(dy, saved.ln0, sm.layer_norm.weight, g[sm.layer_norm.weight], g[sm.layer_norm.bias])
    return dx.view(B, T, D)


class _JDCFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, x, _anchor, need_grad):
        ctx.net, ctx.recompute = net, None
        if need_grad and net.checkpoint_forward:
            # Whole-model gradient checkpointing, the reference's granularity (trainer.py:226-233 wraps the
            # entire model in ONE checkpoint segment): this pass keeps nothing but the input and the dropout
            # stream position; backward re-runs the forward with bookkeeping, then differentiates it.
            drop = net.dropout_cfg
            if drop.inject is not None:
                raise NotImplementedError("gradient checkpointing cannot replay injected dropout masks")
            ctx.recompute = (x, drop.seed, drop.offset, net.training)
            out_cls, out_det, _ = net._forward_impl(x, False)
            ctx.saved = None
            return out_cls, out_det
        out_cls, out_det, saved = net._forward_impl(x, need_grad)
        ctx.saved = saved
        if net.keep_last_context:
            net.last_context = saved
        return out_cls, out_det

    @staticmethod
    def backward(ctx, d_cls, d_det):
        net = ctx.net
        if ctx.recompute is not None:
            x, seed, offset, training = ctx.recompute
            drop = net.dropout_cfg
            now = (drop.seed, drop.offset, net.training)
            drop.seed, drop.offset = seed, offset           # same Philox stream => same masks as the first pass
            net.train(training)
            try:
                # like torch.utils.checkpoint under the reference, the second forward runs in train mode again:
                # BatchNorm running statistics take a second momentum update with the same batch statistics
                _, _, ctx.saved = net._forward_impl(x, True)
            finally:
                drop.seed, drop.offset = now[0], max(now[1], drop.offset)
                net.train(now[2])
            ctx.recompute = None
        if ctx.saved is None:
            raise RuntimeError("JDCNet backward called on a forward that ran without gradient bookkeeping")
        net._backward_impl(ctx.saved, d_cls, d_det)
        ctx.saved = None
        return None, None, None, None


class JDCNet(nn.Module):
    """Joint Detection and Classification network (reference model.py:13-122) on HIP kernels."""

    def __init__(self, num_class=722, leaky_relu_slope=0.01, sequence_model_config=None):
        super().__init__()
        self.num_class = num_class
        self.leaky_relu_slope = leaky_relu_slope
        sequence_model_config = sequence_model_config if sequence_model_config is not None else {}

        self.conv_block = _Slots(_0=_Conv(1, 64, 3), _1=_BatchNorm(64), _3=_Conv(64, 64, 3))
        self.res_block1 = _ResBlock(64, 128)
        self.res_block2 = _ResBlock(128, 192)
        self.res_block3 = _ResBlock(192, 256)
        self.pool_block = _Slots(_0=_BatchNorm(256))
        self.detector_conv = _Slots(_0=_Conv(640, 256, 1), _1=_BatchNorm(256))

        sequence_model_config.setdefault("input_size", 512)          # model.py:59 (mutates the caller's dict)
        self.sequence_classifier = SequenceModel(**sequence_model_config)
        self.sequence_detector = SequenceModel(**sequence_model_config)
        self.classifier = _Linear(self.sequence_classifier.output_dim, num_class)
        self.detector = _Linear(self.sequence_detector.output_dim, 2)

        self.block_dropout = 0.5                                       # model.py:40,56
        self.dropout_cfg = _DropoutCfg()
        self.training_graph_wanted = True
        self._dp = None
        self._dp_cuts = None
        self.keep_last_context = False      # tests: expose the saved tensors (dropout masks) of the last forward
        self.checkpoint_forward = False     # Trainer(gradient_checkpointing=True): recompute the forward in backward
        self.last_context = None
        self._init_weights()
        self._flat = None
        self._grad_flat = None
        self._flatten()

    # ---- initialisation (model.py:124-140) -------------------------------------------------
    def _init_weights(self):
        for m in self.modules():
            if isinstance(m, _Linear):
                nn.init.kaiming_uniform_(m.weight)
                nn.init.constant_(m.bias, 0)
            elif isinstance(m, _Conv):
                nn.init.xavier_normal_(m.weight)
            elif isinstance(m, _SelfAttention):
                nn.init.xavier_uniform_(m.in_proj_weight)
                nn.init.constant_(m.in_proj_bias, 0)
            elif isinstance(m, _LSTMWeights):
                for p in m.parameters():
                    if p.dim() >= 2:
                        nn.init.orthogonal_(p.data)
                    else:
                        nn.init.normal_(p.data)

    # ---- flat parameter / gradient storage --------------------------------------------------
    def _flatten(self):
        params = list(self.parameters())
        if not params:
            return
        device = params[0].device
        offsets, total = [], 0
        for p in params:
            offsets.append(total)
            total += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        # one spare aligned slot behind the last parameter: in the GRADIENT buffer it carries the step's status word
        # (status_slot), so that it rides in the gradient all-reduce and reaches the fused AdamW on the device
        self._status_off = total
        flat = torch.zeros(total + _ALIGN, dtype=torch.float32, device=device)
        for p, off in zip(params, offsets):
            view = flat[off:off + p.numel()].view_as(p)
            view.copy_(p.data)
            p.data = view
        self._flat = flat
        self._flat.requires_grad_(True)
        self._grad_flat = None
        self._param_offsets = {id(p): off for p, off in zip(params, offsets)}
        self._param_list = params
        self._wseg = self._wamax = None

    def _apply(self, fn, recurse=True):
        out = super()._apply(fn, recurse)
        self._flatten()                    # .to(device) / .float() re-materialise parameters: re-pack them
        return out

    def attach_data_parallel(self, dp):
        """``dp``: pitchextractor_amd.distributed.GradientAllReduce over ``flat_gradients()``.  Backward then
        hands it gradient ranges as they become final, in backward order: [sequence models .. heads]
        (93 % of the bytes) right after the temporal-head backward, so that all-reduce runs on its own stream
        underneath the whole convolution backward; then [res_block3 .. detector_conv], res_block2 and res_block1
        as each block's backward finishes, and only [conv_block] (148 KB) after the last kernel."""
        self._dp = dp
        self._dp_cuts = self._block_cuts()

    def _block_cuts(self):
        """Flat-buffer offsets where conv_block | res_block1 | res_block2 | res_block3+pool_block+detector_conv |
        temporal heads begin, or None if the parameter order is not that (then the conv range goes out whole)."""
        def first(mod):
            return min(self._param_offsets[id(q)] for q in mod.parameters())

        def last(mod):
            return max(self._param_offsets[id(q)] for q in mod.parameters())
        blocks = [self.conv_block, self.res_block1, self.res_block2, self.res_block3]
        cuts = [first(b) for b in blocks] + [self._seq_offset()]
        tail_ok = all(cuts[3] < first(m) and last(m) < cuts[4] for m in (self.pool_block, self.detector_conv))
        ordered = cuts[0] == 0 and all(a < b for a, b in zip(cuts, cuts[1:])) and \
            all(last(b) < c for b, c in zip(blocks, cuts[1:]))
        return cuts if (ordered and tail_ok) else None

    def _seq_offset(self):
        first = next(self.sequence_classifier.parameters())
        return self._param_offsets[id(first)]

    @property
    def flat_parameters(self) -> torch.Tensor:
        return self._flat

    def flat_gradients(self) -> torch.Tensor:
        """The flat gradient buffer; ``p.grad`` of every parameter is a view into it."""
        if self._grad_flat is None or self._grad_flat.device != self._flat.device:
            self._grad_flat = torch.zeros_like(self._flat, requires_grad=False)
        return self._grad_flat

    def _refresh_weight_amax(self):
        """"h2" products: the absmax word of every parameter, one launch per forward over the flat buffer."""
        if not ops.h2_active() or self._flat is None or not self._flat.is_cuda:
            self._wamax = None
            return
        if getattr(self, "_wseg", None) is None or self._wseg[0].device != self._flat.device:
            offs = torch.tensor([self._param_offsets[id(p)] for p in self._param_list], dtype=torch.int64)
            lens = torch.tensor([p.numel() for p in self._param_list], dtype=torch.int64)
            self._wseg = (offs.to(self._flat.device), lens.to(self._flat.device))
            self._windex = {id(p): i for i, p in enumerate(self._param_list)}
        self._wamax = ops.absmax_segments(self._flat.detach(), self._wseg[0], self._wseg[1])

    def weight_amax(self, param):
        """1-element view of the absmax word of ``param`` (None outside "h2" mode)."""
        if getattr(self, "_wamax", None) is None:
            return None
        i = self._windex[id(param)]
        return self._wamax[i:i + 1]

    def status_slot(self) -> torch.Tensor:
        """1-element view of the flat gradient buffer behind the last parameter: 0 after a clean backward, non-zero
        when a persistent-LSTM hand-off timed out in this step (summed over ranks by the gradient all-reduce).  The
        fused AdamW skips its update on the device when it is set; the trainer reads it with the loss scalars."""
        return self.flat_gradients()[self._status_off:self._status_off + 1]

    def _write_status(self, dev):
        slot, word = self.status_slot(), ops.persistent_lstm_error_word(dev)
        if word is None:
            slot.zero_()
        else:
            slot.copy_(word != 0)

    def _grad_views(self):
        gflat = self.flat_gradients()
        views = {}
        for p in self._param_list:
            off = self._param_offsets[id(p)]
            g = gflat[off:off + p.numel()].view_as(p)
            if p.grad is None or p.grad.data_ptr() != g.data_ptr():
                p.grad = g
            views[p] = g
        return views

    # ---- forward / backward -----------------------------------------------------------------
    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("JDCNet (HIP) needs device tensors; there is no CPU fallback")
        if x.dim() != 4 or x.shape[1] != 1:
            raise ValueError("expected input of shape (B, 1, T, n_mels)")
        need_grad = torch.is_grad_enabled() and self.training_graph_wanted
        return _JDCFunction.apply(self, x, self._flat, need_grad)

    def _forward_impl(self, x, need_grad):
        train = self.training
        slope = self.leaky_relu_slope
        self._refresh_weight_amax()
        s = _Ctx()
        x_btf = x[:, 0].float()
        B, T, F = x_btf.shape
        s.x_btf = x_btf
        cbk = self.conv_block
        s.y0, y0_stats = ops.conv3x3_c1_fwd(x_btf, cbk[0].weight, bn_stats=train)
        s.bn0 = _bn(cbk[1], s.y0, train, y0_stats)
        s.am_a0 = ops.amax_word()
        s.a0 = ops.bn_act_pool_fwd(s.y0, s.bn0, pool=1, slope=slope, amax_out=s.am_a0)
        wf, s.wd_cb = ops.conv3x3_repack(cbk[3].weight, True, need_grad, amax=self.weight_amax(cbk[3].weight))
        s.cb, st_cb = ops.conv3x3_fwd(s.a0, wf, bn_stats=train, amax=s.am_a0)        # convblock_out
        wa = self.weight_amax
        s.rb1, s.r1, st1 = _res_forward(self.res_block1, s.cb, 2, train, need_grad, slope, st_cb, wa)
        s.rb2, s.r2, st2 = _res_forward(self.res_block2, s.rb1, 2, train, need_grad, slope, st1, wa)
        s.rb3, s.r3, st3 = _res_forward(self.res_block3, s.rb2, 2, train, need_grad, slope, st2, wa)

        # pool_block -> channels [384, 640) of the detector concat (model.py:36-41,90,108)
        Fp = s.rb3.shape[2] // 4
        if Fp != 2:
            raise ValueError("JDCNet geometry expects 80 mel bins (F/40 == 2)")
        s.bnp = _bn(self.pool_block[0], s.rb3, train, st3)
        pooled = ops.bn_act_pool_fwd(s.rb3, s.bnp, pool=4, slope=slope)            # [B,T,2,256]
        s.concat = torch.empty((B, T, 2, 640), dtype=s.rb3.dtype, device=x.device)     # (bf16 under ops.ACT_BF16)
        p_blk = self.block_dropout if train else 0.0
        cslice = s.concat.view(-1, 640)[:, 384:640]
        _, s.mask_pool = _dropout(self.dropout_cfg, _flat2(pooled), p_blk, out2d=cslice)
        seq_c = ops.nhwc_to_seq(s.concat, 256, coff=384)

        # (the positions of the maxima are kept: 25 MB of bytes save the backward pass three 1 GB reads)
        _, s.arg_cb = ops.maxpool_fwd(s.cb, 40, out=s.concat, coff=0, want_argmax=True)
        _, s.arg_rb1 = ops.maxpool_fwd(s.rb1, 20, out=s.concat, coff=64, want_argmax=True)
        _, s.arg_rb2 = ops.maxpool_fwd(s.rb2, 10, out=s.concat, coff=192, want_argmax=True)
        wdet = self.detector_conv[0].weight.view(256, 640)
        s.dconv = ops.gemm_nt(s.concat.view(-1, 640), wdet, amax_b=self.weight_amax(self.detector_conv[0].weight))
        s.dconv = s.dconv.view(B, T, 2, 256)
        s.bnd = _bn(self.detector_conv[1], s.dconv, train)
        dact = ops.bn_act_pool_fwd(s.dconv, s.bnd, pool=1, slope=slope)
        ddrop, s.mask_det = _dropout(self.dropout_cfg, _flat2(dact), p_blk)
        seq_d = ops.nhwc_to_seq(ddrop.view(B, T, 2, 256), 256)

        models = [self.sequence_classifier, self.sequence_detector]
        if models[0].model_type == "bilstm":
            (yc, yd), s.lstm = _lstm_forward(models, [seq_c, seq_d], train, need_grad, self.dropout_cfg, wa)
        else:
            yc, s.tf_c = _tf_forward(models[0], seq_c, train, need_grad, self.dropout_cfg)
            yd, s.tf_d = _tf_forward(models[1], seq_d, train, need_grad, self.dropout_cfg)
        s.yc, s.yd = yc, yd
        D = yc.shape[-1]
        if self.num_class == 1:
            out_cls = ops.head_fwd(yc.view(-1, D), self.classifier.weight, self.classifier.bias).view(B, T, 1)
        else:
            out_cls = ops.gemm_nt(yc.view(-1, D), self.classifier.weight, bias0=self.classifier.bias)
            out_cls = out_cls.view(B, T, self.num_class)
        out_det = ops.head_fwd(yd.view(-1, D), self.detector.weight, self.detector.bias).view(B, T)
        s.shape = (B, T, F)
        return out_cls, out_det, (s if need_grad else None)

    def _backward_impl(self, s, d_cls, d_det):
        g = self._grad_views()
        slope = self.leaky_relu_slope
        B, T, F = s.shape
        D = s.yc.shape[-1]
        dev = s.yc.device
        if d_cls is None:
            d_cls = torch.zeros((B, T, self.num_class), dtype=torch.float32, device=dev)
        if d_det is None:
            d_det = torch.zeros((B, T), dtype=torch.float32, device=dev)
        d_cls = d_cls.contiguous().float()
        d_det = d_det.contiguous().float()
        cls, det = self.classifier, self.detector
        if self.num_class == 1:
            dyc = ops.head_bwd(s.yc.view(-1, D), cls.weight, d_cls.view(-1), g[cls.weight], g[cls.bias])
        else:
            d2 = d_cls.view(-1, self.num_class)
            ops.gemm_tn(d2, s.yc.view(-1, D), out=g[cls.weight])
            ops.colsum(d2, g[cls.bias])
            dyc = ops.gemm_nt(d2, ops.transpose2d(cls.weight))
        dyd = ops.head_bwd(s.yd.view(-1, D), det.weight, d_det.view(-1), g[det.weight], g[det.bias])

        models = [self.sequence_classifier, self.sequence_detector]
        side = _SideWork(dev)               # weight-gradient kernels of the whole backward; joined once, at the end
        if models[0].model_type == "bilstm":
            dseq_c, dseq_d = _lstm_backward(models, s.lstm, [dyc.view(B, T, D), dyd.view(B, T, D)], g, side,
                                            self.weight_amax)
            if LSTM_WGRAD_JOIN_EARLY:
                side.join()
        else:
            dseq_d = _tf_backward(models[1], s.tf_d, dyd.view(B, T, D), g, side)
            dseq_c = _tf_backward(models[0], s.tf_c, dyc.view(B, T, D), g, side)

        self._write_status(dev)             # every recurrence of this step has been queued: its fault word is final
        if self._dp is not None:            # temporal heads + output heads are final once the side stream has
            # finished what is queued on it so far: the collectives are issued from that stream, the main stream does not wait
            self._dp.reduce_range(self._seq_offset(), self._grad_flat.numel(), after=side.pending_stream())

        # detector branch (model.py:103-112)
        p_blk = self.block_dropout
        d_ddrop = torch.empty((B, T, 2, 256), dtype=s.concat.dtype, device=dev)
        ops.seq_to_nhwc(dseq_d, d_ddrop, 256)
        d_dact = _dropout_bwd(_flat2(d_ddrop), p_blk, s.mask_det).view(B, T, 2, 256)
        bn1 = self.detector_conv[1]
        d_dconv = ops.bn_act_pool_bwd(s.dconv, d_dact, s.bnd, g[bn1.weight], g[bn1.bias], pool=1, slope=slope)
        wdet = self.detector_conv[0].weight
        side.run(lambda: ops.gemm_tn(_flat2(d_dconv), s.concat.view(-1, 640), out=g[wdet].view(256, 640)), d_dconv,
                 on=OVERLAP_CONV_WGRAD)
        d_concat = ops.gemm_nt(_flat2(d_dconv), ops.transpose2d(wdet.view(256, 640)), amax_b=self.weight_amax(wdet))
        d_concat = d_concat.view(B, T, 2, 640)
        # classifier branch joins at the pool_block output (channels 384..639 of the concat)
        ops.seq_to_nhwc(dseq_c, d_concat, 256, coff=384, accumulate=True)
        d_pool = torch.empty((B, T, 2, 256), dtype=s.concat.dtype, device=dev)
        _dropout_bwd(d_concat.view(-1, 640)[:, 384:640], p_blk if s.mask_pool is not None else 0.0, s.mask_pool,
                     out2d=_flat2(d_pool))
        bnp = self.pool_block[0]
        am3 = ops.amax_word()
        d_rb3 = ops.bn_act_pool_bwd(s.rb3, d_pool, s.bnp, g[bnp.weight], g[bnp.bias], pool=4, slope=slope, amax_out=am3)

        cuts = self._dp_cuts if self._dp is not None else None

        def block_done(k):                  # gradients of flat range [cuts[k], cuts[k+1]) are queued: reduce them
            if cuts is not None:
                self._dp.reduce_range(cuts[k], cuts[k + 1], after=side.pending_stream())

        # each block-input gradient leaves its absmax word behind; the detector tap's max-pool gradient, added in place
        # afterwards, merges the values it rewrote into the same word
        wa = self.weight_amax
        d_rb2, am2 = _res_backward(self.res_block3, s.r3, d_rb3, 2, slope, g, side, am3, wa)
        block_done(3)                       # res_block3 + pool_block + detector_conv
        ops.maxpool_bwd_add(s.rb2, d_concat, d_rb2, 10, coff=192, amax_out=am2, argmax=s.arg_rb2)
        d_rb1, am1 = _res_backward(self.res_block2, s.r2, d_rb2, 2, slope, g, side, am2, wa)
        block_done(2)
        ops.maxpool_bwd_add(s.rb1, d_concat, d_rb1, 20, coff=64, amax_out=am1, argmax=s.arg_rb1)
        d_cb, am_dcb = _res_backward(self.res_block1, s.r1, d_rb1, 2, slope, g, side, am1, wa)
        block_done(1)
        ops.maxpool_bwd_add(s.cb, d_concat, d_cb, 40, coff=0, amax_out=am_dcb, argmax=s.arg_cb)

        cbk = self.conv_block
        side.run(lambda: ops.conv3x3_wgrad(s.a0, d_cb, g[cbk[3].weight], amax_x=s.am_a0, amax_dy=am_dcb), d_cb, am_dcb,
                 on=OVERLAP_CONV_WGRAD)
        with ops.timer_tag("dgrad"):
            d_a0 = ops.conv3x3_fwd(d_cb, s.wd_cb, amax=am_dcb)
        d_y0 = ops.bn_act_pool_bwd(s.y0, d_a0, s.bn0, g[cbk[1].weight], g[cbk[1].bias], pool=1, slope=slope, dx=d_a0)
        ops.conv3x3_c1_wgrad(s.x_btf, d_y0, g[cbk[0].weight])
        side.join()                             # every weight gradient is final from here on
        if self._dp is not None:
            self._dp.reduce_range(0, cuts[1] if cuts is not None else self._seq_offset())
