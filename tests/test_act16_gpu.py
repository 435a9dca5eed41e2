"""bf16 ACTIVATION STORAGE for mixed precision (the `*_a16` entry points; reference trainer.py:226-235, README.md:36:
autocast keeps conv / linear outputs in half precision).

Every pass of the conv stack with bf16 tensors must equal the fp32-tensor pass of the same name on the same (bf16-
representable) inputs, rounded to bf16 once at the store: arithmetic stays fp32, only the HBM format changes.  The whole
training step with bf16 storage stays within bf16 rounding of the step with fp32 storage, and uses about half the
activation memory."""
import logging

import pytest
import torch
import torch.nn.functional as F

from oracle import model_ref
from pitchextractor_amd import ops
from pitchextractor_amd.model import JDCNet
from pitchextractor_amd.optimizers import build_optimizer
from pitchextractor_amd.trainer import Trainer
from tests.golden.make_golden import SEQ_CFG, TF_CFG, golden_input, golden_targets

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def r16(t):                      # a bf16-representable fp32 tensor
    return t.to(BF).float()


def test_bn_act_pool_passes_bf16_tensors(hip_device):
    B, T, Fq, C = 3, 17, 20, 64
    x = r16(rnd(B, T, Fq, C, seed=1) * 3).to(hip_device)
    gamma, beta = rnd(C, seed=2).to(hip_device) + 1.5, rnd(C, seed=3).to(hip_device)
    st32 = ops.bn_train_stats(x, gamma, beta, None, None)
    st16 = ops.bn_train_stats(x.to(BF), gamma, beta, None, None)
    assert torch.equal(st32.scale, st16.scale) and torch.equal(st32.shift, st16.shift)
    for pool in (1, 2, 4):
        y32 = ops.bn_act_pool_fwd(x, st32, pool=pool)
        y16 = ops.bn_act_pool_fwd(x.to(BF), st32, pool=pool)
        assert y16.dtype == BF and torch.equal(y16, y32.to(BF))
        dy = r16(rnd(B, T, Fq // pool, C, seed=4)).to(hip_device)
        g32 = [torch.empty(C, device=hip_device) for _ in range(2)]
        g16 = [torch.empty(C, device=hip_device) for _ in range(2)]
        dx32 = ops.bn_act_pool_bwd(x, dy, st32, g32[0], g32[1], pool=pool)
        dx16 = ops.bn_act_pool_bwd(x.to(BF), dy.to(BF), st32, g16[0], g16[1], pool=pool)
        assert dx16.dtype == BF and torch.equal(dx16, dx32.to(BF))
        assert torch.equal(g32[0], g16[0]) and torch.equal(g32[1], g16[1])


def test_pool_dropout_relayout_passes_bf16_tensors(hip_device):
    B, T, Fq, C = 2, 9, 20, 64
    x = r16(rnd(B, T, Fq, C, seed=1)).to(hip_device)
    wide32 = torch.zeros(B, T, 2, 640, device=hip_device)
    wide16 = torch.zeros(B, T, 2, 640, device=hip_device, dtype=BF)
    _, a32 = ops.maxpool_fwd(x, 10, out=wide32, coff=64, want_argmax=True)
    _, a16 = ops.maxpool_fwd(x.to(BF), 10, out=wide16, coff=64, want_argmax=True)
    assert torch.equal(a32, a16) and torch.equal(wide16, wide32.to(BF))
    dwide = r16(rnd(B, T, 2, 640, seed=2)).to(hip_device)
    dx32, dx16 = x.clone(), x.to(BF)
    ops.maxpool_bwd_add(x, dwide, dx32, 10, coff=64, argmax=a32)
    ops.maxpool_bwd_add(x, dwide.to(BF), dx16, 10, coff=64, argmax=a16)
    assert torch.equal(dx16, dx32.to(BF))
    x2 = r16(rnd(4096, 256, seed=3)).to(hip_device)
    y32, m = ops.dropout(x2, 0.5, seed=7, offset=3)
    y16, m16 = ops.dropout(x2.to(BF), 0.5, seed=7, offset=3)
    assert torch.equal(m, m16) and torch.equal(y16, y32.to(BF))
    seq32 = ops.nhwc_to_seq(wide32, 64, coff=64)
    seq16 = ops.nhwc_to_seq(wide16, 64, coff=64)
    assert seq16.dtype == torch.float32 and torch.equal(seq16, seq32)
    back16 = torch.zeros(B, T, 2, 640, device=hip_device, dtype=BF)
    ops.seq_to_nhwc(seq32, back16, 64, coff=64)
    assert torch.equal(back16[..., 64:128], wide16[..., 64:128])
    ops.seq_to_nhwc(seq32, back16, 64, coff=64, accumulate=True)
    assert torch.equal(back16[..., 64:128], (wide16[..., 64:128].float() * 2).to(BF))


@pytest.mark.parametrize("B,T,Fq,Ci,Co", [(2, 12, 10, 64, 64), (1, 9, 20, 128, 192), (2, 16, 40, 128, 128),
                                          (1, 3, 80, 64, 64), (1, 4, 5, 256, 256)])
def test_conv_kernels_bf16_tensors(hip_device, B, T, Fq, Ci, Co):
    """3x3 convolution forward (fragment-fed and implicit-GEMM), accumulate, BatchNorm partials and weight gradient
    with bf16 tensors: the fp32-tensor bf16-operand kernels' results, rounded once."""
    x = r16(rnd(B, T, Fq, Ci, seed=1)).to(hip_device)
    w = rnd(Co, Ci, 3, 3, seed=2, scale=0.1).to(hip_device)
    dy = r16(rnd(B, T, Fq, Co, seed=3)).to(hip_device)
    with ops.matmul_bf16(True):
        wf, _ = ops.conv3x3_repack(w, want_dgrad=False)
        y32, p32 = ops.conv3x3_fwd(x, wf, bn_stats=True)
        y16, p16 = ops.conv3x3_fwd(x.to(BF), wf, bn_stats=True)
        assert y16.dtype == BF and torch.equal(y16, y32.to(BF))
        if p32 is not None:      # the partials describe the ROUNDED tensor
            ref = torch.stack([y16.float().double().sum((0, 1, 2)), (y16.float().double() ** 2).sum((0, 1, 2))])
            assert torch.allclose(p16.sum(0), ref, rtol=1e-9, atol=1e-9)
        acc32 = r16(rnd(B, T, Fq, Co, seed=5)).to(hip_device)
        acc16 = acc32.to(BF)
        ops.conv3x3_fwd(x, wf, out=acc32, accumulate=True)
        ops.conv3x3_fwd(x.to(BF), wf, out=acc16, accumulate=True)
        assert torch.equal(acc16, acc32.to(BF))
        plain = ops.PackedWeight(wf.fp32)                      # the implicit-GEMM kernel
        assert torch.equal(ops.conv3x3_fwd(x.to(BF), plain), ops.conv3x3_fwd(x, plain).to(BF))
        dw32, dw16 = torch.empty_like(w), torch.empty_like(w)
        ops.conv3x3_wgrad(x, dy, dw32)
        ops.conv3x3_wgrad(x.to(BF), dy.to(BF), dw16)
        assert torch.equal(dw32, dw16)


def test_first_conv_and_gemms_bf16_tensors(hip_device):
    x = rnd(3, 20, 80, seed=1).to(hip_device)
    w = rnd(64, 1, 3, 3, seed=2, scale=0.3).to(hip_device)
    y32, p32 = ops.conv3x3_c1_fwd(x, w, bn_stats=True)
    with ops.matmul_bf16(True, "bf16", act16=True):
        assert ops.act_dtype() == BF
        y16, p16 = ops.conv3x3_c1_fwd(x, w, bn_stats=True)
    assert y16.dtype == BF and torch.equal(y16, y32.to(BF))
    dy = r16(rnd(3, 20, 80, 64, seed=3)).to(hip_device)
    dw32, dw16 = torch.empty_like(w), torch.empty_like(w)
    ops.conv3x3_c1_wgrad(x, dy, dw32)
    ops.conv3x3_c1_wgrad(x, dy.to(BF), dw16)
    assert torch.equal(dw32, dw16)
    A = r16(rnd(1000, 192, seed=4)).to(hip_device)
    Wt = rnd(128, 192, seed=5, scale=0.1).to(hip_device)
    Bm = r16(rnd(1000, 64, seed=6)).to(hip_device)
    with ops.matmul_bf16(True):
        c32 = ops.gemm_nt(A, Wt)
        c16 = ops.gemm_nt(A.to(BF), Wt)
        assert c16.dtype == BF and torch.equal(c16, c32.to(BF))
        acc32 = r16(rnd(1000, 128, seed=7)).to(hip_device)
        acc16 = acc32.to(BF)
        ops.gemm_nt(A, Wt, out=acc32, accumulate=True)
        ops.gemm_nt(A.to(BF), Wt, out=acc16, accumulate=True)
        assert torch.equal(acc16, acc32.to(BF))
        assert torch.equal(ops.gemm_tn(A, Bm), ops.gemm_tn(A.to(BF), Bm.to(BF)))


def _trainer(net, **kw):
    opt, sched = build_optimizer({"params": net.parameters(), "optimizer_params": {},
                                  "scheduler_params": {"max_lr": 3e-4, "pct_start": 0.0, "epochs": 100,
                                                       "steps_per_epoch": 8}})
    crit = {"l1": torch.nn.SmoothL1Loss(), "ce": torch.nn.BCEWithLogitsLoss()}
    return Trainer(model=net, criterion=crit, optimizer=opt, scheduler=sched, device="cuda:0",
                   loss_config={"lambda_f0": 0.1}, logger=logging.getLogger("t"), **kw)


@pytest.mark.parametrize("head", ["bilstm", "transformer"])
def test_training_steps_with_bf16_activation_storage(hip_device, head):
    """Five optimiser steps, mixed precision: bf16 activation storage against fp32 storage of the same operands.  The two
    differ by one bf16 rounding per stored tensor (2^-9 relative): losses within 1 %, and the storage mode must cut
    the step's peak memory."""
    cfg = dict(SEQ_CFG if head == "bilstm" else TF_CFG)
    cfg["hidden_size"] = 128
    state = model_ref.seeded_state(19, model_type=head, hidden_size=128)
    x = golden_input(6, B=8)
    f0, sil = golden_targets(6, B=8)
    batch = (x.transpose(-1, -2).contiguous(), f0, sil)
    runs, peaks = {}, {}
    torch.cuda.synchronize()
    held = torch.cuda.memory_allocated()        # what earlier tests of the process left cached (workspaces, scale words)
    for storage in ("fp32", "bf16"):
        net = JDCNet(num_class=1, sequence_model_config=dict(cfg))
        net.load_state_dict(state, strict=True)
        net = net.to(hip_device).train()
        net.block_dropout = 0.0
        tr = _trainer(net, use_mixed_precision=True, activation_storage=storage)
        assert tr.act16 == (storage == "bf16")
        tr.run(batch)
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        runs[storage] = [tr.run(batch) for _ in range(4)]
        torch.cuda.synchronize()
        peaks[storage] = torch.cuda.max_memory_allocated() - held
        del net, tr
    for a, b in zip(runs["fp32"], runs["bf16"]):
        for key in ("loss", "f0", "sil"):
            assert abs(a[key] - b[key]) <= 1e-2 * abs(a[key]) + 1e-4, (key, a, b)
    assert runs["bf16"][-1]["loss"] < runs["bf16"][0]["loss"]
    print(f"{head}: peak memory bf16 storage / fp32 storage = {peaks['bf16'] / peaks['fp32']:.3f}")
    assert peaks["bf16"] < 0.85 * peaks["fp32"]
