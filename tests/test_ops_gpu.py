"""Each HIP op, called through the C ABI, against a plain PyTorch fp32/fp64 CPU reference of the same op.

fp32 products run either on the native fp32 MFMA (an exact-f32 fmaf chain) or as the exact three-term
bf16 split ("x3", the default); both are held to 1e-5 of the result scale (north_star: 1e-4 relative
for fp32) and the GEMM / conv tests run in both modes.  Index/layout work is bit-exact.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from pitchextractor_amd import ops

pytestmark = pytest.mark.gpu


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def close(got, ref, tol=1e-5):
    got = got.detach().cpu().double()
    ref = ref.detach().double()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    scale = ref.abs().max().item() + 1e-30
    err = (got - ref).abs().max().item()
    assert err <= tol * scale, (err, scale)


def nhwc(x):      # NCHW -> channels-last contiguous
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


# ------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("M,N,K", [(300, 1536, 384), (1000, 64, 576), (257, 192, 96), (130, 20, 64),
                                   (64, 128, 32), (5, 360, 768), (1024, 256, 640), (33, 1, 4)])
@pytest.mark.parametrize("fp32_mode", ["native", "x3", "h2"])
def test_gemm_nt(hip_device, M, N, K, fp32_mode, monkeypatch):
    monkeypatch.setattr(ops, "FP32_MATMUL", fp32_mode)
    A, B = rnd(M, K, seed=1), rnd(N, K, seed=2)
    b0, b1 = rnd(N, seed=3), rnd(N, seed=4)
    ref = A.double() @ B.double().T
    close(ops.gemm_nt(A.to(hip_device), B.to(hip_device)), ref)
    close(ops.gemm_nt(A.to(hip_device), B.to(hip_device), bias0=b0.to(hip_device), bias1=b1.to(hip_device)),
          ref + b0.double() + b1.double())
    out = rnd(M, N, seed=5).to(hip_device)
    ref2 = ref + out.cpu().double()
    close(ops.gemm_nt(A.to(hip_device), B.to(hip_device), out=out, accumulate=True), ref2)


def bf16r(t):      # what the bf16 kernels see: operands rounded to bf16 (RNE), products exact in fp32
    return t.to(torch.bfloat16).to(torch.float64)


@pytest.mark.parametrize("M,N,K", [(300, 1536, 384), (1000, 64, 576), (257, 192, 96), (130, 20, 64), (33, 1, 4)])
def test_gemm_nt_bf16_operands(hip_device, M, N, K):
    """Mixed-precision variant: equals the fp64 product of the bf16-rounded operands to fp32 accumulation
    accuracy (1e-5 of scale), and is within bf16 rounding (2^-8 relative per operand) of the fp32 product."""
    A, B = rnd(M, K, seed=1), rnd(N, K, seed=2)
    b0 = rnd(N, seed=3)
    with ops.matmul_bf16(True):
        got = ops.gemm_nt(A.to(hip_device), B.to(hip_device), bias0=b0.to(hip_device))
    assert ops.MATMUL_BF16 is False
    close(got, bf16r(A) @ bf16r(B).T + b0.double())
    close(got, A.double() @ B.double().T + b0.double(), tol=2e-2)


def test_conv3x3_bf16_operands(hip_device):
    B, T, Fq, Ci, Co = 2, 12, 10, 64, 128
    x, w = rnd(B, Ci, T, Fq, seed=1), rnd(Co, Ci, 3, 3, seed=2, scale=0.1)
    wf, _ = ops.conv3x3_repack(w.to(hip_device))
    with ops.matmul_bf16(True):
        got = ops.conv3x3_fwd(nhwc(x).to(hip_device), wf)
    close(nchw(got), F.conv2d(bf16r(x), bf16r(w), padding=1))


def test_x3_split_is_as_accurate_as_native_fp32(hip_device):
    """The three-term bf16 split must not be a precision downgrade: against the fp64 product its error is
    held to the same bound as the native fp32 MFMA path (and to 2x the native path's own error + 1 ulp-ish
    slack), on a long-K product with a wide dynamic range."""
    g = torch.Generator().manual_seed(11)
    A = torch.randn(512, 4608, generator=g) * torch.exp(torch.randn(512, 4608, generator=g) * 2)
    B = torch.randn(256, 4608, generator=g) * torch.exp(torch.randn(256, 4608, generator=g) * 2)
    ref = A.double() @ B.double().T
    mag = A.double().abs() @ B.double().abs().T                      # condition-free error scale
    errs = {}
    prev = ops.FP32_MATMUL
    for mode in ("native", "x3", "h2"):
        ops.FP32_MATMUL = mode
        try:
            got = ops.gemm_nt(A.to(hip_device), B.to(hip_device)).cpu().double()
        finally:
            ops.FP32_MATMUL = prev
        errs[mode] = ((got - ref).abs() / mag).max().item()
    assert errs["native"] < 2e-5 and errs["x3"] <= 2 * errs["native"] and errs["h2"] <= 2 * errs["native"], errs


def test_absmax_is_exact_and_covers_strided_views(hip_device):
    """pe_absmax: the IEEE bits of max |x|, exact (integer max of bit patterns), for dense tensors and row-strided
    views, including a maximum sitting in the last element, a negative maximum and an all-zero tensor."""
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1000, 64, generator=g)
    x[999, 63] = -77.25
    xd = x.to(hip_device)
    bits = lambda v: torch.tensor(v, dtype=torch.float32).view(torch.int32).item()     # noqa: E731
    assert ops.absmax(xd).item() == bits(77.25)
    assert ops.absmax(xd[:, 16:48]).item() == bits(float(x[:, 16:48].abs().max()))
    assert ops.absmax(xd.view(10, 100, 64)).item() == bits(77.25)
    assert ops.absmax(torch.zeros(8, 4, device=hip_device)).item() == 0
    big = torch.randn(3000, 1024, generator=g)
    assert ops.absmax(big.to(hip_device)).item() == bits(float(big.abs().max()))


@pytest.mark.parametrize("scale_a,scale_b", [(1e-30, 1e20), (3e15, 2e-12), (1.0, 1e-36), (7e-17, 7e-17)])
def test_h2_products_over_the_exponent_range(hip_device, scale_a, scale_b, monkeypatch):
    """The two-term fp16 split takes its range from the per-tensor power-of-two scale: operands far outside fp16's
    own range (1e-30 .. 1e20, products down to 1e-39) and a 2^40 spread INSIDE one operand come out with the error
    of the native fp32 path, for the NT and the TN product."""
    monkeypatch.setattr(ops, "FP32_MATMUL", "h2")
    g = torch.Generator().manual_seed(5)
    M, N, K = 256, 192, 512
    A = torch.randn(M, K, generator=g) * torch.exp2(-40 * torch.rand(M, K, generator=g)) * scale_a
    B = torch.randn(N, K, generator=g) * scale_b
    ref = A.double() @ B.double().T
    mag = A.double().abs() @ B.double().abs().T
    got = ops.gemm_nt(A.to(hip_device), B.to(hip_device)).cpu().double()
    assert torch.isfinite(got).all()
    assert ((got - ref).abs() / mag).max().item() < 2e-6
    got_tn = ops.gemm_tn(A.t().contiguous().to(hip_device), B.t().contiguous().to(hip_device)).cpu().double()
    assert ((got_tn - ref).abs() / mag).max().item() < 2e-6


def test_h2_all_zero_operand(hip_device, monkeypatch):
    monkeypatch.setattr(ops, "FP32_MATMUL", "h2")
    A = torch.zeros(128, 64, device=hip_device)
    B = torch.randn(96, 64, device=hip_device)
    assert torch.equal(ops.gemm_nt(A, B), torch.zeros(128, 96, device=hip_device))


def test_weight_gradient_products_bf16_operands(hip_device):
    """Mixed precision also rounds the operands of the weight-gradient products (gemm_tn, 3x3 wgrad, dW_hh)
    to bf16, as autocast's backward does: each equals the fp64 product of the rounded operands."""
    A, B = rnd(3001, 192, seed=1), rnd(3001, 128, seed=2)
    with ops.matmul_bf16(True):
        got = ops.gemm_tn(A.to(hip_device), B.to(hip_device))
    close(got, bf16r(A).T @ bf16r(B))
    x, dy = rnd(2, 64, 12, 10, seed=3), rnd(2, 128, 12, 10, seed=4)
    dw = torch.empty(128, 64, 3, 3, device=hip_device)
    with ops.matmul_bf16(True):
        ops.conv3x3_wgrad(nhwc(x).to(hip_device), nhwc(dy).to(hip_device), dw)
    xr = bf16r(x).requires_grad_(False)
    w = torch.zeros(128, 64, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(xr, w, padding=1).backward(bf16r(dy))
    close(dw, w.grad)


def test_gemm_nt_strided_rows(hip_device):
    big = rnd(40, 7, 96, seed=6).to(hip_device)              # rows taken at a fixed time step: ld = 7*96
    A = big[:, 3, 32:96]
    B = rnd(50, 64, seed=7).to(hip_device)
    out = torch.zeros(40, 80, device=hip_device)
    ops.gemm_nt(A, B, out=out[:, 10:60])
    close(out[:, 10:60], A.cpu().double() @ B.cpu().double().T)
    assert (out[:, :10] == 0).all() and (out[:, 60:] == 0).all()


@pytest.mark.parametrize("K,M,N", [(5000, 64, 64), (3001, 192, 128), (777, 1536, 96), (20000, 128, 64), (64, 4, 8)])
@pytest.mark.parametrize("fp32_mode", ["native", "x3", "h2"])
def test_gemm_tn(hip_device, K, M, N, fp32_mode, monkeypatch):
    monkeypatch.setattr(ops, "FP32_MATMUL", fp32_mode)
    A, B = rnd(K, M, seed=1), rnd(K, N, seed=2)
    ref = A.double().T @ B.double()
    close(ops.gemm_tn(A.to(hip_device), B.to(hip_device)), ref)
    out = rnd(M, N, seed=3).to(hip_device)
    ref2 = ref + out.cpu().double()
    close(ops.gemm_tn(A.to(hip_device), B.to(hip_device), out=out, accumulate=True), ref2)


def test_gemm_tn_is_deterministic(hip_device):
    A, B = rnd(30000, 128, seed=1).to(hip_device), rnd(30000, 128, seed=2).to(hip_device)
    assert torch.equal(ops.gemm_tn(A, B), ops.gemm_tn(A, B))


def test_transpose2d(hip_device):
    x = rnd(70, 45, seed=1).to(hip_device)
    assert torch.equal(ops.transpose2d(x).cpu(), x.cpu().T.contiguous())


# ------------------------------------------------------------------ conv
@pytest.mark.parametrize("B,T,Fq,Ci,Co", [(2, 12, 10, 64, 64), (1, 9, 7, 64, 128), (2, 5, 20, 128, 192),
                                          (1, 6, 10, 192, 256), (1, 4, 5, 256, 256), (3, 16, 40, 128, 128),
                                          (1, 3, 80, 64, 64), (1, 5, 45, 64, 128), (1, 4, 50, 64, 128),
                                          (2, 1, 33, 96, 160),
                                          (2, 64, 40, 64, 64), (2, 48, 40, 128, 128)])   # > 2048 pixels: split-K slabs
@pytest.mark.parametrize("fp32_mode", ["native", "x3", "h2"])
def test_conv3x3_fwd_dgrad_wgrad(hip_device, B, T, Fq, Ci, Co, fp32_mode, monkeypatch):
    monkeypatch.setattr(ops, "FP32_MATMUL", fp32_mode)
    x = rnd(B, Ci, T, Fq, seed=1).double().requires_grad_(True)
    w = rnd(Co, Ci, 3, 3, seed=2, scale=0.1).double().requires_grad_(True)
    y = F.conv2d(x, w, padding=1)
    dy = rnd(B, Co, T, Fq, seed=3).double()
    y.backward(dy)
    xd, wd_, dyd = nhwc(x.detach().float()).to(hip_device), w.detach().float().to(hip_device), \
        nhwc(dy.float()).to(hip_device)
    wf, wdg = ops.conv3x3_repack(wd_)
    close(nchw(ops.conv3x3_fwd(xd, wf)), y)
    close(nchw(ops.conv3x3_fwd(dyd, wdg)), x.grad)
    dw = torch.empty_like(wd_)
    close(ops.conv3x3_wgrad(xd, dyd, dw), w.grad)
    acc = rnd(B, T, Fq, Co, seed=4).to(hip_device)
    ref_acc = y.detach() + nchw(acc.cpu()).double()
    close(nchw(ops.conv3x3_fwd(xd, wf, out=acc, accumulate=True)), ref_acc)


def test_weight_fragment_pack_layout(hip_device):
    """pe_wfrag_pack: fragment (kb, nb, term), lane 32 h + r  <->  w[32 nb + r][16 kb + 8 h .. + 7]; the three
    terms are the exact truncation split (hi + mid + lo == w bit for bit), one term = RNE bf16."""
    N, K = 70, 96                                            # N not a multiple of 32: the tail rows pack as zeros
    w = rnd(N, K, seed=3) * torch.exp(rnd(N, K, seed=4) * 3)
    for terms in (3, 1):
        raw = ops.wfrag_pack(w.to(hip_device), terms).cpu()
        frag = raw.view(torch.bfloat16).view(K // 16, 3, terms, 64, 8).float()        # [kb][nb][term][lane][8]
        wpad = torch.zeros(96, K)
        wpad[:N] = w
        ref = wpad.view(3, 32, K // 16, 2, 8).permute(2, 0, 3, 1, 4).reshape(K // 16, 3, 64, 8)   # lane = 32 h + r
        if terms == 3:
            assert torch.equal(frag[:, :, 0] + frag[:, :, 1] + frag[:, :, 2], ref)
            hi = (ref.view(torch.int32) & -65536).view(torch.float32)
            assert torch.equal(frag[:, :, 0], hi)
        else:
            assert torch.equal(frag[:, :, 0], ref.to(torch.bfloat16).float())


@pytest.mark.parametrize("B,T,Fq,Ci,Co", [(3, 16, 40, 128, 128), (2, 5, 20, 128, 192), (1, 3, 80, 64, 64),
                                          (1, 6, 10, 192, 256), (2, 7, 40, 128, 64)])
@pytest.mark.parametrize("mode", ["h2", "x3", "bf16"])
def test_conv3x3_fragment_fed_kernel_matches_implicit_gemm(hip_device, B, T, Fq, Ci, Co, mode, monkeypatch):
    """The halo-staged kernel (weights as pre-packed MFMA fragments from L2) and the implicit-GEMM kernel (im2col
    gathered on the fly, weights through LDS) multiply the same operand terms; they differ only in the order the k
    (tap, channel) blocks are summed, so outputs agree to fp32 accumulation accuracy, forward and data gradient."""
    monkeypatch.setattr(ops, "FP32_MATMUL", mode if mode != "bf16" else "x3")
    x = nhwc(rnd(B, Ci, T, Fq, seed=1)).to(hip_device)
    w = rnd(Co, Ci, 3, 3, seed=2, scale=0.1).to(hip_device)
    outs = {}
    for frag in (True, False):
        monkeypatch.setattr(ops, "CONV_WFRAG", frag)
        with ops.matmul_bf16(mode == "bf16"):
            wf, wd = ops.conv3x3_repack(w)
            assert (wf.frag is not None) == frag
            y = ops.conv3x3_fwd(x, wf)
            y_in = outs[True][0] if outs else y              # both data-gradient products read the SAME tensor
            outs[frag] = (y, ops.conv3x3_fwd(y_in, wd))
    for a, b in zip(outs[True], outs[False]):
        assert (a - b).abs().max().item() <= 2e-6 * b.abs().max().item()


@pytest.mark.parametrize("B,T,Fq,Ci,Co,acc", [(3, 16, 40, 128, 128, False), (2, 5, 20, 128, 192, True), (1, 3, 80, 64, 64, False),
                                              (2, 9, 10, 192, 256, True)])
def test_conv3x3_epilogue_bn_statistics(hip_device, B, T, Fq, Ci, Co, acc, monkeypatch):
    """BatchNorm statistics as a by-product of the convolution epilogue (column sums of the FINAL outputs, incl. the
    residual accumulate) == the separate statistics pass over the stored activation, incl. the running-stat update;
    pixel counts that are not multiples of the 128-pixel tile."""
    monkeypatch.setattr(ops, "FP32_MATMUL", "x3")
    x = nhwc(rnd(B, Ci, T, Fq, seed=1)).to(hip_device)
    w = rnd(Co, Ci, 3, 3, seed=2, scale=0.1).to(hip_device)
    wf, _ = ops.conv3x3_repack(w, True, False)
    out = rnd(B, T, Fq, Co, seed=3).to(hip_device) if acc else None
    y, parts = ops.conv3x3_fwd(x, wf, out=out, accumulate=acc, bn_stats=True)
    assert parts is not None and parts.shape[1:] == (2, Co)
    gamma, beta = (rnd(Co, seed=4).abs() + 0.5).to(hip_device), rnd(Co, seed=5).to(hip_device)
    rm_a, rv_a = rnd(Co, seed=6).to(hip_device), (rnd(Co, seed=7).abs() + 0.5).to(hip_device)
    rm_b, rv_b = rm_a.clone(), rv_a.clone()
    st_a = ops.bn_train_stats(y, gamma, beta, rm_a, rv_a)
    st_b = ops.bn_train_stats(y, gamma, beta, rm_b, rv_b, partials=parts)
    for u, v in ((st_a.mean, st_b.mean), (st_a.invstd, st_b.invstd), (st_a.scale, st_b.scale), (st_a.shift, st_b.shift),
                 (rm_a, rm_b), (rv_a, rv_b)):
        close(v, u.cpu(), 1e-6)


def test_conv3x3_first_layer(hip_device):
    B, T, Fq = 3, 17, 80
    mel = rnd(B, 1, Fq, T, seed=1)                          # (B,1,80,T) as the loader yields it
    x = mel.transpose(-1, -2).double().requires_grad_(False)  # (B,1,T,80) view, what the model receives
    w = rnd(64, 1, 3, 3, seed=2).double().requires_grad_(True)
    y = F.conv2d(x, w, padding=1)
    dy = rnd(B, 64, T, Fq, seed=3).double()
    y.backward(dy)
    xv = mel.to(hip_device).transpose(-1, -2)[:, 0]          # strided (B,T,80)
    assert not xv.is_contiguous()
    close(nchw(ops.conv3x3_c1_fwd(xv, w.detach().float().to(hip_device))), y)
    # with bn_stats the kernel also leaves the BatchNorm batch statistics of its output (no pass over y needed)
    yd, parts = ops.conv3x3_c1_fwd(xv, w.detach().float().to(hip_device), bn_stats=True)
    assert torch.equal(yd, ops.conv3x3_c1_fwd(xv, w.detach().float().to(hip_device))) and parts.dtype == torch.float64
    gamma, beta = torch.ones(64, device=hip_device), torch.zeros(64, device=hip_device)
    st_a = ops.bn_train_stats(yd, gamma, beta, None, None)
    st_b = ops.bn_train_stats(yd, gamma, beta, None, None, partials=parts)
    close(st_b.mean, st_a.mean.cpu(), 1e-6)
    close(st_b.invstd, st_a.invstd.cpu(), 1e-6)
    close(st_b.mean, y.mean(dim=(0, 2, 3)), 1e-5)
    dw = torch.empty(64, 1, 3, 3, device=hip_device)
    close(ops.conv3x3_c1_wgrad(xv, nhwc(dy.float()).to(hip_device), dw), w.grad)


# ------------------------------------------------------------------ BN / LReLU / pool
@pytest.mark.parametrize("C,Fq,pool", [(64, 80, 1), (64, 80, 2), (128, 40, 2), (192, 20, 2), (256, 10, 4), (256, 2, 1)])
def test_bn_lrelu_pool_block(hip_device, C, Fq, pool):
    B, T = 2, 6
    x = (rnd(B, C, T, Fq, seed=1) * 2 + 0.5).double().requires_grad_(True)
    bn = torch.nn.BatchNorm2d(C).double()
    with torch.no_grad():
        bn.weight.copy_(rnd(C, seed=2).abs() + 0.5)
        bn.bias.copy_(rnd(C, seed=3))
        bn.running_mean.copy_(rnd(C, seed=4))
        bn.running_var.copy_(rnd(C, seed=5).abs() + 0.5)
    rm0, rv0 = bn.running_mean.clone(), bn.running_var.clone()
    bn.train()
    y = F.max_pool2d(F.leaky_relu(bn(x), 0.01), (1, pool)) if pool > 1 else F.leaky_relu(bn(x), 0.01)
    dy = rnd(*y.shape, seed=6).double()
    y.backward(dy)

    xd = nhwc(x.detach().float()).to(hip_device)
    g, b = bn.weight.detach().float().to(hip_device), bn.bias.detach().float().to(hip_device)
    rm, rv = rm0.float().to(hip_device), rv0.float().to(hip_device)
    st = ops.bn_train_stats(xd, g, b, rm, rv)
    close(rm, bn.running_mean)
    close(rv, bn.running_var)
    close(nchw(ops.bn_act_pool_fwd(xd, st, pool=pool)), y)
    dg, db = torch.empty(C, device=hip_device), torch.empty(C, device=hip_device)
    dx = ops.bn_act_pool_bwd(xd, nhwc(dy.float()).to(hip_device), st, dg, db, pool=pool)
    close(nchw(dx), x.grad, 2e-5)
    close(dg, bn.weight.grad, 2e-5)
    close(db, bn.bias.grad, 2e-5)

    bn.eval()
    ye = F.leaky_relu(bn(x.detach()), 0.01)
    ste = ops.bn_eval_affine(g, b, rm, rv)
    close(nchw(ops.bn_act_pool_fwd(xd, ste, pool=1)), ye)


def test_bn_block_into_channel_slice(hip_device):
    x = rnd(2, 3, 8, 256, seed=1).to(hip_device)
    st = ops.bn_train_stats(x, torch.ones(256, device=hip_device), torch.zeros(256, device=hip_device), None, None)
    dense = ops.bn_act_pool_fwd(x, st, pool=4)
    wide = torch.full((2, 3, 2, 640), 7.0, device=hip_device)
    ops.bn_act_pool_fwd(x, st, pool=4, out=wide, coff=384)
    assert torch.equal(wide[..., 384:], dense) and (wide[..., :384] == 7).all()


@pytest.mark.parametrize("C,Fq,pool,coff", [(64, 80, 40, 0), (128, 40, 20, 64), (192, 20, 10, 192)])
def test_maxpool_taps(hip_device, C, Fq, pool, coff):
    B, T = 2, 5
    x = rnd(B, C, T, Fq, seed=1).double().requires_grad_(True)
    y = F.max_pool2d(x, (1, pool))
    dy = rnd(*y.shape, seed=2).double()
    y.backward(dy)
    xd = nhwc(x.detach().float()).to(hip_device)
    wide = torch.zeros(B, T, 2, 640, device=hip_device)
    ops.maxpool_fwd(xd, pool, out=wide, coff=coff)
    assert torch.equal(nchw(wide[..., coff:coff + C].contiguous()).cpu(), y.detach().float())
    dwide = torch.zeros(B, T, 2, 640, device=hip_device)
    dwide[..., coff:coff + C] = nhwc(dy.float()).to(hip_device)
    base = rnd(B, T, Fq, C, seed=3).to(hip_device)
    dx = base.clone()
    ops.maxpool_bwd_add(xd, dwide, dx, pool, coff=coff)
    close(nchw(dx - base), x.grad, 1e-6)
    # the same gradient from the recorded positions of the maxima, without reading x
    wide2 = torch.zeros(B, T, 2, 640, device=hip_device)
    _, arg = ops.maxpool_fwd(xd, pool, out=wide2, coff=coff, want_argmax=True)
    assert torch.equal(wide2, wide) and arg.dtype == torch.uint8 and int(arg.max()) < pool
    dx2 = base.clone()
    ops.maxpool_bwd_add(xd, dwide, dx2, pool, coff=coff, argmax=arg)
    assert torch.equal(dx2, dx)


@pytest.mark.parametrize("pool", [1, 2, 4])
def test_bn_pool_passes_leave_the_absmax_of_their_output(hip_device, pool):
    """amax_out of bn_act_pool_fwd / bn_act_pool_bwd: exactly the IEEE bits of max |output|; maxpool_bwd_add merges
    an upper bound of the updated tensor's maximum into the same word."""
    B, T, Fq, C = 3, 17, 20, 64
    x = (rnd(B, T, Fq, C, seed=1) * 3).to(hip_device)
    gamma, beta = rnd(C, seed=2).to(hip_device) + 1.5, rnd(C, seed=3).to(hip_device)
    rm, rv = torch.zeros(C, device=hip_device), torch.ones(C, device=hip_device)
    st = ops.bn_train_stats(x, gamma, beta, rm, rv)
    bits = lambda t: t.abs().max().view(torch.int32).item()                 # noqa: E731
    w = torch.zeros(1, dtype=torch.int32, device=hip_device)
    y = ops.bn_act_pool_fwd(x, st, pool=pool, amax_out=w)
    assert w.item() == bits(y)
    dy = rnd(B, T, Fq // pool, C, seed=4).to(hip_device)
    w2 = torch.zeros(1, dtype=torch.int32, device=hip_device)
    dg, db = torch.empty(C, device=hip_device), torch.empty(C, device=hip_device)
    dx = ops.bn_act_pool_bwd(x, dy, st, dg, db, pool=pool, amax_out=w2)
    assert w2.item() == bits(dx)
    if Fq % 10 == 0:
        wide = (rnd(B, T, 2, 640, seed=5) * 50).to(hip_device)
        ops.maxpool_bwd_add(x, wide, dx, 10, coff=64, amax_out=w2)
        assert w2.item() >= bits(dx) and w2.view(torch.float32).item() <= 2.0 * dx.abs().max().item()


def test_dropout_mask_export_replay_and_rate(hip_device):
    x = rnd(4096, 512, seed=1).to(hip_device)
    y, mask = ops.dropout(x, 0.5, seed=123, offset=0)
    keep = mask.float().mean().item()
    assert abs(keep - 0.5) < 0.01
    assert torch.equal(y, x * mask.float() * 2.0)
    y2, _ = ops.dropout(x, 0.5, mask_in=mask)
    assert torch.equal(y, y2)
    y3, mask3 = ops.dropout(x, 0.5, seed=123, offset=0)
    assert torch.equal(mask, mask3)                            # counter-based: same (seed, offset) -> same mask
    _, mask4 = ops.dropout(x, 0.5, seed=123, offset=x.numel() // 4)
    assert not torch.equal(mask, mask4)
    y5, m5 = ops.dropout(x, 0.1, seed=5)
    assert abs(m5.float().mean().item() - 0.9) < 0.01
    torch.testing.assert_close(y5, x * m5.float() / 0.9)
    wide = torch.zeros(4096, 640, device=hip_device)
    ops.dropout(x[:, :256].contiguous(), 0.5, out2d=wide[:, 384:640], mask_in=mask[:, :256].contiguous())
    assert torch.equal(wide[:, 384:640], x[:, :256] * mask[:, :256].float() * 2.0)


def test_seq_relayout(hip_device):
    B, T, C = 2, 7, 256
    x = rnd(B, C, T, 2, seed=1)                               # NCHW (B,256,T,2)
    ref = x.permute(0, 2, 1, 3).contiguous().view(B, T, 2 * C)   # model.py:93
    wide = torch.zeros(B, T, 2, 640)
    wide[..., 384:] = nhwc(x)
    seq = ops.nhwc_to_seq(wide.to(hip_device), C, coff=384)
    assert torch.equal(seq.cpu(), ref)
    back = torch.zeros(B, T, 2, 640, device=hip_device)
    ops.seq_to_nhwc(seq, back, C, coff=384)
    assert torch.equal(back.cpu(), wide)
    ops.seq_to_nhwc(seq, back, C, coff=384, accumulate=True)
    assert torch.equal(back.cpu(), 2 * wide)


# ------------------------------------------------------------------ LSTM
@pytest.mark.parametrize("persistent", [False, True])
@pytest.mark.parametrize("B,T,In,H", [(5, 7, 96, 64), (70, 4, 64, 32), (3, 11, 128, 96), (130, 9, 64, 384), (66, 6, 64, 128)])
def test_lstm_layer_bidirectional(hip_device, B, T, In, H, persistent, monkeypatch):
    monkeypatch.setattr(ops, "USE_PERSISTENT_LSTM", persistent)
    torch.manual_seed(0)
    ref = torch.nn.LSTM(In, H, num_layers=1, batch_first=True, bidirectional=True).double()
    x = rnd(B, T, In, seed=1).double().requires_grad_(True)
    y, _ = ref(x)
    dy = rnd(B, T, 2 * H, seed=2).double()
    y.backward(dy)

    dev = hip_device
    xd = x.detach().float().to(dev)
    P = {n: p.detach().float().to(dev) for n, p in ref.named_parameters()}
    yd = torch.empty(B, T, 2 * H, device=dev)
    gates, cbuf, whh, ysl = [], [], [], []
    for d, sfx in enumerate(("", "_reverse")):
        g = ops.gemm_nt(xd.view(-1, In), P["weight_ih_l0" + sfx], bias0=P["bias_ih_l0" + sfx],
                        bias1=P["bias_hh_l0" + sfx]).view(B, T, 4 * H)
        gates.append(g); cbuf.append(torch.empty(B, T, H, device=dev)); whh.append(P["weight_hh_l0" + sfx])
        ysl.append(yd[:, :, d * H:(d + 1) * H])
    ops.lstm_fwd(whh, gates, ysl, cbuf, [0, 1], B, T, H)
    close(yd, y, 2e-5)

    dyd = dy.float().to(dev)
    whh_t = [ops.transpose2d(w) for w in whh]
    dsl = [dyd[:, :, d * H:(d + 1) * H] for d in range(2)]
    dcar = [torch.empty(B, H, device=dev) for _ in range(2)]
    # the k-split recurrence kernel also emits the bias gradients as per-batch-tile rows (H = 384, persistent)
    nrows = ops.lstm_bwd_dbias_rows(2, B, T, H, dsl[0].stride(1), dev)
    assert (nrows > 0) == (persistent and H == 384 and ops.FP32_MATMUL in ("x3", "h2"))
    brows = [torch.full((nrows, 4 * H), float("nan"), device=dev) for _ in range(2)] if nrows else None
    amx = [torch.zeros(1, dtype=torch.int32, device=dev) for _ in range(2)] if nrows else None
    assert ops.lstm_bwd(whh_t, gates, cbuf, dsl, dcar, [0, 1], B, T, H, dbias_rows=brows, amax_out=amx) == (nrows > 0)
    if nrows:        # ... and the largest gate-gradient magnitude per cell (scale source of the "h2" products), exactly
        for d in range(2):
            assert amx[d].item() == gates[d].abs().max().view(torch.int32).item()
    dx = torch.empty(B, T, In, device=dev)
    for d, sfx in enumerate(("", "_reverse")):
        dg = gates[d].view(-1, 4 * H)
        dwih = ops.gemm_tn(dg, xd.view(-1, In))
        close(dwih, dict(ref.named_parameters())["weight_ih_l0" + sfx].grad, 5e-5)
        dwhh = torch.empty(4 * H, H, device=dev)
        ops.lstm_whh_grad(gates[d], ysl[d], dwhh, d, B, T, H)
        close(dwhh, dict(ref.named_parameters())["weight_hh_l0" + sfx].grad, 5e-5)
        db0, db1 = torch.empty(4 * H, device=dev), torch.empty(4 * H, device=dev)
        ops.colsum(dg, db0, db1)
        close(db0, dict(ref.named_parameters())["bias_ih_l0" + sfx].grad, 5e-5)
        assert torch.equal(db0, db1)
        if nrows:
            assert nrows == (B + 63) // 64
            close(ops.colsum(brows[d], torch.empty(4 * H, device=dev)),
                  dict(ref.named_parameters())["bias_ih_l0" + sfx].grad, 5e-5)
        ops.gemm_nt(dg, ops.transpose2d(P["weight_ih_l0" + sfx]), out=dx.view(-1, In), accumulate=(d > 0))
    close(dx, x.grad, 5e-5)
    assert not ops.persistent_lstm_error(dev)


@pytest.mark.parametrize("B,T,H", [(130, 9, 384), (5, 7, 64), (66, 6, 128)])
def test_lstm_recurrence_mixed_precision(hip_device, B, T, H):
    """training.mixed_precision also rounds W_hh and the h / dgates rows of the persistent recurrences to bf16
    (fp32 accumulate and cell state): forward output and the recurrent data gradient stay within bf16-level
    error of the fp64 layer."""
    In = 64
    torch.manual_seed(0)
    ref = torch.nn.LSTM(In, H, num_layers=1, batch_first=True, bidirectional=False).double()
    x = rnd(B, T, In, seed=1).double().requires_grad_(True)
    y, _ = ref(x)
    dy = rnd(B, T, H, seed=2).double()
    y.backward(dy)
    dev = hip_device
    P = {n: p.detach().float().to(dev) for n, p in ref.named_parameters()}
    g = ops.gemm_nt(x.detach().float().to(dev).view(-1, In), P["weight_ih_l0"], bias0=P["bias_ih_l0"],
                    bias1=P["bias_hh_l0"]).view(B, T, 4 * H)
    yd, cb = torch.empty(B, T, H, device=dev), torch.empty(B, T, H, device=dev)
    with ops.matmul_bf16(True):
        ops.lstm_fwd([P["weight_hh_l0"]], [g], [yd], [cb], [0], B, T, H)
        close(yd, y, 2e-2)
        if H == 384:                                                        # the persistent bf16 kernels' hidden size
            assert (yd.cpu().double() - y.detach()).abs().max() > 1e-6      # the bf16 path really ran
        dcar = [torch.empty(B, H, device=dev)]
        ops.lstm_bwd([ops.transpose2d(P["weight_hh_l0"])], [g], [cb], [dy.float().to(dev)], dcar, [0], B, T, H)
    dx = ops.gemm_nt(g.view(-1, 4 * H), ops.transpose2d(P["weight_ih_l0"]))
    close(dx.view(B, T, In), x.grad, 3e-2)
    assert not ops.persistent_lstm_error(dev)


def test_mixed_precision_recurrences_are_exactly_batch_and_scale_invariant(hip_device):
    """The persistent bf16 recurrences (forward; backward with its bf16 partial-tile exchange) on 4 cells: run-to-run
    bit-identical, a sample's results do not depend on the batch it sits in (B = 256 of 32 replicas vs B = 8, and
    replica vs replica), and scaling dY by 2^-5 scales every gate gradient by exactly 2^-5 (bf16 rounding is
    scale-free): whatever differs between two batch sizes at model level comes from their inputs, not from here."""
    dev = hip_device
    T, H = 24, 384
    torch.manual_seed(0)
    whh = [torch.randn(4 * H, H, device=dev) * 0.05 for _ in range(4)]
    g8 = [torch.randn(8, T, 4 * H, device=dev) for _ in range(4)]
    dy8 = torch.randn(8, T, 2 * H, device=dev)

    def run(B, scale=1.0):
        reps = B // 8
        gates = [g.repeat(reps, 1, 1).contiguous() for g in g8]
        ys = [torch.empty(B, T, 2 * H, device=dev) for _ in range(2)]
        ysl = [ys[i // 2][:, :, (i % 2) * H:(i % 2 + 1) * H] for i in range(4)]
        cb = [torch.empty(B, T, H, device=dev) for _ in range(4)]
        with ops.matmul_bf16(True):
            assert ops._persistent_ok(4, B, H, dev)
            ops.lstm_fwd(whh, gates, ysl, cb, [0, 1, 0, 1], B, T, H)
            dy = (dy8 * scale).repeat(reps, 1, 1).contiguous()
            dsl = [dy[:, :, (i % 2) * H:(i % 2 + 1) * H] for i in range(4)]
            dc = [torch.empty(B, H, device=dev) for _ in range(4)]
            ops.lstm_bwd([ops.transpose2d(w) for w in whh], gates, cb, dsl, dc, [0, 1, 0, 1], B, T, H)
        return gates, ys

    a, ya = run(256)
    b, yb = run(256)
    c, yc = run(8)
    e, _ = run(256, 1.0 / 32)
    for i in range(4):
        assert torch.equal(a[i], b[i]) and torch.equal(ya[i // 2], yb[i // 2])
        assert torch.equal(a[i][:8], c[i]) and torch.equal(a[i][8:16], a[i][:8]) and torch.equal(ya[i // 2][:8], yc[i // 2])
        assert torch.equal(e[i][:8] * 32, c[i])
    assert not ops.persistent_lstm_error(dev)


# ------------------------------------------------------------------ heads / loss / AdamW
def test_heads(hip_device):
    R, D = 777, 768
    x = rnd(R, D, seed=1).double().requires_grad_(True)
    for n_out in (1, 2):
        w = rnd(n_out, D, seed=2, scale=0.05).double().requires_grad_(True)
        b = rnd(n_out, seed=3).double().requires_grad_(True)
        x.grad = None
        y = (x @ w.T + b).sum(-1)
        dy = rnd(R, seed=4).double()
        y.backward(dy)
        xd, wd, bd = (t.detach().float().to(hip_device) for t in (x, w, b))
        close(ops.head_fwd(xd, wd, bd), y)
        dw, db = torch.empty(n_out, D, device=hip_device), torch.empty(n_out, device=hip_device)
        dx = ops.head_bwd(xd, wd, dy.float().to(hip_device), dw, db)
        close(dx, x.grad)
        close(dw, w.grad, 2e-5)
        close(db, b.grad, 2e-5)


def test_loss_matches_torch_criteria(hip_device):
    R = 4 * 192
    f0p = (rnd(R, seed=1) * 150 + 100).double().requires_grad_(True)
    f0 = torch.where(rnd(R, seed=2) > -0.5, rnd(R, seed=3).abs() * 200 + 80, torch.zeros(R)).double()
    f0[:5] = f0p.detach()[:5] + torch.tensor([0.3, -0.7, 0.999, -1.0, 1.0001], dtype=torch.double)  # both branches
    silp = (rnd(R, seed=4) * 3).double().requires_grad_(True)
    sil = (f0 == 0).double()
    l1 = torch.nn.SmoothL1Loss()(f0p, f0)
    bce = torch.nn.BCEWithLogitsLoss()(silp, sil)
    loss = 0.1 * l1 + bce
    loss.backward()
    out3, d_f0, d_sil = ops.f0_sil_loss(f0p.detach().float().to(hip_device), f0.float().to(hip_device),
                                        silp.detach().float().to(hip_device), sil.float().to(hip_device), 0.1)
    np.testing.assert_allclose(out3.cpu().numpy(), [loss.item(), 0.1 * l1.item(), bce.item()], rtol=2e-6)
    close(d_f0, f0p.grad, 1e-5)
    close(d_sil, silp.grad, 1e-5)


def test_bins_ce_loss_matches_float64_oracle(hip_device):
    """N4: 360-bin classification loss + gradients vs the float64 restatement (oracle/model_ref.py)."""
    from oracle import model_ref
    g = torch.Generator().manual_seed(5)
    R, C = 517, 360
    logits = torch.randn(R, C, generator=g) * 3
    f0 = torch.rand(R, generator=g) * 600 + 40
    f0[torch.rand(R, generator=g) < 0.3] = 0.0
    f0[:4] = torch.tensor([0.0, 5.0, 31.7, 3000.0])                      # unvoiced, below / on / above the grid
    silp, sil = torch.randn(R, generator=g), (f0 == 0).float()
    lg = logits.double().requires_grad_(True)
    sp = silp.double().requires_grad_(True)
    tot, lf0, bce = model_ref.jdc_bins_loss(lg, sp, f0.double(), sil.double(), 0.1)
    tot.backward()
    out4, d_logits, d_sil = ops.f0_bins_ce_loss(logits.to(hip_device), f0.to(hip_device), silp.to(hip_device),
                                                sil.to(hip_device), 0.1)
    np.testing.assert_allclose(out4[:3].cpu().numpy(), [tot.item(), lf0.item(), bce.item()], rtol=2e-6)
    assert int(out4[3].item()) == int((f0 > 0).sum())
    close(d_logits, lg.grad, tol=2e-6)
    close(d_sil, sp.grad, tol=2e-6)
    # bins are index work: bit-exact against the float64 mapping
    onehot_rows = (d_logits.cpu() < 0).float().argmax(dim=1)
    ref_bins = torch.from_numpy(model_ref.f0_to_bins(f0.numpy()))
    voiced = ref_bins >= 0
    assert torch.equal(onehot_rows[voiced], ref_bins[voiced])
    assert (d_logits.cpu()[~voiced] == 0).all()
    # no voiced frame at all: CE term is 0 and its gradient vanishes
    out4, d_logits, _ = ops.f0_bins_ce_loss(logits.to(hip_device), torch.zeros(R, device=hip_device),
                                            silp.to(hip_device), torch.ones(R, device=hip_device), 0.1)
    assert out4[1].item() == 0.0 and (d_logits == 0).all()


@pytest.mark.parametrize("rows,cols", [(1000, 360), (49152 // 8, 1536), (3, 8), (257, 512)])
def test_colsum(hip_device, rows, cols):
    x = rnd(rows, cols, seed=9)
    a, b = torch.empty(cols, device=hip_device), torch.empty(cols, device=hip_device)
    ops.colsum(x.to(hip_device), a, b)
    close(a, x.double().sum(0), tol=1e-6)
    assert torch.equal(a, b)


def test_adamw_matches_torch(hip_device):
    n = 1003
    p0, grads = rnd(n, seed=1), [rnd(n, seed=10 + i) for i in range(6)]
    ref_p = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([ref_p], lr=1e-4, weight_decay=5e-4, betas=(0.9, 0.98), eps=1e-9)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=3e-4, epochs=10, steps_per_epoch=5, pct_start=0.0,
                                                final_div_factor=5)
    p = p0.clone().to(hip_device)
    m, v = torch.zeros(n, device=hip_device), torch.zeros(n, device=hip_device)
    for i, g in enumerate(grads):
        lr, beta1 = opt.param_groups[0]["lr"], opt.param_groups[0]["betas"][0]
        ref_p.grad = g.clone()
        opt.step()
        sched.step()
        ops.adamw_step(p, g.to(hip_device), m, v, lr, beta1, 0.98, 1e-9, 5e-4, i + 1)
        torch.testing.assert_close(p.cpu(), ref_p.detach(), rtol=2e-6, atol=1e-7)
        torch.testing.assert_close(m.cpu(), opt.state[ref_p]["exp_avg"], rtol=2e-6, atol=1e-7)  # fma contraction: 1 ulp at |m| ~ 0.3
        torch.testing.assert_close(v.cpu(), opt.state[ref_p]["exp_avg_sq"], rtol=2e-6, atol=1e-10)


# ------------------------------------------------------------------ fp16 operand mode (reference autocast dtype)
def f16r(t):
    return t.to(torch.float16).to(torch.float64)


def test_fp16_operand_products(hip_device):
    """The *_f16 entry points (same kernels compiled for IEEE half): products equal the fp64 product of the
    fp16-rounded operands to fp32 accumulation accuracy -- GEMM NT / TN, 3x3 conv forward / weight gradient with
    both weight paths."""
    A, B = rnd(300, 384, seed=1), rnd(1536, 384, seed=2)
    with ops.matmul_bf16(True, "f16"):
        close(ops.gemm_nt(A.to(hip_device), B.to(hip_device)), f16r(A) @ f16r(B).T)
        P, Q = rnd(4000, 128, seed=3), rnd(4000, 192, seed=4)
        close(ops.gemm_tn(P.to(hip_device), Q.to(hip_device)), f16r(P).T @ f16r(Q), tol=2e-5)
        x, w = rnd(2, 64, 12, 10, seed=5), rnd(128, 64, 3, 3, seed=6, scale=0.1)
        ref = F.conv2d(f16r(x), f16r(w), padding=1)
        for frag in (True, False):
            prev, ops.CONV_WFRAG = ops.CONV_WFRAG, frag
            try:
                wf, _ = ops.conv3x3_repack(w.to(hip_device))
                assert (wf.frag is not None) == frag
                close(nchw(ops.conv3x3_fwd(nhwc(x).to(hip_device), wf)), ref)
            finally:
                ops.CONV_WFRAG = prev
        dy = rnd(2, 128, 12, 10, seed=7)
        dw = torch.empty(128, 64, 3, 3, device=hip_device)
        xr, dyr = f16r(x).requires_grad_(False), f16r(dy)
        wref = torch.zeros(128, 64, 3, 3, dtype=torch.float64, requires_grad=True)
        F.conv2d(xr, wref, padding=1).backward(dyr)
        close(ops.conv3x3_wgrad(nhwc(x).to(hip_device), nhwc(dy).to(hip_device), dw), wref.grad, tol=2e-5)
    assert ops.MATMUL_BF16 is False and ops.HALF_DTYPE == "bf16"
    # fp16 has 3 more significand bits than bf16: closer to the fp32 product
    with ops.matmul_bf16(True, "bf16"):
        e_bf = (ops.gemm_nt(A.to(hip_device), B.to(hip_device)).cpu().double() - A.double() @ B.double().T).abs().max()
    with ops.matmul_bf16(True, "f16"):
        e_f16 = (ops.gemm_nt(A.to(hip_device), B.to(hip_device)).cpu().double() - A.double() @ B.double().T).abs().max()
    assert e_f16 < 0.3 * e_bf


def test_nonfinite_flag(hip_device):
    g = rnd(100003, seed=1).to(hip_device)
    assert int(ops.nonfinite_flag(g).item()) == 0
    for pos, val in ((0, float("inf")), (77777, float("nan")), (100002, float("-inf"))):
        h = g.clone()
        h[pos] = val
        assert int(ops.nonfinite_flag(h).item()) == 1
