"""Diagnostic: per-parameter gradient error of the HIP step at B=8 and at B=256 (8 samples tiled x32) against the
float64 oracle, both fp32 product modes; and of the Transformer head under mixed precision vs the golden norms."""
import sys
from pathlib import Path
import numpy as np
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from oracle import model_ref
from pitchextractor_amd import ops
from pitchextractor_amd.model import JDCNet
from tests.golden.make_golden import SEQ_CFG, TF_CFG, golden_input, golden_targets
from tests.test_model_gpu import _f64_oracle_grads, build

dev = torch.device("cuda:0")
torch.set_num_threads(16)
state = model_ref.seeded_state(11)
x8 = golden_input(9, B=8); f0, sil = golden_targets(9, B=8)
rc, rd, rl, rg = _f64_oracle_grads(state, dict(SEQ_CFG), x8, f0, sil)


def hip(reps):
    net = build(state, 1, 384, dev).train(); net.block_dropout = 0.0
    x = x8.repeat(reps, 1, 1, 1).to(dev)
    f0b, silb = f0.repeat(reps, 1).to(dev), sil.repeat(reps, 1).to(dev)
    cls, det = net(x)
    out3, d_f0, d_sil = ops.f0_sil_loss(cls.detach().reshape(-1), f0b.reshape(-1), det.detach().reshape(-1), silb.reshape(-1), 0.1)
    torch.autograd.backward([cls, det], [d_f0.view_as(cls), d_sil.view_as(det)])
    return {n: p.grad.detach().cpu().double() for n, p in net.named_parameters()}, out3[0].item()

for mode in ("x3", "native"):
    ops.FP32_MATMUL = mode
    g8, l8 = hip(1); g256, l256 = hip(32)
    print(f"== {mode}: loss oracle {rl:.8f} B8 {l8:.8f} B256 {l256:.8f}")
    worst = []
    for n in g8:
        ref = rg[n]; m = ref.abs().max().item()
        e8 = (g8[n] - ref).abs().max().item() / m; e256 = (g256[n] - ref).abs().max().item() / m
        n8 = abs(g8[n].norm().item() - ref.norm().item()) / ref.norm().item()
        n256 = abs(g256[n].norm().item() - ref.norm().item()) / ref.norm().item()
        worst.append((e256, e8, n256, n8, n))
    worst.sort(reverse=True)
    for e256, e8, n256, n8, n in worst[:12]:
        print(f"  {n:50s} elem256 {e256:.2e} elem8 {e8:.2e} norm256 {n256:.2e} norm8 {n8:.2e}")
    print("  max elem256 %.2e max elem8 %.2e max norm256 %.2e max norm8 %.2e" % (
        max(w[0] for w in worst), max(w[1] for w in worst), max(w[2] for w in worst), max(w[3] for w in worst)))
ops.FP32_MATMUL = "x3"

G = np.load(ROOT / "tests" / "golden" / "model_golden.npz")
st = model_ref.seeded_state(11, model_type="transformer")
for amp in (False, True):
    net = JDCNet(num_class=1, sequence_model_config=dict(TF_CFG)); net.load_state_dict(st); net = net.to(dev).train(); net.block_dropout = 0.0
    f0t, silt = (t.to(dev) for t in golden_targets(3))
    with ops.matmul_bf16(amp):
        cls, det = net(golden_input(3).to(dev))
        out3, d_f0, d_sil = ops.f0_sil_loss(cls.detach().reshape(-1), f0t.reshape(-1), det.detach().reshape(-1), silt.reshape(-1), 0.1)
        torch.autograd.backward([cls, det], [d_f0.view_as(cls), d_sil.view_as(det)])
    norms = dict(zip([str(n) for n in G["tf_f64_grad_names"]], G["tf_f64_grad_norms"]))
    rel = sorted(((abs(p.grad.double().norm().item() - norms[n]) / norms[n], n) for n, p in net.named_parameters()), reverse=True)
    sc = lambda a, b: np.abs(a.detach().cpu().double().numpy() - b).max() / np.abs(b).max()
    print(f"== transformer amp={amp}: cls {sc(cls, G['tf_f64_train_cls']):.2e} det {sc(det, G['tf_f64_train_det']):.2e} loss {out3[0].item():.6f} vs {G['tf_f64_loss'][0]:.6f}")
    for r, n in rel[:8]:
        print(f"  {n:60s} {r:.2e}")
