"""LayerNorm backward at the Transformer head's size (49152 x 512): plain and fused (residual add + dropout) passes.
  PE_LN_BWD_BLOCKS=N python tools/micro/ln_bwd.py"""
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from pitchextractor_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
R, D, p = 256 * 192, 512, 0.1
a, b = torch.randn(R, D, device=dev), torch.randn(R, D, device=dev)
gam, bet = torch.randn(D, device=dev), torch.randn(D, device=dev)
y, st, m = ops.layernorm_dropout_fwd(a, b, gam, bet, p, seed=1, offset=0)
dy, dy2 = torch.randn(R, D, device=dev), torch.randn(R, D, device=dev)
dg, db = torch.empty(D, device=dev), torch.empty(D, device=dev)


def timed(fn, n=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


t_plain = timed(lambda: ops.layernorm_bwd(dy, st, gam, dg, db))
t_fused = timed(lambda: ops.layernorm_bwd(dy, st, gam, dg, db, dy_add=dy2, drop_mask=m, p=p))
mb = R * D * 4 / 1e6
print(f"blocks {os.environ.get('PE_LN_BWD_BLOCKS', '256')}: plain {t_plain:.1f} us ({3 * mb / t_plain:.2f} TB/s)  "
      f"fused {t_fused:.1f} us ({5.25 * mb / t_fused:.2f} TB/s)")
