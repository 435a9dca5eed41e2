"""bf16 mixed precision: fp32 against bf16 activation storage per kernel family and layer (interleaved medians)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pitchextractor_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
BF = torch.bfloat16
LAYERS = [(80, 64, 64), (40, 64, 128), (40, 128, 128), (20, 128, 192), (20, 192, 192), (10, 192, 256), (10, 256, 256)]


def timed(fns, rounds=5):
    ts = {k: [] for k in fns}
    for r in range(rounds + 1):
        for k, fn in fns.items():
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            fn()
            b.record()
            torch.cuda.synchronize()
            if r:
                ts[k].append(a.elapsed_time(b))
    return {k: sorted(v)[len(v) // 2] for k, v in ts.items()}


with ops.matmul_bf16(True):
    for F, Ci, Co in LAYERS:
        x = torch.randn(256, 192, F, Ci, device=dev)
        dy = torch.randn(256, 192, F, Co, device=dev)
        w = torch.randn(Co, Ci, 3, 3, device=dev) * 0.05
        x16, dy16 = x.to(BF), dy.to(BF)
        dw = torch.empty(Co, Ci, 3, 3, device=dev)
        wf, _ = ops.conv3x3_repack(w, want_dgrad=False)
        y32, y16 = torch.empty(256, 192, F, Co, device=dev), torch.empty(256, 192, F, Co, device=dev, dtype=BF)
        t = timed({"wgrad32": lambda: ops.conv3x3_wgrad(x, dy, dw), "wgrad16": lambda: ops.conv3x3_wgrad(x16, dy16, dw),
                   "fwd32": lambda: ops.conv3x3_fwd(x, wf, out=y32), "fwd16": lambda: ops.conv3x3_fwd(x16, wf, out=y16)})
        print(f"F={F:2d} {Ci:3d}->{Co:3d}  " + "  ".join(f"{k} {v:.3f}" for k, v in t.items()), flush=True)
