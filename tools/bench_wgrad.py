"""3x3 weight-gradient kernels on the JDCNet layer shapes at batch 256: time and TFLOP/s per product mode."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pitchextractor_amd import ops  # noqa: E402


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


dev = torch.device("cuda:0")
for (F, Ci, Co) in [(80, 64, 64), (40, 64, 128), (40, 128, 128), (20, 128, 192), (20, 192, 192), (10, 192, 256), (10, 256, 256)]:
    x = torch.randn(256, 192, F, Ci, device=dev)
    dy = torch.randn(256, 192, F, Co, device=dev)
    dw = torch.empty(Co, Ci, 3, 3, device=dev)
    fl = 2.0 * 256 * 192 * F * Co * 9 * Ci
    row = []
    for mode, bf in (("native", False), ("x3", False), ("native", True)):
        ops.FP32_MATMUL = mode
        with ops.matmul_bf16(bf):
            ms = timed(lambda: ops.conv3x3_wgrad(x, dy, dw))
        row.append(f"{'bf16' if bf else mode}: {ms:.3f} ms ({fl / ms / 1e9:.0f} TF)")
    print(f"wgrad F={F} Cin={Ci} Cout={Co}: " + "  ".join(row), flush=True)
