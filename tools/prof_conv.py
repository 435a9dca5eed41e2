"""One conv shape in one precision mode, a few launches: a target for rocprofv3 --pmc / --kernel-trace.
Usage: python3 tools/prof_conv.py {native|x3|bf16} [F C N]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pitchextractor_amd import ops  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "x3"
F, C, N = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (40, 128, 128)
dev = torch.device("cuda:0")
x = torch.randn(256, 192, F, C, device=dev)
w = torch.randn(N, 9 * C, device=dev) * 0.05
out = torch.empty(256, 192, F, N, device=dev)
ops.FP32_MATMUL = "x3" if mode == "x3" else "native"
with ops.matmul_bf16(mode == "bf16"):
    for _ in range(3):
        ops.conv3x3_fwd(x, w, out=out)
torch.cuda.synchronize()
print("done", mode, F, C, N)
