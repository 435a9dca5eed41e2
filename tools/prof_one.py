"""One kernel family at one shape, a few launches: a target for rocprofv3 --pmc (tools/prof_kernel.sh).
Usage: python3 tools/prof_one.py nt M N K | tn M N K | conv F C N | wgrad F Ci Co   [mode]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pitchextractor_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
kind = sys.argv[1]
a, b, c = (int(v) for v in sys.argv[2:5])
ops.FP32_MATMUL = sys.argv[5] if len(sys.argv) > 5 else "h2"
if kind == "nt":
    A, B = torch.randn(a, c, device=dev), torch.randn(b, c, device=dev) * 0.05
    out = torch.empty(a, b, device=dev)
    am = (ops.absmax(A), ops.absmax(B)) if ops.h2_active() else (None, None)
    for _ in range(4):
        ops.gemm_nt(A, B, out=out, amax_a=am[0], amax_b=am[1])
elif kind == "tn":
    A, B = torch.randn(c, a, device=dev), torch.randn(c, b, device=dev)
    out = torch.empty(a, b, device=dev)
    for _ in range(4):
        ops.gemm_tn(A, B, out=out)
elif kind == "conv":
    x = torch.randn(256, 192, a, b, device=dev)
    w = torch.randn(c, b, 3, 3, device=dev) * 0.05
    out = torch.empty(256, 192, a, c, device=dev)
    wf = ops.conv3x3_repack(w, True, False)[0]
    ax = ops.amax_for(x)
    for _ in range(4):
        ops.conv3x3_fwd(x, wf, out=out, amax=ax)
else:
    x = torch.randn(256, 192, a, b, device=dev)
    dy = torch.randn(256, 192, a, c, device=dev)
    dw = torch.empty(c, b, 3, 3, device=dev)
    for _ in range(4):
        ops.conv3x3_wgrad(x, dy, dw)
torch.cuda.synchronize()
print("done")
