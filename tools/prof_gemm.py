"""One gemm_nt / conv shape, a few launches, both weight-operand paths: a target for rocprofv3 --pmc.
Usage: python3 tools/prof_gemm.py gemm M N K | conv F C N"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pitchextractor_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
ops.FP32_MATMUL = "x3"
kind = sys.argv[1]
if kind == "gemm":
    M, N, K = (int(v) for v in sys.argv[2:5])
    A = torch.randn(M, K, device=dev)
    B = torch.randn(N, K, device=dev) * 0.05
    out = torch.empty(M, N, device=dev)
    for frag in (False, True):
        ops.GEMM_WFRAG = frag
        for _ in range(3):
            ops.gemm_nt(A, B, out=out)
else:
    F, C, N = (int(v) for v in sys.argv[2:5])
    x = torch.randn(256, 192, F, C, device=dev)
    w = torch.randn(N, C, 3, 3, device=dev) * 0.05
    out = torch.empty(256, 192, F, N, device=dev)
    for frag in (False, True):
        ops.CONV_WFRAG = frag
        wf = ops.conv3x3_repack(w, True, False)[0]
        for _ in range(3):
            ops.conv3x3_fwd(x, wf, out=out)
torch.cuda.synchronize()
print("done")
