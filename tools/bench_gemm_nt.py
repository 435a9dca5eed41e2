"""gemm_nt shapes of a batch-256 step (LDS-staged kernel), median of `rounds` launches each.  Run twice in one gpurun
call to compare builds / switches, e.g.  PE_GEMM_NT_SCALAR_EPILOGUE=1 python3 tools/bench_gemm_nt.py."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pitchextractor_amd import ops  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "x3"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 7
dev = torch.device("cuda:0")
ops.FP32_MATMUL = "x3"
ops.GEMM_WFRAG = False
R = 256 * 192
shapes = [(R, 1536, 512), (R, 1536, 768), (R, 768, 1536), (R, 512, 1536),                 # LSTM projections fwd / dX
          (R * 40, 128, 64), (R * 20, 192, 128), (R * 10, 256, 192), (R * 2, 256, 640),   # 1x1 convs fwd
          (R * 40, 64, 128), (R * 20, 128, 192), (R * 10, 192, 256), (R * 2, 640, 256)]   # ... and their dX
tot = 0.0
with ops.matmul_bf16(mode == "bf16"):
    for M, N, K in shapes:
        A = torch.randn(M, K, device=dev)
        B = torch.randn(N, K, device=dev) * 0.05
        bias = torch.randn(N, device=dev)
        out = torch.empty(M, N, device=dev)
        ts = []
        for r in range(rounds + 1):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            ops.gemm_nt(A, B, out=out, bias0=bias)
            b.record()
            torch.cuda.synchronize()
            if r:
                ts.append(a.elapsed_time(b))
        med = sorted(ts)[len(ts) // 2]
        tot += med
        print(f"M={M:8d} N={N:4d} K={K:4d}  {med:7.3f} ms {2.0 * M * N * K / med / 1e9:7.1f} TF", flush=True)
        del A, B, out
print(f"total {tot:.2f} ms")
