"""A/B in one process (interleaved rounds): gemm_nt shapes of a batch-256 step with the weight operand (a) staged
through LDS, (b) pre-packed as MFMA fragments and read from L2.  Usage: python3 tools/ab_gemm.py [x3|bf16] [rounds]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pitchextractor_amd import ops  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "x3"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
ops.FP32_MATMUL = "x3"
R = 256 * 192
shapes = [(R, 1536, 512), (R, 1536, 768), (R, 768, 1536), (R, 512, 1536),                 # LSTM projections fwd / dX
          (R * 40, 128, 64), (R * 20, 192, 128), (R * 10, 256, 192), (R * 2, 256, 640),   # 1x1 convs fwd
          (R * 40, 64, 128), (R * 20, 128, 192), (R * 10, 192, 256), (R * 2, 640, 256),   # ... and their dX
          (R, 1536, 1536), (R, 512, 1536)]                                                # transformer-ish
tot = {True: 0.0, False: 0.0}
with ops.matmul_bf16(mode == "bf16"):
    for M, N, K in shapes:
        A = torch.randn(M, K, device=dev)
        B = torch.randn(N, K, device=dev) * 0.05
        out = torch.empty(M, N, device=dev)
        times = {True: [], False: []}
        for r in range(rounds + 1):
            for frag in (True, False):
                ops.GEMM_WFRAG = frag
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                ops.gemm_nt(A, B, out=out)                       # (frag: includes the per-call weight pack)
                b.record()
                torch.cuda.synchronize()
                if r:
                    times[frag].append(a.elapsed_time(b))
        fl = 2.0 * M * N * K
        med = {k: sorted(v)[len(v) // 2] for k, v in times.items()}
        for k in med:
            tot[k] += med[k]
        print(f"M={M:8d} N={N:4d} K={K:4d}  lds {med[False]:7.3f} ms {fl / med[False] / 1e9:7.1f} TF | "
              f"frag {med[True]:7.3f} ms {fl / med[True] / 1e9:7.1f} TF | x{med[False] / med[True]:.3f}", flush=True)
        del A, B, out
print(f"total lds {tot[False]:.2f} ms  frag {tot[True]:.2f} ms  x{tot[False] / tot[True]:.3f}")
