"""A/B in one process (interleaved rounds): conv3x3 forward / data-gradient shapes of a batch-256 step with the
weight operands (a) staged through LDS per tap, (b) pre-packed as MFMA fragments and loaded from L2.
Usage: python3 tools/ab_conv.py [x3|bf16] [rounds]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pitchextractor_amd import ops  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "x3"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
ops.FP32_MATMUL = "x3"
# (F, C, N): forward convs and the data gradients of JDCNet at B=256, T=192
shapes = [(80, 64, 64), (40, 64, 128), (40, 128, 128), (20, 128, 192), (20, 192, 192), (10, 192, 256), (10, 256, 256),
          (40, 128, 64), (20, 192, 128), (10, 256, 192)]
tot = {True: 0.0, False: 0.0}
with ops.matmul_bf16(mode == "bf16"):
    for F, C, N in shapes:
        x = torch.randn(256, 192, F, C, device=dev)
        w = torch.randn(N, C, 3, 3, device=dev) * 0.05
        out = torch.empty(256, 192, F, N, device=dev)
        packed = {}
        for frag in (True, False):
            ops.CONV_WFRAG = frag
            packed[frag] = ops.conv3x3_repack(w, True, False)[0]
        times = {True: [], False: []}
        for r in range(rounds + 1):
            for frag in (True, False):
                ops.CONV_WFRAG = frag
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                ops.conv3x3_fwd(x, packed[frag], out=out)
                b.record()
                torch.cuda.synchronize()
                if r:
                    times[frag].append(a.elapsed_time(b))
        fl = 2.0 * 256 * 192 * F * N * 9 * C
        med = {k: sorted(v)[len(v) // 2] for k, v in times.items()}
        for k in med:
            tot[k] += med[k]
        print(f"F={F:3d} C={C:3d} N={N:3d}  lds {med[False]:7.3f} ms {fl / med[False] / 1e9:7.1f} TF | "
              f"frag {med[True]:7.3f} ms {fl / med[True] / 1e9:7.1f} TF | x{med[False] / med[True]:.3f}", flush=True)
        del x, w, out
print(f"total lds {tot[False]:.2f} ms  frag {tot[True]:.2f} ms  x{tot[False] / tot[True]:.3f}")
