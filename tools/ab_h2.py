"""A/B of kernel variants selected through pe_tune_set (experiment switches), interleaved rounds in one process.
  python3 tools/ab_h2.py conv|nt  key  v0 v1 [v2 ...]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pitchextractor_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
ops.FP32_MATMUL = "h2"
fam, key, vals = sys.argv[1], int(sys.argv[2]), [int(v) for v in sys.argv[3:]]
lib = _lib.load()
R = 256 * 192


def run(cases, rounds=7):
    for name, fn, flop in cases:
        ts = {v: [] for v in vals}
        for r in range(rounds + 1):
            for v in vals:
                lib.pe_tune_set(key, v)
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                fn()
                b.record()
                torch.cuda.synchronize()
                if r:
                    ts[v].append(a.elapsed_time(b))
        lib.pe_tune_set(key, 0)
        print(name + "  " + "  ".join(f"v{v}: {sorted(t)[len(t) // 2]:.3f} ms ({flop / sorted(t)[len(t) // 2] / 1e9:.0f} TF)"
                                      for v, t in ts.items()), flush=True)


cases = []
if fam == "conv":
    for F, Ci, Co in [(80, 64, 64), (40, 64, 128), (40, 128, 128), (20, 128, 192), (20, 192, 192), (10, 192, 256), (10, 256, 256),
                      (40, 128, 64), (20, 192, 128), (10, 256, 192)]:
        x = torch.randn(256, 192, F, Ci, device=dev)
        w = torch.randn(Co, Ci, 3, 3, device=dev) * 0.05
        pw, _ = ops.conv3x3_repack(w, want_dgrad=False)
        out = torch.empty(256, 192, F, Co, device=dev)
        ax = ops.absmax(x)
        cases.append((f"conv F={F} {Ci}->{Co}", (lambda x=x, pw=pw, out=out, ax=ax: ops.conv3x3_fwd(x, pw, out=out, amax=ax)),
                      2.0 * 256 * 192 * F * Co * 9 * Ci))
else:
    for M, N, K in [(R, 1536, 512), (R, 1536, 768), (R, 768, 1536), (R, 512, 1536), (R * 20, 192, 128), (R * 10, 256, 192),
                    (R * 2, 256, 640), (R * 20, 128, 192), (R * 10, 192, 256), (R * 2, 640, 256)]:
        A = torch.randn(M, K, device=dev)
        B = torch.randn(N, K, device=dev) * 0.05
        out = torch.empty(M, N, device=dev)
        am = (ops.absmax(A), ops.absmax(B))
        cases.append((f"nt M={M} N={N} K={K}", (lambda A=A, B=B, out=out, am=am: ops.gemm_nt(A, B, out=out, amax_a=am[0], amax_b=am[1])),
                      2.0 * M * N * K))
run(cases)
