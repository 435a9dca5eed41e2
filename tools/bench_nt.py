"""Error vs fp64 and launch time of the NT engines (native fp32 MFMA, bf16 operands, three-term bf16 split)
on the conv / GEMM shapes of a batch-256 step.  Usage: python tools/bench_nt.py"""
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


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    # accuracy
    A = torch.randn(1024, 2304, generator=g)
    B = torch.randn(256, 2304, generator=g)
    ref = A.double() @ B.double().T
    mag = A.double().abs() @ B.double().abs().T
    for mode, bf in (("native", False), ("x3", False), ("native", True)):
        ops.FP32_MATMUL = mode
        with ops.matmul_bf16(bf):
            got = ops.gemm_nt(A.to(dev), B.to(dev)).cpu().double()
        e = (got - ref).abs()
        print(f"accuracy {'bf16' if bf else mode:7s}: max|err|/sum|a||b| = {(e / mag).max().item():.3e}   "
              f"max|err|/max|ref| = {(e.max() / ref.abs().max()).item():.3e}", flush=True)
    # speed: the conv layers of JDCNet at batch 256 (T=192) and the big GEMMs
    convs = [(256, 192, 80, 64, 64), (256, 192, 40, 64, 128), (256, 192, 40, 128, 128), (256, 192, 20, 128, 192),
             (256, 192, 20, 192, 192), (256, 192, 10, 192, 256), (256, 192, 10, 256, 256)]
    for (Bn, T, F, C, N) in convs:
        x = torch.randn(Bn, T, F, C, device=dev)
        w = torch.randn(N, 9 * C, device=dev) * 0.05
        out = torch.empty(Bn, T, F, N, device=dev)
        fl = 2.0 * Bn * T * F * N * 9 * C
        row = []
        for mode, bf in (("native", False), ("x3", False), ("native", True)):
            ops.FP32_MATMUL = mode
            with ops.matmul_bf16(bf):
                ms = timed(lambda: ops.conv3x3_fwd(x, w, out=out))
            row.append(f"{'bf16' if bf else mode}: {ms:.3f} ms ({fl / ms / 1e9:.0f} TF)")
        print(f"conv F={F} C={C} N={N}: " + "  ".join(row), flush=True)
    for (M, N, K) in [(49152, 1536, 640), (49152, 1536, 768), (49152, 768, 1536), (49152, 640, 1536)]:
        A = torch.randn(M, K, device=dev)
        Bm = torch.randn(N, K, device=dev)
        out = torch.empty(M, N, device=dev)
        fl = 2.0 * M * N * K
        row = []
        for mode, bf in (("native", False), ("x3", False), ("native", True)):
            ops.FP32_MATMUL = mode
            with ops.matmul_bf16(bf):
                ms = timed(lambda: ops.gemm_nt(A, Bm, out=out))
            row.append(f"{'bf16' if bf else mode}: {ms:.3f} ms ({fl / ms / 1e9:.0f} TF)")
        print(f"gemm_nt M={M} N={N} K={K}: " + "  ".join(row), flush=True)
    ops.FP32_MATMUL = "native"


if __name__ == "__main__":
    main()
