"""fp32 product modes side by side on one GPU: "x3" (three bf16 terms, six MFMAs per block) against "h2" (two scaled
fp16 terms, three MFMAs), time per call (median of interleaved rounds) and error against float64 on sampled outputs.

  python3 tools/bench_h2.py [nt|tn|conv|wgrad|whh ...]     default: all families that exist
"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pitchextractor_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
R = 256 * 192
MODES = ("x3", "h2", "native")


def timed(fns, rounds=5):
    """fns: {mode: callable}; interleaved rounds, median ms per mode."""
    ts = {m: [] for m in fns}
    for r in range(rounds + 1):
        for m, fn in fns.items():
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ops.FP32_MATMUL = m
            a.record()
            fn()
            b.record()
            torch.cuda.synchronize()
            if r:
                ts[m].append(a.elapsed_time(b))
    return {m: sorted(v)[len(v) // 2] for m, v in ts.items()}


def wide(shape, spread):
    """Random data with a log-uniform magnitude spread of 2^spread (gradient-like dynamic range)."""
    x = torch.randn(shape, device=dev)
    if spread:
        x = x * torch.exp2(-spread * torch.rand(shape, device=dev))
    return x


def err_nt(A, B, out, rows):
    ref = A[rows].double() @ B.double().t()
    den = A[rows].double().abs() @ B.double().abs().t()
    return float(((out[rows].double() - ref).abs() / den).max())


def fam_nt(modes):
    shapes = [(R, 1536, 512), (R, 1536, 768), (R, 768, 1536), (R, 512, 1536),
              (R * 40, 128, 64), (R * 20, 192, 128), (R * 10, 256, 192), (R * 2, 256, 640),
              (R * 40, 64, 128), (R * 20, 128, 192), (R * 10, 192, 256), (R * 2, 640, 256)]
    tot = {m: 0.0 for m in modes}
    rows = torch.arange(0, 4096, 37, device=dev)
    for M, N, K in shapes:
        A = wide((M, K), 12) * 1e-3
        B = wide((N, K), 6) * 0.05
        outs = {m: torch.empty(M, N, device=dev) for m in modes}
        am = (ops.absmax(A), ops.absmax(B))
        fns = {m: (lambda m=m: ops.gemm_nt(A, B, out=outs[m], amax_a=am[0], amax_b=am[1])) for m in modes}
        t = timed(fns)
        line = f"nt M={M:8d} N={N:4d} K={K:4d} "
        for m in modes:
            tot[m] += t[m]
            line += f" {m}: {t[m]:6.3f} ms {2.0 * M * N * K / t[m] / 1e9:6.1f} TF err {err_nt(A, B, outs[m], rows):.2e} |"
        print(line, flush=True)
        del A, B, outs
    print("nt total " + "  ".join(f"{m} {tot[m]:.2f} ms" for m in modes), flush=True)


def fam_tn(modes):
    shapes = [(1536, 512, R), (1536, 768, R), (128, 64, R * 40), (192, 128, R * 20), (256, 192, R * 10),
              (256, 640, R * 2)]
    tot = {m: 0.0 for m in modes}
    for M, N, K in shapes:
        A = wide((K, M), 12) * 1e-4
        B = wide((K, N), 4)
        outs = {m: torch.empty(M, N, device=dev) for m in modes}
        am = (ops.absmax(A), ops.absmax(B))
        fns = {m: (lambda m=m: ops.gemm_tn(A, B, out=outs[m], amax_a=am[0], amax_b=am[1])) for m in modes}
        t = timed(fns)
        cols = torch.arange(0, M, max(M // 16, 1), device=dev)
        ref = A[:, cols].double().t() @ B.double()
        den = A[:, cols].double().abs().t() @ B.double().abs()
        line = f"tn M={M:5d} N={N:4d} K={K:8d} "
        for m in modes:
            tot[m] += t[m]
            e = float(((outs[m][cols].double() - ref).abs() / den).max())
            line += f" {m}: {t[m]:6.3f} ms {2.0 * M * N * K / t[m] / 1e9:6.1f} TF err {e:.2e} |"
        print(line, flush=True)
        del A, B, outs
    print("tn total " + "  ".join(f"{m} {tot[m]:.2f} ms" for m in modes), flush=True)


LAYERS = [(80, 64, 64), (40, 64, 128), (40, 128, 128), (20, 128, 192), (20, 192, 192), (10, 192, 256), (10, 256, 256)]


def fam_conv(modes):
    import torch.nn.functional as Fn
    tot = {m: 0.0 for m in modes}
    for F, Ci, Co in LAYERS:
        x = wide((256, 192, F, Ci), 8)
        w = wide((Co, Ci, 3, 3), 4) * 0.05
        outs, fns = {}, {}
        for m in modes:
            ops.FP32_MATMUL = m
            pw, _ = ops.conv3x3_repack(w, want_dgrad=False)
            outs[m] = torch.empty(256, 192, F, Co, device=dev)
            ax = ops.absmax(x)
            fns[m] = (lambda m=m, pw=pw, ax=ax: ops.conv3x3_fwd(x, pw, out=outs[m], amax=ax))
        t = timed(fns)
        xs = x[:2].permute(0, 3, 1, 2).double()
        ref = Fn.conv2d(xs, w.double(), padding=1).permute(0, 2, 3, 1)
        den = Fn.conv2d(xs.abs(), w.double().abs(), padding=1).permute(0, 2, 3, 1)
        line = f"conv F={F:2d} Cin={Ci:3d} Cout={Co:3d} "
        fl = 2.0 * 256 * 192 * F * Co * 9 * Ci
        for m in modes:
            tot[m] += t[m]
            e = float(((outs[m][:2].double() - ref).abs() / den).max())
            line += f" {m}: {t[m]:6.3f} ms {fl / t[m] / 1e9:6.1f} TF err {e:.2e} |"
        print(line, flush=True)
        del x, outs
    print("conv total " + "  ".join(f"{m} {tot[m]:.2f} ms" for m in modes), flush=True)


def fam_wgrad(modes):
    import torch.nn.functional as Fn
    tot = {m: 0.0 for m in modes}
    for F, Ci, Co in LAYERS:
        x = wide((256, 192, F, Ci), 8)
        dy = wide((256, 192, F, Co), 12) * 1e-4
        outs = {m: torch.empty(Co, Ci, 3, 3, device=dev) for m in modes}
        am = (ops.absmax(x), ops.absmax(dy))
        fns = {m: (lambda m=m: ops.conv3x3_wgrad(x, dy, outs[m], amax_x=am[0], amax_dy=am[1])) for m in modes}
        t = timed(fns)
        # float64 reference on a few output channels (all pixels)
        co = [0, Co // 2, Co - 1]
        xs = x.permute(0, 3, 1, 2)
        ref = torch.zeros(len(co), Ci, 3, 3, dtype=torch.float64, device=dev)
        den = torch.zeros_like(ref)
        for b0 in range(0, 256, 32):
            xb = Fn.pad(xs[b0:b0 + 32].double(), (1, 1, 1, 1))
            db = dy[b0:b0 + 32][..., co].permute(0, 3, 1, 2).double()
            for kh in range(3):
                for kw in range(3):
                    win = xb[:, :, kh:kh + 192, kw:kw + F]
                    ref[:, :, kh, kw] += torch.einsum("bohw,bihw->oi", db, win)
                    den[:, :, kh, kw] += torch.einsum("bohw,bihw->oi", db.abs(), win.abs())
        line = f"wgrad F={F:2d} Cin={Ci:3d} Cout={Co:3d} "
        fl = 2.0 * 256 * 192 * F * Co * 9 * Ci
        for m in modes:
            tot[m] += t[m]
            e = float(((outs[m][co].double() - ref).abs() / den).max())
            line += f" {m}: {t[m]:6.3f} ms {fl / t[m] / 1e9:6.1f} TF err {e:.2e} |"
        print(line, flush=True)
        del x, dy
    print("wgrad total " + "  ".join(f"{m} {tot[m]:.2f} ms" for m in modes), flush=True)


def fam_whh(modes):
    B, T, H = 256, 192, 384
    dg = wide((B, T, 4 * H), 12) * 1e-4
    y = torch.tanh(torch.randn(B, T, 2 * H, device=dev))
    outs = {m: torch.empty(4 * H, H, device=dev) for m in modes}
    am = (ops.absmax(dg), ops.absmax(y.view(B * T, 2 * H)[:, :H]))
    fns = {m: (lambda m=m: ops.lstm_whh_grad(dg, y[:, :, :H], outs[m], False, B, T, H, amax_dg=am[0], amax_y=am[1]))
           for m in modes}
    t = timed(fns)
    hp = torch.zeros(B, T, H, dtype=torch.float64, device=dev)
    hp[:, 1:] = y[:, :-1, :H].double()
    rows = torch.arange(0, 4 * H, 97, device=dev)
    ref = torch.einsum("btg,bth->gh", dg[..., rows].double(), hp)
    den = torch.einsum("btg,bth->gh", dg[..., rows].double().abs(), hp.abs())
    line = "whh "
    for m in modes:
        e = float(((outs[m][rows].double() - ref).abs() / den).max())
        line += f" {m}: {t[m]:6.3f} ms {2.0 * B * T * 4 * H * H / t[m] / 1e9:6.1f} TF err {e:.2e} |"
    print(line, flush=True)


if __name__ == "__main__":
    fams = [a for a in sys.argv[1:] if not a.startswith("-")] or ["nt", "tn", "conv", "wgrad", "whh"]
    modes = MODES if "--native" in sys.argv else MODES[:2]
    for f in fams:
        globals()["fam_" + f](modes)
