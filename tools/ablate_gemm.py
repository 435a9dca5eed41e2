"""Where does a k-tile of the fragment-fed GEMM spend its time?  Interleaved rounds of the 128x192 x3 kernel with
parts of its loop removed (timing only).  Usage: python3 tools/ablate_gemm.py [M N K]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pitchextractor_amd import _lib, ops  # noqa: E402

M, N, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (49152, 1536, 768)
dev = torch.device("cuda:0")
A = torch.randn(M, K, device=dev)
B = torch.randn(N, K, device=dev) * 0.05
out = torch.empty(M, N, device=dev)
wf = ops.wfrag_pack(B, 3)
lib = _lib.load()
masks = {0: "full", 1: "-B loads", 2: "-A split/store", 6: "-A split/store/global", 8: "-epilogue", 16: "-barrier",
         22: "-A staging -barrier", 32: "-A LDS reads", 55: "MFMA + epilogue only", 63: "MFMA only"}
times = {m: [] for m in masks}
for r in range(8):
    for m in masks:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        _lib.check(lib.pe_gemm_nt_wf_ablate(m, A.data_ptr(), K, wf.data_ptr(), out.data_ptr(), N, M, N, K,
                                            _lib.stream_ptr()), "ablate")
        b.record()
        torch.cuda.synchronize()
        if r:
            times[m].append(a.elapsed_time(b))
fl = 2.0 * M * N * K
for m, name in masks.items():
    t = sorted(times[m])[len(times[m]) // 2]
    print(f"mask {m:2d} {name:28s} {t:7.3f} ms  {fl / t / 1e9:7.1f} TF")
