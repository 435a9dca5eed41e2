"""Time one LSTM layer recurrence (4 cells, B=256, T=192, H=384), forward and backward, persistent vs one launch
per step.  PE_FP32_MATMUL=x3|native selects the recurrent-product form of the persistent kernels."""
import os, sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pitchextractor_amd import ops
dev = torch.device("cuda:0")
if os.environ.get("PE_BENCH_BF16") == "1":      # the mixed-precision build of the recurrences
    ops.matmul_bf16(True, "bf16").__enter__()
B, T, H = 256, 192, 384
whh = [torch.randn(4 * H, H, device=dev) * 0.05 for _ in range(4)]
ys = [torch.empty(B, T, 2 * H, device=dev) for _ in range(2)]
def run(persistent):
    ops.USE_PERSISTENT_LSTM = persistent
    gates = [torch.randn(B, T, 4 * H, device=dev) for _ in range(4)]
    cb = [torch.empty(B, T, H, device=dev) for _ in range(4)]
    ysl = [ys[i // 2][:, :, (i % 2) * H:(i % 2 + 1) * H] for i in range(4)]
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.lstm_fwd(whh, gates, ysl, cb, [0, 1, 0, 1], B, T, H)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)
for p in (True, False):
    run(p)
    ts = [run(p) for _ in range(3)]
    print("persistent" if p else "stepwise", "products", ops.FP32_MATMUL, "ms/layer", min(ts), "us/step", min(ts) / T * 1e3, flush=True)

# backward recurrence
def run_bwd(persistent):
    ops.USE_PERSISTENT_LSTM = persistent
    gates = [torch.rand(B, T, 4 * H, device=dev) for _ in range(4)]
    cb = [torch.randn(B, T, H, device=dev) for _ in range(4)]
    dys = [torch.randn(B, T, 2 * H, device=dev) for _ in range(2)]
    dsl = [dys[i // 2][:, :, (i % 2) * H:(i % 2 + 1) * H] for i in range(4)]
    whh_t = [ops.transpose2d(w) for w in whh]
    dc = [torch.empty(B, H, device=dev) for _ in range(4)]
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.lstm_bwd(whh_t, gates, cb, dsl, dc, [0, 1, 0, 1], B, T, H)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)
for p in (True, False):
    run_bwd(p)
    ts = [run_bwd(p) for _ in range(3)]
    print("bwd persistent" if p else "bwd stepwise", "ms/layer", min(ts), "us/step", min(ts) / T * 1e3, flush=True)
print("error flag:", ops.persistent_lstm_error(dev))
