"""Where does an item of the overlapped persistent LSTM forward spend its cycles?  The stamped instantiation of the
kernel (pe_lstm_configure_stamps) on 2 cells (grid 96 <= 128: the stamp area of the sync buffer), B=256, T=192, H=384; prints per-region
cycles per item (median over workgroups, wave 0)."""
import os
import sys
from pathlib import Path

import torch  # noqa: E402

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pitchextractor_amd import _lib, ops  # noqa: E402

ops.FP32_MATMUL = "x3"
_lib.load().pe_lstm_configure_stamps(1)
if os.environ.get("PE_STAMP_BF16") == "1":      # the mixed-precision build (one bf16 term, bf16 tile exchange)
    ops.matmul_bf16(True, "bf16").__enter__()

dev = torch.device("cuda:0")
B, T, H, NC = 256, 192, 384, 2
whh = [torch.randn(4 * H, H, device=dev) * 0.05 for _ in range(NC)]
ys = [torch.empty(B, T, 2 * H, device=dev) for _ in range(1)]
gates = [torch.randn(B, T, 4 * H, device=dev) for _ in range(NC)]
cb = [torch.empty(B, T, H, device=dev) for _ in range(NC)]
ysl = [ys[0][:, :, (i % 2) * H:(i % 2 + 1) * H] for i in range(NC)]
for _ in range(2):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.lstm_fwd(whh, [g.clone() for g in gates], ysl, cb, [0, 1][:NC], B, T, H)
    e1.record()
    torch.cuda.synchronize()
print("ms/layer", e0.elapsed_time(e1), "us/step", e0.elapsed_time(e1) / T * 1e3)
sync = ops._SYNC[dev].cpu()
grid = NC * (B // 64) * (H // 32)
st = sync[2048:2048 + 16 * grid].view(grid, 16)[:, :10].double() * 16 / (2 * (T - 1))
names = ["R1 issue (MFMA h1 + epilogue)", "store drain vmcnt(5)", "barrier #2", "R2 issue (MFMA h2)", "poll",
         "barrier #3", "R3 (fetch issue, last block, acc->red)", "barrier #0", "commit rows (wait fetch)", "barrier #1"]
tot = 0.0
for k, n in enumerate(names):
    v = st[:, k].median().item()
    tot += v
    print(f"{n:42s} {v:8.0f} cycles/item   (min {st[:, k].min().item():.0f} max {st[:, k].max().item():.0f})")
print(f"{'sum':42s} {tot:8.0f} cycles/item")
print("error flag:", ops.persistent_lstm_error(dev))

# ---- backward (exchange-buffer kernel): regions of an item
whh_t = [ops.transpose2d(w) for w in whh]
for _ in range(2):
    g2 = [torch.rand(B, T, 4 * H, device=dev) for _ in range(NC)]
    cb2 = [torch.randn(B, T, H, device=dev) for _ in range(NC)]
    dys = torch.randn(B, T, 2 * H, device=dev)
    dsl = [dys[:, :, (i % 2) * H:(i % 2 + 1) * H] for i in range(NC)]
    dc = [torch.empty(B, H, device=dev) for _ in range(NC)]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.lstm_bwd(whh_t, g2, cb2, dsl, dc, [0, 1][:NC], B, T, H)
    e1.record()
    torch.cuda.synchronize()
print("bwd ms/layer(2 cells)", e0.elapsed_time(e1), "us/step", e0.elapsed_time(e1) / T * 1e3)
sync = ops._SYNC[dev].cpu()
st = sync[2048:2048 + 16 * grid].view(grid, 16)[:, :10].double() * 16 / (2 * (T - 1))
names = ["product, first half (+ input / tile-store issue)", "poll (per wave)",
         "product, second half (+ tile fetch / store issue)", "gate-gradient update, then tile store drain", "arrive",
         "barrier"]
tot = 0.0
for k, n in enumerate(names):
    v = st[:, k].median().item()
    tot += v
    print(f"{n:42s} {v:8.0f} cycles/item   (min {st[:, k].min().item():.0f} max {st[:, k].max().item():.0f})")
print(f"{'sum':42s} {tot:8.0f} cycles/item")
print("error flag:", ops.persistent_lstm_error(dev))

# absolute times (100 MHz s_memrealtime) of step 100, half 0: flag store, wait entered, wait left -- per group
raw = sync[2048:2048 + 16 * grid].view(grid, 16).long()
NJ = H // 32
ngroups = grid // NJ
blocks = torch.arange(grid)
gidx = torch.where(torch.tensor((ngroups & 7) == 0), (blocks & 7) * (ngroups >> 3) + (blocks >> 3) // NJ, blocks // NJ)
for g in range(min(ngroups, 4)):
    m = gidx == g
    arr, w0, w1 = raw[m, 13], raw[m, 11], raw[m, 12]
    base = arr.min()
    print(f"group {g}: flag stores at {sorted((arr - base).tolist())} x10ns; wait entered {sorted((w0 - base).tolist())}; left {sorted((w1 - base).tolist())}")
