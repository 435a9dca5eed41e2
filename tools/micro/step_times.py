"""Per-step wall times of the bench step (every Trainer.run ends in a device->host copy of the loss, so each step is
synchronous) with and without the bench's event pairs around the conv launches.
  python tools/micro/step_times.py [transformer|bilstm]"""
import logging
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import bench  # noqa: E402
from pitchextractor_amd import ops, synthetic  # noqa: E402
from pitchextractor_amd.mel import DEFAULT_MEL_PARAMS, MelSpectrogram  # noqa: E402
from pitchextractor_amd.model import JDCNet  # noqa: E402
from pitchextractor_amd.optimizers import build_optimizer  # noqa: E402
from pitchextractor_amd.trainer import Trainer  # noqa: E402

head = sys.argv[1] if len(sys.argv) > 1 else "transformer"
dev = torch.device("cuda:0")
torch.manual_seed(1234)
net = JDCNet(num_class=1, sequence_model_config=dict(bench.SEQ_CFG, model_type=head)).to(dev).train()
opt, sched = build_optimizer({"params": net.parameters(), "optimizer_params": {},
                              "scheduler_params": {"max_lr": 3e-4, "pct_start": 0.0, "epochs": 100,
                                                   "steps_per_epoch": 1000}})
tr = Trainer(model=net, criterion={"l1": torch.nn.SmoothL1Loss(), "ce": torch.nn.BCEWithLogitsLoss()}, optimizer=opt,
             scheduler=sched, device=str(dev), loss_config={"lambda_f0": 0.1}, logger=logging.getLogger("t"),
             mel_transform=MelSpectrogram(**DEFAULT_MEL_PARAMS))
w32, f32, s32 = synthetic.batch(0, 32)
batch = tuple(torch.from_numpy(np.tile(a, (8, 1))).to(dev) for a in (w32, f32, s32))


import gc  # noqa: E402
_gc_t0, gc_log = [0.0], []


def _gc_cb(phase, info):
    if phase == "start":
        _gc_t0[0] = time.perf_counter()
    else:
        gc_log.append((info["generation"], (time.perf_counter() - _gc_t0[0]) * 1e3, info["collected"]))


gc.callbacks.append(_gc_cb)


def steps(n, tag):
    ts = []
    gc_log.clear()
    for _ in range(n):
        t0 = time.perf_counter()
        tr.run(batch)
        ts.append((time.perf_counter() - t0) * 1e3)
    print(tag, " ".join(f"{t:.1f}" for t in ts), f"| mean {np.mean(ts):.2f}", flush=True)
    slow = [(g, round(ms, 1), c) for g, ms, c in gc_log if ms > 1.0]
    print(f"           gc: {len(gc_log)} collections, {sum(ms for _, ms, _ in gc_log):.1f} ms in all; > 1 ms: {slow}",
          flush=True)


steps(6, "warm      ")
steps(25, "no timer  ")
ops.TIMER = ops.KernelTimer({"pe_conv3x3_fwd_x3", "pe_conv3x3_fwd_wf_x3", "pe_mel_forward"})
steps(25, "conv timer")
ops.TIMER = None
steps(12, "no timer  ")
alloc = torch.cuda.memory_stats(dev)
print("allocator: reserved GB", alloc["reserved_bytes.all.current"] / 1e9, "num_alloc_retries", alloc["num_alloc_retries"],
      "segments", alloc["segment.all.current"], "cudaMalloc calls", alloc.get("num_device_alloc", -1))
