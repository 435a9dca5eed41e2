"""What the data-parallel wiring costs per step on ONE GPU: backend "nccl" (RCCL) at world size 1 (PE_DP_REHEARSE=1),
where every all-reduce is the identity, so any extra time is wiring: stream waits, host-side enqueue, lost overlap.
  python tools/micro/dp_overhead.py [bilstm|transformer] [fp32|bf16]
Variants: plain (no reducer) / dp (ranges in backward order, issued from the stream that produced them) / dp-one-tail
(conv stack as one range at the end of backward) / dp-reducer-strm (collectives issued from a private reducer stream) /
dp-no-flag (without the cross-rank LSTM fault flag).  Also times one all-reduce of the whole buffer alone."""
import logging
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

os.environ.setdefault("PE_DP_REHEARSE", "1")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import bench  # noqa: E402
from pitchextractor_amd import distributed as pdist  # noqa: E402
from pitchextractor_amd import synthetic  # noqa: E402
from pitchextractor_amd.mel import DEFAULT_MEL_PARAMS, MelSpectrogram  # noqa: E402
from pitchextractor_amd.model import JDCNet  # noqa: E402
from pitchextractor_amd.optimizers import build_optimizer  # noqa: E402
from pitchextractor_amd.trainer import Trainer  # noqa: E402

head = sys.argv[1] if len(sys.argv) > 1 else "bilstm"
payload = sys.argv[2] if len(sys.argv) > 2 else "fp32"
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
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


def steps(n, tag):
    tr.run(batch)
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        tr.run(batch)
        ts.append((time.perf_counter() - t0) * 1e3)
    print(f"{tag:16s}", " ".join(f"{t:.1f}" for t in ts), f"| median {np.median(ts):.2f}", flush=True)


def attach(on, cuts=True, stream=False, flag=True):
    net._dp = dp if on else None
    tr.data_parallel = dp if on else None
    net._dp_cuts = net._block_cuts() if (on and cuts) else None
    if on:
        dp._stream = reducer if stream else None
        dp.any_rank = real_any_rank if flag else (lambda f: bool(f))


for _ in range(4):
    tr.run(batch)
N = 8
steps(N, "before-init")                       # no process group, no RCCL communicator in this process yet
pdist.init_from_env("nccl")
steps(N, "pg-initialised")                    # process group exists, communicator not yet (created by the first collective)
dp = pdist.GradientAllReduce(net.flat_gradients(), opt, flat_param=net.flat_parameters, payload=payload)
reducer, real_any_rank = torch.cuda.Stream(device=dev), dp.any_rank
attach(False); steps(N, "plain")              # communicator live (the constructor broadcast), reducer detached
attach(True); steps(N, "dp")
attach(True, cuts=False); steps(N, "dp-one-tail")
attach(True, stream=True); steps(N, "dp-reducer-strm")
attach(True, flag=False); steps(N, "dp-no-flag")
attach(True); net._dp = None; steps(N, "dp-all-at-end")      # nothing during backward: finish() reduces the whole buffer
attach(False); steps(N, "plain again")
from pitchextractor_amd import model as pe_model  # noqa: E402
pe_model.OVERLAP_CONV_WGRAD = pe_model.OVERLAP_LSTM_WGRAD = pe_model.OVERLAP_TF_WGRAD = False
attach(False); steps(N, "plain serial")       # weight-gradient kernels on the compute stream (no side stream)
attach(True); steps(N, "dp serial")
pe_model.OVERLAP_CONV_WGRAD = pe_model.OVERLAP_LSTM_WGRAD = pe_model.OVERLAP_TF_WGRAD = True

# one whole-buffer all-reduce alone, 32 MB messages
g = net.flat_gradients()
for tag, t in (("fp32", g), ("bf16", g.to(torch.bfloat16))):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    per = (32 << 20) // t.element_size()
    works = [dist.all_reduce(t[a:a + per], async_op=True) for a in range(0, t.numel(), per)]
    t_host = (time.perf_counter() - t0) * 1e3
    for w in works:
        w.wait()
    torch.cuda.synchronize()
    print(f"all-reduce alone {tag}: {t.numel() * t.element_size() / 1e6:.1f} MB in {len(works)} messages: host enqueue "
          f"{t_host:.2f} ms, done after {(time.perf_counter() - t0) * 1e3:.2f} ms", flush=True)
dist.destroy_process_group()
