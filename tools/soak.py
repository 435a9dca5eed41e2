"""Robustness soak: N training steps of the bench workload (B = 256, all side-stream overlaps and the persistent LSTM
kernels live), checking every step that the loss is finite and the persistent-LSTM error word stays clear; then the
same batch is run twice from the same state and the flat gradient buffers compared bit for bit.
With PE_DP_REHEARSE=1 the data-parallel wiring runs too (backend nccl = RCCL at world size 1: 8 bucketed all-reduces
and one flag reduction per step), and the device memory in use must not grow over the run."""
import logging
import os
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
if os.environ.get("PE_DP_REHEARSE") == "1":
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "3")      # as bench.py / train.py do, before the runtime loads
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
from pitchextractor_amd import distributed as pdist  # noqa: E402
from pitchextractor_amd import ops, synthetic  # noqa: E402
from pitchextractor_amd.mel import DEFAULT_MEL_PARAMS, MelSpectrogram  # noqa: E402
from pitchextractor_amd.model import JDCNet  # noqa: E402
from pitchextractor_amd.optimizers import build_optimizer  # noqa: E402
from pitchextractor_amd.trainer import Trainer  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
dev = torch.device("cuda:0")
torch.manual_seed(1)
net = JDCNet(num_class=1, sequence_model_config={"model_type": "bilstm", "hidden_size": 768, "num_layers": 4,
                                                 "dropout": 0.1, "bidirectional": True}).to(dev).train()
opt, sched = build_optimizer({"params": net.parameters(), "optimizer_params": {},
                              "scheduler_params": {"max_lr": 3e-4, "pct_start": 0.0, "epochs": 10, "steps_per_epoch": 100}})
crit = {"l1": torch.nn.SmoothL1Loss(), "ce": torch.nn.BCEWithLogitsLoss()}
dp = None
if pdist.rehearse_single_rank():
    pdist.init_from_env("nccl")
    dp = pdist.GradientAllReduce(net.flat_gradients(), opt, flat_param=net.flat_parameters,
                                 payload=os.environ.get("PE_DP_PAYLOAD"))
    net.attach_data_parallel(dp)
tr = Trainer(model=net, criterion=crit, optimizer=opt, scheduler=sched, device="cuda:0", loss_config={"lambda_f0": 0.1},
             logger=logging.getLogger("soak"), mel_transform=MelSpectrogram(**DEFAULT_MEL_PARAMS), data_parallel=dp)
waves, f0, sil = synthetic.batch(0, 32)
batch = tuple(torch.from_numpy(np.tile(a, (8, 1))).to(dev) for a in (waves, f0, sil))     # raw audio in, as bench.py
first = last = None
mem_mark = None
for i in range(steps):
    if i == 10:
        mem_mark = (torch.cuda.memory_allocated(dev), torch.cuda.memory_reserved(dev))
    out = tr.run(batch)
    assert all(map(lambda v: v == v and abs(v) < 1e6, out.values())), (i, out)
    assert not ops.persistent_lstm_error(dev), i
    first = first or out["loss"]
    last = out["loss"]
print(f"{steps} steps: loss {first:.4f} -> {last:.4f}, error word clear")
if mem_mark is not None:
    now = (torch.cuda.memory_allocated(dev), torch.cuda.memory_reserved(dev))
    print(f"device memory after step 10 / after step {steps}: allocated {mem_mark[0] >> 20} / {now[0] >> 20} MiB, "
          f"reserved {mem_mark[1] >> 20} / {now[1] >> 20} MiB")
    assert now[0] <= mem_mark[0] + (64 << 20) and now[1] <= mem_mark[1] + (256 << 20), "device memory grows per step"
if dp is not None:
    print(f"data-parallel rehearsal: {dp.messages} all-reduces over {torch.distributed.get_backend()} "
          f"({dp.messages / steps:.1f} per step, payload {dp.payload})")
# determinism of one backward with every overlap live
x, f0d, sild = tr._inputs(batch)
grads = []
for _ in range(3):
    net.dropout_cfg.offset = 0
    net.zero_grad(set_to_none=True)
    cls, det = net(x.transpose(-1, -2))          # (B,1,80,T) -> the model's (B,1,T,80), as Trainer._forward_backward
    o3, d_f0, d_sil = ops.f0_sil_loss(cls.detach().reshape(-1), f0d.reshape(-1), det.detach().reshape(-1),
                                      sild.reshape(-1), 0.1)
    torch.autograd.backward([cls, det], [d_f0.view_as(cls), d_sil.view_as(det)])
    grads.append(net.flat_gradients().clone())
assert torch.equal(grads[0], grads[1]) and torch.equal(grads[0], grads[2]), "backward is not run-to-run identical"
print("three backward passes from the same state: gradients bit-identical")
if dp is not None:
    torch.distributed.destroy_process_group()
