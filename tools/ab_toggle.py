"""In-process A/B of a boolean switch of pitchextractor_amd.model on the bench step (B = 256): alternating blocks of
steps with the flag off / on in ONE process on ONE box, so neither box-to-box spread nor clock drift between
processes enters the comparison.

  python tools/ab_toggle.py TF_FUSE_DROPOUT --head transformer [--precision bf16] [--rounds 6] [--block 4]
"""
import argparse
import logging
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("flag")
    ap.add_argument("--head", default="bilstm")
    ap.add_argument("--precision", default="fp32")
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--block", type=int, default=4)
    ap.add_argument("--batch", type=int, default=256)
    args = ap.parse_args()
    from pitchextractor_amd import model as pe_model, synthetic
    from pitchextractor_amd.mel import DEFAULT_MEL_PARAMS, MelSpectrogram
    from pitchextractor_amd.model import JDCNet
    from pitchextractor_amd.optimizers import build_optimizer
    from pitchextractor_amd.trainer import Trainer
    assert isinstance(getattr(pe_model, args.flag), bool), args.flag
    dev = torch.device("cuda:0")
    torch.manual_seed(1234)
    net = JDCNet(num_class=1, sequence_model_config=dict(bench.SEQ_CFG, model_type=args.head)).to(dev).train()
    opt, sched = build_optimizer({"params": net.parameters(), "optimizer_params": {},
                                  "scheduler_params": {"max_lr": 3e-4, "pct_start": 0.0, "epochs": 100,
                                                       "steps_per_epoch": 1000}})
    crit = {"l1": torch.nn.SmoothL1Loss(), "ce": torch.nn.BCEWithLogitsLoss()}
    tr = Trainer(model=net, criterion=crit, optimizer=opt, scheduler=sched, device=str(dev),
                 loss_config={"lambda_f0": 0.1}, logger=logging.getLogger("ab"),
                 mel_transform=MelSpectrogram(**DEFAULT_MEL_PARAMS), use_mixed_precision=args.precision == "bf16")
    w32, f32, s32 = synthetic.batch(0, 32)
    reps = (args.batch + 31) // 32
    batch = tuple(torch.from_numpy(np.tile(a, (reps, 1))[:args.batch]).to(dev) for a in (w32, f32, s32))
    times = {False: [], True: []}
    for on in (False, True, False, True):            # warm both variants (allocator, workspaces, clocks)
        setattr(pe_model, args.flag, on)
        for _ in range(2):
            tr.run(batch)
    for _ in range(args.rounds):
        for on in (False, True):
            setattr(pe_model, args.flag, on)
            tr.run(batch)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(args.block):
                tr.run(batch)
            torch.cuda.synchronize(dev)
            times[on].append((time.perf_counter() - t0) / args.block * 1e3)
    for on in (False, True):
        t = np.array(times[on])
        print(f"{args.flag}={on}: mean {t.mean():.2f} ms/step, min {t.min():.2f}, max {t.max():.2f}  "
              f"({args.head}, {args.precision}, {args.rounds} x {args.block} steps)", flush=True)


if __name__ == "__main__":
    main()
