#!/usr/bin/env python3
"""Headline benchmark: mel-frames/sec training throughput (JDCNet + BiLSTM, 24 kHz, batch 256/GPU).

One step = the whole hot path on one minibatch of synthetic 2 s utterances whose raw audio is
already resident in HBM: fused mel front end -> JDCNet forward -> SmoothL1+BCE loss -> backward ->
(N > 1: RCCL gradient all-reduce) -> fused AdamW -> OneCycle.  Nothing is skipped or cached.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints one JSON line.  ``value`` is the HBM-resident rate (inputs in HBM when the timed region
starts; the PCIe-inclusive rate is the side field ``from_host``).  ``roofline`` prices the dominant kernel family (the
3x3 convolutions: forward + data-gradient launches of ``conv3x3_halo_wf_kernel``) by algorithmic FLOPs over HIP-event
time measured inside the timed region, and carries ``whole_step`` = all of the step's algorithmic FLOPs over the step time; ``roofline_mel`` does the same for the fused mel front end against HBM bandwidth
(1 520 B per frame).  After the timed region (N = 1 only, never part of ``value``):
``kernel_families`` = HIP events around every C-ABI call for a few extra steps (ms/step, share of the step,
achieved rate and fraction of the matching peak, all computed here); ``from_host`` = the same step fed from
pinned host memory with the next batch's H2D in flight on a side stream (SURVEY 8d's step definition);
``fp32_native_mfma`` = the step on the native fp32 MFMA instructions; ``cpu_baseline`` = the CPU oracle
trainer on BASELINE config[0] (B = 4, 10 steps after 2 warm-ups).
"""
import argparse
import json
import logging
import os
import sys
import time
from pathlib import Path

# Data-parallel runs: cap the HIP runtime's hardware queues at 3 (compute, weight-gradient side stream, RCCL) BEFORE
# the runtime loads.  With the default of 4, every step that touches RCCL's stream measured +5 ms on one MI355X
# (+11 ms without the side stream) even with identity collectives; 3 brings it to +0.3 ms
# (tools/micro/dp_overhead.py, profiles/dp_overhead_r02.txt).
if int(os.environ.get("WORLD_SIZE", "1")) > 1 or os.environ.get("PE_DP_REHEARSE") == "1":
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "3")

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

FRAMES = 192
SEQ_CFG = {"model_type": "bilstm", "num_layers": 4, "dropout": 0.1, "nhead": 8, "dim_feedforward": 1536,
           "max_len": 2048}                     # Configs/config.yml:18-24 (hidden_size defaults to 384)
MFMA_BF16_PEAK_TFLOPS = 2500.0                  # MI355X_MICROARCH.md: dense bf16 MFMA peak
MFMA_F32_PEAK_TFLOPS = 157.3                    # MI355X_MICROARCH.md: exact-f32 MFMA = vector rate
HBM_PEAK_GBPS = 8000.0


def host_cores():
    """CPU cores this process may actually use (affinity mask and cgroup quota, not the host total)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 32))


PMC_SUMMARY = "profiles/pmc_r03_summary.json"


def pmc_traffic(conv_key="pe_conv3x3_fwd"):
    """(HBM-side bytes per launch, MFMA-busy share) of the roofline kernel family from the committed rocprofv3 --pmc
    passes of this same command (PMC_SUMMARY, built by tools/prof_step.sh + tools/pmc_build.py: FETCH_SIZE /
    WRITE_SIZE / SQ counters in separate passes, FETCH doubled per the gfx950 correction).  Launch-weighted over the
    family's tile variants; (None, None) when the profile holds no entry for this mode.  These two numbers are read from
    the committed profile, NOT measured in this run: the line carries `traffic_source` next to them."""
    import re
    mode = {"pe_conv3x3_fwd_x3": 2, "pe_conv3x3_fwd_wf_x3": 2, "pe_conv3x3_fwd_h2": 3, "pe_conv3x3_fwd_wf_h2": 3,
            "pe_conv3x3_fwd_bf16": 1, "pe_conv3x3_fwd_wf_bf16": 1}.get(conv_key)
    try:
        d = json.loads((ROOT / PMC_SUMMARY).read_text())["kernels"]
        rows = [v for k, v in d.items() if mode is not None and re.match(r"conv3x3_halo_wf_kernel<\d+, %d, " % mode, k)
                and "hbm_bytes_per_launch" in v and v.get("calls")]
        n = sum(v["calls"] for v in rows)
        return (sum(v["hbm_bytes_per_launch"] * v["calls"] for v in rows) / n,
                sum(v["mfma_busy"] * v["calls"] for v in rows) / n)
    except Exception:
        return None, None


def pmc_source():
    """`<file>@<commit that last touched it>` for the profile the traffic numbers come from."""
    import subprocess
    try:
        rev = subprocess.run(["git", "log", "-1", "--format=%h", "--", PMC_SUMMARY], cwd=str(ROOT), capture_output=True,
                             text=True, timeout=10).stdout.strip()
    except Exception:
        rev = ""
    return PMC_SUMMARY + ("@" + rev if rev else "")


def cpu_baseline(n_steps=10, batch=4, n_warm=2):
    """Reference-equivalent fp32 CPU trainer (oracle port) on a bounded sample, host cores stated."""
    from oracle import mel_ref, model_ref, train_ref
    from pitchextractor_amd import synthetic
    cores = host_cores()
    torch.set_num_threads(cores)
    print(f"[bench] cpu_baseline on {cores} threads ...", file=sys.stderr, flush=True)
    state = model_ref.seeded_state(21)
    cfg = dict(SEQ_CFG, dropout=0.0)
    tr = train_ref.CpuTrainer(state, cfg, fused_lstm=True)   # stock fused LSTM op, as nn.LSTM uses
    waves, f0, sil = synthetic.batch(0, batch)
    t_mel0 = time.perf_counter()
    mels = np.zeros((batch, 1, 80, FRAMES), np.float32)
    for i in range(batch):
        lm = mel_ref.log_mel(waves[i]).astype(np.float32)
        mels[i, 0, :, :lm.shape[1]] = lm
    t_mel = time.perf_counter() - t_mel0
    b = (torch.from_numpy(mels), torch.from_numpy(f0), torch.from_numpy(sil))
    warm = 0.0
    for _ in range(n_warm):                     # 2 warm-up steps, then 10 timed
        t0 = time.perf_counter()
        tr.run(b)
        warm = time.perf_counter() - t0
    if warm * n_steps > 60.0:                   # keep the whole baseline bounded on a slow host
        n_steps = max(1, int(60.0 / warm))
    times = []
    for _ in range(n_steps):
        t0 = time.perf_counter()
        tr.run(b)
        times.append(time.perf_counter() - t0)
    step = float(np.median(times))
    return {"value": batch * FRAMES / (step + t_mel), "unit": "mel-frames/s", "cores": cores, "kind": "port",
            "model_step_only": batch * FRAMES / step,
            "sample": f"BASELINE config[0]: {n_steps} steps after {n_warm} warm-ups of batch {batch} x {FRAMES} frames, "
                      f"default JDCNet+BiLSTM fp32, incl. float64 numpy mel ({t_mel * 1e3:.0f} ms/batch); "
                      f"median step {step:.2f} s"}


def _family_table(summ, steps, step_ms):
    """Per C-ABI entry point: ms/step, share of the step, achieved rate and fraction of the matching peak --
    all from HIP-event sums of this run (MFMA families: algorithmic 2*M*N*K FLOPs; mel: 1 520 B/frame)."""
    rows = {}
    for name, v in sorted(summ.items(), key=lambda kv: -kv[1]["total_ms"]):
        ms = v["total_ms"] / steps
        row = {"ms_per_step": round(ms, 4), "share_of_step": round(ms / step_ms, 4), "calls_per_step": v["calls"] / steps}
        if v["work"]:
            rate = v["work"] / (v["total_ms"] * 1e-3)
            if name.startswith("pe_mel") or name.startswith("pe_resample"):
                row.update(bound="hbm", achieved_gbps=rate / 1e9, frac_of_peak=rate / 1e9 / HBM_PEAK_GBPS)
            else:
                # the entry point's suffix names the pipe its products ran on
                base = name.split("#")[0]           # "#tag": ops.timer_tag (e.g. the overlapped dgrad launches)
                base = base[:-4] if base.endswith("_a16") else base
                peak = (MFMA_BF16_PEAK_TFLOPS if base.endswith("_bf16") else
                        MFMA_BF16_PEAK_TFLOPS / 6.0 if base.endswith("_x3") else
                        MFMA_BF16_PEAK_TFLOPS / 3.0 if base.endswith("_h2") else MFMA_F32_PEAK_TFLOPS)
                row.update(bound="mfma", achieved_tflops=rate / 1e12, peak_tflops=peak, frac_of_peak=rate / 1e12 / peak)
        rows[name] = row
    return rows


class _QuietStdout:
    """Native libraries write to file descriptor 1 behind Python's back (RCCL prints a five-line version banner when
    its first communicator is created), and the contract is ONE JSON line on stdout: everything this process writes
    to fd 1 goes to stderr instead, except inside `line()`."""

    def __init__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)

    def line(self, text):
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        try:
            print(text, flush=True)
        finally:
            os.dup2(2, 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="utterances per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--head", choices=["bilstm", "transformer"], default="bilstm",
                    help="temporal head (default: the reference's default BiLSTM = BASELINE config[1])")
    ap.add_argument("--precision", choices=["fp32", "bf16"], default="fp32",
                    help="fp32 = BASELINE config[1] (headline); bf16 = training.mixed_precision "
                         "(bf16 MFMA operands, fp32 accumulate/state) = BASELINE configs[2]/[3]")
    ap.add_argument("--fp32-matmul", choices=["native", "x3", "h2"], default=None,
                    help="fp32 products on v_mfma_f32_32x32x2_f32 (native), as an exact three-term bf16 split (x3: six "
                         "bf16 MFMAs per block) or as two scaled fp16 terms (h2: three fp16 MFMAs per block); "
                         "default: PE_FP32_MATMUL or the library default (h2)")
    ap.add_argument("--act-storage", choices=["fp32", "bf16"], default=None,
                    help="--precision bf16 only: HBM format of the conv stack's activations and their gradients "
                         "(default bf16 = what the reference's autocast keeps in memory; fp32 = operands rounded only)")
    ap.add_argument("--no-native-ref", action="store_true",
                    help="skip the 7 extra steps that time the native fp32 MFMA form for the fp32_native_mfma field")
    ap.add_argument("--family-timing", action="store_true",
                    help="HIP events around EVERY C-ABI call INSIDE the timed region (costs ~2 %% of the step); by "
                         "default only the roofline kernels are bracketed there and the family table comes from "
                         "--family-steps extra steps after it")
    ap.add_argument("--family-steps", type=int, default=3,
                    help="extra steps after the timed region with events around every call (0 = skip; N = 1 only)")
    ap.add_argument("--dp-payload", choices=["fp32", "bf16"], default=None,
                    help="gradient all-reduce payload (N > 1): fp32 (default) or bf16 buckets (57.8 MB instead of 115.5 MB "
                         "per step on xGMI; meant for --precision bf16 = BASELINE config[3])")
    ap.add_argument("--host-steps", type=int, default=None,
                    help="extra steps fed from pinned host memory with double-buffered H2D (default: --steps; 0 = skip)")
    args = ap.parse_args()
    quiet = _QuietStdout()
    bf16 = args.precision == "bf16"

    from pitchextractor_amd import distributed as pdist
    from pitchextractor_amd import ops, synthetic
    from pitchextractor_amd.mel import DEFAULT_MEL_PARAMS, MelSpectrogram
    from pitchextractor_amd.meldataset import H2DPrefetcher
    from pitchextractor_amd.model import JDCNet
    from pitchextractor_amd.optimizers import build_optimizer
    from pitchextractor_amd.trainer import Trainer
    import torch.distributed as dist

    if args.fp32_matmul:
        ops.FP32_MATMUL = args.fp32_matmul
    rank, world, local = pdist.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    dev = torch.device(f"cuda:{local}")
    torch.cuda.set_device(dev)

    torch.manual_seed(1234)                      # same random-init weights on every rank
    net = JDCNet(num_class=1, sequence_model_config=dict(SEQ_CFG, model_type=args.head)).to(dev).train()
    net.dropout_cfg.seed = 1000 + rank
    opt, sched = build_optimizer({"params": net.parameters(), "optimizer_params": {},
                                  "scheduler_params": {"max_lr": 3e-4, "pct_start": 0.0, "epochs": 100,
                                                       "steps_per_epoch": 1000}})
    dp = None
    dp_on = world > 1 or pdist.rehearse_single_rank()       # PE_DP_REHEARSE=1: the RCCL path at world size 1
    if dp_on:
        buffers = [b for b in net.buffers() if b.dtype.is_floating_point]
        dp = pdist.GradientAllReduce(net.flat_gradients(), opt, flat_param=net.flat_parameters, buffers=buffers,
                                     payload=args.dp_payload)
        net.attach_data_parallel(dp)            # buckets go out as backward finalises them
    crit = {"l1": torch.nn.SmoothL1Loss(), "ce": torch.nn.BCEWithLogitsLoss()}
    log = logging.getLogger("bench")
    tr = Trainer(model=net, criterion=crit, optimizer=opt, scheduler=sched, device=str(dev),
                 loss_config={"lambda_f0": 0.1}, logger=log, mel_transform=MelSpectrogram(**DEFAULT_MEL_PARAMS),
                 data_parallel=dp, use_mixed_precision=bf16, activation_storage=args.act_storage if bf16 else None)

    # this rank's shard of the global minibatch: 32 distinct synthetic utterances tiled to the batch
    lo, _ = pdist.shard_range(args.batch * world, rank, world)
    w32, f32, s32 = synthetic.batch(lo % 32, 32)
    reps = (args.batch + 31) // 32
    host = tuple(torch.from_numpy(np.tile(a, (reps, 1))[:args.batch]).pin_memory() for a in (w32, f32, s32))
    batch = tuple(t.to(dev) for t in host)
    real_frames = 1 + w32.shape[1] // DEFAULT_MEL_PARAMS["hop_length"]

    def barrier():
        if dp_on:
            dist.barrier()
        torch.cuda.synchronize(dev)

    last = None
    for i in range(args.warmup):
        t_w = time.perf_counter()
        last = tr.run(batch)
        if rank == 0:
            print(f"[bench] warmup {i}: {time.perf_counter() - t_w:.3f} s loss {last['loss']:.4f}", file=sys.stderr,
                  flush=True)
    barrier()
    torch.cuda.reset_peak_memory_stats(dev)
    x3 = ops.FP32_MATMUL in ("x3", "h2")        # fp32 results from split 16-bit terms on the bf16 / fp16 MFMA pipe
    h2 = ops.FP32_MATMUL == "h2"
    fp32_mode = ops.FP32_MATMUL
    sfx = "_bf16" if bf16 else "_h2" if h2 else "_x3" if x3 else ""
    conv_key = "pe_conv3x3_fwd" + sfx
    # weights staged through LDS / fed as fragments from L2; "_a16": bf16 activation storage (mixed precision)
    conv_keys = {conv_key, "pe_conv3x3_fwd_wf" + sfx, conv_key + "_a16", "pe_conv3x3_fwd_wf" + sfx + "_a16"}
    ops.TIMER = ops.KernelTimer(None if args.family_timing else conv_keys | {"pe_mel_forward"})
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last = tr.run(batch)
    barrier()
    elapsed = time.perf_counter() - t0
    timer, ops.TIMER = ops.TIMER, None
    peak_mem = int(torch.cuda.max_memory_allocated(dev))
    if rank == 0:
        print(f"[bench] timed {args.steps} steps: {elapsed / args.steps * 1e3:.1f} ms/step", file=sys.stderr,
              flush=True)
    if dp_on:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms = elapsed / args.steps * 1e3

    # ---- everything below runs AFTER the timed region and never enters `value` ----------------------
    # (a) family table: events around every C-ABI call for a few extra steps (N = 1 only)
    fam_summ, fam_ms = (timer.summary(), ms) if args.family_timing else (None, None)
    if world == 1 and not args.family_timing and args.family_steps > 0:
        # with the side-stream overlaps OFF: an event pair around a launch then times that kernel alone (concurrent
        # kernels would each be charged the shared wall time), and the rows add up to a serialised step
        from pitchextractor_amd import model as pe_model
        saved_ov = (pe_model.OVERLAP_CONV_WGRAD, pe_model.OVERLAP_LSTM_WGRAD, pe_model.OVERLAP_TF_WGRAD)
        pe_model.OVERLAP_CONV_WGRAD = pe_model.OVERLAP_LSTM_WGRAD = pe_model.OVERLAP_TF_WGRAD = False
        try:
            ops.TIMER = None
            tr.run(batch)                       # (scratch buffers of the main stream grow once for the serialised order)
            ops.TIMER = ops.KernelTimer(None)
            torch.cuda.synchronize(dev)
            t_f = time.perf_counter()
            for _ in range(args.family_steps):
                tr.run(batch)
            torch.cuda.synchronize(dev)
            fam_ms = (time.perf_counter() - t_f) / args.family_steps * 1e3
            fam_summ, ops.TIMER = ops.TIMER.summary(), None
        finally:
            pe_model.OVERLAP_CONV_WGRAD, pe_model.OVERLAP_LSTM_WGRAD, pe_model.OVERLAP_TF_WGRAD = saved_ov

    # (b) SURVEY 8(d)'s step: from pinned host audio, the next batch's H2D (49 MB) in flight on a side stream
    from_host = None
    host_steps = args.steps if args.host_steps is None else args.host_steps
    if host_steps > 0:
        # data-parallel runs (three hardware queues): copies issued from the weight-gradient side stream, idle in forward
        from pitchextractor_amd import model as pe_model_h
        feeder = H2DPrefetcher(dev, stream=pe_model_h._side_stream(dev) if dp_on else None)
        tr.run(feeder.acquire(feeder.submit(host)))                       # warm the side-stream allocations
        barrier()
        t_h = time.perf_counter()
        ticket = feeder.submit(host)
        for k in range(host_steps):
            nxt = feeder.submit(host) if k + 1 < host_steps else None
            tr.run(feeder.acquire(ticket))
            ticket = nxt
        barrier()
        el_h = time.perf_counter() - t_h
        if dp_on:
            t = torch.tensor([el_h], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el_h = float(t.item())
        from_host = {"value": args.batch * world * FRAMES * host_steps / el_h, "unit": "mel-frames/s",
                     "ms_per_step": el_h / host_steps * 1e3, "steps": host_steps,
                     "h2d_bytes_per_step": int(sum(t.numel() * t.element_size() for t in host)),
                     "note": "step starts at pinned host audio; batch k+1's H2D overlaps step k on a side stream"}

    # (c) the same step with the fp32 products on the native fp32 MFMA instructions, for reference (N = 1 only)
    native_ref = None
    if world == 1 and x3 and not bf16 and not args.no_native_ref:
        ops.FP32_MATMUL = "native"
        try:
            for _ in range(2):
                tr.run(batch)
            torch.cuda.synchronize(dev)
            t_n = time.perf_counter()
            for _ in range(5):
                tr.run(batch)
            torch.cuda.synchronize(dev)
            ms_n = (time.perf_counter() - t_n) / 5 * 1e3
            native_ref = {"ms_per_step": ms_n, "value": args.batch * FRAMES / (ms_n * 1e-3), "unit": "mel-frames/s",
                          "steps": 5, "note": "PE_FP32_MATMUL=native: v_mfma_f32_32x32x2_f32 for every fp32 product"}
        finally:
            ops.FP32_MATMUL = fp32_mode

    if rank == 0:
        frames = args.batch * world * FRAMES * args.steps
        summ = timer.summary()
        from pitchextractor_amd import model as pe_model
        overlapped = pe_model.OVERLAP_CONV_WGRAD      # dgrad launches then share the GPU with side-stream wgrad kernels

        def fold(keys):
            parts = [summ[k] for k in keys if k in summ]
            if not parts:
                return None
            c = {"calls": sum(p["calls"] for p in parts), "total_ms": sum(p["total_ms"] for p in parts),
                 "work": sum(p["work"] for p in parts)}
            c["avg_ms"] = c["total_ms"] / c["calls"]
            return c

        fwd_only, dgrad = fold(conv_keys), fold({k + "#dgrad" for k in conv_keys})
        # the roofline is taken over launches that run ALONE: forward + dgrad when nothing overlaps them, forward only
        # when the dgrad launches share the GPU with the side-stream weight-gradient kernels
        conv = fwd_only if (overlapped or dgrad is None) else fold(conv_keys | {k + "#dgrad" for k in conv_keys})
        wf_used = ("pe_conv3x3_fwd_wf" + sfx) in summ or ("pe_conv3x3_fwd_wf" + sfx + "_a16") in summ
        roof = None
        if conv:
            tflops = conv["work"] / (conv["total_ms"] * 1e-3) / 1e12        # algorithmic 2*M*N*K per launch
            if bf16:
                peak, note = MFMA_BF16_PEAK_TFLOPS, "bf16 dense MFMA peak"
            elif h2:
                peak, note = MFMA_BF16_PEAK_TFLOPS / 3.0, ("fp32 product = 3 fp16 MFMAs (two scaled fp16 terms per operand): "
                                                           "fp16 dense MFMA peak / 3 (builder-defined ceiling); the "
                                                           f"native fp32 MFMA peak is {MFMA_F32_PEAK_TFLOPS} TFLOP/s")
            elif x3:
                peak, note = MFMA_BF16_PEAK_TFLOPS / 6.0, ("fp32 product = 6 bf16 MFMAs (exact 3-term split): "
                                                           "bf16 dense MFMA peak / 6 (builder-defined ceiling); the "
                                                           f"native fp32 MFMA peak is {MFMA_F32_PEAK_TFLOPS} TFLOP/s")
            else:
                peak, note = MFMA_F32_PEAK_TFLOPS, "fp32 MFMA (32x32x2) peak"
            which = ("forward launches, which run alone; the same kernel's dgrad launches share the GPU with side-stream "
                     "weight-gradient kernels: dgrad_overlapped" if (overlapped and dgrad) else "fwd + dgrad launches")
            roof = {"bound": "mfma", "kernel": ("conv3x3_kernel (implicit-GEMM; " + which + ")" if not (bf16 or x3) else
                                                ("conv3x3_halo_wf_kernel (halo-staged activations, weight fragments from L2; "
                                                 + which + ")" if wf_used else
                                                 "conv3x3_kernel (implicit GEMM, 16-bit terms; " + which + ")")),
                    "achieved": tflops, "peak": peak, "unit": "TFLOP/s", "frac": tflops / peak,
                    "peak_note": note, "traffic": pmc_traffic(conv_key)[0], "mfma_busy_pmc": pmc_traffic(conv_key)[1],
                    "traffic_source": pmc_source() + " (rocprofv3 --pmc passes of this command, committed; not "
                                                     "re-measured in this run)",
                    "avg_launch_ms": conv["avg_ms"], "launches_per_step": conv["calls"] / args.steps}
            # the whole step against the same ceiling: SURVEY 8(d)'s 369.6 (BiLSTM) / 344.4 (Transformer) MFLOP per frame
            step_flop = (369.6e6 if args.head == "bilstm" else 344.4e6) * args.batch * FRAMES
            roof["whole_step"] = {"achieved": step_flop / (ms * 1e-3) / 1e12, "unit": "TFLOP/s",
                                  "frac": step_flop / (ms * 1e-3) / 1e12 / peak,
                                  "flop_per_step": step_flop,
                                  "note": "all algorithmic FLOPs of the step (fwd + dgrad + wgrad, every family) over the "
                                          "timed step, same peak as above"}
            if overlapped and dgrad:
                roof["dgrad_overlapped"] = {"avg_launch_ms": dgrad["avg_ms"], "launches_per_step": dgrad["calls"] / args.steps,
                                            "achieved": dgrad["work"] / (dgrad["total_ms"] * 1e-3) / 1e12,
                                            "note": "HIP-event time of a launch that runs concurrently with "
                                                    "conv3x3_wgrad9 / gemm_tn on a side stream (PE_OVERLAP_CONV_WGRAD=0 "
                                                    "serialises them)"}
        mel = summ.get("pe_mel_forward")
        roof_mel = None
        if mel:
            gbps = mel["work"] / (mel["total_ms"] * 1e-3) / 1e9              # 1 520 B x real frames per launch
            roof_mel = {"bound": "hbm", "kernel": "mel_fwd_kernel (frame + Hann + rFFT + power + mel + log + pad)",
                        "achieved": gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": gbps / HBM_PEAK_GBPS,
                        "bytes_per_launch": mel["work"] / mel["calls"], "avg_launch_ms": mel["avg_ms"],
                        "traffic": None,
                        "note": "62.6 MB per launch fits the 256 MiB Infinity Cache: this is the warm in-step rate; "
                                "the cold multi-batch rate is tools/bench_mel.py (profiles/)"}
        line = {
            "metric": "mel-frames/sec training throughput (JDCNet, 24 kHz, batch=256)",
            "value": frames / elapsed, "unit": "mel-frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16" if bf16 else "f32", "data": "synthetic",
            "config": {"workload": (("BASELINE config[2]: batch" if args.head == "transformer" else "BASELINE config[3] per-GPU shape: batch") if bf16 else "BASELINE config[1]: batch") + "=256/GPU, 24 kHz 2 s synthetic glides "
                                   f"({real_frames} real frames zero-padded to 192), JDCNet+{'BiLSTM(4x384)' if args.head == 'bilstm' else 'Transformer(4 layers, 8 heads, ff 1536)'}, " + ("mixed precision (bf16 conv/linear operands, fp32 accumulate, " + ("bf16" if tr.act16 else "fp32") + " activation storage), " if bf16 else f"fp32 (products: {ops.FP32_MATMUL}), ") +
                                   "raw audio resident in HBM -> mel -> fwd -> loss -> bwd -> AdamW",
                       "global_batch": args.batch * world, "frames_per_utterance": FRAMES,
                       "real_frames_per_utterance": real_frames, "parallelism": f"dp{world}"},
            "loss": last["loss"], "roofline": roof, "roofline_mel": roof_mel,
            "peak_memory_bytes": peak_mem,              # torch.cuda.max_memory_allocated over the timed steps, rank 0
        }
        if bf16:
            line["config"]["activation_storage"] = "bf16" if tr.act16 else "fp32"
        if dp is not None:
            line["config"]["gradient_allreduce"] = {
                "backend": dist.get_backend(), "payload": dp.payload, "bucket_bytes": dp.bucket_elems * (2 if dp.payload == "bf16" else 4),
                "bytes_per_step": net.flat_gradients().numel() * (2 if dp.payload == "bf16" else 4),
                "messages_per_step": dp.messages / max(1, tr._runs),
                "rehearsal_world1": bool(world == 1)}
        if from_host is not None:
            line["from_host"] = from_host
        if fam_summ is not None:
            line["kernel_families"] = {"steps": args.steps if args.family_timing else args.family_steps,
                                       "ms_per_step_with_events": fam_ms,
                                       "note": ("extra steps with the side-stream weight-gradient overlaps off: every row "
                                                "is a kernel family timed alone") if not args.family_timing else
                                               "events inside the timed region (concurrent kernels share wall time)",
                                       "rows": _family_table(fam_summ, args.steps if args.family_timing else args.family_steps,
                                                             fam_ms)}
        if native_ref is not None:
            line["fp32_native_mfma"] = native_ref
        if world == 1 and not args.no_cpu_baseline and args.head == "bilstm" and not bf16:
            line["cpu_baseline"] = cpu_baseline()
        quiet.line(json.dumps(line))
    if dp_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
