"""train.py end to end on the HIP path: the reference's CLI (train.py:45-150) with a YAML of the reference's keys,
a tiny synthetic dataset on disk (wav + cached F0, the reference's file layout), two epochs, checkpoints."""
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import yaml

from pitchextractor_amd import synthetic
from tests.test_data_layer import write_wav

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def test_train_py_runs_two_epochs_and_checkpoints(tmp_path, hip_device):
    lines = []
    for i in range(8):
        wave, f0, _ = synthetic.utterance(i, duration=2.0)
        p = tmp_path / f"u{i}.wav"
        write_wav(p, wave, 24000, "float32")
        np.save(str(p) + "_f0.npy", f0)
        lines.append(f"{p}|0\n")
    (tmp_path / "train_list.txt").write_text("".join(lines[:6]))
    (tmp_path / "val_list.txt").write_text("".join(lines[6:]))
    cfg = yaml.safe_load((ROOT / "Configs" / "config.yml").read_text())
    cfg.update(log_dir=str(tmp_path / "ckpt"), save_freq=1, epochs=2, batch_size=2, num_workers=0,
               train_data=str(tmp_path / "train_list.txt"), val_data=str(tmp_path / "val_list.txt"))
    cfg["model_params"]["sequence_model"].update(hidden_size=64, num_layers=2)
    cfg_path = tmp_path / "config.yml"
    cfg_path.write_text(yaml.safe_dump(cfg))
    res = subprocess.run([sys.executable, str(ROOT / "train.py"), "-p", str(cfg_path)], cwd=str(ROOT),
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    log = (tmp_path / "ckpt" / "train.log").read_text()
    assert "--- epoch 2 ---" in log and "train/loss" in log and "eval/loss" in log
    ck = torch.load(tmp_path / "ckpt" / "epoch_00002.pth", map_location="cpu", weights_only=True)
    assert set(ck) == {"optimizer", "scheduler", "steps", "epochs", "model"} and ck["epochs"] == 2
    assert "sequence_classifier.model.weight_hh_l1_reverse" in ck["model"]
    assert all(torch.isfinite(v).all() for v in ck["model"].values() if v.dtype.is_floating_point)


def test_train_py_two_ranks(tmp_path, hip_device):
    """The data-parallel wiring of train.py (rank-sharded file lists, gradient all-reduce, rank-0 logging):
    two ranks rehearsed on one GPU over gloo (the production backend is RCCL, one process per GPU)."""
    import os
    lines = []
    for i in range(11):          # 11 files, 2 ranks x batch 2: not divisible -- ranks must still run equal step counts
        wave, f0, _ = synthetic.utterance(i, duration=2.0)
        p = tmp_path / f"u{i}.wav"
        write_wav(p, wave, 24000, "float32")
        np.save(str(p) + "_f0.npy", f0)
        lines.append(f"{p}|0\n")
    (tmp_path / "train_list.txt").write_text("".join(lines))
    (tmp_path / "val_list.txt").write_text("".join(lines[:3]))
    cfg = yaml.safe_load((ROOT / "Configs" / "config.yml").read_text())
    cfg.update(log_dir=str(tmp_path / "ckpt"), save_freq=1, epochs=2, batch_size=4, num_workers=0,
               train_data=str(tmp_path / "train_list.txt"), val_data=str(tmp_path / "val_list.txt"))
    cfg["model_params"]["sequence_model"].update(hidden_size=64, num_layers=1)
    cfg_path = tmp_path / "config.yml"
    cfg_path.write_text(yaml.safe_dump(cfg))
    env = dict(os.environ, PE_FORCE_DEVICE="0", PE_DIST_BACKEND="gloo")
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29533", str(ROOT / "train.py"), "-p",
                          str(cfg_path)], cwd=str(ROOT), env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    log = (tmp_path / "ckpt" / "train.log").read_text()
    assert "--- epoch 2 ---" in log and "train/loss" in log and "eval/loss" in log
    assert (tmp_path / "ckpt" / "epoch_00002.pth").exists()


def test_train_py_over_rccl_at_world_size_one(tmp_path, hip_device):
    """train.py launched as the multi-GPU command launches it (torch.distributed.run, backend nccl = RCCL) with
    PE_DP_REHEARSE=1 at world size 1: process group, sharded sampler, gradient buckets (bf16 payload, mixed
    precision as the shipped config has it), cross-rank flags and the logging reductions all run through RCCL."""
    import os
    lines = []
    for i in range(7):
        wave, f0, _ = synthetic.utterance(i, duration=2.0)
        p = tmp_path / f"u{i}.wav"
        write_wav(p, wave, 24000, "float32")
        np.save(str(p) + "_f0.npy", f0)
        lines.append(f"{p}|0\n")
    (tmp_path / "train_list.txt").write_text("".join(lines))
    (tmp_path / "val_list.txt").write_text("".join(lines[:3]))
    cfg = yaml.safe_load((ROOT / "Configs" / "config.yml").read_text())
    cfg.update(log_dir=str(tmp_path / "ckpt"), save_freq=1, epochs=2, batch_size=2, num_workers=0,
               train_data=str(tmp_path / "train_list.txt"), val_data=str(tmp_path / "val_list.txt"))
    cfg["model_params"]["sequence_model"].update(hidden_size=64, num_layers=1)
    cfg.setdefault("training", {})["gradient_payload"] = "bf16"
    cfg_path = tmp_path / "config.yml"
    cfg_path.write_text(yaml.safe_dump(cfg))
    env = dict(os.environ, PE_DP_REHEARSE="1")
    env.pop("GPU_MAX_HW_QUEUES", None)
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                          "--master-addr", "127.0.0.1", "--master-port", "29534", str(ROOT / "train.py"), "-p",
                          str(cfg_path)], cwd=str(ROOT), env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    assert "GPU_MAX_HW_QUEUES is unset" not in res.stderr          # train.py sets it before the runtime loads
    log = (tmp_path / "ckpt" / "train.log").read_text()
    assert "--- epoch 2 ---" in log and "train/loss" in log and "eval/loss" in log
    ck = torch.load(tmp_path / "ckpt" / "epoch_00002.pth", map_location="cpu", weights_only=True)
    assert all(torch.isfinite(v).all() for v in ck["model"].values() if v.dtype.is_floating_point)
