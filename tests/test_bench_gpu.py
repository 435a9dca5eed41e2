"""bench.py's contract with the driver: ONE JSON line on stdout carrying the metric fields, the roofline objects and
(N = 1) the CPU baseline; for N > 1 the multi-GPU launch command, rehearsed here with two ranks on one GPU over gloo
(backend RCCL needs one GPU per rank) and with one rank over RCCL itself."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
CONTRACT = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config"}


def _one_line(res):
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, res.stdout[:2000]                   # nothing but the JSON line on stdout
    return json.loads(lines[0])


def test_single_gpu_line_has_the_contract_fields(hip_device):
    res = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--batch", "16", "--steps", "2", "--warmup", "1",
                          "--family-steps", "1", "--host-steps", "1"], cwd=str(ROOT), capture_output=True, text=True,
                         timeout=900)
    d = _one_line(res)
    assert CONTRACT <= set(d) and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["unit"] == "mel-frames/s" and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert d["config"]["global_batch"] == 16 and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 16 * 192 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    roof = d["roofline"]
    assert roof["bound"] == "mfma" and roof["unit"] == "TFLOP/s" and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9
    mel = d["roofline_mel"]
    assert mel["bound"] == "hbm" and mel["peak"] == 8000.0 and 0 < mel["frac"] < 1
    cpu = d["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["cores"] >= 1 and cpu["value"] > 0 and "sample" in cpu
    assert d["from_host"]["ms_per_step"] > 0 and d["kernel_families"]["rows"] and d["fp32_native_mfma"]["ms_per_step"] > 0 and roof["whole_step"]["frac"] > 0 and "traffic_source" in roof


def test_two_rank_launch_over_gloo(hip_device):
    env = dict(os.environ, PE_FORCE_DEVICE="0", PE_DIST_BACKEND="gloo")
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29536", str(ROOT / "bench.py"), "--gpus", "2",
                          "--batch", "8", "--steps", "2", "--warmup", "1", "--host-steps", "1"], cwd=str(ROOT), env=env,
                         capture_output=True, text=True, timeout=900)
    d = _one_line(res)
    assert CONTRACT <= set(d) and d["n_gpus"] == 2 and d["config"]["global_batch"] == 16
    assert d["config"]["parallelism"] == "dp2" and "cpu_baseline" not in d
    ar = d["config"]["gradient_allreduce"]
    assert ar["backend"] == "gloo" and ar["payload"] == "fp32" and ar["messages_per_step"] >= 5
    assert abs(d["value"] - 16 * 192 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]     # whole-job aggregate


def test_rccl_launch_at_world_size_one_keeps_stdout_clean(hip_device):
    """RCCL prints a version banner to fd 1 when its first communicator is created."""
    env = dict(os.environ, PE_DP_REHEARSE="1")
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                          "--master-addr", "127.0.0.1", "--master-port", "29537", str(ROOT / "bench.py"), "--gpus", "1",
                          "--batch", "16", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-native-ref",
                          "--family-steps", "0", "--host-steps", "1", "--precision", "bf16", "--dp-payload", "bf16"],
                         cwd=str(ROOT), env=env, capture_output=True, text=True, timeout=900)
    d = _one_line(res)
    ar = d["config"]["gradient_allreduce"]
    assert ar["backend"] == "nccl" and ar["payload"] == "bf16" and ar["rehearsal_world1"] is True
    assert d["dtype"] == "bf16" and d["n_gpus"] == 1


def test_three_rank_launch_as_the_driver_launches_it(hip_device):
    """The driver's multi-GPU command with an odd world size on one card over gloo (a one-GPU box admits six processes
    on the card, and the test runner and the launcher count: three ranks; the eight-rank arithmetic -- shards, block
    cuts, buckets, MAX-over-ranks -- runs on CPU in tests/test_distributed_cpu.py): one JSON line, whole-job aggregate."""
    env = dict(os.environ, PE_FORCE_DEVICE="0", PE_DIST_BACKEND="gloo")
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3",
                          "--master-addr", "127.0.0.1", "--master-port", "29538", str(ROOT / "bench.py"), "--gpus", "3",
                          "--batch", "8", "--steps", "2", "--warmup", "1", "--host-steps", "0"], cwd=str(ROOT), env=env,
                         capture_output=True, text=True, timeout=1200)
    d = _one_line(res)
    assert CONTRACT <= set(d) and d["n_gpus"] == 3 and d["config"]["global_batch"] == 24
    assert d["config"]["parallelism"] == "dp3" and "cpu_baseline" not in d and "kernel_families" not in d
    assert d["config"]["gradient_allreduce"]["messages_per_step"] >= 5
    assert abs(d["value"] - 24 * 192 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
