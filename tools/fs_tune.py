"""Experiment driver: the B = 256 tiling test (gradient element error against float64) under a switch.
  python3 tools/fs_tune.py nopersist | lstm_native"""
import os
import sys

sys.path.insert(0, "/root/repo")
os.chdir("/root/repo")
import pytest  # noqa: E402

from pitchextractor_amd import ops  # noqa: E402

what = sys.argv[1]
if what == "nopersist":
    ops.USE_PERSISTENT_LSTM = False
elif what == "lstm_native":
    ops.LSTM_X3["fwd"] = ops.LSTM_X3["bwd"] = False
print("SWITCH", sys.argv[1:], flush=True)
sys.exit(pytest.main(["tests/test_model_gpu.py", "-x", "-q", "-m", "gpu", "-s", "-k", "full_size_training_step and h2"]))
