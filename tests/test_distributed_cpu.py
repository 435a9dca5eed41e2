"""Data-parallel wiring on CPU (gloo, world_size 2): gradient buckets are summed across ranks, the
optimizer is told to divide by the world size, parameters/buffers are broadcast from rank 0."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pitchextractor_amd import distributed as pdist


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


class _Opt:
    grad_scale = 1.0


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    r, w, _ = pdist.init_from_env("gloo")
    assert (r, w) == (rank, world)
    n = 100_003
    g = torch.arange(n, dtype=torch.float32) * (rank + 1)            # rank-specific "gradients"
    p = torch.full((n,), float(rank + 7))
    buf = torch.full((5,), float(rank))
    opt = _Opt()
    dp = pdist.GradientAllReduce(g, opt, bucket_bytes=64 << 10, flat_param=p, buffers=[buf])
    assert len(dp.buckets) > 1 and opt.grad_scale == 0.5
    assert torch.equal(p, torch.full((n,), 7.0)) and torch.equal(buf, torch.zeros(5))      # broadcast from rank 0
    dp.reduce_range(*dp.buckets[0])                                                        # early bucket ...
    for lo, hi in dp.buckets[1:]:
        dp.reduce_range(lo, hi)
    dp.finish()                                                                            # ... then the rest
    expect = torch.arange(n, dtype=torch.float32) * 3.0                                    # (1 + 2) * arange
    assert torch.equal(g, expect)
    g2 = torch.ones(10) * (rank + 1)
    dp2 = pdist.GradientAllReduce(g2, None)
    dp2.finish()                                                                           # nothing issued: all buckets
    assert torch.equal(g2, torch.full((10,), 3.0))
    lo, hi = pdist.shard_range(512, rank, world)
    assert (lo, hi) == (256 * rank, 256 * rank + 256)
    np.save(os.path.join(out_dir, f"ok{rank}.npy"), np.array([1]))
    dist.destroy_process_group()


def test_two_rank_gradient_all_reduce(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok0.npy").exists() and (tmp_path / "ok1.npy").exists()


def test_single_process_is_a_no_op():
    g = torch.ones(8)
    opt = _Opt()
    dp = pdist.GradientAllReduce(g, opt)
    dp.finish()
    assert torch.equal(g, torch.ones(8)) and opt.grad_scale == 1.0
