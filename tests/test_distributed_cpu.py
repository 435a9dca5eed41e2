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


def _epoch_worker(rank, world, port, out_dir):
    """What train.py does per epoch, minus the HIP step: sampler-driven DataLoader + one all-reduce per batch."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    pdist.init_from_env("gloo")
    n_files, batch = 11, 2                                  # 11 % (2 ranks * 2) != 0: the advisor's hang case
    sampler = pdist.EpochShardSampler(n_files, batch, rank, world, seed=5)
    loader = torch.utils.data.DataLoader(list(range(n_files)), batch_size=batch, sampler=sampler, drop_last=True)
    assert len(loader) == 2                                 # same steps_per_epoch on every rank
    seen = []
    for epoch in (1, 2):
        sampler.set_epoch(epoch)
        mine = []
        for b in loader:
            t = torch.ones(1)
            dist.all_reduce(t)                              # the step's gradient all-reduce: must pair up
            assert t.item() == world
            mine += b.tolist()
        seen.append(mine)
    stats = pdist.mean_over_ranks({"eval/loss": float(rank + 1)}, weight=float(rank + 1))
    assert abs(stats["eval/loss"] - (1 * 1 + 2 * 2) / 3.0) < 1e-12
    assert pdist.GradientAllReduce(torch.zeros(4), None).any_rank(rank == 1) is True
    dpw = pdist.GradientAllReduce(torch.zeros(4), None)
    assert dpw.any_rank_word(torch.tensor([7 if rank == 1 else 0], dtype=torch.int32)) is True     # one rank faulted
    assert dpw.any_rank_word(torch.zeros(1, dtype=torch.int32)) is False and dpw.any_rank_word(None) is False
    np.save(os.path.join(out_dir, f"seen{rank}.npy"), np.array(seen))
    dist.destroy_process_group()


def test_two_rank_epoch_with_indivisible_file_count_terminates(tmp_path):
    port = _free_port()
    mp.spawn(_epoch_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = np.load(tmp_path / "seen0.npy"), np.load(tmp_path / "seen1.npy")
    assert a.shape == b.shape == (2, 4)
    for e in range(2):
        assert not set(a[e]) & set(b[e])                    # disjoint rank shards of one permutation
    assert a[0].tolist() != a[1].tolist()                   # reshuffled across ranks between epochs
    g = torch.Generator(); g.manual_seed(5 + 1)
    perm = torch.randperm(11, generator=g).tolist()
    assert a[0].tolist() == perm[0:4] and b[0].tolist() == perm[4:8]       # contiguous cuts, common length


def test_shard_sampler_validation_covers_every_item_once():
    parts = [list(pdist.EpochShardSampler(11, 4, r, 3, shuffle=False, drop_last=False)) for r in range(3)]
    assert sorted(sum(parts, [])) == list(range(11)) and [len(p) for p in parts] == [4, 4, 3]
    assert len(pdist.EpochShardSampler(3, 2, 0, 2)) == 0     # cannot fill a batch per rank: train.py refuses


def test_single_process_is_a_no_op():
    g = torch.ones(8)
    opt = _Opt()
    dp = pdist.GradientAllReduce(g, opt)
    dp.finish()
    assert torch.equal(g, torch.ones(8)) and opt.grad_scale == 1.0


def _bf16_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    pdist.init_from_env("gloo")
    n = 70_001
    gen = torch.Generator(); gen.manual_seed(100 + rank)
    g = torch.randn(n, generator=gen)
    mine = g.clone()
    dp = pdist.GradientAllReduce(g, _Opt(), bucket_bytes=64 << 10, payload="bf16")
    assert dp.bucket_elems == (64 << 10) // 2 and g.dtype == torch.float32      # same bytes per message, fp32 buffer
    dp.reduce_range(0, 1000)                                                    # a block range, then the rest
    dp.reduce_range(1000, n)
    dp.finish()
    assert dp.messages == 1 + -(-(n - 1000) // dp.bucket_elems)
    gen2 = torch.Generator(); gen2.manual_seed(100 + (1 - rank))
    other = torch.randn(n, generator=gen2)
    # what the collective computes: each rank's bucket rounded to bf16, summed in bf16, widened back
    expect = (mine.to(torch.bfloat16) + other.to(torch.bfloat16)).float()
    assert torch.equal(g, expect)
    # and how far that is from the fp32 sum: unit roundoff 2^-8 on each term and on the sum
    assert bool(((g - (mine + other)).abs() <= 2.0 ** -7 * (mine.abs() + other.abs()) + 1e-30).all())
    np.save(os.path.join(out_dir, f"bf{rank}.npy"), g.numpy())
    dist.destroy_process_group()


def test_two_rank_bf16_gradient_buckets(tmp_path):
    mp.spawn(_bf16_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert np.array_equal(np.load(tmp_path / "bf0.npy"), np.load(tmp_path / "bf1.npy"))     # replicas agree bit for bit


def _rehearse_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      PE_DP_REHEARSE="1")
    r, w, _ = pdist.init_from_env("gloo")
    assert (r, w) == (0, 1) and dist.is_initialized()       # a process group even at world size 1
    g = torch.arange(5000, dtype=torch.float32)
    opt = _Opt()
    dp = pdist.GradientAllReduce(g, opt, bucket_bytes=4 << 10, flat_param=torch.ones(3))
    assert dp.active and opt.grad_scale == 1.0
    dp.finish()
    assert dp.messages == 5 and torch.equal(g, torch.arange(5000, dtype=torch.float32))     # sum over one rank
    assert dp.any_rank(True) is True and dp.any_rank(False) is False
    assert pdist.mean_over_ranks({"a": 2.0}, weight=3.0) == {"a": 2.0}
    np.save(os.path.join(out_dir, "ok.npy"), np.array([1]))
    dist.destroy_process_group()


def test_world1_rehearsal_runs_the_collective_path(tmp_path):
    """PE_DP_REHEARSE=1: the one-GPU box can execute init_process_group / broadcast / bucketed all-reduce (with
    backend nccl = RCCL there; gloo here)."""
    mp.spawn(_rehearse_worker, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    assert (tmp_path / "ok.npy").exists()


def test_backward_order_block_cuts_of_the_flat_gradient_buffer():
    """JDCNet hands the reducer [heads], [res_block3 + pool_block + detector_conv], res_block2, res_block1, conv_block
    in that order: the cuts must tile the flat buffer in parameter order."""
    from pitchextractor_amd.model import JDCNet
    net = JDCNet(num_class=1, sequence_model_config={"model_type": "bilstm", "num_layers": 1, "hidden_size": 32})
    cuts = net._block_cuts()
    assert cuts is not None and cuts[0] == 0 and cuts == sorted(cuts) and cuts[-1] == net._seq_offset()
    off = net._param_offsets
    for k, blk in enumerate([net.conv_block, net.res_block1, net.res_block2, net.res_block3]):
        assert all(cuts[k] <= off[id(p)] and off[id(p)] + p.numel() <= cuts[k + 1] for p in blk.parameters())
    for blk in (net.pool_block, net.detector_conv):
        assert all(cuts[3] <= off[id(p)] and off[id(p)] + p.numel() <= cuts[4] for p in blk.parameters())
    issued = []

    class _Rec:
        active, world = True, 2
        def reduce_range(self, lo, hi, after=None):
            issued.append((lo, hi))
    net.attach_data_parallel(_Rec())
    assert net._dp_cuts == cuts


def test_shard_sampler_properties():
    """For any list size / batch / world: training shards are disjoint, equally long (whole batches, so every rank
    runs the same number of steps), identical across ranks' view of the epoch permutation, and reshuffled per epoch;
    validation shards cover every item exactly once."""
    from hypothesis import given, settings, strategies as st

    @settings(max_examples=150, deadline=None)
    @given(n=st.integers(0, 300), batch=st.integers(1, 9), world=st.integers(1, 8), seed=st.integers(0, 10_000),
           epoch=st.integers(0, 50))
    def check(n, batch, world, seed, epoch):
        shards = []
        for r in range(world):
            s = pdist.EpochShardSampler(n, batch, r, world, seed=seed)
            s.set_epoch(epoch)
            shards.append(list(s))
            assert len(shards[-1]) == len(s) == (n // (world * batch)) * batch
        flat = sum(shards, [])
        assert len(set(flat)) == len(flat) and all(0 <= i < n for i in flat)
        if n >= 2 * world * batch:            # a permutation, not the identity cut (overwhelmingly likely)
            other = pdist.EpochShardSampler(n, batch, 0, world, seed=seed)
            other.set_epoch(epoch + 1)
            assert len(list(other)) == len(shards[0])
        val = [list(pdist.EpochShardSampler(n, batch, r, world, shuffle=False, drop_last=False)) for r in range(world)]
        assert sorted(sum(val, [])) == list(range(n))
        assert max(map(len, val)) - min(map(len, val)) <= 1

    check()


def _world8_worker(rank, world, port, out_dir):
    """The 8-rank shape of the driver's scaling run, on CPU: JDCNet's backward-order block cuts over its real flat
    gradient buffer (default BiLSTM: 28.9 M elements + the status slot), fp32 and bf16 buckets, the shard arithmetic
    bench.py uses for rank 7, the epoch sampler, and the MAX-over-ranks timing reduction."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    r, w, local = pdist.init_from_env("gloo")
    assert (r, w, local) == (rank, 8, rank)
    from pitchextractor_amd.model import JDCNet
    torch.manual_seed(1234)
    net = JDCNet(num_class=1, sequence_model_config={"model_type": "bilstm", "num_layers": 1, "hidden_size": 32})
    cuts = net._block_cuts()
    n = net.flat_gradients().numel()
    assert cuts is not None and n == net._status_off + 4
    for payload in ("fp32", "bf16"):
        g = net.flat_gradients()
        g.copy_(torch.arange(n, dtype=torch.float32) % 251 * (1.0 if payload == "fp32" else 0.5) + rank)
        net.status_slot().fill_(1.0 if rank == 5 else 0.0)            # one rank reports a fault
        opt = _Opt()
        dp = pdist.GradientAllReduce(g, opt, bucket_bytes=256 << 10, flat_param=net.flat_parameters.data, payload=payload)
        assert opt.grad_scale == 1.0 / 8
        # the order JDCNet._backward_impl issues: temporal + output heads (with the status slot), block 3, 2, 1, conv_block
        dp.reduce_range(net._seq_offset(), n)
        for k in (3, 2, 1):
            dp.reduce_range(cuts[k], cuts[k + 1])
        dp.reduce_range(0, cuts[1])
        dp.finish()
        base = torch.arange(n, dtype=torch.float32) % 251 * (1.0 if payload == "fp32" else 0.5)
        expect = 8 * base + sum(range(8))
        expect[net._status_off] = 1.0                                  # the fault count, seen by every rank
        expect[net._status_off + 1:] = 8 * base[net._status_off + 1:] + sum(range(8))
        if payload == "fp32":
            assert torch.equal(g, expect)
        else:                                                          # small integers / halves: exact in bf16 sums too
            assert torch.allclose(g, expect, rtol=2.0 ** -6, atol=0)
        assert g[net._status_off].item() == 1.0
    lo, hi = pdist.shard_range(8 * 256, rank, world)
    assert (lo, hi) == (256 * rank, 256 * rank + 256) and lo % 32 == 0
    s = pdist.EpochShardSampler(1000, 8, rank, world, seed=3)
    s.set_epoch(2)
    assert len(s) == (1000 // 64) * 8
    t = torch.tensor([float(rank)], dtype=torch.float64)              # bench.py: elapsed = MAX over ranks
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert t.item() == 7.0
    np.save(os.path.join(out_dir, f"w8_{rank}.npy"), np.array(list(s)))
    dist.destroy_process_group()


def test_world_8_block_cuts_buckets_and_sharding(tmp_path):
    mp.spawn(_world8_worker, args=(8, _free_port(), str(tmp_path)), nprocs=8, join=True)
    shards = [np.load(tmp_path / f"w8_{r}.npy") for r in range(8)]
    flat = np.concatenate(shards)
    assert len(set(flat.tolist())) == len(flat) == 8 * 120
