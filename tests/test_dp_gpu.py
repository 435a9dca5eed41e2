"""Data-parallel step on the GPU box: 2 ranks share cuda:0 (gloo carries the CUDA gradient buffer; the
8-GPU RCCL run is the driver's), each runs the HIP trainer on its own shard.  Checks SURVEY 8(e)'s
oracle: the update equals AdamW on the MEAN of per-replica gradients (local-batch BatchNorm), and both
ranks end bit-identical."""
import logging
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

CFG = {"model_type": "bilstm", "num_layers": 2, "dropout": 0.0, "hidden_size": 64}


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _make(state, dev):
    from pitchextractor_amd.model import JDCNet
    from pitchextractor_amd.optimizers import build_optimizer
    net = JDCNet(num_class=1, sequence_model_config=dict(CFG))
    net.load_state_dict(state)
    net = net.to(dev).train()
    net.block_dropout = 0.0
    opt, sched = build_optimizer({"params": net.parameters(), "optimizer_params": {},
                                  "scheduler_params": {"max_lr": 3e-4, "pct_start": 0.0, "epochs": 10,
                                                       "steps_per_epoch": 10}})
    return net, opt, sched


def _batch(rank, dev):
    from tests.golden.make_golden import golden_input, golden_targets
    x = golden_input(40 + rank, B=2).transpose(-1, -2).contiguous()      # loader layout (B,1,80,T)
    f0, sil = golden_targets(40 + rank, B=2)
    return x.to(dev), f0.to(dev), sil.to(dev)


def _worker(rank, world, port, out_dir, amp=False, payload=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0")
    from oracle import model_ref
    from pitchextractor_amd import distributed as pdist
    from pitchextractor_amd import ops
    from pitchextractor_amd.trainer import Trainer
    pdist.init_from_env("gloo")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    state = model_ref.seeded_state(3, hidden_size=64, num_layers=2)
    net, opt, sched = _make(state, dev)
    dp = pdist.GradientAllReduce(net.flat_gradients(), opt, flat_param=net.flat_parameters, bucket_bytes=1 << 20,
                                 payload=payload)
    net.attach_data_parallel(dp)
    crit = {"l1": torch.nn.SmoothL1Loss(), "ce": torch.nn.BCEWithLogitsLoss()}
    tr = Trainer(model=net, criterion=crit, optimizer=opt, scheduler=sched, device="cuda:0",
                 loss_config={"lambda_f0": 0.1}, logger=logging.getLogger("dp"), data_parallel=dp,
                 use_mixed_precision=amp)
    tr.run(_batch(rank, dev))
    torch.cuda.synchronize()
    mine = net.flat_parameters.detach().cpu().numpy()
    np.save(os.path.join(out_dir, f"p{rank}.npy"), mine)
    # validation shards may differ by a batch between ranks: evaluation must not contain a collective
    net.eval()
    for _ in range(2 - rank):
        ev = tr._eval_step(_batch(rank, dev))
        assert np.isfinite(ev["loss"])
    net.train()
    if rank == 0:
        # oracle for DP: mean of the per-replica gradients, each from a fresh replica with local-batch BN
        grads = []
        for r in range(world):
            n2, _, _ = _make(state, dev)
            x, f0, sil = _batch(r, dev)
            with ops.matmul_bf16(amp, "bf16", act16=amp):          # (the trainer stores bf16 activations in this mode)
                cls, det = n2(x.transpose(-1, -2))
            _, d_f0, d_sil = ops.f0_sil_loss(cls.detach().reshape(-1), f0.reshape(-1), det.detach().reshape(-1),
                                             sil.reshape(-1), 0.1)
            with ops.matmul_bf16(amp, "bf16", act16=amp):          # (the trainer stores bf16 activations in this mode)
                torch.autograd.backward([cls, det], [d_f0.view_as(cls), d_sil.view_as(det)])
            grads.append(n2.flat_gradients().clone())
        ref, ropt, _ = _make(state, dev)
        ref._grad_views()
        if payload == "bf16":       # what the collective computes: bf16-rounded replicas summed in bf16
            total = (grads[0].to(torch.bfloat16) + grads[1].to(torch.bfloat16)).float()
        else:
            total = grads[0] + grads[1]
        ref.flat_gradients().copy_(total * 0.5)
        ropt.step()
        np.save(os.path.join(out_dir, "ref.npy"), ref.flat_parameters.detach().cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_step_equals_mean_gradient_update(tmp_path):
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    p0, p1, ref = (np.load(tmp_path / n) for n in ("p0.npy", "p1.npy", "ref.npy"))
    assert np.array_equal(p0, p1)                                   # replicas stay in lock-step
    # (g0 + g1) summed by gloo then scaled inside AdamW vs averaged first: a few ulps apart
    np.testing.assert_allclose(p0, ref, rtol=1e-5, atol=1e-7)
    assert not np.array_equal(p0, np.zeros_like(p0))


def test_two_rank_mixed_precision_step_with_bf16_buckets(tmp_path):
    """BASELINE config[3]'s shape on one GPU: mixed-precision (bf16 operand) replicas exchanging bf16 gradient
    buckets.  The update must equal AdamW on half the bf16 sum of the bf16-rounded per-replica gradients."""
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), True, "bf16"), nprocs=2, join=True)
    p0, p1, ref = (np.load(tmp_path / n) for n in ("p0.npy", "p1.npy", "ref.npy"))
    assert np.array_equal(p0, p1)
    np.testing.assert_allclose(p0, ref, rtol=1e-5, atol=1e-7)


def _rccl_worker(rank, world, port, out_dir):
    """World size 1 over backend "nccl" (= RCCL): the one-GPU box executes the communicator set-up, the broadcasts and
    the bucketed all-reduces on the reducer stream that the 8-GPU launch uses.  A sum over one rank is the identity."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      PE_DP_REHEARSE="1")
    from oracle import model_ref
    from pitchextractor_amd import distributed as pdist
    from pitchextractor_amd.trainer import Trainer
    pdist.init_from_env("nccl")
    assert dist.get_backend() == "nccl"
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    state = model_ref.seeded_state(3, hidden_size=64, num_layers=2)
    crit = {"l1": torch.nn.SmoothL1Loss(), "ce": torch.nn.BCEWithLogitsLoss()}
    result = {}
    for tag, payload in (("plain", None), ("fp32", "fp32"), ("bf16", "bf16")):
        net, opt, sched = _make(state, dev)
        dp = None
        if payload is not None:
            dp = pdist.GradientAllReduce(net.flat_gradients(), opt, flat_param=net.flat_parameters,
                                         buffers=[b for b in net.buffers() if b.dtype.is_floating_point],
                                         bucket_bytes=1 << 20, payload=payload)
            net.attach_data_parallel(dp)
            assert dp.active and net._dp_cuts is not None
        tr = Trainer(model=net, criterion=crit, optimizer=opt, scheduler=sched, device="cuda:0",
                     loss_config={"lambda_f0": 0.1}, logger=logging.getLogger("dp"), data_parallel=dp)
        for step in range(2):
            out = tr.run(_batch(step, dev))
            if step == 0:
                result[tag + "_grad"] = net.flat_gradients().detach().cpu().numpy()
        torch.cuda.synchronize()
        result[tag] = net.flat_parameters.detach().cpu().numpy()
        result[tag + "_loss"] = np.array(out["loss"])
        if dp is not None:
            # per step: heads, res_block3.., res_block2, res_block1, conv_block -- each range in >= 1 message
            result[tag + "_msgs"] = np.array(dp.messages)
    np.savez(os.path.join(out_dir, "rccl.npz"), **result)
    dist.barrier()
    dist.destroy_process_group()


def test_rccl_backend_world1_rehearsal(tmp_path):
    mp.spawn(_rccl_worker, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    r = np.load(tmp_path / "rccl.npz")
    assert np.array_equal(r["plain"], r["fp32"]) and r["plain_loss"] == r["fp32_loss"]      # RCCL sum over one rank
    assert r["fp32_msgs"] >= 2 * 5 and r["bf16_msgs"] >= 2 * 5
    # bf16 buckets: the gradient the optimizer saw in step 1 is exactly the fp32 gradient rounded to bf16
    g, gb = r["fp32_grad"], r["bf16_grad"]
    assert np.array_equal(r["plain_grad"], g) and not np.array_equal(g, gb)
    assert np.array_equal(gb, torch.from_numpy(g).to(torch.bfloat16).float().numpy())
    # two AdamW steps at lr 3e-4: an element whose gradient sits at the bf16 rounding of ~0 can move by lr in opposite
    # directions in each step (eps = 1e-9 makes the update sign-like there): 4 of 2.9 M elements differ by up to 2e-4
    np.testing.assert_allclose(r["bf16"], r["plain"], rtol=0, atol=7e-4)
    assert np.mean(np.abs(r["bf16"] - r["plain"]) > 1e-4) < 1e-5
