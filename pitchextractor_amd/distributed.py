"""Data-parallel wiring: one process per GPU, gradients summed with RCCL over xGMI.

The reference has no distributed code at all; the partition is the natural one for
its workload: utterances are independent, weights are replicated, so each rank runs the whole
step on its own minibatch shard and the only exchange is one sum-all-reduce of the flat gradient
buffer per optimizer step (115.5 MB fp32), divided by the world size inside the fused AdamW
kernel.  BatchNorm keeps per-replica statistics, exactly as N independent reference processes
would (no SyncBN exists in the reference).

The flat gradient buffer is reduced in a few large buckets on RCCL's own stream: ring all-reduce
over point-to-point xGMI is per-link bound, so few large messages beat many small ones.  Buckets
are issued in backward-completion order (sequence models first: 93 % of the bytes) as soon as
JDCNet's backward signals them, from the stream that produced them (no private reducer stream:
see ``reduce_range``), and the optimizer waits on the last one.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def rehearse_single_rank() -> bool:
    """``PE_DP_REHEARSE=1``: run the whole data-parallel wiring (process group, broadcasts, bucketed all-reduce on
    the issuing stream, cross-rank flags) even at world size 1, so a one-GPU box executes the RCCL code path that
    the multi-GPU launch uses (a sum over one rank is the identity: results must equal the plain run bit for bit)."""
    return os.environ.get("PE_DP_REHEARSE", "0") == "1"


def init_from_env(backend: str | None = None):
    """torchrun-style rendezvous (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).  Returns (rank, world, local)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if "PE_FORCE_DEVICE" in os.environ:             # rehearsal on a 1-GPU box: every rank on one device
        local = int(os.environ["PE_FORCE_DEVICE"])
    if (world > 1 or rehearse_single_rank()) and not dist.is_initialized():
        if backend is None:
            backend = os.environ.get("PE_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl" and "GPU_MAX_HW_QUEUES" not in os.environ:
            import warnings
            warnings.warn("GPU_MAX_HW_QUEUES is unset: with the HIP default of 4 hardware queues a step that uses "
                          "RCCL's stream measured ~5 ms slower on MI355X; export GPU_MAX_HW_QUEUES=3 before the "
                          "process starts (bench.py / train.py do it themselves)")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous shard [lo, hi) of a global minibatch for this rank."""
    per = n_items // world
    return rank * per, (rank + 1) * per


class EpochShardSampler(torch.utils.data.Sampler):
    """``DistributedSampler``-equivalent index source: every epoch ONE seeded permutation of the
    whole list (identical on all ranks: ``seed + epoch``), truncated to a multiple of ``world * batch`` and cut
    into ``world`` equal CONTIGUOUS shards -- so every rank runs exactly the same number of optimizer steps
    (a rank with one batch more would wait forever in its last all-reduce) and the union of the shards is a
    different subset-free reshuffle each epoch.  ``shuffle=False`` (validation) keeps the list order.
    ``len(sampler) // batch`` is the per-rank steps per epoch that OneCycleLR must be built with (0 when the
    list cannot fill one batch on every rank: the caller must refuse to train)."""

    def __init__(self, n_items: int, batch: int, rank: int, world: int, seed: int = 0, shuffle: bool = True,
                 drop_last: bool = True):
        if not (0 <= rank < world) or batch < 1:
            raise ValueError("EpochShardSampler: need 0 <= rank < world and batch >= 1")
        self.n_items, self.batch, self.rank, self.world = int(n_items), int(batch), int(rank), int(world)
        self.seed, self.shuffle, self.epoch = int(seed), bool(shuffle), 0
        if drop_last:                    # training: equal shards of whole batches
            self.per_rank = (self.n_items // (self.world * self.batch)) * self.batch
            self.lo = self.rank * self.per_rank
        else:                            # validation (no collective inside the epoch): cover every item once
            base, extra = divmod(self.n_items, self.world)
            self.per_rank = base + (1 if self.rank < extra else 0)
            self.lo = self.rank * base + min(self.rank, extra)

    def set_epoch(self, epoch: int):
        self.epoch = int(epoch)

    def indices(self):
        if self.shuffle:
            g = torch.Generator()
            g.manual_seed(self.seed + self.epoch)
            order = torch.randperm(self.n_items, generator=g).tolist()
        else:
            order = list(range(self.n_items))
        return order[self.lo:self.lo + self.per_rank]

    def __iter__(self):
        return iter(self.indices())

    def __len__(self):
        return self.per_rank


def mean_over_ranks(values: dict, weight: float, device=None, group=None) -> dict:
    """Weighted mean of per-rank scalar dicts (same keys everywhere); logging only."""
    if not (dist.is_initialized() and (dist.get_world_size(group) > 1 or rehearse_single_rank())):
        return dict(values)
    keys = sorted(values)
    t = torch.tensor([float(values[k]) * weight for k in keys] + [float(weight)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    tot = float(t[-1].item())
    return {k: (float(t[i].item()) / tot if tot > 0 else float("nan")) for i, k in enumerate(keys)}


class GradientAllReduce:
    """Sum-all-reduce of a flat gradient buffer, in buckets, on a side stream.

    ``flat_grad`` is the model's flat gradient tensor (``JDCNet.flat_gradients()``);
    ``optimizer.grad_scale`` is set to 1/world so the mean is taken inside the AdamW kernel.
    ``payload="bf16"`` (or ``PE_DP_PAYLOAD=bf16``; BASELINE config[3]) halves the bytes on xGMI: each bucket is
    rounded to bf16 on the issuing stream, summed in bf16 by the collective and widened back into the fp32 buffer
    before the optimizer reads it (115.5 MB -> 57.8 MB per step; the sum of `world` bf16 terms carries ~8 bits, so
    this is an opt-in that belongs with mixed-precision training, never the fp32 parity path)."""

    def __init__(self, flat_grad: torch.Tensor, optimizer=None, group=None, bucket_bytes: int = 32 << 20,
                 flat_param: torch.Tensor | None = None, buffers=(), payload: str | None = None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.active = self.world > 1 or (dist.is_initialized() and rehearse_single_rank())
        self.flat_grad = flat_grad
        payload = (payload or os.environ.get("PE_DP_PAYLOAD") or "fp32").lower()
        if payload not in ("fp32", "bf16"):
            raise ValueError(f"GradientAllReduce: payload must be 'fp32' or 'bf16', not {payload!r}")
        self.payload = payload
        n = flat_grad.numel()
        per = max(1, bucket_bytes // (2 if payload == "bf16" else 4))
        self.bucket_elems = per
        self.buckets = [(lo, min(n, lo + per)) for lo in range(0, n, per)]
        self._pending = []
        self._issued = False
        self.messages = 0                   # collectives issued so far (tests / logging)
        private = flat_grad.is_cuda and os.environ.get("PE_DP_ISSUE", "side") == "reducer"
        self._stream = torch.cuda.Stream(device=flat_grad.device) if private else None
        if optimizer is not None:
            optimizer.grad_scale = 1.0 / self.world
        if self.active:
            if flat_param is not None:
                dist.broadcast(flat_param.data, src=0, group=group)     # identical replicas at step 0
            for b in buffers:
                dist.broadcast(b, src=0, group=group)

    def _all_reduce(self, chunk):
        """One bucket: fp32 in place, or through a bf16 staging tensor widened back by `finish`."""
        if self.payload == "bf16":
            stage = chunk.to(torch.bfloat16)
            work = dist.all_reduce(stage, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            return work, stage, chunk
        return dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group, async_op=True), None, chunk

    def reduce_range(self, lo: int, hi: int, after=None):
        """Start reducing gradient elements [lo, hi), in bucket-sized messages.  They must be final once the work
        queued so far on the current stream -- and on `after`, a second stream that also writes them (the model's
        weight-gradient side stream) -- has run.

        The collectives are ISSUED from `after` once it has been made to wait for the current stream (which every
        later piece of side work does anyway), or from the current stream when there is no `after`: RCCL's own
        stream then waits for exactly the work that produces the range and nothing here blocks the compute stream.
        A private reducer stream (``PE_DP_ISSUE=reducer``) measured 0.5-1 ms/step worse (tools/micro/dp_overhead.py);
        the large effect found there is the hardware-queue count, see bench.py / DESIGN section 5."""
        if not self.active or hi <= lo:
            return
        self._issued = True
        issue_on = None
        if self.flat_grad.is_cuda:
            cur = torch.cuda.current_stream(self.flat_grad.device)
            if self._stream is not None:
                self._stream.wait_stream(cur)
                if after is not None:
                    self._stream.wait_stream(after)
                issue_on = self._stream
            elif after is not None:
                after.wait_stream(cur)
                issue_on = after
            else:
                issue_on = cur
        per = max(1, self.bucket_elems)
        for a in range(lo, hi, per):
            chunk = self.flat_grad[a:min(hi, a + per)]
            if issue_on is not None:
                with torch.cuda.stream(issue_on):
                    self._pending.append(self._all_reduce(chunk))
            else:
                self._pending.append(self._all_reduce(chunk))
            self.messages += 1

    def any_rank(self, flag: bool) -> bool:
        """Logical OR of a per-rank status over the group (fault flags must be acted on by every rank)."""
        if not self.active:
            return bool(flag)
        t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=self.flat_grad.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return bool(int(t.item()))

    def any_rank_word(self, word) -> bool:
        """`any_rank` for a status that already lives on the device (`word`: 1-element integer tensor, or None when
        this process has none -- the same on every rank): max-reduced copy, ONE host read for flag and agreement."""
        if word is None:
            return False
        if not self.active:
            return bool(int(word.item()) != 0)
        t = (word != 0).to(torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return bool(int(t.item()))

    def finish(self):
        """Reduce whatever has not been issued yet and make the compute stream wait for all of it."""
        if not self.active:
            return
        if not self._issued:
            self.reduce_range(0, self.flat_grad.numel())
        for work, stage, chunk in self._pending:
            work.wait()                      # RCCL: the current stream waits for the collective; gloo: the host does
            if stage is not None:
                if stage.is_cuda:            # allocated on the issuing stream, read here on the compute stream
                    stage.record_stream(torch.cuda.current_stream(stage.device))
                chunk.copy_(stage)           # widen the bf16 sum back into the fp32 gradient buffer
        self._pending = []
        self._issued = False
        if self._stream is not None:
            torch.cuda.current_stream(self.flat_grad.device).wait_stream(self._stream)
