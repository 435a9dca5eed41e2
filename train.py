#!/usr/bin/env python3
"""Training entry point: same CLI and ``Configs/config.yml`` keys as the reference's train.py
(train.py:45-150), running every step on the HIP path.

    python train.py --config_path ./Configs/config.yml
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train.py -p cfg.yml

Data-parallel runs shard every minibatch over the ranks (one process per GPU: per-epoch seeded
permutation cut into equal contiguous rank shards) and all-reduce gradients with RCCL; rank 0 logs
and checkpoints.  TensorBoard is optional (absent -> scalars go to
the log file only).
"""
import logging
import os
import os.path as osp
import shutil
import sys
from logging import StreamHandler

# data-parallel runs: three HIP hardware queues (compute, weight-gradient side stream, RCCL), set before the
# runtime loads -- the default of four costs ~5 ms per step once RCCL's stream is in use (see bench.py)
if int(os.environ.get("WORLD_SIZE", "1")) > 1 or os.environ.get("PE_DP_REHEARSE") == "1":
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "3")

import click  # noqa: E402
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402
import yaml  # noqa: E402

from pitchextractor_amd import distributed as pdist  # noqa: E402
from pitchextractor_amd.meldataset import build_dataloader  # noqa: E402
from pitchextractor_amd.model import JDCNet  # noqa: E402
from pitchextractor_amd.optimizers import build_optimizer  # noqa: E402
from pitchextractor_amd.trainer import Trainer  # noqa: E402

logger = logging.getLogger(__name__)
logger.setLevel(logging.DEBUG)
handler = StreamHandler()
handler.setLevel(logging.DEBUG)
logger.addHandler(handler)


def get_data_path_list(train_path=None, val_path=None):
    train_path = train_path or "Data/train_list.txt"
    val_path = val_path or "Data/val_list.txt"
    with open(train_path, "r") as f:
        train_list = f.readlines()
    with open(val_path, "r") as f:
        val_list = f.readlines()
    return train_list, val_list


class _Scalars:
    """SummaryWriter when tensorboard is importable, otherwise a no-op with the same calls."""

    def __init__(self, path):
        try:
            from torch.utils.tensorboard import SummaryWriter
            self.w = SummaryWriter(path)
        except Exception:
            self.w = None

    def add_scalar(self, key, value, step):
        if self.w is not None:
            self.w.add_scalar(key, value, step)


@click.command()
@click.option("-p", "--config_path", default="./Configs/config.yml", type=str)
def main(config_path):
    config = yaml.safe_load(open(config_path))
    rank, world, local = pdist.init_from_env()
    log_dir = config["log_dir"]
    if rank == 0:
        os.makedirs(log_dir, exist_ok=True)
        shutil.copy(config_path, osp.join(log_dir, osp.basename(config_path)))
        file_handler = logging.FileHandler(osp.join(log_dir, "train.log"))
        file_handler.setLevel(logging.DEBUG)
        file_handler.setFormatter(logging.Formatter("%(levelname)s:%(asctime)s: %(message)s"))
        logger.addHandler(file_handler)
    writer = _Scalars(log_dir + "/tensorboard") if rank == 0 else None

    batch_size = config.get("batch_size", 32)
    device = config.get("device", "cuda")
    if torch.device(device).type != "cuda":
        raise SystemExit("this trainer runs on HIP devices only (config 'device' must be cuda)")
    device = f"cuda:{local}"
    torch.cuda.set_device(device)
    epochs = config.get("epochs", 100)
    save_freq = config.get("save_freq", 10)
    num_workers = config.get("num_workers", 8)
    training_config = config.get("training", {})

    train_list, val_list = get_data_path_list(config.get("train_data"), config.get("val_data"))
    # Data parallel: every rank holds the whole list; each epoch one seeded permutation is cut into equal
    # contiguous per-rank shards truncated to a common length (distributed.EpochShardSampler), so all ranks
    # run the same number of steps and OneCycleLR sees the same steps_per_epoch everywhere.
    seed = int(config.get("seed", 1234))
    dp_on = world > 1 or pdist.rehearse_single_rank()
    shard = (rank, world, seed) if dp_on else None
    per_rank_batch = max(1, batch_size // world)
    train_dataloader = build_dataloader(train_list, batch_size=per_rank_batch, num_workers=num_workers,
                                        dataset_config=config.get("dataset_params", {}), device=device, shard=shard)
    val_dataloader = build_dataloader(val_list, batch_size=per_rank_batch, validation=True,
                                      num_workers=num_workers // 2, device=device,
                                      dataset_config=config.get("dataset_params", {}), shard=shard)
    if len(train_dataloader) == 0:
        raise SystemExit(f"train list of {len(train_list)} files cannot fill one batch of {per_rank_batch} on "
                         f"each of {world} rank(s)")

    model_config = config.get("model_params", {})
    model = JDCNet(num_class=model_config.get("num_class", 1),
                   sequence_model_config=model_config.get("sequence_model", {}))
    scheduler_params = {
        "max_lr": float(config["optimizer_params"].get("lr", 5e-4)),
        "pct_start": float(config["optimizer_params"].get("pct_start", 0.0)),
        "epochs": epochs,
        "steps_per_epoch": len(train_dataloader),
    }
    model.to(device)
    model.dropout_cfg.seed = seed + rank           # replicas draw different dropout masks
    optimizer, scheduler = build_optimizer(
        {"params": model.parameters(), "optimizer_params": {}, "scheduler_params": scheduler_params})
    criterion = {"l1": nn.SmoothL1Loss(), "ce": nn.BCEWithLogitsLoss()}
    dp = None
    if dp_on:
        buffers = [b for b in model.buffers() if b.dtype.is_floating_point]
        # training.gradient_payload: "fp32" (default) or "bf16" buckets on xGMI (BASELINE config[3])
        dp = pdist.GradientAllReduce(model.flat_gradients(), optimizer, flat_param=model.flat_parameters,
                                     buffers=buffers, payload=training_config.get("gradient_payload"))
        model.attach_data_parallel(dp)
    trainer = Trainer(model=model, criterion=criterion, optimizer=optimizer, scheduler=scheduler, device=device,
                      train_dataloader=train_dataloader, val_dataloader=val_dataloader,
                      loss_config=config["loss_params"], logger=logger,
                      use_mixed_precision=training_config.get("mixed_precision", True),
                      amp_dtype=training_config.get("precision", "bf16"),
                      gradient_checkpointing=training_config.get("gradient_checkpointing", False),
                      checkpoint_use_reentrant=training_config.get("gradient_checkpointing_use_reentrant"),
                      data_parallel=dp)
    if config.get("pretrained_model", "") != "":
        trainer.load_checkpoint(config["pretrained_model"], load_only_params=config.get("load_only_params", True))

    for epoch in range(1, epochs + 1):
        results = trainer._train_epoch()
        evals = trainer._eval_epoch()
        if dp_on:                                   # logging only: every rank's batches count
            lr = results.pop("train/learning_rate")
            results = pdist.mean_over_ranks(results, len(train_dataloader), device=device)
            results["train/learning_rate"] = lr
            evals = pdist.mean_over_ranks(evals or {"eval/loss": 0.0, "eval/f0": 0.0, "eval/sil": 0.0},
                                          len(val_dataloader) if evals else 0, device=device)
        results.update(evals)
        if rank == 0:
            logger.info("--- epoch %d ---" % epoch)
            for key, value in results.items():
                if isinstance(value, float):
                    logger.info("%-15s: %.4f" % (key, value))
                    writer.add_scalar(key, value, epoch)
            if (epoch % save_freq) == 0:
                trainer.save_checkpoint(osp.join(log_dir, "epoch_%05d.pth" % epoch))
    return 0


if __name__ == "__main__":
    main()
