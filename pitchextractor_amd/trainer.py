"""``Trainer`` drop-in (reference trainer.py:32-291) driving the HIP training step.

Same constructor keywords, same ``run(batch) -> {'loss','f0','sil'}``, ``_train_epoch``,
``_eval_epoch``, ``save_checkpoint`` / ``load_checkpoint`` (same checkpoint dict keys).  One
step = [optional on-device mel] -> JDCNet forward -> fused SmoothL1+BCE loss -> hand-written
backward -> [data-parallel gradient all-reduce] -> fused AdamW -> OneCycle scheduler; the three
scalars come back in a single device->host copy.
"""
from __future__ import annotations

import gc
import logging
import os
from collections import defaultdict

import numpy as np
import torch
from torch import nn

from . import ops

logger = logging.getLogger(__name__)
logger.setLevel(logging.DEBUG)

try:  # progress bars are optional plumbing
    from tqdm import tqdm
except Exception:  # pragma: no cover
    def tqdm(it, **_kw):
        return it


class GradScaler:
    """Dynamic loss scaling with torch.amp.GradScaler's defaults and update rule (reference trainer.py:64-102 builds
    one whenever AMP is on): scale 2^16, halved after a step whose gradients hold inf / nan (that optimizer step is
    skipped), doubled after 2000 consecutive clean steps.  Like the reference, its state is not checkpointed."""

    def __init__(self, init_scale=65536.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000):
        self.scale, self.growth_factor, self.backoff_factor = float(init_scale), growth_factor, backoff_factor
        self.growth_interval, self._good_steps = int(growth_interval), 0
        self.skipped_steps = 0

    def update(self, found_inf: bool):
        if found_inf:
            self.scale *= self.backoff_factor
            self._good_steps = 0
            self.skipped_steps += 1
        else:
            self._good_steps += 1
            if self._good_steps == self.growth_interval:
                self.scale *= self.growth_factor
                self._good_steps = 0


# Host-side hygiene for the step loop.  After the first training steps everything long-lived exists (torch and its
# ~1e6 Python objects, the model, optimizer state, workspaces); a generation-2 collection that walks all of it was
# measured at 115 ms on the bench host -- more than a whole step -- and lands wherever the allocation counters
# happen to trip.  Moving those objects to the permanent generation once (gc.freeze) leaves later collections only
# the objects the steps themselves create.  PE_GC_FREEZE=0 keeps the interpreter's default behaviour.
GC_FREEZE_AFTER_STEPS = 2 if os.environ.get("PE_GC_FREEZE", "1") != "0" else 0
_gc_frozen = False


def _settle_gc():
    global _gc_frozen
    if not _gc_frozen:
        gc.collect()
        gc.freeze()
        _gc_frozen = True


class Trainer(object):
    """Keyword-compatible with the reference constructor (trainer.py:33-48), plus ``mel_transform`` (on-device
    mel for raw-audio batches) and ``data_parallel`` (a ``distributed.GradientAllReduce``)."""

    def __init__(self, model=None, criterion=None, optimizer=None, scheduler=None, config={}, loss_config={},
                 device=torch.device("cpu"), logger=logger, train_dataloader=None, val_dataloader=None,
                 initial_steps=0, initial_epochs=0, use_mixed_precision=False, gradient_checkpointing=False,
                 checkpoint_use_reentrant=None, mel_transform=None, data_parallel=None, amp_dtype="bf16",
                 activation_storage=None):
        kind = torch.device(device).type if isinstance(device, (str, torch.device)) else "cpu"
        if kind != "cuda":
            raise RuntimeError("pitchextractor_amd.Trainer runs the HIP path only: device must be a HIP "
                               "('cuda') device; there is no CPU fallback")
        self._check_criterion(criterion)
        self.model, self.criterion = model, criterion
        self.optimizer, self.scheduler = optimizer, scheduler
        self.train_dataloader, self.val_dataloader = train_dataloader, val_dataloader
        self.config, self.loss_config = config, loss_config
        self.device, self.logger = device, logger
        self.steps, self.epochs = initial_steps, initial_epochs
        self.finish_train = False
        self.mel_transform = mel_transform
        self.data_parallel = data_parallel
        # The reference honours both flags only on an accelerator (trainer.py:63-64,103).
        # mixed_precision: inside a per-step scope every MFMA product -- conv forward / data gradient / weight
        # gradient, Linear and LSTM input projections and their weight gradients, and the persistent LSTM
        # recurrences (W_hh and the h / dgates rows) -- takes bf16-rounded operands and accumulates in fp32;
        # attention scores, normalisations, losses, the LSTM cell state and every tensor in HBM stay fp32.
        # bf16 (default) keeps the fp32 exponent, so no GradScaler is needed; ``amp_dtype="fp16"`` selects the
        # reference's literal autocast default: fp16 operands + GradScaler (trainer.py:64-102,241-244) -- the loss
        # gradients are scaled, non-finite gradients skip the update, the scale backs off / grows.
        # gradient_checkpointing: the reference's single whole-model segment (trainer.py:226-233) -- the forward
        # keeps no activations and backward recomputes it (JDCNet.checkpoint_forward).
        self.use_amp = bool(use_mixed_precision)
        amp_dtype = {"fp16": "f16", "float16": "f16", "half": "f16", "f16": "f16", "bf16": "bf16",
                     "bfloat16": "bf16"}.get(str(amp_dtype).lower())
        if amp_dtype is None:
            raise ValueError("amp_dtype must be 'bf16' or 'fp16'")
        self.amp_dtype = amp_dtype
        self.scaler = GradScaler() if (self.use_amp and amp_dtype == "f16") else None
        # activation_storage: "bf16" keeps the conv stack's activations and their gradients as bf16 tensors in HBM,
        # as autocast keeps conv outputs in half precision (trainer.py:226-235, README.md:36: the VRAM saving);
        # "fp32" rounds MFMA operands only.  Default: bf16 with bf16 mixed precision, fp32 otherwise.
        if activation_storage is None:
            activation_storage = os.environ.get("PE_ACT_STORAGE") or ("bf16" if (self.use_amp and amp_dtype == "bf16")
                                                                      else "fp32")
        if activation_storage not in ("bf16", "fp32"):
            raise ValueError("activation_storage must be 'bf16' or 'fp32'")
        self.act16 = activation_storage == "bf16" and self.use_amp and amp_dtype == "bf16"
        self.gradient_checkpointing = bool(gradient_checkpointing)
        self.gradient_checkpoint_use_reentrant = checkpoint_use_reentrant
        if self.use_amp:
            logger.info("mixed_precision: %s MFMA operands for every matmul-shaped product, fp32 accumulate and state%s",
                        "fp16" if amp_dtype == "f16" else "bf16", " + dynamic loss scaling" if self.scaler else "")
        if self.gradient_checkpointing:
            logger.info("gradient_checkpointing: whole-model segment, forward recomputed in backward")

    @staticmethod
    def _check_criterion(criterion):
        """The fused loss kernel implements exactly train.py:104-106; anything else must fail loudly."""
        if criterion is None:
            return
        l1, ce = criterion.get("l1"), criterion.get("ce")
        ok = (isinstance(l1, nn.SmoothL1Loss) and getattr(l1, "beta", 1.0) == 1.0 and l1.reduction == "mean"
              and isinstance(ce, nn.BCEWithLogitsLoss) and ce.reduction == "mean"
              and ce.pos_weight is None and ce.weight is None)
        if not ok:
            raise NotImplementedError("the HIP loss kernel implements SmoothL1Loss() + BCEWithLogitsLoss() "
                                      "(mean reduction) only")

    # ------------------------------------------------------------------ checkpoints (trainer.py:138-195)
    def save_checkpoint(self, checkpoint_path):
        state_dict = {
            "optimizer": self.optimizer.state_dict(),
            "scheduler": self.scheduler.state_dict(),
            "steps": self.steps,
            "epochs": self.epochs,
            "model": self.model.state_dict(),
        }
        folder = os.path.dirname(checkpoint_path)
        if folder and not os.path.exists(folder):
            os.makedirs(folder)
        torch.save(state_dict, checkpoint_path)

    def load_checkpoint(self, checkpoint_path, load_only_params=False):
        state_dict = torch.load(checkpoint_path, map_location="cpu", weights_only=True)
        self._load(state_dict["model"], self.model)
        if not load_only_params:
            self.steps = state_dict["steps"]
            self.epochs = state_dict["epochs"]
            self.optimizer.load_state_dict(state_dict["optimizer"])
            state_dict["scheduler"].update(**self.config.get("scheduler_params", {}))
            self.scheduler.load_state_dict(state_dict["scheduler"])

    def _load(self, states, model, force_load=True):
        """Tensor-by-tensor copy; unknown keys are skipped, mismatched shapes copy the overlap."""
        own = model.state_dict()
        for key, val in states.items():
            if key not in own:
                continue
            if isinstance(val, nn.Parameter):
                val = val.data
            dst = own[key]
            if val.shape != dst.shape:
                self.logger.info("%s does not have same shape" % key)
                if not force_load:
                    continue
                sl = tuple(slice(0, min(a, b)) for a, b in zip(val.shape, dst.shape))
                dst[sl].copy_(val[sl])
            else:
                dst.copy_(val)

    @staticmethod
    def get_gradient_norm(model):
        flat = model.flat_gradients() if hasattr(model, "flat_gradients") else None
        if flat is not None:
            return float(flat.double().norm().item())
        return float(np.sqrt(sum(p.grad.data.norm(2).item() ** 2 for p in model.parameters())))

    def _get_lr(self):
        for param_group in self.optimizer.param_groups:
            return param_group["lr"]

    # ------------------------------------------------------------------ one step (trainer.py:219-252)
    def _inputs(self, batch):
        batch = [b.to(self.device, non_blocking=True) for b in batch]
        x, f0, sil = batch
        if x.dim() == 2:                                   # raw audio (B, N): mel front end on the device
            if self.mel_transform is None:
                raise RuntimeError("batch carries raw audio but Trainer has no mel_transform")
            x = self.mel_transform.log_mel_batch(x, max_frames=f0.shape[-1])
        return x, f0.contiguous().float(), sil.contiguous().float()

    def _loss(self, f0_pred, sil_pred, f0, sil, want_grads, grad_scale=1.0):
        """num_class == 1: the reference's regression loss (train.py:104-106).  num_class > 1 has no loss in the
        reference; this build defines the CREPE-style bin classification of SURVEY 8f N4 for it
        (ops.f0_bins_ce_loss: voiced frames only, lambda_f0 * CE + BCE)."""
        lam = self.loss_config["lambda_f0"]
        if f0_pred.shape[-1] != 1:
            C = f0_pred.shape[-1]
            out4, d_logits, d_sil = ops.f0_bins_ce_loss(f0_pred.detach().reshape(-1, C), f0.reshape(-1),
                                                        sil_pred.detach().reshape(-1), sil.reshape(-1), lam,
                                                        grad_scale, want_grads)
            return out4[:3], d_logits, d_sil
        return ops.f0_sil_loss(f0_pred.detach().reshape(-1), f0.reshape(-1), sil_pred.detach().reshape(-1),
                               sil.reshape(-1), lam, grad_scale, want_grads)

    def _forward_backward(self, x, f0, sil):
        with ops.matmul_bf16(self.use_amp, self.amp_dtype, self.act16):
            self.model.checkpoint_forward = self.gradient_checkpointing
            try:
                f0_pred, sil_pred = self.model(x.transpose(-1, -2))
                # scaler.scale(loss).backward(): the loss kernel multiplies its gradients by the scale
                out3, d_f0, d_sil = self._loss(f0_pred, sil_pred, f0, sil, True,
                                               self.scaler.scale if self.scaler else 1.0)
                torch.autograd.backward([f0_pred, sil_pred], [d_f0.view_as(f0_pred), d_sil.view_as(sil_pred)])
            finally:
                self.model.checkpoint_forward = False
        if self.data_parallel is not None:
            self.data_parallel.finish()
        return out3

    def _lstm_fault(self, device, collective=True) -> bool:
        """True when a persistent-LSTM group barrier timed out in this step on ANY rank (then every rank's
        results are suspect or its peers would deadlock in the next all-reduce).  Clears the sticky word and
        switches this process to the one-launch-per-time-step kernels.  ``collective=False`` (evaluation: ranks
        may hold different numbers of validation batches, so nothing there may be a collective) looks at this
        rank's word only."""
        if collective and self.data_parallel is not None and self.data_parallel.active:
            bad = self.data_parallel.any_rank_word(ops.persistent_lstm_error_word(device))   # one host read
        else:
            bad = ops.persistent_lstm_error(device)
        if bad:
            ops.clear_persistent_lstm_error(device)
            ops.USE_PERSISTENT_LSTM = False
            self.logger.warning("persistent LSTM kernel: a group barrier timed out (workgroups not co-resident?); "
                                "falling back to the per-time-step kernels for the rest of this run")
        return bad

    def _apply_update(self):
        """optimizer.step() (through the GradScaler in fp16 mode).  The fused AdamW reads the model's status slot on
        the device and leaves everything untouched when it is set."""
        self._stepped = True
        if self.scaler is None:
            self.optimizer.step()
            return
        # scaler.step(optimizer); scaler.update()  (trainer.py:242-244)
        found = bool(ops.nonfinite_flag(self.model.flat_gradients()).item())
        if self.data_parallel is not None and self.data_parallel.active:
            found = self.data_parallel.any_rank(found)
        self._stepped = not found
        if not found:
            self.optimizer.loss_scale_inv = 1.0 / self.scaler.scale      # unscale inside the fused AdamW
            try:
                self.optimizer.step()
            finally:
                self.optimizer.loss_scale_inv = 1.0
        self.scaler.update(found)

    def run(self, batch):
        self._runs = getattr(self, "_runs", 0) + 1
        if self._runs == GC_FREEZE_AFTER_STEPS + 1:
            _settle_gc()
        self.optimizer.zero_grad(set_to_none=True)
        x, f0, sil = self._inputs(batch)
        slot_fn = getattr(self.model, "status_slot", None)
        if slot_fn is None or not hasattr(self.optimizer, "skip_flag"):
            # generic model / optimizer: the fault word is checked on the host before the update
            out3 = self._forward_backward(x, f0, sil)
            if self._lstm_fault(x.device):
                self.optimizer.zero_grad(set_to_none=True)
                out3 = self._forward_backward(x, f0, sil)
            self._apply_update()
            self.scheduler.step()
            loss, loss_f0, loss_sil = out3.tolist()
            return {"loss": loss, "f0": loss_f0, "sil": loss_sil}
        # JDCNet + FusedAdamW: ONE host synchronisation per step.  The persistent-LSTM fault word travels in the flat
        # gradient buffer (summed over ranks by the gradient all-reduce itself), the AdamW kernel is predicated on it
        # on the device, and the host reads it together with the three loss scalars after everything is queued.
        out3 = self._forward_backward(x, f0, sil)
        self.optimizer.skip_flag = slot_fn()
        self._apply_update()
        loss, loss_f0, loss_sil, fault = torch.cat([out3, self.optimizer.skip_flag]).tolist()
        if fault != 0.0:                                   # nothing was updated: redo the step on the safe kernels
            ops.clear_persistent_lstm_error(x.device)
            ops.USE_PERSISTENT_LSTM = False
            self.logger.warning("persistent LSTM kernel: a group barrier timed out (workgroups not co-resident?); "
                                "falling back to the per-time-step kernels for the rest of this run")
            if self._stepped:
                self.optimizer.undo_step_count()
            self.optimizer.zero_grad(set_to_none=True)
            out3 = self._forward_backward(x, f0, sil)
            self.optimizer.skip_flag = slot_fn()
            self._apply_update()
            loss, loss_f0, loss_sil, fault = torch.cat([out3, self.optimizer.skip_flag]).tolist()
            if fault != 0.0:
                raise RuntimeError("LSTM fault word still set after falling back to the per-time-step kernels")
        self.scheduler.step()
        return {"loss": loss, "f0": loss_f0, "sil": loss_sil}

    def _epoch(self, loader, tag, step_fn):
        sums = defaultdict(list)
        for batch in tqdm(loader, desc=f"[{tag}]"):
            for key, value in step_fn(batch).items():
                sums[f"{tag}/{key}"].append(value)
        return {key: float(np.mean(vals)) for key, vals in sums.items()}

    def _train_epoch(self):
        """One pass over ``train_dataloader`` -> {'train/loss','train/f0','train/sil','train/learning_rate'}."""
        self.epochs += 1
        self.model.train()
        if hasattr(self.train_dataloader, "set_epoch"):
            self.train_dataloader.set_epoch(self.epochs)
        out = self._epoch(self.train_dataloader, "train", self.run)
        out["train/learning_rate"] = self._get_lr()
        return out

    @torch.no_grad()
    def _eval_step(self, batch):
        x, f0, sil = self._inputs(batch)
        with ops.matmul_bf16(self.use_amp, self.amp_dtype, self.act16):
            f0_pred, sil_pred = self.model(x.transpose(-1, -2))
        out3, _, _ = self._loss(f0_pred, sil_pred, f0, sil, False)
        word = ops.persistent_lstm_error_word(x.device)        # this rank's word only: no collective in evaluation
        if word is None:
            loss, loss_f0, loss_sil = out3.tolist()
            return {"loss": loss, "f0": loss_f0, "sil": loss_sil}
        loss, loss_f0, loss_sil, fault = torch.cat([out3, word.float()]).tolist()     # one device->host copy
        if fault != 0.0 and self._lstm_fault(x.device, collective=False):
            with ops.matmul_bf16(self.use_amp, self.amp_dtype, self.act16):
                f0_pred, sil_pred = self.model(x.transpose(-1, -2))
            out3, _, _ = self._loss(f0_pred, sil_pred, f0, sil, False)
            loss, loss_f0, loss_sil = out3.tolist()
        return {"loss": loss, "f0": loss_f0, "sil": loss_sil}

    @torch.no_grad()
    def _eval_epoch(self):
        """One pass over ``val_dataloader`` in eval mode -> {'eval/loss','eval/f0','eval/sil'}."""
        self.model.eval()
        return self._epoch(self.val_dataloader, "eval", self._eval_step)
