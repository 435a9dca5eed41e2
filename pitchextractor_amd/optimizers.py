"""``build_optimizer`` drop-in (reference optimizers.py:50-76) with a fused HIP AdamW.

Hyper-parameters are the reference's: AdamW(lr=optimizer_params.get('lr', 1e-4),
weight_decay=optimizer_params.get('weight_decay', 5e-4), betas=(0.9, 0.98), eps=1e-9) and
OneCycleLR(max_lr, epochs, steps_per_epoch, pct_start, final_div_factor=5) whose
``cycle_momentum`` default rewrites beta1 every step (0.85 -> 0.95).  The scheduler object is
torch's own host-side ``OneCycleLR`` (two scalars per step, identical ``state_dict``); the
parameter update itself is one HIP launch over the model's flat parameter buffer.
"""
from __future__ import annotations

import torch
from torch.optim import Optimizer

from . import ops


class FusedAdamW(Optimizer):
    """torch.optim.AdamW semantics, executed by ``pe_adamw_step``.

    When every parameter of a group is a view into one flat buffer (``JDCNet`` lays its
    parameters and gradients out that way) the whole group is updated by a single launch;
    otherwise one launch per tensor.  ``state_dict`` has AdamW's layout (step / exp_avg /
    exp_avg_sq per parameter).  ``grad_scale`` multiplies gradients inside the kernel
    (1/world_size after a summing all-reduce).
    """

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False,
                        maximize=False, foreach=None, capturable=False, differentiable=False, fused=None)
        super().__init__(params, defaults)
        self.grad_scale = 1.0
        self.loss_scale_inv = 1.0            # Trainer's GradScaler: gradients are unscaled inside the kernel
        self.skip_flag = None                # 1-element device float: non-zero when the kernel runs = no update
        self._flat_plans = {}

    # ---- flat-buffer detection -----------------------------------------------------------
    @staticmethod
    def _span(tensors):
        """(storage_ptr, first_elem, last_elem_exclusive, offsets) if all share one storage."""
        ptrs = {t.untyped_storage().data_ptr() for t in tensors}
        if len(ptrs) != 1:
            return None
        offs = [t.storage_offset() for t in tensors]
        lo = min(offs)
        hi = max(o + t.numel() for o, t in zip(offs, tensors))
        return ptrs.pop(), lo, hi, offs

    def _plan(self, gi, params):
        """Flat update plan for a group, or None."""
        if not all(p.is_contiguous() and p.dtype == torch.float32 and p.is_cuda for p in params):
            return None
        ps = self._span([p.data for p in params])
        gs = self._span([p.grad for p in params])
        if ps is None or gs is None:
            return None
        _, plo, phi, poffs = ps
        _, glo, ghi, goffs = gs
        if [o - plo for o in poffs] != [o - glo for o in goffs] or (phi - plo) != (ghi - glo):
            return None
        covered = sum((p.numel() + 3) // 4 * 4 for p in params)
        if covered < (phi - plo) or plo % 4 or glo % 4:
            return None                      # holes larger than the alignment padding: not our layout
        n = phi - plo
        key = (gi, params[0].data.untyped_storage().data_ptr(), plo, n)
        plan = self._flat_plans.get(key)
        if plan is None:
            base_p = params[0].data
            flat_p = torch.as_strided(base_p, (n,), (1,), plo)
            m = torch.zeros(n, dtype=torch.float32, device=base_p.device)
            v = torch.zeros(n, dtype=torch.float32, device=base_p.device)
            # adopt any state loaded through load_state_dict
            for p, off in zip(params, poffs):
                st = self.state.get(p)
                if st and "exp_avg" in st:
                    m[off - plo:off - plo + p.numel()].copy_(st["exp_avg"].reshape(-1))
                    v[off - plo:off - plo + p.numel()].copy_(st["exp_avg_sq"].reshape(-1))
            for p, off in zip(params, poffs):
                st = self.state[p]
                st.setdefault("step", torch.tensor(0.0))
                st["exp_avg"] = m[off - plo:off - plo + p.numel()].view_as(p)
                st["exp_avg_sq"] = v[off - plo:off - plo + p.numel()].view_as(p)
            plan = dict(flat_p=flat_p, m=m, v=v, n=n)
            self._flat_plans = {key: plan}
        g0 = params[0].grad
        plan["flat_g"] = torch.as_strided(g0, (n,), (1,), glo)
        return plan

    def load_state_dict(self, state_dict):
        """torch semantics, then drop the cached flat moment buffers: the next ``step`` re-adopts the loaded
        ``exp_avg`` / ``exp_avg_sq`` instead of continuing with the moments of the run so far."""
        super().load_state_dict(state_dict)
        self._flat_plans = {}

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            params = [p for p in group["params"] if p.grad is not None]
            if not params:
                continue
            beta1, beta2 = group["betas"]
            all_have_grad = len(params) == len(group["params"])
            plan = self._plan(gi, params) if all_have_grad else None
            if plan is not None:
                st0 = self.state[params[0]]
                step = int(st0["step"].item()) + 1
                ops.adamw_step(plan["flat_p"], plan["flat_g"], plan["m"], plan["v"], group["lr"], beta1, beta2,
                               group["eps"], group["weight_decay"], step, self.grad_scale * self.loss_scale_inv,
                               skip_flag=self.skip_flag)
                new_step = torch.tensor(float(step))
                for p in params:
                    self.state[p]["step"] = new_step
                continue
            for p in params:
                st = self.state[p]
                if "exp_avg" not in st:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                step = int(st["step"].item()) + 1
                pd = p.data if p.is_contiguous() else None
                if pd is None or not p.grad.is_contiguous():
                    raise RuntimeError("FusedAdamW needs contiguous parameters and gradients")
                ops.adamw_step(pd.view(-1), p.grad.view(-1), st["exp_avg"].view(-1), st["exp_avg_sq"].view(-1),
                               group["lr"], beta1, beta2, group["eps"], group["weight_decay"], step,
                               self.grad_scale * self.loss_scale_inv, skip_flag=self.skip_flag)
                st["step"] = torch.tensor(float(step))
        return loss

    def undo_step_count(self):
        """The last ``step`` was skipped on the device (``skip_flag`` was set): take its count back so the redone
        step uses the same bias corrections."""
        for group in self.param_groups:
            steps = {id(self.state[p]["step"]): self.state[p]["step"] for p in group["params"] if "step" in self.state.get(p, {})}
            for t in steps.values():
                t -= 1.0


def build_optimizer(parameters):
    optimizer, scheduler = _define_optimizer(parameters)
    return optimizer, scheduler


def _define_optimizer(params):
    optimizer_params = params["optimizer_params"]
    sch_params = params["scheduler_params"]
    optimizer = FusedAdamW(
        params["params"],
        lr=optimizer_params.get("lr", 1e-4),
        weight_decay=optimizer_params.get("weight_decay", 5e-4),
        betas=(0.9, 0.98),
        eps=1e-9)
    scheduler = _define_scheduler(optimizer, sch_params)
    return optimizer, scheduler


def _define_scheduler(optimizer, params):
    return torch.optim.lr_scheduler.OneCycleLR(
        optimizer,
        max_lr=params.get("max_lr", 5e-4),
        epochs=params.get("epochs", 200),
        steps_per_epoch=params.get("steps_per_epoch", 1000),
        pct_start=params.get("pct_start", 0.0),
        final_div_factor=5)
