"""Optimizer + data-parallel glue of the LoRA train step (train_loras.py:284, 308, 314-315).

`Adam` keeps torch.optim.Adam's constructor / zero_grad / step surface but updates the model's
single flat fp32 parameter with ONE fused kernel (vl_adam_step) and re-derives the bf16 GEMM
operands afterwards (vl_lora_commit).  With a process group, `step()` first sums the flat
gradient over ranks with ONE all-reduce (RCCL over xGMI on MI355X; gloo in CPU tests) and
divides by the world size: local losses are means over the local shard.
"""
from __future__ import annotations

from typing import Iterable, Optional

import torch


def allreduce_mean_(flat_grad: torch.Tensor, group=None) -> torch.Tensor:
    """In-place mean of a flat gradient buffer over the ranks of `group` (one collective)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=group)
        flat_grad.div_(dist.get_world_size(group))
    return flat_grad


def shard_batch(n: int, rank: int, world: int):
    """Contiguous [start, stop) of a global batch of n images for `rank` (remainder to the first ranks)."""
    base, rem = divmod(n, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


class Adam:
    """torch.optim.Adam(params, lr, betas, eps) for a vitlora model's flat parameter."""

    def __init__(self, params: Iterable[torch.nn.Parameter], lr=1e-3, betas=(0.9, 0.999), eps=1e-8,
                 weight_decay=0.0, model=None, process_group=None, distributed: Optional[bool] = None):
        if weight_decay:
            raise NotImplementedError("the reference trains with weight_decay = 0 (train_loras.py:284)")
        self.params = [p for p in params]
        if len(self.params) != 1 or self.params[0].dim() != 1:
            raise ValueError("expected the model's single flat parameter (model.parameters())")
        self.lr, self.betas, self.eps = lr, betas, eps
        self.model = model
        self.group = process_group
        self.distributed = distributed
        p = self.params[0]
        self.m1 = torch.zeros_like(p.data)
        self.m2 = torch.zeros_like(p.data)
        self.t = 0
        self.param_groups = [{"params": self.params, "lr": lr, "betas": betas, "eps": eps}]

    def zero_grad(self, set_to_none: bool = True):
        for p in self.params:
            p.grad = None if set_to_none else (p.grad.zero_() if p.grad is not None else None)

    def _engine(self):
        from .model import ViTForImageClassification
        m = self.model
        for _ in range(5):
            if isinstance(m, ViTForImageClassification):
                return m, m._engine()
            m = getattr(m, "_vit", None) or getattr(m, "model", None) or getattr(m, "base_model", None)
        raise TypeError("Adam needs model=<vitlora model or PeftModel>")

    @torch.no_grad()
    def step(self):
        p = self.params[0]
        if p.grad is None:
            return
        g = p.grad.contiguous()
        import torch.distributed as dist
        use_dist = self.distributed if self.distributed is not None else (dist.is_available() and dist.is_initialized())
        if use_dist:
            allreduce_mean_(g, self.group)
        vit, eng = self._engine()
        self.t += 1
        lr = self.param_groups[0]["lr"]
        eng.adam_step(p.data, g, self.m1, self.m2, lr, self.betas[0], self.betas[1], self.eps, self.t)
        vit.mark_dirty()          # bf16 operands are re-derived before the next forward
