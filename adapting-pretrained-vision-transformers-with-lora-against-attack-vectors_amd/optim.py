"""Optimizer + data-parallel glue of the LoRA train step (train_loras.py:284, 308, 314-315).

`Adam` keeps torch.optim.Adam's constructor / zero_grad / step surface but updates the model's
single flat fp32 parameter with ONE fused kernel (vl_adam_step) and re-derives the fp16 GEMM
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


def allreduce_weighted_mean_(flat_grad: torch.Tensor, local_count, global_count, group=None) -> torch.Tensor:
    """Gradient of the GLOBAL batch's mean loss from per-rank gradients of the LOCAL shards' mean losses:
    sum_r (n_r / n) g_r -- one all-reduce; a rank with an empty shard passes zeros and n_r = 0."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        if local_count is not None and global_count:
            # ONE scale (n_r / n) and ONE collective (SUM) on the single flat buffer: the flat-gradient exchange of SURVEY 8e
            flat_grad.mul_(float(local_count) / float(global_count))
            dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=group)
        else:
            allreduce_mean_(flat_grad, group)
    return flat_grad


def shard_batch(n: int, rank: int, world: int):
    """Contiguous [start, stop) of a global batch of n images for `rank` (remainder to the first ranks)."""
    base, rem = divmod(n, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def global_batch_plan(n: int, batch: int, rank: int, world: int, shuffle_seed=None):
    """The step schedule of one epoch, IDENTICAL in length on every rank: the dataset is cut into global batches of
    `batch` samples (the reference's DataLoader batches, train_loras.py:237-243; shuffled with a seed every rank
    shares) and each global batch into contiguous per-rank shards.  Returns [(local_indices, n_global), ...] with one
    entry per optimizer step -- a rank whose shard of the last, ragged batch is empty still gets an entry (it
    contributes a zero gradient and joins the all-reduce), so no rank can run a step the others do not."""
    if shuffle_seed is None:
        order = list(range(n))
    else:
        order = torch.randperm(n, generator=torch.Generator().manual_seed(int(shuffle_seed))).tolist()
    plan = []
    for s0 in range(0, n, batch):
        gb = order[s0:s0 + batch]
        lo, hi = shard_batch(len(gb), rank, world)
        plan.append((gb[lo:hi], len(gb)))
    return plan


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
    def step(self, local_count: Optional[int] = None, global_count: Optional[int] = None):
        """optimizer.step().  Data parallel: pass the number of samples behind this rank's (mean) gradient and
        behind the whole global batch; the all-reduce then yields the gradient of the global-batch mean loss
        also when shards differ in size or a shard is empty (p.grad None = zero contribution)."""
        p = self.params[0]
        import torch.distributed as dist
        use_dist = self.distributed if self.distributed is not None else (dist.is_available() and dist.is_initialized())
        if p.grad is None:
            if not use_dist:
                return
            g = torch.zeros_like(p.data)            # empty shard: still join the collective
        else:
            g = p.grad.contiguous()
        if use_dist:
            # (in place: p.grad is the library's flat gradient buffer, rewritten by the next backward)
            allreduce_weighted_mean_(g, local_count, global_count, self.group)
        vit, eng = self._engine()
        self.t += 1
        lr = self.param_groups[0]["lr"]
        eng.adam_step(p.data, g, self.m1, self.m2, lr, self.betas[0], self.betas[1], self.eps, self.t)
        vit.mark_dirty()          # (vl_adam_step marked the handle itself: the fp16 operands are re-derived
                                  #  by the library before the next forward / attack)
        # p.grad IS the library's flat gradient buffer and the exchange above scaled / summed it in place: it is consumed by
        # this step (unlike torch.optim.Adam, which never writes .grad) -- a second step() without a new backward is a no-op
        # instead of re-applying a rescaled gradient
        p.grad = None

    def drop_last_step(self):
        """The step just taken was dropped by the fused kernel (every element of the reduced gradient non-finite: what
        train_loras.py injects to skip a step whose fp16 backward left the range): do not count it in Adam's bias correction
        -- an AMP skip-step never calls optimizer.step(), so torch.optim.Adam's step count does not advance either
        (round-4 ADVICE).  Identical on every rank: the NaN gradient rode the all-reduce."""
        if self.t > 0:
            self.t -= 1
