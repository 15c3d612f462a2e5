"""Attack facade with the reference's call signatures.

    batched_fgsm_attack(model, images, labels, epsilon, mean, std)   whitebox_attacks.py:22-38
    FGSM(model, eps) / PGD(model, eps, alpha, steps, random_start)   torchattacks, as used at
        .set_normalization_used(mean, std); attack(images, labels)   whitebox_attacks.py:110-113,169-170

All arithmetic runs in the HIP library: the forward/backward chain, the fused sign/project
step (vl_pgd_step), the random start (vl_pgd_init) and the whole PGD loop as one hipGraph per
iteration (vl_pgd_attack).
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch

from .model import LogitsModel, ViTForImageClassification


def _unwrap(model) -> ViTForImageClassification:
    seen = 0
    while not isinstance(model, ViTForImageClassification):
        nxt = getattr(model, "model", None) or getattr(model, "base_model", None)
        if nxt is None or seen > 4:
            raise TypeError("attack needs a vitlora ViTForImageClassification (optionally inside LogitsModel / PeftModel)")
        model, seen = nxt, seen + 1
    return model


def _is_imagenet(mean, std) -> bool:
    from .engine import IMAGENET_MEAN, IMAGENET_STD
    m = [float(v) for v in torch.as_tensor(mean).flatten().tolist()]
    s = [float(v) for v in torch.as_tensor(std).flatten().tolist()]
    return all(abs(a - b) < 1e-6 for a, b in zip(m, IMAGENET_MEAN)) and all(abs(a - b) < 1e-6 for a, b in zip(s, IMAGENET_STD))


def batched_fgsm_attack(model, images, labels, epsilon, mean, std):
    """clamp(x + eps * sign(dCE/dx), 0, 1) with the model fed (x - mean) / std
    (whitebox_attacks.py:22-38).  One forward, CE, backward-to-input and one fused step: exactly one iteration of the PGD
    loop from x with alpha = eps and no random start (bit-identical: tests/test_hip_fullsize.py::test_fgsm_is_pgd1_...), so
    it runs as that -- one captured graph, and at the reference's batch of 32 two half-batch chains (vl_pgd_attack)."""
    vit = _unwrap(model)
    eng = vit._engine()
    vit.sync_params()
    m = [float(v) for v in torch.as_tensor(mean).flatten().tolist()]
    s = [float(v) for v in torch.as_tensor(std).flatten().tolist()]
    eng.set_normalization(m, s)
    x = images.detach().to(device=eng.device, dtype=torch.float32).contiguous()
    return eng.pgd_attack(x, labels, float(epsilon), float(epsilon), 1, random_start=False)


class _Attack:
    """The part of torchattacks.Attack the reference touches."""

    def __init__(self, model):
        self.model = model
        self._norm: Optional[tuple] = None      # set_normalization_used(mean, std)

    def set_normalization_used(self, mean: Sequence[float], std: Sequence[float]):
        """torchattacks contract: the inputs handed to the attack ARE normalised with (mean, std);
        they are mapped back to [0,1], attacked there with the model fed normalised pixels, and
        the result is normalised again.  (The reference calls this on UN-normalised images,
        whitebox_attacks.py:169-170 -- reproduced here by construction.)"""
        self._norm = ([float(v) for v in mean], [float(v) for v in std])

    def _prepare(self, images):
        vit = _unwrap(self.model)
        eng = vit._engine()
        vit.sync_params()                  # adapters written through torch since the last call: never attack stale operands
        x = images.detach().to(device=eng.device, dtype=torch.float32).contiguous()
        if self._norm is not None:
            mean, std = self._norm
            eng.set_normalization(mean, std)
            x = eng.channel_affine(x, std, mean)                 # inverse_normalize: x*std + mean
        return vit, eng, x

    def _finish(self, eng, adv):
        if self._norm is not None:
            mean, std = self._norm
            adv = eng.channel_affine(adv, [1.0 / s for s in std], [-m / s for m, s in zip(mean, std)], out=adv)
        return adv

    def __call__(self, images, labels):
        return self.forward(images, labels)


class FGSM(_Attack):
    """torchattacks.FGSM(model, eps): adv = clamp(x + eps * sign(grad), 0, 1)."""

    def __init__(self, model, eps=8 / 255):
        super().__init__(model)
        self.eps = eps

    def forward(self, images, labels):
        vit, eng, x = self._prepare(images)
        if self._norm is not None:            # the PGD loop feeds the model normalised pixels: one iteration of it, alpha = eps
            return self._finish(eng, eng.pgd_attack(x, labels, float(self.eps), float(self.eps), 1, random_start=False))
        eng.forward(x, normalise=False, train=False)
        eng.loss_ce(labels)
        gx, _ = eng.backward(True, False, tuple(x.shape))
        adv = x.clone()
        eng.pgd_step(adv, x, gx, self.eps, self.eps, 0.0, 1.0)
        return self._finish(eng, adv)


class PGD(_Attack):
    """torchattacks.PGD(model, eps, alpha, steps, random_start) (whitebox_attacks.py:112-113).
    `seed` seeds the library's counter-based random start (torch's RNG stream is not used)."""

    def __init__(self, model, eps=8 / 255, alpha=2 / 255, steps=10, random_start=True, seed=0):
        super().__init__(model)
        self.eps, self.alpha, self.steps, self.random_start, self.seed = eps, alpha, steps, random_start, seed
        self._calls = 0

    def forward(self, images, labels):
        vit, eng, x = self._prepare(images)
        if self._norm is None:
            # no normalisation registered: the model consumes the adversarial image as is
            eng.set_normalization([0.0, 0.0, 0.0], [1.0, 1.0, 1.0])
        adv = eng.pgd_attack(x, labels, self.eps, self.alpha, self.steps, self.random_start,
                             seed=self.seed + self._calls)
        self._calls += 1
        if self._norm is None:
            from .engine import IMAGENET_MEAN, IMAGENET_STD
            eng.set_normalization(IMAGENET_MEAN, IMAGENET_STD)   # back to the library default
        return self._finish(eng, adv)


__all__ = ["batched_fgsm_attack", "FGSM", "PGD", "LogitsModel"]
