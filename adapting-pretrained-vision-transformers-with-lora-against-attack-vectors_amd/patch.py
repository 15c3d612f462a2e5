"""Adversarial-patch attack with expectation over transformations on the HIP engine -- the part of ART's
`AdversarialPatchPyTorch` that the reference uses (patch_attack.py:47-75 constructor arguments, :193-194 `generate`,
:199-208 `apply_patch`), same argument names.

Per optimiser step, all on the device: the host samples one (scale, rotation, shift) per image -- and, with
`distortion_scale_max > 0`, the four displaced corners of torchvision's `perspective` -- and builds the inverse affine
matrices (and homography coefficients); `vl_patch_apply` warps and pastes the patch; `vl_forward` / `vl_loss_ce` / `vl_backward_input` give
d(CE)/d(pixels); `vl_patch_grad` pulls it back onto the patch; with a process group the [3, ps, ps] gradient is summed
over ranks (12 KB all-reduce); `vl_adam_step` + `vl_clamp` update the patch.

ART and torchvision are not installable here: the algorithm is restated from their published sources and checked
against oracle/patch_oracle.py ("parity unpinned").
"""
from __future__ import annotations

import math
from typing import Optional, Sequence, Tuple

import torch


def inverse_affine_matrix(angle_deg: float, translate: Tuple[float, float], scale: float):
    """torchvision `_get_inverse_affine_matrix` for centre = image centre, shear 0 (what `affine(img, angle, translate,
    scale, shear=[0, 0])` uses): maps output pixel offsets from the centre to input offsets."""
    rot = math.radians(angle_deg)
    cs, sn = math.cos(rot), math.sin(rot)
    tx, ty = translate
    m = [cs / scale, sn / scale, 0.0, -sn / scale, cs / scale, 0.0]
    m[2] = m[0] * (-tx) + m[1] * (-ty)
    m[5] = m[3] * (-tx) + m[4] * (-ty)
    return m


def perspective_endpoints(S: int, distortion_scale: float, gen: torch.Generator):
    """ART `_random_overlay` (the corner draw of torchvision `RandomPerspective.get_params` at distortion_scale_max): the
    corners [[0, 0], [S-1, 0], [S-1, S-1], [0, S-1]] of the S x S canvas move inwards by up to int(distortion * S // 2)
    pixels per axis.  Returns [topleft, topright, botright, botleft] as integer [x, y] pairs."""
    half = S // 2
    d = int(distortion_scale * half)
    r = lambda lo, hi: int(torch.randint(lo, hi, (1,), generator=gen).item())
    return [[r(0, d + 1), r(0, d + 1)], [r(S - d - 1, S), r(0, d + 1)],
            [r(S - d - 1, S), r(S - d - 1, S)], [r(0, d + 1), r(S - d - 1, S)]]


def perspective_coeffs(S: int, endpoints):
    """torchvision `_get_perspective_coeffs(startpoints = the canvas corners, endpoints)`: the eight coefficients (a..h) that
    take a pixel of the WARPED canvas to its source, x_src = (a x + b y + c) / (g x + h y + 1), y_src = (d x + e y + f) / (...);
    least squares in float64, returned as float32 values like torchvision does."""
    start = [[0, 0], [S - 1, 0], [S - 1, S - 1], [0, S - 1]]
    a = torch.zeros(8, 8, dtype=torch.float64)
    for i, (p1, p2) in enumerate(zip(endpoints, start)):
        a[2 * i] = torch.tensor([p1[0], p1[1], 1, 0, 0, 0, -p2[0] * p1[0], -p2[0] * p1[1]], dtype=torch.float64)
        a[2 * i + 1] = torch.tensor([0, 0, 0, p1[0], p1[1], 1, -p2[1] * p1[0], -p2[1] * p1[1]], dtype=torch.float64)
    b = torch.tensor(start, dtype=torch.float64).view(8)
    return torch.linalg.lstsq(a, b, driver="gels").solution.to(torch.float32).tolist()


class AdversarialPatchPyTorch:
    """Drop-in for the calls patch_attack.py makes.  `estimator` is a vitlora model (optionally wrapped in LogitsModel /
    NormalizedModel / PeftModel); images are [0, 1] NCHW, the model is fed (x - mean) / std inside the patch gather."""

    def __init__(self, estimator, rotation_max: float = 22.5, scale_min: float = 0.1, scale_max: float = 1.0,
                 distortion_scale_max: float = 0.0, learning_rate: float = 5.0, max_iter: int = 500, batch_size: int = 16,
                 patch_shape: Sequence[int] = (3, 224, 224), patch_location: Optional[Tuple[int, int]] = None,
                 patch_type: str = "circle", optimizer: str = "Adam", targeted: bool = True, verbose: bool = True,
                 seed: int = 0, mean=None, std=None, process_group=None):
        from .attacks import _unwrap
        from .engine import IMAGENET_MEAN, IMAGENET_STD
        if not 0.0 <= distortion_scale_max < 1.0:
            raise ValueError("distortion_scale_max must be in [0, 1)")            # ART's own check
        if patch_type not in ("circle", "square"):
            raise ValueError("patch_type must be 'circle' or 'square'")
        if optimizer not in ("Adam", "pgd"):
            raise ValueError("optimizer must be 'Adam' or 'pgd'")
        if len(patch_shape) != 3 or patch_shape[0] != 3 or patch_shape[1] != patch_shape[2]:
            raise ValueError("patch_shape must be (3, ps, ps)")
        self.vit = _unwrap(estimator)
        self.eng = self.vit._engine()
        # the EoT step is forward / CE / backward-to-pixels through the plain entry points: per-GPU batches of up to 128 images run
        # as two half-batch chains there too (vl_debug_set_option "api_chains"; bit-identical, +7 % on ViT-L/16 at batch 128)
        self.eng.set_option("api_chains", 1)
        self.rotation_max, self.scale_min, self.scale_max = float(rotation_max), float(scale_min), float(scale_max)
        self.distortion_scale_max = float(distortion_scale_max)
        self.learning_rate, self.max_iter, self.batch_size = float(learning_rate), int(max_iter), int(batch_size)
        self.patch_shape, self.patch_location = tuple(patch_shape), patch_location
        self.patch_type, self.optimizer, self.targeted, self.verbose = patch_type, optimizer, bool(targeted), verbose
        self.mean, self.std = list(mean or IMAGENET_MEAN), list(std or IMAGENET_STD)
        self.group = process_group
        # two streams: the transformations drawn per image (rank-specific under a process group: every rank draws for its own
        # shard) and the batch order (the SAME on every rank, or the shards of a global batch would not belong together)
        rank = 0
        try:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                rank = dist.get_rank(process_group)
        except Exception:
            rank = 0
        self._gen = torch.Generator().manual_seed(int(seed) + 1000003 * rank)
        self._gen_order = torch.Generator().manual_seed(int(seed) + 7)
        ps = patch_shape[1]
        # ART: the patch starts at the middle of the classifier's clip_values = (0, 1)
        self._patch = torch.full((3, ps, ps), 0.5, dtype=torch.float32, device=self.eng.device)
        self._m1 = torch.zeros_like(self._patch)
        self._m2 = torch.zeros_like(self._patch)
        self._t = 0
        self.last_params = None
        self.skipped_steps = 0          # fp16 mode: optimiser steps dropped because a gradient left the fp16 range (telemetry)
        self.steps_taken = 0

    # -- sampling (host) --------------------------------------------------------------------------------------
    def sample_params(self, n: int, scale: Optional[float] = None):
        """ART `_random_overlay`: scale ~ U(scale_min, scale_max) unless given; shifts ~ U(-pad, pad), pad = (S - scale*S)/2
        (or fixed by patch_location); rotation ~ U(-rotation_max, rotation_max); with distortion_scale_max > 0 a fifth entry:
        the four displaced canvas corners of the perspective warp (`perspective_endpoints`)."""
        S = self.vit.arch.image_size
        ps = self.patch_shape[1]
        out = []
        for _ in range(n):
            u = torch.rand(4, generator=self._gen, dtype=torch.float64).tolist()
            sc = float(scale) if scale is not None else self.scale_min + (self.scale_max - self.scale_min) * u[0]
            if self.patch_location is None:
                pad = (S - sc * S) / 2.0
                tx, ty = (2 * u[2] - 1) * pad, (2 * u[3] - 1) * pad
            else:
                pad = int(math.floor(S - ps) / 2.0)
                tx, ty = -pad + self.patch_location[0], -pad + self.patch_location[1]
            prm = (sc, (2 * u[1] - 1) * self.rotation_max, tx, ty)
            if self.distortion_scale_max > 0.0:
                prm += (perspective_endpoints(S, self.distortion_scale_max, self._gen),)
            out.append(prm)
        return out

    def _matrices(self, params) -> torch.Tensor:
        m = [inverse_affine_matrix(p[1], (p[2], p[3]), p[0]) for p in params]
        return torch.tensor(m, dtype=torch.float32, device=self.eng.device)

    def _persp(self, params) -> Optional[torch.Tensor]:
        """[B, 8] homography coefficients when the parameter tuples carry corner displacements, else None (no distortion)."""
        if not params or all(len(p) < 5 for p in params):
            return None
        S = self.vit.arch.image_size
        ident = [[0, 0], [S - 1, 0], [S - 1, S - 1], [0, S - 1]]
        q = [perspective_coeffs(S, p[4] if len(p) > 4 else ident) for p in params]
        return torch.tensor(q, dtype=torch.float32, device=self.eng.device)

    # -- one optimiser step -----------------------------------------------------------------------------------
    def train_step(self, images: torch.Tensor, labels: torch.Tensor, params=None, global_count: int = 0) -> torch.Tensor:
        """ART `_train_step`: returns the CE of the patched batch before the update.  Under a process group `images` is this
        rank's shard of a global batch of `global_count` images (possibly empty): the patch gradient of the global mean loss
        is the shard-size-weighted sum of the ranks' gradients, ONE all-reduce of [3, ps, ps] floats."""
        eng = self.eng
        ptype = 1 if self.patch_type == "circle" else 0
        n_local = int(images.shape[0])
        if n_local:
            images = images.to(device=eng.device, dtype=torch.float32).contiguous()
            labels = labels.to(device=eng.device, dtype=torch.int64).contiguous()
            params = params if params is not None else self.sample_params(n_local)
            self.last_params = params
            mats, persp = self._matrices(params), self._persp(params)
            eng.set_normalization(self.mean, self.std)
            patched = eng.patch_apply(images, self._patch, mats, ptype, persp=persp)
            eng.forward(patched, normalise=True, train=False)
            ce = eng.loss_ce(labels)
            gx, _ = eng.backward(True, False, tuple(images.shape))
            g = eng.patch_grad(gx, mats, self.patch_shape[1], ptype, persp=persp)   # d CE / d patch (mean over the local images)
            # fp16 mode: a backward that left the fp16 range is flagged, never silent.  The flag is consumed HERE, in the step it
            # belongs to; the step is then dropped on EVERY rank -- a NaN gradient rides the one all-reduce below (as the LoRA
            # train step does, train_loras.py) -- and the optimisation goes on with the next batch's transformations.
            from ._lib import NonFiniteGradient
            try:
                eng.check()
            except NonFiniteGradient:
                g = torch.full_like(g, float("nan"))
        else:
            ce = torch.zeros((), device=eng.device)
            g = torch.zeros_like(self._patch)
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1:
            if global_count:
                g.mul_(n_local / float(global_count))
                dist.all_reduce(g, group=self.group)                      # 12 KB at ps = 32
            else:                                                         # equal shards assumed
                dist.all_reduce(g, group=self.group)
                g /= dist.get_world_size(self.group)
        if bool(torch.isnan(g).any()):
            self.skipped_steps += 1
            return ce
        self.steps_taken += 1
        # ART minimises loss = -CE (untargeted, Adam) or +CE (targeted); "pgd": patch += / -= lr * sign(grad)
        ascent = not self.targeted
        if self.optimizer == "pgd":
            self._patch.add_(torch.sign(g), alpha=self.learning_rate if ascent else -self.learning_rate)
        else:
            self._t += 1
            flat, gf = self._patch.view(-1), (-g if ascent else g).reshape(-1).contiguous()
            eng.adam_step(flat, gf, self._m1.view(-1), self._m2.view(-1), self.learning_rate, 0.9, 0.999, 1e-8, self._t)
        eng.clamp_(self._patch, 0.0, 1.0)
        return ce

    def generate(self, x, y, **kwargs):
        """`attack.generate(x=x_train, y=y_train)` (patch_attack.py:194): max_iter passes over the data in batches."""
        x = torch.as_tensor(x)
        y = torch.as_tensor(y)
        if y.dim() == 2:
            y = y.argmax(1)
        import torch.distributed as dist
        world, rank = 1, 0
        if dist.is_available() and dist.is_initialized():
            world, rank = dist.get_world_size(self.group), dist.get_rank(self.group)
        for it in range(self.max_iter):
            order = torch.randperm(x.shape[0], generator=self._gen_order)     # ART's DataLoader(shuffle=True); same on every rank
            for s0 in range(0, x.shape[0], self.batch_size):
                idx = order[s0:s0 + self.batch_size]
                if world > 1:                                                 # this rank's images of the global batch
                    ce = self.train_step(x[idx[rank::world]], y[idx[rank::world]], global_count=len(idx))
                else:
                    ce = self.train_step(x[idx], y[idx])
            if self.verbose and (it % 50 == 0 or it == self.max_iter - 1):
                print(f"  patch iter {it + 1}/{self.max_iter}: CE {float(ce):.4f}")
        mask = self.patch_mask()
        return self._patch.detach().cpu().numpy(), mask.numpy()

    def patch_mask(self) -> torch.Tensor:
        ps = self.patch_shape[1]
        if self.patch_type == "square":
            return torch.ones(3, ps, ps)
        lin = torch.linspace(-1, 1, ps)
        xg, yg = torch.meshgrid(lin, lin, indexing="ij")
        return (1 - torch.clamp((xg ** 2 + yg ** 2) ** 40, -1, 1)).expand(3, ps, ps).clone()

    def apply_patch(self, x, scale: float, patch_external=None, params=None):
        """`attack.apply_patch(images_np, scale=scale)` (patch_attack.py:204): random rotation / location, given scale."""
        xt = torch.as_tensor(x).to(device=self.eng.device, dtype=torch.float32).contiguous()
        patch = self._patch if patch_external is None else torch.as_tensor(patch_external).to(self.eng.device).float()
        params = params if params is not None else self.sample_params(xt.shape[0], scale=scale)
        self.last_params = params
        out = self.eng.patch_apply(xt, patch, self._matrices(params), 1 if self.patch_type == "circle" else 0,
                                   persp=self._persp(params))
        return out.cpu().numpy() if not isinstance(x, torch.Tensor) else out


class NormalizedModel(torch.nn.Module):
    """patch_attack.py:16-25: feeds (x - mean) / std to the wrapped model."""

    def __init__(self, model, mean, std):
        super().__init__()
        self.mean = torch.tensor(mean).view(1, 3, 1, 1)
        self.std = torch.tensor(std).view(1, 3, 1, 1)
        self.model = model

    def forward(self, x):
        return self.model((x - self.mean.to(x.device)) / self.std.to(x.device))


__all__ = ["AdversarialPatchPyTorch", "NormalizedModel", "inverse_affine_matrix", "perspective_endpoints", "perspective_coeffs"]
