"""CPU oracle for the adversarial-patch path (BASELINE config 5).  TEST INFRASTRUCTURE ONLY -- same rules as
vit_lora_oracle.py: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product.

PARITY UNPINNED.  The reference delegates this path to ART 1.20.1 (`AdversarialPatchPyTorch`, patch_attack.py:5-6,
47-75, 193-208) which is not installed here and cannot be fetched, and ART in turn calls torchvision's
`transforms.functional.resize / affine` (also absent).  What follows restates their published algorithms in plain
torch and cites the call sites in the reference; no golden vector from the real packages exists.

  * overlay            ART `_random_overlay`: patch and mask resized (bilinear) to the image size, then per image
                       `affine(angle, translate, scale)` (patch: bilinear, mask: nearest, zero fill) and
                       `images * (1 - mask) + patch * mask`; `_predictions` clips the result to clip_values = (0, 1).
  * perspective        distortion_scale_max > 0 (patch_attack.py:95): before the affine, mask and patch canvas both go through
                       torchvision `perspective(startpoints = canvas corners, endpoints; bilinear, fill None)`:
                       `_get_perspective_coeffs` (8 x 8 least squares) + `_perspective_grid` + grid_sample(zeros).
  * inverse matrix     torchvision `_get_inverse_affine_matrix` (centre = image centre, shear 0) + `_gen_affine_grid`
                       + `grid_sample(align_corners=False, padding_mode="zeros")`.
  * circular mask      ART `_get_circular_patch_mask`: 1 - clip((x^2 + y^2)^40, -1, 1) on linspace(-1, 1, ps)^2.
  * sampling           scale ~ U(scale_min, scale_max); shifts ~ U(-pad, pad) with pad = (S - scale * S) / 2;
                       angle ~ U(-rotation_max, rotation_max)          (patch_location None, no external mask)
  * train step         loss = -CE(model(overlay), y) for the untargeted Adam form (patch_attack.py:62,70-72 defaults),
                       torch.optim.Adam([patch], lr), then patch.clamp_(0, 1).
"""
from __future__ import annotations

import math
from typing import Callable, Optional, Tuple

import torch
import torch.nn.functional as F


def circular_mask(ps: int, sharpness: int = 40) -> torch.Tensor:
    x = torch.linspace(-1, 1, ps)
    y = torch.linspace(-1, 1, ps)
    xg, yg = torch.meshgrid(x, y, indexing="ij")
    z = (xg ** 2 + yg ** 2) ** sharpness
    return 1 - torch.clamp(z, -1, 1)


def base_mask(ps: int, patch_type: str) -> torch.Tensor:
    return circular_mask(ps) if patch_type == "circle" else torch.ones(ps, ps)


def inverse_affine_matrix(angle_deg: float, translate: Tuple[float, float], scale: float):
    """torchvision `_get_inverse_affine_matrix(center=[0, 0], angle, translate, scale, shear=[0, 0])`: the matrix that
    maps OUTPUT pixel offsets (from the image centre) to INPUT offsets."""
    rot = math.radians(angle_deg)
    a, b, c, d = math.cos(rot), -math.sin(rot), math.sin(rot), math.cos(rot)
    tx, ty = translate
    m = [d / scale, -b / scale, 0.0, -c / scale, a / scale, 0.0]
    m[2] += m[0] * (-tx) + m[1] * (-ty)
    m[5] += m[3] * (-tx) + m[4] * (-ty)
    return m


def affine(img: torch.Tensor, matrix, mode: str) -> torch.Tensor:
    """torchvision tensor `affine`: `_gen_affine_grid` + grid_sample(align_corners=False, zeros).  img [N, C, H, W]."""
    n, _, h, w = img.shape
    theta = torch.tensor(matrix, dtype=torch.float32).reshape(1, 2, 3)
    xs = torch.linspace(-w * 0.5 + 0.5, w * 0.5 + 0.5 - 1, w)
    ys = torch.linspace(-h * 0.5 + 0.5, h * 0.5 + 0.5 - 1, h)
    base = torch.empty(1, h, w, 3)
    base[..., 0] = xs
    base[..., 1] = ys.unsqueeze(-1)
    base[..., 2] = 1
    rescaled = theta.transpose(1, 2) / torch.tensor([0.5 * w, 0.5 * h])
    grid = base.view(1, h * w, 3).bmm(rescaled).view(1, h, w, 2).expand(n, h, w, 2)
    return F.grid_sample(img, grid, mode=mode, padding_mode="zeros", align_corners=False)


def perspective_coeffs(S: int, endpoints):
    """torchvision `_get_perspective_coeffs`: rows (x', y', 1, 0, 0, 0, -x x', -x y') / (0, 0, 0, x', y', 1, -y x', -y y') per
    corner pair ((x', y') = displaced corner, (x, y) = canvas corner), solved for the 8 coefficients in float64."""
    import numpy as np
    start = [(0, 0), (S - 1, 0), (S - 1, S - 1), (0, S - 1)]
    rows, rhs = [], []
    for (xe, ye), (xs, ys) in zip(endpoints, start):
        rows.append([xe, ye, 1, 0, 0, 0, -xs * xe, -xs * ye])
        rows.append([0, 0, 0, xe, ye, 1, -ys * xe, -ys * ye])
        rhs += [xs, ys]
    sol = np.linalg.lstsq(np.asarray(rows, dtype=np.float64), np.asarray(rhs, dtype=np.float64), rcond=None)[0]
    return [float(np.float32(v)) for v in sol]


def perspective(img: torch.Tensor, coeffs) -> torch.Tensor:
    """torchvision tensor `perspective` (bilinear, fill None): `_perspective_grid` evaluates the homography at pixel centres
    (x + .5, y + .5), normalises by half the canvas size and subtracts 1; grid_sample(align_corners=False, zeros)."""
    n, _, h, w = img.shape
    a, b, c, d, e, f, g, hh = coeffs
    xs = torch.arange(w, dtype=torch.float32) + 0.5
    ys = (torch.arange(h, dtype=torch.float32) + 0.5).unsqueeze(-1)
    den = g * xs + hh * ys + 1.0
    gx = (a * xs + b * ys + c) / (0.5 * w) / den - 1.0
    gy = (d * xs + e * ys + f) / (0.5 * h) / den - 1.0
    grid = torch.stack([gx, gy], dim=-1)[None].expand(n, h, w, 2)
    return F.grid_sample(img, grid, mode="bilinear", padding_mode="zeros", align_corners=False)


def overlay(images: torch.Tensor, patch: torch.Tensor, patch_type: str, params) -> torch.Tensor:
    """images [B,3,S,S] in [0,1]; patch [3,ps,ps]; params: B tuples (scale, angle_deg, x_shift, y_shift[, endpoints]) --
    endpoints = the four displaced canvas corners of the perspective warp (distortion_scale_max > 0).  Differentiable in
    `patch`."""
    B, _, S, _ = images.shape
    ps = patch.shape[-1]
    mask = base_mask(ps, patch_type).expand(3, ps, ps)[None]
    mask_r = F.interpolate(mask, size=(S, S), mode="bilinear", align_corners=False)
    patch_r = F.interpolate(patch[None], size=(S, S), mode="bilinear", align_corners=False)
    outs = []
    for i in range(B):
        sc, ang, tx, ty = params[i][:4]
        m = inverse_affine_matrix(ang, (tx, ty), sc)
        mk_i, pp_i = mask_r, patch_r
        if len(params[i]) > 4:
            q = perspective_coeffs(S, params[i][4])
            mk_i, pp_i = perspective(mask_r, q), perspective(patch_r, q)
        mk = affine(mk_i, m, "nearest")
        pp = affine(pp_i, m, "bilinear")
        outs.append(images[i:i + 1] * (1 - mk) + pp * mk)
    return torch.clamp(torch.cat(outs), 0.0, 1.0)


def sample_params(B: int, S: int, scale_min: float, scale_max: float, rotation_max: float, gen: torch.Generator,
                  scale: Optional[float] = None):
    """ART `_random_overlay` sampling (patch_location None): one (scale, angle, x_shift, y_shift) per image."""
    out = []
    for _ in range(B):
        u = torch.rand(4, generator=gen, dtype=torch.float64).tolist()
        sc = scale if scale is not None else scale_min + (scale_max - scale_min) * u[0]
        pad = (S - sc * S) / 2.0
        out.append((sc, (2 * u[1] - 1) * rotation_max, (2 * u[2] - 1) * pad, (2 * u[3] - 1) * pad))
    return out


def patch_loss_and_grad(model: Callable[[torch.Tensor], torch.Tensor], images, labels, patch, patch_type, params):
    """(CE of the patched batch, d CE / d patch): the gradient `loss.backward()` leaves on ART's `_patch` is the negative of
    this for the untargeted Adam form (loss = -CE)."""
    p = patch.clone().requires_grad_(True)
    logits = model(overlay(images, p, patch_type, params))
    ce = F.cross_entropy(logits, labels)
    (g,) = torch.autograd.grad(ce, p)
    return ce.detach(), g


def generate(model, images, labels, ps: int, patch_type: str, steps: int, lr: float, batch_size: int, scale_min: float,
             scale_max: float, rotation_max: float, seed: int, params_log: Optional[list] = None):
    """ART `generate` / `_train_step` (optimizer = "Adam", untargeted): patch starts at the middle of clip_values,
    max_iter passes over the data in batches, Adam ascent on CE, clamp to [0, 1] after every step."""
    patch = torch.full((3, ps, ps), 0.5, requires_grad=True)
    opt = torch.optim.Adam([patch], lr=lr)
    gen = torch.Generator().manual_seed(seed)
    S = images.shape[-1]
    for _ in range(steps):
        for s0 in range(0, images.shape[0], batch_size):
            xb, yb = images[s0:s0 + batch_size], labels[s0:s0 + batch_size]
            params = sample_params(xb.shape[0], S, scale_min, scale_max, rotation_max, gen)
            if params_log is not None:
                params_log.append(params)
            opt.zero_grad()
            loss = -F.cross_entropy(model(overlay(xb, patch, patch_type, params)), yb)
            loss.backward()
            opt.step()
            with torch.no_grad():
                patch.clamp_(0.0, 1.0)
    return patch.detach()
