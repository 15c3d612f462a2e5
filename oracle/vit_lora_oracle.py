"""CPU oracle for the ViT + LoRA + FGSM/PGD hot path.  TEST INFRASTRUCTURE ONLY.

This file is the checker, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it.  The shipped path (package ``..._amd``) calls the HIP kernels through
the C ABI of ``include/vitlora.h`` and fails loudly if the library is missing.

It is a plain-torch (CPU, fp32) restatement of the arithmetic that the reference
reaches through third-party packages.  Every function cites what it restates:

* ViT forward          -> HF ``ViTForImageClassification`` as built by the reference's
                          ``create_vit_model`` (Utils.py:84-90); op sequence per
                          transformers modeling_vit.py (patch embed :60-69, embeddings
                          :146-157, attention :164-189/:192-238, MLP :241-254, layer
                          :257-286, final LN + CLS head :385, :560-561).
* LoRA linear          -> peft ``LoraConfig`` semantics requested by
                          ``setup_peft_lora`` (train_loras.py:79-95):
                          y = W x + b + (alpha/r) * B(A(dropout(x))).
* FGSM                 -> ``batched_fgsm_attack`` (whitebox_attacks.py:22-38).
* PGD                  -> ``torchattacks.PGD`` call sites whitebox_attacks.py:112-113,
                          169-170 (package absent here: restated from its published
                          algorithm, SURVEY.md section 3.2 -- "parity unpinned").
* Adam                 -> ``torch.optim.Adam(lr=1e-4)`` (train_loras.py:284).
* save_images          -> clamp -> *255 -> uint8 truncation (Utils.py:106-113).

Pinning status (see DESIGN.md "Oracle"):
  * ViT forward / input-gradient and FGSM are PINNED: ``tests/golden/make_golden.py``
    runs the reference's own ``batched_fgsm_attack`` and HF ``ViTForImageClassification``
    in the build container and the committed vectors are re-checked by
    ``tests/test_oracle_golden.py``.
  * LoRA and PGD are "parity unpinned" (peft / torchattacks are not installable
    here); the LoRA trainable-parameter counts printed in infLora.ipynb:163,919
    are the only reference-held known answers and are tested.

``sim16=True`` inserts fp16 round-trips (``SIM_DTYPE``) at the places where the HIP
path stores 16-bit values (GEMM operands, saved activations, gradients between
kernels), so the kernels can be held to a much tighter tolerance than the
fp32-vs-fp16 precision gap would allow.  ``sim16`` may also be a set of site names
(``SITES``) to round only there: tools/error_budget.py uses it to price each hop.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, Iterable, List, Optional, Tuple

import torch
import torch.nn.functional as F

IMAGENET_MEAN = (0.485, 0.456, 0.406)   # Utils.py:92-93
IMAGENET_STD = (0.229, 0.224, 0.225)

# The six Linear modules of one HF-4.55.2 ViT layer, by the module path peft matches on.
LINEAR_MODULES = (
    ("q", "attention.attention.query"),
    ("k", "attention.attention.key"),
    ("v", "attention.attention.value"),
    ("o", "attention.output.dense"),
    ("fc1", "intermediate.dense"),
    ("fc2", "output.dense"),
)


@dataclass
class OracleConfig:
    image_size: int = 224
    patch_size: int = 16
    hidden: int = 768
    layers: int = 12
    heads: int = 12
    mlp: int = 3072
    num_labels: int = 21
    ln_eps: float = 1e-12          # configuration_vit.py:58

    @property
    def tokens(self) -> int:
        return (self.image_size // self.patch_size) ** 2 + 1

    @property
    def head_dim(self) -> int:
        return self.hidden // self.heads


@dataclass
class OracleLora:
    r: int = 8
    alpha: float = 16.0            # setup_peft_lora default, train_loras.py:79
    targets: Tuple[str, ...] = ("q", "k", "v", "o", "fc2")
    # (layer, short target) -> (A [r, in], B [out, r])
    ab: Dict[Tuple[int, str], Tuple[torch.Tensor, torch.Tensor]] = field(default_factory=dict)

    @property
    def scaling(self) -> float:
        return self.alpha / self.r


def resolve_targets(target_modules: Iterable[str]) -> Tuple[str, ...]:
    """peft's rule: a module is adapted when its name equals a target or ends with
    "." + target.  ``["query","key","value","output.dense"]`` (train_loras.py:81)
    therefore adapts q, k, v, the attention out-projection AND the MLP fc2."""
    out = []
    for short, path in LINEAR_MODULES:
        full = "vit.encoder.layer.0." + path
        if any(full == t or full.endswith("." + t) for t in target_modules):
            out.append(short)
    return tuple(out)


def layer_key(i: int, short: str) -> str:
    return f"vit.encoder.layer.{i}." + dict(LINEAR_MODULES)[short]


# ----------------------------------------------------------------------------
# weights
# ----------------------------------------------------------------------------
def init_weights(cfg: OracleConfig, seed: int = 0, std: float = 0.02) -> Dict[str, torch.Tensor]:
    """Seeded random-init weights with HF-4.55.2 state-dict keys (no pretrained
    weights are reachable offline).  N(0, std) like HF ``_init_weights``; LN gains
    are jittered around 1 and biases are non-zero so that no term is vacuous."""
    g = torch.Generator().manual_seed(seed)

    def rn(*shape, s=std):
        return torch.randn(*shape, generator=g) * s

    D, M, P = cfg.hidden, cfg.mlp, cfg.patch_size
    w = {
        "vit.embeddings.cls_token": rn(1, 1, D),
        "vit.embeddings.position_embeddings": rn(1, cfg.tokens, D),
        "vit.embeddings.patch_embeddings.projection.weight": rn(D, 3, P, P),
        "vit.embeddings.patch_embeddings.projection.bias": rn(D),
        "vit.layernorm.weight": 1.0 + rn(D, s=0.1),
        "vit.layernorm.bias": rn(D),
        "classifier.weight": rn(cfg.num_labels, D),
        "classifier.bias": rn(cfg.num_labels),
    }
    for i in range(cfg.layers):
        p = f"vit.encoder.layer.{i}."
        for short, path in LINEAR_MODULES:
            o, k = (M, D) if short == "fc1" else (D, M) if short == "fc2" else (D, D)
            w[p + path + ".weight"] = rn(o, k)
            w[p + path + ".bias"] = rn(o)
        for ln in ("layernorm_before", "layernorm_after"):
            w[p + ln + ".weight"] = 1.0 + rn(D, s=0.1)
            w[p + ln + ".bias"] = rn(D)
    return w


def init_lora(cfg: OracleConfig, r: int, alpha: float = 16.0,
              targets: Tuple[str, ...] = ("q", "k", "v", "o", "fc2"),
              seed: int = 1, b_std: float = 0.02) -> OracleLora:
    """A ~ kaiming-uniform(a=sqrt 5) as peft does; B ~ N(0, b_std) instead of peft's
    zeros when b_std > 0 (zero B makes the LoRA branch vacuous)."""
    g = torch.Generator().manual_seed(seed)
    lora = OracleLora(r=r, alpha=alpha, targets=tuple(targets))
    for i in range(cfg.layers):
        for short in targets:
            o, k = (cfg.mlp, cfg.hidden) if short == "fc1" else \
                   (cfg.hidden, cfg.mlp) if short == "fc2" else (cfg.hidden, cfg.hidden)
            bound = 1.0 / math.sqrt(k)            # kaiming_uniform_(a=sqrt(5)) on [r, k]
            A = (torch.rand(r, k, generator=g) * 2 - 1) * bound
            B = torch.randn(o, r, generator=g) * b_std if b_std > 0 else torch.zeros(o, r)
            lora.ab[(i, short)] = (A, B)
    return lora


def count_parameters(cfg: OracleConfig, lora: Optional[OracleLora]) -> Tuple[int, int]:
    """(trainable, total) as peft's print_trainable_parameters reports for
    task_type=SEQ_CLS (classifier kept trainable AND counted twice: the frozen
    original plus the modules_to_save copy).  Known answers: infLora.ipynb:163,919."""
    D, M, P, C = cfg.hidden, cfg.mlp, cfg.patch_size, cfg.num_labels
    per_layer = 4 * (D * D + D) + (M * D + M) + (D * M + D) + 4 * D
    base = D + cfg.tokens * D + D * 3 * P * P + D + cfg.layers * per_layer + 2 * D
    head = D * C + C
    if lora is None:
        return base + head, base + head
    lp = 0
    for short in lora.targets:
        o, k = (M, D) if short == "fc1" else (D, M) if short == "fc2" else (D, D)
        lp += lora.r * (o + k)
    lp *= cfg.layers
    return lp + head, base + head + lp + head


# ----------------------------------------------------------------------------
# forward
# ----------------------------------------------------------------------------
SIM_DTYPE = torch.float16      # the 16-bit storage type of the HIP path (tools/error_budget.py also tries bfloat16)


class _RoundBF16(torch.autograd.Function):
    """16-bit round trip applied to the value in forward AND to the gradient in
    backward: models a tensor (and its gradient) that the HIP path stores in 16 bits."""

    @staticmethod
    def forward(ctx, x):
        return x.to(SIM_DTYPE).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g.to(SIM_DTYPE).to(torch.float32)


class _RoundFwdOnly(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.to(SIM_DTYPE).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g


class _RoundBwdOnly(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.to(SIM_DTYPE).to(torch.float32)


# ``sim`` is False (fp32 arithmetic), True (every site rounded) or a collection of site names:
# only those sites are rounded (error-budget experiments, tools/error_budget.py).  A name with the
# suffix ":f" / ":b" rounds only the forward value / only the gradient at that site.
# "resid" (round 4): the residual stream and the residual-gradient stream are 16-bit tensors on the HIP path.
SITES = ("patches", "h", "qkv", "probs", "ctx", "delta", "act", "gelu_prime", "dz", "t", "weights", "lora_w", "resid")


def _on(sim, site):
    if isinstance(sim, bool):
        return (sim, sim)
    return (site in sim or site + ":f" in sim, site in sim or site + ":b" in sim)


def _rb(x, sim, site="all"):          # value and gradient rounded
    f, b = _on(sim, site)
    if f and b:
        return _RoundBF16.apply(x)
    if f:
        return _RoundFwdOnly.apply(x)
    if b:
        return _RoundBwdOnly.apply(x)
    return x


def _rf(x, sim, site="lora_w"):       # value rounded, gradient passes
    return _RoundFwdOnly.apply(x) if _on(sim, site)[0] else x


def _wq(w, sim):          # frozen weight as the kernels hold it
    return w.to(SIM_DTYPE).to(torch.float32) if _on(sim, "weights")[0] else w


class _GeluSim(torch.autograd.Function):
    """exact-erf GELU as the HIP fc1 epilogue / fc2-dgrad epilogue pair computes it: the forward
    keeps a = gelu(z) and g' = gelu'(z) (each rounded to 16 bits when its site is on); the backward is
    dz = da * g' with da still in the fp32 accumulator, rounded once (site "dz")."""

    @staticmethod
    def forward(ctx, z, r_act, r_gp, r_dz):
        cdf = 0.5 * (1.0 + torch.erf(z * 0.7071067811865476))
        pdf = 0.3989422804014327 * torch.exp(-0.5 * z * z)
        gp = cdf + z * pdf
        if r_gp:
            gp = gp.to(SIM_DTYPE).to(torch.float32)
        ctx.save_for_backward(gp)
        ctx.r_dz = r_dz
        a = z * cdf
        return a.to(SIM_DTYPE).to(torch.float32) if r_act else a

    @staticmethod
    def backward(ctx, da):
        (gp,) = ctx.saved_tensors
        dz = da * gp
        if ctx.r_dz:
            dz = dz.to(SIM_DTYPE).to(torch.float32)
        return dz, None, None, None


def gelu_sim(z, sim):
    if sim is False:
        return F.gelu(z)
    return _GeluSim.apply(z, _on(sim, "act")[0], _on(sim, "gelu_prime")[0], _on(sim, "dz")[0])


def lora_linear(x, W, b, ab, scaling, sim=False, drop_mask=None):
    """y = x W^T + b + scaling * ((x*mask) A^T) B^T   (peft Linear.forward; the
    dropout mask, when given, is the already-scaled keep mask 1/(1-p) or 0)."""
    y = F.linear(x, _wq(W, sim), b)
    if ab is not None:
        A, B = ab
        xd = x if drop_mask is None else x * drop_mask
        t = _rb(F.linear(xd, _rf(A, sim)), sim, "t")
        y = y + F.linear(t, _rf(B * scaling, sim))
    return y


def vit_forward(w: Dict[str, torch.Tensor], cfg: OracleConfig, x_norm: torch.Tensor,
                lora: Optional[OracleLora] = None, sim16: bool = False,
                return_hidden: bool = False, trace: Optional[dict] = None,
                drop_masks: Optional[dict] = None):
    """logits [B, C] from already-normalised pixels [B, 3, H, W].  ``trace`` (a dict)
    receives the intermediate tensors the HIP path exposes through vl_debug_tensor."""
    sim = sim16
    B = x_norm.shape[0]
    D, H, dh, N = cfg.hidden, cfg.heads, cfg.head_dim, cfg.tokens
    P = cfg.patch_size
    # K2 patchify + embed (Conv2d k=s=P == GEMM over flattened (c,ph,pw) patches)
    g = cfg.image_size // P
    patches = x_norm.reshape(B, 3, g, P, g, P).permute(0, 2, 4, 1, 3, 5).reshape(B, g * g, 3 * P * P)
    patches = _rb(patches, sim, "patches")
    Wpe = w["vit.embeddings.patch_embeddings.projection.weight"].reshape(D, 3 * P * P)
    emb = F.linear(patches, _wq(Wpe, sim), w["vit.embeddings.patch_embeddings.projection.bias"])
    x = torch.cat([w["vit.embeddings.cls_token"].expand(B, -1, -1), emb], dim=1)
    x = _rb(x + w["vit.embeddings.position_embeddings"], sim, "resid")
    if trace is not None:
        trace["xs0"] = x.detach()
    sc = lora.scaling if lora is not None else 0.0

    def ab(i, short):
        return lora.ab.get((i, short)) if lora is not None else None

    for i in range(cfg.layers):
        p = f"vit.encoder.layer.{i}."

        def lin(short, inp):
            k = p + dict(LINEAR_MODULES)[short]
            # train-mode LoRA dropout: explicit keep-masks (0 or 1/(1-p)) keyed by (layer, projection);
            # q, k, v share the mask of the fused qkv projection's input
            dm = None
            if drop_masks is not None:
                dm = drop_masks.get((i, "qkv" if short in ("q", "k", "v") else short))
            return lora_linear(inp, w[k + ".weight"], w[k + ".bias"], ab(i, short), sc, sim, dm)

        h = _rb(F.layer_norm(x, (D,), w[p + "layernorm_before.weight"],
                             w[p + "layernorm_before.bias"], cfg.ln_eps), sim, "h")
        q = _rb(lin("q", h), sim, "qkv").view(B, N, H, dh).transpose(1, 2)
        k_ = _rb(lin("k", h), sim, "qkv").view(B, N, H, dh).transpose(1, 2)
        v = _rb(lin("v", h), sim, "qkv").view(B, N, H, dh).transpose(1, 2)
        s = torch.matmul(q, k_.transpose(2, 3)) * (dh ** -0.5)
        pr = _rb(torch.softmax(s, dim=-1), sim, "probs")
        ctx = _rb(torch.matmul(pr, v).transpose(1, 2).reshape(B, N, D), sim, "ctx")
        if trace is not None:
            trace[f"qkv{i}"] = torch.cat([t_.transpose(1, 2).reshape(B, N, D) for t_ in (q, k_, v)], dim=-1).detach()
            trace[f"ctx{i}"] = ctx.detach()
        # projection output held in 16 bits, added by the next LN pass; the sum is stored in 16 bits (and normalised as stored)
        x = _rb(x + _rb(lin("o", ctx), sim, "delta"), sim, "resid")
        if trace is not None:
            trace[f"xs{2 * i + 1}"] = x.detach()
        h2 = _rb(F.layer_norm(x, (D,), w[p + "layernorm_after.weight"],
                              w[p + "layernorm_after.bias"], cfg.ln_eps), sim, "h")
        a = gelu_sim(lin("fc1", h2), sim)          # exact erf GELU (hidden_act="gelu")
        x = _rb(x + _rb(lin("fc2", a), sim, "delta"), sim, "resid")
        if trace is not None:
            trace[f"xs{2 * i + 2}"] = x.detach()
    xf = F.layer_norm(x[:, 0], (D,), w["vit.layernorm.weight"], w["vit.layernorm.bias"], cfg.ln_eps)
    logits = F.linear(xf, w["classifier.weight"], w["classifier.bias"])
    return (logits, x) if return_hidden else logits


def normalise(x, mean=IMAGENET_MEAN, std=IMAGENET_STD):
    m = torch.tensor(mean, dtype=x.dtype).view(1, 3, 1, 1)
    s = torch.tensor(std, dtype=x.dtype).view(1, 3, 1, 1)
    return (x - m) / s


def loss_and_input_grad(w, cfg, x01, labels, lora=None, sim16=False, normalised=False):
    """(mean CE loss, dLoss/dx, logits) with x in [0,1] pixel space (normalisation
    inside) -- whitebox_attacks.py:24-30 -- or in model space when normalised=True."""
    x = x01.clone().detach().requires_grad_(True)
    logits = vit_forward(w, cfg, x if normalised else normalise(x), lora, sim16)
    loss = F.cross_entropy(logits, labels)
    (gx,) = torch.autograd.grad(loss, x)
    return loss.detach(), gx, logits.detach()


# ----------------------------------------------------------------------------
# attacks
# ----------------------------------------------------------------------------
def pgd_step(adv, x0, grad, eps, alpha, lo=0.0, hi=1.0):
    """adv <- clamp(x0 + clamp(adv + alpha*sign(g) - x0, -eps, eps), lo, hi)  (K10)."""
    adv = adv + alpha * torch.sign(grad)
    delta = torch.clamp(adv - x0, min=-eps, max=eps)
    return torch.clamp(x0 + delta, min=lo, max=hi)


def fgsm(w, cfg, x01, labels, eps, lora=None, sim16=False):
    """whitebox_attacks.py:22-38: clamp(x + eps*sign(dCE/dx), 0, 1)."""
    _, g, _ = loss_and_input_grad(w, cfg, x01, labels, lora, sim16)
    return torch.clamp(x01 + eps * torch.sign(g), 0.0, 1.0)


def pgd(w, cfg, x01, labels, eps, alpha, steps, lora=None, noise=None, sim16=False,
        return_trace=False):
    """Canonical torchattacks.PGD.forward.  ``noise`` in [-1,1] is the
    supplied random start (scaled by eps); None = random_start False."""
    adv = x01.clone()
    if noise is not None:
        adv = torch.clamp(adv + eps * noise, 0.0, 1.0)
    trace = []
    for _ in range(steps):
        loss, g, _ = loss_and_input_grad(w, cfg, adv, labels, lora, sim16)
        adv = pgd_step(adv, x01, g, eps, alpha)
        if return_trace:
            trace.append((loss.item(), g))
    return (adv, trace) if return_trace else adv


def pgd_torchattacks_compat(w, cfg, x01, labels, eps, alpha, steps, lora=None, noise=None,
                            mean=IMAGENET_MEAN, std=IMAGENET_STD):
    """The reference's actual call pattern: set_normalization_used(mean,std) on
    UN-normalised inputs (whitebox_attacks.py:169-170).  torchattacks then
    inverse-normalises the input (x*std+mean), attacks in that space feeding the
    model (x'-mean)/std, and re-normalises the result."""
    m = torch.tensor(mean).view(1, 3, 1, 1)
    s = torch.tensor(std).view(1, 3, 1, 1)
    xp = x01 * s + m
    adv = xp.clone()
    if noise is not None:
        adv = torch.clamp(adv + eps * noise, 0.0, 1.0)
    for _ in range(steps):
        _, g, _ = loss_and_input_grad(w, cfg, adv, labels, lora)   # model sees (adv-m)/s
        adv = pgd_step(adv, xp, g, eps, alpha)
    return (adv - m) / s


# ----------------------------------------------------------------------------
# LoRA training step, Adam, image quantisation
# ----------------------------------------------------------------------------
def lora_train_grads(w, cfg, x_norm, labels, lora: OracleLora, sim16=False,
                     train_classifier=True, drop_masks=None):
    """Gradients of mean CE w.r.t. every LoRA A, B (and the classifier, which peft
    keeps trainable for SEQ_CLS) -- the backward of train_loras.py:310-314."""
    leaves = {}
    lr = OracleLora(r=lora.r, alpha=lora.alpha, targets=lora.targets)
    for key, (A, B) in lora.ab.items():
        A2, B2 = A.clone().requires_grad_(True), B.clone().requires_grad_(True)
        lr.ab[key] = (A2, B2)
        leaves[("A",) + key] = A2
        leaves[("B",) + key] = B2
    w2 = dict(w)
    if train_classifier:
        w2["classifier.weight"] = w["classifier.weight"].clone().requires_grad_(True)
        w2["classifier.bias"] = w["classifier.bias"].clone().requires_grad_(True)
        leaves[("cls", "weight")] = w2["classifier.weight"]
        leaves[("cls", "bias")] = w2["classifier.bias"]
    logits = vit_forward(w2, cfg, x_norm, lr, sim16, drop_masks=drop_masks)
    loss = F.cross_entropy(logits, labels)
    grads = torch.autograd.grad(loss, list(leaves.values()))
    return loss.detach(), logits.detach(), dict(zip(leaves.keys(), grads))


def adam_step(p, g, m, v, t, lr=1e-4, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam single-tensor update (no weight decay, no amsgrad); t is 1-based."""
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    bc1 = 1 - b1 ** t
    bc2 = 1 - b2 ** t
    denom = v.sqrt() / math.sqrt(bc2) + eps
    p = p - (lr / bc1) * (m / denom)
    return p, m, v


def save_images_quant(images: torch.Tensor) -> torch.Tensor:
    """Utils.py:106-113: clamp(0,1) -> HWC -> *255 -> astype(uint8) (truncation)."""
    x = torch.clamp(images, 0, 1).permute(0, 2, 3, 1)
    return (x * 255).to(torch.uint8)
