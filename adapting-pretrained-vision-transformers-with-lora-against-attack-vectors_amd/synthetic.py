"""Seeded synthetic weights / inputs (no pretrained checkpoint or dataset is reachable
offline).  Host-side plumbing for bench.py, smoke() and the CLIs' --synthetic
mode -- values only, no arithmetic of the hot path."""
from __future__ import annotations

import math
from typing import Dict, Tuple

import torch

from .engine import LINEAR_MODULES, ArchConfig, module_shape


def random_state_dict(arch: ArchConfig, seed: int = 0, std: float = 0.02) -> Dict[str, torch.Tensor]:
    """HF-4.55.2-keyed state dict, N(0, std) like HF's initializer_range, LayerNorm gains near 1."""
    g = torch.Generator().manual_seed(seed)

    def rn(*shape, s=std):
        return torch.randn(*shape, generator=g) * s

    D, P = arch.hidden, arch.patch_size
    sd = {
        "vit.embeddings.cls_token": rn(1, 1, D),
        "vit.embeddings.position_embeddings": rn(1, arch.tokens, D),
        "vit.embeddings.patch_embeddings.projection.weight": rn(D, 3, P, P),
        "vit.embeddings.patch_embeddings.projection.bias": rn(D),
        "vit.layernorm.weight": 1.0 + rn(D, s=0.1),
        "vit.layernorm.bias": rn(D),
        "classifier.weight": rn(arch.num_labels, D),
        "classifier.bias": rn(arch.num_labels),
    }
    for i in range(arch.layers):
        p = f"vit.encoder.layer.{i}."
        for short, path in LINEAR_MODULES:
            o, k = module_shape(arch, short)
            sd[p + path + ".weight"] = rn(o, k)
            sd[p + path + ".bias"] = rn(o)
        for ln in ("layernorm_before", "layernorm_after"):
            sd[p + ln + ".weight"] = 1.0 + rn(D, s=0.1)
            sd[p + ln + ".bias"] = rn(D)
    return sd


def random_lora(arch: ArchConfig, r: int, targets: Tuple[str, ...], seed: int = 1, b_std: float = 0.02):
    """{(layer, target): (A [r,in], B [out,r])}: A kaiming-uniform(a=sqrt 5) as peft; B ~ N(0, b_std)
    (peft's B = 0 would make the LoRA branch vacuous in a benchmark)."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for i in range(arch.layers):
        for t in targets:
            o, k = module_shape(arch, t)
            bound = 1.0 / math.sqrt(k)
            A = (torch.rand(r, k, generator=g) * 2 - 1) * bound
            B = torch.randn(o, r, generator=g) * b_std if b_std > 0 else torch.zeros(o, r)
            out[(i, t)] = (A, B)
    return out


def random_batch(arch: ArchConfig, batch: int, seed: int = 0):
    """x ~ U[0,1)^{B x 3 x S x S} (seed), labels ~ randint(0, C) (seed + 1)."""
    gx = torch.Generator().manual_seed(seed)
    gy = torch.Generator().manual_seed(seed + 1)
    x = torch.rand(batch, 3, arch.image_size, arch.image_size, generator=gx)
    y = torch.randint(0, arch.num_labels, (batch,), generator=gy)
    return x, y


ARCHS = {
    # name: (image_size, hidden, layers, heads, mlp)
    "vit_b": (224, 768, 12, 12, 3072),       # google/vit-base-patch16-224 (Utils.py:84-90)
    "vit_l": (224, 1024, 24, 16, 4096),      # ViT-L/16 (BASELINE config 5)
    "tiny": (64, 128, 2, 2, 256),            # plumbing / test runs of the CLIs
}


def arch_by_name(name: str, num_labels: int) -> ArchConfig:
    s, d, l, h, m = ARCHS[name]
    return ArchConfig(image_size=s, hidden=d, layers=l, heads=h, mlp=m, num_labels=int(num_labels))


def write_dataset_tree(root: str, classes, n_per_split=None, image_size: int = 64, seed: int = 0, source: str = "mapillary"):
    """A dataset tree in the reference's on-disk layout (Utils.py:12-82, whitebox_attacks.py:118-133):
    <root>/<split>/images/*.png + <root>/<split>/metadata.csv with columns image_path, unified_class, source.
    Pixels are seeded noise with a class-dependent mean so that a head can learn something."""
    import os

    import pandas as pd
    from PIL import Image
    n_per_split = n_per_split or {"train": 48, "val": 16, "test": 16}
    g = torch.Generator().manual_seed(seed)
    for split, n in n_per_split.items():
        d = os.path.join(root, split, "images")
        os.makedirs(d, exist_ok=True)
        rows = []
        for i in range(n):
            c = i % len(classes)
            img = (torch.rand(image_size, image_size, 3, generator=g) * 0.6 + 0.4 * (c / max(1, len(classes) - 1)))
            fn = f"{split}_{i:05d}.png"
            Image.fromarray((img.clamp(0, 1) * 255).to(torch.uint8).numpy()).save(os.path.join(d, fn))
            rows.append({"image_path": os.path.join(split, "images", fn), "unified_class": classes[c], "source": source})
        pd.DataFrame(rows).to_csv(os.path.join(root, split, "metadata.csv"), index=False)
