"""Shared test plumbing: build the HIP engine from oracle weights, error metrics."""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PKG = "adapting-pretrained-vision-transformers-with-lora-against-attack-vectors_amd"

from oracle import vit_lora_oracle as O  # noqa: E402


def pkg():
    return importlib.import_module(PKG)


def rel_l2(a: torch.Tensor, b: torch.Tensor) -> float:
    a = a.detach().double().cpu().flatten()
    b = b.detach().double().cpu().flatten()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def make_case(image_size=64, hidden=128, layers=2, heads=2, mlp=256, num_labels=10, batch=4, seed=5,
              std=0.05, r=8, targets=("q", "k", "v", "o", "fc2"), b_std=0.05):
    cfg = O.OracleConfig(image_size=image_size, hidden=hidden, layers=layers, heads=heads, mlp=mlp,
                         num_labels=num_labels)
    w = O.init_weights(cfg, seed=seed, std=std)
    lora = O.init_lora(cfg, r=r, targets=targets, seed=seed + 1, b_std=b_std) if r else None
    g = torch.Generator().manual_seed(seed + 2)
    x = torch.rand(batch, 3, image_size, image_size, generator=g)
    y = torch.randint(0, num_labels, (batch,), generator=g)
    return cfg, w, lora, x, y


def make_engine(cfg, w, lora=None, merged=False, dropout=0.0, precision="f16"):
    P = pkg()
    arch = P.ArchConfig(image_size=cfg.image_size, patch_size=cfg.patch_size, hidden=cfg.hidden, layers=cfg.layers,
                        heads=cfg.heads, mlp=cfg.mlp, num_labels=cfg.num_labels, ln_eps=cfg.ln_eps)
    spec = None
    if lora is not None:
        spec = P.LoraSpec(r=lora.r, alpha=lora.alpha, dropout=dropout, targets=tuple(lora.targets), merged=merged)
    eng = P.Engine(arch, spec, precision=precision)
    eng.load_state_dict(w)
    if lora is not None:
        for (i, t), (A, B) in lora.ab.items():
            eng.param(i, t, "A").copy_(A)
            eng.param(i, t, "B").copy_(B)
        eng.commit()
    torch.cuda.synchronize()
    return eng


def fmnist_labels(n=512):
    """Real FashionMNIST test labels (idx1-ubyte: 8-byte header, then one byte per label) from the committed
    copy of the head of the reference's fashion_data/FashionMNIST/raw/t10k-labels-idx1-ubyte."""
    raw = open(os.path.join(ROOT, "tests", "golden", "fmnist_t10k_labels_512.bin"), "rb").read()
    assert raw[:4] == b"\x00\x00\x08\x01"
    return torch.tensor(list(raw[8:8 + n]), dtype=torch.int64)


def fmnist_like_images(n, seed=0, size=224):
    """BASELINE config 1 pixels: the FashionMNIST image files are not in the reference snapshot, so
    seeded 28x28 uint8 noise -> bilinear resize to `size` -> grayscale replicated to 3 channels -> [0,1]
    (the Resize / Grayscale(3) / ToTensor chain of train_bilora.ipynb:28-33)."""
    g = torch.Generator().manual_seed(seed)
    small = torch.randint(0, 256, (n, 1, 28, 28), generator=g).float() / 255.0
    big = torch.nn.functional.interpolate(small, size=(size, size), mode="bilinear", align_corners=False)
    return big.expand(n, 3, size, size).contiguous()
