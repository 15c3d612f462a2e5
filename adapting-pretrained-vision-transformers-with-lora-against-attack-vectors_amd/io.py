"""On-disk formats of the path (host IO, no accelerated arithmetic except the u8 quantiser):

    save_images(images, filenames, output_dir)     Utils.py:106-113  (clamp -> *255 -> uint8 TRUNCATION -> PNG)
    class_mappings.txt  "idx: name" per line       train.py:216-219, whitebox_attacks.py:88-90,135-139
    metadata.csv / create_adv_metadata             Utils.py:95-104,115-120
    <base>/<model>/<source>/<model>_best_model_finetuned.pth   train.py:249-254
"""
from __future__ import annotations

import os
from typing import Dict, List, Sequence

import torch


def save_images(images: torch.Tensor, filenames: Sequence[str], output_dir: str, engine=None):
    """Same bytes as the reference: uint8(clamp(x,0,1)*255) with truncation, HWC, PNG.
    With an engine the quantisation runs on the GPU (vl_quantize_u8) and only bytes cross PCIe."""
    from PIL import Image
    os.makedirs(output_dir, exist_ok=True)
    if engine is not None and images.is_cuda:
        u8 = engine.quantize_u8(images).cpu().numpy()
    else:
        x = torch.clamp(images.detach().float().cpu(), 0, 1).permute(0, 2, 3, 1)
        u8 = (x * 255).to(torch.uint8).numpy()
    for i, fn in enumerate(filenames):
        Image.fromarray(u8[i]).save(os.path.join(output_dir, fn))


def read_class_mappings(path: str) -> Dict[str, int]:
    out = {}
    with open(path) as f:
        for line in f:
            line = line.strip()
            if not line:
                continue
            idx, name = line.split(": ", 1)
            out[name] = int(idx)
    return out


def write_class_mappings(path: str, class_to_idx: Dict[str, int]):
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    with open(path, "w") as f:
        for name, idx in sorted(class_to_idx.items(), key=lambda kv: kv[1]):
            f.write(f"{idx}: {name}\n")


def model_paths(model_base_path: str, model_name: str, source_name: str):
    d = os.path.join(model_base_path, model_name, source_name)
    return os.path.join(d, f"{model_name}_best_model_finetuned.pth"), os.path.join(d, "class_mappings.txt")


def create_adv_metadata(clean_meta_path: str, filenames: List[str], adv_dir: str):
    import pandas as pd
    clean = pd.read_csv(clean_meta_path)
    keep = clean[clean["image_path"].apply(os.path.basename).isin(filenames)].copy()
    keep["image_path"] = keep["image_path"].apply(lambda p: os.path.join(adv_dir, os.path.basename(p)))
    return keep


class FolderDataset(torch.utils.data.Dataset):
    """metadata.csv-driven image folder in the reference's layout (Utils.py:12-82): columns
    image_path, unified_class (and source); images are resized to S and returned in [0,1]."""

    def __init__(self, root_dir: str, metadata_file: str, class_to_idx: Dict[str, int], image_size: int = 224,
                 sources=None, normalise=None):
        import pandas as pd
        df = pd.read_csv(metadata_file)
        if sources and "source" in df.columns:
            df = df[df["source"].isin(sources)]
        cls_col = "unified_class" if "unified_class" in df.columns else "class"
        df = df[df[cls_col].isin(class_to_idx.keys())]
        self.root, self.paths = root_dir, df["image_path"].tolist()
        self.labels = [class_to_idx[c] for c in df[cls_col].tolist()]
        self.filenames = [os.path.basename(p) for p in self.paths]
        self.size, self.normalise = image_size, normalise

    def __len__(self):
        return len(self.paths)

    def __getitem__(self, i):
        from PIL import Image
        import numpy as np
        p = self.paths[i]
        if not os.path.isabs(p) and not os.path.exists(p):
            p = os.path.join(self.root, p)
        img = Image.open(p).convert("RGB")
        # Resize(256) -> CenterCrop(224) of the reference transform (whitebox_attacks.py:129-133)
        s = int(round(self.size * 256 / 224))
        w, h = img.size
        sc = s / min(w, h)
        img = img.resize((max(s, round(w * sc)), max(s, round(h * sc))), Image.BILINEAR)
        w, h = img.size
        l, t = (w - self.size) // 2, (h - self.size) // 2
        img = img.crop((l, t, l + self.size, t + self.size))
        x = torch.from_numpy(np.asarray(img, dtype=np.float32) / 255.0).permute(2, 0, 1).contiguous()
        if self.normalise is not None:
            m, sd = self.normalise
            x = (x - torch.tensor(m).view(3, 1, 1)) / torch.tensor(sd).view(3, 1, 1)
        return x, self.labels[i], self.filenames[i]


def loader_workers(n_items: int) -> int:
    """DataLoader workers for a dataset of n_items images.  Workers are FORKED from a process that already holds a GPU
    context (a large address space to copy page tables of, once per loader and epoch): below a few thousand images the
    forks cost far more than decoding the PNGs in the main process does."""
    import os
    if n_items < 4096:
        return 0
    return min(4, os.cpu_count() or 1)
