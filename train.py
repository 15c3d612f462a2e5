#!/usr/bin/env python3
"""Base-model step of the pipeline -- flag-compatible shell of the reference's train.py
(:301-306: --data_root --output_dir --batch_size --epochs --lr --source).

Scope note (SURVEY.md section 2 / DESIGN.md): full-weight AdamW fine-tuning of all 86 M ViT
weights is NOT part of the accelerated hot path (the path freezes the backbone: attacks and
LoRA training).  What the rest of the pipeline needs from train.py is its two output files, in
its exact formats, and this script produces them:

    <output_dir>/google_vit/<source>/google_vit_best_model_finetuned.pth   state_dict, HF-4.55.2 keys
    <output_dir>/google_vit/<source>/class_mappings.txt                    "idx: name" per line

  * --from_state_dict FILE   re-export an existing checkpoint (HF 4.x or 5.x key names) into that layout;
  * --synthetic              seeded random-init weights (offline runs: no hub access, no dataset);
  * --head_only              train ONLY the classifier head on the frozen backbone with the HIP
                             engine (a LoRA config with no adapters: SEQ_CLS keeps the head trainable).
"""
import argparse
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
V = importlib.import_module("adapting-pretrained-vision-transformers-with-lora-against-attack-vectors_amd")


def main(argv=None):
    p = argparse.ArgumentParser(description="Train Vision Model (checkpoint provider for the HIP path)")
    p.add_argument("--data_root", default=None)
    p.add_argument("--output_dir", default="./base_models")
    p.add_argument("--batch_size", type=int, default=32)
    p.add_argument("--epochs", type=int, default=1)
    p.add_argument("--lr", type=float, default=1e-4)
    p.add_argument("--source", default="mapillary")
    p.add_argument("--from_state_dict", default=None)
    p.add_argument("--synthetic", action="store_true")
    p.add_argument("--num_classes", type=int, default=21)
    p.add_argument("--head_only", action="store_true")
    p.add_argument("--arch", choices=["tiny", "vit_b", "vit_l"], default="vit_b")
    p.add_argument("--seed", type=int, default=0)
    args = p.parse_args(argv)

    iomod = importlib.import_module(V.__name__ + ".io")
    syn = importlib.import_module(V.__name__ + ".synthetic")
    model_name = "google_vit"                               # hard-coded in the reference too (train.py:114)
    out_pth, out_map = iomod.model_paths(args.output_dir, model_name, args.source)
    os.makedirs(os.path.dirname(out_pth), exist_ok=True)

    class_to_idx = None
    if args.data_root and os.path.exists(os.path.join(args.data_root, "train", "metadata.csv")):
        import pandas as pd
        df = pd.read_csv(os.path.join(args.data_root, "train", "metadata.csv"))
        if "source" in df.columns:
            df = df[df["source"] == args.source]
        col = "unified_class" if "unified_class" in df.columns else "class"
        class_to_idx = {c: i for i, c in enumerate(sorted(df[col].unique()))}      # train.py:158-163
    if class_to_idx is None:
        class_to_idx = {f"class_{i}": i for i in range(args.num_classes)}
    C = len(class_to_idx)

    if args.from_state_dict:
        sd = torch.load(args.from_state_dict, map_location="cpu", weights_only=True)
        sd = {V.canonical_key(k): v for k, v in sd.items()}
    elif args.synthetic:
        sd = syn.random_state_dict(syn.arch_by_name(args.arch, C), seed=args.seed)
    else:
        raise SystemExit("full-weight fine-tuning is outside the accelerated path: pass --from_state_dict FILE "
                         "(re-export a checkpoint) or --synthetic (seeded random init); add --head_only to train the head")

    if args.head_only:
        if not args.data_root:
            raise SystemExit("--head_only needs --data_root")
        model = V.create_vit_model(C, arch=syn.arch_by_name(args.arch, C))
        model.load_state_dict(sd, strict=False)
        pm = V.get_peft_model(model, V.LoraConfig(task_type=V.TaskType.SEQ_CLS, r=1, target_modules=[]))
        opt = V.Adam(pm.parameters(), lr=args.lr, model=pm)
        ds = iomod.FolderDataset(args.data_root, os.path.join(args.data_root, "train", "metadata.csv"), class_to_idx,
                                 image_size=syn.arch_by_name(args.arch, C).image_size, sources=[args.source],
                                 normalise=V.get_normalization(model_name))
        crit = torch.nn.CrossEntropyLoss()
        for ep in range(args.epochs):
            pm.train()
            for x, y, _ in torch.utils.data.DataLoader(ds, batch_size=args.batch_size, shuffle=True):
                opt.zero_grad()
                loss = crit(pm.base_model(pixel_values=x.cuda()).logits, y.cuda())
                loss.backward()
                opt.step()
            print(f"epoch {ep + 1}/{args.epochs} loss {loss.item():.4f}")
        sd = pm._vit.state_dict()

    torch.save({k: v.clone() for k, v in sd.items()}, out_pth)
    iomod.write_class_mappings(out_map, class_to_idx)
    print(f"wrote {out_pth}\nwrote {out_map}")


if __name__ == "__main__":
    main()
