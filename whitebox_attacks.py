#!/usr/bin/env python3
"""FGSM / PGD adversarial-example generation on MI355X -- command-line compatible with the
reference's whitebox_attacks.py (flags: whitebox_attacks.py:53-64; directory layout
<out>/<model>/<source>/<split>/<attack>/images/*.png + metadata.csv: :118-124,175-178).

Differences, all opt-in:
  --torchattacks-compat   reproduce the reference's PGD call pattern exactly
                          (set_normalization_used on un-normalised images, :169-170); the
                          default is the canonical attack in [0,1] with the model fed
                          (x - mean) / std, as the reference's own FGSM does (:22-38).
  --synthetic N           no dataset / checkpoint on disk: N seeded random images per split and
                          seeded random-init weights (throughput and plumbing runs).
  --lora_dir DIR          attack the model with a peft-format adapter applied.
  --world-size / RANK     image batches shard over ranks (one process per GPU, no collective).
"""
import argparse
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
V = importlib.import_module("adapting-pretrained-vision-transformers-with-lora-against-attack-vectors_amd")


def build_parser():
    p = argparse.ArgumentParser(description="Generate FGSM and PGD Attacks (MI355X / HIP)")
    p.add_argument("--data_root", required=False, default=None)
    p.add_argument("--models", nargs="+", required=True, help="Model architectures (e.g., google_vit)")
    p.add_argument("--sources", nargs="+", required=True, help="Source datasets (e.g., mapillary)")
    p.add_argument("--model_base_path", default="./Train24", help="Base path for models")
    p.add_argument("--output_dir", required=True)
    p.add_argument("--batch_size", type=int, default=32)
    p.add_argument("--epsilon", type=float, default=8 / 255)
    p.add_argument("--pgd_alpha", type=float, default=3 / 255)
    p.add_argument("--pgd_iters", type=int, default=30)
    p.add_argument("--splits", nargs="+", default=["train", "val", "test"])
    p.add_argument("--attacks", nargs="+", choices=["fgsm", "pgd"], default=["fgsm", "pgd"],
                   help="Which attacks to run (default: both)")
    p.add_argument("--torchattacks-compat", action="store_true")
    p.add_argument("--synthetic", type=int, default=0, metavar="N")
    p.add_argument("--num_classes", type=int, default=21, help="only with --synthetic")
    p.add_argument("--lora_dir", default=None)
    p.add_argument("--seed", type=int, default=0)
    return p


def load_model(args, model_name, source_name, device):
    mean, std = V.get_normalization(model_name)
    if args.synthetic:
        syn = importlib.import_module(V.__name__ + ".synthetic")
        model = V.create_vit_model(args.num_classes, device=device)
        model.load_state_dict(syn.random_state_dict(model.arch, seed=args.seed))
        class_to_idx = {f"class_{i}": i for i in range(args.num_classes)}
    else:
        iomod = importlib.import_module(V.__name__ + ".io")
        model_path, mapping_path = iomod.model_paths(args.model_base_path, model_name, source_name)
        if not os.path.exists(mapping_path):
            print(f"Warning: Class mapping file not found: {mapping_path}")
            return None
        class_to_idx = iomod.read_class_mappings(mapping_path)
        model = V.create_vit_model(len(class_to_idx), device=device)
        try:
            model.load_state_dict(torch.load(model_path, map_location="cpu", weights_only=True))
        except FileNotFoundError:
            print(f"Warning: Model file not found: {model_path}")
            return None
    if args.lora_dir:
        model = V.PeftModel.from_pretrained(model, args.lora_dir)
    model.eval()
    return model, class_to_idx, mean, std


def batches(args, split, class_to_idx, model, rank, world):
    """Yields (images in [0,1], labels, filenames) of this rank's shard."""
    if args.synthetic:
        syn = importlib.import_module(V.__name__ + ".synthetic")
        arch = importlib.import_module(V.__name__ + ".attacks")._unwrap(model).arch
        x, y = syn.random_batch(arch, args.synthetic, seed=args.seed + hash(split) % 1000)
        names = [f"{split}_{i:06d}.png" for i in range(args.synthetic)]
        idx = list(range(rank, args.synthetic, world))
        for s in range(0, len(idx), args.batch_size):
            sel = idx[s:s + args.batch_size]
            yield x[sel], y[sel], [names[i] for i in sel]
        return
    iomod = importlib.import_module(V.__name__ + ".io")
    meta = os.path.join(args.data_root, split, "metadata.csv")
    ds = iomod.FolderDataset(args.data_root, meta, class_to_idx, sources=args.sources)
    sub = torch.utils.data.Subset(ds, list(range(rank, len(ds), world)))
    loader = torch.utils.data.DataLoader(sub, batch_size=args.batch_size, shuffle=False,
                                         num_workers=min(4, os.cpu_count() or 1), pin_memory=True)
    for images, labels, filenames in loader:
        yield images, labels, list(filenames)


def main(argv=None):
    args = build_parser().parse_args(argv)
    if not args.synthetic and not args.data_root:
        raise SystemExit("--data_root is required unless --synthetic N is given")
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0)))
    print(f"Using device: {device} (rank {rank}/{world})")
    iomod = importlib.import_module(V.__name__ + ".io")
    atk = importlib.import_module(V.__name__ + ".attacks")

    for model_name in args.models:
        for source_name in args.sources:
            print(f"\nProcessing model: {model_name}, source: {source_name}")
            loaded = load_model(args, model_name, source_name, device)
            if loaded is None:
                continue
            model, class_to_idx, mean, std = loaded
            wrapped = V.LogitsModel(model)
            engine = atk._unwrap(model)._engine()
            pgd = None
            if "pgd" in args.attacks:
                pgd = V.PGD(wrapped, eps=args.epsilon, alpha=args.pgd_alpha, steps=args.pgd_iters, random_start=True,
                            seed=args.seed)
            for split in args.splits:
                print(f"  Processing {split} split...")
                base_out = os.path.join(args.output_dir, model_name, source_name, split)
                dirs = {a: os.path.join(base_out, a, "images") for a in args.attacks}
                for d in dirs.values():
                    os.makedirs(d, exist_ok=True)
                seen = []
                for images, labels, filenames in batches(args, split, class_to_idx, model, rank, world):
                    images, labels = images.to(device), labels.to(device)
                    seen.extend(filenames)
                    out = {}
                    if "fgsm" in args.attacks:
                        out["fgsm"] = V.batched_fgsm_attack(model, images, labels, args.epsilon, mean, std)
                    if pgd is not None:
                        if args.torchattacks_compat:
                            pgd.set_normalization_used(mean=mean, std=std)       # whitebox_attacks.py:169
                            out["pgd"] = pgd(images, labels)
                        else:
                            # canonical: attack in [0,1], model fed (x-mean)/std
                            engine.set_normalization(mean, std)
                            out["pgd"] = engine.pgd_attack(images, labels, args.epsilon, args.pgd_alpha, args.pgd_iters,
                                                           random_start=True, seed=args.seed + len(seen))
                    for a, adv in out.items():
                        iomod.save_images(adv, filenames, dirs[a], engine=engine)
                if not args.synthetic and rank == 0:
                    clean_meta = os.path.join(args.data_root, split, "metadata.csv")
                    for a in args.attacks:
                        meta = iomod.create_adv_metadata(clean_meta, seen if world == 1 else os.listdir(dirs[a]), dirs[a])
                        meta.to_csv(os.path.join(base_out, a, "metadata.csv"), index=False)
                        print(f"    {a.upper()} results saved to: {os.path.join(base_out, a)}")
                elif rank == 0:
                    print(f"    {len(seen)} images per attack written under {base_out}")


if __name__ == "__main__":
    main()
