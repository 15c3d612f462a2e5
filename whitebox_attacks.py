#!/usr/bin/env python3
"""FGSM / PGD adversarial-example generation on MI355X -- command-line compatible with the
reference's whitebox_attacks.py (flags: whitebox_attacks.py:53-64; directory layout
<out>/<model>/<source>/<split>/<attack>/images/*.png + metadata.csv: :118-124,175-178).

PGD runs with the reference's own call pattern by default: torchattacks' set_normalization_used(mean, std)
on UN-normalised [0,1] images (:169-170), i.e. the attack happens in x*std+mean space and the model sees
x + delta/std.  Extensions, all opt-in:
  --canonical-pgd         the textbook attack instead: perturb in [0,1], model fed (x - mean) / std, as the
                          reference's own FGSM does (:22-38).
  --synthetic N           no dataset / checkpoint on disk: N seeded random images per split and
                          seeded random-init weights (throughput and plumbing runs).
  --arch tiny|vit_b|vit_l architecture of the checkpoint (default vit_b = google/vit-base-patch16-224).
  --lora_dir DIR          attack the model with a peft-format adapter applied.
  torch.distributed.run   image batches shard over ranks (one process per GPU, no data-path collective);
                          ranks meet at a barrier before rank 0 writes metadata.csv.
"""
import argparse
import importlib
import os
import zlib
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
    p.add_argument("--canonical-pgd", action="store_true")
    p.add_argument("--torchattacks-compat", action="store_true", help="(default behaviour; kept for round-1 command lines)")
    p.add_argument("--arch", choices=["tiny", "vit_b", "vit_l"], default="vit_b")
    p.add_argument("--tiny", action="store_true", help="same as --arch tiny")
    p.add_argument("--precision", choices=["f16", "bf16", "f32"], default="f16")
    p.add_argument("--synthetic", type=int, default=0, metavar="N")
    p.add_argument("--num_classes", type=int, default=21, help="only with --synthetic")
    p.add_argument("--lora_dir", default=None)
    p.add_argument("--seed", type=int, default=0)
    return p


def load_model(args, model_name, source_name, device):
    mean, std = V.get_normalization(model_name)
    syn = importlib.import_module(V.__name__ + ".synthetic")
    arch_name = "tiny" if args.tiny else args.arch
    if args.synthetic:
        model = V.create_vit_model(args.num_classes, arch=syn.arch_by_name(arch_name, args.num_classes), device=device,
                                   precision=args.precision)
        model.load_state_dict(syn.random_state_dict(model.arch, seed=args.seed))
        class_to_idx = {f"class_{i}": i for i in range(args.num_classes)}
    else:
        iomod = importlib.import_module(V.__name__ + ".io")
        model_path, mapping_path = iomod.model_paths(args.model_base_path, model_name, source_name)
        if not os.path.exists(mapping_path):
            print(f"Warning: Class mapping file not found: {mapping_path}")
            return None
        class_to_idx = iomod.read_class_mappings(mapping_path)
        model = V.create_vit_model(len(class_to_idx), arch=syn.arch_by_name(arch_name, len(class_to_idx)), device=device,
                                   precision=args.precision)
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
        # zlib.crc32, not hash(): str hashes are salted per process, and every rank must draw the SAME synthetic split
        x, y = syn.random_batch(arch, args.synthetic, seed=args.seed + zlib.crc32(split.encode()) % 1000)
        names = [f"{split}_{i:06d}.png" for i in range(args.synthetic)]
        idx = list(range(rank, args.synthetic, world))
        for s in range(0, len(idx), args.batch_size):
            sel = idx[s:s + args.batch_size]
            yield x[sel], y[sel], [names[i] for i in sel]
        return
    iomod = importlib.import_module(V.__name__ + ".io")
    meta = os.path.join(args.data_root, split, "metadata.csv")
    arch = importlib.import_module(V.__name__ + ".attacks")._unwrap(model).arch
    ds = iomod.FolderDataset(args.data_root, meta, class_to_idx, image_size=arch.image_size, sources=args.sources)
    sub = torch.utils.data.Subset(ds, list(range(rank, len(ds), world)))
    loader = torch.utils.data.DataLoader(sub, batch_size=args.batch_size, shuffle=False,
                                         num_workers=iomod.loader_workers(len(sub)), pin_memory=True)
    for images, labels, filenames in loader:
        yield images, labels, list(filenames)


def main(argv=None):
    args = build_parser().parse_args(argv)
    if not args.synthetic and not args.data_root:
        raise SystemExit("--data_root is required unless --synthetic N is given")
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    # one process per GPU; VITLORA_SHARE_GPU=1 (rehearsal on a one-GPU box, tests/test_hip_pipeline.py) puts every rank on cuda:0
    device = torch.device("cuda", 0 if os.environ.get("VITLORA_SHARE_GPU") == "1" else int(os.environ.get("LOCAL_RANK", 0)))
    print(f"Using device: {device} (rank {rank}/{world})")
    dist = None
    if world > 1:
        # no data-path collective: the group exists for the barrier / file-name exchange before metadata.csv
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if not dist.is_initialized():
            dist.init_process_group("gloo")
    iomod = importlib.import_module(V.__name__ + ".io")
    atk = importlib.import_module(V.__name__ + ".attacks")

    for model_name in args.models:
        for source_name in args.sources:
            print(f"\nProcessing model: {model_name}, source: {source_name}")
            loaded = load_model(args, model_name, source_name, device)
            if loaded is None:
                continue
            model, class_to_idx, mean, std = loaded
            fallback = {}                    # the same model in fp32, built on first need (a batch whose fp16 backward left its range)
            tele = {"batches": 0, "redone": 0}   # fp16 telemetry: VL_ERR_NONFINITE events = batches redone in fp32 (at 1/10 the speed)

            def attack_batch(mdl, images, labels, batch_seed):
                """FGSM / PGD on one batch through `mdl` (whitebox_attacks.py:157-173); raises NonFiniteGradient if flagged."""
                eng = atk._unwrap(mdl)._engine()
                out = {}
                if "fgsm" in args.attacks:
                    out["fgsm"] = V.batched_fgsm_attack(mdl, images, labels, args.epsilon, mean, std)
                if "pgd" in args.attacks:
                    if args.canonical_pgd:
                        # textbook form: attack in [0,1], model fed (x-mean)/std
                        eng.set_normalization(mean, std)
                        out["pgd"] = eng.pgd_attack(images, labels, args.epsilon, args.pgd_alpha, args.pgd_iters,
                                                    random_start=True, seed=batch_seed)
                    else:
                        pgd = V.PGD(V.LogitsModel(mdl), eps=args.epsilon, alpha=args.pgd_alpha, steps=args.pgd_iters,
                                    random_start=True, seed=batch_seed)
                        pgd.set_normalization_used(mean=mean, std=std)       # whitebox_attacks.py:169
                        out["pgd"] = pgd(images, labels)
                eng.check()                  # synchronises; fp16 mode: raises if a gradient left the fp16 range
                return out, eng

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
                    # random start of PGD: seeded by the batch's first file name, i.e. the same noise whichever rank draws it
                    batch_seed = args.seed + zlib.crc32(filenames[0].encode()) % (1 << 20)
                    tele["batches"] += 1
                    try:
                        out, engine = attack_batch(model, images, labels, batch_seed)
                    except V.NonFiniteGradient as e:
                        tele["redone"] += 1
                        print(f"    fp16 gradient out of range in batch starting at {filenames[0]} ({e}); redoing it in fp32")
                        if "model" not in fallback:
                            a32 = argparse.Namespace(**dict(vars(args), precision="f32"))
                            fallback["model"] = load_model(a32, model_name, source_name, device)[0]
                        out, engine = attack_batch(fallback["model"], images, labels, batch_seed)
                    for a, adv in out.items():
                        iomod.save_images(adv, filenames, dirs[a], engine=engine)
                all_seen = seen
                if dist is not None:
                    # every rank has finished writing its PNGs and rank 0 learns all file names: no directory
                    # listing, so stale files of an earlier run cannot leak into metadata.csv
                    gathered = [None] * world
                    dist.all_gather_object(gathered, seen)
                    # dataset order (rank r holds items r, r + world, ...), as the reference's all_filenames (:156-176)
                    all_seen = [part[i] for i in range(max(len(p) for p in gathered)) for part in gathered if i < len(part)]
                if not args.synthetic and rank == 0:
                    clean_meta = os.path.join(args.data_root, split, "metadata.csv")
                    for a in args.attacks:
                        meta = iomod.create_adv_metadata(clean_meta, all_seen, dirs[a])
                        meta.to_csv(os.path.join(base_out, a, "metadata.csv"), index=False)
                        print(f"    {a.upper()} results saved to: {os.path.join(base_out, a)}")
                elif rank == 0:
                    print(f"    {len(all_seen)} images per attack written under {base_out}")
            print(f"  [rank {rank}] fp16 range: {tele['redone']} of {tele['batches']} batches redone in fp32 (VL_ERR_NONFINITE events)")
    if dist is not None:
        dist.barrier()


if __name__ == "__main__":
    main()
