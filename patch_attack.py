#!/usr/bin/env python3
"""Adversarial-patch generation on MI355X -- command-line compatible with the reference's patch_attack.py (flags :80-108;
output <out>/<model>/<source>/<split>/patch_<type>/images/*.png + metadata.csv :150-153, 213-221).

For every patch type and split: optimise ONE patch on a random sample of the split (ART's AdversarialPatchPyTorch
semantics: random scale / rotation / location per image and step = expectation over transformations, Adam lr 5.0 on the
patch, clip to [0, 1]), then paste it on every image of the split at a random scale in [scale_min_apply, scale_max_apply]
(one scale per batch, as the reference does) and write the PNGs.

The EoT loop runs on the HIP engine (patch.py: vl_patch_apply -> vl_forward -> vl_loss_ce -> vl_backward_input ->
vl_patch_grad -> vl_adam_step -> vl_clamp).  Extensions: --synthetic N, --arch tiny|vit_b|vit_l, --lora_dir DIR,
--precision; under torch.distributed.run every rank optimises on its shard of each global batch, the [3, ps, ps] patch
gradient is all-reduced weighted by shard size (RCCL), and the split's batches are patched round-robin by the ranks.
"""
import argparse
import importlib
import os
import random
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
V = importlib.import_module("adapting-pretrained-vision-transformers-with-lora-against-attack-vectors_amd")
iomod = importlib.import_module(V.__name__ + ".io")
syn = importlib.import_module(V.__name__ + ".synthetic")
patch_mod = importlib.import_module(V.__name__ + ".patch")


def build_parser():
    p = argparse.ArgumentParser(description="Generate Adversarial Patch Attacks (MI355X / HIP)")
    p.add_argument("--data_root", default=None)
    p.add_argument("--model", required=True)
    p.add_argument("--source", required=True)
    p.add_argument("--model_path", default=None)
    p.add_argument("--output_dir", required=True)
    p.add_argument("--batch_size", type=int, default=16)
    p.add_argument("--patch_size", type=int, default=24)
    p.add_argument("--patch_sample_size", type=int, default=500)
    p.add_argument("--splits", nargs="+", default=["train", "val", "test"])
    p.add_argument("--scale_min", type=float, default=0.05)
    p.add_argument("--scale_max", type=float, default=1.0)
    p.add_argument("--rotation_max", type=float, default=22.5)
    p.add_argument("--distortion_scale_max", type=float, default=0.0)
    p.add_argument("--learning_rate", type=float, default=5.0)
    p.add_argument("--max_iter", type=int, default=500)
    p.add_argument("--patch_type", nargs="+", default=["circle", "square"], choices=["circle", "square"])
    p.add_argument("--optimizer", type=str, default="Adam", choices=["Adam", "pgd"])
    p.add_argument("--targeted", action="store_true", default=False)
    p.add_argument("--verbose", action="store_true", default=True)
    p.add_argument("--scale_min_apply", type=float, default=0.1)
    p.add_argument("--scale_max_apply", type=float, default=0.5)
    p.add_argument("--patch_location_x", type=int, default=None)
    p.add_argument("--patch_location_y", type=int, default=None)
    p.add_argument("--synthetic", type=int, default=0, metavar="N")
    p.add_argument("--num_classes", type=int, default=21, help="only with --synthetic")
    p.add_argument("--arch", choices=sorted(syn.ARCHS), default="vit_b")
    p.add_argument("--precision", choices=["f16", "bf16", "f32"], default="f16",
                   help="f16: fp16 operands (fast); a gradient that leaves the fp16 range is never silent -- that optimiser step is "
                        "dropped on every rank and counted in the summary line (bf16 / f32 have no such event)")
    p.add_argument("--lora_dir", default=None)
    p.add_argument("--seed", type=int, default=0)
    return p


def main(argv=None):
    args = build_parser().parse_args(argv)
    location = (args.patch_location_x, args.patch_location_y) if args.patch_location_x is not None and args.patch_location_y is not None else None
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    # one process per GPU; VITLORA_SHARE_GPU=1 / VITLORA_DIST_BACKEND=gloo: rehearsal on a one-GPU box
    local = 0 if os.environ.get("VITLORA_SHARE_GPU") == "1" else int(os.environ.get("LOCAL_RANK", 0))
    device = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        if not dist.is_initialized():
            backend = os.environ.get("VITLORA_DIST_BACKEND", "nccl")
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=device)
            else:
                dist.init_process_group(backend)
    random.seed(args.seed)
    mean, std = V.get_normalization(args.model)
    if args.synthetic:
        arch = syn.arch_by_name(args.arch, args.num_classes)
        class_to_idx = {f"class_{i}": i for i in range(args.num_classes)}
        sd = syn.random_state_dict(arch, seed=args.seed)
    else:
        if not (args.data_root and args.model_path):
            raise SystemExit("--data_root and --model_path are required unless --synthetic N is given")
        mapping = os.path.join(os.path.dirname(os.path.abspath(args.model_path)), "class_mappings.txt")
        if not os.path.exists(mapping):
            raise FileNotFoundError(f"Class mapping file not found: {mapping}")
        class_to_idx = iomod.read_class_mappings(mapping)
        arch = syn.arch_by_name(args.arch, len(class_to_idx))
        sd = torch.load(args.model_path, map_location="cpu", weights_only=True)
    base_model = V.create_vit_model(arch.num_labels, arch=arch, device=device, precision=args.precision)
    base_model.load_state_dict(sd)
    if args.lora_dir:
        base_model = V.PeftModel.from_pretrained(base_model, args.lora_dir)
    base_model.eval()
    wrapped = V.LogitsModel(base_model)          # the engine normalises inside the patch gather (NormalizedModel, :16-25)
    engine = importlib.import_module(V.__name__ + ".attacks")._unwrap(base_model)._engine()

    for patch_type in args.patch_type:
        print(f"\n{'=' * 50}\nGenerating patches for shape: {patch_type}\n{'=' * 50}")
        for split in args.splits:
            print(f"\nProcessing {split} split for {patch_type} patches...")
            base_out = os.path.join(args.output_dir, args.model, args.source, split, f"patch_{patch_type}")
            out_dir = os.path.join(base_out, "images")
            os.makedirs(out_dir, exist_ok=True)
            if args.synthetic:
                x_all, y_all = syn.random_batch(arch, args.synthetic, seed=args.seed + sum(map(ord, split)))
                names = [f"{split}_{i:06d}.png" for i in range(args.synthetic)]
                fetch = lambda idx: (x_all[idx], y_all[idx], [names[i] for i in idx])
                name_of = lambda i: names[i]
                n = args.synthetic
            else:
                meta = os.path.join(args.data_root, split, "metadata.csv")
                ds = iomod.FolderDataset(args.data_root, meta, class_to_idx, image_size=arch.image_size, sources=[args.source])
                n = len(ds)
                name_of = lambda i: ds.filenames[i]

                def fetch(idx):
                    items = [ds[i] for i in idx]
                    return torch.stack([it[0] for it in items]), torch.tensor([it[1] for it in items]), [it[2] for it in items]
            indices = list(range(n))
            random.shuffle(indices)                                         # patch_attack.py:177-180
            x_train, y_train, _ = fetch(indices[:args.patch_sample_size])
            attack = patch_mod.AdversarialPatchPyTorch(
                wrapped, rotation_max=args.rotation_max, scale_min=args.scale_min, scale_max=args.scale_max,
                distortion_scale_max=args.distortion_scale_max, learning_rate=args.learning_rate, max_iter=args.max_iter,
                batch_size=args.batch_size, patch_shape=(3, args.patch_size, args.patch_size), patch_location=location,
                patch_type=patch_type, optimizer=args.optimizer, targeted=args.targeted, verbose=args.verbose, seed=args.seed,
                mean=mean, std=std)
            patch, _ = attack.generate(x=x_train, y=y_train)               # every rank ends with the same patch (all-reduced steps)
            if rank == 0:
                print(f"  patch optimisation: {attack.steps_taken} steps taken, {attack.skipped_steps} dropped (fp16 range events)")
            if attack.steps_taken == 0 and attack.skipped_steps > 0:
                raise SystemExit("patch optimisation: EVERY step left the fp16 range; rerun with --precision bf16 or f32")
            if rank == 0:
                np.save(os.path.join(base_out, "patch.npy"), patch)
            all_filenames = []
            for bi, s0 in enumerate(range(0, n, args.batch_size)):
                idx = list(range(s0, min(n, s0 + args.batch_size)))
                scale = random.uniform(args.scale_min_apply, args.scale_max_apply)          # one scale per batch (:201); drawn by every rank
                if bi % world != rank:                                                      # batches are dealt round-robin to the ranks
                    all_filenames.extend(name_of(i) for i in idx)
                    continue
                images, _, filenames = fetch(idx)
                patched = attack.apply_patch(images.to(device), scale=scale)
                all_filenames.extend(filenames)
                iomod.save_images(patched, filenames, out_dir, engine=engine)
            if dist is not None:
                dist.barrier()                                                              # every PNG is on disk before the metadata
            if not args.synthetic and rank == 0:
                meta_out = iomod.create_adv_metadata(os.path.join(args.data_root, split, "metadata.csv"), all_filenames, out_dir)
                meta_out["image_path"] = meta_out["image_path"].apply(lambda p: os.path.abspath(p) if not os.path.isabs(p) else p)
                meta_out.to_csv(os.path.join(base_out, "metadata.csv"), index=False)
            if rank == 0:
                print(f"{patch_type.capitalize()} patch attack results saved to: {base_out}")


if __name__ == "__main__":
    main()
