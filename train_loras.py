#!/usr/bin/env python3
"""LoRA adversarial-defence training on MI355X -- command-line compatible with the reference's
train_loras.py (flags :427-442; adapter layout <out>/<model>/<source>/<attack>/rank_<r>/...).

One peft-style adapter per (attack, rank) is trained on adversarial images, frozen backbone,
trainable LoRA A/B + classifier, Adam(lr), CrossEntropyLoss -- the loop of train_loras.py:295-324
written against the same objects: `peft_model.base_model(pixel_values=x).logits`,
`criterion(logits, labels).backward()`, `optimizer.step()`.

Extensions (opt-in):
  --pgd-inner-steps K   generate the adversarial batch on the fly with PGD-K on the current model
                        (BASELINE config 3) instead of reading pre-generated PNGs.
  --synthetic N         seeded random images / weights, no files needed.
  data parallel         launch with torch.distributed.run: batches shard over ranks and the flat
                        LoRA+classifier gradient is summed with ONE all-reduce per step (RCCL).
"""
import argparse
import importlib
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
V = importlib.import_module("adapting-pretrained-vision-transformers-with-lora-against-attack-vectors_amd")


def build_parser():
    p = argparse.ArgumentParser(description="Train LoRAs for Adversarial Defense (MI355X / HIP)")
    p.add_argument("--models", nargs="+", required=True)
    p.add_argument("--sources", nargs="+", required=True)
    p.add_argument("--attacks", nargs="+", required=True)
    p.add_argument("--model_base_path", default="./Train24/{model}/{source}/{model}_best_model_finetuned.pth")
    p.add_argument("--adv_root", default=None)
    p.add_argument("--data_root", default=None)
    p.add_argument("--output_dir", default="./lora_defenses")
    p.add_argument("--batch_size", type=int, default=32)
    p.add_argument("--lr", type=float, default=1e-4)
    p.add_argument("--epochs", type=int, default=4)
    p.add_argument("--ranks", nargs="+", type=int, default=[8, 16, 32])
    p.add_argument("--lora_dropout", type=float, default=0.1)
    p.add_argument("--pgd-inner-steps", type=int, default=0)
    p.add_argument("--epsilon", type=float, default=8 / 255)
    p.add_argument("--pgd_alpha", type=float, default=2 / 255)
    p.add_argument("--synthetic", type=int, default=0, metavar="N")
    p.add_argument("--num_classes", type=int, default=21)
    p.add_argument("--seed", type=int, default=0)
    return p


def init_distributed():
    world = int(os.environ.get("WORLD_SIZE", 1))
    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    return rank, world, torch.device("cuda", local)


def train_one(args, base_model, train_batches, rank_r, out_dir, device, mean, std, world_rank, world):
    """One adapter: the loop of train_loras.py:281-354."""
    peft_model = V.setup_peft_lora(base_model, rank=rank_r, dropout=args.lora_dropout)
    if world > 1:
        import torch.distributed as dist
        dist.broadcast(peft_model._vit.trainable_flat().data, src=0)      # identical initial adapters
        peft_model._vit.mark_dirty()
    criterion = torch.nn.CrossEntropyLoss()
    optimizer = V.Adam(peft_model.parameters(), lr=args.lr, model=peft_model)
    engine = peft_model._vit._engine()
    hist = {"train_loss": [], "train_acc": []}
    for epoch in range(args.epochs):
        peft_model.train()
        tot_loss = torch.zeros((), device=device)
        tot_ok = torch.zeros((), device=device)
        n = 0
        for images, labels in train_batches():
            images, labels = images.to(device), labels.to(device)
            if args.pgd_inner_steps > 0:
                peft_model.eval()
                engine.set_normalization(mean, std)
                images = engine.pgd_attack(images, labels, args.epsilon, args.pgd_alpha, args.pgd_inner_steps,
                                           random_start=True, seed=args.seed + n)
                images = engine.channel_affine(images, [1.0 / s for s in std], [-m / s for m, s in zip(mean, std)])
                peft_model.train()
            optimizer.zero_grad()
            logits = peft_model.base_model(pixel_values=images).logits
            loss = criterion(logits, labels)
            loss.backward()
            optimizer.step()
            b = images.size(0)
            n += b
            tot_loss += loss.detach() * b                      # no per-step host sync
            tot_ok += (logits.detach().argmax(1) == labels).sum()
        hist["train_loss"].append(float(tot_loss / max(n, 1)))
        hist["train_acc"].append(float(tot_ok / max(n, 1)))
        if world_rank == 0:
            print(f"  rank {rank_r} epoch {epoch + 1}/{args.epochs}: loss {hist['train_loss'][-1]:.4f} acc {hist['train_acc'][-1]:.4f}")
    if world_rank == 0:
        peft_model.save_pretrained(os.path.join(out_dir, f"rank_{rank_r}", "final_lora"))
        with open(os.path.join(out_dir, f"rank_{rank_r}", "results.json"), "w") as f:
            json.dump(hist, f, indent=2)
    return hist


def main(argv=None):
    args = build_parser().parse_args(argv)
    rank, world, device = init_distributed()
    iomod = importlib.import_module(V.__name__ + ".io")
    syn = importlib.import_module(V.__name__ + ".synthetic")
    opt = importlib.import_module(V.__name__ + ".optim")
    results = {}
    for model_name in args.models:
        for source in args.sources:
            mean, std = V.get_normalization(model_name)
            if args.synthetic:
                base = V.create_vit_model(args.num_classes, device=device)
                base.load_state_dict(syn.random_state_dict(base.arch, seed=args.seed))
                class_to_idx = {f"class_{i}": i for i in range(args.num_classes)}
            else:
                path = args.model_base_path.format(model=model_name, source=source)
                mapping = os.path.join(os.path.dirname(path), "class_mappings.txt")
                if not (os.path.exists(path) and os.path.exists(mapping)):
                    print(f"Warning: missing {path} or {mapping}; skipping")
                    continue
                class_to_idx = iomod.read_class_mappings(mapping)
                base = V.create_vit_model(len(class_to_idx), device=device)
                base.load_state_dict(torch.load(path, map_location="cpu", weights_only=True))
            for attack in args.attacks:
                out_dir = os.path.join(args.output_dir, model_name, source, attack)

                def train_batches():
                    if args.synthetic:
                        x, y = syn.random_batch(base.arch, args.synthetic, seed=args.seed + 7)
                        lo, hi = opt.shard_batch(args.synthetic, rank, world)
                        x, y = x[lo:hi], y[lo:hi]
                        m = torch.tensor(mean).view(1, 3, 1, 1)
                        s = torch.tensor(std).view(1, 3, 1, 1)
                        for i in range(0, x.shape[0], args.batch_size):
                            xb = x[i:i + args.batch_size]
                            # pre-generated adversarial PNGs are loaded NORMALISED (train_loras.py:187-192);
                            # with --pgd-inner-steps the attack wants [0,1] images
                            yield (xb if args.pgd_inner_steps > 0 else (xb - m) / s), y[i:i + args.batch_size]
                        return
                    split_dir = os.path.join(args.adv_root, model_name, source, "train", attack)
                    ds = iomod.FolderDataset(split_dir, os.path.join(split_dir, "metadata.csv"), class_to_idx,
                                             normalise=None if args.pgd_inner_steps > 0 else (mean, std))
                    sub = torch.utils.data.Subset(ds, list(range(rank, len(ds), world)))
                    for xb, yb, _ in torch.utils.data.DataLoader(sub, batch_size=args.batch_size, shuffle=True,
                                                                 num_workers=min(4, os.cpu_count() or 1)):
                        yield xb, yb

                for r in args.ranks:
                    try:
                        os.makedirs(os.path.join(out_dir, f"rank_{r}"), exist_ok=True)
                        results[f"{model_name}/{source}/{attack}/rank_{r}"] = train_one(
                            args, base, train_batches, r, out_dir, device, mean, std, rank, world)
                    except Exception as e:            # skip-and-continue, like the reference (:392-395)
                        import traceback
                        traceback.print_exc()
                        print(f"Error training rank {r} for {attack}: {e}")
    if rank == 0:
        os.makedirs(args.output_dir, exist_ok=True)
        with open(os.path.join(args.output_dir, "all_results.json"), "w") as f:
            json.dump(results, f, indent=2)


if __name__ == "__main__":
    main()
