#!/usr/bin/env python3
"""Synthetic-pair generation: the reference's generate_synthetic_data.py (generate_data :33-89) on the HIP engine.

Loads generator_epoch_N.pth strictly into UNetGenerator(3, 3) (generate_synthetic_data.py:46-48), runs the eval-mode
forward and writes img1_synth_{idx}.png / img2_synth_{idx}.png / cm_synth_{idx}.png under images/<city>, labels/<city>
with idx = batch_index * batch_size + j (:74-81), so dataset.scan_dataset(is_synthetic=True) finds them.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import torch  # noqa: E402
from PIL import Image  # noqa: E402
from torch.utils.data import DataLoader  # noqa: E402

import pfa_gan  # noqa: E402
from pfa_gan.data import BaseChangeDetectionDataset, collate_fn_skip_none, scan_dataset, to_uint8_image  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images-dir", required=True)
    ap.add_argument("--labels-dir", required=True)
    ap.add_argument("--checkpoint", required=True, help="generator_epoch_200.pth (generate_synthetic_data.py:18)")
    ap.add_argument("--out-dir", required=True)
    ap.add_argument("--device", default="cuda" if torch.cuda.is_available() else "cpu")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--batch-size", type=int, default=4)     # :22
    ap.add_argument("--size", type=int, default=256)         # :23
    ap.add_argument("--num-downs", type=int, default=7)
    ap.add_argument("--ngf", type=int, default=64)
    ap.add_argument("--workers", type=int, default=2)        # :25
    return ap.parse_args()


@torch.no_grad()
def generate_data(args):
    device = torch.device(args.device)
    samples = scan_dataset(args.images_dir, args.labels_dir, is_synthetic=False)
    ds = BaseChangeDetectionDataset(samples, target_size=(args.size, args.size), augment=False)
    if len(ds) == 0:
        print("Error: Original training dataset is empty. Cannot generate synthetic data.")
        return 0
    loader = DataLoader(ds, batch_size=args.batch_size, shuffle=False, num_workers=args.workers,
                        collate_fn=collate_fn_skip_none)
    gen = pfa_gan.UNetGenerator(3, 3, num_downs=args.num_downs, ngf=args.ngf).to(device).set_precision(args.precision)
    gen.load_state_dict(torch.load(args.checkpoint, map_location=device))      # strict, like the reference
    gen.eval()
    count = 0

    def batches():
        for i, batch in enumerate(loader):
            if batch is not None:
                yield i, batch

    def results():
        """(loader index, batch dict, fake uint8 [B,H,W,3]): on a GPU generator the fixed-shape batches go through the
        captured-graph pipeline (pfa_gan.TilePipeline: pinned double buffering, no blocking .cpu() per batch) and the ragged
        last one through the eager path; elsewhere the plain forward."""
        if device.type == "cuda" and gen.fused():
            tiles = pfa_gan.TilePipeline(gen, args.batch_size, args.size)
            held = []

            def feed():
                for i, batch in batches():
                    held.append((i, batch))
                    yield batch["image1"]
            for _, u8 in tiles(feed()):
                i, batch = held.pop(0)
                yield i, batch, u8
        else:
            for i, batch in batches():
                yield i, batch, gen.generate_uint8(batch["image1"].to(device)).cpu().numpy()

    for i, batch, fake_u8 in results():
        a = batch["image1"]
        for j in range(a.size(0)):
            idx = i * args.batch_size + j
            city = batch["city"][j]
            img_dir = os.path.join(args.out_dir, "images", city)
            lbl_dir = os.path.join(args.out_dir, "labels", city)
            os.makedirs(img_dir, exist_ok=True)
            os.makedirs(lbl_dir, exist_ok=True)
            Image.fromarray(to_uint8_image(a[j] * 0.5 + 0.5)).save(os.path.join(img_dir, f"img1_synth_{idx}.png"))
            Image.fromarray(fake_u8[j]).save(os.path.join(img_dir, f"img2_synth_{idx}.png"))
            Image.fromarray((batch["label"][j].byte() * 255).numpy()).save(os.path.join(lbl_dir, f"cm_synth_{idx}.png"))
            count += 1
    print(f"\nSynthetic data generation finished. Saved {count} samples to {args.out_dir}")
    return count


if __name__ == "__main__":
    generate_data(parse())
