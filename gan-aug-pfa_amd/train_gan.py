#!/usr/bin/env python3
"""Pix2Pix GAN training driver: the reference's train_gan.py (main_gan :95-155, save_samples :78-92) with its
module-level constants as defaults and CLI overrides.  Checkpoints keep the reference's names and layout
(generator_epoch_{E}.pth / discriminator_epoch_{E}.pth = plain state_dict, train_gan.py:149-153).

  python train_gan.py --root /data --device cuda --precision bf16        # fused HIP step on one MI355X
  python -m torch.distributed.run --nproc-per-node 8 train_gan.py ...    # data parallel, one process per GPU
  python train_gan.py --root /data --device cpu --num-downs 6 --size 64  # CPU plumbing (BASELINE config 1)
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

DATASET_SUBDIR = "Onera Satellite Change Detection Dataset"     # train_gan.py:18-21


def parse():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--root", default=os.environ.get("PFA_ROOT", "."))
    ap.add_argument("--images-dir", default=None)
    ap.add_argument("--labels-dir", default=None)
    ap.add_argument("--checkpoint-dir", default=None)
    ap.add_argument("--output-dir", default=None)
    ap.add_argument("--device", default="cuda" if torch.cuda.is_available() else "cpu")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--batch-size", type=int, default=1)            # train_gan.py:26
    ap.add_argument("--epochs", type=int, default=200)              # :27
    ap.add_argument("--lr-g", type=float, default=1e-4)             # :28
    ap.add_argument("--lr-d", type=float, default=1e-4)             # :29
    ap.add_argument("--beta1", type=float, default=0.5)             # :30
    ap.add_argument("--size", type=int, default=256)                # :32
    ap.add_argument("--lambda-l1", type=float, default=100.0)       # :33
    ap.add_argument("--save-every", type=int, default=10)           # :34
    ap.add_argument("--sample-every", type=int, default=5)          # :35
    ap.add_argument("--num-downs", type=int, default=7)
    ap.add_argument("--ngf", type=int, default=64)
    ap.add_argument("--ndf", type=int, default=64)
    ap.add_argument("--workers", type=int, default=2)               # :135
    ap.add_argument("--resume", default=None, help="directory with generator_/discriminator_epoch_N.pth + trainer_epoch_N.pt")
    ap.add_argument("--seed", type=int, default=None)
    return ap.parse_args()


@torch.no_grad()
def save_samples(gen, loader, epoch, output_dir, device):
    """[A | G(A) | B] strip of the first sample, de-normalised (train_gan.py:78-92)."""
    gen.eval()
    data = next(iter(loader))
    a, b = data["image1"].to(device), data["image2"].to(device)
    fake = gen(a)
    grid = torch.cat([t[0] * 0.5 + 0.5 for t in (a, fake, b)], dim=-1)
    path = os.path.join(output_dir, f"sample_{data['city'][0]}_epoch_{epoch:03d}.png")
    Image.fromarray(to_uint8_image(grid)).save(path)
    print(f"Saved sample image to {path}")
    gen.train()


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    device = torch.device(args.device)
    pg = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if device.type == "cuda":
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
            device = torch.device("cuda", torch.cuda.current_device())
        torch.distributed.init_process_group("nccl" if device.type == "cuda" else "gloo")
        pg = torch.distributed.group.WORLD
    images = args.images_dir or os.path.join(args.root, DATASET_SUBDIR, "images", "Onera Satellite Change Detection dataset - Images")
    labels = args.labels_dir or os.path.join(args.root, DATASET_SUBDIR, "train_labels", "Onera Satellite Change Detection dataset - Train Labels")
    ckpt_dir = args.checkpoint_dir or os.path.join(args.root, "gan_checkpoints")
    out_dir = args.output_dir or os.path.join(args.root, "gan_samples")
    if rank == 0:
        os.makedirs(ckpt_dir, exist_ok=True)
        os.makedirs(out_dir, exist_ok=True)
    print(f"Using device: {device}")
    samples = scan_dataset(images, labels, is_synthetic=False)
    ds = BaseChangeDetectionDataset(samples, target_size=(args.size, args.size), augment=False)
    if len(ds) == 0:
        print("Error: GAN Training dataset is empty. Check dataset path and structure.")
        return 1
    sampler = torch.utils.data.distributed.DistributedSampler(ds, world, rank, shuffle=True, drop_last=True) if world > 1 else None
    loader = DataLoader(ds, batch_size=args.batch_size, shuffle=sampler is None, sampler=sampler, num_workers=args.workers,
                        drop_last=True, collate_fn=collate_fn_skip_none)
    sample_loader = DataLoader(ds, batch_size=1, shuffle=True, collate_fn=collate_fn_skip_none)
    if args.seed is not None:
        torch.manual_seed(args.seed)
    gen = pfa_gan.UNetGenerator(3, 3, num_downs=args.num_downs, ngf=args.ngf).to(device).set_precision(args.precision)
    disc = pfa_gan.NLayerDiscriminator(6, ndf=args.ndf).to(device).set_precision(args.precision)
    if device.type == "cuda":
        opt_g = pfa_gan.FusedAdam(gen.parameters(), lr=args.lr_g, betas=(args.beta1, 0.999))
        opt_d = pfa_gan.FusedAdam(disc.parameters(), lr=args.lr_d, betas=(args.beta1, 0.999))
    else:
        opt_g = torch.optim.Adam(gen.parameters(), lr=args.lr_g, betas=(args.beta1, 0.999))
        opt_d = torch.optim.Adam(disc.parameters(), lr=args.lr_d, betas=(args.beta1, 0.999))
    start = 1
    if args.resume:
        start = pfa_gan.load_training_state(args.resume, gen, disc, opt_g, opt_d, device) + 1
    if world > 1:
        pfa_gan.broadcast_model(gen, pg)
        pfa_gan.broadcast_model(disc, pg)
    print("Starting GAN training...")
    for epoch in range(start, args.epochs + 1):
        if sampler is not None:
            sampler.set_epoch(epoch)
        print(f"\nEpoch {epoch}/{args.epochs}")
        loss_d, loss_g = pfa_gan.train_gan_one_epoch(gen, disc, loader, opt_g, opt_d, device=device,
                                                     lambda_l1=args.lambda_l1, process_group=pg)
        print(f"Epoch {epoch} - Avg Loss D: {loss_d:.4f}, Avg Loss G: {loss_g:.4f}")
        if rank == 0 and (epoch % args.sample_every == 0 or epoch == args.epochs):
            save_samples(gen, sample_loader, epoch, out_dir, device)
        if rank == 0 and (epoch % args.save_every == 0 or epoch == args.epochs):
            torch.save(gen.state_dict(), os.path.join(ckpt_dir, f"generator_epoch_{epoch}.pth"))
            torch.save(disc.state_dict(), os.path.join(ckpt_dir, f"discriminator_epoch_{epoch}.pth"))
            # resume state is an EXTENSION kept in its own file: the reference-format .pth files stay untouched
            torch.save(dict(epoch=epoch, opt_g=opt_g.state_dict(), opt_d=opt_d.state_dict()),
                       os.path.join(ckpt_dir, f"trainer_epoch_{epoch}.pt"))
            print(f"GAN Checkpoints saved for epoch {epoch}")
    print("GAN Training finished.")
    if world > 1:
        torch.distributed.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
