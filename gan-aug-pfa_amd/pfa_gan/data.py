"""Batch contract of the reference's loader, without torchvision (dataset.py:162-283).

A batch is {"image1", "image2": float32 [B,3,H,W] in [-1,1] (dataset.py:155-159), "label": int64 [B,H,W] in {0,1}
(dataset.py:27-34), "city": list[str]}.  Directory layout: <images>/<city>/pair/img{1,2}.png and
<labels>/<city>/cm/cm.png for real data, <images>/<city>/img{1,2}_synth_N.png + <labels>/<city>/cm_synth_N.png for
generated data (dataset.py:248-270).  Augmentations (training of the downstream Siamese model only) are not part
of the GAN path - train_gan.py:108 builds its dataset with augment=False - and are not rebuilt here.
"""
import glob
import os

import numpy as np
import torch
import torch.nn.functional as F
from PIL import Image
from torch.utils.data import Dataset


def _readable(path):
    if path is None:
        return True
    try:
        with Image.open(path) as im:
            im.load()
        return True
    except Exception:
        return False


def scan_dataset(data_dir, label_dir=None, is_synthetic=False):
    """List of {"img1","img2","label","city"} dicts, one per usable sample; unreadable or incomplete ones are skipped."""
    out, skipped = [], 0
    for city_dir in sorted(glob.glob(os.path.join(data_dir, "*"))):
        if not os.path.isdir(city_dir):
            continue
        city = os.path.basename(city_dir)
        if is_synthetic:
            cands = []
            for f1 in sorted(glob.glob(os.path.join(city_dir, "img1_synth_*.png"))):
                tail = os.path.basename(f1)[len("img1_"):]
                cands.append((f1, os.path.join(city_dir, "img2_" + tail),
                              os.path.join(label_dir, city, "cm_" + tail) if label_dir else None, city + "_synth"))
        else:
            cands = [(os.path.join(city_dir, "pair", "img1.png"), os.path.join(city_dir, "pair", "img2.png"),
                      os.path.join(label_dir, city, "cm", "cm.png") if label_dir else None, city)]
        for f1, f2, lab, name in cands:
            ok = os.path.exists(f1) and os.path.exists(f2) and (lab is None or os.path.exists(lab))
            if ok and _readable(f1) and _readable(f2) and _readable(lab):
                out.append(dict(img1=f1, img2=f2, label=lab, city=name))
            else:
                skipped += 1
    print(f"Scanned {data_dir}. Found {len(out)} valid samples. Skipped {skipped}.")
    return out


def collate_fn_skip_none(batch):
    batch = [b for b in batch if b is not None]
    if not batch:
        return None
    return torch.utils.data.dataloader.default_collate(batch)


class BaseChangeDetectionDataset(Dataset):
    """PNG pair (+ change mask) -> tensors resized to target_size and normalised to [-1, 1]."""

    def __init__(self, samples_list, target_size=(128, 128), augment=False):
        if augment:
            raise NotImplementedError("augmentations belong to the Siamese pipeline, not the GAN path")
        self.samples, self.target_size = samples_list, tuple(target_size)

    def __len__(self):
        return len(self.samples)

    def _rgb(self, path):
        with Image.open(path) as im:
            a = np.asarray(im.convert("RGB"), dtype=np.float32) / 255.0
        t = torch.from_numpy(a).permute(2, 0, 1)
        if tuple(t.shape[1:]) != self.target_size:
            t = F.interpolate(t[None], size=self.target_size, mode="bilinear", align_corners=False, antialias=True)[0]
        return t * 2.0 - 1.0

    def __getitem__(self, idx):
        s = self.samples[idx]
        try:
            item = {"image1": self._rgb(s["img1"]), "image2": self._rgb(s["img2"]), "city": s["city"]}
            if s["label"] is not None:
                with Image.open(s["label"]) as im:
                    lab = torch.from_numpy((np.asarray(im.convert("L")) > 128).astype(np.int64))
                if tuple(lab.shape) != self.target_size:
                    lab = F.interpolate(lab[None, None].float(), size=self.target_size, mode="nearest")[0, 0].long()
                item["label"] = lab
            else:
                item["label"] = torch.zeros(self.target_size, dtype=torch.int64)
            return item
        except Exception as e:      # same policy as dataset.py:235-237: a bad sample is dropped by the collate
            print(f"Failed to load sample {s.get('city')} at index {idx}: {e}. Returning None.")
            return None


def to_uint8_image(t):
    """De-normalised CHW float tensor in [0,1] -> HWC uint8, the way TF.to_pil_image does it: mul(255).byte()."""
    return t.detach().float().cpu().mul(255).byte().permute(1, 2, 0).numpy()
