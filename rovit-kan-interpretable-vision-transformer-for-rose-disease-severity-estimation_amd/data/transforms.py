"""``cutmix_or_mixup`` on the HIP path.

The reference trainer imports it from ``data.transforms`` (training/trainer.py:9) and calls it as
``images, labels_a, labels_b, lam = cutmix_or_mixup(images, class_labels, use_cutmix=, use_mixup=, cutmix_alpha=,
mixup_alpha=)`` (trainer.py:84-96), then mixes the two losses with ``lam`` (trainer.py:104-111).  The module itself is
not in the reference checkout, so the behaviour here is the published definition of the two augmentations (the same
one timm / the CutMix authors' code use) -- "parity unpinned" with respect to the reference, pinned against a torch
restatement in tests/test_gpu_augment.py:

* both enabled -> CutMix with probability 0.5, else MixUp; only one enabled -> that one
* lam ~ Beta(alpha, alpha); one random permutation of the batch pairs the samples
* CutMix pastes a box of area ratio (1 - lam) centred uniformly at random (clipped to the image) and returns
  lam = 1 - pasted_area / image_area

The host draws the three random numbers (numpy, like the published implementations); the batch is mixed by one HIP
kernel (rovit_mix_images) without leaving the device.
"""
from typing import Optional, Tuple

import numpy as np
import torch

from rovit_hip import native
from rovit_hip.native import call, ptr, stream_ptr


def rand_bbox(height: int, width: int, lam: float, rng=np.random) -> Tuple[int, int, int, int]:
    """Box of area ratio (1 - lam) around a uniformly drawn centre, clipped: returns (y0, y1, x0, x1)."""
    cut = float(np.sqrt(max(0.0, 1.0 - lam)))
    ch, cw = int(height * cut), int(width * cut)
    cy, cx = int(rng.randint(height)), int(rng.randint(width))
    y0, y1 = int(np.clip(cy - ch // 2, 0, height)), int(np.clip(cy + ch // 2, 0, height))
    x0, x1 = int(np.clip(cx - cw // 2, 0, width)), int(np.clip(cx + cw // 2, 0, width))
    return y0, y1, x0, x1


def mix_images(images: torch.Tensor, perm: torch.Tensor, mode: str, lam: float = 1.0,
               box: Tuple[int, int, int, int] = (0, 0, 0, 0)) -> torch.Tensor:
    """The data movement alone: mode 'mixup' -> lam*x + (1-lam)*x[perm]; mode 'cutmix' -> x with x[perm]'s box pasted."""
    if not images.is_cuda:
        raise native.RovitHipError('mix_images runs on the HIP device only (no CPU path)')
    if images.dim() != 4:
        raise native.RovitHipError(f'mix_images expects (B,C,H,W), got {tuple(images.shape)}')
    if mode not in ('mixup', 'cutmix'):
        raise ValueError(f'unknown mode {mode!r}')
    x = images.float().contiguous()
    perm = perm.to(device=x.device, dtype=torch.int64).contiguous()
    if perm.numel() != x.shape[0]:
        raise native.RovitHipError('mix_images: perm must hold one index per sample')
    out = torch.empty_like(x)
    B, C, H, W = x.shape
    y0, y1, x0, x1 = box
    call('rovit_mix_images', ptr(x), ptr(out), ptr(perm), B, C, H, W, 1 if mode == 'cutmix' else 0, float(lam),
         int(y0), int(y1), int(x0), int(x1), stream_ptr())
    return out


def cutmix_or_mixup(images: torch.Tensor, labels: torch.Tensor, use_cutmix: bool = True, use_mixup: bool = True,
                    cutmix_alpha: float = 1.0, mixup_alpha: float = 0.2, rng: Optional[np.random.RandomState] = None):
    """-> (mixed_images, labels_a, labels_b, lam) with the trainer's calling convention (trainer.py:86-92)."""
    rng = rng if rng is not None else np.random
    if not (use_cutmix or use_mixup):
        return images, labels, labels, 1.0
    do_cutmix = use_cutmix and (not use_mixup or rng.rand() < 0.5)
    alpha = cutmix_alpha if do_cutmix else mixup_alpha
    lam = float(rng.beta(alpha, alpha)) if alpha > 0 else 1.0
    B, _, H, W = images.shape
    perm = torch.randperm(B, device=images.device)
    if do_cutmix:
        box = rand_bbox(H, W, lam, rng)
        lam = 1.0 - (box[1] - box[0]) * (box[3] - box[2]) / float(H * W)
        mixed = mix_images(images, perm, 'cutmix', box=box)
    else:
        mixed = mix_images(images, perm, 'mixup', lam=lam)
    return mixed, labels, labels[perm], lam


# ------------------------------------------------------------------------------------------------------------
# Per-image transforms.  The reference's scripts import ``augmented_transforms`` / ``original_transforms`` /
# ``inference_transforms`` from this module (scripts/train.py:13, scripts/evaluate.py:11, scripts/run_ablation.py:12);
# the module is absent from the reference checkout and torchvision is not installed here, so these are plain tensor
# callables with the conventional behaviour (ImageNet normalisation; a random horizontal flip for training) --
# "parity unpinned".  They accept (3,H,W) or (B,3,H,W) float tensors on any device.
# ------------------------------------------------------------------------------------------------------------
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


class Normalize:
    def __init__(self, mean=IMAGENET_MEAN, std=IMAGENET_STD):
        self.mean, self.std = torch.tensor(mean).view(3, 1, 1), torch.tensor(std).view(3, 1, 1)

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        return (x - self.mean.to(x.device)) / self.std.to(x.device)


class RandomHorizontalFlip:
    def __init__(self, p: float = 0.5):
        self.p = p

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        if x.dim() == 3:
            return x.flip(-1) if float(torch.rand(())) < self.p else x
        flip = torch.rand(x.shape[0], device=x.device) < self.p
        return torch.where(flip.view(-1, 1, 1, 1), x.flip(-1), x)


class Compose:
    def __init__(self, fns):
        self.fns = list(fns)

    def __call__(self, x):
        for f in self.fns:
            x = f(x)
        return x


def augmented_transforms():
    return Compose([RandomHorizontalFlip(0.5), Normalize()])


def original_transforms():
    return Compose([Normalize()])


def inference_transforms():
    return Compose([Normalize()])
