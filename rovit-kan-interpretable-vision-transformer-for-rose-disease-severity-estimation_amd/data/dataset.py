"""``data.dataset`` for the HIP path: ``RoseLeafDataset`` and ``create_dataloaders`` with the calling conventions of
the reference's consumers.

The module itself is NOT in the reference checkout (``data/`` is missing); what is known is how
it is called:

* ``scripts/train.py:73-84``  ``train_loader, val_loader, test_loader = create_dataloaders(augmented_root=, original_root=,
  class_names=, severity_map=, augmented_transform=, original_transform=, batch_size=, train_val_split=, num_workers=, seed=)``
* ``scripts/train.py:110-111``  ``train_loader.dataset.dataset.get_class_weights()`` (train loader wraps a ``Subset``)
* ``scripts/evaluate.py:40-46``  ``RoseLeafDataset(root_dir=, class_names=, severity_map=, transform=, mode='original')``
* ``training/trainer.py:79-82``  batches are ``(images, class_labels, severity_labels)``, moved with ``.to(device)``

so the behaviour behind those calls is this repo's own ("parity unpinned"): an image-folder dataset
``root/<class name>/*.{jpg,png,...}`` with severity = ``severity_map[class name]``, an 80/20 seeded split of the
augmented set into train/val, the original set as test, and inverse-frequency class weights.

There is no dataset on the GPU box (no network), so a SYNTHETIC mode stands in for the files: ``synthetic=N`` (or
``$ROVIT_SYNTHETIC_DATA=N`` when the roots do not exist) gives N seeded random images per split, generated once and
kept RESIDENT ON THE DEVICE; the loaders then hand out device tensors (``.to(device)`` in the trainer is a no-op) and
the training step never touches PCIe.
"""
from __future__ import annotations

import os
from pathlib import Path
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch
from torch.utils.data import DataLoader, Dataset, Subset

IMG_EXT = ('.jpg', '.jpeg', '.png', '.bmp', '.webp')
IMAGE_SIZE = 224


class RoseLeafDataset(Dataset):
    """``root_dir/<class name>/<image>`` -> ``(image (3,224,224) float, class index, severity)``.

    ``synthetic=N`` skips the file system: N seeded ``randn`` images with labels drawn uniformly over the classes
    (severity = ``severity_map[class]``, the identity on the reference's class order, configs/config.py:19-24),
    optionally resident on ``device``."""

    def __init__(self, root_dir=None, class_names: Sequence[str] = (), severity_map: Optional[Dict[str, int]] = None,
                 transform: Optional[Callable] = None, mode: str = 'augmented', synthetic: Optional[int] = None,
                 seed: int = 0, device: Optional[torch.device] = None, image_size: int = IMAGE_SIZE):
        self.root_dir = Path(root_dir) if root_dir is not None else None
        self.class_names = list(class_names)
        self.severity_map = dict(severity_map) if severity_map is not None else {c: i for i, c in enumerate(self.class_names)}
        self.transform, self.mode, self.image_size = transform, mode, image_size
        if not self.class_names:
            raise ValueError('RoseLeafDataset needs class_names')
        if synthetic is None and (self.root_dir is None or not self.root_dir.exists()):
            env = os.environ.get('ROVIT_SYNTHETIC_DATA')
            if env:
                synthetic = int(env)
            else:
                raise FileNotFoundError(f'{self.root_dir} does not exist; pass synthetic=N (or set ROVIT_SYNTHETIC_DATA=N) '
                                        'for seeded synthetic images')
        self.synthetic = synthetic
        self.samples: List[Tuple[Path, int]] = []
        self.images: Optional[torch.Tensor] = None
        if synthetic is not None:
            g = torch.Generator().manual_seed(seed)
            self.labels = torch.randint(0, len(self.class_names), (synthetic,), generator=g)
            dev = device if device is not None else torch.device('cpu')
            if dev.type == 'cuda':
                gd = torch.Generator(device=dev).manual_seed(seed)
                self.images = torch.randn(synthetic, 3, image_size, image_size, device=dev, generator=gd)
            else:
                self.images = torch.randn(synthetic, 3, image_size, image_size, generator=g)
            self.labels = self.labels.to(dev)
        else:
            labels = []
            for ci, cname in enumerate(self.class_names):
                d = self.root_dir / cname
                if not d.is_dir():
                    continue
                for f in sorted(d.iterdir()):
                    if f.suffix.lower() in IMG_EXT:
                        self.samples.append((f, ci))
                        labels.append(ci)
            if not self.samples:
                raise FileNotFoundError(f'no images under {self.root_dir}/<class name>/')
            self.labels = torch.tensor(labels, dtype=torch.long)
        sev = torch.tensor([self.severity_map[c] for c in self.class_names], dtype=torch.long, device=self.labels.device)
        self.severities = sev[self.labels]

    def __len__(self) -> int:
        return int(self.labels.numel())

    def _load(self, path: Path) -> torch.Tensor:
        from PIL import Image
        import numpy as np
        with Image.open(path) as im:
            im = im.convert('RGB').resize((self.image_size, self.image_size))
            return torch.from_numpy(np.asarray(im).copy()).permute(2, 0, 1).float().div_(255.0)

    def __getitem__(self, idx: int):
        img = self.images[idx] if self.images is not None else self._load(self.samples[idx][0])
        if self.transform is not None:
            img = self.transform(img)
        return img, self.labels[idx], self.severities[idx]

    def get_class_weights(self) -> torch.Tensor:
        """Inverse-frequency weights, mean 1 over the classes that occur (used as FocalLoss alpha, scripts/train.py:110-121)."""
        counts = torch.bincount(self.labels.cpu(), minlength=len(self.class_names)).float()
        w = counts.sum() / (len(self.class_names) * counts.clamp_min(1.0))
        return w / w[counts > 0].mean()


class DeviceBatchLoader:
    """DataLoader stand-in for a device-resident synthetic dataset: ``.dataset`` is a ``Subset`` (so
    ``loader.dataset.dataset`` is the ``RoseLeafDataset``, as scripts/train.py:110 expects), ``len()`` = batches,
    iteration yields ``(images, class_labels, severity_labels)``: the images sliced on the device (one index_select), the
    two label vectors as HOST tensors like a torch DataLoader's -- the reference's Trainer moves them with ``.to(device)``
    (training/trainer.py:79-82) and its Evaluator calls ``.numpy()`` on them (evaluation/evaluator.py:58,60).
    ``labels_on_device=True`` keeps them on the device instead (no host involvement at all in the step)."""

    def __init__(self, subset: Subset, batch_size: int, shuffle: bool, seed: int = 0, drop_last: bool = False,
                 labels_on_device: bool = False):
        self.dataset, self.batch_size, self.shuffle, self.drop_last = subset, batch_size, shuffle, drop_last
        self.labels_on_device = labels_on_device
        self._gen = torch.Generator().manual_seed(seed)
        base: RoseLeafDataset = subset.dataset
        self._idx = torch.as_tensor(subset.indices, dtype=torch.long)
        self._labels_host, self._sev_host = base.labels.cpu(), base.severities.cpu()

    def __len__(self) -> int:
        n = self._idx.numel()
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        base: RoseLeafDataset = self.dataset.dataset
        order = self._idx
        if self.shuffle:
            order = order[torch.randperm(order.numel(), generator=self._gen)]
        order_dev = order.to(base.images.device, non_blocking=True)
        for b in range(len(self)):
            lo, hi = b * self.batch_size, (b + 1) * self.batch_size
            sel = order_dev[lo:hi]
            imgs = base.images.index_select(0, sel)
            if base.transform is not None:
                imgs = base.transform(imgs)
            if self.labels_on_device:
                yield imgs, base.labels.index_select(0, sel), base.severities.index_select(0, sel)
            else:
                yield imgs, self._labels_host.index_select(0, order[lo:hi]), self._sev_host.index_select(0, order[lo:hi])


def _split(n: int, frac: float, seed: int) -> Tuple[List[int], List[int]]:
    perm = torch.randperm(n, generator=torch.Generator().manual_seed(seed)).tolist()
    k = int(round(n * frac))
    return perm[:k], perm[k:]


def create_dataloaders(augmented_root=None, original_root=None, class_names: Sequence[str] = (),
                       severity_map: Optional[Dict[str, int]] = None, augmented_transform: Optional[Callable] = None,
                       original_transform: Optional[Callable] = None, batch_size: int = 32, train_val_split: float = 0.8,
                       num_workers: int = 0, seed: int = 42, synthetic: Optional[int] = None,
                       device: Optional[torch.device] = None):
    """-> (train_loader, val_loader, test_loader); call as scripts/train.py:73-84 does.  ``synthetic=N`` (or
    ``$ROVIT_SYNTHETIC_DATA``) replaces both folders by N (train+val) and N//4 (test) seeded random images resident on
    ``device`` (default: the current CUDA/HIP device when there is one)."""
    if synthetic is None and not (augmented_root is not None and Path(augmented_root).exists()):
        env = os.environ.get('ROVIT_SYNTHETIC_DATA')
        synthetic = int(env) if env else None
    if synthetic is not None:
        if device is None:
            device = torch.device('cuda', torch.cuda.current_device()) if torch.cuda.is_available() else torch.device('cpu')
        full = RoseLeafDataset(None, class_names, severity_map, augmented_transform, 'augmented', synthetic, seed, device)
        test = RoseLeafDataset(None, class_names, severity_map, original_transform, 'original', max(1, synthetic // 4), seed + 1, device)
        tr, va = _split(len(full), train_val_split, seed)
        return (DeviceBatchLoader(Subset(full, tr), batch_size, True, seed),
                DeviceBatchLoader(Subset(full, va), batch_size, False, seed),
                DeviceBatchLoader(Subset(test, list(range(len(test)))), batch_size, False, seed))
    full = RoseLeafDataset(augmented_root, class_names, severity_map, augmented_transform, 'augmented')
    test = RoseLeafDataset(original_root, class_names, severity_map, original_transform, 'original')
    tr, va = _split(len(full), train_val_split, seed)
    g = torch.Generator().manual_seed(seed)
    kw = dict(batch_size=batch_size, num_workers=num_workers, pin_memory=torch.cuda.is_available())
    return (DataLoader(Subset(full, tr), shuffle=True, generator=g, **kw), DataLoader(Subset(full, va), shuffle=False, **kw),
            DataLoader(test, shuffle=False, **kw))
