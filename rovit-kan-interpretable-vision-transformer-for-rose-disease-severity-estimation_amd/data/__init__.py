"""The pieces of the reference's ``data`` package its training/evaluation loops touch:
batch mixing on the device (``transforms.cutmix_or_mixup``) and the loader surface (``dataset.RoseLeafDataset``,
``dataset.create_dataloaders``) with a device-resident synthetic mode."""
from .transforms import (cutmix_or_mixup, mix_images, rand_bbox, augmented_transforms, original_transforms,  # noqa: F401
                         inference_transforms)
from .dataset import RoseLeafDataset, create_dataloaders, DeviceBatchLoader  # noqa: F401
