"""Device-side pieces of the reference's ``data`` package that sit on the training hot path."""
from .transforms import cutmix_or_mixup, mix_images, rand_bbox  # noqa: F401
