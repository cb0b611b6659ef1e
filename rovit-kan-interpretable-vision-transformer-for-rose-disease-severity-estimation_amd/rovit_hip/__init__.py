"""rovit_hip: MI355X-native kernels (librovit_hip.so) behind the RoViT-KAN nn.Module surface in ``models``."""
from . import native  # noqa: F401
from .native import RovitHipError, LIB_PATH  # noqa: F401
