"""Drop-in for the reference's ``models`` package (same module paths, class names, constructor signatures,
output dict and state_dict keys), computing on hand-written HIP kernels for gfx950."""
from .rovit_kan import RoViTKAN  # noqa: F401
from .backbone import DeiTTinyBackbone, freeze_backbone, get_backbone_output_dim  # noqa: F401
from .kan import BSplineBasis, KANLayer, KANSeverityModule  # noqa: F401
from .heads import ClassificationHead, OrdinalHead, UncertaintyHead  # noqa: F401
