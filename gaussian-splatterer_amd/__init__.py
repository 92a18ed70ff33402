"""gaussian-splatterer_amd — MI355X-native Gaussian-splat training step behind the reference's
ModelSplatsHost / ModelSplatsDevice / Trainer API (src/Trainer.cuh:49-73).

The compute path is the C-ABI library libgsplat_mi355.so (hand-written HIP for gfx950, sources in
csrc/); this package is the Python host-side mirror of the reference interface plus synthetic-input
and camera helpers.  There is no CPU fallback: any device entry point raises when the HIP library
is missing or no GPU is visible.
"""
from . import camera, capi, dist, synth  # noqa: F401
from .model import ModelSplatsDevice, ModelSplatsHost  # noqa: F401
from .trainer import CameraSphere, Project, Trainer  # noqa: F401
from . import driver, fields, io  # noqa: F401,E402
