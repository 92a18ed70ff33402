"""gaussian-splatterer_amd — MI355X-native Gaussian-splat training step behind the reference's
ModelSplatsHost / ModelSplatsDevice / Trainer API (src/Trainer.cuh:49-73).

The compute path is the C-ABI library csrc/libgsplat_mi355.so (hand-written HIP for gfx950);
this package is the Python host-side mirror of the reference interface plus synthetic-input and
camera helpers.  There is no CPU fallback: using any device entry point without the HIP library
raises.
"""
from . import camera, synth  # noqa: F401
