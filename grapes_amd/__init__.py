"""grapes_amd — MI355X-native implementation of GRAPES' sample-then-aggregate training step.

Drop-in module layout mirrors the reference (dfdazac/grapes):
    grapes_amd.modules.utils : sample_neighborhoods_from_probs, get_neighborhoods, slice_adjacency, TensorMap
    grapes_amd.modules.gcn   : GCN
plus the device-resident pieces the reference does not have:
    grapes_amd.graph.DeviceGraph, grapes_amd.step.GrapesTrainer, grapes_amd.dist
All compute runs in hand-written HIP kernels (grapes_amd/csrc) behind the C-ABI of
include/grapes_hip.h; there is no CPU fallback.
"""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
