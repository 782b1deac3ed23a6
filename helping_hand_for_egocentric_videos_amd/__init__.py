"""MI355X-native hot path of Chuhanxx/helping_hand_for_egocentric_videos.

Frozen LaViLa TimeSformer forward + object-query decoder forward/backward + EgoNCE / L1 / GIoU /
Hungarian losses, behind the reference's nn.Module signatures, running on hand-written gfx950 HIP
kernels reached through a C-ABI shared library (include/hh.h).  There is no CPU fallback: modules
raise if the HIP library is missing or tensors are not on the GPU.
"""
from .config import HHConfig, C1, C2, C4, TINY4, TINY16  # noqa: F401

__version__ = "0.1.0"
