"""rgbd_pose_estimation_amd -- MI355X (gfx950) backend of the RGB-D absolute-pose hot path of
ShudaLi/rgbd_pose_estimation.  The product is ``lib/librgbdpose_hip.so`` (hand-written HIP kernels behind the
C ABI of include/rgbd_pose_hip.h) plus the drop-in C++ headers under ``include/pose``; this Python package
only loads the library, generates synthetic scenes and drives multi-GPU sharding."""
from . import _lib  # noqa: F401

__all__ = ["_lib", "api", "simulator", "build", "distributed"]
