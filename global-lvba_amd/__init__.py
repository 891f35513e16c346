"""global-lvba_amd -- MI355X-native LM-refinement hot path of Global-LVBA behind a C-ABI.

The directory name (with a hyphen) is fixed by the build contract; import it with
`importlib.import_module("global-lvba_amd")` or through the `lvba_amd` alias module at the repo root.
"""
from . import _lib
from .balm import BALM2, IMUST, VOX_HESS, BalmProblem, shard_range
from .visual import DepthImages, VisualProblem, fuse_tracks, optimize_camera_poses
from .voxel import Scans, VoxelMap

__all__ = ["BALM2", "IMUST", "VOX_HESS", "BalmProblem", "shard_range", "VisualProblem", "optimize_camera_poses", "VoxelMap", "Scans", "_lib"]
