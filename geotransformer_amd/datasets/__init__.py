"""Dataset side of the hot path's callers: pair loaders, host transforms, benchmark file formats."""
from .pairs import ModelNetPairDataset, OdometryKittiPairDataset, ThreeDMatchPairDataset  # noqa: F401
