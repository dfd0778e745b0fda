from ...pairs import OdometryKittiPairDataset  # noqa: F401
