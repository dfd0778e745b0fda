from ...pairs import ModelNetPairDataset  # noqa: F401
