from ...pairs import ThreeDMatchPairDataset  # noqa: F401
