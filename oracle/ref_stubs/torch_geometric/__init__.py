"""TEST INFRASTRUCTURE ONLY — near-empty stub for the absent torch_geometric (utils.py:4-9)."""
from . import data, datasets, transforms, utils, nn  # noqa: F401
