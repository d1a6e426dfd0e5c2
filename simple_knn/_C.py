"""Drop-in for `simple_knn._C` (MANUS: `from simple_knn._C import distCUDA2`,
src/models/gaussian.py:4,110)."""
from manus_amd.ops import distCUDA2  # noqa: F401
