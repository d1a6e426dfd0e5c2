"""Drop-in for the `diff_gaussian_rasterization` package MANUS imports at
src/utils/gaussian_utils.py:18-21 (installed upstream by setup_env.sh:6-10).
Put the repo root on PYTHONPATH and unmodified MANUS code picks this up."""
from manus_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer  # noqa: F401
