"""manus_amd — MI355X-native articulated-3D-Gaussian rasterizer hot path of MANUS.

Host-side mirror (Python, as the reference is Python) of the reference's operator
interfaces for the one accelerated path, over the C ABI of libmanus_hip.so
(hand-written HIP kernels for gfx950).  See DESIGN.md and INTEGRATION.md.
"""
__version__ = "0.1.0"

from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer, rasterize_views  # noqa: F401
from .ops import distCUDA2, lbs_cov, project_points, sh_colors, skin_weights  # noqa: F401
from .render import calculate_colors_from_sh, render_gaussians  # noqa: F401
