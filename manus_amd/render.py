"""`render_gaussians` / `calculate_colors_from_sh` with the reference's signatures
(brown-ivl/manus src/utils/gaussian_utils.py:349-449), running on the HIP kernels."""
import math

import torch

from . import _lib
from .ops import sh_colors
from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer


def _tf12(tf):
    """(N,4,4) reference layout (passed on as it is: the SH operator reads its rows 0..2 in place) or (N,12)."""
    if tf is None:
        return None
    if tf.dim() == 3 and tuple(tf.shape[-2:]) == (4, 4):
        return tf
    return tf.reshape(tf.shape[0], 12)


def calculate_colors_from_sh(posed_means, cano_features, cano_means, camera, sh_degree, tf):
    """gaussian_utils.py:431-449.  cano_features (N,16,3); camera.camera_center (3,)/(1,3)."""
    if sh_degree != 3:
        raise ValueError("only sh_degree 3 is supported (MANUS fixes it, src/models/gaussian.py:29)")
    dev = posed_means.device
    def pack():
        cc = torch.as_tensor(camera.camera_center, dtype=torch.float32, device=dev).reshape(-1)
        if cc.is_cuda:     # (the SH kernel reads the camera centre only: one launch, no fill + slice copy)
            cams = torch.empty((1, _lib.MGR_CAM_FLOATS), dtype=torch.float32, device=dev)
            _lib.check(_lib.lib().mgr_pack_camera(0.0, 0.0, None, None, _lib.ptr(cc.contiguous()), _lib.ptr(cams), _lib.stream()), "mgr_pack_camera")
        else:
            cams = torch.zeros((1, _lib.MGR_CAM_FLOATS), dtype=torch.float32, device=dev)
            cams[0, 34:37] = cc[:3]
        return cams

    cams = _lib.cached_pack([camera.camera_center], ["sh", str(dev)], pack)
    if tf is not None:
        return sh_colors(cano_features, cano_means, _tf12(tf), cams)[0]
    return sh_colors(cano_features, posed_means, None, cams)[0]


def render_gaussians(posed_means, posed_cov, cano_means, cano_features, cano_opacity, camera, bg_color,
                     colors_precomp=None, sh_degree=3, tf=None, device=None):
    """gaussian_utils.py:349-428: same arguments, same returned dict
    (`render` (H,W,3), `viewspace_points`, `visibility_filter`, `radii`)."""
    device = posed_means.device if device is None else device
    screenspace_points = torch.zeros_like(posed_means, dtype=posed_means.dtype, requires_grad=True,
                                          device=device) + 0
    try:
        screenspace_points.retain_grad()
    except Exception:
        pass
    tanfovx = math.tan(float(camera.fovx) * 0.5)
    tanfovy = math.tan(float(camera.fovy) * 0.5)
    as_dev = lambda t: torch.as_tensor(t, dtype=torch.float32).to(device)
    raster_settings = GaussianRasterizationSettings(
        image_height=int(camera.height), image_width=int(camera.width), tanfovx=tanfovx, tanfovy=tanfovy,
        bg=as_dev(bg_color), scale_modifier=1, viewmatrix=as_dev(camera.world_view_transform),
        projmatrix=as_dev(camera.full_proj_transform), sh_degree=sh_degree,
        campos=as_dev(camera.camera_center), prefiltered=False, debug=False)
    rasterizer = GaussianRasterizer(raster_settings=raster_settings)
    if colors_precomp is None:
        colors_precomp = calculate_colors_from_sh(posed_means, cano_features, cano_means, camera, sh_degree, tf)
    rendered_image, radii = rasterizer(means3D=posed_means, means2D=screenspace_points, shs=None,
                                       colors_precomp=colors_precomp, opacities=cano_opacity, scales=None,
                                       rotations=None, cov3D_precomp=posed_cov)
    rendered_image = torch.permute(rendered_image, (1, 2, 0))
    return {"render": rendered_image, "viewspace_points": screenspace_points,
            "visibility_filter": radii > 0, "radii": radii}
