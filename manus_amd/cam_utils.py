"""Camera matrices in the layout the operator expects (reference
src/utils/cam_utils.py:19-78).  float64 numpy, as the reference computes them."""
import math

import numpy as np


def focal2fov(focal, pixels):
    return 2 * math.atan(pixels / (2 * focal))


def fov2focal(fov, pixels):
    return pixels / (2 * math.tan(fov / 2))


def getProjectionMatrix(znear, zfar, fovX, fovY):
    """cam_utils.py:19-39.  Symmetric frustum: the principal point is dropped."""
    t = math.tan(fovY / 2) * znear
    r = math.tan(fovX / 2) * znear
    P = np.zeros((4, 4))
    P[0, 0] = 2.0 * znear / (r + r)
    P[1, 1] = 2.0 * znear / (t + t)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def get_opengl_camera_attributes(K, extrins, width, height, zfar=100.0, znear=0.01, resize_factor=1.0):
    """cam_utils.py:50-78.  K (3,3), extrins (3,4).  Returns the reference's dict:
    world_view_transform = [E;0001]^T, full_proj_transform = world_view_transform @ P^T,
    camera_center = inv(world_view_transform)[3,:3]."""
    K = np.array(K, dtype=np.float64)
    K[..., :2, :] = K[..., :2, :] * resize_factor
    width = int(width * resize_factor + 0.5)
    height = int(height * resize_factor + 0.5)
    fovx = focal2fov(K[0, 0], width)
    fovy = focal2fov(K[1, 1], height)
    E = np.concatenate([np.asarray(extrins, dtype=np.float64)[:3, :4], np.array([[0, 0, 0, 1.0]])], axis=0)
    wvt = E.T
    proj = getProjectionMatrix(znear=znear, zfar=zfar, fovX=fovx, fovY=fovy).T
    full = wvt @ proj
    center = np.linalg.inv(wvt)[3, :3]
    return {"width": width, "height": height, "fovx": fovx, "fovy": fovy, "K": K, "extr": E,
            "world_view_transform": wvt, "projection_matrix": proj, "full_proj_transform": full,
            "camera_center": center}


def get_scene_extent(cam_centers):
    """cam_utils.py:10-16; cam_centers (3,C)."""
    center = np.mean(cam_centers, axis=1, keepdims=True)
    return float(np.max(np.linalg.norm(cam_centers - center, axis=0, keepdims=True)) * 1.1)
