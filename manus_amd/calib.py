"""Calibration-file cameras of the evaluation dataset (`TestDataset`, mode "acc_gt_eval" with a `.txt` camera file):
/root/reference/src/datasets/brics_dynamic.py:513-533 over /root/reference/src/utils/params.py:31-99.

`read_params` / `get_intr` / `get_extr` restate params.py (numpy only).  `optimal_new_camera_matrix` restates what the
reference obtains from OpenCV, `cv2.getOptimalNewCameraMatrix(intr, dist, (w, h), alpha=0, centerPrincipalPoint=True)`
(params.py:95-99): the published algorithm of OpenCV 4.x `calib3d` (getOptimalNewCameraMatrix -> icvGetRectangles ->
undistortPoints with its five fixed-point iterations) for the 4-coefficient model (k1, k2, p1, p2) params.py passes.
Which OpenCV: the reference installs an unpinned `opencv-python` (setup_env.sh:20), i.e. a current 4.x, whose
icvGetRectangles samples the border grid at x (w - 1) / (n - 1), y (h - 1) / (n - 1) (pixel centres 0 .. w - 1); releases
before the 3.4.14 / 4.5.2 fix sampled x w / (n - 1).  The current form is the default here, `legacy_grid=True` gives
the older one (the two differ by a fraction of a pixel in the rectangles, ~1e-3 relative in the focal lengths).

PARITY UNPINNED: OpenCV is not in this image, params.py cannot be imported without it, and the reference ships no
calibration file or expected values for this leg; tests/test_host_logic.py checks the restatement against closed forms
(no distortion; the centred scale formula; radial distortion inverted by the fixed-point iteration)."""
import numpy as np

_DTYPE = [("cam_id", int), ("width", int), ("height", int), ("fx", float), ("fy", float), ("cx", float), ("cy", float),
          ("k1", float), ("k2", float), ("p1", float), ("p2", float), ("cam_name", "<U22"), ("qvecw", float),
          ("qvecx", float), ("qvecy", float), ("qvecz", float), ("tvecx", float), ("tvecy", float), ("tvecz", float)]


def read_params(path):
    """params.py:57-92 (non-legacy): one row per camera, sorted by camera name."""
    params = np.atleast_1d(np.loadtxt(path, dtype=_DTYPE))
    return np.sort(params, order="cam_name")


def get_intr(param):
    """params.py:31-49 (distorted intrinsics + (k1, k2, p1, p2))."""
    intr = np.eye(3)
    intr[0, 0], intr[1, 1], intr[0, 2], intr[1, 2] = param["fx"], param["fy"], param["cx"], param["cy"]
    return intr, np.asarray([param["k1"], param["k2"], param["p1"], param["p2"]], dtype=np.float64)


def qvec2rotmat(q):
    """params.py:6-28 (w, x, y, z)."""
    w, x, y, z = q
    return np.array([[1 - 2 * y * y - 2 * z * z, 2 * x * y - 2 * w * z, 2 * z * x + 2 * w * y],
                     [2 * x * y + 2 * w * z, 1 - 2 * x * x - 2 * z * z, 2 * y * z - 2 * w * x],
                     [2 * z * x - 2 * w * y, 2 * y * z + 2 * w * x, 1 - 2 * x * x - 2 * y * y]])


def get_extr(param):
    """params.py:52-63: the 3x4 world-to-camera matrix."""
    r = qvec2rotmat([param["qvecw"], param["qvecx"], param["qvecy"], param["qvecz"]])
    t = np.asarray([param["tvecx"], param["tvecy"], param["tvecz"]], dtype=np.float64)
    return np.hstack([r, t[:, None]])


def undistort_points(pts, intr, dist, new_intr, iters=5):
    """OpenCV `undistortPoints` (R = identity, P = new_intr) for (k1, k2, p1, p2): normalise, five fixed-point iterations
    of x <- (x0 - tangential(x)) / radial(x), re-project.  pts (n, 2) pixel coordinates."""
    k1, k2, p1, p2 = [float(v) for v in dist]
    fx, fy, cx, cy = intr[0, 0], intr[1, 1], intr[0, 2], intr[1, 2]
    x0 = (np.asarray(pts[:, 0], np.float64) - cx) / fx
    y0 = (np.asarray(pts[:, 1], np.float64) - cy) / fy
    x, y = x0.copy(), y0.copy()
    for _ in range(iters):
        r2 = x * x + y * y
        icdist = 1.0 / (1.0 + (k2 * r2 + k1) * r2)
        dx = 2.0 * p1 * x * y + p2 * (r2 + 2.0 * x * x)
        dy = p1 * (r2 + 2.0 * y * y) + 2.0 * p2 * x * y
        x, y = (x0 - dx) * icdist, (y0 - dy) * icdist
    return np.stack([x * new_intr[0, 0] + new_intr[0, 2], y * new_intr[1, 1] + new_intr[1, 2]], -1)


def _rectangles(intr, dist, size, n=9, legacy_grid=False):
    """icvGetRectangles: the border grid of n x n points undistorted with the same matrix; the largest inscribed and
    the bounding rectangle (x, y, w, h) of the result.  Grid and results in float32 like OpenCV's CV_32FC2 buffer."""
    w, h = size
    ex, ey = (w, h) if legacy_grid else (w - 1, h - 1)
    gx = (np.arange(n, dtype=np.float32) * np.float32(ex) / np.float32(n - 1))
    gy = (np.arange(n, dtype=np.float32) * np.float32(ey) / np.float32(n - 1))
    grid = np.stack(np.meshgrid(gx, gy, indexing="xy"), -1).reshape(-1, 2)       # row-major: y outer, x inner
    p = undistort_points(grid, intr, dist, intr).astype(np.float32).reshape(n, n, 2)
    ix0, ix1 = p[:, 0, 0].max(), p[:, n - 1, 0].min()
    iy0, iy1 = p[0, :, 1].max(), p[n - 1, :, 1].min()
    ox0, ox1, oy0, oy1 = p[..., 0].min(), p[..., 0].max(), p[..., 1].min(), p[..., 1].max()
    return (ix0, iy0, ix1 - ix0, iy1 - iy0), (ox0, oy0, ox1 - ox0, oy1 - oy0)


def optimal_new_camera_matrix(intr, dist, size, alpha=0.0, legacy_grid=False):
    """cv2.getOptimalNewCameraMatrix(intr, dist, size, alpha, centerPrincipalPoint=True) -> (new_intr, roi): the
    principal point moves to the image centre ((w - 1) / 2, (h - 1) / 2) and both focal lengths are scaled by
    s = s0 (1 - alpha) + s1 alpha, s0 / s1 the scales at which the inscribed / the bounding rectangle of the undistorted
    image fills the frame."""
    w, h = size
    m = np.array(intr, dtype=np.float64)
    cx0, cy0 = m[0, 2], m[1, 2]
    cx, cy = (w - 1) * 0.5, (h - 1) * 0.5
    (ix, iy, iw, ih), (ox, oy, ow, oh) = [tuple(float(v) for v in r) for r in _rectangles(m, dist, size, legacy_grid=legacy_grid)]
    s0 = max(cx / (cx0 - ix), cy / (cy0 - iy), cx / (ix + iw - cx0), cy / (iy + ih - cy0))
    s1 = min(cx / (cx0 - ox), cy / (cy0 - oy), cx / (ox + ow - cx0), cy / (oy + oh - cy0))
    s = s0 * (1.0 - alpha) + s1 * alpha
    m[0, 0] *= s
    m[1, 1] *= s
    m[0, 2], m[1, 2] = cx, cy
    x0, y0 = int(np.floor((ix - cx0) * s + cx)), int(np.floor((iy - cy0) * s + cy))
    x1, y1 = int(np.ceil((ix + iw - cx0) * s + cx)), int(np.ceil((iy + ih - cy0) * s + cy))
    x0, y0, x1, y1 = max(x0, 0), max(y0, 0), min(x1, w), min(y1, h)
    return m, (x0, y0, max(x1 - x0, 0), max(y1 - y0, 0))


def camera_table_from_calibration(path, width, height):
    """brics_dynamic.py:513-531: {"intrs": [fx fy cx cy] of the undistorted optimal matrix, "extrs": 3x4, "cam_name"}."""
    intrs, extrs, names = [], [], []
    for param in read_params(path):
        names.append(str(param["cam_name"]))
        intr, dist = get_intr(param)
        new_intr, _ = optimal_new_camera_matrix(intr, dist, (int(width), int(height)), alpha=0.0)
        intrs.append([new_intr[0, 0], new_intr[1, 1], new_intr[0, 2], new_intr[1, 2]])
        extrs.append(get_extr(param))
    return {"intrs": intrs, "extrs": extrs, "cam_name": names}
