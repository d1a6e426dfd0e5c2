"""CPU oracle package — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
anything from here.  The product path (manus_amd/) never does; it fails loudly
when its HIP extension is missing.

* torch_ref.py      — pure-PyTorch restatement of the reference's LBS / covariance
                      / SH / camera / FK code (pinned by tests/golden/*.npz which
                      were produced by importing the reference).
* mesh_ref.py       — float64 numpy restatement of init_mano_weights (pinned by
                      tests/golden/mano_init.npz) and of the pysdf signed distance
                      it calls (PARITY UNPINNED: pysdf is not in this image).
* raster_oracle.c   — scalar C restatement of the external rasterizer + kNN
                      (PARITY UNPINNED upstream: no reference tests or vectors
                      exist for that boundary; see raster_oracle_impl.h).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("raster_oracle.c", "raster_oracle_impl.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        for sfx in ("_f32", "_f64"):
            getattr(_LIB, "orc_forward" + sfx).restype = ctypes.c_void_p
            getattr(_LIB, "orc_forward_ex" + sfx).restype = ctypes.c_void_p
            getattr(_LIB, "orc_forward_geom" + sfx).restype = ctypes.c_void_p
            getattr(_LIB, "orc_backward_geom" + sfx).restype = None
            getattr(_LIB, "orc_num_rendered" + sfx).restype = ctypes.c_long
            getattr(_LIB, "orc_collect_ambiguous" + sfx).restype = ctypes.c_long
            getattr(_LIB, "orc_reblend_with_overrides" + sfx).restype = None
            for fn in ("orc_free", "orc_get_geom", "orc_get_binning", "orc_get_image_state", "orc_backward"):
                getattr(_LIB, fn + sfx).restype = None
        _LIB.orc_knn3_mean_dist2.restype = None
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


class RasterOracle:
    """One forward (+ optional backward) of the scalar rasterizer.

    view/proj are the (4,4) `world_view_transform` / `full_proj_transform`
    tensors of the reference (row-major storage of the transposed matrices,
    i.e. column-major math matrices; cam_utils.py:58-63)."""

    def __init__(self, W, H, tanfovx, tanfovy, view, proj, means3D, cov3D, colors, opacity, bg,
                 dtype=np.float32, blend=True):
        self.dt = np.dtype(dtype)
        self.sfx = "_f32" if self.dt == np.float32 else "_f64"
        self.c_real = ctypes.c_float if self.dt == np.float32 else ctypes.c_double
        L = lib()
        c = lambda a, shape=None: np.ascontiguousarray(np.asarray(a, dtype=self.dt).reshape(shape or np.shape(a)))
        self.N = int(np.shape(means3D)[0])
        self.W, self.H = int(W), int(H)
        self.gx, self.gy = (self.W + 15) // 16, (self.H + 15) // 16
        view, proj = c(view, (16,)), c(proj, (16,))
        means3D, cov3D, colors = c(means3D, (self.N, 3)), c(cov3D, (self.N, 6)), c(colors, (self.N, 3))
        opacity, bg = c(opacity, (self.N,)), c(bg, (3,))
        self.color = np.zeros((3, self.H, self.W), dtype=self.dt)
        self.radii = np.zeros((self.N,), dtype=np.int32)
        # blend=False: the per-Gaussian preprocess only (radii, geom(), num_rendered; no image, no backward)
        self._h = getattr(L, "orc_forward_ex" + self.sfx)(
            ctypes.c_int(self.W), ctypes.c_int(self.H), self.c_real(tanfovx), self.c_real(tanfovy),
            _p(view), _p(proj), ctypes.c_int(self.N), _p(means3D), _p(cov3D), _p(colors), _p(opacity),
            _p(bg), _p(self.color), _p(self.radii), ctypes.c_int(1 if blend else 0))
        self._h = ctypes.c_void_p(self._h)
        self.num_rendered = int(getattr(L, "orc_num_rendered" + self.sfx)(self._h))

    def geom(self):
        xy = np.zeros((self.N, 2), self.dt); depth = np.zeros((self.N,), self.dt)
        co = np.zeros((self.N, 4), self.dt); tt = np.zeros((self.N,), np.int32)
        rect = np.zeros((self.N, 4), np.int32)
        getattr(lib(), "orc_get_geom" + self.sfx)(self._h, _p(xy), _p(depth), _p(co), _p(tt), _p(rect))
        return dict(xy=xy, depth=depth, conic_opacity=co, tiles_touched=tt, rect=rect)

    def binning(self):
        pl = np.zeros((max(self.num_rendered, 1),), np.int32)
        rg = np.zeros((self.gx * self.gy, 2), np.int32)
        getattr(lib(), "orc_get_binning" + self.sfx)(self._h, _p(pl), _p(rg))
        return pl[: self.num_rendered], rg

    def image_state(self):
        ft = np.zeros((self.H, self.W), self.dt); nc = np.zeros((self.H, self.W), np.int32)
        getattr(lib(), "orc_get_image_state" + self.sfx)(self._h, _p(ft), _p(nc))
        return ft, nc

    def backward(self, dL_dpix):
        g = np.ascontiguousarray(np.asarray(dL_dpix, dtype=self.dt).reshape(3, self.H, self.W))
        out = dict(means3D=np.zeros((self.N, 3), self.dt), means2D=np.zeros((self.N, 3), self.dt),
                   colors=np.zeros((self.N, 3), self.dt), opacity=np.zeros((self.N,), self.dt),
                   cov3D=np.zeros((self.N, 6), self.dt), conic=np.zeros((self.N, 3), self.dt))
        getattr(lib(), "orc_backward" + self.sfx)(
            self._h, _p(g), _p(out["means3D"]), _p(out["means2D"]), _p(out["colors"]),
            _p(out["opacity"]), _p(out["cov3D"]), _p(out["conic"]))
        return out

    # -- test aids for the alpha-threshold flip argument (tests/test_gpu_flips.py) ----------------------------------
    def ambiguous_pairs(self, eps=1e-4):
        """(pixel index, Gaussian, alpha) of the pairs the forward walk evaluated with |255 alpha - 1| <= eps."""
        cap = 1 << 16
        while True:
            pix, gid = np.zeros(cap, np.int32), np.zeros(cap, np.int32)
            al = np.zeros(cap, self.dt)
            c_real = ctypes.c_float if self.dt == np.float32 else ctypes.c_double
            n = int(getattr(lib(), "orc_collect_ambiguous" + self.sfx)(self._h, c_real(eps), ctypes.c_long(cap), _p(pix), _p(gid),
                                                                          _p(al)))
            if n <= cap:
                return pix[:n], gid[:n], al[:n]
            cap = n

    def reblend(self, pix, gid, keep, colors, bg, forced_last=None):
        """Force the outcome of the alpha >= 1/255 test for the given (pixel, Gaussian) pairs -- and, with `forced_last`
        (H,W) int, the Gaussian every pixel's walk ends on as its last contributor (-1: none) -- and composite again; `self.color` and the state the
        backward reads follow the forced decisions.  Returns (pixels whose own T < 1e-4 decision differed, those among
        them whose test value was not within rounding of the threshold)."""
        order = np.lexsort((np.asarray(gid), np.asarray(pix)))
        pix = np.ascontiguousarray(np.asarray(pix, np.int32)[order])
        gid = np.ascontiguousarray(np.asarray(gid, np.int32)[order])
        keep = np.ascontiguousarray(np.asarray(keep, np.int32)[order])
        colors = np.ascontiguousarray(np.asarray(colors, self.dt).reshape(self.N, 3))
        bg = np.ascontiguousarray(np.asarray(bg, self.dt).reshape(3))
        fl = None if forced_last is None else np.ascontiguousarray(np.asarray(forced_last, np.int32).reshape(self.H * self.W))
        sf, sv = ctypes.c_long(0), ctypes.c_long(0)
        getattr(lib(), "orc_reblend_with_overrides" + self.sfx)(self._h, ctypes.c_int(len(pix)), _p(pix), _p(gid), _p(keep), _p(fl),
                                                                _p(colors), _p(bg), _p(self.color), ctypes.byref(sf), ctypes.byref(sv))
        return int(sf.value), int(sv.value)

    def close(self):
        if self._h:
            getattr(lib(), "orc_free" + self.sfx)(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BlendOracle(RasterOracle):
    """The blend half of the scalar rasterizer on given screen-space state: xy (N,2) pixel centres, depth (N),
    conic (N,3) = (A,B,C), opacity (N), radii (N) int (0 = culled), colours (N,3).  `backward` returns the per-Gaussian
    sums the blend produces: means2D (N,3) in NDC-scaled pixels, conic (N,3), colors (N,3), opacity (N)."""

    def __init__(self, W, H, xy, depth, conic, opacity, radii, colors, bg, dtype=np.float32):
        self.dt = np.dtype(dtype)
        self.sfx = "_f32" if self.dt == np.float32 else "_f64"
        L = lib()
        c = lambda a, shape: np.ascontiguousarray(np.asarray(a, dtype=self.dt).reshape(shape))
        self.N = int(np.shape(xy)[0])
        self.W, self.H = int(W), int(H)
        self.gx, self.gy = (self.W + 15) // 16, (self.H + 15) // 16
        co = np.concatenate([c(conic, (self.N, 3)), c(opacity, (self.N, 1))], axis=1)
        self.radii = np.ascontiguousarray(np.asarray(radii, np.int32).reshape(self.N))
        self.color = np.zeros((3, self.H, self.W), dtype=self.dt)
        self._h = ctypes.c_void_p(getattr(L, "orc_forward_geom" + self.sfx)(
            ctypes.c_int(self.W), ctypes.c_int(self.H), ctypes.c_int(self.N), _p(c(xy, (self.N, 2))),
            _p(c(depth, (self.N,))), _p(np.ascontiguousarray(co)), _p(self.radii), _p(c(colors, (self.N, 3))),
            _p(c(bg, (3,))), _p(self.color)))
        self.num_rendered = int(getattr(L, "orc_num_rendered" + self.sfx)(self._h))

    def backward(self, dL_dpix):
        g = np.ascontiguousarray(np.asarray(dL_dpix, dtype=self.dt).reshape(3, self.H, self.W))
        out = dict(means2D=np.zeros((self.N, 3), self.dt), conic=np.zeros((self.N, 3), self.dt),
                   colors=np.zeros((self.N, 3), self.dt), opacity=np.zeros((self.N,), self.dt))
        getattr(lib(), "orc_backward_geom" + self.sfx)(self._h, _p(g), _p(out["means2D"]), _p(out["conic"]),
                                                       _p(out["colors"]), _p(out["opacity"]))
        return out


def knn3_mean_dist2(xyz):
    xyz = np.ascontiguousarray(np.asarray(xyz, dtype=np.float32))
    out = np.zeros((xyz.shape[0],), np.float32)
    lib().orc_knn3_mean_dist2(ctypes.c_int(xyz.shape[0]), _p(xyz), _p(out))
    return out
