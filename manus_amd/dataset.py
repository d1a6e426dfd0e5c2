"""Sequence reader for MANUS-Grasps captures (dataloader half): the reference's
`src/datasets/brics_dynamic.py::Dataset` over the same on-disk schema, feeding `engine.HipViewCompute`.

Schema of one action file (brics_dynamic.py:173-188, 224-263, 343-403):

    frames/<frame no>/images/<camera>     (h, w, 4) uint8 RGBA crop
    frames/<frame no>/bbox/<camera>       (4,) int  xmin, ymin, xmax, ymax of the crop in the full image
    frames/<frame no>/metadata/{bnames, bnames_parent, rest_heads, rest_tails, rest_matrixs, pose_heads, pose_tails,
                                pose_matrixs, eulers, root_translation, root_rotation}
    K/<camera> (3,3)     extr/<camera> (3,4)     mano_rest/<key>

The reference opens `<action>.hdf5` with h5py.  h5py is an optional import here: everything below talks to the file
through the small mapping protocol h5py groups offer (`g[name]`, `g.keys()`, `g.get`, `g.items()`, `dataset[:]`), and
`TreeStore` offers the same protocol over one `.npz` whose keys are the '/'-joined paths -- the container used for the
synthetic sequences of the tests and for captures converted on a machine that has h5py (`convert_hdf5`).
Host-side IO only; nothing here is on the per-step path."""
import json
import collections
import os
import re
from dataclasses import dataclass

import numpy as np
import torch

from . import transforms as T
from .cam_utils import get_opengl_camera_attributes, get_scene_extent


# ---------------------------------------------------------------------------------------------------------------------
# containers
# ---------------------------------------------------------------------------------------------------------------------
class _Node:
    """A group of a TreeStore: children by name, h5py.Group protocol.  `arrays` maps '/'-joined paths to arrays (read
    lazily from the .npz on first access), `index` maps every group prefix to its sorted child names (built once)."""

    def __init__(self, arrays, prefix, index):
        self._a, self._p, self._idx = arrays, prefix, index
        self._names = index.get(prefix, [])  # by name, like h5py's default iteration order

    def keys(self):
        return list(self._names)

    def __iter__(self):
        return iter(self._names)

    def __len__(self):
        return len(self._names)

    def __contains__(self, name):
        return name in self._names

    def __getitem__(self, name):
        full = self._p + name
        if full in self._a:
            return self._a[full]                    # numpy array: `[:]`, `[idx]` behave like an h5py dataset
        if name not in self._names:
            raise KeyError(full)
        return _Node(self._a, full + "/", self._idx)

    def get(self, name, default=None):
        return self[name] if name in self._names else default

    def items(self):
        return [(n, self[n]) for n in self._names]


class TreeStore(_Node):
    """Read-only h5py.File look-alike over an .npz of '/'-joined paths (context manager like h5py.File)."""

    def __init__(self, path, mode="r"):
        if mode != "r":
            raise ValueError("TreeStore is read-only; write with write_tree()")
        self._z = np.load(path, allow_pickle=False)     # kept open: members are decompressed when they are indexed
        index = {}
        for k in self._z.files:
            parts = k.split("/")
            for d in range(len(parts)):
                index.setdefault("/".join(parts[:d]) + ("/" if d else ""), set()).add(parts[d])
        index = {pre: sorted(ch) for pre, ch in index.items()}
        super().__init__(_LazyArrays(self._z), "", index)

    def close(self):
        self._z.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False     # (stays open: open_sequence caches one store per file)


class _LazyArrays:
    """Mapping view of an NpzFile that loads a member on access and keeps the most recently used ones (bounded: a
    capture file holds thousands of image crops; the small shared members -- K, extr, metadata -- are what repeats)."""
    MAX_BYTES = 64 << 20

    def __init__(self, z):
        self._z, self._names = z, set(z.files)
        self._cache, self._bytes = collections.OrderedDict(), 0

    def __contains__(self, k):
        return k in self._names

    def __getitem__(self, k):
        a = self._cache.get(k)
        if a is not None:
            self._cache.move_to_end(k)
            return a
        a = self._z[k]
        self._cache[k] = a
        self._bytes += a.nbytes
        while self._bytes > self.MAX_BYTES and len(self._cache) > 1:
            _, old = self._cache.popitem(last=False)
            self._bytes -= old.nbytes
        return a


def write_tree(path, arrays):
    """arrays: {'/'-joined path: ndarray} -> one uncompressed .npz, written to exactly `path` (np.savez would append
    '.npz')."""
    with open(path, "wb") as f:
        np.savez(f, **arrays)


# One open store per action file AND PROCESS (the Dataset opens the file on every __getitem__).  A DataLoader worker is
# a fork of the process that built the dataset: it must not read through the parent's file descriptor -- zipfile
# serialises seek + read with a per-process lock only, so two workers on one descriptor race -- hence the cache is keyed
# by pid and emptied in every forked child; a file rewritten under the same name replaces (and closes) its stale store.
_STORES = {}


def _drop_stores_after_fork():
    _STORES.clear()     # (the parent's handles stay the parent's: not closed here, just never used by the child)


if hasattr(os, "register_at_fork"):
    os.register_at_fork(after_in_child=_drop_stores_after_fork)


def close_sequences():
    """Close every cached action file of this process."""
    for _, st in _STORES.values():
        st.close()
    _STORES.clear()


def open_sequence(path):
    """An action file by content: .npz container -> TreeStore, otherwise HDF5 through h5py (fails loudly without it)."""
    with open(path, "rb") as f:
        magic = f.read(4)
    if magic[:2] == b"PK":
        key = (os.getpid(), os.path.abspath(path))
        mtime = os.path.getmtime(path)
        ent = _STORES.get(key)
        if ent is not None and ent[0] != mtime:     # rewritten since it was opened
            ent[1].close()
            ent = None
        if ent is None:
            ent = _STORES[key] = (mtime, TreeStore(path))
        return ent[1]
    try:
        import h5py
    except ImportError as e:
        raise RuntimeError("%s is an HDF5 file and h5py is not installed here; convert it with "
                           "manus_amd.dataset.convert_hdf5 on a machine that has h5py" % path) from e
    return h5py.File(path, "r")


def convert_hdf5(src, dst):
    """Flatten an HDF5 action file into the .npz container (needs h5py)."""
    import h5py
    out = {}

    def visit(name, obj):
        if isinstance(obj, h5py.Dataset):
            a = obj[()]
            out[name] = a.astype("S") if a.dtype.kind == "O" else a
    with h5py.File(src, "r") as f:
        f.visititems(visit)
    write_tree(dst, out)


def natural_key(s):
    """Sort key of natsort.natsorted for the names met here (digit runs compare as integers)."""
    return [(0, int(t), "") if t.isdigit() else (1, 0, t) for t in re.split(r"(\d+)", str(s)) if t != ""]


def natsorted(names):
    return sorted(names, key=natural_key)


# ---------------------------------------------------------------------------------------------------------------------
# structures (src/utils/structures.py:8-47)
# ---------------------------------------------------------------------------------------------------------------------
def _index_fields(obj, idx):
    return type(obj)(**{k: (None if v is None else v[idx]) for k, v in obj.__dict__.items()})


from .structures import Bones, Cameras  # noqa: E402,F401  (one definition: src/utils/structures.py:7-47)


def to_tensor(var, dtype=torch.float32):
    """extra.py:56-82 for the cases met here: arrays / lists of numbers become tensors, dataclasses are converted
    field by field (in place, like the reference), strings and dicts of non-numbers stay."""
    if hasattr(var, "__dataclass_fields__"):
        for k, v in var.__dict__.items():
            if v is not None:
                setattr(var, k, to_tensor(v, dtype))
        return var
    if isinstance(var, np.ndarray):
        if var.ndim > 0 and var.dtype.kind in "fiub":
            return torch.tensor(var, dtype=dtype)
        return var
    if isinstance(var, (list, tuple)):
        try:
            return torch.tensor(var, dtype=dtype)
        except (TypeError, ValueError):
            return var
    return var


# ---------------------------------------------------------------------------------------------------------------------
# pose helpers (src/utils/transforms.py:145-198, 371-419, 478-486)
# ---------------------------------------------------------------------------------------------------------------------
DOF_XZ = ("bone_0", "bone_1", "bone_2", "bone_5", "bone_9", "bone_13", "bone_17")
DOF_X = ("bone_3", "bone_6", "bone_7", "bone_10", "bone_11", "bone_14", "bone_15", "bone_18", "bone_19")


def apply_constraints_to_poses(euler, bnames, dof_xz=DOF_XZ, dof_xyz=(), dof_x=DOF_X):
    """(F, J, 3) Euler angles -> (F, n_dof) free angles, bones in `bnames` order (transforms.py:371-419; note the
    single-DoF bones keep the Z angle, column 2)."""
    cols = []
    for i, bn in enumerate(bnames):
        if bn in dof_xyz:
            cols += [euler[:, i, 0], euler[:, i, 1], euler[:, i, 2]]
        elif bn in dof_xz:
            cols += [euler[:, i, 0], euler[:, i, 2]]
        elif bn in dof_x:
            cols += [euler[:, i, 2]]
    width = 2 * len(dof_xz) + 3 * len(dof_xyz) + len(dof_x)
    out = np.zeros((euler.shape[0], width), dtype=np.float32)
    for c, col in enumerate(cols):
        out[:, c] = col
    return out


def matrix_to_quaternion(m):
    """Rotation matrices (..., 3, 3) -> quaternions (..., 4), real part first: of the four algebraically equal
    candidates (one per component taken as the pivot) the one with the largest pivot (transforms.py:145-198)."""
    if m.shape[-2:] != (3, 3):
        raise ValueError(f"Invalid rotation matrix shape {m.shape}.")
    batch = m.shape[:-2]
    f = m.reshape(-1, 9)
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = f.unbind(-1)
    sq = torch.stack([1.0 + m00 + m11 + m22, 1.0 + m00 - m11 - m22, 1.0 - m00 + m11 - m22, 1.0 - m00 - m11 + m22], -1)
    q_abs = torch.sqrt(sq.clamp_min(0.0))
    pivot = q_abs.argmax(-1)
    rows = torch.stack([
        torch.stack([q_abs[:, 0] ** 2, m21 - m12, m02 - m20, m10 - m01], -1),
        torch.stack([m21 - m12, q_abs[:, 1] ** 2, m10 + m01, m02 + m20], -1),
        torch.stack([m02 - m20, m10 + m01, q_abs[:, 2] ** 2, m12 + m21], -1),
        torch.stack([m10 - m01, m20 + m02, m21 + m12, q_abs[:, 3] ** 2], -1)], -2)          # (n, 4 candidates, 4)
    cand = rows / (2.0 * q_abs.clamp_min(0.1))[..., None]
    return cand[torch.arange(f.shape[0]), pivot].reshape(batch + (4,))


def euler_angles_to_quats(euler):
    return matrix_to_quaternion(T.euler_angles_to_matrix(euler, "XYZ", intrinsic=True))


# ---------------------------------------------------------------------------------------------------------------------
# the dataset
# ---------------------------------------------------------------------------------------------------------------------
DEFAULT_OPTS = dict(sequences="all", split_by_action=False, num_time_steps=-1, split_ratio=-1.0,
                    rand_views_per_timestep=-1, n_bones=20, resize_factor=1.0, bg_color="white", width=1920,
                    height=1080, near=0.01, far=100.0, subject="subject")


def _area_resize(img, factor):
    """cv2.resize(..., fx=f, fy=f, INTER_AREA) for f = 1 (identity) and f = 1/k (mean over k x k blocks, rounded like
    cv2 does for uint8).  Other factors need OpenCV."""
    if factor == 1.0:
        return img
    k = round(1.0 / factor)
    if k >= 2 and abs(1.0 / k - factor) < 1e-9 and img.shape[0] % k == 0 and img.shape[1] % k == 0:
        h, w = img.shape[0] // k, img.shape[1] // k
        m = img.reshape(h, k, w, k, -1).astype(np.float64).mean((1, 3))
        return np.floor(m + 0.5).astype(img.dtype) if img.dtype.kind in "ui" else m.astype(img.dtype)
    try:
        import cv2
    except ImportError as e:
        raise RuntimeError("resize_factor %r needs OpenCV (only 1 and 1/k on divisible sizes are built in)" % factor) from e
    return cv2.resize(img, (0, 0), fx=factor, fy=factor, interpolation=cv2.INTER_AREA)


@dataclass
class _Take:
    """One recorded action of the subject: its container file, frame keys and cameras."""
    name: str                 # file name up to the first '.', the key of metadata_dict and of the items
    file: str                 # file name inside the dataset directory
    path: str
    frames_as_stored: list    # frame keys in the container's order
    frames: list              # ... in natural order (the order items are generated in)
    cams: list


def _scan_takes(root_dir, wanted="all"):
    """The takes of a subject directory in natural file order, or those named by `wanted` in the order given."""
    files = natsorted([f for f in os.listdir(root_dir) if f.endswith((".hdf5", ".npz"))])
    if wanted != "all":
        files = [f for w in wanted for f in files if f.rsplit(".", 1)[0] == w]
    takes = []
    for f in files:
        path = os.path.join(root_dir, f)
        with open_sequence(path) as file:
            stored, cams = list(file["frames"].keys()), list(file.get("K").keys())
        takes.append(_Take(f.split(".")[0], f, path, stored, natsorted(stored), cams))
    return takes


def _strided(frames, n_steps):
    """Every (len // n_steps)-th frame -- all of them when n_steps is negative or exceeds their number."""
    if n_steps < 0 or n_steps > len(frames):
        return list(frames)
    return list(frames[::len(frames) // n_steps])


def _split_part(seq, ratio, split):
    """The train (front) or test (back) part of a sequence cut at int(ratio * len); the whole of it for ratio <= 0."""
    if ratio <= 0:
        return list(seq)
    cut = int(ratio * len(seq))
    return list(seq[:cut] if split == "train" else seq[cut:])


class SequenceDataset(torch.utils.data.Dataset):
    """brics_dynamic.py::Dataset: one subject, several actions, every (action, frame, camera) one item (or one item per
    (action, frame) with `rand_views_per_timestep` random cameras)."""

    def __init__(self, root_dir, opts=None, split="train", split_file_dir=None):
        o = dict(DEFAULT_OPTS)
        o.update(opts or {})
        self.opts, self.split, self.root_dir = o, split, root_dir
        self.training = split == "train"
        self.resize_factor, self.bg_color = o["resize_factor"], o["bg_color"]
        self.width, self.height, self.subject_id = o["width"], o["height"], o["subject"]
        self.split_file_dir = split_file_dir
        takes = self._takes_of_split(_scan_takes(root_dir, o["sequences"]), split, o["split_ratio"], o["split_by_action"])
        self.actions = [t.file for t in takes]
        self.metadata_dict = {t.name: self._pose_records(t) for t in takes}
        # the item table: (take, frame) rows in take order, every `stride`-th frame of a take; one item per row and camera,
        # or one per row when the cameras are drawn at random per item (rand_views_per_timestep >= 0)
        rows = [(t, f) for t in takes for f in _strided(t.frames, o["num_time_steps"])]
        per_camera = o["rand_views_per_timestep"] < 0
        items = [(t.name, f, c) for t, f in rows for c in (t.cams if per_camera else (None,))]
        if not o["split_by_action"]:
            # (a single take with split_by_action is the one case where the ratio is ignored altogether: _takes_of_split)
            items = _split_part(items, o["split_ratio"], split)
            if split_file_dir is not None:           # the reference always drops ./<split>_split.json in the cwd
                with open(os.path.join(split_file_dir, "%s_split.json" % split), "w") as f:
                    json.dump(items, f)
        self.index_list = items
        # the camera table holds the rig once per kept frame of the LAST take (brics_dynamic.py:215-263 repeats it like
        # that; items only ever index its first copy through cam2idx)
        self._load_rig(copies=len(_strided(takes[-1].frames, o["num_time_steps"])) if takes else 0)

    def _path(self, action):
        for ext in (".hdf5", ".npz"):
            p = os.path.join(self.root_dir, action + ext)
            if os.path.exists(p):
                return p
        raise FileNotFoundError(os.path.join(self.root_dir, action + ".hdf5"))

    @staticmethod
    def _takes_of_split(takes, split, ratio, by_action):
        """Train / test by whole takes (brics_dynamic.py:156-165): the first int(ratio * n) takes train, the rest test;
        one take alone is never split this way."""
        if by_action and len(takes) > 1:
            return _split_part(takes, ratio, split)
        return takes

    def _pose_records(self, take):
        """{frame: pose record} of EVERY frame of a take (brics_dynamic.py:173-188 reads them all, kept or not)."""
        out = {}
        with open_sequence(take.path) as file:
            for f in take.frames_as_stored:
                rec = self._pose_record(file["frames"][f]["metadata"])
                rec["frame_id"], rec["action"] = f, take.name
                out[f] = rec
        return out

    def _load_rig(self, copies):
        """all_cameras / cam_names / cam2idx / mano_data / extent from the first item's take (one rig per subject)."""
        cols = collections.defaultdict(list)
        with open_sequence(self._path(self.index_list[0][0])) as file:
            mano = file.get("mano_rest")
            self.mano_data = {k: v[:] for k, v in mano.items()} if mano is not None else {}
            Ks, extrs = file.get("K"), file.get("extr")
            self.cam_names = list(Ks.keys())
            self.cam2idx = {c: i for i, c in enumerate(self.cam_names)}
            rig = [(c, get_opengl_camera_attributes(Ks[c][:], extrs[c][:], self.width, self.height, resize_factor=self.resize_factor))
                   for c in self.cam_names]
        for _ in range(copies):
            for cam, attrs in rig:
                for k, v in attrs.items():
                    cols[k].append(v)
                cols["cam_name"].append(cam)
        self.all_cameras = Cameras(**{k: np.stack(v, 0) for k, v in cols.items()})
        self.extent = get_scene_extent(self.all_cameras.camera_center)

    def __len__(self):
        return len(self.index_list)

    def __getitem__(self, idx):
        return self.fetch_data(idx)

    def fetch_data_by_frame(self, action, frame_id, cam_name):
        try:
            return self.fetch_data(self.index_list.index((action, frame_id, cam_name)))
        except (ValueError, KeyError):
            return None

    def _pose_record(self, g):
        """One frame's metadata group -> {"bones_rest", "bones_posed", "pose_latent"} (the dict brics_dynamic.py:279-327
        builds: the first n_bones rows of the rest / posed armature, constrained Euler angles, the kinematic tree, and the
        latent = quaternions of the joint angles followed by the root rotation)."""
        def strings(key):
            return [n[0].decode("UTF-8") if isinstance(n[0], bytes) else str(n[0]) for n in g[key][:].tolist()]
        rows = list(range(self.opts["n_bones"]))
        names = np.array(strings("bnames"))
        armature = {kind: dict(bnames=names, heads=g[kind + "_heads"][rows], tails=g[kind + "_tails"][rows], transforms=g[kind + "_matrixs"][rows])
                    for kind in ("rest", "pose")}
        angles, root_R = g["eulers"][:], g["root_rotation"][:]
        posed = Bones(**armature["pose"], eulers=angles, eulers_c=apply_constraints_to_poses(angles[None], names),
                      root_translation=g["root_translation"][:], root_rotation=root_R, kintree=T.build_kintree(names, strings("bnames_parent")))
        latent = euler_angles_to_quats(torch.tensor(np.concatenate([angles, root_R[None]], 0), dtype=torch.float32)).flatten()
        return {"bones_rest": Bones(**armature["rest"]), "bones_posed": posed, "pose_latent": latent}

    # -- brics_dynamic.py:334-373 ---------------------------------------------------------------------------
    def get_bg_color(self):
        if self.bg_color == "random":
            return np.random.rand(3).astype(np.float32)
        if self.bg_color == "white":
            return np.ones(3, np.float32)
        if self.bg_color == "black":
            return np.zeros(3, np.float32)
        raise ValueError("bg_color %r" % (self.bg_color,))

    def fetch_images(self, data, cam_name):
        """The RGBA crop pasted at its bbox into a full frame, resized, scaled to [0,1] and composited on the background
        colour (the alpha channel is returned untouched: it is the segmentation mask)."""
        img = np.zeros((self.height, self.width, 4), dtype=np.uint8)
        xmin, ymin, xmax, ymax = (int(t) for t in data["bbox"][cam_name][:])
        img[ymin:ymax, xmin:xmax, :] = data["images"][cam_name][:]
        img = _area_resize(img, self.resize_factor) / 255.0
        bkgd = self.get_bg_color()
        alpha = img[..., 3:]
        img[..., :3] = img[..., :3] * alpha + bkgd * (1.0 - alpha)
        return img

    # -- brics_dynamic.py:375-424 ---------------------------------------------------------------------------
    def get_data_from_h5(self, index):
        action, frame_id, cam_name = self.index_list[index]
        if cam_name is None:
            cams = list(np.random.default_rng().choice(self.cam_names, size=self.opts["rand_views_per_timestep"], replace=False))
        else:
            cams = [cam_name]
        with open_sequence(self._path(action)) as file:
            data = file.get("frames")[str(frame_id)]
            rgba = np.array([self.fetch_images(data, c) for c in cams])
        cameras = self.all_cameras[[self.cam2idx[c] for c in cams]]
        return rgba, cameras, self.metadata_dict[action][str(frame_id)], [self.subject_id, action, frame_id, cams]

    def fetch_data(self, index):
        rgba, camera, md, info = self.get_data_from_h5(index)
        rest, posed = Bones(**md["bones_rest"].__dict__), Bones(**md["bones_posed"].__dict__)   # (to_tensor converts in place)
        return {"info": info, "rgb": to_tensor(rgba[..., :3]), "mask": to_tensor(rgba[..., 3:]), "camera": to_tensor(camera),
                "scaling_modifier": 1.0, "bg_color": to_tensor(self.get_bg_color()), "bones_rest": to_tensor(rest),
                "bones_posed": to_tensor(posed), "pose_latent": md["pose_latent"]}

    # -- model initialisation (brics_dynamic.py:69-144) -------------------------------------------------------
    def _first_rest(self):
        action, fno = self.index_list[0][0], self.index_list[0][1]
        return to_tensor(Bones(**self.metadata_dict[action][str(fno)]["bones_rest"].__dict__))

    def sample_gaussians_on_bones(self, sample_size, mano_weights=False, init_type=None, generator=None, device=None):
        """(points, point colours, MANO skin weights or None): `sample_size` Gaussians per bone + half as many per joint
        around the first frame's rest skeleton (train_utils.py:104-139; the draws come from `generator`, not from the
        reference's global RNG stream), their weights from the nearest MANO vertices (`mano_init.init_mano_weights`; GPU).
        The reference's .ply dumps are left to the caller."""
        from . import mano_init, synthetic as S
        b = self._first_rest()
        gen = generator or torch.Generator().manual_seed(0)
        points = S.sample_on_bones(b.heads, b.tails, b.transforms, sample_size, gen)
        colors = torch.rand(points.shape, generator=gen)
        weights = None
        if mano_weights:
            if init_type not in ("mano_init_voxel", "mano_init_points"):
                raise ValueError("init_type must be 'mano_init_voxel' or 'mano_init_points'")
            w, _ = mano_init.init_mano_weights(points, self.mano_data, filter_grid=init_type == "mano_init_voxel", device=device)
            weights = to_tensor(w)
        return points, colors, weights

    def build_voxel_grid(self, grid_boundary=(-1, 1), res=128, ratio=(1, 1, 1), offset=(0, 0, 0), device=None):
        """(scale, center, grid_points, weights, mask) of the skin-weight voxel grid around the first frame's rest
        skeleton (brics_dynamic.py:99-144): `mano_init.build_voxel_grid` on this capture's `mano_rest` group."""
        from . import mano_init
        return mano_init.build_voxel_grid(self._first_rest(), self.mano_data, grid_boundary, res, ratio, offset, device)

    # -- hand-off to the engine --------------------------------------------------------------------------------
    def view_batch(self, indices, device="cpu"):
        """Items -> what `engine.HipViewCompute` consumes for one step: targets (V,3,H,W), masks (V,H,W), the camera
        dicts, posed bone transforms (V,J,4,4) and keypoints (V,J+1,3) = first head + all tails
        (hand_dynamic.py:199-204)."""
        items = [self.fetch_data(i) for i in indices]
        out = _skeleton_batch(items, device, camera_row=0)
        out["targets"] = torch.stack([it["rgb"][0].permute(2, 0, 1) for it in items]).to(device)
        out["masks"] = torch.stack([it["mask"][0, ..., 0] for it in items]).to(device)
        return out


def _skeleton_batch(items, device, camera_row=None):
    """Cameras, posed / rest bone transforms and keypoints of a list of dataset items (train items hold a 1-camera
    table per item: `camera_row=0`; evaluation items hold one camera)."""
    cams = []
    for it in items:
        c = it["camera"].__dict__
        cams.append({k: (v if camera_row is None else v[camera_row]) for k, v in c.items()})
    return {"cameras": cams,
            "posed": torch.stack([it["bones_posed"].transforms for it in items]).to(device),
            "rest": items[0]["bones_rest"].transforms.to(device),
            "keypoints": torch.stack([torch.cat([it["bones_posed"].heads[:1], it["bones_posed"].tails], 0) for it in items]).to(device)}


# ---------------------------------------------------------------------------------------------------------------------
# evaluation trajectories (src/datasets/brics_dynamic.py:485-696)
# ---------------------------------------------------------------------------------------------------------------------
def convert_armature_space_to_world_space(md):
    """Rest / posed skeleton tables from armature space into world space: every 4x4 is left-multiplied by the matching
    `*_matrix_world`, heads and tails are mapped as points (transforms.py:561-590).  Returns a new dict; float64 in,
    float64 out like the reference's numpy einsums."""
    out = dict(md)
    Rw, Pw = np.asarray(md["rest_matrix_world"]), np.asarray(md["pose_matrix_world"])
    out["rest_matrixs"] = Rw @ np.asarray(md["rest_matrixs"])
    out["pose_matrixs"] = Pw @ np.asarray(md["pose_matrixs"])
    for mat, keys in ((Rw, ("rest_tails", "rest_heads")), (Pw, ("pose_tails", "pose_heads"))):
        for k in keys:
            p = np.asarray(md[k])
            h = np.concatenate([p, np.ones(p.shape[:-1] + (1,))], -1)[..., None]
            out[k] = (mat @ h)[..., :3, 0]
    return out


def _load_table(path, size=None):
    """A camera-path / skeleton table: the reference's joblib pickle, the same dict as one `.npz`, or -- cameras only,
    `size` = (width, height) -- a calibration `.txt` (brics_dynamic.py:513-531; manus_amd/calib.py, parity unpinned)."""
    if ".txt" in path and size is not None:      # (the reference's own test, brics_dynamic.py:512)
        from .calib import camera_table_from_calibration
        return camera_table_from_calibration(path, size[0], size[1])
    if path.endswith(".npz"):
        with np.load(path, allow_pickle=False) as z:
            return {k: z[k] for k in z.files}
    import joblib
    return joblib.load(path)


EVAL_OPTS = dict(resize_factor=1.0, color_bkgd_aug="white", frame_sample_rate=1, test_on_canonical_pose=False,
                 subject="subject", contact_render_type="default", n_bones=20, width=1080, height=1080)


class TestDataset(torch.utils.data.Dataset):
    """The reference's `TestDataset` (brics_dynamic.py:485-696): no images, one item per camera of a camera path, each
    paired with a frame of a skeleton trajectory -- what the evaluation / rendering entry points iterate.

    opts: `cam_path` (dict of `intrs` (n,4) fx fy cx cy and `extrs` (n,3,4)), `metadata_path` (skeleton table in armature
    space incl. `rest_matrix_world` / `pose_matrix_world`), `cano_cam_path` (one camera), `frame_sample_rate`,
    `test_on_canonical_pose`, `contact_render_type` ("gt_eval": the last 250 `frame_nums`; "acc_gt_eval": every frame;
    anything else: the camera path thinned by `frame_sample_rate`, skeleton frames stepped to cover it), `color_bkgd_aug`,
    `subject`.  Paths are used as given (the reference resolves them against its own checkout).  The image size is
    1080 x 1080 like the reference's."""
    __test__ = False   # (not a pytest class)

    def __init__(self, opts, split="train"):
        o = dict(EVAL_OPTS)
        o.update(opts or {})
        self.opts = o
        self.resize_factor, self.bg_color = o["resize_factor"], o["color_bkgd_aug"]
        self.frame_sample_rate, self.test_on_canonical_pose = o["frame_sample_rate"], o["test_on_canonical_pose"]
        self.width, self.height, self.n_bones = o["width"], o["height"], o["n_bones"]
        self.subject_id, self.mode = o["subject"], o["contact_render_type"]
        cam_data = _load_table(o["cam_path"], size=(self.width, self.height) if self.mode == "acc_gt_eval" else None)
        cano = _load_table(o["cano_cam_path"])
        md = convert_armature_space_to_world_space(_load_table(o["metadata_path"]))
        parts = os.path.normpath(o["metadata_path"]).split(os.sep)
        action = parts[-2] if len(parts) > 1 else ""
        cam_names = [str(c) for c in cam_data["cam_name"]] if (self.mode == "acc_gt_eval" and "cam_name" in cam_data) else None
        J = self.n_bones
        bnames = [str(n) for n in np.asarray(md["bnames"]).tolist()]
        self.bones_rest = to_tensor(Bones(bnames=bnames, heads=md["rest_heads"][:J], tails=md["rest_tails"][:J],
                                          transforms=md["rest_matrixs"][:J]))
        Ks, extrs = cam_data["intrs"], cam_data["extrs"]
        n_pose = md["pose_tails"].shape[0]
        if self.mode == "gt_eval":
            frame_ids = selected = np.asarray(md["frame_nums"][-250:])
            self.n_frames = len(selected)
        elif self.mode == "acc_gt_eval":
            self.n_frames = len(extrs)
            frame_ids = selected = np.arange(n_pose)
        else:
            self.n_frames = len(extrs[:: self.frame_sample_rate])
            step = 1 if n_pose < self.n_frames else -(-n_pose // self.n_frames)
            frame_ids = selected = np.arange(0, n_pose, step)
        heads, tails, mats = md["pose_heads"][selected], md["pose_tails"][selected], md["pose_matrixs"][selected]
        value = to_tensor(np.concatenate([np.asarray(md["root_rotation"])[:, None, :], np.asarray(md["eulers"])], 1))
        quats = euler_angles_to_quats(value)
        pose_latent = quats.reshape(-1, quats.shape[1] * quats.shape[2])[selected]
        self.infos, self.bones_posed_list, self.pose_latent_list = [], [], []
        cols = {}
        for i in range(self.n_frames):
            idx = min(i, tails.shape[0] - 1)      # a path longer than the trajectory holds the last pose
            self.infos.append([self.subject_id, action, str(frame_ids[idx]), cam_names[i] if cam_names is not None else str(i)])
            self.pose_latent_list.append(pose_latent[idx])
            if self.test_on_canonical_pose:
                self.bones_posed_list.append(self.bones_rest)
            else:
                self.bones_posed_list.append(to_tensor(Bones(bnames=bnames, heads=heads[idx, :J], tails=tails[idx, :J],
                                                             transforms=mats[idx, :J])))
            fx, fy, cx, cy = Ks[i]
            K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1]])
            attr = get_opengl_camera_attributes(K, np.asarray(extrs[i]), self.width, self.height, resize_factor=1.0)
            for k, v in attr.items():
                cols.setdefault(k, []).append(v)
            cols.setdefault("cam_name", []).append(str(i))
        self.all_cameras = to_tensor(Cameras(**{k: np.stack(v, 0) for k, v in cols.items()}))
        cK = cano["intrs"][0]
        cattr = get_opengl_camera_attributes(np.array([[cK[0], 0, cK[2]], [0, cK[1], cK[3]], [0, 0, 1]]),
                                             np.asarray(cano["extrs"][0]), self.width, self.height)
        cc = {k: np.stack([v], 0) for k, v in cattr.items()}
        cc["cam_name"] = np.stack(["0"], 0)
        self.cano_camera = to_tensor(Cameras(**cc))

    def __len__(self):
        return self.n_frames

    def __getitem__(self, index):
        return self.fetch_data(index)

    def fetch_data(self, index):
        return {"idx": index, "info": self.infos[index], "camera": self.all_cameras[index], "cano_camera": self.cano_camera,
                "scaling_modifier": 1.0, "bones_rest": self.bones_rest, "bones_posed": self.bones_posed_list[index],
                "pose_latent": self.pose_latent_list[index],
                "bg_color": torch.tensor([1.0, 1.0, 1.0]) if self.bg_color == "white" else torch.tensor([0.0, 0.0, 0.0])}

    def view_batch(self, indices, device="cpu"):
        """Items -> the cameras / poses / keypoints `hand_scene_from_batch` and `engine.HipViewCompute` consume (no
        targets or masks: this dataset holds no images)."""
        out = _skeleton_batch([self.fetch_data(i) for i in indices], device)
        out["width"], out["height"] = self.width, self.height
        return out


def hand_scene_from_batch(batch, bones_rest, n_gaussians, grid_res=32, seed=0, device="cuda:0", sigma_range=(2e-3, 6e-3),
                          voxel_grid=None):
    """The scene dict `engine.HipViewCompute` / `engine.Trainer` consume, from one `SequenceDataset.view_batch` and the
    capture's rest skeleton (`ds[i]["bones_rest"]`): Gaussians initialised on the bones like `sample_gaussians_on_bones_func`
    (train_utils.py:104-139), a skin-weight voxel grid with `build_voxel_grid`'s geometry (brics_dynamic.py:99-144; the
    weights themselves are the synthetic softmax of `synthetic.make_skin_grid`, MANO's nearest-vertex weights are not
    built here), the per-view bone transforms posed @ inv(rest) + identity (hand_dynamic.py:93-102), the dataset's own
    targets, masks, cameras and keypoints.  `voxel_grid` = the tuple `build_voxel_grid` returns (scale, center, grid points,
    weights (D,H,W,21), mask): the MANO-initialised grid takes the place of the synthetic one, like hand_dynamic.py:43-58 hands
    it to the model.  Returns (scene, targets (V,3,H,W))."""
    import math
    from . import synthetic as S
    gen = torch.Generator().manual_seed(seed)
    heads, tails = bones_rest.heads.cpu().numpy(), bones_rest.tails.cpu().numpy()
    rest = bones_rest.transforms.cpu().float()
    per = max(2, int(math.ceil(n_gaussians / 30.0)))
    xyz = S.sample_on_bones(heads, tails, rest.numpy(), per, gen)
    xyz = xyz[torch.randperm(xyz.shape[0], generator=gen)[:n_gaussians]]
    if voxel_grid is not None:
        g_scale, g_center, _, g_weights, _ = voxel_grid
        dims, center, scale = tuple(g_weights.shape[:3]), g_center.reshape(3).numpy(), g_scale.reshape(3).numpy()
    else:
        dims, center, scale = S.grid_geometry(heads, tails, res=grid_res)
    lo, hi = torch.tensor(center - 0.95 * scale), torch.tensor(center + 0.95 * scale)
    xyz = torch.max(torch.min(xyz, hi), lo).float()
    N = xyz.shape[0]
    lo_s, hi_s = math.log(sigma_range[0]), math.log(sigma_range[1])
    params = {"_xyz": xyz, "_scaling": (torch.rand((N, 3), generator=gen) * (hi_s - lo_s) + lo_s).float(),
              "_rotation": torch.randn((N, 4), generator=gen), "_opacity": 1.5 * torch.randn((N, 1), generator=gen),
              "_features_dc": torch.randn((N, 1, 3), generator=gen), "_features_rest": 0.1 * torch.randn((N, 15, 3), generator=gen)}
    posed = batch["posed"].cpu().float()
    tf = torch.stack([T.bone_transforms(posed[v], rest) for v in range(posed.shape[0])])
    cams = [{k: (v.cpu().numpy() if torch.is_tensor(v) else v) for k, v in c.items()} for c in batch["cameras"]]
    has_img = "targets" in batch
    H, W = batch["targets"].shape[-2:] if has_img else (batch["height"], batch["width"])
    scene = dict(params={k: v.to(device) for k, v in params.items()}, N=N, n_hand=N, kind="hand", grid_dims=dims,
                 grid=(voxel_grid[3].to(device).contiguous() if voxel_grid is not None
                       else S.make_skin_grid(heads, tails, dims, center, scale, device=device)),
                 grid_center=torch.as_tensor(center).to(device), grid_scale=torch.as_tensor(scale).to(device),
                 rest=rest.to(device), posed=posed.to(device), transforms=tf.to(device), cameras=cams,
                 bg=torch.ones(3, device=device), width=int(W), height=int(H), heads=heads, tails=tails,
                 keypoints=batch["keypoints"].float().to(device))
    if not has_img:                       # evaluation trajectory: nothing to compare against
        return scene, None
    scene["masks"] = batch["masks"].to(device)
    return scene, batch["targets"].float().to(device).contiguous()


# ---------------------------------------------------------------------------------------------------------------------
# synthetic sequences in the capture schema (tests, examples)
# ---------------------------------------------------------------------------------------------------------------------
def synthetic_sequence(seed, n_frames=3, n_cams=4, width=64, height=48, n_bones=20, frame_start=8):
    """{path: array} of one action file: a random 20-bone skeleton posed per frame by forward kinematics, `n_cams`
    cameras on a ring, RGBA crops with an elliptical alpha mask."""
    rng = np.random.default_rng(seed)
    arr = {}
    names = ["bone_%d" % i for i in range(n_bones)]
    parents = ["None" if i % 4 == 0 else "bone_%d" % (i - 1) for i in range(n_bones)]
    rest = np.tile(np.eye(4, dtype=np.float32), (n_bones, 1, 1))
    rest[:, :3, 3] = rng.normal(0, 0.05, (n_bones, 3)).astype(np.float32)
    rest[:, :3, :3] = T.euler_angles_to_matrix(torch.tensor(rng.normal(0, 0.4, (n_bones, 3)), dtype=torch.float32), "XYZ").numpy()
    heads = rest[:, :3, 3].copy()
    tails = heads + rest[:, :3, 1] * 0.03
    kintree = T.build_kintree(names, parents)
    for f in range(n_frames):
        fno = str(frame_start + 3 * f)
        eul = rng.normal(0, 0.3, (n_bones, 3)).astype(np.float32)
        r_R = rng.normal(0, 0.2, 3).astype(np.float32)
        r_T = rng.normal(0, 0.02, 3).astype(np.float32)
        pose = T.get_pose_wrt_root(torch.tensor(rest), T.euler_angles_to_matrix(torch.tensor(eul), "XYZ", intrinsic=True)[None],
                                   T.euler_angles_to_matrix(torch.tensor(r_R), "XYZ", intrinsic=True)[None],
                                   torch.tensor(r_T)[None], kintree)[0].numpy()
        ph = pose[:, :3, 3].copy()
        md = {"bnames": np.array(names, dtype="S")[:, None], "bnames_parent": np.array(parents, dtype="S")[:, None],
              "rest_heads": heads, "rest_tails": tails, "rest_matrixs": rest, "pose_heads": ph,
              "pose_tails": ph + pose[:, :3, 1] * 0.03, "pose_matrixs": pose, "eulers": eul, "root_translation": r_T,
              "root_rotation": r_R}
        for k, v in md.items():
            arr["frames/%s/metadata/%s" % (fno, k)] = v
        for c in range(n_cams):
            cam = "cam%02d" % c
            w, h = int(rng.integers(width // 4, width // 2)), int(rng.integers(height // 4, height // 2))
            x0, y0 = int(rng.integers(0, width - w)), int(rng.integers(0, height - h))
            yy, xx = np.mgrid[0:h, 0:w]
            inside = ((xx - w / 2) / (w / 2)) ** 2 + ((yy - h / 2) / (h / 2)) ** 2 <= 1.0
            crop = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
            crop[..., 3] = np.where(inside, 255, rng.integers(0, 2, (h, w)) * 40).astype(np.uint8)
            arr["frames/%s/images/%s" % (fno, cam)] = crop
            arr["frames/%s/bbox/%s" % (fno, cam)] = np.array([x0, y0, x0 + w, y0 + h], np.int64)
    for c in range(n_cams):
        cam = "cam%02d" % c
        a = 2 * np.pi * c / n_cams
        pos = np.array([0.6 * np.cos(a), 0.6 * np.sin(a), 0.25])
        z = -pos / np.linalg.norm(pos)
        x = np.cross(z, [0, 0, 1.0]); x /= np.linalg.norm(x)
        y = np.cross(z, x)
        R = np.stack([x, y, z], 0)
        arr["K/" + cam] = np.array([[90.0 + c, 0, width / 2 - 0.5], [0, 90.0 + c, height / 2 - 0.5], [0, 0, 1]], np.float64)
        arr["extr/" + cam] = np.concatenate([R, (-R @ pos)[:, None]], 1).astype(np.float64)
    arr["mano_rest/verts"] = rng.normal(0, 0.05, (30, 3)).astype(np.float32)
    arr["mano_rest/weights"] = rng.random((30, 16)).astype(np.float32)
    return arr
