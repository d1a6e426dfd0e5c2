"""Boundary dataclasses with the reference's field names (src/utils/structures.py:7-47)."""
from dataclasses import dataclass
from typing import Any


@dataclass
class Bones:
    bnames: Any
    heads: Any
    tails: Any
    transforms: Any
    eulers: Any = None
    eulers_c: Any = None
    root_translation: Any = None
    root_rotation: Any = None
    kintree: Any = None

    def __getitem__(self, idx):
        return Bones(**{k: (v[idx] if v is not None else None) for k, v in self.__dict__.items()})


@dataclass
class Cameras:
    cam_name: Any
    K: Any
    extr: Any
    fovx: Any
    fovy: Any
    width: Any
    height: Any
    world_view_transform: Any
    projection_matrix: Any
    full_proj_transform: Any
    camera_center: Any

    def __getitem__(self, idx):
        return Cameras(**{k: (v[idx] if v is not None else None) for k, v in self.__dict__.items()})
