"""Forward passes of the MANUS training modules on the HIP kernels.

Mirrors (brown-ivl/manus):
  hand_forward      src/modules/hand_dynamic.py:86-137   TrainingModule.forward
  object_forward    src/modules/object.py:32-41
  composite_forward src/modules/composite.py:50-78
Each takes the reference's model object (anything exposing `_xyz`, `_scaling`,
`_rotation`, `get_features`, `get_opacity`, and for the hand `grid_center`,
`grid_scale`, `grid_weights`) and the reference's batch dict, and returns the same
dict of tensors the reference returns (attribute-accessible).
"""
import torch

from .ops import lbs_cov, skin_weights
from .transforms import bone_transforms


class Pred(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def device_grid(model):
    """The reference re-uploads the (D,H,W,21) grid on every call
    (gaussian_utils.py:169); here it is uploaded once and cached on the model."""
    from .ops import SkinGrid
    g = getattr(model, "_mgr_grid_dev", None)
    dev = model._xyz.device
    if g is None or g.data.device != dev:
        g = SkinGrid(model.grid_weights, dev)
        model._mgr_grid_dev = g
    return g


def hand_forward(model, batch, background_transform=True, full_tf=True):
    cano_xyz = model._xyz
    dev = cano_xyz.device
    posed_t = torch.as_tensor(batch["bones_posed"].transforms, dtype=torch.float32).to(dev)
    rest_t = torch.as_tensor(batch["bones_rest"].transforms, dtype=torch.float32).to(dev)
    T = bone_transforms(posed_t, rest_t, background=background_transform)
    w = skin_weights(cano_xyz, device_grid(model),
                     torch.as_tensor(model.grid_center).to(dev), torch.as_tensor(model.grid_scale).to(dev))
    assert w.shape[-1] == T.shape[0]  # hand_dynamic.py:104
    # full_tf: the reference's (N,4,4) -- written in that layout by the LBS kernel and read in place by the SH operator
    # (render.calculate_colors_from_sh): no cat of the constant row, no slice copy, nothing in between in the backward either
    pxyz, pcov, tf = lbs_cov(cano_xyz, model._scaling, model._rotation, w, T, tf44=full_tf)
    return Pred(posed_xyz=pxyz[0], posed_cov=pcov[0], cano_xyz=cano_xyz, cano_features=model.get_features,
                cano_opacity=model.get_opacity, tf=tf[0], skin_wts=w)


def object_forward(model, batch=None):
    pxyz, pcov, _ = lbs_cov(model._xyz, model._scaling, model._rotation, None, None)
    return Pred(posed_xyz=model._xyz, posed_cov=pcov[0], cano_xyz=model._xyz,
                cano_features=model.get_features, cano_opacity=model.get_opacity)


def composite_forward(hand_model, obj_model, batch):
    """Concatenate hand (skinned) and object (identity tf) Gaussians, composite.py:50-78."""
    h = hand_forward(hand_model, batch)
    o = object_forward(obj_model, batch)
    n_o = o.posed_xyz.shape[0]
    eye = torch.eye(4, dtype=torch.float32, device=h.tf.device)[None].expand(n_o, 4, 4)
    return Pred(posed_xyz=torch.cat([h.posed_xyz, o.posed_xyz]), posed_cov=torch.cat([h.posed_cov, o.posed_cov]),
                cano_xyz=torch.cat([h.cano_xyz, o.cano_xyz]),
                cano_features=torch.cat([h.cano_features, o.cano_features]),
                cano_opacity=torch.cat([h.cano_opacity, o.cano_opacity]), tf=torch.cat([h.tf, eye]),
                n_hand=h.posed_xyz.shape[0])
