"""MANUS checkpoint wire format: read and write the Lightning `.ckpt` files the reference
trains to and resumes from, so that real MANUS checkpoints drop into this package and vice versa.

On-disk contract (a `torch.save`d dict):

    state_dict["model._xyz" | "model._features_dc" | "model._features_rest" | "model._scaling" |
               "model._rotation" | "model._opacity"]                       src/models/gaussian.py:120-125
    extra_params["num_gaussians"]                                           src/modules/base.py:73-76
    extra_params["grid_scale" | "grid_center" | "grid_points" | "grid_weights"]   (skin_weights_init_type
        "mano_init_voxel") or extra_params["mano_weights"]                  src/modules/hand_dynamic.py:295-315
    file name  epoch={epoch:03d}-step={step}-loss={loss:.6f}.ckpt           main.py:61-69

Mirrors `load_checkpoint` / `remove_nans_from_checkpoint` (src/utils/train_utils.py:165-204: rows with a NaN in
any leaf are dropped from every leaf, `num_gaussians` updated, the "model." prefix stripped) and
`find_best_checkpoint` (src/utils/extra.py:203-242, including its string comparison of epochs).
Host-side logic only: tensors stay wherever `map_location` puts them.
"""
import glob
import os

import numpy as np
import torch

LEAF_KEYS = ("model._xyz", "model._features_dc", "model._features_rest", "model._scaling", "model._rotation",
             "model._opacity")
GRID_KEYS = ("grid_scale", "grid_center", "grid_points", "grid_weights")


def checkpoint_name(epoch, step, loss):
    """The reference's ModelCheckpoint file name (main.py:61-69, auto_insert_metric_name)."""
    return "epoch=%03d-step=%d-loss=%.6f.ckpt" % (epoch, step, loss)


def remove_nans_from_checkpoint(checkpoint):
    """Drop every Gaussian that has a NaN in any state_dict tensor (train_utils.py:165-190)."""
    sd = checkpoint["state_dict"]
    n = sd["model._xyz"].shape[0]
    nan_mask = torch.zeros((n,), dtype=torch.bool, device=sd["model._xyz"].device)
    for key, value in sd.items():
        if value.dim() < 2 or value.dim() > 4:
            raise ValueError("checkpoint tensor %s has %d dimensions" % (key, value.dim()))
        m = torch.isnan(value).reshape(value.shape[0], -1).any(dim=-1)
        if m.shape[0] != n:
            raise ValueError("checkpoint tensor %s has %d rows, expected %d" % (key, m.shape[0], n))
        nan_mask |= m
    keep = ~nan_mask
    for key in list(sd.keys()):
        sd[key] = sd[key][keep]
    checkpoint.setdefault("extra_params", {})["num_gaussians"] = sd["model._xyz"].shape[0]
    # the optimizer state `save_checkpoint(optimizer=...)` adds is per row too: drop the same rows, so that a resume
    # does not fail (later, in load_state_dict) on a row count that no longer matches the leaves
    opt = checkpoint.get("manus_amd_optimizer")
    if isinstance(opt, dict) and bool(nan_mask.any()):
        def rows(v):
            if torch.is_tensor(v) and v.dim() >= 1 and v.shape[0] == n:
                return v[keep.to(v.device)]
            if isinstance(v, dict):
                return {k: rows(x) for k, x in v.items()}
            if isinstance(v, (list, tuple)):
                return type(v)(rows(x) for x in v)
            return v
        checkpoint["manus_amd_optimizer"] = rows(opt)
    return checkpoint


def load_checkpoint(ckpt_path, device=torch.device("cpu"), allow_pickle=False, return_checkpoint=False):
    """(weights without the "model." prefix, extra_params) of a MANUS checkpoint (train_utils.py:193-204).

    Tensors and plain containers load under torch's safe unpickler.  A real Lightning checkpoint may carry other pickled
    objects (hyper-parameter containers); the reference unpickles those unconditionally (torch.load of a file the user
    trained themselves).  Here that is opt-in: allow_pickle=True -- a file the safe loader rejects is exactly the one that
    can execute code while loading.  return_checkpoint=True also returns the whole dict (optimizer state written by
    `save_checkpoint(optimizer=...)` sits under "manus_amd_optimizer")."""
    import pickle
    try:
        ckpt = torch.load(ckpt_path, map_location=device, weights_only=True)
    except pickle.UnpicklingError as e:     # (a missing or corrupt file raises something else and is not re-labelled)
        if not allow_pickle:
            raise RuntimeError("%s holds pickled objects the safe loader refuses (%s); pass allow_pickle=True if you trust "
                               "the file" % (ckpt_path, e)) from e
        ckpt = torch.load(ckpt_path, map_location=device, weights_only=False)
    ckpt = remove_nans_from_checkpoint(ckpt)
    weights = ckpt["state_dict"]
    for key in list(weights):
        weights[key.replace("model.", "")] = weights.pop(key)
    if return_checkpoint:
        return weights, ckpt.get("extra_params", {}), ckpt
    return weights, ckpt.get("extra_params", {})


def get_num_gaussians_from_checkpoint(ckpt_path, allow_pickle=False):
    return load_checkpoint(ckpt_path, allow_pickle=allow_pickle)[1]["num_gaussians"]


def save_checkpoint(ckpt_dir, params, epoch, step, loss, grid=None, mano_weights=None, extra_state=None, optimizer=None):
    """Write a checkpoint the reference's `load_checkpoint` / `on_load_checkpoint` read back (train_utils.py:193-204,
    hand_dynamic.py:284-293): epoch, global_step, state_dict, extra_params.  (It is not a full Lightning
    `fit(ckpt_path=...)` resume file: no optimizer_states / loops keys.)  params: the six leaves by attribute name (`_xyz`, ...);
    grid: dict with grid_scale / grid_center / grid_points / grid_weights (voxel skin weights) or
    mano_weights: (N,B) per-Gaussian skin weights.  Returns the path."""
    sd = {}
    for key in LEAF_KEYS:
        sd[key] = params[key[len("model."):]].detach().cpu()
    extra = {"num_gaussians": sd["model._xyz"].shape[0]}
    if grid is not None:
        for k in GRID_KEYS:
            extra[k] = grid[k]
    elif mano_weights is not None:
        extra["mano_weights"] = mano_weights.detach().cpu()
    ckpt = {"epoch": int(epoch), "global_step": int(step), "state_dict": sd, "extra_params": extra}
    if optimizer is not None:
        # what a resume needs beyond the leaves: both Adam moments, the per-group step counts and the densification
        # statistics (`GaussianOptimizer.state_dict`); a resume without them restarts the moments from zero
        ckpt["manus_amd_optimizer"] = optimizer.state_dict()
    if extra_state:
        ckpt.update(extra_state)
    os.makedirs(ckpt_dir, exist_ok=True)
    path = os.path.join(ckpt_dir, checkpoint_name(epoch, step, loss))
    torch.save(ckpt, path)
    return path


def find_best_checkpoint(check_dir, sort_by="epoch"):
    """extra.py:203-242.  sort_by="epoch" takes max() over the epoch *strings* (the zero-padded :03d format makes
    that numeric up to epoch 999) and returns the first matching file in glob order; "loss" takes the smallest
    loss, the largest step string among equal losses."""
    all_checkpoints = glob.glob(os.path.join(check_dir, "*.ckpt"))
    if len(all_checkpoints) == 0:
        raise FileNotFoundError("no checkpoint found at %s" % check_dir)
    epochs, steps, loss_strs, losses = [], [], [], []
    for p in all_checkpoints:
        name = p.split("/")[-1]
        epochs.append(name.split("epoch=")[-1].split("-")[0])
        steps.append(name.split("step=")[-1].split("-")[0])
        ls = name.split(".ckpt")[0].split("=")[-1]
        loss_strs.append(ls)
        losses.append(np.array(ls).astype(np.float64))
    if sort_by == "loss":
        min_idx = int(np.argmin(losses))
        min_loss = min(losses)
        mask = np.array(losses, dtype=np.float64) == min_loss
        steps_array = np.array(steps)
        max_step = max(steps_array[mask])
        idx = steps_array.tolist().index(max_step)
        return os.path.join(check_dir, "epoch=%s-step=%s-loss=%s.ckpt" % (epochs[idx], steps[idx], loss_strs[min_idx]))
    if sort_by == "epoch":
        max_epoch = max(epochs)
        for p in all_checkpoints:
            if "epoch=%s" % max_epoch in p:
                return p
    return None
