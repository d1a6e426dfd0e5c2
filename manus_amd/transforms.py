"""Skeleton plumbing of the hot path (reference src/utils/transforms.py): Euler ->
matrix, kinematic-tree FK, kintree construction.  Host-side torch (tiny 20-bone
tensors; the reference also runs these on the host/offline)."""
import torch


_AXIS_ROWS = {   # rotation about one axis as (row, col) -> +cos / -sin / +sin / 1 / 0
    "X": ("1", "0", "0", "0", "c", "-s", "0", "s", "c"),
    "Y": ("c", "0", "s", "0", "1", "0", "-s", "0", "c"),
    "Z": ("c", "-s", "0", "s", "c", "0", "0", "0", "1"),
}


def _axis_angle_rotation(axis, angle):
    """Rotation matrices (..., 3, 3) about one coordinate axis (transforms.py:533-558)."""
    if axis not in _AXIS_ROWS:
        raise ValueError("axis must be X, Y or Z, got %r" % (axis,))
    c, s = torch.cos(angle), torch.sin(angle)
    entry = {"c": c, "s": s, "-s": -s, "1": torch.ones_like(angle), "0": torch.zeros_like(angle)}
    return torch.stack([entry[e] for e in _AXIS_ROWS[axis]], -1).reshape(angle.shape + (3, 3))


def euler_angles_to_matrix(euler_angles, convention, intrinsic=False):
    """Euler angles (..., 3) in radians -> rotation matrices (..., 3, 3), one axis letter per angle
    (transforms.py:489-530).  intrinsic=True: the same rotation read as intrinsic, i.e. the reversed axis order applied
    to the reversed angles."""
    if euler_angles.dim() == 0 or euler_angles.shape[-1] != 3:
        raise ValueError("euler_angles must have a last dimension of 3, got shape %s" % (tuple(euler_angles.shape),))
    axes = str(convention)
    if len(axes) != 3 or any(a not in _AXIS_ROWS for a in axes):
        raise ValueError("convention must be three letters out of X, Y, Z, got %r" % (convention,))
    if axes[1] == axes[0] or axes[1] == axes[2]:
        raise ValueError("convention %r repeats an axis in neighbouring positions" % (convention,))
    if intrinsic:
        axes, euler_angles = axes[::-1], euler_angles.flip(-1)
    r0, r1, r2 = (_axis_angle_rotation(a, e) for a, e in zip(axes, euler_angles.unbind(-1)))
    return r0 @ r1 @ r2


def build_kintree(bnames, bnames_parent):
    """transforms.py:609-623: {str(i): parent index or -1}."""
    bnames = list(bnames)
    bnames_parent = list(bnames_parent)
    tree = {}
    for i, _ in enumerate(bnames):
        p = bnames_parent[i]
        tree[str(i)] = bnames.index(p) if (p is not None and p != "None") else -1
    return tree


def get_pose_wrt_root(rest_pose, pose_param, global_pose, global_t, kintree):
    """transforms.py:233-261.  rest_pose (J,4,4); pose_param (B,J,3,3);
    global_pose (B,3,3); global_t (B,3); kintree dict -> (B,J,4,4)."""
    B, J = pose_param.shape[0], pose_param.shape[1]
    pose = torch.zeros(B, J, 4, 4, dtype=rest_pose.dtype, device=rest_pose.device)
    pose[:, :, :3, :3] = pose_param
    pose[:, :, 3, 3] = 1.0
    G = torch.zeros(B, 4, 4, dtype=rest_pose.dtype, device=rest_pose.device)
    G[:, :3, :3] = global_pose
    G[:, :3, 3] = global_t
    G[:, 3, 3] = 1.0
    M = [None] * J
    for i in range(J):
        if kintree[str(i)] == -1:
            M[i] = G @ rest_pose[i][None] @ pose[:, i]
    for i in range(J):
        p = kintree[str(i)]
        if p == -1:
            continue
        local = torch.linalg.inv(rest_pose[p]) @ rest_pose[i]
        M[i] = M[p] @ (local[None] @ pose[:, i])
    return torch.stack(M, dim=1)


def euler_angles_to_armature_space(pose, kintree, rest_matrixs, global_T):
    """transforms.py:593-606: pose (B,1+J,3) Euler XYZ intrinsic, entry 0 = global."""
    m = euler_angles_to_matrix(pose, "XYZ", intrinsic=True)
    return get_pose_wrt_root(rest_matrixs, m[:, 1:], m[:, 0], global_T, kintree)


def bone_transforms(posed_transforms, rest_transforms, background=True):
    """T_b = posed_b @ inv(rest_b) (+ identity background), hand_dynamic.py:93-102."""
    pt, rt = posed_transforms, rest_transforms
    if pt.is_cuda and rt.is_cuda and pt.dtype == rt.dtype == torch.float32 and pt.dim() == 3 and pt.shape == rt.shape \
            and pt.shape[1:] == (4, 4) and not (pt.requires_grad or rt.requires_grad):
        # on the device (where the reference's batch lives): one launch instead of the ~12 of linalg.inv + einsum + cat
        from ._lib import check, lib, ptr, stream
        B = pt.shape[0]
        out = torch.empty((B + (1 if background else 0), 4, 4), dtype=torch.float32, device=pt.device)
        check(lib().mgr_bone_transforms(B, int(bool(background)), ptr(pt.contiguous()), ptr(rt.contiguous()), ptr(out), stream()),
              "mgr_bone_transforms")
        return out
    T = torch.einsum("nij,njk->nik", posed_transforms, torch.linalg.inv(rest_transforms))
    if background:
        T = torch.cat([T, torch.eye(4, dtype=T.dtype, device=T.device)[None]], dim=0)
    return T
