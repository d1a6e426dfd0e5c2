"""Skeleton plumbing of the hot path (reference src/utils/transforms.py): Euler ->
matrix, kinematic-tree FK, kintree construction.  Host-side torch (tiny 20-bone
tensors; the reference also runs these on the host/offline)."""
import torch


def _axis_angle_rotation(axis, angle):
    """transforms.py:533-558."""
    c, s = torch.cos(angle), torch.sin(angle)
    o, z = torch.ones_like(angle), torch.zeros_like(angle)
    if axis == "X":
        flat = (o, z, z, z, c, -s, z, s, c)
    elif axis == "Y":
        flat = (c, z, s, z, o, z, -s, z, c)
    elif axis == "Z":
        flat = (c, -s, z, s, c, z, z, z, o)
    else:
        raise ValueError("letter must be either X, Y or Z.")
    return torch.stack(flat, -1).reshape(angle.shape + (3, 3))


def euler_angles_to_matrix(euler_angles, convention, intrinsic=False):
    """transforms.py:489-530 (intrinsic = reversed convention on flipped angles)."""
    if intrinsic:
        convention = convention[::-1]
        euler_angles = euler_angles.flip(-1)
    if euler_angles.dim() == 0 or euler_angles.shape[-1] != 3:
        raise ValueError("Invalid input euler angles.")
    if len(convention) != 3:
        raise ValueError("Convention must have 3 letters.")
    if convention[1] in (convention[0], convention[2]):
        raise ValueError(f"Invalid convention {convention}.")
    for letter in convention:
        if letter not in ("X", "Y", "Z"):
            raise ValueError(f"Invalid letter {letter} in convention string.")
    m = [_axis_angle_rotation(c, e) for c, e in zip(convention, torch.unbind(euler_angles, -1))]
    return torch.matmul(torch.matmul(m[0], m[1]), m[2])


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
    T = torch.einsum("nij,njk->nik", posed_transforms, torch.linalg.inv(rest_transforms))
    if background:
        T = torch.cat([T, torch.eye(4, dtype=T.dtype, device=T.device)[None]], dim=0)
    return T
