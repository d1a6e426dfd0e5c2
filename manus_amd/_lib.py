"""ctypes binding of libmanus_hip.so (the C ABI declared in include/manus_hip.h).

The library is mandatory for every compute op of this package: there is NO
CPU or PyTorch fallback.  `lib()` raises if the .so has not been built
(`python -m manus_amd.build`) and the ops raise if their tensors are not on a
GPU.
"""
import ctypes
import os

import weakref

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_VARIANT = os.environ.get("MANUS_HIP_VARIANT", "")   # instrumented builds (tools/instr), never set in production
LIB_PATH = os.path.join(_HERE, "libmanus_hip%s.so" % ("_" + _VARIANT if _VARIANT else ""))
_LIB = None

MGR_CAM_FLOATS = 40
MGR_MAX_BONES = 32

c_int, c_i64, c_f32, c_vp, c_sz = ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t

# name -> (restype, argtypes); must list every symbol of include/manus_hip.h
SIGNATURES = {
    "mgr_version": (c_int, []),
    "mgr_build_variant": (c_int, []),
    "mgr_last_error": (ctypes.c_char_p, []),
    "mgr_raster_workspace_bytes": (c_sz, [c_int, c_int, c_int, c_int, c_i64]),
    "mgr_raster_forward": (c_int, [c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp,
                                   c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_sz, c_i64, c_int, c_vp]),
    "mgr_raster_backward": (c_int, [c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp,
                                    c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz,
                                    c_i64, c_int, c_vp]),
    "mgr_sh_to_half": (c_int, [c_int, c_vp, c_vp, c_vp]),
    "mgr_views_forward": (c_int, [c_int] * 7 + [c_vp] * 12 + [c_vp, c_sz, c_i64, c_int, c_vp]),
    "mgr_views_backward": (c_int, [c_int] * 7 + [c_vp] * 13 + [c_f32] + [c_vp] * 10 + [c_vp, c_sz, c_i64, c_int, c_vp]),
    "mgr_raster_record_bytes": (c_int, []),
    "mgr_raster_layout": (c_int, [c_int, c_int, c_int, c_int, c_i64, ctypes.POINTER(c_sz), c_int]),
    "mgr_raster_status_sync": (c_int, [c_vp, ctypes.POINTER(c_i64), ctypes.POINTER(ctypes.c_int32), c_vp]),
    "mgr_raster_set_status_mirror": (c_int, [c_vp, c_vp]),
    "mgr_raster_set_cut_margin": (c_int, [c_f32, c_int, c_f32, c_f32, c_int]),
    "mgr_raster_set_cut_penalty": (c_int, [c_int]),
    "mgr_raster_status_tiers_sync": (c_int, [c_vp, ctypes.POINTER(c_i64), ctypes.POINTER(ctypes.c_int32),
                                             ctypes.POINTER(ctypes.c_int32), c_vp]),
    "mgr_debug_pair_alpha": (c_int, [c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "mgr_raster_debug_binning_sync": (c_int, [c_vp, c_int, c_int, c_int, c_int, c_i64, c_int, c_vp, c_vp,
                                              c_i64, c_vp]),
    "mgr_skin_weights_fwd": (c_int, [c_int, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp,
                                     c_vp]),
    "mgr_skin_weights_bwd": (c_int, [c_int, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp,
                                     c_vp, c_int, c_vp]),
    "mgr_skin_weights_bwd_indexed": (c_int, [c_int, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp,
                                             c_vp, c_vp, c_vp, c_int, c_vp]),
    "mgr_views_backward_run_lists": (c_int, [c_int]),
    "mgr_views_active_list": (c_int, [c_vp, c_int, c_int, c_int, c_int, c_i64, ctypes.POINTER(c_vp), ctypes.POINTER(c_vp)]),
    "mgr_lbs_cov_fwd": (c_int, [c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "mgr_lbs_cov_bwd": (c_int, [c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                c_vp, c_vp, c_vp, c_vp]),
    "mgr_lbs_cov_fwd_rows": (c_int, [c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_vp]),
    "mgr_lbs_cov_bwd_rows": (c_int, [c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_vp,
                                     c_vp, c_vp, c_vp, c_vp]),
    "mgr_sh_color_fwd_rows": (c_int, [c_int, c_int, c_vp, c_vp, c_i64, c_vp, c_i64, c_int, c_vp, c_vp, c_vp]),
    "mgr_sh_color_bwd_rows": (c_int, [c_int, c_int, c_vp, c_vp, c_i64, c_vp, c_i64, c_int, c_vp, c_vp, c_vp, c_vp, c_vp,
                                      c_vp]),
    "mgr_sh_color_fwd": (c_int, [c_int, c_int, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp]),
    "mgr_sh_color_bwd": (c_int, [c_int, c_int, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp,
                                 c_vp]),
    "mgr_pack_camera": (c_int, [c_f32, c_f32, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "mgr_bone_transforms": (c_int, [c_int, c_int, c_vp, c_vp, c_vp, c_vp]),
    "mgr_project_points": (c_int, [c_int, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "mgr_dilate_mask": (c_int, [c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp]),
    "mgr_points_outside_mask": (c_int, [c_int, c_vp, c_vp, c_vp, c_int, c_int, c_vp, c_int, c_vp, c_vp, c_vp]),
    "mgr_keypoint_far_mask": (c_int, [c_int, c_vp, c_int, c_vp, c_f32, c_vp, c_vp]),
    "mgr_knn3_workspace_bytes": (c_sz, [c_int]),
    "mgr_knn3_mean_dist2": (c_int, [c_int, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "mgr_exchange_mask": (c_int, [c_int, c_vp, c_int, ctypes.POINTER(c_i64), ctypes.POINTER(c_int), c_i64, c_vp, c_vp, c_vp, c_vp]),
    "mgr_exchange_index_workspace_bytes": (c_sz, [c_int]),
    "mgr_exchange_index": (c_int, [c_int, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "mgr_exchange_pack": (c_int, [c_int, c_int, c_vp, c_vp, c_int, ctypes.POINTER(c_i64), ctypes.POINTER(c_int), c_i64, c_vp, c_vp]),
    "mgr_exchange_unpack": (c_int, [c_int, c_int, c_vp, c_vp, c_int, ctypes.POINTER(c_i64), ctypes.POINTER(c_int), c_i64, c_vp, c_vp,
                                    c_i64, c_vp]),
    "mgr_exchange_pack_rows": (c_int, [c_int, c_int, c_vp, c_vp, c_vp, c_int, ctypes.POINTER(c_i64), ctypes.POINTER(c_int), c_i64, c_vp, c_vp]),
    "mgr_exchange_unpack_rows": (c_int, [c_int, c_int, c_vp, c_vp, c_vp, c_int, ctypes.POINTER(c_i64), ctypes.POINTER(c_int), c_i64, c_vp, c_vp,
                                         c_i64, c_vp]),
    "mgr_l1_loss_grad": (c_int, [c_i64, c_vp, c_vp, c_f32, c_vp, c_vp, c_vp]),
    "mgr_image_loss_workspace_bytes": (c_sz, [c_int, c_int, c_int]),
    "mgr_image_loss": (c_int, [c_int, c_int, c_int, c_vp, c_vp, c_f32, c_f32, c_f32, c_f32, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "mgr_image_loss_tiles": (c_int, [c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_f32, c_f32, c_f32, c_f32, c_vp, c_vp, c_vp, c_sz,
                                     c_vp]),
    "mgr_image_loss_target_map_words": (c_sz, [c_int, c_int, c_int]),
    "mgr_image_loss_target_map": (c_int, [c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp]),
    "mgr_image_loss_tiles_list_mapped": (c_int, [c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_sz, c_int, c_vp]),
    "mgr_views_forward_attach_loss_list": (c_int, [c_vp, c_int, c_int, c_int, c_vp, c_vp, c_sz]),
    "mgr_image_loss_tiles_list": (c_int, [c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "mgr_image_loss_tiles_finish": (c_int, [c_int, c_int, c_int, c_vp, c_vp, c_f32, c_f32, c_f32, c_f32, c_vp, c_vp, c_vp, c_sz,
                                            c_vp]),
    "mgr_adam_step": (c_int, [c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, ctypes.c_double, ctypes.c_double,
                               ctypes.c_double, c_vp]),
    "mgr_reset_opacity": (c_int, [c_int, c_vp, c_vp, c_vp, c_vp]),
    "mgr_add_densification_stats": (c_int, [c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "mgr_densify_workspace_bytes": (c_sz, [c_int]),
    "mgr_densify_plan": (c_int, [c_int, c_vp, c_vp, c_vp, c_vp, c_f32, c_f32, c_f32, c_f32, c_f32, c_vp, c_sz, c_vp, c_vp]),
    "mgr_adam_step_groups": (c_int, [c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, ctypes.c_double, ctypes.c_double,
                                      ctypes.c_double, c_vp]),
    "mgr_prune_plan": (c_int, [c_int, c_vp, c_vp, c_sz, c_vp, c_vp]),
    "mgr_gather_rows": (c_int, [c_int, c_i64, c_vp, c_vp, c_vp, c_int, c_vp]),
    "mgr_densify_apply": (c_int, [c_int, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int,
                                   c_vp, c_vp]),
    "mgr_isotropic_reg_workspace_bytes": (c_sz, [c_int]),
    "mgr_isotropic_reg": (c_int, [c_int, c_vp, c_f32, c_f32, c_vp, c_int, c_vp, c_vp, c_sz, c_vp]),
    "mgr_contact_workspace_bytes": (c_sz, [c_int, c_int]),
    "mgr_contact_dist": (c_int, [c_int, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "mgr_knn_mean_rows": (c_int, [c_int, c_vp, c_int, c_vp, c_vp, c_int, c_int, c_vp, c_vp, c_vp]),
    "mgr_mesh_sdf": (c_int, [c_int, c_vp, c_int, c_vp, c_int, c_vp, c_vp, c_vp, c_vp]),
    "mgr_profile_enable": (c_int, [c_int]),
    "mgr_profile_filter": (c_int, [ctypes.c_char_p]),
    "mgr_profile_sample_every": (c_int, [c_int]),
    "mgr_profile_report": (c_int, [ctypes.c_char_p, c_sz, c_vp]),
}


def profile_enable(on, only=None, every=1):
    """HIP events around the library's kernel launches; `only`: just the kernel of that name; `every`: only every n-th
    of those launches."""
    check(lib().mgr_profile_filter(only.encode() if only else None), "mgr_profile_filter")
    check(lib().mgr_profile_sample_every(int(every)), "mgr_profile_sample_every")
    check(lib().mgr_profile_enable(int(bool(on))), "mgr_profile_enable")


def profile_report():
    """{kernel name: (launches, total_ms)} since the last report (synchronises)."""
    buf = ctypes.create_string_buffer(1 << 16)
    check(lib().mgr_profile_report(buf, len(buf), stream()), "mgr_profile_report")
    out = {}
    for line in buf.value.decode().splitlines():
        name, cnt, ms = line.split()
        out[name] = (int(cnt), float(ms))
    return out


class ManusHipError(RuntimeError):
    pass


def lib():
    """Load libmanus_hip.so; raise loudly if it is missing (no fallback)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ManusHipError(
                "libmanus_hip.so is not built (%s). Run `python -m manus_amd.build`; "
                "this package has no CPU/PyTorch fallback." % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


def check(code, what=""):
    if code != 0:
        msg = lib().mgr_last_error()
        raise ManusHipError("%s failed (%d): %s" % (what, code, msg.decode() if msg else ""))


def ptr(t):
    """Device pointer of a contiguous fp32/int32 CUDA(HIP) tensor, or None."""
    if t is None:
        return None
    if not t.is_cuda:
        raise ManusHipError("manus_amd ops need GPU tensors (got %s); there is no CPU fallback" % t.device)
    if not t.is_contiguous():
        raise ManusHipError("tensor must be contiguous")
    return t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def stream():
    """The current HIP stream of the current device, as the integer handle the C ABI takes.  Through torch's raw accessors (two C
    calls): `torch.cuda.current_stream().cuda_stream` walks ~15 Python frames per call (device-index helpers, an os.getenv), and
    the operator route asks eight times per step (tools/instr/dropin_host_profile.py)."""
    if _raw_stream is not None and _raw_device is not None:
        return _raw_stream(_raw_device())
    return torch.cuda.current_stream().cuda_stream


def f32c(t):
    """contiguous fp32 view/copy (the reference passes fp32 everywhere)."""
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


_PACKED = {}      # (ids of the source tensors, scalars) -> (weak references, versions, packed table)


def cached_pack(tensors, scalars, build):
    """`build()` once per set of unmodified source TENSOR OBJECTS: a training loop passes the same camera tensors step after
    step (the reference keeps them on its Camera objects), and packing them again is a handful of launches per forward on a
    route that is bound by launches.  Valid while every source is the same object (weak reference) at the same version
    (in-place writes bump it); anything else -- new tensors, non-tensors -- builds afresh."""
    if not all(torch.is_tensor(t) for t in tensors):
        return build()
    key = tuple(id(t) for t in tensors) + tuple(scalars)
    ent = _PACKED.get(key)
    if ent is not None and all(r() is t and t._version == ver for r, t, ver in zip(ent[0], tensors, ent[1])):
        return ent[2]
    out = build()
    if len(_PACKED) >= 1024:
        _PACKED.clear()
    _PACKED[key] = (tuple(weakref.ref(t) for t in tensors), tuple(t._version for t in tensors), out)
    return out


def pack_cameras(tanfovx, tanfovy, viewmatrix, projmatrix, campos, device):
    """See _pack_cameras; the table of one set of camera tensors is built once (cached_pack)."""
    if isinstance(viewmatrix, (list, tuple)):
        srcs = list(viewmatrix) + list(projmatrix) + list(campos)
        scal = [float(x) for x in tanfovx] + [float(x) for x in tanfovy]
    else:
        srcs, scal = [viewmatrix, projmatrix, campos], [float(tanfovx), float(tanfovy)]
    return cached_pack(srcs, scal + [str(device)], lambda: _pack_cameras(tanfovx, tanfovy, viewmatrix, projmatrix, campos, device))


def _pack_cameras(tanfovx, tanfovy, viewmatrix, projmatrix, campos, device):
    """Build the (V, MGR_CAM_FLOATS) device camera table from reference-style
    camera tensors (each may carry a leading batch dim of 1).
    Lists give V > 1.  No host synchronisation."""
    if not isinstance(viewmatrix, (list, tuple)):
        tanfovx, tanfovy, viewmatrix, projmatrix, campos = [tanfovx], [tanfovy], [viewmatrix], [projmatrix], [campos]
    dev = torch.device(device)

    def on_dev(t, n):
        return torch.is_tensor(t) and t.is_cuda and (dev.index is None or t.device.index == dev.index) and t.dtype == torch.float32 \
            and t.numel() >= n

    if dev.type == "cuda" and all(on_dev(v, 16) and on_dev(p, 16) and on_dev(c, 3) for v, p, c in zip(viewmatrix, projmatrix, campos)):
        # tensors already on the device (the reference's batch is): one launch per camera, the two tangents as kernel
        # arguments -- no host-to-device copy, no cat
        out = torch.empty((len(viewmatrix), MGR_CAM_FLOATS), dtype=torch.float32, device=viewmatrix[0].device)
        for k, (tx, ty, vm, pm, cp) in enumerate(zip(tanfovx, tanfovy, viewmatrix, projmatrix, campos)):
            # (.contiguous(): the reference's world_view_transform is a transposed view -- a copy on the device, still no host-to-device one)
            check(lib().mgr_pack_camera(float(tx), float(ty), ptr(vm.contiguous()), ptr(pm.contiguous()), ptr(cp.contiguous()), out[k].data_ptr(), stream()),
                  "mgr_pack_camera")
        return out
    rows = []
    for tx, ty, vm, pm, cp in zip(tanfovx, tanfovy, viewmatrix, projmatrix, campos):
        head = torch.tensor([float(tx), float(ty)], dtype=torch.float32, device=device)
        rows.append(torch.cat([
            head,
            torch.as_tensor(vm, dtype=torch.float32, device=device).reshape(-1)[:16],
            torch.as_tensor(pm, dtype=torch.float32, device=device).reshape(-1)[:16],
            torch.as_tensor(cp, dtype=torch.float32, device=device).reshape(-1)[:3],
            torch.zeros(3, dtype=torch.float32, device=device)]))
    return torch.stack(rows).contiguous()
