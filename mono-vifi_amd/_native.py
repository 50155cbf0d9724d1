"""ctypes binding of the gfx950 hot-path library (C ABI: include/mvf_hotpath.h).

There is NO CPU fallback: if the library is missing or a tensor is not on a HIP device
the call fails loudly.  Build the library with ``python __graft_entry__.py`` (or
``make -C mono-vifi_amd/csrc``); it is kept in-tree at ``mono-vifi_amd/lib/``.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MVF_HOTPATH_LIB: developer knob (kernel variant builds); the product loads the in-tree library
LIB_PATH = os.environ.get("MVF_HOTPATH_LIB") or os.path.join(_HERE, "lib", "libmvf_hotpath.so")
ABI_VERSION = 14

NO_SSIM, AVG_REPROJ, NO_AUTOMASK = 1, 2, 4
MAX_SRC = 4
MAX_UNITS = 8

_vp, _i, _f, _i64 = C.c_void_p, C.c_int, C.c_float, C.c_int64



class UnitDesc(C.Structure):
    """mvf_unit_desc (include/mvf_hotpath.h): one unit of an mvf_units_fwdbwd launch."""
    _fields_ = [
        ("disp", _vp), ("disp_stride", _i64),
        ("tgt", _vp), ("tgt_stride", _i64),
        ("src", _vp * 2), ("src_stride", _i64 * 2),
        ("T", _vp), ("K", _vp), ("inv_K", _vp),
        ("mask_rec", _vp), ("mask_stride", _i64),
        ("noise", _vp), ("disp_mean_partials", _vp), ("ident_in", _vp),
        ("noise_seed", C.c_uint64),
        ("ident_out", _vp), ("loss", _vp), ("stats", _vp),
        ("g_disp_raw", _vp), ("g_stride", _i64),
        ("g_T_raw", _vp), ("argmin", _vp), ("auto_mask", _vp), ("to_opt", _vp), ("idx_xy", _vp),
        ("noise_out", _vp), ("loss_sum", _vp), ("loss_sum_in", _vp),
    ]


class UnitScaleDesc(C.Structure):
    """mvf_unit_scale_desc: one unit of an mvf_units_fwdbwd_scale launch."""
    _fields_ = [
        ("g_disp_raw", _vp), ("in_stride", _i64),
        ("g_T_raw", _vp), ("stats", _vp), ("g_loss", _vp),
        ("g_disp", _vp), ("out_stride", _i64),
        ("g_T", _vp), ("g_sum", _vp),
    ]


class HeadUnitGrad(C.Structure):
    """mvf_head_unit_grad: one hot-path unit whose raw disparity gradient mvf_disp_head_bwd_units reads in place."""
    _fields_ = [
        ("g_disp_raw", _vp), ("raw_stride", _i64),
        ("stats", _vp), ("g_loss", _vp), ("g_sum", _vp),
        ("smoothness", _f), ("first", C.c_int32), ("step", C.c_int32), ("count", C.c_int32),
    ]


class SilogJob(C.Structure):
    """mvf_silog_job: one SI-log loss of an mvf_silog_many_fwd / _bwd launch."""
    _fields_ = [
        ("pred", _vp), ("pred_stride", _i64), ("target", _vp), ("target_stride", _i64),
        ("mask", _vp), ("mask_stride", _i64), ("g_pred", _vp), ("g_target", _vp),
    ]


MAX_SILOG_JOBS = 16

# name -> argument ctypes (return type is always int = hipError_t unless listed in _RESTYPE)
_SIGNATURES = {
    "mvf_abi_version": [],
    "mvf_error_string": [_i],
    "mvf_workspace_floats": [_i, _i, _i],
    "mvf_disp_to_depth_fwd": [_vp, _vp, _vp, _i64, _f, _f, _vp],
    "mvf_disp_to_depth_bwd": [_vp, _vp, _vp, _vp, _i64, _f, _f, _vp],
    "mvf_backproject_fwd": [_vp, _vp, _vp, _i, _i, _i, _vp],
    "mvf_backproject_bwd": [_vp, _vp, _vp, _i, _i, _i, _vp],
    "mvf_project_fwd": [_vp, _vp, _vp, _vp, _i, _i, _i, _f, _vp],
    "mvf_project_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _vp],
    "mvf_grid_sample_fwd": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "mvf_grid_sample_bwd": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "mvf_warp_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _f, _f, _vp],
    "mvf_warp_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _f, _f,
                     _vp],
    "mvf_ssim_fwd": [_vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "mvf_ssim_bwd": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "mvf_reprojection_fwd": [_vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "mvf_reprojection_bwd": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "mvf_smooth_fwd": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "mvf_smooth_bwd": [_vp, _vp, _vp, _vp, _f, _vp, _i, _i, _i, _i, _i, _vp],
    "mvf_photo_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _vp, _vp, _vp, _vp, _vp, _vp, _i,
                      _i, _i, _vp],
    "mvf_photo_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _vp, _vp, _i, _i, _i, _vp],
    "mvf_unit_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _f, _f, _f, _vp, _vp, _vp,
                     _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    # disp,tgt,src**,T,K,invK,noise,mask, S,flags, smooth,min_disp,range,eps, loss,argmin,auto_mask,
    # to_opt,stats,idx_xy,g_disp_raw,g_T_raw,ws, B,H,W, noise_seed,noise_out,mean_partials, stream
    "mvf_unit_fwdbwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _f, _f, _f, _vp, _vp, _vp,
                        _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, C.c_uint64, _vp, _vp, _vp],
    # units*, n_units, S, flags, smooth, min_disp, range, eps, workspace, tickets, B, H, W, stream
    "mvf_units_fwdbwd": [_vp, _i, _i, _i, _f, _f, _f, _f, _vp, _vp, _i, _i, _i, _vp],
    "mvf_units_fwdbwd_scale": [_vp, _i, _f, _i, _i, _i, _i, _vp],
    "mvf_units_workspace_floats": [_i, _i, _i, _i],
    "mvf_units_ticket_ints": [_i, _i],
    "mvf_up2cat_pad_fwd": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "mvf_up2cat_pad_bwd": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "mvf_pad_act_supported": [_i, _i],
    "mvf_pad_act_workspace_floats": [_i, _i, _i, _i],
    "mvf_reflect_pad1_act_fwd": [_vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "mvf_reflect_pad1_act_bwd": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "mvf_up2cat_pad_act_fwd": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "mvf_up2cat_pad_act_workspace_floats": [_i, _i, _i, _i],
    "mvf_up2cat_pad_act_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "mvf_disp_head_fwd": [_vp, _vp, _vp, _vp, _i, _i, _f, _f, _vp],
    "mvf_disp_head_bwd": [_vp, _vp, _vp, _vp, _i64, _f, _f, _vp],
    "mvf_disp_head_bwd_units": [_vp, _vp, _vp, _vp, _i, _i, _f, _f, _vp, _i, _vp],
    "mvf_bias_act_fwd": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "mvf_bias_act_workspace_floats": [_i, _i, _i],
    "mvf_bias_act_bwd": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "mvf_resize_bilinear_fwd": [_vp, _vp, _i, _i, _i, _i, _i, _f, _f, _i, _vp],
    "mvf_resize_bilinear_bwd": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _f, _i, _vp],
    "mvf_resize_bilinear_bwd_workspace_floats": [_i, _i, _i],
    "mvf_upsample_nearest_fwd": [_vp, _vp, _i, _i, _i, _i, _vp],
    "mvf_upsample_nearest_bwd": [_vp, _vp, _i, _i, _i, _i, _vp],
    "mvf_maxpool3s2_fwd": [_vp, _vp, _vp, _i, _i, _i, _vp],
    "mvf_maxpool3s2_bwd": [_vp, _vp, _vp, _i, _i, _i, _vp],
    "mvf_maxpool3s2_bwd_add": [_vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "mvf_bn_tile": [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp],
    "mvf_bn_fold_running": [_vp, _vp, _vp, _vp, _vp, _f, _i, _i, _vp, _vp],
    "mvf_bn_untile": [_vp, _vp, _vp, _i, _i, _vp],
    "mvf_bn_tile_many": [_vp, _i, _i, _i, _vp],
    "mvf_bn_fold_many": [_vp, _i, _i, _vp, _f, _i, _vp],
    "mvf_sum_act_fwd": [_vp, _i, _vp, _i64, _i, _vp],
    "mvf_regroup_fwd": [_vp, _i, _i, _i64, _i, _vp, _vp, _vp, _vp],
    "mvf_regroup_bwd": [_vp, _vp, _i, _i, _i64, _i, _vp, _vp, _vp, _vp],
    "mvf_interleave_fwd": [_vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i64, _vp],
    "mvf_color_jitter_workspace_floats": [_i],
    "mvf_color_jitter": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "mvf_unit_fwdbwd_scale": [_vp, _vp, _vp, _vp, _f, _vp, _vp, _i, _i, _i, _i, _vp],
    "mvf_unit_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _f, _f, _f, _vp,
                     _vp, _vp, _i, _i, _i, _vp],
    "mvf_pose_fwd": [_vp, _vp, _vp, _i, _i, _vp],
    "mvf_pose_bwd": [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp],
    "mvf_flow_warp_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "mvf_flow_warp_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "mvf_flow_warp_workspace_floats": [_i, _i, _i, _i],
    "mvf_flow_warp_bwd_gather": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "mvf_fusion_prep_floats": [_i, _i, _i],
    "mvf_fusion_prep": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "mvf_fusion_level_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "mvf_fusion_level_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "mvf_fusion_bwd_workspace_ints": [_i, _i, _i],
    "mvf_fusion_level_bwd_gather": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "mvf_fusion_lists_level_ints": [_i, _i, _i],
    "mvf_fusion_lists_scratch_ints": [_i, _i, _vp, _vp],
    "mvf_fusion_lists_build": [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp],
    "mvf_fusion_level_bwd_lists": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "mvf_silog_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _vp],
    "mvf_silog_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _vp],
    "mvf_silog_many_workspace_floats": [_i, _i],
    "mvf_silog_many_fwd": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _vp],
    "mvf_silog_many_bwd": [_vp, _i, _vp, _vp, _vp, _i, _i, _f, _vp],
    "mvf_affine_transform_fwd": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "mvf_affine_transform_views_fwd": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "mvf_affine_restore_strided_fwd": [_vp, _i64, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "mvf_reflect_pad1_fwd": [_vp, _vp, _i, _i, _i, _vp],
    "mvf_reflect_pad1_bwd": [_vp, _vp, _i, _i, _i, _vp],
    "mvf_affine_restore_fwd": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "mvf_affine_restore_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "mvf_profile_enable": [_i],
    "mvf_profile_reset": [],
    "mvf_profile_read": [_i, C.POINTER(C.c_double), C.POINTER(C.c_int64)],
    "mvf_profile_read_work": [_i, C.POINTER(C.c_int64)],
    "mvf_profile_read_launches": [_i, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.c_int64],
    "mvf_profile_name": [_i],
}
(PROF_UNIT_FWD, PROF_UNIT_BWD, PROF_PHOTO_FWD, PROF_PHOTO_BWD, PROF_WARP_FWD, PROF_WARP_BWD,
 PROF_UNIT_FWDBWD) = range(7)
PROF_FIRST_GLUE, PROF_COUNT = 7, 36         # ids >= 7: glue kernels, profile level 2, work = algorithmic bytes
TAG_NAMES = {0: "single_frame", 1: "multi_frame", 2: "affine", 3: "single_frame+affine"}
_RESTYPE = {"mvf_error_string": C.c_char_p, "mvf_profile_name": C.c_char_p, "mvf_profile_read_launches": C.c_int64, "mvf_workspace_floats": C.c_size_t,
            "mvf_flow_warp_workspace_floats": C.c_size_t, "mvf_fusion_prep_floats": C.c_size_t, "mvf_fusion_bwd_workspace_ints": C.c_size_t,
            "mvf_silog_many_workspace_floats": C.c_size_t, "mvf_pad_act_workspace_floats": C.c_size_t,
            "mvf_up2cat_pad_act_workspace_floats": C.c_size_t, "mvf_fusion_lists_level_ints": C.c_size_t, "mvf_fusion_lists_scratch_ints": C.c_size_t,
            "mvf_color_jitter_workspace_floats": C.c_size_t, "mvf_bias_act_workspace_floats": C.c_size_t,
            "mvf_units_workspace_floats": C.c_size_t, "mvf_units_ticket_ints": C.c_size_t,
            "mvf_resize_bilinear_bwd_workspace_floats": C.c_size_t}

EXPORTS = tuple(_SIGNATURES)

_lib = None


class NativeLibraryError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle; raises if the HIP library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryError(
            f"{LIB_PATH} not found: the MI355X hot-path library is not built. Run "
            "`python __graft_entry__.py` (or `make -C mono-vifi_amd/csrc`). "
            "There is no CPU fallback.")
    # PyTorch must have loaded ITS HIP runtime first: the library links libamdhip64 by soname,
    # and if it is dlopen'ed before torch the loader binds /opt/rocm's copy and torch then runs
    # on a second runtime (kernel launches fail with hipErrorNoDevice)
    import torch  # noqa: F401
    handle = C.CDLL(LIB_PATH)
    for name, args in _SIGNATURES.items():
        fn = getattr(handle, name)   # AttributeError if the ABI lost a symbol
        fn.argtypes = args
        fn.restype = _RESTYPE.get(name, C.c_int)
    got = handle.mvf_abi_version()
    if got != ABI_VERSION:
        raise NativeLibraryError(f"ABI mismatch: library {got}, binding {ABI_VERSION}")
    _lib = handle
    return _lib


def check(err, what):
    if err != 0:
        msg = lib().mvf_error_string(err)
        raise RuntimeError(f"{what} failed: HIP error {err} ({msg.decode() if msg else '?'})")


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def ptr_array(tensors):
    """Host array of device pointers (const float* const*)."""
    arr = (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
    return C.cast(arr, C.c_void_p), arr   # keep `arr` alive until the call returns


def require_device(*tensors):
    """All tensors on ONE HIP device, and that device is torch's current device: the kernels
    are enqueued on torch.cuda.current_stream() of the current device and take raw pointers,
    so a tensor of another GPU would be accessed across devices, unordered against the work
    that produced it (one process per GPU: Trainer / bench.py call torch.cuda.set_device)."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError(
                "mono_vifi_amd hot-path ops run on a HIP device (MI355X) only; got a "
                f"{t.device} tensor. There is no CPU fallback.")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError(f"hot-path op got tensors on different devices: {dev} and {t.device}")
    if dev is not None:
        import torch
        cur = torch.cuda.current_device()
        if dev.index is not None and dev.index != cur:
            raise RuntimeError(
                f"hot-path op got {dev} tensors while the current device is cuda:{cur}; call "
                "torch.cuda.set_device(local_rank) (or use `with torch.cuda.device(t.device)`) "
                "so that the kernels are enqueued on the stream of the GPU that owns the data")


def profile_read(kernel_id):
    """(total_ms, launches) of one dominant kernel since the last mvf_profile_reset()."""
    ms, n = C.c_double(0.0), C.c_int64(0)
    check(lib().mvf_profile_read(kernel_id, C.byref(ms), C.byref(n)), "profile_read")
    return ms.value, n.value


def profile_read_work(kernel_id):
    """Pixels (images x H x W over all units of every launch) the recorded launches processed."""
    px = C.c_int64(0)
    check(lib().mvf_profile_read_work(kernel_id, C.byref(px)), "profile_read_work")
    return px.value


def profile_read_launches(kernel_id, cap=65536):
    """Per-launch records of one kernel id since the last reset: list of (ms, work, tag)."""
    ms = (C.c_double * cap)()
    work = (C.c_int64 * cap)()
    tag = (C.c_int32 * cap)()
    n = lib().mvf_profile_read_launches(kernel_id, ms, work, tag, cap)
    if n < 0:
        check(int(-n), "profile_read_launches")
    return [(ms[i], work[i], tag[i]) for i in range(n)]


def profile_name(kernel_id):
    return lib().mvf_profile_name(kernel_id).decode()
