"""ctypes binding of ``libvistracker_hip.so`` (the C ABI declared in ``include/vistracker.h``).

The library is the product: every hot-path op of this package goes through it.  There is no
CPU fallback -- if the shared object is missing or a GPU op is requested without a GPU the
call fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VT_LIB_PATH", os.path.join(_HERE, "libvistracker_hip.so"))   # override only for kernel A/B experiments

vp = C.c_void_p
fp = C.c_void_p  # device pointers are passed as integers (tensor.data_ptr())
ci = C.c_int
cl = C.c_long
cf = C.c_float
cd = C.c_double


class VtMaps(C.Structure):
    _fields_ = [("maps", C.c_void_p * 8), ("res", C.c_int * 8), ("proj", C.c_void_p), ("proj_cols", C.c_int),
                ("act_level", C.c_int), ("proj_level", C.c_int), ("force_fp32", C.c_int)]


# name -> (restype, argtypes); kept in one table so tests can check the export list against the header
SIGNATURES = {
    "vt_last_error": (C.c_char_p, []),
    "vt_version": (ci, []),
    "vt_smplh_create": (ci, [C.POINTER(vp), vp, vp, vp, vp, vp, vp, vp]),
    "vt_smplh_destroy": (None, [vp]),
    "vt_smplh_workspace_floats": (cl, [ci]),
    "vt_smplh_bwd_scratch_floats": (cl, [ci]),
    "vt_smplh_forward": (ci, [vp, fp, fp, fp, ci, fp, fp, fp, fp, vp]),
    "vt_smplh_backward": (ci, [vp, fp, fp, ci, fp, fp, fp, fp, fp, fp, fp, fp, vp]),
    "vt_rodrigues_forward": (ci, [fp, ci, fp, vp]),
    "vt_rodrigues_backward": (ci, [fp, ci, fp, fp, vp]),
    "vt_landmarks_create": (ci, [C.POINTER(vp), vp, vp, vp, ci, ci, vp]),
    "vt_landmarks_destroy": (None, [vp]),
    "vt_landmarks_forward": (ci, [vp, fp, ci, fp, vp]),
    "vt_landmarks_backward": (ci, [vp, fp, ci, fp, ci, vp]),
    "vt_mahalanobis": (ci, [fp, ci, ci, ci, ci, fp, fp, fp, fp, cf, vp]),
    "vt_fill_f64": (ci, [fp, cl, cd, vp]),
    "vt_sum_to_term": (ci, [fp, ci, cf, fp, vp]),
    "vt_sifnet_create": (ci, [C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), vp, vp]),
    "vt_sifnet_destroy": (None, [vp]),
    "vt_nchw_to_nhwc": (ci, [fp, ci, ci, ci, ci, fp, vp]),
    "vt_query_forward": (ci, [vp, C.POINTER(VtMaps), fp, fp, fp, ci, ci, fp, fp, fp, fp, fp, vp]),
    "vt_query_backward": (ci, [vp, C.POINTER(VtMaps), fp, fp, fp, ci, ci, fp, fp, fp, fp, fp, fp, vp]),
    "vt_query_human_loss": (ci, [vp, C.POINTER(VtMaps), fp, fp, fp, ci, ci, fp, fp, cf, cf, fp, fp, vp]),
    "vt_query_object_loss": (ci, [vp, C.POINTER(VtMaps), fp, fp, fp, ci, ci, fp, cf, fp, fp, vp]),
    "vt_query_set_human_kernel": (ci, [ci]),
    "vt_sifnet_set_precision": (ci, [vp, ci]),
    "vt_sifnet_get_precision": (ci, [vp]),
    "vt_groupnorm_nhwc": (ci, [fp, fp, fp, ci, ci, ci, ci, cf, ci, fp, fp, vp]),
    "vt_upsample2x_bicubic_add": (ci, [fp, fp, ci, ci, ci, ci, fp, vp]),
    "vt_conv3x3_create": (ci, [C.POINTER(vp), vp, ci, ci, vp]),
    "vt_conv3x3_destroy": (None, [vp]),
    "vt_conv3x3_forward": (ci, [vp, fp, ci, ci, ci, fp, ci, ci, vp]),
    "vt_conv3x3_forward_gn": (ci, [vp, fp, ci, ci, fp, fp, fp, ci, ci, ci, ci, fp, ci, ci, vp]),
    "vt_groupnorm_stats": (ci, [fp, ci, ci, ci, ci, ci, ci, cf, fp, vp]),
    "vt_groupnorm_workspace_doubles": (C.c_long, [ci, ci, ci, ci]),
    "vt_groupnorm_finalize": (ci, [fp, ci, ci, ci, ci, ci, cf, vp]),
    "vt_conv3x3_tiles": (ci, [ci, ci]),
    "vt_kpts_step": (ci, [vp, fp, fp, fp, ci, ci, vp, cf, cf, fp, fp, fp, ci, vp]),
    "vt_query_human_step": (ci, [vp, C.POINTER(VtMaps), fp, fp, fp, ci, ci, fp, fp, cf, cf, ci, cf, fp, fp, fp, vp]),
    "vt_objstep_head": (ci, [fp, fp, fp, fp, ci, fp, ci, fp, fp, ci, fp, fp, fp, ci, fp, vp]),
    "vt_temporal_loss2": (ci, [fp, ci, ci, cf, fp, cf, fp, fp, ci, vp]),
    "vt_objstep_tail": (ci, [fp, ci, fp, fp, ci, fp, fp, ci, fp, fp, fp, fp, cf, fp, fp, fp, fp,
                             fp, fp, fp, cf, fp, fp, fp, cf, ci, cf, cf, cf,
                             fp, vp, ci, cf, ci, fp, fp, fp, ci, fp, ci, fp, vp]),
    "vt_objstep_tail_temporal": (ci, [fp, cf, fp, cf, fp, ci,
                                      fp, ci, fp, fp, ci, fp, fp, ci, fp, fp, fp, fp, cf, fp, fp, fp, fp,
                                      fp, fp, fp, cf, fp, fp, fp, cf, ci, cf, cf, cf,
                                      fp, vp, ci, cf, ci, fp, fp, fp, ci, fp, ci, fp, vp]),
    "vt_smplstep_tail": (ci, [fp, fp, fp, ci, fp, fp, cf, fp, cf, fp,
                              fp, ci, fp, ci, fp, fp, ci, cf, fp, ci, fp, ci, fp, fp, ci, cf, fp, ci, fp, ci, fp, fp, ci, cf, ci, cf, cf, cf,
                              fp, vp, ci, cf, ci, fp, fp, fp, ci, fp, ci, vp]),
    "vt_stream_set_skip_flag": (ci, [vp, vp]),
    "vt_conv1x1_create": (ci, [C.POINTER(vp), vp, vp, ci, ci, vp]),
    "vt_stem7x7_create": (ci, [C.POINTER(vp), vp, vp, ci, ci, vp]),
    "vt_stem7x7_destroy": (None, [vp]),
    "vt_stem7x7_forward": (ci, [vp, fp, ci, ci, ci, ci, ci, fp, ci, ci, vp]),
    "vt_conv1x1_destroy": (None, [vp]),
    "vt_conv1x1_forward": (ci, [vp, fp, ci, ci, fp, fp, fp, ci, ci, ci, ci, fp, ci, ci, fp, ci, ci, fp, ci, vp]),
    "vt_conv3x3_forward_block": (ci, [vp, fp, ci, ci, fp, fp, fp, ci, ci, ci, ci, fp, ci, ci, fp, ci, ci, fp, ci, ci, fp, ci, vp]),
    "vt_conv3x3_forward_block_stats": (ci, [vp, fp, ci, ci, fp, fp, fp, ci, ci, ci, ci, fp, ci, ci, fp, ci, ci, fp, ci, ci, fp, ci, fp, ci, vp]),
    "vt_avgpool2x2_stats": (ci, [fp, ci, ci, ci, ci, fp, fp, ci, vp]),
    "vt_sweep_blocks": (ci, [ci]),
    "vt_upsample2x_bicubic_add_stats": (ci, [fp, fp, ci, ci, ci, ci, fp, fp, ci, vp]),
    "vt_conv3x3_forward_gn_stats": (ci, [vp, fp, ci, ci, fp, fp, fp, ci, ci, ci, ci, fp, ci, ci, fp, ci, vp]),
    "vt_triplane_render": (ci, [fp, fp, ci, ci, fp, ci, ci, fp, fp, fp, vp]),
    "vt_query_project_step": (ci, [vp, C.POINTER(VtMaps), fp, fp, fp, ci, ci, ci, cf, fp, fp, vp]),
    "vt_query_projection_floats": (C.c_long, [C.POINTER(VtMaps), ci]),
    "vt_query_build_projection": (ci, [vp, C.POINTER(VtMaps), ci, fp, vp]),
    "vt_so3_project_forward": (ci, [fp, fp, ci, fp, vp]),
    "vt_so3_project_backward": (ci, [fp, fp, ci, fp, fp, vp]),
    "vt_rigid_forward": (ci, [fp, ci, fp, fp, fp, ci, ci, fp, vp]),
    "vt_rigid_backward": (ci, [fp, ci, fp, ci, ci, fp, fp, fp, ci, vp]),
    "vt_accel_loss": (ci, [fp, ci, ci, fp, cf, fp, fp, vp]),
    "vt_accel_loss_strided": (ci, [fp, ci, ci, ci, fp, cf, fp, fp, vp]),
    "vt_velocity_loss": (ci, [fp, ci, ci, cf, fp, fp, vp]),
    "vt_kpts_loss": (ci, [fp, fp, fp, ci, ci, ci, vp, cf, cf, fp, fp, vp]),
    "vt_sqdiff_loss": (ci, [fp, ci, fp, ci, ci, ci, cf, cf, fp, fp, vp]),
    "vt_chamfer_ragged": (ci, [fp, fp, fp, fp, ci, cf, fp, fp, fp, vp]),
    "vt_chamfer_ws_bytes": (C.c_long, [C.c_long, C.c_long, ci]),
    "vt_chamfer_ragged_ws": (ci, [fp, fp, C.c_long, fp, fp, C.c_long, ci, cf, fp, fp, fp, fp, vp]),
    "vt_chamfer_ragged_idx": (ci, [fp, fp, C.c_long, fp, fp, fp, C.c_long, ci, cf, fp, fp, fp, ci, vp]),
    "vt_collision_workspace_bytes": (cl, [ci, ci]),
    "vt_collision_loss": (ci, [fp, ci, fp, ci, fp, ci, fp, ci, ci, cf, ci, cf, fp, fp, fp, fp, vp]),
    "vt_nn_distance": (ci, [fp, ci, fp, ci, ci, fp, vp]),
    "vt_sil_workspace_floats": (cl, [ci, ci, ci, ci]),
    "vt_sil_forward": (ci, [fp, ci, ci, fp, ci, fp, ci, fp, fp, fp, vp]),
    "vt_sil_backward": (ci, [fp, ci, ci, fp, ci, fp, ci, fp, fp, cf, fp, fp, vp]),
    "vt_sil_mask_loss": (ci, [fp, fp, fp, fp, ci, ci, cf, fp, fp, fp, vp]),
    "vt_sil_step": (ci, [fp, ci, ci, fp, ci, fp, ci, fp, fp, fp, cf, cf, fp, fp, fp, fp, fp, vp]),
    "vt_gen_round_compact": (ci, [fp, fp, fp, vp, ci, ci, cf, cf, vp, ci, ci, fp, vp, fp, vp, vp, vp]),
    "vt_gen_scatter_heads": (ci, [fp, ci, ci, ci, vp, vp, ci, fp, vp]),
    "vt_gen_resample": (ci, [fp, vp, vp, fp, ci, ci, ci, fp, fp, ci, cf, fp, vp]),
    "vt_sil_setup": (ci, [fp, fp, ci, ci, ci, fp, C.c_double, ci, C.c_double, C.c_double, C.POINTER(C.c_double), C.c_double, fp, fp, fp, fp, vp]),
    "vt_adam_step": (ci, [fp, fp, fp, fp, cl, ci, cf, cf, cf, cf, fp, vp]),
    "vt_adam_step_2d": (ci, [fp, cl, fp, cl, fp, fp, ci, ci, ci, cf, cf, cf, cf, fp, vp]),
    "vt_loss_reduce_and_stop": (ci, [fp, vp, ci, cf, ci, fp, fp, fp, ci, vp]),
    "vt_fill": (ci, [fp, cl, cf, vp]),
    "vt_selftest_mfma": (ci, [fp, fp, fp, vp]),
    "vt_query_set_clock_probe": (ci, [vp]),
    "vt_calibrate_workspace_bytes": (cl, []),
    "vt_calibrate": (ci, [vp, C.POINTER(C.c_double), vp]),
}

_lib = None


class VtError(RuntimeError):
    pass


def lib():
    """Load the HIP library (once).  Raises if it was not built -- there is no fallback path."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise VtError(f"{LIB_PATH} is missing: build it with `make -C vistracker_amd/csrc` "
                          f"(or `python -c 'import __graft_entry__ as g; g.build()'`); there is no CPU fallback")
        # PyTorch-ROCm ships its own libamdhip64 and must be the FIRST to load one: if this library pulled in /opt/rocm's copy before
        # torch was imported the process would hold two HIP runtimes, and the second one finds no device ("no ROCm-capable device is
        # detected" at the first hipMalloc).  With torch loaded first the dynamic linker resolves our dependency to the same runtime
        # object, which is also what makes torch's device pointers and streams valid inside the library.
        import torch  # noqa: F401
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
        _host_tuning()
    return _lib


def _host_tuning():
    """``VT_HOST_THP=off``: transparent huge pages off for this process (``prctl(PR_SET_THP_DISABLE)``).  Opt-in.

    Round 4 chased a 7 % loss of the FIRST process in a fresh container (same kernel times, slower host sections: profiles/r04_cold_process.txt).  THP
    looked like the cause for a while -- the first runs with this prctl, and one after touching 16 GB of anonymous memory, were fast -- but later first
    processes were slow with THP off as well (131.5, 132.0, 131.9 frames/s) and one with THP on was fast: not the cause, or not the only one.  What
    reliably removes the loss is keeping the launching threads out of lockstep (bench.py --stagger, PipelineConfig.fit_stagger_s).  The switch stays for
    hosts where huge-page compaction stalls are known to hurt."""
    if os.environ.get("VT_HOST_THP", "keep") != "off":
        return
    try:
        C.CDLL(None, use_errno=True).prctl(41, 1, 0, 0, 0)          # PR_SET_THP_DISABLE: checked at fault time, covers the mappings that exist already
    except Exception:          # noqa: BLE001 -- not Linux / no prctl: nothing to tune
        pass


VT_OK, VT_ERR_ARG, VT_ERR_HIP, VT_ERR_BUSY = 0, -1, -2, -3        # include/vistracker.h


def check(rc: int):
    if rc != 0:
        raise VtError(f"libvistracker_hip error {rc}: {lib().vt_last_error().decode()}")


def stream_ptr():
    import torch
    return torch.cuda.current_stream().cuda_stream


def dptr(t):
    """device pointer of a contiguous float32/int32/float64 CUDA tensor (or None -> NULL)"""
    if t is None:
        return None
    if not t.is_cuda:
        raise VtError("libvistracker_hip ops need CUDA (HIP) tensors; there is no CPU fallback in the product path")
    if not t.is_contiguous():
        raise VtError("libvistracker_hip ops need contiguous tensors")
    import torch
    if t.device.index != torch.cuda.current_device():
        # stream_ptr() is the current stream of the CURRENT device and HIP launches go to the current device: a tensor that lives
        # elsewhere would be addressed from the wrong GPU.  Callers wrap multi-device use in ``torch.cuda.device(tensor.device)``
        # (FitContext's entry points and the handle constructors do).
        raise VtError(f"tensor on cuda:{t.device.index} but the current device is cuda:{torch.cuda.current_device()}: "
                      "wrap the call in `with torch.cuda.device(tensor.device):`")
    return t.data_ptr()
