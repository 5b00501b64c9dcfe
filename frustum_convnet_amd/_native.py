"""ctypes binding of libfcn_hip.so (see include/fcn_hip.h for the C-ABI).

There is deliberately no fallback: if the shared library is missing, or a call returns non-zero,
this raises -- a PyTorch/CPU substitute would silently void every parity and performance claim.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, os.environ.get("FCN_LIB_NAME", "libfcn_hip.so"))   # (FCN_LIB_NAME: A/B builds)

c_fp = ctypes.c_void_p


class PnDesc(ctypes.Structure):
    _fields_ = [("B", ctypes.c_int32), ("N", ctypes.c_int32), ("L", ctypes.c_int32), ("K", ctypes.c_int32),
                ("C1", ctypes.c_int32), ("C2", ctypes.c_int32), ("C3", ctypes.c_int32),
                ("nvec", ctypes.c_int32), ("training", ctypes.c_int32),
                ("eps", ctypes.c_float), ("momentum", ctypes.c_float), ("nlc", ctypes.c_int32),
                ("precision", ctypes.c_int32), ("grouped", ctypes.c_int32)]


CN_MAXLEV = 5        # FCN_CN_MAXLEV
CN_MAXLAYER = 18     # FCN_CN_MAXLAYER


class CnDesc(ctypes.Structure):
    _fields_ = [("B", ctypes.c_int32), ("L", ctypes.c_int32 * CN_MAXLEV), ("nvec", ctypes.c_int32),
                ("reg_out", ctypes.c_int32), ("training", ctypes.c_int32),
                ("eps", ctypes.c_float), ("momentum", ctypes.c_float), ("prepacked", ctypes.c_int32),
                ("precision", ctypes.c_int32), ("nlev", ctypes.c_int32), ("c1", ctypes.c_int32)]


class CnParams(ctypes.Structure):
    _fields_ = [("W", c_fp * CN_MAXLAYER), ("gamma", c_fp * CN_MAXLAYER), ("beta", c_fp * CN_MAXLAYER),
                ("running_mean", c_fp * CN_MAXLAYER), ("running_var", c_fp * CN_MAXLAYER),
                ("num_batches_tracked", c_fp * CN_MAXLAYER), ("bias", c_fp)]


class CnWs(ctypes.Structure):
    _fields_ = [("y", c_fp), ("dz", c_fp), ("wp", c_fp), ("bn", c_fp), ("stat", c_fp), ("bstat", c_fp),
                ("coef", c_fp), ("partial", c_fp), ("oh64", c_fp), ("flags", c_fp)]


class PnParams(ctypes.Structure):
    _fields_ = [("W", c_fp * 3), ("gamma", c_fp * 3), ("beta", c_fp * 3),
                ("running_mean", c_fp * 3), ("running_var", c_fp * 3), ("num_batches_tracked", c_fp * 3)]


class PnWs(ctypes.Structure):
    _fields_ = [("woff", c_fp), ("ent", c_fp), ("ewin", c_fp), ("tiles", c_fp), ("y2", c_fp), ("y3", c_fp), ("amax", c_fp),
                ("stat", c_fp), ("bn", c_fp), ("gmax", c_fp), ("dy3", c_fp), ("dz2", c_fp), ("bstat", c_fp),
                ("coef", c_fp), ("partial", c_fp), ("nsplit", ctypes.c_int32), ("gmom", c_fp), ("wenc", c_fp), ("flags", c_fp), ("pkey", c_fp),
                ("partial_both", ctypes.c_int32)]


class InpDesc(ctypes.Structure):
    _fields_ = [("B", ctypes.c_int32), ("N", ctypes.c_int32), ("pt_stride", ctypes.c_int32), ("L", ctypes.c_int32 * 4),
                ("stride", ctypes.c_double * 4), ("max_depth", ctypes.c_double),
                ("random_flip", ctypes.c_int32), ("random_shift", ctypes.c_int32)]


class Inp5Desc(ctypes.Structure):
    _fields_ = [("B", ctypes.c_int32), ("N", ctypes.c_int32), ("pt_stride", ctypes.c_int32), ("L", ctypes.c_int32 * 5),
                ("stride", ctypes.c_double * 5), ("max_depth", ctypes.c_double),
                ("random_flip", ctypes.c_int32), ("random_shift", ctypes.c_int32)]


class InpRefineDesc(ctypes.Structure):
    _fields_ = [("B", ctypes.c_int32), ("N", ctypes.c_int32), ("pt_stride", ctypes.c_int32), ("Lpad", ctypes.c_int32 * 4),
                ("stride", ctypes.c_double * 4), ("random_flip", ctypes.c_int32), ("random_shift", ctypes.c_int32)]


EXPORTS = ("fcn_arch", "fcn_build_hash", "fcn_stat_replicas", "fcn_query_depth_point_f32", "fcn_query_depth_point_multi_f32", "fcn_pn_wgrad_rows", "fcn_pn_compact", "fcn_pn_group_compact", "fcn_pn_group_compact2",
           "fcn_pn_pack_weights", "fcn_pn_pack_weights_all", "fcn_pn_forward", "fcn_pn_backward", "fcn_pn_backward2", "fcn_pn_backward3", "fcn_pn_backward_dense", "fcn_pn_conv_fwd", "fcn_det_loss_tail", "fcn_det_loss_tail_rows", "fcn_det_loss_tail_rows2", "fcn_det_iou_metrics",
           "fcn_det_loss_tail_scratch_floats", "fcn_adam_step_f32", "fcn_sgd_step_f32", "fcn_adam_step_slots", "fcn_prepare_inputs", "fcn_prepare_inputs_refine", "fcn_prepare_inputs_sunrgbd", "fcn_stamp", "fcn_stream_capture_id",
           "fcn_convnet_sizes", "fcn_convnet_logits_ld", "fcn_convnet_pack", "fcn_convnet_forward", "fcn_convnet_forward2",
           "fcn_convnet_backward", "fcn_box3d_iou_pair_f32", "fcn_decode_detections", "fcn_rotate_nms_3d")

_lib = None


FLAG_NONFINITE = 1      # FCN_FLAG_NONFINITE


class NativeError(RuntimeError):
    pass


def lib():
    """Loads the library once.  Raises ImportError (loudly) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "frustum_convnet_amd: %s not found -- build it with `python -m frustum_convnet_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU/PyTorch fallback for the hot path." % LIB_PATH)
    # torch bundles its own libamdhip64; load it FIRST so this library binds to the same HIP runtime (streams and
    # device pointers are only meaningful inside one runtime instance).
    import torch  # noqa: F401
    L = ctypes.CDLL(LIB_PATH)
    L.fcn_arch.restype = ctypes.c_int
    if hasattr(L, "fcn_build_hash"):          # (the host emulation of tests/host_harness and older A/B builds lack it)
        L.fcn_build_hash.restype = ctypes.c_char_p
        L.fcn_build_hash.argtypes = []
    L.fcn_arch.argtypes = []
    L.fcn_pn_wgrad_rows.restype = ctypes.c_int
    L.fcn_pn_wgrad_rows.argtypes = []
    if hasattr(L, "fcn_stat_replicas"):
        L.fcn_stat_replicas.restype = ctypes.c_int
        L.fcn_stat_replicas.argtypes = []
    L.fcn_query_depth_point_f32.restype = ctypes.c_int
    L.fcn_query_depth_point_f32.argtypes = [
        c_fp, ctypes.c_int64, ctypes.c_int64, c_fp, ctypes.c_int64, ctypes.c_int64,
        ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_int, c_fp, c_fp, c_fp]
    if hasattr(L, "fcn_query_depth_point_multi_f32"):
        L.fcn_query_depth_point_multi_f32.restype = ctypes.c_int
        L.fcn_query_depth_point_multi_f32.argtypes = [ctypes.c_int, c_fp, ctypes.c_int64, ctypes.c_int64, c_fp, c_fp, c_fp,
                                                      ctypes.c_int, ctypes.c_int, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp]
    L.fcn_pn_compact.restype = ctypes.c_int
    L.fcn_pn_compact.argtypes = [ctypes.POINTER(PnDesc), c_fp, c_fp, c_fp, c_fp, ctypes.POINTER(PnWs), c_fp]
    L.fcn_pn_group_compact.restype = ctypes.c_int
    L.fcn_pn_group_compact.argtypes = [ctypes.c_int, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp]
    L.fcn_pn_group_compact2.restype = ctypes.c_int
    L.fcn_pn_group_compact2.argtypes = [ctypes.c_int, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, ctypes.c_int, c_fp]
    if hasattr(L, "fcn_pn_pack_weights") or "FCN_LIB_NAME" not in os.environ:     # (A/B builds of older kernel sources lack them)
        L.fcn_pn_pack_weights.restype = ctypes.c_int
        L.fcn_pn_pack_weights.argtypes = [ctypes.POINTER(PnDesc), ctypes.POINTER(PnParams), ctypes.POINTER(PnWs), c_fp]
        L.fcn_pn_pack_weights_all.restype = ctypes.c_int
        L.fcn_pn_pack_weights_all.argtypes = [ctypes.c_int, c_fp, c_fp, c_fp, c_fp]
    L.fcn_pn_forward.restype = ctypes.c_int
    L.fcn_pn_forward.argtypes = [ctypes.POINTER(PnDesc), ctypes.POINTER(PnParams), c_fp, c_fp,
                                 ctypes.POINTER(PnWs), c_fp, c_fp]
    L.fcn_pn_backward.restype = ctypes.c_int
    L.fcn_pn_backward.argtypes = [ctypes.POINTER(PnDesc), ctypes.POINTER(PnParams), c_fp,
                                  ctypes.POINTER(PnWs), c_fp * 3, c_fp * 3, c_fp * 3, c_fp]
    L.fcn_pn_backward2.restype = ctypes.c_int
    L.fcn_pn_backward2.argtypes = [ctypes.POINTER(PnDesc), ctypes.POINTER(PnParams), c_fp,
                                   ctypes.POINTER(PnWs), c_fp * 3, c_fp * 3, c_fp * 3, c_fp, c_fp, ctypes.POINTER(c_fp)]
    if hasattr(L, "fcn_pn_backward3"):
        L.fcn_pn_backward3.restype = ctypes.c_int
        L.fcn_pn_backward3.argtypes = [ctypes.POINTER(PnDesc), ctypes.POINTER(PnParams), c_fp,
                                       ctypes.POINTER(PnWs), c_fp * 3, c_fp * 3, c_fp * 3, c_fp, c_fp, c_fp, ctypes.POINTER(c_fp)]
    if hasattr(L, "fcn_pn_backward_dense"):
        L.fcn_pn_backward_dense.restype = ctypes.c_int
        L.fcn_pn_backward_dense.argtypes = [ctypes.POINTER(PnDesc), ctypes.POINTER(PnParams), c_fp,
                                            ctypes.POINTER(PnWs), c_fp * 3, c_fp * 3, c_fp * 3, c_fp]
    L.fcn_pn_conv_fwd.restype = ctypes.c_int
    L.fcn_pn_conv_fwd.argtypes = [ctypes.POINTER(PnDesc), ctypes.POINTER(PnParams), ctypes.POINTER(PnWs),
                                  ctypes.c_int, ctypes.c_int, c_fp]
    L.fcn_det_loss_tail.restype = ctypes.c_int
    L.fcn_det_loss_tail.argtypes = [c_fp] * 9 + [ctypes.c_int] * 4 + [ctypes.c_float] * 4 + [c_fp] * 4
    L.fcn_adam_step_f32.restype = ctypes.c_int
    L.fcn_adam_step_f32.argtypes = [c_fp] * 4 + [ctypes.c_int64] + [c_fp] * 3
    L.fcn_sgd_step_f32.restype = ctypes.c_int
    L.fcn_sgd_step_f32.argtypes = [c_fp] * 3 + [ctypes.c_int64] + [c_fp] * 2
    L.fcn_prepare_inputs.restype = ctypes.c_int
    L.fcn_prepare_inputs.argtypes = [ctypes.POINTER(InpDesc)] + [c_fp] * 13 + [c_fp * 4] + [c_fp] * 7
    L.fcn_prepare_inputs_sunrgbd.restype = ctypes.c_int
    L.fcn_prepare_inputs_sunrgbd.argtypes = [ctypes.POINTER(Inp5Desc)] + [c_fp] * 15 + [c_fp * 5] + [c_fp] * 7
    L.fcn_prepare_inputs_refine.restype = ctypes.c_int
    L.fcn_prepare_inputs_refine.argtypes = [ctypes.POINTER(InpRefineDesc)] + [c_fp] * 12 + [c_fp * 4] + [c_fp] * 8
    L.fcn_stamp.restype = ctypes.c_int
    L.fcn_stamp.argtypes = [c_fp, c_fp]
    L.fcn_stream_capture_id.restype = ctypes.c_int
    L.fcn_stream_capture_id.argtypes = [c_fp, ctypes.POINTER(ctypes.c_uint64)]
    L.fcn_adam_step_slots.restype = ctypes.c_int64
    L.fcn_adam_step_slots.argtypes = [ctypes.c_int64]
    L.fcn_det_loss_tail_rows.restype = ctypes.c_int
    L.fcn_det_loss_tail_rows.argtypes = [c_fp] * 8 + [ctypes.c_int] * 4 + [ctypes.c_float] * 4 + [c_fp] * 3
    L.fcn_det_loss_tail_rows2.restype = ctypes.c_int
    L.fcn_det_loss_tail_rows2.argtypes = [c_fp] * 8 + [ctypes.c_int] * 4 + [ctypes.c_float] * 4 + [c_fp] * 5
    L.fcn_det_iou_metrics.restype = ctypes.c_int
    L.fcn_det_iou_metrics.argtypes = [c_fp, ctypes.c_int] + [c_fp] * 6 + [ctypes.c_int] * 4 + [ctypes.c_float] + [c_fp] * 3
    L.fcn_det_loss_tail_scratch_floats.restype = ctypes.c_int
    L.fcn_det_loss_tail_scratch_floats.argtypes = [ctypes.c_int, ctypes.c_int]
    L.fcn_convnet_sizes.restype = ctypes.c_int
    L.fcn_convnet_sizes.argtypes = [ctypes.POINTER(CnDesc), ctypes.POINTER(ctypes.c_int64 * 6)]
    L.fcn_convnet_logits_ld.restype = ctypes.c_int
    L.fcn_convnet_logits_ld.argtypes = [ctypes.POINTER(CnDesc)]
    L.fcn_convnet_pack.restype = ctypes.c_int
    L.fcn_convnet_pack.argtypes = [ctypes.POINTER(CnDesc), ctypes.POINTER(CnParams), ctypes.POINTER(CnWs), c_fp, c_fp]
    L.fcn_convnet_forward.restype = ctypes.c_int
    L.fcn_convnet_forward.argtypes = [ctypes.POINTER(CnDesc), ctypes.POINTER(CnParams), ctypes.POINTER(CnWs),
                                      c_fp * CN_MAXLEV, c_fp, c_fp, c_fp]
    L.fcn_convnet_forward2.restype = ctypes.c_int
    L.fcn_convnet_forward2.argtypes = [ctypes.POINTER(CnDesc), ctypes.POINTER(CnParams), ctypes.POINTER(CnWs),
                                       c_fp * CN_MAXLEV, c_fp, c_fp, c_fp, ctypes.POINTER(c_fp)]
    L.fcn_convnet_backward.restype = ctypes.c_int
    L.fcn_convnet_backward.argtypes = [ctypes.POINTER(CnDesc), ctypes.POINTER(CnParams), ctypes.POINTER(CnWs),
                                       c_fp * CN_MAXLEV, c_fp, c_fp, c_fp * CN_MAXLEV, c_fp * CN_MAXLAYER, c_fp * CN_MAXLAYER,
                                       c_fp * CN_MAXLAYER, c_fp, c_fp, c_fp, ctypes.POINTER(c_fp)]
    L.fcn_box3d_iou_pair_f32.restype = ctypes.c_int
    L.fcn_box3d_iou_pair_f32.argtypes = [c_fp, c_fp, ctypes.c_int, c_fp, c_fp]
    L.fcn_decode_detections.restype = ctypes.c_int
    L.fcn_decode_detections.argtypes = [c_fp, ctypes.c_int] + [c_fp] * 5 + [ctypes.c_int] * 5 + [c_fp] * 3
    L.fcn_rotate_nms_3d.restype = ctypes.c_int
    L.fcn_rotate_nms_3d.argtypes = [c_fp] * 3 + [ctypes.c_int] * 3 + [ctypes.c_float, ctypes.c_int] + [c_fp] * 3
    _lib = L
    return L


def check(rc, what):
    if rc != 0:
        raise NativeError("%s failed with code %d (hipError_t or FCN_E_*; see include/fcn_hip.h)" % (what, rc))


def ptr(t):
    """data pointer of a tensor (or None -> NULL)."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def current_stream(device=None):
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


_unknown_capture = [0]


def capture_id(device=None):
    """Id of the hipGraph capture the current stream of `device` is part of, 0 when it is not capturing (fcn_stream_capture_id:
    hipStreamGetCaptureInfo asked through libfcn_hip.so, i.e. of the HIP runtime the kernels are launched on -- never of a second
    runtime dlopen'ed by name).  Used to keep state created during one capture from leaking into eager code or another capture
    (PointNetFeat.prefetch).  When the id cannot be determined the answer is a fresh NEGATIVE number every time: two unknown
    ids never compare equal, so state tagged with one is always treated as foreign (dropped, never consumed)."""
    import torch
    if not torch.cuda.is_available() or not torch.cuda.is_current_stream_capturing():
        return 0
    cid = ctypes.c_uint64(0)
    rc = lib().fcn_stream_capture_id(current_stream(device), ctypes.byref(cid))
    if rc != 0 or cid.value == 0:             # capturing (torch says so), id unknown
        _unknown_capture[0] -= 1
        return _unknown_capture[0]
    return int(cid.value)
