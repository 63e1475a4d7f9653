"""ctypes binding of libobjgan_hip.so (the C-ABI declared in include/objgan_hip.h).

The reference binds its native op with cffi through torch.utils.ffi (reference
image_generation/models/roi_align/_ext/roi_align/__init__.py:1-15); torch.utils.ffi no longer
exists and cffi is not installed here, so the zero-dependency equivalent -- ctypes -- is used.
There is NO fallback: if the shared library is missing or a call fails, an exception is raised.
"""
import ctypes
import os

from . import build as _build

_c_int = ctypes.c_int
_c_long = ctypes.c_long
_c_float = ctypes.c_float
_c_double = ctypes.c_double
_ptr = ctypes.c_void_p

# name -> argtypes  (every function returns int: 1 ok, 0 bad args, <0 -hipError)
SIGNATURES = {
    "objgan_roi_align_forward": [_ptr, _ptr, _ptr, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_float, _ptr],
    "objgan_roi_align_backward": [_ptr, _ptr, _ptr, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_float, _ptr],
    "objgan_roi_align_backward_ordered": [_ptr, _ptr, _ptr, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int,
                                          _c_float, _ptr, _c_long, _ptr],
    "objgan_avgpool2s1_forward": [_ptr, _ptr, _c_long, _c_int, _c_int, _ptr],
    "objgan_avgpool2s1_backward": [_ptr, _ptr, _c_long, _c_int, _c_int, _ptr],
    "objgan_conv_igemm": [_ptr, _ptr, _ptr, _ptr, _ptr,
                          _c_int, _c_int, _c_int, _c_int, _c_int, _c_int,
                          _c_int, _c_int, _c_int, _c_int,
                          _c_int, _ptr, _ptr, _ptr,
                          _c_int, _c_int, _c_int,
                          _c_int, _c_int, _c_int, _c_int, _c_int, _c_int,
                          _c_int, _c_int, _c_int, _c_int, _ptr, _ptr, _ptr, _ptr, _c_long, _ptr],
    "objgan_absmax_partials": [_ptr, _c_long, _ptr, _ptr],
    "objgan_h2_records": [_ptr, _ptr, _ptr, _c_int, _c_int, _c_long, _ptr],
    "objgan_reflect_ring_fold": [_ptr, _ptr, _c_long, _c_int, _c_int, _ptr],
    "objgan_conv_dgrad_s2_phases": [_ptr, _ptr, _ptr, _ptr, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int,
                                    _c_int, _ptr, _ptr, _ptr, _c_int, _c_int, _c_int, _c_int, _ptr, _ptr, _c_long, _ptr],
    "objgan_conv_wgrad": [_ptr, _ptr, _ptr, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int,
                          _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _ptr, _ptr, _ptr, _c_long, _ptr],
    "objgan_lstm_bidir_forward": [_ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr,
                                  _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _ptr],
    "objgan_norm_forward": [_ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr,
                            _c_int, _c_int, _c_int, _c_int, _c_int, _c_float, _c_float, _ptr, _ptr],
    "objgan_norm_amax_supported": [_c_int] * 5,
    "objgan_norm_apply": [_ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _c_int, _c_int, _c_int, _c_int, _c_int, _ptr],
    "objgan_norm_backward": [_ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr,
                             _c_int, _c_int, _c_int, _c_int, _c_int, _ptr, _ptr],
    "objgan_act_backward": [_ptr, _ptr, _ptr, _c_long, _c_int, _ptr, _ptr],
    "objgan_channel_sum": [_ptr, _ptr, _c_int, _c_int, _c_int, _ptr, _ptr],
    "objgan_attn_general_forward": [_ptr, _ptr, _ptr, _ptr, _ptr, _c_int, _c_int, _c_int, _c_int, _ptr],
    "objgan_attn_general_backward": [_ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _c_int, _c_int, _c_int, _c_int, _ptr, _ptr],
    "objgan_attn_bu_forward": [_ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_float, _ptr],
    "objgan_attn_bu_backward": [_ptr, _ptr, _ptr, _c_int, _c_int, _c_int, _c_int, _ptr],
    "objgan_masked_max_forward": [_ptr, _ptr, _ptr, _c_int, _c_int, _c_int, _c_int, _c_long, _c_long, _c_long, _ptr],
    "objgan_masked_max_backward": [_ptr, _ptr, _ptr, _ptr, _c_int, _c_int, _c_int, _c_int, _c_long, _c_long, _c_long, _ptr, _ptr],
    "objgan_softmax_strided_forward": [_ptr, _ptr, _c_long, _c_int, _c_long, _c_float, _ptr, _c_int, _ptr, _ptr],
    "objgan_softmax_strided_backward": [_ptr, _ptr, _ptr, _c_long, _c_int, _c_long, _c_float, _ptr],
    "objgan_bilinear_forward": [_ptr, _ptr, _c_long, _c_int, _c_int, _c_int, _c_int, _ptr],
    "objgan_bilinear_backward": [_ptr, _ptr, _c_long, _c_int, _c_int, _c_int, _c_int, _ptr],
    "objgan_sum2x2": [_ptr, _ptr, _c_long, _c_int, _c_int, _ptr],
    "objgan_reflect_fold": [_ptr, _ptr, _c_long, _c_int, _c_int, _ptr],
    "objgan_conv_pack_job_bytes": [],
    "objgan_conv_pack_job": [_ptr, _ptr, _ptr, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int,
                             _c_int, _ptr, _c_int, _c_int, _c_int, _c_int],
    "objgan_conv_pack_job_phase": [_ptr, _ptr, _ptr, _c_int, _c_int, _c_int, _c_int, _ptr, _c_int, _c_int],
    "objgan_conv_pack_job_thin_phase": [_ptr, _ptr, _ptr, _c_int, _c_int, _c_int],
    "objgan_conv_dgrad_s2_thin": [_ptr, _ptr, _ptr, _ptr, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _ptr],
    "objgan_conv_pack_jobs_run": [_ptr, _c_int, _ptr],
    "objgan_bce_const_forward": [_ptr, _ptr, _c_int, _c_float, _ptr],
    "objgan_bce_const_backward": [_ptr, _ptr, _ptr, _c_int, _c_float, _ptr],
    "objgan_bmm_strided": [_ptr, _ptr, _ptr, _c_int, _c_int, _c_int, _c_int] + [_c_long] * 9 + [_ptr],
    "objgan_lift_taps_forward": [_ptr, _ptr, _ptr, _ptr, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int,
                                 _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr],
    "objgan_lift_taps_backward": [_ptr, _ptr, _ptr, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int,
                                  _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr],
    "objgan_pool2d_forward": [_ptr, _ptr, _ptr, _c_long, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _ptr],
    "objgan_pool2d_backward": [_ptr, _ptr, _ptr, _c_long, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _ptr],
    "objgan_adam_step": [_ptr, _ptr, _ptr, _ptr, _c_long, _c_double, _c_double, _c_double, _c_double, _c_int, _c_float, _ptr],
    "objgan_adam_step_gated": [_ptr, _ptr, _ptr, _ptr, _c_long, _c_double, _c_double, _c_double, _c_double,
                               _ptr, _ptr, _ptr, _c_float, _ptr],
    "objgan_ema_update": [_ptr, _ptr, _c_long, _c_float, _c_float, _ptr],
    "objgan_resize_pil_kmax": [_c_int, _c_int],
    "objgan_resize_pil_rgb8": [_ptr, _ptr, _ptr, _ptr, _c_int, _c_int, _c_int, _c_int, _ptr, _ptr, _ptr, _ptr],
    "objgan_mask_resize": [_ptr, _c_int, _c_int, _c_int, _ptr, _ptr, _ptr, _ptr, _ptr],
    "objgan_jpeg_parse": [ctypes.c_char_p, _c_long, _ptr],
    "objgan_jpeg_decode": [_ptr, _ptr, _ptr, _c_int, _ptr, _ptr, _c_long, _ptr, _ptr, _ptr],
    "objgan_prof_enable": [_c_int],
    "objgan_conv_bank_layout": [_c_int] * 10,
    "objgan_conv_wgrad_rec_ok": [_c_int] * 8,
    "objgan_conv_wgrad_bfb_ok": [_c_int] * 8,
    "objgan_nhwc_bf16": [_ptr, _ptr, _c_int, _c_int, _c_long, _ptr],
    "objgan_prof_collect": [_ptr, _ptr, _ptr],
    "objgan_prof_dump": [_ptr, _ptr, _ptr, _c_int, _ptr],
}
LONG_RETURN = {"objgan_conv_packed_floats": [_c_int, _c_int, _c_int],
               "objgan_conv_igemm_ws_floats": [_c_int] * 22,
               "objgan_conv_wgrad_ws_floats": [_c_int] * 13,
               "objgan_conv_dgrad_s2_phases_ws_floats": [_c_int] * 5,
               "objgan_conv_dgrad_s2_thin_floats": [_c_int, _c_int],
               "objgan_h2_records_floats": [_c_int, _c_int, _c_long],
               "objgan_nhwc_bf16_floats": [_c_int, _c_int, _c_long],
               "objgan_norm_ws_floats": [_c_int] * 4,
               "objgan_attn_general_backward_ws_floats": [_c_int] * 4,
               "objgan_masked_max_backward_ws_floats": [_c_int] * 4,
               "objgan_channel_sum_ws_floats": [_c_int] * 3,
               "objgan_roi_align_backward_ws_floats": [_c_int] * 7,
               "objgan_jpeg_desc_bytes": [],
               "objgan_jpeg_seg_bytes": [],
               "objgan_jpeg_plan": [_ptr, _c_int, _ptr, _ptr, _ptr]}

_LIB = None


class ObjganHipError(RuntimeError):
    pass


def lib_path():
    return _build.LIB_PATH


def load(build_if_missing=True):
    """Load (building first if the .so is absent and hipcc is available). Never falls back."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        if not build_if_missing:
            raise ObjganHipError("libobjgan_hip.so not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
        _build.build(verbose=False)
    # torch first: it ships its own libamdhip64, and the library's launches go to torch's streams.
    # Loaded the other way round, libobjgan_hip.so binds the system HIP runtime, the process ends up
    # with two runtimes, and every launch on a torch stream fails with hipErrorNoDevice (100).
    import torch  # noqa: F401
    lib = ctypes.CDLL(path)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError if the symbol is missing: loud by design
        fn.argtypes = argtypes
        fn.restype = _c_int
    for name, argtypes in LONG_RETURN.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = _c_long
    _LIB = lib
    return lib


def call(name, *args):
    """Call an entry point and turn the reference-style return code into an exception."""
    rc = getattr(load(), name)(*args)
    if rc == 1:
        return
    if rc == 0:
        raise ObjganHipError("%s: invalid arguments (returned 0)" % name)
    raise ObjganHipError("%s: HIP launch failed (hipError %d)" % (name, -rc))
