"""ORACLE (test infrastructure): ctypes bindings of the CPU ROIAlign oracles."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_LIB = os.path.join(HERE, "liboracle_roi.so")
REF_LIB = os.path.join(HERE, "_ref", "libref_roi_align.so")
_fp = ctypes.POINTER(ctypes.c_float)


def build():
    """Compile the C restatement (and the reference's own C file when /root/reference exists)."""
    subprocess.check_call(["make", "-s", "-C", HERE, "all"])


def _ptr(a):
    return a.ctypes.data_as(_fp)


def _lib(path):
    if not os.path.exists(path):
        if path == ORACLE_LIB:
            build()
        else:
            return None
    return ctypes.CDLL(path)


def have_reference_lib():
    return os.path.exists(REF_LIB)


def forward(feat, rois, ah, aw, scale):
    """our restatement; feat [B,C,H,W] f32, rois [n,5] f32 -> [n,C,ah,aw] f32"""
    lib = _lib(ORACLE_LIB)
    feat = np.ascontiguousarray(feat, np.float32)
    rois = np.ascontiguousarray(rois, np.float32)
    B, C, H, W = feat.shape
    out = np.zeros((rois.shape[0], C, ah, aw), np.float32)
    lib.oracle_roi_align_forward(_ptr(feat), _ptr(rois), _ptr(out), rois.shape[0], C, H, W, ah, aw,
                                 ctypes.c_float(scale))
    return out


def backward(top_grad, rois, feat_shape, scale):
    lib = _lib(ORACLE_LIB)
    top_grad = np.ascontiguousarray(top_grad, np.float32)
    rois = np.ascontiguousarray(rois, np.float32)
    B, C, H, W = feat_shape
    n, _, ah, aw = top_grad.shape
    out = np.zeros((B, C, H, W), np.float32)
    lib.oracle_roi_align_backward(_ptr(top_grad), _ptr(rois), _ptr(out), n, C, H, W, ah, aw,
                                  ctypes.c_float(scale))
    return out


def reference_forward(feat, rois, ah, aw, scale):
    """the REFERENCE's ROIAlignForwardCpu (roi_align.c:80-136), compiled into oracle/_ref."""
    lib = _lib(REF_LIB)
    if lib is None:
        raise RuntimeError("oracle/_ref/libref_roi_align.so is not built (run `make -C oracle`)")
    feat = np.ascontiguousarray(feat, np.float32)
    rois = np.ascontiguousarray(rois, np.float32)
    B, C, H, W = feat.shape
    out = np.zeros((rois.shape[0], C, ah, aw), np.float32)
    # void ROIAlignForwardCpu(const float* bottom, float scale, int num_rois, int H, int W, int C,
    #                         int AH, int AW, const float* rois, float* top)
    lib.ROIAlignForwardCpu(_ptr(feat), ctypes.c_float(scale), rois.shape[0], H, W, C, ah, aw,
                           _ptr(rois), _ptr(out))
    return out
