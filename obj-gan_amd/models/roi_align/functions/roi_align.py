"""RoIAlignFunction (reference image_generation/models/roi_align/functions/roi_align.py:7-51).

The reference is an old-style (instance) autograd Function calling the cffi extension; here it is
a new-style static Function over the C-ABI `objgan_roi_align_forward/_backward`
(include/objgan_hip.h).  `RoIAlignFunction(ah, aw, scale)(features, rois)` keeps working through
the small callable wrapper below.
"""
from objgan_hip.ops import RoIAlignFunction as _Fn, roi_align


class RoIAlignFunction(object):
    def __init__(self, aligned_height, aligned_width, spatial_scale):
        self.aligned_width = int(aligned_width)
        self.aligned_height = int(aligned_height)
        self.spatial_scale = float(spatial_scale)

    def __call__(self, features, rois):
        return roi_align(features, rois, self.aligned_height, self.aligned_width, self.spatial_scale)


__all__ = ["RoIAlignFunction", "roi_align", "_Fn"]
