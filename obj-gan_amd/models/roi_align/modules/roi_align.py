"""RoIAlign / RoIAlignAvg / RoIAlignMax modules (reference
image_generation/models/roi_align/modules/roi_align.py:6-42): same constructor arguments and
shapes, executed by the gfx950 ROIAlign kernel (csrc/roi_align.hip)."""
from torch.nn.modules.module import Module

from objgan_hip import ops


class RoIAlign(Module):
    def __init__(self, aligned_height, aligned_width, spatial_scale):
        super(RoIAlign, self).__init__()
        self.aligned_width = int(aligned_width)
        self.aligned_height = int(aligned_height)
        self.spatial_scale = float(spatial_scale)

    def forward(self, features, rois):
        return ops.roi_align(features, rois, self.aligned_height, self.aligned_width, self.spatial_scale)


class RoIAlignAvg(RoIAlign):
    """ROIAlign on an (h+1) x (w+1) grid followed by a 2x2 stride-1 average."""

    def forward(self, features, rois):
        x = ops.roi_align(features, rois, self.aligned_height + 1, self.aligned_width + 1,
                          self.spatial_scale)
        return ops.avgpool2s1(x)


class RoIAlignMax(RoIAlign):
    """ROIAlign on an (h+1) x (w+1) grid followed by a 2x2 stride-1 max (off the hot path:
    the pooling itself is delegated to torch)."""

    def forward(self, features, rois):
        import torch.nn.functional as F
        x = ops.roi_align(features, rois, self.aligned_height + 1, self.aligned_width + 1,
                          self.spatial_scale)
        return F.max_pool2d(x, kernel_size=2, stride=1)
