"""ROIAlign forward on the GPU: drop-in for the reference's
``detectron2.layers.roi_align.ROIAlign(output_size, spatial_scale, sampling_ratio, aligned).forward(input, rois)``
(models/richsem/richsem.py:25, :878) -- detectron2 is not needed.  Kernel richsem_amd/csrc/msda_roi.h, C ABI
``msda_roi_align_forward_{f32,f64}``.  Forward only (RichSem pools the frozen CLIP feature map)."""
import torch

from . import _lib


class ROIAlign:
    def __init__(self, output_size, spatial_scale, sampling_ratio, aligned=True):
        self.output_size = (output_size, output_size) if isinstance(output_size, int) else tuple(output_size)
        self.spatial_scale, self.sampling_ratio, self.aligned = float(spatial_scale), int(sampling_ratio), bool(aligned)

    def forward(self, input, rois):
        """input (N, C, H, W); rois (K, 5) = (batch index, x1, y1, x2, y2) -> (K, C, output_h, output_w)"""
        if not input.is_cuda:
            raise RuntimeError("Not implemented on the CPU")
        assert rois.dim() == 2 and rois.size(1) == 5
        dt = input.dtype
        if dt not in (torch.float32, torch.float64):
            raise RuntimeError(f"ROIAlign: float32 / float64 input, got {dt}")
        x, r = input.contiguous(), rois.to(dt).contiguous()
        N, C, H, W = x.shape
        ph, pw = self.output_size
        out = torch.empty((r.shape[0], C, ph, pw), dtype=dt, device=x.device)
        if r.shape[0] == 0:
            return out
        fn = getattr(_lib.load(), "msda_roi_align_forward_" + ("f32" if dt == torch.float32 else "f64"))
        with _lib.on_device(x.device):
            _lib.check(fn(x.data_ptr(), r.data_ptr(), r.shape[0], N, C, H, W, ph, pw, self.spatial_scale, self.sampling_ratio,
                          int(self.aligned), out.data_ptr(), _lib.raw_stream(x.device)))
        return out

    __call__ = forward


def roi_align(input, rois, output_size, spatial_scale=1.0, sampling_ratio=0, aligned=True):
    return ROIAlign(output_size, spatial_scale, sampling_ratio, aligned).forward(input, rois)
