"""ImageUpsample with the reference's API (transforms/imageupsample.py:5-26): nearest x scale, then / scale^2."""
from xmm_superres_denoise.engine import image_upsample as _hip_upsample


class ImageUpsample:
    def __init__(self, scale_factor):
        if int(scale_factor) != scale_factor or scale_factor < 1:
            raise ValueError(f"scale_factor must be a positive integer (got {scale_factor})")
        self.scale_factor = int(scale_factor)

    def __call__(self, x):
        return _hip_upsample(x.contiguous(), self.scale_factor)
