"""Crop: present for API parity with the reference's transforms/__init__.py:1.  The reference disables it
(data/datamodule.py:22-29), so it is off the hot path; this is a plain slicing helper."""


class Crop:
    def __init__(self, crop_p: float, mode: str = "center"):
        self.crop_p = crop_p
        self.mode = mode

    def __call__(self, img):
        h, w = img.shape[-2:]
        ch, cw = int(h * self.crop_p), int(w * self.crop_p)
        if self.mode != "center":
            raise NotImplementedError("only the deterministic 'center' mode is provided")
        y0, x0 = (h - ch) // 2, (w - cw) // 2
        return img[..., y0:y0 + ch, x0:x0 + cw]
