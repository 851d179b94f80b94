from ._lib import XsdError, build, load  # noqa: F401
from .engine import Engine, STRETCH, compose_input, image_upsample, mask_pad_normalize, normalize  # noqa: F401
