from .crop import Crop  # noqa: F401
from .imageupsample import ImageUpsample  # noqa: F401
from .normalize import Normalize  # noqa: F401
