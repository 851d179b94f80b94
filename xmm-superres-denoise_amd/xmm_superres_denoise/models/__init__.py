from .model import Model  # noqa: F401
from .modules.generator_rrdb import GeneratorRRDB_DN, GeneratorRRDB_SR  # noqa: F401
