from .rrdb_blocks import RRDB, ResidualDenseBlock_5C, make_layer  # noqa: F401
