"""Parameter containers with the reference's module tree, so state_dict keys and default initialisation match
`rrdb.{i}.RDB{r}.conv{c}.{weight,bias}` (reference models/modules/rrdb_blocks.py:10-14,22-32,57-64).

These modules own the OIHW fp32 nn.Parameters only.  They are never called: the dense-block arithmetic
(rrdb_blocks.py:37-54,66-70) runs inside libxsd_hip.so on 32-channel NHWC planes, where the torch.cat is free.
"""
from torch import nn


def make_layer(block, n_layers):
    return nn.Sequential(*[block() for _ in range(n_layers)])


class ResidualDenseBlock_5C(nn.Module):
    def __init__(self, nf=64, gc=32, bias=True, memory_efficient: bool = False):
        super().__init__()
        self.mem_efficient = memory_efficient
        for c in range(5):
            setattr(self, f"conv{c + 1}", nn.Conv2d(nf + c * gc, gc if c < 4 else nf, 3, 1, 1, bias=bias))

    def forward(self, x):
        raise RuntimeError("ResidualDenseBlock_5C is a parameter container; call the generator (HIP engine) instead")


class RRDB(nn.Module):
    def __init__(self, nf, gc=32, memory_efficient: bool = False):
        super().__init__()
        self.RDB1 = ResidualDenseBlock_5C(nf, gc, memory_efficient=memory_efficient)
        self.RDB2 = ResidualDenseBlock_5C(nf, gc, memory_efficient=memory_efficient)
        self.RDB3 = ResidualDenseBlock_5C(nf, gc, memory_efficient=memory_efficient)

    def forward(self, x):
        raise RuntimeError("RRDB is a parameter container; call the generator (HIP engine) instead")
