"""`spconv`-v1.2-shaped operator surface backed by libdoda_hip.so.

DODA's model files use exactly: spconv.SparseConvTensor, spconv.SparseSequential,
spconv.SubMConv3d, spconv.SparseConv3d, spconv.SparseInverseConv3d and
spconv.modules.SparseModule (reference model/unet.py:3,35-36,42,94; model/unet_block.py:3-4,
15-29,33,45-48,62-85,89).  Names, constructor arguments, weight layout ([k,k,k,Cin,Cout]) and
forward behaviour follow spconv v1.2 so those files run unchanged.
"""
from .core import IndiceData, SparseConvTensor
from .modules import SparseModule, SparseSequential
from .conv import (SparseConv3d, SparseConvolution, SparseInverseConv3d, SubMConv3d)
from .pool import SparseMaxPool3d
from . import functional, modules, ops

__version__ = "1.2.1+doda_amd"

__all__ = ["IndiceData", "SparseConvTensor", "SparseModule", "SparseSequential",
           "SparseConvolution", "SubMConv3d", "SparseConv3d", "SparseInverseConv3d",
           "SparseMaxPool3d", "functional", "modules", "ops"]
