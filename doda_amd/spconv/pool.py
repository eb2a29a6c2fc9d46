"""SparseMaxPool3d (spconv v1.2 `spconv.pool`): indice-pair max pooling.  Not used by DODA's
network (it down-samples with strided convolutions, model/unet_block.py:70); provided because
north_star names indice-pair pooling.  Geometry: kernel 2, stride 2, padding 0."""
from . import functional as Fsp
from . import ops
from .core import SparseConvTensor
from .modules import SparseModule


class SparseMaxPool3d(SparseModule):
    def __init__(self, kernel_size, stride=1, padding=0, dilation=1, indice_key=None):
        super().__init__()
        self.kernel_size = ops._triple(kernel_size)
        self.stride = ops._triple(stride)
        self.padding = ops._triple(padding)
        self.dilation = ops._triple(dilation)
        self.indice_key = indice_key

    def forward(self, input):
        assert isinstance(input, SparseConvTensor)
        data = input.find_indice_pair(self.indice_key)
        if data is None:
            data = ops.build_down2(input.indices, input.batch_size, input.spatial_shape,
                                   self.kernel_size, self.stride, self.padding, self.dilation)
            if self.indice_key is not None:
                input.indice_dict[self.indice_key] = data
        out_features = Fsp.indice_maxpool(input.features, data)
        out = SparseConvTensor(out_features, data.outids, data.out_spatial_shape, input.batch_size)
        out.indice_dict = input.indice_dict
        out.grid = input.grid
        return out
