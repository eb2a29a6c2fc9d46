"""Definitional oracle for the sparse convolutions — TEST INFRASTRUCTURE ONLY.

spconv v1.2 itself cannot run in the build image, so the *values* a sparse convolution must
produce are pinned by its definition on a dense grid with torch-CPU:

  SubM k3  : F.conv3d(x, W.permute(4,3,0,1,2), padding=1) sampled at the active input sites
  k2 s2    : F.conv3d(x, W.permute(4,3,0,1,2), stride=2) sampled where the 2x2x2 cell is occupied
  inverse  : F.conv_transpose3d(x_coarse, W.permute(3,4,0,1,2), stride=2) sampled at fine sites

(weights are [kD,kH,kW,Cin,Cout]; conv3d is a cross-correlation, as is spconv).  Gradients come
from torch autograd on the same expressions.  Use float64 for a tight reference.
"""
import torch
import torch.nn.functional as F


def scatter_dense(features, indices, spatial_shape, batch_size):
    """[M,C] + int [M,4] -> dense [B,C,X,Y,Z]."""
    idx = torch.as_tensor(indices).long()
    c = features.shape[1]
    dense = torch.zeros([batch_size, c] + [int(s) for s in spatial_shape], dtype=features.dtype)
    dense[idx[:, 0], :, idx[:, 1], idx[:, 2], idx[:, 3]] = features
    return dense


def sample(dense, indices):
    idx = torch.as_tensor(indices).long()
    return dense[idx[:, 0], :, idx[:, 1], idx[:, 2], idx[:, 3]]


def subm_conv(features, indices, spatial_shape, batch_size, weight):
    k = weight.shape[0]
    dense = scatter_dense(features, indices, spatial_shape, batch_size)
    out = F.conv3d(dense, weight.permute(4, 3, 0, 1, 2), padding=k // 2)
    return sample(out, indices)


def down2_conv(features, indices, spatial_shape, batch_size, weight, out_indices):
    dense = scatter_dense(features, indices, spatial_shape, batch_size)
    out = F.conv3d(dense, weight.permute(4, 3, 0, 1, 2), stride=2)
    return sample(out, out_indices)


def down2_sites(indices, spatial_shape, batch_size):
    """Set of occupied output cells of the k2s2 conv (unordered), as a sorted int tensor [*,4]."""
    idx = torch.as_tensor(indices).long()
    out_shape = [(int(s) - 2) // 2 + 1 for s in spatial_shape]
    q = torch.cat([idx[:, :1], idx[:, 1:] // 2], 1)
    keep = (q[:, 1] < out_shape[0]) & (q[:, 2] < out_shape[1]) & (q[:, 3] < out_shape[2])
    return torch.unique(q[keep], dim=0), out_shape


def inverse_conv(coarse_features, coarse_indices, coarse_shape, batch_size, weight, fine_indices,
                 fine_shape):
    dense = scatter_dense(coarse_features, coarse_indices, coarse_shape, batch_size)
    out = F.conv_transpose3d(dense, weight.permute(3, 4, 0, 1, 2), stride=2)
    # conv_transpose output extent is 2*coarse; pad up to the fine shape (odd fine extents)
    pad = []
    for d in (2, 1, 0):
        pad += [0, max(0, int(fine_shape[d]) - out.shape[2 + d])]
    out = F.pad(out, pad)
    return sample(out, fine_indices)
