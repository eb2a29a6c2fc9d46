"""SparseConvTensor and the rulebook record kept in its indice_dict."""
import numpy as np
import torch

from .. import ops as _ops


import os as _os
_TRACE_PAIRS = _os.environ.get("DODA_TRACE_PAIRS", "0") == "1"


class IndiceData:
    """Rulebook of one indice_key.

    Behaves like spconv v1.2's 5-tuple `(outids, indices, indice_pairs, indice_pair_num,
    spatial_shape)` (unpacking and [i] indexing work), but the native record is the gather
    table(s) of include/doda_hip.h; the spconv-format pairs are materialised on first access.

    kind 'subm' : tbl = nbr int32 [27, M]
    kind 'down2': tbl = child int32 [8, M_out]; tbl_rev = par_off int32 [8, M_in]
    """

    def __init__(self, kind, outids, indices, spatial_shape, out_spatial_shape, tbl, tbl_rev=None):
        self.kind = kind
        self.outids = outids
        self.indices = indices
        self.spatial_shape = spatial_shape
        self.out_spatial_shape = out_spatial_shape
        self.tbl = tbl
        self.tbl_rev = tbl_rev
        self._pairs = None
        self._wpairs = None   # (pairs int32 [2,K,n_in] without the -1 fill, pair_num) for the weight gradient

    def _export(self):
        if self._pairs is None:
            n_in = self.indices.shape[0]
            if self.kind == "subm":
                self._pairs = _ops.rulebook_pairs(self.tbl, n_in, flip=True)
            else:
                self._pairs = _ops.rulebook_pairs(self.tbl_rev, n_in, flip=False)
        return self._pairs

    def wgrad_lists(self, inverse=False):
        """(pair_in [K,ld], pair_out [K,ld], pair_num [K], seg [K,nt]) of the pair-list weight gradient
        (doda_spconv_wgrad_multi's pair-list kernel): list o pairs the row of the conv INPUT with the row of the conv
        OUTPUT under offset o, in ascending input row; seg is the lists' per-256-row prefix.  Exported
        once per rulebook (doda_rulebook_pairs without the -1 fill)."""
        if self._wpairs is None:
            n_in = self.indices.shape[0]
            if _TRACE_PAIRS:
                import sys
                import traceback
                sys.stderr.write("[doda] lazy pair-list export: kind %s, %d rows, inverse %s, from %s\n" % (
                    self.kind, n_in, inverse, " <- ".join("%s:%d" % (f.name, f.lineno) for f in reversed(traceback.extract_stack(limit=6)[:-1]))))
            if self.kind == "subm":
                self._wpairs = _ops.rulebook_pairs(self.tbl, n_in, flip=True, pad=False, with_seg=True)
            else:
                self._wpairs = _ops.rulebook_pairs(self.tbl_rev, n_in, flip=False, pad=False, with_seg=True)
        pairs, num, seg = self._wpairs
        # (the inverse convolution reads the same lists with the operand roles swapped; the segment prefix
        # belongs to the ascending list, whichever operand it indexes)
        return (pairs[1], pairs[0], num, seg) if inverse else (pairs[0], pairs[1], num, seg)

    @property
    def indice_pairs(self):
        return self._export()[0]

    @property
    def indice_pair_num(self):
        return self._export()[1]

    def _as_tuple(self):
        return (self.outids, self.indices, self.indice_pairs, self.indice_pair_num,
                self.spatial_shape)

    def __iter__(self):
        return iter(self._as_tuple())

    def __len__(self):
        return 5

    def __getitem__(self, i):
        return self._as_tuple()[i]


class IndiceDict(dict):
    """`SparseConvTensor.indice_dict`: indice_key -> IndiceData, shared by every tensor derived from one
    network input.  `pack_gen` is the weight-pack generation of the forward pass that owns the dictionary
    (doda_amd.spconv.conv: packed weight copies are refreshed once per forward pass); it lives in a slot,
    not under a key, so code that walks the rulebooks never meets it."""
    __slots__ = ("pack_gen",)

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.pack_gen = None


class SparseConvTensor:
    """features [M,C] + indices int32 [M,4] (batch,x,y,z) + spatial_shape + batch_size.

    Mirrors spconv v1.2 SparseConvTensor as used at reference model/unet.py:94 and
    model/unet_block.py:33,89 (rw attributes features/indices/spatial_shape/batch_size,
    indice_dict shared along the network, grid carried but unused by the hash builders)."""

    def __init__(self, features, indices, spatial_shape, batch_size, grid=None):
        self.features = features
        self.indices = indices
        # (a plain list of ints — what the conv modules pass along — skips the numpy round trip)
        self.spatial_shape = list(spatial_shape) if type(spatial_shape) is list else \
            [int(v) for v in np.asarray(spatial_shape).reshape(-1)]
        self.batch_size = int(batch_size)
        self.indice_dict = IndiceDict()
        self.grid = grid

    @property
    def spatial_size(self):
        return int(np.prod(self.spatial_shape))

    def find_indice_pair(self, key):
        if key is None:
            return None
        return self.indice_dict.get(key)

    def dense(self, channels_first=True):
        idx = self.indices.long()
        shape = [self.batch_size] + list(self.spatial_shape) + [self.features.shape[1]]
        out = torch.zeros(shape, dtype=self.features.dtype, device=self.features.device)
        out[idx[:, 0], idx[:, 1], idx[:, 2], idx[:, 3]] = self.features
        if not channels_first:
            return out
        ndim = len(self.spatial_shape)
        return out.permute(0, ndim + 1, *range(1, ndim + 1)).contiguous()

    @property
    def sparity(self):
        return self.indices.shape[0] / max(self.spatial_size * self.batch_size, 1)
