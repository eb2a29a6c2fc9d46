"""CPU ORACLE — TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py cpu_baseline).

Python face of oracle/doda_oracle.c (integer / ordering work, via ctypes) plus the torch-CPU
restatement of spconv v1.2's indiceConv / indiceConvBackward (per-offset gather -> mm ->
scatter-add; spconv is an un-vendored third-party dependency of the reference, see the header of
doda_oracle.c for the pinning status).  Nothing under doda_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "doda_oracle.c")
_LIB = os.path.join(_HERE, "liboracle.so")
_lib = None


def build_oracle(force=False):
    """gcc the C restatement into oracle/liboracle.so (git-ignored, travels to the GPU box)."""
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(_SRC):
        subprocess.run(["gcc", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-std=c99",
                        _SRC, "-o", _LIB], check=True)
    return _LIB


def _l():
    global _lib
    if _lib is None:
        build_oracle()
        _lib = C.CDLL(_LIB)
        _lib.orc_ballquery.restype = C.c_int32
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# ---------------------------------------------------------------------------------------------
def voxelize_idx(coords, mode=4):
    """coords int64 [N,3|4] -> (output_coords int64 [M,ncol], input_map int32 [N],
    output_map int32 [M,1+maxActive]); voxelize.cpp:10-155."""
    coords = np.ascontiguousarray(coords, dtype=np.int64)
    n, ncol = coords.shape
    input_map = np.zeros(n, dtype=np.int32)
    h, na, ma = C.c_void_p(), C.c_int32(), C.c_int32()
    rc = _l().orc_voxelize_idx_begin(_ptr(coords), C.c_int32(n), C.c_int32(ncol), C.c_int32(mode),
                                     _ptr(input_map), C.byref(h), C.byref(na), C.byref(ma))
    assert rc == 0
    out_coords = np.zeros((na.value, ncol), dtype=np.int64)
    out_map = np.zeros((na.value, ma.value + 1), dtype=np.int32)
    _l().orc_voxelize_idx_finish(h, _ptr(coords), _ptr(out_coords), _ptr(out_map))
    return out_coords, input_map, out_map


def voxelize_fp(feats, rules, average=True, out=None):
    feats, rules = _f32(feats), _i32(rules)
    m, w = rules.shape
    c = feats.shape[1]
    out = np.zeros((m, c), dtype=np.float32) if out is None else _f32(out).copy()
    _l().orc_voxelize_fp(_ptr(feats), _ptr(out), _ptr(rules), C.c_int32(m), C.c_int32(w - 1),
                         C.c_int32(c), C.c_int32(int(average)))
    return out


def voxelize_bp(d_out, rules, n_points, average=True, d_feats=None):
    d_out, rules = _f32(d_out), _i32(rules)
    m, w = rules.shape
    c = d_out.shape[1]
    d_feats = np.zeros((n_points, c), dtype=np.float32) if d_feats is None else _f32(d_feats).copy()
    _l().orc_voxelize_bp(_ptr(d_out), _ptr(d_feats), _ptr(rules), C.c_int32(m), C.c_int32(w - 1),
                         C.c_int32(c), C.c_int32(int(average)))
    return d_feats


# ---------------------------------------------------------------------------------------------
def _arr3(v):
    v = [int(x) for x in (v if isinstance(v, (list, tuple, np.ndarray)) else [v] * 3)]
    return (C.c_int32 * 3)(*v), v


def indice_pairs_subm(indices, batch_size, spatial_shape, ksize=3):
    """spconv getIndicePairsSubM (CPU).  -> (pairs int32 [2,K,M] -1 padded, pair_num int32 [K])."""
    indices = _i32(indices)
    m = indices.shape[0]
    k_c, k = _arr3(ksize)
    s_c, _ = _arr3(spatial_shape)
    K = k[0] * k[1] * k[2]
    pairs = np.full((2, K, max(m, 1)), -1, dtype=np.int32)
    pair_num = np.zeros(K, dtype=np.int32)
    rc = _l().orc_indice_pairs_subm(_ptr(indices), C.c_int32(m), C.c_int32(batch_size), s_c, k_c,
                                    _ptr(pairs), _ptr(pair_num))
    assert rc == 0
    return pairs[:, :, :m] if m else pairs[:, :, :0], pair_num


def indice_pairs_conv(indices, batch_size, spatial_shape, ksize=2, stride=2, padding=0, dilation=1):
    """spconv getIndicePairsConv (CPU).  -> (out_indices int32 [M_out,4], pairs [2,K,M],
    pair_num [K], out_shape list)."""
    indices = _i32(indices)
    m = indices.shape[0]
    k_c, k = _arr3(ksize)
    s_c, _ = _arr3(stride)
    p_c, _ = _arr3(padding)
    d_c, _ = _arr3(dilation)
    sh_c, _ = _arr3(spatial_shape)
    K = k[0] * k[1] * k[2]
    out_shape = (C.c_int32 * 3)()
    out_indices = np.zeros((max(m, 1) * K, 4), dtype=np.int32)
    pairs = np.full((2, K, max(m, 1)), -1, dtype=np.int32)
    pair_num = np.zeros(K, dtype=np.int32)
    n_out = C.c_int32()
    rc = _l().orc_indice_pairs_conv(_ptr(indices), C.c_int32(m), C.c_int32(batch_size), sh_c, k_c,
                                    s_c, p_c, d_c, out_shape, _ptr(out_indices), _ptr(pairs),
                                    _ptr(pair_num), C.byref(n_out))
    assert rc == 0
    return (out_indices[:n_out.value].copy(), pairs[:, :, :m] if m else pairs[:, :, :0], pair_num,
            [int(v) for v in out_shape])


# ---------------------------------------------------------------------------------------------
# spconv v1.2 indiceConv / indiceConvBackward (src/spconv/spconv_ops.cc), torch-CPU restatement:
# output = zeros; SubM centre (= argmax pairNum) as one full mm; every other non-empty offset:
# gather rows pairs[inverse][o] -> mm with W[o] -> scatter-add into rows pairs[!inverse][o].
# ---------------------------------------------------------------------------------------------
def indice_conv(features, filters, pairs, pair_num, num_act_out, inverse=False, subm=False):
    features = torch.as_tensor(features)
    filters = torch.as_tensor(filters)
    K = int(np.prod(filters.shape[:-2]))
    cin, cout = filters.shape[-2], filters.shape[-1]
    w = filters.reshape(K, cin, cout)
    pairs = torch.as_tensor(np.asarray(pairs)).long()
    pn = [int(v) for v in np.asarray(pair_num)]
    out = torch.zeros((num_act_out, cout), dtype=features.dtype)
    centre = int(np.argmax(pn)) if subm else -1
    if subm:
        out = torch.mm(features, w[centre])
    inv = int(bool(inverse))
    for o in range(K):
        n_hot = pn[o]
        if n_hot <= 0 or (subm and o == centre):
            continue
        gathered = features[pairs[inv, o, :n_hot]]
        out.index_add_(0, pairs[1 - inv, o, :n_hot], torch.mm(gathered, w[o]))
    return out


def indice_conv_backward(features, filters, out_bp, pairs, pair_num, inverse=False, subm=False):
    features = torch.as_tensor(features)
    filters = torch.as_tensor(filters)
    out_bp = torch.as_tensor(out_bp)
    K = int(np.prod(filters.shape[:-2]))
    cin, cout = filters.shape[-2], filters.shape[-1]
    w = filters.reshape(K, cin, cout)
    pairs = torch.as_tensor(np.asarray(pairs)).long()
    pn = [int(v) for v in np.asarray(pair_num)]
    d_in = torch.zeros_like(features)
    d_w = torch.zeros_like(w)
    centre = int(np.argmax(pn)) if subm else -1
    if subm:
        d_w[centre] = torch.mm(features.t(), out_bp)
        d_in = torch.mm(out_bp, w[centre].t())
    inv = int(bool(inverse))
    for o in range(K):
        n_hot = pn[o]
        if n_hot <= 0 or (subm and o == centre):
            continue
        in_rows, out_rows = pairs[inv, o, :n_hot], pairs[1 - inv, o, :n_hot]
        xg, dyg = features[in_rows], out_bp[out_rows]
        d_w[o] = torch.mm(xg.t(), dyg)
        d_in.index_add_(0, in_rows, torch.mm(dyg, w[o].t()))
    return d_in, d_w.reshape(filters.shape)


def indice_maxpool(features, pairs, pair_num, num_act_out):
    """spconv indice_maxpool forward (out starts at 0; out = max(out, in) over pairs)."""
    features = torch.as_tensor(features)
    pairs = torch.as_tensor(np.asarray(pairs)).long()
    out = torch.zeros((num_act_out, features.shape[1]), dtype=features.dtype)
    for o in range(pairs.shape[1]):
        n_hot = int(pair_num[o])
        for p in range(n_hot):
            i, t = int(pairs[0, o, p]), int(pairs[1, o, p])
            out[t] = torch.maximum(out[t], features[i])
    return out


def indice_maxpool_backward(features, out_features, dout, pairs, pair_num):
    """spconv indice_maxpool backward (maxPoolBwd): every pair whose input equals the pooled output
    receives that output's gradient: din[i, c] += dout[t, c] where features[i, c] == out[t, c]."""
    features, out_features, dout = (torch.as_tensor(a) for a in (features, out_features, dout))
    pairs = torch.as_tensor(np.asarray(pairs)).long()
    din = torch.zeros_like(features)
    for o in range(pairs.shape[1]):
        n_hot = int(pair_num[o])
        if n_hot == 0:
            continue
        i, t = pairs[0, o, :n_hot], pairs[1, o, :n_hot]
        hit = features[i] == out_features[t]
        din.index_add_(0, i, torch.where(hit, dout[t], torch.zeros_like(dout[t])))
    return din


# ---------------------------------------------------------------------------------------------
def knnquery(nsample, xyz, new_xyz, offset_ends, new_offset_ends):
    xyz, new_xyz = _f32(xyz), _f32(new_xyz)
    off, noff = _i32(offset_ends), _i32(new_offset_ends)
    m = new_xyz.shape[0]
    idx = np.zeros((m, nsample), dtype=np.int32)
    d2 = np.zeros((m, nsample), dtype=np.float32)
    _l().orc_knnquery(C.c_int32(m), C.c_int32(nsample), _ptr(xyz), _ptr(new_xyz), _ptr(off),
                      _ptr(noff), _ptr(idx), _ptr(d2))
    return idx, d2


def knn_batch(xyz, query_xyz, batch_idxs, query_batch_offsets, k):
    xyz, query_xyz = _f32(xyz), _f32(query_xyz)
    bi, qo = _i32(batch_idxs), _i32(query_batch_offsets)
    n = xyz.shape[0]
    idx = np.zeros((n, k), dtype=np.int32)
    _l().orc_knn_batch(C.c_int32(n), C.c_int32(k), _ptr(xyz), _ptr(query_xyz), _ptr(bi), _ptr(qo),
                       _ptr(idx))
    return idx


def ballquery(xyz, batch_idxs, batch_offsets, radius, mean_active):
    xyz = _f32(xyz)
    bi, bo = _i32(batch_idxs), _i32(batch_offsets)
    n = xyz.shape[0]
    idx = np.zeros(max(n * mean_active, 1), dtype=np.int32)
    start_len = np.zeros((n, 2), dtype=np.int32)
    total = _l().orc_ballquery(C.c_int32(n), C.c_int32(mean_active), C.c_float(radius), _ptr(xyz),
                               _ptr(bi), _ptr(bo), _ptr(idx), _ptr(start_len))
    return idx, start_len, int(total)
