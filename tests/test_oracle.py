"""CPU tests of the oracle itself: against the known answers recorded from the reference's code
(tests/golden/known_answers.json), against hand-derivable cases, and against the dense conv3d
definition.  These pin the checker before the GPU tests trust it."""
import json
import os

import numpy as np
import torch

from tests.util import random_voxels, surface_voxels

KA = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "known_answers.json")))


def test_voxelize_known_answer(oracle):
    v = KA["voxelize"]
    oc, im, om = oracle.voxelize_idx(np.array(v["coords"], dtype=np.int64), v["mode"])
    assert oc.tolist() == v["voxel_coords"] and im.tolist() == v["p2v"] and om.tolist() == v["v2p"]
    mean = oracle.voxelize_fp(np.array(v["feats"], dtype=np.float32), om, True)
    assert np.allclose(mean, np.array(v["voxel_feats_mean"]), atol=1e-4)


def test_voxelize_modes_hand_derived(oracle):
    c = np.array([[5, 5, 5], [1, 1, 1], [5, 5, 5], [1, 1, 1], [5, 5, 5]], dtype=np.int64)
    oc, im, om = oracle.voxelize_idx(c, 1)   # code: front() = first point
    assert om.tolist() == [[1, 0], [1, 1]] and im.tolist() == [0, 1, 0, 1, 0]
    oc, im, om = oracle.voxelize_idx(c, 2)   # code: back() = last point
    assert om.tolist() == [[1, 4], [1, 3]]
    oc, im, om = oracle.voxelize_idx(c, 3)
    assert om.tolist() == [[3, 0, 2, 4], [2, 1, 3, -1]]
    s = oracle.voxelize_fp(np.arange(10, dtype=np.float32).reshape(5, 2), om, False)
    assert s.tolist() == [[0 + 4 + 8, 1 + 5 + 9], [2 + 6, 3 + 7]]
    # int64 coordinates are narrowed to int32 before comparison (voxelize.cpp:73,90)
    big = np.array([[1 << 32, 0, 0], [0, 0, 0]], dtype=np.int64)
    assert oracle.voxelize_idx(big, 4)[1].tolist() == [0, 0]


def test_voxelize_fp_rounds_product_before_sum(oracle):
    """Appendix D quirk: (1/n)*x is rounded per term, not sum/n."""
    f = np.array([[1.0], [1.0], [1.0]], dtype=np.float32)
    rules = np.array([[3, 0, 1, 2]], dtype=np.int32)
    third = np.float32(1) / np.float32(3)
    expect = np.float32(np.float32(third + third) + third)
    assert oracle.voxelize_fp(f, rules, True)[0, 0] == expect


def test_knnquery_known_answer(oracle):
    k = KA["knnquery"]
    idx, d2 = oracle.knnquery(k["k"], np.array(k["xyz"], np.float32), np.array(k["queries"], np.float32),
                              [5], [2])
    assert idx.tolist() == k["idx"] and np.allclose(d2, np.array(k["dist2"]))


def test_knn_batch_and_ballquery_hand_derived(oracle):
    xyz = np.array([[0, 0, 0], [1, 0, 0], [2, 0, 0], [10, 0, 0], [11, 0, 0]], np.float32)
    bi = np.array([0, 0, 0, 1, 1], np.int32)
    off = np.array([0, 3, 5], np.int32)
    assert oracle.knn_batch(xyz, xyz, bi, off, 2).tolist() == [[0, 1], [1, 0], [2, 1], [3, 4], [4, 3]]
    idx, sl, total = oracle.ballquery(xyz, bi, off, 1.5, 3)
    assert total == 11 and sl.tolist() == [[0, 2], [2, 3], [5, 2], [7, 2], [9, 2]]
    assert idx[:11].tolist() == [0, 1, 0, 1, 2, 1, 2, 3, 4, 3, 4]


def test_rulebook_subm_tiny_hand_derived(oracle):
    # two voxels adjacent along z: offsets 12 (k=(1,1,0)), 13 (centre), 14 (k=(1,1,2))
    idx = np.array([[0, 1, 1, 1], [0, 1, 1, 2]], np.int32)
    pairs, pn = oracle.indice_pairs_subm(idx, 1, [4, 4, 4], 3)
    assert pn.tolist() == [0] * 12 + [1, 2, 1] + [0] * 12
    # offset k: out = in + 1 - k  (k2=0 -> out z+1): input 0 -> output 1
    assert pairs[:, 12, 0].tolist() == [0, 1] and pairs[:, 14, 0].tolist() == [1, 0]
    assert pairs[0, 13, :2].tolist() == [0, 1] and pairs[1, 13, :2].tolist() == [0, 1]


def test_rulebook_down2_first_touch_order(oracle):
    idx = np.array([[0, 3, 3, 3], [0, 0, 0, 0], [0, 2, 2, 2], [0, 1, 0, 1]], np.int32)
    oi, pairs, pn, oshape = oracle.indice_pairs_conv(idx, 1, [4, 4, 4], 2, 2, 0, 1)
    assert oshape == [2, 2, 2]
    assert oi.tolist() == [[0, 1, 1, 1], [0, 0, 0, 0]]          # numbered by first touching input
    assert pn.tolist() == [2, 0, 0, 0, 0, 1, 0, 1]               # offsets 0 (x2), 5, 7
    assert pairs[:, 0, :2].tolist() == [[1, 2], [1, 0]]


def test_convs_match_dense_definition(oracle):
    from oracle import dense_ref
    rng = np.random.default_rng(0)
    for shape, gen in (([12, 11, 13], random_voxels), ([16, 16, 9], surface_voxels)):
        B = 2
        idx = gen(1, 300, B, shape)
        n = idx.shape[0]
        x = torch.tensor(rng.standard_normal((n, 5)), requires_grad=True)
        w = torch.tensor(rng.standard_normal((3, 3, 3, 5, 7)), requires_grad=True)
        pairs, pn = oracle.indice_pairs_subm(idx, B, shape, 3)
        y = oracle.indice_conv(x.detach(), w.detach(), pairs, pn, n, False, True)
        yd = dense_ref.subm_conv(x, idx, shape, B, w)
        assert (y - yd).abs().max() < 1e-12
        g = torch.tensor(rng.standard_normal(tuple(y.shape)))
        yd.backward(g)
        dx, dw = oracle.indice_conv_backward(x.detach(), w.detach(), g, pairs, pn, False, True)
        assert (dx - x.grad).abs().max() < 1e-12 and (dw - w.grad).abs().max() < 1e-11
        # strided + inverse
        oi, p2, pn2, osh = oracle.indice_pairs_conv(idx, B, shape, 2, 2, 0, 1)
        sites, osh2 = dense_ref.down2_sites(idx, shape, B)
        assert osh == osh2 and oi.shape[0] == sites.shape[0]
        assert sorted(map(tuple, oi.tolist())) == sorted(map(tuple, sites.tolist()))
        w2 = torch.tensor(rng.standard_normal((2, 2, 2, 5, 7)))
        y2 = oracle.indice_conv(x.detach(), w2, p2, pn2, oi.shape[0], False, False)
        assert (y2 - dense_ref.down2_conv(x.detach(), idx, shape, B, w2, oi)).abs().max() < 1e-12
        w3 = torch.tensor(rng.standard_normal((2, 2, 2, 7, 5)))
        y3 = oracle.indice_conv(y2, w3, p2, pn2, n, True, False)
        assert (y3 - dense_ref.inverse_conv(y2, oi, osh, B, w3, idx, shape)).abs().max() < 1e-12
