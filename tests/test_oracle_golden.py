"""The numpy oracle (oracle/graph_oracle.py) against the golden vectors produced by importing the
reference's graph constructor (tests/golden/make_golden.py).  CPU only.

Bars: edge lists bit-exact in canonical (row, col) order -- and in the reference's native order for kNN
rows whose k nearest distances are distinct; degree exact; float features to 1e-12 relative (float64
fixtures) / exact after the float32 cast (float32 fixtures)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, golden_files
from oracle import graph_oracle as go

ALL_EDGE = ["point_pair_features", "spatial_euclidean_distance", "velocity_euclidean_distance",
            "relative_position", "relative_velocity"]
ALL_NODE = ["rcs", "time_index", "degree", "velocity_vector_length", "velocity_vector", "spatial_coordinates"]


def _load(name):
    d = np.load(os.path.join(GOLDEN, name))
    routine, k, r, mode, basis = [str(s) for s in d["meta"]]
    return d, routine, int(k), float(r), mode, basis


@pytest.mark.parametrize("name", golden_files("small_"))
def test_small_frames_all_features(name):
    d, routine, k, r, mode, basis = _load(name)
    X, V = d["X"], d["V"]
    dist_basis = X if basis == "X" else np.concatenate((X, V), axis=1)
    E = go.build_edges(dist_basis, routine, k=k, r=r)
    assert E.dtype == np.int32
    assert np.array_equal(go.canonical_edges(E), go.canonical_edges(d["E"]))
    assert np.array_equal(go.undirected_degree(E, X.shape[0]), d["degree"])
    assert np.array_equal(go.time_index(d["timestamp"]).reshape(-1), d["time_index"])
    # features are compared on the REFERENCE's edge order so rows line up
    with np.errstate(all="ignore"):
        ef = go.edge_features(X, V, d["E"], ALL_EDGE, mode)
    np.testing.assert_allclose(ef, d["E_feat"], rtol=1e-12, atol=1e-9, equal_nan=True)
    F = {"rcs": d["rcs"], "time_index": go.time_index(d["timestamp"])}
    xf = go.node_features(X, V, F, go.undirected_degree(E, X.shape[0]), ALL_NODE)
    np.testing.assert_allclose(xf, d["X_feat"], rtol=1e-14, atol=0)


@pytest.mark.parametrize("name", golden_files("small_n40_s2_knn_k10") + golden_files("small_n300_s3_knn_k10"))
def test_knn_native_row_order(name):
    """Within a row the reference lists neighbours by ascending distance (graph.py:61-63 on sklearn's CSR)."""
    d, routine, k, r, mode, basis = _load(name)
    X = d["X"] if basis == "X" else np.concatenate((d["X"], d["V"]), axis=1)
    E = go.knn_edges(X, k)
    ref = d["E"]
    # rows whose k nearest distances are all distinct have a pinned order
    diff = X[ref[:, 0]] - X[ref[:, 1]]
    d2 = (diff * diff).sum(1).reshape(-1, k)
    pinned = np.array([len(np.unique(row)) == k for row in d2])
    assert pinned.mean() > 0.5      # the n40 frame carries two exact duplicate points -> many tied rows
    mask = np.repeat(pinned, k)
    assert np.array_equal(E[mask], ref[mask])


@pytest.mark.parametrize("name", golden_files("rs3000_"))
def test_radarscenes_frame(name):
    d, routine, k, r, mode, basis = _load(name)
    X, V = d["X"], d["V"]
    E = go.build_edges(X, routine, k=k, r=r)
    assert np.array_equal(go.canonical_edges(E), go.canonical_edges(d["E"]))
    if routine == "knn":
        assert np.array_equal(E, d["E"])                      # native order too: no ties in this frame
    deg = go.undirected_degree(E, X.shape[0])
    assert np.array_equal(deg, d["degree"])
    if name.endswith("ppf.npz"):
        efeat, nfeat = ["point_pair_features"], ["rcs", "velocity_vector_length", "time_index", "degree"]
    else:
        efeat, nfeat = ["relative_position"], ["rcs", "velocity_vector", "time_index", "degree"]
    ef = go.edge_features(X, V, d["E"], efeat, mode).astype(np.float32)
    # float32 fixtures: identical after the cast except where the f64 value sits on a rounding boundary
    np.testing.assert_allclose(ef, d["E_feat"], rtol=2e-7, atol=1e-5)
    F = {"rcs": d["rcs"], "time_index": go.time_index(d["timestamp"])}
    xf = go.node_features(X, V, F, deg, nfeat).astype(np.float32)
    assert np.array_equal(xf, d["X_feat"])


def test_reference_known_answers():
    """test/test_graph_constructor.py:6-103 and test/test_preprocessor.py:207-257 of the reference."""
    p1, p2 = np.array([[1.0, 1.0]]), np.array([[3.0, 2.0]])
    v1, v2 = np.array([[0.0, 1.0]]), np.array([[1.0, 0.0]])
    res = [round(float(a[0]), 2) for a in go.point_pair_features(p1, p2, v1, v2, "directed")]
    assert res == [2.24, 90.0, 63.43, 26.57]
    res = [round(float(a[0]), 2) for a in go.point_pair_features(p1, p2, v1, np.zeros((1, 2)), "directed")]
    assert res == [2.24, 90.0, 63.43, 90.0]

    X = np.array([[1.0, 1.0], [3.0, 2.0]])
    V = np.array([[0.0, 1.0], [1.0, 0.0]])
    E = go.build_edges(X, "knn", k=1)
    assert np.array_equal(E, [[0, 1], [1, 0]])
    ef = go.edge_features(X, V, E, ALL_EDGE, "directed")
    assert np.round(ef[0], 2).tolist() == [2.24, 90, 63.43, 26.57, 2.24, 1.41, -2, -1, -1, 1]
    deg = go.undirected_degree(E, 2)
    assert deg.tolist() == [1, 1]
    F = {"rcs": np.array([[1.8], [2.6]]), "time_index": np.array([[100.0], [101.0]])}
    xf = go.node_features(X, V, F, deg, ALL_NODE)
    assert xf[1].tolist() == [2.6, 101, 1, 1, 1, 0, 3, 2]

    # test_preprocessor.py:207-230
    X = np.array([[1.0, 1.0], [3.0, 2.0], [5.0, 8.0]])
    g = go.build_frame_graph(X, np.ones_like(X), np.zeros((3, 1)), np.array([[100.0], [101.0], [102.0]]),
                             "knn", 1, 1, ["spatial_coordinates", "time_index"], ["spatial_euclidean_distance"],
                             "directed", "X")
    assert np.array_equal(g["E"], [[0, 1], [1, 0], [2, 1]])
    assert g["X_feat"][1].tolist() == [3, 2, 1]
    assert g["E_feat"][0, 0] == 5 ** 0.5
    # test_preprocessor.py:233-257: the distance basis changes the edges
    X = np.array([[1.0, 1.0], [2.0, 2.0], [10.0, 10.0]])
    V = np.ones_like(X)
    V[0, :] = 100
    gx = go.build_frame_graph(X, V, np.zeros((3, 1)), np.zeros((3, 1)), "knn", 1, 1, ["spatial_coordinates"],
                              ["spatial_euclidean_distance"], "directed", "X")
    gxv = go.build_frame_graph(X, V, np.zeros((3, 1)), np.zeros((3, 1)), "knn", 1, 1, ["spatial_coordinates"],
                               ["spatial_euclidean_distance"], "directed", "XV")
    assert np.array_equal(gx["E"], [[0, 1], [1, 0], [2, 1]])
    assert np.array_equal(gxv["E"], [[0, 1], [1, 2], [2, 1]])


def test_edge_cases():
    assert go.build_edges(np.zeros((1, 2)), "knn", k=1) is None           # graph.py:45
    assert go.build_edges(np.zeros((0, 2)), "radius", r=1) is None
    with pytest.raises(ValueError):
        go.knn_edges(np.random.rand(5, 2), 5)                             # sklearn: k >= N
    with pytest.raises(Exception, match="Invalid feature"):
        go.edge_features(np.zeros((2, 2)), np.zeros((2, 2)), np.array([[0, 1]]), ["nope"], "directed")
    E = go.radius_edges(np.array([[0.0, 0.0], [1.0, 0.0], [5.0, 5.0]]), 1.0)   # inclusive d <= r, isolated node
    assert np.array_equal(E, [[0, 1], [1, 0]])
    assert go.undirected_degree(E, 3).tolist() == [1, 1, 0]
