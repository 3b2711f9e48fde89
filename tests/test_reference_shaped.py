"""The reference-shaped CPU timing path (oracle/reference_shaped.py: sklearn + networkx + per-edge loop) must agree
with the vectorised oracle, otherwise its time would not be the time of the same computation."""
import numpy as np
import pytest

from oracle import graph_oracle as go
from oracle import reference_shaped as rs
from radargnn_amd import synthetic

sklearn = pytest.importorskip("sklearn")
pytest.importorskip("networkx")


@pytest.mark.parametrize("algo,k,r", [("knn", 5, None), ("radius", None, 6.0)])
def test_reference_shaped_graph_equals_oracle(algo, k, r):
    f = synthetic.nuscenes_frame(1)
    nf = ["rcs", "velocity_vector", "time_index", "degree"]
    ef = ["relative_position", "spatial_euclidean_distance"]
    a = rs.build_frame_graph(f.X, f.V, f.rcs, f.timestamp, algo, k, r, nf, ef, "directed")
    b = go.build_frame_graph(f.X, f.V, f.rcs, f.timestamp, algo, k, r, nf, ef, "directed")
    oa = np.lexsort((a["E"][:, 1], a["E"][:, 0]))
    ob = np.lexsort((b["E"][:, 1], b["E"][:, 0]))
    assert np.array_equal(a["E"][oa], b["E"][ob])
    assert np.array_equal(a["x"], b["x"])
    np.testing.assert_allclose(a["edge_attr"][oa], b["edge_attr"][ob], rtol=1e-6)
