"""Host logic of radargnn_amd.data: the Data / DataLoader surface, the file formats and the batching
oracle.  No GPU needed; the device collation itself is checked in tests/test_gpu_data.py."""
import json
import os
import pickle
import sys
import types

import numpy as np
import pytest
import torch

from oracle import collate_oracle
from radargnn_amd import data as D


def small_graph(n, e, seed):
    g = torch.Generator().manual_seed(seed)
    ei = torch.randint(0, max(n, 1), (2, e), generator=g) if n else torch.zeros((2, 0), dtype=torch.long)
    return D.Data(x=torch.randn(n, 5, generator=g), edge_index=ei, edge_attr=torch.randn(e, 2, generator=g),
                  y=torch.randn(n, 6, generator=g), pos=torch.randn(n, 2, generator=g), vel=torch.randn(n, 2, generator=g))


def test_data_surface_matches_what_the_reference_touches():
    d = small_graph(4, 7, 0)
    assert d.keys == ["x", "edge_index", "edge_attr", "y", "pos", "vel"]        # create_graph_data's keyword order
    assert d.num_nodes == 4 and d.num_edges == 7 and d.num_node_features == 5
    assert d["x"] is d.x and "vel" in d and "foo" not in d
    assert D.Data().x is None and D.Data().edge_index is None                     # unset standard attributes read as None
    with pytest.raises(AttributeError):
        d.nonexistent
    assert d.to("cpu") is d                                                       # inference.py:57 calls .to(device) in place
    assert "x=[4, 5]" in repr(d)


def test_create_graph_data_casts_like_the_reference():
    """dataset_creation.py:786-814: float32 features, int64 edge_index = E.T, y = [label | box]."""
    graph = types.SimpleNamespace(X_feat=np.arange(10, dtype=np.float64).reshape(2, 5), E=np.array([[0, 1], [1, 0]], dtype=np.int32),
                                  E_feat=np.array([[1.5, -2.0], [0.25, 3.0]]))
    pc = types.SimpleNamespace(X_cc=np.array([[1.0, 2.0], [3.0, 4.0]]), V_cc_compensated=np.array([[0.1, 0.2], [0.3, 0.4]]))
    d = D.create_graph_data(graph, np.array([[1.0], [5.0]]), np.arange(10, dtype=np.float64).reshape(2, 5), pc)
    assert d.x.dtype == torch.float32 and d.edge_attr.dtype == torch.float32 and d.y.dtype == torch.float32
    assert d.edge_index.dtype == torch.int64 and d.edge_index.tolist() == [[0, 1], [1, 0]]
    assert d.y.shape == (2, 6) and d.y[:, 0].tolist() == [1.0, 5.0]
    assert d.pos.dtype == torch.float32 and d.vel.shape == (2, 2)


def test_collate_oracle_hand_computed_example():
    """Two graphs (2 and 3 nodes): edge_index of the second is shifted by 2, batch = [0,0,1,1,1], ptr = [0,2,5]."""
    a = {"x": np.array([[1.0], [2.0]]), "edge_index": np.array([[0, 1], [1, 0]]), "edge_attr": np.array([[10.0], [11.0]])}
    b = {"x": np.array([[3.0], [4.0], [5.0]]), "edge_index": np.array([[0, 2, 1], [2, 0, 0]]), "edge_attr": np.array([[12.0], [13.0], [14.0]])}
    out = collate_oracle.collate([a, b])
    assert out["x"].ravel().tolist() == [1, 2, 3, 4, 5]
    assert out["edge_index"].tolist() == [[0, 1, 2, 4, 3], [1, 0, 4, 2, 2]]
    assert out["edge_attr"].ravel().tolist() == [10, 11, 12, 13, 14]
    assert out["batch"].tolist() == [0, 0, 1, 1, 1] and out["ptr"].tolist() == [0, 2, 5]


def test_loader_length_and_order():
    graphs = [small_graph(3, 2, i) for i in range(7)]
    ld = D.DataLoader(graphs, batch_size=3, shuffle=False)
    assert len(ld) == 3 and [list(map(int, b)) for b in ld.batch_ids()] == [[0, 1, 2], [3, 4, 5], [6]]
    assert len(D.DataLoader(graphs, batch_size=3, drop_last=True)) == 2
    g = torch.Generator().manual_seed(5)
    sh = D.DataLoader(graphs, batch_size=4, shuffle=True, generator=g)
    ids = np.concatenate(sh.batch_ids())
    assert sorted(ids.tolist()) == list(range(7))
    assert ids.tolist() == torch.randperm(7, generator=torch.Generator().manual_seed(5)).tolist()   # torch RandomSampler's draw
    with pytest.raises(ValueError):
        D.DataLoader(graphs, batch_size=0)


def test_store_refuses_cpu():
    with pytest.raises(RuntimeError, match="GPU device is required"):
        D.GraphStore([small_graph(3, 2, 0)], device="cpu")


def test_graph_file_roundtrip_and_data_loaders(tmp_path):
    root = tmp_path / "ds"
    (root / "train").mkdir(parents=True)
    (root / "validate").mkdir()
    graphs = [small_graph(3 + i, 4, i) for i in range(11)]
    for i, g in enumerate(graphs):
        D.save_graph(g, str(root / ("train" if i < 8 else "validate") / f"graph_{i:03d}.pt"))
    (root / "config.json").write_text(json.dumps({"GRAPH_CONSTRUCTION_SETTINGS": {"graph_construction_algorithm": "knn"}}))
    loaders, cfg = D.get_data_loaders(["train", "validate"], str(root), batch_size=4, shuffle=False)
    assert cfg["GRAPH_CONSTRUCTION_SETTINGS"]["graph_construction_algorithm"] == "knn"
    assert len(loaders["train"]) == 2 and len(loaders["validate"]) == 1
    back = loaders["train"].dataset[3]
    for k in graphs[3].keys:
        assert torch.equal(back[k], graphs[3][k])
    plain = torch.load(str(root / "train" / "graph_000.pt"))                     # readable without this package
    assert plain["format"] == D.FORMAT_TAG and torch.equal(plain["x"], graphs[0].x)


def _fake_pyg(monkeypatch):
    """Modules shaped like torch_geometric 2.1's ``Data`` / ``GlobalStorage`` (state layout: Data.__dict__ = {'_store':
    GlobalStorage}, GlobalStorage.__dict__ = {'_mapping': {...}, '_parent': Data}) so that a pickle with the real class
    PATHS can be written here; the reader must not need these modules."""
    pkg = types.ModuleType("torch_geometric"); sub = types.ModuleType("torch_geometric.data")
    m_data = types.ModuleType("torch_geometric.data.data"); m_storage = types.ModuleType("torch_geometric.data.storage")

    class GlobalStorage:
        pass

    class Data:
        pass

    GlobalStorage.__module__, GlobalStorage.__qualname__ = "torch_geometric.data.storage", "GlobalStorage"
    Data.__module__, Data.__qualname__ = "torch_geometric.data.data", "Data"
    m_data.Data, m_storage.GlobalStorage = Data, GlobalStorage
    for name, mod in (("torch_geometric", pkg), ("torch_geometric.data", sub), ("torch_geometric.data.data", m_data),
                      ("torch_geometric.data.storage", m_storage)):
        monkeypatch.setitem(sys.modules, name, mod)
    return Data, GlobalStorage


def test_reads_pickles_with_torch_geometric_class_paths(tmp_path, monkeypatch):
    g = small_graph(5, 6, 3)
    Data, GlobalStorage = _fake_pyg(monkeypatch)
    d2, st = Data(), GlobalStorage()
    st.__dict__["_mapping"] = {k: v for k, v in g.items()}
    st.__dict__["_parent"] = d2                                                   # BaseStorage.__getstate__ derefs the weakref
    d2.__dict__["_store"] = st
    p2 = str(tmp_path / "graph_pyg2.pt")
    torch.save(d2, p2)
    d1 = Data()                                                                   # torch_geometric 1.x kept attributes in __dict__
    d1.__dict__.update({k: v for k, v in g.items()})
    p1 = str(tmp_path / "graph_pyg1.pt")
    torch.save(d1, p1)
    for name in list(sys.modules):
        if name.split(".")[0] == "torch_geometric":
            monkeypatch.delitem(sys.modules, name)                                # the reader must work without the package
    for p in (p2, p1):
        back = D.load_graph(p)
        assert back.keys == g.keys
        for k in g.keys:
            assert torch.equal(back[k], g[k])


def test_writes_pickles_with_torch_geometric_class_paths(tmp_path):
    """save_graph(pyg_compatible=True): the file names torch_geometric's classes (what the reference's torch.load expects to
    find) and reads back through load_graph; no torch_geometric module is left behind in sys.modules."""
    import pickletools
    import zipfile
    g = small_graph(7, 9, 3)
    p = str(tmp_path / "graph_0.pt")
    D.save_graph(g, p, pyg_compatible=True)
    try:
        import torch_geometric  # noqa: F401
    except ImportError:
        assert not any(n.split(".")[0] == "torch_geometric" for n in sys.modules)
    with zipfile.ZipFile(p) as z:
        pkl = z.read([n for n in z.namelist() if n.endswith("data.pkl")][0])
    ops = "\n".join(str(a) for _, a, _ in pickletools.genops(pkl) if isinstance(a, str))
    assert "torch_geometric.data.data" in ops and "GlobalStorage" in ops and "_mapping" in ops
    back = D.load_graph(p)
    assert back.keys == g.keys
    for k in g.keys:
        assert torch.equal(back[k], g[k])


def test_pyg_compatible_file_round_trips_through_torch_geometric_if_present(tmp_path):
    tg = pytest.importorskip("torch_geometric")
    g = small_graph(7, 9, 3)
    p = str(tmp_path / "graph_0.pt")
    D.save_graph(g, p, pyg_compatible=True)
    back = torch.load(p, weights_only=False)
    assert isinstance(back, tg.data.Data)
    for k in g.keys:
        assert torch.equal(getattr(back, k), g[k])
