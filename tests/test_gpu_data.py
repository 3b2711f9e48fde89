"""Device-side batching (csrc/collate.hip, radargnn_amd/data.py) against the numpy oracle: bit-exact."""
import numpy as np
import pytest
import torch

from oracle import collate_oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def D():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from radargnn_amd import data
    return data


def make_graphs(D, sizes, seed=0, dn=5, de=2, extra=False):
    g = torch.Generator().manual_seed(seed)
    graphs = []
    for n, e in sizes:
        ei = torch.randint(0, n, (2, e), generator=g) if n and e else torch.zeros((2, 0), dtype=torch.long)
        kw = {}
        if extra:
            kw = dict(stamp=torch.randn(n, generator=g, dtype=torch.float64),               # 8-byte rows, 1-D
                      edge_flag=torch.randint(0, 9, (ei.shape[1], 3), generator=g, dtype=torch.int32),
                      meta=torch.randn(2, 4, generator=g))                                   # 2 rows per graph: its own offsets
        graphs.append(D.Data(x=torch.randn(n, dn, generator=g), edge_index=ei, edge_attr=torch.randn(ei.shape[1], de, generator=g),
                             y=torch.randn(n, 6, generator=g), pos=torch.randn(n, 2, generator=g),
                             vel=torch.randn(n, 2, generator=g), **kw))
    return graphs


def as_numpy(d):
    return {k: v.numpy() for k, v in d.items()}


def check(batch, graphs, ids):
    exp = collate_oracle.collate([as_numpy(graphs[i]) for i in ids])
    for k, v in exp.items():
        got = getattr(batch, k).cpu().numpy()
        assert got.dtype == v.dtype and got.shape == v.shape, (k, got.dtype, v.dtype, got.shape, v.shape)
        assert np.array_equal(got, v), k
    assert batch.num_graphs == len(ids)


@pytest.mark.parametrize("sizes", [[(3, 4)], [(3, 4), (5, 9), (2, 1)], [(1, 0), (4, 12), (0, 0), (7, 3), (0, 0)],
                                   [(300, 2000)] * 9, [(n, 4 * n) for n in range(1, 40)]])
def test_collate_matches_oracle(D, sizes):
    graphs = make_graphs(D, sizes, seed=len(sizes))
    store = D.GraphStore(graphs)
    check(store.collate(range(len(graphs))), graphs, list(range(len(graphs))))
    perm = torch.randperm(len(graphs), generator=torch.Generator().manual_seed(1)).tolist()
    check(store.collate(perm), graphs, perm)                                                 # any order, straight from HBM
    rep = [0] * 3 + perm[:2]
    check(store.collate(rep), graphs, rep)                                                   # a graph may appear twice


def test_collate_other_dtypes_and_own_row_counts(D):
    graphs = make_graphs(D, [(6, 10), (3, 0), (9, 30), (0, 0)], seed=7, extra=True)
    store = D.GraphStore(graphs)
    assert store.kind["stamp"] == "node" and store.kind["edge_flag"] == "edge" and store.kind["meta"] == "own"
    check(store.collate([2, 0, 3, 1]), graphs, [2, 0, 3, 1])


def test_empty_batch_and_bad_ids(D):
    graphs = make_graphs(D, [(3, 4), (2, 2)])
    store = D.GraphStore(graphs)
    b = store.collate([])
    assert b.x.shape == (0, 5) and b.edge_index.shape == (2, 0) and b.ptr.tolist() == [0] and b.num_graphs == 0
    with pytest.raises(IndexError):
        store.collate([2])
    with pytest.raises(NotImplementedError):
        D.GraphStore([D.Data(x=torch.randn(3, 2), edge_index=torch.zeros((2, 0), dtype=torch.long),
                             flag=torch.zeros(3, dtype=torch.bool))])


def test_loader_batches_and_roundtrip(D):
    graphs = make_graphs(D, [(10 + i, 30 + i) for i in range(11)], seed=3)
    loader = D.DataLoader(graphs, batch_size=4, shuffle=False)
    seen = 0
    for b, ids in zip(loader, loader.batch_ids()):
        assert b.x.is_cuda and b.to("cuda") is b
        check(b, graphs, list(map(int, ids)))
        for part, i in zip(b.to_data_list(), ids):
            for k in graphs[i].keys:
                assert torch.equal(part[k].cpu(), graphs[i][k]), k
        seen += b.num_graphs
    assert seen == 11
    sh = D.DataLoader(graphs, batch_size=5, shuffle=True, generator=torch.Generator().manual_seed(9))
    order = torch.randperm(11, generator=torch.Generator().manual_seed(9)).tolist()
    for j, b in enumerate(sh):
        check(b, graphs, order[5 * j:5 * j + 5])


def test_loader_feeds_the_model_like_a_host_collated_batch(D):
    """The batch from HBM and a host-side torch.cat batch give the same forward (same tensors -> same kernels)."""
    from radargnn_amd import gnn
    cfg = gnn.GNNArchitectureConfig(node_feature_dimension=5, edge_feature_dimension=2, conv_layer_dimensions=[16, 8],
                                    classification_head_layer_dimensions=[6], regression_head_layer_dimensions=[8, 5],
                                    initial_node_feature_embedding=True, initial_edge_feature_embedding=True,
                                    node_feature_embedding_layer_dimensions=[8, 16], edge_feature_embedding_layer_dimensions=[4, 8],
                                    conv_layer_type="MPNNConv", batch_norm_in_mlps=False)
    torch.manual_seed(0)
    model = gnn.DetNetBasic(cfg).cuda()
    graphs = make_graphs(D, [(40, 200), (25, 90), (60, 400)], seed=11)
    b = next(iter(D.DataLoader(graphs, batch_size=3)))
    exp = collate_oracle.collate([as_numpy(g) for g in graphs])
    with torch.no_grad():
        c1, bb1 = model(b.x, b.edge_index, b.edge_attr)
        c2, bb2 = model(torch.from_numpy(exp["x"]).cuda(), torch.from_numpy(exp["edge_index"]).cuda(),
                        torch.from_numpy(exp["edge_attr"]).cuda())
    assert torch.equal(c1, c2) and torch.equal(bb1, bb2)
