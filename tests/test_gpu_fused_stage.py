"""The fused launches of round 3 against the launches they replace (same inputs, bit-equal outputs unless stated):
one-block-per-frame binning, the one-launch radius fill with its relative_position attributes, node features written by the
time-index kernel, row lists from the degrees, the kNN write-out's attributes / degree preset, the two-layer tiny MLP.
They replace kernels that stand for scikit-learn / scipy / networkx / numpy calls of graph_constructor/graph.py:52-96,139-275 and
preprocessor/radarscenes/dataset_creation.py:214-223; the replaced launches are themselves pinned against the reference-generated
fixtures (tests/test_gpu_graph.py)."""
import numpy as np
import pytest
import torch

from radargnn_amd import synthetic

pytestmark = pytest.mark.gpu


def _batch(frames):
    from radargnn_amd import frames as fr
    return fr.FrameBatch.from_frames(frames)


def _dense_cluster_frame(n, seed, spread):
    f = synthetic.small_frame(n, seed)
    rng = np.random.Generator(np.random.PCG64(seed))
    f.X[:] = rng.normal(0.0, spread, size=f.X.shape)               # n points within a few `spread`: rows of hundreds of neighbours
    return f


@pytest.mark.parametrize("r,frames", [(1.0, "radar"), (2.5, "mixed"), (1.0, "dense")])
def test_one_launch_radius_stage_equals_the_split_launches(r, frames):
    """rgnn_grid_build_frames (hint) + rgnn_radius_graph_rows against rgnn_grid_build + rgnn_radius_graph_fill +
    rgnn_edge_features: rowptr, col, edge_index and the float32 relative_position attributes bit-equal -- rows from the count
    pass's cache, dense rows ranked from LDS (49 ... 512 neighbours) and from the scratch (> 512)."""
    from radargnn_amd import ops
    if frames == "radar":
        fl = [synthetic.radarscenes_frame(i) for i in range(6)]
    elif frames == "mixed":
        fl = [synthetic.nuscenes_frame(i) for i in range(5)] + [synthetic.small_frame(7, 1), synthetic.radarscenes_frame(2)]
    else:
        fl = [_dense_cluster_frame(400, 3, 0.35), _dense_cluster_frame(900, 4, 0.25), synthetic.radarscenes_frame(1)]
    b = _batch(fl)
    biggest = int(b.frame_sizes.max())
    out = {}
    for fused in (False, True):
        ops.FUSED_RADIUS_ROWS = fused
        try:
            g, rowptr = ops.radius_graph_count(b.X, b.frame_ptr, r, max_frame_points=biggest if fused else 0)
            e = int(rowptr[-1].item())
            col, ei, rel = ops.radius_graph_fill(g, rowptr, r, e, relative_position="directed")
        finally:
            ops.FUSED_RADIUS_ROWS = True
        out[fused] = (rowptr.clone(), col.clone(), ei.clone(), rel.clone())
    for a, c in zip(out[False], out[True]):
        assert torch.equal(a, c)
    deg = (out[True][0][1:] - out[True][0][:-1])
    if frames == "dense":
        assert int(deg.max()) > 512 and int(((deg > 48) & (deg <= 512)).sum()) > 0        # all three row paths ran
    # ... and the cell order the fused build leaves in the workspace is a permutation with its inverse next to it
    order, rank = g.cell_order().long(), g.cell_rank().long()
    assert torch.equal(order[rank], torch.arange(b.num_points, device="cuda")) and torch.equal(rank[order], torch.arange(b.num_points, device="cuda"))


def test_node_features_from_the_time_index_kernel_and_row_lists_from_degrees():
    from radargnn_amd import frames as fr, ops
    fl = [synthetic.radarscenes_frame(i) for i in range(4)] + [synthetic.nuscenes_frame(1), synthetic.small_frame(5, 2)]
    b = _batch(fl)
    cfg = fr.GraphSettings(algorithm="radius", r=1.2)
    g = fr.build_graphs(b, cfg)                                    # (takes the fused launches)
    tidx, _ = ops.time_index(b.timestamp, b.frame_ptr)
    x_split = ops.node_features(b.X, b.V, b.rcs, tidx, g.degree, list(cfg.node_features), dtype=torch.float32)
    assert torch.equal(g.x, x_split)
    x64 = ops.node_features_time_index(b.X, b.V, b.rcs, b.timestamp, b.frame_ptr, g.degree, list(cfg.node_features), dtype=torch.float64)
    assert torch.equal(x64, ops.node_features(b.X, b.V, b.rcs, tidx, g.degree, list(cfg.node_features), dtype=torch.float64))
    # row lists: nodes with / without edges, ascending node ids, from the degrees of the symmetric graph
    assert g.split is not None
    n = b.num_points
    deg = g.degree.cpu().numpy()
    lst_e, cnt_e, slot, lst_ne, cnt_ne = [t.cpu().numpy() for t in g.split]
    ce, cne = int(cnt_e[0]), int(cnt_ne[0])
    assert np.array_equal(lst_ne[:cne], np.nonzero(deg > 0)[0]) and np.array_equal(lst_e[:ce], np.nonzero(deg == 0)[0]) and ce + cne == n
    exp_slot = np.full(n, -1, np.int32)
    exp_slot[deg == 0] = np.arange(ce)
    assert np.array_equal(slot, exp_slot)


@pytest.mark.parametrize("mode", ["directed", "undirected"])
def test_knn_write_out_attributes_and_degree_preset(mode):
    from radargnn_amd import ops
    fl = [synthetic.radarscenes_frame(i) for i in range(3)] + [synthetic.nuscenes_frame(0)]
    b = _batch(fl)
    k = 10
    nbr0, ei0, _ = ops.knn_graph(b.X, b.frame_ptr, k)
    nbr1, ei1, _, rel, deg0 = ops.knn_graph(b.X, b.frame_ptr, k, relative_position=mode, degree_init=True)
    assert torch.equal(nbr0, nbr1) and torch.equal(ei0, ei1)
    exp_rel, _ = ops.edge_features(b.X, b.V, ei0, ["relative_position"], mode, dtype=torch.float32)
    assert torch.equal(rel, exp_rel)
    n = b.num_points
    rowptr = torch.arange(0, n * k + 1, k, dtype=torch.int32, device="cuda")
    assert torch.equal(ops.undirected_degree_preset(rowptr, nbr1.reshape(-1), deg0), ops.undirected_degree(rowptr, nbr0.reshape(-1), n))


def test_tiny_mlp2_equals_gather_and_two_linear_layers():
    from radargnn_amd import ops
    g = torch.Generator().manual_seed(3)
    e = 50_000
    a = torch.randn(e, 2, generator=g).cuda()
    perm = torch.randperm(e, generator=g).int().cuda()
    w1, b1 = torch.randn(4, 2, generator=g).cuda(), torch.randn(4, generator=g).cuda()
    w2, b2 = torch.randn(8, 4, generator=g).cuda(), torch.randn(8, generator=g).cuda()
    exp = ops.linear(ops.linear(ops.gather_rows(a, perm), w1, b1, relu=True), w2, b2, relu=True)
    assert torch.equal(ops.tiny_mlp2(a, perm, w1, b1, True, w2, b2, True), exp)
    assert torch.equal(ops.tiny_mlp2(a, None, w1, None, False, w2, b2, True), ops.linear(ops.linear(a, w1), w2, b2, relu=True))


def test_own_edge_map_of_a_symmetric_graph_replaces_the_twin_search_for_antisymmetric_attributes(monkeypatch):
    """rgnn_csr_by_target_symmetric_own: same rows and sources as the build that searches every edge's twin, and for
    relative_position in directed mode the attributes in target order are EXACTLY minus those of the own edge at the slot -- so the
    HotPath's logits are the same bits with and without the search."""
    from radargnn_amd import frames as fr, ops
    from radargnn_amd.gnn import mpnn_layers
    from radargnn_amd.gnn.mpnn_layers import TargetCSR
    import bench
    batch = fr.FrameBatch.from_frames([synthetic.radarscenes_frame(i) for i in range(4)] + [synthetic.nuscenes_frame(1)])
    cfg = bench.c2_settings()
    g = fr.build_graphs(batch, cfg)
    kw = dict(order=g.cell_order, rank=g.cell_rank, symmetric=True, source_rows=g.rowptr, status=g.status)
    a = TargetCSR(g.edge_index, g.x.shape[0], **kw)
    b = TargetCSR(g.edge_index, g.x.shape[0], own_edges=True, **kw)
    assert b.own_edge is not None and a.own_edge is None
    assert torch.equal(a.rowptr, b.rowptr) and torch.equal(a.src, b.src)
    assert torch.equal(g.edge_attr[a.perm.long()], -g.edge_attr[b.own_edge.long()])
    assert torch.equal(b.perm, a.perm)                                   # (computed on first use)
    # the own edge of slot p leaves the slot's target for the slot's source
    tgt_of_slot = g.edge_index[1][a.perm.long()]
    assert torch.equal(g.edge_index[0][b.own_edge.long()], tgt_of_slot) and torch.equal(g.edge_index[1][b.own_edge.long()], a.src.long())
    outs = {}
    for own in (True, False):
        monkeypatch.setattr(mpnn_layers, "OWN_EDGE_ATTR", own)
        torch.manual_seed(0)
        model = bench.c2_model().cuda()
        cls, bb, gg = fr.HotPath(model, cfg)(batch)
        gg.check()
        outs[own] = (cls.clone(), bb.clone())
    assert torch.equal(outs[True][0], outs[False][0]) and torch.equal(outs[True][1], outs[False][1])
