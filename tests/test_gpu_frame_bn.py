"""BatchNorm statistics per frame in a batched launch (``HotPath(bn_scope="frame")``, ``DetNetBasic.forward(frame_ptr=...)``).

The reference evaluates with ``batch_size=1`` and never puts the model in eval mode (evaluate.py:40,
postprocessor/inference.py:57-62, gnn/gnn_models.py:124-128): each frame is normalised with its own batch statistics.  A batch
of frames with per-frame statistics must therefore equal the single-frame forwards -- to 4e-6 norm-wise against the HIP path
run frame by frame (different summation order of the statistics only), to 1e-5 against the float64 oracle run per frame."""
import copy

import numpy as np
import pytest
import torch

from oracle import gnn_oracle as G
from oracle import graph_oracle as go
from radargnn_amd import synthetic

pytestmark = pytest.mark.gpu


def normwise(a, b) -> float:
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-300)).item()


def _model(bn_in_mlps: bool, seed: int, conv_dims=(224, 128, 64), node_emb=(32, 64, 224)):
    from radargnn_amd import gnn
    cfg = gnn.GNNArchitectureConfig(5, 2, list(conv_dims), [6], [16, 5], True, True, list(node_emb), [4, 8, 16], "MPNNConv", bn_in_mlps)
    torch.manual_seed(seed)
    return gnn.DetNetBasic(cfg)


# (the last case: layer widths that are not whole 32-column steps -- 80 -- keep the conv layers off the frame-padded row lists)
@pytest.mark.parametrize("algo,bn_in_mlps,dims", [("radius", False, None), ("knn", False, None), ("knn", True, None),
                                                  ("radius", False, ((80, 48), (32, 48)))])
def test_batched_frames_with_per_frame_statistics_equal_single_frame_forwards(algo, bn_in_mlps, dims):
    from radargnn_amd import frames as fr
    frames = [synthetic.radarscenes_frame(i) for i in range(5)] + [synthetic.nuscenes_frame(3)]     # ragged: 3000 ... 300 points
    cfg = fr.GraphSettings(algorithm=algo, k=10, r=1.5)
    model = _model(bn_in_mlps, 11) if dims is None else _model(bn_in_mlps, 11, *dims)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model.cuda().train()
    single = copy.deepcopy(model)
    batch = fr.FrameBatch.from_frames(frames)
    cls, bb, g = fr.HotPath(model, cfg, bn_scope="frame")(batch)
    g.check()
    ptr = batch.frame_ptr.cpu().numpy()
    hp1 = fr.HotPath(single, cfg)
    worst_hip = worst_64 = 0.0
    for f, frame in enumerate(frames):
        c1, b1, g1 = hp1(fr.FrameBatch.from_frames([frame]))          # the reference's regime: one frame per forward
        sl = slice(int(ptr[f]), int(ptr[f + 1]))
        worst_hip = max(worst_hip, normwise(cls[sl], c1), normwise(bb[sl], b1))
        ref = go.build_frame_graph(frame.X, frame.V, frame.rcs, frame.timestamp, algo, 10, 1.5, list(cfg.node_features),
                                   list(cfg.edge_features), "directed")
        c64, b64 = G.det_net_basic(torch.from_numpy(ref["x"]), torch.from_numpy(ref["edge_index"]),
                                   torch.from_numpy(ref["edge_attr"]), sd, dtype=torch.float64)
        worst_64 = max(worst_64, normwise(cls[sl], c64), normwise(bb[sl], b64))
    assert worst_hip < 4e-6, worst_hip      # (two fp32 schedules of the same arithmetic: split-K / other pre-scales on one frame)
    assert worst_64 < 1e-5, worst_64
    # the running statistics went through the frames one after the other, like the loop of single-frame forwards
    for (name, a), (_, b) in zip(model.named_buffers(), single.named_buffers()):
        if a.dtype.is_floating_point:
            np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=2e-5, atol=1e-6, err_msg=name)
        else:
            assert torch.equal(a, b), name
    # ... and batch-wide statistics give something else (the two scopes are different functions of the batch)
    cls_b, _, _ = fr.HotPath(copy.deepcopy(single), cfg, bn_scope="batch")(batch)
    assert normwise(cls_b, cls) > 1e-4


def test_public_forward_with_frame_ptr_and_hip_graph_replay():
    """The same through the module surface (``forward(x, edge_index, edge_attr, frame_ptr=Batch.ptr)``) and through a captured
    whole-step HIP graph: bit-equal to the eager HotPath result."""
    from radargnn_amd import frames as fr
    frames = [synthetic.nuscenes_frame(i) for i in range(9)]
    cfg = fr.GraphSettings(algorithm="knn", k=8)
    model = _model(False, 4).cuda().train()
    batch = fr.FrameBatch.from_frames(frames)
    cls, bb, g = fr.HotPath(copy.deepcopy(model), cfg, bn_scope="frame")(batch)
    c2, b2 = copy.deepcopy(model)(g.x, g.edge_index, g.edge_attr, frame_ptr=batch.frame_ptr)
    assert normwise(c2, cls) < 1e-6 and normwise(b2, bb) < 1e-6       # (the public forward has no visiting order: same values,
    hot = fr.HotPath(copy.deepcopy(model), cfg, bn_scope="frame", use_hip_graphs=True)   # another summation order in places)
    for _ in range(4):
        c3, b3, _ = hot(batch)
    torch.cuda.synchronize()
    assert hot._graph is not None and torch.equal(c3, cls) and torch.equal(b3, bb)


@pytest.mark.parametrize("relu", [False, True])
def test_one_pass_segment_batchnorm_equals_statistics_then_apply(relu):
    """rgnn_batchnorm_act_segments (the block that sums a slab normalises it) against rgnn_batchnorm_segments followed by
    rgnn_scale_shift_act_segments, and both against float64 numpy; ragged segments with an empty one and a one-row one,
    a channel count that is not a multiple of the 64-channel slab, a padded row stride, running statistics included."""
    from radargnn_amd import ops
    rng = np.random.default_rng(5)
    sizes = [700, 0, 1, 3001, 17, 64]
    seg = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    m, n = int(seg[-1]), 224 - 19
    xh = (rng.standard_normal((m, n)) * rng.uniform(0.1, 30.0, size=n) + rng.uniform(-5, 5, size=n)).astype(np.float32)
    x = ops.padded_rows(m, n, torch.device("cuda"))
    x.copy_(torch.from_numpy(xh).cuda())
    gamma = torch.from_numpy(rng.uniform(0.5, 1.5, n).astype(np.float32)).cuda()
    beta = torch.from_numpy(rng.uniform(-1, 1, n).astype(np.float32)).cuda()
    segd = torch.from_numpy(seg).cuda()

    def fresh():
        return torch.zeros(n, device="cuda"), torch.ones(n, device="cuda"), torch.zeros((), dtype=torch.int64, device="cuda")

    rm1, rv1, nb1 = fresh()
    table = ops.batchnorm_segments(x, segd, gamma, beta, rm1, rv1, nb1, 0.1, 1e-5)
    want = ops.scale_shift_act_segments(x, table, segd, relu)
    rm2, rv2, nb2 = fresh()
    got = ops.batchnorm_act_segments(x, segd, gamma, beta, rm2, rv2, nb2, 0.1, 1e-5, relu)
    assert got.shape == (m, n)
    assert normwise(got, want) <= 1e-6
    assert int(nb1) == int(nb2)
    np.testing.assert_allclose(rm2.cpu().numpy(), rm1.cpu().numpy(), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(rv2.cpu().numpy(), rv1.cpu().numpy(), rtol=1e-6, atol=1e-7)

    # (a one-row segment has zero variance: torch refuses it in training mode; here x - mean is exactly 0 and the row comes out as
    # beta -- the r03 form x * scale + shift carried the rounding of mean * gamma / sqrt(eps) there)
    ref = got.detach().double().cpu().numpy().copy()
    x64 = xh.astype(np.float64)
    for f in range(len(sizes)):
        a, b = int(seg[f]), int(seg[f + 1])
        if b - a < 1:
            continue
        mu, var = x64[a:b].mean(0), x64[a:b].var(0)
        ref[a:b] = (x64[a:b] - mu) / np.sqrt(var + 1e-5) * gamma.cpu().numpy() + beta.cpu().numpy()
    if relu:
        ref = np.maximum(ref, 0.0)
    assert normwise(got, torch.from_numpy(ref)) <= 2e-6


def test_frame_statistics_from_the_dense_epilogues_equal_the_pass_over_the_activations(monkeypatch):
    """C2-shaped batch (radius graph, ragged frames): per-frame BatchNorm with the statistics taken from the conv layers' own
    epilogues on frame-padded row lists and applied by the next layer's dense launches (gnn_models.FUSE_FRAME_BN) against the
    form that normalises [N, C] in a pass of its own; the fused launches must actually have run; running statistics equal."""
    from radargnn_amd import frames as fr, ops
    from radargnn_amd.gnn import gnn_models
    import bench
    # (40 x 35 + 1600 = 3000 points, ..., 4 x 35 + 160 = 300: ragged frames, the last one barely more than one 256-row tile)
    frames_list = [synthetic.radarscenes_frame(i, n_clusters=c, n_clutter=u)
                   for i, (c, u) in enumerate([(40, 1600), (30, 1450), (10, 350), (40, 1600), (24, 960), (4, 160)])]
    batch = fr.FrameBatch.from_frames(frames_list)
    outs, stats = {}, {}
    for fused in (True, False):
        monkeypatch.setattr(gnn_models, "FUSE_FRAME_BN", fused)
        torch.manual_seed(3)
        model = bench.c2_model().cuda()
        before = ops.COUNTERS.get("fused_a1_affine_segments", 0)
        hot = fr.HotPath(model, bench.c2_settings(), bn_scope="frame")
        cls, bb, g = hot(batch)
        g.check()
        outs[fused] = (cls.clone(), bb.clone())
        stats[fused] = [(b.module.running_mean.clone(), b.module.running_var.clone()) for b in model.batch_norms]
        ran = ops.COUNTERS.get("fused_a1_affine_segments", 0) - before
        assert (ran >= 3 * (len(model.convs) - 1)) if fused else (ran == 0)
    for a, b in zip(outs[True], outs[False]):
        assert torch.isfinite(a).all()
        assert normwise(a, b) <= 4e-6
    for (m1, v1), (m2, v2) in zip(stats[True], stats[False]):
        np.testing.assert_allclose(m1.cpu().numpy(), m2.cpu().numpy(), rtol=2e-5, atol=1e-6)
        np.testing.assert_allclose(v1.cpu().numpy(), v2.cpu().numpy(), rtol=2e-5, atol=1e-6)


def test_pad_list_by_segment_against_numpy():
    """rgnn_pad_list_by_segment: ascending list cut at the segment borders, every segment padded with -1 to whole 256-row tiles;
    an empty segment, a segment without list entries, a list that ends before the last segment."""
    from radargnn_amd import ops
    rng = np.random.default_rng(11)
    seg = np.array([0, 700, 700, 1500, 4000, 4100, 6000], dtype=np.int64)          # segment 1 is empty
    keep = rng.random(6000) < 0.6
    keep[1500:4000] |= rng.random(2500) < 0.9
    keep[4000:4100] = False                                                            # segment 4 has no entry in the list
    keep[5990:] = False
    ids = np.nonzero(keep)[0].astype(np.int32)
    buf = np.full(6000, 123456, dtype=np.int32)                                        # (capacity N, live length on the device)
    buf[:len(ids)] = ids
    out, total, tiles, start = ops.pad_list_by_segment(torch.from_numpy(buf).cuda(), torch.tensor([len(ids)], device="cuda"),
                                                       torch.from_numpy(seg).cuda())
    want, want_tiles, want_start = [], [], []
    for f in range(len(seg) - 1):
        part = ids[(ids >= seg[f]) & (ids < seg[f + 1])]
        plen = (len(part) + 255) // 256 * 256
        want_start.append(len(want) // 128)
        want += list(part) + [-1] * (plen - len(part))
        want_tiles += [f] * (plen // 256)
    want_start.append(len(want) // 128)
    assert int(total) == len(want)
    np.testing.assert_array_equal(out.cpu().numpy()[:len(want)], np.array(want, dtype=np.int32))
    np.testing.assert_array_equal(tiles.cpu().numpy()[:len(want_tiles)], np.array(want_tiles, dtype=np.int32))
    np.testing.assert_array_equal(start.cpu().numpy(), np.array(want_start, dtype=np.int32))


@pytest.mark.parametrize("k2", [0, 464])
def test_dense_launch_on_a_segment_padded_list_with_per_segment_tables(k2):
    """rgnn_linear_fwd with a1_panel_segment: rows named by a segment-padded list (-1 entries skipped), the A1 operand
    relu(x scale_f + shift_f) with the table of the row's segment, column statistics per 128-row panel of the list -- against
    float64 torch; rows outside the list keep what they held; rgnn_batchnorm_segments_from_panels on those statistics against
    per-segment sums of the output."""
    from radargnn_amd import ops
    rng = np.random.default_rng(7 + k2)
    seg = np.array([0, 900, 900, 2100, 5000, 5300], dtype=np.int64)
    n_rows, k1, n = 5300, 224, 160
    x = torch.from_numpy(rng.standard_normal((n_rows, k1)).astype(np.float32)).cuda()
    a2 = torch.from_numpy(rng.standard_normal((n_rows, k2)).astype(np.float32)).cuda() if k2 else None
    w = torch.from_numpy((rng.standard_normal((n, k1 + k2)) / np.sqrt(k1 + k2)).astype(np.float32)).cuda()
    b = torch.from_numpy(rng.standard_normal(n).astype(np.float32)).cuda()
    table = torch.from_numpy(np.stack([np.stack([rng.uniform(-0.5, 0.5, k1), rng.uniform(0.5, 1.5, k1), rng.uniform(-0.5, 0.5, k1)])   # mean_hi, g, t
                                       for _ in range(len(seg) - 1)]).astype(np.float32)).cuda()
    ids = np.nonzero(rng.random(n_rows) < 0.7)[0].astype(np.int32)
    buf = np.full(n_rows, 123456, dtype=np.int32)        # (a list has room for every row of the matrix: rgnn.h, row subsets)
    buf[:len(ids)] = ids
    lst, total, tiles, start = ops.pad_list_by_segment(torch.from_numpy(buf).cuda(), torch.tensor([len(ids)], device="cuda"),
                                                       torch.from_numpy(seg).cuda())
    out = torch.full((n_rows, n), -7.0, device="cuda")
    stats = torch.zeros((max(ops.stat_panels(lst.numel()), 1), ops.STAT_ROWS, n), device="cuda")
    with ops.bound_tracking(x.device):
        ops.set_bound(x, ops.make_bound(x.abs().max())); ops.set_bound(table, ops.make_bound(torch.tensor(8.0, device="cuda")))
        if a2 is not None:
            ops.set_bound(a2, ops.make_bound(a2.abs().max()))
        before = ops.COUNTERS.get("fused_a1_affine_segments", 0)
        ops.linear(x, w, b, a2=a2, out=out, relu=True, row_index=lst, m_dev=total, stats_out=stats, a1_affine=table,
                   a1_affine_tiles=tiles)
        assert ops.COUNTERS.get("fused_a1_affine_segments", 0) == before + 1
    frame_of = np.searchsorted(seg, np.arange(n_rows), side="right") - 1
    xa = torch.relu(ops.apply_table_reference(x.cpu(), table.cpu()[frame_of]))
    full = torch.cat([xa, a2.double().cpu()], dim=1) if a2 is not None else xa
    ref = torch.relu(full @ w.double().cpu().t() + b.double().cpu())
    got = out.cpu()
    listed = np.zeros(n_rows, dtype=bool); listed[ids] = True
    assert normwise(got[listed], ref[listed]) <= 2e-6
    assert (got[~listed] == -7.0).all()
    # column statistics: segment f owns panels [start[f], start[f + 1]) of the list
    sp = start.cpu().numpy()
    for f in range(len(seg) - 1):
        rows = ids[(ids >= seg[f]) & (ids < seg[f + 1])]
        c0, s1, s2 = (t_.cpu().numpy() for t_ in ops.stats_to_sums(stats[sp[f]:sp[f + 1]]))     # panels hold {count, mean, M2}
        assert (c0 == len(rows)).all()
        r = ref[rows].numpy()
        np.testing.assert_allclose(s1, r.sum(0), rtol=2e-5, atol=2e-3)
        np.testing.assert_allclose(s2, (r * r).sum(0), rtol=2e-5, atol=2e-3)
    # ... and the table BatchNorm makes of them: statistics over the LISTED rows of each segment
    cnt = np.array([((ids >= seg[f]) & (ids < seg[f + 1])).sum() for f in range(len(seg) - 1)])
    seg_listed = torch.from_numpy(np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)).cuda()
    gamma = torch.ones(n, device="cuda"); beta = torch.zeros(n, device="cuda")
    tab = ops.batchnorm_segments_from_panels(stats, start, None, None, seg_listed, gamma, beta, None, None, None, 0.1, 1e-5)
    want = ops.batchnorm_segments(out[torch.from_numpy(ids).cuda().long()].contiguous(), seg_listed, gamma, beta, None, None, None, 0.1, 1e-5)
    live = torch.from_numpy(cnt > 1).cuda()
    assert normwise(tab[live], want[live]) <= 1e-5


def test_a_segment_padded_list_is_refused_where_no_kernel_skips_its_absent_rows():
    """Only the LDS-DMA kernel treats a row_index entry of -1 as an absent row; a launch that would run on another kernel (here:
    k1 not in whole 32-column steps next to a second operand block) must be refused, not read row -1."""
    from radargnn_amd import ops
    from radargnn_amd._lib import RgnnError
    n_rows = 6000
    x = torch.randn(n_rows, 208, device="cuda"); a2 = torch.randn(n_rows, 16, device="cuda")
    w = torch.randn(160, 224, device="cuda")
    seg = torch.tensor([0, 2500, 6000], device="cuda")
    lst, total, tiles, start = ops.pad_list_by_segment(torch.arange(n_rows, dtype=torch.int32, device="cuda"),
                                                       torch.tensor([n_rows], device="cuda"), seg)
    out = torch.zeros(n_rows, 160, device="cuda")
    with pytest.raises(RgnnError):
        ops.linear(x, w, None, a2=a2, out=out, row_index=lst, m_dev=total, padded_row_list=True)
    torch.cuda.synchronize()
