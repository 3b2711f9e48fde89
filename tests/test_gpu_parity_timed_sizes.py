"""Parity against float64 AT THE SIZES AND SHAPES bench.py TIMES ("what's weak" 1-2).

The oracle tests elsewhere run the timed code paths on 8 - 24 frames, where the batch is too small for the window kernel
(``k_mpnn_win`` needs 2^18 / 2^19 edges) and the faithful per-edge float64 oracle still finishes in seconds.  Here:

* C2 full batch (64 frames x 3000 points, r = 1 m, 799 078 edges): train mode, the whole step replayed from ONE HIP graph, window
  kernel at D = 464 / 272 asserted through ``ops.COUNTERS``, own-edge CSR, f16x2 dense layers;
* C3: 16 frames (shipped 5-layer model, 11 classes, TRAIN mode) against the faithful oracle, and the full 512-frame batch
  (3 072 000 edges: window kernel at k = 20);
* C4 share: one full 64 x 3000, k = 20 batch (3 840 000 edges);
* C5: a >= 4000-point crop of the stress cloud (point-pair features, 6-layer model) against the faithful oracle, and the full
  100 000-point cloud.

Full sizes compare with ``oracle/gnn_hoisted.py`` -- the same function in float64 through the hoisted algebra, evaluated by
torch's own float64 kernels -- which each test FIRST pins against the faithful ``oracle/gnn_oracle.py`` on a small batch of the
same configuration.  The graph half at full size is checked through the oracle's topology on sampled frames (the O(N^2) numpy
search per 3000-point frame takes ~0.3 s).  Every measured error is printed and recorded (conftest.record_parity); tolerance:
max|a - b| <= 1e-5 max|b| per output tensor (BASELINE.json north_star).
"""
import numpy as np
import pytest
import torch

from conftest import record_parity
from oracle import gnn_hoisted as GH
from oracle import gnn_oracle as G
from oracle import graph_oracle as go
from radargnn_amd import synthetic

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def rg():
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test but no GPU visible")
    from radargnn_amd import frames, gnn, ops
    return frames, gnn, ops


def nerr(a: torch.Tensor, b64: torch.Tensor) -> float:
    return ((a.double().cpu() - b64).abs().max() / b64.abs().max()).item()


def oracle_graphs(frames, cfg):
    return go.collate([go.build_frame_graph(f.X, f.V, f.rcs, f.timestamp, cfg.algorithm, cfg.k, cfg.r, list(cfg.node_features),
                                            list(cfg.edge_features), cfg.edge_mode) for f in frames])


def pin_hoisted(frames, cfg, sd, name):
    """The hoisted float64 evaluation against the faithful per-edge oracle on a small batch of this configuration."""
    ref = oracle_graphs(frames, cfg)
    args = (torch.from_numpy(ref["x"]), torch.from_numpy(ref["edge_index"]), torch.from_numpy(ref["edge_attr"]), sd)
    c0, b0 = G.det_net_basic(*args, dtype=torch.float64)
    c1, b1 = GH.det_net_basic_hoisted(*args, device="cuda")
    ec, eb = nerr(c1, c0), nerr(b1, b0)
    record_parity(name + " [hoisted f64 vs faithful f64]", logits=ec, boxes=eb)
    assert ec < 1e-10 and eb < 1e-10, (ec, eb)


def full_size_check(name, hot_out, frames, cfg, sd, sample):
    """Topology of sampled frames against the oracle (bit-exact), logits / boxes of the WHOLE batch against the hoisted float64
    evaluation of the device's own graph tensors (which the sampled frames tie to the oracle's)."""
    cls, bb, g = hot_out
    g.check()
    ei = g.edge_index.cpu().numpy()
    ptr = np.concatenate([[0], np.cumsum([f.n for f in frames])])
    for f in sample:
        one = go.build_frame_graph(frames[f].X, frames[f].V, frames[f].rcs, frames[f].timestamp, cfg.algorithm, cfg.k, cfg.r,
                                   list(cfg.node_features), list(cfg.edge_features), cfg.edge_mode)
        sel = (ei[0] >= ptr[f]) & (ei[0] < ptr[f + 1])
        assert np.array_equal(ei[:, sel] - ptr[f], one["edge_index"]), f"frame {f}: edge_index differs from the oracle"
        assert np.array_equal(g.x[ptr[f]:ptr[f + 1]].cpu().numpy(), one["x"]), f"frame {f}: node features differ"
        np.testing.assert_allclose(g.edge_attr.cpu().numpy()[sel], one["edge_attr"], rtol=2e-7, atol=1e-6)
    c64, b64 = GH.det_net_basic_hoisted(g.x, g.edge_index, g.edge_attr, sd, device="cuda")
    ec, eb = nerr(cls, c64), nerr(bb, b64)
    record_parity(name, logits=ec, boxes=eb)
    assert ec < TOL and eb < TOL, (ec, eb)


def test_c2_full_batch_as_timed_vs_float64(rg):
    """bench.py's timed path at its own size: C2 model in TRAINING mode, whole step replayed from one HIP graph, 64 frames."""
    import bench
    fr, gnn, ops = rg
    frames = [synthetic.radarscenes_frame(i) for i in range(64)]
    cfg = bench.c2_settings()
    model = bench.c2_model()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model.cuda().train()
    pin_hoisted(frames[:2], cfg, sd, "C2 2 frames")
    batch = fr.FrameBatch.from_frames(frames)
    hot = fr.HotPath(model, cfg, use_hip_graphs=True)
    win0, f160 = ops.COUNTERS.get("mpnn_win", 0), ops.COUNTERS.get("f16x2", 0)
    for _ in range(5):                                          # first sight eager, capture, replays
        out = hot(batch)
    torch.cuda.synchronize()
    assert hot._graph is not None, "the step was not captured"
    assert ops.COUNTERS.get("mpnn_win", 0) - win0 >= 8, "the window kernel did not run (4 layers x eager + capture)"
    assert ops.COUNTERS.get("f16x2", 0) > f160
    assert out[2].edge_index.shape[1] == 799078                # (the synthetic workload is deterministic: DESIGN section 4)
    full_size_check("C2 full batch (64 x 3000, r = 1 m, HIP graph, train mode)", out, frames, cfg, sd, sample=(0, 31, 63))


def test_c3_train_mode_vs_float64(rg):
    fr, gnn, ops = rg
    import bench
    cfg = fr.GraphSettings(algorithm="knn", k=20)
    model = bench.shipped_model([224, 224, 128, 64, 32], 11)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model.cuda().train()
    # (b) of the brief: 16 frames, faithful float64 oracle on the oracle's own graphs
    sub = [synthetic.nuscenes_frame(i) for i in range(16)]
    cls, bb, g = fr.HotPath(model, cfg)(fr.FrameBatch.from_frames(sub))
    g.check()
    ref = oracle_graphs(sub, cfg)
    assert np.array_equal(g.edge_index.cpu().numpy(), ref["edge_index"]) and np.array_equal(g.x.cpu().numpy(), ref["x"])
    c64, b64 = G.det_net_basic(torch.from_numpy(ref["x"]), torch.from_numpy(ref["edge_index"]), torch.from_numpy(ref["edge_attr"]),
                               sd, dtype=torch.float64)
    ec, eb = nerr(cls, c64), nerr(bb, b64)
    record_parity("C3 16 frames (train mode, faithful f64 oracle)", logits=ec, boxes=eb)
    assert ec < TOL and eb < TOL, (ec, eb)
    # the full batch: window kernel at k = 20
    pin_hoisted(sub[:4], cfg, sd, "C3 4 frames")
    frames = [synthetic.nuscenes_frame(i) for i in range(512)]
    win0 = ops.COUNTERS.get("mpnn_win", 0)
    out = fr.HotPath(model, cfg)(fr.FrameBatch.from_frames(frames))
    assert ops.COUNTERS.get("mpnn_win", 0) > win0
    full_size_check("C3 full batch (512 x 300, kNN k = 20, train mode)", out, frames, cfg, sd, sample=(0, 255, 511))


def test_c4_share_full_batch_vs_float64(rg):
    fr, gnn, ops = rg
    import bench
    cfg = fr.GraphSettings(algorithm="knn", k=20)
    model = bench.shipped_model([224, 224, 128, 64, 32], 6)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model.cuda().train()
    frames = [synthetic.radarscenes_frame(100 + i) for i in range(64)]
    pin_hoisted([synthetic.nuscenes_frame(i) for i in range(4)], cfg, sd, "C4 model, 4 small frames")
    win0 = ops.COUNTERS.get("mpnn_win", 0)
    out = fr.HotPath(model, cfg)(fr.FrameBatch.from_frames(frames))
    assert ops.COUNTERS.get("mpnn_win", 0) > win0
    full_size_check("C4 one batch (64 x 3000, kNN k = 20, train mode)", out, frames, cfg, sd, sample=(0, 40))


def _c5_settings(fr):
    return fr.GraphSettings(algorithm="radius", r=1.0, node_features=("rcs", "velocity_vector_length", "time_index", "degree"),
                            edge_features=("point_pair_features",))


def test_c5_six_layer_rotation_invariant_vs_float64(rg):
    fr, gnn, ops = rg
    import bench
    cfg = _c5_settings(fr)
    model = bench.shipped_model([224, 224, 224, 128, 64, 32], 6, node_dim=4, edge_dim=4)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model.cuda().train()
    cloud = synthetic.stress_cloud()
    # (c) of the brief: a crop of the stress cloud with >= 4000 points (crowded: ~34 neighbours per point), faithful oracle
    keep = (cloud.X[:, 0] >= 20.0) & (cloud.X[:, 0] < 40.0) & (cloud.X[:, 1] >= -10.0) & (cloud.X[:, 1] < 10.0)
    crop = synthetic.RadarFrame(cloud.X[keep], cloud.V[keep], cloud.rcs[keep], cloud.timestamp[keep])
    assert crop.n >= 4000, crop.n
    cls, bb, g = fr.HotPath(model, cfg)(fr.FrameBatch.from_frames([crop]))
    g.check()
    ref = oracle_graphs([crop], cfg)
    assert np.array_equal(g.edge_index.cpu().numpy(), ref["edge_index"]) and np.array_equal(g.x.cpu().numpy(), ref["x"])
    np.testing.assert_allclose(g.edge_attr.cpu().numpy(), ref["edge_attr"], rtol=2e-6, atol=2e-5)     # (angles in degrees, f32)
    assert ref["edge_index"].shape[1] > 20 * crop.n            # crowded
    c64, b64 = G.det_net_basic(torch.from_numpy(ref["x"]), torch.from_numpy(ref["edge_index"]), torch.from_numpy(ref["edge_attr"]),
                               sd, dtype=torch.float64)
    ec, eb = nerr(cls, c64), nerr(bb, b64)
    record_parity(f"C5 crop ({crop.n} points, {ref['edge_index'].shape[1]} edges, 6 layers, train mode, faithful f64 oracle)",
                  logits=ec, boxes=eb)
    assert ec < TOL and eb < TOL, (ec, eb)
    # the whole 100 000-point cloud against the hoisted evaluation of the device's graph tensors
    small = synthetic.RadarFrame(crop.X[:600], crop.V[:600], crop.rcs[:600], crop.timestamp[:600])
    pin_hoisted([small], cfg, sd, "C5 600 points")
    cls, bb, g = fr.HotPath(model, cfg)(fr.FrameBatch.from_frames([cloud]))
    g.check()
    c64, b64 = GH.det_net_basic_hoisted(g.x, g.edge_index, g.edge_attr, sd, device="cuda")
    ec, eb = nerr(cls, c64), nerr(bb, b64)
    record_parity(f"C5 full cloud (100 000 points, {g.edge_index.shape[1]} edges, 6 layers, train mode)", logits=ec, boxes=eb)
    assert ec < TOL and eb < TOL, (ec, eb)
