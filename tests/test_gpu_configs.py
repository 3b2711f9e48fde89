"""BASELINE.json configs beyond the bench workload, at their full sizes, through size-independent properties
(the O(N^2) / per-edge-GEMM oracles do not finish in seconds there) plus exact oracle checks on sampled frames.

  C3  512 nuScenes-shaped frames x 300 points, kNN k = 20, shipped 5-layer model, 11 classes
  C4  (one rank's share) 64 RadarScenes-shaped frames, kNN k = 20, shipped 5-layer model + both heads
  C5  100 000-point cloud, radius graph (~5 M edges), 6-layer model on rotation-invariant features
"""
import numpy as np
import pytest
import torch

from oracle import gnn_oracle as G
from oracle import graph_oracle as go
from radargnn_amd import synthetic

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rg():
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test but no GPU visible")
    from radargnn_amd import frames, gnn
    return frames, gnn


def shipped(gnn, dims, k_classes, node_dim=5, edge_dim=2):
    return gnn.GNNArchitectureConfig(node_dim, edge_dim, dims, [k_classes], [16, 5], True, True, [32, 64, 128, 224],
                                     [4, 8, 16], "MPNNConv", False)


def test_c3_many_small_frames_knn(rg):
    fr, gnn = rg
    frames = [synthetic.nuscenes_frame(i) for i in range(512)]
    cfg = fr.GraphSettings(algorithm="knn", k=20)
    torch.manual_seed(0)
    model = gnn.DetNetBasic(shipped(gnn, [224, 224, 128, 64, 32], 11)).cuda().eval()
    batch = fr.FrameBatch.from_frames(frames)
    cls, bb, g = fr.HotPath(model, cfg)(batch)
    g.check()
    ei = g.edge_index.cpu().numpy()
    n = batch.num_points
    assert ei.shape == (2, n * 20)
    assert np.array_equal(ei[0], np.repeat(np.arange(n), 20))                     # k edges per query, rows ascending
    assert (ei[0] // 300 == ei[1] // 300).all() and (ei[0] != ei[1]).all()         # inside the frame, no self loops
    X = batch.X.cpu().numpy()
    d = X[ei[0]] - X[ei[1]]
    d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]).reshape(n, 20)
    assert (np.diff(d2, axis=1) >= 0).all()                                        # distance-ascending rows
    for f in (0, 17, 511):                                                         # exact oracle check on sampled frames
        sl = slice(f * 300 * 20, (f + 1) * 300 * 20)
        assert np.array_equal(ei[:, sl].T - f * 300, go.knn_edges(frames[f].X, 20))
    assert torch.isfinite(cls).all() and torch.isfinite(bb).all() and cls.shape == (n, 11)
    # eval-mode BatchNorm is per-node: a frame's logits do not depend on what else is in the batch
    for f in (3, 400):
        c1, b1, _ = fr.HotPath(model, cfg)(fr.FrameBatch.from_frames([frames[f]]))
        assert torch.allclose(cls[f * 300:(f + 1) * 300], c1, rtol=1e-5, atol=1e-6)
        assert torch.allclose(bb[f * 300:(f + 1) * 300], b1, rtol=1e-5, atol=1e-6)


def test_c4_share_full_model_knn20(rg):
    fr, gnn = rg
    frames = [synthetic.radarscenes_frame(100 + i) for i in range(64)]
    cfg = fr.GraphSettings(algorithm="knn", k=20)
    torch.manual_seed(1)
    model = gnn.DetNetBasic(shipped(gnn, [224, 224, 128, 64, 32], 6))
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model.cuda()
    cls, bb, g = fr.HotPath(model, cfg, with_softmax=True)(fr.FrameBatch.from_frames(frames))
    g.check()
    assert cls.shape == (192000, 6) and bb.shape == (192000, 5)
    assert torch.allclose(cls.sum(1), torch.ones(192000, device="cuda"), atol=1e-5)    # softmax rows (inference.py:62)
    deg = g.degree.cpu().numpy()
    assert deg.min() >= 20                                                         # undirected degree >= out-degree k
    # train-mode parity against the float64 oracle on an 8-frame batch of the same shape
    sub = frames[:8]
    model2 = gnn.DetNetBasic(shipped(gnn, [224, 224, 128, 64, 32], 6))
    model2.load_state_dict(sd)
    model2.cuda()
    c, b, g8 = fr.HotPath(model2, cfg)(fr.FrameBatch.from_frames(sub))
    ref = go.collate([go.build_frame_graph(f.X, f.V, f.rcs, f.timestamp, "knn", 20, None, list(cfg.node_features),
                                           list(cfg.edge_features), "directed") for f in sub])
    assert np.array_equal(g8.edge_index.cpu().numpy(), ref["edge_index"])
    c64, b64 = G.det_net_basic(torch.from_numpy(ref["x"]), torch.from_numpy(ref["edge_index"]),
                               torch.from_numpy(ref["edge_attr"]), sd, dtype=torch.float64)
    assert ((c.double().cpu() - c64).abs().max() / c64.abs().max()).item() < 1e-5
    assert ((b.double().cpu() - b64).abs().max() / b64.abs().max()).item() < 1e-5


def test_c5_stress_cloud_rotation_invariant(rg):
    fr, gnn = rg
    cloud = synthetic.stress_cloud()
    n = cloud.n
    cfg = fr.GraphSettings(algorithm="radius", r=1.0, node_features=("rcs", "velocity_vector_length", "time_index", "degree"),
                           edge_features=("point_pair_features",))
    torch.manual_seed(2)
    model = gnn.DetNetBasic(shipped(gnn, [224, 224, 224, 128, 64, 32], 6, node_dim=4, edge_dim=4)).cuda().eval()
    cls, bb, g = fr.HotPath(model, cfg)(fr.FrameBatch.from_frames([cloud]))
    g.check()
    e = g.edge_index.shape[1]
    assert 3_000_000 < e < 8_000_000, e
    assert torch.isfinite(cls).all() and torch.isfinite(bb).all()
    ea = g.edge_attr.cpu().numpy()
    assert (ea[:, 0] <= 1.0 + 1e-6).all() and (ea[:, 1:] >= 0).all() and (ea[:, 1:] <= 180.0).all()   # d <= r, angles in degrees
    # E(2) invariance of the whole path: rotate + translate the cloud, rebuild, same logits (features are invariant;
    # topology is, too, except where a pair sits within rounding of the radius)
    th = 0.7
    R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
    rot = synthetic.RadarFrame(cloud.X @ R.T + np.array([13.0, -7.0]), cloud.V @ R.T, cloud.rcs, cloud.timestamp)
    cls2, bb2, g2 = fr.HotPath(model, cfg)(fr.FrameBatch.from_frames([rot]))
    same = g2.edge_index.shape[1] == e and bool((g2.edge_index == g.edge_index).all())
    if same:
        assert ((cls2 - cls).abs().max() / cls.abs().max()).item() < 1e-4
    else:                                                    # a handful of borderline pairs flipped: compare the rest
        changed = abs(g2.edge_index.shape[1] - e)
        assert changed < 200
    # permutation equivariance (eval mode, max aggregation).  r05: this cloud runs on the window form of the max aggregation (5 % of
    # its edges go into targets beyond a stream's 64 slots: TargetCSR.wants_window_kernel), whose windows are packed in visiting
    # order -- points of one grid cell are visited in index order, so a permutation re-packs the windows, and the few targets a full
    # window hands to the per-target kernel (fp32 FMA chain instead of the six-product bf16 split: ~2^-22 |z||w| apart, rgnn.h)
    # are other ones: equal to 2e-6 norm-wise there, and EXACT on the per-edge kernel, whose arithmetic does not depend on the order.
    from radargnn_amd.gnn import mpnn_layers
    perm = np.random.default_rng(0).permutation(n)
    pidx = torch.from_numpy(perm).cuda()
    pf = synthetic.RadarFrame(cloud.X[perm], cloud.V[perm], cloud.rcs[perm], cloud.timestamp[perm])
    cls3, bb3, _ = fr.HotPath(model, cfg)(fr.FrameBatch.from_frames([pf]))
    assert ((cls3 - cls[pidx]).abs().max() / cls.abs().max()).item() < 2e-6
    assert ((bb3 - bb[pidx]).abs().max() / bb.abs().max()).item() < 2e-6
    saved = mpnn_layers.USE_WINDOW_KERNEL
    mpnn_layers.USE_WINDOW_KERNEL = False
    try:
        cls4, bb4, _ = fr.HotPath(model, cfg)(fr.FrameBatch.from_frames([cloud]))
        cls5, bb5, _ = fr.HotPath(model, cfg)(fr.FrameBatch.from_frames([pf]))
    finally:
        mpnn_layers.USE_WINDOW_KERNEL = saved
    assert torch.equal(cls5, cls4[pidx]) and torch.equal(bb5, bb4[pidx])
    assert ((cls4 - cls).abs().max() / cls.abs().max()).item() < 1e-5          # the two kernels against each other


def test_bench_runs_under_a_process_group_on_one_gpu():
    """bench.py with RGNN_BENCH_FORCE_DIST=1: RCCL init on a 1-rank group, barrier, all_gather / MAX-reduce of the timing,
    and the JSON line as the LAST line on stdout -- the code path `bench.py --gpus N` takes under torch.distributed.run,
    exercised where only one GPU exists.  No scaling curve is measured by this."""
    import json
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RGNN_BENCH_FORCE_DIST="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT="29533")
    out = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "2",
                          "--no-cpu-baseline", "--no-other-configs", "--launch-mode", "eager"], env=env, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["scaling"] == "weak" and line["value"] > 0
    assert line["config"]["ranks_in_process_group"] == 1
    assert len(line["config"]["per_rank_frames_per_s"]) == 1
    assert abs(line["config"]["per_rank_frames_per_s"][0] - line["value"]) / line["value"] < 0.05
    # the f16x2 dense form executes 226 flop per algorithmic byte, below the machine balance of 312: bound by HBM
    r = line["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and 0 < r["frac"] < 1
    assert r["arithmetic_intensity_flop_per_byte"] < r["machine_balance_flop_per_byte"]
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) / r["achieved"] < 1e-6
    assert r["mfma"]["unit"] == "TFLOP/s" and 0 < r["mfma"]["frac"] < 1
    g = line["roofline_gather"]
    assert 0 < g["compulsory_frac"] < 1 and 0 < g["l2_frac"] < 1
    sr = line["roofline_search"]                    # graph-construction stage against its compulsory bytes
    assert sr["bound"] == "hbm" and 0 < sr["frac"] < 1 and sr["ms_per_batch"] > 0
    assert sr["bytes_per_batch"] == 48 * 192000 + 16 * line["config"]["edges_per_gpu"] + 4 * 2 * line["config"]["edges_per_gpu"]
    assert line["self_check"].startswith("ok")      # the timed steps' outputs: finite, bit-equal to one eager pass
    c4 = line["c4"]                                 # BASELINE.json configs[3]: every rank's 1024-frame share, summed by rank 0
    assert c4["frames_per_s_total"] > 0 and len(c4["per_rank_frames_per_s"]) == 1 and c4["rank0"]["frames"] == 1024
    assert abs(c4["per_rank_frames_per_s"][0] - c4["frames_per_s_total"]) / c4["frames_per_s_total"] < 1e-6
    assert c4["rank0"]["batches"] == 16 and c4["rank0"]["dominant_kernel"]


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus N` without a launcher starts its N ranks itself (bench.self_launch_command): exercised with the one
    GPU there is -- RGNN_BENCH_SELF_LAUNCH=1 sends --gpus 1 through torch.distributed.run too, and RGNN_BENCH_FORCE_DIST=1 makes the
    single rank go through RCCL init / barrier / all_gather like a rank of eight."""
    import json
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(RGNN_BENCH_SELF_LAUNCH="1", RGNN_BENCH_FORCE_DIST="1", MASTER_PORT="29541")
    out = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "2",
                          "--no-cpu-baseline", "--no-other-configs", "--no-c4", "--launch-mode", "eager"], env=env, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["config"]["ranks_in_process_group"] == 1
    assert line["self_check"].startswith("ok")
