"""CPU-only checks of the host side: the C ABI library loads and exports every symbol include/rgnn.h declares, the
module mirrors have the reference's constructor signatures / attribute layout / state_dict keys, the frame
sharding is a partition, and the product refuses CPU tensors instead of falling back."""
import os
import re

import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_abi_exports_every_declared_symbol():
    from radargnn_amd import _lib
    header = open(os.path.join(REPO, "include", "rgnn.h")).read()
    declared = set(re.findall(r"\b(rgnn_[a-z0-9_]+)\s*\(", header))
    declared -= {"rgnn_linear_args", "rgnn_grid"}
    assert declared, "no declarations parsed"
    missing = [name for name in sorted(declared) if not hasattr(_lib.lib, name)]
    assert not missing, f"librgnn.so does not export: {missing}"
    unbound = sorted(declared - set(_lib.SIGNATURES))
    assert not unbound, f"declared in rgnn.h but not bound in _lib.SIGNATURES: {unbound}"
    assert _lib.lib.rgnn_version().startswith(b"rgnn")


def test_c_abi_argument_errors_without_gpu():
    from radargnn_amd import _lib
    assert _lib.lib.rgnn_grid_workspace_bytes(3000, 1, 3) == -1          # dim must be 2 or 4
    assert _lib.lib.rgnn_grid_workspace_bytes(3000, 1, 2) > 0
    assert _lib.lib.rgnn_linear_stat_panels(129) == 2
    for n, e in ((1000, 5000), (3000, 30000), (192000, 800000), (192000, 3840000)):
        w, a = _lib.lib.rgnn_mpnn_work_units(n, e), _lib.lib.rgnn_mpnn_target_weight(n, e)
        assert 8 <= w <= 80 and a >= 2 and w // a <= 63                    # a chunk holds at most 63 targets
        assert _lib.lib.rgnn_mpnn_num_chunks(n, e) == (e + a * n + w - 1) // w + 1
    assert _lib.lib.rgnn_mpnn_work_units(192000, 800000) == 80             # full batches: the measured optimum
    assert _lib.lib.rgnn_mpnn_work_units(3000, 30000) == 16                # one frame: finer chunks, so the chip fills
    rc = _lib.lib.rgnn_linear_fwd(None, None)
    assert rc == -1 and b"null args" in _lib.lib.rgnn_last_error()


def test_state_dict_keys_match_the_reference_contract():
    from radargnn_amd import gnn
    cfg = gnn.GNNArchitectureConfig(5, 2, [224, 224, 128, 64, 32], [6], [16, 5], True, True, [32, 64, 128, 224],
                                    [4, 8, 16], "MPNNConv", False)
    model = gnn.DetNetBasic(cfg)
    keys = set(model.state_dict().keys())
    expected = set()
    for i in (0, 2, 4, 6):
        expected |= {f"node_emb_mlp.{i}.weight", f"node_emb_mlp.{i}.bias"}
    for i in (0, 2, 4):
        expected |= {f"edge_emb_mlp.{i}.weight", f"edge_emb_mlp.{i}.bias"}
    for l in range(5):
        expected |= {f"convs.{l}.pre_mlp.0.weight", f"convs.{l}.pre_mlp.0.bias", f"convs.{l}.post_mlp.0.weight",
                     f"convs.{l}.post_mlp.0.bias"}
        expected |= {f"batch_norms.{l}.module.{n}" for n in ("weight", "bias", "running_mean", "running_var",
                                                             "num_batches_tracked")}
    expected |= {"classification_head.0.weight", "classification_head.0.bias", "regression_head.0.weight",
                 "regression_head.0.bias", "regression_head.2.weight", "regression_head.2.bias"}
    assert keys == expected
    assert sum(p.numel() for p in model.parameters()) == 1_213_503
    assert model.convs[0].pre_mlp[0].weight.shape == (464, 464)              # D = 2C + De
    assert model.convs[0].post_mlp[0].weight.shape == (224, 688)             # [x | m] -> Co
    # batch_norm_in_mlps shifts the Sequential indices by one per inserted BatchNorm (gnn_models.py:161-171)
    cfg.batch_norm_in_mlps = True
    k2 = set(gnn.DetNetBasic(cfg).state_dict().keys())
    assert "node_emb_mlp.1.module.running_mean" in k2 and "node_emb_mlp.3.weight" in k2


def test_layer_constructors_match_reference_tests():
    from radargnn_amd import gnn
    conv = gnn.RadarPointGNNConv(2, 1, "max", 2, 1)                          # test/test_gnn.py:42-76
    assert len(conv.pre_mlp) == 3 and len(conv.post_mlp) == 1
    conv = gnn.MPNNConv(2, 4, 3, post_layers=2)                              # test/test_gnn.py:79-116
    assert len(conv.pre_mlp) == 1 and len(conv.post_mlp) == 3
    assert conv.pre_mlp[0].weight.shape == (7, 7) and conv.post_mlp[0].weight.shape == (4, 9)
    conv = gnn.MPNNConv(1, 4, 2, use_edge_encoder=True)                      # test/test_gnn.py:175-221
    assert conv.pre_mlp[0].weight[0].shape[0] == 3 and conv.edge_encoder.weight.shape == (1, 2)
    mlp = gnn.get_mlp(2, 3, [5], False)                                      # test/test_gnn.py:9-25
    assert mlp[0].weight.shape == (5, 2) and mlp[2].weight.shape == (3, 5)
    with pytest.raises(Exception, match="invalid GNN conv layer type"):
        gnn.DetNetBasic(gnn.GNNArchitectureConfig(2, 3, [2], [3], [3], conv_layer_type="GATConv"))


def test_reference_import_paths_resolve_to_the_hip_modules():
    import gnnradarobjectdetection.gnn.gnn_models as gm
    import gnnradarobjectdetection.gnn.mpnn_layers as ml
    import gnnradarobjectdetection.graph_constructor.graph as gr
    from gnnradarobjectdetection.gnn.configs import GNNArchitectureConfig
    import radargnn_amd.gnn as rg
    assert gm.DetNetBasic is rg.DetNetBasic and ml.MPNNConv is rg.MPNNConv and gm.get_mlp is rg.get_mlp
    assert gr.GeometricGraph.__module__.startswith("radargnn_amd")
    assert GNNArchitectureConfig is rg.GNNArchitectureConfig
    # the rows added around the path
    from gnnradarobjectdetection.utils.data_handling import get_data_loaders
    from gnnradarobjectdetection.postprocessor.configs import PostProcessingConfiguration
    from gnnradarobjectdetection.postprocessor.postprocessing import BoxSuppressor, Postprocessor, PredictionExtractor
    import radargnn_amd.data as rd
    import radargnn_amd.postprocessor as rp
    assert get_data_loaders is rd.get_data_loaders and PostProcessingConfiguration is rp.PostProcessingConfiguration
    assert BoxSuppressor is rp.BoxSuppressor and PredictionExtractor is rp.PredictionExtractor and Postprocessor is rp.Postprocessor


def test_no_cpu_fallback():
    from radargnn_amd import gnn, ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        gnn.Linear(4, 4)(torch.zeros(2, 4))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.csr_by_target(torch.zeros(2, 3, dtype=torch.long), 4)
    if not torch.cuda.is_available():
        import numpy as np
        from radargnn_amd import postprocessor
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            postprocessor.PredictionExtractor.get_predicted_label(np.ones((3, 4), dtype=np.float32))
        from radargnn_amd.graph_constructor import GeometricGraph
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            GeometricGraph().build(np.random.rand(5, 2), "knn", k=1)


def test_graph_build_edge_cases_that_need_no_device():
    import numpy as np
    from radargnn_amd.graph_constructor import GeometricGraph, GraphConstructionConfiguration
    g = GeometricGraph()
    g.build(np.zeros((1, 2)), "knn", k=1)                                    # graph.py:45: nothing happens
    assert g.E is None and g.A is None
    g.build(np.zeros((5, 2)), "bogus")
    assert g.E is None
    with pytest.raises(Exception, match="Invalid graph construction algorithm"):
        GraphConstructionConfiguration("hnsw", {"k": 1}, [], [], "directed", "X")
    c = GraphConstructionConfiguration("knn", {"k": 7, "r": 2}, ["rcs"], ["relative_position"], "directed", "X")
    assert c.k == 7 and c.r is None
    g.add_node_features(np.ones((3, 1)))
    with pytest.raises(Exception, match="Feature dimension not compatible"):
        g.add_node_features(np.ones((4, 1)))


def test_shard_range_is_a_partition():
    from radargnn_amd.frames import shard_range
    for n in (0, 1, 7, 64, 1000):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_fold_cache_key_sees_swapped_parameter_storage():
    """nn.Module._apply (.to(device), .float()) replaces ``param.data`` under the same Parameter object and version counter;
    the folded-weight cache key has to change with it."""
    from radargnn_amd.gnn import mpnn_layers as ml
    p = torch.nn.Parameter(torch.ones(4, 4))
    k0 = ml._cache_key((p,))
    assert ml._same_key(k0, ml._cache_key((p,)))
    p.data = p.data.clone()
    assert not ml._same_key(k0, ml._cache_key((p,)))
    k1 = ml._cache_key((p,))
    with torch.no_grad():
        p.add_(1.0)
    assert not ml._same_key(k1, ml._cache_key((p,)))


def test_bench_decides_when_to_launch_its_own_ranks():
    """bench.py --gpus N: re-run under torch.distributed.run with N ranks unless a launcher already made this process a rank."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    assert bench.self_launch_command(1, {}, ["--gpus", "1"]) is None
    assert bench.self_launch_command(8, {"WORLD_SIZE": "8", "RANK": "3"}, ["--gpus", "8"]) is None      # the driver's torchrun line
    cmd = bench.self_launch_command(8, {}, ["--gpus", "8", "--steps", "5"], script="/x/bench.py", port=29600)
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=8" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29600"
    assert cmd[-5:] == ["/x/bench.py", "--gpus", "8", "--steps", "5"]
    assert bench.self_launch_command(1, {"RGNN_BENCH_SELF_LAUNCH": "1"}, ["--gpus", "1"]) is not None   # the launcher path on 1 GPU


def test_stage_frames_lays_a_batch_out_in_one_block():
    """rgnn_stage_frames (host only): X | V | rcs | timestamp | frame_ptr back to back; empty frames, arrays that need a conversion,
    a block that is too small."""
    import numpy as np
    from radargnn_amd import _lib, frames as fr, synthetic
    host = [synthetic.nuscenes_frame(3), synthetic.RadarFrame(np.zeros((0, 2)), np.zeros((0, 2)), np.zeros((0, 1)), np.zeros((0, 1))),
            synthetic.radarscenes_frame(1), synthetic.nuscenes_frame(4)]
    # a frame whose arrays are float32 / strided: converted on every call, never cached
    odd = synthetic.nuscenes_frame(5)
    wide = np.zeros((odd.n, 4))
    wide[:, ::2] = odd.X
    odd = synthetic.RadarFrame(wide[:, ::2], odd.V.astype(np.float32), odd.rcs, odd.timestamp)
    host.append(odd)
    keep: list = []
    table = np.array([fr.FrameStreamer._addresses(f, keep) for f in host], dtype=np.int64)
    assert len(keep) == 4 and not hasattr(odd, "_rgnn_addr") and hasattr(host[0], "_rgnn_addr")
    assert fr.FrameStreamer._addresses(host[0], keep) == tuple(table[0])           # cached: same addresses
    sizes, addr = np.ascontiguousarray(table[:, 4]), np.ascontiguousarray(table[:, :4])
    n, b = int(sizes.sum()), len(host)
    need = 48 * n + 8 * (b + 1)
    block = torch.full((need + 64,), 0xAB, dtype=torch.uint8)
    assert _lib.lib.rgnn_stage_frames(b, addr.ctypes.data, sizes.ctypes.data, block.data_ptr(), need - 8) == -1
    assert b"block too small" in _lib.lib.rgnn_last_error()
    assert _lib.lib.rgnn_stage_frames(b, addr.ctypes.data, sizes.ctypes.data, block.data_ptr(), need) == 0
    X, V, r, t, ptr = fr.FrameStreamer.views(block, n, b)
    cat, want_ptr = synthetic.concat_frames(host)
    assert np.array_equal(X.numpy(), cat.X) and np.array_equal(V.numpy(), np.asarray(cat.V, dtype=np.float64))
    assert np.array_equal(r.numpy(), cat.rcs.reshape(-1)) and np.array_equal(t.numpy(), cat.timestamp.reshape(-1))
    assert np.array_equal(ptr.numpy(), want_ptr)
    assert bool((block[need:] == 0xAB).all())                                      # nothing written past the block
    assert _lib.lib.rgnn_stage_frames(0, None, None, block.data_ptr(), 8) == 0 and int(block[:8].view(torch.int64)[0]) == 0


def test_bench_derives_traffic_from_a_counter_file(tmp_path):
    """bench.live_traffic's arithmetic on rocprofv3's per-dispatch counter rows: FETCH_SIZE (KB) doubled + WRITE_SIZE (KB) per launch; the
    dense summary takes k_linear_dma<TN >= 3, ., FMT >= 1> only; a step = four edge-kernel launches."""
    import bench
    rows = [("void (anonymous namespace)::k_linear_dma<7, true, 1, 8>((anonymous namespace)::LinParams)", 100.0, 50.0),
            ("void (anonymous namespace)::k_linear_dma<7, true, 1, 8>((anonymous namespace)::LinParams)", 300.0, 150.0),
            ("void (anonymous namespace)::k_linear_dma<2, true, 1, 8>((anonymous namespace)::LinParams)", 1000.0, 1000.0),     # N = 64: not in the set
            ("void (anonymous namespace)::k_linear_dma<5, false, 0, 8>((anonymous namespace)::LinParams)", 1000.0, 1000.0),    # bf16x3 warm-up form
            ("k_grid_frame<2>(double const*)", 10.0, 20.0)]
    rows += [("void (anonymous namespace)::k_mpnn_win<true>((anonymous namespace)::WinParams)", 200.0, 100.0)] * 8             # two steps
    per = {}
    for counter, col in (("FETCH_SIZE", 1), ("WRITE_SIZE", 2)):
        path = tmp_path / f"{counter}.csv"
        with open(path, "w") as f:
            f.write('"Dispatch_Id","Kernel_Name","Counter_Name","Counter_Value"\n')
            for i, r in enumerate(rows):
                f.write(f'{i},"{r[0]}","{counter}",{r[col]}\n')
                f.write(f'{i},"{r[0]}","SOMETHING_ELSE",7\n')
        bench._read_counter_file(str(path), counter, per)
    bench.LIVE_PMC.clear()
    try:
        assert bench._summarise_counters(per) is None
        lin, mp, step = (bench.LIVE_PMC[k] for k in ("pmc_linear_summary.json", "pmc_mpnn_summary.json", "r05_step_traffic.json"))
        assert lin["hbm_bytes_per_launch"] == ((2 * 100 + 50) + (2 * 300 + 150)) / 2 * 1024 and "THIS run" in lin["live"]
        assert mp["hbm_bytes_per_launch"] == (2 * 200 + 100) * 1024 and mp["edge_kernel"] == "k_mpnn_win"
        total = sum(2 * r[1] + r[2] for r in rows) * 1024
        assert step["steps_in_the_profiled_command"] == 2.0 and abs(step["hbm_bytes_per_step"] - total / 2) < 1e-6
        assert bench._pmc_summary("pmc_linear_summary.json") is lin
        assert bench._summarise_counters({"k_grid_frame<2>": {"FETCH_SIZE": [1.0]}}) == "no edge kernel in the counter file"
    finally:
        bench.LIVE_PMC.clear()


def test_distance_basis_is_padded_to_a_compiled_width():
    import numpy as np
    from radargnn_amd.graph_constructor.graph import _distance_basis
    rng = np.random.default_rng(0)
    for w, to in ((1, 2), (2, 2), (3, 4), (4, 4), (5, 8), (7, 8), (8, 8)):
        X = rng.normal(size=(9, w))
        B = _distance_basis(X)
        assert B.shape == (9, to) and np.array_equal(B[:, :w], X) and not B[:, w:].any()
    with pytest.raises(ValueError, match="1 to 8 columns"):
        _distance_basis(rng.normal(size=(4, 9)))
