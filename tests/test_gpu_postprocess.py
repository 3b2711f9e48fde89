"""Post-processor front half on the device (csrc/postprocess.hip through radargnn_amd.postprocessor) against the
reference-generated vectors and the oracle.  Integer results (labels, keep flags, kept order) bit-exact; scores bit-exact
(they are copies of float32 inputs); box corners float64 within 1e-9 absolute on coordinates of size ~100 (device libm
sin / cos / atan2 differ from the host's in the last ulps)."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import postprocess_oracle as O

pytestmark = pytest.mark.gpu
GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "postprocess_*.npz")))
ATOL = 1e-9


@pytest.fixture(scope="module")
def P():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from radargnn_amd import postprocessor
    return postprocessor


def config_of(P, g):
    return P.PostProcessingConfiguration(split="test", iou_for_nms=0.3, min_object_score={f"c{i}": float(v) for i, v in enumerate(g["min_scores"])},
                                         max_score_for_background=float(g["max_bg"]), bg_index=int(g["bg_index"]),
                                         bb_invariance=str(g["invariance"]), adapt_orientation_angle=bool(g["adapt"]))


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[12:-4] for p in GOLDEN])
def test_matches_reference_vectors(P, path):
    g = np.load(path)
    cfg = config_of(P, g)
    boxes, scores, labels = P.PredictionExtractor.get_absolute_object_bounding_box_predictions(g["prob"], g["bb"], g["pos"], cfg)
    kept = g["kept"]
    assert len(boxes) == len(kept) and boxes.is_aligned == (g["bb"].shape[1] == 4)
    assert np.array_equal(labels.cpu().numpy(), g["labels"][kept]) and labels.dtype == torch.float64
    assert np.array_equal(scores.cpu().numpy(), g["scores"][kept])
    np.testing.assert_allclose(boxes.corners.cpu().numpy(), g["corners"], rtol=0, atol=ATOL)
    np.testing.assert_allclose(boxes.two_point().cpu().numpy(), g["two_point"], rtol=0, atol=ATOL)
    assert np.array_equal(P.PredictionExtractor.get_predicted_label(g["prob"]).cpu().numpy(), g["labels"])
    assert np.array_equal(P.PredictionExtractor.get_prediction_scores(g["prob"]).cpu().numpy(), g["scores"])
    assert np.array_equal(P.PredictionExtractor.get_clutter_scores(g["prob"], cfg.bg_index).cpu().numpy(),
                          g["prob"][:, cfg.bg_index].reshape(-1, 1))
    first = boxes[0]
    assert first.corners.shape == (4, 2) and first.is_rotated == (g["bb"].shape[1] == 5)
    assert sum(1 for _ in boxes) == len(kept)


@pytest.mark.parametrize("width,inv,adapt", [(4, "translation", False), (5, "translation", True), (5, "none", False), (5, "en", False)])
def test_large_random_against_oracle(P, width, inv, adapt):
    rng = np.random.default_rng(width * 7 + len(inv))
    n, k = 5000, 6
    logits = rng.normal(size=(n, k)) * 2
    prob = (np.exp(logits) / np.exp(logits).sum(1, keepdims=True)).astype(np.float32)
    prob[::97] = np.float32(1.0 / k)
    pos = rng.uniform(-50, 100, size=(n, 2)).astype(np.float32)
    bb = rng.normal(size=(n, width)).astype(np.float32)
    bb[:, 2:4] = np.abs(bb[:, 2:4]) + 0.5
    if width == 5:
        bb[:, 4] = rng.uniform(-1.5, 1.5, size=n) if adapt else rng.uniform(0, np.pi, size=n)
    cfg = P.PostProcessingConfiguration(split="t", iou_for_nms=0.3, min_object_score={"a": 0.35, "b": 0.5, "c": 0.3, "d": 0.45, "e": 0.25},
                                        max_score_for_background=0.35, bg_index=5, bb_invariance=inv, adapt_orientation_angle=adapt)
    label, score, keep, corners = P.decode(prob, bb, pos, cfg)
    nn = None
    if inv == "en":
        from sklearn.neighbors import NearestNeighbors
        nn = NearestNeighbors(n_neighbors=2).fit(pos.astype(np.float64)).kneighbors(pos.astype(np.float64))[1][:, 1]
    exp_c, exp_s, exp_l, kept = O.absolute_object_boxes(prob, bb, pos, 5, 0.35, [0.35, 0.5, 0.3, 0.45, 0.25], inv, adapt, nn_index=nn)
    assert np.array_equal(np.nonzero(keep.cpu().numpy())[0], kept)
    assert np.array_equal(label.cpu().numpy().astype(np.float64).reshape(-1, 1), O.predicted_label(prob))
    assert np.array_equal(score.cpu().numpy()[kept].reshape(-1, 1).astype(np.float64), exp_s)
    np.testing.assert_allclose(corners.cpu().numpy()[kept], exp_c, rtol=0, atol=ATOL)


def test_device_inputs_batches_and_errors(P):
    """CUDA tensors in (nothing copied to the host), a batch of frames for the en representation (neighbours searched per
    frame), and the reference-side errors."""
    rng = np.random.default_rng(0)
    sizes = [40, 2, 70]
    n = sum(sizes)
    prob = torch.softmax(torch.from_numpy(rng.normal(size=(n, 6))).float(), 1).cuda()
    bb = torch.from_numpy(rng.normal(size=(n, 5))).float().abs().cuda()
    pos_parts = [rng.uniform(0, 50, size=(s, 2)).astype(np.float32) for s in sizes]
    pos = torch.from_numpy(np.concatenate(pos_parts)).cuda()
    cfg = P.PostProcessingConfiguration(split="t", iou_for_nms=0.3, min_object_score={}, max_score_for_background=1.1, bg_index=5,
                                        bb_invariance="en")
    ptr = torch.tensor(np.concatenate(([0], np.cumsum(sizes))), dtype=torch.int64).cuda()
    _, _, _, corners = P.decode(prob, bb, pos, cfg, frame_ptr=ptr)
    off = 0
    for s, pp in zip(sizes, pos_parts):
        _, _, _, c = P.decode(prob[off:off + s], bb[off:off + s], pos[off:off + s], cfg)
        assert torch.equal(c, corners[off:off + s])
        off += s
    with pytest.raises(ValueError, match="n_neighbors"):
        P.decode(prob[:1], bb[:1], pos[:1], cfg)
    with pytest.raises(ValueError):
        P.decode(prob, bb[:, :3], pos, cfg)
    empty = P.PredictionExtractor.get_absolute_object_bounding_box_predictions(prob[:0], bb[:0], pos[:0], cfg)
    assert len(empty[0]) == 0 and empty[1].shape == (0, 1)


# ---- box representations + non-maximum suppression -------------------------------------------------------------------
@pytest.mark.parametrize("path", [p for p in GOLDEN if "rot" in p], ids=lambda p: os.path.basename(p)[12:-4])
def test_box_representations_match_reference_vectors(P, path):
    from radargnn_amd import ops
    g = np.load(path)
    tp, rot = ops.box_representations(torch.from_numpy(g["corners"]).cuda())
    np.testing.assert_allclose(tp.cpu().numpy(), g["two_point"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(rot.cpu().numpy(), g["rotated_repr"], rtol=0, atol=1e-9)


def random_rotated(rng, m, extent):
    return np.stack((rng.uniform(-extent, extent, m), rng.uniform(-extent, extent, m), rng.uniform(1, 6, m), rng.uniform(0.5, 3, m),
                     rng.uniform(0, 180, m)), axis=1)


def test_nms_rotated_reference_known_answer_on_device(P):
    """test/test_postprocessor.py:8-35 of the reference, through rgnn_nms."""
    from radargnn_amd import ops
    boxes = torch.tensor([[1, 2, 1, 1, 90], [1, 2.9, 1, 1, 90]], dtype=torch.float64).cuda()
    scores = torch.tensor([0.2, 0.7], dtype=torch.float64).cuda()
    iou = 0.1 / (2 - 0.1)
    assert ops.nms(boxes, scores, iou - 0.01, rotated=True).tolist() == [1]
    assert ops.nms(boxes, scores, iou + 0.01, rotated=True).tolist() == [1, 0]


def test_nms_rotated_uses_detectron2_corner_convention_on_device(P):
    """Same hand-calculated case as tests/test_postprocess_oracle.py: long side along (cos t, -sin t)."""
    from radargnn_amd import ops
    boxes = torch.tensor([[0, 0, 4, 0.2, 45.0], [1, 1, 4, 0.4, 45.0], [1, -1, 4, 0.4, 45.0]], dtype=torch.float64).cuda()
    scores = torch.tensor([0.9, 0.8, 0.7], dtype=torch.float64).cuda()
    assert ops.nms(boxes, scores, 0.2, rotated=True).tolist() == [0, 1]
    inter = (4 - 2 ** 0.5) * 0.2
    iou = inter / (0.8 + 1.6 - inter)
    assert ops.nms(boxes, scores, iou + 0.01, rotated=True).tolist() == [0, 1, 2]
    assert ops.nms(boxes, scores, iou - 0.01, rotated=True).tolist() == [0, 1]


@pytest.mark.parametrize("m,extent,thr", [(1, 5, 0.3), (64, 6, 0.3), (65, 8, 0.1), (150, 10, 0.3), (200, 6, 0.5)])
def test_nms_rotated_matches_oracle(P, m, extent, thr):
    from radargnn_amd import ops
    rng = np.random.default_rng(m)
    boxes = random_rotated(rng, m, extent)
    scores = rng.uniform(0, 1, m)
    scores[m // 2:] = np.round(scores[m // 2:], 1)                      # ties: order must be the stable descending sort
    got = ops.nms(torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda(), thr, rotated=True)
    assert got.tolist() == O.nms_rotated(boxes, scores, thr).tolist()


@pytest.mark.parametrize("m,extent,thr", [(1, 5, 0.3), (64, 6, 0.3), (129, 8, 0.1), (700, 20, 0.3), (1000, 15, 0.5)])
def test_nms_aligned_matches_oracle(P, m, extent, thr):
    from radargnn_amd import ops
    rng = np.random.default_rng(m + 1)
    lo = rng.uniform(0, extent, (m, 2)).astype(np.float32)
    boxes = np.concatenate((lo, lo + rng.uniform(0.5, 5, (m, 2)).astype(np.float32)), axis=1)
    scores = rng.uniform(0, 1, m).astype(np.float32)
    scores[m // 2:] = np.round(scores[m // 2:], 1)
    got = ops.nms(torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda(), thr, rotated=False)
    assert got.tolist() == O.nms_aligned(boxes, scores, thr).tolist()


def test_nms_large_properties(P):
    """M = 6000 rotated boxes (more than the oracle finishes quickly): suppression is idempotent, the survivors come by
    descending score, and a sample of survivor pairs checked with the oracle's IoU stays below the threshold."""
    from radargnn_amd import ops
    rng = np.random.default_rng(5)
    m, thr = 6000, 0.3
    boxes = random_rotated(rng, m, 60)
    scores = rng.uniform(0, 1, m)
    b, s = torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda()
    keep = ops.nms(b, s, thr, rotated=True)
    k = keep.cpu().numpy()
    assert 0 < len(k) < m and np.all(np.diff(scores[k]) <= 0) and len(set(k.tolist())) == len(k)
    again = ops.nms(b[keep], s[keep], thr, rotated=True)
    assert again.tolist() == list(range(len(k)))
    near = 0
    for a in range(0, len(k), 37):
        d = np.hypot(boxes[k, 0] - boxes[k[a], 0], boxes[k, 1] - boxes[k[a], 1])
        for c in np.nonzero((d < 4) & (np.arange(len(k)) != a))[0][:5]:
            assert O.iou_rotated(boxes[k[a]], boxes[k[c]]) < thr
            near += 1
    assert near > 50
    # every suppressed box overlaps some kept box with a higher (or equal) score
    gone = np.setdiff1d(np.arange(m), k)[:40]
    for j in gone:
        d = np.hypot(boxes[k, 0] - boxes[j, 0], boxes[k, 1] - boxes[j, 1])
        cand = k[(d < 8) & (scores[k] >= scores[j])]
        assert any(O.iou_rotated(boxes[i], boxes[j]) >= thr for i in cand)


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[12:-4] for p in GOLDEN])
def test_box_suppressor_flow(P, path):
    """decode -> apply_nms like Postprocessor.process_one_raw_prediction (postprocessing.py:199-214), against the oracle's
    restatement of the same flow (negative coordinates present: both shift branches run)."""
    g = np.load(path)
    cfg = config_of(P, g)
    boxes, scores, labels = P.PredictionExtractor.get_absolute_object_bounding_box_predictions(g["prob"], g["bb"], g["pos"], cfg)
    kept_boxes, kept_scores, kept_labels = P.BoxSuppressor.apply_nms(boxes, scores, labels, 0.2)
    corners, sc, lb = g["corners"], g["scores"][g["kept"]], g["labels"][g["kept"]]
    if g["bb"].shape[1] == 5:
        mat = O.rotated_representation(corners)
        if mat[:, :2].min() < 0:
            mat[:, :2] += abs(mat[:, :2].min()) + 100
        keep = O.nms_rotated(mat, sc[:, 0], 0.2)
        np.testing.assert_allclose(kept_boxes.corners.cpu().numpy(), corners[keep], rtol=0, atol=ATOL)
        assert np.array_equal(kept_scores.cpu().numpy(), sc[keep])
    else:
        mat = O.two_point(corners)
        shift = abs(mat.min()) + 100 if mat.min() < 0 else 0
        m32 = (mat + shift).astype(np.float32)
        keep = O.nms_aligned(m32, sc[:, 0].astype(np.float32), 0.2)
        back = m32[keep] - np.float32(shift)
        exp = np.stack((back[:, [0, 1]], back[:, [0, 3]], back[:, [2, 1]], back[:, [2, 3]]), axis=1)
        np.testing.assert_allclose(kept_boxes.corners.cpu().numpy(), exp, rtol=0, atol=1e-4)     # float32 at ~200 after the shift
        assert np.array_equal(kept_scores.cpu().numpy(), sc[keep].astype(np.float32))
    assert np.array_equal(kept_labels.cpu().numpy(), lb[keep]) and len(kept_boxes) == len(keep) and 0 < len(keep) <= len(corners)


def test_pipeline_loader_model_softmax_decode_nms(P):
    """The whole inference chain on the device -- resident graphs -> DataLoader batch -> DetNetBasic -> softmax ->
    Postprocessor.process_batch -- against the same chain assembled from the oracles, frame by frame
    (inference.py:48-68 + postprocessing.py:23-79)."""
    from radargnn_amd import data as D, gnn, ops
    from oracle import gnn_oracle
    cfg = gnn.GNNArchitectureConfig(node_feature_dimension=5, edge_feature_dimension=2, conv_layer_dimensions=[16, 8],
                                    classification_head_layer_dimensions=[6], regression_head_layer_dimensions=[8, 5],
                                    initial_node_feature_embedding=True, initial_edge_feature_embedding=True,
                                    node_feature_embedding_layer_dimensions=[8, 16], edge_feature_embedding_layer_dimensions=[4, 8],
                                    conv_layer_type="MPNNConv", batch_norm_in_mlps=False)
    torch.manual_seed(3)
    model = gnn.DetNetBasic(cfg)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model.cuda()
    g = torch.Generator().manual_seed(5)
    graphs = []
    for n, e in [(120, 700), (60, 300), (200, 1500)]:
        graphs.append(D.Data(x=torch.randn(n, 5, generator=g), edge_index=torch.randint(0, n, (2, e), generator=g),
                             edge_attr=torch.randn(e, 2, generator=g), y=torch.zeros(n, 6),
                             pos=torch.rand(n, 2, generator=g) * 40, vel=torch.zeros(n, 2)))
    batch = next(iter(D.DataLoader(graphs, batch_size=3)))
    with torch.no_grad():
        cls, bb = model(batch.x, batch.edge_index, batch.edge_attr)
    prob = ops.softmax_rows(cls)
    pcfg = P.PostProcessingConfiguration(split="t", iou_for_nms=0.1, min_object_score={c: 0.05 for c in "abcde"},
                                         max_score_for_background=0.6, bg_index=5, bb_invariance="translation")
    results = P.Postprocessor.process_batch(pcfg, batch.pos, bb, prob, batch.ptr)
    assert len(results) == 3
    # oracle chain on the host: float64 forward of the whole batch (train-mode BatchNorm spans it), then per frame
    xs = torch.cat([gr.x for gr in graphs]); ea = torch.cat([gr.edge_attr for gr in graphs])
    off = np.concatenate(([0], np.cumsum([gr.num_nodes for gr in graphs])))
    ei = torch.cat([gr.edge_index + int(o) for gr, o in zip(graphs, off)], 1)
    c64, b64 = gnn_oracle.det_net_basic(xs, ei, ea, sd, dtype=torch.float64)
    p64 = torch.softmax(c64, 1)
    assert float((prob.double().cpu() - p64).abs().max()) < 1e-5
    total = 0
    for f, (det, seg) in enumerate(results):
        a, b = int(off[f]), int(off[f + 1])
        # the oracle post-processes the DEVICE's float32 outputs (the filter thresholds act on float32 values)
        pf, bf, xf = prob[a:b].cpu().numpy(), bb[a:b].cpu().numpy(), graphs[f].pos.numpy()
        corners, sc, lb, kept = O.absolute_object_boxes(pf, bf, xf, 5, 0.6, [0.05] * 5, "translation", False)
        mat = O.rotated_representation(corners)
        if len(mat) and mat[:, :2].min() < 0:
            mat[:, :2] += abs(mat[:, :2].min()) + 100
        keep = O.nms_rotated(mat, sc[:, 0], 0.1) if len(mat) else np.zeros(0, dtype=np.int64)
        assert len(det["boxes"]) == len(keep)
        np.testing.assert_allclose(det["boxes"].corners.cpu().numpy(), corners[keep], rtol=0, atol=1e-6)
        assert np.array_equal(det["labels"].cpu().numpy(), lb[keep][:, 0]) and np.array_equal(det["scores"].cpu().numpy(), sc[keep][:, 0])
        assert np.array_equal(seg["labels"].cpu().numpy(), O.predicted_label(pf)[:, 0])
        assert seg["pos"].shape == (b - a, 2) and seg["clutter_scores"].shape == (b - a,)
        total += len(keep)
    assert total > 0


@pytest.mark.parametrize("m", [1, 5, 4096, 5000, 20000, 131072])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_score_order_matches_a_stable_descending_sort(m, dtype):
    """rgnn_sort_scores (the order rgnn_nms suppresses in; postprocessor/postprocessing.py:336-435 leaves it to torchvision /
    detectron2): ids by descending score, ties by ascending id, NaN first -- bit-for-bit what a stable descending library sort
    returns, across one LDS chunk, several chunks and the 131 072-box maximum, with heavy ties and negative / infinite scores."""
    from radargnn_amd import ops
    g = torch.Generator().manual_seed(m)
    s = torch.randn(m, generator=g, dtype=torch.float64)
    s[torch.rand(m, generator=g) < 0.3] = 0.25                    # ties
    s[torch.rand(m, generator=g) < 0.05] = -0.0
    s[torch.rand(m, generator=g) < 0.05] = 0.0
    if m > 4:
        s[1], s[2], s[3] = float("inf"), float("-inf"), float("nan")
    s = s.to(dtype).cuda()
    got = ops.sort_scores(s)
    exp = torch.sort(s, descending=True, stable=True).indices
    assert got.dtype == torch.int64 and torch.equal(got, exp)
