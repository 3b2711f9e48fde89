"""The post-processing oracle (oracle/postprocess_oracle.py) against vectors produced by the reference's own box classes
(tests/golden/postprocess_*.npz, written by tests/golden/make_postprocess_golden.py).  CPU only."""
import glob
import os

import numpy as np
import pytest

from oracle import postprocess_oracle as O

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "postprocess_*.npz")))


def test_fixtures_present():
    assert len(GOLDEN) == 6


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[12:-4] for p in GOLDEN])
def test_oracle_reproduces_reference_vectors(path):
    g = np.load(path)
    prob, bb, pos = g["prob"], g["bb"], g["pos"]
    assert np.array_equal(O.predicted_label(prob), g["labels"])
    assert np.array_equal(O.prediction_scores(prob), g["scores"])
    corners, scores, labels, kept = O.absolute_object_boxes(prob, bb, pos, int(g["bg_index"]), float(g["max_bg"]),
                                                            list(g["min_scores"]), str(g["invariance"]), bool(g["adapt"]),
                                                            nn_index=g["nn_index"])
    assert np.array_equal(kept, g["kept"])
    assert np.array_equal(scores, g["scores"][g["kept"]]) and np.array_equal(labels, g["labels"][g["kept"]])
    # same float64 operations in the same order: agreement to the last few ulps of coordinates of size ~100
    np.testing.assert_allclose(corners, g["corners"], rtol=0, atol=1e-11)
    np.testing.assert_allclose(O.two_point(corners), g["two_point"], rtol=0, atol=1e-11)
    if bb.shape[1] == 5:
        np.testing.assert_allclose(O.rotated_representation(g["corners"]), g["rotated_repr"], rtol=0, atol=1e-12)


def test_first_maximum_wins_and_threshold_dtypes():
    prob = np.array([[0.25, 0.25, 0.25, 0.25], [0.1, 0.4, 0.4, 0.1]], dtype=np.float32)
    assert O.predicted_label(prob).ravel().tolist() == [0.0, 1.0]
    # score float32(0.3) = 0.30000001192...: as float64 it is > 0.3, so `score <= 0.3` does not remove it ...
    p = np.array([[0.3, 0.2, 0.25, 0.25]], dtype=np.float32)
    assert O.removal_indices(p, 3, 0.9, [0.3]).tolist() == []
    # ... while the background test compares in float32: float32(0.3) >= float32(0.3)
    p = np.array([[0.4, 0.2, 0.1, 0.3]], dtype=np.float32)
    assert O.removal_indices(p, 3, 0.3, []).tolist() == [0]


def test_nms_aligned_known_answer():
    """Two unit squares overlapping by 0.1 x 1 (the geometry of the reference's test_nms_rotated, test_postprocessor.py:8-35,
    axis-aligned): IoU = 0.1 / 1.9."""
    boxes = np.array([[0.5, 1.5, 1.5, 2.5], [0.5, 2.4, 1.5, 3.4]])
    scores = np.array([0.2, 0.7])
    iou = 0.1 / 1.9
    assert O.nms_aligned(boxes, scores, iou - 0.01).tolist() == [1]
    assert O.nms_aligned(boxes, scores, iou + 0.01).tolist() == [1, 0]


def test_nms_rotated_reference_known_answer():
    """The reference's own test (test/test_postprocessor.py:8-35): two 1 x 1 boxes at 90 degrees, centres 0.9 apart."""
    boxes = np.array([[1, 2, 1, 1, 90], [1, 2.9, 1, 1, 90]], dtype=np.float64)
    scores = np.array([0.2, 0.7])
    iou = 0.1 / (2 - 0.1)
    assert abs(O.iou_rotated(boxes[0], boxes[1]) - iou) < 1e-12
    assert O.nms_rotated(boxes, scores, iou - 0.01).tolist() == [1]
    assert O.nms_rotated(boxes, scores, iou + 0.01).tolist() == [1, 0]


def test_nms_rotated_uses_detectron2_corner_convention():
    """detectron2 (box_iou_rotated_utils.h) puts the long side of a box along (cos t, -sin t).  Hand calculation for boxes at
    45 degrees: A (4 x 0.2 at (0, 0)) and C (4 x 0.4 at (1, -1)) then lie on the same line x + y = 0, centres sqrt(2) apart:
    they share (4 - sqrt2) of their length over A's whole width, IoU = (4 - sqrt2) 0.2 / (0.8 + 1.6 - (4 - sqrt2) 0.2) =
    0.2747; B (4 x 0.4 at (1, 1)) lies on x + y = 2, sqrt(2) away from A's line: IoU(A, B) = 0.  (With the long side along
    (cos t, +sin t) it is the other way round: B overlaps A and C does not.)"""
    boxes = np.array([[0, 0, 4, 0.2, 45.0], [1, 1, 4, 0.4, 45.0], [1, -1, 4, 0.4, 45.0]], dtype=np.float64)
    inter = (4 - np.sqrt(2)) * 0.2
    assert abs(O.iou_rotated(boxes[0], boxes[2]) - inter / (0.8 + 1.6 - inter)) < 1e-12
    assert O.iou_rotated(boxes[0], boxes[1]) == 0.0
    assert O.nms_rotated(boxes, np.array([0.9, 0.8, 0.7]), 0.2).tolist() == [0, 1]


def test_iou_rotated_closed_forms():
    sq = np.array([0.0, 0.0, 2.0, 2.0, 0.0])
    assert abs(O.iou_rotated(sq, sq) - 1.0) < 1e-12
    diamond = np.array([0.0, 0.0, 2.0, 2.0, 45.0])           # octagon: 8 (sqrt2 - 1)
    inter = 8 * (np.sqrt(2) - 1)
    assert abs(O.iou_rotated(sq, diamond) - inter / (8 - inter)) < 1e-12
    assert O.iou_rotated(sq, np.array([5.0, 5.0, 1.0, 1.0, 30.0])) == 0.0
