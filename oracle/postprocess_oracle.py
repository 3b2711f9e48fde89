"""CPU oracle for the post-processor front half.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module; the
product (``radargnn_amd``) never does.

numpy restatement (per-node Python loops, like the reference) of
* ``PredictionExtractor.get_predicted_label / get_prediction_scores / get_clutter_scores``
  (src/gnnradarobjectdetection/postprocessor/postprocessing.py:177-196),
* ``PredictionExtractor.get_absolute_object_bounding_box_predictions`` (postprocessing.py:198-319),
* the box classes it instantiates (src/gnnradarobjectdetection/preprocessor/bounding_box.py:21-66 absolute rotated, :68-153
  E(n)-invariant, :156-199 relative rotated, :275-312 relative aligned, :566-589 angle inversion),
* ``torchvision.ops.nms`` as called by ``BoxSuppressor.__apply_nms_aligned`` (postprocessing.py:386-431).

Pinned: tests/golden/postprocess_*.npz are produced by tests/golden/make_postprocess_golden.py, which executes the
reference's own box classes (loaded from /root/reference by file path) inside a restated copy of the filter loop; this
oracle is checked against them in tests/test_oracle_golden.py.  Arithmetic: float64 on the float32 predictions, which is
what numpy 1.x scalar promotion gives the reference (numpy 2 keeps float32; the generator casts to float64 first).
The nearest neighbour for the "en" representation comes from sklearn, as in the reference (postprocessing.py:233-237).
torchvision and detectron2 are not installed here: the NMS part restates their published CPU kernels (greedy by
descending score; torchvision.ops.nms suppresses IoU > threshold in float32, detectron2 v0.6 nms_rotated suppresses
IoU >= threshold in the dtype given, float64 here) -- PARITY UNPINNED for that part except for the reference's own
known-answer test (test/test_postprocessor.py:8-35), which tests/test_postprocess_oracle.py repeats.  The rotated IoU is
computed the way detectron2 does (edge intersection points + vertices inside the other box, convex hull, area) with
scipy's hull -- deliberately NOT the half-plane clipping the HIP kernel uses, so the two check each other.
"""
from __future__ import annotations

from math import atan2
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
from scipy.spatial import ConvexHull, QhullError


def predicted_label(prob: np.ndarray) -> np.ndarray:
    labels = np.zeros([prob.shape[0], 1])
    for i in range(prob.shape[0]):
        vec = prob[i, :]
        labels[i, 0] = int(np.where(vec == np.max(vec))[0][0])
    return labels


def prediction_scores(prob: np.ndarray) -> np.ndarray:
    scores = np.zeros([prob.shape[0], 1])
    for i in range(prob.shape[0]):
        scores[i, 0] = np.max(prob[i, :])
    return scores


def removal_indices(prob: np.ndarray, bg_index: int, max_score_for_background: float,
                    min_object_score: Sequence[float]) -> np.ndarray:
    labels, scores = predicted_label(prob), prediction_scores(prob)
    clutter = prob[:, bg_index].reshape(prob.shape[0], 1)
    rm = np.concatenate((np.where(clutter >= np.float32(max_score_for_background))[0], np.where(labels == bg_index)[0]))
    for i, min_score in enumerate(min_object_score):
        rm = np.concatenate((rm, np.where((scores <= min_score) & (labels == i))[0]))
    return np.unique(rm)


def _rotated_corners(cx, cy, l, w, theta_deg) -> np.ndarray:
    orig = np.array([[l / 2, w / 2], [l / 2, -w / 2], [-l / 2, -w / 2], [-l / 2, w / 2]], dtype=np.float64)
    rad = (theta_deg * np.pi) / 180
    rot = np.array([[np.cos(rad), -np.sin(rad)], [np.sin(rad), np.cos(rad)]])
    return np.matmul(rot, orig.T).T + np.array([cx, cy]).reshape(1, 2)


def invert_angle_adaption(t: float) -> float:
    t = max(min(t, 1), -1)
    u = np.arcsin(t)
    return u + np.pi if u < 0 else u


def decode_box(bb: np.ndarray, point: np.ndarray, nn_point: Optional[np.ndarray], invariance: str, adapt_angle: bool) -> np.ndarray:
    """One relative box -> its four absolute corners [4, 2] (float64)."""
    bb = np.asarray(bb, dtype=np.float64)
    point = np.asarray(point, dtype=np.float64)
    if bb.shape[0] == 4:
        cx, cy, dx, dy = point[0] + bb[0], point[1] + bb[1], bb[2], bb[3]
        return np.array([[dx / 2, dy / 2], [dx / 2, -dy / 2], [-dx / 2, -dy / 2], [-dx / 2, dy / 2]]) + np.array([[cx, cy]])
    if invariance != "en":
        theta = (invert_angle_adaption(bb[4]) if adapt_angle else bb[4]) * 180 / np.pi
        if invariance == "translation":
            return _rotated_corners(point[0] + bb[0], point[1] + bb[1], bb[2], bb[3], theta)
        return _rotated_corners(bb[0], bb[1], bb[2], bb[3], theta)
    nn_point = np.asarray(nn_point, dtype=np.float64)
    d, th_pc_rel, l, w, th_dir_rel = bb[0], bb[1] * 180 / np.pi, bb[2], bb[3], bb[4] * 180 / np.pi
    v = (nn_point - point).reshape(2, 1)
    vn = v / np.linalg.norm(v)
    th_nn = atan2(vn[1, 0], vn[0, 0]) * 180 / np.pi
    th_dir = np.round(th_dir_rel + th_nn, 5)
    while th_dir < 0:
        th_dir = 360 + th_dir
    while th_dir >= 180:
        th_dir = th_dir - 180
    th_pc = th_pc_rel + th_nn
    while th_pc > 360:
        th_pc = th_pc - 360
    xc, yc = d * np.cos((th_pc * np.pi) / 180), d * np.sin((th_pc * np.pi) / 180)
    return _rotated_corners(point[0] + xc, point[1] + yc, l, w, th_dir)


def absolute_object_boxes(prob: np.ndarray, bb: np.ndarray, pos: np.ndarray, bg_index: int, max_score_for_background: float,
                          min_object_score: Sequence[float], invariance: str, adapt_angle: bool,
                          nn_index: Optional[np.ndarray] = None) -> Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]:
    """-> (corners [M, 4, 2] f64, scores [M, 1], labels [M, 1], kept node ids [M]) in node order (np.delete keeps it)."""
    labels, scores = predicted_label(prob), prediction_scores(prob)
    rm = removal_indices(prob, bg_index, max_score_for_background, min_object_score)
    kept = np.setdiff1d(np.arange(prob.shape[0]), rm)
    corners = np.zeros((len(kept), 4, 2))
    for r, i in enumerate(kept):
        nn = pos[nn_index[i]] if (invariance == "en" and bb.shape[1] == 5) else None
        corners[r] = decode_box(bb[i], pos[i], nn, invariance, adapt_angle)
    return corners, scores[kept], labels[kept], kept


def two_point(corners: np.ndarray) -> np.ndarray:
    """[M, 4, 2] -> [x_min, y_min, x_max, y_max] per box (BoundingBox.get_to_two_point_representation)."""
    return np.concatenate((corners.min(axis=1), corners.max(axis=1)), axis=1)


def nms_aligned(boxes: np.ndarray, scores: np.ndarray, iou: float) -> np.ndarray:
    """torchvision.ops.nms (CPU kernel) in float32: indices kept, by descending score (stable for ties)."""
    boxes = boxes.astype(np.float32)
    scores = scores.astype(np.float32)
    order = np.argsort(-scores, kind="stable")
    x1, y1, x2, y2 = boxes[:, 0], boxes[:, 1], boxes[:, 2], boxes[:, 3]
    areas = (x2 - x1) * (y2 - y1)
    dead = np.zeros(len(boxes), dtype=bool)
    keep = []
    for a, i in enumerate(order):
        if dead[i]:
            continue
        keep.append(i)
        for j in order[a + 1:]:
            if dead[j]:
                continue
            w = max(np.float32(0), min(x2[i], x2[j]) - max(x1[i], x1[j]))
            h = max(np.float32(0), min(y2[i], y2[j]) - max(y1[i], y1[j]))
            inter = np.float32(w * h)
            if inter / (areas[i] + areas[j] - inter) > np.float32(iou):
                dead[j] = True
    return np.array(keep, dtype=np.int64)


def rotated_representation(corners: np.ndarray) -> np.ndarray:
    """[M, 4, 2] -> [x, y, l, w, theta(deg, 0..180)] per box (BoundingBox.get_absolute_rotated_box_representations,
    preprocessor/bounding_box.py:467-540)."""
    out = np.empty([len(corners), 5])
    for i, c in enumerate(corners):
        p1, p2, p3, p4 = (c[k, :].reshape(1, 2) for k in range(4))
        d = [np.linalg.norm(p1 - p2), np.linalg.norm(p1 - p3), np.linalg.norm(p1 - p4)]
        w = min(d)
        d.remove(w)
        l = min(d)
        ctr = ((p1 + p2 + p3 + p4) / 4).reshape(1, 2)
        if l == np.linalg.norm(p1 - p2):
            v = (p1 - p2).reshape(2, 1)
        elif l == np.linalg.norm(p1 - p3):
            v = (p1 - p3).reshape(2, 1)
        elif l == np.linalg.norm(p1 - p4):
            v = (p1 - p4).reshape(2, 1)
        else:
            out[i, :] = [0, 0, 1, 1, 0]
            continue
        vn = v / np.linalg.norm(v)
        theta = atan2(vn[1, 0], vn[0, 0]) * 180 / np.pi
        if theta < 0:
            theta = 180 + theta
        out[i, :] = [ctr[0, 0], ctr[0, 1], l, w, theta]
    return out


def _rect(box: np.ndarray) -> np.ndarray:
    x, y, l, w, deg = box
    t = deg * np.pi / 180
    # detectron2's corner convention (box_iou_rotated_utils.h, get_rotated_vertices; image frame, y down): the long side
    # points along (cos t, -sin t) and pts[0] = ctr + l/2 (cos, -sin) + w/2 (sin, cos).  The reference passes its y-up
    # theta_x straight in (postprocessing.py:356-370), so its IoUs are those of boxes mirrored about their own centres.
    d, n = np.array([np.cos(t), -np.sin(t)]), np.array([np.sin(t), np.cos(t)])
    c = np.array([x, y])
    return np.array([c + l / 2 * d + w / 2 * n, c - l / 2 * d + w / 2 * n, c - l / 2 * d - w / 2 * n, c + l / 2 * d - w / 2 * n])


def _inside(p: np.ndarray, rect: np.ndarray) -> bool:
    ab, ad = rect[1] - rect[0], rect[3] - rect[0]
    ap = p - rect[0]
    return 0 <= ap @ ab <= ab @ ab and 0 <= ap @ ad <= ad @ ad


def iou_rotated(a: np.ndarray, b: np.ndarray) -> float:
    area_a, area_b = a[2] * a[3], b[2] * b[3]
    if area_a < 1e-14 or area_b < 1e-14:
        return 0.0
    shift = np.array([(a[0] + b[0]) / 2, (a[1] + b[1]) / 2, 0, 0, 0])
    ra, rb = _rect(a - shift), _rect(b - shift)
    pts = [p for p in ra if _inside(p, rb)] + [p for p in rb if _inside(p, ra)]
    for i in range(4):
        p, r = ra[i], ra[(i + 1) % 4] - ra[i]
        for j in range(4):
            q, s = rb[j], rb[(j + 1) % 4] - rb[j]
            det = r[0] * s[1] - r[1] * s[0]
            if abs(det) <= 1e-14:
                continue
            t = ((q - p)[0] * s[1] - (q - p)[1] * s[0]) / det
            u = ((q - p)[0] * r[1] - (q - p)[1] * r[0]) / det
            if 0 <= t <= 1 and 0 <= u <= 1:
                pts.append(p + t * r)
    if len(pts) < 3:
        return 0.0
    try:
        inter = ConvexHull(np.array(pts)).volume          # 2-D hull: `volume` is the area
    except QhullError:
        return 0.0                                         # degenerate (collinear) intersection
    return float(inter / (area_a + area_b - inter))


def nms_rotated(boxes: np.ndarray, scores: np.ndarray, iou: float) -> np.ndarray:
    """detectron2 nms_rotated (CPU) on float64 boxes [x, y, l, w, theta deg]: greedy by descending score, drop IoU >= iou."""
    order = np.argsort(-scores, kind="stable")
    dead = np.zeros(len(boxes), dtype=bool)
    keep = []
    for a, i in enumerate(order):
        if dead[i]:
            continue
        keep.append(i)
        for j in order[a + 1:]:
            if not dead[j] and iou_rotated(boxes[i], boxes[j]) >= iou:
                dead[j] = True
    return np.array(keep, dtype=np.int64)
