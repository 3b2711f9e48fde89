"""Writes tests/golden/postprocess_*.npz: inputs and outputs of the reference's box decoding, produced by EXECUTING the
reference's own box classes (preprocessor/bounding_box.py, loaded from /root/reference by file path -- the package
__init__ imports ray / nuscenes, which are absent) inside a restated copy of the 40-line filter loop of
PredictionExtractor.get_absolute_object_bounding_box_predictions (postprocessor/postprocessing.py:198-319; its module
imports detectron2 / torchvision, absent).  Runs only in the build container; the fixtures are committed.

numpy here is 2.2 (float32 scalars stay float32 under NEP 50); the reference's environment is numpy 1.x, where the same
scalar expressions promote to float64 -- so the predictions are cast to float64 before they enter the reference classes.
"""
import importlib.util
import os
import sys
import types

import numpy as np
from sklearn.neighbors import kneighbors_graph

R = "/root/reference/src/gnnradarobjectdetection"
for name in ("gnnradarobjectdetection", "gnnradarobjectdetection.utils", "gnnradarobjectdetection.preprocessor"):
    m = types.ModuleType(name); m.__path__ = []; sys.modules[name] = m


def load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec); sys.modules[name] = mod; spec.loader.exec_module(mod)
    return mod


load("gnnradarobjectdetection.utils.math", R + "/utils/math.py")
B = load("gnnradarobjectdetection.preprocessor.bounding_box", R + "/preprocessor/bounding_box.py")
HERE = os.path.dirname(os.path.abspath(__file__))


def reference_flow(prob, bb, pos, bg_index, max_bg, min_scores, invariance, adapt):
    n = prob.shape[0]
    labels = np.zeros([n, 1]); scores = np.zeros([n, 1])
    for i in range(n):
        vec = prob[i, :]
        labels[i, 0] = int(np.where(vec == np.max(vec))[0][0])
        scores[i, 0] = np.max(vec)
    clutter = prob[:, bg_index].reshape(n, 1)
    rm = np.concatenate((np.where(clutter >= max_bg)[0], np.where(labels == bg_index)[0]), axis=0)
    for i, ms in enumerate(min_scores):
        rm = np.concatenate((rm, np.where((scores <= ms) & (labels == i))[0]), axis=0)
    rm = np.unique(rm)
    nn_index = np.zeros(n, dtype=np.int64)
    if invariance == "en" and n:
        A = kneighbors_graph(pos, 1, mode="connectivity", include_self=False).toarray()
        nn_index = np.where(A == 1)[1]
        pos_nn = np.delete(pos[nn_index], rm, axis=0)
    bbk = np.delete(bb, rm, axis=0).astype(np.float64)
    posk = np.delete(pos, rm, axis=0).astype(np.float64)
    corners = []
    for i in range(bbk.shape[0]):
        b = bbk[i]
        if bbk.shape[1] == 4:
            box = B.RelativeAlignedBoundingBox(b[0], b[1], b[2], b[3]).get_absolute_bounding_box(posk[i, 0], posk[i, 1])
        elif invariance != "en":
            theta = (B.invert_bb_orientation_angle_adaption(b[4]) if adapt else b[4]) * 180 / np.pi
            if invariance == "translation":
                box = B.RelativeRotatedBoundingBox(b[0], b[1], b[2], b[3], theta).get_absolute_bounding_box(posk[i, 0], posk[i, 1])
            else:
                box = B.AbsoluteRotatedBoundingBox(b[0], b[1], b[2], b[3], theta).get_absolute_bounding_box()
        else:
            box = B.RotationInvariantRelativeRotatedBoundingBox(b[0], b[1] * 180 / np.pi, b[2], b[3], b[4] * 180 / np.pi) \
                .get_absolute_bounding_box(posk[i, :], pos_nn[i, :].astype(np.float64))
        corners.append(box.corners)
    corners = np.array(corners).reshape(-1, 4, 2)
    kept = np.setdiff1d(np.arange(n), rm)
    two_point = B.BoundingBox.get_two_point_representations([B.BoundingBox(c, bbk.shape[1] == 4) for c in corners]) \
        if len(corners) else np.zeros((0, 4))
    rotated_repr = B.BoundingBox.get_absolute_rotated_box_representations([B.BoundingBox(c, False) for c in corners]) \
        if len(corners) else np.zeros((0, 5))
    return dict(labels=labels, scores=scores, kept=kept, corners=corners, nn_index=nn_index, two_point=two_point,
                rotated_repr=rotated_repr)


def case(name, seed, n, k, width, invariance, adapt):
    rng = np.random.default_rng(seed)
    logits = rng.normal(size=(n, k)) * 2.0
    prob = (np.exp(logits) / np.exp(logits).sum(1, keepdims=True)).astype(np.float32)
    prob[3] = prob[3, ::-1].copy(); prob[5, :] = np.float32(1.0 / k)          # an exact tie: first maximum wins
    pos = (rng.uniform(-50, 100, size=(n, 2))).astype(np.float32)
    bb = rng.normal(size=(n, width)).astype(np.float32)
    bb[:, 2:4] = np.abs(bb[:, 2:4]) * 3 + 0.5
    if width == 5:
        bb[:, 4] = rng.uniform(-1.2, 1.2, size=n) if adapt else rng.uniform(0, np.pi, size=n)
        if invariance == "en":
            bb[:, 0] = np.abs(bb[:, 0]) * 2; bb[:, 1] = rng.uniform(0, 2 * np.pi, size=n)
    bg_index = k - 1
    max_bg, min_scores = 0.4, [0.3, 0.5, 0.25, 0.6, 0.2][: k - 1]
    out = reference_flow(prob, bb, pos, bg_index, max_bg, min_scores, invariance, adapt)
    np.savez(os.path.join(HERE, f"postprocess_{name}.npz"), prob=prob, bb=bb, pos=pos, bg_index=bg_index, max_bg=max_bg,
             min_scores=np.array(min_scores), invariance=invariance, adapt=adapt, **out)
    print(name, "kept", len(out["kept"]), "of", n)


if __name__ == "__main__":
    case("aligned", 1, 60, 6, 4, "translation", False)
    case("rot_translation", 2, 60, 6, 5, "translation", False)
    case("rot_translation_adapt", 3, 60, 6, 5, "translation", True)
    case("rot_none", 4, 50, 6, 5, "none", False)
    case("rot_en", 5, 70, 6, 5, "en", False)
    case("rot_en_11cls", 6, 40, 11, 5, "en", False)
