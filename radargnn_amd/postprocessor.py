"""Post-processor front half on the device: mirrors of the reference's
``postprocessor/configs.py::PostProcessingConfiguration`` and ``postprocessor/postprocessing.py::PredictionExtractor``
(labels, scores, background scores, score filtering, relative -> absolute box decoding) on librgnn's
``rgnn_decode_predictions``.  Inputs may be numpy arrays (as the reference passes them) or CUDA tensors (straight
from ``frames.HotPath`` / the model heads, nothing leaves the device); results stay in HBM.

The E(n)-invariant box representation needs every point's nearest neighbour: the k = 1 use of the kNN kernel
(``ops.knn_graph``) instead of sklearn + a dense N x N ``toarray()`` (postprocessing.py:233-237).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import ops

INVARIANCE_CODES = {"none": 0, "translation": 1, "en": 2}


@dataclass
class PostProcessingConfiguration:
    """postprocessor/configs.py:5-27 (same field names and defaults)."""
    split: str = "test"
    iou_for_nms: float = 0.3
    min_object_score: Dict[str, float] = field(default_factory=dict)
    max_score_for_background: float = 1.0
    iou_for_mAP: float = 0.3
    use_point_iou: bool = False
    bg_index: int = 5
    bb_invariance: str = "translation"
    adapt_orientation_angle: bool = False
    get_mAP: bool = True
    get_confusion: bool = True
    get_segmentation_f1: bool = True
    f1_class_averaging: Optional[str] = None


class BoundingBox:
    """One absolute box as the reference's ``preprocessor/bounding_box.py::BoundingBox``: ``corners`` [4, 2] numpy."""

    def __init__(self, corners: np.ndarray, aligned: bool):
        self.corners = corners
        self.is_aligned = aligned
        self.is_rotated = not aligned


class BoundingBoxes:
    """The decoded boxes of one graph in HBM: ``corners`` float64 [M, 4, 2].  Behaves like the list of ``BoundingBox``
    objects the reference returns (``len``, indexing, iteration copy to the host on demand)."""

    def __init__(self, corners: torch.Tensor, aligned: bool):
        self.corners = corners
        self.is_aligned = aligned
        self.is_rotated = not aligned

    def __len__(self) -> int:
        return self.corners.shape[0]

    def __getitem__(self, i) -> BoundingBox:
        return BoundingBox(self.corners[i].cpu().numpy(), self.is_aligned)

    def __iter__(self):
        host = self.corners.cpu().numpy()
        return (BoundingBox(host[i], self.is_aligned) for i in range(host.shape[0]))

    def two_point(self) -> torch.Tensor:
        """[x_min, y_min, x_max, y_max] per box (``BoundingBox.get_two_point_representations``), float64 [M, 4]."""
        return torch.cat((self.corners.min(dim=1).values, self.corners.max(dim=1).values), dim=1)


def _f32_cuda(a, name: str) -> torch.Tensor:
    t = torch.as_tensor(a)
    if t.dtype != torch.float32:
        t = t.to(torch.float32)
    if not t.is_cuda:
        if not torch.cuda.is_available():
            raise RuntimeError(f"{name}: the post-processor kernels need a GPU (no CPU fallback)")
        t = t.cuda()
    return t.contiguous()


def decode(class_probability_prediction, bounding_box_predictions, pos, config: PostProcessingConfiguration,
           nn_index: Optional[torch.Tensor] = None, frame_ptr: Optional[torch.Tensor] = None):
    """Per-node results for a graph or a whole batch, everything on the device:
    -> (label int32 [N], score f32 [N], keep int32 [N], corners f64 [N, 4, 2]).  ``frame_ptr`` (int64 [B+1]): node offsets
    of the frames of a batch, so that nearest neighbours for the "en" representation are searched inside each frame."""
    prob = _f32_cuda(class_probability_prediction, "class_probability_prediction")
    bb = _f32_cuda(bounding_box_predictions, "bounding_box_predictions")
    p = _f32_cuda(pos, "pos")
    n, k = prob.shape
    if bb.shape[0] != n or p.shape != (n, 2) or bb.shape[1] not in (4, 5):
        raise ValueError("shapes: class probabilities [N, K], boxes [N, 4|5], pos [N, 2]")
    if config.bb_invariance not in INVARIANCE_CODES:
        raise ValueError(f"unknown bb_invariance {config.bb_invariance!r}")
    inv = INVARIANCE_CODES[config.bb_invariance]
    if inv == 2 and bb.shape[1] == 5 and n and nn_index is None:
        ptr = frame_ptr if frame_ptr is not None else torch.tensor([0, n], dtype=torch.int64, device=p.device)
        if int((ptr[1:] - ptr[:-1]).min()) <= 1:
            raise ValueError("Expected n_neighbors < n_samples_fit, but n_neighbors = 1, n_samples_fit = 1")   # sklearn's error
        nn_index, _, _ = ops.knn_graph(p.to(torch.float64), ptr, 1, want_edge_index=False)
        nn_index = nn_index.view(-1)
    return ops.decode_predictions(prob, bb, p, nn_index, config.bg_index, config.max_score_for_background,
                                  list(config.min_object_score.values()), inv, config.adapt_orientation_angle)


class PredictionExtractor:
    """postprocessor/postprocessing.py:166-319, same method names and result shapes ([N, 1] columns); tensors in HBM."""

    @staticmethod
    def _prob(class_probability_prediction) -> torch.Tensor:
        return _f32_cuda(class_probability_prediction, "class_probability_prediction")

    @classmethod
    def get_predicted_label(cls, class_probability_prediction) -> torch.Tensor:
        prob = cls._prob(class_probability_prediction)
        label, _ = ops.row_argmax(prob)
        return label.to(torch.float64).view(-1, 1)

    @classmethod
    def get_prediction_scores(cls, class_probability_prediction) -> torch.Tensor:
        prob = cls._prob(class_probability_prediction)
        _, score = ops.row_argmax(prob)
        return score.to(torch.float64).view(-1, 1)

    @classmethod
    def get_clutter_scores(cls, class_probability_prediction, bg_index: int) -> torch.Tensor:
        return cls._prob(class_probability_prediction)[:, bg_index].reshape(-1, 1)

    @classmethod
    def get_absolute_object_bounding_box_predictions(cls, class_probability_prediction, bounding_box_predictions, pos,
                                                     config: PostProcessingConfiguration
                                                     ) -> Tuple[BoundingBoxes, torch.Tensor, torch.Tensor]:
        """-> (boxes of the nodes that survive the score filters, their scores [M, 1] f64, their labels [M, 1] f64), in node
        order like the reference's ``np.delete``.  One host read (M) sizes the outputs."""
        label, score, keep, corners = decode(class_probability_prediction, bounding_box_predictions, pos, config)
        idx = torch.nonzero(keep, as_tuple=False).view(-1)
        aligned = torch.as_tensor(bounding_box_predictions).shape[1] == 4
        return (BoundingBoxes(corners.index_select(0, idx), aligned),
                score.index_select(0, idx).to(torch.float64).view(-1, 1),
                label.index_select(0, idx).to(torch.float64).view(-1, 1))


class BoxSuppressor:
    """postprocessor/postprocessing.py:336-431: non-maximum suppression of one graph's decoded boxes, rotated
    (detectron2 ``nms_rotated`` semantics) or aligned (``torchvision.ops.nms`` semantics) by the kind of the boxes."""

    @classmethod
    def apply_nms(cls, bounding_boxes: BoundingBoxes, box_scores: torch.Tensor, box_labels: torch.Tensor, iou_nms: float):
        """-> (boxes kept, their scores [M', 1], their labels [M', 1]), by descending score like the reference."""
        if len(bounding_boxes) == 0:
            return bounding_boxes, box_scores, box_labels
        corners = bounding_boxes.corners
        if bounding_boxes.is_rotated:
            _, mat = ops.box_representations(corners, two_point=False, rotated=True)
            lo = mat[:, :2].min()                                   # postprocessing.py:358-361: all centres made positive
            if float(lo) < 0:
                mat = mat.clone()
                mat[:, :2] += abs(float(lo)) + 100
            keep = ops.nms(mat, box_scores.reshape(-1).to(torch.float64), iou_nms, rotated=True)
            kept_boxes = BoundingBoxes(corners.index_select(0, keep), False)
            scores = box_scores.reshape(-1).to(torch.float64).index_select(0, keep).view(-1, 1)
        else:
            mat, _ = ops.box_representations(corners, two_point=True, rotated=False)
            lo = float(mat.min())                                   # postprocessing.py:391-394
            shift = abs(lo) + 100 if lo < 0 else 0.0
            mat32 = (mat + shift).to(torch.float32) if shift else mat.to(torch.float32)
            s32 = box_scores.reshape(-1).to(torch.float32)
            keep = ops.nms(mat32, s32, iou_nms, rotated=False)
            kept = mat32.index_select(0, keep)                      # the reference rebuilds the boxes from the float32 matrix
            if shift:
                kept = kept - torch.tensor(shift, dtype=torch.float32, device=kept.device)
            x0, y0, x1, y1 = kept[:, 0], kept[:, 1], kept[:, 2], kept[:, 3]
            rebuilt = torch.stack((x0, y0, x0, y1, x1, y0, x1, y1), dim=1).view(-1, 4, 2)   # corner order of :416-423
            kept_boxes = BoundingBoxes(rebuilt, True)
            scores = s32.index_select(0, keep).view(-1, 1)
        labels = box_labels.reshape(-1).index_select(0, keep).view(-1, 1)
        return kept_boxes, scores, labels


class Postprocessor:
    """postprocessor/postprocessing.py:14-79 (the prediction half): decode + NMS + the per-node segmentation outputs, as the
    two dicts the reference returns -- values are tensors in HBM instead of numpy arrays."""

    @staticmethod
    def process_one_raw_prediction(config: PostProcessingConfiguration, pos, raw_bb_pred, raw_cls_prob_pred):
        label, score, keep, corners = decode(raw_cls_prob_pred, raw_bb_pred, pos, config)
        return Postprocessor._finish(config, _f32_cuda(pos, "pos"), _f32_cuda(raw_cls_prob_pred, "cls"), label, score,
                                     keep, corners, torch.as_tensor(raw_bb_pred).shape[1] == 4)

    @staticmethod
    def _finish(config, pos, prob, label, score, keep, corners, aligned):
        idx = torch.nonzero(keep, as_tuple=False).view(-1)
        boxes = BoundingBoxes(corners.index_select(0, idx), aligned)
        scores = score.index_select(0, idx).to(torch.float64).view(-1, 1)
        labels = label.index_select(0, idx).to(torch.float64).view(-1, 1)
        boxes, scores, labels = BoxSuppressor.apply_nms(boxes, scores, labels, config.iou_for_nms)
        detection = {"boxes": boxes, "scores": scores[:, 0], "labels": labels[:, 0]}
        segmentation = {"pos": pos, "labels": label.to(torch.float64), "scores": score.to(torch.float64),
                        "clutter_scores": prob[:, config.bg_index]}
        return detection, segmentation

    @staticmethod
    def process_batch(config: PostProcessingConfiguration, pos, raw_bb_pred, raw_cls_prob_pred, ptr):
        """The same for a whole batch straight from the model (``Batch.ptr`` / ``FrameBatch.frame_ptr`` node offsets): ONE
        decode launch over all nodes (nearest neighbours for the "en" boxes searched per frame), then suppression frame by
        frame like ``Postprocessor.process`` (postprocessing.py:150-154).  -> list of (detection, segmentation) dicts."""
        ptr_dev = torch.as_tensor(ptr, dtype=torch.int64)
        if not ptr_dev.is_cuda:
            ptr_dev = ptr_dev.cuda()
        label, score, keep, corners = decode(raw_cls_prob_pred, raw_bb_pred, pos, config, frame_ptr=ptr_dev)
        pos32, prob = _f32_cuda(pos, "pos"), _f32_cuda(raw_cls_prob_pred, "cls")
        aligned = torch.as_tensor(raw_bb_pred).shape[1] == 4
        bounds = ptr_dev.cpu().tolist()
        out = []
        for a, b in zip(bounds[:-1], bounds[1:]):
            out.append(Postprocessor._finish(config, pos32[a:b], prob[a:b], label[a:b], score[a:b], keep[a:b], corners[a:b],
                                             aligned))
        return out
