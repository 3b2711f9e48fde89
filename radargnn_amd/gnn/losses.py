"""The trainer's loss on the device: what ``gnn/trainer.py:181-222`` computes with
``CrossEntropyLoss(weight)``, ``HuberLoss()`` and a Python loop over the nodes,

    loss = cls_loss_weight * CE(cls, label) + bb_loss_weight * mean_{i : label_i != bg} Huber(bb_true_i, bb_i),

as two kernel launches (``rgnn_detection_loss``) and, for ``backward()``, one more (``rgnn_detection_loss_bwd``)."""
from __future__ import annotations

from typing import Optional, Sequence, Tuple, Union

import torch

from .. import ops


class _DetectionLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cls, bb, y, class_weight, bg_index, delta, alpha, beta):
        out, sums = ops.detection_loss(cls.detach(), bb.detach(), y, class_weight, bg_index, delta, alpha, beta)
        ctx.save_for_backward(cls.detach(), bb.detach(), y, sums)
        ctx.class_weight = class_weight
        ctx.args = (bg_index, delta, alpha, beta)
        ctx.mark_non_differentiable(sums)
        return out[0], out[1], out[2]

    @staticmethod
    def backward(ctx, g_loss, g_cls, g_bb):
        cls, bb, y, sums = ctx.saved_tensors
        bg_index, delta, alpha, beta = ctx.args
        # the three outputs are loss = alpha lc + beta lb, lc and lb: fold the incoming gradients into the two weights
        d_cls = d_bb = None
        parts = []
        if g_loss is not None:
            parts.append((alpha, beta, g_loss))
        if g_cls is not None and bool((g_cls != 0).any()):
            parts.append((1.0, 0.0, g_cls))
        if g_bb is not None and bool((g_bb != 0).any()):
            parts.append((0.0, 1.0, g_bb))
        for a, b, g in parts:
            dc, db = ops.detection_loss_bwd(cls, bb, y, ctx.class_weight, bg_index, delta, a, b, sums, g)
            d_cls = dc if d_cls is None else d_cls + dc
            d_bb = db if d_bb is None else d_bb + db
        return d_cls, d_bb, None, None, None, None, None, None


def detection_loss(cls: torch.Tensor, bb: torch.Tensor, y: torch.Tensor, bg_index: int,
                   class_weights: Optional[Union[torch.Tensor, Sequence[float]]] = None, cls_loss_weight: float = 1.0,
                   bb_loss_weight: float = 1.0, delta: float = 1.0) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """-> (loss, loss_cls, loss_bb), 0-dim CUDA tensors; ``loss.backward()`` reaches ``cls`` and ``bb``.

    ``y``: float32 [N, 1 + W] as the reference stores it (``graph_batch.y``: class label | box, trainer.py:185-186);
    ``class_weights``: ``TrainingConfig.class_weights.values()`` (trainer.py:95,99) or None.  A batch without object nodes
    or with a NaN box term gets ``loss_bb = 0`` (trainer.py:203-217)."""
    if not cls.is_cuda:
        raise RuntimeError("detection_loss runs on the GPU (no CPU fallback)")
    w = None
    if class_weights is not None:
        w = torch.as_tensor(class_weights, dtype=torch.float32).to(cls.device).contiguous()
    y = y.to(device=cls.device, dtype=torch.float32)
    return _DetectionLoss.apply(cls, bb, y, w, int(bg_index), float(delta), float(cls_loss_weight), float(bb_loss_weight))
