"""CPU oracle for the trainer's loss.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module; the
product (``radargnn_amd``) never does.

Restates src/gnnradarobjectdetection/gnn/trainer.py:181-222 with the same torch modules the reference constructs
(``torch.nn.CrossEntropyLoss(weight=...)``, ``torch.nn.HuberLoss()``, trainer.py:99,106) and the same per-node Python loop,
in float64 on the CPU; differentiable, so gradients come from torch autograd.  torch is present in this image, so this IS
the reference's arithmetic (parity pinned by executing the same library calls)."""
from __future__ import annotations

import numpy as np
import torch


def detection_loss(cls: torch.Tensor, bb: torch.Tensor, y: torch.Tensor, bg_index: int, class_weights=None,
                   cls_loss_weight: float = 1.0, bb_loss_weight: float = 1.0):
    weights = None if class_weights is None else torch.as_tensor(class_weights, dtype=cls.dtype)
    cross_entropy = torch.nn.CrossEntropyLoss(weight=weights)
    huber = torch.nn.HuberLoss()
    label_true = y[:, 0].long()
    bb_true = y[:, 1:]
    loss_cls = cross_entropy(cls, label_true)
    loss_bb = 0
    num_bb = 0
    for i, label in enumerate(label_true):
        if label != bg_index:
            num_bb += 1
            loss_bb = loss_bb + huber(bb_true[i, :], bb[i, :])
    loss_bb = loss_bb / num_bb if num_bb != 0 else 0
    try:
        if np.isnan(loss_bb.item()):
            loss_bb = 0
    except Exception:
        pass
    return cls_loss_weight * loss_cls + bb_loss_weight * loss_bb, loss_cls, loss_bb
