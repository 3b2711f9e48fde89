"""Single-pair front end of the point-pair feature kernel -- mirror of
``graph_constructor/features.py:6-122`` (``get_En_equivariant_point_pair_metrics``)."""
from __future__ import annotations

from typing import Tuple

import numpy as np
import torch

from .. import ops


def get_En_equivariant_point_pair_metrics(p1: np.ndarray, p2: np.ndarray, v1: np.ndarray, v2: np.ndarray,
                                          mode: str) -> Tuple[float, float, float, float]:
    """Distance, angle between the velocity vectors and the two angles between each velocity vector and the
    connecting line, all angles in degrees; zero vectors give 90 degrees.  Inputs are [2,1] (or [2]) arrays."""
    if not torch.cuda.is_available():
        raise RuntimeError("radargnn_amd: no MI355X visible (no CPU fallback)")
    as_row = lambda a: np.asarray(a, dtype=np.float64).reshape(1, -1)[:, :2]
    X = torch.from_numpy(np.concatenate((as_row(p1), as_row(p2)))).cuda()
    V = torch.from_numpy(np.concatenate((as_row(v1), as_row(v2)))).cuda()
    ei = torch.tensor([[0], [1]], dtype=torch.int64, device="cuda")
    out, status = ops.edge_features(X, V, ei, ["point_pair_features"], mode, dtype=torch.float64)
    if status.item() & ops.STATUS_DOT_PRODUCT:
        raise Exception("Error in dot product calculation")
    d, th_v, a, b = out[0].tolist()
    return d, th_v, a, b
