"""CPU oracle for batching graphs.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module; the
product (``radargnn_amd``) never does.

numpy restatement of what ``torch_geometric.loader.DataLoader`` does with the reference's graphs
(src/gnnradarobjectdetection/utils/data_handling.py:30 -> ``Batch.from_data_list`` -> ``collate``).  torch_geometric
2.1.0.post1 (Dockerfile:21-26) is not installable in this image, so the rules are restated from its published
behaviour -- PARITY UNPINNED by a PyG-executed vector; pinned by construction and by the hand-computed example in
tests/test_data_host.py:

* every tensor attribute is concatenated along dim 0 (``Data.__cat_dim__`` = 0) ...
* ... except attributes whose name contains "index" (here: ``edge_index``), concatenated along the LAST dim with the
  cumulative number of nodes of the preceding graphs added (``Data.__inc__`` = ``num_nodes``);
* ``num_nodes`` of a graph = ``x.shape[0]``;
* ``batch`` [N] int64 = position of the node's graph in the list, ``ptr`` [B + 1] int64 = cumulative node counts.
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import numpy as np


def collate(graphs: Sequence[Dict[str, np.ndarray]]) -> Dict[str, np.ndarray]:
    """``graphs``: dicts of numpy arrays with at least ``x``; -> dict of the batch's arrays + ``batch`` + ``ptr``."""
    out: Dict[str, np.ndarray] = {}
    sizes = [g["x"].shape[0] for g in graphs]
    ptr = np.concatenate(([0], np.cumsum(sizes))).astype(np.int64)
    keys: List[str] = list(graphs[0].keys()) if graphs else []
    for k in keys:
        if "index" in k:
            out[k] = np.concatenate([g[k] + ptr[i] for i, g in enumerate(graphs)], axis=-1)
        else:
            out[k] = np.concatenate([g[k] for g in graphs], axis=0)
    out["batch"] = np.repeat(np.arange(len(graphs), dtype=np.int64), sizes)
    out["ptr"] = ptr
    return out
