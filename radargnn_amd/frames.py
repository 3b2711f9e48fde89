"""Batched on-device hot path: B radar frames resident in HBM -> graphs -> DetNetBasic forward.

This is the fused counterpart of what the reference does in two separate programs -- graph construction at
dataset-creation time (preprocessor/radarscenes/dataset_creation.py:190-229,786-814) and the forward pass at
inference time (postprocessor/inference.py:48-68) -- with PyG's ``Batch`` numbering (utils/data_handling.py:30):
all frames of a batch are laid back to back, node indices are global, neighbours are only searched inside a
frame.  Nothing leaves the device between the search and the heads.

Frames are independent, so multi-GPU scaling is a static block partition of the frame list over the ranks with
no collective (``shard_range``).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import ops
from .gnn.mpnn_layers import TargetCSR
from .synthetic import RadarFrame, concat_frames


@dataclass
class GraphSettings:
    """The GRAPH_CONSTRUCTION block of the reference's YAML (configurations/configuration_radarscenes.yml:17-23)."""
    algorithm: str = "knn"                       # "knn" | "radius"
    k: int = 20
    r: float = 1.0
    node_features: Sequence[str] = ("rcs", "velocity_vector", "time_index", "degree")
    edge_features: Sequence[str] = ("relative_position",)
    edge_mode: str = "directed"
    distance_definition: str = "X"               # "X" | "XV"


@dataclass
class FrameBatch:
    """Point clouds of B frames in HBM (float64 like the reference's numpy arrays)."""
    X: torch.Tensor            # [N,2]
    V: torch.Tensor            # [N,2]
    rcs: torch.Tensor          # [N]
    timestamp: torch.Tensor    # [N]
    frame_ptr: torch.Tensor    # [B+1] int64 (device)
    frame_sizes: np.ndarray    # host copy, for argument checks only

    @property
    def num_frames(self) -> int:
        return len(self.frame_sizes)

    @property
    def num_points(self) -> int:
        return self.X.shape[0]

    @staticmethod
    def from_frames(frames: Sequence[RadarFrame], device="cuda") -> "FrameBatch":
        cat, ptr = concat_frames(list(frames))
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(device)
        return FrameBatch(up(cat.X), up(cat.V), up(cat.rcs.reshape(-1)), up(cat.timestamp.reshape(-1)),
                          torch.from_numpy(ptr).to(device), np.diff(ptr))


@dataclass
class GraphBatch:
    """What ``create_graph_data`` + PyG collation hand to the model, plus the CSR the search produced."""
    x: torch.Tensor                 # f32 [N, Dn]
    edge_index: torch.Tensor        # int64 [2, E]   row 0 = query i, row 1 = neighbour j (aggregation target)
    edge_attr: torch.Tensor         # f32 [E, De]
    degree: Optional[torch.Tensor]  # int32 [N]
    status: torch.Tensor            # int32 [1] device-side error flags
    num_frames: int
    cell_order: Optional[torch.Tensor] = None   # int32 [N] rows in grid-cell order (scheduling hint for the convs)

    def check(self) -> None:
        """Synchronises; raises what the reference would have raised on this input."""
        st = int(self.status.item())
        if st & ops.STATUS_KNN_TOO_FEW_POINTS:
            raise ValueError("Expected n_neighbors < n_samples_fit in at least one frame")
        if st & ops.STATUS_DOT_PRODUCT:
            raise Exception("Error in dot product calculation")
        if st & ops.STATUS_TIME_INDEX_OVERFLOW:
            raise RuntimeError("more than 3072 distinct timestamps in one frame")


def build_graphs(batch: FrameBatch, cfg: GraphSettings) -> GraphBatch:
    """Graph construction + feature extraction for every frame of the batch, entirely on the device."""
    dev = batch.X.device
    n = batch.num_points
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    basis = batch.X if cfg.distance_definition == "X" else torch.cat((batch.X, batch.V), dim=1)
    grids: list = []
    if cfg.algorithm == "knn":
        if n and int(batch.frame_sizes.min()) <= cfg.k:
            raise ValueError(f"Expected n_neighbors < n_samples_fit, but n_neighbors = {cfg.k}, "
                             f"n_samples_fit = {int(batch.frame_sizes.min())}")
        nbr, ei, _ = ops.knn_graph(basis, batch.frame_ptr, cfg.k, status=status, grid_out=grids)
        rowptr, col = None, nbr.reshape(-1)
    elif cfg.algorithm == "radius":
        rowptr, col, ei = ops.radius_graph(basis, batch.frame_ptr, cfg.r, grid_out=grids)
    else:
        raise Exception("Invalid graph construction algorithm selected")
    degree = tidx = None
    if "degree" in cfg.node_features:
        if rowptr is None:
            rowptr = torch.arange(0, n * cfg.k + 1, cfg.k, dtype=torch.int32, device=dev)
        degree = ops.undirected_degree(rowptr, col, n)
    if "time_index" in cfg.node_features:
        tidx, _ = ops.time_index(batch.timestamp, batch.frame_ptr, status=status)
    edge_attr, _ = ops.edge_features(batch.X, batch.V, ei, list(cfg.edge_features), cfg.edge_mode, dtype=torch.float32,
                                     status=status)
    x = ops.node_features(batch.X, batch.V, batch.rcs, tidx, degree, list(cfg.node_features), dtype=torch.float32)
    order = grids[0].cell_order() if grids and n else None
    return GraphBatch(x, ei, edge_attr, degree, status, batch.num_frames, order)


class HotPath:
    """graph-build + GNN forward for a batch of frames: the unit BASELINE.json's frames/s is counted in."""

    def __init__(self, model, graph_settings: GraphSettings, with_softmax: bool = False):
        self.model = model
        self.cfg = graph_settings
        self.with_softmax = with_softmax

    def __call__(self, batch: FrameBatch) -> Tuple[torch.Tensor, torch.Tensor, GraphBatch]:
        g = build_graphs(batch, self.cfg)
        graph = TargetCSR(g.edge_index, g.x.shape[0], order=g.cell_order)
        cls, bb = self.model.forward_graph(g.x, graph, graph.sort_edge_attr(g.edge_attr))
        if self.with_softmax:                                   # postprocessor/inference.py:62
            cls = ops.softmax_rows(cls)
        return cls, bb, g


def shard_range(num_items: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Static block partition of a list of independent frames (or batches) over the ranks; no collective."""
    base, rem = divmod(num_items, world_size)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)
