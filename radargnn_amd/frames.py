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

import os

from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import ops
from .gnn.linear import frame_scope
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
    rowptr: Optional[torch.Tensor] = None       # int32 [N+1] radius graphs: the search's rows (edges grouped by edge_index[0])
    cell_rank: Optional[torch.Tensor] = None    # int32 [N] position of node i in cell_order (its inverse permutation)
    split: Optional[tuple] = None               # radius graphs: ops.split_targets(...) of the graph, already computed
    csr: Optional["TargetCSR"] = None           # kNN graphs whose degree feature was computed from the CSR by target: that CSR
    big_edge_fraction: Optional[float] = None   # radius graphs: share of the edges into targets with > 60 in-edges (read with the edge count)
    points: Optional[tuple] = None              # (X, V) of the batch the graph was built from (HotPath: edge attributes in target order)
    edge_side: Optional["torch.cuda.Stream"] = None   # captured steps: the stream the edge side of the step is on (_stage_features)

    def check(self) -> None:
        """Synchronises; raises what the reference would have raised on this input."""
        st = int(self.status.item())
        if st & ops.STATUS_KNN_TOO_FEW_POINTS:
            raise ValueError("Expected n_neighbors < n_samples_fit in at least one frame")
        if st & ops.STATUS_DOT_PRODUCT:
            raise Exception("Error in dot product calculation")
        if st & ops.STATUS_TIME_INDEX_OVERFLOW:
            raise RuntimeError("more than 3072 distinct timestamps in one frame")
        if st & ops.STATUS_EDGE_COUNT_CHANGED:       # (first: the stale edge list then also disagrees with the new search rows)
            raise RuntimeError("the batch's points changed under a captured HIP graph (its radius graph now has a different "
                               "number of edges): build a new FrameBatch instead of modifying one in place")
        if st & ops.STATUS_NOT_SYMMETRIC:
            # (set by rgnn_csr_by_target_symmetric; the own-edge build HotPath uses for its OWN radius graphs with
            #  relative_position attributes does not search for twins and cannot set it -- those graphs are symmetric by
            #  construction, TargetCSR(own_edges=True) is not offered for caller-supplied edge lists: rgnn.h)
            raise RuntimeError("a graph passed as symmetric holds an edge without its reverse")
        if ops.splitk_timeouts(self.status.device):                 # RGNN_STATUS_SPLITK_TIMEOUT (rgnn.h): one more host read
            raise RuntimeError("a dense layer gave up waiting for a partial tile of another work-group (split-K hand-over "
                               "time-out): its output is wrong")


# replayed radius steps: the one-launch search + fill (and the CSR / plan behind it) as a branch of the captured graph beside the node
# side -- four alternating pairs on one box: 1.956 / 1.996 / 1.961 / 1.956 -> 1.949 / 1.948 / 1.947 / 1.948 ms per C2 step
FORK_DIRECT = os.environ.get("RGNN_NO_FORK_DIRECT") is None
DIRECT_ROWS = os.environ.get("RGNN_NO_DIRECT_ROWS") is None     # replayed radius steps: search + fill in one launch at the committed rows


def _stage_search(batch: FrameBatch, cfg: GraphSettings, status: torch.Tensor, static: Optional[dict] = None, grid_only: bool = False):
    """Everything up to the point where the radius graph's edge count is needed on the host (kNN: the whole search).
    ``grid_only`` (radius graphs, replayed steps, ``static`` filled by an earlier full call): only the grid build -- the rows are the
    committed ones and ``_stage_features`` searches and fills in one launch (ops.radius_graph_rows_direct)."""
    if static is not None and "basis" in static:
        basis = static["basis"]
    else:
        basis = batch.X if cfg.distance_definition == "X" else torch.cat((batch.X, batch.V), dim=1)
        if static is not None:
            static["basis"] = basis
    biggest = int(batch.frame_sizes.max()) if len(batch.frame_sizes) else 0      # host copy: lets one launch bin every frame
    if cfg.algorithm == "knn":
        grids: list = []
        # the search's write-out also emits the shipped edge attribute list (relative_position only) and presets the degrees
        want_rel = cfg.edge_mode if (tuple(cfg.edge_features) == ("relative_position",) and cfg.distance_definition in ("X", "XV")) else None
        res = ops.knn_graph(basis, batch.frame_ptr, cfg.k, status=status, grid_out=grids, static=static,
                            max_frame_points=biggest, relative_position=want_rel,
                            degree_init="degree" in cfg.node_features and not (KNN_DEGREE_FROM_CSR and cfg.k <= 64))
        return {"grid": grids[0], "nbr": res[0], "ei": res[1], "rel": res[3] if len(res) > 3 else None,
                "deg0": res[4] if len(res) > 4 else None}
    if cfg.algorithm == "radius":
        sdict = static if static is not None else {}
        if grid_only and static is not None and "grid" in static:
            return {"grid": ops.radius_grid(basis, batch.frame_ptr, cfg.r, static, max_frame_points=biggest), "rowptr": None, "deg": None}
        grid, rowptr = ops.radius_graph_count(basis, batch.frame_ptr, cfg.r, static=sdict, max_frame_points=biggest)
        out = {"grid": grid, "rowptr": rowptr, "deg": sdict["deg"]}
        if static is None:
            # [E, edges into targets with more than 60 in-edges] side by side: ONE host read fetches both (the second number picks the
            # form of the max aggregation, TargetCSR.wants_window_kernel; a radius graph is symmetric: in-degree = row length)
            out["counts"] = ops.radius_counts(sdict["deg"], rowptr, 60)
        return out
    raise Exception("Invalid graph construction algorithm selected")


_UNIFORM_ROWPTR: dict = {}
KNN_DEGREE_FROM_CSR = os.environ.get("RGNN_NO_KNN_DEGREE_FROM_CSR") is None
FORK_EDGE_SIDE = os.environ.get("RGNN_NO_EDGE_SIDE") is None        # captured radius steps: edge side of the step as a graph branch


def _uniform_rowptr(n: int, k: int, dev) -> torch.Tensor:
    """rowptr of a kNN graph (every row holds k entries): built once per shape, read-only afterwards."""
    key = (n, k, str(dev))
    if key not in _UNIFORM_ROWPTR:
        if torch.cuda.is_current_stream_capturing():
            # (a tensor created while a stream is capturing lives in that graph's private pool and is only valid after a replay:
            #  never put it into a process-wide cache -- ADVICE r02)
            return torch.arange(0, n * k + 1, k, dtype=torch.int32, device=dev)
        if len(_UNIFORM_ROWPTR) > 16:
            _UNIFORM_ROWPTR.clear()
        _UNIFORM_ROWPTR[key] = torch.arange(0, n * k + 1, k, dtype=torch.int32, device=dev)
    return _UNIFORM_ROWPTR[key]


def _stage_features(batch: FrameBatch, cfg: GraphSettings, status: torch.Tensor, st: dict, n_edges: int,
                    guarded: bool = False, committed: Optional[torch.Tensor] = None, ordered_csr: bool = True) -> GraphBatch:
    """``guarded``: n_edges is the count of an earlier pass over this batch, not one just read back (captured step); the rows
    everything downstream of the search reads are then ``committed`` -- the rows of the last replay whose count matched
    (rgnn_radius_rows_commit) -- so that a replay on modified points computes on the previous graph (and flags it) instead of
    walking rows that no longer fit the captured buffers."""
    dev = batch.X.device
    n = batch.num_points
    rows_out = edge_attr_fused = edge_side = None
    if cfg.algorithm == "knn":
        ei, col, rowptr = st["ei"], st["nbr"].reshape(-1), None
        edge_attr_fused = st.get("rel")
    else:
        rowptr = st["rowptr"]
        rows_out = rowptr
        direct = guarded and rowptr is None                   # (grid-only search: the rows are the committed ones)
        if direct:
            rowptr = rows_out = committed[0]
            st = dict(st, deg=committed[1])
        elif guarded:
            rows_out = ops.radius_rows_commit(rowptr, n_edges, committed[0], status, st["deg"], committed[1])
            st = dict(st, deg=committed[1])           # (the degrees travel with the rows they are the lengths of)
        # the shipped edge feature list (relative_position only, float32) comes out of the fill launch itself
        fused_attr = tuple(cfg.edge_features) == ("relative_position",) and n_edges > 0
        # Captured steps (guarded): from here the step has two independent chains -- the EDGE side (fill -> CSR by target -> window
        # plan, ~75 us on the C2 batch) and the NODE side (time index / node features -> row lists -> node embedding -> the first
        # layer's isolated-row and source-term launches, which need the degrees and the row lists but no edge).  The edge side goes
        # to the side stream, i.e. into a branch of the captured graph; the model joins it where it first reads the CSR
        # (TargetCSR.join_csr).  Eager steps keep one stream: there the side stream carries the next batch's search.
        # Only where the edge side is long -- feature lists beyond relative_position (a feature launch in edge order and one in target
        # order: the 100 000-point configuration, -70 us); the headline workload measures level with and without (its small kernels
        # fill the chip either way) and keeps the one-branch graph it has always been captured as.
        # (r06: ... and replayed steps whose search and fill are ONE launch at the committed rows -- that launch is a 76-us chain of
        #  dependent loads which leaves most of the chip's wave slots free, and everything on the node side reads the COMMITTED degrees)
        if guarded and FORK_EDGE_SIDE and (not fused_attr or (direct and FORK_DIRECT)) and n_edges > 0 and torch.cuda.is_current_stream_capturing():
            edge_side = ops.ctx().side(dev)
            edge_side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(edge_side):                    # (None: stays on the current stream)
            if direct:
                res = ops.radius_graph_rows_direct(st["grid"], rowptr, cfg.r, n_edges, status,
                                                   relative_position=cfg.edge_mode if fused_attr else None)
            else:
                res = ops.radius_graph_fill(st["grid"], rowptr, cfg.r, n_edges, guard_status=status if guarded else None,
                                            relative_position=cfg.edge_mode if fused_attr else None)
        col, ei = res[0], res[1]
        if fused_attr:
            edge_attr_fused = res[2]
    degree = tidx = csr = None
    if "degree" in cfg.node_features:
        if cfg.algorithm == "knn" and n > 0 and cfg.k <= 64 and KNN_DEGREE_FROM_CSR:
            # the conv layers need the edges sorted by target anyway: build that CSR now and count, per target, the in-edges whose
            # source the target lists itself (ops.knn_degree_from_csr: every target's own row once, no atomics) instead of chasing a
            # different row of the neighbour table for every out-edge (rgnn_undirected_degree_preset: 169 -> see MEASUREMENTS.md 4.3)
            grid = st["grid"]
            csr = TargetCSR(ei, n, order=grid.cell_order(), rank=grid.cell_rank(), all_sources=True, status=status,
                            knn_frames=(batch.frame_ptr, cfg.k, int(batch.frame_sizes.max())), ordered=ordered_csr)
            degree = ops.knn_degree_from_csr(csr.rowptr, csr.src, csr.order, st["nbr"])
        elif cfg.algorithm == "radius":
            # d(i,j) <= r is symmetric, so the directed edge set is symmetric and the undirected degree networkx
            # reports (graph.py:93-96) is simply the out-degree the count pass already produced
            degree = st["deg"]
        else:
            rowptr = _uniform_rowptr(n, cfg.k, dev)
            if st.get("deg0") is not None:
                degree = ops.undirected_degree_preset(rowptr, col, st["deg0"])     # (the search preset the out-degrees)
            else:
                degree = ops.undirected_degree(rowptr, col, n)
    # (the time index goes straight into its feature column: one launch, one block per frame -- unless a frame is so large that
    #  its block would write all those rows alone: one 100 000-point cloud then keeps the two launches)
    fused_tidx = "time_index" in cfg.node_features and n > 0 and int(batch.frame_sizes.max()) <= 16384
    if "time_index" in cfg.node_features and not fused_tidx:
        tidx, _ = ops.time_index(batch.timestamp, batch.frame_ptr, status=status, max_frame_points=int(batch.frame_sizes.max()))
    if edge_attr_fused is not None:
        edge_attr = edge_attr_fused
    else:
        with torch.cuda.stream(edge_side):
            edge_attr, _ = ops.edge_features(batch.X, batch.V, ei, list(cfg.edge_features), cfg.edge_mode, dtype=torch.float32,
                                             status=status)
    split = None
    if fused_tidx:
        # radius graphs are symmetric: the nodes with / without incoming edges follow from the degrees the search counted; the
        # node-feature kernel counts them per frame on the way and one more launch compacts the two lists (TargetCSR would
        # otherwise spend four launches on them)
        per_frame = (torch.empty(batch.num_frames, dtype=torch.int32, device=dev)
                     if (cfg.algorithm == "radius" and degree is not None and ops.SORTED_ROW_LISTS
                         and os.environ.get("RGNN_NO_FUSED_SPLIT") is None) else None)
        x = ops.node_features_time_index(batch.X, batch.V, batch.rcs, batch.timestamp, batch.frame_ptr, degree,
                                         list(cfg.node_features), dtype=torch.float32, status=status, frame_nonempty=per_frame)
        if per_frame is not None:
            split = ops.split_by_degree_frames(degree, batch.frame_ptr, per_frame)
    else:
        x = ops.node_features(batch.X, batch.V, batch.rcs, tidx, degree, list(cfg.node_features), dtype=torch.float32)
    order = st["grid"].cell_order() if n else None         # (views of the grid workspace: no copy, no inversion launch)
    return GraphBatch(x, ei, edge_attr, degree, status, batch.num_frames, order, rows_out,
                      st["grid"].cell_rank() if n else None, split, csr, points=(batch.X, batch.V), edge_side=edge_side)


def _check_knn_sizes(batch: FrameBatch, cfg: GraphSettings) -> None:
    if cfg.algorithm == "knn" and batch.num_points and int(batch.frame_sizes.min()) <= cfg.k:
        raise ValueError(f"Expected n_neighbors < n_samples_fit, but n_neighbors = {cfg.k}, "
                         f"n_samples_fit = {int(batch.frame_sizes.min())}")


def build_graphs(batch: FrameBatch, cfg: GraphSettings, ordered_csr: bool = True) -> GraphBatch:
    """Graph construction + feature extraction for every frame of the batch, entirely on the device.  The radius
    graph reads its edge count back once (count -> scan -> fill); kNN needs no host round trip."""
    _check_knn_sizes(batch, cfg)
    status = torch.zeros(1, dtype=torch.int32, device=batch.X.device)
    st = _stage_search(batch, cfg, status)
    big = None
    if cfg.algorithm == "knn":
        n_edges = batch.num_points * cfg.k
    else:
        n_edges, e_big = st["counts"].tolist()
        big = e_big / max(n_edges, 1)
    g = _stage_features(batch, cfg, status, st, n_edges, ordered_csr=ordered_csr)
    g.big_edge_fraction = big
    return g


class HotPath:
    """graph-build + GNN forward for a batch of frames: the unit BASELINE.json's frames/s is counted in.

    ``use_hip_graphs``: ALL launches of a step -- grid build, search, features, CSR build, the DetNetBasic forward -- are
    captured into ONE HIP graph and replayed: a step is a single graph launch (a one-frame step is ~70 launches of a few
    microseconds each; C1: 0.56 ms eager, 0.37 ms replayed).  A batch runs eagerly the first time it is seen and is captured
    on its next occurrence; one captured graph is kept at a time.  Results are identical either way (same kernels, same
    order).  kNN graphs have E = N k; for radius graphs the capture is sized for the edge count the eager pass found, and
    the fill pass verifies it on the device (rgnn_radius_graph_fill_checked): if the batch's points were modified in place
    so that the count changed, the replay leaves the previous outputs and sets STATUS_EDGE_COUNT_CHANGED
    (``GraphBatch.check()`` raises).

    (ROCm 7.0 runtime: kernels enqueued EAGERLY on a stream between graph launches, and two different graphs launched
    alternately with a host read in between, were not reliably ordered against each other -- memory faults after a few
    steps.  One graph per step, replayed back to back, is; hence no eager search stage and no second graph.)"""

    def __init__(self, model, graph_settings: GraphSettings, with_softmax: bool = False, use_hip_graphs: bool = False,
                 bn_scope: str = "batch"):
        if bn_scope not in ("batch", "frame"):
            raise ValueError("bn_scope must be 'batch' or 'frame'")
        # bn_scope = "frame": every train-mode BatchNorm takes its statistics per frame -- the numbers the reference's
        # one-frame-per-forward inference loop computes (evaluate.py:40; the model is never put in eval mode), at batched
        # throughput.  "batch": statistics over the whole batch, what ONE forward over a 64-frame Batch computes.
        self.bn_scope = bn_scope
        self.model = model
        self.cfg = graph_settings
        self.with_softmax = with_softmax
        self.use_hip_graphs = use_hip_graphs
        # radius graphs hold (s, t) and (t, s) alike (|a - b|^2 is evaluated symmetrically); kNN graphs do not
        self.symmetric_graph = graph_settings.algorithm == "radius"
        # a maximum does not depend on the order of a target's in-edges: the CSR build of kNN batches skips its ranking pass
        self._ordered_csr = not (getattr(model, "aggregation", None) == "max" and os.environ.get("RGNN_ORDERED_CSR") is None)
        self._seen = None          # id of the batch seen last (first sight runs eagerly)
        self._seen_edges = 0       # ... and the edge count that pass found
        self._key = None           # signature of the captured graph
        self._graph = None
        self._static = None        # static buffers of the search stage + outputs of the captured graph

    # ---- the two halves of a step -------------------------------------------------------------------
    def _model(self, g: GraphBatch):
        graph, ea_sorted = self._prepare(g)
        return self._forward(g, graph, ea_sorted)

    def _prepare(self, g: GraphBatch, plan_here: bool = False):
        """The graph-side half of the model stage: edges by target, their attributes in that order, the window plan (started on the
        side stream) -- everything that depends on the graph only, not on the weights."""
        # Radius graphs out of this library's search are symmetric, and every edge feature is a function of the edge's two end points:
        # the attributes in target order need no search for each in-edge's twin (TargetCSR(own_edges=True) leaves the OWN out-edge at
        # every slot).  relative_position in directed mode is antisymmetric under reversal, attr(i -> t) = -attr(t -> i): the first
        # kernel that reads them negates its weights.  Any other feature list: ops.edge_features_reversed computes the reversed
        # edges' features straight into target order -- the same arithmetic on the same end points as the twin's own row, bit for
        # bit (r05; the search was a binary search per edge: 179 us and 1.1 GB of reads on the 100 000-point cloud).
        rel_only = tuple(self.cfg.edge_features) == ("relative_position",) and self.cfg.edge_mode == "directed"
        twin_free = (self.symmetric_graph and g.rowptr is not None and g.points is not None and len(self.cfg.edge_features) > 0
                     and os.environ.get("RGNN_NO_REVERSED_FEATURES") is None)
        edge_side = g.edge_side if g.csr is None else None
        with torch.cuda.stream(edge_side):                      # (None: the current stream; else the edge side of a captured step)
            graph = g.csr if g.csr is not None else TargetCSR(g.edge_index, g.x.shape[0], order=g.cell_order, rank=g.cell_rank, symmetric=self.symmetric_graph,
                              all_sources=self.cfg.algorithm == "knn", source_rows=g.rowptr, status=g.status, split=g.split,
                              knn_frames=((self._frame_ptr, self.cfg.k, self._biggest_frame) if self.cfg.algorithm == "knn" else None),
                              own_edges=rel_only or twin_free, big_edge_fraction=g.big_edge_fraction, ordered=self._ordered_csr)
            if graph.own_edge is not None and not rel_only:
                ea_sorted = ops.edge_features_reversed(g.points[0], g.points[1], g.edge_index, graph.own_edge, list(self.cfg.edge_features),
                                                       self.cfg.edge_mode, dtype=g.edge_attr.dtype, status=g.status)[0]
            else:
                ea_sorted = graph.sort_edge_attr(g.edge_attr, lazy=True)
            if edge_side is not None:
                graph.mark_csr_on(edge_side)                    # (the main stream joins at its first read of the CSR: join_csr)
            # the window plan behind the CSR -- on the side stream beside the embedding launches (no-op unless the rule applies).
            # Started HERE and nowhere else, and joined whatever happens: a caller that only builds graphs never forks, and an
            # exception inside the model cannot leave the side stream writing a plan buffer the allocator has already handed on
            if plan_here:
                graph.build_win_plan_here()
            else:
                graph.start_win_plan()
        return graph, ea_sorted

    def _forward(self, g: GraphBatch, graph, ea_sorted):
        try:
            if self.bn_scope == "frame":
                with frame_scope(self._frame_ptr, g.x.shape[0], graph):
                    cls, bb = self.model.forward_graph(g.x, graph, ea_sorted)
            else:
                cls, bb = self.model.forward_graph(g.x, graph, ea_sorted)
        finally:
            graph.join_csr()
            graph.join_win_plan()                               # (a plan nobody consumed must not leave the side stream forked)
        if self.with_softmax:                                   # postprocessor/inference.py:62
            cls = ops.softmax_rows(cls)
        return cls, bb

    def _eager(self, batch: FrameBatch):
        self._frame_ptr = batch.frame_ptr
        self._biggest_frame = int(batch.frame_sizes.max()) if len(batch.frame_sizes) else 0
        g = build_graphs(batch, self.cfg, ordered_csr=self._ordered_csr)
        cls, bb = self._model(g)
        return cls, bb, g

    # ---- a step in two halves, for callers that stream batches (FrameStreamer) --------------------------------------------
    def begin(self, batch: FrameBatch, after: Optional["torch.cuda.Event"] = None):
        """The search half of an eager step on a SIDE stream, behind ``after`` (the event of the batch's upload) and nothing else:
        grid build, neighbour count and scan (kNN: the whole search), plus -- radius graphs -- an asynchronous copy of the edge count
        into pinned host memory.  ``finish`` runs the rest on the calling stream.  Issued one batch ahead, the count is on the host
        long before ``finish`` needs it: the launching thread never waits for the device while the model stage of the batch before
        is still running (r04: the host read stood between every batch's search and model stages, and the device idled while Python
        enqueued the ~45 launches of the model stage: 0.81 of the resident rate)."""
        _check_knn_sizes(batch, self.cfg)
        dev = batch.X.device
        side = ops.ctx().side(dev, "search")                     # (a hardware queue of its own: ops.independent_stream)
        with torch.cuda.stream(side):
            if after is not None:
                side.wait_event(after)
            status = torch.zeros(1, dtype=torch.int32, device=dev)
            st = _stage_search(batch, self.cfg, status)
            count = None
            if self.cfg.algorithm == "radius":
                count = torch.empty(2, dtype=torch.int32).pin_memory() if getattr(self, "_count_pool", None) is None or not self._count_pool \
                    else self._count_pool.pop()
                count.copy_(st["counts"], non_blocking=True)
            done = torch.cuda.Event()
            done.record(side)
        return {"batch": batch, "status": status, "st": st, "count": count, "done": done, "side": side}

    def finish(self, h) -> Tuple[torch.Tensor, torch.Tensor, GraphBatch]:
        batch, st = h["batch"], h["st"]
        main = torch.cuda.current_stream(batch.X.device)
        if h["count"] is not None:
            h["done"].synchronize()                              # (complete long ago when begin ran a batch ahead)
            n_edges, e_big = int(h["count"][0]), int(h["count"][1])
            big = e_big / max(n_edges, 1)
            if getattr(self, "_count_pool", None) is None:
                self._count_pool = []
            self._count_pool.append(h["count"])
        else:
            n_edges, big = batch.num_points * self.cfg.k, None
        main.wait_event(h["done"])
        # (allocated on the side stream, consumed on this one: the allocator must not hand the blocks out again before this
        #  stream's launches are through)
        #  (... including what the grid object alone holds: with distance_definition "XV" the basis is a torch.cat made on the side
        #   stream and read through the raw grid descriptor by the fill launch of THIS stream -- ADVICE r05)
        for t in [h["status"]] + [v for v in st.values() if torch.is_tensor(v)] + [st["grid"].ws, st["grid"].X, st["grid"].frame_ptr]:
            t.record_stream(main)
        self._frame_ptr = batch.frame_ptr
        self._biggest_frame = int(batch.frame_sizes.max()) if len(batch.frame_sizes) else 0
        g = _stage_features(batch, self.cfg, h["status"], st, n_edges, ordered_csr=self._ordered_csr)
        g.big_edge_fraction = big
        cls, bb = self._model(g)
        return cls, bb, g

    def __call__(self, batch: FrameBatch) -> Tuple[torch.Tensor, torch.Tensor, GraphBatch]:
        if not self.use_hip_graphs:
            return self._eager(batch)
        _check_knn_sizes(batch, self.cfg)
        if self._seen != id(batch):                             # first sight of this batch: plain eager pass
            self._key = self._graph = self._static = None
            self._batch_ref = batch
            out = self._eager(batch)
            self._seen, self._seen_edges, self._seen_big = id(batch), int(out[2].edge_index.shape[1]), out[2].big_edge_fraction
            return out
        n_edges = self._seen_edges
        # the capture bakes in the folded weights / bf16 planes that the eager pass cached (their fold / split kernels are
        # not part of the graph), so the key carries everything those caches are keyed on: an optimizer step or
        # load_state_dict (in-place: version counters), a replaced or moved parameter (storage), an invalidated cache
        key = (id(batch), n_edges, self.model.training, self.bn_scope, ops.CACHE_EPOCH,
               tuple((p.data_ptr(), p._version) for p in self.model.parameters()))
        if self._key != key:
            self._graph = None
            if self._static is None:
                # the static buffers of the search stage come from one eager call OUTSIDE the capture: they outlive
                # re-captures (a capture's own allocations belong to that graph's pool)
                self._static = {"status": torch.zeros(1, dtype=torch.int32, device=batch.X.device), "search": {}}
                st0 = _stage_search(batch, self.cfg, self._static["status"], static=self._static["search"])
                if self.cfg.algorithm == "radius":
                    # the rows downstream kernels read (see _stage_features): start from the rows of the points as they are
                    # NOW; if they no longer give the edge count the eager pass found, the batch was modified in between --
                    # treat it as a new batch (this is the one place where the host may still look)
                    if int(st0["rowptr"][-1].item()) != n_edges:
                        self._seen = None
                        return self.__call__(batch)
                    self._static["rows"] = (st0["rowptr"].clone(), st0["deg"].clone())
            status, sstat = self._static["status"], self._static["search"]
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            self._frame_ptr = batch.frame_ptr
            self._biggest_frame = int(batch.frame_sizes.max()) if len(batch.frame_sizes) else 0
            with torch.cuda.graph(graph):
                status.zero_()
                st = _stage_search(batch, self.cfg, status, static=sstat,
                                   grid_only=DIRECT_ROWS and self.cfg.algorithm == "radius" and n_edges > 0)
                g = _stage_features(batch, self.cfg, status, st, n_edges, guarded=True, committed=self._static.get("rows"),
                                    ordered_csr=self._ordered_csr)
                g.big_edge_fraction = self._seen_big              # (what the eager first pass over this batch read)
                cls, bb = self._model(g)
            self._graph, self._key, self._static["outs"] = graph, key, (cls, bb, g)
        self._graph.replay()
        return self._static["outs"]


def shard_range(num_items: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Static block partition of a list of independent frames (or batches) over the ranks; no collective."""
    base, rem = divmod(num_items, world_size)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


class FrameStreamer:
    """Host-streaming inference: host frames in, host logits / boxes out, at the resident rate.

    The reference moves every frame to the device and every result back, one after the other, on one stream
    (postprocessor/inference.py:48-68: ``data.to(device)``, forward, ``.cpu()``; batches come from the DataLoader of
    utils/data_handling.py:7-36).  Here the three stages run as a pipeline over batches:

    * a loader thread lays the frames of batch i + 1 back to back in PINNED staging buffers (a ring of ``slots``) and issues their
      H2D copies on a copy stream while the compute stream still works on batch i;
    * the calling thread waits for the copy's event ON THE COMPUTE STREAM (no host block), runs ``hot`` (eager launches: every
      batch is a different graph) on device buffers it sees as a ``FrameBatch``;
    * the results of batch i go to pinned host buffers on a third stream behind the compute stream's event, and are handed to the
      caller while batch i + 1 computes.

    ``run`` yields ``(cls, boxes)`` host tensors per batch, in order, ``behind`` batches after they were launched: views of pinned
    ring buffers, valid until ``slots - behind - 1`` further batches have been yielded (clone what must live longer)."""

    def __init__(self, hot: "HotPath", slots: int = 6, lookahead: bool = True, behind: int = 2):
        if slots < 2:
            raise ValueError("at least two staging slots")
        self.hot = hot
        self.slots = slots
        self.behind = behind           # results are handed out this many batches late (clamped to slots - 1; see run)
        # lookahead: the search half of the next batch is launched a batch ahead (HotPath.begin / finish); needs >= 3 slots (a batch
        # whose search is under way occupies one beside the batch in the model stage and the one being staged)
        self.lookahead = lookahead
        self.device = torch.device("cuda", torch.cuda.current_device())
        # (streams on hardware queues of their own -- ops.independent_stream: an upload or a download that shares the queue of the
        #  compute or the search stream serialises with it, +0.2 ms per batch)
        self.copy_stream = ops.ctx().side(self.device, "upload")
        self.down_stream = ops.ctx().side(self.device, "download")
        self._in = [None] * slots          # per slot: dict(cap, host tensors, device tensors, free event)
        self._out = [None] * slots

    # ---- staging ------------------------------------------------------------------------------------
    def _slot(self, i: int, n: int, b: int):
        need = 48 * n + 8 * (b + 1)
        s = self._in[i]
        if s is None or s["bytes"] < need:
            # ONE pinned block and ONE device block per slot: X | V | rcs | timestamp | frame_ptr back to back (rgnn_stage_frames),
            # one H2D copy per batch
            size = max(need, 4096) * 5 // 4
            host = torch.empty(size, dtype=torch.uint8).pin_memory()
            s = self._in[i] = {"bytes": size, "host": host, "dev": torch.empty(size, dtype=torch.uint8, device=self.device)}
        return s

    @staticmethod
    def _addresses(f: RadarFrame, keep: list):
        """(address of X, V, rcs, timestamp; point count) of a frame's arrays.  Cached on the frame, keyed on the identity of the
        four arrays, when they are used as they are; an array that had to be converted (not float64, not contiguous) is converted
        on every call and its copy parked in ``keep`` until the caller is through with the addresses."""
        c = getattr(f, "_rgnn_addr", None)
        if c is not None and c[1] is f.X and c[2] is f.V and c[3] is f.rcs and c[4] is f.timestamp:
            return c[0]
        arrs = (f.X, f.V, f.rcs, f.timestamp)
        conv = [np.ascontiguousarray(a, dtype=np.float64) for a in arrs]
        n = conv[0].shape[0]
        if conv[0].shape != (n, 2) or conv[1].shape != (n, 2) or conv[2].size != n or conv[3].size != n:
            raise ValueError("a frame needs X [n, 2], V [n, 2], rcs [n], timestamp [n]")
        row = (conv[0].ctypes.data, conv[1].ctypes.data, conv[2].ctypes.data, conv[3].ctypes.data, n)
        if all(x is y for x, y in zip(conv, arrs)):
            try:
                f._rgnn_addr = (row,) + arrs
            except AttributeError:                                # (a frame type with __slots__: no cache)
                pass
        else:
            keep.extend(conv)
        return row

    @staticmethod
    def views(block: torch.Tensor, n: int, b: int):
        """X, V, rcs, timestamp, frame_ptr as views of a staged block (layout of rgnn_stage_frames)."""
        f64 = lambda lo, hi: block[lo:hi].view(torch.float64)
        return (f64(0, 16 * n).view(n, 2), f64(16 * n, 32 * n).view(n, 2), f64(32 * n, 40 * n), f64(40 * n, 48 * n),
                block[48 * n:48 * n + 8 * (b + 1)].view(torch.int64))

    def _stage(self, i: int, frames: Sequence[RadarFrame]):
        """Loader thread: frames -> pinned slot i (free: its previous batch's kernels are through) -> device (copy stream).
        Returns what the compute side needs.  The copies into the pinned block are ONE call into librgnn, made without the
        interpreter lock (numpy copies, one per array and frame, handed the lock back and forth with the launching thread 256 times
        per batch: 0.2 ms of ITS time per batch, tools/stream_probe.py)."""
        keep: list = []
        table = np.array([self._addresses(f, keep) for f in frames], dtype=np.int64).reshape(-1, 5)
        sizes = np.ascontiguousarray(table[:, 4])
        addr = np.ascontiguousarray(table[:, :4])
        n, b = int(sizes.sum()), len(frames)
        s = self._slot(i, n, b)
        ops.check(ops.lib.rgnn_stage_frames(b, addr.ctypes.data, sizes.ctypes.data, s["host"].data_ptr(), s["bytes"]))
        del keep
        used = 48 * n + 8 * (b + 1)
        with torch.cuda.stream(self.copy_stream):
            s["dev"][:used].copy_(s["host"][:used], non_blocking=True)
            ready = torch.cuda.Event()
            ready.record(self.copy_stream)
        return i, n, b, sizes, ready

    def run(self, host_batches):
        import queue
        import threading
        q: "queue.Queue" = queue.Queue()
        free: "queue.Queue" = queue.Queue()                       # (slot, event after which its device buffers may be overwritten)
        for i in range(self.slots):
            free.put((i, None))
        dev_index = self.device.index

        def loader():
            torch.cuda.set_device(dev_index)
            try:
                for frames in host_batches:
                    i, consumed = free.get()                      # blocks while all slots hold batches not yet computed
                    if consumed is not None:
                        consumed.synchronize()
                    q.put(self._stage(i, list(frames)))
                q.put(None)
            except BaseException as exc:                           # noqa: BLE001 -- handed to the consumer
                q.put(exc)

        th = threading.Thread(target=loader, daemon=True)
        th.start()
        compute = torch.cuda.current_stream(self.device)
        # results handed out `behind` batches late (slots of the output ring with their download events): with one batch the launching
        # thread waits, every batch, for the download of the batch the device has only just finished -- it can never be more than one
        # batch ahead, and whatever delays it (the loader thread holding the interpreter, a slow host) idles the device; two batches
        # behind it runs up to two ahead.  A slot of the output ring is reused after `slots` batches: behind <= slots - 1.
        behind = max(1, min(self.behind, self.slots - 1))
        pending: list = []
        j = 0
        lookahead = self.lookahead and not self.hot.use_hip_graphs and self.slots >= 3
        ahead = None                                              # (slot, handle) of the batch whose search half is already enqueued
        finished = False
        while True:
            if lookahead:
                # the search half of batch i + 1 goes out (side stream, behind its upload only) BEFORE the model half of batch i is
                # enqueued: its edge count is on the host by the time batch i's ~45 launches are in the queue
                nxt = None
                if not finished:
                    item = q.get()
                    if item is None:
                        finished = True
                    elif isinstance(item, BaseException):
                        raise item
                    else:
                        i2, n2, b2, sizes2, ready2 = item
                        nb = FrameBatch(*self.views(self._in[i2]["dev"], n2, b2), sizes2)
                        nxt = (i2, self.hot.begin(nb, after=ready2))
                if ahead is None:
                    if nxt is None:
                        break
                    ahead = nxt
                    continue
                i, handle = ahead
                ahead = nxt
                cls, bb, g = self.hot.finish(handle)
            else:
                item = q.get()
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise item
                i, n, b, sizes, ready = item
                compute.wait_event(ready)
                batch = FrameBatch(*self.views(self._in[i]["dev"], n, b), sizes)
                cls, bb, g = self.hot(batch)
            done = torch.cuda.Event()
            done.record(compute)
            free.put((i, done))                                   # the loader may refill slot i once this batch's kernels are through
            o = self._out[j % self.slots]
            if o is None or o[0].shape != cls.shape or o[1].shape != bb.shape:
                o = self._out[j % self.slots] = (torch.empty(cls.shape, dtype=cls.dtype).pin_memory(),
                                                 torch.empty(bb.shape, dtype=bb.dtype).pin_memory(), torch.cuda.Event())
            with torch.cuda.stream(self.down_stream):
                self.down_stream.wait_event(done)
                o[0].copy_(cls, non_blocking=True); o[1].copy_(bb, non_blocking=True)
                cls.record_stream(self.down_stream); bb.record_stream(self.down_stream)
                o[2].record(self.down_stream)
            pending.append(o)
            if len(pending) > behind:                             # hand out an earlier batch while this one computes
                old = pending.pop(0)
                old[2].synchronize()
                yield old[0], old[1]
            j += 1
        th.join()
        for old in pending:
            old[2].synchronize()
            yield old[0], old[1]
