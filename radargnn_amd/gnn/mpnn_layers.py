"""``MPNNConv`` and ``RadarPointGNNConv`` -- drop-in mirrors of gnn/mpnn_layers.py:11-101 and :104-184 of the
reference (same constructor signatures, ``pre_mlp`` / ``post_mlp`` / ``edge_encoder`` attributes and state_dict
keys), with the message passing done by librgnn's fused kernels instead of torch_geometric's
gather -> cat -> Linear -> scatter.

Semantics kept from torch_geometric 2.1 ``MessagePassing`` (flow source_to_target): ``x_j = x[edge_index[0]]``,
``x_i = x[edge_index[1]]``, messages are reduced at ``edge_index[1]``; aggregation max | mean | add, empty
segments give 0.

How the message function is evaluated.  With ``pre_layers == 1`` (the shipped setting) the message MLP is one
Linear over ``cat[x_i, x_j, e]`` with weight ``W = [W_i | W_j | W_e]`` (column layout of mpnn_layers.py:98), and

    aggr_e (W_i x_t + W_j x_s + W_e a_e + b)  =  (W_i x_t + b)  (.)  aggr_e (W_j x_s + W_e a_e)

so the [E, D] x [D, D] product of the reference becomes two node-wise projections P = x W_i^T + b and
Q = x W_j^T (one MFMA launch) plus a fused gather / de x D mat-vec / segmented reduce over the CSR keyed on the
target (rgnn_mpnn_aggregate).  Real-arithmetic identical; fp32 deviation ~1e-6 (tests/test_gpu_gnn.py).
With ``pre_layers > 1`` the first layer is evaluated per edge the same way (rgnn_mpnn_edge_hidden), the
remaining Linear layers run on the [E, D] rows and rgnn_segment_reduce aggregates.
"""
from __future__ import annotations

import os

from typing import Optional, Tuple

import torch
from torch import nn
from torch.nn import ReLU, Sequential

from .. import ops
from . import autograd as AG
from .linear import Linear, run_mlp


# Fold the target term of MPNNConv's message into the update GEMM (see MPNNConv._folded_update_weights): saves the
# [N,C]x[C,D] projection P per layer (-26 % dense FLOPs at the shipped widths).  Module-level switch for A/B tests.
FOLD_TARGET_TERM = True
# (the folded layer always runs as two row-subset launches -- rows with / without incoming edges; the r02 form "dense launch +
#  accumulating correction" went with the {sum, sum of squares} statistics it needed)
TRAIN_FOLDED = os.environ.get("RGNN_NO_TRAIN_FOLDED") is None   # training: foldable layers as ONE autograd node (AG.ConvFoldedFn)
# RGNN_ISO_SIDE=1: the isolated-row launch goes to a side stream (measured +1.3 % on C2; off by default so that every
# kernel runs alone on the device and per-kernel durations in profiles mean what they say)
ISO_SIDE_STREAM = os.environ.get("RGNN_ISO_SIDE") is not None
_SIDE = {}


def _side_stream(device):
    key = torch.device(device).index
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=device)
    return _SIDE[key]


OWN_EDGE_ATTR = os.environ.get("RGNN_NO_OWN_EDGE_ATTR") is None
USE_WINDOW_KERNEL = os.environ.get("RGNN_NO_MPNN_WIN") is None     # max aggregation of dense graphs: rgnn_mpnn_aggregate_win
WINDOW_KERNEL_MIN_DEGREE = float(os.environ.get("RGNN_MPNN_WIN_MIN_DEGREE", "2.5"))
WINDOW_KERNEL_MAX_DEGREE = float(os.environ.get("RGNN_MPNN_WIN_MAX_DEGREE", "28"))      # (graphs whose degree distribution is unknown)
# graphs whose caller knows which share of the edges goes into targets with more than 60 in-edges (a stream of the window kernel
# holds 64 slots: those targets go through the per-target kernel): window kernel below this share, whatever the mean degree
WINDOW_KERNEL_MAX_BIG_SHARE = float(os.environ.get("RGNN_MPNN_WIN_MAX_BIG_SHARE", "0.10"))
WINDOW_KERNEL_MIN_EDGES = int(os.environ.get("RGNN_MPNN_WIN_MIN_EDGES", str(1 << 18)))               # 12 or more edges per node
WINDOW_KERNEL_MIN_EDGES_SPARSE = int(os.environ.get("RGNN_MPNN_WIN_MIN_EDGES_SPARSE", str(1 << 19)))  # fewer (r = 1 m batches)
# TargetCSR.start_win_plan: the window plan's kernels on a side stream beside the feature / embedding launches (C4 batch 4.51 -> 4.46 ms,
# C3 3.65 -> 3.61: tools/attic/plan_side_ab.py); they are the only launches of a kNN step that share the device with another kernel
PLAN_ON_SIDE_STREAM = os.environ.get("RGNN_NO_PLAN_SIDE") is None


class UnsortedEdgeAttr:
    """Edge attributes still in edge order plus the graph that knows their target order: lets DetNetBasic fold the re-ordering
    into the first kernel that reads them (ops.tiny_mlp2) instead of a gather pass of its own."""
    __slots__ = ("raw", "graph")

    def __init__(self, raw: torch.Tensor, graph: "TargetCSR"):
        self.raw, self.graph = raw, graph

    def materialize(self) -> torch.Tensor:
        return self.graph.sort_edge_attr(self.raw)


class DeferredEdgeAttr:
    """Edge attributes (embedded, in target order) that are computed where the first conv layer needs them -- in front of its edge
    stage, behind its isolated-row and source-term launches -- instead of in front of the layer: in a captured step whose edge side
    runs as a branch of the graph (frames.HotPath) the main stream then reaches the join with the branch that much later."""
    __slots__ = ("_build", "_val")

    def __init__(self, build):
        self._build, self._val = build, None

    def get(self) -> torch.Tensor:
        if self._build is not None:
            self._val, self._build = self._build(), None
        return self._val


class TargetCSR:
    """Edges of one forward pass sorted by aggregation target (``edge_index[1]``), shared by all conv layers."""

    def __init__(self, edge_index: torch.Tensor, num_nodes: int, order: Optional[torch.Tensor] = None,
                 symmetric: bool = False, all_sources: bool = False, source_rows: Optional[torch.Tensor] = None,
                 status: Optional[torch.Tensor] = None, rank: Optional[torch.Tensor] = None, split=None, knn_frames=None,
                 own_edges: bool = False, big_edge_fraction: Optional[float] = None, ordered: bool = True):
        # ordered=False (frames.HotPath with a max-aggregation model): the in-edges of a target in whatever order the CSR build's
        # atomics left them -- a maximum does not depend on it, and the general builder saves its ranking pass (kNN batches of large
        # frames: 59 us of a 64 x 3000-point, k = 20 batch).  Sums (mean / add) need the stable order for run-to-run equal bits.
        self.num_nodes = num_nodes
        # share of the edges whose target has more than 60 incoming edges, when the caller knows it (frames.HotPath reads it with the
        # edge count of a radius graph): decides between the window and the per-edge form of the max aggregation (wants_window_kernel)
        self.big_edge_fraction = big_edge_fraction
        # all_sources: every node has outgoing edges (kNN graphs) -- the source term is needed on every row
        self.all_sources = all_sources
        # symmetric: every edge (s, t) comes with (t, s) -- radius graphs.  A node without incoming edges then has no
        # outgoing ones either, so the layers skip the source-term GEMM on those rows (nothing gathers them).
        self.symmetric = symmetric
        self.num_edges = edge_index.shape[1]
        # optional visiting order of the targets (int32 [N]); a spatially coherent one (grid-cell order) keeps the
        # gathered rows in L2.  The CSR segments are laid out in that order so the kernels stream them.  Purely a
        # scheduling choice: results do not depend on it.
        self.order = order
        self.edge_index = edge_index
        # (``rank``: the inverse of ``order`` when the caller already has it -- the grid build writes both)
        if order is None:
            rank = None
        elif rank is None:
            rank = ops.invert_permutation(order)
        self._rank = rank
        # source_rows: rowptr of a symmetric graph's edge list grouped by source (what the radius search emits) -- the CSR by
        # target then needs no histogram and no sort (ops.csr_by_target)
        # knn_frames = (frame_ptr int64 [F + 1] on the device, k, largest frame): the batch is F kNN graphs laid back to back
        # (edge e = i k + j) -- the CSR is then built by one launch, one block per frame, which also leaves the in-degrees behind
        self._frames_split = None
        # Worth it for MANY SMALL frames only (measured: 512 frames x 300 points, k = 20: step 4.22 -> 4.16 ms; 64 frames x 3000
        # points: 5.22 -> 5.38 ms; one 3000-point frame: 0.35 -> 0.44 ms -- one block per frame cannot hide the latency of 60 000
        # dependent edge visits, the general builder spreads every phase over the chip)
        if (knn_frames is not None and not symmetric and num_nodes > 0 and edge_index.shape[1] == num_nodes * knn_frames[1]
                and 0 < knn_frames[2] * knn_frames[1] <= 8192 and knn_frames[0].numel() - 1 >= 64
                and os.environ.get("RGNN_NO_CSR_FRAMES") is None):
            fptr, k_nn, biggest = knn_frames
            self.rowptr, self.src, self.perm, indeg, per_frame = ops.csr_by_target_frames(edge_index, num_nodes, k_nn, fptr,
                                                                                          biggest, rank)
            self._frames_split = (indeg, fptr, per_frame)
        elif own_edges and symmetric and source_rows is not None and OWN_EDGE_ATTR:
            # the caller's edge attributes are antisymmetric under reversal (relative_position, directed): the attributes in
            # target order are MINUS those of the own out-edge at each slot, so the CSR build skips the search for the twin's
            # edge id (``own_edge``; ``perm`` is computed on first use by whoever still wants it).  No symmetry check on this
            # path (rgnn.h): ``own_edges=True`` is for edge lists out of this library's own radius search only.
            self._sym_args = (edge_index, num_nodes, rank, source_rows, status)
            self.rowptr, self.src, self.own_edge = ops.csr_by_target(edge_index, num_nodes, rank, symmetric_rows=source_rows,
                                                                     status=status, own_edges=True)
            self._perm = None
        else:
            self.rowptr, self.src, self.perm = ops.csr_by_target(edge_index, num_nodes, rank,
                                                                 symmetric_rows=source_rows if symmetric else None,
                                                                 status=status, ordered=ordered or symmetric)
        # (the chunk table of the per-edge kernels is built by whoever first asks for it: a graph whose aggregation goes through the
        #  window kernel never does -- one launch less per step on the headline workload)
        self._chunks = None

        self._empty = split      # (ops.split_targets(...) when the caller already has it -- frames.HotPath on radius graphs)

    own_edge = None

    @property
    def chunks(self) -> Optional[torch.Tensor]:
        """Work-balanced wave chunks for the fused per-edge message kernels (ops.mpnn_partition), shared by all layers."""
        if self._chunks is None and self.num_nodes > 0:
            self.join_csr()
            self._chunks = ops.mpnn_partition(self.rowptr, self.num_edges)
        return self._chunks

    @property
    def perm(self) -> torch.Tensor:
        if self._perm is None:                             # (built without the twin search: do it now)
            ei, n, rank, rows, status = self._sym_args
            self._perm = ops.csr_by_target(ei, n, rank, symmetric_rows=rows, status=status)[2]
        return self._perm

    @perm.setter
    def perm(self, value: torch.Tensor) -> None:
        self._perm = value

    def sort_edge_attr(self, edge_attr: torch.Tensor, lazy: bool = False):
        if lazy and not AG.is_recording():
            return UnsortedEdgeAttr(edge_attr, self)
        if AG.is_recording() and edge_attr.requires_grad:
            if getattr(self, "_inv_perm", None) is None:
                self._inv_perm = ops.invert_permutation(self.perm)
            return AG.PermuteRowsFn.apply(edge_attr, self.perm, self._inv_perm)   # differentiable w.r.t. the edge attributes
        return ops.gather_rows(edge_attr, self.perm)

    def wants_window_kernel(self) -> bool:
        """Whether the max aggregation of this graph goes through the window kernel (ops.mpnn_aggregate_win): it pays where a
        window of consecutive targets shares its sources -- measured 1.2x (D = 464) to 1.65x (D = 144, 272) on 64 frames with
        k = 20, and 144 us against 188 - 200 per launch inside the model on the r = 1 m batches of the headline workload (4 edges
        per node; the captured step 2.16 -> 2.06 ms with the plan on the side stream; profiles/r04_mpnn_win_bench.txt).  Not on
        crowded clouds (34 neighbours on average: a stream holds one or two targets, 3 % of the edges belong to targets too large
        for a stream and go through the per-target kernel: 4.99 vs 4.86 ms on the 100 000-point cloud).
        Rule: 2.5 <= edges per node, at most 10 % of the edges in targets with more than 60 in-edges where the caller knows that share
        (else: fewer than 28 edges per node) and at least 2^18 edges -- 2^19 below 12 edges per node: the captured C2-model step by batch
        size, per-edge / window: 32 frames (0.40 M edges) 1.393 / 1.411 ms, 48 frames (0.60 M) 1.805 / 1.798, 64 frames (0.80 M)
        2.266 / 2.152 (tools/win_threshold_probe.py) -- smaller launches do not pay for the plan and leave work-groups idle."""
        dense = self.num_edges >= 12 * self.num_nodes
        if getattr(self, "big_edge_fraction", None) is not None:
            # r05 (tools/density_sweep.py, profiles/r05_density_sweep.txt): what decides is not the MEAN degree but the share of
            # the edges in targets too large for a 64-slot stream, which the per-target kernel takes one by one -- RadarScenes-
            # shaped batches with r = 2 ... 5 m and the stress cloud with r = 0.5 ... 1.5 m: window / per-edge 1.05 - 1.9 x up to a
            # share of 7 %, level at 13 %, 0.9 - 0.3 x from 19 %
            crowded_ok = self.big_edge_fraction < WINDOW_KERNEL_MAX_BIG_SHARE
        else:
            crowded_ok = self.num_edges < WINDOW_KERNEL_MAX_DEGREE * self.num_nodes
        # ... and only with a visiting ORDER (grid-cell order from the search: consecutive targets are neighbours in space and share
        # their sources).  In node order -- the model's public forward on a bare edge_index; sensors do not deliver points cluster
        # by cluster -- a window's ~450 slots name ~450 different rows, more than a stage holds, and whole windows go through the
        # per-target kernel (r05, tools/profile_train.sh: 652 us per launch against 198 us of the per-edge kernel).
        return (USE_WINDOW_KERNEL and self.num_nodes > 0 and self.order is not None
                and self.num_edges >= (WINDOW_KERNEL_MIN_EDGES if dense else WINDOW_KERNEL_MIN_EDGES_SPARSE)
                and WINDOW_KERNEL_MIN_DEGREE * self.num_nodes <= self.num_edges and crowded_ok
                and self.num_nodes < (1 << 24))

    def start_win_plan(self) -> None:
        """Build the window plan NOW on a side stream (ops.ctx().side_stream), behind everything the calling stream has queued: the
        plan needs the CSR only, and its kernels are short dependent launches of a few hundred waves (greedy packing, a one-block
        scan) that leave the chip to the feature / embedding / first dense launches that follow on the calling stream.  ``win_plan``
        (first aggregation) or ``join_win_plan`` makes the calling stream wait for it.  Works inside a stream capture (fork / join)."""
        if getattr(self, "_win_plan", None) is not None or not (PLAN_ON_SIDE_STREAM and self.wants_window_kernel()):
            return
        main = torch.cuda.current_stream(self.rowptr.device)
        side = ops.ctx().side(self.rowptr.device)
        plan = ops.mpnn_win_plan_buffer(self.rowptr, self.src)              # allocated on the calling stream: freed and reused there
        if side.cuda_stream != main.cuda_stream:                             # (the CSR itself may have been built on the side stream:
            side.wait_stream(main)                                           #  frames.HotPath, edge side of a captured step)
        with torch.cuda.stream(side):
            ops.mpnn_win_plan(self.rowptr, self.src, self.order, out=plan)
        self._win_plan, self._win_plan_pending = plan, side

    def build_win_plan_here(self) -> None:
        """The window plan on the CALLING stream, now (frames.HotPath, pipelined steps: the whole graph stage already runs on a branch
        of its own beside the previous batch's model stage; a branch forked from a branch ends a HIP capture in a segfault)."""
        if getattr(self, "_win_plan", None) is None and self.wants_window_kernel():
            self._win_plan = ops.mpnn_win_plan(self.rowptr, self.src, self.order)

    def mark_csr_on(self, side: "torch.cuda.Stream") -> None:
        """The CSR (and whatever else the caller made of the edges: their attributes) was built on ``side`` (frames.HotPath: the edge
        side of a captured step).  ``join_csr`` makes the calling stream wait for it -- placed in front of the first launch that
        reads rowptr / src / own_edge / the edge attributes (DetNetBasic._forward_graph, MPNNConv._forward_folded)."""
        ev = torch.cuda.Event()
        ev.record(side)
        self._csr_pending = ev

    def join_csr(self) -> None:
        ev = getattr(self, "_csr_pending", None)
        if ev is not None:
            torch.cuda.current_stream(self.rowptr.device).wait_event(ev)
            self._csr_pending = None

    def join_win_plan(self) -> None:
        side = getattr(self, "_win_plan_pending", None)
        if side is not None:
            cur = torch.cuda.current_stream(self.rowptr.device)
            if cur.cuda_stream != side.cuda_stream:              # (a stream waiting for itself ends a HIP capture in a segfault)
                cur.wait_stream(side)
            self._win_plan_pending = None

    def win_plan(self) -> torch.Tensor:
        """The window plan (ops.mpnn_win_plan), built on first use (or ahead of it: start_win_plan), once per graph, shared by all
        layers."""
        if getattr(self, "_win_plan", None) is None:
            self._win_plan = ops.mpnn_win_plan(self.rowptr, self.src, self.order)
        self.join_win_plan()
        return self._win_plan

    def in_degree(self) -> torch.Tensor:
        """float32 [N, 1] number of incoming edges per node (node numbering, not visiting order)."""
        if getattr(self, "_deg", None) is None:
            self.join_csr()                                   # (reads rowptr: the CSR may have been built on a side branch)
            seg = (self.rowptr[1:] - self.rowptr[:-1]).to(torch.float32)
            if self.order is not None:
                deg = torch.empty_like(seg)
                deg[self.order.long()] = seg
            else:
                deg = seg
            self._deg = deg.view(-1, 1)
        return self._deg

    def source_rows(self):
        """(ids int32 [N], count int64 [1] on the device) of the nodes that have OUTGOING edges -- the only rows of the source
        term Q = x W_j^T that the edge stage gathers -- or None when every node is one (``all_sources``).  Symmetric graphs:
        the nodes with incoming edges.  Otherwise found once per graph from the out-degrees (rgnn_source_rowptr)."""
        if self.all_sources or self.num_nodes == 0:
            return None
        if self.symmetric:
            return self.split_targets()[3:5]
        if getattr(self, "_src_rows", None) is None:
            self.join_csr()
            rowptr_s = self._source[0] if getattr(self, "_source", None) is not None else \
                ops.source_rowptr(self.edge_index, self.num_nodes, self._rank)
            sp = ops.split_targets(rowptr_s, self.order, rank=self._rank, by_node=True)
            self._src_rows = (sp[3], sp[4])
        return self._src_rows

    def edge_maps(self):
        """For the backward of the max aggregation (rgnn_mpnn_max_bwd), once per graph: (tgt_sorted int32 [E] target node of
        every row of the target-sorted edge list, eloc_sorted int32 [E] its index inside the target's segment, tloc int32 [E]
        that index for every out-edge of the source CSR) -- or None if an in-degree exceeds 65 535 (the winners are recorded
        as uint16 in-segment indices; one host read per graph)."""
        if getattr(self, "_edge_maps", None) is None:
            if self.num_edges == 0 or int((self.rowptr[1:] - self.rowptr[:-1]).max().item()) > 65535:
                self._edge_maps = False
            else:
                tgt = self.edge_index[1][self.perm.long()]                                  # node id of the sorted edge's target
                seg = tgt if self._rank is None else self._rank.long()[tgt]                 # ... and its segment
                eloc = (torch.arange(self.num_edges, device=tgt.device) - self.rowptr.long()[seg]).to(torch.int32)
                tpos = self.source_csr()[2]
                self._edge_maps = (tgt.to(torch.int32).contiguous(), eloc.contiguous(), eloc[tpos.long()].contiguous())
        return self._edge_maps or None

    def source_csr(self):
        """The same edges keyed on their SOURCE, for the backward pass (gradients w.r.t. the gathered rows become a
        gather instead of atomics): (rowptr_s int32 [N+1], tnode int32 [E] target of each out-edge, tpos int32 [E]
        position of that edge in this object's target-sorted order).  Built on first use, once per graph."""
        if getattr(self, "_source", None) is None:
            rowptr_s, tnode, perm_s = ops.csr_by_target(self.edge_index.flip(0).contiguous(), self.num_nodes, self._rank)
            if getattr(self, "_inv_perm", None) is None:
                self._inv_perm = ops.invert_permutation(self.perm)
            inv_t = self._inv_perm                                        # original edge id -> position in target order
            self._source = (rowptr_s, tnode, inv_t[perm_s.long()].contiguous())
        return self._source

    def split_targets(self):
        """(ids of nodes without incoming edges int32 [N], their count int64 [1] on the device, slot of a node in that
        list int32 [N], ids of the nodes WITH incoming edges int32 [N], their count int64 [1]); once per graph.  The lists
        ascend by node id (what the row-subset dense launches read fastest), not by visiting order."""
        if self._empty is None:
            self.join_csr()                                   # (computed from rowptr / the in-degrees the CSR build left: ADVICE r05)
            if self._frames_split is not None and ops.SORTED_ROW_LISTS:
                self._empty = ops.split_by_degree_frames(*self._frames_split)     # (in-degrees left behind by the CSR build)
            else:
                self._empty = ops.split_targets(self.rowptr, self.order, rank=self._rank, by_node=True)
        return self._empty

    def empty_targets(self):
        return self.split_targets()[:3]


def _cache_key(tensors):
    """Identity of a set of weight tensors for the folded-weight caches: the tensors themselves (kept alive by the key, so
    their storage -- and with it the address -- cannot be recycled by a replacement Parameter), their version counters, the
    global epoch, and where their data lives right now: ``module.to(device)`` / ``.float()`` swap ``param.data`` under the
    SAME Parameter object without touching its version counter."""
    return (ops.CACHE_EPOCH, tuple(tensors), tuple(t._version for t in tensors),
            tuple((t.data_ptr(), t.device, t.dtype) for t in tensors))


def _same_tensor(x: torch.Tensor, y: torch.Tensor) -> bool:
    # `.detach()` hands out a new Python object over the same memory on every call (DetNetBasic passes the edge-embedding
    # tail that way), so identity is storage address + geometry, not `is`
    return x is y or (x.data_ptr() == y.data_ptr() and x.shape == y.shape and x.stride() == y.stride() and x.dtype == y.dtype)


def _same_key(a, b) -> bool:
    return (a is not None and a[0] == b[0] and len(a[1]) == len(b[1]) and all(_same_tensor(x, y) for x, y in zip(a[1], b[1]))
            and a[2] == b[2] and a[3] == b[3])


def _message_mlp(dim: int, layers: int) -> Sequential:
    mods = [Linear(dim, dim)]
    for _ in range(layers - 1):
        mods += [ReLU(), Linear(dim, dim)]
    return Sequential(*mods)


def _update_mlp(in_dim: int, out_dim: int, layers: int) -> Sequential:
    mods = [Linear(in_dim, out_dim)]
    for _ in range(layers - 1):
        mods += [ReLU(), Linear(out_dim, out_dim)]
    return Sequential(*mods)


def _fold_edge_tail(We: torch.Tensor, p_bias: Optional[torch.Tensor], edge_tail):
    """W_e (W a + b) = (W_e W) a + W_e b: apply a trailing Linear of the edge embedding to the [D, De] weight once
    instead of to every edge (halves the per-edge work when the embedding ends 8 -> 16)."""
    if edge_tail is None:
        return We, p_bias
    tw, tb = edge_tail                                          # [De, Dz], [De]
    if tb is not None:
        extra = ops.linear(We, tb.view(1, -1)).view(-1)         # W_e b  [D]
        p_bias = extra if p_bias is None else p_bias + extra
    return ops.linear(We, tw.t().contiguous()), p_bias          # W_e W  [D, Dz]


# per-instance caches of weight-derived tensors (folded weights, summed biases): recomputed on first use, keyed on parameter identity
# and version -- never part of a pickle (torch.save(model) / copy.deepcopy(model), gnn/trainer.py:128-130,342-354) or of a state_dict
_CACHE_ATTRS = ("_fold_val", "_fold_key", "_edge_fold_val", "_edge_fold_key", "_tail_val", "_tail_key", "_sum_bias_val", "_sum_bias_key",
                "_sum_bias_keep", "_neg_cache", "_heads_val", "_heads_key")


def _state_without_caches(module: nn.Module) -> dict:
    return {k: v for k, v in module.__dict__.items() if k not in _CACHE_ATTRS}


class _ConvBase(nn.Module):
    aggr: str

    def __getstate__(self):
        return _state_without_caches(self)

    def reset_parameters(self):
        if getattr(self, "use_edge_encoder", False):
            self.edge_encoder.reset_parameters()
        for seq in (self.pre_mlp, self.post_mlp):
            for m in seq:
                if hasattr(m, "reset_parameters"):
                    m.reset_parameters()

    # ---- shared edge stage -------------------------------------------------------------------------
    def _aggregate(self, P, p_bias, Q, We, ea_sorted, graph: TargetCSR, skip_empty_rows: bool = False) -> torch.Tensor:
        linears = [m for m in self.pre_mlp if isinstance(m, Linear)]
        wide = ea_sorted is not None and ea_sorted.shape[1] > ops.MAX_FUSED_EDGE_WIDTH and graph.num_edges > 0
        if len(linears) == 1 and not wide:
            if (P is None and self.aggr == "max" and graph.wants_window_kernel() and Q.shape[1] <= 2048 and Q.stride(0) % 4 == 0
                    and (ea_sorted is None or ea_sorted.shape[1] <= 8)
                    # (the limits of rgnn_mpnn_aggregate_win, mirrored: it refuses -- RGNN_ERR_UNSUPPORTED -- rows that do not start
                    #  on 16-byte addresses, rows of 2^24 bytes, matrices of 2 GiB: those launches stay on the per-edge kernel)
                    and Q.data_ptr() % 16 == 0 and Q.stride(0) * 4 < (1 << 24)
                    and graph.num_nodes * max(Q.stride(0), (Q.shape[1] + 31) // 32 * 32) * 4 < (1 << 31)):
                # dense neighbourhoods (k = 20, crowded clouds): the distinct source rows of a window of targets staged in LDS
                # (rgnn_mpnn_aggregate_win) instead of one row gather per edge
                return ops.mpnn_aggregate_win(p_bias, Q, We, ea_sorted, graph.rowptr, graph.src, graph.win_plan(),
                                              node_order=graph.order, skip_empty_rows=skip_empty_rows)
            return ops.mpnn_aggregate(P, p_bias, Q, We, ea_sorted, graph.rowptr, graph.src, self.aggr,
                                      node_order=graph.order, chunks=graph.chunks, skip_empty_rows=skip_empty_rows)
        if wide:
            # more edge attributes than the fused kernels take (wider than any embedding the reference ships): the edge term
            # as a dense launch over the edge rows, added to the gathered node terms
            hidden = ops.mpnn_edge_hidden(P, p_bias, Q, None, None, graph.rowptr, graph.src, relu=False, node_order=graph.order,
                                          chunks=graph.chunks)
            hidden += ops.linear(ea_sorted, We.contiguous())
            if len(linears) == 1:
                return ops.segment_reduce(hidden, graph.rowptr, self.aggr, node_order=graph.order)
            hidden = torch.relu_(hidden)
        else:
            hidden = ops.mpnn_edge_hidden(P, p_bias, Q, We, ea_sorted, graph.rowptr, graph.src, relu=True,
                                          node_order=graph.order, chunks=graph.chunks)
        for j, lin in enumerate(linears[1:]):
            last = j == len(linears) - 2
            hidden = ops.linear(hidden, lin.weight.detach(), lin.bias.detach(), relu=not last)
        return ops.segment_reduce(hidden, graph.rowptr, self.aggr, node_order=graph.order)

    def _needs_grad(self, x, ea_sorted, edge_tail=None) -> bool:
        return AG.is_recording()

    def _aggregate_grad(self, P, p_bias, Q, We, ea_sorted, graph: TargetCSR) -> torch.Tensor:
        """Differentiable form of ``_aggregate``: the aggregate of the source / edge part runs in the fused kernel
        (AG.aggregate), the target term is added by autograd-visible elementwise ops."""
        linears = [m for m in self.pre_mlp if isinstance(m, Linear)]
        if len(linears) != 1:
            # deeper message MLP: first layer per edge, the remaining Linears on the [E, D] rows, segmented reduce
            hidden = AG.edge_rows(P, p_bias, Q, We, ea_sorted, graph, True)
            for j, lin in enumerate(linears[1:]):
                hidden = AG.linear(hidden, lin.weight, lin.bias, relu=j != len(linears) - 2)
            return AG.SegmentReduceFn.apply(hidden, graph, self.aggr)
        M = AG.aggregate(Q, We, ea_sorted, graph, self.aggr)
        term = P
        if p_bias is not None:
            term = p_bias.view(1, -1) if term is None else term + p_bias.view(1, -1)
        if term is None:
            return M
        deg = graph.in_degree()
        scale = deg if self.aggr in ("add", "sum") else (deg > 0).to(torch.float32)
        return M + scale * term

    def forward(self, x: torch.Tensor, edge_index: torch.Tensor, edge_attr: torch.Tensor) -> torch.Tensor:
        graph = TargetCSR(edge_index, x.shape[0])

        def run(x_, ea_):
            return (self.forward_sorted(x_, graph, graph.sort_edge_attr(ea_))[0],)

        params = list(self.parameters())
        if not AG.is_recording() and AG.grad_mode(x, edge_attr, *params):
            if x.requires_grad or edge_attr.requires_grad:     # a training step (see DetNetBasic.forward): record once
                with AG.recording(direct=True):
                    return run(x, edge_attr)[0]
            return AG.checkpointed(run, (x, edge_attr), params)[0]
        return run(x, edge_attr)[0]


class MPNNConv(_ConvBase):
    """General MPNN layer with edge features (reference: gnn/mpnn_layers.py:11-101)."""

    def __init__(self, in_channels: int, out_channels: int, edge_dim: int, aggr: str = "max",
                 pre_layers: int = 1, post_layers: int = 1, use_edge_encoder: bool = False):
        super().__init__()
        if aggr not in ops.AGGR_CODES:
            raise ValueError(f"unknown aggregation {aggr!r}")
        self.aggr = aggr
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.edge_dim = edge_dim
        self.use_edge_encoder = use_edge_encoder
        if use_edge_encoder:                                   # mpnn_layers.py:54-60
            self.edge_encoder = Linear(edge_dim, in_channels)
            msg_dim = 3 * in_channels
        else:
            msg_dim = 2 * in_channels + edge_dim
        self.pre_mlp = _message_mlp(msg_dim, pre_layers)       # mpnn_layers.py:64-68
        self.post_mlp = _update_mlp(msg_dim + in_channels, out_channels, post_layers)   # :70-74
        self.reset_parameters()

    def can_fold_input_tail(self, x: torch.Tensor) -> bool:
        """Can this layer take its input as ``x @ W^T + b`` with the Linear (W, b) folded into its own weights (``x_tail``)?  The
        folded inference form reads the input only through linear maps (W_i, W_j, W_post,x)."""
        return (not AG.is_recording() and self._can_fold_target_term()
                and os.environ.get("RGNN_NO_INPUT_TAIL_FOLD") is None)

    def frames_fusable(self, x: torch.Tensor, graph: TargetCSR, k1: Optional[int] = None) -> bool:
        """Can this layer run on frame-padded row lists (``forward_sorted(..., frames=...)``): the folded inference form, a
        symmetric graph (its source rows are its targets with edges), all three dense launches on the LDS-DMA kernel -- the only
        one that skips the -1 entries of such a list (ops.linear refuses any other).  ``k1``: width of the node matrix the layer
        reads (its own input width, or the narrower one in front of a folded node-embedding tail)."""
        k1 = self.in_channels if k1 is None else k1
        return (graph.symmetric and not graph.all_sources and self._can_fold_target_term()
                and not self._needs_grad(x, None) and self._dense_kernels_take_affine(x) and k1 <= 512
                and k1 % 32 == 0                # (the update launches read [x | m]: the LDS-DMA kernel wants k1 in whole 32-column steps then)
                and (not ISO_SIDE_STREAM))

    def forward_sorted(self, x: torch.Tensor, graph: TargetCSR, ea_sorted: torch.Tensor, want_stats: bool = False,
                       edge_tail=None, x_affine: Optional[torch.Tensor] = None, x_tail=None, frames=None
                       ) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
        """``ea_sorted``: edge attributes already in ``graph`` order.  ``edge_tail = (W, b)``: the edge attributes
        this layer is defined on are ``ea_sorted @ W^T + b`` (the last Linear of DetNetBasic's edge embedding); it is
        folded into W_e here instead of being applied to every edge.  ``x_affine`` [ops.AFFINE_ROWS, C]: the layer input is
        relu(BatchNorm(x)) with that apply table -- the BatchNorm + ReLU DetNetBasic applies after the previous conv -- left to this layer's
        dense kernels (the folded inference form) instead of a pass of its own."""
        if frames is not None:
            # per-frame BatchNorm statistics on frame-padded row lists (gnn.linear.frame_scope.padded_split): x_affine is the
            # previous BatchNorm's [F, AFFINE_ROWS, C] table, the statistics come back per frame (FrameStats); callers ask frames_fusable first
            return self._forward_folded(x, graph, ea_sorted, want_stats, edge_tail, x_affine, x_tail, frames=frames)
        if isinstance(ea_sorted, DeferredEdgeAttr) and (self._needs_grad(x, None, edge_tail) or not self._can_fold_target_term()):
            ea_sorted = ea_sorted.get()                 # (only the folded inference form below knows a later place for it)
        if x_affine is not None and (self._needs_grad(x, ea_sorted, edge_tail) or not self._can_fold_target_term()
                                     or not self._dense_kernels_take_affine(x)):
            x, x_affine = ops.scale_shift_act(x, x_affine, relu=True), None
        if self._needs_grad(x, ea_sorted, edge_tail):
            return self._forward_grad(x, graph, ea_sorted, want_stats, edge_tail)
        if x_tail is not None and not self.can_fold_input_tail(x):
            raise ValueError("x_tail needs the folded inference form (MPNNConv.can_fold_input_tail)")
        if self._can_fold_target_term():
            return self._forward_folded(x, graph, ea_sorted, want_stats, edge_tail, x_affine, x_tail)
        c = self.in_channels
        lin0 = self.pre_mlp[0]
        W = lin0.weight.detach()
        b = lin0.bias.detach()
        d = W.shape[0]
        # P = x W_i^T + b (target term), Q = x W_j^T (source term): one launch, output [N, 2D]
        pq = ops.linear(x, W[:, :c], b, w2=W[:, c:2 * c])
        P, Q = pq[:, :d], pq[:, d:]
        We = W[:, 2 * c:]
        p_bias = None
        if self.use_edge_encoder:                              # mpnn_layers.py:96-97 folded into W_e
            enc_w = self.edge_encoder.weight.detach()          # [C, De]
            enc_b = self.edge_encoder.bias.detach()
            p_bias = ops.linear(We, enc_b.view(1, -1)).view(-1)                 # W_e b_enc  [D]
            We = ops.linear(We, enc_w.t().contiguous())                        # W_e W_enc  [D, De]
        We, p_bias = _fold_edge_tail(We, p_bias, edge_tail)
        m = self._aggregate(P, p_bias, Q, We, ea_sorted, graph)
        return run_mlp(self.post_mlp, x, a2=m, want_stats=want_stats)          # post_mlp(cat[x, m]) :89-90

    def _dense_kernels_take_affine(self, x) -> bool:
        """Will all three dense launches of the folded form (source term, the two row-split updates) run on the LDS-DMA kernel,
        which can apply a scale / shift to its A1 fragments?  (Widths in whole k-steps of 16, more than 32 output columns,
        enough rows for the bf16x3 path; ops.linear still checks every launch and falls back to a separate pass.)"""
        c, d = self.in_channels, self.pre_mlp[0].weight.shape[0]
        return (ops.FUSE_A1_AFFINE and ops.USE_BF16X3 and c % 16 == 0 and d % 16 == 0 and d > ops.BF16X3_MIN_COLS
                and self.out_channels > ops.BF16X3_MIN_COLS and self.out_channels % 4 == 0 and x.shape[0] >= ops.BF16X3_MIN_ROWS)

    # ---- training form: every parameter stays visible to autograd ----------------------------------------------
    def _forward_grad(self, x, graph, ea_sorted, want_stats, edge_tail):
        if (self._can_fold_target_term() and TRAIN_FOLDED
                and (ea_sorted is None or ea_sorted.shape[1] <= ops.MAX_FUSED_EDGE_WIDTH_BWD)):   # (wider: the general form below)
            return self._forward_grad_folded(x, graph, ea_sorted, want_stats, edge_tail)
        c = self.in_channels
        lin0 = self.pre_mlp[0]
        W, b = lin0.weight, lin0.bias
        d = W.shape[0]
        pq = AG.linear(x, torch.cat([W[:, :c], W[:, c:2 * c]], dim=0), torch.cat([b, torch.zeros_like(b)]))
        P, Q = pq[:, :d], pq[:, d:]
        We, p_bias = W[:, 2 * c:], None
        # (the small weight folds on the HIP kernels too: AG.matmul, not torch's `@`, which would go to the BLAS)
        if self.use_edge_encoder:
            p_bias = AG.matmul(We, self.edge_encoder.bias.view(-1, 1)).view(-1)
            We = AG.matmul(We, self.edge_encoder.weight)
        if edge_tail is not None:
            tw, tb = edge_tail
            if tb is not None:
                extra = AG.matmul(We, tb.view(-1, 1)).view(-1)
                p_bias = extra if p_bias is None else p_bias + extra
            We = AG.matmul(We, tw)
        m = self._aggregate_grad(P, p_bias, Q, We, ea_sorted, graph)
        return run_mlp(self.post_mlp, x, a2=m, want_stats=want_stats)

    def _forward_grad_folded(self, x, graph, ea_sorted, want_stats, edge_tail):
        """Training form of the folded layer: the [C, .]-sized folds are differentiable products of the parameters (on the
        HIP kernels: AG.matmul), everything that touches node / edge data is ONE autograd node whose forward launches the
        inference kernels (AG.ConvFoldedFn)."""
        c = self.in_channels
        W, b = self.pre_mlp[0].weight, self.pre_mlp[0].bias
        post = self.post_mlp[0]
        Wi, Wj, We = W[:, :c], W[:, c:2 * c], W[:, 2 * c:]
        Wpx, Wpm = post.weight[:, :c], post.weight[:, c:]
        wcomb = torch.cat([Wpx + AG.matmul(Wpm, Wi), Wpm], dim=1)                 # [Co, C + D]
        bcomb = post.bias + AG.matmul(Wpm, b.view(-1, 1)).view(-1)
        p_bias = None
        if self.use_edge_encoder:
            p_bias = AG.matmul(We, self.edge_encoder.bias.view(-1, 1)).view(-1)
            We = AG.matmul(We, self.edge_encoder.weight)
        if edge_tail is not None:
            tw, tb = edge_tail
            if tb is not None:
                extra = AG.matmul(We, tb.view(-1, 1)).view(-1)
                p_bias = extra if p_bias is None else p_bias + extra
            We = AG.matmul(We, tw)
        h, stats = AG.ConvFoldedFn.apply(x, ea_sorted, Wj, We, p_bias, wcomb, bcomb, Wpx, post.bias, graph, self.aggr, want_stats)
        return h, stats

    # ---- folded form: the target term W_i x + b never becomes a tensor -----------------------------------------
    def _can_fold_target_term(self) -> bool:
        return (FOLD_TARGET_TERM and self.aggr in ("max", "mean") and len(self.pre_mlp) == 1 and len(self.post_mlp) == 1)

    def _folded_update_weights(self):
        """post_mlp(cat[x, m]) with m = 1[deg>0] (W_i x + b + M):

            h = (W_px + W_pm W_i) x + W_pm M' + (b_post + W_pm b)      for targets with incoming edges
            h =  W_px x + b_post                                       for isolated targets (m = 0)

        (M' = aggregated source/edge part, exactly 0 for isolated targets).  Returns the combined weight
        [W_px + W_pm W_i | W_pm] and the combined bias.  Cached until a parameter is modified in place / replaced."""
        pre, post = self.pre_mlp[0], self.post_mlp[0]
        key = _cache_key((pre.weight, pre.bias, post.weight, post.bias))
        if not _same_key(getattr(self, "_fold_key", None), key):
            c = self.in_channels
            W, b = pre.weight.detach(), pre.bias.detach()
            Wp, bp = post.weight.detach(), post.bias.detach()
            Wi = W[:, :c]
            Wpx, Wpm = Wp[:, :c], Wp[:, c:]
            wfold = ops.linear(Wpm, Wi.t().contiguous())                 # W_pm W_i   [Co, C]
            bfold = ops.linear(Wpm, b.view(1, -1)).view(-1)              # W_pm b     [Co]
            self._fold_val = (torch.cat([Wpx + wfold, Wpm], dim=1).contiguous(), (bp + bfold).contiguous())
            self._fold_key = key
        return self._fold_val

    def _folded_edge_weights(self, edge_tail):
        """(W_e with the edge encoder / the edge-embedding tail folded in, its bias term) -- [D, De']-sized products of
        parameters only, so they are cached until a parameter changes (4 small launches per layer and step otherwise)."""
        tensors = [self.pre_mlp[0].weight]
        if self.use_edge_encoder:
            tensors += [self.edge_encoder.weight, self.edge_encoder.bias]
        if edge_tail is not None:
            tensors += [t for t in edge_tail if t is not None]
        key = _cache_key(tensors)
        if not _same_key(getattr(self, "_edge_fold_key", None), key):
            c = self.in_channels
            We, p_bias = self.pre_mlp[0].weight.detach()[:, 2 * c:], None
            if self.use_edge_encoder:
                enc_w, enc_b = self.edge_encoder.weight.detach(), self.edge_encoder.bias.detach()
                p_bias = ops.linear(We, enc_b.view(1, -1)).view(-1)
                We = ops.linear(We, enc_w.t().contiguous())
            self._edge_fold_val = _fold_edge_tail(We, p_bias, edge_tail)
            self._edge_fold_key = key
        return self._edge_fold_val

    def _input_tail_weights(self, x_tail, edge_tail):
        """The layer's weights with a Linear in front of it folded in: the layer input is x = t W_t^T + b_t (the last Linear of
        DetNetBasic's node embedding, gnn_models.py:137-178: no activation follows it), and the folded inference form reads x only
        through linear maps, so

            Q = x W_j^T            = t (W_j W_t)^T + W_j b_t         (the constant joins the per-target bias: it passes the max)
            h = [x | M] W_comb^T   = [t | M] [W_comb,x W_t | W_comb,m]^T + W_comb,x b_t
            h_isolated = x W_px^T  = t (W_px W_t)^T + W_px b_t

        -- the [N, C] embedding output is never computed (its GEMM, 128 -> 224 at the shipped widths, was 65 us and 270 MB per
        step) and all three dense launches of the layer shrink from K = C to the tail's input width.  Exact in real arithmetic
        like the other folds; cached per weight version."""
        tw, tb = x_tail
        pre, post = self.pre_mlp[0], self.post_mlp[0]
        tensors = [pre.weight, pre.bias, post.weight, post.bias, tw] + ([tb] if tb is not None else [])
        key = _cache_key(tensors)
        if not _same_key(getattr(self, "_tail_key", None), key):
            c = self.in_channels
            wcomb, bcomb = self._folded_update_weights()
            Wj = pre.weight.detach()[:, c:2 * c]
            Wpx = post.weight.detach()[:, :c]
            twt = tw.detach().t().contiguous()                                  # [c0, C] -> ops.linear(A, B) = A B^T
            wj_t = ops.linear(Wj.contiguous(), twt)                              # W_j W_t        [D, c0]
            wpx_t = ops.linear(Wpx.contiguous(), twt)                            # W_px W_t       [Co, c0]
            wcx_t = ops.linear(wcomb[:, :c].contiguous(), twt)                   # W_comb,x W_t   [Co, c0]
            wcomb_t = torch.cat([wcx_t, wcomb[:, c:]], dim=1).contiguous()
            if tb is not None:
                b = tb.detach().view(1, -1)
                qb = ops.linear(Wj.contiguous(), b).view(-1)                     # W_j b_t        [D]
                bcomb_t = (bcomb + ops.linear(wcomb[:, :c].contiguous(), b).view(-1)).contiguous()
                biso_t = (post.bias.detach() + ops.linear(Wpx.contiguous(), b).view(-1)).contiguous()
            else:
                qb, bcomb_t, biso_t = None, bcomb, post.bias.detach()
            self._tail_val = (wj_t.contiguous(), qb, wcomb_t, bcomb_t, wpx_t.contiguous(), biso_t)
            self._tail_key = key
        return self._tail_val

    def _forward_folded(self, x, graph, ea_sorted, want_stats, edge_tail, x_affine=None, x_tail=None, frames=None):
        c = self.in_channels
        n = x.shape[0]
        W = self.pre_mlp[0].weight.detach()
        post = self.post_mlp[0]
        wcomb, bcomb = self._folded_update_weights()
        w_src, w_iso, b_iso, q_bias = W[:, c:2 * c], post.weight.detach()[:, :c], post.bias.detach(), None
        if x_tail is not None:
            w_src, q_bias, wcomb, bcomb, w_iso, b_iso = self._input_tail_weights(x_tail, edge_tail)
        lst_e, cnt_e, _, lst_ne, cnt_ne = graph.split_targets()
        tiles_ne = tiles_e = None
        n_list = n
        if frames is not None:                      # the same lists, every frame in 256-row tiles of its own (-1 entries pad them)
            lst_ne, cnt_ne, tiles_ne, _ = frames["ne"]
            lst_e, cnt_e, tiles_e, _ = frames["e"]
            n_list = max(lst_ne.numel(), lst_e.numel())
            if x_affine is None:
                tiles_ne = tiles_e = None
        stats = main_stats = iso_stats = None
        if want_stats:
            panels = max(ops.stat_panels(n_list), 1)
            # one panel set per launch, uninitialised: BatchNorm reads only the panels the launch's row count reaches
            stats = torch.empty((2 * panels, ops.STAT_ROWS, wcomb.shape[0]), dtype=torch.float32, device=x.device)
            main_stats, iso_stats = stats[:panels], stats[panels:]
            stats = ops.StatParts([(main_stats, cnt_ne), (iso_stats, cnt_e)]) if frames is None else FrameStats(main_stats, iso_stats)
        side = None
        # Two row-subset launches, each with the weights its rows need: targets with incoming edges get the folded
        # update (K = C + D), isolated targets (m = 0) the plain W_px x + b_post (K = C).  The second one needs x
        # only; optionally (ISO_SIDE_STREAM) it runs on a side stream beside the source-term GEMM and the edge kernel.
        h = torch.empty((n, wcomb.shape[0]), dtype=torch.float32, device=x.device)
        if ISO_SIDE_STREAM:
            side = _side_stream(x.device)
            side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                                 # (side = None: stays on the current stream)
            if side is not None:
                with ops.no_splitk_workspace():                       # (may overlap main-stream launches that use the scratch)
                    ops.linear(x, w_iso, b_iso, out=h, row_index=lst_e, m_dev=cnt_e, stats_out=iso_stats, a1_affine=x_affine)
            else:
                ops.linear(x, w_iso, b_iso, out=h, row_index=lst_e, m_dev=cnt_e, stats_out=iso_stats, a1_affine=x_affine,
                           a1_affine_tiles=tiles_e, padded_row_list=frames is not None)
        src_rows = graph.source_rows()
        if frames is not None:
            src_rows = (lst_ne, cnt_ne)                 # (symmetric graph: the sources are the targets with edges)
        if src_rows is not None:
            # source term only on the nodes that have outgoing edges: nothing gathers the other rows of Q
            Q = ops.linear(x, w_src, row_index=src_rows[0], m_dev=src_rows[1], a1_affine=x_affine, a1_affine_tiles=tiles_ne,
                           padded_row_list=frames is not None, out=ops.padded_rows(n, w_src.shape[0], x.device))
        else:
            Q = ops.linear(x, w_src, a1_affine=x_affine, out=ops.padded_rows(n, w_src.shape[0], x.device))   # source term only: [N, D]
        if isinstance(ea_sorted, DeferredEdgeAttr):
            ea_sorted = ea_sorted.get()                                   # (joins the edge side of a captured step: TargetCSR.join_csr)
        graph.join_csr()
        We, p_bias = self._folded_edge_weights(edge_tail)
        if q_bias is not None:                                            # (a constant per channel passes the max / mean)
            p_bias = q_bias if p_bias is None else self._sum_bias(p_bias, q_bias)
        # 1[deg>0] (p_bias + aggr_e(Q[s] + W_e a_e)); the update below reads M on the targets with edges only
        M = self._aggregate(None, p_bias, Q, We, ea_sorted, graph, skip_empty_rows=True)
        ops.linear(x, wcomb, bcomb, a2=M, out=h, row_index=lst_ne, m_dev=cnt_ne, stats_out=main_stats, a1_affine=x_affine,
                   a1_affine_tiles=tiles_ne, padded_row_list=frames is not None)
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)
        return h, stats

    def _sum_bias(self, a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
        """a + b for two [D] bias vectors, cached on their identities (one elementwise launch per weight version, not per step)."""
        key = (a.data_ptr(), a._version, b.data_ptr(), b._version)
        if getattr(self, "_sum_bias_key", None) != key:
            self._sum_bias_val, self._sum_bias_key, self._sum_bias_keep = (a + b).contiguous(), key, (a, b)
        return self._sum_bias_val

    def message(self, x_i: torch.Tensor, x_j: torch.Tensor, edge_attr: torch.Tensor) -> torch.Tensor:
        """Per-edge message exactly as the reference spells it (mpnn_layers.py:94-101); not used by forward."""
        if self.use_edge_encoder:
            edge_attr = self.edge_encoder(edge_attr)
        return run_mlp(self.pre_mlp, torch.cat([x_i, x_j, edge_attr], dim=-1))[0]


class FrameStats:
    """Column statistics of a conv layer's two row-split update launches on frame-padded row lists: one partial per 128-row panel
    of each list (targets with edges / isolated targets), summed per frame by BatchNorm.scale_shift_frames."""

    def __init__(self, main: torch.Tensor, iso: torch.Tensor):
        self.main, self.iso = main, iso


class RadarPointGNNConv(_ConvBase):
    """Radar-PointGNN convolution with edge features and a residual connection
    (reference: gnn/mpnn_layers.py:104-184).  Output width == input width."""

    def __init__(self, init_node_dim: int, init_edge_dim: int, aggr: str = "max", pre_layers: int = 1,
                 post_layers: int = 1):
        super().__init__()
        if aggr not in ops.AGGR_CODES:
            raise ValueError(f"unknown aggregation {aggr!r}")
        self.aggr = aggr
        self.in_channels = init_node_dim
        self.out_channels = init_node_dim
        self.init_node_dim = init_node_dim
        self.init_edge_dim = init_edge_dim
        msg_dim = init_node_dim + init_edge_dim
        self.pre_mlp = _message_mlp(msg_dim, pre_layers)
        self.post_mlp = _update_mlp(msg_dim + init_node_dim, init_node_dim, post_layers)
        self.reset_parameters()

    def forward_sorted(self, x: torch.Tensor, graph: TargetCSR, ea_sorted: torch.Tensor, want_stats: bool = False,
                       edge_tail=None, x_affine: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
        if x_affine is not None:                               # (x is also the residual: this layer wants it materialised)
            x = ops.scale_shift_act(x, x_affine, relu=True)
        c = self.in_channels
        lin0 = self.pre_mlp[0]
        if self._needs_grad(x, ea_sorted, edge_tail):
            W = lin0.weight
            Q = AG.linear(x, W[:, :c])
            We, p_bias = W[:, c:], lin0.bias
            if edge_tail is not None:
                tw, tb = edge_tail
                if tb is not None:
                    p_bias = p_bias + AG.matmul(We, tb.view(-1, 1)).view(-1)
                We = AG.matmul(We, tw)
            m = self._aggregate_grad(None, p_bias, Q, We, ea_sorted, graph)
            return run_mlp(self.post_mlp, x, a2=m, residual=x, want_stats=want_stats)
        W = lin0.weight.detach()
        Q = ops.linear(x, W[:, :c])                            # message = pre_mlp(cat[x_j, e])  :181-182
        We, p_bias = _fold_edge_tail(W[:, c:], lin0.bias.detach(), edge_tail)
        m = self._aggregate(None, p_bias, Q, We, ea_sorted, graph)
        return run_mlp(self.post_mlp, x, a2=m, residual=x, want_stats=want_stats)   # post_mlp(cat[x, m]) + x  :174-177

    def message(self, x_j: torch.Tensor, edge_attr: torch.Tensor) -> torch.Tensor:
        return run_mlp(self.pre_mlp, torch.cat([x_j, edge_attr], dim=-1))[0]
