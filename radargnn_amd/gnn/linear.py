"""``Linear`` and ``BatchNorm`` with the parameter layout of the torch_geometric 2.1 modules the reference builds
its networks from (``torch_geometric.nn.dense.linear.Linear``, ``torch_geometric.nn.BatchNorm``; call sites
gnn/gnn_models.py:5,8,71,86,162,170 and gnn/mpnn_layers.py:4,55,64-74), executing on the HIP path.

``run_mlp`` walks an ``nn.Sequential`` of Linear / BatchNorm / ReLU and fuses what the reference executes op by
op: a ReLU that directly follows a Linear goes into the GEMM epilogue, a BatchNorm that follows a Linear gets
its column statistics from the same epilogue."""
from __future__ import annotations

import math
import os
from typing import Optional, Tuple

import torch
from torch import nn

from .. import ops
from . import autograd as AG


def current_frame_scope():
    """The frame_scope in force for the calling thread (ops.ForwardContext), or None."""
    return ops.ctx().frame_scope


class frame_scope:
    """``with frame_scope(frame_ptr, n_nodes, graph)``: every train-mode BatchNorm executed inside takes its statistics PER FRAME
    (rows [frame_ptr[f], frame_ptr[f + 1]) of a node matrix; the edges into those nodes for an edge matrix) instead of over
    the whole batch.  That is what the reference computes at inference time -- one frame per forward, the model never put in
    eval mode (evaluate.py:40, postprocessor/inference.py:57-62, gnn/gnn_models.py:124-128) -- at batched throughput.
    ``frame_ptr``: int64 [F + 1] on the device (PyG ``Batch.ptr``); ``graph``: the TargetCSR of the forward pass (edge rows)."""

    def __init__(self, frame_ptr: torch.Tensor, n_nodes: int, graph=None):
        self.node_ptr = frame_ptr.to(torch.int64).contiguous()
        self.n_nodes = int(n_nodes)
        self.graph = graph
        self._edge_ptr = None
        self._padded = None
        self.kind = None           # "node" | "edge": what the matrices the BatchNorms see right now hold one row of (rows_of)

    def rows_of(self, kind: Optional[str]):
        """``with scope.rows_of("edge"):`` -- the BatchNorm inputs inside are edge matrices (or "node").  Callers that know say
        so; without it the row count decides, which is ambiguous for a batch with as many edges as nodes."""
        scope = self

        class _Kind:
            def __enter__(self_):
                self_.prev, scope.kind = scope.kind, kind

            def __exit__(self_, *exc):
                scope.kind = self_.prev
                return False
        return _Kind()

    def seg_ptr_for(self, rows: int) -> torch.Tensor:
        n_edges = None if self.graph is None else self.graph.num_edges
        kind = self.kind
        if kind is None and rows == self.n_nodes and rows == n_edges:
            raise ValueError(f"frame_scope: a BatchNorm input with {rows} rows could be the node matrix or the edge matrix of this "
                             "batch (as many edges as nodes): say which with frame_scope.rows_of('node' | 'edge')")
        if kind == "node" or (kind is None and rows == self.n_nodes):
            if rows != self.n_nodes:
                raise ValueError(f"frame_scope: a node matrix has {self.n_nodes} rows, this BatchNorm input has {rows}")
            return self.node_ptr
        if self.graph is not None and rows == n_edges:
            if self._edge_ptr is None:
                # frames are contiguous both in node numbering and in the visiting order of the CSR by target, so the edges
                # into frame f are the CSR positions [rowptr[frame_ptr[f]], rowptr[frame_ptr[f + 1]])
                self._edge_ptr = self.graph.rowptr.index_select(0, self.node_ptr).to(torch.int64).contiguous()
            return self._edge_ptr
        raise ValueError(f"frame_scope: a BatchNorm input with {rows} rows is neither the node matrix ({self.n_nodes} rows) nor "
                         "the edge matrix of the batch")

    def padded_split(self):
        """The graph's two target lists (with / without incoming edges, ascending node ids) cut at the frame borders and padded
        so that every frame starts a 256-row tile of its own (ops.pad_list_by_segment), once per scope:
        ``{"ne": (list, count, tiles, stat_start), "e": (...)}``.  With them the conv layers' dense launches leave column statistics
        per frame behind and apply the previous layer's per-frame BatchNorm on their way in -- no pass over [N, C] for either."""
        if self._padded is None:
            lst_e, cnt_e, _, lst_ne, cnt_ne = self.graph.split_targets()
            ne, e = ops.pad_list_pair_by_segment(lst_ne, cnt_ne, lst_e, cnt_e, self.node_ptr)
            self._padded = {"ne": ne, "e": e}
        return self._padded

    def __enter__(self):
        c = ops.ctx()
        self.prev, c.frame_scope = c.frame_scope, self
        return self

    def __exit__(self, *exc):
        ops.ctx().frame_scope = self.prev
        return False


class Linear(nn.Module):
    """y = x W^T + b with ``weight`` [out, in] and ``bias`` [out] (keys ``weight`` / ``bias``)."""

    def __init__(self, in_channels: int, out_channels: int, bias: bool = True):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels))
        if bias:
            self.bias = nn.Parameter(torch.empty(out_channels))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self):
        # torch_geometric default: kaiming-uniform(a=sqrt(5)) on fan_in = in_channels -> U(-1/sqrt(in), 1/sqrt(in))
        bound = 1.0 / math.sqrt(self.in_channels) if self.in_channels > 0 else 0.0
        with torch.no_grad():
            self.weight.uniform_(-bound, bound)
            if self.bias is not None:
                self.bias.uniform_(-bound, bound)

    def _lazy_load_hook(self, *args, **kwargs):
        """torch_geometric's Linear registers a state_dict pre-hook of this name (lazy in_channels = -1); a reference whole-module
        pickle stores it as a bound method of the layer, so the name must resolve when such a pickle is read with this class standing
        in (radargnn_amd.checkpoint.install_reference_pickle_shims).  Widths are never lazy here: nothing to do."""
        return None

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        squeeze = x.dim() == 1
        if squeeze:                                   # test/test_gnn.py:18 feeds a single feature vector
            x = x.unsqueeze(0)
        if AG.grad_mode(x, self.weight, self.bias):
            y = AG.linear(x, self.weight, self.bias)
        else:
            y = ops.linear(x, self.weight.detach(), None if self.bias is None else self.bias.detach())
        return y.squeeze(0) if squeeze else y

    def extra_repr(self) -> str:
        return f"{self.in_channels}, {self.out_channels}, bias={self.bias is not None}"


class BatchNorm(nn.Module):
    """Parameters live in ``self.module`` (a ``torch.nn.BatchNorm1d`` used as a container only), so the keys are
    ``module.weight|bias|running_mean|running_var|num_batches_tracked`` like the reference's checkpoints.
    ``self.training`` selects batch statistics (the reference never calls ``.eval()``)."""

    def __init__(self, in_channels: int, eps: float = 1e-5, momentum: float = 0.1, affine: bool = True,
                 track_running_stats: bool = True):
        super().__init__()
        self.module = nn.BatchNorm1d(in_channels, eps, momentum, affine, track_running_stats)
        self.in_channels = in_channels

    def reset_parameters(self):
        self.module.reset_parameters()

    def __setstate__(self, state):
        # (a torch_geometric 2.1 BatchNorm pickled inside a reference model keeps nothing but `module`: checkpoint.py)
        super().__setstate__(state)
        if "in_channels" not in self.__dict__ and "module" in self._modules:
            self.in_channels = self._modules["module"].num_features

    def empty_or_single(self, rows: int, width: int) -> bool:
        """What torch's batch_norm does with fewer than two rows while it takes batch statistics: ONE row is an error
        (``Expected more than 1 value per channel when training``, torch/nn/functional.py), ZERO rows pass through -- the
        output is empty, the running statistics stay, num_batches_tracked still counts the call (an edge MLP with BatchNorm on a
        frame without edges).  True: the caller skips the layer."""
        mod = self.module
        if rows > 1 or not (self.training or mod.running_mean is None):
            return False
        if rows == 1:
            raise ValueError(f"Expected more than 1 value per channel when training, got input size torch.Size([1, {width}])")
        if self.training and mod.track_running_stats and mod.num_batches_tracked is not None and not AG.is_reexecution():
            mod.num_batches_tracked += 1
        return True

    def scale_shift(self, stats: Optional[torch.Tensor], m: int, in_bound: Optional[torch.Tensor] = None) -> torch.Tensor:
        """[ops.AFFINE_ROWS, C] apply table (mean_hi, g, t: y = (x - mean_hi) g + t) of this layer for a batch whose column statistics are ``stats``;
        updates the running statistics exactly once (train mode).  ``in_bound``: device word bounding the layer's input
        (ops.bound_of); the table then carries the bound of the normalised values for the f16x2 dense form."""
        mod = self.module
        use_batch = self.training or mod.running_mean is None
        if mod.momentum is None:
            raise NotImplementedError("cumulative-moving-average BatchNorm (momentum=None) is not supported")
        d = lambda t: None if t is None else t.detach()
        update = self.training and mod.track_running_stats and not AG.is_reexecution()   # (re-execution for backward)
        return ops.batchnorm_finalize(stats if use_batch else None, m, self.in_channels, d(mod.weight), d(mod.bias),
                                      mod.running_mean if (update or not use_batch) else None,
                                      mod.running_var if (update or not use_batch) else None,
                                      mod.num_batches_tracked if update else None, use_batch, mod.momentum, mod.eps,
                                      in_bound=in_bound)

    def uses_frame_scope(self) -> bool:
        """Inside ``frame_scope`` with batch statistics in use: normalise every frame with its own statistics."""
        return current_frame_scope() is not None and (self.training or self.module.running_mean is None)

    def scale_shift_frames(self, frame_stats, in_bound=None) -> torch.Tensor:
        """[F, ops.AFFINE_ROWS, C] apply table of this BatchNorm with per-frame statistics, from the column statistics a conv layer's dense
        launches left per 128-row panel of the frame-padded row lists (``frame_stats``: FrameStats of MPNNConv); running statistics
        are updated frame after frame.  The NEXT layer's dense launches apply the table (ops.linear a1_affine_tiles)."""
        mod = self.module
        if mod.momentum is None:
            raise NotImplementedError("cumulative-moving-average BatchNorm (momentum=None) is not supported")
        d = lambda t: None if t is None else t.detach()
        update = self.training and mod.track_running_stats and not AG.is_reexecution()
        pad = current_frame_scope().padded_split()
        return ops.batchnorm_segments_from_panels(frame_stats.main, pad["ne"][3], frame_stats.iso, pad["e"][3], current_frame_scope().node_ptr,
                                                  d(mod.weight), d(mod.bias), mod.running_mean if update else None,
                                                  mod.running_var if update else None, mod.num_batches_tracked if update else None,
                                                  mod.momentum, mod.eps, in_bound=in_bound)

    def apply_frames(self, x: torch.Tensor, relu: bool) -> torch.Tensor:
        """act(BatchNorm(x)) with per-frame statistics (ops.batchnorm_segments); running statistics are updated frame after
        frame, as a loop of single-frame forwards would."""
        if AG.is_recording():
            raise NotImplementedError("per-frame BatchNorm statistics (frame_scope) are an inference feature: no backward pass")
        if x.shape[0] == 0 and self.empty_or_single(0, x.shape[1]):
            return torch.relu(x) if relu else x
        mod = self.module
        if mod.momentum is None:
            raise NotImplementedError("cumulative-moving-average BatchNorm (momentum=None) is not supported")
        seg = current_frame_scope().seg_ptr_for(x.shape[0])
        d = lambda t: None if t is None else t.detach()
        update = self.training and mod.track_running_stats and not AG.is_reexecution()
        if os.environ.get("RGNN_BN_SEG_SPLIT") is not None:     # (the three-launch form: statistics, finish, apply)
            table = ops.batchnorm_segments(x, seg, d(mod.weight), d(mod.bias), mod.running_mean if update else None,
                                           mod.running_var if update else None, mod.num_batches_tracked if update else None,
                                           mod.momentum, mod.eps)
            return ops.scale_shift_act_segments(x, table, seg, relu)
        return ops.batchnorm_act_segments(x, seg, d(mod.weight), d(mod.bias), mod.running_mean if update else None,
                                          mod.running_var if update else None, mod.num_batches_tracked if update else None,
                                          mod.momentum, mod.eps, relu)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.uses_frame_scope():
            return self.apply_frames(x, relu=False)
        if self.empty_or_single(x.shape[0], x.shape[1]):
            return x
        if AG.grad_mode(x, self.module.weight, self.module.bias):
            return AG.batch_norm_act(x, self, relu=False)
        use_batch = self.training or self.module.running_mean is None
        stats = ops.column_stats(x) if use_batch else None
        return ops.scale_shift_act(x, self.scale_shift(stats, x.shape[0]), relu=False)

    def __repr__(self) -> str:
        return f"{self.__class__.__name__}({self.in_channels})"


def run_mlp(seq: nn.Sequential, x: torch.Tensor, *, a2: Optional[torch.Tensor] = None,
            residual: Optional[torch.Tensor] = None, want_stats: bool = False
            ) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """Execute Sequential[Linear, (BatchNorm), ReLU, ...] with fused epilogues.

    ``a2``: second input block of the FIRST Linear ([x | a2] without materialising the concatenation).
    ``residual`` / ``want_stats`` apply to the LAST Linear (h + x of RadarPointGNNConv; statistics for the
    BatchNorm that DetNetBasic applies after every conv)."""
    mods = list(seq)
    stats = None
    i = 0
    last_linear = max((j for j, m_ in enumerate(mods) if isinstance(m_, Linear)), default=-1)
    if AG.is_recording():
        return _run_mlp_grad(mods, x, a2, residual, want_stats, last_linear)
    while i < len(mods):
        m_ = mods[i]
        if isinstance(m_, Linear):
            nxt = mods[i + 1] if i + 1 < len(mods) else None
            is_last = i == last_linear
            fuse_relu = isinstance(nxt, nn.ReLU)
            fuse_bn = isinstance(nxt, BatchNorm)
            w = m_.weight.detach()
            b = None if m_.bias is None else m_.bias.detach()
            res = residual if is_last else None
            per_frame = fuse_bn and nxt.uses_frame_scope()
            need_stats = (fuse_bn and not per_frame) or (is_last and want_stats)
            out = ops.linear(x, w, b, a2=a2, relu=fuse_relu and res is None, residual=res, want_stats=need_stats)
            a2 = None
            if need_stats:
                x, st = out
                if is_last and want_stats and not fuse_bn:
                    stats = st
            else:
                x = out
            i += 1
            if fuse_relu and res is None:
                i += 1
            elif fuse_bn:
                relu_after = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
                if not per_frame and nxt.empty_or_single(x.shape[0], x.shape[1]):
                    pass                                   # (no rows: nothing to normalise -- torch returns the empty matrix)
                elif per_frame:
                    x = nxt.apply_frames(x, relu_after)
                else:
                    ss = nxt.scale_shift(st, x.shape[0], in_bound=ops.bound_of(x))
                    x = ops.scale_shift_act(x, ss, relu=relu_after)
                i += 2 if relu_after else 1
        elif isinstance(m_, nn.ReLU):
            x = torch.relu(x)          # only reached for hand-built Sequentials that do not start with a Linear
            i += 1
        else:
            x = m_(x)
            i += 1
    return x, stats


def _run_mlp_grad(mods, x, a2, residual, want_stats, last_linear):
    """``run_mlp`` with autograd recording (same fusion decisions; radargnn_amd/gnn/autograd.py)."""
    stats = None
    i = 0
    while i < len(mods):
        m_ = mods[i]
        if isinstance(m_, Linear):
            nxt = mods[i + 1] if i + 1 < len(mods) else None
            is_last = i == last_linear
            fuse_relu = isinstance(nxt, nn.ReLU)
            fuse_bn = isinstance(nxt, BatchNorm)
            res = residual if is_last else None
            need_stats = (fuse_bn and (nxt.training or nxt.module.running_mean is None)) or (is_last and want_stats)
            out = AG.linear(x, m_.weight, m_.bias, a2=a2, relu=fuse_relu and res is None, want_stats=need_stats,
                            residual=res)
            a2 = None
            st = None
            if need_stats:
                x, st = out
                if is_last and want_stats and not fuse_bn:
                    stats = st
            else:
                x = out
            i += 1
            if fuse_relu and res is None:
                i += 1
            elif fuse_bn:
                relu_after = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
                x = AG.batch_norm_act(x, nxt, stats=st, relu=relu_after)
                i += 2 if relu_after else 1
        elif isinstance(m_, nn.ReLU):
            x = torch.relu(x)
            i += 1
        else:
            x = m_(x)
            i += 1
    return x, stats
