"""``DetNetBasic`` and ``get_mlp`` -- mirrors of gnn/gnn_models.py:15-134 and :137-178 of the reference: node / edge
embedding MLPs, L x [conv -> BatchNorm -> ReLU], a classification head and a box-regression head.  Same
attribute names and ``state_dict`` keys, so reference checkpoints load with ``load_state_dict``.

Forward pass on the MI355X (one CSR-by-target build per call, shared by all layers):

    x  = node_emb_mlp(x)                                  MFMA GEMMs, ReLU in the epilogue
    ea = edge_emb_mlp(edge_attr[perm])                    edge attributes are re-ordered ONCE into target order
    per layer:  [P|Q] = x [W_i;W_j]^T (+b)                one MFMA launch
                m      = P (.) aggr_e(Q[src] + W_e ea)    fused gather / mat-vec / segmented reduce
                h      = [x|m] W_post^T + b               MFMA, column statistics for BatchNorm in the epilogue
                x      = relu(h * scale + shift)          train-mode BatchNorm (batch statistics) + ReLU
    c, bb = classification_head(x), regression_head(x)
"""
from __future__ import annotations

from typing import List

import torch
from torch import nn
from torch.nn import ModuleList, ReLU, Sequential

from .. import ops
from .configs import GNNArchitectureConfig
from . import autograd as AG
from .linear import BatchNorm, Linear, frame_scope, run_mlp
from .mpnn_layers import DeferredEdgeAttr, MPNNConv, RadarPointGNNConv, TargetCSR, UnsortedEdgeAttr, _cache_key, _same_key, _state_without_caches
from . import linear as _lin_mod


class _rows_of:
    """Inside an active ``frame_scope``: the BatchNorm inputs of the block are node / edge matrices (frame_scope.rows_of)."""

    def __init__(self, kind: str):
        self.kind, self.ctx = kind, None

    def __enter__(self):
        if _lin_mod.current_frame_scope() is not None:
            self.ctx = _lin_mod.current_frame_scope().rows_of(self.kind)
            self.ctx.__enter__()

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
        return False

# per-frame BatchNorm statistics (frame_scope) from the conv layers' own epilogues on frame-padded row lists, applied by the next
# layer's dense launches, instead of a statistics + apply pass over [N, C] per layer
FUSE_FRAME_BN = __import__("os").environ.get("RGNN_NO_FUSED_FRAME_BN") is None
FUSE_EMBED3 = __import__("os").environ.get("RGNN_NO_EMBED3") is None        # node embedding 5 -> 32 -> 64 -> 128 in one launch
FUSE_HEADS = __import__("os").environ.get("RGNN_NO_FUSED_HEADS") is None   # first Linears of both heads in one launch (inference)


def get_mlp(in_size: int, out_size: int, hidden_layer_sizes: List[int], batch_norm: bool) -> Sequential:
    """Linear(in, h0), then for every further width [BatchNorm], ReLU, Linear, and finally [BatchNorm], ReLU,
    Linear(h_last, out); no activation after the last Linear; ``hidden_layer_sizes == []`` gives one Linear
    (reference: gnn/gnn_models.py:137-178)."""
    widths = list(hidden_layer_sizes)
    if not widths:
        return Sequential(Linear(in_size, out_size))
    mods: List[nn.Module] = [Linear(in_size, widths[0])]
    for prev, width in zip(widths, widths[1:] + [out_size]):
        if batch_norm:
            mods.append(BatchNorm(prev))
        mods.append(ReLU())
        mods.append(Linear(prev, width))
    return Sequential(*mods)


class DetNetBasic(nn.Module):
    """GNN for per-point semantic segmentation + bounding-box regression on radar point clouds."""

    def __init__(self, config: GNNArchitectureConfig):
        super().__init__()
        self.batch_norm_mlps = config.batch_norm_in_mlps
        self.node_feat_dim = config.node_feature_dimension
        self.edge_feat_dim = config.edge_feature_dimension
        self.conv_layer_dimensions = config.conv_layer_dimensions
        self.initial_node_feature_embedding = config.initial_node_feature_embedding
        self.initial_edge_feature_embedding = config.initial_edge_feature_embedding
        self.conv_pre_mlp_layers = config.conv_pre_mlp_layer_number
        self.conv_post_mlp_layers = config.conv_post_mlp_layer_number
        self.conv_use_edge_encoder = config.conv_use_edge_encoder
        self.aggregation = config.aggregation_function

        if config.initial_node_feature_embedding:               # gnn_models.py:42-46
            dims = config.node_feature_embedding_layer_dimensions
            self.node_emb_mlp = get_mlp(self.node_feat_dim, dims[-1], dims[:-1], self.batch_norm_mlps)
            self.node_feat_dim = dims[-1]
        if config.initial_edge_feature_embedding:               # gnn_models.py:48-52
            dims = config.edge_feature_embedding_layer_dimensions
            self.edge_emb_mlp = get_mlp(self.edge_feat_dim, dims[-1], dims[:-1], self.batch_norm_mlps)
            self.edge_feat_dim = dims[-1]

        if config.conv_layer_type not in ("MPNNConv", "RadarPointGNNConv"):
            raise Exception(f"{config.conv_layer_type} is invalid GNN conv layer type. "
                            "Chose either MPNNConv or RadarPointGNNConv")
        self.convs = ModuleList()
        self.batch_norms = ModuleList()
        width_in = self.node_feat_dim
        for i, width_out in enumerate(self.conv_layer_dimensions):
            if config.conv_layer_type == "MPNNConv":
                conv = MPNNConv(width_in, width_out, self.edge_feat_dim, aggr=self.aggregation,
                                pre_layers=self.conv_pre_mlp_layers, post_layers=self.conv_post_mlp_layers,
                                use_edge_encoder=self.conv_use_edge_encoder)
                bn_width = width_out
            else:
                # this layer cannot change the width: every conv works on the embedded node width, and only
                # the FIRST BatchNorm is sized by it (the reference sizes the others from the config list,
                # gnn_models.py:62-66,81-88)
                conv = RadarPointGNNConv(self.node_feat_dim, self.edge_feat_dim, aggr=self.aggregation,
                                         pre_layers=self.conv_pre_mlp_layers, post_layers=self.conv_post_mlp_layers)
                bn_width = self.node_feat_dim if i == 0 else width_out
            self.convs.append(conv)
            self.batch_norms.append(BatchNorm(bn_width))
            width_in = width_out

        final_dim = self.conv_layer_dimensions[-1]              # gnn_models.py:92-102
        dims = config.classification_head_layer_dimensions
        self.classification_head = get_mlp(final_dim, dims[-1], dims[:-1], self.batch_norm_mlps)
        dims = config.regression_head_layer_dimensions
        self.regression_head = get_mlp(final_dim, dims[-1], dims[:-1], self.batch_norm_mlps)

    def __getstate__(self):
        # (whole-module pickles -- gnn/trainer.py:342-354, evaluate.py:46-52 -- and deepcopy carry parameters and buffers, not the
        #  weight-derived caches an inference pass leaves on the instance)
        return _state_without_caches(self)

    def forward(self, x: torch.Tensor, edge_index: torch.Tensor, edge_attr: torch.Tensor, frame_ptr: torch.Tensor = None):
        """-> (class logits [N, K], boxes [N, 4|5]); reference: gnn/gnn_models.py:104-134.
        ``frame_ptr`` (extension; int64 [F + 1], PyG ``Batch.ptr``): the batch holds F graphs laid back to back and every
        train-mode BatchNorm takes its statistics per graph -- the result of F single-graph forwards (how the reference runs
        inference, evaluate.py:40) from one batched call.  Inference only."""
        graph = TargetCSR(edge_index, x.shape[0])
        if frame_ptr is not None:
            if AG.is_recording() or (torch.is_grad_enabled() and (x.requires_grad or edge_attr.requires_grad)):
                # (silently detached outputs with updated running statistics are worse than an error)
                raise NotImplementedError("per-frame BatchNorm statistics (frame_ptr) are an inference feature: call under "
                                          "torch.no_grad() / with inputs that do not require gradients")
            with frame_scope(frame_ptr.to(x.device), x.shape[0], graph), torch.no_grad():
                return self.forward_graph(x.detach(), graph, graph.sort_edge_attr(edge_attr.detach()))

        def run(x_, ea_):
            return self.forward_graph(x_, graph, graph.sort_edge_attr(ea_))

        params = list(self.parameters())
        if not AG.is_recording() and AG.grad_mode(x, edge_attr, *params):
            if x.requires_grad or edge_attr.requires_grad:
                # a training step: the reference's trainer marks the inputs (gnn/trainer.py:179-180 x.requires_grad_(),
                # edge_attr.requires_grad_()) before the forward.  Record the autograd nodes directly -- for the shipped
                # layer shapes they launch the inference kernels -- so nothing runs twice.
                with AG.recording(direct=True):
                    return run(x, edge_attr)
            # only the parameters require gradients: how the reference runs inference (postprocessor/inference.py:57-62,
            # autograd left on, backward never called).  Inference kernels now, differentiable re-execution only if
            # backward is ever called (autograd.py) -- a training loop that does not mark its inputs still works, it pays
            # a second forward.
            return AG.checkpointed(run, (x, edge_attr), params)
        return run(x, edge_attr)

    def forward_graph(self, x: torch.Tensor, graph: TargetCSR, edge_attr_sorted: torch.Tensor):
        """Same as ``forward`` for callers that already hold the target-sorted graph (radargnn_amd.frames)."""
        # inference: the kernels track a bound of every activation they produce, which lets the dense layers run in the f16x2 form
        # (three matrix-pipe products per fp32 product instead of six; ops.bound_tracking).  A recorded (training) forward can do the
        # same (ops.TRAIN_F16X2) -- its autograd nodes keep the pool and their backward launches go on tracking in it
        # (gnn/autograd.py).  The weights change every step and the backward pass multiplies by one-off transposed copies, so the
        # f16 planes are rebuilt for every launch: that pays since their build is two short launches (MEASUREMENTS.md section 8)
        if not AG.is_recording() or ops.TRAIN_F16X2:
            with ops.bound_tracking(x.device):
                return self._forward_graph(x, graph, edge_attr_sorted)
        return self._forward_graph(x, graph, edge_attr_sorted)

    def _forward_graph(self, x: torch.Tensor, graph: TargetCSR, edge_attr_sorted: torch.Tensor):
        node_tail = None
        if self.initial_node_feature_embedding:
            mods = list(self.node_emb_mlp)
            first_conv = self.convs[0] if len(self.convs) else None
            if (len(mods) > 1 and isinstance(mods[-1], Linear) and isinstance(first_conv, MPNNConv)
                    and first_conv.can_fold_input_tail(x) and first_conv.in_channels == mods[-1].out_channels):
                # the embedding's last Linear has no activation behind it (gnn_models.py:137-178) and the first conv reads its
                # input only through linear maps: the Linear is folded into that layer's weights (MPNNConv._input_tail_weights),
                # the [N, C] embedding output is never computed
                with _rows_of("node"):
                    x = self._embedding_front(mods[:-1], x)
                node_tail = (mods[-1].weight.detach(), None if mods[-1].bias is None else mods[-1].bias.detach())
            else:
                with _rows_of("node"):
                    x, _ = run_mlp(self.node_emb_mlp, x)
        ea = edge_attr_sorted
        emb = list(self.edge_emb_mlp) if self.initial_edge_feature_embedding else []
        if not (isinstance(ea, UnsortedEdgeAttr) and emb and isinstance(emb[-1], Linear) and self._tiny_edge_hidden(emb[:-1])):
            graph.join_csr()                        # (everything but the deferred tiny embedding below reads the edges from here on)
        lazy = isinstance(ea, UnsortedEdgeAttr)     # edge attributes still in edge order (frames.HotPath): re-ordered by whoever reads them first
        edge_tail = None
        if self.initial_edge_feature_embedding:
            # every conv consumes the embedded edge attributes through a Linear map, so the embedding's last Linear
            # (no activation follows it, gnn_models.py:137-178) is folded into the convs' W_e: the edge stage reads
            # the 8-wide hidden activations instead of the 16-wide embedding and does half the multiply-adds
            mods = list(self.edge_emb_mlp)
            last = mods[-1]
            if isinstance(last, Linear):
                hidden = mods[:-1]
                if lazy and self._tiny_edge_hidden(hidden):
                    # the shipped shape (2 -> 4 -> 8, ReLU after each): gather + both layers in one pass over the edges
                    l1, l2 = hidden[0], hidden[2]
                    d = lambda t: None if t is None else t.detach()
                    raw = ea.raw

                    def embed():
                        graph.join_csr()
                        if graph.own_edge is not None:
                            # (antisymmetric attributes, CSR built without the twin search: the in-edge's attributes are minus the
                            #  own edge's, and relu(W (-a) + b) = relu((-W) a + b))
                            return ops.tiny_mlp2(raw, graph.own_edge, self._negated(l1.weight), d(l1.bias), True, d(l2.weight),
                                                 d(l2.bias), True)
                        return ops.tiny_mlp2(raw, graph.perm, d(l1.weight), d(l1.bias), True, d(l2.weight), d(l2.bias), True)
                    # (the edge side of a captured step is still under way on its branch: the embedding waits until the first conv
                    #  layer's edge stage asks for it -- behind that layer's node-only launches)
                    defer = (getattr(graph, "_csr_pending", None) is not None and len(self.convs) > 0
                             and all(isinstance(c, MPNNConv) for c in self.convs))
                    ea = DeferredEdgeAttr(embed) if defer else embed()
                    lazy = False
                else:
                    if lazy:
                        ea, lazy = ea.materialize(), False
                    if hidden:
                        with _rows_of("edge"):
                            ea, _ = run_mlp(hidden, ea)
                if AG.is_recording():
                    edge_tail = (last.weight, last.bias)      # stays on the autograd tape (folded with torch matmuls)
                else:
                    edge_tail = (last.weight.detach(), None if last.bias is None else last.bias.detach())
            else:
                if lazy:
                    ea, lazy = ea.materialize(), False
                with _rows_of("edge"):
                    ea, _ = run_mlp(mods, ea)
        if lazy:
            ea = ea.materialize()
        pending = None      # [AFFINE_ROWS, C] apply table of a BatchNorm + ReLU that the NEXT conv applies to its input (inference form)
        frames = None       # frame-padded row lists: per-frame statistics without a pass over [N, C] (frame_scope.padded_split)
        from . import linear as _lin
        if (not AG.is_recording() and FUSE_FRAME_BN and len(self.convs) and all(bn.uses_frame_scope() for bn in self.batch_norms)
                and all(isinstance(c, MPNNConv) for c in self.convs) and _lin.current_frame_scope().graph is graph
                and self._frames_fusable(x, graph, node_tail)):
            frames = _lin.current_frame_scope().padded_split()
        for conv, bn in zip(self.convs, self.batch_norms):
            use_batch = bn.training or bn.module.running_mean is None
            if AG.is_recording():
                h, stats = conv.forward_sorted(x, graph, ea, want_stats=use_batch, edge_tail=edge_tail)
                x = AG.batch_norm_act(h, bn, stats=stats, relu=True)
            elif frames is not None:
                # per-frame statistics from the launches' own epilogues, applied by the next layer's launches: [F, AFFINE_ROWS, C] tables
                h, fstats = conv.forward_sorted(x, graph, ea, want_stats=True, edge_tail=edge_tail, x_affine=pending, frames=frames,
                                                **({"x_tail": node_tail} if node_tail is not None else {}))
                node_tail = None
                x, pending = h, bn.scale_shift_frames(fstats, in_bound=ops.bound_of(h))
            elif bn.uses_frame_scope():
                # per-frame statistics (frame_scope): the normalised activations are materialised by the segmented apply pass
                h, _ = conv.forward_sorted(x, graph, ea, want_stats=False, edge_tail=edge_tail, x_affine=pending,
                                           **({"x_tail": node_tail} if node_tail is not None else {}))
                node_tail = None
                with _rows_of("node"):
                    x, pending = bn.apply_frames(h, relu=True), None
            else:
                # batch_norm + F.relu (:126-128): the scale / shift come out of the statistics the conv's GEMMs left behind;
                # applying them is left to the dense kernels of the next conv (their A-operand path), which deletes a
                # read + write pass over [N, C] per layer.  The last conv's output is materialised for the heads.
                h, stats = conv.forward_sorted(x, graph, ea, want_stats=use_batch, edge_tail=edge_tail, x_affine=pending,
                                               **({"x_tail": node_tail} if node_tail is not None else {}))
                node_tail = None
                x, pending = h, bn.scale_shift(stats, h.shape[0], in_bound=ops.bound_of(h))
        if pending is not None and frames is not None:     # the last BatchNorm's per-frame tables: one pass for the heads
            x, pending = ops.scale_shift_act_segments(x, pending, _lin.current_frame_scope().node_ptr, True), None
        if pending is not None:
            fused = self._fused_heads(x, pending) if FUSE_HEADS else None
            if fused is not None:
                return fused
            x = ops.scale_shift_act(x, pending, relu=True)
        with _rows_of("node"):
            c, _ = run_mlp(self.classification_head, x)
            bb, _ = run_mlp(self.regression_head, x)
        return c, bb

    def _frames_fusable(self, x, graph, node_tail) -> bool:
        """Every conv layer qualifies for frame-padded row lists (MPNNConv.frames_fusable, at its own input width) and every frame
        is large enough that padding it to whole tiles costs little (>= 256 nodes on average)."""
        from . import linear as _lin
        f = _lin.current_frame_scope().node_ptr.numel() - 1
        if f < 1 or x.shape[0] < 256 * f:
            return False
        # (behind a folded node-embedding tail the first layer reads x as it is: the embedding's narrower hidden layer)
        return all(conv.frames_fusable(x, graph, k1=x.shape[1] if (i == 0 and node_tail is not None) else None)
                   for i, conv in enumerate(self.convs))

    @staticmethod
    def _embedding_front(mods, x: torch.Tensor) -> torch.Tensor:
        """The node embedding without its (folded) last Linear: [Linear, ReLU] x 3 of the shipped widths in ONE launch
        (ops.embed3: the [N, 32] and [N, 64] intermediates never reach HBM), anything else layer by layer."""
        if (FUSE_EMBED3 and not AG.is_recording() and len(mods) == 6 and all(isinstance(mods[i], Linear) for i in (0, 2, 4))
                and all(isinstance(mods[i], ReLU) for i in (1, 3, 5)) and x.shape[0] > 0):
            d = lambda t: None if t is None else t.detach()
            l1, l2, l3 = mods[0], mods[2], mods[4]
            out = ops.embed3(x, d(l1.weight), d(l1.bias), d(l2.weight), d(l2.bias), d(l3.weight), d(l3.bias), True)
            if out is not None:
                return out
        return run_mlp(mods, x)[0]

    def _negated(self, w: torch.Tensor) -> torch.Tensor:
        """-w, cached until w changes (one elementwise launch per weight version, outside captured steps)."""
        key = (w.data_ptr(), w._version, ops.CACHE_EPOCH)
        cache = self.__dict__.get("_neg_cache")
        if cache is None or cache[0] != key:
            # (kept in the instance dict, NOT through nn.Module.__setattr__: assigning the Parameter `w` as an attribute registered it
            #  as a second parameter "_neg_keep" -- an extra key in state_dict() after the first inference pass, and a strict
            #  load_state_dict of the original checkpoint failed)
            cache = self.__dict__["_neg_cache"] = (key, (-w.detach()).contiguous(), w)
        return cache[1]

    @staticmethod
    def _tiny_edge_hidden(hidden) -> bool:
        """[Linear, ReLU, Linear, ReLU] with at most 8 inputs, 8 hidden and 16 output features: what ops.tiny_mlp2 computes."""
        return (not AG.is_recording() and len(hidden) == 4 and isinstance(hidden[0], Linear) and isinstance(hidden[1], ReLU)
                and isinstance(hidden[2], Linear) and isinstance(hidden[3], ReLU) and hidden[0].in_channels <= 8
                and hidden[0].out_channels <= 8 and hidden[2].out_channels <= 16
                and __import__("os").environ.get("RGNN_NO_TINY_MLP2") is None)

    def _fused_heads(self, h: torch.Tensor, scale_shift: torch.Tensor):
        """Inference form of the two heads (:131-132) behind the last BatchNorm + ReLU: their first Linears read the same node
        features, so ONE launch computes both from one pass over ``h`` -- concatenated weights, the BatchNorm's scale / shift
        applied to the A operand inside the kernel, ReLU only on the columns of the head whose first Linear is followed by
        one -- instead of a normalisation pass and two launches that each read its result.  None when the heads do not have
        that shape (a BatchNorm behind a first Linear, no ReLU pattern one column boundary can express)."""
        heads = []
        for seq in (self.classification_head, self.regression_head):
            mods = list(seq)
            if not mods or not isinstance(mods[0], Linear) or (len(mods) > 1 and not isinstance(mods[1], (ReLU, Linear))):
                return None
            relu = len(mods) > 1 and isinstance(mods[1], ReLU)
            heads.append((mods[0], relu, mods[2:] if relu else mods[1:]))
        order = (0, 1) if heads[0][1] <= heads[1][1] else (1, 0)            # the head WITHOUT a ReLU first
        first, second = heads[order[0]], heads[order[1]]
        lins = (first[0], second[0])
        if any(l.bias is None for l in lins):
            return None
        key = _cache_key((lins[0].weight, lins[0].bias, lins[1].weight, lins[1].bias))
        n0, n1 = lins[0].weight.shape[0], lins[1].weight.shape[0]
        pad = (-n0) % 4                  # zero columns between the heads: the second head's block starts 16-byte aligned
        if not _same_key(getattr(self, "_heads_key", None), key):
            w0, b0 = lins[0].weight.detach(), lins[0].bias.detach()
            self._heads_val = (torch.cat([w0, w0.new_zeros((pad, w0.shape[1])), lins[1].weight.detach()], 0).contiguous(),
                               torch.cat([b0, b0.new_zeros(pad), lins[1].bias.detach()], 0).contiguous())
            self._heads_key = key
        w, b = self._heads_val
        relu_any = first[1] or second[1]
        out = ops.linear(h, w, b, relu=relu_any, relu_from=0 if first[1] else n0 + pad, a1_affine=scale_shift, a1_relu=True)
        res = [None, None]
        for which, (_, _, rest), view in ((order[0], first, out[:, :n0]), (order[1], second, out[:, n0 + pad:n0 + pad + n1])):
            res[which] = run_mlp(rest, view)[0] if rest else view
        return res[0], res[1]
