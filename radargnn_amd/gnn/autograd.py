"""Differentiable forms of the fused HIP operators, so ``loss.backward()`` works through ``DetNetBasic`` / ``MPNNConv`` /
``RadarPointGNNConv`` the way the reference's trainer expects (gnn/trainer.py:176-231: ``requires_grad_``, forward,
``loss.backward()``, ``optimizer.step()``) -- SURVEY.md section 8(f) row 1.

Each ``torch.autograd.Function`` runs the same forward kernel as inference and a hand-written backward:

* ``LinearFn``      out = act([a1|a2] W^T + b): dA = g W on rgnn_linear_fwd (transposed weight, a small device-side copy),
                    [dW | db] = g^T [a1|a2|1] on rgnn_wgrad (bf16x3 MFMA), g = relu'(out) dy (rgnn_relu_bwd);
* ``ConvFoldedFn``  a whole MPNNConv layer in its folded inference form (row-split update, source term on the rows with
                    edges, fused edge kernel) with a hand-scheduled backward over the same row lists;
* ``BatchNormActFn`` train-mode BatchNorm1d (+ReLU) from the column statistics of the GEMM epilogue
                    (rgnn_bn_bwd_stats / rgnn_bn_bwd_apply; the [C]-sized coefficient algebra in float64 torch ops);
* ``AggregateFn``   M[t] = aggr_e(Q[s_e] + W_e a_e): rgnn_mpnn_aggregate / rgnn_mpnn_aggregate_bwd.

The weight folds of the inference path (edge encoder, edge-embedding tail) are differentiable products of [D, De]-sized
matrices here (``matmul`` below: the same HIP kernels, no BLAS), so their parameters receive gradients through autograd; the target-term fold is not used
when gradients are required (P is produced by the GEMM and added to the aggregate).
"""
from __future__ import annotations

from typing import Optional

import torch

from .. import ops


def grad_mode(*tensors) -> bool:
    """True when autograd must record: grad enabled and some tensor / parameter requires it."""
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


# ---- whole-forward checkpointing ------------------------------------------------------------------------------------
# The reference runs inference with autograd enabled and simply never calls backward (postprocessor/inference.py:57-62),
# so "gradients could be required" must not cost anything: the public forward passes (DetNetBasic.forward,
# MPNNConv.forward, ...) always execute the fused inference kernels and return outputs attached to ONE autograd node
# (``_Checkpointed``).  Only if backward actually reaches that node is the forward re-executed in its differentiable
# form (``is_recording()`` is true inside that re-execution: run_mlp / forward_sorted / forward_graph then pick the
# autograd Functions below) and differentiated.  Training pays one extra forward; inference pays nothing.
#
# A forward whose INPUTS require gradients -- the reference's trainer marks them, gnn/trainer.py:179-180 -- records
# directly instead: it runs the differentiable Functions once (``recording(direct=True)``); for the layer shapes the
# reference ships these launch the same fused kernels as inference (ConvFoldedFn), and nothing is executed twice.
_RECORDING = False
_REEXECUTION = False


def is_recording() -> bool:
    return _RECORDING


def is_reexecution() -> bool:
    """True inside the backward-time re-execution of a checkpointed forward (side effects such as BatchNorm's running
    statistics already happened in the first execution)."""
    return _REEXECUTION


class recording:
    def __init__(self, direct: bool = False):
        self.direct = direct

    def __enter__(self):
        global _RECORDING, _REEXECUTION
        self.prev = (_RECORDING, _REEXECUTION)
        _RECORDING, _REEXECUTION = True, not self.direct

    def __exit__(self, *exc):
        global _RECORDING, _REEXECUTION
        _RECORDING, _REEXECUTION = self.prev


_Recording = recording


class _Checkpointed(torch.autograd.Function):
    @staticmethod
    def forward(ctx, fn, n_inputs, *tensors):
        ctx.fn, ctx.n_inputs = fn, n_inputs
        ctx.save_for_backward(*tensors)
        with torch.no_grad():
            outs = fn(*tensors[:n_inputs])
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        tensors = ctx.saved_tensors
        n = ctx.n_inputs
        inputs = [t.detach().requires_grad_(t.requires_grad) for t in tensors[:n]]
        params = list(tensors[n:])                               # the live Parameters (leaves)
        with torch.enable_grad(), _Recording():
            outs = ctx.fn(*inputs)
        pairs = [(o, g) for o, g in zip(outs, grads) if g is not None and o.requires_grad]
        wanted = [t for t in inputs + params if t.requires_grad]
        got = torch.autograd.grad([o for o, _ in pairs], wanted, [g for _, g in pairs], allow_unused=True) if pairs and wanted else ()
        it = iter(got)
        res = [next(it) if t.requires_grad else None for t in inputs + params]
        return (None, None, *res)


def checkpointed(fn, inputs, params):
    """outs = fn(*inputs) on the inference kernels now; differentiable re-execution if backward is called."""
    return _Checkpointed.apply(fn, len(inputs), *inputs, *params)


class LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a1, a2, weight, bias, relu: bool, want_stats: bool, residual):
        # (a weight that is itself computed -- a fold of parameters -- is a new tensor every step: its planes are not worth a cache entry)
        out = ops.linear(a1, weight.contiguous(), None if bias is None else bias.contiguous(), a2=a2, relu=relu,
                         residual=residual, want_stats=want_stats, cache_planes=weight.is_leaf)
        stats = None
        if want_stats:
            out, stats = out
            ctx.mark_non_differentiable(stats)
        ctx.relu = relu
        ctx.pool = ops.ctx().bounds                              # (backward runs on autograd's thread: ops.using_bounds)
        ctx.has_a2 = a2 is not None
        ctx.has_bias = bias is not None
        ctx.has_res = residual is not None
        ctx.save_for_backward(a1, a2, weight, out if relu else None)
        return out, stats

    @staticmethod
    def backward(ctx, dy, _dstats):
        with ops.using_bounds(ctx.pool):
            return LinearFn._backward(ctx, dy)

    @staticmethod
    def _backward(ctx, dy):
        a1, a2, weight, out = ctx.saved_tensors
        g = dy.contiguous()
        if ctx.relu:
            g = ops.relu_bwd(g, out)
        k1 = a1.shape[1]
        needs = ctx.needs_input_grad
        da1 = da2 = dw = db = dres = None
        wt = None
        if needs[0] or (ctx.has_a2 and needs[1]):
            wt = weight.t().contiguous()                         # [K, N]: dA = g @ W = linear(g, W^T)
        if needs[0]:
            da1 = ops.linear(g, wt[:k1], cache_planes=False)           # one-off transposed weight: do not cache its planes
        if ctx.has_a2 and needs[1]:
            da2 = ops.linear(g, wt[k1:], cache_planes=False)
        want_db = ctx.has_bias and needs[3]
        if needs[2] or want_db:
            # dW = g^T [a1 | a2] and db = column sums of g in ONE launch (the bias gradient is the product with a column of
            # ones): rgnn_wgrad, bf16x3 MFMA, any widths
            dwb = ops.linear_wgrad(g, a1, a2 if ctx.has_a2 else None, with_bias=want_db)
            if want_db:
                dw, db = dwb[:, :-1], dwb[:, -1]
            else:
                dw = dwb
            if not needs[2]:
                dw = None
        if ctx.has_res and needs[6]:
            dres = dy
        return da1, da2, dw, db, None, None, dres


def linear(a1, weight, bias=None, *, a2=None, relu=False, want_stats=False, residual=None):
    out, stats = LinearFn.apply(a1, a2, weight, bias, relu, want_stats, residual)
    return (out, stats) if want_stats else out


def matmul(a, b):
    """a @ b for small parameter-sized matrices on the HIP kernels, differentiable in both (the weight folds of the
    training path; torch.matmul would go to the BLAS)."""
    return LinearFn.apply(a, None, b.t(), None, False, False, None)[0]


class ConvFoldedFn(torch.autograd.Function):
    """One MPNNConv layer (single-Linear message and update MLPs, max / mean) in the folded form the inference path runs:

        Q  = x W_j^T                                   on the rows that have outgoing edges (TargetCSR.source_rows)
        M  = 1[deg>0] (p_bias + aggr_e(Q[s_e] + W_e a_e))                          fused edge kernel
        h  = [x | M] W_comb^T + b_comb   on targets with incoming edges,   h = x W_px^T + b_post   on isolated targets

    ``W_comb = [W_px + W_pm W_i | W_pm]``, ``b_comb``, ``W_e``, ``p_bias`` arrive as differentiable functions of the
    parameters (small torch-visible products), so autograd distributes their gradients; this Function owns everything that
    touches [N, .] or [E, .] data.  Backward, over the same row lists: dM and dx by rgnn_linear_fwd on transposed weights
    (dx of a row = one GEMM over [dh | dQ]), the edge stage by rgnn_mpnn_aggregate_bwd, the four weight gradients (bias
    gradients as columns of ones) by rgnn_wgrad."""

    @staticmethod
    def forward(ctx, x, ea_sorted, Wj, We, p_bias, Wcomb, bcomb, Wpx, bp, graph, aggr: str, want_stats: bool):
        x = x.contiguous()
        n, c = x.shape
        co = Wcomb.shape[0]
        lst_e, cnt_e, _, lst_ne, cnt_ne = graph.split_targets()
        Wj_c, We_c = Wj.contiguous(), We.contiguous()
        Wcomb_c, Wpx_c = Wcomb.contiguous(), Wpx.contiguous()
        src_rows = graph.source_rows()                                           # rows of Q that the edge stage gathers
        if src_rows is not None:
            Q = ops.linear(x, Wj_c, row_index=src_rows[0], m_dev=src_rows[1], cache_planes=Wj.is_leaf)
        else:
            Q = ops.linear(x, Wj_c, cache_planes=Wj.is_leaf)
        arg = None
        if aggr == "max" and ea_sorted is not None and ea_sorted.shape[1] > 0 and graph.edge_maps() is not None:
            # the winners are recorded while aggregating (one int per target and channel): the backward pass then routes
            # every channel gradient without repeating the gather
            M, arg = ops.mpnn_aggregate_max_arg(p_bias, Q, We_c, ea_sorted, graph.rowptr, graph.src, node_order=graph.order,
                                                chunks=graph.chunks, skip_empty_rows=True)
        else:
            M = ops.mpnn_aggregate(None, p_bias, Q, We_c, ea_sorted, graph.rowptr, graph.src, aggr, node_order=graph.order,
                                   chunks=graph.chunks, skip_empty_rows=True)
        stats = main_stats = iso_stats = None
        if want_stats:
            panels = max(ops.stat_panels(n), 1)
            stats = torch.zeros((2 * panels, ops.STAT_ROWS, co), dtype=torch.float32, device=x.device)   # (zero panels count nothing)
            main_stats, iso_stats = stats[:panels], stats[panels:]
            ctx.mark_non_differentiable(stats)
        h = torch.empty((n, co), dtype=torch.float32, device=x.device)
        ops.linear(x, Wpx_c, bp.contiguous(), out=h, row_index=lst_e, m_dev=cnt_e, stats_out=iso_stats, cache_planes=Wpx.is_leaf)
        ops.linear(x, Wcomb_c, bcomb.contiguous(), a2=M, out=h, row_index=lst_ne, m_dev=cnt_ne, stats_out=main_stats,
                   cache_planes=Wcomb.is_leaf)
        ctx.graph, ctx.aggr, ctx.has_pb, ctx.has_arg = graph, aggr, p_bias is not None, arg is not None
        ctx.pool = ops.ctx().bounds
        ctx.fwd_bounds = (ops.bound_of(x), ops.bound_of(M))      # (kept beside the saved tensors: the weight gradients' f16x2 form)
        ctx.save_for_backward(x, ea_sorted, Q, M, Wj_c, We_c, Wcomb_c, Wpx_c, arg)
        return h, stats

    @staticmethod
    def backward(ctx, dh, _dstats):
        # the dgrad launches take the f16x2 form when their operands carry bounds: dh from rgnn_bn_bwd_apply_absmax, dM from the
        # launch that wrote it, dQ from rgnn_mpnn_max_bwd_absmax -- tracked in the pool of the forward pass
        with ops.using_bounds(ctx.pool):
            return ConvFoldedFn._backward(ctx, dh)

    @staticmethod
    def _backward(ctx, dh):
        x, ea, Q, M, Wj, We, Wcomb, Wpx, arg = ctx.saved_tensors
        bx, bM = ctx.fwd_bounds
        g = ctx.graph
        c = x.shape[1]
        lst_e, cnt_e, _, lst_ne, cnt_ne = g.split_targets()
        needs = ctx.needs_input_grad
        dh = dh.contiguous()
        WcT = Wcomb.t().contiguous()                                             # [C + D, Co]
        # dM = dh W_comb[:, C:] on the targets with edges (the only rows the edge stage reads)
        dM = ops.linear(dh, WcT[c:], row_index=lst_ne, m_dev=cnt_ne, cache_planes=False)
        scale = None
        if ctx.aggr == "mean":
            scale = (1.0 / g.in_degree().clamp(min=1.0)).view(-1).contiguous()
        dQ, dea, dWe = ops.mpnn_aggregate_bwd(dM, Q, We, ea, g.rowptr, g.src, ctx.aggr, g.source_csr(), node_order=g.order,
                                              target_scale=scale, edge_maps=g.edge_maps(), arg=arg)
        dx = None
        if needs[0]:
            dx = torch.empty_like(x)
            w_ne = torch.cat([WcT[:c], Wj.t()], dim=1).contiguous()          # [C, Co + D]: dx = [dh | dQ] [W_comb_x ; W_j]
            ops.linear(dh, w_ne, a2=dQ, out=dx, row_index=lst_ne, m_dev=cnt_ne, cache_planes=False)
            if g.symmetric:                                                      # isolated rows gather nothing and nobody gathers them
                ops.linear(dh, Wpx.t().contiguous(), out=dx, row_index=lst_e, m_dev=cnt_e, cache_planes=False)
            else:
                w_e = torch.cat([Wpx.t(), Wj.t()], dim=1).contiguous()
                ops.linear(dh, w_e, a2=dQ, out=dx, row_index=lst_e, m_dev=cnt_e, cache_planes=False)
        dWcomb = dbcomb = dWpx = dbp = dWj = dpb = None
        want_pb = ctx.has_pb and needs[4]
        if needs[5] or needs[6] or want_pb:
            t = ops.linear_wgrad(dh, x, M, with_bias=True, row_index=lst_ne, m_dev=cnt_ne, bounds=(ops.bound_of(dh), bx, bM))
            dWcomb, dbcomb = t[:, :-1], t[:, -1]
            if want_pb:
                # p_bias is added to M on the targets with edges: d p_bias = sum_ne dM = (sum_ne dh) W_comb[:, C:] -- the
                # column sums of dh over those rows are the bias column of the product above
                dpb = ops.linear(dbcomb.reshape(1, -1).contiguous(), WcT[c:], cache_planes=False).view(-1)
        if needs[7] or needs[8]:
            t = ops.linear_wgrad(dh, x, None, with_bias=True, row_index=lst_e, m_dev=cnt_e, bounds=(ops.bound_of(dh), bx, None))
            dWpx, dbp = t[:, :-1], t[:, -1]
        if needs[2]:
            sr = g.source_rows()                                                 # dQ is zero on the rows nobody gathered
            bq = (ops.bound_of(dQ), bx, None)
            dWj = (ops.linear_wgrad(dQ, x, None, bounds=bq) if sr is None
                   else ops.linear_wgrad(dQ, x, None, row_index=sr[0], m_dev=sr[1], bounds=bq))
        return (dx, dea if needs[1] else None, dWj, dWe if needs[3] else None, dpb, dWcomb, dbcomb, dWpx, dbp, None, None,
                None)


class BatchNormActFn(torch.autograd.Function):
    """y = act(BatchNorm1d(h)) with batch statistics (train mode) or running statistics (eval)."""

    @staticmethod
    def forward(ctx, h, gamma, beta, stats, bn_module, relu: bool):
        mod = bn_module.module
        use_batch = bn_module.training or mod.running_mean is None
        m, c = h.shape
        if use_batch and stats is None:
            stats = ops.column_stats(h)
        ss = bn_module.scale_shift(stats, m, in_bound=ops.bound_of(h))   # also updates the running statistics (train mode)
        y = ops.scale_shift_act(h, ss, relu=relu)                # (carries the table's bound: the next layer's f16x2 launches)
        ctx.use_batch, ctx.relu, ctx.m, ctx.eps = use_batch, relu, m, mod.eps
        ctx.pool = ops.ctx().bounds
        ctx.affine = gamma is not None
        # the backward coefficients come from the forward column statistics (or the running ones as they are NOW: eval mode
        # does not change them) -- one kernel, float64 inside (rgnn_bn_bwd_coef)
        # the ReLU mask of the backward pass: recomputed from h and the apply table (the bits of y) instead of reading y again
        ctx.mask_from_table = relu and ops.BN_BWD_MASK_FROM_TABLE and ss.dim() == 2
        ctx.table = ss if ctx.mask_from_table else None
        ctx.save_for_backward(h, y if (relu and not ctx.mask_from_table) else None, gamma, stats if use_batch else None,
                              None if use_batch else mod.running_mean.detach().clone(),
                              None if use_batch else mod.running_var.detach().clone())
        return y

    @staticmethod
    def backward(ctx, dy):
        h, y, gamma, stats, rmean, rvar = ctx.saved_tensors
        part = ops.bn_bwd_stats(dy, y, h, table=ctx.table)                          # [panels, 2, C]: sum g, sum g h
        coef, dgamma, dbeta = ops.bn_bwd_coef(stats, rmean, rvar, part, ctx.m, gamma, ctx.eps, ctx.use_batch)
        with ops.using_bounds(ctx.pool):
            dh = ops.bn_bwd_apply(dy, y, h, coef, table=ctx.table) if ctx.needs_input_grad[0] else None
        return (dh, dgamma if (ctx.affine and ctx.needs_input_grad[1]) else None,
                dbeta if (ctx.affine and ctx.needs_input_grad[2]) else None, None, None, None)


def batch_norm_act(h, bn_module, stats=None, relu=False):
    mod = bn_module.module
    if bn_module.empty_or_single(h.shape[0], h.shape[1]):       # (no rows: torch passes the empty matrix through; one row: an error)
        return torch.relu(h) if relu else h
    return BatchNormActFn.apply(h, mod.weight, mod.bias, stats, bn_module, relu)


class PermuteRowsFn(torch.autograd.Function):
    """rows = x[perm] for a PERMUTATION perm (the edge attributes in target order, TargetCSR.sort_edge_attr): the gradient is
    the same gather with the inverse permutation -- one coalesced-write kernel instead of torch's index backward (a sort +
    accumulate of 800 k rows, 216 us per training step)."""

    @staticmethod
    def forward(ctx, x, perm, inv_perm):
        ctx.save_for_backward(inv_perm)
        return ops.gather_rows(x, perm)

    @staticmethod
    def backward(ctx, g):
        (inv_perm,) = ctx.saved_tensors
        return ops.gather_rows(g.contiguous(), inv_perm), None, None


class AggregateFn(torch.autograd.Function):
    """M[t] = aggr_{e -> t} (Q[src_e] + We a_e), 0 for targets without incoming edges."""

    @staticmethod
    def forward(ctx, Q, We, ea_sorted, graph, aggr: str):
        Q = Q.contiguous() if Q.stride(1) != 1 else Q
        ctx.graph, ctx.aggr = graph, aggr
        ctx.has_edge = ea_sorted is not None and ea_sorted.shape[1] > 0
        We_c = We.contiguous() if We is not None else None
        ctx.save_for_backward(Q, We_c, ea_sorted)
        return ops.mpnn_aggregate(None, None, Q, We_c, ea_sorted, graph.rowptr, graph.src, aggr,
                                  node_order=graph.order, chunks=graph.chunks)

    @staticmethod
    def backward(ctx, dM):
        Q, We, ea = ctx.saved_tensors
        g = ctx.graph
        scale = None
        if ctx.aggr == "mean":
            scale = (1.0 / g.in_degree().clamp(min=1.0)).view(-1).contiguous()
        dQ, dea, dWe = ops.mpnn_aggregate_bwd(dM.contiguous(), Q, We, ea, g.rowptr, g.src, ctx.aggr, g.source_csr(),
                                              node_order=g.order, target_scale=scale, edge_maps=g.edge_maps())
        needs = ctx.needs_input_grad
        return (dQ if needs[0] else None, dWe if (ctx.has_edge and needs[1]) else None,
                dea if (ctx.has_edge and needs[2]) else None, None, None)


def edge_rows(P, p_bias, Q, We, ea_sorted, graph, relu: bool):
    """act(P[t_e] + p_bias + Q[s_e] + We a_e) per edge, [E, D].  The fused kernel takes up to 32 edge attributes; wider ones (an
    edge embedding wider than anything the reference ships) go through a dense launch over the edge rows instead."""
    if ea_sorted is None or ea_sorted.shape[1] <= ops.MAX_FUSED_EDGE_WIDTH:
        return EdgeHiddenFn.apply(P, p_bias, Q, None if We is None else We.contiguous(), ea_sorted, graph, relu)
    rows = EdgeHiddenFn.apply(P, p_bias, Q, None, None, graph, False) + linear(ea_sorted, We.contiguous())
    return torch.relu(rows) if relu else rows


def aggregate(Q, We, ea_sorted, graph, aggr: str):
    if ea_sorted is not None and ea_sorted.shape[1] > ops.MAX_FUSED_EDGE_WIDTH_BWD:
        # (the backward kernels of the fused aggregate stop at 16 edge attributes: per-edge rows + segmented reduce instead)
        return SegmentReduceFn.apply(edge_rows(None, None, Q, We, ea_sorted, graph, False), graph, aggr)
    return AggregateFn.apply(Q, We, ea_sorted, graph, aggr)


class SegmentReduceFn(torch.autograd.Function):
    """M[t] = aggr over the rows of segment t (general message path, pre_layers > 1)."""

    @staticmethod
    def forward(ctx, rows, graph, aggr: str):
        rows = rows.contiguous()
        ctx.graph, ctx.aggr = graph, aggr
        ctx.save_for_backward(rows)
        return ops.segment_reduce(rows, graph.rowptr, aggr, node_order=graph.order)

    @staticmethod
    def backward(ctx, dM):
        (rows,) = ctx.saved_tensors
        g = ctx.graph
        return ops.segment_reduce_bwd(dM.contiguous(), rows, g.rowptr, ctx.aggr, node_order=g.order), None, None


class EdgeHiddenFn(torch.autograd.Function):
    """H[e] = act(P[t_e] + p_bias + Q[s_e] + We a_e), rows in CSR-by-target order (first layer of a deeper message MLP)."""

    @staticmethod
    def forward(ctx, P, p_bias, Q, We, ea_sorted, graph, relu: bool):
        H = ops.mpnn_edge_hidden(P, p_bias, Q, We, ea_sorted, graph.rowptr, graph.src, relu=relu, node_order=graph.order,
                                 chunks=graph.chunks)
        ctx.graph, ctx.relu = graph, relu
        ctx.has_p, ctx.has_b = P is not None, p_bias is not None
        ctx.has_edge = ea_sorted is not None and ea_sorted.shape[1] > 0
        ctx.save_for_backward(H if relu else None, We, ea_sorted)
        return H

    @staticmethod
    def backward(ctx, dH):
        H, We, ea = ctx.saved_tensors
        g = ctx.graph
        G = dH.contiguous()
        if ctx.relu:
            G = ops.relu_bwd(G, H)
        needs = ctx.needs_input_grad
        dP = dpb = dQ = dWe = dea = None
        if ctx.has_p and needs[0]:
            dP = ops.segment_reduce(G, g.rowptr, "add", node_order=g.order)          # sum over the edges INTO each target
        if ctx.has_b and needs[1]:
            dpb = ops.column_sums(G)
        if needs[2]:
            rowptr_s, _, tpos = g.source_csr()
            dQ = ops.segment_reduce(ops.gather_rows(G, tpos), rowptr_s, "add", node_order=g.order)   # ... OUT of each source
        if ctx.has_edge and needs[3]:
            dWe = ops.linear_wgrad(G, ea)                        # (any row count and widths: rgnn_wgrad pads; no BLAS)
        if ctx.has_edge and needs[4]:
            dea = ops.linear(G, We.t().contiguous(), cache_planes=False)
        return dP, dpb, dQ, dWe, dea, None, None


def has_incoming(graph) -> torch.Tensor:
    """float32 [N, 1]: 1 for targets with at least one incoming edge (CSR segments are in visiting order)."""
    if getattr(graph, "_has_in", None) is None:
        seg = (graph.rowptr[1:] > graph.rowptr[:-1]).to(torch.float32)          # per segment p
        if graph.order is not None:
            mask = torch.empty_like(seg)
            mask[graph.order.long()] = seg
        else:
            mask = seg
        graph._has_in = mask.view(-1, 1)
    return graph._has_in
