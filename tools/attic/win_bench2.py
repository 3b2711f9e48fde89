"""Window kernel (rgnn_mpnn_aggregate_win) against the per-edge kernel on the graphs the configurations build, rows of Q padded to
128-byte lines as inside the model (tools only; float64 on sampled targets as the checker).
    python tools/attic/win_bench2.py [-r rounds] [--cases c2,c4,c3,c5] [--dims 464,272,144]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from radargnn_amd import frames as fr, ops, synthetic
from radargnn_amd.gnn.mpnn_layers import TargetCSR


def graph(case):
    if case == "c2":
        fs = [synthetic.radarscenes_frame(i) for i in range(64)]
        g = fr.build_graphs(fr.FrameBatch.from_frames(fs), fr.GraphSettings(algorithm="radius", r=1.0))
        return g, TargetCSR(g.edge_index, g.x.shape[0], order=g.cell_order, rank=g.cell_rank, symmetric=True, source_rows=g.rowptr)
    if case == "c4":
        fs = [synthetic.radarscenes_frame(i) for i in range(64)]
    elif case == "c3":
        fs = [synthetic.nuscenes_frame(i) for i in range(512)]
    elif case == "c5":
        g = fr.build_graphs(fr.FrameBatch.from_frames([synthetic.stress_cloud()]), fr.GraphSettings(algorithm="radius", r=1.0))
        return g, TargetCSR(g.edge_index, g.x.shape[0], order=g.cell_order, rank=g.cell_rank, symmetric=True, source_rows=g.rowptr)
    g = fr.build_graphs(fr.FrameBatch.from_frames(fs), fr.GraphSettings(algorithm="knn", k=20))
    return g, (g.csr if g.csr is not None else TargetCSR(g.edge_index, g.x.shape[0], order=g.cell_order, rank=g.cell_rank, all_sources=True))


def main():
    argv = sys.argv[1:]
    rounds, cases, dims, track = 5, ["c2", "c4", "c3"], [464, 272, 144], False
    while argv:
        a = argv.pop(0)
        if a == "-r": rounds = int(argv.pop(0))
        elif a == "--cases": cases = argv.pop(0).split(",")
        elif a == "--dims": dims = [int(v) for v in argv.pop(0).split(",")]
        elif a == "--amax": track = True          # launches track max |out| like inside the model (ops.bound_tracking)
    if track:
        ops.bound_tracking("cuda").__enter__()
        ops.ctx().bounds.buf = torch.zeros((4096, ops.BOUND_SLOTS), dtype=torch.float32, device="cuda")
    for case in cases:
        g, csr = graph(case)
        csr.join_win_plan()
        n, e = g.x.shape[0], csr.num_edges
        plan = csr.win_plan() if getattr(csr, "_win_plan", None) is not None else ops.mpnn_win_plan(csr.rowptr, csr.src, csr.order)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ops.mpnn_win_plan(csr.rowptr, csr.src, csr.order, out=plan)
        e1.record(); torch.cuda.synchronize()
        print(f"[{case}] N={n} E={e}: plan {e0.elapsed_time(e1) / 5 * 1e3:.0f} us")
        for d in dims:
            torch.manual_seed(0)
            Q = ops.padded_rows(n, d, "cuda"); Q.copy_(torch.randn(n, d, device="cuda"))
            We = (torch.randn(d, 8, device="cuda") * 0.3).contiguous()
            ea = torch.randn(e, 8, device="cuda").relu_()
            bias = torch.randn(d, device="cuda")
            old = lambda: ops.mpnn_aggregate(None, bias, Q, We, ea, csr.rowptr, csr.src, "max", node_order=csr.order, chunks=csr.chunks, skip_empty_rows=True)
            new = lambda: ops.mpnn_aggregate_win(bias, Q, We, ea, csr.rowptr, csr.src, plan, node_order=csr.order, skip_empty_rows=True)
            oa, ob = old(), new()
            torch.cuda.synchronize()
            # float64 on sampled targets (positions in visiting order -> node ids)
            rp = csr.rowptr.cpu(); order = csr.order.cpu().long() if csr.order is not None else torch.arange(n)
            gen = torch.Generator().manual_seed(1)
            pos = torch.randint(0, n, (400,), generator=gen)
            errs_a, errs_b, big = 0.0, 0.0, 0.0
            srcs, Q64, We64, ea64 = csr.src.cpu().long(), Q.double().cpu(), We.double().cpu(), ea.double().cpu()
            for p_ in pos.tolist():
                a_, b_ = int(rp[p_]), int(rp[p_ + 1])
                if a_ == b_:
                    continue
                m = (Q64[srcs[a_:b_]] + ea64[a_:b_] @ We64.t()).max(0).values + bias.double().cpu()
                node = int(order[p_])
                errs_a = max(errs_a, float((oa[node].double().cpu() - m).abs().max())); errs_b = max(errs_b, float((ob[node].double().cpu() - m).abs().max()))
                big = max(big, float(m.abs().max()))
            ta, tb = [], []
            for _ in range(rounds):
                for f, acc in ((old, ta), (new, tb)):
                    f()
                    e0.record()
                    for _ in range(5):
                        f()
                    e1.record(); torch.cuda.synchronize(); acc.append(e0.elapsed_time(e1) / 5)
            ta.sort(); tb.sort()
            ma, mb = ta[len(ta) // 2], tb[len(tb) // 2]
            print(f"[{case}] D={d}: per-edge {ma * 1e3:7.1f} us   window {mb * 1e3:7.1f} us   x{ma / mb:.3f}   err vs f64 (norm-wise): per-edge {errs_a / big:.1e} window {errs_b / big:.1e}")


if __name__ == "__main__":
    main()
