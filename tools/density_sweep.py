"""Re-validates the dispatch thresholds of the max aggregation (gnn/mpnn_layers.py: window kernel iff MIN_DEGREE <= E / N < MAX_DEGREE and
E >= MIN_EDGES / MIN_EDGES_SPARSE) on graphs OTHER than the five benchmark workloads: RadarScenes-shaped batches with the radius swept
so that the mean degree runs from ~1.5 to ~60, at batch sizes from 2^16 to 2^22 edges; per point the per-edge kernel and the window
kernel (plan excluded: it is built once per graph on a side stream; printed beside) at D = 464 and D = 144.  Tools only.
    python tools/density_sweep.py > profiles/rNN_density_sweep.txt"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from radargnn_amd import frames as fr, ops, synthetic
from radargnn_amd.gnn.mpnn_layers import TargetCSR


def time_it(f, reps=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(3):
        e0.record()
        for _ in range(reps):
            f()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / reps)
    return sorted(ts)[1] * 1e3


def main():
    print("frames  radius   N        E        E/N    D    per-edge us  window us  ratio   plan us  rule")
    for n_frames in (4, 16, 64, 128):
        frames = [synthetic.radarscenes_frame(i) for i in range(n_frames)]
        batch = fr.FrameBatch.from_frames(frames)
        for r in (0.5, 0.8, 1.0, 1.5, 2.0, 2.5, 3.0, 4.0, 5.0):
            g = fr.build_graphs(batch, fr.GraphSettings(algorithm="radius", r=r))
            n, e = g.x.shape[0], g.edge_index.shape[1]
            if e < (1 << 15) or e > (1 << 23):
                continue
            csr = TargetCSR(g.edge_index, n, order=g.cell_order, rank=g.cell_rank, symmetric=True, source_rows=g.rowptr)
            rule = csr.wants_window_kernel()
            plan = ops.mpnn_win_plan(csr.rowptr, csr.src, csr.order)
            t_plan = time_it(lambda: ops.mpnn_win_plan(csr.rowptr, csr.src, csr.order, out=plan))
            for d in (464, 144):
                torch.manual_seed(0)
                Q = ops.padded_rows(n, d, "cuda"); Q.copy_(torch.randn(n, d, device="cuda"))
                We = (torch.randn(d, 8, device="cuda") * 0.3).contiguous()
                ea = torch.randn(e, 8, device="cuda").relu_()
                bias = torch.randn(d, device="cuda")
                t_old = time_it(lambda: ops.mpnn_aggregate(None, bias, Q, We, ea, csr.rowptr, csr.src, "max", node_order=csr.order,
                                                           chunks=csr.chunks, skip_empty_rows=True))
                t_new = time_it(lambda: ops.mpnn_aggregate_win(bias, Q, We, ea, csr.rowptr, csr.src, plan, node_order=csr.order,
                                                               skip_empty_rows=True))
                good = (t_new < t_old) == rule
                print(f"{n_frames:5d}  {r:6.2f}  {n:7d}  {e:8d}  {e / n:5.1f}  {d:4d}  {t_old:10.1f}  {t_new:9.1f}  {t_old / t_new:5.2f}  {t_plan:7.0f}  "
                      f"{'window' if rule else 'per-edge'}{'' if good else '   <-- rule picks the slower kernel'}", flush=True)
            del csr, plan, g
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
