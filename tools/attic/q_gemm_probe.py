"""Why is the source-term GEMM slower inside the step than in tools/x3_bench?  Times ops.linear on the C2 graph's real row list
against a random row list of the same length, with / without the fused BatchNorm-apply, with random / ReLU-sparse inputs."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from radargnn_amd import frames as fr, ops, synthetic
from radargnn_amd.gnn.mpnn_layers import TargetCSR


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(n):
        e0.record(); fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


batch = fr.FrameBatch.from_frames([synthetic.radarscenes_frame(i) for i in range(64)])
g = fr.build_graphs(batch, fr.GraphSettings(algorithm="radius", r=1.0))
n = g.x.shape[0]
graph = TargetCSR(g.edge_index, n, order=g.cell_order, symmetric=True, source_rows=g.rowptr, status=g.status)
rows, cnt = graph.source_rows()
m = int(cnt.item())
gen = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(n, 224, device="cuda", generator=gen)
w = torch.randn(464, 224, device="cuda", generator=gen) / 15
aff = torch.stack([torch.rand(224, device="cuda", generator=gen) + 0.5, torch.randn(224, device="cuda", generator=gen)])
rnd = torch.full((n,), -1, dtype=torch.int32, device="cuda")
rnd[:m] = torch.randperm(n, device="cuda", generator=gen)[:m].sort().values.to(torch.int32)
srt = torch.full((n,), -1, dtype=torch.int32, device="cuda")
srt[:m] = rows[:m].sort().values
out = torch.empty(n, 464, device="cuda")
print(f"N = {n}, rows with edges = {m}")
for name, lst in (("real list (visiting order)", rows), ("real list, sorted by node id", srt), ("random rows, sorted", rnd)):
    for aname, a in (("plain", None), ("fused BatchNorm-apply", aff)):
        med, mn = timeit(lambda: ops.linear(x, w, out=out, row_index=lst, m_dev=cnt, a1_affine=a))
        print(f"  {name:30s} {aname:22s} {med:7.1f} us (min {mn:7.1f})", flush=True)
