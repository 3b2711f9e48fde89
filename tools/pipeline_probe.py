"""What a captured C2 step costs when the graph stage of the NEXT batch runs beside the model stage of the current one:
(a) the captured step as it is (search -> features -> CSR / plan -> model, one chain), (b) the model stage alone on a prepared graph,
(c) the model stage with a complete graph stage (search, fill, features, CSR by target, window plan) as a second branch of the same
captured graph.    python tools/pipeline_probe.py [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from radargnn_amd import frames as fr, ops, synthetic

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
settings = bench.c2_settings()
model = bench.c2_model().cuda()
batch = fr.FrameBatch.from_frames([synthetic.radarscenes_frame(i) for i in range(64)])
dev = batch.X.device


def timeit(fn, n=steps):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


hot = fr.HotPath(model, settings, use_hip_graphs=True)
for _ in range(3):
    ref = hot(batch)
full_ms = timeit(lambda: hot(batch))
ref_cls = ref[0].clone()
n_edges = hot._seen_edges


def graph_stage(static, status, rows):
    st = fr._stage_search(batch, settings, status, static=static)
    g = fr._stage_features(batch, settings, status, st, n_edges, guarded=True, committed=rows)
    g.big_edge_fraction = hot._seen_big
    graph, ea = hot._prepare(g, plan_here=True)
    return g, graph, ea


def make_set():
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    static = {}
    st0 = fr._stage_search(batch, settings, status, static=static)
    rows = (st0["rowptr"].clone(), st0["deg"].clone())
    return static, status, rows


hot._frame_ptr = batch.frame_ptr
hot._biggest_frame = int(batch.frame_sizes.max())
sa, sta, ra = make_set()
gA, graphA, eaA = graph_stage(sa, sta, ra)
graphA.join_csr(); graphA.join_win_plan()
torch.cuda.synchronize()

gm = torch.cuda.CUDAGraph()
with torch.cuda.graph(gm):
    outs_m = hot._forward(gA, graphA, eaA)
gm.replay(); torch.cuda.synchronize()
print("model-only output equals the captured step's:", bool(torch.equal(outs_m[0], ref_cls)))
model_ms = timeit(gm.replay)

sb, stb, rb = make_set()
torch.cuda.synchronize()
side = ops.ctx().side(dev, "prep")
go = torch.cuda.CUDAGraph()
with torch.cuda.graph(go):
    main = torch.cuda.current_stream(dev)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        stb.zero_()
        gB, graphB, eaB = graph_stage(sb, stb, rb)
        graphB.join_csr(); graphB.join_win_plan()
    outs_o = hot._forward(gA, graphA, eaA)
    main.wait_stream(side)
go.replay(); torch.cuda.synchronize()
print("overlapped output equals:", bool(torch.equal(outs_o[0], ref_cls)))
over_ms = timeit(go.replay)
res = []
for _ in range(3):
    res.append((round(timeit(lambda: hot(batch)), 4), round(timeit(gm.replay), 4), round(timeit(go.replay), 4)))
print({"captured step (one chain)": round(full_ms, 4), "model stage alone": round(model_ms, 4),
       "model stage beside a whole graph stage": round(over_ms, 4), "alternating repeats (step, model, overlapped)": res})
