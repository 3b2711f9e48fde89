import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from radargnn_amd import frames as fr, ops, synthetic
v = sys.argv[1]
settings = bench.c2_settings(); model = bench.c2_model().cuda()
batch = fr.FrameBatch.from_frames([synthetic.radarscenes_frame(i) for i in range(int(os.environ.get("NF", "64")))])
dev = batch.X.device
hot = fr.HotPath(model, settings, use_hip_graphs=True)
for _ in range(3): ref = hot(batch)
n_edges = hot._seen_edges
hot._frame_ptr = batch.frame_ptr; hot._biggest_frame = int(batch.frame_sizes.max())
def make_set():
    status = torch.zeros(1, dtype=torch.int32, device=dev); static = {}
    st0 = fr._stage_search(batch, settings, status, static=static)
    return static, status, (st0["rowptr"].clone(), st0["deg"].clone())
def graph_stage(static, status, rows, prep=True):
    st = fr._stage_search(batch, settings, status, static=static)
    g = fr._stage_features(batch, settings, status, st, n_edges, guarded=True, committed=rows)
    g.big_edge_fraction = hot._seen_big
    if not prep: return g, None, None
    graph, ea = hot._prepare(g)
    return g, graph, ea
sa, sta, ra = make_set()
gA, graphA, eaA = graph_stage(sa, sta, ra)
graphA.join_csr(); graphA.join_win_plan()
sb, stb, rb = make_set()
torch.cuda.synchronize()
side = ops.ctx().side(dev, "prep")
go = torch.cuda.CUDAGraph()
with torch.cuda.graph(go):
    main = torch.cuda.current_stream(dev)
    if "dummy" in v: sta.zero_()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        stb.zero_()
        if "searchonly" in v:
            st = fr._stage_search(batch, settings, stb, static=sb)
        else:
            gB, graphB, eaB = graph_stage(sb, stb, rb, prep="noprep" not in v)
            if graphB is not None: graphB.join_csr(); graphB.join_win_plan()
    outs = hot._forward(gA, graphA, eaA)
    main.wait_stream(side)
print(v, "captured", flush=True)
go.replay(); torch.cuda.synchronize()
print(v, "ok", bool(torch.equal(outs[0], ref[0])), flush=True)
