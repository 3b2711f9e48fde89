"""Time one training step (trainer.py:175-231 shape: forward, cross-entropy + Huber, backward, Adam) of the C2 workload
on the HIP path.  Not the headline metric (that is bench.py, inference); a measurement for MEASUREMENTS.md section 8."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from radargnn_amd import frames as fr, synthetic
from radargnn_amd.gnn.mpnn_layers import TargetCSR

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
model = bench.c2_model().cuda()
settings = fr.GraphSettings(algorithm="radius", k=0, r=1.0)
batch = fr.FrameBatch.from_frames([synthetic.radarscenes_frame(i) for i in range(64)])
g = fr.build_graphs(batch, settings)
n = g.x.shape[0]
label = torch.randint(0, 6, (n,), device="cuda")
box = torch.randn(n, 5, device="cuda")
y = torch.cat((label.float().view(-1, 1), box), 1)              # graph_batch.y: label | box (trainer.py:185-186)
from radargnn_amd.gnn.losses import detection_loss
opt = torch.optim.Adam(model.parameters(), lr=1e-3)
x, ei, ea = g.x, g.edge_index, g.edge_attr


def step():
    opt.zero_grad()
    x.requires_grad_(); ea.requires_grad_()
    c, bb = model(x, ei, ea)
    loss, _, _ = detection_loss(c, bb, y, 5, [1.0, 1.0, 1.0, 1.0, 1.0, 0.3])   # trainer.py:181-222 on the device
    loss.backward()
    opt.step()
    return loss


for _ in range(2):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    l = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
with torch.no_grad():
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        model(x, ei, ea)
    torch.cuda.synchronize()
    df = (time.perf_counter() - t0) / steps
print(f"train step {dt*1e3:.2f} ms ({64/dt:.0f} frames/s), forward only {df*1e3:.2f} ms, loss {l.item():.4f}, N={n}, E={ei.shape[1]}")
