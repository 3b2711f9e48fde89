"""The trainer's loss (gnn/trainer.py:181-222) on the C2 batch: device kernels against the reference's per-node loop."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radargnn_amd.gnn.losses import detection_loss
from oracle import loss_oracle   # cpu leg only

g = torch.Generator().manual_seed(0)
n = 192000
cls = (torch.randn(n, 6, generator=g) * 2).cuda().requires_grad_(True)
bb = torch.randn(n, 5, generator=g).cuda().requires_grad_(True)
label = torch.where(torch.rand(n, generator=g) < 0.3, torch.randint(0, 5, (n,), generator=g), torch.full((n,), 5))
y = torch.cat((label.float().view(-1, 1), torch.randn(n, 5, generator=g)), 1).cuda()


def step():
    loss, _, _ = detection_loss(cls, bb, y, 5, [1.0, 1.0, 1.0, 1.0, 1.0, 0.3])
    loss.backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    step()
e1.record(); torch.cuda.synchronize()
dev_ms = e0.elapsed_time(e1) / 20
sub = 3000
c64 = cls.detach().cpu()[:sub].double().requires_grad_(True); b64 = bb.detach().cpu()[:sub].double().requires_grad_(True)
t0 = time.perf_counter()
l = loss_oracle.detection_loss(c64, b64, y.cpu()[:sub].double(), 5, [1.0, 1.0, 1.0, 1.0, 1.0, 0.3])[0]
l.backward()
cpu_s = time.perf_counter() - t0
nbytes = n * (4 * (6 + 5 + 6) + 4 * (6 + 5 + 6) + 4 * (6 + 5))       # forward read, backward read + gradient write
print(json.dumps({"what": "detection loss forward + backward, C2 batch (192 000 nodes)", "device_ms": dev_ms,
                  "algorithmic_bytes": nbytes, "GBps": nbytes / dev_ms / 1e6,
                  "cpu_reference_loop_ms_per_3000_nodes": cpu_s * 1e3, "cpu_reference_loop_ms_scaled_to_batch": cpu_s * 64e3}))
