"""Which ATen ops (and from where) run inside one C2 training step: python tools/prof_aten_ops_train.py"""
import sys, os, collections
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from radargnn_amd import frames as fr, synthetic
from radargnn_amd.gnn.losses import detection_loss
model = bench.c2_model().cuda()
batch = fr.FrameBatch.from_frames([synthetic.radarscenes_frame(i) for i in range(16)])
g = fr.build_graphs(batch, bench.c2_settings())
n = g.x.shape[0]
y = torch.cat((torch.randint(0, 6, (n, 1), device="cuda").float(), torch.randn(n, 5, device="cuda")), 1)
opt = torch.optim.Adam(model.parameters(), lr=1e-3)
x, ei, ea = g.x, g.edge_index, g.edge_attr
def step():
    opt.zero_grad(); x.requires_grad_(); ea.requires_grad_()
    c, bb = model(x, ei, ea)
    loss, _, _ = detection_loss(c, bb, y, 5, [1.0] * 5 + [0.3])
    loss.backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(); torch.cuda.synchronize()
skip = {"aten::empty", "aten::view", "aten::as_strided", "aten::empty_strided", "aten::slice", "aten::select", "aten::detach", "aten::reshape",
        "aten::t", "aten::transpose", "aten::_unsafe_view", "aten::alias", "aten::numel", "aten::is_nonzero", "aten::item", "aten::_local_scalar_dense",
        "aten::stride", "aten::lift_fresh", "aten::unsqueeze", "aten::expand", "aten::resize_", "aten::empty_like", "aten::narrow", "aten::squeeze", "aten::result_type", "aten::to"}
cnt = collections.Counter()
for e in prof.events():
    if e.name.startswith("aten::") and e.name not in skip:
        st = [s.split("/")[-1] for s in (e.stack or []) if "radargnn_amd" in s or "optim" in s]
        cnt[(e.name, st[0] if st else "?")] += 1
for (name, where), c in cnt.most_common(45):
    print(f"{c:4d} {name:28s} {where}")
