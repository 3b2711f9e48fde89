import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
from radargnn_amd import frames as fr, synthetic
model = bench.c2_model().cuda()
batch = fr.FrameBatch.from_frames([synthetic.radarscenes_frame(i) for i in range(8)])
hot = fr.HotPath(model, bench.c2_settings())
for _ in range(3): hot(batch)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    hot(batch); torch.cuda.synchronize()
for e in prof.events():
    if e.name.startswith("aten::") and e.name not in ("aten::empty", "aten::view", "aten::as_strided", "aten::empty_strided", "aten::slice", "aten::select", "aten::detach", "aten::reshape", "aten::t", "aten::transpose", "aten::_unsafe_view", "aten::alias", "aten::numel", "aten::is_nonzero", "aten::item", "aten::_local_scalar_dense", "aten::contiguous", "aten::stride", "aten::lift_fresh", "aten::unsqueeze", "aten::expand", "aten::resize_"):
        st = [s for s in (e.stack or []) if "radargnn_amd" in s or "bench" in s]
        print(e.name, st[:2])
