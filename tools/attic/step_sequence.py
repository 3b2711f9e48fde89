"""Five eager C2 steps for a kernel trace whose LAST step tools/attic/list_step_kernels.py prints:
    rocprofv3 --kernel-trace --output-format csv -d out -o ks -- python tools/attic/step_sequence.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from radargnn_amd import frames as fr, synthetic
model = bench.c2_model().cuda()
batch = fr.FrameBatch.from_frames([synthetic.radarscenes_frame(i) for i in range(64)])
hot = fr.HotPath(model, bench.c2_settings())
for _ in range(6):
    hot(batch)
torch.cuda.synchronize()
