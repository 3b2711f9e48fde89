"""Ten captured C2 steps with PER-FRAME BatchNorm statistics (HotPath(bn_scope="frame")) for a kernel trace:
    rocprofv3 --kernel-trace --stats --output-format csv -d out -o fb -- python tools/attic/frame_bn_profile.py [batch|frame]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from radargnn_amd import frames as fr, synthetic
scope = sys.argv[1] if len(sys.argv) > 1 else "frame"
model = bench.c2_model().cuda()
batch = fr.FrameBatch.from_frames([synthetic.radarscenes_frame(i) for i in range(64)])
hot = fr.HotPath(model, bench.c2_settings(), bn_scope=scope)
for _ in range(10):
    hot(batch)
torch.cuda.synchronize()
