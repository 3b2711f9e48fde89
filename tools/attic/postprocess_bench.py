"""Post-processor front half timed on the C2 batch shape.  One JSON line: the decode kernel against its
HBM roofline (algorithmic bytes = 4 (K + W + 2) read + 76 written per node), NMS of one frame's boxes (pairs / s), and the
oracle's per-node Python loops (the reference's form) on a bounded sample beside them."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from radargnn_amd import ops, postprocessor as P  # noqa: E402
from oracle import postprocess_oracle as O  # noqa: E402  (cpu baseline leg only)


def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    rng = np.random.default_rng(0)
    n, k, w = 192000, 6, 5
    prob = torch.softmax(torch.from_numpy(rng.normal(size=(n, k)) * 2).float(), 1).cuda()
    bb = torch.from_numpy(np.abs(rng.normal(size=(n, w))) + 0.3).float().cuda()
    pos = torch.from_numpy(rng.uniform(0, 100, size=(n, 2))).float().cuda()
    cfg = P.PostProcessingConfiguration(split="t", iou_for_nms=0.3, min_object_score={c: 0.3 for c in "abcde"},
                                        max_score_for_background=0.4, bg_index=5, bb_invariance="translation")
    ms_decode = timed(lambda: P.decode(prob, bb, pos, cfg), 50)
    dec_bytes = n * (4 * (k + w + 2) + 76)
    cfg_en = P.PostProcessingConfiguration(split="t", iou_for_nms=0.3, min_object_score={}, max_score_for_background=0.4, bg_index=5,
                                           bb_invariance="en")
    ptr = torch.arange(0, n + 1, 3000, dtype=torch.int64).cuda()
    ms_decode_en = timed(lambda: P.decode(prob, bb, pos, cfg_en, frame_ptr=ptr), 20)
    out = {"what": "post-processor front half, C2 batch (64 frames x 3000 nodes)", "decode_ms": ms_decode,
           "decode_algorithmic_bytes": dec_bytes, "decode_GBps": dec_bytes / ms_decode / 1e6, "decode_frac_of_8TBps": dec_bytes / ms_decode / 1e6 / 8000,
           "decode_en_with_k1_search_ms": ms_decode_en}
    for m in (500, 2000, 6000):
        boxes = np.stack((rng.uniform(0, 100, m), rng.uniform(-50, 50, m), rng.uniform(1, 6, m), rng.uniform(0.5, 3, m), rng.uniform(0, 180, m)), 1)
        b = torch.from_numpy(boxes).cuda(); s = torch.from_numpy(rng.uniform(0, 1, m)).cuda()
        ms = timed(lambda: ops.nms(b, s, 0.3, rotated=True), 10)
        out[f"nms_rotated_{m}_ms"] = ms
        out[f"nms_rotated_{m}_Mpairs_per_s"] = m * (m - 1) / 2 / ms / 1e3
        tp = torch.from_numpy(np.concatenate((boxes[:, :2], boxes[:, :2] + boxes[:, 2:4]), 1)).float().cuda()
        out[f"nms_aligned_{m}_ms"] = timed(lambda: ops.nms(tp, s.float(), 0.3, rotated=False), 10)
    # CPU: the reference-shaped per-node loops (oracle) on one frame, scaled to the batch
    sub = slice(0, 3000)
    pc, bc, xc = prob[sub].cpu().numpy(), bb[sub].cpu().numpy(), pos[sub].cpu().numpy()
    t0 = time.perf_counter()
    O.absolute_object_boxes(pc, bc, xc, 5, 0.4, [0.3] * 5, "translation", False)
    cpu_frame = time.perf_counter() - t0
    boxes = np.stack((rng.uniform(0, 100, 500), rng.uniform(-50, 50, 500), rng.uniform(1, 6, 500), rng.uniform(0.5, 3, 500), rng.uniform(0, 180, 500)), 1)
    t0 = time.perf_counter()
    O.nms_rotated(boxes, rng.uniform(0, 1, 500), 0.3)
    out.update({"cpu_port_decode_ms_per_frame_3000_nodes": cpu_frame * 1e3, "cpu_port_decode_ms_per_batch_scaled": cpu_frame * 64e3,
                "cpu_port_nms_rotated_500_ms": (time.perf_counter() - t0) * 1e3, "cpu_port_kind": "oracle (numpy loops, 1 core)"})
    print(json.dumps(out))


if __name__ == "__main__":
    main()
