"""Two models driven from two Python threads, each on its own stream, interleaved: every thread gets exactly the bits the same calls
give alone.  What a forward pass carries besides its tensors -- the bound pool of the f16x2 dense form, the
frame scope of per-frame BatchNorm statistics, the launch profiler -- lives in ops.ForwardContext, one per thread."""
import threading

import pytest
import torch

from radargnn_amd import synthetic

pytestmark = pytest.mark.gpu


def test_two_models_on_two_streams_from_two_threads():
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test but no GPU visible")
    from radargnn_amd import frames as fr, gnn, ops
    torch.manual_seed(0)
    cfg_a = gnn.GNNArchitectureConfig(5, 2, [32, 64], [4, 8, 16], [16, 5], True, False, [224, 64], [6], "MPNNConv", False)
    cfg_b = gnn.GNNArchitectureConfig(5, 2, [32, 64, 128], [4, 8, 16], [16, 5], True, True, [96, 32], [11], "MPNNConv", False)
    jobs = [
        # (model, settings, per-frame statistics?, frames): wide layers (f16x2 form, bounds) / BatchNorm inside the MLPs, kNN
        (gnn.DetNetBasic(cfg_a).cuda(), fr.GraphSettings(algorithm="radius", r=2.5), "frame",
         [synthetic.radarscenes_frame(i) for i in range(6)]),
        (gnn.DetNetBasic(cfg_b).cuda(), fr.GraphSettings(algorithm="knn", k=8), "batch",
         [synthetic.nuscenes_frame(i) for i in range(40)]),
    ]
    rounds = 6

    def run(job, out, stream=None):
        model, settings, scope, frames = job
        hot = fr.HotPath(model, settings, bn_scope=scope)
        batch = fr.FrameBatch.from_frames(frames)
        ctxm = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream())
        with ctxm:
            for _ in range(rounds):                       # train-mode BatchNorm: the running statistics move every round
                cls, bb, g = hot(batch)
                out.append((cls.clone(), bb.clone()))
            g.check()
            torch.cuda.current_stream().synchronize()

    # alone, one after the other (fresh copies of the models: the running statistics are part of the state)
    import copy
    alone = [[], []]
    for k, job in enumerate(jobs):
        run((copy.deepcopy(job[0]),) + job[1:], alone[k])
    # together
    both, errors = [[], []], []

    def worker(k):
        try:
            torch.cuda.set_device(0)
            run(jobs[k], both[k], torch.cuda.Stream())
        except BaseException as exc:                      # noqa: BLE001
            errors.append(exc)

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert ops.ctx().bounds is None and ops.ctx().frame_scope is None
    for k in range(2):
        assert len(both[k]) == rounds
        for (c0, b0), (c1, b1) in zip(alone[k], both[k]):
            assert torch.equal(c0, c1) and torch.equal(b0, b1)


def test_context_is_per_thread():
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test but no GPU visible")
    from radargnn_amd import ops
    seen = {}
    with ops.bound_tracking(torch.device("cuda", 0)):
        mine = ops.ctx().bounds

        def other():
            seen["bounds"] = ops.ctx().bounds
            seen["same_ctx"] = ops.ctx() is main_ctx

        main_ctx = ops.ctx()
        t = threading.Thread(target=other)
        t.start(); t.join()
    assert (mine is not None) == ops.USE_F16X2 and seen["bounds"] is None and seen["same_ctx"] is False
    assert ops.ctx().bounds is None
