"""frames.FrameStreamer (host frames -> pinned staging -> copy stream -> hot path -> download stream -> host results, the pipeline
form of postprocessor/inference.py:48-68) hands out, batch by batch, exactly what the resident path computes."""
import pytest
import torch

from radargnn_amd import synthetic

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("algo", ["radius", "knn"])
def test_streamed_batches_equal_resident_results(algo):
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test but no GPU visible")
    from radargnn_amd import frames as fr, gnn
    torch.manual_seed(0)
    cfg = gnn.GNNArchitectureConfig(5, 2, [32, 64], [4, 8, 16], [16, 5], True, False, [64, 32], [6], "MPNNConv", False)
    model = gnn.DetNetBasic(cfg).cuda().eval()               # eval: no running statistics move between the two passes
    settings = fr.GraphSettings(algorithm=algo, k=6, r=2.5)
    sizes = [5, 3, 6, 4, 5, 2, 7, 4]                          # ragged batches: the staging slots grow and are reused
    host = [[synthetic.nuscenes_frame(100 * b + i) for i in range(sizes[b])] for b in range(8)]
    hot = fr.HotPath(model, settings)
    resident = []
    for frames in host:
        cls, bb, g = hot(fr.FrameBatch.from_frames(frames))
        g.check()
        resident.append((cls.cpu(), bb.cpu()))
    streamed = [(c.clone(), b.clone()) for c, b in fr.FrameStreamer(hot, slots=3).run(iter(host))]
    assert len(streamed) == len(resident)
    for (c0, b0), (c1, b1) in zip(resident, streamed):
        assert torch.equal(c0, c1) and torch.equal(b0, b1)
    # a second run on the same streamer (slots and streams are reused), two slots
    again = [(c.clone(), b.clone()) for c, b in fr.FrameStreamer(hot, slots=2).run(iter(host[:3]))]
    assert all(torch.equal(a[0], r[0]) and torch.equal(a[1], r[1]) for a, r in zip(again, resident))
    # the defaults (six slots, results handed out two batches late), and the form without the search half a batch ahead
    for kw in ({}, {"lookahead": False}, {"slots": 4, "behind": 3}):
        out = [(c.clone(), b.clone()) for c, b in fr.FrameStreamer(hot, **kw).run(iter(host))]
        assert len(out) == len(resident) and all(torch.equal(a[0], r[0]) and torch.equal(a[1], r[1]) for a, r in zip(out, resident))


def test_streamer_passes_on_the_loaders_exception():
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test but no GPU visible")
    from radargnn_amd import frames as fr, gnn
    cfg = gnn.GNNArchitectureConfig(5, 2, [32], [8], [16, 5], True, False, [32], [6], "MPNNConv", False)
    hot = fr.HotPath(gnn.DetNetBasic(cfg).cuda().eval(), fr.GraphSettings(algorithm="radius", r=2.0))

    def batches():
        yield [synthetic.nuscenes_frame(0)]
        raise KeyError("no such sequence")

    with pytest.raises(KeyError):
        list(fr.FrameStreamer(hot).run(batches()))
