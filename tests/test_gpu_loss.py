"""The trainer's loss on the device (csrc/loss.hip, radargnn_amd/gnn/losses.py) against the reference's own formulation
(oracle/loss_oracle.py: the same torch modules and per-node loop, float64): values and gradients."""
import numpy as np
import pytest
import torch

from oracle import loss_oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from radargnn_amd.gnn import losses
    return losses


def make(n, k, w, bg, seed, obj_frac=0.4):
    g = torch.Generator().manual_seed(seed)
    cls = torch.randn(n, k, generator=g) * 2
    bb = torch.randn(n, w, generator=g) * 1.5
    label = torch.where(torch.rand(n, generator=g) < obj_frac, torch.randint(0, k, (n,), generator=g), torch.full((n,), bg))
    y = torch.cat((label.float().view(-1, 1), torch.randn(n, w, generator=g) * 1.5), 1)
    return cls, bb, y


@pytest.mark.parametrize("n,k,w,bg,weights,alpha,beta", [(700, 6, 5, 5, [0.5, 2.0, 1.0, 3.0, 0.7, 0.1], 1.0, 1.0),
                                                        (300, 11, 5, 10, None, 0.3, 2.5), (257, 6, 4, 5, [1.0] * 6, 1.0, 0.5),
                                                        (1, 6, 5, 5, None, 1.0, 1.0)])
def test_loss_and_gradients_match_the_reference_formulation(L, n, k, w, bg, weights, alpha, beta):
    cls, bb, y = make(n, k, w, bg, n + k)
    c64 = cls.double().requires_grad_(True); b64 = bb.double().requires_grad_(True)
    ref, ref_c, ref_b = loss_oracle.detection_loss(c64, b64, y.double(), bg, weights, alpha, beta)
    ref.backward()
    cg = cls.cuda().requires_grad_(True); bg_ = bb.cuda().requires_grad_(True)
    loss, lc, lb = L.detection_loss(cg, bg_, y.cuda(), bg, weights, alpha, beta)
    assert abs(float(loss.detach()) - float(ref.detach())) <= 2e-6 * max(1.0, abs(float(ref.detach())))
    assert abs(float(lc) - float(ref_c)) <= 2e-6 * max(1.0, abs(float(ref_c)))
    assert abs(float(lb) - float(ref_b)) <= 2e-6 * max(1.0, abs(float(ref_b)))
    loss.backward()
    for got, exp in ((cg.grad, c64.grad), (bg_.grad, b64.grad if b64.grad is not None else torch.zeros_like(b64))):
        err = (got.double().cpu() - exp).abs().max() / max(float(exp.abs().max()), 1e-12)
        assert float(err) < 2e-5, float(err)


def test_no_objects_nan_boxes_and_scaled_upstream_gradient(L):
    cls, bb, y = make(500, 6, 5, 5, 3, obj_frac=0.0)             # background only: loss_bb = 0, no box gradient
    cg = cls.cuda().requires_grad_(True); bg_ = bb.cuda().requires_grad_(True)
    loss, lc, lb = L.detection_loss(cg, bg_, y.cuda(), 5)
    ref, ref_c, _ = loss_oracle.detection_loss(cls.double(), bb.double(), y.double(), 5)
    assert float(lb) == 0.0 and abs(float(loss) - float(ref)) < 2e-6
    (3.0 * loss).backward()
    assert float(bg_.grad.abs().max()) == 0.0
    c64 = cls.double().requires_grad_(True)
    (3.0 * loss_oracle.detection_loss(c64, bb.double(), y.double(), 5)[0]).backward()
    assert float((cg.grad.double().cpu() - c64.grad).abs().max()) < 1e-7
    cls, bb, y = make(400, 6, 5, 5, 4)                            # a NaN in one object box: the batch's box loss is ignored
    bb[int((y[:, 0] != 5).nonzero()[0]), 2] = float("nan")
    cg = cls.cuda().requires_grad_(True); bg_ = bb.cuda().requires_grad_(True)
    loss, lc, lb = L.detection_loss(cg, bg_, y.cuda(), 5)
    ref = loss_oracle.detection_loss(cls.double(), bb.double(), y.double(), 5)[0]
    assert float(lb) == 0.0 and abs(float(loss) - float(ref)) < 2e-6
    loss.backward()
    assert float(torch.nan_to_num(bg_.grad).abs().max()) == 0.0 and torch.isfinite(cg.grad).all()


def test_training_step_through_model_and_loss(L):
    """forward -> detection_loss -> backward -> Adam on the HIP path: the loss goes down (trainer.py:176-231 in five lines)."""
    from radargnn_amd import gnn
    cfg = gnn.GNNArchitectureConfig(node_feature_dimension=5, edge_feature_dimension=2, conv_layer_dimensions=[16, 8],
                                    classification_head_layer_dimensions=[6], regression_head_layer_dimensions=[8, 5],
                                    initial_node_feature_embedding=True, initial_edge_feature_embedding=True,
                                    node_feature_embedding_layer_dimensions=[8, 16], edge_feature_embedding_layer_dimensions=[4, 8],
                                    conv_layer_type="MPNNConv", batch_norm_in_mlps=False)
    torch.manual_seed(0)
    model = gnn.DetNetBasic(cfg).cuda()
    g = torch.Generator().manual_seed(1)
    n, e = 400, 2400
    x = torch.randn(n, 5, generator=g).cuda(); ei = torch.randint(0, n, (2, e), generator=g).cuda()
    ea = torch.randn(e, 2, generator=g).cuda()
    _, _, y = make(n, 6, 5, 5, 9)
    opt = torch.optim.Adam(model.parameters(), lr=5e-3)
    losses = []
    for _ in range(15):
        opt.zero_grad()
        c, b = model(x, ei, ea)
        loss, _, _ = L.detection_loss(c, b, y.cuda(), 5, [1.0, 1.0, 1.0, 1.0, 1.0, 0.2])
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert losses[-1] < 0.8 * losses[0], losses
