"""Whole-module pickles of the reference's trainer (gnn/trainer.py:342-354 ``torch.save(self.model, ".../trained_model.pt")``, read back by
evaluate.py:46-52) without torch_geometric -- radargnn_amd/checkpoint.py.  CPU only.

torch_geometric cannot be imported in this image, so the pickle is HAND-BUILT: stand-in classes are registered under the real class
paths with the attribute layout of torch_geometric 2.1.0 / the reference (``MessagePassing``: aggr, aggr_module, flow, node_dim,
inspector, __user_args__, hook dictionaries ...; ``Linear``: in/out_channels, initialisers, a state_dict pre-hook stored as a BOUND
METHOD; ``BatchNorm``: nothing but ``module``), a model is assembled from them with random weights, saved with ``torch.save`` and the
stand-ins are removed again before loading.  PARITY UNPINNED against a file written by the real package."""
import collections
import copy
import inspect
import io
import sys
import types

import pytest
import torch
from torch import nn

from radargnn_amd import checkpoint, gnn


# ---- the stand-in "reference environment" ------------------------------------------------------------------------------------------
def _install_fake_reference():
    names = ("torch_geometric", "torch_geometric.nn", "torch_geometric.nn.dense", "torch_geometric.nn.dense.linear",
             "torch_geometric.nn.norm", "torch_geometric.nn.norm.batch_norm", "torch_geometric.nn.aggr", "torch_geometric.nn.aggr.basic",
             "torch_geometric.nn.conv", "torch_geometric.nn.conv.utils", "torch_geometric.nn.conv.utils.inspector",
             "fakeref", "fakeref.gnn", "fakeref.gnn.gnn_models", "fakeref.gnn.mpnn_layers")
    saved = {n: sys.modules.get(n) for n in names}
    mods = {n: types.ModuleType(n) for n in names}

    def cls(module, name, bases, body):
        c = type(name, bases, dict(body, __module__=module))
        setattr(mods[module], name, c)
        return c

    class _LinearBody(nn.Module):
        def __init__(self, in_channels, out_channels, bias=True):
            super().__init__()
            self.in_channels, self.out_channels = in_channels, out_channels
            self.weight_initializer, self.bias_initializer = None, None
            self.weight = nn.Parameter(torch.randn(out_channels, in_channels) / in_channels ** 0.5)
            self.bias = nn.Parameter(torch.randn(out_channels) * 0.1) if bias else None
            # torch_geometric 2.1: self._load_hook = self._register_load_state_dict_pre_hook(self._lazy_load_hook)
            self._load_hook = self._register_load_state_dict_pre_hook(self._lazy_load_hook)

        def _lazy_load_hook(self, *a, **k):
            pass

    Linear = cls("torch_geometric.nn.dense.linear", "Linear", (_LinearBody,), {})

    class _BNBody(nn.Module):
        def __init__(self, in_channels):
            super().__init__()
            self.module = nn.BatchNorm1d(in_channels)

    BatchNorm = cls("torch_geometric.nn.norm.batch_norm", "BatchNorm", (_BNBody,), {})
    MaxAggregation = cls("torch_geometric.nn.aggr.basic", "MaxAggregation", (nn.Module,), {})
    MeanAggregation = cls("torch_geometric.nn.aggr.basic", "MeanAggregation", (nn.Module,), {})
    Inspector = cls("torch_geometric.nn.conv.utils.inspector", "Inspector", (object,), {})

    class _MP(nn.Module):
        def __init__(self, aggr):
            super().__init__()
            self.aggr = aggr
            self.aggr_module = MaxAggregation() if aggr == "max" else MeanAggregation()
            self.flow, self.node_dim, self.decomposed_layers = "source_to_target", -2, 1
            insp = Inspector()
            insp.base_class = self
            insp.params = {"message": collections.OrderedDict(x_i=inspect.Parameter("x_i", inspect.Parameter.POSITIONAL_OR_KEYWORD))}
            self.inspector = insp
            self.__user_args__, self.__fused_user_args__, self.fuse = {"x_i", "x_j", "edge_attr"}, set(), False
            self._explain, self._edge_mask, self._loop_mask, self._apply_sigmoid = False, None, None, True
            for h in ("_propagate_forward_pre_hooks", "_propagate_forward_hooks", "_message_forward_pre_hooks", "_message_forward_hooks",
                      "_aggregate_forward_pre_hooks", "_aggregate_forward_hooks"):
                setattr(self, h, collections.OrderedDict())

    class _MPNNBody(_MP):
        def __init__(self, in_channels, out_channels, edge_dim, aggr="max", pre_layers=1, post_layers=1, use_edge_encoder=False):
            super().__init__(aggr)
            self.in_channels, self.out_channels, self.edge_dim, self.use_edge_encoder = in_channels, out_channels, edge_dim, use_edge_encoder
            if use_edge_encoder:
                self.edge_encoder = Linear(edge_dim, in_channels)
                d = 3 * in_channels
            else:
                d = 2 * in_channels + edge_dim
            m = [Linear(d, d)]
            for _ in range(pre_layers - 1):
                m += [nn.ReLU(), Linear(d, d)]
            self.pre_mlp = nn.Sequential(*m)
            m = [Linear(d + in_channels, out_channels)]
            for _ in range(post_layers - 1):
                m += [nn.ReLU(), Linear(out_channels, out_channels)]
            self.post_mlp = nn.Sequential(*m)

    MPNNConv = cls("fakeref.gnn.mpnn_layers", "MPNNConv", (_MPNNBody,), {})

    def get_mlp(i, o, hidden, bn):
        if not hidden:
            return nn.Sequential(Linear(i, o))
        m = [Linear(i, hidden[0])]
        for a, b in zip(hidden, hidden[1:] + [o]):
            if bn:
                m.append(BatchNorm(a))
            m += [nn.ReLU(), Linear(a, b)]
        return nn.Sequential(*m)

    class _DetBody(nn.Module):
        def __init__(self, cfg):
            super().__init__()
            self.batch_norm_mlps = cfg.batch_norm_in_mlps
            self.node_feat_dim, self.edge_feat_dim = cfg.node_feature_dimension, cfg.edge_feature_dimension
            self.conv_layer_dimensions = cfg.conv_layer_dimensions
            self.initial_node_feature_embedding = cfg.initial_node_feature_embedding
            self.initial_edge_feature_embedding = cfg.initial_edge_feature_embedding
            self.conv_pre_mlp_layers, self.conv_post_mlp_layers = cfg.conv_pre_mlp_layer_number, cfg.conv_post_mlp_layer_number
            self.conv_use_edge_encoder, self.aggregation = cfg.conv_use_edge_encoder, cfg.aggregation_function
            if cfg.initial_node_feature_embedding:
                dims = cfg.node_feature_embedding_layer_dimensions
                self.node_emb_mlp = get_mlp(self.node_feat_dim, dims[-1], dims[:-1], self.batch_norm_mlps)
                self.node_feat_dim = dims[-1]
            if cfg.initial_edge_feature_embedding:
                dims = cfg.edge_feature_embedding_layer_dimensions
                self.edge_emb_mlp = get_mlp(self.edge_feat_dim, dims[-1], dims[:-1], self.batch_norm_mlps)
                self.edge_feat_dim = dims[-1]
            self.convs, self.batch_norms = nn.ModuleList(), nn.ModuleList()
            w = self.node_feat_dim
            for width in cfg.conv_layer_dimensions:
                self.convs.append(MPNNConv(w, width, self.edge_feat_dim, aggr=self.aggregation, pre_layers=self.conv_pre_mlp_layers,
                                           post_layers=self.conv_post_mlp_layers, use_edge_encoder=self.conv_use_edge_encoder))
                self.batch_norms.append(BatchNorm(width))
                w = width
            dims = cfg.classification_head_layer_dimensions
            self.classification_head = get_mlp(w, dims[-1], dims[:-1], self.batch_norm_mlps)
            dims = cfg.regression_head_layer_dimensions
            self.regression_head = get_mlp(w, dims[-1], dims[:-1], self.batch_norm_mlps)

    DetNetBasic = cls("fakeref.gnn.gnn_models", "DetNetBasic", (_DetBody,), {})
    sys.modules.update(mods)

    def restore():
        for n, m in saved.items():
            if m is None:
                sys.modules.pop(n, None)
            else:
                sys.modules[n] = m
    return DetNetBasic, restore


def _save_fake_reference_model(cfg, path, train=True):
    """torch.save(model) in the stand-in environment, pickled under the reference's REAL class paths
    (gnnradarobjectdetection.gnn.gnn_models.DetNetBasic, ...mpnn_layers.MPNNConv, torch_geometric.nn...)."""
    Det, restore = _install_fake_reference()
    keep = {}
    try:
        import fakeref.gnn.gnn_models as fm, fakeref.gnn.mpnn_layers as fl
        real = {"gnnradarobjectdetection.gnn.gnn_models": fm, "gnnradarobjectdetection.gnn.mpnn_layers": fl}
        keep = {n: sys.modules.get(n) for n in real}
        fm.DetNetBasic.__module__ = "gnnradarobjectdetection.gnn.gnn_models"
        fl.MPNNConv.__module__ = "gnnradarobjectdetection.gnn.mpnn_layers"
        sys.modules.update(real)
        torch.manual_seed(3)
        model = Det(cfg)
        model.train(train)
        with torch.no_grad():
            for bn in model.batch_norms:
                bn.module.running_mean.normal_()
                bn.module.running_var.uniform_(0.5, 2.0)
                bn.module.num_batches_tracked.fill_(7)
        sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
        torch.save(model, str(path))
        data = open(str(path), "rb").read()
        # (the layout the loaders must survive: PyG class paths, a bound method in a hook dictionary, the aggregation module)
        assert b"torch_geometric" in data and b"_lazy_load_hook" in data and b"MaxAggregation" in data or b"MeanAggregation" in data
    finally:
        for n, m in keep.items():
            if m is None:
                sys.modules.pop(n, None)
            else:
                sys.modules[n] = m
        restore()
    return sd


CFGS = [gnn.GNNArchitectureConfig(5, 2, [32, 24], [6], [16, 5], True, True, [16, 32], [4, 8], "MPNNConv", False),
        gnn.GNNArchitectureConfig(4, 3, [24], [11], [8, 4], False, False, [4], [3], "MPNNConv", True, 2, 2, True, "mean")]


@pytest.mark.parametrize("cfg", CFGS)
def test_load_reference_model_from_a_stand_in_pickle(cfg, tmp_path):
    path = tmp_path / "trained_model.pt"
    sd = _save_fake_reference_model(cfg, path, train=False)
    checkpoint.remove_reference_pickle_shims()                   # (the structural loader needs no module under the pickled paths)
    assert "fakeref" not in sys.modules
    model = checkpoint.load_reference_model(str(path))
    assert isinstance(model, gnn.DetNetBasic) and model.training is False
    got = model.state_dict()
    assert set(got) == set(sd)
    for k, v in sd.items():
        assert torch.equal(got[k], v), k
    assert int(model.batch_norms[0].module.num_batches_tracked) == 7
    ref = gnn.DetNetBasic(cfg)
    assert [tuple(p.shape) for p in model.parameters()] == [tuple(p.shape) for p in ref.parameters()]
    assert type(model.convs[0]).__name__ == cfg.conv_layer_type and model.aggregation == cfg.aggregation_function
    assert model.conv_pre_mlp_layers == cfg.conv_pre_mlp_layer_number and model.batch_norm_mlps == cfg.batch_norm_in_mlps
    assert model.conv_use_edge_encoder == cfg.conv_use_edge_encoder


def test_the_reference_own_torch_load_resolves_through_the_shims(tmp_path):
    """evaluate.py:46-52 calls ``torch.load(".../trained_model.pt")`` itself.  With the shim package imported the torch_geometric
    paths resolve to the HIP ``Linear`` / ``BatchNorm`` and parameter-free stand-ins, and the reference's own class paths to the HIP
    ``DetNetBasic`` / ``MPNNConv``: the unpickled object IS a HIP model carrying the pickled weights."""
    cfg = CFGS[0]
    path = tmp_path / "trained_model.pt"
    sd = _save_fake_reference_model(cfg, path)
    for n in [n for n in sys.modules if n.split(".")[0] == "gnnradarobjectdetection"]:
        del sys.modules[n]
    import importlib.util
    import gnnradarobjectdetection
    assert not getattr(sys.modules.get("torch_geometric"), "_rgnn_stand_in", False)      # importing the shim package installs nothing
    assert gnnradarobjectdetection.enable_reference_pickles()                             # the explicit opt-in (no real package here)
    tg = sys.modules["torch_geometric"]
    assert importlib.util.find_spec("torch_geometric") is not None and not hasattr(tg, "no_such_name")
    try:
        loaded = torch.load(str(path), map_location="cpu", weights_only=False)     # (torch >= 2.6 defaults to weights_only=True)
        assert isinstance(loaded, gnn.DetNetBasic) and isinstance(loaded.convs[0], gnn.MPNNConv)
        assert isinstance(loaded.node_emb_mlp[0], gnn.Linear) and isinstance(loaded.batch_norms[0], gnn.BatchNorm)
        assert loaded.batch_norms[0].in_channels == cfg.conv_layer_dimensions[0]
        got = loaded.state_dict()
        assert set(got) == set(sd) and all(torch.equal(got[k], v) for k, v in sd.items())
        fresh = gnn.DetNetBasic(cfg)
        fresh.load_state_dict(got)                                              # the key contract both ways
    finally:
        checkpoint.remove_reference_pickle_shims()


def test_whole_module_pickle_of_the_hip_model_carries_no_caches():
    """trainer.py:128-130 deep-copies the best model, :342-354 pickles it: the weight-derived caches an inference pass leaves on the
    instances (folded weights, negated / concatenated weights) stay out of both."""
    cfg = CFGS[0]
    model = gnn.DetNetBasic(cfg)
    model.__dict__["_neg_cache"] = ("key", torch.zeros(3), None)
    model._heads_val, model._heads_key = (torch.zeros(2), torch.zeros(2)), "k"
    model.convs[0]._fold_val, model.convs[0]._fold_key = (torch.zeros(4), torch.zeros(4)), "k"
    buf = io.BytesIO()
    torch.save(model, buf)
    buf.seek(0)
    back = torch.load(buf, weights_only=False)
    for m in (back, copy.deepcopy(model)):
        assert "_neg_cache" not in m.__dict__ and not hasattr(m, "_heads_val") and not hasattr(m.convs[0], "_fold_val")
        assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), model.state_dict().values()))
    assert set(back.state_dict()) == set(model.state_dict())


def test_load_reference_model_refuses_globals_a_model_file_has_no_reason_to_name(tmp_path):
    """weights_only=False unpickling runs whatever a file names: the structural loader resolves torch, numpy, collections, this
    package and (as opaque stand-ins) the reference's / torch_geometric's class paths -- nothing else."""
    import os
    import pickle

    class Evil:
        def __reduce__(self):
            return (os.system, ("true",))
    path = tmp_path / "trained_model.pt"
    torch.save(Evil(), str(path))
    with pytest.raises(pickle.UnpicklingError, match="no reason"):
        checkpoint.load_reference_model(str(path))


def test_reference_pickle_shims_context_manager_leaves_nothing_behind():
    import importlib.util
    assert "torch_geometric" not in sys.modules or not getattr(sys.modules["torch_geometric"], "_rgnn_stand_in", False)
    with checkpoint.reference_pickle_shims():
        import torch_geometric.nn.dense.linear as tl
        assert tl.Linear is gnn.Linear and importlib.util.find_spec("torch_geometric.nn") is not None
    assert not [n for n in sys.modules if n.split(".")[0] == "torch_geometric"]
