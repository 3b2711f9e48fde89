"""Loading what the reference's trainer wrote (SURVEY.md sections 5 / 7.3: the whole-module pickle is part of the ABI).

``Trainer.save_results`` (gnn/trainer.py:342-354) saves the trained model twice: ``trained_model_state_dict.pt`` -- loaded with
``DetNetBasic(cfg).load_state_dict(...)``, the key contract of SURVEY section 8(b) -- and ``trained_model.pt``, a pickle of the WHOLE
module, which ``evaluate.py:46-52`` reads back with ``torch.load``.  That pickle names its classes by import path:

    gnnradarobjectdetection.gnn.gnn_models.DetNetBasic, gnnradarobjectdetection.gnn.mpnn_layers.MPNNConv / RadarPointGNNConv,
    torch_geometric.nn.dense.linear.Linear, torch_geometric.nn.norm.batch_norm.BatchNorm, torch_geometric.nn.aggr.basic.*,
    torch_geometric.nn.conv.utils.inspector.Inspector, torch.nn containers

so loading it needs those paths importable -- in the reference's environment: torch_geometric 2.1.  Two ways here, neither needs
torch_geometric:

* ``load_reference_model(path)`` decodes the pickle STRUCTURALLY: every class that is not torch's own is replaced by an opaque
  stand-in while unpickling (the pattern of ``data.load_graph``), the parameter / buffer tensors are collected by walking
  ``_modules`` / ``_parameters`` / ``_buffers`` (which is all a ``state_dict`` is), the architecture is read off the attributes
  ``DetNetBasic.__init__`` stores (gnn/gnn_models.py:22-40) and the tensor shapes, and a fresh HIP ``DetNetBasic`` is built and
  loaded strictly.  Nothing of the pickled objects' code or state survives except tensors, ints, bools and strings.
* ``install_reference_pickle_shims()`` registers stand-in MODULES under the torch_geometric paths (only when the real package cannot
  be imported) whose classes are the HIP ``Linear`` / ``BatchNorm`` plus parameter-free stand-ins for the aggregation modules and the
  inspector, so that the reference's own ``torch.load(".../trained_model.pt")`` resolves every name.  Opt-in, never at import
  time: ``with reference_pickle_shims(): ...``, ``gnnradarobjectdetection.enable_reference_pickles()`` or RGNN_REFERENCE_PICKLES=1.  The unpickled objects are then the HIP classes carrying the pickled ``__dict__``; their
  ``__setstate__`` fills in what the HIP classes keep beside the reference's attributes.

UNPINNED in this image: there is no torch_geometric here to write a real ``trained_model.pt`` with; the tests build the pickle
against stand-in classes registered under the real class paths with the attribute layout of torch_geometric 2.1.0
(tests/test_checkpoint_host.py), the same caveat as the ``graph_*.pt`` reader carries.
"""
from __future__ import annotations

import pickle
import sys
import types
from typing import Dict, List, Optional

import torch
from torch import nn

FOREIGN_ROOTS = ("torch_geometric", "gnnradarobjectdetection", "torch_scatter", "torch_sparse", "torch_cluster")


class _Opaque:
    """Stand-in for a class of the reference / torch_geometric met while unpickling: keeps whatever state the pickle carries
    (``nn.Module`` subclasses pickle their ``__dict__``: ``_parameters``, ``_buffers``, ``_modules``, plain attributes)."""

    def __init__(self, *args, **kwargs):
        pass

    def __setstate__(self, state):
        if isinstance(state, dict):
            self.__dict__.update(state)
        else:
            self.__dict__["_state"] = state

    # (bound methods stored in hook dictionaries are pickled as getattr(obj, name): any name resolves to a no-op)
    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return _noop


def _noop(*args, **kwargs):
    return None


# What a whole-module pickle of a torch model legitimately names besides the foreign classes: torch itself, the containers' helpers
# and this package.  Anything else is refused -- ``weights_only=False`` unpickling executes whatever global a file names, and a
# ``trained_model.pt`` from an untrusted source must not get further than this list.
_ALLOWED_ROOTS = ("torch", "collections", "numpy", "radargnn_amd", "_codecs", "functools")
# (torch_geometric's Inspector keeps the signatures of message / aggregate / update: inspect.Parameter objects and their kinds)
_ALLOWED_GLOBALS = {("inspect", "Parameter"), ("inspect", "_ParameterKind"), ("inspect", "_empty"), ("inspect", "Signature"),
                    ("typing", "Any"), ("typing", "Optional"), ("typing", "Union"), ("typing", "Tuple"), ("typing", "List")}
_ALLOWED_BUILTINS = {"set", "frozenset", "dict", "list", "tuple", "slice", "range", "complex", "bytearray", "getattr", "int", "float",
                     "bool", "str", "bytes", "object"}


class _ModelUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        root = module.split(".")[0]
        if root in FOREIGN_ROOTS:
            return type(name, (_Opaque,), {"__module__": module, "_rgnn_path": f"{module}.{name}"})
        if root in _ALLOWED_ROOTS or (module, name) in _ALLOWED_GLOBALS or (module in ("builtins", "__builtin__") and name in _ALLOWED_BUILTINS):
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f"load_reference_model: the file names {module}.{name}, which a trained_model.pt has no reason to "
                                     "(allowed: torch, numpy, collections, this package, the reference's and torch_geometric's class paths)")


_PICKLE = types.SimpleNamespace(__name__="pickle", Unpickler=_ModelUnpickler, load=lambda f, **kw: _ModelUnpickler(f, **kw).load(),
                                loads=pickle.loads, dump=pickle.dump, dumps=pickle.dumps, Pickler=pickle.Pickler)


def _children(obj) -> Dict[str, object]:
    mods = getattr(obj, "__dict__", {}).get("_modules")
    return dict(mods) if mods else {}


def _collect_state(obj, prefix: str, out: Dict[str, torch.Tensor]) -> None:
    """What ``nn.Module.state_dict`` would return, from the pickled structure alone."""
    d = getattr(obj, "__dict__", {})
    for name, p in (d.get("_parameters") or {}).items():
        if p is not None:
            out[prefix + name] = p.detach() if torch.is_tensor(p) else p
    skip = d.get("_non_persistent_buffers_set") or set()
    for name, b in (d.get("_buffers") or {}).items():
        if b is not None and name not in skip:
            out[prefix + name] = b
    for name, child in _children(obj).items():
        if child is not None:
            _collect_state(child, f"{prefix}{name}.", out)


def _class_name(obj) -> str:
    return getattr(type(obj), "_rgnn_path", f"{type(obj).__module__}.{type(obj).__name__}").rsplit(".", 1)[-1]


def _linear_widths(sd: Dict[str, torch.Tensor], prefix: str) -> List[int]:
    """Output widths of the Linear layers of the Sequential under ``prefix`` (get_mlp: Linear, [BatchNorm], ReLU, Linear, ...)."""
    idx = sorted({int(k[len(prefix):].split(".")[0]) for k in sd if k.startswith(prefix) and k.endswith(".weight")
                  and sd[k].dim() == 2})
    return [int(sd[f"{prefix}{i}.weight"].shape[0]) for i in idx]


def config_from_reference_module(obj, sd: Dict[str, torch.Tensor]):
    """The ``GNNArchitectureConfig`` that builds the pickled ``DetNetBasic``: from the attributes its constructor stores
    (gnn/gnn_models.py:22-40: ``node_feat_dim`` / ``edge_feat_dim`` are overwritten by the embedding widths there, so the input
    widths come from the first weights) and the shapes of the tensors."""
    from .gnn.configs import GNNArchitectureConfig
    d = obj.__dict__
    convs = list(_children(_children(obj).get("convs")).values()) if _children(obj).get("convs") is not None else []
    if not convs:
        raise ValueError("the pickled module has no `convs`: not a DetNetBasic")
    conv_type = _class_name(convs[0])
    if conv_type not in ("MPNNConv", "RadarPointGNNConv"):
        raise ValueError(f"unknown conv layer class {conv_type!r}")
    node_emb = bool(d.get("initial_node_feature_embedding", any(k.startswith("node_emb_mlp.") for k in sd)))
    edge_emb = bool(d.get("initial_edge_feature_embedding", any(k.startswith("edge_emb_mlp.") for k in sd)))
    bn_mlps = bool(d.get("batch_norm_mlps", False))
    first_lin = lambda prefix: sd[min((k for k in sd if k.startswith(prefix) and k.endswith(".weight") and sd[k].dim() == 2),
                                      key=lambda k: int(k[len(prefix):].split(".")[0]))]
    node_dims = _linear_widths(sd, "node_emb_mlp.") if node_emb else []
    edge_dims = _linear_widths(sd, "edge_emb_mlp.") if edge_emb else []
    pre0 = sd["convs.0.pre_mlp.0.weight"]
    use_enc = bool(d.get("conv_use_edge_encoder", "convs.0.edge_encoder.weight" in sd))
    if node_emb:
        node_in = int(first_lin("node_emb_mlp.").shape[1])
    else:                                                       # width of x entering the first conv
        node_in = int(d["node_feat_dim"])
    if edge_emb:
        edge_in = int(first_lin("edge_emb_mlp.").shape[1])
    else:
        edge_in = int(d["edge_feat_dim"])
    conv_dims = [int(v) for v in d["conv_layer_dimensions"]]
    cfg = GNNArchitectureConfig(
        node_feature_dimension=node_in, edge_feature_dimension=edge_in, conv_layer_dimensions=conv_dims,
        classification_head_layer_dimensions=_linear_widths(sd, "classification_head."),
        regression_head_layer_dimensions=_linear_widths(sd, "regression_head."),
        initial_node_feature_embedding=node_emb, initial_edge_feature_embedding=edge_emb,
        node_feature_embedding_layer_dimensions=node_dims or [node_in], edge_feature_embedding_layer_dimensions=edge_dims or [edge_in],
        conv_layer_type=conv_type, batch_norm_in_mlps=bn_mlps,
        conv_pre_mlp_layer_number=int(d.get("conv_pre_mlp_layers", len(_linear_widths(sd, "convs.0.pre_mlp.")))),
        conv_post_mlp_layer_number=int(d.get("conv_post_mlp_layers", len(_linear_widths(sd, "convs.0.post_mlp.")))),
        conv_use_edge_encoder=use_enc, aggregation_function=str(d.get("aggregation", "max")))
    del pre0
    return cfg


def load_reference_model(path: str, map_location="cpu"):
    """``trained_model.pt`` as gnn/trainer.py:342-344 writes it (``torch.save(model)`` of the reference's ``DetNetBasic`` on
    torch_geometric layers) -> the HIP ``DetNetBasic`` with the same weights, buffers (running statistics, batch counters) and
    training flag.  Also reads whole-module pickles of THIS package's ``DetNetBasic`` and plain ``state_dict`` files next to a
    whole-module pickle are not needed.  torch_geometric is not required."""
    from .gnn.gnn_models import DetNetBasic
    obj = torch.load(path, map_location=map_location, pickle_module=_PICKLE, weights_only=False)
    if isinstance(obj, DetNetBasic):
        return obj
    if isinstance(obj, dict):
        raise ValueError(f"{path} holds a dictionary (a state_dict?): build DetNetBasic(config) and call load_state_dict on it")
    sd: Dict[str, torch.Tensor] = {}
    _collect_state(obj, "", sd)
    if not sd:
        raise ValueError(f"{path}: no parameters found in the pickled object ({type(obj).__name__})")
    cfg = config_from_reference_module(obj, sd)
    model = DetNetBasic(cfg)
    model.load_state_dict({k: v.to("cpu") for k, v in sd.items()}, strict=True)
    model.train(bool(obj.__dict__.get("training", True)))
    if map_location not in (None, "cpu") and str(map_location) != "cpu":
        model.to(map_location)
    return model


# ---- stand-in modules for the reference's own torch.load -------------------------------------------------------------------------
class _ParameterFree(nn.Module):
    """torch_geometric.nn.aggr.* as pickled inside a MessagePassing layer (``aggr_module``): no parameters, never called here."""

    def forward(self, *args, **kwargs):
        raise RuntimeError("a torch_geometric stand-in module was called: the HIP layers do their own aggregation")


class _OpaqueObject(_Opaque):
    pass


# Classes that torch_geometric 2.1 pickles inside a MessagePassing layer besides Linear / BatchNorm / the aggregation modules: plain
# objects (no parameters) -- named explicitly; a pickle that names anything else under these paths fails with the missing name
# (a catch-all module __getattr__ makes hasattr(torch_geometric, anything) true for the whole process)
_OPAQUE_NAMES = {
    "torch_geometric.nn.conv.utils.inspector": ("Inspector",),
    "torch_geometric.nn.conv.message_passing": ("MessagePassing",),
    "torch_geometric.nn.conv.utils": ("Inspector",),
    "torch_geometric.nn.inits": (),
}


def install_reference_pickle_shims(force: bool = False) -> bool:
    """Make the class paths a reference ``trained_model.pt`` names importable without torch_geometric (no-op when the real package
    imports; ``force`` is for tests).  Returns whether stand-ins were installed.

    NOT done at import time: while installed, ``import torch_geometric`` succeeds with a stand-in, so this is an explicit
    opt-in -- ``with reference_pickle_shims(): torch.load(...)``, ``gnnradarobjectdetection.enable_reference_pickles()`` for scripts
    that call ``torch.load`` themselves (evaluate.py:46-52), or RGNN_REFERENCE_PICKLES=1 in the environment of an unchanged script.
    The stand-in modules carry a real ``ModuleSpec`` (``importlib.util.find_spec`` works) and an explicit list of names."""
    if not force:
        try:
            import torch_geometric  # noqa: F401
            if not getattr(torch_geometric, "_rgnn_stand_in", False):
                return False
        except Exception:
            pass
    if getattr(sys.modules.get("torch_geometric"), "_rgnn_stand_in", False):
        return True
    import importlib.machinery
    from .gnn.linear import BatchNorm, Linear
    names = ["torch_geometric", "torch_geometric.nn", "torch_geometric.nn.dense", "torch_geometric.nn.dense.linear",
             "torch_geometric.nn.norm", "torch_geometric.nn.norm.batch_norm", "torch_geometric.nn.aggr", "torch_geometric.nn.aggr.basic",
             "torch_geometric.nn.aggr.base", "torch_geometric.nn.conv", "torch_geometric.nn.conv.message_passing",
             "torch_geometric.nn.conv.utils", "torch_geometric.nn.conv.utils.inspector", "torch_geometric.nn.inits"]
    packages = {n for n in names if any(o.startswith(n + ".") for o in names)}

    def module(name, **attrs):
        m = types.ModuleType(name, "stand-in installed by radargnn_amd.checkpoint.install_reference_pickle_shims (no torch_geometric in this environment)")
        m._rgnn_stand_in = True
        m.__spec__ = importlib.machinery.ModuleSpec(name, loader=None, is_package=name in packages)
        if name in packages:
            m.__path__ = []
        m.__dict__.update(attrs)
        for attr in _OPAQUE_NAMES.get(name, ()):
            setattr(m, attr, type(attr, (_OpaqueObject,), {"__module__": name}))
        sys.modules[name] = m
        parent, _, leaf = name.rpartition(".")
        if parent in sys.modules:
            setattr(sys.modules[parent], leaf, m)
        return m

    aggr = {n: type(n, (_ParameterFree,), {"__module__": "torch_geometric.nn.aggr.basic"})
            for n in ("MaxAggregation", "MeanAggregation", "SumAggregation", "MinAggregation", "MulAggregation", "VarAggregation",
                      "StdAggregation", "SoftmaxAggregation", "PowerMeanAggregation")}
    attrs = {"torch_geometric.nn": dict(Linear=Linear, BatchNorm=BatchNorm), "torch_geometric.nn.dense.linear": dict(Linear=Linear),
             "torch_geometric.nn.norm": dict(BatchNorm=BatchNorm), "torch_geometric.nn.norm.batch_norm": dict(BatchNorm=BatchNorm),
             "torch_geometric.nn.aggr": aggr, "torch_geometric.nn.aggr.basic": aggr,
             "torch_geometric.nn.aggr.base": dict(Aggregation=type("Aggregation", (_ParameterFree,), {"__module__": "torch_geometric.nn.aggr.base"}))}
    for n in names:
        module(n, **attrs.get(n, {}))
    return True


class reference_pickle_shims:
    """``with reference_pickle_shims(): model = torch.load(".../trained_model.pt", weights_only=False)`` -- the stand-ins exist for
    the duration of the block only (and not at all when the real torch_geometric imports)."""

    def __enter__(self):
        self._mine = not getattr(sys.modules.get("torch_geometric"), "_rgnn_stand_in", False) and install_reference_pickle_shims()
        return self

    def __exit__(self, *exc):
        if self._mine:
            remove_reference_pickle_shims()
        return False


def remove_reference_pickle_shims() -> None:
    for name in [n for n, m in sys.modules.items() if n.split(".")[0] == "torch_geometric" and getattr(m, "_rgnn_stand_in", False)]:
        del sys.modules[name]
