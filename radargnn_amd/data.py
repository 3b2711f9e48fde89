"""Graph containers and batching -- the device-resident counterpart of what the reference gets from torch_geometric
(SURVEY §8f row 2): ``Data`` (one graph as written by ``create_graph_data``,
preprocessor/radarscenes/dataset_creation.py:786-814 / nuscenes/dataset_creation.py:280-308), ``Batch`` (PyG's
``Batch.from_data_list`` numbering), ``DataLoader`` (utils/data_handling.py:30) and ``get_data_loaders``
(utils/data_handling.py:7-36).

MI355X-first difference: the reference keeps the graphs on the host, collates each batch on the host and copies it to
the device (postprocessor/inference.py:57).  Here the whole split is concatenated into HBM ONCE (``GraphStore``; a
RadarScenes split is a few GB, the device has 288) and a batch is built on the device by two segmented-copy kernels
(csrc/collate.hip) from a list of graph ids -- per batch the host only uploads one small offset table.

Collation rules reproduced (torch_geometric 2.1 ``collate``): every tensor attribute is concatenated along dim 0, except
``edge_index`` which is concatenated along dim 1 with the cumulative node count added; ``batch`` holds the graph slot of
every node, ``ptr`` the node offsets; ``num_nodes`` of a graph is ``x.shape[0]``.
"""
from __future__ import annotations

import glob
import json
import pickle
import types
from typing import Dict, Iterable, Iterator, List, Optional, Sequence

import numpy as np
import torch

from . import ops

FORMAT_TAG = "radargnn_amd.graph/1"


class Data:
    """Attribute bag with the ``torch_geometric.data.Data`` surface the reference touches: keyword construction,
    attribute access, ``keys``, ``to(device)``, ``num_nodes`` / ``num_edges`` / ``num_node_features``."""

    def __init__(self, x=None, edge_index=None, edge_attr=None, y=None, pos=None, **kwargs):
        self._keys: List[str] = []
        for k, v in (("x", x), ("edge_index", edge_index), ("edge_attr", edge_attr), ("y", y), ("pos", pos)):
            if v is not None:
                setattr(self, k, v)
        for k, v in kwargs.items():
            if v is not None:
                setattr(self, k, v)

    def __setattr__(self, key, value):
        if not key.startswith("_") and key not in self._keys:
            self._keys.append(key)
        object.__setattr__(self, key, value)

    def __getattr__(self, key):                       # only reached when the attribute does not exist
        if key in ("x", "edge_index", "edge_attr", "y", "pos"):
            return None                               # PyG returns None for the unset standard attributes
        raise AttributeError(f"'{type(self).__name__}' object has no attribute '{key}'")

    @property
    def keys(self) -> List[str]:
        return list(self._keys)

    def __getitem__(self, key):
        return getattr(self, key)

    def __contains__(self, key):
        return key in self._keys

    def items(self):
        return [(k, getattr(self, k)) for k in self._keys]

    @property
    def num_nodes(self) -> int:
        if self.x is not None:
            return self.x.shape[0]
        if self.pos is not None:
            return self.pos.shape[0]
        ei = self.edge_index
        return int(ei.max()) + 1 if ei is not None and ei.numel() else 0

    @property
    def num_edges(self) -> int:
        return 0 if self.edge_index is None else self.edge_index.shape[1]

    @property
    def num_node_features(self) -> int:
        return 0 if self.x is None else (1 if self.x.dim() == 1 else self.x.shape[1])

    def to(self, device, non_blocking: bool = False):
        for k in self._keys:
            v = getattr(self, k)
            if torch.is_tensor(v):
                object.__setattr__(self, k, v.to(device, non_blocking=non_blocking))
        return self

    def cpu(self):
        return self.to("cpu")

    def cuda(self, device="cuda"):
        return self.to(device)

    def __repr__(self):
        parts = [f"{k}={list(v.shape)}" if torch.is_tensor(v) else f"{k}={v!r}" for k, v in self.items()]
        return f"{type(self).__name__}({', '.join(parts)})"


class Batch(Data):
    """Several graphs as one disconnected graph (PyG numbering) + ``batch`` [N] int64, ``ptr`` [B+1] int64."""

    @property
    def num_graphs(self) -> int:
        return int(self.ptr.numel()) - 1

    def to_data_list(self) -> List[Data]:
        """The graphs of the batch again (graph-local edge numbering); synchronises (offsets are read on the host)."""
        ptr = self.ptr.cpu().numpy()
        eptr = self._edge_ptr.cpu().numpy() if getattr(self, "_edge_ptr", None) is not None else None
        out = []
        for g in range(len(ptr) - 1):
            d = Data()
            for k in self._keys:
                if k in ("batch", "ptr"):
                    continue
                v = getattr(self, k)
                if k == "edge_index":
                    d.edge_index = v[:, eptr[g]:eptr[g + 1]] - int(ptr[g])
                elif torch.is_tensor(v) and eptr is not None and "edge" in k:
                    setattr(d, k, v[eptr[g]:eptr[g + 1]])
                elif torch.is_tensor(v):
                    setattr(d, k, v[ptr[g]:ptr[g + 1]])
            out.append(d)
        return out


def create_graph_data(graph, target: np.ndarray, bounding_box: np.ndarray, point_cloud) -> Data:
    """One graph sample as the reference stores it (dataset_creation.py:786-814): x f32 [N, Dn], edge_index int64 [2, E]
    (= ``graph.E.T``), edge_attr f32 [E, De], y f32 [N, 1 + box] (class label | box), pos / vel f32 [N, 2]."""
    merged = np.concatenate((target, bounding_box), axis=1)
    f32 = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32)
    return Data(x=f32(graph.X_feat), edge_index=torch.tensor(np.asarray(graph.E).T, dtype=torch.long),
                edge_attr=f32(graph.E_feat), y=f32(merged), pos=f32(point_cloud.X_cc),
                vel=f32(point_cloud.V_cc_compensated))


# ---- the resident dataset ------------------------------------------------------------------------------------------
class GraphStore:
    """All graphs of a split concatenated in HBM; ``collate(ids)`` builds a ``Batch`` on the device."""

    def __init__(self, graphs: Sequence[Data], device="cuda"):
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("GraphStore keeps the dataset in HBM and collates with HIP kernels: a GPU device is "
                               "required (there is no CPU collation path)")
        self.device = device
        self.num_graphs = len(graphs)
        keys = graphs[0].keys if graphs else []
        for g in graphs:
            if g.keys != keys:
                raise ValueError(f"all graphs must carry the same attributes ({keys} vs {g.keys})")
        self.keys = [k for k in keys if torch.is_tensor(graphs[0][k])]
        self.node_sizes = np.array([g.num_nodes for g in graphs], dtype=np.int64)
        self.edge_sizes = np.array([g.num_edges for g in graphs], dtype=np.int64)
        self.node_ptr = np.concatenate(([0], np.cumsum(self.node_sizes)))
        self.edge_ptr = np.concatenate(([0], np.cumsum(self.edge_sizes)))
        self.resident: Dict[str, torch.Tensor] = {}
        self.kind: Dict[str, str] = {}                 # "node" | "edge" | "own" (attribute with its own row counts)
        self.own_ptr: Dict[str, np.ndarray] = {}
        for k in self.keys:
            parts = [g[k] for g in graphs]
            if k == "edge_index":
                for p in parts:
                    if p.dtype != torch.int64 or p.dim() != 2 or p.shape[0] != 2:
                        raise ValueError("edge_index must be int64 [2, E]")
                self.resident[k] = torch.cat(parts, dim=1).contiguous().to(device)
                self.kind[k] = "edge_index"
                continue
            rows = np.array([p.shape[0] for p in parts], dtype=np.int64)
            cat = torch.cat(parts, dim=0).contiguous()
            ops._words_per_row(cat)                    # raises for dtypes the copy kernel cannot move
            self.resident[k] = cat.to(device)
            if "edge" in k and np.array_equal(rows, self.edge_sizes):
                self.kind[k] = "edge"
            elif np.array_equal(rows, self.node_sizes):
                self.kind[k] = "node"
            else:
                self.kind[k] = "own"
                self.own_ptr[k] = np.concatenate(([0], np.cumsum(rows)))

    def __len__(self) -> int:
        return self.num_graphs

    def nbytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self.resident.values())

    def collate(self, ids: Sequence[int]) -> Batch:
        ids = np.asarray(ids, dtype=np.int64).reshape(-1)
        if ids.size and (ids.min() < 0 or ids.max() >= self.num_graphs):
            raise IndexError("graph id out of range")
        b = ids.size
        # offset tables for the node-like / edge-like / own-count attributes, uploaded as ONE int64 array
        tabs, where = [], {}

        def table(name, src_ptr):
            sizes = src_ptr[ids + 1] - src_ptr[ids]
            dst = np.concatenate(([0], np.cumsum(sizes)))
            where[name] = (len(tabs), int(dst[-1]))
            tabs.append(np.concatenate((src_ptr[ids], dst)))           # [b] source starts | [b + 1] batch offsets
            return dst

        node_dst = table("node", self.node_ptr)
        table("edge", self.edge_ptr)
        for k, p in self.own_ptr.items():
            table(k, p)
        host = np.stack(tabs) if b or tabs else np.zeros((0, 1), dtype=np.int64)
        dev = torch.from_numpy(np.ascontiguousarray(host)).to(self.device)

        def tab(name):
            i, total = where[name]
            return dev[i, :b], dev[i, b:], total

        out = Batch()
        n_src, n_dst, n_total = tab("node")
        e_src, e_dst, e_total = tab("edge")
        batch_vec = None
        for k in self.keys:
            kind, res = self.kind[k], self.resident[k]
            if kind == "edge_index":
                val = ops.collate_edges(res, e_src, e_dst, n_dst[:b].contiguous(), e_total) if b else res[:, :0].clone()
            elif kind == "node":
                if batch_vec is None and b and ops._words_per_row(res) > 0:
                    val, batch_vec = ops.collate_rows(res, n_src, n_dst, n_total, want_batch=True)
                else:
                    val = ops.collate_rows(res, n_src, n_dst, n_total) if b else res[:0].clone()
            elif kind == "edge":
                val = ops.collate_rows(res, e_src, e_dst, e_total) if b else res[:0].clone()
            else:
                s, d, total = tab(k)
                val = ops.collate_rows(res, s, d, total) if b else res[:0].clone()
            setattr(out, k, val)
        if batch_vec is None:                          # no node attribute present: derive it from the offsets
            batch_vec = torch.repeat_interleave(torch.arange(b, device=self.device),
                                                torch.from_numpy(np.diff(node_dst)).to(self.device))
        out.batch = batch_vec
        out.ptr = n_dst.clone() if b else torch.zeros(1, dtype=torch.int64, device=self.device)
        object.__setattr__(out, "_edge_ptr", e_dst if b else torch.zeros(1, dtype=torch.int64, device=self.device))
        return out


class DataLoader:
    """``torch_geometric.loader.DataLoader(graph_list, batch_size, shuffle)`` for graphs resident on the device: iterating
    yields ``Batch`` objects already in HBM.  ``len()`` = number of batches; the last one may be smaller (drop_last=False);
    ``shuffle`` draws a fresh ``torch.randperm`` per epoch (torch ``RandomSampler``)."""

    def __init__(self, dataset, batch_size: int = 1, shuffle: bool = False, device="cuda",
                 generator: Optional[torch.Generator] = None, drop_last: bool = False):
        self.dataset = dataset if isinstance(dataset, GraphStore) else list(dataset)
        self.batch_size = int(batch_size)
        if self.batch_size < 1:
            raise ValueError("batch_size must be a positive integer")
        self.shuffle = shuffle
        self.device = device
        self.generator = generator
        self.drop_last = drop_last
        self._store: Optional[GraphStore] = dataset if isinstance(dataset, GraphStore) else None

    @property
    def store(self) -> GraphStore:
        if self._store is None:
            self._store = GraphStore(self.dataset, self.device)
        return self._store

    def _num_graphs(self) -> int:
        return len(self.dataset)

    def __len__(self) -> int:
        n = self._num_graphs()
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def batch_ids(self) -> List[np.ndarray]:
        n = self._num_graphs()
        order = torch.randperm(n, generator=self.generator).numpy() if self.shuffle else np.arange(n)
        return [order[i * self.batch_size:(i + 1) * self.batch_size] for i in range(len(self))]

    def __iter__(self) -> Iterator[Batch]:
        store = self.store
        for ids in self.batch_ids():
            yield store.collate(ids)


# ---- files -----------------------------------------------------------------------------------------------------------
def save_graph(data: Data, path: str, pyg_compatible: bool = False) -> None:
    """One graph on disk.  Default: a plain dict of CPU tensors (``torch.save``), readable without this package.

    ``pyg_compatible=True`` writes what the reference's dataset creation writes (``torch.save(Data(...), path)``,
    preprocessor/radarscenes/dataset_creation.py:121-123,786-814) so that the reference's ``torch.load`` in
    utils/data_handling.py:27 gets a ``torch_geometric.data.Data``: with torch_geometric importable the real class is used;
    without it the pickle is written against stand-in classes registered under the real class paths, with the state layout of
    torch_geometric 2.x (``Data.__dict__ = {'_store': GlobalStorage}``, ``GlobalStorage.__dict__ = {'_mapping': {...},
    '_parent': <the Data>}`` -- BaseStorage.__getstate__ dereferences its weak parent).  UNPINNED in this image: there is no
    torch_geometric to load such a file with; ``load_graph`` reads it back, and tests/test_data_host.py carries a self-skipping
    test that round-trips through the real package when it exists."""
    tensors = {k: (v.detach().cpu() if torch.is_tensor(v) else v) for k, v in data.items()}
    if not pyg_compatible:
        payload = {"format": FORMAT_TAG}
        payload.update(tensors)
        torch.save(payload, path)
        return
    try:
        from torch_geometric.data import Data as PygData          # the real thing, when it exists
        torch.save(PygData(**tensors), path)
        return
    except ImportError:
        pass
    import sys
    import warnings
    warnings.warn("save_graph(pyg_compatible=True) without torch_geometric installed: the file is written against stand-in classes "
                  "with the torch_geometric 2.0 - 2.3 state layout (Data._store / GlobalStorage._mapping, _parent); it has never been "
                  "loaded by a real torch_geometric in this environment -- verify it where the package exists", stacklevel=2)
    names = ("torch_geometric", "torch_geometric.data", "torch_geometric.data.data", "torch_geometric.data.storage")
    mods = {n: types.ModuleType(n) for n in names}
    store_cls = type("GlobalStorage", (), {"__module__": "torch_geometric.data.storage"})
    data_cls = type("Data", (), {"__module__": "torch_geometric.data.data"})
    mods["torch_geometric.data.storage"].GlobalStorage = store_cls
    mods["torch_geometric.data.data"].Data = data_cls
    obj, store = data_cls(), store_cls()
    store.__dict__["_mapping"] = tensors
    store.__dict__["_parent"] = obj
    obj.__dict__["_store"] = store
    saved = {n: sys.modules.get(n) for n in names}
    sys.modules.update(mods)                                       # pickle resolves classes by module path while writing
    try:
        torch.save(obj, path)
    finally:
        for n, m in saved.items():
            if m is None:
                sys.modules.pop(n, None)
            else:
                sys.modules[n] = m


class _Opaque:
    """Stand-in for any torch_geometric class met while unpickling a reference ``graph_*.pt`` (torch_geometric itself is
    not needed): keeps whatever state the pickle carries."""

    def __init__(self, *args, **kwargs):
        pass

    def __setstate__(self, state):
        if isinstance(state, dict):
            self.__dict__.update(state)
        else:
            self.__dict__["_state"] = state


class _GraphUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module.split(".")[0] == "torch_geometric":
            return type(name, (_Opaque,), {})
        return super().find_class(module, name)


_PICKLE = types.SimpleNamespace(__name__="pickle", Unpickler=_GraphUnpickler, load=lambda f, **kw: _GraphUnpickler(f, **kw).load(),
                                loads=pickle.loads, dump=pickle.dump, dumps=pickle.dumps, Pickler=pickle.Pickler)


def _tensors_of(obj) -> Dict[str, object]:
    if isinstance(obj, dict):
        return {k: v for k, v in obj.items() if k != "format"}
    d = dict(getattr(obj, "__dict__", {}))
    store = d.get("_store")
    if store is not None:                              # torch_geometric >= 2.0: Data -> GlobalStorage -> _mapping
        return dict(getattr(store, "__dict__", {}).get("_mapping", {}))
    return {k: v for k, v in d.items() if not k.startswith("_")}       # torch_geometric 1.x: attributes in __dict__


def load_graph(path: str) -> Data:
    """Reads a graph written by ``save_graph`` or a pickled ``torch_geometric.data.Data`` as the reference's dataset
    creation writes them (``torch.save(data, f"graph_{i}.pt")``).  The latter is decoded structurally (2.x:
    ``_store._mapping``; 1.x: ``__dict__``) and is UNPINNED here: no torch_geometric in this image to write a real file."""
    obj = torch.load(path, map_location="cpu", pickle_module=_PICKLE, weights_only=False)
    fields = {k: v for k, v in _tensors_of(obj).items() if v is not None}
    if "edge_index" not in fields and "x" not in fields:
        raise ValueError(f"{path}: no graph attributes found")
    return Data(**fields)


def get_data_loaders(splits: Iterable[str], root: str, batch_size: int, shuffle: bool, device="cuda"):
    """``utils/data_handling.py:7-36``: every ``{root}/{split}/*.pt`` (sorted by name) -> one DataLoader per split, plus
    the dataset description ``{root}/config.json``."""
    loaders = {}
    for split in splits:
        names = sorted(glob.glob(f"{root}/{split}/*.pt"))
        loaders[split] = DataLoader([load_graph(n) for n in names], batch_size=batch_size, shuffle=shuffle, device=device)
    with open(f"{root}/config.json", "r") as f:
        config = json.load(f)
    return loaders, config
