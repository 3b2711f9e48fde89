"""Torch-tensor front end of librgnn.so.  PyTorch is used here for device memory, streams and nothing else:
every function validates its tensors, allocates outputs and forwards raw device pointers plus the current
HIP stream to the C ABI (include/rgnn.h).  CPU tensors are rejected -- there is no fallback path."""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import RgnnError, RgnnGrid, RgnnLinearArgs, check, lib

EDGE_FEATURE_CODES = {
    "point_pair_features": 0, "spatial_euclidean_distance": 1, "velocity_euclidean_distance": 2,
    "relative_position": 3, "relative_velocity": 4,
}
EDGE_FEATURE_WIDTH = {"point_pair_features": 4, "spatial_euclidean_distance": 1, "velocity_euclidean_distance": 1,
                      "relative_position": 2, "relative_velocity": 2}
NODE_FEATURE_CODES = {"rcs": 0, "time_index": 1, "degree": 2, "velocity_vector_length": 3, "velocity_vector": 4,
                      "spatial_coordinates": 5}
NODE_FEATURE_WIDTH = {"rcs": 1, "time_index": 1, "degree": 1, "velocity_vector_length": 1, "velocity_vector": 2,
                      "spatial_coordinates": 2}
AGGR_CODES = {"max": 0, "mean": 1, "add": 2, "sum": 2}
MAX_FUSED_EDGE_WIDTH = 32        # edge attributes per edge the fused message kernels take (rgnn_mpnn_aggregate / _edge_hidden)
MAX_FUSED_EDGE_WIDTH_BWD = 16    # ... and their backward (rgnn_mpnn_aggregate_bwd)

class ForwardContext:
    """What one forward pass carries besides its tensors, per calling THREAD (``ctx()``): two models driven from two Python
    threads, each on its own stream, share none of it (tests/test_gpu_threads.py).
      bounds       the pool the launches of this pass take their bound words from (``bound_tracking``; None: no f16x2 form)
      frame_scope  gnn.linear.frame_scope in force (per-frame BatchNorm statistics), or None
      side()       a side stream per device for graph-only work that overlaps the main stream (forked and joined by events)
      profiler     optional launch profiler (bench.py): an object with ``.begin(kind)`` -> token and ``.end(token, **work)``;
                   ``begin`` arms a pair of HIP events through rgnn_profile_next_launch(), which the library records immediately
                   around the kernel launch (no Python between the event and the launch).  None = no overhead.
    Process-wide state that remains: the weight-plane caches (keyed on the weights; entries carry the event of the stream that
    filled them), the split-K scratch (one per device and stream), COUNTERS (diagnostics for the tests)."""
    __slots__ = ("bounds", "frame_scope", "profiler", "_side", "_finalizer", "__weakref__")

    def __init__(self):
        self.bounds = self.frame_scope = self.profiler = None
        self._side = {}
        self._finalizer = None

    def side(self, device, role: str = "plan") -> "torch.cuda.Stream":
        """This thread's side stream on ``device`` for ``role``: "plan" (TargetCSR.start_win_plan: graph-only work beside the main
        stream) and "search" (HotPath.begin: the search half of the next batch) are ONE stream -- in it the search of batch j + 1
        precedes the plan of batch j, which waits for the main stream to reach batch j's model stage: nothing is held up -- and
        "upload" / "download" (FrameStreamer) one each: with the main stream four, as many as there are hardware queues, each on
        its own (``independent_stream``)."""
        key = (torch.device(device).index, "side" if role in ("plan", "search") else role)
        if key not in self._side:
            self._side[key] = independent_stream(device, owner=self)
            if self._finalizer is None:                       # the streams die with the context (a thread that ends: its thread-local)
                import weakref
                self._finalizer = weakref.finalize(self, _release_streams, self._side)
        return self._side[key]


# ---- streams on hardware queues of their own ----------------------------------------------------------------------------------
# ROCm maps the HIP streams of a process onto 4 hardware queues per priority level (GPU_MAX_HW_QUEUES), assigned at a stream's first
# use; torch hands out streams of a pool of 32 per level.  Two streams that land on ONE queue run strictly one after the other,
# whatever their events say: a download parked behind the event of batch j on the queue that also carries the main stream holds up
# batch j + 1's kernels, an upload does the same to the search stage (tools/hw_queue_probe.py shows the map; the streamed C2 step
# measured 2.16 or 2.35 ms per batch depending on which streams a run happened to get, tools/stream_probe.py).  So a side stream is
# taken only after it has been SEEN to run beside every stream already in use: a spinning kernel occupies the other stream, a
# one-word fill on the candidate must complete meanwhile.  (Streams of different priority never share a queue, but a high-priority
# stream is no way out: with the plan and the search on one the streamed step took 3.1 ms -- while a high-priority queue holds a
# packet, even one that only waits for an event, the normal queues are not served.)
_INDEPENDENT = {}          # device index -> streams handed out (the default stream first); never destroyed: process lifetime
_INDEPENDENT_LOCK = __import__("threading").Lock()


def _new_stream(dev) -> "torch.cuda.Stream":
    """A non-blocking HIP stream of our own (rgnn_stream_create) seen by torch as an ExternalStream: torch's pool of 32 streams per
    device hands the same streams out again after 32 requests, and a candidate taken from it could not be given back."""
    h = C.c_void_p()
    with torch.cuda.device(dev):
        check(lib.rgnn_stream_create(C.byref(h)))
    return torch.cuda.ExternalStream(h.value, device=dev)


def _runs_beside(busy: "torch.cuda.Stream", cand: "torch.cuda.Stream", word: torch.Tensor) -> bool:
    import time
    with torch.cuda.stream(busy):
        torch.cuda._sleep(6_000_000)                          # ~3 ms
    ev = torch.cuda.Event()
    with torch.cuda.stream(cand):
        word.fill_(1.0)
        ev.record(cand)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.0015 and not ev.query():
        pass
    ok = ev.query()
    torch.cuda.synchronize(busy.device)
    return ok


_PROBE_OWNER = {}          # device index -> id of the ForwardContext whose streams are verified by running (the first one to ask)


def _release_streams(side: dict) -> None:
    """Finalizer of a ForwardContext: its streams are destroyed (HIP releases a stream's resources once its queued work is through)
    and leave the list of streams in use -- thread churn no longer leaks a stream per thread and role."""
    with _INDEPENDENT_LOCK:
        for (dev_index, _role), st in list(side.items()):
            taken = _INDEPENDENT.get(dev_index, [])
            _INDEPENDENT[dev_index] = [t for t in taken if t.cuda_stream != st.cuda_stream]
            try:
                lib.rgnn_stream_destroy(C.c_void_p(st.cuda_stream))
            except Exception:
                pass
        side.clear()


def independent_stream(device, tries: int = 10, owner=None) -> "torch.cuda.Stream":
    """A stream on ``device`` that shares its hardware queue with no stream handed out here before, nor with the default or the
    current stream.  Verified by running (a few ms per stream in use, once; skipped while a capture is under way, when torch has no
    spin kernel and under RGNN_NO_QUEUE_CHECK -- the stream is then merely new).  Candidates that fail stay alive until one passes
    (a destroyed stream's queue would be the next one's) and are destroyed then; after ``tries`` the last one is taken as it is."""
    dev = torch.device(device)
    if torch.cuda.is_current_stream_capturing() or not hasattr(torch.cuda, "_sleep") or os.environ.get("RGNN_NO_QUEUE_CHECK"):
        return _new_stream(dev)
    # The probe (spin kernels on the streams in use, a device-wide synchronize per candidate) runs for ONE context per process and
    # device -- the first to ask, normally the main thread before anything is captured.  Contexts of further threads get plain new
    # streams: their probes would inject ~3 ms kernels into streams other threads are using and synchronise the device under
    # them, which can also invalidate a capture running in another thread.
    with _INDEPENDENT_LOCK:
        first = _PROBE_OWNER.setdefault(dev.index, id(owner) if owner is not None else 0)
    if owner is not None and first != id(owner):
        st = _new_stream(dev)
        with _INDEPENDENT_LOCK:
            _INDEPENDENT.setdefault(dev.index, [torch.cuda.default_stream(dev)]).append(st)
        return st
    with _INDEPENDENT_LOCK:
        taken = _INDEPENDENT.setdefault(dev.index, [torch.cuda.default_stream(dev)])
        others = list(taken)
        cur = torch.cuda.current_stream(dev)
        if all(cur.cuda_stream != st.cuda_stream for st in others):
            others.append(cur)
        if len(others) >= int(os.environ.get("GPU_MAX_HW_QUEUES", "4") or 4):
            cand = _new_stream(dev)                           # (every hardware queue already carries one of `others`: nothing to find)
            taken.append(cand)
            return cand
        word = torch.zeros(1, device=dev)
        rejected = []
        cand = _new_stream(dev)
        for _ in range(tries - 1):
            if all(_runs_beside(st, cand, word) for st in others):
                break
            rejected.append(cand)
            cand = _new_stream(dev)
        for st in rejected:
            lib.rgnn_stream_destroy(C.c_void_p(st.cuda_stream))
        taken.append(cand)
        return cand


_TLS = __import__("threading").local()


def reload_env() -> None:
    """After changing an RGNN_* switch that the LIBRARY reads (os.environ inside a running process): the library caches its
    environment reads per call site (rgnn_env_reload, rgnn.h)."""
    lib.rgnn_env_reload()


def ctx() -> ForwardContext:
    c = getattr(_TLS, "c", None)
    if c is None:
        c = _TLS.c = ForwardContext()
    return c


def arm_profile_events(start_event: "torch.cuda.Event", stop_event: "torch.cuda.Event") -> None:
    lib.rgnn_profile_next_launch(C.c_void_p(start_event.cuda_event), C.c_void_p(stop_event.cuda_event))

STATUS_KNN_TOO_FEW_POINTS = 1
STATUS_DOT_PRODUCT = 2
STATUS_TIME_INDEX_OVERFLOW = 4
STATUS_EDGE_COUNT_CHANGED = 8
STATUS_NOT_SYMMETRIC = 16
STATUS_SPLITK_TIMEOUT = 32
SPLITK_TIMEOUT_WORD = 1000          # rgnn.h RGNN_SPLITK_TIMEOUT_WORD


def _dev(t: torch.Tensor, name: str, dtype=None) -> torch.Tensor:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"radargnn_amd: `{name}` must be a tensor on the MI355X (got "
                           f"{'a CPU tensor' if isinstance(t, torch.Tensor) else type(t).__name__}); "
                           "the HIP path has no CPU fallback")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"radargnn_amd: `{name}` must be {dtype}, got {t.dtype}")
    return t


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream_id(device=None) -> int:
    """The current stream's handle as an int (what ``torch.cuda.current_stream(device).cuda_stream`` returns, at 0.3 us instead of 8)."""
    if _raw_stream is not None and _raw_device is not None:
        if device is None:
            return _raw_stream(_raw_device())
        idx = torch.device(device).index
        return _raw_stream(_raw_device() if idx is None else idx)
    return torch.cuda.current_stream(device).cuda_stream


def _stream():
    # ~130 launches per step: the raw accessors cost 0.3 us instead of the 8 us of torch.cuda.current_stream()
    if _raw_stream is not None and _raw_device is not None:
        return C.c_void_p(_raw_stream(_raw_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _rowmajor(t: torch.Tensor, name: str) -> torch.Tensor:
    """2-D tensor whose rows are contiguous (stride(1) == 1); the row stride may exceed the width (views)."""
    if t.dim() != 2:
        raise ValueError(f"`{name}` must be 2-D")
    if t.shape[1] > 1 and t.stride(1) != 1 or (t.shape[0] > 1 and t.stride(0) < t.shape[1]):
        t = t.contiguous()
    return t


def _ld(t: torch.Tensor) -> int:
    return t.stride(0) if t.shape[0] > 1 else max(t.shape[1], 1)


# ------------------------------------------------------------------------------------------------ primitives
def exclusive_scan_i32(x: torch.Tensor, out: Optional[torch.Tensor] = None, tmp: Optional[torch.Tensor] = None) -> torch.Tensor:
    _dev(x, "x", torch.int32)
    n = x.numel()
    if out is None:
        out = torch.empty(n + 1, dtype=torch.int32, device=x.device)
    if tmp is None:
        tmp = torch.empty(max(lib.rgnn_scan_tmp_bytes(n), 256), dtype=torch.uint8, device=x.device)
    check(lib.rgnn_exclusive_scan_i32(_ptr(x.contiguous()), _ptr(out), n, _ptr(tmp), _stream()))
    return out


# ------------------------------------------------------------------------------------------------ graph
class GridHash:
    """Uniform-grid binning of a batch of frames (rgnn_grid_build); keeps the workspace alive."""

    def __init__(self, X: torch.Tensor, frame_ptr: torch.Tensor):
        _dev(X, "X", torch.float64)
        _dev(frame_ptr, "frame_ptr", torch.int64)
        if X.dim() != 2 or X.shape[1] not in (2, 4, 8):
            raise ValueError("X must be [N,2], [N,4] or [N,8]")
        self.X = X.contiguous()
        self.frame_ptr = frame_ptr.contiguous()
        self.n = self.X.shape[0]
        self.n_frames = self.frame_ptr.numel() - 1
        nbytes = lib.rgnn_grid_workspace_bytes(self.n, self.n_frames, self.X.shape[1])
        self.ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=X.device)
        self.desc = RgnnGrid(_ptr(self.X), self.X.shape[1], self.n, _ptr(self.frame_ptr), self.n_frames,
                             _ptr(self.ws), self.ws.numel())

    def build(self, cell_size: float = 0.0, pts_per_cell: float = 2.0, max_frame_points: int = 0) -> "GridHash":
        """``max_frame_points`` > 0: the caller knows (host copy of the frame sizes) that no frame is larger -- moderately sized
        frames are then binned by one launch instead of five (rgnn_grid_build_frames)."""
        check(lib.rgnn_grid_build_frames(C.byref(self.desc), float(cell_size), float(pts_per_cell), int(max_frame_points),
                                         _stream()))
        return self

    def _order_views(self):
        if getattr(self, "_views", None) is None:
            o, r = C.c_int64(0), C.c_int64(0)
            check(lib.rgnn_grid_order_offsets(self.n, self.n_frames, self.X.shape[1], C.byref(o), C.byref(r)))
            self._views = (self.ws[o.value:o.value + 4 * self.n].view(torch.int32), self.ws[r.value:r.value + 4 * self.n].view(torch.int32))
        return self._views

    def cell_order(self) -> torch.Tensor:
        """int32 [n]: point rows in grid-cell order (a spatially coherent visiting order) -- a VIEW of the workspace (valid
        until the next ``build`` on this object; the tensor keeps the workspace alive)."""
        return self._order_views()[0]

    def cell_rank(self) -> torch.Tensor:
        """int32 [n]: position of point i in the cell order (the inverse permutation), a view like ``cell_order``."""
        return self._order_views()[1]


def radius_graph_count(X: torch.Tensor, frame_ptr: torch.Tensor, r: float, static: Optional[dict] = None,
                       max_frame_points: int = 0):
    """Pass 1 of the radius graph (no host read): -> GridHash, rowptr int32 [N+1] (rowptr[-1] = E on the device).
    ``static``: a dict that keeps the grid workspace and the output buffers alive across calls (same addresses every
    time, as a captured HIP graph of the later stages needs)."""
    if static is not None and "grid" in static:
        g, deg, rowptr, tmp = static["grid"], static["deg"], static["rowptr"], static["tmp"]
    else:
        g = GridHash(X, frame_ptr)
        deg = torch.empty(g.n, dtype=torch.int32, device=X.device)
        rowptr = torch.empty(g.n + 1, dtype=torch.int32, device=X.device)
        tmp = torch.empty(max(lib.rgnn_scan_tmp_bytes(g.n), 256), dtype=torch.uint8, device=X.device)
        if static is not None:
            static.update(grid=g, deg=deg, rowptr=rowptr, tmp=tmp)
    g.build(cell_size=float(r) if r > 0 else 1e-300, max_frame_points=max_frame_points)
    check(lib.rgnn_radius_graph_count(C.byref(g.desc), float(r), _ptr(deg), _stream()))
    return g, exclusive_scan_i32(deg, out=rowptr, tmp=tmp)


def radius_grid(X: torch.Tensor, frame_ptr: torch.Tensor, r: float, static: dict, max_frame_points: int = 0) -> "GridHash":
    """The grid build alone, on the static workspace an earlier ``radius_graph_count(..., static=...)`` left in ``static`` (replayed
    steps: the rows come from ``radius_graph_rows_direct``)."""
    g = static["grid"]
    g.build(cell_size=float(r) if r > 0 else 1e-300, max_frame_points=max_frame_points)
    return g


def radius_graph_rows_direct(g: "GridHash", rowptr_committed: torch.Tensor, r: float, n_edges: int, status: torch.Tensor,
                             want_edge_index: bool = True, relative_position: Optional[str] = None):
    """Search + fill in one launch for a replayed step (rgnn_radius_graph_rows_direct): rows of the committed lengths are written at
    the committed places, any other row keeps its contents and sets STATUS_EDGE_COUNT_CHANGED.  -> (col, edge_index[, rel])."""
    _dev(rowptr_committed, "rowptr_committed", torch.int32)
    col = torch.empty(n_edges, dtype=torch.int32, device=rowptr_committed.device)
    ei = torch.empty((2, n_edges), dtype=torch.int64, device=rowptr_committed.device) if want_edge_index else None
    rel = torch.empty((n_edges, 2), dtype=torch.float32, device=rowptr_committed.device) if relative_position else None
    tmp = torch.empty(max(n_edges, 1), dtype=torch.int32, device=rowptr_committed.device)
    check(lib.rgnn_radius_graph_rows_direct(C.byref(g.desc), float(r), _ptr(rowptr_committed), _ptr(col), _ptr(ei), n_edges, _ptr(tmp),
                                            _ptr(status), _ptr(rel), 1 if relative_position == "undirected" else 0, _stream()))
    return (col, ei, rel) if relative_position else (col, ei)


FUSED_RADIUS_ROWS = __import__("os").environ.get("RGNN_RADIUS_SPLIT_FILL") is None


def radius_graph_fill(g: "GridHash", rowptr: torch.Tensor, r: float, n_edges: int, want_edge_index: bool = True,
                      guard_status: Optional[torch.Tensor] = None, relative_position: Optional[str] = None):
    """Pass 2: -> col int32 [E] (ascending per row), edge_index int64 [2,E].
    ``guard_status``: n_edges was not just read back from rowptr (captured step): the kernels verify rowptr[n] == n_edges on
    the device, write nothing otherwise and set STATUS_EDGE_COUNT_CHANGED in it (rgnn_radius_graph_fill_checked).
    ``relative_position`` ("directed" | "undirected"): also return the relative_position edge attributes float32 [E, 2] as a
    third value -- written by the same launch (rgnn_radius_graph_rows)."""
    col = torch.empty(n_edges, dtype=torch.int32, device=rowptr.device)
    ei = torch.empty((2, n_edges), dtype=torch.int64, device=rowptr.device) if want_edge_index else None
    if FUSED_RADIUS_ROWS:
        rel = torch.empty((n_edges, 2), dtype=torch.float32, device=rowptr.device) if relative_position else None
        tmp = torch.empty(max(n_edges, 1), dtype=torch.int32, device=rowptr.device)
        check(lib.rgnn_radius_graph_rows(C.byref(g.desc), float(r), _ptr(rowptr), _ptr(col), _ptr(ei), n_edges, _ptr(tmp),
                                         _ptr(guard_status), _ptr(rel), 1 if relative_position == "undirected" else 0, _stream()))
        return (col, ei, rel) if relative_position else (col, ei)
    tmp = torch.empty(max(2 * n_edges, 1), dtype=torch.int32, device=rowptr.device)
    if guard_status is not None:
        check(lib.rgnn_radius_graph_fill_checked(C.byref(g.desc), float(r), _ptr(rowptr), _ptr(col), _ptr(ei), n_edges,
                                                 _ptr(tmp), _ptr(guard_status), _stream()))
    else:
        check(lib.rgnn_radius_graph_fill(C.byref(g.desc), float(r), _ptr(rowptr), _ptr(col), _ptr(ei), n_edges, _ptr(tmp),
                                         _stream()))
    if relative_position:
        rel, _ = edge_features(g.X, g.X, ei, ["relative_position"], relative_position, dtype=torch.float32)   # (V is not read)
        return col, ei, rel
    return col, ei


def radius_graph(X: torch.Tensor, frame_ptr: torch.Tensor, r: float, want_edge_index: bool = True,
                 grid_out: Optional[list] = None):
    """-> rowptr int32 [N+1], col int32 [E] (ascending per row), edge_index int64 [2,E] (row 0 = query i,
    row 1 = neighbour j).  One device->host read of E (count -> scan -> fill protocol).  ``grid_out``: a list the
    GridHash is appended to (for ``cell_order``)."""
    g = GridHash(X, frame_ptr).build(cell_size=float(r) if r > 0 else 1e-300)
    if grid_out is not None:
        grid_out.append(g)
    n = g.n
    deg = torch.empty(n, dtype=torch.int32, device=X.device)
    check(lib.rgnn_radius_graph_count(C.byref(g.desc), float(r), _ptr(deg), _stream()))
    rowptr = exclusive_scan_i32(deg)
    n_edges = int(rowptr[-1].item()) if n > 0 else 0
    col, ei = radius_graph_fill(g, rowptr, r, n_edges, want_edge_index)
    return rowptr, col, ei


def knn_graph(X: torch.Tensor, frame_ptr: torch.Tensor, k: int, status: Optional[torch.Tensor] = None,
              want_edge_index: bool = True, pts_per_cell: Optional[float] = None, grid_out: Optional[list] = None,
              static: Optional[dict] = None, max_frame_points: int = 0, relative_position: Optional[str] = None,
              degree_init: bool = False):
    """-> nbr int32 [N,k] (distance asc, index asc), edge_index int64 [2, N*k], status int32 [1].
    ``static``: a dict that keeps the grid workspace and the output buffers alive across calls (same addresses every
    time, as a captured HIP graph of the later stages needs).
    ``pts_per_cell``: target occupancy of a grid cell (a scheduling choice: the rows do not depend on it); default 2, 3 from
    k = 16 (profiles/r04_knn_cell_probe.txt: 512 x 300 points, k = 20: 299 -> 275 us; 64 x 3000: level; k = 40: 591 -> 552 us)."""
    if pts_per_cell is None:
        pts_per_cell = 3.0 if k >= 16 else 2.0
    if static is not None and "grid" in static:
        g, nbr, ei = static["grid"], static["nbr"], static["ei"]
        g.build(cell_size=0.0, pts_per_cell=pts_per_cell, max_frame_points=max_frame_points)
    else:
        g = GridHash(X, frame_ptr).build(cell_size=0.0, pts_per_cell=pts_per_cell, max_frame_points=max_frame_points)
        nbr = torch.empty((g.n, k), dtype=torch.int32, device=X.device)
        ei = torch.empty((2, g.n * k), dtype=torch.int64, device=X.device) if want_edge_index else None
        if static is not None:
            static.update(grid=g, nbr=nbr, ei=ei)
    if grid_out is not None:
        grid_out.append(g)
    n = g.n
    if status is None:
        status = torch.zeros(1, dtype=torch.int32, device=X.device)
    if (relative_position or degree_init) and k <= 64:
        # the write-out of the search also emits the relative_position attributes ("directed" | "undirected") and presets the
        # undirected-degree array with the out-degrees (rgnn_knn_graph_attrs): returned as 4th / 5th value
        rel = torch.empty((g.n * k, 2), dtype=torch.float32, device=X.device) if relative_position else None
        deg = torch.empty(g.n, dtype=torch.int32, device=X.device) if degree_init else None
        # (with the largest frame known: small frames are searched by brute force per frame, rgnn_knn_graph_frames)
        check(lib.rgnn_knn_graph_frames(C.byref(g.desc), int(k), int(max_frame_points), _ptr(nbr), _ptr(ei), _ptr(status), _ptr(rel),
                                        1 if relative_position == "undirected" else 0, _ptr(deg), _stream()))
        return nbr, ei, status, rel, deg
    check(lib.rgnn_knn_graph_frames(C.byref(g.desc), int(k), int(max_frame_points), _ptr(nbr), _ptr(ei), _ptr(status), None, 0, None,
                                    _stream()))
    return (nbr, ei, status, None, None) if (relative_position or degree_init) else (nbr, ei, status)


def undirected_degree(rowptr: torch.Tensor, col: torch.Tensor, n: int) -> torch.Tensor:
    _dev(rowptr, "rowptr", torch.int32)
    _dev(col, "col", torch.int32)
    deg = torch.empty(n, dtype=torch.int32, device=rowptr.device)
    check(lib.rgnn_undirected_degree(_ptr(rowptr), _ptr(col.contiguous()), n, None, _ptr(deg), _stream()))
    return deg


def knn_degree_from_csr(rowptr_t: torch.Tensor, src_sorted: torch.Tensor, node_order: Optional[torch.Tensor], nbr: torch.Tensor) -> torch.Tensor:
    """Undirected degree of a kNN graph (rows ``nbr`` int32 [n, k]) from its CSR by target: one launch, no atomics
    (rgnn_knn_degree_from_csr)."""
    _dev(nbr, "nbr", torch.int32)
    n, k = nbr.shape
    deg = torch.empty(n, dtype=torch.int32, device=nbr.device)
    check(lib.rgnn_knn_degree_from_csr(_ptr(rowptr_t), _ptr(src_sorted), _ptr(node_order), _ptr(nbr.contiguous()), n, k, _ptr(deg), _stream()))
    return deg


def undirected_degree_preset(rowptr: torch.Tensor, col: torch.Tensor, degree: torch.Tensor) -> torch.Tensor:
    """``undirected_degree`` when ``degree`` already holds the out-degrees (knn_graph(degree_init=True)): one launch."""
    check(lib.rgnn_undirected_degree_preset(_ptr(rowptr), _ptr(col.contiguous()), degree.numel(), _ptr(degree), _stream()))
    return degree


def invert_permutation(order: torch.Tensor) -> torch.Tensor:
    _dev(order, "order", torch.int32)
    rank = torch.empty_like(order)
    check(lib.rgnn_invert_permutation(_ptr(order.contiguous()), order.numel(), _ptr(rank), _stream()))
    return rank


def source_rowptr(edge_index: torch.Tensor, n: int, rank: Optional[torch.Tensor] = None) -> torch.Tensor:
    """int32 [n + 1]: out-degrees (by edge_index[0]) as a rowptr, positions in visiting order when ``rank`` is given."""
    _dev(edge_index, "edge_index", torch.int64)
    ei = edge_index.contiguous()
    e = ei.shape[1]
    rowptr_s = torch.empty(n + 1, dtype=torch.int32, device=ei.device)
    tmp = torch.empty(max(lib.rgnn_csr_by_target_tmp_bytes(n, 0), 256), dtype=torch.uint8, device=ei.device)
    check(lib.rgnn_source_rowptr(_ptr(ei), n, e, _ptr(rank), _ptr(rowptr_s), _ptr(tmp), _stream()))
    return rowptr_s


def csr_by_target(edge_index: torch.Tensor, n: int, target_rank: Optional[torch.Tensor] = None,
                  symmetric_rows: Optional[torch.Tensor] = None, status: Optional[torch.Tensor] = None, own_edges: bool = False,
                  ordered: bool = True):
    """-> rowptr_t int32 [n+1], src_sorted int32 [E], perm int32 [E]; with ``target_rank`` the segments are laid
    out in visiting order (segment p = edges into the node with rank p).

    ``symmetric_rows`` int32 [n+1]: the caller vouches that the graph is symmetric and that its edges are grouped by source
    with this rowptr, targets ascending inside a group (the output of the radius search) -- same result from three small
    kernels instead of six (rgnn_csr_by_target_symmetric); a missing twin edge sets STATUS_NOT_SYMMETRIC in ``status``."""
    _dev(edge_index, "edge_index", torch.int64)
    if edge_index.dim() != 2 or edge_index.shape[0] != 2:
        raise ValueError("edge_index must be [2,E]")
    ei = edge_index.contiguous()
    e = ei.shape[1]
    dev = ei.device
    rowptr_t = torch.empty(n + 1, dtype=torch.int32, device=dev)
    src = torch.empty(e, dtype=torch.int32, device=dev)
    perm = torch.empty(e, dtype=torch.int32, device=dev)
    tmp = torch.empty(max(lib.rgnn_csr_by_target_tmp_bytes(n, e), 256), dtype=torch.uint8, device=dev)
    if target_rank is not None:
        _dev(target_rank, "target_rank", torch.int32)
    if symmetric_rows is not None:
        _dev(symmetric_rows, "symmetric_rows", torch.int32)
        if symmetric_rows.numel() != n + 1:
            raise ValueError("symmetric_rows must be [n + 1]")
        if own_edges:
            # (no twin search: the third result is the OWN out-edge at the slot of every in-edge -- rgnn_csr_by_target_symmetric_own)
            check(lib.rgnn_csr_by_target_symmetric_own(_ptr(ei), _ptr(symmetric_rows), n, e, _ptr(target_rank), _ptr(rowptr_t),
                                                       _ptr(src), _ptr(perm), _ptr(tmp), _ptr(status), _stream()))
            return rowptr_t, src, perm
        check(lib.rgnn_csr_by_target_symmetric(_ptr(ei), _ptr(symmetric_rows), n, e, _ptr(target_rank), _ptr(rowptr_t),
                                               _ptr(src), _ptr(perm), _ptr(tmp), _ptr(status), _stream()))
        return rowptr_t, src, perm
    # ``ordered=False``: the order of a target's in-edges is whatever the fill's atomics made it (rgnn_csr_by_target_unordered) -- for
    # callers whose reduction does not depend on it (max aggregation): one pass over the edges less
    fn = lib.rgnn_csr_by_target if ordered else lib.rgnn_csr_by_target_unordered
    check(fn(_ptr(ei), n, e, _ptr(target_rank), _ptr(rowptr_t), _ptr(src), _ptr(perm), _ptr(tmp), _stream()))
    return rowptr_t, src, perm


def csr_by_target_frames(edge_index: torch.Tensor, n: int, k: int, frame_ptr: torch.Tensor, max_frame_points: int,
                         target_rank: Optional[torch.Tensor] = None):
    """``csr_by_target`` of a batch of kNN graphs (uniform out-degree ``k``, edge e = i k + j) in ONE launch, one block per frame
    (rgnn_csr_by_target_frames) -> rowptr_t, src_sorted, perm, in_degree int32 [n], frame_nonempty int32 [F]."""
    _dev(edge_index, "edge_index", torch.int64)
    ei = edge_index.contiguous()
    e = ei.shape[1]
    dev = ei.device
    rowptr_t = torch.empty(n + 1, dtype=torch.int32, device=dev)
    src = torch.empty(e, dtype=torch.int32, device=dev)
    perm = torch.empty(e, dtype=torch.int32, device=dev)
    tmp = torch.empty(max(e, 1), dtype=torch.int32, device=dev)
    indeg = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    f = frame_ptr.numel() - 1
    per_frame = torch.empty(f, dtype=torch.int32, device=dev)
    check(lib.rgnn_csr_by_target_frames(_ptr(ei), n, e, int(k), _ptr(frame_ptr.contiguous()), f, int(max_frame_points),
                                        _ptr(target_rank), _ptr(rowptr_t), _ptr(src), _ptr(perm), _ptr(tmp), _ptr(indeg),
                                        _ptr(per_frame), _stream()))
    return rowptr_t, src, perm, indeg, per_frame


CSR_FRAMES_MAX_POINTS = 24 * 1024          # rgnn_csr_by_target_frames: a frame's in-degree histogram lives in LDS


# ------------------------------------------------------------------------------------------------ features
def _codes(names: Sequence[str], table: dict, what: str):
    codes = []
    for nm in names:
        if nm not in table:
            raise Exception(f"Invalid {what}feature specified" if what else "Invalid feature specified")
        codes.append(table[nm])
    if len(codes) > 16:
        raise ValueError("at most 16 feature names")
    return (C.c_int32 * max(len(codes), 1))(*codes), len(codes)


def edge_features(X, V, edge_index, names: Sequence[str], edge_mode: str = "directed", dtype=torch.float32,
                  status: Optional[torch.Tensor] = None):
    _dev(X, "X", torch.float64); _dev(V, "V", torch.float64); _dev(edge_index, "edge_index", torch.int64)
    if edge_mode not in ("directed", "undirected"):
        raise ValueError(edge_mode)
    arr, n_codes = _codes(names, EDGE_FEATURE_CODES, "")
    width = sum(EDGE_FEATURE_WIDTH[nm] for nm in names)
    e = edge_index.shape[1]
    out = torch.empty((e, width), dtype=dtype, device=X.device)
    if status is None:
        status = torch.zeros(1, dtype=torch.int32, device=X.device)
    Xc, Vc = X[:, :2].contiguous(), V[:, :2].contiguous()
    check(lib.rgnn_edge_features(_ptr(Xc), _ptr(Vc), _ptr(edge_index.contiguous()), e, arr, n_codes,
                                 1 if edge_mode == "undirected" else 0, _ptr(out), 1 if dtype == torch.float64 else 0,
                                 _ptr(status), _stream()))
    return out, status


def edge_features_reversed(X, V, edge_index, reversed_of: torch.Tensor, names: Sequence[str], edge_mode: str = "directed",
                           dtype=torch.float32, status: Optional[torch.Tensor] = None):
    """Row s = the features of the REVERSE of edge ``reversed_of[s]`` (rgnn_edge_features_reversed): with TargetCSR.own_edge the
    attribute list of a symmetric graph in target order, without the twin search."""
    _dev(X, "X", torch.float64); _dev(V, "V", torch.float64); _dev(edge_index, "edge_index", torch.int64)
    _dev(reversed_of, "reversed_of", torch.int32)
    if edge_mode not in ("directed", "undirected"):
        raise ValueError(edge_mode)
    arr, n_codes = _codes(names, EDGE_FEATURE_CODES, "")
    width = sum(EDGE_FEATURE_WIDTH[nm] for nm in names)
    rows = reversed_of.numel()
    out = torch.empty((rows, width), dtype=dtype, device=X.device)
    if status is None:
        status = torch.zeros(1, dtype=torch.int32, device=X.device)
    check(lib.rgnn_edge_features_reversed(_ptr(X[:, :2].contiguous()), _ptr(V[:, :2].contiguous()), _ptr(edge_index.contiguous()),
                                          edge_index.shape[1], _ptr(reversed_of.contiguous()), rows, arr, n_codes,
                                          1 if edge_mode == "undirected" else 0, _ptr(out), 1 if dtype == torch.float64 else 0,
                                          _ptr(status), _stream()))
    return out, status


def node_features(X, V, rcs, time_index, degree, names: Sequence[str], dtype=torch.float32):
    _dev(X, "X", torch.float64)
    arr, n_codes = _codes(names, NODE_FEATURE_CODES, "node ")
    width = sum(NODE_FEATURE_WIDTH[nm] for nm in names)
    n = X.shape[0]
    out = torch.empty((n, width), dtype=dtype, device=X.device)
    f64 = lambda t: None if t is None else _dev(t, "feature", torch.float64).reshape(-1).contiguous()
    deg = None if degree is None else _dev(degree, "degree", torch.int32).contiguous()
    Vc = None if V is None else V[:, :2].contiguous()
    check(lib.rgnn_node_features(_ptr(X[:, :2].contiguous()), _ptr(Vc), _ptr(f64(rcs)), _ptr(f64(time_index)), _ptr(deg),
                                 n, arr, n_codes, _ptr(out), 1 if dtype == torch.float64 else 0, _stream()))
    return out


def node_features_time_index(X, V, rcs, timestamp, frame_ptr, degree, names: Sequence[str], dtype=torch.float32,
                             status: Optional[torch.Tensor] = None, frame_nonempty: Optional[torch.Tensor] = None):
    """``node_features`` with the time index computed on the way (one launch, one block per frame): same rows as
    ``node_features(..., time_index(timestamp, frame_ptr)[0], ...)``."""
    _dev(X, "X", torch.float64)
    arr, n_codes = _codes(names, NODE_FEATURE_CODES, "node ")
    width = sum(NODE_FEATURE_WIDTH[nm] for nm in names)
    n = X.shape[0]
    out = torch.empty((n, width), dtype=dtype, device=X.device)
    f64 = lambda t: None if t is None else _dev(t, "feature", torch.float64).reshape(-1).contiguous()
    deg = None if degree is None else _dev(degree, "degree", torch.int32).contiguous()
    Vc = None if V is None else V[:, :2].contiguous()
    if status is None:
        status = torch.zeros(1, dtype=torch.int32, device=X.device)
    _dev(frame_ptr, "frame_ptr", torch.int64)
    check(lib.rgnn_node_features_time_index(_ptr(X[:, :2].contiguous()), _ptr(Vc), _ptr(f64(rcs)), _ptr(f64(timestamp)),
                                            _ptr(frame_ptr.contiguous()), frame_ptr.numel() - 1, _ptr(deg), n, arr, n_codes,
                                            _ptr(out), 1 if dtype == torch.float64 else 0, _ptr(status), _ptr(frame_nonempty),
                                            _stream()))
    return out


def split_by_degree_frames(degree: torch.Tensor, frame_ptr: torch.Tensor, frame_nonempty: torch.Tensor):
    """``split_targets(..., by_node=True)`` of a symmetric graph from its degrees and the per-frame counts of
    ``node_features_time_index``: one launch (rgnn_split_by_degree_frames)."""
    n, dev = degree.numel(), degree.device
    lst = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    cnt = torch.empty(1, dtype=torch.int64, device=dev)
    slot = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    lst_ne = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    cnt_ne = torch.empty(1, dtype=torch.int64, device=dev)
    check(lib.rgnn_split_by_degree_frames(_ptr(degree), _ptr(frame_ptr.contiguous()), frame_ptr.numel() - 1, _ptr(frame_nonempty),
                                          _ptr(lst), _ptr(cnt), _ptr(slot), _ptr(lst_ne), _ptr(cnt_ne), _stream()))
    return lst, cnt, slot, lst_ne, cnt_ne


def radius_counts(deg: torch.Tensor, rowptr: torch.Tensor, threshold: int = 60) -> torch.Tensor:
    """int32 [2] on the device: (edge count rowptr[n], edges in rows longer than ``threshold``) -- rgnn_radius_counts, one launch."""
    _dev(deg, "deg", torch.int32); _dev(rowptr, "rowptr", torch.int32)
    out = torch.empty(2, dtype=torch.int32, device=deg.device)
    check(lib.rgnn_radius_counts(_ptr(deg.contiguous()), deg.numel(), _ptr(rowptr.contiguous()), threshold, _ptr(out), _stream()))
    return out


def time_index(timestamp: torch.Tensor, frame_ptr: torch.Tensor, status: Optional[torch.Tensor] = None, max_frame_points: int = 0):
    """``max_frame_points`` > 16 384 (the caller's host-side knowledge of its largest frame): the frames' points are spread over the
    chip (rgnn_time_index_ws) instead of one block per frame."""
    _dev(timestamp, "timestamp", torch.float64); _dev(frame_ptr, "frame_ptr", torch.int64)
    ts = timestamp.reshape(-1).contiguous()
    out = torch.empty_like(ts)
    if status is None:
        status = torch.zeros(1, dtype=torch.int32, device=ts.device)
    n_frames = frame_ptr.numel() - 1
    if max_frame_points > 16384 and n_frames <= 1024:
        ws = torch.empty(int(lib.rgnn_time_index_ws_bytes(n_frames)), dtype=torch.uint8, device=ts.device)
        check(lib.rgnn_time_index_ws(_ptr(ts), _ptr(frame_ptr.contiguous()), n_frames, ts.numel(), _ptr(out), _ptr(status), _ptr(ws),
                                     ws.numel(), _stream()))
        return out, status
    check(lib.rgnn_time_index(_ptr(ts), _ptr(frame_ptr.contiguous()), frame_ptr.numel() - 1, _ptr(out), _ptr(status),
                              _stream()))
    return out, status


# ------------------------------------------------------------------------------------------------ dense
STAT_ROWS = 4      # rgnn.h RGNN_STAT_ROWS: a column-statistics panel holds {count, pivot, sum (v - pivot), sum (v - pivot)^2} per column
AFFINE_ROWS = 3    # rgnn.h RGNN_AFFINE_ROWS: a BatchNorm-apply table holds {mean_hi, g, t} per column, y = (x - mean_hi) g + t


def stat_panels(m: int) -> int:
    return int(lib.rgnn_linear_stat_panels(m))


# bf16x3 path of the dense layer (linear.hip, k_linear_x3): the weight is handed over as three bf16 planes; splitting
# costs one small launch, so the planes are cached per (storage, version, geometry) of the weight tensors.
USE_BF16X3 = True
BF16X3_MIN_ROWS = 128
# layers with at most this many output columns stay on the fp32 MFMA kernel: the 3-way split of an activation tile is
# paid once per tile row whatever the tile's width, and a 64-column tile does not amortise it (C2: -1.4 % step time)
COUNTERS = {"fused_a1_affine": 0}          # launches per kind since import (tests check which path ran)
FUSE_A1_AFFINE = __import__("os").environ.get("RGNN_NO_FUSED_BN_APPLY") is None   # BatchNorm-apply inside the consumer GEMM's A path
USE_MAX_BWD = __import__("os").environ.get("RGNN_MPNN_BWD_OLD") is None     # max aggregation backward: rgnn_mpnn_max_bwd where it applies
BF16X3_MIN_COLS = int(__import__('os').environ.get('RGNN_X3_MIN_COLS', '32'))
_PLANES = {}
CACHE_EPOCH = 0          # part of every weight-derived cache key (planes here, folded weights in gnn/mpnn_layers.py)


_SPLITK_WS = {}
USE_STREAM_K = True
_SPLITK_SUPPRESS = 0


class no_splitk_workspace:
    """Dense launches inside go without the stream-K / split-K hand-over scratch (static tile schedule, undivided k-loops: same
    results).  For launches that may overlap launches of another stream which use the scratch -- a side stream inside a captured
    step resolves to the same buffer as the main stream."""

    def __enter__(self):
        global _SPLITK_SUPPRESS
        _SPLITK_SUPPRESS += 1

    def __exit__(self, *exc):
        global _SPLITK_SUPPRESS
        _SPLITK_SUPPRESS -= 1
        return False

# ---- f16x2 form of the dense kernel (linear_dma.hip, FMT 1): two f16 terms per operand, three MFMA products instead of six.
# It needs an upper bound of every activation operand in device memory (the exact power-of-two pre-scale is derived from it).
# While a BoundPool is active (`with bound_tracking(device)`: DetNetBasic's inference forward), the kernels that produce
# activations track max |out| into a word of the pool and the result tensor carries it as `_rgnn_bound`; a dense launch
# whose operands all carry a bound takes the f16x2 form, any other launch the bf16x3 form as before.
USE_F16X2 = __import__("os").environ.get("RGNN_NO_F16X2") is None
CHECK_BOUNDS = __import__("os").environ.get("RGNN_CHECK_BOUNDS") is not None
WGRAD_F16X2 = __import__("os").environ.get("RGNN_NO_WGRAD_F16X2") is None       # weight gradients in the f16x2 form inside a bound pool
BN_BWD_MASK_FROM_TABLE = __import__("os").environ.get("RGNN_BN_BWD_READS_Y") is None   # BatchNorm backward: ReLU mask from h + apply table
TRAIN_F16X2 = __import__("os").environ.get("RGNN_NO_TRAIN_F16X2") is None      # recorded (training) forwards and their backward track bounds too
_PLANES16 = {}


BOUND_SLOTS = 256        # rgnn.h RGNN_BOUND_SLOTS: a bound is 256 float words, producers raise one slot per work-group


class BoundPool:
    """Bounds (256 float32 slots each, rgnn.h) in device memory, zeroed by ONE fill per forward pass (inside a captured step:
    re-zeroed by every replay)."""

    def __init__(self, device, n: int = 96):
        self.buf = torch.zeros((n, BOUND_SLOTS), dtype=torch.float32, device=device)
        self.used = 0

    def word(self) -> Optional[torch.Tensor]:
        if self.used >= self.buf.shape[0]:
            return None                                  # (a very deep model: the remaining layers stay on the bf16x3 form)
        self.used += 1
        return self.buf[self.used - 1]


def make_bound(value: torch.Tensor) -> torch.Tensor:
    """A bound holding ``value`` (a scalar tensor on the device): for operands whose bound the caller knows."""
    b = torch.zeros(BOUND_SLOTS, dtype=torch.float32, device=value.device)
    b[0] = value
    return b


class bound_tracking:
    def __init__(self, device):
        self.device = device

    def __enter__(self):
        c = ctx()
        self.prev = c.bounds
        c.bounds = BoundPool(self.device) if USE_F16X2 else None
        return c.bounds

    def __exit__(self, *exc):
        ctx().bounds = self.prev
        return False


class using_bounds:
    """Put an existing pool in force on THIS thread: the backward pass of a recorded forward runs on autograd's device thread, whose
    ForwardContext knows nothing of the pool the forward pass tracked its bounds in (gnn/autograd.py keeps it on the node)."""

    def __init__(self, pool):
        self.pool = pool

    def __enter__(self):
        c = ctx()
        self.prev = c.bounds
        c.bounds = self.pool if USE_F16X2 else None
        return c.bounds

    def __exit__(self, *exc):
        ctx().bounds = self.prev
        return False


def bound_of(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    """The bound attached to ``t`` by the kernel that wrote it -- or None if ``t`` was written through torch since (its version
    counter moved): autograd's InputBuffer sums the gradients of a tensor with several consumers IN PLACE into the first one that
    arrived and hands on that same Python object, attribute and all -- a sum must not inherit one contributor's bound.
    librgnn's own launches write through raw pointers and leave the counter alone; whoever launches them calls ``set_bound``."""
    if t is None:
        return None
    word = getattr(t, "_rgnn_bound", None)
    if word is not None and getattr(t, "_rgnn_bound_version", None) != t._version:
        return None
    return word


def set_bound(t: torch.Tensor, word: Optional[torch.Tensor]) -> None:
    if word is not None:
        t._rgnn_bound = word
        t._rgnn_bound_version = t._version
    elif hasattr(t, "_rgnn_bound"):
        del t._rgnn_bound


def _splitk_ws(device, wanted: bool):
    """(pointer, bytes) of the stream-K / split-K scratch of the LDS-DMA dense kernel, one per (device, stream): allocated and
    zeroed once (the kernel keeps its flag words at zero between launches).  Launches of one stream run one after the other
    and share it; launches on DIFFERENT streams may overlap and must not exchange partial accumulators through the same slots."""
    if not (wanted and USE_STREAM_K) or _SPLITK_SUPPRESS:
        return None, 0
    key = (device, _stream_id(device))
    ws = _SPLITK_WS.get(key)
    if ws is None and torch.cuda.is_current_stream_capturing():
        # a capture runs on its own stream: take the scratch the eager pass before it used (allocating here would put a 64 MB
        # memset into every replay); a replay is ordered like the stream it is launched on
        ws = next((w for (d, _), w in _SPLITK_WS.items() if d == device), None)
    if ws is None and sum(1 for (d, _) in _SPLITK_WS if d == device) >= 8:
        # a process that keeps creating streams: no further 64 MB buffers -- its later streams go without the hand-over
        # workspace (static tile schedule, undivided k-loops; never a buffer another stream may still be using)
        return None, 0
    if ws is None:
        if torch.cuda.is_current_stream_capturing():
            # (no eager pass has run on this device yet: an allocation here would belong to the capturing graph's pool and put a
            #  64 MB memset into every replay -- go without the hand-over workspace: static tile schedule, same results)
            return None, 0
        ws = _SPLITK_WS[key] = torch.zeros(int(lib.rgnn_linear_splitk_ws_bytes()), dtype=torch.uint8, device=device)
    return ws.data_ptr(), ws.numel()


def splitk_timeouts(device) -> int:
    """How many times a dense launch on ``device`` gave up waiting for another work-group's partial tile (and computed a wrong
    one) since the scratch buffers were allocated: one host read per buffer.  0 in any healthy run."""
    dev = torch.device(device)
    total = 0
    for (d, _), ws in _SPLITK_WS.items():
        if torch.device(d) == dev or (torch.device(d).type == dev.type and dev.index is None):
            off = ws.numel() - 4096 + 4 * SPLITK_TIMEOUT_WORD
            total += int(ws[off:off + 4].view(torch.int32).item())
    return total


def radius_rows_commit(rowptr_new: torch.Tensor, n_edges: int, committed: torch.Tensor, status: torch.Tensor,
                       deg_new: Optional[torch.Tensor] = None, deg_committed: Optional[torch.Tensor] = None) -> torch.Tensor:
    """rgnn_radius_rows_commit: ``committed`` takes ``rowptr_new`` if its total is ``n_edges``, else keeps its rows and
    ``status`` gets STATUS_EDGE_COUNT_CHANGED (replayed steps: everything downstream of the search reads ``committed``)."""
    _dev(rowptr_new, "rowptr_new", torch.int32)
    _dev(committed, "committed", torch.int32)
    check(lib.rgnn_radius_rows_commit(_ptr(rowptr_new), rowptr_new.numel() - 1, int(n_edges), _ptr(committed), _ptr(status),
                                      _ptr(deg_new), _ptr(deg_committed), _stream()))
    return committed


def invalidate_weight_caches() -> None:
    """Drop everything derived from weight values.  The caches are keyed on tensor version counters, which in-place
    operations bump (optimizer steps, ``load_state_dict``, ``copy_`` under ``no_grad``) -- but writes through ``.data``
    do not: call this after such a write."""
    global CACHE_EPOCH
    CACHE_EPOCH += 1
    _PLANES.clear()
    _PLANES16.clear()
    _LATEST.clear()
    _LATEST16.clear()


_LATEST, _LATEST16 = {}, {}      # (weight views without their version) -> key of the newest cache entry made for them


def _replace_older(latest: dict, cache: dict, key) -> None:
    """A new entry for the same weight views supersedes the one made for an older version (an optimizer step bumps the version of
    every parameter: without this a training loop parks 18 dead plane sets per step until the 128-entry sweep)."""
    gkey = tuple(None if k is None else (k[0],) + k[2:] for k in key[:2])
    old = latest.get(gkey)
    if old is not None and old != key:
        cache.pop(old, None)
    latest[gkey] = key


def _wkey(w: Optional[torch.Tensor]):
    if w is None:
        return None, None
    st = w.untyped_storage()
    return st, (st._cdata, w._version, w.storage_offset(), tuple(w.shape), w.stride())


def _filled_on_this_stream():
    """(event, stream id) recorded behind the launch that just filled a cache entry: a later hit from ANOTHER stream waits for the
    event first (the same model driven on two streams; entries made inside a stream capture carry none -- a capture's own
    launches are ordered by the graph)."""
    st = torch.cuda.current_stream()
    if torch.cuda.is_current_stream_capturing():
        return None, st.cuda_stream
    ev = torch.cuda.Event()
    ev.record(st)
    return ev, st.cuda_stream


def _order_behind(ev, stream_id, planes: Optional[torch.Tensor] = None) -> None:
    """A cache hit from a stream other than the one that filled the entry: wait for the fill, and tell the caching allocator that
    THIS stream reads the planes too -- an entry evicted (replaced by a newer weight version, the 128-entry sweep,
    invalidate_weight_caches) while a launch of this stream still reads it must not have its block handed out again before that
    launch is through (the allocator only orders reuse on the allocating stream)."""
    if _stream_id() == stream_id:                     # (the common case, ~40 cache hits per step: no stream object is built)
        return
    st = torch.cuda.current_stream()
    if st.cuda_stream != stream_id:
        if ev is not None:
            st.wait_event(ev)
        if planes is not None and not torch.cuda.is_current_stream_capturing():
            planes.record_stream(st)


def weight_planes(w1: torch.Tensor, w2: Optional[torch.Tensor], k: int, cache: bool = True):
    """Three bf16 planes of [w1; w2] (rgnn_linear_split_weights), cached per weight storage, version (in-place updates
    bump it, also through detached aliases and views) and view geometry.  The entry keeps the storage alive, so its
    address cannot be recycled for another tensor while the entry exists; the cache is dropped when it reaches 128
    entries (temporaries such as the transposed weights of the backward pass)."""
    s1, k1_ = _wkey(w1)
    s2, k2_ = _wkey(w2)
    key = (k1_, k2_, CACHE_EPOCH)
    hit = _PLANES.get(key) if cache else None
    if hit is not None:
        _order_behind(hit[4], hit[5], hit[0])
        return hit[0], hit[1]
    if cache and len(_PLANES) >= 128:
        _PLANES.clear()
        _LATEST.clear()
    n1 = w1.shape[0]
    n = n1 + (0 if w2 is None else w2.shape[0])
    kp = int(lib.rgnn_linear_planes_kp(k))
    planes = torch.empty((3, n, kp), dtype=torch.bfloat16, device=w1.device)
    check(lib.rgnn_linear_split_weights(_ptr(w1), _ptr(w2), _ld(w1), n1, n, k, _ptr(planes), _stream()))
    if cache:
        _replace_older(_LATEST, _PLANES, key)
        _PLANES[key] = (planes, kp, s1, s2, *_filled_on_this_stream())
    return planes, kp


PAD_ROWS = __import__("os").environ.get("RGNN_NO_PADDED_ROWS") is None


def padded_rows(m: int, n: int, device) -> torch.Tensor:
    """An uninitialised float32 [m, n] matrix whose rows start on 128-byte lines: the row stride is n rounded up to 32 floats
    (464 -> 480).  For the intermediates of a conv layer that are written once and read once or gathered by row (the source
    term Q, the aggregated messages M): with a stride of 1 856 bytes every other 128-byte store segment and every row gather
    straddles two cache lines."""
    ld = (n + 31) // 32 * 32 if (PAD_ROWS and n % 32 and m > 0) else n
    if ld == n:
        return torch.empty((m, n), dtype=torch.float32, device=device)
    return torch.empty((m, ld), dtype=torch.float32, device=device)[:, :n]


def weight_planes_f16(w1: torch.Tensor, w2: Optional[torch.Tensor], k: int, cache: bool = True) -> torch.Tensor:
    """The two f16 planes of [w1; w2] * 2^sw plus their footer (rgnn_linear_split_weights_f16), cached like ``weight_planes``."""
    s1, k1_ = _wkey(w1)
    s2, k2_ = _wkey(w2)
    key = (k1_, k2_, CACHE_EPOCH)
    hit = _PLANES16.get(key) if cache else None
    if hit is not None:
        _order_behind(hit[3], hit[4], hit[0])
        return hit[0]
    if cache and len(_PLANES16) >= 128:
        _PLANES16.clear()
        _LATEST16.clear()
    n1 = w1.shape[0]
    n = n1 + (0 if w2 is None else w2.shape[0])
    planes = torch.empty(int(lib.rgnn_linear_planes_f16_bytes(n, k)), dtype=torch.uint8, device=w1.device)
    check(lib.rgnn_linear_split_weights_f16(_ptr(w1), _ptr(w2), _ld(w1), n1, n, k, _ptr(planes), _stream()))
    if cache:
        _replace_older(_LATEST16, _PLANES16, key)
        _PLANES16[key] = (planes, s1, s2, *_filled_on_this_stream())
    return planes


def linear(a1: torch.Tensor, w1: torch.Tensor, bias1: Optional[torch.Tensor] = None, *, a2: Optional[torch.Tensor] = None,
           w2: Optional[torch.Tensor] = None, bias2: Optional[torch.Tensor] = None, relu: bool = False,
           residual: Optional[torch.Tensor] = None, want_stats: bool = False, out: Optional[torch.Tensor] = None,
           row_index: Optional[torch.Tensor] = None, m_dev: Optional[torch.Tensor] = None, accumulate: bool = False,
           stats_out: Optional[torch.Tensor] = None, gather_only: bool = False,
           residual_index: Optional[torch.Tensor] = None, cache_planes: bool = True,
           a1_affine: Optional[torch.Tensor] = None, a1_relu: bool = True, relu_from: int = 0,
           a1_affine_tiles: Optional[torch.Tensor] = None, padded_row_list: bool = False):
    """out = act([a1|a2] @ [w1;w2]^T + [bias1;bias2]) (+ residual).  ``w1`` [n1, k1+k2] and ``w2`` [n2, k1+k2] may be
    column views of a larger weight (row stride = ldw).  Returns out or (out, col_stats).
    ``a1_affine`` float32 [AFFINE_ROWS, k1] (rows mean_hi, g, t of ``batchnorm_finalize``): the layer's first input block is
    act((a1 - mean_hi) * g + t) -- the BatchNorm (+ReLU) that precedes it -- applied inside the dense kernel where it can (rgnn_linear_fwd_fuses_a1_affine), by
    ``scale_shift_act`` in front of it otherwise.  ``relu_from``: with ``relu``, only the output columns >= relu_from are
    clamped (one launch for the first Linear of two heads)."""
    a1 = _rowmajor(_dev(a1, "a1", torch.float32), "a1")
    w1 = _rowmajor(_dev(w1, "w1", torch.float32), "w1")
    m, k1 = a1.shape
    k2 = 0
    if a2 is not None:
        a2 = _rowmajor(_dev(a2, "a2", torch.float32), "a2")
        if a2.shape[0] != m:
            raise ValueError("a1/a2 row mismatch")
        k2 = a2.shape[1]
    if w1.shape[1] != k1 + k2:
        raise ValueError(f"weight has {w1.shape[1]} input features, activations have {k1 + k2}")
    n1 = w1.shape[0]
    n = n1
    ldw = _ld(w1)
    if w2 is not None:
        w2 = _rowmajor(_dev(w2, "w2", torch.float32), "w2")
        if w2.shape[1] != k1 + k2:
            raise ValueError("w2 input width mismatch")
        if w2.shape[0] > 1 and w1.shape[0] > 1 and _ld(w2) != ldw:
            w2 = w2.contiguous(); w1 = w1.contiguous(); ldw = _ld(w1)
            if _ld(w2) != ldw:
                raise ValueError("w1/w2 row strides differ")
        n += w2.shape[0]
    if out is None:
        out = torch.empty((m, n), dtype=torch.float32, device=a1.device)
    else:
        _dev(out, "out", torch.float32)
        if out.shape != (m, n) or (n > 1 and out.stride(1) != 1):
            raise ValueError("bad `out`")
    stats = stats_out
    if want_stats and stats is None:
        stats = (torch.zeros if m == 0 else torch.empty)((max(stat_panels(m), 1), STAT_ROWS, n), dtype=torch.float32, device=a1.device)
    if row_index is not None:
        _dev(row_index, "row_index", torch.int32)
        _dev(m_dev, "m_dev", torch.int64)
        m = min(m, row_index.numel())                 # upper bound for the launch geometry; the kernel reads m_dev
    for b_, nm in ((bias1, "bias1"), (bias2, "bias2")):
        if b_ is not None:
            _dev(b_, nm, torch.float32)
            if not b_.is_contiguous():
                raise ValueError(f"{nm} must be contiguous")
    if residual is not None:
        residual = _rowmajor(_dev(residual, "residual", torch.float32), "residual")
    planes, kp = None, 0
    if (USE_BF16X3 and m >= BF16X3_MIN_ROWS and n > BF16X3_MIN_COLS and residual is None and (k1 + k2) % 4 == 0 and n % 4 == 0
            and (row_index is None or not (accumulate or gather_only or residual_index is not None))):
        planes, kp = weight_planes(w1, w2, k1 + k2, cache_planes)
    # f16x2 form: every activation block must carry a bound (the bound of an operand behind a1_affine lives on the table)
    b1 = bound_of(a1_affine) if a1_affine is not None else bound_of(a1)
    b2 = bound_of(a2)
    if CHECK_BOUNDS and not torch.cuda.is_current_stream_capturing():      # (a capture repeats launches an eager pass already checked)
        # debug net (RGNN_CHECK_BOUNDS=1; tests/test_gpu_threads.py): a bound travels as an attribute of the tensor it describes --
        # an in-place write that raises |values| behind it would make the f16x2 pre-scale overflow silently.  Verified here, at
        # the only place bounds are consumed (one reduction + host read per operand: never in a timed run).
        for t_, b_, nm in ((a1 if a1_affine is None else None, b1, "a1"), (a2, b2, "a2")):
            if t_ is not None and b_ is not None and t_.numel():
                rows = t_ if row_index is None else t_[row_index[:int(m_dev.item())].long()]
                if rows.numel() and float(rows.abs().max()) > float(b_.max()) * (1 + 1e-6):
                    raise RuntimeError(f"ops.linear: operand {nm} exceeds the bound attached to it "
                                       f"({float(rows.abs().max())} > {float(b_.max())})")
    track = ctx().bounds is not None and planes is not None
    def make_args(a1_, aff):
        return RgnnLinearArgs(_ptr(a1_), _ld(a1_), k1, _ptr(a2), 0 if a2 is None else _ld(a2), k2,
                              _ptr(w1), _ptr(w2), ldw, n1, _ptr(bias1), _ptr(bias2),
                              _ptr(residual), 0 if residual is None else _ld(residual),
                              _ptr(out), _ld(out) if out.shape[0] > 1 else n, m, n, 1 if relu else 0, _ptr(stats),
                              _ptr(row_index), _ptr(m_dev), 1 if accumulate else 0, 1 if gather_only else 0,
                              _ptr(residual_index), _ptr(planes), kp, *_splitk_ws(a1.device, planes is not None),
                              _ptr(aff), 1 if a1_relu else 0, int(relu_from), None, None, None, None,
                              _ptr(a1_affine_tiles) if aff is not None else None)

    def add_f16(args_) -> bool:
        """Fill in the f16x2 operands when the launch can take that form -- BEFORE asking whether the BatchNorm-apply fits the
        kernel's LDS next to its tiles: the two forms have different budgets (rgnn_linear_fwd_fuses_a1_affine)."""
        if (track and USE_F16X2 and b1 is not None and (a2 is None or b2 is not None)
                and lib.rgnn_linear_fwd_path(C.byref(args_)) != 0):
            planes16 = weight_planes_f16(w1, w2, k1 + k2, cache_planes)
            args_.W_planes_f16, args_.a1_bound, args_.a2_bound = _ptr(planes16), _ptr(b1), _ptr(b2)
            args_._keep = planes16
            return True
        return False

    if a1_affine_tiles is not None:
        # per-segment tables [S, AFFINE_ROWS, k1] (frames normalised with their own statistics): ``row_index`` is a list padded per segment
        # (pad_list_by_segment) and ``a1_affine_tiles`` names the table of each of its 256-row tiles.  No second form of this
        # launch exists: the caller decides beforehand whether its layers qualify (rgnn_linear_fwd_fuses_a1_affine).
        if a1_affine is None or row_index is None:
            raise ValueError("a1_affine_tiles goes with a1_affine [S, AFFINE_ROWS, k1] and a segment-padded row_index")
        a1_affine = _dev(a1_affine, "a1_affine", torch.float32).contiguous()
        _dev(a1_affine_tiles, "a1_affine_tiles", torch.int32)
        if a1_affine.dim() != 3 or a1_affine.shape[1:] != (AFFINE_ROWS, k1):
            raise ValueError("a1_affine must be [S, AFFINE_ROWS, k1] with a1_affine_tiles")
        args = make_args(a1, a1_affine)
        f16 = add_f16(args)
        if not lib.rgnn_linear_fwd_fuses_a1_affine(C.byref(args)):
            raise RgnnError("this launch cannot apply per-segment scale / shift tables (not on the LDS-DMA kernel)")
        COUNTERS["fused_a1_affine_segments"] = COUNTERS.get("fused_a1_affine_segments", 0) + 1
    elif a1_affine is not None:
        a1_affine = _dev(a1_affine, "a1_affine", torch.float32).contiguous()
        if a1_affine.shape != (AFFINE_ROWS, k1):
            raise ValueError("a1_affine must be [AFFINE_ROWS, k1]")
        args = make_args(a1, a1_affine)
        f16 = add_f16(args)
        if FUSE_A1_AFFINE and lib.rgnn_linear_fwd_fuses_a1_affine(C.byref(args)):
            COUNTERS["fused_a1_affine"] += 1
        else:
            # (a narrow / odd-width layer on another kernel: one pass over a1 first -- same arithmetic, same kernels as before)
            args = make_args(scale_shift_act(a1, a1_affine, relu=a1_relu), None)
            f16 = add_f16(args)
    else:
        args = make_args(a1, None)
        f16 = add_f16(args)
    if (padded_row_list or a1_affine_tiles is not None) and lib.rgnn_linear_fwd_path(C.byref(args)) == 0:
        # (``row_index`` may hold -1 entries -- a segment-padded list: only the LDS-DMA kernel skips them, any other kernel would
        #  read row -1)
        raise RgnnError("a segment-padded row list needs the LDS-DMA kernel, which this launch does not qualify for")
    word = None
    # (a row-subset launch that could not track max |out| leaves rows in `out` no bound covers -- a later launch
    #  into the same `out` must not attach a fresh word that only knows its own rows)
    if (track and lib.rgnn_linear_fwd_path(C.byref(args)) != 0       # the LDS-DMA kernel: it can track max |out|
            and not (row_index is not None and getattr(out, "_rgnn_rows_without_bound", False))):
        # (row-subset launches into a shared `out` -- the two halves of a conv layer's update -- share one word)
        word = bound_of(out) if row_index is not None else None
        if word is None:
            word = ctx().bounds.word()
        args.out_absmax = _ptr(word)
    tok = ctx().profiler.begin("linear") if ctx().profiler is not None else None
    check(lib.rgnn_linear_fwd(C.byref(args), _stream()))
    set_bound(out, word)                                  # (a launch that does not track invalidates an older bound)
    if row_index is not None and word is None:
        out._rgnn_rows_without_bound = True
    elif row_index is None and hasattr(out, "_rgnn_rows_without_bound"):
        del out._rgnn_rows_without_bound
    COUNTERS["f16x2" if f16 else "other_dense"] = COUNTERS.get("f16x2" if f16 else "other_dense", 0) + 1
    if tok is not None:
        ctx().profiler.end(tok, m=m if m_dev is None else m_dev, n=n, k=k1 + k2, x3=planes is not None, f16=f16)   # row subsets: true count lives on the device
    return (out, stats) if (want_stats or stats_out is not None) else out


def embed3(x: torch.Tensor, w1, b1, w2, b2, w3, b3, relu3: bool) -> Optional[torch.Tensor]:
    """act3(W3 relu(W2 relu(W1 x + b1) + b2) + b3) in one pass (rgnn_embed3: <= 8 inputs, layers of 32, 64 and 128 columns -- the
    shipped node embedding), or None when the shape does not qualify.  The result carries its bound."""
    x = _rowmajor(_dev(x, "x", torch.float32), "x")
    if not lib.rgnn_embed3_supported(x.shape[1], w1.shape[0], w2.shape[0], w3.shape[0]):
        return None
    if w1.shape[1] != x.shape[1] or w2.shape[1] != w1.shape[0] or w3.shape[1] != w2.shape[0]:
        raise ValueError("layer widths do not chain")
    w1 = _rowmajor(_dev(w1, "w1", torch.float32), "w1")
    p2 = weight_planes_f16(w2, None, w2.shape[1])
    p3 = weight_planes_f16(w3, None, w3.shape[1])
    m = x.shape[0]
    out = torch.empty((m, w3.shape[0]), dtype=torch.float32, device=x.device)
    word = ctx().bounds.word() if ctx().bounds is not None else None
    check(lib.rgnn_embed3(_ptr(x), _ld(x), x.shape[1], _ptr(w1), _ld(w1), _ptr(b1), w1.shape[0], _ptr(p2), _ptr(b2), w2.shape[0],
                          _ptr(p3), _ptr(b3), w3.shape[0], 1 if relu3 else 0, m, _ptr(out), _ld(out), _ptr(word), _stream()))
    set_bound(out, word)
    return out


def tiny_mlp2(a: torch.Tensor, row_index: Optional[torch.Tensor], w1, b1, relu1: bool, w2, b2, relu2: bool) -> torch.Tensor:
    """act2(W2 act1(W1 a[row_index] + b1) + b2) in one pass (rgnn_tiny_mlp2): <= 8 inputs, <= 8 hidden, <= 16 outputs."""
    a = _rowmajor(_dev(a, "a", torch.float32), "a")
    w1 = _rowmajor(_dev(w1, "w1", torch.float32), "w1"); w2 = _rowmajor(_dev(w2, "w2", torch.float32), "w2")
    m = a.shape[0] if row_index is None else row_index.numel()
    if row_index is not None:
        _dev(row_index, "row_index", torch.int32)
    out = torch.empty((m, w2.shape[0]), dtype=torch.float32, device=a.device)
    check(lib.rgnn_tiny_mlp2(_ptr(a), _ld(a), a.shape[1], _ptr(row_index), m, _ptr(w1), _ld(w1), _ptr(b1), w1.shape[0],
                             1 if relu1 else 0, _ptr(w2), _ld(w2), _ptr(b2), w2.shape[0], 1 if relu2 else 0, _ptr(out),
                             _ld(out), _stream()))
    return out


def empty_targets(rowptr_t: torch.Tensor, node_order: Optional[torch.Tensor]):
    """-> (list int32 [n] of node ids whose CSR segment is empty, count int64 [1] on the device, slot int32 [n]: the
    position of a node in that list or -1)."""
    _dev(rowptr_t, "rowptr_t", torch.int32)
    n = rowptr_t.numel() - 1
    dev = rowptr_t.device
    flags = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    pos = torch.empty(n + 1, dtype=torch.int32, device=dev)
    tmp = torch.empty(max(lib.rgnn_scan_tmp_bytes(n), 256), dtype=torch.uint8, device=dev)
    lst = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    cnt = torch.empty(1, dtype=torch.int64, device=dev)
    slot = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    check(lib.rgnn_empty_targets(_ptr(rowptr_t), _ptr(node_order), n, _ptr(flags), _ptr(pos), _ptr(tmp), _ptr(lst),
                                 _ptr(cnt), _ptr(slot), _stream()))
    return lst, cnt, slot


SORTED_ROW_LISTS = __import__("os").environ.get("RGNN_VISITING_ORDER_LISTS") is None


def split_targets(rowptr_t: torch.Tensor, node_order: Optional[torch.Tensor], rank: Optional[torch.Tensor] = None,
                  by_node: bool = False):
    """-> (empty list int32 [n], its count int64 [1], slot int32 [n], non-empty list int32 [n], its count int64 [1]); all on
    the device, lists in visiting order (rgnn_split_targets) or -- ``by_node`` with ``rank`` = the inverse of ``node_order`` (None
    when there is no visiting order) -- in ascending node order (rgnn_split_targets_by_node): the order the row-subset dense
    launches prefer."""
    _dev(rowptr_t, "rowptr_t", torch.int32)
    n = rowptr_t.numel() - 1
    dev = rowptr_t.device
    flags = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    pos = torch.empty(n + 1, dtype=torch.int32, device=dev)
    tmp = torch.empty(max(lib.rgnn_scan_tmp_bytes(n), 256), dtype=torch.uint8, device=dev)
    lst = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    cnt = torch.empty(1, dtype=torch.int64, device=dev)
    slot = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    lst_ne = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    cnt_ne = torch.empty(1, dtype=torch.int64, device=dev)
    if by_node and SORTED_ROW_LISTS and (node_order is None or rank is not None):
        if rank is not None:
            _dev(rank, "rank", torch.int32)
        check(lib.rgnn_split_targets_by_node(_ptr(rowptr_t), _ptr(rank), n, _ptr(flags), _ptr(pos), _ptr(tmp), _ptr(lst),
                                             _ptr(cnt), _ptr(slot), _ptr(lst_ne), _ptr(cnt_ne), _stream()))
        return lst, cnt, slot, lst_ne, cnt_ne
    check(lib.rgnn_split_targets(_ptr(rowptr_t), _ptr(node_order), n, _ptr(flags), _ptr(pos), _ptr(tmp), _ptr(lst), _ptr(cnt),
                                 _ptr(slot), _ptr(lst_ne), _ptr(cnt_ne), _stream()))
    return lst, cnt, slot, lst_ne, cnt_ne


def column_sums(x: torch.Tensor) -> torch.Tensor:
    """Column sums of ``x`` (float32 [n]) from its statistics panels, combined in float64."""
    return stats_to_sums(column_stats(x))[1].to(torch.float32)


def column_stats(x: torch.Tensor) -> torch.Tensor:
    """Column statistics of ``x`` in the panel layout of the dense epilogue: [panels, STAT_ROWS, n] per 128-row panel
    (rgnn_column_stats)."""
    x = _rowmajor(_dev(x, "x", torch.float32), "x")
    m, n = x.shape
    if m == 0:                                            # (no rows: a panel that counts nothing -- not an uninitialised one)
        return torch.zeros((1, STAT_ROWS, n), dtype=torch.float32, device=x.device)
    stats = torch.empty((max(stat_panels(m), 1), STAT_ROWS, n), dtype=torch.float32, device=x.device)
    check(lib.rgnn_column_stats(_ptr(x), _ld(x), m, n, _ptr(stats), _stream()))
    return stats


def stats_to_sums(stats: torch.Tensor):
    """(rows, column sums, column sums of squares) in float64 from statistics panels [panels, STAT_ROWS, n] -- what the
    {count, pivot, s1, s2} layout stands for (tests and tools; the BatchNorm kernels combine the panels themselves)."""
    st = stats.double()
    cnt, piv, s1, s2 = st[:, 0, :], st[:, 1, :], st[:, 2, :], st[:, 3, :]
    return cnt.sum(0), (s1 + cnt * piv).sum(0), (s2 + 2.0 * piv * s1 + cnt * piv * piv).sum(0)


def apply_table_reference(x: torch.Tensor, table: torch.Tensor) -> torch.Tensor:
    """(x - mean_hi) g + t in float64 for a BatchNorm-apply table [AFFINE_ROWS, n] (or one row of tables per row of x,
    [m, AFFINE_ROWS, n]): the arithmetic the kernels run in float32, for tests."""
    t = table.double()
    if t.dim() == 2:
        return (x.double() - t[0]) * t[1] + t[2]
    return (x.double() - t[:, 0]) * t[:, 1] + t[:, 2]


class StatParts:
    """Column statistics of one layer output left by up to two row-subset launches, each in a buffer of its own:
    ``parts`` = [(stats [panels, STAT_ROWS, C], row count int64 [1] on the device or None), ...].  Only the panels a launch really
    wrote are summed (rgnn_batchnorm_finalize_parts), so the buffers are allocated uninitialised."""
    __slots__ = ("parts",)

    def __init__(self, parts):
        if not 1 <= len(parts) <= 2:
            raise ValueError("one or two parts")
        self.parts = list(parts)

    @property
    def device(self):
        return self.parts[0][0].device


def batchnorm_finalize(stats, m: int, n: int, gamma, beta, running_mean, running_var,
                       num_batches_tracked, training: bool, momentum: float, eps: float,
                       in_bound: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``in_bound``: device word bounding |x| of the matrix the layer normalises; with an active BoundPool the returned table then
    carries the bound of |(x - mean_hi) g + t| (``_rgnn_bound``), which lets the dense layer that applies it run in the f16x2 form."""
    dev = (stats if stats is not None else running_mean).device
    ss = torch.empty((AFFINE_ROWS, n), dtype=torch.float32, device=dev)
    out_bound = ctx().bounds.word() if (ctx().bounds is not None and in_bound is not None) else None
    if out_bound is not None:
        if isinstance(stats, StatParts):
            (sa, ra), (sb, rb) = stats.parts[0], (stats.parts[1] if len(stats.parts) > 1 else (None, None))
        else:
            sa, ra, sb, rb = stats, None, None, None
        check(lib.rgnn_batchnorm_finalize_bound(_ptr(sa), 0 if sa is None else sa.shape[0], _ptr(ra), _ptr(sb),
                                                0 if sb is None else sb.shape[0], _ptr(rb), m, n, _ptr(gamma), _ptr(beta),
                                                _ptr(running_mean), _ptr(running_var), _ptr(num_batches_tracked),
                                                1 if training else 0, float(momentum), float(eps), _ptr(ss), _ptr(in_bound),
                                                _ptr(out_bound), _stream()))
        set_bound(ss, out_bound)
        return ss
    if isinstance(stats, StatParts):
        (sa, ra), (sb, rb) = stats.parts[0], (stats.parts[1] if len(stats.parts) > 1 else (None, None))
        check(lib.rgnn_batchnorm_finalize_parts(_ptr(sa), sa.shape[0], _ptr(ra), _ptr(sb), 0 if sb is None else sb.shape[0],
                                                _ptr(rb), m, n, _ptr(gamma), _ptr(beta), _ptr(running_mean),
                                                _ptr(running_var), _ptr(num_batches_tracked), 1 if training else 0,
                                                float(momentum), float(eps), _ptr(ss), _stream()))
        return ss
    panels = 0 if stats is None else stats.shape[0]
    check(lib.rgnn_batchnorm_finalize(_ptr(stats), panels, m, n, _ptr(gamma), _ptr(beta), _ptr(running_mean),
                                      _ptr(running_var), _ptr(num_batches_tracked), 1 if training else 0,
                                      float(momentum), float(eps), _ptr(ss), _stream()))
    return ss


def scale_shift_act(x: torch.Tensor, scale_shift: torch.Tensor, relu: bool, out: Optional[torch.Tensor] = None):
    x = _rowmajor(_dev(x, "x", torch.float32), "x")
    m, n = x.shape
    if out is None:
        out = torch.empty((m, n), dtype=torch.float32, device=x.device)
    check(lib.rgnn_scale_shift_act(_ptr(x), _ld(x), _ptr(scale_shift), m, n, 1 if relu else 0, _ptr(out), _ld(out),
                                   _stream()))
    set_bound(out, bound_of(scale_shift))                 # (the table's bound is that of |(x - mean_hi) g + t|)
    return out


def batchnorm_segments(x: torch.Tensor, seg_ptr: torch.Tensor, gamma, beta, running_mean, running_var, num_batches_tracked,
                       momentum: float, eps: float) -> torch.Tensor:
    """Apply table [F, AFFINE_ROWS, C] of a train-mode BatchNorm whose statistics are taken per segment of rows
    (rgnn_batchnorm_segments; ``seg_ptr`` int64 [F + 1] on the device: one segment per frame).  Running statistics, when given,
    are updated segment after segment.  With an active BoundPool and a bound on ``x`` the table carries the bound of the
    normalised values."""
    x = _rowmajor(_dev(x, "x", torch.float32), "x")
    _dev(seg_ptr, "seg_ptr", torch.int64)
    m, n = x.shape
    f = seg_ptr.numel() - 1
    table = torch.empty((f, AFFINE_ROWS, n), dtype=torch.float32, device=x.device)
    sums = torch.empty((f, 2, n), dtype=torch.float64, device=x.device)
    in_bound = bound_of(x)
    out_bound = ctx().bounds.word() if (ctx().bounds is not None and in_bound is not None) else None
    check(lib.rgnn_batchnorm_segments(_ptr(x), _ld(x), _ptr(seg_ptr.contiguous()), f, n, _ptr(gamma), _ptr(beta), _ptr(running_mean),
                                      _ptr(running_var), _ptr(num_batches_tracked), float(momentum), float(eps), _ptr(sums),
                                      _ptr(table), _ptr(in_bound) if out_bound is not None else None, _ptr(out_bound), _stream()))
    set_bound(table, out_bound)
    return table


def pad_list_by_segment(lst: torch.Tensor, count: torch.Tensor, seg_ptr: torch.Tensor):
    """An ascending int32 row list (``count`` int64 [1] on the device: how many entries are live) cut at the segment borders
    ``seg_ptr`` int64 [S + 1] and padded with -1 so that every segment starts a 256-row tile of its own (rgnn_pad_list_by_segment)
    -> (list int32 [len(lst) + 256 S], its live length int64 [1], tile_segment int32 [tiles], stat_panel_start int32 [S + 1])."""
    _dev(lst, "lst", torch.int32); _dev(count, "count", torch.int64); _dev(seg_ptr, "seg_ptr", torch.int64)
    s = seg_ptr.numel() - 1
    cap = (lst.numel() + 256 * s + 255) // 256 * 256
    out = torch.empty(cap, dtype=torch.int32, device=lst.device)
    total = torch.empty(1, dtype=torch.int64, device=lst.device)
    tiles = torch.empty(cap // 256, dtype=torch.int32, device=lst.device)
    start = torch.empty(s + 1, dtype=torch.int32, device=lst.device)
    check(lib.rgnn_pad_list_by_segment(_ptr(lst.contiguous()), _ptr(count), _ptr(seg_ptr.contiguous()), s, _ptr(out), _ptr(total),
                                       _ptr(tiles), _ptr(start), _stream()))
    return out, total, tiles, start


def pad_list_pair_by_segment(lst_a, count_a, lst_b, count_b, seg_ptr):
    """``pad_list_by_segment`` for two lists over the same segments in one launch -> (result of a, result of b)."""
    for t_, nm, dt in ((lst_a, "lst_a", torch.int32), (lst_b, "lst_b", torch.int32), (count_a, "count_a", torch.int64),
                       (count_b, "count_b", torch.int64), (seg_ptr, "seg_ptr", torch.int64)):
        _dev(t_, nm, dt)
    s = seg_ptr.numel() - 1
    res = []
    for lst in (lst_a, lst_b):
        cap = (lst.numel() + 256 * s + 255) // 256 * 256
        res.append((torch.empty(cap, dtype=torch.int32, device=lst.device), torch.empty(1, dtype=torch.int64, device=lst.device),
                    torch.empty(cap // 256, dtype=torch.int32, device=lst.device), torch.empty(s + 1, dtype=torch.int32, device=lst.device)))
    (oa, ta, tia, sa), (ob, tb, tib, sb) = res
    check(lib.rgnn_pad_list_pair_by_segment(_ptr(lst_a.contiguous()), _ptr(count_a), _ptr(lst_b.contiguous()), _ptr(count_b),
                                            _ptr(seg_ptr.contiguous()), s, _ptr(oa), _ptr(ta), _ptr(tia), _ptr(sa), _ptr(ob), _ptr(tb),
                                            _ptr(tib), _ptr(sb), _stream()))
    return res[0], res[1]


def batchnorm_segments_from_panels(stats_a: torch.Tensor, start_a: torch.Tensor, stats_b: Optional[torch.Tensor],
                                   start_b: Optional[torch.Tensor], seg_ptr: torch.Tensor, gamma, beta, running_mean, running_var,
                                   num_batches_tracked, momentum: float, eps: float, in_bound=None) -> torch.Tensor:
    """The [S, AFFINE_ROWS, C] apply table of ``batchnorm_segments`` from the per-panel column statistics that dense launches on
    segment-padded row lists left behind (rgnn_batchnorm_segments_from_panels): no pass over the activations."""
    _dev(stats_a, "stats_a", torch.float32); _dev(start_a, "start_a", torch.int32); _dev(seg_ptr, "seg_ptr", torch.int64)
    n = stats_a.shape[-1]
    s = seg_ptr.numel() - 1
    table = torch.empty((s, AFFINE_ROWS, n), dtype=torch.float32, device=stats_a.device)
    sums = torch.empty((s, 2, n), dtype=torch.float64, device=stats_a.device)
    out_bound = ctx().bounds.word() if (ctx().bounds is not None and in_bound is not None) else None
    check(lib.rgnn_batchnorm_segments_from_panels(_ptr(stats_a), _ptr(start_a), _ptr(stats_b), _ptr(start_b),
                                                  _ptr(seg_ptr.contiguous()), s, n, _ptr(gamma), _ptr(beta), _ptr(running_mean),
                                                  _ptr(running_var), _ptr(num_batches_tracked), float(momentum), float(eps),
                                                  _ptr(sums), _ptr(table), _ptr(in_bound) if out_bound is not None else None,
                                                  _ptr(out_bound), _stream()))
    set_bound(table, out_bound)
    return table


def batchnorm_act_segments(x: torch.Tensor, seg_ptr: torch.Tensor, gamma, beta, running_mean, running_var, num_batches_tracked,
                           momentum: float, eps: float, relu: bool) -> torch.Tensor:
    """act(BatchNorm(x)) with statistics per segment of rows: ``scale_shift_act_segments(x, batchnorm_segments(x, ...))`` in two
    launches instead of three (rgnn_batchnorm_act_segments); the bound of the result travels with it."""
    x = _rowmajor(_dev(x, "x", torch.float32), "x")
    _dev(seg_ptr, "seg_ptr", torch.int64)
    m, n = x.shape
    f = seg_ptr.numel() - 1
    table = torch.empty((f, AFFINE_ROWS, n), dtype=torch.float32, device=x.device)
    sums = torch.empty((f, 2, n), dtype=torch.float64, device=x.device)
    out = torch.empty((m, n), dtype=torch.float32, device=x.device)
    in_bound = bound_of(x)
    out_bound = ctx().bounds.word() if (ctx().bounds is not None and in_bound is not None) else None
    check(lib.rgnn_batchnorm_act_segments(_ptr(x), _ld(x), _ptr(seg_ptr.contiguous()), f, n, _ptr(gamma), _ptr(beta),
                                          _ptr(running_mean), _ptr(running_var), _ptr(num_batches_tracked), float(momentum),
                                          float(eps), 1 if relu else 0, _ptr(sums), _ptr(table), _ptr(out), _ld(out),
                                          _ptr(in_bound) if out_bound is not None else None, _ptr(out_bound), _stream()))
    set_bound(out, out_bound)
    return out


def scale_shift_act_segments(x: torch.Tensor, table: torch.Tensor, seg_ptr: torch.Tensor, relu: bool) -> torch.Tensor:
    x = _rowmajor(_dev(x, "x", torch.float32), "x")
    m, n = x.shape
    out = torch.empty((m, n), dtype=torch.float32, device=x.device)
    check(lib.rgnn_scale_shift_act_segments(_ptr(x), _ld(x), _ptr(table), _ptr(seg_ptr.contiguous()), seg_ptr.numel() - 1, m, n,
                                            1 if relu else 0, _ptr(out), _ld(out), _stream()))
    set_bound(out, bound_of(table))
    return out


def softmax_rows(x: torch.Tensor) -> torch.Tensor:
    x = _rowmajor(_dev(x, "x", torch.float32), "x")
    y = torch.empty_like(x, memory_format=torch.contiguous_format)
    check(lib.rgnn_softmax_rows(_ptr(x), _ld(x), x.shape[0], x.shape[1], _ptr(y), _ld(y), _stream()))
    return y


# ------------------------------------------------------------------------------------------------ message passing
def gather_rows(x: torch.Tensor, perm: torch.Tensor) -> torch.Tensor:
    x = _rowmajor(_dev(x, "x", torch.float32), "x")
    _dev(perm, "perm", torch.int32)
    out = torch.empty((perm.numel(), x.shape[1]), dtype=torch.float32, device=x.device)
    check(lib.rgnn_gather_rows_f32(_ptr(x), _ld(x), _ptr(perm), perm.numel(), x.shape[1], _ptr(out), max(x.shape[1], 1),
                                   _stream()))
    return out


def _mp_common(P, p_bias, Q, We, ea_sorted, rowptr_t, src_sorted):
    Q = _rowmajor(_dev(Q, "Q", torch.float32), "Q")
    if P is not None:
        P = _rowmajor(_dev(P, "P", torch.float32), "P")
    de = 0
    if ea_sorted is not None and ea_sorted.shape[1] > 0:
        ea_sorted = _dev(ea_sorted, "edge_attr_sorted", torch.float32).contiguous()
        We = _rowmajor(_dev(We, "We", torch.float32), "We")
        de = ea_sorted.shape[1]
        if We.shape[1] != de:
            raise ValueError("We / edge_attr width mismatch")
    else:
        ea_sorted, We = None, None
    _dev(rowptr_t, "rowptr_t", torch.int32); _dev(src_sorted, "src_sorted", torch.int32)
    if ea_sorted is not None and ea_sorted.shape[0] == 0:
        ea_sorted, We, de = None, None, 0             # a graph without edges: every segment is empty, nothing to gather
    return P, Q, We, ea_sorted, de


def mpnn_partition(rowptr_t: torch.Tensor, n_edges: int) -> torch.Tensor:
    """Work-balanced chunk boundaries over the CSR-by-target (int32 [n_chunks + 1])."""
    _dev(rowptr_t, "rowptr_t", torch.int32)
    n = rowptr_t.numel() - 1
    nc = int(lib.rgnn_mpnn_num_chunks(n, n_edges))
    out = torch.empty(nc + 1 + 1024, dtype=torch.int32, device=rowptr_t.device)   # table + ticket counters
    check(lib.rgnn_mpnn_partition(_ptr(rowptr_t), n, n_edges, _ptr(out), _stream()))
    return out


def mpnn_aggregate(P, p_bias, Q, We, ea_sorted, rowptr_t, src_sorted, aggr: str,
                   node_order: Optional[torch.Tensor] = None, chunks: Optional[torch.Tensor] = None,
                   skip_empty_rows: bool = False) -> torch.Tensor:
    """m[t] = P[t] (+p_bias) (.) reduce_{e -> t}( Q[src_e] + We a_e ); empty segments -> 0.
    ``skip_empty_rows``: the caller never reads the rows of targets without incoming edges; they may stay unwritten."""
    P, Q, We, ea_sorted, de = _mp_common(P, p_bias, Q, We, ea_sorted, rowptr_t, src_sorted)
    n, d = rowptr_t.numel() - 1, Q.shape[1]
    out = padded_rows(n, d, Q.device)                        # rows start on 128-byte lines (see padded_rows)
    word = ctx().bounds.word() if ctx().bounds is not None else None     # max |out|: the update GEMM's A2 bound (f16x2 form)
    tok = ctx().profiler.begin("mpnn_aggregate") if ctx().profiler is not None else None
    check(lib.rgnn_mpnn_aggregate_absmax(_ptr(P), 0 if P is None else _ld(P), _ptr(p_bias), _ptr(Q), _ld(Q), _ptr(We),
                                         0 if We is None else _ld(We), _ptr(ea_sorted), de, _ptr(rowptr_t), _ptr(src_sorted if src_sorted.numel() else rowptr_t),
                                         _ptr(node_order), _ptr(chunks), 0 if chunks is None else chunks.numel() - 1025, n, d,
                                         AGGR_CODES[aggr], _ptr(out), _ld(out), 1 if skip_empty_rows else 0, _ptr(word), _stream()))
    if tok is not None:
        ctx().profiler.end(tok, n=n, d=d, de=de, e=src_sorted.numel())
    set_bound(out, word)
    return out


def mpnn_win_plan_buffer(rowptr_t: torch.Tensor, src_sorted: torch.Tensor) -> torch.Tensor:
    return torch.empty(int(lib.rgnn_mpnn_win_plan_ints(rowptr_t.numel() - 1, src_sorted.numel())), dtype=torch.int32, device=rowptr_t.device)


def mpnn_win_plan(rowptr_t: torch.Tensor, src_sorted: torch.Tensor, node_order: Optional[torch.Tensor] = None,
                  out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The window plan of a graph for ``mpnn_aggregate_win`` (rgnn_mpnn_win_plan): once per graph, no host synchronisation."""
    n, e = rowptr_t.numel() - 1, src_sorted.numel()
    plan = out if out is not None else mpnn_win_plan_buffer(rowptr_t, src_sorted)
    check(lib.rgnn_mpnn_win_plan(_ptr(rowptr_t), _ptr(src_sorted if e else rowptr_t), _ptr(node_order), n, e, _ptr(plan), _stream()))
    return plan


WIN_IMAGE_CACHE = os.environ.get("RGNN_NO_WIN_IMAGE_CACHE") is None


def _win_operand_image(We: Optional[torch.Tensor], p_bias: Optional[torch.Tensor], d: int, de: int) -> Optional[torch.Tensor]:
    """The window kernel's operand image of (We, p_bias) (rgnn_mpnn_win_wplanes), kept ON the We tensor object -- the layers hand in
    their cached folded weights, so the image lives exactly as long as the weights it was made from and is rebuilt when they were
    written in place or come with another bias.  None: no edge weights, or a first sight inside a stream capture (an image made there
    would belong to that graph's pool) -- the launch then builds its own image in the plan's scratch."""
    if We is None or de == 0 or not WIN_IMAGE_CACHE:
        return None
    hit = getattr(We, "_rgnn_wplanes", None)
    if (hit is not None and hit[1] == We._version and hit[2] is p_bias and (p_bias is None or hit[3] == p_bias._version)
            and hit[4] == CACHE_EPOCH):
        _order_behind(hit[5], hit[6], hit[0])
        return hit[0]
    if torch.cuda.is_current_stream_capturing():
        return None
    planes = torch.empty(int(lib.rgnn_mpnn_win_wplanes_bytes(d)), dtype=torch.uint8, device=We.device)
    check(lib.rgnn_mpnn_win_wplanes(_ptr(We), _ld(We), de, d, _ptr(p_bias), _ptr(planes), _stream()))
    ev, sid = _filled_on_this_stream()
    We._rgnn_wplanes = (planes, We._version, p_bias, None if p_bias is None else p_bias._version, CACHE_EPOCH, ev, sid)
    return planes


def mpnn_aggregate_win(p_bias, Q, We, ea_sorted, rowptr_t, src_sorted, plan: torch.Tensor,
                       node_order: Optional[torch.Tensor] = None, skip_empty_rows: bool = False) -> torch.Tensor:
    """m[t] = p_bias + max_{e -> t}(Q[src_e] + We a_e) by the window kernel (rgnn_mpnn_aggregate_win): distinct source rows of a
    window of targets staged in LDS, MFMA mat-vec on exact three-term bf16 splits; de <= 8."""
    n, d = rowptr_t.numel() - 1, Q.shape[1]
    de = 0 if ea_sorted is None else ea_sorted.shape[1]
    out = padded_rows(n, d, Q.device)
    word = ctx().bounds.word() if ctx().bounds is not None else None
    tok = ctx().profiler.begin("mpnn_aggregate") if ctx().profiler is not None else None
    planes = _win_operand_image(We, p_bias, d, de)
    common = (_ptr(p_bias), _ptr(Q), _ld(Q), _ptr(We), 0 if We is None else _ld(We), _ptr(ea_sorted), de,
              _ptr(rowptr_t), _ptr(src_sorted if src_sorted.numel() else rowptr_t), _ptr(node_order), _ptr(plan), n,
              src_sorted.numel(), d, _ptr(out), _ld(out), 1 if skip_empty_rows else 0, _ptr(word))
    if planes is not None:
        check(lib.rgnn_mpnn_aggregate_win_planes(*common, _ptr(planes), _stream()))
    else:
        check(lib.rgnn_mpnn_aggregate_win(*common, _stream()))
    if tok is not None:
        ctx().profiler.end(tok, n=n, d=d, de=de, e=src_sorted.numel(), win=True)
    COUNTERS["mpnn_win"] = COUNTERS.get("mpnn_win", 0) + 1
    set_bound(out, word)
    return out


def mpnn_aggregate_max_arg(p_bias, Q, We, ea_sorted, rowptr_t, src_sorted, node_order=None, chunks=None,
                           skip_empty_rows: bool = False):
    """Max aggregation that also records the winning edge per (target, channel) for the backward pass
    (rgnn_mpnn_aggregate_max_arg) -> (M [n, d], arg int16 [n, d] (uint16 in-segment indices) or None when the fused max kernel
    does not cover the shape).  The caller guarantees in-degrees below 65 536 (TargetCSR.edge_maps())."""
    _, Q, We, ea_sorted, de = _mp_common(None, p_bias, Q, We, ea_sorted, rowptr_t, src_sorted)
    n, d = rowptr_t.numel() - 1, Q.shape[1]
    out = torch.empty((n, d), dtype=torch.float32, device=Q.device)
    arg = torch.empty((n, d), dtype=torch.int16, device=Q.device)
    written = C.c_int32(0)
    word = ctx().bounds.word() if ctx().bounds is not None else None     # max |out|: the update GEMM's A2 bound (f16x2 form)
    tok = ctx().profiler.begin("mpnn_aggregate") if ctx().profiler is not None else None
    check(lib.rgnn_mpnn_aggregate_max_arg_absmax(_ptr(p_bias), _ptr(Q), _ld(Q), _ptr(We), 0 if We is None else _ld(We), _ptr(ea_sorted),
                                                 de, _ptr(rowptr_t), _ptr(src_sorted if src_sorted.numel() else rowptr_t), _ptr(node_order),
                                                 _ptr(chunks), 0 if chunks is None else chunks.numel() - 1025, n, d, _ptr(out), d, _ptr(arg),
                                                 1 if skip_empty_rows else 0, C.byref(written), _ptr(word), _stream()))
    if tok is not None:
        ctx().profiler.end(tok, n=n, d=d, de=de, e=src_sorted.numel())
    set_bound(out, word if (written.value & 2) else None)
    return out, (arg if (written.value & 1) else None)


def mpnn_edge_hidden(P, p_bias, Q, We, ea_sorted, rowptr_t, src_sorted, relu: bool,
                     node_order: Optional[torch.Tensor] = None, chunks: Optional[torch.Tensor] = None) -> torch.Tensor:
    P, Q, We, ea_sorted, de = _mp_common(P, p_bias, Q, We, ea_sorted, rowptr_t, src_sorted)
    n, d = rowptr_t.numel() - 1, Q.shape[1]
    e = src_sorted.numel()
    out = torch.empty((e, d), dtype=torch.float32, device=Q.device)
    if e == 0:                                            # (a graph without edges: nothing to compute -- and no buffer to hand over)
        return out
    check(lib.rgnn_mpnn_edge_hidden(_ptr(P), 0 if P is None else _ld(P), _ptr(p_bias), _ptr(Q), _ld(Q), _ptr(We),
                                    0 if We is None else _ld(We), _ptr(ea_sorted), de, _ptr(rowptr_t), _ptr(src_sorted if src_sorted.numel() else rowptr_t),
                                    _ptr(node_order), _ptr(chunks), 0 if chunks is None else chunks.numel() - 1025, n, d,
                                    1 if relu else 0, _ptr(out), d, _stream()))
    return out


def segment_reduce(rows: torch.Tensor, rowptr_t: torch.Tensor, aggr: str,
                   node_order: Optional[torch.Tensor] = None) -> torch.Tensor:
    rows = _rowmajor(_dev(rows, "rows", torch.float32), "rows")
    n, d = rowptr_t.numel() - 1, rows.shape[1]
    if rows.shape[0] == 0:                                # (no rows: every segment is empty -> exactly 0, torch-scatter's convention)
        return torch.zeros((n, d), dtype=torch.float32, device=rows.device)
    out = torch.empty((n, d), dtype=torch.float32, device=rows.device)
    check(lib.rgnn_segment_reduce(_ptr(rows), _ld(rows), _ptr(rowptr_t), _ptr(node_order), n, d, AGGR_CODES[aggr],
                                  _ptr(out), d, _stream()))
    return out


# ------------------------------------------------------------------------------------------------ backward pass
def relu_bwd(dy: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """dx = (y > 0) ? dy : 0 (ReLU fused into a forward epilogue; ``y`` is the saved output)."""
    dy = _dev(dy, "dy", torch.float32).contiguous()
    y = _dev(y, "y", torch.float32).contiguous()
    dx = torch.empty_like(dy)
    check(lib.rgnn_relu_bwd(_ptr(dy), _ptr(y), _ptr(dx), dy.numel(), _stream()))
    return dx


def bn_bwd_stats(dy: torch.Tensor, y: Optional[torch.Tensor], h: torch.Tensor, table: Optional[torch.Tensor] = None) -> torch.Tensor:
    """-> float32 [panels, 2, C]: column sums of g and g*h with g = dy masked by (y > 0) when ``y`` is given -- or, with ``table``
    (the [AFFINE_ROWS, C] apply table of the forward pass), by fmaf(h - mean_hi, g, t) > 0: the same bits without reading y."""
    dy = _rowmajor(_dev(dy, "dy", torch.float32), "dy")
    h = _rowmajor(_dev(h, "h", torch.float32), "h")
    if y is not None:
        y = _rowmajor(_dev(y, "y", torch.float32), "y")
    if table is not None:
        table = _dev(table, "table", torch.float32).contiguous()
    m, n = h.shape
    part = torch.empty((max(stat_panels(m), 1), 2, n), dtype=torch.float32, device=h.device)
    check(lib.rgnn_bn_bwd_stats_table(_ptr(dy), _ld(dy), _ptr(y), 0 if y is None else _ld(y), _ptr(table), _ptr(h), _ld(h), m, n,
                                      _ptr(part), _stream()))
    return part


def bn_bwd_coef(fwd_stats: Optional[torch.Tensor], running_mean, running_var, bwd_part: torch.Tensor, m: int, gamma, eps: float,
                use_batch: bool):
    """-> (coef float32 [3, C] for bn_bwd_apply, d gamma [C], d beta [C]) in one launch (rgnn_bn_bwd_coef)."""
    n = bwd_part.shape[2]
    dev = bwd_part.device
    coef = torch.empty((3, n), dtype=torch.float32, device=dev)
    dgamma = torch.empty(n, dtype=torch.float32, device=dev)
    dbeta = torch.empty(n, dtype=torch.float32, device=dev)
    check(lib.rgnn_bn_bwd_coef(_ptr(fwd_stats), 0 if fwd_stats is None else fwd_stats.shape[0], _ptr(running_mean), _ptr(running_var),
                               _ptr(bwd_part), bwd_part.shape[0], m, n, _ptr(gamma), float(eps), 1 if use_batch else 0, _ptr(coef),
                               _ptr(dgamma), _ptr(dbeta), _stream()))
    return coef, dgamma, dbeta


def bn_bwd_apply(dy: torch.Tensor, y: Optional[torch.Tensor], h: torch.Tensor, coef: torch.Tensor,
                 table: Optional[torch.Tensor] = None) -> torch.Tensor:
    dy = _rowmajor(_dev(dy, "dy", torch.float32), "dy")
    h = _rowmajor(_dev(h, "h", torch.float32), "h")
    if y is not None:
        y = _rowmajor(_dev(y, "y", torch.float32), "y")
    if table is not None:
        table = _dev(table, "table", torch.float32).contiguous()
    m, n = h.shape
    coef = _dev(coef, "coef", torch.float32).contiguous()
    dx = torch.empty((m, n), dtype=torch.float32, device=h.device)
    word = ctx().bounds.word() if ctx().bounds is not None else None     # max |dx|: the dgrad launches that read it (f16x2 form)
    check(lib.rgnn_bn_bwd_apply_table(_ptr(dy), _ld(dy), _ptr(y), 0 if y is None else _ld(y), _ptr(table), _ptr(h), _ld(h), _ptr(coef),
                                      m, n, _ptr(dx), n, _ptr(word), _stream()))
    set_bound(dx, word)
    return dx


def mpnn_aggregate_bwd(dM, Q, We, ea_sorted, rowptr_t, src_sorted, aggr: str, source_csr, node_order=None,
                       target_scale: Optional[torch.Tensor] = None, edge_maps=None, arg: Optional[torch.Tensor] = None):
    """Gradients of M[t] = aggr_{e->t}(Q[src_e] + We a_e) -> (dQ [n,d], d_edge_attr [E,de] or None, dWe [d,de] or None).
    ``source_csr`` = (rowptr_s, tnode, tpos): the same edges keyed on their source (see rgnn.h).  ``edge_maps`` =
    (tgt_sorted, eloc_sorted, tloc) -- target and in-segment index of every sorted edge, in-segment index of every out-edge
    (TargetCSR.edge_maps(); None when an in-degree exceeds 65 535) -- enables the lane-local kernels of rgnn_mpnn_max_bwd on
    the shapes they cover; ``arg``: the winners the forward pass recorded (uint16 in-segment indices; else recomputed)."""
    dM = _rowmajor(_dev(dM, "dM", torch.float32), "dM")
    _, Q, We, ea_sorted, de = _mp_common(None, None, Q, We, ea_sorted, rowptr_t, src_sorted)
    n, d = rowptr_t.numel() - 1, Q.shape[1]
    rowptr_s, tnode, tpos = source_csr
    for t_, nm in ((rowptr_s, "rowptr_s"), (tnode, "tnode"), (tpos, "tpos")):
        _dev(t_, nm, torch.int32)
    dev = Q.device
    n_edges = src_sorted.numel()
    if n_edges == 0:                                      # (a graph without edges: no source received anything from anybody)
        return (torch.zeros((n, d), dtype=torch.float32, device=dev),
                torch.zeros((0, de), dtype=torch.float32, device=dev) if de else None,
                torch.zeros((d, de), dtype=torch.float32, device=dev) if de else None)
    dQ = torch.empty((n, d), dtype=torch.float32, device=dev)
    if (AGGR_CODES[aggr] == 0 and edge_maps is not None and n_edges > 0 and USE_MAX_BWD and lib.rgnn_mpnn_max_bwd_supported(d, de)
            and dM.stride(0) % 4 == 0 and Q.stride(0) % 4 == 0):
        tgt_sorted, eloc_sorted, tloc = edge_maps
        for t_, nm in ((tgt_sorted, "tgt_sorted"), (eloc_sorted, "eloc_sorted"), (tloc, "tloc")):
            _dev(t_, nm, torch.int32)
        dea = torch.empty((n_edges, de), dtype=torch.float32, device=dev)
        dWe = torch.empty((d, de), dtype=torch.float32, device=dev)
        part = torch.empty((int(lib.rgnn_mpnn_bwd_slots(n)), d, de), dtype=torch.float32, device=dev)
        have_arg = arg is not None
        if have_arg:
            _dev(arg, "arg", torch.int16)
        else:
            arg = torch.empty((n, d), dtype=torch.int16, device=dev)
        word = ctx().bounds.word() if ctx().bounds is not None else None  # max |dQ|: the dx launch reads [dh | dQ] (f16x2 form)
        check(lib.rgnn_mpnn_max_bwd_absmax(_ptr(dM), _ld(dM), _ptr(Q), _ld(Q), _ptr(We), _ld(We), _ptr(ea_sorted), de, _ptr(rowptr_t),
                                           _ptr(src_sorted), _ptr(tgt_sorted), _ptr(eloc_sorted), _ptr(node_order), n, d, _ptr(rowptr_s),
                                           _ptr(tnode), _ptr(tloc), n_edges, _ptr(arg), 1 if have_arg else 0, _ptr(part), _ptr(dQ), d,
                                           _ptr(dea), _ptr(dWe), _ptr(word), _stream()))
        set_bound(dQ, word)
        return dQ, dea, dWe
    dea = torch.empty((n_edges, de), dtype=torch.float32, device=dev) if de else None
    cs = int(lib.rgnn_mpnn_bwd_split(d))
    dea_part = torch.empty((cs, n_edges, de), dtype=torch.float32, device=dev) if (de and cs > 1) else None
    dWe = torch.empty((d, de), dtype=torch.float32, device=dev) if de else None
    part = torch.empty((int(lib.rgnn_mpnn_bwd_slots(n)), d, de), dtype=torch.float32, device=dev) if de else None
    arg = torch.empty((n, d), dtype=torch.int32, device=dev) if AGGR_CODES[aggr] == 0 else None
    if AGGR_CODES[aggr] == 1:
        _dev(target_scale, "target_scale", torch.float32)
    check(lib.rgnn_mpnn_aggregate_bwd(_ptr(dM), _ld(dM), _ptr(Q), _ld(Q), _ptr(We), 0 if We is None else _ld(We),
                                      _ptr(ea_sorted), de, _ptr(rowptr_t), _ptr(src_sorted), _ptr(node_order), n, d,
                                      AGGR_CODES[aggr], _ptr(rowptr_s), _ptr(tnode), _ptr(tpos), _ptr(target_scale),
                                      n_edges, _ptr(arg), _ptr(part), _ptr(dea_part), _ptr(dQ), d, _ptr(dea), _ptr(dWe),
                                      _stream()))
    return dQ, dea, dWe


def linear_wgrad_supported(g: torch.Tensor, a1: torch.Tensor, a2: Optional[torch.Tensor]) -> bool:
    """(every row-major fp32 operand is supported since the bf16x3 kernel reads single floats)"""
    ts = [g, a1] + ([a2] if a2 is not None else [])
    return all(t.dim() == 2 and (t.shape[1] <= 1 or t.stride(1) == 1) for t in ts)


def linear_wgrad(g: torch.Tensor, a1: torch.Tensor, a2: Optional[torch.Tensor] = None, with_bias: bool = False,
                 row_index: Optional[torch.Tensor] = None, m_dev: Optional[torch.Tensor] = None, bounds=None) -> torch.Tensor:
    """dW [N, K1 + K2 (+ 1)] = g^T [a1 | a2 (| 1)] over all rows or the rows of a device-side row list (rgnn_wgrad: bf16x3
    MFMA products, fp32 accumulate).  ``with_bias``: the last column is the bias gradient (column sums of g).
    Inside a bound pool (``using_bounds``: the backward pass of a recorded forward) the launch takes the f16x2 form when every
    operand carries a bound -- ``bounds`` = (of g, of a1, of a2) where the caller kept them, else the operands' own."""
    bg, b1, b2 = bounds if bounds is not None else (bound_of(g), bound_of(a1), bound_of(a2))
    g = _rowmajor(_dev(g, "g", torch.float32), "g")
    a1 = _rowmajor(_dev(a1, "a1", torch.float32), "a1")
    n = g.shape[1]
    k1 = a1.shape[1]
    k2 = 0
    if a2 is not None:
        a2 = _rowmajor(_dev(a2, "a2", torch.float32), "a2")
        k2 = a2.shape[1]
    m = g.shape[0] if row_index is None else row_index.numel()
    if row_index is not None:
        _dev(row_index, "row_index", torch.int32)
    if m_dev is not None:
        _dev(m_dev, "m_dev", torch.int64)
    kt = k1 + k2 + (1 if with_bias else 0)
    if m == 0 or g.shape[0] == 0:                         # (no rows: the sum over nothing -- e.g. the edge MLPs of a graph without edges)
        return torch.zeros((n, kt), dtype=torch.float32, device=g.device)
    slabs = int(lib.rgnn_wgrad_slabs(m, n, k1, k2, 1 if with_bias else 0))
    part = torch.empty((slabs, n, kt), dtype=torch.float32, device=g.device)
    dw = torch.empty((n, kt), dtype=torch.float32, device=g.device)
    f16 = (ctx().bounds is not None and USE_F16X2 and WGRAD_F16X2 and bg is not None and (k1 == 0 or b1 is not None)
           and (k2 == 0 or b2 is not None))
    if f16 and CHECK_BOUNDS and not torch.cuda.is_current_stream_capturing():     # (debug net, as in ``linear``)
        for t_, b_, nm in ((g, bg, "g"), (a1, b1, "a1"), (a2, b2, "a2")):
            if t_ is not None and t_.numel():
                rows = t_ if row_index is None else t_[row_index[:int(m_dev.item()) if m_dev is not None else None].long()]
                if rows.numel() and float(rows.abs().max()) > float(b_.max()) * (1 + 1e-6):
                    raise RuntimeError(f"ops.linear_wgrad: operand {nm} exceeds the bound attached to it "
                                       f"({float(rows.abs().max())} > {float(b_.max())})")
    check(lib.rgnn_wgrad_bounds(_ptr(g), _ld(g), n, _ptr(a1), _ld(a1), k1, _ptr(a2), 0 if a2 is None else _ld(a2), k2,
                                1 if with_bias else 0, m, _ptr(row_index), _ptr(m_dev), _ptr(bg) if f16 else None,
                                _ptr(b1) if f16 else None, _ptr(b2) if f16 else None, _ptr(part), _ptr(dw), _stream()))
    COUNTERS["wgrad_f16x2" if f16 else "wgrad_other"] = COUNTERS.get("wgrad_f16x2" if f16 else "wgrad_other", 0) + 1
    return dw


def linear_wgrad_fp32(g: torch.Tensor, a1: torch.Tensor, a2: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The fp32-MFMA weight-gradient kernel (rgnn_linear_wgrad; widths and strides multiples of 4 floats) -- kept as the
    reference the bf16x3 kernel is measured against."""
    g = _rowmajor(_dev(g, "g", torch.float32), "g")
    a1 = _rowmajor(_dev(a1, "a1", torch.float32), "a1")
    m, n = g.shape
    k1 = a1.shape[1]
    k2 = 0
    if a2 is not None:
        a2 = _rowmajor(_dev(a2, "a2", torch.float32), "a2")
        k2 = a2.shape[1]
    k = k1 + k2
    slabs = int(lib.rgnn_linear_wgrad_slabs(m, n, k))
    part = torch.empty((slabs, n, k), dtype=torch.float32, device=g.device)
    dw = torch.empty((n, k), dtype=torch.float32, device=g.device)
    check(lib.rgnn_linear_wgrad(_ptr(g), _ld(g), _ptr(a1), _ld(a1), k1, _ptr(a2), 0 if a2 is None else _ld(a2), k2, m, n,
                                _ptr(part), _ptr(dw), _stream()))
    return dw


def segment_reduce_bwd(dM: torch.Tensor, rows: torch.Tensor, rowptr_t: torch.Tensor, aggr: str,
                       node_order: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Gradient of ``segment_reduce`` w.r.t. its [E, d] input rows."""
    dM = _rowmajor(_dev(dM, "dM", torch.float32), "dM")
    rows = _rowmajor(_dev(rows, "rows", torch.float32), "rows")
    n, d = rowptr_t.numel() - 1, rows.shape[1]
    out = torch.empty((rows.shape[0], d), dtype=torch.float32, device=rows.device)
    if rows.shape[0] == 0:
        return out
    check(lib.rgnn_segment_reduce_bwd(_ptr(dM), _ld(dM), _ptr(rows), _ld(rows), _ptr(rowptr_t), _ptr(node_order), n, d,
                                      AGGR_CODES[aggr], _ptr(out), d, _stream()))
    return out


# ---- batches of graphs resident in HBM (csrc/collate.hip) -----------------------------------------------------------
def _words_per_row(t: torch.Tensor) -> int:
    row_bytes = math.prod(t.shape[1:]) * t.element_size()
    if row_bytes % 4:
        raise NotImplementedError(f"collation moves rows as 4-byte words; dtype {t.dtype} with row size {row_bytes} B "
                                  "is not supported")
    return row_bytes // 4


def collate_rows(src: torch.Tensor, seg_src_row: torch.Tensor, seg_dst_ptr: torch.Tensor, n_rows: int,
                 want_batch: bool = False):
    """Segmented row copy out[dst_ptr[s] + r] = src[src_row[s] + r] (rgnn_collate_rows).  ``src``: resident [R, ...]
    tensor of a 4- or 8-byte dtype, contiguous.  -> out [n_rows, ...] (and PyG's ``batch`` vector int64 [n_rows])."""
    if not src.is_cuda:
        raise RuntimeError("collate_rows: the resident tensor must live on the GPU (no CPU fallback)")
    if not src.is_contiguous():
        raise ValueError("collate_rows: resident tensor must be contiguous")
    _dev(seg_src_row, "seg_src_row", torch.int64); _dev(seg_dst_ptr, "seg_dst_ptr", torch.int64)
    n_seg = seg_src_row.numel()
    if seg_dst_ptr.numel() != n_seg + 1:
        raise ValueError("seg_dst_ptr must have one more entry than seg_src_row")
    width = _words_per_row(src)
    out = torch.empty((n_rows,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    batch = torch.empty(n_rows, dtype=torch.int64, device=src.device) if want_batch else None
    if width == 0:
        if want_batch and n_rows:
            raise ValueError("cannot derive the batch vector from a zero-width tensor")
        return (out, batch) if want_batch else out
    check(lib.rgnn_collate_rows(_ptr(src), width, width, _ptr(seg_src_row), _ptr(seg_dst_ptr), n_seg, n_rows, _ptr(out),
                                width, _ptr(batch), _stream()))
    return (out, batch) if want_batch else out


def collate_edges(src_edge_index: torch.Tensor, seg_src_edge: torch.Tensor, seg_dst_eptr: torch.Tensor,
                  seg_node_shift: torch.Tensor, n_edges: int) -> torch.Tensor:
    """edge_index of the batch: both rows copied per selected graph, shifted by the graph's first batch row
    (rgnn_collate_edges).  ``src_edge_index`` int64 [2, E_all] contiguous, graph-local numbering."""
    _dev(src_edge_index, "src_edge_index", torch.int64)
    if src_edge_index.dim() != 2 or src_edge_index.shape[0] != 2 or not src_edge_index.is_contiguous():
        raise ValueError("src_edge_index must be a contiguous int64 [2, E] tensor")
    for t, nm in ((seg_src_edge, "seg_src_edge"), (seg_dst_eptr, "seg_dst_eptr"), (seg_node_shift, "seg_node_shift")):
        _dev(t, nm, torch.int64)
    out = torch.empty((2, n_edges), dtype=torch.int64, device=src_edge_index.device)
    check(lib.rgnn_collate_edges(_ptr(src_edge_index), src_edge_index.shape[1], _ptr(seg_src_edge), _ptr(seg_dst_eptr),
                                 _ptr(seg_node_shift), seg_src_edge.numel(), n_edges, _ptr(out), n_edges, _stream()))
    return out


# ---- post-processor front half (csrc/postprocess.hip) ---------------------------------------------------------------
def decode_predictions(prob: torch.Tensor, boxes: torch.Tensor, pos: torch.Tensor, nn_index: Optional[torch.Tensor],
                       bg_index: int, max_score_for_background: float, min_object_score: Sequence[float], invariance: int,
                       adapt_orientation_angle: bool):
    """-> (label int32 [N], score f32 [N], keep int32 [N], corners f64 [N, 4, 2]) (rgnn_decode_predictions)."""
    prob = _rowmajor(_dev(prob, "prob", torch.float32), "prob")
    boxes = _rowmajor(_dev(boxes, "boxes", torch.float32), "boxes")
    _dev(pos, "pos", torch.float32)
    if not pos.is_contiguous():
        raise ValueError("pos must be contiguous [N, 2]")
    n, k = prob.shape
    dev = prob.device
    if nn_index is not None:
        _dev(nn_index, "nn_index", torch.int32)
    ms = torch.tensor(list(min_object_score), dtype=torch.float64, device=dev) if len(min_object_score) else None
    label = torch.empty(n, dtype=torch.int32, device=dev)
    score = torch.empty(n, dtype=torch.float32, device=dev)
    keep = torch.empty(n, dtype=torch.int32, device=dev)
    corners = torch.empty((n, 4, 2), dtype=torch.float64, device=dev)
    check(lib.rgnn_decode_predictions(_ptr(prob), _ld(prob) if n > 1 else k, k, _ptr(boxes), _ld(boxes) if n > 1 else boxes.shape[1],
                                      boxes.shape[1], _ptr(pos), _ptr(nn_index), n, int(bg_index),
                                      float(max_score_for_background), _ptr(ms), 0 if ms is None else ms.numel(),
                                      int(invariance), 1 if adapt_orientation_angle else 0, _ptr(label), _ptr(score),
                                      _ptr(keep), _ptr(corners), _stream()))
    return label, score, keep, corners


def row_argmax(prob: torch.Tensor):
    """(first index of the row maximum int32 [N], the maximum f32 [N]) -- the label / score part of the decode kernel."""
    n, k = prob.shape
    dummy_box = torch.zeros((n, 4), dtype=torch.float32, device=prob.device)
    dummy_pos = torch.zeros((n, 2), dtype=torch.float32, device=prob.device)
    label, score, _, _ = decode_predictions(prob, dummy_box, dummy_pos, None, -1, 2.0, [], 1, False)
    return label, score


def box_representations(corners: torch.Tensor, two_point: bool = True, rotated: bool = True):
    """corners f64 [M, 4, 2] -> (two_point f64 [M, 4] | None, rotated f64 [M, 5] | None) (rgnn_box_representations)."""
    _dev(corners, "corners", torch.float64)
    corners = corners.contiguous()
    m = corners.shape[0]
    tp = torch.empty((m, 4), dtype=torch.float64, device=corners.device) if two_point else None
    rot = torch.empty((m, 5), dtype=torch.float64, device=corners.device) if rotated else None
    check(lib.rgnn_box_representations(_ptr(corners), m, _ptr(tp), _ptr(rot), _stream()))
    return tp, rot


def sort_scores(scores: torch.Tensor) -> torch.Tensor:
    """int64 [M]: ids by descending score, ties by ascending id, NaN first (rgnn_sort_scores) -- what
    ``torch.sort(scores, descending=True, stable=True).indices`` returns."""
    sc = scores.reshape(-1)
    if not sc.is_cuda:
        raise RuntimeError("scores must be a CUDA tensor")
    if sc.dtype not in (torch.float32, torch.float64):
        sc = sc.to(torch.float32)
    sc = sc.contiguous()
    m = sc.numel()
    order = torch.empty(m, dtype=torch.int64, device=sc.device)
    tmp = torch.empty(int(lib.rgnn_sort_scores_tmp_bytes(m)), dtype=torch.uint8, device=sc.device)
    check(lib.rgnn_sort_scores(_ptr(sc), 1 if sc.dtype == torch.float64 else 0, m, _ptr(order), _ptr(tmp), _stream()))
    return order


def nms(boxes: torch.Tensor, scores: torch.Tensor, iou_threshold: float, rotated: bool) -> torch.Tensor:
    """Greedy NMS on the device -> int64 ids kept, by descending score.  ``boxes``: f32 [M, 4] two-point (aligned,
    torchvision semantics) or f64 [M, 5] [x, y, l, w, theta deg] (rotated, detectron2 semantics).  One host read (count)."""
    boxes = _dev(boxes, "boxes", torch.float64 if rotated else torch.float32).contiguous()
    m = boxes.shape[0]
    if boxes.dim() != 2 or boxes.shape[1] != (5 if rotated else 4) or scores.shape[0] != m:
        raise ValueError("boxes [M, 4] float32 (aligned) or [M, 5] float64 (rotated); one score per box")
    if not scores.is_cuda:
        raise RuntimeError("scores must be a CUDA tensor")
    order = sort_scores(scores)
    mask = torch.empty(max(int(lib.rgnn_nms_mask_words(m)), 1), dtype=torch.int64, device=boxes.device)
    keep = torch.empty(max(m, 1), dtype=torch.int64, device=boxes.device)
    count = torch.empty(1, dtype=torch.int64, device=boxes.device)
    check(lib.rgnn_nms(_ptr(boxes), 1 if rotated else 0, _ptr(order), m, float(iou_threshold), _ptr(mask), _ptr(keep),
                       _ptr(count), _stream()))
    return keep[:int(count.item())]


# ---- detection loss (csrc/loss.hip) ----------------------------------------------------------------------------------
def detection_loss(cls: torch.Tensor, boxes: torch.Tensor, y: torch.Tensor, class_weight: Optional[torch.Tensor],
                   bg_index: int, delta: float, cls_loss_weight: float, bb_loss_weight: float):
    """-> (out f32 [3] = loss, loss_cls, loss_bb; sums f64 [4] for the backward pass) (rgnn_detection_loss)."""
    cls = _rowmajor(_dev(cls, "cls", torch.float32), "cls")
    boxes = _rowmajor(_dev(boxes, "boxes", torch.float32), "boxes")
    y = _rowmajor(_dev(y, "y", torch.float32), "y")
    n, k = cls.shape
    w = boxes.shape[1]
    if boxes.shape[0] != n or y.shape != (n, 1 + w):
        raise ValueError("shapes: cls [N, K], boxes [N, W], y [N, 1 + W]")
    if class_weight is not None:
        _dev(class_weight, "class_weight", torch.float32)
        if class_weight.numel() != k or not class_weight.is_contiguous():
            raise ValueError("class_weight must be a contiguous [K] tensor")
    nb = int(lib.rgnn_detection_loss_blocks(n))
    partial = torch.empty(4 * nb, dtype=torch.float64, device=cls.device)
    sums = torch.empty(4, dtype=torch.float64, device=cls.device)
    out = torch.empty(3, dtype=torch.float32, device=cls.device)
    check(lib.rgnn_detection_loss(_ptr(cls), _ld(cls) if n > 1 else k, k, _ptr(boxes), _ld(boxes) if n > 1 else w, w, _ptr(y),
                                  _ld(y) if n > 1 else 1 + w, _ptr(class_weight), n, int(bg_index), float(delta),
                                  float(cls_loss_weight), float(bb_loss_weight), _ptr(partial), _ptr(sums), _ptr(out), _stream()))
    return out, sums


def detection_loss_bwd(cls, boxes, y, class_weight, bg_index, delta, cls_loss_weight, bb_loss_weight, sums, grad_loss):
    cls = _rowmajor(cls, "cls"); boxes = _rowmajor(boxes, "boxes"); y = _rowmajor(y, "y")
    n, k = cls.shape
    w = boxes.shape[1]
    d_cls = torch.empty((n, k), dtype=torch.float32, device=cls.device)
    d_bb = torch.empty((n, w), dtype=torch.float32, device=cls.device)
    g = None if grad_loss is None else grad_loss.reshape(1).to(torch.float32).contiguous()
    check(lib.rgnn_detection_loss_bwd(_ptr(cls), _ld(cls) if n > 1 else k, k, _ptr(boxes), _ld(boxes) if n > 1 else w, w,
                                      _ptr(y), _ld(y) if n > 1 else 1 + w, _ptr(class_weight), n, int(bg_index), float(delta),
                                      float(cls_loss_weight), float(bb_loss_weight), _ptr(sums), _ptr(g), _ptr(d_cls), k,
                                      _ptr(d_bb), w, _stream()))
    return d_cls, d_bb
