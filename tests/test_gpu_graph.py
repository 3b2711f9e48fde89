"""HIP graph construction / feature kernels (through the C ABI) against the numpy oracle and the golden
vectors generated from the reference.  Bars: topology bit-exact; float64 features 1e-12 relative; float32
features identical to the oracle's cast up to 1 ulp."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, golden_files
from oracle import graph_oracle as go
from radargnn_amd import synthetic

pytestmark = pytest.mark.gpu

ALL_EDGE = ["point_pair_features", "spatial_euclidean_distance", "velocity_euclidean_distance",
            "relative_position", "relative_velocity"]
ALL_NODE = ["rcs", "time_index", "degree", "velocity_vector_length", "velocity_vector", "spatial_coordinates"]


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test but no GPU visible")
    from radargnn_amd import ops as _ops
    return _ops


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def batch(frames):
    cat, ptr = synthetic.concat_frames(frames)
    return cat, ptr


def oracle_batch_edges(frames, routine, k=None, r=None, basis="X"):
    out, off = [], 0
    for f in frames:
        X = f.X if basis == "X" else np.concatenate((f.X, f.V), axis=1)
        E = go.build_edges(X, routine, k=k, r=r)
        if E is not None and E.shape[0]:
            out.append(E.astype(np.int64) + off)
        off += f.n
    return np.concatenate(out) if out else np.zeros((0, 2), np.int64)


def test_scan(ops):
    for n in (0, 1, 5, 2048, 2049, 1_000_003, 4_194_304, 4_200_001):        # (> 2048 tiles: the three-launch chain)
        x = torch.randint(0, 7, (n,), dtype=torch.int32, device="cuda")
        got = ops.exclusive_scan_i32(x).cpu().numpy()
        exp = np.concatenate([[0], np.cumsum(x.cpu().numpy(), dtype=np.int64)]).astype(np.int32)
        assert np.array_equal(got, exp)


@pytest.mark.parametrize("name", golden_files("rs3000_") + golden_files("small_n300") + golden_files("small_n40"))
def test_topology_vs_reference_golden(ops, name):
    d = np.load(os.path.join(GOLDEN, name))
    routine, k, r, mode, basis = [str(s) for s in d["meta"]]
    k, r = int(k), float(r)
    X = d["X"] if basis == "X" else np.concatenate((d["X"], d["V"]), axis=1)
    n = X.shape[0]
    ptr = dev(np.array([0, n], dtype=np.int64))
    if routine == "radius":
        rowptr, col, ei = ops.radius_graph(dev(X), ptr, r)
        E = ei.t().cpu().numpy()
        assert np.array_equal(E, go.canonical_edges(d["E"]))            # rows ascending, cols ascending
        deg = ops.undirected_degree(rowptr, col, n)
    else:
        nbr, ei, status = ops.knn_graph(dev(X), ptr, k)
        assert status.item() == 0
        E = ei.t().cpu().numpy()
        assert np.array_equal(go.canonical_edges(E), go.canonical_edges(d["E"]))
        assert np.array_equal(E, go.knn_edges(X, k))                    # (distance asc, index asc) like the oracle
        rowptr = torch.arange(0, n * k + 1, k, dtype=torch.int32, device="cuda")
        deg = ops.undirected_degree(rowptr, nbr.reshape(-1), n)
    assert np.array_equal(deg.cpu().numpy(), d["degree"])
    ti, st = ops.time_index(dev(d["timestamp"]), ptr)
    assert st.item() == 0
    assert np.array_equal(ti.cpu().numpy(), d["time_index"])


@pytest.mark.parametrize("name", golden_files("small_"))
def test_features_vs_reference_golden(ops, name):
    d = np.load(os.path.join(GOLDEN, name))
    routine, k, r, mode, basis = [str(s) for s in d["meta"]]
    X, V = d["X"], d["V"]
    n = X.shape[0]
    ei = dev(d["E"].T.astype(np.int64))
    ef, st = ops.edge_features(dev(X), dev(V), ei, ALL_EDGE, mode, dtype=torch.float64)
    assert st.item() == 0
    np.testing.assert_allclose(ef.cpu().numpy(), d["E_feat"], rtol=1e-12, atol=1e-9, equal_nan=True)
    ptr = dev(np.array([0, n], dtype=np.int64))
    ti, _ = ops.time_index(dev(d["timestamp"]), ptr)
    xf = ops.node_features(dev(X), dev(V), dev(d["rcs"]), ti, dev(d["degree"]), ALL_NODE, dtype=torch.float64)
    np.testing.assert_allclose(xf.cpu().numpy(), d["X_feat"], rtol=1e-14, atol=0)


def test_float32_handover_matches_reference(ops):
    """create_graph_data dtypes (dataset_creation.py:804-806) on the 3000-point frame."""
    for name, efeat, nfeat in [("rs3000_knn_k20_r1.npz", ["relative_position"], ["rcs", "velocity_vector", "time_index", "degree"]),
                               ("rs3000_radius_ppf.npz", ["point_pair_features"], ["rcs", "velocity_vector_length", "time_index", "degree"])]:
        d = np.load(os.path.join(GOLDEN, name))
        X, V = d["X"], d["V"]
        ei = dev(d["E"].T.astype(np.int64))
        ef, st = ops.edge_features(dev(X), dev(V), ei, efeat, "directed", dtype=torch.float32)
        assert st.item() == 0
        np.testing.assert_allclose(ef.cpu().numpy(), d["E_feat"], rtol=2e-7, atol=1e-5)
        ti, _ = ops.time_index(dev(d["timestamp"]), dev(np.array([0, X.shape[0]], dtype=np.int64)))
        xf = ops.node_features(dev(X), dev(V), dev(d["rcs"]), ti, dev(d["degree"]), nfeat, dtype=torch.float32)
        assert np.array_equal(xf.cpu().numpy(), d["X_feat"])


@pytest.mark.parametrize("routine,k,r,basis", [("radius", None, 1.0, "X"), ("radius", None, 2.5, "XV"),
                                               ("knn", 1, None, "X"), ("knn", 10, None, "X"), ("knn", 20, None, "X"),
                                               ("knn", 5, None, "XV")])
def test_batched_ragged_frames(ops, routine, k, r, basis):
    """Several frames of different sizes in one batch (incl. a 1-point and an empty frame for radius)."""
    frames = [synthetic.radarscenes_frame(1), synthetic.nuscenes_frame(2), synthetic.small_frame(40, 5, duplicates=2),
              synthetic.radarscenes_frame(3, n_clusters=10, pts_per_cluster=20, n_clutter=100)]
    if routine == "radius":
        frames.insert(1, synthetic.small_frame(1, 9))
        frames.insert(3, synthetic.RadarFrame(np.zeros((0, 2)), np.zeros((0, 2)), np.zeros((0, 1)), np.zeros((0, 1))))
    cat, ptr = batch(frames)
    X = cat.X if basis == "X" else np.concatenate((cat.X, cat.V), axis=1)
    exp = oracle_batch_edges(frames, routine, k=k, r=r, basis=basis)
    if routine == "radius":
        rowptr, col, ei = ops.radius_graph(dev(X), dev(ptr), r)
        assert np.array_equal(ei.t().cpu().numpy(), exp)
        assert rowptr[-1].item() == exp.shape[0]
        deg = ops.undirected_degree(rowptr, col, X.shape[0]).cpu().numpy()
    else:
        nbr, ei, status = ops.knn_graph(dev(X), dev(ptr), k)
        assert status.item() == 0
        assert np.array_equal(ei.t().cpu().numpy(), exp)
        rowptr = torch.arange(0, X.shape[0] * k + 1, k, dtype=torch.int32, device="cuda")
        deg = ops.undirected_degree(rowptr, nbr.reshape(-1), X.shape[0]).cpu().numpy()
    off, exp_deg = 0, []
    for f in frames:
        Ef = exp[(exp[:, 0] >= off) & (exp[:, 0] < off + f.n)] - off
        exp_deg.append(go.undirected_degree(Ef, f.n))
        off += f.n
    assert np.array_equal(deg, np.concatenate(exp_deg))
    # time index per frame
    ti, st = ops.time_index(dev(cat.timestamp), dev(ptr))
    assert st.item() == 0
    exp_ti = np.concatenate([go.time_index(f.timestamp).reshape(-1) if f.n else np.zeros(0) for f in frames])
    assert np.array_equal(ti.cpu().numpy(), exp_ti)


@pytest.mark.parametrize("k", [3, 33, 64, 65, 90])
def test_knn_team_kernel_equals_the_one_thread_kernel_and_the_oracle(ops, k, monkeypatch):
    """k_knn_team (64 lanes per query, rank-counting prune; k <= 64) against k_knn (one thread per query; also what k > 64
    takes) and the KD-tree-faithful oracle: ragged frames with duplicates, clusters denser than the candidate buffer, sparse
    clutter whose rings run to the frame border."""
    frames = [synthetic.radarscenes_frame(4, n_clusters=6, pts_per_cluster=150, n_clutter=120), synthetic.nuscenes_frame(3),
              synthetic.small_frame(k + 2, 7, duplicates=3)]
    cat, ptr = batch(frames)
    exp = oracle_batch_edges(frames, "knn", k=k, r=None, basis="X")
    monkeypatch.setenv("RGNN_KNN_TEAM", "0")
    __import__("radargnn_amd.ops").ops.reload_env()
    nbr0, ei0, st0 = ops.knn_graph(dev(cat.X), dev(ptr), k)
    monkeypatch.delenv("RGNN_KNN_TEAM")
    nbr1, ei1, st1 = ops.knn_graph(dev(cat.X), dev(ptr), k)
    assert st0.item() == 0 and st1.item() == 0
    assert torch.equal(nbr0, nbr1) and torch.equal(ei0, ei1)
    assert np.array_equal(ei1.t().cpu().numpy(), exp)


def test_time_index_of_large_frames_spread_over_the_chip(ops):
    """rgnn_time_index_ws (frames beyond 16 384 points: hash set in global memory, one sort per frame, one look-up per point) gives
    the indices of the one-block-per-frame kernel and of the oracle (rank among np.unique of the frame): ragged batch with a
    20 000-point frame, an empty frame, -0.0 / +0.0 and ~3 000 distinct values in one frame; too many distinct values are flagged."""
    rng = np.random.default_rng(3)
    sizes = [20000, 0, 37, 5000]
    ts = [rng.integers(0, 40, size=sizes[0]).astype(np.float64) * 17.0 + 1e6, np.zeros(0),
          np.array([0.0, -0.0, 1.0] * 12 + [2.0]), rng.integers(0, 3000, size=sizes[3]).astype(np.float64) * 0.5 - 700.0]
    ptr = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    cat = np.concatenate(ts)
    small, st0 = ops.time_index(dev(cat), dev(ptr))
    big, st1 = ops.time_index(dev(cat), dev(ptr), max_frame_points=20000)
    assert st0.item() == 0 and st1.item() == 0
    exp = np.concatenate([go.time_index(t.reshape(-1, 1)).reshape(-1) if t.size else np.zeros(0) for t in ts])
    assert np.array_equal(small.cpu().numpy(), exp) and np.array_equal(big.cpu().numpy(), exp)
    many = np.arange(40000, dtype=np.float64)
    _, st = ops.time_index(dev(many), dev(np.array([0, 40000], dtype=np.int64)), max_frame_points=40000)
    assert st.item() & ops.STATUS_TIME_INDEX_OVERFLOW


@pytest.mark.parametrize("k", [3, 10, 20, 32])
@pytest.mark.parametrize("sizes", [(300, 300, 41, 300), (500, 480), (1000, 700, 33)])
def test_knn_frame_kernel_equals_the_team_kernel_and_the_oracle(ops, k, sizes, monkeypatch):
    """k_knn_frame (r05: brute force per small frame, one wave per query, k-th smallest by a binary search over the distance bits)
    against k_knn_team on the same batch and against the KD-tree-faithful oracle: ragged frames in all three register layouts
    (<= 320, <= 512, <= 1 024 points), EXACT DUPLICATES (zero distances, equal keys broken by index), a lattice of points with
    many equal distances at the k-th place, distance basis X and XV, with the relative_position / out-degree outputs."""
    rng = np.random.default_rng(11)
    frames = []
    for j, n in enumerate(sizes):
        if j == 1:                                     # a lattice: ties at the k-th distance everywhere
            side = int(np.ceil(np.sqrt(n)))
            gx, gy = np.meshgrid(np.arange(side, dtype=np.float64), np.arange(side, dtype=np.float64))
            Xl = np.stack([gx.ravel(), gy.ravel()], 1)[:n] * 0.5
            frames.append(synthetic.RadarFrame(Xl, rng.normal(size=(n, 2)).round(1), rng.normal(size=(n, 1)), np.zeros((n, 1))))
        else:
            f = synthetic.nuscenes_frame(20 + j) if n == 300 else synthetic.small_frame(n, 3 + j, duplicates=min(9, n // 4))
            if f.n != n:
                f = synthetic.small_frame(n, 5 + j, duplicates=min(9, n // 4))
            frames.append(f)
    if min(f.n for f in frames) <= k:
        pytest.skip("a frame with <= k points: covered by test_knn_too_few_points_sets_status")
    cat, ptr = batch(frames)
    biggest = max(f.n for f in frames)
    for basis_name in ("X", "XV"):
        basis = cat.X if basis_name == "X" else np.concatenate([cat.X, cat.V], 1)
        exp = oracle_batch_edges(frames, "knn", k=k, r=None, basis=basis_name)
        nbr_t, ei_t, st_t = ops.knn_graph(dev(basis), dev(ptr), k)                                   # no frame size: the grid walk
        nbr_f, ei_f, st_f, rel, deg = ops.knn_graph(dev(basis), dev(ptr), k, max_frame_points=biggest,
                                                    relative_position="directed", degree_init=True)   # small frames: brute force
        assert st_t.item() == 0 and st_f.item() == 0
        assert torch.equal(nbr_t, nbr_f) and torch.equal(ei_t, ei_f), basis_name
        # rows with a tie at the k-th place: the oracle (stable argsort) and the kernels both break ties by index
        assert np.array_equal(ei_f.t().cpu().numpy(), exp), basis_name
        assert (deg == k).all()
        d = basis[ei_f[0].cpu().numpy()][:, :2] - basis[ei_f[1].cpu().numpy()][:, :2]
        assert np.array_equal(rel.cpu().numpy(), d.astype(np.float32))


def test_knn_too_few_points_sets_status(ops):
    f = synthetic.small_frame(5, 1)
    nbr, ei, status = ops.knn_graph(dev(f.X), dev(np.array([0, 5], dtype=np.int64)), 5)
    assert status.item() & ops.STATUS_KNN_TOO_FEW_POINTS


def test_csr_by_target(ops):
    rng = np.random.default_rng(0)
    n, e = 1000, 20000
    ei = np.stack([rng.integers(0, n, e), rng.integers(0, n, e)]).astype(np.int64)
    ei[1, :50] = 7                                       # a hub and duplicates
    rowptr, src, perm = ops.csr_by_target(dev(ei), n)
    order = np.argsort(ei[1], kind="stable")             # stable counting sort on the target
    assert np.array_equal(perm.cpu().numpy(), order)
    assert np.array_equal(src.cpu().numpy(), ei[0][order])
    assert np.array_equal(rowptr.cpu().numpy(), np.concatenate([[0], np.cumsum(np.bincount(ei[1], minlength=n))]))


@pytest.mark.parametrize("with_rank", [False, True])
def test_csr_by_target_of_a_symmetric_graph_from_the_rows_of_the_search(ops, with_rank):
    """rgnn_csr_by_target_symmetric: a radius graph's edge list is grouped by its query and symmetric, so the CSR by target
    follows from the search's rowptr without histogram / atomics / sort -- the same three arrays as the general builder, with
    and without a visiting order; an edge whose twin is missing is reported."""
    from radargnn_amd import frames as F
    fr = [synthetic.radarscenes_frame(i) for i in range(6)]
    b = F.FrameBatch.from_frames(fr)
    cfg = F.GraphSettings(algorithm="radius", r=1.0, node_features=("rcs", "degree"), edge_features=("relative_position",))
    g = F.build_graphs(b, cfg)
    n = g.x.shape[0]
    assert g.rowptr is not None and g.rowptr.numel() == n + 1
    rank = ops.invert_permutation(g.cell_order) if with_rank else None
    ref = ops.csr_by_target(g.edge_index, n, rank)
    status = torch.zeros(1, dtype=torch.int32).cuda()
    got = ops.csr_by_target(g.edge_index, n, rank, symmetric_rows=g.rowptr, status=status)
    for a_, b_ in zip(ref, got):
        assert torch.equal(a_, b_)
    assert int(status.item()) == 0
    # break the symmetry: redirect one edge to a node that is no neighbour of its source
    ei = g.edge_index.clone()
    far = int((ei[1] != ei[1, 0]).nonzero()[-1])
    ei[1, 0] = ei[0, far]
    ops.csr_by_target(ei, n, rank, symmetric_rows=g.rowptr, status=status)
    assert int(status.item()) & ops.STATUS_NOT_SYMMETRIC


def test_split_targets_in_node_order_holds_the_same_nodes_as_in_visiting_order(ops):
    """rgnn_split_targets_by_node: the nodes with / without incoming edges as ascending lists (what the row-subset dense
    launches stream fastest) -- the same sets as the visiting-order lists, counts on the device, slot[] consistent."""
    rng = np.random.default_rng(3)
    n, e = 5000, 9000
    ei = np.stack([rng.integers(0, n, e), rng.integers(0, n // 2, e)]).astype(np.int64)      # half of the nodes get no in-edge
    order = torch.from_numpy(rng.permutation(n).astype(np.int32)).cuda()
    rank = ops.invert_permutation(order)
    rowptr, _, _ = ops.csr_by_target(dev(ei), n, rank)
    a = ops.split_targets(rowptr, order)
    b = ops.split_targets(rowptr, order, rank=rank, by_node=True)
    for (la, ca), (lb, cb) in (((a[0], a[1]), (b[0], b[1])), ((a[3], a[4]), (b[3], b[4]))):
        m = int(ca.item())
        assert m == int(cb.item())
        got = lb[:m].cpu().numpy()
        assert np.array_equal(got, np.sort(la[:m].cpu().numpy())) and np.all(np.diff(got) > 0)
    has_in = np.bincount(ei[1], minlength=n) > 0
    assert np.array_equal(np.sort(b[3][:int(b[4].item())].cpu().numpy()), np.nonzero(has_in)[0])
    slot = b[2].cpu().numpy()
    lst = b[0][:int(b[1].item())].cpu().numpy()
    assert np.array_equal(slot[lst], np.arange(len(lst))) and np.all(slot[has_in] == -1)
    # no visiting order at all
    rowptr0, _, _ = ops.csr_by_target(dev(ei), n)
    c = ops.split_targets(rowptr0, None, by_node=True)
    assert np.array_equal(c[3][:int(c[4].item())].cpu().numpy(), np.nonzero(has_in)[0])


def test_new_graph_entry_points_on_empty_and_edge_free_inputs(ops):
    """Empty and edge-free inputs through the entry points added for the radius path: a graph without edges (every row of the
    search empty), no nodes at all, and BatchNorm statistics with one part that holds no rows."""
    n = 7
    ei = torch.zeros((2, 0), dtype=torch.int64).cuda()
    rows = torch.zeros(n + 1, dtype=torch.int32).cuda()
    status = torch.zeros(1, dtype=torch.int32).cuda()
    rowptr, src, perm = ops.csr_by_target(ei, n, None, symmetric_rows=rows, status=status)
    assert rowptr.cpu().tolist() == [0] * (n + 1) and src.numel() == 0 and perm.numel() == 0 and int(status.item()) == 0
    lst, cnt, slot, lst_ne, cnt_ne = ops.split_targets(rowptr, None, by_node=True)
    assert int(cnt.item()) == n and int(cnt_ne.item()) == 0 and lst[:n].cpu().tolist() == list(range(n))
    assert slot.cpu().tolist() == list(range(n))
    rowptr0, _, _ = ops.csr_by_target(ei, 0, None, symmetric_rows=torch.zeros(1, dtype=torch.int32).cuda(), status=status)
    assert rowptr0.cpu().tolist() == [0]
    e0 = ops.split_targets(rowptr0, None, by_node=True)
    assert int(e0[1].item()) == 0 and int(e0[4].item()) == 0
    # statistics: part A holds all rows, part B none (its buffer is never read: NaN)
    c, m = 16, 300
    x = torch.randn(m, c).cuda()
    st_a = ops.column_stats(x)
    st_b = torch.full_like(st_a, float("nan"))
    parts = ops.StatParts([(st_a, torch.tensor([m]).cuda()), (st_b, torch.tensor([0]).cuda())])
    ss = ops.batchnorm_finalize(parts, m, c, None, None, None, None, None, True, 0.1, 1e-5)
    ref = ops.batchnorm_finalize(st_a, m, c, None, None, None, None, None, True, 0.1, 1e-5)
    assert torch.equal(ss, ref) and bool(torch.isfinite(ss).all())


def test_dot_product_error_flag(ops):
    """features.py:49-56,70-77,84-91 raise "Error in dot product calculation" when a dot product of two normalised vectors
    leaves [-1 - 1e-3, 1 + 1e-3].  The kernel normalises like the oracle, v / sqrt(vx^2 + vy^2); that can only fail when the
    sum of squares underflows: |v| ~ 1e-200 -> norm 0 -> "unit" vector (inf, inf) -> dot = inf.  The device then sets
    RGNN_STATUS_DOT_PRODUCT, GraphBatch.check() raises the reference's exception, and the oracle raises on the same input.
    (The reference itself normalises [2, 1]-shaped arrays with np.linalg.norm(ord=2), the SVD-based matrix norm, which does
    not underflow: for velocities below ~1e-154 m/s it still returns angles.  Physically meaningless input; noted in DESIGN.)
    A valid input leaves the status word clear, and an unknown feature name raises like graph.py:219-220."""
    from oracle import graph_oracle as go
    from radargnn_amd import frames as fr
    f = synthetic.small_frame(6, 0)
    ei = dev(np.array([[0, 1], [1, 0]], dtype=np.int64))
    _, status = ops.edge_features(dev(f.X), dev(f.V), ei, ["point_pair_features"], "directed")
    assert int(status.item()) == 0
    V = f.V.copy()
    V[0] = [1e-200, 1e-200]
    V[1] = [1e-200, 1e-200]
    _, status = ops.edge_features(dev(f.X), dev(V), ei, ["point_pair_features"], "directed")
    assert int(status.item()) & ops.STATUS_DOT_PRODUCT
    g = fr.GraphBatch(None, None, None, None, status, 1)
    with pytest.raises(Exception, match="Error in dot product calculation"):
        g.check()
    with np.errstate(all="ignore"), pytest.raises(go.DotProductError):
        go.edge_features(f.X, V, np.array([[0, 1], [1, 0]]), ["point_pair_features"], "directed")
    with pytest.raises(Exception, match="Invalid feature specified"):
        ops.edge_features(dev(f.X), dev(f.V), ei, ["bogus"], "directed")


def test_cpu_tensors_are_rejected(ops):
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.radius_graph(torch.zeros(4, 2, dtype=torch.float64), torch.tensor([0, 4]), 1.0)


def test_stress_cloud_radius_properties(ops):
    """100 000-point cloud (config 5): symmetry of the radius graph, inclusive bound, sortedness -- properties that
    do not need the O(N^2) oracle -- plus an exact oracle check on a 2 000-query sample."""
    c = synthetic.stress_cloud()
    X = c.X
    n = X.shape[0]
    rowptr, col, ei = ops.radius_graph(dev(X), dev(np.array([0, n], dtype=np.int64)), 1.0)
    E = ei.t().cpu().numpy()
    d = X[E[:, 0]] - X[E[:, 1]]
    assert (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] <= 1.0).all()
    key = E[:, 0] * n + E[:, 1]
    assert (np.diff(key) > 0).all()                                    # sorted, no duplicates, no self loops needed below
    assert (E[:, 0] != E[:, 1]).all()
    assert np.array_equal(np.sort(E[:, 1] * n + E[:, 0]), key)          # symmetric edge set
    rng = np.random.default_rng(1)
    q = np.sort(rng.choice(n, 2000, replace=False))
    d2 = go._reduced_distances(X[q], X)
    hit = d2 <= 1.0
    hit[np.arange(len(q)), q] = False
    rp = rowptr.cpu().numpy()
    cc = col.cpu().numpy()
    for a, i in enumerate(q):
        assert np.array_equal(cc[rp[i]:rp[i + 1]], np.nonzero(hit[a])[0])


def test_csr_by_target_frames_equals_the_general_builder():
    """rgnn_csr_by_target_frames (one block per frame, kNN batches of many small frames) against rgnn_csr_by_target: the same
    rowptr_t / src_sorted / perm with and without a visiting order, in-degrees and per-frame counts consistent with them, and
    the row lists built from those equal rgnn_split_targets_by_node's."""
    from radargnn_amd import frames as fr, ops
    frames = [synthetic.nuscenes_frame(i) for i in range(70)] + [synthetic.small_frame(25, 3)]
    batch = fr.FrameBatch.from_frames(frames)
    k = 12
    g = fr.build_graphs(batch, fr.GraphSettings(algorithm="knn", k=k))
    n = g.x.shape[0]
    biggest = int(batch.frame_sizes.max())
    for rank in (None, g.cell_rank):
        r0, s0, p0 = ops.csr_by_target(g.edge_index, n, rank)
        r1, s1, p1, indeg, per_frame = ops.csr_by_target_frames(g.edge_index, n, k, batch.frame_ptr, biggest, rank)
        assert torch.equal(r0, r1) and torch.equal(s0, s1) and torch.equal(p0, p1)
        exp_deg = torch.bincount(g.edge_index[1], minlength=n).int()
        assert torch.equal(indeg, exp_deg)
        ptr = batch.frame_ptr.cpu()
        assert per_frame.cpu().tolist() == [int((exp_deg[ptr[f]:ptr[f + 1]] > 0).sum()) for f in range(len(frames))]
        a = ops.split_by_degree_frames(indeg, batch.frame_ptr, per_frame)
        b = ops.split_targets(r0, None if rank is None else g.cell_order, rank=rank, by_node=True)
        ce, cne = int(a[1].item()), int(a[4].item())
        assert ce == int(b[1].item()) and cne == int(b[4].item()) and ce + cne == n
        assert torch.equal(a[0][:ce], b[0][:ce]) and torch.equal(a[3][:cne], b[3][:cne]) and torch.equal(a[2], b[2])


@pytest.mark.parametrize("mode", ["directed", "undirected"])
def test_reversed_edge_features_equal_the_twins_rows(mode):
    """ops.edge_features_reversed at the own-edge list of a symmetric CSR (built WITHOUT the twin search) = the attribute rows gathered
    through the twin ids of the CSR built WITH it, bit for bit, for every feature and both modes (graph.py:139-223 on swapped end points)."""
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test but no GPU visible")
    from radargnn_amd import frames as fr, ops, synthetic
    from radargnn_amd.gnn.mpnn_layers import TargetCSR
    feats = ["point_pair_features", "spatial_euclidean_distance", "velocity_euclidean_distance", "relative_position", "relative_velocity"]
    batch = fr.FrameBatch.from_frames([synthetic.radarscenes_frame(i, n_clusters=6, pts_per_cluster=20, n_clutter=150) for i in range(5)])
    cfg = fr.GraphSettings(algorithm="radius", r=2.0, edge_features=tuple(feats), edge_mode=mode)
    g = fr.build_graphs(batch, cfg)
    assert g.edge_index.shape[1] > 1000
    kw = dict(order=g.cell_order, rank=g.cell_rank, symmetric=True, source_rows=g.rowptr, status=g.status)
    with_search = TargetCSR(g.edge_index, g.x.shape[0], **kw)
    without = TargetCSR(g.edge_index, g.x.shape[0], own_edges=True, **kw)
    assert without.own_edge is not None and torch.equal(with_search.src, without.src) and torch.equal(with_search.rowptr, without.rowptr)
    want = g.edge_attr[with_search.perm.long()]
    got, _ = ops.edge_features_reversed(batch.X, batch.V, g.edge_index, without.own_edge, feats, mode, status=g.status)
    assert got.shape == want.shape and torch.equal(got.view(torch.int32), want.view(torch.int32))
    g.check()


def test_radius_counts_in_one_launch():
    """ops.radius_counts = (rowptr[n], sum of the degrees above the threshold): what frames.build_graphs reads back per batch."""
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test but no GPU visible")
    from radargnn_amd import ops
    g = torch.Generator(device="cuda").manual_seed(3)
    for n in (0, 1, 1000, 200_003):
        deg = torch.randint(0, 90, (n,), device="cuda", dtype=torch.int32, generator=g)
        rowptr = torch.cat((torch.zeros(1, dtype=torch.int32, device="cuda"), torch.cumsum(deg, 0).to(torch.int32)))
        got = ops.radius_counts(deg, rowptr, 60).tolist()
        assert got == [int(deg.sum()), int(deg[deg > 60].sum())]


@pytest.mark.parametrize("general_path", [False, True])
def test_a_cell_of_thousands_of_coincident_points(ops, general_path, monkeypatch):
    """3 000 points on ONE spot (a cell far beyond GRID_ORDERED_CELL_MAX: it keeps the atomics' arrival order instead of the quadratic
    rank by counting) beside ordinary ones: the search still returns the oracle's neighbours (ties by index), on the per-frame grid
    kernel and on the general path, and does so at once."""
    import time
    if general_path:
        monkeypatch.setenv("RGNN_GRID_SPLIT", "1")
        __import__("radargnn_amd.ops").ops.reload_env()
    rng = np.random.default_rng(4)
    X = np.concatenate([np.full((3000, 2), 7.25), np.round(rng.normal(size=(200, 2)) * 3.0, 2)])
    X = X[rng.permutation(len(X))]
    f = synthetic.RadarFrame(X, np.zeros_like(X), np.zeros((len(X), 1)), np.zeros((len(X), 1)))
    frames = [f, synthetic.nuscenes_frame(3)]
    cat, ptr = batch(frames)
    exp = oracle_batch_edges(frames, "knn", k=4, r=None, basis="X")
    t0 = time.perf_counter()
    nbr, ei, st = ops.knn_graph(dev(cat.X), dev(ptr), 4, max_frame_points=0 if general_path else max(fr.n for fr in frames))
    torch.cuda.synchronize()
    assert time.perf_counter() - t0 < 5.0 and st.item() == 0
    got = ei.cpu().numpy().T
    assert np.array_equal(go.canonical_edges(got), go.canonical_edges(exp))


@pytest.mark.gpu
def test_register_resident_grid_block_equals_the_general_block(ops, monkeypatch):
    """k_grid_frame_reg (r06: frames of up to 4 096 points with a two-column basis keep their points in registers through the binning)
    against k_grid_frame on the same frames -- ordinary frames, a tiny one, duplicates, a cell of coincident points beyond the
    ordered-cell limit: the same cell order, the same inverse, hence the same radius and kNN graphs, entry for entry."""
    rng = np.random.default_rng(11)
    base = synthetic.radarscenes_frame(5)
    dup = np.concatenate([np.full((2500, 2), 3.5), base.X[:900]])
    frames = [synthetic.radarscenes_frame(1), synthetic.nuscenes_frame(2),
              synthetic.RadarFrame(base.X[:3], base.V[:3], base.rcs[:3], base.timestamp[:3]),
              synthetic.RadarFrame(dup, np.zeros_like(dup), np.zeros((len(dup), 1)), np.zeros((len(dup), 1))),
              synthetic.RadarFrame(base.X[rng.integers(0, 3000, 4000)], base.V[:1].repeat(4000, 0), base.rcs[:1].repeat(4000, 0),
                                   base.timestamp[:1].repeat(4000, 0))]
    cat, ptr = batch(frames)
    X, P = dev(cat.X), dev(ptr)
    biggest = max(f.n for f in frames)
    outs = []
    for no_reg in (False, True):
        if no_reg:
            monkeypatch.setenv("RGNN_GRID_NO_REG", "1")
        __import__("radargnn_amd.ops").ops.reload_env()
        g = ops.GridHash(X, P).build(cell_size=1.0, max_frame_points=biggest)
        order, rank = g.cell_order().clone(), g.cell_rank().clone()
        gk, rowptr = ops.radius_graph_count(X, P, 1.0, max_frame_points=biggest)
        n_edges = int(rowptr[-1].item())
        col, ei = ops.radius_graph_fill(gk, rowptr, 1.0, n_edges)
        nbr, kei, st = ops.knn_graph(X, P, 2, max_frame_points=biggest)
        torch.cuda.synchronize()
        outs.append((order, rank, rowptr.clone(), col.clone(), ei.clone(), nbr.clone()))
    # (the cell of 2 500 coincident points keeps the atomics' arrival order on either path -- GRID_ORDERED_CELL_MAX -- so the cell
    #  ORDER is compared on the other frames' points; the graphs are compared everywhere)
    lo, hi = int(ptr[3]), int(ptr[4])
    for k, (a, b) in enumerate(zip(*outs)):
        if k == 0:
            keep = lambda o: o[(o < lo) | (o >= hi)]
            assert torch.equal(keep(a), keep(b))
        elif k == 1:
            assert torch.equal(a[:lo], b[:lo]) and torch.equal(a[hi:], b[hi:])
        else:
            assert torch.equal(a, b)
    order = outs[0][0].long()
    assert torch.equal(torch.sort(order).values, torch.arange(order.numel(), device=order.device))


@pytest.mark.gpu
def test_search_and_fill_in_one_launch_at_known_rows(ops):
    """rgnn_radius_graph_rows_direct (r06, replayed steps): with the rows of the graph known, one launch searches and writes what count ->
    scan -> fill write -- col, both rows of edge_index, the relative_position attributes, entry for entry (rows of a few neighbours, rows
    beyond the 128-entry LDS stage in a crowded frame); on points that no longer give those rows it raises the status bit and stays
    inside the buffers."""
    frames = [synthetic.radarscenes_frame(i) for i in range(3)]
    crowd = synthetic.radarscenes_frame(7)
    frames.append(synthetic.RadarFrame(crowd.X * 0.05, crowd.V, crowd.rcs, crowd.timestamp))     # hundreds of neighbours per point
    cat, ptr = batch(frames)
    X, P = dev(cat.X), dev(ptr)
    biggest = max(f.n for f in frames)
    static = {}
    g, rowptr = ops.radius_graph_count(X, P, 1.0, static=static, max_frame_points=biggest)
    n_edges = int(rowptr[-1].item())
    assert int((rowptr[1:] - rowptr[:-1]).max()) > 128
    col, ei, rel = ops.radius_graph_fill(g, rowptr, 1.0, n_edges, relative_position="directed")
    rows = rowptr.clone()
    status = torch.zeros(1, dtype=torch.int32, device=X.device)
    g2 = ops.radius_grid(X, P, 1.0, static, max_frame_points=biggest)
    col2, ei2, rel2 = ops.radius_graph_rows_direct(g2, rows, 1.0, n_edges, status, relative_position="directed")
    torch.cuda.synchronize()
    assert int(status.item()) == 0
    assert torch.equal(col2, col) and torch.equal(ei2, ei) and torch.equal(rel2, rel)
    X.mul_(0.5)                                                  # denser: most rows grow
    g3 = ops.radius_grid(X, P, 1.0, static, max_frame_points=biggest)
    col3, ei3, _ = ops.radius_graph_rows_direct(g3, rows, 1.0, n_edges, status, relative_position="directed")
    torch.cuda.synchronize()
    assert int(status.item()) & ops.STATUS_EDGE_COUNT_CHANGED
    assert col3.shape == col.shape and ei3.shape == ei.shape
