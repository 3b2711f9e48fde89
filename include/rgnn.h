/*
 * rgnn.h -- C-ABI of librgnn.so: the MI355X (gfx950) hot path of RadarGNN.
 *
 * Scope: per-frame graph construction (k-NN / radius), undirected degree,
 * edge / node feature extraction and the DetNetBasic forward pass (MPNNConv / RadarPointGNNConv
 * message passing, train-mode BatchNorm, dense MLPs).  Every entry point below names the piece of
 * the reference (paths relative to /root/reference/src/gnnradarobjectdetection/) or of its
 * third-party native dependency that it replaces.
 *
 * Conventions
 *  - plain C, no C++ types, no exceptions cross the boundary;
 *  - every pointer marked [dev] is a device (HBM) pointer owned by the caller; the library never
 *    allocates, frees or retains memory (the caller -- PyTorch in this repo -- owns all buffers);
 *  - `stream` is a hipStream_t passed as void*; all work is enqueued on it, no entry point
 *    synchronises the device or the stream;
 *  - return value 0 = ok, negative = error (see rgnn_last_error(), thread-local);
 *  - data-dependent output sizes use the count -> scan -> fill protocol so the caller allocates
 *    exactly (radius graph);
 *  - index dtype: int32 for CSR structures (E < 2^31), int64 only where the reference hands
 *    int64 over (`edge_index`, preprocessor/radarscenes/dataset_creation.py:805);
 *  - device-side failures (k >= frame size, |dot| > 1 + 1e-3, hash overflow) are reported through
 *    a caller-provided [dev] int32 status word that the caller reads when it next synchronises.
 */
#ifndef RGNN_H
#define RGNN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* rgnn_stream_t; /* hipStream_t */

/* ---------------------------------------------------------------- error codes / status bits */
#define RGNN_OK 0
#define RGNN_ERR_INVALID_ARGUMENT (-1)
#define RGNN_ERR_LAUNCH (-2)
#define RGNN_ERR_UNSUPPORTED (-3)

#define RGNN_STATUS_KNN_TOO_FEW_POINTS 1 /* a frame has n_f <= k: sklearn raises ValueError              */
#define RGNN_STATUS_DOT_PRODUCT 2        /* graph_constructor/features.py:56,77,91 "Error in dot product" */
#define RGNN_STATUS_TIME_INDEX_OVERFLOW 4 /* more distinct timestamps in a frame than the LDS table holds  */
#define RGNN_STATUS_EDGE_COUNT_CHANGED 8 /* rgnn_radius_graph_fill_checked: rowptr[n] != the n_edges the caller sized for */
#define RGNN_STATUS_NOT_SYMMETRIC 16     /* rgnn_csr_by_target_symmetric: an edge (s,t) without its twin (t,s)               */
#define RGNN_STATUS_SPLITK_TIMEOUT 32    /* a dense launch gave up waiting for a partial tile of another work-group (see splitk_ws) */

#define RGNN_SPLITK_TIMEOUT_WORD 1000    /* index of the time-out counter in the flag area of rgnn_linear_args.splitk_ws */

const char* rgnn_version(void);
const char* rgnn_last_error(void);

/* The library reads its environment switches (RGNN_LINEAR_FP32, RGNN_DMA_TN, ...: experiment and A/B knobs, none needed in normal
 * use) once per call site and caches them.  A process that changes one of them after its first launch calls this to have every
 * site read its variable again on its next use (tools/x3_bench, bench.py's fp32-MFMA line).  Thread-safe; no reference counterpart
 * (the reference has no such switches). */
void rgnn_env_reload(void);

/* Profiling hook: arms two hipEvent_t (passed as void*) that the NEXT call of rgnn_linear_fwd / rgnn_mpnn_aggregate on
 * this thread records immediately before / after its kernel launch on the launch stream (bench.py uses it to time the
 * dominant kernels without Python between the event and the launch).  NULL disarms. */
void rgnn_profile_next_launch(void* ev_start, void* ev_stop);

/* Side streams of the host layer (radargnn_amd/ops.py independent_stream; the reference runs everything on one stream,
 * postprocessor/inference.py:48-68): a plain non-blocking HIP stream on the CURRENT device / its destruction (NULL: no-op).  The host
 * layer creates candidates here, keeps the one it has seen running beside every stream already in use (ROCm maps streams onto four
 * hardware queues; two streams on one queue serialise) and destroys the others -- torch's own pool of 32 streams cannot be given back. */
int rgnn_stream_create(rgnn_stream_t* out);
int rgnn_stream_destroy(rgnn_stream_t stream);

/* ================================================================ generic device primitives */

/* out[i] = sum_{j<i} in[j], i in [0, n]; out has n+1 entries (out[n] = total).  tmp: [dev] scratch of
 * rgnn_scan_tmp_bytes(n) bytes.  Used for cell offsets, radius rowptr, CSR-by-target rowptr. */
int64_t rgnn_scan_tmp_bytes(int64_t n);
int rgnn_exclusive_scan_i32(const int32_t* in, int32_t* out, int64_t n, void* tmp, rgnn_stream_t stream);

/* ================================================================ graph construction
 * Replaces graph_constructor/graph.py:52-82 (sklearn kneighbors_graph / radius_neighbors_graph on a
 * KDTree64 + scipy toarray()/nonzero()).  A batch is B frames laid back to back: X is [n, dim] row-major
 * float64 (dim 2 = "X", dim 4 = "XV": radarscenes/dataset_creation.py:203-206), frame_ptr [B+1] int64.
 * Neighbours are only searched inside a point's own frame; indices written are GLOBAL row numbers (PyG
 * batch numbering, utils/data_handling.py:30).
 *
 * Distances are float64, sum_j (a_j-b_j)^2 accumulated in dimension order with separate multiply and
 * add (no FMA), radius test inclusive d2 <= r*r -- bit-identical decisions to the KD-tree.            */

/* Host-side descriptor of one batch and its grid-hash workspace; filled by the caller, never retained. */
typedef struct rgnn_grid {
  const double* X;          /* [dev] [n, dim] float64 */
  int32_t dim;              /* 2 ("X"), 4 ("XV") or 8 (wider distance bases, zero-padded to 8 columns by the caller: the reference
                             * measures over ALL columns it is given, graph.py:45-50,57-58; cells are binned on the first two) */
  int64_t n;
  const int64_t* frame_ptr; /* [dev] [n_frames + 1] */
  int64_t n_frames;
  void* ws;                 /* [dev] opaque, rgnn_grid_workspace_bytes(n, n_frames, dim) bytes */
  int64_t ws_bytes;
} rgnn_grid;

int64_t rgnn_grid_workspace_bytes(int64_t n, int64_t n_frames, int32_t dim);

/* Bin the points of every frame into a uniform grid on the first two coordinates.
 *   cell_size > 0 : radius mode, cells no smaller than cell_size (3x3 neighbourhood covers radius r = cell_size)
 *   cell_size <= 0: kNN mode, cell edge chosen per frame for ~pts_per_cell points per cell             */
int rgnn_grid_build(const rgnn_grid* g, double cell_size, double pts_per_cell, rgnn_stream_t stream);
/* The same with a hint: max_frame_points > 0 = no frame of the batch holds more points than this (the caller's host copy of
 * the frame sizes).  Frames of moderate size are then binned by ONE launch -- one block per frame: bounding box, grid, cell
 * histogram, local scan, cell order -- instead of five; 0 = no promise (the general path).  Same cells either way. */
int rgnn_grid_build_frames(const rgnn_grid* g, double cell_size, double pts_per_cell, int64_t max_frame_points,
                           rgnn_stream_t stream);
/* Where the cell order (int32 [n]: point rows in grid-cell order) and its inverse (int32 [n]: position of point i in that
 * order) lie inside the workspace, as byte offsets: valid after rgnn_grid_build*, for as long as the workspace is. */
int rgnn_grid_order_offsets(int64_t n, int64_t n_frames, int32_t dim, int64_t* order_offset /*host*/, int64_t* rank_offset /*host*/);

/* Replayed (captured) steps: search + fill of a radius graph whose rows are already known -- `rowptr_committed` [n + 1] as
 * rgnn_radius_rows_commit keeps it (rowptr_committed[n] must equal n_edges).  ONE launch behind the grid build: a point's row is
 * searched once and, if it has the committed length, written in ascending order to col / edge_index (/ relative_position) at the
 * committed place; a row of another length keeps its previous contents and raises RGNN_STATUS_EDGE_COUNT_CHANGED in `status`
 * (the points changed under the captured graph).  Replaces, for those steps, rgnn_radius_graph_count + the scan +
 * rgnn_radius_rows_commit + rgnn_radius_graph_rows; same rows, same order, same values.  tmp: int32 [n_edges] scratch (rows of more
 * than 512 neighbours).  Reference: graph.py:68-82 (radius_neighbors_graph + nonzero), graph.py:199-200 (relative_position). */
int rgnn_radius_graph_rows_direct(const rgnn_grid* g, double r, const int32_t* rowptr_committed, int32_t* col, int64_t* edge_index,
                                  int64_t n_edges, int32_t* tmp, int32_t* status, float* relative_position, int32_t undirected,
                                  rgnn_stream_t stream);

/* Radius graph, pass 1: deg[i] = |{j != i in frame(i) : d2(i,j) <= r*r}|  (int32 [n]).  Also leaves the first 32
 * neighbours of every point in the grid workspace for pass 2. */
int rgnn_radius_graph_count(const rgnn_grid* g, double r, int32_t* deg /*[dev]*/, rgnn_stream_t stream);
/* Radius graph, pass 2: rowptr = exclusive scan of deg ([n+1]); col[rowptr[i]..rowptr[i+1]) = neighbours of
 * i, ascending.  Optionally also writes edge_index int64 [2,E] (row 0 = i "query", row 1 = j "neighbour",
 * graph.py:61-63 + dataset_creation.py:805); pass NULL to skip.  E = rowptr[n] is passed by the caller.
 * Must follow rgnn_radius_graph_count on the same grid (same workspace, same r): rows of up to 32 neighbours are copied
 * from what that pass found, denser rows are searched again. */
int rgnn_radius_graph_fill(const rgnn_grid* g, double r, const int32_t* rowptr /*[dev]*/, int32_t* col /*[dev]*/,
                           int64_t* edge_index /*[dev] or NULL*/, int64_t n_edges, int32_t* tmp /*[dev] int32 [2*E]*/,
                           rgnn_stream_t stream);
/* The same pass for a caller that sized col / edge_index for an edge count it did NOT just read back (a captured HIP
 * graph replayed on a resident batch): if rowptr[n] differs from n_edges the kernels write nothing -- the buffers keep
 * their previous contents -- and status gets RGNN_STATUS_EDGE_COUNT_CHANGED. */
int rgnn_radius_graph_fill_checked(const rgnn_grid* g, double r, const int32_t* rowptr /*[dev]*/, int32_t* col /*[dev]*/,
                                   int64_t* edge_index /*[dev] or NULL*/, int64_t n_edges, int32_t* tmp /*[dev] int32 [2*E]*/,
                                   int32_t* status /*[dev]*/, rgnn_stream_t stream);
/* Fill pass in ONE launch: finished rows (col ascending, edge_index) and, when relative_position != NULL, the edge attributes
 * [x_i - x_j, y_i - y_j] (graph.py:199-200; |.| of both when undirected != 0) as float [n_edges, 2] in edge order -- what
 * rgnn_radius_graph_fill / _fill_checked + rgnn_edge_features(relative_position) produce in three launches.  status != NULL:
 * the guarded form (rowptr[n] must equal n_edges, else nothing is written and RGNN_STATUS_EDGE_COUNT_CHANGED is set).
 * tmp: [dev] int32 [n_edges] scratch. */
int rgnn_radius_graph_rows(const rgnn_grid* g, double r, const int32_t* rowptr /*[dev]*/, int32_t* col /*[dev]*/,
                           int64_t* edge_index /*[dev] or NULL*/, int64_t n_edges, int32_t* tmp /*[dev]*/,
                           int32_t* status /*[dev] or NULL*/, float* relative_position /*[dev] or NULL*/, int32_t undirected,
                           rgnn_stream_t stream);
/* Companion of the checked fill for everything DOWNSTREAM of it in a replayed step (CSR by target, chunk table, the edge
 * kernels: all sized for n_edges): rowptr_new is what this replay's count + scan produced.  If rowptr_new[n] == n_edges it is
 * copied to rowptr_committed; otherwise rowptr_committed keeps the rows of the last replay that matched -- consistent with the
 * col / edge_index the checked fill then leaves untouched -- and status gets RGNN_STATUS_EDGE_COUNT_CHANGED (also when
 * n_edges == 0).  Downstream kernels read rowptr_committed only, so a replay on modified points computes on the previous
 * graph instead of walking rows that no longer fit its buffers. */
int rgnn_radius_rows_commit(const int32_t* rowptr_new /*[dev] n+1*/, int64_t n, int64_t n_edges,
                            int32_t* rowptr_committed /*[dev] n+1*/, int32_t* status /*[dev]*/,
                            const int32_t* deg_new /*[dev] n or NULL*/, int32_t* deg_committed /*[dev] n or NULL: the row
                            lengths travel with the rows (degree feature, lists of nodes with / without edges)*/,
                            rgnn_stream_t stream);

/* k nearest neighbours excluding self; nbr int32 [n,k], each row ordered (distance asc, index asc).
 * Optionally writes edge_index int64 [2, n*k].  status gets RGNN_STATUS_KNN_TOO_FEW_POINTS if a frame has
 * <= k points (rows of that frame are filled with -1). */
int rgnn_knn_graph(const rgnn_grid* g, int32_t k, int32_t* nbr /*[dev]*/, int64_t* edge_index /*[dev] or NULL*/,
                   int32_t* status /*[dev]*/, rgnn_stream_t stream);

/* rgnn_knn_graph with the write-out doing more (the team kernel has the query's coordinates and the neighbour ids in hand):
 * relative_position != NULL: float [n k, 2] edge attributes [x_i - x_j, y_i - y_j] in edge order (graph.py:199-200; |.| when
 * undirected != 0) -- what rgnn_edge_features(relative_position) computes in a launch of its own; degree_init != NULL: int32 [n]
 * preset with the out-degree k, the starting point of rgnn_undirected_degree_preset.  k <= 64 with either of them. */
int rgnn_knn_graph_attrs(const rgnn_grid* g, int32_t k, int32_t* nbr, int64_t* edge_index, int32_t* status,
                         float* relative_position, int32_t undirected, int32_t* degree_init, rgnn_stream_t stream);
/* rgnn_knn_graph_attrs for a grid built by rgnn_grid_build_frames whose largest frame the caller knows: frames of at most 1 024
 * points (nuScenes-shaped sweeps: ~300) with 3 <= k <= 32 are searched by brute force per frame -- one wave per query, every point
 * of the frame evaluated with the KD-tree's float64 arithmetic, the k-th smallest distance found by a binary search over the
 * distance bits -- instead of the walk over the grid; same rows bit for bit (distance asc, index asc).  Anything else is handed
 * to rgnn_knn_graph_attrs.  Replaces sklearn kneighbors_graph behind graph_constructor/graph.py:52-66. */
int rgnn_knn_graph_frames(const rgnn_grid* g, int32_t k, int64_t max_frame_points, int32_t* nbr, int64_t* edge_index, int32_t* status,
                          float* relative_position, int32_t undirected, int32_t* degree_init, rgnn_stream_t stream);
/* rgnn_undirected_degree for a `degree` array that already holds the out-degrees (rgnn_knn_graph_attrs): one launch. */
int rgnn_undirected_degree_preset(const int32_t* rowptr, const int32_t* col, int64_t n, int32_t* degree, rgnn_stream_t stream);
/* The same number for the rows of a kNN search (nbr int32 [n, k]) from the CSR by target of its edge list (rgnn_csr_by_target:
 * rowptr_t / src_sorted / node_order): k + in-degree - |{in-edges (s -> i) with s in nbr[i]}|; no atomics, every target's own row
 * is fetched once.  k <= 64.  Matches Graph.get_degree on the kNN adjacency (graph_constructor/graph.py:93-96). */
int rgnn_knn_degree_from_csr(const int32_t* rowptr_t, const int32_t* src_sorted, const int32_t* node_order, const int32_t* nbr, int64_t n,
                             int32_t k, int32_t* degree, rgnn_stream_t stream);
/* order[p] = global row of the p-th point in grid-cell order (frames back to back, cells row-major inside a
 * frame): a spatially coherent visiting order for the message-passing kernels (rgnn_mpnn_aggregate). */
int rgnn_grid_cell_order(const rgnn_grid* g, int32_t* order /*[dev] [n]*/, rgnn_stream_t stream);

/* Undirected degree: |{j : (i,j) in E or (j,i) in E}| for a CSR (rowptr,col) of the directed edges.
 * Replaces graph.py:93-96 (networkx Graph built from the dense adjacency).  in_deg_tmp: unused (kept for
 * ABI stability; may be NULL). */
int rgnn_undirected_degree(const int32_t* rowptr, const int32_t* col, int64_t n, int32_t* in_deg_tmp,
                           int32_t* degree_out, rgnn_stream_t stream);

/* Out-degree of every node as a rowptr: rowptr_s[p + 1] - rowptr_s[p] = number of edges whose SOURCE (edge_index row 0) is
 * the node at position p (position = rank[node], or the node id when rank is NULL).  With rgnn_split_targets this gives the
 * list of nodes that have outgoing edges -- the only rows of the source term Q = X W_j^T anybody gathers.  tmp: as for
 * rgnn_csr_by_target. */
int rgnn_source_rowptr(const int64_t* edge_index /*[dev] [2,E]*/, int64_t n, int64_t n_edges, const int32_t* rank /*[dev] or NULL*/,
                       int32_t* rowptr_s /*[dev] [n+1]*/, void* tmp, rgnn_stream_t stream);

/* rank[order[p]] = p (inverse of a visiting order such as rgnn_grid_cell_order's). */
int rgnn_invert_permutation(const int32_t* order, int64_t n, int32_t* rank, rgnn_stream_t stream);

/* CSR keyed on the aggregation target edge_index[1] (PyG flow source_to_target: messages of edge e are
 * reduced at edge_index[1][e]).  Stable: inside a segment edges keep ascending edge id, so sums are
 * deterministic.  Outputs rowptr_t int32 [n+1], src_sorted int32 [E] (= edge_index[0][perm]), perm int32 [E].
 * target_rank (optional, [dev] int32 [n]): segments are laid out in visiting order -- segment p holds the
 * edges whose target t has target_rank[t] == p -- so a kernel that visits targets in that order streams the
 * edge arrays contiguously.  NULL = natural order (segment t = target t).
 * tmp: [dev] of rgnn_csr_by_target_tmp_bytes(n, E) bytes. */
int64_t rgnn_csr_by_target_tmp_bytes(int64_t n, int64_t n_edges);
int rgnn_csr_by_target(const int64_t* edge_index /*[dev] [2,E]*/, int64_t n, int64_t n_edges,
                       const int32_t* target_rank, int32_t* rowptr_t, int32_t* src_sorted, int32_t* perm, void* tmp,
                       rgnn_stream_t stream);
/* The same without the stable order inside the segments: the places the fill's atomics hand out are the result (perm and
 * src_sorted agree with each other; which of a target's in-edges comes first varies from run to run).  For reductions that do not
 * depend on that order -- max aggregation (torch-scatter max behind mpnn_layers.py:88): one pass over the edges less. */
int rgnn_csr_by_target_unordered(const int64_t* edge_index /*[dev] [2,E]*/, int64_t n, int64_t n_edges,
                                 const int32_t* target_rank /*[dev] [n] or NULL*/, int32_t* rowptr_t, int32_t* src_sorted,
                                 int32_t* perm, void* tmp, rgnn_stream_t stream);

/* rgnn_csr_by_target for a batch of frames whose graph has UNIFORM out-degree k with edges grouped by source (kNN graphs:
 * edge e = i k + j; n_edges == n k): ONE launch, one block per frame -- histogram, scan, fill and stable ordering are local to a
 * frame because its edges are a contiguous slice and all their targets lie inside it.  max_frame_points: the caller's largest
 * frame (<= 24 576).  Optionally leaves in_degree int32 [n] (node numbering) and frame_nonempty int32 [n_frames] (nodes with
 * incoming edges per frame) behind -- the inputs of rgnn_split_by_degree_frames.  perm_tmp: [dev] int32 [n_edges] scratch.
 * Same rowptr_t / src_sorted / perm as rgnn_csr_by_target. */
int rgnn_csr_by_target_frames(const int64_t* edge_index, int64_t n, int64_t n_edges, int64_t k, const int64_t* frame_ptr,
                              int64_t n_frames, int64_t max_frame_points, const int32_t* target_rank /*[dev] or NULL*/,
                              int32_t* rowptr_t, int32_t* src_sorted, int32_t* perm, int32_t* perm_tmp,
                              int32_t* in_degree /*or NULL*/, int32_t* frame_nonempty /*or NULL*/, rgnn_stream_t stream);
/* The same result for a SYMMETRIC graph ((s,t) present <=> (t,s) present: radius graphs) whose edges are grouped by their
 * source in ascending source order with ascending targets inside a group -- exactly what rgnn_radius_graph_fill emits;
 * rowptr_src [dev] int32 [n+1] is that grouping (the search's rowptr).  A node's in-degree is then its row length and an
 * edge's place in its target's segment is the rank of its source in the target's row: no histogram, no atomics, no sort --
 * three small kernels instead of six.  Identical outputs to rgnn_csr_by_target (tests/test_gpu_graph.py).  If some (t,s) is
 * missing, *status (optional) gets RGNN_STATUS_NOT_SYMMETRIC and that edge is left out.  tmp: as above. */
int rgnn_csr_by_target_symmetric(const int64_t* edge_index /*[dev] [2,E]*/, const int32_t* rowptr_src, int64_t n,
                                 int64_t n_edges, const int32_t* target_rank, int32_t* rowptr_t, int32_t* src_sorted,
                                 int32_t* perm, void* tmp, int32_t* status /*[dev] or NULL*/, rgnn_stream_t stream);
/* rgnn_csr_by_target_symmetric without the twin search (r03): instead of perm (edge id of the in-edge at every slot) it writes
 * own_edge: the id of the OUT-edge (t -> i) at the slot of its twin (i -> t).  For callers whose edge attributes are
 * antisymmetric under reversal -- relative_position in directed mode: attr(i -> t) = -attr(t -> i), exactly -- the target-ordered
 * attributes are then -attr[own_edge[slot]] and the binary search per edge is not needed.
 * SYMMETRY IS NOT CHECKED on this path (no twin search: an edge (t -> i) without (i -> t) would silently appear as an in-edge
 * with negated attributes; RGNN_STATUS_NOT_SYMMETRIC is never set): only for edge lists that are symmetric by construction --
 * the output of rgnn_radius_graph_fill.  Anything else goes through rgnn_csr_by_target_symmetric, which checks. */
int rgnn_csr_by_target_symmetric_own(const int64_t* edge_index, const int32_t* rowptr_src, int64_t n, int64_t n_edges,
                                     const int32_t* target_rank, int32_t* rowptr_t, int32_t* src_sorted, int32_t* own_edge,
                                     void* tmp, int32_t* status, rgnn_stream_t stream);

/* ================================================================ features
 * Edge feature codes, concatenated in list order (graph.py:139-223).                                  */
#define RGNN_EF_POINT_PAIR 0          /* 4: d, theta(v1,v2), theta(d,v1), theta(d,v2)  features.py:6-122 */
#define RGNN_EF_SPATIAL_DISTANCE 1    /* 1 */
#define RGNN_EF_VELOCITY_DISTANCE 2   /* 1 */
#define RGNN_EF_RELATIVE_POSITION 3   /* 2: X_i - X_j, i = E[:,0], j = E[:,1] */
#define RGNN_EF_RELATIVE_VELOCITY 4   /* 2 */
/* Node feature codes (graph.py:225-275). */
#define RGNN_NF_RCS 0
#define RGNN_NF_TIME_INDEX 1
#define RGNN_NF_DEGREE 2
#define RGNN_NF_VELOCITY_LENGTH 3
#define RGNN_NF_VELOCITY_VECTOR 4     /* 2 */
#define RGNN_NF_SPATIAL_COORDINATES 5 /* 2 */
#define RGNN_MAX_FEATURE_CODES 16

/* out[e, :] for every edge; X,V float64 [n,2]; edge_index int64 [2,E]; out row-major [E, width] where width
 * is implied by the codes; out_is_f64 selects float64 (GeometricGraph.E_feat) or float32 (create_graph_data,
 * dataset_creation.py:806).  undirected != 0 selects edge_mode "undirected". */
int rgnn_edge_features(const double* X, const double* V, const int64_t* edge_index, int64_t n_edges,
                       const int32_t* codes /*host*/, int32_t n_codes, int32_t undirected, void* out,
                       int32_t out_is_f64, int32_t* status /*[dev]*/, rgnn_stream_t stream);

/* The same features for the REVERSED edges at the rows of a list: out[s] = features(E[1][e], E[0][e]) with e = reversed_of[s]
 * (int32 [n_rows]).  With reversed_of = the own_edge list of rgnn_csr_by_target_symmetric_own this is the attribute list in target
 * order of a symmetric graph -- bit-identical to gathering every in-edge's own row (same arithmetic, same end points) -- without the
 * search for the twin's edge id.  Replaces the same reference lines as rgnn_edge_features (graph.py:139-223) plus the re-ordering that
 * torch_geometric's scatter makes unnecessary there. */
int rgnn_edge_features_reversed(const double* X, const double* V, const int64_t* edge_index, int64_t n_edges,
                                const int32_t* reversed_of, int64_t n_rows, const int32_t* codes, int32_t n_codes,
                                int32_t undirected, void* out, int32_t out_is_f64, int32_t* status, rgnn_stream_t stream);

/* Node feature matrix (graph.py:225-275): any of rcs/time_index [n] float64, degree int32 [n] may be NULL when
 * its code is not requested. */
int rgnn_node_features(const double* X, const double* V, const double* rcs, const double* time_index,
                       const int32_t* degree, int64_t n, const int32_t* codes /*host*/, int32_t n_codes, void* out,
                       int32_t out_is_f64, rgnn_stream_t stream);

/* time_index[i] = rank of timestamp[i] among the sorted distinct timestamps of its frame
 * (radarscenes/dataset_creation.py:214-223, nuscenes/conversion.py:94-103). */
int rgnn_time_index(const double* timestamp, const int64_t* frame_ptr, int64_t n_frames, double* time_index,
                    int32_t* status /*[dev]*/, rgnn_stream_t stream);
/* rgnn_time_index for batches with LARGE frames (one 100 000-point cloud: one block per frame walks it alone in 160 us): the points
 * are spread over the chip -- a hash set of the frame's distinct timestamps in the caller's workspace (rgnn_time_index_ws_bytes),
 * one sort per frame, one rank look-up per point; same indices (dataset_creation.py:214-223: rank among np.unique of the frame). */
int64_t rgnn_time_index_ws_bytes(int64_t n_frames);
int rgnn_time_index_ws(const double* timestamp, const int64_t* frame_ptr, int64_t n_frames, int64_t n, double* time_index,
                       int32_t* status, void* ws /*[dev] rgnn_time_index_ws_bytes, 16-byte aligned*/, int64_t ws_bytes,
                       rgnn_stream_t stream);
/* rgnn_time_index + rgnn_node_features in one launch (one block per frame): the time index of a point goes straight into its
 * column of the feature row; same values as the two calls (graph.py:225-275, dataset_creation.py:214-223). */
int rgnn_node_features_time_index(const double* X, const double* V, const double* rcs, const double* timestamp,
                                  const int64_t* frame_ptr, int64_t n_frames, const int32_t* degree /*[dev] or NULL*/, int64_t n,
                                  const int32_t* codes /*host*/, int32_t n_codes, void* out, int32_t out_is_f64,
                                  int32_t* status /*[dev]*/, int32_t* frame_nonempty /*[dev] int32 [n_frames] or NULL: nodes with
                                  degree > 0 per frame, for rgnn_split_by_degree_frames*/, rgnn_stream_t stream);
/* rgnn_split_targets_by_node for a SYMMETRIC graph (radius graphs: a node has incoming edges iff its own row is not empty) from
 * the degrees and the per-frame counts rgnn_node_features_time_index left behind: one launch (one block per frame) instead of
 * four.  Same lists, counts and slots. */
int rgnn_split_by_degree_frames(const int32_t* degree, const int64_t* frame_ptr, int64_t n_frames, const int32_t* frame_nonempty,
                                int32_t* list_empty, int64_t* count_empty, int32_t* slot_of_node /*or NULL*/,
                                int32_t* list_nonempty, int64_t* count_nonempty, rgnn_stream_t stream);

/* What the host reads back of a radius graph before it sizes the edge arrays (frames.build_graphs: one 8-byte copy per batch):
 * out2[0] = rowptr[n] (the edge count of rgnn_radius_graph_count + scan), out2[1] = edges in rows longer than `threshold` (symmetric
 * graph: in-degree = row length; radargnn_amd/gnn/mpnn_layers.py picks the form of the max aggregation by that share).  One launch. */
int rgnn_radius_counts(const int32_t* deg, int64_t n, const int32_t* rowptr, int32_t threshold, int32_t* out2, rgnn_stream_t stream);

/* ================================================================ dense layers (fp32, MFMA)
 * out[m, n] = act( sum_k A'[m,k] * W[n,k] + bias[n] ) (+ residual[m,n])
 * Replaces ATen addmm behind torch_geometric.nn.dense.linear.Linear (gnn/gnn_models.py:137-178,
 * gnn/mpnn_layers.py:64-74,89-90).  A' is the row-concatenation [A1 | A2] (torch.cat([x, m_emb], -1) at
 * mpnn_layers.py:89 without materialising it); W rows n < w_split come from W1, rows n >= w_split from W2
 * (lets one launch produce several projections of the same input).  If col_stats != NULL the kernel also
 * writes the column statistics (RGNN_STAT_ROWS floats per column and 128-row panel) of the stored `out`, the input of the
 * train-mode BatchNorm that follows every conv (gnn_models.py:126). */
/* Column statistics are kept per panel of 128 rows as RGNN_STAT_ROWS floats per column: {count of rows, a pivot taken from the
 * data, s1 = sum of (v - pivot), s2 = sum of (v - pivot)^2}.  Sums of small differences are exact or nearly so, so a column
 * whose spread is tiny against its mean loses nothing to cancellation or to the rounding of a mean (a constant column: s1 == s2
 * == 0 exactly); consumers form mean and variance in float64 (a zero-filled panel counts nothing).  (r03 kept {sum, sum of
 * squares} in float32: 1e-7 (mean / std)^2 relative in the variance.) */
#define RGNN_STAT_ROWS 4
/* A BatchNorm-apply table is RGNN_AFFINE_ROWS rows of n floats: {mean_hi, g, t} with y = (x - mean_hi) g + t, mean_hi = fl(mean),
 * g = fl(gamma / sqrt(var + eps)), t = fl(beta - (mean - mean_hi) g).  (r03 kept {scale, shift} and evaluated x scale + shift,
 * like ATen's CPU kernel: the rounding of mean * scale shows as 6e-8 |mean| / std in the normalised value.) */
#define RGNN_AFFINE_ROWS 3
typedef struct rgnn_linear_args {
  const float* A1; int64_t lda1; int32_t k1;    /* [M,k1]  */
  const float* A2; int64_t lda2; int32_t k2;    /* [M,k2] or NULL/0 */
  const float* W1; const float* W2; int64_t ldw; int32_t w_split; /* W rows have k1+k2 contiguous floats */
  const float* bias1; const float* bias2;       /* bias for rows < w_split / >= w_split; NULL = 0 */
  const float* residual; int64_t ldr;           /* added after the activation (RadarPointGNNConv h + x) */
  float* out; int64_t ldo;
  int64_t m; int32_t n;
  int32_t relu_out;
  float* col_stats;                             /* [panels, RGNN_STAT_ROWS, n] fp32 or NULL; panels = rgnn_linear_stat_panels(m) */
  /* Row subsets (all optional): tile row r works on matrix row row_index[r] of A1/A2/residual/out; the number of
   * rows is read from device memory (m_dev, with m an upper bound used for the launch geometry); accumulate: out +=
   * result (no col_stats with it).  Panels of col_stats that receive no rows are not written: zero the buffer first, or hand
   * the launch's row count to rgnn_batchnorm_finalize_parts. */
  const int32_t* row_index;
  const int64_t* m_dev;
  int32_t accumulate;
  int32_t gather_only;               /* with row_index: gather the A rows only, write a compact [count, n] result */
  const int32_t* residual_index;     /* per output row: row of `residual` to add, -1 = none (NULL: same row) */
  /* Optional: the weight [W1; W2] pre-split into three bf16 planes by rgnn_linear_split_weights (3 n kp uint16, laid
   * out [kp / 32][3][n][32], kp = rgnn_linear_planes_kp(k1 + k2)).  When given (and no residual / row_index is used) the product runs on the
   * bf16 matrix pipe as a six-term split product with fp32 accumulation -- as accurate as the fp32 MFMA path (see
   * linear.hip), about twice as fast.  W1 / W2 must still be passed (they define the layer). */
  const void* W_planes;
  int32_t w_planes_kp;
  /* Optional [dev] scratch of rgnn_linear_splitk_ws_bytes() bytes, ZEROED once by the caller and then left to the library
   * (every launch leaves its flag words zero again).  With it the LDS-DMA kernel divides the (tile, k-step) units of a launch
   * evenly over the work-groups instead of dealing whole tiles (no partial tile rounds); a tile cut between two work-groups
   * is handed over through this scratch and accumulated in the order of the undivided tile: results are bit-identical with
   * and without it.  One launch at a time per scratch buffer (launches on one stream qualify).
   * A work-group that waits for a partial tile longer than ~1 s gives up (a wrong tile is a better failure than a device that
   * never comes back) and counts it in int32 word RGNN_SPLITK_TIMEOUT_WORD of the flag area, i.e. at byte offset
   * rgnn_linear_splitk_ws_bytes() - 4096 + 4 * RGNN_SPLITK_TIMEOUT_WORD of the scratch: callers that read device status words
   * back should read this one too (radargnn_amd: GraphBatch.check() raises RGNN_STATUS_SPLITK_TIMEOUT). */
  void* splitk_ws;
  int64_t splitk_ws_bytes;
  /* Optional: the A1 operand is act((A1 - mean) g + t) per column -- the train-mode BatchNorm + ReLU that precedes the layer
   * (gnn_models.py:126-128), applied to the activation fragment on its way into the matrix pipe instead of in a pass of its
   * own over [M, k1].  a1_scale_shift: [dev] float [RGNN_AFFINE_ROWS, k1] (what rgnn_batchnorm_finalize writes: A1' = act((A1 - mean_hi) g + t));
   * a1_relu: clamp at 0 afterwards.  The LDS-DMA kernel and the fp32 kernel on buffer-descriptor operands do this: ask rgnn_linear_fwd_fuses_a1_affine(args) first and
   * otherwise apply rgnn_scale_shift_act to A1 (rgnn_linear_fwd returns RGNN_ERR_UNSUPPORTED rather than ignore it). */
  const float* a1_scale_shift;
  int32_t a1_relu;
  /* relu_out applies to the output columns >= relu_from_col only (0: all).  Lets one launch compute the first Linear of
   * both heads (gnn_models.py:131-132: logits without activation | hidden layer of the box head with ReLU) from one pass
   * over the node features.  Not on the <= 8-wide-input kernel (the call then takes the general one). */
  int32_t relu_from_col;
  /* Optional (r03): the f16x2 form of the LDS-DMA kernel -- every fp32 operand as TWO f16 terms after an exact power-of-two
   * pre-scale, THREE matrix-pipe products per fp32 product instead of the six of the bf16x3 form (a = h + l carries 22
   * significand bits; the dropped l l' term is <= 2^-22 |a a'|: as accurate as the fp32 MFMA kernel).  Taken when the launch
   * qualifies for the LDS-DMA kernel (rgnn_linear_fwd_path) and all of these are given:
   *   W_planes_f16: rgnn_linear_planes_f16_bytes(n, k1 + k2) bytes written by rgnn_linear_split_weights_f16;
   *   a1_bound / a2_bound: [dev] bounds (RGNN_BOUND_SLOTS floats each, see below) holding an UPPER BOUND of |A1'| (A1 after a1_scale_shift / a1_relu) and of |A2|
   *     (a2_bound only when k2 > 0).  Elements beyond the bound overflow to infinity; a bound up to 2^19 above the tensor's
   *     typical magnitude costs no accuracy.  Producers: out_absmax of the launch that wrote the operand,
   *     rgnn_mpnn_aggregate_flags' out_absmax, rgnn_batchnorm_bound for an operand behind a1_scale_shift.
   * out_absmax: [dev] bound (RGNN_BOUND_SLOTS floats), raised to max |out| over everything this launch stores (zero it first; LDS-DMA kernel
   *   only, either form: rgnn_linear_fwd returns RGNN_ERR_UNSUPPORTED if another kernel would take the launch). */
  const void* W_planes_f16;
  const float* a1_bound;
  const float* a2_bound;
  float* out_absmax;
  /* Optional (r03), with row_index AND a1_scale_shift: per-SEGMENT scale / shift -- the train-mode BatchNorm of a batch whose
   * frames are normalised with their OWN statistics (the reference's one-frame-per-forward inference loop, evaluate.py:40 +
   * gnn_models.py:124-128), still applied on the way into the matrix pipe.  a1_scale_shift is then [S, RGNN_AFFINE_ROWS, k1]
   * (rgnn_batchnorm_segments_from_panels) and a1_panel_segment: [dev] int32 [ceil(m / 256)] names the table of every 256-row
   * tile of the row list -- the list keeps a segment's rows inside tiles of their own, padded with -1 entries
   * (rgnn_pad_list_by_segment writes list and map).  LDS-DMA kernel only: rgnn_linear_fwd_fuses_a1_affine says whether the
   * launch qualifies.  A row_index entry of -1 is an absent row there: nothing is read, stored or counted for it -- ON THE LDS-DMA
   * KERNEL ONLY (rgnn_linear_fwd_path != 0), with or without tables; no other kernel checks, so ask before handing over a padded
   * list.  As for every
   * row list, m bounds the matrix rows the list may name; the padded list itself may be longer (*m_dev > m is fine), and
   * col_stats then needs rgnn_linear_stat_panels(*m_dev) panels. */
  const int32_t* a1_panel_segment;
} rgnn_linear_args;
/* A "bound" in this header is an array of RGNN_BOUND_SLOTS floats in device memory, zeroed by the caller before its first
 * producer runs: producers raise individual slots (atomic max, one slot per work-group), consumers take the maximum over all
 * slots.  To hand in a bound computed elsewhere, write it to slot 0 and zero the rest. */
#define RGNN_BOUND_SLOTS 256
enum { RGNN_LINEAR_PATH_OTHER = 0, RGNN_LINEAR_PATH_DMA_BF16X3 = 1, RGNN_LINEAR_PATH_DMA_F16X2 = 2 };
/* Which kernel family rgnn_linear_fwd would launch for these arguments (host-side, no launch). */
int32_t rgnn_linear_fwd_path(const rgnn_linear_args* args /*host*/);
int64_t rgnn_linear_planes_f16_bytes(int32_t n, int32_t k);
int rgnn_linear_split_weights_f16(const float* W1, const float* W2, int64_t ldw, int32_t w_split, int32_t n, int32_t k,
                                  void* planes /*[dev] rgnn_linear_planes_f16_bytes(n, k) bytes*/, rgnn_stream_t stream);
int32_t rgnn_linear_fwd_fuses_a1_affine(const rgnn_linear_args* args /*host*/);
int64_t rgnn_linear_splitk_ws_bytes(void);
int64_t rgnn_linear_stat_panels(int64_t m);
int32_t rgnn_linear_planes_kp(int32_t k);
int rgnn_linear_split_weights(const float* W1, const float* W2, int64_t ldw, int32_t w_split, int32_t n, int32_t k,
                              void* planes /*[dev] uint16, 3 n kp of them*/, rgnn_stream_t stream);
int rgnn_linear_fwd(const rgnn_linear_args* args /*host*/, rgnn_stream_t stream);

/* Three Linear layers back to back in one pass: out = act3(W3 relu(W2 relu(W1 x + b1) + b2) + b3) for k0 <= 8 inputs and layers
 * of 32, 64 and 128 columns -- DetNetBasic's node embedding (gnn_models.py:137-178, node_feature_embedding_layer_dimensions
 * [32, 64, 128, ...]; rgnn_embed3_supported says whether a shape qualifies).  W1 fp32 [n1, k0] (row stride ldw1); W2 / W3 as the
 * f16 planes rgnn_linear_split_weights_f16 writes for [n2, n1] / [n3, n2]; layers 2 and 3 run in the f16x2 form of
 * rgnn_linear_fwd, pre-scaled per 32-row block from maxima computed on the way.  out_absmax: optional bound of |out| (see
 * RGNN_BOUND_SLOTS). */
int32_t rgnn_embed3_supported(int32_t k0, int32_t n1, int32_t n2, int32_t n3);
int rgnn_embed3(const float* x, int64_t ldx, int32_t k0, const float* W1, int64_t ldw1, const float* b1, int32_t n1,
                const void* W2_planes_f16, const float* b2, int32_t n2, const void* W3_planes_f16, const float* b3, int32_t n3,
                int32_t relu3, int64_t m, float* out, int64_t ldo, float* out_absmax, rgnn_stream_t stream);
/* Two tiny Linear layers back to back on (optionally gathered) rows: out[r] = act2(W2 act1(W1 a[row_index[r]] + b1) + b2),
 * k0 <= 8 inputs, n1 <= 8 hidden, n2 <= 16 outputs; row_index NULL = identity.  The hidden layers of DetNetBasic's edge
 * embedding (gnn_models.py:48-52,137-178) on the edge attributes in CSR-by-target order: one pass instead of
 * rgnn_gather_rows_f32 + two rgnn_linear_fwd launches, same bits. */
int rgnn_tiny_mlp2(const float* A, int64_t lda, int32_t k0, const int32_t* row_index /*[dev] or NULL*/, int64_t m,
                   const float* W1, int64_t ldw1, const float* b1, int32_t n1, int32_t relu1, const float* W2, int64_t ldw2,
                   const float* b2, int32_t n2, int32_t relu2, float* out, int64_t ldo, rgnn_stream_t stream);

/* ================================================================ BatchNorm1d (gnn_models.py:71-73,126-128)
 * Training: combine col_stats (float64) -> batch mean / biased variance -> the apply table {mean_hi, g, t} (RGNN_AFFINE_ROWS);
 * running stats updated with momentum and the unbiased variance; num_batches_tracked += 1.
 * Eval (training == 0): the table from the running statistics, col_stats ignored. */
int rgnn_batchnorm_finalize(const float* col_stats, int64_t panels, int64_t m, int32_t n, const float* gamma,
                            const float* beta, float* running_mean, float* running_var,
                            int64_t* num_batches_tracked, int32_t training, float momentum, float eps,
                            float* scale_shift /*[RGNN_AFFINE_ROWS,n]*/, rgnn_stream_t stream);
/* The same for a layer whose rows were produced by TWO row-subset launches (mpnn_layers.py:89-90 on the targets with and
 * without incoming edges), each with a col_stats buffer of its own: rows_a / rows_b ([dev], optional) are the m_dev row
 * counts of those launches; only the ceil(rows / 128) panels a launch really wrote are read, so the buffers need no zero fill.
 * stats_b may be NULL (one part). */
int rgnn_batchnorm_finalize_parts(const float* stats_a, int64_t panels_a, const int64_t* rows_a /*[dev] or NULL*/,
                                  const float* stats_b, int64_t panels_b, const int64_t* rows_b /*[dev] or NULL*/,
                                  int64_t m, int32_t n, const float* gamma, const float* beta, float* running_mean,
                                  float* running_var, int64_t* num_batches_tracked, int32_t training, float momentum,
                                  float eps, float* scale_shift /*[RGNN_AFFINE_ROWS,n]*/, rgnn_stream_t stream);
/* The same, additionally propagating a bound for the f16x2 dense form (rgnn_linear_args.a1_bound): in_bound [dev] (a bound, RGNN_BOUND_SLOTS floats) holds an upper
 * bound B of |x| over the matrix the statistics were taken of (the producing launches' out_absmax); out_bound ([dev] bound,
 * zeroed by the caller) receives max over the columns of |g| (B + |mean_hi|) + |t| -- an upper bound of |(x - mean_hi) g + t|, hence of
 * the ReLU of it.  stats_b / rows_* / the bounds may be NULL. */
int rgnn_batchnorm_finalize_bound(const float* stats_a, int64_t panels_a, const int64_t* rows_a /*[dev] or NULL*/,
                                  const float* stats_b, int64_t panels_b, const int64_t* rows_b /*[dev] or NULL*/,
                                  int64_t m, int32_t n, const float* gamma, const float* beta, float* running_mean,
                                  float* running_var, int64_t* num_batches_tracked, int32_t training, float momentum,
                                  float eps, float* scale_shift /*[RGNN_AFFINE_ROWS,n]*/, const float* in_bound, float* out_bound,
                                  rgnn_stream_t stream);
/* Train-mode BatchNorm with statistics PER SEGMENT of rows -- segment f = rows [seg_ptr[f], seg_ptr[f + 1]), one segment per
 * radar frame of a batch.  The reference runs inference one frame per forward and never calls .eval() (evaluate.py:40,
 * postprocessor/inference.py:57-62, gnn/gnn_models.py:124-128): every frame is normalised with its own batch statistics; this
 * reproduces that in a batched launch.  table [dev] float [n_seg, RGNN_AFFINE_ROWS, n] receives the apply table of every segment; the running
 * statistics are walked through the segments in order (what a loop of single-frame forwards does), num_batches_tracked grows
 * by the number of non-empty segments.  seg_sums: [dev] scratch, double [n_seg, 2, n].  in_bound / out_bound: as in
 * rgnn_batchnorm_finalize_bound (optional). */
int rgnn_batchnorm_segments(const float* x, int64_t ldx, const int64_t* seg_ptr /*[dev] n_seg + 1*/, int64_t n_seg, int32_t n,
                            const float* gamma, const float* beta, float* running_mean, float* running_var,
                            int64_t* num_batches_tracked, float momentum, float eps, double* seg_sums, float* table,
                            const float* in_bound, float* out_bound, rgnn_stream_t stream);
/* rgnn_batchnorm_segments + rgnn_scale_shift_act_segments in two launches instead of three: the block that sums a (segment,
 * 64-channel slab) also normalises it (its second pass over the slab hits L2), y = act((x - mean_hi) g + t); the second launch walks
 * the running statistics, writes the table and the bound as rgnn_batchnorm_segments does.  y != x. */
int rgnn_batchnorm_act_segments(const float* x, int64_t ldx, const int64_t* seg_ptr, int64_t n_seg, int32_t n, const float* gamma,
                                const float* beta, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                                float momentum, float eps, int32_t relu, double* seg_sums, float* table, float* y, int64_t ldy,
                                const float* in_bound, float* out_bound, rgnn_stream_t stream);
/* The same table from column statistics the dense launches already left behind (rgnn_linear_args.col_stats: one partial per
 * 128-row panel of the launch's row list) instead of a pass over x -- for row lists padded per segment
 * (rgnn_pad_list_by_segment): segment f owns panels [panel_start_a[f], panel_start_a[f + 1]) of list a and likewise of list b
 * (b optional: the conv layers run two launches, targets with and without edges).  seg_ptr: rows per segment as above (the
 * counts the variances divide by).  seg_sums: scratch, double [n_seg, 2, n]. */
int rgnn_batchnorm_segments_from_panels(const float* stats_a, const int32_t* panel_start_a, const float* stats_b,
                                        const int32_t* panel_start_b, const int64_t* seg_ptr, int64_t n_seg, int32_t n,
                                        const float* gamma, const float* beta, float* running_mean, float* running_var,
                                        int64_t* num_batches_tracked, float momentum, float eps, double* seg_sums, float* table,
                                        const float* in_bound, float* out_bound, rgnn_stream_t stream);
/* An ascending row list cut at the segment borders and padded with -1 so that every segment starts a 256-row tile of its own:
 * list int32 [*count] ascending, seg_ptr int64 [n_seg + 1] (row ranges of the segments) -> out_list (room for *count + 256 n_seg
 * entries), *out_count, tile_segment int32 [out_count / 256] (rgnn_linear_args.a1_panel_segment), stat_panel_start int32
 * [n_seg + 1] in units of the 128-row statistics panels (rgnn_batchnorm_segments_from_panels).  All [dev]. */
int rgnn_pad_list_by_segment(const int32_t* list, const int64_t* count, const int64_t* seg_ptr, int64_t n_seg, int32_t* out_list,
                             int64_t* out_count, int32_t* tile_segment, int32_t* stat_panel_start, rgnn_stream_t stream);
/* Two lists over the same segments in ONE launch (a batch's targets with / without edges). */
int rgnn_pad_list_pair_by_segment(const int32_t* list_a, const int64_t* count_a, const int32_t* list_b, const int64_t* count_b,
                                  const int64_t* seg_ptr, int64_t n_seg, int32_t* out_list_a, int64_t* out_count_a,
                                  int32_t* tile_segment_a, int32_t* stat_panel_start_a, int32_t* out_list_b, int64_t* out_count_b,
                                  int32_t* tile_segment_b, int32_t* stat_panel_start_b, rgnn_stream_t stream);
/* y[r] = (x[r] - mean_hi[seg(r)]) g[seg(r)] + t[seg(r)], optional ReLU, with the table of rgnn_batchnorm_segments; in place allowed. */
int rgnn_scale_shift_act_segments(const float* x, int64_t ldx, const float* table, const int64_t* seg_ptr, int64_t n_seg,
                                  int64_t m, int32_t n, int32_t relu, float* y, int64_t ldy, rgnn_stream_t stream);
/* Column statistics of an arbitrary [m,n] matrix in the panel layout above (BatchNorm on an input that did not
 * come out of rgnn_linear_fwd, e.g. BatchNorm modules called on their own). */
int rgnn_column_stats(const float* x, int64_t ldx, int64_t m, int32_t n, float* col_stats /*[panels,RGNN_STAT_ROWS,n]*/,
                      rgnn_stream_t stream);
/* y = (x - mean_hi) g + t with a table of rgnn_batchnorm_finalize, optional ReLU (F.relu, gnn_models.py:128); in place allowed. */
int rgnn_scale_shift_act(const float* x, int64_t ldx, const float* scale_shift, int64_t m, int32_t n, int32_t relu,
                         float* y, int64_t ldy, rgnn_stream_t stream);

/* ================================================================ message passing
 * Fused gather + per-edge linear + segmented reduce for a message MLP that is a single Linear
 * (pre_layers == 1, mpnn_layers.py:64-68), using linearity:
 *   reduce_e( W_i x_t + W_j x_s + W_e a_e + b ) = P[t] (.) + reduce_e( Q[s_e] + W_e a_e )
 * P (may be NULL: RadarPointGNNConv has no x_i term, only `p_bias`) and Q are node-wise projections
 * produced by rgnn_linear_fwd.  Edges are visited in CSR-by-target order; `edge_attr_sorted` is
 * edge_attr[perm].  aggr: 0 = max, 1 = mean, 2 = add; empty segments give exactly 0 (torch-scatter).
 * `node_order` (with a CSR built with the matching `target_rank`) only changes the order in which targets are
 * processed (cache locality), never the result: segment p of the CSR belongs to target node_order[p].
 * Replaces MessagePassing.propagate (index_select gathers + cat + addmm + scatter) at
 * mpnn_layers.py:88,94-101 / :173,179-184. */
#define RGNN_AGGR_MAX 0
#define RGNN_AGGR_MEAN 1
#define RGNN_AGGR_ADD 2
int rgnn_mpnn_aggregate(const float* P, int64_t ldp, const float* p_bias, const float* Q, int64_t ldq,
                        const float* We /*[d, de], row stride ldwe*/, int64_t ldwe, const float* edge_attr_sorted,
                        int32_t de, const int32_t* rowptr_t, const int32_t* src_sorted,
                        const int32_t* node_order /*[dev] [n] visiting order of the targets, or NULL*/,
                        const int32_t* chunk_start /*[dev] from rgnn_mpnn_partition, or NULL*/, int32_t n_chunks, int64_t n,
                        int32_t d, int32_t aggr, float* out, int64_t ldo, rgnn_stream_t stream);

/* Same with options.  RGNN_MPNN_SKIP_EMPTY_ROWS: rows of `out` that belong to targets without incoming edges may be left
 * unwritten (callers that run the update only on the other rows -- rgnn_split_targets -- never read them; it saves their
 * share of the output traffic). */
#define RGNN_MPNN_SKIP_EMPTY_ROWS 1
int rgnn_mpnn_aggregate_flags(const float* P, int64_t ldp, const float* p_bias, const float* Q, int64_t ldq,
                              const float* We, int64_t ldwe, const float* edge_attr_sorted, int32_t de,
                              const int32_t* rowptr_t, const int32_t* src_sorted, const int32_t* node_order,
                              const int32_t* chunk_start, int32_t n_chunks, int64_t n, int32_t d, int32_t aggr, float* out,
                              int64_t ldo, int32_t flags, rgnn_stream_t stream);
/* Same, additionally tracking out_absmax ([dev] bound of RGNN_BOUND_SLOTS floats, zeroed by the caller; NULL = not wanted): the maximum of |out|
 * over the rows written -- the bound the f16x2 dense form wants for its A2 operand (rgnn_linear_args.a2_bound).  The fused
 * max kernel tracks it per lane at no measurable cost; other kernels are followed by one pass over `out`. */
int rgnn_mpnn_aggregate_absmax(const float* P, int64_t ldp, const float* p_bias, const float* Q, int64_t ldq,
                               const float* We, int64_t ldwe, const float* edge_attr_sorted, int32_t de,
                               const int32_t* rowptr_t, const int32_t* src_sorted, const int32_t* node_order,
                               const int32_t* chunk_start, int32_t n_chunks, int64_t n, int32_t d, int32_t aggr, float* out,
                               int64_t ldo, int32_t flags, float* out_absmax, rgnn_stream_t stream);
/* Max aggregation without a target term, recording the winners for the backward pass: arg_out uint16 [n, d] receives, per
 * target with incoming edges and channel, the index INSIDE the target's segment of the first edge that attains the maximum
 * (in-degrees must stay below 65 536).  Only the fused max kernel can record them (d % 8 == 0, 16-byte aligned rows, chunk
 * table given, de <= 8): *arg_written (host) says whether it ran -- if 0, out is complete but arg_out is untouched
 * (rgnn_mpnn_max_bwd then recomputes the winners). */
int rgnn_mpnn_aggregate_max_arg(const float* p_bias, const float* Q, int64_t ldq, const float* We, int64_t ldwe,
                                const float* edge_attr_sorted, int32_t de, const int32_t* rowptr_t, const int32_t* src_sorted,
                                const int32_t* node_order, const int32_t* chunk_start, int32_t n_chunks, int64_t n, int32_t d,
                                float* out, int64_t ldo, uint16_t* arg_out, int32_t flags, int32_t* arg_written /*host*/,
                                rgnn_stream_t stream);
/* ... the same launch also tracking max |out| into out_absmax (a bound, RGNN_BOUND_SLOTS words; NULL: rgnn_mpnn_aggregate_max_arg),
 * so that the update GEMM of a TRAINING forward takes the f16x2 form like the inference one.  *arg_written: bit 0 = winners
 * recorded, bit 1 = max |out| tracked (over the rows written). */
int rgnn_mpnn_aggregate_max_arg_absmax(const float* p_bias, const float* Q, int64_t ldq, const float* We, int64_t ldwe,
                                       const float* edge_attr_sorted, int32_t de, const int32_t* rowptr_t, const int32_t* src_sorted,
                                       const int32_t* node_order, const int32_t* chunk_start, int32_t n_chunks, int64_t n, int32_t d,
                                       float* out, int64_t ldo, uint16_t* arg_out, int32_t flags, int32_t* arg_written /*host*/,
                                       float* out_absmax, rgnn_stream_t stream);

/* ---- window form of the max aggregation (mpnn_tiles.hip; what the folded layers run on graphs it pays for) ---------------------
 * The same sum as rgnn_mpnn_aggregate_absmax with aggr = max, P = NULL and de <= 8 (the folded layers the models ship), computed
 * 32 edges x 32 channels at a time: the gathered Q values are the accumulator's initial value of v_mfma_f32_32x32x16_bf16 and
 * W_e z_e comes from the matrix pipe, z and W_e each split exactly into three bf16 terms (six products, fp32 accumulate:
 * ~2^-22 |z||w| from the fp32 chain of the per-edge kernel -- a frame's low bits therefore depend on which kernel its batch's
 * density and size select).  The rows of Q are not gathered per edge: a window of ~15-40 consecutive targets (<= 8 streams of
 * <= 64 slots, every target padded to a multiple of 4 slots with repeated edges and packed whole into a stream) stages its
 * DISTINCT sources (<= 176) in LDS once per 32-channel tile (LDS-DMA, double-buffered) and the accumulators are initialised from
 * there.  In grid-cell order a window names each source ~5 times, so the L2 -> CU traffic falls from E rows to ~E / 5.  Windows
 * are pipelined across each other (r05): the next window's descriptors and first row stage are requested while this one runs.
 * Targets with more than 64 padded slots and targets a full window leaves over go through a per-target kernel (none on
 * r = 1 m graphs, a few dozen per k = 20 batch).
 *   plan: [dev] int32 [rgnn_mpnn_win_plan_ints(n, n_edges)], 16-byte aligned, filled by rgnn_mpnn_win_plan from the CSR by target
 *         (rowptr_t / src_sorted / node_order as for rgnn_mpnn_aggregate); once per graph, shared by all layers; holds the ticket
 *         counters and a weight scratch: ONE LAUNCH AT A TIME PER PLAN (two streams need two plans).
 * Returns RGNN_ERR_UNSUPPORTED for de > 8, d > 2048, n >= 2^24, Q rows not 16-byte aligned (ldq % 4, base address), rows of
 * 2^24 bytes or more, matrices or a plan of 2 GiB or more: the caller then takes rgnn_mpnn_aggregate_absmax
 * (radargnn_amd/gnn/mpnn_layers.py mirrors these limits).
 * flags / out_absmax as for rgnn_mpnn_aggregate_absmax.  Matches gnn/mpnn_layers.py:94-101 + torch-scatter max (hoisted form). */
int64_t rgnn_mpnn_win_plan_ints(int64_t n, int64_t n_edges);
/* diagnostics: the words of a built plan that hold the number of targets left to the per-target kernel / of windows made */
void rgnn_mpnn_win_plan_counters(int64_t n, int64_t n_edges, int64_t* leftover_word /*host*/, int64_t* windows_word /*host*/);
int rgnn_mpnn_win_plan(const int32_t* rowptr_t, const int32_t* src_sorted, const int32_t* node_order, int64_t n, int64_t n_edges,
                       int32_t* plan, rgnn_stream_t stream);
int rgnn_mpnn_aggregate_win(const float* p_bias, const float* Q, int64_t ldq, const float* We, int64_t ldwe,
                            const float* edge_attr_sorted, int32_t de, const int32_t* rowptr_t, const int32_t* src_sorted,
                            const int32_t* node_order, int32_t* plan, int64_t n, int64_t n_edges, int32_t d, float* out, int64_t ldo,
                            int32_t flags, float* out_absmax, rgnn_stream_t stream);
/* The kernel's operand image of one layer's weights -- per 32-channel tile the three bf16 terms of W_e [d, de <= 8] and the bias --
 * depends on the weights only.  rgnn_mpnn_aggregate_win builds it into the plan's scratch in front of every launch (5 us);
 * a caller that keeps it per weight version builds it once with rgnn_mpnn_win_wplanes ([dev] rgnn_mpnn_win_wplanes_bytes(d) bytes,
 * 16-byte aligned) and calls rgnn_mpnn_aggregate_win_planes with the SAME We / p_bias (the per-target kernel still reads them). */
int64_t rgnn_mpnn_win_wplanes_bytes(int32_t d);
int rgnn_mpnn_win_wplanes(const float* We, int64_t ldwe, int32_t de, int32_t d, const float* p_bias, void* planes, rgnn_stream_t stream);
int rgnn_mpnn_aggregate_win_planes(const float* p_bias, const float* Q, int64_t ldq, const float* We, int64_t ldwe,
                                   const float* edge_attr_sorted, int32_t de, const int32_t* rowptr_t, const int32_t* src_sorted,
                                   const int32_t* node_order, int32_t* plan, int64_t n, int64_t n_edges, int32_t d, float* out,
                                   int64_t ldo, int32_t flags, float* out_absmax, const void* wplanes, rgnn_stream_t stream);

/* Targets without incoming edges, in visiting order: list[0..count) = node ids (node_order[p] or p) of the empty CSR
 * segments; count is written to device memory (int64).  Deterministic (scan based).  flags_tmp: int32 [n],
 * scan_tmp: rgnn_scan_tmp_bytes(n) bytes, pos_tmp: int32 [n+1]. */
int rgnn_empty_targets(const int32_t* rowptr_t, const int32_t* node_order, int64_t n, int32_t* flags_tmp,
                       int32_t* pos_tmp, void* scan_tmp, int32_t* list, int64_t* count,
                       int32_t* slot_of_node /*[n] or NULL: position in `list`, -1 for targets with edges*/,
                       rgnn_stream_t stream);

/* Same, additionally the complement: list_nonempty[0..count_nonempty) = the targets WITH incoming edges, in visiting
 * order.  On a symmetric edge set (radius graphs) these are also the only nodes that occur as a source, so the dense
 * layers of a conv can be restricted to them (rgnn_linear_fwd row_index / m_dev) and the others take a cheaper path. */
int rgnn_split_targets(const int32_t* rowptr_t, const int32_t* node_order, int64_t n, int32_t* flags_tmp, int32_t* pos_tmp,
                       void* scan_tmp, int32_t* list, int64_t* count, int32_t* slot_of_node, int32_t* list_nonempty,
                       int64_t* count_nonempty, rgnn_stream_t stream);

/* The same two lists in ascending NODE order instead of visiting order (rank [dev] int32 [n]: CSR segment of node i is
 * rank[i]; NULL = identity).  What the row-subset launches of rgnn_linear_fwd want: they then stream the activation matrix
 * front to back (the grid-cell visiting order jumps around inside every frame; measured 7 % slower). */
int rgnn_split_targets_by_node(const int32_t* rowptr_t, const int32_t* rank, int64_t n, int32_t* flags_tmp, int32_t* pos_tmp,
                               void* scan_tmp, int32_t* list, int64_t* count, int32_t* slot_of_node,
                               int32_t* list_nonempty, int64_t* count_nonempty, rgnn_stream_t stream);

/* Work-balanced split of the CSR-by-target into chunks of ~80 units of (edges + 2 targets): chunk_start int32
 * [rgnn_mpnn_num_chunks(n, E) + 1 + 1024]: the table, followed by 1024 ints of ticket counters (persistent waves pull
 * chunks per XCD) that rgnn_mpnn_partition zeroes and every rgnn_mpnn_aggregate / rgnn_mpnn_edge_hidden launch leaves
 * zeroed again -- one launch at a time per table.  Computed once per graph, shared by all layers. */
int32_t rgnn_mpnn_num_chunks(int64_t n, int64_t n_edges);
/* The tuning values behind the chunk count: work units per chunk (80 on full batches, finer -- down to 16 -- on graphs
 * too small to fill the chip at 80) and the weight of one target in units
 * (num_chunks = ceil((E + weight * n) / units) + 1).  Exposed so callers and tests need not hard-code them. */
int32_t rgnn_mpnn_work_units(int64_t n, int64_t n_edges);
int32_t rgnn_mpnn_target_weight(int64_t n, int64_t n_edges);
int rgnn_mpnn_partition(const int32_t* rowptr_t, int64_t n, int64_t n_edges, int32_t* chunk_start, rgnn_stream_t stream);

/* General path (pre_layers > 1): first message layer per edge, hidden[e,:] = relu?(P[t_e] + Q[s_e] + W_e a_e),
 * rows in CSR-by-target order, then rgnn_linear_fwd on the [E,d] rows, then rgnn_segment_reduce. */
int rgnn_mpnn_edge_hidden(const float* P, int64_t ldp, const float* p_bias, const float* Q, int64_t ldq,
                          const float* We, int64_t ldwe, const float* edge_attr_sorted, int32_t de,
                          const int32_t* rowptr_t, const int32_t* src_sorted, const int32_t* node_order,
                          const int32_t* chunk_start, int32_t n_chunks, int64_t n, int32_t d, int32_t relu, float* hidden,
                          int64_t ldh, rgnn_stream_t stream);
int rgnn_segment_reduce(const float* rows, int64_t ldr, const int32_t* rowptr_t, const int32_t* node_order, int64_t n,
                        int32_t d, int32_t aggr, float* out, int64_t ldo, rgnn_stream_t stream);

/* out[i,:] = in[perm[i],:] for rows of `width` 4-byte elements (edge_attr -> CSR-by-target order). */
int rgnn_gather_rows_f32(const float* in, int64_t ldi, const int32_t* perm, int64_t n_rows, int32_t width, float* out,
                         int64_t ldo, rgnn_stream_t stream);

/* Row softmax of the class logits (postprocessor/inference.py:46,62). */
int rgnn_softmax_rows(const float* x, int64_t ldx, int64_t m, int32_t n, float* y, int64_t ldy, rgnn_stream_t stream);

/* ================================================================ batches of graphs resident in HBM
 * Device-side replacement of torch_geometric's DataLoader collation (utils/data_handling.py:30 -> Batch.from_data_list)
 * and of the per-batch host-to-device copy (postprocessor/inference.py:57): the processed dataset lives in HBM as
 * concatenated tensors, a batch is a list of n_seg graph ids.  All offset tables are int64 on the device:
 *   seg_src_row[s]  first row (node or edge) of selected graph s in the resident tensor,
 *   seg_dst_ptr[s]  first row of graph s in the batch, seg_dst_ptr[n_seg] = n_rows (non-decreasing; empty graphs allowed).
 * rgnn_collate_rows copies rows of `width` 4-byte words (x, edge_attr, y, pos, vel) and, when batch_out != NULL, writes
 * PyG's `batch` vector (graph slot of every row).  rgnn_collate_edges copies both rows of the int64 edge_index [2, E]
 * (row stride ld) and adds seg_node_shift[s] = first batch row of graph s (PyG: `edge_index += cumulative num_nodes`). */
int rgnn_collate_rows(const void* src, int64_t ld_src, int32_t width, const int64_t* seg_src_row, const int64_t* seg_dst_ptr,
                      int32_t n_seg, int64_t n_rows, void* out, int64_t ld_out, int64_t* batch_out, rgnn_stream_t stream);
int rgnn_collate_edges(const int64_t* src_edge_index, int64_t ld_src, const int64_t* seg_src_edge,
                       const int64_t* seg_dst_eptr, const int64_t* seg_node_shift, int32_t n_seg, int64_t n_edges,
                       int64_t* out, int64_t ld_out, rgnn_stream_t stream);

/* Host side of a STREAMED batch (no device work, no stream; the reference collates on the host: utils/data_handling.py:30 DataLoader
 * -> Batch.from_data_list, then `data.to(device)`, postprocessor/inference.py:57).  The float64 point arrays of n_frames frames, wherever
 * they lie in host memory, are laid back to back into ONE block, so that a batch goes up in one copy:
 *   block = X [n, 2] | V [n, 2] | rcs [n] | timestamp [n] | frame_ptr [n_frames + 1] (int64),   n = sum n_points,
 *   byte offsets 0, 16 n, 32 n, 40 n, 48 n;  block_bytes >= 48 n + 8 (n_frames + 1)  (RGNN_INVALID_ARGUMENT otherwise).
 * addr: [n_frames][4] host pointers {X, V, rcs, timestamp} of contiguous float64 arrays ([n_f, 2], [n_f, 2], [n_f], [n_f]).
 * Meant to be called WITHOUT the interpreter lock from a loader thread (ctypes releases it). */
int rgnn_stage_frames(int64_t n_frames, const void* const* addr, const int64_t* n_points, void* block, int64_t block_bytes);

/* ================================================================ post-processor front half
 * Per node of one graph / batch: predicted label (first index of the row maximum of class_prob [n, n_classes]), its
 * score, the keep flag of PredictionExtractor.get_absolute_object_bounding_box_predictions
 * (postprocessor/postprocessing.py:198-319: removed when class_prob[:, bg_index] >= max_score_for_background, when the
 * label is bg_index, or when score <= min_object_score[label] for label < n_min_scores) and the four corners
 * (c1x, c1y, ..., c4x, c4y; float64) of the absolute box decoded from boxes [n, 4] (relative aligned) or [n, 5]
 * (invariance 0: absolute rotated, 1: relative rotated, 2: E(n)-invariant, which needs nn_index = nearest other point of
 * every node, rgnn_knn_graph with k = 1); box algebra of preprocessor/bounding_box.py.  pos: float32 [n, 2] contiguous.
 * adapt_orientation_angle: the angle was trained as sin(theta shifted to [-pi/2, pi/2]) (bounding_box.py:566-589). */
int rgnn_decode_predictions(const float* class_prob, int64_t ldp, int32_t n_classes, const float* boxes, int64_t ldb,
                            int32_t box_width, const float* pos, const int32_t* nn_index, int64_t n, int32_t bg_index,
                            float max_score_for_background, const double* min_object_score, int32_t n_min_scores,
                            int32_t invariance, int32_t adapt_orientation_angle, int32_t* label, float* score,
                            int32_t* keep, double* corners, rgnn_stream_t stream);

/* Non-maximum suppression of one graph's boxes (BoxSuppressor.apply_nms, postprocessor/postprocessing.py:336-431).
 * kind 0: aligned boxes [x_min, y_min, x_max, y_max] float32 [m, 4], torchvision.ops.nms semantics (suppress IoU > t);
 * kind 1: rotated boxes [x, y, l, w, theta in degrees] float64 [m, 5], detectron2 nms_rotated semantics (IoU >= t).
 * order: int64 [m] box ids by descending score (stable); mask_tmp: rgnn_nms_mask_words(m) 64-bit words of workspace;
 * keep: int64 [m] receives the ids kept, by descending score; count: their number (device). */
/* Box ids by descending score, ties by ascending id (a stable descending sort; NaN first): the order rgnn_nms wants -- what
 * torchvision.ops.nms / detectron2 nms_rotated compute with a library sort before their suppression pass
 * (postprocessor/postprocessing.py:336-435).  scores: [dev] float32 or float64 [m]; order: [dev] int64 [m];
 * tmp: [dev] rgnn_sort_scores_tmp_bytes(m) bytes. */
int64_t rgnn_sort_scores_tmp_bytes(int64_t m);
int rgnn_sort_scores(const void* scores, int32_t is_f64, int64_t m, int64_t* order, void* tmp, rgnn_stream_t stream);
int64_t rgnn_nms_mask_words(int64_t m);
/* corners float64 [m, 4, 2] -> two_point float64 [m, 4] ([x_min, y_min, x_max, y_max]) and / or rotated float64 [m, 5]
 * ([x, y, l, w, theta in degrees, 0..180]); BoundingBox.get_two_point_representations /
 * get_absolute_rotated_box_representations, preprocessor/bounding_box.py:447-540.  Either output may be NULL. */
int rgnn_box_representations(const double* corners, int64_t m, double* two_point, double* rotated, rgnn_stream_t stream);
int rgnn_nms(const void* boxes, int32_t kind, const int64_t* order, int64_t m, double iou_threshold, uint64_t* mask_tmp,
             int64_t* keep, int64_t* count, rgnn_stream_t stream);

/* ================================================================ backward pass (training: gnn/trainer.py:176-231)
 * What autograd derives for the reference's op-by-op forward, for the fused forward kernels above.  The dense-layer
 * gradients are GEMMs: dX = dY W runs on rgnn_linear_fwd with the transposed weight, dW = dY^T X on rgnn_wgrad
 * (csrc/wgrad.hip: split-K MFMA kernel of this library; no BLAS anywhere in the product since r02). */

/* ReLU fused into a dense-layer epilogue: dx = (y > 0) ? dy : 0 over `count` contiguous floats (16-byte aligned). */
int rgnn_relu_bwd(const float* dy, const float* y, float* dx, int64_t count, rgnn_stream_t stream);

/* Train-mode BatchNorm1d backward, reduction half: per 128-row panel the column sums of g and g*h, where
 * g = dy (y == NULL) or (y > 0 ? dy : 0) (BatchNorm followed by the fused ReLU, y = its output) and h = the
 * BatchNorm input.  partial: float [panels, 2, n], panels = rgnn_linear_stat_panels(m). */
int rgnn_bn_bwd_stats(const float* dy, int64_t lddy, const float* y, int64_t ldy, const float* h, int64_t ldh, int64_t m,
                      int32_t n, float* partial, rgnn_stream_t stream);

/* ... elementwise half: dx = A[c] g + B[c] h + C[c], coef = [A | B | C] (3 n floats) with
 * A = gamma rstd, B = -gamma rstd^2 S/m, C = -gamma rstd sum(g)/m + gamma rstd^2 mean S/m, S = sum g xhat. */
int rgnn_bn_bwd_apply(const float* dy, int64_t lddy, const float* y, int64_t ldy, const float* h, int64_t ldh,
                      const float* coef, int64_t m, int32_t n, float* dx, int64_t lddx, rgnn_stream_t stream);
/* ... also raising max |dx| into dx_absmax (a bound, RGNN_BOUND_SLOTS words zeroed by the caller; NULL: rgnn_bn_bwd_apply): the dgrad
 * launches that read dx then take the f16x2 form. */
/* Both halves with the ReLU mask recomputed from h and the forward pass's apply table instead of read from y: table =
 * [RGNN_AFFINE_ROWS, n] of rgnn_batchnorm_finalize*, mask = fmaf(h - mean_hi, g, t) > 0 -- the bits of y as rgnn_scale_shift_act
 * (or the A-operand path of rgnn_linear_fwd) produced them; a quarter to a third fewer bytes per pass.  table NULL: y as above. */
int rgnn_bn_bwd_stats_table(const float* dy, int64_t lddy, const float* y, int64_t ldy, const float* table, const float* h, int64_t ldh,
                            int64_t m, int32_t n, float* partial, rgnn_stream_t stream);
int rgnn_bn_bwd_apply_table(const float* dy, int64_t lddy, const float* y, int64_t ldy, const float* table, const float* h, int64_t ldh,
                            const float* coef, int64_t m, int32_t n, float* dx, int64_t lddx, float* dx_absmax, rgnn_stream_t stream);
int rgnn_bn_bwd_apply_absmax(const float* dy, int64_t lddy, const float* y, int64_t ldy, const float* h, int64_t ldh,
                             const float* coef, int64_t m, int32_t n, float* dx, int64_t lddx, float* dx_absmax, rgnn_stream_t stream);
/* The [3, n] coefficients of rgnn_bn_bwd_apply plus d gamma / d beta in one launch (float64 inside): from the forward column
 * statistics of the layer input (fwd_stats [panels_f, RGNN_STAT_ROWS, n], use_batch = 1) or the running statistics (use_batch = 0) and the
 * partial sums of rgnn_bn_bwd_stats (bwd_part [panels_b, 2, n]).  dgamma / dbeta may be NULL. */
int rgnn_bn_bwd_coef(const float* fwd_stats, int64_t panels_f, const float* running_mean, const float* running_var,
                     const float* bwd_part, int64_t panels_b, int64_t m, int32_t n, const float* gamma, float eps,
                     int32_t use_batch, float* coef, float* dgamma, float* dbeta, rgnn_stream_t stream);

/* Backward of rgnn_segment_reduce: d_rows [E, d] (every row is written; rows of the forward input are needed for the
 * arg-max: the first row of a segment attaining the maximum receives dM[t, c]; mean / add: every row, / deg). */
int rgnn_segment_reduce_bwd(const float* dM, int64_t lddm, const float* rows, int64_t ldr, const int32_t* rowptr_t,
                            const int32_t* node_order, int64_t n, int32_t d, int32_t aggr, float* d_rows, int64_t lddr,
                            rgnn_stream_t stream);

/* The same backward for max aggregation on the shipped shapes (d % 8 == 0, d <= 512, 1 <= de <= 8, in-degrees < 65 536), with
 * the edge half as two kernels that keep their reductions inside a lane (dW_e: lanes = channels; d_edge_attr: lanes = edges
 * scanning their target's arg row).  Per sorted edge e: tgt_sorted[e] its target node, eloc_sorted[e] its index inside the
 * target's segment; per out-edge j of the source CSR: tloc[j] = eloc_sorted[tpos[j]].  arg: uint16 [n, d], the index (inside
 * the segment) of the edge that won channel c of target t -- recorded by the forward pass (arg_is_valid = 1) or recomputed
 * here from Q / W_e / the attributes (0).  dwe_partial: float [rgnn_mpnn_bwd_slots(n), d, de]. */
int32_t rgnn_mpnn_max_bwd_supported(int32_t d, int32_t de);
int rgnn_mpnn_max_bwd(const float* dM, int64_t lddm, const float* Q, int64_t ldq, const float* We, int64_t ldwe,
                      const float* edge_attr_sorted, int32_t de, const int32_t* rowptr_t, const int32_t* src_sorted,
                      const int32_t* tgt_sorted, const int32_t* eloc_sorted, const int32_t* node_order, int64_t n, int32_t d,
                      const int32_t* rowptr_s, const int32_t* tnode, const int32_t* tloc, int64_t n_edges, uint16_t* arg,
                      int32_t arg_is_valid, float* dwe_partial, float* dQ, int64_t lddq, float* d_edge_attr, float* dWe,
                      rgnn_stream_t stream);
/* ... also raising max |dQ| into dq_absmax (a bound; NULL: rgnn_mpnn_max_bwd). */
int rgnn_mpnn_max_bwd_absmax(const float* dM, int64_t lddm, const float* Q, int64_t ldq, const float* We, int64_t ldwe,
                             const float* edge_attr_sorted, int32_t de, const int32_t* rowptr_t, const int32_t* src_sorted,
                             const int32_t* tgt_sorted, const int32_t* eloc_sorted, const int32_t* node_order, int64_t n, int32_t d,
                             const int32_t* rowptr_s, const int32_t* tnode, const int32_t* tloc, int64_t n_edges, uint16_t* arg,
                             int32_t arg_is_valid, float* dwe_partial, float* dQ, int64_t lddq, float* d_edge_attr, float* dWe,
                             float* dq_absmax, rgnn_stream_t stream);

/* Weight gradient of a dense layer: dW[n, k] = sum_m G[m, n] * [A1 | A2][m, k] (G = gradient of the layer output after
 * the activation mask, A = the layer input), fp32 MFMA, reduction over the rows split into
 * rgnn_linear_wgrad_slabs(m, n, k1 + k2) slabs whose partial tiles go to `partial` (float [slabs, n, k1 + k2]) and are
 * summed into dW [n, k1 + k2] (row-major, contiguous) by a second kernel: no atomics, deterministic.
 * Widths and row strides must be multiples of 4 floats, pointers 16-byte aligned. */
int32_t rgnn_linear_wgrad_slabs(int64_t m, int32_t n, int32_t k);
int rgnn_linear_wgrad(const float* G, int64_t ldg, const float* A1, int64_t lda1, int32_t k1, const float* A2, int64_t lda2,
                      int32_t k2, int64_t m, int32_t n, float* partial, float* dW, rgnn_stream_t stream);

/* The same gradient on the bf16 matrix pipe (three-term split of both operands, six MFMA products per fp32 product, fp32
 * accumulate: the forward kernels' arithmetic), for any widths and row strides:
 *   dW[n, k] = sum_r G[row(r), n] * [A1 | A2 | 1][row(r), k],   row(r) = row_index ? row_index[r] : r,  r < (m_dev ? *m_dev : m)
 * with_ones appends a column of ones to the input, so dW[:, k1 + k2] is the bias gradient (column sums of G over the same
 * rows).  partial: float [rgnn_wgrad_slabs(m, n, k1, k2, with_ones), n, k1 + k2 + with_ones]; dW [n, k1 + k2 + with_ones]
 * row-major.  Deterministic (slab partials summed by a second kernel, no atomics).  m = 0: dW is zeroed (the sum over nothing;
 * G / A may be NULL then). */
int32_t rgnn_wgrad_slabs(int64_t m, int32_t n, int32_t k1, int32_t k2, int32_t with_ones);
int rgnn_wgrad(const float* G, int64_t ldg, int32_t n, const float* A1, int64_t lda1, int32_t k1, const float* A2, int64_t lda2,
               int32_t k2, int32_t with_ones, int64_t m, const int32_t* row_index /*[dev] or NULL*/,
               const int64_t* m_dev /*[dev] or NULL*/, float* partial, float* dW, rgnn_stream_t stream);
/* ... in the f16x2 form when every operand block comes with a bound (g_bound, a1_bound if k1 > 0, a2_bound if k2 > 0: arrays of
 * RGNN_BOUND_SLOTS floats bounding |G| / |A1| / |A2| over the rows read): both operands pre-scaled by exact powers of two, split into
 * two f16 terms, three MFMA products (l h', h l', h h') -- the forward kernels' f16x2 arithmetic; the column of ones is carried as
 * 2^14 after the pre-scale.  Any bound NULL: the bf16x3 form of rgnn_wgrad. */
int rgnn_wgrad_bounds(const float* G, int64_t ldg, int32_t n, const float* A1, int64_t lda1, int32_t k1, const float* A2, int64_t lda2,
                      int32_t k2, int32_t with_ones, int64_t m, const int32_t* row_index /*[dev] or NULL*/,
                      const int64_t* m_dev /*[dev] or NULL*/, const float* g_bound, const float* a1_bound, const float* a2_bound,
                      float* partial, float* dW, rgnn_stream_t stream);

/* Backward of rgnn_mpnn_aggregate without its target term: M[t] = aggr_{e -> t}(Q[s_e] + W_e a_e) (0 for empty
 * segments).  max: the gradient of (t, c) goes to the first edge attaining the maximum (torch-scatter arg_out).
 * Two kernels, no atomics on the node gradient:
 *   edge half  -- over the CSR by target (same rowptr_t / src_sorted / node_order as the forward): d_edge_attr [E, de]
 *                 and dWe [d, de] (both written), and for max the winning edge
 *                 position per (t, c) in arg_tmp int32 [n, d];
 *   node half  -- over the CSR by SOURCE of the same edges (rowptr_s [n+1], segments in the same node_order;
 *                 tnode[j] = target of out-edge j, tpos[j] = position of that edge in the target-sorted list):
 *                 dQ [n, d] is written for every node.
 * target_scale: float [n], 1 / in-degree (mean only, else NULL).  dwe_partial: float [rgnn_mpnn_bwd_slots(n), d, de]
 * workspace (every persistent wave group stores its share of dW_e, a last kernel sums them into dWe: no atomics). */
int64_t rgnn_mpnn_bwd_slots(int64_t n);
/* waves per segment of the edge half (1, 2 or 4); dea_partial: float [split, E, de] workspace when split > 1 */
int32_t rgnn_mpnn_bwd_split(int32_t d);
int rgnn_mpnn_aggregate_bwd(const float* dM, int64_t lddm, const float* Q, int64_t ldq, const float* We, int64_t ldwe,
                            const float* edge_attr_sorted, int32_t de, const int32_t* rowptr_t, const int32_t* src_sorted,
                            const int32_t* node_order, int64_t n, int32_t d, int32_t aggr, const int32_t* rowptr_s,
                            const int32_t* tnode, const int32_t* tpos, const float* target_scale, int64_t n_edges,
                            int32_t* arg_tmp, float* dwe_partial, float* dea_partial, float* dQ, int64_t lddq,
                            float* d_edge_attr, float* dWe, rgnn_stream_t stream);

/* ================================================================ detection loss (training: gnn/trainer.py:181-222)
 * loss = cls_loss_weight * CrossEntropyLoss(weight = class_weight)(cls, label) + bb_loss_weight * mean over the nodes
 * with label != bg_index of HuberLoss(delta)(y[:, 1:], boxes) -- the reference's per-node Python loop (trainer.py:193-201)
 * as one pass.  y: float32 [n, 1 + box_width] (label | box), class_weight: float32 [n_classes] or NULL.
 * partial_tmp: 4 * rgnn_detection_loss_blocks(n) doubles; sums: 4 doubles (sum w nll, sum w, sum huber, number of object
 * nodes; kept for the backward pass); loss_out: 3 floats (loss, loss_cls, loss_bb).  A batch without objects or with a NaN
 * box term contributes loss_bb = 0 (trainer.py:203-217).  _bwd writes d loss / d cls and d loss / d boxes scaled by
 * grad_loss[0] (NULL: 1). */
int64_t rgnn_detection_loss_blocks(int64_t n);
int rgnn_detection_loss(const float* cls, int64_t ldc, int32_t n_classes, const float* boxes, int64_t ldb, int32_t box_width,
                        const float* y, int64_t ldy, const float* class_weight, int64_t n, int32_t bg_index, float delta,
                        float cls_loss_weight, float bb_loss_weight, double* partial_tmp, double* sums, float* loss_out,
                        rgnn_stream_t stream);
int rgnn_detection_loss_bwd(const float* cls, int64_t ldc, int32_t n_classes, const float* boxes, int64_t ldb,
                            int32_t box_width, const float* y, int64_t ldy, const float* class_weight, int64_t n,
                            int32_t bg_index, float delta, float cls_loss_weight, float bb_loss_weight, const double* sums,
                            const float* grad_loss, float* d_cls, int64_t lddc, float* d_boxes, int64_t lddb,
                            rgnn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* RGNN_H */
