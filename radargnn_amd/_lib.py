"""ctypes binding of librgnn.so (include/rgnn.h).  There is NO fallback: if the HIP library is missing or does
not export a declared symbol, importing this module fails loudly."""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  -- must be imported first: librgnn.so binds to the HIP runtime torch has already loaded

_HERE = os.path.dirname(os.path.abspath(__file__))
# RGNN_LIB: another build of the SAME library (tools: A/B of kernel variants in one gpurun call); never a different implementation
LIB_PATH = os.environ.get("RGNN_LIB") or os.path.join(_HERE, "librgnn.so")

c_i32, c_i64, c_f32, c_f64, c_vp = C.c_int32, C.c_int64, C.c_float, C.c_double, C.c_void_p


class RgnnGrid(C.Structure):
    _fields_ = [("X", c_vp), ("dim", c_i32), ("n", c_i64), ("frame_ptr", c_vp), ("n_frames", c_i64),
                ("ws", c_vp), ("ws_bytes", c_i64)]


class RgnnLinearArgs(C.Structure):
    _fields_ = [("A1", c_vp), ("lda1", c_i64), ("k1", c_i32),
                ("A2", c_vp), ("lda2", c_i64), ("k2", c_i32),
                ("W1", c_vp), ("W2", c_vp), ("ldw", c_i64), ("w_split", c_i32),
                ("bias1", c_vp), ("bias2", c_vp),
                ("residual", c_vp), ("ldr", c_i64),
                ("out", c_vp), ("ldo", c_i64),
                ("m", c_i64), ("n", c_i32),
                ("relu_out", c_i32),
                ("col_stats", c_vp),
                ("row_index", c_vp), ("m_dev", c_vp), ("accumulate", c_i32), ("gather_only", c_i32),
                ("residual_index", c_vp),
                ("W_planes", c_vp), ("w_planes_kp", c_i32),
                ("splitk_ws", c_vp), ("splitk_ws_bytes", c_i64),
                ("a1_scale_shift", c_vp), ("a1_relu", c_i32), ("relu_from_col", c_i32),
                ("W_planes_f16", c_vp), ("a1_bound", c_vp), ("a2_bound", c_vp), ("out_absmax", c_vp), ("a1_panel_segment", c_vp)]


# name -> (restype, argtypes); one entry per function declared in include/rgnn.h
SIGNATURES = {
    "rgnn_version": (C.c_char_p, []),
    "rgnn_last_error": (C.c_char_p, []),
    "rgnn_env_reload": (None, []),
    "rgnn_profile_next_launch": (None, [c_vp, c_vp]),
    "rgnn_stream_create": (c_i32, [C.POINTER(c_vp)]),
    "rgnn_stream_destroy": (c_i32, [c_vp]),
    "rgnn_scan_tmp_bytes": (c_i64, [c_i64]),
    "rgnn_exclusive_scan_i32": (c_i32, [c_vp, c_vp, c_i64, c_vp, c_vp]),
    "rgnn_grid_workspace_bytes": (c_i64, [c_i64, c_i64, c_i32]),
    "rgnn_grid_build": (c_i32, [C.POINTER(RgnnGrid), c_f64, c_f64, c_vp]),
    "rgnn_grid_build_frames": (c_i32, [C.POINTER(RgnnGrid), c_f64, c_f64, c_i64, c_vp]),
    "rgnn_grid_order_offsets": (c_i32, [c_i64, c_i64, c_i32, C.POINTER(c_i64), C.POINTER(c_i64)]),
    "rgnn_radius_graph_count": (c_i32, [C.POINTER(RgnnGrid), c_f64, c_vp, c_vp]),
    "rgnn_radius_graph_fill": (c_i32, [C.POINTER(RgnnGrid), c_f64, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp]),
    "rgnn_radius_graph_fill_checked": (c_i32, [C.POINTER(RgnnGrid), c_f64, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp]),
    "rgnn_radius_graph_rows": (c_i32, [C.POINTER(RgnnGrid), c_f64, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_i32, c_vp]),
    "rgnn_radius_graph_rows_direct": (c_i32, [C.POINTER(RgnnGrid), c_f64, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_i32, c_vp]),
    "rgnn_radius_rows_commit": (c_i32, [c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "rgnn_knn_graph": (c_i32, [C.POINTER(RgnnGrid), c_i32, c_vp, c_vp, c_vp, c_vp]),
    "rgnn_knn_graph_attrs": (c_i32, [C.POINTER(RgnnGrid), c_i32, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp]),
    "rgnn_knn_graph_frames": (c_i32, [C.POINTER(RgnnGrid), c_i32, c_i64, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp]),
    "rgnn_knn_degree_from_csr": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_vp, c_vp]),
    "rgnn_undirected_degree_preset": (c_i32, [c_vp, c_vp, c_i64, c_vp, c_vp]),
    "rgnn_grid_cell_order": (c_i32, [C.POINTER(RgnnGrid), c_vp, c_vp]),
    "rgnn_undirected_degree": (c_i32, [c_vp, c_vp, c_i64, c_vp, c_vp, c_vp]),
    "rgnn_csr_by_target_tmp_bytes": (c_i64, [c_i64, c_i64]),
    "rgnn_invert_permutation": (c_i32, [c_vp, c_i64, c_vp, c_vp]),
    "rgnn_source_rowptr": (c_i32, [c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp]),
    "rgnn_csr_by_target": (c_i32, [c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "rgnn_csr_by_target_unordered": (c_i32, [c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "rgnn_csr_by_target_frames": (c_i32, [c_vp, c_i64, c_i64, c_i64, c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "rgnn_csr_by_target_symmetric": (c_i32, [c_vp, c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "rgnn_embed3_supported": (c_i32, [c_i32, c_i32, c_i32, c_i32]),
    "rgnn_embed3": (c_i32, [c_vp, c_i64, c_i32, c_vp, c_i64, c_vp, c_i32, c_vp, c_vp, c_i32, c_vp, c_vp, c_i32, c_i32, c_i64, c_vp, c_i64,
                            c_vp, c_vp]),
    "rgnn_csr_by_target_symmetric_own": (c_i32, [c_vp, c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "rgnn_edge_features": (c_i32, [c_vp, c_vp, c_vp, c_i64, C.POINTER(c_i32), c_i32, c_i32, c_vp, c_i32, c_vp, c_vp]),
    "rgnn_edge_features_reversed": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, C.POINTER(c_i32), c_i32, c_i32, c_vp, c_i32, c_vp, c_vp]),
    "rgnn_node_features": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, C.POINTER(c_i32), c_i32, c_vp, c_i32, c_vp]),
    "rgnn_node_features_time_index": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, C.POINTER(c_i32), c_i32, c_vp, c_i32,
                                              c_vp, c_vp, c_vp]),
    "rgnn_split_by_degree_frames": (c_i32, [c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "rgnn_radius_counts": (c_i32, [c_vp, c_i64, c_vp, c_i32, c_vp, c_vp]),
    "rgnn_time_index": (c_i32, [c_vp, c_vp, c_i64, c_vp, c_vp, c_vp]),
    "rgnn_time_index_ws_bytes": (c_i64, [c_i64]),
    "rgnn_time_index_ws": (c_i32, [c_vp, c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "rgnn_linear_stat_panels": (c_i64, [c_i64]),
    "rgnn_linear_fwd": (c_i32, [C.POINTER(RgnnLinearArgs), c_vp]),
    "rgnn_linear_fwd_fuses_a1_affine": (c_i32, [C.POINTER(RgnnLinearArgs)]),
    "rgnn_linear_planes_kp": (c_i32, [c_i32]),
    "rgnn_linear_fwd_path": (c_i32, [C.POINTER(RgnnLinearArgs)]),
    "rgnn_linear_planes_f16_bytes": (c_i64, [c_i32, c_i32]),
    "rgnn_linear_split_weights_f16": (c_i32, [c_vp, c_vp, c_i64, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "rgnn_linear_splitk_ws_bytes": (c_i64, []),
    "rgnn_linear_split_weights": (c_i32, [c_vp, c_vp, c_i64, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "rgnn_tiny_mlp2": (c_i32, [c_vp, c_i64, c_i32, c_vp, c_i64, c_vp, c_i64, c_vp, c_i32, c_i32, c_vp, c_i64, c_vp, c_i32, c_i32, c_vp,
                       c_i64, c_vp]),
    "rgnn_batchnorm_finalize": (c_i32, [c_vp, c_i64, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_f32, c_f32,
                                        c_vp, c_vp]),
    "rgnn_batchnorm_finalize_parts": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32,
                                              c_f32, c_f32, c_vp, c_vp]),
    "rgnn_batchnorm_finalize_bound": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32,
                                              c_f32, c_f32, c_vp, c_vp, c_vp, c_vp]),
    "rgnn_batchnorm_segments": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_f32, c_f32, c_vp, c_vp,
                                        c_vp, c_vp, c_vp]),
    "rgnn_batchnorm_act_segments": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_f32, c_f32, c_i32, c_vp,
                                            c_vp, c_vp, c_i64, c_vp, c_vp, c_vp]),
    "rgnn_batchnorm_segments_from_panels": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_f32,
                                                    c_f32, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "rgnn_pad_list_by_segment": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "rgnn_pad_list_pair_by_segment": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "rgnn_scale_shift_act_segments": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_i64, c_i64, c_i32, c_i32, c_vp, c_i64, c_vp]),
    "rgnn_column_stats": (c_i32, [c_vp, c_i64, c_i64, c_i32, c_vp, c_vp]),
    "rgnn_scale_shift_act": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_i32, c_i32, c_vp, c_i64, c_vp]),
    "rgnn_mpnn_aggregate": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_i32,
                                    c_i64, c_i32, c_i32, c_vp, c_i64, c_vp]),
    "rgnn_mpnn_aggregate_flags": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_i32,
                                          c_i64, c_i32, c_i32, c_vp, c_i64, c_i32, c_vp]),
    "rgnn_mpnn_aggregate_absmax": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_i32,
                                           c_i64, c_i32, c_i32, c_vp, c_i64, c_i32, c_vp, c_vp]),
    "rgnn_mpnn_aggregate_max_arg": (c_i32, [c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_i32, c_i64, c_i32,
                                             c_vp, c_i64, c_vp, c_i32, C.POINTER(c_i32), c_vp]),
    "rgnn_mpnn_aggregate_max_arg_absmax": (c_i32, [c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_i32, c_i64, c_i32,
                                                    c_vp, c_i64, c_vp, c_i32, C.POINTER(c_i32), c_vp, c_vp]),
    "rgnn_mpnn_win_plan_ints": (c_i64, [c_i64, c_i64]),
    "rgnn_mpnn_win_plan_counters": (None, [c_i64, c_i64, C.POINTER(c_i64), C.POINTER(c_i64)]),
    "rgnn_mpnn_win_plan": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i64, c_vp, c_vp]),
    "rgnn_mpnn_aggregate_win": (c_i32, [c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i32, c_vp,
                                        c_i64, c_i32, c_vp, c_vp]),
    "rgnn_mpnn_win_wplanes_bytes": (c_i64, [c_i32]),
    "rgnn_mpnn_win_wplanes": (c_i32, [c_vp, c_i64, c_i32, c_i32, c_vp, c_vp, c_vp]),
    "rgnn_mpnn_aggregate_win_planes": (c_i32, [c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i32, c_vp,
                                               c_i64, c_i32, c_vp, c_vp, c_vp]),
    "rgnn_empty_targets": (c_i32, [c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "rgnn_split_targets": (c_i32, [c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "rgnn_split_targets_by_node": (c_i32, [c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "rgnn_mpnn_num_chunks": (c_i32, [c_i64, c_i64]),
    "rgnn_mpnn_work_units": (c_i32, [c_i64, c_i64]),
    "rgnn_mpnn_target_weight": (c_i32, [c_i64, c_i64]),
    "rgnn_mpnn_partition": (c_i32, [c_vp, c_i64, c_i64, c_vp, c_vp]),
    "rgnn_mpnn_edge_hidden": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_i32,
                                      c_i64, c_i32, c_i32, c_vp, c_i64, c_vp]),
    "rgnn_segment_reduce": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_i64, c_i32, c_i32, c_vp, c_i64, c_vp]),
    "rgnn_gather_rows_f32": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_i32, c_vp, c_i64, c_vp]),
    "rgnn_softmax_rows": (c_i32, [c_vp, c_i64, c_i64, c_i32, c_vp, c_i64, c_vp]),
    "rgnn_relu_bwd": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_vp]),
    "rgnn_bn_bwd_stats": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_i64, c_i32, c_vp, c_vp]),
    "rgnn_bn_bwd_apply": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_i32, c_vp, c_i64, c_vp]),
    "rgnn_bn_bwd_stats_table": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_i64, c_i64, c_i32, c_vp, c_vp]),
    "rgnn_bn_bwd_apply_table": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_i64, c_i32, c_vp, c_i64, c_vp, c_vp]),
    "rgnn_bn_bwd_apply_absmax": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_i32, c_vp, c_i64, c_vp, c_vp]),
    "rgnn_bn_bwd_coef": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_i64, c_i32, c_vp, c_f32, c_i32, c_vp, c_vp, c_vp, c_vp]),
    "rgnn_mpnn_aggregate_bwd": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_i32, c_vp, c_vp, c_vp, c_i64, c_i32,
                                        c_i32, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp]),
    "rgnn_mpnn_max_bwd_supported": (c_i32, [c_i32, c_i32]),
    "rgnn_mpnn_max_bwd": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_vp,
                                   c_vp, c_vp, c_i64, c_vp, c_i32, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp]),
    "rgnn_mpnn_max_bwd_absmax": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_vp,
                                          c_vp, c_vp, c_i64, c_vp, c_i32, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp]),
    "rgnn_mpnn_bwd_split": (c_i32, [c_i32]),
    "rgnn_segment_reduce_bwd": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_i64, c_i32, c_i32, c_vp, c_i64, c_vp]),
    "rgnn_linear_wgrad_slabs": (c_i32, [c_i64, c_i32, c_i32]),
    "rgnn_linear_wgrad": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_i32, c_vp, c_i64, c_i32, c_i64, c_i32, c_vp, c_vp, c_vp]),
    "rgnn_wgrad_slabs": (c_i32, [c_i64, c_i32, c_i32, c_i32, c_i32]),
    "rgnn_wgrad": (c_i32, [c_vp, c_i64, c_i32, c_vp, c_i64, c_i32, c_vp, c_i64, c_i32, c_i32, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "rgnn_wgrad_bounds": (c_i32, [c_vp, c_i64, c_i32, c_vp, c_i64, c_i32, c_vp, c_i64, c_i32, c_i32, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                   c_vp, c_vp]),
    "rgnn_mpnn_bwd_slots": (c_i64, [c_i64]),
    "rgnn_decode_predictions": (c_i32, [c_vp, c_i64, c_i32, c_vp, c_i64, c_i32, c_vp, c_vp, c_i64, c_i32, C.c_float, c_vp, c_i32,
                                        c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "rgnn_box_representations": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_vp]),
    "rgnn_sort_scores_tmp_bytes": (c_i64, [c_i64]),
    "rgnn_sort_scores": (c_i32, [c_vp, c_i32, c_i64, c_vp, c_vp, c_vp]),
    "rgnn_nms_mask_words": (c_i64, [c_i64]),
    "rgnn_nms": (c_i32, [c_vp, c_i32, c_vp, c_i64, c_f64, c_vp, c_vp, c_vp, c_vp]),
    "rgnn_detection_loss_blocks": (c_i64, [c_i64]),
    "rgnn_detection_loss": (c_i32, [c_vp, c_i64, c_i32, c_vp, c_i64, c_i32, c_vp, c_i64, c_vp, c_i64, c_i32, c_f32, c_f32, c_f32,
                                    c_vp, c_vp, c_vp, c_vp]),
    "rgnn_detection_loss_bwd": (c_i32, [c_vp, c_i64, c_i32, c_vp, c_i64, c_i32, c_vp, c_i64, c_vp, c_i64, c_i32, c_f32, c_f32,
                                        c_f32, c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp]),
    "rgnn_collate_rows": (c_i32, [c_vp, c_i64, c_i32, c_vp, c_vp, c_i32, c_i64, c_vp, c_i64, c_vp, c_vp]),
    "rgnn_collate_edges": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_vp, c_i32, c_i64, c_vp, c_i64, c_vp]),
    "rgnn_stage_frames": (c_i32, [c_i64, c_vp, c_vp, c_vp, c_i64]),
}


class RgnnError(RuntimeError):
    pass


def _load() -> C.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP extension has not been built and radargnn_amd has no CPU fallback. "
            "Run `python -m radargnn_amd.build` (hipcc, gfx950).")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def check(rc: int) -> None:
    if rc != 0:
        raise RgnnError(f"librgnn error {rc}: {lib.rgnn_last_error().decode()}")
