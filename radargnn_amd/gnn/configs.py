"""Architecture description consumed by ``DetNetBasic`` -- field-compatible with the reference's
``gnn/configs.py:4-30`` (``GNNArchitectureConfig``) so YAML files, JSON dumps and positional construction
(``GNNArchitectureConfig(2, 3, [5], [3], [3], conv_layer_type=...)``, test/test_gnn.py:28-39) keep working."""
from dataclasses import dataclass
from typing import List, Optional


@dataclass
class GNNArchitectureConfig:
    # widths of the raw node / edge feature vectors entering the network
    node_feature_dimension: int
    edge_feature_dimension: int
    # output width of every graph-convolution layer, then the two heads (last entry = output width)
    conv_layer_dimensions: List[int]
    classification_head_layer_dimensions: List[int]
    regression_head_layer_dimensions: List[int]
    # optional embedding MLPs in front of the convolutions (last entry = embedded width)
    initial_node_feature_embedding: bool = False
    initial_edge_feature_embedding: bool = False
    node_feature_embedding_layer_dimensions: Optional[List[int]] = None
    edge_feature_embedding_layer_dimensions: Optional[List[int]] = None
    conv_layer_type: str = "MPNNConv"          # or "RadarPointGNNConv"
    # inside the MLPs / convolutions
    batch_norm_in_mlps: bool = True
    conv_pre_mlp_layer_number: int = 1
    conv_post_mlp_layer_number: int = 1
    conv_use_edge_encoder: bool = False
    aggregation_function: str = "max"          # "max" | "mean" | "add"
