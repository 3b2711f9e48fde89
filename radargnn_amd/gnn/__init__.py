"""Host-side mirror of ``gnnradarobjectdetection.gnn``: same class names, constructor signatures, attribute
layout and ``state_dict`` keys; the arithmetic runs in librgnn.so on the MI355X."""
from .configs import GNNArchitectureConfig  # noqa: F401
from .gnn_models import DetNetBasic, get_mlp  # noqa: F401
from .linear import BatchNorm, Linear, frame_scope  # noqa: F401
from .mpnn_layers import MPNNConv, RadarPointGNNConv  # noqa: F401
from .losses import detection_loss  # noqa: F401
